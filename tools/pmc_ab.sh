#!/bin/bash
# usage (GPU box): tools/pmc_ab.sh <tag> <lib.so> ...  -- instruction mix, instruction-cache and wait counters of the
# needle-major find_kernel (NM_CMIN from the environment) over one launch of 300 k Geonames-scale needles, per library
tag=$1; shift
for lib in "$@"; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    echo "## $(basename $lib): $grp"
    BLURRILY_LIB=$GRAFT_REPO_ROOT/blurrily_amd/$lib WSWEEP=0 bash $GRAFT_REPO_ROOT/tools/pmc_one.sh $tag/$(basename $lib .so) "$grp" 1.0 300000
  done
done
