#!/bin/bash
# usage (GPU box): tools/pmc_step.sh <tag> <lib.so> ...   -- SQ instruction / activity counters of the needle-major
# find_kernel over ONE launch of 300 k Geonames-scale needles, one rocprofv3 --pmc pass per counter group and library
tag=$1; shift
for lib in "$@"; do
  for grp in ${PMC_GROUPS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY"}; do
    echo "## $(basename $lib): $grp"
    BLURRILY_LIB=$GRAFT_REPO_ROOT/blurrily_amd/$lib WSWEEP=0 bash $GRAFT_REPO_ROOT/tools/pmc_one.sh $tag/$(basename $lib .so) "$grp" 1.0 300000
  done
done
