#!/bin/bash
# usage (GPU box): tools/ws_pmc.sh <outdir> <scale> <needles>  -- PMC passes over tools/ws_run.py
out=$GRAFT_REPO_ROOT/gpurun_out/$1; scale=$2; nq=$3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace -d $out/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/ws_run.py $scale $nq 1 > $out/p$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $out/p$i -name "*.db" | head -1) > $out/p$i.txt 2>&1
  find $out/p$i -name "*.db" -delete
done
grep -h "wsweep\|find_kernel<unsigned char, 1024, false, true>" $out/p*.txt | cut -c1-200
