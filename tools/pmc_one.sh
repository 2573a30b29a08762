#!/bin/bash
# usage (GPU box): tools/pmc_one.sh <tag> "<counters>" <scale> <needles>  -- one rocprofv3 --pmc pass over tools/ws_run.py
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace -d $out/p -o pmc -- python $GRAFT_REPO_ROOT/tools/ws_run.py $3 $4 1 > $out/p.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $out/p -name "*.db" | head -1) | grep "find_kernel<unsigned char, 1024, false, true>" | cut -c48-120
find $out -name "*.db" -delete
