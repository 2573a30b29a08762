#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args...]   (run on the GPU box)
# One rocprofv3 --pmc pass of bench.py; leaves gpurun_out/pmc_<tag>/ with the counter CSVs.
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace -d $out -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --latency-probes 0 "$@" > $out/bench.json 2> $out/bench.log
echo "rc=$?"
