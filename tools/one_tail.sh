#!/bin/bash
# usage (GPU box, repo root): tools/one_tail.sh <tag> -- the single find's latency distribution, kernel time joined in
tag=${1:-one_tail}; root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for case in "geonames 10" "geonames 100" "geonames_x4 10" "skewed 100"; do
  set -- $case
  d=$out/$1_$2; mkdir -p $d
  rocprofv3 --kernel-trace -d $d -o t -- python $root/tools/one_tail.py $1 $2 $d/finds.json 2>&1 | grep -v amdgpu.ids | grep "p50"
  python $root/tools/one_tail.py --join $d/finds.json $(find $d -name "*.db" | head -1) >> $out/summary.md
  find $d -name "*.db" -delete
done
cat $out/summary.md
