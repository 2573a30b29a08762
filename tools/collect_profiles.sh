#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round-tag>
# Produces under gpurun_out/<tag>/ everything profiles/ is refreshed from:
#   bench_{geonames,words,skewed}.json   python bench.py (geonames with the CPU baseline)
#   stats/                                rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline`
#   pmc_{fetch,write,tcc}/                rocprofv3 --pmc passes, one counter group per run
#   traffic.json                          tools/traffic_summary.py over the three PMC passes
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py > $out/bench_geonames.json 2> $out/bench_geonames.log
python $root/bench.py --workload words --no-cpu-baseline > $out/bench_words.json 2> $out/bench_words.log
python $root/bench.py --workload skewed --no-cpu-baseline > $out/bench_skewed.json 2> $out/bench_skewed.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $root/bench.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats_bench.log
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  mkdir -p $out/pmc_$name
  rocprofv3 --pmc $ctrs --kernel-trace -d $out/pmc_$name -o pmc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --latency-probes 0 > $out/pmc_$name/bench.json 2> $out/pmc_$name/bench.log
done
python $root/tools/traffic_summary.py $out > $out/traffic.json
find $out -name "*.db" -size +20M -delete     # keep the merge-back small
ls -la $out
