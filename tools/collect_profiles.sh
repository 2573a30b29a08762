#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round-tag>
# Produces under gpurun_out/<tag>/ everything profiles/ is refreshed from:
#   bench.json / bench.log             python bench.py (configs[2] with cpu_baseline + extra_configs)
#   stats/                             rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-extra`
#   stats_skewed/                      the same for --workload skewed (the window-major sweep's kernels)
#   pmc_{fetch,write,tcc}_<workload>/  rocprofv3 --pmc passes, one counter group per run, per workload
#   traffic.json                       tools/traffic_summary.py over those passes
# (PROFILE_COMMIT=<short hash> in the environment stamps traffic.json: the GPU box has no .git)
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py > $out/bench.json 2> $out/bench.log
# (--latency-probes 0: no single finds, no host-buffer batch -- whose chunks are launches of the same kernel --
#  so that the kernel's average duration in the summary is the timed steps' and the warm-up's)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $root/bench.py --no-cpu-baseline --no-extra --latency-probes 0 > $out/stats_bench.json 2> $out/stats_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_skewed -o bench -- python $root/bench.py --workload skewed --no-cpu-baseline --latency-probes 0 > $out/stats_skewed.json 2> $out/stats_skewed.log
for wl in geonames words skewed; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    d=$out/pmc_${name}_$wl
    mkdir -p $d
    rocprofv3 --pmc $ctrs --kernel-trace -d $d -o pmc -- python $root/bench.py --workload $wl --steps 1 --warmup 0 --static-choice --no-cpu-baseline --latency-probes 0 > $d/bench.json 2> $d/bench.log
  done
done
python $root/tools/traffic_summary.py $out > $out/traffic.json
find $out -name "*.db" -delete                      # keep the merge-back small
find $out -name "*_kernel_trace.csv" -size +5M -delete
ls -la $out $out/stats
