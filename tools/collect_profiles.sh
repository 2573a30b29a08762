#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round-tag>
# Produces under gpurun_out/<tag>/ everything profiles/ is refreshed from:
#   bench.json / bench.log             python bench.py (configs[2] with cpu_baseline + extra_configs)
#   stats/                             rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-extra`
#   stats_skewed/, stats_x4/           the same for --workload skewed (the window-major sweep's kernels) and geonames_x4 (the
#                                      needle-major sweep that leaves slices out: an image seven times the Infinity Cache)
#   pmc_{fetch,write,tcc}_<workload>/  rocprofv3 --pmc passes, one counter group per run, per workload
#   traffic.json                       tools/traffic_summary.py over those passes
# (PROFILE_COMMIT=<short hash> in the environment stamps traffic.json: the GPU box has no .git)
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py --detail $out/bench_detail.json > $out/bench.json 2> $out/bench.log   # (bench.json: the short line; bench_detail.json: the record)
# (--latency-probes 0: no single finds, no host-buffer batch -- whose chunks are launches of the same kernel --
#  so that the kernel's average duration in the summary is the timed steps' and the warm-up's; --force-sweep 3: the sweep
#  the bench run's measurement picks, without the measurement -- since round 6 the plain sweep and the one that leaves
#  slices out are ONE kernel symbol, and the measurement's plain runs would be averaged in)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $root/bench.py --force-sweep 3 --no-cpu-baseline --no-extra --latency-probes 0 --detail $out/stats_bench_detail.json > $out/stats_bench.json 2> $out/stats_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_skewed -o bench -- python $root/bench.py --workload skewed --force-sweep 3 --no-cpu-baseline --latency-probes 0 --detail $out/stats_skewed_detail.json > $out/stats_skewed.json 2> $out/stats_skewed.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x4 -o bench -- python $root/bench.py --workload geonames_x4 --force-sweep 3 --no-cpu-baseline --latency-probes 0 --detail $out/stats_x4_detail.json > $out/stats_x4.json 2> $out/stats_x4.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_words -o bench -- python $root/bench.py --workload words --no-cpu-baseline --latency-probes 0 --detail $out/stats_words_detail.json > $out/stats_words.json 2> $out/stats_words.log
# the sweep the bench run above took for a workload (its measured choice): the PMC passes force the same one
sweep_of() { python - "$out/bench_detail.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); wl = sys.argv[2]
r = (d if wl == "geonames" else d["extra_configs"][wl])["roofline"]["sweep"]
print(3 if r.startswith("needle-major, dense") else 2 if r.startswith("window") else 4 if r.startswith("small") else 1)
PY
}
for wl in geonames words skewed geonames_x4 geonames_miss; do
  fs=$(sweep_of $wl)
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    d=$out/pmc_${name}_$wl
    mkdir -p $d
    rocprofv3 --pmc $ctrs --kernel-trace -d $d -o pmc -- python $root/bench.py --workload $wl --steps 1 --warmup 0 --force-sweep ${fs:-0} --no-cpu-baseline --latency-probes 0 --detail $d/detail.json > $d/bench.json 2> $d/bench.log
  done
done
python $root/tools/traffic_summary.py $out > $out/traffic.json
find $out -name "*.db" -delete                      # keep the merge-back small
find $out -name "*_kernel_trace.csv" -size +5M -delete
ls -la $out $out/stats
