root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r06c; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $root/bench.py --force-sweep 3 --no-cpu-baseline --no-extra --latency-probes 0 --detail $out/stats_bench_detail.json > $out/stats_bench.json 2> $out/stats_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_skewed -o bench -- python $root/bench.py --workload skewed --force-sweep 3 --no-cpu-baseline --latency-probes 0 --detail $out/stats_skewed_detail.json > $out/stats_skewed.json 2> $out/stats_skewed.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x4 -o bench -- python $root/bench.py --workload geonames_x4 --force-sweep 3 --no-cpu-baseline --latency-probes 0 --detail $out/stats_x4_detail.json > $out/stats_x4.json 2> $out/stats_x4.log
find $out -name "*_kernel_trace.csv" -size +5M -delete
head -3 $out/stats/bench_kernel_stats.csv | cut -c1-150; cat $out/stats_bench.json | cut -c1-400
