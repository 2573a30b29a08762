# usage (GPU box): bash tools/run_latency.sh <tag> [lib.so]  -- single-find latency breakdown + per-kernel trace of the single finds
tag=$1; mkdir -p gpurun_out/$tag
[ -n "$2" ] && export BLURRILY_LIB=$PWD/blurrily_amd/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/trace -o lat -- python $GRAFT_REPO_ROOT/tools/latency_probe.py 1.0 > $GRAFT_REPO_ROOT/gpurun_out/$tag/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/latency_probe.py 1.0 > gpurun_out/$tag/latency.log 2>&1
find gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats.csv \;
find gpurun_out/$tag/trace -name "*memory_copy_stats.csv" -exec cp {} gpurun_out/$tag/memory_copy_stats.csv \;
find gpurun_out/$tag/trace -name "*_trace.csv" -size +2M -delete
find gpurun_out/$tag/trace -name "*.db" -delete
cat gpurun_out/$tag/latency.log; cut -c1-200 gpurun_out/$tag/kernel_stats.csv; cat gpurun_out/$tag/memory_copy_stats.csv
