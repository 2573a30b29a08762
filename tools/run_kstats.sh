# usage (GPU box): bash tools/run_kstats.sh <tag> <lib.so|-> <script.py> [env assignments...]  -- a probe script under rocprofv3 --kernel-trace --stats
tag=$1; lib=$2; script=$3; shift; shift; shift
mkdir -p gpurun_out/$tag
[ "$lib" != "-" ] && export BLURRILY_LIB=$PWD/blurrily_amd/$lib
for kv in "$@"; do export "$kv"; done
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/trace -o k -- python $root/$script > $root/gpurun_out/$tag/out.log 2>&1
cd $root
find gpurun_out/$tag/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats.csv \;
rm -rf gpurun_out/$tag/trace
grep -v amdgpu.ids gpurun_out/$tag/out.log | tail -5; cut -d, -f1-4 gpurun_out/$tag/kernel_stats.csv | cut -c1-150
