# usage (GPU box): bash tools/run_leave_ab.sh <tag> lib1.so lib2.so ...   -- tools/leave_ab.py for each build (environment passed through)
tag=$1; shift
mkdir -p gpurun_out/$tag
( for lib in "$@"; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/leave_ab.py; done ) > gpurun_out/$tag/ab.log 2>&1
grep "kernel ms\|Error\|error" gpurun_out/$tag/ab.log
