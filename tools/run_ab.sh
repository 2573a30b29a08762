# usage (GPU box): bash tools/run_ab.sh <tag> lib1.so lib2.so ...   -- tools/ab_probe.py for each build, twice round
tag=$1; shift
mkdir -p gpurun_out/$tag
( for r in 1 2; do for lib in "$@"; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/ab_probe.py; done; done ) > gpurun_out/$tag/ab.log 2>&1
grep "kernel ms" gpurun_out/$tag/ab.log | awk '{print $1, $NF}' | sed 's/.*blurrily_amd.//' | awk '{a[$1]=a[$1]" "sprintf("%.1f",$2)} END{for(k in a) print k, a[k]}'
