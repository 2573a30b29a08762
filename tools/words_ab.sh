# usage (GPU box): bash tools/words_ab.sh <tag> lib1.so lib2.so ...  -- configs[1] (bench.py --workload words, 20 steps) for each build of the library
tag=$1; shift; mkdir -p gpurun_out/$tag
for lib in "$@"; do
  for rep in 1 2; do
    BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python bench.py --workload words --no-extra --no-cpu-baseline --latency-probes 0 --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],3),'ms')"
  done
done | tee gpurun_out/$tag/ab.txt
