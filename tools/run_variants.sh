# usage (GPU box): bash tools/run_variants.sh <tag> "<nm_probe grid>" lib1.so lib2.so ...  -- tools/nm_probe.py (300 k needles) per build
tag=$1; grid=$2; shift; shift
mkdir -p gpurun_out/$tag
for lib in "$@"; do
  echo "## $lib"
  BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/nm_probe.py ${NM_NEEDLES:-300000} 1.0 "$grid" 2>&1 | grep "^cmin" | sed 's/ postings.*rows==baseline/ rows==baseline/'
done > gpurun_out/$tag/variants.log 2>&1
cat gpurun_out/$tag/variants.log
