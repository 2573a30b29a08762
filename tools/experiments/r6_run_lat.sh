mkdir -p gpurun_out/r6e
( for lib in libx_lattr.so libx_lattrn.so; do for lt in "" 1 2 4; do
    echo "== $lib BLURRILY_LT=$lt"; if [ -z "$lt" ]; then unset BLURRILY_LT; else export BLURRILY_LT=$lt; fi; BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/experiments/r6_trace_lat.py
  done; done ) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/trace_lat.log
cat gpurun_out/r6e/trace_lat.log
