mkdir -p gpurun_out/r6d
( for lib in libx_r5.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
  for lib in libx_r5.so libblurrily_hip.so; do AB_SWEEP=1 BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
  for wl in geonames_x4 skewed geonames_miss; do for lib in libx_r5.so libblurrily_hip.so; do AB_WORKLOAD=$wl BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done; done
) 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6d/ab.log
