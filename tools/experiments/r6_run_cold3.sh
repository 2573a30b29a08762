mkdir -p gpurun_out/r6e
( AB_REPS=4 LIBS="libblurrily_hip.so libx_cold3.so libblurrily_hip.so libx_cold3.so" bash tools/experiments/r6_run_ab3.sh
  for wl in geonames_x4 skewed geonames_miss words; do for lib in libblurrily_hip.so libx_cold3.so; do AB_WORKLOAD=$wl AB_N=100000 AB_REPS=3 BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py ""; done; done
  export MID_N="32 64 128 224" MID_REPS=60
  for lib in libblurrily_hip.so libx_cold3.so; do echo "== $lib"; BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/mid_probe.py; done
) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6e/cold3.log; cat gpurun_out/r6e/cold3.log
