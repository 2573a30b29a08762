mkdir -p gpurun_out/r6b
( for lib in libx_r5.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
  BLURRILY_LIB=$PWD/blurrily_amd/libblurrily_hip.so python tools/opt_ab.py "nm_pow2=0" "nm_pow2=1" "nm_pow2=2" "nm_pow2=0,nm_cmin=4" "nm_pow2=0,nm_cmin=2" "nm_cmin=3"
  for lib in libx_noscan.so libx_noharvest.so libx_noatomics.so libx_nocount.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
  AB_WORKLOAD=geonames_x4 BLURRILY_LIB=$PWD/blurrily_amd/libblurrily_hip.so python tools/opt_ab.py "nm_pow2=0" "nm_pow2=1"
  AB_WORKLOAD=skewed BLURRILY_LIB=$PWD/blurrily_amd/libblurrily_hip.so python tools/opt_ab.py "nm_pow2=0" "nm_pow2=1"
) 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6b/ab.log
