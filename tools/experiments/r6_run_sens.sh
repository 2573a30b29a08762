mkdir -p gpurun_out/r6c
( for lib in libx_r5.so libblurrily_hip.so libx_valu64.so libx_salu64.so libx_bar2.so libx_atom3.so libx_scan2.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
) 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6c/ab.log
