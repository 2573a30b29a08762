mkdir -p gpurun_out/r6e
( for lib in $LIBS; do BLURRILY_LIB=$PWD/blurrily_amd/$lib python tools/opt_ab.py "" ; done
) 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6e/ab.log
