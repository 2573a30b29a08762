mkdir -p gpurun_out/wsab
( for lib in "$@"; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/ws_ab.py; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/wsab/ws_ab.log
cat gpurun_out/wsab/ws_ab.log
