mkdir -p gpurun_out/wsab
timeout 900 python -m pytest tests/test_gpu_wsweep.py -x -q 2>&1 | tail -3
( for lib in libblurrily_hip_prev.so libblurrily_hip.so libblurrily_hip_prev.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/ws_ab.py; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/wsab/ws_ab.log
cat gpurun_out/wsab/ws_ab.log
