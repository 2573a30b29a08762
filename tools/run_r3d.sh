export BLURRILY_DIGESTS_PENDING=1
mkdir -p gpurun_out/r3d
( for lib in libblurrily_hip_prev.so libblurrily_hip.so libblurrily_hip_prev.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/ws_ab.py; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3d/ws_ab.log
timeout 900 python -m pytest tests/test_gpu_wsweep.py -x -q > gpurun_out/r3d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/pytest.log
cat gpurun_out/r3d/ws_ab.log; tail -3 gpurun_out/r3d/pytest.log
