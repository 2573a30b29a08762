export BLURRILY_DIGESTS_PENDING=1
mkdir -p gpurun_out/r3a
( for lib in blurrily_amd/libblurrily_hip_old.so blurrily_amd/libblurrily_hip.so blurrily_amd/libblurrily_hip_old.so blurrily_amd/libblurrily_hip.so; do BLURRILY_LIB=$PWD/$lib timeout 300 python tools/ab_probe.py; done ) > gpurun_out/r3a/ab.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
timeout 300 python tools/phase_profile.py 1.0 100000 > gpurun_out/r3a/phase.log 2>&1
tail -5 gpurun_out/r3a/ab.log gpurun_out/r3a/pytest.log
