mkdir -p gpurun_out/r6e
timeout 1500 python -m pytest tests/test_gpu_find_one.py tests/test_gpu_parity.py tests/test_gpu_frontend.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
( export MID_N="25 28 32 40 48 56 64 96 128" MID_REPS=60
  echo "== defaults"; python tools/mid_probe.py
  echo "== trace"; MID_N="32 128" BLURRILY_LIB=$PWD/blurrily_amd/libx_lattr.so python tools/experiments/r6_trace_lat.py
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/mid4.log
cat gpurun_out/r6e/mid4.log
