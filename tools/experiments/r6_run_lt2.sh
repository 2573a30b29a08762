mkdir -p gpurun_out/r6e
( export MID_N="17 20 24 28 32 40 48 56 64 96 128" MID_REPS=60
  echo "== find_few up to 128"; MID_OPTS=few_max=128 python tools/mid_probe.py
  echo "== latency mode"; MID_OPTS=few_max=1 python tools/mid_probe.py
  echo "== latency mode (again)"; MID_OPTS=few_max=1 python tools/mid_probe.py
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/lt2.log
cat gpurun_out/r6e/lt2.log
timeout 900 python -m pytest tests/test_gpu_find_one.py tests/test_gpu_parity.py tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | tail -3
