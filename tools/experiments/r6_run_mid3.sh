mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests/test_gpu_find_one.py tests/test_gpu_normalise.py tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | tail -8
( export MID_N="8 16 17 24 25 28 32 40 48 56 64 96 128 129" MID_REPS=60
  echo "== defaults"; python tools/mid_probe.py
  echo "== mid_max 0 (the batch's way beyond few_max)"; MID_OPTS=mid_max=0 python tools/mid_probe.py
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/mid3.log
cat gpurun_out/r6e/mid3.log
