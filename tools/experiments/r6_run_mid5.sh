mkdir -p gpurun_out/r6e
timeout 1500 python -m pytest tests/test_gpu_find_one.py tests/test_gpu_parity.py tests/test_gpu_leave.py -x -q -m gpu 2>&1 | tail -3
( export MID_N="32 48 56 64 96 128" MID_REPS=60
  for lt in 0 1 2 3 4; do echo "== random sets, latency_tasks $lt"; MID_OPTS=latency_tasks=$lt python tools/mid_probe.py; done
  export MID_N="32 64 128" MID_REPS=20
  for lt in 0 2 3 4; do echo "== the bench's needles, latency_tasks $lt"; MID_FIXED=1 MID_OPTS=latency_tasks=$lt python tools/mid_probe.py; done
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/mid5.log
cat gpurun_out/r6e/mid5.log
