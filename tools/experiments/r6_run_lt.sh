mkdir -p gpurun_out/r6e
( export BLURRILY_LIB=$PWD/blurrily_amd/libx_lt.so MID_N="17 24 32 48 64 96 128" MID_REPS=60
  echo "== find_few up to 128"; MID_OPTS=few_max=128 python tools/mid_probe.py
  for lt in 1 2 4; do echo "== latency mode, $lt task(s) per workgroup"; BLURRILY_LT=$lt MID_OPTS=few_max=1 python tools/mid_probe.py; done
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6e/lt.log
cat gpurun_out/r6e/lt.log
