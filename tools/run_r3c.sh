export BLURRILY_DIGESTS_PENDING=1
mkdir -p gpurun_out/r3c
timeout 900 python tools/gate_probe.py > gpurun_out/r3c/gate.md 2> gpurun_out/r3c/gate.jsonl
timeout 300 python tools/ws_probe.py 1.0 100000 skewed > gpurun_out/r3c/ws_skewed.log 2>&1
timeout 300 python tools/ws_probe.py 1.0 300000 geonames > gpurun_out/r3c/ws_geonames.log 2>&1
timeout 600 python -m pytest tests/test_gpu_bench_batch.py::test_bench_n1_under_the_launcher_is_the_plain_n1_line -x -q > gpurun_out/r3c/pytest.log 2>&1
cat gpurun_out/r3c/gate.md; tail -3 gpurun_out/r3c/pytest.log
