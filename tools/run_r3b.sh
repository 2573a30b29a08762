export BLURRILY_DIGESTS_PENDING=1
mkdir -p gpurun_out/r3b
( for lib in libblurrily_hip.so libblurrily_hip_scan4.so libblurrily_hip_load4.so libblurrily_hip_both.so libblurrily_hip.so; do BLURRILY_LIB=$PWD/blurrily_amd/$lib timeout 300 python tools/ab_probe.py; done ) > gpurun_out/r3b/ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_bench_batch.py::test_bench_n1_under_the_launcher_is_the_plain_n1_line tests/test_gpu_bench_batch.py::test_bench_two_ranks_plumbing -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>&1 | grep -i -B1 -A3 "dram\|hbm\|mall\|EA_RDREQ_DRAM\|_DRAM" | head -150) > gpurun_out/r3b/counters.log 2>&1
tail -3 gpurun_out/r3b/pytest.log; grep "kernel ms" gpurun_out/r3b/ab.log | awk '{print $1, $NF}' | sed 's/.*blurrily_amd.//'
