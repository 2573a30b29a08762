mkdir -p gpurun_out/r3k
GATE_PCTS=25,33,40,50 GATE_GEONAMES=0 timeout 600 python tools/gate_probe.py 4000000 16384 65536 262144 > gpurun_out/r3k/gate10.md 2> gpurun_out/r3k/gate10.jsonl
GATE_LIMIT=100 GATE_PCTS=25,40,50,100 GATE_GEONAMES=0 timeout 600 python tools/gate_probe.py 4000000 16384 100000 > gpurun_out/r3k/gate100.md 2> gpurun_out/r3k/gate100.jsonl
cat gpurun_out/r3k/gate10.md gpurun_out/r3k/gate100.md
bash tools/pmc_step.sh r3k/pmc libblurrily_hip_old.so libblurrily_hip.so > gpurun_out/r3k/pmc.txt 2>&1
cat gpurun_out/r3k/pmc.txt
timeout 600 python -m pytest tests/test_gpu_wsweep.py -k gate -x -q 2>&1 | tail -3
