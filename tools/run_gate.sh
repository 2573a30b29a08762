mkdir -p gpurun_out/gate
GATE_PCTS=0,25,33,40,50,75,100 timeout 1200 python tools/gate_probe.py 4000000 16384 65536 262144 1000000 > gpurun_out/gate/limit10.md 2> gpurun_out/gate/limit10.jsonl
GATE_LIMIT=100 GATE_PCTS=25,40,50,75,100 GATE_GEONAMES=0 timeout 900 python tools/gate_probe.py 4000000 16384 100000 > gpurun_out/gate/limit100.md 2> gpurun_out/gate/limit100.jsonl
cat gpurun_out/gate/limit10.md gpurun_out/gate/limit100.md
