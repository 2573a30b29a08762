mkdir -p gpurun_out/r3l
GATE_PCTS=0,25,50,100 timeout 900 python tools/gate_probe.py 4000000 65536 262144 > gpurun_out/r3l/gate10.md 2> gpurun_out/r3l/gate10.jsonl
cat gpurun_out/r3l/gate10.md
