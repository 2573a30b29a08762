#!/bin/bash
# usage (GPU box, from the repo root): tools/lds_pipe.sh <tag>
# How busy the CU's pipes are under the kernels the bench line's batch rates come from: find_kernel<..., LEAVE> on
# configs[2], four times its haystack and configs[4] (sweep 3), find_small_kernel on configs[1] (sweep 4) --
# SQ_LDS_IDX_ACTIVE (every cycle the LDS array works: atomics, the scan's reads and clears, bank-conflict replays) and
# the VALU's issue slots (SQ_ACTIVE_INST_VALU: quad-cycles of a SIMD, four SIMDs a CU) over SQ_BUSY_CU_CYCLES, ONE
# timed step each, one rocprofv3 --pmc pass per counter group (no other trace domain beside --kernel-trace).
# Leaves gpurun_out/<tag>/lds_pipe.json (tools/lds_pipe_summary.py); PROFILE_COMMIT stamps it like traffic.json.
tag=${1:-lds}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in "geonames:3" "words:4" "geonames_x4:3" "skewed:3" "geonames_miss:3"; do
  name=${wl%%:*}; fs=${wl#*:}
  for pass in "idx:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES" "inst:SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "mix:SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "valu:SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "salu:SQ_INST_CYCLES_SALU"; do
    p=${pass%%:*}; ctrs=${pass#*:}
    d=$out/pmc_${p}_$name
    mkdir -p $d
    timeout 300 rocprofv3 --pmc $ctrs --kernel-trace -d $d -o pmc -- python $root/bench.py --workload $name --steps 1 --warmup 0 --force-sweep $fs --no-cpu-baseline --no-extra --latency-probes 0 --detail $d/detail.json > $d/bench.json 2> $d/bench.log
  done
done
python $root/tools/lds_pipe_summary.py $out > $out/lds_pipe.json
find $out -name "*.db" -delete
find $out -name "*_kernel_trace.csv" -size +2M -delete
cat $out/lds_pipe.json
