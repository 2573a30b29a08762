#!/bin/bash
# usage (GPU box, repo root): tools/pmc_insts.sh <tag> [lib.so ...] -- SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS of the dominant
# kernel over ONE step of configs[2] (1 M needles, sweep 3) per build of the library, one rocprofv3 --pmc pass
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "${@:-libblurrily_hip.so}"; do
  d=$out/insts_$lib; mkdir -p $d
  BLURRILY_LIB=$root/blurrily_amd/$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $d -o pmc -- \
    python $root/bench.py --workload geonames --steps 1 --warmup 0 --force-sweep 3 --no-cpu-baseline --no-extra --latency-probes 0 --detail $d/detail.json > $d/bench.json 2> $d/bench.log
  python - "$d" "$lib" <<'P'
import glob, os, sqlite3, sys, json
d, lib = sys.argv[1], sys.argv[2]
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
tot = {}
for k, n, v, disp in rows:
    if "find_kernel<unsigned char, 1024, false, true>" in k and "counted::" not in k:
        tot[n] = tot.get(n, 0) + v; tot["_disp"] = disp
steps = json.load(open(os.path.join(d, "detail.json")))["roofline"]["counters"]["steps"]
disp = tot.pop("_disp", 1)
print(lib, "dispatches", disp, "steps/launch", steps, " per step:", {k: round(v / disp / steps, 1) for k, v in tot.items()})
P
  find $d -name "*.db" -delete
done 2>&1 | grep -v amdgpu.ids | tee $out/insts.log
