# usage (GPU box): bash tools/run_latency.sh <tag>  -- 1 000 single finds under rocprofv3 --kernel-trace --memory-copy-trace, and what
# the trace holds from the first find on: kernels and copies per find (gpurun_out/<tag>/latency_trace.csv)
tag=$1; mkdir -p gpurun_out/$tag
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $root/gpurun_out/$tag/trace -o lat -- python $root/tools/single_find_loop.py 1000 10 > $root/gpurun_out/$tag/trace.log 2>&1
cd $root
python - $tag <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
kt = glob.glob(f"gpurun_out/{tag}/trace/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(f"gpurun_out/{tag}/trace/**/*memory_copy_trace.csv", recursive=True)
rows = list(csv.DictReader(open(kt)))
first = min(int(r["Start_Timestamp"]) for r in rows if "find_one_kernel" in r["Kernel_Name"])
by = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) >= first: by[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
copies = [r for r in (csv.DictReader(open(mc[0])) if mc else []) if int(r["Start_Timestamp"]) >= first]
n_one = len(by[[k for k in by if "find_one_kernel" in k][0]])
with open(f"gpurun_out/{tag}/latency_trace.csv", "w") as f:
    f.write("what,calls_from_the_first_single_find_on,per_find,average_ns\n")
    for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])): f.write(f"\"kernel {k}\",{len(v)},{len(v)/n_one:.3f},{sum(v)/len(v):.0f}\n")
    f.write(f"\"memory copies (any direction)\",{len(copies)},{len(copies)/n_one:.3f},\n")
print(open(f"gpurun_out/{tag}/latency_trace.csv").read())
PY
grep "single finds" gpurun_out/$tag/trace.log
python tools/single_find_loop.py 1000 10 | grep "single finds"
python tools/single_find_loop.py 1000 100 | grep "single finds"
rm -rf gpurun_out/$tag/trace
