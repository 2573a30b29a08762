#!/usr/bin/env python3
"""The resident single-find grid (option "one_persistent") against the launch per find: same rows for every needle, the host
clock's p50 / p99 of both, and what ends and restarts the grid.   python tools/svc_probe.py [n_finds] [limit] [workload]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
n_finds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = sys.argv[3] if len(sys.argv) > 3 else "geonames"
hay, off = W.bench_haystack(wl, 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, n_finds, 7)
needles = W.unpack(q, qo)
lib = _native.lib()
def run(tag):
    rows = (_native.TrigramMatch * limit)()
    lat, out = [], []
    for nd in needles[:5]: lib.blurrily_storage_find(m.handle, nd, limit, rows)
    for nd in needles:
        t = time.perf_counter(); k = lib.blurrily_storage_find(m.handle, nd, limit, rows); lat.append(time.perf_counter() - t)
        out.append([(rows[i].reference, rows[i].matches, rows[i].weight) for i in range(k)])
    lat = np.array(lat) * 1e6
    print(f"{tag}: p50 {np.median(lat):.1f} us  p10 {np.percentile(lat, 10):.1f}  p90 {np.percentile(lat, 90):.1f}  p99 {np.percentile(lat, 99):.1f}", flush=True)
    return out
a = run("a launch per find ")
m.set_option("one_persistent", limit)
b = run("the resident grid  ")
print("rows equal:", a == b, " answered by the grid:", m.get_option("one_service_taken"), flush=True)
rows, counts = m.find_batch_packed(q, qo[:301], limit)          # a batch ends the grid ...
c = run("... after a batch  ")                                   # ... and the next single find starts it again
print("rows equal:", a == c, " answered by the grid:", m.get_option("one_service_taken"), flush=True)
m.set_option("one_persistent", 0)
d = run("a launch again     ")
print("rows equal:", a == d)
m.close()
print("closed")
