#!/usr/bin/env python3
"""N single blurrily_storage_find calls on the configs[2] haystack (what tools/run_latency.sh traces): python
tools/single_find_loop.py [n_finds] [limit] [workload]; prints the host clock's p50.   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
n_finds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = sys.argv[3] if len(sys.argv) > 3 else "geonames"
hay, off = W.bench_haystack(wl, 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, n_finds, 7)
needles = W.unpack(q, qo)
lib = _native.lib()
rows = (_native.TrigramMatch * limit)()
lat = []
for nd in needles:
    t = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, limit, rows); lat.append(time.perf_counter() - t)
print(f"{wl}: {n_finds} single finds, limit {limit}: p50 {np.median(lat)*1e6:.1f} us  p10 {np.percentile(lat,10)*1e6:.1f}  p90 {np.percentile(lat,90)*1e6:.1f}; one launch each: {m.get_option('one_taken')}")
