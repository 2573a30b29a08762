#!/usr/bin/env python3
"""The single find's own launch (c_abi.hip: find_one) against the batch's way on the same map: rows compared needle by
needle for several limits, then the host clock around blurrily_storage_find both ways.  python tools/one_probe.py
[workload] [scale]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native

wl = sys.argv[1] if len(sys.argv) > 1 else "geonames"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
hay, off = W.bench_haystack(wl, scale)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 400, 7)
needles = W.unpack(q, qo)
lib = _native.lib()
bad = 0
for limit in (1, 10, 64, 100, 120, 121):
    rows = (_native.TrigramMatch * limit)()
    for nd in needles[:200]:
        got = []
        for one in (1, 0):
            m.set_option("one_launch", one)
            c = lib.blurrily_storage_find(m.handle, nd, limit, rows)
            got.append([(rows[i].reference, rows[i].matches, rows[i].weight) for i in range(c)])
        if got[0] != got[1]:
            bad += 1
            if bad < 5: print("MISMATCH", nd, limit, got[0][:3], got[1][:3])
print(f"{wl} x{scale}: {n} strings, windows {m.device_info()['n_windows']}, mismatches {bad}, taken {m.get_option('one_taken')}")
rows = (_native.TrigramMatch * 10)()
for limit in (10, 100):
    rows = (_native.TrigramMatch * limit)()
    for one in (1, 0, 1, 0):
        m.set_option("one_launch", one)
        lat = []
        for nd in needles:
            t = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, limit, rows); lat.append(time.perf_counter() - t)
        print(f"limit {limit} one_launch={one}: p50 {np.median(lat)*1e6:.1f} us  p10 {np.percentile(lat,10)*1e6:.1f}  p90 {np.percentile(lat,90)*1e6:.1f}")
