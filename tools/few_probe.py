#!/usr/bin/env python3
"""Host-buffer batches of a handful of needles on configs[2]'s haystack: one launch shared (find_few) against latency mode
(option "one_launch" 0).   python tools/few_probe.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
for batch in (1, 2, 4, 8, 12, 16, 17, 32, 64):
    out = []
    for one in (1, 0):
        m.set_option("one_launch", one)
        t = []
        for rep in range(40):
            q, qo = W.queries(hay, off, batch, 100 + rep)
            t0 = time.perf_counter(); rows, counts = m.find_batch_packed(q, qo, 10); t.append(time.perf_counter() - t0)
        out.append(np.median(t) * 1e6)
    print(f"batch {batch:3d}: shared launch {out[0]:7.1f} us   latency mode {out[1]:7.1f} us")
