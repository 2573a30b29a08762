#!/usr/bin/env python3
"""Host-buffer batches of 17 .. 128 needles (a server's coalesced FINDs) on configs[2]'s haystack: host clock around
blurrily_storage_find_batch, p50 over MID_REPS calls.   MID_N="17 32 64 128" python tools/mid_probe.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
hay, off = W.bench_haystack(os.environ.get("MID_WORKLOAD", "geonames"), 1.0)
n = len(off) - 1
limit = int(os.environ.get("MID_LIMIT", "10"))
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
for k, v in (kv.split("=") for kv in os.environ.get("MID_OPTS", "").split(",") if kv):
    m.set_option(k, int(v))
reps = int(os.environ.get("MID_REPS", "40"))
for batch in [int(x) for x in os.environ.get("MID_N", "8 16 17 32 64 128 256").split()]:
    t = []
    sets = [W.queries(hay, off, batch, 100 + rep) for rep in range(reps + 3)]      # (generated ahead: the calls follow each other)
    if os.environ.get("MID_FIXED"):                                                 # bench.py's way: the step batch's first needles, again and again
        bq, bo = W.bench_needles(hay, off, os.environ.get("MID_WORKLOAD", "geonames"), 1.0, 0, 1)
        sets = [(bq, np.ascontiguousarray(bo[:batch + 1]))] * (reps + 3)
    for rep in range(reps + 3):
        q, qo = sets[rep]
        t0 = time.perf_counter(); rows, counts = m.find_batch_packed(q, qo, limit); dt = time.perf_counter() - t0
        if rep >= 3:
            t.append(dt)
    print(f"batch {batch:4d}: p50 {np.median(t) * 1e6:7.1f} us  p90 {np.percentile(t, 90) * 1e6:7.1f} us  kernels {'+'.join(m.last_kernels())}", flush=True)
