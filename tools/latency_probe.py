#!/usr/bin/env python3
"""Where a single blurrily_storage_find spends its time (host clock around the C-ABI call and
HIP-event times of the kernels).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(8423769 * scale)
hay, off = W.geonames(n, max(1000, int(500000 * min(1.0, scale * 4))), 3)
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 300, 7)
needles = W.unpack(q, qo)
lib = _native.lib()
rows = (_native.TrigramMatch * 10)()
for timing in (False, True):
    m.set_timing(timing)
    lat, ker, tok = [], [], []
    for nd in needles:
        t = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, 10, rows); lat.append(time.perf_counter() - t)
        if timing:
            i = m.device_info(); ker.append(i["last_find_kernel_ms"] * 1e3); tok.append(i["last_tokenise_kernel_ms"] * 1e3)
    print(f"timing={timing}: p50 {np.median(lat)*1e6:.0f} us  p10 {np.percentile(lat,10)*1e6:.0f}  p90 {np.percentile(lat,90)*1e6:.0f}"
          + (f"   find kernels p50 {np.median(ker):.0f} us, tokenise p50 {np.median(tok):.0f} us" if timing else ""))
for batch in (1, 8, 64, 255, 256, 1024):
    qq, qqo = W.queries(hay, off, batch, 9)
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); m.find_batch_packed(qq, qqo, 10); t.append(time.perf_counter() - t0)
    print(f"batch {batch}: p50 {np.median(t)*1e6:.0f} us  -> {batch/np.median(t):.0f} needles/s")
