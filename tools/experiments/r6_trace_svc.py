#!/usr/bin/env python3
"""The resident grid's marks for single finds (trace build): when the workgroups see the command, what the steps take, when the rows are out."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 40, 7)
needles = W.unpack(q, qo)
lib = _native.lib(); lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rows = (_native.TrigramMatch * 10)()
G = m.device_info()["n_windows"]
names = ["seen", "ready", "table", "counted", "cnt barrier", "selected", "bisected", "compacted", "parts+fence", "flags seen", "m:loaded", "m:H+filter", "m:sorted", "m:rows", "m:sysfence"]
for mode in (0, 10):
    m.set_option("one_persistent", mode)
    acc = []
    for nd in needles:
        for _ in range(3): lib.blurrily_storage_find(m.handle, nd, 10, rows)
        t0 = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, 10, rows); host = (time.perf_counter() - t0) * 1e6
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
        t = buf[:G * 16].reshape(G, 16).astype(np.int64)
        base = t[:, 0].min()
        us = (t - base) / 100.0
        last = int(np.argmax(t[:, 10]))
        acc.append([np.median(us[:, i]) for i in range(10)] + [us[:, i].max() for i in range(10)] + [us[last, i] for i in range(10, 15)] + [host])
    a = np.median(np.array(acc), axis=0)
    print(f"one_persistent {mode}: microseconds from the first workgroup's mark 0 (median over {len(needles)} needles); host clock {a[25]:.1f} us")
    for i in range(10): print(f"  {names[i]:12s} median wg {a[i]:6.2f}   last wg {a[10 + i]:6.2f}")
    for i in range(10, 15): print(f"  {names[i]:12s} merging wg {a[10 + i]:6.2f}")
