#!/usr/bin/env python3
"""Timeline of find_one_kernel from the TRACE build (BLURRILY_LIB=.../libblurrily_hip_tr.so): the device's 100 MHz wall
clock at sixteen marks per workgroup, over single finds of the Geonames-scale haystack.  Prints, in microseconds from
the first workgroup's start: when workgroups start / reach each mark (median over workgroups, and the last one), and
the merging workgroup's marks.   (GPU box)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
wl = sys.argv[1] if len(sys.argv) > 1 else "geonames"
hay, off = W.bench_haystack(wl, 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 64, 7)
needles = W.unpack(q, qo)
lib = _native.lib(); lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
limit = int(os.environ.get("LIMIT", "10"))
rows = (_native.TrigramMatch * limit)()
G = m.device_info()["n_windows"]
names = ["start", "zeroed", "table", "counted", "cnt barrier", "selected", "bisected", "compacted", "parts+fence", "ticket",
         "m:loaded", "m:H+filter", "m:sorted", "m:rows", "m:sysfence"]
acc = []
for nd in needles:
    for _ in range(3): lib.blurrily_storage_find(m.handle, nd, limit, rows)     # warm
    t0 = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, limit, rows); host = (time.perf_counter() - t0) * 1e6
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
    t = buf[:G * 16].reshape(G, 16).astype(np.int64)
    base = t[:, 0].min()
    us = (t - base) / 100.0
    last = int(np.argmax(t[:, 10]))                      # the merging workgroup wrote mark 10
    row = [np.median(us[:, i]) for i in range(10)] + [us[:, i].max() for i in range(10)] + [us[last, i] for i in range(10, 15)] + [host] + [us[last, 15]]
    acc.append(row)
a = np.median(np.array(acc), axis=0)
print(f"{wl}: {G} workgroups, limit {limit}; microseconds from the first workgroup's start (median over {len(needles)} needles)")
for i in range(10): print(f"  {names[i]:12s} median wg {a[i]:6.2f}   last wg {a[10 + i]:6.2f}")
for i in range(10, 15): print(f"  {names[i]:12s} merging wg {a[10 + i]:6.2f}")
print(f"  m:gathered   merging wg {a[26]:6.2f}   (between m:H+filter and m:sorted)")
print(f"  host clock around the call {a[25]:.1f} us")
