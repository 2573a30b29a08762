#!/usr/bin/env python3
"""find_one_kernel's marks over a batch of MID_N needles (libx_midtr.so): when workgroups start, how long a step takes,
when the lists are out, when the merges end.  (GPU box)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
for k, v in (kv.split("=") for kv in os.environ.get("MID_OPTS", "").split(",") if kv):
    m.set_option(k, int(v))
lib = _native.lib(); lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
nw = m.device_info()["n_windows"]
for batch in [int(x) for x in os.environ.get("MID_N", "16 32 64 128").split()]:
    m.set_option("few_max", 128)
    acc = []
    for rep in range(8):
        q, qo = W.queries(hay, off, batch, 100 + rep)
        for _ in range(2): m.find_batch_packed(q, qo, 10)
        t0 = time.perf_counter(); m.find_batch_packed(q, qo, 10); host = (time.perf_counter() - t0) * 1e6
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
        t = buf.reshape(8192, 16).astype(np.int64)
        live = t[:, 0] != 0
        t = t[live]
        base = t[:, 0].min()
        us = (t - base) / 100.0
        us[t == 0] = np.nan
        merged = ~np.isnan(us[:, 14])
        acc.append([live.sum(), np.median(us[:, 0]), us[:, 0].max(), np.median(us[:, 1] - us[:, 0]), np.median(us[:, 7] - us[:, 1]),
                    np.nanmax(us[:, 7] - us[:, 1]), np.median(us[:, 8] - us[:, 7]), np.nanmax(us[:, 8]),
                    np.median(us[merged, 14] - us[merged, 9]), np.nanmax(us[:, 14]), host, merged.sum()])
    a = np.median(np.array(acc), axis=0)
    print(f"batch {batch}: {a[0]:.0f} workgroups ({nw} windows), kernels {'+'.join(m.last_kernels())}\n"
          f"  start: median {a[1]:.1f} last {a[2]:.1f} us;  zeroing {a[3]:.1f};  steps (all of a workgroup's): median {a[4]:.1f} max {a[5]:.1f};"
          f"  list out {a[6]:.1f} (last at {a[7]:.1f});  merges {a[11]:.0f}, each {a[8]:.1f}, last done at {a[9]:.1f};  host {a[10]:.1f} us", flush=True)
