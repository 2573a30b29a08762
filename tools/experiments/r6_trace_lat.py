#!/usr/bin/env python3
"""Latency mode's tasks on the device's wall clock (libx_lattr.so / libx_lattrn.so): when tasks start, what set-up, the
learning sweep, the range's sweep and the keys cost, when the last one ends; the host clock around the call.  (GPU box)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
m.set_option("few_max", 1); m.set_option("mid_max", 0)
lib = _native.lib(); lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for batch in [int(x) for x in os.environ.get("MID_N", "32 128").split()]:
    acc = []
    for rep in range(8):
        q, qo = W.queries(hay, off, batch, 100 + rep)
        for _ in range(2): m.find_batch_packed(q, qo, 10)
        t0 = time.perf_counter(); m.find_batch_packed(q, qo, 10); host = (time.perf_counter() - t0) * 1e6
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
        t = buf.reshape(8192, 16).astype(np.int64)
        t = t[t[:, 0] != 0]
        us = (t - t[:, 0].min()) / 100.0
        us[t == 0] = np.nan
        learned = ~np.isnan(us[:, 2])
        if rep == 7 and os.environ.get("LAT_DUMP"):
            dur = us[:, 4] - us[:, 0]
            info = buf.reshape(8192, 16)[:, 5][buf.reshape(8192, 16)[:, 0] != 0]
            order = np.argsort(-dur)[:24]
            print("  slowest tasks: us (learn, range) | T range own w0 w1 qs")
            for k in order:
                v = int(info[k])
                print(f"    {dur[k]:6.1f} ({(us[k, 2] - us[k, 1]) if learned[k] else 0:5.1f}, {us[k, 3] - (us[k, 2] if learned[k] else us[k, 1]):5.1f}) | T {v & 0xFF:3d} r {(v >> 8) & 0xFFF:3d} own {(v >> 20) & 1} w [{(v >> 24) & 0xFFF}, {(v >> 36) & 0xFFF}) qs {(v >> 48) & 0xFFF}")
            Tt = np.array([int(x) & 0xFF for x in info]); own = np.array([(int(x) >> 20) & 1 for x in info]); w0s = np.array([(int(x) >> 24) & 0xFFF for x in info])
            rng_t = us[:, 3] - np.where(learned, us[:, 2], us[:, 1])
            for name, sel in (("T<=15", Tt <= 15), ("T>15 nib part", (Tt > 15) & (w0s < 64)), ("T>15 byte part", (Tt > 15) & (w0s >= 64)), ("own range", own == 1)):
                if sel.any(): print(f"  {name}: {sel.sum()} tasks, range sweep median {np.median(rng_t[sel]):.1f} p90 {np.percentile(rng_t[sel], 90):.1f} max {rng_t[sel].max():.1f}")
        acc.append([len(t), np.median(us[:, 0]), us[:, 0].max(), np.median(us[:, 1] - us[:, 0]),
                    np.median(us[learned, 2] - us[learned, 1]) if learned.any() else 0.0, learned.mean(),
                    np.median(us[:, 3] - np.where(learned, us[:, 2], us[:, 1])), np.nanmax(us[:, 3] - np.where(learned, us[:, 2], us[:, 1])),
                    np.median(us[:, 4] - us[:, 3]), np.nanmax(us[:, 4]), np.median(us[:, 4] - us[:, 0]), np.nanmax(us[:, 4] - us[:, 0]), host])
    a = np.median(np.array(acc), axis=0)
    print(f"batch {batch}: {a[0]:.0f} tasks, kernels {'+'.join(m.last_kernels())}\n"
          f"  task start: median {a[1]:.1f} last {a[2]:.1f} us;  set-up {a[3]:.1f};  learning sweep {a[4]:.1f} ({a[5]:.2f} of the tasks);"
          f"  range sweep median {a[6]:.1f} max {a[7]:.1f};  keys {a[8]:.1f};  a task median {a[10]:.1f} max {a[11]:.1f};  last task done at {a[9]:.1f};  host {a[12]:.1f} us", flush=True)
