"""Quick A/B of the window-major sweep against the needle-major one (run on the GPU box):
python tools/ws_probe.py [scale] [needles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
name = sys.argv[3] if len(sys.argv) > 3 else "geonames"
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name, scale)
n = len(off) - 1
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.set_option("ws_min_slice", 0)          # (before the image is built: it then carries the bitmaps)
m.set_option("ws_static_slice", 0); m.set_option("ws_autotune", 0)      # the sweep asked for, not the measured choice
m.sync_device()
print("windows", m.device_info()["n_windows"], "bytes", m.device_info()["device_bytes"], flush=True)
q, qo = W.queries(hay, off, nq, 3000)
m.set_option("ws_min_slice", 0)
res = {}
for label, opts in (("needle-major", {"wsweep": 0}), ("window-major cmin2", {"wsweep": 1, "ws_cmin": 2}),
                    ("window-major cmin1", {"ws_cmin": 1}), ("window-major cmin3", {"ws_cmin": 3})):
    for k, v in opts.items():
        m.set_option(k, v)
    m.find_batch_packed(q, qo, limit)
    m.set_timing(True)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); rows, counts = m.find_batch_packed(q, qo, limit); ts.append(time.perf_counter() - t)
    kms = m.device_info()["last_find_kernel_ms"]
    m.set_timing(False)
    m.set_stats(True); m.find_batch_packed(q, qo, limit); st = m.find_stats()
    import ctypes as C
    out16 = (C.c_uint64 * 16)()
    m._lib.blurrily_debug_find_stats16.argtypes = [C.c_void_p, C.c_void_p]
    m._lib.blurrily_debug_find_stats16(m.handle, out16)
    clk = [int(v) for v in out16[8:16]]
    tot = max(1, sum(clk))
    st["clock%"] = {k: round(100 * v / tot, 1) for k, v in zip(("filter", "setup", "cnt-barrier", "scan", "probe", "select", "skipsel", "units+count"), clk)}
    st["clk/task"] = round(tot / max(1, st["tasks"]))
    m.set_stats(False)
    res[label] = (rows.copy(), counts.copy())
    print(f"{label:22s} kernel {kms:9.2f} ms  {nq / (kms * 1e-3) / 1e6:7.3f} M needles/s  host {min(ts) * 1e3:8.1f} ms  {st}", flush=True)
a = res["needle-major"]
for k, b in res.items():
    live = np.arange(limit)[None, :] < a[1][:, None].astype(np.int64)
    same = np.array_equal(a[1], b[1]) and np.array_equal(np.where(live[:, :, None], a[0], 0), np.where(live[:, :, None], b[0], 0))
    print(k, "rows equal to needle-major:", same)
