"""The window-major sweep on configs[4] by `ws_cmin` (GPU box): python tools/ws_cmin_probe.py [cmin ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

name, nq = "skewed", 100000
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name)
m = RawMap()
m.set_option("ws_min_slice", 0); m.set_option("ws_static_slice", 0); m.set_option("ws_autotune", 0)
m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, nq, 3000)
for cmin in [int(x) for x in sys.argv[1:]] or [2, 3, 4, 5, 6]:
    m.set_option("ws_cmin", cmin)
    m.find_batch_packed(q, qo, limit)
    m.set_timing(True)
    ms = []
    for _ in range(3):
        m.find_batch_packed(q, qo, limit)
        ms.append(m.device_info()["last_find_kernel_ms"])
    m.set_timing(False)
    print("ws_cmin", cmin, "kernel ms", " ".join(f"{x:.1f}" for x in ms), flush=True)
