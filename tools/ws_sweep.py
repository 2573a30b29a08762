"""Tuning sweep of the window-major sweep (GPU box): dense-slice threshold x cmin, kernel ms.
python tools/ws_sweep.py <workload> <needles> [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

name = sys.argv[1]; nq = int(sys.argv[2]); scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name, scale)
n = len(off) - 1
q, qo = W.queries(hay, off, nq, 3000)
for dense in (512, 1024, 2048, 4096):
    m = RawMap()
    m.set_option("ws_min_slice", 0); m.set_option("ws_static_slice", 0); m.set_option("ws_autotune", 0)
    m.set_option("dense_min", dense)
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    m.sync_device()
    m.set_timing(True)
    for cmin in (2, 3, 4, 5):
        m.set_option("ws_cmin", cmin)
        m.find_batch_packed(q, qo, limit)
        m.find_batch_packed(q, qo, limit)
        print(f"{name} dense>={dense} cmin={cmin}: {m.device_info()['last_find_kernel_ms']:.1f} ms  index {m.device_info()['device_bytes'] / 1e6:.0f} MB", flush=True)
    m.close()
