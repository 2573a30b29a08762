"""Run the window-major sweep a few times (for rocprofv3): python tools/ws_run.py [scale] [needles] [reps] [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
name = sys.argv[4] if len(sys.argv) > 4 else "geonames"
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name, scale)
n = len(off) - 1
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.set_option("ws_min_slice", int(os.environ.get("WS_MIN_SLICE", "0")))
m.set_option("ws_static_slice", 0); m.set_option("ws_autotune", 0)
m.set_option("wsweep", int(os.environ.get("WSWEEP", "1")))     # WSWEEP=0: the needle-major sweep
for key in ("nm_cmin", "nm_dense"):                               # NM_CMIN=0: nothing left out of the needle-major count
    if os.environ.get(key.upper()):
        try:
            m.set_option(key, int(os.environ[key.upper()]))
        except OSError:                                           # (a build from before round 4)
            pass
m.sync_device()
q, qo = W.queries(hay, off, nq, 3000)
m.set_timing(True)
for _ in range(reps):
    m.find_batch_packed(q, qo, limit)
    print("kernel ms", m.device_info()["last_find_kernel_ms"], flush=True)
