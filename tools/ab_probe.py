import os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np, ctypes as C
import workloads as W
from blurrily_amd import RawMap
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, 500000, 3000)
m.set_timing(True)
for _ in range(4):
    m.find_batch_packed(q, qo, 10)
    print(os.environ.get("BLURRILY_LIB", "current"), "kernel ms", m.device_info()["last_find_kernel_ms"], flush=True)
