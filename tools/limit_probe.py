"""The needle-major sweep on configs[2]'s haystack by limit, for one build of the library (BLURRILY_LIB): kernel ms of
200 k needles.  python tools/limit_probe.py [limit ...]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap()
m.set_option("wsweep", 0)
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, 200000, 3000)
m.set_timing(True)
for limit in [int(x) for x in sys.argv[1:]] or [1, 10, 16, 30, 60, 100]:
    ms = []
    for _ in range(3):
        m.find_batch_packed(q, qo, limit)
        ms.append(m.device_info()["last_find_kernel_ms"])
    print(os.path.basename(os.environ.get("BLURRILY_LIB", "current")), "limit", limit, "kernel ms", " ".join(f"{x:.1f}" for x in ms[1:]), flush=True)
