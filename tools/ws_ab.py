"""Same-box timing of the window-major sweep for one build of the library (BLURRILY_LIB): configs[4] at its own
limit, and the Geonames-scale haystack with the sweep forced; rows of the two builds are compared through a hash.
python tools/ws_ab.py   (GPU box)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

for name, nq in (("skewed", 100000), ("geonames", 300000)):
    limit = W.BENCH_WORKLOADS[name]["limit"]
    hay, off = W.bench_haystack(name)
    m = RawMap()
    m.set_option("ws_min_slice", 0); m.set_option("ws_static_slice", 0); m.set_option("ws_autotune", 0)
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    m.sync_device()
    q, qo = W.queries(hay, off, nq, 3000)
    m.find_batch_packed(q, qo, limit)
    m.set_timing(True)
    ms = []
    for _ in range(3):
        rows, counts = m.find_batch_packed(q, qo, limit)
        ms.append(m.device_info()["last_find_kernel_ms"])
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    h = hashlib.sha256(np.where(live[:, :, None], rows, 0).tobytes() + counts.tobytes()).hexdigest()[:12]
    print(os.path.basename(os.environ.get("BLURRILY_LIB", "current")), name, nq, "needles, limit", limit,
          "window-major kernel ms", " ".join(f"{x:.1f}" for x in ms), "rows", h, flush=True)
    m.close()
