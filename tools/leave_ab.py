"""Same-box timing of ONE sweep of the needle-major kernel for a build of the library (BLURRILY_LIB): configs[2]'s
haystack, AB_N needles (300 000), AB_SWEEP 1 plain / 3 dense slices left out (default), AB_LIMIT (10).  With AB_CHECK=1
the rows are compared with those of the plain sweep of the same build.   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

n_q = int(os.environ.get("AB_N", "300000"))
sweep = int(os.environ.get("AB_SWEEP", "3"))
limit = int(os.environ.get("AB_LIMIT", "10"))
scale = float(os.environ.get("AB_SCALE", "1.0"))
hay, off = W.bench_haystack("geonames", scale)
n = len(off) - 1
m = RawMap()
m.set_option("ws_autotune", 0)
m.set_option("wsweep", 0)
m.set_option("nm_min_windows", 0 if sweep == 3 else 1 << 20)
for key in ("nm_cmin", "nm_dense"):
    if os.environ.get(key.upper()):
        m.set_option(key, int(os.environ[key.upper()]))
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
if os.environ.get("AB_MISS"):
    f_hay, f_off = W.geonames(200000, 500000, 1003)
    q, qo = W.queries(f_hay, f_off, n_q, 3000)
else:
    q, qo = W.queries(hay, off, n_q, 3000)
m.set_timing(True)
ms = []
for _ in range(4):
    rows, counts = m.find_batch_packed(q, qo, limit)
    ms.append(m.device_info()["last_find_kernel_ms"])
tag = os.path.basename(os.environ.get("BLURRILY_LIB", "current"))
import zlib
live_ = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
crc = zlib.crc32(np.ascontiguousarray(np.where(live_[:, :, None], rows, 0)).tobytes()) ^ zlib.crc32(counts.tobytes())
out = f"{tag} crc {crc:08x} sweep {m.get_option('last_sweep')} kernel ms " + " ".join(f"{x:.1f}" for x in ms) + f"  min {min(ms):.1f}"
if os.environ.get("AB_CHECK"):
    m.set_option("nm_min_windows", 1 << 20)
    rows1, counts1 = m.find_batch_packed(q, qo, limit)
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    same = bool(np.array_equal(counts, counts1) and np.array_equal(np.where(live[:, :, None], rows, 0), np.where(live[:, :, None], rows1, 0)))
    out += f"  rows==plain {same} (plain {m.device_info()['last_find_kernel_ms']:.1f} ms)"
print(out, flush=True)
