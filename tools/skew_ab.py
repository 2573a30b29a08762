"""A bench workload (SK_WORKLOAD: skewed = configs[4], hot-trigram haystack, limit 100; words; geonames) through each sweep of the library, rows compared with the plain sweep's.
python tools/skew_ab.py   (GPU box; SK_N needles = 100000, SK_LIMIT = 100)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

n_q = int(os.environ.get("SK_N", "100000")); limit = int(os.environ.get("SK_LIMIT", "100"))
wl = os.environ.get("SK_WORKLOAD", "skewed")
hay, off = W.bench_haystack(wl, 1.0)
n = len(off) - 1
q, qo = W.bench_needles(hay, off, wl, 1.0, 0, 1)
q, qo = q[:int(qo[n_q])], qo[:n_q + 1]
base = None
for name, opts in (("plain", dict(wsweep=0, nm_cmin=0, small_sweep=0)),
                   ("window-major", dict(ws_min_slice=0, ws_static_slice=0, ws_min_windows=1, small_sweep=0)),
                   ("slices left out", dict(wsweep=0, nm_min_windows=0, small_sweep=0)), ("small haystack", dict())):
    m = RawMap()
    m.set_option("ws_autotune", 0)
    for k, v in opts.items():
        m.set_option(k, v)
    for k in ("nm_cmin", "nm_dense"):
        if os.environ.get(k.upper()) and name == "slices left out":
            m.set_option(k, int(os.environ[k.upper()]))
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    m.sync_device()
    m.set_timing(True)
    ms = []
    for _ in range(3):
        rows, counts = m.find_batch_packed(q, qo, limit)
        ms.append(m.device_info()["last_find_kernel_ms"])
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    rows = np.where(live[:, :, None], rows, 0)
    if base is None:
        base = (rows, counts)
    same = bool(np.array_equal(counts, base[1]) and np.array_equal(rows, base[0]))
    m.set_timing(False); m.set_stats(True)
    m.find_batch_packed(q, qo, limit)
    st = m.find_stats()
    print(f"{name}: last_sweep {m.get_option('last_sweep')} kernel ms {' '.join(f'{x:.1f}' for x in ms)} rows==plain {same} "
          f"postings/needle {st['posting_entries'] / n_q:.0f} steps/needle {st['steps'] / n_q:.1f} resweeps {st['resweeps']} "
          f"compactions/needle {st['compactions'] / n_q:.2f} probes/needle {st.get('probes', 0) / n_q:.0f}", flush=True)
    m.close()
