"""The needle-major sweep with slices left out of the count (options "nm_cmin", "nm_dense") on configs[2]'s haystack:
kernel ms, postings read, bitmap probes and steps per needle along a grid, every variant's rows compared with the
rows of nm_cmin 0 (nothing left out).    python tools/nm_probe.py [needles = 300000] [scale = 1.0] [grid]   (GPU box)
grid: "cmin:dense,cmin:dense,..." (default: a sweep)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
grid = [(0, 4096)] + [(c, d) for c in (2, 3, 4) for d in (1024, 2048, 4096, 8192)]
if len(sys.argv) > 3:
    grid = [(0, 4096)] + [tuple(int(x) for x in g.split(":")) for g in sys.argv[3].split(",")]
limit = int(os.environ.get("NM_LIMIT", "10"))
hay, off = W.bench_haystack("geonames", scale)
n = len(off) - 1
m = RawMap()
m.set_option("wsweep", 0)
if os.environ.get("NM_DENSE_MIN"):                    # which slices are dense at all (65536: none, no bitmaps in the image)
    m.set_option("dense_min", int(os.environ["NM_DENSE_MIN"]))
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
info = m.device_info()
print("strings", n, "windows", info["n_windows"], "bitmaps", info["n_bitmaps"], "image MB", info["device_bytes"] / 1e6, flush=True)
if os.environ.get("NM_MISS"):                         # needles of another vocabulary: no close match in the haystack
    f_hay, f_off = W.geonames(200000, 500000, 1003)
    q, qo = W.queries(f_hay, f_off, n_q, 3000)
else:
    q, qo = W.queries(hay, off, n_q, 3000)
base = None
for cmin, dense in grid:
    try:
        m.set_option("nm_cmin", cmin)
        m.set_option("nm_dense", dense)
    except OSError:                                   # (a build of the library from before round 4: BLURRILY_LIB)
        if cmin:
            continue
    m.set_timing(True)
    ms = []
    for _ in range(3):
        rows, counts = m.find_batch_packed(q, qo, limit)
        ms.append(m.device_info()["last_find_kernel_ms"])
    m.set_timing(False)
    m.set_stats(True)
    rows2, counts2 = m.find_batch_packed(q, qo, limit)
    st = m.find_stats()
    flags = m.find_path_flags(n_q)
    m.set_stats(False)
    st.setdefault("probes", 0)
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    rows = np.where(live[:, :, None], rows, 0)
    rows2 = np.where(live[:, :, None], rows2, 0)
    ok = bool(np.array_equal(counts, counts2) and np.array_equal(rows, rows2))
    if base is None:
        base = (rows, counts)
    same = bool(np.array_equal(counts, base[1]) and np.array_equal(rows, base[0]))
    bad = int((~((rows == base[0]).all(axis=(1, 2)) & (counts == base[1]))).sum()) if not same else 0
    print(f"cmin {cmin} dense {dense}: kernel ms {' '.join(f'{x:.1f}' for x in ms)}  postings/needle {st['posting_entries'] / n_q:.0f} "
          f"probes/needle {st['probes'] / n_q:.1f} steps/needle {st['steps'] / n_q:.1f} compactions/needle {st['compactions'] / n_q:.2f} "
          f"resweeps {st['resweeps']} walked {st['units']} left_out_needles {int(((flags >> 21) & 1).sum())} "
          f"counted==timed {ok} rows==baseline {same} (differing needles {bad})", flush=True)
