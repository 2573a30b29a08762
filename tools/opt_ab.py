"""Same-process A/B of OPTION sets on one build of the library (BLURRILY_LIB): configs[2]'s haystack (or AB_WORKLOAD),
AB_N needles (300 000), the sweep forced by AB_SWEEP (3: dense slices left out), every set timed AB_REPS times in turn
(interleaved: set A, set B, set A, ...), rows' crc per set.     python tools/opt_ab.py "nm_pow2=0" "nm_pow2=1" ...   (GPU box)"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
n_q = int(os.environ.get("AB_N", "300000"))
sweep = int(os.environ.get("AB_SWEEP", "3"))
reps = int(os.environ.get("AB_REPS", "4"))
name = os.environ.get("AB_WORKLOAD", "geonames")
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name, 1.0)
n = len(off) - 1
m = RawMap()
m.set_option("ws_autotune", 0)
m.set_option("wsweep", 0)
m.set_option("small_sweep", 0)
m.set_option("nm_min_windows", 0 if sweep == 3 else 1 << 20)
if sweep == 1:
    m.set_option("nm_cmin", 0)
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.bench_needles(hay, off, name, 1.0, 0, 1)
if n_q < len(qo) - 1:
    q, qo = q[:int(qo[n_q])], qo[:n_q + 1]
n_q = len(qo) - 1
m.set_timing(True)
ms = [[] for _ in sets]
crc = [None] * len(sets)
crcs = [set() for _ in sets]
for r in range(reps):
    for i, st in enumerate(sets):
        for k, v in st.items():
            m.set_option(k, v)
        rows, counts = m.find_batch_packed(q, qo, limit)
        ms[i].append(m.device_info()["last_find_kernel_ms"])
        live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
        crc[i] = zlib.crc32(np.ascontiguousarray(np.where(live[:, :, None], rows, 0)).tobytes()) ^ zlib.crc32(counts.tobytes())
        crcs[i].add(crc[i])
tag = os.path.basename(os.environ.get("BLURRILY_LIB", "current"))
for i, st in enumerate(sets):
    print(f"{tag} {name} n={n_q} sweep {m.get_option('last_sweep')} {st}: crc {crc[i]:08x}{' UNSTABLE ' + ','.join(f'{c:08x}' for c in crcs[i]) if len(crcs[i]) > 1 else ''} kernel ms " +
          " ".join(f"{x:.1f}" for x in ms[i]) + f"  min {min(ms[i]):.1f}", flush=True)
