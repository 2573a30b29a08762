"""configs[2]'s needles split by distinct-trigram count: those the 4-bit sweep serves whole (T <= 15, two windows a
step) against the rest (byte counters, one window a step, behind the 4-bit prefix of the windows).  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle

hay, off = W.bench_haystack("geonames", 1.0)
m = RawMap(); m.set_option("wsweep", 0)
m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 200000, 3000)
needles = W.unpack(q, qo)
T = np.array([len(set(Oracle.tokenise(nd))) for nd in needles])


def pack(lst):
    o = np.zeros(len(lst) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(x) for x in lst])
    return np.frombuffer(b"".join(lst), dtype=np.uint8), o


m.set_timing(True)
for label, sel in (("all", T >= 0), ("T <= 15", T <= 15), ("T > 15", T > 15)):
    lst = [nd for nd, s in zip(needles, sel) if s]
    p, o = pack(lst)
    ms = []
    for _ in range(3):
        m.find_batch_packed(p, o, 10)
        ms.append(m.device_info()["last_find_kernel_ms"])
    m.set_stats(True); m.find_batch_packed(p, o, 10); st = m.find_stats(); m.set_stats(False)
    print(f"{label:8s} {len(lst):7d} needles, mean T {T[sel].mean():.1f}: kernel ms {ms[-1]:.1f} = {ms[-1] * 1e3 / len(lst):.3f} us per needle x 512 workgroups; "
          f"steps per needle {st['steps'] / len(lst):.1f}, posting slots per needle {st['posting_entries'] / len(lst):.0f}", flush=True)
