"""Does the ORDER of the needles in a batch matter to the needle-major sweep?  (GPU box)  Workgroups take needles in
batch order; a sweep starts at the window of the needle's own length class, so a batch sorted by length has the
workgroups that run together sweep the same windows at about the same time (shared slices then meet in the L2).
python tools/order_probe.py   -- configs[2]'s haystack, 500 k needles: as generated, by length descending, ascending"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, 500000, 3000)
needles = W.unpack(q, qo)


def pack(lst):
    o = np.zeros(len(lst) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(x) for x in lst])
    return np.frombuffer(b"".join(lst), dtype=np.uint8), o


m.set_timing(True)
for label, lst in (("as generated", needles), ("by length, descending", sorted(needles, key=len, reverse=True)),
                   ("by length, ascending", sorted(needles, key=len)),
                   ("by length descending in blocks of 4096", None)):
    if lst is None:
        lst = []
        for i in range(0, len(needles), 4096):
            lst += sorted(needles[i:i + 4096], key=len, reverse=True)
    p, o = pack(lst)
    ms = []
    for _ in range(3):
        m.find_batch_packed(p, o, 10)
        ms.append(m.device_info()["last_find_kernel_ms"])
    print(f"{label:42s} kernel ms", " ".join(f"{x:.1f}" for x in ms), flush=True)
