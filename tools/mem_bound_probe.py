"""Is the needle-major sweep waiting for memory?  The same multiset of needles in two orders: every
distinct needle 512 times in a row (the 512 resident workgroups sweep the same needle at the same time:
its postings come from L2) against the same needles shuffled (postings from HBM / Infinity Cache).
python tools/mem_bound_probe.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
from blurrily_amd.map import _pack

hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap()
m.set_option("wsweep", 0)
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, 1024, 3000)
distinct = W.unpack(q, qo)
reps = 512
blocked = [nd for nd in distinct for _ in range(reps)]
rng = np.random.default_rng(1)
shuffled = [blocked[i] for i in rng.permutation(len(blocked))]
m.set_timing(True)
for label, batch in (("blocked (L2)", blocked), ("shuffled (HBM)", shuffled), ("blocked (L2)", blocked)):
    p, o = _pack(batch)
    for _ in range(2):
        m.find_batch_packed(p, o, 10)
    print(label, len(batch), "needles: kernel ms", round(m.device_info()["last_find_kernel_ms"], 2), flush=True)
