#!/usr/bin/env python3
"""Batches made of nothing but needles a launch passes over -- nothing to find, too long for the launch -- through every
sweep: whole needles, the window-major sweep forced, the small-haystack sweep.  Each case prints when it is done (run
under `timeout`: the point is that it comes back) and compares the rows with the oracle's.   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle
which = sys.argv[1]
n = {"big": 700_000, "small": 200_000}[which]
hay, off = (W.geonames(n, 90000, 51) if which == "big" else W.words(n, 55))
strings = W.unpack(hay, off)
m, o = RawMap(), Oracle()
m.put_many_packed(hay, off, np.arange(1, len(strings) + 1, dtype=np.uint32)); o.put_many(hay, off)
m.sync_device()
print(which, m.device_info()["n_windows"], "windows", flush=True)
def pack(needles):
    offs = np.zeros(len(needles) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in needles])
    return np.frombuffer(b"".join(needles) or b"\0", dtype=np.uint8), offs
def run(tag, needles, limit=10):
    print("->", tag, len(needles), flush=True)
    p, q = pack(needles); rows, counts = m.find_batch_packed(p, q, limit)
    for i in (0, len(needles) // 2, len(needles) - 1):
        assert rows[i, :counts[i]].tolist() == o.find(needles[i], limit), (tag, i)
    print("   done", m.last_kernels(), "last_sweep", m.get_option("last_sweep"), int(counts.sum()), flush=True)
nothing = [b"", b"\x01\x02\x03", b"~~~~", b"{|}{|}"]
long70 = b" ".join(strings[7 * k] for k in range(14))[:110]
long200 = b" ".join(strings[11 * k] for k in range(40))[:250]
mid20 = b" ".join(strings[5 * k] for k in range(4))[:40]
print(len(set(Oracle.tokenise(long70))), len(set(Oracle.tokenise(long200))), len(set(Oracle.tokenise(mid20))), flush=True)
for forced in ((), (("wsweep", 1), ("ws_min_needles", 0), ("ws_min_windows", 0), ("ws_min_slice", 0), ("ws_static_slice", 0), ("ws_autotune", 0), ("small_sweep", 0))):
    for k, v in forced: m.set_option(k, v)
    for size in (20000, 5000):
        run(f"{forced and 'ws forced, ' or ''}nothing to find", (nothing * size)[:size])
        run(f"{forced and 'ws forced, ' or ''}65..127 trigrams only", [long70] * size)
        run(f"{forced and 'ws forced, ' or ''}more than 127 only", [long200] * (size // 10))
        run(f"{forced and 'ws forced, ' or ''}16..64 trigrams only", [mid20] * size)
        if forced: run("ws forced, a mix that is mostly nothing", ((nothing * 5 + [mid20, strings[3], long70]) * size)[:size])
print("all done")
