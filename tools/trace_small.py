"""(temporary) timeline of find_small_kernel from the trace build: wave 0's stamps over needles 50000..50063 of the words batch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
hay, off = W.bench_haystack("words", 1.0)
n = len(off) - 1
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.bench_needles(hay, off, "words", 1.0, 0, 1)
m.set_timing(True)
for _ in range(2): m.find_batch_packed(q, qo, 10)
print("kernel ms", m.device_info()["last_find_kernel_ms"], "sweep", m.get_option("last_sweep"))
lib = _native.lib(); lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
t = buf[:64 * 64 * 2 * 8].reshape(64, 64, 2, 8)[:, :, 0, :].astype(np.int64)
ok = (t[:, 7, 0] != 0) & (t[:, 7, 2] != 0)
print("needles traced", int(ok.sum()))
nd = t[ok]
print("needle: setup->sweep end %.0f  emit %.0f  total %.0f" % ((nd[:, 7, 1] - nd[:, 7, 0]).mean(), (nd[:, 7, 2] - nd[:, 7, 1]).mean(), (nd[:, 7, 2] - nd[:, 7, 0]).mean()))
gap = [nd[k + 1, 7, 0] - nd[k, 7, 2] for k in range(len(nd) - 1)]
for i in range(4):
    w = nd[:, i, :]
    full = (w != 0).all(axis=1)
    if full.sum() == 0: print("window", i, "never full"); continue
    w = w[full]
    names = ["advance", "count issue", "head of next", "barrier", "read+clear", "bisect", "harvest+barrier"]
    print("window %d (%d): " % (i, full.sum()) + "  ".join("%s %.0f" % (nm, (w[:, k + 1] - w[:, k]).mean()) for k, nm in enumerate(names)) + "  total %.0f" % (w[:, 7] - w[:, 0]).mean())
    if i < 3:
        nxt = nd[full][:, i + 1, 0]
        print("   to next window top: %.0f" % (nxt[nxt != 0] - w[nxt != 0][:, 7]).mean())
