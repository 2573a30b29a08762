"""(temporary) Step timeline of the needle-major sweep that leaves slices out, from the trace build of the library
(make EXTRA=-DBLURRILY_TRACE): shader-clock stamps of wave 0 and the manager wave over the steps of 64 needles in
mid-batch.   BLURRILY_LIB=.../libblurrily_hip_tr.so python tools/trace_steps.py   (GPU box)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native

sweep = int(os.environ.get("AB_SWEEP", "3"))
hay, off = W.bench_haystack("geonames", 1.0)
n = len(off) - 1
m = RawMap()
m.set_option("ws_autotune", 0); m.set_option("wsweep", 0)
m.set_option("nm_min_windows", 0 if sweep == 3 else 1 << 20)
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, 300000, 3000)
m.set_timing(True)
for _ in range(2):
    m.find_batch_packed(q, qo, 10)
print("kernel ms", m.device_info()["last_find_kernel_ms"])
lib = _native.lib()
lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.blurrily_debug_phase_clocks(m.handle, buf.ctypes.data, 8192) == 0
t = buf[:64 * 64 * 2 * 8].reshape(64, 64, 2, 8).astype(np.int64)
wk, mg = t[:, :, 0, :], t[:, :, 1, :]
ok = (wk[:, :, :7] != 0).all(axis=2) & (mg[:, :, [0, 1, 2, 4, 5, 6, 7]] != 0).all(axis=2)
ok[:, 1:] &= ok[:, :-1]                       # (the step before was a plain hot-loop step too)
ok[:, 0] = False
print("steps traced", int(ok.sum()))
def mean(x): return float(x[ok].mean())
print("worker (wave 0):  count %.0f  wait(count) %.0f  scan %.0f  preload %.0f  wait(scan) %.0f  glance %.0f" % (
    mean(wk[:, :, 1] - wk[:, :, 0]), mean(wk[:, :, 2] - wk[:, :, 1]), mean(wk[:, :, 3] - wk[:, :, 2]),
    mean(wk[:, :, 4] - wk[:, :, 3]), mean(wk[:, :, 5] - wk[:, :, 4]), mean(wk[:, :, 6] - wk[:, :, 5])))
print("manager:          publish %.0f  choose+fetch %.0f  wait(count) %.0f  settle %.0f  wait(scan) %.0f  glance %.0f" % (
    mean(mg[:, :, 7] - mg[:, :, 0]), mean(mg[:, :, 1] - mg[:, :, 7]), mean(mg[:, :, 2] - mg[:, :, 1]),
    mean(mg[:, :, 4] - mg[:, :, 2]), mean(mg[:, :, 5] - mg[:, :, 4]), mean(mg[:, :, 6] - mg[:, :, 5])))
step = wk[:, 1:, 0] - wk[:, :-1, 0]
print("step (wave 0, top to top): mean %.0f  median %.0f" % (float(step[ok[:, 1:]].mean()), float(np.median(step[ok[:, 1:]]))))
top = wk[:, 1:, 0] - wk[:, :-1, 6]
print("glance -> next top: %.0f" % float(top[ok[:, 1:]].mean()))
# who the barriers wait for: arrival of the manager minus arrival of wave 0
print("count barrier: manager arrives %.0f clocks after wave 0 (mean; >0: the manager is later);  late share %.2f" % (
    mean(mg[:, :, 1] - wk[:, :, 1]), float(((mg[:, :, 1] - wk[:, :, 1])[ok] > 0).mean())))
print("scan barrier:  manager arrives %.0f clocks after wave 0;  late share %.2f" % (
    mean(mg[:, :, 4] - wk[:, :, 4]), float(((mg[:, :, 4] - wk[:, :, 4])[ok] > 0).mean())))
