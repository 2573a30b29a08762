"""Host-buffer batch (blurrily_storage_find_batch) against the device-resident one, by chunk size (GPU box):
python tools/host_pipe_probe.py [needles]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
hay, off = W.bench_haystack("geonames")
m = RawMap()
m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
m.sync_device()
q, qo = W.queries(hay, off, nq, 3000)
lib = _native.lib()
rows = np.ones((nq, 10, 3), dtype=np.uint32)
counts = np.ones(nq, dtype=np.uint32)
m.set_timing(True)
m.find_batch_packed(q[:int(qo[1000])], qo[:1001], 10)
m.set_option("host_chunk", 0)
for _ in range(2):
    lib.blurrily_storage_find_batch(m.handle, q.ctypes.data, qo.ctypes.data, nq, 10, rows.ctypes.data, counts.ctypes.data)
k_ms = m.device_info()["last_find_kernel_ms"]
print(f"kernel alone (timing mode, one piece): {k_ms:.1f} ms -> {nq / k_ms / 1e3:.3f} M needles/s", flush=True)
m.set_timing(False)
for chunk in (0, 32768, 65536, 131072, 262144):
    m.set_option("host_chunk", chunk)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        lib.blurrily_storage_find_batch(m.handle, q.ctypes.data, qo.ctypes.data, nq, 10, rows.ctypes.data, counts.ctypes.data)
        ts.append(time.perf_counter() - t)
    print(f"host_chunk {chunk:7d}: {min(ts) * 1e3:7.1f} ms -> {nq / min(ts) / 1e6:.3f} M needles/s ({100 * (min(ts) * 1e3 / k_ms - 1):+.1f} % vs kernel)", flush=True)
