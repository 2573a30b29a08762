import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap, _native
n = 8423769
hay, off = W.geonames(n, 500000, 3)
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
q, qo = W.queries(hay, off, 300, 7)
needles = W.unpack(q, qo)
lib = _native.lib()
rows = (_native.TrigramMatch * 10)()
lat = []
for nd in needles:
    t = time.perf_counter(); lib.blurrily_storage_find(m.handle, nd, 10, rows); lat.append(time.perf_counter() - t)
print(f"single: p50 {np.median(lat)*1e6:.0f} us  p10 {np.percentile(lat,10)*1e6:.0f}  p90 {np.percentile(lat,90)*1e6:.0f}")
for batch in (int(x) for x in os.environ.get("BATCHES", "2,8,32,64,128,256,384,512").split(",")):
    qq, qqo = W.queries(hay, off, batch, 9)
    t = []
    for _ in range(12):
        t0 = time.perf_counter(); m.find_batch_packed(qq, qqo, 10); t.append(time.perf_counter() - t0)
    print(f"batch {batch}: p50 {np.median(t)*1e6:.0f} us  -> {batch/np.median(t):.0f} needles/s", flush=True)
