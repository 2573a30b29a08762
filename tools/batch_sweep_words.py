"""host-buffer batches of several sizes on configs[1]'s haystack (235 886 words): p50 of blurrily_storage_find_batch.   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap
hay, off = W.bench_haystack("words", 1.0)
n = len(off) - 1
m = RawMap()
if os.environ.get('SMALL_MIN'): m.set_option('small_min_needles', int(os.environ['SMALL_MIN']))
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
for batch in (int(x) for x in os.environ.get("BATCHES", "64,128,192,256,384,512,1024,2048,4096").split(",")):
    qq, qqo = W.queries(hay, off, batch, 9)
    t = []
    for _ in range(14):
        t0 = time.perf_counter(); m.find_batch_packed(qq, qqo, 10); t.append(time.perf_counter() - t0)
    print(f"batch {batch}: p50 {np.median(t)*1e6:.0f} us  -> {batch/np.median(t):.0f} needles/s  sweep {m.get_option('last_sweep')}", flush=True)
