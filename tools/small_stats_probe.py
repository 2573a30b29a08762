import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tools", "/root/repo/tests"]
import numpy as np, workloads as W
from blurrily_amd import RawMap
hay, off = W.bench_haystack("words", 1.0)
n = len(off) - 1
m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
q, qo = W.queries(hay, off, 100000, 2)
m.find_batch_packed(q, qo, 10)
m.set_stats(True); m.find_batch_packed(q, qo, 10); st = m.find_stats(); m.set_stats(False)
print({k: v for k, v in st.items() if not isinstance(v, (list, tuple))})
print("last_sweep", m.get_option("last_sweep"))
