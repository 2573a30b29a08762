#!/usr/bin/env python3
"""A reader of the compiled reference (oracle/_ref) in a process of its own: bench.py's cpu_baseline leg hands the rows
beyond its timed sample -- and the all-cores figure -- to such readers instead of forking itself (a process that has HIP
and torch initialised is not one to fork).  Test infrastructure, like everything that touches oracle/.

    python tools/ref_reader.py <file.trigrams> <needles.npz> <lo> <hi> <limit> <out.npz>

needles.npz: packed (uint8, NUL-terminated C strings), starts (uint32).  Writes rows [hi-lo, limit, 3], counts [hi-lo] and
the wall-clock span of the finds (t0, t1: time.time())."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from helpers import Reference

path, needles, lo, hi, limit, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
z = np.load(needles)
packed, starts = z["packed"], np.ascontiguousarray(z["starts"][lo:hi])
ref = Reference(path)
S = Reference.shim()
rows = np.zeros((hi - lo, max(limit, 1), 3), dtype=np.uint32)
counts = np.zeros(hi - lo, dtype=np.uint32)
t0 = time.time()
S.ref_find_many(ref.h, packed.ctypes.data, starts.ctypes.data, hi - lo, limit, rows.ctypes.data, counts.ctypes.data)
t1 = time.time()
np.savez(out, rows=rows, counts=counts, span=np.array([t0, t1]))
ref.close()
