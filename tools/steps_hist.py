#!/usr/bin/env python3
"""Steps per needle of the needle-major sweep that leaves slices out (the counted build keeps a needle's step count in bits
31:23 of its path word), by the needle's trigram count and by the match count its answer ends at -- where the steps of a
large haystack go and which of them a better visiting order could spare.   python tools/steps_hist.py [workload] [needles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import workloads as W
from blurrily_amd import RawMap

name = sys.argv[1] if len(sys.argv) > 1 else "geonames_x4"
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
limit = W.BENCH_WORKLOADS[name]["limit"]
hay, off = W.bench_haystack(name, 1.0)
n = len(off) - 1
m = RawMap()
m.set_option("ws_autotune", 0); m.set_option("wsweep", 0); m.set_option("small_sweep", 0); m.set_option("nm_min_windows", 0)
m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
m.sync_device()
q, qo = W.bench_needles(hay, off, name, 1.0, 0, 1)
q, qo = q[:int(qo[n_q])], qo[:n_q + 1]
m.set_stats(True)
rows, counts = m.find_batch_packed(q, qo, limit)
st = m.find_stats()
flags = m.find_path_flags(n_q)
m.set_stats(False)
steps = (flags >> 23).astype(np.int64)
info = m.device_info()
W_ = int(info["n_windows"])
T = np.array([W.count_trigrams(q[int(qo[i]):int(qo[i + 1])], np.array([0, int(qo[i + 1] - qo[i])], dtype=np.uint64)) for i in range(min(n_q, 20000))])
sub = slice(0, len(T))
last = np.where(counts > 0, rows[np.arange(n_q), np.maximum(counts.astype(np.int64) - 1, 0), 1], 0)[sub]   # matches of the last row kept
best = np.where(counts > 0, rows[:, 0, 1], 0)[sub]
print(f"## {name}: {n} strings, {W_} windows, {n_q} needles, limit {limit}; sweep {m.get_option('last_sweep')}\n")
print(f"steps per needle: mean {steps.mean():.1f} (counter: {st['steps'] / n_q:.1f}), median {np.median(steps):.0f}, p10 {np.percentile(steps, 10):.0f}, p90 {np.percentile(steps, 90):.0f}, max {steps.max()}; "
      f"all windows would be {(W_ + 1) // 2} steps for a needle of up to 15 trigrams\n")
print("| needle's trigrams | needles | steps mean | p10 | p90 | full sweep would be | share stepped over |")
print("|---|---|---|---|---|---|---|")
nib = None
for lo, hi in ((1, 8), (9, 12), (13, 15), (16, 20), (21, 30), (31, 64)):
    sel = (T >= lo) & (T <= hi)
    if sel.sum() < 20:
        continue
    s_ = steps[sub][sel]
    full = (W_ + 1) // 2 if hi <= 15 else None
    print(f"| {lo}..{hi} | {int(sel.sum())} | {s_.mean():.1f} | {np.percentile(s_, 10):.0f} | {np.percentile(s_, 90):.0f} | {full if full else 'pairs up to nib_windows, single windows beyond'} | "
          f"{(1 - s_.mean() / full):.2f} |" if full else f"| {lo}..{hi} | {int(sel.sum())} | {s_.mean():.1f} | {np.percentile(s_, 10):.0f} | {np.percentile(s_, 90):.0f} | pairs up to nib_windows, single windows beyond | -- |")
print("\n| matches of the answer's last row (the final threshold) as a share of the needle's trigrams | needles | steps mean |")
print("|---|---|---|")
frac = np.where(T > 0, last / np.maximum(T, 1), 0)
for lo, hi in ((0, 0.25), (0.25, 0.4), (0.4, 0.55), (0.55, 0.7), (0.7, 1.01)):
    sel = (frac >= lo) & (frac < hi) & (T <= 15)
    if sel.sum() >= 20:
        print(f"| {lo:.2f}..{hi:.2f} (needles of up to 15 trigrams) | {int(sel.sum())} | {steps[sub][sel].mean():.1f} |")
