#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc sqlite output (gpurun_out/pmc_<tag>/pmc_results.db): per kernel,
per counter: sum over dispatches and per-dispatch mean for the batched find launch."""
import sqlite3
import sys


def summarise(path, only=None):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), max(duration) "
                     "from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for k, n, v, disp, dur in rows:
        short = k.replace("void ", "").replace("blurrily::(anonymous namespace)::", "").split("(")[0]
        if only and only not in short:
            continue
        out.setdefault(short, {})[n] = (v, disp, dur)
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("==", p)
        for k, ctr in summarise(p).items():
            for n, (v, disp, dur) in sorted(ctr.items()):
                print(f"  {k:40s} {n:24s} sum={v:.6g} dispatches={disp} max_dispatch_ns={dur}")
