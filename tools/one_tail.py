#!/usr/bin/env python3
"""Where the single find's tail goes: N finds one after the other, host clock around blurrily_storage_find, per find the
needle's trigram count; run under `rocprofv3 --kernel-trace` the kernels' own durations (in launch order) are joined with
the host latencies afterwards (ONE_TAIL_DB=<results.db>).

    python tools/one_tail.py <workload> <limit> <out.json>            # measure (GPU box; under rocprofv3 for the join)
    python tools/one_tail.py --join <out.json> <results.db>           # join + summary (markdown on stdout)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np


def measure(workload, limit, out):
    import workloads as W
    from blurrily_amd import RawMap, _native
    hay, off = W.bench_haystack(workload, 1.0)
    n = len(off) - 1
    m = RawMap(); m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32)); m.sync_device()
    q, qo = W.bench_needles(hay, off, workload, 1.0, 0, 1)
    needles = W.unpack(q, qo[:int(os.environ.get("ONE_TAIL_N", "1000")) + 4])
    lib = _native.lib()
    rows = (_native.TrigramMatch * limit)()
    codes = (np.zeros(512, dtype=np.uint16))
    rec = []
    for i, nd in enumerate(needles):
        t = time.perf_counter()
        c = lib.blurrily_storage_find(m.handle, nd, limit, rows)
        dt = time.perf_counter() - t
        T = lib.blurrily_tokeniser_parse_string(nd, codes.ctypes.data)
        if i >= 4:                                    # (the first finds set the stream and the pinned page up)
            rec.append({"us": dt * 1e6, "T": int(T), "rows": int(c), "len": len(nd), "matches0": int(rows[0].matches) if c else 0})
    info = m.device_info()
    json.dump({"workload": workload, "limit": limit, "windows": int(info["n_windows"]), "one_taken": int(m.get_option("one_taken")), "finds": rec},
              open(out, "w"))
    lat = np.array([r["us"] for r in rec])
    print(f"{workload} limit {limit}: p50 {np.median(lat):.1f} p90 {np.percentile(lat, 90):.1f} p99 {np.percentile(lat, 99):.1f} max {lat.max():.1f} us")


def join(path, db):
    import sqlite3
    d = json.load(open(path))
    c = sqlite3.connect(db)
    ks = c.execute("select name, start, end from kernels where name like '%find_one_kernel%' order by start").fetchall()
    finds = d["finds"]
    ks = ks[-len(finds):]                              # (the warm-up finds' launches come first)
    assert len(ks) == len(finds), (len(ks), len(finds))
    host = np.array([f["us"] for f in finds]); kern = np.array([(e - s) / 1e3 for _, s, e in ks])
    gap = host - kern
    T = np.array([f["T"] for f in finds])
    def pct(x): return f"p50 {np.median(x):.1f} / p90 {np.percentile(x, 90):.1f} / p99 {np.percentile(x, 99):.1f} / max {x.max():.1f}"
    print(f"### {d['workload']}, limit {d['limit']} ({d['windows']} windows, {len(finds)} finds)\n")
    print(f"* host clock around blurrily_storage_find (us): {pct(host)}")
    print(f"* find_one_kernel's own duration (us):          {pct(kern)}")
    print(f"* the rest -- launch, PCIe write, poll, copy (us): {pct(gap)}")
    slow = host >= np.percentile(host, 99)
    print(f"* the slowest 1 % of the finds ({int(slow.sum())}): kernel {np.median(kern[slow]):.1f} us (all: {np.median(kern):.1f}), "
          f"rest {np.median(gap[slow]):.1f} us (all: {np.median(gap):.1f}); trigrams {np.median(T[slow]):.0f} (all: {np.median(T):.0f})")
    print(f"* correlation of the host latency with the kernel's duration {np.corrcoef(host, kern)[0, 1]:.2f}, with the rest {np.corrcoef(host, gap)[0, 1]:.2f}, "
          f"of the kernel's duration with the needle's trigrams {np.corrcoef(kern, T)[0, 1]:.2f}")
    for lo, hi in ((0, 15), (16, 31), (32, 64)):
        sel = (T >= lo) & (T <= hi)
        if sel.sum() >= 5:
            print(f"* needles of {lo}..{hi} trigrams ({int(sel.sum())}): host p50 {np.median(host[sel]):.1f} / p99 {np.percentile(host[sel], 99):.1f}, kernel p50 {np.median(kern[sel]):.1f} / p99 {np.percentile(kern[sel], 99):.1f}")
    print()


if __name__ == "__main__":
    if sys.argv[1] == "--join":
        join(sys.argv[2], sys.argv[3])
    else:
        measure(sys.argv[1], int(sys.argv[2]), sys.argv[3])
