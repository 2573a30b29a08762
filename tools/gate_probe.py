#!/usr/bin/env python3
"""Where does the window-major sweep overtake the needle-major one?  (GPU box)

    python tools/gate_probe.py [n_strings] [batch sizes ...]          default 4000000  16384 100000 1000000

Haystacks between plain and hot-trigram -- the skewed generator with its hot prefixes / suffixes on 0 .. 100 per
cent of the strings (tools/synth.cpp: synth_skewed_mix), plus the Geonames-scale one -- each with the figure the
choice is gated on (DeviceIndex::mean_hit_slice: postings a needle trigram finds per window, on average); both
sweeps timed on the same box for every batch size (kernel ms of the second of two calls), rows compared.
Prints a markdown table and one JSON line per point."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from blurrily_amd import RawMap  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    batches = [int(a) for a in sys.argv[2:]] or [16384, 100_000, 1_000_000]
    limit = int(os.environ.get("GATE_LIMIT", "10"))
    pcts = [int(x) for x in os.environ.get("GATE_PCTS", "0,25,50,75,100").split(",")]
    points = [("skew %d%%" % p, lambda p=p: W.skewed_mix(n, p, 5)) for p in pcts]
    if os.environ.get("GATE_GEONAMES", "1") != "0":
        points.append(("geonames", lambda: W.bench_haystack("geonames")))
        points.append(("geonames 1/2", lambda: W.bench_haystack("geonames", 0.5)))
        points.append(("words 4M", lambda: W.words(4_000_000, 9)))
    print(f"| haystack | strings | windows | mean_hit_slice | dense_share | ws_gain | " +
          " | ".join(f"{b} needles: needle-major / window-major ms (ratio)" for b in batches) + " |")
    print("|---|---|---|---|---|---|" + "---|" * len(batches))
    for label, gen in points:
        hay, off = gen()
        m = RawMap()
        m.set_option("ws_min_slice", 0)                 # bitmaps built whatever the slice sizes: both sweeps can run
        m.set_option("ws_static_slice", 0)
        m.set_option("ws_autotune", 0)                  # the sweep asked for ("wsweep" 0 / 1), not the measured choice
        m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
        m.sync_device()
        info = m.device_info()
        cells = []
        for nq in batches:
            q, qo = W.queries(hay, off, nq, 3000)
            res = {}
            for name, on in (("needle", 0), ("window", 1)):
                m.set_option("wsweep", on)
                m.set_option("ws_min_needles", 1)
                m.find_batch_packed(q, qo, limit)
                m.set_timing(True)
                rows, counts = m.find_batch_packed(q, qo, limit)
                res[name] = (m.device_info()["last_find_kernel_ms"], rows, counts)
                m.set_timing(False)
            a, b = res["needle"], res["window"]
            live = np.arange(limit)[None, :] < a[2][:, None].astype(np.int64)
            same = bool(np.array_equal(a[2], b[2]) and
                        np.array_equal(np.where(live[:, :, None], a[1], 0), np.where(live[:, :, None], b[1], 0)))
            cells.append(f"{a[0]:.1f} / {b[0]:.1f} ({a[0] / b[0]:.2f}x){'' if same else ' ROWS DIFFER'}")
            print(json.dumps({"haystack": label, "strings": len(off) - 1, "windows": info["n_windows"],
                              "mean_hit_slice": info["mean_hit_slice"], "dense_share": info["dense_share"],
                              "ws_gain": info["ws_gain"], "needles": nq, "limit": limit,
                              "needle_major_ms": a[0], "window_major_ms": b[0], "rows_equal": same}), file=sys.stderr)
        print(f"| {label} | {len(off) - 1} | {info['n_windows']} | {info['mean_hit_slice']:.0f} | {info['dense_share']:.2f} | "
              f"{info['ws_gain']:.0f} | " + " | ".join(cells) + " |",
              flush=True)
        m.close()


if __name__ == "__main__":
    main()
