#!/usr/bin/env python3
"""Regenerates tests/golden/digest_*.json: SHA-256 digests of ORACLE-produced result blocks of the
benched batches ("SHA-256 of large result arrays").

    python tools/make_digests.py [words] [geonames] [skewed]      (CPU only; hours on a few cores)

For each workload of tools/workloads.py::BENCH_WORKLOADS the haystack and the step batch bench.py
times are rebuilt from their seeds, oracle/blurrily_oracle.c (the CPU restatement, pinned against the
reference's own C by tests/test_oracle_pinning.py) answers the first `needles` needles of the batch,
and every block of `block` consecutive needles is hashed (helpers.block_digests: per needle its
count, then its rows, little-endian u32).  tests/test_gpu_digests.py hashes what the GPU wrote for
the same needles the same way.  A sample of every workload is also answered by the compiled
reference (oracle/_ref) when it is present, and must agree with the oracle before a file is written.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import workloads as W  # noqa: E402
from helpers import GOLDEN, Oracle, Reference, block_digests  # noqa: E402

# workload -> needles covered (a prefix of the step batch), block length
PLAN = {"words": (100_000, 1000), "geonames": (100_000, 1000), "skewed": (100_000, 1000)}


def main():
    names = sys.argv[1:] or list(PLAN)
    threads = int(os.environ.get("DIGEST_THREADS", os.cpu_count() or 1))
    for name in names:
        n_cover, block = PLAN[name]
        spec = W.BENCH_WORKLOADS[name]
        limit = spec["limit"]
        t0 = time.time()
        hay, off = W.bench_haystack(name)
        qp, qo = W.bench_needles(hay, off, name)
        n_cover = min(n_cover, len(qo) - 1)
        o = Oracle()
        o.put_many(hay, off)
        print(f"{name}: oracle built in {time.time() - t0:.0f}s; {n_cover} needles, limit {limit}, {threads} threads",
              flush=True)
        rows = np.zeros((n_cover, limit, 3), dtype=np.uint32)
        counts = np.zeros(n_cover, dtype=np.uint32)
        step = 2000
        t1 = time.time()
        for lo in range(0, n_cover, step):
            hi = min(n_cover, lo + step)
            got = o.batch(qp, qo, idx=np.arange(lo, hi, dtype=np.uint32), limit=limit, threads=threads)
            rows[lo:hi], counts[lo:hi] = got["rows"], got["counts"]
            el = time.time() - t1
            print(f"  {hi}/{n_cover}  {el:.0f}s  eta {el / hi * (n_cover - hi):.0f}s", flush=True)
        # a sample through the reference's own C
        ref_checked = 0
        if Reference.available():
            from blurrily_amd import RawMap
            m = RawMap()
            m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
            path = f"/tmp/digest_{os.getpid()}.trigrams"
            m.save(path)
            m.close()
            ref = Reference(path)
            rng = np.random.default_rng(7)
            budget = time.time() + 120
            for q in rng.choice(n_cover, size=min(n_cover, 400), replace=False):
                nd = qp[int(qo[q]):int(qo[q + 1])].tobytes()
                want = ref.find(nd, limit)
                assert rows[q, :counts[q]].tolist() == want, (name, int(q), nd)
                ref_checked += 1
                if time.time() > budget:
                    break
            ref.close()
            os.unlink(path)
        out = {
            "workload": name, "label": spec["label"], "limit": limit, "needles": int(n_cover), "block": block,
            "haystack": {k: spec[k] for k in ("kind", "n", "hay_seed") if k in spec},
            "needle_seed": 3000, "hash": "sha256 per block: for each needle u32 count, then count x (u32 reference, "
                                         "u32 matches, u32 weight), little endian",
            "produced_by": "tools/make_digests.py: the ORACLE (oracle/blurrily_oracle.c through oracle_batch), not the compiled reference -- the "
                           "reference takes 0.2-0.8 s per needle at these sizes; the oracle is pinned to the reference (tests/test_oracle_pinning.py) "
                           "and `reference_sample_checked` needles of this very file were answered by oracle/_ref as well before it was written",
            "reference_sample_checked": ref_checked,
            "sum_counts": int(counts.astype(np.int64).sum()),
            "digests": block_digests(rows, counts, block),
        }
        path = os.path.join(GOLDEN, f"digest_{name}.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=0)
        print(f"{name}: wrote {path} ({len(out['digests'])} blocks, {time.time() - t0:.0f}s, "
              f"{ref_checked} needles also through oracle/_ref)", flush=True)


if __name__ == "__main__":
    main()
