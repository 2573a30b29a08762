#!/usr/bin/env python3
"""Generate tests/golden/ref_*.json from the reference's own C compiled in place.

Runs only in the container that has /root/reference (oracle/_ref must be built:
`make -C oracle`).  The fixtures are data -- seeded inputs and the outputs the reference
produced for them -- so the GPU box can pin oracle and HIP path without the reference tree.

    python tools/make_golden.py

How a haystack reaches the reference: this repo's blurrily_storage_put/save write a
`.trigrams` file; the reference's blurrily_storage_load maps it (oracle/ref_shim.c).  The
reference's find and tokeniser then produce the expected values stored here.
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from blurrily_amd import RawMap  # noqa: E402
from helpers import GOLDEN, Reference, golden_haystack  # noqa: E402


def case(name, kind, n, hay_seed, q_seed, n_q, limit, refs="seq", weights="zero"):
    meta = {"kind": kind, "n": n, "seed": hay_seed, "refs": refs, "weights": weights, "ref_seed": hay_seed}
    hay, off, ref_arr, w_arr = golden_haystack(meta)
    m = RawMap()
    m.put_many_packed(hay, off, ref_arr, w_arr)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "h.trigrams")
        m.save(path)
        ref = Reference(path)
        q, qo = W.queries(hay, off, n_q, q_seed)
        needles = W.unpack(q, qo) + [b"", b"zzzz", b"a", b"the quick brown fox"]
        expected = [ref.find(nd, limit) for nd in needles]
        stats = ref.stats()
        ref.close()
    out = {
        "generator": "tools/make_golden.py (reference C: oracle/_ref/libblurrily_ref.so)",
        "haystack": meta,
        "stats": stats, "limit": limit,
        "needles_hex": [nd.hex() for nd in needles],
        "expected": expected,
    }
    with open(os.path.join(GOLDEN, f"ref_find_{name}.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(name, len(needles), "needles")


def tokeniser_vectors():
    rng = np.random.default_rng(42)
    needles = [b"", b"london", b"new york", b"a", b"aa", b"aaa", b"aaaa", b"port au prince", b"  ", b"a  b",
               b"Mixed CASE", b"caf\xc3\xa9", b"***", b"z" * 40, b"abcdefghijklmnopqrstuvwxyz" * 6]
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz  *-AZ\xc3\xa9", dtype=np.uint8)
    for _ in range(200):
        ln = int(rng.integers(0, 40))
        needles.append(bytes(rng.choice(alphabet, size=ln).tolist()).replace(b"\0", b"a"))
    out = {"generator": "tools/make_golden.py (blurrily_tokeniser_parse_string of the reference)",
           "vectors": [{"needle_hex": nd.hex(), "codes": Reference.tokenise(nd)} for nd in needles]}
    with open(os.path.join(GOLDEN, "ref_tokeniser.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("tokeniser", len(needles), "vectors")


if __name__ == "__main__":
    assert Reference.available(), "build oracle/_ref first (make -C oracle) -- needs /root/reference"
    os.makedirs(GOLDEN, exist_ok=True)
    tokeniser_vectors()
    case("words3k", "words", 3000, 21, 22, 120, 10)
    case("words3k_limit100", "words", 3000, 21, 23, 40, 100)
    case("geo20k", "geonames", 20000, 31, 32, 120, 10)
    case("skewed_ties", "skewed", 8000, 41, 42, 80, 25, refs="sparse", weights="small")
