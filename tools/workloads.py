"""ctypes front-end of tools/synth.cpp: the seeded workloads of BASELINE.json's configs.

Bench/test infrastructure (not part of the product).  Every generator returns
``(packed: np.ndarray[uint8], offsets: np.ndarray[uint64, n+1])``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _synth():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libblurrily_synth.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.synth_max_bytes.restype = C.c_uint64
        L.synth_max_bytes.argtypes = [C.c_uint32]
        for name, args in {
            "synth_words": [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p],
            "synth_geonames": [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p],
            "synth_skewed": [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p],
            "synth_skewed_mix": [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p],
            "synth_queries": [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p],
            "synth_count_trigrams": [C.c_void_p, C.c_void_p, C.c_uint32],
        }.items():
            fn = getattr(L, name)
            fn.restype = C.c_uint64
            fn.argtypes = args
        _lib = L
    return _lib


def _run(n, call):
    L = _synth()
    buf = np.empty(int(L.synth_max_bytes(n)), dtype=np.uint8)
    off = np.empty(n + 1, dtype=np.uint64)
    used = call(L, buf.ctypes.data, off.ctypes.data)
    return buf[:used].copy(), off


def words(n=235886, seed=1):
    """configs[0..1]: distinct pseudo-words (stand-in for /usr/share/dict/words)."""
    return _run(n, lambda L, b, o: L.synth_words(seed, n, b, o))


def geonames(n=8423769, vocab=500000, seed=3):
    """configs[2..3]: Geonames-scale multi-word haystack."""
    return _run(n, lambda L, b, o: L.synth_geonames(seed, n, vocab, b, o))


def skewed(n=4000000, seed=5):
    """configs[4]: hot-trigram haystack with massive (matches, weight) ties."""
    return _run(n, lambda L, b, o: L.synth_skewed(seed, n, b, o))


def skewed_mix(n=4000000, hot_pct=100, seed=5):
    """The skewed haystack with its hot prefixes / suffixes on `hot_pct` per cent of the strings that would take
    them (100: `skewed`; 0: bare stems): haystacks between plain and hot-trigram (tools/gate_probe.py)."""
    return _run(n, lambda L, b, o: L.synth_skewed_mix(seed, n, hot_pct, b, o))


def queries(hay, hay_off, n, seed):
    """Haystack samples with 0..2 random edits."""
    n_hay = len(hay_off) - 1
    return _run(n, lambda L, b, o: L.synth_queries(seed, hay.ctypes.data, hay_off.ctypes.data, n_hay, n, b, o))


def unpack(packed, offsets):
    raw = packed.tobytes()
    return [raw[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]


def count_trigrams(packed, offsets):
    """Sum of distinct-trigram counts over the needles."""
    return int(_synth().synth_count_trigrams(packed.ctypes.data, offsets.ctypes.data, len(offsets) - 1))


# ---- the inputs bench.py times (shared with tests/test_gpu_bench_batch.py, which checks that very
# ---- call on those very inputs) ------------------------------------------------------------------
BENCH_WORKLOADS = {
    # name: haystack generator arguments, needles per rank, limit, the BASELINE.json config it is
    "geonames": dict(kind="geonames", n=8423769, vocab=500000, hay_seed=3, queries=1_000_000, limit=10,
                     label="configs[2]: synthetic Geonames-scale haystack, 1M batched needles"),
    "words":    dict(kind="words", n=235886, hay_seed=1, queries=100_000, limit=10,
                     label="configs[1]: 235k-word haystack, 100k batched needles"),
    "skewed":   dict(kind="skewed", n=4_000_000, hay_seed=5, queries=100_000, limit=100,
                     label="configs[4]: adversarial hot-trigram haystack, limit=100"),
    # round 4: past the Infinity Cache (four times configs[2]'s haystack: the image is >= 7 x the 256 MiB cache) ...
    "geonames_x4":   dict(kind="geonames", n=4 * 8423769, vocab=500000, hay_seed=3, queries=100_000, limit=10,
                          label="four times configs[2]'s haystack (33.7M strings), 100k batched needles"),
    # ... and needles WITHOUT a close match: configs[2]'s haystack, needles cut from a haystack of another seed's
    # vocabulary -- what the reference's own bench does with its fixed city names on any dataset (bin/bench:24-25,160-162)
    # configs[0-1] name /usr/share/dict/words: benched where the box has the file (bench.py records its SHA-256, or
    # that it looked and found none); `words` above is its seeded stand-in, which the digests of tests/golden hold
    "dict_words": dict(kind="dict", n=0, hay_seed=0, queries=100_000, limit=10,
                       label="configs[1] on the box's own /usr/share/dict/words, 100k batched needles"),
    "geonames_miss": dict(kind="geonames", n=8423769, vocab=500000, hay_seed=3, queries=100_000, limit=10,
                          needles_from=dict(n=200_000, vocab=500000, seed=1003),
                          label="configs[2]'s haystack, 100k needles from a foreign vocabulary (no close match)"),
}


DICT_WORDS_PATH = "/usr/share/dict/words"       # BASELINE.json configs[0-1] name it; SURVEY.md 8(d): "if the box has it"


def dict_words(path=DICT_WORDS_PATH):
    """The box's own word list as a haystack, or None where there is none: one string per line, as Blurrily::Map#put
    would index it (lib/blurrily/map.rb:40-47 for ASCII: downcase, everything but a-z a space, squeezed, stripped;
    lines with a byte >= 0x80 and lines that come out empty are left out).  Returns (packed, offsets, sha256 of the file)."""
    import hashlib
    import re
    if not os.path.isfile(path):
        return None
    raw = open(path, "rb").read()
    out = []
    for line in raw.split(b"\n"):
        if not line or max(line) >= 0x80:
            continue
        s = b" ".join(re.sub(rb"[^a-z]", b" ", line.lower()).split())
        if s:
            out.append(s)
    packed = np.frombuffer(b"".join(out), dtype=np.uint8).copy()
    off = np.zeros(len(out) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in out])
    return packed, off, hashlib.sha256(raw).hexdigest()


# the eight needles of the reference's own benchmark (bin/bench:24-25), as Map#find hands them to the C call
PUBLISHED_NEEDLES = [b"london", b"paris", b"rome", b"luxembourg", b"lonndon", b"pari", b"roma", b"luxenbour"]
# record counts of its six datasets (doc/bench.numbers): cities .. world
PUBLISHED_RECORDS = [131_002, 347_014, 474_695, 828_647, 2_158_158, 8_423_769]
# what the synthetic haystacks of that curve end with, so that the needles have something to find (the real datasets
# hold these places; a haystack of pseudo-words does not)
PUBLISHED_PLACES = [b"london", b"paris", b"rome", b"luxembourg", b"london city airport", b"paris texas", b"roma termini",
                    b"luxembourg ville"]


def published_haystack(records):
    """A Geonames-kind haystack of `records` strings (same generator and vocabulary rule as configs[2]) whose last eight
    strings are PUBLISHED_PLACES."""
    hay, off = geonames(records, max(1000, min(500000, records // 16)), 3)
    strings_end = int(off[records - len(PUBLISHED_PLACES)])
    tail = b"".join(PUBLISHED_PLACES)
    packed = np.concatenate([hay[:strings_end], np.frombuffer(tail, dtype=np.uint8)])
    off = off.copy()
    off[records - len(PUBLISHED_PLACES) + 1:] = strings_end + np.cumsum([len(s) for s in PUBLISHED_PLACES]).astype(np.uint64)
    return packed, off


def bench_haystack(name, scale=1.0):
    """(packed, offsets) of the haystack bench.py indexes for `name` (refs are 1..n, weight 0)."""
    spec = BENCH_WORKLOADS[name]
    if spec["kind"] == "dict":
        got = dict_words()
        if got is None:
            raise FileNotFoundError(DICT_WORDS_PATH)
        return got[0], got[1]
    n = max(1000, int(spec["n"] * scale))
    if spec["kind"] == "geonames":
        return geonames(n, max(1000, int(spec["vocab"] * min(1.0, scale * 4))), spec["hay_seed"])
    if spec["kind"] == "words":
        return words(n, spec["hay_seed"])
    return skewed(n, spec["hay_seed"])


def bench_needles(hay, hay_off, name, scale=1.0, rank=0, world=1):
    """This rank's shard of the step batch: world x n_q needles in contiguous shards, seeded per rank."""
    spec = BENCH_WORKLOADS[name]
    n_q = max(100, int(spec["queries"] * scale))
    seed = (3 if world == 1 else 4) * 1000 + rank
    if "needles_from" in spec:                      # needles of another vocabulary: no close match in `hay`
        src = spec["needles_from"]
        f_hay, f_off = geonames(max(1000, int(src["n"] * min(1.0, scale * 4))), src["vocab"], src["seed"])
        return queries(f_hay, f_off, n_q, seed)
    return queries(hay, hay_off, n_q, seed)
