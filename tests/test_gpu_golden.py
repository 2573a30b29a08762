"""GPU parity against data produced by the reference's own C: the committed fixtures
(tests/golden/ref_find_*.json) and, where oracle/_ref travelled with the snapshot, the live
reference on larger haystacks; plus size-independent properties at BASELINE.json's full
Geonames scale, with a sample checked row for row against the oracle."""
import os
import tempfile

import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle, Reference, golden_find_files, golden_haystack, load_golden

pytestmark = pytest.mark.gpu


def _batch(m, needles, limit):
    packed = np.frombuffer(b"".join(needles), dtype=np.uint8)
    off = np.zeros(len(needles) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in needles])
    rows, counts = m.find_batch_packed(packed, off, limit)
    return [rows[i, :counts[i]].tolist() for i in range(len(needles))]


@pytest.mark.parametrize("name", golden_find_files())
def test_hip_matches_reference_fixture(name):
    g = load_golden(name)
    hay, off, refs, weights = golden_haystack(g["haystack"])
    m = RawMap()
    m.put_many_packed(hay, off, refs, weights)
    assert m.stats() == g["stats"]
    needles = [bytes.fromhex(h) for h in g["needles_hex"]]
    assert _batch(m, needles, g["limit"]) == g["expected"]
    taken = m.get_option("one_taken")
    for nd, want in zip(needles, g["expected"]):                     # EVERY needle through the single-needle entry point too
        rows = (np.zeros((g["limit"], 3), dtype=np.uint32))
        n = m._lib.blurrily_storage_find(m.handle, nd, g["limit"], rows.ctypes.data)
        assert rows[:n].tolist() == want
    if g["limit"] <= 120:                                            # ... which went as ONE launch (c_abi.hip: find_one)
        assert m.get_option("one_taken") - taken >= sum(1 for w in g["expected"] if w)


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref did not travel with the snapshot")
@pytest.mark.parametrize("kind,n,limit", [("geonames", 300000, 10), ("skewed", 200000, 100), ("words", 235886, 10)])
def test_hip_vs_live_reference(kind, n, limit):
    hay, off = {"words": W.words, "skewed": W.skewed}.get(kind, lambda n, s: W.geonames(n, 30000, s))(n, 17)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    q, qo = W.queries(hay, off, 300, 18)
    needles = W.unpack(q, qo)
    got = _batch(m, needles, limit)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "h.trigrams")
        m.save(path)
        ref = Reference(path)
        for nd, rows in zip(needles, got):
            assert rows == ref.find(nd, limit), nd
        ref.close()


def test_full_geonames_scale_properties(geonames_full):
    """configs[2] haystack (8 423 769 strings): properties that need no full-size checker, and a
    sample of needles checked row for row against the oracle."""
    n = 8423769
    hay, off = geonames_full.hay, geonames_full.off
    assert len(off) - 1 == n
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    rng = np.random.default_rng(1)
    picks = rng.integers(0, n, size=2000)
    strings = [hay[int(off[i]):int(off[i + 1])].tobytes() for i in picks]
    q, qo = W.queries(hay, off, 2000, 99)
    needles = strings + W.unpack(q, qo)
    limit = 10
    got = _batch(m, needles, limit)
    again = _batch(m, needles, limit)
    assert got == again                                            # idempotent
    for nd, rows in zip(needles, got):
        T = len(Oracle.tokenise(nd))
        assert len(rows) <= limit
        keys = [(-r[1], r[2], r[0]) for r in rows]
        assert keys == sorted(keys) and len(set(r[0] for r in rows)) == len(rows)   # the total order, no dup refs
        assert all(1 <= r[1] <= T for r in rows)
    for i, (nd, rows) in enumerate(zip(strings, got)):             # an indexed string finds itself
        T = len(Oracle.tokenise(nd))
        own = (-T, len(nd), int(picks[i]) + 1)                     # all trigrams match; weight = strlen
        keys = [(-r[1], r[2], r[0]) for r in rows]
        assert rows[0][1] == T                                     # nothing can match more than T trigrams
        assert own in keys or (len(keys) == limit and max(keys) < own)
    o = geonames_full.oracle
    for nd, rows in list(zip(needles, got))[1990:2030]:
        assert rows == o.find(nd, limit), nd
    # the single find at this scale -- 129 workgroups, one list each, merged by find_one_kernel's last workgroup
    # (one_merge: up to 129 x 120 slots) -- at the limits where the lists are long
    taken = m.get_option("one_taken")
    longest = max((x for x in strings if len(Oracle.tokenise(x)) <= 64), key=lambda x: len(Oracle.tokenise(x)))   # many levels
    singles = needles[1996:2007] + [longest]
    for lim in (10, 100, 120):
        for nd in singles:
            out = np.zeros((lim, 3), dtype=np.uint32)
            k = m._lib.blurrily_storage_find(m.handle, nd, lim, out.ctypes.data)
            assert out[:k].tolist() == o.find(nd, lim), (nd, lim)
    assert m.get_option("one_taken") - taken == 3 * len(singles)


def test_full_skewed_scale_properties():
    """configs[4] haystack (4 000 000 hot-trigram strings, limit 100): massive (matches, weight)
    ties; order, uniqueness and idempotence on every row, a sample row for row against the oracle."""
    n = 4_000_000
    hay, off = W.skewed(n, 5)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    q, qo = W.queries(hay, off, 1500, 55)
    needles = W.unpack(q, qo)
    limit = 100
    got = _batch(m, needles, limit)
    assert got == _batch(m, needles, limit)
    for nd, rows in zip(needles, got):
        T = len(Oracle.tokenise(nd))
        keys = [(-r[1], r[2], r[0]) for r in rows]
        assert keys == sorted(keys) and len(set(r[0] for r in rows)) == len(rows) and len(rows) <= limit
        assert all(1 <= r[1] <= T for r in rows)
    o = Oracle()
    o.put_many(hay, off)
    for nd, rows in list(zip(needles, got))[:24]:
        assert rows == o.find(nd, limit), nd
    # ... and the single find on this haystack of massive ties (62 lists of up to 120 keys of a handful of levels)
    for lim in (100, 120, 7):
        for nd in needles[:8]:
            out = np.zeros((lim, 3), dtype=np.uint32)
            k = m._lib.blurrily_storage_find(m.handle, nd, lim, out.ctypes.data)
            assert out[:k].tolist() == o.find(nd, lim), (nd, lim)


def test_config2_words_100k_batch():
    """configs[1]: the 235 886-word haystack and ONE batch of 100 000 needles (seed 2); every
    needle's rows are checked for order/uniqueness, 4 000 of them row for row against the oracle."""
    hay, off = W.words(235886, 1)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, 235887, dtype=np.uint32))
    q, qo = W.queries(hay, off, 100_000, 2)
    rows, counts = m.find_batch_packed(q, qo, 10)
    assert counts.max() <= 10 and counts.min() >= 1          # every needle is an edited haystack word
    keys_ok = True
    r = rows.astype(np.int64)
    for k in range(9):                                       # vectorised order check: (-matches, weight, ref) ascending
        valid = counts > k + 1
        a, b = r[valid, k], r[valid, k + 1]
        lt = (a[:, 1] > b[:, 1]) | ((a[:, 1] == b[:, 1]) & ((a[:, 2] < b[:, 2]) | ((a[:, 2] == b[:, 2]) & (a[:, 0] < b[:, 0]))))
        keys_ok &= bool(lt.all())
    assert keys_ok
    o = Oracle()
    o.put_many(hay, off)
    needles = W.unpack(q, qo)
    for i in range(0, 100_000, 25):
        assert rows[i, :counts[i]].tolist() == o.find(needles[i], 10), needles[i]
