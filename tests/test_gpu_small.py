"""The small-haystack sweep (find_kernels.hip: find_small_kernel; c_abi.hip: sweep 4) against the oracle, row for row.
An image of at most eight windows serves large batches at limits up to 64 with four waves and one window's 4-bit
counters per needle; needles of 16..64 distinct trigrams are listed by it and follow through the byte-counter kernel.
What it must compute is what every sweep must (storage.c:477-580): every reference's match count, the best `limit`
by matches, weight, reference."""
import os

import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap
from blurrily_amd.map import _pack
from helpers import Oracle

pytestmark = pytest.mark.gpu
SMALL = 1 << 22               # kPathSmall
COLD = 1 << 2                 # kPathColdStart
RESWEEP = 1 << 3              # kPathResweep


def _pair(hay, off):
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    o.put_many(hay, off)
    return m, o


def _check(m, o, packed, off, limit, upto=None):
    m.set_stats(True)
    rows, counts = m.find_batch_packed(packed, off, limit)
    flags = m.find_path_flags(len(off) - 1)
    m.set_stats(False)
    assert m.get_option("last_sweep") == 4, m.get_option("last_sweep")
    # (several limits over one batch: the oracle runs once at the largest, helpers.Oracle.batch_upto)
    want = o.batch(packed, off, limit=limit) if upto is None else o.batch_upto(packed, off, limit, upto)
    assert np.array_equal(counts, want["counts"])
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    bad = np.nonzero((np.where(live[:, :, None], rows, 0) != np.where(live[:, :, None], want["rows"], 0)).any(axis=(1, 2)))[0]
    if len(bad):
        q = int(bad[0])
        nd = bytes(packed[int(off[q]):int(off[q + 1])])
        raise AssertionError((len(bad), q, nd, rows[q, :counts[q]].tolist(), want["rows"][q, :counts[q]].tolist()))
    # ... and the timed build of the same kernels writes the same rows
    rows_t, counts_t = m.find_batch_packed(packed, off, limit)
    assert np.array_equal(counts_t, counts) and np.array_equal(np.where(live[:, :, None], rows_t, 0), np.where(live[:, :, None], rows, 0))
    return flags


def _mixed(hay, off, n_q, seed):
    """Edited haystack strings plus the awkward ones: whole long strings (16 and more trigrams: the byte-counter launch),
    gibberish, empty, one letter."""
    q, qo = W.queries(hay, off, n_q, seed)
    needles = W.unpack(q, qo)
    rng = np.random.default_rng(seed)
    lens = (off[1:] - off[:-1]).astype(np.int64)
    for i in np.argsort(lens)[-300:]:
        needles.append(bytes(hay[int(off[i]):int(off[i + 1])]))
    for _ in range(300):
        needles.append(bytes(rng.choice(list(b"qxzjkvw"), size=int(rng.integers(1, 9))).tolist()))
    needles += [b"", b"a", b" ", b"zzzzzzzz", b"e", b"the", b"a" * 70, b"abcdefghijklmnopqrstuvwxyz" * 6]
    order = rng.permutation(len(needles))
    return _pack([needles[i] for i in order])


@pytest.mark.parametrize("limit", [1, 10, 37, 64])
def test_words_all_rows_vs_oracle(limit):
    hay, off = W.words(200000, 61)                              # 4 windows
    m, o = _pair(hay, off)
    m.sync_device()
    assert m.device_info()["n_windows"] <= 8
    q, qo = _mixed(hay, off, 12000, 62)
    flags = _check(m, o, q, qo, limit)
    assert (flags & SMALL).sum() > 8000 and (flags & COLD).sum() > 8000
    assert ((flags & SMALL) == 0).sum() > 100                   # the long needles went the other way
    m.close()


def test_hot_trigrams_floods_of_ties_and_eight_windows():
    """configs[4]'s kind of haystack at a size of eight windows: ties by the thousand at the cold start's bound (the
    candidate list overflows: every thread admits its own hits), pool overflows (the window swept again); eight
    windows: the sweep starts at the needle's own length class."""
    hay, off = W.skewed(500000, 63)
    m, o = _pair(hay, off)
    m.sync_device()
    assert m.device_info()["n_windows"] == 8
    q, qo = W.queries(hay, off, 4500, 64)
    for limit in (64, 10):
        flags = _check(m, o, q, qo, limit, upto=64)
        assert (flags & RESWEEP).any()
    m.close()


def test_limits_above_64_small_batches_and_nine_windows_are_left_alone():
    hay, off = W.words(200000, 65)
    m, o = _pair(hay, off)
    q, qo = W.queries(hay, off, 8000, 66)
    rows, counts = m.find_batch_packed(q, qo, 65)
    assert m.get_option("last_sweep") != 4
    q2, qo2 = W.queries(hay, off, 2000, 67)                     # under "small_min_needles"
    rows, counts = m.find_batch_packed(q2, qo2, 10)
    assert m.get_option("last_sweep") != 4
    m.set_option("small_sweep", 0)
    rows, counts = m.find_batch_packed(q, qo, 10)
    assert m.get_option("last_sweep") != 4
    m.close()
    hay, off = W.geonames(600000, 90000, 68)                    # 10 windows
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    rows, counts = m.find_batch_packed(q, qo, 10)
    assert m.device_info()["n_windows"] > 8 and m.get_option("last_sweep") != 4
    m.close()


def test_tombstones_and_pending_puts_under_the_small_sweep():
    """Deletes since the image was built (tombstone bits where a candidate is admitted: no cold start then) and puts since
    then (the delta image, merged in) -- storage.c:584-612, :398-473 semantics under this sweep."""
    hay, off = W.words(150000, 69)
    strings = W.unpack(hay, off)
    m, o = _pair(hay, off)
    q, qo = W.queries(hay, off, 6000, 70)
    _check(m, o, q, qo, 10)
    for ref in range(1, 20000, 7):
        assert m.delete(ref) == o.delete(ref)
    for k, s in enumerate(strings[:300]):
        assert m.put(s + b"x", 900000 + k, 0) == o.put(s + b"x", 900000 + k, 0)
    _check(m, o, q, qo, 10)
    assert m.device_info()["base_builds"] == 1
    m.close()


_FUZZ_FIRST = int(os.environ.get("BLURRILY_FUZZ_FIRST", "0"))           # soak runs: BLURRILY_FUZZ_FIRST=4 BLURRILY_FUZZ_SEEDS=40


@pytest.mark.parametrize("seed", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("BLURRILY_FUZZ_SEEDS", "4"))))
def test_randomised_small_images_both_new_sweeps(seed):
    """Seeded sweep over small images (one to eight windows) and batches large enough for the small-haystack sweep: the
    three generators, custom weights (ranks then do not follow length), sparse references, deletes before the first
    find (tombstones: no cold start) and puts after it (the delta image), any limit up to 64 -- and the same batch
    through the needle-major sweep that leaves slices out, forced, at limits up to 149."""
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([900, 30000, 120000, 300000, 500000]))
    gen = [lambda: W.geonames(n, max(500, n // 12), 150 + seed), lambda: W.words(n, 160 + seed),
           lambda: W.skewed(n, 170 + seed)][seed % 3]
    hay, off = gen()
    n = len(off) - 1
    refs = (np.sort(rng.choice(2**31 - 2, size=n, replace=False)) + 1).astype(np.uint32) if seed % 2 else \
        np.arange(1, n + 1, dtype=np.uint32)
    weights = None
    if seed % 4 >= 2:
        weights = rng.integers(0, 40, size=n).astype(np.uint32)
        weights[rng.random(n) < 0.3] = 0
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, refs, weights)
    if weights is None:
        o.put_many(hay, off, refs)
    else:
        for s_, r_, w_ in zip(W.unpack(hay, off), refs.tolist(), weights.tolist()):
            o.put(s_, r_, w_)
    if seed % 3 == 0:
        for r in rng.choice(refs, size=min(150, n // 4), replace=False).tolist():
            assert m.delete(r) == o.delete(r)
    q, qo = _mixed(hay, off, 4500, 180 + seed)
    limit = int(rng.choice([1, 5, 10, 33, 64]))
    _check(m, o, q, qo, limit)
    m.put(b"an entirely new entry", 2**31 - 1, 0); o.put(b"an entirely new entry", 2**31 - 1, 0)
    _check(m, o, q, qo, limit)
    # the same batch through the sweep that leaves slices out (forced; small "nm_dense": there are dense slices to leave)
    for k, v in dict(small_sweep=0, ws_autotune=0, wsweep=0, nm_min_windows=0, nm_cmin=int(rng.integers(1, 5)),
                     nm_dense=int(rng.choice([128, 512, 2048])), dense_min=128).items():
        m.set_option(k, v)
    limit2 = int(rng.choice([10, 64, 100, 149]))
    rows, counts = m.find_batch_packed(q, qo, limit2)
    assert m.get_option("last_sweep") == 3
    want = o.batch(q, qo, limit=limit2)
    assert np.array_equal(counts, want["counts"])
    live = np.arange(limit2)[None, :] < counts[:, None].astype(np.int64)
    assert np.array_equal(np.where(live[:, :, None], rows, 0), np.where(live[:, :, None], want["rows"], 0))
    m.close()
