"""GPU parity of the window-major sweep (find_kernels.hip: wsweep_kernel -- hot slices left out of the
count and consulted through bitmaps, one launch per window, needle states in global memory) against
the oracle, bit-exact, and against the needle-major sweep it replaces for large batches.

The path is taken for batches of >= 16 384 needles (option ws_min_needles) over >= 8 windows
(ws_min_windows) with limit <= 128 whose slices are big (ws_min_slice: the path pays a fixed price per
(needle, window)); the tests lower those bounds per map through blurrily_storage_set_option to reach it
with haystacks the oracle checks in seconds."""
import os

import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap
from blurrily_amd.map import _pack
from helpers import Oracle

pytestmark = pytest.mark.gpu


def _ws(m, **kw):
    """Lower the sweep's bounds on map `m` (blurrily_storage_set_option) so that it is reached with haystacks the
    oracle checks in seconds."""
    kw.setdefault("ws_min_slice", 0)                           # whatever the haystack's slice sizes
    kw.setdefault("ws_static_slice", 0)
    kw.setdefault("ws_autotune", 0)                            # ... and without measuring: the sweep it is
    kw.setdefault("small_sweep", 0)                            # (small images have a sweep of their own: tests/test_gpu_small.py)
    for k, v in kw.items():
        m.set_option(k, v)


def _pair(hay, off):
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    m.set_option("small_sweep", 0)                 # (small images: this file is about the window-major sweep and the choice)
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    o.put_many(hay, off)
    return m, o


def _check_all(m, o, packed, off, limit, took_ws=True, upto=None):
    m.set_stats(True)
    rows, counts = m.find_batch_packed(packed, off, limit)
    st = m.find_stats()
    m.set_stats(False)
    # the window-major path really ran (or did not): "last_sweep" says which sweep a large batch took; its own
    # units counter is the window-major sweep's alone (the needle-major sweep probes bitmaps too since round 4)
    assert (m.get_option("last_sweep") == 2) == took_ws, (m.get_option("last_sweep"), st)
    assert st["probes"] > 0 or not took_ws, st
    # (several limits over one batch: the oracle runs once at the largest, helpers.Oracle.batch_upto)
    want = o.batch(packed, off, limit=limit) if upto is None else o.batch_upto(packed, off, limit, upto)
    assert np.array_equal(counts, want["counts"])
    live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    bad = np.nonzero((np.where(live[:, :, None], rows, 0) != np.where(live[:, :, None], want["rows"], 0)).any(axis=(1, 2)))[0]
    if len(bad):
        q = int(bad[0])
        nd = bytes(packed[int(off[q]):int(off[q + 1])])
        raise AssertionError((len(bad), q, nd, rows[q, :counts[q]].tolist(), want["rows"][q, :counts[q]].tolist()))
    return rows, counts


def _mixed_needles(hay, off, n_q, seed):
    """Edited haystack strings plus the awkward ones: whole long strings (many trigrams: byte counters
    over window halves), gibberish (no threshold for a long time), empty, one letter, > 64 trigrams."""
    q, qo = W.queries(hay, off, n_q, seed)
    needles = W.unpack(q, qo)
    rng = np.random.default_rng(seed)
    lens = (off[1:] - off[:-1]).astype(np.int64)
    longest = np.argsort(lens)[-400:]
    for i in longest:
        needles.append(bytes(hay[int(off[i]):int(off[i + 1])]))
    for _ in range(300):
        needles.append(bytes(rng.choice(list(b"qxzjkvw"), size=int(rng.integers(1, 9))).tolist()))
    needles += [b"", b"a", b" ", b"zzzzzzzz", b"e", b"the", b"abcdefghij klmnopqrst uvwxyz abcdefgh ijklmnopqr stuvwxyz zyxwvutsrq ponmlkji"]
    order = rng.permutation(len(needles))
    return _pack([needles[i] for i in order])


@pytest.fixture(scope="module")
def medium():
    hay, off = W.geonames(600000, 80000, 41)                   # 10 windows
    m, o = _pair(hay, off)
    packed, offs = _mixed_needles(hay, off, 6000, 42)
    yield m, o, packed, offs
    m.close()


@pytest.mark.parametrize("limit,cmin", [(10, 2), (10, 1), (10, 3), (100, 2), (128, 2), (1, 2), (3, 2)])
def test_geonames_medium_all_rows_vs_oracle(limit, cmin, medium):
    m, o, packed, offs = medium                                # (one image, one oracle: "ws_cmin" is read per find)
    _ws(m, ws_min_windows=4, ws_min_needles=1000, ws_cmin=cmin)
    _check_all(m, o, packed, offs, limit, upto=128)


def test_window_major_equals_needle_major():
    hay, off = W.geonames(900000, 120000, 43)
    m, _ = _pair(hay, off)
    _ws(m, ws_min_windows=4, ws_min_needles=1000)
    packed, offs = _mixed_needles(hay, off, 30000, 44)
    a_rows, a_counts = m.find_batch_packed(packed, offs, 10)
    m.set_option("wsweep", 0)
    b_rows, b_counts = m.find_batch_packed(packed, offs, 10)
    assert np.array_equal(a_counts, b_counts)
    live = np.arange(10)[None, :] < a_counts[:, None].astype(np.int64)
    assert np.array_equal(np.where(live[:, :, None], a_rows, 0), np.where(live[:, :, None], b_rows, 0))


def test_skewed_ties_and_limit_100():
    """Massive (matches, weight) ties: floods of equally good candidates, pool overflows, re-sweeps."""
    hay, off = W.skewed(600000, 45)
    m, o = _pair(hay, off)
    _ws(m, ws_min_windows=4, ws_min_needles=1000)
    q, qo = W.queries(hay, off, 2500, 46)
    _check_all(m, o, q, qo, 100, upto=100)
    _check_all(m, o, q, qo, 10, upto=100)


def test_words_many_windows():
    hay, off = W.words(400000, 47)                             # 7 windows of single words
    m, o = _pair(hay, off)
    _ws(m, ws_min_windows=4, ws_min_needles=1000)
    q, qo = W.queries(hay, off, 8000, 48)
    _check_all(m, o, q, qo, 10)


def test_tombstones_and_pending_puts_under_the_window_major_sweep():
    hay, off = W.geonames(500000, 60000, 49)
    m, o = _pair(hay, off)
    _ws(m, ws_min_windows=4, ws_min_needles=1000)
    q, qo = W.queries(hay, off, 4000, 50)
    _check_all(m, o, q, qo, 10)                                # builds the base image
    needles = W.unpack(q, qo)
    rng = np.random.default_rng(51)
    # delete the best match of many needles (tombstones on the base image), add new strings (delta image)
    rows, counts = m.find_batch_packed(q, qo, 10)
    victims = sorted({int(rows[i, 0, 0]) for i in rng.choice(len(needles), size=600, replace=False) if counts[i]})
    for ref in victims:
        assert m.delete(ref) == o.delete(ref)
    n = len(off) - 1
    for j, i in enumerate(rng.choice(len(needles), size=300, replace=False)):
        ref = n + 1 + j
        assert m.put(needles[i], ref, 0) == o.put(needles[i], ref, 0)
    info = m.device_info()
    _check_all(m, o, q, qo, 10)
    info = m.device_info()
    assert info["base_builds"] == 1 and info["n_tombstones"] == len(victims) and info["n_pending"] == 300


def test_small_batches_and_small_haystacks_keep_the_needle_major_path():
    hay, off = W.geonames(600000, 80000, 41)
    m, o = _pair(hay, off)
    _ws(m)
    q, qo = W.queries(hay, off, 500, 52)                       # default bounds: 500 needles is a small batch
    _check_all(m, o, q, qo, 10, took_ws=False)


_FUZZ_FIRST = int(os.environ.get("BLURRILY_FUZZ_FIRST", "0"))           # soak runs: BLURRILY_FUZZ_FIRST=6 BLURRILY_FUZZ_SEEDS=40


@pytest.mark.parametrize("seed", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("BLURRILY_FUZZ_SEEDS", "6"))))
def test_randomised_window_major_configurations(seed):
    """Seeded random configurations of the window-major sweep -- haystack kind and size (5 to 11 windows),
    limit, cmin, reference numbering (dense / sparse), a sprinkle of deletes -- every row against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    kind = ["geonames", "skewed", "words"][seed % 3]
    n = int(rng.integers(300_000, 700_000))
    limit = int(rng.choice([1, 2, 7, 10, 33, 100, 128]))
    cmin = int(rng.integers(1, 5))
    hay, off = {"geonames": lambda: W.geonames(n, 50000, 60 + seed), "skewed": lambda: W.skewed(n, 60 + seed),
                "words": lambda: W.words(n, 60 + seed)}[kind]()
    refs = np.arange(1, n + 1, dtype=np.uint32)
    if seed % 2:
        refs = (np.sort(rng.choice(2**31 - 2, size=n, replace=False)) + 1).astype(np.uint32)   # sparse references
    m, o = RawMap(), Oracle()
    _ws(m, ws_min_windows=2, ws_min_needles=500, ws_cmin=cmin)
    m.put_many_packed(hay, off, refs)
    o.put_many(hay, off, refs)
    q, qo = W.queries(hay, off, 1500, 70 + seed)
    _check_all(m, o, q, qo, limit)
    for ref in rng.choice(refs, size=200, replace=False):       # tombstones on the base image
        assert m.delete(int(ref)) == o.delete(int(ref))
    _check_all(m, o, q, qo, limit)


@pytest.mark.parametrize("hot_pct", [0, 25, 50, 75, 100])
def test_both_sweeps_agree_along_the_gate(hot_pct):
    """The haystacks the sweep-choice gate was measured on (tools/gate_probe.py: the skewed generator with its hot
    prefixes / suffixes on 0 .. 100 per cent of the strings; mean_hit_slice from plain to hot-trigram), smaller:
    whichever sweep the gate picks, the rows are the same -- window-major against needle-major, row for row, and a
    sample against the oracle."""
    hay, off = W.skewed_mix(600000, hot_pct, 81)                # 10 windows
    m, o = _pair(hay, off)
    _ws(m, ws_min_windows=4, ws_min_needles=1000)
    q, qo = W.queries(hay, off, 12000, 82)
    for limit in (10, 100):
        m.set_option("wsweep", 1)
        m.set_stats(True)
        a_rows, a_counts = m.find_batch_packed(q, qo, limit)
        assert m.get_option("last_sweep") == 2
        assert m.find_stats()["probes"] > 0 or hot_pct == 0       # (bare stems: hardly a dense slice to leave out)
        m.set_stats(False)
        m.set_option("wsweep", 0)
        b_rows, b_counts = m.find_batch_packed(q, qo, limit)
        assert np.array_equal(a_counts, b_counts)
        live = np.arange(limit)[None, :] < a_counts[:, None].astype(np.int64)
        assert np.array_equal(np.where(live[:, :, None], a_rows, 0), np.where(live[:, :, None], b_rows, 0))
    idx = np.arange(0, 12000, 12, dtype=np.uint32)
    want = o.batch(q, qo, idx=idx, limit=100)
    assert np.array_equal(b_counts[idx], want["counts"])
    live = np.arange(100)[None, :] < want["counts"][:, None].astype(np.int64)
    assert np.array_equal(np.where(live[:, :, None], b_rows[idx], 0), np.where(live[:, :, None], want["rows"], 0))
    info = m.device_info()
    assert info["n_bitmaps"] > 0 and info["mean_hit_slice"] > 0


def test_the_sweep_is_chosen_by_measurement_and_the_static_rule_holds_without_it():
    """Default options.  Three sweeps can serve a large batch's short needles: needle-major (1), window-major (2: an
    image whose mean_hit_slice reaches "ws_min_slice", limits up to 128), needle-major with dense slices left out of a
    step's count (3: limits up to 64).  The first batch of a class (limit up to / above 32; by batch size, from 16 384
    needles) runs every sweep it can take and notes the fastest ("ws_choice", two bits per class); the rows are the
    oracle's whichever way.  With "ws_autotune" 0 the static rules of DESIGN.md section 5 apply: window-major by
    mean_hit_slice against "ws_static_slice" (x1.7 for a batch under 65 536 needles, x1.7 for a limit above 32, x4
    for both), slices left out from 256 windows on."""
    for gen, kw in ((W.skewed, dict(n=600000, seed=45)), (W.geonames, dict(n=600000, vocab=80000, seed=41))):
        hay, off = gen(**kw)
        m, o = _pair(hay, off)
        m.sync_device()
        info = m.device_info()
        mhs = info["mean_hit_slice"]
        eligible = mhs >= 1550 and info["n_windows"] >= 8
        assert info["n_bitmaps"] > 0 or not eligible, info
        # ---- measured choice ---------------------------------------------------------------------------
        assert m.get_option("ws_choice") == 0
        for n_q, limit, cls in ((70000, 10, 1), (20000, 10, 0), (20000, 100, 3), (8000, 10, None)):
            q, qo = W.queries(hay, off, n_q, 90)
            rows, counts = m.find_batch_packed(q, qo, limit)                  # (the class's first batch: both sweeps)
            choice = m.get_option("ws_choice")
            if cls is not None:
                picked = (choice >> (2 * cls)) & 3
                measured = eligible or (limit <= 149 and info["n_bitmaps"] > 0)     # something besides the plain sweep can run
                assert (picked != 0) == measured, (gen.__name__, n_q, limit, choice)
                assert picked != 2 or eligible
                assert picked != 3 or limit <= 149
            rows2, counts2 = m.find_batch_packed(q, qo, limit)                # (the chosen one)
            assert m.get_option("ws_choice") == choice
            live = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
            assert np.array_equal(counts, counts2)
            assert np.array_equal(np.where(live[:, :, None], rows, 0), np.where(live[:, :, None], rows2, 0))
            idx = np.arange(0, n_q, 50, dtype=np.uint32)
            want = o.batch(q, qo, idx=idx, limit=limit)
            assert np.array_equal(counts[idx], want["counts"])
            live_s = np.arange(limit)[None, :] < want["counts"][:, None].astype(np.int64)
            assert np.array_equal(np.where(live_s[:, :, None], rows[idx], 0), np.where(live_s[:, :, None], want["rows"], 0))
        m.set_option("ws_choice", 0)
        assert m.get_option("ws_choice") == 0
        # ---- the static rule ---------------------------------------------------------------------------
        m.set_option("ws_autotune", 0)
        for n_q, limit in ((70000, 10), (20000, 10), (70000, 100), (20000, 100), (8000, 10)):
            factor = (4.0 if limit > 32 else 1.7) if n_q < 65536 else (1.7 if limit > 32 else 1.0)
            expect_ws = eligible and n_q >= 16384 and mhs >= factor * 2200
            q, qo = W.queries(hay, off, n_q, 90)
            m.set_stats(True)
            m.find_batch_packed(q, qo, limit)
            took = m.get_option("last_sweep")
            m.set_stats(False)
            assert (took == 2) == expect_ws, (gen.__name__, mhs, n_q, limit, took)
            assert took != 3                                    # (fewer than 256 windows: nothing left out by the static rule)
        m.close()
