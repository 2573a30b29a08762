"""GPU parity: the HIP find path (through the C ABI) against the oracle, bit-exact.

Reads like spec/blurrily/map_spec.rb '#find' (:118-210) plus seeded random haystacks.
"""
import os

import numpy as np
import pytest

import workloads as W
from blurrily_amd import Map, RawMap
from helpers import Oracle, build_pair

pytestmark = pytest.mark.gpu


def test_device_present():
    m = Map()
    m.put("london", 1)
    m.sync_device()
    assert m.device_info()["device_ordinal"] >= 0


# ---- spec/blurrily/map_spec.rb:118-210 -------------------------------------------------

def test_empty_map_returns_no_results():
    assert Map().find("london", 10) == []


def test_empty_string_returns_no_results():
    assert Map().find("", 10) == []


def test_limit_option():
    m = Map()
    for idx in range(5):
        m.put("london", idx, 0)
    assert len(m.find("london", 2)) == 2


def test_duplicated_references():
    m = Map()
    m.put("london", 123)
    m.put("london2", 123)
    r = m.find("london", 10)
    assert len(r) == 1 and r[0][0] == 123


def test_perfect_match():
    m = Map()
    m.put("london", 123, 0)
    assert m.find("london")[0] == [123, 7, 6]


def test_favours_exact_matches():
    m = Map()
    m.put("lon", 125, 0)
    m.put("london city airport", 124, 0)
    m.put("london", 123, 0)
    assert m.find("london")[0][0] == 123


@pytest.mark.parametrize("needle", ["lonXdon", "lodon", "lodnon"])
def test_misspelt(needle):
    m = Map()
    m.put("london", 123, 0)
    assert m.find(needle) != []


def test_sorts_by_descending_matchiness():
    m = Map()
    m.put("New York", 1001, 0)
    m.put("Yorkshire", 1002, 0)
    m.put("York", 1003, 0)
    m.put("Yorkisthan", 1004, 0)
    assert m.find("York") == [[1003, 5, 4], [1001, 4, 8], [1002, 4, 9], [1004, 4, 10]]


def test_favours_lighter():
    m = Map()
    m.put("london", 103, 103)
    m.put("london", 101, 101)
    m.put("london", 102, 102)
    assert [r[0] for r in m.find("london")] == [101, 102, 103]


def test_tie_is_reference_ascending():
    """spec/integration_spec.rb:31-42."""
    m = Map()
    m.put("paris", 456)
    m.put("paris", 123)
    assert m.find("paris") == [[123, 6, 5], [456, 6, 5]]
    assert m.find("pariis")[0] == [123, 5, 5]


def test_command_processor_vector():
    """spec/blurrily/command_processor_spec.rb:15-19."""
    m = Map()
    m.put("great london", 12)
    m.put("greater masovian", 13)
    assert m.find("great") == [[12, 6, 12], [13, 5, 16]]


# ---- the reference's stress checks (spec/blurrily/map_spec.rb:355-403), at its own 1 024 iterations,
# ---- with the oracle replaying every step and the device image served by delta + tombstones --------
STRESS_COUNT = 1024          # map_spec.rb:361 "enough cycles to force reallocations"


def _stress_pair():
    from helpers import Oracle
    return Map(), Oracle()


def test_stress_puts():
    """map_spec.rb:363-367."""
    m, o = _stress_pair()
    for index in range(STRESS_COUNT):
        assert m.put("Port-au-Prince", index) == o.put(b"port au prince", index, 0)
    assert m.stats() == o.stats() and m.stats()["references"] == STRESS_COUNT
    assert m.find("Port-au-Prince") == o.find(b"port au prince", 10) != []
    assert m.find("Port-au-Prince", 1024) == o.find(b"port au prince", 1024)      # all of them, in order


def test_stress_put_delete_find():
    """map_spec.rb:369-376."""
    m, o = _stress_pair()
    for index in range(STRESS_COUNT):
        m.put("Port-au-Prince", index)
        o.put(b"port au prince", index, 0)
        assert m.delete(index) == o.delete(index)
        assert m.stats() == {"references": 0, "trigrams": 0}
        assert m.find("Port-au-Prince") == []
    assert m.device_info()["base_builds"] == 1            # served by the mutation log, not by rebuilds


def test_stress_put_find_delete():
    """map_spec.rb:377-384."""
    m, o = _stress_pair()
    for index in range(STRESS_COUNT):
        m.put("Port-au-Prince", index)
        o.put(b"port au prince", index, 0)
        assert m.stats()["references"] == 1
        rows = m.find("Port-au-Prince")
        assert rows == o.find(b"port au prince", 10) and rows[0][0] == index
        m.delete(index)
        o.delete(index)
    info = m.device_info()
    assert info["base_builds"] == 1 and info["n_tombstones"] == 1      # only reference 0 ever reached the base image


def test_stress_puts_then_many_deletes():
    """map_spec.rb:386-391, with a find between the deletes so that every tombstone is exercised."""
    m, o = _stress_pair()
    for index in range(STRESS_COUNT):
        m.put("Port-au-Prince", index)
        o.put(b"port au prince", index, 0)
    assert m.find("Port-au-Prince", 5) == o.find(b"port au prince", 5)     # base image holds all 1 024
    for index in range(STRESS_COUNT):
        assert m.delete(index) == o.delete(index)
        if index % 16 == 0 or index > STRESS_COUNT - 8:
            assert m.find("Port-au-Prince", 20) == o.find(b"port au prince", 20)
    assert m.stats() == {"references": 0, "trigrams": 0}
    assert m.find("Port-au-Prince") == []
    assert m.device_info()["base_builds"] == 1


def test_stress_puts_reload_many_deletes(tmp_path):
    """map_spec.rb:393-403."""
    m, o = _stress_pair()
    for index in range(STRESS_COUNT):
        m.put("Port-au-Prince", index)
        o.put(b"port au prince", index, 0)
    path = str(tmp_path / "stress.trigrams")
    m.save(path)
    m = Map.load(path)
    assert m.find("Port-au-Prince", 7) == o.find(b"port au prince", 7)
    for index in range(STRESS_COUNT):
        assert m.delete(index) == o.delete(index)
        if index % 64 == 0:
            assert m.find("Port-au-Prince", 7) == o.find(b"port au prince", 7)
    assert m.stats() == {"references": 0, "trigrams": 0}
    assert m.find("Port-au-Prince") == []
    assert m.device_info()["base_builds"] == 1


def test_stress_put_save_load_cycles(tmp_path):
    """map_spec.rb:407-437 (100 iterations): cold loads, put/save/load/delete, put/save/load."""
    path = str(tmp_path / "cycle.trigrams")
    m = Map()
    for index in range(100):
        m.put("Port-au-Prince", index)
        m.save(path)
        m = Map.load(path)
        assert m.stats()["references"] == index + 1
    assert [r[0] for r in m.find("Port-au-Prince", 100)] == list(range(100))
    m = Map()
    for index in range(100):
        m.put("Port-au-Prince", index)
        m.save(path)
        m = Map.load(path)
        assert m.find("Port-au-Prince")[0][0] == index
        m.delete(index)
        assert m.stats()["references"] == 0 and m.find("Port-au-Prince") == []


# ---- seeded random haystacks vs the oracle ------------------------------------------------

def _check_batch(m, o, needles, limit):
    packed = b"".join(needles)
    off = np.zeros(len(needles) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in needles])
    rows, counts = m.find_batch_packed(packed, off, limit)
    for i, nd in enumerate(needles):
        want = o.find(nd, limit)
        got = rows[i, :counts[i]].tolist()
        assert got == want, (nd, limit, got[:5], want[:5])


@pytest.mark.parametrize("n,limit", [(1, 10), (50, 3), (3000, 10), (70000, 10), (140000, 100)])
def test_words_vs_oracle(n, limit):
    hay, off = W.words(n, seed=100 + n)
    strings = W.unpack(hay, off)
    m, o = build_pair(strings)
    q, qo = W.queries(hay, off, 400, seed=7)
    _check_batch(m, o, W.unpack(q, qo) + [b"", b"zzzz", b"a"], limit)


def test_sparse_refs_weights_and_ties():
    rng = np.random.default_rng(5)
    hay, off = W.skewed(30000, seed=9)
    strings = W.unpack(hay, off)
    refs = rng.choice(2**31 - 1, size=len(strings), replace=False).astype(np.int64) + 1
    weights = rng.integers(0, 4, size=len(strings))           # 0 -> strlen, else tiny: massive ties
    m, o = build_pair(strings, refs.tolist(), weights.tolist())
    q, qo = W.queries(hay, off, 300, seed=10)
    for limit in (1, 10, 100, 1024):
        _check_batch(m, o, W.unpack(q, qo)[:100], limit)


def test_large_limits_multi_pass():
    hay, off = W.skewed(5000, seed=19)
    m, o = build_pair(W.unpack(hay, off))
    q, qo = W.queries(hay, off, 20, seed=20)
    for limit in (1025, 3000, 65535):
        _check_batch(m, o, W.unpack(q, qo), limit)


def test_long_needles_use_wide_counters():
    hay, off = W.geonames(20000, 3000, seed=3)
    strings = W.unpack(hay, off)
    m, o = build_pair(strings)
    long1 = b" ".join(strings[:40])            # > 127 distinct trigrams
    long2 = (b"abcdefghijklmnopqrstuvwxyz " * 40)[:1000]
    _check_batch(m, o, [long1, long2, strings[0], b"x" * 300], 10)
    _check_batch(m, o, [long1, long2], 300)


def test_arbitrary_bytes_needles():
    hay, off = W.words(5000, seed=77)
    m, o = build_pair(W.unpack(hay, off))
    needles = [b"Lond\xc3\xa9n", b"a*b c", b"***", b"  ", b"UPPER lower", bytes(range(1, 60))]
    _check_batch(m, o, needles, 10)


def test_interleaved_mutations_are_served_by_delta_image_and_tombstones():
    """put/delete after the device image exists (storage.c:398-473, :584-612 semantics): small
    logs must not trigger a rebuild, and results stay bit-exact -- including a reference that
    is deleted and put again with another string."""
    rng = np.random.default_rng(21)
    hay, off = W.words(60000, seed=31)
    strings = W.unpack(hay, off)
    m, o = RawMap(), Oracle()
    refs = np.arange(1, len(strings) + 1, dtype=np.uint32)
    m.put_many_packed(hay, off, refs)
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, 120, seed=32)
    needles = W.unpack(q, qo)
    _check_batch(m, o, needles, 10)
    assert m.device_info()["base_builds"] == 1
    extra, eo = W.words(4000, seed=33)
    extra = W.unpack(extra, eo)
    m2, o2 = m, o
    live = set(int(r) for r in refs)
    used_extra = 0
    for rnd in range(10):
        for _ in range(30):
            r = int(rng.choice(sorted(live)))
            assert m2.delete(r) == o2.delete(r) > 0
            live.discard(r)
        for _ in range(50):
            s = extra[used_extra]; used_extra += 1
            r = (10**6 + used_extra) if rng.random() < 0.7 else int(rng.integers(1, 60001))
            w = int(rng.integers(0, 3))
            a = m2.put(s, r, w)
            assert a == o2.put(s, r, w)
            if a:
                live.add(r)
        assert m2.stats() == o2.stats()
        _check_batch(m2, o2, needles + extra[:used_extra][-20:], 10)
        _check_batch(m2, o2, needles[:10], 100)
    info = m2.device_info()
    assert info["base_builds"] == 1 and info["n_pending"] > 0 and info["n_tombstones"] > 0
    # a log past its budget folds into a rebuilt base image
    big, bo = W.words(9000, seed=34)
    m2.put_many_packed(big, bo, np.arange(2 * 10**6, 2 * 10**6 + 9000, dtype=np.uint32))
    for i, s in enumerate(W.unpack(big, bo)):
        o2.put(s, 2 * 10**6 + i, 0)
    _check_batch(m2, o2, needles[:40], 10)
    info = m2.device_info()
    assert info["base_builds"] == 2 and info["n_pending"] == 0 and info["n_tombstones"] == 0


def test_value_range_edges():
    """Extremes of the reference's validated ranges (lib/blurrily/defaults.rb:7-9): reference 0
    and 2^31-1, weights up to 2^31-1, strings that are empty or all non-letters, limit 0 and the
    largest uint16_t limit, a needle cut at an embedded NUL."""
    strings = [b"", b"   ", b"a", b"zz", b"london", b"london", b"londres", b"l", b"***", b"lon don"]
    refs = [0, 1, 2, 2**31 - 1, 77, 78, 2**30, 5, 6, 7]
    weights = [0, 2**31 - 1, 1, 0, 2**31 - 1, 1, 0, 0, 3, 2]
    m, o = build_pair(strings, refs, weights)
    needles = [b"", b"london", b"lon", b"l", b"zz", b"a", b" ", b"don lon", b"londonlondon"]
    for limit in (1, 2, 10, 65535):
        _check_batch(m, o, needles, limit)
    # limit 0 (storage.c:569: min(0, nb_matches)) through the C ABI
    rows = (np.zeros((4, 3), dtype=np.uint32))
    assert m._lib.blurrily_storage_find(m.handle, b"london", 0, rows.ctypes.data) == 0
    # a batched needle is cut at its first NUL, like the C string the reference sees
    packed = np.frombuffer(b"london\0garbage" + b"lon", dtype=np.uint8)
    off = np.array([0, 14, 17], dtype=np.uint64)
    got, counts = m.find_batch_packed(packed, off, 10)
    assert got[0, :counts[0]].tolist() == o.find(b"london", 10)
    assert got[1, :counts[1]].tolist() == o.find(b"lon", 10)


def test_very_long_needles_and_haystack_strings():
    """Thousands of distinct trigrams in one needle (16-bit counters, staged slice table) against a
    haystack that also holds very long strings."""
    rng = np.random.default_rng(8)
    hay, off = W.geonames(30000, 4000, seed=13)
    strings = W.unpack(hay, off)
    long_strings = [b" ".join(strings[i:i + 300]) for i in range(0, 3000, 300)]       # ~4000 chars each
    all_strings = strings + long_strings
    m, o = build_pair(all_strings)
    needles = [long_strings[0], long_strings[3][:1500], b" ".join(strings[5000:5400]),
               bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz ", dtype=np.uint8), size=6000).tolist()),
               strings[10], b"x" * 5000]
    assert max(len(Oracle.tokenise(nd)) for nd in needles) > 1000
    _check_batch(m, o, needles, 10)
    _check_batch(m, o, needles[:3], 700)


@pytest.mark.parametrize("n", [65519, 65520, 65521, 65535, 65536, 131039, 131040, 131041])
def test_window_boundaries_and_cross_window_ties(n):
    """Haystack sizes around the 65 520-rank window (and the 65 535 it used to be): every string ties with every other on
    (matches, weight), so the order is purely reference-ascending across window boundaries,
    including limits that cross a window and the multi-pass path."""
    m, o = RawMap(), Oracle()
    strings = [b"london"] * n
    refs = np.arange(1, n + 1, dtype=np.uint32)
    packed = np.frombuffer(b"".join(strings), dtype=np.uint8)
    off = (np.arange(n + 1, dtype=np.uint64) * 6)
    m.put_many_packed(packed, off, refs)
    o.put_many(packed, off, refs)
    m.put("paris", n + 10, 0); o.put(b"paris", n + 10, 0)          # lives alone at the end of the ranks
    for limit in (1, 10, 1024, 1025, 65535):
        _check_batch(m, o, [b"london", b"londno", b"paris", b"pa"], limit)


def test_dense_and_empty_windows_mixed():
    """Windows whose slices hold more units than the kernel's unit ring (one string repeated
    70 000 times) next to ordinary windows and windows with nothing for the needle: the
    ring-overflow walk, the ring path and the skipped steps in one sweep."""
    hay, off = W.words(150000, seed=21)
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    refs = np.arange(1, n + 1, dtype=np.uint32)
    m.put_many_packed(hay, off, refs)
    o.put_many(hay, off, refs)
    rep = 70000
    packed = np.frombuffer(b"london" * rep, dtype=np.uint8)
    roff = np.arange(rep + 1, dtype=np.uint64) * 6
    rrefs = np.arange(n + 1, n + rep + 1, dtype=np.uint32)
    m.put_many_packed(packed, roff, rrefs)
    o.put_many(packed, roff, rrefs)
    m.put("zzzzqqqqzzzzqqqqzzzzqqqq", 10**6, 0); o.put(b"zzzzqqqqzzzzqqqqzzzzqqqq", 10**6, 0)
    needles = [b"london", b"londno", b"lon", b"zzzzqqqq", b"zq", b"don"] + W.unpack(hay, off)[:200]
    for limit in (1, 10, 300):
        _check_batch(m, o, needles, limit)


def test_every_batch_size_regime_on_a_many_window_haystack():
    """Eleven windows: a handful of needles is cut into window-pair ranges over many workgroups
    (latency mode, byte and 4-bit counters alike), a few hundred into fewer ranges, a thousand
    run whole -- the same needles must give the same rows in every regime."""
    hay, off = W.geonames(700000, 60000, seed=31)
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    refs = np.arange(1, n + 1, dtype=np.uint32)
    m.put_many_packed(hay, off, refs)
    o.put_many(hay, off, refs)
    q, qo = W.queries(hay, off, 1100, seed=12)
    needles = W.unpack(q, qo)
    assert m.device_info()["n_windows"] is not None
    packed = b"".join(needles)
    offs = np.zeros(len(needles) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in needles])
    whole_rows, whole_counts = m.find_batch_packed(packed, offs, 10)          # 1100 needles: whole needles
    assert m.device_info()["n_windows"] == 11
    for i in range(0, 300):                                                     # the oracle on a sample
        assert whole_rows[i, :whole_counts[i]].tolist() == o.find(needles[i], 10), needles[i]
    for size in (1, 2, 7, 40, 97, 130, 300, 512, 600):
        sub = needles[:size]
        p = b"".join(sub)
        so = np.zeros(size + 1, dtype=np.uint64)
        so[1:] = np.cumsum([len(x) for x in sub])
        rows, counts = m.find_batch_packed(p, so, 10)
        assert np.array_equal(counts, whole_counts[:size]), size
        assert np.array_equal(rows, whole_rows[:size]), size
    _check_batch(m, o, needles[:60], 100)
    _check_batch(m, o, needles[:5], 1000)


def test_long_needles_over_short_reference_windows():
    """Needles with 16..64 distinct trigrams sweep the windows of short references (<= 15 trigrams
    each) with 4-bit counters, two windows per step, and the rest with byte counters: both kinds in
    one sweep, every trigram of the needle counted in both."""
    rng = np.random.default_rng(41)
    hay, off = W.geonames(400000, 40000, seed=29)            # 7 windows, the first ones all short strings
    strings = W.unpack(hay, off)
    n = len(strings)
    m, o = RawMap(), Oracle()
    refs = np.arange(1, n + 1, dtype=np.uint32)
    m.put_many_packed(hay, off, refs)
    o.put_many(hay, off, refs)
    needles = []
    for _ in range(150):                                       # several haystack strings glued together
        k = int(rng.integers(2, 6))
        needles.append(b" ".join(strings[int(i)] for i in rng.integers(0, n, size=k))[:int(rng.integers(16, 64))])
    assert min(len(set(Oracle.tokenise(nd))) for nd in needles) >= 10
    assert max(len(set(Oracle.tokenise(nd))) for nd in needles) > 40
    _check_batch(m, o, needles, 10)                            # latency mode (ranges)
    _check_batch(m, o, needles[:40], 120)
    big = needles * 8                                          # 1200 needles: whole needles per workgroup
    packed = b"".join(big)
    offs = np.zeros(len(big) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in big])
    rows, counts = m.find_batch_packed(packed, offs, 10)
    for i, nd in enumerate(needles):
        assert rows[i, :counts[i]].tolist() == o.find(nd, 10), nd
        assert np.array_equal(rows[i], rows[i + len(needles)])


_FUZZ_FIRST = int(os.environ.get("BLURRILY_FUZZ_FIRST", "0"))           # soak runs: BLURRILY_FUZZ_FIRST=10 BLURRILY_FUZZ_SEEDS=50


@pytest.mark.parametrize("seed", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("BLURRILY_FUZZ_SEEDS", "10"))))
def test_randomised_configurations(seed):
    """Seeded sweep over what a haystack and a batch can look like: size (one window to several), the
    three generators, custom weights (ranks then do not follow length, so the 4-bit window prefix is
    short or empty), sparse references, deletes before and after the first find, any limit, short
    and long needles, a handful of needles (ranges) and a thousand (whole needles)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([900, 40000, 70000, 150000, 260000]))
    gen = [lambda: W.geonames(n, max(500, n // 12), 50 + seed), lambda: W.words(n, 60 + seed),
           lambda: W.skewed(n, 70 + seed)][seed % 3]
    hay, off = gen()
    strings = W.unpack(hay, off)
    n = len(strings)
    refs = (rng.choice(2**31 - 2, size=n, replace=False) + 1).astype(np.uint32) if seed % 2 else \
        np.arange(1, n + 1, dtype=np.uint32)
    weights = None
    if seed % 4 >= 2:                                          # custom weights, zeros (-> strlen) mixed in
        weights = rng.integers(0, 40, size=n).astype(np.uint32)
        weights[rng.random(n) < 0.3] = 0
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, refs, weights)
    if weights is None:
        o.put_many(hay, off, refs)
    else:
        for s, r, w in zip(strings, refs.tolist(), weights.tolist()):
            o.put(s, r, w)
    for r in rng.choice(refs, size=min(200, n // 4), replace=False).tolist():    # deletes before the first find
        assert m.delete(r) == o.delete(r)
    q, qo = W.queries(hay, off, 110, seed=80 + seed)
    needles = W.unpack(q, qo) + [b"", b"q", strings[0] + b" " + strings[-1] + b" " + strings[n // 2]]
    # (needles of more than 64 and of more than 127 distinct trigrams: the byte-counter and the 16-bit launches behind the
    # first one -- in the small batches below as well, where round 6 found the ranged launch waiting for ever)
    needles += [b" ".join(strings[(7 * k + seed) % n] for k in range(12))[:120], b" ".join(strings[(11 * k + seed) % n] for k in range(30))[:250]]
    limit = int(rng.choice([1, 2, 10, 33, 100, 257]))
    _check_batch(m, o, needles, limit)
    for r in rng.choice(refs, size=min(50, n // 8), replace=False).tolist():     # and after (tombstones)
        assert m.delete(r) == o.delete(r)
    m.put(b"an entirely new entry", 2**31 - 1, 0); o.put(b"an entirely new entry", 2**31 - 1, 0)
    _check_batch(m, o, needles[:20] + needles[-2:] + [b"an entirely new"], limit)
    k = int(rng.integers(25, 225))                             # latency mode under mutation: tombstones in its scans, the delta image beside it
    some = (needles * 2)[:k]                                   # (up to 128 needles over the pinned page, beyond by the batch's copies)
    packed = b"".join(some)
    offs = np.zeros(len(some) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in some])
    rows, counts = m.find_batch_packed(packed, offs, limit)
    for i, nd in enumerate(some[:len(needles)]):
        assert rows[i, :counts[i]].tolist() == o.find(nd, limit), (nd, limit, k)
    for i in range(len(needles), k):                           # (the second copy of a needle says what the first says)
        assert rows[i, :counts[i]].tolist() == rows[i - len(needles), :counts[i - len(needles)]].tolist()
    big = needles * 10                                         # > 1000 needles: whole needles per workgroup
    packed = b"".join(big)
    offs = np.zeros(len(big) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in big])
    rows, counts = m.find_batch_packed(packed, offs, limit)
    for i, nd in enumerate(needles):
        assert rows[i, :counts[i]].tolist() == o.find(nd, limit), (nd, limit)
