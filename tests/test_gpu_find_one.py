"""blurrily_storage_find as ONE launch (c_abi.hip: find_one, find_kernels.hip: find_one_kernel) -- the reference's
only call shape (ext/blurrily/map_ext.c:131-162 -> storage.c:477-580) -- row for row against the live compiled
reference (oracle/_ref, where it travelled) or the oracle, and against the batch's way on the same map:
  * hypothesis-generated needles at limits 1 .. 120 on a haystack of five windows full of duplicate words (ties);
  * thousands of IDENTICAL strings: a window whose counters tie by the thousand at the bound (storage.c:566's order:
    the lowest references win);
  * several windows per workgroup (option "one_windows_per_wg": the steps after the first arrive with a threshold);
  * mutations the device image has not absorbed: tombstones inside the select, pending puts by a second launch over
    the delta image, merged on the host;
  * what must NOT take the launch: limits of 0 and above 120, needles of more than 64 distinct trigrams, timing mode
    -- same rows, by the batch's way ("one_taken" tells which way a find went)."""
import os
import tempfile

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle, Reference

pytestmark = pytest.mark.gpu


def _find(m, needle, limit):
    rows = np.zeros((max(limit, 1), 3), dtype=np.uint32)
    n = m._lib.blurrily_storage_find(m.handle, needle, limit, rows.ctypes.data)
    assert n >= 0
    return rows[:n].tolist()


class _Checker:
    """the live reference over a saved file when oracle/_ref is there, else the oracle"""

    def __init__(self, m, hay, off):
        self.ref, self.o, self.dir = None, None, None
        if Reference.available():
            self.dir = tempfile.TemporaryDirectory()
            path = os.path.join(self.dir.name, "h.trigrams")
            m.save(path)
            self.ref = Reference(path)
        else:
            self.o = Oracle()
            self.o.put_many(hay, off)

    def find(self, needle, limit):
        return self.ref.find(needle, limit) if self.ref else self.o.find(needle, limit)


@pytest.fixture(scope="module")
def geo():
    n = 300_000                                              # five windows; 3 000 words: every string has twins
    hay, off = W.geonames(n, 3000, 41)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    m.sync_device()
    chk = _Checker(m, hay, off)
    strings = W.unpack(hay, off)
    return m, chk, strings


def test_every_way_in_gives_the_reference_rows(geo):
    m, chk, strings = geo
    q, qo = W.queries(*_pack(strings[:4000]), 300, 5)
    taken0 = m.get_option("one_taken")
    n_one = 0
    for limit in (1, 2, 10, 64, 100, 120):
        for nd in W.unpack(q, qo)[:120] + [strings[7], strings[70000], strings[299999], b"", b"a", b"zzzzqqqq"]:
            want = chk.find(nd, limit)
            assert _find(m, nd, limit) == want, (nd, limit)
            n_one += 1
    assert m.get_option("one_taken") - taken0 >= n_one - 6 * 3      # (needles without a posting return before any launch)
    # the batch's way gives the same rows (one_launch 0)
    m.set_option("one_launch", 0)
    t = m.get_option("one_taken")
    for nd in W.unpack(q, qo)[:40]:
        assert _find(m, nd, 10) == chk.find(nd, 10)
    assert m.get_option("one_taken") == t
    m.set_option("one_launch", 1)


def _pack(strings):
    packed = np.frombuffer(b"".join(strings), dtype=np.uint8)
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return packed, off


@settings(max_examples=int(os.environ.get("BLURRILY_FUZZ_ONE", "150")), deadline=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(needle=st.text(alphabet="abcdefghijklmnopqrstuvwxyz ", min_size=0, max_size=48),
       pick=st.integers(0, 299_999), edits=st.integers(0, 3), limit=st.sampled_from([1, 3, 10, 37, 64, 120]),
       use_pick=st.booleans())
def test_hypothesis_single_finds_against_the_reference(geo, needle, pick, edits, limit, use_pick):
    m, chk, strings = geo
    nd = needle.encode()
    if use_pick:                                             # a haystack string, a few characters dropped: close matches and ties
        nd = bytearray(strings[pick])
        for k in range(min(edits, max(len(nd) - 1, 0))):
            del nd[(pick * 7 + k * 13) % len(nd)]
        nd = bytes(nd)
    assert _find(m, nd, limit) == chk.find(nd, limit), (nd, limit)


def test_thousands_of_identical_strings_tie_at_the_bound():
    """windows whose counters tie by the thousand: the lowest references win (storage.c:566 over ref-sorted
    input, spec/integration_spec.rb:37-42), whichever window they sit in"""
    rng = np.random.default_rng(3)
    base = [b"san jose", b"san juan", b"santa fe", b"saint john", b"sao jose dos campos"]
    hay_strings = [base[i % 5] if (i % 3) else bytes(rng.integers(97, 123, size=int(rng.integers(3, 14))).astype(np.uint8))
                   for i in range(200_000)]
    packed, off = _pack(hay_strings)
    refs = rng.permutation(np.arange(1, 200_001)).astype(np.uint32)          # references in no order: ranks decide
    m, o = RawMap(), Oracle()
    m.put_many_packed(packed, off, refs)
    o.put_many(packed, off, refs)
    taken0 = m.get_option("one_taken")
    for nd in (b"san jose", b"san juan", b"san", b"saint", b"jose", b"sao jose", b"s"):
        for limit in (1, 10, 64, 120):
            assert _find(m, nd, limit) == o.find(nd, limit), (nd, limit)
    assert m.get_option("one_taken") - taken0 == 7 * 4
    for per in (2, 3, 64):                                   # several windows per workgroup: steps that arrive with a threshold
        m.set_option("one_windows_per_wg", per)
        for nd in (b"san jose", b"santa", b"saint john"):
            for limit in (1, 10, 120):
                assert _find(m, nd, limit) == o.find(nd, limit), (nd, limit, per)
    m.set_option("one_windows_per_wg", 0)


def test_what_goes_the_batchs_way_instead(geo):
    m0, chk, strings = geo
    hay_strings = strings[:30_000]
    packed, off = _pack(hay_strings)
    m, o = RawMap(), Oracle()
    m.put_many_packed(packed, off, np.arange(1, 30_001, dtype=np.uint32))
    o.put_many(packed, off)
    nd = hay_strings[11]

    def goes_one(needle, limit):
        t = m.get_option("one_taken")
        assert _find(m, needle, limit) == o.find(needle, limit), (needle, limit)
        return m.get_option("one_taken") - t

    assert goes_one(nd, 10) == 1
    assert goes_one(nd, 120) == 1
    assert goes_one(nd, 121) == 0 and goes_one(nd, 1024) == 0               # limits above the launch's
    assert _find(m, nd, 0) == []                                            # limit 0: no rows (the C entry point takes it as it is)
    long_needle = b" ".join(hay_strings[k] for k in range(20, 40))[:250]    # more than 64 distinct trigrams
    assert len(set(Oracle.tokenise(long_needle))) > 64 and goes_one(long_needle, 10) == 0
    # timing mode describes the batch's launches: not taken
    m.set_timing(True)
    assert goes_one(nd, 10) == 0
    m.set_timing(False)


def test_mutations_the_image_has_not_absorbed(geo):
    """Deletes since the base image was built are tombstone bits the select looks at (a deleted reference takes no place:
    storage.c:584-612), puts since then live in the delta image, searched by a second launch and merged on the host
    (storage.c:398-473) -- the single find stays ONE launch per image, and says what the reference says."""
    m0, chk, strings = geo
    hay_strings = strings[:150_000]                            # three windows, full of twins
    packed, off = _pack(hay_strings)
    m, o = RawMap(), Oracle()
    m.put_many_packed(packed, off, np.arange(1, 150_001, dtype=np.uint32))
    o.put_many(packed, off)
    rng = np.random.default_rng(12)
    needles = [hay_strings[int(k)] for k in rng.integers(0, 150_000, size=60)]
    needles += [nd[:-1] for nd in needles[:20]] + [b" ".join(hay_strings[k] for k in range(300, 303))[:60]]   # (the last: over 15 trigrams)
    for nd in needles[:5]:
        assert _find(m, nd, 10) == o.find(nd, 10)              # builds the base image
    # delete the best matches of the needles (what a find would return first), and a spread of others
    victims = set()
    for nd in needles:
        for row in o.find(nd, 3):
            victims.add(row[0])
    victims |= set(range(7, 150_000, 997))
    for ref in sorted(victims):
        assert m.delete(ref) == o.delete(ref)
    # new strings: twins of haystack strings (ties with base references) and fresh ones
    for k, i in enumerate(rng.integers(0, 150_000, size=200)):
        s_new = hay_strings[int(i)] if k % 2 else hay_strings[int(i)] + b" nova"
        assert m.put(s_new, 200_000 + k, 0) == o.put(s_new, 200_000 + k, 0)
    info = m.device_info()
    taken = m.get_option("one_taken")
    for limit in (1, 10, 64, 120):
        for nd in needles:
            assert _find(m, nd, limit) == o.find(nd, limit), (nd, limit)
    assert m.get_option("one_taken") - taken == 4 * len(needles)            # every one of them one launch per image
    info = m.device_info()
    assert info["base_builds"] == 1 and info["n_tombstones"] > 0 and info["n_pending"] == 200
    # a handful at a time, and several windows per workgroup
    m.set_option("one_windows_per_wg", 2)
    pk, po = _pack(needles[:16])
    rows, counts = m.find_batch_packed(pk, po, 10)
    for i, nd in enumerate(needles[:16]):
        assert rows[i, :counts[i]].tolist() == o.find(nd, 10), nd
    m.set_option("one_windows_per_wg", 0)
    # ... and a few dozen: the base image in latency mode over the pinned page (tombstones in its scans), the delta image by
    # find_one_kernel, a needle's two lists merged on the host
    pk, po = _pack(needles[:40])
    for limit in (10, 64):
        rows, counts = m.find_batch_packed(pk, po, limit)
        assert "merge_parts_pinned_kernel" in m.last_kernels()
        for i, nd in enumerate(needles[:40]):
            assert rows[i, :counts[i]].tolist() == o.find(nd, limit), (nd, limit)
    # delete a pending put, put a deleted reference back: the log keeps up
    assert m.delete(200_001) == o.delete(200_001)
    back = sorted(victims)[0]
    assert m.put(b"zanzibar city", back, 0) == o.put(b"zanzibar city", back, 0)
    for nd in needles[:30] + [b"zanzibar city"]:
        assert _find(m, nd, 10) == o.find(nd, 10), nd
    m.close()


def test_a_handful_of_needles_share_one_launch(geo):
    """blurrily_storage_find_batch with up to "few_max" needles (24; the kernel takes up to 128): one launch, a row of the
    grid per needle (c_abi.hip: find_few) -- each element still exactly one blurrily_storage_find; needles without a
    posting take no row.  Up to sixteen needles travel as kernel arguments and the row's last workgroup merges; more are
    read from the pinned page and the workgroup that finishes last merges (tickets).  Beyond few_max and up to "mid_max"
    (128) needles: latency mode's ranges with the needles read from the pinned page and the merged rows written back
    into it (two launches, no copy); a limit beyond 120 goes the batch's way, copies and all."""
    m, chk, strings = geo
    rng = np.random.default_rng(8)
    few_max = m.get_option("few_max")
    assert few_max == 24
    for n, limit in [(2, 10), (5, 1), (16, 10), (16, 120), (9, 64), (17, 10), (24, 10), (24, 120), (25, 10), (32, 10), (3, 121)]:
        picks = [strings[int(k)] for k in rng.integers(0, len(strings), size=n)]
        needles = [p[: max(1, len(p) - 1)] for p in picks]
        if n >= 5:
            needles[1] = b""                                   # no posting: no rows, no row of the grid
            needles[3] = b"qqqqzzzzxxxx"
        packed, off = _pack(needles)
        taken = m.get_option("one_taken")
        rows, counts = m.find_batch_packed(packed, off, limit)
        went = m.get_option("one_taken") - taken
        for i, nd in enumerate(needles):
            assert rows[i, :counts[i]].tolist() == chk.find(nd, limit), (n, limit, nd)
        rows_wanted = sum(1 for nd in needles if chk.find(nd, 1))
        kernels = m.last_kernels()
        if limit > 120:
            assert went == 0 and not any("pinned" in k or "find_one" in k for k in kernels)   # the batch's way
        elif rows_wanted <= few_max:
            assert went == rows_wanted and kernels == ["find_one_kernel<1024>"]   # one row per needle that has any posting
        else:
            assert went == rows_wanted and kernels == ["find_kernel<uint8_t,1024,true,true>", "merge_parts_pinned_kernel"]


@pytest.mark.parametrize("n", [27, 32, 47, 64, 128])
def test_a_servers_coalesced_finds_take_latency_mode_without_copies(geo, n):
    """More than few_max and up to mid_max needles (defaults: 25 .. 128 needles with a posting): find_kernel<..., RANGED> over needle arrays in the
    pinned page, merge_parts_pinned_kernel writing rows and sequence words back into it -- rows equal the live reference's
    for every needle, at two limits, twice in a row (the queue word returns to zero), empty needles and needles without a
    posting among them; with mid_max 0 the same batch goes the batch's way and says the same."""
    m, chk, strings = geo
    rng = np.random.default_rng(180 + n)
    assert m.get_option("mid_max") == 128
    for limit in (10, 100):
        for rep in range(2):
            picks = [strings[int(k)] for k in rng.integers(0, len(strings), size=n)]
            needles = [p[: max(1, len(p) - 1)] if k % 3 else p for k, p in enumerate(picks)]
            needles[2] = b""
            needles[5] = b"qqqqzzzzxxxx"
            needles[7] = b" ".join(picks[7:10])[:60]             # (over 15 trigrams: byte counters beyond nib_windows)
            packed, off = _pack(needles)
            taken = m.get_option("one_taken")
            rows, counts = m.find_batch_packed(packed, off, limit)
            assert m.get_option("one_taken") - taken == sum(1 for nd in needles if chk.find(nd, 1))
            assert m.last_kernels() == ["find_kernel<uint8_t,1024,true,true>", "merge_parts_pinned_kernel"]
            for i, nd in enumerate(needles):
                assert rows[i, :counts[i]].tolist() == chk.find(nd, limit), (n, limit, nd)
    m.set_option("mid_max", 0)
    try:
        taken = m.get_option("one_taken")
        rows2, counts2 = m.find_batch_packed(packed, off, 100)
        assert m.get_option("one_taken") == taken and "merge_parts_pinned_kernel" not in m.last_kernels()
        assert np.array_equal(counts2, counts) and all(np.array_equal(rows2[i, :counts[i]], rows[i, :counts[i]]) for i in range(n))
    finally:
        m.set_option("mid_max", 128)


def test_latency_mode_over_the_pinned_page_at_its_edges(geo):
    """The corners of find_few's latency-mode half: two needles cut into as many ranges as there are window pairs ("few_max" 1),
    sixteen tasks aimed at per workgroup, limits 1 and 120, exactly 25 and exactly 128 needles with postings, a batch of nothing
    but needles without a posting (no launch at all), a needle of more than 64 distinct trigrams in the batch (the whole batch
    goes the batch's way) and one of exactly the most the launch takes -- every row against the live reference."""
    m, chk, strings = geo
    rng = np.random.default_rng(91)
    pick = lambda k: [strings[int(i)] for i in rng.integers(0, len(strings), size=k)]

    def check(needles, limit, pinned):
        packed, off = _pack(needles)
        rows, counts = m.find_batch_packed(packed, off, limit)
        if pinned is not None:
            assert ("merge_parts_pinned_kernel" in m.last_kernels()) == pinned, (len(needles), limit, m.last_kernels())
        for i, nd in enumerate(needles):
            assert rows[i, :counts[i]].tolist() == chk.find(nd, limit), (len(needles), limit, nd)

    try:
        m.set_option("few_max", 1)
        for limit in (1, 10, 120):
            check(pick(2), limit, True)                        # two needles, a range per window pair
        m.set_option("latency_tasks", 16)
        check(pick(40), 10, True)
        m.set_option("latency_tasks", 0)
        m.set_option("few_max", 24)
        check(pick(25), 1, True)
        check(pick(25), 120, True)
        check(pick(128), 10, True)
        taken = m.get_option("one_taken")
        check([b""] * 30, 10, None)                            # nothing to search for: no launch
        assert m.get_option("one_taken") == taken
        long_one = b" ".join(pick(8))[:120]                    # (well over 64 distinct trigrams)
        assert len(set(Oracle.tokenise(long_one))) > 64
        check(pick(29) + [long_one], 10, False)
        # (a needle of more than 64 distinct trigrams where ranges are taken -- alone, or among a few dozen: the ranged launch
        # leaves it to the launches behind it.  Through round 6 that launch never came back: needle_major.inc, find_kernel)
        grow = b" ".join(pick(30))
        for want in (65, 100, 127, 128, 140):
            nd = next(grow[:k] for k in range(30, 400) if len(set(Oracle.tokenise(grow[:k]))) >= want)
            check([nd], 10, False)
            assert _find(m, nd, 10) == chk.find(nd, 10)
            check(pick(3) + [nd] + pick(2), 10, False)
        base = b" ".join(pick(8))
        full = max((base[:k] for k in range(40, 110) if len(set(Oracle.tokenise(base[:k]))) <= 64), key=len)
        assert len(set(Oracle.tokenise(full))) >= 56           # (as many distinct trigrams as the launch takes, or nearly)
        check(pick(29) + [full], 10, True)
        m.set_option("mid_max", 10)                            # (below few_max: the shared launch still takes up to few_max)
        check(pick(20), 10, False)
        assert m.last_kernels() == ["find_one_kernel<1024>"]
        check(pick(30), 10, False)                             # ... and beyond it the batch's way
    finally:
        for key, value in (("few_max", 24), ("mid_max", 128), ("latency_tasks", 0)):
            m.set_option(key, value)


@pytest.mark.parametrize("n", [17, 32, 64, 128])
def test_a_servers_coalesced_finds_share_one_launch(geo, n):
    """The same launch for up to 128 needles (option "few_max" raised to the kernel's limit): codes from the pinned page, a
    ticket per row, a handful of window pairs per workgroup ("mid_workgroups" picks how many) -- rows equal the live
    reference's for every needle, at two limits and two grid shapes, twice in a row (the tickets return to zero)."""
    m, chk, strings = geo
    rng = np.random.default_rng(80 + n)
    m.set_option("few_max", 128)
    try:
        for limit, wgs in [(10, 1024), (100, 256)]:
            m.set_option("mid_workgroups", wgs)
            for rep in range(2):
                picks = [strings[int(k)] for k in rng.integers(0, len(strings), size=n)]
                needles = [p[: max(1, len(p) - 1)] if k % 3 else p for k, p in enumerate(picks)]
                needles[2] = b""
                packed, off = _pack(needles)
                taken = m.get_option("one_taken")
                rows, counts = m.find_batch_packed(packed, off, limit)
                assert m.get_option("one_taken") - taken == sum(1 for nd in needles if chk.find(nd, 1))
                for i, nd in enumerate(needles):
                    assert rows[i, :counts[i]].tolist() == chk.find(nd, limit), (n, limit, wgs, nd)
    finally:
        m.set_option("few_max", 24)
        m.set_option("mid_workgroups", 1024)
