"""The needle-major sweep that LEAVES dense slices out of a step's count (find_kernels.hip: sweep_role -- a manager
wave and fifteen workers; c_abi.hip: sweep 3) against the oracle, row for row.  Once a needle has a threshold, the
largest slices of at least "nm_dense" postings of a window (at most need - "nm_cmin" of them, four at most) are not
counted; a reference that could still reach the threshold on its best case waits in the step's pending list and is
settled through the left-out slices' bitmaps by the manager during the next step (storage.c:545-573 is what it must
still compute: every reference's match count, the best `limit` by matches, weight, reference).  By default the sweep is
taken where a measurement or the static rule ("nm_min_windows" 256) says so; here it is forced onto haystacks the oracle
answers in seconds: "ws_autotune" 0, "wsweep" 0, "nm_min_windows" 0, small "dense_min" / "nm_dense"."""
import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap
from blurrily_amd.map import _pack
from helpers import Oracle

pytestmark = pytest.mark.gpu
LEFT_OUT = 1 << 21            # kPathNmLeftOut


def _pair(hay, off, **opts):
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    for k, v in dict(ws_autotune=0, wsweep=0, nm_min_windows=0, small_sweep=0, **opts).items():
        m.set_option(k, v)
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    o.put_many(hay, off)
    return m, o


def _check(m, o, packed, off, limit, expect_left_out=True, upto=None):
    m.set_stats(True)
    rows, counts = m.find_batch_packed(packed, off, limit)
    st, flags = m.find_stats(), m.find_path_flags(len(off) - 1)
    m.set_stats(False)
    assert m.get_option("last_sweep") == 3, m.get_option("last_sweep")
    if expect_left_out:
        assert st["probes"] > 0 and (flags & LEFT_OUT).any(), st
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
    """Edited haystack strings plus the awkward ones: whole long strings, gibberish (rare trigrams: runs of empty
    steps, no threshold for a long time), empty, one letter."""
    q, qo = W.queries(hay, off, n_q, seed)
    needles = W.unpack(q, qo)
    rng = np.random.default_rng(seed)
    lens = (off[1:] - off[:-1]).astype(np.int64)
    for i in np.argsort(lens)[-300:]:
        needles.append(bytes(hay[int(off[i]):int(off[i + 1])]))
    for _ in range(300):
        needles.append(bytes(rng.choice(list(b"qxzjkvw"), size=int(rng.integers(1, 9))).tolist()))
    needles += [b"", b"a", b" ", b"zzzzzzzz", b"e", b"the"]
    order = rng.permutation(len(needles))
    return _pack([needles[i] for i in order])


@pytest.fixture(scope="module")
def medium():
    hay, off = W.geonames(700000, 90000, 51)                   # 11 windows
    m, o = _pair(hay, off, dense_min=256, nm_cmin=3, nm_dense=256)
    q, qo = _mixed(hay, off, 4000, 52)
    yield m, o, q, qo
    m.close()


@pytest.mark.parametrize("limit,cmin,dense", [(10, 3, 512), (10, 1, 256), (10, 2, 1024), (1, 3, 256), (3, 2, 512), (64, 3, 256),
                                              (33, 3, 512), (10, 5, 256), (100, 3, 256), (128, 2, 512), (149, 3, 256)])
def test_geonames_medium_all_rows_vs_oracle(limit, cmin, dense, medium):
    m, o, q, qo = medium                                       # (one image, one oracle: "nm_cmin" / "nm_dense" are read per find)
    m.set_option("nm_cmin", cmin)
    m.set_option("nm_dense", dense)
    flags = _check(m, o, q, qo, limit, upto=149)
    assert (flags & LEFT_OUT).sum() > 1000


def test_hot_trigram_haystack_and_massive_ties():
    """configs[4]'s kind of haystack: a few trigrams in half the strings, lengths that collide -- the left-out slices are
    the hot ones, the pending lists and the pool's tail fill up (overflows: the step swept again, every slice counted)."""
    hay, off = W.skewed(500000, 53)
    m, o = _pair(hay, off, dense_min=512, nm_cmin=2, nm_dense=512)
    q, qo = W.queries(hay, off, 3000, 54)
    _check(m, o, q, qo, 100, upto=100)                          # (configs[4]'s limit: the 1 024-entry pool, a tail of 256)
    _check(m, o, q, qo, 10, upto=100)
    _check(m, o, q, qo, 64, upto=100)
    m.close()


def test_single_words_small_windows_and_latency_mode_is_left_alone():
    hay, off = W.words(200000, 55)                              # 4 windows
    m, o = _pair(hay, off, dense_min=128, nm_cmin=2, nm_dense=128)
    q, qo = W.queries(hay, off, 20000, 56)
    _check(m, o, q, qo, 10)
    # a handful of needles: latency mode (ranges), which leaves nothing out; a limit whose pool has no room for the
    # settled candidates' tail (from 150 on) neither; limit 100 does (the 1 024-entry pool, a tail of 256)
    q2, qo2 = W.queries(hay, off, 40, 57)
    rows, counts = m.find_batch_packed(q2, qo2, 10)
    assert m.last_kernels() == ["find_kernel<uint8_t,1024,true,true>", "merge_parts_pinned_kernel"]   # (over the pinned page: c_abi.hip, find_few)
    want = o.batch(q2, qo2, limit=10)
    assert np.array_equal(counts, want["counts"])
    m.set_option("mid_max", 0)                                  # ... and the batch's way, copies and all
    rows, counts = m.find_batch_packed(q2, qo2, 10)
    assert m.get_option("last_sweep") == 0 and m.last_kernels() == ["find_kernel<uint8_t,1024,true,true>"]
    assert np.array_equal(counts, want["counts"])
    m.set_option("mid_max", 128)
    _check(m, o, q[:int(qo[4000])], qo[:4001], 100, expect_left_out=False)   # (the hundredth-best match of a word is a poor one: little to leave out)
    assert m.get_option("last_sweep") == 3
    rows, counts = m.find_batch_packed(q, qo, 200)
    assert m.get_option("last_sweep") == 1
    m.close()


def test_tombstones_and_pending_puts_under_the_leaving_sweep():
    """Deletes since the image was built (tombstone bits: a pending candidate is looked up where it is harvested) and
    puts since then (the delta image, merged in) -- storage.c:584-612, :398-473 semantics under this sweep."""
    hay, off = W.geonames(400000, 60000, 58)
    strings = W.unpack(hay, off)
    m, o = _pair(hay, off, dense_min=256, nm_cmin=2, nm_dense=256)
    q, qo = W.queries(hay, off, 5000, 59)
    _check(m, o, q, qo, 10)
    for ref in range(1, 30000, 23):
        assert m.delete(ref) == o.delete(ref)
    for k, s in enumerate(strings[:300]):
        assert m.put(s + b" x", 900000 + k, 0) == o.put(s + b" x", 900000 + k, 0)
    _check(m, o, q, qo, 10)
    assert m.device_info()["base_builds"] == 1
    m.close()


def test_tune_measures_a_class_ahead_of_time():
    """blurrily_storage_tune: the measurement the first batch of a class would otherwise make inside its find call --
    so that blurrily_storage_find_batch_device on that class never waits for one (include/blurrily_storage.h)."""
    hay, off = W.geonames(600000, 80000, 61)
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    o.put_many(hay, off)
    m.sync_device()
    assert m.get_option("tuned_class") == -1 and m.get_option("ws_choice") == 0
    q, qo = W.queries(hay, off, 3000, 62)
    m.tune(q, qo, 70000, 10)                                    # 3 000 needles given, 70 000 run: class 1
    assert m.get_option("tuned_class") == 1
    picked = (m.get_option("ws_choice") >> 2) & 3
    assert picked in (1, 2, 3) and m.get_option("tuned_nm_us") > 0 and m.get_option("tuned_leave_us") > 0
    q2, qo2 = W.queries(hay, off, 70000, 63)
    rows, counts = m.find_batch_packed(q2, qo2, 10)
    assert m.get_option("last_sweep") == picked and (m.get_option("ws_choice") >> 2) & 3 == picked
    idx = np.arange(0, 70000, 40, dtype=np.uint32)
    want = o.batch(q2, qo2, idx=idx, limit=10)
    live = np.arange(10)[None, :] < want["counts"][:, None].astype(np.int64)
    assert np.array_equal(counts[idx], want["counts"])
    assert np.array_equal(np.where(live[:, :, None], rows[idx], 0), np.where(live[:, :, None], want["rows"], 0))
    m.close()


def test_device_info_for_a_caller_of_another_header_version():
    """blurrily_storage_device_info_sized writes no more than the caller's structure holds and says how large the
    library's is (the structure only grows at its end)."""
    import ctypes as C
    from blurrily_amd import _native
    m = RawMap()
    m.put(b"london", 1, 0)
    m.find(b"london", 1)
    lib = _native.lib()
    buf = (C.c_ubyte * 96)(*([0xEE] * 96))
    full = lib.blurrily_storage_device_info_sized(m.handle, buf, 16)
    assert full == C.sizeof(_native.DeviceInfo)
    assert bytes(buf[16:]) == b"\xEE" * 80                      # nothing behind the 16 bytes the caller declared
    assert int.from_bytes(bytes(buf[4:8]), "little") == 1       # n_refs
    m.close()


def test_a_bad_first_sample_of_the_measured_choice_is_corrected(geonames_full):
    """The sweep that serves a class of batches is MEASURED on the class's first batch (c_abi.hip: run_find_on) -- every
    sweep twice, the better run counting -- and WATCHED afterwards: two batches of the class in a row that run over 10 %
    slower per needle than the measurement saw have the class measured again.  Here the first measurement is given a bad sample on
    purpose (option "tune_inject": the plain sweep at half its time, which makes it win); the second batch then runs
    the plain sweep at its real speed, so does the third; the fourth finds that out and measures again."""
    n = 8423769
    hay, off = geonames_full.hay, geonames_full.off
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    q, qo = W.queries(hay, off, 70000, 31)
    cls_shift = 2 * 1                                        # class 1: limit <= 32, 65 536 .. 262 143 needles
    rows0, counts0 = m.find_batch_packed(q, qo, 10)          # a clean measurement first
    clean = (m.get_option("ws_choice") >> cls_shift) & 3
    if clean == 1:
        pytest.skip("the plain sweep wins this class on this box: nothing a halved plain sample could get wrong")
    m.set_option("ws_choice", 0)                             # forget it; the next measurement gets the bad sample
    m.set_option("tune_inject", 1)
    m.find_batch_packed(q, qo, 10)
    assert (m.get_option("ws_choice") >> cls_shift) & 3 == 1 and m.get_option("retunes") == 0
    rows, counts = m.find_batch_packed(q, qo, 10)            # batch 2: the plain sweep, watched
    assert m.get_option("last_sweep") == 1
    rows, counts = m.find_batch_packed(q, qo, 10)            # batch 3: batch 2 was slow against the sample -- once is noise
    assert m.get_option("last_sweep") == 1 and m.get_option("retunes") == 0
    rows, counts = m.find_batch_packed(q, qo, 10)            # batch 4: so was batch 3 -> measured again
    assert m.get_option("retunes") == 1
    assert (m.get_option("ws_choice") >> cls_shift) & 3 == clean
    rows, counts = m.find_batch_packed(q, qo, 10)
    assert m.get_option("last_sweep") == clean and m.get_option("retunes") == 1
    assert np.array_equal(counts, counts0) and np.array_equal(rows, rows0)
    m.close()


def test_mid_size_batches_have_their_sweep_measured_too():
    """Batches of 129 .. 16 383 needles -- a server's coalesced FINDs -- are a class of their own (6; 7 above limit 32)
    whose first batch measures the sweeps like a large one's (c_abi.hip: run_find_on); smaller batches keep the static
    rule.  Same rows whatever is chosen."""
    hay, off = W.geonames(700000, 90000, 51)                   # 11 windows: the static rule alone would never leave a slice out
    n = len(off) - 1
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, 3000, 71)
    assert m.get_option("ws_choice") == 0
    rows, counts = m.find_batch_packed(q, qo, 10)
    picked = (m.get_option("ws_choice") >> 12) & 3
    # (the measuring batch ends with the plain sweep: the rows in place are its -- "last_sweep" 1 whatever was picked)
    assert picked in (1, 3) and m.get_option("tuned_class") == 6 and m.get_option("last_sweep") == 1
    assert m.get_option("tuned_nm_us") > 0 and m.get_option("tuned_leave_us") > 0
    want = o.batch(q, qo, limit=10)
    live = np.arange(10)[None, :] < want["counts"][:, None].astype(np.int64)
    assert np.array_equal(counts, want["counts"])
    assert np.array_equal(np.where(live[:, :, None], rows, 0), np.where(live[:, :, None], want["rows"], 0))
    rows2, counts2 = m.find_batch_packed(q, qo, 10)            # the choice serves the class
    assert m.get_option("last_sweep") == picked and np.array_equal(rows2, rows) and np.array_equal(counts2, counts)
    q3, qo3 = W.queries(hay, off, 100, 72)                     # a hundred needles (latency mode): nothing is measured
    before = m.get_option("ws_choice")
    m.find_batch_packed(q3, qo3, 10)
    assert m.get_option("ws_choice") == before
    m.close()
