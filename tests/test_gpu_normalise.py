"""GPU parity of the device-side needle normaliser against
oracle/normalize_oracle.c -- Blurrily::Map#normalize_string restated in C from the Ruby text
(lib/blurrily/map.rb:40-47), independent of both the kernel and the product's Python mirror
(which tests/test_normalize.py checks against the same oracle and the spec's vectors)."""
import ctypes as C

import numpy as np
import pytest

import workloads as W
from blurrily_amd import Map, RawMap, _native, normalize_string
from blurrily_amd.map import _pack
from helpers import Oracle
from test_normalize import EDGE_NEEDLES, random_ascii_needles

pytestmark = pytest.mark.gpu

def _seen_by_c(b):
    """What the C side reads of a needle: the bytes up to the first NUL."""
    return b.split(b"\0", 1)[0]


class _Hip:
    """Device buffers through the HIP runtime the library is bound to (no torch needed here)."""

    def __init__(self):
        self.rt = _native.hip_runtime()
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def to_device(self, arr):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(arr.nbytes, 16)) == 0
        assert self.rt.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def to_host(self, p, like):
        out = np.empty_like(like)
        assert self.rt.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0      # synchronises the null stream
        return out

    def free(self, *ps):
        for p in ps:
            self.rt.hipFree(p)


def _device_normalise(needles):
    lib = _native.lib()
    hip = _Hip()
    packed, off = _pack(needles)
    buf = np.frombuffer(packed, dtype=np.uint8) if len(packed) else np.zeros(0, dtype=np.uint8)
    h_in = np.concatenate([buf, np.zeros(16, dtype=np.uint8)])
    h_flags = np.zeros(len(needles), dtype=np.uint32)
    d_in, d_off = hip.to_device(h_in), hip.to_device(off)
    d_out, d_flags = hip.to_device(np.full_like(h_in, 0x55)), hip.to_device(h_flags)
    assert lib.blurrily_normalize_batch_device(d_in, d_off, len(needles), d_out, d_flags, None) == 0, C.get_errno()
    out, flags = hip.to_host(d_out, h_in).tobytes(), hip.to_host(d_flags, h_flags)
    # in place as well, flags not asked for
    assert lib.blurrily_normalize_batch_device(d_in, d_off, len(needles), d_in, None, None) == 0
    inplace = hip.to_host(d_in, h_in).tobytes()
    hip.free(d_in, d_off, d_out, d_flags)
    got, got_inplace = [], []
    for i in range(len(needles)):
        a, b = int(off[i]), int(off[i + 1])
        got.append(_seen_by_c(out[a:b]))
        got_inplace.append(_seen_by_c(inplace[a:b]))
    return got, got_inplace, flags


def test_spec_vectors_and_edges():
    needles = list(EDGE_NEEDLES)
    got, got_inplace, flags = _device_normalise(needles)
    for nd, g, gi in zip(needles, got, got_inplace):
        want = _seen_by_c(Oracle.normalize_ascii(nd))
        assert g == want, (nd, g, want)
        assert gi == want, (nd, gi, want)
    assert not flags.any()
    # the reference-held vectors (every spec string that reaches normalize_string, tests/golden/spec_vectors.json): the
    # ASCII ones through the device normaliser, held to the form the spec's own assertion implies
    from helpers import load_golden
    vecs = [v for v in load_golden("spec_vectors.json")["normalize"] if all(ord(c) < 0x80 for c in v["raw"])]
    assert len(vecs) >= 12
    got, got_inplace, flags = _device_normalise([v["raw"].encode() for v in vecs])
    for v, g, gi in zip(vecs, got, got_inplace):
        assert g == gi == v["normalized"].encode(), (v, g, gi)
    assert not flags.any()


def test_random_ascii_needles_match_the_oracle_normaliser():
    needles = random_ascii_needles(77, 20000)
    got, got_inplace, flags = _device_normalise(needles)
    want = [_seen_by_c(Oracle.normalize_ascii(nd)) for nd in needles]
    assert got == want
    assert got_inplace == want
    assert not flags.any()


def test_non_ascii_needles_are_flagged_not_guessed():
    needles = ["café".encode(), b"plain", "@€%é".encode(), b"\x80", b"ok too"]
    _, _, flags = _device_normalise(needles)
    assert flags.tolist() == [1, 0, 1, 1, 0]


def test_find_batch_raw_equals_find_over_host_normalised_needles():
    rng = np.random.default_rng(5)
    hay, off = W.geonames(60000, 8000, seed=17)
    strings = W.unpack(hay, off)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, len(strings) + 1, dtype=np.uint32))
    raw = []
    for s in strings[:3000]:
        t = bytearray(s.upper() if rng.random() < 0.3 else s)
        for _ in range(int(rng.integers(0, 3))):
            t.insert(int(rng.integers(0, len(t) + 1)), int(rng.choice(list(b" -_.,'\t"))))
        raw.append(bytes(t))
    raw += [b"", b"  ", b"!!!", b"Two\nLines"]
    packed, offsets = _pack(raw)
    rows_raw, counts_raw, flags = m.find_batch_raw_packed(packed, offsets, 10)
    host = [normalize_string(r.decode("latin1")).encode("latin1") for r in raw]
    hp, ho = _pack(host)
    rows, counts = m.find_batch_packed(hp, ho, 10)
    assert not flags.any()
    assert np.array_equal(counts_raw, counts)
    for i in range(len(raw)):
        assert np.array_equal(rows_raw[i, :counts[i]], rows[i, :counts[i]]), raw[i]


def test_map_find_batch_mixes_device_and_host_normalisation():
    m = Map()
    for i, s in enumerate(["london", "paris", "sao paulo", "cafe de flore", "new york"], start=1):
        m.put(s, i)
    needles = ["LONDON", "São  Paulo", "café de Flore", "New-York", "@€%é"]
    assert m.find_batch(needles, 3) == [m.find(s, 3) for s in needles]


def test_a_handful_of_raw_needles_is_normalised_on_the_host_and_shares_one_launch():
    """blurrily_storage_find_batch_raw with up to "few_max" needles: normalised by the library on the host -- the C++ twin of
    normalise_kernel -- and served by find_one_kernel's launch (c_abi.hip: normalise_one, find_few): same rows and flags as
    the device's normalisation gives the same needles in a larger batch, and as finds over host-normalised needles."""
    rng = np.random.default_rng(77)
    hay, off = W.geonames(60000, 8000, seed=5)
    strings = W.unpack(hay, off)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, len(strings) + 1, dtype=np.uint32))
    alphabet = list(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOP0123456789 \t\n\r\x0b\x0c.,;-'!?\x00\x7f")
    raw = []
    for k in range(400):
        t = bytearray(strings[int(rng.integers(0, len(strings)))])
        for _ in range(int(rng.integers(0, 5))):
            pos = int(rng.integers(0, len(t) + 1))
            t[pos:pos] = bytes([alphabet[int(rng.integers(0, len(alphabet)))]])
        if k % 7 == 0:
            t = bytearray(bytes(t).upper())
        raw.append(bytes(t))
    raw += [b"", b"  ", b"!!!", b"Two\nLines", b"a\x00b", b"\x00", b"x  y", b"caf\xc3\xa9"]
    big_p, big_o = _pack(raw)
    want_rows, want_counts, want_flags = m.find_batch_raw_packed(big_p, big_o, 10)       # (408 needles: the device normalises)
    assert m.get_option("few_max") == 24
    taken = m.get_option("one_taken")
    for lo in range(0, len(raw), 17):                                                  # batches of 17 (and a shorter last one)
        part = raw[lo:lo + 17]
        p_, o_ = _pack(part)
        rows, counts, flags = m.find_batch_raw_packed(p_, o_, 10)
        assert np.array_equal(counts, want_counts[lo:lo + len(part)]) and np.array_equal(flags, want_flags[lo:lo + len(part)])
        for i in range(len(part)):
            assert np.array_equal(rows[i, :counts[i]], want_rows[lo + i, :counts[i]]), part[i]
    assert m.get_option("one_taken") > taken                                           # ... through the shared launch
    for lo in range(0, len(raw), 61):                                                  # batches of 61 (beyond few_max: the pinned page still, one window here: find_one_kernel's rows)
        part = raw[lo:lo + 61]
        p_, o_ = _pack(part)
        rows, counts, flags = m.find_batch_raw_packed(p_, o_, 10)
        assert np.array_equal(counts, want_counts[lo:lo + len(part)]) and np.array_equal(flags, want_flags[lo:lo + len(part)])
        for i in range(len(part)):
            assert np.array_equal(rows[i, :counts[i]], want_rows[lo + i, :counts[i]]), part[i]
    # and the single raw needle
    for i in (0, 5, len(raw) - 5, len(raw) - 1):
        p_, o_ = _pack([raw[i]])
        rows, counts, flags = m.find_batch_raw_packed(p_, o_, 10)
        assert counts[0] == want_counts[i] and np.array_equal(rows[0, :counts[0]], want_rows[i, :counts[0]]) and flags[0] == want_flags[i]
    m.close()
