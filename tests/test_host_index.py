"""Host side of the map through the C ABI (no GPU needed): put / delete / stats / save / load /
close, written after spec/blurrily/map_spec.rb, plus byte-level checks of the file format
against the reference's own loader/saver where oracle/_ref is present."""
import errno
import hashlib
import os

import numpy as np
import pytest

import workloads as W
from blurrily_amd import ClosedError, Map, RawMap
from helpers import Oracle, Reference

PAGE = 4096
HEADER = 32 + 25 * 21952            # packed trigram_map_t (storage.c:62-74): 548 832 bytes


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


# ---- '#stats' / '#put' (map_spec.rb:15-76) ------------------------------------------------

def test_stats_keys():
    st = Map().stats()
    assert isinstance(st["references"], int) and isinstance(st["trigrams"], int)


def test_put_stores_references():
    m = Map()
    m.put("foobar", 123, 0)
    assert m.stats() == {"references": 1, "trigrams": 7}


def test_put_returns_number_of_added_trigrams():
    m = Map()
    assert m.put("foobar", 123) == 7
    assert m.put("foobar", 123) == 0


def test_put_does_not_store_duplicate_references():
    m = Map()
    for _ in range(2):
        m.put("foobar", 123, 0)
    assert m.stats() == {"references": 1, "trigrams": 7}


def test_put_accepts_empty_strings():
    m = Map()
    m.put("", 123, 0)
    assert m.stats() == {"references": 1, "trigrams": 1}


def test_put_accepts_non_letter_characters():
    m = Map()
    m.put("@€%é", 123, 0)
    assert m.stats() == {"references": 1, "trigrams": 2}


def test_put_ignores_dupes_after_save_load_cycle(tmp_path):
    m = Map()
    m.put("london", 123)
    p = tmp_path / "map.test"
    m.save(p)
    m2 = Map.load(p)
    assert m2.put("paris", 123) == 0
    assert m2.stats() == {"references": 1, "trigrams": 7}


def test_put_makes_map_dirty(tmp_path):
    m = Map()
    p = tmp_path / "map.test"
    m.save(p)
    p.unlink()
    m.put("london", 123)
    m.save(p)
    assert p.exists()


def test_put_rejects_out_of_range_integers():
    m = Map()
    with pytest.raises(OverflowError):
        m.put("x", -1, 0)
    with pytest.raises(OverflowError):
        m.put("x", 1, 1 << 32)


# ---- '#delete' (map_spec.rb:78-116) ------------------------------------------------------

def test_delete_removes_references():
    m = Map()
    m.put("london", 123, 0)
    assert m.delete(123) == 7
    assert m.stats() == {"references": 0, "trigrams": 0}


def test_delete_makes_map_dirty(tmp_path):
    m = Map()
    p = tmp_path / "map.test"
    m.put("london", 123, 0)
    m.save(p)
    p.unlink()
    m.delete(123)
    m.save(p)
    assert p.exists()


def test_delete_ignores_missing_references():
    m = Map()
    assert m.delete(123) == 0
    assert m.stats()["trigrams"] == 0


def test_delete_permits_re_adds():
    m = Map()
    m.put("london", 1337)
    m.delete(1337)
    assert m.put("paris", 1337) == 6


# ---- '#save' (map_spec.rb:213-278) -------------------------------------------------------

@pytest.fixture
def three(tmp_path):
    m = Map()
    m.put("london", 10, 0)
    m.put("paris", 11, 0)
    m.put("monaco", 12, 0)
    return m, tmp_path / "map.test"


def test_save_creates_file(three):
    m, p = three
    m.save(p)
    assert p.exists()


def test_save_raises_when_directory_does_not_exist(three):
    m, _ = three
    with pytest.raises(FileNotFoundError):
        m.save("/var/nonexistent/foo")


def test_save_uses_magic_header(three):
    m, p = three
    m.save(p)
    head = p.read_bytes()[:8]
    assert head[:6] == b"trigra"
    assert head[6:7] == (b"\x01" if np.little_endian else b"\x02")
    assert head[7:8] == b"\x08"


def test_save_is_idempotent(three):
    m, p = three
    hashes = []
    for _ in range(3):
        RawMap.save(m, p)                      # bypass the clean-path elision, like `perform` on a dirty map
        hashes.append(md5(p))
    assert hashes[0] == hashes[1] == hashes[2]


def test_save_makes_map_clean(three):
    m, p = three
    m.save(p)
    p.unlink()
    m.save(p)
    assert not p.exists()


def test_file_layout(three):
    """storage.c:299-377: packed header padded to 134 pages with 0xFF, then one page-aligned
    block per non-empty bucket holding all 512 slots (unused slots 0xAA)."""
    m, p = three
    m.save(p)
    raw = p.read_bytes()
    n_buckets = len({c for s in (b"london", b"paris", b"monaco") for c in Oracle.tokenise(s)})
    assert len(raw) == 134 * PAGE + n_buckets * PAGE
    assert raw[HEADER:134 * PAGE] == b"\xff" * (134 * PAGE - HEADER)
    assert int.from_bytes(raw[8:12], "little") == 3 and int.from_bytes(raw[12:16], "little") == 20
    assert raw[16:32] == b"\0" * 16                           # mapped_size, refs
    code = Oracle.tokenise(b"london")[0]
    d = raw[32 + 25 * code: 32 + 25 * (code + 1)]
    slots, used = int.from_bytes(d[0:4], "little"), int.from_bytes(d[4:8], "little")
    offset = int.from_bytes(d[16:24], "little")
    assert (slots, used, d[8:16], d[24]) == (512, 1, b"\0" * 8, 0)
    block = raw[offset:offset + PAGE]
    assert block[:8] == (10).to_bytes(4, "little") + (6).to_bytes(4, "little")
    assert block[8:] == b"\xaa" * (PAGE - 8)


# ---- '.load' (map_spec.rb:281-330) -------------------------------------------------------

@pytest.fixture
def saved(tmp_path):
    m = Map()
    m.put("london", 10, 0)
    m.put("paris", 11, 0)
    m.put("monaco", 12, 0)
    p = tmp_path / "map.test"
    m.save(p)
    return p


def test_load_then_saves_identical_file(saved, tmp_path):
    alt = tmp_path / "map2.test"
    Map.load(saved).save(alt)
    assert md5(saved) == md5(alt)


def test_load_missing_file(saved):
    saved.unlink()
    with pytest.raises(FileNotFoundError):
        Map.load(saved)


def test_load_incorrect_file(saved):
    saved.write_bytes(b"foo")
    with pytest.raises(OSError) as e:
        Map.load(saved)
    assert e.value.errno == errno.EPROTO


def test_load_corrupt_file(saved):
    os.truncate(saved, 128)
    with pytest.raises(OSError) as e:
        Map.load(saved)
    assert e.value.errno == errno.EPROTO


def test_load_bad_magic_and_out_of_bounds(saved):
    raw = bytearray(saved.read_bytes())
    bad = bytearray(raw)
    bad[0:6] = b"trigrb"
    saved.write_bytes(bad)
    with pytest.raises(OSError) as e:
        Map.load(saved)
    assert e.value.errno == errno.EPROTO
    bad = bytearray(raw)
    code = Oracle.tokenise(b"london")[0]
    bad[32 + 25 * code + 16: 32 + 25 * code + 24] = (len(raw) + 4096).to_bytes(8, "little")
    saved.write_bytes(bad)
    with pytest.raises(OSError) as e:
        Map.load(saved)
    assert e.value.errno == errno.EPROTO


def test_load_clean_map(saved):
    m = Map.load(saved)
    saved.unlink()
    m.save(saved)
    assert not saved.exists()


# ---- '#close' (map_spec.rb:332-353) ------------------------------------------------------

def test_close():
    m = Map()
    m.close()
    for call in (m.close, lambda: m.put("london", 123), lambda: m.find("london"), lambda: m.save("foo"),
                 lambda: m.delete(1), m.stats):
        with pytest.raises(ClosedError):
            call()
    assert issubclass(ClosedError, RuntimeError) and Map.ClosedError is ClosedError


# ---- stress (map_spec.rb:355-438) --------------------------------------------------------

def test_stress_puts_and_deletes_force_reallocations(tmp_path):
    count = 1024
    m = Map()
    for i in range(count):
        m.put("Port-au-Prince", i)
    assert m.stats()["references"] == count
    p = tmp_path / "s.trigrams"
    m.save(p)
    # 1024 entries per bucket: capacity 512 -> 682 -> 909 -> 1212 (x4/3, storage.c:431)
    raw = p.read_bytes()
    code = Oracle.tokenise(b"port au prince")[0]
    d = raw[32 + 25 * code: 32 + 25 * (code + 1)]
    assert int.from_bytes(d[0:4], "little") == 1212 and int.from_bytes(d[4:8], "little") == 1024
    m2 = Map.load(p)
    for i in range(count):
        assert m2.delete(i) == m.delete(i) > 0
    assert m.stats() == m2.stats() == {"references": 0, "trigrams": 0}


def test_stress_put_save_load(tmp_path):
    p = tmp_path / "s.trigrams"
    m = Map()
    for i in range(100):
        m.put("Port-au-Prince", i)
        m.save(p)
        m = Map.load(p)
        assert m.stats()["references"] == i + 1


def test_put_many_equals_puts():
    hay, off = W.words(2000, 5)
    a, b = RawMap(), RawMap()
    refs = np.arange(10, 2010, dtype=np.uint32)
    total = a.put_many_packed(hay, off, refs)
    assert total == sum(b.put(s, int(r), 0) for s, r in zip(W.unpack(hay, off), refs))
    assert a.stats() == b.stats()


# ---- byte compatibility with the reference's own load / save -------------------------------

needs_ref = pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (no reference tree)")


@needs_ref
def test_reference_loads_and_resaves_our_file_byte_identically(tmp_path):
    hay, off = W.geonames(20000, 2000, 3)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, 20001, dtype=np.uint32))
    for r in range(1, 20001, 7):                   # deletions leave 0xFF-scribbled tail slots
        m.delete(r)
    m.put("after the deletions", 900001, 0)
    ours = tmp_path / "ours.trigrams"
    m.save(ours)
    ref = Reference(ours)
    assert ref.stats() == m.stats()
    theirs = tmp_path / "theirs.trigrams"
    assert ref.save(theirs) == 0
    ref.close()
    assert md5(ours) == md5(theirs)
    again = tmp_path / "again.trigrams"
    RawMap.load(theirs).save(again)                # and we re-save the reference's file identically
    assert md5(again) == md5(ours)


def test_bulk_put_many_is_byte_identical_to_single_puts(tmp_path):
    """Batches of 65 536 strings or more take the parallel host path (HostIndex::put_many): the saved
    file must equal, byte for byte, what the same puts one at a time leave -- duplicate references
    inside the batch and against the map, explicit and defaulted weights, buckets that already hold
    dead slots from deletes, several growth steps."""
    rng = np.random.default_rng(17)
    hay, off = W.geonames(90000, 9000, 33)
    strings = W.unpack(hay, off)
    n = len(strings)
    refs = np.arange(1000, 1000 + n, dtype=np.uint32)
    refs[rng.integers(0, n, size=500)] = refs[rng.integers(0, n, size=500)]      # duplicates inside the batch
    refs[:50] = np.arange(1, 51, dtype=np.uint32)                                 # and against the map below
    weights = rng.integers(0, 30, size=n).astype(np.uint32)
    weights[rng.random(n) < 0.5] = 0
    a, b = RawMap(), RawMap()
    for m in (a, b):                                                              # a map with history
        for r in range(1, 801):
            m.put(strings[(r * 7) % n], r, r % 5)
        for r in range(100, 700, 3):
            m.delete(r)
    total_a = a.put_many_packed(hay, off, refs, weights)
    total_b = sum(b.put(s, int(r), int(w)) for s, r, w in zip(strings, refs.tolist(), weights.tolist()))
    assert total_a == total_b
    assert a.stats() == b.stats()
    a.save(tmp_path / "a.trigrams")
    b.save(tmp_path / "b.trigrams")
    assert (tmp_path / "a.trigrams").read_bytes() == (tmp_path / "b.trigrams").read_bytes()
    # and a second bulk on top of the first (buckets now well past their first allocation)
    hay2, off2 = W.geonames(70000, 9000, 34)
    refs2 = np.arange(500000, 570000, dtype=np.uint32)
    assert a.put_many_packed(hay2, off2, refs2) == sum(b.put(s, int(r), 0) for s, r in zip(W.unpack(hay2, off2), refs2))
    a.save(tmp_path / "a2.trigrams")
    b.save(tmp_path / "b2.trigrams")
    assert (tmp_path / "a2.trigrams").read_bytes() == (tmp_path / "b2.trigrams").read_bytes()
