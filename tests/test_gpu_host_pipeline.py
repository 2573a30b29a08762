"""Large host-buffer batches go through blurrily_storage_find_batch in chunks -- needles to the device, search,
rows back, on three streams, two slots by turns (c_abi.hip: find_batch_chunked).  Every element is still one
blurrily_storage_find: rows must equal those of the batch taken in one piece, and the oracle's."""
import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle

pytestmark = pytest.mark.gpu


def _live(rows, counts):
    limit = rows.shape[1]
    mask = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    return np.where(mask[:, :, None], rows, 0)


@pytest.mark.parametrize("chunk,n_q,limit", [(1000, 5000, 10), (1024, 2049, 3), (700, 9001, 100), (512, 4096, 0)])
def test_chunked_batch_equals_one_piece_and_the_oracle(chunk, n_q, limit):
    hay, off = W.words(60000, 21)
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, n_q, 22)
    m.set_option("host_chunk", 0)
    rows1, counts1 = m.find_batch_packed(q, qo, limit)
    m.set_option("host_chunk", chunk)
    assert m.get_option("host_chunk") == chunk
    for _ in range(2):                                     # twice: the slots and staging are reused
        rows2, counts2 = m.find_batch_packed(q, qo, limit)
        assert np.array_equal(counts1, counts2)
        assert np.array_equal(_live(rows1, counts1), _live(rows2, counts2))
    if limit:
        want = o.batch(q, qo, limit=limit)
        assert np.array_equal(counts2, want["counts"])
        assert np.array_equal(_live(rows2, counts2), _live(want["rows"], want["counts"]))
    else:
        assert not counts2.any()


def test_chunked_raw_batch_and_flags():
    """The un-normalised entry point in chunks: normalisation on the device per chunk, flags per needle."""
    hay, off = W.words(30000, 23)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    needles = [b"  " + nd.upper() + b"!!" for nd in W.unpack(*W.queries(hay, off, 3000, 24))]
    needles[17] = "café".encode()                    # non-ASCII: flagged, not guessed
    needles[2999] = b""
    from blurrily_amd.map import _pack
    packed, offs = _pack(needles)
    m.set_option("host_chunk", 0)
    r1, c1, f1 = m.find_batch_raw_packed(packed, offs, 10)
    m.set_option("host_chunk", 600)
    r2, c2, f2 = m.find_batch_raw_packed(packed, offs, 10)
    assert np.array_equal(c1, c2) and np.array_equal(f1, f2) and f2[17] == 1 and f2.sum() == 1
    assert np.array_equal(_live(r1, c1), _live(r2, c2))


def test_mutations_between_chunked_batches():
    """Tombstones and pending puts (base + delta images merged per chunk) under the pipeline."""
    hay, off = W.words(40000, 25)
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, 4000, 26)
    m.set_option("host_chunk", 900)
    rows, counts = m.find_batch_packed(q, qo, 10)
    needles = W.unpack(q, qo)
    for i in range(0, 4000, 40):                           # delete best matches, add the needles themselves
        if counts[i]:
            ref = int(rows[i, 0, 0])
            assert m.delete(ref) == o.delete(ref)
        assert m.put(needles[i], 100000 + i, 0) == o.put(needles[i], 100000 + i, 0)
    rows, counts = m.find_batch_packed(q, qo, 10)
    want = o.batch(q, qo, limit=10)
    assert np.array_equal(counts, want["counts"])
    assert np.array_equal(_live(rows, counts), _live(want["rows"], want["counts"]))
    assert m.device_info()["base_builds"] == 1
