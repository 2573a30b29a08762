"""In-process multi-GPU behind the C ABI (option "devices", c_abi.hip: run_find_multi): the device image replicated
on the first n visible devices, a large batch sharded contiguously over them, every replica's block of rows sent
into the caller's buffers by peer copies.  The drop-in host is ONE process (lib/blurrily/server.rb:19-30), so this is
how it reaches more than one GPU.  With a single GPU on the box the replicas share it (replica k lives on device
(primary + 1 + k) mod visible): separate images, scratch and streams, the same code path -- the answer must be the
one-piece call's, and the oracle's, whatever the number of shards."""
import ctypes as C

import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap, _native
from helpers import Oracle

pytestmark = pytest.mark.gpu


def _live(rows, counts):
    limit = rows.shape[1]
    mask = np.arange(limit)[None, :] < counts[:, None].astype(np.int64)
    return np.where(mask[:, :, None], rows, 0)


@pytest.fixture(scope="module")
def pair():
    hay, off = W.geonames(300000, 60000, 71)
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    o.put_many(hay, off)
    yield m, o, hay, off
    m.close()


@pytest.mark.parametrize("devices,n_q,limit", [(2, 20000, 10), (8, 20011, 10), (8, 9000, 100), (3, 16384, 1)])
def test_sharded_batch_equals_one_piece_and_the_oracle(pair, devices, n_q, limit):
    m, o, hay, off = pair
    q, qo = W.queries(hay, off, n_q, 72 + devices)
    m.set_option("devices", 1)
    rows1, counts1 = m.find_batch_packed(q, qo, limit)
    m.set_option("devices", devices)
    assert m.get_option("devices") == devices
    for _ in range(2):                                     # twice: replicas, staging and events are reused
        rows2, counts2 = m.find_batch_packed(q, qo, limit)
        assert np.array_equal(counts1, counts2)
        assert np.array_equal(_live(rows1, counts1), _live(rows2, counts2))
    idx = np.arange(0, n_q, 7, dtype=np.uint32)
    want = o.batch(q, qo, idx=idx, limit=limit)
    assert np.array_equal(counts2[idx], want["counts"])
    assert np.array_equal(_live(rows2[idx], counts2[idx]), _live(want["rows"], want["counts"]))
    m.set_option("devices", 1)


def test_device_resident_batch_over_replicas_with_timing_and_nb_entries(pair):
    """blurrily_storage_find_batch_device with "devices" 4: device pointers on the primary in, rows / counts /
    nb_entries on the primary out, the caller's stream waits for the replicas."""
    import torch
    m, o, hay, off = pair
    n_q, limit = 30000, 10
    q, qo = W.queries(hay, off, n_q, 91)
    dev = torch.device("cuda", 0)
    d_packed = torch.from_numpy(q).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    lib = _native.lib()
    out = {}
    for devices in (1, 4):
        m.set_option("devices", devices)
        rows = torch.zeros((n_q, limit, 3), dtype=torch.int32, device=dev)
        counts = torch.zeros((n_q,), dtype=torch.int32, device=dev)
        nb = torch.zeros((n_q,), dtype=torch.int32, device=dev)
        m.set_timing(devices == 4)
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.blurrily_storage_find_batch_device(m.handle, d_packed.data_ptr(), int(qo[-1]), d_off.data_ptr(), n_q, limit,
                                                    rows.data_ptr(), counts.data_ptr(), nb.data_ptr(), stream)
        assert rc == 0, C.get_errno()
        torch.cuda.synchronize()
        out[devices] = (rows.cpu().numpy().view(np.uint32), counts.cpu().numpy().view(np.uint32), nb.cpu().numpy())
        if devices == 4:
            assert m.device_info()["last_find_kernel_ms"] > 0
        m.set_timing(False)
    assert np.array_equal(out[1][1], out[4][1]) and np.array_equal(out[1][2], out[4][2])
    assert np.array_equal(_live(out[1][0], out[1][1]), _live(out[4][0], out[4][1]))
    m.set_option("devices", 1)


def test_replicas_follow_puts_and_deletes():
    """Mutations after the replicas exist: tombstones and the delta image of pending puts reach every replica (device to
    device), a rebuild of the base image is cloned again."""
    hay, off = W.words(120000, 75)
    strings = W.unpack(hay, off)
    m, o = RawMap(), Oracle()
    m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
    o.put_many(hay, off)
    m.set_option("devices", 4)
    q, qo = W.queries(hay, off, 12000, 76)

    def check():
        rows, counts = m.find_batch_packed(q, qo, 10)
        want = o.batch(q, qo, limit=10)
        assert np.array_equal(counts, want["counts"])
        assert np.array_equal(_live(rows, counts), _live(want["rows"], want["counts"]))

    check()
    builds = m.device_info()["base_builds"]
    for ref in range(1, 4000, 3):                          # deletes of base references: tombstones
        assert m.delete(ref) == o.delete(ref)
    check()
    for k, s in enumerate(strings[:500]):                  # puts of new references: the delta image
        assert m.put(s + b"x", 500000 + k, 0) == o.put(s + b"x", 500000 + k, 0)
    check()
    for ref in range(500000, 500100):                      # deletes of pending references
        assert m.delete(ref) == o.delete(ref)
    check()
    assert m.device_info()["base_builds"] == builds        # (all of it served without a rebuild)
    extra_hay, extra_off = W.words(70000, 77)              # a bulk import: the base is rebuilt, the replicas cloned again
    refs = np.arange(2000000, 2000000 + len(extra_off) - 1, dtype=np.uint32)
    m.put_many_packed(extra_hay, extra_off, refs)
    o.put_many(extra_hay, extra_off, refs)
    check()
    assert m.device_info()["base_builds"] == builds + 1
    m.close()


def test_device_info_says_what_the_replicas_sit_on(pair):
    """`n_gpus: 2` with both replicas on one GPU.  The library reports the PHYSICAL devices holding a copy
    (PCI bus ids), which replicas share the primary's device and which reach it by peer access; "devices" back to 1
    drops the replicas at once."""
    import torch
    m, o, hay, off = pair
    visible = torch.cuda.device_count()
    q, qo = W.queries(hay, off, 8192, 5)
    m.set_option("devices", 4)
    m.find_batch_packed(q, qo, 10)
    info = m.device_info()
    assert info["n_replicas"] == 3 and info["distinct_devices"] == min(4, visible)
    assert info["pci_bus_id"] and len(info["pci_bus_id"]) >= 7
    # replica k lives on device (primary + 1 + k) mod visible: with one GPU all three share the primary's
    want_same = sum(1 << k for k in range(3) if (info["device_ordinal"] + 1 + k) % visible == info["device_ordinal"])
    assert info["same_device_mask"] == want_same
    assert info["peer_access_mask"] & info["same_device_mask"] == 0
    m.set_option("devices", 1)
    info = m.device_info()
    assert info["n_replicas"] == 0 and info["distinct_devices"] == 1
