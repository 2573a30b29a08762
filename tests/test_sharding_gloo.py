"""The N>1 path on CPU: world_size 2 over gloo.  The shard/gather code of
blurrily_amd/sharding.py is exercised with the oracle standing in for the per-rank GPU find
(the HIP path itself is covered by the -m gpu tests)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blurrily_amd.sharding import ResultBlock, block_bytes, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_configs3_batch_of_eight_million_over_eight_ranks():
    """BASELINE.json configs[3]: an 8 M-needle batch sharded over the 8 GPUs of a node (bench.py --scaling strong):
    contiguous shards of one million each, and what ONE gather then moves (~124 MB per peer)."""
    spans = [shard_bounds(8_000_000, 8, r) for r in range(8)]
    assert spans == [(r * 1_000_000, (r + 1) * 1_000_000) for r in range(8)]
    assert block_bytes(1_000_000, 10) == 124_000_000
    # a batch that does not divide: the first ranks take one more, nothing is lost or doubled
    spans = [shard_bounds(8_000_003, 8, r) for r in range(8)]
    assert [b - a for a, b in spans] == [1_000_001] * 3 + [1_000_000] * 5 and spans[-1][1] == 8_000_003
    for n in (2, 4, 8):
        assert sum(b - a for a, b in (shard_bounds(8_000_000, n, r) for r in range(n))) == 8_000_000


def test_result_block_is_one_buffer_of_fixed_stride():
    """limit x 12 B + 4 B per needle, rows and counts views of one allocation."""
    b = ResultBlock(7, 10)
    assert b.buf.numel() * 4 == block_bytes(7, 10) == 7 * (10 * 12 + 4)
    b.rows[6, 9, 2] = 77
    b.counts[6] = 5
    assert int(b.buf[7 * 30 - 1]) == 77 and int(b.buf[-1]) == 5
    assert b.rows.data_ptr() == b.buf.data_ptr() and b.counts.data_ptr() == b.buf.data_ptr() + 7 * 120


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_needles, limit, out_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tools"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import workloads as W
    from helpers import Oracle
    from blurrily_amd.sharding import find_batch_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hay, off = W.words(3000, 5)                       # index replicated on every rank
    o = Oracle()
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, n_needles, 6)
    needles = W.unpack(q, qo)

    calls = []
    real_gather = dist.gather

    def counting_gather(*a, **k):                      # the path has ONE exchange step
        calls.append(1)
        return real_gather(*a, **k)
    dist.gather = counting_gather

    def find_fn(batch, lim, rows, counts):             # stand-in for the per-rank GPU batch: fills the block in place
        assert rows.shape == (len(batch), lim, 3) and counts.shape == (len(batch),)
        for i, nd in enumerate(batch):
            r = o.find(nd, lim)
            counts[i] = len(r)
            if r:
                rows[i, :len(r)] = torch.tensor(r, dtype=torch.int64).to(torch.int32)

    got = find_batch_sharded(dist, find_fn, needles, limit, rank, world)
    dist.gather = real_gather
    assert len(calls) == 1, calls
    if rank == 0:
        rows, counts = got
        ok = rows.shape[0] == n_needles
        for i, nd in enumerate(needles):
            want = o.find(nd, limit)
            ok = ok and rows[i, :int(counts[i])].tolist() == want
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_needles", [64, 37])       # even split, and a ragged last shard
def test_world_size_2_gather_reassembles_in_order(tmp_path, n_needles):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), n_needles, 5, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _pipelined_worker(rank, world, port, steps, out_path):
    """bench.py's N > 1 loop in miniature: two blocks by turns, the gather of one step in flight while the next
    step fills the other block, a block reused only behind its gather's wait()."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from blurrily_amd.sharding import gather_blocks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, limit = 50, 4
    blocks = [ResultBlock(n, limit), ResultBlock(n, limit)]
    gathered = [torch.empty((world, blocks[0].buf.numel()), dtype=torch.int32) if rank == 0 else None for _ in range(2)]
    in_flight = [None, None]
    seen = []
    for step in range(steps):
        i = step % 2
        if in_flight[i] is not None:
            in_flight[i].wait()
            if rank == 0:
                seen.append((step - 2, gathered[i].clone()))
        blocks[i].rows[:] = 1000 * step + rank                 # "the search": every step writes different rows
        blocks[i].counts[:] = step
        in_flight[i] = gather_blocks(dist, blocks[i], gathered[i], rank, async_op=True)
    for k in range(2):                                          # the fence of the timed region
        i = (steps + k) % 2
        if in_flight[i] is not None:
            in_flight[i].wait()
            if rank == 0:
                seen.append((steps - 2 + k, gathered[i].clone()))
    if rank == 0:
        ok = sorted(s for s, _ in seen) == list(range(steps))
        for step, g in seen:
            for r in range(world):
                peer = ResultBlock(n, limit, buf=g[r])
                ok = ok and bool((peer.rows == 1000 * step + r).all()) and bool((peer.counts == step).all())
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gathers_overlap_the_next_step(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_pipelined_worker, args=(2, _free_port(), 5, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("n_needles", [64, 61, 5])      # even; ragged (shards of 8 and 7); fewer needles than ranks
def test_world_size_8_gather_reassembles_in_order(tmp_path, n_needles):
    """configs[3]'s rank count: eight contiguous shards (ragged, some empty), one gather."""
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(8, _free_port(), n_needles, 5, str(out)), nprocs=8, join=True)
    assert out.read_text() == "ok"


def test_world_size_8_gathers_overlap_the_next_step(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_pipelined_worker, args=(8, _free_port(), 5, str(out)), nprocs=8, join=True)
    assert out.read_text() == "ok"
