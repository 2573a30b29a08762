"""Multi-GPU find: replicate the index, shard the needle batch, gather the result blocks.

Needles are independent once the index is built, so the only
exchange step is collecting every rank's fixed-size result block on rank 0: ONE gather over
RCCL/xGMI (backend "nccl" is RCCL on ROCm), no all-reduce.  A rank's block is one contiguous
buffer of ``n * (limit * 12 + 4)`` bytes -- the ``limit`` packed 12-byte rows of every needle
(include/blurrily_storage.h: trigram_match_t) followed by the per-needle row counts -- which
the find kernels write in place (`ResultBlock.rows` / `.counts` are views of it), so nothing is
repacked before the collective.  The same code runs on CPU tensors over gloo, which is how
tests cover world_size 2 without GPUs.
"""


def shard_bounds(n_total, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def block_bytes(n, limit):
    """Size of one rank's result block: `limit` 12-byte rows and one 4-byte count per needle."""
    return int(n) * (int(limit) * 12 + 4)


class ResultBlock:
    """One rank's results as ONE buffer: int32 words [n*limit*3 rows | n counts]."""

    def __init__(self, n, limit, device=None, buf=None):
        import torch
        self.n, self.limit = int(n), int(limit)
        words = block_bytes(n, limit) // 4
        self.buf = torch.zeros(words, dtype=torch.int32, device=device) if buf is None else buf
        assert self.buf.numel() == words and self.buf.dtype == torch.int32
        self.rows = self.buf[:self.n * self.limit * 3].view(self.n, self.limit, 3)
        self.counts = self.buf[self.n * self.limit * 3:]


def gather_blocks(dist, block, gathered, rank, dst=0, async_op=False):
    """THE collective of the path: every rank's block to rank `dst` in one gather.

    `block` is a ResultBlock (all ranks: same n and limit -- pad the last shard); `gathered` is
    an int32 tensor [world, words] on `dst` (None elsewhere).  With `async_op` the call returns the
    collective's work handle at once: the gather of one batch then travels over xGMI while the next
    batch is being searched (into another block -- `block` and `gathered` stay the collective's until
    `wait()`)."""
    if rank == dst:
        return dist.gather(block.buf, gather_list=list(gathered.unbind(0)), dst=dst, async_op=async_op)
    return dist.gather(block.buf, gather_list=None, dst=dst, async_op=async_op)


def find_batch_sharded(dist, find_fn, needles, limit, rank, world, dst=0, device=None):
    """Host-level helper: every rank holds the full `needles` list, computes its contiguous
    shard with `find_fn(list_of_needles, limit, rows_out[n, limit, 3], counts_out[n])` (int32
    views of this rank's ResultBlock, to be filled in place) and rank `dst` returns the
    reassembled (rows[n_total, limit, 3], counts[n_total]); other ranks return None."""
    import torch
    n_total = len(needles)
    lo, hi = shard_bounds(n_total, world, rank)
    width = shard_bounds(n_total, world, 0)[1]            # widest shard: every block has this many slots
    block = ResultBlock(width, limit, device=device)
    find_fn(needles[lo:hi], limit, block.rows[:hi - lo], block.counts[:hi - lo])
    gathered = None
    if rank == dst:
        gathered = torch.empty((world, block.buf.numel()), dtype=torch.int32, device=block.buf.device)
    gather_blocks(dist, block, gathered, rank, dst)
    if rank != dst:
        return None
    out_r, out_c = [], []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        peer = ResultBlock(width, limit, buf=gathered[r])
        out_r.append(peer.rows[:b - a])
        out_c.append(peer.counts[:b - a])
    return torch.cat(out_r), torch.cat(out_c)
