"""Multi-GPU find: replicate the index, shard the needle batch, gather the result blocks.

Needles are independent once the index is built, so the only
exchange step is collecting every rank's fixed-stride result block on rank 0: ONE gather over
RCCL/xGMI (backend "nccl" is RCCL on ROCm), no all-reduce.  The same code runs on CPU tensors
over gloo, which is how tests cover world_size 2 without GPUs.
"""


def shard_bounds(n_total, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_results(dist, results, counts, gathered, rank, dst=0):
    """Collect per-rank (results[n, limit, 3], counts[n]) on rank `dst`.

    `gathered` is (out_results[world, n, limit, 3], out_counts[world, n]) on `dst`, None
    elsewhere.  Every rank passes equally shaped tensors (pad the last shard if needed).
    """
    if rank == dst:
        out_r, out_c = gathered
        dist.gather(results, gather_list=list(out_r.unbind(0)), dst=dst)
        dist.gather(counts, gather_list=list(out_c.unbind(0)), dst=dst)
    else:
        dist.gather(results, gather_list=None, dst=dst)
        dist.gather(counts, gather_list=None, dst=dst)


def find_batch_sharded(dist, find_fn, needles, limit, rank, world, dst=0):
    """Host-level helper: every rank holds the full `needles` list, computes its contiguous
    shard with `find_fn(list_of_needles, limit) -> (rows[n, limit, 3], counts[n])` (torch
    tensors on the rank's device) and rank `dst` returns the reassembled
    (rows[n_total, limit, 3], counts[n_total]); other ranks return None."""
    import torch
    n_total = len(needles)
    lo, hi = shard_bounds(n_total, world, rank)
    width = shard_bounds(n_total, world, 0)[1]            # widest shard
    rows, counts = find_fn(needles[lo:hi], limit)
    pad = width - (hi - lo)
    if pad:
        rows = torch.cat([rows, rows.new_zeros((pad,) + tuple(rows.shape[1:]))])
        counts = torch.cat([counts, counts.new_zeros((pad,))])
    rows, counts = rows.contiguous(), counts.contiguous()
    gathered = None
    if rank == dst:
        gathered = (rows.new_empty((world,) + tuple(rows.shape)), counts.new_empty((world,) + tuple(counts.shape)))
    gather_results(dist, rows, counts, gathered, rank, dst)
    if rank != dst:
        return None
    out_r, out_c = [], []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        out_r.append(gathered[0][r, :b - a])
        out_c.append(gathered[1][r, :b - a])
    return torch.cat(out_r), torch.cat(out_c)
