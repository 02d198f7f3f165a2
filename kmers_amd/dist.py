"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on the GPUs,
"gloo" in the CPU tests).  Round-1 sharding (Multi-GPU): reads are split into contiguous
shards, every rank builds the read DB of its shard against a replicated reference DB -- the reference's
own multi-file search convention (`reads.db.1 .. .N`, Database/Matcher.java:111-128).  The only
cross-rank quantities are prefix sums: the global index of a shard's first read (ReadIDMapping numbers
reads across the whole input, OtherFiles/ReadIDMapping.java:14-23) and job totals.
`balanced_key_ranges` is the planning step of the key-range exchange, not used yet.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def exclusive_offset(value, device="cpu", group=None):
    """-> (sum of `value` over lower ranks, sum over all ranks).  One all_gather of a single int64."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, int(value)
    world = dist.get_world_size(group)
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    vals = [int(p.item()) for p in parts]
    r = dist.get_rank(group)
    return sum(vals[:r]), sum(vals)


def global_read_indices(local_read_index, device="cpu", group=None):
    """Shift a shard's read numbering (kmg_build_add's read_index_out, -1 = no k-mer) to the numbering
    the reference would give the whole input: add the number of reads numbered by lower ranks."""
    local_read_index = np.asarray(local_read_index, dtype=np.int64)
    n_local = int(local_read_index.max()) + 1 if local_read_index.size and local_read_index.max() >= 0 else 0
    base, total = exclusive_offset(n_local, device, group)
    out = np.where(local_read_index >= 0, local_read_index + base, -1)
    return out, base, total


def allreduce_sum(array, device="cpu", group=None):
    """Element-wise sum of an int64 numpy array over ranks (bucket histograms, hit counts)."""
    t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.int64)).to(device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def balanced_key_ranges(bucket_hist, world):
    """Split the 4^l key buckets into `world` contiguous ranges of near-equal record counts (skew-aware:
    poly-A piles into bucket 0).  -> list of (first_key, end_key).  Every rank computes the same answer
    from the all-reduced histogram, so no further communication is needed."""
    h = np.asarray(bucket_hist, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(h)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        k = int(np.searchsorted(csum, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), len(h)))
    cuts.append(len(h))
    return [(cuts[i], cuts[i + 1]) for i in range(world)]
