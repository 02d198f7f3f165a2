"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on the GPUs,
"gloo" in the CPU tests).  Round-1 sharding (Multi-GPU): reads are split into contiguous
shards, every rank builds the read DB of its shard against a replicated reference DB -- the reference's
own multi-file search convention (`reads.db.1 .. .N`, Database/Matcher.java:111-128).  The only
cross-rank quantities are prefix sums: the global index of a shard's first read (ReadIDMapping numbers
reads across the whole input, OtherFiles/ReadIDMapping.java:14-23) and job totals.

Key-range sharding, for databases too large to replicate: rank g owns a contiguous range of the
4^l key buckets (`balanced_key_ranges` over the all-reduced bucket histogram).  Every rank builds the database
of its own input shard, cuts it at the range bounds -- the k-mers are sorted, so every destination's part is a
contiguous CSR slice, no scatter kernel is needed -- and one all-to-all-v per array (`exchange_csr`) sends the
parts to their owners, which merge them on the GPU (`kmg_db_merge`: counts add, read lists unite).  Matching
against a range-sharded reference is the same cut on the search k-mers, one all-to-all of keys, a local join,
and the reverse all-to-all of the hit indices (`match_by_key_range`).

On GPUs both exchange steps live under the C ABI (csrc/comm.h: kmg_match_sharded, kmg_db_exchange_by_key_range --
grouped ncclSend / ncclRecv on the library's stream), so that a JNI / FFM caller reaches them too; `Comm` below is
their Python face and needs torch.distributed only to hand rank 0's RCCL id to the other ranks.  The tensor versions
(`exchange_csr_tensors`, `match_by_key_range_tensors`; `torch.distributed.all_to_all_single`) are the same
arithmetic on torch tensors: the CPU tests run them on gloo against the single-process oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


class Comm:
    """kmg_comm: this rank's end of the RCCL communicator the library's exchange steps run on (one per context)."""

    def __init__(self, ctx, rank=None, world=None, group=None):
        import ctypes as C
        self.ctx = ctx
        self.rank = (dist.get_rank(group) if dist.is_initialized() else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if dist.is_initialized() else 1) if world is None else world
        ident = np.zeros(128, np.uint8)
        if self.rank == 0:
            rc = ctx.lib.kmg_comm_unique_id(ident.ctypes.data_as(C.c_void_p))
            if rc != 0:
                raise RuntimeError("kmg_comm_unique_id failed (%d): is RCCL installed?" % rc)
        if self.world > 1:
            # rank 0's id travels over the process group the caller already has (any backend)
            t = torch.from_numpy(ident.astype(np.int32))
            backend = dist.get_backend(group)
            t = t.to(torch.device("cuda", ctx.device)) if backend == "nccl" else t
            dist.broadcast(t, src=0, group=group)
            ident = t.cpu().numpy().astype(np.uint8)
        h = C.c_void_p()
        ctx.check(ctx.lib.kmg_comm_create(ctx.h, self.world, self.rank, ident.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h

    @staticmethod
    def bounds(ranges):
        """(first, end) bucket ranges of the ranks, contiguous and ascending -> the nranks + 1 bounds the C ABI takes."""
        b = [int(r[0]) for r in ranges] + [int(ranges[-1][1])]
        for (a0, a1), nxt in zip(ranges, b[1:]):
            if int(a1) != nxt:
                raise ValueError("key ranges must be contiguous")
        return np.array(b, np.uint32)

    def match_sharded(self, search_db, ref_shard, ranges, out=None):
        """kmg_match_sharded -> int64 tensor on the context's GPU: for every k-mer of search_db its hit index inside
        the OWNER's shard, -1 = none."""
        import ctypes as C
        dev = torch.device("cuda", self.ctx.device)
        if out is None:
            out = torch.empty(search_db.n_unique, dtype=torch.int64, device=dev)
        b = self.bounds(ranges)
        torch.cuda.current_stream(dev).synchronize()          # `out` may still be in use on torch's stream
        self.ctx.check(self.ctx.lib.kmg_match_sharded(self.h, search_db.h, ref_shard.h, b.ctypes.data_as(C.c_void_p),
                                                      out.data_ptr() if out.numel() else None))
        torch.cuda.ExternalStream(self.ctx.stream, device=dev).synchronize()
        return out

    def exchange_by_key_range(self, local_db, ranges):
        """kmg_db_exchange_by_key_range -> KmerDatabase: every rank's k-mers of THIS rank's key range, merged."""
        import ctypes as C
        from .api import KmerDatabase
        b = self.bounds(ranges)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.kmg_db_exchange_by_key_range(self.h, local_db.h, b.ctypes.data_as(C.c_void_p), C.byref(h)))
        return KmerDatabase(self.ctx, h)

    def close(self):
        if self.h:
            self.ctx.lib.kmg_comm_destroy(self.h)
            self.h = None


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


def taxa_counts_sharded(taxon, match, unique, device="cpu", group=None):
    """TaxaCounts over a key-range-sharded reference: every rank counts its own shard
    (kmg_taxacounts: ascending taxa with their match / unique counts); a k-mer lives in exactly one shard, so
    the whole database's counts are the per-taxon sums.  The taxon lists differ between ranks, so the triples
    are all-gathered (padded to the longest list) and summed by taxon on every rank.
    -> (taxon[], match[], unique[]) ascending by taxon, identical on all ranks."""
    taxon = np.asarray(taxon, np.int64)
    rows = np.stack([taxon, np.asarray(match, np.int64), np.asarray(unique, np.int64)], axis=1) if len(taxon) else np.zeros((0, 3), np.int64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        n = torch.tensor([len(rows)], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(x.item()) for x in sizes]
        width = max(max(sizes), 1)
        mine = torch.zeros((width, 3), dtype=torch.int64, device=device)
        if len(rows):
            mine[:len(rows)] = torch.from_numpy(rows).to(device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        rows = np.concatenate([p.cpu().numpy()[:sz] for p, sz in zip(parts, sizes)]) if sum(sizes) else np.zeros((0, 3), np.int64)
    if len(rows) == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64)
    t, inv = np.unique(rows[:, 0], return_inverse=True)
    m = np.bincount(inv, weights=rows[:, 1], minlength=len(t)).astype(np.int64)
    q = np.bincount(inv, weights=rows[:, 2], minlength=len(t)).astype(np.int64)
    return t.astype(np.int32), m, q


def balanced_key_ranges(bucket_hist, world):
    """Split the 4^l key buckets into `world` contiguous ranges of near-equal record counts (skew-aware:
    poly-A piles into bucket 0).  -> list of (first_key, end_key).  Every rank computes the same answer
    from the all-reduced histogram, so no further communication is needed.  A range can come out empty
    (first == end) when one bucket holds more than a rank's share or the world is larger than the populated
    buckets: kmg_build_set_key_range takes it (an empty database), the exchange sends that rank nothing."""
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


def range_lower_bounds(ranges, key_len):
    """First possible k-mer (left-aligned u64) of every range, plus the end sentinel."""
    shift = np.uint64(64 - 2 * key_len)
    lows = [np.uint64(r[0]) << shift for r in ranges]
    return np.array(lows, dtype=np.uint64)


def cut_points(sorted_keys, ranges, key_len):
    """Indices cutting ascending `sorted_keys` into one contiguous slice per range: slice g = [c[g], c[g+1])."""
    keys = np.asarray(sorted_keys, dtype=np.uint64)
    lows = range_lower_bounds(ranges, key_len)
    c = np.searchsorted(keys, lows, side="left").astype(np.int64)
    c[0] = 0
    return np.concatenate([c, [len(keys)]])


def all_to_all_tensor(t, send_counts, group=None):
    """1-D tensor `t` (on the CPU under gloo, on this rank's GPU under nccl), cut into consecutive chunks of
    send_counts[g] elements for rank g  ->  (tensor of everything received, in rank order; recv_counts).
    Two collectives: the sizes, then the data; nothing leaves the tensor's device."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    send_counts = [int(x) for x in send_counts]
    assert sum(send_counts) == t.numel()
    if world == 1:
        return t, send_counts
    assert len(send_counts) == world
    ss = torch.tensor(send_counts, dtype=torch.int64, device=t.device)
    rs_t = torch.zeros(world, dtype=torch.int64, device=t.device)
    dist.all_to_all_single(rs_t, ss, group=group)
    rs = [int(x) for x in rs_t.cpu().tolist()]
    recv = torch.empty(sum(rs), dtype=t.dtype, device=t.device)
    dist.all_to_all_single(recv, t.contiguous(), output_split_sizes=rs, input_split_sizes=send_counts, group=group)
    return recv, rs


def all_to_all_v(chunks, dtype, device="cpu", group=None):
    """numpy front of all_to_all_tensor: chunks[g] = 1-D array for rank g -> list of arrays, one from every rank."""
    chunks = [np.ascontiguousarray(c, dtype=dtype) for c in chunks]
    # torch has no uint64 collectives: ship the bits as int64 / int32
    wire = {8: np.int64, 4: np.int32}[np.dtype(dtype).itemsize]
    flat = np.concatenate(chunks) if chunks else np.zeros(0, dtype)
    recv, rs = all_to_all_tensor(torch.from_numpy(flat.view(wire).copy()).to(device), [len(c) for c in chunks], group)
    out = recv.cpu().numpy().view(dtype)
    offs = np.concatenate([[0], np.cumsum(rs)])
    return [out[offs[i]:offs[i + 1]] for i in range(len(rs))]


def exchange_csr_tensors(keys, val_off, val_a, val_b, cuts, group=None):
    """The exchange itself, on tensors that stay where they are (GPU tensors under nccl: RCCL moves them over xGMI
    peer to peer; CPU tensors under gloo).  keys int64[U] (the u64 bits), val_off int64[U+1], val_a / val_b
    int32[V]; cuts[g] .. cuts[g+1] = the k-mers that go to rank g.  -> list of (keys, val_off, val_a, val_b), one
    complete CSR (val_off from 0) per source rank.  Four data collectives, whatever the number of ranks."""
    world = len(cuts) - 1
    cuts_t = torch.as_tensor(cuts, dtype=torch.int64, device=keys.device)
    vcuts = [int(x) for x in val_off[cuts_t].cpu().tolist()]
    kc = [int(cuts[g + 1] - cuts[g]) for g in range(world)]
    vc = [vcuts[g + 1] - vcuts[g] for g in range(world)]
    lo, hi = int(cuts[0]), int(cuts[-1])
    rk, rkc = all_to_all_tensor(keys[lo:hi], kc, group)
    rn, _ = all_to_all_tensor(val_off[lo + 1:hi + 1] - val_off[lo:hi], kc, group)        # values per k-mer
    ra, rvc = all_to_all_tensor(val_a[vcuts[0]:vcuts[-1]], vc, group)
    rb, _ = all_to_all_tensor(val_b[vcuts[0]:vcuts[-1]], vc, group)
    parts, ko, vo = [], 0, 0
    for nk, nv in zip(rkc, rvc):
        off = torch.zeros(nk + 1, dtype=torch.int64, device=keys.device)
        torch.cumsum(rn[ko:ko + nk], 0, out=off[1:])
        parts.append((rk[ko:ko + nk], off, ra[vo:vo + nv], rb[vo:vo + nv]))
        ko += nk
        vo += nv
    return parts


def exchange_csr(csr, ranges, key_len, device="cpu", group=None):
    """csr = dict(keys, val_off, val_a, val_b) of this rank's database as host arrays (kmg_db_fetch layout).
    Sends every key range's slice to its owner; returns the list of CSR parts received, one per source rank,
    each again a complete CSR (val_off starting at 0) ready for kmg_db_load + kmg_db_merge."""
    keys = np.ascontiguousarray(csr["keys"], np.uint64)
    c = cut_points(keys, ranges, key_len)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(device)
    parts = exchange_csr_tensors(t(keys, np.int64), t(np.asarray(csr["val_off"], np.uint64), np.int64),
                                 t(np.asarray(csr["val_a"], np.int32), np.int32),
                                 t(np.asarray(csr["val_b"], np.int32), np.int32), c, group)
    return [dict(keys=k.cpu().numpy().view(np.uint64).copy(), val_off=o.cpu().numpy().view(np.uint64).copy(),
                 val_a=a.cpu().numpy().copy(), val_b=b.cpu().numpy().copy()) for k, o, a, b in parts]


def build_key_range_shard(ctx, local_db, ranges, device=None, group=None, comm=None):
    """(comm given: kmg_db_exchange_by_key_range does all of it under the C ABI.)
    The exchange step of a sharded build: local_db (KmerDatabase of this rank's input shard) -> KmerDatabase
    holding every rank's k-mers of THIS rank's key range, merged on the GPU.  Device-resident end to end: the
    CSR arrays are copied out of the library into torch tensors on the context's GPU (kmg_db_copy_dev), cut at
    the bucket offsets the database already holds, exchanged GPU to GPU and handed back with kmg_db_load_dev."""
    from .api import KmerDatabase, merge_databases
    if comm is not None:
        return comm.exchange_by_key_range(local_db, ranges)
    dev = torch.device("cuda", ctx.device) if device is None else torch.device(device)
    keys, val_off, val_a, val_b = local_db.to_tensors(dev)
    boff = local_db.bucket_offsets()
    cuts = [int(boff[r[0]]) for r in ranges] + [int(boff[ranges[-1][1]])]
    parts = exchange_csr_tensors(keys, val_off, val_a, val_b, cuts, group)
    torch.cuda.synchronize(dev)                    # the library reads the parts on its own stream
    dbs = [KmerDatabase.load_tensors(ctx, local_db.k, local_db.key_length, local_db.rc, local_db.kind, *p) for p in parts]
    out = dbs[0] if len(dbs) == 1 else merge_databases(dbs)
    for d in dbs:
        if d is not out:
            d.close()
    return out


def match_by_key_range_tensors(keys, cuts, local_match, group=None, want_owner=True):
    """Join ascending search k-mers (int64 tensor holding the u64 bits, on the CPU or on this rank's GPU) against a
    key-range-sharded reference: keys[cuts[g] : cuts[g+1]] go to rank g, every owner joins what it received in ONE
    call -- local_match(tensor of everything received, counts per source rank: one ascending run per source, all of
    them inside the owner's range) -> int64 hit indices into its shard, -1 = miss -- and the reverse all-to-all
    brings the hits back in the order of `keys`.  -> (owner int32[n], hit int64[n])."""
    world = len(cuts) - 1
    kc = [int(cuts[g + 1] - cuts[g]) for g in range(world)]
    recv, rc = all_to_all_tensor(keys[int(cuts[0]):int(cuts[-1])], kc, group)
    hits = local_match(recv, rc)
    back, _ = all_to_all_tensor(hits, rc, group)
    if not want_owner:          # (the owner of k-mer i is the range i falls in: cuts says it all)
        return None, back
    # (one slice fill per rank: torch.repeat_interleave took 18 ms for 35 M k-mers, these take microseconds)
    owner = torch.empty(int(sum(kc)), dtype=torch.int32, device=keys.device)
    o = 0
    for g_, n in enumerate(kc):
        owner[o:o + n] = g_
        o += n
    return owner, back


def match_by_key_range(search_keys, ranges, key_len, local_match, device="cpu", group=None):
    """numpy front of match_by_key_range_tensors.  local_match(keys) -> int64 hit index into THIS rank's reference
    shard (-1 = miss) for an ascending u64 array (called once per source rank's run).  Returns (owner, hit): for
    every search k-mer the rank that owns its range and the hit index inside that rank's shard (-1 = miss)."""
    keys = np.ascontiguousarray(search_keys, np.uint64)
    c = cut_points(keys, ranges, key_len)

    def lm(t, counts):
        a = t.cpu().numpy().view(np.uint64)
        h = np.empty(len(a), np.int64)
        o = 0
        for n in counts:
            if n:
                h[o:o + n] = np.asarray(local_match(a[o:o + n]), np.int64)
            o += n
        return torch.from_numpy(h).to(t.device)

    owner, hit = match_by_key_range_tensors(torch.from_numpy(keys.view(np.int64).copy()).to(device), c, lm, group)
    return owner.cpu().numpy(), hit.cpu().numpy()


def shard_matcher(ctx, like_db, ref_shard):
    """-> local_match for match_by_key_range_tensors on the GPU: everything the owner received (one ascending run
    per source rank) is joined against ref_shard in one launch (kmg_match_keys_dev), straight from the receive
    buffer."""
    import ctypes as C
    dev = torch.device("cuda", ctx.device)

    def local_match(recv, counts):
        recv = recv.contiguous()
        out = torch.empty(recv.numel(), dtype=torch.int64, device=dev)
        if recv.numel() == 0:
            return out
        off = np.concatenate([[0], np.cumsum(np.asarray(counts, np.uint64))]).astype(np.uint64)
        torch.cuda.current_stream(dev).synchronize()         # the collective that filled `recv` ran on torch's stream
        ctx.check(ctx.lib.kmg_match_keys_dev(ctx.h, recv.data_ptr(), off.ctypes.data_as(C.c_void_p), len(counts), like_db.k,
                                             ref_shard.h, out.data_ptr()))
        torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
        return out

    return local_match


def match_key_range_shard(ctx, search_db, ref_shard, ranges, group=None, want_owner=True, comm=None):
    """The sharded Matcher join, device-resident: search_db's k-mers (this rank's read database) are cut at the
    key-range bounds, exchanged GPU to GPU, joined by every owner against its reference shard and the hit indices
    come back the same way.  -> (owner int32 tensor, hit int64 tensor) on the context's GPU, one entry per k-mer
    of search_db.  comm (a Comm): the whole step runs under the C ABI (kmg_match_sharded); without one, the torch
    version of the same arithmetic."""
    dev = torch.device("cuda", ctx.device)
    if comm is not None:
        hit = comm.match_sharded(search_db, ref_shard, ranges)
        if not want_owner:
            return None, hit
        boff = search_db.bucket_offsets()
        owner = torch.full((search_db.n_unique,), -1, dtype=torch.int32, device=dev)
        for g_, r in enumerate(ranges):
            owner[int(boff[r[0]]):int(boff[r[1]])] = g_
        return owner, hit
    keys = torch.empty(search_db.n_unique, dtype=torch.int64, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    ctx.check(ctx.lib.kmg_db_copy_dev(search_db.h, 0, keys.data_ptr() if keys.numel() else None, 0, keys.numel()))
    boff = search_db.bucket_offsets()
    cuts = [int(boff[r[0]]) for r in ranges] + [int(boff[ranges[-1][1]])]
    return match_by_key_range_tensors(keys, cuts, shard_matcher(ctx, search_db, ref_shard), group, want_owner)
