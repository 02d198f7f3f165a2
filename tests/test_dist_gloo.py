"""N > 1 path on CPU: world_size-2 gloo.  The shards' read DBs (built here by the CPU oracle standing in
for the GPU build) with globally shifted read indices must add up to the single-process database."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fq_records, k, key_len, out):
    import torch.distributed as dist
    from kmers_amd import dist as kd
    from oracle.c_oracle import OracleDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = kd.shard_range(len(fq_records), rank, world)
    db = OracleDB(b"".join(fq_records[lo:hi]), 1, k, key_len)
    n_local = len(db.names)
    base, total = kd.exclusive_offset(n_local)
    hist = np.diff(db.val_off.astype(np.int64)[db.bucket_off.astype(np.int64)])   # records per key bucket
    ghist = kd.allreduce_sum(hist)
    # expand the shard's CSR to (key, read + base, pos) rows
    counts = np.diff(db.val_off.astype(np.int64))
    rows = np.stack([np.repeat(db.keys, counts).astype(np.uint64).view(np.int64),
                     db.val_a.astype(np.int64) + base, db.val_b.astype(np.int64)], axis=1)
    out.put((rank, rows, db.names, base, total, ghist))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_read_shards_add_up_to_the_whole():
    from kmers_amd import dist as kd, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(5, 3, 3000)
    reads = synth.random_reads(6, genomes, 301, 80, 0.02, n_rate=0.01)
    reads[7, :] = ord("N")                          # a read that gets no index
    recs = [synth.fastq_bytes(reads[i:i + 1], first_index=i) for i in range(reads.shape[0])]
    k, key_len, world = 15, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, recs, k, key_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = OracleDB(b"".join(recs), 1, k, key_len)
    counts = np.diff(whole.val_off.astype(np.int64))
    want = np.stack([np.repeat(whole.keys, counts).astype(np.uint64).view(np.int64),
                     whole.val_a.astype(np.int64), whole.val_b.astype(np.int64)], axis=1)
    rows = np.concatenate([g[1] for g in got])
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0].view(np.uint64)))]
    assert np.array_equal(key(rows), key(want))
    assert got[0][2] + got[1][2] == whole.names                      # read-name map = shards' maps in order
    assert got[0][3] == 0 and got[1][3] == len(got[0][2]) and got[0][4] == got[1][4] == len(whole.names)
    assert np.array_equal(got[0][5], np.diff(whole.val_off.astype(np.int64)[whole.bucket_off.astype(np.int64)]))
    ranges = kd.balanced_key_ranges(got[0][5], 4)
    assert ranges[0][0] == 0 and ranges[-1][1] == 4 ** key_len
    assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))


def test_shard_range_and_balanced_ranges():
    from kmers_amd import dist as kd
    assert [kd.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [kd.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    hist = np.array([1000, 1, 1, 1, 1, 1, 1, 994])                   # skew at both ends
    r = kd.balanced_key_ranges(hist, 2)
    assert r == [(0, 1), (1, 8)]
    assert kd.balanced_key_ranges(np.zeros(16, int), 4)[-1][1] == 16


# ---------------------------------------------------------------------------- key-range exchange
def _merge_csr_numpy(parts, taxon_kind=True):
    """What kmg_db_merge computes, in numpy: (key, taxon) pairs of all parts, equal pairs' counts added."""
    rows = []
    for p in parts:
        n = np.diff(p["val_off"].astype(np.int64))
        rows.append(np.stack([np.repeat(p["keys"], n).view(np.int64), p["val_a"].astype(np.int64),
                              p["val_b"].astype(np.int64)], axis=1))
    rows = np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)
    order = np.lexsort((rows[:, 1], rows[:, 0].view(np.uint64)))
    rows = rows[order]
    if len(rows) == 0:
        return dict(keys=np.zeros(0, np.uint64), val_off=np.zeros(1, np.uint64), val_a=np.zeros(0, np.int32),
                    val_b=np.zeros(0, np.int32))
    newpair = np.ones(len(rows), bool)
    newpair[1:] = (rows[1:, 0] != rows[:-1, 0]) | (rows[1:, 1] != rows[:-1, 1])
    pair_id = np.cumsum(newpair) - 1
    counts = np.bincount(pair_id, weights=rows[:, 2]).astype(np.int64)
    pk, pt = rows[newpair, 0], rows[newpair, 1]
    newkey = np.ones(len(pk), bool)
    newkey[1:] = pk[1:] != pk[:-1]
    val_off = np.concatenate([np.nonzero(newkey)[0], [len(pk)]]).astype(np.uint64)
    return dict(keys=pk[newkey].view(np.uint64), val_off=val_off, val_a=pt.astype(np.int32), val_b=counts.astype(np.int32))


def _exchange_worker(rank, world, port, fastas, fq, k, key_len, out):
    import torch.distributed as dist
    from kmers_amd import dist as kd
    from oracle.c_oracle import OracleDB
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = OracleDB(fastas[rank], 0, k, key_len)            # stands in for the GPU build of this rank's shard
    hist = np.diff(db.val_off.astype(np.int64)[db.bucket_off.astype(np.int64)])
    ranges = kd.balanced_key_ranges(kd.allreduce_sum(hist), world)
    csr = dict(keys=db.keys, val_off=db.val_off, val_a=db.val_a, val_b=db.val_b)
    parts = kd.exchange_csr(csr, ranges, key_len)
    shard = _merge_csr_numpy(parts)
    # sharded match: this rank's half of the read k-mers against the range-sharded reference
    reads = OracleDB(fq, 1, k, key_len)
    lo, hi = kd.shard_range(len(reads.keys), rank, world)
    mine = reads.keys[lo:hi]

    def local_match(q):
        i = np.searchsorted(shard["keys"], q)
        ok = (i < len(shard["keys"])) & (shard["keys"][np.minimum(i, max(len(shard["keys"]) - 1, 0))] == q) if len(shard["keys"]) else np.zeros(len(q), bool)
        return np.where(ok, i, -1)

    owner, hit = kd.match_by_key_range(mine, ranges, key_len, local_match)
    # TaxaCounts of the shard (what kmg_taxacounts returns for it), then the cross-rank sum
    n_tax = np.diff(shard["val_off"].astype(np.int64))
    st, sm = np.unique(shard["val_a"], return_counts=True)
    su = np.zeros(len(st), np.int64)
    single = shard["val_a"][shard["val_off"].astype(np.int64)[:-1][n_tax == 1]]
    ut, uc = np.unique(single, return_counts=True)
    su[np.searchsorted(st, ut)] = uc
    counts = kd.taxa_counts_sharded(st, sm, su)
    # a read database goes through the same exchange: this rank holds the occurrences of every other read
    sv = reads.val_off.astype(np.int64)
    kidx = np.repeat(np.arange(len(reads.keys)), np.diff(sv))
    mine_v = reads.val_a % world == rank
    cnt = np.bincount(kidx[mine_v], minlength=len(reads.keys))
    has = cnt > 0
    part = dict(keys=reads.keys[has], val_off=np.concatenate([[0], np.cumsum(cnt[has])]).astype(np.uint64),
                val_a=reads.val_a[mine_v], val_b=reads.val_b[mine_v])
    got_parts = kd.exchange_csr(part, ranges, key_len)
    rows = np.concatenate([np.stack([np.repeat(p["keys"], np.diff(p["val_off"].astype(np.int64))).view(np.int64),
                                     p["val_a"].astype(np.int64), p["val_b"].astype(np.int64)], axis=1) for p in got_parts])
    rows = rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0].view(np.uint64)))]          # k-mer, read, pos
    out.put((rank, ranges, shard, lo, owner, hit, counts, rows))
    dist.barrier()
    dist.destroy_process_group()


def test_key_range_exchange_build_and_match():
    """Two ranks, each with half of the reference FASTA: after the exchange + merge, rank g holds exactly the
    whole database's k-mers of its key range (counts of k-mers present in both halves added), and the sharded
    match returns, for every read k-mer, the same reference k-mer as the single-process join."""
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(15, 4, 2500, shared_frac=0.5)
    genomes = list(genomes) + [genomes[0]]                       # taxon 0's genome again in the second half
    taxa = list(taxa) + [taxa[0]]
    reads = synth.random_reads(16, genomes[:4], 200, 60, 0.01)
    k, key_len, world = 13, 3, 2
    fastas = [synth.fasta_bytes(taxa[:2], genomes[:2]), synth.fasta_bytes(taxa[2:], genomes[2:])]
    fq = synth.fastq_bytes(reads)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, fastas, fq, k, key_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = OracleDB(synth.fasta_bytes(taxa, genomes), 0, k, key_len)
    ranges = got[0][1]
    assert ranges == got[1][1] and ranges[0][0] == 0 and ranges[-1][1] == 4 ** key_len
    starts = []
    for g in range(world):
        u0, u1 = int(whole.bucket_off[ranges[g][0]]), int(whole.bucket_off[ranges[g][1]])
        starts.append(u0)
        shard = got[g][2]
        assert np.array_equal(shard["keys"], whole.keys[u0:u1])
        v0, v1 = int(whole.val_off[u0]), int(whole.val_off[u1])
        assert np.array_equal(shard["val_off"].astype(np.int64), whole.val_off[u0:u1 + 1].astype(np.int64) - v0)
        assert np.array_equal(shard["val_a"], whole.val_a[v0:v1]) and np.array_equal(shard["val_b"], whole.val_b[v0:v1])
    assert whole.val_b.max() >= 2                                 # counts really were added across ranks
    o_reads = OracleDB(fq, 1, k, key_len)
    o_hit, o_nh = o_reads.match(whole)
    glob = np.full(len(o_reads.keys), -2, np.int64)
    for g in range(world):
        lo, owner, hit = got[g][3], got[g][4], got[g][5]
        glob[lo:lo + len(hit)] = np.where(hit >= 0, hit + np.array(starts)[owner], -1)
    assert np.array_equal(glob, o_hit) and o_nh > 0
    t, m, q = whole.taxacounts()
    for g in range(world):                                         # every rank ends up with the whole database's counts
        gt, gm, gq = got[g][6]
        assert np.array_equal(gt, t) and np.array_equal(gm, m) and np.array_equal(gq, q)
    assert (np.asarray(q) < np.asarray(m)).any()
    # read database: rank g ends up with every occurrence of the k-mers of its range, whoever held them
    rsv = o_reads.val_off.astype(np.int64)
    for g in range(world):
        shift = np.uint64(64 - 2 * key_len)
        lo_key, hi_key = np.uint64(ranges[g][0]) << shift, (np.uint64(ranges[g][1]) << shift if ranges[g][1] < 4 ** key_len else None)
        sel = (o_reads.keys >= lo_key) & ((o_reads.keys < hi_key) if hi_key is not None else True)
        idx = np.nonzero(sel)[0]
        want_rows = np.stack([np.repeat(o_reads.keys[idx], rsv[idx + 1] - rsv[idx]).view(np.int64),
                              np.concatenate([o_reads.val_a[rsv[i]:rsv[i + 1]] for i in idx]).astype(np.int64),
                              np.concatenate([o_reads.val_b[rsv[i]:rsv[i + 1]] for i in idx]).astype(np.int64)], axis=1)
        assert np.array_equal(got[g][7], want_rows)
