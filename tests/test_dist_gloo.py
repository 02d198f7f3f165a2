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
