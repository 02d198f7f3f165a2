"""One rank of tests/test_exchange_two_ranks.py (started as a fresh process, `python tests/_exchange_rank.py rank nranks
dir`).  Maps the test-only transport double (tests/transport_double: the nine RCCL symbols csrc/comm.h looks up, host
staged through shared memory) BEFORE the library's first kmg_comm_* call, makes its own kmg_ctx on GPU 0 and runs the
library's two exchange steps -- kmg_match_sharded, kmg_db_exchange_by_key_range -- as rank `rank` of `nranks`, every
result against the CPU oracle.  Never imports torch (PyTorch would bring the real librccl.so.1 into the process)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DOUBLE = os.path.join(ROOT, "tests", "transport_double", "librccl.so.1")


def main():
    rank, nranks, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    real = os.environ.get("KMG_TEST_REAL_RCCL") == "1"        # two real GPUs: the same script on RCCL itself, GPU = rank
    double = None if real else C.CDLL(DOUBLE, mode=C.RTLD_GLOBAL)
    import kmers_amd as km
    from kmers_amd import synth
    from kmers_amd._lib import KMG_EINVAL, KMG_ESTATE
    from oracle.c_oracle import OracleDB
    assert "torch" not in sys.modules

    ctx = km.Context(rank if real else 0)
    lib = ctx.lib
    # rank 0 makes the id (the double's ncclGetUniqueId, through kmg_comm_unique_id), the others read it from a file
    id_path = os.path.join(workdir, "id.bin")
    ident = np.zeros(128, np.uint8)
    if rank == 0:
        assert lib.kmg_comm_unique_id(ident.ctypes.data_as(C.c_void_p)) == 0
        assert real or bytes(ident[:21]) == b"/kmg-transport-double", "the library did not pick the transport double up"
        with open(id_path + ".tmp", "wb") as f:
            f.write(ident.tobytes())
        os.rename(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            assert time.time() - t0 < 120, "rank 0 never wrote the id"
            time.sleep(0.05)
        ident = np.frombuffer(open(id_path, "rb").read(), np.uint8).copy()
    comm = C.c_void_p()
    ctx.check(lib.kmg_comm_create(ctx.h, nranks, rank, ident.ctypes.data_as(C.c_void_p), C.byref(comm)))
    nr, rk = C.c_int(), C.c_int()
    assert lib.kmg_comm_info(comm, C.byref(nr), C.byref(rk)) == 0 and (nr.value, rk.value) == (nranks, rank)

    k, l = (int(sys.argv[4]) if len(sys.argv) > 4 else 21), 4          # (k > 32: two words a k-mer, the second travels too)
    wide = k > 32
    nb = 4 ** l
    taxa, genomes = synth.random_genomes(301, 5, 30000, shared_frac=0.3)
    fa_whole = synth.fasta_bytes(taxa, genomes)
    o_ref = OracleDB(fa_whole, 0, k, l)
    # every rank searches its own reads
    reads = synth.random_reads(302 + rank, genomes, 3000, 100, 0.01 if not wide else 0.004, n_rate=0.002)
    fq = synth.fastq_bytes(reads)
    rdb, _ = km.make_read_db(ctx, fq, k, l)
    o_reads = OracleDB(fq, 1, k, l)
    o_hit, _ = o_reads.match(o_ref)
    assert (o_hit >= 0).sum() > 1000

    def bounds_arr(b):
        return np.array(b, np.uint32)

    # device memory for the hit indices: the HIP runtime itself (no torch in this process)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipSetDevice(rank if real else 0) == 0
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    d_hit = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_hit), (rdb.n_unique + 1) * 8) == 0

    def fetch_hits():
        assert hip.hipDeviceSynchronize() == 0
        out = np.full(rdb.n_unique, -7, np.int64)
        assert hip.hipMemcpy(out.ctypes.data, d_hit, rdb.n_unique * 8, 2) == 0          # hipMemcpyDeviceToHost
        return out

    cases = 0
    # ---- kmg_match_sharded: the reference range-partitioned over the ranks --------------------------------------
    layouts = [
        [0] + [nb * (g + 1) // nranks for g in range(nranks)],                      # the whole key space
        [37] + [37 + (nb - 48) * (g + 1) // nranks for g in range(nranks)],         # first and last buckets without an owner
        [0] * nranks + [nb],                                                        # every rank but the last owns nothing
        [nb // 3] * (nranks + 1),                                                   # nobody owns anything
    ]
    for b in layouts:
        lo, hi = b[rank], b[rank + 1]
        shard = km.make_reference_db(ctx, fa_whole, k, l, key_range=(lo, hi))
        a = bounds_arr(b)
        ctx.check(lib.kmg_match_sharded(comm, rdb.h, shard.h, a.ctypes.data_as(C.c_void_p), d_hit))
        got = fetch_hits()
        # expected: the whole-database join, re-indexed into the owner's shard; -1 outside every range
        want = np.full(rdb.n_unique, -1, np.int64)
        rb = o_reads.bucket_off
        for g in range(nranks):
            glo, ghi = b[g], b[g + 1]
            s0, s1 = int(rb[glo]), int(rb[ghi])
            seg = o_hit[s0:s1]
            want[s0:s1] = np.where(seg >= 0, seg - int(o_ref.bucket_off[glo]), -1)
        assert np.array_equal(got, want), ("match_sharded", b, int((got != want).sum()))
        cases += 1
        shard.close()

    # ---- a rank whose checks fail: every rank gets an error, nobody waits, the communicator stays usable ----------
    shard = km.make_reference_db(ctx, fa_whole, k, l, key_range=(layouts[0][rank], layouts[0][rank + 1]))
    a = bounds_arr(layouts[0])
    bad_ptr = None if rank == nranks - 1 else d_hit          # the last rank passes no hit array
    rc = lib.kmg_match_sharded(comm, rdb.h, shard.h, a.ctypes.data_as(C.c_void_p), bad_ptr)
    assert rc == (KMG_EINVAL if rank == nranks - 1 else KMG_ESTATE), rc
    msg = lib.kmg_last_error(ctx.h).decode()
    assert ("null hit_index" in msg) if rank == nranks - 1 else ("rank %d failed its checks" % (nranks - 1) in msg), msg
    ctx.check(lib.kmg_match_sharded(comm, rdb.h, shard.h, a.ctypes.data_as(C.c_void_p), d_hit))
    assert (fetch_hits() >= 0).sum() > 100
    cases += 1
    shard.close()

    # ---- kmg_db_exchange_by_key_range: every rank built the database of ITS OWN input shard -----------------------
    # rank r holds genomes r, r + nranks, ... and everybody holds genome 0 (its counts add up across the ranks)
    mine = sorted(set([0] + list(range(rank, len(genomes), nranks))))
    fa_mine = synth.fasta_bytes([taxa[i] for i in mine], [genomes[i] for i in mine])
    local = km.make_reference_db(ctx, fa_mine, k, l)
    parts = []
    for r in range(nranks):
        idx = sorted(set([0] + list(range(r, len(genomes), nranks))))
        parts.append(synth.fasta_bytes([taxa[i] for i in idx], [genomes[i] for i in idx]))
    o_all = OracleDB(b"".join(parts), 0, k, l)                      # what one process would build from all the input
    for b in layouts[:3]:
        out = C.c_void_p()
        a = bounds_arr(b)
        ctx.check(lib.kmg_db_exchange_by_key_range(comm, local.h, a.ctypes.data_as(C.c_void_p), C.byref(out)))
        got = km.KmerDatabase(ctx, out)
        h = got.fetch()
        lo, hi = b[rank], b[rank + 1]
        s0, s1 = int(o_all.bucket_off[lo]), int(o_all.bucket_off[hi])
        v0, v1 = int(o_all.val_off[s0]), int(o_all.val_off[s1])
        assert np.array_equal(h["keys"], o_all.keys[s0:s1]), ("exchange keys", b)
        assert not wide or np.array_equal(h["keys_lo"], o_all.keys_lo[s0:s1]), ("exchange second words", b)
        assert np.array_equal(h["val_off"], o_all.val_off[s0:s1 + 1] - o_all.val_off[s0])
        assert np.array_equal(h["val_a"], o_all.val_a[v0:v1]) and np.array_equal(h["val_b"], o_all.val_b[v0:v1])
        if hi > lo and s1 > s0:
            assert int(h["val_b"].max()) >= 2          # genome 0's k-mers were counted on every rank
        cases += 1
        got.close()
    # read databases: the sets of (read, position) unite (every rank numbers its reads from 0: the values interleave)
    out = C.c_void_p()
    a = bounds_arr(layouts[0])
    ctx.check(lib.kmg_db_exchange_by_key_range(comm, rdb.h, a.ctypes.data_as(C.c_void_p), C.byref(out)))
    got = km.KmerDatabase(ctx, out)
    h = got.fetch()
    pairs = {}
    lo, hi = layouts[0][rank], layouts[0][rank + 1]
    for r in range(nranks):
        o_r = OracleDB(synth.fastq_bytes(synth.random_reads(302 + r, genomes, 3000, 100, 0.01 if not wide else 0.004, n_rate=0.002)), 1, k, l)
        s0, s1 = int(o_r.bucket_off[lo]), int(o_r.bucket_off[hi])
        for i in range(s0, s1):
            pairs.setdefault((int(o_r.keys[i]), int(o_r.keys_lo[i]) if wide else 0), []).extend(
                zip(o_r.val_a[int(o_r.val_off[i]):int(o_r.val_off[i + 1])].tolist(), o_r.val_b[int(o_r.val_off[i]):int(o_r.val_off[i + 1])].tolist()))
    want_keys = np.array([p[0] for p in sorted(pairs)], np.uint64)
    assert np.array_equal(h["keys"], want_keys)
    assert not wide or np.array_equal(h["keys_lo"], np.array([p[1] for p in sorted(pairs)], np.uint64))
    want_vals = [p for key in sorted(pairs) for p in sorted(pairs[key])]
    assert h["val_a"].tolist() == [p[0] for p in want_vals] and h["val_b"].tolist() == [p[1] for p in want_vals]
    cases += 1
    got.close()

    stats = (C.c_uint64 * 4)()
    if double is not None:
        double.kmg_transport_double_stats(stats)
    lib.kmg_comm_destroy(comm)
    hip.hipFree(d_hit)
    for d in (local, rdb):
        d.close()
    ctx.close()
    with open(os.path.join(workdir, "rank%d.json" % rank), "w") as f:
        json.dump({"ok": True, "cases": cases, "sends": int(stats[0]), "receives": int(stats[1]), "allgathers": int(stats[2]),
                   "rounds": int(stats[3])}, f)


if __name__ == "__main__":
    main()
