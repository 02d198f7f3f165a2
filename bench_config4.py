"""bench.py --config 4: ONE rank's share of BASELINE config 4 on ONE MI355X (secondary line, not the headline).

Config 4 = 8 x MI355X: a ~30 Gbp reference (300 000 pseudo-genomes x 100 kbp) range-sharded by k-mer prefix, 10^9 reads
(1.25 * 10^8 per GPU), the distinct read k-mers exchanged to their owners with an RCCL all-to-all.  Eight GPUs are
the driver's to run; what one GPU of that job does is measurable on one GPU, and that is this line:

  reference : the WHOLE 30 Gbp input (resident in HBM, generated on the device) scanned with kmg_build_set_key_range
              for the 512 of 4096 key buckets rank r owns (r = 3 by default) -- both strands, ~7.5 * 10^9 records.
              One build holds < 2^32-16 records and needs both sort buffers beside the shard already built, so the
              range is built in P key-range passes (P = 8 at full size; the record limit alone would ask for 2), each a
              scan of the whole input in batches of 10 000 genomes; the passes' databases are disjoint ascending
              pieces of the shard (the reference's file-set form, name.1 .. name.N, Matcher.java:80-89) and stay
              resident (24 bytes per distinct k-mer: ~180 GB for the shard).
  reads     : the rank's own 1.25 * 10^8 x 150 bp reads (1 % substitutions) in 25 chunks of 5 M, each built into a read
              database; the chunk's distinct k-mers inside the rank's range (a slice: the database is sorted) are joined
              against the shard's pieces (kmg_match_keys_dev, one ascending run per piece); the k-mers outside it are
              what this rank would send to the seven other owners -- counted, with the hit indices that come back,
              as exchange bytes against the xGMI bound (7 links x 153 GB/s), not moved: one GPU has no peer.

Correctness at this size, in the run itself:
  * every piece's k-mers strictly ascending, inside its pass's buckets, pieces ascending end to end;
  * the multiset of records, independently: the k-mers of every window of the input (and their reverse complements)
    are recomputed with torch integer ops, those whose bucket lies in the rank's range counted and hashed -- the count
    equals the sum of all counts in the shard, the hash sum equals the sum over the shard of hash(k-mer) x count;
  * equality with the CPU oracle on a slice: the genomes the first 5 000 reads were drawn from (~0.5 Gbp) built by the
    oracle for 41 key buckets of the rank's range -- every (k-mer, taxon, count) of those taxa in those buckets of the
    shard equals the oracle's, and none is missing or extra --, and the first 5 000 reads' k-mers in those buckets
    joined: a read k-mer hits a k-mer of those taxa in the shard exactly where the oracle's join hits.
"""
import ctypes as C
import json
import os
import time

import bench_config3 as b3

K, KEY_LEN = 31, 6
GLEN, RLEN = b3.GLEN, b3.RLEN
NODE_GPUS = 8
HBM_PEAK_GBS = 8000.0
XGMI_GBS = 7 * 153.0
BATCH_TAXA = 10_000                 # genomes per kmg_build_add_dev (1 Gbp)
N_SAMPLE_READS = 5_000


def _kmers_of_slice(torch, codes):
    """codes: int64 base codes (A0 C1 T2 G3) of positions p .. p+n+29 -> left-aligned k-mers of the n windows (K = 31),
    by doubling: 2-, 4-, 8-, 16-mers, then 16 + 8 + 4 + 2 + 1."""
    n = codes.numel() - (K - 1)
    v1 = codes
    v2 = (v1[:-1] << 2) | v1[1:]
    v4 = (v2[:-2] << 4) | v2[2:]
    v8 = (v4[:-4] << 8) | v4[4:]
    v16 = (v8[:-8] << 16) | v8[8:]
    key = (v16[:n] << 30) | (v8[16:16 + n] << 14) | (v4[24:24 + n] << 6) | (v2[28:28 + n] << 2) | v1[30:30 + n]
    return key << 2


def _independent_records(torch, dev, bases, lo_bucket, hi_bucket):
    """Count and hash, with torch integer ops only, every record of the rank's range: forward k-mers of all windows
    inside a genome, and their reverse complements, whose Kmer.key(6) lies in [lo_bucket, hi_bucket)."""
    n = bases.numel()
    step = 1 << 26
    cnt, hsum = 0, 0
    for a in range(0, n - (K - 1), step):
        m = min(step, n - (K - 1) - a)
        codes = ((bases[a:a + m + K - 1] >> 1) & 3).to(torch.int64)
        key = _kmers_of_slice(torch, codes)
        del codes
        p = torch.arange(a, a + m, device=dev, dtype=torch.int64) % GLEN
        inside = p <= GLEN - K
        del p
        for kk in (key, b3._rc_keys(torch, key)):
            bucket = (kk >> (64 - 2 * KEY_LEN)) & (4 ** KEY_LEN - 1)
            keep = inside & (bucket >= lo_bucket) & (bucket < hi_bucket)
            cnt += int(keep.sum().item())
            hsum += int((b3._hash(torch, kk) * keep).sum().item())
            del bucket, keep
        del key, inside
    return cnt, hsum & ((1 << 64) - 1)


def run(args, dev):
    import numpy as np
    import torch
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB

    scale = args.scale
    rank = int(os.environ.get("KMERS_CFG4_RANK", "3"))
    n_taxa = max(40, int(round(300_000 * scale)))
    n_chunks = 25
    chunk_reads = max(N_SAMPLE_READS, int(round(5_000_000 * scale)))
    nkeys = 4 ** KEY_LEN
    r_lo, r_hi = nkeys * rank // NODE_GPUS, nkeys * (rank + 1) // NODE_GPUS
    log = lambda *a: print("[cfg4 %6.1fs]" % (time.perf_counter() - t_start), *a, flush=True)
    t_start = time.perf_counter()

    # ---- the whole reference input, on the device
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    bases = torch.empty(n_taxa * GLEN, dtype=torch.uint8, device=dev)
    for a in range(0, bases.numel(), 1 << 30):
        bases[a:a + (1 << 30)].random_(0, 4, generator=g)
    b3._ascii_from_codes(torch, bases)
    taxa = torch.arange(n_taxa, dtype=torch.int32, device=dev) + 100_001
    bt = min(BATCH_TAXA, n_taxa)
    d_off = torch.arange(bt + 1, dtype=torch.int64, device=dev) * GLEN
    torch.cuda.synchronize()
    n_ref_win = n_taxa * (GLEN - K + 1)
    expect_records = 2 * n_ref_win // NODE_GPUS
    log("reference input on the device: %d genomes, %.2f Gbp; rank %d of %d owns key buckets [%d, %d): ~%.2f G records"
        % (n_taxa, bases.numel() / 1e9, rank, NODE_GPUS, r_lo, r_hi, expect_records / 1e9))

    ctx = km.Context(dev.index)
    lib = ctx.lib
    # passes: the record limit of one build asks for min_passes; memory (both sort buffers beside the resident shard)
    # for records-per-pass of about 10^9
    min_passes = -(-expect_records // (2 ** 32 - 16))
    passes = int(os.environ.get("KMERS_CFG4_PASSES", "0"))
    if passes <= 0:
        passes = 1
        while expect_records / passes > 1.0e9 and passes < r_hi - r_lo:
            passes *= 2
    bounds = [r_lo + (r_hi - r_lo) * p // passes for p in range(passes + 1)]

    def build_range(first, end):
        ctx.check(lib.kmg_build_begin(ctx.h, K, KEY_LEN, 1, 0))
        ctx.check(lib.kmg_build_set_key_range(ctx.h, first, end))
        for t0_ in range(0, n_taxa, bt):
            n = min(bt, n_taxa - t0_)
            ctx.check(lib.kmg_build_add_dev(ctx.h, bases[t0_ * GLEN:].data_ptr(), n * GLEN, d_off.data_ptr(),
                                            taxa[t0_:].data_ptr(), None, n, None))
        db, nw = C.c_void_p(), C.c_uint64()
        ctx.check(lib.kmg_build_finish(ctx.h, C.byref(db), C.byref(nw)))
        return db, nw.value

    # ---- the reference shard, pass by pass (pass 0 once before, untimed: it grows the pool)
    h, _ = build_range(bounds[0], bounds[1])
    lib.kmg_db_free(h)
    torch.cuda.synchronize()
    pieces, pass_ms, kept_fwd = [], [], 0
    prof_ref = None
    for p in range(passes):
        if p == passes - 1:
            ctx.prof_enable(True)
            ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h, nw = build_range(bounds[p], bounds[p + 1])
        torch.cuda.synchronize()
        pass_ms.append((time.perf_counter() - t0) * 1e3)
        if p == passes - 1:
            prof_ref = {k_: v for k_, v in ctx.prof().items() if v["launches"]}
            ctx.prof_enable(False)
        db = km.KmerDatabase(ctx, h)
        pieces.append(db)
        kept_fwd += nw
        log("pass %d/%d: key buckets [%d, %d): %d distinct k-mers, %d (k-mer, taxon) pairs, %.1f ms"
            % (p + 1, passes, bounds[p], bounds[p + 1], db.n_unique, db.n_values, pass_ms[-1]))
    t_ref = sum(pass_ms) * 1e-3
    U = sum(d.n_unique for d in pieces)
    V = sum(d.n_values for d in pieces)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    log("shard built: %d distinct k-mers in %d pieces, %.1f s; device memory in use %.1f of %.1f GB"
        % (U, passes, t_ref, (total_b - free_b) / 1e9, total_b / 1e9))

    # ---- reads: this rank's chunks, the k-mers of its range joined against the shard's pieces
    d_reads = torch.empty(chunk_reads * RLEN, dtype=torch.uint8, device=dev)
    d_roff = torch.arange(chunk_reads + 1, dtype=torch.int64, device=dev) * RLEN
    n_chunk_win = chunk_reads * (RLEN - K + 1)
    max_in_range = n_chunk_win // 4 + 1024          # (an eighth is expected)
    d_keys = torch.empty(max_in_range, dtype=torch.int64, device=dev)
    d_hit = torch.empty(max_in_range, dtype=torch.int64, device=dev)
    t_chunks, join_ms, u_total, in_range_total, hits_total = [], [], 0, 0, 0
    first_reads, first_src = None, None
    prof_reads = None

    def build_reads(d_bases, d_seq_off, toks, nseq):
        ctx.check(lib.kmg_build_begin(ctx.h, K, KEY_LEN, 0, 1))
        ctx.check(lib.kmg_build_add_dev(ctx.h, d_bases.data_ptr(), d_bases.numel(), d_seq_off.data_ptr(), None, toks.data_ptr(), nseq, None))
        db, nw = C.c_void_p(), C.c_uint64()
        ctx.check(lib.kmg_build_finish(ctx.h, C.byref(db), C.byref(nw)))
        return km.KmerDatabase(ctx, db), nw.value

    def join_in_range(rdb):
        """-> (search k-mers in the rank's range, hits); their keys in d_keys, hit indices (into the piece) in d_hit."""
        rb = rdb.bucket_offsets()
        s0, s1 = int(rb[r_lo]), int(rb[r_hi])
        assert s1 - s0 <= max_in_range
        if s1 > s0:
            ctx.check(lib.kmg_db_copy_dev(rdb.h, 0, d_keys.data_ptr(), s0, s1 - s0))
        cuts = [int(rb[b]) - s0 for b in bounds]
        for p, piece in enumerate(pieces):
            n = cuts[p + 1] - cuts[p]
            if n:
                run_off = (C.c_uint64 * 2)(0, n)
                ctx.check(lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr() + 8 * cuts[p], run_off, 1, K, piece.h,
                                                 d_hit.data_ptr() + 8 * cuts[p]))
        return s1 - s0, cuts

    for ci in range(n_chunks + 1):                      # chunk 0 twice: the first pass warms the pool
        chunk = max(ci - 1, 0)
        if ci != 1:
            b3.make_reads(torch, dev, bases, n_taxa, chunk_reads, 6 + chunk, d_reads)
        if first_reads is None:
            first_reads = d_reads[:N_SAMPLE_READS * RLEN].cpu().numpy().reshape(-1, RLEN).copy()
        d_tok = torch.arange(chunk_reads, dtype=torch.int64, device=dev) + chunk * chunk_reads
        if ci == n_chunks:
            ctx.prof_enable(True)
            ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rdb, nw = build_reads(d_reads, d_roff, d_tok, chunk_reads)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_in, _ = join_in_range(rdb)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert nw == n_chunk_win
        hits = int((d_hit[:n_in] >= 0).sum().item())
        u = rdb.n_unique
        rdb.close()
        if ci == n_chunks:
            prof_reads = {k_: v for k_, v in ctx.prof().items() if v["launches"]}
            ctx.prof_enable(False)
        if ci >= 1:
            t_chunks.append(t2 - t0)
            join_ms.append((t2 - t1) * 1e3)
            u_total += u
            in_range_total += n_in
            hits_total += hits
        if ci <= 2 or ci == n_chunks:
            log("chunk %d%s: %d reads, %d distinct k-mers, %d in the rank's range, %d hits, build %.1f ms + slice and join %.1f ms"
                % (chunk, " (warm-up)" if ci == 0 else "", chunk_reads, u, n_in, hits, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    t_reads = sum(t_chunks)
    n_read_win = n_chunk_win * n_chunks
    sent = u_total - in_range_total                      # distinct k-mers this rank would send to the other owners
    exchange_bytes = 16 * sent                           # 8 B k-mer out, 8 B hit index back
    log("reads: %d chunks, %.1f ms each; %d of %d distinct k-mers leave for other owners" % (n_chunks, t_reads / n_chunks * 1e3, sent, u_total))

    # ---- properties of the shard at this size
    checks = {}
    M = (1 << 64) - 1
    asc, in_bucket, s_hash, s_cnt, last = True, True, 0, 0, None
    step = 1 << 26
    for p, piece in enumerate(pieces):
        for a in range(0, piece.n_unique, step):
            n = min(step, piece.n_unique - a)
            keys = torch.empty(n, dtype=torch.int64, device=dev)
            voff = torch.empty(n + 1, dtype=torch.int64, device=dev)
            ctx.check(lib.kmg_db_copy_dev(piece.h, 0, keys.data_ptr(), a, n))
            ctx.check(lib.kmg_db_copy_dev(piece.h, 1, voff.data_ptr(), a, n + 1))
            v0, v1 = int(voff[0].item()), int(voff[-1].item())
            vb = torch.empty(v1 - v0, dtype=torch.int32, device=dev)
            ctx.check(lib.kmg_db_copy_dev(piece.h, 3, vb.data_ptr(), v0, v1 - v0))
            cs = torch.zeros(v1 - v0 + 1, dtype=torch.int64, device=dev)
            cs[1:] = torch.cumsum(vb, 0, dtype=torch.int64)
            tot = cs[voff[1:] - v0] - cs[voff[:-1] - v0]          # occurrences of every k-mer (all taxa)
            sx = keys ^ -(1 << 63)
            asc = asc and bool((sx[1:] > sx[:-1]).all().item()) and (last is None or last < int(sx[0].item()))
            last = int(sx[-1].item())
            bucket = (keys >> (64 - 2 * KEY_LEN)) & (nkeys - 1)
            in_bucket = in_bucket and bool(((bucket >= bounds[p]) & (bucket < bounds[p + 1])).all().item())
            s_cnt += int(tot.sum().item())
            s_hash += int((b3._hash(torch, keys) * tot).sum().item())
            del keys, voff, vb, cs, tot, sx, bucket
    checks["keys_strictly_ascending_across_pieces"] = asc
    checks["every_kmer_in_its_pass_buckets"] = in_bucket
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    want_cnt, want_hash = _independent_records(torch, dev, bases, r_lo, r_hi)
    checks["sum_of_counts_is_the_records_of_the_range"] = s_cnt == want_cnt
    checks["hash_sum_over_records_equals_independent_recomputation"] = (s_hash & M) == want_hash
    log("properties: %s (records of the range %d; recomputed with torch in %.1f s)" % (checks, want_cnt, time.perf_counter() - t0))
    assert all(checks.values()), checks

    # ---- the CPU leg: the oracle on the genomes the first reads came from, a slice of the rank's key range
    t0 = time.perf_counter()
    slice_lo = r_lo + (r_hi - r_lo) // 3
    SLICE = (slice_lo, min(slice_lo + 41, r_hi))
    gs = torch.Generator(device=dev)
    gs.manual_seed(6)                                    # the draws make_reads made for chunk 0: taxon first
    src = torch.randint(0, n_taxa, (min(1_000_000, chunk_reads),), device=dev, generator=gs)[:N_SAMPLE_READS]
    sub = np.unique(src.cpu().numpy())
    sub_t = torch.from_numpy(sub).to(dev)
    host = bases.view(n_taxa, GLEN)[sub_t].cpu().numpy()
    sub_ids = (sub + 100_001).astype(np.int32)
    fa = b"".join(b">%d\n%s\n" % (int(t), host[i].tobytes()) for i, t in enumerate(sub_ids))
    threads = min(16, os.cpu_count() or 1)
    o_ref = OracleDB(fa, 0, K, KEY_LEN, threads=threads, key_range=SLICE)
    del fa, host
    # the shard's (k-mer, taxon, count) entries of those taxa in those buckets
    p_of = [p for p in range(passes) if bounds[p] < SLICE[1] and bounds[p + 1] > SLICE[0]]
    got_k, got_a, got_b = [], [], []
    member = torch.zeros(n_taxa + 1, dtype=torch.bool, device=dev)
    member[sub_t] = True
    for p in p_of:
        piece = pieces[p]
        pb = piece.bucket_offsets()
        u0, u1 = int(pb[max(SLICE[0], bounds[p])]), int(pb[min(SLICE[1], bounds[p + 1])])
        if u1 == u0:
            continue
        keys = torch.empty(u1 - u0, dtype=torch.int64, device=dev)
        voff = torch.empty(u1 - u0 + 1, dtype=torch.int64, device=dev)
        ctx.check(lib.kmg_db_copy_dev(piece.h, 0, keys.data_ptr(), u0, u1 - u0))
        ctx.check(lib.kmg_db_copy_dev(piece.h, 1, voff.data_ptr(), u0, u1 - u0 + 1))
        v0, v1 = int(voff[0].item()), int(voff[-1].item())
        va = torch.empty(v1 - v0, dtype=torch.int32, device=dev)
        vb = torch.empty(v1 - v0, dtype=torch.int32, device=dev)
        ctx.check(lib.kmg_db_copy_dev(piece.h, 2, va.data_ptr(), v0, v1 - v0))
        ctx.check(lib.kmg_db_copy_dev(piece.h, 3, vb.data_ptr(), v0, v1 - v0))
        key_of_val = torch.repeat_interleave(keys, voff[1:] - voff[:-1])
        sel = member[(va - 100_001).long()]
        got_k.append(key_of_val[sel].cpu().numpy().view(np.uint64))
        got_a.append(va[sel].cpu().numpy())
        got_b.append(vb[sel].cpu().numpy())
        del keys, voff, va, vb, key_of_val, sel
    got_k = np.concatenate(got_k) if got_k else np.zeros(0, np.uint64)
    got_a = np.concatenate(got_a) if got_a else np.zeros(0, np.int32)
    got_b = np.concatenate(got_b) if got_b else np.zeros(0, np.int32)
    o_key_of_val = np.repeat(o_ref.keys, np.diff(o_ref.val_off.astype(np.int64)))
    checks["oracle_slice_reference"] = bool(o_ref.V > 0 and np.array_equal(got_k, o_key_of_val) and np.array_equal(got_a, o_ref.val_a)
                                            and np.array_equal(got_b, o_ref.val_b))
    # the first reads' k-mers of those buckets: hits on k-mers of those taxa, where the oracle's join hits
    o_reads = OracleDB(synth.fastq_bytes(first_reads), 1, K, KEY_LEN, threads=threads, key_range=SLICE)
    o_hit, o_nh = o_reads.match(o_ref)
    d_first = torch.from_numpy(first_reads.reshape(-1)).to(dev)
    rdb, _ = build_reads(d_first, d_roff[:N_SAMPLE_READS + 1].contiguous(), torch.arange(N_SAMPLE_READS, dtype=torch.int64, device=dev), N_SAMPLE_READS)
    n_in, cuts = join_in_range(rdb)
    torch.cuda.synchronize()
    rb = rdb.bucket_offsets()
    s0 = int(rb[r_lo])
    a0, a1 = int(rb[SLICE[0]]) - s0, int(rb[SLICE[1]]) - s0
    got_keys = d_keys[a0:a1].cpu().numpy().view(np.uint64)
    got_mask = np.zeros(a1 - a0, dtype=bool)
    for p in p_of:                                        # a hit counts when the k-mer it names holds a taxon of the subset
        lo, hi = max(a0, cuts[p]), min(a1, cuts[p + 1])
        if hi <= lo:
            continue
        hit = d_hit[lo:hi].cpu().numpy()
        piece = pieces[p]
        ok = np.zeros(hi - lo, dtype=bool)
        vo2 = torch.empty(2, dtype=torch.int64, device=dev)
        for i in np.nonzero(hit >= 0)[0].tolist():
            ctx.check(lib.kmg_db_copy_dev(piece.h, 1, vo2.data_ptr(), int(hit[i]), 2))
            f_, e_ = (int(x) for x in vo2.tolist())
            va = torch.empty(e_ - f_, dtype=torch.int32, device=dev)
            ctx.check(lib.kmg_db_copy_dev(piece.h, 2, va.data_ptr(), f_, e_ - f_))
            ok[i] = bool(np.isin(va.cpu().numpy(), sub_ids).any())
        got_mask[lo - a0:hi - a0] = ok
    checks["oracle_slice_read_hits"] = bool(o_reads.U == a1 - a0 and np.array_equal(got_keys, o_reads.keys)
                                            and np.array_equal(got_mask, o_hit >= 0) and o_nh > 0)
    rdb.close()
    oracle_s = time.perf_counter() - t0
    log("oracle slice: reference %s (%d entries of %d taxa in buckets [%d, %d)), read hits %s (%d hits), %.1f s"
        % (checks["oracle_slice_reference"], o_ref.V, len(sub), SLICE[0], SLICE[1], checks["oracle_slice_read_hits"], o_nh, oracle_s))
    assert checks["oracle_slice_reference"] and checks["oracle_slice_read_hits"], checks

    # ---- the line
    prof = {}
    for name, v in prof_ref.items():
        prof[name if name.endswith("_ref") else name + "_ref"] = v
    prof.update(prof_reads)
    cand = {k_: v for k_, v in prof.items() if v["bytes"]}
    dom = max(cand, key=lambda k_: cand[k_]["ms"])
    d = cand[dom]
    avg_ms, bpl = d["ms"] / d["launches"], d["bytes"] / d["launches"]
    achieved = bpl / (avg_ms * 1e-3) / 1e9
    kernels = {k_: {"ms": round(v["ms"], 3), "launches": v["launches"],
                    "algo_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None} for k_, v in prof.items()}
    total_s = t_ref + t_reads
    n_sub_win = len(sub) * (GLEN - K + 1)
    out = {
        "metric": "k-mers/sec built + matched (reference build + read build + match, node)",
        "value": (kept_fwd + n_read_win) / total_s, "unit": "k-mers/s",
        "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": total_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "one rank's share (rank %d of %d) of BASELINE config 4 on one GPU: %d genomes x 100 kbp (%.2f Gbp) scanned for "
                               "key buckets [%d, %d) in %d key-range passes, %d x 150 bp reads in %d chunks built and their k-mers of "
                               "that range joined against the shard; k=31, key length 6; the exchange itself is not run (one GPU)"
                               % (rank, NODE_GPUS, n_taxa, n_taxa * GLEN / 1e9, r_lo, r_hi, passes, chunk_reads * n_chunks, n_chunks),
                   "scale": scale, "rank": rank, "node_gpus": NODE_GPUS, "ref_windows_in_input": n_ref_win,
                   "ref_forward_windows_kept": kept_fwd, "ref_records_of_the_range": want_cnt, "read_windows": n_read_win,
                   "shard_distinct_kmers": U, "shard_values": V, "read_distinct_kmers": u_total,
                   "read_distinct_kmers_in_range": in_range_total, "hits": hits_total},
        "passes": {"run": passes, "needed_by_the_record_limit": int(min_passes), "records_per_build_limit": 2 ** 32 - 16,
                   "why_more": "both sort buffers of a pass must fit beside the resident input and the shard already built",
                   "ms_each": [round(x, 1) for x in pass_ms], "input_scanned_per_pass_Gbp": round(n_taxa * GLEN / 1e9, 2),
                   "input_bases_scanned_per_s": n_taxa * GLEN * passes / t_ref},
        "stages": {"ref_shard_build_s": round(t_ref, 3), "ref_records_built_per_s": want_cnt / t_ref,
                   "read_chunk_ms_mean": round(t_reads / n_chunks * 1e3, 2), "read_kmers_built_per_s": n_read_win / t_reads,
                   "slice_and_join_ms_mean": round(sum(join_ms) / len(join_ms), 3),
                   "timing": "wall clock around synchronised stages; pool warm (pass 0 and chunk 0 run once before)"},
        "exchange_not_run": {"distinct_kmers_leaving_this_rank": sent, "bytes_out_plus_back": exchange_bytes,
                             "xgmi_bound_ms": round(exchange_bytes / 2 / (XGMI_GBS * 1e9) * 1e3, 2),
                             "note": "8 B per k-mer out and 8 B per hit index back, each direction over 7 links x 153 GB/s; "
                                     "kmg_match_sharded runs it (tests/test_exchange_two_ranks.py), one GPU has no peer"},
        "memory": {"device_total_GB": round(total_b / 1e9, 1), "in_use_after_the_shard_GB": round((total_b - free_b) / 1e9, 1)},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                     "algo_bytes_per_launch": int(bpl), "launches": d["launches"]},
        "kernels": kernels,
        "verified": dict(checks, oracle_slice="key buckets [%d, %d) of 4096; the %d genomes the first %d reads were drawn from"
                                              % (SLICE[0], SLICE[1], len(sub), N_SAMPLE_READS), oracle_seconds=round(oracle_s, 1)),
        "cpu_baseline": {"value": (n_sub_win + N_SAMPLE_READS * (RLEN - K + 1)) / oracle_s,
                         "unit": "k-mers/s (windows extracted; the k-mers of 41 key buckets sorted and joined)",
                         "cores": threads, "kind": "port",
                         "sample": "%d genomes (%.2f Gbp) extracted on both strands, key buckets [%d, %d) built and joined with the first "
                                   "%d reads: %.1f s" % (len(sub), len(sub) * GLEN / 1e9, SLICE[0], SLICE[1], N_SAMPLE_READS, oracle_s)},
    }
    for piece in pieces:
        piece.close()
    ctx.close()
    return out
