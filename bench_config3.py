"""bench.py --config 3: BASELINE config 3 at full size on ONE MI355X (secondary line, not the headline).

  reference : 10 000 taxa x 100 kbp = 1 Gbp, 20 % of every genome copied from its sibling taxon (multi-taxon k-mers,
              unique < match), k = 31, key length 6, both strands: 2 * 10^9 records in one build
  reads     : 10^8 x 150 bp, 1 % substitutions, streamed as 10 chunks of 10 M reads; every chunk is built into its
              own read database (the reference's file-set convention, reads.db.1 .. .N, Matcher.java:111-128) and joined
              against the RESIDENT reference database
  TaxaCounts: kmg_taxacounts over the built reference (10 000 taxa: the 16 384-row LDS histogram path)

Everything is generated on the device (seeded torch generators; a 1 Gbp FASTA never exists as a file) and is resident
in HBM before anything is timed.  Correctness at this size, in the run itself:
  * size-independent properties of the reference database: strictly ascending k-mers, sum of counts = 2 * windows,
    closure under reverse complement (two hash sums over {k-mer} and over {k-mer x number of taxa} equal those over the
    reverse complements), TaxaCounts rows: sum(match) = V, unique <= match, sum(unique) = k-mers with one taxon;
  * equality with the CPU oracle on a 1 % slice: key buckets [2000, 2041) of 4096 of the reference database -- every
    k-mer, taxon and count -- and, for the first 100 000 reads, every hit of the read k-mers in those buckets.
"""
import ctypes as C
import json
import os
import time

K, KEY_LEN = 31, 6
GLEN, SHARED, RLEN = 100_000, 20_000, 150
SLICE = (2000, 2041)          # 41 of 4096 key buckets = 1 %
HBM_PEAK_GBS = 8000.0


def _ascii_from_codes(torch, codes):
    """uint8 codes 0..3 -> ASCII A C G T, in place, without a 64-bit index tensor."""
    step = 1 << 27
    for a in range(0, codes.numel(), step):
        c = codes[a:a + step]
        c.copy_(65 + 2 * (c == 1).to(torch.uint8) + 6 * (c == 2).to(torch.uint8) + 19 * (c == 3).to(torch.uint8))
    return codes


def make_reference(torch, np, dev, n_taxa):
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    bases = _ascii_from_codes(torch, torch.randint(0, 4, (n_taxa * GLEN,), dtype=torch.uint8, device=dev, generator=g))
    rng = np.random.Generator(np.random.PCG64(3))
    dst = rng.integers(0, GLEN - SHARED + 1, size=n_taxa)
    src = rng.integers(0, GLEN - SHARED + 1, size=n_taxa)
    for i in range(1, n_taxa):          # genome i takes 20 % from genome i - 1 (as that one is by now)
        a, b = i * GLEN + int(dst[i]), (i - 1) * GLEN + int(src[i])
        bases[a:a + SHARED] = bases[b:b + SHARED]
    taxa = np.arange(n_taxa, dtype=np.int32) + 100_001
    return bases, taxa


def make_reads(torch, dev, bases, n_taxa, n_reads, seed, out):
    """n_reads x 150 bp sampled from a random taxon / offset / strand with 1 % substitutions, written into `out`."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(RLEN, device=dev)
    step = 1_000_000
    for lo in range(0, n_reads, step):
        n = min(step, n_reads - lo)
        start = torch.randint(0, n_taxa, (n,), device=dev, generator=g) * GLEN + torch.randint(0, GLEN - RLEN + 1, (n,), device=dev, generator=g)
        strand = torch.randint(0, 2, (n,), device=dev, generator=g).bool()
        j = torch.where(strand[:, None], RLEN - 1 - ar[None, :], ar[None, :])
        win = bases[start[:, None] + j]
        win = torch.where(strand[:, None], comp[win.long()], win)
        err = torch.rand((n, RLEN), device=dev, generator=g) < 0.01
        sub = acgt[torch.randint(0, 4, (n, RLEN), device=dev, generator=g)]
        out[lo * RLEN:(lo + n) * RLEN] = torch.where(err, sub, win).reshape(-1)


def _rc_keys(torch, x):
    """Kmer.inplaceRC on left-aligned 2-bit k-mers held as int64 bits: reverse the 2-bit groups, complement = code ^ 2."""
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
        x = ((x >> sh) & m) | ((x & m) << sh)
    x = ((x >> 32) & 0xFFFFFFFF) | (x << 32)
    x = x << (64 - 2 * K)
    return x ^ (-0x5555555555555556 & -(1 << (64 - 2 * K)))      # 0xAAAA.. on the k-mer's bits


def _hash(torch, x):
    y = x * -0x61C8864680B583EB          # 0x9E3779B97F4A7C15 as int64
    return y ^ ((y >> 31) & 0x1FFFFFFFF)


def _pmc_traffic(kernel_id):
    """HBM bytes per launch of the dominant kernel from profiles/*cfg3*_pmc_traffic.json (tools/pmc_traffic.py <tag>
    --config 3 --no-cpu-baseline), if that file was measured on the sources this library was built from."""
    import glob
    from bench import KERNEL_SYMBOL, ROOT, source_hash
    sym = KERNEL_SYMBOL.get(kernel_id)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*cfg3*_pmc_traffic.json")), key=os.path.getmtime)
    if not sym or not files:
        return None, "no measurement"
    data = json.load(open(files[-1]))
    if data.get("source_hash") != source_hash():
        return None, "%s was measured on other kernel sources" % os.path.basename(files[-1])
    syms = [sym] + (["ls_sortidx_kernel<unsigned int, false"] if kernel_id == "sort_local_ref" else [])
    hits = [int(v["traffic_bytes_per_launch"]) for name, v in data["kernels"].items() if any(s_ in name for s_ in syms)]
    if not hits:
        return None, "no measurement"
    # (the reference build's launch is the larger one at this config: 2 G records against 1.2 G per read chunk)
    return (max(hits) if kernel_id.endswith("_ref") else min(hits)), os.path.basename(files[-1])


def run(args, dev):
    import numpy as np
    import torch
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB

    scale = args.scale
    n_taxa = max(20, int(round(10_000 * scale)))
    # KMERS_CFG3_KERNELS_ONLY=1: one reference build and one chunk at full size, nothing checked or timed -- the run
    # tools/pmc_traffic.py counts HBM bytes over (counter collection serialises every launch: the whole job would take
    # the better part of an hour under it)
    kernels_only = os.environ.get("KMERS_CFG3_KERNELS_ONLY") == "1"
    n_chunks = 1 if kernels_only else 10
    chunk_reads = max(1000, int(round(10_000_000 * scale)))
    log = lambda *a: print("[cfg3 %6.1fs]" % (time.perf_counter() - t_start), *a, flush=True)
    t_start = time.perf_counter()

    bases, taxa = make_reference(torch, np, dev, n_taxa)
    d_off = (torch.arange(n_taxa + 1, dtype=torch.int64, device=dev) * GLEN)
    d_id = torch.from_numpy(taxa).to(dev)
    n_ref_win = n_taxa * (GLEN - K + 1)
    torch.cuda.synchronize()
    log("reference input on the device: %d taxa, %.3f Gbp" % (n_taxa, bases.numel() / 1e9))

    ctx_ref, ctx = km.Context(dev.index), km.Context(dev.index)
    lib = ctx.lib

    def build(c, d_bases, d_seq_off, ids, toks, nseq, kind, rc):
        c.check(lib.kmg_build_begin(c.h, K, KEY_LEN, rc, kind))
        c.check(lib.kmg_build_add_dev(c.h, d_bases.data_ptr(), d_bases.numel(), d_seq_off.data_ptr(),
                                      ids.data_ptr() if ids is not None else None,
                                      toks.data_ptr() if toks is not None else None, nseq, None))
        db, nw = C.c_void_p(), C.c_uint64()
        c.check(lib.kmg_build_finish(c.h, C.byref(db), C.byref(nw)))
        return db, nw.value

    # ---- reference build: once to warm the pool (hipMalloc of ~100 GB), then timed, then once more under the hook
    ref_ms = []
    n_rep = 1 if kernels_only else 3
    for rep in range(n_rep):
        if rep == 2:
            ctx_ref.prof_enable(True)
            ctx_ref.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h, nw = build(ctx_ref, bases, d_off, d_id, None, n_taxa, 0, 1)
        torch.cuda.synchronize()
        ref_ms.append((time.perf_counter() - t0) * 1e3)
        assert nw == n_ref_win, (nw, n_ref_win)
        if rep < n_rep - 1:
            lib.kmg_db_free(h)
    prof_ref = {k_: v for k_, v in ctx_ref.prof().items() if v["launches"]}
    ctx_ref.prof_enable(False)
    ref = km.KmerDatabase(ctx_ref, h)
    t_ref = ref_ms[min(1, len(ref_ms) - 1)] * 1e-3
    log("reference built: %d distinct k-mers, %d (k-mer, taxon) pairs; %s ms" % (ref.n_unique, ref.n_values, " / ".join("%.1f" % m for m in ref_ms)))

    # ---- reads: 10 chunks against the resident reference
    d_reads = torch.empty(chunk_reads * RLEN, dtype=torch.uint8, device=dev)
    d_roff = torch.arange(chunk_reads + 1, dtype=torch.int64, device=dev) * RLEN
    n_chunk_win = chunk_reads * (RLEN - K + 1)
    d_hit = torch.empty(n_chunk_win + 16, dtype=torch.int64, device=dev)
    if os.environ.get("KMERS_CFG3_DEBUG"):
        d_hit.fill_(-5)
    t_chunks, u_total, hits_total, match_ms = [], 0, 0, []
    first_reads = None
    prof_reads = None
    for ci in range(1 if kernels_only else 0, n_chunks + 1):     # chunk 0 is run twice: the first pass warms the pool
        chunk = max(ci - 1, 0)
        if ci <= 1:
            make_reads(torch, dev, bases, n_taxa, chunk_reads, 4, d_reads)
            if first_reads is None:
                first_reads = d_reads[:min(100_000, chunk_reads) * RLEN].cpu().numpy().reshape(-1, RLEN).copy()
        else:
            make_reads(torch, dev, bases, n_taxa, chunk_reads, 4 + chunk, d_reads)
        d_tok = torch.arange(chunk_reads, dtype=torch.int64, device=dev) + chunk * chunk_reads
        if ci == n_chunks:
            ctx.prof_enable(True)
            ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rdb, nw = build(ctx, d_reads, d_roff, None, d_tok, chunk_reads, 1, 0)
        t1 = time.perf_counter()
        ctx.check(lib.kmg_match_dev(ctx.h, rdb, ref.h, d_hit.data_ptr()))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert nw == n_chunk_win
        u = C.c_uint64()
        lib.kmg_db_info(rdb, None, None, None, None, C.byref(u), None)
        hits = int((d_hit[:u.value] >= 0).sum().item())
        if os.environ.get("KMERS_CFG3_DEBUG"):
            log("debug: min %d max %d, entries >= U_ref %d" % (int(d_hit[:u.value].min().item()), int(d_hit[:u.value].max().item()),
                                                            int((d_hit[:u.value] >= ref.n_unique).sum().item())))
        lib.kmg_db_free(rdb)
        if ci == n_chunks:
            prof_reads = {k_: v for k_, v in ctx.prof().items() if v["launches"]}
            ctx.prof_enable(False)
        if ci >= 1:
            t_chunks.append(t2 - t0)
            match_ms.append((t2 - t1) * 1e3)
            u_total += u.value
            hits_total += hits
        log("chunk %d%s: %d reads, %d distinct k-mers, %d hits, build %.1f ms + join %.1f ms" %
            (chunk, " (warm-up)" if ci == 0 else "", chunk_reads, u.value, hits, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    t_reads = sum(t_chunks)
    n_read_win = n_chunk_win * n_chunks
    if kernels_only:
        return {"kernels_only": True, "ref_ms": ref_ms, "chunk_ms": [t * 1e3 for t in t_chunks]}

    # ---- TaxaCounts
    tc_ms = []
    for rep in range(4):
        t0 = time.perf_counter()
        t_ids, t_match, t_unique = km.taxa_counts(ref)
        tc_ms.append((time.perf_counter() - t0) * 1e3)
    tc = sum(tc_ms[1:]) / 3 * 1e-3
    log("TaxaCounts: %d taxa, %.1f ms" % (len(t_ids), tc * 1e3))

    # ---- properties of the reference database at this size
    U, V = ref.n_unique, ref.n_values
    checks = {}
    keys = torch.empty(U, dtype=torch.int64, device=dev)
    ctx_ref.check(lib.kmg_db_copy_dev(ref.h, 0, keys.data_ptr(), 0, U))
    voff = torch.empty(U + 1, dtype=torch.int64, device=dev)
    ctx_ref.check(lib.kmg_db_copy_dev(ref.h, 1, voff.data_ptr(), 0, U + 1))
    step = 1 << 27
    asc, s_k, s_rc, w_k, w_rc, single = True, 0, 0, 0, 0, 0
    for a in range(0, U, step):
        x = keys[a:a + step]
        nv = voff[a + 1:a + step + 1] - voff[a:a + x.numel()]
        if a > 0:
            asc = asc and bool(((keys[a - 1] ^ -(1 << 63)) < (x[0] ^ -(1 << 63))).item())
        sx = x ^ -(1 << 63)                         # unsigned order as signed order
        asc = asc and bool((sx[1:] > sx[:-1]).all().item())
        hk, hr = _hash(torch, x), _hash(torch, _rc_keys(torch, x))
        s_k += int(hk.sum().item()); s_rc += int(hr.sum().item())
        w_k += int((hk * nv).sum().item()); w_rc += int((hr * nv).sum().item())
        single += int((nv == 1).sum().item())
    M = (1 << 64) - 1
    checks["keys_strictly_ascending"] = asc
    checks["closed_under_reverse_complement"] = (s_k & M) == (s_rc & M) and (w_k & M) == (w_rc & M)
    del keys
    vb = torch.empty(V, dtype=torch.int32, device=dev)
    ctx_ref.check(lib.kmg_db_copy_dev(ref.h, 3, vb.data_ptr(), 0, V))
    checks["sum_of_counts_is_twice_the_windows"] = int(vb.sum(dtype=torch.int64).item()) == 2 * n_ref_win
    del vb
    checks["taxacounts_sum_match_is_V"] = int(t_match.sum()) == V
    checks["taxacounts_unique_le_match"] = bool((t_unique <= t_match).all()) and bool((t_unique < t_match).any())
    checks["taxacounts_sum_unique_is_single_taxon_kmers"] = int(t_unique.sum()) == single
    checks["taxacounts_rows"] = len(t_ids) == n_taxa and bool((np.diff(t_ids) > 0).all())
    log("properties:", checks)
    assert all(checks.values()), checks

    # ---- the CPU leg (cpu_baseline): the oracle on a bounded sample, timed; its output checks the 1 % slice
    t0 = time.perf_counter()
    host = bases.cpu().numpy().reshape(n_taxa, GLEN)
    fa = b"".join(b">%d\n%s\n" % (int(t), host[i].tobytes()) for i, t in enumerate(taxa))
    threads = min(16, os.cpu_count() or 1)
    o_ref = OracleDB(fa, 0, K, KEY_LEN, threads=threads, key_range=SLICE)
    del fa
    boff = ref.bucket_offsets()
    u0, u1 = int(boff[SLICE[0]]), int(boff[SLICE[1]])
    v0, v1 = int(voff[u0].item()), int(voff[u1].item())

    def part(which, first, count, dt):
        t = torch.empty(count, dtype=dt, device=dev)
        ctx_ref.check(lib.kmg_db_copy_dev(ref.h, which, t.data_ptr(), first, count))
        return t.cpu().numpy()
    same = (o_ref.U == u1 - u0 and o_ref.V == v1 - v0
            and np.array_equal(part(0, u0, u1 - u0, torch.int64).view(np.uint64), o_ref.keys)
            and np.array_equal(part(1, u0, u1 - u0 + 1, torch.int64) - v0, o_ref.val_off.astype(np.int64))
            and np.array_equal(part(2, v0, v1 - v0, torch.int32), o_ref.val_a)
            and np.array_equal(part(3, v0, v1 - v0, torch.int32), o_ref.val_b))
    checks["oracle_slice_reference"] = bool(same)
    # reads: the first 100 000 reads' k-mers of those buckets, hit for hit
    n_first = first_reads.shape[0]
    o_reads = OracleDB(synth.fastq_bytes(first_reads), 1, K, KEY_LEN, threads=threads, key_range=SLICE)
    o_hit, o_nh = o_reads.match(o_ref)
    d_first = torch.from_numpy(first_reads.reshape(-1)).to(dev)
    torch.cuda.synchronize()
    rdb, _ = build(ctx, d_first, d_roff[:n_first + 1].contiguous(), None, torch.arange(n_first, dtype=torch.int64, device=dev), n_first, 1, 0)
    ctx.check(lib.kmg_match_dev(ctx.h, rdb, ref.h, d_hit.data_ptr()))
    torch.cuda.synchronize()
    r = km.KmerDatabase(ctx, rdb)
    rb = r.bucket_offsets()
    s0, s1 = int(rb[SLICE[0]]), int(rb[SLICE[1]])
    got = d_hit[s0:s1].cpu().numpy()
    rk = torch.empty(s1 - s0, dtype=torch.int64, device=dev)
    ctx.check(lib.kmg_db_copy_dev(r.h, 0, rk.data_ptr(), s0, s1 - s0))
    checks["oracle_slice_read_hits"] = bool(o_reads.U == s1 - s0 and np.array_equal(rk.cpu().numpy().view(np.uint64), o_reads.keys)
                                            and np.array_equal(np.where(got >= 0, got - u0, -1), o_hit) and o_nh > 0)
    r.close()
    oracle_s = time.perf_counter() - t0
    log("oracle slice: reference %s, read hits %s (%d hits in the slice), %.1f s" %
        (checks["oracle_slice_reference"], checks["oracle_slice_read_hits"], o_nh, oracle_s))
    assert checks["oracle_slice_reference"] and checks["oracle_slice_read_hits"], checks

    # ---- roofline of the dominant kernel at this shape
    prof = {}
    for name, v in prof_ref.items():
        prof[name if name.endswith("_ref") else name + "_ref"] = v
    prof.update(prof_reads)
    cand = {k_: v for k_, v in prof.items() if v["bytes"]}
    dom = max(cand, key=lambda k_: cand[k_]["ms"])
    d = cand[dom]
    avg_ms, bpl = d["ms"] / d["launches"], d["bytes"] / d["launches"]
    achieved = bpl / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = _pmc_traffic(dom)
    kernels = {k_: {"ms": round(v["ms"], 3), "launches": v["launches"],
                    "algo_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None} for k_, v in prof.items()}
    total_s = t_ref + t_reads
    out = {
        "metric": "k-mers/sec built + matched (reference build + read build + match, node)",
        "value": (n_ref_win + n_read_win) / total_s, "unit": "k-mers/s",
        "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": total_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BASELINE config 3 on one GPU: %d taxa x 100 kbp reference (%.2f Gbp, 20 %% sibling copies), %d x 150 bp "
                               "reads in %d chunks of %d against the resident reference, k=31, key length 6, then TaxaCounts"
                               % (n_taxa, n_taxa * GLEN / 1e9, chunk_reads * n_chunks, n_chunks, chunk_reads),
                   "scale": scale, "ref_windows": n_ref_win, "read_windows": n_read_win,
                   "ref_distinct_kmers": U, "ref_values": V, "read_distinct_kmers": u_total, "hits": hits_total},
        "stages": {"ref_build_ms": round(ref_ms[1], 2), "ref_kmers_built_per_s": n_ref_win / t_ref,
                   "read_chunk_ms_mean": round(t_reads / n_chunks * 1e3, 2),
                   "read_kmers_built_and_matched_per_s": n_read_win / t_reads,
                   "join_ms_mean": round(sum(match_ms) / len(match_ms), 3),
                   "distinct_read_kmers_matched_per_s": u_total / (sum(match_ms) * 1e-3),
                   "taxacounts_ms": round(tc * 1e3, 2), "taxacounts_kmers_per_s": U / tc, "taxacounts_values_per_s": V / tc,
                   "taxacounts_algo_GBps": round((8 * (U + 1) + 4 * V) / tc / 1e9, 1),
                   "timing": "wall clock around synchronised stages; pool warm (second build / second pass of chunk 0)"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": int(bpl), "launches": d["launches"]},
        "kernels": kernels,
        "verified": dict(checks, oracle_slice="key buckets [%d, %d) of 4096; first %d reads" % (SLICE[0], SLICE[1], n_first),
                         oracle_seconds=round(oracle_s, 1)),
        # the CPU leg of this line: the C oracle (`port`), timed on a bounded sample of the same workload -- it reads and
        # extracts the WHOLE reference (both strands) but sorts, reduces and joins only the k-mers of 1 % of the key
        # buckets, plus the first reads -- and its output is what the slice above was compared with
        "cpu_baseline": {"value": (n_ref_win + n_first * (RLEN - K + 1)) / oracle_s, "unit": "k-mers/s (windows extracted; 1 % of them sorted and joined)",
                         "cores": threads, "kind": "port",
                         "sample": "whole %.2f Gbp reference extracted on both strands, key buckets [%d, %d) of 4096 built and joined, "
                                   "first %d reads: %.1f s" % (n_taxa * GLEN / 1e9, SLICE[0], SLICE[1], n_first, oracle_s)},
    }
    ref.close()
    ctx.close()
    ctx_ref.close()
    return out
