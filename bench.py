#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer hot path on MI355X.

One step = BASELINE config 2 on one GPU, inputs already resident in HBM:
    build the reference DB (10 taxa x 500 kbp, both strands)      kmg_build_add_dev + kmg_build_finish
  + build the read DB (1M x 150 bp reads, 1 % substitutions)      kmg_build_add_dev + kmg_build_finish
  + match the read DB against the reference DB                    kmg_match_dev
at k = 31, key length 6.  value = (reference windows + read windows) / step time, whole job.
The two builds are independent and are submitted on two contexts (two HIP streams): the small reference build
runs underneath the reads' extraction; the match starts when both databases are complete.
N > 1: one process per GPU, every rank builds the (small) reference and its own shard of 1M reads
(weak scaling, no data-path collective -- see DESIGN.md "Multi-GPU"); value sums over ranks.

Adds "roofline" (dominant kernel, algorithmic bytes / HIP-event time measured live on the library's
stream) and, on rank 0 at N = 1, "cpu_baseline" (the C restatement of the reference's algorithm timed
on a bounded sample on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
K, KEY_LEN = 31, 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="scale factor on config sizes (1.0 = BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calib", action="store_true", help="also launch the known-traffic copy kernel (PMC calibration)")
    ap.add_argument("--exchange", action="store_true",
                    help="key-range mode (SURVEY 8e) instead of the replicated reference: the reference is range-partitioned by "
                         "k-mer prefix -- every rank builds only the key range it owns (kmg_build_set_key_range) -- and the "
                         "read k-mers are bucketed and exchanged to the owners for the join (RCCL all-to-all, hits come back "
                         "the same way).  Opt-in: not the headline line.")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: RCCL prints a version banner on file descriptor 1 when the
    # communicator comes up, so everything but the result is sent to stderr at the descriptor level
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # KMERS_BENCH_FORCE_DIST=1: run the collective code path with a single rank too (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("KMERS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    import kmers_amd as km
    from kmers_amd import synth

    # ---- synthetic inputs, resident in HBM before anything is timed ------------------------------
    glen = max(1000, int(500_000 * args.scale))
    nreads = max(100, int(1_000_000 * args.scale))
    taxa, genomes = synth.random_genomes(1, 10, glen)
    reads = synth.random_reads(2 + rank, genomes, nreads, 150, 0.01)
    my_genomes, my_taxa = genomes, taxa          # every rank holds the whole reference input
    ref_bases, ref_off = synth.batch_from_seqs(my_genomes)
    read_off = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(150)
    d_ref_bases = torch.from_numpy(ref_bases).to(dev)
    d_ref_off = torch.from_numpy(ref_off.view(np.int64)).to(dev)
    d_ref_id = torch.tensor(my_taxa, dtype=torch.int32, device=dev)
    d_read_bases = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).to(dev)
    d_read_off = torch.from_numpy(read_off.view(np.int64)).to(dev)
    d_read_tok = torch.arange(nreads, dtype=torch.int64, device=dev)
    n_ref_win = sum(len(g) - K + 1 for g in genomes)
    n_read_win = nreads * (150 - K + 1)
    d_hit = torch.empty(n_read_win + 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    # Two contexts (= two HIP streams) on the GPU: the reference build and the read build are independent, so a
    # step submits both extractions before it finishes either; the small reference sort then runs underneath
    # the reads' extraction.  The match (on the reads' context) starts when both databases are complete.
    ctx = km.Context(local_rank)
    ctx_ref = km.Context(local_rank)
    lib, h, h_ref = ctx.lib, ctx.h, ctx_ref.h

    def submit(c, bases, off, ids, toks, nseq, kind, rc):
        c.check(lib.kmg_build_begin(c.h, K, KEY_LEN, rc, kind))
        c.check(lib.kmg_build_add_dev(c.h, bases.data_ptr(), bases.numel(), off.data_ptr(),
                                      ids.data_ptr() if ids is not None else None,
                                      toks.data_ptr() if toks is not None else None, nseq, None))

    def finish(c):
        db = C.c_void_p()
        nw = C.c_uint64()
        c.check(lib.kmg_build_finish(c.h, C.byref(db), C.byref(nw)))
        return db, nw.value

    stage_s = {"ref_build": 0.0, "read_build": 0.0, "match": 0.0}

    ranges = None
    if args.exchange:
        # The partition of the key space is a property of the reference, planned once: balanced ranges from the bucket
        # histogram of one complete build (every rank holds the whole input, so all ranks find the same ranges).
        from kmers_amd import dist as kd
        submit(ctx_ref, d_ref_bases, d_ref_off, d_ref_id, None, len(my_genomes), 0, 1)
        whole = km.KmerDatabase(ctx_ref, finish(ctx_ref)[0])
        ranges = kd.balanced_key_ranges(np.diff(whole.bucket_offsets().astype(np.int64)), world)
        whole.close()

    def step_exchange():
        """Key-range mode: this rank's range of the reference is built from the whole input (no reference k-mer leaves
        its GPU), the reads are built locally, their distinct k-mers are cut at the range bounds, exchanged, joined by
        the owners and the hit indices exchanged back (kmers_amd/dist.py; RCCL all-to-all under nccl)."""
        ctx_ref.check(lib.kmg_build_begin(h_ref, K, KEY_LEN, 1, 0))
        ctx_ref.check(lib.kmg_build_set_key_range(h_ref, ranges[rank][0], ranges[rank][1]))
        ctx_ref.check(lib.kmg_build_add_dev(h_ref, d_ref_bases.data_ptr(), d_ref_bases.numel(), d_ref_off.data_ptr(),
                                            d_ref_id.data_ptr(), None, len(my_genomes), None))
        submit(ctx, d_read_bases, d_read_off, None, d_read_tok, nreads, 1, 0)
        ref_h, nw_ref = finish(ctx_ref)
        shard = km.KmerDatabase(ctx_ref, ref_h)
        rdb_h, nw_reads = finish(ctx)
        rdb = km.KmerDatabase(ctx, rdb_h)
        _, hit = kd.match_key_range_shard(ctx, rdb, shard, ranges, want_owner=False)
        nw_all = int(kd.allreduce_sum(np.array([nw_ref]), device=dev if use_dist else "cpu")[0])
        rdb.close()
        shard.close()
        return nw_all, nw_reads

    def step(timed_stages=False):
        if args.exchange:
            return step_exchange()
        t0 = time.perf_counter()
        submit(ctx_ref, d_ref_bases, d_ref_off, d_ref_id, None, len(genomes), 0, 1)
        if timed_stages:                      # stage-by-stage wall clock: one stage at a time
            ref, nw_ref = finish(ctx_ref)
            t1 = time.perf_counter()
            submit(ctx, d_read_bases, d_read_off, None, d_read_tok, nreads, 1, 0)
            rdb, nw_reads = finish(ctx)
        else:
            submit(ctx, d_read_bases, d_read_off, None, d_read_tok, nreads, 1, 0)
            ref, nw_ref = finish(ctx_ref)
            t1 = time.perf_counter()
            rdb, nw_reads = finish(ctx)
        t2 = time.perf_counter()
        ctx.check(lib.kmg_match_dev(h, rdb, ref, d_hit.data_ptr()))
        if timed_stages:
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            stage_s["ref_build"] += t1 - t0
            stage_s["read_build"] += t2 - t1
            stage_s["match"] += t3 - t2
        info = (nw_ref, nw_reads)
        lib.kmg_db_free(rdb)
        lib.kmg_db_free(ref)
        return info

    if args.calib:
        ctx.check(lib.kmg_calib_copy(h, 1 << 27))   # 1 GiB read + 1 GiB written, known exactly
    for _ in range(args.warmup):
        info = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert info == (n_ref_win, n_read_win), (info, n_ref_win, n_read_win)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-stage rates and the roofline of the dominant kernel (extra, untimed steps) -----------
    prof_steps = 3
    for _ in range(prof_steps):
        step(timed_stages=True)
    for c in (ctx, ctx_ref):
        c.prof_enable(True)
        c.prof_reset()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    # the read build + match ran on `ctx`, the reference build on `ctx_ref`: kept apart ("<kernel>_ref")
    prof = {k: v for k, v in ctx.prof().items() if v["launches"]}
    for name, v in ctx_ref.prof().items():
        if v["launches"]:
            prof[name if name.endswith("_ref") else name + "_ref"] = v
    for c in (ctx, ctx_ref):
        c.prof_enable(False)
    cand = {k: v for k, v in prof.items() if v["launches"] and v["bytes"]}
    dom = max(cand, key=lambda k: cand[k]["ms"])
    d = cand[dom]
    avg_ms = d["ms"] / d["launches"]
    bytes_per_launch = d["bytes"] / d["launches"]
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = pmc_traffic(dom)
    # measured device-copy ceiling of this box (report the fraction of it next to the spec peak)
    ca = torch.empty(1 << 27, dtype=torch.int64, device=dev)
    cb = torch.ones(1 << 27, dtype=torch.int64, device=dev)
    ca.copy_(cb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ca.copy_(cb)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del ca, cb
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": int(bytes_per_launch),
                "launches_per_step": d["launches"] // prof_steps,
                "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4)}
    kernels = {k: {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": v["launches"] // prof_steps,
                   "algo_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None}
               for k, v in prof.items() if v["launches"]}

    if rank == 0:
        # key-range mode builds the reference once across the node; the replicated mode builds it on every GPU
        total_windows = n_ref_win + n_read_win * world if args.exchange else (n_ref_win + n_read_win) * world
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "k-mers/sec built + matched (reference build + read build + match, node)",
            "value": total_windows * args.steps / elapsed,
            "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2 in key-range mode: 5 Mbp reference (10 taxa) range-partitioned by k-mer "
                                    "prefix (every GPU builds the range it owns), 1M x 150 bp reads per GPU, their k-mers "
                                    "exchanged to the owners and joined there, k=31, key length 6"
                                    if args.exchange else
                                    "BASELINE config 2: 5 Mbp reference (10 taxa) + 1M x 150 bp reads per GPU, k=31, "
                                    "key length 6; build both DBs + exact match; inputs resident in HBM"),
                       "k": K, "key_len": KEY_LEN, "scale": args.scale,
                       "ref_windows": n_ref_win, "read_windows_per_gpu": n_read_win},
            "stages": None if args.exchange else {"ref_kmers_built_per_s": n_ref_win * prof_steps / stage_s["ref_build"],
                       "read_kmers_built_per_s": n_read_win * prof_steps / stage_s["read_build"],
                       "read_kmers_matched_per_s": n_read_win * prof_steps / stage_s["match"]},
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, taxa, genomes, reads)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    ctx_ref.close()
    if use_dist:
        dist.destroy_process_group()


# kernel id of the measurement hook -> kernel symbol in the rocprofv3 traces
# (the read build carries 32-bit payloads on this workload, so it shares the <unsigned int> instantiations with
# the reference build; tools/pmc_traffic.py keeps launches with different grids apart and the read build's is
# the larger one)
KERNEL_SYMBOL = {
    "sort_scatter_reads": "rs_scatter_kernel<unsigned int>", "sort_scatter_ref": "rs_scatter_kernel<unsigned int>",
    "sort_local_reads": "ls_sortidx_kernel<unsigned int, true>", "sort_local_ref": "ls_sort_kernel<unsigned int>",
    "sort_hist_reads": "rs_hist_kernel<unsigned int>", "sort_hist_ref": "rs_hist_kernel<unsigned int>",
    "extract_write": "extract_part_kernel<true, unsigned int>", "extract_write_ref": "extract_part_kernel<true, unsigned int>",
    "extract_count": "extract_part_kernel<false, unsigned int>", "extract_count_ref": "extract_part_kernel<false, unsigned int>",
    "match": "match_kernel",
}


def pmc_traffic(kernel_id):
    """HBM bytes per launch of the dominant kernel from the committed PMC measurement
    (profiles/*_pmc_traffic.json, produced by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE
    passes, calibrated on a copy of known size).  None when no measurement covers the kernel."""
    import glob
    sym = KERNEL_SYMBOL.get(kernel_id)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not sym or not files:
        return None
    try:
        data = json.load(open(files[-1]))
        hits = [int(v["traffic_bytes_per_launch"]) for name, v in data["kernels"].items() if sym in name]
        if hits:
            return min(hits) if kernel_id.endswith("_ref") else max(hits)
    except Exception:
        return None
    return None


def cpu_baseline(synth, taxa, genomes, reads):
    """The C restatement of the reference's algorithm (oracle/kmers_oracle.c: char-at-a-time extractor,
    per-key-bucket sort + reduce on a thread pool, two-pointer join) on a bounded sample of the same
    workload (the full reference + the first 400k reads, a few seconds to a few tens of seconds of CPU
    work depending on the host).  kind = "port": Kmers.jar needs a JVM, which neither box has."""
    from oracle.c_oracle import OracleDB
    threads = min(16, os.cpu_count() or 1)
    sample_reads = min(400_000, reads.shape[0])
    fa = synth.fasta_bytes(taxa, genomes)
    fq = synth.fastq_bytes(reads[:sample_reads])
    t0 = time.perf_counter()
    ref = OracleDB(fa, 0, K, KEY_LEN, threads=threads)
    rdb = OracleDB(fq, 1, K, KEY_LEN, threads=threads)
    rdb.match(ref, threads=threads)
    dt = time.perf_counter() - t0
    windows = ref.n_windows + rdb.n_windows
    return {"value": windows / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": "full 5 Mbp reference + first %d of the reads (%d windows) in %.1f s; extraction is "
                      "single-threaded as in the reference, sort/join use %d threads" % (sample_reads, windows, dt, threads)}


if __name__ == "__main__":
    main()
