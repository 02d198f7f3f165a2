#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer hot path on MI355X.

One step = BASELINE config 2 on one GPU, inputs already resident in HBM:
    build the reference DB (10 taxa x 500 kbp, both strands)      kmg_build_add_dev + kmg_build_finish
  + build the read DB (1M x 150 bp reads, 1 % substitutions)      kmg_build_add_dev + kmg_build_finish
  + match the read DB against the reference DB                    kmg_match_dev
at k = 31, key length 6.  value = (reference windows + read windows) / step time, whole job.
The two builds are independent and are submitted on two contexts (two HIP streams): the small reference build
runs underneath the reads' extraction; the match starts when both databases are complete.

N > 1 (`--gpus N`; one process per GPU, started here when no launcher has set WORLD_SIZE, else taken from the
launcher's RANK / LOCAL_RANK / WORLD_SIZE): the north-star split is the headline -- the reference database is
range-partitioned by k-mer prefix (every GPU builds only the key range it owns, kmg_build_set_key_range), every GPU
builds the read database of its own 1M reads, and the distinct read k-mers are bucketed by owner and exchanged with an
RCCL all-to-all; owners join what they received in one launch (kmg_match_keys_dev) and the hit indices travel back.
The replicated-reference mode (every GPU builds the whole small reference, no data-path collective) is timed beside
it and reported under "replicated".  After the timed loops one checked step asserts that the hits summed over ranks
are the same in both modes.

`--config 3` is the secondary line: BASELINE config 3 at full size on one GPU (1 Gbp / 10 k taxa reference, 100 M
reads streamed in chunks against the resident reference, TaxaCounts) -- see bench_config3().
`--config 4`: one rank's share of BASELINE config 4 (8 GPUs) on one GPU -- see bench_config4.py.

Adds "roofline" (dominant kernel, algorithmic bytes / HIP-event time measured live on the library's stream) and, on
rank 0 at N = 1, "cpu_baseline" (the C restatement of the reference's algorithm timed on this box's host cores).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~5 TB/s is the measured copy ceiling
K, KEY_LEN = 31, 6


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4),
                    help="BASELINE config: 2 = headline, 3 = 1 Gbp + 100 M reads + TaxaCounts (one GPU), 4 = one rank's share of the "
                         "8-GPU config (30 Gbp scanned for an eighth of the key buckets, 125 M reads) on one GPU")
    ap.add_argument("--scale", type=float, default=1.0, help="scale factor on config sizes (1.0 = the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calib", action="store_true", help="also launch the known-traffic copy kernel (PMC calibration)")
    ap.add_argument("--mode", choices=("auto", "key-range", "replicated"), default="auto",
                    help="N > 1: which split is the headline (auto = key-range, the north-star split; the other is "
                         "reported beside it).  KMERS_BENCH_FORCE_DIST=1 runs the key-range code path on one GPU too.")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n, script=None, argv=None, need_gpus=True, grace=20):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes BEFORE this process makes any GPU
    call (it never does), one per GPU, with the launch contract's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT); they inherit stdout, and only rank 0 writes the JSON line.  Fails loudly
    when the box has fewer than N GPUs.  -> exit status (0 = every rank succeeded)."""
    if need_gpus:
        import torch
        have = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this box -- not running\n" % (n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] +
                                      (sys.argv[1:] if argv is None else list(argv)), env=env))
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0):
                rc = rc or p.returncode
                deadline = deadline or time.time() + grace          # a rank died: the others would wait in a collective
        if deadline and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    for p in procs:
        rc = rc or p.returncode
    return rc


def source_hash():
    """Hash of the kernel sources the loaded library was built from (kmers_amd/csrc/*): committed PMC measurements
    carry it, and a measurement of other sources is not reported as this build's traffic."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kmers_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ one rank
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries the one JSON line and nothing else: RCCL prints a version banner on file descriptor 1 when the
    # communicator comes up, so everything but the result is sent to stderr at the descriptor level
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d\n" % (args.gpus, world, world))

    import numpy as np
    import torch
    import torch.distributed as dist

    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d wants GPU %d, %d visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.config in (3, 4):
        if world != 1:
            sys.stderr.write("bench.py: --config %d is a one-GPU line\n" % args.config)
            sys.exit(2)
        out = bench_config3(args, dev) if args.config == 3 else bench_config4(args, dev)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        return
    # KMERS_BENCH_FORCE_DIST=1: run the collective code path with a single rank too (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("KMERS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if "RANK" not in os.environ:           # the one-rank rehearsal without a launcher
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
            sk.close()
        dist.init_process_group("nccl", device_id=dev)
    headline = "single" if not use_dist else ("replicated" if args.mode == "replicated" else "key-range")

    import kmers_amd as km
    from kmers_amd import dist as kd, synth

    # ---- synthetic inputs, resident in HBM before anything is timed ------------------------------
    glen = max(1000, int(500_000 * args.scale))
    nreads = max(100, int(1_000_000 * args.scale))
    taxa, genomes = synth.random_genomes(1, 10, glen)
    reads = synth.random_reads(2 + rank, genomes, nreads, 150, 0.01)
    ref_bases, ref_off = synth.batch_from_seqs(genomes)          # every rank holds the whole reference input
    read_off = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(150)
    d_ref_bases = torch.from_numpy(ref_bases).to(dev)
    d_ref_off = torch.from_numpy(ref_off.view(np.int64)).to(dev)
    d_ref_id = torch.tensor(taxa, dtype=torch.int32, device=dev)
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

    def submit(c, bases, off, ids, toks, nseq, kind, rc, key_range=None):
        c.check(lib.kmg_build_begin(c.h, K, KEY_LEN, rc, kind))
        if key_range is not None:
            c.check(lib.kmg_build_set_key_range(c.h, key_range[0], key_range[1]))
        c.check(lib.kmg_build_add_dev(c.h, bases.data_ptr(), bases.numel(), off.data_ptr(),
                                      ids.data_ptr() if ids is not None else None,
                                      toks.data_ptr() if toks is not None else None, nseq, None))

    def finish(c):
        db = C.c_void_p()
        nw = C.c_uint64()
        c.check(lib.kmg_build_finish(c.h, C.byref(db), C.byref(nw)))
        return db, nw.value

    def n_unique(db):
        u = C.c_uint64()
        lib.kmg_db_info(db, None, None, None, None, C.byref(u), None)
        return u.value

    stage_s = {"ref_build": 0.0, "read_build": 0.0, "match": 0.0}

    ranges, comm = None, None
    if use_dist:
        # the library's own RCCL communicator (csrc/comm.h): the exchange runs under the C ABI.  Should it not come up
        # on some rank, every rank takes the torch.distributed form of the same exchange instead (kmers_amd/dist.py).
        exchange = "kmg_match_sharded (grouped ncclSend/ncclRecv inside the library)"
        try:
            if os.environ.get("KMERS_BENCH_EXCHANGE") == "torch":          # rehearsal of the other path
                raise RuntimeError("KMERS_BENCH_EXCHANGE=torch")
            comm = kd.Comm(ctx)
        except Exception as e:                  # noqa: BLE001 -- whatever it was, the ranks must agree on the path
            sys.stderr.write("bench.py: rank %d: kmg_comm_create failed (%s)\n" % (rank, e))
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None
            exchange = "torch.distributed all_to_all_single (the library's communicator did not come up)"
        # The partition of the key space is a property of the reference, planned once: balanced ranges from the bucket
        # histogram of one complete build (every rank holds the whole input, so all ranks find the same ranges).
        submit(ctx_ref, d_ref_bases, d_ref_off, d_ref_id, None, len(genomes), 0, 1)
        whole = km.KmerDatabase(ctx_ref, finish(ctx_ref)[0])
        ranges = kd.balanced_key_ranges(np.diff(whole.bucket_offsets().astype(np.int64)), world)
        whole.close()

    def step_key_range(keep_hits=False):
        """Key-range mode: this rank's range of the reference is built from the whole input (no reference k-mer leaves
        its GPU), the reads are built locally, their distinct k-mers are cut at the range bounds, exchanged, joined by
        the owners in one launch and the hit indices exchanged back (kmers_amd/dist.py; RCCL all-to-all under nccl)."""
        submit(ctx_ref, d_ref_bases, d_ref_off, d_ref_id, None, len(genomes), 0, 1, key_range=ranges[rank])
        submit(ctx, d_read_bases, d_read_off, None, d_read_tok, nreads, 1, 0)
        ref_h, nw_ref = finish(ctx_ref)
        shard = km.KmerDatabase(ctx_ref, ref_h)
        rdb_h, nw_reads = finish(ctx)
        rdb = km.KmerDatabase(ctx, rdb_h)
        _, hit = kd.match_key_range_shard(ctx, rdb, shard, ranges, want_owner=False, comm=comm)
        u = rdb.n_unique
        rdb.close()
        shard.close()
        return (nw_ref, nw_reads, u), (hit if keep_hits else None)

    def step_local(timed_stages=False):
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
        info = (nw_ref, nw_reads, n_unique(rdb))
        lib.kmg_db_free(rdb)      # (right behind the asynchronous join: the library orders the release behind it)
        lib.kmg_db_free(ref)
        return info

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(args.warmup):
            info = fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            info = fn()
        sync_all()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, info

    if args.calib:
        ctx.check(lib.kmg_calib_copy(h, 1 << 27))   # 1 GiB read + 1 GiB written, known exactly

    # ---- one checked step (untimed): the hits behind the headline figure.  In key-range mode it runs BEFORE anything is
    # timed and decides what is timed: the exchange has to give the whole-reference join's hits on every rank and the
    # same total over the ranks; if the library's exchange does not, the torch.distributed form of it is tried, and if
    # that does not either, the replicated mode becomes the headline and the line says so.
    def checked_step(with_key_range):
        info = step_local()
        torch.cuda.synchronize()
        u_reads = info[2]
        hits_local = int((d_hit[:u_reads] >= 0).sum().item())
        verified = {"distinct_read_kmers": u_reads, "hits": hits_local}
        if not with_key_range:
            return verified, True
        ok, hits_kr, nw_ref_kept = True, 0, 0
        try:
            (nw_ref_kept, _, u_kr), hit_kr = step_key_range(keep_hits=True)
            hits_kr = int((hit_kr >= 0).sum().item())
            # the same k-mers must be found (as a mask; the indices differ: one is into the shard, one into the whole)
            ok = u_kr == u_reads and hit_kr.numel() == u_reads and bool(((hit_kr >= 0) == (d_hit[:u_reads] >= 0)).all().item())
        except Exception as e:                  # noqa: BLE001 -- reported, and the ranks agree on what happens next
            sys.stderr.write("bench.py: rank %d: key-range step failed (%s)\n" % (rank, e))
            ok = False
        tot = torch.tensor([hits_local, hits_kr, nw_ref_kept, 0 if ok else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        tot = [int(x) for x in tot.tolist()]
        verified.update(hits_all_ranks_whole_reference=tot[0], hits_all_ranks_key_range=tot[1])
        return verified, tot[3] == 0 and tot[0] == tot[1] and tot[2] == n_ref_win

    verified = None
    if headline == "key-range":
        verified, ok = checked_step(True)
        if not ok and comm is not None:
            comm.close()
            comm = None
            exchange = "torch.distributed all_to_all_single (the library's exchange failed the checked step)"
            verified, ok = checked_step(True)
        if not ok:
            headline = "replicated"
            verified["key_range"] = "failed the checked step on this node: not timed, the replicated mode is the headline"
    runs = {}
    if headline == "single":
        runs["single"] = timed(step_local)
    else:
        if headline == "key-range":
            runs["key-range"] = timed(lambda: step_key_range()[0])
        runs["replicated"] = timed(step_local)
    elapsed, info = runs[headline]
    for mode, (_, inf) in runs.items():
        want_ref = n_ref_win if mode != "key-range" else inf[0]     # (a range build counts the windows it keeps)
        assert inf[:2] == (want_ref, n_read_win), (mode, inf, n_ref_win, n_read_win)
    if verified is None:
        verified, _ = checked_step(False)
    u_reads, hits_local = verified["distinct_read_kmers"], verified["hits"]

    # ---- per-stage rates and the roofline of the dominant kernel (extra, untimed steps) -----------
    headline_step = (lambda: step_key_range()[0]) if headline == "key-range" else step_local
    prof_steps = 10
    if not use_dist:
        for _ in range(3):
            step_local(timed_stages=True)
    for c in (ctx, ctx_ref):
        c.prof_enable(True)
        c.prof_reset()
    for _ in range(prof_steps):
        headline_step()
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
    traffic, traffic_src = pmc_traffic(dom)
    copy_gbs = measured_copy_gbs(torch, dev)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": int(bytes_per_launch),
                "launches_per_step": d["launches"] // prof_steps,
                "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4)}
    kernels = {k: {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": v["launches"] // prof_steps,
                   "algo_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None}
               for k, v in prof.items() if v["launches"]}
    roofline["stages"] = stage_fractions(lib, ctx, ctx_ref, submit, finish, prof, prof_steps, n_read_win, n_ref_win, u_reads,
                                         (d_ref_bases, d_ref_off, d_ref_id, len(genomes)), (d_read_bases, d_read_off, d_read_tok, nreads),
                                         ranges[rank] if headline == "key-range" else None)
    # matched rate = k-mers presented to the join (the DISTINCT read k-mers) / time of stage 3, with HIP
    # events on the library's stream, over prof_steps repetitions
    match_ms = prof["match"]["ms"] / max(prof["match"]["launches"], 1)
    stages = {"distinct_read_kmers_matched_per_s": u_reads / (match_ms * 1e-3), "match_ms": round(match_ms, 4),
              "match_timing": "HIP events around the join's launches, mean of %d" % prof["match"]["launches"]}
    if not use_dist:
        stages.update(ref_kmers_built_per_s=n_ref_win * 3 / stage_s["ref_build"],
                      read_kmers_built_per_s=n_read_win * 3 / stage_s["read_build"],
                      build_timing="wall clock, one stage at a time, mean of 3")

    if rank == 0:
        # key-range mode builds the reference once across the node; the replicated mode builds it on every GPU
        def line(mode, el):
            total = n_ref_win + n_read_win * world if mode == "key-range" else (n_ref_win + n_read_win) * world
            return total * args.steps / el, el / args.steps * 1e3
        value, ms_per_step = line(headline, elapsed)
        workload = {
            "single": "BASELINE config 2: 5 Mbp reference (10 taxa) + 1M x 150 bp reads, k=31, key length 6; build both DBs + "
                      "exact match; inputs resident in HBM",
            "replicated": "BASELINE config 2, replicated reference: every GPU builds the 5 Mbp reference and the DB of its own 1M x "
                          "150 bp reads and joins locally (no data-path collective), k=31, key length 6",
            "key-range": "BASELINE config 2 in key-range mode (the config-4 split): 5 Mbp reference (10 taxa) range-partitioned by "
                         "k-mer prefix (every GPU builds the range it owns), 1M x 150 bp reads per GPU, their distinct k-mers "
                         "exchanged to the owners by RCCL all-to-all, joined there in one launch, hit indices sent back; k=31, "
                         "key length 6"}[headline]
        out = {
            "metric": "k-mers/sec built + matched (reference build + read build + match, node)",
            "value": value,
            "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "mode": headline, "k": K, "key_len": KEY_LEN, "scale": args.scale,
                       "ref_windows": n_ref_win, "read_windows_per_gpu": n_read_win,
                       **({"exchange": exchange} if use_dist else {})},
            "stages": stages,
            "verified": verified,
            "roofline": roofline,
            "kernels": kernels,
        }
        for mode, (el, _) in runs.items():
            if mode != headline:
                v, ms = line(mode, el)
                out[mode.replace("-", "_")] = {"value": v, "ms_per_step": ms, "unit": "k-mers/s"}
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            out["cpu_baseline"], o_u, o_nh = cpu_baseline(synth, taxa, genomes, reads)
            if o_u is not None:      # the whole config went through the oracle: the step's results are pinned by it
                assert (u_reads, hits_local) == (o_u, o_nh), ("GPU step vs oracle", u_reads, hits_local, o_u, o_nh)
                out["verified"]["against"] = "oracle (cpu_baseline leg): distinct read k-mers and hit count equal"
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    ctx.close()
    ctx_ref.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def measured_copy_gbs(torch, dev):
    """Device-copy ceiling of this box (report the fraction of it next to the spec peak)."""
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
    return 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9


ACHIEVABLE_COPY_GBS = 6290.0     # MI355X_MICROARCH.md: what a streaming copy reaches of the 8 TB/s spec peak

# measurement-hook kernel ids per stage of SURVEY 8(d) (read build = the step's critical path; "<id>_ref" = the
# reference build, which runs underneath it on the second stream).  Every prefix sum -- the fused first digit's table,
# the global pass's table, the run heads -- counts as sort + dedup: they exist because of the partition.
STAGE_KERNELS = {
    "extract": ["extract_tileseq", "extract_count", "extract_write", "number_reads"],
    "sort_dedup": ["scan", "sort_hist_reads", "sort_hist_ref", "sort_scan", "sort_scatter_reads", "sort_scatter_ref", "sort_segments",
                   "sort_local_reads", "sort_local_ref", "reduce_flag", "reduce_write", "bucket_offsets"],
    "match": ["match"],
}
# kernel symbols of the rocprofv3 counter runs per stage (both builds: the counters cannot tell them apart by name)
STAGE_SYMBOLS = {
    "extract": ["extract_", "seq_maxlen", "number_", "sum_u32", "id_bits"],
    "sort_dedup": ["rs_", "ls_", "rd_", "scan_", "big_copy", "set_u64"],
    "match": ["match_"],
}


def stage_fractions(lib, ctx, ctx_ref, submit, finish, prof, prof_steps, n_read_win, n_ref_win, u_reads, ref_in, read_in, key_range):
    """SURVEY 8(d)'s headline per stage: B_stage / t_stage / 8e12 with its formulas, P as stored on the device
    (extract: 1 B per base in + (8 + P) B per record out; sort + dedup: 2 (8 + P) B per record + 16 B per distinct k-mer
    + P' B per value; match: 8 B per search and per reference k-mer in + 8 B per search k-mer out).  t_stage = HIP-event
    time of the stage's launches on the read build's stream (the step's critical path); the reference build's bytes and
    times are given beside it.  traffic = the stage's kernels summed over the newest PMC file of these sources."""
    (rb, ro, ri, rn), (qb, qo, qt, qn) = ref_in, read_in
    submit(ctx_ref, rb, ro, ri, None, rn, 0, 1, key_range=key_range)
    submit(ctx, qb, qo, None, qt, qn, 1, 0)
    ref, _ = finish(ctx_ref)
    rdb, _ = finish(ctx)
    u, v = C.c_uint64(), C.c_uint64()
    lib.kmg_db_info(ref, None, None, None, None, C.byref(u), C.byref(v))
    u_ref, v_ref = u.value, v.value
    lib.kmg_db_info(rdb, None, None, None, None, C.byref(u), C.byref(v))
    v_reads = v.value
    p_reads = 4 if lib.kmg_db_payload_bits(rdb) == 32 else 8
    lib.kmg_db_free(rdb)
    lib.kmg_db_free(ref)
    n_ref_rec = 2 * n_ref_win                      # both strands (odd k: no palindromes)
    algo = {
        "extract": (qb.numel() + (8 + p_reads) * n_read_win, rb.numel() + 12 * n_ref_rec),
        "sort_dedup": (2 * (8 + p_reads) * n_read_win + 16 * u_reads + p_reads * v_reads, 2 * 12 * n_ref_rec + 16 * u_ref + 8 * v_ref),
        "match": (8 * (u_reads + u_ref) + 8 * u_reads, 0),
    }
    traffic = pmc_stage_traffic()
    out = {}
    for st, ids in STAGE_KERNELS.items():
        # (ids ending in _ref are the reference build's own instantiations; the others' reference-build launches were
        # renamed <id>_ref when the two contexts' tables were merged)
        ms = sum(prof[i]["ms"] for i in ids if i in prof and not i.endswith("_ref")) / prof_steps
        ms_ref = (sum(prof[i]["ms"] for i in ids if i in prof and i.endswith("_ref")) +
                  sum(prof[i + "_ref"]["ms"] for i in ids if not i.endswith("_ref") and i + "_ref" in prof)) / prof_steps
        b, b_ref = algo[st]
        gbs = b / (ms * 1e-3) / 1e9 if ms else 0.0
        out[st] = {"algo_bytes": int(b), "ms": round(ms, 4), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                   "frac_of_achievable_copy": round(gbs / ACHIEVABLE_COPY_GBS, 4),
                   "reference_build_underneath": {"algo_bytes": int(b_ref), "ms": round(ms_ref, 4)},
                   "traffic_both_builds": traffic.get(st), "algo_bytes_both_builds": int(b + b_ref)}
        if traffic.get(st):
            out[st]["traffic_over_algo"] = round(traffic[st] / (b + b_ref), 3)
    out["sort_dedup"]["passes"] = {"global_hbm": prof.get("sort_scatter_reads", {}).get("launches", 0) // prof_steps,
                                   "fused_into_the_extractor": 1, "local_lds": 1,
                                   "note": "effective traffic = (global + fused write + local) x 2 x (8 + P) x records"}
    out["payload_bytes_as_stored"] = p_reads
    out["timing"] = "HIP events around every launch of the read build's stream, mean of %d steps" % prof_steps
    return out


def _pmc_file():
    """The committed PMC measurement of the default workload for these kernel sources: the newest
    profiles/*_pmc_traffic.json (config 3's own measurements aside) whose source hash is this library's.
    -> (parsed file, name), or (None, reason)."""
    import glob
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")) if "cfg3" not in f),
                   key=os.path.getmtime)
    if not files:
        return None, "no measurement"
    h = source_hash()
    for f in reversed(files):
        try:
            data = json.load(open(f))
        except Exception:
            continue
        if data.get("source_hash") == h:
            return data, os.path.basename(f)
    return None, "%s was measured on other kernel sources" % os.path.basename(files[-1])


def pmc_stage_traffic():
    """HBM bytes per step and stage from the newest committed PMC file, if it was measured on these sources."""
    data, _ = _pmc_file()
    if data is None:
        return {}
    try:
        ks = data["kernels"]
        steps = max(v["launches"] for name, v in ks.items() if "extract_part_count_kernel<false, false" in name)
        out = {}
        for st, pats in STAGE_SYMBOLS.items():
            tot = 0.0
            for name, v in ks.items():
                base = name.split("void ")[-1]
                if any(base.startswith(p_) for p_ in pats):
                    tot += v["traffic_bytes_per_launch"] * v["launches"] / steps
            out[st] = int(tot)
        return out
    except Exception:
        return {}


# kernel id of the measurement hook -> kernel symbol in the rocprofv3 traces
# (the read build carries 32-bit payloads on this workload, so it shares the <unsigned int> instantiations with
# the reference build; tools/pmc_traffic.py keeps launches with different grids apart and the read build's is
# the larger one)
KERNEL_SYMBOL = {
    "sort_scatter_reads": "rs_scatter_kernel<unsigned int>", "sort_scatter_ref": "rs_scatter_kernel<unsigned int>",
    "sort_local_reads": "ls_sortidx_kernel<unsigned int, true", "sort_local_ref": "ls_sort_kernel<unsigned int",
    "sort_hist_reads": "rs_hist_kernel<unsigned int>", "sort_hist_ref": "rs_hist_kernel<unsigned int>",
    "extract_write": "extract_part_write_kernel<unsigned int", "extract_write_ref": "extract_part_write_kernel<unsigned int",
    "extract_count": "extract_part_count_kernel<false, false", "extract_count_ref": "extract_part_count_kernel<true, false",
    "match": "match_kernel",
}


def pmc_traffic(kernel_id):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC measurement
    (profiles/*_pmc_traffic.json, produced by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE
    passes, calibrated on a copy of known size).  -> (bytes, file name), or (None, reason) when no measurement covers
    the kernel or the measurement was taken on other kernel sources than the ones this library was built from."""
    sym = KERNEL_SYMBOL.get(kernel_id)
    if not sym:
        return None, "no measurement"
    data, name = _pmc_file()
    if data is None:
        return None, name
    try:
        hits = [int(v["traffic_bytes_per_launch"]) for kname, v in data["kernels"].items() if sym in kname]
        if hits:
            return (min(hits) if kernel_id.endswith("_ref") else max(hits)), name
    except Exception:
        pass
    return None, "no measurement"


def cpu_baseline(synth, taxa, genomes, reads):
    """The C restatement of the reference's algorithm (oracle/kmers_oracle.c: char-at-a-time extractor,
    per-key-bucket sort + reduce on a thread pool, two-pointer join) on the WHOLE of config 2 (125 M windows, a few
    seconds to a few tens of seconds of CPU work depending on the host; a host slower than ~3 M windows/s gets the first
    400 k reads only).  kind = "port": Kmers.jar needs a JVM, which neither box has.
    -> (record, distinct read k-mers, hits) -- the last two None unless the whole config was run."""
    from oracle.c_oracle import OracleDB
    threads = min(16, os.cpu_count() or 1)
    fa = synth.fasta_bytes(taxa, genomes)
    t0 = time.perf_counter()
    ref = OracleDB(fa, 0, K, KEY_LEN, threads=threads)
    t_ref = time.perf_counter() - t0
    whole = ref.n_windows / max(t_ref, 1e-9) > 3e6
    sample_reads = reads.shape[0] if whole else min(400_000, reads.shape[0])
    fq = synth.fastq_bytes(reads[:sample_reads])
    t0 = time.perf_counter()
    rdb = OracleDB(fq, 1, K, KEY_LEN, threads=threads)
    _, nh = rdb.match(ref, threads=threads)
    dt = t_ref + time.perf_counter() - t0
    windows = ref.n_windows + rdb.n_windows
    rec = {"value": windows / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
           "sample": "%s: full 5 Mbp reference + %d reads (%d windows) in %.1f s; extraction is "
                     "single-threaded as in the reference, sort/join use %d threads"
                     % ("the whole config" if sample_reads == reads.shape[0] else "bounded sample", sample_reads, windows, dt, threads)}
    if sample_reads == reads.shape[0]:
        return rec, int(rdb.U), int(nh)
    return rec, None, None


def bench_config3(args, dev):
    import bench_config3 as b3
    return b3.run(args, dev)


def bench_config4(args, dev):
    import bench_config4 as b4
    return b4.run(args, dev)


if __name__ == "__main__":
    main()
