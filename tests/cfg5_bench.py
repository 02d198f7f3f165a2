#!/usr/bin/env python3
"""BASELINE config 5 at size on one MI355X (run ON the GPU box; not collected by pytest -- it lives under tests/ because
with `check` it calls the oracle, which is a checker): python tests/cfg5_bench.py [scale [check]] > gpurun_out/cfg5.json
Long, noisy reads (10 kbp, 15 % substitutions, 0.5 % N), genome sizes over three orders of magnitude, 5 % of the bases in
low-complexity tracts (a few key prefixes hold most of those k-mers: the oversize-segment paths of the sort), k = 21, 25,
31.  Per k: reference build, read build, join -- wall clock from host buffers (PCIe included), the kernels' own time from
the library's event hook -- and, with `check`, both databases and every hit index against the C oracle."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402
from kmers_amd.api import SequenceBatch  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
check = len(sys.argv) > 2 and sys.argv[2] == "check"
L = 6
t0 = time.perf_counter()
taxa, genomes, reads = synth.cfg5(scale)
sys.stderr.write("generated in %.1f s: %d taxa, %.1f Mbp reference (largest %.1f Mbp), %d reads x %d\n" % (
    time.perf_counter() - t0, len(taxa), sum(len(g) for g in genomes) / 1e6, len(genomes[0]) / 1e6, reads.shape[0], reads.shape[1]))
ref_bases, ref_off = synth.batch_from_seqs(genomes)
ref_batch = SequenceBatch(ref_bases, ref_off, seq_id=np.array(taxa, np.int32))
n, rl = reads.shape
read_batch = SequenceBatch(np.ascontiguousarray(reads.reshape(-1)), (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)),
                           seq_tok=np.arange(n, dtype=np.int64))
ctx = km.Context(0)


def timed(f):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t


def build(kind, batch, k):
    fc = km.FileCreator(ctx, L, k, kind, rc=kind == "ref")
    fc.add_kmers(batch, min_length=k, max_length=k)
    db = fc.create()
    db.n_windows = fc.n_windows
    return db


out = {"config": {"workload": "BASELINE config 5 x %g: %d taxa, %.1f Mbp reference (Zipf sizes, 5 %% low-complexity tracts), "
                  "%d reads x 10 kbp, 15 %% substitutions, 0.5 %% N; key length 6" % (scale, len(taxa), ref_bases.size / 1e6, n),
                  "checked_against_oracle": check}, "k": {}}
for k in (21, 25, 31):
    for rep in range(2):                       # the second repetition is the one reported (pools warm)
        ctx.prof_enable(True)
        ctx.prof_reset()
        ref, t_ref = timed(lambda: build("ref", ref_batch, k))
        p_ref = ctx.prof()
        ctx.prof_reset()
        rdb, t_reads = timed(lambda: build("read", read_batch, k))
        p_reads = ctx.prof()
        ctx.prof_reset()
        (hit, nh), t_match = timed(lambda: km.DB([ref]).get_nearest_kmers(rdb))
        p_match = ctx.prof()
        if rep == 0:
            ref.close()
            rdb.close()
    ksum = lambda p: sum(v["ms"] for v in p.values())
    top = lambda p: {kk: round(v["ms"], 3) for kk, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:6]}
    rec = {"ref_windows": int(ref.n_windows), "ref_kmers": int(ref.n_unique), "read_windows": int(rdb.n_windows),
           "read_kmers": int(rdb.n_unique), "hits": int(nh),
           "ref_build_s": round(t_ref, 4), "read_build_s": round(t_reads, 4), "match_s": round(t_match, 4),
           "ref_kernels_ms": round(ksum(p_ref), 2), "read_kernels_ms": round(ksum(p_reads), 2), "match_kernels_ms": round(ksum(p_match), 2),
           "read_kmers_built_per_s_kernels": round(rdb.n_windows / (ksum(p_reads) / 1e3)),
           "ref_top_kernels_ms": top(p_ref), "read_top_kernels_ms": top(p_reads),
           "read_payload_bits": int(ctx.lib.kmg_db_payload_bits(rdb.h))}
    if check:
        from oracle.c_oracle import OracleDB
        t = time.perf_counter()
        fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
        o_ref, o_reads = OracleDB(fa, 0, k, L, threads=16), OracleDB(fq, 1, k, L, threads=16)
        same = True
        for got, want in ((ref, o_ref), (rdb, o_reads)):
            h = got.fetch()
            same &= bool(np.array_equal(h["keys"], want.keys) and np.array_equal(h["val_off"], want.val_off)
                         and np.array_equal(h["val_a"], want.val_a) and np.array_equal(h["val_b"], want.val_b))
        o_hit, o_nh = o_reads.match(o_ref)
        same &= bool(nh == o_nh and np.array_equal(hit, o_hit))
        rec["identical_to_oracle"] = same
        rec["oracle_s"] = round(time.perf_counter() - t, 1)
        del o_ref, o_reads
    out["k"][str(k)] = rec
    sys.stderr.write("k=%d %s\n" % (k, json.dumps({a: b for a, b in rec.items() if not a.endswith("_top_kernels_ms")})))
    ref.close()
    rdb.close()
print(json.dumps(out))
