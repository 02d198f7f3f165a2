#!/usr/bin/env python3
"""Two-word k-mers (33 <= k <= 64, csrc/wide.h) at BASELINE config 2's size on one MI355X (run ON the GPU box; not collected
by pytest -- it lives under tests/ because it calls the oracle, which is a checker):
    python tests/wide_bench.py [scale [k,k,... [min_k]]] > gpurun_out/wide.json
(min_k: k-mers of several lengths, min_k .. k -- the join then runs on both sides restricted to k - 3 bases)
The workload is config 2's (64 genomes, 5 Mbp, 1 M reads x 150 bp at scale 1) with k = 31 (the one-word path, for scale),
33, 47, 63 and 64.  Per k: reference build, read build, join -- wall clock from host buffers (PCIe included) and the
kernels' own time from the library's event hook (the two index sorts and the rank kernels of the two-word path are
not under the hook: wall clock is the figure to read there) -- and both databases and every hit index against the C
oracle's two-word build, which is timed beside it."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402
from kmers_amd.api import SequenceBatch  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
L = 6
taxa, genomes = synth.random_genomes(1, 64, int(5_000_000 * scale / 64))
reads = synth.random_reads(2, genomes, int(1_000_000 * scale), 150, 0.01)
ref_bases, ref_off = synth.batch_from_seqs(genomes)
ref_batch = SequenceBatch(ref_bases, ref_off, seq_id=np.array(taxa, np.int32))
n, rl = reads.shape
read_batch = SequenceBatch(np.ascontiguousarray(reads.reshape(-1)), (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)),
                           seq_tok=np.arange(n, dtype=np.int64))
fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
ctx = km.Context(0)


def timed(f):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t


MIN_K = int(sys.argv[3]) if len(sys.argv) > 3 else None


def build(kind, batch, k):
    fc = km.FileCreator(ctx, L, k, kind, rc=kind == "ref")
    fc.add_kmers(batch, min_length=k if MIN_K is None else MIN_K, max_length=k)
    db = fc.create()
    db.n_windows = fc.n_windows
    return db


out = {"config": {"workload": "BASELINE config 2 x %g at several k: %d genomes, %.1f Mbp, %d reads x %d; key length %d"
                  % (scale, len(taxa), ref_bases.size / 1e6, n, rl, L)}, "k": {}}
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [31, 33, 47, 63, 64]
for k in ks:
    for rep in range(2):                       # the second repetition is the one reported (pools warm)
        ref, t_ref = timed(lambda: build("ref", ref_batch, k))
        rdb, t_reads = timed(lambda: build("read", read_batch, k))
        if MIN_K is not None:
            (r2, s2), t_restrict = timed(lambda: (ref.restrict(k - 3), rdb.restrict(k - 3)))
            (hit, nh), t_match = timed(lambda: km.DB([r2]).get_nearest_kmers(s2))
            r2.close(); s2.close()
        else:
            t_restrict = 0.0
            (hit, nh), t_match = timed(lambda: km.DB([ref]).get_nearest_kmers(rdb))
        if rep == 0:
            ref.close()
            rdb.close()
    windows = int(ref.n_windows) + int(rdb.n_windows)
    rec = {"ref_windows": int(ref.n_windows), "ref_kmers": int(ref.n_unique), "read_windows": int(rdb.n_windows),
           "read_kmers": int(rdb.n_unique), "hits": int(nh),
           "ref_build_s": round(t_ref, 4), "read_build_s": round(t_reads, 4), "restrict_s": round(t_restrict, 4), "match_s": round(t_match, 4),
           "gpu_kmers_per_s_wall_from_host_buffers": round(windows / (t_ref + t_reads + t_restrict + t_match))}
    from oracle.c_oracle import OracleDB
    t = time.perf_counter()
    o_ref, o_reads = OracleDB(fa, 0, k, L, threads=16, min_k=MIN_K), OracleDB(fq, 1, k, L, threads=16, min_k=MIN_K)
    if MIN_K is not None:
        o_hit, o_nh = o_reads.restrict(k - 3).match(o_ref.restrict(k - 3), threads=16)
    else:
        o_hit, o_nh = o_reads.match(o_ref, threads=16)
    rec["oracle_s"] = round(time.perf_counter() - t, 2)
    rec["oracle_kmers_per_s_16_threads"] = round(windows / (time.perf_counter() - t))
    same = True
    for got, want in ((ref, o_ref), (rdb, o_reads)):
        h = got.fetch()
        same &= bool(np.array_equal(h["keys"], want.keys) and np.array_equal(h["val_off"], want.val_off)
                     and np.array_equal(h["val_a"], want.val_a) and np.array_equal(h["val_b"], want.val_b))
        if k > 32:
            same &= bool(np.array_equal(h["keys_lo"], want.keys_lo))
        if MIN_K is not None:
            same &= bool(np.array_equal(got.lens(), want.lens))
    same &= bool(nh == o_nh and np.array_equal(hit, o_hit))
    rec["identical_to_oracle"] = same
    del o_ref, o_reads
    out["k"][str(k)] = rec
    sys.stderr.write("k=%d %s\n" % (k, json.dumps(rec)))
    ref.close()
    rdb.close()
print(json.dumps(out))
