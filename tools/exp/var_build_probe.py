#!/usr/bin/env python3
"""Time the DEFAULT-options builds (-K 32 -k 24: k-mers of 24 .. 32 bases) at BASELINE config 2's size, inputs resident in
HBM, next to the one-length build (k = 31) -- run ON the GPU box: python tools/exp/var_build_probe.py [scale]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
glen, nreads = int(500_000 * scale), int(1_000_000 * scale)
taxa, genomes = synth.random_genomes(1, 10, glen)
reads = synth.random_reads(2, genomes, nreads, 150, 0.01)
ref_bases, ref_off = synth.batch_from_seqs(genomes)
d_ref = torch.from_numpy(ref_bases).to(dev)
d_ref_off = torch.from_numpy(ref_off.view(np.int64)).to(dev)
d_ref_id = torch.tensor(taxa, dtype=torch.int32, device=dev)
d_reads = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).to(dev)
d_read_off = torch.from_numpy((np.arange(nreads + 1, dtype=np.uint64) * np.uint64(150)).view(np.int64)).to(dev)
d_tok = torch.arange(nreads, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx = km.Context(0)
lib = ctx.lib


def build(k, min_k, which):
    bases, off, ids, toks, nseq, kind, rc = (d_ref, d_ref_off, d_ref_id, None, len(genomes), 0, 1) if which == "ref" else \
        (d_reads, d_read_off, None, d_tok, nreads, 1, 0)
    ctx.check(lib.kmg_build_begin(ctx.h, k, 6, rc, kind))
    if min_k != k:
        ctx.check(lib.kmg_build_set_min_k(ctx.h, min_k))
    ctx.check(lib.kmg_build_add_dev(ctx.h, bases.data_ptr(), bases.numel(), off.data_ptr(), ids.data_ptr() if ids is not None else None,
                                    toks.data_ptr() if toks is not None else None, nseq, None))
    db, nw = C.c_void_p(), C.c_uint64()
    ctx.check(lib.kmg_build_finish(ctx.h, C.byref(db), C.byref(nw)))
    d = km.KmerDatabase(ctx, db)
    u, v = d.n_unique, d.n_values
    d.close()
    return nw.value, u, v


for k, min_k in ((31, 31), (32, 32), (32, 24)):
    for which in ("ref", "reads"):
        build(k, min_k, which)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            nw, u, v = build(k, min_k, which)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print("K=%d k=%d %-5s %8.2f ms  windows %d  distinct %d  values %d  -> %.2f G records/s" % (k, min_k, which, ms, nw, u, v, nw / ms / 1e6))
