#!/usr/bin/env python3
"""Timing probe for ONE read build alone on the GPU (config-2 reads resident in HBM; no reference build on a second
stream underneath): kmg_build_begin + kmg_build_add_dev + kmg_build_finish in a loop with the library's launch timing on.
    KMERS_GPU_LIB=kmers_amd/libkmers_gpu_old.so python tools/exp/build_probe.py     # A/B against another build"""
import os
import sys
import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kmers_amd as km
from kmers_amd import synth

K, KEY_LEN = 31, 6
nreads = int(os.environ.get("PROBE_READS", "1000000"))
dev = torch.device("cuda", 0)
taxa, genomes = synth.random_genomes(1, 10, 500_000)
reads = synth.random_reads(2, genomes, nreads, 150, 0.01)
d_bases = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).to(dev)
d_off = torch.from_numpy((np.arange(nreads + 1, dtype=np.uint64) * np.uint64(150)).view(np.int64)).to(dev)
d_tok = torch.arange(nreads, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
if os.environ.get("PROBE_KIND") == "ref":          # config 2's reference instead: 10 genomes x 500 kbp, both strands, taxon payloads
    nreads = len(genomes)
    d_bases = torch.from_numpy(np.ascontiguousarray(np.concatenate(genomes))).to(dev)
    d_off = torch.from_numpy((np.arange(nreads + 1, dtype=np.uint64) * np.uint64(len(genomes[0]))).view(np.int64)).to(dev)
    d_ids = torch.arange(1, nreads + 1, dtype=torch.int32, device=dev)
ctx = km.Context(0)
lib = ctx.lib
ctx.prof_enable(True)
for it in range(13):
    if it == 3:
        torch.cuda.synchronize()
        ctx.prof_reset()
    if os.environ.get("PROBE_KIND") == "ref":
        ctx.check(lib.kmg_build_begin(ctx.h, K, KEY_LEN, 1, 0))
        ctx.check(lib.kmg_build_add_dev(ctx.h, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), d_ids.data_ptr(), None, nreads, None))
    else:
        ctx.check(lib.kmg_build_begin(ctx.h, K, KEY_LEN, 0, 1))
        ctx.check(lib.kmg_build_add_dev(ctx.h, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), None, d_tok.data_ptr(), nreads, None))
    h = C.c_void_p()
    nw = C.c_uint64()
    ctx.check(lib.kmg_build_finish(ctx.h, C.byref(h), C.byref(nw)))
    lib.kmg_db_free(h)
torch.cuda.synchronize()
p = ctx.prof()
tot = 0.0
out = []
for k_, v in p.items():
    if v["launches"]:
        ms = v["ms"] / 10
        tot += ms
        out.append("%s %.4f" % (k_, ms))
print(os.environ.get("KMERS_GPU_LIB", "default"), "sum %.4f |" % tot, " ".join(out))
