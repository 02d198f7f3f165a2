#!/usr/bin/env python3
"""Timing probe for the extraction kernels alone (config-2 reads, resident in HBM): kmg_build_begin + kmg_build_add_dev
in a loop with the library's launch timing on, no finish.  Used with library variants built with -DEP_PROBE=n (the
write pass leaves after phase n; results are wrong, only the times mean something):
    KMERS_GPU_LIB=kmers_amd/libkmers_gpu_3.so python tools/exp/extract_probe.py"""
import os
import sys

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
ctx = km.Context(0)
lib = ctx.lib
ctx.prof_enable(True)
for it in range(13):
    if it == 3:
        torch.cuda.synchronize()
        ctx.prof_reset()
    ctx.check(lib.kmg_build_begin(ctx.h, K, KEY_LEN, 0, 1))
    ctx.check(lib.kmg_build_add_dev(ctx.h, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), None, d_tok.data_ptr(), nreads, None))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
torch.cuda.synchronize()
p = ctx.prof()
print(os.environ.get("KMERS_GPU_LIB", "default"), " ".join("%s %.4f" % (k, p[k]["ms"] / max(p[k]["launches"], 1)) for k in ("extract_count", "extract_write", "number_reads", "scan", "extract_tileseq")))
