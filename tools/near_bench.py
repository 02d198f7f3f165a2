#!/usr/bin/env python3
"""kmg_match_near on BASELINE config 2 (run ON the GPU box): every distinct read k-mer against the reference with one
and with two differences allowed; the sizing call (look-ups + scan) and the filling call (look-ups again, per-k-mer sort,
copy out -- the offsets of the sizing call are reused).
    python tools/near_bench.py [scale] > gpurun_out/near.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
taxa, genomes, reads = synth.cfg1(scale)
ctx = km.Context(0)
ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), 31, 6)
rdb, _ = km.make_read_db(ctx, synth.fastq_bytes(reads), 31, 6)
out = {"search_kmers": int(rdb.n_unique), "reference_kmers": int(ref.n_unique)}
for n in (0, 1, 2):
    cnt = C.c_uint64()
    t = time.perf_counter()
    ctx.check(ctx.lib.kmg_match_near(ctx.h, rdb.h, ref.h, n, None, None, None, 0, C.byref(cnt)))
    t_count = time.perf_counter() - t
    t = time.perf_counter()
    off, idx, dist = km.match_near(rdb, ref, n)          # its own sizing call finds the offsets already there
    t_full = time.perf_counter() - t
    probes = rdb.n_unique * (1 + (3 * 31 if n >= 1 else 0) + (9 * 31 * 30 // 2 if n >= 2 else 0))
    out["max_diff_%d" % n] = {"pairs": int(cnt.value), "sizing_call_s": round(t_count, 4), "filling_call_s": round(t_full, 4),
                              "lookups": int(probes), "lookups_per_s": round(probes / t_count, 0),
                              "by_distance": np.bincount(dist, minlength=3).tolist()}
print(json.dumps(out))
