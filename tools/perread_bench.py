#!/usr/bin/env python3
"""Per-read aggregation on BASELINE config 2 (run ON the GPU box): builds both databases as bench.py does, then
times kmg_collect_by_read (count only = join + record count; full = + emit, sort, unpack, copy out) and
kmg_classify_reads, and prints the kernels' HIP-event times.  tools/perread_bench.py [scale]"""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
K, L = 31, 6
taxa, genomes = synth.random_genomes(1, 10, 500_000, first_taxon=1001)
reads = synth.random_reads(2, genomes, int(1_000_000 * scale), 150, 0.01)
ctx = km.Context(0)
ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), K, L)
rdb, names = km.make_read_db(ctx, synth.fastq_bytes(reads), K, L)
out = {"reads": len(names), "read_kmers": rdb.n_unique, "read_occurrences": rdb.n_values}
for rep in range(2):
    n = C.c_uint64()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.kmg_collect_by_read(ctx.h, rdb.h, ref.h, C.byref(n), None, None, None))
    t1 = time.perf_counter()
    ctx.prof_enable(True)
    ctx.prof_reset()
    r, p, i = km.collect_by_read(rdb, ref)
    t2 = time.perf_counter()
    prof = {k: round(v["ms"], 3) for k, v in ctx.prof().items() if v["launches"]}
    ctx.prof_reset()
    call = km.classify_reads(rdb, ref, len(names))
    t3 = time.perf_counter()
    prof2 = {k: round(v["ms"], 3) for k, v in ctx.prof().items() if v["launches"]}
    ctx.prof_enable(False)
out.update(records=int(n.value), count_only_ms=round((t1 - t0) * 1e3, 2), collect_total_ms=round((t2 - t1) * 1e3, 2),
           collect_kernels_ms=prof, classify_total_ms=round((t3 - t2) * 1e3, 2), classify_kernels_ms=prof2,
           reads_called=int((call >= 0).sum()))
print(json.dumps(out))
