#!/usr/bin/env python3
"""Where the sharded join's time goes on one GPU (run ON the GPU box): the steps of dist.match_key_range_shard --
copying the search k-mers out of the library, loading them back as a keys-only database, the join -- timed one by
one on BASELINE config 2, then the whole call with and without the owner array.
    python tools/exchange_timing.py"""
import sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import kmers_amd as km
from kmers_amd import synth, dist as kd
dev = torch.device('cuda', 0)
taxa, genomes = synth.random_genomes(1, 10, 500000)
reads = synth.random_reads(2, genomes, 1000000, 150, 0.01)
ctx = km.Context(0)
ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), 31, 6)
rdb, _ = km.make_read_db(ctx, synth.fastq_bytes(reads), 31, 6)
ranges = [(0, 4096)]
def T(label, f, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); print(label, round((time.perf_counter() - t) / n * 1e3, 3), 'ms'); return r
keys = torch.empty(rdb.n_unique, dtype=torch.int64, device=dev)
T('copy keys', lambda: ctx.check(ctx.lib.kmg_db_copy_dev(rdb.h, 0, keys.data_ptr(), 0, keys.numel())))
T('bucket_offsets', lambda: rdb.bucket_offsets())
lm = kd.shard_matcher(ctx, rdb, ref)
T('local_match', lambda: lm(keys, [keys.numel()]))
no_vals = torch.empty(0, dtype=torch.int32, device=dev)
z = T('zeros', lambda: torch.zeros(keys.numel() + 1, dtype=torch.int64, device=dev))
q = T('load_tensors', lambda: km.KmerDatabase.load_tensors(ctx, 31, 6, False, 1, keys, z, no_vals, no_vals))
out = torch.empty(keys.numel(), dtype=torch.int64, device=dev)
from kmers_amd.api import match_into
T("match_into", lambda: match_into(ctx, q, ref, out))
T("close", lambda: q.close(), 1)
T('whole', lambda: kd.match_key_range_shard(ctx, rdb, ref, ranges, want_owner=False))
T('whole+owner', lambda: kd.match_key_range_shard(ctx, rdb, ref, ranges))
