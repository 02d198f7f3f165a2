#!/usr/bin/env python3
"""One-off exactness check above the sizes the test suite covers (run ON the GPU box): builds BASELINE config 2 scaled
by `scale` on the GPU and with the C oracle and compares every array.  Not collected by pytest (the oracle is a checker: it lives under tests/).
python tests/scale_check.py [scale [batches]]
scale 1.7 -> 204 M read records: 17 prefix bits, i.e. three prefix digits (fused first digit + TWO global passes)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kmers_amd as km  # noqa: E402
from kmers_amd import synth  # noqa: E402
from oracle.c_oracle import OracleDB  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.7
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # > 1: the reads arrive in that many add calls
K, L = 31, 6
taxa, genomes, reads = synth.cfg1(scale)
fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
ctx = km.Context(0)
t0 = time.perf_counter()
ref = km.make_reference_db(ctx, fa, K, L)
if batches == 1:
    rdb, names = km.make_read_db(ctx, fq, K, L)
else:
    from kmers_amd import parse
    pf = parse.parse(fq, parse.FASTQ, K)
    b = pf.batch
    fc = km.FileCreator(ctx, L, K, "read", rc=False)
    cuts = [b.nseq * i // batches for i in range(batches + 1)]
    ridx = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
        ridx.append(fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi])))
    rdb = fc.create()
    ridx = np.concatenate(ridx)
    names = [pf.names[int(t)] for t in b.seq_tok[np.nonzero(ridx >= 0)[0]]]
hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
t1 = time.perf_counter()
o_ref = OracleDB(fa, 0, K, L, threads=16)
o_reads = OracleDB(fq, 1, K, L, threads=16)
o_hit, o_nh = o_reads.match(o_ref, threads=16)
t2 = time.perf_counter()
ok = True
for name, g, o in (("ref", ref, o_ref), ("reads", rdb, o_reads)):
    h = g.fetch()
    for arr in ("keys", "val_off", "val_a", "val_b", "bucket_off"):
        same = np.array_equal(h[arr], getattr(o, arr))
        ok &= same
        if not same:
            print("MISMATCH", name, arr)
ok &= names == o_reads.names and nh == o_nh and np.array_equal(hit, o_hit)
print(json.dumps({"scale": scale, "batches": batches, "read_records": int(rdb.n_values), "read_kmers": int(rdb.n_unique), "hits": int(nh),
                  "payload_bits": ctx.lib.kmg_db_payload_bits(rdb.h), "gpu_incl_host_parse_s": round(t1 - t0, 2),
                  "oracle_s": round(t2 - t1, 2), "identical": bool(ok)}))
sys.exit(0 if ok else 1)
