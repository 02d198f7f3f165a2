"""one-off: the failing fuzz configuration of the fold-first local sort, with details of what differs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kmers_amd as km
from kmers_amd import parse, synth
from oracle.c_oracle import OracleDB

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(1000 + seed)
k = int(rng.integers(8, 33)); l = int(rng.integers(1, min(k, 6) + 1))
n_gen, glen = int(rng.integers(1, 6)), int(rng.integers(500, 40000))
taxa, genomes = synth.random_genomes(seed, n_gen, glen, first_taxon=int(rng.integers(-5, 1000)), shared_frac=float(rng.choice([0.0, 0.3, 0.9])))
if rng.random() < 0.3:
    genomes[0][: glen // 2] = ord("A")
n_reads, rlen = int(rng.integers(50, 6000)), int(rng.integers(k, 260))
reads = synth.random_reads(seed + 1, genomes, n_reads, rlen, 0.02, n_rate=float(rng.choice([0.0, 0.01])))
fq = synth.fastq_bytes(reads)
ctx = km.Context(0)
b = parse.parse(fq, parse.FASTQ, k).batch
want = OracleDB(fq, 1, k, l, threads=4)
for cuts in ([0, b.nseq], [0, 4147, b.nseq]):
    fc = km.FileCreator(ctx, l, k, "read", rc=False)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
        fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi]))
    got = fc.create()
    h = got.fetch()
    print("cuts", cuts, "U", got.n_unique, want.U, "keys", np.array_equal(h["keys"], want.keys), "off", np.array_equal(h["val_off"], want.val_off),
          "a", np.array_equal(h["val_a"], want.val_a), "b", np.array_equal(h["val_b"], want.val_b))
    if np.array_equal(h["val_off"], want.val_off) and not np.array_equal(h["val_a"], want.val_a):
        bad = np.nonzero((h["val_a"] != want.val_a) | (h["val_b"] != want.val_b))[0]
        print("bad values", len(bad), "of", len(want.val_a), "first", bad[:10])
        ki = np.searchsorted(want.val_off, bad, side="right") - 1
        shown = 0
        for u in np.unique(ki)[:6]:
            s, e = int(want.val_off[u]), int(want.val_off[u + 1])
            g = list(zip(h["val_a"][s:e].tolist(), h["val_b"][s:e].tolist()))
            w = list(zip(want.val_a[s:e].tolist(), want.val_b[s:e].tolist()))
            print(" kmer", u, hex(int(want.keys[u])), "n", e - s, "same set", sorted(g) == sorted(w))
            print("   got ", g[:14]); print("   want", w[:14])
        segs = np.unique(want.keys[ki] >> np.uint64(64 - 9))
        print("prefixes hit (9 bits):", segs[:20], len(segs))
    got.close()
