import sys; sys.path.insert(0,__import__('os').getcwd()); sys.path.insert(0,__import__('os').getcwd()+'/tests')
import numpy as np
import kmers_amd as km
from oracle.c_oracle import OracleDB
import test_gpu_parity as T
rng = np.random.default_rng(31)
fa = T._rand_fasta(rng, 40, 60, b"AACCGTN", [5, 9, 1234567, 77])
fq = T._rand_fastq(rng, 300, 30, b"AACGTN")
ctx = km.Context(0)
rdb, names = km.make_read_db(ctx, fq, 5, 2, min_k=3)
want = OracleDB(fq, 1, 5, 2, min_k=3)
g = rdb.fetch()
print("U", rdb.n_unique, len(want.keys), "V", rdb.n_values, len(want.val_a))
print("keys eq", np.array_equal(g["keys"], want.keys), "off eq", np.array_equal(g["val_off"], want.val_off))
bad = np.nonzero(g["val_a"] != want.val_a)[0]
print("bad val_a", len(bad), bad[:10])
badb = np.nonzero(g["val_b"] != want.val_b)[0]
print("bad val_b", len(badb), badb[:10])
lens = rdb.lens()
for i in bad[:5]:
    u = np.searchsorted(g["val_off"], i, side="right") - 1
    print("value", i, "kmer", u, "len", lens[u], "got", g["val_a"][i], g["val_b"][i], "want", want.val_a[i], want.val_b[i],
          "row got", list(zip(g["val_a"][g["val_off"][u]:g["val_off"][u+1]], g["val_b"][g["val_off"][u]:g["val_off"][u+1]]))[:8],
          "row want", list(zip(want.val_a[want.val_off[u]:want.val_off[u+1]], want.val_b[want.val_off[u]:want.val_off[u+1]]))[:8])
