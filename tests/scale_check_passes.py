#!/usr/bin/env python3
"""One-off check of key-range builds ABOVE what one build can hold (run ON the GPU box; not collected by pytest):
random reads whose k-mer windows number more than 2^32 are built twice, in P1 and in P2 key-range passes
(kmg_build_set_key_range), the input resident in HBM and added in several batches per pass.
    python tests/scale_check_passes.py [million_reads [P1 [P2]]]
Checked: every pass's k-mers ascend strictly and lie in the pass's buckets; the passes' value counts add up to the
number of windows (every base is valid); both decompositions agree on the number of distinct k-mers and on
the sum of all k-mers mod 2^64; and the first pass holds, value for value, what the C oracle builds from the first
200 000 reads for those buckets (random 31-mers do not recur, so the later reads add nothing to those lists)."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import kmers_amd as km  # noqa: E402
from oracle.c_oracle import OracleDB  # noqa: E402

mreads = float(sys.argv[1]) if len(sys.argv) > 1 else 36.0
P1 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
P2 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K, L, RLEN = 31, 6, 150
nreads = int(mreads * 1e6)
per_batch = int(float(__import__("os").environ.get("KMERS_BATCH_READS", 6e6)))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(11)
letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
batches = []
for lo in range(0, nreads, per_batch):
    n = min(per_batch, nreads - lo)
    bases = letters[torch.randint(0, 4, (n * RLEN,), generator=g, device=dev)]
    off = torch.arange(n + 1, dtype=torch.int64, device=dev) * RLEN
    tok = torch.arange(lo, lo + n, dtype=torch.int64, device=dev)
    batches.append((bases, off, tok, n))
torch.cuda.synchronize()
windows = nreads * (RLEN - K + 1)
ctx = km.Context(0)
lib = ctx.lib
nkeys = 4 ** L


def build(first, end):
    ctx.check(lib.kmg_build_begin(ctx.h, K, L, 0, 1))
    ctx.check(lib.kmg_build_set_key_range(ctx.h, first, end))
    for bases, off, tok, n in batches:
        ctx.check(lib.kmg_build_add_dev(ctx.h, bases.data_ptr(), bases.numel(), off.data_ptr(), None, tok.data_ptr(), n, None))
    db, nw = C.c_void_p(), C.c_uint64()
    ctx.check(lib.kmg_build_finish(ctx.h, C.byref(db), C.byref(nw)))
    return km.KmerDatabase(ctx, db), nw.value


def run(passes, keep_first=False):
    tot_u = tot_v = tot_w = 0
    s1 = 0
    ok = True
    first_db = None
    for ps in range(passes):
        first, end = nkeys * ps // passes, nkeys * (ps + 1) // passes
        db, nw = build(first, end)
        keys = torch.empty(db.n_unique, dtype=torch.int64, device=dev)
        ctx.check(lib.kmg_db_copy_dev(db.h, 0, keys.data_ptr(), 0, db.n_unique))
        torch.cuda.synchronize()
        bucket = (keys >> (64 - 2 * L)) & (nkeys - 1)
        ok &= bool(((bucket >= first) & (bucket < end)).all())
        flipped = keys ^ (-2 ** 63)                                   # unsigned order in a signed tensor
        ok &= bool((flipped[1:] > flipped[:-1]).all())
        s1 = (s1 + int(keys.sum())) & (2 ** 64 - 1)                   # wraps: a sum mod 2^64
        tot_u += db.n_unique
        tot_v += db.n_values
        tot_w += nw
        if keep_first and ps == 0:
            first_db = (db, first, end)
        else:
            db.close()
        del keys, bucket, flipped
    return dict(unique=tot_u, values=tot_v, windows=tot_w, sum=s1, ok=ok), first_db


t0 = time.perf_counter()
a, first_db = run(P1, keep_first=True)
t1 = time.perf_counter()
b, _ = run(P2)
t2 = time.perf_counter()
ok = a["ok"] and b["ok"] and a["values"] == b["values"] == windows == a["windows"] == b["windows"]
ok &= a["unique"] == b["unique"] and a["sum"] == b["sum"]

# the first pass against the oracle on the first reads
db, first, end = first_db
sample = 200_000
sb = batches[0][0][:sample * RLEN].cpu().numpy().reshape(sample, RLEN)
fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(r), b"I" * RLEN) for i, r in enumerate(sb))
o = OracleDB(fq, 1, K, L, threads=16)
h = db.fetch()
bk = (o.keys >> np.uint64(64 - 2 * L)).astype(np.int64)
sel = np.nonzero((bk >= first) & (bk < end))[0]
pos = np.searchsorted(h["keys"], o.keys[sel])
found = pos < len(h["keys"])
found[found] &= h["keys"][pos[found]] == o.keys[sel][found]
ok &= bool(found.all())
if found.all():
    same = True
    for i, p in zip(sel[:200000].tolist(), pos[:200000].tolist()):
        a0, a1 = int(o.val_off[i]), int(o.val_off[i + 1])
        b0, b1 = int(h["val_off"][p]), int(h["val_off"][p + 1])
        if a1 - a0 != b1 - b0 or not np.array_equal(o.val_a[a0:a1], h["val_a"][b0:b1]) or not np.array_equal(o.val_b[a0:a1], h["val_b"][b0:b1]):
            same = False
            break
    ok &= same
db.close()
print(json.dumps({"reads": nreads, "windows": windows, "above_2_32": windows > 2 ** 32, "passes": [P1, P2],
                  "distinct_kmers": a["unique"], "values": a["values"], "sum_mod_2_64": a["sum"],
                  "seconds": [round(t1 - t0, 2), round(t2 - t1, 2)], "oracle_sample_keys": int(len(sel)), "consistent": bool(ok)}))
sys.exit(0 if ok else 1)
