"""kmg_match_dev_noalloc (the join for callers that cannot allocate) against kmg_match_dev, through the C ABI.

Bit-exact bar: the same hit index for every search k-mer.  Two calls in a row on one context, the second with a
smaller input, so that whatever the first call left in the GPU's scratch array is stale for the second.  The
oracle's join pins both.  Collected last (test_zz_*): the round's end runs `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(ctx, fa, fq, k, l):
    import kmers_amd as km
    ref = km.make_reference_db(ctx, fa, k, l)
    rdb, _ = km.make_read_db(ctx, fq, k, l)
    return ref, rdb


def _both(ctx, rdb, ref):
    dev = torch.device("cuda", ctx.device)
    n = rdb.n_unique
    want = torch.full((n + 1,), -7, dtype=torch.int64, device=dev)
    got = torch.full((n + 1,), -7, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    ctx.check(ctx.lib.kmg_match_dev(ctx.h, rdb.h, ref.h, want.data_ptr()))
    ctx.check(ctx.lib.kmg_match_dev_noalloc(ctx.h, rdb.h, ref.h, got.data_ptr()))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    assert int(want[n]) == -7 and int(got[n]) == -7          # nothing written past the end
    return want[:n].cpu().numpy(), got[:n].cpu().numpy()


def test_noalloc_match_equals_match_dev_twice():
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    ctx = km.Context(0)
    try:
        # first call: ~100 tiles of search k-mers; second call: a fraction of one tile
        for seed, glen, nreads in ((31, 60000, 4000), (32, 3000, 40)):
            taxa, genomes = synth.random_genomes(seed, 3, glen, shared_frac=0.2)
            reads = synth.random_reads(seed + 100, genomes, nreads, 150, 0.01, n_rate=0.002)
            fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
            ref, rdb = _build(ctx, fa, fq, 31, 6)
            want, got = _both(ctx, rdb, ref)
            o_hit, o_nh = OracleDB(fq, 1, 31, 6).match(OracleDB(fa, 0, 31, 6))
            assert np.array_equal(got, want)
            assert np.array_equal(got, o_hit) and int((got >= 0).sum()) == o_nh and o_nh > 0
            rdb.close()
            ref.close()
    finally:
        ctx.close()


def test_noalloc_rejects_foreign_and_mismatched():
    import kmers_amd as km
    from kmers_amd import _lib, synth
    ctx, other = km.Context(0), km.Context(0)
    try:
        taxa, genomes = synth.random_genomes(5, 2, 2000)
        fa = synth.fasta_bytes(taxa, genomes)
        fq = synth.fastq_bytes(synth.random_reads(6, genomes, 20, 100, 0.0))
        ref, rdb = _build(ctx, fa, fq, 21, 4)
        ref_other = km.make_reference_db(other, fa, 21, 4)
        ref_k = km.make_reference_db(ctx, fa, 19, 4)
        out = torch.empty(rdb.n_unique, dtype=torch.int64, device="cuda:0")
        assert ctx.lib.kmg_match_dev_noalloc(ctx.h, rdb.h, ref_other.h, out.data_ptr()) == _lib.KMG_EINVAL
        assert ctx.lib.kmg_match_dev_noalloc(ctx.h, rdb.h, ref_k.h, out.data_ptr()) == _lib.KMG_EINVAL
        assert ctx.lib.kmg_match_dev_noalloc(ctx.h, rdb.h, ref.h, None) == _lib.KMG_EINVAL
        for d in (ref, rdb, ref_other, ref_k):
            d.close()
    finally:
        ctx.close()
        other.close()


def test_noalloc_two_contexts_back_to_back():
    """Two contexts of one GPU share the GPU's scratch array but have a stream each: their joins, queued one
    right behind the other with nothing waited for in between, must not overwrite each other's tile bounds -- the library
    orders them with an event per context.  Several rounds, a large join (many tiles) on one context and a different one
    on the other."""
    import kmers_amd as km
    from kmers_amd import synth
    a, b = km.Context(0), km.Context(0)
    dev = torch.device("cuda", 0)
    try:
        dbs = []
        for ctx, seed, glen, nreads in ((a, 41, 80000, 6000), (b, 42, 50000, 3000)):
            taxa, genomes = synth.random_genomes(seed, 3, glen, shared_frac=0.2)
            reads = synth.random_reads(seed + 100, genomes, nreads, 150, 0.01)
            ref, rdb = _build(ctx, synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads), 31, 6)
            want = torch.empty(rdb.n_unique, dtype=torch.int64, device=dev)
            ctx.check(ctx.lib.kmg_match_dev(ctx.h, rdb.h, ref.h, want.data_ptr()))
            dbs.append((ctx, ref, rdb, want, torch.empty(rdb.n_unique, dtype=torch.int64, device=dev)))
        torch.cuda.synchronize(dev)
        for ctx, *_ in dbs:
            torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
        for rnd in range(6):
            for ctx, ref, rdb, want, got in (dbs if rnd % 2 == 0 else dbs[::-1]):
                got.fill_(-9)
            torch.cuda.synchronize(dev)
            for ctx, ref, rdb, want, got in (dbs if rnd % 2 == 0 else dbs[::-1]):
                ctx.check(ctx.lib.kmg_match_dev_noalloc(ctx.h, rdb.h, ref.h, got.data_ptr()))
            for ctx, *_ in dbs:
                torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
            for ctx, ref, rdb, want, got in dbs:
                assert torch.equal(got, want), "round %d" % rnd
                assert int((got >= 0).sum()) > 0
        for ctx, ref, rdb, want, got in dbs:
            rdb.close()
            ref.close()
    finally:
        a.close()
        b.close()
