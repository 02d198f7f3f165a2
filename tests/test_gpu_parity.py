"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bar: bit-exact (integer / byte / index work).  The oracle is the C restatement (oracle/kmers_oracle.c),
itself checked byte for byte against the literal Python restatement and the Appendix-B known answers
(tests/test_oracle_*.py); parity with Kmers.jar itself is UNPINNED (no JVM, no reference fixtures).
Read-position lists are compared in the canonical ascending order (the Java HashSet order is
nondeterministic).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kmers_amd as km
    c = km.Context(0)
    yield c
    c.close()


def _same_db(got, want):
    h = got.fetch()
    assert got.n_unique == want.U and got.n_values == want.V
    assert np.array_equal(h["keys"], want.keys)
    assert np.array_equal(h["val_off"], want.val_off)
    assert np.array_equal(h["val_a"], want.val_a)
    assert np.array_equal(h["val_b"], want.val_b)
    assert np.array_equal(h["bucket_off"], want.bucket_off)


def _rand_fasta(rng, n_seq, max_len, alphabet, taxa):
    parts = []
    for _ in range(n_seq):
        L = int(rng.integers(0, max_len))
        s = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), size=L).tolist())
        parts.append(b">%d d\n%s\n" % (taxa[int(rng.integers(0, len(taxa)))], s))
    return b"".join(parts)


def _rand_fastq(rng, n_seq, max_len, alphabet):
    parts = []
    for i in range(n_seq):
        L = int(rng.integers(0, max_len))
        s = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), size=L).tolist())
        name = b"r%d" % (i - 1 if i % 7 == 3 else i)      # some neighbours share a header
        parts.append(b"@%s\n%s\n+\n%s\n" % (name, s, b"I" * L))
    return b"".join(parts)


# ---------------------------------------------------------------------------- known answers (App. B)
def test_kat_b12_b15(ctx):
    import kmers_amd as km
    ref = km.make_reference_db(ctx, b">7\nACGTNACGTA\n", 4, 2)
    blocks, off = ref.encode_blocks()
    blk = lambda k: blocks[int(off[k]):int(off[k + 1])]
    assert ref.meta() == bytes.fromhex("040402000000010000080001")
    assert blk(1) == bytes.fromhex("00041E00010000000704")
    assert blk(7) == bytes.fromhex("00047800010000000701")
    assert blk(8) == bytes.fromhex("00048700010000000701")
    assert all(blk(k) == b"" for k in range(16) if k not in (1, 7, 8))
    fq = b"@r1\nACGTA\n+\nIIIII\n@r2\nNNNNN\n+\nIIIII\n@r3\nTTTTT\n+\nIIIII\n"
    rdb, names = km.make_read_db(ctx, fq, 4, 2)
    assert names == ["r1", "r3"]
    assert rdb.meta() == bytes.fromhex("04040200000002000001010000080400")
    rb, roff = rdb.encode_blocks()
    assert rb[int(roff[10]):int(roff[11])] == bytes.fromhex("0004AA00000002000000010000000000010001")
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    assert km.match_lines(rdb, ref, hit) == ["ACGT\t0:0\t-|7x4", "CGTA\t0:1\t-|7x1"]
    assert km.taxa_count_lines(ref) == ["7\t3\t3"]


# ---------------------------------------------------------------------------- build parity
@pytest.mark.parametrize("k,key_len,seed,alphabet", [
    (4, 2, 1, b"AACCGTN"), (5, 2, 2, b"ACGTN"), (6, 3, 3, b"ACGT"), (8, 4, 4, b"ACGTacgtN"),
    (21, 6, 5, b"ACGT"), (31, 6, 6, b"ACGTN"), (32, 6, 7, b"ACGT"), (1, 1, 8, b"ACGTN"), (16, 8, 9, b"AC"),
])
def test_reference_build_matches_oracle(ctx, k, key_len, seed, alphabet):
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    fa = _rand_fasta(rng, 40, 700, alphabet, (5, 7, 11, -3, 70000, 2 ** 31 - 1))
    got = km.make_reference_db(ctx, fa, k, key_len)
    want = OracleDB(fa, 0, k, key_len)
    _same_db(got, want)
    assert got.n_windows == want.n_windows
    gb, goff = got.encode_blocks()
    wb, woff = want.encode_blocks()
    assert gb == wb and np.array_equal(goff, woff)
    got.close()


@pytest.mark.parametrize("k,key_len,seed,alphabet", [
    (4, 2, 11, b"AACGTN"), (7, 3, 12, b"ACGTN"), (31, 6, 13, b"ACGTN"), (32, 5, 14, b"ACGT"), (12, 6, 15, b"AC"),
])
def test_read_build_matches_oracle(ctx, k, key_len, seed, alphabet):
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    fq = _rand_fastq(rng, 300, 260, alphabet)
    got, names = km.make_read_db(ctx, fq, k, key_len)
    want = OracleDB(fq, 1, k, key_len)
    _same_db(got, want)
    assert names == want.names
    gb, goff = got.encode_blocks()
    wb, woff = want.encode_blocks()
    assert gb == wb and np.array_equal(goff, woff)
    got.close()


def test_both_rank_paths_of_the_fused_extractor(monkeypatch):
    """The fused extractor ranks a tile's records by digit either with returning LDS adds (where the device's probe at
    kmg_create says their return values are stable ranks) or with wave-level digit matching (KMG_RANK_BY_MATCH forces
    it): both against the oracle, reads (deep coverage: many equal k-mers whose positions must stay ascending) and a
    reference with both strands."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(71, 3, 30000, shared_frac=0.3)
    reads = synth.random_reads(72, genomes, 6000, 150, 0.01, n_rate=0.002)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    want_ref, want_reads = OracleDB(fa, 0, 31, 6), OracleDB(fq, 1, 31, 6)
    for by_match in (False, True):
        if by_match:
            monkeypatch.setenv("KMG_RANK_BY_MATCH", "1")
        c = km.Context(0)
        try:
            ref = km.make_reference_db(c, fa, 31, 6)
            rdb, names = km.make_read_db(c, fq, 31, 6)
            _same_db(ref, want_ref)
            _same_db(rdb, want_reads)
            assert names == want_reads.names
            ref.close()
            rdb.close()
        finally:
            c.close()


def test_edge_cases(ctx):
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    for fa in (b"", b">1\n\n", b">1\nACG\n>2\nAC\n", b">1\n" + b"N" * 100 + b"\n", b">1\nACGTACGTAC",
               b">4\n" + b"A" * 5000 + b"\n", b">1\nACGT\n" * 3000):
        got = km.make_reference_db(ctx, fa, 5, 2)
        want = OracleDB(fa, 0, 5, 2)
        _same_db(got, want)
        got.close()
    for fq in (b"", b"@a\n\n+\n\n", b"@a\nACG\n+\nIII\n", b"@a\nNNNNNNN\n+\nIIIIIII\n"):
        got, names = km.make_read_db(ctx, fq, 5, 2)
        want = OracleDB(fq, 1, 5, 2)
        _same_db(got, want)
        assert names == want.names
        got.close()


def test_long_read_position_wraps_like_java_short(ctx):
    """(short)(pos - maxK), KmersFromFile.java:165: offsets past 32767 wrap negative."""
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(5)
    s = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=40000).tolist())
    fq = b"@long\n%s\n+\n%s\n" % (s, b"I" * len(s))
    got, _ = km.make_read_db(ctx, fq, 31, 6)
    want = OracleDB(fq, 1, 31, 6)
    _same_db(got, want)
    assert got.fetch()["val_b"].min() < 0
    got.close()


def test_compact_read_payload_limits(ctx):
    """Read builds carry (read, pos) in 32 bits when the batch allows it (pos bits from the longest read, the rest
    for the read number).  Both sides of the limits, against the oracle: short reads (compact), neighbouring records
    with equal headers (one read index shared: the payload digits are sorted too), a read longer than 65536 + k
    (not compact, positions wrap as Java's short does), and compact followed by a second batch (widened)."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(71, 3, 20000, shared_frac=0.2)
    reads = synth.random_reads(72, genomes, 3000, 80, 0.01)
    k, l = 21, 4
    fq = synth.fastq_bytes(reads)
    # pairs of records with the same header
    recs = fq.split(b"@")[1:]
    paired = b"".join(b"@" + (recs[i - (i & 1)].split(b"\n", 1)[0]) + b"\n" + recs[i].split(b"\n", 1)[1] for i in range(len(recs)))
    rng = np.random.default_rng(73)
    long_seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=70_000).tolist())
    with_long = fq + b"@long\n%s\n+\n%s\n" % (long_seq, b"I" * len(long_seq))
    for data, bits in ((fq, 32), (paired, 32), (with_long, 64)):
        got, names = km.make_read_db(ctx, data, k, l)
        assert ctx.lib.kmg_db_payload_bits(got.h) == bits
        want = OracleDB(data, 1, k, l)
        _same_db(got, want)
        assert names == want.names
        gb, goff = got.encode_blocks()
        wb, woff = want.encode_blocks()
        assert gb == wb and np.array_equal(goff, woff)
        got.close()
    b = parse.parse(fq, parse.FASTQ, k).batch
    fc = km.FileCreator(ctx, l, k, "read", rc=False)
    for lo, hi in ((0, 2900), (2900, 3000)):
        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
        fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi]))
    got = fc.create()
    assert ctx.lib.kmg_db_payload_bits(got.h) == 64
    _same_db(got, OracleDB(fq, 1, k, l))
    got.close()


def test_many_batches_equal_one_batch(ctx):
    """addKmers may be called many times (FileCreator.java:65); read numbering carries across calls."""
    import kmers_amd as km
    from kmers_amd import parse
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(21)
    fq = _rand_fastq(rng, 500, 200, b"ACGTN")
    pf = parse.parse(fq, parse.FASTQ, 9)
    b = pf.batch
    fc = km.FileCreator(ctx, 3, 9, "read", rc=False)
    cuts = [0, 1, 2, 150, 151, 333, b.nseq]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
        fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi]))
    got = fc.create()
    _same_db(got, OracleDB(fq, 1, 9, 3))
    got.close()


def test_many_empty_reads_inside_one_tile(ctx):
    """70 000 empty FASTQ records between real reads: more sequence starts inside one 4096-base tile than the
    extractor's 16-bit per-position sequence counter holds (clamped there, looked up instead)."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(61, 2, 3000)
    reads = synth.random_reads(62, genomes, 120, 90, 0.01)
    recs = [synth.fastq_bytes(reads[i:i + 1], first_index=i) for i in range(reads.shape[0])]
    empties = b"".join(b"@e%d\n\n+\n\n" % i for i in range(70_000))
    fq = b"".join(recs[:40]) + empties + b"".join(recs[40:])
    for k, l in ((17, 4), (31, 6)):
        got, names = km.make_read_db(ctx, fq, k, l)
        want = OracleDB(fq, 1, k, l)
        _same_db(got, want)
        assert names == want.names and len(names) == 120
        got.close()


def test_later_batches_of_key_range_builds_with_empty_reads(ctx):
    """Later batches of a build are written from the count pass's tile image, by a workgroup per tile or -- where a key
    range keeps little -- by a wave per tile with a lane per record.  Both against the oracle's slice: three batches, a
    narrow and a wide key range, tiles in which thousands of empty reads start at one position (sequences looked up by
    search instead of by counting start bits), reads whose headers repeat across the batch cuts."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(63, 3, 20000, shared_frac=0.2)
    reads = synth.random_reads(64, genomes, 900, 120, 0.01, n_rate=0.003)
    recs = [synth.fastq_bytes(reads[i:i + 1], first_index=i) for i in range(reads.shape[0])]
    empties = b"".join(b"@e%d\n\n+\n\n" % i for i in range(5000))
    fq = b"".join(recs[:300]) + empties + b"".join(recs[300:650]) + empties.replace(b"@e", b"@f") + b"".join(recs[650:])
    k, l = 27, 4
    want = OracleDB(fq, 1, k, l, threads=4)
    b = parse.parse(fq, parse.FASTQ, k).batch
    cuts = [0, 280, 5400, b.nseq]
    nkeys = 4 ** l
    for first, end in ((0, nkeys), (7, 9), (9, 200), (200, nkeys)):
        fc = km.FileCreator(ctx, l, k, "read", rc=False, key_range=(first, end))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
            fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi]))
        got = fc.create()
        h = got.fetch()
        u0, u1 = int(want.bucket_off[first]), int(want.bucket_off[end])
        v0, v1 = int(want.val_off[u0]), int(want.val_off[u1])
        assert np.array_equal(h["keys"], want.keys[u0:u1])
        assert np.array_equal(h["val_off"].astype(np.int64), want.val_off[u0:u1 + 1].astype(np.int64) - v0)
        assert np.array_equal(h["val_a"], want.val_a[v0:v1]) and np.array_equal(h["val_b"], want.val_b[v0:v1])
        got.close()


def test_large_first_batch_then_more_batches(ctx):
    """The first batch of a build is written already partitioned by the sort's first digit; a second batch
    voids that plan (window-order append) and finish() must run every pass.  Both DB kinds, sizes well above
    one sort segment."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(55, 5, 30000, shared_frac=0.3)
    reads = synth.random_reads(56, genomes, 4000, 100, 0.01, n_rate=0.002)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    for data, mode, kind, rc, cuts in ((fa, parse.FASTA, "ref", True, [0, 3, 4, 5]), (fq, parse.FASTQ, "read", False, [0, 2500, 2501, 4000])):
        b = parse.parse(data, mode, 23).batch
        fc = km.FileCreator(ctx, 5, 23, kind, rc=rc)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
            fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                          seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                          seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi]))
        got = fc.create()
        _same_db(got, OracleDB(data, 0 if kind == "ref" else 1, 23, 5, threads=4))
        got.close()


@pytest.mark.parametrize("k,l,shared", [(31, 6, 26), (32, 6, 29), (21, 4, 19)])
def test_kmers_that_differ_only_in_their_last_bases(ctx, k, l, shared):
    """Every other read starts with the same `shared` bases, so thousands of DIFFERENT k-mers agree on all leading
    digits of the sort and only the lowest ones tell them apart (each of them twice: ties must keep their order) --
    next to ordinary reads.  Both kinds of database, single batch (fused path) and two batches."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(1000 + k)
    taxa, genomes = synth.random_genomes(61, 3, 4000)
    head = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, shared)]
    tails = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (2000, k + 6 - shared))]
    tails[1000:] = tails[:1000]                                  # every such k-mer twice: ties must keep their order
    same_start = np.concatenate([np.tile(head, (2000, 1)), tails], axis=1)
    plain = synth.random_reads(62, genomes, 1500, k + 6, 0.01)
    reads = np.concatenate([same_start, plain])[rng.permutation(3500)]
    fq = synth.fastq_bytes(reads)
    fa = b"".join(b">%d\n%s\n" % (70 + (i % 5), bytes(r)) for i, r in enumerate(reads))
    for data, mode, kind, rc in ((fq, parse.FASTQ, "read", False), (fa, parse.FASTA, "ref", True)):
        want = OracleDB(data, 0 if kind == "ref" else 1, k, l, threads=4)
        b = parse.parse(data, mode, k).batch
        for cuts in ([0, 3500], [0, 1700, 3500]):
            fc = km.FileCreator(ctx, l, k, kind, rc=rc)
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
                fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                              seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                              seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi]))
            got = fc.create()
            _same_db(got, want)
            got.close()


@pytest.mark.parametrize("k,l,cuts", [(21, 3, [0, 5, 6, 40, 64]), (32, 5, [0, 300, 1024]), (7, 2, [0, 1, 2, 2, 15, 16])])
def test_key_range_builds_are_slices_of_the_whole(ctx, k, l, cuts):
    """kmg_build_set_key_range: a build restricted to key buckets [first, end) holds exactly the whole database's
    k-mers of those buckets, values and all (an empty range, first == end, builds an empty database: a rank of a world
    larger than the populated buckets) -- forward and reverse-complement records are judged one by one (an RC
    usually lies in another range than its forward k-mer) -- and numbers its reads as the whole build does.  Laid end
    to end the range builds are the oracle's database.  Reference and read databases, one batch (extraction fused
    with the partition) and two batches (plain extraction), -D / -R filters on."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(200 + k, 4, 6000, shared_frac=0.3)
    reads = synth.random_reads(201 + k, genomes, 1500, 90, 0.01, n_rate=0.004)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    for data, mode, kind, rc, filt in ((fa, parse.FASTA, "ref", True, {}), (fq, parse.FASTQ, "read", False, {}),
                                       (fa, parse.FASTA, "ref", True, dict(dust=3, runofsame=6))):
        want = OracleDB(data, 0 if kind == "ref" else 1, k, l, threads=4, **filt)
        b = parse.parse(data, mode, k).batch
        for batches in (1, 2):
            keys, offs, va, vb, windows, ridx_all, vbase = [], [np.zeros(1, np.uint64)], [], [], 0, [], np.uint64(0)
            for first, end in zip(cuts[:-1], cuts[1:]):
                fc = km.FileCreator(ctx, l, k, kind, rc=rc, key_range=(first, end), **filt)
                half = b.nseq // 2 if batches == 2 else b.nseq
                ridx = []
                for lo, hi in ((0, half), (half, b.nseq)):
                    if hi > lo:
                        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
                        ridx.append(fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                                                  seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                                                  seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi])))
                part = fc.create()
                windows += fc.n_windows
                h = part.fetch()
                bucket = (h["keys"] >> np.uint64(64 - 2 * l)).astype(np.int64)
                assert np.all((bucket >= first) & (bucket < end))
                keys.append(h["keys"])
                offs.append(h["val_off"][1:] + vbase)
                vbase += h["val_off"][-1]
                va.append(h["val_a"]); vb.append(h["val_b"])
                if kind == "read":
                    ridx_all.append(np.concatenate(ridx))
                part.close()
            assert np.array_equal(np.concatenate(keys), want.keys)
            assert np.array_equal(np.concatenate(offs), want.val_off)
            assert np.array_equal(np.concatenate(va), want.val_a) and np.array_equal(np.concatenate(vb), want.val_b)
            if not rc:
                assert windows == want.n_windows
            for r in ridx_all[1:]:
                assert np.array_equal(r, ridx_all[0])
    fc = km.FileCreator(ctx, l, k, "ref", rc=True)
    with pytest.raises(km.InconsistentDataException):
        ctx.check(ctx.lib.kmg_build_set_key_range(ctx.h, 4, 3))
    with pytest.raises(km.InconsistentDataException):
        ctx.check(ctx.lib.kmg_build_set_key_range(ctx.h, 0, 4 ** l + 1))
    fc.create().close()


@pytest.mark.parametrize("k,l,seed", [(9, 2, 5), (8, 2, 6), (31, 6, 8)])
def test_inexact_search_matches_trie_restatement(ctx, k, l, seed):
    """Matcher -n 1 | 2 | 3 | 5 [-j] for k-mers of one length: kmg_match_near + the per-k-mer steps of api.near_match_lines
    against the reference's radix trie restated in oracle/slow_ref.py (Root, ClosestInfoCollector,
    processNearestCommonKey) -- itself checked against an all-pairs search in tests/test_oracle_kat.py.  Three taxa a
    few bases apart, reads with errors, one database file and a set of two (merged on the GPU / inserted file by
    file into the trie), even k with palindromes."""
    import kmers_amd as km
    from oracle import slow_ref as R
    from test_oracle_kat import _near_inputs
    fa1, fa2, fq = _near_inputs(seed, min(k, 12))
    if k == 31:                                                       # long k-mers: same genomes, longer reads
        fq = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, fa1.split(b"\n")[1][o:o + 60], b"I" * 60) for i, o in enumerate(range(0, 200, 7)))
        fq = fq.replace(b"ACG", b"ACT", 3)
    o_reads, _ = R.build_read_db(fq, k, l)
    o_files = [R.build_ref_db(fa1, k, l), R.build_ref_db(fa2, k, l)]
    reads, _ = km.make_read_db(ctx, fq, k, l)
    refs = [km.make_reference_db(ctx, fa1, k, l), km.make_reference_db(ctx, fa2, k, l)]
    for nfiles in (1, 2):
        db = km.DB(refs[:nfiles])
        for max_diff in (1, 2, 3) + ((5,) if k < 31 else ()):       # (beyond two: the walk over the sorted array)
            for just in (False, True):
                want = R.near_match_lines(o_reads, R.DB(o_files[:nfiles]), max_diff, just)
                assert km.near_match_lines(reads, db, max_diff, just) == want, (nfiles, max_diff, just)
                assert len(want) > 5
        off, idx, dist = km.match_near(reads, db, 0)                  # no difference allowed = the exact join
        hit, nh = db.get_nearest_kmers(reads)
        assert len(idx) == nh and np.array_equal(idx, hit[hit >= 0]) and not dist.any()
        assert np.array_equal(np.diff(off.astype(np.int64)), (hit >= 0).astype(np.int64))
        if nfiles == 2:
            db.merged.close()
    for d in refs + [reads]:
        d.close()


@pytest.mark.parametrize("seed", range(12))
def test_inexact_search_pairs_against_all_pairs_distances(ctx, seed):
    """kmg_match_near's raw output -- every (search k-mer, reference k-mer) pair within max_diff differences, ascending
    per search k-mer, with its distance -- against Hamming distances computed for ALL pairs (numpy, on the packed
    k-mers): random k and key length, related genomes, low-complexity stretches, reads with errors."""
    import kmers_amd as km
    from kmers_amd import synth
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(6, 33))
    l = int(rng.integers(1, min(k, 6) + 1))
    taxa, genomes = synth.random_genomes(300 + seed, 3, int(rng.integers(300, 2500)), shared_frac=0.5)
    if seed % 3 == 0:
        genomes[0][:200] = ord("A")                                   # poly-A: many neighbours of one another
        genomes[1][100:300] = np.frombuffer(b"AC" * 100, np.uint8)
    reads = synth.random_reads(400 + seed, genomes, int(rng.integers(20, 120)), max(k + 5, 60), 0.03)
    ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), k, l)
    rdb, _ = km.make_read_db(ctx, synth.fastq_bytes(reads), k, l)
    sk, rk = rdb.fetch()["keys"], ref.fetch()["keys"]
    x = sk[:, None] ^ rk[None, :]
    nz = (x | (x >> np.uint64(1))) & np.uint64(0x5555555555555555)    # one bit per differing base
    ham = np.zeros(nz.shape, np.int32)
    for sh in range(0, 64, 2):
        ham += ((nz >> np.uint64(sh)) & np.uint64(1)).astype(np.int32)
    for max_diff in (0, 1, 2, 3, 4, 7, k):
        off, idx, dist = km.match_near(rdb, ref, max_diff)
        rows, cols = np.nonzero(ham <= max_diff)                      # row-major: ascending reference index per row
        assert np.array_equal(np.diff(off.astype(np.int64)), np.bincount(rows, minlength=len(sk)))
        assert np.array_equal(idx, cols) and np.array_equal(dist, ham[rows, cols])
    ref.close()
    rdb.close()


def test_concurrent_callers_on_one_context(ctx):
    """The reference calls addKmers on ONE FileCreator from up to -t pool threads at once in -p mode
    (MakeDatabase.java:220-236,259-265) and quickMatchCommonKey from `threads` pool threads (DB.java:191-207), so a
    context has to take concurrent kmg_build_add / kmg_match / kmg_taxacounts calls (threading).  Eight
    threads add eight slices of a reference in whatever order they get there -- a reference database does not depend
    on it -- then eight threads join and count at once; everything equals the oracle's single-threaded answer."""
    import threading
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(57, 24, 6000, shared_frac=0.3)
    reads = synth.random_reads(58, genomes, 3000, 100, 0.01)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    b = parse.parse(fa, parse.FASTA, 21).batch
    fc = km.FileCreator(ctx, 5, 21, "ref", rc=True)
    errors = []

    def add(lo, hi):
        try:
            a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
            fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_id=b.seq_id[lo:hi]))
        except Exception as e:                      # noqa: BLE001 - reported by the main thread
            errors.append(e)

    th = [threading.Thread(target=add, args=(3 * i, 3 * i + 3)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    ref = fc.create()
    o_ref, o_reads = OracleDB(fa, 0, 21, 5, threads=4), OracleDB(fq, 1, 21, 5, threads=4)
    _same_db(ref, o_ref)
    rdb, _ = km.make_read_db(ctx, fq, 21, 5)
    o_hit, o_nh = o_reads.match(o_ref)
    o_t = o_ref.taxacounts()
    results = [None] * 8

    def query(i):
        try:
            hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
            results[i] = (hit, nh, km.taxa_counts(ref))
        except Exception as e:                      # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=query, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for hit, nh, (t_, m_, q_) in results:
        assert nh == o_nh and np.array_equal(hit, o_hit)
        assert np.array_equal(t_, o_t[0]) and np.array_equal(m_, o_t[1]) and np.array_equal(q_, o_t[2])
    ref.close()
    rdb.close()


# ---------------------------------------------------------------------------- match + counts parity
@pytest.mark.parametrize("k,key_len,seed", [(5, 2, 31), (9, 3, 32), (31, 6, 33)])
def test_match_and_taxacounts_match_oracle(ctx, k, key_len, seed):
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(seed, 6, 3000, first_taxon=900, shared_frac=0.3)
    reads = synth.random_reads(seed + 1, genomes, 400, 100, 0.02, n_rate=0.003)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, k, key_len)
    rdb, _ = km.make_read_db(ctx, fq, k, key_len)
    o_ref, o_reads = OracleDB(fa, 0, k, key_len), OracleDB(fq, 1, k, key_len)
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(o_ref)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    assert "\n".join(km.match_lines(rdb, ref, hit)) + "\n" == o_reads.match_lines(o_ref, o_hit).decode()
    t, m, q = km.taxa_counts(ref)
    ot, om, oq = o_ref.taxacounts()
    assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    # reference-vs-reference search (Matcher.java:182-210): every k-mer finds itself
    hit2, nh2 = km.DB([ref]).get_nearest_kmers(ref)
    assert nh2 == ref.n_unique and np.array_equal(hit2, np.arange(ref.n_unique))
    ref.close()
    rdb.close()


@pytest.mark.parametrize("n_reads,genome_len", [(40, 150_000), (6000, 3_000)])
def test_match_sparse_and_dense_search(ctx, n_reads, genome_len):
    """The join kernel stages the reference range of a search tile in LDS when it fits: few reads against a
    large reference (range >> stage: searched in place) and many reads against a small one (range << stage)."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(77, 3, genome_len, first_taxon=5)
    reads = synth.random_reads(78, genomes, n_reads, 120, 0.01)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, 21, 5)
    rdb, _ = km.make_read_db(ctx, fq, 21, 5)
    o_ref, o_reads = OracleDB(fa, 0, 21, 5, threads=8), OracleDB(fq, 1, 21, 5, threads=8)
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(o_ref, threads=8)
    assert nh == o_nh and nh > 0 and np.array_equal(hit, o_hit)
    # and the other way round (reference k-mers searched in the read set)
    rh = rdb.fetch()
    as_ref = km.KmerDatabase.load(ctx, 21, 5, True, 1, rh["keys"], rh["val_off"], rh["val_a"], rh["val_b"])
    hit2, nh2 = km.DB([as_ref]).get_nearest_kmers(ref)
    want = np.searchsorted(rh["keys"], ref.fetch()["keys"])
    found = (want < len(rh["keys"])) & (rh["keys"][np.minimum(want, len(rh["keys"]) - 1)] == ref.fetch()["keys"])
    assert nh2 == int(found.sum()) and np.array_equal(hit2, np.where(found, want, -1))
    for d in (ref, rdb, as_ref):
        d.close()


# ---------------------------------------------------------------------------- multi-file databases
def test_merge_of_reference_files_equals_one_build(ctx):
    """DB.kmers over name.1 .. name.N (DB.java:99-109): merging the databases of the parts of a FASTA gives
    the database of the whole FASTA -- counts of equal (k-mer, taxon) pairs add, taxa stay ascending."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(91, 6, 4000, first_taxon=-3, shared_frac=0.4)      # negative taxa too
    genomes = list(genomes) + [genomes[0], genomes[1][:1500]]                              # repeated sequences:
    taxa = list(taxa) + [taxa[0], taxa[3]]                                                 # counts > 1 and shared k-mers
    parts = [(taxa[:3], genomes[:3]), (taxa[3:5], genomes[3:5]), (taxa[5:], genomes[5:])]
    for k, l in ((9, 3), (31, 6)):
        dbs = [km.make_reference_db(ctx, synth.fasta_bytes(t, g), k, l) for t, g in parts]
        db = km.DB(dbs)
        want = OracleDB(synth.fasta_bytes(taxa, genomes), 0, k, l)
        _same_db(db.merged, want)
        t, m, q = km.taxa_counts(db)
        ot, om, oq = want.taxacounts()
        assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
        reads = synth.random_reads(92, genomes[:6], 300, 90, 0.01)
        rdb, _ = km.make_read_db(ctx, synth.fastq_bytes(reads), k, l)
        hit, nh = db.get_nearest_kmers(rdb)
        o_hit, o_nh = OracleDB(synth.fastq_bytes(reads), 1, k, l).match(want)
        assert nh == o_nh and np.array_equal(hit, o_hit)
        with pytest.raises(km.InconsistentDataException):                                   # DB.java:36-42
            km.DB([dbs[0], rdb])
        for d in dbs + [db.merged, rdb]:
            d.close()


def test_merge_of_read_files_unites_position_lists(ctx):
    """Read DBs of two halves of a FASTQ, second half's read indices shifted as the multi-process path does
    (dist.global_read_indices): the merge holds every (read, pos) of the whole file, ascending."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(93, 2, 2500)
    reads = synth.random_reads(94, genomes, 400, 70, 0.01)
    k, l = 15, 4
    a, na = km.make_read_db(ctx, synth.fastq_bytes(reads[:170]), k, l)
    b, nb = km.make_read_db(ctx, synth.fastq_bytes(reads[170:], first_index=170), k, l)
    hb = b.fetch()
    shifted = km.KmerDatabase.load(ctx, k, l, False, 1, hb["keys"], hb["val_off"], hb["val_a"] + len(na), hb["val_b"])
    whole = OracleDB(synth.fastq_bytes(reads), 1, k, l)
    assert na + nb == whole.names
    merged = km.merge_databases([a, shifted])
    _same_db(merged, whole)
    for d in (a, b, shifted, merged):
        d.close()


def test_key_range_shards_merge_on_the_gpu(ctx):
    """The local step of the multi-GPU key-range exchange (kmers_amd/dist.py): CSR slices of two ranks' local
    databases, cut at the balanced range bounds, loaded and merged with kmg_db_merge == the whole database's
    k-mers of that range.  (The collectives themselves are covered on gloo in tests/test_dist_gloo.py.)"""
    import kmers_amd as km
    from kmers_amd import dist as kd, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(97, 4, 6000, shared_frac=0.5)
    k, l, world = 21, 4, 3
    halves = [km.make_reference_db(ctx, synth.fasta_bytes(taxa[:2], genomes[:2]), k, l),
              km.make_reference_db(ctx, synth.fasta_bytes(taxa[2:], genomes[2:]), k, l)]
    whole = OracleDB(synth.fasta_bytes(taxa, genomes), 0, k, l)
    hist = np.diff(whole.val_off.astype(np.int64)[whole.bucket_off.astype(np.int64)])
    ranges = kd.balanced_key_ranges(hist, world)
    assert kd.build_key_range_shard(ctx, halves[0], [(0, 4 ** l)]).n_unique == halves[0].n_unique   # world of one
    for g in range(world):
        parts = []
        for h in halves:
            f = h.fetch()
            c = kd.cut_points(f["keys"], ranges, l)
            vo = f["val_off"].astype(np.int64)
            a, b = int(vo[c[g]]), int(vo[c[g + 1]])
            parts.append(km.KmerDatabase.load(ctx, k, l, True, 0, f["keys"][c[g]:c[g + 1]],
                                              (vo[c[g]:c[g + 1] + 1] - a).astype(np.uint64), f["val_a"][a:b], f["val_b"][a:b]))
        shard = km.merge_databases(parts)
        u0, u1 = int(whole.bucket_off[ranges[g][0]]), int(whole.bucket_off[ranges[g][1]])
        v0, v1 = int(whole.val_off[u0]), int(whole.val_off[u1])
        s = shard.fetch()
        assert np.array_equal(s["keys"], whole.keys[u0:u1])
        assert np.array_equal(s["val_a"], whole.val_a[v0:v1]) and np.array_equal(s["val_b"], whole.val_b[v0:v1])
        for d in parts + [shard]:
            d.close()
        # the same slice the device-resident way: tensors on the GPU, cut at the stored bucket offsets, CSR rebuilt
        # by exchange_csr_tensors (a world of one here: the slice comes straight back), kmg_db_load_dev
        parts = []
        for h in halves:
            boff = h.bucket_offsets()
            cut = [int(boff[ranges[g][0]]), int(boff[ranges[g][1]])]
            (p,) = kd.exchange_csr_tensors(*h.to_tensors(), cut)
            torch.cuda.synchronize()
            parts.append(km.KmerDatabase.load_tensors(ctx, k, l, True, 0, *p))
        shard = km.merge_databases(parts)
        s2 = shard.fetch()
        for name in ("keys", "val_off", "val_a", "val_b", "bucket_off"):
            assert np.array_equal(s2[name], s[name]), name
        for d in parts + [shard]:
            d.close()
    for h in halves:
        h.close()


def test_key_range_sharded_match_on_the_gpu(ctx):
    """Device-resident sharded join (dist.match_key_range_shard and its pieces): the read k-mers cut at the range
    bounds, each slice joined against that range's reference shard, give the hits of the whole-database join."""
    import kmers_amd as km
    from kmers_amd import dist as kd, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(51, 4, 5000, shared_frac=0.3)
    reads = synth.random_reads(52, genomes, 500, 90, 0.01)
    k, l, world = 19, 4, 3
    ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), k, l)
    rdb, _ = km.make_read_db(ctx, synth.fastq_bytes(reads), k, l)
    o_ref, o_reads = OracleDB(synth.fasta_bytes(taxa, genomes), 0, k, l), OracleDB(synth.fastq_bytes(reads), 1, k, l)
    o_hit, o_nh = o_reads.match(o_ref)
    assert o_nh > 0
    # a world of one: the whole thing through the public entry
    owner, hit = kd.match_key_range_shard(ctx, rdb, ref, [(0, 4 ** l)])
    assert np.array_equal(hit.cpu().numpy(), o_hit) and int(owner.max()) == 0
    # three ranges, every owner's part by hand
    hist = np.diff(o_ref.val_off.astype(np.int64)[o_ref.bucket_off.astype(np.int64)])
    ranges = kd.balanced_key_ranges(hist, world)
    keys = rdb.to_tensors()[0]
    sb, rb = rdb.bucket_offsets(), ref.bucket_offsets()
    got = np.full(rdb.n_unique, -2, np.int64)
    for g in range(world):
        (p,) = kd.exchange_csr_tensors(*ref.to_tensors(), [int(rb[ranges[g][0]]), int(rb[ranges[g][1]])])
        torch.cuda.synchronize()
        shard = km.KmerDatabase.load_tensors(ctx, k, l, True, 0, *p)
        c0, c1 = int(sb[ranges[g][0]]), int(sb[ranges[g][1]])
        _, h = kd.match_by_key_range_tensors(keys, [c0, c1], kd.shard_matcher(ctx, rdb, shard))
        h = h.cpu().numpy()
        got[c0:c1] = np.where(h >= 0, h + int(rb[ranges[g][0]]), -1)
        shard.close()
    assert np.array_equal(got, o_hit)
    ref.close()
    rdb.close()


def test_match_keys_in_runs(ctx):
    """kmg_match_keys_dev: the owner's side of the key-range exchange -- several ascending runs of search k-mers back
    to back (one per source GPU, all inside one key range, an empty one among them, run lengths on both sides of a
    tile) joined in one launch; every element's hit index equals the oracle's for that k-mer.  Also the one-run
    form against a whole database, and the argument checks."""
    import ctypes as C
    import kmers_amd as km
    from kmers_amd import _lib, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(61, 3, 40000, shared_frac=0.2)
    reads = synth.random_reads(62, genomes, 3000, 120, 0.01)
    k, l = 23, 5
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, k, l)
    o_ref, o_reads = OracleDB(fa, 0, k, l), OracleDB(fq, 1, k, l)
    o_hit, _ = o_reads.match(o_ref)
    rng = np.random.default_rng(5)
    n = o_reads.U
    assert n > 5 * 2048
    # deal the (ascending) search k-mers to 5 "sources": each source's share stays ascending
    src = rng.integers(0, 5, size=n)
    src[src == 3] = 4                                  # source 3 sends nothing
    src[:3] = [0, 1, 2]
    order = np.concatenate([np.nonzero(src == g)[0] for g in range(5)])
    counts = [int((src == g).sum()) for g in range(5)]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    dev = torch.device("cuda", 0)
    d_keys = torch.from_numpy(o_reads.keys[order].view(np.int64).copy()).to(dev)
    d_hit = torch.full((n + 1,), -7, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr(), off.ctypes.data_as(C.c_void_p), 5, k, ref.h, d_hit.data_ptr()))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    got = d_hit.cpu().numpy()
    assert got[n] == -7 and np.array_equal(got[:n], o_hit[order])
    # one run = the plain join
    one = np.array([0, n], np.uint64)
    d_keys = torch.from_numpy(o_reads.keys.view(np.int64).copy()).to(dev)
    d_hit.fill_(-7)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr(), one.ctypes.data_as(C.c_void_p), 1, k, ref.h, d_hit.data_ptr()))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    assert np.array_equal(d_hit.cpu().numpy()[:n], o_hit)
    bad = np.array([5, 0], np.uint64)
    assert ctx.lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr(), bad.ctypes.data_as(C.c_void_p), 1, k, ref.h, d_hit.data_ptr()) == _lib.KMG_EINVAL
    assert ctx.lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr(), one.ctypes.data_as(C.c_void_p), 1, k - 1, ref.h, d_hit.data_ptr()) == _lib.KMG_EINVAL
    assert ctx.lib.kmg_match_keys_dev(ctx.h, d_keys.data_ptr(), one.ctypes.data_as(C.c_void_p), 65, k, ref.h, d_hit.data_ptr()) == _lib.KMG_EINVAL
    ref.close()


def test_device_resident_copy_and_load(ctx):
    """kmg_db_copy_dev / kmg_db_load_dev: a database copied into torch tensors and loaded back is the same
    database; slices are bounds-checked; the device-side checks reject what kmg_db_load rejects on the host."""
    import kmers_amd as km
    from kmers_amd import synth
    taxa, genomes = synth.random_genomes(43, 3, 3000)
    reads = synth.random_reads(44, genomes, 300, 80, 0.01)
    for db in (km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), 17, 5),
               km.make_read_db(ctx, synth.fastq_bytes(reads), 17, 5)[0]):
        keys, val_off, val_a, val_b = db.to_tensors()
        h = db.fetch()
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), h["keys"])
        assert np.array_equal(val_off.cpu().numpy().view(np.uint64), h["val_off"])
        assert np.array_equal(val_a.cpu().numpy(), h["val_a"]) and np.array_equal(val_b.cpu().numpy(), h["val_b"])
        again = km.KmerDatabase.load_tensors(ctx, db.k, db.key_length, db.rc, db.kind, keys, val_off, val_a, val_b)
        g = again.fetch()
        for name in ("keys", "val_off", "val_a", "val_b", "bucket_off"):
            assert np.array_equal(g[name], h[name]), name
        again.close()
        # a slice
        part = torch.empty(10, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.check(ctx.lib.kmg_db_copy_dev(db.h, 0, part.data_ptr(), 5, 10))
        assert np.array_equal(part.cpu().numpy().view(np.uint64), h["keys"][5:15])
        with pytest.raises(km.InconsistentDataException):
            ctx.check(ctx.lib.kmg_db_copy_dev(db.h, 0, part.data_ptr(), db.n_unique - 5, 10))
        with pytest.raises(km.InconsistentDataException):
            ctx.check(ctx.lib.kmg_db_copy_dev(db.h, 9, part.data_ptr(), 0, 1))
        # descending keys / val_off not ending at n_values
        bad = keys.clone()
        bad[[3, 4]] = bad[[4, 3]]
        torch.cuda.synchronize()
        with pytest.raises(km.InconsistentDataException):
            km.KmerDatabase.load_tensors(ctx, db.k, db.key_length, db.rc, db.kind, bad, val_off, val_a, val_b)
        bad_off = val_off.clone()
        bad_off[-1] -= 1
        torch.cuda.synchronize()
        with pytest.raises(km.InconsistentDataException):
            km.KmerDatabase.load_tensors(ctx, db.k, db.key_length, db.rc, db.kind, keys, bad_off, val_a, val_b)
        db.close()
    # an empty database
    empty = km.KmerDatabase.load_tensors(ctx, 17, 5, True, 0, torch.empty(0, dtype=torch.int64, device="cuda"),
                                         torch.zeros(1, dtype=torch.int64, device="cuda"),
                                         torch.empty(0, dtype=torch.int32, device="cuda"),
                                         torch.empty(0, dtype=torch.int32, device="cuda"))
    assert empty.n_unique == 0 and empty.n_values == 0
    empty.close()


def test_db_load_roundtrip_and_errors(ctx):
    import kmers_amd as km
    from kmers_amd import synth
    taxa, genomes = synth.random_genomes(41, 3, 2000)
    ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), 15, 4)
    h = ref.fetch()
    again = km.KmerDatabase.load(ctx, 15, 4, True, 0, h["keys"], h["val_off"], h["val_a"], h["val_b"])
    g = again.fetch()
    for name in ("keys", "val_off", "val_a", "val_b", "bucket_off"):
        assert np.array_equal(g[name], h[name])
    other = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), 14, 4)
    with pytest.raises(km.InconsistentDataException):       # DB.java:78-81
        km.DB([ref]).get_nearest_kmers(other)
    with pytest.raises(km.KmersGpuError):                   # k > 64 -> shim falls back to Java
        km.FileCreator(ctx, 6, 65, "ref", True)
    for d in (ref, again, other):
        d.close()


# ---------------------------------------------------------------------------- full-size properties
def test_cfg1_scale_properties(ctx):
    """BASELINE config 1/2 at full size (5 Mbp, 1M x 150 bp, k=31): size-independent properties, plus exact
    equality of both databases, the read-name map and every hit index with the C oracle (multi-threaded)."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes, reads = synth.cfg1(1.0)
    fa = synth.fasta_bytes(taxa, genomes)
    ref = km.make_reference_db(ctx, fa, 31, 6)
    h = ref.fetch()
    keys = h["keys"]
    assert np.all(keys[1:] > keys[:-1])                                  # strictly ascending, unique
    assert int(h["val_b"].sum()) == 2 * ref.n_windows                    # odd k: every window + its RC
    assert ref.n_windows == sum(len(g) - 30 for g in genomes)
    # the stored set is closed under reverse complement: re-running the join on itself finds all
    hit, nh = km.DB([ref]).get_nearest_kmers(ref)
    assert nh == ref.n_unique
    want = OracleDB(fa, 0, 31, 6, threads=8)
    _same_db(ref, want)
    nreads = reads.shape[0]                                              # all 1M reads: 120 M records
    fq = synth.fastq_bytes(reads[:nreads])
    rdb, names = km.make_read_db(ctx, fq, 31, 6)
    assert rdb.n_values == nreads * 120 == rdb.n_windows
    o_reads = OracleDB(fq, 1, 31, 6, threads=16)
    _same_db(rdb, o_reads)
    assert names == o_reads.names and len(names) == nreads
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(want, threads=16)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    assert 0.1 < nh / rdb.n_unique < 0.9                                 # error k-mers are mostly unique misses
    t, m, q = km.taxa_counts(ref)
    ot, om, oq = want.taxacounts()
    assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    # per-read aggregation at full size, against numpy on the oracle's arrays: the transposed join ...
    read, pos, idx = km.collect_by_read(rdb, ref)
    sv = o_reads.val_off.astype(np.int64)
    per_kmer = np.where(o_hit >= 0, sv[1:] - sv[:-1], 0)
    assert len(read) == int(per_kmer.sum())
    okey = np.repeat(np.arange(len(o_hit)), sv[1:] - sv[:-1])                      # search k-mer of every occurrence
    keep = o_hit[okey] >= 0
    want_rp = o_reads.val_a.astype(np.int64)[keep] * 65536 + o_reads.val_b.astype(np.int64)[keep]
    order = np.argsort(want_rp, kind="stable")
    assert np.array_equal(read.astype(np.int64) * 65536 + pos, want_rp[order])
    assert np.array_equal(idx, o_hit[okey][keep][order])
    # ... and the classifier: every taxon set here is a singleton or a pair, so processRead in numpy is short
    rv = want.val_off.astype(np.int64)
    nt = (rv[1:] - rv[:-1])[idx]
    first = want.val_a[rv[idx]]
    single = nt == 1
    mn = np.full(nreads, np.iinfo(np.int32).max, np.int64)
    mx = np.full(nreads, np.iinfo(np.int32).min, np.int64)
    np.minimum.at(mn, read[single], first[single])
    np.maximum.at(mx, read[single], first[single])
    cand = np.where(mn == mx, mn, -1)
    assert int(nt.max()) <= 2
    second = want.val_a[np.minimum(rv[idx] + 1, len(want.val_a) - 1)]
    has = (first == cand[read]) | ((nt == 2) & (second == cand[read]))
    bad = np.zeros(nreads, bool)
    bad[read[~has]] = True
    want_call = np.where((cand >= 0) & ~bad, cand, -1)
    call = km.classify_reads(rdb, ref, nreads)
    assert np.array_equal(call, want_call) and (call >= 0).sum() > 0.9 * nreads
    ref.close()
    rdb.close()


# ---------------------------------------------------------------------------- skew (config 5 flavour)
def _skewed_reads(rng, n_reads, read_len):
    """Reads with homopolymer / dinucleotide tracts: a few key prefixes hold most of the k-mers."""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = acgt[rng.integers(0, 4, size=(n_reads, read_len))]
    for i in range(n_reads):
        r = rng.random()
        if r < 0.25:
            out[i, :] = ord("A")                                   # poly-A: every window is the same k-mer
        elif r < 0.40:
            out[i, ::2], out[i, 1::2] = ord("A"), ord("C")         # (AC)n
        elif r < 0.55:
            s = int(rng.integers(0, read_len // 2))
            out[i, s:s + read_len // 2] = ord("T")                 # poly-T tract inside a random read
    return out


@pytest.mark.parametrize("k,n_reads", [(21, 3000), (31, 20000)])
def test_skewed_reads_match_oracle(ctx, k, n_reads):
    """Oversize sort segments (some sorted slice by slice, or the whole-array LSD path) stay bit-exact."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(k)
    reads = _skewed_reads(rng, n_reads, 120)
    fq = synth.fastq_bytes(reads)
    got, names = km.make_read_db(ctx, fq, k, 6)
    want = OracleDB(fq, 1, k, 6, threads=8)
    _same_db(got, want)
    assert names == want.names
    got.close()
    # the same sequences as a reference (taxon payload, both strands)
    fa = b"".join(b">%d\n%s\n" % (100 + i % 7, reads[i].tobytes()) for i in range(min(n_reads, 4000)))
    ref = km.make_reference_db(ctx, fa, k, 6)
    _same_db(ref, OracleDB(fa, 0, k, 6, threads=8))
    ref.close()


# ---------------------------------------------------------------------------- -D / -R filters
@pytest.mark.parametrize("k,dust,run", [(9, 2, None), (9, None, 4), (12, 1, 3), (31, 14, 5), (31, None, 2), (5, 0, 1),
                                        (32, 3, 32), (4, 1, None)])
def test_dust_and_runofsame_filters_match_oracle(ctx, k, dust, run):
    """MakeDatabase -D / -R fused into the extraction kernel (Dust.java:12-30, RunOfSame.java:12-36)."""
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(200 + k)
    fa = _rand_fasta(rng, 30, 600, b"AAAACGT", (3, 4, 9))
    fq = _rand_fastq(rng, 200, 200, b"AAACCGTN")
    od, orr = (-1 if dust is None else dust), (-1 if run is None else run)
    ref = km.make_reference_db(ctx, fa, k, 3, dust=dust, runofsame=run)
    want = OracleDB(fa, 0, k, 3, dust=od, runofsame=orr)
    _same_db(ref, want)
    assert ref.n_windows == want.n_windows
    rdb, names = km.make_read_db(ctx, fq, k, 3, dust=dust, runofsame=run)
    wr = OracleDB(fq, 1, k, 3, dust=od, runofsame=orr)
    _same_db(rdb, wr)
    assert names == wr.names          # reads are numbered before the filters
    ref.close()
    rdb.close()


# ---------------------------------------------------------------------------- randomised configurations
# KMERS_FUZZ_SEEDS=N widens the sweep (a one-off 3000-seed run is recorded in DESIGN.md)
@pytest.mark.parametrize("seed", range(int(os.environ.get("KMERS_FUZZ_SEEDS", "24"))))
def test_fuzz_against_oracle(ctx, seed):
    """Random k, key length, input shape, batch cuts and filters -- sized so that the sort plans 1..12 prefix
    bits (fused first digit of 1..8 bits, index local sort on and off, segments near their cap)."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(8, 33))
    l = int(rng.integers(1, min(k, 6) + 1))
    n_gen, glen = int(rng.integers(1, 6)), int(rng.integers(500, 40000))
    taxa, genomes = synth.random_genomes(seed, n_gen, glen, first_taxon=int(rng.integers(-5, 1000)),
                                         shared_frac=float(rng.choice([0.0, 0.3, 0.9])))
    if rng.random() < 0.3:                                              # low complexity: skewed prefixes
        genomes[0][: glen // 2] = ord("A")
    n_reads, rlen = int(rng.integers(50, 6000)), int(rng.integers(k, 260))
    reads = synth.random_reads(seed + 1, genomes, n_reads, rlen, 0.02, n_rate=float(rng.choice([0.0, 0.01])))
    if rng.random() < 0.5:
        reads[rng.integers(0, n_reads, size=n_reads // 10)] |= 0x20     # lower case
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    dust = int(rng.integers(1, 4)) if rng.random() < 0.25 else None
    run = int(rng.integers(3, 9)) if rng.random() < 0.25 else None
    for data, mode, kind, rc in ((fa, parse.FASTA, "ref", True), (fq, parse.FASTQ, "read", False)):
        b = parse.parse(data, mode, k).batch
        ncut = int(rng.integers(0, 3))
        cuts = [0] + sorted(int(x) for x in rng.integers(0, b.nseq + 1, size=ncut)) + [b.nseq]
        fc = km.FileCreator(ctx, l, k, kind, rc=rc, dust=dust, runofsame=run)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
            fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                          seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                          seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi]))
        got = fc.create()
        want = OracleDB(data, 0 if kind == "ref" else 1, k, l, threads=4, dust=-1 if dust is None else dust,
                        runofsame=-1 if run is None else run)
        _same_db(got, want)
        if seed % 3 == 0:
            # the same database once more, in two key ranges cut at a random bucket (kmg_build_set_key_range)
            cut_at = int(rng.integers(0, 4 ** l + 1))
            pk, pa, pb, nvals = [], [], [], 0
            for first, end in ((0, cut_at), (cut_at, 4 ** l)):
                if first == end:
                    continue
                fc = km.FileCreator(ctx, l, k, kind, rc=rc, dust=dust, runofsame=run, key_range=(first, end))
                for lo, hi in zip(cuts[:-1], cuts[1:]):
                    a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
                    fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                                  seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                                  seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi]))
                part = fc.create()
                h = part.fetch()
                pk.append(h["keys"]); pa.append(h["val_a"]); pb.append(h["val_b"])
                nvals += part.n_values
                part.close()
            assert np.array_equal(np.concatenate(pk), want.keys) and nvals == want.V
            assert np.array_equal(np.concatenate(pa), want.val_a) and np.array_equal(np.concatenate(pb), want.val_b)
        if kind == "ref":
            ref, o_ref = got, want
        else:
            hit, nh = km.DB([ref]).get_nearest_kmers(got)
            o_hit, o_nh = want.match(o_ref)
            assert nh == o_nh and np.array_equal(hit, o_hit)
            got.close()
    ref.close()


@pytest.mark.parametrize("k,l,n_reads,rlen", [(32, 6, 120_000, 150), (24, 5, 60_000, 250), (17, 4, 150_000, 60), (31, 6, 30_000, 1000)])
def test_medium_builds_against_oracle(ctx, k, l, n_reads, rlen):
    """5-30 M records: 11-14 prefix bits, the one-word local sort with its k-mer-run epilogue at k = 32 (no spare
    low key bits) and below, reads longer than a tile's share."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(k, 4, 200_000, shared_frac=0.2)
    reads = synth.random_reads(k + 1, genomes, n_reads, rlen, 0.01, n_rate=0.001)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, k, l)
    rdb, names = km.make_read_db(ctx, fq, k, l)
    o_ref, o_reads = OracleDB(fa, 0, k, l, threads=16), OracleDB(fq, 1, k, l, threads=16)
    _same_db(ref, o_ref)
    _same_db(rdb, o_reads)
    assert names == o_reads.names
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(o_ref, threads=16)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    ref.close()
    rdb.close()


def _classifier_blocks(call, collect_blocks):
    """ReadClassifier's output blocks from the dense calls: the reads of each CollectByRead block, in order."""
    out = {}
    for key, text in collect_blocks.items():
        reads = []
        for line in (text.decode().split("\n") if text else []):
            r = int(line.split(":", 1)[0])
            if not reads or reads[-1] != r:
                reads.append(r)
        out[key] = "".join("%d\t%d\n" % (r, int(call[r])) for r in reads if call[r] >= 0).encode()
    return out


@pytest.mark.parametrize("k,l,n_reads,rlen,paired", [(21, 4, 2500, 90, False), (31, 6, 6000, 150, False), (15, 3, 1800, 70, True)])
def test_collect_by_read_and_classifier_match_oracle(ctx, k, l, n_reads, rlen, paired):
    """kmg_collect_by_read / kmg_classify_reads against the restated CollectByRead + ReadClassifier run on the match
    lines of the same join.  Genomes share 40 % of their sequence (multi-taxon hits), reads carry errors, and in
    the `paired` case neighbouring records share a header (one read index, (read, pos) pairs occurring twice)."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle import slow_ref as R
    taxa, genomes = synth.random_genomes(81 + k, 6, 12000, shared_frac=0.4)
    reads = synth.random_reads(82 + k, genomes, n_reads, rlen, 0.01)
    fq = synth.fastq_bytes(reads)
    if paired:
        recs = fq.split(b"@")[1:]
        fq = b"".join(b"@" + (recs[i - (i & 1)].split(b"\n", 1)[0]) + b"\n" + recs[i].split(b"\n", 1)[1] for i in range(len(recs)))
    ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), k, l)
    rdb, names = km.make_read_db(ctx, fq, k, l)
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    assert nh > 1000
    lines = km.match_lines(rdb, ref, hit)
    want = R.collect_by_read(lines)
    got = km.collect_by_read_blocks(rdb, ref)
    assert got.keys() == want.keys()
    for key in want:
        assert got[key] == want[key], key
    read, pos, idx = km.collect_by_read(rdb, ref)
    assert len(read) == int(rdb.fetch()["val_off"][np.nonzero(hit >= 0)[0] + 1].astype(np.int64).sum()
                            - rdb.fetch()["val_off"][np.nonzero(hit >= 0)[0]].astype(np.int64).sum())
    order = read.astype(np.int64) * 65536 + (pos.astype(np.int64) + 32768)
    assert np.all(np.diff(order) >= 0) and (paired or np.all(np.diff(order) > 0))
    call = km.classify_reads(rdb, ref, len(names))
    assert _classifier_blocks(call, want) == R.read_classifier(want)
    assert (call >= 0).sum() > 100 and (call < 0).sum() > 0
    # kmg_classify_reads_hits: the same calls + which reads have a matched k-mer = ReadClassifier's file without the
    # match text and CollectByRead's file in between
    call2, seen = km.classify_reads(rdb, ref, len(names), with_hits=True)
    assert np.array_equal(call2, call)
    assert np.array_equal(np.nonzero(seen)[0], np.unique(read))
    assert km.read_call_blocks(call2, seen) == R.read_classifier(want)
    with pytest.raises(km.InconsistentDataException):
        km.classify_reads(rdb, ref, len(names) - 1)                  # a read index outside n_reads
    with pytest.raises(km.InconsistentDataException):
        km.collect_by_read(ref, ref)                                 # not a read database
    # kmg_classify_reads_rows: the calls from answers found elsewhere -- here the exact join's own hit list against the
    # reference's own values, which must give kmg_classify_reads_hits' result; an index outside the rows is refused
    hit_all, _ = km.DB([ref]).get_nearest_kmers(rdb)
    call3, seen3 = km.classify_reads_rows(rdb, ref, hit_all, len(names))
    assert np.array_equal(call3, call) and np.array_equal(seen3, seen)
    none3, seen_none = km.classify_reads_rows(rdb, ref, np.full(rdb.n_unique, -1, np.int64), len(names))
    assert np.all(none3 == -1) and not seen_none.any()
    bad_hit = hit_all.copy()
    bad_hit[len(bad_hit) // 2] = ref.n_unique
    with pytest.raises(km.InconsistentDataException):
        km.classify_reads_rows(rdb, ref, bad_hit, len(names))
    ref.close()
    rdb.close()


def test_collect_by_read_skew_and_empty(ctx):
    """One k-mer that occurs in every read (a shared adapter) -- a value list of thousands of entries handled by one
    wave -- and the empty join."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle import slow_ref as R
    taxa, genomes = synth.random_genomes(91, 3, 5000)
    adapter = genomes[0][:40]
    reads = synth.random_reads(92, genomes, 3000, 60, 0.0)
    reads = np.concatenate([np.tile(adapter, (reads.shape[0], 1)), reads], axis=1)
    k, l = 17, 4
    ref = km.make_reference_db(ctx, synth.fasta_bytes(taxa, genomes), k, l)
    rdb, names = km.make_read_db(ctx, synth.fastq_bytes(reads), k, l)
    hit, _ = km.DB([ref]).get_nearest_kmers(rdb)
    want = R.collect_by_read(km.match_lines(rdb, ref, hit))
    assert km.collect_by_read_blocks(rdb, ref) == want
    assert _classifier_blocks(km.classify_reads(rdb, ref, len(names)), want) == R.read_classifier(want)
    other_taxa, other = synth.random_genomes(93, 1, 4000)
    stranger, _ = km.make_read_db(ctx, synth.fastq_bytes(synth.random_reads(94, other, 50, 60, 0.0)), k, l)
    r, p, i = km.collect_by_read(stranger, ref)
    assert len(r) == 0 and km.collect_by_read_blocks(stranger, ref) == {}
    assert np.all(km.classify_reads(stranger, ref, 50) == -1)
    c, s = km.classify_reads(stranger, ref, 50, with_hits=True)
    assert np.all(c == -1) and not s.any() and km.read_call_blocks(c, s) == {}
    for d in (ref, rdb, stranger):
        d.close()


# ---------------------------------------------------------------------------- BASELINE configs 3 and 5, scaled down
def test_cfg3_flavour_many_taxa(ctx):
    """Config 3 in small: 3000 taxa x 2 kbp, 20 % of each genome copied from its sibling (multi-taxon k-mers,
    unique < match), more taxa than the LDS histogram of TaxaCounts holds; reads in several batches; the whole chain
    (both databases, match, TaxaCounts, per-read calls) against the C oracle."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    taxa, genomes = synth.random_genomes(3, 3000, 2000, first_taxon=5, shared_frac=0.2)
    reads = synth.random_reads(4, genomes, 30000, 150, 0.01)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    k, l = 31, 6
    ref = km.make_reference_db(ctx, fa, k, l)
    o_ref = OracleDB(fa, 0, k, l, threads=8)
    _same_db(ref, o_ref)
    b = parse.parse(fq, parse.FASTQ, k).batch
    fc = km.FileCreator(ctx, l, k, "read", rc=False)
    for lo, hi in ((0, 12000), (12000, 12001), (12001, 30000)):                       # streamed in chunks
        a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
        fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo], seq_tok=b.seq_tok[lo:hi]))
    rdb = fc.create()
    o_reads = OracleDB(fq, 1, k, l, threads=8)
    _same_db(rdb, o_reads)
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(o_ref)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    t, m, q = km.taxa_counts(ref)
    ot, om, oq = o_ref.taxacounts()
    assert len(t) == 3000 and np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    assert (q < m).sum() > 2000                                                        # shared k-mers everywhere
    call = km.classify_reads(rdb, ref, reads.shape[0])
    assert (call >= 0).sum() > 0.5 * reads.shape[0] and (call < 0).sum() > 0
    ref.close()
    rdb.close()


def test_taxacounts_many_taxa_and_the_whole_int32_range(ctx):
    """TaxaCounts beyond the LDS histograms (20 000 taxa: global counters) and with taxon ids from one end of int32
    to the other, negative ones included (the presence bitmap then spans 2^32 bits) -- rows ascending by taxon as
    TaxaCounts.java:132-150 prints them, against the oracle."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(77)
    _, genomes = synth.random_genomes(78, 20000, 120, shared_frac=0.3)
    for ids in (np.arange(20000) * 3 + 7,
                np.concatenate([[-2 ** 31, -5, 0, 2 ** 31 - 1], rng.integers(-2 ** 31, 2 ** 31, size=19996)])):
        taxa = [int(x) for x in ids]
        fa = synth.fasta_bytes(taxa, genomes)
        ref = km.make_reference_db(ctx, fa, 31, 6)
        o_ref = OracleDB(fa, 0, 31, 6, threads=8)
        t, m, q = km.taxa_counts(ref)
        ot, om, oq = o_ref.taxacounts()
        assert len(t) == len(set(taxa)) and np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
        assert int(m.sum()) == ref.n_values and (q <= m).all() and (q < m).any()
        ref.close()


@pytest.mark.parametrize("n_taxa,stride", [(40, 1), (40, 100003), (3000, 5), (3000, 700001), (2049, 1), (16384, 1)])
def test_taxacounts_row_addressing_modes(ctx, n_taxa, stride):
    """TaxaCounts' LDS rows are addressed by taxon - min when the span of the ids fits them and by rank otherwise:
    dense and sparse ids on both histogram sizes, and the sizes' edges, against the oracle."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    _, genomes = synth.random_genomes(500 + n_taxa, n_taxa, 150, shared_frac=0.3)
    taxa = [-17 + i * stride for i in range(n_taxa)]
    fa = synth.fasta_bytes(taxa, genomes)
    ref = km.make_reference_db(ctx, fa, 25, 5)
    o_ref = OracleDB(fa, 0, 25, 5, threads=8)
    t, m, q = km.taxa_counts(ref)
    ot, om, oq = o_ref.taxacounts()
    assert len(t) == n_taxa and np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    assert int(m.sum()) == ref.n_values and (q < m).any()
    ref.close()


@pytest.mark.parametrize("k", [21, 25, 31])
def test_cfg5_flavour_long_noisy_reads(ctx, k):
    """Config 5 in small: 10 kbp reads with 15 % substitutions and 0.5 % N, genome sizes spread over two orders of
    magnitude, 5 % of the bases in homopolymer / dinucleotide tracts (bucket skew), k in {21, 25, 31}."""
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(700 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    sizes = [10_000, 14_000, 25_000, 60_000, 150_000, 400_000]                          # Zipf-like spread
    genomes = []
    for n in sizes:
        g = acgt[rng.integers(0, 4, size=n)]
        for _ in range(max(1, n // 4000)):                                             # low-complexity tracts, ~5 %
            s, ln = int(rng.integers(0, n - 250)), int(rng.integers(100, 250))
            if rng.random() < 0.5:
                g[s:s + ln] = acgt[int(rng.integers(0, 4))]
            else:
                g[s:s + ln:2], g[s + 1:s + ln:2] = acgt[int(rng.integers(0, 4))], acgt[int(rng.integers(0, 4))]
        genomes.append(g)
    taxa = list(range(9000, 9000 + len(sizes)))
    reads = synth.random_reads(701 + k, genomes, 300, 10_000, 0.15, n_rate=0.005)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, k, 6)
    o_ref = OracleDB(fa, 0, k, 6, threads=8)
    _same_db(ref, o_ref)
    rdb, names = km.make_read_db(ctx, fq, k, 6)
    o_reads = OracleDB(fq, 1, k, 6, threads=8)
    _same_db(rdb, o_reads)
    assert names == o_reads.names
    assert ctx.lib.kmg_db_payload_bits(rdb.h) == 32                                    # 14 position bits + 9 read bits
    hit, nh = km.DB([ref]).get_nearest_kmers(rdb)
    o_hit, o_nh = o_reads.match(o_ref)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    ref.close()
    rdb.close()


def test_packed_input_equals_letters(ctx):
    """kmg_build_add_packed (2-bit packed sequences, MakeDatabase -p records) against the same sequences as letters:
    lengths that are and are not multiples of four, a sequence shorter than k, an empty one; and its argument checks."""
    import kmers_amd as km
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(17)
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=n)) for n in (4001, 64, 3, 0, 1234, 8192, 777)]
    ids = [11, 12, 12, 13, 14, 11, 15]
    k, l = 19, 4
    packed = b"".join(R.sequence_compress(s)[4:] for s in seqs)
    lens = np.array([len(s) for s in seqs], np.uint64)
    seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    byte_off = np.concatenate([[0], np.cumsum((lens + np.uint64(3)) // np.uint64(4))]).astype(np.uint64)
    fc = km.FileCreator(ctx, l, k, "ref", rc=True)
    fc.add_packed(np.frombuffer(packed, np.uint8), seq_off, byte_off, np.array(ids, np.int32))
    got = fc.create()
    want = OracleDB(b"".join(b">%d\n%s\n" % (t, s.encode()) for t, s in zip(ids, seqs)), 0, k, l)
    _same_db(got, want)
    got.close()
    fc = km.FileCreator(ctx, l, k, "ref", rc=True)
    with pytest.raises(km.InconsistentDataException):
        fc.add_packed(np.frombuffer(packed, np.uint8), seq_off, byte_off + np.uint64(1), np.array(ids, np.int32))
    fc.create().close()
    fc = km.FileCreator(ctx, l, k, "read", rc=False)
    with pytest.raises(km.InconsistentDataException):
        fc.add_packed(np.frombuffer(packed, np.uint8), seq_off, byte_off, np.array(ids, np.int32))
    fc.create().close()


# ---------------------------------------------------------------------------- the bench step's shape
def test_bench_step_shape_two_contexts(ctx):
    """What bench.py times, at small scale and checked: kmg_build_add_dev from device tensors without a stream sync,
    the reference on a second context (a second stream) underneath the reads' extraction, kmg_match_dev across the two
    contexts into a device tensor, and kmg_db_free right behind it -- for several steps in a row, so the pools hand the
    just-freed arrays of one step to the next while the join of the step before may still be reading them (the library
    orders that: kmg_db::foreign_use).  Both databases and every hit index against the oracle, every step."""
    import ctypes as C
    import kmers_amd as km
    from kmers_amd import synth
    from oracle.c_oracle import OracleDB
    K, L = 31, 6
    dev = torch.device("cuda", 0)
    ctx_ref = km.Context(0)
    lib = ctx.lib
    try:
        steps = []
        for seed in (1, 2, 3):
            taxa, genomes = synth.random_genomes(seed, 4, 30000, shared_frac=0.1)
            reads = synth.random_reads(seed + 10, genomes, 6000, 150, 0.01)
            ref_bases, ref_off = synth.batch_from_seqs(genomes)
            t = dict(
                ref_bases=torch.from_numpy(ref_bases).to(dev), ref_off=torch.from_numpy(ref_off.view(np.int64)).to(dev),
                ref_id=torch.tensor(taxa, dtype=torch.int32, device=dev), nref=len(genomes),
                read_bases=torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).to(dev),
                read_off=torch.from_numpy((np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(150)).view(np.int64)).to(dev),
                read_tok=torch.arange(len(reads), dtype=torch.int64, device=dev), nreads=len(reads),
                o_ref=OracleDB(synth.fasta_bytes(taxa, genomes), 0, K, L), o_reads=OracleDB(synth.fastq_bytes(reads), 1, K, L))
            t["o_hit"], t["o_nh"] = t["o_reads"].match(t["o_ref"])
            t["hit"] = torch.full((t["o_reads"].U + 1,), -7, dtype=torch.int64, device=dev)
            steps.append(t)
        torch.cuda.synchronize()

        def finish(c):
            db, nw = C.c_void_p(), C.c_uint64()
            c.check(lib.kmg_build_finish(c.h, C.byref(db), C.byref(nw)))
            return db, nw.value

        for rep in range(2):
            for i, t in enumerate(steps):
                ctx_ref.check(lib.kmg_build_begin(ctx_ref.h, K, L, 1, 0))
                ctx_ref.check(lib.kmg_build_add_dev(ctx_ref.h, t["ref_bases"].data_ptr(), t["ref_bases"].numel(),
                                                    t["ref_off"].data_ptr(), t["ref_id"].data_ptr(), None, t["nref"], None))
                ctx.check(lib.kmg_build_begin(ctx.h, K, L, 0, 1))
                ctx.check(lib.kmg_build_add_dev(ctx.h, t["read_bases"].data_ptr(), t["read_bases"].numel(),
                                                t["read_off"].data_ptr(), None, t["read_tok"].data_ptr(), t["nreads"], None))
                ref, nw_ref = finish(ctx_ref)
                rdb, nw_reads = finish(ctx)
                assert nw_ref == t["o_ref"].n_windows and nw_reads == t["o_reads"].n_windows
                ctx.check(lib.kmg_match_dev(ctx.h, rdb, ref, t["hit"].data_ptr()))
                if rep == 0:
                    # first round: look at the databases too (fetch synchronises their contexts)
                    _same_db(km.KmerDatabase(ctx_ref, ref), t["o_ref"])
                    _same_db(km.KmerDatabase(ctx, rdb), t["o_reads"])
                # second round: free right behind the asynchronous join, then go straight on to the next step
                lib.kmg_db_free(rdb)
                lib.kmg_db_free(ref)
            torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
            torch.cuda.ExternalStream(ctx_ref.stream, device=dev).synchronize()
            for t in steps:
                n = t["o_reads"].U
                got = t["hit"].cpu().numpy()
                assert got[n] == -7
                assert np.array_equal(got[:n], t["o_hit"]) and int((got[:n] >= 0).sum()) == t["o_nh"]
                t["hit"].fill_(-7)
            torch.cuda.synchronize()
    finally:
        ctx_ref.close()


# ---------------------------------------------------------------------------- variable-length k-mers (minK < maxK)
def _same_var_db(got, want):
    _same_db(got, want)
    assert got.min_length == want.min_k and got.max_length == want.k
    assert np.array_equal(got.lens(), want.lens)
    assert got.encode_blocks()[0] == want.encode_blocks()[0]
    assert np.array_equal(got.encode_blocks()[1], want.encode_blocks()[1])


@pytest.mark.parametrize("min_k,k,key_len,seed", [(3, 5, 2, 31), (4, 8, 2, 32), (24, 32, 3, 33), (20, 31, 3, 34), (6, 7, 3, 35),
                                                  (1, 4, 1, 36), (24, 32, 6, 37), (31, 32, 5, 38), (12, 20, 4, 39)])
def test_variable_length_builds_match_oracle(ctx, min_k, k, key_len, seed):
    """MakeDatabase -k min_k -K k (the reference's default is -k 24 -K 32, MakeDatabase.java:106-107): the suffixes of
    KmersFromFile.java:90-133,167-176 next to the full windows, (k-mer, length) order (Sequence.java:165-187), a length
    byte per record in the blocks (FileCreator.java:312-347), reads numbered at their first k-mer of any length."""
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    small = k <= 8
    fa = _rand_fasta(rng, 40, 160 if not small else 60, b"AACCGTN" if small else b"ACGTacgtACGTACGTN", [5, 9, 1234567, 77])
    fq = _rand_fastq(rng, 300, 100 if not small else 30, b"AACGTN" if small else b"ACGTACGTACGTACGTN")
    ref = km.make_reference_db(ctx, fa, k, key_len, min_k=min_k)
    want = OracleDB(fa, 0, k, key_len, min_k=min_k)
    _same_var_db(ref, want)
    assert ref.n_windows == want.n_windows
    assert len(set(want.lens.tolist())) > 1
    rdb, names = km.make_read_db(ctx, fq, k, key_len, min_k=min_k)
    wantr = OracleDB(fq, 1, k, key_len, min_k=min_k)
    _same_var_db(rdb, wantr)
    assert names == wantr.names
    t, m, q = km.taxa_counts(ref)                       # TaxaCounts walks every stored k-mer, whatever its length
    ot, om, oq = want.taxacounts()
    assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    ref.close(); rdb.close()


def test_variable_length_edge_cases(ctx):
    """Sequences shorter than min_k, of exactly min_k .. k bases, an invalid last window (its suffixes come out twice:
    KmersFromFile.java:128-133 after :167-176), filters, key ranges, several add calls."""
    import kmers_amd as km
    from kmers_amd import parse
    from kmers_amd.api import FileCreator
    from oracle.c_oracle import OracleDB
    fa = (b">1\nAC\n>2\nACG\n>3\nACGT\n>4\nACGTA\n>5\nACGTAC\n>6\nACGTACG\n>7\nACGNACGTTGCA\n>8\nACGTTGCANAC\n"
          b">9\nNNNNNNNN\n>10\nACGTACGTNACGTACGTNACGT\n>11\nTTTTTTTTTTTTTTTTT\n>12\nACGTTGCAN\n>13\nACGTTGCANA\n")
    for min_k, k, kl in ((3, 6, 2), (2, 6, 1), (5, 6, 2), (3, 4, 2)):
        _same_var_db(km.make_reference_db(ctx, fa, k, kl, min_k=min_k), OracleDB(fa, 0, k, kl, min_k=min_k))
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i // 2, s, b"I" * len(s)) for i, s in enumerate(
        [b"ACGTAC", b"ACGNACGTTGCA", b"AC", b"NNNNNN", b"ACGTTGCANAC", b"ACG", b"TTTTTTTTTT", b"ACGTN"]))
    rdb, names = km.make_read_db(ctx, fq, 6, 2, min_k=3)
    want = OracleDB(fq, 1, 6, 2, min_k=3)
    _same_var_db(rdb, want)
    assert names == want.names
    rng = np.random.default_rng(5)
    fa2 = _rand_fasta(rng, 200, 300, b"AAAAACCGTN", [3, 4])
    for dust, run in ((2, None), (None, 5), (3, 6)):
        got = km.make_reference_db(ctx, fa2, 12, 3, dust=dust, runofsame=run, min_k=7)
        _same_var_db(got, OracleDB(fa2, 0, 12, 3, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run, min_k=7))
    for rng_ in ((0, 20), (20, 64), (64, 64)):
        got = km.make_reference_db(ctx, fa2, 12, 3, key_range=rng_, min_k=7)
        w = OracleDB(fa2, 0, 12, 3, key_range=rng_, min_k=7) if rng_[0] < rng_[1] else None
        if w is not None:
            _same_var_db(got, w)
        else:
            assert got.n_unique == 0
    # the same input in three add calls
    pf = parse.parse(fa2, parse.FASTA, 12)
    b = pf.batch
    fc = FileCreator(ctx, 3, 12, "ref", rc=True)
    cuts = [0, b.nseq // 3, b.nseq // 2, b.nseq]
    for a, e in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(b.seq_off[a]), int(b.seq_off[e])
        fc.add_kmers(km.SequenceBatch(b.bases[lo:hi], b.seq_off[a:e + 1] - b.seq_off[a], seq_id=b.seq_id[a:e]), min_length=7, max_length=12)
    _same_var_db(fc.create(), OracleDB(fa2, 0, 12, 3, min_k=7))


@pytest.mark.parametrize("min_k,k,L,key_len,seed", [(3, 6, 4, 2, 41), (3, 6, 3, 2, 42), (3, 6, 6, 2, 43), (20, 31, 25, 3, 44),
                                                    (24, 32, 24, 6, 45), (24, 32, 32, 6, 46), (24, 32, 28, 6, 47)])
def test_restricted_match_matches_oracle(ctx, min_k, k, L, key_len, seed):
    """Matcher -k L -K L over variable-length databases: both sides through KmerUtils.restrictedStream
    (KmerUtils.java:59-74; kmg_db_restrict), then the quick branch's join (DB.java:129-137)."""
    import kmers_amd as km
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    small = k <= 8
    fa = _rand_fasta(rng, 30, 300 if not small else 80, b"ACGT" * 12 + b"N" if not small else b"AACCGTN", [11, 12, 13])
    body = b"".join(l for l in fa.split(b"\n") if not l.startswith(b">"))
    recs = []
    for i in range(200):
        a = int(rng.integers(0, max(len(body) - 60, 1)))
        sq = body[a:a + int(rng.integers(min_k, 60))]
        recs.append(b"@q%d\n%s\n+\n%s\n" % (i, sq, b"I" * len(sq)))
    fq = b"".join(recs)
    ref = km.make_reference_db(ctx, fa, k, key_len, min_k=min_k)
    rdb, _ = km.make_read_db(ctx, fq, k, key_len, min_k=min_k)
    with pytest.raises(km.KmersGpuError):                   # DB.java:83: several lengths never take the quick branch
        km.DB([ref]).get_nearest_kmers(rdb)
    cref = OracleDB(fa, 0, k, key_len, min_k=min_k).restrict(L)
    creads = OracleDB(fq, 1, k, key_len, min_k=min_k).restrict(L)
    rref, rreads = ref.restrict(L), rdb.restrict(L)
    _same_db(rref, cref)
    _same_db(rreads, creads)
    assert rref.min_length == rref.max_length == L
    hit, nh = km.DB([rref]).get_nearest_kmers(rreads)
    ohit, onh = creads.match(cref)
    assert nh == onh and nh > 0 and np.array_equal(hit, ohit)
    assert "\n".join(km.match_lines(rreads, rref, hit)) + "\n" == creads.match_lines(cref, ohit).decode()
    with pytest.raises(km.InconsistentDataException):       # DB.java:78-81
        ref.restrict(min_k - 1)


@pytest.mark.parametrize("min_k,k,key_len,seed", [(3, 6, 2, 81), (24, 32, 4, 82), (20, 31, 3, 83)])
def test_variable_length_sets_cut_at_key_bounds_merge(ctx, min_k, k, key_len, seed):
    """A set of variable-length files that share no key (what MakeDatabase -S / -G write: name.1 .. name.N cut at key
    bounds) is one database again (kmg_db_merge lays the files end to end in key order; DB.kmers, DB.java:99-109):
    equal to the whole build and to the oracle, for both payload kinds, and the trie search over the set gives the
    whole database's lines.  Files that share a key are refused (the trie takes a key's k-mers file by file)."""
    import kmers_amd as km
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    small = k <= 8
    fa, fq = _trie_case(rng, b"AACCGTN" if small else b"ACGT" * 8 + b"N", 12 if small else 30, 60 if small else 300,
                        40 if small else 150, 25 if small else 60, [3, 4, 5])
    nkeys = 4 ** key_len
    cuts = [0, nkeys // 5, nkeys // 5, nkeys // 2 + 1, nkeys]               # (one range is empty)
    for kind, data, make in (("ref", fa, lambda kr: km.make_reference_db(ctx, fa, k, key_len, min_k=min_k, key_range=kr)),
                             ("read", fq, lambda kr: km.make_read_db(ctx, fq, k, key_len, min_k=min_k, key_range=kr)[0])):
        parts = [make((a, b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        whole = km.merge_databases(parts[::-1])                               # (the order of the files does not matter)
        want = OracleDB(data, 0 if kind == "ref" else 1, k, key_len, min_k=min_k)
        _same_var_db(whole, want)
        if kind == "ref":
            ref_set, ref_whole = km.DB(parts), whole
        else:
            rows, hit = km.match_trie(whole, ref_set)
            got = km.match_lines(whole, rows, hit, ref_stored=True)
            assert got == R.near_match_lines(R.build_read_db(fq, k, key_len, min_k=min_k)[0],
                                             R.DB([R.build_ref_db(fa, k, key_len, min_k=min_k)]), 0, False) and len(got) > 10
            t, m, q = km.taxa_counts(ref_set)
            ot, om, oq = OracleDB(fa, 0, k, key_len, min_k=min_k).taxacounts()
            assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
            rows.close()
            # two files holding the same keys merge as well (test_variable_length_sets_that_share_keys_merge), and the
            # search over several lengths takes the files one after the other as the reference's trie does: every k-mer of
            # the second file walks the pieces the first file's k-mers were split into
            twice = km.merge_databases([ref_whole, ref_whole])
            rows2, hit2 = km.match_trie(whole, twice)
            lit = R.build_ref_db(fa, k, key_len, min_k=min_k)
            assert km.match_lines(whole, rows2, hit2, ref_stored=True) == \
                R.near_match_lines(R.build_read_db(fq, k, key_len, min_k=min_k)[0], R.DB([lit, lit]), 0, False)
            rows2.close()
            twice.close()
            whole.close()
            for d in parts:
                d.close()


@pytest.mark.parametrize("min_k,k,key_len,seed", [(3, 6, 2, 91), (24, 32, 4, 92), (20, 31, 3, 93), (5, 5, 2, 94), (2, 9, 1, 95)])
def test_variable_length_sets_that_share_keys_merge(ctx, min_k, k, key_len, seed):
    """Files of k-mers of several lengths that SHARE keys -- databases built from different inputs and handed to Matcher /
    TaxaCounts as several -d files -- are one database as DB.kmers makes it (DB.java:99-109: k-way merge in
    Sequence.compareTo order = (zero-padded value, length), equal k-mers reduced with addAll): three reference files
    merged on the GPU equal the database of the three inputs built as one, the literal DB.kmers over the three files,
    and TaxaCounts / `Matcher -k L -K L` over the set give the whole's rows and hits.  Read databases unite their lists
    (the reference's sets hold ReadPos objects without equals: both copies stay)."""
    import kmers_amd as km
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(seed)
    small = k <= 9
    alphabet = b"AACCGTN" if small else b"ACGT" * 8 + b"N"
    fas = [_rand_fasta(rng, 14 if small else 25, 50 if small else 260, alphabet, [3, 4, 5, 77][:2 + i]) for i in range(3)]
    recs = fas[0].split(b">")[1:]
    fas[1] += b"".join(b">" + r for r in recs[:len(recs) // 2])          # shared sequences: counts add, taxa sets unite
    parts = [km.make_reference_db(ctx, fa, k, key_len, min_k=min_k) for fa in fas]
    merged = km.merge_databases(parts)
    want = OracleDB(b"".join(fas), 0, k, key_len, min_k=min_k)
    if min_k < k:
        _same_var_db(merged, want)
    else:
        _same_db(merged, want)
    # the literal DB.kmers over the three files, key by key
    files = [R.build_ref_db(fa, k, key_len, min_k=min_k) for fa in fas]
    lit = R.DB(files)
    h, lens = merged.fetch(), (merged.lens() if min_k < k else np.full(merged.n_unique, k, np.uint8))
    u = 0
    for key in range(4 ** key_len):
        for kmer, data in lit.kmers(key):
            text = str(kmer)
            assert len(text) == int(lens[u])
            got_text = "".join("ACTG"[(int(h["keys"][u]) >> (62 - 2 * i)) & 3] for i in range(len(text)))
            assert got_text == text
            a, b = int(h["val_off"][u]), int(h["val_off"][u + 1])
            assert dict(zip(h["val_a"][a:b].tolist(), h["val_b"][a:b].tolist())) == data
            u += 1
    assert u == merged.n_unique
    # TaxaCounts over the set, and the exact join after restriction to one length
    t, m, q = km.taxa_counts(km.DB(parts))
    ot, om, oq = want.taxacounts()
    assert np.array_equal(t, ot) and np.array_equal(m, om) and np.array_equal(q, oq)
    fq = _rand_fastq(rng, 200, 30 if small else 110, alphabet)
    rdb = km.make_read_db(ctx, fq, k, key_len, min_k=min_k)[0]
    o_reads = OracleDB(fq, 1, k, key_len, min_k=min_k)
    L = (min_k + k) // 2
    rref, rreads = merged.restrict(L), rdb.restrict(L)
    cref, creads = want.restrict(L), o_reads.restrict(L)
    _same_db(rref, cref)
    hit, nh = km.DB([rref]).get_nearest_kmers(rreads)
    ohit, onh = creads.match(cref)
    assert nh == onh and np.array_equal(hit, ohit)
    if min_k < k:
        # the search over several lengths: the reference's trie takes a key's k-mers file by file (DB.java:154-158) --
        # kmg_match_trie over the files the merged database kept, against the literal trie over the three files, in
        # both orders of the files (the answers depend on it)
        reads_lit = R.build_read_db(fq, k, key_len, min_k=min_k)[0]
        for order in ([0, 1, 2], [2, 0, 1]):
            mset = km.merge_databases([parts[i] for i in order])
            rows, hit = km.match_trie(rdb, mset)
            got = km.match_lines(rdb, rows, hit, ref_stored=True)
            assert got == R.near_match_lines(reads_lit, R.DB([files[i] for i in order]), 0, False)
            rows.close()
            mset.close()
        # read databases: the same file twice -- every list doubled, ascending
        twice = km.merge_databases([rdb, rdb])
        with pytest.raises(km.KmersGpuError):                 # search files that share k-mers are not merged by the reference
            km.match_trie(twice, merged)
        h1, h2 = rdb.fetch(), twice.fetch()
        assert np.array_equal(h2["keys"], h1["keys"]) and np.array_equal(twice.lens(), rdb.lens())
        assert np.array_equal(h2["val_off"], 2 * h1["val_off"])
        assert np.array_equal(h2["val_a"], np.repeat(h1["val_a"], 2)) and np.array_equal(h2["val_b"], np.repeat(h1["val_b"], 2))
        twice.close()
    for d in parts + [merged, rdb, rref, rreads]:
        d.close()


def test_trie_search_over_file_sets_matches_literal_trie(ctx):
    """kmg_match_trie over sets of two to four files that share keys, small alphabets (long shared prefixes: edges split by
    later files' k-mers, objects that take one k-mer in more than once), the search side sometimes narrower than the
    database: every line against the literal trie filled file by file (oracle/slow_ref.py Root, DB.java:154-158).  The
    derivation itself is held on the CPU by tests/test_trie_closed_form.py (1500 seeds in a one-off sweep)."""
    import random
    import kmers_amd as km
    from oracle import slow_ref as R
    lines = differ = 0
    for seed in range(int(os.environ.get("KMERS_TRIE_SET_SEEDS", "40"))):
        rng = random.Random(7000 + seed)
        alpha = rng.choice(["ACGT", "AC", "AAAC", "ACGTN", "AG"])
        max_k = rng.randint(3, 12)
        min_k = rng.randint(1, max_k - 1)
        key_len = rng.randint(1, min(min_k, 3))
        pool = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 70))) for _ in range(rng.randint(2, 8))]
        fas = []
        for _ in range(rng.randint(2, 4)):
            seqs = [rng.choice(pool) if rng.random() < 0.6 else "".join(rng.choice(alpha) for _ in range(rng.randint(0, 70)))
                    for _ in range(rng.randint(1, 6))]
            fas.append("".join(">%d\n%s\n" % (rng.randint(1, 5), q) for q in seqs).encode())
        fq = "".join("@r%d\n%s\n+\n%s\n" % (i, q, "I" * len(q)) for i, q in
                     enumerate((rng.choice(pool) if rng.random() < 0.5 else "".join(rng.choice(alpha) for _ in range(rng.randint(0, 50))))
                               for _ in range(rng.randint(1, 10)))).encode()
        s_min = rng.randint(min_k, max_k)
        s_max = rng.randint(s_min, max_k)
        if rng.random() < 0.5:
            s_min, s_max = min_k, max_k
        if key_len > s_min or s_min == s_max:
            continue
        parts = [km.make_reference_db(ctx, fa, max_k, key_len, min_k=min_k) for fa in fas]
        if sum(d.n_unique for d in parts) == 0:
            continue
        rdb = km.make_read_db(ctx, fq, s_max, key_len, min_k=s_min)[0]
        mset = km.merge_databases(parts)
        rows, hit = km.match_trie(rdb, mset)
        got = km.match_lines(rdb, rows, hit, ref_stored=True)
        files = [R.build_ref_db(fa, max_k, key_len, min_k=min_k) for fa in fas]
        want = R.near_match_lines(R.build_read_db(fq, s_max, key_len, min_k=s_min)[0], R.DB(files), 0, False)
        assert got == want, (seed, alpha, min_k, max_k, s_min, s_max, key_len, len(fas))
        lines += len(want)
        differ += want != R.near_match_lines(R.build_read_db(fq, s_max, key_len, min_k=s_min)[0], R.DB(files[::-1]), 0, False)
        for d in parts + [rdb, mset, rows]:
            d.close()
    assert lines > 300 and differ > 3          # (the cases bite: the files' order changes the answers)


def _trie_case(rng, alphabet, n_seq, seq_len, n_reads, read_len, taxa):
    fa = _rand_fasta(rng, n_seq, seq_len, alphabet, taxa)
    body = b"".join(l for l in fa.split(b"\n") if not l.startswith(b">"))
    recs = []
    for i in range(n_reads):
        a = int(rng.integers(0, max(len(body) - read_len, 1)))
        sq = bytearray(body[a:a + int(rng.integers(1, read_len + 1))])
        if len(sq) and rng.random() < 0.3:                   # a changed base: suffixes / prefixes still match
            sq[int(rng.integers(0, len(sq)))] = alphabet[int(rng.integers(0, len(alphabet)))]
        recs.append(b"@q%d\n%s\n+\n%s\n" % (i, bytes(sq), b"I" * len(sq)))
    return fa, b"".join(recs)


@pytest.mark.parametrize("min_k,k,key_len,alphabet,seed", [
    (3, 6, 2, b"AACCGTN", 61), (2, 9, 1, b"AC", 62), (1, 5, 1, b"AAAC", 63), (4, 8, 3, b"ACGT", 64), (3, 7, 2, b"AG", 65),
    (20, 31, 3, b"ACGT" * 12 + b"N", 66), (24, 32, 6, b"ACGT" * 12 + b"N", 67), (24, 32, 6, b"ACGT", 68), (31, 32, 4, b"ACGT", 69)])
def test_trie_search_matches_literal_trie(ctx, min_k, k, key_len, alphabet, seed):
    """Matcher with its default -k / -K on variable-length files: DB.getNearestKmers' non-quick branch at maxDiff = 0
    (DB.java:83,139-184) -- kmg_match_trie's closed form against the literal restatement of Root.addKmer / closestKmers
    (oracle/slow_ref.py Root; shared data objects included), line by line.  Small alphabets force long shared prefixes
    (answers summed over many stored k-mers); the search side also narrower than the database."""
    import kmers_amd as km
    from oracle import slow_ref as R
    rng = np.random.default_rng(seed)
    small = k <= 9
    fa, fq = _trie_case(rng, alphabet, 12 if small else 30, 60 if small else 300, 40 if small else 200, 25 if small else 60,
                        [11, 12, 13])
    ref = km.make_reference_db(ctx, fa, k, key_len, min_k=min_k)
    o_ref = R.build_ref_db(fa, k, key_len, min_k=min_k)
    total = 0
    for s_min, s_max in {(min_k, k), (min(min_k + 1, k), k), (min_k, max(k - 1, min_k)), (max(min_k, k - 2), k)}:
        if key_len > s_min:
            continue
        rdb, _ = km.make_read_db(ctx, fq, s_max, key_len, min_k=s_min)
        o_reads, _ = R.build_read_db(fq, s_max, key_len, min_k=s_min)
        want = R.near_match_lines(o_reads, R.DB([o_ref]), 0, False)
        rows, hit = km.match_trie(rdb, ref, s_min)
        got = km.match_lines(rdb, rows, hit, ref_stored=True)
        assert got == want, (s_min, s_max, len(got), len(want))
        assert int((hit >= 0).sum()) == len(want) == rows.n_unique
        total += len(want)
        rows.close()
        # ... and with differences allowed (-n 1 | 2 [-j]): every T-base prefix of a stored k-mer within reach, each with
        # what the trie holds at depth T on its path
        for max_diff, just in ((1, False), (2, True)) + (((3, False),) if small else ()):
            near = R.near_match_lines(o_reads, R.DB([o_ref]), max_diff, just)
            assert km.near_match_lines_var(rdb, ref, max_diff, just, min_length=s_min) == near, (s_min, s_max, max_diff, just)
            assert len(near) >= len(want)
        rdb.close()
    assert total > 0
    if min_k > key_len:
        rdb, _ = km.make_read_db(ctx, fq, k, key_len, min_k=min_k)
        with pytest.raises(km.InconsistentDataException):   # DB.java:78-81: search lengths outside the database's
            km.match_trie(rdb, ref, min_k - 1)
        rdb.close()
    ref.close()


@pytest.mark.parametrize("min_k,k,key_len,alphabet,seed", [(3, 8, 2, b"AACCGTN", 91), (2, 9, 1, b"AC", 92), (24, 32, 4, b"ACGT", 93)])
def test_length_windows_other_than_the_files_own(ctx, min_k, k, key_len, alphabet, seed):
    """Matcher -k a -K b with a < b: the search side goes through KmerUtils.restrictedStream(a, b) (KmerUtils.java:59-74;
    kmg_db_restrict_range: shorter dropped, longer cut and folded, the rest untouched), the trie is built with (a, b) over
    the whole database.  The restricted database entry by entry against the restated stream function, then the lines
    with and without differences against the literal trie."""
    import kmers_amd as km
    from oracle import slow_ref as R
    rng = np.random.default_rng(seed)
    small = k <= 9
    fa, fq = _trie_case(rng, alphabet, 12 if small else 30, 60 if small else 300, 40 if small else 150, 25 if small else 60, [6, 7])
    ref = km.make_reference_db(ctx, fa, k, key_len, min_k=min_k)
    rdb, _ = km.make_read_db(ctx, fq, k, key_len, min_k=min_k)
    o_ref, (o_reads, _) = R.build_ref_db(fa, k, key_len, min_k=min_k), R.build_read_db(fq, k, key_len, min_k=min_k)
    windows = {(min_k, k - 1), (min_k + 1, k), (min_k + 1, k - 1), (max(min_k, key_len), min_k + 1)}
    for a, b in sorted(w for w in windows if key_len <= w[0] < w[1] <= k):
        for db, blocks, kind in ((ref, o_ref, "ref"), (rdb, o_reads, "read")):
            cut = db.restrict(a, b)
            assert (cut.min_length, cut.max_length) == (a, b)
            want = []
            for key in range(4 ** key_len):
                want.extend(R.restricted_stream(R.decode_block(blocks.get(key, b""), kind), min_k, k, a, b, kind))
            got, lens = cut.fetch(), cut.lens()
            assert cut.n_unique == len(want)
            names = km.kmer_strings(got["keys"], 32)
            assert [n[:l] for n, l in zip(names, lens.tolist())] == [str(kk) for kk, _ in want]
            for i in rng.integers(0, max(len(want), 1), size=min(len(want), 200)).tolist():
                c0, c1 = int(got["val_off"][i]), int(got["val_off"][i + 1])
                vals = list(zip(got["val_a"][c0:c1].tolist(), got["val_b"][c0:c1].tolist()))
                assert vals == (sorted(want[i][1].items()) if kind == "ref" else sorted(want[i][1])), (a, b, kind, i)
            if kind == "read":
                for n_, just in ((0, False), (1, False), (2, True)):
                    lines = R.near_match_lines(o_reads, R.DB([o_ref]), n_, just, a, b)
                    if n_ == 0:
                        rows, hit = km.match_trie(cut, ref, a)
                        assert km.match_lines(cut, rows, hit, ref_stored=True) == lines and len(lines) > 5
                        rows.close()
                    else:
                        assert km.near_match_lines_var(cut, ref, n_, just, min_length=a) == lines
            cut.close()
    ref.close()
    rdb.close()


def test_trie_search_counts_and_reference_search_side(ctx):
    """Counts beyond a file's byte are cut to 255 BEFORE the trie adds them up (the reference reads them from the file,
    CountDataType.java:40), and a reference database can be the search side (Matcher.java:182-210): `TxC` on both
    sides of the line."""
    import kmers_amd as km
    from oracle import slow_ref as R
    unit = b"ACGTTGCAAGGCTTAACCGGATAT"
    fa = (b">7\n" + (unit + b"N") * 300 + b"\n>7\n" + (unit[:21] + b"GGGN") * 300 + b"\n>8\n" + unit + b"ACCA\n>9\n" +
          unit[:20] + b"TTTTGG\n")
    q = b">1\n" + unit + b"AC\n>2\n" + unit[:22] + b"\n"
    min_k, k, l = 20, 24, 3
    ref = km.make_reference_db(ctx, fa, k, l, min_k=min_k)
    sdb = km.make_reference_db(ctx, q, k, l, min_k=min_k)
    o_ref, o_s = R.build_ref_db(fa, k, l, min_k=min_k), R.build_ref_db(q, k, l, min_k=min_k)
    want = R.near_match_lines(o_s, R.DB([o_ref]), 0, False)
    rows, hit = km.match_trie(sdb, ref, min_k)
    got = km.match_lines(sdb, rows, hit, ref_stored=True)
    assert got == want
    assert max(int(x.split("x")[-1]) for w in want for x in w.split("\t")[2].split("|")[1:]) > 255     # sums pass 255
    for d in (rows, ref, sdb):
        d.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KMERS_FUZZ_VAR_SEEDS", "12"))))
def test_fuzz_variable_length_against_oracle(ctx, seed):
    """Random minK < maxK, key length, input shape, batch cuts, filters and key ranges; then Matcher -k L -K L at a
    random length."""
    import kmers_amd as km
    from kmers_amd import parse, synth
    from oracle.c_oracle import OracleDB
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(6, 33))
    mn = int(rng.integers(max(1, k - 10), k))
    l = int(rng.integers(1, min(mn, 6) + 1))
    n_gen, glen = int(rng.integers(1, 5)), int(rng.integers(300, 20000))
    taxa, genomes = synth.random_genomes(500 + seed, n_gen, glen, first_taxon=int(rng.integers(-5, 1000)),
                                         shared_frac=float(rng.choice([0.0, 0.3])))
    if rng.random() < 0.3:
        genomes[0][: glen // 2] = ord("A")
    n_reads, rlen = int(rng.integers(50, 4000)), int(rng.integers(mn, 200))
    reads = synth.random_reads(600 + seed, genomes, n_reads, rlen, 0.02, n_rate=float(rng.choice([0.0, 0.01, 0.05])))
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    # ragged read lengths around minK .. maxK: cut some reads short
    recs = fq.split(b"\n")
    for i in range(0, len(recs) - 3, 4):
        if rng.random() < 0.3:
            n = int(rng.integers(0, min(len(recs[i + 1]), k + 3) + 1))
            recs[i + 1], recs[i + 3] = recs[i + 1][:n], recs[i + 3][:n]
    fq = b"\n".join(recs)
    dust = int(rng.integers(1, 4)) if rng.random() < 0.25 else None
    run = int(rng.integers(3, 9)) if rng.random() < 0.25 else None
    key_range = None
    if rng.random() < 0.3:
        a = int(rng.integers(0, 4 ** l))
        key_range = (a, int(rng.integers(a + 1, 4 ** l + 1)))
    built = {}
    for data, mode, kind, rc in ((fa, parse.FASTA, "ref", True), (fq, parse.FASTQ, "read", False)):
        b = parse.parse(data, mode, k).batch
        ncut = int(rng.integers(0, 3))
        cuts = [0] + sorted(int(x) for x in rng.integers(0, b.nseq + 1, size=ncut)) + [b.nseq]
        fc = km.FileCreator(ctx, l, k, kind, rc=rc, dust=dust, runofsame=run, key_range=key_range)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a0, a1 = int(b.seq_off[lo]), int(b.seq_off[hi])
            fc.add_kmers(km.SequenceBatch(b.bases[a0:a1], b.seq_off[lo:hi + 1] - b.seq_off[lo],
                                          seq_id=None if b.seq_id is None else b.seq_id[lo:hi],
                                          seq_tok=None if b.seq_tok is None else b.seq_tok[lo:hi]), min_length=mn, max_length=k)
        got = fc.create()
        want = OracleDB(data, 0 if kind == "ref" else 1, k, l, threads=4, dust=-1 if dust is None else dust,
                        runofsame=-1 if run is None else run, key_range=key_range, min_k=mn)
        _same_var_db(got, want)
        assert fc.n_windows == want.n_windows
        built[kind] = (got, want)
    L = int(rng.integers(max(mn, l), k + 1))
    (ref, o_ref), (rdb, o_reads) = built["ref"], built["read"]
    rr, rs, orr, ors = ref.restrict(L), rdb.restrict(L), o_ref.restrict(L), o_reads.restrict(L)
    _same_db(rr, orr)
    _same_db(rs, ors)
    hit, nh = km.DB([rr]).get_nearest_kmers(rs)
    o_hit, o_nh = ors.match(orr)
    assert nh == o_nh and np.array_equal(hit, o_hit)
    if len(fa) + len(fq) < 250000:
        # ... and Matcher with its default -k / -K: all lengths at once through the trie (the literal restatement is
        # Python over every stored k-mer, so only the smaller cases)
        from oracle import slow_ref as R
        want = R.near_match_lines(_oracle_var_blocks(o_reads), R.DB([_oracle_var_blocks(o_ref)]), 0, False)
        rows, thit = km.match_trie(rdb, ref)
        assert km.match_lines(rdb, rows, thit, ref_stored=True) == want
        rows.close()
    for d in (ref, rdb, rr, rs):
        d.close()


def _oracle_var_blocks(odb):
    """The C oracle's database as the {key: block bytes, -1: meta} dict the Python restatement reads
    (meta = [minK][maxK][keyLength][n ids][ids][rc], FileCreator.java:153-164)."""
    blob, off = odb.encode_blocks()
    ids = [2048] if odb.kind == 0 else [257, 2052]
    blocks = {-1: bytes([odb.min_k, odb.k, odb.key_len]) + len(ids).to_bytes(4, "big") + b"".join(i.to_bytes(4, "big") for i in ids) +
              bytes([1 if odb.kind == 0 else 0])}
    for key in range(4 ** odb.key_len):
        blocks[key] = blob[int(off[key]):int(off[key + 1])]
    return blocks


def test_exchange_under_the_c_abi_world_of_one(ctx):
    """kmg_comm_create / kmg_match_sharded / kmg_db_exchange_by_key_range with one rank: the grouped ncclSend / ncclRecv
    are sends to self, everything else (cuts, count exchange, one-launch join of the received runs, reverse exchange,
    rebuilding and merging the received parts) is the code the N-GPU run takes.  Against the oracle's whole-database
    join, with ranges that leave some buckets without an owner."""
    import kmers_amd as km
    from kmers_amd import dist as kd, synth
    from oracle.c_oracle import OracleDB
    k, l = 21, 4
    taxa, genomes = synth.random_genomes(301, 4, 30000, shared_frac=0.3)
    reads = synth.random_reads(302, genomes, 4000, 100, 0.01, n_rate=0.002)
    fa, fq = synth.fasta_bytes(taxa, genomes), synth.fastq_bytes(reads)
    ref = km.make_reference_db(ctx, fa, k, l)
    rdb, _ = km.make_read_db(ctx, fq, k, l)
    o_ref, o_reads = OracleDB(fa, 0, k, l), OracleDB(fq, 1, k, l)
    o_hit, _ = o_reads.match(o_ref)
    try:
        comm = kd.Comm(ctx, rank=0, world=1)
    except km.KmersGpuError as e:
        pytest.skip("no RCCL on this machine: %s" % e)
    # the whole key space owned by the one rank: the plain join
    hit = comm.match_sharded(rdb, ref, [(0, 4 ** l)])
    assert np.array_equal(hit.cpu().numpy(), o_hit)
    owner, hit2 = kd.match_key_range_shard(ctx, rdb, ref, [(0, 4 ** l)], comm=comm)
    assert np.array_equal(hit2.cpu().numpy(), o_hit) and int(owner.min()) == 0 == int(owner.max())
    # a range that leaves the first and the last buckets without an owner: those k-mers find nothing
    lo, hi = 37, 4 ** l - 11
    sub = comm.match_sharded(rdb, ref, [(lo, hi)])
    boff = o_reads.bucket_off
    want = o_hit.copy()
    want[:int(boff[lo])] = -1
    want[int(boff[hi]):] = -1
    assert np.array_equal(sub.cpu().numpy(), want) and (want >= 0).sum() > 1000
    # the database exchange: the rank's own slice comes back, merged into one database
    shard = comm.exchange_by_key_range(ref, [(lo, hi)])
    h = shard.fetch()
    a, b = int(o_ref.bucket_off[lo]), int(o_ref.bucket_off[hi])
    va, vb = int(o_ref.val_off[a]), int(o_ref.val_off[b])
    assert np.array_equal(h["keys"], o_ref.keys[a:b]) and np.array_equal(h["val_off"], o_ref.val_off[a:b + 1] - o_ref.val_off[a])
    assert np.array_equal(h["val_a"], o_ref.val_a[va:vb]) and np.array_equal(h["val_b"], o_ref.val_b[va:vb])
    whole = kd.build_key_range_shard(ctx, rdb, [(0, 4 ** l)], comm=comm)          # a read database, all of it
    _same_db(whole, o_reads)
    empty = comm.exchange_by_key_range(ref, [(5, 5)])
    assert empty.n_unique == 0 and empty.n_values == 0
    for d in (shard, whole, empty, ref, rdb):
        d.close()
    comm.close()
