"""The two independent CPU restatements (literal Python vs C) must agree byte for byte.

Parity with Kmers.jar itself is UNPINNED (no JVM here, no fixtures in the reference): this test
is the substitute the survey prescribes -- two restatements written independently from the Java
source agreeing on block bytes, read-name maps, match lines and TaxaCounts lines.
"""
import numpy as np
import pytest

from oracle import slow_ref as R
from oracle.c_oracle import OracleDB


def _rand_fasta(rng, n_seq, max_len, alphabet=b"ACGTacgtN", taxa=(5, 7, 11, -3)):
    parts = []
    for _ in range(n_seq):
        L = int(rng.integers(0, max_len))
        s = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), size=L).tolist())
        parts.append(b">%d some description\n%s\n" % (taxa[int(rng.integers(0, len(taxa)))], s))
    return b"".join(parts)


def _rand_fastq(rng, n_seq, max_len, alphabet=b"ACGTN"):
    parts = []
    for i in range(n_seq):
        L = int(rng.integers(0, max_len))
        s = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), size=L).tolist())
        name = b"r%d/x y" % (i // 2 if i % 5 == 0 else i)
        parts.append(b"@%s\n%s\n+\n%s\n" % (name, s, b"I" * L))
    return b"".join(parts)


def _py_blocks_concat(blocks, key_len):
    return b"".join(blocks[k] for k in range(4 ** key_len))


@pytest.mark.parametrize("k,key_len,seed", [(4, 2, 1), (5, 2, 2), (8, 3, 3), (31, 3, 4), (32, 3, 5), (6, 6, 6)])
def test_ref_build_blocks_agree(k, key_len, seed):
    rng = np.random.default_rng(seed)
    # low-entropy alphabet at small k => many duplicates and palindromes (even k)
    fa = _rand_fasta(rng, 12, 200, alphabet=b"ACGTacgtN" if k > 8 else b"AACCGTN")
    py = R.build_ref_db(fa, k, key_len)
    c = OracleDB(fa, 0, k, key_len)
    cb, off = c.encode_blocks()
    assert cb == _py_blocks_concat(py, key_len)
    for key in range(4 ** key_len):
        assert cb[int(off[key]):int(off[key + 1])] == py[key]


@pytest.mark.parametrize("k,key_len,seed", [(4, 2, 11), (7, 3, 12), (31, 3, 13)])
def test_read_build_blocks_and_names_agree(k, key_len, seed):
    rng = np.random.default_rng(seed)
    fq = _rand_fastq(rng, 40, 90, alphabet=b"ACGTN" if k > 8 else b"AACGTN")
    py, names = R.build_read_db(fq, k, key_len)
    c = OracleDB(fq, 1, k, key_len)
    cb, _ = c.encode_blocks()
    assert cb == _py_blocks_concat(py, key_len)
    assert c.names == names


@pytest.mark.parametrize("k,key_len,seed", [(5, 2, 21), (9, 3, 22)])
def test_match_and_taxacounts_agree(k, key_len, seed):
    rng = np.random.default_rng(seed)
    fa = _rand_fasta(rng, 10, 300, alphabet=b"ACGT")
    fq = _rand_fastq(rng, 30, 60, alphabet=b"ACGT")
    pyref = R.build_ref_db(fa, k, key_len)
    pyreads, _ = R.build_read_db(fq, k, key_len)
    db = R.DB([pyref])
    cref = OracleDB(fa, 0, k, key_len)
    creads = OracleDB(fq, 1, k, key_len)
    hit, nh = creads.match(cref)
    lines = creads.match_lines(cref, hit).decode().splitlines()
    assert lines == R.match_lines(pyreads, db)
    assert nh == len(lines) and nh > 0
    t, m, q = cref.taxacounts()
    assert ["%d\t%d\t%d" % x for x in zip(t, m, q)] == R.taxa_counts(db)


def test_no_trailing_newline_quirk_agrees():
    fa = b">9\nACGTACGTTGCA"
    py = R.build_ref_db(fa, 4, 2)
    c = OracleDB(fa, 0, 4, 2)
    assert c.encode_blocks()[0] == _py_blocks_concat(py, 2)


def test_multithreaded_build_is_identical():
    rng = np.random.default_rng(7)
    fa = _rand_fasta(rng, 50, 2000, alphabet=b"ACGT")
    a = OracleDB(fa, 0, 21, 4, threads=1)
    b = OracleDB(fa, 0, 21, 4, threads=4)
    assert np.array_equal(a.keys, b.keys) and np.array_equal(a.val_a, b.val_a)
    assert np.array_equal(a.val_b, b.val_b) and np.array_equal(a.val_off, b.val_off)


@pytest.mark.parametrize("k,dust,run", [(9, 2, None), (9, None, 4), (12, 1, 3), (31, 14, 5)])
def test_dust_and_runofsame_filters_agree(k, dust, run):
    """MakeDatabase -D / -R (MakeDatabase.java:279-291, Dust.java, RunOfSame.java)."""
    rng = np.random.default_rng(100 + k)
    fa = _rand_fasta(rng, 12, 300, alphabet=b"AAAACGT")
    fq = _rand_fastq(rng, 30, 120, alphabet=b"AAACCGT")
    py = R.build_ref_db(fa, k, 3, dust=dust, runofsame=run)
    c = OracleDB(fa, 0, k, 3, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run)
    assert c.encode_blocks()[0] == _py_blocks_concat(py, 3)
    unfiltered = OracleDB(fa, 0, k, 3)
    assert c.V < unfiltered.V                       # the filters did drop something
    pyr, names = R.build_read_db(fq, k, 3, dust=dust, runofsame=run)
    cr = OracleDB(fq, 1, k, 3, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run)
    assert cr.encode_blocks()[0] == _py_blocks_concat(pyr, 3)
    assert cr.names == names                        # reads are numbered before the filter sees their k-mers


# ---------------------------------------------------------------------------- variable-length k-mers (minK < maxK)
def _lens_from_py(py, key_len, kind):
    out = []
    for key in range(4 ** key_len):
        out.extend(k.length() for k, _ in R.decode_block(py[key], kind))
    return out


@pytest.mark.parametrize("min_k,k,key_len,seed", [(3, 5, 2, 31), (4, 8, 2, 32), (24, 32, 3, 33), (20, 31, 3, 34), (6, 7, 3, 35)])
def test_variable_length_builds_agree(min_k, k, key_len, seed):
    """MakeDatabase -k min_k -K k (the reference's default is -k 24 -K 32): suffixes of length k-1 .. min_k at the
    end of every sequence and behind every invalid base (KmersFromFile.java:90-133,167-176), records ordered by
    (k-mer, length) (Sequence.java:165-187), every record with its own length byte (FileCreator.java:312-347)."""
    rng = np.random.default_rng(seed)
    small = k <= 8
    fa = _rand_fasta(rng, 14, 160 if not small else 60, alphabet=b"AACCGTN" if small else b"ACGTacgtACGTACGTN")
    fq = _rand_fastq(rng, 40, 100 if not small else 30, alphabet=b"AACGTN" if small else b"ACGTACGTACGTACGTN")
    py = R.build_ref_db(fa, k, key_len, min_k=min_k)
    c = OracleDB(fa, 0, k, key_len, min_k=min_k)
    assert c.encode_blocks()[0] == _py_blocks_concat(py, key_len)
    assert c.lens.tolist() == _lens_from_py(py, key_len, "ref") and min(c.lens) >= min_k and max(c.lens) <= k
    assert len(set(c.lens.tolist())) > 1
    pyr, names = R.build_read_db(fq, k, key_len, min_k=min_k)
    cr = OracleDB(fq, 1, k, key_len, min_k=min_k)
    assert cr.encode_blocks()[0] == _py_blocks_concat(pyr, key_len)
    assert cr.names == names and cr.lens.tolist() == _lens_from_py(pyr, key_len, "read")


def test_variable_length_without_trailing_newline_agrees():
    for fa in (b">9\nACGTACGTTGCA", b">9\nACGTACGTTGCA\n>4\nACNTACGTTG", b">9\nACGTANGTTGCATT"):
        py = R.build_ref_db(fa, 6, 2, min_k=3)
        c = OracleDB(fa, 0, 6, 2, min_k=3)
        assert c.encode_blocks()[0] == _py_blocks_concat(py, 2)


@pytest.mark.parametrize("min_k,k,L,key_len,seed", [(3, 6, 4, 2, 41), (3, 6, 3, 2, 42), (3, 6, 6, 2, 43), (20, 31, 25, 3, 44)])
def test_restricted_match_agrees(min_k, k, L, key_len, seed):
    """Matcher -k L -K L over variable-length databases: both sides through KmerUtils.restrictedStream
    (KmerUtils.java:59-74), then the quick branch's join (DB.java:129-137)."""
    rng = np.random.default_rng(seed)
    small = k <= 8
    fa = _rand_fasta(rng, 10, 200 if not small else 80, alphabet=b"ACGT" * 12 + b"N" if not small else b"AACCGTN")
    # reads cut from the reference text so that long k-mers match too
    body = b"".join(l for l in fa.split(b"\n") if not l.startswith(b">"))
    recs = []
    for i in range(40):
        a = int(rng.integers(0, max(len(body) - 60, 1)))
        sq = body[a:a + int(rng.integers(min_k, 60))]
        recs.append(b"@q%d\n%s\n+\n%s\n" % (i, sq, b"I" * len(sq)))
    fq = b"".join(recs)
    pyref = R.build_ref_db(fa, k, key_len, min_k=min_k)
    pyreads, _ = R.build_read_db(fq, k, key_len, min_k=min_k)
    want = R.match_lines(pyreads, R.DB([pyref]), min_k=L, max_k=L)
    cref = OracleDB(fa, 0, k, key_len, min_k=min_k).restrict(L)
    creads = OracleDB(fq, 1, k, key_len, min_k=min_k).restrict(L)
    hit, nh = creads.match(cref)
    assert creads.match_lines(cref, hit).decode().splitlines() == want
    assert nh == len(want) and nh > 0


# ---- two-word k-mers (k > 32): the restatement built with -DKO_WIDE ---------------------------------------------------
@pytest.mark.parametrize("k,key_len,seed", [(4, 2, 51), (31, 3, 52), (32, 3, 53)])
def test_wide_build_of_the_restatement_equals_the_narrow_one_up_to_32(k, key_len, seed):
    rng = np.random.default_rng(seed)
    fa = _rand_fasta(rng, 12, 200, alphabet=b"ACGTacgtN" if k > 8 else b"AACCGTN")
    fq = _rand_fastq(rng, 40, 90, alphabet=b"ACGTN" if k > 8 else b"AACGTN")
    for data, mode in ((fa, 0), (fq, 1)):
        a, b = OracleDB(data, mode, k, key_len), OracleDB(data, mode, k, key_len, wide=True)
        for f in ("keys", "val_off", "val_a", "val_b", "bucket_off"):
            assert np.array_equal(getattr(a, f), getattr(b, f))
        assert not b.keys_lo.any()
        assert a.encode_blocks()[0] == b.encode_blocks()[0] and a.names == b.names


@pytest.mark.parametrize("k,key_len,seed,alphabet", [(33, 3, 61, b"ACGTacgtN"), (40, 2, 62, b"ACGT"), (47, 3, 63, b"AC"),
                                                     (63, 3, 64, b"ACGTN"), (64, 3, 65, b"ACGT"), (64, 2, 66, b"AT"),
                                                     (36, 2, 67, b"A")])
def test_wide_builds_agree_with_the_literal_restatement(k, key_len, seed, alphabet):
    """slow_ref.py keeps a k-mer as a list of bases of any length, as the reference does."""
    rng = np.random.default_rng(seed)
    fa = _rand_fasta(rng, 10, 300, alphabet=alphabet)
    py = R.build_ref_db(fa, k, key_len)
    c = OracleDB(fa, 0, k, key_len)
    cb, off = c.encode_blocks()
    assert cb == _py_blocks_concat(py, key_len)
    fq = _rand_fastq(rng, 30, 200, alphabet=alphabet)
    pyr, names = R.build_read_db(fq, k, key_len)
    cr = OracleDB(fq, 1, k, key_len)
    assert cr.encode_blocks()[0] == _py_blocks_concat(pyr, key_len)
    assert cr.names == names
    # the exact join and its lines, TaxaCounts
    hit, n = cr.match(c)
    want = R.match_lines(pyr, R.DB([py]))
    got = cr.match_lines(c, hit).decode().splitlines()
    assert got == want and n == len(want)
    t, m, q = c.taxacounts()
    assert ["%d\t%d\t%d" % (a, b, cc) for a, b, cc in zip(t, m, q)] == R.taxa_counts(R.DB([py]))


@pytest.mark.parametrize("min_k,k,key_len,seed,alphabet", [(24, 40, 3, 71, b"ACGTN" * 3 + b"ACGT" * 6), (30, 64, 2, 72, b"ACGT"),
                                                           (33, 36, 2, 73, b"AC"), (20, 33, 3, 74, b"ACGTacgtN" + b"ACGT" * 5),
                                                           (40, 47, 4, 75, b"AT")])
def test_wide_builds_of_several_lengths_agree_with_the_literal_restatement(min_k, k, key_len, seed, alphabet):
    """MakeDatabase -k min_k -K k with k > 32, and Matcher -k L -K L over the two databases (KmerUtils.restrictedStream)."""
    rng = np.random.default_rng(seed)
    fa = _rand_fasta(rng, 12, 300, alphabet=alphabet)
    fq = _rand_fastq(rng, 40, 200, alphabet=alphabet)
    py = R.build_ref_db(fa, k, key_len, min_k=min_k)
    c = OracleDB(fa, 0, k, key_len, min_k=min_k)
    assert c.encode_blocks()[0] == _py_blocks_concat(py, key_len)
    pyr, names = R.build_read_db(fq, k, key_len, min_k=min_k)
    cr = OracleDB(fq, 1, k, key_len, min_k=min_k)
    assert cr.encode_blocks()[0] == _py_blocks_concat(pyr, key_len) and cr.names == names
    assert len(set(c.lens.tolist())) > 1
    for L in (min_k, (min_k + k) // 2, k):
        want = R.match_lines(pyr, R.DB([py]), min_k=L, max_k=L)
        s_, d_ = cr.restrict(L), c.restrict(L)
        hit, _ = s_.match(d_)
        assert s_.match_lines(d_, hit).decode().splitlines() == want


@pytest.mark.parametrize("min_k,k,dust,run,alphabet", [(40, 40, 2, None, b"AAAACCGT"), (40, 40, None, 6, b"AAAACGT"), (30, 50, 3, 5, b"AAACCGT"),
                                                       (64, 64, 1, 4, b"AACGT"), (12, 12, 1, 3, b"AAACGT")])
def test_wide_filters_agree_with_the_literal_restatement(min_k, k, dust, run, alphabet):
    rng = np.random.default_rng(k + (dust or 0))
    fa = _rand_fasta(rng, 12, 300, alphabet=alphabet)
    py = R.build_ref_db(fa, k, 2, min_k=min_k, dust=dust, runofsame=run)
    c = OracleDB(fa, 0, k, 2, min_k=min_k, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run, wide=True)
    assert c.encode_blocks()[0] == _py_blocks_concat(py, 2)
    assert c.U <= OracleDB(fa, 0, k, 2, min_k=min_k, wide=True).U
