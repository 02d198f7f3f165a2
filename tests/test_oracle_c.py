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
