"""The frozen vectors of tests/golden/ (written by tests/golden/make_golden.py from oracle/slow_ref.py -- the reference
has no fixtures of its own and cannot run here, so these pin the restatements and the GPU path to each other, not to
Kmers.jar).  CPU: both restatements still produce them.  GPU: the C ABI produces them."""
import glob
import json
import os

import numpy as np
import pytest

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "case_*.json")))


def _load(path):
    g = json.load(open(path))
    g["ref_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["ref_blocks"].items()}
    g["read_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["read_blocks"].items()}
    g["collect_by_read"] = {int(k): v.encode() for k, v in g["collect_by_read"].items()}
    g["read_classifier"] = {int(k): v.encode() for k, v in g["read_classifier"].items()}
    g["collect_by_read_near_n1"] = {int(k): v.encode() for k, v in g["collect_by_read_near_n1"].items()}
    return g


def test_golden_files_exist():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-5] for p in CASES])
def test_restatements_reproduce_golden(path):
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    g = _load(path)
    k, l, dust, run = g["k"], g["key_len"], g["dust"], g["runofsame"]
    fa, fq = g["fasta"].encode(), g["fastq"].encode()
    ref = R.build_ref_db(fa, k, l, dust=dust, runofsame=run)
    reads, names = R.build_read_db(fq, k, l, dust=dust, runofsame=run)
    assert ref == g["ref_blocks"] and reads == g["read_blocks"] and names == g["read_names"]
    db = R.DB([ref])
    lines = R.match_lines(reads, db)
    assert lines == g["match_lines"] and R.taxa_counts(db) == g["taxa_counts"]
    collected = R.collect_by_read(lines)
    assert collected == g["collect_by_read"] and R.read_classifier(collected) == g["read_classifier"]
    near1 = R.near_match_lines(reads, db, 1, False)                  # Matcher -n 1 / -n 2 -j through the restated trie
    assert near1 == g["near_match_lines_n1"] and R.near_match_lines(reads, db, 2, True) == g["near_match_lines_n2_just"]
    assert R.collect_by_read(near1) == g["collect_by_read_near_n1"]
    cd, cr = (-1 if dust is None else dust), (-1 if run is None else run)
    c_ref, c_reads = OracleDB(fa, 0, k, l, dust=cd, runofsame=cr), OracleDB(fq, 1, k, l, dust=cd, runofsame=cr)
    blob, off = c_ref.encode_blocks()
    for key in range(4 ** l):
        assert blob[int(off[key]):int(off[key + 1])] == g["ref_blocks"][key]
    blob, off = c_reads.encode_blocks()
    for key in range(4 ** l):
        assert blob[int(off[key]):int(off[key + 1])] == g["read_blocks"][key]
    assert c_reads.names == g["read_names"]
    hit, _ = c_reads.match(c_ref)
    assert c_reads.match_lines(c_ref, hit).decode().splitlines() == g["match_lines"]
    t, m, q = c_ref.taxacounts()
    assert ["%d\t%d\t%d" % x for x in zip(t, m, q)] == g["taxa_counts"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-5] for p in CASES])
def test_gpu_reproduces_golden(path):
    import kmers_amd as km
    g = _load(path)
    k, l, dust, run = g["k"], g["key_len"], g["dust"], g["runofsame"]
    ctx = km.Context(0)
    ref = km.make_reference_db(ctx, g["fasta"].encode(), k, l, dust=dust, runofsame=run)
    reads, names = km.make_read_db(ctx, g["fastq"].encode(), k, l, dust=dust, runofsame=run)
    for db, want in ((ref, g["ref_blocks"]), (reads, g["read_blocks"])):
        blob, off = db.encode_blocks()
        assert db.meta() == want[-1]
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key], key
    assert names == g["read_names"]
    hit, nh = km.DB([ref]).get_nearest_kmers(reads)
    assert km.match_lines(reads, ref, hit) == g["match_lines"] and nh == len(g["match_lines"])
    assert km.taxa_count_lines(ref) == g["taxa_counts"]
    assert km.collect_by_read_blocks(reads, ref) == g["collect_by_read"]
    call = km.classify_reads(reads, ref, len(names))
    want_calls = {}
    for key, text in g["collect_by_read"].items():
        rs = []
        for line in text.decode().split("\n"):
            r = int(line.split(":", 1)[0])
            if not rs or rs[-1] != r:
                rs.append(r)
        want_calls[key] = "".join("%d\t%d\n" % (r, int(call[r])) for r in rs if call[r] >= 0).encode()
    assert want_calls == g["read_classifier"]
    # inexact search (kmg_match_near + the per-k-mer host steps)
    assert km.near_match_lines(reads, ref, 1, False) == g["near_match_lines_n1"]
    assert km.near_match_lines(reads, ref, 2, True) == g["near_match_lines_n2_just"]
    # the same blocks without CollectByRead's records in between (kmg_classify_reads_hits)
    call2, seen = km.classify_reads(reads, ref, len(names), with_hits=True)
    assert km.read_call_blocks(call2, seen) == g["read_classifier"]
    for d in (ref, reads):
        d.close()
    # the databases again, built in three key ranges (kmg_build_set_key_range): the frozen blocks, range by range
    nkeys = 4 ** l
    for lo, hi in ((0, nkeys // 3), (nkeys // 3, nkeys - 1), (nkeys - 1, nkeys)):
        part_ref = km.make_reference_db(ctx, g["fasta"].encode(), k, l, dust=dust, runofsame=run, key_range=(lo, hi))
        part_reads, part_names = km.make_read_db(ctx, g["fastq"].encode(), k, l, dust=dust, runofsame=run, key_range=(lo, hi))
        assert part_names == g["read_names"]
        for db, want in ((part_ref, g["ref_blocks"]), (part_reads, g["read_blocks"])):
            blob, off = db.encode_blocks()
            for key in range(nkeys):
                assert blob[int(off[key]):int(off[key + 1])] == (want[key] if lo <= key < hi else b""), key
            db.close()
    ctx.close()


# ---- variable-length k-mers (MakeDatabase -k minK -K maxK; Matcher -k L -K L) ----------------------------------------
VAR_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "var_*.json")))


def _load_var(path):
    g = json.load(open(path))
    g["ref_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["ref_blocks"].items()}
    g["read_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["read_blocks"].items()}
    return g


@pytest.mark.parametrize("path", VAR_CASES, ids=[os.path.basename(p)[4:-5] for p in VAR_CASES])
def test_restatements_reproduce_variable_length_golden(path):
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    g = _load_var(path)
    mn, k, l, L = g["min_k"], g["k"], g["key_len"], g["restrict"]
    fa, fq = g["fasta"].encode(), g["fastq"].encode()
    ref = R.build_ref_db(fa, k, l, min_k=mn)
    reads, names = R.build_read_db(fq, k, l, min_k=mn)
    assert ref == g["ref_blocks"] and reads == g["read_blocks"] and names == g["read_names"]
    assert ref[-1][0] == mn and ref[-1][1] == k
    db = R.DB([ref])
    assert R.match_lines(reads, db, min_k=L, max_k=L) == g["match_lines_restricted"] and R.taxa_counts(db) == g["taxa_counts"]
    assert R.near_match_lines(reads, db, 0, False) == g["match_lines_trie"]          # the literal trie (Root.java)
    c_ref, c_reads = OracleDB(fa, 0, k, l, min_k=mn), OracleDB(fq, 1, k, l, min_k=mn)
    for c, want in ((c_ref, g["ref_blocks"]), (c_reads, g["read_blocks"])):
        blob, off = c.encode_blocks()
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key]
    assert c_reads.names == g["read_names"]
    cr, cs = c_ref.restrict(L), c_reads.restrict(L)
    hit, _ = cs.match(cr)
    assert cs.match_lines(cr, hit).decode().splitlines() == g["match_lines_restricted"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", VAR_CASES, ids=[os.path.basename(p)[4:-5] for p in VAR_CASES])
def test_gpu_reproduces_variable_length_golden(path):
    import kmers_amd as km
    g = _load_var(path)
    mn, k, l, L = g["min_k"], g["k"], g["key_len"], g["restrict"]
    ctx = km.Context(0)
    ref = km.make_reference_db(ctx, g["fasta"].encode(), k, l, min_k=mn)
    reads, names = km.make_read_db(ctx, g["fastq"].encode(), k, l, min_k=mn)
    for db, want in ((ref, g["ref_blocks"]), (reads, g["read_blocks"])):
        blob, off = db.encode_blocks()
        assert db.meta() == want[-1]
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key], key
    assert names == g["read_names"]
    assert km.taxa_count_lines(ref) == g["taxa_counts"]
    rr, rs = ref.restrict(L), reads.restrict(L)
    hit, nh = km.DB([rr]).get_nearest_kmers(rs)
    assert km.match_lines(rs, rr, hit) == g["match_lines_restricted"] and nh == len(g["match_lines_restricted"])
    rows, thit = km.match_trie(reads, ref)                                           # Matcher's default -k / -K
    assert km.match_lines(reads, rows, thit, ref_stored=True) == g["match_lines_trie"]
    rows.close()
    # the same in two key ranges
    nkeys = 4 ** l
    for lo, hi in ((0, nkeys // 2), (nkeys // 2, nkeys)):
        part = km.make_reference_db(ctx, g["fasta"].encode(), k, l, key_range=(lo, hi), min_k=mn)
        blob, off = part.encode_blocks()
        for key in range(nkeys):
            assert blob[int(off[key]):int(off[key + 1])] == (g["ref_blocks"][key] if lo <= key < hi else b""), key
        part.close()
    ctx.close()


# ---- k-mers of more than 32 bases (two words per k-mer on the GPU; Sequence.java:15-35,90-115) ------------------------
WIDE_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "wide_*.json")))


def _load_wide(path):
    g = json.load(open(path))
    g["ref_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["ref_blocks"].items()}
    g["read_blocks"] = {int(k): bytes.fromhex(v) for k, v in g["read_blocks"].items()}
    g["collect_by_read"] = {int(k): v.encode() for k, v in g["collect_by_read"].items()}
    g["read_classifier"] = {int(k): v.encode() for k, v in g["read_classifier"].items()}
    return g


def test_wide_golden_files_exist():
    assert len(WIDE_CASES) >= 3


@pytest.mark.parametrize("path", WIDE_CASES, ids=[os.path.basename(p)[5:-5] for p in WIDE_CASES])
def test_restatements_reproduce_wide_golden(path):
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    g = _load_wide(path)
    k, l = g["k"], g["key_len"]
    fa, fq = g["fasta"].encode(), g["fastq"].encode()
    ref = R.build_ref_db(fa, k, l)
    reads, names = R.build_read_db(fq, k, l)
    assert ref == g["ref_blocks"] and reads == g["read_blocks"] and names == g["read_names"]
    db = R.DB([ref])
    lines = R.match_lines(reads, db)
    assert lines == g["match_lines"] and R.taxa_counts(db) == g["taxa_counts"]
    collected = R.collect_by_read(lines)
    assert collected == g["collect_by_read"] and R.read_classifier(collected) == g["read_classifier"]
    assert R.near_match_lines(reads, db, 1, False) == g["near_match_lines_n1"]          # Matcher -n 1 / -n 2 -j: the restated trie
    assert R.near_match_lines(reads, db, 2, True) == g["near_match_lines_n2_just"]
    c_ref, c_reads = OracleDB(fa, 0, k, l), OracleDB(fq, 1, k, l)
    for c, want in ((c_ref, g["ref_blocks"]), (c_reads, g["read_blocks"])):
        blob, off = c.encode_blocks()
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key]
    assert c_reads.names == g["read_names"]
    hit, _ = c_reads.match(c_ref)
    assert c_reads.match_lines(c_ref, hit).decode().splitlines() == g["match_lines"]
    t, m, q = c_ref.taxacounts()
    assert ["%d\t%d\t%d" % x for x in zip(t, m, q)] == g["taxa_counts"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", WIDE_CASES, ids=[os.path.basename(p)[5:-5] for p in WIDE_CASES])
def test_gpu_reproduces_wide_golden(path):
    import kmers_amd as km
    g = _load_wide(path)
    k, l = g["k"], g["key_len"]
    ctx = km.Context(0)
    ref = km.make_reference_db(ctx, g["fasta"].encode(), k, l)
    reads, names = km.make_read_db(ctx, g["fastq"].encode(), k, l)
    for db, want in ((ref, g["ref_blocks"]), (reads, g["read_blocks"])):
        blob, off = db.encode_blocks()
        assert db.meta() == want[-1]
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key], key
    assert names == g["read_names"]
    hit, nh = km.DB([ref]).get_nearest_kmers(reads)
    assert km.match_lines(reads, ref, hit) == g["match_lines"] and nh == len(g["match_lines"])
    assert km.taxa_count_lines(ref) == g["taxa_counts"]
    assert km.collect_by_read_blocks(reads, ref) == g["collect_by_read"]
    call2, seen = km.classify_reads(reads, ref, len(names), with_hits=True)
    assert km.read_call_blocks(call2, seen) == g["read_classifier"]
    # inexact search on two words (kmg_match_near's walk + the per-k-mer host steps)
    assert km.near_match_lines(reads, ref, 1, False) == g["near_match_lines_n1"]
    assert km.near_match_lines(reads, ref, 2, True) == g["near_match_lines_n2_just"]
    for d in (ref, reads):
        d.close()
    ctx.close()


# ---- ... and of several lengths (MakeDatabase -K 40 with the default -k 24; Matcher -k L -K L) ------------------------------
WIDE_VAR_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "widevar_*.json")))


def test_wide_var_golden_files_exist():
    assert len(WIDE_VAR_CASES) >= 2


@pytest.mark.parametrize("path", WIDE_VAR_CASES, ids=[os.path.basename(p)[8:-5] for p in WIDE_VAR_CASES])
def test_restatements_reproduce_wide_variable_length_golden(path):
    from oracle import slow_ref as R
    from oracle.c_oracle import OracleDB
    g = _load_var(path)
    mn, k, l, L = g["min_k"], g["k"], g["key_len"], g["restrict"]
    fa, fq = g["fasta"].encode(), g["fastq"].encode()
    ref = R.build_ref_db(fa, k, l, min_k=mn)
    reads, names = R.build_read_db(fq, k, l, min_k=mn)
    assert ref == g["ref_blocks"] and reads == g["read_blocks"] and names == g["read_names"]
    db = R.DB([ref])
    assert R.match_lines(reads, db, min_k=L, max_k=L) == g["match_lines_restricted"] and R.taxa_counts(db) == g["taxa_counts"]
    c_ref, c_reads = OracleDB(fa, 0, k, l, min_k=mn), OracleDB(fq, 1, k, l, min_k=mn)
    for c, want in ((c_ref, g["ref_blocks"]), (c_reads, g["read_blocks"])):
        blob, off = c.encode_blocks()
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key]
    cr, cs = c_ref.restrict(L), c_reads.restrict(L)
    hit, _ = cs.match(cr)
    assert cs.match_lines(cr, hit).decode().splitlines() == g["match_lines_restricted"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", WIDE_VAR_CASES, ids=[os.path.basename(p)[8:-5] for p in WIDE_VAR_CASES])
def test_gpu_reproduces_wide_variable_length_golden(path):
    import kmers_amd as km
    g = _load_var(path)
    mn, k, l, L = g["min_k"], g["k"], g["key_len"], g["restrict"]
    ctx = km.Context(0)
    ref = km.make_reference_db(ctx, g["fasta"].encode(), k, l, min_k=mn)
    reads, names = km.make_read_db(ctx, g["fastq"].encode(), k, l, min_k=mn)
    for db, want in ((ref, g["ref_blocks"]), (reads, g["read_blocks"])):
        blob, off = db.encode_blocks()
        assert db.meta() == want[-1]
        for key in range(4 ** l):
            assert blob[int(off[key]):int(off[key + 1])] == want[key], key
    assert names == g["read_names"]
    assert km.taxa_count_lines(ref) == g["taxa_counts"]
    rref, rreads = ref.restrict(L), reads.restrict(L)
    hit, nh = km.DB([rref]).get_nearest_kmers(rreads)
    assert km.match_lines(rreads, rref, hit) == g["match_lines_restricted"] and nh == len(g["match_lines_restricted"])
    for d in (ref, reads, rref, rreads):
        d.close()
    ctx.close()
