#!/usr/bin/env python3
"""Writes the golden vectors tests/golden/case_*.json.

The reference (danielmoney/kmers) ships no tests, fixtures or golden files and cannot be run here (Java, no JVM;
its prebuilt jar is never executed), so these vectors do NOT come from the reference: they are the outputs of
the literal restatement oracle/slow_ref.py -- every function of which cites the reference lines it follows -- on
small seeded inputs, frozen so that neither the restatements nor the GPU path can drift unnoticed.  Parity against
Kmers.jar itself stays unpinned.  At generation time each case is also checked against the
independent C restatement (oracle/kmers_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import slow_ref as R  # noqa: E402
from oracle.c_oracle import OracleDB  # noqa: E402

CASES = [
    # name, k, key_len, seed, #genomes, genome length, #reads, read length, fasta alphabet, dust, runofsame, paired headers
    ("basic_k11", 11, 2, 101, 3, 300, 40, 60, b"ACGT", None, None, False),
    ("messy_k31", 31, 3, 102, 4, 400, 50, 90, b"ACGTacgtN", None, None, True),
    ("even_k8_palindromes", 8, 2, 103, 3, 250, 30, 50, b"AACGTT", None, None, False),
    ("filters_k12", 12, 3, 104, 3, 300, 40, 70, b"AAACGT", 1, 3, False),
]


def make_inputs(seed, ng, glen, nr, rlen, alphabet, paired):
    rng = np.random.default_rng(seed)
    al = np.frombuffer(alphabet, np.uint8)
    genomes = [bytes(rng.choice(al, size=glen).tolist()) for _ in range(ng)]
    genomes[-1] = genomes[0][:glen // 2] + genomes[-1][glen // 2:]          # shared k-mers between two taxa
    fa = b"".join(b">%d\n%s\n" % (500 + i, g) for i, g in enumerate(genomes)) + b">77\n\n"   # + an empty sequence
    recs = []
    for i in range(nr):
        g = genomes[int(rng.integers(ng))].upper().replace(b"N", b"A")
        o = int(rng.integers(0, glen - rlen))
        s = bytearray(g[o:o + rlen])
        for j in np.nonzero(rng.random(rlen) < 0.02)[0].tolist():
            s[j] = int(rng.choice(np.frombuffer(b"ACGTN", np.uint8)))
        name = b"r%d" % (i - (i & 1) if paired else i)
        recs.append(b"@%s\n%s\n+\n%s\n" % (name, bytes(s), b"I" * rlen))
    return fa, b"".join(recs)


# variable-length k-mers (MakeDatabase -k minK -K maxK; the reference's defaults are -k 24 -K 32, MakeDatabase.java:106-107)
VAR_CASES = [
    # name, minK, maxK, key_len, restricted length of the match (Matcher -k L -K L), seed, fasta alphabet
    ("default_k24_32", 24, 32, 3, 27, 201, b"ACGTACGTACGTACGTN"),
    ("short_k3_6", 3, 6, 2, 4, 202, b"AACCGTN"),
]


def main_var():
    for name, mn, mx, l, L, seed, alphabet in VAR_CASES:
        fa, fq = make_inputs(seed, 3, 260, 30, 70, alphabet, True)
        first = fa.split(b"\n")[1]
        fa += b">9\n%s\n>9\n%s\n" % (first[:mx - 2], first[5:5 + mn])      # sequences of maxK-2 and of minK bases
        ref = R.build_ref_db(fa, mx, l, min_k=mn)
        reads, names = R.build_read_db(fq, mx, l, min_k=mn)
        db = R.DB([ref])
        lines = R.match_lines(reads, db, min_k=L, max_k=L)
        counts = R.taxa_counts(db)
        c_ref, c_reads = OracleDB(fa, 0, mx, l, min_k=mn), OracleDB(fq, 1, mx, l, min_k=mn)
        assert c_ref.encode_blocks()[0] == b"".join(ref[key] for key in range(4 ** l)), name
        assert c_reads.encode_blocks()[0] == b"".join(reads[key] for key in range(4 ** l)) and c_reads.names == names, name
        cr, cs = c_ref.restrict(L), c_reads.restrict(L)
        hit, _ = cs.match(cr)
        assert cs.match_lines(cr, hit).decode().splitlines() == lines and len(lines) > 20, name
        t, m, q = c_ref.taxacounts()
        assert ["%d\t%d\t%d" % x for x in zip(t, m, q)] == counts, name
        # Matcher with its default -k / -K (the files' own lengths): the literal trie, no differences
        trie = R.near_match_lines(reads, db, 0, False)
        assert len(trie) > 20 and len({len(x.split("\t")[0]) for x in trie}) > 1, name
        out = {
            "comment": "outputs of oracle/slow_ref.py (restatement of the reference), not of Kmers.jar; see make_golden.py",
            "match_lines_trie": trie,
            "min_k": mn, "k": mx, "key_len": l, "restrict": L, "fasta": fa.decode(), "fastq": fq.decode(),
            "ref_blocks": {str(key): ref[key].hex() for key in sorted(ref)},
            "read_blocks": {str(key): reads[key].hex() for key in sorted(reads)},
            "read_names": names, "match_lines_restricted": lines, "taxa_counts": counts,
        }
        path = os.path.join(ROOT, "tests", "golden", "var_%s.json" % name)
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, len(lines), "match lines,", os.path.getsize(path), "bytes")


def main():
    for name, k, l, seed, ng, glen, nr, rlen, alphabet, dust, run, paired in CASES:
        fa, fq = make_inputs(seed, ng, glen, nr, rlen, alphabet, paired)
        ref = R.build_ref_db(fa, k, l, dust=dust, runofsame=run)
        reads, names = R.build_read_db(fq, k, l, dust=dust, runofsame=run)
        db = R.DB([ref])
        lines = R.match_lines(reads, db)
        counts = R.taxa_counts(db)
        collected = R.collect_by_read(lines)
        calls = R.read_classifier(collected)
        near1 = R.near_match_lines(reads, db, 1, False)              # Matcher -n 1 / -n 2 -j: the restated trie
        near2j = R.near_match_lines(reads, db, 2, True)
        # the independent C restatement must agree before anything is frozen
        c_ref = OracleDB(fa, 0, k, l, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run)
        c_reads = OracleDB(fq, 1, k, l, dust=-1 if dust is None else dust, runofsame=-1 if run is None else run)
        assert c_ref.encode_blocks()[0] == b"".join(ref[key] for key in range(4 ** l)), name
        assert c_reads.encode_blocks()[0] == b"".join(reads[key] for key in range(4 ** l)) and c_reads.names == names, name
        hit, _ = c_reads.match(c_ref)
        assert c_reads.match_lines(c_ref, hit).decode().splitlines() == lines, name
        out = {
            "comment": "outputs of oracle/slow_ref.py (restatement of the reference), not of Kmers.jar; see make_golden.py",
            "k": k, "key_len": l, "dust": dust, "runofsame": run,
            "fasta": fa.decode(), "fastq": fq.decode(),
            "ref_blocks": {str(key): ref[key].hex() for key in sorted(ref)},
            "read_blocks": {str(key): reads[key].hex() for key in sorted(reads)},
            "read_names": names, "match_lines": lines, "taxa_counts": counts,
            "collect_by_read": {str(key): collected[key].decode() for key in sorted(collected)},
            "read_classifier": {str(key): calls[key].decode() for key in sorted(calls)},
            "near_match_lines_n1": near1, "near_match_lines_n2_just": near2j,
            "collect_by_read_near_n1": {str(key): v.decode() for key, v in sorted(R.collect_by_read(near1).items())},
        }
        path = os.path.join(ROOT, "tests", "golden", "case_%s.json" % name)
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, len(lines), "match lines,", os.path.getsize(path), "bytes")


# k-mers of more than 32 bases (the reference's Sequence is a byte array of any length, Sequence.java:15-35,90-115)
WIDE_CASES = [
    # name, k, key_len, seed, #genomes, genome length, #reads, read length, fasta alphabet, paired headers
    ("k40", 40, 3, 301, 3, 400, 40, 110, b"ACGT" * 8 + b"acgtN", True),
    ("k64_even", 64, 2, 302, 3, 420, 40, 120, b"ACGT", False),
    ("k33_low_entropy", 33, 2, 303, 3, 300, 30, 90, b"AAT", False),
]


def main_wide():
    for name, k, l, seed, ng, glen, nr, rlen, alphabet, paired in WIDE_CASES:
        fa, fq = make_inputs(seed, ng, glen, nr, rlen, alphabet, paired)
        if k % 2 == 0:                           # a palindrome of even length has no reverse complement record (Kmer.isOwnRC)
            half = fa.split(b"\n")[1][:k // 2].upper().replace(b"N", b"A")
            fa += b">9\n%s\n" % (half + half[::-1])
        ref = R.build_ref_db(fa, k, l)
        reads, names = R.build_read_db(fq, k, l)
        db = R.DB([ref])
        lines = R.match_lines(reads, db)
        counts = R.taxa_counts(db)
        collected = R.collect_by_read(lines)
        calls = R.read_classifier(collected)
        c_ref, c_reads = OracleDB(fa, 0, k, l), OracleDB(fq, 1, k, l)
        assert c_ref.encode_blocks()[0] == b"".join(ref[key] for key in range(4 ** l)), name
        assert c_reads.encode_blocks()[0] == b"".join(reads[key] for key in range(4 ** l)) and c_reads.names == names, name
        hit, _ = c_reads.match(c_ref)
        assert c_reads.match_lines(c_ref, hit).decode().splitlines() == lines and len(lines) > 20, name
        near1 = R.near_match_lines(reads, db, 1, False)              # Matcher -n 1 / -n 2 -j: the restated trie
        near2j = R.near_match_lines(reads, db, 2, True)
        assert len(near1) >= len(lines), name
        out = {
            "comment": "outputs of oracle/slow_ref.py (restatement of the reference), not of Kmers.jar; see make_golden.py",
            "near_match_lines_n1": near1, "near_match_lines_n2_just": near2j,
            "k": k, "key_len": l, "fasta": fa.decode(), "fastq": fq.decode(),
            "ref_blocks": {str(key): ref[key].hex() for key in sorted(ref)},
            "read_blocks": {str(key): reads[key].hex() for key in sorted(reads)},
            "read_names": names, "match_lines": lines, "taxa_counts": counts,
            "collect_by_read": {str(key): collected[key].decode() for key in sorted(collected)},
            "read_classifier": {str(key): calls[key].decode() for key in sorted(calls)},
        }
        path = os.path.join(ROOT, "tests", "golden", "wide_%s.json" % name)
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, len(lines), "match lines,", os.path.getsize(path), "bytes")


# ... and of several lengths (MakeDatabase -K 40 with the default -k 24)
WIDE_VAR_CASES = [
    # name, minK, maxK, key_len, restricted length of the match (Matcher -k L -K L), seed, fasta alphabet
    ("var_k24_40", 24, 40, 3, 35, 401, b"ACGTACGTACGTACGTACGTACGTN"),
    ("var_k30_64", 30, 64, 2, 31, 402, b"ACGT"),
]


def main_wide_var():
    for name, mn, mx, l, L, seed, alphabet in WIDE_VAR_CASES:
        fa, fq = make_inputs(seed, 3, 320, 30, 110, alphabet, True)
        first = fa.split(b"\n")[1]
        fa += b">9\n%s\n>9\n%s\n" % (first[:mx - 2], first[5:5 + mn])      # sequences of maxK-2 and of minK bases
        ref = R.build_ref_db(fa, mx, l, min_k=mn)
        reads, names = R.build_read_db(fq, mx, l, min_k=mn)
        db = R.DB([ref])
        lines = R.match_lines(reads, db, min_k=L, max_k=L)
        counts = R.taxa_counts(db)
        c_ref, c_reads = OracleDB(fa, 0, mx, l, min_k=mn), OracleDB(fq, 1, mx, l, min_k=mn)
        assert c_ref.encode_blocks()[0] == b"".join(ref[key] for key in range(4 ** l)), name
        assert c_reads.encode_blocks()[0] == b"".join(reads[key] for key in range(4 ** l)) and c_reads.names == names, name
        cr, cs = c_ref.restrict(L), c_reads.restrict(L)
        hit, _ = cs.match(cr)
        assert cs.match_lines(cr, hit).decode().splitlines() == lines and len(lines) > 20, name
        out = {
            "comment": "outputs of oracle/slow_ref.py (restatement of the reference), not of Kmers.jar; see make_golden.py",
            "min_k": mn, "k": mx, "key_len": l, "restrict": L, "fasta": fa.decode(), "fastq": fq.decode(),
            "ref_blocks": {str(key): ref[key].hex() for key in sorted(ref)},
            "read_blocks": {str(key): reads[key].hex() for key in sorted(reads)},
            "read_names": names, "match_lines_restricted": lines, "taxa_counts": counts,
        }
        path = os.path.join(ROOT, "tests", "golden", "widevar_%s.json" % name[4:])
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, len(lines), "match lines,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    import sys
    if "wide" not in sys.argv[1:]:
        main()
        main_var()
    main_wide()
    main_wide_var()
