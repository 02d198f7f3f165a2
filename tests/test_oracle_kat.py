"""Known-answer vectors for the CPU oracle (oracle/slow_ref.py).

The reference has no tests or fixtures, so these are the hand-derived vectors of
SURVEY.md Appendix B; each cites the reference lines it was derived from.  Parity vs Kmers.jar
itself is UNPINNED (no JVM in the build container).
"""
import pytest

from oracle import slow_ref as R


def hx(s):
    return bytes.fromhex(s.replace(" ", ""))


def K(s):
    return R.Kmer(list(s.encode()))


def test_b1_b3_packing():  # Kmer.java:90-97, Sequence.java:90-115
    assert K("ACGT").compressed_bytes() == hx("04 1E")
    assert K("ACGTA").compressed_bytes() == hx("05 1E 00")
    assert K("ACGTG").compressed_bytes() == hx("05 1E C0")
    assert K("G" * 31).compressed_bytes() == hx("1F FF FF FF FF FF FF FF FC")


def test_b4_key():  # Kmer.java:47-55
    assert K("ACGTACGGT").key(6) == 481


def test_b5_rc():  # Kmer.java:142-208
    k = K("ACGTA")
    k.inplace_rc()
    assert str(k) == "TACGT"
    k = K("ACGT")
    k.inplace_rc()
    assert str(k) == "ACGT"


def test_b6_is_own_rc():  # Kmer.java:210-225 -- palindrome test, not RC identity
    assert K("ACCA").is_own_rc()
    assert not K("ACGT").is_own_rc()
    assert not K("AAAAA").is_own_rc()


def test_b7_order():  # Sequence.java:165-187
    assert K("ACGT").compare_to(K("ACGG")) < 0
    assert K("ACG").compare_to(K("ACGA")) < 0
    assert K("ACGT").compare_to(K("ACGT")) == 0


def test_lowercase_and_invalid():  # Base.java:22-41
    assert str(K("acgt")) == "ACGT"
    with pytest.raises(R.InvalidBaseException):
        K("ACNT")


def test_roundtrip_compressed():  # Sequence.java:236-257
    for s in ["A", "ACGT", "ACGTA", "GATTACAGATTACAGATTACAGATTACAGAT"]:
        assert str(R.Kmer.create_from_compressed(K(s).compressed_bytes())) == s


def test_b8_temp_record():  # KmerWithDataDataType.java:62-72, ReadPosDataType.java:22-28
    assert R.kwd_compress(K("ACGT"), R.int_compress(9606)) == hx("04 1E 00 00 25 86")
    assert R.kwd_compress(K("ACGT"), R.readpos_compress((3, 7))) == hx("04 1E 00 00 00 03 00 07")


def test_b9_count_codec():  # CountDataType.java:31-46
    assert R.count_compress({9606: 1}) == hx("00 01 00 00 25 86 01")
    assert R.count_compress({9606: 2, 10090: 1}) == hx("00 02 00 00 25 86 02 00 00 27 6A 01")
    assert R.count_compress({7: 300}) == hx("00 01 00 00 00 07 FF")


def test_b10_prefix_delta_block():  # FileCreator.java:312-347, KmerFile.java:200-243
    fc = R.FileCreator(2, "ref", False)
    fc.add_kmers([(K("ACGG"), 9606), (K("ACGT"), 9606), (K("ACGG"), 10090), (K("ACGG"), 9606)], 4, 4)
    block, _ = fc.make_key(K("ACGT").key(2))
    assert block == hx("00 04 1E 00 01 00 00 25 86 01  01 1F 00 02 00 00 25 86 02 00 00 27 6A 01")
    dec = R.decode_block(block, "ref")
    assert [(str(k), dict(d)) for k, d in dec] == [("ACGT", {9606: 1}), ("ACGG", {9606: 2, 10090: 1})]


def test_b11_meta():  # FileCreator.java:153-164
    fc = R.FileCreator(6, "ref", True)
    fc.min_k = fc.max_k = 31
    assert fc.meta() == hx("1F 1F 06 00 00 00 01 00 00 08 00 01")
    fc = R.FileCreator(6, "read", False)
    fc.min_k = fc.max_k = 31
    assert fc.meta() == hx("1F 1F 06 00 00 00 02 00 00 01 01 00 00 08 04 00")
    assert R.parse_meta(fc.meta()) == (31, 31, 6, [257, 2052], False)


B12_FA = b">7\nACGTNACGTA\n"
B13_FQ = b"@r1\nACGTA\n+\nIIIII\n@r2\nNNNNN\n+\nIIIII\n@r3\nTTTTT\n+\nIIIII\n"


def test_b12_fasta_build():  # KmersFromFile.java:84-208, FileCreator.java
    kw = R.kmers_from_file(B12_FA, R.fa_state_changer(), 4, 4, R.fa_to_refdb_mapper)
    assert [(str(k), d) for k, d in kw] == [("ACGT", 7), ("ACGT", 7), ("CGTA", 7)]
    blocks = R.build_ref_db(B12_FA, 4, 2)
    assert blocks[-1] == hx("04 04 02 00 00 00 01 00 00 08 00 01")
    assert blocks[1] == hx("00 04 1E 00 01 00 00 00 07 04")
    assert blocks[7] == hx("00 04 78 00 01 00 00 00 07 01")
    assert blocks[8] == hx("00 04 87 00 01 00 00 00 07 01")
    assert sorted(blocks) == [-1] + list(range(16))
    assert all(blocks[k] == b"" for k in range(16) if k not in (1, 7, 8))


def test_b13_fastq_build():  # ReadIDMapping.java:14-23, KmersFromFileStateChanger.java:60-68
    blocks, names = R.build_read_db(B13_FQ, 4, 2)
    assert names == ["r1", "r3"]
    assert blocks[-1] == hx("04 04 02 00 00 00 02 00 00 01 01 00 00 08 04 00")
    assert blocks[10] == hx("00 04 AA 00 00 00 02 00 00 00 01 00 00 00 00 00 01 00 01")
    allk = []
    for key in range(16):
        allk += [(str(k), d) for k, d in R.decode_block(blocks[key], "read")]
    assert allk == [("ACGT", [(0, 0)]), ("CGTA", [(0, 1)]), ("TTTT", [(1, 0), (1, 1)])]


def test_b14_match_b15_taxacounts():  # DB.java:76-137, Matcher.java:235-250, TaxaCounts.java
    ref = R.build_ref_db(B12_FA, 4, 2)
    reads, _ = R.build_read_db(B13_FQ, 4, 2)
    db = R.DB([ref])
    assert R.match_lines(reads, db) == ["ACGT\t0:0\t-|7x4", "CGTA\t0:1\t-|7x1"]
    assert R.taxa_counts(db) == ["7\t3\t3"]


def test_b16_dust():  # Dust.java:12-30
    assert R.dust_test(K("A" * 31), 14)
    assert not R.dust_test(K("A" * 31), 13)


def test_quirk_no_trailing_newline_duplicates_last_kmer():
    kw = R.kmers_from_file(b">7\nACGTA", R.fa_state_changer(), 4, 4, R.fa_to_refdb_mapper)
    assert [str(k) for k, _ in kw] == ["ACGT", "CGTA", "CGTA"]


def test_quirk_wrapped_fasta_drops_every_other_line():
    kw = R.kmers_from_file(b">7\nACGTA\nCCCCC\nGGGGG\n", R.fa_state_changer(), 4, 4, R.fa_to_refdb_mapper)
    assert [str(k) for k, _ in kw] == ["ACGT", "CGTA", "GGGG", "GGGG"]


def test_quirk_read_index_reuse_and_skip():  # ReadIDMapping.java:14-23
    fq = b"@a\nACGTA\n+\nIIIII\n@b\nNN\n+\nII\n@a\nCCCCC\n+\nIIIII\n"
    blocks, names = R.build_read_db(fq, 4, 2)
    assert names == ["a"]          # 'b' emits nothing; second 'a' equals the last *emitting* id
    allk = []
    for key in range(16):
        allk += [(str(k), d) for k, d in R.decode_block(blocks[key], "read")]
    assert ("CCCC", [(0, 0), (0, 1)]) in allk


def test_multi_file_db_counts_add_without_saturation():  # SURVEY App. C.9, DB.java:99-109
    fa = b">5\n" + b"ACGT" * 80 + b"\n"        # ACGT occurs 80x fwd + 80x as own RC = 160... per file
    a = R.build_ref_db(fa, 4, 2)
    db = R.DB([a, a])
    got = dict((str(k), d) for k, d in db.kmers(R.Kmer(list(b"ACGT")).key(2)))
    one = dict((str(k), d) for k, d in R.DB([a]).kmers(R.Kmer(list(b"ACGT")).key(2)))
    assert got["ACGT"][5] == 2 * one["ACGT"][5]


def test_collect_by_read_and_read_classifier_hand_example():
    """CollectByRead.java:105-124,159-184 and ReadClassifier.java:95-220 on five match lines worked by hand:
    counts are capped at 255 on the way through the temp file (CountDataType.java:40), records are filed under
    (read / 1000) * 1000 and ordered by (read, (short) pos), a read is called only if all its single-taxon hits
    agree and every hit's taxon set contains that taxon."""
    lines = ["AAC\t0:0|0:1\t-|5x2|7x300",
             "ACC\t0:2|1001:0\t-|5x1",
             "CCC\t1001:5\t-|7x1",
             "GGG\t2:-32768|2:7\t-|9x1|10x1",
             "GGT\t3:4\t-|9x1|10x1"]
    blocks = R.collect_by_read(lines)
    assert blocks == {
        0: b"0:0\t0:5x2|7x255\n0:1\t0:5x2|7x255\n0:2\t0:5x1\n2:-32768\t0:9x1|10x1\n2:7\t0:9x1|10x1\n3:4\t0:9x1|10x1",
        1000: b"1001:0\t0:5x1\n1001:5\t0:7x1"}
    # read 0: {5,7} {5,7} {5} -> 5; read 2, 3: no single-taxon hit -> none; read 1001: {5} vs {7} -> none
    assert R.read_classifier(blocks) == {0: b"0\t5\n", 1000: b""}
    # a single-taxon hit that another hit's set does not contain
    assert R.read_classifier({0: b"4:0\t0:5x1\n4:1\t0:6x1|7x1"}) == {0: b""}
    assert R.read_classifier({0: b"4:0\t0:5x1\n4:1\t0:5x3|7x1\n5:0\t0:8x1"}) == {0: b"4\t5\n5\t8\n"}


def _brute_near_lines(R, reads, ref_files, max_diff, just, k):
    """Independent of the trie: every reference k-mer within max_diff differences, by comparing all pairs."""
    db = R.DB(ref_files)
    l = db.key_length
    refs = [(km, data) for key in range(4 ** l) for km, data in db.kmers(key)]
    lines = []
    for key in range(4 ** l):
        for kmer, sdata in R.decode_block(reads.get(key, b""), "read"):
            cand = []
            for rk, data in refs:                                    # ascending reference k-mer
                d = sum(1 for a, b in zip(kmer.chars, rk.chars) if a != b)
                if d <= max_diff:
                    cand.append((rk, data, d))
            if just and cand:
                best = min(c[2] for c in cand)
                cand = [c for c in cand if c[2] == best]
            kept = {}
            for rk, data, d in cand:
                me, rc = tuple(rk.chars), tuple(c ^ 2 for c in reversed(rk.chars))
                if rc in kept:
                    if d < kept[rc][1]:
                        del kept[rc]
                        kept[me] = (data, d)
                else:
                    kept[me] = (data, d)
            ents = []
            for me, (data, d) in kept.items():
                diffs = [(i, me[i]) for i in range(k) if me[i] != kmer.chars[i]]
                ents.append(((len(diffs), diffs), data))
            ents.sort(key=lambda e: e[0])
            if ents:
                lines.append("%s\t%s\t%s" % (kmer, "|".join("%d:%d" % rp for rp in sorted(sdata)), " ".join(
                    "%s|%s" % (R.kmer_diff_text(df), "|".join("%dx%d" % (t, data[t]) for t in sorted(data))) for (_, df), data in ents)))
    return lines


def _near_inputs(seed, k):
    import numpy as np
    rng = np.random.default_rng(seed)
    al = np.frombuffer(b"ACGT", np.uint8)
    base = rng.choice(al, size=260)
    genomes = []
    for g in range(3):                                   # three taxa that differ in a few bases: near matches abound
        x = base.copy()
        for j in rng.choice(260, size=12, replace=False).tolist():
            x[j] = int(rng.choice(al))
        genomes.append(bytes(x.tolist()))
    pal = b"ACGTTGCAACGT"[:k] if k <= 12 else b""
    fa1 = b">11\n%s\n>12\n%s\n" % (genomes[0], genomes[1] + pal)
    fa2 = b">13\n%s\n>11\n%s\n" % (genomes[2], genomes[0][:80])
    recs = []
    for i in range(30):
        o = int(rng.integers(0, 260 - 40))
        s_ = bytearray(genomes[int(rng.integers(3))][o:o + 40])
        for j in np.nonzero(rng.random(40) < 0.04)[0].tolist():
            s_[j] = int(rng.choice(al))
        recs.append(b"@q%d\n%s\n+\n%s\n" % (i, bytes(s_), b"I" * 40))
    return fa1, fa2, b"".join(recs)


def test_trie_restatement_equals_all_pairs_search():
    """Root / ClosestInfoCollector restated (oracle/slow_ref.py) against a search that knows nothing of tries: for
    k-mers of one length the trie returns the stored k-mers within maxDiff differences, no more and no fewer --
    one database file and two (unmerged insertion, DB.java:155-158), -j on and off, one and two differences, even
    k (where a k-mer can be its own reverse complement's neighbour)."""
    from oracle import slow_ref as R
    for k, l, seed in ((9, 2, 5), (8, 2, 6), (11, 3, 7)):
        fa1, fa2, fq = _near_inputs(seed, k)
        reads, _ = R.build_read_db(fq, k, l)
        files = [R.build_ref_db(fa1, k, l), R.build_ref_db(fa2, k, l)]
        for ref_files in (files[:1], files):
            for max_diff in (1, 2):
                for just in (False, True):
                    got = R.near_match_lines(reads, R.DB(ref_files), max_diff, just)
                    assert got == _brute_near_lines(R, reads, ref_files, max_diff, just, k), (k, len(ref_files), max_diff, just)
                    assert any(" " in ln.split("\t")[2] for ln in got) or just
        exact = R.match_lines(reads, R.DB(files))
        assert [ln for ln in R.near_match_lines(reads, R.DB(files), 1, True) if ln.split("\t")[2].startswith("-|")] == exact
