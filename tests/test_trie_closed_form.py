"""The closed form kmg_match_trie computes (csrc/match.h trie_find_kernel), written out in Python over the
oracle's decoded blocks, against the literal restatement of Root.addKmer / closestKmers (oracle/slow_ref.py Root,
Root.java:29-170) at maxDiff = 0, on random variable-length databases with heavy prefix sharing.  CPU only: this pins
the derivation; the kernel itself is held against the literal restatement by the -m gpu tests."""
import random

from oracle import slow_ref as R


STATS = {'lines': 0, 'm_lt_T': 0, 'entries': 0}


def closed_form_lines(search_blocks, ref_blocks):
    min_k, max_k, key_len, ids, rc = R.parse_meta(search_blocks[-1])
    kind = "read" if ids == R.READ_ID else "ref"
    lines = []
    for key in range(4 ** key_len):
        group = R.decode_block(search_blocks.get(key, b""), kind)
        ents = R.decode_block(ref_blocks.get(key, b""), "ref")      # (value, length) order
        strs = [tuple(k.chars) for k, _ in ents]
        for kmer, sdata in group:
            t = tuple(kmer.chars)
            T = len(t)
            # first stored string >= t in prefix-first lexicographic order
            i0 = None
            for i, x in enumerate(strs):
                if x >= t:        # python tuple order: prefix first, A = 0 smallest
                    i0 = i
                    break
            if i0 is None or strs[i0][:T] != t:
                continue
            x0 = strs[i0]
            L = 0
            if i0 > 0:
                p = strs[i0 - 1]
                while L < min(len(p), len(x0)) and p[L] == x0[L]:
                    L += 1
            m = max(min_k, L + 1)
            data = {}
            STATS['lines'] += 1
            STATS['m_lt_T'] += m < T
            for x, (_, d) in zip(strs[i0:], ents[i0:]):
                if x[:m] != x0[:m]:
                    break
                STATS['entries'] += 1
                for tx, c in d.items():
                    data[tx] = data.get(tx, 0) + c
            if kind == "read":
                s = "|".join("%d:%d" % (r_, p_) for r_, p_ in sorted(sdata))
            else:
                s = "|".join("%dx%d" % (tx, sdata[tx]) for tx in sorted(sdata))
            lines.append("%s\t%s\t-|%s" % (kmer, s, "|".join("%dx%d" % (tx, data[tx]) for tx in sorted(data))))
    return lines


def _run(seeds):
    bad = 0
    for seed in range(seeds):
        rng = random.Random(seed)
        alpha = rng.choice(["ACGT", "AC", "AAAC", "ACGTN", "AG"])
        max_k = rng.randint(2, 9)
        min_k = rng.randint(1, max_k)
        key_len = rng.randint(1, min(min_k, 3))
        fa = "".join(">%d\n%s\n" % (rng.randint(1, 5), "".join(rng.choice(alpha) for _ in range(rng.randint(0, 60)))) for _ in range(rng.randint(1, 6)))
        fq = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in
                     enumerate("".join(rng.choice(alpha) for _ in range(rng.randint(0, 40))) for _ in range(rng.randint(1, 8))))
        ref = R.build_ref_db(fa.encode(), max_k, key_len, min_k=min_k)
        # the search side may be narrower than the database
        s_min = rng.randint(min_k, max_k)
        s_max = rng.randint(s_min, max_k)
        if rng.random() < 0.5:
            s_min, s_max = min_k, max_k
        if key_len > s_min:
            continue
        reads, _ = R.build_read_db(fq.encode(), s_max, key_len, min_k=s_min)
        db = R.DB([ref])
        want = R.near_match_lines(reads, db, 0, False)
        got = closed_form_lines(reads, ref)
        if want != got:
            bad += 1
            print("seed", seed, "alpha", alpha, "k", min_k, max_k, "search", s_min, s_max, "key", key_len, "MISMATCH", len(want), len(got))
            for a, b in zip(want, got):
                if a != b:
                    print("  want", a)
                    print("  got ", b)
                    break
    print("seeds", seeds, "bad", bad, STATS)
    return bad


def test_closed_form_equals_literal_trie():
    STATS.update(lines=0, m_lt_T=0, entries=0)
    assert _run(120) == 0
    # the cases must bite: answers summed over several stored k-mers, at a depth below the search k-mer's length
    assert STATS["lines"] > 1000 and STATS["m_lt_T"] > 200 and STATS["entries"] > 2 * STATS["lines"]


# ---- several files that share keys ------------------------------------------------------------------------------------
# The reference puts a key's k-mers into its trie file by file (DB.java:154-158), each file in ascending order.  With
# more than one file the insertion is no longer in the trie's preorder: a later file's k-mer can split an edge that an
# earlier file's k-mer made, after which BOTH pieces hold that k-mer's data object (n1.data = n.data, Root.java:79-80),
# and a k-mer that then walks through both pieces is merged into the object twice.  In closed form: the object of x --
# the earliest inserted stored k-mer that has t as a prefix -- ends as
#     D_x + sum over boundary depths e in [max(L_x + 1, shortest), |x|] of
#           sum of D_z over the k-mers z inserted at or after the boundary's creator with lcp(x, z) >= e,
# L_x = the longest prefix x shares with anything inserted before it; the boundary depths are |x| (made by x itself) and
# every lcp(x, y) < |x| of a k-mer y inserted after x (made by the first such y).
STATS_F = {'lines': 0, 'multi': 0, 'twice': 0}


def _lcp(a, b):
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    return n


def closed_form_lines_files(search_blocks, files):
    min_k, max_k, key_len, ids, rc = R.parse_meta(search_blocks[-1])
    kind = "read" if ids == R.READ_ID else "ref"
    lines = []
    for key in range(4 ** key_len):
        group = R.decode_block(search_blocks.get(key, b""), kind)
        ents = [[(tuple(k.chars), d) for k, d in R.decode_block(f.get(key, b""), "ref")] for f in files]
        for kmer, sdata in group:
            t = tuple(kmer.chars)
            T = len(t)
            # x: the earliest inserted k-mer with prefix t = first file that holds one, its smallest there
            f0 = i0 = None
            for fi, es in enumerate(ents):
                for i, (s, _) in enumerate(es):
                    if s[:T] == t:
                        f0, i0 = fi, i
                        break
                if f0 is not None:
                    break
            if f0 is None:
                continue
            x, dx = ents[f0][i0]
            before = [s for fi in range(f0) for s, _ in ents[fi]] + [s for s, _ in ents[f0][:i0]]
            L = max([_lcp(x, w) for w in before], default=0)
            m = max(min_k, L + 1)
            after = [(fi, i, s, d) for fi in range(f0, len(ents)) for i, (s, d) in enumerate(ents[fi]) if (fi, i) > (f0, i0)]
            data = dict(dx)
            terms = 0
            for e in range(m, len(x) + 1):
                if e == len(x):
                    start = (f0, i0)
                else:
                    makers = [(fi, i) for fi, i, s, _ in after if _lcp(x, s) == e]
                    if not makers:
                        continue
                    start = min(makers)
                for fi, i, s, d in after:
                    if (fi, i) >= start and _lcp(x, s) >= e:
                        terms += 1
                        for tx, c in d.items():
                            data[tx] = data.get(tx, 0) + c
            STATS_F['lines'] += 1
            STATS_F['multi'] += f0 < len(ents) - 1 and terms > 0
            STATS_F['twice'] += terms > len([1 for _, _, s, _ in after if _lcp(x, s) >= m])
            if kind == "read":
                sd = "|".join("%d:%d" % (r_, p_) for r_, p_ in sorted(sdata))
            else:
                sd = "|".join("%dx%d" % (tx, sdata[tx]) for tx in sorted(sdata))
            lines.append("%s\t%s\t-|%s" % (kmer, sd, "|".join("%dx%d" % (tx, data[tx]) for tx in sorted(data))))
    return lines


def _run_files(seeds):
    bad = 0
    for seed in range(seeds):
        rng = random.Random(1000 + seed)
        alpha = rng.choice(["ACGT", "AC", "AAAC", "ACGTN", "AG"])
        max_k = rng.randint(2, 9)
        min_k = rng.randint(1, max_k)
        key_len = rng.randint(1, min(min_k, 3))
        pool = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 50))) for _ in range(rng.randint(2, 8))]
        files = []
        for _ in range(rng.randint(2, 4)):
            seqs = [rng.choice(pool) if rng.random() < 0.6 else "".join(rng.choice(alpha) for _ in range(rng.randint(0, 50)))
                    for _ in range(rng.randint(1, 5))]
            fa = "".join(">%d\n%s\n" % (rng.randint(1, 5), s) for s in seqs)
            files.append(R.build_ref_db(fa.encode(), max_k, key_len, min_k=min_k))
        fq = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in
                     enumerate((rng.choice(pool) if rng.random() < 0.5 else "".join(rng.choice(alpha) for _ in range(rng.randint(0, 40))))
                               for _ in range(rng.randint(1, 8))))
        s_min = rng.randint(min_k, max_k)
        s_max = rng.randint(s_min, max_k)
        if rng.random() < 0.5:
            s_min, s_max = min_k, max_k
        if key_len > s_min:
            continue
        reads, _ = R.build_read_db(fq.encode(), s_max, key_len, min_k=s_min)
        want = R.near_match_lines(reads, R.DB(files), 0, False)
        got = closed_form_lines_files(reads, files)
        if want != got:
            bad += 1
            print("seed", seed, "alpha", alpha, "k", min_k, max_k, "search", s_min, s_max, "key", key_len, "files", len(files), "MISMATCH", len(want), len(got))
            for a, b in zip(want, got):
                if a != b:
                    print("  want", a)
                    print("  got ", b)
                    break
    print("seeds", seeds, "bad", bad, STATS_F)
    return bad


def test_closed_form_for_several_files_equals_literal_trie():
    STATS_F.update(lines=0, multi=0, twice=0)
    assert _run_files(150) == 0
    # the cases must bite: answers that take k-mers of later files in, some of them more than once
    assert STATS_F["lines"] > 1000 and STATS_F["multi"] > 300 and STATS_F["twice"] > 50
