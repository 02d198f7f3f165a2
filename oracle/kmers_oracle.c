/* CPU restatement (plain C) of the k-mer hot path of danielmoney/kmers.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, as the checker / reported baseline -- never by the product path.
 *
 * PARITY UNPINNED: the reference has no tests, fixtures or golden vectors and Kmers.jar cannot be
 * run here (no JVM).  Pinned only by SURVEY.md Appendix B known answers and by byte-for-byte
 * agreement with the independent literal restatement oracle/slow_ref.py (tests/test_oracle_c.py).
 *
 * Follows the reference algorithm step by step (paths relative to /root/reference/src):
 *   - char-at-a-time ID/KMER/OTHER state machine, sliding maxK-byte window that is shifted per
 *     base and re-encoded per window           OtherFiles/KmersFromFile.java:84-208,
 *                                              OtherFiles/KmersFromFileStateChanger.java:46-68
 *   - base codes A0 C1 T2 G3, 2-bit MSB-first packing   Kmers/Base.java:7-41, Sequence.java:90-115
 *   - RC record unless palindrome (even k)               Kmers/Kmer.java:142-225, FileCreator.java:77-95
 *   - bucket by key(keyLength), per bucket sort + dedup + reduce, prefix-delta encode
 *                                                        KmerFiles/FileCreator.java:65-110,271-406
 *   - CountDataType / SetCompressor<ReadPos> codecs      Counts/CountDataType.java:31-46, jar
 *   - per-key two-pointer inner join                     Database/DB.java:76-90,129-137, jar StreamUtils
 *   - match line text                                    DataTypes/ResultsDataType.java:15-35
 *   - TaxaCounts default mode                            Utils/TaxaCounts.java:119-150,386-395
 *   - variable-length k-mers (minK < maxK): the `ending` branch of tryAdvance (suffixes of length maxK-1 .. minK at
 *     a sequence's end and behind an invalid base), (k-mer, length) order, KmerUtils.restrictedStream
 *                                                        KmersFromFile.java:90-133,167-176, Sequence.java:165-187,
 *                                                        KmerUtils.java:59-74
 * Restrictions (documented): k <= 32, input bytes < 128.
 * Compiled with -DKO_WIDE (a second library, _build/libkmers_oracle_wide.so): k <= 64 -- the reference's Sequence is a
 * byte array of any length (Sequence.java:15-35,90-115); here a k-mer then takes two words, `kmer` = bases 0 .. 31 and
 * `lo` = bases 32 .. k-1, both left-aligned, and every step below works on the pair.  For k > 32 that build takes
 * the same options; for k <= 32 it gives what the narrow one gives.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef KO_WIDE
#define KO_MAXK 64
#else
#define KO_MAXK 32
#endif
typedef struct {
    uint64_t kmer;  /* cBytes() zero-padded to 8 bytes, read big-endian == left-aligned 2-bit */
#ifdef KO_WIDE
    uint64_t lo;    /* ... and the next 8 bytes */
#endif
    int32_t a;      /* taxon | read index */
    int32_t b;      /* bits 15..0: (unused) | pos (Java short); bits 23..16: k-mer length */
} ko_rec;
#define REC_LEN(r) (((r).b >> 16) & 0xFF)
#define REC_POS(r) ((int16_t)((r).b & 0xFFFF))

typedef struct {
    int k, min_k, key_len, kind /*0 ref, 1 read*/, rc;   /* k = maxK */
    uint8_t *lens;        /* U: length of every k-mer (all k when min_k == k) */
    uint64_t n_windows;   /* valid forward windows emitted by the extractor */
    uint64_t n_records;   /* temp-bucket records (fwd + rc) */
    uint64_t U, V;
    uint64_t *keys;       /* U  */
    uint64_t *keys_lo;    /* U (KO_WIDE only, else NULL): bases 32 .. k-1 */
    uint64_t *val_off;    /* U+1 */
    int32_t *val_a;       /* V: taxon | read */
    int32_t *val_b;       /* V: count (unsaturated, clamped to int32) | pos */
    uint64_t *bucket_off; /* 4^key_len + 1, in units of unique k-mers */
    /* read DB only: read-name map (ReadIDMapping.java) */
    uint64_t n_reads;
    uint64_t *name_off;   /* n_reads+1 offsets into names */
    char *names;
} ko_db;

enum { ST_ID = 0, ST_KMER = 1, ST_OTHER = 2 };

static const int8_t *code_table(void) {
    static int8_t t[256];
    static int init = 0;
    if (!init) {
        memset(t, -1, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['T'] = t['t'] = 2; t['G'] = t['g'] = 3;
        init = 1;
    }
    return t;
}

typedef struct { ko_rec *p; size_t n, cap; } recvec;
static void rv_push(recvec *v, ko_rec r) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : (1u << 20);
        v->p = (ko_rec *)realloc(v->p, v->cap * sizeof(ko_rec));
        if (!v->p) { fprintf(stderr, "ko: out of memory\n"); abort(); }
    }
    v->p[v->n++] = r;
}

#ifdef KO_WIDE
#define REC_LO(r) ((r).lo)
static int rec_base(const ko_rec *r, int i) { return (int)(((i < 32 ? r->kmer : r->lo) >> (62 - 2 * (i & 31))) & 3); }
static void rec_set_base(ko_rec *r, int i, int c) {
    if (i < 32) r->kmer |= (uint64_t)c << (62 - 2 * i); else r->lo |= (uint64_t)c << (62 - 2 * (i - 32));
}
/* Kmer.inplaceRC (Kmer.java:142-208) and Kmer.isOwnRC (:210-225), base by base */
static ko_rec rec_rc(const ko_rec *x, int k) {
    ko_rec q = *x; q.kmer = 0; q.lo = 0;
    for (int i = 0; i < k; i++) rec_set_base(&q, k - 1 - i, rec_base(x, i) ^ 2);
    return q;
}
static int rec_is_palindrome(const ko_rec *x, int k) {
    if (k & 1) return 0;
    for (int i = 0; i < k / 2; i++) if (rec_base(x, i) != rec_base(x, k - 1 - i)) return 0;
    return 1;
}
#else
#define REC_LO(r) 0ull
#endif
#ifndef KO_WIDE
static uint64_t rc_packed(uint64_t x, int k) {
    /* Kmer.inplaceRC: reverse, complement = code ^ 2 */
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        uint64_t c = (x >> (62 - 2 * i)) & 3;
        r |= (c ^ 2) << (62 - 2 * (k - 1 - i));
    }
    return r;
}
static int is_palindrome(uint64_t x, int k) { /* Kmer.isOwnRC */
    if (k & 1) return 0;
    for (int i = 0; i < k / 2; i++)
        if (((x >> (62 - 2 * i)) & 3) != ((x >> (62 - 2 * (k - 1 - i))) & 3)) return 0;
    return 1;
}
#endif

static int cmp_rec(const void *pa, const void *pb) {
    const ko_rec *a = (const ko_rec *)pa, *b = (const ko_rec *)pb;
    /* Sequence.compareTo (Sequence.java:165-187): base by base, a proper prefix first == zero-padded value, then
     * length (the pad symbol A = 0 is the smallest base) */
    if (a->kmer != b->kmer) return a->kmer < b->kmer ? -1 : 1;
    if (REC_LO(*a) != REC_LO(*b)) return REC_LO(*a) < REC_LO(*b) ? -1 : 1;
    if (REC_LEN(*a) != REC_LEN(*b)) return REC_LEN(*a) < REC_LEN(*b) ? -1 : 1;
    if (a->a != b->a) return a->a < b->a ? -1 : 1;               /* TreeCountMap key order / canonical */
    if (REC_POS(*a) != REC_POS(*b)) return REC_POS(*a) < REC_POS(*b) ? -1 : 1;
    return 0;
}

void ko_free(ko_db *d) {
    if (!d) return;
    free(d->keys); free(d->keys_lo); free(d->val_off); free(d->val_a); free(d->val_b); free(d->bucket_off); free(d->lens);
    free(d->name_off); free(d->names); free(d);
}

#ifndef KO_WIDE
/* Kmers/Dust.java:12-30: keep iff sum c(c-1) over the 64 triplet counts < threshold * k * 2 */
static int dust_keep(uint64_t packed, int k, int threshold) {
    int counts[64];
    memset(counts, 0, sizeof counts);
    for (int i = 0; i < k - 2; i++) {
        int a = (packed >> (62 - 2 * i)) & 3, b = (packed >> (62 - 2 * (i + 1))) & 3, c = (packed >> (62 - 2 * (i + 2))) & 3;
        counts[a * 16 + b * 4 + c]++;
    }
    int score = 0;
    for (int i = 0; i < 64; i++) score += counts[i] * (counts[i] - 1);
    return score < threshold * k * 2;
}
/* Kmers/RunOfSame.java:12-36: drop iff some run of equal bases reaches `length` (checked on increments only) */
static int runofsame_keep(uint64_t packed, int k, int length) {
    int c = (packed >> 62) & 3;
    int l = 1;
    for (int i = 1; i < k; i++) {
        int b = (packed >> (62 - 2 * i)) & 3;
        if (b == c) { l++; if (l == length) return 0; }
        else { c = b; l = 1; }
    }
    return 1;
}
#endif

#ifdef KO_WIDE
/* the same two predicates on a k-mer of up to 64 bases, base by base */
static int rec_dust_keep(const ko_rec *r, int k, int threshold) {
    int counts[64];
    memset(counts, 0, sizeof counts);
    for (int i = 0; i < k - 2; i++) counts[rec_base(r, i) * 16 + rec_base(r, i + 1) * 4 + rec_base(r, i + 2)]++;
    int score = 0;
    for (int i = 0; i < 64; i++) score += counts[i] * (counts[i] - 1);
    return score < threshold * k * 2;
}
static int rec_runofsame_keep(const ko_rec *r, int k, int length) {
    int c = rec_base(r, 0), l = 1;
    for (int i = 1; i < k; i++) {
        int b = rec_base(r, i);
        if (b == c) { l++; if (l == length) return 0; }
        else { c = b; l = 1; }
    }
    return 1;
}
#endif

static int g_dust = -1, g_runofsame = -1;
/* MakeDatabase -D / -R (MakeDatabase.java:279-291): filters applied to the forward k-mer stream of the
 * NEXT ko_build call(s); -1 = off.  (Test infrastructure: a process-wide setting is good enough.) */
void ko_set_filters(int dust_threshold, int runofsame_length) { g_dust = dust_threshold; g_runofsame = runofsame_length; }

/* Key slice: keep only the records whose key bucket (Kmer.key(keyLength), Kmer.java:47-55) lies in [lo, hi) -- the
 * part of the database MakeAndWriteKey writes for those keys (FileCreator.java:271-406 works key by key; every
 * record is bucketed by its own k-mer, FileCreator.java:82-87, so a reverse complement can be kept where its forward
 * k-mer is not).  Lets a test check a slice of a database too large to restate whole.  hi == 0: everything.
 * n_windows then counts the forward windows kept.  Applies to the NEXT ko_build call(s). */
static uint32_t g_range_lo = 0, g_range_hi = 0;
void ko_set_key_range(uint32_t lo, uint32_t hi) { g_range_lo = lo; g_range_hi = hi; }
static int in_range(uint64_t kmer, int key_len) {
    if (!g_range_hi) return 1;
    uint32_t b = (uint32_t)(kmer >> (64 - 2 * key_len));
    return b >= g_range_lo && b < g_range_hi;
}

/* MakeDatabase -k: the shortest k-mer (minK) of the NEXT ko_build call(s); 0 = the same as k (maxK). */
static int g_min_k = 0;
void ko_set_min_k(int min_k) { g_min_k = min_k; }

/* mode 0 = FASTA -> reference DB (MakeDatabase -a), 1 = FASTQ -> read DB (MakeDatabase -q).  k = maxK. */
ko_db *ko_build(const uint8_t *file, uint64_t n, int mode, int k, int key_len, int threads) {
    const int min_k = g_min_k > 0 ? g_min_k : k;
    if (k < 1 || k > KO_MAXK || min_k < 1 || min_k > k || key_len < 1 || key_len > min_k || key_len > 12) return NULL;
    const int8_t *code = code_table();
    ko_db *d = (ko_db *)calloc(1, sizeof(ko_db));
    d->k = k; d->min_k = min_k; d->key_len = key_len; d->kind = mode; d->rc = (mode == 0);
    uint32_t nkeys = 1u << (2 * key_len);

    /* ---- transition tables: KmersFromFileStateChanger.java:46-68 */
    int8_t trans[3][256];
    memset(trans, -1, sizeof trans);
    if (mode == 0) {
        trans[ST_ID][' '] = ST_OTHER; trans[ST_ID]['\n'] = ST_KMER; trans[ST_OTHER]['\n'] = ST_KMER;
        trans[ST_KMER]['\n'] = ST_OTHER; trans[ST_OTHER]['>'] = ST_ID;
    } else {
        trans[ST_OTHER]['@'] = ST_ID; trans[ST_KMER]['@'] = ST_ID; trans[ST_ID]['\n'] = ST_KMER;
        trans[ST_KMER]['+'] = ST_OTHER;
    }

    /* ---- HOT LOOP 1: tryAdvance + FileCreator.addKmers, single thread like the reference.
     * The outer loop is one tryAdvance call, the inner one its do { } while (kwd == null) (KmersFromFile.java:84-208). */
    recvec recs = {0, 0, 0};
    uint8_t window[KO_MAXK];
    memset(window, 0, sizeof window);
    int state = ST_OTHER;
    long pos = 0;
    uint64_t id_start = 0, id_len = 0;           /* current header id = file[id_start .. +id_len) */
    uint64_t old_start = 0, old_len = 0;         /* oldid */
    uint64_t tax_of = (uint64_t)-1; int32_t taxon = 0;   /* Integer.parseInt(id), cached per header */
    /* ReadIDMapping state: curReadS = "" and curReadI = -1 to begin with (ReadIDMapping.java:10-12) */
    uint64_t cur_name_start = 0, cur_name_len = 0; int32_t cur_read = -1;
    size_t names_cap = 1 << 16, names_len = 0, noff_cap = 1 << 12;
    if (mode == 1) {
        d->names = (char *)malloc(names_cap);
        d->name_off = (uint64_t *)malloc(noff_cap * sizeof(uint64_t));
        d->name_off[0] = 0;
    }
    int have_kwd = 0; ko_rec kwd; memset(&kwd, 0, sizeof kwd);    /* the `kwd` FIELD: it survives from one call to the next (quirk C.7) */
    int ending = 0; long cur_k = 0;
    uint64_t ip = 0;
    for (;;) {                                    /* tryAdvance */
        for (;;) {                                /* do { ... } while (kwd == null) */
            int at_eof = 0, try_len = 0; uint64_t ms = 0, ml = 0; long mpos = 0;
            if (ending) {                         /* :90-111 */
                if (cur_k < min_k) { ending = 0; have_kwd = 0; }
                else { try_len = (int)cur_k; ms = old_start; ml = old_len; mpos = pos - cur_k; cur_k--; }
            } else if (ip >= n) {                 /* in.read() == -1, :116-123 */
                ending = 1; old_start = id_start; old_len = id_len;
                cur_k = (pos - 1 < k - 1) ? pos - 1 : k - 1;
                at_eof = 1;
            } else {
                uint8_t c = file[ip++];
                int ns = trans[state][c];
                if (ns >= 0) {                    /* :126-148 */
                    if (state == ST_KMER && pos >= min_k) {
                        old_start = id_start; old_len = id_len; ending = 1;
                        cur_k = pos < k - 1 ? pos : k - 1;
                    }
                    state = ns;
                    if (state == ST_KMER) pos = 0;                  /* kwd NOT cleared here (:137-139) */
                    else if (state == ST_ID) { id_start = ip; id_len = 0; have_kwd = 0; }
                    else have_kwd = 0;
                } else if (state == ST_ID) id_len++;
                else if (state == ST_KMER) {
                    if (c > 32) {
                        pos++;
                        memmove(window, window + 1, (size_t)k - 1);   /* System.arraycopy shift, :157 */
                        window[k - 1] = c;
                        if (pos >= k) { try_len = k; ms = id_start; ml = id_len; mpos = pos - k; }
                    } else have_kwd = 0;
                }
            }
            if (try_len) {
                /* new Kmer(newbytes) on the last try_len bytes of the window: per-window re-encode,
                 * InvalidBaseException => no k-mer; the mapper runs only after the k-mer was built */
                uint64_t packed = 0, packed_lo = 0; int ok = 1;
                for (int i = 0; i < try_len; i++) {
                    uint8_t w = window[k - try_len + i];
                    int v = w < 128 ? code[w] : -1;
                    if (v < 0) { ok = 0; break; }
                    if (i < 32) packed |= (uint64_t)v << (62 - 2 * i);
                    else packed_lo |= (uint64_t)v << (62 - 2 * (i - 32));
                }
                (void)packed_lo;
                if (ok) {
                    kwd.kmer = packed;
#ifdef KO_WIDE
                    kwd.lo = packed_lo;
#endif
                    if (mode == 0) {
                        if (tax_of != ms) {       /* Integer.parseInt(id) */
                            char buf[32]; size_t l = ml < 31 ? ml : 31;
                            memcpy(buf, file + ms, l); buf[l] = 0;
                            taxon = (int32_t)strtol(buf, NULL, 10); tax_of = ms;
                        }
                        kwd.a = taxon; kwd.b = try_len << 16;
                    } else {                      /* ReadIDMapping.geNext: compares with the last *emitting* id */
                        if (cur_name_len != ml || memcmp(file + cur_name_start, file + ms, ml) != 0) {
                            cur_name_start = ms; cur_name_len = ml; cur_read++;
                            while (names_len + ml + 1 > names_cap) { names_cap *= 2; d->names = (char *)realloc(d->names, names_cap); }
                            memcpy(d->names + names_len, file + ms, ml); names_len += ml;
                            if ((uint64_t)cur_read + 2 > noff_cap) { noff_cap *= 2; d->name_off = (uint64_t *)realloc(d->name_off, noff_cap * sizeof(uint64_t)); }
                            d->name_off[cur_read + 1] = names_len;
                        }
                        kwd.a = cur_read;
                        kwd.b = (try_len << 16) | (int32_t)((uint16_t)(int16_t)mpos);   /* (short)(pos - length) */
                    }
                    have_kwd = 1;
                } else if (try_len == k) {        /* :167-176 (in the ending branch the exception changes nothing) */
                    have_kwd = 0;
                    if (pos > min_k) {
                        ending = 1; old_start = id_start; old_len = id_len;
                        cur_k = (pos - 1 < k - 1) ? pos - 1 : k - 1;
                    }
                }
            }
            if (at_eof || have_kwd) break;
        }
        if (!have_kwd) break;                     /* return false */
        /* consumer.accept(kwd): kstream.filter(Dust).filter(RunOfSame), then FileCreator.addKmers (:77-95) */
        const int len = REC_LEN(kwd);
#ifdef KO_WIDE
        if (g_dust >= 0 && !rec_dust_keep(&kwd, len, g_dust)) continue;
        if (g_runofsame >= 0 && !rec_runofsame_keep(&kwd, len, g_runofsame)) continue;
#else
        if (g_dust >= 0 && !dust_keep(kwd.kmer, len, g_dust)) continue;
        if (g_runofsame >= 0 && !runofsame_keep(kwd.kmer, len, g_runofsame)) continue;
#endif
        if (in_range(kwd.kmer, key_len)) { rv_push(&recs, kwd); d->n_windows++; }
#ifdef KO_WIDE
        if (d->rc && !rec_is_palindrome(&kwd, len)) {
            ko_rec q = rec_rc(&kwd, len);
            if (in_range(q.kmer, key_len)) rv_push(&recs, q);
        }
#else
        if (d->rc && !is_palindrome(kwd.kmer, len)) {
            ko_rec q = kwd; q.kmer = rc_packed(kwd.kmer, len);
            if (in_range(q.kmer, key_len)) rv_push(&recs, q);
        }
#endif
    }
    d->n_reads = (uint64_t)(cur_read + 1);
    d->n_records = recs.n;

    /* ---- temp-bucket scatter by key(keyLength): FileCreator.java:82, jar IntegerIndexedOutputFileCache */
    uint64_t *boff = (uint64_t *)calloc((size_t)nkeys + 1, sizeof(uint64_t));
    int sh = 64 - 2 * key_len;
    for (size_t i = 0; i < recs.n; i++) boff[(recs.p[i].kmer >> sh) + 1]++;
    for (uint32_t i = 0; i < nkeys; i++) boff[i + 1] += boff[i];
    ko_rec *sorted = (ko_rec *)malloc((recs.n ? recs.n : 1) * sizeof(ko_rec));
    {
        uint64_t *cur = (uint64_t *)malloc((size_t)nkeys * sizeof(uint64_t));
        memcpy(cur, boff, (size_t)nkeys * sizeof(uint64_t));
        for (size_t i = 0; i < recs.n; i++) sorted[cur[recs.p[i].kmer >> sh]++] = recs.p[i];
        free(cur);
    }
    free(recs.p);

    /* ---- HOT LOOP 2: MakeAndWriteKey per key on a pool (LimitedQueueExecutor) */
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    uint64_t *ucount = (uint64_t *)calloc((size_t)nkeys + 1, sizeof(uint64_t));
    uint64_t *vcount = (uint64_t *)calloc((size_t)nkeys + 1, sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t key = 0; key < (int64_t)nkeys; key++) {
        ko_rec *p = sorted + boff[key]; size_t m = boff[key + 1] - boff[key];
        qsort(p, m, sizeof(ko_rec), cmp_rec);     /* TreeMap<Kmer,..> + TreeCountMap ordering */
        uint64_t u = 0, v = 0;
        for (size_t i = 0; i < m; i++) {
            int newk = (i == 0) || p[i].kmer != p[i - 1].kmer || REC_LO(p[i]) != REC_LO(p[i - 1]) || REC_LEN(p[i]) != REC_LEN(p[i - 1]);
            if (newk) u++;
            if (d->kind == 1 || newk || p[i].a != p[i - 1].a) v++;
        }
        ucount[key + 1] = u; vcount[key + 1] = v;
    }
    for (uint32_t i = 0; i < nkeys; i++) { ucount[i + 1] += ucount[i]; vcount[i + 1] += vcount[i]; }
    d->U = ucount[nkeys]; d->V = vcount[nkeys];
    d->keys = (uint64_t *)malloc((d->U + 1) * sizeof(uint64_t));
    d->val_off = (uint64_t *)malloc((d->U + 1) * sizeof(uint64_t));
    d->val_a = (int32_t *)malloc((d->V + 1) * sizeof(int32_t));
    d->val_b = (int32_t *)malloc((d->V + 1) * sizeof(int32_t));
    d->lens = (uint8_t *)malloc(d->U + 1);
#ifdef KO_WIDE
    d->keys_lo = (uint64_t *)malloc((d->U + 1) * sizeof(uint64_t));
#endif
    d->bucket_off = ucount;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t key = 0; key < (int64_t)nkeys; key++) {
        ko_rec *p = sorted + boff[key]; size_t m = boff[key + 1] - boff[key];
        uint64_t u = ucount[key], v = vcount[key];
        for (size_t i = 0; i < m; i++) {
            int newk = (i == 0) || p[i].kmer != p[i - 1].kmer || REC_LO(p[i]) != REC_LO(p[i - 1]) || REC_LEN(p[i]) != REC_LEN(p[i - 1]);
            if (newk) {
                d->keys[u] = p[i].kmer; d->lens[u] = (uint8_t)REC_LEN(p[i]); d->val_off[u] = v;
#ifdef KO_WIDE
                d->keys_lo[u] = p[i].lo;
#endif
                u++;
            }
            if (d->kind == 1) { d->val_a[v] = p[i].a; d->val_b[v] = REC_POS(p[i]); v++; }
            else if (newk || p[i].a != p[i - 1].a) { d->val_a[v] = p[i].a; d->val_b[v] = 1; v++; }
            else if (d->val_b[v - 1] < INT32_MAX) d->val_b[v - 1]++;   /* TreeCountMap.add */
        }
    }
    d->val_off[d->U] = d->V;
    free(vcount); free(boff); free(sorted);
    return d;
}

/* accessors for ctypes */
uint64_t ko_U(const ko_db *d) { return d->U; }
uint64_t ko_V(const ko_db *d) { return d->V; }
uint64_t ko_n_windows(const ko_db *d) { return d->n_windows; }
uint64_t ko_n_records(const ko_db *d) { return d->n_records; }
uint64_t ko_n_reads(const ko_db *d) { return d->n_reads; }
int ko_min_k(const ko_db *d) { return d->min_k; }
const uint8_t *ko_lens(const ko_db *d) { return d->lens; }
const uint64_t *ko_keys(const ko_db *d) { return d->keys; }
const uint64_t *ko_keys_lo(const ko_db *d) { return d->keys_lo; }
int ko_max_k(void) { return KO_MAXK; }
const uint64_t *ko_val_off(const ko_db *d) { return d->val_off; }
const int32_t *ko_val_a(const ko_db *d) { return d->val_a; }
const int32_t *ko_val_b(const ko_db *d) { return d->val_b; }
const uint64_t *ko_bucket_off(const ko_db *d) { return d->bucket_off; }
const uint64_t *ko_name_off(const ko_db *d) { return d->name_off; }
const char *ko_names(const ko_db *d) { return d->names; }

/* ---- block encoding: MakeAndWriteKey (FileCreator.java:312-347) + meta (:153-164).
 * Writes all 4^l blocks back to back into out (cap bytes), block_off[4^l+1]; returns bytes needed. */
static void be32(uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static void be16(uint8_t *p, uint16_t v) { p[0] = v >> 8; p[1] = v; }

uint64_t ko_encode_blocks(const ko_db *d, uint8_t *out, uint64_t cap, uint64_t *block_off) {
    uint32_t nkeys = 1u << (2 * d->key_len);
    uint64_t w = 0;
    uint8_t kc[17], prev[17];
    for (uint32_t key = 0; key < nkeys; key++) {
        if (block_off) block_off[key] = w;
        memset(prev, 0, sizeof prev);             /* byte[] prev = new byte[1] */
        int prev_n = 1;
        for (uint64_t u = d->bucket_off[key]; u < d->bucket_off[key + 1]; u++) {
            const int l = (d->lens[u] - 1) / 4 + 1;   /* packed bytes of this k-mer */
            kc[0] = d->lens[u];                       /* Kmer.compressedBytes: [length byte][packed] */
            for (int i = 0; i < 8; i++) kc[1 + i] = (uint8_t)(d->keys[u] >> (56 - 8 * i));
            for (int i = 0; i < 8; i++) kc[9 + i] = d->keys_lo ? (uint8_t)(d->keys_lo[u] >> (56 - 8 * i)) : 0;
            /* shared(): leading bytes in common with the previous record's compressedBytes, within both lengths
             * (FileCreator.java:244-252); prev = new byte[1] at the start of a key, and a record that shares
             * nothing is written whole (:318-330) */
            int s = 0;
            while (s < prev_n && s < l + 1 && prev[s] == kc[s]) s++;
            memcpy(prev, kc, 17); prev_n = l + 1;
            uint64_t nv = d->val_off[u + 1] - d->val_off[u];
            uint64_t need = 1 + (uint64_t)(l + 1 - s) + (d->kind == 0 ? 2 + 5 * nv : 4 + 6 * nv);
            if (out && w + need <= cap) {
                uint8_t *p = out + w;
                *p++ = (uint8_t)s;
                memcpy(p, kc + s, (size_t)(l + 1 - s)); p += l + 1 - s;
                if (d->kind == 0) {               /* CountDataType.compress */
                    be16(p, (uint16_t)nv); p += 2;
                    for (uint64_t v = d->val_off[u]; v < d->val_off[u + 1]; v++) {
                        be32(p, (uint32_t)d->val_a[v]); p += 4;
                        *p++ = (uint8_t)(d->val_b[v] > 255 ? 255 : d->val_b[v]);
                    }
                } else {                          /* SetCompressor<ReadPos>, canonical ascending */
                    be32(p, (uint32_t)nv); p += 4;
                    for (uint64_t v = d->val_off[u]; v < d->val_off[u + 1]; v++) {
                        be32(p, (uint32_t)d->val_a[v]); p += 4;
                        be16(p, (uint16_t)d->val_b[v]); p += 2;
                    }
                }
            }
            w += need;
        }
    }
    if (block_off) block_off[nkeys] = w;
    return w;
}

/* ---- KmerUtils.restrictedStream(stream, L, L, dataType) (KmerUtils.java:59-74) over a whole database: drop the
 * k-mers shorter than L, cut the rest to L (Kmer.limitTo, Kmer.java:277-289), fold runs of equal neighbours with the
 * data type's reducer -- counts add per taxon (CountDataType.getMerger: TreeCountMap.addAll), read-position sets
 * unite (every ReadPos is its own object: nothing is ever dropped; canonical order ascending).  What
 * quickMatchCommonKey does to the database side (DB.java:134) and KmerFile.allRestrictedKmers to the search side
 * (KmerFile.java:80-83, Matcher.java:241) under Matcher -k L -K L. */
typedef struct { int32_t a, b; } ko_val;
static int cmp_val(const void *pa, const void *pb) {
    const ko_val *x = (const ko_val *)pa, *y = (const ko_val *)pb;
    if (x->a != y->a) return x->a < y->a ? -1 : 1;
    if (x->b != y->b) return x->b < y->b ? -1 : 1;
    return 0;
}
ko_db *ko_restrict(const ko_db *s, int L) {
    if (L < s->min_k || L > s->k) return NULL;     /* DB.java:78-81 */
    ko_db *d = (ko_db *)calloc(1, sizeof(ko_db));
    d->k = L; d->min_k = L; d->key_len = s->key_len; d->kind = s->kind; d->rc = s->rc;
    d->n_windows = s->n_windows; d->n_records = s->n_records;
    const uint32_t nkeys = 1u << (2 * s->key_len);
    const uint64_t mask = L >= 32 ? ~0ull : ~(~0ull >> (2 * L));
    /* (two-word k-mers: the second word keeps bases 32 .. L-1) */
    const uint64_t mask_lo = L <= 32 ? 0ull : (L >= 64 ? ~0ull : ~(~0ull >> (2 * (L - 32))));
    if (s->keys_lo && L > 32) d->keys_lo = (uint64_t *)malloc((s->U + 1) * sizeof(uint64_t));
    d->keys = (uint64_t *)malloc((s->U + 1) * sizeof(uint64_t));
    d->val_off = (uint64_t *)malloc((s->U + 1) * sizeof(uint64_t));
    d->val_a = (int32_t *)malloc((s->V + 1) * sizeof(int32_t));
    d->val_b = (int32_t *)malloc((s->V + 1) * sizeof(int32_t));
    d->lens = (uint8_t *)malloc(s->U + 1);
    d->bucket_off = (uint64_t *)calloc((size_t)nkeys + 1, sizeof(uint64_t));
    ko_val *tmp = (ko_val *)malloc((s->V + 1) * sizeof(ko_val));
    uint64_t U = 0, V = 0;
    for (uint64_t u = 0; u < s->U;) {
        if (s->lens[u] < L) { u++; continue; }
        const uint64_t key = s->keys[u] & mask, key_lo = s->keys_lo ? s->keys_lo[u] & mask_lo : 0;
        uint64_t nt = 0;
        /* the run of neighbours with the same cut k-mer (shorter ones in between are filtered out first) */
        while (u < s->U && (s->lens[u] < L || ((s->keys[u] & mask) == key && (s->keys_lo ? s->keys_lo[u] & mask_lo : 0) == key_lo))) {
            if (s->lens[u] >= L)
                for (uint64_t v = s->val_off[u]; v < s->val_off[u + 1]; v++) { tmp[nt].a = s->val_a[v]; tmp[nt].b = s->val_b[v]; nt++; }
            u++;
        }
        qsort(tmp, nt, sizeof(ko_val), cmp_val);
        d->keys[U] = key; d->lens[U] = (uint8_t)L; d->val_off[U] = V;
        if (d->keys_lo) d->keys_lo[U] = key_lo;
        d->bucket_off[(key >> (64 - 2 * s->key_len)) + 1]++;
        U++;
        for (uint64_t i = 0; i < nt; i++) {
            if (s->kind == 0 && i > 0 && tmp[i].a == tmp[i - 1].a) {         /* TreeCountMap.addAll */
                int64_t c = (int64_t)d->val_b[V - 1] + tmp[i].b;
                d->val_b[V - 1] = c > INT32_MAX ? INT32_MAX : (int32_t)c;
            } else { d->val_a[V] = tmp[i].a; d->val_b[V] = tmp[i].b; V++; }
        }
    }
    free(tmp);
    for (uint32_t i = 0; i < nkeys; i++) d->bucket_off[i + 1] += d->bucket_off[i];
    d->U = U; d->V = V; d->val_off[U] = V;
    return d;
}

/* ---- HOT LOOP 3: per-key two-pointer inner join (DB.quickMatchCommonKey + matchTwoStreams).
 * hit[i] = index of the reference k-mer equal to search k-mer i, or -1.  Returns #hits. */
uint64_t ko_match(const ko_db *s, const ko_db *r, int64_t *hit, int threads) {
    /* the quick branch: search k-mers of one length (DB.java:83), the database restricted to it (:134) */
    if (s->k != r->k || s->key_len != r->key_len || s->min_k != s->k || r->min_k != r->k) return (uint64_t)-1;
    uint32_t nkeys = 1u << (2 * s->key_len);
    uint64_t hits = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : hits)
    for (int64_t key = 0; key < (int64_t)nkeys; key++) {
        uint64_t a = s->bucket_off[key], ae = s->bucket_off[key + 1];
        uint64_t b = r->bucket_off[key], be = r->bucket_off[key + 1];
        for (uint64_t i = a; i < ae; i++) hit[i] = -1;
        while (a < ae && b < be) {
            const uint64_t sl = s->keys_lo ? s->keys_lo[a] : 0, rl = r->keys_lo ? r->keys_lo[b] : 0;
            if (s->keys[a] == r->keys[b] && sl == rl) { hit[a] = (int64_t)b; hits++; a++; b++; }
            else if (s->keys[a] < r->keys[b] || (s->keys[a] == r->keys[b] && sl < rl)) a++;
            else b++;
        }
    }
    return hits;
}

/* Match line text (ResultsDataType.toString) for every hit, ascending; returns bytes needed. */
uint64_t ko_match_lines(const ko_db *s, const ko_db *r, const int64_t *hit, char *out, uint64_t cap) {
    static const char B[4] = {'A', 'C', 'T', 'G'};
    uint64_t w = 0;
    char tmp[80];
    for (uint64_t i = 0; i < s->U; i++) {
        if (hit[i] < 0) continue;
#define PUT(str, len) do { if (out && w + (len) <= cap) memcpy(out + w, (str), (len)); w += (len); } while (0)
        for (int j = 0; j < s->k; j++) tmp[j] = B[((j < 32 ? s->keys[i] : s->keys_lo[i]) >> (62 - 2 * (j & 31))) & 3];
        PUT(tmp, (uint64_t)s->k); PUT("\t", 1);
        for (uint64_t v = s->val_off[i]; v < s->val_off[i + 1]; v++) {
            int n = s->kind == 1 ? snprintf(tmp, sizeof tmp, "%s%d:%d", v > s->val_off[i] ? "|" : "", s->val_a[v], s->val_b[v])
                                 : snprintf(tmp, sizeof tmp, "%s%dx%d", v > s->val_off[i] ? "|" : "", s->val_a[v], s->val_b[v] > 255 ? 255 : s->val_b[v]);
            PUT(tmp, (uint64_t)n);
        }
        PUT("\t-", 2);
        uint64_t h = (uint64_t)hit[i];
        for (uint64_t v = r->val_off[h]; v < r->val_off[h + 1]; v++) {
            /* counts come back from the DB file, i.e. saturated at 255 (CountDataType.java:40) */
            int n = snprintf(tmp, sizeof tmp, "|%dx%d", r->val_a[v], r->val_b[v] > 255 ? 255 : r->val_b[v]);
            PUT(tmp, (uint64_t)n);
        }
        PUT("\n", 1);
#undef PUT
    }
    return w;
}

/* ---- HOT LOOP 4: TaxaCounts default mode.  taxon[] ascending; returns number of taxa (<= cap filled). */
static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : x > y;
}
uint64_t ko_taxacounts(const ko_db *r, int32_t *taxon, int64_t *match, int64_t *unique, uint64_t cap) {
    int32_t *t = (int32_t *)malloc((r->V + 1) * sizeof(int32_t));
    memcpy(t, r->val_a, r->V * sizeof(int32_t));
    qsort(t, r->V, sizeof(int32_t), cmp_i32);
    uint64_t nt = 0;
    for (uint64_t i = 0; i < r->V; i++) if (i == 0 || t[i] != t[i - 1]) t[nt++] = t[i];
    int64_t *m = (int64_t *)calloc(nt + 1, sizeof(int64_t)), *q = (int64_t *)calloc(nt + 1, sizeof(int64_t));
    for (uint64_t u = 0; u < r->U; u++) {
        uint64_t a = r->val_off[u], b = r->val_off[u + 1];
        for (uint64_t v = a; v < b; v++) {
            int32_t *f = (int32_t *)bsearch(&r->val_a[v], t, nt, sizeof(int32_t), cmp_i32);
            m[f - t]++;
            if (b - a == 1) q[f - t]++;
        }
    }
    for (uint64_t i = 0; i < nt && i < cap; i++) { taxon[i] = t[i]; match[i] = m[i]; unique[i] = q[i]; }
    free(t); free(m); free(q);
    return nt;
}
