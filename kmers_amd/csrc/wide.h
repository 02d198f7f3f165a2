// wide.h -- k-mers of 33 .. 64 bases: two 64-bit words per k-mer.
//
// The reference's k-mer is a byte array of any length (Kmers/Sequence.java:15-35,90-115; Kmer.java:47-55,90-97,142-225);
// its order is base by base (Sequence.compareTo, Sequence.java:165-187).  Here a k-mer of k > 32 bases is (hi, lo): hi =
// bases 0 .. 31, lo = bases 32 .. k-1 left-aligned and zero-padded, 2 bits per base (A0 C1 T2 G3) -- (hi, lo) ascending
// as a pair of unsigned words is the reference's order, and hi then lo, big-endian, cut to ceil(k / 4) bytes, is
// Sequence.cBytes.  Kmer.key(keyLength <= 12) lies in hi's top bits, so buckets work as for one-word k-mers.
//
// The build reuses the one-word machinery instead of doubling it (BASELINE's configs all have k = 31: this path is
// about coverage of the reference's interface, every kernel here is HBM-bound byte work like its one-word twin):
//   1. wide_extract_kernel: windows -> (hi, lo, payload) records, count pass + write pass, window order;
//   2. two stable index sorts (radix.h, hybrid_sort_records<u32>): by lo, then by hi -> the records' order by k-mer;
//   3. heads + scan over that order -> every record's RANK among the distinct k-mers, and the distinct k-mers;
//   4. the ranks, left-aligned, are one-word keys with the k-mers' order and equality: the ordinary sort + dedup
//      (sort_reduce_class) builds the CSR from (rank, payload) records -- taxa order, read sets and all;
//   5. the CSR's keys (0 .. U-1) are swapped for the distinct k-mers' (hi, lo).
// The exact join looks every search k-mer up in the piece of the reference its tile can meet, staged in LDS.
#pragma once
#include "extract.h"

#define EXW_EXT 64

struct WideArgs {
    ExtractArgs e;        // k = 33 .. 64; out_keys = the hi words
    u64 *out_lo;
};

// ---- MakeDatabase -D / -R on a k-mer of L bases in two words (extract.h's dust_keep / runofsame_keep on 128 bits) -------------
__device__ __forceinline__ void wide_shl(u64 hi, u64 lo, int s, u64 &oh, u64 &ol) {      // (hi:lo) << s, 0 <= s < 128
    if (s == 0) { oh = hi; ol = lo; }
    else if (s < 64) { oh = (hi << s) | (lo >> (64 - s)); ol = lo << s; }
    else if (s == 64) { oh = lo; ol = 0; }
    else { oh = lo << (s - 64); ol = 0; }
}
// the first `groups` two-bit groups of a 128-bit string
__device__ __forceinline__ void wide_top_groups(int groups, u64 &mh, u64 &ml) {
    mh = groups >= 32 ? ~0ull : (groups <= 0 ? 0ull : ~(~0ull >> (2 * groups)));
    ml = groups <= 32 ? 0ull : (groups >= 64 ? ~0ull : ~(~0ull >> (2 * (groups - 32))));
}
// Kmers/RunOfSame.java:12-36: drop the k-mer iff some run of equal bases reaches `length`
__device__ __forceinline__ bool wide_runofsame_keep(u64 hi, u64 lo, int L, int length) {
    const int R = length - 1;                          // consecutive equal neighbour pairs needed
    if (R < 1 || R > L - 1) return true;
    const u64 lowbits = 0x5555555555555555ull;
    u64 sh, sl, mh, ml;
    wide_shl(hi, lo, 2, sh, sl);
    const u64 xh = hi ^ sh, xl = lo ^ sl;              // group i: base_i ^ base_{i+1}
    u64 eh = ~(xh | (xh >> 1)) & lowbits, el = ~(xl | (xl >> 1)) & lowbits;
    wide_top_groups(L - 1, mh, ml);                    // pairs 0 .. L-2 only
    eh &= mh; el &= ml;
    u64 th = eh, tl = el;
    int s = 1;
    while (s * 2 <= R) { wide_shl(th, tl, 2 * s, sh, sl); th &= sh; tl &= sl; s *= 2; }      // t[i] = e[i .. i+s-1] all set
    if (R > s) { wide_shl(th, tl, 2 * (R - s), sh, sl); th &= sh; tl &= sl; }
    return (th | tl) == 0;
}
// Kmers/Dust.java:12-30: score = sum c(c-1) over the triplet counts = twice the pairs of equal triplets; keep iff
// pairs < threshold * L
__device__ __forceinline__ bool wide_dust_keep(u64 hi, u64 lo, int L, int threshold) {
    const int nt = L - 2;
    int pairs = 0;
    const u64 lowbits = 0x5555555555555555ull;
    for (int d = 1; d < nt; d++) {
        u64 sh, sl, mh, ml;
        wide_shl(hi, lo, 2 * d, sh, sl);
        const u64 yh = hi ^ sh, yl = lo ^ sl;          // group i: base_i ^ base_{i+d}
        const u64 eh = ~(yh | (yh >> 1)) & lowbits, el = ~(yl | (yl >> 1)) & lowbits;
        u64 ah, al, bh, bl;
        wide_shl(eh, el, 2, ah, al);
        wide_shl(eh, el, 4, bh, bl);
        wide_top_groups(nt - d, mh, ml);               // i = 0 .. nt-1-d
        pairs += __popcll(eh & ah & bh & mh) + __popcll(el & al & bl & ml);
    }
    return pairs < threshold * L;
}
__device__ __forceinline__ bool wide_filters_keep(const ExtractArgs &a, u64 hi, u64 lo, int L) {
    if (a.runofsame >= 0 && !wide_runofsame_keep(hi, lo, L, a.runofsame)) return false;
    if (a.dust >= 0 && !wide_dust_keep(hi, lo, L, a.dust)) return false;
    return true;
}

struct WideSite {
    u64 hi, lo, rhi, rlo;
    bool valid, ef, er;
    u32 seq;
    u32 pos;
};

// the window of W bases at position p of the tile (pk / inv / stb / seqrel: ex_load_tile_ext<EXW_EXT>)
__device__ __forceinline__ WideSite wide_site(const ExtractArgs &a, const u32 *pk, const u32 *inv, const u32 *stb,
                                              const u32 *seqrel, u32 s_lo, u64 t0, u32 p) {
    WideSite x;
    x.valid = x.ef = x.er = false;
    x.hi = x.lo = x.rhi = x.rlo = 0; x.seq = 0; x.pos = 0;
    const int W = a.k;
    const u64 gp = t0 + p;
    if (gp + (u64)W > a.nbases) return x;
    const u32 w2 = p >> 5, r2 = p & 31;
    const u64 b0 = ((u64)inv[w2] << 32) | inv[w2 + 1], b1 = (u64)inv[w2 + 2] << 32;
    const u64 s0 = ((u64)stb[w2] << 32) | stb[w2 + 1], s1 = (u64)stb[w2 + 2] << 32;
    const u64 bad = r2 ? (b0 << r2) | (b1 >> (64 - r2)) : b0;     // bit 63 - i: base p + i is invalid
    const u64 st = r2 ? (s0 << r2) | (s1 >> (64 - r2)) : s0;      // bit 63 - i: a sequence starts at p + i
    if ((bad >> (64 - W)) != 0) return x;                         // Sequence.java:15-35 would throw: no k-mer
    if (((st << 1) >> (65 - W)) != 0) return x;                   // the window crosses into the next sequence
    x.valid = true;
    const u32 w = p >> 4, sh = 2 * (p & 15);
    const u64 a0 = ((u64)pk[w] << 32) | pk[w + 1], a1 = ((u64)pk[w + 2] << 32) | pk[w + 3], a2 = (u64)pk[w + 4] << 32;
    const int nlo = W - 32;
    const u64 mask_lo = nlo == 32 ? ~0ull : ~(~0ull >> (2 * nlo));
    x.hi = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
    x.lo = (sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1) & mask_lo;
    x.seq = s_lo + seqrel[p];
    x.pos = (u32)(gp - a.seq_off[x.seq]);
    // (the filters judge the forward k-mer, its reverse complement goes with it; reads are numbered before them)
    if ((a.dust >= 0 || a.runofsame >= 0) && !wide_filters_keep(a, x.hi, x.lo, W)) return x;
    x.ef = true;
    if (a.ranged) {
        const u32 fb = (u32)(x.hi >> a.range_shift);
        x.ef = fb >= a.range_lo && fb < a.range_hi;
    }
    if (a.rc) {
        // the W bases in reverse order, left-aligned (Kmer.inplaceRC, Kmer.java:142-208: reversed, code ^ 2)
        const u64 rh = revpairs64(x.lo), rl = revpairs64(x.hi);
        const int s = 2 * (64 - W);
        const u64 vh = s ? (rh << s) | (rl >> (64 - s)) : rh, vl = rl << s;
        x.er = !(((W & 1) == 0) && vh == x.hi && vl == x.lo);     // Kmer.isOwnRC (Kmer.java:210-225): a palindrome of even length
        x.rhi = vh ^ 0xAAAAAAAAAAAAAAAAull;
        x.rlo = vl ^ (0xAAAAAAAAAAAAAAAAull & mask_lo);
        if (x.er && a.ranged) {
            const u32 rb = (u32)(x.rhi >> a.range_shift);
            x.er = rb >= a.range_lo && rb < a.range_hi;
        }
    }
    return x;
}

// Count pass (WRITE = false): records and forward windows per tile, has_kmer flags.  Write pass: the records of tile t
// behind the ST_NREC records of earlier batches, at tile_off[t], in window order, the forward record of a window before its reverse complement's.
template <bool WRITE>
__global__ __launch_bounds__(EX_BLOCK) void wide_extract_kernel(WideArgs wa) {
    __shared__ u32 pk[(EX_TILE + EXW_EXT) / 16 + 2];
    __shared__ u32 inv[(EX_TILE + EXW_EXT) / 32 + 2];
    __shared__ u32 stb[(EX_TILE + EXW_EXT) / 32 + 2];
    __shared__ u32 seqrel[EX_TILE + EXW_EXT];
    __shared__ u32 wave_cnt[EX_ROUNDS][EX_BLOCK / 64];
    __shared__ u32 scan_smem[4];
    __shared__ u32 blk_fwd;
    const ExtractArgs &a = wa.e;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 t0 = (u64)blockIdx.x * EX_TILE;
    if (tid == 0) blk_fwd = 0;
    const u32 s_lo = ex_load_tile_ext<EXW_EXT>(a, blockIdx.x, pk, inv, stb, seqrel, scan_smem);
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u32 fmask = 0, rmask = 0, wave_fwd = 0;
    for (int j = 0; j < EX_ROUNDS; j++) {
        const u32 p = j * EX_BLOCK + tid;
        const WideSite x = wide_site(a, pk, inv, stb, seqrel, s_lo, t0, p);
        // reads are numbered by the mapper, which runs when the k-mer is built: whatever a key range keeps of it
        if (!WRITE && x.valid && a.has_kmer) a.has_kmer[x.seq] = 1;
        fmask |= (u32)x.ef << j;
        rmask |= (u32)x.er << j;
        const u64 bf = __ballot(x.ef), br = __ballot(x.er);
        if (lane == 0) wave_cnt[j][wave] = (u32)__popcll(bf) + (u32)__popcll(br);
        wave_fwd += (u32)__popcll(bf);
    }
    if (!WRITE && lane == 0 && wave_fwd) atomicAdd(&blk_fwd, wave_fwd);
    __syncthreads();
    if (tid < 64) {     // 64 (round, wave) counts in position order: exclusive scan
        const u32 v = wave_cnt[tid >> 2][tid & 3];
        u32 inc = v;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        if (WRITE) wave_cnt[tid >> 2][tid & 3] = inc - v;
        else if (tid == 63) {
            a.tile_count[blockIdx.x] = inc;
            a.tile_fwd[blockIdx.x] = blk_fwd;
        }
    }
    if (!WRITE) return;
    __syncthreads();
    const u64 tile_base = a.state[ST_NREC] + a.tile_off[blockIdx.x];
    for (int j = 0; j < EX_ROUNDS; j++) {
        const bool ef = (fmask >> j) & 1u, er = (rmask >> j) & 1u;
        const u64 bf = __ballot(ef), br = __ballot(er);
        if (!(ef || er)) continue;
        const u32 p = j * EX_BLOCK + tid;
        const WideSite x = wide_site(a, pk, inv, stb, seqrel, s_lo, t0, p);
        u64 slot = tile_base + wave_cnt[j][wave] + (u32)__popcll(bf & lt) + (u32)__popcll(br & lt);
        const u64 val = a.kind == KMG_PAYLOAD_TAXON ? (u64)(u32)a.seq_id[x.seq]
                                                    : (((u64)(u32)a.read_index[x.seq] << 16) | (u64)(x.pos & 0xFFFFu));
        for (int half = 0; half < 2; half++) {
            if (!(half ? er : ef)) continue;
            a.out_keys[slot] = half ? x.rhi : x.hi;
            wa.out_lo[slot] = half ? x.rlo : x.lo;
            if (a.kind == KMG_PAYLOAD_TAXON) reinterpret_cast<u32 *>(a.out_vals)[slot] = (u32)val;
            else reinterpret_cast<u64 *>(a.out_vals)[slot] = val;
            slot++;
        }
    }
}

// ---- the k-mers SHORTER than k of a build of several lengths (kmg_build_set_min_k with k > 32) ---------------------------
// extract_short.h's rules -- the `ending` branch of KmersFromFileSpliterator.tryAdvance, OtherFiles/KmersFromFile.java:90-133,
// 167-176: the suffixes of length min(pos, k - 1) .. minK that end at a sequence's last base (C), and those that end at the
// last base of a full window with an invalid base in it (B), both when the last full window of a sequence is invalid --
// for windows of up to 64 bases.  A suffix of L <= 32 bases is (hi, 0), a longer one (hi, lo); its length travels
// beside it.  Sites are rare (one position in a hundred on reads): a thread owns a position and walks its items.
struct WideShortArgs {
    ExtractArgs e;        // k = maxK (33 .. 64); out_keys = the hi words; tile_count / tile_fwd / tile_off: the short records'
    int min_k;
    u64 *out_lo;
    uint8_t *out_len;     // length of every record of the build (the write pass of the full windows leaves k there)
};

// the L bases (1 .. 64) from tile position q, left-aligned and zero-padded
__device__ __forceinline__ void wide_kmer_at(const u32 *pk, u32 q, int L, u64 &hi, u64 &lo) {
    const u32 w = q >> 4, sh = 2 * (q & 15);
    const u64 a0 = ((u64)pk[w] << 32) | pk[w + 1], a1 = ((u64)pk[w + 2] << 32) | pk[w + 3], a2 = (u64)pk[w + 4] << 32;
    hi = sh ? (a0 << sh) | (a1 >> (64 - sh)) : a0;
    lo = sh ? (a1 << sh) | (a2 >> (64 - sh)) : a1;
    if (L <= 32) { lo = 0; hi &= L == 32 ? ~0ull : ~(~0ull >> (2 * L)); }
    else lo &= L == 64 ? ~0ull : ~(~0ull >> (2 * (L - 32)));
}
// reverse complement of a k-mer of L bases (Kmer.inplaceRC), and Kmer.isOwnRC's palindrome test
__device__ __forceinline__ void wide_rc(u64 hi, u64 lo, int L, u64 &rhi, u64 &rlo, bool &pal) {
    const u64 rh = revpairs64(lo), rl = revpairs64(hi);      // the 64 base slots reversed: the L bases sit at the bottom
    const int s = 2 * (64 - L);
    u64 vh, vl;
    if (s == 0) { vh = rh; vl = rl; }
    else if (s < 64) { vh = (rh << s) | (rl >> (64 - s)); vl = rl << s; }
    else if (s == 64) { vh = rl; vl = 0; }
    else { vh = rl << (s - 64); vl = 0; }
    pal = (L & 1) == 0 && vh == hi && vl == lo;
    const u64 mh = L >= 32 ? ~0ull : ~(~0ull >> (2 * L));
    const u64 ml = L <= 32 ? 0ull : (L == 64 ? ~0ull : ~(~0ull >> (2 * (L - 32))));
    rhi = vh ^ (0xAAAAAAAAAAAAAAAAull & mh);
    rlo = vl ^ (0xAAAAAAAAAAAAAAAAull & ml);
}

struct WideShortSite {
    int n;            // bases the suffixes are cut from (k, or the whole of a sequence shorter than k)
    int d, nB;        // first nB items: lengths d, d - 1, ...
    int top, nC;      // next nC items: lengths top, top - 1, ...
    u32 seq;
    u64 seq_start;
};
__device__ __forceinline__ WideShortSite wide_short_site(const ExtractArgs &a, int m, const u32 *inv, const u32 *stb,
                                                         const u32 *seqrel, u32 s_lo, u64 t0, u32 p) {
    WideShortSite x;
    x.n = a.k; x.d = x.nB = x.top = x.nC = 0; x.seq = 0; x.seq_start = 0;
    const int W = a.k;
    const u64 gp = t0 + p;
    if (gp >= a.nbases) return x;
    const u32 w2 = p >> 5, r2 = p & 31;
    const u64 b0 = ((u64)inv[w2] << 32) | inv[w2 + 1], b1 = (u64)inv[w2 + 2] << 32;
    const u64 s0 = ((u64)stb[w2] << 32) | stb[w2 + 1], s1 = (u64)stb[w2 + 2] << 32;
    const u64 bad = r2 ? (b0 << r2) | (b1 >> (64 - r2)) : b0;     // bit 63 - i: base p + i is invalid
    const u64 st = r2 ? (s0 << r2) | (s1 >> (64 - r2)) : s0;      // bit 63 - i: a sequence starts at p + i
    x.seq = s_lo + seqrel[p];
    x.seq_start = a.seq_off[x.seq];
    const bool inside = ((st << 1) >> (65 - W)) == 0 && gp + (u64)W <= a.nbases;
    if (inside) {
        const u64 badW = bad >> (64 - W);
        x.d = badW ? __ffsll((long long)badW) - 1 : W;                       // valid bases in a row at the window's end
        if (badW && x.d >= m) x.nB = x.d - m + 1;                            // (B): lengths d .. m
        const u32 e = p + (u32)W;                                            // a sequence starts right behind the window?
        const bool last = gp + (u64)W == a.nbases || ((stb[e >> 5] >> (31 - (e & 31))) & 1u);
        x.top = x.d < W - 1 ? x.d : W - 1;
        if (last && x.top >= m) x.nC = x.top - m + 1;                        // (C), sequence of n >= W bases
    } else if (x.seq_start == gp) {
        // a sequence of fewer than W bases starts here: its (C), if it has at least m bases
        const u64 n64 = a.seq_off[x.seq + 1] - gp;
        if (n64 >= (u64)m && n64 < (u64)W) {
            x.n = (int)n64;
            const u64 badn = bad >> (64 - x.n);
            x.d = badn ? __ffsll((long long)badn) - 1 : x.n;
            if (x.d >= m) x.nB = x.d - m + 1;
        }
    }
    return x;
}

template <bool WRITE>
__global__ __launch_bounds__(EX_BLOCK) void wide_short_kernel(WideShortArgs sa) {
    __shared__ u32 pk[(EX_TILE + EXW_EXT) / 16 + 8];
    __shared__ u32 inv[(EX_TILE + EXW_EXT) / 32 + 2];
    __shared__ u32 stb[(EX_TILE + EXW_EXT) / 32 + 2];
    __shared__ u32 seqrel[EX_TILE + EXW_EXT];
    __shared__ u32 scan_smem[4];
    __shared__ u32 blk_fwd;
    const ExtractArgs &a = sa.e;
    const int tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * EX_TILE;
    if (tid == 0) blk_fwd = 0;
    if (tid < 8) pk[(EX_TILE + EXW_EXT) / 16 + tid] = 0;       // (words behind the image: read, then masked off)
    const u32 s_lo = ex_load_tile_ext<EXW_EXT>(a, blockIdx.x, pk, inv, stb, seqrel, scan_smem);
    const u64 tile_base = WRITE ? a.state[ST_NREC] + a.tile_off[blockIdx.x] : 0;
    u32 running = 0, my_fwd = 0;
    for (int j = 0; j < EX_ROUNDS; j++) {
        const u32 p = j * EX_BLOCK + tid;
        const WideShortSite x = wide_short_site(a, sa.min_k, inv, stb, seqrel, s_lo, t0, p);
        const int T = x.nB + x.nC;
        u32 cnt = 0;
        if (T) {
            // reads are numbered by the mapper, which runs when the k-mer is built: whatever a key range keeps of it
            if (!WRITE && a.has_kmer) a.has_kmer[x.seq] = 1;
            for (int it = 0; it < T; it++) {
                const int L = it < x.nB ? x.d - it : x.top - (it - x.nB);
                u64 hi, lo, rhi, rlo;
                bool pal;
                wide_kmer_at(pk, p + (u32)(x.n - L), L, hi, lo);
                const bool keep = !(a.dust >= 0 || a.runofsame >= 0) || wide_filters_keep(a, hi, lo, L);
                bool ef = keep, er = false;
                if (ef && a.ranged) { const u32 fb = (u32)(hi >> a.range_shift); ef = fb >= a.range_lo && fb < a.range_hi; }
                if (keep && a.rc) {
                    wide_rc(hi, lo, L, rhi, rlo, pal);
                    er = !pal;
                    if (er && a.ranged) { const u32 rb = (u32)(rhi >> a.range_shift); er = rb >= a.range_lo && rb < a.range_hi; }
                }
                cnt += (u32)ef + (u32)er;
                my_fwd += (u32)ef;
            }
        }
        u32 total;
        const u32 incl = block_scan_inclusive(cnt, OpSum(), scan_smem, &total);
        if (WRITE && cnt) {
            u64 slot = tile_base + running + (incl - cnt);
            for (int it = 0; it < T; it++) {
                const int L = it < x.nB ? x.d - it : x.top - (it - x.nB);
                u64 hi, lo, rhi = 0, rlo = 0;
                bool pal = true;
                wide_kmer_at(pk, p + (u32)(x.n - L), L, hi, lo);
                const bool keep = !(a.dust >= 0 || a.runofsame >= 0) || wide_filters_keep(a, hi, lo, L);
                bool ef = keep, er = false;
                if (ef && a.ranged) { const u32 fb = (u32)(hi >> a.range_shift); ef = fb >= a.range_lo && fb < a.range_hi; }
                if (keep && a.rc) {
                    wide_rc(hi, lo, L, rhi, rlo, pal);
                    er = !pal;
                    if (er && a.ranged) { const u32 rb = (u32)(rhi >> a.range_shift); er = rb >= a.range_lo && rb < a.range_hi; }
                }
                const u32 pos = (u32)(t0 + p + (u64)(x.n - L) - x.seq_start);
                const u64 val = a.kind == KMG_PAYLOAD_TAXON ? (u64)(u32)a.seq_id[x.seq]
                                                            : (((u64)(u32)a.read_index[x.seq] << 16) | (u64)(pos & 0xFFFFu));
                for (int half = 0; half < 2; half++) {
                    if (!(half ? er : ef)) continue;
                    a.out_keys[slot] = half ? rhi : hi;
                    sa.out_lo[slot] = half ? rlo : lo;
                    sa.out_len[slot] = (uint8_t)L;
                    if (a.kind == KMG_PAYLOAD_TAXON) reinterpret_cast<u32 *>(a.out_vals)[slot] = (u32)val;
                    else reinterpret_cast<u64 *>(a.out_vals)[slot] = val;
                    slot++;
                }
            }
        }
        running += total;
    }
    if (!WRITE) {
        if (my_fwd) atomicAdd(&blk_fwd, my_fwd);
        __syncthreads();
        if (tid == 0) {
            a.tile_count[blockIdx.x] = running;
            a.tile_fwd[blockIdx.x] = blk_fwd;
        }
    }
}

// ---- ranks ----------------------------------------------------------------------------------------------------------
__global__ void wide_iota_kernel(u32 *__restrict__ idx, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}
__global__ void wide_gather_kernel(const u64 *__restrict__ src, const u32 *__restrict__ idx, u32 n, u64 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ void wide_gather8_kernel(const uint8_t *__restrict__ src, const u32 *__restrict__ idx, u32 n, uint8_t *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
// the length as a sort key of its own (k-mers of several lengths: (value, length) is Sequence.compareTo's order)
// k-mers of one length, ordered by their first words already (a stable sort: equal first words in record order): the
// runs of equal first words are short in sequencing data (the reads that cover one place), so each run is put
// in order by the second words where it lies -- stable, as a sort would be -- instead of a
// whole sort by the second words ahead of the first (every entry counts its place in its run: the run's entries are a few
// cache lines its neighbours read too).  wide_longrun_kernel says whether a run exceeds WO_RUN (low-complexity
// input: then the two sorts are taken).
#define WO_RUN 128
__global__ void wide_longrun_kernel(const u64 *__restrict__ his, u32 n, u32 *__restrict__ flag) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= WO_RUN && j < n && his[j] == his[j - WO_RUN]) *flag = 1u;
}
#define WR_BLOCK 256
#define WR_TILE 1024
// LENS: k-mers of several lengths -- inside a run the order is (second word, length), Sequence.compareTo's
template <bool LENS>
__global__ __launch_bounds__(WR_BLOCK) void wide_runs_kernel(const u64 *__restrict__ his, const u64 *__restrict__ los,
                                                             const uint8_t *__restrict__ lens, const u32 *__restrict__ order, u32 n,
                                                             u64 *__restrict__ los_out, uint8_t *__restrict__ lens_out,
                                                             u32 *__restrict__ order_out) {
    // the tile's words and WO_RUN on either side (no run reaches further: wide_longrun_kernel) in LDS
    __shared__ u64 l_h[WR_TILE + 2 * WO_RUN], l_l[WR_TILE + 2 * WO_RUN];
    __shared__ uint8_t l_n[LENS ? WR_TILE + 2 * WO_RUN : 1];
    const long long base = (long long)blockIdx.x * WR_TILE - WO_RUN;
    for (int t = threadIdx.x; t < WR_TILE + 2 * WO_RUN; t += WR_BLOCK) {
        const long long g = base + t;
        const bool in = g >= 0 && g < (long long)n;
        l_h[t] = in ? his[g] : 0ull;
        l_l[t] = in ? los[g] : 0ull;
        if (LENS) l_n[t] = in ? lens[g] : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WR_TILE / WR_BLOCK; r++) {
        const int t = WO_RUN + r * WR_BLOCK + threadIdx.x;
        const long long i = base + t;
        if (i >= (long long)n) break;
        const u64 h = l_h[t], l = l_l[t];
        const uint8_t ln = LENS ? l_n[t] : (uint8_t)0;
        // its place in its run: behind the earlier entries that are not larger and the later ones that are smaller (stable)
        int s = t, e = t + 1;
        u32 before = 0;
        while (base + s > 0 && l_h[s - 1] == h) {
            s--;
            before += LENS ? (l_l[s] < l || (l_l[s] == l && l_n[s] <= ln)) : (l_l[s] <= l);
        }
        while (base + e < (long long)n && l_h[e] == h) {
            before += LENS ? (l_l[e] < l || (l_l[e] == l && l_n[e] < ln)) : (l_l[e] < l);
            e++;
        }
        const long long at = base + s + before;
        los_out[at] = l;
        if (LENS) lens_out[at] = ln;
        order_out[at] = order[i];
    }
}
__global__ void wide_lenkey_kernel(const uint8_t *__restrict__ len, u32 n, u64 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u64)len[i] << 56;
}
// his / los (/ lens, may be null: one length): the records' words in k-mer order.  flag[j] = record j starts a new k-mer
__global__ void wide_heads_kernel(const u64 *__restrict__ his, const u64 *__restrict__ los, const uint8_t *__restrict__ lens, u32 n,
                                  u32 *__restrict__ flag) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[j] = j == 0 || his[j] != his[j - 1] || los[j] != los[j - 1] || (lens && lens[j] != lens[j - 1]);
}
// before[j] = heads before j (exclusive scan of flag).  rank_key of record perm[j] = its k-mer's rank, left-aligned in
// `rbits` bits; the distinct k-mers in order.
__global__ void wide_rank_kernel(const u64 *__restrict__ his, const u64 *__restrict__ los, const u32 *__restrict__ perm,
                                 const u32 *__restrict__ flag, const u64 *__restrict__ before, u32 n, int rbits,
                                 u64 *__restrict__ rank_key, u64 *__restrict__ d_hi, u64 *__restrict__ d_lo,
                                 const uint8_t *__restrict__ lens = nullptr, uint8_t *__restrict__ d_len = nullptr) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u64 r = before[j] + flag[j] - 1;
    rank_key[perm[j]] = r << (64 - rbits);
    if (flag[j]) { d_hi[r] = his[j]; d_lo[r] = los[j]; if (d_len) d_len[r] = lens[j]; }
}

// Read databases whose records arrive with ascending (read, pos) (no two sequences shared a read index): the two index
// sorts are stable, so inside a k-mer the positions already ascend and the CSR follows from the heads alone --
// val_off[r] = first record of k-mer r, the values = the payloads gathered in k-mer order.
__global__ void wide_csr_reads_kernel(const u64 *__restrict__ his, const u64 *__restrict__ los, const u32 *__restrict__ flag,
                                      const u64 *__restrict__ before, u32 n, u64 *__restrict__ d_hi, u64 *__restrict__ d_lo,
                                      u64 *__restrict__ val_off, const uint8_t *__restrict__ lens = nullptr,
                                      uint8_t *__restrict__ d_len = nullptr) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !flag[j]) return;
    const u64 r = before[j];
    d_hi[r] = his[j]; d_lo[r] = los[j]; val_off[r] = j;
    if (d_len) d_len[r] = lens[j];
}

// KmerUtils.restrictedStream on two-word k-mers (KmerUtils.java:59-74): the entries of at least L bases (keep[i] != 0),
// cut to L bases (wide_cut_kernel's rule), side by side in their order, and where each came from
// (lens / out_len, may be null: with them the cut k-mers keep min(length, L) -- KmerUtils.restrictedStream with two lengths
// leaves the shorter ones as they are; L <= 32 then cuts into the first word)
__global__ void wide_compact_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo, const u32 *__restrict__ keep,
                                    const u64 *__restrict__ pos, u64 n, int L, u64 *__restrict__ out_hi, u64 *__restrict__ out_lo,
                                    u32 *__restrict__ from, const uint8_t *__restrict__ lens = nullptr, int len_all = 0,
                                    uint8_t *__restrict__ out_len = nullptr) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const u64 j = pos[i];
    out_hi[j] = L >= 32 ? hi[i] : hi[i] & ~(~0ull >> (2 * L));
    out_lo[j] = L >= 64 ? lo[i] : (L <= 32 ? 0ull : lo[i] & ~(~0ull >> (2 * (L - 32))));
    from[j] = (u32)i;
    if (out_len) { const int len = lens ? (int)lens[i] : len_all; out_len[j] = (uint8_t)(len < L ? len : L); }
}
__global__ void wide_spread_kernel(const u64 *__restrict__ vals, const u32 *__restrict__ to, u64 n, u64 *__restrict__ out) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[to[j]] = vals[j];
}
// Kmer.limitTo (Kmer.java:277-289) on the second word: the bases behind the L-th (32 < L <= 64) go
__global__ void wide_cut_kernel(const u64 *__restrict__ lo, u64 n, int L, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = L >= 64 ? lo[i] : lo[i] & ~(~0ull >> (2 * (L - 32)));
}

// ---- exact join (DB.quickMatchCommonKey, Database/DB.java:129-137: equal k-mers of one key bucket) -----------------------
// hit[i] = index of search k-mer i in the reference, or -1.  Both sides ascend as (hi, lo) pairs, so the reference
// k-mers that a tile of WM_TILE consecutive search k-mers can equal are one contiguous piece: two threads find it (a
// lower and an upper bound over the whole reference), the workgroup loads it into LDS if it has at most WM_CAP k-mers,
// and every search k-mer is looked up there; a longer piece (a reference much denser than the search side) is searched
// in place, inside the piece.
#define WM_BLOCK 256
#define WM_ITEMS 4
#define WM_TILE (WM_BLOCK * WM_ITEMS)
#define WM_CAP 2048
__device__ __forceinline__ bool wide_less(u64 ah, u64 al, u64 bh, u64 bl) { return ah < bh || (ah == bh && al < bl); }
__global__ __launch_bounds__(WM_BLOCK) void wide_match_kernel(const u64 *__restrict__ s_hi, const u64 *__restrict__ s_lo, u64 n,
                                                               const u64 *__restrict__ r_hi, const u64 *__restrict__ r_lo, u64 R,
                                                               long long *__restrict__ hit, u64 *__restrict__ nhits) {
    __shared__ u64 l_hi[WM_CAP], l_lo[WM_CAP];
    __shared__ u64 piece[2];
    const int tid = threadIdx.x;
    const u64 i0 = (u64)blockIdx.x * WM_TILE, i1 = min(n, i0 + (u64)WM_TILE);
    if (tid < 128) {
        // two waves find the piece, one bound each: 64 samples of the range a round (the ballot's count of samples still
        // before the bound names the sixty-fourth that holds it) -- four or five dependent loads where a bisection takes 23
        const bool upper = tid >= 64;
        const int lane = tid & 63;
        const u64 kh = upper ? s_hi[i1 - 1] : s_hi[i0], kl = upper ? s_lo[i1 - 1] : s_lo[i0];
        u64 lo = 0, hi = R;
        while (lo < hi) {
            const u64 step = (hi - lo) / 64 + 1;
            const u64 pos = lo + (u64)lane * step;
            bool before = false;
            if (pos < hi) {
                const u64 mh = r_hi[pos], ml = r_lo[pos];
                // lower bound: first reference k-mer >= key; upper bound: first one > key
                before = upper ? !wide_less(kh, kl, mh, ml) : wide_less(mh, ml, kh, kl);
            }
            const int cnt = __popcll(__ballot(before));
            const u64 nhi = min(hi, lo + (u64)cnt * step);
            lo = cnt ? lo + (u64)(cnt - 1) * step + 1 : lo;
            hi = nhi;
        }
        if (lane == 0) piece[upper ? 1 : 0] = lo;
    }
    __syncthreads();
    const u64 p0 = piece[0], p1 = piece[1];
    const bool in_lds = p1 - p0 <= WM_CAP;
    const u32 span = in_lds ? (u32)(p1 - p0) : 0u;
    for (u32 j = tid; j < span; j += WM_BLOCK) { l_hi[j] = r_hi[p0 + j]; l_lo[j] = r_lo[p0 + j]; }
    __syncthreads();
    u32 found_n = 0;
    // a thread's WM_ITEMS search k-mers are neighbours: the place of one bounds the next from below, which lies a step or
    // two further on (an exponential search from there, then a bisection of the last stride).  All of them are loaded
    // before the first is looked up (clamped, not guarded: guarded loads wait one by one).
    u64 kh[WM_ITEMS], kl[WM_ITEMS];
    long long res[WM_ITEMS];
#pragma unroll
    for (int r = 0; r < WM_ITEMS; r++) {
        const u64 i = i0 + (u64)tid * WM_ITEMS + r;
        const u64 c = i < i1 ? i : i1 - 1;
        kh[r] = s_hi[c]; kl[r] = s_lo[c];
        res[r] = -1;
    }
    u32 from = 0;
    u64 gfrom = p0;
#pragma unroll
    for (int r = 0; r < WM_ITEMS; r++) {
        const u64 h = kh[r], l = kl[r];
        if (in_lds) {
            u32 a = from, b = from, step = 1;
            while (b < span && wide_less(l_hi[b], l_lo[b], h, l)) { a = b + 1; b += step; step <<= 1; }
            b = min(b, span);
            while (a < b) {
                const u32 mid = (a + b) >> 1;
                if (wide_less(l_hi[mid], l_lo[mid], h, l)) a = mid + 1; else b = mid;
            }
            if (a < span && l_hi[a] == h && l_lo[a] == l) res[r] = (long long)(p0 + a);
            from = a;
        } else {
            u64 a = gfrom, b = p1;
            while (a < b) {
                const u64 mid = (a + b) >> 1;
                if (wide_less(r_hi[mid], r_lo[mid], h, l)) a = mid + 1; else b = mid;
            }
            if (a < p1 && r_hi[a] == h && r_lo[a] == l) res[r] = (long long)a;
            gfrom = a;
        }
    }
#pragma unroll
    for (int r = 0; r < WM_ITEMS; r++) {
        const u64 i = i0 + (u64)tid * WM_ITEMS + r;
        if (i < i1) { hit[i] = res[r]; found_n += res[r] >= 0; }
    }
    if (nhits) {          // (uniform) one addition a workgroup
        __shared__ u32 l_found[WM_BLOCK / 64];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) found_n += __shfl_down(found_n, d, 64);
        if ((tid & 63) == 0) l_found[tid >> 6] = found_n;
        __syncthreads();
        if (tid == 0) {
            u32 total = 0;
            for (int w = 0; w < WM_BLOCK / 64; w++) total += l_found[w];
            if (total) atomicAdd((unsigned long long *)nhits, (unsigned long long)total);
        }
    }
}

// ---- search over k-mers of several lengths, no differences: match.h's trie_find_kernel -- the closed form of what the
// reference's radix trie hands back (DB.java:83,139-184 over Root.java:29-170; derivation in DESIGN 6b) -- on two words.
// x0 = the first stored k-mer that has the search k-mer t (T bases) as a prefix; the answer's data object has absorbed
// the stored k-mers [x0, first entry whose first m bases are larger), m = max(shortest, L + 1), L = the bases x0 shares
// with its predecessor in the key.  One file, or a set whose files share no key.
__global__ void trie_find_wide_kernel(const u64 *__restrict__ s_hi, const u64 *__restrict__ s_lo, const uint8_t *__restrict__ slens,
                                      int s_len_all, u64 ns, const u64 *__restrict__ r_hi, const u64 *__restrict__ r_lo,
                                      const uint8_t *__restrict__ rlens, int r_len_all, const u64 *__restrict__ rbucket,
                                      const u64 *__restrict__ rval_off, int key_len, int shortest, long long *__restrict__ first,
                                      u32 *__restrict__ found, u32 *__restrict__ nvals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const u64 th = s_hi[i], tl = s_lo ? s_lo[i] : 0ull;      // (a search side of at most 32 bases has no second word)
    const int T = slens ? (int)slens[i] : s_len_all;
    const u64 b = th >> (64 - 2 * key_len);
    const u64 lo = rbucket[b], hi = rbucket[b + 1];
    u64 l = lo, h = hi;
    while (l < h) {                                // first entry >= (t, T) in (value, length) order
        const u64 mid = (l + h) >> 1;
        const u64 kh = r_hi[mid], kl = r_lo[mid];
        const int lm = rlens ? (int)rlens[mid] : r_len_all;
        if (wide_less(kh, kl, th, tl) || (kh == th && kl == tl && lm < T)) l = mid + 1; else h = mid;
    }
    first[i] = -1; found[i] = 0; nvals[i] = 0;
    if (l == hi) return;
    const u64 xh = r_hi[l], xl = r_lo[l];
    const int len0 = rlens ? (int)rlens[l] : r_len_all;
    // t a prefix of x0?  (if not: of no stored k-mer)
    const bool pre = T <= 32 ? ((xh ^ th) >> (64 - 2 * T)) == 0 : (xh == th && (T >= 64 ? xl == tl : ((xl ^ tl) >> (64 - 2 * (T - 32))) == 0));
    if (len0 < T || !pre) return;
    int L = 0;
    if (l > lo) {
        const u64 dh = xh ^ r_hi[l - 1], dl = xl ^ r_lo[l - 1];
        const int lp = rlens ? (int)rlens[l - 1] : r_len_all;
        const int c = dh ? (int)(__clzll((long long)dh) >> 1) : 32 + (dl ? (int)(__clzll((long long)dl) >> 1) : 32);
        L = min(c, min(len0, lp));
    }
    const int m = max(shortest, L + 1);
    // the largest value with x0's first m bases
    const u64 toph = m >= 32 ? xh : (xh | (~0ull >> (2 * m)));
    const u64 topl = m >= 64 ? xl : (m > 32 ? (xl | (~0ull >> (2 * (m - 32)))) : ~0ull);
    u64 a = l + 1, e = hi;
    while (a < e) {
        const u64 mid = (a + e) >> 1;
        if (!wide_less(toph, topl, r_hi[mid], r_lo[mid])) a = mid + 1; else e = mid;
    }
    first[i] = (long long)l;
    found[i] = 1;
    const u64 nv = rval_off[a] - rval_off[l];
    nvals[i] = nv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)nv;
}

// ---- block encoding (FileCreator.java:312-347) of two-word k-mers ---------------------------------------------------------
// compressedBytes = [k][cBytes: hi then lo, big-endian, l = ceil(k / 4) bytes]; s = bytes shared with the previous k-mer of
// the bucket (FileCreator.shared, :244-252)
__device__ __forceinline__ int wide_enc_shared(u64 ph, u64 pl, u64 ch, u64 cl, int l) {
    int same;
    const u64 xh = ph ^ ch;
    if (xh) same = __clzll((long long)xh) >> 3;
    else {
        const u64 xl = pl ^ cl;
        same = 8 + (xl ? (__clzll((long long)xl) >> 3) : 8);
    }
    if (same > l) same = l;
    return 1 + same;
}
__device__ __forceinline__ uint8_t wide_byte(u64 h, u64 l, int i) {
    return (uint8_t)(i < 8 ? h >> (56 - 8 * i) : l >> (56 - 8 * (i - 8)));
}
// (lens, may be null: every k-mer has k bases; k-mers of different lengths share nothing, their length bytes differ)
template <bool WRITE>
__global__ void wide_enc_kernel(const u64 *__restrict__ k_hi, const u64 *__restrict__ k_lo, const uint8_t *__restrict__ lens,
                                const u64 *__restrict__ val_off,
                                const int32_t *__restrict__ val_a, const int32_t *__restrict__ val_b, u64 U, int k_all, int key_len,
                                int kind, u32 *__restrict__ sizes, const u64 *__restrict__ enc_off, uint8_t *__restrict__ out) {
    const u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int k = lens ? lens[u] : k_all;
    const int l = (k - 1) / 4 + 1;
    const int sh = 64 - 2 * key_len;
    const u64 ch = k_hi[u], cl = k_lo[u];
    int s = 0;
    if (u > 0) {
        const u64 ph = k_hi[u - 1];
        if ((ph >> sh) == (ch >> sh) && (!lens || lens[u - 1] == k)) s = wide_enc_shared(ph, k_lo[u - 1], ch, cl, l);
    }
    const u64 a = val_off[u], b = val_off[u + 1];
    const u64 nv = b - a;
    if (!WRITE) {
        sizes[u] = (u32)(1 + (u64)(l + 1 - s) + (kind == KMG_PAYLOAD_TAXON ? 2 + 5 * nv : 4 + 6 * nv));
        return;
    }
    uint8_t *p = out + enc_off[u];
    *p++ = (uint8_t)s;
    if (s == 0) *p++ = (uint8_t)k;
    for (int i = (s == 0 ? 0 : s - 1); i < l; i++) *p++ = wide_byte(ch, cl, i);
    if (kind == KMG_PAYLOAD_TAXON) {
        *p++ = (uint8_t)(nv >> 8);
        *p++ = (uint8_t)nv;
        for (u64 v = a; v < b; v++) {
            const u32 t = (u32)val_a[v];
            const int32_t c = val_b[v];
            *p++ = (uint8_t)(t >> 24); *p++ = (uint8_t)(t >> 16); *p++ = (uint8_t)(t >> 8); *p++ = (uint8_t)t;
            *p++ = (uint8_t)(c > 255 ? 255 : c);   // CountDataType.java:40
        }
    } else {
        *p++ = (uint8_t)(nv >> 24); *p++ = (uint8_t)(nv >> 16); *p++ = (uint8_t)(nv >> 8); *p++ = (uint8_t)nv;
        for (u64 v = a; v < b; v++) {
            const u32 r = (u32)val_a[v];
            const u32 q = (u32)val_b[v];
            *p++ = (uint8_t)(r >> 24); *p++ = (uint8_t)(r >> 16); *p++ = (uint8_t)(r >> 8); *p++ = (uint8_t)r;
            *p++ = (uint8_t)(q >> 8); *p++ = (uint8_t)q;
        }
    }
}

// a loaded database's k-mers must ascend strictly as (hi, lo) pairs
__global__ void wide_ascending_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo, u64 n, u32 *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i < n && (hi[i - 1] > hi[i] || (hi[i - 1] == hi[i] && lo[i - 1] >= lo[i]))) atomicAdd(bad, 1u);
}

// ---- inexact search (Matcher -n) on two words: the trie walk of Root.closestKmers (Root.java:115-170) over the sorted
// (hi, lo) pairs, as near_walk_kernel (match.h) does on one word -- a node = (range of stored k-mers sharing a prefix,
// depth, differences so far), children by the next base in base order, so the matches come out ascending.  Every number
// of differences takes this walk (the enumeration of neighbours that serves one or two differences on one word is not
// built for two).
#define WNW_MAXK 64
__device__ __forceinline__ void wide_prefix_mask(int d, u64 &mh, u64 &ml) {             // the first d bases, 0 <= d <= 64
    mh = d >= 32 ? ~0ull : (d == 0 ? 0ull : ~(~0ull >> (2 * d)));
    ml = d <= 32 ? 0ull : (d >= 64 ? ~0ull : ~(~0ull >> (2 * (d - 32))));
}
__device__ __forceinline__ u64 wide_lower(const u64 *__restrict__ rh, const u64 *__restrict__ rl, u64 lo, u64 hi, u64 wh, u64 wl) {
    while (lo < hi) {                                  // first pair in [lo, hi) that is not below (wh, wl)
        const u64 mid = lo + ((hi - lo) >> 1);
        if (wide_less(rh[mid], rl[mid], wh, wl)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <bool FILL>
__global__ __launch_bounds__(128) void wide_near_walk_kernel(const u64 *__restrict__ s_hi, const u64 *__restrict__ s_lo, u64 ns, int k,
                                                             int max_diff, const u64 *__restrict__ r_hi, const u64 *__restrict__ r_lo,
                                                             u64 nr, const u64 *__restrict__ rbucket, int key_len,
                                                             u32 *__restrict__ cnt, const u64 *__restrict__ off,
                                                             long long *__restrict__ out_ref, uint8_t *__restrict__ out_dist) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns || nr == 0) return;
    const u64 kh = s_hi[s], kl = s_lo[s];
    u64 found = 0;
    const u64 base_out = FILL ? off[s] : 0;
    auto emit = [&](u64 r, int d) {
        if (FILL) { out_ref[base_out + found] = (long long)r; out_dist[base_out + found] = (uint8_t)d; }
        found++;
    };
    auto differing = [&](u64 r) {
        const u64 x = r_hi[r] ^ kh, y = r_lo[r] ^ kl;
        return __popcll((x | (x >> 1)) & 0x5555555555555555ull) + __popcll((y | (y >> 1)) & 0x5555555555555555ull);
    };
    u64 f_lo[WNW_MAXK + 1], f_hi[WNW_MAXK + 1], f_cur[WNW_MAXK + 1];
    uint8_t f_next[WNW_MAXK + 1], f_diff[WNW_MAXK + 1];
    int d = 0;
    f_lo[0] = 0; f_hi[0] = nr; f_cur[0] = 0; f_next[0] = 0; f_diff[0] = 0;
    while (d >= 0) {
        const u64 lo = f_lo[d], hi = f_hi[d];
        const int used = f_diff[d];
        if (f_next[d] == 0) {
            if (hi - lo == 1) {
                const int dist = differing(lo);
                if (dist <= max_diff) emit(lo, dist);
                d--;
                continue;
            }
            if (used == max_diff) {                    // the rest must be the search k-mer's own
                u64 mh, ml;
                wide_prefix_mask(d, mh, ml);
                const u64 wh = (r_hi[lo] & mh) | (kh & ~mh), wl = (r_lo[lo] & ml) | (kl & ~ml);
                const u64 at = wide_lower(r_hi, r_lo, lo, hi, wh, wl);
                if (at < hi && r_hi[at] == wh && r_lo[at] == wl) emit(at, used);
                d--;
                continue;
            }
            f_cur[d] = lo;
        }
        if (f_next[d] == 4) { d--; continue; }
        const int c = f_next[d]++;
        const u64 c_lo = f_cur[d];
        u64 c_hi;
        if (c == 3) c_hi = hi;
        else if (d < key_len) {                        // (the key lies in the first word)
            const u64 pre = (r_hi[lo] >> (64 - 2 * (d + 1)) & ~3ull) | (u64)c;
            c_hi = rbucket[(pre + 1) << (2 * (key_len - d - 1))];
            c_hi = c_hi < c_lo ? c_lo : (c_hi > hi ? hi : c_hi);
        } else {
            u64 mh, ml, nh, nl;
            wide_prefix_mask(d, mh, ml);
            wide_prefix_mask(d + 1, nh, nl);
            u64 ph = r_hi[lo] & mh, pl = r_lo[lo] & ml;
            if (d < 32) ph |= (u64)c << (62 - 2 * d); else pl |= (u64)c << (62 - 2 * (d - 32));
            const u64 lh = ph | ~nh, ll = pl | ~nl;    // the largest pair with that prefix
            c_hi = wide_lower(r_hi, r_lo, c_lo, hi, lh, ll);
            if (c_hi < hi && r_hi[c_hi] == lh && r_lo[c_hi] == ll) c_hi++;
        }
        f_cur[d] = c_hi;
        if (c_hi == c_lo) continue;
        const int sb = d < 32 ? (int)((kh >> (62 - 2 * d)) & 3) : (int)((kl >> (62 - 2 * (d - 32))) & 3);
        const int cost = sb != c;
        if (used + cost > max_diff) continue;
        if (d + 1 == k) { emit(c_lo, used + cost); continue; }
        d++;
        f_lo[d] = c_lo; f_hi[d] = c_hi; f_next[d] = 0; f_diff[d] = (uint8_t)(used + cost);
    }
    if (!FILL) cnt[s] = (u32)(found > 0xFFFFFFFFull ? 0xFFFFFFFFull : found);
}

// ---- the search over several lengths across a SET of files that share keys, on two words: match.h's trie_files_kernel
// (the second closed form, derived in DESIGN 6b/6c: the answer's object belongs to the earliest inserted stored k-mer x that
// has the search k-mer as a prefix, and has absorbed, per boundary depth on x's path, the k-mers inserted at or after the
// boundary's maker that share that many bases with x) with (hi, lo) compares, prefixes and common-prefix lengths.
struct WideTrieFile {
    TrieFile t;               // keys = the first words
    const u64 *keys_lo;
};
__device__ __forceinline__ int tfw_lcp(u64 ah, u64 al, int la, u64 bh, u64 bl, int lb) {
    const u64 x = ah ^ bh, y = al ^ bl;
    const int c = x ? (int)(__clzll((long long)x) >> 1) : (y ? 32 + (int)(__clzll((long long)y) >> 1) : 64);
    return min(c, min(la, lb));
}
// first entry of [lo, hi) that is >= (v, len) in (value, length) order
__device__ __forceinline__ u64 tfw_lower(const WideTrieFile &f, u64 lo, u64 hi, u64 vh, u64 vl, int len) {
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        const u64 kh = f.t.keys[mid], kl = f.keys_lo[mid];
        if (wide_less(kh, kl, vh, vl) || (kh == vh && kl == vl && tf_len(f.t, mid) < len)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the entries of [lo, hi) that have the first e bases of x: -> [a, b)
__device__ __forceinline__ void tfw_prefix_range(const WideTrieFile &f, u64 lo, u64 hi, u64 xh, u64 xl, int e, u64 &a, u64 &b) {
    u64 mh, ml;
    wide_prefix_mask(e, mh, ml);
    const u64 vh = xh & mh, vl = xl & ml, th = vh | ~mh, tl = vl | ~ml;
    a = tfw_lower(f, lo, hi, vh, vl, e);
    u64 l = a, h = hi;
    while (l < h) {
        const u64 mid = (l + h) >> 1;
        if (!wide_less(th, tl, f.t.keys[mid], f.keys_lo[mid])) l = mid + 1; else h = mid;
    }
    b = l;
}

template <bool EMIT>
__global__ void trie_files_wide_kernel(const u64 *__restrict__ s_hi, const u64 *__restrict__ s_lo /* null: one word */,
                                       const uint8_t *__restrict__ slens, int s_len_all, u64 ns,
                                       const WideTrieFile *__restrict__ files, int nf, int key_len, int shortest,
                                       u32 *__restrict__ found, u64 *__restrict__ nvals, const u64 *__restrict__ row,
                                       const u64 *__restrict__ dst, int cap, u64 *__restrict__ out_keys, u64 *__restrict__ out_vals,
                                       long long *__restrict__ hit) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const u64 th = s_hi[i], tl = s_lo ? s_lo[i] : 0ull;
    const int T = slens ? (int)slens[i] : s_len_all;
    const u64 b = th >> (64 - 2 * key_len);
    u64 tmh, tml;
    wide_prefix_mask(T, tmh, tml);
    // x: the first file that holds a k-mer with prefix t, its smallest there
    int f0 = -1;
    u64 l0 = 0;
    for (int f = 0; f < nf && f0 < 0; f++) {
        const WideTrieFile &F = files[f];
        const u64 lo = F.t.bucket[b], hi = F.t.bucket[b + 1];
        const u64 l = tfw_lower(F, lo, hi, th, tl, T);
        if (l < hi && tf_len(F.t, l) >= T && ((F.t.keys[l] ^ th) & tmh) == 0 && ((F.keys_lo[l] ^ tl) & tml) == 0) { f0 = f; l0 = l; }
    }
    if (f0 < 0) {
        if (EMIT) hit[i] = -1;
        else { found[i] = 0; nvals[i] = 0; }
        return;
    }
    const u64 xh = files[f0].t.keys[l0], xl = files[f0].keys_lo[l0];
    const int X = tf_len(files[f0].t, l0);
    // L: the longest prefix x shares with anything inserted before it
    int L = 0;
    {
        const WideTrieFile &F = files[f0];
        if (l0 > F.t.bucket[b]) L = tfw_lcp(xh, xl, X, F.t.keys[l0 - 1], F.keys_lo[l0 - 1], tf_len(F.t, l0 - 1));
        for (int f = 0; f < f0; f++) {
            const WideTrieFile &G = files[f];
            const u64 lo = G.t.bucket[b], hi = G.t.bucket[b + 1];
            const u64 p = tfw_lower(G, lo, hi, xh, xl, X);
            if (p > lo) L = max(L, tfw_lcp(xh, xl, X, G.t.keys[p - 1], G.keys_lo[p - 1], tf_len(G.t, p - 1)));
            if (p < hi) L = max(L, tfw_lcp(xh, xl, X, G.t.keys[p], G.keys_lo[p], tf_len(G.t, p)));
        }
    }
    const int m = max(shortest, L + 1);
    u64 count = 0, o = EMIT ? dst[i] : 0;
    const u64 key = EMIT ? (row[i] << 32) : 0;
    tf_take<EMIT>(files[f0].t, l0, l0 + 1, count, key, cap, out_keys, out_vals, o);          // D_x itself
    u64 made[2] = {0, 0};                      // bit e (0 .. 64): some k-mer at or after x has made a boundary at depth e
    made[X >> 6] |= 1ull << (X & 63);
    {
        const WideTrieFile &F = files[f0];
        const u64 hi = F.t.bucket[b + 1];
        u64 prev = l0 + 1;
        for (int e = X; e >= m; e--) {
            u64 a_, b_;
            tfw_prefix_range(F, prev, hi, xh, xl, e, a_, b_);
            if (b_ > prev) {
                tf_take<EMIT>(F.t, prev, b_, count, key, cap, out_keys, out_vals, o);
                made[e >> 6] |= 1ull << (e & 63);
            }
            prev = max(prev, b_);
        }
    }
    for (int f = f0 + 1; f < nf; f++) {
        const WideTrieFile &F = files[f];
        u64 ra, rb;
        tfw_prefix_range(F, F.t.bucket[b], F.t.bucket[b + 1], xh, xl, m, ra, rb);
        for (int e = m; e <= X && ra < rb; e++) {
            if (e == X) {
                tf_take<EMIT>(F.t, ra, rb, count, key, cap, out_keys, out_vals, o);
                break;
            }
            u64 na, nb;
            tfw_prefix_range(F, ra, rb, xh, xl, e + 1, na, nb);
            if (made[e >> 6] & (1ull << (e & 63))) tf_take<EMIT>(F.t, ra, rb, count, key, cap, out_keys, out_vals, o);
            else if (ra < na) {
                tf_take<EMIT>(F.t, ra, rb, count, key, cap, out_keys, out_vals, o);
                made[e >> 6] |= 1ull << (e & 63);
            } else if (nb < rb) {
                tf_take<EMIT>(F.t, nb, rb, count, key, cap, out_keys, out_vals, o);
                made[e >> 6] |= 1ull << (e & 63);
            }
            ra = na; rb = nb;
        }
    }
    if (EMIT) hit[i] = (long long)row[i];
    else { found[i] = 1; nvals[i] = count; }
}
