// match.h -- stage 3: sorted-set join of search k-mers against a reference DB, and TaxaCounts.
//
// Replaces DB.quickMatchCommonKey + StreamUtils.matchTwoStreams (Database/DB.java:129-137, jar A.6: a
// two-pointer inner join per key bucket of two strictly ascending k-mer sequences) and
// TaxaCounts.CountDataAll.add (Utils/TaxaCounts.java:386-395).  Both sides are unique and sorted, so the
// join result is exactly "is this search k-mer present in the reference, and where".
#pragma once
#include "common.h"

#define MT_BLOCK 256
#define MT_ITEMS 8
#define MT_TILE (MT_BLOCK * MT_ITEMS)   // search k-mers per workgroup
#define MT_RMAX 3072                    // reference k-mers staged in LDS per tile (24 KB)

// bounds[t] = index of the first reference k-mer that is >= the first search k-mer of tile t (t < ntiles),
// bounds[ntiles] = nr.  Search k-mers of tile t lie in [first_t, first_{t+1}), so every reference k-mer that can
// match one of them lies in [bounds[t], bounds[t+1]).  One thread per tile: a binary search inside the k-mer's
// key bucket (rbucket), i.e. ~log2(bucket size) probes.
__global__ void match_bounds_kernel(const u64 *__restrict__ skeys, u64 ns, const u64 *__restrict__ rkeys, u64 nr,
                                    const u64 *__restrict__ rbucket, int key_len, u32 ntiles, u64 *__restrict__ bounds) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) { bounds[t] = nr; return; }
    const u64 key = skeys[(u64)t * MT_TILE];
    const u64 b = key >> (64 - 2 * key_len);
    u64 lo = rbucket[b], hi = rbucket[b + 1];
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    bounds[t] = lo;
}

// One workgroup joins MT_TILE consecutive search k-mers.  Both sides are ascending, so the reference k-mers
// that can match them form one contiguous range (match_bounds_kernel); the block stages it in LDS (it is a
// few hundred k-mers when the two sets are of similar density) and every search k-mer is then binary-searched
// in LDS instead of through L2.  A range that does not fit goes through LDS in pieces (a reference denser than the
// search set: BASELINE config 3 has 1.7 reference k-mers per search k-mer); only ranges beyond MT_CHUNKS pieces are
// searched in place.  hit[i] = index of the k-mer in the reference, or -1.
#define MT_CHUNKS 8                     // a tile's reference range is staged in up to this many LDS-sized pieces
__device__ __forceinline__ void match_tile(const u64 *__restrict__ skeys, u64 q0, u64 q1, const u64 *__restrict__ rkeys,
                                           u64 r_lo, u64 r_hi, long long *__restrict__ hit, u64 *__restrict__ nhits) {
    __shared__ u64 sref[MT_RMAX];
    __shared__ u32 smem[4];
    const u64 span = r_hi - r_lo;
    u32 nfound = 0;
    if (span <= (u64)MT_RMAX * MT_CHUNKS) {
        // Both sides ascend: the range goes through LDS piece by piece, and a search k-mer is looked up in the first
        // piece whose last k-mer is not below it (every earlier piece ends below it, every later one starts above).
        u64 key[MT_ITEMS];
        long long res[MT_ITEMS];
        u32 open = 0;
#pragma unroll
        for (int j = 0; j < MT_ITEMS; j++) {
            const u64 i = q0 + (u64)j * MT_BLOCK + threadIdx.x;
            res[j] = -1;
            key[j] = skeys[i < q1 ? i : q1 - 1];            // (clamped, not guarded: guarded loads wait one by one)
            if (i < q1) open |= 1u << j;
        }
        for (u64 c0 = 0; c0 < span; c0 += MT_RMAX) {
            const u32 cnt = (u32)(span - c0 < MT_RMAX ? span - c0 : MT_RMAX);
            if (c0) __syncthreads();                        // the previous piece's readers are done
            for (u32 b = 0; b < cnt; b += 4 * MT_BLOCK) {     // four loads in flight per lane
                u64 t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const u32 i = b + u * MT_BLOCK + threadIdx.x;
                    t[u] = rkeys[r_lo + c0 + (i < cnt ? i : cnt - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const u32 i = b + u * MT_BLOCK + threadIdx.x;
                    if (i < cnt) sref[i] = t[u];
                }
            }
            __syncthreads();
            const u64 last = sref[cnt - 1];
#pragma unroll
            for (int j = 0; j < MT_ITEMS; j++) {
                if (((open >> j) & 1) && key[j] <= last) {
                    u32 lo = 0, hi = cnt;
                    while (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        if (sref[mid] < key[j]) lo = mid + 1; else hi = mid;
                    }
                    if (lo < cnt && sref[lo] == key[j]) res[j] = (long long)(r_lo + c0 + lo);
                    open &= ~(1u << j);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MT_ITEMS; j++) {
            const u64 i = q0 + (u64)j * MT_BLOCK + threadIdx.x;
            if (i < q1) { hit[i] = res[j]; nfound += res[j] >= 0; }
        }
    } else {
        // a sparse search set over a dense reference: staging would read far more than the probes do
#pragma unroll
        for (int j = 0; j < MT_ITEMS; j++) {
            const u64 i = q0 + (u64)j * MT_BLOCK + threadIdx.x;
            if (i < q1) {
                const u64 key = skeys[i];
                long long res = -1;
                u64 lo = r_lo, hi = r_hi;
                while (lo < hi) {
                    const u64 mid = (lo + hi) >> 1;
                    if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo < r_hi && rkeys[lo] == key) res = (long long)lo;
                hit[i] = res;
                nfound += res >= 0;
            }
        }
    }
    if (nhits) {   // uniform
        const u64 bal0 = __ballot(nfound & 1), bal1 = __ballot(nfound & 2), bal2 = __ballot(nfound & 4), bal3 = __ballot(nfound & 8);
        const u32 wsum = __popcll(bal0) + 2 * __popcll(bal1) + 4 * __popcll(bal2) + 8 * __popcll(bal3);
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const u32 total = smem[0] + smem[1] + smem[2] + smem[3];
            if (total) atomicAdd(nhits, (u64)total);
        }
    }
}

__global__ __launch_bounds__(MT_BLOCK) void match_kernel(const u64 *__restrict__ skeys, u64 ns,
                                                          const u64 *__restrict__ rkeys,
                                                          const u64 *__restrict__ bounds,
                                                          long long *__restrict__ hit, u64 *__restrict__ nhits) {
    const u64 q0 = (u64)blockIdx.x * MT_TILE;
    const u64 q1 = q0 + MT_TILE < ns ? q0 + MT_TILE : ns;
    match_tile(skeys, q0, q1, rkeys, bounds[blockIdx.x], bounds[blockIdx.x + 1], hit, nhits);
}

// ---- the same join for search k-mers that arrive as several ascending runs back to back (what the owner of a key
// range receives from the other GPUs in the key-range exchange: one ascending slice per source, all inside the owner's
// range, so the concatenation is not ascending).  Tiles never straddle a run boundary; every tile carries its own
// search range and reference range, so one launch joins all runs.
#define MT_MAX_RUNS 64
struct MatchRuns {
    u32 n_runs;
    u32 tile0[MT_MAX_RUNS + 1];   // first tile of every run (runs without k-mers have no tile)
    u64 off[MT_MAX_RUNS + 1];     // first search k-mer of every run
};
struct MatchTile { u64 q0, q1, r_lo, r_hi; };

__global__ void match_runs_bounds_kernel(const u64 *__restrict__ skeys, MatchRuns runs, const u64 *__restrict__ rkeys,
                                         const u64 *__restrict__ rbucket, int key_len, u32 ntiles,
                                         MatchTile *__restrict__ tiles) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    u32 r = 0;
    while (r + 1 < runs.n_runs && t >= runs.tile0[r + 1]) r++;
    const u64 q0 = runs.off[r] + (u64)(t - runs.tile0[r]) * MT_TILE;
    const u64 q1 = q0 + MT_TILE < runs.off[r + 1] ? q0 + MT_TILE : runs.off[r + 1];
    const u64 first = skeys[q0], last = skeys[q1 - 1];
    u64 b = first >> (64 - 2 * key_len);
    u64 lo = rbucket[b], hi = rbucket[b + 1];
    while (lo < hi) {                         // first reference k-mer >= the tile's first search k-mer
        const u64 mid = (lo + hi) >> 1;
        if (rkeys[mid] < first) lo = mid + 1; else hi = mid;
    }
    const u64 r_lo = lo;
    b = last >> (64 - 2 * key_len);
    lo = rbucket[b]; hi = rbucket[b + 1];
    while (lo < hi) {                         // first reference k-mer > the tile's last search k-mer
        const u64 mid = (lo + hi) >> 1;
        if (rkeys[mid] <= last) lo = mid + 1; else hi = mid;
    }
    MatchTile mt;
    mt.q0 = q0; mt.q1 = q1; mt.r_lo = r_lo; mt.r_hi = lo < r_lo ? r_lo : lo;      // (a run that is not ascending finds nothing)
    tiles[t] = mt;
}

__global__ __launch_bounds__(MT_BLOCK) void match_runs_kernel(const u64 *__restrict__ skeys, const u64 *__restrict__ rkeys,
                                                               const MatchTile *__restrict__ tiles,
                                                               long long *__restrict__ hit) {
    const MatchTile mt = tiles[blockIdx.x];
    match_tile(skeys, mt.q0, mt.q1, rkeys, mt.r_lo, mt.r_hi, hit, nullptr);
}

// ---- search over k-mers of several lengths, no differences (Matcher on variable-length files with its default
// -k / -K: DB.getNearestKmers' non-quick branch, DB.java:83,139-184, over Root.addKmer / closestKmers,
// Root.java:29-170) ------------------------------------------------------------------------------------------------
// What the reference's radix trie hands back for a search k-mer t of T bases, in closed form (derivation and the CPU
// check against the literal restatement: DESIGN 6b, tools/exp/trie_closed_form.py).  The k-mers of one key enter the
// trie in (zero-padded value, length) order, which is prefix-first lexicographic order, so every split happens on the
// rightmost path and a data object is held by exactly one node of that path at a time.  The node that answers t is
// the one covering depth T on t's path; it belongs to x0 = the first stored k-mer that has t as a prefix (none: no
// line), and its data object has absorbed (Root.java:48-51,95-102; the object is shared between a split node and its
// tail, :79-80) every stored k-mer that shares x0's first m = max(shortest, L + 1) bases, L = the bases x0 shares
// with its predecessor in the key (0 for the key's first k-mer).  Those k-mers are the contiguous entries
// [x0, first entry whose first m bases are larger).  So: two binary searches and one look at the predecessor.
__global__ void trie_find_kernel(const u64 *__restrict__ skeys, const uint8_t *__restrict__ slens, int s_len_all, u64 ns,
                                 const u64 *__restrict__ rkeys, const uint8_t *__restrict__ rlens, int r_len_all,
                                 const u64 *__restrict__ rbucket, const u64 *__restrict__ rval_off, int key_len, int shortest,
                                 long long *__restrict__ first, u32 *__restrict__ found, u32 *__restrict__ nvals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const u64 t = skeys[i];
    const int T = slens ? (int)slens[i] : s_len_all;
    const u64 b = t >> (64 - 2 * key_len);
    const u64 lo = rbucket[b], hi = rbucket[b + 1];
    u64 l = lo, h = hi;
    while (l < h) {                                // first entry >= (t, T) in (value, length) order
        const u64 mid = (l + h) >> 1;
        const u64 km = rkeys[mid];
        const int lm = rlens ? (int)rlens[mid] : r_len_all;
        if (km < t || (km == t && lm < T)) l = mid + 1; else h = mid;
    }
    first[i] = -1; found[i] = 0; nvals[i] = 0;
    if (l == hi) return;
    const u64 x0 = rkeys[l];
    const int len0 = rlens ? (int)rlens[l] : r_len_all;
    if (len0 < T || ((x0 ^ t) >> (64 - 2 * T)) != 0) return;      // t is not a prefix of it: of no stored k-mer then
    int L = 0;
    if (l > lo) {
        const u64 p = rkeys[l - 1];
        const int lp = rlens ? (int)rlens[l - 1] : r_len_all;
        const u64 x = x0 ^ p;
        const int c = x ? (int)(__clzll((long long)x) >> 1) : 32;
        L = min(c, min(len0, lp));
    }
    const int m = max(shortest, L + 1);
    const u64 top = m >= 32 ? x0 : (x0 | (~0ull >> (2 * m)));      // the largest value with x0's first m bases
    u64 a = l + 1, e = hi;
    while (a < e) {
        const u64 mid = (a + e) >> 1;
        if (rkeys[mid] <= top) a = mid + 1; else e = mid;
    }
    first[i] = (long long)l;
    found[i] = 1;
    const u64 nv = rval_off[a] - rval_off[l];
    nvals[i] = nv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)nv;         // (the host refuses totals beyond 2^32-16)
}

// the answers' values as (row, taxon, count) records for the merge pipeline (reduce.h, mg_*), which adds the counts
// per taxon as TreeCountMap.addAll does.  row = rank of the search k-mer among the answered ones, in the records' key
// field; cap > 0: every stored count is first cut to what its file byte holds (CountDataType.java:40).
__global__ void trie_expand_kernel(const long long *__restrict__ first, const u32 *__restrict__ nvals, const u64 *__restrict__ row,
                                   const u64 *__restrict__ dst, u64 ns, const u64 *__restrict__ rval_off,
                                   const int32_t *__restrict__ rval_a, const int32_t *__restrict__ rval_b, int cap,
                                   u64 *__restrict__ out_keys, u64 *__restrict__ out_vals, long long *__restrict__ hit) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    if (first[i] < 0) { hit[i] = -1; return; }
    hit[i] = (long long)row[i];
    const u64 key = row[i] << 32;
    u64 o = dst[i];
    const u64 v0 = rval_off[first[i]];
    for (u64 v = v0; v < v0 + nvals[i]; v++, o++) {
        int32_t c = rval_b[v];
        if (cap > 0 && c > cap) c = cap;
        out_keys[o] = key;
        out_vals[o] = ((u64)((u32)rval_a[v] ^ 0x80000000u) << 32) | (u32)c;
    }
}

// ---- the same search over a SET of files that share keys ---------------------------------------------------------------
// The reference puts a key's k-mers into its trie file by file, each file in ascending order (DB.java:154-158).  With
// several files the insertion is no longer in the trie's preorder: a later file's k-mer can split an edge that an earlier
// file's k-mer made, after which BOTH pieces hold that k-mer's data object (n1.data = n.data, Root.java:79-80), and a
// k-mer that then walks through both pieces is merged into the object twice.  In closed form (derived in DESIGN 6b, held
// against the literal trie on the CPU by tests/test_trie_closed_form.py): the answer for t is the object of x = the
// earliest inserted stored k-mer that has t as a prefix -- the first file that holds one, its smallest there --, which
// ends as
//     D_x + sum over the boundary depths e in [max(L_x + 1, shortest), |x|] of
//           sum of D_z over the k-mers z inserted at or after the boundary's maker that share x's first e bases,
// L_x = the longest prefix x shares with anything inserted before it.  The boundary depths are |x| (made by x) and every
// lcp(x, y) < |x| of a k-mer y inserted after x (made by the first such y).  Per file the k-mers that share x's first e
// bases are one range of the sorted array, nested for growing e; a maker at depth e is an entry of the range of e that is
// not in the range of e + 1.  One thread per search k-mer walks files x depths with binary searches: a corner of the
// search (several -d files built from different inputs), not a hot path.
struct TrieFile {
    const u64 *keys;
    const uint8_t *lens;
    const u64 *bucket;
    const u64 *val_off;
    const int32_t *val_a, *val_b;
    int len_all, pad;
};

__device__ __forceinline__ int tf_len(const TrieFile &f, u64 i) { return f.lens ? (int)f.lens[i] : f.len_all; }
// leading bases two zero-padded k-mers of la / lb bases have in common
__device__ __forceinline__ int tf_lcp(u64 a, int la, u64 b, int lb) {
    const u64 x = a ^ b;
    const int c = x ? (int)(__clzll((long long)x) >> 1) : 32;
    return min(c, min(la, lb));
}
// first entry of [lo, hi) that is >= (v, len) in (value, length) order
__device__ __forceinline__ u64 tf_lower(const TrieFile &f, u64 lo, u64 hi, u64 v, int len) {
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        const u64 km = f.keys[mid];
        if (km < v || (km == v && tf_len(f, mid) < len)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the entries of [lo, hi) that have the first e bases of x: -> [a, b)
__device__ __forceinline__ void tf_prefix_range(const TrieFile &f, u64 lo, u64 hi, u64 x, int e, u64 &a, u64 &b) {
    const u64 mask = e >= 32 ? ~0ull : ~(~0ull >> (2 * e));
    const u64 v = x & mask, top = v | ~mask;
    a = tf_lower(f, lo, hi, v, e);
    u64 l = a, h = hi;
    while (l < h) {
        const u64 mid = (l + h) >> 1;
        if (f.keys[mid] <= top) l = mid + 1; else h = mid;
    }
    b = l;
}

template <bool EMIT>
__device__ __forceinline__ void tf_take(const TrieFile &f, u64 a, u64 b, u64 &count, u64 key, int cap, u64 *__restrict__ out_keys,
                                        u64 *__restrict__ out_vals, u64 &o) {
    if (a >= b) return;
    const u64 v0 = f.val_off[a], v1 = f.val_off[b];
    count += v1 - v0;
    if (EMIT) {
        for (u64 v = v0; v < v1; v++, o++) {
            int32_t c = f.val_b[v];
            if (cap > 0 && c > cap) c = cap;
            out_keys[o] = key;
            out_vals[o] = ((u64)((u32)f.val_a[v] ^ 0x80000000u) << 32) | (u32)c;
        }
    }
}

// EMIT = false: found[i], nvals[i] (values the answer takes in, with multiplicity).  EMIT = true: the values as (row, taxon,
// count) records at dst[i], hit[i] = row[i] or -1.
template <bool EMIT>
__global__ void trie_files_kernel(const u64 *__restrict__ skeys, const uint8_t *__restrict__ slens, int s_len_all, u64 ns,
                                  const TrieFile *__restrict__ files, int nf, int key_len, int shortest, u32 *__restrict__ found,
                                  u64 *__restrict__ nvals, const u64 *__restrict__ row, const u64 *__restrict__ dst, int cap,
                                  u64 *__restrict__ out_keys, u64 *__restrict__ out_vals, long long *__restrict__ hit) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const u64 t = skeys[i];
    const int T = slens ? (int)slens[i] : s_len_all;
    const u64 b = t >> (64 - 2 * key_len);
    // x: the first file that holds a k-mer with prefix t, its smallest there
    int f0 = -1;
    u64 l0 = 0;
    for (int f = 0; f < nf && f0 < 0; f++) {
        const TrieFile &F = files[f];
        const u64 lo = F.bucket[b], hi = F.bucket[b + 1];
        const u64 l = tf_lower(F, lo, hi, t, T);
        if (l < hi && tf_len(F, l) >= T && ((F.keys[l] ^ t) >> (64 - 2 * T)) == 0) { f0 = f; l0 = l; }
    }
    if (f0 < 0) {
        if (EMIT) hit[i] = -1;
        else { found[i] = 0; nvals[i] = 0; }
        return;
    }
    const u64 x = files[f0].keys[l0];
    const int X = tf_len(files[f0], l0);
    // L: the longest prefix x shares with anything inserted before it -- its predecessor in its own file, its neighbours in
    // the files before
    int L = 0;
    {
        const TrieFile &F = files[f0];
        if (l0 > F.bucket[b]) L = tf_lcp(x, X, F.keys[l0 - 1], tf_len(F, l0 - 1));
        for (int f = 0; f < f0; f++) {
            const TrieFile &G = files[f];
            const u64 lo = G.bucket[b], hi = G.bucket[b + 1];
            const u64 p = tf_lower(G, lo, hi, x, X);
            if (p > lo) L = max(L, tf_lcp(x, X, G.keys[p - 1], tf_len(G, p - 1)));
            if (p < hi) L = max(L, tf_lcp(x, X, G.keys[p], tf_len(G, p)));
        }
    }
    const int m = max(shortest, L + 1);
    u64 count = 0, o = EMIT ? dst[i] : 0;
    const u64 key = EMIT ? (row[i] << 32) : 0;
    // D_x itself
    tf_take<EMIT>(files[f0], l0, l0 + 1, count, key, cap, out_keys, out_vals, o);
    u64 made = 1ull << X;                      // bit e: some k-mer at or after x has made a boundary at depth e
    {
        // x's own file: the k-mers behind x.  Those that share x's first e bases end at b_e, nested for shrinking e; the
        // ones between b_(e+1) and b_e make (or find) the boundary at e and count once for it.
        const TrieFile &F = files[f0];
        const u64 hi = F.bucket[b + 1];
        u64 prev = l0 + 1;
        for (int e = X; e >= m; e--) {
            u64 a_, b_;
            tf_prefix_range(F, prev, hi, x, e, a_, b_);          // (everything in [l0 + 1, prev) shares more than e bases)
            if (b_ > prev) {
                tf_take<EMIT>(F, prev, b_, count, key, cap, out_keys, out_vals, o);
                made |= 1ull << e;
            }
            prev = max(prev, b_);
        }
    }
    for (int f = f0 + 1; f < nf; f++) {
        const TrieFile &F = files[f];
        u64 ra, rb;
        tf_prefix_range(F, F.bucket[b], F.bucket[b + 1], x, m, ra, rb);
        for (int e = m; e <= X && ra < rb; e++) {
            if (e == X) {                                            // x as a prefix (or x again): counts for x's own boundary
                tf_take<EMIT>(F, ra, rb, count, key, cap, out_keys, out_vals, o);
                break;
            }
            u64 na, nb;
            tf_prefix_range(F, ra, rb, x, e + 1, na, nb);            // the entries that share more than e bases
            if (made & (1ull << e)) tf_take<EMIT>(F, ra, rb, count, key, cap, out_keys, out_vals, o);
            else if (ra < na) {                                      // the range's first entry makes the boundary: all of it counts
                tf_take<EMIT>(F, ra, rb, count, key, cap, out_keys, out_vals, o);
                made |= 1ull << e;
            } else if (nb < rb) {                                    // made behind the deeper entries: only the tail counts
                tf_take<EMIT>(F, nb, rb, count, key, cap, out_keys, out_vals, o);
                made |= 1ull << e;
            }
            ra = na; rb = nb;
        }
    }
    if (EMIT) hit[i] = (long long)row[i];
    else { found[i] = 1; nvals[i] = count; }
}

// ---- TaxaCounts ----------------------------------------------------------------------------------------
// The distinct taxa of a database come from a presence bitmap over [min taxon, max taxon] (at most 2^32 bits =
// 512 MB; BASELINE's taxon ids span a few thousand): one pass marks, a popcount scan ranks the words, and the rank
// of a taxon -- its row in the ascending output, TaxaCounts.java:132-150 -- is prefix[word] + popcount of the lower
// bits: two cached loads instead of a binary search, and no sort of the value array.  Counting is one pass over
// the CSR, thread per k-mer: match[t]++ for every taxon of the k-mer, unique[t]++ if it is the only one
// (CountDataAll.add, TaxaCounts.java:386-395), into an LDS histogram (<= 2048 taxa: 256-thread blocks; <= 16384
// taxa: one 1024-thread block per CU with 128 KB of counters) or straight into global memory beyond that.
#define TX_LDS_SMALL 2048
#define TX_LDS_BIG 16384

__global__ void taxa_minmax_init_kernel(int32_t *mm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { mm[0] = 0x7FFFFFFF; mm[1] = (int32_t)0x80000000; }
}
__global__ __launch_bounds__(256) void taxa_minmax_kernel(const int32_t *__restrict__ val_a, u64 V, int32_t *__restrict__ mm) {
    int32_t lo = 0x7FFFFFFF, hi = (int32_t)0x80000000;
    // 16 bytes per lane over the aligned middle, the few values either side of it one by one
    const u64 head = std::min<u64>(V, (u64)((16 - ((uintptr_t)val_a & 15)) & 15) / 4);
    const u64 n4 = (V - head) / 4;
    const int4 *__restrict__ body = (const int4 *)(val_a + head);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (u64)gridDim.x * blockDim.x) {
        const int4 t = body[i];
        lo = min(min(lo, t.x), min(min(t.y, t.z), t.w));
        hi = max(max(hi, t.x), max(max(t.y, t.z), t.w));
    }
    if (blockIdx.x == 0) {
        for (u64 i = threadIdx.x; i < V - n4 * 4; i += blockDim.x) {
            const int32_t t = val_a[i < head ? i : n4 * 4 + i];
            lo = t < lo ? t : lo;
            hi = t > hi ? t : hi;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int32_t ol = __shfl_xor(lo, d, 64), oh = __shfl_xor(hi, d, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}
__global__ __launch_bounds__(256) void taxa_present_kernel(const int32_t *__restrict__ val_a, u64 V, int32_t tmin, u32 *__restrict__ bits) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (u64)gridDim.x * blockDim.x) {
        const u32 r = (u32)val_a[i] - (u32)tmin;
        const u32 bit = 1u << (r & 31);
        if (!(bits[r >> 5] & bit)) atomicOr(&bits[r >> 5], bit);     // (a stale read only costs a redundant atomic)
    }
}
__global__ void taxa_popc_kernel(const u32 *__restrict__ bits, u64 nwords, u32 *__restrict__ cnt) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) cnt[w] = __popc(bits[w]);
}
__global__ void taxa_list_kernel(const u32 *__restrict__ bits, const u32 *__restrict__ prefix, u64 nwords, int32_t tmin,
                                 int32_t *__restrict__ taxa) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    u32 b = bits[w], o = prefix[w];
    while (b) {
        const int j = __ffs(b) - 1;
        b &= b - 1;
        taxa[o++] = (int32_t)((u32)tmin + (u32)(w * 32 + j));
    }
}

// dense path (the span of the ids fits the LDS rows): presence comes from the counted rows themselves
__global__ void taxa_rows_present_kernel(const u64 *__restrict__ dm, u32 range, u32 *__restrict__ bits) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 32 >= range) return;
    u32 b = 0;
    for (u32 j = 0; j < 32 && w * 32 + j < range; j++) b |= (dm[w * 32 + j] != 0 ? 1u : 0u) << j;
    bits[w] = b;
}
__global__ void taxa_rows_gather_kernel(const u64 *__restrict__ dm, const u64 *__restrict__ du, u32 range,
                                        const u32 *__restrict__ bits, const u32 *__restrict__ prefix, u64 *__restrict__ match,
                                        u64 *__restrict__ unique) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= range || dm[i] == 0) return;
    const u32 idx = prefix[i >> 5] + __popc(bits[i >> 5] & ((1u << (i & 31)) - 1u));
    match[idx] = dm[i];
    unique[idx] = du[i];
}

// MODE 0: LDS histogram of TX_LDS_SMALL rows, 1: of TX_LDS_BIG rows (1024 threads, one block per CU), 2: global.
// DIRECT: the whole span [min taxon, max taxon] fits the rows, so a value's row is taxon - min (match / unique are then
// indexed by taxon - min too); otherwise the row is the taxon's rank (two dependent cached loads per value).
// A thread takes TX_UNROLL k-mers per round: their offsets first, then their first values, then the counting, so
// that several loads are in flight per lane (one block of 1024 per CU in MODE 1 leaves little else to hide latency).
#ifndef TX_UNROLL
#define TX_UNROLL 8
#endif
template <int MODE, bool DIRECT>
__global__ __launch_bounds__(MODE == 1 ? 1024 : 256) void taxa_count_kernel(const u64 *__restrict__ val_off,
                                                                            const int32_t *__restrict__ val_a, u64 U, u64 V,
                                                                            const u32 *__restrict__ bits,
                                                                            const u32 *__restrict__ prefix, int32_t tmin, u32 rows,
                                                                            u64 *__restrict__ match, u64 *__restrict__ unique) {
    constexpr u32 ROWS = MODE == 0 ? TX_LDS_SMALL : (MODE == 1 ? TX_LDS_BIG : 1);
    __shared__ u32 s_match[ROWS];
    __shared__ u32 s_unique[ROWS];
    if (MODE != 2) {
        for (u32 i = threadIdx.x; i < rows; i += blockDim.x) { s_match[i] = 0; s_unique[i] = 0; }
        __syncthreads();
    }
    auto add = [&](int32_t t, bool single) {
        const u32 r = (u32)t - (u32)tmin;
        const u32 idx = DIRECT ? r : prefix[r >> 5] + __popc(bits[r >> 5] & ((1u << (r & 31)) - 1u));
        if (MODE != 2) {
            atomicAdd(&s_match[idx], 1u);
            if (single) atomicAdd(&s_unique[idx], 1u);
        } else {
            atomicAdd(&match[idx], 1ull);
            if (single) atomicAdd(&unique[idx], 1ull);
        }
    };
    const u64 step = (u64)blockDim.x * TX_UNROLL;
    for (u64 base = (u64)blockIdx.x * step; base < U; base += (u64)gridDim.x * step) {
        // (indices clamped instead of guarded: guarded loads compile to a branch and a wait apiece)
        u64 a[TX_UNROLL], b[TX_UNROLL];
        int32_t t0[TX_UNROLL], t1[TX_UNROLL];
#pragma unroll
        for (int j = 0; j < TX_UNROLL; j++) {
            const u64 u = base + (u64)j * blockDim.x + threadIdx.x;
            const u64 uc = u < U ? u : U - 1;
            a[j] = val_off[uc];
            b[j] = val_off[uc + 1];
            if (u >= U) b[j] = a[j];
        }
#pragma unroll
        for (int j = 0; j < TX_UNROLL; j++) {
            t0[j] = val_a[a[j] < V ? a[j] : V - 1];
            t1[j] = val_a[a[j] + 1 < V ? a[j] + 1 : V - 1];
        }
#pragma unroll
        for (int j = 0; j < TX_UNROLL; j++) {
            const u64 n = b[j] - a[j];
            if (n >= 1) add(t0[j], n == 1);
            if (n >= 2) add(t1[j], false);
            for (u64 v = a[j] + 2; v < b[j]; v++) add(val_a[v], false);
        }
    }
    if (MODE != 2) {
        // (a block's share of the values is far below 2^32: the u32 rows cannot wrap)
        __syncthreads();
        for (u32 i = threadIdx.x; i < rows; i += blockDim.x) {
            if (!s_match[i]) continue;
            atomicAdd(&match[i], (u64)s_match[i]);
            if (s_unique[i]) atomicAdd(&unique[i], (u64)s_unique[i]);
        }
    }
}

// ---- inexact search, fixed k (Matcher -n 1 | 2) ----------------------------------------------------------------
// Replaces DB.processNearestCommonKey + Root.addKmers / closestKmers (Database/DB.java:139-184, Root.java:22-170) for
// k-mers of one length: with minK == maxK no node of the reference's radix trie carries merged data (Root.java:94-102
// clears it above depth minK), so closestKmers returns exactly the stored k-mers within Hamming distance maxDiff of
// the search k-mer, and KmerUtils.getCloseKeys (KmerUtils.java:18-47) only names the key buckets they can lie in.
// Here every neighbour of the search k-mer -- 3k of them at distance 1, 9 k (k-1) / 2 at distance 2 -- is looked up
// by binary search inside its own key bucket.  Thread (s, j) owns the variants of search k-mer s whose FIRST changed
// base is j (j == 0 also takes the unchanged k-mer).  `just`, the reverse-complement rule of ClosestInfoCollector.add
// and the KmerDiff text are the caller's (a handful of matches per k-mer).
__device__ __forceinline__ long long near_find(const u64 *__restrict__ rkeys, const u64 *__restrict__ rbucket, int key_len, u64 key) {
    const u64 b = key >> (64 - 2 * key_len);
    u64 lo = rbucket[b], hi = rbucket[b + 1];
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < rbucket[b + 1] && rkeys[lo] == key) ? (long long)lo : -1ll;
}

// matches noted down by the sizing pass, in no order: (search k-mer, reference k-mer, differences)
struct NearLog {
    u32 *s, *ref;
    uint8_t *dist;
    u64 cap;
    unsigned long long *n;      // NULL: no log
};

// the logged matches to their places: what the filling pass of near_kernel does, without the look-ups
__global__ void near_place_kernel(NearLog log, u64 n, u32 *__restrict__ cursor, const u64 *__restrict__ off,
                                  long long *__restrict__ out_ref, uint8_t *__restrict__ out_dist) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u32 s = log.s[e];
    const u64 at = off[s] + atomicAdd(&cursor[s], 1u);
    out_ref[at] = (long long)log.ref[e];
    out_dist[at] = log.dist[e];
}

// first index in [lo, hi) whose k-mer is >= key
__device__ __forceinline__ u64 near_lower(const u64 *__restrict__ rkeys, u64 lo, u64 hi, u64 key) {
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// [lo, hi) narrowed to the reference k-mers that share the first `plen` bases of x (1 <= plen <= 32)
__device__ __forceinline__ void near_narrow(const u64 *__restrict__ rkeys, u64 &lo, u64 &hi, u64 x, int plen) {
    const u64 mask = plen == 32 ? ~0ull : ~(~0ull >> (2 * plen));
    const u64 first = x & mask, last = first | ~mask;
    lo = near_lower(rkeys, lo, hi, first);
    hi = last == ~0ull ? hi : near_lower(rkeys, lo, hi, last + 1);
}

// FILL = false: cnt[s] += matches.  FILL = true: matches stored at off[s] + (cursor[s]++), unordered.
// Two differences: a variant whose first change is at j and second at j2 shares its first j2 bases with the
// one-difference variant x1, so the candidates are followed base by base as j2 moves on -- a range of reference
// k-mers that only shrinks -- and the walk stops as soon as nothing shares the prefix any more.  In a reference of
// random sequence that is after about log4(n) bases, so nearly all of the 9 (k - 1 - j) second changes are never
// looked up (148 G lookups on config 2 if they were).
template <bool FILL>
__global__ __launch_bounds__(256) void near_kernel(const u64 *__restrict__ skeys, u64 ns, int k, int max_diff,
                                                   const u64 *__restrict__ rkeys, const u64 *__restrict__ rbucket, int key_len,
                                                   u32 *__restrict__ cnt, const u64 *__restrict__ off,
                                                   long long *__restrict__ out_ref, uint8_t *__restrict__ out_dist,
                                                   NearLog log) {
    // the block's logged matches are gathered in LDS and take their places in the log with ONE global atomic
    // (one per match, all on the same counter, cost 0.3 s for 32 M matches)
    __shared__ u32 l_n, l_s[256], l_ref[256];
    __shared__ uint8_t l_d[256];
    __shared__ unsigned long long l_base;
    if (threadIdx.x == 0) l_n = 0;
    __syncthreads();
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 s = idx / (u32)k;
    const int j = (int)(idx % (u32)k);
    const bool live = s < ns;
    const u64 key = live ? skeys[s] : 0;
    u32 found = 0;
    auto emit = [&](u64 r, int d) {
        if (FILL) {
            const u64 at = off[s] + atomicAdd(&cnt[s], 1u);
            out_ref[at] = (long long)r;
            out_dist[at] = (uint8_t)d;
        } else {
            found++;
            if (log.n) {                      // remember the match, so that the filling call need not look it up again
                const u32 p = atomicAdd(&l_n, 1u);
                if (p < 256) { l_s[p] = (u32)s; l_ref[p] = (u32)r; l_d[p] = (uint8_t)d; }
                else {                        // (a block with more than 256 matches: straight to the log)
                    const u64 e = atomicAdd(log.n, 1ull);
                    if (e < log.cap) { log.s[e] = (u32)s; log.ref[e] = (u32)r; log.dist[e] = (uint8_t)d; }
                }
            }
        }
    };
    if (live && j == 0) {
        const long long r = near_find(rkeys, rbucket, key_len, key);
        if (r >= 0) emit((u64)r, 0);
    }
    const int sj = 62 - 2 * j;
    for (u64 b1 = 1; b1 < 4 && max_diff >= 1 && live; b1++) {
        const u64 x1 = key ^ (b1 << sj);
        if (max_diff < 2) {          // one difference: a plain look-up in x1's bucket
            const long long r = near_find(rkeys, rbucket, key_len, x1);
            if (r >= 0) emit((u64)r, 1);
            continue;
        }
        // reference k-mers that share x1's first j + 1 bases (at least key_len of them name the bucket)
        u64 lo, hi;
        if (j + 1 >= key_len) {
            const u64 b = x1 >> (64 - 2 * key_len);
            lo = rbucket[b]; hi = rbucket[b + 1];
            if (j + 1 > key_len) near_narrow(rkeys, lo, hi, x1, j + 1);
        } else {
            // the change lies inside the key: the k-mers with this prefix span whole buckets
            const int sh = 2 * (key_len - (j + 1));
            const u64 b = x1 >> (64 - 2 * key_len) >> sh << sh;
            lo = rbucket[b]; hi = rbucket[b + (1ull << sh)];
        }
        if (lo >= hi) continue;
        {
            const u64 at = near_lower(rkeys, lo, hi, x1);
            if (at < hi && rkeys[at] == x1) emit(at, 1);
        }
        for (int j2 = j + 1; j2 < k; j2++) {
            if (j2 > j + 1) {
                near_narrow(rkeys, lo, hi, x1, j2);           // now sharing x1's first j2 bases
                if (lo >= hi) break;
            }
            for (u64 b2 = 1; b2 < 4; b2++) {
                const u64 x2 = x1 ^ (b2 << (62 - 2 * j2));
                const u64 at = near_lower(rkeys, lo, hi, x2);
                if (at < hi && rkeys[at] == x2) emit(at, 2);
            }
        }
    }
    if (!FILL && found) atomicAdd(&cnt[s], found);
    if (FILL || !log.n) return;               // (block-uniform)
    __syncthreads();
    const u32 m = l_n < 256u ? l_n : 256u;
    if (threadIdx.x == 0 && m) l_base = atomicAdd(log.n, (unsigned long long)m);
    __syncthreads();
    if (threadIdx.x < m) {
        const u64 e = l_base + threadIdx.x;
        if (e < log.cap) { log.s[e] = l_s[threadIdx.x]; log.ref[e] = l_ref[threadIdx.x]; log.dist[e] = l_d[threadIdx.x]; }
    }
}

// ---- any number of differences (Matcher -n > 2): the trie walk of Root.closestKmers (Root.java:115-170) over the sorted
// array.  The stored k-mers that share a prefix are one contiguous range; a node of the walk = (range, depth, differences
// so far); its four children split the range by the next base (three binary searches; above the key length the key
// table gives the bounds), a child whose base differs from the search k-mer's costs one difference, empty ranges and
// children beyond the budget are dropped.  Two short cuts end most paths early: a range of ONE k-mer is compared as a
// whole (differing bases = popcount of the XOR folded to one bit per base), and with the budget spent the rest must
// match exactly -- one binary search.  Children are taken in base order, so the matches come out in ascending k-mer
// order, which is the order the trie hands them to ClosestInfoCollector.  One thread per search k-mer (the work per
// k-mer is what the reference's walk does; it grows quickly with the number of differences, as there).
#define NW_MAXK 32
template <bool FILL>
__global__ __launch_bounds__(128) void near_walk_kernel(const u64 *__restrict__ skeys, u64 ns, int k, int max_diff,
                                                        const u64 *__restrict__ rkeys, u64 nr, const u64 *__restrict__ rbucket,
                                                        int key_len, u32 *__restrict__ cnt, const u64 *__restrict__ off,
                                                        long long *__restrict__ out_ref, uint8_t *__restrict__ out_dist) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns || nr == 0) return;
    const u64 key = skeys[s];
    u64 found = 0;
    const u64 base_out = FILL ? off[s] : 0;
    auto emit = [&](u64 r, int d) {
        if (FILL) { out_ref[base_out + found] = (long long)r; out_dist[base_out + found] = (uint8_t)d; }
        found++;
    };
    auto bases_differing = [&](u64 a, u64 b) {
        const u64 x = a ^ b;
        return __popcll((x | (x >> 1)) & 0x5555555555555555ull);
    };
    // explicit stack: one frame per depth
    u64 f_lo[NW_MAXK + 1], f_hi[NW_MAXK + 1];      // range of the node at this depth
    u64 f_cur[NW_MAXK + 1];                        // start of the next child's range
    uint8_t f_next[NW_MAXK + 1], f_diff[NW_MAXK + 1];
    int d = 0;
    f_lo[0] = 0; f_hi[0] = nr; f_cur[0] = 0; f_next[0] = 0; f_diff[0] = 0;
    while (d >= 0) {
        const u64 lo = f_lo[d], hi = f_hi[d];
        const int used = f_diff[d];
        if (f_next[d] == 0) {
            // first visit of the node: the short cuts
            if (hi - lo == 1) {
                const int dist = bases_differing(rkeys[lo], key);
                if (dist <= max_diff) emit(lo, dist);
                d--;
                continue;
            }
            if (used == max_diff) {
                // (the node's k-mers share the first d bases of one variant of the search k-mer with `used` changes; the
                // rest must be the search k-mer's own: that variant = the node's prefix + the search k-mer's tail)
                const u64 pm = d == 0 ? 0ull : ~(~0ull >> (2 * d));
                const u64 want = (rkeys[lo] & pm) | (key & ~pm);
                const u64 at = near_lower(rkeys, lo, hi, want);
                if (at < hi && rkeys[at] == want) emit(at, used);
                d--;
                continue;
            }
            f_cur[d] = lo;
        }
        if (f_next[d] == 4) { d--; continue; }
        const int c = f_next[d]++;
        // the child's range: the node's k-mers whose base at position d is c -- they follow the earlier children's
        const u64 c_lo = f_cur[d];
        u64 c_hi;
        if (c == 3) c_hi = hi;
        else if (d < key_len) {
            // prefix of d + 1 bases -> first bucket of the next prefix
            const u64 pre = (rkeys[lo] >> (64 - 2 * (d + 1)) & ~3ull) | (u64)c;          // (d + 1 bases, right-aligned)
            c_hi = rbucket[(pre + 1) << (2 * (key_len - d - 1))];
            c_hi = c_hi < c_lo ? c_lo : (c_hi > hi ? hi : c_hi);
        } else {
            const u64 pm = d + 1 >= 32 ? ~0ull : ~(~0ull >> (2 * (d + 1)));
            const u64 pre = (rkeys[lo] & (d == 0 ? 0ull : ~(~0ull >> (2 * d)))) | ((u64)c << (62 - 2 * d));
            const u64 last = pre | ~pm;                                                  // the largest k-mer with that prefix
            c_hi = near_lower(rkeys, c_lo, hi, last);
            if (c_hi < hi && rkeys[c_hi] == last) c_hi++;
        }
        f_cur[d] = c_hi;
        if (c_hi == c_lo) continue;
        const int cost = (int)((key >> (62 - 2 * d)) & 3) != c;
        if (used + cost > max_diff) continue;
        if (d + 1 == k) {                          // a whole k-mer: k-mers are distinct, the range holds one
            emit(c_lo, used + cost);
            continue;
        }
        d++;
        f_lo[d] = c_lo; f_hi[d] = c_hi; f_next[d] = 0; f_diff[d] = (uint8_t)(used + cost);
    }
    if (!FILL) cnt[s] = (u32)(found > 0xFFFFFFFFull ? 0xFFFFFFFFull : found);
}

// every search k-mer's matches into ascending reference order (= ascending k-mer: the order the trie hands them out)
__global__ void near_sort_kernel(const u64 *__restrict__ off, u64 ns, long long *__restrict__ ref, uint8_t *__restrict__ dist) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const u64 a = off[s], e = off[s + 1];
    for (u64 i = a + 1; i < e; i++) {
        const long long r = ref[i];
        const uint8_t d = dist[i];
        u64 p = i;
        while (p > a && ref[p - 1] > r) { ref[p] = ref[p - 1]; dist[p] = dist[p - 1]; p--; }
        ref[p] = r;
        dist[p] = d;
    }
}
