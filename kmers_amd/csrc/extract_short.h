// extract_short.h -- the k-mers SHORTER than maxK of a variable-length build (MakeDatabase -k minK -K maxK with
// minK < maxK; the reference's own default is -k 24 -K 32, MakeDatabase.java:106-107).
//
// Replaces the `ending` branch of KmersFromFileSpliterator.tryAdvance (OtherFiles/KmersFromFile.java:90-111), which
// hands out suffixes of the sliding window, longest first, from min(pos, maxK - 1) down to minK, each one skipped on
// its own if it holds an invalid base, in two situations:
//   (C) the sequence line ends with pos >= minK (:128-133): the suffixes that end at the sequence's last base;
//   (B) a full window turned out invalid (:167-176; pos > minK always holds there): the suffixes that end at the
//       window's last base -- valid ones exist once the bad base lies at least minK bases back.
// When the LAST full window of a sequence is invalid both fire, one after the other, and every suffix comes out
// twice (two records: the counts double, a read k-mer holds the position twice).
// With W = maxK, m = minK and d(e) = number of valid bases in a row that end at base e (inside the sequence):
//   (B) at every end e of an invalid full window: lengths m .. d(e)                (d(e) < W there)
//   (C) at the last base of a sequence of n >= m bases: lengths m .. min(d, n, W - 1)
// Position of a suffix of length L that ends behind base e: (e + 1 - L) - sequence start, as a Java short.
//
// Thread p of a tile owns the full window [p, p + W) -- both situations for that window -- and, if a sequence of
// fewer than W bases starts at p, that sequence's (C).  Everything it needs lies in [p, p + W), i.e. in the tile's
// LDS image (extract.h).  Records leave in window order with their length beside them; kmg_build_finish sorts the
// records of every length on their own (a fixed-length build each) and interleaves the databases by rank.
#pragma once
#include "extract.h"

struct ShortArgs {
    ExtractArgs e;        // k = maxK; out_keys / out_vals = the short-record buffers; tile_count / tile_off as usual
    int min_k;
    uint8_t *out_len;     // length of every short record
    u64 out_base;         // records already in the buffers (earlier batches)
    u32 *jw_count;        // [tile][round][wave]: records of 64 consecutive positions (count pass -> write pass)
};

// What position p of a tile hands out, as a list of items (one per length, in the order the reference hands them out:
// situation B before C, longest first); item i is evaluated by lane i of the wave that owns p, so the eight lengths
// of a read's end cost one step of 8 busy lanes instead of eight steps of one (a lane walking its own lengths one after
// the other was two thirds of the first version's time).  At most 2 x 31 items.
struct ShortSite {
    u64 win;          // 32 bases from p, first in the top bits
    u64 seq_start;
    u32 seq;          // sequence of p
    int n;            // bases the suffixes are cut from (W, or the whole of a sequence shorter than W)
    int d, nB;        // first nB items: lengths d, d - 1, ...
    int top, nC;      // next nC items: lengths top, top - 1, ...
};
__device__ __forceinline__ ShortSite short_site(const ShortArgs &sa, const u32 *pk, const u32 *inv, const u32 *stb,
                                                const u32 *seqrel, u32 s_lo, u64 t0, u32 p) {
    const ExtractArgs &a = sa.e;
    const int W = a.k, m = sa.min_k;
    ShortSite x;
    x.nB = x.nC = 0; x.d = x.top = 0; x.n = W; x.win = 0; x.seq_start = 0; x.seq = 0;
    const u64 gp = t0 + p;
    if (gp >= a.nbases) return x;
    const u32 w = p >> 4, r = p & 15;
    const u64 hi = ((u64)pk[w] << 32) | pk[w + 1];
    x.win = (hi << (2 * r)) | ((u64)pk[w + 2] >> (32 - 2 * r));
    const u32 w2 = p >> 5, r2 = p & 31;
    const u64 bad = (((u64)inv[w2] << 32) | inv[w2 + 1]) << r2;             // bit 63 - i: base p + i is invalid
    const u64 st = (((u64)stb[w2] << 32) | stb[w2 + 1]) << r2;              // bit 63 - i: a sequence starts at p + i
    x.seq = s_lo + seqrel[p];
    x.seq_start = a.seq_off[x.seq];
    const bool inside = (W == 1 || ((st << 1) >> (65 - W)) == 0) && gp + (u64)W <= a.nbases;
    if (inside) {
        const u64 badW = bad >> (64 - W);
        x.d = badW ? __ffsll((long long)badW) - 1 : W;                       // valid bases in a row at the window's end
        if (badW && x.d >= m) x.nB = x.d - m + 1;                            // (B): lengths d .. m
        const bool last = gp + (u64)W == a.nbases || ((st >> (63 - W)) & 1);
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

#define ES_STAGE 1024                    // short records of a tile staged in LDS before they are stored
template <bool WRITE>
__global__ __launch_bounds__(EX_BLOCK) void extract_short_kernel(ShortArgs sa) {
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 inv[EX_TILE / 32 + 2];
    __shared__ u32 stb[EX_TILE / 32 + 2];
    __shared__ u32 seqrel[EX_TILE + EX_EXT];
    __shared__ u32 wave_cnt[EX_ROUNDS][EX_BLOCK / 64];     // records of (round, wave): 64 consecutive positions, in position order
    __shared__ u32 scan_smem[4];
    __shared__ u32 blk_total, blk_fwd;
    __shared__ u32 cand[EX_TILE / 32];
    __shared__ u64 st_key[WRITE ? ES_STAGE : 1], st_val[WRITE ? ES_STAGE : 1];
    __shared__ uint8_t st_len[WRITE ? ES_STAGE : 1];
    const ExtractArgs &a = sa.e;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 t0 = (u64)blockIdx.x * EX_TILE;
    const u32 s_lo = ex_load_tile(a, blockIdx.x, pk, inv, stb, seqrel, scan_smem);
    const int W = a.k;

    // Positions that can hand out a short k-mer at all, 32 per thread and bit-parallel (a superset; short_site decides):
    // a window with an invalid base, a window that ends where a sequence ends, a sequence start (a sequence of fewer
    // than W bases), the tile's first position.  On reads that is one position in a hundred.
    if (tid < EX_TILE / 32) {
        const u64 B = ((u64)inv[tid] << 32) | inv[tid + 1];
        const u64 S = ((u64)stb[tid] << 32) | stb[tid + 1];
        u32 c = (u32)(ep_or_span(B, W) >> 32) | (u32)(S >> 32) | (u32)((S << W) >> 32);       // (W <= 32: bit p + W lies inside S)
        const long long eoi = (long long)a.nbases - W - (long long)t0 - 32ll * tid;             // the window that ends the input
        if (eoi >= 0 && eoi < 32) c |= 0x80000000u >> (u32)eoi;
        if (tid == 0) c |= 0x80000000u;
        cand[tid] = c;
    }
    if (tid == 0) blk_fwd = 0;
    if (WRITE && tid < EX_ROUNDS * (EX_BLOCK / 64)) {
        // the count pass left the records of every (round, wave): their exclusive scan places them
        const u32 v = sa.jw_count[(u64)blockIdx.x * (EX_ROUNDS * (EX_BLOCK / 64)) + tid];
        u32 inc = v;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        wave_cnt[tid >> 2][tid & 3] = inc - v;
        if (tid == 63) blk_total = inc;
    }
    __syncthreads();
    const u64 tile_base = WRITE ? sa.out_base + a.tile_off[blockIdx.x] : 0;
    const bool staged = WRITE && blk_total <= ES_STAGE;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u32 wave_fwd = 0;
    for (int j = 0; j < EX_ROUNDS; j++) {
        const u32 p0 = j * EX_BLOCK + wave * 64;
        const u32 pl = p0 + lane;
        u64 todo = __ballot((cand[pl >> 5] >> (31 - (pl & 31))) & 1u);
        u32 running = 0;
        const u32 base = WRITE ? wave_cnt[j][wave] : 0;
        while (todo) {
            const u32 p = p0 + (u32)(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            const ShortSite x = short_site(sa, pk, inv, stb, seqrel, s_lo, t0, p);          // (the same for all lanes)
            const int T = x.nB + x.nC;
            if (T == 0) continue;
            // reads are numbered by the mapper, which runs when the k-mer is built: before the filters
            if (!WRITE && a.has_kmer && lane == 0) a.has_kmer[x.seq] = 1;
            // item `lane`: one suffix of length L of the n bases that start at p
            bool ef = false, er = false;
            u64 key = 0, rk = 0;
            int L = 0;
            if (lane < T) {
                L = lane < x.nB ? x.d - lane : x.top - (lane - x.nB);
                const u64 maskn = x.n == 32 ? ~0ull : ~(~0ull >> (2 * x.n));
                key = (x.win & maskn) << (2 * (x.n - L));
                bool keep = true;
                if (a.runofsame >= 0 && !runofsame_keep(key, L, a.runofsame)) keep = false;
                if (keep && a.dust >= 0 && !dust_keep(key, L, a.dust)) keep = false;
                if (keep) {
                    ef = true;
                    if (a.ranged) {
                        const u32 fb = (u32)(key >> a.range_shift);
                        ef = fb >= a.range_lo && fb < a.range_hi;
                    }
                    if (a.rc) {
                        const u64 maskL = ~(~0ull >> (2 * L));                          // L < 32 here
                        const u64 rev = revpairs64(key >> (64 - 2 * L));
                        er = !(((L & 1) == 0) && rev == key);                            // Kmer.isOwnRC: a palindrome of even length
                        rk = rev ^ (0xAAAAAAAAAAAAAAAAull & maskL);
                        if (er && a.ranged) {
                            const u32 rb = (u32)(rk >> a.range_shift);
                            er = rb >= a.range_lo && rb < a.range_hi;
                        }
                    }
                }
            }
            const u64 bf = __ballot(ef), br = __ballot(er);
            if (WRITE && (ef || er)) {
                u32 local = base + running + (u32)__popcll(bf & lt) + (u32)__popcll(br & lt);
                const u32 pos = (u32)(t0 + p + (u64)(x.n - L) - x.seq_start);
                const u64 val = a.kind == KMG_PAYLOAD_TAXON ? (u64)(u32)a.seq_id[x.seq]
                                                            : (((u64)(u32)a.read_index[x.seq] << 16) | (u64)(pos & 0xFFFFu));
                for (int half = 0; half < 2; half++) {
                    if (!(half ? er : ef)) continue;
                    const u64 kk = half ? rk : key;
                    if (staged) { st_key[local] = kk; st_val[local] = val; st_len[local] = (uint8_t)L; }
                    else {
                        const u64 slot = tile_base + local;
                        a.out_keys[slot] = kk;
                        sa.out_len[slot] = (uint8_t)L;
                        if (a.kind == KMG_PAYLOAD_TAXON) reinterpret_cast<u32 *>(a.out_vals)[slot] = (u32)val;
                        else reinterpret_cast<u64 *>(a.out_vals)[slot] = val;
                    }
                    local++;
                }
            }
            running += (u32)__popcll(bf) + (u32)__popcll(br);
            wave_fwd += (u32)__popcll(bf);
        }
        if (!WRITE && lane == 0) sa.jw_count[(u64)blockIdx.x * (EX_ROUNDS * (EX_BLOCK / 64)) + j * (EX_BLOCK / 64) + wave] = running;
        if (!WRITE && lane == 0) wave_cnt[j][wave] = running;
    }
    if (!WRITE) {
        if (lane == 0 && wave_fwd) atomicAdd(&blk_fwd, wave_fwd);
        __syncthreads();
        if (tid < 64) {
            u32 v = wave_cnt[tid >> 2][tid & 3];
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) v += __shfl_xor(v, dd, 64);
            if (tid == 0) {
                a.tile_count[blockIdx.x] = v;
                a.tile_fwd[blockIdx.x] = blk_fwd;
            }
        }
        return;
    }
    if (!staged) return;
    __syncthreads();
    for (u32 i = tid; i < blk_total; i += EX_BLOCK) {
        const u64 slot = tile_base + i;
        a.out_keys[slot] = st_key[i];
        sa.out_len[slot] = st_len[i];
        if (a.kind == KMG_PAYLOAD_TAXON) reinterpret_cast<u32 *>(a.out_vals)[slot] = (u32)st_val[i];
        else reinterpret_cast<u64 *>(a.out_vals)[slot] = st_val[i];
    }
}

// ---- per-length classes -----------------------------------------------------------------------------------------
// hist[L] += number of short records of length L
__global__ __launch_bounds__(256) void short_hist_kernel(const uint8_t *__restrict__ len, u64 n, u32 *__restrict__ hist) {
    __shared__ u32 h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&h[len[i] & 63], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// Up to four consecutive lengths L0 .. L1-1 are sorted as ONE class: a k-mer of at most 31 bases leaves the two lowest
// bits of its 64-bit form free, and with length - L0 there the plain 64-bit order is (zero-padded value, length) --
// Sequence.compareTo -- so the class sorts and reduces as a build of 32-mers; short_split_kernel takes the tag off again.
// flag[i] = record i belongs to the class (input of the exclusive scan that places it, in record order: stable)
__global__ void short_flag_kernel(const uint8_t *__restrict__ len, u64 n, int L0, int L1, u32 *__restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = len[i] >= L0 && len[i] < L1;
}
template <typename V>
__global__ void short_gather_kernel(const uint8_t *__restrict__ len, const u32 *__restrict__ dst, u64 n, int L0, int L1,
                                    const u64 *__restrict__ keys, const V *__restrict__ vals, u64 *__restrict__ out_keys,
                                    V *__restrict__ out_vals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && len[i] >= L0 && len[i] < L1) { out_keys[dst[i]] = keys[i] | (u64)(len[i] - L0); out_vals[dst[i]] = vals[i]; }
}
__global__ void short_split_kernel(u64 *__restrict__ keys, u64 n, int L0, uint8_t *__restrict__ lens) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    lens[i] = (uint8_t)(L0 + (int)(k & 3));
    keys[i] = k & ~3ull;
}

// ---- interleaving two databases by rank ---------------------------------------------------------------------------
// Sequence.compareTo (Kmers/Sequence.java:165-187) orders k-mers base by base, a proper prefix first: that is the
// zero-padded packed value first (the pad symbol A = 0 is the smallest base) and the length second.  X holds shorter
// k-mers than Y throughout, both ascending: X_i goes before Y_j iff key(X_i) <= key(Y_j).  Hence
//   place(X_i) = i + #{ j : key(Y_j) <  key(X_i) }          place(Y_j) = j + #{ i : key(X_i) <= key(Y_j) }.
// FROM_X: the entries of X are placed (lower bound in Y's keys), else those of Y (upper bound in X's keys).
// Tile-wise (the two sides are ascending, so the entries of the other side that a tile of ILP_TILE
// consecutive entries can land between are one contiguous piece -- il_bounds_kernel finds it, one thread per tile -- and
// the piece is searched in LDS instead of through the whole array; pieces beyond ILP_CAP entries are searched in place).
#define ILP_TILE 1024
#define ILP_CAP 3072
__global__ void il_bounds_kernel(const u64 *__restrict__ keys, u64 n, const u64 *__restrict__ other, u64 n_other,
                                 u64 *__restrict__ lo_out, u64 *__restrict__ hi_out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t * ILP_TILE >= n) return;
    const u64 first = keys[t * ILP_TILE], last = keys[min(n, (t + 1) * ILP_TILE) - 1];
    u64 lo = 0, hi = n_other;
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (other[mid] < first) lo = mid + 1; else hi = mid; }
    lo_out[t] = lo;
    hi = n_other;
    u64 l2 = lo;
    while (l2 < hi) { const u64 mid = (l2 + hi) >> 1; if (other[mid] <= last) l2 = mid + 1; else hi = mid; }
    hi_out[t] = l2;
}
template <bool FROM_X>
__global__ __launch_bounds__(256) void il_place_tiled_kernel(const u64 *__restrict__ keys, const uint8_t *__restrict__ lens, int len_all,
                                                             const u64 *__restrict__ val_off, u64 n, const u64 *__restrict__ other,
                                                             const u64 *__restrict__ t_lo, const u64 *__restrict__ t_hi,
                                                             u64 *__restrict__ out_keys, uint8_t *__restrict__ out_lens,
                                                             u32 *__restrict__ out_cnt, u64 *__restrict__ place) {
    __shared__ u64 s_other[ILP_CAP];
    const u64 olo = t_lo[blockIdx.x], ohi = t_hi[blockIdx.x];
    const u32 span = (u32)(ohi - olo <= ILP_CAP ? ohi - olo : 0);
    const bool in_lds = ohi - olo <= ILP_CAP;
    for (u32 i = threadIdx.x; i < span; i += 256) s_other[i] = other[olo + i];
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * ILP_TILE;
#pragma unroll
    for (int r = 0; r < ILP_TILE / 256; r++) {
        const u64 i = i0 + r * 256 + threadIdx.x;
        if (i >= n) break;
        const u64 key = keys[i];
        u64 lo;
        if (in_lds) {
            u32 a = 0, b = span;
            while (a < b) {
                const u32 mid = (a + b) >> 1;
                const u64 o = s_other[mid];
                if (FROM_X ? o < key : o <= key) a = mid + 1; else b = mid;
            }
            lo = olo + a;
        } else {
            u64 a = olo, b = ohi;
            while (a < b) {
                const u64 mid = (a + b) >> 1;
                const u64 o = other[mid];
                if (FROM_X ? o < key : o <= key) a = mid + 1; else b = mid;
            }
            lo = a;
        }
        const u64 at = i + lo;
        place[i] = at;
        out_keys[at] = key;
        out_lens[at] = lens ? lens[i] : (uint8_t)len_all;
        out_cnt[at] = (u32)(val_off[i + 1] - val_off[i]);
    }
}

// the values of every entry to their place in the interleaved value arrays.  A block takes 256 consecutive entries: their
// values are one contiguous piece of the source, walked by all lanes side by side (coalesced reads, and writes that
// only break where an entry of the other database goes in between); the entry a value belongs to is found by a binary
// search over the block's 257 offsets in LDS.
#define ILV_BLOCK 256
__global__ __launch_bounds__(ILV_BLOCK) void il_values_kernel(const u64 *__restrict__ val_off, const int32_t *__restrict__ val_a,
                                                               const int32_t *__restrict__ val_b, u64 n,
                                                               const u64 *__restrict__ place, const u64 *__restrict__ out_off,
                                                               int32_t *__restrict__ out_a, int32_t *__restrict__ out_b) {
    __shared__ u64 s_off[ILV_BLOCK + 1];
    __shared__ u64 s_dst[ILV_BLOCK];
    const u64 e0 = (u64)blockIdx.x * ILV_BLOCK;
    const u32 cnt = (u32)(n - e0 < ILV_BLOCK ? n - e0 : ILV_BLOCK);
    if (threadIdx.x <= cnt) s_off[threadIdx.x] = val_off[e0 + threadIdx.x];
    if (threadIdx.x == 0 && cnt == ILV_BLOCK) s_off[ILV_BLOCK] = val_off[e0 + ILV_BLOCK];
    if (threadIdx.x < cnt) s_dst[threadIdx.x] = out_off[place[e0 + threadIdx.x]];
    __syncthreads();
    const u64 v0 = s_off[0], v1 = s_off[cnt];
    for (u64 v = v0 + threadIdx.x; v < v1; v += ILV_BLOCK) {
        u32 lo = 0, hi = cnt;                      // last entry whose offset is <= v (entries without values are skipped over)
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (s_off[mid] <= v) lo = mid; else hi = mid;
        }
        const u64 o = s_dst[lo] + (v - s_off[lo]);
        out_a[o] = val_a[v];
        out_b[o] = val_b[v];
    }
}

// ---- KmerUtils.restrictedStream (Kmers/KmerUtils.java:59-74) ---------------------------------------------------------
// keep[i] = number of values of entry i if its k-mer has at least L bases, else 0
// (lens == null: every k-mer has len_all bases)
// (H: only k-mers of fewer than H bases -- L + 1 picks the k-mers of exactly L bases, 64 all from L on)
__global__ void rs_keep_kernel(const uint8_t *__restrict__ lens, int len_all, const u64 *__restrict__ val_off, u64 n, int L, int H,
                               u32 *__restrict__ keep) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len = lens ? (int)lens[i] : len_all;
    keep[i] = (len >= L && len < H) ? (u32)(val_off[i + 1] - val_off[i]) : 0u;
}
// the kept entries' values as (k-mer cut to L bases, payload) records for the merge pipeline (reduce.h, mg_*)
__global__ void rs_expand_kernel(const u64 *__restrict__ keys, const uint8_t *__restrict__ lens, int len_all,
                                 const u64 *__restrict__ val_off, const int32_t *__restrict__ val_a,
                                 const int32_t *__restrict__ val_b, u64 n, int L, int H, const u64 *__restrict__ dst, u64 base, int kind,
                                 u64 *__restrict__ out_keys, u64 *__restrict__ out_vals, bool cut = true) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len = lens ? (int)lens[i] : len_all;
    if (len < L || len >= H) return;
    // Kmer.limitTo (Kmer.java:277-289); cut = false: the keys stand in for k-mers that were cut already (wide.h's ranks)
    const u64 key = !cut || L >= 32 ? keys[i] : keys[i] & ~(~0ull >> (2 * L));
    u64 o = base + dst[i];
    for (u64 v = val_off[i]; v < val_off[i + 1]; v++, o++) {
        out_keys[o] = key;
        out_vals[o] = kind == KMG_PAYLOAD_TAXON ? (((u64)((u32)val_a[v] ^ 0x80000000u) << 32) | (u32)val_b[v])
                                                : (((u64)(u32)val_a[v] << 16) | (u64)((u32)val_b[v] & 0xFFFFu));
    }
}

// ---- a set of variable-length databases whose files share no key (what -S and -G write: sets cut at key bounds) laid
// end to end in key order.  sum_boff / sum_voff[b] = sum over the files of their own offsets at bucket b = the merged
// database's offsets (exclusive sums add); holders[b] = files with k-mers in bucket b.
__global__ void vm_sum_kernel(const u64 *__restrict__ bucket_off, const u64 *__restrict__ val_off, u32 nbuckets,
                              u64 *__restrict__ sum_boff, u64 *__restrict__ sum_voff, u32 *__restrict__ holders) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbuckets) return;
    const u64 o = bucket_off[b];
    sum_boff[b] += o;
    sum_voff[b] += val_off[o];
    if (b < nbuckets && bucket_off[b + 1] > o) holders[b]++;
}
__global__ void vm_clash_kernel(const u32 *__restrict__ holders, u32 nbuckets, u32 *__restrict__ clash) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbuckets && holders[b] > 1) *clash = 1;
}
__global__ void vm_place_kernel(const u64 *__restrict__ keys, const uint8_t *__restrict__ lens, int len_all,
                                const u64 *__restrict__ val_off, const int32_t *__restrict__ val_a,
                                const int32_t *__restrict__ val_b, const u64 *__restrict__ bucket_off, u64 n, int key_len,
                                const u64 *__restrict__ sum_boff, const u64 *__restrict__ sum_voff, u64 *__restrict__ out_keys,
                                uint8_t *__restrict__ out_lens, u64 *__restrict__ out_val_off, int32_t *__restrict__ out_a,
                                int32_t *__restrict__ out_b) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 b = keys[i] >> (64 - 2 * key_len);
    const u64 first = bucket_off[b];
    const u64 dst = sum_boff[b] + (i - first);
    u64 vdst = sum_voff[b] + (val_off[i] - val_off[first]);
    out_keys[dst] = keys[i];
    out_lens[dst] = lens ? lens[i] : (uint8_t)len_all;
    out_val_off[dst] = vdst;
    for (u64 v = val_off[i]; v < val_off[i + 1]; v++, vdst++) { out_a[vdst] = val_a[v]; out_b[vdst] = val_b[v]; }
}

// ---- KmerUtils.restrictedStream with minLength < maxLength: the entries of lo <= length < hi stay as they are
__global__ void vr_flag_kernel(const uint8_t *__restrict__ lens, int len_all, const u64 *__restrict__ val_off, u64 n, int lo, int hi,
                               u32 *__restrict__ keep, u32 *__restrict__ nvals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len = lens ? (int)lens[i] : len_all;
    const bool k = len >= lo && len < hi;
    keep[i] = k;
    nvals[i] = k ? (u32)(val_off[i + 1] - val_off[i]) : 0u;
}
__global__ void vr_gather_kernel(const u64 *__restrict__ keys, const uint8_t *__restrict__ lens, int len_all,
                                 const u64 *__restrict__ val_off, const int32_t *__restrict__ val_a, const int32_t *__restrict__ val_b,
                                 u64 n, const u32 *__restrict__ keep, const u64 *__restrict__ dstk, const u64 *__restrict__ dstv,
                                 u64 *__restrict__ out_keys, uint8_t *__restrict__ out_lens, u64 *__restrict__ out_val_off,
                                 int32_t *__restrict__ out_a, int32_t *__restrict__ out_b) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const u64 o = dstk[i];
    u64 v = dstv[i];
    out_keys[o] = keys[i];
    out_lens[o] = lens ? lens[i] : (uint8_t)len_all;
    out_val_off[o] = v;
    for (u64 s = val_off[i]; s < val_off[i + 1]; s++, v++) { out_a[v] = val_a[s]; out_b[v] = val_b[s]; }
}
