// radix.h -- stage 2a: stable LSD radix sort of (k-mer key, payload) records.
//
// Replaces the TreeMap<Kmer, A> insertion of MakeAndWriteKey.call (KmerFiles/FileCreator.java:273-303):
// sorting by the left-aligned 2-bit value is sorting by Sequence.compareTo (Kmers/Sequence.java:165-187).
// HBM-bound: each pass reads every record twice (histogram, scatter) and writes it once.
//
// One pass = three launches:
//   hist    : per tile, LDS histogram of the pass's digit -> table[digit][tile]
//   scan    : exclusive scan of the table (digit-major), scan.h
//   scatter : per tile, stable local ranks by wave-level digit matching (RADIX_BITS ballots per key,
//             no per-thread counters), records staged through LDS in digit order so that each digit's
//             run leaves the block as one contiguous coalesced store.
// The wave owns 64*RS_ITEMS consecutive records, so (wave, round, lane) order == input order and the
// sort is stable.  Passes whose digit is the same for every record are skipped by the host driver.
#pragma once
#include "common.h"
#include "scan.h"

#define RS_BITS 8
#define RS_DIGITS (1 << RS_BITS)

// Wave-level digit matching: for this lane's digit d, `below` = number of lower lanes (with `valid`) holding
// the same digit and `total` = number of such lanes in the whole wave.  Per digit bit: the bit sign-extended
// to a 0 / ~0 word B (one v_bfe_i32), its ballot, and "keep the lanes whose bit equals mine" folded into the
// running mask as m & ~(ballot ^ B) (one three-input bit-op per half).
template <int BITS = RS_BITS>
__device__ __forceinline__ void wave_match(u32 d, bool valid, u32 &below, u32 &total) {
    const u64 v = __ballot(valid);
    u32 mlo = (u32)v, mhi = (u32)(v >> 32);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        int B = ((int)(d << (31 - b))) >> 31;
        asm("" : "+v"(B));                     // keep B a value of its own: the ballot is one compare of it
        const u64 bal = __ballot(B != 0);
        mlo &= ~((u32)bal ^ (u32)B);
        mhi &= ~((u32)(bal >> 32) ^ (u32)B);
        asm("" : "+v"(mlo), "+v"(mhi));        // one three-input op per half and bit, not a re-associated chain
    }
    below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    total = __popc(mlo) + __popc(mhi);
}
#define RS_BLOCK 256
#define RS_WAVES (RS_BLOCK / 64)
#define RS_ITEMS 16
#define RS_TILE (RS_BLOCK * RS_ITEMS)  // 4096 records per tile

// Digit of a record for this pass.  src 0: from the k-mer key.  src 1: from the sign-flipped 32-bit
// payload, used to order taxa inside equal k-mers (TreeCountMap<Integer> key order).  src 2: from the raw
// 64-bit payload (read << 16 | pos), only needed when two sequences share a read index.
#define RS_SRC_KEY 0
#define RS_SRC_VAL32 1
#define RS_SRC_VAL64 2
template <typename V>
__device__ __forceinline__ u32 rs_digit(u64 key, V val, int shift, u32 mask, int src) {
    if (src == RS_SRC_VAL32) return (u32)((((u32)val) ^ 0x80000000u) >> shift) & mask;
    if (src == RS_SRC_VAL64) return (u32)(((u64)val) >> shift) & mask;
    return (u32)(key >> shift) & mask;
}

// One block histograms RSH_TILES consecutive tiles at a time, so that each digit's RSH_TILES counters -- adjacent in
// the digit-major table -- leave as one 64-byte store instead of sixteen stray 4-byte ones (measured with
// WRITE_SIZE: the one-tile-per-block version wrote 8x the table's size).  The tiles are dealt out evenly over a grid
// that is resident all at once (rs_hist_blocks): block b takes tiles [b * ntiles / nblocks, (b+1) * ntiles / nblocks).
// (Measured, tools/exp/hist_probe.hip: the kernel runs at the speed of its loads alone -- LDS adds on random digits
// retire ~6.6 per clock and CU, tools/exp/lds_hist_rate.hip, a tenth of the time -- and a grid of 1.8 waves of blocks
// cost 30 % over one of a single wave.)
#define RSH_TILES 16
static inline u32 rs_hist_blocks(u32 ntiles) {
    const u32 resident = 2048;                       // 256 CUs x 8 blocks of 256 threads
    return std::max<u32>(1, std::min<u32>(resident, ntiles));
}
template <typename V>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const u64 *__restrict__ keys, const V *__restrict__ vals,
                                                            u32 n, int shift, u32 mask, int src,
                                                            u32 *__restrict__ table, u32 ntiles) {
    __shared__ u32 h[RS_DIGITS][RSH_TILES + 1];
    const u32 t_lo = (u32)((u64)blockIdx.x * ntiles / gridDim.x), t_hi = (u32)((u64)(blockIdx.x + 1) * ntiles / gridDim.x);
    for (u32 tile0 = t_lo; tile0 < t_hi; tile0 += RSH_TILES) {
        const int tpb = (int)min((u32)RSH_TILES, t_hi - tile0);
        for (int t = 0; t < tpb; t++) h[threadIdx.x][t] = 0;
        __syncthreads();
        for (int t = 0; t < tpb; t++) {
            const u32 base = (tile0 + t) * RS_TILE;
            if (n - base >= RS_TILE) {
                // full tile: all of a thread's loads leave before the first counter is touched (a load under
                // `if (idx < n)` compiles to a branch with a wait behind it: one memory round trip per record)
                if (src) {
                    V v[RS_ITEMS];
#pragma unroll
                    for (int i = 0; i < RS_ITEMS; i++) v[i] = vals[base + i * RS_BLOCK + threadIdx.x];
#pragma unroll
                    for (int i = 0; i < RS_ITEMS; i++) atomicAdd(&h[rs_digit<V>(0, v[i], shift, mask, src)][t], 1u);
                } else {
                    u64 kk[RS_ITEMS];
#pragma unroll
                    for (int i = 0; i < RS_ITEMS; i++) kk[i] = keys[base + i * RS_BLOCK + threadIdx.x];
#pragma unroll
                    for (int i = 0; i < RS_ITEMS; i++) atomicAdd(&h[(u32)(kk[i] >> shift) & mask][t], 1u);
                }
                continue;
            }
            for (int i = 0; i < RS_ITEMS; i++) {
                u32 idx = base + i * RS_BLOCK + threadIdx.x;
                if (idx < n) {
                    u32 d = src ? rs_digit<V>(0, vals[idx], shift, mask, src) : rs_digit<V>(keys[idx], (V)0, shift, mask, 0);
                    atomicAdd(&h[d][t], 1u);
                }
            }
        }
        __syncthreads();
        u32 *row = table + (u64)threadIdx.x * ntiles + tile0;
        for (int t = 0; t < tpb; t++) row[t] = h[threadIdx.x][t];
        __syncthreads();
    }
}

// Scatter geometry: 512 threads x 8 records.  Eight waves per block and a 32-KB staging buffer that is
// used for the keys first and the payloads second keep three blocks (24 waves) resident per CU, so one
// block's loads/stores overlap another's ranking.
#define RSC_BLOCK 512
#define RSC_WAVES (RSC_BLOCK / 64)
#define RSC_ITEMS (RS_TILE / RSC_BLOCK)

// (Ranks from returning LDS adds, as the fused extractor takes them where the device allows -- extract_part.h, ATOM --,
// were measured here too: 0.649 -> 0.674 ms.  This kernel waits for its loads and stores, not for its VALU.)
template <typename V>
__global__ __launch_bounds__(RSC_BLOCK) void rs_scatter_kernel(const u64 *__restrict__ keys_in,
                                                                const V *__restrict__ vals_in,
                                                                u64 *__restrict__ keys_out, V *__restrict__ vals_out,
                                                                u32 n, int shift, u32 mask, int src,
                                                                const u32 *__restrict__ table, u32 ntiles) {
    __shared__ u32 wcnt[RSC_WAVES][RS_DIGITS];  // per-wave digit counts -> exclusive prefix over waves
    __shared__ u32 dstart[RS_DIGITS];           // tile-local start of each digit's run
    __shared__ u32 gdelta[RS_DIGITS];           // global position = local position + gdelta[digit] (mod 2^32)
    __shared__ u64 stage[RS_TILE];              // keys, then payloads, in digit order
    __shared__ uint8_t sdig[RS_TILE];           // digit of each staged slot
    __shared__ u32 scan_smem[RSC_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so block b and b+8 share an
    // L2.  Giving each XCD a contiguous range of tiles puts neighbouring tiles -- whose runs of one digit
    // are adjacent in the output -- behind the same L2, where their partial cache lines merge.
    const u32 per_xcd = (ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const u32 tile_base = tile * RS_TILE;
    const u32 tile_n = min((u32)RS_TILE, n - tile_base);

    if (tid < RS_DIGITS) {
#pragma unroll
        for (int w = 0; w < RSC_WAVES; w++) wcnt[w][tid] = 0;
    }
    u64 key[RSC_ITEMS];
    V val[RSC_ITEMS];
    u32 lpos[RSC_ITEMS];   // rank among the wave's records of the same digit, later the staged slot
    const u32 wave_base = wave * (64 * RSC_ITEMS);
#pragma unroll
    for (int j = 0; j < RSC_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const bool valid = li < tile_n;
        key[j] = valid ? keys_in[tile_base + li] : 0;
        val[j] = valid ? vals_in[tile_base + li] : (V)0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RSC_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const bool valid = li < tile_n;
        const u32 d = rs_digit<V>(key[j], val[j], shift, mask, src);
        u32 below, total;
        wave_match(d, valid, below, total);
        const u32 before = wcnt[wave][d];
        lpos[j] = before + below;
        // the lowest lane of each digit group publishes the group's size; a wave's LDS operations
        // execute in program order, so the read above precedes this write for every lane
        if (valid && below == 0) wcnt[wave][d] = before + total;
    }
    __syncthreads();

    // per digit (thread == digit): exclusive prefix over waves, tile-local run start, global delta
    {
        u32 run = 0;
        if (tid < RS_DIGITS) {
#pragma unroll
            for (int w = 0; w < RSC_WAVES; w++) {
                u32 c = wcnt[w][tid];
                wcnt[w][tid] = run;
                run += c;
            }
        }
        u32 total;
        u32 incl = block_scan_inclusive<u32, OpSum, RSC_WAVES>(run, OpSum(), scan_smem, &total);
        if (tid < RS_DIGITS) {
            u32 start = incl - run;
            dstart[tid] = start;
            // (loading this entry at kernel entry instead was measured slower: 1.06 vs 1.02 ms, it delays the record loads)
            gdelta[tid] = table[(u64)tid * ntiles + tile] - start;
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RSC_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        if (li < tile_n) {
            const u32 d = rs_digit<V>(key[j], val[j], shift, mask, src);
            lpos[j] += dstart[d] + wcnt[wave][d];
            stage[lpos[j]] = key[j];
            sdig[lpos[j]] = (uint8_t)d;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RSC_ITEMS; i++) {
        const u32 lp = i * RSC_BLOCK + tid;
        if (lp < tile_n) keys_out[lp + gdelta[sdig[lp]]] = stage[lp];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RSC_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        if (li < tile_n) reinterpret_cast<V *>(stage)[lpos[j]] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RSC_ITEMS; i++) {
        const u32 lp = i * RSC_BLOCK + tid;
        if (lp < tile_n) vals_out[lp + gdelta[sdig[lp]]] = reinterpret_cast<V *>(stage)[lp];
    }
}

struct SortPass {
    int shift;
    int bits;
    int src;
};

#define LS_BLOCK 512
#define LS_WAVES (LS_BLOCK / 64)
#ifndef LS_ITEMS
#define LS_ITEMS 5
#endif
#ifndef LS_MINWAVES
#define LS_MINWAVES 8
#endif
#define LS_MAX (LS_BLOCK * LS_ITEMS)   // 2560 records per segment at most (48 KB of LDS: 3 blocks per CU)

// number of prefix bits sorted by global passes for n records of `kbits` key bits
static inline int choose_pbits(u64 n, int kbits, int cap = 24) {
    // average segment <= 2300 records: LS_MAX = 2560 is then > 5 sigma away on uniform keys, and segments are as
    // full as the local sort's 512-record rounds allow
    int p = 0;
    while (p < kbits && p < cap && (n >> p) > (u64)(LS_BLOCK * LS_ITEMS) * 9 / 10) p++;
    return p;
}

// Where the segments of a sort lie.  A build of the whole key space: 2^pbits segments, number = the key's top pbits.
// A key-range build (kmg_build_set_key_range) holds its records in a slice of the prefixes: the plan counts the
// prefix bits for the slice's density (up to 30 of them), numbers only the slice's segments -- segment s holds prefix
// seg_base + s -- and spends no global pass on the `shared_top` leading bits that every key of the range has in common.
struct SegPlan {
    int pbits;
    int shared_top;
    u32 seg_base;
    u32 nseg;
};
static inline SegPlan whole_space_plan(int pbits) { return {pbits, 0, 0u, 1u << pbits}; }
// keys of [first_key, last_key] (both inclusive) only
static inline SegPlan key_slice_plan(int pbits, u64 first_key, u64 last_key) {
    if (pbits <= 0) return {0, 0, 0u, 1u};
    const u64 diff = first_key ^ last_key;
    const int common = diff ? __builtin_clzll(diff) : 64;
    const u32 lo = (u32)(first_key >> (64 - pbits)), hi = (u32)(last_key >> (64 - pbits));
    return {pbits, std::min(common, pbits), lo, hi - lo + 1};
}

// the global passes' digits, lowest first (none for the shared_top leading bits)
static inline std::vector<SortPass> prefix_digits(int pbits, int shared_top = 0) {
    std::vector<SortPass> prefix;
    const int eff = pbits - shared_top;
    if (eff > 0) {
        const int np = (eff + RS_BITS - 1) / RS_BITS;
        int shift = 64 - pbits;
        for (int i = 0; i < np; i++) {
            int bits = eff / np + (i < eff % np ? 1 : 0);
            prefix.push_back({shift, bits, RS_SRC_KEY});
            shift += bits;
        }
    }
    return prefix;
}

// Sorts n records.  Buffers ping-pong between (k0,v0) and (k1,v1); *result_in_1 tells where the
// sorted records ended up.  `table` must hold RS_DIGITS * ntiles u32 and `agg` the scan scratch.
template <typename V>
static int radix_sort_records(kmg_ctx *ctx, u64 *k0, V *v0, u64 *k1, V *v1, u32 n, const std::vector<SortPass> &passes,
                              u32 *table, u32 *agg, bool *result_in_1) {
    bool flip = false;
    if (n == 0) {
        *result_in_1 = false;
        return KMG_OK;
    }
    const u32 ntiles = div_up_u64(n, RS_TILE);
    for (const SortPass &p : passes) {
        const u64 *ki = flip ? k1 : k0;
        const V *vi = flip ? v1 : v0;
        u64 *ko = flip ? k0 : k1;
        V *vo = flip ? v0 : v1;
        const u32 mask = (1u << p.bits) - 1u;
        {
            ProfScope ps(ctx, sizeof(V) == 8 || ctx->sort_as_reads ? KP_SORT_HIST_V64 : KP_SORT_HIST_V32, (u64)n * (p.src ? sizeof(V) : 8));
            hipLaunchKernelGGL((rs_hist_kernel<V>), dim3(rs_hist_blocks(ntiles)), dim3(RS_BLOCK), 0, ctx->stream, ki, vi, n, p.shift,
                               mask, p.src, table, ntiles);
        }
        KMG_TRY((device_scan<u32, u32, OpSum, true>(ctx, table, table, (u64)RS_DIGITS * ntiles, OpSum(), 0u, agg,
                                                    nullptr, false, KP_SORT_SCAN)));
        {
            ProfScope ps(ctx, sizeof(V) == 8 || ctx->sort_as_reads ? KP_SORT_SCATTER_V64 : KP_SORT_SCATTER_V32, (u64)n * (8 + sizeof(V)) * 2);
            hipLaunchKernelGGL((rs_scatter_kernel<V>), dim3(((ntiles + 7) / 8) * 8), dim3(RSC_BLOCK), 0, ctx->stream, ki, vi, ko, vo, n,
                               p.shift, mask, p.src, table, ntiles);
        }
        KMG_HIP(hipGetLastError());
        flip = !flip;
    }
    *result_in_1 = flip;
    return KMG_OK;
}


// ------------------------------------------------------------------------------------------------------
// Hybrid sort: a few global LSD passes over the TOP prefix bits of the key, then every segment of equal
// prefix (a few thousand records) is sorted on all remaining digits inside LDS by one workgroup.
// For n records this moves each record through HBM ceil(P/8) + 1 times instead of once per key byte.
// Correctness: the prefix passes are a stable sort by prefix; a stable sort of each equal-prefix segment
// by (payload digits, low key bits) then yields the same order as the full LSD sort.
// ------------------------------------------------------------------------------------------------------
#define LS_MAX_PASSES 16

#define LS_TOP 2        // key digits sorted by counting passes before the comparison finish
#define LS_RUNCAP 384   // longest equal-prefix run the comparison finish accepts

struct LocalPasses {
    int n;
    int fast_from;   // > 0: run only passes [fast_from, n), then finish by comparison (see ls_sort_kernel)
    int cmp_shift;   // lowest key bit covered by those passes
    int pay_mode;    // payload ordering inside equal keys: 0 none (stable), 1 signed 32-bit, 2 raw 64-bit
    int shift[LS_MAX_PASSES];
    int bits[LS_MAX_PASSES];
    int src[LS_MAX_PASSES];
    u32 seg_base;    // prefix value of segment 0 (key-range builds, SegPlan)
};

// seg_off[s] = first record whose top `pbits` key bits are >= seg_base + s (s = 0 .. nseg), and the largest segment size
// (atomicMax into *max_seg).  keys must be sorted by that prefix.  A block finds 256 boundaries and so the sizes of 255
// segments: boundaries LSEG_PER * b .. LSEG_PER * b + 255.  `table` (may be null): the scanned digit-major offsets of
// the last global pass, whose digit was the key's top `top_bits` bits -- table[d * ntiles_tab] is where digit d's
// records start, so the search for a boundary runs inside its digit's range only.
#define LSEG_PER 255
__global__ __launch_bounds__(256) void ls_segments_kernel(const u64 *__restrict__ keys, u32 n, int pbits, u32 *__restrict__ seg_off,
                                                          u32 nseg, u32 *max_seg, const u32 *__restrict__ table, u32 ntiles_tab,
                                                          int top_bits, u32 seg_base) {
    __shared__ u32 bnd[256];
    const u32 s = blockIdx.x * LSEG_PER + threadIdx.x;
    u32 lo = 0, hi = n;
    if (s >= nseg) lo = n;
    else if (pbits > 0) {
        if (table) {
            const u32 d = s >> (pbits - top_bits);
            lo = table[(u64)d * ntiles_tab];
            hi = d + 1 < (1u << top_bits) ? table[(u64)(d + 1) * ntiles_tab] : n;
        }
        while (lo < hi) {
            u32 mid = lo + ((hi - lo) >> 1);
            if ((u32)(keys[mid] >> (64 - pbits)) < seg_base + s) lo = mid + 1; else hi = mid;
        }
    }
    bnd[threadIdx.x] = lo;
    if (s <= nseg && (threadIdx.x < LSEG_PER || s == nseg)) seg_off[s] = lo;
    __syncthreads();
    u32 m = (threadIdx.x < LSEG_PER && s < nseg) ? bnd[threadIdx.x + 1] - lo : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (u32)__shfl_down(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_seg, m);
}

template <typename V>
__device__ __forceinline__ u64 ls_pay_ord(V v, int mode) {
    return mode == 1 ? (u64)(((u32)v) ^ 0x80000000u) : (u64)v;
}
// full order of two records: key, then (if the payload is ordered) payload
template <typename V>
__device__ __forceinline__ bool ls_greater(u64 ka, V va, u64 kb, V vb, int mode) {
    if (ka != kb) return ka > kb;
    return mode != 0 && ls_pay_ord<V>(va, mode) > ls_pay_ord<V>(vb, mode);
}

// One workgroup sorts one equal-prefix segment (<= LS_MAX records) entirely in LDS: stable counting passes
// on the top LS_TOP digits of the remaining key bits, then a comparison finish inside the short runs that
// still agree on those digits.  A segment with an over-long run is sorted with the complete digit list
// (payload digits, then every key digit), which is correct from any arrangement the stable steps produce.
// (compiled once per number of rounds, as ls_sortidx_body: straight-line loops over a thread's records)
template <typename V, int ROUNDS, bool ATOM>
__device__ __forceinline__ void ls_sort_body(const u64 *__restrict__ keys_in, const V *__restrict__ vals_in,
                                             u64 *__restrict__ keys_out, V *__restrict__ vals_out, const LocalPasses &lp, u32 s0, u32 m,
                                             uint16_t (*wcnt)[RS_DIGITS], uint16_t (*wbase)[RS_DIGITS], u64 *skeys, V *svals) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (the segment's wave-rounds dealt out evenly over the waves, a wave without the last round skips it: as in
    // ls_sortidx_body)
    const u32 extra = ((m + 63) >> 6) - (ROUNDS - 1) * LS_WAVES;          // 1 .. LS_WAVES
    const bool last = (u32)wave < extra;
    const u32 wave_base = 64u * ((u32)wave * (ROUNDS - 1) + ((u32)wave < extra ? (u32)wave : extra));
#define LS_ACT(j) ((j) < ROUNDS - 1 || last)

    u64 key[ROUNDS];
    V val[ROUNDS];
    u32 rank[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const u32 lc = li < m ? li : m - 1;          // (clamped, not guarded: see ls_sortidx_body)
        key[j] = keys_in[s0 + lc];
        val[j] = vals_in[s0 + lc];
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
        if (wave_base + j * 64 + lane >= m || !LS_ACT(j)) { key[j] = 0; val[j] = (V)0; }
    }
    if (lp.n == 0) {
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const u32 li = wave_base + j * 64 + lane;
            if (LS_ACT(j) && li < m) { keys_out[s0 + li] = key[j]; vals_out[s0 + li] = val[j]; }
        }
        return;
    }
    int p = lp.fast_from, p_end = lp.n;
    bool fast = lp.fast_from > 0;
    *reinterpret_cast<uint2 *>(&wcnt[wave][4 * lane]) = make_uint2(0, 0);   // own row; only this wave writes it
    for (;;) {
        // ---- one stable counting pass: registers -> LDS in digit order.  Two barriers per pass: every wave
        // derives its own slot bases from all waves' counts (redundantly, 4 digits per lane) instead of
        // waiting for a block-wide scan, and clears its own counter row for the next pass.
        const int shift = lp.shift[p], src = lp.src[p], nbits = lp.bits[p];
        const u32 mask = (1u << nbits) - 1u;
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            if (LS_ACT(j)) {
                const u32 li = wave_base + j * 64 + lane;
                const bool valid = li < m;
                const u32 d = rs_digit<V>(key[j], val[j], shift, mask, src);
                if (ATOM) {                              // (as ls_word_pass: the rank a returning LDS add hands back)
                    const u32 old = valid ? atomicAdd(reinterpret_cast<u32 *>(&wcnt[wave][d & ~1u]), (d & 1u) ? 0x10000u : 1u) : 0u;
                    rank[j] = (d & 1u) ? (old >> 16) : (old & 0xFFFFu);
                } else {
                    u32 below, total;
                    wave_match(d, valid, below, total);
                    const u32 before = wcnt[wave][d];
                    rank[j] = before + below;
                    if (valid && below == 0) wcnt[wave][d] = before + total;
                }
            }
        }
        __syncthreads();
        {
            // lane owns digits 4*lane .. 4*lane+3
            u32 tot[4] = {0, 0, 0, 0}, mine[4] = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < LS_WAVES; w++) {
                const uint2 c = *reinterpret_cast<const uint2 *>(&wcnt[w][4 * lane]);   // 4 x u16
                if (w == wave) { mine[0] = tot[0]; mine[1] = tot[1]; mine[2] = tot[2]; mine[3] = tot[3]; }
                tot[0] += c.x & 0xFFFFu; tot[1] += c.x >> 16; tot[2] += c.y & 0xFFFFu; tot[3] += c.y >> 16;
            }
            const u32 s4 = tot[0] + tot[1] + tot[2] + tot[3];
            u32 inc = s4;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const u32 o = __shfl_up(inc, dd, 64);
                if (lane >= dd) inc += o;
            }
            u32 start = inc - s4;
            u32 b0 = start + mine[0]; start += tot[0];
            u32 b1 = start + mine[1]; start += tot[1];
            u32 b2 = start + mine[2]; start += tot[2];
            u32 b3 = start + mine[3];
            *reinterpret_cast<uint2 *>(&wbase[wave][4 * lane]) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
        }
        // same-wave LDS accesses execute in order: wbase is complete for this wave's reads below
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const u32 li = wave_base + j * 64 + lane;
            if (LS_ACT(j) && li < m) {
                const u32 d = rs_digit<V>(key[j], val[j], shift, mask, src);
                const u32 pos = wbase[wave][d] + rank[j];
                skeys[pos] = key[j];
                svals[pos] = val[j];
            }
        }
        __syncthreads();
        // every wave has read all counter rows (before the barrier above): clear the own row for the next pass
        *reinterpret_cast<uint2 *>(&wcnt[wave][4 * lane]) = make_uint2(0, 0);
        p++;
        if (p == p_end) {
            if (!fast) break;
            // ---- finish by comparison: the segment is now sorted on the bits >= lp.cmp_shift.  Inside every
            // run of records that agree on those bits, a record's final slot is
            //   run start + #(records of the run that are smaller) + #(equal records that precede it),
            // found by walking the run in LDS.  Runs are short (a k-mer's copies plus the few k-mers that
            // share its top digits), lanes of a wave sit in the same or neighbouring runs, and no further
            // counting pass over the low digits (or the payload digits) is needed.  Runs longer than
            // LS_RUNCAP would make this quadratic: such segments take the complete digit list instead.
            fast = false;
            bool toolong = false;
            u32 npos[ROUNDS];
#pragma unroll
            for (int j = 0; j < ROUNDS; j++) {
                const u32 li = wave_base + j * 64 + lane;
                npos[j] = li;
                if (LS_ACT(j) && li < m) {
                    const u64 x = skeys[li];
                    const V xv = svals[li];
                    const u64 px = x >> lp.cmp_shift;
                    u32 less = 0, eq = 0, t = li, steps = 0;
                    while (t > 0) {
                        const u64 y = skeys[t - 1];
                        if ((y >> lp.cmp_shift) != px) break;
                        t--;
                        const V yv = svals[t];
                        if (ls_greater<V>(x, xv, y, yv, lp.pay_mode)) less++;
                        else if (!ls_greater<V>(y, yv, x, xv, lp.pay_mode)) eq++;
                        if (++steps > LS_RUNCAP) { toolong = true; break; }
                    }
                    const u32 run_start = t;
                    t = li + 1;
                    while (t < m) {
                        const u64 y = skeys[t];
                        if ((y >> lp.cmp_shift) != px) break;
                        if (ls_greater<V>(x, xv, y, svals[t], lp.pay_mode)) less++;
                        t++;
                        if (++steps > LS_RUNCAP) { toolong = true; break; }
                    }
                    npos[j] = run_start + less + eq;
                    key[j] = x;
                    val[j] = xv;
                }
            }
            const bool fallback = __syncthreads_or(toolong);   // also: every LDS read above is done
            if (!fallback) {
#pragma unroll
                for (int j = 0; j < ROUNDS; j++) {
                    const u32 li = wave_base + j * 64 + lane;
                    if (LS_ACT(j) && li < m) { skeys[npos[j]] = key[j]; svals[npos[j]] = val[j]; }
                }
                __syncthreads();
                break;
            }
            p = 0;   // complete digit list, starting from the (stably permuted) current arrangement
        }
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const u32 li = wave_base + j * 64 + lane;
            if (LS_ACT(j) && li < m) { key[j] = skeys[li]; val[j] = svals[li]; }
        }
        // the next pass's first barrier orders these reads before its LDS writes
    }
    for (u32 i = tid; i < m; i += LS_BLOCK) {
        keys_out[s0 + i] = skeys[i];
        vals_out[s0 + i] = svals[i];
    }
}
#undef LS_ACT

template <typename V, bool ATOM>
__global__ __launch_bounds__(LS_BLOCK, 6) void ls_sort_kernel(const u64 *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                            u64 *__restrict__ keys_out, V *__restrict__ vals_out,
                                                            const u32 *__restrict__ seg_off, LocalPasses lp) {
    __shared__ __align__(8) uint16_t wcnt[LS_WAVES][RS_DIGITS];    // per-wave digit counts of the current pass (<= LS_MAX)
    __shared__ uint16_t wbase[LS_WAVES][RS_DIGITS];   // per-wave first LDS slot of each digit
    __shared__ u64 skeys[LS_MAX];
    __shared__ V svals[LS_MAX];

    const u32 s0 = seg_off[blockIdx.x];
    const u32 m = seg_off[blockIdx.x + 1] - s0;
    if (m == 0 || m > LS_MAX) return;   // oversize segments: the host falls back to full LSD passes
    switch ((m + LS_BLOCK - 1) / LS_BLOCK) {
        case 1: ls_sort_body<V, 1, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, s0, m, wcnt, wbase, skeys, svals); break;
        case 2: ls_sort_body<V, 2, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, s0, m, wcnt, wbase, skeys, svals); break;
        case 3: ls_sort_body<V, 3, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, s0, m, wcnt, wbase, skeys, svals); break;
        case 4: ls_sort_body<V, 4, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, s0, m, wcnt, wbase, skeys, svals); break;
        default: ls_sort_body<V, 5, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, s0, m, wcnt, wbase, skeys, svals); break;
    }
}

// Key-digit-only variant (no payload passes): what moves through LDS in every pass is one 64-bit word per
// record -- the key shifted left by pbits (its top pbits are the segment number, the same for the whole
// segment) with the record's load index in the freed low 12 bits -- instead of key + payload.  The payload
// is stored to LDS once and gathered through the index when the segment is written out.  Half the LDS
// traffic of ls_sort_kernel per pass and no payload registers.  Needs pbits + low_bit >= 12.
//
// HEADS (read DBs): the k-mer runs are found while the sorted segment is still in LDS, so no later pass has
// to read the sorted keys again.  The block writes, compacted at the start of its own slice of the INPUT
// buffers (whose records it holds in LDS by then), the segment's distinct k-mers (keys_in[s0 ..]) and the
// index of each one's first record (vals_in[s0 ..], in the payload's own width), their number to ucount[segment], and
// only the payloads to vals_out: the per-record sorted keys are never needed again.
// (the body is compiled once per number of rounds -- ROUNDS x 512 slots cover the segment -- so that every loop over a
// thread's records is straight-line code: the rounds' ballot chains and counter look-ups overlap instead of queueing
// behind one scalar branch per round)
// One stable counting pass of the one-word-per-record local sort: the wave's words (registers) -> dst[] in digit order,
// and, if `reload`, back into the registers in the new order.  Round j of this wave holds slots iff j < ALWAYS (known
// at compile time) or j < nact (wave-uniform).  Three barriers; wcnt rows must be zero on entry and are zero on exit.
template <int R, int ALWAYS, bool ATOM>
__device__ __forceinline__ void ls_word_pass(u64 (&w)[R], u32 nact, int shift, u32 mask, uint16_t (*wcnt)[RS_DIGITS],
                                             uint16_t (*wbase)[RS_DIGITS], u64 *dst, u32 wave_base, bool reload) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define LSW_ACT(j) ((j) < ALWAYS || (u32)(j) < nact)
    u32 rank[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        if (!LSW_ACT(j)) continue;
        const u32 d = (u32)(w[j] >> shift) & mask;
        if (ATOM) {
            // the rank = what a returning LDS add on the wave's counter hands back (two 16-bit counters per 32-bit word, a
            // wave holds at most LS_ITEMS * 64 records: the halves never carry) -- where the device's probe allows it, see
            // extract_part.h (ATOM).  ~45 VALU instructions of wave_match become one LDS instruction: 0.927 -> 0.905 ms.
            const u32 old = atomicAdd(reinterpret_cast<u32 *>(&wcnt[wave][d & ~1u]), (d & 1u) ? 0x10000u : 1u);
            rank[j] = (d & 1u) ? (old >> 16) : (old & 0xFFFFu);
        } else {
            u32 below, total;
            wave_match(d, true, below, total);
            const u32 before = wcnt[wave][d];
            rank[j] = before + below;
            wcnt[wave][d] = before + total;      // every lane of the digit stores the same sum: no branch around the store
        }
    }
    __syncthreads();
    if (wave == 0) {
        // One wave turns the 8 x 256 per-wave counts into per-wave slot bases; lane owns digits 4*lane .. +3.
        // Counts are 16-bit and every partial sum stays <= LS_MAX, so four of them are added at once as two
        // 32-bit words without carries between the halves.
        uint2 run = make_uint2(0, 0), pre[LS_WAVES];
#pragma unroll
        for (int wv = 0; wv < LS_WAVES; wv++) {
            const uint2 c = *reinterpret_cast<const uint2 *>(&wcnt[wv][4 * lane]);
            pre[wv] = run;
            run.x += c.x; run.y += c.y;
        }
        const u32 t0 = run.x & 0xFFFFu, t1 = run.x >> 16, t2 = run.y & 0xFFFFu, t3 = run.y >> 16;
        const u32 s4 = t0 + t1 + t2 + t3;
        u32 inc = s4;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        const u32 b0 = inc - s4, b1 = b0 + t0, b2 = b1 + t1, b3 = b2 + t2;
        const uint2 start = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
#pragma unroll
        for (int wv = 0; wv < LS_WAVES; wv++)
            *reinterpret_cast<uint2 *>(&wbase[wv][4 * lane]) = make_uint2(start.x + pre[wv].x, start.y + pre[wv].y);
    }
    __syncthreads();
    u32 pos[R];
#pragma unroll
    for (int j = 0; j < R; j++)
        if (LSW_ACT(j)) pos[j] = wbase[wave][(u32)(w[j] >> shift) & mask] + rank[j];
#pragma unroll
    for (int j = 0; j < R; j++)
        if (LSW_ACT(j)) dst[pos[j]] = w[j];
    __syncthreads();
    *reinterpret_cast<uint2 *>(&wcnt[wave][4 * lane]) = make_uint2(0, 0);
    if (reload) {
#pragma unroll
        for (int j = 0; j < R; j++)
            if (LSW_ACT(j)) w[j] = dst[wave_base + j * 64 + lane];
    }
#undef LSW_ACT
}

// Read sets are deep: at 30-fold coverage two thirds of a segment's records are copies of a k-mer that is already
// there.  FOLD (HEADS only): after the two lowest digit passes equal k-mers stand next to each other (records that
// agree on those 16 bits and differ above them are rare), so the runs of equal words are found there -- "groups" -- and
// only one word per group takes the remaining passes; the records then move to their group's final place in one step.
// A k-mer whose copies were split into several groups by a stranger with the same low digits is put together again by
// the stable group passes (the groups end up adjacent, in their order), and the distinct k-mers are told apart on the
// sorted groups.  Segments of mostly distinct k-mers (> 5/8 groups per record, long reads with many errors) carry on
// with the plain passes: what the check costs them is one neighbour look per record.
#ifndef LS_FOLD
#define LS_FOLD 1
#endif
#define LS_FOLD_AFTER 2          // passes before the groups are looked for
#define LS_GROUPS_MAX 1024       // two group words per thread; a 12-bit group number

template <typename V, bool HEADS, int ROUNDS, bool ATOM>
__device__ __forceinline__ u32 ls_sortidx_body(u64 *keys_in, V *vals_in, u64 *__restrict__ keys_out, V *__restrict__ vals_out,
                                                const LocalPasses &lp, int pbits, u32 *__restrict__ ucount, u32 s0, u32 m,
                                                uint16_t (*wcnt)[RS_DIGITS], uint16_t (*wbase)[RS_DIGITS], u64 *swords, V *svals) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int rounds = ROUNDS;
    // The segment's ceil(m / 64) wave-rounds dealt out evenly: every wave takes ROUNDS - 1 of them, the first `extra`
    // waves one more -- in slot order, so (wave, round, lane) still reads the records in their order.  A wave without
    // the last round skips it (wave-uniform): on average half a block-round of all-sentinel work less per pass.
    const u32 extra = ((m + 63) >> 6) - (ROUNDS - 1) * LS_WAVES;          // 1 .. LS_WAVES
    const bool last = (u32)wave < extra;
    const u32 nact = (u32)(ROUNDS - 1) + (last ? 1u : 0u);
    const u32 wave_base = 64u * ((u32)wave * (ROUNDS - 1) + ((u32)wave < extra ? (u32)wave : extra));
#define LS_ACT(j) ((j) < ROUNDS - 1 || last)

    // Slots from m to the end of m's wave-round hold all-ones sentinels: they sort behind every record (stably, they
    // are loaded last), so no step below needs a per-lane validity branch.
    // (loads at clamped indices, all issued before the first is used: a load under `if (valid)` compiles to a branch
    // with a wait behind it, one memory round trip per round)
    u64 sw[ROUNDS];
    V vraw[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const u32 lc = li < m ? li : m - 1;
        sw[j] = keys_in[s0 + lc];
        vraw[j] = vals_in[s0 + lc];
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const bool valid = li < m && LS_ACT(j);
        sw[j] = valid ? ((sw[j] << pbits) | li) : ~0ull;
        if (valid) svals[li] = vraw[j];
    }
    *reinterpret_cast<uint2 *>(&wcnt[wave][4 * lane]) = make_uint2(0, 0);
    const u64 seg_prefix = pbits > 0 ? ((u64)(blockIdx.x + lp.seg_base) << (64 - pbits)) : 0ull;
    for (int p = 0; p < lp.n; p++) {
        ls_word_pass<ROUNDS, ROUNDS - 1, ATOM>(sw, nact, lp.shift[p] + pbits, (1u << lp.bits[p]) - 1u, wcnt, wbase, swords, wave_base, p + 1 < lp.n);
        if (LS_FOLD && HEADS && p + 1 == LS_FOLD_AFTER && lp.n >= LS_FOLD_AFTER + 2) {
            // ---- groups: runs of equal words (the index bits aside) in the order of the two lowest digits -------------
            u32 *wtot = reinterpret_cast<u32 *>(&wbase[0][0]);   // heads per wave-round, in slot order (wbase is free between passes)
            u32 gbelow[ROUNDS];
            u32 ghead = 0;
            // opaque copy of the slot base: keeps this block's address arithmetic out of the pass loop's registers
            // (hoisted, it sat in a dozen registers across every pass: 61 -> 80 and a wave per SIMD less)
            u32 wb = wave_base;
            asm volatile("" : "+v"(wb));
#pragma unroll
            for (int j = 0; j < ROUNDS; j++) {
                if (!LS_ACT(j)) continue;
                const u32 q = wb + j * 64 + lane;
                const u64 prev = swords[q ? q - 1 : 0];
                const bool head = q < m && (q == 0 || ((prev ^ sw[j]) >> 12) != 0);
                const u64 hb = __ballot(head);
                if (lane == 0) wtot[q >> 6] = __popcll(hb);
                gbelow[j] = __builtin_amdgcn_mbcnt_hi((u32)(hb >> 32), __builtin_amdgcn_mbcnt_lo((u32)hb, 0u));
                ghead |= (u32)head << j;
            }
            __syncthreads();
            const u32 nwr = (m + 63) >> 6;                        // wave-rounds of the segment (<= 40)
            const u32 c = (u32)lane < nwr ? wtot[lane] : 0u;
            u32 inc = c;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const u32 o = __shfl_up(inc, dd, 64);
                if (lane >= dd) inc += o;
            }
            const u32 G = __shfl(inc, 63, 64);
            if (G < LS_GROUPS_MAX && 8 * G <= 5 * m) {       // (G + 1 group starts must end below the records' numbers)
                u64 *gbuf = swords;                                              // [G] group words: key bits | group number
                uint16_t *gstart = reinterpret_cast<uint16_t *>(swords + LS_GROUPS_MAX);   // [G + 1] first slot of every group
                // (the barrier above ordered every wave's reads of swords before these stores into it; wtot is not
                // written over before the group passes' first barrier)
                // per record, in the order of the slots: group number << 12 | load index -- in the upper half of swords, which
                // neither the group words (<= 8 KB + 2 KB of group starts) nor 32-bit payloads in final order (<= 10 KB) reach
                u32 *recs = reinterpret_cast<u32 *>(swords + LS_MAX / 2);
                const u32 exc = inc - c;
#pragma unroll
                for (int j = 0; j < ROUNDS; j++) {
                    if (!LS_ACT(j)) continue;
                    const u32 q = wb + j * 64 + lane;
                    const u32 base = __shfl(exc, q >> 6, 64);
                    const bool head = (ghead >> j) & 1;
                    const u32 gid = base + gbelow[j] - (head ? 0u : 1u);
                    recs[q] = (gid << 12) | ((u32)sw[j] & 0xFFFu);
                    if (head) {
                        gbuf[gid] = (sw[j] & ~0xFFFull) | gid;
                        gstart[gid] = (uint16_t)q;
                    }
                }
                if (tid == 0) gstart[G] = (uint16_t)m;
                return G;          // the rest happens once for all round counts: ls_fold_tail
            }
            // (wtot lies in wbase, which the next pass writes behind its first barrier: every wave has read it by then)
        }
    }
    if (!HEADS) {
        for (u32 i = tid; i < m; i += LS_BLOCK) {
            const u64 w = swords[i];
            keys_out[s0 + i] = seg_prefix | ((w & ~0xFFFull) >> pbits);
            vals_out[s0 + i] = svals[(u32)w & 0xFFFu];
        }
        return 0;
    }
    // the last pass's closing barrier ordered every load of this block's input slice before these stores
    u32 *wtot = reinterpret_cast<u32 *>(&wbase[0][0]);   // head counts per (round, wave), in record order
    const u64 below_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 hw[ROUNDS];
    u32 hbelow[ROUNDS];
    u32 hmask = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        if (r < rounds) {
            const u32 i = r * LS_BLOCK + tid;
            const bool inb = i < m;
            const u64 w = inb ? swords[i] : 0ull;
            const bool head = inb && (i == 0 || ((swords[i - 1] ^ w) >> 12) != 0);
            const u64 hb = __ballot(head);
            if (lane == 0) wtot[r * LS_WAVES + wave] = __popcll(hb);
            if (inb) vals_out[s0 + i] = svals[(u32)w & 0xFFFu];
            hw[r] = w;
            hbelow[r] = __popcll(hb & below_mask);
            hmask |= (u32)head << r;
        }
    }
    __syncthreads();
    // every wave scans the (at most 40) counts itself: lane = round * 8 + wave
    const u32 c = lane < rounds * LS_WAVES ? wtot[lane] : 0u;
    u32 inc = c;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const u32 o = __shfl_up(inc, dd, 64);
        if (lane >= dd) inc += o;
    }
    const u32 exc = inc - c;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        if (r < rounds) {
            const u32 base = __shfl(exc, r * LS_WAVES + wave, 64);
            if ((hmask >> r) & 1) {
                const u32 slot = s0 + base + hbelow[r];
                keys_in[slot] = seg_prefix | ((hw[r] & ~0xFFFull) >> pbits);
                vals_in[slot] = (V)(s0 + r * LS_BLOCK + tid);
            }
        }
    }
    const u32 nheads = __shfl(inc, 63, 64);
    if (tid == 0) ucount[blockIdx.x] = nheads;
    return 0;
}

// The second half of the fold (see ls_sortidx_body): G group words in swords[0 .. G), the groups' first slots behind
// them, every record's (group number << 12 | load index) in the upper half of swords.  Compiled once, behind the
// switch over the round counts: nothing of the body's registers is live here.
template <typename V, bool ATOM>
__device__ __forceinline__ void ls_fold_tail(u64 *keys_in, V *vals_in, V *__restrict__ vals_out, const LocalPasses &lp, int pbits,
                                             u32 *__restrict__ ucount, u32 s0, u32 m, u32 G, uint16_t (*wcnt)[RS_DIGITS],
                                             uint16_t (*wbase)[RS_DIGITS], u64 *swords, V *svals) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64 *gbuf = swords;
    uint16_t *gstart = reinterpret_cast<uint16_t *>(swords + LS_GROUPS_MAX);
    uint16_t *delta = reinterpret_cast<uint16_t *>(&wbase[0][0]);    // [G] final place - place now (mod 2^16)
    u32 *recs = reinterpret_cast<u32 *>(swords + LS_MAX / 2);
    const u64 seg_prefix = pbits > 0 ? ((u64)(blockIdx.x + lp.seg_base) << (64 - pbits)) : 0ull;
    __syncthreads();
    // ---- the remaining passes over one word per group --------------------------------------------------------
    const u32 gwr = (G + 63) >> 6;                                   // 1 .. 16 wave-rounds
    const u32 grounds = (gwr + LS_WAVES - 1) / LS_WAVES;             // 1 or 2 per wave
    const u32 gextra = gwr - (grounds - 1) * LS_WAVES;
    const u32 gnact = (grounds - 1) + ((u32)wave < gextra ? 1u : 0u);
    const u32 gwave_base = 64u * ((u32)wave * (grounds - 1) + ((u32)wave < gextra ? (u32)wave : gextra));
    u64 gw[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const u32 i = gwave_base + j * 64 + lane;
        gw[j] = ((u32)j < gnact && i < G) ? gbuf[i] : ~0ull;
    }
    // The group words are all different (their group numbers are), and their order as plain 64-bit numbers -- key bits,
    // then group number -- is the order wanted: groups of one k-mer keep the order they were found in.  So no stable
    // pass per digit is needed here: ONE counting pass on the top eight bits leaves buckets of a few words each, and
    // a word's place inside its bucket is the number of smaller words there (a short walk through LDS).  Five barriers
    // instead of twelve for the four passes this replaces: with a few hundred words per segment a pass is all latency.
    ls_word_pass<2, 0, ATOM>(gw, gnact, 56, 0xFFu, wcnt, wbase, gbuf, gwave_base, true);
    u32 npos[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        npos[j] = 0;
        if ((u32)j < gnact) {
            const u64 w = gw[j];
            const u32 d = (u32)(w >> 56);
            const u32 bs = wbase[0][d];                                   // wave 0's slot base of a digit = the bucket's start
            const u32 be = d == 255u ? gwr * 64u : (u32)wbase[0][d + 1];  // (the last bucket ends with the sentinels)
            u32 smaller = 0;
            for (u32 t = bs; t < be; t += 4) {
                // (four reads in flight, at clamped slots; a word is not smaller than itself)
                const u64 x0 = gbuf[t], x1 = gbuf[min(t + 1, be - 1)], x2 = gbuf[min(t + 2, be - 1)], x3 = gbuf[min(t + 3, be - 1)];
                smaller += (x0 < w) + (t + 1 < be && x1 < w) + (t + 2 < be && x2 < w) + (t + 3 < be && x3 < w);
            }
            npos[j] = bs + smaller;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++)
        if ((u32)j < gnact && gw[j] != ~0ull) gbuf[npos[j]] = gw[j];
    __syncthreads();
    // ---- sorted groups -> their first slots (scan of the sizes) and the distinct k-mers -------------------------
    u32 *wsum = reinterpret_cast<u32 *>(&wbase[LS_WAVES / 2][0]);    // (wbase: delta in its first half, this behind it)
    u64 x[2];
    u32 v[2], gg[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const u32 i = 2 * tid + e;
        v[e] = 0; gg[e] = 0; x[e] = 0;
        if (i < G) {
            x[e] = gbuf[i];
            gg[e] = (u32)x[e] & 0xFFFu;
            const u64 pv = i ? gbuf[i - 1] : ~x[e];
            const u32 distinct = ((pv ^ x[e]) >> 12) != 0;
            v[e] = ((u32)gstart[gg[e] + 1] - (u32)gstart[gg[e]]) | (distinct << 16);
        }
    }
    const u32 both = v[0] + v[1];
    u32 winc = both;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const u32 o = __shfl_up(winc, dd, 64);
        if (lane >= dd) winc += o;
    }
    if (lane == 63) wsum[wave] = winc;
    __syncthreads();
    u32 wpre = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < LS_WAVES; wv++) {
        const u32 t = wsum[wv];
        if (wv < wave) wpre += t;
        tot += t;
    }
    u32 ex = wpre + winc - both;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const u32 i = 2 * tid + e;
        if (i < G) {
            const u32 first = ex & 0xFFFFu;
            delta[gg[e]] = (uint16_t)(first - (u32)gstart[gg[e]]);
            if (v[e] >> 16) {
                const u32 slot = s0 + (ex >> 16);
                keys_in[slot] = seg_prefix | ((x[e] & ~0xFFFull) >> pbits);
                vals_in[slot] = (V)(s0 + first);
            }
        }
        ex += v[e];
    }
    if (tid == 0) ucount[blockIdx.x] = tot >> 16;
    __syncthreads();
    // ---- every record to its group's final place ------------------------------------------------------------------
    if (sizeof(V) == 4) {
        // through LDS (the lower half of swords: the group words are done with), then out in order
        V *fvals = reinterpret_cast<V *>(swords);
        for (u32 q = tid; q < m; q += LS_BLOCK) {
            const u32 rec = recs[q];
            fvals[(q + delta[rec >> 12]) & 0xFFFFu] = svals[rec & 0xFFFu];
        }
        __syncthreads();
        for (u32 i = tid; i < m; i += LS_BLOCK) vals_out[s0 + i] = fvals[i];
    } else {
        // (64-bit payloads in final order would not fit beside the records' group numbers: straight to memory,
        // a group's records side by side)
        for (u32 q = tid; q < m; q += LS_BLOCK) {
            const u32 rec = recs[q];
            vals_out[s0 + ((q + delta[rec >> 12]) & 0xFFFFu)] = svals[rec & 0xFFFu];
        }
    }
}

#undef LS_ACT

// Oversize segments: their records to and from one compact array, a chunk of a segment per workgroup.
#define BIG_CHUNK 4096u
struct BigChunk { u32 slice, compact, count; };      // first record in the segmented array, in the compact array, records
template <typename V, bool GATHER>
__global__ __launch_bounds__(256) void big_copy_kernel(const BigChunk *__restrict__ chunks, const u64 *__restrict__ ks,
                                                       const V *__restrict__ vs, u64 *__restrict__ kd, V *__restrict__ vd) {
    const BigChunk c = chunks[blockIdx.x];
    const u32 from = GATHER ? c.slice : c.compact, to = GATHER ? c.compact : c.slice;
    for (u32 i = threadIdx.x; i < c.count; i += 256) {
        kd[to + i] = ks[from + i];
        vd[to + i] = vs[from + i];
    }
}

template <typename V, bool HEADS, bool ATOM>
__global__ __launch_bounds__(LS_BLOCK, LS_MINWAVES) void ls_sortidx_kernel(u64 *keys_in, V *vals_in, u64 *__restrict__ keys_out,
                                                               V *__restrict__ vals_out, const u32 *__restrict__ seg_off,
                                                               LocalPasses lp, int pbits, u32 *__restrict__ ucount) {
    __shared__ __align__(8) uint16_t wcnt[LS_WAVES][RS_DIGITS];
    __shared__ uint16_t wbase[LS_WAVES][RS_DIGITS];
    __shared__ u64 swords[LS_MAX];
    __shared__ V svals[LS_MAX];

    const u32 s0 = seg_off[blockIdx.x];
    const u32 m = seg_off[blockIdx.x + 1] - s0;
    if (m == 0 || m > LS_MAX) {
        if (HEADS && threadIdx.x == 0) ucount[blockIdx.x] = 0;   // (the host never asks for HEADS with oversize segments)
        return;
    }
    u32 G = 0;
    switch ((m + LS_BLOCK - 1) / LS_BLOCK) {
        case 1: G = ls_sortidx_body<V, HEADS, 1, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, pbits, ucount, s0, m, wcnt, wbase, swords, svals); break;
        case 2: G = ls_sortidx_body<V, HEADS, 2, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, pbits, ucount, s0, m, wcnt, wbase, swords, svals); break;
        case 3: G = ls_sortidx_body<V, HEADS, 3, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, pbits, ucount, s0, m, wcnt, wbase, swords, svals); break;
        case 4: G = ls_sortidx_body<V, HEADS, 4, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, pbits, ucount, s0, m, wcnt, wbase, swords, svals); break;
        default: G = ls_sortidx_body<V, HEADS, 5, ATOM>(keys_in, vals_in, keys_out, vals_out, lp, pbits, ucount, s0, m, wcnt, wbase, swords, svals); break;
    }
    if (LS_FOLD && HEADS && G) ls_fold_tail<V, ATOM>(keys_in, vals_in, vals_out, lp, pbits, ucount, s0, m, G, wcnt, wbase, swords, svals);
}

// Gathers what ls_sortidx_kernel<.., true> left per segment into the DB arrays: one block per segment.
template <typename V>
__global__ __launch_bounds__(256) void ls_heads_gather_kernel(const u64 *__restrict__ head_keys, const V *__restrict__ head_first,
                                                             const u32 *__restrict__ seg_off, const u32 *__restrict__ ucount,
                                                             const u64 *__restrict__ ubase, u64 *__restrict__ ukeys,
                                                             u64 *__restrict__ val_off) {
    const u32 seg = blockIdx.x;
    const u32 u = ucount[seg], s0 = seg_off[seg];
    const u64 b = ubase[seg];
    for (u32 i = threadIdx.x; i < u; i += 256) {
        ukeys[b + i] = head_keys[s0 + i];
        val_off[b + i] = head_first[s0 + i];
    }
}

// Optional by-product of the local sort for read DBs (see ls_sortidx_kernel): the caller provides ucount
// (2^pbits u32); when `done` comes back true the sorted KEYS were not written at all -- the distinct k-mers
// and their first-record indices sit per segment in the sort's other buffer pair instead.
struct SortHeads {
    u32 *ucount;
    bool done;
    const u64 *head_keys;
    const void *head_first;   // u64 or u32 per distinct k-mer, as the payload type of the sort
    int first_bytes;
    const u32 *seg_off;
    u32 nseg;
};

// Sorts n records by (key bits [low_bit, 64), then payload passes).  `payload_passes` are the least
// significant digits (taxa inside a k-mer / shared read indices).  Result location in *result_in_1.
// scratch: table (RS_DIGITS * ntiles u32), agg (scan), seg (2^pbits + 2 u32 + 1 for the max).
template <typename V>
static int hybrid_sort_records(kmg_ctx *ctx, u64 *k0, V *v0, u64 *k1, V *v1, u32 n, int low_bit,
                               const std::vector<SortPass> &payload_passes, u32 *table, u32 *agg, u32 *seg,
                               const SegPlan &plan, bool skip_first, bool *result_in_1, SortHeads *heads = nullptr,
                               bool allow_cmp_finish = true) {
    *result_in_1 = false;
    if (heads) heads->done = false;
    if (n == 0) return KMG_OK;
    const int pbits = plan.pbits;
    // global LSD passes over the top pbits of the key
    const std::vector<SortPass> prefix = prefix_digits(pbits, plan.shared_top);
    // `skip_first`: the records already arrive partitioned by the first prefix digit (extract_part.h)
    std::vector<SortPass> todo(prefix.begin() + (skip_first && !prefix.empty() ? 1 : 0), prefix.end());
    bool in1 = false;
    KMG_TRY(radix_sort_records<V>(ctx, k0, v0, k1, v1, n, todo, table, agg, &in1));
    u64 *ka = in1 ? k1 : k0, *kb = in1 ? k0 : k1;
    V *va = in1 ? v1 : v0, *vb = in1 ? v0 : v1;

    const u32 nseg = plan.nseg;
    u32 *max_seg = seg + nseg + 1;
    KMG_HIP(hipMemsetAsync(max_seg, 0, 4, ctx->stream));
    {
        ProfScope ps(ctx, KP_SORT_SEGMENTS, 0);
        // (the last global pass of this call leaves its scanned table behind: digit = the key's top bits)
        // (only when that digit is the key's very top: not behind leading bits a key-range build left out of its passes)
        const bool narrow = !todo.empty() && plan.shared_top == 0 && plan.seg_base == 0;
        hipLaunchKernelGGL(ls_segments_kernel, dim3(nseg / LSEG_PER + 1), dim3(256), 0, ctx->stream, ka, n, pbits,
                           seg, nseg, max_seg, narrow ? (const u32 *)table : (const u32 *)nullptr, div_up_u64(n, RS_TILE),
                           narrow ? todo.back().bits : 0, plan.seg_base);
    }
    u32 h_max = 0;
    KMG_HIP(read_back(ctx, &h_max, max_seg, 4));

    // remaining digits, least significant first: payload, then key bits [low_bit, 64 - pbits)
    std::vector<SortPass> rest = payload_passes;
    {
        const int lbits = 64 - pbits - low_bit;
        if (lbits > 0) {
            const int np = (lbits + RS_BITS - 1) / RS_BITS;
            int shift = low_bit;
            for (int i = 0; i < np; i++) {
                int bits = lbits / np + (i < lbits % np ? 1 : 0);
                rest.push_back({shift, bits, RS_SRC_KEY});
                shift += bits;
            }
        }
    }
    // Segments that do not fit one workgroup's LDS (skew: poly-A, low-complexity repeats pile into a few
    // prefixes -- with sequencing errors into a few hundred).  Their records are gathered into one array, that array is
    // sorted by plain LSD passes on the whole key (stable, so it keeps the segments in their order and each segment's
    // records end up where its slice expects them), and copied back into the slices: the passes run over the few per
    // cent of the records that need them, in a handful of launches however many such segments there are.  If they hold
    // most of the data, the whole array takes the plain LSD path below.
    std::vector<std::pair<u32, u32>> big;   // (first record, size)
    bool local_ok = rest.size() <= LS_MAX_PASSES;
    if (local_ok && h_max > LS_MAX) {
        std::vector<u32> h_seg((size_t)nseg + 1);
        KMG_HIP(hipMemcpyAsync(h_seg.data(), seg, ((size_t)nseg + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        KMG_HIP(hipStreamSynchronize(ctx->stream));
        u64 total_big = 0;
        for (u32 sgi = 0; sgi < nseg; sgi++) {
            const u32 sz = h_seg[sgi + 1] - h_seg[sgi];
            if (sz > LS_MAX) { big.push_back({h_seg[sgi], sz}); total_big += sz; }
        }
        if (total_big > (u64)n / 2) local_ok = false;
    }
    if (local_ok) {
        LocalPasses lp;
        lp.seg_base = plan.seg_base;
        lp.n = (int)rest.size();
        // Reference DBs: counting passes on the top LS_TOP key digits only, the rest is finished by comparison
        // in LDS (measured 0.32 -> 0.14 ms on config 2, and the taxon digits need no pass at all).
        // Read DBs always run the full digit list: deep read sets hold ~9 copies of every k-mer plus its
        // sequencing-error variants, so every run is ~10-20 records long and walking it is a chain of
        // dependent LDS reads (measured 2.6 -> 5.9 ms).
        lp.fast_from = (sizeof(V) == 4 && !heads && allow_cmp_finish && lp.n > LS_TOP && (lp.n - (int)payload_passes.size()) >= LS_TOP) ? lp.n - LS_TOP : 0;
        lp.cmp_shift = lp.fast_from > 0 ? rest[lp.fast_from].shift : 0;
        lp.pay_mode = payload_passes.empty() ? 0 : (payload_passes[0].src == RS_SRC_VAL32 ? 1 : 2);
        for (int i = 0; i < lp.n; i++) { lp.shift[i] = rest[i].shift; lp.bits[i] = rest[i].bits; lp.src[i] = rest[i].src; }
        if (lp.n == 0) { *result_in_1 = in1; return KMG_OK; }
        {
            const bool idx_kernel = lp.fast_from == 0 && payload_passes.empty() && pbits + low_bit >= 12 && pbits < 64;
            const bool with_heads = idx_kernel && heads && big.empty();
            // algorithmic bytes: records in, records out -- with heads only the payloads go out (+ 16 B per distinct
            // k-mer, added by the caller once their number is known)
            ProfScope ps(ctx, sizeof(V) == 8 || ctx->sort_as_reads ? KP_SORT_LOCAL_V64 : KP_SORT_LOCAL_V32,
                         (u64)n * (8 + sizeof(V)) + (u64)n * (with_heads ? sizeof(V) : 8 + sizeof(V)));
            // key digits only and room for a 12-bit index under the key: the one-word-per-record kernel
            if (idx_kernel) {
                if (with_heads) {
                    if (ctx->lds_rank_ok)
                        hipLaunchKernelGGL((ls_sortidx_kernel<V, true, true>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg,
                                           lp, pbits, heads->ucount);
                    else
                        hipLaunchKernelGGL((ls_sortidx_kernel<V, true, false>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg,
                                           lp, pbits, heads->ucount);
                    heads->done = true;
                    heads->head_keys = ka;
                    heads->head_first = va;
                    heads->first_bytes = (int)sizeof(V);
                    heads->seg_off = seg;
                    heads->nseg = nseg;
                } else
                    if (ctx->lds_rank_ok)
                        hipLaunchKernelGGL((ls_sortidx_kernel<V, false, true>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg,
                                           lp, pbits, (u32 *)nullptr);
                    else
                        hipLaunchKernelGGL((ls_sortidx_kernel<V, false, false>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg,
                                           lp, pbits, (u32 *)nullptr);
            } else
                if (ctx->lds_rank_ok) hipLaunchKernelGGL((ls_sort_kernel<V, true>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg, lp);
                else hipLaunchKernelGGL((ls_sort_kernel<V, false>), dim3(nseg), dim3(LS_BLOCK), 0, ctx->stream, ka, va, kb, vb, seg, lp);
            KMG_HIP(hipGetLastError());
        }
        if (!big.empty()) {
            std::vector<BigChunk> chunks;
            u64 total_big = 0;
            for (const auto &bs : big) {
                for (u32 c = 0; c < bs.second; c += BIG_CHUNK)
                    chunks.push_back({bs.first + c, (u32)total_big + c, std::min<u32>(BIG_CHUNK, bs.second - c)});
                total_big += bs.second;
            }
            void *d_chunks = nullptr, *ck0 = nullptr, *ck1 = nullptr, *cv0 = nullptr, *cv1 = nullptr;
            int rc = ctx->pool.alloc(&d_chunks, chunks.size() * sizeof(BigChunk));
            if (rc == KMG_OK) rc = ctx->pool.alloc(&ck0, total_big * 8);
            if (rc == KMG_OK) rc = ctx->pool.alloc(&ck1, total_big * 8);
            if (rc == KMG_OK) rc = ctx->pool.alloc(&cv0, total_big * sizeof(V));
            if (rc == KMG_OK) rc = ctx->pool.alloc(&cv1, total_big * sizeof(V));
            auto give_back = [&]() {      // (behind the launches: the pool's blocks are reused on this stream only)
                for (void *q : {d_chunks, ck0, ck1, cv0, cv1}) if (q) ctx->pool.release(q);
            };
            if (rc != KMG_OK) { give_back(); return rc; }
            hipError_t he = hipMemcpyAsync(d_chunks, chunks.data(), chunks.size() * sizeof(BigChunk), hipMemcpyHostToDevice, ctx->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);      // `chunks` is pageable host memory
            if (he != hipSuccess) { give_back(); return KMG_EDEVICE; }
            {
                ProfScope ps(ctx, KP_SORT_SEGMENTS, 0);
                hipLaunchKernelGGL((big_copy_kernel<V, true>), dim3((u32)chunks.size()), dim3(256), 0, ctx->stream,
                                   (const BigChunk *)d_chunks, (const u64 *)ka, (const V *)va, (u64 *)ck0, (V *)cv0);
            }
            std::vector<SortPass> all = rest;
            for (const SortPass &p : prefix) all.push_back(p);
            bool in_c1 = false;
            rc = radix_sort_records<V>(ctx, (u64 *)ck0, (V *)cv0, (u64 *)ck1, (V *)cv1, (u32)total_big, all, table, agg, &in_c1);
            if (rc != KMG_OK) { give_back(); return rc; }
            {
                ProfScope ps(ctx, KP_SORT_SEGMENTS, 0);
                hipLaunchKernelGGL((big_copy_kernel<V, false>), dim3((u32)chunks.size()), dim3(256), 0, ctx->stream,
                                   (const BigChunk *)d_chunks, (const u64 *)(in_c1 ? ck1 : ck0), (const V *)(in_c1 ? cv1 : cv0), kb, vb);
            }
            give_back();
            KMG_HIP(hipGetLastError());
        }
        *result_in_1 = !in1;
        return KMG_OK;
    }
    // Skewed input (a segment does not fit one workgroup's LDS): finish with plain LSD passes.  Starting
    // from the prefix-sorted array is fine -- those passes were stable -- but the prefix digits must be
    // redone after the low digits.
    std::vector<SortPass> all = rest;
    for (const SortPass &p : prefix) all.push_back(p);
    bool in1b = false;
    KMG_TRY(radix_sort_records<V>(ctx, ka, va, kb, vb, n, all, table, agg, &in1b));
    *result_in_1 = in1b ? !in1 : in1;
    return KMG_OK;
}
