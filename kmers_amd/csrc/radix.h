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

template <typename V>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const u64 *__restrict__ keys, const V *__restrict__ vals,
                                                            u32 n, int shift, u32 mask, int src,
                                                            u32 *__restrict__ table, u32 ntiles) {
    __shared__ u32 h[RS_DIGITS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        u32 idx = base + i * RS_BLOCK + threadIdx.x;
        if (idx < n) {
            u32 d = src ? rs_digit<V>(0, vals[idx], shift, mask, src) : rs_digit<V>(keys[idx], (V)0, shift, mask, 0);
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    table[(u64)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <typename V>
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const u64 *__restrict__ keys_in,
                                                               const V *__restrict__ vals_in,
                                                               u64 *__restrict__ keys_out, V *__restrict__ vals_out,
                                                               u32 n, int shift, u32 mask, int src,
                                                               const u32 *__restrict__ table, u32 ntiles) {
    __shared__ u32 wcnt[RS_WAVES][RS_DIGITS];   // per-wave digit counts -> exclusive prefix over waves
    __shared__ u32 dstart[RS_DIGITS];           // tile-local start of each digit's run
    __shared__ int32_t gdelta[RS_DIGITS];       // global position = local position + gdelta[digit] (mod 2^32)
    __shared__ u64 skeys[RS_TILE];
    __shared__ V svals[RS_TILE];
    __shared__ u32 scan_smem[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile_base = blockIdx.x * RS_TILE;
    const u32 tile_n = min((u32)RS_TILE, n - tile_base);
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) wcnt[w][tid] = 0;
    __syncthreads();

    u64 key[RS_ITEMS];
    V val[RS_ITEMS];
    u32 rank[RS_ITEMS];   // rank of the record among its wave's records with the same digit
    u32 dig[RS_ITEMS];
    const u32 wave_base = wave * (64 * RS_ITEMS);
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const bool valid = li < tile_n;
        key[j] = valid ? keys_in[tile_base + li] : 0;
        val[j] = valid ? vals_in[tile_base + li] : (V)0;
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        const bool valid = li < tile_n;
        const u32 d = rs_digit<V>(key[j], val[j], shift, mask, src);
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_BITS; b++) {
            const bool bit = (d >> b) & 1;
            const u64 bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        // m = lanes (valid) holding the same digit as this lane
        const u32 before = wcnt[wave][d];
        rank[j] = before + __popcll(m & lt);
        dig[j] = d;
        // the lowest lane of each digit group publishes the group's size; a wave's LDS operations
        // execute in program order, so the read above precedes this write for every lane
        if (valid && (m & lt) == 0) wcnt[wave][d] = before + __popcll(m);
    }
    __syncthreads();

    // per digit (thread == digit): exclusive prefix over waves, tile-local run start, global delta
    {
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            u32 c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
        u32 total;
        u32 incl = block_scan_inclusive(run, OpSum(), scan_smem, &total);
        u32 start = incl - run;
        dstart[tid] = start;
        gdelta[tid] = (int32_t)(table[(u64)tid * ntiles + blockIdx.x] - start);
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const u32 li = wave_base + j * 64 + lane;
        if (li < tile_n) {
            const u32 lp = dstart[dig[j]] + wcnt[wave][dig[j]] + rank[j];
            skeys[lp] = key[j];
            svals[lp] = val[j];
        }
    }
    __syncthreads();

#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        const u32 lp = i * RS_BLOCK + tid;
        if (lp < tile_n) {
            const u64 kk = skeys[lp];
            const V vv = svals[lp];
            const u32 d = rs_digit<V>(kk, vv, shift, mask, src);
            const u32 gp = lp + (u32)gdelta[d];
            keys_out[gp] = kk;
            vals_out[gp] = vv;
        }
    }
}

struct SortPass {
    int shift;
    int bits;
    int src;
};

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
            ProfScope ps(ctx, KP_SORT_HIST, (u64)n * (p.src ? sizeof(V) : 8));
            hipLaunchKernelGGL((rs_hist_kernel<V>), dim3(ntiles), dim3(RS_BLOCK), 0, ctx->stream, ki, vi, n, p.shift,
                               mask, p.src, table, ntiles);
        }
        KMG_TRY((device_scan<u32, u32, OpSum, true>(ctx, table, table, (u64)RS_DIGITS * ntiles, OpSum(), 0u, agg,
                                                    nullptr, false, KP_SORT_SCAN)));
        {
            ProfScope ps(ctx, KP_SORT_SCATTER, (u64)n * (8 + sizeof(V)) * 2);
            hipLaunchKernelGGL((rs_scatter_kernel<V>), dim3(ntiles), dim3(RS_BLOCK), 0, ctx->stream, ki, vi, ko, vo, n,
                               p.shift, mask, p.src, table, ntiles);
        }
        KMG_HIP(hipGetLastError());
        flip = !flip;
    }
    *result_in_1 = flip;
    return KMG_OK;
}
