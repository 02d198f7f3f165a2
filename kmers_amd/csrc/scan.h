// scan.h -- device-wide prefix scans (reduce per tile -> scan of tile aggregates -> rescan).
// Hand-written for 64-wide wavefronts: per-lane serial run + __shfl_up wave scan + LDS across waves.
#pragma once
#include "common.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

struct OpSum {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct OpMax {
    template <typename T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};

// Inclusive scan of one value per thread across a 256-thread block.  Returns the inclusive prefix;
// *block_total = aggregate of the whole block.  `smem` must hold WAVES T's.
template <typename T, typename Op, int WAVES = SCAN_BLOCK / 64>
__device__ __forceinline__ T block_scan_inclusive(T v, Op op, T *smem, T *block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v = op(o, v);
    }
    if (lane == 63) smem[wave] = v;
    __syncthreads();
    T total = smem[0];
    T carry = smem[0];
    bool have = false;
#pragma unroll
    for (int w = 1; w < WAVES; w++) {
        if (w == wave) { carry = total; have = true; }
        total = op(total, smem[w]);
    }
    if (have) v = op(carry, v);
    *block_total = total;
    __syncthreads();
    return v;
}

template <typename InT, typename OutT, typename Op>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_reduce(const InT *__restrict__ in, OutT *__restrict__ agg,
                                                                u64 n, Op op, OutT identity) {
    __shared__ OutT smem[4];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    OutT acc = identity;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) acc = op(acc, (OutT)in[base + i]);
    OutT total;
    block_scan_inclusive(acc, op, smem, &total);
    if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// Single block: exclusive scan of the tile aggregates in place; total to *total_out (may be null).
template <typename OutT, typename Op>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_aggregates(OutT *__restrict__ agg, u32 ntiles, Op op, OutT identity,
                                                               OutT *__restrict__ total_out) {
    __shared__ OutT smem[4];
    __shared__ OutT line[SCAN_BLOCK];
    OutT carry = identity;
    for (u32 base = 0; base < ntiles; base += SCAN_BLOCK) {
        u32 i = base + threadIdx.x;
        OutT v = i < ntiles ? agg[i] : identity;
        OutT total;
        OutT inc = block_scan_inclusive(v, op, smem, &total);
        line[threadIdx.x] = inc;
        __syncthreads();
        // exclusive[i] = carry (+) inclusive[i-1]  (no inverse needed, so it also works for max)
        OutT ex = threadIdx.x == 0 ? carry : op(carry, line[threadIdx.x - 1]);
        if (i < ntiles) agg[i] = ex;
        carry = op(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// Rescan each tile with its carry-in.  EXCLUSIVE selects exclusive/inclusive output.
// If write_total, out[n] receives the grand total (exclusive scans only; caller sizes out n+1).
template <typename InT, typename OutT, typename Op, bool EXCLUSIVE>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_apply(const InT *__restrict__ in, OutT *__restrict__ out,
                                                               const OutT *__restrict__ agg, u64 n, Op op,
                                                               OutT identity, bool write_total) {
    __shared__ OutT smem[4];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    OutT v[SCAN_ITEMS];
    OutT acc = identity;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? (OutT)in[base + i] : identity;
        acc = op(acc, v[i]);
    }
    OutT total;
    OutT inc = block_scan_inclusive(acc, op, smem, &total);
    // exclusive prefix of this thread = carry-in (+) inclusive of the previous thread
    __shared__ OutT line[SCAN_BLOCK];
    line[threadIdx.x] = inc;
    __syncthreads();
    OutT run = agg[blockIdx.x];
    if (threadIdx.x > 0) run = op(run, line[threadIdx.x - 1]);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (EXCLUSIVE) {
            if (base + i < n) out[base + i] = run;
            run = op(run, v[i]);
        } else {
            run = op(run, v[i]);
            if (base + i < n) out[base + i] = run;
        }
    }
    if (EXCLUSIVE && write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = run;
}

// Host driver.  `agg` scratch must hold ceil(n / SCAN_TILE) OutT's.  In-place (in == out) is allowed
// when InT and OutT have the same size.
template <typename InT, typename OutT, typename Op, bool EXCLUSIVE>
static int device_scan(kmg_ctx *ctx, const InT *in, OutT *out, u64 n, Op op, OutT identity, OutT *agg,
                       OutT *total_out, bool write_total, int prof_id) {
    if (n == 0) {
        if (EXCLUSIVE && write_total) KMG_HIP(hipMemsetAsync(out, 0, sizeof(OutT), ctx->stream));
        if (total_out) KMG_HIP(hipMemsetAsync(total_out, 0, sizeof(OutT), ctx->stream));
        return KMG_OK;
    }
    u32 ntiles = div_up_u64(n, SCAN_TILE);
    ProfScope ps(ctx, prof_id, 0);
    hipLaunchKernelGGL((scan_tile_reduce<InT, OutT, Op>), dim3(ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, in, agg, n,
                       op, identity);
    hipLaunchKernelGGL((scan_aggregates<OutT, Op>), dim3(1), dim3(SCAN_BLOCK), 0, ctx->stream, agg, ntiles, op,
                       identity, total_out);
    hipLaunchKernelGGL((scan_tile_apply<InT, OutT, Op, EXCLUSIVE>), dim3(ntiles), dim3(SCAN_BLOCK), 0, ctx->stream,
                       in, out, agg, n, op, identity, write_total);
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}
