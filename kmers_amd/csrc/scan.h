// scan.h -- device-wide prefix scans (reduce per tile -> scan of tile aggregates -> rescan).
// Hand-written for 64-wide wavefronts: per-lane serial run + __shfl_up wave scan + LDS across waves.
#pragma once
#include "common.h"
#include <algorithm>
#include <type_traits>

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

// ---- single-pass scan (chained tiles, decoupled look-back) ---------------------------------------------------------
// The three-launch form above reads the input twice and costs three launches; for the digit tables of the sort
// (256 counters per 4096 records, scanned once per pass) and the many short scans around them that is most of what
// the bookkeeping between the big kernels costs.  Here one launch does it: tiles take their numbers from a counter
// in the order they start (so every predecessor of a running tile is running or done: the waits below cannot
// deadlock), a tile publishes its aggregate at once, adds up its predecessors' words -- 256 at a time, four per lane --
// until it meets one that already holds an inclusive prefix, publishes its own inclusive prefix and writes its slice.
// Status word: value (46 bits: every count here is below 2^46) | state (2 bits: 1 aggregate, 2 inclusive prefix) |
// epoch (16 bits).  The epoch names the launch, so the status array needs no clearing between launches; the host
// clears it when the epoch wraps.  ctl[0] hands out tile numbers, ctl[1] counts finished tiles; the last one to
// finish sets both back to zero for the next launch on the stream.
#define CS_BLOCK 256
#ifndef CS_ITEMS
#define CS_ITEMS 64          // 16384 elements per tile: a few hundred tiles for the sort's digit tables
#endif
#ifndef CS_LOOK
#define CS_LOOK 4
#endif
#define CS_TILE (CS_BLOCK * CS_ITEMS)
#define CS_VAL_BITS 46
#define CS_VAL_MASK ((1ull << CS_VAL_BITS) - 1ull)

__device__ __forceinline__ u64 cs_word(u32 epoch, u32 state, u64 value) {
    return ((u64)epoch << 48) | ((u64)state << CS_VAL_BITS) | (value & CS_VAL_MASK);
}

// Called by every lane of ONE wave of the tile.  -> sum of the aggregates of tiles 0 .. tile-1 (all lanes).
__device__ __forceinline__ u64 cs_lookback(u64 *status, u32 epoch, u32 tile, u64 aggregate) {
    const int lane = threadIdx.x & 63;
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&status[0], cs_word(epoch, 2, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&status[tile], cs_word(epoch, 1, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 excl = 0;
    long long look = (long long)tile - 1;
    for (;;) {
        // CS_LOOK words per lane and step, all asked for at once: one memory round trip covers 64 * CS_LOOK predecessors
        // (element e = lane * CS_LOOK + j is tile look - e: nearest first)
        u64 w[CS_LOOK];
        bool ready;
        do {
            ready = true;
#pragma unroll
            for (int j = 0; j < CS_LOOK; j++) {
                const long long idx = look - (long long)(lane * CS_LOOK + j);
                w[j] = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : cs_word(epoch, 2, 0);       // before the first tile: an inclusive prefix of zero
            }
#pragma unroll
            for (int j = 0; j < CS_LOOK; j++)
                ready = ready && (u32)(w[j] >> 48) == epoch && ((w[j] >> CS_VAL_BITS) & 3u) != 0;
        } while (!ready);
        // the nearest predecessor that holds an inclusive prefix ends the walk: it and every aggregate before it count
        int jfirst = CS_LOOK;
#pragma unroll
        for (int j = CS_LOOK - 1; j >= 0; j--)
            if (((w[j] >> CS_VAL_BITS) & 3u) == 2u) jfirst = j;
        const u64 incl = __ballot(jfirst < CS_LOOK);
        const int first = incl ? __ffsll((unsigned long long)incl) - 1 : 64;
        u64 v = 0;
#pragma unroll
        for (int j = 0; j < CS_LOOK; j++)
            if (lane < first || (lane == first && j <= jfirst)) v += w[j] & CS_VAL_MASK;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
        excl += __shfl(v, 0, 64);
        if (incl) break;
        look -= 64 * CS_LOOK;
    }
    if (lane == 0)
        __hip_atomic_store(&status[tile], cs_word(epoch, 2, excl + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// The tile's number (dynamic: the order in which tiles start).  Ends behind a barrier.
__device__ __forceinline__ u32 cs_take_tile(u32 *ctl, u32 *s_tile) {
    if (threadIdx.x == 0) *s_tile = atomicAdd(&ctl[0], 1u);
    __syncthreads();
    return *s_tile;
}
// Last statement of a chained kernel (thread 0 of every tile).  -> true for the tile that finishes last.
__device__ __forceinline__ bool cs_tile_done(u32 *ctl, u32 ntiles) {
    // (no fence: what a tile wrote is read by later kernels, and what the last tile reads from the others are status
    // words, stored and loaded as agent-scope atomics; a release fence here would write the XCD's whole L2 back per tile)
    if (atomicAdd(&ctl[1], 1u) != ntiles - 1) return false;
    ctl[0] = 0;
    ctl[1] = 0;
    return true;
}

template <typename InT, typename OutT, bool EXCLUSIVE>
__global__ __launch_bounds__(CS_BLOCK) void scan_chained_kernel(const InT *__restrict__ in, OutT *__restrict__ out, u64 n,
                                                                 u64 *__restrict__ status, u32 *__restrict__ ctl, u32 epoch,
                                                                 u32 ntiles, bool write_total, OutT *__restrict__ total_out,
                                                                 u64 *__restrict__ total_add, u64 *__restrict__ total_add2) {
    __shared__ OutT smem[4];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    const u32 tile = cs_take_tile(ctl, &s_tile);
    const u64 base = (u64)tile * CS_TILE + (u64)threadIdx.x * CS_ITEMS;
    OutT v[CS_ITEMS];
    const bool full = base + CS_ITEMS <= n;
    if (full && sizeof(InT) == 4 && (((uintptr_t)in) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < CS_ITEMS; i += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(reinterpret_cast<const u32 *>(in) + base + i);
            v[i] = (OutT)(InT)q.x; v[i + 1] = (OutT)(InT)q.y; v[i + 2] = (OutT)(InT)q.z; v[i + 3] = (OutT)(InT)q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CS_ITEMS; i++) v[i] = base + i < n ? (OutT)in[base + i] : (OutT)0;
    }
    OutT acc = 0;
#pragma unroll
    for (int i = 0; i < CS_ITEMS; i++) acc += v[i];
    OutT total;
    const OutT inc = block_scan_inclusive(acc, OpSum(), smem, &total);
    if (threadIdx.x < 64) {
        const u64 e = cs_lookback(status, epoch, tile, (u64)total);
        if (threadIdx.x == 0) s_excl = e;
    }
    __syncthreads();
    OutT run = (OutT)s_excl + inc - acc;
    if (full && sizeof(OutT) == 4 && (((uintptr_t)out) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < CS_ITEMS; i += 4) {
            u32 o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (EXCLUSIVE) { o[j] = (u32)run; run += v[i + j]; }
                else { run += v[i + j]; o[j] = (u32)run; }
            }
            *reinterpret_cast<uint4 *>(reinterpret_cast<u32 *>(out) + base + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CS_ITEMS; i++) {
            if (EXCLUSIVE) {
                if (base + i < n) out[base + i] = run;
                run += v[i];
            } else {
                run += v[i];
                if (base + i < n) out[base + i] = run;
            }
        }
    }
    if (tile == ntiles - 1 && threadIdx.x == CS_BLOCK - 1) {
        // (the last thread of the last tile: its running sum is the grand total)
        if (EXCLUSIVE && write_total) out[n] = run;
        if (total_out) *total_out = run;
        if (total_add) atomicAdd((unsigned long long *)total_add, (unsigned long long)run);
        if (total_add2) atomicAdd((unsigned long long *)total_add2, (unsigned long long)run);
    }
    if (threadIdx.x == 0) cs_tile_done(ctl, ntiles);
}

// The context's status array and control words for chained kernels, and the next epoch.  -> epoch via *epoch.
static int chained_begin(kmg_ctx *ctx, u32 ntiles, u32 *epoch) {
    if (!ctx->scan_ctl) {
        KMG_HIP(hipMalloc((void **)&ctx->scan_ctl, 64));
        KMG_HIP(hipMemsetAsync(ctx->scan_ctl, 0, 64, ctx->stream));
    }
    if (ntiles > ctx->scan_status_cap) {
        const size_t cap = std::max<size_t>((size_t)ntiles * 2, 4096);
        if (ctx->scan_status) {
            KMG_HIP(hipStreamSynchronize(ctx->stream));
            KMG_HIP(hipFree(ctx->scan_status));
            ctx->scan_status = nullptr;
            ctx->scan_status_cap = 0;
        }
        KMG_HIP(hipMalloc((void **)&ctx->scan_status, cap * 8));
        KMG_HIP(hipMemsetAsync(ctx->scan_status, 0, cap * 8, ctx->stream));
        ctx->scan_status_cap = cap;
        ctx->scan_epoch = 0;
    }
    if (ctx->scan_epoch >= 0xFFFFu) {
        KMG_HIP(hipMemsetAsync(ctx->scan_status, 0, ctx->scan_status_cap * 8, ctx->stream));
        ctx->scan_epoch = 0;
    }
    *epoch = ++ctx->scan_epoch;             // 1 .. 65535: never the zero of a cleared word
    return KMG_OK;
}

// Host driver.  OpSum scans take the single-pass kernel; anything else the three-launch form, for which `agg` scratch
// must hold ceil(n / SCAN_TILE) OutT's.  In-place (in == out) is allowed when InT and OutT have the same size.
// total_add, total_add2 (sums only): the grand total is also added to these 64-bit words.
template <typename InT, typename OutT, typename Op, bool EXCLUSIVE>
static int device_scan(kmg_ctx *ctx, const InT *in, OutT *out, u64 n, Op op, OutT identity, OutT *agg,
                       OutT *total_out, bool write_total, int prof_id, u64 *total_add = nullptr, u64 *total_add2 = nullptr) {
    if (n == 0) {
        if (EXCLUSIVE && write_total) KMG_HIP(hipMemsetAsync(out, 0, sizeof(OutT), ctx->stream));
        if (total_out) KMG_HIP(hipMemsetAsync(total_out, 0, sizeof(OutT), ctx->stream));
        return KMG_OK;
    }
    ProfScope ps(ctx, prof_id, 0);
    if constexpr (std::is_same<Op, OpSum>::value && std::is_unsigned<OutT>::value) {
        const u32 ntiles = div_up_u64(n, CS_TILE);
        u32 epoch = 0;
        KMG_TRY(chained_begin(ctx, ntiles, &epoch));
        hipLaunchKernelGGL((scan_chained_kernel<InT, OutT, EXCLUSIVE>), dim3(ntiles), dim3(CS_BLOCK), 0, ctx->stream, in, out, n,
                           ctx->scan_status, ctx->scan_ctl, epoch, ntiles, write_total, total_out, total_add, total_add2);
        KMG_HIP(hipGetLastError());
        return KMG_OK;
    }
    u32 ntiles = div_up_u64(n, SCAN_TILE);
    hipLaunchKernelGGL((scan_tile_reduce<InT, OutT, Op>), dim3(ntiles), dim3(SCAN_BLOCK), 0, ctx->stream, in, agg, n,
                       op, identity);
    hipLaunchKernelGGL((scan_aggregates<OutT, Op>), dim3(1), dim3(SCAN_BLOCK), 0, ctx->stream, agg, ntiles, op,
                       identity, total_out);
    hipLaunchKernelGGL((scan_tile_apply<InT, OutT, Op, EXCLUSIVE>), dim3(ntiles), dim3(SCAN_BLOCK), 0, ctx->stream,
                       in, out, agg, n, op, identity, write_total);
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}
