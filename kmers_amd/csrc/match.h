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

// number of threads of the block for which pred holds (all threads must call)
__device__ __forceinline__ u32 mt_block_count(bool pred, u32 *smem) {
    const u64 bal = __ballot(pred);
    __syncthreads();                       // previous readers of smem are done
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

// The whole block searches [lo, hi) of the ascending array a for the first element that is not `less`
// than key (STRICT: first element > key).  256-ary: every round each thread probes one position.
template <bool STRICT>
__device__ __forceinline__ u64 mt_block_bound(const u64 *__restrict__ a, u64 lo, u64 hi, u64 key, u32 *smem) {
    const u32 t = threadIdx.x;
    while (hi > lo) {
        const u64 n = hi - lo;
        if (n <= MT_BLOCK) {
            bool before = false;
            if (t < n) { const u64 v = a[lo + t]; before = STRICT ? v <= key : v < key; }
            return lo + mt_block_count(before, smem);
        }
        // probes split [lo, hi) into 257 parts; predicates are 1..1 0..0 along t
        const u64 p = lo + (n * (u64)(t + 1)) / (MT_BLOCK + 1);
        const u64 v = a[p];
        const bool before = STRICT ? v <= key : v < key;
        const u32 c = mt_block_count(before, smem);
        const u64 nlo = c == 0 ? lo : lo + (n * (u64)c) / (MT_BLOCK + 1) + 1;
        const u64 nhi = c == MT_BLOCK ? hi : lo + (n * (u64)(c + 1)) / (MT_BLOCK + 1);
        lo = nlo; hi = nhi;
    }
    return lo;
}

// One workgroup joins MT_TILE consecutive search k-mers.  Both sides are ascending, so the reference
// k-mers that can match them form one contiguous range [lower_bound(first), upper_bound(last)); the block
// finds it cooperatively, stages it in LDS (it is a few hundred k-mers when the two sets are of similar
// density) and every search k-mer is then binary-searched in LDS instead of through L2.  Ranges that do not
// fit are searched in place.  hit[i] = index of the k-mer in the reference, or -1.
__global__ __launch_bounds__(MT_BLOCK) void match_kernel(const u64 *__restrict__ skeys, u64 ns,
                                                          const u64 *__restrict__ rkeys, u64 nr,
                                                          long long *__restrict__ hit, u64 *__restrict__ nhits) {
    __shared__ u64 sref[MT_RMAX];
    __shared__ u32 smem[4];
    const u64 q0 = (u64)blockIdx.x * MT_TILE;
    const u64 q1 = q0 + MT_TILE < ns ? q0 + MT_TILE : ns;
    const u64 r_lo = mt_block_bound<false>(rkeys, 0, nr, skeys[q0], smem);
    const u64 r_hi = mt_block_bound<true>(rkeys, r_lo, nr, skeys[q1 - 1], smem);
    const u32 cnt = (u32)(r_hi - r_lo < MT_RMAX ? r_hi - r_lo : MT_RMAX);
    const bool staged = r_hi - r_lo <= MT_RMAX;
    if (staged)
        for (u32 i = threadIdx.x; i < cnt; i += MT_BLOCK) sref[i] = rkeys[r_lo + i];
    __syncthreads();
    u32 nfound = 0;
#pragma unroll
    for (int j = 0; j < MT_ITEMS; j++) {
        const u64 i = q0 + (u64)j * MT_BLOCK + threadIdx.x;
        if (i < q1) {
            const u64 key = skeys[i];
            long long res = -1;
            if (staged) {
                u32 lo = 0, hi = cnt;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (sref[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo < cnt && sref[lo] == key) res = (long long)(r_lo + lo);
            } else {
                u64 lo = r_lo, hi = r_hi;
                while (lo < hi) {
                    const u64 mid = (lo + hi) >> 1;
                    if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo < r_hi && rkeys[lo] == key) res = (long long)lo;
            }
            hit[i] = res;
            nfound += res >= 0;
        }
    }
    if (nhits) {   // uniform
        // per-thread counts <= MT_ITEMS: sum the block with one more counting round per bit
        u32 total = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) total += mt_block_count((nfound >> b) & 1, smem) << b;
        if (threadIdx.x == 0 && total) atomicAdd(nhits, (u64)total);
    }
}

// ---- TaxaCounts ----------------------------------------------------------------------------------------
// taxa[] = ascending distinct taxa of the DB (built by sorting val_a once).  Thread per k-mer.
#define TX_LDS_MAX 2048

__device__ __forceinline__ u32 taxa_find(const int32_t *taxa, u32 nt, int32_t t) {
    u32 lo = 0, hi = nt;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (taxa[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void taxa_count_kernel(const u64 *__restrict__ val_off,
                                                          const int32_t *__restrict__ val_a, u64 U,
                                                          const int32_t *__restrict__ taxa, u32 nt,
                                                          u64 *__restrict__ match, u64 *__restrict__ unique) {
    __shared__ int32_t s_taxa[USE_LDS ? TX_LDS_MAX : 1];
    __shared__ u32 s_match[USE_LDS ? TX_LDS_MAX : 1];
    __shared__ u32 s_unique[USE_LDS ? TX_LDS_MAX : 1];
    if (USE_LDS) {
        for (u32 i = threadIdx.x; i < nt; i += blockDim.x) { s_taxa[i] = taxa[i]; s_match[i] = 0; s_unique[i] = 0; }
        __syncthreads();
    }
    for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (u64)gridDim.x * blockDim.x) {
        const u64 a = val_off[u], b = val_off[u + 1];
        for (u64 v = a; v < b; v++) {
            const int32_t t = val_a[v];
            if (USE_LDS) {
                u32 idx = taxa_find(s_taxa, nt, t);
                atomicAdd(&s_match[idx], 1u);
                if (b - a == 1) atomicAdd(&s_unique[idx], 1u);
            } else {
                u32 idx = taxa_find(taxa, nt, t);
                atomicAdd(&match[idx], 1ull);
                if (b - a == 1) atomicAdd(&unique[idx], 1ull);
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < nt; i += blockDim.x) {
            if (s_match[i]) atomicAdd(&match[i], (u64)s_match[i]);
            if (s_unique[i]) atomicAdd(&unique[i], (u64)s_unique[i]);
        }
    }
}

// helpers to derive the distinct taxa: expand val_a to sortable keys, then flag run heads
__global__ void taxa_expand_kernel(const int32_t *__restrict__ val_a, u64 V, u64 *__restrict__ keys, u32 *__restrict__ vals) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) { keys[i] = (u64)((u32)val_a[i] ^ 0x80000000u); vals[i] = 0; }
}
__global__ void taxa_headflag_kernel(const u64 *__restrict__ keys, u64 V, u32 *__restrict__ flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void taxa_compact_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ flag,
                                    const u32 *__restrict__ pos, u64 V, int32_t *__restrict__ taxa) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V && flag[i]) taxa[pos[i]] = (int32_t)((u32)keys[i] ^ 0x80000000u);
}
