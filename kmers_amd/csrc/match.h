// match.h -- stage 3: sorted-set join of search k-mers against a reference DB, and TaxaCounts.
//
// Replaces DB.quickMatchCommonKey + StreamUtils.matchTwoStreams (Database/DB.java:129-137, jar A.6: a
// two-pointer inner join per key bucket of two strictly ascending k-mer sequences) and
// TaxaCounts.CountDataAll.add (Utils/TaxaCounts.java:386-395).  Both sides are unique and sorted, so the
// join result is exactly "is this search k-mer present in the reference, and where".
#pragma once
#include "common.h"

#define MT_BLOCK 256

// One thread per search k-mer: binary search inside the reference's bucket for the k-mer's
// Kmer.key(key_len) prefix.  Neighbouring threads hold neighbouring (sorted) keys, so their probes walk
// the same cache lines of the reference array.
__global__ __launch_bounds__(MT_BLOCK) void match_kernel(const u64 *__restrict__ skeys, u64 ns,
                                                          const u64 *__restrict__ rkeys,
                                                          const u64 *__restrict__ rbucket, int key_len,
                                                          long long *__restrict__ hit, u64 *__restrict__ nhits) {
    u64 i = (u64)blockIdx.x * MT_BLOCK + threadIdx.x;
    bool found = false;
    if (i < ns) {
        const u64 key = skeys[i];
        const u64 b = key >> (64 - 2 * key_len);
        u64 lo = rbucket[b], hi = rbucket[b + 1];
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (rkeys[mid] < key) lo = mid + 1; else hi = mid;
        }
        found = lo < rbucket[b + 1] && rkeys[lo] == key;
        hit[i] = found ? (long long)lo : -1ll;
    }
    if (nhits) {
        u64 bal = __ballot(found);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(nhits, (u64)__popcll(bal));
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
