// collect.h -- per-read aggregation downstream of the exact match.
//
// Utils/CollectByRead.java:159-184 turns the match result around: every (read, pos) of every matched search
// k-mer becomes a record carrying that k-mer's reference hit, and the records are put in ReadPos order
// (Reads/ReadPos.java:27-35: read, then pos as a signed short; the TreeSet of CollectByRead.java:105-124).
// Utils/ReadClassifier.java:178-220 then reduces the records of each read to one call.  The reference does both
// through text files; here they consume the join directly:
//   cbr_count / scan / cbr_emit : matched occurrences -> (sortable ReadPos key, hit index) records
//   hybrid_sort_records<u32>    : the same sort the database build uses, on the key's used bits only
//   cbr_unpack                  : keys back to (read, pos)
//   rc_single / rc_check / rc_call : processRead as two passes of per-read atomics, no sort at all
#pragma once
#include "common.h"

// values of search k-mer s that take part: all of them if it has a hit
__global__ void cbr_count_kernel(const u64 *__restrict__ val_off, const long long *__restrict__ hit, u64 U, u32 *__restrict__ cnt) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < U) cnt[s] = hit[s] >= 0 ? (u32)(val_off[s + 1] - val_off[s]) : 0u;
}

// largest read index among the values (for the number of key bits the sort has to look at)
__global__ void cbr_maxread_kernel(const int32_t *__restrict__ val_a, u64 V, u32 *out) {
    u32 m = 0;
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (u64)gridDim.x * blockDim.x) m = max(m, (u32)val_a[v]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (u32)__shfl_down(m, d, 64));
    __shared__ u32 wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > *reinterpret_cast<volatile u32 *>(out)) atomicMax(out, m);
    }
}

// CB_LANES lanes per search k-mer (lists hold 3-4 occurrences on average -- measured 8: 1.73, 4: 1.26, 2: 1.14, 1: 1.80 ms -- but a low-complexity k-mer can hold
// thousands): record = key (read << 16 | pos16 with the sign bit flipped, left-aligned by `shift`) + hit index.
#ifndef CB_LANES
#define CB_LANES 4
#endif
__global__ __launch_bounds__(256) void cbr_emit_kernel(const u64 *__restrict__ val_off, const int32_t *__restrict__ val_a,
                                                       const int32_t *__restrict__ val_b, const long long *__restrict__ hit,
                                                       const u64 *__restrict__ out_off, u64 U, int shift,
                                                       u64 *__restrict__ out_keys, u32 *__restrict__ out_hit) {
    const u64 s = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / CB_LANES;
    const u32 lane = threadIdx.x % CB_LANES;
    if (s >= U) return;
    const long long h = hit[s];
    if (h < 0) return;
    const u64 v0 = val_off[s], n = val_off[s + 1] - v0, o = out_off[s];
    for (u64 j = lane; j < n; j += CB_LANES) {
        const u64 rp = ((u64)(u32)val_a[v0 + j] << 16) | (u64)(((u32)val_b[v0 + j] ^ 0x8000u) & 0xFFFFu);
        out_keys[o + j] = rp << shift;
        out_hit[o + j] = (u32)h;
    }
}

__global__ void cbr_unpack_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ hits, u64 n, int shift,
                                  int32_t *__restrict__ read, int32_t *__restrict__ pos, long long *__restrict__ ref_index) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 rp = keys[i] >> shift;
    read[i] = (int32_t)(rp >> 16);
    pos[i] = (int32_t)(int16_t)((u32)(rp & 0xFFFFu) ^ 0x8000u);
    ref_index[i] = (long long)hits[i];
}

// ---- ReadClassifier.processRead (ReadClassifier.java:178-220) -------------------------------------------------
// call = the one taxon t >= 0 such that every single-taxon hit of the read names t and every hit's taxon set
// contains t; no call otherwise.  Per read one int2 = (min, max) of its single-taxon hits, starting at
// (INT_MAX, INT_MIN): one 8-byte random access per occurrence instead of two 4-byte ones.
__global__ __launch_bounds__(256) void rc_single_kernel(const u64 *__restrict__ s_off, const int32_t *__restrict__ s_read,
                                                        const long long *__restrict__ hit, u64 U,
                                                        const u64 *__restrict__ r_off, const int32_t *__restrict__ r_taxon,
                                                        u64 n_reads, int2 *mm, u32 *err, uint8_t *seen) {
    const u64 s = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / CB_LANES;
    const u32 lane = threadIdx.x % CB_LANES;
    if (s >= U) return;
    const long long h = hit[s];
    if (h < 0) return;
    const u64 v0 = s_off[s], n = s_off[s + 1] - v0;
    const bool single = r_off[h + 1] - r_off[h] == 1;
    const int32_t t = single ? r_taxon[r_off[h]] : 0;
    for (u64 j = lane; j < n; j += CB_LANES) {
        const u32 r = (u32)s_read[v0 + j];
        if (r >= n_reads) { atomicOr(err, 1u); continue; }
        // (same look-first rule: a read's ~100 occurrences would all store the same byte)
        if (seen && !*reinterpret_cast<volatile uint8_t *>(&seen[r])) seen[r] = 1;
        if (single) {
            // mn only falls and mx only rises, so a plain look first is safe and spares most of the ~100 atomics
            // a read's occurrences would aim at its two words
            const u64 w = *reinterpret_cast<volatile u64 *>(&mm[r]);
            if ((int32_t)(u32)w > t) atomicMin(&mm[r].x, t);
            if ((int32_t)(u32)(w >> 32) < t) atomicMax(&mm[r].y, t);
        }
    }
}

__global__ __launch_bounds__(256) void rc_check_kernel(const u64 *__restrict__ s_off, const int32_t *__restrict__ s_read,
                                                       const long long *__restrict__ hit, u64 U,
                                                       const u64 *__restrict__ r_off, const int32_t *__restrict__ r_taxon,
                                                       u64 n_reads, const int2 *__restrict__ mm, u32 *bad) {
    const u64 s = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / CB_LANES;
    const u32 lane = threadIdx.x % CB_LANES;
    if (s >= U) return;
    const long long h = hit[s];
    if (h < 0) return;
    const u64 v0 = s_off[s], n = s_off[s + 1] - v0;
    const u64 t0 = r_off[h], tn = r_off[h + 1] - t0;
    for (u64 j = lane; j < n; j += CB_LANES) {
        const u32 r = (u32)s_read[v0 + j];
        if (r >= n_reads) continue;
        const int2 m = mm[r];
        const int32_t t = m.x;
        if (t != m.y) continue;                       // no single candidate: no call whatever the other hits hold
        u64 lo = 0, hi = tn;                          // taxa of a k-mer are ascending (TreeCountMap order)
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (r_taxon[t0 + mid] < t) lo = mid + 1; else hi = mid;
        }
        if (!(lo < tn && r_taxon[t0 + lo] == t)) bad[r] = 1u;
    }
}

__global__ void rc_call_kernel(const int2 *__restrict__ mm, const u32 *__restrict__ bad, u64 n_reads, int32_t *__restrict__ call) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int2 m = mm[r];
    call[r] = (m.x == m.y && m.x >= 0 && !bad[r]) ? m.x : -1;
}

// flag[0] = 1 if any hit index is neither -1 nor one of the n_rows rows
__global__ void hit_range_kernel(const long long *__restrict__ hit, u64 n, long long n_rows, u32 *__restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (hit[i] < -1 || hit[i] >= n_rows)) flag[0] = 1u;
}

__global__ void rc_init_kernel(int2 *mm, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mm[i] = make_int2(0x7FFFFFFF, (int)0x80000000);
}

