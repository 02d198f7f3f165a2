// reduce.h -- stage 2b: deduplicate the sorted records into a CSR database, bucket offsets, block encoding.
//
// Replaces the reducer half of MakeAndWriteKey.call (KmerFiles/FileCreator.java:294-301: TreeCountMap.add
// per taxon | HashSet<ReadPos>.add) and its prefix-delta encoder (:312-347).  One read of the sorted
// records, one write of the CSR: HBM-bound streaming with a two-level prefix sum for the output slots.
#pragma once
#include "common.h"
#include "scan.h"

#define RD_BLOCK 256
#define RD_ITEMS 8
#define RD_TILE (RD_BLOCK * RD_ITEMS)

// A record starts a new k-mer run if its key differs from its predecessor's; a new (k-mer, taxon) run if
// additionally / alternatively the payload differs.  For read DBs every record is its own value.
// Coalesced: lane i of a round reads record base + i (and its predecessor, which the neighbouring lane just
// loaded: a cache hit); heads are ranked with ballots, so the lanes of a wave store to consecutive slots.
#define RD_WAVES (RD_BLOCK / 64)
template <typename V, bool IS_REF>
__global__ __launch_bounds__(RD_BLOCK) void rd_count_kernel(const u64 *__restrict__ keys, const V *__restrict__ vals,
                                                             u32 n, u32 *__restrict__ cnt_k, u32 *__restrict__ cnt_v) {
    __shared__ u32 smem[4];
    const u32 base = blockIdx.x * RD_TILE;
    u32 ck = 0, cv = 0;
#pragma unroll
    for (int r = 0; r < RD_ITEMS; r++) {
        const u32 idx = base + r * RD_BLOCK + threadIdx.x;
        bool hk = false, hv = false;
        if (idx < n) {
            const u64 kk = keys[idx];
            hk = idx == 0 || kk != keys[idx - 1];
            if (IS_REF) hv = hk || vals[idx] != vals[idx - 1];
        }
        ck += __popcll(__ballot(hk));       // wave-uniform counts
        if (IS_REF) cv += __popcll(__ballot(hv));
    }
    const bool first = (threadIdx.x & 63) == 0;
    u32 tk, tv = 0;
    block_scan_inclusive(first ? ck : 0u, OpSum(), smem, &tk);
    if (IS_REF) block_scan_inclusive(first ? cv : 0u, OpSum(), smem, &tv);
    if (threadIdx.x == 0) {
        cnt_k[blockIdx.x] = tk;
        if (IS_REF) cnt_v[blockIdx.x] = tv;
    }
}

template <typename V, bool IS_REF>
__global__ __launch_bounds__(RD_BLOCK) void rd_write_kernel(const u64 *__restrict__ keys, const V *__restrict__ vals,
                                                             u32 n, const u64 *__restrict__ off_k,
                                                             const u64 *__restrict__ off_v, u64 *__restrict__ ukeys,
                                                             u64 *__restrict__ val_off, int32_t *__restrict__ val_a,
                                                             int32_t *__restrict__ val_b, u32 *__restrict__ vstart) {
    __shared__ u32 wck[RD_ITEMS][RD_WAVES], wcv[RD_ITEMS][RD_WAVES];     // heads per (round, wave), in record order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 base = blockIdx.x * RD_TILE;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 kk[RD_ITEMS];
    V vv[RD_ITEMS];
    u32 hk = 0, hv = 0;
    uint8_t prek[RD_ITEMS], prev[RD_ITEMS];
    u64 kp[RD_ITEMS];
    V vp[RD_ITEMS];
#pragma unroll
    for (int r = 0; r < RD_ITEMS; r++) {          // every load first, at clamped indices: guarded loads wait one by one
        const u32 idx = base + r * RD_BLOCK + tid;
        const u32 ic = idx < n ? idx : n - 1, ip = ic ? ic - 1 : 0;
        kk[r] = keys[ic];
        vv[r] = vals[ic];
        kp[r] = keys[ip];
        if (IS_REF) vp[r] = vals[ip];
    }
#pragma unroll
    for (int r = 0; r < RD_ITEMS; r++) {
        const u32 idx = base + r * RD_BLOCK + tid;
        const bool h = idx < n && (idx == 0 || kk[r] != kp[r]);
        const bool h2 = IS_REF && idx < n && (h || vv[r] != vp[r]);
        const u64 bk = __ballot(h);
        prek[r] = (uint8_t)__popcll(bk & lt);
        hk |= (u32)h << r;
        if (lane == 0) wck[r][wave] = __popcll(bk);
        if (IS_REF) {
            const u64 bv = __ballot(h2);
            prev[r] = (uint8_t)__popcll(bv & lt);
            hv |= (u32)h2 << r;
            if (lane == 0) wcv[r][wave] = __popcll(bv);
        }
    }
    __syncthreads();
    if (tid < RD_ITEMS * RD_WAVES) {                    // 32 (round, wave) counts: exclusive scan by one half-wave
        u32 v = wck[tid >> 2][tid & 3], inc = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            u32 o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        wck[tid >> 2][tid & 3] = inc - v;
    } else if (IS_REF && tid >= 64 && tid < 64 + RD_ITEMS * RD_WAVES) {
        const int t = tid - 64;
        u32 v = wcv[t >> 2][t & 3], inc = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            u32 o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        wcv[t >> 2][t & 3] = inc - v;
    }
    __syncthreads();
    const u64 ubase = off_k[blockIdx.x];
    const u64 vbase = IS_REF ? off_v[blockIdx.x] : 0;
#pragma unroll
    for (int r = 0; r < RD_ITEMS; r++) {
        const u32 idx = base + r * RD_BLOCK + tid;
        if (idx >= n) break;
        if (IS_REF) {
            if ((hv >> r) & 1) {
                const u64 v = vbase + wcv[r][wave] + prev[r];
                if ((hk >> r) & 1) {
                    const u64 u = ubase + wck[r][wave] + prek[r];
                    ukeys[u] = kk[r];
                    val_off[u] = v;
                }
                val_a[v] = (int32_t)vv[r];
                vstart[v] = idx;
            }
        } else {
            if ((hk >> r) & 1) {
                const u64 u = ubase + wck[r][wave] + prek[r];
                ukeys[u] = kk[r];
                val_off[u] = idx;
            }
            // payload = read << 16 | pos16  ->  ReadPos(int read, short pos)
            u64 p = (u64)vv[r];
            val_a[idx] = (int32_t)(p >> 16);
            val_b[idx] = (int32_t)(int16_t)(p & 0xFFFFu);
        }
    }
}

// ---- read DBs: only the keys are needed to find the k-mer runs; every record is its own value and the
// sorted payload array (read << 16 | pos16) IS the value array (unpacked on demand by rd_unpack_kernel).
// Coalesced: lane i of a round reads key base + i (and its predecessor), heads are ranked with ballots.
#define RDK_ROUNDS 8
#define RDK_TILE (RD_BLOCK * RDK_ROUNDS)

__global__ __launch_bounds__(RD_BLOCK) void rdk_count_kernel(const u64 *__restrict__ keys, u32 n, u32 *__restrict__ cnt_k) {
    __shared__ u32 smem[4];
    const u32 base = blockIdx.x * RDK_TILE;
    u32 c = 0;
#pragma unroll
    for (int r = 0; r < RDK_ROUNDS; r++) {
        const u32 idx = base + r * RD_BLOCK + threadIdx.x;
        const u32 ic = idx < n ? idx : n - 1, ip = ic ? ic - 1 : 0;       // clamped, not guarded
        const bool h = idx < n && (idx == 0 || keys[ic] != keys[ip]);
        c += __popcll(__ballot(h));       // wave-uniform count
    }
    u32 total;
    block_scan_inclusive((threadIdx.x & 63) == 0 ? c : 0u, OpSum(), smem, &total);
    if (threadIdx.x == 0) cnt_k[blockIdx.x] = total;
}

__global__ __launch_bounds__(RD_BLOCK) void rdk_write_kernel(const u64 *__restrict__ keys, u32 n,
                                                              const u64 *__restrict__ off_k, u64 *__restrict__ ukeys,
                                                              u64 *__restrict__ val_off) {
    __shared__ u32 wc[RDK_ROUNDS][RD_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 base = blockIdx.x * RDK_TILE;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 kk[RDK_ROUNDS];
    u64 kp[RDK_ROUNDS];
    u32 hmask = 0, pre[RDK_ROUNDS];
#pragma unroll
    for (int r = 0; r < RDK_ROUNDS; r++) {
        const u32 idx = base + r * RD_BLOCK + tid;
        const u32 ic = idx < n ? idx : n - 1, ip = ic ? ic - 1 : 0;
        kk[r] = keys[ic];
        kp[r] = keys[ip];
    }
#pragma unroll
    for (int r = 0; r < RDK_ROUNDS; r++) {
        const u32 idx = base + r * RD_BLOCK + tid;
        const bool h = idx < n && (idx == 0 || kk[r] != kp[r]);
        const u64 bal = __ballot(h);
        pre[r] = __popcll(bal & lt);
        hmask |= (u32)h << r;
        if (lane == 0) wc[r][wave] = __popcll(bal);
    }
    __syncthreads();
    if (tid < RDK_ROUNDS * (RD_BLOCK / 64)) {     // 32 (round, wave) counts in record order
        u32 v = wc[tid >> 2][tid & 3], inc = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            u32 o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        wc[tid >> 2][tid & 3] = inc - v;
    }
    __syncthreads();
    const u64 ubase = off_k[blockIdx.x];
#pragma unroll
    for (int r = 0; r < RDK_ROUNDS; r++) {
        if ((hmask >> r) & 1) {
            const u64 u = ubase + wc[r][wave] + pre[r];
            ukeys[u] = kk[r];
            val_off[u] = base + r * RD_BLOCK + tid;
        }
    }
}

// ReadPos(int read, short pos) from the packed payload
__global__ void rd_unpack_kernel(const u64 *__restrict__ packed, u64 V, int32_t *__restrict__ val_a, int32_t *__restrict__ val_b) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) {
        u64 p = packed[i];
        val_a[i] = (int32_t)(p >> 16);
        val_b[i] = (int32_t)(int16_t)(p & 0xFFFFu);
    }
}

// ... and from the compact 32-bit form (read << pos_bits | pos)
__global__ void rd_unpack32_kernel(const u32 *__restrict__ packed, u64 V, int pos_bits, int32_t *__restrict__ val_a,
                                   int32_t *__restrict__ val_b) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) {
        const u32 p = packed[i];
        val_a[i] = (int32_t)(p >> pos_bits);
        val_b[i] = (int32_t)(int16_t)(p & ((1u << pos_bits) - 1u));
    }
}

// length of the longest sequence of a batch
__global__ void seq_maxlen_kernel(const u64 *__restrict__ seq_off, u64 nseq, u32 *out) {
    u32 m = 0;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < nseq; s += (u64)gridDim.x * blockDim.x) {
        const u64 len = seq_off[s + 1] - seq_off[s];
        m = max(m, len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (u32)__shfl_down(m, d, 64));
    // one atomic per block, and only if it can raise the maximum (a million same-address atomics took ~100 us)
    __shared__ u32 wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > *reinterpret_cast<volatile u32 *>(out)) atomicMax(out, m);
    }
}

// compact payloads back to the 64-bit form (a second batch arrived)
__global__ void rd_widen_kernel(const u32 *__restrict__ in, u64 n, int pos_bits, u64 *__restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u32 p = in[i];
        out[i] = ((u64)(p >> pos_bits) << 16) | (u64)(p & ((1u << pos_bits) - 1u));
    }
}

// counts of a reference DB: occurrences of each (k-mer, taxon) = distance to the next run start
__global__ void rd_counts_kernel(const u32 *__restrict__ vstart, const u64 *__restrict__ totals /*[0]=U,[1]=V*/,
                                 u32 n, int32_t *__restrict__ val_b) {
    u64 V = totals[1];
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (u64)gridDim.x * blockDim.x) {
        u32 next = v + 1 < V ? vstart[v + 1] : n;
        u32 c = next - vstart[v];
        val_b[v] = c > 0x7FFFFFFFu ? 0x7FFFFFFF : (int32_t)c;
    }
}

// val_off[U] = V ; bucket_off[b] = first k-mer whose Kmer.key(key_len) (Kmers/Kmer.java:47-55) is >= b
__global__ void rd_buckets_kernel(const u64 *__restrict__ ukeys, const u64 *__restrict__ totals,
                                  u64 *__restrict__ val_off, u64 *__restrict__ bucket_off, u32 nbuckets, int key_len) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 U = totals[0];
    if (b == 0) val_off[U] = totals[1];
    if (b > nbuckets) return;
    if (b == nbuckets) { bucket_off[b] = U; return; }
    const u64 target = (u64)b << (64 - 2 * key_len);
    u64 lo = 0, hi = U;
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (ukeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    bucket_off[b] = lo;
}

// ---- block encoding (FileCreator.java:312-347) -----------------------------------------------------------
// Per k-mer: [s = bytes shared with the previous compressedBytes in this bucket][compressedBytes[s:]]
// then CountDataType.compress (Counts/CountDataType.java:31-46) or SetCompressor<ReadPos> (jar, A.4).
__device__ __forceinline__ int enc_shared(u64 prev, u64 cur, int l) {
    // compressedBytes = [len][packed bytes]; called for k-mers of equal length (the length byte matches)
    u64 x = prev ^ cur;
    int same = x == 0 ? 8 : (__clzll((long long)x) >> 3);
    if (same > l) same = l;
    return 1 + same;
}

// lens (may be null: every k-mer has k bases) = the length byte of every k-mer; k-mers of different lengths share
// nothing, their first bytes differ (FileCreator.java:244-252).
__global__ void enc_size_kernel(const u64 *__restrict__ ukeys, const uint8_t *__restrict__ lens, const u64 *__restrict__ val_off,
                                u64 U, int k, int key_len, int kind, u32 *__restrict__ sizes) {
    u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int len = lens ? lens[u] : k;
    const int l = (len - 1) / 4 + 1;
    const int sh = 64 - 2 * key_len;
    u64 cur = ukeys[u];
    int s = 0;
    if (u > 0) {
        u64 prev = ukeys[u - 1];
        if ((prev >> sh) == (cur >> sh) && (!lens || lens[u - 1] == len)) s = enc_shared(prev, cur, l);
    }
    u64 nv = val_off[u + 1] - val_off[u];
    u64 sz = 1 + (u64)(l + 1 - s) + (kind == KMG_PAYLOAD_TAXON ? 2 + 5 * nv : 4 + 6 * nv);
    sizes[u] = (u32)sz;
}

__global__ void enc_write_kernel(const u64 *__restrict__ ukeys, const uint8_t *__restrict__ lens, const u64 *__restrict__ val_off,
                                 const int32_t *__restrict__ val_a, const int32_t *__restrict__ val_b, u64 U, int k,
                                 int key_len, int kind, const u64 *__restrict__ enc_off, uint8_t *__restrict__ out) {
    u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int len = lens ? lens[u] : k;
    const int l = (len - 1) / 4 + 1;
    const int sh = 64 - 2 * key_len;
    u64 cur = ukeys[u];
    int s = 0;
    if (u > 0) {
        u64 prev = ukeys[u - 1];
        if ((prev >> sh) == (cur >> sh) && (!lens || lens[u - 1] == len)) s = enc_shared(prev, cur, l);
    }
    uint8_t *p = out + enc_off[u];
    *p++ = (uint8_t)s;
    if (s == 0) *p++ = (uint8_t)len;
    for (int i = (s == 0 ? 0 : s - 1); i < l; i++) *p++ = (uint8_t)(cur >> (56 - 8 * i));
    const u64 a = val_off[u], b = val_off[u + 1];
    const u64 nv = b - a;
    if (kind == KMG_PAYLOAD_TAXON) {
        *p++ = (uint8_t)(nv >> 8);
        *p++ = (uint8_t)nv;
        for (u64 v = a; v < b; v++) {
            u32 t = (u32)val_a[v];
            int32_t c = val_b[v];
            *p++ = (uint8_t)(t >> 24); *p++ = (uint8_t)(t >> 16); *p++ = (uint8_t)(t >> 8); *p++ = (uint8_t)t;
            *p++ = (uint8_t)(c > 255 ? 255 : c);   // CountDataType.java:40
        }
    } else {
        *p++ = (uint8_t)(nv >> 24); *p++ = (uint8_t)(nv >> 16); *p++ = (uint8_t)(nv >> 8); *p++ = (uint8_t)nv;
        for (u64 v = a; v < b; v++) {
            u32 r = (u32)val_a[v];
            u32 q = (u32)val_b[v];
            *p++ = (uint8_t)(r >> 24); *p++ = (uint8_t)(r >> 16); *p++ = (uint8_t)(r >> 8); *p++ = (uint8_t)r;
            *p++ = (uint8_t)(q >> 8); *p++ = (uint8_t)q;
        }
    }
}

__global__ void enc_blockoff_kernel(const u64 *__restrict__ bucket_off, const u64 *__restrict__ enc_off, u32 nbuckets,
                                    u64 *__restrict__ block_off) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nbuckets) block_off[b] = enc_off[bucket_off[b]];
}


// ---- merge of databases (DB.kmers over several files, Database/DB.java:99-109: k-way merge of the files'
// ascending k-mer streams, equal k-mers reduced with addAll -- CountMap counts add, read sets unite) --------
// Expand one CSR into (key, payload) records appended at out[base ..): one thread per value.
//   taxon DB : payload = (taxon ^ sign bit) << 32 | count      (sorted by the upper half inside a k-mer)
//   read DB  : payload = read << 16 | pos16                     (sorted whole inside a k-mer)
__global__ void mg_expand_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ val_off, u64 U, u64 V,
                                 const int32_t *__restrict__ val_a, const int32_t *__restrict__ val_b,
                                 const u64 *__restrict__ packed, int kind, u64 *__restrict__ out_keys,
                                 u64 *__restrict__ out_vals, u64 base) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    u64 lo = 0, hi = U;                       // last k-mer whose list starts at or before v
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (val_off[mid] <= v) lo = mid; else hi = mid;
    }
    out_keys[base + v] = keys[lo];
    u64 pay;
    if (kind == KMG_PAYLOAD_TAXON) pay = ((u64)((u32)val_a[v] ^ 0x80000000u) << 32) | (u32)val_b[v];
    else pay = packed ? packed[v] : (((u64)(u32)val_a[v] << 16) | (u64)((u32)val_b[v] & 0xFFFFu));
    out_vals[base + v] = pay;
}

// sorted (key, taxon, count) records: head flags of k-mers and of (k-mer, taxon) pairs
__global__ void mg_flags_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ vals, u64 n,
                                u32 *__restrict__ fk, u32 *__restrict__ fp) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool hk = i == 0 || keys[i] != keys[i - 1];
    fk[i] = hk;
    fp[i] = hk || (vals[i] >> 32) != (vals[i - 1] >> 32);
}
// pk / pp = exclusive scans of the flags.  Counts of equal pairs add (saturating at INT32_MAX: the CSR holds
// int32 counts; the reference adds longs and saturates at 255 only when it writes the byte).
__global__ void mg_write_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ vals, u64 n,
                                const u32 *__restrict__ fk, const u32 *__restrict__ fp, const u64 *__restrict__ pk,
                                const u64 *__restrict__ pp, u64 *__restrict__ ukeys, u64 *__restrict__ val_off,
                                int32_t *__restrict__ val_a, u32 *__restrict__ val_b) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 pair = pp[i] + fp[i] - 1;
    if (fk[i]) { ukeys[pk[i]] = keys[i]; val_off[pk[i]] = pair; }
    if (fp[i]) val_a[pair] = (int32_t)((u32)(vals[i] >> 32) ^ 0x80000000u);
    atomicAdd(&val_b[pair], (u32)vals[i]);   // clamped to INT32_MAX afterwards (mg_clamp_kernel)
}
__global__ void mg_clamp_kernel(u32 *__restrict__ val_b, const u64 *__restrict__ totals) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < totals[1] && val_b[i] > 0x7FFFFFFFu) val_b[i] = 0x7FFFFFFFu;
}
