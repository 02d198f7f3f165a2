// extract_part.h -- stage 1 fused with the first partition pass of the sort.
//
// For the first batch of a build the extractor does not write its records in window order for a later
// histogram + scatter pass to read back; it scatters them straight into the order of the sort's first
// (lowest) prefix digit.  That removes one full read + write of every record and one histogram read:
//   count pass : per extraction tile, the digit histogram of its forward (and RC) records -> table
//   scan       : digit-major exclusive scan of the table (scan.h)
//   write pass : windows again; stable ranks by wave-level digit matching (as rs_scatter_kernel), records
//                staged through LDS in digit order, runs stored contiguously at the scanned offsets.
// Forward and RC records of a tile form two table columns (2t, 2t+1), so a sub-pass never stages more than
// EX_TILE records; both are emitted in window order, which keeps the partition stable.
// Same window logic as extract_kernel (extract.h); a wave owns 64 * windows-per-thread consecutive positions.
#pragma once
#include "common.h"
#include "extract.h"
#include "radix.h"

#define EP_BLOCK 512
#define EP_WAVES (EP_BLOCK / 64)
#define EP_ROUNDS (EX_TILE / EP_BLOCK)   // 8 windows per thread
#ifndef EP_SQCAP
#define EP_SQCAP 96                      // sequences of a tile whose payload data is cached in LDS
#endif

struct PartArgs {
    ExtractArgs e;
    int dshift;        // the partition digit: (key >> dshift) & dmask
    u32 dmask;
    u32 *table;        // [digit][column] counts (count pass) / exclusive offsets (write pass)
    u32 ncols;         // columns = tiles * (rc ? 2 : 1)
    u32 ntiles;
    u32 out_base;      // records already in the buffers (0: the fused path is the first batch of a build)
};

// the k-mer starting at tile position p, from the packed LDS image (3 words that the whole wave shares)
__device__ __forceinline__ u64 ep_window(const u32 *pk, u32 p, u64 kmask) {
    const u32 w = p >> 4, r = p & 15;
    const u64 hi = ((u64)pk[w] << 32) | pk[w + 1];
    return ((hi << (2 * r)) | ((u64)pk[w + 2] >> (32 - 2 * r))) & kmask;
}

// OR of x << j for j = 0 .. n-1 (1 <= n <= 32), by doubling: bit p of the result (MSB first) says whether any of the
// n positions p .. p+n-1 is set in x
__device__ __forceinline__ u64 ep_or_span(u64 x, int n) {
    int have = 1;
    while (have * 2 <= n) { x |= x << have; have *= 2; }
    if (have < n) x |= x << (n - have);
    return x;
}

// seqrel is kept in 16 bits; a tile that sees 65535 or more sequence starts (tens of thousands of EMPTY
// sequences inside 4 K bases) stores the clamp value and those positions look their sequence up instead:
// the sequence of the base at absolute position g is the largest s with seq_off[s] <= g.
#define EP_REL_CLAMP 0xFFFFu
__device__ __forceinline__ u32 ep_seq_rel(const ExtractArgs &a, u32 s_lo, u32 s_end, u64 g) {
    u64 lo = s_lo, hi = (u64)s_end + 1;   // first index in (s_lo, s_end] with seq_off > g
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (a.seq_off[mid] <= g) lo = mid; else hi = mid;
    }
    return (u32)(lo - s_lo);
}

// Register budget: the write pass keeps only validity masks and ranks per window and re-reads the k-mer
// from the LDS image where it needs it (broadcast reads), which fits 3 workgroups per CU (<= 80 VGPRs,
// ~52 KB of LDS: LDS is handed out in 1280-byte granules on gfx950, 42 granules = 53,760 B is the most a
// workgroup may take for three to fit -- 54,160 B measured as two per CU and 10 % slower).
template <bool WRITE, typename V>
__global__ __launch_bounds__(EP_BLOCK, WRITE ? 6 : 8) void extract_part_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 inv[EX_TILE / 32 + 2];
    __shared__ u32 stb[EX_TILE / 32 + 2];
    __shared__ uint16_t seqrel[EX_TILE + EX_EXT];      // #sequences (beyond the tile's first) starting at or before p
    __shared__ u32 cnt32[WRITE ? 1 : EX_TILE + EX_EXT]; // sequence starts per position (write pass: inside `stage`)
    __shared__ u32 hist[2][WRITE ? 1 : RS_DIGITS];     // count pass: forward / RC digit histograms
    __shared__ uint16_t wcnt[WRITE ? EP_WAVES : 1][RS_DIGITS];
    __shared__ uint16_t dstart[RS_DIGITS];             // <= EX_TILE
    __shared__ u32 gdelta[RS_DIGITS];
    __shared__ u64 stage[WRITE ? EX_TILE : 1];
    __shared__ uint8_t sdig[WRITE ? EX_TILE : 1];
    __shared__ u32 scan_smem[EP_WAVES];
    __shared__ u32 s_tot[2];                           // count pass: forward / RC records of the tile
    // write pass: payload source of the tile's first EP_SQCAP sequences (taxon | read index) and the low 32 bits
    // of t0 - seq_off[s], so that building the payloads needs no global load (later sequences: global path)
    __shared__ u32 sq_val[WRITE ? EP_SQCAP : 1];
    __shared__ u32 sq_neg[WRITE ? EP_SQCAP : 1];
    u32 *starts = WRITE ? reinterpret_cast<u32 *>(stage) : cnt32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (see rs_scatter_kernel): neighbouring tiles' runs are adjacent in the output
    const u32 per_xcd = (pa.ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= pa.ntiles) return;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 col0 = tile * (a.rc ? 2u : 1u);
    // Global loads that later phases depend on are issued up front, so that their latency overlaps the packing:
    // the tile's sequence range, this thread's sequence start, the scanned table entries of digit `tid`.
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const u64 my_s = (u64)s_lo + 1 + tid;
    const u64 my_start = my_s <= s_end ? a.seq_off[my_s] : 0;
    u32 tbl0 = 0, tbl1 = 0, out_base = 0;
    if (WRITE) {
        out_base = pa.out_base;
        if (tid < RS_DIGITS) {
            tbl0 = pa.table[(u64)tid * pa.ncols + col0];
            if (a.rc) tbl1 = pa.table[(u64)tid * pa.ncols + col0 + 1];
        }
        if (tid < EP_SQCAP && (u64)s_lo + tid <= s_end) {
            const u64 sq = (u64)s_lo + tid;
            sq_val[tid] = a.kind == KMG_PAYLOAD_TAXON ? (u32)a.seq_id[sq] : (u32)a.read_index[sq];
            sq_neg[tid] = (u32)(t0 - a.seq_off[sq]);
        }
    }

    // ---- 1. load + pack (as extract_kernel) --------------------------------------------------------
    const bool aligned = (((uintptr_t)a.bases) & 15) == 0;
    for (int chunk = tid; chunk < (EX_TILE + EX_EXT) / 16; chunk += EP_BLOCK) {
        u64 g = t0 + (u64)chunk * 16;
        u32 q[4];
        if (aligned && g + 16 <= a.nbases) {
            uint4 v = *reinterpret_cast<const uint4 *>(a.bases + g);
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u32 x = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    u64 gi = g + i * 4 + b;
                    u32 c = gi < a.nbases ? a.bases[gi] : 0u;
                    x |= c << (8 * b);
                }
                q[i] = x;
            }
        }
        u32 word = 0, bad = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            u32 c = (q[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            u32 up = c & 0xDFu;
            u32 ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
            word = (word << 2) | ((c >> 1) & 3u);
            bad = (bad << 1) | (ok ^ 1u);
        }
        pk[chunk] = word;
        reinterpret_cast<uint16_t *>(inv)[chunk ^ 1] = (uint16_t)bad;
    }
    for (int i = tid; i < EX_TILE + EX_EXT; i += EP_BLOCK) starts[i] = 0;
    if (tid < RS_DIGITS) {
        if (WRITE) {
#pragma unroll
            for (int w = 0; w < EP_WAVES; w++) wcnt[w][tid] = 0;
        } else {
            hist[0][tid] = 0;
            hist[1][tid] = 0;
            if (tid < 2) s_tot[tid] = 0;
        }
    }
    __syncthreads();

    // ---- 2. sequence starts -> per-position sequence index ------------------------------------------
    if (my_s <= s_end) atomicAdd(&starts[(u32)(my_start - t0)], 1u);
    for (u64 s = my_s + EP_BLOCK; s <= s_end; s += EP_BLOCK) atomicAdd(&starts[(u32)(a.seq_off[s] - t0)], 1u);
    __syncthreads();
    {
        // blocked: thread owns positions 8*tid .. 8*tid+7 = byte `tid` of the MSB-first start bitmap
        u32 c[8];
        u32 sum = 0, bits = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c[i] = starts[tid * 8 + i];
            bits = (bits << 1) | (c[i] != 0);
            sum += c[i];
        }
        reinterpret_cast<uint8_t *>(stb)[(tid & ~3) + (3 - (tid & 3))] = (uint8_t)bits;
        if (tid < EX_EXT / 8) {
            u32 eb = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) eb = (eb << 1) | (starts[EX_TILE + tid * 8 + i] != 0);
            const int bn = EX_TILE / 8 + tid;
            reinterpret_cast<uint8_t *>(stb)[(bn & ~3) + (3 - (bn & 3))] = (uint8_t)eb;
        }
        u32 total;
        u32 incl = block_scan_inclusive<u32, OpSum, EP_WAVES>(sum, OpSum(), scan_smem, &total);
        u32 run = incl - sum;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            run += c[i];
            seqrel[tid * 8 + i] = (uint16_t)(run < EP_REL_CLAMP ? run : EP_REL_CLAMP);
        }
    }
    // Window validity for the whole tile, 32 positions per thread and bit-parallel: a window [p, p+k) is good when
    // it holds no invalid base, no sequence starts at p+1 .. p+k-1, and it ends inside the input.  (stb is complete:
    // the scan above had barriers after its last store.)  The bitmap replaces `inv`, which nothing reads afterwards.
    u32 vw = 0;
    if (tid < EX_TILE / 32) {
        const u64 B = ((u64)inv[tid] << 32) | inv[tid + 1];
        const u64 S = ((u64)stb[tid] << 32) | stb[tid + 1];
        u64 bad_any = ep_or_span(B, k);
        if (k > 1) bad_any |= ep_or_span(S << 1, k - 1);
        const long long last = (long long)a.nbases - k - (long long)t0 - 32ll * tid;   // last good position of this word
        const u32 in_range = last < 0 ? 0u : (last >= 31 ? ~0u : ~(~0u >> (u32)(last + 1)));
        vw = ~(u32)(bad_any >> 32) & in_range;
    }
    __syncthreads();
    if (tid < EX_TILE / 32) inv[tid] = vw;
    __syncthreads();

    // ---- 3. windows: a wave owns positions wave*512 .. +511, round j covers 64 consecutive ones ------
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    const u32 pbase = wave * (64 * EP_ROUNDS) + lane;
    u32 vmask = 0, rmask = 0, nfwd = 0, nrc = 0;
    // (not unrolled in the write pass: the eight copies of the window logic with both filters inlined cost
    // registers the rank and staging steps need -- spills there are real memory traffic)
#pragma unroll WRITE ? 1 : EP_ROUNDS
    for (int j = 0; j < EP_ROUNDS; j++) {
        const u32 p = pbase + j * 64;
        const u64 key = ep_window(pk, p, kmask);
        bool ok = (inv[p >> 5] >> (31 - (p & 31))) & 1u;
        if (!WRITE && ok && a.has_kmer) {
            u32 rel = seqrel[p];
            if (rel == EP_REL_CLAMP) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
            a.has_kmer[s_lo + rel] = 1;
        }
        if (ok && a.runofsame >= 0) ok = runofsame_keep(key, k, a.runofsame);
        if (ok && a.dust >= 0) ok = dust_keep(key, k, a.dust);
        bool emit_rc = false;
        u64 rk = 0;
        if (a.rc && ok) {
            const u64 rev = revpairs64(key >> (64 - 2 * k));
            emit_rc = !(((k & 1) == 0) && rev == key);
            rk = rev ^ rcx;
        }
        if (a.ranged) {
            const u32 fb = (u32)(key >> a.range_shift), rb = (u32)(rk >> a.range_shift);
            emit_rc = emit_rc && rb >= a.range_lo && rb < a.range_hi;
            ok = ok && fb >= a.range_lo && fb < a.range_hi;
        }
        vmask |= (u32)ok << j;
        rmask |= (u32)emit_rc << j;
        if (!WRITE) {
            if (ok) atomicAdd(&hist[0][(u32)(key >> pa.dshift) & pa.dmask], 1u);
            if (emit_rc) atomicAdd(&hist[1][(u32)(rk >> pa.dshift) & pa.dmask], 1u);
            nfwd += __popcll(__ballot(ok));          // wave-uniform running totals
            nrc += __popcll(__ballot(emit_rc));
        }
    }
    if (!WRITE && lane == 0) {
        atomicAdd(&s_tot[0], nfwd);
        atomicAdd(&s_tot[1], nrc);
    }
    __syncthreads();

    if (!WRITE) {
        // histogram columns + the tile's record / forward-window totals
        u32 hf = 0, hr = 0;
        if (tid < RS_DIGITS) {
            hf = hist[0][tid];
            hr = hist[1][tid];
            pa.table[(u64)tid * pa.ncols + col0] = hf;
            if (a.rc) pa.table[(u64)tid * pa.ncols + col0 + 1] = hr;
        }
        if (tid == 0) {
            a.tile_count[tile] = s_tot[0] + s_tot[1];
            a.tile_fwd[tile] = s_tot[0];
        }
        return;
    }

    V *out_vals = reinterpret_cast<V *>(a.out_vals);
    for (int sp = 0; sp < (a.rc ? 2 : 1); sp++) {
        const u32 imask = sp == 0 ? vmask : rmask;
        // opaque copy of the position base: keeps per-window address arithmetic inside the loop body
        // (hoisted out of the sub-pass loop it would sit in 16+ registers across every barrier)
        u32 pb = pbase;
        asm volatile("" : "+v"(pb));
        u32 lpos[EP_ROUNDS];   // digit << 16 | slot
        // ---- stable rank of every record among the tile's records with the same digit
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            const bool valid = (imask >> j) & 1;
            u64 key = ep_window(pk, pb + j * 64, kmask);
            if (sp) key = revpairs64(key >> (64 - 2 * k)) ^ rcx;
            const u32 d = (u32)(key >> pa.dshift) & pa.dmask;
            u32 below, total;
            wave_match(d, valid, below, total);
            const u32 before = wcnt[wave][d];
            lpos[j] = (d << 16) | (before + below);
            if (valid && below == 0) wcnt[wave][d] = (uint16_t)(before + total);
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        u32 tile_n;
        u32 tdx = tid;                      // opaque copy, as pb: no addresses hoisted out of the loop
        asm volatile("" : "+v"(tdx));
        {
            u32 run = 0;
            if (tdx < RS_DIGITS) {
#pragma unroll
                for (int w = 0; w < EP_WAVES; w++) {
                    u32 c = wcnt[w][tdx];
                    wcnt[w][tdx] = (uint16_t)run;
                    run += c;
                }
            }
            u32 incl = block_scan_inclusive<u32, OpSum, EP_WAVES>(run, OpSum(), scan_smem, &tile_n);
            if (tdx < RS_DIGITS) {
                const u32 start = incl - run;
                dstart[tdx] = (uint16_t)start;
                gdelta[tdx] = out_base + (sp ? tbl1 : tbl0) - start;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            if ((imask >> j) & 1) {
                u64 key = ep_window(pk, pb + j * 64, kmask);
                if (sp) key = revpairs64(key >> (64 - 2 * k)) ^ rcx;
                const u32 d = lpos[j] >> 16;
                const u32 slot = (lpos[j] & 0xFFFFu) + dstart[d] + wcnt[wave][d];
                lpos[j] = slot;
                stage[slot] = key;
                sdig[slot] = (uint8_t)d;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < EP_ROUNDS; i++) {
            const u32 lp = i * EP_BLOCK + tid;
            if (lp < tile_n) a.out_keys[lp + gdelta[sdig[lp]]] = stage[lp];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            if ((imask >> j) & 1) {
                // payload of the window (the sequence tables are tiny and cached: a tile spans few sequences)
                const u32 p = pb + j * 64;
                u32 rel = seqrel[p];
                if (rel == EP_REL_CLAMP) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
                u32 sv, neg;
                if (rel < EP_SQCAP) { sv = sq_val[rel]; neg = sq_neg[rel]; }
                else {
                    const u64 sq = (u64)s_lo + rel;
                    sv = a.kind == KMG_PAYLOAD_TAXON ? (u32)a.seq_id[sq] : (u32)a.read_index[sq];
                    neg = (u32)(t0 - a.seq_off[sq]);
                }
                V v;
                if (a.kind == KMG_PAYLOAD_TAXON) v = (V)sv;
                else v = (V)(((u64)sv << a.pos_bits) | (u64)((p + neg) & a.pos_mask));
                reinterpret_cast<V *>(stage)[lpos[j]] = v;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < EP_ROUNDS; i++) {
            const u32 lp = i * EP_BLOCK + tid;
            if (lp < tile_n) out_vals[lp + gdelta[sdig[lp]]] = reinterpret_cast<V *>(stage)[lp];
        }
        __syncthreads();
        if (sp == 0 && a.rc && tid < RS_DIGITS) {
#pragma unroll
            for (int w = 0; w < EP_WAVES; w++) wcnt[w][tid] = 0;
        }
        __syncthreads();
    }
}
