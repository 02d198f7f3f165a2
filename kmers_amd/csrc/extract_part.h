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

// The count pass leaves every tile's prepared image behind -- packed bases, sequence-start bitmap, and per thread
// the "this window is a record" bits after validity, filters and key range -- so that the write pass neither
// loads nor packs the bases again, nor repeats the sequence-start scan, the validity bitmap, or the -D / -R
// predicates: it starts at the ranking step.  2.6 KB per 4 KB of bases, written once and read once.
#define EP_IMG_PK 0                                   // EX_TILE / 16 + 4 packed words
#define EP_IMG_STB (EX_TILE / 16 + 4)                 // EX_TILE / 32 + 2 start-bitmap words (MSB first), padded to + 4
#define EP_IMG_VM (EP_IMG_STB + EX_TILE / 32 + 4)     // EP_BLOCK bytes: forward-record bits of thread tid (bit j = round j)
#define EP_IMG_RM (EP_IMG_VM + EP_BLOCK / 4)          // EP_BLOCK bytes: reverse-complement-record bits
#define EP_IMG_FLAG (EP_IMG_RM + EP_BLOCK / 4)        // 1 = some position starts several sequences: look sequences up
#define EP_IMG_WORDS (EP_IMG_FLAG + 4)

struct PartArgs {
    ExtractArgs e;
    u32 *img;          // [tile][EP_IMG_WORDS]
    int dshift;        // the partition digit: (key >> dshift) & dmask
    u32 dmask;
    u32 *table;        // [digit][column] counts (count pass) / exclusive offsets (write pass)
    u32 ncols;         // columns = tiles * (rc ? 2 : 1)
    u32 ntiles;
    u32 out_base;      // records already in the buffers (0: the fused path is the first batch of a build)
};

// the k-mer starting at tile position p, from the packed LDS image (3 words that the whole wave shares)
// (the write pass keeps the 64-bit form: the funnel-shift form costs it registers it does not have -- 6 spills, +13 %)
__device__ __forceinline__ u64 ep_window64(const u32 *pk, u32 p, u64 kmask) {
    const u32 w = p >> 4, r = p & 15;
    const u64 hi = ((u64)pk[w] << 32) | pk[w + 1];
    return ((hi << (2 * r)) | ((u64)pk[w + 2] >> (32 - 2 * r))) & kmask;
}
// upper word of the k-mer starting at p (its first 16 bases): all the partition digit of a forward record needs
__device__ __forceinline__ u32 ep_window_hi(const u32 *pk, u32 p, u64 kmask) {
    const u32 w = p >> 4;
    return __funnelshift_l(pk[w + 1], pk[w], 2 * (p & 15)) & (u32)(kmask >> 32);
}
// (two 32-bit funnel shifts: 64-bit shifts by a lane-varying amount run at a quarter of the rate)
__device__ __forceinline__ u64 ep_window(const u32 *pk, u32 p, u64 kmask) {
    const u32 w = p >> 4, r2 = 2 * (p & 15);
    const u32 a = pk[w], b = pk[w + 1], c = pk[w + 2];
    const u32 hi = __funnelshift_l(b, a, r2), lo = __funnelshift_l(c, b, r2);
    return (((u64)hi << 32) | lo) & kmask;
}

// OR of x << j for j = 0 .. n-1 (1 <= n <= 32), by doubling: bit p of the result (MSB first) says whether any of the
// n positions p .. p+n-1 is set in x
__device__ __forceinline__ u64 ep_or_span(u64 x, int n) {
    int have = 1;
    while (have * 2 <= n) { x |= x << have; have *= 2; }
    if (have < n) x |= x << (n - have);
    return x;
}

// The sequence of the base at absolute position g is the largest s with seq_off[s] <= g.  Tiles count the start
// bits at or before a position instead (wpre + popcount); this search is for tiles in which several sequences start
// at one position (empty sequences), where bits cannot count them.
__device__ __forceinline__ u32 ep_seq_rel(const ExtractArgs &a, u32 s_lo, u32 s_end, u64 g) {
    u64 lo = s_lo, hi = (u64)s_end + 1;   // first index in (s_lo, s_end] with seq_off > g
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (a.seq_off[mid] <= g) lo = mid; else hi = mid;
    }
    return (u32)(lo - s_lo);
}

// ---- count pass: everything up to "which windows are records", the digit histograms, and the tile image -------
// RC: reverse-complement records asked for.  FILT: -D / -R filters or a key range are on (else none of that code is
// even compiled in: the common path is branch-free per window).
template <bool RC, bool FILT>
__global__ __launch_bounds__(EP_BLOCK, 8) void extract_part_count_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 inv[EX_TILE / 32 + 2];
    __shared__ u32 stb[EX_TILE / 32 + 2];              // "a sequence starts here" bitmap, MSB first
    __shared__ u32 wpre[EX_TILE / 32];                 // sequence starts before bitmap word w
    __shared__ u32 hist[2][RS_DIGITS];                 // forward / RC digit histograms
    __shared__ u32 s_tot[2];                           // forward / RC records of the tile
    __shared__ u32 s_dup;                              // some position starts more than one sequence (empty sequences)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (see rs_scatter_kernel): neighbouring tiles' runs are adjacent in the output
    const u32 per_xcd = (pa.ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= pa.ntiles) return;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 col0 = tile * (a.rc ? 2u : 1u);
    // Global loads that later phases depend on are issued up front, so that their latency overlaps the packing
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const u64 my_s = (u64)s_lo + 1 + tid;
    const u64 my_start = my_s <= s_end ? a.seq_off[my_s] : 0;

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
            // A C G T (either case) = 0x41 + {0, 2, 6, 19}: one look into a 32-bit set instead of four compares
            const u32 t = (c & 0xDFu) - 0x41u;
            const u32 ok = (0x00080045u >> (t < 31u ? t : 31u)) & 1u;
            word = (word << 2) | ((c >> 1) & 3u);
            bad = (bad << 1) | (ok ^ 1u);
        }
        pk[chunk] = word;
        reinterpret_cast<uint16_t *>(inv)[chunk ^ 1] = (uint16_t)bad;
    }
    if (tid < RS_DIGITS) {
        hist[0][tid] = 0;
        hist[1][tid] = 0;
        if (tid < EX_TILE / 32 + 2) stb[tid] = 0;
        if (tid < 2) s_tot[tid] = 0;
        if (tid == 2) s_dup = 0;
    }
    __syncthreads();

    // ---- 2. sequence starts -> bitmap; the sequence of position p = the tile's first + the starts at or before p ---
    // (a start that finds its bit set already shares its position with another one -- an empty sequence: such
    // tiles look every sequence up by binary search instead of counting bits)
    for (u64 sq = my_s; sq <= s_end; sq += EP_BLOCK) {
        const u32 pos = (u32)((sq == my_s ? my_start : a.seq_off[sq]) - t0);
        const u32 bit = 0x80000000u >> (pos & 31);
        if (atomicOr(&stb[pos >> 5], bit) & bit) s_dup = 1;
    }
    __syncthreads();
    if (wave == EP_WAVES - 1) {
        const u32 c0 = __popc(stb[2 * lane]), c1 = __popc(stb[2 * lane + 1]);
        u32 inc = c0 + c1;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        wpre[2 * lane] = inc - c0 - c1;
        wpre[2 * lane + 1] = inc - c1;
    }
    // Window validity for the whole tile, 32 positions per thread and bit-parallel: a window [p, p+k) is good when
    // it holds no invalid base, no sequence starts at p+1 .. p+k-1, and it ends inside the input.  The bitmap replaces
    // `inv`, which nothing reads afterwards.
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
    const bool dup = s_dup != 0;
    // A sequence's first k-mer (of this tile) sits behind a window that is none: a window across its start (k > 1), one
    // with an invalid base, or the tile's edge -- marking there marks every sequence that has a k-mer.  Those windows
    // are one in a read's length: the threads that hold the validity words walk their few set bits, and the window loop
    // below carries none of this.
    if (a.has_kmer && tid < EX_TILE / 32) {
        u32 fw = vw;
        if (k > 1) fw &= ~((vw >> 1) | (tid ? (inv[tid - 1] << 31) : 0u));
        while (fw) {
            const u32 b = (u32)__clz((int)fw), p = 32u * tid + b;
            u32 rel;
            if (dup) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
            else rel = wpre[p >> 5] + __popc(stb[p >> 5] >> (31 - (p & 31)));
            a.has_kmer[s_lo + rel] = 1;
            fw &= ~(0x80000000u >> b);
        }
    }

    // ---- 3. windows: a wave owns positions wave*512 .. +511, round j covers 64 consecutive ones ------
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    const u32 pbase = wave * (64 * EP_ROUNDS) + lane;
    u32 vmask = 0, rmask = 0, nfwd = 0, nrc = 0;
#pragma unroll
    for (int j = 0; j < EP_ROUNDS; j++) {
        const u32 p = pbase + j * 64;
        // (without reverse complements or filters only the digit is needed here: the upper word)
        const u64 key = (RC || FILT) ? ep_window(pk, p, kmask) : ((u64)ep_window_hi(pk, p, kmask) << 32);
        bool ok = (inv[p >> 5] >> (31 - (p & 31))) & 1u;
        if (FILT) {
            if (ok && a.runofsame >= 0) ok = runofsame_keep(key, k, a.runofsame);
            if (ok && a.dust >= 0) ok = dust_keep(key, k, a.dust);
        }
        bool emit_rc = false;
        u64 rk = 0;
        if (RC) {
            const u64 rev = revpairs64(key >> (64 - 2 * k));
            emit_rc = ok && !(((k & 1) == 0) && rev == key);
            rk = rev ^ rcx;
        }
        if (FILT && a.ranged) {
            const u32 fb = (u32)(key >> a.range_shift), rb = (u32)(rk >> a.range_shift);
            emit_rc = emit_rc && rb >= a.range_lo && rb < a.range_hi;
            ok = ok && fb >= a.range_lo && fb < a.range_hi;
        }
        vmask |= (u32)ok << j;
        if (RC) rmask |= (u32)emit_rc << j;
        // (the partition digit lies in the key's upper word: dshift >= 32 for every plan, see choose_pbits)
        if (ok) atomicAdd(&hist[0][((u32)(key >> 32) >> (pa.dshift - 32)) & pa.dmask], 1u);
        nfwd += __popcll(__ballot(ok));          // wave-uniform running totals
        if (RC) {
            if (emit_rc) atomicAdd(&hist[1][((u32)(rk >> 32) >> (pa.dshift - 32)) & pa.dmask], 1u);
            nrc += __popcll(__ballot(emit_rc));
        }
    }
    if (lane == 0) {
        atomicAdd(&s_tot[0], nfwd);
        atomicAdd(&s_tot[1], nrc);
    }
    // the tile image for the write pass
    u32 *img = pa.img + (u64)tile * EP_IMG_WORDS;
    if (tid < (EX_TILE + EX_EXT) / 16) img[EP_IMG_PK + tid] = pk[tid];
    if (tid < EX_TILE / 32 + 2) img[EP_IMG_STB + tid] = stb[tid];
    reinterpret_cast<uint8_t *>(img + EP_IMG_VM)[tid] = (uint8_t)vmask;
    if (RC) reinterpret_cast<uint8_t *>(img + EP_IMG_RM)[tid] = (uint8_t)rmask;
    __syncthreads();

    // histogram columns + the tile's record / forward-window totals
    if (tid < RS_DIGITS) {
        pa.table[(u64)tid * pa.ncols + col0] = hist[0][tid];
        if (RC) pa.table[(u64)tid * pa.ncols + col0 + 1] = hist[1][tid];
    }
    if (tid == 0) {
        a.tile_count[tile] = s_tot[0] + s_tot[1];
        a.tile_fwd[tile] = s_tot[0];
        img[EP_IMG_FLAG] = dup ? 1u : 0u;
    }
}

// ---- write pass: from the tile image straight to ranking, staging and the partitioned stores ---------------------
// Register budget: only the record bits and ranks live per window; the k-mer is re-read from the LDS image where it
// is needed (broadcast reads), which fits 3 workgroups per CU (<= 80 VGPRs, < 53,760 B of LDS: LDS is handed out in
// 1280-byte granules on gfx950).
template <typename V>
__global__ __launch_bounds__(EP_BLOCK, 6) void extract_part_write_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 stb[EX_TILE / 32 + 4];
    __shared__ u32 wpre[EX_TILE / 32];                 // sequence starts before bitmap word w
    __shared__ uint16_t wcnt[EP_WAVES][RS_DIGITS];
    __shared__ uint16_t dstart[RS_DIGITS];             // <= EX_TILE
    __shared__ u32 gdelta[RS_DIGITS];
    __shared__ u64 stage[EX_TILE];
    __shared__ uint8_t sdig[EX_TILE];
    __shared__ u32 scan_smem[EP_WAVES];
    // payload source of the tile's first EP_SQCAP sequences (taxon | read index) and the low 32 bits of
    // t0 - seq_off[s], so that building the payloads needs no global load (later sequences: global path)
    __shared__ u32 sq_val[EP_SQCAP];
    __shared__ u32 sq_neg[EP_SQCAP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 per_xcd = (pa.ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= pa.ntiles) return;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 col0 = tile * (a.rc ? 2u : 1u);
    const u32 *img = pa.img + (u64)tile * EP_IMG_WORDS;
    // everything the tile needs from global memory is asked for up front
    const u32 img_w = tid < EP_IMG_VM ? img[tid] : 0u;               // packed bases, then the start bitmap
    const u32 vmask = reinterpret_cast<const uint8_t *>(img + EP_IMG_VM)[tid];
    const u32 rmask = a.rc ? reinterpret_cast<const uint8_t *>(img + EP_IMG_RM)[tid] : 0u;
    const bool dup = img[EP_IMG_FLAG] != 0;
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const bool lds_only = !dup && s_end - s_lo < EP_SQCAP;             // every sequence of the tile is in sq_val / sq_neg
    u32 tbl0 = 0, tbl1 = 0;
    const u32 out_base = pa.out_base;
    if (tid < RS_DIGITS) {
        tbl0 = pa.table[(u64)tid * pa.ncols + col0];
        if (a.rc) tbl1 = pa.table[(u64)tid * pa.ncols + col0 + 1];
#pragma unroll
        for (int w = 0; w < EP_WAVES; w++) wcnt[w][tid] = 0;
    }
    if (tid < EP_SQCAP && (u64)s_lo + tid <= s_end) {
        const u64 sq = (u64)s_lo + tid;
        sq_val[tid] = a.kind == KMG_PAYLOAD_TAXON ? (u32)a.seq_id[sq] : (u32)a.read_index[sq];
        sq_neg[tid] = (u32)(t0 - a.seq_off[sq]);
    }
    if (tid < EP_IMG_STB) pk[tid] = img_w;
    else if (tid < EP_IMG_VM) stb[tid - EP_IMG_STB] = img_w;
    __syncthreads();
    if (wave == 0) {
        // starts before each bitmap word: lane owns words 2*lane, 2*lane + 1
        const u32 c0 = __popc(stb[2 * lane]), c1 = __popc(stb[2 * lane + 1]);
        u32 inc = c0 + c1;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        wpre[2 * lane] = inc - c0 - c1;
        wpre[2 * lane + 1] = inc - c1;
    }
    // (wpre is read in the payload step, behind the ranking step's barriers)

    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    const u32 pbase = wave * (64 * EP_ROUNDS) + lane;
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
            u32 khi;                                     // upper word of the record's key: the digit lies there
            if (sp) khi = (u32)((revpairs64(ep_window64(pk, pb + j * 64, kmask) >> (64 - 2 * k)) ^ rcx) >> 32);
            else khi = ep_window_hi(pk, pb + j * 64, kmask);
            const u32 d = (khi >> (pa.dshift - 32)) & pa.dmask;                  // dshift >= 32 (choose_pbits <= 24)
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
                u64 key = ep_window64(pk, pb + j * 64, kmask);
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
        // payload of the window: the window's sequence = the tile's first + the sequence starts at or before p.
        // Two copies of the loop behind a block-uniform test: the common one touches LDS only.  (A global load
        // anywhere in the loop, even on a path no lane takes, puts a wait for ALL outstanding vector memory
        // operations behind it -- the counter is shared with stores on gfx950 -- and the key stores just issued
        // would have to drain before the first payload is built.)
        if (lds_only) {
#pragma unroll
            for (int j = 0; j < EP_ROUNDS; j++) {
                if ((imask >> j) & 1) {
                    const u32 p = pb + j * 64;
                    const u32 rel = wpre[p >> 5] + __popc(stb[p >> 5] >> (31 - (p & 31)));
                    const u32 sv = sq_val[rel], neg = sq_neg[rel];
                    V v;
                    if (a.kind == KMG_PAYLOAD_TAXON) v = (V)sv;
                    else v = (V)(((u64)sv << a.pos_bits) | (u64)((p + neg) & a.pos_mask));
                    reinterpret_cast<V *>(stage)[lpos[j]] = v;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < EP_ROUNDS; j++) {
                if ((imask >> j) & 1) {
                    const u32 p = pb + j * 64;
                    u32 rel;
                    if (dup) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
                    else rel = wpre[p >> 5] + __popc(stb[p >> 5] >> (31 - (p & 31)));
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
