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
// A wave of the write pass owns 64 * windows-per-thread consecutive positions.
#pragma once
#include "common.h"
#include "extract.h"
#include "radix.h"
#include <type_traits>

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

// Four bases in the bytes of q (first base lowest) -> their 2-bit codes, first base highest (8 bits), and their "not one
// of ACGTacgt" bits the same way round (4 bits).  A C G T (either case) = 0x41 + {0, 2, 6, 19}: bits 2:1 of the letter
// are the code (A0 C1 T2 G3), and a byte is a base iff, without its case bit, it IS the letter of its code -- one byte
// permute looks the four letters up.  Bytes past the end of the input are 0: code 0, not a base.
__device__ __forceinline__ void ep_pack4(u32 q, u32 &code8, u32 &bad4) {
    const u32 c = (q >> 1) & 0x03030303u;
    const u32 diff = __builtin_amdgcn_perm(0u, 0x47544341u, c) ^ (q & 0xDFDFDFDFu);
    const u32 nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;   // 1 per byte that differs
    // gather towards the top byte; the copies left further down collide with nothing that reaches it
    u32 u = c | (c << 10);
    u |= u << 20;
    code8 = u >> 24;
    u32 b = nz | (nz << 9);
    b |= b << 18;
    bad4 = b >> 24;
}

// bit `bit` of each of the 8 bytes of v -> 8 bits, byte j's at bit j
__device__ __forceinline__ u32 ep_bit_column(uint2 v, u32 bit) {
    u32 x = (v.x >> bit) & 0x01010101u, y = (v.y >> bit) & 0x01010101u;
    x |= x >> 7; x |= x >> 14;
    y |= y >> 7; y |= y >> 14;
    return (x & 0xFu) | ((y & 0xFu) << 4);
}

// ---- count pass: everything up to "which windows are records", the digit histograms, and the tile image -------
// RC: reverse-complement records asked for.  FILT: -D / -R filters or a key range are on (else none of that code is
// even compiled in: the common path is branch-free per window).
// HIST: the partition digit's histograms are wanted (the fused first batch of a build).  Later batches append their
// records in window order (extract_img_write_kernel): they take everything else -- record bits, tile totals, image.
template <bool RC, bool FILT, bool HIST = true>
__global__ __launch_bounds__(EP_BLOCK, 8) void extract_part_count_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 inv[EX_TILE / 32 + 2];
    __shared__ u32 stb[EX_TILE / 32 + 2];              // "a sequence starts here" bitmap, MSB first
    __shared__ u32 wpre[EX_TILE / 32];                 // sequence starts before bitmap word w
    __shared__ u32 hist[2][RS_DIGITS];                 // forward / RC digit histograms
    __shared__ u32 s_tot[2];                           // forward / RC records of the tile
    __shared__ u32 s_dup;                              // some position starts more than one sequence (empty sequences)
    __shared__ __align__(8) uint8_t tbits[2][EP_BLOCK];   // record bits on their way from the lanes that found them to the image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (see rs_scatter_kernel): neighbouring tiles' runs are adjacent in the output
    const u32 per_xcd = (pa.ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= pa.ntiles) return;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 col0 = tile * (a.rc ? 2u : 1u);
    // ---- 1. load + pack: 8 bases per thread, four at a time inside one 32-bit word (ep_pack4) ---------
    // One chunk per thread; the EX_EXT bases past the tile are four more chunks, which threads 0-3 take as well.  Every
    // global load the tile needs is asked for before the first is waited for, the bases first: the whole tile -- not
    // every thread, not behind a branch (a second trip, or a load under `if`, is a second memory round trip on every
    // tile's critical path; the threads without a second chunk load their first one again).
    static_assert((EX_TILE + EX_EXT) / 8 <= EP_BLOCK + 4 && EX_TILE / 8 == EP_BLOCK, "chunks per thread");
    const bool fast = (((uintptr_t)a.bases) & 7) == 0 && t0 + EX_TILE + EX_EXT <= a.nbases;
    const int chunk_b = tid < EX_EXT / 8 ? tid + EP_BLOCK : tid;
    u32 qq[2][2];
    if (fast) {
        const uint2 v0 = *reinterpret_cast<const uint2 *>(a.bases + t0 + (u64)tid * 8);
        const uint2 v1 = *reinterpret_cast<const uint2 *>(a.bases + t0 + (u64)chunk_b * 8);
        qq[0][0] = v0.x; qq[0][1] = v0.y; qq[1][0] = v1.x; qq[1][1] = v1.y;
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const u64 g = t0 + (u64)(i ? chunk_b : tid) * 8;
            qq[i][0] = qq[i][1] = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const u64 gi = g + b;
                qq[i][0] |= (gi < a.nbases ? (u32)a.bases[gi] : 0u) << (8 * b);
                qq[i][1] |= (gi + 4 < a.nbases ? (u32)a.bases[gi + 4] : 0u) << (8 * b);
            }
        }
    }
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const u64 my_s = (u64)s_lo + 1 + tid;
    const u64 my_start = my_s <= s_end ? a.seq_off[my_s] : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int chunk = i ? chunk_b : tid;
        u32 c0, b0, c1, b1;
        ep_pack4(qq[i][0], c0, b0);
        ep_pack4(qq[i][1], c1, b1);
        reinterpret_cast<uint16_t *>(pk)[chunk ^ 1] = (uint16_t)((c0 << 8) | c1);      // base 16w at bits 31:30 of word w
        reinterpret_cast<uint8_t *>(inv)[chunk ^ 3] = (uint8_t)((b0 << 4) | b1);       // base 32w at bit 31 of word w
    }
    if (tid < RS_DIGITS) {
        hist[0][tid] = 0;
        hist[1][tid] = 0;
        if (tid < EX_TILE / 32 + 2) stb[tid] = 0;
        if (tid < 2) s_tot[tid] = 0;
        if (tid == 2) s_dup = 0;
    }
    __syncthreads();
#ifdef EPC_PROBE
    if (EPC_PROBE == 1) return;
#endif

    // ---- 2. sequence starts -> bitmap; the sequence of position p = the tile's first + the starts at or before p ---
    // (a start that finds its bit set already shares its position with another one -- an empty sequence: such
    // tiles look every sequence up by binary search instead of counting bits)
    for (u64 sq = my_s; sq <= s_end; sq += EP_BLOCK) {
        const u32 pos = (u32)((sq == my_s ? my_start : a.seq_off[sq]) - t0);
        const u32 bit = 0x80000000u >> (pos & 31);
        if (atomicOr(&stb[pos >> 5], bit) & bit) s_dup = 1;
    }
    __syncthreads();
#ifdef EPC_PROBE
    if (EPC_PROBE == 2) return;
#endif
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
#ifdef EPC_PROBE
    if (EPC_PROBE == 3) return;
#endif
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

    // ---- 3. windows.  Here a lane owns the EIGHT CONSECUTIVE positions from q on (a count does not care who sees which
    // window): their bases are 3 packed words read once and shifted so that the stretch starts at q -- window j's k-mer is
    // the stretch moved on by 2j bits, a constant --, their validity is one byte, and the reverse complements of all eight
    // are pieces of ONE reversed stretch (window j = its bases 48-j-k .. 48-j-1) instead of a 64-bit reversal each.
    static_assert(EP_ROUNDS == 8, "a lane's windows are one byte of the validity bitmap");
#ifdef EPC_PROBE
    if (EPC_PROBE == 4) return;
#endif
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    const u32 q = (u32)wave * (64 * EP_ROUNDS) + (u32)lane * EP_ROUNDS;
    u32 y0, y1, y2;
    {
        const u32 w = q >> 4, r2 = (q & 15) * 2;
        const u32 a0 = pk[w], a1 = pk[w + 1], a2 = pk[w + 2];
        y0 = __funnelshift_l(a1, a0, r2);
        y1 = __funnelshift_l(a2, a1, r2);
        y2 = a2 << r2;                                   // (q + 39 bases at most are looked at: 16 + 16 + 8)
    }
    const u32 okb = (inv[q >> 5] >> (24 - (q & 24))) & 0xFFu;          // bit 7 = position q
    u32 r0 = 0, r1 = 0, r2w = 0;                         // the stretch reversed and complemented, 48 bases
    if (RC) {
        r0 = __brev(y2); r1 = __brev(y1); r2w = __brev(y0);
        r0 = (((r0 >> 1) & 0x55555555u) | ((r0 & 0x55555555u) << 1)) ^ 0xAAAAAAAAu;
        r1 = (((r1 >> 1) & 0x55555555u) | ((r1 & 0x55555555u) << 1)) ^ 0xAAAAAAAAu;
        r2w = (((r2w >> 1) & 0x55555555u) | ((r2w & 0x55555555u) << 1)) ^ 0xAAAAAAAAu;
    }
    u32 vmb = 0, rmb = 0;                                // bit j = window q + j is a forward / reverse-complement record
    // A record's partition digit and its key bucket lie in the k-mer's upper word (dshift >= 32 for every plan, see
    // choose_pbits; key lengths stop at 12 bases): the lower word is formed only where something reads it -- a -D / -R
    // filter, or the palindrome test of an even k --, and masks, compares and shifts are 32-bit ones (half the
    // instructions of their 64-bit forms; the shift itself is not the cost: tools/exp/shift64_rate.hip puts a 64-bit
    // shift by a wave-uniform amount at ~1.2 plain instructions).
    const u32 km_hi = (u32)(kmask >> 32), km_lo = (u32)kmask;
    const u32 rcx_hi = (u32)(rcx >> 32), rcx_lo = (u32)rcx;
    const bool k_even = (k & 1) == 0;
    const bool filters = FILT && (a.runofsame >= 0 || a.dust >= 0);
    const int bshift = a.range_shift - 32;
#pragma unroll
    for (int j = 0; j < EP_ROUNDS; j++) {
        const u32 hi = __funnelshift_l(y1, y0, 2 * j) & km_hi;
        u32 lo = 0;
        if ((RC && k_even) || filters) lo = __funnelshift_l(y2, y1, 2 * j) & km_lo;
        bool ok = (okb >> (7 - j)) & 1u;
        if (filters) {
            const u64 key = (((u64)hi) << 32) | lo;
            if (ok && a.runofsame >= 0) ok = runofsame_keep(key, k, a.runofsame);
            if (ok && a.dust >= 0) ok = dust_keep(key, k, a.dust);
        }
        bool emit_rc = false;
        u32 rh = 0;
        if (RC) {
            const int o = 48 - j - k;                    // first base of the window's reverse complement in r0:r1:r2w
            if (o < 16) rh = __funnelshift_l(r1, r0, 2 * o);
            else if (o < 32) rh = __funnelshift_l(r2w, r1, 2 * (o - 16));
            else rh = __funnelshift_l(0u, r2w, 2 * (o - 32));
            rh &= km_hi;
            emit_rc = ok;
            if (k_even) {                                // Kmer.isOwnRC (Kmer.java:210-225) is a palindrome test on even lengths
                u32 rl;
                if (o < 16) rl = __funnelshift_l(r2w, r1, 2 * o);
                else if (o < 32) rl = __funnelshift_l(0u, r2w, 2 * (o - 16));
                else rl = 0;
                rl &= km_lo;
                emit_rc = ok && !((rh ^ rcx_hi) == hi && (rl ^ rcx_lo) == lo);
            }
        }
        if (FILT && a.ranged) {
            const u32 fb = hi >> bshift, rb = rh >> bshift;
            emit_rc = emit_rc && rb >= a.range_lo && rb < a.range_hi;
            ok = ok && fb >= a.range_lo && fb < a.range_hi;
        }
        vmb |= (u32)ok << j;
        const u32 fd = (hi >> (pa.dshift - 32)) & pa.dmask;
        if (HIST) {
            if (RC || FILT) {
                if (ok) atomicAdd(&hist[0][fd], 1u);
            } else atomicAdd(&hist[0][fd], (u32)ok);      // branch-free: nearly every window of a read is a record
        }
        if (RC) {
            rmb |= (u32)emit_rc << j;
            if (HIST && emit_rc) atomicAdd(&hist[1][(rh >> (pa.dshift - 32)) & pa.dmask], 1u);
        }
    }
#ifdef EPC_PROBE
    if (EPC_PROBE == 5) return;
#endif
    // the wave's totals, one add per wave (an LDS add of a different value per lane to ONE address is turned by the
    // compiler into a scalar loop over the 64 lanes -- measured: half of this kernel's time): a lane's count has four
    // bits, each worth a ballot
    {
        const u32 cf = __popc(vmb), cr = __popc(rmb);
        u32 nf = 0, nr = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            nf += (u32)__popcll(__ballot((cf >> b) & 1u)) << b;
            if (RC) nr += (u32)__popcll(__ballot((cr >> b) & 1u)) << b;
        }
        if (lane == 0) {
            atomicAdd(&s_tot[0], nf);
            if (RC) atomicAdd(&s_tot[1], nr);
        }
    }
#ifdef EPC_PROBE
    if (EPC_PROBE == 51) return;
#endif
    // The write pass ranks its records round by round over 64 consecutive positions (thread tid: positions
    // wave*512 + j*64 + lane, bit j of its byte): the bits change hands through LDS, laid out so that a thread's eight
    // bytes -- one from each lane that holds one of its positions -- are one 8-byte read.
    tbits[0][wave * 64 + (lane & 7) * 8 + (lane >> 3)] = (uint8_t)vmb;
    if (RC) tbits[1][wave * 64 + (lane & 7) * 8 + (lane >> 3)] = (uint8_t)rmb;
#ifdef EPC_PROBE
    if (EPC_PROBE == 52) return;
#endif
    // the tile image for the write pass
    u32 *img = pa.img + (u64)tile * EP_IMG_WORDS;
    if (tid < (EX_TILE + EX_EXT) / 16) img[EP_IMG_PK + tid] = pk[tid];
#ifdef EPC_PROBE
    if (EPC_PROBE == 53) return;
#endif
    if (tid < EX_TILE / 32 + 2) img[EP_IMG_STB + tid] = stb[tid];
    __syncthreads();
#ifdef EPC_PROBE
    if (EPC_PROBE == 6) return;
#endif
    {
        const u32 at = wave * 64 + (lane >> 3) * 8;
        reinterpret_cast<uint8_t *>(img + EP_IMG_VM)[tid] =
            (uint8_t)ep_bit_column(*reinterpret_cast<const uint2 *>(&tbits[0][at]), lane & 7);
        if (RC) reinterpret_cast<uint8_t *>(img + EP_IMG_RM)[tid] =
            (uint8_t)ep_bit_column(*reinterpret_cast<const uint2 *>(&tbits[1][at]), lane & 7);
    }

    // histogram columns + the tile's record / forward-window totals
    if (HIST && tid < RS_DIGITS) {
        pa.table[(u64)tid * pa.ncols + col0] = hist[0][tid];
        if (RC) pa.table[(u64)tid * pa.ncols + col0 + 1] = hist[1][tid];
    }
    if (tid == 0) {
        a.tile_count[tile] = s_tot[0] + s_tot[1];
        a.tile_fwd[tile] = s_tot[0];
        img[EP_IMG_FLAG] = dup ? 1u : 0u;
    }
}

// Is the value a returning LDS add hands back a stable rank on this device?  Every wave runs rounds of adds on digits of
// several distributions (uniform over 256, over 16, over 2, all equal) and holds each returned value against the rank
// wave_match gives; *bad counts the differences.
__global__ __launch_bounds__(256) void lds_order_probe_kernel(u32 *bad, int rounds) {
    __shared__ u32 cnt_a[4][RS_DIGITS];
    __shared__ u32 cnt_m[4][RS_DIGITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * RS_DIGITS; i += 256) { (&cnt_a[0][0])[i] = 0; (&cnt_m[0][0])[i] = 0; }
    __syncthreads();
    u32 x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    u32 wrong = 0;
    for (int r = 0; r < rounds; r++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const int mode = (blockIdx.x + r) & 3;
        const u32 d = mode == 0 ? (x >> 8) & 255u : mode == 1 ? (x >> 8) & 15u : mode == 2 ? ((x >> 8) & 1u) * 33u : 77u;
        const bool valid = ((x >> 20) & 7u) != 0;                 // some lanes sit a round out
        const u32 got = valid ? atomicAdd(&cnt_a[wave][d], 1u) : 0u;
        u32 below, total;
        wave_match(d, valid, below, total);
        const u32 before = cnt_m[wave][d];
        if (valid && got != before + below) wrong++;
        if (valid && below == 0) cnt_m[wave][d] = before + total;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// ---- write pass: from the tile image straight to ranking, staging and the partitioned stores ---------------------
// Register budget: only the record bits and ranks live per window; the k-mer is re-read from the LDS image where it
// is needed (broadcast reads), which fits 3 workgroups per CU (<= 80 VGPRs, < 53,760 B of LDS: LDS is handed out in
// 1280-byte granules on gfx950).
// ATOM: a record's rank among its wave's records of the same digit is what a returning LDS add on the wave's counter
// of that digit hands back -- same-address adds of one wave instruction are applied in ascending lane order on gfx950
// and a wave's LDS instructions in program order, so (round, lane) order = window order is kept.  That order is no
// documented property: lds_order_probe_kernel checks it once per device (kmg_create), and where it fails -- or the
// library is built with -DEP_RANK_ATOMIC=0 -- the ranks come from wave-level digit matching (wave_match: 8 ballots per
// round, ~45 VALU instructions where the add is one LDS instruction; measured 0.16 -> 0.0x ms of this pass).
#ifndef EP_RANK_ATOMIC
#define EP_RANK_ATOMIC 1
#endif
template <typename V, bool ATOM>
__global__ __launch_bounds__(EP_BLOCK, 6) void extract_part_write_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 stb[EX_TILE / 32 + 4];
    __shared__ u32 wpre[EX_TILE / 32];                 // sequence starts before bitmap word w
    __shared__ typename std::conditional<ATOM, u32, uint16_t>::type wcnt[EP_WAVES][RS_DIGITS];
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
#ifdef EPW_PROBE
    if (EPW_PROBE == 1) return;
#endif
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
            if (ATOM) {
                const u32 r = valid ? atomicAdd(reinterpret_cast<u32 *>(&wcnt[wave][d]), 1u) : 0u;
                lpos[j] = (d << 16) | r;
            } else {
                u32 below, total;
                wave_match(d, valid, below, total);
                const u32 before = wcnt[wave][d];
                lpos[j] = (d << 16) | (before + below);
                if (valid && below == 0) wcnt[wave][d] = (uint16_t)(before + total);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#ifdef EPW_PROBE
    if (EPW_PROBE == 2) return;
#endif
        u32 tile_n;
        u32 tdx = tid;                      // opaque copy, as pb: no addresses hoisted out of the loop
        asm volatile("" : "+v"(tdx));
        {
            u32 run = 0;
            if (tdx < RS_DIGITS) {
#pragma unroll
                for (int w = 0; w < EP_WAVES; w++) {
                    u32 c = wcnt[w][tdx];
                    wcnt[w][tdx] = static_cast<typename std::conditional<ATOM, u32, uint16_t>::type>(run);
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
#ifdef EPW_PROBE
    if (EPW_PROBE == 3) return;
#endif
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
#ifdef EPW_PROBE
    if (EPW_PROBE == 4) return;
#endif
#pragma unroll 4
        for (int i = 0; i < EP_ROUNDS; i++) {
            const u32 lp = i * EP_BLOCK + tid;
            if (lp < tile_n) a.out_keys[lp + gdelta[sdig[lp]]] = stage[lp];
        }
        __syncthreads();
#ifdef EPW_PROBE
    if (EPW_PROBE == 5) return;
#endif
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
#ifdef EPW_PROBE
    if (EPW_PROBE == 6) return;
#endif
#pragma unroll 4
        for (int i = 0; i < EP_ROUNDS; i++) {
            const u32 lp = i * EP_BLOCK + tid;
            if (lp < tile_n) out_vals[lp + gdelta[sdig[lp]]] = reinterpret_cast<V *>(stage)[lp];
        }
        __syncthreads();
#ifdef EPW_PROBE
    if (EPW_PROBE == 7) return;
#endif
        if (sp == 0 && a.rc && tid < RS_DIGITS) {
#pragma unroll
            for (int w = 0; w < EP_WAVES; w++) wcnt[w][tid] = 0;
        }
        __syncthreads();
#ifdef EPW_PROBE
    if (EPW_PROBE == 8) return;
#endif
    }
}

// ---- write pass of a batch that is NOT partitioned (every batch of a build but the first): the records of a tile in
// window order -- a window's forward record, then its reverse complement -- at tile_off[tile] behind the records the
// build already holds.  From the count pass's tile image: nothing is loaded, packed or filtered again, and a tile that
// keeps nothing (key-range builds keep a slice: most windows of most tiles go) costs one load.
template <typename V>
__global__ __launch_bounds__(EP_BLOCK) void extract_img_write_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 stb[EX_TILE / 32 + 4];
    __shared__ u32 wpre[EX_TILE / 32];
    __shared__ u32 wave_cnt[EP_WAVES * EP_ROUNDS];       // records of (wave, round), then their exclusive prefix
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile = blockIdx.x;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 *img = pa.img + (u64)tile * EP_IMG_WORDS;
    const u32 img_w = tid < EP_IMG_VM ? img[tid] : 0u;               // packed bases, then the start bitmap
    const u32 vmask = reinterpret_cast<const uint8_t *>(img + EP_IMG_VM)[tid];
    const u32 rmask = a.rc ? reinterpret_cast<const uint8_t *>(img + EP_IMG_RM)[tid] : 0u;
    const bool dup = img[EP_IMG_FLAG] != 0;
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const u64 out_base = a.state[ST_NREC] + a.tile_off[tile];
    if (a.tile_count[tile] == 0) return;        // (asked for with the rest: one round trip, not one of its own)
    if (tid < EP_IMG_STB) pk[tid] = img_w;
    else if (tid < EP_IMG_VM) stb[tid - EP_IMG_STB] = img_w;
    // records per (wave, round), in window order: thread tid holds positions wave * 512 + j * 64 + lane
#pragma unroll
    for (int j = 0; j < EP_ROUNDS; j++) {
        const u32 c = (u32)__popcll(__ballot((vmask >> j) & 1u)) + (u32)__popcll(__ballot((rmask >> j) & 1u));
        if (lane == 0) wave_cnt[wave * EP_ROUNDS + j] = c;
    }
    __syncthreads();
    if (wave == 0) {
        static_assert(EP_WAVES * EP_ROUNDS == 64, "one wave scans the (wave, round) counts");
        const u32 v = wave_cnt[lane];
        u32 inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        wave_cnt[lane] = inc - v;
    } else if (wave == 1) {
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
    __syncthreads();
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    V *ov = reinterpret_cast<V *>(a.out_vals);
#pragma unroll
    for (int j = 0; j < EP_ROUNDS; j++) {
        const bool ok = (vmask >> j) & 1u, erc = (rmask >> j) & 1u;
        const u64 bal_v = __ballot(ok), bal_r = __ballot(erc);
        if ((bal_v | bal_r) == 0) continue;
        if (ok || erc) {
            // (the RC record can be there without the forward one: key-range builds keep each by its own bucket)
            const u32 p = (u32)wave * (64 * EP_ROUNDS) + (u32)j * 64 + (u32)lane;
            const u64 slot = out_base + wave_cnt[wave * EP_ROUNDS + j] + __popcll(bal_v & lt) + __popcll(bal_r & lt);
            const u64 rslot = slot + (ok ? 1 : 0);
            const u64 key = ep_window64(pk, p, kmask);
            if (ok) a.out_keys[slot] = key;
            if (erc) a.out_keys[rslot] = revpairs64(key >> (64 - 2 * k)) ^ rcx;
            u32 rel;
            if (dup) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
            else rel = wpre[p >> 5] + __popc(stb[p >> 5] >> (31 - (p & 31)));
            const u64 sq = (u64)s_lo + rel;
            V v;
            if (a.kind == KMG_PAYLOAD_TAXON) v = (V)(u32)a.seq_id[sq];
            else v = (V)(((u64)(u32)a.read_index[sq] << 16) | (u64)((t0 + p - a.seq_off[sq]) & 0xFFFFu));   // ReadPos(read, (short)pos)
            if (ok) ov[slot] = v;
            if (erc) ov[rslot] = v;
        }
    }
}

// ---- the same for batches that keep little (key-range builds: a slice of the key space, ~1 window in 64 at 64 of 4096
// buckets).  A 512-thread workgroup per tile spends its time waiting for a 2.6 KB image to arrive and on 64 wave-rounds
// with a lane or two at work; here ONE WAVE takes a tile -- eight times the tiles in flight per CU, no workgroup
// barrier -- and a LANE takes a record: the wave first notes, per (wave, round) of the image's layout, which lanes hold a
// forward / reverse-complement record (two 64-bit masks) and how many records precede the round; record r of the tile
// then finds its round by a binary search over those counts and its lane by one over the masks' bit counts, and the
// stores of 64 consecutive records leave together.  Same order as extract_img_write_kernel: window by window, a
// window's forward record before its reverse complement.
#define EPS_WAVES 4                                    // tiles per workgroup (one per wave)
__device__ __forceinline__ void eps_wave_sync() {     // a wave's LDS traffic is in program order: keep the compiler to it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename V>
__global__ __launch_bounds__(64 * EPS_WAVES) void extract_img_write_sparse_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 s_pk[EPS_WAVES][EX_TILE / 16 + 4];
    __shared__ u32 s_stb[EPS_WAVES][EX_TILE / 32 + 4];
    __shared__ u32 s_wpre[EPS_WAVES][EX_TILE / 32];
    __shared__ u64 s_bv[EPS_WAVES][EP_WAVES * EP_ROUNDS];   // lanes with a forward record, per (wave, round) of the image
    __shared__ u64 s_br[EPS_WAVES][EP_WAVES * EP_ROUNDS];   // ... with a reverse-complement record
    __shared__ u32 s_before[EPS_WAVES][EP_WAVES * EP_ROUNDS + 1];   // records before the round; [64] = the tile's total
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 tile = blockIdx.x * EPS_WAVES + wv;
    if (tile >= pa.ntiles) return;
    u32 *pk = s_pk[wv], *stb = s_stb[wv], *wpre = s_wpre[wv], *before = s_before[wv];
    u64 *bv = s_bv[wv], *br = s_br[wv];
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;
    const u32 *img = pa.img + (u64)tile * EP_IMG_WORDS;
    // everything the tile needs, asked for at once: the image's words (6 + 1 per lane), the record bits of the image's
    // thread w * 64 + lane for w = 0 .. 7
    u32 iw[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const u32 at = (u32)i * 64 + (u32)lane;
        iw[i] = at < EP_IMG_VM ? img[at] : 0u;
    }
    u32 vmw[EP_WAVES], rmw[EP_WAVES];
#pragma unroll
    for (int w = 0; w < EP_WAVES; w++) {
        vmw[w] = reinterpret_cast<const uint8_t *>(img + EP_IMG_VM)[w * 64 + lane];
        rmw[w] = a.rc ? reinterpret_cast<const uint8_t *>(img + EP_IMG_RM)[w * 64 + lane] : 0u;
    }
    const bool dup = img[EP_IMG_FLAG] != 0;
    const u32 s_lo = a.tile_seq[tile];
    const u32 s_end = a.tile_end[tile];
    const u64 out_base = a.state[ST_NREC] + a.tile_off[tile];
    const u32 n = a.tile_count[tile];
    if (n == 0) return;
    static_assert(EP_IMG_VM <= 7 * 64 && EP_IMG_STB == EX_TILE / 16 + 4, "image words per lane");
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const u32 at = (u32)i * 64 + (u32)lane;
        if (at < EP_IMG_STB) pk[at] = iw[i];
        else if (at < EP_IMG_VM) stb[at - EP_IMG_STB] = iw[i];
    }
    // the rounds' masks and the records before each round, in the image's (wave, round) order = window order
    u32 run = 0;
#pragma unroll
    for (int w = 0; w < EP_WAVES; w++) {
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            const u64 mv = __ballot((vmw[w] >> j) & 1u), mr = __ballot((rmw[w] >> j) & 1u);
            if (lane == 0) {
                bv[w * EP_ROUNDS + j] = mv;
                br[w * EP_ROUNDS + j] = mr;
                before[w * EP_ROUNDS + j] = run;
            }
            run += (u32)__popcll(mv) + (u32)__popcll(mr);
        }
    }
    if (lane == 0) before[EP_WAVES * EP_ROUNDS] = run;
    eps_wave_sync();
    {
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
    eps_wave_sync();
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    V *ov = reinterpret_cast<V *>(a.out_vals);
    for (u32 r0 = 0; r0 < run; r0 += 64) {
        const u32 r = r0 + (u32)lane;
        if (r >= run) continue;
        // the round: the last one with no more than r records before it (it holds records: the next round has more)
        u32 lo = 0, hi = EP_WAVES * EP_ROUNDS;             // before[lo] <= r < before[hi]
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const u32 mid = (lo + hi) >> 1;
            if (before[mid] <= r) lo = mid; else hi = mid;
        }
        const u32 q = r - before[lo];
        const u64 mv = bv[lo], mr = br[lo];
        // the lane: the first t with more than q records in lanes 0 .. t
        u32 tl = 0, th = 63;
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const u32 mid = (tl + th) >> 1;
            const u64 upto = mid == 63 ? ~0ull : ((1ull << (mid + 1)) - 1);
            const u32 c = (u32)__popcll(mv & upto) + (u32)__popcll(mr & upto);
            if (c > q) th = mid; else tl = mid + 1;
        }
        const u32 t = tl;
        const u64 below = t == 0 ? 0ull : ((1ull << t) - 1);
        const u32 cb = (u32)__popcll(mv & below) + (u32)__popcll(mr & below);
        const bool has_f = (mv >> t) & 1ull;
        const bool is_rc = !has_f || q != cb;                // the forward record comes first where both are there
        const u32 p = (lo >> 3) * (64 * EP_ROUNDS) + (lo & 7u) * 64 + t;
        u64 key = ep_window64(pk, p, kmask);
        if (is_rc) key = revpairs64(key >> (64 - 2 * k)) ^ rcx;
        u32 rel;
        if (dup) rel = ep_seq_rel(a, s_lo, s_end, t0 + p);
        else rel = wpre[p >> 5] + __popc(stb[p >> 5] >> (31 - (p & 31)));
        const u64 sq = (u64)s_lo + rel;
        V v;
        if (a.kind == KMG_PAYLOAD_TAXON) v = (V)(u32)a.seq_id[sq];
        else v = (V)(((u64)(u32)a.read_index[sq] << 16) | (u64)((t0 + p - a.seq_off[sq]) & 0xFFFFu));   // ReadPos(read, (short)pos)
        a.out_keys[out_base + r] = key;
        ov[out_base + r] = v;
    }
}
