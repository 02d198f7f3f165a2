// extract.h -- stage 1: 2-bit encode + k-mer windows + reverse complement + read numbering.
//
// Replaces the per-character loop of KmersFromFileSpliterator.tryAdvance
// (OtherFiles/KmersFromFile.java:150-178) and the RC branch of FileCreator.addKmers
// (KmerFiles/FileCreator.java:77-95).  HBM-bound byte work: every base is read once per pass and every
// record written once; nothing here is a contraction, so no MFMA.
//
// A tile is EX_TILE consecutive base positions of the concatenated batch.  The block
//   1. loads its bytes with 16-B coalesced loads and packs them into LDS: 2 bits per base (A0 C1 T2 G3,
//      Kmers/Base.java:7-10) plus one "not ACGTacgt" bit per base,
//   2. counts sequence starts per position in LDS and prefix-scans them, which gives every position its
//      sequence and a "sequence starts here" bitmask,
//   3. forms every window with two funnel shifts out of the packed LDS words (no per-base rolling,
//      no redundant re-encoding): a window is a k-mer iff no bad base in [p, p+k) and no sequence
//      start in (p, p+k),
//   4. count pass: per-tile record counts (+ has_kmer flags per sequence);  write pass: records at
//      exclusive-scanned offsets, in window order (the later stable sort relies on that order).
#pragma once
#include "common.h"
#include "scan.h"

#define EX_BLOCK 256
#define EX_ROUNDS 16
#define EX_TILE (EX_BLOCK * EX_ROUNDS)   // 4096 window-start positions per tile
#define EX_EXT 32                        // bases past the tile that its last windows may cover

// tile_seq[t] = largest s in [0, nseq] with seq_off[s] <= t * EX_TILE;
// tile_end[t] (optional) = largest s with seq_off[s] < (t + 1) * EX_TILE + EX_EXT: the last sequence start a
// tile's windows can see, so that the kernels read exactly the offsets they need.
__global__ void extract_tileseq_kernel(const u64 *__restrict__ seq_off, u64 nseq, u32 *__restrict__ tile_seq,
                                       u32 *__restrict__ tile_end, u32 ntiles_plus1) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles_plus1) return;
    u64 pos = (u64)t * EX_TILE;
    u64 lo = 0, hi = nseq + 1;  // first index with seq_off > pos
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (seq_off[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    tile_seq[t] = (u32)(lo == 0 ? 0 : lo - 1);
    if (tile_end) {
        const u64 lim = pos + EX_TILE + EX_EXT;
        hi = nseq + 1;              // first index with seq_off >= lim (not below the first search's result)
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (seq_off[mid] < lim) lo = mid + 1; else hi = mid;
        }
        tile_end[t] = (u32)(lo == 0 ? 0 : lo - 1);
    }
}

__device__ __forceinline__ u64 revpairs64(u64 x) {
    // reverse the order of the 32 two-bit groups of x
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// Kmers/RunOfSame.java:12-36 on the packed k-mer: drop the k-mer iff some run of equal bases reaches
// `length` (the reference only checks after an increment, so length < 2 never drops anything).
__device__ __forceinline__ bool runofsame_keep(u64 key, int k, int length) {
    const int R = length - 1;                          // consecutive equal neighbour pairs needed
    if (R < 1 || R > k - 1) return true;
    const u64 x = key ^ (key << 2);                    // group i (from the top): base_i ^ base_{i+1}
    const u64 lowbits = 0x5555555555555555ull;
    u64 eq = ~(x | (x >> 1)) & lowbits;                // 1 per group whose two bases are equal
    eq &= (k - 1) >= 32 ? ~0ull : ~(~0ull >> (2 * (k - 1)));   // pairs 0 .. k-2 only
    u64 t = eq;
    int s = 1;
    while (s * 2 <= R) { t &= t << (2 * s); s *= 2; }  // t[i] = eq[i .. i+s-1] all set
    if (R > s) t &= t << (2 * (R - s));
    return t == 0;
}

// Kmers/Dust.java:12-30 on the packed k-mer.  score = sum c(c-1) over the triplet counts = twice the
// number of unordered pairs of equal triplets, so "score < threshold * k * 2" is "pairs < threshold * k".
__device__ __forceinline__ bool dust_keep(u64 key, int k, int threshold) {
    const int nt = k - 2;                              // triplets at positions 0 .. nt-1
    int pairs = 0;
    const u64 lowbits = 0x5555555555555555ull;
    for (int d = 1; d < nt; d++) {
        const u64 y = key ^ (key << (2 * d));          // group i: base_i ^ base_{i+d}
        const u64 e = ~(y | (y >> 1)) & lowbits;       // base_i == base_{i+d}
        u64 t3 = e & (e << 2) & (e << 4);              // ... and the next two bases as well
        t3 &= ~(~0ull >> (2 * (nt - d)));              // i = 0 .. nt-1-d
        pairs += __popcll(t3);
    }
    return pairs < threshold * k;
}

struct ExtractArgs {
    const uint8_t *bases;
    u64 nbases;
    const u64 *seq_off;
    u64 nseq;
    const int32_t *seq_id;       // taxon per sequence (kind 0)
    const int32_t *read_index;   // read number per sequence (kind 1, write pass)
    const u32 *tile_seq;
    const u32 *tile_end;         // last sequence start visible from the tile (extract_part.h)
    int k;
    int rc;
    int kind;
    int dust;                    // MakeDatabase -D threshold, -1 = off
    int runofsame;               // MakeDatabase -R length, -1 = off
    // key-range builds (kmg_build_set_key_range): a record is kept iff its OWN key bucket -- the forward k-mer's
    // for the forward record, the reverse complement's for the RC record -- lies in [range_lo, range_hi)
    int ranged;
    int range_shift;             // 64 - 2 * key_len
    u32 range_lo, range_hi;
    // count pass outputs
    u32 *tile_count;
    u32 *tile_fwd;               // valid forward windows per tile
    uint8_t *has_kmer;
    // write pass
    const u64 *tile_off;         // exclusive scan of tile_count
    u64 *out_keys;
    void *out_vals;
    u64 *state;                  // device build state (ST_*)
    int pos_bits;                // read payload = read << pos_bits | (pos & pos_mask): 16 / 0xFFFF unless compact
    u32 pos_mask;
};

// Steps 1 and 2 of a tile for the short-k-mer kernels of extract_short.h (the full windows: extract_part.h): the tile's bytes
// packed into LDS (pk: 2 bits per base, inv: "not ACGTacgt" bits, both with EX_EXT positions past the tile), the
// "some sequence starts here" bitmap stb, and seqrel[p] = number of sequences beyond the tile's first one that start
// at or before position p.  -> index of the tile's first sequence.  Ends behind a __syncthreads().
// (EXT = bases past the tile that its windows may cover: EX_EXT for k <= 32, 64 for the two-word k-mers of wide.h; the
// arrays hold (EX_TILE + EXT) / 16 words of pk, (EX_TILE + EXT) / 32 of inv and stb, EX_TILE + EXT of seqrel)
template <int EXT>
__device__ __forceinline__ u32 ex_load_tile_ext(const ExtractArgs &a, const u32 tile, u32 *pk, u32 *inv, u32 *stb, u32 *seqrel,
                                                u32 *scan_smem) {
    const int tid = threadIdx.x;
    const u64 t0 = (u64)tile * EX_TILE;
    // ---- 1. load + pack ---------------------------------------------------------------------------
    const bool aligned = (((uintptr_t)a.bases) & 15) == 0;
    for (int chunk = tid; chunk < (EX_TILE + EXT) / 16; chunk += EX_BLOCK) {
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
            // Base.fromCharacterByte accepts both cases; A C G T = 0x41 + {0, 2, 6, 19}: one look into a 32-bit set
            const u32 t = (c & 0xDFu) - 0x41u;
            const u32 ok = (0x00080045u >> (t < 31u ? t : 31u)) & 1u;
            word = (word << 2) | ((c >> 1) & 3u);  // 'A'>>1&3=0 'C'=1 'T'=2 'G'=3
            bad = (bad << 1) | (ok ^ 1u);
        }
        pk[chunk] = word;
        reinterpret_cast<uint16_t *>(inv)[chunk ^ 1] = (uint16_t)bad;
    }
    for (int i = tid; i < EX_TILE + EXT; i += EX_BLOCK) seqrel[i] = 0;
    __syncthreads();

    // ---- 2. sequence starts -> per-position sequence index ------------------------------------------
    const u32 s_lo = a.tile_seq[tile];
    for (u64 s = (u64)s_lo + 1 + tid; s <= a.nseq; s += EX_BLOCK) {
        u64 o = a.seq_off[s];
        if (o >= t0 + EX_TILE + EXT) break;
        atomicAdd(&seqrel[(u32)(o - t0)], 1u);
    }
    __syncthreads();
    {
        // blocked: thread owns positions 16*tid .. 16*tid+15 (threads 0,1 also own the extension)
        u32 c[16];
        u32 sum = 0, bits = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            c[i] = seqrel[tid * 16 + i];
            bits = (bits << 1) | (c[i] != 0);
            sum += c[i];
        }
        reinterpret_cast<uint16_t *>(stb)[tid ^ 1] = (uint16_t)bits;
        if (tid < EXT / 16) {
            u32 eb = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) eb = (eb << 1) | (seqrel[EX_TILE + tid * 16 + i] != 0);
            reinterpret_cast<uint16_t *>(stb)[(EX_TILE / 16 + tid) ^ 1] = (uint16_t)eb;
        }
        u32 total;
        u32 incl = block_scan_inclusive(sum, OpSum(), scan_smem, &total);
        u32 run = incl - sum;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            run += c[i];
            seqrel[tid * 16 + i] = run;
        }
    }
    __syncthreads();

    return s_lo;
}
__device__ __forceinline__ u32 ex_load_tile(const ExtractArgs &a, const u32 tile, u32 *pk, u32 *inv, u32 *stb, u32 *seqrel,
                                            u32 *scan_smem) {
    return ex_load_tile_ext<EX_EXT>(a, tile, pk, inv, stb, seqrel, scan_smem);
}

// After a write pass: advance the running record count by this batch's total (tile_off[ntiles]).
__global__ void extract_advance_kernel(u64 *state, const u64 *tile_off, u32 ntiles) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[ST_NREC] += tile_off[ntiles];
}

// out += sum(in[0..n)): one 64-bit atomic per block of 2048 elements.
__global__ __launch_bounds__(256) void sum_u32_kernel(const u32 *__restrict__ in, u32 n, u64 *out) {
    __shared__ u64 smem[4];
    u64 acc = 0;
    u32 base = blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 idx = base + i * 256;
        if (idx < n) acc += in[idx];
    }
    u64 total;
    block_scan_inclusive(acc, OpSum(), smem, &total);
    if (threadIdx.x == 0 && total) atomicAdd(out, total);
}

// OR / AND of (taxon ^ sign bit) over a batch: value bits that never vary need no sort pass.
__global__ void id_bits_kernel(const int32_t *__restrict__ ids, u64 n, u64 *state) {
    u32 o = 0, an = 0xFFFFFFFFu, mn = 0xFFFFFFFFu, mx = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 v = (u32)ids[i] ^ 0x80000000u;
        o |= v;
        an &= v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        o |= __shfl_down(o, d, 64);
        an &= __shfl_down(an, d, 64);
        const u32 omn = __shfl_down(mn, d, 64), omx = __shfl_down(mx, d, 64);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicOr(&state[ST_ID_OR], (u64)o);
        atomicAnd(&state[ST_ID_AND], (u64)an);
        atomicMin((unsigned long long *)&state[ST_ID_MIN], (unsigned long long)mn);
        atomicMax((unsigned long long *)&state[ST_ID_MAX], (unsigned long long)mx);
    }
}

// ---- read numbering (ReadIDMapping.geNext, OtherFiles/ReadIDMapping.java:14-23) ----------------------
// A sequence that produced a k-mer gets the next read index unless its header token equals that of the last sequence
// that produced one (the carried state across batches: ST_CARRY_TOK / ST_HAVE_CARRY / ST_READ_BASE).  One launch: a
// tile of NR_TILE sequences finds each emitter's predecessor (inside the tile by a max-scan; the tile's first emitter
// walks back through the flags before the tile, 64 a step -- one step unless reads without a k-mer pile up there),
// counts the emitters whose token differs from their predecessor's, and takes its offset from the tiles before it
// through the chained look-back of scan.h.  The tile that finishes last advances the carried state.
#define NR_BLOCK 256
#ifndef NR_ITEMS
#define NR_ITEMS 8
#endif
#define NR_TILE (NR_BLOCK * NR_ITEMS)

// last sequence at or before `from` that produced a k-mer, or -1 (called by every lane of one wave)
__device__ __forceinline__ long long nr_walk_back(const uint8_t *__restrict__ has_kmer, long long from) {
    const int lane = threadIdx.x & 63;
    for (long long hi = from; hi >= 0; hi -= 64) {
        const long long i = hi - lane;
        const u64 b = __ballot(i >= 0 && has_kmer[i >= 0 ? i : 0] != 0);
        if (b) return hi - (long long)(__ffsll((unsigned long long)b) - 1);
    }
    return -1;
}

__global__ __launch_bounds__(NR_BLOCK) void number_reads_kernel(const uint8_t *__restrict__ has_kmer, const int64_t *__restrict__ tok,
                                                                 u64 n, int32_t *__restrict__ read_index, u64 *state,
                                                                 u64 *__restrict__ status, u32 *__restrict__ ctl, u32 epoch, u32 ntiles) {
    __shared__ long long smax[4];
    __shared__ long long line[NR_BLOCK];
    __shared__ u32 ssum[4];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    __shared__ long long s_prev;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 tile = cs_take_tile(ctl, &s_tile);
    // the carried state as the batch found it (only the tile that finishes last writes it, after every tile has read)
    const u64 read_base = state[ST_READ_BASE], have_carry = state[ST_HAVE_CARRY];
    const int64_t carry_tok = (int64_t)state[ST_CARRY_TOK];
    const u64 t0 = (u64)tile * NR_TILE, i0 = t0 + (u64)tid * NR_ITEMS;
    if (tid < 64) {
        const long long p = t0 ? nr_walk_back(has_kmer, (long long)t0 - 1) : -1;
        if (lane == 0) s_prev = p;
    }
    // (every load of the thread asked for up front, at clamped indices: a load under `if (h[i])` compiles to a branch
    // with a wait behind it, one memory round trip per sequence)
    bool h[NR_ITEMS];
    int64_t tk[NR_ITEMS];
    long long last = -1;
#pragma unroll
    for (int i = 0; i < NR_ITEMS; i++) {
        const u64 ic = i0 + i < n ? i0 + i : n - 1;
        h[i] = has_kmer[ic] != 0;
        tk[i] = tok[ic];
    }
#pragma unroll
    for (int i = 0; i < NR_ITEMS; i++) {
        h[i] = h[i] && i0 + i < n;
        if (h[i]) last = (long long)(i0 + i);
    }
    long long tot_unused;
    line[tid] = block_scan_inclusive<long long, OpMax, NR_BLOCK / 64>(last, OpMax(), smax, &tot_unused);
    __syncthreads();
    long long prev = tid ? line[tid - 1] : -1;       // last emitter of the tile before this thread's sequences
    if (prev < 0) prev = s_prev;
    u32 nw = 0, cnt = 0;
    bool dup = false;
    int64_t prev_tok = prev >= 0 ? tok[prev] : 0;
#pragma unroll
    for (int i = 0; i < NR_ITEMS; i++) {
        if (h[i]) {
            const int64_t t = tk[i];
            const bool is_new = prev >= 0 ? t != prev_tok : !(have_carry && carry_tok == t);
            // two sequences that share a read index: their (read, pos) pairs are no longer generated in ascending
            // order, so the sort must also order the payload (rare; costs extra passes)
            dup = dup || !is_new;
            nw |= (u32)is_new << i;
            cnt += is_new;
            prev = (long long)(i0 + i);
            prev_tok = t;
        }
    }
    if (dup) state[ST_DUP_READS] = 1;
    u32 total;
    const u32 inc = block_scan_inclusive<u32, OpSum, NR_BLOCK / 64>(cnt, OpSum(), ssum, &total);
    if (tid < 64) {
        const u64 e = cs_lookback(status, epoch, tile, (u64)total);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
    u64 run = read_base + s_excl + inc - cnt;
#pragma unroll
    for (int i = 0; i < NR_ITEMS; i++) {
        if (i0 + i < n) {
            run += (nw >> i) & 1u;
            read_index[i0 + i] = h[i] ? (int32_t)run - 1 : -1;
        }
    }
    if (tid < 64) {
        int fin = 0;
        if (lane == 0) fin = cs_tile_done(ctl, ntiles) ? 1 : 0;
        if (__shfl(fin, 0, 64)) {
            // every tile is done: the batch's total is the last tile's inclusive prefix
            const u64 w = __hip_atomic_load(&status[ntiles - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long lastem = nr_walk_back(has_kmer, (long long)n - 1);
            if (lane == 0) {
                if (lastem >= 0) {
                    state[ST_CARRY_TOK] = (u64)tok[lastem];
                    state[ST_HAVE_CARRY] = 1;
                }
                state[ST_READ_BASE] = read_base + (w & CS_VAL_MASK);
            }
        }
    }
}


// taxon builds that asked for it (kmg_build_add's read_index_out): 0 = the sequence produced a k-mer, -1 = it did not
__global__ void has_kmer_flags_kernel(const uint8_t *__restrict__ has_kmer, u64 n, int32_t *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = has_kmer[i] ? 0 : -1;
}

// MakeDatabase -p input as it lies in the file: Sequence.cBytes (Sequence.java:90-115) of every sequence, 2 bits per
// base, first base in the top bits of a byte, every sequence starting on a byte boundary.  One thread per packed byte
// finds its sequence in byte_off (ascending, nseq + 1 entries) and writes up to four base letters.
__global__ void unpack2bit_kernel(const uint8_t *__restrict__ packed, u64 nbytes, const u64 *__restrict__ byte_off,
                                  const u64 *__restrict__ seq_off, u64 nseq, uint8_t *__restrict__ out) {
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbytes) return;
    u64 lo = 0, hi = nseq;                       // last sequence with byte_off <= b
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (byte_off[mid] <= b) lo = mid; else hi = mid;
    }
    const u64 first = seq_off[lo] + 4 * (b - byte_off[lo]), end = seq_off[lo + 1];
    const u32 v = packed[b];
    const u32 letters = 0x47544341u;             // 'A' 'C' 'T' 'G' for codes 0..3 (Base.java:7-41)
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (first + i < end) out[first + i] = (uint8_t)(letters >> (8 * ((v >> (6 - 2 * i)) & 3u)));
}
