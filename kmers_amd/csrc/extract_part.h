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
// Same window logic as extract_kernel (extract.h), 512 threads: a wave owns 512 consecutive positions.
#pragma once
#include "common.h"
#include "extract.h"
#include "radix.h"

#define EP_BLOCK 512
#define EP_WAVES (EP_BLOCK / 64)
#define EP_ROUNDS (EX_TILE / EP_BLOCK)   // 8 windows per thread

struct PartArgs {
    ExtractArgs e;
    int dshift;        // the partition digit: (key >> dshift) & dmask
    u32 dmask;
    u32 *table;        // [digit][column] counts (count pass) / exclusive offsets (write pass)
    u32 ncols;         // columns = tiles * (rc ? 2 : 1)
    u32 ntiles;
};

template <bool WRITE, typename V>
__global__ __launch_bounds__(EP_BLOCK, 4) void extract_part_kernel(PartArgs pa) {
    const ExtractArgs &a = pa.e;
    __shared__ u32 pk[EX_TILE / 16 + 4];
    __shared__ u32 inv[EX_TILE / 32 + 2];
    __shared__ u32 stb[EX_TILE / 32 + 2];
    __shared__ u32 seqrel[EX_TILE + EX_EXT];
    __shared__ u32 wcnt[EP_WAVES][RS_DIGITS];          // count pass: rows 0/1 = forward / RC histograms
    __shared__ u32 dstart[RS_DIGITS];
    __shared__ u32 gdelta[RS_DIGITS];
    __shared__ u64 stage[WRITE ? EX_TILE : 1];
    __shared__ uint8_t sdig[WRITE ? EX_TILE : 1];
    __shared__ u32 scan_smem[EP_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (see rs_scatter_kernel): neighbouring tiles' runs are adjacent in the output
    const u32 per_xcd = (pa.ntiles + 7) / 8;
    const u32 tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= pa.ntiles) return;
    const u64 t0 = (u64)tile * EX_TILE;
    const int k = a.k;

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
    for (int i = tid; i < EX_TILE + EX_EXT; i += EP_BLOCK) seqrel[i] = 0;
    if (tid < RS_DIGITS) {
#pragma unroll
        for (int w = 0; w < EP_WAVES; w++) wcnt[w][tid] = 0;
    }
    __syncthreads();

    // ---- 2. sequence starts -> per-position sequence index ------------------------------------------
    const u32 s_lo = a.tile_seq[tile];
    for (u64 s = (u64)s_lo + 1 + tid; s <= a.nseq; s += EP_BLOCK) {
        u64 o = a.seq_off[s];
        if (o >= t0 + EX_TILE + EX_EXT) break;
        atomicAdd(&seqrel[(u32)(o - t0)], 1u);
    }
    __syncthreads();
    {
        // blocked: thread owns positions 8*tid .. 8*tid+7 = byte `tid` of the MSB-first start bitmap
        u32 c[8];
        u32 sum = 0, bits = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c[i] = seqrel[tid * 8 + i];
            bits = (bits << 1) | (c[i] != 0);
            sum += c[i];
        }
        reinterpret_cast<uint8_t *>(stb)[(tid & ~3) + (3 - (tid & 3))] = (uint8_t)bits;
        if (tid < EX_EXT / 8) {
            u32 eb = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) eb = (eb << 1) | (seqrel[EX_TILE + tid * 8 + i] != 0);
            const int bn = EX_TILE / 8 + tid;
            reinterpret_cast<uint8_t *>(stb)[(bn & ~3) + (3 - (bn & 3))] = (uint8_t)eb;
        }
        u32 total;
        u32 incl = block_scan_inclusive<u32, OpSum, EP_WAVES>(sum, OpSum(), scan_smem, &total);
        u32 run = incl - sum;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            run += c[i];
            seqrel[tid * 8 + i] = run;
        }
    }
    __syncthreads();

    // ---- 3. windows: a wave owns positions wave*512 .. +511, round j covers 64 consecutive ones ------
    const u64 kmask = k == 32 ? ~0ull : ~(~0ull >> (2 * k));
    const u64 rcx = 0xAAAAAAAAAAAAAAAAull & kmask;
    u64 fkey[EP_ROUNDS];
    V val[EP_ROUNDS];
    u32 vmask = 0, rmask = 0;
#pragma unroll
    for (int j = 0; j < EP_ROUNDS; j++) {
        const u32 p = wave * (64 * EP_ROUNDS) + j * 64 + lane;
        const u32 w = p >> 4, r = p & 15;
        u64 hi = ((u64)pk[w] << 32) | pk[w + 1];
        u64 win = (hi << (2 * r)) | ((u64)pk[w + 2] >> (32 - 2 * r));
        u64 key = win & kmask;
        const u32 w2 = p >> 5, r2 = p & 31;
        u64 bad = (((u64)inv[w2] << 32) | inv[w2 + 1]) << r2;
        u64 st = (((u64)stb[w2] << 32) | stb[w2 + 1]) << r2;
        bool ok = (bad >> (64 - k)) == 0;
        if (k > 1) ok = ok && (((st << 1) >> (65 - k)) == 0);
        ok = ok && (t0 + p + (u64)k <= a.nbases);
        if (!WRITE && ok && a.has_kmer) a.has_kmer[s_lo + seqrel[p]] = 1;
        if (ok && a.runofsame >= 0) ok = runofsame_keep(key, k, a.runofsame);
        if (ok && a.dust >= 0) ok = dust_keep(key, k, a.dust);
        fkey[j] = key;
        bool emit_rc = false;
        if (a.rc && ok) {
            u64 rev = revpairs64(key >> (64 - 2 * k));
            emit_rc = !(((k & 1) == 0) && rev == key);
        }
        vmask |= (u32)ok << j;
        rmask |= (u32)emit_rc << j;
        val[j] = 0;
        if (WRITE && ok) {
            const u32 s = s_lo + seqrel[p];
            if (a.kind == KMG_PAYLOAD_TAXON) val[j] = (V)(u32)a.seq_id[s];
            else val[j] = (V)(((u64)(u32)a.read_index[s] << 16) | (u64)((t0 + p - a.seq_off[s]) & 0xFFFFu));
        }
        if (!WRITE) {
            if (ok) atomicAdd(&wcnt[0][(u32)(key >> pa.dshift) & pa.dmask], 1u);
            if (emit_rc) {
                const u64 rk = revpairs64(key >> (64 - 2 * k)) ^ rcx;
                atomicAdd(&wcnt[1][(u32)(rk >> pa.dshift) & pa.dmask], 1u);
            }
        }
    }
    __syncthreads();
    const u32 ncol_per_tile = a.rc ? 2u : 1u;
    const u32 col0 = tile * ncol_per_tile;

    if (!WRITE) {
        // histogram columns + the tile's record / forward-window totals
        u32 hf = 0, hr = 0;
        if (tid < RS_DIGITS) {
            hf = wcnt[0][tid];
            hr = wcnt[1][tid];
            pa.table[(u64)tid * pa.ncols + col0] = hf;
            if (a.rc) pa.table[(u64)tid * pa.ncols + col0 + 1] = hr;
        }
        u32 tf, tr;
        block_scan_inclusive<u32, OpSum, EP_WAVES>(hf, OpSum(), scan_smem, &tf);
        block_scan_inclusive<u32, OpSum, EP_WAVES>(hr, OpSum(), scan_smem, &tr);
        if (tid == 0) {
            a.tile_count[tile] = tf + tr;
            a.tile_fwd[tile] = tf;
        }
        return;
    }

    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const u32 out_base = (u32)a.state[ST_NREC];
    V *out_vals = reinterpret_cast<V *>(a.out_vals);
    for (int sp = 0; sp < (a.rc ? 2 : 1); sp++) {
        const u32 imask = sp == 0 ? vmask : rmask;
        u32 lpos[EP_ROUNDS];
        // ---- stable rank of every record among the tile's records with the same digit
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            const bool valid = (imask >> j) & 1;
            const u64 key = sp == 0 ? fkey[j] : (revpairs64(fkey[j] >> (64 - 2 * k)) ^ rcx);
            const u32 d = (u32)(key >> pa.dshift) & pa.dmask;
            u64 m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < RS_BITS; b++) {
                const bool bit = (d >> b) & 1;
                const u64 bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const u32 before = wcnt[wave][d];
            lpos[j] = before + __popcll(m & lt);
            if (valid && (m & lt) == 0) wcnt[wave][d] = before + __popcll(m);
        }
        __syncthreads();
        u32 tile_n;
        {
            u32 run = 0;
            if (tid < RS_DIGITS) {
#pragma unroll
                for (int w = 0; w < EP_WAVES; w++) {
                    u32 c = wcnt[w][tid];
                    wcnt[w][tid] = run;
                    run += c;
                }
            }
            u32 incl = block_scan_inclusive<u32, OpSum, EP_WAVES>(run, OpSum(), scan_smem, &tile_n);
            if (tid < RS_DIGITS) {
                const u32 start = incl - run;
                dstart[tid] = start;
                gdelta[tid] = out_base + pa.table[(u64)tid * pa.ncols + col0 + sp] - start;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++) {
            if ((imask >> j) & 1) {
                const u64 key = sp == 0 ? fkey[j] : (revpairs64(fkey[j] >> (64 - 2 * k)) ^ rcx);
                const u32 d = (u32)(key >> pa.dshift) & pa.dmask;
                lpos[j] += dstart[d] + wcnt[wave][d];
                stage[lpos[j]] = key;
                sdig[lpos[j]] = (uint8_t)d;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EP_ROUNDS; i++) {
            const u32 lp = i * EP_BLOCK + tid;
            if (lp < tile_n) a.out_keys[lp + gdelta[sdig[lp]]] = stage[lp];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EP_ROUNDS; j++)
            if ((imask >> j) & 1) reinterpret_cast<V *>(stage)[lpos[j]] = val[j];
        __syncthreads();
#pragma unroll
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
