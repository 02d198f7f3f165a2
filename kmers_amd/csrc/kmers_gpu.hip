// kmers_gpu.hip -- C ABI (include/kmers_gpu.h) and host orchestration of the MI355X k-mer hot path.
// gfx950 only.  Stages: extract.h (encode + windows + RC + read numbering), radix.h (stable LSD sort),
// reduce.h (dedup -> CSR, bucket offsets, block encoding), match.h (sorted join, TaxaCounts).
#include "common.h"
#include "scan.h"
#include "extract.h"
#include "extract_part.h"
#include "extract_short.h"
#include "radix.h"
#include "reduce.h"
#include "match.h"
#include "collect.h"
#include "wide.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

// ------------------------------------------------------------------------------------------------ utils
__global__ void set_u64_kernel(u64 *p, u64 v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}

struct PoolBuf {
    // RAII: device scratch that goes back to the context's pool when the scope ends.  Safe with
    // asynchronous work because every kernel of a context runs on its one stream, so any later reuse
    // of the block is ordered after the work that still reads it.
    kmg_ctx *ctx;
    void *p = nullptr;
    explicit PoolBuf(kmg_ctx *c) : ctx(c) {}
    ~PoolBuf() { ctx->pool.release(p); }
    int alloc(size_t bytes) { return ctx->pool.alloc(&p, bytes); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
    void *detach() { void *q = p; p = nullptr; return q; }
    void free() { ctx->pool.release(p); p = nullptr; }
};

static int fail(kmg_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define KMG_ALLOC(buf, bytes)                                                             \
    do {                                                                                  \
        if ((buf).alloc(bytes) != KMG_OK) return fail(ctx, KMG_ENOMEM, "device allocation failed"); \
    } while (0)

// ------------------------------------------------------------------------------------------------ context
static int resolve_match_scratch(kmg_ctx *ctx);
static void forget_match_scratch_user(kmg_ctx *ctx);

// Once per device and process: does a returning LDS add hand back a stable rank here (extract_part.h, ATOM)?  A device
// where the probe cannot run, or finds one difference, takes the wave_match kernels.
static bool lds_rank_order_holds(kmg_ctx *ctx) {
    static std::mutex mu;
    static std::map<int, bool> known;
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(ctx->device);
    if (it != known.end()) return it->second;
    bool ok = false;
    void *d_bad = nullptr;
    if (ctx->pool.alloc(&d_bad, 64) == KMG_OK) {
        u32 bad = 1;
        if (hipMemsetAsync(d_bad, 0, 4, ctx->stream) == hipSuccess) {
            hipLaunchKernelGGL(lds_order_probe_kernel, dim3(1024), dim3(256), 0, ctx->stream, (u32 *)d_bad, 64);
            if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess)
                ok = bad == 0;
        }
        ctx->pool.release(d_bad);
    }
    (void)hipGetLastError();
    known[ctx->device] = ok;
    return ok;
}

extern "C" int kmg_create(int device, kmg_ctx **out) {
    if (!out) return KMG_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KMG_EDEVICE;
    if (device < 0 || device >= ndev) return KMG_EINVAL;
    kmg_ctx *ctx = new (std::nothrow) kmg_ctx();
    if (!ctx) return KMG_ENOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return KMG_EDEVICE;
    }
    void *st = nullptr;
    if (ctx->pool.alloc(&st, ST_WORDS * sizeof(u64)) != KMG_OK) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return KMG_ENOMEM;
    }
    ctx->d_state = (u64 *)st;
    if (resolve_match_scratch(ctx) != KMG_OK) {
        ctx->pool.destroy();
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return KMG_EDEVICE;
    }
    if (hipEventCreateWithFlags(&ctx->ev_scratch, hipEventDisableTiming) != hipSuccess) ctx->ev_scratch = nullptr;
    // (KMG_RANK_BY_MATCH in the environment: the wave_match kernels whatever the probe says -- for the test that holds the two
    // against each other)
    ctx->lds_rank_ok = EP_RANK_ATOMIC && !getenv("KMG_RANK_BY_MATCH") && lds_rank_order_holds(ctx);
    if (hipHostMalloc(&ctx->h_pin, PIN_BYTES, hipHostMallocDefault) != hipSuccess) ctx->h_pin = nullptr;   // optional
    if (hipHostMalloc(&ctx->h_snap, PIN_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_len, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_snap, hipEventDisableTiming) != hipSuccess) {
        if (ctx->h_snap) (void)hipHostFree(ctx->h_snap);
        ctx->h_snap = nullptr;                                                                             // optional too
    }
    *out = ctx;
    return KMG_OK;
}

extern "C" void kmg_destroy(kmg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    forget_match_scratch_user(ctx);
    if (ctx->ev_scratch) (void)hipEventDestroy(ctx->ev_scratch);
    prof_drain(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    ctx->pool.destroy();
    if (ctx->scan_status) (void)hipFree(ctx->scan_status);
    if (ctx->scan_ctl) (void)hipFree(ctx->scan_ctl);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_snap) (void)hipHostFree(ctx->h_snap);
    if (ctx->ev_len) (void)hipEventDestroy(ctx->ev_len);
    if (ctx->ev_snap) (void)hipEventDestroy(ctx->ev_snap);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *kmg_last_error(kmg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void *kmg_stream(kmg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ------------------------------------------------------------------------------------------------ build
extern "C" int kmg_build_begin(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (k < 1 || key_len < 1) return fail(ctx, KMG_EINVAL, "k and key_len must be positive");
    if (k > 64) return fail(ctx, KMG_EUNSUPPORTED, "k > 64 is not supported on the GPU path");
    if (key_len > k || key_len > 12) return fail(ctx, KMG_EUNSUPPORTED, "key_len must be <= min(k, 12)");
    if (payload_kind != KMG_PAYLOAD_TAXON && payload_kind != KMG_PAYLOAD_READPOS)
        return fail(ctx, KMG_EINVAL, "unknown payload kind");
    KMG_HIP(hipSetDevice(ctx->device));
    ctx->k = k; ctx->key_len = key_len; ctx->rc = rc ? 1 : 0; ctx->kind = payload_kind;
    ctx->min_k = k; ctx->sh_n = 0;
    ctx->dust = -1; ctx->runofsame = -1;
    ctx->ranged = false; ctx->range_lo = 0; ctx->range_hi = 0;
    ctx->rec_upper = 0;
    ctx->n_adds = 0; ctx->prepart = false; ctx->plan_pbits = 0; ctx->compact_pb = 0; ctx->snap_valid = false;
    // stream-ordered, no host round trip
    KMG_HIP(hipMemsetAsync(ctx->d_state, 0, ST_WORDS * sizeof(u64), ctx->stream));
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state + ST_ID_AND, (u64)0xFFFFFFFFull);
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state + ST_ID_MIN, (u64)0xFFFFFFFFull);
    KMG_HIP(hipGetLastError());
    ctx->building = true;
    return KMG_OK;
}

extern "C" int kmg_build_set_filters(kmg_ctx *ctx, int dust_threshold, int runofsame_length) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_set_filters before kmg_build_begin");
    if (ctx->rec_upper) return fail(ctx, KMG_ESTATE, "filters must be set before the first kmg_build_add");
    ctx->dust = dust_threshold < 0 ? -1 : dust_threshold;
    ctx->runofsame = runofsame_length < 0 ? -1 : runofsame_length;
    return KMG_OK;
}

extern "C" int kmg_build_set_min_k(kmg_ctx *ctx, int min_k) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_set_min_k before kmg_build_begin");
    if (ctx->n_adds) return fail(ctx, KMG_ESTATE, "the shortest k-mer length must be set before the first kmg_build_add");
    if (min_k < 1 || min_k > ctx->k) return fail(ctx, KMG_EINVAL, "min_k must lie in 1 .. k");
    if (min_k < ctx->key_len) return fail(ctx, KMG_EUNSUPPORTED, "key_len must be <= min_k");      // Kmer.key(keyLength) of a shorter k-mer throws
    ctx->min_k = min_k;
    return KMG_OK;
}

extern "C" int kmg_build_set_key_range(kmg_ctx *ctx, uint32_t first_bucket, uint32_t end_bucket) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_set_key_range before kmg_build_begin");
    if (ctx->n_adds) return fail(ctx, KMG_ESTATE, "the key range must be set before the first kmg_build_add");
    const u32 nbuckets = 1u << (2 * ctx->key_len);
    // (first == end is an empty range: the build keeps nothing and finishes with an empty database -- what a rank
    // gets when there are more ranks than populated buckets; failing there would leave the other ranks waiting)
    if (first_bucket > end_bucket || end_bucket > nbuckets) return fail(ctx, KMG_EINVAL, "key range must be first <= end <= 4^key_len");
    ctx->ranged = !(first_bucket == 0 && end_bucket == nbuckets);
    ctx->range_lo = first_bucket;
    ctx->range_hi = end_bucket;
    return KMG_OK;
}

static int ensure_record_capacity(kmg_ctx *ctx, u64 need) {
    if (need <= ctx->rec_cap && ctx->rec_keys) return KMG_OK;
    u64 ncap = std::max(need, ctx->rec_cap + ctx->rec_cap / 2);
    const size_t vb = val_bytes(ctx->kind);
    const bool wide = ctx->k > 32, lens = wide && ctx->min_k < ctx->k;
    void *nk = nullptr, *nv = nullptr, *nl = nullptr, *nn = nullptr;
    if (ctx->pool.alloc(&nk, ncap * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "record buffer allocation failed");
    if (ctx->pool.alloc(&nv, ncap * vb) != KMG_OK || (wide && ctx->pool.alloc(&nl, ncap * 8) != KMG_OK) ||
        (lens && ctx->pool.alloc(&nn, ncap + 64) != KMG_OK)) {
        ctx->pool.release(nk);
        ctx->pool.release(nv);
        ctx->pool.release(nl);
        return fail(ctx, KMG_ENOMEM, "record buffer allocation failed");
    }
    // (every record has k bases unless the short k-mers' write pass says otherwise)
    if (lens) KMG_HIP(hipMemsetAsync(nn, ctx->k, ncap + 64, ctx->stream));
    if (ctx->rec_keys && ctx->rec_upper) {
        KMG_HIP(hipMemcpyAsync(nk, ctx->rec_keys, ctx->rec_upper * 8, hipMemcpyDeviceToDevice, ctx->stream));
        KMG_HIP(hipMemcpyAsync(nv, ctx->rec_vals, ctx->rec_upper * vb, hipMemcpyDeviceToDevice, ctx->stream));
        if (wide) KMG_HIP(hipMemcpyAsync(nl, ctx->rec_lo, ctx->rec_upper * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (lens) KMG_HIP(hipMemcpyAsync(nn, ctx->rec_len, ctx->rec_upper, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ctx->pool.release(ctx->rec_keys);
    ctx->pool.release(ctx->rec_vals);
    ctx->pool.release(ctx->rec_lo);
    ctx->pool.release(ctx->rec_len);
    ctx->rec_keys = (u64 *)nk;
    ctx->rec_vals = nv;
    ctx->rec_lo = (u64 *)nl;
    ctx->rec_len = (uint8_t *)nn;
    ctx->rec_cap = ncap;
    return KMG_OK;
}

// Segment plan of a build's sort (radix.h, SegPlan): the whole key space, or the slice a key-range build keeps
static SegPlan build_seg_plan(const kmg_ctx *ctx, int pbits) {
    if (!ctx->ranged || pbits <= 0) return whole_space_plan(pbits);
    const int s = 64 - 2 * ctx->key_len;
    const u64 first = (u64)ctx->range_lo << s, last = ((u64)ctx->range_hi << s) - 1;      // (range_hi = 4^key_len wraps to ~0)
    return key_slice_plan(pbits, first, last);
}
// prefix bits for n records of this build: planned for the density inside the slice a key-range build keeps
static int build_pbits(const kmg_ctx *ctx, u64 n, int kbits) {
    if (!ctx->ranged) return choose_pbits(n, kbits);
    const u64 n_dense = n * ((u64)1 << (2 * ctx->key_len)) / std::max<u32>(ctx->range_hi - ctx->range_lo, 1u);
    return choose_pbits(n_dense, kbits, 30);
}

// A batch of a build of two-word k-mers (k > 32, wide.h): count pass, numbering, scan, write pass -- the records are
// appended in window order behind those of the earlier batches.  Arguments checked by build_add_locked.
static int build_add_wide(kmg_ctx *ctx, const uint8_t *d_bases, u64 nbases, const u64 *d_seq_off, const int32_t *d_seq_id,
                          const int64_t *d_seq_tok, u64 nseq, int32_t *d_read_index_out) {
    const bool reads = ctx->kind == KMG_PAYLOAD_READPOS;
    const bool want_flags = !reads && d_read_index_out;
    const u64 upper = ctx->rec_upper + nbases * (ctx->rc ? 2 : 1);
    KMG_TRY(ensure_record_capacity(ctx, upper));
    const u32 ntiles = div_up_u64(nbases, EX_TILE);
    PoolBuf tile_seq(ctx), tile_count(ctx), tile_fwd(ctx), tile_off(ctx), agg(ctx), has_kmer(ctx), read_index(ctx);
    KMG_ALLOC(tile_seq, (size_t)(ntiles + 1) * 4);
    KMG_ALLOC(tile_count, (size_t)ntiles * 4);
    KMG_ALLOC(tile_fwd, (size_t)ntiles * 4);
    KMG_ALLOC(tile_off, (size_t)(ntiles + 1) * 8);
    KMG_ALLOC(agg, (size_t)(std::max<u64>(div_up_u64(ntiles, SCAN_TILE), div_up_u64(nseq, SCAN_TILE)) + 1) * 8);
    WideArgs wa;
    memset(&wa, 0, sizeof wa);
    ExtractArgs &a = wa.e;
    a.bases = d_bases; a.nbases = nbases; a.seq_off = d_seq_off; a.nseq = nseq; a.seq_id = d_seq_id;
    a.tile_seq = tile_seq.as<u32>(); a.k = ctx->k; a.rc = ctx->rc; a.kind = ctx->kind;
    a.dust = ctx->dust; a.runofsame = ctx->runofsame;
    a.ranged = ctx->ranged ? 1 : 0; a.range_shift = 64 - 2 * ctx->key_len; a.range_lo = ctx->range_lo; a.range_hi = ctx->range_hi;
    a.tile_count = tile_count.as<u32>(); a.tile_fwd = tile_fwd.as<u32>(); a.tile_off = tile_off.as<u64>();
    a.out_keys = ctx->rec_keys; a.out_vals = ctx->rec_vals; a.state = ctx->d_state;
    a.pos_bits = 16; a.pos_mask = 0xFFFFu;
    wa.out_lo = ctx->rec_lo;
    {
        ProfScope ps(ctx, KP_EXTRACT_TILESEQ, 0);
        hipLaunchKernelGGL(extract_tileseq_kernel, dim3(div_up_u64(ntiles + 1, 256)), dim3(256), 0, ctx->stream,
                           d_seq_off, nseq, tile_seq.as<u32>(), (u32 *)nullptr, ntiles + 1);
    }
    int32_t *ridx = nullptr;
    if (want_flags || reads) {
        KMG_ALLOC(has_kmer, nseq);
        KMG_HIP(hipMemsetAsync(has_kmer.p, 0, nseq, ctx->stream));
        a.has_kmer = has_kmer.as<uint8_t>();
    }
    if (reads) {
        if (d_read_index_out) ridx = d_read_index_out;
        else { KMG_ALLOC(read_index, nseq * 4); ridx = read_index.as<int32_t>(); }
        a.read_index = ridx;
    }
    ctx->prepart = false; ctx->plan_pbits = 0; ctx->snap_valid = false;
    ctx->n_adds++;
    {
        ProfScope ps(ctx, KP_EXTRACT_COUNT, nbases);
        hipLaunchKernelGGL((wide_extract_kernel<false>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, wa);
        hipLaunchKernelGGL(sum_u32_kernel, dim3(div_up_u64(ntiles, 2048)), dim3(256), 0, ctx->stream, tile_fwd.as<u32>(),
                           ntiles, ctx->d_state + ST_NWIN);
    }
    // a build of several lengths: count the k-mers shorter than k -- before the reads are numbered, because a read gets its
    // index at its first k-mer of ANY length
    const bool var = ctx->min_k < ctx->k;
    PoolBuf sh_count(ctx), sh_fwd(ctx), sh_off(ctx);
    WideShortArgs sa;
    memset(&sa, 0, sizeof sa);
    if (var) {
        KMG_ALLOC(sh_count, (size_t)ntiles * 4);
        KMG_ALLOC(sh_fwd, (size_t)ntiles * 4);
        KMG_ALLOC(sh_off, (size_t)(ntiles + 1) * 8);
        sa.e = a;
        sa.min_k = ctx->min_k;
        sa.e.tile_count = sh_count.as<u32>(); sa.e.tile_fwd = sh_fwd.as<u32>(); sa.e.tile_off = sh_off.as<u64>();
        ProfScope ps(ctx, KP_EXTRACT_COUNT, nbases);
        hipLaunchKernelGGL((wide_short_kernel<false>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, sa);
        hipLaunchKernelGGL(sum_u32_kernel, dim3(div_up_u64(ntiles, 2048)), dim3(256), 0, ctx->stream, sh_fwd.as<u32>(),
                           ntiles, ctx->d_state + ST_NWIN);
    }
    if (reads) {
        ProfScope ps(ctx, KP_NUMBER_READS, 0);
        const u32 nr_tiles = div_up_u64(nseq, NR_TILE);
        u32 epoch = 0;
        KMG_TRY(chained_begin(ctx, nr_tiles, &epoch));
        hipLaunchKernelGGL(number_reads_kernel, dim3(nr_tiles), dim3(NR_BLOCK), 0, ctx->stream, has_kmer.as<uint8_t>(), d_seq_tok, nseq,
                           ridx, ctx->d_state, ctx->scan_status, ctx->scan_ctl, epoch, nr_tiles);
    } else {
        hipLaunchKernelGGL(id_bits_kernel, dim3(std::min<u32>(div_up_u64(nseq, 256), 1024)), dim3(256), 0, ctx->stream,
                           d_seq_id, nseq, ctx->d_state);
        if (want_flags)
            hipLaunchKernelGGL(has_kmer_flags_kernel, dim3(div_up_u64(nseq, 256)), dim3(256), 0, ctx->stream, has_kmer.as<uint8_t>(),
                               nseq, d_read_index_out);
    }
    KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, tile_count.as<u32>(), tile_off.as<u64>(), ntiles, OpSum(), 0ull,
                                                agg.as<u64>(), nullptr, true, KP_SCAN)));
    {
        ProfScope ps(ctx, KP_EXTRACT_WRITE, nbases);
        hipLaunchKernelGGL((wide_extract_kernel<true>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, wa);
        hipLaunchKernelGGL(extract_advance_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, tile_off.as<u64>(), ntiles);
    }
    KMG_HIP(hipGetLastError());
    ctx->rec_upper = upper;
    if (var) {
        // the short k-mers go behind the batch's full windows, their lengths beside them
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, sh_count.as<u32>(), sh_off.as<u64>(), ntiles, OpSum(), 0ull,
                                                    agg.as<u64>(), nullptr, true, KP_SCAN)));
        u64 added = 0;
        KMG_HIP(read_back(ctx, &added, sh_off.as<u64>() + ntiles, 8));
        if (added) {
            KMG_TRY(ensure_record_capacity(ctx, upper + added));
            sa.e.read_index = ridx;
            sa.e.out_keys = ctx->rec_keys; sa.e.out_vals = ctx->rec_vals; sa.out_lo = ctx->rec_lo; sa.out_len = ctx->rec_len;
            ProfScope ps(ctx, KP_EXTRACT_WRITE, nbases + added * (17 + val_bytes(ctx->kind)));
            hipLaunchKernelGGL((wide_short_kernel<true>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, sa);
            hipLaunchKernelGGL(extract_advance_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, sh_off.as<u64>(), ntiles);
            KMG_HIP(hipGetLastError());
            ctx->rec_upper = upper + added;
        }
    }
    return KMG_OK;
}

// All pointers are device pointers.  Caller holds ctx->mu.
static int build_add_locked(kmg_ctx *ctx, const uint8_t *d_bases, u64 nbases, const u64 *d_seq_off,
                            const int32_t *d_seq_id, const int64_t *d_seq_tok, u64 nseq, int32_t *d_read_index_out) {
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_add before kmg_build_begin");
    const bool reads = ctx->kind == KMG_PAYLOAD_READPOS;
    if (nseq == 0 || nbases == 0) {
        if (d_read_index_out && nseq)
            KMG_HIP(hipMemsetAsync(d_read_index_out, 0xFF, nseq * sizeof(int32_t), ctx->stream));
        return KMG_OK;
    }
    // taxon builds: read_index_out, if given, receives 0 / -1 per sequence: it produced a k-mer / it did not
    const bool want_flags = !reads && d_read_index_out;
    if (!d_bases || !d_seq_off) return fail(ctx, KMG_EINVAL, "null bases / seq_off");
    if (!reads && !d_seq_id) return fail(ctx, KMG_EINVAL, "seq_id required for taxon payload");
    if (reads && !d_seq_tok) return fail(ctx, KMG_EINVAL, "seq_tok required for read payload");
    if (nseq >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "too many sequences in one batch");
    if (nbases >= (u64)0xFFFFFFF0ull * 1024) return fail(ctx, KMG_EUNSUPPORTED, "batch too large");

    if (ctx->k > 32) return build_add_wide(ctx, d_bases, nbases, d_seq_off, d_seq_id, d_seq_tok, nseq, d_read_index_out);

    if (ctx->compact_pb) {
        // a second batch: the first one's 32-bit payloads go back to the 64-bit form every other path uses
        void *wide = nullptr;
        if (ctx->pool.alloc(&wide, ctx->rec_cap * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "record buffer allocation failed");
        if (ctx->rec_upper)
            hipLaunchKernelGGL(rd_widen_kernel, dim3(div_up_u64(ctx->rec_upper, 256)), dim3(256), 0, ctx->stream,
                               (const u32 *)ctx->rec_vals, ctx->rec_upper, ctx->compact_pb, (u64 *)wide);
        ctx->pool.release(ctx->rec_vals);
        ctx->rec_vals = wide;
        ctx->compact_pb = 0;
    }
    // Record buffers: sized for every window of the batch -- except in a key-range build, which keeps a fraction
    // and exists for inputs whose full record set does not fit: there the count pass's total is fetched first
    // (`exact` below) and the buffers hold exactly what the write pass will store.
    const bool exact = ctx->ranged;
    u64 upper = ctx->rec_upper + nbases * (ctx->rc ? 2 : 1);
    if (!exact) KMG_TRY(ensure_record_capacity(ctx, upper));

    const u32 ntiles = div_up_u64(nbases, EX_TILE);
    PoolBuf tile_seq(ctx), tile_count(ctx), tile_fwd(ctx), tile_off(ctx), agg(ctx), has_kmer(ctx), read_index(ctx);
    KMG_ALLOC(tile_seq, (size_t)(ntiles + 1) * 8);   // tile_seq[ntiles + 1] then tile_end[ntiles + 1]
    KMG_ALLOC(tile_count, (size_t)ntiles * 4);
    KMG_ALLOC(tile_fwd, (size_t)ntiles * 4);
    KMG_ALLOC(tile_off, (size_t)(ntiles + 1) * 8);
    KMG_ALLOC(agg, (size_t)(std::max<u64>(div_up_u64(ntiles, SCAN_TILE), div_up_u64(nseq, SCAN_TILE)) + 1) * 8);

    ExtractArgs a;
    memset(&a, 0, sizeof a);
    a.bases = d_bases; a.nbases = nbases; a.seq_off = d_seq_off; a.nseq = nseq; a.seq_id = d_seq_id;
    a.tile_seq = tile_seq.as<u32>(); a.tile_end = tile_seq.as<u32>() + ntiles + 1; a.k = ctx->k; a.rc = ctx->rc; a.kind = ctx->kind;
    a.dust = ctx->dust; a.runofsame = ctx->runofsame;
    a.ranged = ctx->ranged ? 1 : 0; a.range_shift = 64 - 2 * ctx->key_len; a.range_lo = ctx->range_lo; a.range_hi = ctx->range_hi;
    a.tile_count = tile_count.as<u32>(); a.tile_fwd = tile_fwd.as<u32>(); a.tile_off = tile_off.as<u64>();
    a.out_keys = ctx->rec_keys; a.out_vals = ctx->rec_vals; a.state = ctx->d_state;
    a.pos_bits = 16; a.pos_mask = 0xFFFFu;

    {
        ProfScope ps(ctx, KP_EXTRACT_TILESEQ, 0);
        hipLaunchKernelGGL(extract_tileseq_kernel, dim3(div_up_u64(ntiles + 1, 256)), dim3(256), 0, ctx->stream,
                           d_seq_off, nseq, tile_seq.as<u32>(), tile_seq.as<u32>() + ntiles + 1, ntiles + 1);
    }
    int32_t *ridx = nullptr;
    if (want_flags) {
        KMG_ALLOC(has_kmer, nseq);
        KMG_HIP(hipMemsetAsync(has_kmer.p, 0, nseq, ctx->stream));
        a.has_kmer = has_kmer.as<uint8_t>();
    }
    if (reads) {
        KMG_ALLOC(has_kmer, nseq);
        KMG_HIP(hipMemsetAsync(has_kmer.p, 0, nseq, ctx->stream));
        a.has_kmer = has_kmer.as<uint8_t>();
        if (d_read_index_out) ridx = d_read_index_out;
        else { KMG_ALLOC(read_index, nseq * 4); ridx = read_index.as<int32_t>(); }
        a.read_index = ridx;
    }
    // First batch: plan the sort's prefix bits from the record upper bound and let the extractor write the
    // records partitioned by the first prefix digit.  Later batches append in window order (and void it).
    // (planned from the expected record count -- every sequence long enough, every window valid -- rather than
    // from the upper bound: any plan is correct, a tight one fills the local sort's segments)
    const u64 short_by = std::min<u64>(nbases, nseq * (u64)(ctx->k - 1));
    const u64 expect = std::max<u64>(nbases - short_by, nbases / 16) * (ctx->rc ? 2 : 1);
    // (a variable-length build's full windows take the same path; its shorter k-mers are counted before the reads are
    // numbered and before the state snapshot below, and written into buffers of their own afterwards)
    const bool var = ctx->min_k < ctx->k;
    const int plan = ctx->n_adds == 0 ? choose_pbits(expect, 2 * ctx->k) : 0;
    // (a key-range build whose keys agree on all of the planned prefix bits has no global pass to fuse)
    const std::vector<SortPass> plan_digits = prefix_digits(plan, build_seg_plan(ctx, plan).shared_top);
    const bool part = plan > 0 && plan <= 32 && nbases < 0xFFFFFFF0ull / 2 && !plan_digits.empty();   // (<= 32: the partition digit lies in the key's upper word)
    ctx->prepart = part;
    ctx->plan_pbits = plan;
    ctx->n_adds++;
    ctx->snap_valid = false;
    PartArgs pa;
    memset(&pa, 0, sizeof pa);
    PoolBuf table(ctx), tagg(ctx), mx(ctx), img(ctx);
    bool len_pending = false;
    if (part && reads) {
        // Compact payloads: when every (read, pos) of the batch fits 32 bits -- pos < 2^pb from the longest
        // sequence, read < nseq < 2^(32 - pb) -- the records carry read << pb | pos in 4 bytes through the
        // partition, the sort and into the database (12-byte records instead of 16).  The longest sequence is
        // found first and fetched behind an event; the host looks at it only before the write pass, with the
        // count pass, the numbering and the table scan already queued.
        KMG_ALLOC(mx, 64);
        KMG_HIP(hipMemsetAsync(mx.p, 0, 4, ctx->stream));
        hipLaunchKernelGGL(seq_maxlen_kernel, dim3(std::min<u32>(div_up_u64(nseq, 1024), 1024)), dim3(256), 0, ctx->stream,
                           d_seq_off, nseq, mx.as<u32>());
        if (ctx->h_snap) {
            KMG_HIP(hipMemcpyAsync((char *)ctx->h_snap + 128, mx.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            KMG_HIP(hipEventRecord(ctx->ev_len, ctx->stream));
        }
        len_pending = true;
    }
    // the count pass's tile images, read back by the write pass (either kind of batch)
    KMG_ALLOC(img, (size_t)ntiles * EP_IMG_WORDS * 4);
    pa.img = img.as<u32>();
    pa.ntiles = ntiles; pa.ncols = ntiles * (ctx->rc ? 2 : 1);
    pa.dshift = 32; pa.dmask = 0;
    if (part) {
        const SortPass d0 = plan_digits[0];
        pa.dshift = d0.shift; pa.dmask = (1u << d0.bits) - 1u;
        KMG_ALLOC(table, (size_t)RS_DIGITS * pa.ncols * 4);
        KMG_ALLOC(tagg, (size_t)(div_up_u64((u64)RS_DIGITS * pa.ncols, SCAN_TILE) + 1) * 8);
        pa.table = table.as<u32>();
        // forward and RC records of a tile are written as two runs: (read, pos) order inside a k-mer is
        // restored by the payload passes
        if (reads && ctx->rc) KMG_HIP(hipMemsetAsync((char *)(ctx->d_state + ST_DUP_READS), 1, 1, ctx->stream));
    }
    const dim3 pgrid(((ntiles + 7) / 8) * 8);
    {
        ProfScope ps(ctx, KP_EXTRACT_COUNT, nbases);
        pa.e = a;
        // (the payload type only matters to the write pass)
        const bool filt = ctx->dust >= 0 || ctx->runofsame >= 0 || ctx->ranged;
        if (part) {
            if (ctx->rc) {
                if (filt) hipLaunchKernelGGL((extract_part_count_kernel<true, true>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
                else hipLaunchKernelGGL((extract_part_count_kernel<true, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            } else {
                if (filt) hipLaunchKernelGGL((extract_part_count_kernel<false, true>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
                else hipLaunchKernelGGL((extract_part_count_kernel<false, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            }
        } else {
            // a later batch of the build: the same pass without the partition digit's histograms
            if (ctx->rc) {
                if (filt) hipLaunchKernelGGL((extract_part_count_kernel<true, true, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
                else hipLaunchKernelGGL((extract_part_count_kernel<true, false, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            } else {
                if (filt) hipLaunchKernelGGL((extract_part_count_kernel<false, true, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
                else hipLaunchKernelGGL((extract_part_count_kernel<false, false, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            }
        }
        // forward windows of the batch.  (A fused batch without reverse complements holds nothing else: there the
        // total of the table scan below is both the record count and the window count, and no launch is spent on it.)
        if (!part || ctx->rc)
            hipLaunchKernelGGL(sum_u32_kernel, dim3(div_up_u64(ntiles, 2048)), dim3(256), 0, ctx->stream, tile_fwd.as<u32>(),
                               ntiles, ctx->d_state + ST_NWIN);
    }
    // variable-length build: count the k-mers shorter than k (extract_short.h) -- before the reads are numbered,
    // because a read gets its index at its first k-mer of ANY length
    PoolBuf sh_count(ctx), sh_fwd(ctx), sh_off(ctx), sh_jw(ctx);
    ShortArgs sa;
    memset(&sa, 0, sizeof sa);
    if (var) {
        KMG_ALLOC(sh_count, (size_t)ntiles * 4);
        KMG_ALLOC(sh_fwd, (size_t)ntiles * 4);
        KMG_ALLOC(sh_off, (size_t)(ntiles + 1) * 8);
        KMG_ALLOC(sh_jw, (size_t)ntiles * (EX_ROUNDS * (EX_BLOCK / 64)) * 4);
        sa.jw_count = sh_jw.as<u32>();
        sa.e = a;
        sa.min_k = ctx->min_k;
        sa.e.tile_count = sh_count.as<u32>(); sa.e.tile_fwd = sh_fwd.as<u32>(); sa.e.tile_off = sh_off.as<u64>();
        ProfScope ps(ctx, KP_EXTRACT_COUNT, nbases);
        hipLaunchKernelGGL((extract_short_kernel<false>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, sa);
        hipLaunchKernelGGL(sum_u32_kernel, dim3(div_up_u64(ntiles, 2048)), dim3(256), 0, ctx->stream, sh_fwd.as<u32>(),
                           ntiles, ctx->d_state + ST_NWIN);
    }
    if (reads) {
        // ReadIDMapping.geNext over the batch, one launch (extract.h)
        ProfScope ps(ctx, KP_NUMBER_READS, 0);
        const u32 nr_tiles = div_up_u64(nseq, NR_TILE);
        u32 epoch = 0;
        KMG_TRY(chained_begin(ctx, nr_tiles, &epoch));
        hipLaunchKernelGGL(number_reads_kernel, dim3(nr_tiles), dim3(NR_BLOCK), 0, ctx->stream, has_kmer.as<uint8_t>(), d_seq_tok, nseq,
                           ridx, ctx->d_state, ctx->scan_status, ctx->scan_ctl, epoch, nr_tiles);
    } else {
        hipLaunchKernelGGL(id_bits_kernel, dim3(std::min<u32>(div_up_u64(nseq, 256), 1024)), dim3(256), 0, ctx->stream,
                           d_seq_id, nseq, ctx->d_state);
        if (want_flags)
            hipLaunchKernelGGL(has_kmer_flags_kernel, dim3(div_up_u64(nseq, 256)), dim3(256), 0, ctx->stream, has_kmer.as<uint8_t>(),
                               nseq, d_read_index_out);
    }
    if (part) {
        // the record count is final once the tiles are counted (the write pass starts at record 0: first batch);
        // with it the build state is complete, and finish() reads this snapshot instead of stopping the stream
        // (the scan's grand total is the batch's record count: added to the state by the scan's last tile)
        KMG_TRY((device_scan<u32, u32, OpSum, true>(ctx, table.as<u32>(), table.as<u32>(), (u64)RS_DIGITS * pa.ncols, OpSum(),
                                                    0u, tagg.as<u32>(), nullptr, false, KP_SCAN, ctx->d_state + ST_NREC,
                                                    ctx->rc ? nullptr : ctx->d_state + ST_NWIN)));
        if (ctx->h_snap) {
            KMG_HIP(hipMemcpyAsync(ctx->h_snap, ctx->d_state, ST_WORDS * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
            KMG_HIP(hipEventRecord(ctx->ev_snap, ctx->stream));
            ctx->snap_valid = true;
        }
        if (exact) {
            u64 nrec = 0;
            KMG_HIP(read_back(ctx, &nrec, ctx->d_state + ST_NREC, 8));
            upper = std::max<u64>(nrec, 1);
            KMG_TRY(ensure_record_capacity(ctx, upper));
            a.out_keys = ctx->rec_keys; a.out_vals = ctx->rec_vals;
        }
        if (len_pending) {
            u32 maxlen = 0;
            if (ctx->h_snap) {
                KMG_HIP(hipEventSynchronize(ctx->ev_len));
                memcpy(&maxlen, (char *)ctx->h_snap + 128, 4);
            } else KMG_HIP(read_back(ctx, &maxlen, mx.p, 4));
            const u32 maxpos = maxlen > (u32)ctx->k ? maxlen - (u32)ctx->k : 0;
            int pb = 1;
            while (pb < 32 && (maxpos >> pb) != 0) pb++;
            if (pb <= 16 && nseq < ((u64)1 << (32 - pb))) {
                ctx->compact_pb = pb;
                a.pos_bits = pb; a.pos_mask = (1u << pb) - 1u;
            }
        }
        const bool compact = ctx->compact_pb > 0;
        ProfScope ps(ctx, KP_EXTRACT_WRITE, nbases /* + records * rec_b, added at finish when N is known */);
        pa.e = a;
        pa.out_base = 0;
#ifndef EPC_PROBE          // (timing probes of the count pass leave the table unfinished: nothing may be written from it)
        if (ctx->lds_rank_ok) {
            if (reads && !compact) hipLaunchKernelGGL((extract_part_write_kernel<u64, true>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            else hipLaunchKernelGGL((extract_part_write_kernel<u32, true>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
        } else {
            if (reads && !compact) hipLaunchKernelGGL((extract_part_write_kernel<u64, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
            else hipLaunchKernelGGL((extract_part_write_kernel<u32, false>), pgrid, dim3(EP_BLOCK), 0, ctx->stream, pa);
        }
#endif
    } else {
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, tile_count.as<u32>(), tile_off.as<u64>(), ntiles, OpSum(), 0ull,
                                                    agg.as<u64>(), nullptr, true, KP_SCAN)));
        u64 batch_added = 0;
        if (exact) {
            u64 before = 0, added = 0;
            KMG_HIP(read_back(ctx, &before, ctx->d_state + ST_NREC, 8));
            KMG_HIP(read_back(ctx, &added, tile_off.as<u64>() + ntiles, 8));
            batch_added = added;
            upper = std::max<u64>(before + added, 1);
            KMG_TRY(ensure_record_capacity(ctx, upper));
            a.out_keys = ctx->rec_keys; a.out_vals = ctx->rec_vals;
        }
        ProfScope ps(ctx, KP_EXTRACT_WRITE, nbases /* + records * rec_b, added at finish when N is known */);
        pa.e = a;
        // a batch that keeps little (key-range builds know their count by now): one wave per tile, a lane per record
        const bool sparse = exact && batch_added < (u64)ntiles * 1024;
        if (sparse) {
            const dim3 sgrid(div_up_u64(ntiles, EPS_WAVES)), sblk(64 * EPS_WAVES);
            if (reads) hipLaunchKernelGGL((extract_img_write_sparse_kernel<u64>), sgrid, sblk, 0, ctx->stream, pa);
            else hipLaunchKernelGGL((extract_img_write_sparse_kernel<u32>), sgrid, sblk, 0, ctx->stream, pa);
        } else if (reads) hipLaunchKernelGGL((extract_img_write_kernel<u64>), dim3(ntiles), dim3(EP_BLOCK), 0, ctx->stream, pa);
        else hipLaunchKernelGGL((extract_img_write_kernel<u32>), dim3(ntiles), dim3(EP_BLOCK), 0, ctx->stream, pa);
        hipLaunchKernelGGL(extract_advance_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, tile_off.as<u64>(), ntiles);
    }
    if (var) {
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, sh_count.as<u32>(), sh_off.as<u64>(), ntiles, OpSum(), 0ull,
                                                    agg.as<u64>(), nullptr, true, KP_SCAN)));
        u64 added = 0;
        KMG_HIP(read_back(ctx, &added, sh_off.as<u64>() + ntiles, 8));
        if (added) {
            const size_t vb = val_bytes(ctx->kind);
            if (ctx->sh_n + added > ctx->sh_cap || !ctx->sh_keys) {
                const u64 ncap = std::max<u64>(ctx->sh_n + added, ctx->sh_cap + ctx->sh_cap / 2);
                void *nk = nullptr, *nv = nullptr, *nl = nullptr;
                if (ctx->pool.alloc(&nk, ncap * 8) != KMG_OK || ctx->pool.alloc(&nv, ncap * vb) != KMG_OK ||
                    ctx->pool.alloc(&nl, ncap) != KMG_OK) {
                    ctx->pool.release(nk); ctx->pool.release(nv); ctx->pool.release(nl);
                    return fail(ctx, KMG_ENOMEM, "record buffer allocation failed");
                }
                if (ctx->sh_n) {
                    KMG_HIP(hipMemcpyAsync(nk, ctx->sh_keys, ctx->sh_n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                    KMG_HIP(hipMemcpyAsync(nv, ctx->sh_vals, ctx->sh_n * vb, hipMemcpyDeviceToDevice, ctx->stream));
                    KMG_HIP(hipMemcpyAsync(nl, ctx->sh_len, ctx->sh_n, hipMemcpyDeviceToDevice, ctx->stream));
                }
                ctx->pool.release(ctx->sh_keys); ctx->pool.release(ctx->sh_vals); ctx->pool.release(ctx->sh_len);
                ctx->sh_keys = (u64 *)nk; ctx->sh_vals = nv; ctx->sh_len = (uint8_t *)nl; ctx->sh_cap = ncap;
            }
            sa.e.read_index = ridx;
            sa.e.out_keys = ctx->sh_keys; sa.e.out_vals = ctx->sh_vals; sa.out_len = ctx->sh_len; sa.out_base = ctx->sh_n;
            ProfScope ps(ctx, KP_EXTRACT_WRITE, nbases + added * (9 + vb));
            hipLaunchKernelGGL((extract_short_kernel<true>), dim3(ntiles), dim3(EX_BLOCK), 0, ctx->stream, sa);
            ctx->sh_n += added;
        }
    }
    KMG_HIP(hipGetLastError());
    ctx->rec_upper = upper;
    return KMG_OK;
}

extern "C" int kmg_build_add_dev(kmg_ctx *ctx, const uint8_t *d_bases, uint64_t nbases, const uint64_t *d_seq_off,
                                 const int32_t *d_seq_id, const int64_t *d_seq_tok, uint64_t nseq,
                                 int32_t *d_read_index_out) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    return build_add_locked(ctx, d_bases, nbases, (const u64 *)d_seq_off, d_seq_id, d_seq_tok, nseq, d_read_index_out);
}

extern "C" int kmg_build_add(kmg_ctx *ctx, const uint8_t *bases, uint64_t nbases, const uint64_t *seq_off,
                             const int32_t *seq_id, const int64_t *seq_tok, uint64_t nseq, int32_t *read_index_out) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_add before kmg_build_begin");
    if (nseq == 0) return KMG_OK;
    if (!seq_off || (nbases && !bases)) return fail(ctx, KMG_EINVAL, "null bases / seq_off");
    const bool reads = ctx->kind == KMG_PAYLOAD_READPOS;
    if (!reads && !seq_id) return fail(ctx, KMG_EINVAL, "seq_id required for taxon payload");
    if (reads && !seq_tok) return fail(ctx, KMG_EINVAL, "seq_tok required for read payload");
    if (seq_off[0] != 0 || seq_off[nseq] != nbases) return fail(ctx, KMG_EINVAL, "seq_off must run from 0 to nbases");
    for (u64 i = 0; i < nseq; i++)
        if (seq_off[i] > seq_off[i + 1]) return fail(ctx, KMG_EINVAL, "seq_off must be non-decreasing");
    KMG_HIP(hipSetDevice(ctx->device));
    if (nbases == 0) {
        if (read_index_out) for (u64 i = 0; i < nseq; i++) read_index_out[i] = -1;
        return KMG_OK;
    }
    PoolBuf d_bases(ctx), d_off(ctx), d_id(ctx), d_tok(ctx), d_ridx(ctx);
    KMG_ALLOC(d_bases, nbases + 64);
    KMG_ALLOC(d_off, (nseq + 1) * 8);
    KMG_HIP(hipMemcpyAsync(d_bases.p, bases, nbases, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipMemcpyAsync(d_off.p, seq_off, (nseq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (read_index_out) KMG_ALLOC(d_ridx, nseq * 4);
    if (reads) {
        KMG_ALLOC(d_tok, nseq * 8);
        KMG_HIP(hipMemcpyAsync(d_tok.p, seq_tok, nseq * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        KMG_ALLOC(d_id, nseq * 4);
        KMG_HIP(hipMemcpyAsync(d_id.p, seq_id, nseq * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    KMG_TRY(build_add_locked(ctx, d_bases.as<uint8_t>(), nbases, d_off.as<u64>(), d_id.as<int32_t>(),
                             d_tok.as<int64_t>(), nseq, d_ridx.as<int32_t>()));
    if (read_index_out) {
        KMG_HIP(hipMemcpyAsync(read_index_out, d_ridx.p, nseq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    // the host buffers may be freed or moved by the caller as soon as we return
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

extern "C" int kmg_build_add_packed(kmg_ctx *ctx, const uint8_t *packed, uint64_t nbytes, const uint64_t *seq_off,
                                    const uint64_t *byte_off, const int32_t *seq_id, uint64_t nseq) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_add_packed before kmg_build_begin");
    if (ctx->kind != KMG_PAYLOAD_TAXON) return fail(ctx, KMG_EINVAL, "packed input carries taxon ids (MakeDatabase -p)");
    if (nseq == 0) return KMG_OK;
    if (!seq_off || !byte_off || !seq_id || (nbytes && !packed)) return fail(ctx, KMG_EINVAL, "null argument");
    if (seq_off[0] != 0 || byte_off[0] != 0 || byte_off[nseq] != nbytes) return fail(ctx, KMG_EINVAL, "offsets must run from 0 to the end");
    for (u64 i = 0; i < nseq; i++) {
        if (seq_off[i] > seq_off[i + 1] || byte_off[i] > byte_off[i + 1])
            return fail(ctx, KMG_EINVAL, "offsets must be non-decreasing");
        if ((seq_off[i + 1] - seq_off[i] + 3) / 4 != byte_off[i + 1] - byte_off[i])
            return fail(ctx, KMG_EINVAL, "a sequence of n bases takes ceil(n / 4) bytes");
    }
    const u64 nbases = seq_off[nseq];
    if (nbases == 0) return KMG_OK;
    KMG_HIP(hipSetDevice(ctx->device));
    PoolBuf d_packed(ctx), d_boff(ctx), d_off(ctx), d_id(ctx), d_bases(ctx);
    KMG_ALLOC(d_packed, nbytes + 64);
    KMG_ALLOC(d_boff, (nseq + 1) * 8);
    KMG_ALLOC(d_off, (nseq + 1) * 8);
    KMG_ALLOC(d_id, nseq * 4);
    KMG_ALLOC(d_bases, nbases + 64);
    KMG_HIP(hipMemcpyAsync(d_packed.p, packed, nbytes, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipMemcpyAsync(d_boff.p, byte_off, (nseq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipMemcpyAsync(d_off.p, seq_off, (nseq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipMemcpyAsync(d_id.p, seq_id, nseq * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(unpack2bit_kernel, dim3(div_up_u64(nbytes, 256)), dim3(256), 0, ctx->stream, d_packed.as<uint8_t>(), nbytes,
                       d_boff.as<u64>(), d_off.as<u64>(), nseq, d_bases.as<uint8_t>());
    KMG_HIP(hipGetLastError());
    KMG_TRY(build_add_locked(ctx, d_bases.as<uint8_t>(), nbases, d_off.as<u64>(), d_id.as<int32_t>(), nullptr, nseq, nullptr));
    KMG_HIP(hipStreamSynchronize(ctx->stream));      // the host buffers may go away as soon as we return
    return KMG_OK;
}

// `user` has just queued work on its stream that reads `db`, a database of another context: note it down
// (see kmg_db::foreign_use).  Re-recording the context's event is enough -- its stream runs in order.
static int note_foreign_use(kmg_ctx *user, const kmg_db *cdb) {
    kmg_db *db = const_cast<kmg_db *>(cdb);
    if (db->ctx == user) return KMG_OK;
    kmg_ctx *ctx = user;
    std::lock_guard<std::mutex> g(db->foreign_mu);
    hipEvent_t &ev = db->foreign_use[user];
    if (!ev) KMG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    KMG_HIP(hipEventRecord(ev, user->stream));
    return KMG_OK;
}

static void db_release(kmg_db *db) {
    if (!db) return;
    kmg_ctx *ctx = db->ctx;
    {
        // every reuse of the arrays happens on the owner's stream (or after a hipFree, which waits for the device)
        std::lock_guard<std::mutex> g(db->foreign_mu);
        for (auto &kv : db->foreign_use) {
            if (hipStreamWaitEvent(ctx->stream, kv.second, 0) != hipSuccess) (void)hipEventSynchronize(kv.second);
            (void)hipEventDestroy(kv.second);     // (released by the runtime once the wait has been served)
        }
        db->foreign_use.clear();
    }
    ctx->pool.release(db->keys);
    ctx->pool.release(db->keys_lo);
    ctx->pool.release(db->lens);
    ctx->pool.release(db->val_off);
    ctx->pool.release(db->val_a);
    ctx->pool.release(db->val_b);
    ctx->pool.release(db->vals_packed);
    ctx->pool.release(db->vals_packed32);
    ctx->pool.release(db->bucket_off);
    ctx->pool.release(db->near_off);
    ctx->pool.release(db->near_log_s);
    ctx->pool.release(db->near_log_ref);
    ctx->pool.release(db->near_log_dist);
    for (kmg_db *part : db->parts) db_release(part);
    delete db;
}

// a copy of the arrays the trie search reads (taxon databases), for a merged set that keeps its files (see kmg_db::parts)
static int clone_for_trie(kmg_ctx *ctx, const kmg_db *src, kmg_db **out) {
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> guard(db, db_release);
    db->ctx = ctx; db->k = src->k; db->min_k = src->min_k; db->key_len = src->key_len; db->rc = src->rc; db->kind = src->kind;
    db->U = src->U; db->V = src->V;
    const u32 nbuckets = 1u << (2 * src->key_len);
    struct { void **dst; const void *from; size_t bytes; } arrays[] = {
        {(void **)&db->keys, src->keys, (src->U + 1) * 8},       {(void **)&db->lens, src->lens, src->lens ? src->U + 8 : 0},
        {(void **)&db->keys_lo, src->keys_lo, src->keys_lo ? (src->U + 1) * 8 : 0},
        {(void **)&db->val_off, src->val_off, (src->U + 2) * 8}, {(void **)&db->val_a, src->val_a, (src->V + 1) * 4},
        {(void **)&db->val_b, src->val_b, (src->V + 1) * 4},     {(void **)&db->bucket_off, src->bucket_off, ((size_t)nbuckets + 1) * 8}};
    for (auto &a : arrays) {
        if (!a.bytes || !a.from) continue;
        void *p;
        if (ctx->pool.alloc(&p, a.bytes) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        *a.dst = p;
    }
    // (element counts, not the padded sizes above: the sources hold exactly these)
    if (src->U) KMG_HIP(hipMemcpyAsync(db->keys, src->keys, src->U * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (src->lens && src->U) KMG_HIP(hipMemcpyAsync(db->lens, src->lens, src->U, hipMemcpyDeviceToDevice, ctx->stream));
    if (src->keys_lo && src->U) KMG_HIP(hipMemcpyAsync(db->keys_lo, src->keys_lo, src->U * 8, hipMemcpyDeviceToDevice, ctx->stream));
    KMG_HIP(hipMemcpyAsync(db->val_off, src->val_off, (src->U + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (src->V) {
        KMG_HIP(hipMemcpyAsync(db->val_a, src->val_a, src->V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        KMG_HIP(hipMemcpyAsync(db->val_b, src->val_b, src->V * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    KMG_HIP(hipMemcpyAsync(db->bucket_off, src->bucket_off, ((size_t)nbuckets + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    *out = guard.release();
    return KMG_OK;
}

extern "C" void kmg_db_free(kmg_db *db) {
    if (!db) return;
    kmg_ctx *ctx = db->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    (void)hipSetDevice(ctx->device);
    db_release(db);
}

template <typename V, bool IS_REF>
static int reduce_records(kmg_ctx *ctx, const u64 *keys, const V *vals, u32 n, kmg_db *db, u64 *d_totals) {
    // allocate the CSR at its upper bound (n) so that no host round trip sits between count and write
    void *p;
    if (ctx->pool.alloc(&p, ((size_t)n + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, ((size_t)n + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_off = (u64 *)p;
    if (ctx->pool.alloc(&p, ((size_t)n + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_a = (int32_t *)p;
    if (ctx->pool.alloc(&p, ((size_t)n + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_b = (int32_t *)p;
    const u32 nbuckets = 1u << (2 * db->key_len);
    if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->bucket_off = (u64 *)p;

    const u32 ntiles = div_up_u64(n, RD_TILE);
    PoolBuf cnt_k(ctx), cnt_v(ctx), off_k(ctx), off_v(ctx), agg(ctx), vstart(ctx);
    KMG_ALLOC(cnt_k, (size_t)(ntiles + 1) * 4);
    KMG_ALLOC(cnt_v, (size_t)(ntiles + 1) * 4);
    KMG_ALLOC(off_k, (size_t)(ntiles + 1) * 8);
    KMG_ALLOC(off_v, (size_t)(ntiles + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(ntiles, SCAN_TILE) + 1) * 8);
    if (IS_REF) KMG_ALLOC(vstart, ((size_t)n + 1) * 4);
    const u64 rec_b = 8 + sizeof(V);
    if (n > 0) {
        {
            ProfScope ps(ctx, KP_REDUCE_FLAG, (u64)n * rec_b);
            hipLaunchKernelGGL((rd_count_kernel<V, IS_REF>), dim3(ntiles), dim3(RD_BLOCK), 0, ctx->stream, keys, vals, n,
                               cnt_k.as<u32>(), cnt_v.as<u32>());
        }
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt_k.as<u32>(), off_k.as<u64>(), ntiles, OpSum(), 0ull,
                                                    agg.as<u64>(), d_totals + 0, false, KP_SCAN)));
        if (IS_REF) {
            KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt_v.as<u32>(), off_v.as<u64>(), ntiles, OpSum(), 0ull,
                                                        agg.as<u64>(), d_totals + 1, false, KP_SCAN)));
        } else {
            hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, d_totals + 1, (u64)n);
        }
        {
            ProfScope ps(ctx, KP_REDUCE_WRITE, (u64)n * rec_b /* + CSR bytes, added when U,V are known */);
            hipLaunchKernelGGL((rd_write_kernel<V, IS_REF>), dim3(ntiles), dim3(RD_BLOCK), 0, ctx->stream, keys, vals, n,
                               off_k.as<u64>(), off_v.as<u64>(), db->keys, db->val_off, db->val_a, db->val_b,
                               vstart.as<u32>());
            if (IS_REF)
                hipLaunchKernelGGL(rd_counts_kernel, dim3(std::min<u32>(div_up_u64(n, 256), 4096)), dim3(256), 0,
                                   ctx->stream, vstart.as<u32>(), d_totals, n, db->val_b);
        }
    } else {
        KMG_HIP(hipMemsetAsync(d_totals, 0, 16, ctx->stream));
    }
    {
        ProfScope ps(ctx, KP_BUCKETS, 0);
        hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                           d_totals, db->val_off, db->bucket_off, nbuckets, db->key_len);
    }
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// Read DB: k-mer runs from the sorted keys alone; the sorted payload array becomes the value array.
static int reduce_read_keys(kmg_ctx *ctx, const u64 *keys, u32 n, kmg_db *db, u64 *d_totals) {
    void *p;
    if (ctx->pool.alloc(&p, ((size_t)n + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, ((size_t)n + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_off = (u64 *)p;
    const u32 nbuckets = 1u << (2 * db->key_len);
    if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->bucket_off = (u64 *)p;
    const u32 ntiles = div_up_u64(n, RDK_TILE);
    PoolBuf cnt_k(ctx), off_k(ctx), agg(ctx);
    KMG_ALLOC(cnt_k, (size_t)(ntiles + 1) * 4);
    KMG_ALLOC(off_k, (size_t)(ntiles + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(ntiles, SCAN_TILE) + 1) * 8);
    if (n > 0) {
        {
            ProfScope ps(ctx, KP_REDUCE_FLAG, (u64)n * 8);
            hipLaunchKernelGGL(rdk_count_kernel, dim3(ntiles), dim3(RD_BLOCK), 0, ctx->stream, keys, n, cnt_k.as<u32>());
        }
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt_k.as<u32>(), off_k.as<u64>(), ntiles, OpSum(), 0ull,
                                                    agg.as<u64>(), d_totals + 0, false, KP_SCAN)));
        hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, d_totals + 1, (u64)n);
        {
            ProfScope ps(ctx, KP_REDUCE_WRITE, (u64)n * 8 /* + 16 B per unique k-mer, added when U is known */);
            hipLaunchKernelGGL(rdk_write_kernel, dim3(ntiles), dim3(RD_BLOCK), 0, ctx->stream, keys, n, off_k.as<u64>(),
                               db->keys, db->val_off);
        }
    } else {
        KMG_HIP(hipMemsetAsync(d_totals, 0, 16, ctx->stream));
    }
    {
        ProfScope ps(ctx, KP_BUCKETS, 0);
        hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                           d_totals, db->val_off, db->bucket_off, nbuckets, db->key_len);
    }
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// Read DB whose k-mer runs were already found by the local sort (SortHeads): scan the per-segment counts and
// gather the distinct k-mers and their first-record indices into the DB arrays.
static int reduce_read_heads(kmg_ctx *ctx, const SortHeads &h, u32 n, kmg_db *db, u64 *d_totals) {
    void *p;
    if (ctx->pool.alloc(&p, ((size_t)n + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, ((size_t)n + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_off = (u64 *)p;
    const u32 nbuckets = 1u << (2 * db->key_len);
    if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->bucket_off = (u64 *)p;
    PoolBuf ubase(ctx), agg(ctx);
    KMG_ALLOC(ubase, ((size_t)h.nseg + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(h.nseg, SCAN_TILE) + 1) * 8);
    KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, h.ucount, ubase.as<u64>(), h.nseg, OpSum(), 0ull, agg.as<u64>(),
                                                d_totals + 0, false, KP_SCAN)));
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, d_totals + 1, (u64)n);
    {
        ProfScope ps(ctx, KP_REDUCE_WRITE, 0 /* 32 B per unique k-mer, added when U is known */);
        if (h.first_bytes == 8)
            hipLaunchKernelGGL((ls_heads_gather_kernel<u64>), dim3(h.nseg), dim3(256), 0, ctx->stream, h.head_keys,
                               (const u64 *)h.head_first, h.seg_off, h.ucount, ubase.as<u64>(), db->keys, db->val_off);
        else
            hipLaunchKernelGGL((ls_heads_gather_kernel<u32>), dim3(h.nseg), dim3(256), 0, ctx->stream, h.head_keys,
                               (const u32 *)h.head_first, h.seg_off, h.ucount, ubase.as<u64>(), db->keys, db->val_off);
    }
    {
        ProfScope ps(ctx, KP_BUCKETS, 0);
        hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                           d_totals, db->val_off, db->bucket_off, nbuckets, db->key_len);
    }
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// val_a / val_b of a read DB are materialised from the packed payloads the first time they are needed.
// ctx must be the database's own context (its pool holds the arrays, its lock guards the handle): every caller
// either takes db->ctx or has rejected foreign databases.
static int ensure_vals(kmg_ctx *ctx, kmg_db *db) {
    if (db->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    if (db->val_a || (!db->vals_packed && !db->vals_packed32)) return KMG_OK;
    void *a = nullptr, *b = nullptr;
    if (ctx->pool.alloc(&a, (db->V + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    if (ctx->pool.alloc(&b, (db->V + 1) * 4) != KMG_OK) { ctx->pool.release(a); return fail(ctx, KMG_ENOMEM, "db allocation failed"); }
    db->val_a = (int32_t *)a;
    db->val_b = (int32_t *)b;
    if (db->V && db->vals_packed)
        hipLaunchKernelGGL(rd_unpack_kernel, dim3(div_up_u64(db->V, 256)), dim3(256), 0, ctx->stream, db->vals_packed, db->V,
                           db->val_a, db->val_b);
    else if (db->V)
        hipLaunchKernelGGL(rd_unpack32_kernel, dim3(div_up_u64(db->V, 256)), dim3(256), 0, ctx->stream, db->vals_packed32, db->V,
                           db->pos_bits, db->val_a, db->val_b);
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// Sort + dedup the n records (keys, vals) of k-mers of db->k bases into db's CSR arrays (db->k / key_len / kind set by
// the caller).  When the sorted payloads become a read database's value array, `vals` changes owner and is nulled.
static int sort_reduce_class(kmg_ctx *ctx, u64 *keys, void *&vals, u32 n, bool compact, int pos_bits,
                             const std::vector<SortPass> &payload, const SegPlan &plan, bool prepart, kmg_db *db, bool *heads_done,
                             int key_bits = 0 /* bits of the keys that count, from the top; 0: 2 * db->k */) {
    const bool reads = db->kind == KMG_PAYLOAD_READPOS;
    const size_t vb = compact ? 4 : val_bytes(db->kind);
    PoolBuf k1(ctx), v1(ctx), table(ctx), agg(ctx), totals(ctx), seg(ctx), ucount(ctx);
    SortHeads heads;
    memset(&heads, 0, sizeof heads);
    const u32 ntiles = n ? div_up_u64(n, RS_TILE) : 1;
    KMG_ALLOC(totals, 64);
    int r = KMG_OK;
    bool in1 = false;
    if (n > 0) {
        if (!keys || !vals) return fail(ctx, KMG_ESTATE, "no record buffer");
        KMG_ALLOC(k1, (size_t)n * 8);
        KMG_ALLOC(v1, (size_t)n * vb);
        KMG_ALLOC(table, (size_t)RS_DIGITS * ntiles * 4);
        KMG_ALLOC(agg, (size_t)(div_up_u64((u64)RS_DIGITS * ntiles, SCAN_TILE) + 1) * 8);
        KMG_ALLOC(seg, ((size_t)plan.nseg + 4) * 4);
        const int low_bit = 64 - (key_bits ? key_bits : 2 * db->k);
        if (reads) {
            KMG_ALLOC(ucount, ((size_t)plan.nseg + 1) * 4);
            heads.ucount = ucount.as<u32>();
            if (compact) {
                ctx->sort_as_reads = true;
                r = hybrid_sort_records<u32>(ctx, keys, (u32 *)vals, k1.as<u64>(), v1.as<u32>(), n, low_bit, payload, table.as<u32>(), agg.as<u32>(), seg.as<u32>(), plan, prepart, &in1, &heads);
                ctx->sort_as_reads = false;
            } else
                r = hybrid_sort_records<u64>(ctx, keys, (u64 *)vals, k1.as<u64>(), v1.as<u64>(), n, low_bit, payload, table.as<u32>(), agg.as<u32>(), seg.as<u32>(), plan, prepart, &in1, &heads);
        }
        else r = hybrid_sort_records<u32>(ctx, keys, (u32 *)vals, k1.as<u64>(), v1.as<u32>(), n, low_bit, payload, table.as<u32>(), agg.as<u32>(), seg.as<u32>(), plan, prepart, &in1);
    }
    if (r == KMG_OK) {
        const u64 *sk = in1 ? k1.as<u64>() : keys;
        const void *sv = in1 ? v1.p : vals;
        if (reads) {
            r = heads.done ? reduce_read_heads(ctx, heads, n, db, totals.as<u64>())
                           : reduce_read_keys(ctx, sk, n, db, totals.as<u64>());
            // the sorted payload buffer changes owner: it is the DB's value array now
            void *sorted_vals = nullptr;
            if (in1) sorted_vals = v1.detach();
            else if (n > 0) { sorted_vals = vals; vals = nullptr; }
            if (compact) { db->vals_packed32 = (u32 *)sorted_vals; db->pos_bits = pos_bits; }
            else db->vals_packed = (u64 *)sorted_vals;
        } else r = reduce_records<u32, true>(ctx, sk, (const u32 *)sv, n, db, totals.as<u64>());
    }
    u64 tot[2] = {0, 0};
    if (r == KMG_OK) {
        hipError_t e = read_back(ctx, tot, totals.p, 16);
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); r = KMG_EDEVICE; }
    }
    if (r != KMG_OK) return r;
    db->U = tot[0];
    db->V = tot[1];
    if (heads_done) *heads_done = heads.done;
    return KMG_OK;
}

// X and Y ascending, every k-mer of X shorter than every k-mer of Y (extract_short.h: interleaving by rank) -> out
static int interleave_dbs(kmg_ctx *ctx, kmg_db *X, kmg_db *Y, kmg_db *out) {
    KMG_TRY(ensure_vals(ctx, X));
    KMG_TRY(ensure_vals(ctx, Y));
    const u64 U = X->U + Y->U, V = X->V + Y->V;
    void *p;
    if (ctx->pool.alloc(&p, (U + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    out->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, (U + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    out->val_off = (u64 *)p;
    if (ctx->pool.alloc(&p, (V + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    out->val_a = (int32_t *)p;
    if (ctx->pool.alloc(&p, (V + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    out->val_b = (int32_t *)p;
    if (ctx->pool.alloc(&p, U + 1) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    out->lens = (uint8_t *)p;
    PoolBuf cnt(ctx), px(ctx), py(ctx), agg(ctx);
    KMG_ALLOC(cnt, (U + 1) * 4);
    KMG_ALLOC(px, (X->U + 1) * 8);
    KMG_ALLOC(py, (Y->U + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    // (tile-wise: the piece of the other side a tile can land in is found first and searched in LDS)
    PoolBuf blo(ctx), bhi(ctx);
    const u64 xt = div_up_u64(X->U, ILP_TILE), yt = div_up_u64(Y->U, ILP_TILE);
    KMG_ALLOC(blo, (std::max(xt, yt) + 1) * 8);
    KMG_ALLOC(bhi, (std::max(xt, yt) + 1) * 8);
    if (X->U) {
        hipLaunchKernelGGL(il_bounds_kernel, dim3(div_up_u64(xt, 256)), dim3(256), 0, ctx->stream, X->keys, X->U, Y->keys, Y->U,
                           blo.as<u64>(), bhi.as<u64>());
        hipLaunchKernelGGL((il_place_tiled_kernel<true>), dim3(xt), dim3(256), 0, ctx->stream, X->keys, X->lens, X->k, X->val_off, X->U,
                           Y->keys, blo.as<u64>(), bhi.as<u64>(), out->keys, out->lens, cnt.as<u32>(), px.as<u64>());
    }
    if (Y->U) {
        hipLaunchKernelGGL(il_bounds_kernel, dim3(div_up_u64(yt, 256)), dim3(256), 0, ctx->stream, Y->keys, Y->U, X->keys, X->U,
                           blo.as<u64>(), bhi.as<u64>());
        hipLaunchKernelGGL((il_place_tiled_kernel<false>), dim3(yt), dim3(256), 0, ctx->stream, Y->keys, Y->lens, Y->k, Y->val_off, Y->U,
                           X->keys, blo.as<u64>(), bhi.as<u64>(), out->keys, out->lens, cnt.as<u32>(), py.as<u64>());
    }
    if (U)
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt.as<u32>(), out->val_off, U, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
    if (X->U)
        hipLaunchKernelGGL(il_values_kernel, dim3(div_up_u64(X->U, ILV_BLOCK)), dim3(ILV_BLOCK), 0, ctx->stream, X->val_off, X->val_a, X->val_b,
                           X->U, px.as<u64>(), out->val_off, out->val_a, out->val_b);
    if (Y->U)
        hipLaunchKernelGGL(il_values_kernel, dim3(div_up_u64(Y->U, ILV_BLOCK)), dim3(ILV_BLOCK), 0, ctx->stream, Y->val_off, Y->val_a, Y->val_b,
                           Y->U, py.as<u64>(), out->val_off, out->val_a, out->val_b);
    KMG_HIP(hipGetLastError());
    out->U = U;
    out->V = V;
    return KMG_OK;
}

// bucket_off (and val_off[U]) of a database whose U and V the host knows
static int db_make_buckets(kmg_ctx *ctx, kmg_db *db) {
    const u32 nbuckets = 1u << (2 * db->key_len);
    if (!db->bucket_off) {
        void *p;
        if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->bucket_off = (u64 *)p;
    }
    PoolBuf totals(ctx);
    KMG_ALLOC(totals, 64);
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, totals.as<u64>() + 0, (u64)db->U);
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, totals.as<u64>() + 1, (u64)db->V);
    hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                       totals.as<u64>(), db->val_off, db->bucket_off, nbuckets, db->key_len);
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// The k-mers shorter than k of a variable-length build: every length is a fixed-length build of its own (gathered in
// emission order, so read positions stay ascending), then the databases are interleaved by rank, shortest first, the
// full-length database last.  `full` is replaced by the interleaved database.
static int finish_variable(kmg_ctx *ctx, std::unique_ptr<kmg_db, void (*)(kmg_db *)> &full,
                           const std::vector<SortPass> &payload) {
    const u64 ns = ctx->sh_n;
    const size_t vb = val_bytes(ctx->kind);
    u32 hist[64];
    memset(hist, 0, sizeof hist);
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> acc(nullptr, db_release);
    if (ns) {
        if (ns >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 records in one build");
        PoolBuf d_hist(ctx), flag(ctx), dst(ctx), agg(ctx);
        KMG_ALLOC(d_hist, 64 * 4);
        KMG_HIP(hipMemsetAsync(d_hist.p, 0, 64 * 4, ctx->stream));
        hipLaunchKernelGGL(short_hist_kernel, dim3(std::min<u32>(div_up_u64(ns, 1024), 2048)), dim3(256), 0, ctx->stream, ctx->sh_len, ns,
                           d_hist.as<u32>());
        KMG_HIP(read_back(ctx, hist, d_hist.p, sizeof hist));
        KMG_ALLOC(flag, ns * 4);
        KMG_ALLOC(dst, (ns + 1) * 4);
        KMG_ALLOC(agg, (size_t)(div_up_u64(ns, SCAN_TILE) + 1) * 8);
        for (int L0 = ctx->min_k; L0 < ctx->k; L0 += 4) {
            const int L1 = std::min(L0 + 4, ctx->k);             // the class: lengths L0 .. L1 - 1 (all below 32)
            u64 n64 = 0;
            for (int L = L0; L < L1; L++) n64 += hist[L];
            if (!n64) continue;
            const u32 n = (u32)n64;
            PoolBuf ck(ctx), cv(ctx);
            KMG_ALLOC(ck, (size_t)n * 8);
            KMG_ALLOC(cv, (size_t)n * vb);
            const dim3 grid(div_up_u64(ns, 256)), blk(256);
            hipLaunchKernelGGL(short_flag_kernel, grid, blk, 0, ctx->stream, ctx->sh_len, ns, L0, L1, flag.as<u32>());
            KMG_TRY((device_scan<u32, u32, OpSum, true>(ctx, flag.as<u32>(), dst.as<u32>(), ns, OpSum(), 0u, agg.as<u32>(), nullptr,
                                                        false, KP_SCAN)));
            if (vb == 4)
                hipLaunchKernelGGL((short_gather_kernel<u32>), grid, blk, 0, ctx->stream, ctx->sh_len, dst.as<u32>(), ns, L0, L1, ctx->sh_keys,
                                   (const u32 *)ctx->sh_vals, ck.as<u64>(), cv.as<u32>());
            else
                hipLaunchKernelGGL((short_gather_kernel<u64>), grid, blk, 0, ctx->stream, ctx->sh_len, dst.as<u32>(), ns, L0, L1, ctx->sh_keys,
                                   (const u64 *)ctx->sh_vals, ck.as<u64>(), cv.as<u64>());
            KMG_HIP(hipGetLastError());
            kmg_db *c = new (std::nothrow) kmg_db();
            if (!c) return fail(ctx, KMG_ENOMEM, "host allocation failed");
            std::unique_ptr<kmg_db, void (*)(kmg_db *)> cls(c, db_release);
            // (sorted as 32-mers: the length tag in the two lowest bits takes part in the order and in the runs)
            c->ctx = ctx; c->k = 32; c->min_k = 32; c->key_len = ctx->key_len; c->rc = ctx->rc; c->kind = ctx->kind;
            void *vals = cv.detach();
            int r = sort_reduce_class(ctx, ck.as<u64>(), vals, n, false, 16, payload, build_seg_plan(ctx, build_pbits(ctx, n, 64)), false, c, nullptr);
            ctx->pool.release(vals);
            KMG_TRY(r);
            void *lp;
            if (ctx->pool.alloc(&lp, c->U + 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
            c->lens = (uint8_t *)lp;
            if (c->U)
                hipLaunchKernelGGL(short_split_kernel, dim3(div_up_u64(c->U, 256)), blk, 0, ctx->stream, c->keys, c->U, L0, c->lens);
            c->k = L1 - 1; c->min_k = L0;
            if (!acc) acc = std::move(cls);
            else {
                kmg_db *m = new (std::nothrow) kmg_db();
                if (!m) return fail(ctx, KMG_ENOMEM, "host allocation failed");
                std::unique_ptr<kmg_db, void (*)(kmg_db *)> merged(m, db_release);
                m->ctx = ctx; m->k = L1 - 1; m->min_k = acc->min_k; m->key_len = ctx->key_len; m->rc = ctx->rc; m->kind = ctx->kind;
                KMG_TRY(interleave_dbs(ctx, acc.get(), c, m));
                acc = std::move(merged);
            }
        }
    }
    kmg_db *f = full.get();
    if (acc) {
        kmg_db *m = new (std::nothrow) kmg_db();
        if (!m) return fail(ctx, KMG_ENOMEM, "host allocation failed");
        std::unique_ptr<kmg_db, void (*)(kmg_db *)> merged(m, db_release);
        m->ctx = ctx; m->k = ctx->k; m->key_len = ctx->key_len; m->rc = ctx->rc; m->kind = ctx->kind;
        KMG_TRY(interleave_dbs(ctx, acc.get(), f, m));
        KMG_TRY(db_make_buckets(ctx, m));
        full = std::move(merged);
        f = full.get();
    } else {
        // no shorter k-mer came up: the full-length database with a length beside every k-mer
        void *p;
        if (ctx->pool.alloc(&p, f->U + 1) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        f->lens = (uint8_t *)p;
        KMG_HIP(hipMemsetAsync(f->lens, ctx->k, f->U + 1, ctx->stream));
    }
    f->min_k = ctx->min_k;
    KMG_HIP(hipStreamSynchronize(ctx->stream));     // the class databases go back to the pool behind this
    return KMG_OK;
}

// The records of a build of two-word k-mers (ctx->rec_keys / rec_lo / rec_vals, window order) -> db's CSR.  wide.h, steps 2-5.
// n two-word k-mers (hi[i], lo[i]) in any order -> their order as k-mers (wide.h, steps 2-3): `order[j]` = index of the
// j-th smallest, equal ones in index order (both sorts are stable); his / los = the words in that order; flag[j] = j
// starts a new k-mer, before[j] = heads before j; U = distinct k-mers, rbits = bits their ranks take.
struct WideOrder {
    PoolBuf ka, kb, kc, i0, i1, flag, before, lens;      // lens: the lengths in k-mer order (k-mers of several lengths only)
    const u64 *his = nullptr;
    u64 *los = nullptr;
    const u32 *order = nullptr;
    u64 U = 0;
    int rbits = 1;
    explicit WideOrder(kmg_ctx *c) : ka(c), kb(c), kc(c), i0(c), i1(c), flag(c), before(c), lens(c) {}
};
static int wide_order(kmg_ctx *ctx, const u64 *hi, const u64 *lo, u32 n, int k, WideOrder &w,
                      const uint8_t *len = nullptr /* k-mers of several lengths: (hi, lo, len) is their order */) {
    PoolBuf table(ctx), agg(ctx), seg(ctx);
    const u32 ntiles = div_up_u64(n, RS_TILE);
    const int lo_bits = 2 * (k - 32);
    const int p_lo = choose_pbits(n, lo_bits), p_hi = choose_pbits(n, 64);
    KMG_ALLOC(w.ka, (size_t)n * 8);
    KMG_ALLOC(w.kb, (size_t)n * 8);
    KMG_ALLOC(w.i0, (size_t)n * 4);
    KMG_ALLOC(w.i1, (size_t)n * 4);
    KMG_ALLOC(table, (size_t)RS_DIGITS * ntiles * 4);
    KMG_ALLOC(agg, (size_t)(div_up_u64(std::max<u64>((u64)RS_DIGITS * ntiles, n), SCAN_TILE) + 1) * 8);
    KMG_ALLOC(seg, (((size_t)1 << std::max(p_lo, p_hi)) + 4) * 4);
    const dim3 grid(div_up_u64(n, 256)), blk(256);
    const std::vector<SortPass> none;
    // (by length first, where lengths differ,) by lo (any order among equal words), then -- stable -- by hi: every launch
    // below is a kernel of radix.h
    hipLaunchKernelGGL(wide_iota_kernel, grid, blk, 0, ctx->stream, w.i0.as<u32>(), n);
    bool in1 = false;
    u32 *perm = w.i0.as<u32>(), *perm_other = w.i1.as<u32>();
    bool ordered = false;
    {
        // by hi alone, then every run of equal first words by lo in place (wide_runs_kernel) -- unless a run
        // is long (low-complexity input), which the two sorts below take
        PoolBuf longrun(ctx);
        KMG_ALLOC(longrun, 64);
        KMG_HIP(hipMemsetAsync(longrun.p, 0, 4, ctx->stream));
        KMG_HIP(hipMemcpyAsync(w.ka.p, hi, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        KMG_TRY(hybrid_sort_records<u32>(ctx, w.ka.as<u64>(), perm, w.kb.as<u64>(), perm_other, n, 0, none, table.as<u32>(), agg.as<u32>(),
                                         seg.as<u32>(), whole_space_plan(p_hi), false, &in1, nullptr, false));
        const u64 *his = in1 ? w.kb.as<u64>() : w.ka.as<u64>();
        hipLaunchKernelGGL(wide_longrun_kernel, grid, blk, 0, ctx->stream, his, n, longrun.as<u32>());
        u32 h_long = 0;
        KMG_HIP(read_back(ctx, &h_long, longrun.p, 4));
        if (!h_long) {
            w.his = his;
            u64 *gathered = in1 ? w.ka.as<u64>() : w.kb.as<u64>();
            const u32 *order = in1 ? perm_other : perm;
            u32 *order_out = in1 ? perm : perm_other;
            KMG_ALLOC(w.kc, (size_t)n * 8);
            hipLaunchKernelGGL(wide_gather_kernel, grid, blk, 0, ctx->stream, lo, order, n, gathered);
            if (len) {
                // several lengths: (hi, lo, length) is the order -- the lengths travel with the second words
                PoolBuf lens_g(ctx);
                KMG_ALLOC(lens_g, (size_t)n + 64);
                KMG_ALLOC(w.lens, (size_t)n + 64);
                hipLaunchKernelGGL(wide_gather8_kernel, grid, blk, 0, ctx->stream, len, order, n, lens_g.as<uint8_t>());
                hipLaunchKernelGGL((wide_runs_kernel<true>), dim3(div_up_u64(n, WR_TILE)), dim3(WR_BLOCK), 0, ctx->stream, w.his,
                                   (const u64 *)gathered, (const uint8_t *)lens_g.as<uint8_t>(), order, n, w.kc.as<u64>(),
                                   w.lens.as<uint8_t>(), order_out);
                KMG_HIP(hipGetLastError());          // (lens_g goes back to the pool in stream order)
            } else
                hipLaunchKernelGGL((wide_runs_kernel<false>), dim3(div_up_u64(n, WR_TILE)), dim3(WR_BLOCK), 0, ctx->stream, w.his,
                                   (const u64 *)gathered, (const uint8_t *)nullptr, order, n, w.kc.as<u64>(), (uint8_t *)nullptr, order_out);
            w.los = w.kc.as<u64>();
            w.order = order_out;
            ordered = true;
        } else {
            hipLaunchKernelGGL(wide_iota_kernel, grid, blk, 0, ctx->stream, w.i0.as<u32>(), n);
            in1 = false;
        }
    }
    if (!ordered) {
    if (len) {
        hipLaunchKernelGGL(wide_lenkey_kernel, grid, blk, 0, ctx->stream, len, n, w.ka.as<u64>());
        KMG_TRY(hybrid_sort_records<u32>(ctx, w.ka.as<u64>(), perm, w.kb.as<u64>(), perm_other, n, 56, none, table.as<u32>(), agg.as<u32>(),
                                         seg.as<u32>(), whole_space_plan(choose_pbits(n, 8)), false, &in1, nullptr, false));
        if (in1) std::swap(perm, perm_other);
        hipLaunchKernelGGL(wide_gather_kernel, grid, blk, 0, ctx->stream, lo, (const u32 *)perm, n, w.ka.as<u64>());
    } else KMG_HIP(hipMemcpyAsync(w.ka.p, lo, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    KMG_TRY(hybrid_sort_records<u32>(ctx, w.ka.as<u64>(), perm, w.kb.as<u64>(), perm_other, n, 64 - lo_bits, none,
                                     table.as<u32>(), agg.as<u32>(), seg.as<u32>(), whole_space_plan(p_lo), false, &in1, nullptr, false));
    if (in1) std::swap(perm, perm_other);
    hipLaunchKernelGGL(wide_gather_kernel, grid, blk, 0, ctx->stream, hi, (const u32 *)perm, n, w.ka.as<u64>());
    KMG_TRY(hybrid_sort_records<u32>(ctx, w.ka.as<u64>(), perm, w.kb.as<u64>(), perm_other, n, 0, none, table.as<u32>(), agg.as<u32>(),
                                     seg.as<u32>(), whole_space_plan(p_hi), false, &in1, nullptr, false));
    w.his = in1 ? w.kb.as<u64>() : w.ka.as<u64>();
    w.los = in1 ? w.ka.as<u64>() : w.kb.as<u64>();
    w.order = in1 ? perm_other : perm;
    hipLaunchKernelGGL(wide_gather_kernel, grid, blk, 0, ctx->stream, lo, w.order, n, w.los);
    }
    if (len && !ordered) {
        KMG_ALLOC(w.lens, (size_t)n + 64);
        hipLaunchKernelGGL(wide_gather8_kernel, grid, blk, 0, ctx->stream, len, w.order, n, w.lens.as<uint8_t>());
    }
    KMG_ALLOC(w.flag, (size_t)n * 4);
    KMG_ALLOC(w.before, ((size_t)n + 1) * 8);
    hipLaunchKernelGGL(wide_heads_kernel, grid, blk, 0, ctx->stream, w.his, (const u64 *)w.los, (const uint8_t *)w.lens.p, n, w.flag.as<u32>());
    KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, w.flag.as<u32>(), w.before.as<u64>(), n, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                KP_SCAN)));
    KMG_HIP(read_back(ctx, &w.U, w.before.as<u64>() + n, 8));
    while (w.rbits < 32 && ((u64)1 << w.rbits) < w.U) w.rbits++;
    return KMG_OK;
}

static int finish_wide(kmg_ctx *ctx, u32 n, const std::vector<SortPass> &payload, kmg_db *db) {
    PoolBuf rank(ctx), d_hi(ctx), d_lo(ctx), d_len(ctx);
    u64 U = 0;
    int rbits = 1;
    const bool var = ctx->min_k < ctx->k;        // k-mers of several lengths: their lengths travel with them
    db->min_k = ctx->min_k;
    if (n > 0) {
        WideOrder w(ctx);
        KMG_TRY(wide_order(ctx, ctx->rec_keys, ctx->rec_lo, n, ctx->k, w, var ? ctx->rec_len : nullptr));
        U = w.U; rbits = w.rbits;
        const dim3 grid(div_up_u64(n, 256)), blk(256);
        KMG_ALLOC(d_hi, (U + 1) * 8);
        KMG_ALLOC(d_lo, (U + 1) * 8);
        if (var) KMG_ALLOC(d_len, U + 64);
        if (db->kind == KMG_PAYLOAD_READPOS && payload.empty()) {
            // the records' (read, pos) ascend as they were written and both sorts were stable: the CSR from the heads
            PoolBuf pay(ctx), voff(ctx);
            KMG_ALLOC(pay, ((size_t)n + 1) * 8);
            KMG_ALLOC(voff, (U + 2) * 8);
            hipLaunchKernelGGL(wide_gather_kernel, grid, blk, 0, ctx->stream, (const u64 *)ctx->rec_vals, w.order, n, pay.as<u64>());
            hipLaunchKernelGGL(wide_csr_reads_kernel, grid, blk, 0, ctx->stream, w.his, (const u64 *)w.los, w.flag.as<u32>(),
                               w.before.as<u64>(), n, d_hi.as<u64>(), d_lo.as<u64>(), voff.as<u64>(), (const uint8_t *)w.lens.p,
                               d_len.as<uint8_t>());
            KMG_HIP(hipGetLastError());
            db->U = U; db->V = n;
            db->keys = (u64 *)d_hi.detach();
            db->keys_lo = (u64 *)d_lo.detach();
            db->lens = (uint8_t *)d_len.detach();
            db->val_off = (u64 *)voff.detach();
            db->vals_packed = (u64 *)pay.detach();
            return db_make_buckets(ctx, db);      // (and val_off[U] = V)
        }
        KMG_ALLOC(rank, (size_t)n * 8);
        hipLaunchKernelGGL(wide_rank_kernel, grid, blk, 0, ctx->stream, w.his, (const u64 *)w.los, w.order, w.flag.as<u32>(),
                           w.before.as<u64>(), n, rbits, rank.as<u64>(), d_hi.as<u64>(), d_lo.as<u64>(), (const uint8_t *)w.lens.p,
                           d_len.as<uint8_t>());
        KMG_HIP(hipGetLastError());
    } else {
        KMG_ALLOC(d_hi, 8);
        KMG_ALLOC(d_lo, 8);
        if (var) KMG_ALLOC(d_len, 64);
    }
    // (rank, payload) records: the ranks order and compare as the k-mers do
    KMG_TRY(sort_reduce_class(ctx, rank.as<u64>(), ctx->rec_vals, n, false, 16, payload, whole_space_plan(choose_pbits(n, rbits)), false, db,
                              nullptr, rbits));
    if (db->U != U) return fail(ctx, KMG_EDEVICE, "two-word build: the ranks' distinct count differs from the k-mers'");
    ctx->pool.release(db->keys);
    db->keys = (u64 *)d_hi.detach();
    db->keys_lo = (u64 *)d_lo.detach();
    db->lens = (uint8_t *)d_len.detach();
    return db_make_buckets(ctx, db);
}

extern "C" int kmg_build_finish(kmg_ctx *ctx, kmg_db **out, uint64_t *n_windows) {
    if (!ctx || !out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (!ctx->building) return fail(ctx, KMG_ESTATE, "kmg_build_finish before kmg_build_begin");
    KMG_HIP(hipSetDevice(ctx->device));
    ctx->building = false;

    u64 st[ST_WORDS];
    if (ctx->snap_valid && ctx->n_adds == 1) {
        // single fused batch: the state was final before the write pass and is already here (see build_add_locked)
        KMG_HIP(hipEventSynchronize(ctx->ev_snap));
        memcpy(st, ctx->h_snap, sizeof st);
    } else KMG_HIP(read_back(ctx, st, ctx->d_state, sizeof st));
    ctx->snap_valid = false;
    const u64 N = st[ST_NREC];
    if (n_windows) *n_windows = st[ST_NWIN];
    if (N >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 records in one build");
    const u32 n = (u32)N;
    const bool reads = ctx->kind == KMG_PAYLOAD_READPOS;
    const bool compact = reads && ctx->compact_pb > 0;          // single-batch read build with 32-bit payloads
    const int pos_bits = compact ? ctx->compact_pb : 16;
    const size_t vb = compact ? 4 : val_bytes(ctx->kind);
    if (ctx->prof_on) ctx->prof[KP_EXTRACT_WRITE].bytes += N * (8 + vb);

    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    db->ctx = ctx; db->k = ctx->k; db->min_k = ctx->k; db->key_len = ctx->key_len; db->rc = ctx->rc; db->kind = ctx->kind;

    // payload digits (least significant): taxa order inside a k-mer / shared read indices
    std::vector<SortPass> payload;
    if (!reads) {
        const u32 varying = (u32)st[ST_ID_OR] ^ (u32)st[ST_ID_AND];
        for (int b = 0; b < 4; b++)
            if ((varying >> (8 * b)) & 0xFFu) payload.push_back({8 * b, 8, RS_SRC_VAL32});
    } else if (st[ST_DUP_READS]) {
        // canonical (read, pos) order inside a k-mer when sequences shared a read index
        int bits = pos_bits;
        while (bits < pos_bits + 32 && (st[ST_READ_BASE] >> (bits - pos_bits)) != 0) bits++;
        for (int shift = 0; shift < bits; shift += RS_BITS) payload.push_back({shift, std::min(RS_BITS, bits - shift), RS_SRC_VAL64});
    }
    if (ctx->k > 32) {
        int r = finish_wide(ctx, n, payload, db);
        ctx->pool.release(ctx->rec_keys); ctx->pool.release(ctx->rec_vals); ctx->pool.release(ctx->rec_lo); ctx->pool.release(ctx->rec_len);
        ctx->rec_keys = nullptr; ctx->rec_vals = nullptr; ctx->rec_lo = nullptr; ctx->rec_len = nullptr; ctx->rec_cap = 0; ctx->rec_upper = 0;
        if (r != KMG_OK) return r;
        if (!reads && st[ST_ID_MIN] <= st[ST_ID_MAX]) {
            db->tax_bounds = true;
            db->tax_lo = (int32_t)((u32)st[ST_ID_MIN] ^ 0x80000000u);
            db->tax_hi = (int32_t)((u32)st[ST_ID_MAX] ^ 0x80000000u);
        }
        *out = db_guard.release();
        return KMG_OK;
    }
    // prefix bits for the global passes: segments of ~1-2 K records on uniform data
    // (a single-batch build keeps the bits its extractor already partitioned the records by)
    const bool prepart = ctx->prepart && ctx->n_adds == 1;
    // (a key-range build holds its records in a slice of the prefixes: the segments are planned for the density there,
    // only the slice's segments exist, and the bits its keys share take no pass -- build_seg_plan)
    const int pbits = prepart ? ctx->plan_pbits : build_pbits(ctx, n, 2 * ctx->k);

    bool heads_done = false;
    int r = sort_reduce_class(ctx, ctx->rec_keys, ctx->rec_vals, n, compact, pos_bits, payload, build_seg_plan(ctx, pbits), prepart, db, &heads_done);
    // the record buffers go back to the pool: the next build (or the DB arrays of this one) reuses them
    ctx->pool.release(ctx->rec_keys);
    ctx->pool.release(ctx->rec_vals);
    ctx->rec_keys = nullptr; ctx->rec_vals = nullptr; ctx->rec_cap = 0; ctx->rec_upper = 0; ctx->compact_pb = 0;
    if (r == KMG_OK && ctx->min_k < ctx->k) {
        // the shorter k-mers carry (read << 16 | pos) in 64 bits whatever form the full windows took
        std::vector<SortPass> payload_short = payload;
        if (reads && st[ST_DUP_READS] && pos_bits != 16) {
            payload_short.clear();
            int bits = 16;
            while (bits < 16 + 32 && (st[ST_READ_BASE] >> (bits - 16)) != 0) bits++;
            for (int shift = 0; shift < bits; shift += RS_BITS) payload_short.push_back({shift, std::min(RS_BITS, bits - shift), RS_SRC_VAL64});
        }
        r = finish_variable(ctx, db_guard, payload_short);
    }
    ctx->pool.release(ctx->sh_keys); ctx->pool.release(ctx->sh_vals); ctx->pool.release(ctx->sh_len);
    ctx->sh_keys = nullptr; ctx->sh_vals = nullptr; ctx->sh_len = nullptr; ctx->sh_n = 0; ctx->sh_cap = 0;
    if (r != KMG_OK) return r;
    db = db_guard.get();
    if (!reads && st[ST_ID_MIN] <= st[ST_ID_MAX]) {
        // every taxon of the database is some sequence's id: TaxaCounts need not look for the span
        db->tax_bounds = true;
        db->tax_lo = (int32_t)((u32)st[ST_ID_MIN] ^ 0x80000000u);
        db->tax_hi = (int32_t)((u32)st[ST_ID_MAX] ^ 0x80000000u);
    }
    if (ctx->prof_on)
    {
        ctx->prof[KP_REDUCE_WRITE].bytes += db->U * (heads_done ? (8 + vb) + 16 : 16) + (reads ? 0 : db->V * 8);
        if (heads_done) ctx->prof[KP_SORT_LOCAL_V64].bytes += db->U * (8 + vb);
    }
    *out = db_guard.release();
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ db
extern "C" int kmg_db_info(const kmg_db *db, int *k, int *key_len, int *rc, int *payload_kind, uint64_t *n_unique,
                           uint64_t *n_values) {
    if (!db) return KMG_EINVAL;
    if (k) *k = db->k;
    if (key_len) *key_len = db->key_len;
    if (rc) *rc = db->rc;
    if (payload_kind) *payload_kind = db->kind;
    if (n_unique) *n_unique = db->U;
    if (n_values) *n_values = db->V;
    return KMG_OK;
}

extern "C" int kmg_db_min_k(const kmg_db *db) { return db ? db->min_k : 0; }

extern "C" int kmg_db_fetch_lens(const kmg_db *db, uint8_t *lens) {
    if (!db || !lens) return KMG_EINVAL;
    kmg_ctx *ctx = db->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (!db->lens) { memset(lens, db->k, db->U); return KMG_OK; }
    if (db->U) KMG_HIP(hipMemcpyAsync(lens, db->lens, db->U, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

extern "C" int kmg_db_payload_bits(const kmg_db *db) {
    if (!db) return 0;
    return db->vals_packed32 ? 32 : (db->vals_packed ? 64 : 0);
}

extern "C" int kmg_db_fetch(const kmg_db *db, uint64_t *keys, uint64_t *val_off, int32_t *val_a, int32_t *val_b,
                            uint64_t *bucket_off) {
    if (!db) return KMG_EINVAL;
    kmg_ctx *ctx = db->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (val_a || val_b) KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(db)));
    if (keys && db->U) KMG_HIP(hipMemcpyAsync(keys, db->keys, db->U * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (val_off) KMG_HIP(hipMemcpyAsync(val_off, db->val_off, (db->U + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (val_a && db->V) KMG_HIP(hipMemcpyAsync(val_a, db->val_a, db->V * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (val_b && db->V) KMG_HIP(hipMemcpyAsync(val_b, db->val_b, db->V * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bucket_off)
        KMG_HIP(hipMemcpyAsync(bucket_off, db->bucket_off, (((size_t)1 << (2 * db->key_len)) + 1) * 8,
                               hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

extern "C" int kmg_db_fetch_keys_lo(const kmg_db *db, uint64_t *keys_lo) {
    if (!db || !keys_lo) return KMG_EINVAL;
    kmg_ctx *ctx = db->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (!db->keys_lo) { memset(keys_lo, 0, db->U * 8); return KMG_OK; }      // k <= 32: no base behind the 32nd
    if (db->U) KMG_HIP(hipMemcpyAsync(keys_lo, db->keys_lo, db->U * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

// device check of a key array handed over by the caller: *bad += 1 wherever keys[i - 1] >= keys[i]
__global__ void keys_ascending_kernel(const u64 *__restrict__ keys, u64 n, u32 *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i < n && keys[i - 1] >= keys[i]) atomicAdd(bad, 1u);
}

// shared by kmg_db_load (host arrays) and kmg_db_load_dev (device arrays); caller holds ctx->mu and has
// validated the arguments it can see
static int db_load_common(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind, const u64 *keys, u64 n_unique,
                          const u64 *val_off, const int32_t *val_a, const int32_t *val_b, u64 V, hipMemcpyKind how,
                          kmg_db **out, const u64 *keys_lo = nullptr /* two-word k-mers: the second words, where keys are */) {
    KMG_HIP(hipSetDevice(ctx->device));
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    db->ctx = ctx; db->k = k; db->min_k = k; db->key_len = key_len; db->rc = rc ? 1 : 0; db->kind = payload_kind;
    db->U = n_unique; db->V = V;
    const u32 nbuckets = 1u << (2 * key_len);
    void *p;
    int r = KMG_OK;
    if (ctx->pool.alloc(&p, (n_unique + 1) * 8) == KMG_OK) db->keys = (u64 *)p; else r = KMG_ENOMEM;
    if (r == KMG_OK && ctx->pool.alloc(&p, (n_unique + 2) * 8) == KMG_OK) db->val_off = (u64 *)p; else r = KMG_ENOMEM;
    if (r == KMG_OK && ctx->pool.alloc(&p, (V + 1) * 4) == KMG_OK) db->val_a = (int32_t *)p; else r = KMG_ENOMEM;
    if (r == KMG_OK && ctx->pool.alloc(&p, (V + 1) * 4) == KMG_OK) db->val_b = (int32_t *)p; else r = KMG_ENOMEM;
    if (r == KMG_OK && ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) == KMG_OK) db->bucket_off = (u64 *)p; else r = KMG_ENOMEM;
    if (keys_lo) { if (r == KMG_OK && ctx->pool.alloc(&p, (n_unique + 1) * 8) == KMG_OK) db->keys_lo = (u64 *)p; else r = KMG_ENOMEM; }
    PoolBuf totals(ctx);
    if (r == KMG_OK && totals.alloc(64) != KMG_OK) r = KMG_ENOMEM;
    if (r != KMG_OK) { db_release(db); return fail(ctx, KMG_ENOMEM, "device allocation failed"); }
    hipError_t e = hipSuccess;
    if (n_unique) e = hipMemcpyAsync(db->keys, keys, n_unique * 8, how, ctx->stream);
    if (e == hipSuccess && n_unique && keys_lo) e = hipMemcpyAsync(db->keys_lo, keys_lo, n_unique * 8, how, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db->val_off, val_off, (n_unique + 1) * 8, how, ctx->stream);
    if (e == hipSuccess && V) e = hipMemcpyAsync(db->val_a, val_a, V * 4, how, ctx->stream);
    if (e == hipSuccess && V) e = hipMemcpyAsync(db->val_b, val_b, V * 4, how, ctx->stream);
    u64 tot[4] = {n_unique, V, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(totals.p, tot, 32, hipMemcpyHostToDevice, ctx->stream);
    u64 check[3] = {0, V, 0};   // val_off[0], val_off[n_unique], #descents in keys
    if (e == hipSuccess) {
        if (how == hipMemcpyDeviceToDevice) {
            if (n_unique > 1 && keys_lo)
                hipLaunchKernelGGL(wide_ascending_kernel, dim3(div_up_u64(n_unique, 256)), dim3(256), 0, ctx->stream, (const u64 *)db->keys,
                                   (const u64 *)db->keys_lo, n_unique, reinterpret_cast<u32 *>(totals.as<u64>() + 2));
            else if (n_unique > 1)
                hipLaunchKernelGGL(keys_ascending_kernel, dim3(div_up_u64(n_unique, 256)), dim3(256), 0, ctx->stream, db->keys,
                                   n_unique, reinterpret_cast<u32 *>(totals.as<u64>() + 2));
            e = hipMemcpyAsync(&check[0], db->val_off, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&check[1], db->val_off + n_unique, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&check[2], totals.as<u64>() + 2, 8, hipMemcpyDeviceToHost, ctx->stream);
        }
        hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                           totals.as<u64>(), db->val_off, db->bucket_off, nbuckets, key_len);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); db_release(db); return KMG_EDEVICE; }
    if (check[0] != 0 || check[1] != V) { db_release(db); return fail(ctx, KMG_EINVAL, "val_off must run from 0 to n_values"); }
    if (check[2] != 0) { db_release(db); return fail(ctx, KMG_EINVAL, "keys must be strictly ascending"); }
    *out = db;
    return KMG_OK;
}

extern "C" int kmg_db_load(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                           uint64_t n_unique, const uint64_t *val_off, const int32_t *val_a, const int32_t *val_b,
                           kmg_db **out) {
    if (!ctx || !out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (k < 1 || key_len < 1) return fail(ctx, KMG_EINVAL, "k and key_len must be positive");
    if (k > 32 || key_len > k || key_len > 12) return fail(ctx, KMG_EUNSUPPORTED, "k > 32 or key_len > min(k, 12)");
    if (!val_off || (n_unique && !keys)) return fail(ctx, KMG_EINVAL, "null keys / val_off");
    const u64 V = val_off[n_unique];
    if (V && (!val_a || !val_b)) return fail(ctx, KMG_EINVAL, "null values");
    for (u64 i = 1; i < n_unique; i++)
        if (keys[i - 1] >= keys[i]) return fail(ctx, KMG_EINVAL, "keys must be strictly ascending");
    return db_load_common(ctx, k, key_len, rc, payload_kind, reinterpret_cast<const u64 *>(keys), n_unique,
                          reinterpret_cast<const u64 *>(val_off), val_a, val_b, V, hipMemcpyHostToDevice, out);
}

extern "C" int kmg_db_load_wide(kmg_ctx *ctx, int k, int min_k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                                const uint64_t *keys_lo, const uint8_t *lens, uint64_t n_unique, const uint64_t *val_off,
                                const int32_t *val_a, const int32_t *val_b, kmg_db **out) {
    if (!ctx || !out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (k <= 32 || key_len < 1 || min_k < 1 || min_k > k)
        return fail(ctx, KMG_EINVAL, "k-mers of 33 .. 64 bases (kmg_db_load / kmg_db_load_var take the shorter ones); key_len and min_k positive, min_k <= k");
    if (k > 64 || key_len > 12 || key_len > min_k) return fail(ctx, KMG_EUNSUPPORTED, "k > 64 or key_len > min(min_k, 12)");
    if (!val_off || (n_unique && (!keys || !keys_lo || (min_k < k && !lens)))) return fail(ctx, KMG_EINVAL, "null keys / lens / val_off");
    const u64 V = val_off[n_unique];
    if (V && (!val_a || !val_b)) return fail(ctx, KMG_EINVAL, "null values");
    for (u64 i = 0; i < n_unique; i++) {
        const int L = min_k < k ? lens[i] : k;
        if (L < min_k || L > k) return fail(ctx, KMG_EINVAL, "k-mer length outside min_k .. k");
        const bool clean = L >= 64 ? true
                         : L > 32 ? (keys_lo[i] << (2 * (L - 32))) == 0
                         : L == 32 ? keys_lo[i] == 0 : (keys_lo[i] == 0 && (keys[i] << (2 * L)) == 0);
        if (!clean) return fail(ctx, KMG_EINVAL, "k-mer bits behind its length must be zero");
        // Sequence.compareTo order (Sequence.java:165-187): zero-padded value, then length
        if (i) {
            const int Lp = min_k < k ? lens[i - 1] : k;
            const bool less = keys[i - 1] < keys[i] || (keys[i - 1] == keys[i] && (keys_lo[i - 1] < keys_lo[i] || (keys_lo[i - 1] == keys_lo[i] && Lp < L)));
            if (!less) return fail(ctx, KMG_EINVAL, "k-mers must be strictly ascending");
        }
    }
    KMG_TRY(db_load_common(ctx, k, key_len, rc, payload_kind, reinterpret_cast<const u64 *>(keys), n_unique,
                           reinterpret_cast<const u64 *>(val_off), val_a, val_b, V, hipMemcpyHostToDevice, out));
    kmg_db *db = *out;
    db->min_k = min_k;
    void *p, *q = nullptr;
    if (ctx->pool.alloc(&p, (n_unique + 1) * 8) != KMG_OK || (min_k < k && ctx->pool.alloc(&q, n_unique + 64) != KMG_OK)) {
        db_release(db); *out = nullptr;
        return fail(ctx, KMG_ENOMEM, "device allocation failed");
    }
    db->keys_lo = (u64 *)p;
    db->lens = (uint8_t *)q;
    hipError_t e = n_unique ? hipMemcpyAsync(db->keys_lo, keys_lo, n_unique * 8, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
    if (e == hipSuccess && q && n_unique) e = hipMemcpyAsync(db->lens, lens, n_unique, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); db_release(db); *out = nullptr; return KMG_EDEVICE; }
    return KMG_OK;
}

extern "C" int kmg_db_load_dev(kmg_ctx *ctx, int k, int key_len, int rc, int payload_kind, const uint64_t *d_keys,
                               uint64_t n_unique, const uint64_t *d_val_off, const int32_t *d_val_a,
                               const int32_t *d_val_b, uint64_t n_values, kmg_db **out) {
    if (!ctx || !out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (k < 1 || key_len < 1) return fail(ctx, KMG_EINVAL, "k and key_len must be positive");
    if (k > 32 || key_len > k || key_len > 12) return fail(ctx, KMG_EUNSUPPORTED, "k > 32 or key_len > min(k, 12)");
    if (!d_val_off || (n_unique && !d_keys)) return fail(ctx, KMG_EINVAL, "null keys / val_off");
    if (n_values && (!d_val_a || !d_val_b)) return fail(ctx, KMG_EINVAL, "null values");
    return db_load_common(ctx, k, key_len, rc, payload_kind, reinterpret_cast<const u64 *>(d_keys), n_unique,
                          reinterpret_cast<const u64 *>(d_val_off), d_val_a, d_val_b, n_values,
                          hipMemcpyDeviceToDevice, out);
}

extern "C" int kmg_db_load_var(kmg_ctx *ctx, int k, int min_k, int key_len, int rc, int payload_kind, const uint64_t *keys,
                               const uint8_t *lens, uint64_t n_unique, const uint64_t *val_off, const int32_t *val_a,
                               const int32_t *val_b, kmg_db **out) {
    if (!ctx || !out) return KMG_EINVAL;
    std::unique_lock<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (k < 1 || key_len < 1 || min_k < 1 || min_k > k) return fail(ctx, KMG_EINVAL, "k, min_k and key_len must be positive, min_k <= k");
    if (k > 32 || key_len > min_k || key_len > 12) return fail(ctx, KMG_EUNSUPPORTED, "k > 32 or key_len > min(min_k, 12)");
    if (!val_off || (n_unique && (!keys || !lens))) return fail(ctx, KMG_EINVAL, "null keys / lens / val_off");
    const u64 V = val_off[n_unique];
    if (V && (!val_a || !val_b)) return fail(ctx, KMG_EINVAL, "null values");
    for (u64 i = 0; i < n_unique; i++) {
        const int L = lens[i];
        if (L < min_k || L > k) return fail(ctx, KMG_EINVAL, "k-mer length outside min_k .. k");
        if (L < 32 && (keys[i] << (2 * L)) != 0) return fail(ctx, KMG_EINVAL, "k-mer bits behind its length must be zero");
        // Sequence.compareTo order (Sequence.java:165-187): zero-padded value, then length
        if (i && (keys[i - 1] > keys[i] || (keys[i - 1] == keys[i] && lens[i - 1] >= lens[i])))
            return fail(ctx, KMG_EINVAL, "k-mers must be strictly ascending");
    }
    // (db_load_common checks strictly ascending keys on device arrays only: host arrays were checked above)
    KMG_TRY(db_load_common(ctx, k, key_len, rc, payload_kind, reinterpret_cast<const u64 *>(keys), n_unique,
                           reinterpret_cast<const u64 *>(val_off), val_a, val_b, V, hipMemcpyHostToDevice, out));
    kmg_db *db = *out;
    db->min_k = min_k;
    if (min_k < k) {
        void *p;
        if (ctx->pool.alloc(&p, n_unique + 1) != KMG_OK) { db_release(db); *out = nullptr; return fail(ctx, KMG_ENOMEM, "device allocation failed"); }
        db->lens = (uint8_t *)p;
        hipError_t e = n_unique ? hipMemcpyAsync(db->lens, lens, n_unique, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); db_release(db); *out = nullptr; return KMG_EDEVICE; }
    }
    return KMG_OK;
}

extern "C" int kmg_db_copy_dev(const kmg_db *db, int which, void *d_dst, uint64_t first, uint64_t count) {
    if (!db || (count && !d_dst)) return KMG_EINVAL;
    kmg_ctx *ctx = db->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    const void *src = nullptr;
    size_t elem = 8;
    u64 limit = 0;
    switch (which) {
        case KMG_DB_KEYS: src = db->keys; limit = db->U; break;
        case KMG_DB_KEYS_LO:
            if (!db->keys_lo) return fail(ctx, KMG_EINVAL, "k-mers of at most 32 bases have no second word");
            src = db->keys_lo; limit = db->U; break;
        case KMG_DB_VAL_OFF: src = db->val_off; limit = db->U + 1; break;
        case KMG_DB_VAL_A:
        case KMG_DB_VAL_B:
            KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(db)));
            src = which == KMG_DB_VAL_A ? db->val_a : db->val_b; elem = 4; limit = db->V; break;
        case KMG_DB_BUCKET_OFF: src = db->bucket_off; limit = ((u64)1 << (2 * db->key_len)) + 1; break;
        default: return fail(ctx, KMG_EINVAL, "unknown array id");
    }
    if (first > limit || count > limit - first) return fail(ctx, KMG_EINVAL, "slice outside the array");
    if (count)
        KMG_HIP(hipMemcpyAsync(d_dst, (const char *)src + first * elem, count * elem, hipMemcpyDeviceToDevice, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

extern "C" int kmg_db_encode(kmg_ctx *ctx, const kmg_db *db, uint8_t *out, uint64_t cap, uint64_t *n_bytes,
                             uint64_t *block_off) {
    if (!ctx || !db || !n_bytes) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (db->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    KMG_HIP(hipSetDevice(ctx->device));
    const u64 U = db->U;
    const u32 nbuckets = 1u << (2 * db->key_len);
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(db)));
    PoolBuf sizes(ctx), enc_off(ctx), agg(ctx), d_out(ctx), d_boff(ctx);
    KMG_ALLOC(sizes, (U + 1) * 4);
    KMG_ALLOC(enc_off, (U + 2) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    ProfScope ps(ctx, KP_ENCODE, 0);
    const bool wide = db->keys_lo != nullptr;
    if (U && wide) {
        hipLaunchKernelGGL((wide_enc_kernel<false>), dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, db->keys, db->keys_lo, db->lens, db->val_off,
                           db->val_a, db->val_b, U, db->k, db->key_len, db->kind, sizes.as<u32>(), (const u64 *)nullptr, (uint8_t *)nullptr);
    } else if (U) {
        hipLaunchKernelGGL(enc_size_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, db->keys, db->lens, db->val_off, U,
                           db->k, db->key_len, db->kind, sizes.as<u32>());
    }
    KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, sizes.as<u32>(), enc_off.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(),
                                                nullptr, true, KP_SCAN)));
    u64 total = 0;
    KMG_HIP(hipMemcpyAsync(&total, enc_off.as<u64>() + U, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    *n_bytes = total;
    if (block_off) {
        KMG_ALLOC(d_boff, ((size_t)nbuckets + 1) * 8);
        hipLaunchKernelGGL(enc_blockoff_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream,
                           db->bucket_off, enc_off.as<u64>(), nbuckets, d_boff.as<u64>());
        KMG_HIP(hipMemcpyAsync(block_off, d_boff.p, ((size_t)nbuckets + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (out) {
        if (cap < total) return fail(ctx, KMG_EINVAL, "encode buffer too small");
        if (total) {
            KMG_ALLOC(d_out, total);
            if (wide)
                hipLaunchKernelGGL((wide_enc_kernel<true>), dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, db->keys, db->keys_lo,
                                   db->lens, db->val_off, db->val_a, db->val_b, U, db->k, db->key_len, db->kind, (u32 *)nullptr, enc_off.as<u64>(),
                                   d_out.as<uint8_t>());
            else
            hipLaunchKernelGGL(enc_write_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, db->keys, db->lens,
                               db->val_off, db->val_a, db->val_b, U, db->k, db->key_len, db->kind, enc_off.as<u64>(),
                               d_out.as<uint8_t>());
            KMG_HIP(hipMemcpyAsync(out, d_out.p, total, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    KMG_HIP(hipGetLastError());
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

// Second half of kmg_db_merge and kmg_db_restrict: n expanded (k-mer, payload) records in (k0, v0) -> sorted, equal
// k-mers folded with the data type's reducer (counts add per taxon, read-position sets unite) into db's CSR arrays.
static int merge_records(kmg_ctx *ctx, kmg_db *db, PoolBuf &k0, PoolBuf &v0, PoolBuf &k1, PoolBuf &v1, u32 n, PoolBuf &totals,
                         int key_bits = 0 /* bits of the keys that count, from the top; 0: 2 * db->k */) {
    if (!key_bits) key_bits = 2 * db->k;
    const bool reads = db->kind == KMG_PAYLOAD_READPOS;
    const u32 nbuckets = 1u << (2 * db->key_len);
    PoolBuf table(ctx), agg(ctx), seg(ctx), fk(ctx), fp(ctx), pk(ctx), pp(ctx);
    bool in1 = false;
    if (n > 0) {
        // inside a k-mer: taxa ascending (upper payload half) | (read, pos) ascending (whole payload)
        std::vector<SortPass> payload;
        if (reads) for (int sh = 0; sh < 48; sh += RS_BITS) payload.push_back({sh, RS_BITS, RS_SRC_VAL64});
        else for (int sh = 32; sh < 64; sh += RS_BITS) payload.push_back({sh, RS_BITS, RS_SRC_VAL64});
        const int pbits = choose_pbits(n, key_bits);
        const u32 ntiles = div_up_u64(n, RS_TILE);
        KMG_ALLOC(table, (size_t)RS_DIGITS * ntiles * 4);
        KMG_ALLOC(agg, (size_t)(div_up_u64(std::max<u64>((u64)RS_DIGITS * ntiles, n), SCAN_TILE) + 1) * 8);
        KMG_ALLOC(seg, (((size_t)1 << pbits) + 4) * 4);
        KMG_TRY(hybrid_sort_records<u64>(ctx, k0.as<u64>(), v0.as<u64>(), k1.as<u64>(), v1.as<u64>(), n, 64 - key_bits, payload,
                                         table.as<u32>(), agg.as<u32>(), seg.as<u32>(), whole_space_plan(pbits), false, &in1));
    } else {
        KMG_ALLOC(agg, 64);
    }
    const u64 *sk = in1 ? k1.as<u64>() : k0.as<u64>();
    const u64 *sv = in1 ? v1.as<u64>() : v0.as<u64>();
    if (reads) {
        KMG_TRY(reduce_read_keys(ctx, sk, n, db, totals.as<u64>()));
        db->vals_packed = (u64 *)(in1 ? v1.detach() : v0.detach());
    } else {
        void *p;
        if (ctx->pool.alloc(&p, ((size_t)n + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->keys = (u64 *)p;
        if (ctx->pool.alloc(&p, ((size_t)n + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->val_off = (u64 *)p;
        if (ctx->pool.alloc(&p, ((size_t)n + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->val_a = (int32_t *)p;
        if (ctx->pool.alloc(&p, ((size_t)n + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->val_b = (int32_t *)p;
        if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        db->bucket_off = (u64 *)p;
        if (n > 0) {
            KMG_ALLOC(fk, (size_t)n * 4); KMG_ALLOC(fp, (size_t)n * 4);
            KMG_ALLOC(pk, ((size_t)n + 1) * 8); KMG_ALLOC(pp, ((size_t)n + 1) * 8);
            KMG_HIP(hipMemsetAsync(db->val_b, 0, ((size_t)n + 1) * 4, ctx->stream));
            const dim3 grid(div_up_u64(n, 256)), blk(256);
            hipLaunchKernelGGL(mg_flags_kernel, grid, blk, 0, ctx->stream, sk, sv, (u64)n, fk.as<u32>(), fp.as<u32>());
            KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, fk.as<u32>(), pk.as<u64>(), n, OpSum(), 0ull, agg.as<u64>(),
                                                        totals.as<u64>() + 0, false, KP_SCAN)));
            KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, fp.as<u32>(), pp.as<u64>(), n, OpSum(), 0ull, agg.as<u64>(),
                                                        totals.as<u64>() + 1, false, KP_SCAN)));
            hipLaunchKernelGGL(mg_write_kernel, grid, blk, 0, ctx->stream, sk, sv, (u64)n, fk.as<u32>(), fp.as<u32>(), pk.as<u64>(),
                               pp.as<u64>(), db->keys, db->val_off, db->val_a, reinterpret_cast<u32 *>(db->val_b));
            hipLaunchKernelGGL(mg_clamp_kernel, grid, blk, 0, ctx->stream, reinterpret_cast<u32 *>(db->val_b), totals.as<u64>());
        }
        hipLaunchKernelGGL(rd_buckets_kernel, dim3(div_up_u64(nbuckets + 1, 256)), dim3(256), 0, ctx->stream, db->keys,
                           totals.as<u64>(), db->val_off, db->bucket_off, nbuckets, db->key_len);
    }
    u64 tot[2] = {0, 0};
    KMG_HIP(hipMemcpyAsync(tot, totals.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    KMG_HIP(hipGetLastError());
    db->U = tot[0];
    db->V = tot[1];
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ merge
// Variable-length databases whose files SHARE keys: DB.kmers' merge (DB.java:99-109: mergeSortedStreams +
// groupAndReduceStream(addAll)) in Sequence.compareTo order = (zero-padded value, length).  Length by length: the
// entries of that length of every file as (k-mer, payload) records, folded by the merge pipeline into a database of one
// length; the lengths' databases interleaved by rank, shortest first, as a build's length classes are
// (extract_short.h).  Not a hot path: a handful of launches per length and file.
static int merge_var_shared(kmg_ctx *ctx, const kmg_db *const *dbs, int n_dbs, kmg_db **out) {
    const kmg_db *d0 = dbs[0];
    int rc = 1;
    for (int i = 0; i < n_dbs; i++) {
        rc &= dbs[i]->rc;
        KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(dbs[i])));
    }
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> acc(nullptr, db_release);
    for (int L = d0->min_k; L <= d0->k; L++) {
        std::vector<std::unique_ptr<PoolBuf>> dsts;
        std::vector<u64> cnt(n_dbs, 0);
        u64 total = 0;
        for (int i = 0; i < n_dbs; i++) {
            const kmg_db *d = dbs[i];
            dsts.emplace_back(new PoolBuf(ctx));
            if (d->U == 0) continue;
            PoolBuf keep(ctx), agg(ctx);
            KMG_ALLOC(keep, (d->U + 1) * 4);
            KMG_ALLOC(agg, (size_t)(div_up_u64(d->U, SCAN_TILE) + 1) * 8);
            KMG_ALLOC(*dsts[i], (d->U + 2) * 8);
            hipLaunchKernelGGL(rs_keep_kernel, dim3(div_up_u64(d->U, 256)), dim3(256), 0, ctx->stream, d->lens, d->k, d->val_off, d->U, L, L + 1,
                               keep.as<u32>());
            KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, keep.as<u32>(), dsts[i]->as<u64>(), d->U, OpSum(), 0ull, agg.as<u64>(), nullptr,
                                                        true, KP_SCAN)));
            KMG_HIP(read_back(ctx, &cnt[i], dsts[i]->as<u64>() + d->U, 8));
            total += cnt[i];
        }
        if (total == 0) continue;
        if (total >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 values of one k-mer length in the merged database");
        const u32 n = (u32)total;
        PoolBuf k0(ctx), v0(ctx), k1(ctx), v1(ctx), totals(ctx);
        KMG_ALLOC(totals, 64);
        KMG_HIP(hipMemsetAsync(totals.p, 0, 64, ctx->stream));
        KMG_ALLOC(k0, ((size_t)n + 1) * 8); KMG_ALLOC(v0, ((size_t)n + 1) * 8);
        KMG_ALLOC(k1, ((size_t)n + 1) * 8); KMG_ALLOC(v1, ((size_t)n + 1) * 8);
        u64 base = 0;
        for (int i = 0; i < n_dbs; i++) {
            const kmg_db *d = dbs[i];
            if (!cnt[i]) continue;
            hipLaunchKernelGGL(rs_expand_kernel, dim3(div_up_u64(d->U, 256)), dim3(256), 0, ctx->stream, d->keys, d->lens, d->k, d->val_off,
                               d->val_a, d->val_b, d->U, L, L + 1, dsts[i]->as<u64>(), base, d->kind, k0.as<u64>(), v0.as<u64>());
            base += cnt[i];
        }
        KMG_HIP(hipGetLastError());
        kmg_db *c = new (std::nothrow) kmg_db();
        if (!c) return fail(ctx, KMG_ENOMEM, "host allocation failed");
        std::unique_ptr<kmg_db, void (*)(kmg_db *)> cls(c, db_release);
        c->ctx = ctx; c->k = L; c->min_k = L; c->key_len = d0->key_len; c->rc = rc; c->kind = d0->kind;
        KMG_TRY(merge_records(ctx, c, k0, v0, k1, v1, n, totals));                 // (ends with a stream synchronisation)
        if (!acc) acc = std::move(cls);
        else {
            kmg_db *m = new (std::nothrow) kmg_db();
            if (!m) return fail(ctx, KMG_ENOMEM, "host allocation failed");
            std::unique_ptr<kmg_db, void (*)(kmg_db *)> merged(m, db_release);
            m->ctx = ctx; m->k = L; m->min_k = acc->min_k; m->key_len = d0->key_len; m->rc = rc; m->kind = d0->kind;
            KMG_TRY(interleave_dbs(ctx, acc.get(), c, m));
            KMG_HIP(hipStreamSynchronize(ctx->stream));         // acc and the class go back to the pool behind this
            acc = std::move(merged);
        }
    }
    if (!acc) return fail(ctx, KMG_ESTATE, "merge of databases that share keys found no k-mer");
    kmg_db *f = acc.get();
    if (!f->lens) {                                             // k-mers of one length only
        void *p;
        if (ctx->pool.alloc(&p, f->U + 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
        f->lens = (uint8_t *)p;
        KMG_HIP(hipMemsetAsync(f->lens, f->k, f->U + 1, ctx->stream));
    }
    f->k = d0->k; f->min_k = d0->min_k; f->rc = rc;
    f->shared_key_merge = true;
    if (d0->kind == KMG_PAYLOAD_TAXON)
        for (int i = 0; i < n_dbs; i++) {
            // the files as they are, in their order: what kmg_match_trie searches (a set merged from sets keeps the files)
            if (dbs[i]->parts.empty()) {
                kmg_db *part = nullptr;
                KMG_TRY(clone_for_trie(ctx, dbs[i], &part));
                f->parts.push_back(part);
            } else
                for (const kmg_db *q : dbs[i]->parts) {
                    kmg_db *part = nullptr;
                    KMG_TRY(clone_for_trie(ctx, q, &part));
                    f->parts.push_back(part);
                }
        }
    KMG_TRY(ensure_vals(ctx, f));
    KMG_TRY(db_make_buckets(ctx, f));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    KMG_HIP(hipGetLastError());
    *out = acc.release();
    return KMG_OK;
}

// Variable-length databases: a set whose files share no key (the sets MakeDatabase -S / -G write are cut at key
// bounds) is laid end to end in key order; files that share a key take merge_var_shared.
static int merge_disjoint_keys(kmg_ctx *ctx, const kmg_db *const *dbs, int n_dbs, kmg_db **out) {
    const kmg_db *d0 = dbs[0];
    const u32 nbuckets = 1u << (2 * d0->key_len);
    u64 Ut = 0, Vt = 0;
    int rc = 1;
    for (int i = 0; i < n_dbs; i++) { Ut += dbs[i]->U; Vt += dbs[i]->V; rc &= dbs[i]->rc; }
    KMG_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n_dbs; i++) KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(dbs[i])));
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    db->ctx = ctx; db->k = d0->k; db->min_k = d0->min_k; db->key_len = d0->key_len; db->rc = rc; db->kind = d0->kind;
    db->U = Ut; db->V = Vt;
    void *p;
    if (ctx->pool.alloc(&p, (Ut + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, Ut + 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->lens = (uint8_t *)p;
    if (ctx->pool.alloc(&p, (Ut + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_off = (u64 *)p;
    if (ctx->pool.alloc(&p, (Vt + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_a = (int32_t *)p;
    if (ctx->pool.alloc(&p, (Vt + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->val_b = (int32_t *)p;
    if (ctx->pool.alloc(&p, ((size_t)nbuckets + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    db->bucket_off = (u64 *)p;
    PoolBuf sum_v(ctx), holders(ctx), clash(ctx);
    KMG_ALLOC(sum_v, ((size_t)nbuckets + 1) * 8);
    KMG_ALLOC(holders, ((size_t)nbuckets + 1) * 4);
    KMG_ALLOC(clash, 64);
    KMG_HIP(hipMemsetAsync(db->bucket_off, 0, ((size_t)nbuckets + 1) * 8, ctx->stream));
    KMG_HIP(hipMemsetAsync(sum_v.p, 0, ((size_t)nbuckets + 1) * 8, ctx->stream));
    KMG_HIP(hipMemsetAsync(holders.p, 0, ((size_t)nbuckets + 1) * 4, ctx->stream));
    KMG_HIP(hipMemsetAsync(clash.p, 0, 64, ctx->stream));
    const dim3 bgrid(div_up_u64((u64)nbuckets + 1, 256)), blk(256);
    for (int i = 0; i < n_dbs; i++)
        hipLaunchKernelGGL(vm_sum_kernel, bgrid, blk, 0, ctx->stream, dbs[i]->bucket_off, dbs[i]->val_off, nbuckets, db->bucket_off,
                           sum_v.as<u64>(), holders.as<u32>());
    hipLaunchKernelGGL(vm_clash_kernel, bgrid, blk, 0, ctx->stream, holders.as<u32>(), nbuckets, clash.as<u32>());
    u32 h_clash = 0;
    KMG_HIP(read_back(ctx, &h_clash, clash.p, 4));
    if (h_clash) {
        db_guard.reset();
        return merge_var_shared(ctx, dbs, n_dbs, out);
    }
    for (int i = 0; i < n_dbs; i++) {
        const kmg_db *d = dbs[i];
        if (d->U == 0) continue;
        hipLaunchKernelGGL(vm_place_kernel, dim3(div_up_u64(d->U, 256)), blk, 0, ctx->stream, d->keys, d->lens, d->k, d->val_off, d->val_a,
                           d->val_b, d->bucket_off, d->U, d->key_len, db->bucket_off, sum_v.as<u64>(), db->keys, db->lens, db->val_off,
                           db->val_a, db->val_b);
    }
    KMG_HIP(hipMemcpyAsync(db->val_off + Ut, &Vt, 8, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipGetLastError());
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    *out = db_guard.release();
    return KMG_OK;
}

extern "C" int kmg_db_merge(kmg_ctx *ctx, const kmg_db *const *dbs, int n_dbs, kmg_db **out) {
    if (!ctx || !dbs || !out || n_dbs < 1) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    const kmg_db *d0 = dbs[0];
    u64 Vt = 0;
    int rc = 1;
    for (int i = 0; i < n_dbs; i++) {
        const kmg_db *d = dbs[i];
        if (!d || d->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
        if (d->kind != d0->kind) return fail(ctx, KMG_EINVAL, "Files contain different datatypes");          // DB.java:36-42
        if (d->k != d0->k || d->min_k != d0->min_k || d->key_len != d0->key_len)                           // DB.java:57-64
            return fail(ctx, KMG_EINVAL, "Files contains different kmer parameters (min/max length, key length or reverse complement included");
        Vt += d->V;
        rc &= d->rc;
    }
    if (d0->min_k != d0->k && d0->k <= 32) return merge_disjoint_keys(ctx, dbs, n_dbs, out);
    if (Vt >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 values in the merged database");
    KMG_HIP(hipSetDevice(ctx->device));
    const bool reads = d0->kind == KMG_PAYLOAD_READPOS;
    const u32 n = (u32)Vt;

    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    db->ctx = ctx; db->k = d0->k; db->min_k = d0->k; db->key_len = d0->key_len; db->rc = rc; db->kind = d0->kind;

    PoolBuf k0(ctx), v0(ctx), k1(ctx), v1(ctx), totals(ctx);
    KMG_ALLOC(totals, 64);
    KMG_HIP(hipMemsetAsync(totals.p, 0, 64, ctx->stream));
    KMG_ALLOC(k0, ((size_t)n + 1) * 8); KMG_ALLOC(v0, ((size_t)n + 1) * 8);
    KMG_ALLOC(k1, ((size_t)n + 1) * 8); KMG_ALLOC(v1, ((size_t)n + 1) * 8);
    u64 base = 0;
    if (d0->k > 32) {
        // two-word k-mers (wide.h): the files' k-mers, end to end, are ranked among the distinct ones; the ranks stand in
        // for the k-mers through the one-word merge, and the merged database gets the distinct k-mers for its keys
        u64 Ut = 0;
        for (int i = 0; i < n_dbs; i++) Ut += dbs[i]->U;
        if (Ut >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 k-mers in the files of the set");
        // (files of several lengths: the lengths are part of the k-mers' order and identity, DB.java:99-109 merges in
        // Sequence.compareTo order; a set whose files share a key keeps its files for the search over several lengths)
        const bool var = d0->min_k != d0->k;
        db->min_k = d0->min_k;
        PoolBuf ch(ctx), cl(ctx), cn(ctx), rank(ctx), d_hi(ctx), d_lo(ctx), d_len(ctx);
        KMG_ALLOC(ch, (Ut + 1) * 8); KMG_ALLOC(cl, (Ut + 1) * 8); KMG_ALLOC(rank, (Ut + 1) * 8);
        if (var) KMG_ALLOC(cn, Ut + 64);
        u64 at = 0;
        for (int i = 0; i < n_dbs; i++) {
            const kmg_db *d = dbs[i];
            if (!d->U) continue;
            KMG_HIP(hipMemcpyAsync(ch.as<u64>() + at, d->keys, d->U * 8, hipMemcpyDeviceToDevice, ctx->stream));
            KMG_HIP(hipMemcpyAsync(cl.as<u64>() + at, d->keys_lo, d->U * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (var) {
                if (d->lens) KMG_HIP(hipMemcpyAsync(cn.as<uint8_t>() + at, d->lens, d->U, hipMemcpyDeviceToDevice, ctx->stream));
                else KMG_HIP(hipMemsetAsync(cn.as<uint8_t>() + at, d->k, d->U, ctx->stream));
            }
            at += d->U;
        }
        int rbits = 1;
        u64 U = 0;
        if (Ut) {
            WideOrder w(ctx);
            KMG_TRY(wide_order(ctx, ch.as<u64>(), cl.as<u64>(), (u32)Ut, d0->k, w, var ? cn.as<uint8_t>() : nullptr));
            U = w.U; rbits = w.rbits;
            KMG_ALLOC(d_hi, (U + 1) * 8); KMG_ALLOC(d_lo, (U + 1) * 8);
            if (var) KMG_ALLOC(d_len, U + 64);
            hipLaunchKernelGGL(wide_rank_kernel, dim3(div_up_u64(Ut, 256)), dim3(256), 0, ctx->stream, w.his, (const u64 *)w.los, w.order,
                               w.flag.as<u32>(), w.before.as<u64>(), (u32)Ut, rbits, rank.as<u64>(), d_hi.as<u64>(), d_lo.as<u64>(),
                               (const uint8_t *)w.lens.p, d_len.as<uint8_t>());
            KMG_HIP(hipGetLastError());
        } else { KMG_ALLOC(d_hi, 8); KMG_ALLOC(d_lo, 8); if (var) KMG_ALLOC(d_len, 64); }
        at = 0;
        for (int i = 0; i < n_dbs; i++) {
            const kmg_db *d = dbs[i];
            if (d->V) {
                if (reads && d->vals_packed32) KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(d)));
                if (reads && !d->vals_packed && !d->val_a) return fail(ctx, KMG_ESTATE, "database has no values");
                hipLaunchKernelGGL(mg_expand_kernel, dim3(div_up_u64(d->V, 256)), dim3(256), 0, ctx->stream, rank.as<u64>() + at, d->val_off,
                                   d->U, d->V, d->val_a, d->val_b, reads ? d->vals_packed : nullptr, d->kind, k0.as<u64>(), v0.as<u64>(), base);
                base += d->V;
            }
            at += d->U;
        }
        KMG_TRY(merge_records(ctx, db, k0, v0, k1, v1, n, totals, rbits));
        if (db->U != U) return fail(ctx, KMG_EDEVICE, "two-word merge: the ranks' distinct count differs from the k-mers'");
        if (var && n_dbs > 1) {
            // the search over several lengths takes a key's k-mers file by file (DB.java:154-158): a set whose files share a
            // key is marked and keeps its files
            const u32 nbk = 1u << (2 * d0->key_len);
            PoolBuf sb(ctx), sv(ctx), holders(ctx), clash(ctx);
            KMG_ALLOC(sb, ((size_t)nbk + 1) * 8); KMG_ALLOC(sv, ((size_t)nbk + 1) * 8); KMG_ALLOC(holders, ((size_t)nbk + 1) * 4); KMG_ALLOC(clash, 64);
            KMG_HIP(hipMemsetAsync(sb.p, 0, ((size_t)nbk + 1) * 8, ctx->stream));
            KMG_HIP(hipMemsetAsync(sv.p, 0, ((size_t)nbk + 1) * 8, ctx->stream));
            KMG_HIP(hipMemsetAsync(holders.p, 0, ((size_t)nbk + 1) * 4, ctx->stream));
            KMG_HIP(hipMemsetAsync(clash.p, 0, 4, ctx->stream));
            for (int i = 0; i < n_dbs; i++)
                hipLaunchKernelGGL(vm_sum_kernel, dim3(div_up_u64(nbk + 1, 256)), dim3(256), 0, ctx->stream, dbs[i]->bucket_off, dbs[i]->val_off, nbk,
                                   sb.as<u64>(), sv.as<u64>(), holders.as<u32>());
            hipLaunchKernelGGL(vm_clash_kernel, dim3(div_up_u64(nbk, 256)), dim3(256), 0, ctx->stream, holders.as<u32>(), nbk, clash.as<u32>());
            u32 h_clash = 0;
            KMG_HIP(read_back(ctx, &h_clash, clash.p, 4));
            db->shared_key_merge = h_clash != 0;
            if (db->shared_key_merge && d0->kind == KMG_PAYLOAD_TAXON)
                for (int i = 0; i < n_dbs; i++) {
                    // the files as they are, in their order: what kmg_match_trie searches (trie_files_wide_kernel)
                    if (dbs[i]->parts.empty()) {
                        kmg_db *part = nullptr;
                        KMG_TRY(clone_for_trie(ctx, dbs[i], &part));
                        db->parts.push_back(part);
                    } else
                        for (const kmg_db *q : dbs[i]->parts) {
                            kmg_db *part = nullptr;
                            KMG_TRY(clone_for_trie(ctx, q, &part));
                            db->parts.push_back(part);
                        }
                }
        }
        ctx->pool.release(db->keys);
        db->keys = (u64 *)d_hi.detach();
        db->keys_lo = (u64 *)d_lo.detach();
        db->lens = (uint8_t *)d_len.detach();
        KMG_TRY(db_make_buckets(ctx, db));
        *out = db_guard.release();
        return KMG_OK;
    }
    for (int i = 0; i < n_dbs; i++) {
        const kmg_db *d = dbs[i];
        if (d->V == 0) continue;
        if (reads && d->vals_packed32) KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(d)));
        if (reads && !d->vals_packed && !d->val_a) return fail(ctx, KMG_ESTATE, "database has no values");
        hipLaunchKernelGGL(mg_expand_kernel, dim3(div_up_u64(d->V, 256)), dim3(256), 0, ctx->stream, d->keys, d->val_off, d->U, d->V,
                           d->val_a, d->val_b, reads ? d->vals_packed : nullptr, d->kind, k0.as<u64>(), v0.as<u64>(), base);
        base += d->V;
    }
    KMG_TRY(merge_records(ctx, db, k0, v0, k1, v1, n, totals));
    *out = db_guard.release();
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ restrict
static int restrict_locked(kmg_ctx *ctx, const kmg_db *src, int length, kmg_db **out);

extern "C" int kmg_db_restrict(kmg_ctx *ctx, const kmg_db *src, int length, kmg_db **out) {
    if (!ctx || !src || !out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (src->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    if (length < src->min_k || length > src->k)                                                     // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (length < src->key_len) return fail(ctx, KMG_EINVAL, "length below the key length");
    return restrict_locked(ctx, src, length, out);
}

// The same for a database of two-word k-mers (wide.h): the entries of at least min_length bases, those beyond max_length
// cut to it, with their lengths min(length, max_length), are ranked in (value, length) order among the distinct ones;
// their values go through the fold under their ranks; the result holds the distinct (k-mer, length) pairs -- one word
// each if max_length <= 32.  Caller holds ctx->mu.
static int restrict_range_wide(kmg_ctx *ctx, const kmg_db *src, int min_length, int max_length, kmg_db **out) {
    if (min_length > max_length || min_length < src->key_len || max_length > 64)
        return fail(ctx, KMG_EINVAL, "key length <= min_length <= max_length <= 64");
    KMG_HIP(hipSetDevice(ctx->device));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(src)));
    const u64 U = src->U;
    if (U >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 k-mers in the database");
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    db->ctx = ctx; db->k = max_length; db->min_k = min_length; db->key_len = src->key_len; db->rc = src->rc; db->kind = src->kind;
    PoolBuf kflag(ctx), kcnt(ctx), kpos(ctx), dst(ctx), agg(ctx), ch(ctx), cl(ctx), cn(ctx), from(ctx), rank_k(ctx), rank(ctx), d_hi(ctx), d_lo(ctx),
        d_len(ctx), k0(ctx), v0(ctx), k1(ctx), v1(ctx), totals(ctx);
    KMG_ALLOC(totals, 64);
    KMG_HIP(hipMemsetAsync(totals.p, 0, 64, ctx->stream));
    KMG_ALLOC(kflag, (U + 1) * 4); KMG_ALLOC(kcnt, (U + 1) * 4); KMG_ALLOC(kpos, (U + 2) * 8); KMG_ALLOC(dst, (U + 2) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    KMG_ALLOC(rank, (U + 1) * 8);
    u64 nk = 0, nvals = 0, Uw = 0;
    int rbits = 1;
    if (U) {
        // kflag = the entry stays, kcnt = its values
        hipLaunchKernelGGL(vr_flag_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, src->lens, src->k, src->val_off, U, min_length, 65,
                           kflag.as<u32>(), kcnt.as<u32>());
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, kflag.as<u32>(), kpos.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true, KP_SCAN)));
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, kcnt.as<u32>(), dst.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true, KP_SCAN)));
        KMG_HIP(read_back(ctx, &nk, kpos.as<u64>() + U, 8));
        KMG_HIP(read_back(ctx, &nvals, dst.as<u64>() + U, 8));
    }
    if (nvals >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 values in the restricted database");
    const u32 n = (u32)nvals;
    if (nk) {
        KMG_ALLOC(ch, nk * 8); KMG_ALLOC(cl, nk * 8); KMG_ALLOC(cn, nk + 64); KMG_ALLOC(from, nk * 4); KMG_ALLOC(rank_k, nk * 8);
        hipLaunchKernelGGL(wide_compact_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, (const u64 *)src->keys, (const u64 *)src->keys_lo,
                           kflag.as<u32>(), kpos.as<u64>(), U, max_length, ch.as<u64>(), cl.as<u64>(), from.as<u32>(), (const uint8_t *)src->lens,
                           src->k, cn.as<uint8_t>());
        WideOrder w(ctx);
        // (the order's second word has 2 * (k - 32) bits: 33 stands for "none beyond the first word")
        KMG_TRY(wide_order(ctx, ch.as<u64>(), cl.as<u64>(), (u32)nk, std::max(max_length, 33), w, cn.as<uint8_t>()));
        Uw = w.U; rbits = w.rbits;
        KMG_ALLOC(d_hi, (Uw + 1) * 8); KMG_ALLOC(d_lo, (Uw + 1) * 8); KMG_ALLOC(d_len, Uw + 64);
        hipLaunchKernelGGL(wide_rank_kernel, dim3(div_up_u64(nk, 256)), dim3(256), 0, ctx->stream, w.his, (const u64 *)w.los, w.order,
                           w.flag.as<u32>(), w.before.as<u64>(), (u32)nk, rbits, rank_k.as<u64>(), d_hi.as<u64>(), d_lo.as<u64>(),
                           (const uint8_t *)w.lens.p, d_len.as<uint8_t>());
        hipLaunchKernelGGL(wide_spread_kernel, dim3(div_up_u64(nk, 256)), dim3(256), 0, ctx->stream, (const u64 *)rank_k.as<u64>(),
                           (const u32 *)from.as<u32>(), nk, rank.as<u64>());
        KMG_HIP(hipGetLastError());
    } else { KMG_ALLOC(d_hi, 8); KMG_ALLOC(d_lo, 8); KMG_ALLOC(d_len, 64); }
    KMG_ALLOC(k0, ((size_t)n + 1) * 8); KMG_ALLOC(v0, ((size_t)n + 1) * 8);
    KMG_ALLOC(k1, ((size_t)n + 1) * 8); KMG_ALLOC(v1, ((size_t)n + 1) * 8);
    if (n)
        hipLaunchKernelGGL(rs_expand_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, (const u64 *)rank.as<u64>(), (const uint8_t *)src->lens,
                           src->k, src->val_off, src->val_a, src->val_b, U, min_length, 65, dst.as<u64>(), 0ull, src->kind, k0.as<u64>(),
                           v0.as<u64>(), false);
    KMG_HIP(hipGetLastError());
    KMG_TRY(merge_records(ctx, db, k0, v0, k1, v1, n, totals, rbits));
    if (db->U != Uw) return fail(ctx, KMG_EDEVICE, "two-word restriction: the ranks' distinct count differs from the k-mers'");
    ctx->pool.release(db->keys);
    db->keys = (u64 *)d_hi.detach();
    if (max_length > 32) db->keys_lo = (u64 *)d_lo.detach();
    db->lens = (uint8_t *)d_len.detach();
    KMG_TRY(db_make_buckets(ctx, db));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    *out = db_guard.release();
    return KMG_OK;
}

// KmerUtils.restrictedStream(stream, minLength, maxLength, dataType) with minLength < maxLength (KmerUtils.java:59-74):
// the k-mers of at least maxLength bases, cut and folded, are a database of one length (restrict_locked); those of
// minLength .. maxLength - 1 bases stay as they are; the two interleave by rank as the length classes of a build do.
extern "C" int kmg_db_restrict_range(kmg_ctx *ctx, const kmg_db *src, int min_length, int max_length, kmg_db **out) {
    if (!ctx || !src || !out) return KMG_EINVAL;
    if (min_length == max_length) return kmg_db_restrict(ctx, src, min_length, out);
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (src->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    if (src->k > 32) return restrict_range_wide(ctx, src, min_length, max_length, out);
    if (min_length > max_length || min_length < src->key_len || max_length > 32)
        return fail(ctx, KMG_EINVAL, "key length <= min_length <= max_length <= 32");
    KMG_HIP(hipSetDevice(ctx->device));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(src)));
    const int cut = std::min(max_length, src->k);          // (nothing is longer than src->k: nothing else is cut)
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> Y(nullptr, db_release), X(new (std::nothrow) kmg_db(), db_release),
        res(new (std::nothrow) kmg_db(), db_release);
    if (!X || !res) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    if (cut >= min_length) {
        kmg_db *y = nullptr;
        KMG_TRY(restrict_locked(ctx, src, cut, &y));
        Y.reset(y);
    } else {                                               // every stored k-mer is shorter than min_length: nothing stays
        Y.reset(new (std::nothrow) kmg_db());
        if (!Y) return fail(ctx, KMG_ENOMEM, "host allocation failed");
        Y->ctx = ctx; Y->k = src->k; Y->min_k = src->k; Y->key_len = src->key_len; Y->rc = src->rc; Y->kind = src->kind;
    }
    // X: the entries of min_length <= length < cut
    const u64 U = src->U;
    X->ctx = ctx; X->k = src->k; X->min_k = src->min_k; X->key_len = src->key_len; X->rc = src->rc; X->kind = src->kind;
    PoolBuf keep(ctx), nv(ctx), dk(ctx), dv(ctx), agg(ctx);
    KMG_ALLOC(keep, (U + 1) * 4); KMG_ALLOC(nv, (U + 1) * 4);
    KMG_ALLOC(dk, (U + 2) * 8); KMG_ALLOC(dv, (U + 2) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    u64 xu = 0, xv = 0;
    if (U && cut > min_length) {
        hipLaunchKernelGGL(vr_flag_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, src->lens, src->k, src->val_off, U, min_length,
                           cut, keep.as<u32>(), nv.as<u32>());
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, keep.as<u32>(), dk.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true, KP_SCAN)));
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, nv.as<u32>(), dv.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true, KP_SCAN)));
        KMG_HIP(read_back(ctx, &xu, dk.as<u64>() + U, 8));
        KMG_HIP(read_back(ctx, &xv, dv.as<u64>() + U, 8));
    }
    void *p;
    if (ctx->pool.alloc(&p, (xu + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    X->keys = (u64 *)p;
    if (ctx->pool.alloc(&p, xu + 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    X->lens = (uint8_t *)p;
    if (ctx->pool.alloc(&p, (xu + 2) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    X->val_off = (u64 *)p;
    if (ctx->pool.alloc(&p, (xv + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    X->val_a = (int32_t *)p;
    if (ctx->pool.alloc(&p, (xv + 1) * 4) != KMG_OK) return fail(ctx, KMG_ENOMEM, "db allocation failed");
    X->val_b = (int32_t *)p;
    X->U = xu; X->V = xv;
    if (xu)
        hipLaunchKernelGGL(vr_gather_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, src->keys, src->lens, src->k, src->val_off,
                           src->val_a, src->val_b, U, keep.as<u32>(), dk.as<u64>(), dv.as<u64>(), X->keys, X->lens, X->val_off, X->val_a,
                           X->val_b);
    KMG_HIP(hipMemcpyAsync(X->val_off + xu, &xv, 8, hipMemcpyHostToDevice, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));            // (xv is on this frame)
    res->ctx = ctx; res->k = max_length; res->min_k = min_length; res->key_len = src->key_len; res->rc = src->rc; res->kind = src->kind;
    KMG_TRY(interleave_dbs(ctx, X.get(), Y.get(), res.get()));
    KMG_TRY(db_make_buckets(ctx, res.get()));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    KMG_HIP(hipGetLastError());
    *out = res.release();
    return KMG_OK;
}

static int restrict_locked(kmg_ctx *ctx, const kmg_db *src, int length, kmg_db **out) {
    KMG_HIP(hipSetDevice(ctx->device));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(src)));
    const bool reads = src->kind == KMG_PAYLOAD_READPOS;
    const u64 U = src->U;
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    db->ctx = ctx; db->k = length; db->min_k = length; db->key_len = src->key_len; db->rc = src->rc; db->kind = src->kind;
    PoolBuf keep(ctx), dst(ctx), agg(ctx), k0(ctx), v0(ctx), k1(ctx), v1(ctx), totals(ctx);
    KMG_ALLOC(totals, 64);
    KMG_HIP(hipMemsetAsync(totals.p, 0, 64, ctx->stream));
    KMG_ALLOC(keep, (U + 1) * 4);
    KMG_ALLOC(dst, (U + 2) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    u64 nkept = 0;
    // two-word k-mers cut to more than 32 bases (wide.h): the cut k-mers are ranked among the distinct ones and the ranks
    // go through the fold as one-word keys; cut to 32 bases or fewer, the first word is all that is left of them
    const bool wide_out = length > 32;
    PoolBuf cut_lo(ctx), rank(ctx), d_hi(ctx), d_lo(ctx);
    int rbits = 0;
    u64 Uw = 0;
    if (wide_out) {
        KMG_ALLOC(rank, (U + 1) * 8);
        if (U >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 k-mers in the database");
        // the entries of at least `length` bases (all of them in a database of one length), cut, side by side
        PoolBuf kflag(ctx), kcnt(ctx), kpos(ctx), ch(ctx), from(ctx), rank_k(ctx);
        u64 nk = 0;
        if (U) {
            KMG_ALLOC(kflag, (U + 1) * 4); KMG_ALLOC(kcnt, (U + 1) * 4); KMG_ALLOC(kpos, (U + 2) * 8);
            hipLaunchKernelGGL(vr_flag_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, src->lens, src->k, src->val_off, U, length, 65,
                               kflag.as<u32>(), kcnt.as<u32>());
            KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, kflag.as<u32>(), kpos.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                        KP_SCAN)));
            KMG_HIP(read_back(ctx, &nk, kpos.as<u64>() + U, 8));
        }
        if (nk) {
            KMG_ALLOC(ch, nk * 8); KMG_ALLOC(cut_lo, nk * 8); KMG_ALLOC(from, nk * 4); KMG_ALLOC(rank_k, nk * 8);
            hipLaunchKernelGGL(wide_compact_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, (const u64 *)src->keys,
                               (const u64 *)src->keys_lo, kflag.as<u32>(), kpos.as<u64>(), U, length, ch.as<u64>(), cut_lo.as<u64>(), from.as<u32>());
            WideOrder w(ctx);
            KMG_TRY(wide_order(ctx, ch.as<u64>(), cut_lo.as<u64>(), (u32)nk, length, w));
            Uw = w.U; rbits = w.rbits;
            KMG_ALLOC(d_hi, (Uw + 1) * 8); KMG_ALLOC(d_lo, (Uw + 1) * 8);
            hipLaunchKernelGGL(wide_rank_kernel, dim3(div_up_u64(nk, 256)), dim3(256), 0, ctx->stream, w.his, (const u64 *)w.los, w.order,
                               w.flag.as<u32>(), w.before.as<u64>(), (u32)nk, rbits, rank_k.as<u64>(), d_hi.as<u64>(), d_lo.as<u64>(),
                               (const uint8_t *)nullptr, (uint8_t *)nullptr);
            hipLaunchKernelGGL(wide_spread_kernel, dim3(div_up_u64(nk, 256)), dim3(256), 0, ctx->stream, (const u64 *)rank_k.as<u64>(),
                               (const u32 *)from.as<u32>(), nk, rank.as<u64>());
            KMG_HIP(hipGetLastError());
        } else { KMG_ALLOC(d_hi, 8); KMG_ALLOC(d_lo, 8); rbits = 1; }
    }
    // (what the record kernels see: the ranks are "k-mers of 32 bases" that nothing is cut from)
    const u64 *rkeys = wide_out ? rank.as<u64>() : src->keys;
    const uint8_t *rlens = src->lens;
    const int rlen_all = src->k, rcut = length;
    if (U) {
        hipLaunchKernelGGL(rs_keep_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, rlens, rlen_all, src->val_off, U, rcut, 65,
                           keep.as<u32>());
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, keep.as<u32>(), dst.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_HIP(read_back(ctx, &nkept, dst.as<u64>() + U, 8));
    }
    if (nkept >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 values in the restricted database");
    const u32 n = (u32)nkept;
    KMG_ALLOC(k0, ((size_t)n + 1) * 8); KMG_ALLOC(v0, ((size_t)n + 1) * 8);
    KMG_ALLOC(k1, ((size_t)n + 1) * 8); KMG_ALLOC(v1, ((size_t)n + 1) * 8);
    if (n)
        hipLaunchKernelGGL(rs_expand_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, rkeys, rlens, rlen_all, src->val_off,
                           src->val_a, src->val_b, U, rcut, 65, dst.as<u64>(), 0ull, src->kind, k0.as<u64>(), v0.as<u64>(), !wide_out);
    KMG_HIP(hipGetLastError());
    (void)reads;
    KMG_TRY(merge_records(ctx, db, k0, v0, k1, v1, n, totals, rbits));
    if (wide_out) {
        if (db->U != Uw) return fail(ctx, KMG_EDEVICE, "two-word restriction: the ranks' distinct count differs from the k-mers'");
        ctx->pool.release(db->keys);
        db->keys = (u64 *)d_hi.detach();
        db->keys_lo = (u64 *)d_lo.detach();
        KMG_TRY(db_make_buckets(ctx, db));
    }
    *out = db_guard.release();
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ trie search
extern "C" int kmg_match_trie(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int min_k, int count_cap, kmg_db **rows,
                              int64_t *hit_row, uint64_t *n_hits) {
    if (!ctx || !search || !ref || !rows || (search->U && !hit_row)) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *rows = nullptr;
    if (n_hits) *n_hits = 0;
    if (search->ctx != ctx || ref->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    if (ref->kind != KMG_PAYLOAD_TAXON) return fail(ctx, KMG_EINVAL, "the database searched holds taxon counts");
    // (two-word k-mers, wide.h: one file, or a set whose files share no key; the search side one word or two as the database)
    const bool wide = ref->k > 32;
    if (search->key_len != ref->key_len) return fail(ctx, KMG_EINVAL, "key lengths differ");
    // DB.java:78-81 on the search stream's declared lengths [min_k, search->k]
    if (min_k < ref->min_k || search->k > ref->k || min_k > search->min_k || min_k < ref->key_len)
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    // The reference puts a key's k-mers into its trie FILE BY FILE, unmerged (DB.java:154-158): with files that share a key
    // the insertion is no longer in the trie's own preorder, nodes are split by later files' k-mers and a data object can
    // receive one k-mer more than once -- the single-file closed form does not hold for that; the database side then goes
    // through match.h's second closed form over the files the merged database kept (trie_files_kernel).
    // The SEARCH side is a stream the reference does not fold at all (mergeSortedStreams without a reduce,
    // Matcher.java:241: a k-mer two files hold is asked twice), which a merged handle cannot stand for: refused.
    if (search->shared_key_merge)
        return fail(ctx, KMG_EUNSUPPORTED, "search files of several k-mer lengths that share a key: not on the GPU path");
    const bool by_files = ref->shared_key_merge;
    if (by_files && ref->parts.empty()) return fail(ctx, KMG_ESTATE, "merged database without its files");
    KMG_HIP(hipSetDevice(ctx->device));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(ref)));
    const u64 ns = search->U;
    kmg_db *db = new (std::nothrow) kmg_db();
    if (!db) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    std::unique_ptr<kmg_db, void (*)(kmg_db *)> db_guard(db, db_release);
    // the rows' "k-mer" is the row number in 32 bits: a database of 16-base keys as far as the merge pipeline cares
    db->ctx = ctx; db->k = 16; db->min_k = 16; db->key_len = 1; db->rc = true; db->kind = KMG_PAYLOAD_TAXON;
    PoolBuf first(ctx), found(ctx), nvals(ctx), row(ctx), dst(ctx), agg(ctx), k0(ctx), v0(ctx), k1(ctx), v1(ctx), totals(ctx), hit(ctx);
    KMG_ALLOC(totals, 64);
    KMG_HIP(hipMemsetAsync(totals.p, 0, 64, ctx->stream));
    KMG_ALLOC(first, (ns + 1) * 8); KMG_ALLOC(found, (ns + 1) * 4); KMG_ALLOC(nvals, (ns + 1) * 4);
    KMG_ALLOC(row, (ns + 2) * 8); KMG_ALLOC(dst, (ns + 2) * 8); KMG_ALLOC(hit, (ns + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(ns, SCAN_TILE) + 1) * 8);
    u64 nrows = 0, nrec = 0;
    PoolBuf d_files(ctx), nvals64(ctx);
    const int nf = (int)ref->parts.size();
    if (ns && by_files && wide) {
        std::vector<WideTrieFile> h_files;
        for (const kmg_db *q : ref->parts) {
            if (!q->keys_lo) return fail(ctx, KMG_ESTATE, "a file of the set holds one-word k-mers");
            h_files.push_back(WideTrieFile{TrieFile{q->keys, q->lens, q->bucket_off, q->val_off, q->val_a, q->val_b, q->k, 0}, q->keys_lo});
        }
        KMG_ALLOC(d_files, h_files.size() * sizeof(WideTrieFile));
        KMG_ALLOC(nvals64, (ns + 1) * 8);
        KMG_HIP(hipMemcpyAsync(d_files.p, h_files.data(), h_files.size() * sizeof(WideTrieFile), hipMemcpyHostToDevice, ctx->stream));
        KMG_HIP(hipStreamSynchronize(ctx->stream));             // (h_files is pageable and on this frame)
        hipLaunchKernelGGL((trie_files_wide_kernel<false>), dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, (const u64 *)search->keys,
                           (const u64 *)search->keys_lo, (const uint8_t *)search->lens, search->k, ns, d_files.as<WideTrieFile>(), nf, ref->key_len,
                           min_k, found.as<u32>(), nvals64.as<u64>(), (const u64 *)nullptr, (const u64 *)nullptr, count_cap, (u64 *)nullptr,
                           (u64 *)nullptr, (long long *)nullptr);
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, found.as<u32>(), row.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_TRY((device_scan<u64, u64, OpSum, true>(ctx, nvals64.as<u64>(), dst.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_HIP(read_back(ctx, &nrows, row.as<u64>() + ns, 8));
        KMG_HIP(read_back(ctx, &nrec, dst.as<u64>() + ns, 8));
    } else if (ns && by_files) {
        std::vector<TrieFile> h_files;
        for (const kmg_db *q : ref->parts)
            h_files.push_back(TrieFile{q->keys, q->lens, q->bucket_off, q->val_off, q->val_a, q->val_b, q->k, 0});
        KMG_ALLOC(d_files, h_files.size() * sizeof(TrieFile));
        KMG_ALLOC(nvals64, (ns + 1) * 8);
        KMG_HIP(hipMemcpyAsync(d_files.p, h_files.data(), h_files.size() * sizeof(TrieFile), hipMemcpyHostToDevice, ctx->stream));
        KMG_HIP(hipStreamSynchronize(ctx->stream));             // (h_files is pageable and on this frame)
        hipLaunchKernelGGL((trie_files_kernel<false>), dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, search->keys, search->lens,
                           search->k, ns, d_files.as<TrieFile>(), nf, ref->key_len, min_k, found.as<u32>(), nvals64.as<u64>(),
                           (const u64 *)nullptr, (const u64 *)nullptr, count_cap, (u64 *)nullptr, (u64 *)nullptr, (long long *)nullptr);
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, found.as<u32>(), row.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_TRY((device_scan<u64, u64, OpSum, true>(ctx, nvals64.as<u64>(), dst.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_HIP(read_back(ctx, &nrows, row.as<u64>() + ns, 8));
        KMG_HIP(read_back(ctx, &nrec, dst.as<u64>() + ns, 8));
    } else if (ns) {
        if (wide)
            hipLaunchKernelGGL(trie_find_wide_kernel, dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, search->keys, (const u64 *)search->keys_lo,
                               search->lens, search->k, ns, ref->keys, ref->keys_lo, ref->lens, ref->k, ref->bucket_off, ref->val_off,
                               ref->key_len, min_k, first.as<long long>(), found.as<u32>(), nvals.as<u32>());
        else
        hipLaunchKernelGGL(trie_find_kernel, dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, search->keys, search->lens, search->k,
                           ns, ref->keys, ref->lens, ref->k, ref->bucket_off, ref->val_off, ref->key_len, min_k,
                           first.as<long long>(), found.as<u32>(), nvals.as<u32>());
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, found.as<u32>(), row.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, nvals.as<u32>(), dst.as<u64>(), ns, OpSum(), 0ull, agg.as<u64>(), nullptr, true,
                                                    KP_SCAN)));
        KMG_HIP(read_back(ctx, &nrows, row.as<u64>() + ns, 8));
        KMG_HIP(read_back(ctx, &nrec, dst.as<u64>() + ns, 8));
    }
    if (nrec >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 values in the answers of one call: search in key slices");
    const u32 n = (u32)nrec;
    KMG_ALLOC(k0, ((size_t)n + 1) * 8); KMG_ALLOC(v0, ((size_t)n + 1) * 8);
    KMG_ALLOC(k1, ((size_t)n + 1) * 8); KMG_ALLOC(v1, ((size_t)n + 1) * 8);
    if (ns && by_files && wide)
        hipLaunchKernelGGL((trie_files_wide_kernel<true>), dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, (const u64 *)search->keys,
                           (const u64 *)search->keys_lo, (const uint8_t *)search->lens, search->k, ns, d_files.as<WideTrieFile>(), nf, ref->key_len,
                           min_k, (u32 *)nullptr, (u64 *)nullptr, row.as<u64>(), dst.as<u64>(), count_cap, k0.as<u64>(), v0.as<u64>(),
                           hit.as<long long>());
    else if (ns && by_files)
        hipLaunchKernelGGL((trie_files_kernel<true>), dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, search->keys, search->lens,
                           search->k, ns, d_files.as<TrieFile>(), nf, ref->key_len, min_k, (u32 *)nullptr, (u64 *)nullptr,
                           row.as<u64>(), dst.as<u64>(), count_cap, k0.as<u64>(), v0.as<u64>(), hit.as<long long>());
    else if (ns)
        hipLaunchKernelGGL(trie_expand_kernel, dim3(div_up_u64(ns, 256)), dim3(256), 0, ctx->stream, first.as<long long>(), nvals.as<u32>(),
                           row.as<u64>(), dst.as<u64>(), ns, ref->val_off, ref->val_a, ref->val_b, count_cap, k0.as<u64>(), v0.as<u64>(),
                           hit.as<long long>());
    KMG_HIP(hipGetLastError());
    if (ns) KMG_HIP(hipMemcpyAsync(hit_row, hit.p, ns * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_TRY(merge_records(ctx, db, k0, v0, k1, v1, n, totals));          // (ends with a stream synchronisation)
    if (db->U != nrows) return fail(ctx, KMG_ESTATE, "trie search: rows and answered k-mers differ");
    if (n_hits) *n_hits = nrows;
    *rows = db_guard.release();
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ match
static int match_locked(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, long long *d_hit, u64 *d_nhits) {
    // databases of another context of the same device are fine (they are complete once their build has
    // finished): a caller may build the reference and the reads on two contexts side by side
    if (search->ctx->device != ctx->device || ref->ctx->device != ctx->device)
        return fail(ctx, KMG_EINVAL, "database belongs to another device");
    // DB.java:78-81: search k-mers must have the database's length; keys are compared per key bucket
    if (search->k != ref->k || search->key_len != ref->key_len)
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    // DB.java:83: the quick branch takes k-mers of one length -- restrict both sides first (kmg_db_restrict)
    if (search->min_k != search->k || ref->min_k != ref->k)
        return fail(ctx, KMG_EUNSUPPORTED, "k-mers of several lengths: restrict the databases to one length first");
    if (search->U == 0) return KMG_OK;
    if (search->k > 32) {
        // two-word k-mers (wide.h): every tile of search k-mers against the piece of the reference it can meet
        ProfScope ps(ctx, KP_MATCH, 16 * (search->U + ref->U) + 8 * search->U);
        hipLaunchKernelGGL(wide_match_kernel, dim3(div_up_u64(search->U, WM_TILE)), dim3(WM_BLOCK), 0, ctx->stream, search->keys,
                           search->keys_lo, search->U, ref->keys, ref->keys_lo, ref->U, d_hit, d_nhits);
        KMG_HIP(hipGetLastError());
        KMG_TRY(note_foreign_use(ctx, search));
        KMG_TRY(note_foreign_use(ctx, ref));
        return KMG_OK;
    }
    ProfScope ps(ctx, KP_MATCH, 8 * (search->U + ref->U) + 8 * search->U);
    const u32 ntiles = div_up_u64(search->U, MT_TILE);
    PoolBuf bounds(ctx);
    KMG_ALLOC(bounds, ((size_t)ntiles + 1) * 8);
    hipLaunchKernelGGL(match_bounds_kernel, dim3(div_up_u64(ntiles + 1, 256)), dim3(256), 0, ctx->stream, search->keys, search->U,
                       ref->keys, ref->U, ref->bucket_off, ref->key_len, ntiles, bounds.as<u64>());
    hipLaunchKernelGGL(match_kernel, dim3(ntiles), dim3(MT_BLOCK), 0, ctx->stream, search->keys, search->U, ref->keys,
                       bounds.as<u64>(), d_hit, d_nhits);
    KMG_HIP(hipGetLastError());
    KMG_TRY(note_foreign_use(ctx, search));
    KMG_TRY(note_foreign_use(ctx, ref));
    return KMG_OK;
}

extern "C" int kmg_match_dev(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *d_hit_index) {
    if (!ctx || !search || !ref || !d_hit_index) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    return match_locked(ctx, search, ref, (long long *)d_hit_index, nullptr);
}

static int match_keys_locked(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *run_off, int n_runs, int k,
                             const kmg_db *ref, int64_t *d_hit_index);

// the same for two-word k-mers (wide.h): every ascending run is a search side of its own for wide_match_kernel
static int match_keys_wide_locked(kmg_ctx *ctx, const u64 *d_hi, const u64 *d_lo, const uint64_t *run_off, int n_runs, int k,
                                  const kmg_db *ref, int64_t *d_hit_index) {
    if (ref->ctx->device != ctx->device) return fail(ctx, KMG_EINVAL, "database belongs to another device");
    if (k != ref->k || !ref->keys_lo)                                                            // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (ref->min_k != ref->k) return fail(ctx, KMG_EUNSUPPORTED, "k-mers of several lengths: restrict the database to one length first");
    for (int r = 0; r < n_runs; r++) {
        if (run_off[r] > run_off[r + 1]) return fail(ctx, KMG_EINVAL, "run_off must be non-decreasing");
        const u64 a = run_off[r], m = run_off[r + 1] - a;
        if (m == 0) continue;
        if (div_up_u64(m, WM_TILE) >= 0x7FFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "too many search k-mers for one call");
        hipLaunchKernelGGL(wide_match_kernel, dim3(div_up_u64(m, WM_TILE)), dim3(WM_BLOCK), 0, ctx->stream, d_hi + a, d_lo + a, m,
                           (const u64 *)ref->keys, (const u64 *)ref->keys_lo, ref->U, (long long *)d_hit_index + a, (u64 *)nullptr);
    }
    KMG_HIP(hipGetLastError());
    KMG_TRY(note_foreign_use(ctx, ref));
    return KMG_OK;
}

extern "C" int kmg_match_keys_wide_dev(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_keys_lo, const uint64_t *run_off,
                                       int n_runs, int k, const kmg_db *ref, int64_t *d_hit_index) {
    if (!ctx || !ref || !run_off) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (n_runs < 1 || n_runs > MT_MAX_RUNS) return fail(ctx, KMG_EINVAL, "between 1 and 64 runs");
    if (k <= 32 || !ref->keys_lo) return fail(ctx, KMG_EINVAL, "k-mers of 33 .. 64 bases (kmg_match_keys_dev takes the shorter ones)");
    if (run_off[n_runs] > run_off[0] && (!d_keys || !d_keys_lo || !d_hit_index)) return fail(ctx, KMG_EINVAL, "null keys / hit_index");
    return match_keys_wide_locked(ctx, (const u64 *)d_keys, (const u64 *)d_keys_lo, run_off, n_runs, k, ref, d_hit_index);
}

extern "C" int kmg_match_keys_dev(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *run_off, int n_runs, int k,
                                  const kmg_db *ref, int64_t *d_hit_index) {
    if (!ctx || !ref || !run_off) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    return match_keys_locked(ctx, d_keys, run_off, n_runs, k, ref, d_hit_index);
}

static int match_keys_locked(kmg_ctx *ctx, const uint64_t *d_keys, const uint64_t *run_off, int n_runs, int k,
                             const kmg_db *ref, int64_t *d_hit_index) {
    if (ref->ctx->device != ctx->device) return fail(ctx, KMG_EINVAL, "database belongs to another device");
    if (ref->k > 32) return fail(ctx, KMG_EUNSUPPORTED, "k-mers of more than 32 bases: kmg_match_keys_wide_dev takes both words");
    if (k != ref->k)                                                                             // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (ref->min_k != ref->k) return fail(ctx, KMG_EUNSUPPORTED, "k-mers of several lengths: restrict the database to one length first");
    if (n_runs < 1 || n_runs > MT_MAX_RUNS) return fail(ctx, KMG_EINVAL, "between 1 and 64 runs");
    MatchRuns runs;
    memset(&runs, 0, sizeof runs);
    runs.n_runs = (u32)n_runs;
    u64 tiles = 0;
    for (int r = 0; r < n_runs; r++) {
        if (run_off[r] > run_off[r + 1]) return fail(ctx, KMG_EINVAL, "run_off must be non-decreasing");
        runs.tile0[r] = (u32)tiles;
        runs.off[r] = run_off[r];
        tiles += (run_off[r + 1] - run_off[r] + MT_TILE - 1) / MT_TILE;
        if (tiles >= 0x7FFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "too many search k-mers for one call");
    }
    runs.tile0[n_runs] = (u32)tiles;
    runs.off[n_runs] = run_off[n_runs];
    const u64 n = run_off[n_runs] - run_off[0];
    if (n == 0) return KMG_OK;
    if (!d_keys || !d_hit_index) return fail(ctx, KMG_EINVAL, "null keys / hit_index");
    const u32 ntiles = (u32)tiles;
    ProfScope ps(ctx, KP_MATCH, 8 * (n + ref->U) + 8 * n);
    PoolBuf tb(ctx);
    KMG_ALLOC(tb, (size_t)ntiles * sizeof(MatchTile));
    hipLaunchKernelGGL(match_runs_bounds_kernel, dim3(div_up_u64(ntiles, 256)), dim3(256), 0, ctx->stream, (const u64 *)d_keys, runs,
                       ref->keys, ref->bucket_off, ref->key_len, ntiles, tb.as<MatchTile>());
    hipLaunchKernelGGL(match_runs_kernel, dim3(ntiles), dim3(MT_BLOCK), 0, ctx->stream, (const u64 *)d_keys, ref->keys,
                       tb.as<MatchTile>(), (long long *)d_hit_index);
    KMG_HIP(hipGetLastError());
    KMG_TRY(note_foreign_use(ctx, ref));
    return KMG_OK;
}

// ---- the join for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid it) ----
// kmg_match_dev's only scratch is the per-tile bounds array (one u64 per MT_TILE search k-mers + 1).  Here it lives
// in a file-scope device array: every GPU the library runs on holds its own copy (1 MiB) until the process ends.
// In host code the array's name is a stand-in, not a device address -- kmg_create resolves the address once per
// device (hipGetSymbolAddress, that device current) and only that pointer is ever handed to a kernel.
#define KMG_NOALLOC_TILES 131071u
static __device__ __align__(256) u64 kmg_match_scratch[KMG_NOALLOC_TILES + 1];
#define KMG_MAX_DEVICES 64
static u64 *g_match_scratch[KMG_MAX_DEVICES];          // per device: address of ITS copy of kmg_match_scratch
// per device: the context whose join used the array last.  Another context's join waits for that context's event
// (made with the context, recorded behind every no-allocation join) before it writes the array: calls from several
// contexts of one GPU are ordered one behind the other instead of racing -- still without allocating.
static kmg_ctx *g_match_scratch_user[KMG_MAX_DEVICES];
static std::mutex g_match_scratch_mu;

// called by kmg_create with ctx->device current
static int resolve_match_scratch(kmg_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= KMG_MAX_DEVICES) return KMG_OK;       // no-alloc join not offered there
    std::lock_guard<std::mutex> g(g_match_scratch_mu);
    if (!g_match_scratch[ctx->device]) {
        void *p = nullptr;
        KMG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(kmg_match_scratch)));
        if (!p) return fail(ctx, KMG_EDEVICE, "hipGetSymbolAddress returned no address");
        g_match_scratch[ctx->device] = (u64 *)p;
    }
    ctx->match_scratch = g_match_scratch[ctx->device];
    return KMG_OK;
}

// a context that goes away (its stream has been synchronised: its joins are done) is nobody's predecessor any more
static void forget_match_scratch_user(kmg_ctx *ctx) {
    std::lock_guard<std::mutex> g(g_match_scratch_mu);
    for (int d = 0; d < KMG_MAX_DEVICES; d++)
        if (g_match_scratch_user[d] == ctx) g_match_scratch_user[d] = nullptr;
}

extern "C" int kmg_match_dev_noalloc(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *d_hit_index) {
    if (!ctx || !search || !ref || !d_hit_index) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (!ctx->match_scratch) return fail(ctx, KMG_EDEVICE, "no scratch array on this device");
    // (a foreign database would need an event of its own to order its release: that is an allocation)
    if (search->ctx != ctx || ref->ctx != ctx) return fail(ctx, KMG_EINVAL, "both databases must belong to this context");
    if (search->k != ref->k || search->key_len != ref->key_len)                                  // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (search->min_k != search->k || ref->min_k != ref->k)
        return fail(ctx, KMG_EUNSUPPORTED, "k-mers of several lengths: restrict the databases to one length first");
    if (search->U > (u64)KMG_MATCH_NOALLOC_MAX_SEARCH)
        return fail(ctx, KMG_EINVAL, "more search k-mers than the no-allocation join holds bounds for (KMG_MATCH_NOALLOC_MAX_SEARCH)");
    if (search->U == 0) return KMG_OK;
    if (search->k > 32) {
        // two-word k-mers (wide.h): that join finds every tile's piece of the reference inside the kernel -- no bounds
        // array, so nothing of the scratch array is touched and no other context has to be waited for
        hipLaunchKernelGGL(wide_match_kernel, dim3(div_up_u64(search->U, WM_TILE)), dim3(WM_BLOCK), 0, ctx->stream, (const u64 *)search->keys,
                           (const u64 *)search->keys_lo, search->U, (const u64 *)ref->keys, (const u64 *)ref->keys_lo, ref->U,
                           (long long *)d_hit_index, (u64 *)nullptr);
        KMG_HIP(hipGetLastError());
        return KMG_OK;
    }
    const u32 ntiles = div_up_u64(search->U, MT_TILE);          // <= KMG_NOALLOC_TILES by the check above
    u64 *bounds = ctx->match_scratch;                            // every word [0, ntiles] is written before it is read
    std::lock_guard<std::mutex> gs(g_match_scratch_mu);          // (queueing order = the order of the waits below)
    kmg_ctx *last = g_match_scratch_user[ctx->device];
    if (last && last != ctx && ctx->ev_scratch && last->ev_scratch)
        KMG_HIP(hipStreamWaitEvent(ctx->stream, last->ev_scratch, 0));
    hipLaunchKernelGGL(match_bounds_kernel, dim3(div_up_u64(ntiles + 1, 256)), dim3(256), 0, ctx->stream, search->keys, search->U,
                       ref->keys, ref->U, ref->bucket_off, ref->key_len, ntiles, bounds);
    hipLaunchKernelGGL(match_kernel, dim3(ntiles), dim3(MT_BLOCK), 0, ctx->stream, search->keys, search->U, ref->keys,
                       bounds, (long long *)d_hit_index, (u64 *)nullptr);
    KMG_HIP(hipGetLastError());
    if (ctx->ev_scratch) {
        KMG_HIP(hipEventRecord(ctx->ev_scratch, ctx->stream));
        g_match_scratch_user[ctx->device] = ctx;
    }
    return KMG_OK;
}
static_assert((u64)KMG_NOALLOC_TILES * MT_TILE == KMG_MATCH_NOALLOC_MAX_SEARCH, "header limit and scratch size disagree");

extern "C" int kmg_match(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int64_t *hit_index, uint64_t *n_hits) {
    if (!ctx || !search || !ref || !hit_index) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    PoolBuf d_hit(ctx), d_n(ctx);
    KMG_ALLOC(d_hit, (search->U + 1) * 8);
    KMG_ALLOC(d_n, 64);
    KMG_HIP(hipMemsetAsync(d_n.p, 0, 8, ctx->stream));
    KMG_TRY(match_locked(ctx, search, ref, d_hit.as<long long>(), d_n.as<u64>()));
    u64 nh = 0;
    if (search->U) KMG_HIP(hipMemcpyAsync(hit_index, d_hit.p, search->U * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipMemcpyAsync(&nh, d_n.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    if (n_hits) *n_hits = nh;
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ inexact match
static void near_log_release(kmg_ctx *ctx, kmg_db *db) {
    ctx->pool.release(db->near_log_s); ctx->pool.release(db->near_log_ref); ctx->pool.release(db->near_log_dist);
    db->near_log_s = db->near_log_ref = nullptr;
    db->near_log_dist = nullptr;
}

extern "C" int kmg_match_near(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, int max_diff, uint64_t *pair_off,
                              int64_t *ref_index, uint8_t *dist, uint64_t capacity, uint64_t *n_pairs) {
    if (!ctx || !search || !ref || !n_pairs) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    *n_pairs = 0;
    if (max_diff < 0) return fail(ctx, KMG_EINVAL, "max_diff must not be negative");
    // up to two differences: the neighbours are enumerated (near_kernel); more, and every two-word search: the trie walk
    // over the sorted array
    const bool wide = search->keys_lo != nullptr;      // (k-mers of the same length below: the other side is as wide)
    const bool walk = max_diff > 2 || wide;
    if (search->ctx->device != ctx->device || ref->ctx->device != ctx->device)
        return fail(ctx, KMG_EINVAL, "database belongs to another device");
    if (search->k != ref->k || search->key_len != ref->key_len)      // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (search->min_k != search->k || ref->min_k != ref->k)     // (the trie over several lengths shares data objects)
        return fail(ctx, KMG_EUNSUPPORTED, "inexact search over k-mers of several lengths is not supported on the GPU path");
    if ((ref_index == nullptr) != (dist == nullptr)) return fail(ctx, KMG_EINVAL, "ref_index and dist go together");
    const u64 U = search->U;
    if (U == 0) { if (pair_off) pair_off[0] = 0; return KMG_OK; }
    if (U * (u64)search->k >= 0xFFFFFFFFull * 256) return fail(ctx, KMG_EUNSUPPORTED, "too many search k-mers for one call");
    PoolBuf cnt(ctx), agg(ctx), tot(ctx), d_ref(ctx), d_dist(ctx);
    KMG_ALLOC(cnt, U * 4);
    const dim3 grid(div_up_u64(U * (u64)search->k, 256)), blk(256);
    kmg_db *sdb = const_cast<kmg_db *>(search);          // (the cached offsets are bookkeeping, not content)
    if (sdb->ctx != ctx) return fail(ctx, KMG_EINVAL, "the search database belongs to another context");
    u64 n = 0;
    if (sdb->near_off && sdb->near_ref_id == ref->id && sdb->near_diff == max_diff) n = sdb->near_pairs;
    else {
        if (!sdb->near_off) {
            void *p = nullptr;
            if (ctx->pool.alloc(&p, (U + 1) * 8) != KMG_OK) return fail(ctx, KMG_ENOMEM, "allocation failed");
            sdb->near_off = (u64 *)p;
        }
        sdb->near_diff = -1;
        near_log_release(ctx, sdb);
        KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
        KMG_ALLOC(tot, 64);
        KMG_HIP(hipMemsetAsync(cnt.p, 0, U * 4, ctx->stream));
        KMG_HIP(hipMemsetAsync(tot.p, 0, 64, ctx->stream));
        // The matches are noted down as they are found (two per search k-mer's worth of room: typical data holds about
        // one), so that the filling call places them instead of looking every neighbour up again; if the room does not
        // suffice, or cannot be had, that call falls back to the look-ups.
        NearLog log = {nullptr, nullptr, nullptr, 0, nullptr};
        if (!walk && ref->U < 0xFFFFFFF0ull && U < 0xFFFFFFF0ull) {
            const u64 cap = std::max<u64>(2 * U, 1024);
            void *a = nullptr, *b = nullptr, *c = nullptr;
            if (ctx->pool.alloc(&a, cap * 4) == KMG_OK && ctx->pool.alloc(&b, cap * 4) == KMG_OK && ctx->pool.alloc(&c, cap) == KMG_OK) {
                sdb->near_log_s = (u32 *)a; sdb->near_log_ref = (u32 *)b; sdb->near_log_dist = (uint8_t *)c;
                log = {sdb->near_log_s, sdb->near_log_ref, sdb->near_log_dist, cap, reinterpret_cast<unsigned long long *>(tot.as<u64>() + 1)};
            } else { ctx->pool.release(a); ctx->pool.release(b); ctx->pool.release(c); }
        }
        if (wide)
            hipLaunchKernelGGL((wide_near_walk_kernel<false>), dim3(div_up_u64(U, 128)), dim3(128), 0, ctx->stream, search->keys,
                               search->keys_lo, U, search->k, std::min(max_diff, search->k), ref->keys, ref->keys_lo, ref->U,
                               ref->bucket_off, ref->key_len, cnt.as<u32>(), (const u64 *)nullptr, (long long *)nullptr,
                               (uint8_t *)nullptr);
        else if (walk)
            hipLaunchKernelGGL((near_walk_kernel<false>), dim3(div_up_u64(U, 128)), dim3(128), 0, ctx->stream, search->keys, U, search->k,
                               std::min(max_diff, search->k), ref->keys, ref->U, ref->bucket_off, ref->key_len, cnt.as<u32>(),
                               (const u64 *)nullptr, (long long *)nullptr, (uint8_t *)nullptr);
        else
            hipLaunchKernelGGL((near_kernel<false>), grid, blk, 0, ctx->stream, search->keys, U, search->k, max_diff, ref->keys,
                               ref->bucket_off, ref->key_len, cnt.as<u32>(), (const u64 *)nullptr, (long long *)nullptr, (uint8_t *)nullptr, log);
        KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt.as<u32>(), sdb->near_off, U, OpSum(), 0ull, agg.as<u64>(),
                                                    tot.as<u64>(), true, KP_SCAN)));
        u64 t2[2] = {0, 0};
        KMG_HIP(read_back(ctx, t2, tot.p, 16));
        n = t2[0];
        if (log.n && t2[1] > log.cap) near_log_release(ctx, sdb);          // did not fit: the filling call looks up again
        sdb->near_ref_id = ref->id; sdb->near_diff = max_diff; sdb->near_pairs = n;
    }
    u64 *off_p = sdb->near_off;
    *n_pairs = n;
    if (pair_off) KMG_HIP(hipMemcpyAsync(pair_off, off_p, (U + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (ref_index && n) {
        if (capacity < n) return fail(ctx, KMG_EINVAL, "capacity below the number of matches");
        KMG_ALLOC(d_ref, n * 8);
        KMG_ALLOC(d_dist, n);
        KMG_HIP(hipMemsetAsync(cnt.p, 0, U * 4, ctx->stream));
        if (wide) {
            hipLaunchKernelGGL((wide_near_walk_kernel<true>), dim3(div_up_u64(U, 128)), dim3(128), 0, ctx->stream, search->keys,
                               search->keys_lo, U, search->k, std::min(max_diff, search->k), ref->keys, ref->keys_lo, ref->U,
                               ref->bucket_off, ref->key_len, cnt.as<u32>(), off_p, d_ref.as<long long>(), d_dist.as<uint8_t>());
        } else if (walk) {
            hipLaunchKernelGGL((near_walk_kernel<true>), dim3(div_up_u64(U, 128)), dim3(128), 0, ctx->stream, search->keys, U, search->k,
                               std::min(max_diff, search->k), ref->keys, ref->U, ref->bucket_off, ref->key_len, cnt.as<u32>(), off_p,
                               d_ref.as<long long>(), d_dist.as<uint8_t>());
        } else if (sdb->near_log_s) {
            const NearLog log = {sdb->near_log_s, sdb->near_log_ref, sdb->near_log_dist, n, nullptr};
            hipLaunchKernelGGL(near_place_kernel, dim3(div_up_u64(n, 256)), blk, 0, ctx->stream, log, n, cnt.as<u32>(), off_p,
                               d_ref.as<long long>(), d_dist.as<uint8_t>());
        } else {
            const NearLog none = {nullptr, nullptr, nullptr, 0, nullptr};
            hipLaunchKernelGGL((near_kernel<true>), grid, blk, 0, ctx->stream, search->keys, U, search->k, max_diff, ref->keys,
                               ref->bucket_off, ref->key_len, cnt.as<u32>(), off_p, d_ref.as<long long>(), d_dist.as<uint8_t>(), none);
        }
        near_log_release(ctx, sdb);                   // (stream-ordered: whoever gets the memory next runs after the kernel)
        if (!walk)                                    // (the walk hands its matches out in ascending order already)
            hipLaunchKernelGGL(near_sort_kernel, dim3(div_up_u64(U, 256)), blk, 0, ctx->stream, off_p, U, d_ref.as<long long>(),
                               d_dist.as<uint8_t>());
        KMG_HIP(hipGetLastError());
        KMG_HIP(hipMemcpyAsync(ref_index, d_ref.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        KMG_HIP(hipMemcpyAsync(dist, d_dist.p, n, hipMemcpyDeviceToHost, ctx->stream));
    }
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    KMG_HIP(hipGetLastError());
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ taxa counts
// ------------------------------------------------------------------------------------------------ per read
// the join both per-read entry points start from: hit index of every search k-mer, on the device
static int join_for_reads(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, PoolBuf &d_hit) {
    if (search->kind != KMG_PAYLOAD_READPOS || ref->kind != KMG_PAYLOAD_TAXON)
        return fail(ctx, KMG_EINVAL, "per-read aggregation joins a read database against a reference database");
    if (ref->U >= 0xFFFFFFF0ull || search->V >= 0xFFFFFFF0ull)
        return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 reference k-mers or read occurrences");
    // (its values are unpacked into the owner's pool, under the owner's lock: ensure_vals)
    if (search->ctx != ctx) return fail(ctx, KMG_EINVAL, "the search database belongs to another context");
    KMG_ALLOC(d_hit, (search->U + 1) * 8);
    KMG_TRY(match_locked(ctx, search, ref, d_hit.as<long long>(), nullptr));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(search)));
    return KMG_OK;
}

extern "C" int kmg_collect_by_read(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t *n_out,
                                   int32_t *read, int32_t *pos, int64_t *ref_index) {
    if (!ctx || !search || !ref || !n_out) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    *n_out = 0;
    PoolBuf d_hit(ctx);
    KMG_TRY(join_for_reads(ctx, search, ref, d_hit));
    const u64 U = search->U, V = search->V;
    if (U == 0 || V == 0) { KMG_HIP(hipStreamSynchronize(ctx->stream)); return KMG_OK; }

    PoolBuf cnt(ctx), off(ctx), agg(ctx), small(ctx);
    KMG_ALLOC(cnt, U * 4);
    KMG_ALLOC(off, (U + 1) * 8);
    KMG_ALLOC(agg, (size_t)(div_up_u64(U, SCAN_TILE) + 1) * 8);
    KMG_ALLOC(small, 64);                     // [0] = number of records (u64), [1] = largest read index (u32)
    KMG_HIP(hipMemsetAsync(small.p, 0, 64, ctx->stream));
    hipLaunchKernelGGL(cbr_count_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, search->val_off,
                       d_hit.as<long long>(), U, cnt.as<u32>());
    KMG_TRY((device_scan<u32, u64, OpSum, true>(ctx, cnt.as<u32>(), off.as<u64>(), U, OpSum(), 0ull, agg.as<u64>(),
                                                small.as<u64>() + 0, false, KP_SCAN)));
    hipLaunchKernelGGL(cbr_maxread_kernel, dim3(std::min<u32>(div_up_u64(V, 1024), 1024)), dim3(256), 0, ctx->stream,
                       search->val_a, V, reinterpret_cast<u32 *>(small.as<u64>() + 1));
    u64 hs[2] = {0, 0};
    KMG_HIP(read_back(ctx, hs, small.p, 16));
    const u64 n = hs[0];
    *n_out = n;
    if (n == 0 || !read) return KMG_OK;       // count only
    if (!pos || !ref_index) return fail(ctx, KMG_EINVAL, "read, pos and ref_index go together");

    int rbits = 1;
    while (rbits < 32 && (((u32)hs[1]) >> rbits) != 0) rbits++;
    const int nbits = 16 + rbits, shift = 64 - nbits;
    const int pbits = choose_pbits(n, nbits);
    const u32 ntiles = div_up_u64(n, RS_TILE);
    PoolBuf k0(ctx), v0(ctx), k1(ctx), v1(ctx), table(ctx), sagg(ctx), seg(ctx);
    KMG_ALLOC(k0, (n + 1) * 8); KMG_ALLOC(v0, (n + 1) * 4);
    KMG_ALLOC(k1, (n + 1) * 8); KMG_ALLOC(v1, (n + 1) * 4);
    KMG_ALLOC(table, (size_t)RS_DIGITS * ntiles * 4);
    KMG_ALLOC(sagg, (size_t)(div_up_u64((u64)RS_DIGITS * ntiles, SCAN_TILE) + 1) * 8);
    KMG_ALLOC(seg, (((size_t)1 << pbits) + 4) * 4);
    {
        // in: offsets + hit + output offset per k-mer, (read, pos) per occurrence; out: 12-byte records
        ProfScope ps(ctx, KP_COLLECT_EMIT, U * 24 + V * 8 + n * 12);
        hipLaunchKernelGGL(cbr_emit_kernel, dim3(div_up_u64(U * CB_LANES, 256)), dim3(256), 0, ctx->stream, search->val_off,
                           search->val_a, search->val_b, d_hit.as<long long>(), off.as<u64>(), U, shift, k0.as<u64>(), v0.as<u32>());
    }
    KMG_HIP(hipGetLastError());
    // ReadPos order = unsigned order of the used key bits; equal (read, pos) pairs (two sequences sharing a read
    // index) keep k-mer order -- the sort is stable
    bool in1 = false;
    KMG_TRY(hybrid_sort_records<u32>(ctx, k0.as<u64>(), v0.as<u32>(), k1.as<u64>(), v1.as<u32>(), (u32)n, shift, {},
                                     table.as<u32>(), sagg.as<u32>(), seg.as<u32>(), whole_space_plan(pbits), false, &in1, nullptr,
                                     /* counting passes only: the low key bits (pos) repeat in long runs */ false));
    const u64 *sk = in1 ? k1.as<u64>() : k0.as<u64>();
    const u32 *sv = in1 ? v1.as<u32>() : v0.as<u32>();
    // the other buffer pair is free now: unpack into it, then copy out
    int32_t *d_read = reinterpret_cast<int32_t *>(in1 ? k0.p : k1.p);
    int32_t *d_pos = d_read + n;
    PoolBuf d_ref(ctx);
    KMG_ALLOC(d_ref, n * 8);
    {
        ProfScope ps(ctx, KP_COLLECT_UNPACK, n * 12 + n * 16);
        hipLaunchKernelGGL(cbr_unpack_kernel, dim3(div_up_u64(n, 256)), dim3(256), 0, ctx->stream, sk, sv, n, shift, d_read, d_pos,
                           d_ref.as<long long>());
    }
    KMG_HIP(hipGetLastError());
    KMG_HIP(hipMemcpyAsync(read, d_read, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipMemcpyAsync(pos, d_pos, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipMemcpyAsync(ref_index, d_ref.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

// the per-read calls from a finished join: d_hit[i] = index of search k-mer i's answer among `ref`'s k-mers (its taxa
// are ref->val_a[ref->val_off[..]]), -1 = none
static int classify_from_hits(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, const long long *d_hit_index, uint64_t n_reads,
                              int32_t *call, uint8_t *has_hit);

static int classify_locked(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t n_reads, int32_t *call,
                           uint8_t *has_hit) {
    KMG_HIP(hipSetDevice(ctx->device));
    if (n_reads >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "too many reads");
    PoolBuf d_hit(ctx);
    KMG_TRY(join_for_reads(ctx, search, ref, d_hit));
    return classify_from_hits(ctx, search, ref, d_hit.as<long long>(), n_reads, call, has_hit);
}

static int classify_from_hits(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, const long long *d_hit_index, uint64_t n_reads,
                              int32_t *call, uint8_t *has_hit) {
    if (n_reads == 0) {
        if (search->V) return fail(ctx, KMG_EINVAL, "read index outside n_reads");
        return KMG_OK;
    }
    PoolBuf mm(ctx), bad(ctx), d_call(ctx), seen(ctx);
    if (has_hit) {
        KMG_ALLOC(seen, n_reads);
        KMG_HIP(hipMemsetAsync(seen.p, 0, n_reads, ctx->stream));
    }
    KMG_ALLOC(mm, n_reads * 8);
    KMG_ALLOC(bad, (n_reads + 16) * 4);               // [n_reads] = error flag
    KMG_ALLOC(d_call, n_reads * 4);
    const dim3 gr(div_up_u64(n_reads, 256)), b(256);
    hipLaunchKernelGGL(rc_init_kernel, gr, b, 0, ctx->stream, mm.as<int2>(), n_reads);
    KMG_HIP(hipMemsetAsync(bad.p, 0, (n_reads + 16) * 4, ctx->stream));
    u32 *err = bad.as<u32>() + n_reads;
    const u64 U = search->U;
    if (U) {
        const dim3 gk(div_up_u64(U * CB_LANES, 256));
        // two passes over the join: offsets + hit per search k-mer, one read index per occurrence (+ the hit k-mers'
        // taxa, cache-resident), per-read words updated in place
        ProfScope ps(ctx, KP_CLASSIFY, 2 * (U * 16 + search->V * 4) + n_reads * 16);
        hipLaunchKernelGGL(rc_single_kernel, gk, b, 0, ctx->stream, search->val_off, search->val_a, d_hit_index, U,
                           ref->val_off, ref->val_a, n_reads, mm.as<int2>(), err, seen.as<uint8_t>());
        hipLaunchKernelGGL(rc_check_kernel, gk, b, 0, ctx->stream, search->val_off, search->val_a, d_hit_index, U,
                           ref->val_off, ref->val_a, n_reads, mm.as<int2>(), bad.as<u32>());
    }
    hipLaunchKernelGGL(rc_call_kernel, gr, b, 0, ctx->stream, mm.as<int2>(), bad.as<u32>(), n_reads, d_call.as<int32_t>());
    KMG_HIP(hipGetLastError());
    u32 h_err = 0;
    KMG_HIP(hipMemcpyAsync(call, d_call.p, n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (has_hit) KMG_HIP(hipMemcpyAsync(has_hit, seen.p, n_reads, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(read_back(ctx, &h_err, err, 4));
    if (h_err) return fail(ctx, KMG_EINVAL, "read index outside n_reads");
    return KMG_OK;
}

extern "C" int kmg_classify_reads(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t n_reads, int32_t *call) {
    if (!ctx || !search || !ref || (n_reads && !call)) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    return classify_locked(ctx, search, ref, n_reads, call, nullptr);
}

extern "C" int kmg_classify_reads_hits(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref, uint64_t n_reads, int32_t *call,
                                       uint8_t *has_hit) {
    if (!ctx || !search || !ref || (n_reads && (!call || !has_hit))) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    return classify_locked(ctx, search, ref, n_reads, call, has_hit);
}

// The calls from answers found elsewhere: `rows` and hit_row as kmg_match_trie hands them out (or any database of taxon
// values with, per search k-mer, the index of its entry there or -1).
extern "C" int kmg_classify_reads_rows(kmg_ctx *ctx, const kmg_db *search, const kmg_db *rows, const int64_t *hit_row,
                                       uint64_t n_reads, int32_t *call, uint8_t *has_hit) {
    if (!ctx || !search || !rows || (search->U && !hit_row) || (n_reads && (!call || !has_hit))) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    if (n_reads >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "too many reads");
    if (search->kind != KMG_PAYLOAD_READPOS || rows->kind != KMG_PAYLOAD_TAXON)
        return fail(ctx, KMG_EINVAL, "per-read aggregation joins a read database against taxon values");
    if (search->ctx != ctx || rows->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
    if (search->V >= 0xFFFFFFF0ull) return fail(ctx, KMG_EUNSUPPORTED, "more than 2^32-16 read occurrences");
    const u64 U = search->U;
    PoolBuf d_hit(ctx), d_bad(ctx);
    KMG_ALLOC(d_hit, (U + 1) * 8);
    KMG_ALLOC(d_bad, 64);
    KMG_HIP(hipMemsetAsync(d_bad.p, 0, 64, ctx->stream));
    if (U) {
        KMG_HIP(hipMemcpyAsync(d_hit.p, hit_row, U * 8, hipMemcpyHostToDevice, ctx->stream));
        // every index must name one of the rows (the kernels below follow it)
        hipLaunchKernelGGL(hit_range_kernel, dim3(div_up_u64(U, 256)), dim3(256), 0, ctx->stream, d_hit.as<long long>(), U,
                           (long long)rows->U, d_bad.as<u32>());
        u32 h_bad = 0;
        KMG_HIP(read_back(ctx, &h_bad, d_bad.p, 4));      // (also: hit_row may be pageable -- it has been read by now)
        if (h_bad) return fail(ctx, KMG_EINVAL, "hit_row holds an index outside the rows");
    }
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(search)));
    KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(rows)));
    return classify_from_hits(ctx, search, rows, d_hit.as<long long>(), n_reads, call, has_hit);
}

extern "C" int kmg_taxacounts(kmg_ctx *ctx, const kmg_db *db, int32_t *taxon, int64_t *match, int64_t *unique,
                              uint64_t *n_taxa) {
    if (!ctx || !db || !n_taxa) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (db->kind != KMG_PAYLOAD_TAXON) return fail(ctx, KMG_EINVAL, "TaxaCounts needs a reference (taxon) database");
    KMG_HIP(hipSetDevice(ctx->device));
    const u64 cap = *n_taxa;
    *n_taxa = 0;
    const u64 V = db->V;
    if (V == 0 || db->U == 0) return KMG_OK;
    ProfScope ps(ctx, KP_TAXA, 8 * (db->U + 1) + 4 * V);

    // distinct taxa from a presence bitmap over [min, max] (match.h), then one counting pass over the CSR
    PoolBuf mm(ctx), bits(ctx), cnt(ctx), prefix(ctx), agg(ctx), total(ctx), taxa(ctx), cm(ctx), cu(ctx);
    KMG_ALLOC(mm, 64);
    KMG_ALLOC(total, 64);
    const u32 vgrid = std::min<u32>(div_up_u64(V, 1024), 4096);
    int32_t h_mm[2] = {db->tax_lo, db->tax_hi};
    if (!db->tax_bounds) {                      // (a database built here knows the span of its taxa from the sequence ids)
        hipLaunchKernelGGL(taxa_minmax_init_kernel, dim3(1), dim3(64), 0, ctx->stream, mm.as<int32_t>());
        hipLaunchKernelGGL(taxa_minmax_kernel, dim3(vgrid), dim3(256), 0, ctx->stream, db->val_a, V, mm.as<int32_t>());
        KMG_HIP(read_back(ctx, h_mm, mm.p, 8));
    }
    const int32_t tmin = h_mm[0];
    const u64 range = (u64)((long long)h_mm[1] - (long long)tmin) + 1;      // <= 2^32
    const u64 nwords = (range + 31) / 32;
    const bool dense = range <= TX_LDS_BIG;     // rows addressed by taxon - min; presence read off the counts afterwards
    PoolBuf dm(ctx), du(ctx);
    KMG_ALLOC(bits, nwords * 4);
    KMG_ALLOC(cnt, nwords * 4);
    KMG_ALLOC(prefix, (nwords + 1) * 4);
    KMG_ALLOC(agg, (size_t)(div_up_u64(nwords, SCAN_TILE) + 1) * 8);
#define TAXA_COUNT(MODE, DIRECT, GRID, BLOCK, ROWS_, M, Q)                                                                  \
    hipLaunchKernelGGL((taxa_count_kernel<MODE, DIRECT>), dim3(GRID), dim3(BLOCK), 0, ctx->stream, db->val_off, db->val_a, db->U, \
                       V, bits.as<u32>(), prefix.as<u32>(), tmin, (u32)(ROWS_), M.as<u64>(), Q.as<u64>())
    if (dense) {
        KMG_ALLOC(dm, range * 8);
        KMG_ALLOC(du, range * 8);
        KMG_HIP(hipMemsetAsync(dm.p, 0, range * 8, ctx->stream));
        KMG_HIP(hipMemsetAsync(du.p, 0, range * 8, ctx->stream));
        if (range <= TX_LDS_SMALL) TAXA_COUNT(0, true, std::min<u32>(div_up_u64(db->U, 256 * TX_UNROLL), 2048), 256, range, dm, du);
        else TAXA_COUNT(1, true, std::min<u32>(div_up_u64(db->U, 1024 * TX_UNROLL), 256), 1024, range, dm, du);
        hipLaunchKernelGGL(taxa_rows_present_kernel, dim3(div_up_u64(nwords, 256)), dim3(256), 0, ctx->stream, dm.as<u64>(), (u32)range,
                           bits.as<u32>());
    } else {
        KMG_HIP(hipMemsetAsync(bits.p, 0, nwords * 4, ctx->stream));
        hipLaunchKernelGGL(taxa_present_kernel, dim3(vgrid), dim3(256), 0, ctx->stream, db->val_a, V, tmin, bits.as<u32>());
    }
    hipLaunchKernelGGL(taxa_popc_kernel, dim3(div_up_u64(nwords, 256)), dim3(256), 0, ctx->stream, bits.as<u32>(), nwords, cnt.as<u32>());
    KMG_TRY((device_scan<u32, u32, OpSum, true>(ctx, cnt.as<u32>(), prefix.as<u32>(), nwords, OpSum(), 0u, agg.as<u32>(),
                                                total.as<u32>(), false, KP_SCAN)));
    u32 nt = 0;
    KMG_HIP(read_back(ctx, &nt, total.p, 4));
    KMG_ALLOC(taxa, (size_t)nt * 4);
    KMG_ALLOC(cm, (size_t)nt * 8);
    KMG_ALLOC(cu, (size_t)nt * 8);
    hipLaunchKernelGGL(taxa_list_kernel, dim3(div_up_u64(nwords, 256)), dim3(256), 0, ctx->stream, bits.as<u32>(), prefix.as<u32>(),
                       nwords, tmin, taxa.as<int32_t>());
    if (dense) {
        hipLaunchKernelGGL(taxa_rows_gather_kernel, dim3(div_up_u64(range, 256)), dim3(256), 0, ctx->stream, dm.as<u64>(), du.as<u64>(),
                           (u32)range, bits.as<u32>(), prefix.as<u32>(), cm.as<u64>(), cu.as<u64>());
    } else {
        KMG_HIP(hipMemsetAsync(cm.p, 0, (size_t)nt * 8, ctx->stream));
        KMG_HIP(hipMemsetAsync(cu.p, 0, (size_t)nt * 8, ctx->stream));
        if (nt <= TX_LDS_SMALL) TAXA_COUNT(0, false, std::min<u32>(div_up_u64(db->U, 256 * TX_UNROLL), 2048), 256, nt, cm, cu);
        else if (nt <= TX_LDS_BIG) TAXA_COUNT(1, false, std::min<u32>(div_up_u64(db->U, 1024 * TX_UNROLL), 256), 1024, nt, cm, cu);
        else TAXA_COUNT(2, false, std::min<u32>(div_up_u64(db->U, 256 * TX_UNROLL), 4096), 256, nt, cm, cu);
    }
#undef TAXA_COUNT
    KMG_HIP(hipGetLastError());
    const u64 w = std::min<u64>(nt, cap);
    if (w) {
        if (taxon) KMG_HIP(hipMemcpyAsync(taxon, taxa.p, w * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (match) KMG_HIP(hipMemcpyAsync(match, cm.p, w * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (unique) KMG_HIP(hipMemcpyAsync(unique, cu.p, w * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    *n_taxa = nt;
    return KMG_OK;
}

// ------------------------------------------------------------------------------------------------ profiling
extern "C" int kmg_prof_enable(kmg_ctx *ctx, int on) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->prof_on = on != 0;
    return KMG_OK;
}
extern "C" int kmg_prof_reset(kmg_ctx *ctx) {
    if (!ctx) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_drain(ctx);
    for (int i = 0; i < KP_COUNT_; i++) ctx->prof[i] = ProfSlot();
    return KMG_OK;
}
extern "C" int kmg_prof_get(kmg_ctx *ctx, int id, double *total_ms, uint64_t *launches, uint64_t *algo_bytes) {
    if (!ctx || id < 0 || id >= KP_COUNT_) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof[id].ms;
    if (launches) *launches = ctx->prof[id].launches;
    if (algo_bytes) *algo_bytes = ctx->prof[id].bytes;
    return KMG_OK;
}
// 8-byte-per-lane streaming copy of n words: a launch whose HBM bytes are known exactly (8n read, 8n
// written), used to calibrate the FETCH_SIZE / WRITE_SIZE counters for this library's access width.
__global__ __launch_bounds__(256) void calib_copy_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}
extern "C" int kmg_calib_copy(kmg_ctx *ctx, uint64_t n_words) {
    if (!ctx || n_words == 0) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    PoolBuf a(ctx), b(ctx);
    KMG_ALLOC(a, n_words * 8);
    KMG_ALLOC(b, n_words * 8);
    KMG_HIP(hipMemsetAsync(a.p, 0x5A, n_words * 8, ctx->stream));
    hipLaunchKernelGGL(calib_copy_kernel, dim3(div_up_u64(n_words, 256 * 8)), dim3(256), 0, ctx->stream, a.as<u64>(),
                       b.as<u64>(), (u64)n_words);
    KMG_HIP(hipGetLastError());
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

extern "C" const char *kmg_prof_name(int id) { return (id >= 0 && id < KP_COUNT_) ? kp_names[id] : ""; }
extern "C" int kmg_prof_count(void) { return KP_COUNT_; }

// ------------------------------------------------------------------------------------------------ multi-GPU
#include "comm.h"
