// comm.h -- the multi-GPU exchange step under the C ABI: one process per GPU, RCCL over xGMI.
//
// The reference is a single process; what it offers for a database in several pieces is the file set name.1 ..
// name.N (Matcher.java:80-89) whose k-mers DB.kmers walks in ascending order (DB.java:99-109).  Here piece g is
// the contiguous range of key buckets (Kmer.key(keyLength), Kmer.java:47-55) that GPU g owns, and the two exchange
// steps are grouped ncclSend / ncclRecv (an all-to-all-v: every GPU talks to every other one over its own xGMI
// link at once) on the context's stream:
//   kmg_match_sharded            read k-mers -> owners, join there in one launch, hit indices back
//   kmg_db_exchange_by_key_range database slices -> owners, merged there (counts add, read sets unite)
// RCCL is looked up at run time (dlopen), so the library loads -- and everything single-GPU works -- where no RCCL is
// installed, and a process that already holds one (PyTorch's) shares it.
#pragma once
#include <dlfcn.h>

#include "common.h"

typedef struct { char internal[128]; } kmg_nccl_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *kmg_nccl_comm;
enum { KMG_NCCL_INT64 = 4, KMG_NCCL_UINT64 = 5, KMG_NCCL_INT32 = 2, KMG_NCCL_UINT32 = 3 };   // ncclDataType_t values

struct RcclApi {
    int (*GetUniqueId)(kmg_nccl_id *);
    int (*CommInitRank)(kmg_nccl_comm *, int, kmg_nccl_id, int);
    int (*CommDestroy)(kmg_nccl_comm);
    int (*Send)(const void *, size_t, int, int, kmg_nccl_comm, hipStream_t);
    int (*Recv)(void *, size_t, int, int, kmg_nccl_comm, hipStream_t);
    int (*AllGather)(const void *, void *, size_t, int, kmg_nccl_comm, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char *(*GetErrorString)(int);
    bool ok = false;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // one the process already holds (PyTorch's)
            if (h) break;
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) return;
#define KMG_SYM(field, sym) *(void **)(&api.field) = dlsym(h, sym); if (!api.field) return
        KMG_SYM(GetUniqueId, "ncclGetUniqueId");
        KMG_SYM(CommInitRank, "ncclCommInitRank");
        KMG_SYM(CommDestroy, "ncclCommDestroy");
        KMG_SYM(Send, "ncclSend");
        KMG_SYM(Recv, "ncclRecv");
        KMG_SYM(AllGather, "ncclAllGather");
        KMG_SYM(GroupStart, "ncclGroupStart");
        KMG_SYM(GroupEnd, "ncclGroupEnd");
        KMG_SYM(GetErrorString, "ncclGetErrorString");
#undef KMG_SYM
        api.ok = true;
    });
    return api.ok ? &api : nullptr;
}

#define COMM_MAX_COLS 3     // count columns of an exchange (k-mers, values) + the status column
struct kmg_comm {
    kmg_ctx *ctx = nullptr;
    int nranks = 0, rank = 0;
    kmg_nccl_comm comm = nullptr;
    // device words of the count exchange (mine: nranks x COMM_MAX_COLS, all: nranks times that), allocated when the
    // communicator is made: nothing that can fail locally stands between a call's checks and its first collective
    u64 *d_mine = nullptr, *d_all = nullptr;
};

#define KMG_NCCL(expr)                                                                        \
    do {                                                                                      \
        int _e = (expr);                                                                      \
        if (_e != 0) {                                                                        \
            ctx->err = std::string(#expr) + ": " + rccl_api()->GetErrorString(_e);            \
            return KMG_EDEVICE;                                                               \
        }                                                                                     \
    } while (0)

extern "C" int kmg_comm_unique_id(uint8_t *id) {
    if (!id) return KMG_EINVAL;
    RcclApi *api = rccl_api();
    if (!api) return KMG_EUNSUPPORTED;
    kmg_nccl_id u;
    if (api->GetUniqueId(&u) != 0) return KMG_EDEVICE;
    memcpy(id, u.internal, sizeof u.internal);
    return KMG_OK;
}

extern "C" int kmg_comm_create(kmg_ctx *ctx, int nranks, int rank, const uint8_t *id, kmg_comm **out) {
    if (!ctx || !out || !id) return KMG_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, KMG_EINVAL, "rank must lie in 0 .. nranks - 1");
    RcclApi *api = rccl_api();
    if (!api) return fail(ctx, KMG_EUNSUPPORTED, "RCCL (librccl.so) was not found on this machine");
    KMG_HIP(hipSetDevice(ctx->device));
    kmg_comm *c = new (std::nothrow) kmg_comm();
    if (!c) return fail(ctx, KMG_ENOMEM, "host allocation failed");
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    kmg_nccl_id u;
    memcpy(u.internal, id, sizeof u.internal);
    const size_t per = (size_t)nranks * COMM_MAX_COLS;
    void *pm = nullptr, *pa = nullptr;
    if (ctx->pool.alloc(&pm, per * 8) != KMG_OK || ctx->pool.alloc(&pa, per * nranks * 8) != KMG_OK) {
        ctx->pool.release(pm);
        delete c;
        return fail(ctx, KMG_ENOMEM, "device allocation failed");
    }
    c->d_mine = (u64 *)pm; c->d_all = (u64 *)pa;
    const int e = api->CommInitRank(&c->comm, nranks, u, rank);
    if (e != 0) {
        ctx->err = std::string("ncclCommInitRank: ") + api->GetErrorString(e);
        ctx->pool.release(pm); ctx->pool.release(pa);
        delete c;
        return KMG_EDEVICE;
    }
    *out = c;
    return KMG_OK;
}

extern "C" void kmg_comm_destroy(kmg_comm *c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> g(c->ctx->mu);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
        c->ctx->pool.release(c->d_mine);
        c->ctx->pool.release(c->d_all);
    }
    delete c;
}

extern "C" int kmg_comm_info(const kmg_comm *c, int *nranks, int *rank) {
    if (!c) return KMG_EINVAL;
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return KMG_OK;
}

// cut[g] = index of the first entry of `db` in key bucket bounds[g] (g = 0 .. nranks): bounds are ascending bucket
// numbers, GPU g owns buckets [bounds[g], bounds[g + 1]).  A database is sorted, so each owner's part is a slice.
static int comm_cuts(kmg_ctx *ctx, const kmg_db *db, const uint32_t *bounds, int nranks, std::vector<u64> &cut) {
    const u32 nbuckets = 1u << (2 * db->key_len);
    cut.assign(nranks + 1, 0);
    for (int g = 0; g <= nranks; g++) {
        if (bounds[g] > nbuckets || (g && bounds[g] < bounds[g - 1])) return fail(ctx, KMG_EINVAL, "bounds must ascend within 0 .. 4^key_len");
        KMG_HIP(hipMemcpyAsync(&cut[g], db->bucket_off + bounds[g], 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    return KMG_OK;
}

// Every rank's `ncols` send counts to every rank: m[(src * nranks + dst) * ncols + col] on the host of every rank.
// The same all-gather carries every rank's verdict on its own checks (`my_status`, 0 = fine): a rank whose arguments or
// local preparation failed still takes part -- with zero counts -- and EVERY rank leaves the call with an error, instead
// of one rank returning early while its peers wait in the collective.
static int comm_counts(kmg_comm *c, const std::vector<u64> &mine, int ncols, int my_status, std::vector<u64> &m) {
    kmg_ctx *ctx = c->ctx;
    const int n = c->nranks;
    if (n == 1) { m = mine; return my_status; }
    const int cols = ncols + 1;                       // + status
    const size_t per = (size_t)n * cols;
    std::vector<u64> row(per, 0), all(per * n, 0);
    for (int r = 0; r < n; r++) {
        if (my_status == KMG_OK)
            for (int col = 0; col < ncols; col++) row[(size_t)r * cols + col] = mine[(size_t)r * ncols + col];
        row[(size_t)r * cols + ncols] = (u64)(uint32_t)(-my_status);
    }
    const std::string own_err = ctx->err;
    // (a failure of these two copies is a device failure: the peers cannot be told)
    KMG_HIP(hipMemcpyAsync(c->d_mine, row.data(), per * 8, hipMemcpyHostToDevice, ctx->stream));
    KMG_NCCL(rccl_api()->AllGather(c->d_mine, c->d_all, per, KMG_NCCL_UINT64, c->comm, ctx->stream));
    KMG_HIP(hipMemcpyAsync(all.data(), c->d_all, per * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KMG_HIP(hipStreamSynchronize(ctx->stream));
    m.assign((size_t)n * n * ncols, 0);
    int first_bad = -1;
    for (int s = 0; s < n; s++)
        for (int r = 0; r < n; r++) {
            for (int col = 0; col < ncols; col++) m[((size_t)s * n + r) * ncols + col] = all[((size_t)s * n + r) * cols + col];
            if (all[((size_t)s * n + r) * cols + ncols] != 0 && first_bad < 0) first_bad = s;
        }
    if (my_status != KMG_OK) { ctx->err = own_err; return my_status; }
    if (first_bad >= 0) {
        ctx->err = "rank " + std::to_string(first_bad) + " failed its checks before the exchange: nothing was exchanged";
        return KMG_ESTATE;
    }
    return KMG_OK;
}

// one all-to-all-v: slice g of `send` (elements soff[g] .. soff[g + 1]) to rank g, rank s's slice to recv + roff[s]
static int comm_alltoallv(kmg_comm *c, const void *send, const std::vector<u64> &soff, void *recv, const std::vector<u64> &roff,
                          size_t elem, int dtype) {
    kmg_ctx *ctx = c->ctx;
    RcclApi *api = rccl_api();
    // this rank's own slice never leaves the GPU: a device copy (HBM speed) instead of a send to self
    {
        const int g = c->rank;
        const u64 ns = soff[g + 1] - soff[g];
        if (ns != roff[g + 1] - roff[g]) return fail(ctx, KMG_ESTATE, "all-to-all: own slice counted differently on its two sides");
        if (ns) KMG_HIP(hipMemcpyAsync((char *)recv + roff[g] * elem, (const char *)send + soff[g] * elem, ns * elem, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (c->nranks == 1) return KMG_OK;
    KMG_NCCL(api->GroupStart());
    // (an error between GroupStart and GroupEnd must not leave the communicator inside an open group)
    int bad = 0;
    const char *what = "ncclSend";
    for (int g = 0; g < c->nranks && !bad; g++) {
        if (g == c->rank) continue;
        const u64 ns = soff[g + 1] - soff[g], nr = roff[g + 1] - roff[g];
        if (ns) bad = api->Send((const char *)send + soff[g] * elem, ns, dtype, g, c->comm, ctx->stream);
        if (nr && !bad) { bad = api->Recv((char *)recv + roff[g] * elem, nr, dtype, g, c->comm, ctx->stream); if (bad) what = "ncclRecv"; }
    }
    std::string msg;
    if (bad) msg = std::string(what) + ": " + api->GetErrorString(bad);
    const int end = api->GroupEnd();
    if (bad) { ctx->err = msg; return KMG_EDEVICE; }
    if (end != 0) { ctx->err = std::string("ncclGroupEnd: ") + api->GetErrorString(end); return KMG_EDEVICE; }
    return KMG_OK;
}

// the checks of kmg_match_sharded that need no other rank; the cut points come out with them
static int match_sharded_checks(kmg_ctx *ctx, const kmg_db *search, const kmg_db *ref_shard, const uint32_t *bounds, int n,
                                const int64_t *d_hit_index, std::vector<u64> &cut) {
    if (search->ctx->device != ctx->device || ref_shard->ctx->device != ctx->device)
        return fail(ctx, KMG_EINVAL, "database belongs to another device");
    if (search->k != ref_shard->k || search->key_len != ref_shard->key_len)                          // DB.java:78-81
        return fail(ctx, KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database");
    if (search->min_k != search->k || ref_shard->min_k != ref_shard->k)
        return fail(ctx, KMG_EUNSUPPORTED, "k-mers of several lengths: restrict the databases to one length first");
    if (search->U && !d_hit_index) return fail(ctx, KMG_EINVAL, "null hit_index");
    return comm_cuts(ctx, search, bounds, n, cut);
}

extern "C" int kmg_match_sharded(kmg_comm *c, const kmg_db *search, const kmg_db *ref_shard, const uint32_t *bounds,
                                 int64_t *d_hit_index) {
    if (!c || !search || !ref_shard || !bounds) return KMG_EINVAL;
    kmg_ctx *ctx = c->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    KMG_HIP(hipSetDevice(ctx->device));
    const int n = c->nranks;
    std::vector<u64> cut(n + 1, 0), mine(n, 0), all;
    // a rank whose checks fail still meets its peers in the count exchange, and every rank returns an error
    const int st = match_sharded_checks(ctx, search, ref_shard, bounds, n, d_hit_index, cut);
    if (st == KMG_OK) for (int r = 0; r < n; r++) mine[r] = cut[r + 1] - cut[r];
    KMG_TRY(comm_counts(c, mine, 1, st, all));
    std::vector<u64> roff(n + 1, 0);
    for (int s = 0; s < n; s++) roff[s + 1] = roff[s] + all[(size_t)s * n + c->rank];
    const u64 nrecv = roff[n];
    // (from here on a local failure -- out of device memory -- leaves the peers waiting in the grouped exchange: the
    // receive buffers can only be sized once the counts are known)
    PoolBuf rkeys(ctx), rlo(ctx), rhits(ctx);
    const bool wide = search->keys_lo != nullptr;          // two-word k-mers (wide.h): the second words travel as well
    KMG_ALLOC(rkeys, (nrecv + 1) * 8);
    KMG_ALLOC(rhits, (nrecv + 1) * 8);
    if (wide) KMG_ALLOC(rlo, (nrecv + 1) * 8);
    // k-mers outside every range have no owner: no match
    if (search->U) KMG_HIP(hipMemsetAsync(d_hit_index, 0xFF, search->U * 8, ctx->stream));
    KMG_TRY(comm_alltoallv(c, search->keys, cut, rkeys.p, roff, 8, KMG_NCCL_UINT64));
    if (wide) KMG_TRY(comm_alltoallv(c, search->keys_lo, cut, rlo.p, roff, 8, KMG_NCCL_UINT64));
    // what this owner received: one ascending run per source GPU, all inside its range -> one launch
    if (nrecv && wide)
        KMG_TRY(match_keys_wide_locked(ctx, rkeys.as<u64>(), rlo.as<u64>(), (const uint64_t *)roff.data(), n, search->k, ref_shard,
                                       rhits.as<int64_t>()));
    else if (nrecv) {
        if (n <= MT_MAX_RUNS) KMG_TRY(match_keys_locked(ctx, rkeys.as<uint64_t>(), (const uint64_t *)roff.data(), n, search->k, ref_shard, rhits.as<int64_t>()));
        else
            for (int s = 0; s < n; s += MT_MAX_RUNS) {
                const int m = std::min(MT_MAX_RUNS, n - s);
                KMG_TRY(match_keys_locked(ctx, rkeys.as<uint64_t>(), (const uint64_t *)roff.data() + s, m, search->k, ref_shard, rhits.as<int64_t>()));
            }
    }
    KMG_TRY(comm_alltoallv(c, rhits.p, roff, d_hit_index, cut, 8, KMG_NCCL_INT64));
    KMG_TRY(note_foreign_use(ctx, search));
    return KMG_OK;
}

// counts of a slice of CSR offsets: n[i] = val_off[first + i + 1] - val_off[first + i]
__global__ void comm_counts_kernel(const u64 *__restrict__ val_off, u64 first, u64 n, u32 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u32)(val_off[first + i + 1] - val_off[first + i]);
}

extern "C" int kmg_db_exchange_by_key_range(kmg_comm *c, const kmg_db *local, const uint32_t *bounds, kmg_db **out) {
    if (!c || !local || !bounds || !out) return KMG_EINVAL;
    kmg_ctx *ctx = c->ctx;
    *out = nullptr;
    const int n = c->nranks;
    std::vector<kmg_db *> parts;
    auto drop_parts = [&]() { for (kmg_db *d : parts) db_release(d); parts.clear(); };
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        KMG_HIP(hipSetDevice(ctx->device));
        std::vector<u64> cut(n + 1, 0), vcut(n + 1, 0), mine(2 * (size_t)n, 0), all;
        // the checks and preparation that need no other rank; their verdict travels with the counts (comm_counts)
        const int st = [&]() -> int {
            if (local->ctx != ctx) return fail(ctx, KMG_EINVAL, "database belongs to another context");
            if (local->min_k != local->k) return fail(ctx, KMG_EUNSUPPORTED, "databases of k-mers of several lengths are not exchanged");
            KMG_TRY(ensure_vals(ctx, const_cast<kmg_db *>(local)));
            KMG_TRY(comm_cuts(ctx, local, bounds, n, cut));
            for (int r = 0; r <= n; r++) KMG_HIP(hipMemcpyAsync(&vcut[r], local->val_off + cut[r], 8, hipMemcpyDeviceToHost, ctx->stream));
            KMG_HIP(hipStreamSynchronize(ctx->stream));
            return KMG_OK;
        }();
        if (st == KMG_OK) for (int r = 0; r < n; r++) { mine[2 * r] = cut[r + 1] - cut[r]; mine[2 * r + 1] = vcut[r + 1] - vcut[r]; }
        KMG_TRY(comm_counts(c, mine, 2, st, all));
        std::vector<u64> koff(n + 1, 0), voff(n + 1, 0);
        for (int s = 0; s < n; s++) {
            koff[s + 1] = koff[s] + all[((size_t)s * n + c->rank) * 2];
            voff[s + 1] = voff[s] + all[((size_t)s * n + c->rank) * 2 + 1];
        }
        const u64 nk = koff[n], nv = voff[n], sk = cut[n] - cut[0];
        PoolBuf scnt(ctx), rkeys(ctx), rlo(ctx), rcnt(ctx), ra(ctx), rb(ctx), off(ctx), agg(ctx);
        const bool wide = local->keys_lo != nullptr;       // two-word k-mers (wide.h): a fifth all-to-all-v, the second words
        KMG_ALLOC(scnt, (sk + 1) * 4);
        KMG_ALLOC(rkeys, (nk + 1) * 8);
        if (wide) KMG_ALLOC(rlo, (nk + 1) * 8);
        KMG_ALLOC(rcnt, (nk + 1) * 4);
        KMG_ALLOC(ra, (nv + 1) * 4);
        KMG_ALLOC(rb, (nv + 1) * 4);
        KMG_ALLOC(off, (nk + 2) * 8);
        KMG_ALLOC(agg, (size_t)(div_up_u64(nk, SCAN_TILE) + 1) * 8);
        if (sk) hipLaunchKernelGGL(comm_counts_kernel, dim3(div_up_u64(sk, 256)), dim3(256), 0, ctx->stream, local->val_off, cut[0], sk, scnt.as<u32>());
        // four all-to-all-v, whatever the number of GPUs: k-mers, values per k-mer, the two value arrays
        std::vector<u64> crel(n + 1), vrel(n + 1);
        for (int r = 0; r <= n; r++) { crel[r] = cut[r] - cut[0]; vrel[r] = vcut[r] - vcut[0]; }
        KMG_TRY(comm_alltoallv(c, local->keys + cut[0], crel, rkeys.p, koff, 8, KMG_NCCL_UINT64));
        if (wide) KMG_TRY(comm_alltoallv(c, local->keys_lo + cut[0], crel, rlo.p, koff, 8, KMG_NCCL_UINT64));
        KMG_TRY(comm_alltoallv(c, scnt.p, crel, rcnt.p, koff, 4, KMG_NCCL_UINT32));
        KMG_TRY(comm_alltoallv(c, local->val_a + vcut[0], vrel, ra.p, voff, 4, KMG_NCCL_INT32));
        KMG_TRY(comm_alltoallv(c, local->val_b + vcut[0], vrel, rb.p, voff, 4, KMG_NCCL_INT32));
        // every source's part becomes a database of its own (offsets rebuilt from the counts), then they merge
        for (int s = 0; s < n; s++) {
            const u64 pk = koff[s + 1] - koff[s], pv = voff[s + 1] - voff[s];
            if (pk == 0) continue;
            int r = device_scan<u32, u64, OpSum, true>(ctx, rcnt.as<u32>() + koff[s], off.as<u64>(), pk, OpSum(), 0ull, agg.as<u64>(), nullptr,
                                                       true, KP_SCAN);
            kmg_db *part = nullptr;
            if (r == KMG_OK)
                r = db_load_common(ctx, local->k, local->key_len, local->rc, local->kind, rkeys.as<u64>() + koff[s], pk, off.as<u64>(),
                                   ra.as<int32_t>() + voff[s], rb.as<int32_t>() + voff[s], pv, hipMemcpyDeviceToDevice, &part,
                                   wide ? rlo.as<u64>() + koff[s] : nullptr);
            if (r != KMG_OK) { drop_parts(); return r; }
            parts.push_back(part);
        }
        if (parts.empty()) {
            // nothing arrived: an empty database of the same shape
            const u64 zero = 0;
            kmg_db *part = nullptr;
            const int r = db_load_common(ctx, local->k, local->key_len, local->rc, local->kind, nullptr, 0, &zero, nullptr, nullptr, 0,
                                         hipMemcpyHostToDevice, &part, wide ? &zero : nullptr);
            if (r != KMG_OK) return r;
            *out = part;
            return KMG_OK;
        }
        if (parts.size() == 1) { *out = parts[0]; return KMG_OK; }
    }
    const int r = kmg_db_merge(ctx, parts.data(), (int)parts.size(), out);     // (takes the lock itself)
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        drop_parts();
    }
    return r;
}
