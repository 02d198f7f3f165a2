// common.h -- context, device-memory pool, launch timing and error plumbing shared by the stages.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmers_gpu.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define KMG_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return _e == hipErrorOutOfMemory ? KMG_ENOMEM : KMG_EDEVICE;                   \
        }                                                                                  \
    } while (0)

#define KMG_TRY(expr)                    \
    do {                                 \
        int _r = (expr);                 \
        if (_r != KMG_OK) return _r;     \
    } while (0)

// Kernel ids for the measurement hook (kmg_prof_get).
enum {
    KP_EXTRACT_TILESEQ = 0,
    KP_EXTRACT_COUNT,
    KP_EXTRACT_WRITE,
    KP_NUMBER_READS,
    KP_SCAN,
    KP_SORT_HIST_V32,
    KP_SORT_HIST_V64,
    KP_SORT_SCAN,
    KP_SORT_SCATTER_V32,
    KP_SORT_SCATTER_V64,
    KP_SORT_SEGMENTS,
    KP_SORT_LOCAL_V32,
    KP_SORT_LOCAL_V64,
    KP_REDUCE_FLAG,
    KP_REDUCE_WRITE,
    KP_BUCKETS,
    KP_MATCH,
    KP_TAXA,
    KP_ENCODE,
    KP_COLLECT_EMIT,
    KP_COLLECT_UNPACK,
    KP_CLASSIFY,
    KP_COUNT_
};

static const char *const kp_names[KP_COUNT_] = {
    "extract_tileseq", "extract_count", "extract_write", "number_reads", "scan",
    "sort_hist_ref",   "sort_hist_reads", "sort_scan", "sort_scatter_ref", "sort_scatter_reads",
    "sort_segments",   "sort_local_ref", "sort_local_reads", "reduce_flag",  "reduce_write",
    "bucket_offsets",  "match",         "taxacounts",    "encode",
    "collect_emit",    "collect_unpack", "classify_reads"};

// Grow-only caching device allocator: blocks freed to the pool are reused by later requests of a
// similar size, so a steady-state loop (bench steps, repeated builds) calls hipMalloc only once.
struct DevPool {
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> live;
    size_t total = 0;

    int alloc(void **out, size_t bytes) {
        if (bytes == 0) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
            *out = it->second;
            live[it->second] = it->first;
            free_blocks.erase(it);
            return KMG_OK;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            // give everything cached back and retry once.  The failed call leaves hipErrorOutOfMemory as the
            // runtime's last error: taken off here, or the next hipGetLastError() behind a kernel launch would
            // report an allocation failure that was recovered from
            (void)hipGetLastError();
            trim();
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return KMG_ENOMEM;
            }
        }
        total += bytes;
        live[p] = bytes;
        *out = p;
        return KMG_OK;
    }
    void release(void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.insert({it->second, p});
        live.erase(it);
    }
    void trim() {
        for (auto &kv : free_blocks) {
            (void)hipFree(kv.second);
            total -= kv.first;
        }
        free_blocks.clear();
    }
    void destroy() {
        trim();
        for (auto &kv : live) (void)hipFree(kv.first);
        live.clear();
    }
};

struct ProfSlot {
    double ms = 0;
    u64 launches = 0;
    u64 bytes = 0;
};

struct PendingEvent {
    int id;
    hipEvent_t a, b;
};

// Device-resident build state (one 64-bit word each), so that kmg_build_add_dev never has to
// synchronise with the host.
enum {
    ST_NREC = 0,      // records written so far (fwd + rc)
    ST_NWIN,          // valid forward windows so far
    ST_READ_BASE,     // reads numbered so far (ReadIDMapping.curReadI + 1)
    ST_CARRY_TOK,     // token of the last read that produced a k-mer (ReadIDMapping.curReadS)
    ST_HAVE_CARRY,    // 0 until some read has produced a k-mer
    ST_ID_OR,         // OR / AND of every (taxon ^ 0x80000000): constant value bits skip sort passes
    ST_ID_AND,
    ST_ID_MIN,        // smallest / largest (taxon ^ 0x80000000) of the sequences handed over: bounds of the database's taxa
    ST_ID_MAX,
    ST_DUP_READS,     // 1 once two sequences shared a read index (equal neighbouring headers)
    ST_TMP0,          // scratch totals written by scans
    ST_TMP1,
    ST_WORDS = 16
};

struct kmg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    DevPool pool;

    // build state
    bool building = false;
    int k = 0, key_len = 0, rc = 0, kind = 0;
    int dust = -1, runofsame = -1;   // MakeDatabase -D / -R, -1 = off
    bool lds_rank_ok = false;        // returning LDS adds give stable ranks on this device (checked in kmg_create)
    bool ranged = false;             // kmg_build_set_key_range: keep the records of key buckets [range_lo, range_hi)
    u32 range_lo = 0, range_hi = 0;
    u64 *rec_keys = nullptr;     // record keys (left-aligned 2-bit k-mers; k > 32: bases 0 .. 31)
    u64 *rec_lo = nullptr;       // k > 32 (wide.h): bases 32 .. k-1 of every record, left-aligned
    uint8_t *rec_len = nullptr;  // k > 32 and min_k < k: bases of every record (k for the full windows)
    void *rec_vals = nullptr;    // u32 taxon | u64 (read << 16 | pos16)
    u64 rec_cap = 0;             // capacity in records
    u64 rec_upper = 0;           // host-known upper bound of records added so far
    // first batch of a build: records are written already partitioned by the first prefix digit
    // (extract_part.h); a second batch voids that and finish() runs every pass
    int n_adds = 0;
    bool prepart = false;
    int plan_pbits = 0;          // prefix bits the partition digit was planned for
    // read builds whose (read, pos) pairs fit 32 bits keep them as (read << compact_pb | pos) through the sort:
    // 12-byte records instead of 16.  0 = off (payloads are read << 16 | pos16 in 64 bits).
    int compact_pb = 0;
    bool sort_as_reads = false;  // profiling only: attribute 32-bit-payload sort launches to the read kernels
    u64 *d_state = nullptr;      // ST_WORDS device words
    void *h_pin = nullptr;       // PIN_BYTES of pinned host memory: landing pad of the small result readbacks
    // Readbacks that do not stop the stream: the copy into h_snap is queued where its value is final, an event
    // follows it, and the host waits for the event only when it needs the value -- later kernels are queued by then.
    void *h_snap = nullptr;      // [0, 128) build state after the count-time kernels, [128, 132) longest sequence
    hipEvent_t ev_len = nullptr, ev_snap = nullptr;
    bool snap_valid = false;     // h_snap holds the state finish() needs (single fused batch)
    std::vector<void *> staged;  // device copies of host batches, released at finish
    // variable-length builds (kmg_build_set_min_k): the k-mers shorter than k, in emission order, every batch appended
    int min_k = 0;               // == k unless set
    u64 *sh_keys = nullptr;      // left-aligned, zero-padded
    void *sh_vals = nullptr;     // u32 taxon | u64 (read << 16 | pos16)
    uint8_t *sh_len = nullptr;
    u64 sh_n = 0, sh_cap = 0;
    u64 *match_scratch = nullptr; // this device's copy of the file-scope bounds array (kmg_match_dev_noalloc)
    hipEvent_t ev_scratch = nullptr;   // recorded behind this context's last no-allocation join (orders other contexts' joins)
    // single-pass (chained) scans and their relatives, scan.h: tile status words, two control words, the launch epoch
    u64 *scan_status = nullptr;
    size_t scan_status_cap = 0;
    u32 *scan_ctl = nullptr;
    u32 scan_epoch = 0;

    // profiling
    bool prof_on = false;
    ProfSlot prof[KP_COUNT_];
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
};

static inline u64 next_db_id() {
    static std::atomic<u64> n{1};
    return n.fetch_add(1);
}

struct kmg_db {
    kmg_ctx *ctx = nullptr;
    const u64 id = next_db_id();   // never reused: names a database where a pointer could be a later one's
    int k = 0, key_len = 0, rc = 0, kind = 0;
    int min_k = 0;              // shortest k-mer (== k: every k-mer has k bases and lens is null)
    uint8_t *lens = nullptr;    // U: length of every k-mer of a variable-length database, ascending inside equal keys
    // merged from variable-length files that shared keys: the search over several lengths needs the files as they were
    // (the reference's trie takes a key's k-mers file by file), so the merged database keeps copies of them, in order
    bool shared_key_merge = false;
    std::vector<kmg_db *> parts;
    bool tax_bounds = false;    // taxon databases: every taxon lies in [tax_lo, tax_hi] (known from the build's sequence ids)
    int32_t tax_lo = 0, tax_hi = 0;
    u64 U = 0, V = 0;
    u64 *keys = nullptr;        // U; zero-padded to k bases: (keys, lens) ascending == Sequence.compareTo order
    u64 *keys_lo = nullptr;     // k > 32 only (wide.h): bases 32 .. k-1, left-aligned; (keys, keys_lo) ascends strictly
    u64 *val_off = nullptr;     // U + 1
    int32_t *val_a = nullptr;   // V
    int32_t *val_b = nullptr;   // V
    u64 *vals_packed = nullptr; // read DBs built here: V x (read << 16 | pos16); val_a/val_b unpacked on demand
    u32 *vals_packed32 = nullptr; // ... or V x (read << pos_bits | pos) when the build ran compact
    int pos_bits = 16;
    u64 *bucket_off = nullptr;  // 4^key_len + 1
    // kmg_match_near's sizing call leaves its offsets here for the filling call that follows (same reference, same
    // max_diff): the neighbour look-ups of the count are not repeated
    u64 *near_off = nullptr;    // U + 1
    u64 near_ref_id = 0, near_pairs = 0;
    int near_diff = -1;
    u32 *near_log_s = nullptr, *near_log_ref = nullptr;   // ... and the matches themselves, unordered (freed by the filling call)
    uint8_t *near_log_dist = nullptr;
    // Work another context queued on ITS stream that reads this database's arrays (kmg_match_dev with a database
    // of another context of the same GPU): one event per such context, recorded after its last launch.  The
    // owner's stream waits for them before the arrays go back to the owner's pool (db_release), so a later build
    // on the owner's stream cannot overwrite what the foreign kernel is still reading.
    std::mutex foreign_mu;
    std::map<kmg_ctx *, hipEvent_t> foreign_use;
};

#define PIN_BYTES 256
// Small device -> host readback + stream sync.  Through the context's pinned block: a pageable destination makes
// the runtime stage the copy with a blit kernel and a second wait (visible as ~50 us GPU-idle gaps in the trace).
static inline hipError_t read_back(kmg_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    if (!ctx->h_pin || bytes > PIN_BYTES) {
        hipError_t e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
    }
    hipError_t e = hipMemcpyAsync(ctx->h_pin, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) memcpy(dst, ctx->h_pin, bytes);
    return e;
}

static inline int val_bytes(int kind) { return kind == KMG_PAYLOAD_TAXON ? 4 : 8; }

// ---- launch timing ------------------------------------------------------------------------------
struct ProfScope {
    kmg_ctx *ctx;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(kmg_ctx *c, int id_, u64 algo_bytes) : ctx(c), id(id_) {
        if (!ctx->prof_on) return;
        ctx->prof[id].bytes += algo_bytes;
        a = get();
        b = get();
        if (a && b) (void)hipEventRecord(a, ctx->stream);
    }
    ~ProfScope() {
        if (!ctx->prof_on || !a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({id, a, b});
    }
    hipEvent_t get() {
        if (!ctx->event_pool.empty()) {
            hipEvent_t e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

static inline void prof_drain(kmg_ctx *ctx) {
    for (auto &p : ctx->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof[p.id].ms += ms;
            ctx->prof[p.id].launches += 1;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

static inline u32 div_up_u64(u64 a, u64 b) { return (u32)((a + b - 1) / b); }
