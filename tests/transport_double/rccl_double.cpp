// rccl_double.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for the nine RCCL entry points csrc/comm.h looks up by name
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclAllGather, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString), so that the library's exchange steps can run with SEVERAL ranks on ONE GPU.
// RCCL itself refuses two ranks on one device; the GPU boxes of this project have one.
//
// Transport: every rank owns a mailbox in one POSIX shared-memory object (named by the unique id).  A round --
// one ncclGroupEnd, or one call outside a group -- is: wait for the stream, copy every send to the own mailbox
// (device -> host), barrier, copy what the peers addressed to this rank out of their mailboxes (host -> device),
// barrier.  Synchronous where RCCL is stream-ordered; the results are the same.  It is strict where RCCL would hang
// or corrupt: a receive whose element count or type differs from the matching send, a send nobody receives, or a
// peer that does not arrive within the time limit fails the call with a message.
//
// Built by tests/transport_double/Makefile with the soname librccl.so.1: a test process maps it (RTLD_GLOBAL) BEFORE
// the first kmg_comm_* call, and comm.h's dlopen("librccl.so.1", RTLD_NOLOAD) finds it.  Never installed, never on a
// library path, never loaded by the package: only tests/_exchange_rank.py maps it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int MAX_RANKS = 16;
constexpr int MAX_ENT = 256;
constexpr uint64_t BOX_BYTES = 96ull << 20;      // per rank and round (pages are only touched when used)
constexpr int WAIT_SECONDS = 120;

struct Entry { int32_t dst; int32_t dtype; uint64_t count, off, bytes; };
struct Box {
    uint32_t nent;
    uint32_t pad[15];
    Entry ent[MAX_ENT];
    alignas(4096) uint8_t data[BOX_BYTES];
};
struct Shared {
    std::atomic<uint32_t> arrived, generation, attached, failed;
    uint32_t nranks;
    alignas(4096) Box box[1];     // nranks of them
};

struct Op { int kind; const void *src; void *dst; uint64_t count; int dtype, peer; hipStream_t stream; };   // kind 0 send, 1 recv, 2 all-gather

struct Comm {
    Shared *sh = nullptr;
    size_t bytes = 0;
    int nranks = 0, rank = 0;
    std::string name;
};

thread_local std::string g_err = "no error";
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_group_comm = nullptr;
Comm *g_last_comm = nullptr;            // an empty group still meets the peers' round: on the process's communicator
std::atomic<uint64_t> g_stats[4];        // sends, receives, all-gathers, rounds

int fail(int code, const std::string &m) { g_err = "transport double: " + m; return code; }

size_t dtype_size(int dt) {
    switch (dt) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 0; }
}

int barrier(Comm *c) {
    Shared *s = c->sh;
    const uint32_t gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == (uint32_t)c->nranks) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (s->failed.load()) return fail(6, "a peer failed its side of the round");
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(WAIT_SECONDS)) {
            s->failed.store(1);
            return fail(6, "a peer did not arrive at the barrier within the time limit");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return 0;
}

int round_of(Comm *c, std::vector<Op> &ops) {
    g_stats[3]++;
    Box *mine = &c->sh->box[c->rank];
    std::vector<hipStream_t> seen;
    for (const Op &o : ops) {
        bool dup = false;
        for (hipStream_t s : seen) dup = dup || s == o.stream;
        if (dup) continue;
        seen.push_back(o.stream);
        if (hipStreamSynchronize(o.stream) != hipSuccess) { c->sh->failed.store(1); return fail(1, "hipStreamSynchronize failed"); }
    }
    int rc = 0;
    uint64_t off = 0;
    mine->nent = 0;
    for (const Op &o : ops) {
        if (o.kind == 1) continue;
        const uint64_t bytes = o.count * dtype_size(o.dtype);
        if (mine->nent >= MAX_ENT || off + bytes > BOX_BYTES) { rc = fail(3, "a round's sends do not fit the mailbox"); break; }
        if (bytes && hipMemcpy(mine->data + off, o.src, bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(1, "copy out of the send buffer failed"); break; }
        mine->ent[mine->nent++] = Entry{o.kind == 2 ? -1 : o.peer, o.dtype, o.count, off, bytes};
        off += (bytes + 255) & ~255ull;
        g_stats[o.kind == 2 ? 2 : 0]++;
    }
    if (rc) c->sh->failed.store(1);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (int b = barrier(c)) return rc ? rc : b;
    if (!rc) {
        std::vector<uint32_t> cursor(c->nranks, 0);
        for (const Op &o : ops) {
            if (o.kind == 1) {
                if (o.peer < 0 || o.peer >= c->nranks) { rc = fail(4, "receive from a rank that does not exist"); break; }
                const Box *pb = &c->sh->box[o.peer];
                uint32_t &i = cursor[o.peer];
                while (i < pb->nent && pb->ent[i].dst != c->rank) i++;
                if (i >= pb->nent) { rc = fail(5, "receive from rank " + std::to_string(o.peer) + " without a matching send"); break; }
                const Entry &e = pb->ent[i++];
                if (e.count != o.count || e.dtype != o.dtype) {
                    rc = fail(5, "receive of " + std::to_string(o.count) + " elements (type " + std::to_string(o.dtype) + ") from rank " +
                                     std::to_string(o.peer) + " meets a send of " + std::to_string(e.count) + " (type " + std::to_string(e.dtype) + ")");
                    break;
                }
                if (e.bytes && hipMemcpy(o.dst, pb->data + e.off, e.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(1, "copy into the receive buffer failed"); break; }
                g_stats[1]++;
            } else if (o.kind == 2) {
                const uint64_t bytes = o.count * dtype_size(o.dtype);
                for (int p = 0; p < c->nranks && !rc; p++) {
                    const Box *pb = &c->sh->box[p];
                    const Entry *e = nullptr;
                    for (uint32_t i = 0; i < pb->nent; i++) if (pb->ent[i].dst == -1) { e = &pb->ent[i]; break; }
                    if (!e || e->count != o.count || e->dtype != o.dtype) { rc = fail(5, "all-gather: rank " + std::to_string(p) + " brought another size or type"); break; }
                    if (bytes && hipMemcpy((char *)o.dst + (uint64_t)p * bytes, pb->data + e->off, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = fail(1, "copy into the gather buffer failed");
                }
                if (rc) break;
            }
        }
        if (!rc) {
            // every send addressed to this rank must have met a receive
            for (int p = 0; p < c->nranks && !rc; p++) {
                const Box *pb = &c->sh->box[p];
                uint32_t i = cursor[p];
                for (; i < pb->nent; i++)
                    if (pb->ent[i].dst == c->rank) { rc = fail(5, "rank " + std::to_string(p) + " sent " + std::to_string(pb->ent[i].count) + " elements that no receive took"); break; }
            }
        }
        if (rc) c->sh->failed.store(1);
    }
    if (int b = barrier(c)) return rc ? rc : b;
    return rc;
}

int submit(Comm *c, const Op &o) {
    if (!c || !c->sh) return fail(4, "null communicator");
    if (g_depth > 0) {
        if (g_group_comm && g_group_comm != c) return fail(5, "one communicator per group in this double");
        g_group_comm = c;
        g_ops.push_back(o);
        return 0;
    }
    std::vector<Op> one{o};
    return round_of(c, one);
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return fail(4, "null id");
    memset(id->internal, 0, sizeof id->internal);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/kmg-transport-double-%d-%llx", (int)getpid(), (unsigned long long)now);
    return 0;
}

int ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return fail(4, "bad rank / number of ranks");
    id.internal[sizeof id.internal - 1] = 0;
    if (id.internal[0] != '/') return fail(4, "this id was not made by the double's ncclGetUniqueId");
    Comm *c = new Comm();
    c->nranks = nranks; c->rank = rank; c->name = id.internal;
    c->bytes = sizeof(Shared) + (size_t)(nranks - 1) * sizeof(Box);
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return fail(2, "shm_open failed"); }
    if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return fail(2, "ftruncate failed"); }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return fail(2, "mmap failed"); }
    c->sh = (Shared *)p;               // a fresh object reads as zeros: counters start at 0
    c->sh->nranks = (uint32_t)nranks;
    c->sh->attached.fetch_add(1);
    // everybody present before anybody goes on (ncclCommInitRank is collective)
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < (uint32_t)nranks) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(WAIT_SECONDS)) {
            munmap(p, c->bytes);
            shm_unlink(c->name.c_str());
            delete c;
            return fail(6, "not every rank reached ncclCommInitRank within the time limit");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    *out = c;
    g_last_comm = c;
    return 0;
}

int ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    if (g_last_comm == c) g_last_comm = nullptr;
    if (c->sh) {
        const bool last = c->sh->attached.fetch_sub(1) == 1;
        munmap(c->sh, c->bytes);
        if (last) shm_unlink(c->name.c_str());
    }
    delete c;
    return 0;
}

int ncclGroupStart() { g_depth++; return 0; }

int ncclGroupEnd() {
    if (g_depth <= 0) return fail(5, "ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return 0;
    Comm *c = g_group_comm;
    std::vector<Op> ops;
    ops.swap(g_ops);
    g_group_comm = nullptr;
    if (!c) c = g_last_comm;     // an empty group: this rank neither sends nor receives, its peers' round still counts it in
    if (!c) return 0;
    return round_of(c, ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!dtype_size(dtype)) return fail(4, "unknown data type");
    return submit((Comm *)comm, Op{0, buf, nullptr, count, dtype, peer, stream});
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!dtype_size(dtype)) return fail(4, "unknown data type");
    return submit((Comm *)comm, Op{1, nullptr, buf, count, dtype, peer, stream});
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t comm, hipStream_t stream) {
    if (!dtype_size(dtype)) return fail(4, "unknown data type");
    return submit((Comm *)comm, Op{2, send, recv, count, dtype, -1, stream});
}

const char *ncclGetErrorString(int) { return g_err.c_str(); }

// test hook: how many sends / receives / all-gathers / rounds went through this process's copy
void kmg_transport_double_stats(uint64_t out[4]) { for (int i = 0; i < 4; i++) out[i] = g_stats[i].load(); }

}  // extern "C"
