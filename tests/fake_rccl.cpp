// fake_rccl.cpp -- TEST DOUBLE of librccl (test infrastructure; never loaded by the product unless PANSIM_RCCL_LIBRARY
// names it).  It implements the nine NCCL entry points pansim_amd/csrc/exchange_rccl.h resolves with dlsym, for several
// RANKS INSIDE ONE PROCESS ON ONE GPU (one host thread per rank), so that the K > 1 code of ps_exchange_rccl -- slice
// offsets, the pad of lengths K does not divide, the OR over K received slices, the gather offsets, the scratch sets --
// executes on a one-GPU box.  Real RCCL refuses two ranks on one device, and the GPU boxes allow only six processes on
// the card, so worlds of 8 can only be played by threads.
//
// Semantics kept from NCCL: ncclCommInitRank blocks until every rank of the id has joined; sends and receives issued
// between ncclGroupStart / ncclGroupEnd are matched as a set (no ordering deadlock); a receive copies exactly the words
// the matching send posted, on the receiver's stream; ncclAllGather places rank r's block at recvbuff + r * count.
// Simplification: every call completes before it returns (host-synchronous), which NCCL permits but does not promise.
// The calls are checked harder than RCCL would: mismatched counts, peers out of range, a group left open or a
// communicator used after destruction fail with ncclInvalidUsage, and PANSIM_FAKE_RCCL_FAIL=send|recv|groupend|allgather
// makes that call fail once (the error paths of exchange_rccl.h).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
}

namespace {

struct Post {               // one posted send, or one rank's all-gather block
    const void *buf = nullptr;
    size_t bytes = 0;
    bool full = false;
};

struct World {
    int n = 0, joined = 0, left = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Post> p2p;          // [src * n + dst]
    std::vector<Post> gather;       // [rank]
    int bar_count = 0;
    uint64_t bar_gen = 0;
    bool broken = false;

    bool barrier(std::unique_lock<std::mutex> &lk)
    {
        const uint64_t g = bar_gen;
        if (++bar_count == n) {
            bar_count = 0;
            bar_gen++;
            cv.notify_all();
            return true;
        }
        return cv.wait_for(lk, std::chrono::seconds(60), [&] { return bar_gen != g || broken; }) && !broken;
    }
};

struct Comm {
    World *w = nullptr;
    int rank = 0;
    bool alive = true;
};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *c;
    hipStream_t st;
};

std::mutex g_mu;
std::map<uint64_t, World *> g_worlds;
uint64_t g_next_id = 0x5eed0001;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local bool t_group_failed = false;

size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
        case 0: case 1: return 1;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        case 6: case 9: return 2;
        default: return 0;
    }
}

bool inject(const char *what)
{
    static std::mutex mu;
    static bool used = false;
    const char *e = getenv("PANSIM_FAKE_RCCL_FAIL");
    if (!e || strcmp(e, what) != 0) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (used) return false;
    used = true;
    return true;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    // sends first (a post never blocks), then the receives, then every send waits until it has been consumed -- so the
    // caller may overwrite its send buffer when the call returns, and a (src, dst) slot is free for the next call
    for (Op &o : ops)
        if (o.send) {
            if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;     // the data is complete
            World *w = o.c->w;
            std::unique_lock<std::mutex> lk(w->mu);
            Post &p = w->p2p[(size_t)o.c->rank * w->n + o.peer];
            if (p.full) return ncclInvalidUsage;            // two sends to one peer inside one group
            p.buf = o.buf;
            p.bytes = o.bytes;
            p.full = true;
            w->cv.notify_all();
        }
    for (Op &o : ops)
        if (!o.send) {
            World *w = o.c->w;
            const void *src = nullptr;
            {
                std::unique_lock<std::mutex> lk(w->mu);
                Post &p = w->p2p[(size_t)o.peer * w->n + o.c->rank];
                if (!w->cv.wait_for(lk, std::chrono::seconds(60), [&] { return p.full || w->broken; }) || w->broken) return ncclSystemError;
                if (p.bytes != o.bytes) {
                    w->broken = true;
                    w->cv.notify_all();
                    return ncclInvalidUsage;                // send and receive disagree about the count
                }
                src = p.buf;
            }
            if (hipMemcpyAsync(o.buf, src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return ncclUnhandledCudaError;
            if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
            std::unique_lock<std::mutex> lk(w->mu);
            w->p2p[(size_t)o.peer * w->n + o.c->rank].full = false;
            w->cv.notify_all();
        }
    for (Op &o : ops)
        if (o.send) {
            World *w = o.c->w;
            std::unique_lock<std::mutex> lk(w->mu);
            Post &p = w->p2p[(size_t)o.c->rank * w->n + o.peer];
            if (!w->cv.wait_for(lk, std::chrono::seconds(60), [&] { return !p.full || w->broken; }) || w->broken) return ncclSystemError;
        }
    return ncclSuccess;
}

ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t t, int peer, Comm *c, hipStream_t st)
{
    if (!c || !c->alive || !c->w) return ncclInvalidUsage;
    if (peer < 0 || peer >= c->w->n || !dtype_bytes(t) || (!buf && count)) return ncclInvalidArgument;
    Op o{send, buf, count * dtype_bytes(t), peer, c, st};
    if (t_depth > 0) {
        t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_ops(one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    memcpy(id->internal + 8, "fake_rccl", 9);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(Comm **comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal + 8, "fake_rccl", 9) != 0) return ncclInvalidArgument;
    uint64_t key;
    memcpy(&key, id.internal, sizeof(key));
    World *w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        World *&slot = g_worlds[key];
        if (!slot) {
            slot = new World();
            slot->n = nranks;
            slot->p2p.resize((size_t)nranks * nranks);
            slot->gather.resize((size_t)nranks);
        }
        w = slot;
    }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->n != nranks) return ncclInvalidArgument;
    w->joined++;
    w->cv.notify_all();
    // like RCCL, the call returns when every rank of the id is inside it
    if (!w->cv.wait_for(lk, std::chrono::seconds(60), [&] { return w->joined >= w->n; })) return ncclSystemError;
    Comm *c = new Comm();
    c->w = w;
    c->rank = rank;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(Comm *c)
{
    if (!c || !c->alive) return ncclInvalidArgument;
    c->alive = false;               // (the Comm itself is kept: a use after destroy is reported, not a crash)
    World *w = c->w;
    std::unique_lock<std::mutex> lk(w->mu);
    w->left++;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    if (t_depth == 0) {
        t_ops.clear();
        t_group_failed = false;
    }
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (inject("groupend")) return ncclInternalError;
    if (t_group_failed) return ncclSuccess;         // (a call inside the group already failed: nothing is launched)
    return run_ops(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, Comm *c, hipStream_t st)
{
    if (inject("send")) {
        t_group_failed = true;
        return ncclInternalError;
    }
    return p2p(true, (void *)buf, count, t, peer, c, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, Comm *c, hipStream_t st)
{
    if (inject("recv")) {
        t_group_failed = true;
        return ncclInternalError;
    }
    return p2p(false, buf, count, t, peer, c, st);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t t, Comm *c, hipStream_t st)
{
    if (!c || !c->alive || !c->w) return ncclInvalidUsage;
    if (t_depth > 0) return ncclInvalidUsage;       // (not needed by the library: kept out of groups)
    const size_t bytes = count * dtype_bytes(t);
    if (!dtype_bytes(t) || ((!sendbuff || !recvbuff) && count)) return ncclInvalidArgument;
    if (inject("allgather")) return ncclInternalError;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    World *w = c->w;
    std::vector<Post> all;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->gather[(size_t)c->rank] = Post{sendbuff, bytes, true};
        if (!w->barrier(lk)) return ncclSystemError;
        all = w->gather;
    }
    for (int r = 0; r < w->n; r++) {
        if (all[(size_t)r].bytes != bytes) return ncclInvalidUsage;
        void *dst = (uint8_t *)recvbuff + (size_t)r * bytes;
        if (dst == all[(size_t)r].buf) continue;        // in place
        if (hipMemcpyAsync(dst, all[(size_t)r].buf, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(w->mu);
    if (!w->barrier(lk)) return ncclSystemError;      // nobody reuses its block before everyone has copied it
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "a peer never arrived (fake_rccl, 60 s)";
        case ncclInternalError: return "injected failure (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        default: return "unknown result (fake_rccl)";
    }
}

}  // extern "C"
