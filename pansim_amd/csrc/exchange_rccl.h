// exchange_rccl.h -- the native provider of ps_exchange_fn: the OR of the donor-sharded HGT deltas over RCCL,
// one process per GPU, for a host that has no torch (a Rust main() binds ps_rccl_* like every other entry of
// include/pansim_hip.h).  Included by pansim_capi.hip (one translation unit, its error helpers).
//
// RCCL has no OR reduction, so the exchange is the same three steps as pansim_amd/distributed.py::or_all_reduce:
//   1. all-to-all of the K row slices (ncclSend / ncclRecv to every peer inside one group: slice k of every rank's
//      delta arrives at rank k), 2. a local OR of the K received slices, 3. the merged slices to everyone: a second
//      group of direct sends (round 5; PANSIM_RCCL_GATHER=ring selects ncclAllGather instead).
// Per rank and call 2 (K - 1) / K x the buffer is sent and as much received, point to point over xGMI: both steps put one
// slice on each of the K - 1 links of a fully connected node at the same time.
// librccl.so is opened with dlopen at the first use: libpansim_hip.so itself links no RCCL and includes no RCCL header
// (the few ABI types the calls need are declared below), so a box without RCCL builds the library unchanged and gets
// PS_ERR_NO_DEVICE from ps_rccl_* and nothing else.  Failures of RCCL calls themselves are PS_ERR_STATE.
// PANSIM_RCCL_LIBRARY names another library to open instead (tests/fake_rccl.cpp: a test double that runs several
// ranks on one GPU, so that K > 1 of this file executes on a one-GPU box).
#pragma once

#include <dlfcn.h>

// The part of the NCCL / RCCL C ABI used here (rccl.h: NCCL_UNIQUE_ID_BYTES 128, ncclResult_t ncclSuccess = 0,
// ncclDataType_t ncclUint64 = 5, ncclComm_t an opaque pointer).  Every function is resolved through dlsym.
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclUint64 = 5;

namespace ps_rccl {

struct api {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;        // why loading failed
};

static api g_api;

static api *load()
{
    static std::mutex mu;
    static bool tried = false;
    api &a = g_api;
    std::lock_guard<std::mutex> lock(mu);
    if (tried) return a.so ? &a : nullptr;
    tried = true;
    const char *override_path = getenv("PANSIM_RCCL_LIBRARY");
    const char *names[] = { override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void *so = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
        const char *e = dlerror();
        a.why = e ? e : "dlopen failed";
    }
    if (!so) return nullptr;
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(so, name);
        if (!p) {
            ok = false;
            a.why = std::string("librccl has no symbol ") + name;
        }
        return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(so);
        return nullptr;
    }
    a.so = so;
    return &a;
}

static std::string why_not()
{
    return "librccl.so could not be loaded (" + (g_api.why.empty() ? std::string("no reason recorded") : g_api.why)
           + "): the RCCL exchange provider is unavailable; PANSIM_RCCL_LIBRARY overrides the path";
}

}  // namespace ps_rccl

struct ps_rccl_scratch {
    // work buffers of one buffer length, allocated at its first call and kept: the padded delta / the gathered result
    // (K slices of `part` words), the K received slices, their OR
    uint64_t n_words = 0, part = 0;
    uint64_t *send = nullptr, *recv = nullptr, *mine = nullptr;
};

struct ps_rccl_exchange {
    ps_rccl::api *api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    // a run exchanges buffers of two lengths per generation (the HGT delta matrix; the N average distances of a
    // --competition_strength run): one scratch set per length, so that alternating calls never reallocate
    ps_rccl_scratch scratch[4];
    uint64_t calls = 0, bytes = 0;
    bool ring_gather = false;       // PANSIM_RCCL_GATHER=ring: ncclAllGather instead of the direct sends (A/B on real links)
};

#define NCCLCHK(x, expr)                                                                                         \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return ps_fail(PS_ERR_STATE, "%s failed: %s (%s:%d)", #expr, (x)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// mine[w] = OR over k of recv[k * part + w]
__global__ void __launch_bounds__(256) rccl_or_slices_kernel(uint64_t *mine, const uint64_t *recv, uint64_t part, uint32_t K)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= part) return;
    uint64_t v = recv[w];
    for (uint32_t k = 1; k < K; k++) v |= recv[(uint64_t)k * part + w];
    mine[w] = v;
}

extern "C" int ps_rccl_available(void)
{
    return ps_rccl::load() ? 1 : 0;
}

extern "C" int ps_rccl_unique_id(uint8_t *id_out)
{
    if (!id_out) return ps_fail(PS_ERR_INVALID, "null argument");
    ps_rccl::api *a = ps_rccl::load();
    if (!a) return ps_fail(PS_ERR_NO_DEVICE, "%s", ps_rccl::why_not().c_str());
    ncclUniqueId id;
    NCCLCHK(a, a->GetUniqueId(&id));
    static_assert(sizeof(id) == PS_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, sizeof(id));
    return PS_OK;
}

extern "C" void ps_rccl_exchange_destroy(ps_rccl_exchange *x)
{
    if (!x) return;
    (void)hipSetDevice(x->device);
    for (ps_rccl_scratch &c : x->scratch) {
        if (c.send) (void)hipFree(c.send);
        if (c.recv) (void)hipFree(c.recv);
        if (c.mine) (void)hipFree(c.mine);
    }
    if (x->comm && x->api) (void)x->api->CommDestroy(x->comm);
    delete x;
}

extern "C" int ps_rccl_exchange_create(const uint8_t *id, int rank, int world, int device, ps_rccl_exchange **out)
{
    if (!id || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return ps_fail(PS_ERR_INVALID, "bad rank %d of %d", rank, world);
    ps_rccl::api *a = ps_rccl::load();
    if (!a) return ps_fail(PS_ERR_NO_DEVICE, "%s", ps_rccl::why_not().c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return ps_fail(PS_ERR_NO_DEVICE, "no HIP device is visible: libpansim_hip has no CPU path");
    if (device < 0) HIPCHK(hipGetDevice(&device));
    if (device >= ndev) return ps_fail(PS_ERR_INVALID, "device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    ps_rccl_exchange *x = new ps_rccl_exchange();
    x->api = a;
    x->rank = rank;
    x->world = world;
    x->device = device;
    if (const char *e = getenv("PANSIM_RCCL_GATHER")) x->ring_gather = strcmp(e, "ring") == 0;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = a->CommInitRank(&x->comm, world, uid, rank);       // (collective: every rank of the run is inside it)
    if (r != ncclSuccess) {
        x->comm = nullptr;
        ps_rccl_exchange_destroy(x);
        return ps_fail(PS_ERR_STATE, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, a->GetErrorString(r));
    }
    *out = x;
    return PS_OK;
}

// the ps_exchange_fn: ctx is the ps_rccl_exchange of this rank
extern "C" int ps_exchange_rccl(void *ctx, void *d_words, uint64_t n_words, void *hip_stream)
{
    ps_rccl_exchange *x = (ps_rccl_exchange *)ctx;
    if (!x || !x->comm) return ps_fail(PS_ERR_INVALID, "null exchange handle");
    if (!d_words || !n_words) return ps_fail(PS_ERR_INVALID, "empty delta buffer");
    hipStream_t st = (hipStream_t)hip_stream;
    const uint64_t K = (uint64_t)x->world;     // (K = 1 runs the same calls: a send / receive to itself, a gather of one)
    HIPCHK(hipSetDevice(x->device));
    ps_rccl_scratch *sc = nullptr;
    for (ps_rccl_scratch &c : x->scratch)
        if (c.n_words == n_words) { sc = &c; break; }
    if (!sc) {
        for (ps_rccl_scratch &c : x->scratch)
            if (c.n_words == 0) { sc = &c; break; }
        if (!sc) {                               // all four sets taken by other lengths: recycle the first
            sc = &x->scratch[0];
            HIPCHK(hipStreamSynchronize(st));
            if (sc->send) HIPCHK(hipFree(sc->send));
            if (sc->recv) HIPCHK(hipFree(sc->recv));
            if (sc->mine) HIPCHK(hipFree(sc->mine));
            *sc = ps_rccl_scratch{};
        }
        const uint64_t part = (n_words + K - 1) / K;
        HIPCHK(hipMalloc(&sc->send, K * part * 8));
        HIPCHK(hipMalloc(&sc->recv, K * part * 8));
        HIPCHK(hipMalloc(&sc->mine, part * 8));
        // the pad words behind the delta are zeroed once and stay zero (the all-gather writes the OR of zeros there)
        HIPCHK(hipMemsetAsync(sc->send, 0, K * part * 8, st));
        sc->part = part;
        sc->n_words = n_words;
    }
    const uint64_t part = sc->part;
    ps_rccl::api *a = x->api;
    HIPCHK(hipMemcpyAsync(sc->send, d_words, n_words * 8, hipMemcpyDeviceToDevice, st));
    NCCLCHK(a, a->GroupStart());
    // a failure inside the bracket must not leave the group open on this thread (later RCCL calls -- the destroy, a
    // retry, torch's own collectives on the same librccl -- would queue into a group that never closes): keep the
    // first error, always close the group, then fail
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (uint64_t k = 0; k < K && first == ncclSuccess; k++) {
        first = a->Send(sc->send + k * part, part, ncclUint64, (int)k, x->comm, st);
        what = "ncclSend";
        if (first != ncclSuccess) break;
        first = a->Recv(sc->recv + k * part, part, ncclUint64, (int)k, x->comm, st);
        what = "ncclRecv";
    }
    const ncclResult_t closed = a->GroupEnd();
    if (first != ncclSuccess)
        return ps_fail(PS_ERR_STATE, "%s failed inside the all-to-all group: %s", what, a->GetErrorString(first));
    if (closed != ncclSuccess)
        return ps_fail(PS_ERR_STATE, "ncclGroupEnd failed: %s", a->GetErrorString(closed));
    rccl_or_slices_kernel<<<(uint32_t)((part + 255) / 256), 256, 0, st>>>(sc->mine, sc->recv, part, (uint32_t)K);
    HIPCHK(hipGetLastError());
    if (x->ring_gather) {
        NCCLCHK(a, a->AllGather(sc->mine, sc->send, part, ncclUint64, x->comm, st));
    } else {
        // the all-gather as a second DIRECT all-to-all: this rank's merged slice goes to every peer over that peer's own
        // xGMI link, all links at once (slice / link rate), where ncclAllGather's ring passes every slice through K - 1
        // hops and is bound by ONE link ((K - 1) / K x buffer / link rate: 7 x longer at K = 8)
        NCCLCHK(a, a->GroupStart());
        first = ncclSuccess;
        for (uint64_t k = 0; k < K && first == ncclSuccess; k++) {
            first = a->Send(sc->mine, part, ncclUint64, (int)k, x->comm, st);
            what = "ncclSend";
            if (first != ncclSuccess) break;
            first = a->Recv(sc->send + k * part, part, ncclUint64, (int)k, x->comm, st);
            what = "ncclRecv";
        }
        const ncclResult_t closed2 = a->GroupEnd();
        if (first != ncclSuccess)
            return ps_fail(PS_ERR_STATE, "%s failed inside the gather group: %s", what, a->GetErrorString(first));
        if (closed2 != ncclSuccess)
            return ps_fail(PS_ERR_STATE, "ncclGroupEnd failed: %s", a->GetErrorString(closed2));
    }
    HIPCHK(hipMemcpyAsync(d_words, sc->send, n_words * 8, hipMemcpyDeviceToDevice, st));
    x->calls++;
    x->bytes += 2 * part * 8 * (K - 1);
    return PS_OK;
}

extern "C" int ps_rccl_exchange_stats(ps_rccl_exchange *x, int reset, uint64_t *calls, uint64_t *bytes)
{
    if (!x) return ps_fail(PS_ERR_INVALID, "null handle");
    if (calls) *calls = x->calls;
    if (bytes) *bytes = x->bytes;
    if (reset) x->calls = x->bytes = 0;
    return PS_OK;
}
