// comm.hip — the exchange step of the sharded path inside ONE process: several contexts (GPUs), one x replica each.
//
// Reference: CSRMatrixMatVectorNuma gives every NUMA node a private copy of the whole x with
// memcpy(p[i].X, x.values, ...) (src/mat_vec.cpp:257,266) and leaves the partial y vectors where they are
// (:287-296).  Here the replicas are assembled on the devices: participant i holds its own slice of x (what a solver
// iteration produced there, or what the host uploaded once), and an all-gather fills in the slices of the others over
// xGMI — no detour through host memory.  Two transports:
//   "rccl"       one ncclComm per participant (ncclCommInitAll); the all-gather is a group of n broadcasts, one per
//                slice, so the slices may differ in length (the reference's last shard takes the remainder, :245-246).
//                RCCL is loaded with dlopen: the engine carries no link-time dependency on it.
//   "peer-copy"  every participant pulls the other slices with hipMemcpyPeerAsync on pull streams of its own, one per
//                peer, so on an xGMI mesh the n-1 links of a GPU carry their slices at the same time (a ring would be
//                bound by one link).  Always available; also the path when participants share a
//                device (several shards on one GPU: plain device-to-device copies).
// Ordering is by events, never by host waits: a pull starts when the slice's owner has reached the call on its
// stream, the receiver's stream continues when its pulls are in, and an owner may overwrite its slice only after
// everybody has pulled it.
#include <dlfcn.h>

#include <string>
#include <vector>

#include "common.hpp"

using namespace spmv;

namespace
{
// the handful of RCCL entry points used (rccl.h: ncclResult_t = int, 0 = success; ncclFloat64 = 8)
struct Rccl
{
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist)                                                       = nullptr;
    int (*CommDestroy)(void* comm)                                                                                       = nullptr;
    int (*GroupStart)()                                                                                                  = nullptr;
    int (*GroupEnd)()                                                                                                    = nullptr;
    int (*Broadcast)(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream)  = nullptr;
    const char* (*GetErrorString)(int)                                                                                   = nullptr;
    bool load()
    {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return false;
        CommInitAll    = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy    = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart     = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd       = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Broadcast      = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
    }
};
constexpr int kNcclFloat64 = 8;
}  // namespace

struct spmv_comm
{
    int                                   n = 0;
    std::vector<spmv_ctx*>                ctx;
    std::vector<hipEvent_t>               ready;   // participant i has reached the call on its stream
    std::vector<hipEvent_t>               pulled;  // participant i has every slice it wanted
    std::vector<std::vector<hipStream_t>> pull;    // pull[d][k]: streams of participant d, one per peer (k < n - 1)
    std::vector<std::vector<hipEvent_t>>  landed;  // landed[d][k]: the copy on pull[d][k] is done
    Rccl                                  rccl;
    std::vector<void*>                    nccl;  // one communicator per participant ("rccl" transport) or empty
    std::string                           backend = "peer-copy";
};

static void comm_free(spmv_comm* c)
{
    if (!c) return;
    for (int i = 0; i < c->n; ++i)
    {
        (void)hipSetDevice(c->ctx[(size_t)i]->device);
        if (i < (int)c->ready.size() && c->ready[(size_t)i]) (void)hipEventDestroy(c->ready[(size_t)i]);
        if (i < (int)c->pulled.size() && c->pulled[(size_t)i]) (void)hipEventDestroy(c->pulled[(size_t)i]);
        if (i < (int)c->pull.size())
            for (hipStream_t s : c->pull[(size_t)i])
                if (s) (void)hipStreamDestroy(s);
        if (i < (int)c->landed.size())
            for (hipEvent_t e : c->landed[(size_t)i])
                if (e) (void)hipEventDestroy(e);
    }
    for (void* k : c->nccl)
        if (k && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(k);
    if (c->rccl.lib) (void)dlclose(c->rccl.lib);
    delete c;
}

extern "C" int spmv_comm_create(spmv_ctx* const* ctxs, int32_t n, spmv_comm** out)
{
    SPMV_REQUIRE(ctxs && out && n >= 1, "spmv_comm_create: bad argument");
    for (int i = 0; i < n; ++i) SPMV_REQUIRE(ctxs[i], "spmv_comm_create: context %d is null", i);
    spmv_comm* c = new (std::nothrow) spmv_comm();
    if (!c) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    c->n = n;
    c->ctx.assign(ctxs, ctxs + n);
    c->ready.assign((size_t)n, nullptr);
    c->pulled.assign((size_t)n, nullptr);
    c->pull.assign((size_t)n, {});
    c->landed.assign((size_t)n, {});
    bool distinct = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && ctxs[i]->device != ctxs[j]->device;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; ++i)
    {
        e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready[(size_t)i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pulled[(size_t)i], hipEventDisableTiming);
        c->pull[(size_t)i].assign((size_t)std::max(n - 1, 0), nullptr);
        c->landed[(size_t)i].assign((size_t)std::max(n - 1, 0), nullptr);
        for (int k = 0; k < n - 1 && e == hipSuccess; ++k)
        {
            e = hipStreamCreateWithFlags(&c->pull[(size_t)i][(size_t)k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->landed[(size_t)i][(size_t)k], hipEventDisableTiming);
        }
        // direct loads/stores between the devices where the fabric allows it; a refusal only means staged copies
        for (int j = 0; j < n && e == hipSuccess; ++j)
            if (ctxs[j]->device != ctxs[i]->device)
            {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device) == hipSuccess && can)
                    if (hipDeviceEnablePeerAccess(ctxs[j]->device, 0) != hipSuccess) (void)hipGetLastError();  // "already enabled" is fine
            }
    }
    if (e != hipSuccess)
    {
        comm_free(c);
        SPMV_FAIL(SPMV_ERR_HIP, "spmv_comm_create: %s", hipGetErrorString(e));
    }
    // RCCL when every participant has a GPU of its own (its communicators are one per device) unless SPMV_COMM=peer
    const char* want = getenv("SPMV_COMM");
    if (n >= 2 && distinct && !(want && !strcmp(want, "peer")) && c->rccl.load())
    {
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
        c->nccl.assign((size_t)n, nullptr);
        const int rc = c->rccl.CommInitAll(c->nccl.data(), n, devs.data());
        if (rc == 0)
            c->backend = "rccl";
        else
            c->nccl.clear();  // fall back to peer copies; not an error
    }
    *out = c;
    return SPMV_OK;
}

extern "C" void spmv_comm_destroy(spmv_comm* c) { comm_free(c); }

extern "C" const char* spmv_comm_backend(const spmv_comm* c) { return c ? c->backend.c_str() : ""; }

// copy src[src_offset .. +n) -> dst[dst_offset .. +n) between any two contexts, ordered behind the work already queued
// on the source's stream, queued on the destination's stream
extern "C" int spmv_vec_copy(spmv_vec* dst, int64_t dst_offset, const spmv_vec* src, int64_t src_offset, int64_t n)
{
    SPMV_REQUIRE(dst && src && n >= 0 && dst_offset >= 0 && src_offset >= 0 && dst_offset + n <= dst->n && src_offset + n <= src->n,
                 "spmv_vec_copy: range outside the vectors");
    if (n == 0) return SPMV_OK;
    spmv_ctx *dc = dst->ctx, *sc = src->ctx;
    if (sc != dc)
    {
        // the source's queued work first (its own event; created per call: this is not the hot path)
        hipEvent_t ev = nullptr;
        SPMV_HIP(hipSetDevice(sc->device));
        SPMV_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, sc->stream);
        if (e == hipSuccess) e = hipSetDevice(dc->device);
        if (e == hipSuccess) e = hipStreamWaitEvent(dc->stream, ev, 0);
        (void)hipEventDestroy(ev);  // deferred by the runtime until the wait has been satisfied
        if (e != hipSuccess) SPMV_FAIL(SPMV_ERR_HIP, "spmv_vec_copy: %s", hipGetErrorString(e));
    }
    SPMV_HIP(hipSetDevice(dc->device));
    if (sc->device == dc->device)
        SPMV_HIP(hipMemcpyAsync(dst->d + dst_offset, src->d + src_offset, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, dc->stream));
    else
        SPMV_HIP(hipMemcpyPeerAsync(dst->d + dst_offset, dc->device, src->d + src_offset, sc->device, sizeof(double) * (size_t)n, dc->stream));
    return SPMV_OK;
}

extern "C" int spmv_comm_allgather(spmv_comm* c, spmv_vec* const* vecs, const int64_t* offsets)
{
    SPMV_REQUIRE(c && vecs && offsets, "spmv_comm_allgather: null argument");
    const int     n     = c->n;
    const int64_t total = offsets[n];
    for (int i = 0; i < n; ++i)
    {
        SPMV_REQUIRE(vecs[i] && vecs[i]->ctx == c->ctx[(size_t)i], "spmv_comm_allgather: vector %d does not belong to participant %d", i, i);
        SPMV_REQUIRE(vecs[i]->n >= total && offsets[i] >= 0 && offsets[i] <= offsets[i + 1],
                     "spmv_comm_allgather: vector %d holds %lld entries, the slices end at %lld", i, (long long)vecs[i]->n, (long long)total);
    }
    if (n == 1) return SPMV_OK;
    if (!c->nccl.empty())
    {
        // n broadcasts in one group: slice r from participant r to everybody, in place
        int rc = c->rccl.GroupStart();
        for (int r = 0; r < n && rc == 0; ++r)
        {
            const size_t count = (size_t)(offsets[r + 1] - offsets[r]);
            if (count == 0) continue;
            for (int i = 0; i < n && rc == 0; ++i)
            {
                double* p = vecs[i]->d + offsets[r];
                rc        = c->rccl.Broadcast(p, p, count, kNcclFloat64, r, c->nccl[(size_t)i], c->ctx[(size_t)i]->stream);
            }
        }
        const int rc_end = c->rccl.GroupEnd();
        if (rc == 0) rc = rc_end;
        if (rc != 0) SPMV_FAIL(SPMV_ERR_HIP, "spmv_comm_allgather (rccl): %s", c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "error");
        return SPMV_OK;
    }
    // peer copies
    for (int i = 0; i < n; ++i)
    {
        SPMV_HIP(hipSetDevice(c->ctx[(size_t)i]->device));
        SPMV_HIP(hipEventRecord(c->ready[(size_t)i], c->ctx[(size_t)i]->stream));
    }
    for (int d = 0; d < n; ++d)
    {
        spmv_ctx* dc = c->ctx[(size_t)d];
        SPMV_HIP(hipSetDevice(dc->device));
        int k = 0;
        for (int s = 0; s < n; ++s)
        {
            if (s == d) continue;
            const int64_t cnt = offsets[s + 1] - offsets[s];
            hipStream_t   h   = c->pull[(size_t)d][(size_t)k];
            hipEvent_t    ev  = c->landed[(size_t)d][(size_t)k];
            ++k;
            if (cnt == 0) continue;
            spmv_ctx* sc = c->ctx[(size_t)s];
            SPMV_HIP(hipStreamWaitEvent(h, c->ready[(size_t)s], 0));  // the owner has produced its slice
            SPMV_HIP(hipStreamWaitEvent(h, c->ready[(size_t)d], 0));  // the receiver no longer reads the old replica
            double*       to   = vecs[d]->d + offsets[s];
            const double* from = vecs[s]->d + offsets[s];
            if (sc->device == dc->device)
                SPMV_HIP(hipMemcpyAsync(to, from, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, h));
            else
                SPMV_HIP(hipMemcpyPeerAsync(to, dc->device, from, sc->device, sizeof(double) * (size_t)cnt, h));
            SPMV_HIP(hipEventRecord(ev, h));
            SPMV_HIP(hipStreamWaitEvent(dc->stream, ev, 0));
        }
        SPMV_HIP(hipEventRecord(c->pulled[(size_t)d], dc->stream));
    }
    // an owner may overwrite its slice only when everybody has pulled it
    for (int s = 0; s < n; ++s)
    {
        SPMV_HIP(hipSetDevice(c->ctx[(size_t)s]->device));
        for (int d = 0; d < n; ++d)
            if (d != s) SPMV_HIP(hipStreamWaitEvent(c->ctx[(size_t)s]->stream, c->pulled[(size_t)d], 0));
    }
    return SPMV_OK;
}
