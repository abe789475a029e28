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
#include <rccl/rccl.h>  // prototypes only (decltype); the library itself is dlopen'ed

#include <string>
#include <type_traits>
#include <vector>

#include "common.hpp"

using namespace spmv;

namespace
{
// The handful of RCCL entry points used.  Their types are taken from rccl.h itself (decltype of the declarations), so
// the pointers dlsym returns are called through exactly the prototypes of the installed RCCL: nothing is declared by
// hand and nothing can drift.  The header is only read at compile time; the library is still loaded with dlopen.
struct Rccl
{
    void* lib = nullptr;
    decltype(&ncclCommInitAll)    CommInitAll    = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
    decltype(&ncclGroupStart)     GroupStart     = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
    decltype(&ncclBroadcast)      Broadcast      = nullptr;
    decltype(&ncclAllGather)      AllGather      = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion)     GetVersion     = nullptr;
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
        AllGather      = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        GetVersion     = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && AllGather;
    }
};
// the shapes this file relies on, checked against the header (a change in RCCL's API fails the build, not the 8-GPU run)
static_assert(std::is_same_v<decltype(&ncclBroadcast), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>,
              "ncclBroadcast(sendbuff, recvbuff, count, datatype, root, comm, stream)");
static_assert(std::is_same_v<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>,
              "ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream)");
static_assert(std::is_same_v<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t*, int, const int*)>, "ncclCommInitAll(comms, ndev, devlist)");
static_assert(ncclFloat64 == 8 && ncclSuccess == 0, "rccl.h: ncclFloat64 / ncclSuccess");

// leaves the calling thread's current HIP device as it found it (callers share the process with torch)
struct DeviceRestore
{
    int  dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};
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
    std::vector<ncclComm_t>               nccl;  // one communicator per participant ("rccl" transport) or empty
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
    for (ncclComm_t k : c->nccl)
        if (k && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(k);
    if (c->rccl.lib) (void)dlclose(c->rccl.lib);
    delete c;
}

// The all-gather over RCCL, queued on the participants' streams: one in-place ncclAllGather when the slices are equal
// (the usual case: rows / n per shard), else a group of n broadcasts, one per slice, so the slices may differ in length
// (the reference's last shard takes the remainder, src/mat_vec.cpp:245-246).  One host thread drives every
// communicator, so the calls of all participants sit in ONE group.
// force_broadcasts: take the ragged form whatever the slices (the self-check's second pass: with ONE participant any
// offsets are "equal", and the group of ncclBroadcast calls would otherwise never run on a one-GPU box).
static int rccl_allgather(spmv_comm* c, spmv_vec* const* vecs, const int64_t* offsets, bool force_broadcasts = false)
{
    const int n     = c->n;
    bool      equal = offsets[0] == 0 && !force_broadcasts;
    for (int r = 0; r < n; ++r) equal = equal && offsets[r + 1] - offsets[r] == offsets[1] - offsets[0];
    ncclResult_t rc = c->rccl.GroupStart();
    if (equal)
    {
        const size_t count = (size_t)(offsets[1] - offsets[0]);
        for (int i = 0; i < n && rc == ncclSuccess && count > 0; ++i)
            rc = c->rccl.AllGather(vecs[i]->d + offsets[i], vecs[i]->d, count, ncclFloat64, c->nccl[(size_t)i], c->ctx[(size_t)i]->stream);
    }
    else
        for (int r = 0; r < n && rc == ncclSuccess; ++r)
        {
            const size_t count = (size_t)(offsets[r + 1] - offsets[r]);
            if (count == 0) continue;
            for (int i = 0; i < n && rc == ncclSuccess; ++i)
            {
                double* p = vecs[i]->d + offsets[r];
                rc        = c->rccl.Broadcast(p, p, count, ncclFloat64, r, c->nccl[(size_t)i], c->ctx[(size_t)i]->stream);
            }
        }
    const ncclResult_t rc_end = c->rccl.GroupEnd();
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) SPMV_FAIL(SPMV_ERR_HIP, "spmv_comm_allgather (rccl): %s", c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "error");
    return SPMV_OK;
}

// Fresh communicators prove themselves before they carry x: participant i writes i + 1 into its slot(s) of a small
// vector, both forms of the all-gather run (equal slices: ncclAllGather; ragged: the broadcasts), and every participant
// must end up with every slot.  A transport that fails this is dropped for peer copies (or reported, if it was forced).
static bool rccl_self_check(spmv_comm* c, std::string& why)
{
    const int n = c->n;
    for (int ragged = 0; ragged < 2; ++ragged)
    {
        std::vector<int64_t> off((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) off[(size_t)i + 1] = off[(size_t)i] + (ragged ? 3 + 2 * i : 4);
        const int64_t           total = off[(size_t)n];
        std::vector<spmv_vec>   store((size_t)n);
        std::vector<spmv_vec*>  vecs((size_t)n, nullptr);
        std::vector<double>     host((size_t)total);
        bool                    ok = true;
        for (int i = 0; i < n && ok; ++i)
        {
            spmv_ctx* x = c->ctx[(size_t)i];
            ok          = hipSetDevice(x->device) == hipSuccess && hipMalloc(&store[(size_t)i].d, sizeof(double) * (size_t)total) == hipSuccess;
            if (!ok) break;
            store[(size_t)i].ctx = x;
            store[(size_t)i].n   = total;
            vecs[(size_t)i]      = &store[(size_t)i];
            for (int64_t k = 0; k < total; ++k) host[(size_t)k] = (k >= off[(size_t)i] && k < off[(size_t)i + 1]) ? (double)(i + 1) : -1.0;
            ok = hipMemcpyAsync(store[(size_t)i].d, host.data(), sizeof(double) * (size_t)total, hipMemcpyHostToDevice, x->stream) == hipSuccess &&
                 hipStreamSynchronize(x->stream) == hipSuccess;
        }
        if (ok && rccl_allgather(c, vecs.data(), off.data(), /*force_broadcasts=*/ragged != 0) != SPMV_OK)
        {
            why = spmv_last_error();
            ok  = false;
        }
        for (int i = 0; i < n && ok; ++i)
        {
            spmv_ctx* x = c->ctx[(size_t)i];
            ok = hipSetDevice(x->device) == hipSuccess &&
                 hipMemcpyAsync(host.data(), store[(size_t)i].d, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, x->stream) == hipSuccess &&
                 hipStreamSynchronize(x->stream) == hipSuccess;
            for (int r = 0; r < n && ok; ++r)
                for (int64_t k = off[(size_t)r]; k < off[(size_t)r + 1] && ok; ++k)
                    if (host[(size_t)k] != (double)(r + 1))
                    {
                        ok  = false;
                        why = "the self-check all-gather delivered wrong data";
                    }
        }
        for (int i = 0; i < n; ++i)
            if (store[(size_t)i].d)
            {
                (void)hipSetDevice(c->ctx[(size_t)i]->device);
                (void)hipFree(store[(size_t)i].d);
            }
        if (!ok)
        {
            if (why.empty()) why = std::string("self-check: ") + hipGetErrorString(hipGetLastError());
            return false;
        }
    }
    return true;
}

extern "C" int spmv_comm_create(spmv_ctx* const* ctxs, int32_t n, spmv_comm** out)
{
    SPMV_REQUIRE(ctxs && out && n >= 1, "spmv_comm_create: bad argument");
    DeviceRestore keep_device;
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
    // Transport.  SPMV_COMM = "peer": copies only; "rccl": RCCL or fail (also with ONE participant, where the default
    // would not bother: that is how a single-GPU box exercises ncclCommInitAll / the group of collectives); unset: RCCL
    // when there are two or more participants, each with a GPU of its own (RCCL wants one device per communicator),
    // and the communicators pass a self-check, else copies.
    const char* want  = getenv("SPMV_COMM");
    const bool  force = want && !strcmp(want, "rccl");
    const bool  never = want && !strcmp(want, "peer");
    if (force && !distinct)
    {
        comm_free(c);
        SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "spmv_comm_create: SPMV_COMM=rccl needs one GPU per participant (RCCL has one communicator per device)");
    }
    if (!never && distinct && (force || n >= 2))
    {
        std::string why;
        if (!c->rccl.load())
            why = "librccl.so could not be loaded";
        else
        {
            std::vector<int> devs((size_t)n);
            for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
            c->nccl.assign((size_t)n, nullptr);
            const ncclResult_t rc = c->rccl.CommInitAll(c->nccl.data(), n, devs.data());
            if (rc != ncclSuccess)
            {
                why = std::string("ncclCommInitAll: ") + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "error");
                c->nccl.clear();
            }
            else
            {
                c->backend = "rccl";
                if (!rccl_self_check(c, why))
                {
                    for (ncclComm_t k : c->nccl)
                        if (k) (void)c->rccl.CommDestroy(k);
                    c->nccl.clear();
                    c->backend = "peer-copy";
                }
            }
        }
        if (force && c->nccl.empty())
        {
            comm_free(c);
            SPMV_FAIL(SPMV_ERR_HIP, "spmv_comm_create: SPMV_COMM=rccl but the RCCL transport is not usable: %s", why.c_str());
        }
        // (not forced: fall back to peer copies; not an error)
    }
    *out = c;
    return SPMV_OK;
}

extern "C" void spmv_comm_destroy(spmv_comm* c)
{
    DeviceRestore keep_device;
    comm_free(c);
}

extern "C" const char* spmv_comm_backend(const spmv_comm* c) { return c ? c->backend.c_str() : ""; }

// copy src[src_offset .. +n) -> dst[dst_offset .. +n) between any two contexts, ordered behind the work already queued
// on the source's stream, queued on the destination's stream
extern "C" int spmv_vec_copy(spmv_vec* dst, int64_t dst_offset, const spmv_vec* src, int64_t src_offset, int64_t n)
{
    SPMV_REQUIRE(dst && src && n >= 0 && dst_offset >= 0 && src_offset >= 0 && dst_offset + n <= dst->n && src_offset + n <= src->n,
                 "spmv_vec_copy: range outside the vectors");
    if (n == 0) return SPMV_OK;
    DeviceRestore keep_device;
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
    DeviceRestore keep_device;
    const int     n     = c->n;
    const int64_t total = offsets[n];
    for (int i = 0; i < n; ++i)
    {
        SPMV_REQUIRE(vecs[i] && vecs[i]->ctx == c->ctx[(size_t)i], "spmv_comm_allgather: vector %d does not belong to participant %d", i, i);
        SPMV_REQUIRE(vecs[i]->n >= total && offsets[i] >= 0 && offsets[i] <= offsets[i + 1],
                     "spmv_comm_allgather: vector %d holds %lld entries, the slices end at %lld", i, (long long)vecs[i]->n, (long long)total);
    }
    if (!c->nccl.empty()) return rccl_allgather(c, vecs, offsets);
    if (n == 1) return SPMV_OK;
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
