// abi.hip — the extern "C" entry points declared in include/spmv_abi.h.
//
// This file is glue: argument checking, handle life cycle, host<->device copies, and dispatch to the
// format kernels (kernels_*.hip), the conversions (convert.hip) and the generators (generate.hip).
// There is deliberately no CPU implementation of anything behind these entry points.
#include <algorithm>
#include <new>
#include <vector>

#include "common.hpp"

namespace spmv
{
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int ensure_scratch(spmv_ctx* ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes) return SPMV_OK;
    if (ctx->scratch)
    {
        SPMV_HIP(hipStreamSynchronize(ctx->stream));
        SPMV_HIP(hipFree(ctx->scratch));
        ctx->scratch       = nullptr;
        ctx->scratch_bytes = 0;
    }
    const size_t want = std::max<size_t>(bytes, 1 << 16);
    SPMV_HIP(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return SPMV_OK;
}

int mat_alloc(spmv_ctx* ctx, int32_t format, int32_t nrow, int32_t ncol, int64_t nnz, int32_t k, size_t a_count,
              size_t b_count, size_t v_count, spmv_mat** out)
{
    spmv_mat* m = new (std::nothrow) spmv_mat();
    if (!m) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    m->ctx    = ctx;
    m->format = format;
    m->nrow   = nrow;
    m->ncol   = ncol;
    m->nnz    = nnz;
    m->k      = k;
    m->owned  = true;
    void *a = nullptr, *b = nullptr, *v = nullptr;
    hipError_t e = hipSuccess;
    // a zero-length array still gets a valid (tiny) allocation so that kernels never see nullptr
    if (e == hipSuccess) e = hipMalloc(&a, std::max<size_t>(a_count, 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&b, std::max<size_t>(b_count, 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&v, std::max<size_t>(v_count, 1) * sizeof(double));
    if (e != hipSuccess)
    {
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (v) (void)hipFree(v);
        delete m;
        SPMV_FAIL(SPMV_ERR_ALLOC, "device allocation of %zu+%zu int32 and %zu fp64 failed: %s", a_count, b_count,
                  v_count, hipGetErrorString(e));
    }
    m->a            = (const int32_t*)a;
    m->b            = (const int32_t*)b;
    m->v            = (const double*)v;
    m->device_bytes = (int64_t)((a_count + b_count) * sizeof(int32_t) + v_count * sizeof(double));
    *out            = m;
    return SPMV_OK;
}

void mat_free(spmv_mat* m)
{
    if (!m) return;
    if (m->owned)
    {
        if (m->a) (void)hipFree(const_cast<int32_t*>(m->a));
        if (m->b) (void)hipFree(const_cast<int32_t*>(m->b));
        if (m->v) (void)hipFree(const_cast<double*>(m->v));
    }
    if (m->win_lo) (void)hipFree(m->win_lo);
    if (m->win_span) (void)hipFree(m->win_span);
    if (m->ell_diag) (void)hipFree(m->ell_diag);
    if (m->ell_diag_mask) (void)hipFree(m->ell_diag_mask);
    ell_free_tiles(m);
    ell_free_dia_order(m);
    csr_panel_free(m);
    csr_twophase_free(m);
    csr_segscan_free(m);
    csr_split_free(m);
    csr_ell_copy_free(m);
    symgs_free(m);
    if (m->coo_csr) mat_free(m->coo_csr);
    coo_free_bins(m);
    delete m;
}

static int use_device(const spmv_ctx* ctx)
{
    SPMV_HIP(hipSetDevice(ctx->device));
    return SPMV_OK;
}

static int upload(void* dst, const void* src, size_t bytes, spmv_ctx* ctx)
{
    if (bytes == 0) return SPMV_OK;
    SPMV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));  // pageable host memory: the caller may free it on return
    return SPMV_OK;
}

static int wrap(spmv_ctx* ctx, int32_t format, int32_t nrow, int32_t ncol, int64_t nnz, int32_t k, const int32_t* a,
                const int32_t* b, const double* v, spmv_mat** out)
{
    spmv_mat* m = new (std::nothrow) spmv_mat();
    if (!m) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    m->ctx    = ctx;
    m->format = format;
    m->nrow   = nrow;
    m->ncol   = ncol;
    m->nnz    = nnz;
    m->k      = k;
    m->a      = a;
    m->b      = b;
    m->v      = v;
    m->owned  = false;
    *out      = m;
    return SPMV_OK;
}

// spmv_ctx_set_plan: the handle a public entry point is about to create takes the context's plan (plan.hip: plan_take_armed, at
// the handle's analysis); whatever the entry point does, the flag is down again when it returns
struct plan_arm
{
    spmv_ctx* c;
    explicit plan_arm(spmv_ctx* ctx) : c(ctx)
    {
        if (c && !c->plan_blob.empty()) c->plan_armed = true;
    }
    ~plan_arm()
    {
        if (c) c->plan_armed = false;
    }
};

static int finish(spmv_mat* m, spmv_mat** out)
{
    // The analysis and layout builders index by row and column (histograms in LDS, scatter cursors): a malformed
    // matrix must be refused before they run — an out-of-bounds access on the GPU can take the whole node down.
    int rc = mat_validate(m);
    if (rc == SPMV_OK && m->format == SPMV_FMT_CSR) rc = csr_analyse(m);
    if (rc == SPMV_OK && m->format == SPMV_FMT_COO) rc = coo_analyse(m);
    if (rc == SPMV_OK && m->format == SPMV_FMT_CSC) rc = csc_analyse(m);
    if (rc == SPMV_OK && m->format == SPMV_FMT_ELL) rc = ell_analyse(m);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    *out = m;
    return SPMV_OK;
}
// Does a product of this handle add into y with device atomics (global_atomic_add_f64)?  Decided by the kernel that RUNS,
// followed through the copies a handle may run from: the COO scan (in place or over column bins) and the CSC scatter; CSR
// handles under SPMV_CSR_SEGSCAN (the same scan over a row index per entry) and SPMV_CSR_SPLIT in chunk mode (one atomic add
// per chunk of a long row; the virtual-row mode adds its partial sums up in a scratch vector of its own and onto y with a plain
// read and store); and any handle whose row-grouped copy, short-row copy or ELL copy runs one of those.  Everything else
// touches every y_i once with a plain read and a plain store.  spmv_apply_host decides from this where y may live.
bool adds_into_y_with_atomics(const spmv_mat* A)
{
    if (!A) return false;
    switch (A->format)
    {
        case SPMV_FMT_COO:
        case SPMV_FMT_CSC:
            return A->coo_csr && A->kernel == SPMV_CSR_PANEL ? adds_into_y_with_atomics(A->coo_csr) : true;
        case SPMV_FMT_ELL: return A->coo_csr && A->kernel == SPMV_CSR_PANEL ? adds_into_y_with_atomics(A->coo_csr) : false;
        case SPMV_FMT_CSR:
            switch (A->kernel)
            {
                case SPMV_CSR_SEGSCAN: return true;
                case SPMV_CSR_SPLIT:
                    // the short rows' copy picks a kernel of its own; long rows: chunks add atomically, virtual rows do not
                    return adds_into_y_with_atomics(A->coo_csr) || (!A->split_long && A->split_nchunks > 0);
                case SPMV_CSR_ELL: return adds_into_y_with_atomics(A->ell_copy);
                default: return false;
            }
        default: return false;  // DIA
    }
}

}  // namespace spmv

using namespace spmv;

extern "C" {

int         spmv_abi_version(void) { return SPMV_ABI_VERSION; }
const char* spmv_last_error(void) { return g_last_error; }

int spmv_device_count(int* count)
{
    SPMV_REQUIRE(count, "count is null");
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
    {
        *count = 0;
        SPMV_FAIL(SPMV_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SPMV_OK;
}

static void xcd_probe(spmv_ctx* ctx);
static bool host_stores_reach_kernels(spmv_ctx* ctx);
static int ctx_create_common(int device, void* borrowed_stream, bool borrow, spmv_ctx** out)
{
    SPMV_REQUIRE(out, "out is null");
    *out  = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        SPMV_FAIL(SPMV_ERR_NO_DEVICE, "no HIP device is visible; libspmv_hip has no CPU fallback");
    if (device < 0 || device >= n) SPMV_FAIL(SPMV_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    SPMV_HIP(hipSetDevice(device));
    spmv_ctx* ctx = new (std::nothrow) spmv_ctx();
    if (!ctx) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    ctx->device = device;
    if (borrow)
    {
        ctx->stream      = (hipStream_t)borrowed_stream;
        ctx->owns_stream = false;
    }
    else
    {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            delete ctx;
            SPMV_FAIL(SPMV_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->owns_stream = true;
    }
    hipError_t e = hipEventCreate(&ctx->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_end);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->host_pinned, 64, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->dev_scalars, sizeof(double) * kDotDoubles);
    if (e != hipSuccess)
    {
        spmv_ctx_destroy(ctx);
        SPMV_FAIL(SPMV_ERR_HIP, "context setup: %s", hipGetErrorString(e));
    }
    xcd_probe(ctx);  // one small launch: do workgroups b and b + 8 share an XCD on this device? (see below)
    {
        int large = 0;
        if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large)
        {
            unsigned* reg = nullptr;
            if (hipDeviceGetAttribute((int*)&reg, hipDeviceAttributeHdpMemFlushCntl, device) == hipSuccess) ctx->hdp_flush = reg;
            const char* off = getenv("SPMV_HOST_STORES");  // SPMV_HOST_STORES=0: never store into device memory from the CPU (A/B; read once per context)
            ctx->large_bar  = !(off && off[0] == '0');
            // ... and only where this context has SEEN it work: CPU stores into a device buffer, read by a kernel, twice (the second
            // round right after a kernel read the first contents - the case a stale cache line would get wrong)
            if (ctx->large_bar && !host_stores_reach_kernels(ctx)) ctx->large_bar = 0;
        }
        (void)hipGetLastError();
    }
    *out = ctx;
    return SPMV_OK;
}

// ---- where do the workgroups of a launch land? --------------------------------------------------------------------------------
// HIP promises nothing about workgroup -> XCD placement; what is OBSERVED on MI355X is round-robin dealing (workgroup b on XCD
// (b + c) % 8 for some c that changes from launch to launch).  coo_segscan_bins_kernel (workgroup w scans bin w % 8) and the
// panel kernel's per-XCD bookkeeping lean on "b and b + 8 share an XCD" for speed.  This probe checks it once per context:
// 2048 workgroups write the XCC_ID hardware register (bits 3:0 of hwreg 20 on gfx942 / gfx950); the host requires that equal
// labels b % 8 saw ONE id each and different labels different ids.
namespace
{
__global__ void xcd_probe_kernel(int* __restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // GETREG_IMMED(size - 1, offset, XCC_ID)
}
}  // namespace

static void xcd_probe(spmv_ctx* ctx)
{
    constexpr int kProbeBlocks = 2048;
    ctx->xcd_round_robin = -1;
    if (ensure_scratch(ctx, sizeof(int) * kProbeBlocks) != SPMV_OK) return;
    int* d = (int*)ctx->scratch;
    std::vector<int> h(kProbeBlocks, -1);
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(kProbeBlocks), dim3(64), 0, ctx->stream, d);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h.data(), d, sizeof(int) * kProbeBlocks, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
    {
        (void)hipGetLastError();
        return;
    }
    int  id_of_label[8];
    bool ok = true;
    unsigned seen = 0;
    for (int b = 0; b < kProbeBlocks; ++b)
    {
        seen |= 1u << (h[b] & 15);
        if (b < 8)
            id_of_label[b] = h[b];
        else if (h[b] != id_of_label[b & 7])
            ok = false;
    }
    for (int a = 0; a < 8 && ok; ++a)
        for (int b = a + 1; b < 8; ++b)
            if (id_of_label[a] == id_of_label[b]) ok = false;
    ctx->xcds_seen       = __builtin_popcount(seen);
    ctx->xcd_round_robin = ok ? 1 : 0;
}

// ---- may the CPU store x straight into device memory? --------------------------------------------------------------------------
// spmv_apply_host writes a small x into its device buffer with memcpy where the device reports a large BAR.  The attribute says
// the memory is mapped; that a KERNEL sees such stores (write-combining buffers drained, HDP flushed, no stale line in L2) is
// checked here once per context before the path is switched on: 512 doubles stored by the CPU, summed by a kernel, twice with
// different contents.  A mismatch (or any error) leaves large_bar off and x goes through the pinned staging buffer instead.
namespace
{
__global__ void host_store_probe_kernel(const double* __restrict__ in, int n, double* __restrict__ out)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += in[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) *out = s;
}
}  // namespace

static bool host_stores_reach_kernels(spmv_ctx* ctx)
{
    constexpr int kN = 512;
    double* buf = nullptr;
    if (hipMalloc((void**)&buf, sizeof(double) * kN) != hipSuccess)
    {
        (void)hipGetLastError();
        return false;
    }
    bool   ok = true;
    double host[kN];
    for (int round = 0; round < 2 && ok; ++round)
    {
        double want = 0.0;
        for (int i = 0; i < kN; ++i)
        {
            host[i] = (double)((i * 7 + round * 13) % 32) + (round ? 0.5 : 0.25);  // (sums of these are exact in fp64 in any order)
            want += host[i];
        }
        memcpy(buf, host, sizeof(host));
        __sync_synchronize();
        if (ctx->hdp_flush) *ctx->hdp_flush = 1u;
        hipLaunchKernelGGL(host_store_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, buf, kN, ctx->dev_scalars);
        double got = -1.0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&got, ctx->dev_scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
        {
            (void)hipGetLastError();
            ok = false;
        }
        else
            ok = got == want;
    }
    (void)hipFree(buf);
    return ok;
}

int spmv_ctx_xcd_round_robin(spmv_ctx* ctx, int32_t* round_robin, int32_t* xcds_seen)
{
    SPMV_REQUIRE(ctx, "ctx is null");
    if (round_robin) *round_robin = ctx->xcd_round_robin;
    if (xcds_seen) *xcds_seen = ctx->xcds_seen;
    return SPMV_OK;
}

int spmv_ctx_get_param(const spmv_ctx* ctx, const char* name, int64_t* value)
{
    SPMV_REQUIRE(ctx && name && value, "spmv_ctx_get_param: null argument");
    if (!strcmp(name, "host_stores"))
        *value = ctx->large_bar ? 1 : 0;
    else if (!strcmp(name, "xcd_round_robin"))
        *value = ctx->xcd_round_robin;
    else if (!strcmp(name, "xcds_seen"))
        *value = ctx->xcds_seen;
    else if (!strcmp(name, "trial_arena_bytes"))
        *value = (int64_t)ctx->arena_bytes;
    else
        SPMV_FAIL(SPMV_ERR_INVALID, "unknown context parameter '%s'", name);
    return SPMV_OK;
}

int spmv_ctx_create(int device, spmv_ctx** out) { return ctx_create_common(device, nullptr, false, out); }
int spmv_ctx_create_on_stream(int device, void* hip_stream, spmv_ctx** out)
{
    return ctx_create_common(device, hip_stream, true, out);
}

int spmv_ctx_destroy(spmv_ctx* ctx)
{
    if (!ctx) return SPMV_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->stage_x) (void)hipFree(ctx->stage_x);
    if (ctx->stage_y) (void)hipFree(ctx->stage_y);
    if (ctx->stage_pinned) (void)hipHostFree(ctx->stage_pinned);
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->dev_scalars) (void)hipFree(ctx->dev_scalars);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SPMV_OK;
}

int spmv_ctx_mem_info(spmv_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes)
{
    SPMV_REQUIRE(ctx && free_bytes && total_bytes, "spmv_ctx_mem_info: null argument");
    SPMV_HIP(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    SPMV_HIP(hipMemGetInfo(&f, &t));
    *free_bytes  = (int64_t)f;
    *total_bytes = (int64_t)t;
    return SPMV_OK;
}

int spmv_sync(spmv_ctx* ctx)
{
    SPMV_REQUIRE(ctx, "ctx is null");
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    return SPMV_OK;
}

int spmv_ctx_device(const spmv_ctx* ctx, int* device)
{
    SPMV_REQUIRE(ctx && device, "null argument");
    *device = ctx->device;
    return SPMV_OK;
}

// ---- vectors ----------------------------------------------------------------------------------------------
int spmv_vec_create(spmv_ctx* ctx, int64_t n, spmv_vec** out)
{
    SPMV_REQUIRE(ctx && out && n >= 0, "spmv_vec_create: bad argument");
    SPMV_TRY(use_device(ctx));
    spmv_vec* v = new (std::nothrow) spmv_vec();
    if (!v) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    hipError_t e = hipMalloc((void**)&v->d, std::max<int64_t>(n, 1) * sizeof(double));
    if (e != hipSuccess)
    {
        delete v;
        SPMV_FAIL(SPMV_ERR_ALLOC, "device allocation of %lld fp64 failed: %s", (long long)n, hipGetErrorString(e));
    }
    v->ctx   = ctx;
    v->n     = n;
    v->owned = true;
    *out     = v;
    return SPMV_OK;
}

int spmv_vec_wrap_device(spmv_ctx* ctx, int64_t n, double* device_ptr, spmv_vec** out)
{
    SPMV_REQUIRE(ctx && out && n >= 0 && (device_ptr || n == 0), "spmv_vec_wrap_device: bad argument");
    spmv_vec* v = new (std::nothrow) spmv_vec();
    if (!v) SPMV_FAIL(SPMV_ERR_ALLOC, "out of host memory");
    v->ctx   = ctx;
    v->n     = n;
    v->d     = device_ptr;
    v->owned = false;
    *out     = v;
    return SPMV_OK;
}

int spmv_vec_destroy(spmv_vec* v)
{
    if (!v) return SPMV_OK;
    if (v->owned && v->d)
    {
        (void)hipSetDevice(v->ctx->device);
        (void)hipFree(v->d);
    }
    delete v;
    return SPMV_OK;
}

int spmv_vec_size(const spmv_vec* v, int64_t* n)
{
    SPMV_REQUIRE(v && n, "null argument");
    *n = v->n;
    return SPMV_OK;
}

int spmv_vec_device_ptr(const spmv_vec* v, double** device_ptr)
{
    SPMV_REQUIRE(v && device_ptr, "null argument");
    *device_ptr = v->d;
    return SPMV_OK;
}

int spmv_vec_upload(spmv_vec* v, int64_t offset, int64_t n, const double* host)
{
    SPMV_REQUIRE(v && offset >= 0 && n >= 0 && offset + n <= v->n && (host || n == 0),
                 "spmv_vec_upload: range [%lld,+%lld) outside vector of %lld", (long long)offset, (long long)n,
                 v ? (long long)v->n : -1LL);
    SPMV_TRY(use_device(v->ctx));
    return upload(v->d + offset, host, (size_t)n * sizeof(double), v->ctx);
}

int spmv_vec_download(const spmv_vec* v, int64_t offset, int64_t n, double* host)
{
    SPMV_REQUIRE(v && offset >= 0 && n >= 0 && offset + n <= v->n && (host || n == 0),
                 "spmv_vec_download: range [%lld,+%lld) outside vector of %lld", (long long)offset, (long long)n,
                 v ? (long long)v->n : -1LL);
    if (n == 0) return SPMV_OK;
    SPMV_TRY(use_device(v->ctx));
    SPMV_HIP(hipMemcpyAsync(host, v->d + offset, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
    SPMV_HIP(hipStreamSynchronize(v->ctx->stream));
    return SPMV_OK;
}

int spmv_vec_fill(spmv_vec* v, double a)
{
    SPMV_REQUIRE(v, "null vector");
    SPMV_TRY(use_device(v->ctx));
    return vec_fill(v->ctx, v->d, v->n, a);
}

// ---- matrices ---------------------------------------------------------------------------------------------
int spmv_csr_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* row_ptr, const int32_t* col_ind,
                    const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && row_ptr, "spmv_csr_upload: bad argument");
    const int64_t nnz = (int64_t)row_ptr[nrow] - row_ptr[0];
    SPMV_REQUIRE(row_ptr[0] == 0 && nnz >= 0 && (nnz == 0 || (col_ind && values)),
                 "spmv_csr_upload: row_ptr[0]=%d, nnz=%lld", row_ptr[0], (long long)nnz);
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSR, nrow, ncol, nnz, 0, (size_t)nrow + 1, (size_t)nnz, (size_t)nnz, &m));
    int rc = upload(const_cast<int32_t*>(m->a), row_ptr, sizeof(int32_t) * ((size_t)nrow + 1), ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<int32_t*>(m->b), col_ind, sizeof(int32_t) * (size_t)nnz, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values, sizeof(double) * (size_t)nnz, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    return finish(m, out);
}

int spmv_csr_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* d_row_ptr, const int32_t* d_col_ind,
                         const double* d_values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && d_row_ptr, "spmv_csr_wrap_device: bad argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    int32_t ends[2] = {0, 0};
    SPMV_HIP(hipMemcpyAsync(&ends[0], d_row_ptr, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipMemcpyAsync(&ends[1], d_row_ptr + nrow, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    SPMV_REQUIRE(ends[0] == 0 && ends[1] >= 0, "spmv_csr_wrap_device: row_ptr[0]=%d row_ptr[nrow]=%d", ends[0], ends[1]);
    spmv_mat* m = nullptr;
    SPMV_TRY(wrap(ctx, SPMV_FMT_CSR, nrow, ncol, ends[1], 0, d_row_ptr, d_col_ind, d_values, &m));
    return finish(m, out);
}

int spmv_csr_upload_shard(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol, const int64_t* row_ptr64,
                          const int32_t* col_ind, const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && row_ptr64 && row_begin >= 0 && row_end >= row_begin && row_end - row_begin <= INT32_MAX,
                 "spmv_csr_upload_shard: bad row range");
    const int32_t nrow = (int32_t)(row_end - row_begin);
    const int64_t base = row_ptr64[row_begin];
    const int64_t nnz  = row_ptr64[row_end] - base;
    SPMV_REQUIRE(nnz >= 0 && nnz <= INT32_MAX, "spmv_csr_upload_shard: shard holds %lld entries (int32 offsets)",
                 (long long)nnz);
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    // rebase exactly like src/mat_vec.cpp:260-263
    std::vector<int32_t> sub((size_t)nrow + 1);
    for (int32_t j = 0; j <= nrow; ++j) sub[j] = (int32_t)(row_ptr64[row_begin + j] - base);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSR, nrow, ncol, nnz, 0, (size_t)nrow + 1, (size_t)nnz, (size_t)nnz, &m));
    int rc = upload(const_cast<int32_t*>(m->a), sub.data(), sizeof(int32_t) * sub.size(), ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<int32_t*>(m->b), col_ind + base, sizeof(int32_t) * (size_t)nnz, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values + base, sizeof(double) * (size_t)nnz, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    m->row_begin = row_begin;
    return finish(m, out);
}

int spmv_coo_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int64_t nnz, const int32_t* row_ind,
                    const int32_t* col_ind, const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && nnz >= 0 && (nnz == 0 || (row_ind && col_ind && values)),
                 "spmv_coo_upload: bad argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_COO, nrow, ncol, nnz, 0, (size_t)nnz, (size_t)nnz, (size_t)nnz, &m));
    int rc = upload(const_cast<int32_t*>(m->a), row_ind, sizeof(int32_t) * (size_t)nnz, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<int32_t*>(m->b), col_ind, sizeof(int32_t) * (size_t)nnz, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values, sizeof(double) * (size_t)nnz, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    return finish(m, out);
}

int spmv_coo_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int64_t nnz, const int32_t* d_row_ind,
                         const int32_t* d_col_ind, const double* d_values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && nnz >= 0, "spmv_coo_wrap_device: bad argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(wrap(ctx, SPMV_FMT_COO, nrow, ncol, nnz, 0, d_row_ind, d_col_ind, d_values, &m));
    return finish(m, out);
}

int spmv_ell_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, int64_t nnz, const int32_t* col_ind,
                    const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && k >= 0, "spmv_ell_upload: bad argument");
    const size_t total = (size_t)nrow * (size_t)k;
    SPMV_REQUIRE(total == 0 || (col_ind && values), "spmv_ell_upload: null arrays");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_ELL, nrow, ncol, nnz, k, 0, total, total, &m));
    int rc = upload(const_cast<int32_t*>(m->b), col_ind, sizeof(int32_t) * total, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values, sizeof(double) * total, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    m->max_row_nnz = k;
    return finish(m, out);
}

int spmv_ell_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, int64_t nnz, const int32_t* d_col_ind,
                         const double* d_values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && k >= 0, "spmv_ell_wrap_device: bad argument");
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(wrap(ctx, SPMV_FMT_ELL, nrow, ncol, nnz, k, nullptr, d_col_ind, d_values, &m));
    m->max_row_nnz = k;
    return finish(m, out);
}

int spmv_csc_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* col_ptr, const int32_t* row_ind,
                    const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && col_ptr, "spmv_csc_upload: bad argument");
    const int64_t nnz = (int64_t)col_ptr[ncol];
    SPMV_REQUIRE(col_ptr[0] == 0 && nnz >= 0 && (nnz == 0 || (row_ind && values)), "spmv_csc_upload: bad col_ptr");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSC, nrow, ncol, nnz, 0, (size_t)ncol + 1, (size_t)nnz, (size_t)nnz, &m));
    int rc = upload(const_cast<int32_t*>(m->a), col_ptr, sizeof(int32_t) * ((size_t)ncol + 1), ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<int32_t*>(m->b), row_ind, sizeof(int32_t) * (size_t)nnz, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values, sizeof(double) * (size_t)nnz, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    return finish(m, out);
}

int spmv_dia_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t ndiags, const int32_t* offsets,
                    const double* values, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out && nrow >= 0 && ncol >= 0 && ndiags >= 0, "spmv_dia_upload: bad argument");
    const size_t total = (size_t)nrow * (size_t)ndiags;
    SPMV_REQUIRE(ndiags == 0 || offsets, "spmv_dia_upload: null offsets");
    SPMV_REQUIRE(total == 0 || values, "spmv_dia_upload: null values");
    SPMV_TRY(use_device(ctx));
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_DIA, nrow, ncol, 0, ndiags, (size_t)ndiags, 0, total, &m));
    if (ndiags > 0)
    {
        m->dia_off_known = true;
        m->dia_off_min   = *std::min_element(offsets, offsets + ndiags);
        m->dia_off_max   = *std::max_element(offsets, offsets + ndiags);
    }
    int rc = upload(const_cast<int32_t*>(m->a), offsets, sizeof(int32_t) * (size_t)ndiags, ctx);
    if (rc == SPMV_OK) rc = upload(const_cast<double*>(m->v), values, sizeof(double) * total, ctx);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    return finish(m, out);
}

int spmv_mat_destroy(spmv_mat* m)
{
    if (!m) return SPMV_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    mat_free(m);
    return SPMV_OK;
}

int spmv_mat_validate(const spmv_mat* m)
{
    SPMV_REQUIRE(m, "null matrix");
    SPMV_TRY(use_device(m->ctx));
    return mat_validate(m);
}

int spmv_mat_get_info(const spmv_mat* m, spmv_mat_info* info)
{
    SPMV_REQUIRE(m && info, "null argument");
    info->format        = m->format;
    info->nrow          = m->nrow;
    info->ncol          = m->ncol;
    info->ell_k         = m->k;
    info->nnz           = m->nnz;
    info->row_begin     = m->row_begin;
    info->max_row_nnz   = m->max_row_nnz;
    info->kernel        = m->kernel;
    info->lanes_per_row = m->lanes_per_row;
    info->sorted_rows   = m->sorted_rows;
    info->device_bytes  = m->device_bytes;
    return SPMV_OK;
}

int spmv_mat_set_kernel(spmv_mat* m, int32_t kernel, int32_t lanes_per_row)
{
    SPMV_REQUIRE(m, "null matrix");
    SPMV_REQUIRE(kernel >= SPMV_CSR_AUTO && kernel <= SPMV_CSR_ELL, "unknown kernel id %d", kernel);
    SPMV_REQUIRE(lanes_per_row == 0 || (lanes_per_row >= 1 && lanes_per_row <= 64 &&
                                        (lanes_per_row & (lanes_per_row - 1)) == 0),
                 "lanes_per_row must be 0 or a power of two in 1..64, got %d", lanes_per_row);
    if (lanes_per_row > 0) m->lanes_per_row = lanes_per_row;
    if (m->format == SPMV_FMT_COO)
    {
        // COO: AUTO = panel layout when it pays, VECTOR = the segmented scan, PANEL = build the panel layout now
        SPMV_REQUIRE(kernel == SPMV_CSR_AUTO || kernel == SPMV_CSR_VECTOR || kernel == SPMV_CSR_PANEL,
                     "COO handles take kernel AUTO (0), VECTOR (1: segmented scan) or PANEL (4), got %d", kernel);
        SPMV_HIP(hipSetDevice(m->ctx->device));
        m->kernel_forced = kernel != SPMV_CSR_AUTO;
        if (kernel == SPMV_CSR_VECTOR)
        {
            // the scan runs over a copy of the entries in column bins when x is beyond an XCD's L2 ("coo_column_bins" = 0 drops it)
            m->kernel = SPMV_CSR_VECTOR;
            if (!m->cb_bins)
            {
                const int rc = coo_build_bins(m, 0, /*only_if_worth=*/true);
                if (rc != SPMV_OK && rc != SPMV_ERR_ALLOC) return rc;  // (no room for the copy: the scan runs over the handle's own arrays)
                (void)hipGetLastError();
            }
        }
        else
        {
            if (kernel == SPMV_CSR_AUTO)
                SPMV_TRY(coo_select_kernel(m));  // the scan or the row-grouped copy, timed (select.hip)
            else
                SPMV_TRY(coo_build_panel(m, /*only_if_worth=*/false));
            m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
            if (m->kernel == SPMV_CSR_PANEL)
            {
                SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
                coo_free_bins(m);
            }
        }
        return SPMV_OK;
    }
    if (m->format == SPMV_FMT_CSC)
    {
        // CSC: AUTO = the scatter or the row-grouped copy, timed; VECTOR = the scatter over the columns; PANEL = regroup now, panel layout
        SPMV_REQUIRE(kernel == SPMV_CSR_AUTO || kernel == SPMV_CSR_VECTOR || kernel == SPMV_CSR_PANEL,
                     "CSC handles take kernel AUTO (0), VECTOR (1: scatter over the columns) or PANEL (4), got %d", kernel);
        SPMV_HIP(hipSetDevice(m->ctx->device));
        m->kernel_forced = kernel != SPMV_CSR_AUTO;
        if (kernel == SPMV_CSR_VECTOR)
            m->kernel = SPMV_CSR_VECTOR;
        else if (kernel == SPMV_CSR_AUTO)
            SPMV_TRY(csc_select_kernel(m));
        else
        {
            SPMV_TRY(csc_build_rowgrouped(m, SPMV_CSR_PANEL));
            m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
        }
        return SPMV_OK;
    }
    if (m->format == SPMV_FMT_ELL)
    {
        // ELL: AUTO = panel layout when the columns are scattered, VECTOR = one lane per row, PANEL = build it now
        SPMV_REQUIRE(kernel == SPMV_CSR_AUTO || kernel == SPMV_CSR_VECTOR || kernel == SPMV_CSR_PANEL,
                     "ELL handles take kernel AUTO (0), VECTOR (1: one lane per row) or PANEL (4), got %d", kernel);
        SPMV_HIP(hipSetDevice(m->ctx->device));
        m->kernel_forced = kernel != SPMV_CSR_AUTO;
        if (kernel == SPMV_CSR_VECTOR)
        {
            m->kernel      = SPMV_CSR_VECTOR;
            m->ell_variant = 0;  // lanes_per_row (below) picks the variant of the format's own kernel
            if (m->ell_dia_order_req < 0 && m->ell_rval)
            {
                SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
                ell_free_dia_order(m);  // (a DIA-order copy the trial had kept: 8 bytes per slot nobody multiplies from now)
            }
        }
        else if (kernel == SPMV_CSR_AUTO)
            SPMV_TRY(ell_select_kernel(m));  // the format's own kernels and, where it is a candidate, the row-grouped copy: timed
        else
        {
            SPMV_TRY(ell_build_panel(m, /*only_if_worth=*/false));
            m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
        }
        return SPMV_OK;
    }
    if (m->format == SPMV_FMT_CSR && m->nnz > 0 && (!m->b || !m->v))
        SPMV_REQUIRE((kernel == SPMV_CSR_PANEL && m->pb_val) || (kernel == SPMV_CSR_TWOPHASE && m->tp_val) || (kernel == SPMV_CSR_ELL && m->ell_copy) ||
                         (kernel == SPMV_CSR_AUTO && (m->kernel == SPMV_CSR_PANEL || m->kernel == SPMV_CSR_TWOPHASE || m->kernel == SPMV_CSR_ELL)),
                     "this handle gave up its CSR arrays (panel_keep_csr = 0): only the product it was built for is left");
    if (kernel == SPMV_CSR_AUTO)
    {
        m->kernel_forced = false;
        if (m->format == SPMV_FMT_CSR && m->b && m->v)
        {
            SPMV_HIP(hipSetDevice(m->ctx->device));
            return csr_select_kernel(m);  // the model and, where it pays, a trial of the candidates (select.hip); builds what it picks
        }
    }
    else
    {
        m->kernel         = kernel;
        m->kernel_forced  = true;
        m->split_auto_low = false;  // (a forced SPLIT takes "split_row_threshold" or its default, not what AUTO found)
    }
    if (m->kernel != SPMV_CSR_SEGSCAN) csr_segscan_free(m);
    if (m->format == SPMV_FMT_CSR && m->kernel != SPMV_CSR_SPLIT) csr_split_free(m);
    if (m->kernel != SPMV_CSR_ELL) csr_ell_copy_free(m);
    if (m->format == SPMV_FMT_CSR && m->kernel == SPMV_CSR_PANEL)
    {
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_panel_build(m));  // (re)build with the current parameters
    }
    if (m->format == SPMV_FMT_CSR && m->kernel == SPMV_CSR_TWOPHASE)
    {
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_twophase_build(m));
    }
    if (m->kernel == SPMV_CSR_SEGSCAN)
    {
        SPMV_REQUIRE(m->format == SPMV_FMT_CSR, "kernel SEGSCAN (6) is a CSR kernel (a COO handle's VECTOR is the same scan)");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_segscan_build(m));
    }
    if (m->kernel == SPMV_CSR_SPLIT)
    {
        SPMV_REQUIRE(m->format == SPMV_FMT_CSR, "kernel SPLIT (7) is a CSR kernel");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_split_build(m));
    }
    if (m->kernel == SPMV_CSR_ELL)
    {
        SPMV_REQUIRE(m->format == SPMV_FMT_CSR, "kernel ELL (8) is a CSR kernel: the ELL copy of a CSR handle");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_ell_copy_build(m));
    }
    return SPMV_OK;
}

int spmv_mat_set_param(spmv_mat* m, const char* name, int64_t value)
{
    SPMV_REQUIRE(m && name, "null argument");
    if (!strcmp(name, "panel_rows"))
        m->pb_group_rows = (int32_t)value;
    else if (!strcmp(name, "panel_width"))
        m->pb_panel_width = (int32_t)value;
    else if (!strcmp(name, "panel_sort"))
        m->pb_sort = (int32_t)value;
    else if (!strcmp(name, "panel_unroll"))
        m->pb_unroll = (int32_t)value;
    else if (!strcmp(name, "panel_aos"))
        m->pb_aos = (int32_t)value;
    else if (!strcmp(name, "panel_pipe"))
        m->pb_pipe = (int32_t)value;
    else if (!strcmp(name, "panel_two_per_cu"))
        m->pb_two_per_cu = (int32_t)value;
    else if (!strcmp(name, "panel_rounds"))
    {
        SPMV_REQUIRE(value >= 0 && value <= 16, "panel_rounds must be 0 (automatic), 1 (the fewest groups) or a multiple up to 16, got %lld", (long long)value);
        m->pb_rounds_req = (int32_t)value;
    }
    else if (!strcmp(name, "split_row_threshold"))
    {
        // rows of this many entries and more are "long" under kernel SPLIT (0: the default); takes effect at the next spmv_mat_set_kernel
        SPMV_REQUIRE(value >= 0 && value <= INT32_MAX, "split_row_threshold must be 0 (default) or a row length, got %lld", (long long)value);
        m->split_threshold = (int32_t)value;
    }
    else if (!strcmp(name, "split_mode"))
    {
        // how kernel SPLIT runs its long rows: 1 chunks of 4096 entries over the handle's own arrays, 2 virtual rows of 64 entries in a
        // matrix of their own, 0 by their density (kernels_csr_split.hip); takes effect at the next spmv_mat_set_kernel
        SPMV_REQUIRE(value >= 0 && value <= 2, "split_mode must be 0 (by density), 1 (chunks) or 2 (virtual rows), got %lld", (long long)value);
        m->split_mode = (int32_t)value;
    }
    else if (!strcmp(name, "panel_keep_csr"))
    {
        // 0: release col_ind / values of a CSR handle whose product runs from the panel layout (which holds the same
        // entries re-ordered; row_ptr stays).  Memory goes from 2x to 1x the matrix; what needs the arrays afterwards
        // (download, another kernel, a re-build with other parameters, conversions, the Jacobi diagonal) is refused.
        SPMV_REQUIRE(value == 0 || (m->b && m->v) || m->nnz == 0, "panel_keep_csr: the arrays are gone already");
        if (value == 0 && m->b && m->v)
        {
            SPMV_REQUIRE(m->format == SPMV_FMT_CSR && m->owned &&
                             ((m->kernel == SPMV_CSR_PANEL && m->pb_val) || (m->kernel == SPMV_CSR_TWOPHASE && m->tp_val) ||
                              (m->kernel == SPMV_CSR_ELL && m->ell_copy)),
                         "panel_keep_csr = 0 needs an owned CSR handle whose panel or two-phase layout or ELL copy is built");
            SPMV_HIP(hipSetDevice(m->ctx->device));
            SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
            // a two-phase handle first offers the gigabytes it is about to release to its product stream's piece search
            // (kernels_csr_twophase.hip: memory of another moment of the allocator's history, at no transient cost)
            bool keep_b = false, keep_v = false;
            if (m->kernel == SPMV_CSR_TWOPHASE) SPMV_TRY(csr_twophase_offer_csr_copy(m, &keep_b, &keep_v));
            if (!keep_b) (void)hipFree(const_cast<int32_t*>(m->b));
            if (!keep_v) (void)hipFree(const_cast<double*>(m->v));
            m->b = nullptr;
            m->v = nullptr;
            m->device_bytes -= m->nnz * 12;
        }
    }
    else if (!strcmp(name, "ell_tiled_values"))
    {
        // ELL whose slots are diagonals: 1 = keep a copy of the values in tiles of 512 rows for the product (never made unasked:
        // 8 bytes per slot for 1-7 %), 0 = drop it and multiply from the column-major array
        SPMV_REQUIRE(m->format == SPMV_FMT_ELL && (value == 0 || value == 1), "ell_tiled_values: an ELL handle and 0 or 1");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
        if (value == 0)
            ell_free_tiles(m);
        else
            SPMV_TRY(ell_build_tiles(m, /*only_if_worth=*/false));
    }
    else if (!strcmp(name, "ell_dia_order"))
    {
        // ELL whose slots are diagonals: 1 = keep the values once more in DIA order (row-major) and multiply with the DIA kernel,
        // now; 0 = drop the copy and never build it; -1 = a candidate of AUTO's trial (the default).  8 bytes per slot.
        SPMV_REQUIRE(m->format == SPMV_FMT_ELL && value >= -1 && value <= 1, "ell_dia_order: an ELL handle and -1, 0 or 1");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
        m->ell_dia_order_req = (int32_t)value;
        if (value == 1)
        {
            SPMV_TRY(ell_build_dia_order(m, /*only_if_worth=*/false));
            if (m->coo_csr && m->kernel == SPMV_CSR_PANEL) m->kernel = SPMV_CSR_VECTOR;  // (the format's own kernel runs: this variant of it)
            m->ell_variant = 3;
        }
        else
            ell_free_dia_order(m);
    }
    else if (!strcmp(name, "coo_column_bins"))
    {
        // COO, segmented scan: bins per XCD of the copy the scan runs over (1..8), 0 = no copy (the scan reads the handle's
        // own arrays in their order), -1 = as many as keep a slice of x inside an XCD's L2
        SPMV_REQUIRE(m->format == SPMV_FMT_COO && value >= -1 && value <= 8, "coo_column_bins: a COO handle and -1 .. 8");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
        if (value == 0)
            coo_free_bins(m);
        else
            SPMV_TRY(coo_build_bins(m, value < 0 ? 0 : (int)value, /*only_if_worth=*/false));
    }
    else if (!strcmp(name, "dia_col_bound"))
    {
        SPMV_REQUIRE(m->format == SPMV_FMT_DIA && value >= 0 && value <= m->ncol, "dia_col_bound: a DIA handle and 0 <= bound <= ncol");
        m->dia_col_bound = (int32_t)value;
    }
    else if (!strcmp(name, "panel_sync"))
        m->pb_sync = (int32_t)value;
    else if (!strcmp(name, "twophase_offer_csr_copy"))
        m->tp_offer_csr = value < 0 || value > 2 ? 1 : (int32_t)value;  // (2: tests - a carved piece is taken whatever the timings say)
    else if (!strcmp(name, "panel_trial"))
        m->pb_trial = (int32_t)value;
    else if (!strcmp(name, "symgs_order"))  // 1 multicolour, 0 the matrix's own row order; takes effect at the next set-up / sweep
    {
        SPMV_REQUIRE(value == 0 || value == 1, "symgs_order: 0 (row order) or 1 (multicolour), got %lld", (long long)value);
        m->gs_order = (int32_t)value;
    }
    else if (!strcmp(name, "twophase_panel_cols"))  // takes effect at the next spmv_mat_set_kernel(TWOPHASE)
        m->tp_pcols_req = (int32_t)value;
    else if (!strcmp(name, "twophase_placement_budget_mb"))
    {
        // memory (MB) the piece search of the two-phase layout may hold beyond the product stream while it runs; 0 = no
        // search; -1 = the default (SPMV_TP_PLACEMENT_BUDGET_MB or 8192).  Takes effect at the next spmv_mat_set_kernel(TWOPHASE)
        // that builds the layout.
        SPMV_REQUIRE(value >= -1 && value <= (1 << 20), "twophase_placement_budget_mb: -1 (default), 0 (no search) or megabytes");
        m->tp_place_budget_mb = (int32_t)value;
    }
    else if (!strcmp(name, "twophase_pool_alloc") || !strcmp(name, "twophase_pool_config"))  // experiments (tools/probe_twophase_pairs.py)
    {
        const char* e_exp = getenv("SPMV_EXPERIMENTS");
        SPMV_REQUIRE(e_exp && e_exp[0] == '1', "%s is an experiment: set SPMV_EXPERIMENTS=1", name);
        SPMV_HIP(hipSetDevice(m->ctx->device));
        if (!strcmp(name, "twophase_pool_alloc"))
            SPMV_TRY(csr_twophase_pool_alloc(m, (int)value));
        else
            SPMV_TRY(csr_twophase_pool_config(m, value));
    }
    else if (!strcmp(name, "twophase_choose_pieces"))  // run the piece search of a built two-phase layout (again) with the current budget
    {
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_TRY(csr_twophase_choose_again(m));
    }
    else if (!strcmp(name, "twophase_rotate"))  // starting points of the expand phase's workgroups inside their panels: 256 (default), 0 = all at the start
    {
        SPMV_REQUIRE(value >= 0 && value <= 256, "twophase_rotate: 0 .. 256 starting points");
        m->tp_rotate = (int32_t)value;
    }
    else if (!strcmp(name, "twophase_only"))
    {
        // experiment (tools/tune_twophase.py): run phase A (1) or phase B (2) alone.  THE PRODUCT IS THEN WRONG, so the
        // switch exists only under SPMV_EXPERIMENTS=1 (read here, once, not on the product's path).
        const char* e_exp = getenv("SPMV_EXPERIMENTS");
        SPMV_REQUIRE(value == 0 || (e_exp && e_exp[0] == '1'), "twophase_only is an experiment: set SPMV_EXPERIMENTS=1");
        SPMV_REQUIRE(value >= 0 && value <= 2, "twophase_only: 0, 1 (phase A alone) or 2 (phase B alone)");
        m->tp_only = (int32_t)value;
    }
    else if (!strcmp(name, "twophase_realloc"))
    {
        // experiment (tools/probe_twophase_placement.py): move streams of the two-phase layout to fresh allocations, the
        // old ones freed only afterwards so that other memory is handed out.  Bits: 1 products, 2 values, 4 columns, 8 rows
        const char* e_exp = getenv("SPMV_EXPERIMENTS");
        SPMV_REQUIRE(e_exp && e_exp[0] == '1', "twophase_realloc is an experiment: set SPMV_EXPERIMENTS=1");
        SPMV_REQUIRE(m->tp_val && m->tp_padded > 0, "twophase_realloc: the two-phase layout is not built");
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
        auto move = [&](void** slot, size_t bytes) -> int {
            void* fresh = nullptr;
            if (hipMalloc(&fresh, bytes) != hipSuccess) SPMV_FAIL(SPMV_ERR_ALLOC, "twophase_realloc: out of device memory");
            if (const hipError_t e = hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice); e != hipSuccess)
            {
                (void)hipFree(fresh);
                SPMV_FAIL(SPMV_ERR_HIP, "twophase_realloc: copy failed: %s", hipGetErrorString(e));
            }
            (void)hipFree(*slot);
            *slot = fresh;
            return SPMV_OK;
        };
        const size_t np = (size_t)m->tp_padded;
        // (a pool of an experiment owns the pieces: replacing them here would leave the pool holding freed pointers)
        SPMV_REQUIRE(!((value & 1) && m->tp_pool), "twophase_realloc bit 1: this handle's pieces belong to a pool (twophase_pool_alloc)");
        if (value & 1)  // the product stream's pieces: fresh allocations (its contents need no copy: phase A rewrites all of it)
            for (int i = 0; i < m->tp_npieces; ++i)
            {
                double* fresh = nullptr;
                if (hipMalloc(&fresh, i + 1 < m->tp_npieces ? (size_t)1 << 30 : (size_t)m->tp_last_piece_bytes) != hipSuccess) SPMV_FAIL(SPMV_ERR_ALLOC, "twophase_realloc: out of device memory");
                (void)hipFree(m->tp_piece[i]);
                m->tp_piece[i] = fresh;
            }
        if (value & 2) SPMV_TRY(move((void**)&m->tp_val, sizeof(double) * np));
        if (value & 4) SPMV_TRY(move((void**)&m->tp_col, sizeof(uint16_t) * np));
        if (value & 8) SPMV_TRY(move((void**)&m->tp_row, sizeof(uint16_t) * np));
        if (value & 16) SPMV_TRY(move((void**)&m->tp_blk, sizeof(int32_t) * 2 * ((np + 15) / 16)));
    }
    else
        SPMV_FAIL(SPMV_ERR_INVALID, "unknown parameter '%s'", name);
    return SPMV_OK;
}

int spmv_mat_get_param(const spmv_mat* m, const char* name, int64_t* value)
{
    SPMV_REQUIRE(m && name && value, "null argument");
    if (!strncmp(name, "symgs_", 6))
    {
        SPMV_REQUIRE(symgs_info(m, name, value) == SPMV_OK, "unknown parameter '%s'", name);
        return SPMV_OK;
    }
    if (!strcmp(name, "panel_keep_csr"))
        *value = (m->b && m->v) || m->nnz == 0 ? 1 : 0;
    else if (!strcmp(name, "device_bytes"))
        *value = m->device_bytes;
    else if (!strcmp(name, "ell_tiled_values"))  // ELL: 1 if the product reads the values from the copy in tiles of 512 rows
        *value = m->ell_tval ? 1 : 0;
    else if (!strcmp(name, "ell_dia_order"))  // ELL: 1 if the product runs the DIA kernel over the DIA-order copy of the values
        *value = m->ell_variant == 3 && m->ell_rval ? 1 : 0;
    else if (!strcmp(name, "ell_non_conforming_rows"))  // ... and the rows the side kernel does
        *value = m->ell_rval ? m->ell_nc_count : 0;
    else if (!strcmp(name, "coo_column_bins"))  // bins of the copy the segmented scan runs over (8 x bins per XCD), 0: none
        *value = m->cb_bins;
    else if (!strcmp(name, "coo_bins_padded"))
        *value = m->cb_padded;
    else if (!strcmp(name, "panel_rows"))
        *value = m->pb_built_rows;
    else if (!strcmp(name, "panel_width"))
        *value = m->pb_built_width;
    else if (!strcmp(name, "panel_sort"))
        *value = m->pb_built_sort;
    else if (!strcmp(name, "panel_groups"))
        *value = m->pb_ngroups;
    else if (!strcmp(name, "panel_unroll"))
        *value = std::min(8, m->pb_unroll > 0 ? m->pb_unroll : (m->pb_unroll_tuned > 0 ? m->pb_unroll_tuned : 8));  // (16 runs 8 since round 4)
    else if (!strcmp(name, "panel_bytes"))
        *value = m->pb_bytes;
    else if (!strcmp(name, "panel_pipe"))
        *value = m->pb_pipe >= 0 ? m->pb_pipe : (m->pb_pipe_tuned > 0 ? m->pb_pipe_tuned : 1);
    else if (!strcmp(name, "panel_sync"))
        *value = (m->pb_sync >= 0 ? m->pb_sync : m->pb_sync_tuned) == 2 ? 3 : (m->pb_sync >= 0 ? m->pb_sync : m->pb_sync_tuned);  // (2, the split barrier, runs 3 since round 5)
    else if (!strcmp(name, "panel_layout"))  // layout in memory: 0 three arrays, 1 records, 3 packed 12-byte entries
        *value = m->pb_pack ? (m->pb_pair ? 4 : 3) : 0;
    else if (!strcmp(name, "ell_diagonal_slots"))  // ELL: 1 if the slots were found to be diagonals (no column stream for conforming rows)
        *value = m->ell_diag ? 1 : 0;
    else if (!strcmp(name, "twophase_panel_cols"))
        *value = m->tp_pcols;
    else if (!strcmp(name, "twophase_pieces_carved"))  // pieces of the product stream that lie inside the released CSR copy's allocations
    {
        *value = 0;
        for (int i = 0; i < m->tp_npieces; ++i) *value += m->tp_piece_carved[i] ? 1 : 0;
    }
    else if (!strcmp(name, "twophase_placements_timed"))
        *value = m->tp_place_seen;
    else if (!strcmp(name, "twophase_placement_spread"))  // time as built / time with the pieces the search kept, in 1/1000
        *value = m->tp_place_gain;
    else if (!strcmp(name, "twophase_padded"))
        *value = m->tp_padded;
    else if (!strcmp(name, "twophase_pieces"))  // 1 GB pieces the product stream consists of
        *value = m->tp_npieces;
    else if (!strcmp(name, "twophase_pieces_exchanged"))  // of them: exchanged for other pieces by the search
        *value = m->tp_pieces_exchanged;
    else if (!strcmp(name, "twophase_placement_budget_mb"))
        *value = m->tp_place_budget_mb;
    else if (!strcmp(name, "window_max_span"))
        *value = m->win_max_span;
    else if (!strcmp(name, "window_avg_span"))
        *value = (int64_t)m->win_avg_span;
    else if (!strcmp(name, "contiguous_permille"))
        *value = (int64_t)(m->contig_frac * 1000.0 + 0.5);
    else if (!strcmp(name, "select_candidates"))
        *value = m->sel_candidates;
    else if (!strcmp(name, "select_rounds"))  // rounds the handle's last trial went through until its candidates' minima stood still (2 .. 6; 0: no trial)
        *value = m->sel_rounds;
    else if (!strncmp(name, "select_us_", 10))
    {
        static const char* const kNames[] = {"", "vector", "ldswin", "scalar", "panel", "twophase", "variant1", "variant2"};
        int slot = -1;
        for (int i = 1; i < 8; ++i)
            if (!strcmp(name + 10, kNames[i])) slot = i;
        if (!strcmp(name + 10, "segscan")) slot = SPMV_CSR_SEGSCAN;  // (CSR handles; the slots are an ELL handle's "variant1" / "variant2")
        if (!strcmp(name + 10, "split")) slot = SPMV_CSR_SPLIT;
        if (!strcmp(name + 10, "split_low")) slot = 0;
        if (!strcmp(name + 10, "dia_order")) slot = 9;  // ELL handles: the DIA-order copy of the values
        if (!strcmp(name + 10, "ell")) slot = SPMV_CSR_ELL;  // CSR handles: the ELL copy of (nearly) equal rows  // kernel SPLIT with rows of 256 entries and more split off (timed from 8M entries on)
        SPMV_REQUIRE(slot >= 0, "unknown parameter '%s'", name);
        *value = (int64_t)(m->sel_us[slot] + 0.5f);
    }
    else if (!strcmp(name, "rowgrouped_kernel"))
        *value = m->coo_csr && m->kernel == SPMV_CSR_PANEL ? m->coo_csr->kernel : 0;
    else if (!strcmp(name, "ell_variant"))
        *value = m->ell_variant;
    else if (!strcmp(name, "panel_rounds"))
        *value = m->pb_built_rounds;
    else if (!strcmp(name, "panel_rounds_us_one"))
        *value = (int64_t)(m->pb_rounds_us[0] + 0.5f);
    else if (!strcmp(name, "panel_rounds_us_more"))
        *value = (int64_t)(m->pb_rounds_us[1] + 0.5f);
    else if (!strcmp(name, "split_row_threshold"))
        *value = m->format == SPMV_FMT_CSR ? csr_split_threshold(m) : 0;
    else if (!strcmp(name, "split_long_rows"))
        *value = m->split_long_rows;
    else if (!strcmp(name, "split_mode"))
        *value = m->split_built_mode ? m->split_built_mode : m->split_mode;
    else if (!strcmp(name, "split_virtual_rows"))
        *value = m->split_vrows;
    else if (!strcmp(name, "split_long_kernel"))
        *value = m->split_long ? m->split_long->kernel : 0;
    else if (!strcmp(name, "split_long_entries"))
        *value = m->split_long_nnz;
    else if (!strcmp(name, "ell_copy_slots"))
        *value = m->ell_copy ? (int64_t)m->ell_copy->nrow * m->ell_copy->k : 0;
    else if (!strcmp(name, "ell_copy_diagonal_slots"))
        *value = m->ell_copy && m->ell_copy->ell_diag ? 1 : 0;
    else if (!strcmp(name, "ell_copy_variant"))
        *value = m->ell_copy ? m->ell_copy->ell_variant : 0;
    else if (!strcmp(name, "min_row_entries"))
        *value = m->min_row_nnz;
    else if (!strcmp(name, "adds_into_y_with_atomics"))  // 1: the product adds into y with device atomics (spmv_apply_host stages y in device memory)
        *value = adds_into_y_with_atomics(m) ? 1 : 0;
    else if (!strcmp(name, "split_inner_kernel"))
        *value = m->format == SPMV_FMT_CSR && m->kernel == SPMV_CSR_SPLIT && m->coo_csr ? m->coo_csr->kernel : 0;
    else
        SPMV_FAIL(SPMV_ERR_INVALID, "unknown parameter '%s'", name);
    return SPMV_OK;
}

int spmv_mat_set_flags(spmv_mat* m, uint32_t flags)
{
    SPMV_REQUIRE(m, "null matrix");
    m->flags = flags;
    return SPMV_OK;
}

int spmv_mat_download(const spmv_mat* m, int32_t* a, int32_t* b, double* v)
{
    SPMV_REQUIRE(m, "null matrix");
    SPMV_TRY(use_device(m->ctx));
    size_t na = 0, nb = 0, nv = 0;
    switch (m->format)
    {
        case SPMV_FMT_CSR: na = (size_t)m->nrow + 1; nb = nv = (size_t)m->nnz; break;
        case SPMV_FMT_COO: na = nb = nv = (size_t)m->nnz; break;
        case SPMV_FMT_ELL: nb = nv = (size_t)m->nrow * (size_t)m->k; break;
        case SPMV_FMT_CSC: na = (size_t)m->ncol + 1; nb = nv = (size_t)m->nnz; break;
        case SPMV_FMT_DIA: na = (size_t)m->k; nv = (size_t)m->nrow * (size_t)m->k; break;
        default: SPMV_FAIL(SPMV_ERR_INVALID, "unknown format %d", m->format);
    }
    SPMV_REQUIRE(!((b && nb && !m->b) || (v && nv && !m->v)), "spmv_mat_download: this handle gave up its arrays (panel_keep_csr = 0)");
    hipStream_t s = m->ctx->stream;
    if (a && na) SPMV_HIP(hipMemcpyAsync(a, m->a, na * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (b && nb) SPMV_HIP(hipMemcpyAsync(b, m->b, nb * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (v && nv) SPMV_HIP(hipMemcpyAsync(v, m->v, nv * sizeof(double), hipMemcpyDeviceToHost, s));
    SPMV_HIP(hipStreamSynchronize(s));
    return SPMV_OK;
}

int spmv_mat_device_ptrs(const spmv_mat* m, const int32_t** a, const int32_t** b, const double** v)
{
    SPMV_REQUIRE(m, "null matrix");
    if (a) *a = m->a;
    if (b) *b = m->b;
    if (v) *v = m->v;
    return SPMV_OK;
}

// ---- the hot path -----------------------------------------------------------------------------------------
static int apply_checked(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y)
{
    switch (A->format)
    {
        case SPMV_FMT_CSR: return csr_apply(ctx, A, x->d, y->d);
        case SPMV_FMT_ELL: return ell_apply(ctx, A, x->d, y->d);
        case SPMV_FMT_COO: return coo_apply(ctx, A, x->d, y->d);
        case SPMV_FMT_CSC: return csc_apply(ctx, A, x->d, y->d);
        case SPMV_FMT_DIA: return dia_apply(ctx, A, x->d, y->d);
        default: SPMV_FAIL(SPMV_ERR_INVALID, "unknown format %d", A->format);
    }
}

static int check_apply_args(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y)
{
    SPMV_REQUIRE(ctx && A && x && y, "spmv_apply: null argument");
    SPMV_REQUIRE(x->n == A->ncol, "spmv_apply: x has %lld entries, matrix has %d columns", (long long)x->n, A->ncol);
    SPMV_REQUIRE(y->n == A->nrow, "spmv_apply: y has %lld entries, matrix (shard) has %d rows", (long long)y->n,
                 A->nrow);
    SPMV_REQUIRE(x->d != y->d || x->n == 0, "spmv_apply: x and y must not alias");
    return SPMV_OK;
}

int spmv_apply(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y)
{
    SPMV_TRY(check_apply_args(ctx, A, x, y));
    SPMV_TRY(use_device(ctx));
    return apply_checked(ctx, A, x, y);
}

int spmv_apply_timed(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y, int32_t reps,
                     double* ms_per_apply)
{
    SPMV_TRY(check_apply_args(ctx, A, x, y));
    SPMV_REQUIRE(reps > 0 && ms_per_apply, "spmv_apply_timed: reps=%d", reps);
    SPMV_TRY(use_device(ctx));
    SPMV_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    for (int32_t i = 0; i < reps; ++i) SPMV_TRY(apply_checked(ctx, A, x, y));
    SPMV_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    SPMV_HIP(hipEventSynchronize(ctx->ev_end));
    float ms = 0.f;
    SPMV_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    *ms_per_apply = (double)ms / reps;
    return SPMV_OK;
}

// ---- the reference's own call shape: host vectors in, host vector out ---------------------------------------------------------
// CSRMatrixMatVector(A, x, y) and its siblings take HOST vectors on every call (include/mat_vec.h:7-11; main.cpp:56-59 calls
// them 50 times).  Behind that signature a product costs two hand-overs whatever the kernel does; for small matrices they ARE the
// cost (C1: kernel 3 us).  Rounds 1-4 paid three synchronous hipMemcpy of pageable memory per call (~17 us each: 67 us per C1
// product, 4.8 GFLOP/s against the reference's 5.6 on one CPU thread).  Here, for vectors up to 1 MB together (tools/probe_apply_host_sizes.py: beyond ~1.5 MB the copies' fixed cost is paid back by their higher rate): the host copies
// y into a pinned, device-mapped staging buffer (memcpy: 80 KB in 3 us) and stores x straight into a device buffer where the
// platform lets the CPU do that (large BAR: 80 KB in 2 us; elsewhere x goes through the staging buffer and one kernel pulls it
// over the host link), the product gathers x from device memory and updates y IN the staging buffer (kernels that add into y
// with device atomics - the COO scan, the CSC scatter - get y through a device buffer and two more launches instead), the
// host polls the stream (hipStreamQuery: no interrupt wake-up) and copies y out.  ONE launch, no hipMemcpy: what is left of a
// C1 product is the 12-17 us a kernel launch takes from doorbell to completion signal on this platform.  Larger vectors take asynchronous copies from / to the caller's memory
// (the PCIe time dominates there).  The caller's arrays are never registered or mapped: they may be freed or re-allocated
// between calls without a stale mapping being left behind.
static int grow(double** p, size_t* have, size_t want)
{
    if (*have >= want) return SPMV_OK;
    if (*p) SPMV_HIP(hipFree(*p));
    *p    = nullptr;
    *have = 0;
    const size_t n = std::max<size_t>(want + want / 4, 4096);
    if (hipMalloc(p, sizeof(double) * n) != hipSuccess)
    {
        (void)hipGetLastError();
        SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_apply_host: no device memory for a staging vector of %zu entries", n);
    }
    *have = n;
    return SPMV_OK;
}

// The end of a staged call: hipStreamSynchronize.  Measured for an empty kernel (tools/probe_launch_floor.hip,
// profiles/r06_probe_launch_floor.txt): launch + hipStreamQuery spin (round 5) 16.9 us, launch + hipStreamSynchronize 10.5,
// launch + a word of pinned host memory the kernel writes and the host spins on 6.4.  The last one does not survive contact with a
// real call: the word has to come AFTER the product's stores, i.e. from a second launch (23.9 us per C1 call against 22.2 with
// the plain synchronize, same box, profiles/r06_dropin_small_products.txt) or from the product's own last workgroup behind
// system-wide fences of every lane (32 us).  So: the call HIP guarantees, and nothing to go wrong.
static int host_wait(spmv_ctx* ctx)
{
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    return SPMV_OK;
}

int spmv_apply_host(spmv_ctx* ctx, const spmv_mat* A, const double* x_host, double* y_host)
{
    SPMV_REQUIRE(ctx && A && (x_host || A->ncol == 0) && (y_host || A->nrow == 0), "spmv_apply_host: null argument");
    SPMV_REQUIRE(A->ctx == ctx, "spmv_apply_host: the matrix belongs to another context");
    SPMV_TRY(use_device(ctx));
    const size_t nx = (size_t)A->ncol, ny = (size_t)A->nrow;
    if (ny == 0) return SPMV_OK;
    // (No wait for the stream here: the staging buffers are touched by this function alone, and it returns only when its own
    // launches are done with them; work queued earlier by others runs before ours by stream order.  Growing a buffer waits.)
    if (ctx->stage_x_n < std::max<size_t>(nx, 1) || ctx->stage_y_n < ny) SPMV_HIP(hipStreamSynchronize(ctx->stream));
    SPMV_TRY(grow(&ctx->stage_x, &ctx->stage_x_n, std::max<size_t>(nx, 1)));
    SPMV_TRY(grow(&ctx->stage_y, &ctx->stage_y_n, ny));
    spmv_vec vx, vy;
    vx.ctx = vy.ctx = ctx;
    vx.n            = (int64_t)nx;
    vx.d            = ctx->stage_x;
    vy.n            = (int64_t)ny;
    vy.d            = ctx->stage_y;
    // (SPMV_HOST_STAGED_MB: the limit in megabytes, for tools/probe_apply_host_sizes.py; read per call, this is not a product's path
    // that anything is measured on)
    static const size_t kStagedLimit = [] {
        const char* e = getenv("SPMV_HOST_STAGED_MB");
        return ((size_t)(e && atoi(e) > 0 ? atoi(e) : 1) << 20) / sizeof(double);
    }();
    if (nx + ny <= kStagedLimit)
    {
        if (ctx->stage_pinned_n < nx + ny)
        {
            SPMV_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->stage_pinned) (void)hipHostFree(ctx->stage_pinned);
            ctx->stage_pinned     = nullptr;
            ctx->stage_pinned_dev = nullptr;
            ctx->stage_pinned_n   = 0;
            const size_t n        = std::max<size_t>(2 * (nx + ny), 32768);
            void*        dev      = nullptr;
            if (hipHostMalloc((void**)&ctx->stage_pinned, sizeof(double) * n, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer(&dev, ctx->stage_pinned, 0) != hipSuccess)
            {
                if (ctx->stage_pinned) (void)hipHostFree(ctx->stage_pinned);
                ctx->stage_pinned = nullptr;
                SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_apply_host: no pinned host memory for %zu staged entries: %s", n, hipGetErrorString(hipGetLastError()));
            }
            ctx->stage_pinned_dev = (double*)dev;
            ctx->stage_pinned_n   = n;
        }
        double* hx = ctx->stage_pinned;
        double* hy = ctx->stage_pinned + nx;
        // Kernels that touch every y_i once with a plain read and a plain store (the row-parallel, LDS-window, scalar, panel and
        // two-phase CSR kernels, the ELL and DIA kernels, and handles running from a copy that runs one of those) update y IN
        // the staging buffer over the host link: two launches.  Kernels that add into y with device atomics (the COO scan, the
        // CSC scatter, CSR under SEGSCAN or SPLIT's chunks - whether the handle's own or its copy's: adds_into_y_with_atomics)
        // get y through a device buffer: fp64 atomics on host memory over the link are platform behaviour, not a HIP guarantee.
        // (experiment, tools/probe_apply_host_atomics.py: SPMV_EXPERIMENTS=1 SPMV_HOST_Y_IN_PLACE=1 keeps y in the staging buffer
        // whatever the kernel - to SEE what device atomics on mapped host memory do on a given box; never a product's path)
        static const bool kForceInPlace = [] {
            const char *e = getenv("SPMV_EXPERIMENTS"), *f = getenv("SPMV_HOST_Y_IN_PLACE");
            return e && e[0] == '1' && f && f[0] == '1';
        }();
        const bool y_in_place = kForceInPlace || !adds_into_y_with_atomics(A);
        // Round 6 (tools/probe_launch_floor.hip, tools/probe_small_host_calls.py; profiles/r06_dropin_small_products.txt):
        //   * the call ends in hipStreamSynchronize, not in a hipStreamQuery spin (host_wait above), and does not begin with one;
        //   * y NEED NOT TRAVEL TO THE DEVICE.  The CSR kernels that form a row's sum and add it to y in one step - row-parallel,
        //     panel, two-phase: y_i = y_i + sum_i, the reference's own shape (src/mat_vec.cpp:59-64: private sum, one +=) - write
        //     sum_i alone (their "overwrite" mode) and the host does y_host[i] += sum_i while it copies out: the same IEEE add,
        //     bit for bit what the kernel would have stored, one memcpy and one crossing of the host link less.  Kernels whose
        //     accumulator starts AT y_i (ELL, DIA, scalar CSR: the reference's order y0 + p0 + p1 + ...) keep y in place.
        const spmv_mat* K = A;  // the handle whose kernel runs
        while (K->format != SPMV_FMT_CSR && K->coo_csr && K->kernel == SPMV_CSR_PANEL) K = K->coo_csr;
        const bool sum_then_add = y_in_place && !kForceInPlace && K->format == SPMV_FMT_CSR && K->nnz > 0 &&
                                  (K->kernel == SPMV_CSR_VECTOR || K->kernel == SPMV_CSR_AUTO || K->kernel == SPMV_CSR_PANEL || K->kernel == SPMV_CSR_TWOPHASE);
        if (nx && !ctx->large_bar) memcpy(hx, x_host, sizeof(double) * nx);
        if (!sum_then_add) memcpy(hy, y_host, sizeof(double) * ny);
        // x: where the CPU can store into device memory (large BAR) it writes x into the device buffer itself - 80 KB in 2 us,
        // no launch (tools/probe_host_write_vram.hip: the next kernel sees the stores, also right after a kernel that read the
        // previous contents; the HDP flush register is written behind them as the platform prescribes for such stores; the
        // context checked it once for itself when it was created).  Elsewhere one kernel pulls x out of the staging buffer.
        const bool direct = ctx->large_bar != 0;
        if (direct && nx)
        {
            memcpy(ctx->stage_x, x_host, sizeof(double) * nx);
            __sync_synchronize();
            if (ctx->hdp_flush) *ctx->hdp_flush = 1u;
        }
        if (y_in_place)
        {
            vy.d = ctx->stage_pinned_dev + nx;
            if (!direct) SPMV_TRY(vec_copy2(ctx, ctx->stage_x, ctx->stage_pinned_dev, (int64_t)nx, nullptr, nullptr, 0));
            if (sum_then_add)
            {
                apply_extra ex;
                ex.overwrite = true;
                SPMV_TRY(mat_apply_ex(ctx, A, vx.d, vy.d, ex));
            }
            else
                SPMV_TRY(apply_checked(ctx, A, &vx, &vy));
        }
        else
        {
            SPMV_TRY(vec_copy2(ctx, ctx->stage_x, ctx->stage_pinned_dev, direct ? 0 : (int64_t)nx, ctx->stage_y, ctx->stage_pinned_dev + nx, (int64_t)ny));
            SPMV_TRY(apply_checked(ctx, A, &vx, &vy));
            SPMV_TRY(vec_copy2(ctx, ctx->stage_pinned_dev + nx, ctx->stage_y, (int64_t)ny, nullptr, nullptr, 0));
        }
        SPMV_TRY(host_wait(ctx));
        if (sum_then_add)
            for (size_t i = 0; i < ny; ++i) y_host[i] += hy[i];
        else
            memcpy(y_host, hy, sizeof(double) * ny);
        return SPMV_OK;
    }
    if (nx) SPMV_HIP(hipMemcpyAsync(ctx->stage_x, x_host, sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
    SPMV_HIP(hipMemcpyAsync(ctx->stage_y, y_host, sizeof(double) * ny, hipMemcpyHostToDevice, ctx->stream));
    SPMV_TRY(apply_checked(ctx, A, &vx, &vy));
    SPMV_HIP(hipMemcpyAsync(y_host, ctx->stage_y, sizeof(double) * ny, hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    return SPMV_OK;
}

// ---- BLAS-1 -----------------------------------------------------------------------------------------------
int spmv_dot(spmv_ctx* ctx, const spmv_vec* x, const spmv_vec* y, double* result)
{
    SPMV_REQUIRE(ctx && x && y && result, "spmv_dot: null argument");
    SPMV_REQUIRE(x->n == y->n, "spmv_dot: sizes %lld and %lld differ", (long long)x->n, (long long)y->n);
    SPMV_TRY(use_device(ctx));
    return vec_dot(ctx, x->d, y->d, x->n, result);
}

int spmv_axpby(spmv_ctx* ctx, double alpha, const spmv_vec* x, double beta, const spmv_vec* y, spmv_vec* w)
{
    SPMV_REQUIRE(ctx && x && y && w, "spmv_axpby: null argument");
    // the reference sizes the loop by w (src/vec_vec.cpp:33)
    SPMV_REQUIRE(x->n >= w->n && y->n >= w->n, "spmv_axpby: w has %lld entries, x %lld, y %lld", (long long)w->n,
                 (long long)x->n, (long long)y->n);
    SPMV_TRY(use_device(ctx));
    return vec_axpby(ctx, alpha, x->d, beta, y->d, w->d, w->n);
}

// ---- solver step (solver.hip) -----------------------------------------------------------------------------
int spmv_apply_dot(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y, int32_t overwrite, const spmv_vec* w,
                   double* dot)
{
    SPMV_TRY(check_apply_args(ctx, A, x, y));
    SPMV_REQUIRE(w && dot, "spmv_apply_dot: null argument");
    SPMV_REQUIRE(w->n == A->nrow, "spmv_apply_dot: w has %lld entries, matrix has %d rows", (long long)w->n, A->nrow);
    SPMV_REQUIRE(w->d != y->d || y->n == 0, "spmv_apply_dot: w and y must not alias");
    SPMV_TRY(use_device(ctx));
    double* out = ctx->dev_scalars;
    SPMV_HIP(hipMemsetAsync(out, 0, sizeof(double) * kDotDoubles, ctx->stream));
    apply_extra ex;
    ex.overwrite = overwrite != 0;
    ex.dot_w     = w->d;
    ex.dot_out   = out;
    SPMV_TRY(mat_apply_ex(ctx, A, x->d, y->d, ex));
    std::vector<double> slots(kDotDoubles);
    SPMV_HIP(hipMemcpyAsync(slots.data(), out, sizeof(double) * kDotDoubles, hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    for (int i = 0; i < kDotSlots; ++i) total += slots[(size_t)i * kDotStride];
    *dot = total;
    return SPMV_OK;
}

int spmv_cg(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* b, spmv_vec* x, int32_t max_iter, double rel_tol,
            int32_t check_every, int32_t precond, int32_t* iters, double* rel_resid)
{
    SPMV_REQUIRE(ctx && A && b && x && iters && rel_resid, "spmv_cg: null argument");
    SPMV_REQUIRE(A->nrow == A->ncol, "spmv_cg: the matrix is %d x %d, not square", A->nrow, A->ncol);
    SPMV_REQUIRE(b->n == A->nrow && x->n == A->nrow, "spmv_cg: b has %lld and x %lld entries, the matrix %d rows",
                 (long long)b->n, (long long)x->n, A->nrow);
    SPMV_REQUIRE(b->d != x->d || x->n == 0, "spmv_cg: b and x must not alias");
    SPMV_REQUIRE(max_iter >= 0 && rel_tol >= 0.0, "spmv_cg: max_iter=%d rel_tol=%g", max_iter, rel_tol);
    SPMV_REQUIRE(precond == SPMV_PRECOND_NONE || precond == SPMV_PRECOND_JACOBI || precond == SPMV_PRECOND_SYMGS,
                 "spmv_cg: unknown preconditioner %d", precond);
    SPMV_TRY(use_device(ctx));
    return cg_solve(ctx, A, b->d, x->d, max_iter, rel_tol, check_every, precond, iters, rel_resid);
}

int spmv_symgs_setup(spmv_ctx* ctx, spmv_mat* A)
{
    SPMV_REQUIRE(ctx && A, "spmv_symgs_setup: null argument");
    SPMV_REQUIRE(A->ctx == ctx, "spmv_symgs_setup: the matrix belongs to another context");
    SPMV_TRY(use_device(ctx));
    return symgs_setup(A);
}

int spmv_symgs_order(spmv_ctx* ctx, const spmv_mat* A, int32_t* order)
{
    SPMV_REQUIRE(ctx && A && (order || A->nrow == 0), "spmv_symgs_order: null argument");
    SPMV_TRY(use_device(ctx));
    return symgs_sequence(A, order);
}

int spmv_symgs(spmv_ctx* ctx, spmv_mat* A, const spmv_vec* b, spmv_vec* x, int32_t sweeps)
{
    SPMV_REQUIRE(ctx && A && b && x, "spmv_symgs: null argument");
    SPMV_REQUIRE(A->ctx == ctx, "spmv_symgs: the matrix belongs to another context");
    SPMV_REQUIRE(b->n == A->nrow && x->n == A->nrow, "spmv_symgs: b has %lld and x %lld entries, the matrix %d rows", (long long)b->n,
                 (long long)x->n, A->nrow);
    SPMV_REQUIRE(b->d != x->d || x->n == 0, "spmv_symgs: b and x must not alias");
    SPMV_REQUIRE(sweeps >= 0, "spmv_symgs: sweeps = %d", sweeps);
    SPMV_TRY(use_device(ctx));
    SPMV_TRY(symgs_setup(A));  // first call: split, levels, schedule (synchronous); later calls: nothing
    for (int k = 0; k < sweeps; ++k) SPMV_TRY(symgs_sweep(ctx, A, b->d, x->d, /*zero_guess=*/false));
    return SPMV_OK;
}

// ---- conversions ------------------------------------------------------------------------------------------
// a converted ELL handle gets the same analysis as an uploaded one (kernel choice, regrouped copy when it pays)
static int analyse_new_ell(spmv_mat** out)
{
    const int rc = ell_analyse(*out);
    if (rc != SPMV_OK)
    {
        mat_free(*out);
        *out = nullptr;
    }
    return rc;
}

int spmv_coo_to_csr(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out_csr)
{
    SPMV_REQUIRE(ctx && coo && out_csr, "spmv_coo_to_csr: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    return coo_to_csr(ctx, coo, out_csr);
}

int spmv_csr_to_ell(spmv_ctx* ctx, const spmv_mat* csr, spmv_mat** out_ell)
{
    SPMV_REQUIRE(ctx && csr && out_ell, "spmv_csr_to_ell: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    SPMV_TRY(csr_to_ell(ctx, csr, out_ell));
    return analyse_new_ell(out_ell);
}

int spmv_csr_split_columns(spmv_ctx* ctx, const spmv_mat* csr, int32_t col_begin, int32_t col_end, spmv_mat** out_inside,
                           spmv_mat** out_outside)
{
    SPMV_REQUIRE(ctx && csr && out_inside && out_outside, "spmv_csr_split_columns: null argument");
    SPMV_TRY(use_device(ctx));
    spmv_mat *in = nullptr, *outm = nullptr;
    SPMV_TRY(csr_split_columns(ctx, csr, col_begin, col_end, &in, &outm));
    int rc = finish(in, out_inside);  // validates and analyses like an uploaded handle (frees on failure)
    if (rc != SPMV_OK)
    {
        mat_free(outm);
        return rc;
    }
    rc = finish(outm, out_outside);
    if (rc != SPMV_OK)
    {
        mat_free(*out_inside);
        *out_inside = nullptr;
    }
    return rc;
}

int spmv_coo_to_ell(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out_ell)
{
    SPMV_REQUIRE(ctx && coo && out_ell, "spmv_coo_to_ell: null argument");
    SPMV_TRY(use_device(ctx));
    spmv_mat* csr = nullptr;
    SPMV_TRY(coo_to_csr(ctx, coo, &csr));
    int rc = csr_to_ell(ctx, csr, out_ell);
    mat_free(csr);
    plan_arm arm(ctx);  // (for the ELL handle: the intermediate CSR form above selected as usual)
    return rc == SPMV_OK ? analyse_new_ell(out_ell) : rc;
}

// ---- sharding ---------------------------------------------------------------------------------------------
int spmv_partition_rows(int64_t nrow, int32_t nparts, int32_t part, int64_t* row_begin, int64_t* row_end)
{
    SPMV_REQUIRE(nrow >= 0 && nparts > 0 && part >= 0 && part < nparts && row_begin && row_end,
                 "spmv_partition_rows: nrow=%lld nparts=%d part=%d", (long long)nrow, nparts, part);
    // src/mat_vec.cpp:233,245-246: rows_per_thread = nrow / nthreads, the last part takes the remainder
    const int64_t per = nrow / nparts;
    *row_begin        = (int64_t)part * per;
    *row_end          = (part == nparts - 1) ? nrow : *row_begin + per;
    return SPMV_OK;
}

int spmv_partition_rows_balanced(int64_t nrow, const int64_t* row_ptr64, int32_t nparts, int64_t* bounds)
{
    SPMV_REQUIRE(nrow >= 0 && row_ptr64 && nparts > 0 && bounds, "spmv_partition_rows_balanced: bad argument");
    const int64_t nnz = row_ptr64[nrow] - row_ptr64[0];
    bounds[0]         = 0;
    for (int32_t p = 1; p < nparts; ++p)
    {
        // the row boundary whose offset lies NEAREST to p / nparts of the entries (a row is never split; ties go to the later
        // boundary), kept monotonic
        const int64_t  target = row_ptr64[0] + (int64_t)(((__int128)nnz * p) / nparts);
        const int64_t* it     = std::lower_bound(row_ptr64, row_ptr64 + nrow + 1, target);
        int64_t        r      = it - row_ptr64;
        if (r > nrow) r = nrow;
        if (r > 0 && target - row_ptr64[r - 1] < row_ptr64[r] - target) --r;
        if (r < bounds[p - 1]) r = bounds[p - 1];
        bounds[p] = r;
    }
    bounds[nparts] = nrow;
    return SPMV_OK;
}

// bounds of a row partition of a device-resident handle (columns for CSC, which the reference shards by column)
int spmv_mat_partition_rows(const spmv_mat* m, int32_t nparts, int32_t balance_entries, int64_t* bounds)
{
    SPMV_REQUIRE(m && nparts > 0 && bounds, "spmv_mat_partition_rows: bad argument");
    const int64_t n = m->format == SPMV_FMT_CSC ? m->ncol : m->nrow;
    // padded formats store the same number of slots for every row: equal rows ARE equal work
    if (!balance_entries || m->format == SPMV_FMT_ELL || m->format == SPMV_FMT_DIA || n == 0)
    {
        for (int32_t p = 0; p < nparts; ++p)
        {
            int64_t b = 0, e = 0;
            SPMV_TRY(spmv_partition_rows(n, nparts, p, &b, &e));
            bounds[p]     = b;
            bounds[p + 1] = e;
        }
        return SPMV_OK;
    }
    std::vector<int64_t> rp64((size_t)n + 1);
    if (m->format == SPMV_FMT_COO)
        SPMV_TRY(coo_row_offsets(m, rp64.data()));
    else
    {
        SPMV_REQUIRE(m->a, "spmv_mat_partition_rows: the handle has no offset array");
        std::vector<int32_t> rp((size_t)n + 1);
        SPMV_HIP(hipSetDevice(m->ctx->device));
        SPMV_HIP(hipMemcpyAsync(rp.data(), m->a, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, m->ctx->stream));
        SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
        for (size_t i = 0; i < rp.size(); ++i) rp64[i] = rp[i];
    }
    return spmv_partition_rows_balanced(n, rp64.data(), nparts, bounds);
}

int spmv_csr_extract_rows(spmv_ctx* dst_ctx, const spmv_mat* csr, int64_t row_begin, int64_t row_end, spmv_mat** out)
{
    SPMV_REQUIRE(dst_ctx && csr && out, "spmv_csr_extract_rows: null argument");
    int before = -1;
    if (hipGetDevice(&before) != hipSuccess) before = -1;
    spmv_mat* m  = nullptr;
    plan_arm  arm(dst_ctx);
    int       rc = csr_extract_rows(dst_ctx, csr, row_begin, row_end, &m);
    if (rc == SPMV_OK) rc = finish(m, out);  // validates and analyses like an uploaded shard (frees on failure)
    if (before >= 0) (void)hipSetDevice(before);
    return rc;
}

// ---- generators -------------------------------------------------------------------------------------------
int spmv_gen_csr_uniform(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol, int32_t k, int32_t band,
                         uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out, "spmv_gen_csr_uniform: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    return gen_csr_uniform(ctx, row_begin, row_end, ncol, k, band, seed, out);
}

int spmv_gen_ell_banded(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out, "spmv_gen_ell_banded: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    SPMV_TRY(gen_ell_banded(ctx, nrow, ncol, k, seed, out));
    return analyse_new_ell(out);
}

int spmv_gen_dia_banded(spmv_ctx* ctx, int32_t nrow, int32_t k, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out, "spmv_gen_dia_banded: null argument");
    SPMV_TRY(use_device(ctx));
    return gen_dia_banded(ctx, nrow, k, seed, out);
}

int spmv_gen_coo_powerlaw(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out, "spmv_gen_coo_powerlaw: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    return gen_coo_powerlaw(ctx, nrow, ncol, max_len, seed, false, out);
}

int spmv_gen_coo_powerlaw_sorted(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(ctx && out, "spmv_gen_coo_powerlaw_sorted: null argument");
    SPMV_TRY(use_device(ctx));
    plan_arm arm(ctx);
    return gen_coo_powerlaw(ctx, nrow, ncol, max_len, seed, true, out);
}

int spmv_gen_vec_uniform(spmv_ctx* ctx, spmv_vec* v, int64_t index_offset, uint64_t seed)
{
    SPMV_REQUIRE(ctx && v, "spmv_gen_vec_uniform: null argument");
    SPMV_TRY(use_device(ctx));
    return gen_vec_uniform(ctx, v->d, v->n, index_offset, seed);
}

}  // extern "C"
