// kernels_misc.hip — BLAS-1, fill, and the CSC / DIA products on CDNA4 (gfx950).
//
// Replaces vec_dot / vec_axpby (reference src/vec_vec.cpp:15-29, :31-94), Vector::Fill
// (src/vector.cpp:59-63), CSCMatrixMatVector (src/mat_vec.cpp:69-95) and DIAMatrixMatVector
// (src/mat_vec.cpp:123-146).  All are HBM-bound streaming kernels: grid-stride, 16-byte accesses where
// alignment allows, at most kMaxGrid workgroups.
#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(kBlock) void fill_kernel(double* __restrict__ d, int64_t n, double a)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) d[i] = a;
}

// ---- dot: per-lane fma chain -> wavefront sum -> workgroup sum -> one partial per workgroup ---------------
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double s_part[kBlock / kWave];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kBlock / kWave; ++w) total += s_part[w];
    return total;  // valid in thread 0
}

__global__ __launch_bounds__(kBlock) void dot_partial_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                             int64_t n, double* __restrict__ partial)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        acc = fma(load_stream(x + i), load_stream(y + i), acc);
    const double total = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void dot_final_kernel(const double* __restrict__ partial, int count,
                                                           double* __restrict__ out)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
    const double total = block_sum(acc);
    if (threadIdx.x == 0) *out = total;
}

// ---- axpby: the reference's seven branches (src/vec_vec.cpp:38-93), chosen on the host -------------------
// alpha == 0 never reads x and beta == 0 never reads y, so NaN/Inf there cannot leak into w.
enum AxpbyMode
{
    kBetaY = 0,   // alpha == 0       : w = beta*y
    kAlphaX,      // beta == 0        : w = alpha*x
    kXPlusBy,     // alpha == 1       : w = beta*y + x
    kByMinusX,    // alpha == -1      : w = beta*y - x
    kAxPlusY,     // beta == 1        : w = alpha*x + y
    kAxMinusY,    // beta == -1       : w = alpha*x - y
    kGeneral      //                  : w = alpha*x + beta*y
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void axpby_kernel(int64_t n, double alpha, const double* __restrict__ x,
                                                       double beta, const double* __restrict__ y,
                                                       double* __restrict__ w)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        double r;
        if constexpr (MODE == kBetaY) r = beta * y[i];
        if constexpr (MODE == kAlphaX) r = alpha * x[i];
        if constexpr (MODE == kXPlusBy) r = fma(beta, y[i], x[i]);
        if constexpr (MODE == kByMinusX) r = fma(beta, y[i], -x[i]);
        if constexpr (MODE == kAxPlusY) r = fma(alpha, x[i], y[i]);
        if constexpr (MODE == kAxMinusY) r = fma(alpha, x[i], -y[i]);
        if constexpr (MODE == kGeneral) r = fma(alpha, x[i], beta * y[i]);
        w[i] = r;
    }
}

// Two elements per lane with 16-byte accesses (the streaming width that reaches HBM speed on this chip); the same
// element-wise arithmetic, so the results are bit-identical to the one-element kernel.  NT: nontemporal loads and
// stores for vectors far beyond the caches.
template <int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void axpby2_kernel(int64_t npairs, double alpha, const f64x2_t* __restrict__ x,
                                                        double beta, const f64x2_t* __restrict__ y, f64x2_t* __restrict__ w)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock)
    {
        f64x2_t xv = {0.0, 0.0}, yv = {0.0, 0.0}, r;
        if constexpr (MODE != kBetaY) xv = NT ? __builtin_nontemporal_load(x + i) : x[i];
        if constexpr (MODE != kAlphaX) yv = NT ? __builtin_nontemporal_load(y + i) : y[i];
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            if constexpr (MODE == kBetaY) r[e] = beta * yv[e];
            if constexpr (MODE == kAlphaX) r[e] = alpha * xv[e];
            if constexpr (MODE == kXPlusBy) r[e] = fma(beta, yv[e], xv[e]);
            if constexpr (MODE == kByMinusX) r[e] = fma(beta, yv[e], -xv[e]);
            if constexpr (MODE == kAxPlusY) r[e] = fma(alpha, xv[e], yv[e]);
            if constexpr (MODE == kAxMinusY) r[e] = fma(alpha, xv[e], -yv[e]);
            if constexpr (MODE == kGeneral) r[e] = fma(alpha, xv[e], beta * yv[e]);
        }
        if constexpr (NT)
            __builtin_nontemporal_store(r, w + i);
        else
            w[i] = r;
    }
}

// ---- CSC: scatter.  LPC lanes share one column; every entry is one fp64 atomic on y --------------------------
template <int LPC>
__global__ __launch_bounds__(kBlock) void csc_kernel(int ncol, const int32_t* __restrict__ col_ptr,
                                                     const int32_t* __restrict__ row, const double* __restrict__ val,
                                                     const double* __restrict__ x, double* __restrict__ y)
{
    const int c = blockIdx.x * (kBlock / LPC) + threadIdx.x / LPC;
    if (c >= ncol) return;
    const double xc  = x[c];
    const int    end = col_ptr[c + 1];
    for (int j = col_ptr[c] + threadIdx.x % LPC; j < end; j += LPC)
        unsafeAtomicAdd(y + load_stream(row + j), load_stream(val + j) * xc);
}

// ---- DIA: one lane per row, diagonals left to right from y[i] (bit-identical to orc_dia_spmv_fma) ------------
// The column bound is min(nrow, ncol): the reference checks `j < nrow` (src/mat_vec.cpp:140) while x has ncol entries,
// so for nrow > ncol it reads past the end of x (times a stored 0.0: DIAMatrix(CSR) never stores such an entry); here
// those slots are skipped, which is the same sum whenever the reference's overread is finite and cannot fault.  For
// nrow < ncol the reference's bound drops columns [nrow, ncol) and so does this kernel.
// The reference stores the diagonals ROW-major (values[i*ndiags + d], src/matrix.cpp:721): a lane walking its own
// row would make every load instruction touch 64 different lines.  So a workgroup copies a tile of 256 rows x 16
// diagonals (one 128-byte line per row) into LDS with 16 consecutive lanes per line, and every lane then reads its
// row from LDS (row stride 17 doubles: conflict-free).  x[i + offset] is contiguous across lanes.
constexpr int kDiaChunk = 16;

// WIDE: 16-byte loads (two adjacent diagonals per lane, 8 lanes per 128-byte line); needs an even ndiags so that every
// pair is 16-byte aligned.  8-byte accesses reach only ~0.65x of the streaming rate on this chip.
// XWIN: the offsets lie within a narrow band (off_min .. off_max): the stretch x[r0 + off_min .. r0 + 255 + off_max] the
// 256 rows of the workgroup multiply with is copied into LDS once, and the ndiags reads of x per row come from there
// instead of from global memory (64 vector loads per row otherwise: the vector memory pipe, not HBM, was the limit).
// skip_rows (may be null): bit i set = row i is somebody else's (the DIA-order copy of an ELL handle leaves its non-conforming
// rows to a side kernel, kernels_ell.hip): its y is neither read nor written here.
template <bool WIDE, bool XWIN>
__global__ __launch_bounds__(kBlock) void dia_kernel(int nrow, int jmax, int ndiags, const int32_t* __restrict__ offsets,
                                                     const double* __restrict__ val, const double* __restrict__ x,
                                                     double* __restrict__ y, int off_min, int off_max,
                                                     const unsigned long long* __restrict__ skip_rows, int stride)
{
    // stride: doubles between two rows' values (ndiags for a DIA handle - the reference's layout; the DIA-order copy of an ELL
    // handle with an odd number of slots pads its rows to an even stride so that the 16-byte path applies: the pad is never
    // consumed, the loop below runs to ndiags)
    __shared__ double tile[kBlock * (kDiaChunk + 1)];
    extern __shared__ double xs[];  // XWIN: kBlock + off_max - off_min entries of x
    constexpr int PER   = WIDE ? 2 : 1;                  // diagonals per lane and load
    constexpr int LPR   = kDiaChunk / PER;               // lanes per row of the tile
    constexpr int RPP   = kBlock / LPR;                  // rows per pass of the workgroup
    constexpr int NPASS = kBlock / RPP;                  // passes per tile
    const int r0 = blockIdx.x * kBlock;
    const int i  = r0 + threadIdx.x;
    const bool mine = i < nrow && !(skip_rows && ((skip_rows[i >> 6] >> (i & 63)) & 1ull));
    double    acc = mine ? y[i] : 0.0;
    // element j of this lane's share of a chunk: tile position (r, d) = (r_mine + RPP j, d_mine [+ 1])
    const int d_mine = (threadIdx.x % LPR) * PER;
    const int r_mine = threadIdx.x / LPR;
    double    stage[NPASS * PER];
    auto fetch = [&](int d0) {
#pragma unroll
        for (int j = 0; j < NPASS; ++j)
        {
            const int     r = r_mine + j * RPP;
            const double* p = val + (size_t)(r0 + r) * stride + d0 + d_mine;
            if constexpr (WIDE)
            {
                f64x2_t v = {0.0, 0.0};
                if (r0 + r < nrow && d0 + d_mine < ndiags) v = __builtin_nontemporal_load((const f64x2_t*)p);  // stride even: the pair is 16-byte aligned and inside the row
                stage[2 * j]     = v[0];
                stage[2 * j + 1] = v[1];
            }
            else
                stage[j] = (r0 + r < nrow && d0 + d_mine < ndiags) ? load_stream(p) : 0.0;
        }
    };
    fetch(0);
    if constexpr (XWIN)
    {
        const int w = kBlock + off_max - off_min;
        for (int k = threadIdx.x; k < w; k += kBlock)
        {
            const int64_t j = (int64_t)r0 + off_min + k;
            xs[k]           = (j >= 0 && j < jmax) ? x[j] : 0.0;
        }
        // (visible after the barriers of the first chunk below)
    }
    for (int d0 = 0; d0 < ndiags; d0 += kDiaChunk)
    {
        const int dn = min(kDiaChunk, ndiags - d0);
        __syncthreads();  // the previous chunk has been consumed
#pragma unroll
        for (int j = 0; j < NPASS; ++j)
#pragma unroll
            for (int e = 0; e < PER; ++e) tile[(r_mine + j * RPP) * (kDiaChunk + 1) + d_mine + e] = stage[PER * j + e];
        __syncthreads();
        if (d0 + kDiaChunk < ndiags) fetch(d0 + kDiaChunk);  // in flight while this chunk is consumed
        if (mine)
            for (int d = 0; d < dn; ++d)
            {
                const int off = offsets[d0 + d];  // wave-uniform address: scalar load
                const int j   = i + off;
                if (j >= 0 && j < jmax)
                    acc = fma(tile[threadIdx.x * (kDiaChunk + 1) + d], XWIN ? xs[(int)threadIdx.x + off - off_min] : x[j], acc);
            }
    }
    if (mine) y[i] = acc;
}

inline int stream_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock))); }
}  // namespace

namespace
{
// dst0[0..n0) = src0, dst1[0..n1) = src1 in one launch (16-byte accesses where both sides are aligned): what moves the caller's
// small host vectors between pinned host memory and the device in spmv_apply_host - a kernel reading mapped host memory
// starts within microseconds, a hipMemcpy costs 15-20 us of runtime per call
__global__ __launch_bounds__(kBlock) void copy2_kernel(double* __restrict__ dst0, const double* __restrict__ src0, int64_t n0, double* __restrict__ dst1,
                                                       const double* __restrict__ src1, int64_t n1)
{
    const int64_t total = n0 + n1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock)
    {
        if (i < n0)
            dst0[i] = src0[i];
        else
            dst1[i - n0] = src1[i - n0];
    }
}
}  // namespace

int vec_copy2(spmv_ctx* ctx, double* dst0, const double* src0, int64_t n0, double* dst1, const double* src1, int64_t n1)
{
    if (n0 + n1 <= 0) return SPMV_OK;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div(n0 + n1, kBlock)));
    hipLaunchKernelGGL(copy2_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, dst0, src0, n0, dst1, src1, n1);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int vec_fill(spmv_ctx* ctx, double* d, int64_t n, double a)
{
    if (n == 0) return SPMV_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, d, n, a);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int vec_dot(spmv_ctx* ctx, const double* x, const double* y, int64_t n, double* result)
{
    const int grid = stream_grid(n);
    SPMV_TRY(ensure_scratch(ctx, sizeof(double) * (size_t)(grid + 8)));
    double* partial = (double*)ctx->scratch;
    double* out     = partial + grid;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, x, y, n, partial);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial, grid, out);
    SPMV_HIP(hipGetLastError());
    SPMV_HIP(hipMemcpyAsync(ctx->host_pinned, out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    *result = ctx->host_pinned[0];
    return SPMV_OK;
}

template <int MODE>
static void launch_axpby(hipStream_t s, int64_t n, double alpha, const double* x, double beta, const double* y, double* w)
{
    // 16-byte path when every array that is touched is 16-byte aligned; an odd last element goes to the scalar kernel
    const bool    aligned = ((((uintptr_t)w) | (MODE != kBetaY ? (uintptr_t)x : 0) | (MODE != kAlphaX ? (uintptr_t)y : 0)) & 15) == 0;
    const int64_t npairs  = aligned ? n / 2 : 0;
    if (npairs > 0)
    {
        const dim3 grid(stream_grid(npairs)), block(kBlock);
        if (n >= (int64_t)(8 << 20))  // three vectors of 64 MiB and more: nothing of them survives in a cache anyway
            hipLaunchKernelGGL((axpby2_kernel<MODE, true>), grid, block, 0, s, npairs, alpha, (const f64x2_t*)x, beta,
                               (const f64x2_t*)y, (f64x2_t*)w);
        else
            hipLaunchKernelGGL((axpby2_kernel<MODE, false>), grid, block, 0, s, npairs, alpha, (const f64x2_t*)x, beta,
                               (const f64x2_t*)y, (f64x2_t*)w);
    }
    const int64_t done = npairs * 2;
    if (done < n)
        hipLaunchKernelGGL(axpby_kernel<MODE>, dim3(stream_grid(n - done)), dim3(kBlock), 0, s, n - done, alpha, x + done, beta,
                           y + done, w + done);
}

int vec_axpby(spmv_ctx* ctx, double alpha, const double* x, double beta, const double* y, double* w, int64_t n)
{
    if (n == 0) return SPMV_OK;
    hipStream_t s = ctx->stream;
    // same branch order as src/vec_vec.cpp:38-93
    if (alpha == 0)
        launch_axpby<kBetaY>(s, n, alpha, x, beta, y, w);
    else if (beta == 0)
        launch_axpby<kAlphaX>(s, n, alpha, x, beta, y, w);
    else if (alpha == 1)
        launch_axpby<kXPlusBy>(s, n, alpha, x, beta, y, w);
    else if (alpha == -1)
        launch_axpby<kByMinusX>(s, n, alpha, x, beta, y, w);
    else if (beta == 1)
        launch_axpby<kAxPlusY>(s, n, alpha, x, beta, y, w);
    else if (beta == -1)
        launch_axpby<kAxMinusY>(s, n, alpha, x, beta, y, w);
    else
        launch_axpby<kGeneral>(s, n, alpha, x, beta, y, w);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

// Large CSC: one fp64 atomic per entry is an order of magnitude slower than the row-grouped panel product, so the
// handle is regrouped by row once (expand col_ptr to per-entry columns, then the same device path as a COO handle:
// spmv_coo_to_csr + panel layout); the scatter kernel stays for small matrices.  Same sums, different order.
template <int LPC>
__global__ __launch_bounds__(kBlock) void csc_expand_cols_kernel(int ncol, const int32_t* __restrict__ col_ptr,
                                                                 int32_t* __restrict__ col_of_entry)
{
    const int c = blockIdx.x * (kBlock / LPC) + threadIdx.x / LPC;
    if (c >= ncol) return;
    const int end = col_ptr[c + 1];
    for (int j = col_ptr[c] + threadIdx.x % LPC; j < end; j += LPC) col_of_entry[j] = c;
}

namespace
{
int csc_scatter_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    constexpr int LPC = 8;
    if (!launch_fits(A->ncol, LPC)) SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "CSC product: %d columns are more than one launch holds", A->ncol);
    hipLaunchKernelGGL(csc_kernel<LPC>, dim3((unsigned)ceil_div(A->ncol, kBlock / LPC)), dim3(kBlock), 0, ctx->stream,
                       A->ncol, A->a, A->b, A->v, x, y);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace

void csc_drop_rowgrouped(spmv_mat* m)
{
    if (!m->coo_csr) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    m->device_bytes -= m->coo_csr->device_bytes;
    mat_free(m->coo_csr);
    m->coo_csr = nullptr;
    if (m->kernel == SPMV_CSR_PANEL) m->kernel = SPMV_CSR_VECTOR;
}

// The entries of a CSC handle grouped by row on the device (column expansion + spmv_coo_to_csr through a borrowed COO view of
// the same arrays: duplicates and the column-major order inside a row kept) as an internal CSR handle; force_kernel AUTO: that
// copy picks its kernel like any CSR handle (select.hip), PANEL: the panel layout (spmv_mat_set_kernel(csc, SPMV_CSR_PANEL)).
int csc_build_rowgrouped(spmv_mat* m, int32_t force_kernel)
{
    if (m->coo_csr && (force_kernel == SPMV_CSR_AUTO || m->coo_csr->kernel == force_kernel))
    {
        m->kernel = SPMV_CSR_PANEL;
        return SPMV_OK;
    }
    if (m->nnz == 0 || m->nnz > (int64_t)INT32_MAX - 65536 || !launch_fits(m->ncol, 8)) return SPMV_OK;
    csc_drop_rowgrouped(m);
    spmv_ctx* ctx  = m->ctx;
    int32_t*  cols = nullptr;
    if (hipMalloc(&cols, sizeof(int32_t) * (size_t)m->nnz) != hipSuccess)
    {
        (void)hipGetLastError();
        SPMV_FAIL(SPMV_ERR_ALLOC, "no device memory for the column indices of %lld CSC entries", (long long)m->nnz);
    }
    constexpr int LPC = 8;
    hipLaunchKernelGGL(csc_expand_cols_kernel<LPC>, dim3((unsigned)ceil_div(m->ncol, kBlock / LPC)), dim3(kBlock), 0,
                       ctx->stream, m->ncol, m->a, cols);
    // a borrowed COO view of the same entries: rows = row_ind (column-major order), columns = expanded
    spmv_mat view;
    view.ctx    = ctx;
    view.format = SPMV_FMT_COO;
    view.nrow   = m->nrow;
    view.ncol   = m->ncol;
    view.nnz    = m->nnz;
    view.a      = m->b;  // CSC row_ind
    view.b      = cols;
    view.v      = m->v;
    view.owned  = false;
    view.pb_trial      = m->pb_trial;
    view.kernel_forced = true;  // nothing is selected or built for the temporary view itself
    int       rc  = coo_analyse(&view);  // sortedness of the row indices
    view.plan_base = m->plan_base;  // (a CSC handle built from a plan: spmv_coo_to_csr hands the copy's node down from the view -
    view.plan_at   = m->plan_at;    // set only now: the view itself is nobody's handle and must not act on the plan)
    spmv_mat* csr = nullptr;
    if (rc == SPMV_OK) rc = coo_to_csr(ctx, &view, &csr, force_kernel);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(cols);
    if (rc != SPMV_OK) return rc;
    // the panel and two-phase layouts read row_ptr and their own arrays only
    if ((csr->kernel == SPMV_CSR_PANEL || csr->kernel == SPMV_CSR_TWOPHASE || csr->kernel == SPMV_CSR_ELL) && csr->b && csr->v)
    {
        (void)hipFree(const_cast<int32_t*>(csr->b));
        (void)hipFree(const_cast<double*>(csr->v));
        csr->device_bytes -= (int64_t)csr->nnz * 12;
        csr->b = nullptr;
        csr->v = nullptr;
    }
    m->coo_csr = csr;
    m->kernel  = SPMV_CSR_PANEL;  // reported for CSC as "runs from the row-grouped copy"
    m->device_bytes += csr->device_bytes;
    return SPMV_OK;
}

// AUTO for a CSC handle (round 5, select.hip): the scatter over the columns (one fp64 atomic on y per entry) or the copy grouped
// by row.  Model: the copy from 1.5M entries on; from 64K entries on both are timed.
int csc_select_kernel(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    select_reset(m);
    csc_drop_rowgrouped(m);
    m->kernel = SPMV_CSR_VECTOR;
    if (m->nnz == 0 || m->nrow <= 0 || m->ncol <= 0) return SPMV_OK;
    const bool model_copy = m->nnz >= ((int64_t)3 << 19);
    if (!select_trials_enabled(m) || m->nnz < kSelectMinNnz) return model_copy ? csc_build_rowgrouped(m, SPMV_CSR_AUTO) : SPMV_OK;
    select_scratch sv;
    if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return model_copy ? csc_build_rowgrouped(m, SPMV_CSR_AUTO) : SPMV_OK;
    float t_own = 1e30f, t_copy = 1e30f;
    int   rc    = csc_build_rowgrouped(m, SPMV_CSR_AUTO);
    if (rc == SPMV_ERR_ALLOC)
    {
        (void)hipGetLastError();
        return SPMV_OK;  // no memory for the copy: the scatter runs
    }
    if (rc != SPMV_OK) return rc;
    if (!m->coo_csr) return SPMV_OK;
    {
        // the copy is built; the two are timed in rounds until their minima stand still (select.hip)
        float t[2] = {-1.f, -1.f};
        rc = select_rounds(ctx, 2, [&](int j) { return j == 0 ? csr_apply(ctx, m->coo_csr, sv.x, sv.y) : csc_scatter_apply(ctx, m, sv.x, sv.y); }, t, &m->sel_rounds);
        if (rc == SPMV_OK)
        {
            t_copy = t[0] >= 0.f ? t[0] : 1e30f;
            t_own  = t[1] >= 0.f ? t[1] : 1e30f;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != SPMV_OK) return rc;
    select_note(m, SPMV_CSR_PANEL, t_copy);
    select_note(m, SPMV_CSR_VECTOR, t_own);
    const bool keep_copy = model_copy ? t_copy <= t_own * 1.02f : t_copy < t_own * 0.98f;
    if (!keep_copy) csc_drop_rowgrouped(m);
    m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
    return SPMV_OK;
}

int csc_analyse(spmv_mat* m)
{
    m->kernel = SPMV_CSR_VECTOR;  // reported for CSC as "scatter over the columns"
    const bool from_ctx = plan_take_armed(m);
    if (plan_of(m))
    {
        const int rc = csc_apply_plan(m);
        plan_clear(m);
        if (rc == SPMV_OK || !from_ctx) return rc;
        (void)hipGetLastError();  // (a context's plan that does not fit this matrix: the handle selects by itself)
    }
    if (m->kernel_forced) return SPMV_OK;
    return csc_select_kernel(m);
}

// A plan on a CSC handle (plan.hip): the scatter, or the row-grouped copy with the kernel and layout its own node names
int csc_apply_plan(spmv_mat* m)
{
    const plan_node& p = *plan_of(m);
    select_reset(m);
    csc_drop_rowgrouped(m);
    m->kernel = SPMV_CSR_VECTOR;
    if (m->nnz == 0 || m->nrow <= 0 || m->ncol <= 0 || p.kernel != SPMV_CSR_PANEL) return SPMV_OK;
    SPMV_TRY(csc_build_rowgrouped(m, SPMV_CSR_AUTO));
    m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
    return SPMV_OK;
}

int csc_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->ncol == 0 || A->nnz == 0) return SPMV_OK;
    if (A->coo_csr && A->kernel == SPMV_CSR_PANEL) return csr_apply(ctx, A->coo_csr, x, y);
    return csc_scatter_apply(ctx, A, x, y);
}

int dia_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nrow == 0 || A->k == 0) return SPMV_OK;
    // a row shard (offsets shifted by its first row) keeps the bound of the whole matrix; never past the end of x
    const int jmax = std::min(A->dia_col_bound > 0 ? A->dia_col_bound : std::min(A->nrow, A->ncol), A->ncol);
    return dia_rows_apply(ctx, A->nrow, jmax, A->k, A->a, A->v, x, y, A->dia_off_known, A->dia_off_min, A->dia_off_max, A->flags, nullptr, A->k);
}

// the DIA product over row-major values (a DIA handle's own, or the DIA-order copy of an ELL handle's values: kernels_ell.hip)
int dia_rows_apply(spmv_ctx* ctx, int nrow, int jmax, int k, const int32_t* offsets, const double* values, const double* x, double* y, bool off_known,
                   int off_min, int off_max, uint32_t flags, const unsigned long long* skip_rows, int stride)
{
    if (nrow == 0 || k == 0) return SPMV_OK;
    const dim3 grid((unsigned)ceil_div(nrow, kBlock));
    const bool wide = stride % 2 == 0 && stride >= k + (k & 1) && (((uintptr_t)values) & 15) == 0;  // (an odd k needs its pad: a pair never leaves the row)
    // offsets within a band of at most 1792 (known from the upload / the generator): x goes through LDS
    const bool   xwin = off_known && off_min <= off_max && (int64_t)off_max - off_min <= 1792 && !(flags & SPMV_FLAG_DIA_GLOBAL_X);
    const size_t lds  = xwin ? sizeof(double) * (size_t)(kBlock + off_max - off_min) : 0;
#define SPMV_DIA(W, X)                                                                                                 \
    hipLaunchKernelGGL((dia_kernel<W, X>), grid, dim3(kBlock), lds, ctx->stream, nrow, jmax, k, offsets, values, x, y, \
                       xwin ? off_min : 0, xwin ? off_max : 0, skip_rows, stride)
    if (wide && xwin)
        SPMV_DIA(true, true);
    else if (wide)
        SPMV_DIA(true, false);
    else if (xwin)
        SPMV_DIA(false, true);
    else
        SPMV_DIA(false, false);
#undef SPMV_DIA
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
