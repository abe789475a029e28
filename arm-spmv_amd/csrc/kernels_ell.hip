// kernels_ell.hip — y += A*x for column-major ELL on CDNA4 (gfx950).
//
// Replaces ELLMatrixMatVector (reference src/mat_vec.cpp:97-121).  The reference walks the slots in
// the outer loop and read-modify-writes y once per slot (K passes over y); here one lane owns one row,
// keeps the accumulator in a register and touches y once.  The per-row addition order is the same
// (y0 + p0 + p1 + ...), so with fma the result is bit-identical to the oracle's orc_ell_spmv_fma.
//
// Layout (include/matrix.h:70, src/matrix.cpp:487-488): element (row i, slot s) at i + s*nrow, so
// the 64 lanes of a wavefront read 64 consecutive column indices (256 B) and 64 consecutive values
// (512 B) per slot: perfectly coalesced.  Padding slots hold col 0 / val 0.0 (src/matrix.cpp:473-474)
// and are multiplied like any other slot, as in the reference.
//
// Roofline: HBM-bound; algorithmic bytes per application = 12*nrow*K + 8*ncol + 16*nrow.
#include <algorithm>
#include <vector>


#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
// UNROLL slots are fetched before the first fma so that UNROLL gathers of x are in flight per lane.
// MASKED (round 6): the handle is the ELL COPY of a CSR handle (csr_ell_copy_build).  Its padding slots carry value 0.0 and a
// NEGATIVE column, the complement of the row's own last column (convert.hip): such a slot takes no part in the sum - a row that
// reads x[c] = +-inf ends at +-inf as in the reference's CSR loop (src/mat_vec.cpp:57-65), not at the NaN of 0.0 * inf, and a row
// without entries is not touched in its value - while the gather the kernel issues for it anyway goes to a column the row reads
// itself.  No length array, no mask words: the information rides in the index stream the kernel reads in any case, and the
// diagonal-slot path (which reads no index) never meets a padding slot - a conforming slot holds a real column by construction.
// (Two earlier forms, measured against the unmasked copy on one box, tools/ab_ell_copy_masked.py: row lengths read off row_ptr
// +5-6 % on 5- and 7-point stencils - 4 bytes per row are 6 % of their traffic; a bit per row and lengths for padded rows only
// +1.5 / +5 %: the dependent loads stand in front of every wavefront's short life.)  A true ELL handle multiplies its padding as
// the reference's ELL loop does (0.0 * x[0], src/mat_vec.cpp:108-117): MASKED = false is the code of rounds 1-5, unchanged.
template <int UNROLL, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void ell_kernel(int nrow, int k, const int32_t* __restrict__ col,
                                                     const double* __restrict__ val,
                                                     const double* __restrict__ x, double* __restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrow) return;
    double       acc    = y[i];
    const size_t stride = (size_t)nrow;
    size_t       at     = (size_t)i;
    int          s      = 0;
    for (; s + UNROLL <= k; s += UNROLL)
    {
        int    c[UNROLL];
        double v[UNROLL];
        double xv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            c[u] = load_stream(col + at + (size_t)u * stride);
            v[u] = load_stream(val + at + (size_t)u * stride);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) xv[u] = x[MASKED && c[u] < 0 ? ~c[u] : c[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (!MASKED || c[u] >= 0) acc = fma(v[u], xv[u], acc);
        at += (size_t)UNROLL * stride;
    }
    for (; s < k; ++s)
    {
        const int    c  = load_stream(col + at);
        const double v  = load_stream(val + at), xv = x[MASKED && c < 0 ? ~c : c];
        if (!MASKED || c >= 0) acc = fma(v, xv, acc);
        at += stride;
    }
    y[i] = acc;  // (a row without entries gets back the y it had: the same bits)
}

// Two adjacent rows per lane: 8-byte column loads and 16-byte value loads (1 KiB per wavefront
// instruction).  Needs nrow even so that every slot column starts 16-byte aligned.
template <int UNROLL, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void ell_kernel_x2(int nrow, int k, const int32_t* __restrict__ col,
                                                        const double* __restrict__ val,
                                                        const double* __restrict__ x, double* __restrict__ y)
{
    const int i = 2 * (blockIdx.x * kBlock + threadIdx.x);
    if (i >= nrow) return;  // nrow even: i+1 < nrow too
    f64x2        acc    = *reinterpret_cast<const f64x2*>(y + i);
    const size_t stride = (size_t)nrow;
    size_t       at     = (size_t)i;
    int          s      = 0;
    for (; s + UNROLL <= k; s += UNROLL)
    {
        i32x2   c[UNROLL];
        f64x2   v[UNROLL];
        double  xa[UNROLL], xb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            c[u] = load_stream(reinterpret_cast<const i32x2*>(col + at + (size_t)u * stride));
            v[u] = load_stream(reinterpret_cast<const f64x2*>(val + at + (size_t)u * stride));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            xa[u] = x[MASKED && c[u].x < 0 ? ~c[u].x : c[u].x];
            xb[u] = x[MASKED && c[u].y < 0 ? ~c[u].y : c[u].y];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            if (!MASKED || c[u].x >= 0) acc.x = fma(v[u].x, xa[u], acc.x);
            if (!MASKED || c[u].y >= 0) acc.y = fma(v[u].y, xb[u], acc.y);
        }
        at += (size_t)UNROLL * stride;
    }
    for (; s < k; ++s)
    {
        const i32x2   c = load_stream(reinterpret_cast<const i32x2*>(col + at));
        const f64x2   v = load_stream(reinterpret_cast<const f64x2*>(val + at));
        const double  xa = x[MASKED && c.x < 0 ? ~c.x : c.x], xb = x[MASKED && c.y < 0 ? ~c.y : c.y];
        if (!MASKED || c.x >= 0) acc.x = fma(v.x, xa, acc.x);
        if (!MASKED || c.y >= 0) acc.y = fma(v.y, xb, acc.y);
        at += stride;
    }
    *reinterpret_cast<f64x2*>(y + i) = acc;
}
// ---- ELL whose slots are diagonals ---------------------------------------------------------------------------------
// ELL is what stencil and band matrices are stored in, and there slot s of most rows holds the same diagonal:
// col[i + s*nrow] == i + off[s].  The values of such a slot ARE a diagonal stored contiguously, so for those rows the
// column index need not be read at all — a third of the stream (C3: 3.07 GB -> 2.05 GB) — and the entries of x a row
// block needs are one contiguous stretch that goes through LDS once (the DIA kernel's trick, kernels_misc.hip).  When a
// handle is analysed, off[s] is taken from the middle row and every (pair of rows, slot) gets one bit: both rows
// conform.  The bits of a wavefront's 64 row pairs for one slot are one 64-bit word, fetched with a scalar load and
// used as the lane predicate: conforming lanes compute their columns, the others (boundary rows with padding, wrap-
// around rows, rows with arbitrary columns) read them as before.  Same products in the same order: bit-identical to
// ell_kernel_x2.  Cost: 1 bit per 24 bytes of matrix.
using u64 = unsigned long long;

// one lane per row pair; mask[wave * k + s] = ballot of "both rows of the pair hold column i + off[s] in slot s"
constexpr int kRefCandidates = 16, kRefSample = 2048;
// candidate c for the reference row: 16 rows spread over the sampled stretch around the middle, at a spacing that does
// not line up with round grid sizes (the middle row itself is often a boundary row of the grid, and so are its neighbours)
__host__ __device__ inline int ref_candidate(int nrow, int c)
{
    const int count = nrow < kRefSample ? nrow : kRefSample, first = nrow / 2 - count / 2 > 0 ? nrow / 2 - count / 2 : 0;
    const int r = first + (c * 131 + 17) % count;
    return r < nrow ? r : nrow - 1;
}

// which row near the middle speaks for the most others?  hits[c] = entries of the sampled rows that sit where
// candidate row c says
__global__ __launch_bounds__(kBlock) void ell_ref_row_kernel(int nrow, int k, const int32_t* __restrict__ col, int first, int count,
                                                             int32_t* __restrict__ hits)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= count) return;
    const int i = first + t;
    for (int c = 0; c < kRefCandidates; ++c)
    {
        const int ref = ref_candidate(nrow, c);
        int       ok  = 0;
        for (int s = 0; s < k; ++s) ok += (col[(size_t)i + (size_t)s * nrow] - i == col[(size_t)ref + (size_t)s * nrow] - ref) ? 1 : 0;
        if (ok) atomicAdd(hits + c, ok);
    }
}

__global__ __launch_bounds__(kBlock) void ell_diag_scan_kernel(int nrow, int k, int ref_row, const int32_t* __restrict__ col, int32_t* __restrict__ off,
                                                               u64* __restrict__ mask, u64* __restrict__ covered)
{
    const int  pair = blockIdx.x * kBlock + threadIdx.x;
    const int  i    = 2 * pair;
    const int  wave = pair >> 6, lane = threadIdx.x & 63;
    const bool on   = i + 1 < nrow;
    u64        mine = 0;
    for (int s = 0; s < k; ++s)
    {
        const int o = col[(size_t)ref_row + (size_t)s * nrow] - ref_row;  // uniform: scalar load
        if (pair == 0) off[s] = o;
        bool conf = false;
        if (on)
        {
            const i32x2 c = *reinterpret_cast<const i32x2*>(col + (size_t)i + (size_t)s * nrow);
            conf          = c.x >= 0 && c.y >= 0 && c.x == i + o && c.y == i + 1 + o;  // (>= 0: a padding slot of a CSR handle's ELL copy holds a complement)
        }
        const u64 b = __ballot(conf);
        if (lane == 0 && i < nrow)  // (the grid is rounded up to whole workgroups: wavefronts past the last row pair own no words)
        {
            mask[(size_t)wave * k + s] = b;
            mine += (u64)__popcll(b);
        }
    }
    if (lane == 0 && mine) atomicAdd(covered, mine);
}

// The stretches of x a block of 2 * kBlock rows needs, one per CLUSTER of nearby offsets (a 7-point stencil on a grid:
// {-m^2}, {-m}, {-1, 0, 1}, {m}, {m^2}), copied into LDS once per workgroup.  Descriptor array (int32, after off[K]):
//   xbase[K]                       LDS index of x[r0 + off[s]] for slot s (r0 = first row of the block)
//   ncl, then {lo, len, base} per cluster: x[r0 + lo .. r0 + lo + len) sits at xs[base ..)
__device__ __forceinline__ void stage_x_windows(double* xs, const int32_t* __restrict__ desc, int k, const double* __restrict__ x, int r0, int ncol)
{
    const int ncl = desc[2 * k];
    for (int c = 0; c < ncl; ++c)
    {
        const int     lo = desc[2 * k + 1 + 3 * c], len = desc[2 * k + 2 + 3 * c], base = desc[2 * k + 3 + 3 * c];
        const int64_t j0 = (int64_t)r0 + lo;
        if (j0 >= 0 && j0 + len <= ncol && ((j0 | base | len) & 1) == 0 && (((uintptr_t)x) & 15) == 0)
        {
            // whole stretch inside x, even start: 16-byte loads (half the vector-memory instructions)
            const f64x2* __restrict__ x2 = reinterpret_cast<const f64x2*>(x + j0);
            f64x2*                    s2 = reinterpret_cast<f64x2*>(xs + base);
            for (int t = threadIdx.x; t < len / 2; t += kBlock) s2[t] = x2[t];
        }
        else
            for (int t = threadIdx.x; t < len; t += kBlock)
            {
                const int64_t j = j0 + t;
                xs[base + t]    = (j >= 0 && j < ncol) ? x[j] : 0.0;
            }
    }
    __syncthreads();
}

// (Round 4 double-buffered the value stream and requested its first group before the x stretches go through LDS: the same
// time on the same box, 0.398 ms on C3 - profiles/r04_tune_ell_c3_double_buffered.txt - so the simpler form stays.)
//
// TILED (round 4): `val` is the engine's copy of the values in TILES of 512 rows - tile b, slot s, row r at
// (b * k + s) * 512 + r (ell_build_tiles) - so that what a workgroup reads is one contiguous stretch of 512 * k * 8 bytes
// instead of k stretches of 4 KB that lie nrow * 8 bytes (C3: 32 MB) apart.  Same loads, same order of the slots, same
// bits.  C3, A/B in one process: 0.345-0.360 -> 0.331-0.334 ms on a box where the column-major form is fast, 0.4045 ->
// 0.400 on one where it is slow - what separates the boxes is not the stride.  Opt-in ("ell_tiled_values"): 8 more bytes
// per slot are a poor price for that.
template <int UNROLL, bool XWIN, bool TILED, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void ell_diag_kernel_x2(int nrow, int k, const int32_t* __restrict__ col,
                                                             const double* __restrict__ val, const double* __restrict__ x,
                                                             double* __restrict__ y, const int32_t* __restrict__ off,
                                                             const u64* __restrict__ mask, int ncol)
{
    extern __shared__ double xs[];  // XWIN: the stretches of x this block's conforming entries read (stage_x_windows)
    const int r0 = 2 * kBlock * (int)blockIdx.x;
    if constexpr (XWIN) stage_x_windows(xs, off, k, x, r0, ncol);
    const int i = r0 + 2 * (int)threadIdx.x;
    if (i >= nrow) return;  // nrow even: i+1 < nrow too
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kBlock + threadIdx.x) >> 6));
    const u64* __restrict__ wm = mask + (size_t)wave * k;  // this wavefront's words: scalar loads
    const u64 bit = 1ull << (threadIdx.x & 63);
    const int32_t* __restrict__ xbase = off + k;
    f64x2        acc    = *reinterpret_cast<const f64x2*>(y + i);
    const size_t stride = TILED ? (size_t)(2 * kBlock) : (size_t)nrow;
    size_t       at     = TILED ? (size_t)blockIdx.x * (2 * kBlock) * k + 2 * threadIdx.x : (size_t)i;
    for (int s0 = 0; s0 < k; s0 += UNROLL)
    {
        f64x2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (s0 + u < k) v[u] = load_stream(reinterpret_cast<const f64x2*>(val + at + (size_t)u * stride));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (s0 + u < k)
            {
                const int s = s0 + u;
                double    x0, x1;
                bool      live0 = true, live1 = true;  // (MASKED: a padding slot of a CSR handle's ELL copy - see ell_kernel)
                if (wm[s] & bit)  // nearly always the whole wavefront; a conforming slot holds a real column
                {
                    if constexpr (XWIN)
                    {
                        const int at_x = xbase[s] + 2 * (int)threadIdx.x;
                        x0             = xs[at_x];
                        x1             = xs[at_x + 1];
                    }
                    else
                    {
                        const int c0 = i + off[s];
                        x0           = x[c0];
                        x1           = x[c0 + 1];
                    }
                }
                else
                {
                    const i32x2 c = load_stream(reinterpret_cast<const i32x2*>(col + (size_t)i + (size_t)s * nrow));
                    if constexpr (MASKED)
                    {
                        live0 = c.x >= 0;
                        live1 = c.y >= 0;
                    }
                    x0 = x[MASKED && c.x < 0 ? ~c.x : c.x];
                    x1 = x[MASKED && c.y < 0 ? ~c.y : c.y];
                }
                if (!MASKED || live0) acc.x = fma(v[u].x, x0, acc.x);
                if (!MASKED || live1) acc.y = fma(v[u].y, x1, acc.y);
            }
        at += (size_t)UNROLL * stride;
    }
    *reinterpret_cast<f64x2*>(y + i) = acc;
}

// ---- large ELL with non-local columns ----------------------------------------------------------------------
// One lane per row is the right kernel when neighbouring rows touch neighbouring columns (C3's band: the x window of
// a row block sits in L2).  With scattered columns every gather misses L2 exactly as in the row-parallel CSR kernel.
// Such a handle gets the panel layout too (kernels_csr_panel.hip): the slots are copied row by row — ALL of them,
// padding included, so that the sums are the reference's (its loop multiplies the padding's 0.0 by x[0] as well,
// src/mat_vec.cpp:108-117) — and only row_ptr and the panel arrays are kept.
__global__ __launch_bounds__(kBlock) void ell_window_scan_kernel(int nrow, int k, const int32_t* __restrict__ col,
                                                                 const double* __restrict__ val,
                                                                 unsigned long long* __restrict__ span_sum)
{
    __shared__ int s_lo, s_hi;
    if (threadIdx.x == 0)
    {
        s_lo = INT32_MAX;
        s_hi = -1;
    }
    __syncthreads();
    const int i  = blockIdx.x * kBlock + threadIdx.x;
    int       lo = INT32_MAX, hi = -1;
    if (i < nrow)
        for (int s = 0; s < k; ++s)
        {
            const size_t e = (size_t)i + (size_t)s * nrow;
            if (val[e] != 0.0)  // padding (col 0, val 0.0) says nothing about where the row's columns are
            {
                const int c = col[e];
                lo          = min(lo, c);
                hi          = max(hi, c);
            }
        }
    if (hi >= 0)
    {
        atomicMin(&s_lo, lo);
        atomicMax(&s_hi, hi);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_hi >= 0) atomicAdd(span_sum, (unsigned long long)(s_hi - s_lo + 1));
}

// The row-grouped copy of an ELL handle.  Every slot takes part in the reference's sum, the padding too: it adds 0.0 * x[0]
// (src/mat_vec.cpp:108-117), which is nothing unless x[0] is not finite.  ONE such term per row says the same as many, so the
// copy keeps every slot that is not (column 0, value 0.0) and, for a row that has such slots, one entry (0, 0.0) where the first
// of them stood: a mesh matrix of 16-64 entries per row in 64 slots is copied at 1.03x its entries instead of 1.6x.
// FILL = false: count[i] = entries row i gets (count[nrow] = 0); FILL = true: write them behind row_ptr[i].
template <bool FILL>
__global__ __launch_bounds__(kBlock) void ell_to_csr_kernel(int nrow, int k, const int32_t* __restrict__ ecol,
                                                            const double* __restrict__ eval, int32_t* __restrict__ count,
                                                            const int32_t* __restrict__ row_ptr, int32_t* __restrict__ ccol,
                                                            double* __restrict__ cval)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i > nrow) return;
    if (i == nrow)
    {
        if constexpr (!FILL) count[i] = 0;
        return;
    }
    int    n   = 0;
    bool   pad = false;
    size_t at  = FILL ? (size_t)row_ptr[i] : 0;
    for (int s = 0; s < k; ++s)
    {
        const size_t e = (size_t)i + (size_t)s * nrow;
        const int    c = ecol[e];
        const double v = eval[e];
        const bool   p = c == 0 && v == 0.0;
        if (p && pad) continue;  // (a second 0.0 * x[0]: says nothing the first did not)
        pad = pad || p;
        if constexpr (FILL)
        {
            ccol[at + n] = c;
            cval[at + n] = v;
        }
        ++n;
    }
    if constexpr (!FILL) count[i] = n;
}
}  // namespace

namespace
{
// mean number of columns a block of 256 consecutive rows spans (padding ignored); < 0 on failure
double ell_mean_block_span(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (ensure_scratch(ctx, 64) != SPMV_OK) return -1.0;
    unsigned long long* d_sum = (unsigned long long*)ctx->scratch;
    unsigned long long  h_sum = 0;
    const unsigned      nblk  = (unsigned)ceil_div(m->nrow, kBlock);
    if (hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) return -1.0;
    hipLaunchKernelGGL(ell_window_scan_kernel, dim3(nblk), dim3(kBlock), 0, ctx->stream, m->nrow, m->k, m->b, m->v, d_sum);
    if (hipMemcpyAsync(&h_sum, d_sum, sizeof(h_sum), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return -1.0;
    return (double)h_sum / nblk;
}

void ell_drop_rowgrouped(spmv_mat* m)
{
    if (!m->coo_csr) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    m->device_bytes -= m->coo_csr->device_bytes;
    mat_free(m->coo_csr);
    m->coo_csr = nullptr;
    if (m->kernel == SPMV_CSR_PANEL) m->kernel = SPMV_CSR_VECTOR;
}

// every slot that says something - the padding's `0.0 * x[0]` of the reference (src/mat_vec.cpp:108-117) once per row, see
// ell_to_csr_kernel - copied row by row into a CSR handle; force_kernel AUTO: the copy picks its kernel like any CSR handle
int ell_make_rowgrouped(spmv_mat* m, int32_t force_kernel)
{
    spmv_ctx*     ctx   = m->ctx;
    SPMV_REQUIRE((int64_t)m->nrow * m->k <= (int64_t)INT32_MAX - 65536, "the row-grouped copy of an ELL handle of %d rows x %d slots: shard it (int32 offsets)",
                 m->nrow, m->k);
    spmv_mat*      csr   = nullptr;
    int32_t *      cnt = nullptr, *rp = nullptr;
    const unsigned grid = (unsigned)ceil_div((int64_t)m->nrow + 1, kBlock);
    int32_t        total = 0;
    int            rc0   = SPMV_OK;
    if (hipMalloc(&cnt, sizeof(int32_t) * ((size_t)m->nrow + 1)) != hipSuccess || hipMalloc(&rp, sizeof(int32_t) * ((size_t)m->nrow + 1)) != hipSuccess)
        rc0 = SPMV_ERR_ALLOC;
    if (rc0 == SPMV_OK)
    {
        hipLaunchKernelGGL((ell_to_csr_kernel<false>), dim3(grid), dim3(kBlock), 0, ctx->stream, m->nrow, m->k, m->b, m->v, cnt, (const int32_t*)nullptr,
                           (int32_t*)nullptr, (double*)nullptr);
        if (hipGetLastError() != hipSuccess) rc0 = SPMV_ERR_HIP;
    }
    if (rc0 == SPMV_OK) rc0 = exclusive_scan_i32(ctx, cnt, rp, (int64_t)m->nrow + 1);
    if (rc0 == SPMV_OK && (hipMemcpyAsync(&total, rp + m->nrow, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                           hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc0 = SPMV_ERR_HIP;
    if (rc0 == SPMV_OK) rc0 = mat_alloc(ctx, SPMV_FMT_CSR, m->nrow, m->ncol, total, 0, (size_t)m->nrow + 1, (size_t)total, (size_t)total, &csr);
    if (rc0 == SPMV_OK)
    {
        (void)hipMemcpyAsync(const_cast<int32_t*>(csr->a), rp, sizeof(int32_t) * ((size_t)m->nrow + 1), hipMemcpyDeviceToDevice, ctx->stream);
        hipLaunchKernelGGL((ell_to_csr_kernel<true>), dim3(grid), dim3(kBlock), 0, ctx->stream, m->nrow, m->k, m->b, m->v, (int32_t*)nullptr, rp,
                           const_cast<int32_t*>(csr->b), const_cast<double*>(csr->v));
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (cnt) (void)hipFree(cnt);
    if (rp) (void)hipFree(rp);
    if (rc0 != SPMV_OK)
    {
        (void)hipGetLastError();
        if (csr) mat_free(csr);
        if (rc0 == SPMV_ERR_ALLOC) SPMV_FAIL(rc0, "no device memory for the row-grouped copy of an ELL handle (%lld slots)", (long long)((int64_t)m->nrow * m->k));
        SPMV_FAIL(rc0, "building the row-grouped copy of an ELL handle failed");
    }
    const int64_t slots = total;  // entries of the copy (below: what its col_ind / values hold)
    int rc = hipGetLastError() == hipSuccess ? SPMV_OK : SPMV_ERR_HIP;
    if (rc == SPMV_OK)
    {
        if (force_kernel != SPMV_CSR_AUTO)
        {
            csr->kernel_forced = true;
            csr->kernel        = force_kernel;
        }
        csr->pb_trial   = m->pb_trial;
        csr->sel_no_ell = true;  // (an ELL copy of the CSR copy of an ELL handle would be this handle again)
        plan_hand_down(m, csr, kPlanChildRowgrouped);
        rc              = csr_analyse(csr);  // row statistics, kernel (selected or forced), layout
    }
    if (rc != SPMV_OK)
    {
        mat_free(csr);
        if (rc == SPMV_ERR_HIP) set_error("building the row-grouped copy of an ELL handle failed");
        return rc;
    }
    // the panel and two-phase kernels read row_ptr and their own arrays only
    if ((csr->kernel == SPMV_CSR_PANEL || csr->kernel == SPMV_CSR_TWOPHASE || csr->kernel == SPMV_CSR_ELL) && csr->b && csr->v)
    {
        (void)hipFree(const_cast<int32_t*>(csr->b));
        (void)hipFree(const_cast<double*>(csr->v));
        csr->device_bytes -= slots * 12;
        csr->b = nullptr;
        csr->v = nullptr;
    }
    m->coo_csr = csr;
    m->kernel  = SPMV_CSR_PANEL;  // reported for ELL as "runs from the row-grouped copy"
    m->device_bytes += csr->device_bytes;
    return SPMV_OK;
}
int ell_own_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
}  // namespace

// the row-grouped copy with the PANEL kernel forced on it (spmv_mat_set_kernel(ell, SPMV_CSR_PANEL)); only_if_worth: the
// model's gate (no launches): >= 2M slots whose row blocks span more than 1 MiB of an x beyond L2
int ell_build_panel(spmv_mat* m, bool only_if_worth)
{
    if (m->coo_csr && m->coo_csr->kernel == SPMV_CSR_PANEL)
    {
        m->kernel = SPMV_CSR_PANEL;
        return SPMV_OK;
    }
    const int64_t slots = (int64_t)m->nrow * m->k;
    if (slots == 0 || slots > (int64_t)INT32_MAX - 65536) return SPMV_OK;
    if (only_if_worth)
    {
        if (slots < ((int64_t)2 << 20) || m->k < 2 || (double)m->ncol * 8.0 <= 4.0 * 1048576.0) return SPMV_OK;
        const double span = ell_mean_block_span(m);  // how far apart are the columns of 256 consecutive rows, on average?
        if (span < 0.0) SPMV_FAIL(SPMV_ERR_HIP, "ELL window scan failed: %s", hipGetErrorString(hipGetLastError()));
        if (span <= 131072.0) return SPMV_OK;  // a row block's x window is at most 1 MiB: stays in L2
    }
    ell_drop_rowgrouped(m);
    return ell_make_rowgrouped(m, SPMV_CSR_PANEL);
}

// AUTO for an ELL handle (select.hip; tools/sweep_structures.py).  The format's own kernels: two rows per lane (with the
// slots' diagonals where they were found: no index stream), one row per lane (more wavefronts: wins below ~200K rows, 2x on a
// 27-point stencil of 64000 rows), two rows per lane reading every index.  And the row-grouped copy, a candidate where one
// lane per row cannot work: FEW LONG ROWS (5000 rows x 160 slots: 0.058 ms against 0.008 - a lane walks 160 slots while
// most of the chip idles) or SCATTERED columns (a row block spanning more than 16 columns per row: 4M x 100K uniform,
// 0.165 against 0.084; the panel kernel orders the gathers by x line) - not where the slots are diagonals (their x
// stretches go through LDS) unless the rows are few.  The candidates are timed; without trials the model's gate decides.
int ell_select_kernel(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    select_reset(m);
    ell_drop_rowgrouped(m);
    m->kernel      = SPMV_CSR_VECTOR;
    m->ell_variant = 0;
    const int64_t slots = (int64_t)m->nrow * m->k;
    if (slots == 0 || m->nrow <= 0) return SPMV_OK;
    if (!select_trials_enabled(m) || slots < kSelectMinNnz)
        return (m->ell_diag || m->sel_no_rowgrouped) ? SPMV_OK : ell_build_panel(m, /*only_if_worth=*/true);
    select_scratch sv;
    if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return (m->ell_diag || m->sel_no_rowgrouped) ? SPMV_OK : ell_build_panel(m, true);
    // is the row-grouped copy a candidate?  (decided from the matrix, not from a timing: it is built BEFORE anything is timed, so
    // that no allocation falls between two timings - select.hip)
    bool candidate = !m->sel_no_rowgrouped && slots <= (int64_t)INT32_MAX - 65536;  // (the ELL copy of a CSR handle: that handle IS the row-grouped form)
    if (candidate)
    {
        const bool few_rows = m->nrow <= 65536 && m->k >= 16;
        // rows of unequal length (a finite-element mesh: 16 to 64 entries per row in 64 slots): a lane per row walks the padding
        // of the short rows while its neighbours work - 200000 rows, 8M entries in 12.8M slots: 0.046 ms against 0.032 through
        // the copy, which carries the same padding but spreads it over a workgroup
        const bool ragged = m->nnz > 0 && slots * 5 > m->nnz * 6;
        if (!few_rows && !ragged)
        {
            if (m->ell_diag)
                candidate = false;
            else
            {
                const double span = ell_mean_block_span(m);
                candidate         = span > 16.0 * 256.0;
            }
        }
    }
    int rc = SPMV_OK;
    if (candidate)
    {
        rc = ell_make_rowgrouped(m, SPMV_CSR_AUTO);
        if (rc == SPMV_ERR_ALLOC)
        {
            (void)hipGetLastError();
            rc = SPMV_OK;  // no memory for the copy: the format's own kernels are the candidates
        }
        if (rc != SPMV_OK) return rc;
    }
    // candidates 0 .. 2: the format's own variants; 3: the copy.  Timed in rounds until their minima stand still (select.hip:
    // this is what replaced round 5's 2 ms of sleep in front of small handles' trials)
    const bool has_v2 = m->ell_diag && m->ell_diag_mask;  // (without diagonal slots variant 0 reads the indices already)
    // the DIA-order copy of the values (variant 3), where the slots are diagonals and it is not forbidden ("ell_dia_order" 0)
    if (m->ell_dia_order_req != 0) SPMV_TRY(ell_build_dia_order(m, /*only_if_worth=*/m->ell_dia_order_req < 0));
    int        ids[5], n = 0;  // 0, 1, 2: the format's own variants; 4: the DIA-order copy (variant 3); 3: the row-grouped copy, last
    ids[n++] = 0;
    ids[n++] = 1;
    if (has_v2) ids[n++] = 2;
    if (m->ell_rval) ids[n++] = 4;
    if (m->coo_csr) ids[n++] = 3;
    float t[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
    rc = select_rounds(ctx, n,
                       [&](int j) {
                           if (ids[j] == 3) return csr_apply(ctx, m->coo_csr, sv.x, sv.y);
                           m->ell_variant = ids[j] == 4 ? 3 : ids[j];
                           return ell_own_apply(ctx, m, sv.x, sv.y);
                       },
                       t, &m->sel_rounds);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != SPMV_OK) return rc;
    float best_ms = 1e30f;
    int   best_v  = 0;
    for (int j = 0; j < n; ++j)
    {
        if (ids[j] == 3 || t[j] < 0.f) continue;
        const int v = ids[j] == 4 ? 3 : ids[j];
        select_note(m, v == 0 ? SPMV_CSR_VECTOR : (v == 3 ? 9 : 5 + v), t[j]);  // slots 1, 6 ("variant1"), 7 ("variant2"), 9 ("dia_order")
        // two rows per lane is the model's pick: another variant has to win by 2 % - the DIA-order copy, which costs 8 bytes per
        // slot of memory, by 5 %
        if (t[j] < best_ms * (v == 3 ? 0.95f : (v ? 0.98f : 1.0f)))
        {
            best_ms = t[j];
            best_v  = v;
        }
    }
    m->ell_variant = best_v;
    if (best_v != 3 && m->ell_dia_order_req < 0) ell_free_dia_order(m);  // (a copy that was asked for stays, used or not)
    if (m->coo_csr)
    {
        const float t_copy = t[n - 1];
        if (t_copy >= 0.f) select_note(m, SPMV_CSR_PANEL, t_copy);
        if (!(t_copy >= 0.f && t_copy < best_ms * 0.98f)) ell_drop_rowgrouped(m);
    }
    m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
    return SPMV_OK;
}

// ---- the ELL copy of a CSR handle (SPMV_CSR_ELL) --------------------------------------------------------------------
// tools/sweep_structures.py, every family: where the rows are (nearly) equally long - stencils, bands, block diagonals - the
// same matrix runs faster through an ELL handle than through a CSR handle's best kernel: a tridiagonal matrix of 8M rows
// 0.0557 against 0.0873 ms, a band of 33 0.105 against 0.163, dense 8 x 8 blocks 0.077 against 0.100, the 5- / 7-point stencils
// 0.048 / 0.065 against 0.056 / 0.074 (one lane per two rows streams values and indices coalesced, and slots recognised as
// diagonals read no index at all).  The reference's user calls CSRMatrixMatVector on whatever matrix she has
// (src/mat_vec.cpp:44-67): a CSR handle whose padding to its longest row stays below a quarter and that has no empty row
// therefore times an ELL copy of itself as one more candidate - where its columns are local (csr_ell_copy_worth).  The copy's padding carries value 0.0 and the row's own LAST
// column, not the reference's column 0 (convert.hip), so that the gathers issued for it stay inside x and near the row's other
// gathers - and since round 6 it takes NO part in the sums: the copy's kernels are the MASKED instances (ell_kernel above), which
// read a row's own length off the CSR handle's row_ptr.  Round 5's copy multiplied its padding: a row reading x[c] = inf got NaN
// (0.0 * inf) where the reference's CSR loop gets inf, and an empty row read x[0].
bool csr_ell_copy_worth(const spmv_mat* m)
{
    if (m->format != SPMV_FMT_CSR || m->sel_no_ell || m->nrow < 2 || m->nnz < kSelectMinNnz || !m->b || !m->v) return false;
    const int64_t slots = (int64_t)m->nrow * m->max_row_nnz;
    // local columns only: one lane per row gathers x like the row-parallel kernel does - fine while the x window of a row block
    // (mean over blocks of 256 rows) or all of x stays in an XCD's L2, hopeless on scattered columns (C2 has exactly 32 entries
    // in every row too: its handle must not build and time 3.84 GB of ELL copy to find that out)
    const bool local = m->win_avg_span * 8.0 <= 2.0 * 1048576.0 || (double)m->ncol * 8.0 <= 4.0 * 1048576.0;
    return local && m->min_row_nnz >= 1 && m->max_row_nnz <= 1024 && slots * 4 <= m->nnz * 5 && slots <= (int64_t)INT32_MAX - 65536;
}

void csr_ell_copy_free(spmv_mat* m)
{
    if (!m->ell_copy) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    m->device_bytes -= m->ell_copy->device_bytes;
    mat_free(m->ell_copy);
    m->ell_copy = nullptr;
}

int csr_ell_copy_build(spmv_mat* m)
{
    if (m->ell_copy || m->nnz == 0 || m->nrow == 0) return SPMV_OK;
    SPMV_REQUIRE(m->format == SPMV_FMT_CSR && m->a && m->b && m->v, "the ELL copy is made from a CSR handle's own arrays");
    SPMV_REQUIRE((int64_t)m->nrow * m->max_row_nnz <= (int64_t)INT32_MAX - 65536 && (int64_t)m->nrow * m->max_row_nnz <= 16 * std::max<int64_t>(m->nnz, 1),
                 "an ELL copy of %d rows x %d slots for %lld entries: the padding is out of proportion", m->nrow, m->max_row_nnz, (long long)m->nnz);
    spmv_mat* ell = nullptr;
    SPMV_TRY(csr_to_ell(m->ctx, m, &ell, /*pad_own_column=*/true));
    // the copy's padding slots carry a negative column (convert.hip: PAD_OWN) and its kernels - the MASKED instances - leave them out
    // of the sums: the copy is the CSR matrix in non-finite arithmetic too (a row that reads x[c] = inf ends at inf, a row without
    // entries keeps its y)
    ell->ell_pad_marked = true;
    ell->pb_trial          = m->pb_trial;
    ell->sel_no_rowgrouped = true;
    plan_hand_down(m, ell, kPlanChildEll);
    const int rc           = ell_analyse(ell);  // diagonal slots, the variant that is timed fastest
    if (rc != SPMV_OK)
    {
        mat_free(ell);
        return rc;
    }
    m->ell_copy = ell;
    m->device_bytes += ell->device_bytes;
    return SPMV_OK;
}

// are the slots diagonals?  kept when at least half of the entries lie in conforming row pairs
static int ell_detect_diagonals(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (m->ell_diag || m->nrow < 4 * kBlock || m->nrow % 2 != 0 || m->k < 1 || m->k > 4096 || ((uintptr_t)m->b % 8) != 0) return SPMV_OK;
    const int     k      = m->k;
    const int64_t npairs = m->nrow / 2, nwaves = ceil_div(npairs, 64);
    int32_t*      off    = nullptr;
    u64*          mask   = nullptr;
    SPMV_TRY(ensure_scratch(ctx, 256));
    u64*     d_cov  = (u64*)ctx->scratch;
    int32_t* d_hits = (int32_t*)((char*)ctx->scratch + 64);
    u64      h_cov  = 0;
    if (hipMalloc(&off, sizeof(int32_t) * (size_t)k) != hipSuccess || hipMalloc(&mask, sizeof(u64) * (size_t)(nwaves * k)) != hipSuccess)
    {
        if (off) (void)hipFree(off);
        SPMV_FAIL(SPMV_ERR_ALLOC, "out of device memory for the slot descriptors of an ELL handle (%lld words)", (long long)(nwaves * k));
    }
    std::vector<int32_t> h_off((size_t)k);
    // the reference row: of 16 rows from the middle on, the one most of 2048 sampled rows agree with
    int ref_row = m->nrow / 2;
    {
        int32_t    h_hits[kRefCandidates] = {0};
        const int  count = std::min(m->nrow, kRefSample), first = std::max(0, m->nrow / 2 - count / 2);
        hipError_t e0    = hipMemsetAsync(d_hits, 0, sizeof(h_hits), ctx->stream);
        hipLaunchKernelGGL(ell_ref_row_kernel, dim3((unsigned)ceil_div(count, kBlock)), dim3(kBlock), 0, ctx->stream, m->nrow, k, m->b, first, count, d_hits);
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(h_hits, d_hits, sizeof(h_hits), hipMemcpyDeviceToHost, ctx->stream);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(ctx->stream);
        if (e0 != hipSuccess)
        {
            (void)hipFree(off);
            (void)hipFree(mask);
            SPMV_FAIL(SPMV_ERR_HIP, "sampling the slots of an ELL handle failed: %s", hipGetErrorString(e0));
        }
        ref_row = ref_candidate(m->nrow, (int)(std::max_element(h_hits, h_hits + kRefCandidates) - h_hits));
    }
    hipError_t           e = hipMemsetAsync(d_cov, 0, sizeof(u64), ctx->stream);
    hipLaunchKernelGGL(ell_diag_scan_kernel, dim3((unsigned)ceil_div(nwaves * 64, kBlock)), dim3(kBlock), 0, ctx->stream, m->nrow, k, ref_row, m->b, off, mask,
                       d_cov);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_cov, d_cov, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_off.data(), off, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || (int64_t)h_cov * 2 < npairs * k)
    {
        (void)hipFree(off);
        (void)hipFree(mask);
        if (e != hipSuccess) SPMV_FAIL(SPMV_ERR_HIP, "scanning the slots of an ELL handle failed: %s", hipGetErrorString(e));
        return SPMV_OK;  // not a stencil / band: the column indices are needed
    }
    // clusters of nearby offsets share one stretch of x in LDS (stage_x_windows); at most 40 KB in all, else no windows
    std::vector<int> order((size_t)k);
    for (int s2 = 0; s2 < k; ++s2) order[(size_t)s2] = s2;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return h_off[(size_t)a] < h_off[(size_t)b]; });
    std::vector<int32_t> desc((size_t)2 * k + 1, 0);
    for (int s2 = 0; s2 < k; ++s2) desc[(size_t)s2] = h_off[(size_t)s2];
    std::vector<int32_t> cl;  // lo, len, base per cluster
    int64_t total = 0;
    for (int q = 0; q < k;)
    {
        int e2 = q;
        while (e2 + 1 < k && (int64_t)h_off[(size_t)order[(size_t)e2 + 1]] - h_off[(size_t)order[(size_t)q]] <= 1024) ++e2;
        const int lo  = h_off[(size_t)order[(size_t)q]] & ~1;  // even start and length: 16-byte staging loads
        const int len = (2 * kBlock + (h_off[(size_t)order[(size_t)e2]] - lo) + 1) & ~1;
        for (int t = q; t <= e2; ++t) desc[(size_t)k + order[(size_t)t]] = (int32_t)total + (h_off[(size_t)order[(size_t)t]] - lo);
        cl.push_back(lo);
        cl.push_back(len);
        cl.push_back((int32_t)total);
        total += len;
        q = e2 + 1;
    }
    const bool windows = total * 8 <= 40 * 1024;
    desc[(size_t)2 * k] = windows ? (int32_t)(cl.size() / 3) : 0;
    if (windows) desc.insert(desc.end(), cl.begin(), cl.end());
    int32_t* d_desc = nullptr;
    if (hipMalloc(&d_desc, sizeof(int32_t) * desc.size()) != hipSuccess ||
        hipMemcpyAsync(d_desc, desc.data(), sizeof(int32_t) * desc.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
    {
        if (d_desc) (void)hipFree(d_desc);
        (void)hipFree(off);
        (void)hipFree(mask);
        SPMV_FAIL(SPMV_ERR_ALLOC, "out of device memory for the slot descriptors of an ELL handle");
    }
    (void)hipFree(off);  // (the offsets live at the head of the descriptor)
    m->ell_diag      = d_desc;
    m->ell_diag_mask = mask;
    m->ell_diag_lds  = windows ? (int32_t)total : 0;
    m->device_bytes += (int64_t)sizeof(int32_t) * (int64_t)desc.size() + (int64_t)sizeof(u64) * nwaves * k;
    return SPMV_OK;
}

namespace
{
constexpr int kEllTileRows = 2 * kBlock;  // the rows of one workgroup of ell_diag_kernel_x2
__global__ __launch_bounds__(kBlock) void ell_tile_values_kernel(int nrow, int k, const double* __restrict__ val, double* __restrict__ tval)
{
    const int64_t slots = (int64_t)nrow * k;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < slots; e += (int64_t)gridDim.x * kBlock)
    {
        const int s = (int)(e / nrow), i = (int)(e - (int64_t)s * nrow);
        tval[((int64_t)(i / kEllTileRows) * k + s) * kEllTileRows + (i % kEllTileRows)] = val[e];
    }
}
}  // namespace

void ell_free_tiles(spmv_mat* m)
{
    if (!m->ell_tval) return;
    (void)hipFree(m->ell_tval);
    m->ell_tval = nullptr;
    m->device_bytes -= (int64_t)ceil_div(m->nrow, kEllTileRows) * kEllTileRows * m->k * (int64_t)sizeof(double);
}

// The values once more, in tiles of 512 rows, for the product over slots that are diagonals (the kernel's TILED form).  Costs
// 8 bytes per slot of device memory for 1-7 % of C3's time (see the kernel), so it is made only when asked for
// ("ell_tiled_values" = 1).
int ell_build_tiles(spmv_mat* m, bool only_if_worth)
{
    if (m->ell_tval || !m->ell_diag || m->nrow % 2 != 0) return SPMV_OK;
    const int64_t slots = (int64_t)m->nrow * m->k;
    if (only_if_worth && slots < ((int64_t)1 << 20)) return SPMV_OK;  // (a few megabytes: the strides do not matter)
    spmv_ctx*     ctx    = m->ctx;
    const int64_t padded = (int64_t)ceil_div(m->nrow, kEllTileRows) * kEllTileRows * m->k;
    double*       t      = nullptr;
    if (hipMalloc(&t, sizeof(double) * (size_t)padded) != hipSuccess)
    {
        (void)hipGetLastError();
        if (only_if_worth) return SPMV_OK;
        SPMV_FAIL(SPMV_ERR_ALLOC, "out of device memory for the tiled copy of the values of an ELL handle (%lld bytes)", (long long)(padded * 8));
    }
    hipError_t e = hipMemsetAsync(t, 0, sizeof(double) * (size_t)padded, ctx->stream);
    hipLaunchKernelGGL(ell_tile_values_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(slots, kBlock))), dim3(kBlock), 0, ctx->stream, m->nrow, m->k,
                       m->v, t);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
    {
        (void)hipFree(t);
        SPMV_FAIL(SPMV_ERR_HIP, "tiling the values of an ELL handle failed: %s", hipGetErrorString(e));
    }
    m->ell_tval = t;
    m->device_bytes += padded * (int64_t)sizeof(double);
    return SPMV_OK;
}

// ---- the DIA-order copy of an ELL handle whose slots are diagonals (ell_variant 3) -------------------------------------------
// BASELINE's C3 (4M rows x 64 slots, a circulant band) in one process, handles interleaved (tools/probe_c3_layouts.py,
// profiles/r06_probe_c3_layouts.txt): the ELL kernel over the column-major values 0.373-0.394 ms, over the values in tiles of
// 512 rows 0.338-0.393, the engine's DIA kernel over ROW-major values of the same band 0.316-0.336 - in memory allocated at the
// same moment.  A workgroup of the DIA kernel streams ONE contiguous stretch (256 rows x k values) in 128-byte lines through
// an LDS tile and takes x from an LDS window; the ELL kernel reads k stretches that lie nrow * 8 bytes apart.  So an ELL handle
// whose slots are diagonals may keep its values once more in DIA order - row * k + slot - and run the DIA kernel over them
// (kernels_misc.hip: dia_rows_apply).  Rows in which ANY slot is not its diagonal (C3: the 64 wrap-around rows; boundary rows of
// a stencil with their padding) are skipped there - a bit per row - and done by a side kernel over the handle's own column-major
// arrays.  Either way a row's products are added in slot order onto y[i] with fma: bit-identical to the ELL kernels and to the
// oracle's orc_ell_spmv_fma.  Cost: 8 bytes per slot of device memory (C3: +2.05 GB on 3.07 GB).  A candidate of the trial
// (ell_select_kernel), never a default by model; "ell_dia_order" 1 / 0 forces / forbids it.
namespace
{
constexpr int kRmChunk = 16;  // slots per tile of the transposition

// rval[i * kp + s] = val[i + s * nrow] (kp: the row stride, k rounded up to even): tiles of 256 rows x 16 slots through LDS, both
// sides in whole 128-byte lines
__global__ __launch_bounds__(kBlock) void ell_rowmajor_kernel(int nrow, int k, int kp, const double* __restrict__ val, double* __restrict__ rval)
{
    __shared__ double tile[kBlock * (kRmChunk + 1)];
    const int r0 = blockIdx.x * kBlock, s0 = blockIdx.y * kRmChunk;
    const int i  = r0 + (int)threadIdx.x;
    for (int j = 0; j < kRmChunk; ++j) tile[threadIdx.x * (kRmChunk + 1) + j] = (i < nrow && s0 + j < k) ? val[(size_t)i + (size_t)(s0 + j) * nrow] : 0.0;
    __syncthreads();
    const int d = threadIdx.x % kRmChunk, rr = threadIdx.x / kRmChunk;
    for (int p = 0; p < kRmChunk; ++p)
    {
        const int r = rr + p * (kBlock / kRmChunk);
        if (r0 + r < nrow && s0 + d < k) rval[(size_t)(r0 + r) * kp + s0 + d] = tile[r * (kRmChunk + 1) + d];
    }
}

// skip[i / 64] bit i % 64 = row i has a slot that is not its diagonal (col != i + off[s]); the grid covers whole words
__global__ __launch_bounds__(kBlock) void ell_row_conform_kernel(int nrow, int k, const int32_t* __restrict__ col, const int32_t* __restrict__ off,
                                                                 u64* __restrict__ skip)
{
    const int i  = blockIdx.x * kBlock + (int)threadIdx.x;
    bool      nc = false;
    if (i < nrow)
        for (int s = 0; s < k; ++s)
        {
            const int c = col[(size_t)i + (size_t)s * nrow];
            nc |= c < 0 || c != i + off[s];  // (c < 0: a padding slot of a CSR handle's ELL copy: the row goes to the side kernel, which leaves it out)
        }
    const u64 b = __ballot(nc);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (nrow + 63) / 64) skip[i >> 6] = b;
}

// the rows the DIA pass skipped, over the column-major arrays: one WAVEFRONT per listed row.  The 64 lanes fetch 64 slots' values,
// columns and x at once; the sum is then formed in slot order - acc = fma(v_s, x_s, acc), s = 0, 1, ... - from lane to lane, so
// that the bits are the ELL kernels' (one lane per row walking its slots alone took 25 us for C3's 63 rows x 64 slots: a chain of
// 64 dependent gathers - a twelfth of the DIA pass it follows; this form takes the 3 us of a launch)
__global__ __launch_bounds__(kBlock) void ell_rows_list_kernel(int nlist, const int32_t* __restrict__ rows, int nrow, int k, const int32_t* __restrict__ col,
                                                               const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    const int t = (int)((blockIdx.x * kBlock + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
    if (t >= nlist) return;  // (uniform over the wavefront)
    const int i   = rows[t];
    double    acc = y[i];
    for (int s0 = 0; s0 < k; s0 += kWave)
    {
        const int s = s0 + lane;
        double    v = 0.0, xv = 0.0;
        int       c = 0;  // (c < 0: a padding slot of a CSR handle's ELL copy - no part of the sum; a true ELL handle has none)
        if (s < k)
        {
            c  = col[(size_t)i + (size_t)s * nrow];
            v  = val[(size_t)i + (size_t)s * nrow];
            xv = x[c < 0 ? ~c : c];
        }
        const int n = min(kWave, k - s0);
        for (int j = 0; j < n; ++j)  // every lane forms the same sum, in slot order
        {
            const double vj = __shfl(v, j), xj = __shfl(xv, j);
            if (__shfl(c, j) >= 0) acc = fma(vj, xj, acc);
        }
    }
    if (lane == 0) y[i] = acc;
}
}  // namespace

void ell_free_dia_order(spmv_mat* m)
{
    if (m->format != SPMV_FMT_ELL || !m->ell_rval) return;
    (void)hipFree(m->ell_rval);
    if (m->ell_skip) (void)hipFree(m->ell_skip);
    if (m->ell_nc_rows) (void)hipFree(m->ell_nc_rows);
    m->device_bytes -= (int64_t)sizeof(double) * m->nrow * (m->k + (m->k & 1)) + (int64_t)sizeof(u64) * ((m->nrow + 63) / 64) + (int64_t)sizeof(int32_t) * std::max(m->ell_nc_count, 1);
    m->ell_rval     = nullptr;
    m->ell_skip     = nullptr;
    m->ell_nc_rows  = nullptr;
    m->ell_nc_count = 0;
    if (m->ell_variant == 3) m->ell_variant = 0;
}

// only_if_worth: the trial's gate - diagonal slots found, a million slots and more (below, a product is launch latency),
// at most 1/16 of the rows non-conforming, and device memory to spare (the copy is 8 bytes per slot)
int ell_build_dia_order(spmv_mat* m, bool only_if_worth)
{
    if (m->ell_rval) return SPMV_OK;
    spmv_ctx*     ctx   = m->ctx;
    const int64_t slots = (int64_t)m->nrow * m->k;
    if (!m->ell_diag || slots == 0 || !m->b || !m->v)
    {
        if (only_if_worth) return SPMV_OK;
        SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "ell_dia_order: the slots of this ELL handle were not found to be diagonals");
    }
    if (only_if_worth)
    {
        if (slots < ((int64_t)1 << 20) || m->k > 1024) return SPMV_OK;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)slots * 8 * 4 > free_b) return SPMV_OK;
    }
    const int         k = m->k, nrow = m->nrow, kp = m->k + (m->k & 1);
    const size_t      words = (size_t)(nrow + 63) / 64;
    std::vector<int32_t> h_off((size_t)k);
    std::vector<u64>     h_skip(words);
    double*   rval = nullptr;
    u64*      skip = nullptr;
    int32_t*  list = nullptr;
    int       rc   = SPMV_OK;
    do
    {
        if (hipMemcpyAsync(h_off.data(), m->ell_diag, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = SPMV_ERR_HIP; break; }
        if (hipMalloc(&skip, sizeof(u64) * words) != hipSuccess) { skip = nullptr; rc = SPMV_ERR_ALLOC; break; }
        hipLaunchKernelGGL(ell_row_conform_kernel, dim3((unsigned)ceil_div((int64_t)words * 64, kBlock)), dim3(kBlock), 0, ctx->stream, nrow, k, m->b, m->ell_diag, skip);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h_skip.data(), skip, sizeof(u64) * words, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = SPMV_ERR_HIP; break; }
        std::vector<int32_t> nc;
        for (size_t w = 0; w < words; ++w)
            for (u64 b = h_skip[w]; b; b &= b - 1) nc.push_back((int32_t)(w * 64 + (size_t)__builtin_ctzll(b)));
        if (only_if_worth && (int64_t)nc.size() * 16 > nrow) break;  // too many rows for the side kernel: not a candidate (rc OK, nothing built)
        // (rows padded to an even stride: the DIA kernel's 16-byte path needs every pair inside its row; the pad holds 0.0 and is never consumed)
        if (hipMalloc(&rval, sizeof(double) * (size_t)nrow * (size_t)kp) != hipSuccess) { rval = nullptr; rc = SPMV_ERR_ALLOC; break; }
        if (kp != k && hipMemsetAsync(rval, 0, sizeof(double) * (size_t)nrow * (size_t)kp, ctx->stream) != hipSuccess) { rc = SPMV_ERR_HIP; break; }
        if (hipMalloc(&list, sizeof(int32_t) * std::max<size_t>(nc.size(), 1)) != hipSuccess) { list = nullptr; rc = SPMV_ERR_ALLOC; break; }
        if (!nc.empty() && hipMemcpyAsync(list, nc.data(), sizeof(int32_t) * nc.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = SPMV_ERR_HIP; break; }
        hipLaunchKernelGGL(ell_rowmajor_kernel, dim3((unsigned)ceil_div(nrow, kBlock), (unsigned)ceil_div(k, kRmChunk)), dim3(kBlock), 0, ctx->stream, nrow, k, kp, m->v, rval);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = SPMV_ERR_HIP; break; }  // (nc, host, goes out of scope)
        m->ell_rval     = rval;
        m->ell_skip     = skip;
        m->ell_nc_rows  = list;
        m->ell_nc_count = (int32_t)nc.size();
        m->ell_off_min  = *std::min_element(h_off.begin(), h_off.end());
        m->ell_off_max  = *std::max_element(h_off.begin(), h_off.end());
        m->device_bytes += (int64_t)sizeof(double) * nrow * kp + (int64_t)sizeof(u64) * (int64_t)words + (int64_t)sizeof(int32_t) * std::max(m->ell_nc_count, 1);
        return SPMV_OK;
    } while (0);
    if (rval) (void)hipFree(rval);
    if (skip) (void)hipFree(skip);
    if (list) (void)hipFree(list);
    if (rc == SPMV_OK) return SPMV_OK;  // not worth it
    (void)hipGetLastError();
    if (only_if_worth && rc == SPMV_ERR_ALLOC) return SPMV_OK;  // no room: not a candidate
    if (rc == SPMV_ERR_ALLOC) SPMV_FAIL(rc, "out of device memory for the DIA-order copy of an ELL handle (%lld bytes)", (long long)(slots * 8));
    SPMV_FAIL(rc, "building the DIA-order copy of an ELL handle failed");
}

int ell_analyse(spmv_mat* m)
{
    m->kernel = SPMV_CSR_VECTOR;  // reported for ELL as "one lane per row"
    SPMV_TRY(ell_detect_diagonals(m));
    const bool from_ctx = plan_take_armed(m);
    if (plan_of(m))
    {
        const int rc = ell_apply_plan(m);
        plan_clear(m);
        if (rc == SPMV_OK || !from_ctx) return rc;
        (void)hipGetLastError();  // (a context's plan that does not fit this matrix: the handle selects by itself)
    }
    if (!m->kernel_forced) SPMV_TRY(ell_select_kernel(m));
    return SPMV_OK;
}

// A plan on an ELL handle (plan.hip): the variant of the format's own kernel, the tiled values, or the row-grouped copy with
// the kernel and layout its own node names - no timing launch.  (Whether the slots are diagonals is found out from the matrix
// as always: that is analysis, not a decision.)
int ell_apply_plan(spmv_mat* m)
{
    const plan_node& p = *plan_of(m);
    select_reset(m);
    ell_drop_rowgrouped(m);
    m->kernel        = SPMV_CSR_VECTOR;
    m->ell_variant   = p.ell_variant >= 0 && p.ell_variant <= 3 ? p.ell_variant : 0;
    if (m->ell_variant == 3)
    {
        SPMV_TRY(ell_build_dia_order(m, /*only_if_worth=*/false));  // (refused where the slots are no diagonals: the plan does not fit)
        m->ell_variant = 3;
    }
    else
    {
        (void)hipStreamSynchronize(m->ctx->stream);
        ell_free_dia_order(m);
    }
    m->lanes_per_row = p.lanes_per_row;
    m->flags         = p.flags;
    if ((int64_t)m->nrow * m->k == 0) return SPMV_OK;
    if (p.kernel == SPMV_CSR_PANEL)
    {
        SPMV_TRY(ell_make_rowgrouped(m, SPMV_CSR_AUTO));  // (hands the copy's node down)
        m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
    }
    if (p.ell_tiled)
        SPMV_TRY(ell_build_tiles(m, /*only_if_worth=*/false));
    else
    {
        (void)hipStreamSynchronize(m->ctx->stream);
        ell_free_tiles(m);
    }
    return SPMV_OK;
}

int ell_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nrow == 0) return SPMV_OK;
    if (A->coo_csr && A->kernel == SPMV_CSR_PANEL) return csr_apply(ctx, A->coo_csr, x, y);
    return ell_own_apply(ctx, A, x, y);
}

namespace
{
int ell_own_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->ell_variant == 3 && A->ell_rval && A->ell_diag && !(A->flags & SPMV_FLAG_ELL_READ_COLUMNS))
    {
        // the DIA kernel over the row-major copy (every row whose slots are all diagonals), then the few other rows from the
        // column-major arrays; every column of a conforming row is a real column: the bound is ncol
        SPMV_TRY(dia_rows_apply(ctx, A->nrow, A->ncol, A->k, A->ell_diag, A->ell_rval, x, y, true, A->ell_off_min, A->ell_off_max, A->flags, A->ell_skip, A->k + (A->k & 1)));
        if (A->ell_nc_count > 0)
        {
            hipLaunchKernelGGL(ell_rows_list_kernel, dim3((unsigned)ceil_div((int64_t)A->ell_nc_count * kWave, kBlock)), dim3(kBlock), 0, ctx->stream, A->ell_nc_count, A->ell_nc_rows,
                               A->nrow, A->k, A->b, A->v, x, y);
            SPMV_HIP(hipGetLastError());
        }
        return SPMV_OK;
    }
    const bool aligned = (A->nrow % 2 == 0) && (((uintptr_t)A->b % 8) == 0) && (((uintptr_t)A->v % 16) == 0) &&
                         (((uintptr_t)y % 16) == 0);
    // lanes_per_row == 1 (spmv_mat_set_kernel) or the variant AUTO timed fastest (ell_variant 1) select the one-row kernel
    const bool x2 = aligned && !(A->lanes_per_row == 1) && A->ell_variant != 1;
    if (x2 && A->ell_diag && A->ell_diag_mask && !(A->flags & SPMV_FLAG_ELL_READ_COLUMNS) && A->ell_variant != 2)
    {
        const unsigned grid = (unsigned)ceil_div(A->nrow / 2, kBlock);
        const bool     xwin = A->ell_diag_lds > 0 && A->ell_diag_lds <= 5120;  // the x stretches of 512 rows fit 40 KB of LDS
        const size_t   lds  = xwin ? sizeof(double) * (size_t)A->ell_diag_lds : 0;
#define SPMV_ELL_DIAG_M(U, W, T, M)                                                                                                              \
    hipLaunchKernelGGL((ell_diag_kernel_x2<U, W, T, M>), dim3(grid), dim3(kBlock), lds, ctx->stream, A->nrow, A->k, A->b, T ? A->ell_tval : A->v, x, y, \
                       A->ell_diag, (const u64*)A->ell_diag_mask, A->ncol)
#define SPMV_ELL_DIAG(U, W)                                    \
    do                                                         \
    {                                                          \
        if (A->ell_pad_marked) /* the ELL copy of a CSR handle: padding left out */ \
        {                                                      \
            if (A->ell_tval)                                   \
                SPMV_ELL_DIAG_M(U, W, true, true);             \
            else                                               \
                SPMV_ELL_DIAG_M(U, W, false, true);            \
        }                                                      \
        else if (A->ell_tval)                                  \
            SPMV_ELL_DIAG_M(U, W, true, false);                \
        else                                                   \
            SPMV_ELL_DIAG_M(U, W, false, false);               \
    } while (0)
        // slots in flight per lane: 4 by default; lanes_per_row 4 / 8 select 8 / 2 (tools/tune.py ell: A/B)
        if (A->lanes_per_row == 4)
        {
            if (xwin) SPMV_ELL_DIAG(8, true); else SPMV_ELL_DIAG(8, false);
        }
        else if (A->lanes_per_row == 8)
        {
            if (xwin) SPMV_ELL_DIAG(2, true); else SPMV_ELL_DIAG(2, false);
        }
        else
        {
            if (xwin) SPMV_ELL_DIAG(4, true); else SPMV_ELL_DIAG(4, false);
        }
#undef SPMV_ELL_DIAG
#undef SPMV_ELL_DIAG_M
    }
    else if (x2)
    {
        const unsigned grid = (unsigned)ceil_div(A->nrow / 2, kBlock);
        // slots in flight per lane: 4 by default; lanes_per_row 4 / 8 select 8 / 2 (tools/tune.py ell: A/B)
#define SPMV_ELL_X2(U)                                                                                                                                     \
    do                                                                                                                                                     \
    {                                                                                                                                                      \
        if (A->ell_pad_marked)                                                                                                 \
            hipLaunchKernelGGL((ell_kernel_x2<U, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y);  \
        else                                                                                                                   \
            hipLaunchKernelGGL((ell_kernel_x2<U, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y); \
    } while (0)
        if (A->lanes_per_row == 4)
            SPMV_ELL_X2(8);
        else if (A->lanes_per_row == 8)
            SPMV_ELL_X2(2);
        else
            SPMV_ELL_X2(4);
#undef SPMV_ELL_X2
    }
    else
    {
        const unsigned grid = (unsigned)ceil_div(A->nrow, kBlock);
        if (A->ell_pad_marked)
            hipLaunchKernelGGL((ell_kernel<8, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y);
        else
            hipLaunchKernelGGL((ell_kernel<8, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y);
    }
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace
}  // namespace spmv
