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
#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
// UNROLL slots are fetched before the first fma so that UNROLL gathers of x are in flight per lane.
template <int UNROLL>
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
        for (int u = 0; u < UNROLL; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = fma(v[u], xv[u], acc);
        at += (size_t)UNROLL * stride;
    }
    for (; s < k; ++s)
    {
        acc = fma(load_stream(val + at), x[load_stream(col + at)], acc);
        at += stride;
    }
    y[i] = acc;
}

// Two adjacent rows per lane: 8-byte column loads and 16-byte value loads (1 KiB per wavefront
// instruction).  Needs nrow even so that every slot column starts 16-byte aligned.
template <int UNROLL>
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
            xa[u] = x[c[u].x];
            xb[u] = x[c[u].y];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            acc.x = fma(v[u].x, xa[u], acc.x);
            acc.y = fma(v[u].y, xb[u], acc.y);
        }
        at += (size_t)UNROLL * stride;
    }
    for (; s < k; ++s)
    {
        const i32x2   c = load_stream(reinterpret_cast<const i32x2*>(col + at));
        const f64x2   v = load_stream(reinterpret_cast<const f64x2*>(val + at));
        acc.x           = fma(v.x, x[c.x], acc.x);
        acc.y           = fma(v.y, x[c.y], acc.y);
        at += stride;
    }
    *reinterpret_cast<f64x2*>(y + i) = acc;
}
}  // namespace

int ell_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nrow == 0) return SPMV_OK;
    const bool aligned = (A->nrow % 2 == 0) && (((uintptr_t)A->b % 8) == 0) && (((uintptr_t)A->v % 16) == 0) &&
                         (((uintptr_t)y % 16) == 0);
    const bool x2 = aligned && !(A->lanes_per_row == 1);  // lanes_per_row==1 forces the one-row kernel
    if (x2)
    {
        const unsigned grid = (unsigned)ceil_div(A->nrow / 2, kBlock);
        hipLaunchKernelGGL(ell_kernel_x2<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y);
    }
    else
    {
        const unsigned grid = (unsigned)ceil_div(A->nrow, kBlock);
        hipLaunchKernelGGL(ell_kernel<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, A->nrow, A->k, A->b, A->v, x, y);
    }
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
