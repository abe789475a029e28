// solver.hip — the step around the hot path: the consumer the reference's vec_dot / vec_axpby
// (src/vec_vec.cpp:15-29, :31-94) and the `diagonal // for SymGS` fields (include/matrix.h:36,81) were written for is
// a Krylov iteration.  The reference never got as far as calling them; here is that loop, device-resident:
//
//   mat_apply_ex   y = A*x or y += A*x with the dot product w.y of the updated y riding along (fused into the
//                  write-back of the panel kernel: saves a pass over w and y; other kernels run a dot pass behind)
//   cg_solve       conjugate gradients for symmetric positive definite A, plain, Jacobi- or symmetric-Gauss-Seidel-
//                  preconditioned (symgs.hip).  Three launches per iteration
//                  (product+dot, x/r update+dot, direction update); alpha and beta are computed on the device from
//                  scalars that never leave it, so iterations queue up without a host round trip; the host looks at
//                  the residual every `check_every` iterations only.  (A hipGraph replay of the iterations in
//                  between was built and measured slower than plain launches: kept behind SPMV_CG_GRAPH=1.)
//
// Not part of the reference's API: the results are checked against the oracle's product (residual of the solution)
// in tests/test_gpu_solver.py.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
inline int stream_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock))); }

// sum over the workgroup, valid in thread 0
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double s_part[kBlock / kWave];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kBlock / kWave; ++w) total += s_part[w];
    return total;
}

// Totals of two slotted accumulators, for every thread of the workgroup: lanes 0-31 of the first wavefront read the
// slots of `a`, lanes 32-63 those of `b` (one load latency instead of a chain of 32), halves are summed, LDS broadcast.
__device__ __forceinline__ void slot_sum2_block(const double* a, const double* b, double* ta, double* tb)
{
    static_assert(kDotSlots == 32, "one slot per lane of a 32-lane half");
    __shared__ double s_tot[2];
    if (threadIdx.x < kWave)
    {
        const int    lane = threadIdx.x;
        const double v    = (lane < 32 ? a : b)[(lane & 31) * kDotStride];
        const double t    = group_sum_swizzle<32>(v);
        if ((lane & 31) == 0) s_tot[lane >> 5] = t;
    }
    __syncthreads();
    *ta = s_tot[0];
    *tb = s_tot[1];
    __syncthreads();  // s_tot may be reused (block_sum has its own array)
}

// the same for four accumulators: the first two wavefronts read, one barrier pair
__device__ __forceinline__ void slot_sum4_block(const double* a, const double* b, const double* c, const double* d, double (&t)[4])
{
    static_assert(kDotSlots == 32 && kBlock >= 2 * kWave, "one slot per lane of a 32-lane half, two wavefronts");
    __shared__ double s_tot4[4];
    if (threadIdx.x < 2 * kWave)
    {
        const int     lane = threadIdx.x & 63, which = (int)(threadIdx.x >> 5);
        const double* acc  = which == 0 ? a : which == 1 ? b : which == 2 ? c : d;
        const double  v    = acc[(lane & 31) * kDotStride];
        const double  tot  = group_sum_swizzle<32>(v);
        if ((lane & 31) == 0) s_tot4[which] = tot;
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) t[i] = s_tot4[i];
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void dot_accumulate_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                                int64_t n, double* __restrict__ out)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        acc = fma(x[i], y[i], acc);
    const double total = block_sum(acc);
    if (threadIdx.x == 0) slot_add(out, total);
}

// Scalars of the iteration, on the device, each a slotted accumulator (common.hpp: kDotSlots partial sums).
// rr[k & 3] = r_k . r_k, pq[k & 3] = p_k . A p_k; ring slot k + 2 is cleared during iteration k, long after its last
// reader and before its next writer.
struct CgScalars
{
    double rr[4][kDotDoubles];
    double pq[4][kDotDoubles];
    double rz[4][kDotDoubles];  // Jacobi-preconditioned runs: r_k . z_k with z = D^-1 r (else unused: z = r, r.z = r.r)
    double bb[kDotDoubles];  // b . b
    double status;           // != 0: breakdown (p . A p <= 0: the matrix is not positive definite)
    double alpha[4];         // two-launch iteration: alpha_k, kept for the recurrence of iteration k + 1
    double noise_floor;      // two-launch iteration: 1e-28 b.b; at or below it r.r is rounding noise and the updates stop
};

// r = b - q (q = A x0), p = z = r (PRE: D^-1 r), rr[0] = r.r, rz[0] = r.z, bb = b.b
template <bool PRE>
__global__ __launch_bounds__(kBlock) void cg_init_kernel(int64_t n, const double* __restrict__ b, const double* __restrict__ q,
                                                         double* __restrict__ r, double* __restrict__ p, CgScalars* __restrict__ s,
                                                         const double* __restrict__ dinv)
{
    double rr = 0.0, bb = 0.0, rz = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        const double bi = b[i];
        const double ri = bi - q[i];
        const double zi = PRE ? ri * dinv[i] : ri;
        r[i]            = ri;
        p[i]            = zi;
        rr              = fma(ri, ri, rr);
        bb              = fma(bi, bi, bb);
        if (PRE) rz = fma(ri, zi, rz);
    }
    const double t_rr = block_sum(rr);
    __syncthreads();
    const double t_bb = block_sum(bb);
    __syncthreads();
    const double t_rz = PRE ? block_sum(rz) : 0.0;
    if (threadIdx.x == 0)
    {
        slot_add(s->rr[0], t_rr);
        slot_add(s->bb, t_bb);
        if (PRE) slot_add(s->rz[0], t_rz);
    }
}

// Jacobi-preconditioned pair (scalar width): alpha = rz_k / pq_k; x += alpha p; r -= alpha q; rr_{k+1} += r.r;
// rz_{k+1} += r . D^-1 r   and   beta = rz_{k+1} / rz_k; p = D^-1 r + beta p
__global__ __launch_bounds__(kBlock) void pcg_update_kernel(int64_t n, int k, const double* __restrict__ p,
                                                            const double* __restrict__ q, double* __restrict__ x,
                                                            double* __restrict__ r, CgScalars* __restrict__ s,
                                                            const double* __restrict__ dinv)
{
    double pq, rz_k;
    slot_sum2_block(s->pq[k & 3], s->rz[k & 3], &pq, &rz_k);
    if (!(pq > 0.0))
    {
        // Breakdown (the matrix is not positive definite) only while there is a residual to speak of.  Once r is zero
        // or has shrunk to where r.r and p.Ap underflow (the system was solved between two looks of the host), p.Ap = 0
        // is the end of the iteration, not an error: x and r stay, r.r of the next iteration stays at its cleared 0,
        // so every queued iteration ends here as well and the host reads a residual of 0.
        if (blockIdx.x == 0 && threadIdx.x == 0 && slot_sum(s->rr[k & 3]) > 1e-60 * slot_sum(s->bb)) s->status = 1.0;
        return;
    }
    const double alpha = rz_k / pq;
    double       rr = 0.0, rz = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        x[i]            = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i]            = ri;
        rr              = fma(ri, ri, rr);
        rz              = fma(ri * dinv[i], ri, rz);
    }
    const double t_rr = block_sum(rr);
    __syncthreads();
    const double t_rz = block_sum(rz);
    if (threadIdx.x == 0)
    {
        slot_add(s->rr[(k + 1) & 3], t_rr);
        slot_add(s->rz[(k + 1) & 3], t_rz);
    }
}

__global__ __launch_bounds__(kBlock) void pcg_direction_kernel(int64_t n, int k, const double* __restrict__ r,
                                                               double* __restrict__ p, CgScalars* __restrict__ s,
                                                               const double* __restrict__ dinv)
{
    double rz_k, rz_next;
    slot_sum2_block(s->rz[k & 3], s->rz[(k + 1) & 3], &rz_k, &rz_next);
    const double beta = rz_k > 0.0 ? rz_next / rz_k : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        p[i] = fma(beta, p[i], r[i] * dinv[i]);
    if (blockIdx.x == 0 && threadIdx.x < kDotSlots)
    {
        s->rr[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->rz[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->pq[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
    }
}

// General preconditioner (z = M^-1 r comes from a sweep between the two kernels, r . z from a dot pass behind it):
// alpha = rz_k / pq_k; x += alpha p; r -= alpha q; rr_{k+1} += r.r    and    beta = rz_{k+1} / rz_k; p = z + beta p
__global__ __launch_bounds__(kBlock) void gcg_update_kernel(int64_t n, int k, const double* __restrict__ p, const double* __restrict__ q,
                                                            double* __restrict__ x, double* __restrict__ r, CgScalars* __restrict__ s)
{
    double pq, rz_k;
    slot_sum2_block(s->pq[k & 3], s->rz[k & 3], &pq, &rz_k);
    if (!(pq > 0.0))
    {
        // a breakdown only while there is a residual to speak of (see pcg_update_kernel)
        if (blockIdx.x == 0 && threadIdx.x == 0 && slot_sum(s->rr[k & 3]) > 1e-60 * slot_sum(s->bb)) s->status = 1.0;
        return;
    }
    const double alpha = rz_k / pq;
    double       rr    = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        x[i]            = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i]            = ri;
        rr              = fma(ri, ri, rr);
    }
    const double total = block_sum(rr);
    if (threadIdx.x == 0) slot_add(s->rr[(k + 1) & 3], total);
}

__global__ __launch_bounds__(kBlock) void gcg_direction_kernel(int64_t n, int k, const double* __restrict__ z, double* __restrict__ p,
                                                               CgScalars* __restrict__ s)
{
    double rz_k, rz_next;
    slot_sum2_block(s->rz[k & 3], s->rz[(k + 1) & 3], &rz_k, &rz_next);
    const double beta = rz_k > 0.0 ? rz_next / rz_k : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = fma(beta, p[i], z[i]);
    if (blockIdx.x == 0 && threadIdx.x < kDotSlots)
    {
        s->rr[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->rz[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->pq[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
    }
}

// 1 / a_ii of a CSR handle (duplicates of the diagonal entry are summed, as the product would); flag != 0: a zero or
// missing diagonal entry
__global__ __launch_bounds__(kBlock) void csr_inv_diag_kernel(int nrow, int64_t row_begin, const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ col, const double* __restrict__ val,
                                                              double* __restrict__ dinv, int* __restrict__ flag)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrow) return;
    double    d   = 0.0;
    const int end = row_ptr[i + 1];
    for (int j = row_ptr[i]; j < end; ++j)
        if ((int64_t)col[j] == row_begin + i) d += val[j];
    if (d == 0.0)
    {
        atomicOr(flag, 1);
        d = 1.0;
    }
    dinv[i] = 1.0 / d;
}

// alpha = rr_k / pq_k;  x += alpha p;  r -= alpha q;  rr_{k+1} += r.r
__global__ __launch_bounds__(kBlock) void cg_update_kernel(int64_t n, int k, const double* __restrict__ p,
                                                           const double* __restrict__ q, double* __restrict__ x,
                                                           double* __restrict__ r, CgScalars* __restrict__ s)
{
    double pq, rr_k;
    slot_sum2_block(s->pq[k & 3], s->rr[k & 3], &pq, &rr_k);
    if (!(pq > 0.0))
    {
        // a breakdown only while there is a residual to speak of (see pcg_update_kernel)
        if (blockIdx.x == 0 && threadIdx.x == 0 && rr_k > 1e-60 * slot_sum(s->bb)) s->status = 1.0;
        return;  // uniform over the grid: every thread read the same scalar
    }
    const double alpha = rr_k / pq;
    double       rr    = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        x[i]            = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i]            = ri;
        rr              = fma(ri, ri, rr);
    }
    const double total = block_sum(rr);
    if (threadIdx.x == 0) slot_add(s->rr[(k + 1) & 3], total);
}

// The same two kernels with two elements per lane and 16-byte accesses (used when the vectors are 16-byte aligned);
// an odd last element is handled by one extra lane.
typedef double f64x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kBlock) void cg_update2_kernel(int64_t n, int k, const double* __restrict__ p,
                                                            const double* __restrict__ q, double* __restrict__ x,
                                                            double* __restrict__ r, CgScalars* __restrict__ s)
{
    double pq, rr_k;
    slot_sum2_block(s->pq[k & 3], s->rr[k & 3], &pq, &rr_k);
    if (!(pq > 0.0))
    {
        // a breakdown only while there is a residual to speak of (see pcg_update_kernel)
        if (blockIdx.x == 0 && threadIdx.x == 0 && rr_k > 1e-60 * slot_sum(s->bb)) s->status = 1.0;
        return;
    }
    const double  alpha  = rr_k / pq;
    const int64_t npairs = n / 2;
    double        rr     = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock)
    {
        const f64x2_t pv = ((const f64x2_t*)p)[i], qv = ((const f64x2_t*)q)[i];
        f64x2_t       xv = ((f64x2_t*)x)[i], rv = ((f64x2_t*)r)[i];
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            xv[e] = fma(alpha, pv[e], xv[e]);
            rv[e] = fma(-alpha, qv[e], rv[e]);
            rr    = fma(rv[e], rv[e], rr);
        }
        ((f64x2_t*)x)[i] = xv;
        ((f64x2_t*)r)[i] = rv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    {
        const int64_t i  = n - 1;
        x[i]             = fma(alpha, p[i], x[i]);
        const double ri  = fma(-alpha, q[i], r[i]);
        r[i]             = ri;
        rr               = fma(ri, ri, rr);
    }
    const double total = block_sum(rr);
    if (threadIdx.x == 0) slot_add(s->rr[(k + 1) & 3], total);
}

__global__ __launch_bounds__(kBlock) void cg_direction2_kernel(int64_t n, int k, const double* __restrict__ r,
                                                               double* __restrict__ p, CgScalars* __restrict__ s)
{
    double rr_k, rr_next;
    slot_sum2_block(s->rr[k & 3], s->rr[(k + 1) & 3], &rr_k, &rr_next);
    const double  beta   = rr_k > 0.0 ? rr_next / rr_k : 0.0;
    const int64_t npairs = n / 2;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock)
    {
        const f64x2_t rv = ((const f64x2_t*)r)[i];
        f64x2_t       pv = ((f64x2_t*)p)[i];
        pv[0]            = fma(beta, pv[0], rv[0]);
        pv[1]            = fma(beta, pv[1], rv[1]);
        ((f64x2_t*)p)[i] = pv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) p[n - 1] = fma(beta, p[n - 1], r[n - 1]);
    if (blockIdx.x == 0 && threadIdx.x < kDotSlots)
    {
        s->rr[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->pq[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
    }
}

// beta = rr_{k+1} / rr_k;  p = r + beta p;  clear the slots of iteration k + 2
__global__ __launch_bounds__(kBlock) void cg_direction_kernel(int64_t n, int k, const double* __restrict__ r,
                                                              double* __restrict__ p, CgScalars* __restrict__ s)
{
    double rr_k, rr_next;
    slot_sum2_block(s->rr[k & 3], s->rr[(k + 1) & 3], &rr_k, &rr_next);
    const double beta = rr_k > 0.0 ? rr_next / rr_k : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        p[i] = fma(beta, p[i], r[i]);
    if (blockIdx.x == 0 && threadIdx.x < kDotSlots)
    {
        s->rr[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->pq[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
    }
}
// ---- two launches per iteration (round 4) ----------------------------------------------------------------------------------
// The three-launch iteration needs r.r of the UPDATED residual before it can form beta, hence a kernel boundary between the
// x / r update and the direction update.  The Chronopoulos-Gear arrangement of the same recurrences moves the product onto
// the (preconditioned) residual and carries s = A p along, so that everything behind the product is ONE pass:
//   launch 1   w = A u (u = M^-1 r; u = r without a preconditioner), delta_k = u . w fused into the product's write-back
//   launch 2   beta = gamma_k / gamma_{k-1};  alpha = gamma_k / (delta_k - beta gamma_k / alpha_{k-1})
//              p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  u = D^-1 r;  gamma_{k+1} += r . u;  rr_{k+1} += r . r
// with gamma_k = r_k . u_k in rz[k & 3] (unpreconditioned: u = r and gamma = r . r).  Same iterates in exact arithmetic; the
// same bytes per iteration as the two kernels it replaces (5 reads + 4 writes of a vector against 6 + 3), one launch less:
// what small systems, which are launch-bound, are made of.  One more work vector (s).
template <bool PRE, bool WIDE>
__global__ __launch_bounds__(kBlock) void cg_fused_kernel(int64_t n, int k, const double* __restrict__ w, double* __restrict__ u,
                                                          double* __restrict__ p, double* __restrict__ sv, double* __restrict__ x,
                                                          double* __restrict__ r, CgScalars* __restrict__ s, const double* __restrict__ dinv)
{
    double (*const G)[kDotDoubles] = PRE ? s->rz : s->rr;  // gamma_k = r_k . u_k lives in rz; without a preconditioner it IS r . r
    double t[4];
    slot_sum4_block(G[k & 3], s->pq[k & 3], G[(k + 3) & 3], s->rr[k & 3], t);
    const double gamma = t[0], delta = t[1], gamma_old = t[2], rr_k = t[3];
    // the ring slots of iteration k + 2 are cleared whatever happens below (nobody reads or writes them in this launch): an
    // iteration that passes quietly must still leave r.r = 0 where the host will look for it
    if (blockIdx.x == 0 && threadIdx.x < kDotSlots)
    {
        s->rr[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->rz[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
        s->pq[(k + 2) & 3][threadIdx.x * kDotStride] = 0.0;
    }
    // Below |r| = 1e-14 |b| (s->noise_floor = 1e-28 b.b, written by the host once b.b is known) the residual is rounding noise
    // of the recurrence and this arrangement's alpha - a difference of two nearly equal numbers - is noise too: the iterations
    // queued behind an exactly solved system pass without touching anything.
    // (uniform over the grid: every thread read the same scalars)
    if (rr_k != rr_k)
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) s->status = 2.0;  // r.r is NaN: b, x0 or the matrix hold non-finite numbers
        return;
    }
    if (!(rr_k > s->noise_floor))
    {
        // the residual the recurrence ATTAINED travels on with the ring (slot k + 1 was cleared two launches ago and nobody
        // else writes it in a quiet launch), so the host reports ~1e-14 and not an exact 0 it never reached
        if (blockIdx.x == 0 && threadIdx.x == 0) s->rr[(k + 1) & 3][0] = rr_k;
        return;
    }
    // k = 0 needs no special case: gamma_old is the 0 the set-up wrote into slot 3 (beta_0 = 0), which is also what lets a
    // captured graph of four iterations be replayed at k = 4, 8, ... (k enters through k & 3 alone)
    const double beta  = gamma_old > 0.0 ? gamma / gamma_old : 0.0;
    const double denom = beta != 0.0 ? delta - beta * gamma / s->alpha[(k + 3) & 3] : delta;
    if (!(denom > 0.0) || !(gamma > 0.0))
    {
        // a residual to speak of and no descent direction: p.Ap <= 0, or r.M^-1 r <= 0 (a Jacobi diagonal with a negative
        // entry: -I), or one of them NaN - not positive definite, and never a quiet pass
        if (blockIdx.x == 0 && threadIdx.x == 0) s->status = 1.0;
        return;
    }
    const double alpha = gamma / denom;
    double       rr = 0.0, rz = 0.0;
    auto one = [&](double wi, double ui, double& pi, double& si, double& xi, double& ri, double di, double& u_out) {
        pi    = fma(beta, pi, ui);
        si    = fma(beta, si, wi);
        xi    = fma(alpha, pi, xi);
        ri    = fma(-alpha, si, ri);
        u_out = PRE ? ri * di : ri;
        rr    = fma(ri, ri, rr);
        if (PRE) rz = fma(ri, u_out, rz);
    };
    if constexpr (WIDE)
    {
        const int64_t npairs = n / 2;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * kBlock)
        {
            const f64x2_t wv = ((const f64x2_t*)w)[i];
            const f64x2_t uv = PRE ? ((const f64x2_t*)u)[i] : ((const f64x2_t*)r)[i];
            const f64x2_t dv = PRE ? ((const f64x2_t*)dinv)[i] : f64x2_t{1.0, 1.0};
            const f64x2_t pv = ((const f64x2_t*)p)[i], sw = ((const f64x2_t*)sv)[i], xv = ((const f64x2_t*)x)[i], rv = ((const f64x2_t*)r)[i];
            double        pe[2] = {pv[0], pv[1]}, se[2] = {sw[0], sw[1]}, xe[2] = {xv[0], xv[1]}, re[2] = {rv[0], rv[1]}, ue[2];
            one(wv[0], uv[0], pe[0], se[0], xe[0], re[0], dv[0], ue[0]);
            one(wv[1], uv[1], pe[1], se[1], xe[1], re[1], dv[1], ue[1]);
            ((f64x2_t*)p)[i]  = f64x2_t{pe[0], pe[1]};
            ((f64x2_t*)sv)[i] = f64x2_t{se[0], se[1]};
            ((f64x2_t*)x)[i]  = f64x2_t{xe[0], xe[1]};
            ((f64x2_t*)r)[i]  = f64x2_t{re[0], re[1]};
            if (PRE) ((f64x2_t*)u)[i] = f64x2_t{ue[0], ue[1]};
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        {
            const int64_t i = n - 1;
            double        uo;
            one(w[i], PRE ? u[i] : r[i], p[i], sv[i], x[i], r[i], PRE ? dinv[i] : 1.0, uo);
            if (PRE) u[i] = uo;
        }
    }
    else
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        {
            double uo;
            one(w[i], PRE ? u[i] : r[i], p[i], sv[i], x[i], r[i], PRE ? dinv[i] : 1.0, uo);
            if (PRE) u[i] = uo;
        }
    const double t_rr = block_sum(rr);
    __syncthreads();
    const double t_rz = PRE ? block_sum(rz) : 0.0;
    if (threadIdx.x == 0)
    {
        slot_add(s->rr[(k + 1) & 3], t_rr);
        if (PRE) slot_add(s->rz[(k + 1) & 3], t_rz);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) s->alpha[k & 3] = alpha;
}
}  // namespace

int vec_dot_accumulate(spmv_ctx* ctx, const double* x, const double* y, int64_t n, double* device_out)
{
    if (n == 0) return SPMV_OK;
    hipLaunchKernelGGL(dot_accumulate_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, x, y, n, device_out);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int mat_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    switch (A->format)
    {
        case SPMV_FMT_CSR: return csr_apply(ctx, A, x, y);
        case SPMV_FMT_ELL: return ell_apply(ctx, A, x, y);
        case SPMV_FMT_COO: return coo_apply(ctx, A, x, y);
        case SPMV_FMT_CSC: return csc_apply(ctx, A, x, y);
        case SPMV_FMT_DIA: return dia_apply(ctx, A, x, y);
        default: SPMV_FAIL(SPMV_ERR_INVALID, "unknown format %d", A->format);
    }
}

int mat_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex)
{
    // COO / ELL / CSC handles that run from their row-grouped copy: that copy is a CSR handle with a kernel of its own
    if ((A->format == SPMV_FMT_COO || A->format == SPMV_FMT_ELL || A->format == SPMV_FMT_CSC) && A->coo_csr && A->kernel == SPMV_CSR_PANEL && A->nnz > 0 &&
        A->nrow > 0 && A->ncol > 0)
        return mat_apply_ex(ctx, A->coo_csr, x, y, ex);
    // the panel kernel does all of it in its write-back
    if (A->format == SPMV_FMT_CSR && A->kernel == SPMV_CSR_PANEL && A->nrow > 0 && A->nnz > 0) return csr_panel_apply_ex(ctx, A, x, y, ex);
    if (A->format == SPMV_FMT_CSR && A->kernel == SPMV_CSR_TWOPHASE) return csr_twophase_apply_ex(ctx, A, x, y, ex);
    if (A->format == SPMV_FMT_CSR && A->kernel == SPMV_CSR_SPLIT && A->coo_csr)
    {
        // the short rows with the overwrite fused into their kernel, the long rows added on top, the dot product over the finished y
        apply_extra first;
        first.overwrite = ex.overwrite;
        SPMV_TRY(mat_apply_ex(ctx, A->coo_csr, x, y, first));
        SPMV_TRY(csr_split_long_rows_apply(ctx, A, x, y));
        if (ex.dot_w) SPMV_TRY(vec_dot_accumulate(ctx, ex.dot_w, y, A->nrow, ex.dot_out));
        return SPMV_OK;
    }
    int rc = SPMV_OK;
    if (A->format == SPMV_FMT_CSR && csr_vector_apply_ex(ctx, A, x, y, ex, &rc)) return rc;  // row-parallel kernel: fused too
    if (ex.overwrite) SPMV_TRY(vec_fill(ctx, y, A->nrow, 0.0));
    SPMV_TRY(mat_apply(ctx, A, x, y));
    if (ex.dot_w) SPMV_TRY(vec_dot_accumulate(ctx, ex.dot_w, y, A->nrow, ex.dot_out));
    return SPMV_OK;
}

int cg_solve(spmv_ctx* ctx, const spmv_mat* A, const double* b, double* x, int max_iter, double rel_tol, int check_every,
             int precond, int* iters, double* rel_resid)
{
    const int64_t n = A->nrow;
    *iters          = 0;
    *rel_resid      = 0.0;
    if (n == 0) return SPMV_OK;
    hipStream_t st = ctx->stream;
    double *    r = nullptr, *p = nullptr, *q = nullptr, *dinv = nullptr, *z = nullptr, *sv = nullptr, *u = nullptr;
    CgScalars*  s = nullptr;
    auto        release = [&]() {
        if (sv) (void)hipFree(sv);
        if (u) (void)hipFree(u);
        if (z) (void)hipFree(z);
        if (r) (void)hipFree(r);
        if (p) (void)hipFree(p);
        if (q) (void)hipFree(q);
        if (s) (void)hipFree(s);
        if (dinv) (void)hipFree(dinv);
    };
    if (precond == SPMV_PRECOND_SYMGS)
    {
        // one symmetric Gauss-Seidel sweep from z = 0 per iteration (symgs.hip); the plan is built once and stays in the handle
        SPMV_TRY(symgs_setup(const_cast<spmv_mat*>(A)));
        if (hipMalloc(&z, sizeof(double) * (size_t)n) != hipSuccess)
            SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_cg: out of device memory for the preconditioned residual (%lld entries)", (long long)n);
    }
    else if (precond)
    {
        // Jacobi: the diagonal the reference's containers carry "for SymGS" (include/matrix.h:36); taken from the CSR arrays
        if (A->format != SPMV_FMT_CSR || !A->b || !A->v)
            SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "spmv_cg: the Jacobi preconditioner reads the diagonal of a CSR handle");
        SPMV_TRY(ensure_scratch(ctx, 64));
        int* flag   = (int*)ctx->scratch;
        int  h_flag = 0;
        if (hipMalloc(&dinv, sizeof(double) * (size_t)n) != hipSuccess)
            SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_cg: out of device memory for the diagonal (%lld entries)", (long long)n);
        (void)hipMemsetAsync(flag, 0, sizeof(int), st);
        hipLaunchKernelGGL(csr_inv_diag_kernel, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, st, (int)n, A->row_begin,
                           A->a, A->b, A->v, dinv, flag);
        if (hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
            h_flag != 0)
        {
            release();
            SPMV_FAIL(SPMV_ERR_INVALID, "spmv_cg: the matrix has a zero or missing diagonal entry (Jacobi preconditioner)");
        }
    }
    if (hipMalloc(&r, sizeof(double) * (size_t)n) != hipSuccess || hipMalloc(&p, sizeof(double) * (size_t)n) != hipSuccess ||
        hipMalloc(&q, sizeof(double) * (size_t)n) != hipSuccess || hipMalloc(&s, sizeof(CgScalars)) != hipSuccess)
    {
        release();
        SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_cg: out of device memory for three work vectors of %lld entries", (long long)n);
    }
    // Two launches per iteration (cg_fused_kernel) unless a sweep stands between the product and the updates (symmetric
    // Gauss-Seidel) or SPMV_CG_THREE_LAUNCHES=1 asks for the textbook arrangement (A/B; read once per solve).
    const char* e_three = getenv("SPMV_CG_THREE_LAUNCHES");
    const bool  fused   = !z && !(e_three && e_three[0] == '1');
    if (fused && (hipMalloc(&sv, sizeof(double) * (size_t)n) != hipSuccess || (dinv && hipMalloc(&u, sizeof(double) * (size_t)n) != hipSuccess)))
    {
        release();
        SPMV_FAIL(SPMV_ERR_ALLOC, "spmv_cg: out of device memory for the work vectors of %lld entries", (long long)n);
    }
    const int  grid  = stream_grid(n);
    const int  grid2 = stream_grid(std::max<int64_t>(1, n / 2));
    const bool wide  = (((uintptr_t)x) & 15) == 0 && n >= 2;  // r, p, q are fresh allocations (256-byte aligned)
    int        rc    = SPMV_OK;
    std::vector<double> hbuf(sizeof(CgScalars) / sizeof(double));
    CgScalars&          h = *reinterpret_cast<CgScalars*>(hbuf.data());
    auto                host_sum = [](const double* acc) {
        double t = 0.0;
        for (int i = 0; i < kDotSlots; ++i) t += acc[i * kDotStride];
        return t;
    };
    // ring < 0: everything (once, after the set-up); else the r.r accumulator of that ring slot and the status word
    auto      fetch = [&](int ring) -> int {
        hipError_t e = hipSuccess;
        if (ring < 0)
            e = hipMemcpyAsync(&h, s, sizeof(CgScalars), hipMemcpyDeviceToHost, st);
        else
        {
            e = hipMemcpyAsync(h.rr[ring], s->rr[ring], sizeof(double) * kDotDoubles, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(&h.status, &s->status, sizeof(double), hipMemcpyDeviceToHost, st);
        }
        if (e != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        {
            set_error("spmv_cg: reading the iteration scalars failed: %s", hipGetErrorString(hipGetLastError()));
            return SPMV_ERR_HIP;
        }
        return SPMV_OK;
    };
    do
    {
        if (hipMemsetAsync(s, 0, sizeof(CgScalars), st) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        apply_extra first;
        first.overwrite = true;
        if ((rc = mat_apply_ex(ctx, A, x, q, first)) != SPMV_OK) break;  // q = A x0
        if (dinv)
            hipLaunchKernelGGL(cg_init_kernel<true>, dim3(grid), dim3(kBlock), 0, st, n, b, q, r, p, s, dinv);
        else
            hipLaunchKernelGGL(cg_init_kernel<false>, dim3(grid), dim3(kBlock), 0, st, n, b, q, r, p, s, dinv);
        if (z)
        {
            // z_0 = M^-1 r_0, rz_0 = r_0 . z_0, p_0 = z_0
            if ((rc = symgs_sweep(ctx, A, r, z, true)) != SPMV_OK) break;
            if ((rc = vec_dot_accumulate(ctx, r, z, n, s->rz[0])) != SPMV_OK) break;
            if (hipMemcpyAsync(p, z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
        }
        if (fused)
        {
            // s = A p starts at 0 (beta_0 = 0 multiplies it); the preconditioned residual u_0 = z_0 is what init left in p
            if (hipMemsetAsync(sv, 0, sizeof(double) * (size_t)n, st) != hipSuccess ||
                (u && hipMemcpyAsync(u, p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess))
            {
                rc = SPMV_ERR_HIP;
                break;
            }
        }
        if ((rc = fetch(-1)) != SPMV_OK) break;
        const double bb    = host_sum(h.bb);
        const double limit = rel_tol * rel_tol * bb;  // compare squared norms
        double       rr    = host_sum(h.rr[0]);
        int          k     = 0;
        if (!std::isfinite(bb) || !std::isfinite(rr))
        {
            set_error("spmv_cg: b.b = %g, r0.r0 = %g: b, x0 or the matrix hold non-finite numbers", bb, rr);
            rc = SPMV_ERR_INVALID;
            break;
        }
        if (!(bb > 0.0) || rr <= limit)
        {
            *rel_resid = bb > 0.0 ? sqrt(rr / bb) : 0.0;  // b = 0: x0 solves it if r = 0 (else the caller sees iters = 0)
            break;
        }
        if (fused)
        {
            const double floor_rr = 1e-28 * bb;
            if (hipMemcpyAsync(&s->noise_floor, &floor_rr, sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
        }
        const int every = std::max(1, check_every);
        // one iteration = two (fused) or three launches on the stream (k enters the kernels only through k & 3 and k > 0)
        auto iteration = [&](int kk) -> int {
            if (fused)
            {
                const double* in = u ? u : r;  // the product runs on the (preconditioned) residual
                apply_extra   fx;
                fx.overwrite = true;
                fx.dot_w     = in;
                fx.dot_out   = s->pq[kk & 3];
                SPMV_TRY(mat_apply_ex(ctx, A, in, q, fx));  // w = A u, delta_k = u . w
                const int kq = kk & 3;  // (k & 3 is all the kernel looks at: a captured graph of four iterations replays at any k % 4 == 0)
#define SPMV_CG_FUSED(PRE, WIDE, GRID) \
    hipLaunchKernelGGL((cg_fused_kernel<PRE, WIDE>), dim3(GRID), dim3(kBlock), 0, st, n, kq, q, u, p, sv, x, r, s, dinv)
                if (dinv)
                {
                    if (wide) SPMV_CG_FUSED(true, true, grid2); else SPMV_CG_FUSED(true, false, grid);
                }
                else
                {
                    if (wide) SPMV_CG_FUSED(false, true, grid2); else SPMV_CG_FUSED(false, false, grid);
                }
#undef SPMV_CG_FUSED
                return SPMV_OK;
            }
            apply_extra ex;
            ex.overwrite = true;
            ex.dot_w     = p;
            ex.dot_out   = s->pq[kk & 3];
            SPMV_TRY(mat_apply_ex(ctx, A, p, q, ex));  // q = A p, pq_k = p . q
            if (z)
            {
                hipLaunchKernelGGL(gcg_update_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, p, q, x, r, s);
                SPMV_TRY(symgs_sweep(ctx, A, r, z, true));
                SPMV_TRY(vec_dot_accumulate(ctx, r, z, n, s->rz[(kk + 1) & 3]));
                hipLaunchKernelGGL(gcg_direction_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, z, p, s);
            }
            else if (dinv)
            {
                hipLaunchKernelGGL(pcg_update_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, p, q, x, r, s, dinv);
                hipLaunchKernelGGL(pcg_direction_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, r, p, s, dinv);
            }
            else if (wide)
            {
                hipLaunchKernelGGL(cg_update2_kernel, dim3(grid2), dim3(kBlock), 0, st, n, kk, p, q, x, r, s);
                hipLaunchKernelGGL(cg_direction2_kernel, dim3(grid2), dim3(kBlock), 0, st, n, kk, r, p, s);
            }
            else
            {
                hipLaunchKernelGGL(cg_update_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, p, q, x, r, s);
                hipLaunchKernelGGL(cg_direction_kernel, dim3(grid), dim3(kBlock), 0, st, n, kk, r, p, s);
            }
            return SPMV_OK;
        };
        // Between two looks at the residual nothing depends on the host, and the scalar slots repeat with period 4, so
        // four iterations can be captured once into a hipGraph and replayed with one launch.  MEASURED SLOWER than the
        // plain stream of launches on ROCm 7.2 (profiles/r01_tune_cg_graph.txt: 61 vs 23 us per iteration at n = 10^4,
        // 214 vs 168 us at n = 4M), so it is off unless SPMV_CG_GRAPH=1 asks for it.
        hipGraph_t     graph = nullptr;
        hipGraphExec_t exec  = nullptr;
        const char*    want_graph = getenv("SPMV_CG_GRAPH");
        if (want_graph && want_graph[0] == '1' && every >= 4 && max_iter >= 4 &&
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess)
        {
            int crc = SPMV_OK;
            for (int kk = 0; kk < 4 && crc == SPMV_OK; ++kk) crc = iteration(kk);
            const hipError_t e_end = hipStreamEndCapture(st, &graph);
            if (crc != SPMV_OK || e_end != hipSuccess || graph == nullptr ||
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
            {
                exec = nullptr;
                (void)hipGetLastError();
            }
        }
        while (k < max_iter)
        {
            const int until = std::min(max_iter, (k / every + 1) * every);  // the next look at the residual
            if (exec && (k & 3) == 0 && k + 4 <= until)
            {
                if (hipGraphLaunch(exec, st) != hipSuccess)
                {
                    rc = SPMV_ERR_HIP;
                    set_error("spmv_cg: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
                    break;
                }
                k += 4;
            }
            else
            {
                if ((rc = iteration(k)) != SPMV_OK) break;
                ++k;
            }
            if (k % every == 0 || k == max_iter)
            {
                if ((rc = fetch(k & 3)) != SPMV_OK) break;
                rr = host_sum(h.rr[k & 3]);
                // the status word is only set with r != 0 (an exactly solved system ends the queued iterations quietly, see
                // the update kernels); it is looked at first because a breakdown leaves the next r.r at its cleared 0
                if (h.status == 2.0 || !std::isfinite(rr))
                {
                    set_error("spmv_cg: the residual is not finite at or before iteration %d (non-finite numbers in b, x0 or the matrix, or overflow)", k);
                    rc = SPMV_ERR_INVALID;
                    break;
                }
                if (h.status != 0.0)
                {
                    set_error("spmv_cg: p.Ap <= 0 (or r.M^-1 r <= 0) at or before iteration %d: the matrix is not positive definite", k);
                    rc = SPMV_ERR_INVALID;
                    break;
                }
                if (rr <= limit) break;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc == SPMV_OK && hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
        *iters     = k;
        *rel_resid = sqrt(rr / bb);
    } while (0);
    (void)hipStreamSynchronize(st);
    release();
    return rc;
}
}  // namespace spmv
