// validate.hip — optional structural check of a handle's index arrays (spmv_mat_validate).
//
// The reference never checks indices (a bad column index reads out of bounds, src/mat_vec.cpp:62); the products here
// do not either, for speed.  This is the explicit check a caller runs once on untrusted input: every index inside
// its range, offsets non-decreasing and consistent with the entry count.  One pass over the index arrays.
#include "common.hpp"

namespace spmv
{
namespace
{
// flags: bit 0 = index out of range, bit 1 = offsets decrease, bit 2 = offsets do not start at 0 / end at nnz
__global__ __launch_bounds__(kBlock) void check_range_kernel(const int32_t* __restrict__ idx, int64_t n, int32_t limit,
                                                             int32_t* __restrict__ flags)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        bad |= (uint32_t)idx[i] >= (uint32_t)limit;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

__global__ __launch_bounds__(kBlock) void check_offsets_kernel(const int32_t* __restrict__ ptr, int64_t n /* entries of ptr - 1 */,
                                                               int64_t nnz, int32_t* __restrict__ flags)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        bad |= ptr[i] > ptr[i + 1];
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 2);
    if (blockIdx.x == 0 && threadIdx.x == 0 && (ptr[0] != 0 || ptr[n] != nnz)) atomicOr(flags, 4);
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock))); }
}  // namespace

int mat_validate(const spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    // a handle that released its index arrays (panel_keep_csr = 0) has nothing left to check: refuse on the host, the
    // kernels below would dereference a null pointer on the GPU
    const int64_t entries = m->format == SPMV_FMT_ELL ? (int64_t)m->nrow * m->k : m->nnz;
    if (m->format != SPMV_FMT_DIA)
    {
        const bool need_a = m->format != SPMV_FMT_ELL && (m->format != SPMV_FMT_COO || entries > 0);
        SPMV_REQUIRE(!(need_a && !m->a) && !(entries > 0 && !m->b),
                     "spmv_mat_validate: this handle gave up its index arrays (panel_keep_csr = 0): validate before releasing them");
    }
    SPMV_TRY(ensure_scratch(ctx, 64));
    int32_t*    flags = (int32_t*)ctx->scratch;
    hipStream_t s     = ctx->stream;
    SPMV_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t), s));
    const int64_t nnz = m->nnz;
    switch (m->format)
    {
        case SPMV_FMT_CSR:
            hipLaunchKernelGGL(check_offsets_kernel, dim3(grid_for(m->nrow)), dim3(kBlock), 0, s, m->a, (int64_t)m->nrow, nnz, flags);
            if (nnz) hipLaunchKernelGGL(check_range_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, m->b, nnz, m->ncol, flags);
            break;
        case SPMV_FMT_CSC:
            hipLaunchKernelGGL(check_offsets_kernel, dim3(grid_for(m->ncol)), dim3(kBlock), 0, s, m->a, (int64_t)m->ncol, nnz, flags);
            if (nnz) hipLaunchKernelGGL(check_range_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, m->b, nnz, m->nrow, flags);
            break;
        case SPMV_FMT_COO:
            if (nnz)
            {
                hipLaunchKernelGGL(check_range_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, m->a, nnz, m->nrow, flags);
                hipLaunchKernelGGL(check_range_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, m->b, nnz, m->ncol, flags);
            }
            break;
        case SPMV_FMT_ELL:
        {
            const int64_t total = (int64_t)m->nrow * m->k;
            if (total) hipLaunchKernelGGL(check_range_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, m->b, total, m->ncol, flags);
            break;
        }
        case SPMV_FMT_DIA: break;  // the product bounds every column by min(nrow, ncol) itself (kernels_misc.hip)
        default: SPMV_FAIL(SPMV_ERR_INVALID, "unknown format %d", m->format);
    }
    SPMV_HIP(hipGetLastError());
    int32_t h = 0;
    SPMV_HIP(hipMemcpyAsync(&h, flags, sizeof(h), hipMemcpyDeviceToHost, s));
    SPMV_HIP(hipStreamSynchronize(s));
    if (h)
        SPMV_FAIL(SPMV_ERR_INVALID, "matrix structure is invalid:%s%s%s", (h & 1) ? " index out of range;" : "",
                  (h & 2) ? " offsets decrease;" : "", (h & 4) ? " offsets do not run from 0 to the entry count;" : "");
    return SPMV_OK;
}
}  // namespace spmv
