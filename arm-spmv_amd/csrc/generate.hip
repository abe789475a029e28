// generate.hip — synthetic inputs generated directly in device memory.
//
// The reference ships no matrices; its harness reads Matrix Market files (src/data_io.cpp:45-105) and
// fills x with unseeded rand() (src/vector.cpp:65-69).  The benchmark configurations are far too large
// to go through text files, so the inputs are drawn on the device from a counter-based generator:
//     draw(seed, stream, index) = splitmix64( key(seed, stream) + index )
// Every element depends only on its global index, so any shard of any size can be regenerated
// anywhere (GPU kernel here, numpy twin in the Python package, used by the parity tests and by the
// CPU-baseline sample) and is identical bit for bit.
#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
// one thread per entry; entry e of the shard is (row e / k, slot e % k); global index = grow*k + slot
__global__ __launch_bounds__(kBlock) void gen_csr_uniform_kernel(int64_t row_begin, int32_t nrow, int32_t ncol,
                                                                 int32_t k, int32_t band, uint64_t key_col,
                                                                 uint64_t key_val, int32_t* __restrict__ row_ptr,
                                                                 int32_t* __restrict__ col, double* __restrict__ val)
{
    const int64_t total = (int64_t)nrow * k;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock)
    {
        const int64_t  grow = row_begin + e / k;
        const uint64_t gidx = (uint64_t)grow * (uint64_t)k + (uint64_t)(e % k);
        const uint64_t rc   = splitmix64(key_col + gidx);
        int32_t        c;
        if (band <= 0)
            c = (int32_t)u64_to_range(rc, (uint32_t)ncol);
        else
        {
            // uniform in a window of `band` columns centred on the diagonal, wrapping around
            int64_t cc = (grow % ncol) + (int64_t)u64_to_range(rc, (uint32_t)band) - band / 2;
            cc %= ncol;
            if (cc < 0) cc += ncol;
            c = (int32_t)cc;
        }
        col[e] = c;
        val[e] = u64_to_sym(splitmix64(key_val + gidx));
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= nrow; i += (int64_t)gridDim.x * kBlock)
        row_ptr[i] = (int32_t)(i * k);
}

// column-major ELL: (row i, slot d) at i + d*nrow; col = (i + d - k/2) mod ncol
__global__ __launch_bounds__(kBlock) void gen_ell_banded_kernel(int32_t nrow, int32_t ncol, int32_t k, uint64_t key_val,
                                                                int32_t* __restrict__ col, double* __restrict__ val)
{
    const int64_t total = (int64_t)nrow * k;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock)
    {
        const int32_t i = (int32_t)(e % nrow);
        const int32_t d = (int32_t)(e / nrow);
        int64_t       c = ((int64_t)i + d - k / 2) % ncol;
        if (c < 0) c += ncol;
        col[e] = (int32_t)c;
        val[e] = u64_to_sym(splitmix64(key_val + (uint64_t)i * (uint64_t)k + (uint64_t)d));
    }
}

// sorted_by_length: the same distribution taken at its quantiles, u_i = (i + 1) / nrow, instead of drawn: the rows come sorted
// by length, the longest first - every heavy row at one end of the matrix, the positional skew an equal-rows partition
// handles worst ("an nnz-balanced split as an option for C4-like skew")
__global__ __launch_bounds__(kBlock) void gen_row_len_kernel(int32_t nrow, int32_t max_len, uint64_t key_len, int sorted_by_length,
                                                             int32_t* __restrict__ len)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= nrow; i += (int64_t)gridDim.x * kBlock)
    {
        int32_t l = 0;
        if (i < nrow)
        {
            const double u = sorted_by_length ? (double)(i + 1) / (double)nrow : 1.0 - u64_to_unit(splitmix64(key_len + (uint64_t)i));  // (0,1]
            const double q = floor(8.0 / u);
            l              = q >= (double)max_len ? max_len : (int32_t)q;
        }
        len[i] = l;  // len[nrow] = 0 so that the scan leaves nnz in row_ptr[nrow]
    }
}

// LPR lanes per row fill the row's entries; entry s of row i has generator index i*max_len + s
template <int LPR>
__global__ __launch_bounds__(kBlock) void gen_coo_fill_kernel(int32_t nrow, int32_t ncol, int32_t max_len,
                                                              uint64_t key_col, uint64_t key_val,
                                                              const int32_t* __restrict__ row_ptr,
                                                              int32_t* __restrict__ row, int32_t* __restrict__ col,
                                                              double* __restrict__ val)
{
    const int r = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR;
    if (r >= nrow) return;
    const int begin = row_ptr[r], len = row_ptr[r + 1] - begin;
    for (int s = threadIdx.x % LPR; s < len; s += LPR)
    {
        const uint64_t gidx = (uint64_t)r * (uint64_t)max_len + (uint64_t)s;
        row[begin + s]      = r;
        col[begin + s]      = (int32_t)u64_to_range(splitmix64(key_col + gidx), (uint32_t)ncol);
        val[begin + s]      = u64_to_sym(splitmix64(key_val + gidx));
    }
}

__global__ __launch_bounds__(kBlock) void gen_vec_kernel(double* __restrict__ d, int64_t n, int64_t index_offset,
                                                         uint64_t key)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        d[i] = u64_to_unit(splitmix64(key + (uint64_t)(index_offset + i)));
}

inline unsigned stream_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock))); }
}  // namespace

int gen_csr_uniform(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol, int32_t k, int32_t band,
                    uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(row_begin >= 0 && row_end >= row_begin && row_end - row_begin <= INT32_MAX, "bad row range");
    SPMV_REQUIRE(ncol > 0 && k >= 0 && band >= 0 && band <= ncol, "bad ncol/k/band");
    const int32_t nrow = (int32_t)(row_end - row_begin);
    const int64_t nnz  = (int64_t)nrow * k;
    SPMV_REQUIRE(nnz <= INT32_MAX, "shard of %d rows x %d entries does not fit int32 offsets", nrow, k);
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSR, nrow, ncol, nnz, 0, (size_t)nrow + 1, (size_t)nnz, (size_t)nnz, &m));
    hipLaunchKernelGGL(gen_csr_uniform_kernel, dim3(stream_grid(std::max<int64_t>(nnz, nrow + 1))), dim3(kBlock), 0,
                       ctx->stream, row_begin, nrow, ncol, k, band, stream_key(seed, kStreamCol),
                       stream_key(seed, kStreamVal), const_cast<int32_t*>(m->a), const_cast<int32_t*>(m->b),
                       const_cast<double*>(m->v));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        mat_free(m);
        SPMV_FAIL(SPMV_ERR_HIP, "gen_csr_uniform: %s", hipGetErrorString(e));
    }
    m->row_begin = row_begin;
    int rc       = csr_analyse(m);
    if (rc != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    *out = m;
    return SPMV_OK;
}

int gen_ell_banded(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(nrow >= 0 && ncol > 0 && k >= 0, "bad nrow/ncol/k");
    const size_t total = (size_t)nrow * (size_t)k;
    spmv_mat*    m     = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_ELL, nrow, ncol, (int64_t)total, k, 0, total, total, &m));
    hipLaunchKernelGGL(gen_ell_banded_kernel, dim3(stream_grid((int64_t)total)), dim3(kBlock), 0, ctx->stream, nrow,
                       ncol, k, stream_key(seed, kStreamVal), const_cast<int32_t*>(m->b), const_cast<double*>(m->v));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
    {
        mat_free(m);
        SPMV_FAIL(SPMV_ERR_HIP, "gen_ell_banded: %s", hipGetErrorString(e));
    }
    m->max_row_nnz = k;
    *out           = m;
    return SPMV_OK;
}

int gen_coo_powerlaw(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed, bool sorted_by_length, spmv_mat** out)
{
    SPMV_REQUIRE(nrow >= 0 && ncol > 0 && max_len > 0, "bad nrow/ncol/max_len");
    int32_t* len     = nullptr;
    int32_t* row_ptr = nullptr;
    SPMV_HIP(hipMalloc(&len, sizeof(int32_t) * ((size_t)nrow + 1)));
    if (hipMalloc(&row_ptr, sizeof(int32_t) * ((size_t)nrow + 1)) != hipSuccess)
    {
        (void)hipFree(len);
        SPMV_FAIL(SPMV_ERR_ALLOC, "gen_coo_powerlaw: out of device memory");
    }
    spmv_mat* m  = nullptr;
    int       rc = SPMV_OK;
    do
    {
        hipLaunchKernelGGL(gen_row_len_kernel, dim3(stream_grid(nrow + 1)), dim3(kBlock), 0, ctx->stream, nrow, max_len,
                           stream_key(seed, kStreamLen), sorted_by_length ? 1 : 0, len);
        if ((rc = exclusive_scan_i32(ctx, len, row_ptr, (int64_t)nrow + 1)) != SPMV_OK) break;
        int32_t nnz = 0;
        if (hipMemcpyAsync(&nnz, row_ptr + nrow, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (nnz < 0)
        {
            rc = SPMV_ERR_INVALID;  // int32 overflow in the scan
            break;
        }
        if ((rc = mat_alloc(ctx, SPMV_FMT_COO, nrow, ncol, nnz, 0, (size_t)nnz, (size_t)nnz, (size_t)nnz, &m)) != SPMV_OK)
            break;
        constexpr int LPR = 8;
        if (!launch_fits(nrow, LPR))
        {
            set_error("spmv_gen_coo_powerlaw: %d rows are more than one launch of the fill step holds", nrow);
            rc = SPMV_ERR_UNSUPPORTED;
            break;
        }
        if (nrow > 0)
            hipLaunchKernelGGL(gen_coo_fill_kernel<LPR>, dim3((unsigned)ceil_div(nrow, kBlock / LPR)), dim3(kBlock), 0,
                               ctx->stream, nrow, ncol, max_len, stream_key(seed, kStreamCol),
                               stream_key(seed, kStreamVal), row_ptr, const_cast<int32_t*>(m->a),
                               const_cast<int32_t*>(m->b), const_cast<double*>(m->v));
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    (void)hipFree(len);
    (void)hipFree(row_ptr);
    if (rc != SPMV_OK)
    {
        if (m) mat_free(m);
        SPMV_FAIL(rc, "gen_coo_powerlaw failed");
    }
    if ((rc = coo_analyse(m)) != SPMV_OK)
    {
        mat_free(m);
        return rc;
    }
    *out = m;
    return SPMV_OK;
}

// DIA twin of gen_ell_banded: k diagonals with offsets d - k/2, values (i, d) from the same draw as the ELL
// generator (index i*k + d), row-major as the reference stores them.  (No wrap-around: entries whose column
// falls outside [0, nrow) exist in the array but are skipped by the product, src/mat_vec.cpp:140.)
__global__ __launch_bounds__(kBlock) void gen_dia_banded_kernel(int32_t nrow, int32_t k, uint64_t key_val,
                                                                int32_t* __restrict__ offsets, double* __restrict__ val)
{
    const int64_t total = (int64_t)nrow * k;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock)
        val[e] = u64_to_sym(splitmix64(key_val + (uint64_t)e));  // e = i*k + d
    for (int d = blockIdx.x * kBlock + threadIdx.x; d < k; d += gridDim.x * kBlock) offsets[d] = d - k / 2;
}

int gen_dia_banded(spmv_ctx* ctx, int32_t nrow, int32_t k, uint64_t seed, spmv_mat** out)
{
    SPMV_REQUIRE(nrow >= 0 && k >= 0, "bad nrow/k");
    const size_t total = (size_t)nrow * (size_t)k;
    spmv_mat*    m     = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_DIA, nrow, nrow, (int64_t)total, k, (size_t)k, 0, total, &m));
    hipLaunchKernelGGL(gen_dia_banded_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(kMaxGrid, ceil_div((int64_t)total, kBlock)))),
                       dim3(kBlock), 0, ctx->stream, nrow, k, stream_key(seed, kStreamVal), const_cast<int32_t*>(m->a),
                       const_cast<double*>(m->v));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
    {
        mat_free(m);
        SPMV_FAIL(SPMV_ERR_HIP, "gen_dia_banded: %s", hipGetErrorString(e));
    }
    if (k > 0)
    {
        m->dia_off_known = true;  // offsets d - k / 2, d = 0 .. k-1
        m->dia_off_min   = -(k / 2);
        m->dia_off_max   = k - 1 - k / 2;
    }
    *out = m;
    return SPMV_OK;
}

int gen_vec_uniform(spmv_ctx* ctx, double* d, int64_t n, int64_t index_offset, uint64_t seed)
{
    if (n == 0) return SPMV_OK;
    hipLaunchKernelGGL(gen_vec_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, d, n, index_offset,
                       stream_key(seed, kStreamVec));
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
