// convert_sort.hip — stable ordering of COO entries by row for inputs with very long unsorted rows.
//
// CSRMatrix(COOMatrix) keeps the COO order inside every row (src/matrix.cpp:140-144, a stable counting sort).  The
// placement of convert.hip (claim a slot, then rank the entry ids inside the row) is O(len^2) per row: fine for the
// rows of a sparse matrix, an effective hang for a hub row of 10^5..10^6 unsorted entries.  For those inputs the entry
// ids are sorted by row with rocPRIM's LSD radix sort instead — stable, so ids stay ascending inside a row, which IS
// the COO order — and the entries are then copied through the permutation.  Integer work off the hot path; kept in a
// translation unit of its own because the rocPRIM headers are slow to compile.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace spmv
{
namespace
{
__global__ __launch_bounds__(kBlock) void iota_kernel(int32_t* __restrict__ p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = (int32_t)i;
}
__global__ __launch_bounds__(kBlock) void permute_entries_kernel(int64_t n, const int32_t* __restrict__ perm,
                                                                 const int32_t* __restrict__ col, const double* __restrict__ val,
                                                                 int32_t* __restrict__ out_col, double* __restrict__ out_val)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    {
        const int32_t e = perm[i];
        out_col[i]      = col[e];
        out_val[i]      = val[e];
    }
}
}  // namespace

// out_ids[k] = the k-th id of 0 .. n-1 in ascending key order, ids of equal keys ascending (stable); keys < 2^bits
int sort_ids_by_key(spmv_ctx* ctx, const int32_t* keys, int64_t n, int bits, int32_t* out_ids)
{
    if (n == 0) return SPMV_OK;
    hipStream_t s = ctx->stream;
    int32_t *   ids = nullptr, *keys_sorted = nullptr;
    void*       temp = nullptr;
    size_t      temp_bytes = 0;
    int         rc = SPMV_OK;
    do
    {
        if (hipMalloc(&ids, sizeof(int32_t) * (size_t)n) != hipSuccess || hipMalloc(&keys_sorted, sizeof(int32_t) * (size_t)n) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock))), dim3(kBlock), 0, s, ids, n);
        if (rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_sorted, ids, out_ids, (size_t)n, 0u, (unsigned)bits, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (hipMalloc(&temp, std::max<size_t>(temp_bytes, 16)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, ids, out_ids, (size_t)n, 0u, (unsigned)bits, s) != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    (void)hipStreamSynchronize(s);
    if (ids) (void)hipFree(ids);
    if (keys_sorted) (void)hipFree(keys_sorted);
    if (temp) (void)hipFree(temp);
    if (rc != SPMV_OK) SPMV_FAIL(rc, "ordering %lld ids by key failed (%s)", (long long)n, hipGetErrorString(hipGetLastError()));
    return SPMV_OK;
}

// out_col/out_val[k] = col/val of the k-th entry in (row, entry id) order
int coo_place_by_stable_sort(spmv_ctx* ctx, int64_t nnz, int32_t nrow, const int32_t* row, const int32_t* col, const double* val,
                             int32_t* out_col, double* out_val)
{
    if (nnz == 0) return SPMV_OK;
    hipStream_t s = ctx->stream;
    int32_t *   ids = nullptr, *ids_sorted = nullptr, *rows_sorted = nullptr;
    void*       temp = nullptr;
    size_t      temp_bytes = 0;
    int         rc = SPMV_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(kMaxGrid, ceil_div(nnz, kBlock));
    int bits = 1;
    while (bits < 31 && (1LL << bits) < (long long)nrow) ++bits;
    do
    {
        if (hipMalloc(&ids, sizeof(int32_t) * (size_t)nnz) != hipSuccess || hipMalloc(&ids_sorted, sizeof(int32_t) * (size_t)nnz) != hipSuccess ||
            hipMalloc(&rows_sorted, sizeof(int32_t) * (size_t)nnz) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(kBlock), 0, s, ids, nnz);
        if (rocprim::radix_sort_pairs(nullptr, temp_bytes, row, rows_sorted, ids, ids_sorted, (size_t)nnz, 0u, (unsigned)bits, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (hipMalloc(&temp, std::max<size_t>(temp_bytes, 16)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (rocprim::radix_sort_pairs(temp, temp_bytes, row, rows_sorted, ids, ids_sorted, (size_t)nnz, 0u, (unsigned)bits, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(permute_entries_kernel, dim3(grid), dim3(kBlock), 0, s, nnz, ids_sorted, col, val, out_col, out_val);
        if (hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    (void)hipStreamSynchronize(s);
    if (ids) (void)hipFree(ids);
    if (ids_sorted) (void)hipFree(ids_sorted);
    if (rows_sorted) (void)hipFree(rows_sorted);
    if (temp) (void)hipFree(temp);
    if (rc != SPMV_OK) SPMV_FAIL(rc, "ordering %lld COO entries by row failed (%s)", (long long)nnz, hipGetErrorString(hipGetLastError()));
    return SPMV_OK;
}
}  // namespace spmv
