// convert.hip — format conversion on the device.
//
// Replaces the converting constructors of the reference:
//   CSRMatrix(const COOMatrix&)  src/matrix.cpp:115-154  histogram, prefix sum, backward stable scatter
//   ELLMatrix(const COOMatrix&)  src/matrix.cpp:450-500  K = longest row, zero-padded column-major fill
// Both keep the COO order of the entries inside a row, so the arrays produced here are identical to the
// reference's (tests compare them element for element), not just an equivalent matrix.
//
// The packed `diagonal` array the reference also fills (src/matrix.cpp:146-153, "for SymGS") is not
// consumed by any product; the C++ compat shim builds it on the host when asked.
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
// ---- exclusive prefix sum of int32 (tile = 256 threads x 8 items) ------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile  = kBlock * kScanItems;

__global__ __launch_bounds__(kBlock) void scan_tile_kernel(const int32_t* in, int32_t* out, int64_t n,
                                                           int32_t* tile_sum)  // in may alias out: no restrict
{
    __shared__ int32_t s_wave[kBlock / kWave];
    const int64_t      base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int32_t            item[kScanItems];
    int32_t            local = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
    {
        item[i] = (base + i < n) ? in[base + i] : 0;
        local += item[i];
    }
    // inclusive scan of the per-thread totals across the wavefront
    const int lane = lane_id();
    int32_t   incl = local;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1)
    {
        const int32_t up = bpermute(incl, max(lane - d, 0));
        if (lane >= d) incl += up;
    }
    if (lane == kWave - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t wave_off = 0, total = 0;
    for (int w = 0; w < kBlock / kWave; ++w)
    {
        if (w < (int)(threadIdx.x >> 6)) wave_off += s_wave[w];
        total += s_wave[w];
    }
    int32_t run = wave_off + incl - local;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
    {
        if (base + i < n) out[base + i] = run;
        run += item[i];
    }
    if (threadIdx.x == 0 && tile_sum) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(int32_t* __restrict__ out, int64_t n,
                                                          const int32_t* __restrict__ tile_off)
{
    const int32_t off  = tile_off[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    for (int i = threadIdx.x; i < kScanTile; i += kBlock)
        if (base + i < n) out[base + i] += off;
}

// ---- COO -> CSR ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void row_histogram_kernel(int64_t nnz, const int32_t* __restrict__ row,
                                                               int32_t* __restrict__ count)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock)
        atomicAdd(count + row[e], 1);
}

// unsorted input, step 1: claim any free slot of the row, remember which entry sits there
__global__ __launch_bounds__(kBlock) void claim_slot_kernel(int64_t nnz, const int32_t* __restrict__ row,
                                                            const int32_t* __restrict__ row_ptr,
                                                            int32_t* __restrict__ cursor, int32_t* __restrict__ perm)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock)
    {
        const int r               = row[e];
        const int slot            = atomicAdd(cursor + r, 1);
        perm[row_ptr[r] + slot]   = (int32_t)e;
    }
}

// unsorted input, step 2: LPR lanes per row put the row's entries in ascending entry order (= COO order).
// Entry ids are distinct, so the rank of an id among the row's ids is its final slot.
template <int LPR>
__global__ __launch_bounds__(kBlock) void order_rows_kernel(int nrow, const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ col,
                                                            const double* __restrict__ val,
                                                            int32_t* __restrict__ out_col, double* __restrict__ out_val)
{
    const int r = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR;
    if (r >= nrow) return;
    const int begin = row_ptr[r], end = row_ptr[r + 1];
    for (int j = begin + threadIdx.x % LPR; j < end; j += LPR)
    {
        const int e    = perm[j];
        int       rank = 0;
        for (int q = begin; q < end; ++q) rank += perm[q] < e;
        out_col[begin + rank] = col[e];
        out_val[begin + rank] = val[e];
    }
}

// ---- CSR -> ELL (column-major, zero padded) --------------------------------------------------------------------
// PAD_OWN: every slot of the row is written; the slots beyond its entries get value 0.0 and, as their column, the COMPLEMENT of
// the row's own last column (~c = -1 - c < 0; ~0 for an empty row) instead of the reference's column 0 - the internal ELL copy
// of a CSR handle (kernels_ell.hip: csr_ell_copy_build).  A negative column says "padding: no part of the sum" to the copy's
// kernels (their MASKED instances), and decodes to a column the row reads anyway for the gather they issue regardless
template <int LPR, bool PAD_OWN>
__global__ __launch_bounds__(kBlock) void csr_to_ell_kernel(int nrow, int k, const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col,
                                                            const double* __restrict__ val,
                                                            int32_t* __restrict__ ell_col, double* __restrict__ ell_val)
{
    const int r = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR;
    if (r >= nrow) return;
    const int begin = row_ptr[r], len = row_ptr[r + 1] - begin;
    const int pad   = PAD_OWN ? ~(len > 0 ? col[begin + len - 1] : 0) : 0;
    for (int s = threadIdx.x % LPR; s < (PAD_OWN ? k : len); s += LPR)
    {
        const size_t at = (size_t)r + (size_t)s * (size_t)nrow;
        ell_col[at]     = s < len ? col[begin + s] : pad;
        ell_val[at]     = s < len ? val[begin + s] : 0.0;
    }
}
}  // namespace

int exclusive_scan_i32(spmv_ctx* ctx, const int32_t* in, int32_t* out, int64_t n)
{
    if (n <= 0) return SPMV_OK;
    const int64_t tiles = ceil_div(n, kScanTile);
    if (tiles == 1)
    {
        hipLaunchKernelGGL(scan_tile_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, in, out, n, (int32_t*)nullptr);
        SPMV_HIP(hipGetLastError());
        return SPMV_OK;
    }
    int32_t* tile_sum = nullptr;
    SPMV_HIP(hipMalloc(&tile_sum, sizeof(int32_t) * (size_t)tiles));
    hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, in, out, n, tile_sum);
    int rc = exclusive_scan_i32(ctx, tile_sum, tile_sum, tiles);  // in place: each tile reads before it writes
    if (rc == SPMV_OK)
    {
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, out, n, tile_sum);
        if (hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tile_sum);
    if (rc != SPMV_OK) SPMV_FAIL(rc, "exclusive_scan_i32 failed");
    return SPMV_OK;
}

int coo_to_csr(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out, int32_t force_kernel)
{
    SPMV_REQUIRE(coo->format == SPMV_FMT_COO, "spmv_coo_to_csr: input is not COO");
    SPMV_REQUIRE(coo->nnz <= INT32_MAX, "spmv_coo_to_csr: %lld entries do not fit int32 row_ptr", (long long)coo->nnz);
    const int     nrow = coo->nrow;
    const int64_t nnz  = coo->nnz;
    spmv_mat*     csr  = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSR, nrow, coo->ncol, nnz, 0, (size_t)nrow + 1, (size_t)nnz, (size_t)nnz, &csr));
    int32_t* row_ptr = const_cast<int32_t*>(csr->a);
    int32_t* out_col = const_cast<int32_t*>(csr->b);
    double*  out_val = const_cast<double*>(csr->v);
    int32_t* count   = nullptr;
    int32_t* perm    = nullptr;
    int      rc      = SPMV_OK;
    hipStream_t s    = ctx->stream;
    do
    {
        if (hipMalloc(&count, sizeof(int32_t) * ((size_t)nrow + 1)) != hipSuccess) { rc = SPMV_ERR_ALLOC; break; }
        (void)hipMemsetAsync(count, 0, sizeof(int32_t) * ((size_t)nrow + 1), s);
        if (nnz > 0)
            hipLaunchKernelGGL(row_histogram_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(nnz, kBlock))),
                               dim3(kBlock), 0, s, nnz, coo->a, count);
        if ((rc = exclusive_scan_i32(ctx, count, row_ptr, (int64_t)nrow + 1)) != SPMV_OK) break;
        if (nnz == 0) break;
        if (coo->sorted_rows)
        {
            // row-sorted COO is already in CSR order
            (void)hipMemcpyAsync(out_col, coo->b, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(out_val, coo->v, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, s);
        }
        else
        {
            // The slot-claim + rank placement below is O(len^2) per row; rows beyond kRankRowLimit entries (a hub row of an
            // unsorted file) go through a stable sort of the entry ids by row instead (convert_sort.hip).
            constexpr int32_t kRankRowLimit = 2048;
            int32_t           longest       = 0;
            if ((rc = reduce_max_i32(ctx, count, nrow, &longest)) != SPMV_OK) break;
            if (longest > kRankRowLimit)
            {
                rc = coo_place_by_stable_sort(ctx, nnz, nrow, coo->a, coo->b, coo->v, out_col, out_val);
                break;
            }
            if (hipMalloc(&perm, sizeof(int32_t) * (size_t)nnz) != hipSuccess) { rc = SPMV_ERR_ALLOC; break; }
            (void)hipMemsetAsync(count, 0, sizeof(int32_t) * ((size_t)nrow + 1), s);  // reuse as per-row cursor
            hipLaunchKernelGGL(claim_slot_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(nnz, kBlock))),
                               dim3(kBlock), 0, s, nnz, coo->a, row_ptr, count, perm);
            constexpr int LPR = 8;
            if (!launch_fits(nrow, LPR))
            {
                set_error("spmv_coo_to_csr: %d rows are more than one launch of the ordering step holds", nrow);
                rc = SPMV_ERR_UNSUPPORTED;
                break;
            }
            hipLaunchKernelGGL(order_rows_kernel<LPR>, dim3((unsigned)ceil_div(nrow, kBlock / LPR)), dim3(kBlock), 0, s,
                               nrow, row_ptr, perm, coo->b, coo->v, out_col, out_val);
        }
        if (hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    (void)hipStreamSynchronize(s);
    if (count) (void)hipFree(count);
    if (perm) (void)hipFree(perm);
    if (rc != SPMV_OK)
    {
        mat_free(csr);
        SPMV_FAIL(rc, "spmv_coo_to_csr failed (%s)", hipGetErrorString(hipGetLastError()));
    }
    csr->row_begin = coo->row_begin;
    if (force_kernel == kCsrAutoNoSegscan)
        csr->sel_no_segscan = true;  // (the COO handle that asks has that scan itself, over its own arrays)
    else if (force_kernel != SPMV_CSR_AUTO)
    {
        csr->kernel_forced = true;
        csr->kernel        = force_kernel;
    }
    csr->pb_trial = coo->pb_trial;  // ("panel_trial" of the source handle holds for what is built from it)
    plan_hand_down(coo, csr, kPlanChildRowgrouped);  // (a source that is being built from a plan: the copy takes its node)
    if ((rc = csr_analyse(csr)) != SPMV_OK)
    {
        mat_free(csr);
        return rc;
    }
    *out = csr;
    return SPMV_OK;
}

int csr_to_ell(spmv_ctx* ctx, const spmv_mat* csr, spmv_mat** out, bool pad_own_column)
{
    SPMV_REQUIRE(csr->format == SPMV_FMT_CSR, "spmv_csr_to_ell: input is not CSR");
    SPMV_REQUIRE(csr->nnz == 0 || (csr->b && csr->v), "spmv_csr_to_ell: the handle gave up its CSR arrays (panel_keep_csr = 0)");
    const int    nrow  = csr->nrow;
    const int    k     = csr->max_row_nnz;
    const size_t total = (size_t)nrow * (size_t)k;
    SPMV_REQUIRE(total < ((size_t)1 << 40), "spmv_csr_to_ell: %d rows x %d slots is unreasonably large", nrow, k);
    spmv_mat* ell = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_ELL, nrow, csr->ncol, csr->nnz, k, 0, total, total, &ell));
    if (total > 0)
    {
        // padding: col 0, val +0.0 (src/matrix.cpp:473-474, value-initialised new[])
        if (!pad_own_column)
        {
            (void)hipMemsetAsync(const_cast<int32_t*>(ell->b), 0, sizeof(int32_t) * total, ctx->stream);
            (void)hipMemsetAsync(const_cast<double*>(ell->v), 0, sizeof(double) * total, ctx->stream);
        }
        constexpr int LPR = 8;
        if (!launch_fits(nrow, LPR))
        {
            mat_free(ell);
            SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "spmv_csr_to_ell: %d rows are more than one launch of the fill step holds", nrow);
        }
        if (pad_own_column)
            hipLaunchKernelGGL((csr_to_ell_kernel<LPR, true>), dim3((unsigned)ceil_div(nrow, kBlock / LPR)), dim3(kBlock), 0, ctx->stream, nrow, k, csr->a,
                               csr->b, csr->v, const_cast<int32_t*>(ell->b), const_cast<double*>(ell->v));
        else
            hipLaunchKernelGGL((csr_to_ell_kernel<LPR, false>), dim3((unsigned)ceil_div(nrow, kBlock / LPR)), dim3(kBlock), 0, ctx->stream, nrow, k, csr->a,
                               csr->b, csr->v, const_cast<int32_t*>(ell->b), const_cast<double*>(ell->v));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
        {
            mat_free(ell);
            SPMV_FAIL(SPMV_ERR_HIP, "spmv_csr_to_ell: %s", hipGetErrorString(e));
        }
    }
    ell->max_row_nnz = k;
    ell->row_begin   = csr->row_begin;
    *out             = ell;
    return SPMV_OK;
}
// ---- CSR split by column range (sharded solver step: local columns first, the rest after the exchange) -----------
namespace
{
// per row: entries with a column inside [c0, c1)
__global__ __launch_bounds__(kBlock) void split_count_kernel(int nrow, const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ col, int c0, int c1,
                                                             int32_t* __restrict__ n_in, int32_t* __restrict__ n_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i > nrow) return;
    int inside = 0, total = 0;
    if (i < nrow)
    {
        const int end = row_ptr[i + 1];
        total         = end - row_ptr[i];
        for (int j = row_ptr[i]; j < end; ++j)
        {
            const int c = col[j];
            inside += (c >= c0 && c < c1);
        }
    }
    n_in[i]  = inside;  // entry nrow = 0: the scans run over nrow + 1 values
    n_out[i] = total - inside;
}

__global__ __launch_bounds__(kBlock) void split_fill_kernel(int nrow, const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col, const double* __restrict__ val,
                                                            int c0, int c1, const int32_t* __restrict__ rp_in,
                                                            int32_t* __restrict__ col_in, double* __restrict__ val_in,
                                                            const int32_t* __restrict__ rp_out, int32_t* __restrict__ col_out,
                                                            double* __restrict__ val_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrow) return;
    int       a = rp_in[i], b = rp_out[i];
    const int end = row_ptr[i + 1];
    for (int j = row_ptr[i]; j < end; ++j)  // the order inside a row is kept in both parts
    {
        const int    c = col[j];
        const double v = val[j];
        if (c >= c0 && c < c1)
        {
            col_in[a] = c - c0;  // rebased: the inside part multiplies the caller's own slice of x
            val_in[a] = v;
            ++a;
        }
        else
        {
            col_out[b] = c;
            val_out[b] = v;
            ++b;
        }
    }
}
}  // namespace

int csr_split_columns(spmv_ctx* ctx, const spmv_mat* csr, int32_t c0, int32_t c1, spmv_mat** out_in, spmv_mat** out_out)
{
    SPMV_REQUIRE(csr->format == SPMV_FMT_CSR && csr->b && csr->v, "spmv_csr_split_columns: input is not a CSR handle with its arrays");
    SPMV_REQUIRE(c0 >= 0 && c0 <= c1 && c1 <= csr->ncol, "spmv_csr_split_columns: column range [%d, %d) outside [0, %d)", c0, c1,
                 csr->ncol);
    const int   nrow = csr->nrow;
    hipStream_t s    = ctx->stream;
    int32_t *   n_in = nullptr, *n_out = nullptr;
    spmv_mat *  A_in = nullptr, *A_out = nullptr;
    int         rc   = SPMV_OK;
    do
    {
        if (hipMalloc(&n_in, sizeof(int32_t) * ((size_t)nrow + 1)) != hipSuccess ||
            hipMalloc(&n_out, sizeof(int32_t) * ((size_t)nrow + 1)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        const unsigned grid = (unsigned)ceil_div((int64_t)nrow + 1, kBlock);
        hipLaunchKernelGGL(split_count_kernel, dim3(grid), dim3(kBlock), 0, s, nrow, csr->a, csr->b, c0, c1, n_in, n_out);
        // totals first (sizes of the two parts), then the offsets
        int32_t last_in = 0, last_out = 0;
        if ((rc = exclusive_scan_i32(ctx, n_in, n_in, (int64_t)nrow + 1)) != SPMV_OK) break;
        if ((rc = exclusive_scan_i32(ctx, n_out, n_out, (int64_t)nrow + 1)) != SPMV_OK) break;
        if (hipMemcpyAsync(&last_in, n_in + nrow, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&last_out, n_out + nrow, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if ((rc = mat_alloc(ctx, SPMV_FMT_CSR, nrow, c1 - c0, last_in, 0, (size_t)nrow + 1, (size_t)last_in, (size_t)last_in, &A_in)) != SPMV_OK) break;
        if ((rc = mat_alloc(ctx, SPMV_FMT_CSR, nrow, csr->ncol, last_out, 0, (size_t)nrow + 1, (size_t)last_out, (size_t)last_out, &A_out)) != SPMV_OK) break;
        (void)hipMemcpyAsync(const_cast<int32_t*>(A_in->a), n_in, sizeof(int32_t) * ((size_t)nrow + 1), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(const_cast<int32_t*>(A_out->a), n_out, sizeof(int32_t) * ((size_t)nrow + 1), hipMemcpyDeviceToDevice, s);
        if (nrow > 0)
            hipLaunchKernelGGL(split_fill_kernel, dim3((unsigned)ceil_div(nrow, kBlock)), dim3(kBlock), 0, s, nrow, csr->a, csr->b,
                               csr->v, c0, c1, A_in->a, const_cast<int32_t*>(A_in->b), const_cast<double*>(A_in->v), A_out->a,
                               const_cast<int32_t*>(A_out->b), const_cast<double*>(A_out->v));
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    if (n_in) (void)hipFree(n_in);
    if (n_out) (void)hipFree(n_out);
    if (rc != SPMV_OK)
    {
        if (A_in) mat_free(A_in);
        if (A_out) mat_free(A_out);
        SPMV_FAIL(rc, "spmv_csr_split_columns failed: %s", hipGetErrorString(hipGetLastError()));
    }
    // `inside` has its columns rebased to c0: its row i (global row row_begin + i) meets its own diagonal at column
    // row_begin + i - c0.  For the block of a shard's own rows (c0 = row_begin) that is a square matrix starting at 0, which
    // is what the Jacobi diagonal and the Gauss-Seidel sweep of the sharded solver work on.
    A_out->row_begin = csr->row_begin;
    A_in->row_begin  = csr->row_begin - c0;
    *out_in  = A_in;
    *out_out = A_out;
    return SPMV_OK;
}

// Rows [r0, r1) of a device-resident CSR handle as a shard of their own on `dst` (any context: the same GPU or a peer) -
// src/mat_vec.cpp:250-265 on the devices: rebased int32 row_ptr, global columns, the entries copied device to device.
// Synchronous; ordered behind the work queued on the source's stream.
int csr_extract_rows(spmv_ctx* dst, const spmv_mat* csr, int64_t r0, int64_t r1, spmv_mat** out)
{
    SPMV_REQUIRE(csr->format == SPMV_FMT_CSR && csr->a && (csr->nnz == 0 || (csr->b && csr->v)),
                 "spmv_csr_extract_rows: input is not a CSR handle with its arrays (panel_keep_csr = 0 released them?)");
    SPMV_REQUIRE(r0 >= 0 && r0 <= r1 && r1 <= csr->nrow, "spmv_csr_extract_rows: rows [%lld, %lld) outside [0, %d)", (long long)r0, (long long)r1, csr->nrow);
    spmv_ctx*            sc   = csr->ctx;
    const int32_t        nrow = (int32_t)(r1 - r0);
    std::vector<int32_t> rp((size_t)nrow + 1);
    SPMV_HIP(hipSetDevice(sc->device));
    SPMV_HIP(hipMemcpyAsync(rp.data(), csr->a + r0, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, sc->stream));
    SPMV_HIP(hipStreamSynchronize(sc->stream));
    const int32_t base = rp[0];
    const int64_t nnz  = (int64_t)rp[(size_t)nrow] - base;
    SPMV_REQUIRE(nnz >= 0, "spmv_csr_extract_rows: the handle's row_ptr decreases between rows %lld and %lld", (long long)r0, (long long)r1);
    for (int32_t& v : rp) v -= base;
    SPMV_HIP(hipSetDevice(dst->device));
    spmv_mat* m = nullptr;
    SPMV_TRY(mat_alloc(dst, SPMV_FMT_CSR, nrow, csr->ncol, nnz, 0, (size_t)nrow + 1, (size_t)nnz, (size_t)nnz, &m));
    hipError_t e = hipMemcpyAsync(const_cast<int32_t*>(m->a), rp.data(), sizeof(int32_t) * rp.size(), hipMemcpyHostToDevice, dst->stream);
    if (e == hipSuccess && nnz > 0)
    {
        if (sc->device == dst->device)
        {
            e = hipMemcpyAsync(const_cast<int32_t*>(m->b), csr->b + base, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, dst->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(const_cast<double*>(m->v), csr->v + base, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, dst->stream);
        }
        else
        {
            e = hipMemcpyPeerAsync(const_cast<int32_t*>(m->b), dst->device, csr->b + base, sc->device, sizeof(int32_t) * (size_t)nnz, dst->stream);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(const_cast<double*>(m->v), dst->device, csr->v + base, sc->device, sizeof(double) * (size_t)nnz, dst->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(dst->stream);  // (rp is a host vector about to go out of scope)
    if (e != hipSuccess)
    {
        mat_free(m);
        SPMV_FAIL(SPMV_ERR_HIP, "spmv_csr_extract_rows: %s", hipGetErrorString(e));
    }
    m->row_begin = csr->row_begin + r0;
    *out         = m;
    return SPMV_OK;
}

// row_ptr64[i] = entries of a COO handle in rows < i (a histogram of the row indices + a scan, on the device; the entries
// may be in any order).  What an entry-balanced row partition of a COO handle is cut from.
int coo_row_offsets(const spmv_mat* coo, int64_t* row_ptr64)
{
    SPMV_REQUIRE(coo->format == SPMV_FMT_COO, "coo_row_offsets: not a COO handle");
    spmv_ctx*   ctx  = coo->ctx;
    const int   nrow = coo->nrow;
    hipStream_t s    = ctx->stream;
    SPMV_REQUIRE(coo->nnz <= INT32_MAX, "coo_row_offsets: %lld entries do not fit the int32 histogram", (long long)coo->nnz);
    int32_t* count = nullptr;
    SPMV_HIP(hipSetDevice(ctx->device));
    if (hipMalloc(&count, sizeof(int32_t) * ((size_t)nrow + 1)) != hipSuccess) SPMV_FAIL(SPMV_ERR_ALLOC, "coo_row_offsets: out of device memory");
    std::vector<int32_t> host((size_t)nrow + 1);
    int                  rc = SPMV_OK;
    (void)hipMemsetAsync(count, 0, sizeof(int32_t) * ((size_t)nrow + 1), s);
    if (coo->nnz > 0)
        hipLaunchKernelGGL(row_histogram_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(coo->nnz, kBlock))), dim3(kBlock), 0, s,
                           coo->nnz, coo->a, count);
    rc = exclusive_scan_i32(ctx, count, count, (int64_t)nrow + 1);
    if (rc == SPMV_OK && (hipMemcpyAsync(host.data(), count, sizeof(int32_t) * host.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
                          hipStreamSynchronize(s) != hipSuccess))
        rc = SPMV_ERR_HIP;
    (void)hipFree(count);
    if (rc != SPMV_OK) SPMV_FAIL(rc, "coo_row_offsets failed: %s", hipGetErrorString(hipGetLastError()));
    for (size_t i = 0; i < host.size(); ++i) row_ptr64[i] = host[i];
    return SPMV_OK;
}
}  // namespace spmv
