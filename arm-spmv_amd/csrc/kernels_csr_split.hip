// kernels_csr_split.hip - a CSR handle whose few LONG rows go their own way (SPMV_CSR_SPLIT).
//
// tools/sweep_structures.py "odd" (profiles/r05_sweep_structures_odd_shapes_*.txt): 1M rows of 32 entries and ONE row with an
// entry in every column, 33M entries in all.  Row-wise kernels leave that row to one lane group or (panel layout) to the one
// workgroup that owns its row group: 1.26 ms, of which the other 32M entries need ~0.13.  The segmented scan (SPMV_CSR_SEGSCAN)
// spreads the long row but walks ALL entries in row order, gathering x line by line from an 8 MB vector: 0.47 ms.  Here:
//   * rows of `split_threshold` entries and more stay in the handle's own arrays and are cut into chunks of 4096 entries; one
//     256-lane workgroup per chunk adds its products up (the columns of a long row are sorted and close together: the gathers
//     of x are nearly coalesced) and ends in ONE atomic add on the row's y;
//   * every other row goes into a copy of the CSR arrays without the long rows (their lengths are 0 there) - an ordinary
//     CSR handle that picks ITS kernel as any other does (the panel layout, which then gives back the copy's col_ind / values).
// A product is the copy's product followed by the long-row kernel on the same stream.  Rows that cross chunks are joined by
// atomic adds in arrival order (the last bits of a LONG row's y may differ between two calls, as with PANEL).
//
// The reference has nothing to mirror here: its CSR loop gives a row to one thread whatever its length (src/mat_vec.cpp:57-65).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kLongChunk = 4096;  // entries per workgroup: 16 per lane, 4 loads in flight each

__global__ __launch_bounds__(kBlock) void csr_long_rows_kernel(const int32_t* __restrict__ chunk_row, const int32_t* __restrict__ chunk_beg,
                                                               const int32_t* __restrict__ chunk_end, const int32_t* __restrict__ col,
                                                               const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double part[kBlock / kWave];
    const int c = blockIdx.x;
    const int b = chunk_beg[c], e = chunk_end[c];
    double    s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int       j = b + (int)threadIdx.x;
    for (; j + 3 * kBlock < e; j += 4 * kBlock)
    {
        const int    c0 = load_stream(col + j), c1 = load_stream(col + j + kBlock), c2 = load_stream(col + j + 2 * kBlock), c3 = load_stream(col + j + 3 * kBlock);
        const double v0 = load_stream(val + j), v1 = load_stream(val + j + kBlock), v2 = load_stream(val + j + 2 * kBlock), v3 = load_stream(val + j + 3 * kBlock);
        s0 += v0 * x[c0];
        s1 += v1 * x[c1];
        s2 += v2 * x[c2];
        s3 += v3 * x[c3];
    }
    for (; j < e; j += kBlock) s0 += load_stream(val + j) * x[load_stream(col + j)];
    const double w = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = part[0];
#pragma unroll
        for (int i = 1; i < kBlock / kWave; ++i) t += part[i];
        unsafeAtomicAdd(y + chunk_row[c], t);
    }
}

// the rows the copy keeps: one group of LPR lanes per row
template <int LPR>
__global__ __launch_bounds__(kBlock) void split_copy_kernel(int nrow, const int32_t* __restrict__ src_ptr, const int32_t* __restrict__ dst_ptr,
                                                            const int32_t* __restrict__ col, const double* __restrict__ val, int32_t* __restrict__ out_col,
                                                            double* __restrict__ out_val)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int     r   = (int)(gid / LPR);
    const int     l   = (int)(gid % LPR);
    if (r >= nrow) return;
    const int d = dst_ptr[r], n = dst_ptr[r + 1] - d, s = src_ptr[r];
    for (int j = l; j < n; j += LPR)
    {
        out_col[d + j] = col[s + j];
        out_val[d + j] = val[s + j];
    }
}
}  // namespace

void csr_split_free(spmv_mat* m)
{
    if (m->format != SPMV_FMT_CSR) return;
    if (m->split_chunks)
    {
        (void)hipFree(m->split_chunks);
        m->device_bytes -= (int64_t)sizeof(int32_t) * 3 * m->split_nchunks;
        m->split_chunks = nullptr;
    }
    if (m->coo_csr)
    {
        m->device_bytes -= m->coo_csr->device_bytes;
        mat_free(m->coo_csr);
        m->coo_csr = nullptr;
    }
    m->split_nchunks = 0;
    m->split_long_rows = 0;
    m->split_long_nnz = 0;
    m->split_built_threshold = 0;
}

int csr_split_threshold(const spmv_mat* m)
{
    if (m->split_threshold > 0) return m->split_threshold;
    // a sixteenth of the longest row, at least a chunk: what stays behind is at most a few workgroup-chunks of work per row
    return std::max(kLongChunk, m->max_row_nnz / 16);
}

int csr_split_build(spmv_mat* m)
{
    SPMV_REQUIRE(m->format == SPMV_FMT_CSR && m->a && m->b && m->v, "the long-row split is built from a CSR handle's own arrays");
    const int T = csr_split_threshold(m);
    if (m->coo_csr && m->split_built_threshold == T) return SPMV_OK;
    (void)hipStreamSynchronize(m->ctx->stream);
    csr_split_free(m);
    spmv_ctx*   ctx = m->ctx;
    hipStream_t s   = ctx->stream;
    const int   n   = m->nrow;
    // row lengths through the host: one pass, one-off (8M rows: 32 MB)
    std::vector<int32_t> rp((size_t)n + 1), dst((size_t)n + 1), chunks;
    SPMV_HIP(hipMemcpyAsync(rp.data(), m->a, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
    SPMV_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> crow, cbeg, cend;
    int64_t              kept = 0, long_nnz = 0;
    int                  long_rows = 0;
    for (int r = 0; r < n; ++r)
    {
        const int len  = rp[(size_t)r + 1] - rp[(size_t)r];
        dst[(size_t)r] = (int32_t)kept;
        if (len >= T)
        {
            ++long_rows;
            long_nnz += len;
            for (int b = rp[(size_t)r]; b < rp[(size_t)r + 1]; b += kLongChunk)
            {
                crow.push_back(r);
                cbeg.push_back(b);
                cend.push_back(std::min(b + kLongChunk, rp[(size_t)r + 1]));
            }
        }
        else
            kept += len;
    }
    dst[(size_t)n] = (int32_t)kept;
    spmv_mat* rest = nullptr;
    SPMV_TRY(mat_alloc(ctx, SPMV_FMT_CSR, n, m->ncol, kept, 0, (size_t)n + 1, (size_t)kept, (size_t)kept, &rest));
    int rc = SPMV_OK;
    do
    {
        if (hipMemcpyAsync(const_cast<int32_t*>(rest->a), dst.data(), sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (n > 0 && kept > 0)
        {
            constexpr int LPR = 8;
            if (!launch_fits(n, LPR))
            {
                set_error("long-row split: %d rows are more than one launch of the copy holds", n);
                rc = SPMV_ERR_UNSUPPORTED;
                break;
            }
            hipLaunchKernelGGL(split_copy_kernel<LPR>, dim3((unsigned)ceil_div((int64_t)n * LPR, kBlock)), dim3(kBlock), 0, s, n, m->a, rest->a, m->b, m->v,
                               const_cast<int32_t*>(rest->b), const_cast<double*>(rest->v));
        }
        const size_t nc = crow.size();
        if (nc > 0)
        {
            if (hipMalloc(&m->split_chunks, sizeof(int32_t) * 3 * nc) != hipSuccess)
            {
                m->split_chunks = nullptr;
                rc              = SPMV_ERR_ALLOC;
                break;
            }
            m->split_nchunks = (int32_t)nc;
            m->device_bytes += (int64_t)sizeof(int32_t) * 3 * (int64_t)nc;
            if (hipMemcpyAsync(m->split_chunks, crow.data(), sizeof(int32_t) * nc, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipMemcpyAsync(m->split_chunks + nc, cbeg.data(), sizeof(int32_t) * nc, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipMemcpyAsync(m->split_chunks + 2 * nc, cend.data(), sizeof(int32_t) * nc, hipMemcpyHostToDevice, s) != hipSuccess)
                rc = SPMV_ERR_HIP;
        }
        if (hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    if (hipStreamSynchronize(s) != hipSuccess && rc == SPMV_OK) rc = SPMV_ERR_HIP;  // (the host vectors go out of scope below)
    if (rc == SPMV_OK)
    {
        rest->row_begin      = m->row_begin;
        rest->pb_trial       = m->pb_trial;
        rest->sel_no_split   = true;  // (its longest row is below the threshold by construction; and no split of a split)
        rest->sel_no_segscan = true;  // what the scan is for went out with the long rows
        rc                   = csr_analyse(rest);  // picks the copy's kernel and builds its layout
    }
    if (rc != SPMV_OK)
    {
        mat_free(rest);
        csr_split_free(m);
        if (rc == SPMV_ERR_HIP) set_error("building the long-row split failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc == SPMV_ERR_ALLOC) set_error("no device memory for the long-row split of %lld entries", (long long)m->nnz);
        return rc;
    }
    // the panel and two-phase layouts read row_ptr and their own arrays only
    if ((rest->kernel == SPMV_CSR_PANEL || rest->kernel == SPMV_CSR_TWOPHASE) && rest->b && rest->v && kept > 0)
    {
        (void)hipFree(const_cast<int32_t*>(rest->b));
        (void)hipFree(const_cast<double*>(rest->v));
        rest->device_bytes -= kept * 12;
        rest->b = nullptr;
        rest->v = nullptr;
    }
    m->coo_csr               = rest;
    m->device_bytes         += rest->device_bytes;
    m->split_long_rows       = long_rows;
    m->split_long_nnz        = long_nnz;
    m->split_built_threshold = T;
    return SPMV_OK;
}

int csr_split_long_rows_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->split_nchunks == 0) return SPMV_OK;
    const size_t nc = (size_t)A->split_nchunks;
    hipLaunchKernelGGL(csr_long_rows_kernel, dim3((unsigned)nc), dim3(kBlock), 0, ctx->stream, A->split_chunks, A->split_chunks + nc, A->split_chunks + 2 * nc,
                       A->b, A->v, x, y);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int csr_split_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (!A->coo_csr) SPMV_FAIL(SPMV_ERR_INVALID, "long-row split selected but never built");
    SPMV_TRY(csr_apply(ctx, A->coo_csr, x, y));
    return csr_split_long_rows_apply(ctx, A, x, y);
}
}  // namespace spmv
