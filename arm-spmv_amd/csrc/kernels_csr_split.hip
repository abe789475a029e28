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
// That is "split_mode" 1, right for DENSE long rows (an entry in most 128-byte lines of x they span).  Long rows of a power law
// or an R-MAT graph are long AND sparse: 8192 entries spread over 1M columns gather a line of x per entry, in row order, from
// an x beyond L2 - what the panel layout exists to avoid (tools/probe_split_threshold.py: the more rows went through the
// chunks, the slower the product).  "split_mode" 2 therefore turns every long row into V = ceil(len / 64) VIRTUAL rows with
// the entries dealt out in turn (entry j of the row goes to virtual row j % V): neighbours in x land in different virtual
// rows, every virtual row looks like an ordinary short row, and the virtual rows of all long rows form a second CSR matrix
// with a handle - and a kernel, as a rule the panel layout - of its own.  Its product goes to a scratch vector (overwrite),
// and one wavefront per long row adds the V partial sums up in a fixed order and onto y.  Mode 0 picks: chunks where the long
// rows hold an entry per 2 columns on average, virtual rows otherwise.
//
// The reference has nothing to mirror here: its CSR loop gives a row to one thread whatever its length (src/mat_vec.cpp:57-65).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "split_rows.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kLongChunk = 4096;  // entries per workgroup: 16 per lane, 4 loads in flight each
constexpr int kVirtualLen = 64;  // mode 2: entries per virtual row

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

// mode 2: deal the entries of the long rows out to their virtual rows.  One workgroup per chunk of a long row (the chunk
// table's row column holds the INDEX of the long row here); lrow = first entry | V | first virtual row, per long row
__global__ __launch_bounds__(kBlock) void split_deal_kernel(const int32_t* __restrict__ chunk_long, const int32_t* __restrict__ chunk_beg,
                                                            const int32_t* __restrict__ chunk_end, const int32_t* __restrict__ lbeg,
                                                            const int32_t* __restrict__ lv, const int32_t* __restrict__ lbase,
                                                            const int32_t* __restrict__ vptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                            int32_t* __restrict__ out_col, double* __restrict__ out_val)
{
    const int c = blockIdx.x, li = chunk_long[c];
    const int first = lbeg[li], V = lv[li], base = lbase[li];
    for (int j = chunk_beg[c] + (int)threadIdx.x; j < chunk_end[c]; j += kBlock)
    {
        const int k = j - first;
        const int d = vptr[base + k % V] + k / V;
        out_col[d]  = col[j];
        out_val[d]  = val[j];
    }
}

// mode 2: y[row] += the partial sums of the row's virtual rows, one wavefront per long row, a fixed order
__global__ __launch_bounds__(kBlock) void split_combine_kernel(int nlong, const int32_t* __restrict__ lrow, const int32_t* __restrict__ lv,
                                                               const int32_t* __restrict__ lbase, const double* __restrict__ yl, double* __restrict__ y)
{
    const int li = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
    if (li >= nlong) return;  // (uniform over the wavefront)
    const int V = lv[li], base = lbase[li];
    double    s = 0.0;
    for (int v = lane_id(); v < V; v += kWave) s += yl[base + v];
    s = wave_sum(s);
    if (lane_id() == 0) y[lrow[li]] += s;
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
    if (m->split_long)
    {
        m->device_bytes -= m->split_long->device_bytes;
        mat_free(m->split_long);
        m->split_long = nullptr;
    }
    if (m->split_yl)
    {
        (void)hipFree(m->split_yl);
        m->device_bytes -= (int64_t)sizeof(double) * m->split_vrows;
        m->split_yl = nullptr;
    }
    if (m->split_rows)
    {
        (void)hipFree(m->split_rows);
        m->device_bytes -= (int64_t)sizeof(int32_t) * 4 * m->split_long_rows;
        m->split_rows = nullptr;
    }
    m->split_vrows = 0;
    m->split_built_mode = 0;
    m->split_built_for_mode = 0;
    m->split_nchunks = 0;
    m->split_long_rows = 0;
    m->split_long_nnz = 0;
    m->split_built_threshold = 0;
}

int csr_split_threshold(const spmv_mat* m)
{
    if (m->split_threshold > 0) return m->split_threshold;
    if (m->split_auto_low) return 256;  // what AUTO found faster on this handle (select.hip)
    // a sixteenth of the longest row, at least a chunk: what stays behind is at most a few workgroup-chunks of work per row
    return std::max(kLongChunk, m->max_row_nnz / 16);
}

int csr_split_build(spmv_mat* m)
{
    SPMV_REQUIRE(m->format == SPMV_FMT_CSR && m->a && m->b && m->v, "the long-row split is built from a CSR handle's own arrays");
    const int T = csr_split_threshold(m);
    if (m->coo_csr && m->split_built_threshold == T && m->split_mode == m->split_built_for_mode) return SPMV_OK;  // (the request it was built for)
    (void)hipStreamSynchronize(m->ctx->stream);
    csr_split_free(m);
    spmv_ctx*   ctx = m->ctx;
    hipStream_t s   = ctx->stream;
    const int   n   = m->nrow;
    // row lengths through the host: one pass, one-off (8M rows: 32 MB)
    std::vector<int32_t> rp((size_t)n + 1), dst((size_t)n + 1);
    SPMV_HIP(hipMemcpyAsync(rp.data(), m->a, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
    SPMV_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> lrow;  // the long rows
    int64_t              kept = 0, long_nnz = 0;
    for (int r = 0; r < n; ++r)
    {
        const int len  = rp[(size_t)r + 1] - rp[(size_t)r];
        dst[(size_t)r] = (int32_t)kept;
        if (len >= T)
        {
            lrow.push_back(r);
            long_nnz += len;
        }
        else
            kept += len;
    }
    dst[(size_t)n]     = (int32_t)kept;
    const size_t nlong = lrow.size();
    // chunks or virtual rows: dense long rows read x nearly coalesced in row order; sparse ones want the panel layout's order
    int mode = m->split_mode;
    // (1M rows, lengths min(500000, 8 / u): the 260 longest rows hold an entry per 9 columns - chunks 0.46 ms, virtual rows 0.28;
    // 8 rows with an entry in every column among 500000 of 32: chunks 0.079, virtual rows 0.119)
    if (mode != 1 && mode != 2) mode = (double)long_nnz * 2.0 >= (double)nlong * (double)m->ncol ? 1 : 2;
    // chunk table (mode 1: row | begin | end; mode 2: index of the long row | begin | end, for the deal kernel)
    std::vector<int32_t> crow, cbeg, cend, lbeg(nlong), lv, lbase, vptr;
    std::vector<int64_t> llen(nlong);
    for (size_t i = 0; i < nlong; ++i)
    {
        const int r = lrow[i], b0 = rp[(size_t)r], e0 = rp[(size_t)r + 1];
        for (int b = b0; b < e0; b += kLongChunk)
        {
            crow.push_back(mode == 1 ? r : (int32_t)i);
            cbeg.push_back(b);
            cend.push_back(std::min(b + kLongChunk, e0));
        }
        lbeg[i] = b0;
        llen[i] = e0 - b0;
    }
    split_virtual_row_ptr(llen, kVirtualLen, &lv, &lbase, &vptr);  // (split_rows.hpp: V = ceil(len / 64) virtual rows per long row)
    const int64_t nv = (int64_t)vptr.size() - 1;
    spmv_mat *rest = nullptr, *lng = nullptr;
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
        if (nc == 0) break;
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
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (mode == 1) break;
        // mode 2: the virtual rows as a CSR matrix of their own
        if (hipMalloc(&m->split_rows, sizeof(int32_t) * 4 * nlong) != hipSuccess)
        {
            m->split_rows = nullptr;
            rc            = SPMV_ERR_ALLOC;
            break;
        }
        m->split_long_rows = (int32_t)nlong;  // (csr_split_free accounts for split_rows with it)
        m->device_bytes += (int64_t)sizeof(int32_t) * 4 * (int64_t)nlong;
        if (hipMalloc(&m->split_yl, sizeof(double) * (size_t)nv) != hipSuccess)
        {
            m->split_yl = nullptr;
            rc          = SPMV_ERR_ALLOC;
            break;
        }
        m->split_vrows = (int32_t)nv;
        m->device_bytes += (int64_t)sizeof(double) * nv;
        if ((rc = mat_alloc(ctx, SPMV_FMT_CSR, (int32_t)nv, m->ncol, long_nnz, 0, (size_t)nv + 1, (size_t)long_nnz, (size_t)long_nnz, &lng)) != SPMV_OK) break;
        int32_t* sr = m->split_rows;
        if (hipMemcpyAsync(sr, lrow.data(), sizeof(int32_t) * nlong, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(sr + nlong, lbeg.data(), sizeof(int32_t) * nlong, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(sr + 2 * nlong, lv.data(), sizeof(int32_t) * nlong, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(sr + 3 * nlong, lbase.data(), sizeof(int32_t) * nlong, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(const_cast<int32_t*>(lng->a), vptr.data(), sizeof(int32_t) * ((size_t)nv + 1), hipMemcpyHostToDevice, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(split_deal_kernel, dim3((unsigned)nc), dim3(kBlock), 0, s, m->split_chunks, m->split_chunks + nc, m->split_chunks + 2 * nc, sr + nlong,
                           sr + 2 * nlong, sr + 3 * nlong, lng->a, m->b, m->v, const_cast<int32_t*>(lng->b), const_cast<double*>(lng->v));
    } while (0);
    if (rc == SPMV_OK && hipGetLastError() != hipSuccess) rc = SPMV_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess && rc == SPMV_OK) rc = SPMV_ERR_HIP;  // (the host vectors go out of scope below)
    for (spmv_mat* part : {rest, lng})
        if (rc == SPMV_OK && part)
        {
            part->row_begin      = part == rest ? m->row_begin : 0;
            part->pb_trial       = m->pb_trial;
            part->sel_no_split   = true;  // (their longest rows are short by construction; and no split of a split)
            part->sel_no_segscan = true;  // what the scan is for went out with the long rows
            plan_hand_down(m, part, part == rest ? kPlanChildRowgrouped : kPlanChildLong);
            rc                   = csr_analyse(part);  // picks the part's kernel and builds its layout
            // the panel and two-phase layouts read row_ptr and their own arrays only
            if (rc == SPMV_OK && (part->kernel == SPMV_CSR_PANEL || part->kernel == SPMV_CSR_TWOPHASE || part->kernel == SPMV_CSR_ELL) && part->b && part->v && part->nnz > 0)
            {
                (void)hipFree(const_cast<int32_t*>(part->b));
                (void)hipFree(const_cast<double*>(part->v));
                part->device_bytes -= part->nnz * 12;
                part->b = nullptr;
                part->v = nullptr;
            }
        }
    if (rc != SPMV_OK)
    {
        mat_free(rest);
        if (lng) mat_free(lng);
        csr_split_free(m);
        if (rc == SPMV_ERR_HIP) set_error("building the long-row split failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc == SPMV_ERR_ALLOC) set_error("no device memory for the long-row split of %lld entries", (long long)m->nnz);
        return rc;
    }
    m->coo_csr = rest;
    m->device_bytes += rest->device_bytes;
    if (lng)
    {
        m->split_long = lng;
        m->device_bytes += lng->device_bytes;
        // the chunk table was for the deal kernel only
        (void)hipFree(m->split_chunks);
        m->device_bytes -= (int64_t)sizeof(int32_t) * 3 * m->split_nchunks;
        m->split_chunks  = nullptr;
        m->split_nchunks = 0;
    }
    m->split_long_rows       = (int32_t)nlong;
    m->split_long_nnz        = long_nnz;
    m->split_built_threshold = T;
    m->split_built_mode      = mode;
    m->split_built_for_mode  = m->split_mode;
    return SPMV_OK;
}

int csr_split_long_rows_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->split_long)
    {
        const size_t nl = (size_t)A->split_long_rows;
        apply_extra  over;
        over.overwrite = true;
        SPMV_TRY(mat_apply_ex(ctx, A->split_long, x, A->split_yl, over));
        hipLaunchKernelGGL(split_combine_kernel, dim3((unsigned)ceil_div((int64_t)nl * kWave, kBlock)), dim3(kBlock), 0, ctx->stream, (int)nl, A->split_rows,
                           A->split_rows + 2 * nl, A->split_rows + 3 * nl, A->split_yl, y);
        SPMV_HIP(hipGetLastError());
        return SPMV_OK;
    }
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
