// kernels_csr.hip — y += A*x for CSR on CDNA4 (gfx950).
//
// Replaces CSRMatrixMatVector (reference src/mat_vec.cpp:44-67; hot loop :60-63) and the
// per-thread body of its NUMA driver (src/mat_vec.cpp:507-530).
//
// Roofline: HBM-bound.  Algorithmic bytes per application:
//     12*nnz (values + column indices, read once)  + 4*(nrow+1) (row_ptr)
//   +  8*ncol (x, counted once)                    + 16*nrow    (y read + write, `+=`)
//
// Kernels
//   csr_vector_kernel<LPR>  LPR = 2^k lanes cooperate on one row (a 64-lane wavefront holds 64/LPR
//                           rows); lane l of the group walks entries l, l+LPR, ... so each load
//                           instruction reads 64/LPR contiguous runs of LPR*8 bytes; four
//                           independent (col, val, x) triples are in flight per lane before the first
//                           fma; partial sums meet through ds_swizzle/ds_bpermute (or DPP).
//   csr_scalar_kernel       one lane per row, strictly left to right with fma: bit-identical to the
//                           oracle's orc_csr_spmv_fma.  Used for parity pinning and for very short rows.
//   csr_ldswin_kernel       banded matrices: a workgroup stages the x window its rows touch into LDS
//                           with coalesced loads, then gathers from LDS instead of L2/HBM.
#include <atomic>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
// Blocks are dealt round-robin to the 8 XCDs (block b and b+8 share an L2).  With XCD_REMAP each XCD
// walks one contiguous eighth of the rows, so neighbouring row blocks — which on banded matrices
// touch overlapping x windows — share an L2.  Speed only; any placement is correct.
__device__ __forceinline__ int remap_block(int bid, int nblocks)
{
    const int per = nblocks / kNumXcd;  // blocks in the evenly divisible part
    const int cut = per * kNumXcd;
    if (bid >= cut) return bid;  // ragged tail keeps its id
    return (bid % kNumXcd) * per + bid / kNumXcd;
}

template <int LPR, bool USE_DPP, bool XCD_REMAP>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(
    int nrow, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, int overwrite,
    const double* __restrict__ dot_w, double* __restrict__ dot_out)
{
    constexpr int ROWS = kBlock / LPR;
    int bid = blockIdx.x;
    if constexpr (XCD_REMAP) bid = remap_block(bid, gridDim.x);
    const int row = bid * ROWS + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;

    double sum = 0.0;
    if (row < nrow)
    {
        int       j   = row_ptr[row] + sub;
        const int end = row_ptr[row + 1];
        for (; j + 3 * LPR < end; j += 4 * LPR)
        {
            const int    c0 = load_stream(col + j);
            const int    c1 = load_stream(col + j + LPR);
            const int    c2 = load_stream(col + j + 2 * LPR);
            const int    c3 = load_stream(col + j + 3 * LPR);
            const double v0 = load_stream(val + j);
            const double v1 = load_stream(val + j + LPR);
            const double v2 = load_stream(val + j + 2 * LPR);
            const double v3 = load_stream(val + j + 3 * LPR);
            const double x0 = x[c0];
            const double x1 = x[c1];
            const double x2 = x[c2];
            const double x3 = x[c3];
            sum = fma(v0, x0, sum);
            sum = fma(v1, x1, sum);
            sum = fma(v2, x2, sum);
            sum = fma(v3, x3, sum);
        }
        for (; j < end; j += LPR) sum = fma(load_stream(val + j), x[load_stream(col + j)], sum);
    }
    // every lane of the wave takes part in the cross-lane step (no early exit above)
    sum = group_sum<LPR, USE_DPP>(sum);
    // y += sum (the reference's op), or y = sum; the solver's dot product w . y_new rides along (spmv_apply_dot)
    double part = 0.0;
    if (row < nrow && sub == 0)
    {
        const double yn = overwrite ? sum : y[row] + sum;
        y[row]          = yn;
        if (dot_w) part = dot_w[row] * yn;
    }
    if (dot_w)  // uniform over the grid
    {
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0) slot_add(dot_out, part);
    }
}

__global__ __launch_bounds__(kBlock) void csr_scalar_kernel(
    int nrow, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= nrow) return;
    double    sum = 0.0;
    const int end = row_ptr[row + 1];
    for (int j = row_ptr[row]; j < end; ++j) sum = fma(val[j], x[col[j]], sum);
    y[row] += sum;
}

// ---- LDS-window kernel ---------------------------------------------------------------------------------
// A workgroup owns WIN_ROWS consecutive rows.  csr_window_scan_kernel (run once, at analysis) records the
// smallest column `lo` and the span touched by those rows.  If the span fits the LDS tile the workgroup
// copies x[lo, lo+span) into LDS with fully coalesced loads (each x element is fetched once per
// workgroup instead of once per nonzero) and the gathers become ds_read_b64.
constexpr int kWinRows    = 256;       // rows per workgroup (LPR lanes each -> loop over row slabs)
constexpr int kWinDoubles = 16 * 1024; // 128 KiB x tile; leaves 32 KiB of the CU's 160 KiB unused

// Also counts the entries whose column is the previous entry's + 1 (`contig`; the first entry of a block never counts, an
// entry that continues the previous ROW's last column does - one in a row's length, immaterial): rows made of contiguous
// runs read x coalesced under the row-parallel kernel (dense blocks: tools/sweep_structures.py).
__global__ __launch_bounds__(kBlock) void csr_window_scan_kernel(
    int nrow, int win_rows, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    int32_t* __restrict__ win_lo, int32_t* __restrict__ win_span, unsigned long long* __restrict__ contig)
{
    const int b     = blockIdx.x;
    const int r0    = b * win_rows;
    const int r1    = min(nrow, r0 + win_rows);
    const int begin = row_ptr[r0];
    const int end   = row_ptr[r1];
    int       lo = INT32_MAX, hi = -1;
    unsigned  runs = 0;
    for (int j = begin + threadIdx.x; j < end; j += kBlock)
    {
        const int c = col[j];
        lo          = min(lo, c);
        hi          = max(hi, c);
        if (j > begin && col[j - 1] + 1 == c) ++runs;
    }
    for (int off = 32; off > 0; off >>= 1) runs += __shfl_xor(runs, off);
    if ((threadIdx.x & 63) == 0 && runs) atomicAdd(contig, (unsigned long long)runs);
    __shared__ int s_lo[kBlock / kWave], s_hi[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1)
    {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0)
    {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int w = 1; w < kBlock / kWave; ++w)
        {
            lo = min(lo, s_lo[w]);
            hi = max(hi, s_hi[w]);
        }
        if (hi < 0)
        {
            lo = 0;
            hi = -1;
        }
        win_lo[b]   = lo;
        win_span[b] = hi - lo + 1;
    }
}

template <int LPR, bool USE_DPP>
__global__ __launch_bounds__(kBlock) void csr_ldswin_kernel(
    int nrow, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
    const int32_t* __restrict__ win_lo, const int32_t* __restrict__ win_span)
{
    extern __shared__ double xs[];  // win_max_span doubles (<= kWinDoubles)
    int bid = remap_block(blockIdx.x, gridDim.x);
    const int lo   = win_lo[bid];
    const int span = win_span[bid];
    // stage x[lo, lo+span): lane-contiguous 8-byte loads; x stays default-policy (it is re-read by
    // the neighbouring workgroups from L2)
    for (int i = threadIdx.x; i < span; i += kBlock) xs[i] = x[lo + i];
    __syncthreads();

    constexpr int ROWS = kBlock / LPR;
    const int     sub  = threadIdx.x % LPR;
    const int     r0   = bid * kWinRows;
#pragma unroll 1
    for (int slab = 0; slab < kWinRows; slab += ROWS)
    {
        const int row = r0 + slab + threadIdx.x / LPR;
        double    sum = 0.0;
        if (row < nrow)
        {
            int       j   = row_ptr[row] + sub;
            const int end = row_ptr[row + 1];
            for (; j + 3 * LPR < end; j += 4 * LPR)
            {
                const int    c0 = load_stream(col + j) - lo;
                const int    c1 = load_stream(col + j + LPR) - lo;
                const int    c2 = load_stream(col + j + 2 * LPR) - lo;
                const int    c3 = load_stream(col + j + 3 * LPR) - lo;
                const double v0 = load_stream(val + j);
                const double v1 = load_stream(val + j + LPR);
                const double v2 = load_stream(val + j + 2 * LPR);
                const double v3 = load_stream(val + j + 3 * LPR);
                sum = fma(v0, xs[c0], sum);
                sum = fma(v1, xs[c1], sum);
                sum = fma(v2, xs[c2], sum);
                sum = fma(v3, xs[c3], sum);
            }
            for (; j < end; j += LPR)
                sum = fma(load_stream(val + j), xs[load_stream(col + j) - lo], sum);
        }
        sum = group_sum<LPR, USE_DPP>(sum);
        if (row < nrow && sub == 0) y[row] += sum;
    }
}

// out[0] = longest row, out[1] = shortest row (initialised to 0 / INT32_MAX by the caller)
__global__ __launch_bounds__(kBlock) void row_len_max_kernel(int nrow, const int32_t* __restrict__ row_ptr,
                                                             int32_t* __restrict__ out)
{
    int mx = 0, mn = INT32_MAX;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nrow; i += (int64_t)gridDim.x * kBlock)
    {
        const int len = row_ptr[i + 1] - row_ptr[i];
        mx            = max(mx, len);
        mn            = min(mn, len);
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        mx = max(mx, __shfl_xor(mx, off));
        mn = min(mn, __shfl_xor(mn, off));
    }
    if ((threadIdx.x & 63) == 0)
    {
        if (mx > 0) atomicMax(out, mx);
        atomicMin(out + 1, mn);
    }
}

__global__ void sum_i32_kernel(const int32_t* __restrict__ in, int n, unsigned long long* __restrict__ out_sum)
{
    unsigned long long acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += (unsigned)in[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out_sum, acc);
}

__global__ void max_i32_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out_max)
{
    int mx = INT32_MIN;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) mx = max(mx, in[i]);
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_max, mx);
}

int pick_lanes(double mean_row)
{
    // about four entries per lane: enough independent gathers per lane to cover the latency of x,
    // few enough lanes per row that 64/LPR rows keep a wavefront's loads spread over few lines
    int lanes = 1;
    while (lanes < 64 && lanes * 4 < mean_row) lanes <<= 1;
    return lanes;
}

template <bool USE_DPP, bool XCD_REMAP>
int launch_vector(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, int lanes, const apply_extra& ex = apply_extra{})
{
    const int   nrow = A->nrow;
    hipStream_t s    = ctx->stream;
    // one lane group per row: more than 2^32 lanes in a launch wrap around silently — fewer lanes per row then
    while (lanes > 1 && (int64_t)nrow * lanes >= ((int64_t)1 << 32) - kBlock) lanes >>= 1;
#define SPMV_LAUNCH_LPR(L)                                                                              \
    case L:                                                                                             \
        hipLaunchKernelGGL((csr_vector_kernel<L, USE_DPP, XCD_REMAP>), dim3((unsigned)ceil_div(nrow, kBlock / L)), \
                           dim3(kBlock), 0, s, nrow, A->a, A->b, A->v, x, y, ex.overwrite ? 1 : 0,      \
                           ex.dot_w, ex.dot_out);                                                       \
        break;
    switch (lanes)
    {
        SPMV_LAUNCH_LPR(1)
        SPMV_LAUNCH_LPR(2)
        SPMV_LAUNCH_LPR(4)
        SPMV_LAUNCH_LPR(8)
        SPMV_LAUNCH_LPR(16)
        SPMV_LAUNCH_LPR(32)
        SPMV_LAUNCH_LPR(64)
        default: SPMV_FAIL(SPMV_ERR_INVALID, "lanes_per_row must be a power of two in 1..64, got %d", lanes);
    }
#undef SPMV_LAUNCH_LPR
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int launch_ldswin(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, int lanes)
{
    const int    nblocks = (int)ceil_div(A->nrow, kWinRows);
    const size_t lds     = (size_t)A->win_max_span * sizeof(double);
    hipStream_t  s       = ctx->stream;
    // more than 64 KiB of dynamic LDS has to be granted per kernel; once per instantiation is enough
#define SPMV_LAUNCH_LPR(L)                                                                                      \
    case L:                                                                                                     \
    {                                                                                                           \
        static std::atomic<unsigned long long> granted{0}; /* bit per device */                                                                            \
        if (!((granted.load(std::memory_order_relaxed) >> ctx->device) & 1ull))                                                                                         \
        {                                                                                                       \
            SPMV_HIP(hipFuncSetAttribute((const void*)csr_ldswin_kernel<L, false>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kWinDoubles * 8));         \
            granted.fetch_or(1ull << ctx->device, std::memory_order_relaxed);                                                                                     \
        }                                                                                                       \
        hipLaunchKernelGGL((csr_ldswin_kernel<L, false>), dim3(nblocks), dim3(kBlock), lds, s, A->nrow, A->a, A->b, \
                           A->v, x, y, A->win_lo, A->win_span);                                                 \
        break;                                                                                                  \
    }
    switch (lanes)
    {
        SPMV_LAUNCH_LPR(1)
        SPMV_LAUNCH_LPR(2)
        SPMV_LAUNCH_LPR(4)
        SPMV_LAUNCH_LPR(8)
        SPMV_LAUNCH_LPR(16)
        SPMV_LAUNCH_LPR(32)
        SPMV_LAUNCH_LPR(64)
        default: SPMV_FAIL(SPMV_ERR_INVALID, "lanes_per_row must be a power of two in 1..64, got %d", lanes);
    }
#undef SPMV_LAUNCH_LPR
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace

int reduce_max_i32(spmv_ctx* ctx, const int32_t* in, int64_t n, int32_t* result)
{
    SPMV_TRY(ensure_scratch(ctx, 64));
    int32_t* d  = (int32_t*)ctx->scratch;
    int32_t  lo = INT32_MIN;
    SPMV_HIP(hipMemcpyAsync(d, &lo, sizeof(lo), hipMemcpyHostToDevice, ctx->stream));
    if (n > 0)
    {
        const int grid = (int)std::min<int64_t>(kMaxGrid, ceil_div(n, kBlock));
        hipLaunchKernelGGL(max_i32_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, in, (int)n, d);
        SPMV_HIP(hipGetLastError());
    }
    SPMV_HIP(hipMemcpyAsync(result, d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    return SPMV_OK;
}

// AUTO, the model (no launches; select.hip times candidates on top of it where that pays).  Measured at 32 entries/row in
// round 1 (profiles/r01_tune_*) and audited on stencils, dense blocks, R-MAT graphs, rectangles and permutations in round 5
// (tools/sweep_structures.py, profiles/r05_sweep_structures_*.txt):
//   * enough entries to occupy 256 workgroups of 1024 lanes -> the panel layout.  It beats the row-parallel kernel
//     without column locality (N = 10M uniform: 1.63 vs 5.9 ms), with it (0.76 vs 1.63 ms in a 4096-wide band,
//     0.76 vs 2.0 ms at 65536; the column-sorted walk makes neighbouring lanes share x lines), while x still fits
//     L2 (N = 100k..1.5M: 1.3x..2.8x), and it beats the LDS-window kernel where that applies (N = 4M band 4096
//     without wrap-around rows: 0.32 vs 0.43 ms).  "Enough" is 1.5M entries (round 5: 27-point stencil with 1.64M entries
//     0.0087 vs 0.0116 ms, 200000 x 5000 with 1.6M 0.0088 vs 0.0114; 8-wide blocks with 1.28M 0.0083 vs 0.0066), whatever
//     the mean row length (a permutation of 8M rows: 0.136 vs 0.171; the two-phase layout 0.092);
//   * EXCEPT long rows made of contiguous runs (dense blocks of 32 and more: 64 x 64 blocks 0.093 vs 0.116): the
//     row-parallel kernel reads their x coalesced and adds in registers;
//   * a few hub rows among short ones (R-MAT: 8436 entries in one row, 16 on average) serialise on the lanes of ONE row
//     group under the row-parallel kernel (0.177 ms vs 0.016): the panel layout from 64K entries on;
//   * smaller: stage x in LDS when every row block's column window fits the tile and is re-used, else gather
//     through L1/L2 with the row-parallel kernel.
void csr_choose_kernel(spmv_mat* m)
{
    const double mean       = m->nrow > 0 ? (double)m->nnz / (double)m->nrow : 0.0;
    const bool   big_enough = m->nnz >= (int64_t)3 << 19 && mean >= 0.5;
    const bool   fits       = m->win_max_span > 0 && m->win_max_span <= kWinDoubles;
    const double reuse      = m->win_max_span > 0 ? mean * kWinRows / (double)m->win_max_span : 0.0;
    const bool   dense_runs = m->contig_frac >= 0.8 && mean >= 24.0;
    const bool   hub_rows   = m->nnz >= (int64_t)64 << 10 && m->max_row_nnz >= 1024 && (double)m->max_row_nnz >= 32.0 * std::max(mean, 1.0);
    // one row with 1/64 of the entries, and more of them than a workgroup gets through in the time of a product (arrow shapes,
    // dense constraint rows: 4 rows of 200000 among 200000 of 16, panel 0.256 ms, segmented scan 0.049, long rows split off 0.027)
    const bool   long_row   = !m->sel_no_split && m->max_row_nnz >= 65536 && (int64_t)m->max_row_nnz * 64 >= m->nnz;
    if (long_row)
        m->kernel = SPMV_CSR_SPLIT;
    else if (big_enough && !dense_runs)
        m->kernel = csr_twophase_worth(m) ? SPMV_CSR_TWOPHASE : SPMV_CSR_PANEL;
    else if (hub_rows)
        m->kernel = SPMV_CSR_PANEL;
    else if (fits && reuse >= 2.0 && !big_enough)
        m->kernel = SPMV_CSR_LDSWIN;
    else
        m->kernel = SPMV_CSR_VECTOR;
}

int csr_ldswin_capacity() { return kWinDoubles; }

// Row statistics + kernel choice.  Runs once when a CSR handle is created.
int csr_analyse(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    // the kernels step their entry index in int32 (j + 3*LPR, e + 3*1024): keep clear of the wrap
    SPMV_REQUIRE(m->nnz <= (int64_t)INT32_MAX - 65536, "CSR handle with %lld entries: shard it (int32 offsets)",
                 (long long)m->nnz);
    SPMV_TRY(ensure_scratch(ctx, 64));
    int32_t* d_max = (int32_t*)ctx->scratch;
    int32_t h_len[2] = {0, INT32_MAX};
    SPMV_HIP(hipMemcpyAsync(d_max, h_len, sizeof(h_len), hipMemcpyHostToDevice, ctx->stream));
    if (m->nrow > 0)
    {
        const int grid = (int)std::min<int64_t>(kMaxGrid, ceil_div(m->nrow, kBlock));
        hipLaunchKernelGGL(row_len_max_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, m->nrow, m->a, d_max);
        SPMV_HIP(hipGetLastError());
    }
    SPMV_HIP(hipMemcpyAsync(h_len, d_max, sizeof(h_len), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    m->max_row_nnz = h_len[0];
    m->min_row_nnz = m->nrow > 0 ? h_len[1] : 0;

    const double mean = m->nrow > 0 ? (double)m->nnz / (double)m->nrow : 0.0;
    m->lanes_per_row  = pick_lanes(mean);

    // column windows per block of kWinRows rows
    m->win_rows = kWinRows;
    if (m->nrow > 0 && m->nnz > 0)
    {
        const int nblocks = (int)ceil_div(m->nrow, kWinRows);
        SPMV_HIP(hipMalloc(&m->win_lo, sizeof(int32_t) * nblocks));
        SPMV_HIP(hipMalloc(&m->win_span, sizeof(int32_t) * nblocks));
        m->device_bytes += 2 * (int64_t)sizeof(int32_t) * nblocks;
        unsigned long long* d_contig = (unsigned long long*)((char*)ctx->scratch + 32);  // (scratch holds >= 64 bytes)
        unsigned long long  h_contig = 0;
        SPMV_HIP(hipMemsetAsync(d_contig, 0, sizeof(*d_contig), ctx->stream));
        hipLaunchKernelGGL(csr_window_scan_kernel, dim3(nblocks), dim3(kBlock), 0, ctx->stream, m->nrow, kWinRows,
                           m->a, m->b, m->win_lo, m->win_span, d_contig);
        SPMV_HIP(hipGetLastError());
        SPMV_HIP(hipMemcpyAsync(&h_contig, d_contig, sizeof(h_contig), hipMemcpyDeviceToHost, ctx->stream));
        SPMV_HIP(hipStreamSynchronize(ctx->stream));
        m->contig_frac = (double)h_contig / (double)m->nnz;
        SPMV_TRY(reduce_max_i32(ctx, m->win_span, nblocks, &m->win_max_span));
        unsigned long long* d_sum = (unsigned long long*)ctx->scratch;
        unsigned long long  total = 0;
        SPMV_HIP(hipMemsetAsync(d_sum, 0, sizeof(*d_sum), ctx->stream));
        hipLaunchKernelGGL(sum_i32_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(nblocks, kBlock))),
                           dim3(kBlock), 0, ctx->stream, m->win_span, nblocks, d_sum);
        SPMV_HIP(hipMemcpyAsync(&total, d_sum, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
        SPMV_HIP(hipStreamSynchronize(ctx->stream));
        m->win_avg_span = (double)total / nblocks;
    }
    // (a handle created under spmv_ctx_set_plan takes the context's plan; copies get their node handed down by whoever builds them)
    const bool from_ctx = plan_take_armed(m);
    if (plan_of(m))
    {
        const int rc = csr_apply_plan(m);  // the plan's kernel and layout, no timing launch (select.hip)
        plan_clear(m);
        if (rc == SPMV_OK || !from_ctx) return rc;
        (void)hipGetLastError();  // a context's plan that does not fit THIS matrix is no reason to refuse the handle: it selects by itself
        plan_reset_requests(m);
    }
    if (!m->kernel_forced) return csr_select_kernel(m);  // the model, and where it pays a trial of the candidates (select.hip)
    if (m->kernel == SPMV_CSR_PANEL) SPMV_TRY(csr_panel_build(m));
    if (m->kernel == SPMV_CSR_TWOPHASE) SPMV_TRY(csr_twophase_build(m));
    if (m->kernel == SPMV_CSR_SEGSCAN) SPMV_TRY(csr_segscan_build(m));
    if (m->kernel == SPMV_CSR_SPLIT) SPMV_TRY(csr_split_build(m));
    if (m->kernel == SPMV_CSR_ELL) SPMV_TRY(csr_ell_copy_build(m));
    return SPMV_OK;
}

// the row-parallel kernel with the solver's extras fused into its write-back; false if another kernel is selected
bool csr_vector_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex, int* rc)
{
    if (A->nrow == 0 || (A->kernel != SPMV_CSR_VECTOR && A->kernel != SPMV_CSR_AUTO)) return false;
    const int  lanes = A->lanes_per_row > 0 ? A->lanes_per_row : 8;
    const bool dpp = A->flags & SPMV_FLAG_DPP_REDUCE, remap = A->flags & SPMV_FLAG_XCD_REMAP;
    if (dpp && remap)
        *rc = launch_vector<true, true>(ctx, A, x, y, lanes, ex);
    else if (dpp)
        *rc = launch_vector<true, false>(ctx, A, x, y, lanes, ex);
    else if (remap)
        *rc = launch_vector<false, true>(ctx, A, x, y, lanes, ex);
    else
        *rc = launch_vector<false, false>(ctx, A, x, y, lanes, ex);
    return true;
}

int csr_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nrow == 0) return SPMV_OK;
    const int lanes = A->lanes_per_row > 0 ? A->lanes_per_row : 8;
    switch (A->kernel)
    {
        case SPMV_CSR_SCALAR:
            hipLaunchKernelGGL(csr_scalar_kernel, dim3((unsigned)ceil_div(A->nrow, kBlock)), dim3(kBlock), 0,
                               ctx->stream, A->nrow, A->a, A->b, A->v, x, y);
            SPMV_HIP(hipGetLastError());
            return SPMV_OK;
        case SPMV_CSR_LDSWIN:
            if (A->win_max_span <= 0 || A->win_max_span > kWinDoubles)
                SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "LDS-window kernel: widest block window is %d columns, tile holds %d",
                          A->win_max_span, kWinDoubles);
            return launch_ldswin(ctx, A, x, y, lanes);
        case SPMV_CSR_PANEL: return csr_panel_apply(ctx, A, x, y);
        case SPMV_CSR_TWOPHASE: return csr_twophase_apply_ex(ctx, A, x, y, apply_extra{});
        case SPMV_CSR_SEGSCAN: return csr_segscan_apply(ctx, A, x, y);
        case SPMV_CSR_SPLIT: return csr_split_apply(ctx, A, x, y);
        case SPMV_CSR_ELL:
            if (!A->ell_copy) SPMV_FAIL(SPMV_ERR_INVALID, "ELL copy selected but never built");
            return ell_apply(ctx, A->ell_copy, x, y);
        case SPMV_CSR_VECTOR:
        case SPMV_CSR_AUTO:
        default:
        {
            const bool dpp = A->flags & SPMV_FLAG_DPP_REDUCE, remap = A->flags & SPMV_FLAG_XCD_REMAP;
            if (dpp && remap) return launch_vector<true, true>(ctx, A, x, y, lanes);
            if (dpp) return launch_vector<true, false>(ctx, A, x, y, lanes);
            if (remap) return launch_vector<false, true>(ctx, A, x, y, lanes);
            return launch_vector<false, false>(ctx, A, x, y, lanes);
        }
    }
}
}  // namespace spmv
