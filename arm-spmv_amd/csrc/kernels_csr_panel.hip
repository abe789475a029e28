// kernels_csr_panel.hip — CSR `y += A*x` for matrices whose x does not fit L2 and whose columns have no
// locality (BASELINE config 2: N = 10M, 32 uniform-random columns per row).
//
// Why the plain row-parallel kernel is slow there (measured, profiles/r01_*): every 8-byte gather of x
// misses the 4 MiB L2 of its XCD and drags a 128-byte line across the fabric — 320M gathers move ~43 GB
// for 4.1 GB of algorithmic traffic, and the kernel runs at the fabric's ~7 TB/s, i.e. 8.6 % of the
// algorithmic roofline.  A table that fits L2 is gathered 4.6x faster (tools/probe_gather: 265 vs 58
// Ggather/s).
//
// Layout built here once per matrix (the reference does the same kind of work before its timed loop when it
// builds the per-node sub-matrices, src/mat_vec.cpp:240-268):
//   * rows are cut into GROUPS of G consecutive rows (G*8 bytes = the group's y accumulators fit the
//     160 KiB LDS of one CU);
//   * inside a group the entries are re-ordered by column PANEL (W columns, W*8 bytes << L2) and, inside a
//     panel, by 128-byte line of x — entries keep their value, their global column and a 16-bit row index
//     local to the group (14 bytes per entry instead of CSR's 12).
// Kernel: one 1024-thread workgroup per CU walks its group's entries front to back, so at any moment all
// 256 workgroups gather from the same few panels of x: the lines are fetched from HBM/MALL once per XCD and
// round, and every other gather hits L2.  Products are added into the group's accumulators in LDS with
// ds_add_f64; at the end the accumulators are added to y with coalesced accesses.  Because consecutive
// entries of a panel are sorted by x line, lanes of one wavefront instruction often share a line, which cuts
// the number of L1<->L2 line transfers — the next bottleneck (one 128-byte line per ~2 clocks per CU).
//
// Algorithmic bytes are still counted with CSR's 12 bytes per entry, so the extra 2 bytes and
// the repeated x sweeps show up as a lower roofline fraction, not as hidden traffic.
#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kPanelThreads = 1024;  // 16 wavefronts: one workgroup per CU (LDS-limited)
constexpr int kLineDoubles  = 16;    // 128-byte line of x

// ---- build step 1: entries per (group, panel) + exclusive scan inside the group -----------------------------
__global__ __launch_bounds__(256) void panel_count_kernel(int nrow, int G, int W, int P,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col,
                                                          int32_t* __restrict__ tile_ptr /* [ngroups][P+1] */)
{
    extern __shared__ int32_t hist[];  // P + 1
    const int g  = blockIdx.x;
    const int r0 = g * G, r1 = min(nrow, r0 + G);
    for (int i = threadIdx.x; i <= P; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int begin = row_ptr[r0], end = row_ptr[r1];
    for (int e = begin + threadIdx.x; e < end; e += blockDim.x) atomicAdd(&hist[col[e] / W], 1);
    __syncthreads();
    // exclusive scan of hist[0..P) by wavefront 0, 64 bins at a time
    if (threadIdx.x < kWave)
    {
        const int lane  = threadIdx.x;
        int       carry = 0;
        for (int base = 0; base < P; base += kWave)
        {
            const int v    = (base + lane < P) ? hist[base + lane] : 0;
            int       incl = v;
            for (int d = 1; d < kWave; d <<= 1)
            {
                const int up = bpermute(incl, max(lane - d, 0));
                if (lane >= d) incl += up;
            }
            if (base + lane < P) tile_ptr[(size_t)g * (P + 1) + base + lane] = carry + incl - v;
            carry += bpermute(incl, kWave - 1);
        }
        if (lane == 0) tile_ptr[(size_t)g * (P + 1) + P] = carry;
    }
}

// ---- build step 2: scatter the group's entries into their panel (order inside a panel: any) ----------------
__global__ __launch_bounds__(256) void panel_scatter_kernel(int nrow, int G, int W, int P,
                                                            const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col,
                                                            const double* __restrict__ val,
                                                            const int32_t* __restrict__ tile_ptr,
                                                            int32_t* __restrict__ out_col, uint16_t* __restrict__ out_row,
                                                            double* __restrict__ out_val)
{
    extern __shared__ int32_t cursor[];  // P
    constexpr int LPR = 8;
    const int     g   = blockIdx.x;
    const int     r0 = g * G, r1 = min(nrow, r0 + G);
    for (int i = threadIdx.x; i < P; i += blockDim.x) cursor[i] = tile_ptr[(size_t)g * (P + 1) + i];
    __syncthreads();
    const int base = row_ptr[r0];
    for (int r = r0 + threadIdx.x / LPR; r < r1; r += blockDim.x / LPR)
    {
        const int end = row_ptr[r + 1];
        for (int j = row_ptr[r] + threadIdx.x % LPR; j < end; j += LPR)
        {
            const int c   = col[j];
            const int pos = base + atomicAdd(&cursor[c / W], 1);
            out_col[pos]  = c;
            out_row[pos]  = (uint16_t)(r - r0);
            out_val[pos]  = val[j];
        }
    }
}

// ---- build step 3: inside each (group, panel) tile, bucket the entries by 128-byte line of x --------------
// counting sort with one bin per line of the panel (W/16 bins in LDS); src -> dst
__global__ __launch_bounds__(256) void panel_line_sort_kernel(int G, int W, int P, int nrow,
                                                              const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ tile_ptr,
                                                              const int32_t* __restrict__ src_col,
                                                              const uint16_t* __restrict__ src_row,
                                                              const double* __restrict__ src_val,
                                                              int32_t* __restrict__ dst_col, uint16_t* __restrict__ dst_row,
                                                              double* __restrict__ dst_val)
{
    extern __shared__ int32_t bins[];  // W/16 + 1
    const int nb   = W / kLineDoubles;
    const int g    = blockIdx.x / P;
    const int p    = blockIdx.x % P;
    const int base = row_ptr[min(nrow, g * G)];
    const int t0   = base + tile_ptr[(size_t)g * (P + 1) + p];
    const int t1   = base + tile_ptr[(size_t)g * (P + 1) + p + 1];
    if (t0 == t1) return;
    const int c0 = p * W;
    for (int i = threadIdx.x; i <= nb; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    for (int e = t0 + threadIdx.x; e < t1; e += blockDim.x) atomicAdd(&bins[(src_col[e] - c0) / kLineDoubles], 1);
    __syncthreads();
    // exclusive scan of bins[0..nb) by wavefront 0
    if (threadIdx.x < kWave)
    {
        const int lane  = threadIdx.x;
        int       carry = 0;
        for (int b = 0; b < nb; b += kWave)
        {
            const int v    = (b + lane < nb) ? bins[b + lane] : 0;
            int       incl = v;
            for (int d = 1; d < kWave; d <<= 1)
            {
                const int up = bpermute(incl, max(lane - d, 0));
                if (lane >= d) incl += up;
            }
            if (b + lane < nb) bins[b + lane] = carry + incl - v;
            carry += bpermute(incl, kWave - 1);
        }
    }
    __syncthreads();
    for (int e = t0 + threadIdx.x; e < t1; e += blockDim.x)
    {
        const int c   = src_col[e];
        const int pos = t0 + atomicAdd(&bins[(c - c0) / kLineDoubles], 1);
        dst_col[pos]  = c;
        dst_row[pos]  = src_row[e];
        dst_val[pos]  = src_val[e];
    }
}

// ---- the product --------------------------------------------------------------------------------------------
// UNROLL entries per lane are in flight between the streamed loads and the LDS adds.  With PIPE the streamed
// loads of the NEXT batch are issued before the gathers of the current one, so the HBM latency of the
// stream and the L2 latency of the gathers overlap instead of adding up (one workgroup per CU = 16
// wavefronts is all the thread-level parallelism the LDS footprint allows).
template <int UNROLL>
struct PanelBatch
{
    int      c[UNROLL];
    unsigned r[UNROLL];
    double   v[UNROLL];
    __device__ __forceinline__ void load(const int32_t* __restrict__ pcol, const uint16_t* __restrict__ prow,
                                         const double* __restrict__ pval, int e)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            c[u] = load_stream(pcol + e + u * kPanelThreads);
            r[u] = load_stream(prow + e + u * kPanelThreads);
            v[u] = load_stream(pval + e + u * kPanelThreads);
        }
    }
    __device__ __forceinline__ void apply(const double* __restrict__ x, double* acc) const
    {
        double xv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) atomicAdd(&acc[r[u]], v[u] * xv[u]);  // ds_add_f64
    }
};

template <int UNROLL, bool PIPE>
__global__ __launch_bounds__(kPanelThreads) void csr_panel_kernel(int nrow, int G, int ngroups,
                                                                  const int32_t* __restrict__ row_ptr,
                                                                  const int32_t* __restrict__ pcol,
                                                                  const uint16_t* __restrict__ prow,
                                                                  const double* __restrict__ pval,
                                                                  const double* __restrict__ x, double* __restrict__ y)
{
    extern __shared__ double acc[];  // G accumulators
    constexpr int STEP = UNROLL * kPanelThreads;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int r0   = g * G;
        const int rows = min(G, nrow - r0);
        for (int i = threadIdx.x; i < rows; i += kPanelThreads) acc[i] = 0.0;
        __syncthreads();
        const int begin = row_ptr[r0], end = row_ptr[r0 + rows];
        const int nfull = (end - begin) / STEP;  // batches in which every lane has UNROLL valid entries
        int       e     = begin + threadIdx.x;
        if constexpr (PIPE)
        {
            PanelBatch<UNROLL> cur, nxt;
            if (nfull > 0) cur.load(pcol, prow, pval, e);
            for (int b = 0; b < nfull; ++b)
            {
                if (b + 1 < nfull) nxt.load(pcol, prow, pval, e + STEP);
                cur.apply(x, acc);
                cur = nxt;
                e += STEP;
            }
        }
        else
        {
            for (int b = 0; b < nfull; ++b)
            {
                PanelBatch<UNROLL> cur;
                cur.load(pcol, prow, pval, e);
                cur.apply(x, acc);
                e += STEP;
            }
        }
        for (; e < end; e += kPanelThreads)
            atomicAdd(&acc[load_stream(prow + e)], load_stream(pval + e) * x[load_stream(pcol + e)]);
        __syncthreads();
        for (int i = threadIdx.x; i < rows; i += kPanelThreads) y[r0 + i] += acc[i];
        __syncthreads();
    }
}
}  // namespace

void csr_panel_free(spmv_mat* m)
{
    if (m->pb_col) hipFree(m->pb_col);
    if (m->pb_row) hipFree(m->pb_row);
    if (m->pb_val) hipFree(m->pb_val);
    m->pb_col = nullptr;
    m->pb_row = nullptr;
    m->pb_val = nullptr;
    m->pb_built_sort = -1;
    m->device_bytes -= m->pb_bytes;
    m->pb_bytes = 0;
}

// Choose G so that the groups fill whole rounds of 256 workgroups: G = ceil(nrow / (256*m)) for the smallest
// m that keeps G*8 bytes inside the LDS budget.
static int pick_group_rows(int nrow, int cap)
{
    for (int m = 1;; ++m)
    {
        const int64_t g = ceil_div(nrow, (int64_t)kNumCu * m);
        if (g <= cap) return (int)std::max<int64_t>(g, 1);
    }
}

int csr_panel_build(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (m->nrow == 0 || m->nnz == 0) return SPMV_OK;
    constexpr int kCapRows = 20000;  // 160,000 B of the CU's 163,840 B LDS
    int G = m->pb_group_rows > 0 ? std::min(m->pb_group_rows, kCapRows) : pick_group_rows(m->nrow, kCapRows);
    int W = m->pb_panel_width > 0 ? m->pb_panel_width : 128 * 1024;
    W     = std::max(kLineDoubles, (W / kLineDoubles) * kLineDoubles);
    while (ceil_div(m->ncol, W) > 8192) W *= 2;  // the per-group histogram lives in LDS
    const bool sort = m->pb_sort != 0;
    if (m->pb_col && m->pb_built_rows == G && m->pb_built_width == W && m->pb_built_sort == (int)sort)
        return SPMV_OK;  // the layout in memory was built with these parameters
    csr_panel_free(m);
    const int ngroups = (int)ceil_div(m->nrow, G);
    const int P       = (int)ceil_div(m->ncol, W);
    const size_t nnz  = (size_t)m->nnz;

    int32_t*  tile_ptr = nullptr;
    int32_t*  tcol     = nullptr;
    uint16_t* trow     = nullptr;
    double*   tval     = nullptr;
    int       rc       = SPMV_OK;
    hipStream_t s      = ctx->stream;
    do
    {
        if (hipMalloc(&m->pb_col, nnz * sizeof(int32_t)) != hipSuccess || hipMalloc(&m->pb_row, nnz * sizeof(uint16_t)) != hipSuccess ||
            hipMalloc(&m->pb_val, nnz * sizeof(double)) != hipSuccess ||
            hipMalloc(&tile_ptr, sizeof(int32_t) * (size_t)ngroups * (P + 1)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (sort && (hipMalloc(&tcol, nnz * sizeof(int32_t)) != hipSuccess || hipMalloc(&trow, nnz * sizeof(uint16_t)) != hipSuccess ||
                     hipMalloc(&tval, nnz * sizeof(double)) != hipSuccess))
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        hipLaunchKernelGGL(panel_count_kernel, dim3(ngroups), dim3(256), sizeof(int32_t) * (P + 1), s, m->nrow, G, W, P,
                           m->a, m->b, tile_ptr);
        int32_t*  scol = sort ? tcol : m->pb_col;
        uint16_t* srow = sort ? trow : m->pb_row;
        double*   sval = sort ? tval : m->pb_val;
        hipLaunchKernelGGL(panel_scatter_kernel, dim3(ngroups), dim3(256), sizeof(int32_t) * P, s, m->nrow, G, W, P, m->a,
                           m->b, m->v, tile_ptr, scol, srow, sval);
        if (sort)
            hipLaunchKernelGGL(panel_line_sort_kernel, dim3((unsigned)ngroups * P), dim3(256),
                               sizeof(int32_t) * (W / kLineDoubles + 1), s, G, W, P, m->nrow, m->a, tile_ptr, tcol, trow,
                               tval, m->pb_col, m->pb_row, m->pb_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    if (tile_ptr) hipFree(tile_ptr);
    if (tcol) hipFree(tcol);
    if (trow) hipFree(trow);
    if (tval) hipFree(tval);
    if (rc != SPMV_OK)
    {
        csr_panel_free(m);
        SPMV_FAIL(rc, "building the panel layout (%d groups of %d rows, %d panels of %d columns) failed: %s", ngroups, G, P,
                  W, hipGetErrorString(hipGetLastError()));
    }
    m->pb_built_rows  = G;
    m->pb_built_width = W;
    m->pb_built_sort  = (int)sort;
    m->pb_ngroups     = ngroups;
    m->pb_bytes       = (int64_t)(nnz * 14);
    m->device_bytes += m->pb_bytes;
    return SPMV_OK;
}

int csr_panel_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (!A->pb_col) SPMV_FAIL(SPMV_ERR_INVALID, "panel kernel selected but its layout was not built");
    const int    G   = A->pb_built_rows;
    const size_t lds = (size_t)G * sizeof(double);
    // two workgroups share a CU when their accumulators fit twice into the 160 KiB LDS
    const int per_cu = (lds <= 80000 && A->pb_two_per_cu) ? 2 : 1;
    const int grid   = std::min(A->pb_ngroups, kNumCu * per_cu);
    const int unroll = A->pb_unroll > 0 ? A->pb_unroll : 8;
    const bool pipe  = A->pb_pipe != 0;
#define SPMV_PANEL_CASE(U, P)                                                                                        \
    if (unroll == U && pipe == P)                                                                                    \
    {                                                                                                                \
        static bool granted = false;                                                                                 \
        if (!granted)                                                                                                \
        {                                                                                                            \
            SPMV_HIP(hipFuncSetAttribute((const void*)csr_panel_kernel<U, P>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         160000));                                                                   \
            granted = true;                                                                                          \
        }                                                                                                            \
        hipLaunchKernelGGL((csr_panel_kernel<U, P>), dim3(grid), dim3(kPanelThreads), lds, ctx->stream, A->nrow, G,    \
                           A->pb_ngroups, A->a, A->pb_col, A->pb_row, A->pb_val, x, y);                               \
        SPMV_HIP(hipGetLastError());                                                                                 \
        return SPMV_OK;                                                                                              \
    }
    SPMV_PANEL_CASE(2, false)
    SPMV_PANEL_CASE(4, false)
    SPMV_PANEL_CASE(8, false)
    SPMV_PANEL_CASE(16, false)
    SPMV_PANEL_CASE(2, true)
    SPMV_PANEL_CASE(4, true)
    SPMV_PANEL_CASE(8, true)
#undef SPMV_PANEL_CASE
    SPMV_FAIL(SPMV_ERR_INVALID, "panel kernel: unroll=%d pipe=%d is not instantiated", unroll, (int)pipe);
}
}  // namespace spmv
