// kernels_coo.hip — y += A*x for COO on CDNA4 (gfx950): wavefront segmented scan.
//
// Replaces COOMatirxMatVector (reference src/mat_vec.cpp:18-42), which issues one `omp atomic` per
// nonzero.  Here a wavefront owns a contiguous chunk of kIters*64 entries; iteration t covers entries
// base + t*64 + lane (coalesced 4/4/8-byte loads).  Products are combined with a 64-lane segmented
// inclusive scan keyed on "same row as my left neighbour" (ds_bpermute shifts), so each maximal run of
// equal row indices produces ONE update of y, issued by the run's last lane.  The run that is still open
// at lane 63 is carried into the next iteration in registers.
//
// Which updates need a hardware fp64 atomic (global_atomic_add_f64):
//   - row-sorted input (detected once, coo_analyse): only runs that touch the first or the last entry of
//     the wavefront's chunk can share their row with another wavefront — they use the atomic; every other
//     run owns its row and does a plain read-modify-write.  That is <= 2 atomics per kIters*64 entries.
//   - unsorted input (legal in the reference: file order, duplicates allowed, include/matrix.h:7-25):
//     a row may reappear anywhere, so every run end uses the atomic.  Still correct, just slower.
//
// Roofline: HBM-bound; algorithmic bytes per application = 16*nnz + 8*ncol + 16*nrow.
#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kIters      = 8;                     // iterations of 64 entries per wavefront
constexpr int kWaveChunk  = kIters * kWave;        // 512 entries
constexpr int kBlockChunk = kWaveChunk * (kBlock / kWave);  // 2048 entries per workgroup
constexpr int kNoRow      = INT32_MAX;             // padding lanes beyond nnz

__device__ __forceinline__ void add_to_y(double* y, int row, double v, bool atomic)
{
    if (atomic)
        unsafeAtomicAdd(y + row, v);
    else
        y[row] += v;
}

template <bool SORTED>
__global__ __launch_bounds__(kBlock) void coo_segscan_kernel(int64_t nnz, const int32_t* __restrict__ row,
                                                             const int32_t* __restrict__ col,
                                                             const double* __restrict__ val,
                                                             const double* __restrict__ x, double* __restrict__ y)
{
    const int     lane = lane_id();
    const int64_t base = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWaveChunk;
    if (base >= nnz) return;  // wave-uniform

    // issue every load of the chunk before the first scan
    int    r[kIters];
    double p[kIters];
    {
        int    c[kIters];
        double v[kIters];
#pragma unroll
        for (int t = 0; t < kIters; ++t)
        {
            const int64_t e  = base + t * kWave + lane;
            const bool    ok = e < nnz;
            r[t]             = ok ? load_stream(row + e) : kNoRow;
            c[t]             = ok ? load_stream(col + e) : 0;
            v[t]             = ok ? load_stream(val + e) : 0.0;
        }
#pragma unroll
        for (int t = 0; t < kIters; ++t) p[t] = (r[t] != kNoRow) ? v[t] * x[c[t]] : 0.0;
    }

    int    carry_row  = kNoRow;  // run left open at lane 63 of the previous iteration (wave-uniform)
    double carry_val  = 0.0;
    bool   chain_open = true;    // the open run reaches back to the chunk's first entry

#pragma unroll
    for (int t = 0; t < kIters; ++t)
    {
        const int rr = r[t];
        double    pp = p[t];
        // lane 0: continue the carried run, or close it
        bool head0 = false;
        if (lane == 0 && carry_row != kNoRow)
        {
            if (rr == carry_row)
                pp += carry_val;
            else
            {
                add_to_y(y, carry_row, carry_val, !SORTED || chain_open);
                head0 = true;
            }
        }
        const int left = bpermute(rr, max(lane - 1, 0));
        int       head = (lane == 0) ? (int)head0 : (int)(left != rr);
        // segmented inclusive scan (Hillis-Steele with head-flag OR)
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1)
        {
            const int    src = max(lane - d, 0);
            const double pu  = bpermute(pp, src);
            const int    hu  = bpermute(head, src);
            if (lane >= d)
            {
                if (!head) pp += pu;
                head |= hu;
            }
        }
        // run ends where the right neighbour has another row; lane 63 stays open (carried)
        const int right = bpermute(rr, min(lane + 1, kWave - 1));
        if (lane < kWave - 1 && right != rr && rr != kNoRow)
            add_to_y(y, rr, pp, !SORTED || (chain_open && !head));
        // wave-uniform carry-out
        carry_row = bpermute(rr, kWave - 1);
        carry_val = bpermute(pp, kWave - 1);
        chain_open = chain_open && (bpermute(head, kWave - 1) == 0);
    }
    // the run that contains the chunk's last entry may continue in the next wavefront's chunk
    if (lane == 0 && carry_row != kNoRow) add_to_y(y, carry_row, carry_val, true);
}

__global__ __launch_bounds__(kBlock) void coo_sorted_check_kernel(int64_t nnz, const int32_t* __restrict__ row,
                                                                  int32_t* __restrict__ unsorted_flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i + 1 < nnz; i += (int64_t)gridDim.x * kBlock)
        bad |= row[i] > row[i + 1];
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(unsorted_flag, 1);
}
}  // namespace

int coo_analyse(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    SPMV_TRY(ensure_scratch(ctx, 64));
    int32_t* flag = (int32_t*)ctx->scratch;
    SPMV_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
    if (m->nnz > 1)
    {
        const int grid = (int)std::min<int64_t>(kMaxGrid, ceil_div(m->nnz, kBlock));
        hipLaunchKernelGGL(coo_sorted_check_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, m->nnz, m->a, flag);
        SPMV_HIP(hipGetLastError());
    }
    int32_t unsorted = 0;
    SPMV_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));
    m->sorted_rows = unsorted ? 0 : 1;
    m->kernel      = SPMV_CSR_VECTOR;  // reported for COO as "segmented scan"
    if (!m->kernel_forced) SPMV_TRY(coo_build_panel(m, /*only_if_worth=*/true));
    return SPMV_OK;
}

// Large COO with an x beyond L2 is gather-bound in entry order exactly like CSR (C4: 13 % of roofline with the
// segmented scan).  Such a handle gets the panel layout too: the entries are grouped by row on the device
// (spmv_coo_to_csr, duplicates and file order inside a row kept) and re-ordered as in kernels_csr_panel.hip; the
// product then runs csr_panel_kernel (C4: 0.51 ms instead of 1.81).  Only row_ptr and the panel arrays are kept.
int coo_build_panel(spmv_mat* m, bool only_if_worth)
{
    if (m->coo_csr) return SPMV_OK;
    const bool worth = m->nnz >= ((int64_t)2 << 20) && m->nrow > 0 && m->nnz / m->nrow >= 2;
    if (only_if_worth && !worth) return SPMV_OK;
    if (m->nnz == 0 || m->nnz > (int64_t)INT32_MAX - 65536) return SPMV_OK;
    spmv_mat* csr = nullptr;
    SPMV_TRY(coo_to_csr(m->ctx, m, &csr));  // csr_analyse inside picks (and builds) the panel layout when x is large
    if (csr->kernel != SPMV_CSR_PANEL)
    {
        csr_twophase_free(csr);  // csr_analyse may have chosen (and built) the two-phase layout: not the one that runs here
        csr->kernel_forced = true;
        csr->kernel        = SPMV_CSR_PANEL;
        int rc             = csr_panel_build(csr);
        if (rc != SPMV_OK)
        {
            mat_free(csr);
            return rc;
        }
    }
    // the panel kernel reads row_ptr and its own arrays only: drop the CSR copies of col_ind / values
    (void)hipFree(const_cast<int32_t*>(csr->b));
    (void)hipFree(const_cast<double*>(csr->v));
    csr->device_bytes -= (int64_t)csr->nnz * 12;
    csr->b     = nullptr;
    csr->v     = nullptr;
    m->coo_csr = csr;
    m->kernel  = SPMV_CSR_PANEL;
    m->device_bytes += csr->device_bytes;
    return SPMV_OK;
}

int coo_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nnz == 0) return SPMV_OK;
    if (A->coo_csr && A->kernel == SPMV_CSR_PANEL) return csr_panel_apply(ctx, A->coo_csr, x, y);
    const unsigned grid = (unsigned)ceil_div(A->nnz, kBlockChunk);
    if (A->sorted_rows)
        hipLaunchKernelGGL(coo_segscan_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, A->nnz, A->a, A->b, A->v,
                           x, y);
    else
        hipLaunchKernelGGL(coo_segscan_kernel<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, A->nnz, A->a, A->b,
                           A->v, x, y);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
