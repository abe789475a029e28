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
constexpr int kIters      = 8;                     // iterations of 64 entries per wavefront (a power of two: coo_bin_place_kernel)
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

// one wavefront's chunk: entries base .. base + kWaveChunk - 1 (those at or beyond nnz are padding)
template <bool SORTED>
__device__ __forceinline__ void segscan_chunk(int64_t base, int64_t nnz, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                              const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y)
{
    const int lane = lane_id();

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

template <bool SORTED>
__global__ __launch_bounds__(kBlock) void coo_segscan_kernel(int64_t nnz, const int32_t* __restrict__ row,
                                                             const int32_t* __restrict__ col,
                                                             const double* __restrict__ val,
                                                             const double* __restrict__ x, double* __restrict__ y)
{
    const int64_t base = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWaveChunk;
    if (base >= nnz) return;  // wave-uniform
    segscan_chunk<SORTED>(base, nnz, row, col, val, x, y);
}

// The scan over the copy in column bins (coo_build_bins).  Workgroups go to the XCDs round-robin (workgroup w runs on
// XCD w % 8), each XCD has an L2 of its own, and an x beyond 4 MB gathered in entry order misses it with every entry:
// C4 moved 12.9 GB for the 1.89 GB of its entries (128-byte lines from the fabric, 1.81 ms).  Here XCD c scans only the
// entries whose columns lie in ITS bins, one bin after the other, so the slice of x it gathers from (<= 2 MB) stays in
// its L2 (counters: 1.98 GB).  A row is spread over up to `bins` runs now: every run end is an atomic on y.
//
// With the gathers served the scan itself was the bound (0.92 ms; 1.16 ms in place with an x of 0.8 MB): six ds_bpermute
// steps on a double and a flag for every 64 entries.  The copy is the engine's own, so it is laid out for a cheaper
// scan: inside a wavefront's chunk of 512 entries, entry q of the order sits at (q % 8) * 64 + q / 8 - the loads are the
// same coalesced ones, and lane l holds the 8 CONSECUTIVE entries 8 l .. 8 l + 7.  A lane reduces its strip serially
// (runs that begin and end inside it go to y from there), and one 64-lane segmented scan per 512 entries joins the runs
// that cross lanes: lane l contributes the run open at its end (row rb, sum b); the run open at its start (row ra, sum a,
// the same run when the strip is a single one) is closed by the lane in which it ends.
struct coo_bins_tab
{
    int32_t region[8];
    int32_t chunks[8];
};
// What bounds it now (tools/probe_coo_bins.py, C4 over 16 bins): with every update of y left out the kernel takes 0.65 ms -
// 115M gathers of 8 bytes from an L2-resident slice are 115M L2 line operations beside the 15M lines of the streams, the
// same line-rate bound as the CSR panel product, and entries in row order cannot be sorted by x line.  The
// updates add 0.10 (8 bins) to 0.23 ms (16): they are device-scope atomics carried out by the memory side, one per run.
// Tried instead, for row-sorted input with one bin per XCD (a run interior to a chunk is then all of its row in the bin):
// plain stores to a partial vector per XCD plus a kernel that adds the eight to y - 1.29 + 0.05 ms against 0.75 (the
// 8-byte stores allocate in the L2 the slice of x lives in; nontemporal ones: 0.97 against 0.88 over 16 bins).  Not kept.
__global__ __launch_bounds__(kBlock) void coo_segscan_bins_kernel(const coo_bins_tab tab, const int32_t* __restrict__ row,
                                                                  const int32_t* __restrict__ col, const double* __restrict__ val,
                                                                  const double* __restrict__ x, double* __restrict__ y)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= tab.chunks[xcd]) return;
    const int     lane = lane_id();
    const int64_t base = ((int64_t)(tab.region[xcd] + slot) * (kBlock / kWave) + (threadIdx.x >> 6)) * kWaveChunk;
    int           r[kIters];
    double        p[kIters];
    {
        int    c[kIters];
        double v[kIters];
#pragma unroll
        for (int t = 0; t < kIters; ++t)
        {
            const int64_t e = base + t * kWave + lane;  // (the copy is padded to whole chunks: no bound to check)
            r[t]            = load_stream(row + e);
            c[t]            = load_stream(col + e);
            v[t]            = load_stream(val + e);
        }
#pragma unroll
        for (int t = 0; t < kIters; ++t) p[t] = v[t] * x[c[t]];  // (padding: 0.0 * x[0], never added to y)
    }
    // the lane's strip, serially
    const int ra     = r[0];
    int       rb     = r[0];
    double    a      = 0.0, b = p[0];
    bool      single = true;
#pragma unroll
    for (int t = 1; t < kIters; ++t)
    {
        if (r[t] != rb)
        {
            if (single)
            {
                a      = b;
                single = false;
            }
            else if (rb != kNoRow)
                unsafeAtomicAdd(y + rb, b);  // begins and ends inside the strip
            rb = r[t];
            b  = p[t];
        }
        else
            b += p[t];
    }
    // across the lanes
    const int  prev_rb = bpermute(rb, max(lane - 1, 0));
    const bool cont    = lane > 0 && ra == prev_rb;   // the run open at my start began before me
    int        head    = (!single || !cont) ? 1 : 0;  // the run open at my end begins in my strip
    double     sum     = b;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1)
    {
        const int    src = max(lane - d, 0);
        const double su  = bpermute(sum, src);
        const int    hu  = bpermute(head, src);
        if (lane >= d)
        {
            if (!head) sum += su;
            head |= hu;
        }
    }
    const double sum_prev = bpermute(sum, max(lane - 1, 0));
    const int    next_ra  = bpermute(ra, min(lane + 1, kWave - 1));
    if (!single && ra != kNoRow) unsafeAtomicAdd(y + ra, a + (cont ? sum_prev : 0.0));           // the run open at my start ends in my strip
    if ((lane == kWave - 1 || next_ra != rb) && rb != kNoRow) unsafeAtomicAdd(y + rb, sum);  // the run open at my end ends with it
}

constexpr int kColBuckets = 4096;  // the bins are unions of column buckets of equal width
// (a column outside 0 .. ncol-1 is the caller's error and the product's problem, as in the reference; the build stays inside its tables)
__device__ __forceinline__ int bucket_of(int c, int shift) { return min(max(c >> shift, 0), kColBuckets - 1); }
__global__ __launch_bounds__(kBlock) void coo_col_histogram_kernel(int64_t nnz, const int32_t* __restrict__ col, int shift, unsigned long long* __restrict__ hist)
{
    __shared__ unsigned int h[kColBuckets];
    for (int i = threadIdx.x; i < kColBuckets; i += kBlock) h[i] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock) atomicAdd(&h[bucket_of(col[e], shift)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < kColBuckets; i += kBlock)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
__global__ __launch_bounds__(kBlock) void coo_bin_keys_kernel(int64_t nnz, const int32_t* __restrict__ col, int shift, const uint8_t* __restrict__ bin_of_bucket,
                                                              int32_t* __restrict__ key)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock) key[e] = bin_of_bucket[bucket_of(col[e], shift)];
}
__global__ __launch_bounds__(kBlock) void coo_bin_fill_kernel(int64_t padded, int32_t* __restrict__ row, int32_t* __restrict__ col, double* __restrict__ val)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < padded; i += (int64_t)gridDim.x * kBlock)
    {
        row[i] = kNoRow;
        col[i] = 0;
        val[i] = 0.0;
    }
}
// position i of the order by (bin, entry id) goes to the bin's copy, transposed inside each wavefront's chunk (see coo_segscan_bins_kernel)
__global__ __launch_bounds__(kBlock) void coo_bin_place_kernel(int64_t nnz, const int32_t* __restrict__ perm, const int32_t* __restrict__ row,
                                                               const int32_t* __restrict__ col, const double* __restrict__ val, int shift,
                                                               const uint8_t* __restrict__ bin_of_bucket, const int64_t* __restrict__ bin_src,
                                                               const int64_t* __restrict__ bin_dst, int32_t* __restrict__ out_row,
                                                               int32_t* __restrict__ out_col, double* __restrict__ out_val)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock)
    {
        const int32_t e = perm[i];
        const int32_t c = col[e];
        const int     b = bin_of_bucket[bucket_of(c, shift)];
        const int64_t q = i - bin_src[b];  // position in the bin's order; inside a chunk of 512, entry q sits at (q % 8) * 64 + q / 8
        const int64_t d = bin_dst[b] + (q & ~(int64_t)(kWaveChunk - 1)) + ((q & (kIters - 1)) << 6) + ((q & (kWaveChunk - 1)) >> 3);
        out_row[d]      = row[e];
        out_col[d]      = c;
        out_val[d]      = val[e];
    }
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
    const bool from_ctx = plan_take_armed(m);
    if (plan_of(m))
    {
        const int rc = coo_apply_plan(m);
        plan_clear(m);
        if (rc == SPMV_OK || !from_ctx) return rc;
        (void)hipGetLastError();  // (a context's plan that does not fit this matrix: the handle selects by itself)
    }
    if (!m->kernel_forced) SPMV_TRY(coo_select_kernel(m));
    return SPMV_OK;
}

void coo_drop_rowgrouped(spmv_mat* m)
{
    if (!m->coo_csr) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    m->device_bytes -= m->coo_csr->device_bytes;
    mat_free(m->coo_csr);
    m->coo_csr = nullptr;
    if (m->kernel == SPMV_CSR_PANEL) m->kernel = SPMV_CSR_VECTOR;
}

namespace
{
// adopts a CSR handle as the row-grouped copy; the layouts that do not read col_ind / values give them back
void adopt_rowgrouped(spmv_mat* m, spmv_mat* csr)
{
    if ((csr->kernel == SPMV_CSR_PANEL || csr->kernel == SPMV_CSR_TWOPHASE || csr->kernel == SPMV_CSR_ELL) && csr->b && csr->v && csr->owned)
    {
        (void)hipFree(const_cast<int32_t*>(csr->b));
        (void)hipFree(const_cast<double*>(csr->v));
        csr->device_bytes -= (int64_t)csr->nnz * 12;
        csr->b = nullptr;
        csr->v = nullptr;
    }
    m->coo_csr = csr;
    m->kernel  = SPMV_CSR_PANEL;  // reported for COO as "runs from the row-grouped copy" (whichever CSR kernel that copy picked)
    m->device_bytes += csr->device_bytes;
}
int coo_scan_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
}  // namespace

// AUTO for a COO handle: the segmented scan over the entries as they are, or a copy grouped by row (spmv_coo_to_csr: duplicates
// and the order inside a row kept) which picks ITS kernel like any CSR handle - row-parallel, LDS window, panel, two-phase
// (select.hip).  Model: the copy from 1.5M entries on (C4: 0.28 ms against 1.81 for the scan in place).  From 64K entries on
// both are timed (tools/sweep_structures.py: below 2M entries the copy won 2x on every family but one - an R-MAT graph
// with a hub row, where the scan won 1.7x; above, the scan won 6-15 % on dense blocks and wide rectangles).
int coo_select_kernel(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    select_reset(m);
    coo_drop_rowgrouped(m);
    (void)hipStreamSynchronize(ctx->stream);
    coo_free_bins(m);
    m->kernel = SPMV_CSR_VECTOR;
    if (m->nnz == 0 || m->nrow <= 0 || m->nnz > (int64_t)INT32_MAX - 65536) return SPMV_OK;
    const bool model_copy = m->nnz >= ((int64_t)3 << 19);
    auto       build_copy = [&]() -> int {
        spmv_mat* csr = nullptr;
        const int rc  = coo_to_csr(ctx, m, &csr, kCsrAutoNoSegscan);  // (csr_analyse inside selects the copy's kernel)
        if (rc == SPMV_OK) adopt_rowgrouped(m, csr);
        return rc;
    };
    if (!select_trials_enabled(m) || m->nnz < kSelectMinNnz) return model_copy ? build_copy() : SPMV_OK;
    select_scratch sv;
    if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return model_copy ? build_copy() : SPMV_OK;
    // the copy is built first, then the two are timed in rounds - the model's pick first - until their minima stand still
    // (select.hip: no allocation between two timings)
    int rc = build_copy();
    if (rc == SPMV_ERR_ALLOC && !model_copy)
    {
        (void)hipGetLastError();
        rc = SPMV_OK;  // no memory for the copy: the scan runs
    }
    if (rc != SPMV_OK) return rc;
    float t_scan = 1e30f, t_copy = 1e30f;
    {
        // candidate 0 is the model's pick; without a copy the scan is timed alone (its figure is reported)
        const bool copy_first = model_copy && m->coo_csr;
        float      t[2]       = {-1.f, -1.f};
        const int  n          = m->coo_csr ? 2 : 1;
        rc = select_rounds(ctx, n,
                           [&](int j) {
                               const bool copy = m->coo_csr && ((j == 0) == copy_first);
                               return copy ? csr_apply(ctx, m->coo_csr, sv.x, sv.y) : coo_scan_apply(ctx, m, sv.x, sv.y);
                           },
                           t, &m->sel_rounds);
        (void)hipStreamSynchronize(ctx->stream);
        if (rc != SPMV_OK) return rc;
        if (m->coo_csr)
        {
            t_copy = copy_first ? t[0] : t[1];
            t_scan = copy_first ? t[1] : t[0];
            if (t_copy >= 0.f) select_note(m, SPMV_CSR_PANEL, t_copy); else t_copy = 1e30f;
        }
        else
            t_scan = t[0];
        if (t_scan >= 0.f) select_note(m, SPMV_CSR_VECTOR, t_scan); else t_scan = 1e30f;
    }
    // Third candidate: the scan over a copy of the entries in column bins, one per XCD (coo_build_bins; 16 bytes per entry).
    // Worth a try where the scan in place is gather-bound but not hopeless (within 4x of the row-grouped copy; C4's is 6.5x behind, and its scan over bins loses too): local
    // columns under an x beyond an XCD's L2 - dense blocks, bands (8 x 8 blocks, 32M entries: 0.097 ms against the copy's 0.108).
    float t_bins = 1e30f;
    if (t_scan < 4.0f * t_copy && (int64_t)m->ncol * 8 > ((int64_t)3 << 20) && m->nnz >= ((int64_t)2 << 20))
    {
        rc = coo_build_bins(m, 0, /*only_if_worth=*/true);
        if (rc == SPMV_OK && m->cb_bins)
        {
            // the scan over the bins and (again, as a minimum to improve on) the copy, in rounds: the bins' first timing follows
            // their build - allocations and frees a moment ago - the later ones do not
            float     t[2] = {-1.f, m->coo_csr && t_copy < 1e29f ? t_copy : -1.f};
            const int n    = m->coo_csr ? 2 : 1;
            rc = select_rounds(ctx, n, [&](int j) { return j == 0 ? coo_scan_apply(ctx, m, sv.x, sv.y) : csr_apply(ctx, m->coo_csr, sv.x, sv.y); }, t, nullptr);
            if (rc == SPMV_OK)
            {
                t_bins = t[0] >= 0.f ? t[0] : 1e30f;
                if (t[0] >= 0.f) select_note(m, 6, t_bins);  // "select_us_variant1"
                if (m->coo_csr && t[1] >= 0.f)
                {
                    t_copy                    = t[1];
                    m->sel_us[SPMV_CSR_PANEL] = t_copy * 1000.f;
                }
            }
        }
        else if (rc == SPMV_ERR_ALLOC)
        {
            (void)hipGetLastError();
            rc = SPMV_OK;
        }
        (void)hipStreamSynchronize(ctx->stream);
        if (rc != SPMV_OK) return rc;
        if (!(t_bins < 0.98f * std::min(t_scan, t_copy))) coo_free_bins(m);
    }
    const float t_own     = std::min(t_scan, m->cb_bins ? t_bins : 1e30f);
    const bool  keep_copy = m->coo_csr && (model_copy ? t_copy <= t_own * 1.02f : t_copy < t_own * 0.98f);  // the second one has to win by 2 %
    if (!keep_copy)
        coo_drop_rowgrouped(m);
    else
        coo_free_bins(m);
    m->kernel = m->coo_csr ? SPMV_CSR_PANEL : SPMV_CSR_VECTOR;
    return SPMV_OK;
}

// A plan on a COO handle (plan.hip): the scan in place, the scan over column bins, or the row-grouped copy whose own node says
// which kernel and layout IT runs - no timing launch.
int coo_apply_plan(spmv_mat* m)
{
    const plan_node& p = *plan_of(m);
    spmv_ctx*        ctx = m->ctx;
    select_reset(m);
    coo_drop_rowgrouped(m);
    (void)hipStreamSynchronize(ctx->stream);
    coo_free_bins(m);
    m->kernel = SPMV_CSR_VECTOR;
    if (m->nnz == 0 || m->nrow <= 0) return SPMV_OK;
    if (p.kernel == SPMV_CSR_PANEL)
    {
        SPMV_REQUIRE(m->nnz <= (int64_t)INT32_MAX - 65536, "plan: a row-grouped copy of %lld COO entries does not fit int32 offsets", (long long)m->nnz);
        spmv_mat* csr = nullptr;
        SPMV_TRY(coo_to_csr(ctx, m, &csr, kCsrAutoNoSegscan));  // (hands the copy's node down)
        adopt_rowgrouped(m, csr);
    }
    else if (p.coo_bins_per_xcd > 0)
    {
        const int rc = coo_build_bins(m, p.coo_bins_per_xcd, /*only_if_worth=*/false);
        if (rc != SPMV_OK && rc != SPMV_ERR_ALLOC) return rc;  // (no room for the copy: the scan runs over the handle's own arrays)
        (void)hipGetLastError();
    }
    return SPMV_OK;
}

// The row-grouped copy with the PANEL kernel forced on it (spmv_mat_set_kernel(coo, SPMV_CSR_PANEL), and the CSC handles'
// regrouping, kernels_misc.hip).  Large COO with an x beyond L2 is gather-bound in entry order exactly like CSR (C4: 13 % of
// roofline with the segmented scan in place); the entries are grouped by row on the device (spmv_coo_to_csr, duplicates and
// file order inside a row kept) and re-ordered as in kernels_csr_panel.hip.  Only row_ptr and the panel arrays are kept.
int coo_build_panel(spmv_mat* m, bool only_if_worth)
{
    if (m->coo_csr && m->coo_csr->kernel == SPMV_CSR_PANEL)
    {
        m->kernel = SPMV_CSR_PANEL;
        return SPMV_OK;
    }
    const bool worth = m->nnz >= ((int64_t)3 << 19) && m->nrow > 0;
    if (only_if_worth && !worth) return SPMV_OK;
    if (m->nnz == 0 || m->nnz > (int64_t)INT32_MAX - 65536) return SPMV_OK;
    coo_drop_rowgrouped(m);  // (a copy AUTO made with another kernel)
    spmv_mat* csr = nullptr;
    SPMV_TRY(coo_to_csr(m->ctx, m, &csr, SPMV_CSR_PANEL));
    adopt_rowgrouped(m, csr);
    return SPMV_OK;
}

void coo_free_bins(spmv_mat* m)
{
    if (m->cb_row) (void)hipFree(m->cb_row);
    if (m->cb_col) (void)hipFree(m->cb_col);
    if (m->cb_val) (void)hipFree(m->cb_val);
    if (m->cb_bins) m->device_bytes -= m->cb_padded * 16;
    m->cb_row = m->cb_col = nullptr;
    m->cb_val = nullptr;
    m->cb_bins = 0;
    m->cb_padded = 0;
}

// The copy in column bins for the segmented scan.  The bins are chosen from a histogram of the columns so that each holds
// about the same number of ENTRIES (the XCDs finish together whatever the distribution of the columns); bin j belongs to
// XCD j / bins_per_xcd.  The order inside a bin is the order of the handle's entries (a stable sort by bin), so the runs of
// row-sorted input stay runs.  Integer work done once per handle; costs 16 bytes per entry of device memory.
int coo_build_bins(spmv_mat* m, int bins_per_xcd, bool only_if_worth)
{
    spmv_ctx* ctx = m->ctx;
    coo_free_bins(m);
    // worth it: an x that does not fit an XCD's L2 beside the streams, and enough entries to matter
    const bool worth = (int64_t)m->ncol * 8 > ((int64_t)3 << 20) && m->nnz >= ((int64_t)2 << 20);
    if (only_if_worth && !worth) return SPMV_OK;
    // one bin per XCD is worth its 16 bytes per entry only if workgroups w and w + 8 share an XCD on this device (the start-up
    // probe of the context: abi.hip xcd_probe).  Where they do not - a partitioned device, another dispatch order - the scan
    // runs over the handle's own arrays: same results, no copy.
    if (ctx->xcd_round_robin != 1)
    {
        static bool told = false;
        if (!told)
        {
            fprintf(stderr, "libspmv_hip: workgroups are not dealt round-robin over 8 XCDs on device %d (probe: %d, %d XCD ids seen): "
                            "the COO scan stays on the entries as stored (coo_column_bins = 0)\n", ctx->device, ctx->xcd_round_robin, ctx->xcds_seen);
            told = true;
        }
        if (only_if_worth) return SPMV_OK;
        SPMV_FAIL(SPMV_ERR_UNSUPPORTED, "coo_column_bins: workgroups w and w + 8 do not share an XCD on this device (spmv_ctx_xcd_round_robin)");
    }
    if (m->nnz == 0 || m->ncol <= 0) return SPMV_OK;
    // a handle too large for the copy's 32-bit addressing keeps the scan in place when nobody asked for bins by name
    if (only_if_worth && m->nnz > (int64_t)INT32_MAX - 64 * kBlockChunk) return SPMV_OK;
    SPMV_REQUIRE(m->nnz <= (int64_t)INT32_MAX - 64 * kBlockChunk, "the copy in column bins addresses its entries with 32 bits: %lld entries are too many", (long long)m->nnz);
    if (bins_per_xcd <= 0) bins_per_xcd = (int)std::min<int64_t>(8, std::max<int64_t>(1, ceil_div((int64_t)m->ncol * 8, (int64_t)16 << 20)));
    SPMV_REQUIRE(bins_per_xcd <= 8, "coo_column_bins: at most 8 bins per XCD, got %d", bins_per_xcd);
    const int nbins = 8 * bins_per_xcd;
    int       shift = 0;
    while (((int64_t)(m->ncol - 1) >> shift) >= kColBuckets) ++shift;

    hipStream_t         s = ctx->stream;
    unsigned long long* d_hist = nullptr;
    uint8_t*            d_bin  = nullptr;
    int64_t*            d_off  = nullptr;  // bin_src[nbins] | bin_dst[nbins]
    int32_t *           d_key = nullptr, *d_perm = nullptr;
    int                 rc = SPMV_OK;
    const unsigned      grid = (unsigned)std::min<int64_t>(kMaxGrid, ceil_div(m->nnz, kBlock));
    do
    {
        if (hipMalloc(&d_hist, sizeof(unsigned long long) * kColBuckets) != hipSuccess || hipMalloc(&d_bin, kColBuckets) != hipSuccess ||
            hipMalloc(&d_off, sizeof(int64_t) * 2 * (size_t)nbins) != hipSuccess || hipMalloc(&d_key, sizeof(int32_t) * (size_t)m->nnz) != hipSuccess ||
            hipMalloc(&d_perm, sizeof(int32_t) * (size_t)m->nnz) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * kColBuckets, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(coo_col_histogram_kernel, dim3(grid), dim3(kBlock), 0, s, m->nnz, m->b, shift, d_hist);
        std::vector<unsigned long long> hist(kColBuckets);
        if (hipMemcpyAsync(hist.data(), d_hist, sizeof(unsigned long long) * kColBuckets, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        // bin j ends with the bucket at which the running count reaches (j + 1) / nbins of the entries
        std::vector<uint8_t> bin_of(kColBuckets);
        std::vector<int64_t> count((size_t)nbins, 0), off(2 * (size_t)nbins, 0);
        {
            unsigned long long run = 0;
            int                j   = 0;
            for (int b = 0; b < kColBuckets; ++b)
            {
                bin_of[(size_t)b] = (uint8_t)j;
                count[(size_t)j] += (int64_t)hist[(size_t)b];
                run += hist[(size_t)b];
                while (j + 1 < nbins && run * (unsigned long long)nbins >= (unsigned long long)m->nnz * (unsigned long long)(j + 1)) ++j;
            }
        }
        int64_t src = 0, dst = 0;
        for (int j = 0; j < nbins; ++j)
        {
            if (j % bins_per_xcd == 0) m->cb_region[j / bins_per_xcd] = (int32_t)(dst / kBlockChunk);
            off[(size_t)j]         = src;
            off[(size_t)(nbins + j)] = dst;
            src += count[(size_t)j];
            dst += ceil_div(count[(size_t)j], (int64_t)kBlockChunk) * kBlockChunk;
            if ((j + 1) % bins_per_xcd == 0) m->cb_chunks[j / bins_per_xcd] = (int32_t)(dst / kBlockChunk) - m->cb_region[j / bins_per_xcd];
        }
        const int64_t padded = dst;
        if (hipMemcpyAsync(d_bin, bin_of.data(), kColBuckets, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(d_off, off.data(), sizeof(int64_t) * 2 * (size_t)nbins, hipMemcpyHostToDevice, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(coo_bin_keys_kernel, dim3(grid), dim3(kBlock), 0, s, m->nnz, m->b, shift, d_bin, d_key);
        int bits = 3;
        while ((1 << bits) < nbins) ++bits;
        if (sort_ids_by_key(ctx, d_key, m->nnz, bits, d_perm) != SPMV_OK)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        (void)hipFree(d_key);
        d_key = nullptr;
        if (hipMalloc(&m->cb_row, sizeof(int32_t) * (size_t)padded) != hipSuccess || hipMalloc(&m->cb_col, sizeof(int32_t) * (size_t)padded) != hipSuccess ||
            hipMalloc(&m->cb_val, sizeof(double) * (size_t)padded) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        hipLaunchKernelGGL(coo_bin_fill_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(padded, kBlock))), dim3(kBlock), 0, s, padded, m->cb_row,
                           m->cb_col, m->cb_val);
        hipLaunchKernelGGL(coo_bin_place_kernel, dim3(grid), dim3(kBlock), 0, s, m->nnz, d_perm, m->a, m->b, m->v, shift, d_bin, d_off, d_off + nbins,
                           m->cb_row, m->cb_col, m->cb_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        m->cb_bins   = nbins;
        m->cb_padded = padded;
        m->device_bytes += padded * 16;
    } while (0);
    if (d_hist) (void)hipFree(d_hist);
    if (d_bin) (void)hipFree(d_bin);
    if (d_off) (void)hipFree(d_off);
    if (d_key) (void)hipFree(d_key);
    if (d_perm) (void)hipFree(d_perm);
    if (rc != SPMV_OK)
    {
        coo_free_bins(m);  // (cb_bins is still 0: nothing was accounted yet)
        if (rc == SPMV_ERR_ALLOC) SPMV_FAIL(rc, "no device memory for the copy of %lld COO entries in column bins", (long long)m->nnz);
        SPMV_FAIL(rc, "building the copy in column bins failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return SPMV_OK;
}

// ---- the same scan over a CSR handle (SPMV_CSR_SEGSCAN) --------------------------------------------------------------
// Every other CSR kernel gives a row to one lane, one group of lanes or (panel layout: its row group) one workgroup; a row
// that holds a large share of the entries - the dense row of an arrow matrix, a constraint row, a hub - then runs on one
// CU while 255 wait (tools/sweep_structures.py "odd": 1M entries in a row, 1.26 ms under the panel kernel, 74 ms
// row-parallel).  The scan cuts the ENTRIES into equal pieces whatever their rows; what it needs is the row of every
// entry, 4 bytes each, written once here.  Rows without entries are never touched (y += 0).
namespace
{
__global__ __launch_bounds__(kBlock) void csr_expand_rows_kernel(int nrow, int64_t nnz, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ row)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock)
    {
        // the last r with row_ptr[r] <= e (row_ptr[0] = 0 and row_ptr[nrow] = nnz are checked when the handle is made)
        int lo = 0, hi = nrow;
        while (hi - lo > 1)
        {
            const int mid = lo + (hi - lo) / 2;
            if ((int64_t)row_ptr[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        row[e] = lo;
    }
}
}  // namespace

int csr_segscan_build(spmv_mat* m)
{
    if (m->seg_row || m->nnz == 0 || m->nrow == 0) return SPMV_OK;
    SPMV_REQUIRE(m->format == SPMV_FMT_CSR && m->a && m->b && m->v, "the segmented scan runs over a CSR handle's own arrays");
    spmv_ctx* ctx = m->ctx;
    if (hipMalloc(&m->seg_row, sizeof(int32_t) * (size_t)m->nnz) != hipSuccess)
    {
        (void)hipGetLastError();
        m->seg_row = nullptr;
        SPMV_FAIL(SPMV_ERR_ALLOC, "no device memory for the row index of %lld entries (segmented scan)", (long long)m->nnz);
    }
    hipLaunchKernelGGL(csr_expand_rows_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid * 4, ceil_div(m->nnz, kBlock))), dim3(kBlock), 0, ctx->stream,
                       (int)m->nrow, m->nnz, m->a, m->seg_row);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    {
        csr_segscan_free(m);
        SPMV_FAIL(SPMV_ERR_HIP, "writing the row index per entry failed: %s", hipGetErrorString(hipGetLastError()));
    }
    m->device_bytes += (int64_t)sizeof(int32_t) * m->nnz;
    return SPMV_OK;
}

void csr_segscan_free(spmv_mat* m)
{
    if (!m->seg_row) return;
    (void)hipFree(m->seg_row);
    m->seg_row = nullptr;
    m->device_bytes -= (int64_t)sizeof(int32_t) * m->nnz;
}

int csr_segscan_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nnz == 0) return SPMV_OK;
    if (!A->seg_row || !A->b || !A->v) SPMV_FAIL(SPMV_ERR_INVALID, "segmented scan selected but its row index was never built");
    hipLaunchKernelGGL(coo_segscan_kernel<true>, dim3((unsigned)ceil_div(A->nnz, kBlockChunk)), dim3(kBlock), 0, ctx->stream, A->nnz, A->seg_row, A->b,
                       A->v, x, y);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

int coo_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->nnz == 0) return SPMV_OK;
    if (A->coo_csr && A->kernel == SPMV_CSR_PANEL) return csr_apply(ctx, A->coo_csr, x, y);
    return coo_scan_apply(ctx, A, x, y);
}

namespace
{
int coo_scan_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    if (A->cb_bins)
    {
        coo_bins_tab tab;
        int          most = 0;
        for (int c = 0; c < 8; ++c)
        {
            tab.region[c] = A->cb_region[c];
            tab.chunks[c] = A->cb_chunks[c];
            most          = std::max(most, A->cb_chunks[c]);
        }
        if (most == 0) return SPMV_OK;
        hipLaunchKernelGGL(coo_segscan_bins_kernel, dim3(8u * (unsigned)most), dim3(kBlock), 0, ctx->stream, tab, A->cb_row, A->cb_col, A->cb_val, x, y);
        SPMV_HIP(hipGetLastError());
        return SPMV_OK;
    }
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
}  // namespace
}  // namespace spmv
