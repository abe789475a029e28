// kernels_csr_twophase.hip — CSR `y += A*x` in two streaming phases, for matrices whose x is so large against the
// rows one GPU holds that the panel kernel's sweeps of x dominate: the shard one rank owns in BASELINE config 5
// (10M rows x 80M columns: x = 640 MB, swept 2 rounds x 8 XCDs = 16 times = 10 GB over the fabric next to 3.85 GB
// of matrix).  Replaces CSRMatrixMatVector (src/mat_vec.cpp:44-67) for that shape; same sums, other order.
//
// The LDS of a CU can hold the accumulators of ~20000 rows OR ~20000 entries of x, never both sides of a scattered
// matrix.  So the product is cut where the two meet, and what crosses the cut travels as a dense stream:
//   phase A (expand, x-stationary)  one workgroup loads a PANEL of 20000 consecutive entries of x into LDS (coalesced,
//            every line of x exactly once per product) and writes, for every matrix entry whose column lies in the
//            panel, the PRODUCT value * x[col] to a stream `xg` — entries ordered by (panel, row group), so this is a
//            flat pass (2-byte index + 8-byte value in, 8-byte product out) with random access only inside LDS;
//   phase B (reduce, y-stationary)  one workgroup owns a row GROUP (<= 20000 rows, accumulators in LDS) and walks the
//            RUNS (panel p, group g) of that order: the product and a 2-byte local row per entry, contiguous inside a
//            run; products go into LDS with ds_add_f64 and y is touched once at the end.
// Bytes per entry: A reads 2 + 8, writes 8; B reads 8 + 2 = 28 against the panel kernel's 12 + x sweeps.  No gather
// ever leaves LDS, so there is nothing to keep in step and no dependence on where the columns fall; x is read once.
// Worth it when the sweeps would cost more than the 16 extra bytes per entry: see twophase_worth().
//
// Layout (built once per handle, like the reference's shard construction before its timed loop, src/mat_vec.cpp:240-268):
//   tp_val[e], tp_col[e] (uint16: column - panel base), tp_row[e] (uint16: row - group base)   e in (panel, group) order
//   tp_panel_ptr[P + 1]       first entry of every panel
//   tp_run[g * P + p] = {first entry, entries} of run (p, g)        (group-major: what a phase-B workgroup walks)
//   tp_xg[nnz]                the stream between the phases (scratch owned by the handle)
#include <atomic>
#include <cmath>
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kTpThreads   = 1024;
constexpr int kTpPanelCols = 20000;  // 160,000 B of x in LDS
constexpr int kTpGroupRows = 20000;  // 160,000 B of accumulators in LDS

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
constexpr int kExpandUnroll = 8;

// ---- build ----------------------------------------------------------------------------------------------------
// one lane per row: group of the row by binary search in gstart, then one key per entry
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void tp_place_kernel(int nrow, int ngroups, int P, int pcols, const int32_t* __restrict__ gstart,
                                                          const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                          const double* __restrict__ val, int32_t* __restrict__ count,
                                                          const int32_t* __restrict__ start, int32_t* __restrict__ cursor,
                                                          unsigned short* __restrict__ out_col, unsigned short* __restrict__ out_row,
                                                          double* __restrict__ out_val)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nrow) return;
    int lo = 0, hi = ngroups - 1;  // last g with gstart[g] <= r
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) / 2;
        if (gstart[mid] <= r)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int g = lo;
    for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j)
    {
        const int c   = col[j];
        const int p   = c / pcols;
        const int key = p * ngroups + g;
        if constexpr (COUNT)
            atomicAdd(count + key, 1);
        else
        {
            const int pos = start[key] + atomicAdd(cursor + key, 1);
            out_col[pos]  = (unsigned short)(c - p * pcols);
            out_row[pos]  = (unsigned short)(r - gstart[g]);
            out_val[pos]  = val[j];
        }
    }
}

// runs in group-major order + the panel boundaries
__global__ __launch_bounds__(kBlock) void tp_tables_kernel(int ngroups, int P, const int32_t* __restrict__ start /* [P*G + 1] */,
                                                           i32x2* __restrict__ run /* [G*P] */, int32_t* __restrict__ panel_ptr /* [P+1] */)
{
    const int64_t total = (int64_t)ngroups * P;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock)
    {
        const int p = (int)(i / ngroups), g = (int)(i % ngroups);
        i32x2     v;
        v.x                        = start[i];
        v.y                        = start[i + 1] - start[i];
        run[(size_t)g * P + p]     = v;
        if (g == 0) panel_ptr[p] = start[i];
        if (i == total - 1) panel_ptr[P] = start[total];
    }
}

// ---- phase A: xg[e] = tp_val[e] * x[panel base + tp_col[e]] --------------------------------------------------------
// Two entries per lane: 4-byte index loads, 16-byte value loads and 16-byte product stores, every wavefront
// instruction contiguous (256 B / 1 KiB / 1 KiB).  kExpandUnroll pairs per lane in flight: a workgroup is all the
// parallelism a CU has here (one 160 KB panel per CU).
__global__ __launch_bounds__(kTpThreads) void tp_expand_kernel(int P, int pcols, int ncol, const int32_t* __restrict__ panel_ptr,
                                                               const unsigned short* __restrict__ tp_col, const double* __restrict__ tp_val,
                                                               const double* __restrict__ x, double* __restrict__ xg)
{
    extern __shared__ double xs[];  // pcols entries of x
    for (int p = blockIdx.x; p < P; p += gridDim.x)
    {
        const int c0 = p * pcols;
        const int n  = min(pcols, ncol - c0);
        // the panel: 16-byte loads (panel bases are multiples of 20000 entries; x itself is checked), five in flight
        const double* __restrict__ xp = x + c0;
        if ((reinterpret_cast<uintptr_t>(xp) & 15) == 0)
        {
            const int    pairs = n / 2;
            const f64x2* x2    = reinterpret_cast<const f64x2*>(xp);
            f64x2*       s2    = reinterpret_cast<f64x2*>(xs);
            for (int i0 = 0; i0 < pairs; i0 += kTpThreads * 5)
            {
                f64x2 t[5];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const int i = i0 + k * kTpThreads + (int)threadIdx.x;
                    if (i < pairs) t[k] = x2[i];
                }
#pragma unroll
                for (int k = 0; k < 5; ++k)
                {
                    const int i = i0 + k * kTpThreads + (int)threadIdx.x;
                    if (i < pairs) s2[i] = t[k];
                }
            }
            if ((n & 1) && threadIdx.x == 0) xs[n - 1] = xp[n - 1];
        }
        else
            for (int i = threadIdx.x; i < n; i += kTpThreads) xs[i] = xp[i];
        __syncthreads();
        const int b = panel_ptr[p], e = panel_ptr[p + 1];
        const int a0 = min((b + 1) & ~1, e), a1 = max(e & ~1, a0);  // whole pairs
        if (threadIdx.x == 0 && b < a0) xg[b] = tp_val[b] * xs[tp_col[b]];
        if (threadIdx.x == 1 && a1 < e) xg[a1] = tp_val[a1] * xs[tp_col[a1]];
        const u16x2* __restrict__ c2 = reinterpret_cast<const u16x2*>(tp_col);
        const f64x2* __restrict__ v2 = reinterpret_cast<const f64x2*>(tp_val);
        f64x2* __restrict__ o2       = reinterpret_cast<f64x2*>(xg);
        const int t_end = a1 / 2;
        if (a0 / 2 < t_end)
        {
            // software-pipelined: the loads of the next kExpandUnroll pairs are issued before this set is multiplied and
            // stored, so the HBM latency is paid once per panel, not once per set
            u16x2 c[2][kExpandUnroll];
            f64x2 v[2][kExpandUnroll];
            auto  fetch = [&](int t0, u16x2(&cc)[kExpandUnroll], f64x2(&vv)[kExpandUnroll]) {
#pragma unroll
                for (int k = 0; k < kExpandUnroll; ++k)
                {
                    const int t = min(t0 + k * kTpThreads + (int)threadIdx.x, t_end - 1);  // past the end: re-read the last pair
                    cc[k]       = __builtin_nontemporal_load(c2 + t);
                    vv[k]       = __builtin_nontemporal_load(v2 + t);
                }
            };
            auto emit = [&](int t0, const u16x2(&cc)[kExpandUnroll], const f64x2(&vv)[kExpandUnroll]) {
#pragma unroll
                for (int k = 0; k < kExpandUnroll; ++k)
                {
                    const int t = t0 + k * kTpThreads + (int)threadIdx.x;
                    if (t < t_end)
                    {
                        f64x2 o;
                        o.x   = vv[k].x * xs[cc[k].x];
                        o.y   = vv[k].y * xs[cc[k].y];
                        o2[t] = o;
                    }
                }
            };
            constexpr int SET = kTpThreads * kExpandUnroll;
            fetch(a0 / 2, c[0], v[0]);
            for (int t0 = a0 / 2; t0 < t_end; t0 += 2 * SET)
            {
                fetch(t0 + SET, c[1], v[1]);  // (clamped when past the end)
                emit(t0, c[0], v[0]);
                fetch(t0 + 2 * SET, c[0], v[0]);
                emit(t0 + SET, c[1], v[1]);
            }
        }
        __syncthreads();  // the panel is replaced next
    }
}

// ---- phase B: y[group] += sum over the group's runs of the products --------------------------------------------------------
// A wavefront takes RUNS_IN_FLIGHT runs at a time (runs p = wave, wave + 16, ...) and keeps one load of each of
// the two streams per run in flight, so a workgroup has 16 x 16 x 2 loads out: the HBM latency is covered although a
// run is only a couple of hundred entries long.
constexpr int kRunsInFlight = 16;
__global__ __launch_bounds__(kTpThreads) void tp_reduce_kernel(const int32_t* __restrict__ gstart, int ngroups, int P,
                                                               const i32x2* __restrict__ run,
                                                               const unsigned short* __restrict__ tp_row, const double* __restrict__ xg,
                                                               double* __restrict__ y, int overwrite, const double* __restrict__ dot_w,
                                                               double* __restrict__ dot_out)
{
    extern __shared__ double acc[];
    constexpr int NW = kTpThreads / kWave;
    const int     lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int r0   = gstart[g];
        const int rows = gstart[g + 1] - r0;
        for (int i = threadIdx.x; i < rows; i += kTpThreads) acc[i] = 0.0;
        __syncthreads();
        const i32x2* __restrict__ rg = run + (size_t)g * P;
        for (int p0 = wave * kRunsInFlight; p0 < P; p0 += NW * kRunsInFlight)
        {
            int pos[kRunsInFlight], end[kRunsInFlight];
            int longest = 0;
#pragma unroll
            for (int k = 0; k < kRunsInFlight; ++k)
            {
                i32x2 d;
                d.x = 0;
                d.y = 0;
                if (p0 + k < P) d = rg[p0 + k];  // wave-uniform: scalar loads
                pos[k]  = d.x + lane;
                end[k]  = d.x + d.y;
                longest = max(longest, d.y);
            }
            // (software-pipelining the steps with half as many runs in flight was measured: 1.44 ms against 0.82)
            for (int t = 0; t < longest; t += kWave)
            {
                double   xv[kRunsInFlight];
                unsigned r[kRunsInFlight];
#pragma unroll
                for (int k = 0; k < kRunsInFlight; ++k)
                {
                    const bool on = pos[k] < end[k];
                    xv[k] = on ? load_stream(xg + pos[k]) : 0.0;
                    r[k]  = on ? (unsigned)load_stream(tp_row + pos[k]) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int k = 0; k < kRunsInFlight; ++k)
                {
                    if (r[k] != 0xFFFFFFFFu) atomicAdd(&acc[r[k]], xv[k]);  // ds_add_f64
                    pos[k] += kWave;
                }
            }
        }
        __syncthreads();
        double part = 0.0;
        for (int i = threadIdx.x; i < rows; i += kTpThreads)
        {
            const double yn = overwrite ? acc[i] : y[r0 + i] + acc[i];
            y[r0 + i]       = yn;
            if (dot_w) part = fma(dot_w[r0 + i], yn, part);
        }
        if (dot_w)
        {
            part = wave_sum(part);
            if (lane == 0) slot_add(dot_out, part);
        }
        __syncthreads();
    }
}
}  // namespace

void csr_twophase_free(spmv_mat* m)
{
    for (void** p : {(void**)&m->tp_val, (void**)&m->tp_col, (void**)&m->tp_row, (void**)&m->tp_xg, (void**)&m->tp_run,
                     (void**)&m->tp_panel_ptr, (void**)&m->tp_gstart})
        if (*p)
        {
            (void)hipFree(*p);
            *p = nullptr;
        }
    m->device_bytes -= m->tp_bytes;
    m->tp_bytes = 0;
}

// x sweeps of the panel kernel (8 XCDs x rounds x all of x) against the 16 extra bytes per entry of the two phases
bool csr_twophase_worth(const spmv_mat* m)
{
    if (m->nnz < ((int64_t)2 << 20) || m->nrow <= 0) return false;
    const double rounds = std::max(1.0, std::ceil((double)m->nrow / ((double)kNumCu * kTpGroupRows)));
    const double sweeps = 8.0 * (double)m->ncol * kNumXcd * rounds;
    return sweeps > 16.0 * (double)m->nnz * 1.25 + 8.0 * (double)m->ncol;
}

int csr_twophase_build(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (m->tp_val || m->nrow == 0 || m->nnz == 0) return SPMV_OK;
    SPMV_REQUIRE(m->b && m->v, "the two-phase layout needs the CSR arrays (panel_keep_csr = 0 released them)");
    hipStream_t s = ctx->stream;
    // row groups: whole rounds of 256 workgroups, equal rows (the entries of a row may be anywhere: nothing to balance
    // by column), at most kTpGroupRows
    int per = 1;
    for (int rounds = 1;; ++rounds)
    {
        per = (int)ceil_div(m->nrow, (int64_t)kNumCu * rounds);
        if (per <= kTpGroupRows) break;
    }
    const int            ngroups = (int)ceil_div(m->nrow, per);
    std::vector<int32_t> gstart((size_t)ngroups + 1);
    for (int g = 0; g <= ngroups; ++g) gstart[(size_t)g] = (int32_t)std::min<int64_t>((int64_t)g * per, m->nrow);
    const int     pcols = kTpPanelCols;
    const int     P     = (int)ceil_div(m->ncol, pcols);
    const int64_t keys  = (int64_t)P * ngroups;
    SPMV_REQUIRE(keys < ((int64_t)1 << 28), "two-phase layout: %d panels x %d groups is too fine", P, ngroups);
    int32_t *count = nullptr, *start = nullptr;
    int      rc    = SPMV_OK;
    const size_t nnz = (size_t)m->nnz;
    do
    {
        if (hipMalloc(&m->tp_gstart, sizeof(int32_t) * gstart.size()) != hipSuccess || hipMalloc(&count, sizeof(int32_t) * (size_t)(keys + 1)) != hipSuccess ||
            hipMalloc(&start, sizeof(int32_t) * (size_t)(keys + 1)) != hipSuccess || hipMalloc(&m->tp_val, sizeof(double) * nnz) != hipSuccess ||
            hipMalloc(&m->tp_col, sizeof(unsigned short) * nnz + 8) != hipSuccess || hipMalloc(&m->tp_row, sizeof(unsigned short) * nnz + 8) != hipSuccess ||
            hipMalloc(&m->tp_xg, sizeof(double) * nnz) != hipSuccess || hipMalloc(&m->tp_run, sizeof(i32x2) * (size_t)keys) != hipSuccess ||
            hipMalloc(&m->tp_panel_ptr, sizeof(int32_t) * ((size_t)P + 1)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (hipMemcpyAsync(m->tp_gstart, gstart.data(), sizeof(int32_t) * gstart.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)(keys + 1), s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        const unsigned rgrid = (unsigned)ceil_div(m->nrow, kBlock);
        hipLaunchKernelGGL(tp_place_kernel<true>, dim3(rgrid), dim3(kBlock), 0, s, m->nrow, ngroups, P, pcols, m->tp_gstart, m->a, m->b, m->v,
                           count, (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned short*)nullptr, (unsigned short*)nullptr, (double*)nullptr);
        if ((rc = exclusive_scan_i32(ctx, count, start, keys + 1)) != SPMV_OK) break;
        if (hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)(keys + 1), s) != hipSuccess)  // re-used as the cursors
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(tp_place_kernel<false>, dim3(rgrid), dim3(kBlock), 0, s, m->nrow, ngroups, P, pcols, m->tp_gstart, m->a, m->b, m->v,
                           (int32_t*)nullptr, start, count, m->tp_col, m->tp_row, m->tp_val);
        hipLaunchKernelGGL(tp_tables_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div(keys, kBlock))), dim3(kBlock), 0, s, ngroups, P,
                           start, (i32x2*)m->tp_run, m->tp_panel_ptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;  // gstart (host) is done with
    } while (0);
    if (count) (void)hipFree(count);
    if (start) (void)hipFree(start);
    if (rc != SPMV_OK)
    {
        (void)hipStreamSynchronize(s);
        csr_twophase_free(m);
        SPMV_FAIL(rc, "building the two-phase layout (%d groups x %d panels) failed: %s", ngroups, P, hipGetErrorString(hipGetLastError()));
    }
    m->tp_ngroups   = ngroups;
    m->tp_panels    = P;
    m->tp_pcols     = pcols;
    m->tp_max_rows  = per;
    m->tp_bytes     = (int64_t)nnz * 20 + keys * 8 + (int64_t)(P + 1 + ngroups + 1) * 4 + 16;
    m->device_bytes += m->tp_bytes;
    return SPMV_OK;
}

int csr_twophase_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex)
{
    if (A->nrow == 0 || A->nnz == 0)
    {
        if (ex.overwrite && A->nrow > 0) SPMV_TRY(vec_fill(ctx, y, A->nrow, 0.0));
        return SPMV_OK;
    }
    // the kernels dereference exactly these: refuse on the host rather than fault on the GPU
    if (!A->tp_val || !A->tp_col || !A->tp_row || !A->tp_xg || !A->tp_run || !A->tp_panel_ptr || !A->tp_gstart || !x || !y || A->tp_panels <= 0 ||
        A->tp_pcols <= 0 || A->tp_pcols > kTpPanelCols || A->tp_max_rows > kTpGroupRows)
        SPMV_FAIL(SPMV_ERR_INVALID, "two-phase kernel selected but its layout was not built");
    static std::atomic<unsigned long long> granted{0};  // bit per device
    if (!((granted.load(std::memory_order_relaxed) >> ctx->device) & 1ull))
    {
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160008));
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160008));
        granted.fetch_or(1ull << ctx->device, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(tp_expand_kernel, dim3((unsigned)std::min(A->tp_panels, kNumCu)), dim3(kTpThreads), sizeof(double) * (size_t)A->tp_pcols,
                       ctx->stream, A->tp_panels, A->tp_pcols, A->ncol, A->tp_panel_ptr, (const unsigned short*)A->tp_col, (const double*)A->tp_val, x, A->tp_xg);
    hipLaunchKernelGGL(tp_reduce_kernel, dim3((unsigned)std::min(A->tp_ngroups, kNumCu)), dim3(kTpThreads), sizeof(double) * (size_t)A->tp_max_rows,
                       ctx->stream, A->tp_gstart, A->tp_ngroups, A->tp_panels, (const i32x2*)A->tp_run, (const unsigned short*)A->tp_row,
                       (const double*)A->tp_xg, y, ex.overwrite ? 1 : 0, ex.dot_w, ex.dot_out);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
