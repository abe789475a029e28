// kernels_csr_twophase.hip — CSR `y += A*x` in two streaming phases, for matrices whose x is so large against the
// rows one GPU holds that the panel kernel's sweeps of x dominate: the shard one rank owns in BASELINE config 5
// (10M rows x 80M columns: x = 640 MB, swept 2 rounds x 8 XCDs = 16 times = 10 GB over the fabric next to 3.85 GB
// of matrix).  Replaces CSRMatrixMatVector (src/mat_vec.cpp:44-67) for that shape; same sums, other order.
//
// The LDS of a CU can hold the accumulators of ~20000 rows OR ~20000 entries of x, never both sides of a scattered
// matrix.  So the product is cut where the two meet, and what crosses the cut travels as a dense stream:
//   phase A (expand, x-stationary)  one workgroup loads a PANEL of consecutive entries of x into LDS (coalesced,
//            every line of x exactly once per product) and writes, for every matrix entry whose column lies in the
//            panel, the PRODUCT value * x[col] to a stream `xg`.  It READS the entries in (panel, row group) order — a
//            flat pass: 2-byte index + 8-byte value — and WRITES the products in (row group, panel) order: the entries
//            of one RUN (panel p, group g) are contiguous in both, runs are padded to whole 128-byte lines of
//            products (16 entries), and a table holds the destination line of every source line;
//   phase B (reduce, y-stationary)  one workgroup owns a row GROUP (<= 20000 rows, accumulators in LDS) and streams
//            its stretch of xg with a 2-byte local row per entry, flat from the first entry to the last; products go
//            into LDS with ds_add_f64 and y is touched once at the end.
// Bytes per entry: A reads 2 + 8 (+ 0.25 for the table), writes 8; B reads 8 + 2: 28.25, times the padding (the C5
// shard: runs of 160 entries, +5 %), against the panel kernel's 12 + x sweeps.  No gather ever leaves LDS, so there
// is nothing to keep in step and no dependence on where the columns fall; x is read once.  The scatter between the
// two orders is carried by phase A's stores (whole lines, nothing waits for them), so both phases read flat streams.
// Worth it when the sweeps would cost more than the 16 extra bytes per entry: see csr_twophase_worth().
//
// Layout (built once per handle, like the reference's shard construction before its timed loop, src/mat_vec.cpp:240-268):
//   tp_val[e], tp_col[e] (uint16: column - panel base)     e in (panel, group) order, runs padded to 16 (value 0)
//   tp_row[e'] (uint16: row - group base; 0xFFFF = padding) e' in (group, panel) order
//   tp_blk[e / 16]            destination line e' / 16 of source line e / 16
//   tp_panel_ptr[P + 1], tp_group_ptr[G + 1]     first entry of every panel in e, of every group in e'
//   tp_xg[padded nnz]         the stream between the phases (scratch owned by the handle), in e' order
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kTpPanelCols = 20000;  // 160,000 B of x in LDS
constexpr int kTpGroupRows = 20000;  // 160,000 B of accumulators in LDS
constexpr int kTpLine      = 16;     // products per 128-byte line: the unit runs are padded to
constexpr unsigned kTpPadRow = 0xFFFFu;

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));

// ---- build ----------------------------------------------------------------------------------------------------
// one lane per row: group of the row by binary search in gstart, then one key per entry
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void tp_place_kernel(int nrow, int ngroups, int P, int pcols, const int32_t* __restrict__ gstart,
                                                          const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                          const double* __restrict__ val, int32_t* __restrict__ count,
                                                          const int32_t* __restrict__ start_pg, const int32_t* __restrict__ start_gp,
                                                          int32_t* __restrict__ cursor, unsigned short* __restrict__ out_col,
                                                          unsigned short* __restrict__ out_row, double* __restrict__ out_val)
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
            const int k   = atomicAdd(cursor + key, 1);
            const int src = start_pg[key] + k;
            const int dst = start_gp[(size_t)g * P + p] + k;
            out_col[src]  = (unsigned short)(c - p * pcols);
            out_val[src]  = val[j];
            out_row[dst]  = (unsigned short)(r - gstart[g]);
        }
    }
}

// run lengths padded to whole lines, in both orders
__global__ __launch_bounds__(kBlock) void tp_pad_kernel(int ngroups, int P, int32_t* __restrict__ count_pg /* [P*G + 1], in place */,
                                                        int32_t* __restrict__ count_gp /* [G*P + 1] */)
{
    const int64_t total = (int64_t)ngroups * P;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= total; i += (int64_t)gridDim.x * kBlock)
    {
        if (i == total)
        {
            count_pg[i] = 0;
            count_gp[i] = 0;
            continue;
        }
        const int p = (int)(i / ngroups), g = (int)(i % ngroups);
        const int c = (count_pg[i] + kTpLine - 1) / kTpLine * kTpLine;
        count_pg[i] = c;
        count_gp[(size_t)g * P + p] = c;
    }
}

// destination line of every source line + where panels (source order) and groups (destination order) begin
__global__ __launch_bounds__(kBlock) void tp_tables_kernel(int ngroups, int P, const int32_t* __restrict__ start_pg /* [P*G + 1] */,
                                                           const int32_t* __restrict__ start_gp /* [G*P + 1] */,
                                                           int32_t* __restrict__ blk /* [padded / 16] */,
                                                           int32_t* __restrict__ panel_ptr /* [P+1] */, int32_t* __restrict__ group_ptr /* [G+1] */)
{
    const int64_t total = (int64_t)ngroups * P;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock)
    {
        const int p = (int)(i / ngroups), g = (int)(i % ngroups);
        const int s = start_pg[i] / kTpLine, n = (start_pg[i + 1] - start_pg[i]) / kTpLine;
        const int d = start_gp[(size_t)g * P + p] / kTpLine;
        for (int b = 0; b < n; ++b) blk[s + b] = d + b;
        if (g == 0) panel_ptr[p] = start_pg[i];
        if (p == 0) group_ptr[g] = start_gp[(size_t)g * P];
        if (i == total - 1)
        {
            panel_ptr[P]       = start_pg[total];
            group_ptr[ngroups] = start_gp[total];
        }
    }
}

// ---- phase A: xg[dst(e)] = tp_val[e] * x[panel base + tp_col[e]] ----------------------------------------------------
// Two entries per lane: 4-byte index loads, 16-byte value loads and 16-byte product stores; eight lanes fill one
// 128-byte line of products, whose destination line comes from tp_blk (one 4-byte load shared by the eight).  The
// loads of the next set of UNROLL pairs per lane are issued before this set is multiplied and stored, and the first
// set of a panel before its x is fetched, so the HBM latency is paid once per panel (UNROLL 6: 114 VGPRs; 8 spills).
// THREADS = 1024: one workgroup
// per CU with a panel of up to 20000 columns; 512: two per CU with up to 10000 each, one streaming while the other
// changes its panel.
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void tp_expand_kernel(int P, int pcols, int ncol, const int32_t* __restrict__ panel_ptr,
                                                            const unsigned short* __restrict__ tp_col, const double* __restrict__ tp_val,
                                                            const int32_t* __restrict__ tp_blk, const double* __restrict__ x,
                                                            double* __restrict__ xg, int rotate)
{
    extern __shared__ double xs[];  // pcols entries of x
    const u16x2* __restrict__ c2 = reinterpret_cast<const u16x2*>(tp_col);
    const f64x2* __restrict__ v2 = reinterpret_cast<const f64x2*>(tp_val);
    f64x2* __restrict__ o2       = reinterpret_cast<f64x2*>(xg);
    constexpr int SET = THREADS * UNROLL;
    for (int p = blockIdx.x; p < P; p += gridDim.x)
    {
        const int c0 = p * pcols;
        const int n  = min(pcols, ncol - c0);
        const int t_begin = panel_ptr[p] / 2, t_end = panel_ptr[p + 1] / 2;  // pairs: runs are padded to 16 entries
        const int len = t_end - t_begin;
        // rotate: workgroup b starts b / 256 of the way through its panel and wraps around, so that at any moment the
        // workgroups write into different row groups' stretches of xg instead of all into the same one (4-5 % on
        // average over allocations: profiles/r02_probe_twophase_placement.txt)
        const int rot = rotate ? (int)(((int64_t)len * (int)(blockIdx.x % 256u) / 256) & ~7) : 0;
        auto      phys = [&](int u) { return t_begin + (u + rot >= len ? u + rot - len : u + rot); };
        u16x2 c[2][UNROLL];
        f64x2 v[2][UNROLL];
        int   d[2][UNROLL];
        auto  fetch = [&](int u0, u16x2(&cc)[UNROLL], f64x2(&vv)[UNROLL], int(&dd)[UNROLL]) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
            {
                const int t = phys(min(u0 + k * THREADS + (int)threadIdx.x, len - 1));  // past the end: re-read the last pair
                cc[k]       = __builtin_nontemporal_load(c2 + t);
                vv[k]       = __builtin_nontemporal_load(v2 + t);
                dd[k]       = __builtin_nontemporal_load(tp_blk + (t >> 3));
            }
        };
        auto emit = [&](int u0, const u16x2(&cc)[UNROLL], const f64x2(&vv)[UNROLL], const int(&dd)[UNROLL]) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
            {
                const int u = u0 + k * THREADS + (int)threadIdx.x;
                if (u < len)
                {
                    const int t = phys(u);
                    f64x2     o;
                    o.x = vv[k].x * xs[cc[k].x];
                    o.y = vv[k].y * xs[cc[k].y];
                    __builtin_nontemporal_store(o, o2 + ((size_t)dd[k] * 8 + (t & 7)));  // read again only 2.7 GB later: 2-3 %
                }
            }
        };
        if (len > 0) fetch(0, c[0], v[0], d[0]);  // (workgroup-uniform)
        // the panel: 16-byte loads when x allows (panel bases are multiples of pcols entries)
        const double* __restrict__ xp = x + c0;
        if ((reinterpret_cast<uintptr_t>(xp) & 15) == 0)
        {
            const int    pairs = n / 2;
            const f64x2* x2    = reinterpret_cast<const f64x2*>(xp);
            f64x2*       s2    = reinterpret_cast<f64x2*>(xs);
            constexpr int XU   = 5;
            for (int i0 = 0; i0 < pairs; i0 += THREADS * XU)
            {
                f64x2 t[XU];
#pragma unroll
                for (int k = 0; k < XU; ++k)
                {
                    const int i = i0 + k * THREADS + (int)threadIdx.x;
                    if (i < pairs) t[k] = x2[i];
                }
#pragma unroll
                for (int k = 0; k < XU; ++k)
                {
                    const int i = i0 + k * THREADS + (int)threadIdx.x;
                    if (i < pairs) s2[i] = t[k];
                }
            }
            if ((n & 1) && threadIdx.x == 0) xs[n - 1] = xp[n - 1];
        }
        else
            for (int i = threadIdx.x; i < n; i += THREADS) xs[i] = xp[i];
        __syncthreads();
        for (int u0 = 0; u0 < len; u0 += 2 * SET)
        {
            fetch(u0 + SET, c[1], v[1], d[1]);  // (clamped when past the end)
            emit(u0, c[0], v[0], d[0]);
            fetch(u0 + 2 * SET, c[0], v[0], d[0]);
            emit(u0 + SET, c[1], v[1], d[1]);
        }
        __syncthreads();  // the panel is replaced next
    }
}

// ---- phase B: y[group] += the group's stretch of the product stream ---------------------------------------------------
// Flat: a lane takes pairs t, t + 1024, ... of the stretch; 16-byte product loads, 4-byte row loads, the next set in
// flight while this one goes into LDS.
constexpr int kTpThreads     = 1024;
constexpr int kReduceUnroll  = 8;
__global__ __launch_bounds__(kTpThreads) void tp_reduce_kernel(const int32_t* __restrict__ gstart, int ngroups,
                                                               const int32_t* __restrict__ group_ptr,
                                                               const unsigned short* __restrict__ tp_row, const double* __restrict__ xg,
                                                               double* __restrict__ y, int overwrite, const double* __restrict__ dot_w,
                                                               double* __restrict__ dot_out)
{
    extern __shared__ double acc[];
    const int lane = threadIdx.x & 63;
    const u16x2* __restrict__ r2 = reinterpret_cast<const u16x2*>(tp_row);
    const f64x2* __restrict__ p2 = reinterpret_cast<const f64x2*>(xg);
    constexpr int U = kReduceUnroll, SET = kTpThreads * U;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int r0   = gstart[g];
        const int rows = gstart[g + 1] - r0;
        const int t_begin = group_ptr[g] / 2, t_end = group_ptr[g + 1] / 2;
        u16x2 r[2][U];
        f64x2 v[2][U];
        auto  fetch = [&](int t0, u16x2(&rr)[U], f64x2(&vv)[U]) {
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int t = min(t0 + k * kTpThreads + (int)threadIdx.x, t_end - 1);
                rr[k]       = __builtin_nontemporal_load(r2 + t);
                vv[k]       = __builtin_nontemporal_load(p2 + t);
            }
        };
        auto add = [&](int t0, const u16x2(&rr)[U], const f64x2(&vv)[U]) {
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int t = t0 + k * kTpThreads + (int)threadIdx.x;
                if (t < t_end)
                {
                    if (rr[k].x != kTpPadRow) atomicAdd(&acc[rr[k].x], vv[k].x);  // ds_add_f64
                    if (rr[k].y != kTpPadRow) atomicAdd(&acc[rr[k].y], vv[k].y);
                }
            }
        };
        if (t_begin < t_end) fetch(t_begin, r[0], v[0]);
        for (int i = threadIdx.x; i < rows; i += kTpThreads) acc[i] = 0.0;
        __syncthreads();
        for (int t0 = t_begin; t0 < t_end; t0 += 2 * SET)
        {
            fetch(t0 + SET, r[1], v[1]);
            add(t0, r[0], v[0]);
            fetch(t0 + 2 * SET, r[0], v[0]);
            add(t0 + SET, r[1], v[1]);
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
    for (void** p : {(void**)&m->tp_val, (void**)&m->tp_col, (void**)&m->tp_row, (void**)&m->tp_xg, (void**)&m->tp_blk,
                     (void**)&m->tp_panel_ptr, (void**)&m->tp_group_ptr, (void**)&m->tp_gstart})
        if (*p)
        {
            (void)hipFree(*p);
            *p = nullptr;
        }
    m->device_bytes -= m->tp_bytes;
    m->tp_bytes  = 0;
    m->tp_padded = 0;
}

namespace
{
int tp_groups_per(const spmv_mat* m)  // rows per group: whole rounds of 256 workgroups, equal rows, at most kTpGroupRows
{
    for (int rounds = 1;; ++rounds)
    {
        const int per = (int)ceil_div(m->nrow, (int64_t)kNumCu * rounds);
        if (per <= kTpGroupRows) return std::max(per, 1);
    }
}
int tp_panel_cols(const spmv_mat* m) { return m->tp_pcols_req > 0 ? std::min(m->tp_pcols_req, kTpPanelCols) : kTpPanelCols; }
}  // namespace

// The panel kernel sweeps x once per XCD and round (8 x rounds x 8 ncol bytes over the fabric); the two phases move
// ~20 more bytes per entry than it and read x once.  Measured on 10M rows x 32 (profiles/r02_tune_csr_c5_*): panel
// 1.13 / 1.63 / 1.98 / 2.37 ms at ncol = 10M / 20M / 40M / 80M against a flat 1.8-1.9 ms: the two phases win from
// ~3x more columns than rows on, i.e. when the sweeps exceed 12 bytes per entry.  And the runs (panel x row group)
// must be long enough that padding them to whole lines costs little.
bool csr_twophase_worth(const spmv_mat* m)
{
    if (m->nnz < ((int64_t)8 << 20) || m->nrow <= 0) return false;
    // local columns (bands, stencils: the mean column window of 256 rows is a small part of x): the panel kernel only
    // sweeps what its groups touch, the model below does not apply (tools/sweep_sizes.py: a 20M x 16 band matrix took
    // 1.89 ms through the two phases against 0.7 through the panel kernel)
    if (m->win_avg_span < 0.5 * (double)m->ncol) return false;
    const double rounds = std::max(1.0, std::ceil((double)m->nrow / ((double)kNumCu * kTpGroupRows)));
    const double sweeps = 8.0 * (double)m->ncol * kNumXcd * rounds;
    const double runs   = (double)ceil_div(m->ncol, tp_panel_cols(m)) * (double)ceil_div(m->nrow, tp_groups_per(m));
    if ((double)m->nnz < 64.0 * runs) return false;
    if ((double)m->nnz + (kTpLine - 1) * runs >= 2147483648.0 || runs >= 134217728.0) return false;  // the padded layout must stay within int32
    return sweeps > 12.0 * (double)m->nnz;
}

int csr_twophase_build(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (m->nrow == 0 || m->nnz == 0) return SPMV_OK;
    const int pcols = tp_panel_cols(m);
    if (m->tp_val && m->tp_pcols == pcols) return SPMV_OK;
    SPMV_REQUIRE(m->b && m->v, "the two-phase layout needs the CSR arrays (panel_keep_csr = 0 released them)");
    csr_twophase_free(m);
    hipStream_t s = ctx->stream;
    const int            per     = tp_groups_per(m);
    const int            ngroups = (int)ceil_div(m->nrow, per);
    std::vector<int32_t> gstart((size_t)ngroups + 1);
    for (int g = 0; g <= ngroups; ++g) gstart[(size_t)g] = (int32_t)std::min<int64_t>((int64_t)g * per, m->nrow);
    const int     P    = (int)ceil_div(m->ncol, pcols);
    const int64_t keys = (int64_t)P * ngroups;
    SPMV_REQUIRE(keys < ((int64_t)1 << 27) && m->nnz + keys * (kTpLine - 1) < ((int64_t)1 << 31),
                 "two-phase layout: %d panels x %d groups is too fine for %lld entries", P, ngroups, (long long)m->nnz);
    int32_t *cnt_pg = nullptr, *cnt_gp = nullptr, *start_pg = nullptr, *start_gp = nullptr;
    int      rc     = SPMV_OK;
    const size_t kbytes = sizeof(int32_t) * (size_t)(keys + 1);
    do
    {
        if (hipMalloc(&m->tp_gstart, sizeof(int32_t) * gstart.size()) != hipSuccess || hipMalloc(&cnt_pg, kbytes) != hipSuccess ||
            hipMalloc(&cnt_gp, kbytes) != hipSuccess || hipMalloc(&start_pg, kbytes) != hipSuccess || hipMalloc(&start_gp, kbytes) != hipSuccess ||
            hipMalloc(&m->tp_panel_ptr, sizeof(int32_t) * ((size_t)P + 1)) != hipSuccess ||
            hipMalloc(&m->tp_group_ptr, sizeof(int32_t) * ((size_t)ngroups + 1)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (hipMemcpyAsync(m->tp_gstart, gstart.data(), sizeof(int32_t) * gstart.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemsetAsync(cnt_pg, 0, kbytes, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        const unsigned rgrid = (unsigned)ceil_div(m->nrow, kBlock);
        const unsigned kgrid = (unsigned)std::min<int64_t>(kMaxGrid, ceil_div(keys + 1, kBlock));
        hipLaunchKernelGGL(tp_place_kernel<true>, dim3(rgrid), dim3(kBlock), 0, s, m->nrow, ngroups, P, pcols, m->tp_gstart, m->a, m->b, m->v,
                           cnt_pg, (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (unsigned short*)nullptr,
                           (unsigned short*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL(tp_pad_kernel, dim3(kgrid), dim3(kBlock), 0, s, ngroups, P, cnt_pg, cnt_gp);
        if ((rc = exclusive_scan_i32(ctx, cnt_pg, start_pg, keys + 1)) != SPMV_OK) break;
        if ((rc = exclusive_scan_i32(ctx, cnt_gp, start_gp, keys + 1)) != SPMV_OK) break;
        int32_t padded = 0;
        if (hipMemcpyAsync(&padded, start_pg + keys, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (padded < m->nnz || padded % kTpLine != 0 || (int64_t)padded > m->nnz + keys * (kTpLine - 1))
        {
            rc = SPMV_ERR_HIP;  // (the scan disagrees with the counts)
            break;
        }
        const size_t np = (size_t)padded;
        if (hipMalloc(&m->tp_val, sizeof(double) * np) != hipSuccess || hipMalloc(&m->tp_col, sizeof(unsigned short) * np) != hipSuccess ||
            hipMalloc(&m->tp_row, sizeof(unsigned short) * np) != hipSuccess || hipMalloc(&m->tp_xg, sizeof(double) * np) != hipSuccess ||
            hipMalloc(&m->tp_blk, sizeof(int32_t) * (np / kTpLine)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        // padding: value 0 at column 0 of the panel on the source side, row 0xFFFF on the destination side; cnt_pg becomes the cursors
        if (hipMemsetAsync(m->tp_val, 0, sizeof(double) * np, s) != hipSuccess || hipMemsetAsync(m->tp_col, 0, sizeof(unsigned short) * np, s) != hipSuccess ||
            hipMemsetAsync(m->tp_row, 0xFF, sizeof(unsigned short) * np, s) != hipSuccess || hipMemsetAsync(cnt_pg, 0, kbytes, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(tp_place_kernel<false>, dim3(rgrid), dim3(kBlock), 0, s, m->nrow, ngroups, P, pcols, m->tp_gstart, m->a, m->b, m->v,
                           (int32_t*)nullptr, start_pg, start_gp, cnt_pg, m->tp_col, m->tp_row, m->tp_val);
        hipLaunchKernelGGL(tp_tables_kernel, dim3(kgrid), dim3(kBlock), 0, s, ngroups, P, start_pg, start_gp, m->tp_blk, m->tp_panel_ptr, m->tp_group_ptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;  // gstart (host) is done with
        m->tp_padded = padded;
    } while (0);
    for (int32_t* p : {cnt_pg, cnt_gp, start_pg, start_gp})
        if (p) (void)hipFree(p);
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
    m->tp_bytes     = m->tp_padded * 20 + m->tp_padded / kTpLine * 4 + (int64_t)(P + 1 + 2 * (ngroups + 1)) * 4;
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
    if (!A->tp_val || !A->tp_col || !A->tp_row || !A->tp_xg || !A->tp_blk || !A->tp_panel_ptr || !A->tp_group_ptr || !A->tp_gstart || !x || !y ||
        A->tp_panels <= 0 || A->tp_pcols <= 0 || A->tp_pcols > kTpPanelCols || A->tp_max_rows > kTpGroupRows || A->tp_padded % kTpLine != 0)
        SPMV_FAIL(SPMV_ERR_INVALID, "two-phase kernel selected but its layout was not built");
    static std::atomic<unsigned long long> granted{0};  // bit per device
    if (!((granted.load(std::memory_order_relaxed) >> ctx->device) & 1ull))
    {
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_expand_kernel<1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160008));
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_expand_kernel<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160008));
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_expand_kernel<512, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 80008));
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_expand_kernel<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80008));
        SPMV_HIP(hipFuncSetAttribute((const void*)tp_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160008));
        granted.fetch_or(1ull << ctx->device, std::memory_order_relaxed);
    }
    // a panel of at most 10000 columns: two workgroups of 512 share a CU
    const bool   half  = A->tp_pcols * 2 <= kTpPanelCols;
    const int    unr   = A->tp_unroll == 4 ? 4 : 6;
    const size_t xlds  = sizeof(double) * (size_t)A->tp_pcols;
    const dim3   egrid((unsigned)std::min(A->tp_panels, half ? 2 * kNumCu : kNumCu));
    static const int rotate = [] { const char* e = getenv("SPMV_TP_ROTATE"); return e ? atoi(e) : 1; }();  // (0: A/B)
#define SPMV_TP_EXPAND(T, U)                                                                                                                  \
    hipLaunchKernelGGL((tp_expand_kernel<T, U>), egrid, dim3(T), xlds, ctx->stream, A->tp_panels, A->tp_pcols, A->ncol, A->tp_panel_ptr,       \
                       (const unsigned short*)A->tp_col, (const double*)A->tp_val, (const int32_t*)A->tp_blk, x, A->tp_xg, rotate)
    if (half && unr == 6)
        SPMV_TP_EXPAND(512, 6);
    else if (half)
        SPMV_TP_EXPAND(512, 4);
    else if (unr == 6)
        SPMV_TP_EXPAND(1024, 6);
    else
        SPMV_TP_EXPAND(1024, 4);
#undef SPMV_TP_EXPAND
    hipLaunchKernelGGL(tp_reduce_kernel, dim3((unsigned)std::min(A->tp_ngroups, kNumCu)), dim3(kTpThreads), sizeof(double) * (size_t)A->tp_max_rows,
                       ctx->stream, A->tp_gstart, A->tp_ngroups, A->tp_group_ptr, (const unsigned short*)A->tp_row, (const double*)A->tp_xg, y,
                       ex.overwrite ? 1 : 0, ex.dot_w, ex.dot_out);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
