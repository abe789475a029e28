// symgs.hip — symmetric Gauss-Seidel on a CSR handle: the sweep the reference's `diagonal // for SymGS` fields
// (include/matrix.h:36,81) were reserved for and that the reference never wrote.  There is no
// reference code to follow; the definition is the textbook one in the matrix's own row order (oracle/spmv_oracle.c:
// orc_symgs), so that results can be compared number by number:
//
//   forward   for i = 0 .. n-1:   x_i = (b_i - sum_{j<i} a_ij x_j(new) - sum_{j>i} a_ij x_j(old)) / a_ii
//   backward  for i = n-1 .. 0:   x_i = (b_i - sum_{j<i} a_ij x_j(old) - sum_{j>i} a_ij x_j(new)) / a_ii
//
// and, because that order is only as parallel as the matrix lets it be, the same sweep over a MULTICOLOUR order of
// the rows (the default; spmv_mat_set_param "symgs_order" = 0 asks for the matrix's own order): rows are coloured so
// that no two coupled rows share a colour — the greedy colouring in row order, colour(i) = smallest colour no coupled
// row j < i has, found on the device by relaxing to its fixed point — and swept colour by colour, ascending row index
// inside a colour (spmv_symgs_order returns the sequence; orc_symgs takes it).  Two colours for a 7-point Laplacian
// (red-black), i.e. 6 launches per sweep instead of ~790; a different but equally valid Gauss-Seidel preconditioner.
//
// A sweep in a given order is a triangular solve, and a triangular solve is only as parallel as the order lets it be.
// The matrix is split once into the part L before the diagonal IN SWEEP ORDER, the diagonal D and the part U after it,
// and each half sweep becomes
//   t = b - U x          (a row-parallel product, fully parallel, reads the OLD x only)
//   (L + D) x = t        (rows in LEVELS: level(i) = 1 + max level(j) over the entries j of row i in L; the rows of one
//                         level depend on earlier levels only and are solved together)
// and the mirror image for the backward half.  The split keeps the sweep exact for any pattern: a row never reads an
// entry of x that another row of its level is writing (with the whole row in one kernel that holds for structurally
// symmetric matrices only).  Levels are found on the device by relaxing level(i) = max(level(j) + 1) to its fixed
// point, rows are ordered by level with a stable radix sort (ascending row index inside a level: neighbouring lanes
// read neighbouring rows), and the schedule — one launch per level, runs of small levels folded into one launch of a
// single workgroup that steps through them with barriers — is fixed at set-up.  The cost is the dependency chain: a
// 7-point Laplacian on 160^3 points has 478 levels each way, i.e. ~0.7k launches of a few microseconds per sweep.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "wave.hpp"

namespace spmv
{
struct tri_part
{
    int32_t* ptr     = nullptr;  // [n + 1]
    int32_t* col     = nullptr;  // [nnz]
    double*  val     = nullptr;  // [nnz]
    int32_t* order   = nullptr;  // [n] rows by level, ascending inside a level
    int32_t* lvl_ptr = nullptr;  // [levels + 1] into order
    int64_t  nnz     = 0;
    int32_t  levels  = 0;
    int32_t  lanes   = 1;  // lanes per row in the solve and product kernels
    struct segment
    {
        int32_t first_level, nlevels, first_row, rows;  // nlevels > 1: one workgroup steps through them
    };
    std::vector<segment> schedule;
};

struct symgs_plan
{
    tri_part lo, up;
    double*  diag  = nullptr;  // [n]
    double*  t     = nullptr;  // [n] right-hand side of the triangular solves
    int32_t* seq   = nullptr;  // [n] multicolour order: the k-th row of a forward sweep (null: the matrix's own order)
    int32_t  mode    = 0;      // 0 the matrix's own order, 1 multicolour
    int32_t  colours = 0;
    int64_t  bytes = 0;
    // proper colourings (no two coupled rows share a colour: every level is a colour) sweep from a copy of the WHOLE rows in
    // sweep order - rows of a colour contiguous, columns and vectors in the matrix's own numbering - one launch per colour
    int32_t*             cs_ptr = nullptr;  // [n + 1]
    int32_t*             cs_col = nullptr;  // [nnz]
    double*              cs_val = nullptr;  // [nnz]
    std::vector<int32_t> cs_first;          // [colours + 1] first sweep position of every colour
    int32_t              cs_lanes = 4;
};

namespace
{
constexpr int kSolveThreads = 1024;
constexpr int kSmallLevel   = 4096;  // lanes: levels up to this many (rows x lanes per row) are folded into one workgroup

// ---- split A = L + D + U ------------------------------------------------------------------------------------------
// pos: position of every row in the sweep (null: the row index itself); an entry belongs to L if its column comes
// earlier in the sweep than its row
__global__ __launch_bounds__(kBlock) void split_count_kernel(int n, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                             const double* __restrict__ val, const int32_t* __restrict__ pos,
                                                             int32_t* __restrict__ lo_cnt, int32_t* __restrict__ up_cnt,
                                                             double* __restrict__ diag, int* __restrict__ flag)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    if (i == n)
    {
        lo_cnt[n] = 0;
        up_cnt[n] = 0;
        return;
    }
    int       nl = 0, nu = 0;
    double    d  = 0.0;
    const int pi = pos ? pos[i] : i;
    for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j)
    {
        const int c = col[j];
        if (c == i)
            d += val[j];  // duplicates of the diagonal entry are summed, as the product would
        else if ((pos ? pos[c] : c) < pi)
            ++nl;
        else
            ++nu;
    }
    lo_cnt[i] = nl;
    up_cnt[i] = nu;
    diag[i]   = d;
    if (d == 0.0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(kBlock) void split_fill_kernel(int n, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                            const double* __restrict__ val, const int32_t* __restrict__ pos,
                                                            const int32_t* __restrict__ lo_ptr, const int32_t* __restrict__ up_ptr,
                                                            int32_t* __restrict__ lo_col, double* __restrict__ lo_val,
                                                            int32_t* __restrict__ up_col, double* __restrict__ up_val)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int       pl = lo_ptr[i], pu = up_ptr[i];
    const int pi = pos ? pos[i] : i;
    for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j)  // the order inside the row is kept
    {
        const int c = col[j];
        if (c == i) continue;
        if ((pos ? pos[c] : c) < pi)
        {
            lo_col[pl] = c;
            lo_val[pl] = val[j];
            ++pl;
        }
        else
        {
            up_col[pu] = c;
            up_val[pu] = val[j];
            ++pu;
        }
    }
}

// ---- multicolour order ----------------------------------------------------------------------------------------------
// Greedy colouring in row order by relaxation: colour(i) = smallest colour none of the rows j < i coupled to i has.
// A row is final once the rows before it are (induction over the dependency levels), final rows never change again,
// and a pass that changes nothing is the fixed point — the same colours a sequential greedy pass would give.  Rows
// j > i coupled to i avoid colour(i) in their own turn when the pattern is symmetric; where it is not, two coupled
// rows may share a colour and the level analysis below simply keeps them apart.
__global__ __launch_bounds__(kBlock) void colour_relax_kernel(int n, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                              int32_t* colour, int* __restrict__ changed)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int b = row_ptr[i], e = row_ptr[i + 1];
    int       m = 0;
    for (int base = 0;; base += 64)  // (ends: a row has fewer coupled rows than colours tried)
    {
        unsigned long long used = 0ull;
        for (int j = b; j < e; ++j)
        {
            const int c = col[j];
            if (c >= i) continue;
            const int cj = __builtin_nontemporal_load(colour + c) - base;
            if (cj >= 0 && cj < 64) used |= 1ull << cj;
        }
        if (~used)
        {
            m = base + __builtin_ctzll(~used);
            break;
        }
    }
    if (m != colour[i])
    {
        colour[i] = m;
        *changed  = 1;
    }
}

__global__ __launch_bounds__(kBlock) void invert_order_kernel(int n, const int32_t* __restrict__ seq, int32_t* __restrict__ pos)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n) pos[seq[k]] = k;
}

// ---- levels -----------------------------------------------------------------------------------------------------
// One relaxation pass, in place: values only grow and never pass the true level, so any interleaving of the lanes
// ends at the same fixed point; a pass that changes nothing has reached it.
__global__ __launch_bounds__(kBlock) void level_relax_kernel(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                             int32_t* lev, int* __restrict__ changed)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int m = 0;
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) m = max(m, __builtin_nontemporal_load(lev + col[j]) + 1);
    if (m > lev[i])
    {
        lev[i]   = m;
        *changed = 1;
    }
}

__global__ __launch_bounds__(kBlock) void level_hist_kernel(int n, const int32_t* __restrict__ lev, int32_t* __restrict__ hist, int nlevels)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n && lev[i] < nlevels) atomicAdd(hist + lev[i], 1);
}

__global__ __launch_bounds__(kBlock) void level_max_kernel(int n, const int32_t* __restrict__ lev, int32_t* __restrict__ out)
{
    int m = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) m = max(m, lev[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ---- the sweep ----------------------------------------------------------------------------------------------------
// t = b - T x over the rows of a triangle, LANES lanes per row
template <int LANES>
__global__ __launch_bounds__(kBlock) void tri_residual_kernel(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                              const double* __restrict__ val, const double* __restrict__ b,
                                                              const double* __restrict__ x, double* __restrict__ t)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int i = gid / LANES, l = gid % LANES;
    double    acc = 0.0;
    if (i < n)
        for (int j = ptr[i] + l; j < ptr[i + 1]; j += LANES) acc = fma(val[j], x[col[j]], acc);
    acc = group_sum<LANES, true>(acc);
    if (i < n && l == 0) t[i] = b[i] - acc;
}

// ---- the sweep of a proper colouring: one launch per colour, the residual fused into the solve --------------------------------
// Rows of a colour are not coupled to each other, so x_i = (b_i - sum_{j != i} a_ij x_j) / a_ii for all of them at once IS the
// Gauss-Seidel update in sweep order (every x_j it reads belongs to another colour: already updated in this half sweep if
// that colour came earlier, still the old value if it comes later).  The rows are read from a copy in sweep order (cs_*:
// contiguous per colour), the vectors stay in the matrix's numbering (x gathers by column as the product does, the result goes
// to x[seq[k]]): one pass over a colour's rows per launch, against a product over one triangle plus a solve over the other in
// the general scheme.  Round 3 stored the TRIANGLES in sweep order and permuted the vectors on entry and exit: the
// permutations cost what the contiguous rows saved (profiles/r03_tune_symgs_sweep_order_storage.txt); here nothing is permuted.
// flag |= 1 if two coupled rows share a colour (any stored entry (i, c), c != i, with colour[i] == colour[c])
__global__ __launch_bounds__(kBlock) void gs_proper_kernel(int n, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ colour, int* __restrict__ flag)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int ci  = colour[i];
    bool      bad = false;
    for (int j = row_ptr[i]; j < row_ptr[i + 1]; ++j) bad = bad || (col[j] != i && colour[col[j]] == ci);
    if (bad) atomicOr(flag, 1);
}
__global__ __launch_bounds__(kBlock) void gs_row_lengths_kernel(int n, const int32_t* __restrict__ seq, const int32_t* __restrict__ row_ptr,
                                                                int32_t* __restrict__ len)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k > n) return;
    len[k] = k < n ? row_ptr[seq[k] + 1] - row_ptr[seq[k]] : 0;
}
__global__ __launch_bounds__(kBlock) void gs_copy_rows_kernel(int n, const int32_t* __restrict__ seq, const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ col, const double* __restrict__ val,
                                                              const int32_t* __restrict__ cs_ptr, int32_t* __restrict__ cs_col,
                                                              double* __restrict__ cs_val)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int k = gid / 4, l = gid % 4;  // four lanes per row
    if (k >= n) return;
    const int i = seq[k], src = row_ptr[i], dst = cs_ptr[k], len = row_ptr[i + 1] - src;
    for (int j = l; j < len; j += 4)
    {
        cs_col[dst + j] = col[src + j];
        cs_val[dst + j] = val[src + j];
    }
}
template <int LANES>
__global__ __launch_bounds__(kBlock) void gs_colour_kernel(int first, int rows, const int32_t* __restrict__ seq, const int32_t* __restrict__ ptr,
                                                           const int32_t* __restrict__ col, const double* __restrict__ val,
                                                           const double* __restrict__ diag, const double* __restrict__ b, double* x)
{
    const int  gid = blockIdx.x * kBlock + threadIdx.x;
    const int  r = gid / LANES, l = gid % LANES;
    const bool on = r < rows;
    const int  k = first + (on ? r : 0), i = seq[k];
    double     acc = 0.0;
    if (on)
        for (int j = ptr[k] + l; j < ptr[k + 1]; j += LANES)
        {
            const int c = col[j];
            if (c != i) acc = fma(val[j], x[c], acc);  // (x[c] is never written by this launch: c belongs to another colour)
        }
    acc = group_sum<LANES, true>(acc);
    if (on && l == 0) x[i] = (b[i] - acc) / diag[i];
}

template <int LANES>
__device__ __forceinline__ void solve_row(int i, int l, bool on, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                          const double* __restrict__ val, const double* __restrict__ diag, const double* t, double* x)
{
    double acc = 0.0;
    if (on)
        for (int j = ptr[i] + l; j < ptr[i + 1]; j += LANES) acc = fma(val[j], x[col[j]], acc);
    acc = group_sum<LANES, true>(acc);
    if (on && l == 0) x[i] = (t[i] - acc) / diag[i];
}

// one level: rows order[first .. first + rows)
template <int LANES>
__global__ __launch_bounds__(kBlock) void tri_solve_level_kernel(int first, int rows, const int32_t* __restrict__ order,
                                                                 const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                                 const double* __restrict__ val, const double* __restrict__ diag,
                                                                 const double* t, double* x)
{
    const int  gid = blockIdx.x * kBlock + threadIdx.x;
    const int  r = gid / LANES, l = gid % LANES;
    const bool on = r < rows;
    solve_row<LANES>(on ? order[first + r] : 0, l, on, ptr, col, val, diag, t, x);
}

// a run of small levels in one workgroup: what a level wrote is read by the next after the barrier (same CU, same L1)
template <int LANES>
__global__ __launch_bounds__(kSolveThreads) void tri_solve_run_kernel(int first_level, int nlevels, const int32_t* __restrict__ lvl_ptr,
                                                                      const int32_t* __restrict__ order, const int32_t* __restrict__ ptr,
                                                                      const int32_t* __restrict__ col, const double* __restrict__ val,
                                                                      const double* __restrict__ diag, const double* t, double* x)
{
    const int r0 = threadIdx.x / LANES, l = threadIdx.x % LANES;
    for (int lv = first_level; lv < first_level + nlevels; ++lv)
    {
        const int first = lvl_ptr[lv], rows = lvl_ptr[lv + 1] - first;
        for (int base = 0; base < rows; base += kSolveThreads / LANES)  // (uniform bounds: every lane reaches the barrier)
        {
            const int  r  = base + r0;
            const bool on = r < rows;
            solve_row<LANES>(on ? order[first + r] : 0, l, on, ptr, col, val, diag, t, x);
        }
        __syncthreads();
    }
}

void free_part(tri_part& p)
{
    for (void** q : {(void**)&p.ptr, (void**)&p.col, (void**)&p.val, (void**)&p.order, (void**)&p.lvl_ptr})
        if (*q)
        {
            (void)hipFree(*q);
            *q = nullptr;
        }
    p.schedule.clear();
}

// levels of one triangle, rows by level, the launch schedule
int analyse_part(spmv_ctx* ctx, int n, tri_part& p, const char* which)
{
    hipStream_t s    = ctx->stream;
    int32_t *   lev  = nullptr, *hist = nullptr;
    int         rc   = SPMV_OK;
    const unsigned grid = (unsigned)ceil_div(n, kBlock);
    SPMV_TRY(ensure_scratch(ctx, 64));
    int* flag = (int*)ctx->scratch;
    do
    {
        if (hipMalloc(&lev, sizeof(int32_t) * (size_t)n) != hipSuccess || hipMalloc(&p.order, sizeof(int32_t) * (size_t)n) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        if (hipMemsetAsync(lev, 0, sizeof(int32_t) * (size_t)n, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        // relax to the fixed point: `kPasses` passes between two looks at the flag; as many passes as the longest chain
        constexpr int kPasses = 8, kMaxRounds = 1 << 15;
        int  round = 0, h_flag = 1;
        for (; round < kMaxRounds && h_flag; ++round)
        {
            (void)hipMemsetAsync(flag, 0, sizeof(int), s);
            for (int k = 0; k < kPasses; ++k) hipLaunchKernelGGL(level_relax_kernel, dim3(grid), dim3(kBlock), 0, s, n, p.ptr, p.col, lev, flag);
            if (hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
            // the flag of the LAST pass alone would do; any pass of the round is a safe over-estimate
        }
        if (rc != SPMV_OK) break;
        if (h_flag)
        {
            set_error("spmv_symgs: the %s triangle has dependency chains longer than %d rows: a sweep in row order is sequential there",
                      which, kPasses * kMaxRounds);
            rc = SPMV_ERR_UNSUPPORTED;
            break;
        }
        int32_t h_max = 0;
        (void)hipMemsetAsync(flag, 0, sizeof(int), s);
        hipLaunchKernelGGL(level_max_kernel, dim3((unsigned)std::min<int64_t>(1024, grid)), dim3(kBlock), 0, s, n, lev, (int32_t*)flag);
        if (hipMemcpyAsync(&h_max, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        const int levels = h_max + 1;
        p.levels         = levels;
        if (hipMalloc(&hist, sizeof(int32_t) * ((size_t)levels + 1)) != hipSuccess || hipMalloc(&p.lvl_ptr, sizeof(int32_t) * ((size_t)levels + 1)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        (void)hipMemsetAsync(hist, 0, sizeof(int32_t) * ((size_t)levels + 1), s);
        hipLaunchKernelGGL(level_hist_kernel, dim3(grid), dim3(kBlock), 0, s, n, lev, hist, levels);
        if ((rc = exclusive_scan_i32(ctx, hist, p.lvl_ptr, (int64_t)levels + 1)) != SPMV_OK) break;
        int bits = 1;
        while (bits < 31 && (1LL << bits) < (long long)levels) ++bits;
        if ((rc = sort_ids_by_key(ctx, lev, n, bits, p.order)) != SPMV_OK) break;
        std::vector<int32_t> lp((size_t)levels + 1);
        if (hipMemcpyAsync(lp.data(), p.lvl_ptr, sizeof(int32_t) * lp.size(), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (lp[0] != 0 || lp[(size_t)levels] != n)
        {
            set_error("spmv_symgs: the level table of the %s triangle does not cover the rows (%d of %d)", which, lp[(size_t)levels], n);
            rc = SPMV_ERR_HIP;
            break;
        }
        const double avg = n > 0 ? (double)p.nnz / n : 0.0;
        p.lanes          = avg <= 2.5 ? 1 : (avg <= 12.0 ? 4 : 16);
        // schedule: runs of small levels share one workgroup, every other level is a launch of its own
        p.schedule.clear();
        for (int lv = 0; lv < levels;)
        {
            const int rows = lp[(size_t)lv + 1] - lp[(size_t)lv];
            if ((int64_t)rows * p.lanes > kSmallLevel)
            {
                p.schedule.push_back({lv, 1, lp[(size_t)lv], rows});
                ++lv;
                continue;
            }
            int end = lv;
            while (end < levels && (int64_t)(lp[(size_t)end + 1] - lp[(size_t)end]) * p.lanes <= kSmallLevel) ++end;
            p.schedule.push_back({lv, end - lv, lp[(size_t)lv], lp[(size_t)end] - lp[(size_t)lv]});
            lv = end;
        }
    } while (0);
    (void)hipStreamSynchronize(s);
    if (lev) (void)hipFree(lev);
    if (hist) (void)hipFree(hist);
    return rc;
}

template <int LANES>
void launch_solve(hipStream_t s, const tri_part& p, const double* diag, const double* t, double* x)
{
    for (const tri_part::segment& g : p.schedule)
    {
        if (g.rows == 0) continue;
        if (g.nlevels > 1 || (int64_t)g.rows * LANES <= kSolveThreads)
            hipLaunchKernelGGL(tri_solve_run_kernel<LANES>, dim3(1), dim3(kSolveThreads), 0, s, g.first_level, g.nlevels, p.lvl_ptr, p.order, p.ptr,
                               p.col, p.val, diag, t, x);
        else
            hipLaunchKernelGGL(tri_solve_level_kernel<LANES>, dim3((unsigned)ceil_div((int64_t)g.rows * LANES, kBlock)), dim3(kBlock), 0, s,
                               g.first_row, g.rows, p.order, p.ptr, p.col, p.val, diag, t, x);
    }
}
void solve(hipStream_t s, const tri_part& p, const double* diag, const double* t, double* x)
{
    if (p.lanes == 1)
        launch_solve<1>(s, p, diag, t, x);
    else if (p.lanes == 4)
        launch_solve<4>(s, p, diag, t, x);
    else
        launch_solve<16>(s, p, diag, t, x);
}
void residual(hipStream_t s, int n, const tri_part& p, const double* b, const double* x, double* t)
{
    const unsigned grid = (unsigned)ceil_div((int64_t)n * p.lanes, kBlock);
    if (p.lanes == 1)
        hipLaunchKernelGGL(tri_residual_kernel<1>, dim3(grid), dim3(kBlock), 0, s, n, p.ptr, p.col, p.val, b, x, t);
    else if (p.lanes == 4)
        hipLaunchKernelGGL(tri_residual_kernel<4>, dim3(grid), dim3(kBlock), 0, s, n, p.ptr, p.col, p.val, b, x, t);
    else
        hipLaunchKernelGGL(tri_residual_kernel<16>, dim3(grid), dim3(kBlock), 0, s, n, p.ptr, p.col, p.val, b, x, t);
}
}  // namespace

void symgs_free(spmv_mat* m)
{
    if (!m->gs) return;
    free_part(m->gs->lo);
    free_part(m->gs->up);
    if (m->gs->diag) (void)hipFree(m->gs->diag);
    if (m->gs->t) (void)hipFree(m->gs->t);
    if (m->gs->seq) (void)hipFree(m->gs->seq);
    for (void* q : {(void*)m->gs->cs_ptr, (void*)m->gs->cs_col, (void*)m->gs->cs_val})
        if (q) (void)hipFree(q);
    m->device_bytes -= m->gs->bytes;
    delete m->gs;
    m->gs = nullptr;
}

int symgs_setup(spmv_mat* m)
{
    if (m->gs && m->gs->mode == (m->gs_order != 0 ? 1 : 0)) return SPMV_OK;
    symgs_free(m);  // (another order was asked for since)
    spmv_ctx* ctx = m->ctx;
    SPMV_REQUIRE(m->format == SPMV_FMT_CSR, "spmv_symgs: a CSR handle is needed (format %d)", m->format);
    SPMV_REQUIRE(m->nrow == m->ncol && m->row_begin == 0, "spmv_symgs: the whole square matrix is needed (%d x %d, first row %lld)", m->nrow,
                 m->ncol, (long long)m->row_begin);
    SPMV_REQUIRE(m->nnz == 0 || (m->b && m->v), "spmv_symgs: the CSR arrays are gone (panel_keep_csr = 0 released them)");
    const int   n = m->nrow;
    hipStream_t s = ctx->stream;
    SPMV_TRY(ensure_scratch(ctx, 64));  // (before the plan is attached: a failure here leaves no half-built plan behind)
    symgs_plan* g = new symgs_plan();
    g->mode       = m->gs_order != 0 ? 1 : 0;
    m->gs         = g;
    if (n == 0) return SPMV_OK;
    int32_t *lo_cnt = nullptr, *up_cnt = nullptr, *colour = nullptr, *pos = nullptr;
    int      rc     = SPMV_OK;
    int* flag = (int*)ctx->scratch;
    do
    {
        if (g->mode == 1)
        {
            if (hipMalloc(&colour, sizeof(int32_t) * (size_t)n) != hipSuccess || hipMalloc(&pos, sizeof(int32_t) * (size_t)n) != hipSuccess ||
                hipMalloc(&g->seq, sizeof(int32_t) * (size_t)n) != hipSuccess)
            {
                rc = SPMV_ERR_ALLOC;
                break;
            }
            (void)hipMemsetAsync(colour, 0, sizeof(int32_t) * (size_t)n, s);
            constexpr int kPasses = 8, kMaxRounds = 1 << 15;
            const unsigned grid = (unsigned)ceil_div(n, kBlock);
            int  h_changed = 1, round = 0;
            for (; round < kMaxRounds && h_changed; ++round)
            {
                (void)hipMemsetAsync(flag, 0, sizeof(int), s);
                for (int k = 0; k < kPasses; ++k) hipLaunchKernelGGL(colour_relax_kernel, dim3(grid), dim3(kBlock), 0, s, n, m->a, m->b, colour, flag);
                if (hipMemcpyAsync(&h_changed, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
                {
                    rc = SPMV_ERR_HIP;
                    break;
                }
            }
            if (rc != SPMV_OK) break;
            if (h_changed)
            {
                set_error("spmv_symgs: colouring did not settle in %d passes (dependency chains that long); symgs_order = 0 will not either",
                          kPasses * kMaxRounds);
                rc = SPMV_ERR_UNSUPPORTED;
                break;
            }
            int32_t h_max = 0;
            (void)hipMemsetAsync(flag, 0, sizeof(int), s);
            hipLaunchKernelGGL(level_max_kernel, dim3((unsigned)std::min<int64_t>(1024, grid)), dim3(kBlock), 0, s, n, colour, (int32_t*)flag);
            if (hipMemcpyAsync(&h_max, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
            g->colours = h_max + 1;
            int bits   = 1;
            while (bits < 31 && (1LL << bits) < (long long)g->colours) ++bits;
            if ((rc = sort_ids_by_key(ctx, colour, n, bits, g->seq)) != SPMV_OK) break;  // by colour, ascending row inside
            hipLaunchKernelGGL(invert_order_kernel, dim3(grid), dim3(kBlock), 0, s, n, g->seq, pos);
        }
        if (hipMalloc(&lo_cnt, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess || hipMalloc(&up_cnt, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess ||
            hipMalloc(&g->lo.ptr, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess || hipMalloc(&g->up.ptr, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess ||
            hipMalloc(&g->diag, sizeof(double) * (size_t)n) != hipSuccess || hipMalloc(&g->t, sizeof(double) * (size_t)n) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        (void)hipMemsetAsync(flag, 0, sizeof(int), s);
        hipLaunchKernelGGL(split_count_kernel, dim3((unsigned)ceil_div((int64_t)n + 1, kBlock)), dim3(kBlock), 0, s, n, m->a, m->b, m->v, pos, lo_cnt,
                           up_cnt, g->diag, flag);
        int h_flag = 0;  // (read before the scans: they use the context's scratch too)
        if (hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (h_flag)
        {
            set_error("spmv_symgs: the matrix has a zero or missing diagonal entry");
            rc = SPMV_ERR_INVALID;
            break;
        }
        if ((rc = exclusive_scan_i32(ctx, lo_cnt, g->lo.ptr, (int64_t)n + 1)) != SPMV_OK) break;
        if ((rc = exclusive_scan_i32(ctx, up_cnt, g->up.ptr, (int64_t)n + 1)) != SPMV_OK) break;
        int32_t h_lo = 0, h_up = 0;
        if (hipMemcpyAsync(&h_lo, g->lo.ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&h_up, g->up.ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (h_lo < 0 || h_up < 0 || (int64_t)h_lo + h_up > m->nnz)
        {
            set_error("spmv_symgs: splitting the matrix gave %d + %d entries of %lld", h_lo, h_up, (long long)m->nnz);
            rc = SPMV_ERR_HIP;
            break;
        }
        g->lo.nnz = h_lo;
        g->up.nnz = h_up;
        if (hipMalloc(&g->lo.col, sizeof(int32_t) * std::max<size_t>(1, (size_t)h_lo)) != hipSuccess ||
            hipMalloc(&g->lo.val, sizeof(double) * std::max<size_t>(1, (size_t)h_lo)) != hipSuccess ||
            hipMalloc(&g->up.col, sizeof(int32_t) * std::max<size_t>(1, (size_t)h_up)) != hipSuccess ||
            hipMalloc(&g->up.val, sizeof(double) * std::max<size_t>(1, (size_t)h_up)) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        hipLaunchKernelGGL(split_fill_kernel, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, s, n, m->a, m->b, m->v, pos, g->lo.ptr,
                           g->up.ptr, g->lo.col, g->lo.val, g->up.col, g->up.val);
        if (hipGetLastError() != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if ((rc = analyse_part(ctx, n, g->lo, "lower")) != SPMV_OK) break;
        if ((rc = analyse_part(ctx, n, g->up, "upper")) != SPMV_OK) break;
        // A proper colouring (every level of either triangle is one colour): the fused sweep's copy of the rows in sweep order.
        // SPMV_GS_FUSED=0 keeps the general scheme (A/B; read here, at set-up).
        const char* e_fused = getenv("SPMV_GS_FUSED");
        int h_improper = 1;
        if (g->mode == 1 && g->colours > 0 && g->lo.levels == g->colours && g->up.levels == g->colours && !(e_fused && e_fused[0] == '0') &&
            m->nnz < INT32_MAX)
        {
            // (equal counts of levels and colours are necessary, not sufficient: the colouring itself is checked, entry by entry)
            (void)hipMemsetAsync(flag, 0, sizeof(int), s);
            hipLaunchKernelGGL(gs_proper_kernel, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, s, n, m->a, m->b, colour, flag);
            if (hipMemcpyAsync(&h_improper, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
        }
        if (!h_improper)
        {
            g->cs_first.assign((size_t)g->colours + 1, 0);
            int32_t* len = nullptr;
            if (hipMalloc(&len, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess || hipMalloc(&g->cs_ptr, sizeof(int32_t) * ((size_t)n + 1)) != hipSuccess ||
                hipMalloc(&g->cs_col, sizeof(int32_t) * std::max<size_t>(1, (size_t)m->nnz)) != hipSuccess ||
                hipMalloc(&g->cs_val, sizeof(double) * std::max<size_t>(1, (size_t)m->nnz)) != hipSuccess)
            {
                if (len) (void)hipFree(len);
                rc = SPMV_ERR_ALLOC;
                break;
            }
            hipLaunchKernelGGL(gs_row_lengths_kernel, dim3((unsigned)ceil_div((int64_t)n + 1, kBlock)), dim3(kBlock), 0, s, n, g->seq, m->a, len);
            rc = exclusive_scan_i32(ctx, len, g->cs_ptr, (int64_t)n + 1);
            if (rc == SPMV_OK)
            {
                hipLaunchKernelGGL(gs_copy_rows_kernel, dim3((unsigned)ceil_div((int64_t)n * 4, kBlock)), dim3(kBlock), 0, s, n, g->seq, m->a, m->b,
                                   m->v, g->cs_ptr, g->cs_col, g->cs_val);
                // the forward triangle's levels ARE the colours (ascending row index inside): its level pointers are the colours' first positions
                if (hipMemcpyAsync(g->cs_first.data(), g->lo.lvl_ptr, sizeof(int32_t) * ((size_t)g->colours + 1), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess)
                    rc = SPMV_ERR_HIP;
            }
            (void)hipFree(len);
            if (rc != SPMV_OK) break;
            if (g->cs_first.front() != 0 || g->cs_first.back() != n) g->cs_first.clear();  // (never: the general scheme then)
            g->cs_lanes = m->nnz / std::max(1, n) >= 24 ? 16 : (m->nnz / std::max(1, n) >= 3 ? 4 : 1);
        }
    } while (0);
    (void)hipStreamSynchronize(s);
    for (int32_t* q : {lo_cnt, up_cnt, colour, pos})
        if (q) (void)hipFree(q);
    if (rc != SPMV_OK)
    {
        symgs_free(m);
        if (rc == SPMV_ERR_ALLOC) set_error("spmv_symgs: out of device memory splitting a matrix of %lld entries", (long long)m->nnz);
        if (rc == SPMV_ERR_HIP && hipGetLastError() != hipSuccess) set_error("spmv_symgs: set-up failed: %s", hipGetErrorString(hipGetLastError()));
        return rc;
    }
    g->bytes = (g->lo.nnz + g->up.nnz) * 12 + (int64_t)n * (8 + 8 + 4 + 4 + 8 + 8 + (g->seq ? 4 : 0)) + (int64_t)(g->lo.levels + g->up.levels) * 4 +
               (g->cs_ptr ? m->nnz * 12 + ((int64_t)n + 1) * 4 : 0);
    m->device_bytes += g->bytes;
    return SPMV_OK;
}

// One symmetric sweep on x.  zero_guess: x is taken as 0 on entry (its contents are ignored): the forward half then
// needs no product, which is how the sweep is used as a preconditioner z = M^-1 r.
int symgs_sweep(spmv_ctx* ctx, const spmv_mat* A, const double* b, double* x, bool zero_guess)
{
    const symgs_plan* g = A->gs;
    if (!g) SPMV_FAIL(SPMV_ERR_INVALID, "spmv_symgs: the handle was not set up");
    const int n = A->nrow;
    if (n == 0) return SPMV_OK;
    if (!g->lo.ptr || !g->up.ptr || !g->lo.order || !g->up.order || !g->lo.lvl_ptr || !g->up.lvl_ptr || !g->diag || !g->t || !b || !x)
        SPMV_FAIL(SPMV_ERR_INVALID, "spmv_symgs: the plan of this handle is incomplete");
    hipStream_t s = ctx->stream;
    if (g->cs_ptr && g->seq && !g->cs_first.empty())
    {
        // a proper colouring: forward through the colours, backward from the last but one (the last colour's rows would be
        // recomputed from the very same values)
        if (zero_guess) SPMV_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, s));
        auto colour = [&](int c) {
            const int first = g->cs_first[(size_t)c], rows = g->cs_first[(size_t)c + 1] - first;
            if (rows <= 0) return;
            const unsigned grid = (unsigned)ceil_div((int64_t)rows * g->cs_lanes, kBlock);
            if (g->cs_lanes == 1)
                hipLaunchKernelGGL(gs_colour_kernel<1>, dim3(grid), dim3(kBlock), 0, s, first, rows, g->seq, g->cs_ptr, g->cs_col, g->cs_val, g->diag, b, x);
            else if (g->cs_lanes == 4)
                hipLaunchKernelGGL(gs_colour_kernel<4>, dim3(grid), dim3(kBlock), 0, s, first, rows, g->seq, g->cs_ptr, g->cs_col, g->cs_val, g->diag, b, x);
            else
                hipLaunchKernelGGL(gs_colour_kernel<16>, dim3(grid), dim3(kBlock), 0, s, first, rows, g->seq, g->cs_ptr, g->cs_col, g->cs_val, g->diag, b, x);
        };
        for (int c = 0; c < g->colours; ++c) colour(c);
        for (int c = g->colours - 2; c >= 0; --c) colour(c);
        SPMV_HIP(hipGetLastError());
        return SPMV_OK;
    }
    if (zero_guess)
        solve(s, g->lo, g->diag, b, x);  // t = b - U 0
    else
    {
        residual(s, n, g->up, b, x, g->t);
        solve(s, g->lo, g->diag, g->t, x);
    }
    residual(s, n, g->lo, b, x, g->t);
    solve(s, g->up, g->diag, g->t, x);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}

// the k-th row of a forward sweep, to the host
int symgs_sequence(const spmv_mat* m, int32_t* out)
{
    const symgs_plan* g = m->gs;
    if (!g) SPMV_FAIL(SPMV_ERR_INVALID, "spmv_symgs_order: the handle was not set up (spmv_symgs_setup)");
    if (!g->seq)
    {
        for (int32_t i = 0; i < m->nrow; ++i) out[i] = i;
        return SPMV_OK;
    }
    SPMV_HIP(hipMemcpyAsync(out, g->seq, sizeof(int32_t) * (size_t)m->nrow, hipMemcpyDeviceToHost, m->ctx->stream));
    SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
    return SPMV_OK;
}

int symgs_info(const spmv_mat* m, const char* what, int64_t* value)
{
    const symgs_plan* g = m->gs;
    if (!strcmp(what, "symgs_levels_forward"))
        *value = g ? g->lo.levels : 0;
    else if (!strcmp(what, "symgs_levels_backward"))
        *value = g ? g->up.levels : 0;
    else if (!strcmp(what, "symgs_launches"))  // per sweep from a non-zero x: two products and the two schedules; fused: 2 colours - 1
        *value = g ? (g->cs_ptr && !g->cs_first.empty() ? (int64_t)(2 * g->colours - 1) : (int64_t)(2 + g->lo.schedule.size() + g->up.schedule.size())) : 0;
    else if (!strcmp(what, "symgs_fused"))  // 1: a proper colouring swept with one launch per colour from the rows in sweep order
        *value = g && g->cs_ptr && !g->cs_first.empty() ? 1 : 0;
    else if (!strcmp(what, "symgs_bytes"))
        *value = g ? g->bytes : 0;
    else if (!strcmp(what, "symgs_colours"))  // 0: the matrix's own order
        *value = g ? g->colours : 0;
    else if (!strcmp(what, "symgs_order"))
        *value = m->gs_order != 0 ? 1 : 0;
    else
        return SPMV_ERR_INVALID;
    return SPMV_OK;
}
}  // namespace spmv
