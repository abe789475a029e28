// kernels_csr_panel.hip — CSR `y += A*x` through a re-ordered copy of the matrix; the kernel behind BASELINE
// config 2 (N = 10M, 32 uniform-random columns per row), and — measured — the fastest CSR path for every matrix with
// enough entries to occupy the chip (banded ones included).  Replaces CSRMatrixMatVector, src/mat_vec.cpp:44-67.
//
// Why the plain row-parallel kernel is slow there (measured, profiles/r01_*): every 8-byte gather of x
// misses the 4 MiB L2 of its XCD and drags a 128-byte line across the fabric — 320M gathers move ~43 GB
// for 4.1 GB of algorithmic traffic, and the kernel runs at the fabric's ~7 TB/s, i.e. 8.6 % of the
// algorithmic roofline.  A table that fits L2 is gathered 4.6x faster (tools/probe_gather: 265 vs 58
// Ggather/s).
//
// Layout built here once per matrix (the reference does the same kind of work before its timed loop when it
// builds the per-node sub-matrices, src/mat_vec.cpp:240-268):
//   * rows are cut into GROUPS of consecutive rows (at most 20000: the group's y accumulators fill the 160 KiB
//     LDS of one CU), with cuts that balance the entries per group;
//   * inside a group the entries are re-ordered by column PANEL (W columns) and, inside a panel, by 128-byte
//     line of x;
//   * the ordered entries are stored packed, 12 bytes each (fp64 value + one word: local row | column relative to
//     the base of the entry's 1024-entry slice) — see panel_cut_kernel / panel_expand_kernel; where that does not
//     pay, as three arrays (value, int32 column, uint16 local row: 14 bytes).
// Kernel: one 1024-thread workgroup per CU walks its group's entries front to back, so at any moment all
// 256 workgroups gather from the same few panels of x: the lines are fetched from HBM/MALL once per XCD and
// round, and every other gather hits L2.  Products are added into the group's accumulators in LDS with
// ds_add_f64; at the end the accumulators are added to y with coalesced accesses.  Because consecutive
// entries of a panel are sorted by x line, lanes of one wavefront instruction often share a line, which cuts
// the number of L2->L1 line transfers — the bound once the fabric traffic is under control: a divergent gather
// costs one 128-byte line transfer per distinct line (~258 Glines/s chip-wide = the L2's ~33 TB/s), whatever the
// load flavour.
// Staying "in step" inside an XCD is what makes the gathers hit L2: see "keeping in step" below (a workgroup barrier
// per chunk; the clock pace of round 1 is kept as an option), the two pipeline orders of a chunk, and
// panel_choose_pace, which picks chunk size, order and barrier placement by timing a handful of launches.
// What bounds C2 now (profiles/r02_pmc_bench_kernels.txt): the L2.  One product makes 234M L1->L2 read requests (202M
// gather lines + 30M streamed lines + y) and 41M fills; at the ~268 G line operations per second the eight L2s
// sustain (tools/probe_gather) that is 1.03 ms, and the kernel takes 1.13-1.15.  A CU cannot overlap the HBM stream
// with the gathers either (tools/probe_mix.hip: one CU runs them at the SUM of their times, separate CUs at the max).
// Experiments that lost were removed from this file in round 2; their logs stay in profiles/ (16-byte records,
// system-scope stream loads, pace slack, wavefront stagger, counter gate, ring pipelines, split barrier).
//
// Algorithmic bytes are counted with CSR's 12 bytes per entry; the packed layout streams exactly
// that, so what is left between achieved and roofline is the x traffic and the gather path, not layout overhead.
#include <vector>

#include <atomic>

#include "common.hpp"
#include "panel_groups.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kPanelThreads = 1024;  // 16 wavefronts: one workgroup per CU (LDS-limited)
constexpr int kLineDoubles  = 16;    // 128-byte line of x

// ---- build step 1: entries per (group, panel) + exclusive scan inside the group -----------------------------
__global__ __launch_bounds__(256) void panel_count_kernel(const int32_t* __restrict__ gstart, int W, int P,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col,
                                                          int32_t* __restrict__ tile_ptr /* [ngroups][P+1] */)
{
    extern __shared__ int32_t hist[];  // P + 1
    const int g  = blockIdx.x;
    const int r0 = gstart[g], r1 = gstart[g + 1];
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
__global__ __launch_bounds__(256) void panel_scatter_kernel(const int32_t* __restrict__ gstart, int W, int P,
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
    const int     r0 = gstart[g], r1 = gstart[g + 1];
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
__global__ __launch_bounds__(256) void panel_line_sort_kernel(int64_t ntiles, const int32_t* __restrict__ gstart, int W, int P,
                                                              const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ tile_ptr,
                                                              const int32_t* __restrict__ src_col,
                                                              const uint16_t* __restrict__ src_row,
                                                              const double* __restrict__ src_val,
                                                              int32_t* __restrict__ dst_col, uint16_t* __restrict__ dst_row,
                                                              double* __restrict__ dst_val)
{
    extern __shared__ int32_t bins[];  // W/16 + 1
    const int nb = W / kLineDoubles;
    // tiles beyond the grid are taken in turns: groups x panels can exceed what one launch may hold (a grid of more
    // than 2^32 / 256 workgroups of 256 wraps around silently)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
    const int g    = (int)(tile / P);
    const int p    = (int)(tile % P);
    const int base = row_ptr[gstart[g]];
    const int t0   = base + tile_ptr[(size_t)g * (P + 1) + p];
    const int t1   = base + tile_ptr[(size_t)g * (P + 1) + p + 1];
    if (t0 == t1) continue;  // (uniform over the workgroup)
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
    __syncthreads();  // the bins are cleared for the next tile
    }
}

// ---- the product --------------------------------------------------------------------------------------------
// One chunk = UNROLL x 1024 consecutive entries of the group: every lane has UNROLL entries in flight between
// the streamed loads and the LDS adds (one workgroup per CU = 16 wavefronts is all the thread-level
// parallelism the LDS footprint allows, so the memory-level parallelism has to come from here).
// LAYOUT 0: three arrays (value fp64, column int32, local row uint16): 14 bytes per entry, three load instructions.
// LAYOUT 4: LAYOUT 3 with the slices stored in interleaved pairs, read with 8- and 16-byte loads (see load_raw).
// LAYOUT 3: two arrays (value fp64, one packed 32-bit word): 12 bytes per entry, two load instructions.  The word
//           holds the local row in its low `rowbits` bits and, above them, the column relative to the base of the
//           entry's slice (1024 packed entries; the entries are ordered by x line, so a slice spans few
//           columns; see panel_cut_kernel).  The bases are one int32 per slice, read through the scalar cache.
// (Tried and dropped, logs in profiles/r01_*: 16-byte {value, column, row} records — one load instruction per entry,
// 3.2-3.6 ms on C2 — and system-scope loads for the stream.)
using u32x2p = unsigned __attribute__((ext_vector_type(2)));

template <int UNROLL, int LAYOUT>
struct PanelBatch
{
    int      c[UNROLL];
    unsigned r[UNROLL];
    double   v[UNROLL];
    // LAYOUT 3: turn the packed words into column and local row.  Kept apart from the loads so that the pipelined
    // loop can leave the words untouched (and their loads in flight) until the iteration that uses them.
    // sb: bases of the UNROLL slices this batch reads, uniform over the workgroup
    __device__ __forceinline__ void unpack(const int32_t* __restrict__ sb, int rowbits)
    {
        if constexpr (LAYOUT >= 3)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const unsigned w = (unsigned)c[u];
                c[u]             = sb[u] + (int)(w >> rowbits);
                r[u]             = w & ((1u << rowbits) - 1u);
            }
        }
    }
    __device__ __forceinline__ void load(const int32_t* __restrict__ pcol, const uint16_t* __restrict__ prow,
                                         const double* __restrict__ pval, int e, const int32_t* __restrict__ sb = nullptr,
                                         int rowbits = 0)
    {
        load_raw(pcol, prow, pval, e);
        unpack(sb, rowbits);
    }
    // e: index of this lane's entry in the first slice of the batch (slice * 1024 + lane)
    __device__ __forceinline__ void load_raw(const int32_t* __restrict__ pcol, const uint16_t* __restrict__ prow,
                                             const double* __restrict__ pval, int e)
    {
        if constexpr (LAYOUT == 4 && UNROLL >= 2)
        {
            // slices 2k and 2k+1 are stored interleaved (entry t of slice 2k at 2t, of slice 2k+1 at 2t + 1 of their
            // 2048-entry block): one 8-byte and one 16-byte load bring this lane's entries of both — half the vector
            // memory instructions for the stream (a wavefront instruction costs the same ~32 clocks of the CU's
            // address pipe whether it moves 4 or 16 bytes per lane)
            const int     t  = e & (kPanelThreads - 1);
            const size_t  p0 = (size_t)(e >> 11) * kPanelThreads + t;  // pair index: (slice / 2) * 1024 + lane; the batch starts at an even slice
            const u32x2p* w2 = reinterpret_cast<const u32x2p*>(pcol) + p0;
            const f64x2*  v2 = reinterpret_cast<const f64x2*>(pval) + p0;
#pragma unroll
            for (int u = 0; u < UNROLL / 2; ++u)
            {
                const u32x2p w = load_stream(w2 + (size_t)u * kPanelThreads);
                const f64x2  d = load_stream(v2 + (size_t)u * kPanelThreads);
                c[2 * u]       = (int)w.x;
                c[2 * u + 1]   = (int)w.y;
                v[2 * u]       = d.x;
                v[2 * u + 1]   = d.y;
            }
        }
        else if constexpr (LAYOUT == 4)
        {
            const int    t  = e & (kPanelThreads - 1), sl = e >> 10;
            const size_t at = ((size_t)(sl >> 1) << 11) + 2 * (size_t)t + (size_t)(sl & 1);
            c[0]            = load_stream(pcol + at);
            v[0]            = load_stream(pval + at);
        }
        else
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                c[u] = load_stream(pcol + e + u * kPanelThreads);  // LAYOUT 3: the packed word, see unpack()
                if constexpr (LAYOUT < 3) r[u] = load_stream(prow + e + u * kPanelThreads);
                v[u] = load_stream(pval + e + u * kPanelThreads);
            }
        }
    }
    __device__ __forceinline__ void apply(const double* __restrict__ x, double* acc) const
    {
        double xv[UNROLL];
        gather(x, xv);
        add(acc, xv);
    }
    // the two halves of apply(), for the gather-first pipeline
    __device__ __forceinline__ void gather(const double* __restrict__ x, double (&xv)[UNROLL]) const
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) xv[u] = x[c[u]];
    }
    __device__ __forceinline__ void add(double* acc, const double (&xv)[UNROLL]) const
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) atomicAdd(&acc[r[u]], v[u] * xv[u]);  // ds_add_f64
    }
    // single entry (the ragged tail of a group)
    __device__ static __forceinline__ void one(const int32_t* __restrict__ pcol, const uint16_t* __restrict__ prow,
                                               const double* __restrict__ pval, int e, const double* __restrict__ x,
                                               double* acc, const int32_t* __restrict__ sb = nullptr, int rowbits = 0)
    {
        PanelBatch<1, LAYOUT> b;
        b.load(pcol, prow, pval, e, sb, rowbits);
        b.apply(x, acc);
    }
};

// LAYOUT 3 build.  A group's entries (ordered by x line) are cut into slices of at most 1024 entries that span
// fewer than 2^colbits columns; every slice is stored as exactly 1024 packed entries (the rest are pads: value 0,
// column = the slice base, local row = `pad_row`, a spare accumulator past the last row of the fullest group that
// is never written back: a pad may add 0 * inf = NaN there without touching y), so a group is a whole number of
// 1024-entry slices and a column gap inside a group (band wrap-around, far couplings) costs at most one partly
// filled slice.
struct PanelPacked
{
    const int32_t* sbase;    // [slices] first column (line-aligned) of every slice
    const int32_t* soff;     // [ngroups + 1] first slice of every group
    int            rowbits;  // low bits of the packed word that hold the local row
};

// One thread per group walks its entries.  Pass 1 (soff == nullptr) counts the slices, pass 2 records them.
__global__ void panel_cut_kernel(const int32_t* __restrict__ gstart, int ngroups, const int32_t* __restrict__ row_ptr,
                                 const int32_t* __restrict__ col, int colbits, const int32_t* __restrict__ soff,
                                 int32_t* __restrict__ nslices, int32_t* __restrict__ sbase, int32_t* __restrict__ ssrc,
                                 int32_t* __restrict__ scount)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int begin = row_ptr[gstart[g]], end = row_ptr[gstart[g + 1]];
    int       pos = begin, k = 0;
    while (pos < end)
    {
        const int base_line  = col[pos] / kLineDoubles;
        const int limit_line = base_line + (1 << (colbits - 4));  // kLineDoubles == 16 columns per line
        const int most       = min(kPanelThreads, end - pos);
        int       take       = most;
        if (col[pos + most - 1] / kLineDoubles >= limit_line)
        {
            // lines are non-decreasing along the group: largest take with line(col[pos + take - 1]) < limit_line
            int lo = 1, hi = most - 1;  // entry pos itself always fits
            while (lo < hi)
            {
                const int mid = (lo + hi + 1) / 2;
                if (col[pos + mid - 1] / kLineDoubles < limit_line)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            take = lo;
        }
        if (soff)
        {
            const int idx = soff[g] + k;
            sbase[idx]    = base_line * kLineDoubles;
            ssrc[idx]     = pos;
            scount[idx]   = take;
        }
        pos += take;
        ++k;
    }
    if (!soff) nslices[g] = k;
}

__global__ __launch_bounds__(256) void panel_expand_kernel(int total, const int32_t* __restrict__ sbase,
                                                           const int32_t* __restrict__ ssrc,
                                                           const int32_t* __restrict__ scount, const int32_t* __restrict__ col,
                                                           const uint16_t* __restrict__ row, const double* __restrict__ val,
                                                           int rowbits, unsigned pad_row, int pair,
                                                           uint32_t* __restrict__ packed, double* __restrict__ pval)
{
    const unsigned pad = pad_row;
    for (int sl = blockIdx.x; sl < total; sl += gridDim.x)
    {
        const int base = sbase[sl], src = ssrc[sl], cnt = scount[sl];
        for (int j = threadIdx.x; j < kPanelThreads; j += 256)
        {
            // pair: slices 2k, 2k+1 interleaved entry by entry (LAYOUT 4)
            const size_t dst = pair ? ((size_t)(sl >> 1) << 11) + 2 * (size_t)j + (size_t)(sl & 1) : (size_t)sl * kPanelThreads + j;
            if (j < cnt)
            {
                packed[dst] = ((unsigned)(col[src + j] - base) << rowbits) | (unsigned)row[src + j];
                pval[dst]   = val[src + j];
            }
            else
            {
                packed[dst] = pad;
                pval[dst]   = 0.0;
            }
        }
    }
}

__global__ void group_nnz_max_kernel(const int32_t* __restrict__ gstart, int ngroups, const int32_t* __restrict__ row_ptr,
                                     int32_t* __restrict__ out_max)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int r0 = gstart[g], r1 = gstart[g + 1];
    atomicMax(out_max, row_ptr[r1] - row_ptr[r0]);
}

// ---- keeping in step ------------------------------------------------------------------------------------------
// The x lines one CU pulls into its XCD's L2 serve the other 31 CUs of the XCD only while all of them gather from the
// same stretch of x.  What keeps them there (round 2's trace build, profiles/r02_trace_*):
//   * inside a workgroup, a barrier per chunk (SYNCT below).  Without it the oldest-first issue arbitration lets
//     wavefront 0 start chunk b+1 while wavefront 15 is still issuing chunk b; 2 % of the chunks then take 15 us
//     instead of 7, the workgroup falls behind its XCD and its gathers start missing L2;
//   * between workgroups, nothing explicit: who leads the sweep takes the L2 misses and slows down, the others close
//     up.  (Round 1 throttled every chunk to a clock instead; it compensated for the missing barrier and cost 15 %.  That
//     pace, its run-time guard, the per-XCD offsets, a split barrier through an LDS counter and the trace build lived on as
//     options inside the chunk loop through round 4 and were deleted in round 5 together with the kernel that carried them:
//     the headline's schedule depended on that dead code being there - see below.  A counter gate per XCD and schedules
//     with the wavefronts or the halves of an XCD apart were measured and dropped earlier: profiles/r01_pmc_gate_variants.txt,
//     r01_tune_csr_stagger.txt, r02_tune_csr_c2_ring_paces.txt.)

// ---- the chunk pipeline with its order written down (round 5) ------------------------------------------------------------------
// Rounds 2-4 ran the panel product through ONE kernel with run-time switches inside its chunk loop (clock pace, pace guard,
// trace stamps, barrier placement), and its C2 instance owed its schedule to code it never ran: the (run-time dead) pace
// block split a chunk into basic blocks, and because of that split the compiler (a) copied the next chunk's registers into
// the current ones AFTER the workgroup barrier instead of before it - so that no wavefront waited for its HBM loads in front
// of the barrier - and (b) kept the order gathers -> next chunk's stream -> LDS adds.  With the dead block compiled out the
// chunk became one basic block, the copies (and with them `s_waitcnt vmcnt(0)`) moved in front of the barrier and the loads
// were interleaved: 1.108 -> 1.25 ms (profiles/r03_tune_csr_c2_no_pace_code.txt).  This kernel depends on no such accident:
//   * two register sets used in turn (chunk b in A, b + 1 in B, b + 2 in A ...): nothing is copied, so nothing has to have
//     arrived at the loop's back edge;
//   * the groups of a chunk's memory instructions - the gathers of x, the next chunk's stream, the LDS adds - are kept apart
//     by scheduling fences (__builtin_amdgcn_sched_barrier), and nothing of a chunk moves in front of its workgroup barrier;
//   * ORDER 2 (gather-first): gathers -> next chunk's stream -> adds: vector loads return in order, so the adds wait for the
//     gathers alone and the HBM latency of the stream runs under them (scattered columns: C2);
//     ORDER 1 (stream-first): next chunk's stream -> gathers -> adds (local columns: bands; shards whose x is 2-4x their rows);
//     ORDER 0: no prefetch (kept as the plain reference order);
//   * SYNCT 1: a workgroup barrier at the top of every chunk, 3: between a chunk's loads and its LDS adds, 0: none - what keeps
//     the 16 wavefronts of a workgroup in the same chunk and thereby the workgroups of an XCD in step;
//   * no pace, no guard, no trace, no run-time switch inside the chunk loop.
// A/B on one box against the round-4 kernel (tools/ab_panel_order.py, profiles/r05_ab_panel_order.txt): C2 1.100 against 1.106 ms,
// barrier placement 3: 1.114 against 1.146, U = 4: 1.225 against 1.336, the N = 2 shard shape gather-first 1.485 against 1.602.
// LAYOUT 0: three arrays (value, int32 column, uint16 local row); 3: 12-byte packed entries; 4: the same in paired slices.
// (The three-array layout - the fallback for groups too sparse to pack - is the one place where the round-4 kernel's accidental
// schedule stays ahead: C2 forced into it 1.373 against 1.278 ms; with the fences left out 1.39-1.52: profiles/r05_probe_panel_three_array_layout_fences.txt.)
// TRIAL: the same code under another name, so that build-time trial launches show up apart from products in a kernel trace.
template <int UNROLL, int LAYOUT, int ORDER, bool TRIAL, int SYNCT>
__global__ __launch_bounds__(kPanelThreads) void csr_panel_pp_kernel(const int32_t* __restrict__ gstart, int ngroups, const int32_t* __restrict__ row_ptr,
                                                                     const int32_t* __restrict__ pcol, const uint16_t* __restrict__ prow,
                                                                     const double* __restrict__ pval, const double* __restrict__ x,
                                                                     double* __restrict__ y, const int32_t* __restrict__ sbase,
                                                                     const int32_t* __restrict__ soff, int rowbits, int overwrite,
                                                                     const double* __restrict__ dot_w, double* __restrict__ dot_out)
{
    extern __shared__ double acc[];  // the group's accumulators
    constexpr int STEP = UNROLL * kPanelThreads;
    const int     lane = threadIdx.x & 63;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int r0   = gstart[g];
        const int rows = gstart[g + 1] - r0;
        for (int i = threadIdx.x; i < rows; i += kPanelThreads) acc[i] = 0.0;
        __syncthreads();
        // the packed layouts keep their own (padded) entry numbering: whole slices of 1024
        const int begin = LAYOUT >= 3 ? soff[g] * kPanelThreads : row_ptr[r0];
        const int end   = LAYOUT >= 3 ? soff[g + 1] * kPanelThreads : row_ptr[r0 + rows];
        const int nfull = (end - begin) / STEP;  // chunks in which every lane has UNROLL valid entries
        const int e0    = begin + threadIdx.x;
        const int32_t* __restrict__ sb = LAYOUT >= 3 ? sbase + soff[g] : pcol;  // slice bases of this group (else unused)
        PanelBatch<UNROLL, LAYOUT> A, B;
        // one chunk: `cur` holds chunk b (its loads requested one chunk ago), chunk b + 1 is requested into `nxt` (the last
        // chunk re-reads itself: a load behind a branch would make the adds wait for it)
        auto chunk = [&](PanelBatch<UNROLL, LAYOUT>& cur, PanelBatch<UNROLL, LAYOUT>& nxt, int b) __attribute__((always_inline)) {
            if constexpr (SYNCT == 1)
            {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);  // nothing of this chunk (its unpacking waits for HBM) moves in front of the barrier
            }
            const int e_next = e0 + (b + 1 < nfull ? b + 1 : b) * STEP;
            if constexpr (ORDER == 0) cur.load_raw(pcol, prow, pval, e0 + b * STEP);
            if constexpr (ORDER == 1)
            {
                nxt.load_raw(pcol, prow, pval, e_next);
                __builtin_amdgcn_sched_barrier(0);  // the next chunk's stream is requested before ...
            }
            cur.unpack(sb + b * UNROLL, rowbits);
            double xv[UNROLL];
            cur.gather(x, xv);
            __builtin_amdgcn_sched_barrier(0);  // ... the gathers (stream-first); the gathers before ...
            if constexpr (ORDER == 2)
            {
                nxt.load_raw(pcol, prow, pval, e_next);
                __builtin_amdgcn_sched_barrier(0);  // ... the next chunk's stream (gather-first); and all loads before the adds
            }
            if constexpr (SYNCT == 3)
            {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            cur.add(acc, xv);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (ORDER != 0 && nfull > 0) A.load_raw(pcol, prow, pval, e0);
        int b = 0;
        for (; b + 1 < nfull; b += 2)
        {
            chunk(A, B, b);
            chunk(B, A, b + 1);
        }
        if (b < nfull) chunk(A, B, b);
        int e = e0 + nfull * STEP;
        for (int t = nfull * UNROLL; e < end; e += kPanelThreads, ++t) PanelBatch<1, LAYOUT>::one(pcol, prow, pval, e, x, acc, sb + t, rowbits);
        __syncthreads();
        // write-back: y += (or =) the group's sums; optionally the solver's dot product w . y_new rides along
        // (spmv_apply_dot: saves the separate pass over w and y)
        double part = 0.0;
        for (int i = threadIdx.x; i < rows; i += kPanelThreads)
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

void csr_panel_free(spmv_mat* m)
{
    if (m->pb_col) (void)hipFree(m->pb_col);
    if (m->pb_row) (void)hipFree(m->pb_row);
    if (m->pb_val) (void)hipFree(m->pb_val);
    if (m->pb_pack) (void)hipFree(m->pb_pack);
    if (m->pb_sbase) (void)hipFree(m->pb_sbase);
    if (m->pb_soff) (void)hipFree(m->pb_soff);
    if (m->pb_gstart) (void)hipFree(m->pb_gstart);
    m->pb_gstart = nullptr;
    m->pb_pack   = nullptr;
    m->pb_pair   = false;
    m->pb_sbase  = nullptr;
    m->pb_soff   = nullptr;
    m->pb_col = nullptr;
    m->pb_row = nullptr;
    m->pb_val = nullptr;
    m->pb_built_sort = -1;
    m->pb_tuned_key = 0;
    m->pb_unroll_tuned = 0;
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

// Re-store the (line-ordered) three-array layout as 12-byte packed entries.  Not an error when it does not work
// out (no memory, or so many column gaps that the padding would outweigh the two bytes saved): the three arrays
// stay and the product runs from them.
static void panel_pack(spmv_mat* m, int ngroups, int max_rows, bool pair)
{
    spmv_ctx*   ctx     = m->ctx;
    hipStream_t s       = ctx->stream;
    const int   rowbits = 32 - __builtin_clz((unsigned)std::max(1, max_rows));  // local rows 0..max_rows (max_rows = pads)
    const int   colbits = 32 - rowbits;
    int32_t *d_n = nullptr, *ssrc = nullptr, *scount = nullptr;
    uint32_t* pack = nullptr;
    double*   pval = nullptr;
    int32_t * sbase = nullptr, *d_soff = nullptr;
    bool      ok   = false;
    do
    {
        if (rowbits > 16 || hipMalloc(&d_n, sizeof(int32_t) * (size_t)ngroups) != hipSuccess) break;
        const unsigned gb = (unsigned)ceil_div(ngroups, 64);
        hipLaunchKernelGGL(panel_cut_kernel, dim3(gb), dim3(64), 0, s, m->pb_gstart, ngroups, m->a, m->pb_col, colbits,
                           (const int32_t*)nullptr, d_n, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
        std::vector<int32_t> soff((size_t)ngroups + 1, 0);
        if (hipMemcpyAsync(soff.data() + 1, d_n, sizeof(int32_t) * (size_t)ngroups, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            break;
        int64_t total = 0;
        for (int g = 1; g <= ngroups; ++g)
        {
            // pair: an even number of slices per group (the odd one out gets an empty partner: count 0, base 0)
            total += pair ? (soff[(size_t)g] + 1) / 2 * 2 : soff[(size_t)g];
            if (total > INT32_MAX / kPanelThreads) break;
            soff[(size_t)g] = (int32_t)total;
        }
        // padded entries: whole slices.  Worth it while 12 B x padded < 14 B x stored (with a margin)
        const int64_t padded = total * kPanelThreads;
        if (total > INT32_MAX / kPanelThreads || padded * 12 > m->nnz * 13 + (int64_t)ngroups * kPanelThreads * 12 * (pair ? 2 : 1)) break;
        if (hipMalloc(&ssrc, sizeof(int32_t) * (size_t)total) != hipSuccess ||
            hipMalloc(&scount, sizeof(int32_t) * (size_t)total) != hipSuccess ||
            hipMalloc(&sbase, sizeof(int32_t) * (size_t)total) != hipSuccess ||
            hipMalloc(&d_soff, sizeof(int32_t) * soff.size()) != hipSuccess ||
            hipMalloc(&pack, sizeof(uint32_t) * (size_t)padded) != hipSuccess || hipMalloc(&pval, sizeof(double) * (size_t)padded) != hipSuccess)
            break;
        if (hipMemcpyAsync(d_soff, soff.data(), sizeof(int32_t) * soff.size(), hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemsetAsync(sbase, 0, sizeof(int32_t) * (size_t)total, s) != hipSuccess || hipMemsetAsync(ssrc, 0, sizeof(int32_t) * (size_t)total, s) != hipSuccess ||
            hipMemsetAsync(scount, 0, sizeof(int32_t) * (size_t)total, s) != hipSuccess)
            break;
        hipLaunchKernelGGL(panel_cut_kernel, dim3(gb), dim3(64), 0, s, m->pb_gstart, ngroups, m->a, m->pb_col, colbits,
                           (const int32_t*)d_soff, (int32_t*)nullptr, sbase, ssrc, scount);
        hipLaunchKernelGGL(panel_expand_kernel, dim3((unsigned)std::min<int64_t>(total, kMaxGrid * 4)), dim3(256), 0, s,
                           (int)total, sbase, ssrc, scount, m->pb_col, m->pb_row, m->pb_val, rowbits, (unsigned)max_rows, pair ? 1 : 0, pack,
                           pval);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;  // soff (host) is done with
        ok = true;
        // the packed words and padded values replace the three arrays
        (void)hipFree(m->pb_col);
        (void)hipFree(m->pb_row);
        (void)hipFree(m->pb_val);
        m->pb_col     = nullptr;
        m->pb_row     = nullptr;
        m->pb_val     = pval;
        m->pb_pack    = pack;
        m->pb_sbase   = sbase;
        m->pb_soff    = d_soff;
        m->pb_rowbits = rowbits;
        m->pb_pair    = pair;
        m->pb_slices  = (int32_t)total;
        m->pb_bytes   = padded * 12 + (int64_t)sizeof(int32_t) * (total + ngroups + 1);
    } while (0);
    if (d_n) (void)hipFree(d_n);
    if (ssrc) (void)hipFree(ssrc);
    if (scount) (void)hipFree(scount);
    if (!ok)
    {
        if (pack) (void)hipFree(pack);
        if (pval) (void)hipFree(pval);
        if (sbase) (void)hipFree(sbase);
        if (d_soff) (void)hipFree(d_soff);
        (void)hipGetLastError();
    }
}

int csr_panel_build(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (m->nrow == 0 || m->nnz == 0) return SPMV_OK;
    constexpr int kCapRows = 20000;  // 160,000 B of the CU's 163,840 B LDS
    int G = m->pb_group_rows > 0 ? std::min(m->pb_group_rows, kCapRows) : pick_group_rows(m->nrow, kCapRows);
    G     = std::max<int64_t>(G, ceil_div(m->nrow, (int64_t)1 << 21));  // at most 2M groups: one workgroup of 256 per group in the build kernels
    int W = m->pb_panel_width > 0 ? m->pb_panel_width : 128 * 1024;
    W     = std::min(524288, std::max(kLineDoubles, (W / kLineDoubles) * kLineDoubles));  // (the line sort keeps W / 16 + 1 counters of a panel in LDS)
    while (ceil_div(m->ncol, W) > 8192) W *= 2;  // the per-group histogram lives in LDS
    const bool sort = m->pb_sort != 0;
    const bool pack = (m->pb_aos == 3 || m->pb_aos == 4) && sort;  // 12-byte entries (needs the line order); kept only if every slice fits
    if (m->pb_val && m->pb_built_rows == G && m->pb_built_width == W && m->pb_built_sort == (int)sort && m->pb_built_layout == m->pb_aos &&
        (m->pb_rounds_req == 0 || m->pb_rounds_req == m->pb_built_rounds))
        return panel_choose_pace(m);  // the layout in memory was built with these parameters
    if (!m->b || !m->v) SPMV_FAIL(SPMV_ERR_INVALID, "the panel layout cannot be re-built: this handle gave up its CSR arrays (panel_keep_csr = 0)");
    csr_panel_free(m);
    // Row groups.  Requested size (panel_rows): equal groups of G rows.  Otherwise the boundaries balance the
    // ENTRIES per group — a workgroup's time is its group's entry count, and the slowest of a round sets the pace
    // (C4's power-law rows: equal-row groups differ by 15 %) — under the row cap that LDS imposes.  Smallest bound T
    // such that a greedy cut (entries <= T, rows <= cap) needs no more groups than equal groups of G rows would.
    std::vector<int32_t> gstart;
    int                  try_rounds = 0;  // groups for this many rounds are worth a timing against the single round
    {
        std::vector<int32_t> rp((size_t)m->nrow + 1);
        SPMV_HIP(hipMemcpyAsync(rp.data(), m->a, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, ctx->stream));
        SPMV_HIP(hipStreamSynchronize(ctx->stream));
        const int want = (int)ceil_div(m->nrow, G);
        if (m->pb_group_rows > 0)
            (void)panel_cut(rp, m->nrow, (int64_t)INT32_MAX, G, &gstart);  // a requested size is exact: equal groups of G rows
        else
        {
            // light groups may take up to the LDS cap so heavy ones can shrink (panel_groups.hpp)
            const double busiest = panel_balanced_cut(rp, m->nrow, want, kCapRows, kNumCu, &gstart);
            // Many rows AND skewed lengths (R-MAT scale 22: 4.2M rows, 210 groups' worth of rows under the LDS cap, so 256 groups
            // leave the heavy rows 46 to spread over): the busiest group holds 2.24x the mean and sets the product's time.  More
            // groups than one round let the heavy stretches be cut finer (512: 1.21x; the pairing b, b + 256 puts a heavy and a
            // light stretch on the same CU) - but every further round is another sweep of x by workgroups that are no longer in
            // step, and entries per CU are not the whole cost: that matrix went from 0.333 to 0.267 ms, what is left of it after
            // its 254 longest rows were split off (1.97x) from 0.327 to 0.402.  So the alternative is TIMED against one round
            // (below, after the build); "panel_rounds" k forces k rounds' worth of groups, 1 the single round.
            if (m->pb_rounds_req > 1)
                (void)panel_balanced_cut(rp, m->nrow, want * m->pb_rounds_req, kCapRows, kNumCu, &gstart);
            else if (m->pb_rounds_req == 0 && select_trials_enabled(m))
                try_rounds = panel_rounds_worth_a_trial(rp, m->nrow, want, kCapRows, kNumCu, busiest);
        }
    }
    const int ngroups = (int)gstart.size() - 1;
    int       max_rows = 0;
    for (int g = 0; g < ngroups; ++g) max_rows = std::max(max_rows, gstart[(size_t)g + 1] - gstart[(size_t)g]);
    SPMV_HIP(hipMalloc(&m->pb_gstart, sizeof(int32_t) * gstart.size()));
    SPMV_HIP(hipMemcpyAsync(m->pb_gstart, gstart.data(), sizeof(int32_t) * gstart.size(), hipMemcpyHostToDevice, ctx->stream));
    SPMV_HIP(hipStreamSynchronize(ctx->stream));  // gstart (host) goes out of use only after the copy
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
        hipLaunchKernelGGL(panel_count_kernel, dim3(ngroups), dim3(256), sizeof(int32_t) * (P + 1), s, m->pb_gstart, W, P,
                           m->a, m->b, tile_ptr);
        int32_t*  scol = sort ? tcol : m->pb_col;
        uint16_t* srow = sort ? trow : m->pb_row;
        double*   sval = sort ? tval : m->pb_val;
        hipLaunchKernelGGL(panel_scatter_kernel, dim3(ngroups), dim3(256), sizeof(int32_t) * P, s, m->pb_gstart, W, P, m->a,
                           m->b, m->v, tile_ptr, scol, srow, sval);
        if (sort && sizeof(int32_t) * (W / kLineDoubles + 1) > 65536)  // one bin per x line of the panel, in LDS
            (void)hipFuncSetAttribute((const void*)panel_line_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160008);
        if (sort)
            hipLaunchKernelGGL(panel_line_sort_kernel, dim3((unsigned)std::min<int64_t>((int64_t)ngroups * P, (int64_t)1 << 22)), dim3(256),
                               sizeof(int32_t) * (W / kLineDoubles + 1), s, (int64_t)ngroups * P, m->pb_gstart, W, P, m->a, tile_ptr, tcol, trow,
                               tval, m->pb_col, m->pb_row, m->pb_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;
    } while (0);
    if (tile_ptr) (void)hipFree(tile_ptr);
    if (tcol) (void)hipFree(tcol);
    if (trow) (void)hipFree(trow);
    if (tval) (void)hipFree(tval);
    if (rc != SPMV_OK)
    {
        csr_panel_free(m);
        SPMV_FAIL(rc, "building the panel layout (%d groups of %d rows, %d panels of %d columns) failed: %s", ngroups, G, P,
                  W, hipGetErrorString(hipGetLastError()));
    }
    m->pb_built_rows  = G;
    m->pb_max_rows    = max_rows;
    m->pb_built_width = W;
    m->pb_built_sort  = (int)sort;
    m->pb_ngroups     = ngroups;
    {
        // entries of the fullest group bound the number of gate slots per round
        SPMV_TRY(ensure_scratch(ctx, 64));
        int32_t* d_max = (int32_t*)ctx->scratch;
        int32_t  h_max = 0;
        SPMV_HIP(hipMemsetAsync(d_max, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL(group_nnz_max_kernel, dim3((unsigned)ceil_div(ngroups, 256)), dim3(256), 0, s, m->pb_gstart, ngroups,
                           m->a, d_max);
        SPMV_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SPMV_HIP(hipStreamSynchronize(s));
        m->pb_max_group_nnz = h_max;
    }
    m->pb_bytes       = (int64_t)(nnz * 14);
    m->pb_built_layout = m->pb_aos;
    if (pack) panel_pack(m, ngroups, max_rows, m->pb_aos == 4);  // keeps the three arrays when packing does not pay
    m->device_bytes += m->pb_bytes;
    m->pb_built_rounds = m->pb_rounds_req > 1 ? m->pb_rounds_req : 1;
    SPMV_TRY(panel_choose_pace(m));
    if (try_rounds > 1)
    {
        // one round (just built) against `try_rounds`: the layout is built again for the timing, and a third time if the
        // single round wins (one-off; 0.05-0.1 s each at 67M entries)
        select_scratch sv;
        if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return SPMV_OK;  // no room to try: the single round stays
        float t_one = 0.f, t_more = 0.f;
        int   rc    = select_time(ctx, [&] { return csr_panel_apply(ctx, m, sv.x, sv.y); }, 1e30f, &t_one);
        if (rc != SPMV_OK) return rc;
        m->pb_rounds_req = try_rounds;
        if ((rc = csr_panel_build(m)) != SPMV_OK)
        {
            (void)hipGetLastError();
            m->pb_rounds_req = 1;  // (no memory for the finer cut: back to the single round)
            rc               = csr_panel_build(m);
            m->pb_rounds_req = 0;
            return rc;
        }
        rc = select_time(ctx, [&] { return csr_panel_apply(ctx, m, sv.x, sv.y); }, 1e30f, &t_more);
        if (rc == SPMV_OK && !(t_more < 0.97f * t_one))
        {
            m->pb_rounds_req = 1;
            rc               = csr_panel_build(m);
        }
        m->pb_rounds_req = 0;  // (automatic again; the layout in memory stays until a parameter changes)
        m->pb_rounds_us[0] = t_one * 1000.f;
        m->pb_rounds_us[1] = t_more * 1000.f;
        return rc;
    }
    return SPMV_OK;
}

static int panel_launch(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, bool trial, const apply_extra& ex);

// Chooses, by timing a few launches on scratch vectors (the gather addresses, not the values, set the time), the chunk
// size, the order of a chunk's loads and how the 16 wavefronts of a workgroup are kept together.  Part of the one-off
// analysis, like the reference's shard construction before its timed loop (src/mat_vec.cpp:240-268).
//   scattered columns (C2, C4): U = 8, gather-first, a workgroup barrier at the top of every chunk.  With the barrier the
//     workgroups of an XCD fall into step by themselves (whoever leads the sweep of x takes the L2 misses and slows down);
//     without it wavefront 0 starts the next chunk while wavefront 15 is still issuing this one and starves it (round 2's
//     trace), which is what the clock pace of round 1 was really compensating;
//   local columns (bands): U = 4, gather-first, the barrier between a chunk's loads and its adds (round 5: 0.615 ms on the
//     65536 band against 0.630-0.647 for every candidate of the round-4 list);
//   shards whose x is 2-4x their rows: U = 8, stream-first, barrier at the top.
// The list is what tools/sweep_panel_configs.py ranks first on those shapes (profiles/r05_sweep_panel_configs_*.txt).
// (The function keeps its name from round 1, when what it chose was the pace of a clock throttle.)
int panel_choose_pace(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    const int key = 1000 + std::max(m->pb_unroll, 0) * 10 + (m->pb_pipe + 1) + ((m->pb_sync + 1) & 7) * 1000000;
    if (m->pb_tuned_key == key) return SPMV_OK;  // already tried for this layout and these requests
    m->pb_tuned_key    = 0;
    m->pb_unroll_tuned = 0;
    m->pb_pipe_tuned   = 0;
    m->pb_sync_tuned   = 0;
    if (m->pb_unroll > 0 && m->pb_pipe >= 0 && m->pb_sync >= 0) return SPMV_OK;  // nothing left to choose
    const bool worth = (double)m->ncol * 8.0 > 4.0 * 1048576.0 && m->pb_max_group_nnz >= 8LL * 8 * kPanelThreads;
    if (!worth) return SPMV_OK;  // x fits L2 or the groups are a few chunks long: nothing to keep in step
    // SPMV_PANEL_TRIAL=0 (or panel_trial = 0): no timing launches at all; take what wins on scattered columns and costs
    // a few per cent on local ones (U = 8, gather-first, barrier at the top of the chunk)
    if (!select_trials_enabled(m))
    {
        m->pb_unroll_tuned = m->pb_unroll <= 0 ? 8 : 0;
        m->pb_pipe_tuned   = m->pb_pack ? 2 : 1;
        m->pb_sync_tuned   = 1;
        m->pb_tuned_key    = key;
        return SPMV_OK;
    }
    select_scratch sv;  // zeroed x and y: from the context's trial arena where they fit (select.hip), else allocations of their own
    if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return SPMV_OK;  // no room to try: defaults
    double *x = sv.x, *y = sv.y;
    int  rc    = SPMV_OK;
    auto timed = [&](int launches, float* ms) -> int {
        int r = panel_launch(ctx, m, x, y, true, apply_extra{});  // warm
        if (r != SPMV_OK) return r;
        (void)hipEventRecord(ctx->ev_begin, ctx->stream);
        for (int i = 0; i < launches && r == SPMV_OK; ++i) r = panel_launch(ctx, m, x, y, true, apply_extra{});
        (void)hipEventRecord(ctx->ev_end, ctx->stream);
        if (r == SPMV_OK && (hipEventSynchronize(ctx->ev_end) != hipSuccess ||
                             hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end) != hipSuccess))
            r = SPMV_ERR_HIP;
        *ms /= (float)launches;
        return r;
    };
    struct Try
    {
        int unroll, pipe, sync;
    };
    const Try packed_tries[] = {{8, 2, 1}, {8, 2, 3}, {4, 2, 3}, {8, 1, 1}, {8, 1, 3}, {4, 2, 1}, {4, 1, 0}, {8, 1, 0}};
    const Try plain_tries[]  = {{8, 1, 1}, {4, 1, 1}, {4, 1, 3}, {8, 2, 1}, {8, 1, 0}, {4, 1, 0}, {2, 1, 0}};
    const Try* tries  = m->pb_pack ? packed_tries : plain_tries;
    const int  ntries = m->pb_pack ? 8 : 7;
    Try    best{m->pb_unroll > 0 ? m->pb_unroll : 8, m->pb_pipe >= 0 ? m->pb_pipe : 1, m->pb_sync >= 0 ? m->pb_sync : 0};
    double best_ms = 1e30;
    int    tried   = 0;
    for (int pass = 0; pass < 2 && rc == SPMV_OK; ++pass)
    {
        // pass 1 only if no candidate matches the requests: the requested values with the remaining defaults
        for (int i = 0; i < (pass == 0 ? ntries : 1) && rc == SPMV_OK; ++i)
        {
            const Try t = pass == 0 ? tries[i] : best;
            if (pass == 0 && ((m->pb_unroll > 0 && m->pb_unroll != t.unroll) || (m->pb_pipe >= 0 && m->pb_pipe != t.pipe) ||
                              (m->pb_sync >= 0 && m->pb_sync != t.sync)))
                continue;
            if (pass == 1 && tried > 0) break;
            m->pb_unroll_tuned = t.unroll;
            m->pb_pipe_tuned   = t.pipe;
            m->pb_sync_tuned   = t.sync;
            float a = 0.f, b = 0.f;
            if ((rc = timed(4, &a)) != SPMV_OK || (rc = timed(4, &b)) != SPMV_OK) break;
            ++tried;
            const double ms = std::min(a, b);
            if (ms < best_ms * (best_ms < 1e29 ? 0.99 : 1.0))  // later candidates have to win by 1 %
            {
                best_ms = ms;
                best    = t;
            }
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    sv.release();
    m->pb_unroll_tuned = rc == SPMV_OK && m->pb_unroll <= 0 ? best.unroll : 0;
    m->pb_pipe_tuned   = rc == SPMV_OK ? best.pipe : 0;
    m->pb_sync_tuned   = rc == SPMV_OK ? best.sync : 0;
    m->pb_tuned_key    = rc == SPMV_OK ? key : 0;
    return rc;
}

int csr_panel_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y)
{
    return panel_launch(ctx, A, x, y, false, apply_extra{});
}
int csr_panel_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex)
{
    return panel_launch(ctx, A, x, y, false, ex);
}

static int panel_launch(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, bool trial, const apply_extra& ex)
{
    if (!A->pb_val) SPMV_FAIL(SPMV_ERR_INVALID, "panel kernel selected but its layout was not built");
    // the fullest group's accumulators (+ the spare one the pads of the packed layout add into)
    const size_t lds = ((size_t)A->pb_max_rows + (A->pb_pack ? 1 : 0)) * sizeof(double);
    // two workgroups share a CU when their accumulators fit twice into the 160 KiB LDS
    const int per_cu = (lds <= 80000 && A->pb_two_per_cu) ? 2 : 1;
    const int grid   = std::min(A->pb_ngroups, kNumCu * per_cu);
    // chunks of 2, 4 or 8 x 1024 entries.  16 existed through round 3: every instance of it spilled registers to scratch and
    // ran slower (C2: 1.70 ms against 1.13, the x window of a chunk leaves L2); a request for 16 runs 8.
    const int unroll_rq = A->pb_unroll > 0 ? A->pb_unroll : (A->pb_unroll_tuned > 0 ? A->pb_unroll_tuned : 8);
    const int unroll    = unroll_rq >= 16 ? 8 : (unroll_rq >= 8 ? 8 : (unroll_rq >= 4 ? 4 : 2));
    const int sync_rq   = (A->pb_sync >= 0 ? A->pb_sync : A->pb_sync_tuned) & 3;
    const int sync      = sync_rq == 2 ? 3 : sync_rq;  // (2 was the split barrier through an LDS counter: measured no better than 3, deleted in round 5)
    const int layout    = A->pb_pack ? (A->pb_pair ? 4 : 3) : 0;
    // the kernel dereferences exactly these arrays: refuse on the host rather than fault on the GPU
    const bool have = layout >= 3 ? (A->pb_pack && A->pb_sbase && A->pb_soff && A->pb_val && A->pb_rowbits > 0 && A->pb_rowbits < 32)
                                  : (A->pb_col && A->pb_row && A->pb_val);
    if (!A->pb_gstart || !x || !y || !have)
        SPMV_FAIL(SPMV_ERR_INVALID, "panel kernel: layout %d is selected but its arrays are not there", layout);
    const int32_t* arg_col = layout >= 3 ? (const int32_t*)A->pb_pack : A->pb_col;
    const int      pipe_rq = A->pb_pipe >= 0 ? A->pb_pipe : (A->pb_pipe_tuned > 0 ? A->pb_pipe_tuned : 1);
    const int      pipe    = std::max(0, std::min(pipe_rq, 2));
#define SPMV_PANEL_PP(U, LY, OR, TR, SY)                                                                                               \
    {                                                                                                                                  \
        static std::atomic<unsigned long long> granted{0}; /* bit per device */                                                        \
        if (!((granted.load(std::memory_order_relaxed) >> ctx->device) & 1ull))                                                        \
        {                                                                                                                              \
            SPMV_HIP(hipFuncSetAttribute((const void*)csr_panel_pp_kernel<U, LY, OR, TR, SY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160008)); \
            granted.fetch_or(1ull << ctx->device, std::memory_order_relaxed);                                                          \
        }                                                                                                                              \
        hipLaunchKernelGGL((csr_panel_pp_kernel<U, LY, OR, TR, SY>), dim3(grid), dim3(kPanelThreads), lds, ctx->stream, A->pb_gstart,  \
                           A->pb_ngroups, A->a, arg_col, A->pb_row, A->pb_val, x, y, A->pb_sbase, A->pb_soff, A->pb_rowbits,           \
                           ex.overwrite ? 1 : 0, ex.dot_w, ex.dot_out);                                                                \
        SPMV_HIP(hipGetLastError());                                                                                                   \
        return SPMV_OK;                                                                                                                \
    }
#define SPMV_PANEL_PP_SY(U, LY, OR)                                                                        \
    if (unroll == U && layout == LY && pipe == OR)                                                         \
    {                                                                                                      \
        if (sync == 0) { if (trial) SPMV_PANEL_PP(U, LY, OR, true, 0) SPMV_PANEL_PP(U, LY, OR, false, 0) } \
        if (sync == 1) { if (trial) SPMV_PANEL_PP(U, LY, OR, true, 1) SPMV_PANEL_PP(U, LY, OR, false, 1) } \
        if (sync == 3) { if (trial) SPMV_PANEL_PP(U, LY, OR, true, 3) SPMV_PANEL_PP(U, LY, OR, false, 3) } \
    }
#define SPMV_PANEL_PP_OR(U, LY) SPMV_PANEL_PP_SY(U, LY, 0) SPMV_PANEL_PP_SY(U, LY, 1) SPMV_PANEL_PP_SY(U, LY, 2)
    SPMV_PANEL_PP_OR(8, 4) SPMV_PANEL_PP_OR(4, 4) SPMV_PANEL_PP_OR(2, 4)
    SPMV_PANEL_PP_OR(8, 3) SPMV_PANEL_PP_OR(4, 3) SPMV_PANEL_PP_OR(2, 3)
    SPMV_PANEL_PP_OR(8, 0) SPMV_PANEL_PP_OR(4, 0) SPMV_PANEL_PP_OR(2, 0)
#undef SPMV_PANEL_PP_OR
#undef SPMV_PANEL_PP_SY
#undef SPMV_PANEL_PP
    SPMV_FAIL(SPMV_ERR_INVALID, "panel kernel: unroll=%d layout=%d order=%d sync=%d is not instantiated", unroll, layout, pipe, sync);
}
}  // namespace spmv
