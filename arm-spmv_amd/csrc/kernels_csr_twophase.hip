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
//            of one RUN (panel p, group g) are contiguous in both;
//   phase B (reduce, y-stationary)  one workgroup owns a row GROUP (<= 20000 rows, accumulators in LDS) and streams
//            its stretch of xg with a 2-byte local row per entry, flat from the first entry to the last; products go
//            into LDS with ds_add_f64 and y is touched once at the end.
// Bytes per entry: A reads 2 + 8 (+ 0.5 for the table), writes 8; B reads 8 + 2: 28.5.  No gather ever leaves LDS, so
// there is nothing to keep in step and no dependence on where the columns fall; x is read once.  The scatter between
// the two orders is carried by phase A's stores (16-byte pairs, consecutive lanes consecutive pairs inside a run), so
// both phases read flat streams.  Worth it when the sweeps would cost more than the 16 extra bytes per entry: see
// csr_twophase_worth().
//
// Runs and padding (round 3).  Both phases move PAIRS of entries (16-byte values / products, 4-byte index pairs).  A run
// is padded to a multiple of 8 entries - whole 64-byte pieces of the product stream, the size of the write requests the
// L2 sends to memory - and to at least 16, which puts at most one run boundary into any 16 entries (a LINE of the source
// order).  Round 2 padded to whole 128-byte lines (+7.5 entries per run: 4.8 % of every stream on the C5 shard, whose
// runs hold 156 entries); 8 costs 2.2 %.  Padding to pairs only (0.3 %) was built and measured too: slower by 0.15 ms in
// phase A - run ends then share 64-byte pieces with their neighbours, which other workgroups write at other times, and
// partial pieces cost the memory side a read-modify-write (profiles/r03_tune_twophase_run_padding.txt).
// What the coarser padding bought was a one-word table (destination line of a source line); the table is now two words
// per source line:
//   word 0 = (delta0 << 3) | bpos     word 1 = delta1
// destination pair = source pair + delta0 for the pairs of the line before position bpos, + delta1 from bpos on (bpos = 0:
// no boundary inside the line).  Eight lanes share one 8-byte load.  delta0 has 29 bits: the layout holds < 2^29 entries.
// Every wavefront access of either phase covers whole 128-byte lines of the arrays it reads: panels and group stretches
// are walked from the 512-byte stretch (32 pairs) that holds their first pair, so that the loads of the 4-byte index pairs
// cover whole lines too (phase B 0.58 -> 0.545 ms against walks aligned to the products' lines only).
//
// Layout (built once per handle, like the reference's shard construction before its timed loop, src/mat_vec.cpp:240-268):
//   tp_val[e], tp_col[e] (uint16: column - panel base)     e in (panel, group) order, padding: value 0, column 0
//   tp_row[e'] (uint16: row - group base; 0xFFFF = padding) e' in (group, panel) order
//   tp_blk[2 * (e / 16) + {0, 1}]  the two table words of every source line
//   tp_panel_ptr[P + 1], tp_group_ptr[G + 1]     first entry of every panel in e, of every group in e'
//   tp_piece[i][...]          the stream between the phases (scratch owned by the handle), in e' order, in pieces of 2^26
//                             pairs (1 GB): pair t lives at tp_piece[t >> 26] + (t & (2^26 - 1)); see tp_choose_pieces()
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "placement_math.hpp"
#include "wave.hpp"

namespace spmv
{
namespace
{
constexpr int kTpPanelCols = 20000;  // 160,000 B of x in LDS
constexpr int kTpGroupRows = 20000;  // 160,000 B of accumulators in LDS
constexpr int kTpMinRun    = 16;     // entries: a non-empty run is at least one source line long
constexpr int kTpPad       = 8;      // entries: runs are padded to whole 64-byte pieces of the product stream
constexpr int kTpLine      = 16;     // entries per source line (8 pairs: what one table entry describes)
constexpr unsigned kTpPadRow = 0xFFFFu;
constexpr int64_t kTpMaxPadded = (int64_t)1 << 29;  // delta0 keeps 29 bits
// The product stream lives in PIECES: plain allocations of 2^kTpPieceShift pairs (1 GB), at most kTpMaxPieces of them
// (the layout holds < 2^28 pairs).  Both phases look the piece of a pair up in a table of base addresses that sits in
// front of the x panel / the accumulators in LDS.  Why pieces: see tp_choose_pieces().
constexpr int kTpPieceShift = 26;  // pairs per piece, log2: 2^26 x 16 B = 1 GB
constexpr int kTpMaxPieces  = 4;
constexpr int kTpTabDoubles = 16;  // the table's place in LDS: 128 bytes, so that what follows keeps its 16-byte alignment
struct tp_piece_tab
{
    double* base[kTpMaxPieces];
};

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
using i32x2 = int __attribute__((ext_vector_type(2)));

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

// run lengths with their padding (even, at least kTpMinRun; an empty run stays empty), in both orders
__global__ __launch_bounds__(kBlock) void tp_pad_kernel(int ngroups, int P, int pad /* 2, 8 or 16 */,
                                                        int32_t* __restrict__ count_pg /* [P*G + 1], in place */,
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
        const int n = count_pg[i];
        const int c = n == 0 ? 0 : max(kTpMinRun, (n + pad - 1) / pad * pad);
        count_pg[i] = c;
        count_gp[(size_t)g * P + p] = c;
    }
}

// per run: the table words of the source lines it touches (delta of the run covering a line's first pair -> word 0;
// a run that begins inside a line -> its position and delta, packed by tp_pack_kernel) + where panels (source order)
// and groups (destination order) begin
__global__ __launch_bounds__(kBlock) void tp_tables_kernel(int ngroups, int P, const int32_t* __restrict__ start_pg /* [P*G + 1] */,
                                                           const int32_t* __restrict__ start_gp /* [G*P + 1] */,
                                                           int32_t* __restrict__ tbl /* 2 per line */, int32_t* __restrict__ bpos /* per line */,
                                                           int32_t* __restrict__ panel_ptr /* [P+1] */, int32_t* __restrict__ group_ptr /* [G+1] */)
{
    const int64_t total = (int64_t)ngroups * P;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock)
    {
        const int p = (int)(i / ngroups), g = (int)(i % ngroups);
        const int sp = start_pg[i] / 2, np = (start_pg[i + 1] - start_pg[i]) / 2;  // pairs
        const int delta = start_gp[(size_t)g * P + p] / 2 - sp;
        if (np > 0)
            for (int L = sp >> 3; L <= (sp + np - 1) >> 3; ++L)
            {
                if ((L << 3) >= sp)
                    tbl[2 * (size_t)L] = delta;
                else
                {
                    bpos[L]                = sp - (L << 3);  // 1..7
                    tbl[2 * (size_t)L + 1] = delta;
                }
            }
        if (g == 0) panel_ptr[p] = start_pg[i];
        if (p == 0) group_ptr[g] = start_gp[(size_t)g * P];
        if (i == total - 1)
        {
            panel_ptr[P]       = start_pg[total];
            group_ptr[ngroups] = start_gp[total];
        }
    }
}
__global__ __launch_bounds__(kBlock) void tp_pack_kernel(int64_t nlines, int32_t* __restrict__ tbl, const int32_t* __restrict__ bpos)
{
    for (int64_t L = (int64_t)blockIdx.x * kBlock + threadIdx.x; L < nlines; L += (int64_t)gridDim.x * kBlock)
        tbl[2 * L] = (int32_t)(((uint32_t)tbl[2 * L] << 3) | (uint32_t)bpos[L]);
}

// ---- phase A: xg[dst(e)] = tp_val[e] * x[panel base + tp_col[e]] ----------------------------------------------------
// Two entries per lane: 4-byte index loads, 16-byte value loads and 16-byte product stores, the table words of a source
// line in one 8-byte load shared by its eight lanes.  A workgroup treats its panels as ONE stream of sets of
// THREADS x UNROLL pairs: the set that follows the one being multiplied is always in flight, also when it belongs to the
// next panel, and the next panel's x waits in registers (XR 16-byte pairs per lane, loaded when the current panel starts)
// until the barrier pair that swaps it into LDS - so neither the HBM latency nor the 160 KB of x is waited for at a panel
// change (round 2 stopped the stream there: 1.31 -> 1.27 ms on the C5 shard, profiles/r03_tune_twophase_phases.txt).
// rotate: workgroup b starts b / 256 of the way through each panel and wraps around, so that at any moment the
// workgroups write into different row groups' stretches of xg instead of all into the same one.
// Control flow is workgroup-uniform throughout.
// ALIGNED: x is 16-byte aligned (every vector the engine allocates is; a wrapped device pointer may not be) - its own
// instance, so that the aligned kernel carries neither the 8-byte path nor its hoisted index arithmetic.
template <int THREADS, int UNROLL, bool ALIGNED>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS / 256, THREADS / 256))) void tp_expand_kernel(int P, int pcols, int ncol, const int32_t* __restrict__ panel_ptr,
                                                            const unsigned short* __restrict__ tp_col, const double* __restrict__ tp_val,
                                                            const int32_t* __restrict__ tp_blk, const double* __restrict__ x,
                                                            const tp_piece_tab pieces, int rotate)
{
    extern __shared__ double lds_a[];  // [kTpTabDoubles] piece table | [pcols] entries of x
    char**  tab = reinterpret_cast<char**>(lds_a);
    double* xs  = lds_a + kTpTabDoubles;
    const u16x2* __restrict__ c2 = reinterpret_cast<const u16x2*>(tp_col);
    const f64x2* __restrict__ v2 = reinterpret_cast<const f64x2*>(tp_val);
    const i32x2* __restrict__ b2 = reinterpret_cast<const i32x2*>(tp_blk);
    f64x2*              s2       = reinterpret_cast<f64x2*>(xs);
    if (threadIdx.x < kTpMaxPieces) tab[threadIdx.x] = reinterpret_cast<char*>(pieces.base[threadIdx.x]);  // (before the first barrier)
    constexpr int SET = THREADS * UNROLL;
    constexpr int XR  = (kTpPanelCols / 2 + THREADS - 1) / THREADS;  // 16-byte pairs of x per lane
    struct Pan
    {
        int p, t_begin, len, skip, rot;  // panel, where its walk starts (a 32-pair boundary), pairs from there, pairs walked before the panel's first, rotation
    };
    const int  tid     = (int)threadIdx.x;
    // (panel bases are even: x + p * pcols keeps x's alignment)
    // the next panel of this workgroup that holds entries, from q on (panels without entries need no x either)
    // A panel's pairs are walked from the 32-pair boundary before its first pair (512 bytes of values, 128 of column pairs;
    // the pairs before the panel are skipped in emit), and the rotation is a multiple of 32 pairs too: every wavefront load
    // covers whole 128-byte lines of all three source arrays.
    auto find = [&](int q) -> Pan {
        for (; q < P; q += (int)gridDim.x)
        {
            const int tb = panel_ptr[q] / 2, te = panel_ptr[q + 1] / 2;
            if (te > tb)
            {
                const int t0  = tb & ~31;
                const int len = te - t0;
                // rotate = number of distinct starting points: 256 (default: every workgroup its own), 0 / 1 none, else workgroup b
                // takes phase b % rotate (experiments: "twophase_rotate")
                const int ph = rotate > 1 ? (int)(blockIdx.x % (unsigned)rotate) : 0;
                return Pan{q, t0, len, tb - t0, rotate > 1 ? (int)(((int64_t)len * ph / rotate) & ~31) : 0};
            }
        }
        return Pan{P, 0, 0, 0, 0};
    };
    f64x2  xr[XR];
    double xr_last = 0.0;
    auto   load_x  = [&](int p) {
        const int     c0 = p * pcols, n = min(pcols, ncol - c0), pairs = n / 2;
        const double* xp = x + c0;
        if constexpr (ALIGNED)
        {
            const f64x2* x2 = reinterpret_cast<const f64x2*>(xp);
#pragma unroll
            for (int k = 0; k < XR; ++k)
            {
                const int i = k * THREADS + tid;
                if (i < pairs) xr[k] = x2[i];
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < XR; ++k)
            {
                const int i = k * THREADS + tid;
                if (i < pairs)
                {
                    xr[k].x = xp[2 * i];
                    xr[k].y = xp[2 * i + 1];
                }
            }
        }
        if ((n & 1) && tid == 0) xr_last = xp[n - 1];
    };
    auto store_x = [&](int p) {
        const int c0 = p * pcols, n = min(pcols, ncol - c0), pairs = n / 2;
#pragma unroll
        for (int k = 0; k < XR; ++k)
        {
            const int i = k * THREADS + tid;
            if (i < pairs) s2[i] = xr[k];
        }
        if ((n & 1) && tid == 0) xs[n - 1] = xr_last;
    };
    u16x2 c[2][UNROLL];
    f64x2 v[2][UNROLL];
    i32x2 d[2][UNROLL];
    auto  phys  = [](const Pan& a, int u) { return a.t_begin + (u + a.rot >= a.len ? u + a.rot - a.len : u + a.rot); };
    // Addresses are an SGPR base + a 32-bit BYTE offset per lane (the layout holds < 2^28 pairs: 16 B x pair < 2^32), so a
    // load costs one address register instead of a 64-bit pair and no 64-bit adds (round 3's 64-bit addressing spilled 24
    // registers at the panel change; tools/kernel_resources.py --check now fails the build when this kernel uses scratch).
    const char* cb = reinterpret_cast<const char*>(c2);
    const char* vb = reinterpret_cast<const char*>(v2);
    const char* bb = reinterpret_cast<const char*>(b2);
    auto  fetch = [&](const Pan& a, int u0, u16x2(&cc)[UNROLL], f64x2(&vv)[UNROLL], i32x2(&dd)[UNROLL]) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
        {
            const uint32_t t = (uint32_t)phys(a, min(u0 + k * THREADS + tid, a.len - 1));  // past the end: re-read the last pair
            cc[k]            = __builtin_nontemporal_load(reinterpret_cast<const u16x2*>(cb + (t << 2)));
            vv[k]            = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(vb + (t << 4)));
            dd[k]            = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(bb + ((t >> 3) << 3)));
        }
    };
    auto emit = [&](const Pan& a, int u0, const u16x2(&cc)[UNROLL], const f64x2(&vv)[UNROLL], const i32x2(&dd)[UNROLL]) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
        {
            const int u = u0 + k * THREADS + tid;
            const int t = phys(a, u);
            if (u < a.len && t >= a.t_begin + a.skip)
            {
                const int      bpos = dd[k].x & 7;
                const uint32_t dst  = (uint32_t)(t + ((bpos != 0 && (t & 7) >= bpos) ? dd[k].y : (dd[k].x >> 3)));
                f64x2          o;
                o.x = vv[k].x * xs[cc[k].x];
                o.y = vv[k].y * xs[cc[k].y];
                // the pair's piece from the table in LDS, its place inside the piece in 28 bits.  (A scalar base picked with
                // readfirstlane + ballot where a wavefront writes into one piece - nearly always - and an SGPR-base store were
                // measured: phase A 1.43 ms against 1.22 with the look-up for every pair; round 4, not kept.)
                char* const ob = tab[dst >> kTpPieceShift];
                // (nontemporal or plain stores: the same time in fast and slow placements alike, profiles/r04_probe_twophase_classes.txt)
                __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(ob + ((dst & ((1u << kTpPieceShift) - 1u)) << 4)));
            }
        }
    };
    Pan cur = find((int)blockIdx.x);
    if (cur.p >= P) return;
    fetch(cur, 0, c[0], v[0], d[0]);
    load_x(cur.p);
    store_x(cur.p);
    Pan nxt = find(cur.p + (int)gridDim.x);
    if (nxt.p < P) load_x(nxt.p);
    __syncthreads();
    int u = 0;
    // one step: the set after (cur, u) goes into `f*`, (cur, u) is multiplied out of `e*`; false when nothing follows
    auto step = [&](const u16x2(&ec)[UNROLL], const f64x2(&ev)[UNROLL], const i32x2(&ed)[UNROLL], u16x2(&fc)[UNROLL], f64x2(&fv)[UNROLL],
                    i32x2(&fd)[UNROLL]) -> bool {
        const bool cross = u + SET >= cur.len;
        const bool last  = cross && nxt.p >= P;
        if (!last)
        {
            if (cross)
                fetch(nxt, 0, fc, fv, fd);
            else
                fetch(cur, u + SET, fc, fv, fd);
        }
        emit(cur, u, ec, ev, ed);
        if (last) return false;
        if (cross)
        {
            __syncthreads();  // every wavefront is done with the current panel's x
            store_x(nxt.p);
            cur = nxt;
            nxt = find(cur.p + (int)gridDim.x);
            if (nxt.p < P) load_x(nxt.p);
            __syncthreads();
            u = 0;
        }
        else
            u += SET;
        return true;
    };
    for (;;)
    {
        if (!step(c[0], v[0], d[0], c[1], v[1], d[1])) break;
        if (!step(c[1], v[1], d[1], c[0], v[0], d[0])) break;
    }
}

// ---- phase B: y[group] += the group's stretch of the product stream ---------------------------------------------------
// Flat: a lane takes pairs t, t + 1024, ... of the stretch; 16-byte product loads, 4-byte row loads, the next set in
// flight while this one goes into LDS.  Runs at the rate of its reads (3.5 GB in 0.57-0.60 ms on the C5 shard; with the
// LDS atomics replaced by plain adds, a timing experiment, it takes the same time: profiles/r03_tune_twophase_phases.txt).
constexpr int kTpThreads    = 1024;
constexpr int kReduceUnroll = 8;
__global__ __launch_bounds__(kTpThreads) void tp_reduce_kernel(const int32_t* __restrict__ gstart, int ngroups,
                                                               const int32_t* __restrict__ group_ptr,
                                                               const unsigned short* __restrict__ tp_row, const tp_piece_tab pieces,
                                                               double* __restrict__ y, int overwrite, const double* __restrict__ dot_w,
                                                               double* __restrict__ dot_out)
{
    extern __shared__ double lds_b[];  // [kTpTabDoubles] piece table | accumulators
    const char** tab = reinterpret_cast<const char**>(lds_b);
    double*      acc = lds_b + kTpTabDoubles;
    const int lane = threadIdx.x & 63;
    const u16x2* __restrict__ r2 = reinterpret_cast<const u16x2*>(tp_row);
    if (threadIdx.x < kTpMaxPieces) tab[threadIdx.x] = reinterpret_cast<const char*>(pieces.base[threadIdx.x]);
    __syncthreads();
    constexpr int U = kReduceUnroll, SET = kTpThreads * U;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        const int r0   = gstart[g];
        const int rows = gstart[g + 1] - r0;
        // the stretch is walked from the 32-pair boundary before its first pair (the pairs before it belong to the previous
        // group and are skipped): every wavefront load covers whole lines of the products AND of the 4-byte row pairs
        const int t_first = group_ptr[g] / 2, t_end = group_ptr[g + 1] / 2;
        const int t_begin = t_first & ~31;
        u16x2 r[2][U];
        f64x2 v[2][U];
        auto  fetch = [&](int t0, u16x2(&rr)[U], f64x2(&vv)[U]) {
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int t = min(t0 + k * kTpThreads + (int)threadIdx.x, t_end - 1);
                rr[k]       = __builtin_nontemporal_load(r2 + t);
                vv[k]       = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(tab[(unsigned)t >> kTpPieceShift] + (((unsigned)t & ((1u << kTpPieceShift) - 1u)) << 4)));
            }
        };
        auto add = [&](int t0, const u16x2(&rr)[U], const f64x2(&vv)[U]) {
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                const int t = t0 + k * kTpThreads + (int)threadIdx.x;
                if (t < t_end && t >= t_first)
                {
                    if (rr[k].x != kTpPadRow) atomicAdd(&acc[rr[k].x], vv[k].x);  // ds_add_f64
                    if (rr[k].y != kTpPadRow) atomicAdd(&acc[rr[k].y], vv[k].y);
                }
            }
        };
        if (t_first < t_end) fetch(t_begin, r[0], v[0]);
        for (int i = threadIdx.x; i < rows; i += kTpThreads) acc[i] = 0.0;
        __syncthreads();
        for (int t0 = t_begin; t0 < t_end && t_first < t_end; t0 += 2 * SET)
        {
            fetch(t0 + SET, r[1], v[1]);
            add(t0, r[0], v[0]);
            fetch(t0 + 2 * SET, r[0], v[0]);
            add(t0 + SET, r[1], v[1]);
        }
        __syncthreads();
        // (batching the loads of y here was measured: 0.5 % of the phase, profiles/r03_tune_writeback_batches.txt - not kept)
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

struct tp_pool  // experiments: pieces held besides (and including) the stream's
{
    std::vector<double*> piece;
};

void csr_twophase_free(spmv_mat* m)
{
    if (m->tp_pool)  // (experiments) the pool owns every piece, the stream's current ones included
    {
        auto* pool = (tp_pool*)m->tp_pool;
        for (double* p : pool->piece) (void)hipFree(p);
        for (int i = 0; i < kTpMaxPieces; ++i) m->tp_piece[i] = nullptr;
        delete pool;
        m->tp_pool = nullptr;
    }
    for (int i = 0; i < kTpMaxPieces; ++i)
    {
        if (m->tp_piece[i] && !m->tp_piece_carved[i]) (void)hipFree(m->tp_piece[i]);  // (a carved piece lies inside a parent, below)
        m->tp_piece[i]        = nullptr;
        m->tp_piece_carved[i] = false;
    }
    for (void*& parent : m->tp_parent)
        if (parent)
        {
            (void)hipFree(parent);
            parent = nullptr;
        }
    m->tp_npieces = 0;
    for (void** p : {(void**)&m->tp_val, (void**)&m->tp_col, (void**)&m->tp_row, (void**)&m->tp_blk, (void**)&m->tp_panel_ptr,
                     (void**)&m->tp_group_ptr, (void**)&m->tp_gstart})
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
int tp_panel_cols(const spmv_mat* m)  // even: the panels of x are moved in 16-byte pairs
{
    return m->tp_pcols_req > 0 ? std::max(2, std::min(m->tp_pcols_req, kTpPanelCols) & ~1) : kTpPanelCols;
}
}  // namespace

// The panel kernel sweeps x once per XCD and round (8 x rounds x 8 ncol bytes over the fabric); the two phases move
// ~20 more bytes per entry than it and read x once.  Measured on 10M rows x 32 (profiles/r02_tune_csr_c5_*): panel
// 1.13 / 1.63 / 1.98 / 2.37 ms at ncol = 10M / 20M / 40M / 80M against a flat 1.8-1.9 ms: the two phases win from
// ~3x more columns than rows on, i.e. when the sweeps exceed 12 bytes per entry.  And the runs (panel x row group)
// must be long enough that their padding (to an even count, at least 16) costs little.
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
    if ((double)m->nnz < 48.0 * runs) return false;
    if ((double)m->nnz + kTpMinRun * runs >= (double)kTpMaxPadded || runs >= 134217728.0) return false;  // the table's delta keeps 29 bits
    return sweeps > 12.0 * (double)m->nnz;
}

namespace
{
// the expand instances that can be launched: {threads, pairs per lane in flight}; the first is the default
template <int THREADS, int UNROLL>
void tp_expand_launch(spmv_ctx* ctx, const spmv_mat* A, const double* x, const tp_piece_tab& tab)
{
    const size_t xlds = sizeof(double) * ((size_t)A->tp_pcols + kTpTabDoubles);
    const dim3   grid((unsigned)std::min(A->tp_panels, kNumCu));
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL((tp_expand_kernel<THREADS, UNROLL, true>), grid, dim3(THREADS), xlds, ctx->stream, A->tp_panels, A->tp_pcols, A->ncol,
                           A->tp_panel_ptr, (const unsigned short*)A->tp_col, (const double*)A->tp_val, (const int32_t*)A->tp_blk, x, tab,
                           (int)A->tp_rotate);
    else
        hipLaunchKernelGGL((tp_expand_kernel<THREADS, UNROLL, false>), grid, dim3(THREADS), xlds, ctx->stream, A->tp_panels, A->tp_pcols, A->ncol,
                           A->tp_panel_ptr, (const unsigned short*)A->tp_col, (const double*)A->tp_val, (const int32_t*)A->tp_blk, x, tab,
                           (int)A->tp_rotate);
}
template <int THREADS, int UNROLL>
void tp_expand_grant()
{
    (void)hipFuncSetAttribute((const void*)tp_expand_kernel<THREADS, UNROLL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160008 + 8 * kTpTabDoubles);
    (void)hipFuncSetAttribute((const void*)tp_expand_kernel<THREADS, UNROLL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160008 + 8 * kTpTabDoubles);
}

void tp_grant_lds(spmv_ctx* ctx)
{
    static std::atomic<unsigned long long> granted{0};  // bit per device
    if ((granted.load(std::memory_order_relaxed) >> ctx->device) & 1ull) return;
    tp_expand_grant<1024, 3>();
    (void)hipFuncSetAttribute((const void*)tp_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160008 + 8 * kTpTabDoubles);
    granted.fetch_or(1ull << ctx->device, std::memory_order_relaxed);
}

// One instance: 1024 threads, 3 pairs per lane in flight.  Round 4 measured 1024 x 4, 512 x 6 and 512 x 8 (all without
// scratch once the addressing was 32-bit) on the C5 shard: 1.31-1.35 ms where this one takes 1.32 in the same slow
// placement - the phase runs at the memory's rate whatever the shape (profiles/r04_tune_twophase_expand_shapes.txt).
tp_piece_tab tp_table_of(const spmv_mat* A)
{
    tp_piece_tab t{};
    for (int i = 0; i < kTpMaxPieces; ++i) t.base[i] = A->tp_piece[i];
    return t;
}
void tp_launch_expand(spmv_ctx* ctx, const spmv_mat* A, const double* x, const tp_piece_tab& tab)
{
    tp_expand_launch<1024, 3>(ctx, A, x, tab);
}

void tp_launch_reduce(spmv_ctx* ctx, const spmv_mat* A, double* y, const apply_extra& ex, const tp_piece_tab& tab)
{
    hipLaunchKernelGGL(tp_reduce_kernel, dim3((unsigned)std::min(A->tp_ngroups, kNumCu)), dim3(kTpThreads),
                       sizeof(double) * ((size_t)A->tp_max_rows + kTpTabDoubles), ctx->stream, A->tp_gstart, A->tp_ngroups, A->tp_group_ptr,
                       (const unsigned short*)A->tp_row, tab, y, ex.overwrite ? 1 : 0, ex.dot_w, ex.dot_out);
}

// Where the PRODUCT stream lies in PHYSICAL memory decides a tenth of the product.  With everything else in place, phase A
// takes 1.14-1.17 ms on the C5 shard on some memory and 1.32-1.34 ms on other (phase B, which reads the products back,
// 0.565 against 0.548: the opposite way, a fifth as much); nothing else matters - not where the values, columns, rows, table,
// x or y lie, not the virtual address, not how the memory was allocated, not the translation (the same 2.4e5 UTCL1 misses
// of 1.5e8 requests either way), not nontemporal against plain stores.  Round 4 found the rule
// (profiles/r04_probe_twophase_classes.txt, r04_probe_twophase_placement_cause.txt, r04_pmc_twophase_placement.txt): the
// device's memory comes in THREE CLASSES - gigabytes of one class are slow beside each other under the stream and fast
// beside gigabytes of another; "slow together" is an equivalence relation with exactly three classes (19 / 14 / 27 of 60
// consecutively allocated gigabytes, in runs of 2 to 17) - and a stream is fast when its pieces come from different classes
// (1.70 / 1.78 / 1.835 / 1.89 ms per product from three classes down to one).  Three ranks per 12-high HBM3E stack would
// look like this: the scattered 64-byte writes of phase A lean on one rank's row activations or on three.  A process can
// neither see nor ask for the class of the memory it is handed.  (Round 4 also tried HIP's virtual memory management - one
// reserved range, windows of physical pieces mapped in turn - and found that on this ROCm (7.2) a virtual address keeps
// reaching the FIRST piece ever mapped there, after hipMemUnmap, hipMemAddressFree and a new reservation alike:
// tools/probe_vm_remap.hip.  Nothing here maps memory.)
//
// So the stream is not one allocation but PIECES of 1 GB (kernels: a table of base addresses in LDS), and which pieces it
// consists of is CHOSEN by measurement: `extra` more pieces than the stream needs are allocated (plain hipMalloc; the budget
// is m->tp_place_budget_mb, 8 GB by default, never more than a quarter of the free memory) and configurations of pieces are
// timed as they are (1 warm-up + 2 products of both phases on zeroed scratch vectors): coordinate descent - for every slot of
// the stream in turn, every piece outside the current configuration is tried in that slot and the best configuration seen
// replaces the current one when it wins by more than the noise; two rounds at most (3 slots x 8 pieces x 2 = 48 configurations
// of 6 ms each on the C5 shard: 0.3 s).  Nothing is inferred about single pieces: the time of a configuration is NOT a sum over
// its pieces (a piece that is fast beside two others can be slow beside a third, and the whole stream aliased onto one piece
// behaves differently again - profiles/r04_tune_twophase_piece_search.txt).  Every piece not kept is freed before the
// function returns.
// Transient footprint: the stream + the budget (8 GB) + scratch x and y.  Afterwards: the stream rounded up to whole pieces.
// Round 3 held up to 16 whole candidate streams with 1-4 GB spacers between them - up to 3/4 of the free memory.
// The outcome is still a draw from what the allocator hands out, and it is recorded: "twophase_placements_timed"
// (configurations timed), "twophase_placement_spread" (as built / kept, in 1/1000), "twophase_pieces_exchanged".
// "panel_trial" 0 / SPMV_PANEL_TRIAL=0 / a budget of 0: no search, no timing launches.  Streams below 512 MB: no search.
// A failing timing launch is an error (SPMV_ERR_HIP), not a silent 1e30.
// how many pieces beyond the stream's own the search may hold (0: no search; the last piece is then allocated at its exact size)
int tp_search_extra(const spmv_mat* m)
{
    const size_t bytes   = sizeof(double) * (size_t)m->tp_padded;
    const size_t piece   = (size_t)16 << kTpPieceShift;
    const char*  e_trial = getenv("SPMV_PANEL_TRIAL");  // (read when the layout is built, never on the product's path)
    const char*  e_mb    = getenv("SPMV_TP_PLACEMENT_BUDGET_MB");
    int64_t      budget_mb = m->tp_place_budget_mb >= 0 ? m->tp_place_budget_mb : (e_mb ? atoll(e_mb) : 8192);
    if (m->pb_trial == 0 || (m->pb_trial < 0 && e_trial && e_trial[0] == '0')) budget_mb = 0;
    if (budget_mb <= 0 || bytes < ((size_t)512 << 20)) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    const size_t budget = std::min((size_t)budget_mb << 20, free_b / 4);
    const int    extra  = (int)std::min<size_t>(budget / piece, 256);
    return extra >= 2 ? extra : 0;  // two reference pieces besides the stream's own are the least the scheme needs
}

// `early`: pieces the build allocated BEFORE the layout's own arrays (half the budget): together with the ones allocated here,
// behind them, the pool spans the build's ~5 GB of other allocations as well - the allocator's runs of one class are long
// (2-17 GB), so the span is what finds a second and a third class.  Consumed (kept or freed) here.
// `carve`: allocations the handle is ABOUT TO RELEASE (the CSR copy's values and column indices, at panel_keep_csr = 0): whole
// gigabytes inside them are candidates too - memory of another moment of the allocator's history at no transient cost at
// all.  A parent that ends up under the stream is kept by the layout instead of being released (tp_parent) and the pieces
// it replaced are freed; the others go back to the caller untouched (their contents are garbage afterwards: the search
// writes products into them, which is why they are only offered when they are about to be released).
struct tp_carve
{
    void*  base;
    size_t bytes;
};
int tp_choose_pieces(spmv_mat* m, std::vector<double*> early = {}, std::vector<tp_carve> carve = {}, bool fresh = true)
{
    spmv_ctx*    ctx   = m->ctx;
    const size_t piece = (size_t)16 << kTpPieceShift;
    const int    need  = m->tp_npieces;
    const int    extra = tp_search_extra(m);
    m->tp_place_seen   = 0;
    m->tp_place_gain   = 0;
    m->tp_pieces_exchanged  = 0;
    m->tp_carve_taken  = 0;
    if (extra < 2 || need <= 0 || m->tp_last_piece_bytes != (int64_t)piece)  // (pieces must be interchangeable)
    {
        for (double* p : early) (void)hipFree(p);
        return SPMV_OK;
    }
    // carved candidates: whole pieces inside the allocations on offer; parent_of[i] = index into `carve`, -1 = an allocation of its own
    std::vector<int> parent_of((size_t)need, -1);
    for (int i = 0; i < need; ++i)
        if (m->tp_piece_carved[i]) parent_of[(size_t)i] = -2;  // carved earlier: lies in a parent the layout already keeps

    hipStream_t s = ctx->stream;
    double *    x = nullptr, *y = nullptr;
    hipEvent_t  e0 = nullptr, e1 = nullptr;
    int         rc = SPMV_OK;
    std::vector<double*> cand(m->tp_piece, m->tp_piece + need);  // the stream's own pieces are candidates like the others
    cand.insert(cand.end(), early.begin(), early.end());
    parent_of.resize(cand.size(), -1);
    for (size_t k = 0; k < carve.size(); ++k)
        for (size_t off = 0; off + piece <= carve[k].bytes; off += piece)
        {
            cand.push_back((double*)((char*)carve[k].base + off));
            parent_of.push_back((int)k);
        }
    apply_extra plain;
    // ms per product of the configuration `slots` (piece index per slot; slots past `need` repeat the last: never addressed)
    auto time_config = [&](const std::vector<int>& slots, float* out) -> bool {
        tp_piece_tab t{};
        for (int i = 0; i < kTpMaxPieces; ++i) t.base[i] = cand[(size_t)slots[(size_t)std::min(i, need - 1)]];
        tp_launch_expand(ctx, m, x, t);  // warm-up
        tp_launch_reduce(ctx, m, y, plain, t);
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < 2; ++r)
        {
            tp_launch_expand(ctx, m, x, t);
            tp_launch_reduce(ctx, m, y, plain, t);
        }
        (void)hipEventRecord(e1, s);
        float t_ms = 0.f;
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t_ms, e0, e1) != hipSuccess)
        {
            set_error("two-phase layout: a timing launch of the piece search failed: %s", hipGetErrorString(hipGetLastError()));
            return false;
        }
        *out = t_ms / 2;
        return true;
    };
    const char*        e_v     = getenv("SPMV_TP_PLACEMENT_VERBOSE");
    const bool         verbose = e_v && e_v[0] == '1';
    float              t_first = 0.f, t_kept = 0.f;
    bool               searched = false;
    do
    {
        if (hipMalloc(&x, sizeof(double) * (size_t)m->ncol) != hipSuccess || hipMalloc(&y, sizeof(double) * (size_t)m->nrow) != hipSuccess ||
            hipMemsetAsync(x, 0, sizeof(double) * (size_t)m->ncol, s) != hipSuccess || hipMemsetAsync(y, 0, sizeof(double) * (size_t)m->nrow, s) != hipSuccess ||
            hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
            break;  // no room for the scratch vectors: the pieces stay as they are
        for (int i = (int)early.size(); fresh && i < extra; ++i)
        {
            double* p = nullptr;
            if (hipMalloc(&p, piece) != hipSuccess)
            {
                (void)hipGetLastError();
                break;
            }
            cand.push_back(p);
            parent_of.push_back(-1);
        }
        const int n = (int)cand.size();
        if (n < need + (carve.empty() ? 2 : 1)) break;
        // A large pool is SAMPLED: the classes of physical memory come in runs of gigabytes to tens of gigabytes, so
        // every k-th piece of a 64 GB pool reaches as far as all of them; the descent then times ~16 pieces per slot instead of
        // 64 (bench.py's grant at N > 1: 390 configurations and 4.8 s of set-up per rank in round 4, ~100 and ~1 s now).
        // The stream's own pieces and the carved ones always take part.
        std::vector<bool> in_play((size_t)n, true);
        {
            int pool = 0;
            for (int d = need; d < n; ++d) pool += parent_of[(size_t)d] == -1 ? 1 : 0;
            const int stride = pool > 24 ? (pool + 15) / 16 : 1;
            int       seen   = 0;
            for (int d = need; d < n; ++d)
                if (parent_of[(size_t)d] == -1 && (seen++ % stride) != 0) in_play[(size_t)d] = false;
        }
        tp_grant_lds(ctx);
        // the configuration the layout was built with
        std::vector<int> slots((size_t)need);
        for (int i = 0; i < need; ++i) slots[(size_t)i] = i;
        if (!time_config(slots, &t_first))
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        // Coordinate descent over REAL configurations (the time of a configuration is not a sum over its pieces - a piece that
        // is fast beside two others can be slow beside a third - so nothing is inferred, only measured): for every slot in
        // turn, every piece outside the configuration is tried in that slot; the best configuration seen replaces the
        // current one when it wins by more than the noise; two rounds, or until a round changes nothing.
        float t_cur = t_first;
        int   tried = 0;
        bool  ok    = true;
        for (int round = 0; ok && round < 2; ++round)
        {
            bool changed = false;
            for (int slot = 0; ok && slot < need; ++slot)
            {
                int   best_d = -1;
                float best_t = t_cur;
                for (int d = 0; ok && d < n; ++d)
                {
                    if (!in_play[(size_t)d] || std::find(slots.begin(), slots.end(), d) != slots.end()) continue;
                    std::vector<int> c = slots;
                    c[(size_t)slot]   = d;
                    float t           = 0.f;
                    ok                = time_config(c, &t);
                    ++tried;
                    // a carved piece keeps its whole parent allocated (2.56 GB of values for one gigabyte of stream on the C5
                    // shard): it has to be worth 2 % of the product, not the 0.3 % a fresh gigabyte has to be
                    if (parent_of[(size_t)d] >= 0) t *= 1.02f;
                    if (ok && t < best_t)
                    {
                        best_t = t;
                        best_d = d;
                    }
                }
                if (ok && best_d >= 0 && best_t < t_cur * 0.997f)
                {
                    if (verbose) fprintf(stderr, "  slot %d: piece #%d for #%d: %.4f -> %.4f ms\n", slot, best_d, slots[(size_t)slot], t_cur, best_t);
                    slots[(size_t)slot] = best_d;
                    t_cur               = best_t;
                    changed             = true;
                }
            }
            if (!changed) break;
        }
        if (!ok)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        // The verdict once more, the configuration as built and the one found timed in turn (the minimum of three each): a
        // search misled by the noise of its two-product timings falls back to the pieces the layout was built with, so what
        // the handle keeps is never the slower of the two by the last measurement taken.
        std::vector<int> built((size_t)need);
        for (int i = 0; i < need; ++i) built[(size_t)i] = i;
        t_kept = t_first;
        // tests ("twophase_offer_csr_copy" 2): a carved piece goes under slot 0 whatever the timings say, so that the kernels,
        // the accounting and the release of a stream that lies partly inside the old CSR copy are exercised on every box
        bool forced = false;
        if (m->tp_offer_csr == 2 && !carve.empty())
            for (int d = need; d < n && !forced; ++d)
                if (parent_of[(size_t)d] >= 0 && std::find(slots.begin(), slots.end(), d) == slots.end())
                {
                    bool has = false;
                    for (int sidx : slots) has = has || parent_of[(size_t)sidx] >= 0;
                    if (!has) slots[0] = d;
                    forced = true;
                }
        if (slots != built && !forced)
        {
            float t_b = 1e30f, t_k = 1e30f;
            for (int rep = 0; ok && rep < 3; ++rep)
            {
                float t = 0.f;
                ok      = time_config(built, &t);
                t_b     = std::min(t_b, t);
                ok      = ok && time_config(slots, &t);
                t_k     = std::min(t_k, t);
            }
            if (!ok)
            {
                rc = SPMV_ERR_HIP;
                break;
            }
            tried += 6;
            t_first = t_b;
            t_kept  = t_k;
            if (t_k >= t_b)
            {
                if (verbose) fprintf(stderr, "  the configuration found does not hold up (%.4f against %.4f ms as built): the pieces stay\n", t_k, t_b);
                slots  = built;
                t_kept = t_b;
            }
        }
        if (verbose)
        {
            fprintf(stderr, "two-phase product stream: %d pieces of %zu MB needed, %d to choose from, %d configurations timed; as built %.4f ms, kept", need,
                    piece >> 20, n, tried, t_first);
            for (int i = 0; i < need; ++i) fprintf(stderr, " #%d", slots[(size_t)i]);
            fprintf(stderr, ": %.4f ms\n", t_kept);
        }
        std::vector<bool> keep((size_t)n, false);
        for (int i = 0; i < need; ++i)
        {
            const int d                    = slots[(size_t)i];
            m->tp_piece[i]                 = cand[(size_t)d];
            m->tp_piece_carved[i]          = parent_of[(size_t)d] != -1;
            keep[(size_t)d]                = true;
            if (d != i) ++m->tp_pieces_exchanged;  // (pieces exchanged by the search)
            if (parent_of[(size_t)d] >= 0)
            {
                // the allocation this piece lies in stays with the layout (freed with it); once per parent
                tp_carve& c = carve[(size_t)parent_of[(size_t)d]];
                if (c.base)
                {
                    for (void*& slot : m->tp_parent)
                        if (!slot)
                        {
                            slot = c.base;
                            m->tp_bytes += (int64_t)c.bytes;
                            m->device_bytes += (int64_t)c.bytes;
                            c.base = nullptr;  // (taken: the caller must not free it)
                            break;
                        }
                }
            }
        }
        // allocations of their own under the stream: before (the first `need` candidates that are no carved pieces) and after
        int own_before = 0, own_after = 0;
        for (int d = 0; d < need; ++d) own_before += parent_of[(size_t)d] == -1 ? 1 : 0;
        for (int d = 0; d < n; ++d)
        {
            if (parent_of[(size_t)d] != -1) continue;
            if (keep[(size_t)d])
                ++own_after;
            else
                (void)hipFree(cand[(size_t)d]);
        }
        // (an own piece exchanged for a fresh one changes nothing; one that made way for a carved piece is a gigabyte less -
        // the parent's bytes were added above)
        m->tp_bytes -= (int64_t)(own_before - own_after) * (int64_t)piece;
        m->device_bytes -= (int64_t)(own_before - own_after) * (int64_t)piece;
        m->tp_carve_taken = 0;
        for (const tp_carve& c : carve) m->tp_carve_taken |= c.base ? 0 : 1 << (int)(&c - carve.data());
        m->tp_place_seen = tried;
        m->tp_place_gain = tp_spread_permille(t_first, t_kept, slots == built);
        searched         = true;
    } while (0);
    if (!searched)
        for (size_t i = (size_t)need; i < cand.size(); ++i)
            if (parent_of[i] == -1) (void)hipFree(cand[i]);  // the stream keeps the pieces it was built with
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (x) (void)hipFree(x);
    if (y) (void)hipFree(y);
    (void)hipGetLastError();
    return rc;
}
}  // namespace

// experiments ("twophase_pool_alloc" / "twophase_pool_config", SPMV_EXPERIMENTS=1): a pool of pieces held by the handle and
// any configuration of them under the stream, chosen from outside (tools/probe_twophase_pairs.py)
int csr_twophase_pool_alloc(spmv_mat* m, int extra)
{
    SPMV_REQUIRE(m->tp_val && m->tp_npieces > 0 && m->tp_last_piece_bytes == ((int64_t)16 << kTpPieceShift), "needs a built layout with whole pieces");
    for (int i = 0; i < m->tp_npieces; ++i) SPMV_REQUIRE(!m->tp_piece_carved[i], "twophase_pool_alloc: a piece of this stream lies inside the released CSR copy");
    SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
    if (!m->tp_pool)
    {
        auto* pool = new tp_pool;
        for (int i = 0; i < m->tp_npieces; ++i) pool->piece.push_back(m->tp_piece[i]);
        m->tp_pool = pool;
    }
    auto* pool = (tp_pool*)m->tp_pool;
    for (int i = 0; i < extra; ++i)
    {
        double* p = nullptr;
        if (hipMalloc(&p, (size_t)16 << kTpPieceShift) != hipSuccess)
        {
            (void)hipGetLastError();
            SPMV_FAIL(SPMV_ERR_ALLOC, "twophase_pool_alloc: out of device memory after %d pieces", i);
        }
        pool->piece.push_back(p);
    }
    return SPMV_OK;
}
int csr_twophase_pool_config(spmv_mat* m, int64_t code)  // 10 bits per slot
{
    SPMV_REQUIRE(m->tp_pool, "twophase_pool_config: no pool (twophase_pool_alloc first)");
    auto* pool = (tp_pool*)m->tp_pool;
    SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
    for (int i = 0; i < m->tp_npieces; ++i)
    {
        const size_t idx = (size_t)((code >> (10 * i)) & 1023);
        SPMV_REQUIRE(idx < pool->piece.size(), "twophase_pool_config: piece %zu of %zu", idx, pool->piece.size());
        m->tp_piece[i] = pool->piece[idx];
    }
    return SPMV_OK;
}

// "twophase_choose_pieces": run the piece search (again) on a built layout with the handle's current budget
int csr_twophase_choose_again(spmv_mat* m)
{
    SPMV_REQUIRE(m->tp_val && m->tp_padded > 0 && m->tp_npieces > 0, "the two-phase layout is not built");
    SPMV_REQUIRE(!m->tp_pool, "twophase_choose_pieces: this handle's pieces belong to an experiment's pool (twophase_pool_alloc)");
    SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
    const int64_t whole = (int64_t)16 << kTpPieceShift;
    if (m->tp_last_piece_bytes != whole && tp_search_extra(m) > 0)
    {
        // the layout was built without a search: its last piece has the stream's exact size; pieces must be interchangeable
        double* fresh = nullptr;
        if (hipMalloc(&fresh, (size_t)whole) != hipSuccess)
        {
            (void)hipGetLastError();
            SPMV_FAIL(SPMV_ERR_ALLOC, "twophase_choose_pieces: no memory for a whole last piece");
        }
        (void)hipFree(m->tp_piece[m->tp_npieces - 1]);
        m->tp_piece[m->tp_npieces - 1] = fresh;
        m->tp_bytes += whole - m->tp_last_piece_bytes;
        m->device_bytes += whole - m->tp_last_piece_bytes;
        m->tp_last_piece_bytes = whole;
    }
    return tp_choose_pieces(m);
}

// panel_keep_csr = 0 on a two-phase handle: before col_ind / values of the CSR copy are released, whole gigabytes inside them
// stand as candidates for the product stream beside the pieces it has (no fresh allocation: zero transient footprint).  Returns
// in *keep_b / *keep_v whether the layout took the allocation over (then the caller must not free it).
int csr_twophase_offer_csr_copy(spmv_mat* m, bool* keep_b, bool* keep_v)
{
    *keep_b = *keep_v = false;
    if (!m->tp_val || m->tp_npieces <= 0 || m->tp_pool || !m->b || !m->v || !m->owned || !m->tp_offer_csr) return SPMV_OK;
    if (m->tp_parent[0] || m->tp_parent[1]) return SPMV_OK;  // (once)
    SPMV_HIP(hipStreamSynchronize(m->ctx->stream));
    std::vector<tp_carve> carve{{(void*)m->v, sizeof(double) * (size_t)m->nnz}, {(void*)m->b, sizeof(int32_t) * (size_t)m->nnz}};
    const int seen0 = m->tp_place_seen, gain0 = m->tp_place_gain, exch0 = m->tp_pieces_exchanged;
    const int rc    = tp_choose_pieces(m, {}, carve, /*fresh=*/false);
    if (rc != SPMV_OK) return rc;
    *keep_v = (m->tp_carve_taken & 1) != 0;
    *keep_b = (m->tp_carve_taken & 2) != 0;
    // the record keeps both searches: configurations add up, the spread multiplies, exchanges add up
    m->tp_place_seen += seen0;
    m->tp_place_gain = gain0 > 0 && m->tp_place_gain > 0 ? (int32_t)((int64_t)gain0 * m->tp_place_gain / 1000) : std::max(gain0, m->tp_place_gain);
    m->tp_pieces_exchanged += exch0;
    return SPMV_OK;
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
    SPMV_REQUIRE(keys < ((int64_t)1 << 27) && m->nnz + keys * kTpMinRun < kTpMaxPadded,
                 "two-phase layout: %d panels x %d groups is too fine for %lld entries (or the layout would pass 2^29 entries)", P, ngroups,
                 (long long)m->nnz);
    int32_t *cnt_pg = nullptr, *cnt_gp = nullptr, *start_pg = nullptr, *start_gp = nullptr, *bpos = nullptr;
    int      rc     = SPMV_OK;
    const size_t kbytes = sizeof(int32_t) * (size_t)(keys + 1);
    // half of the piece search's extra pieces are allocated now, before the layout's arrays (see tp_choose_pieces); the count
    // comes from the unpadded entry count (the search itself decides with the padded one; a handful of pieces either way)
    std::vector<double*> early;
    {
        const int64_t saved = m->tp_padded;
        m->tp_padded        = m->nnz;
        const int extra     = tp_search_extra(m);
        m->tp_padded        = saved;
        for (int i = 0; i < extra / 2; ++i)
        {
            double* p = nullptr;
            if (hipMalloc(&p, (size_t)16 << kTpPieceShift) != hipSuccess)
            {
                (void)hipGetLastError();
                break;
            }
            early.push_back(p);
        }
    }
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
        // runs are padded to whole 64-byte pieces of the product stream (8 entries); SPMV_TP_PAD = 2 / 16: to pairs / to whole
        // 128-byte lines instead (A/B switch, see the header of this file)
        const char* e_pad = getenv("SPMV_TP_PAD");
        const int   pad   = e_pad && (atoi(e_pad) == 2 || atoi(e_pad) == 16) ? atoi(e_pad) : kTpPad;
        hipLaunchKernelGGL(tp_pad_kernel, dim3(kgrid), dim3(kBlock), 0, s, ngroups, P, pad, cnt_pg, cnt_gp);
        if ((rc = exclusive_scan_i32(ctx, cnt_pg, start_pg, keys + 1)) != SPMV_OK) break;
        if ((rc = exclusive_scan_i32(ctx, cnt_gp, start_gp, keys + 1)) != SPMV_OK) break;
        int32_t padded = 0;
        if (hipMemcpyAsync(&padded, start_pg + keys, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        if (padded < m->nnz || padded % 2 != 0 || (int64_t)padded > m->nnz + keys * kTpMinRun)
        {
            rc = SPMV_ERR_HIP;  // (the scan disagrees with the counts)
            break;
        }
        const size_t np     = (size_t)padded;
        const size_t nlines = (np + kTpLine - 1) / kTpLine;
        if (hipMalloc(&m->tp_val, sizeof(double) * np) != hipSuccess || hipMalloc(&m->tp_col, sizeof(unsigned short) * np) != hipSuccess ||
            hipMalloc(&m->tp_row, sizeof(unsigned short) * np) != hipSuccess ||
            hipMalloc(&m->tp_blk, sizeof(int32_t) * 2 * nlines) != hipSuccess || hipMalloc(&bpos, sizeof(int32_t) * nlines) != hipSuccess)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        m->tp_padded = padded;
        // the product stream: ceil(pairs / 2^26) pieces of 1 GB (the last one as well: pieces are interchangeable)
        m->tp_npieces = (int32_t)(((np / 2) + ((size_t)1 << kTpPieceShift) - 1) >> kTpPieceShift);
        bool got = m->tp_npieces <= kTpMaxPieces;
        // whole pieces when the search will run (pieces must be interchangeable), else the last one at its exact size
        const size_t whole = (size_t)16 << kTpPieceShift;
        const size_t last  = tp_search_extra(m) > 0 ? whole : (np * sizeof(double) - (size_t)(m->tp_npieces - 1) * whole + 255) / 256 * 256;
        m->tp_last_piece_bytes = (int64_t)last;
        for (int i = 0; got && i < m->tp_npieces; ++i) got = hipMalloc(&m->tp_piece[i], i + 1 < m->tp_npieces ? whole : last) == hipSuccess;
        if (!got)
        {
            rc = SPMV_ERR_ALLOC;
            break;
        }
        // padding: value 0 at column 0 of the panel on the source side, row 0xFFFF on the destination side; cnt_pg becomes the cursors
        if (hipMemsetAsync(m->tp_val, 0, sizeof(double) * np, s) != hipSuccess || hipMemsetAsync(m->tp_col, 0, sizeof(unsigned short) * np, s) != hipSuccess ||
            hipMemsetAsync(m->tp_row, 0xFF, sizeof(unsigned short) * np, s) != hipSuccess || hipMemsetAsync(cnt_pg, 0, kbytes, s) != hipSuccess ||
            hipMemsetAsync(m->tp_blk, 0, sizeof(int32_t) * 2 * nlines, s) != hipSuccess || hipMemsetAsync(bpos, 0, sizeof(int32_t) * nlines, s) != hipSuccess)
        {
            rc = SPMV_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(tp_place_kernel<false>, dim3(rgrid), dim3(kBlock), 0, s, m->nrow, ngroups, P, pcols, m->tp_gstart, m->a, m->b, m->v,
                           (int32_t*)nullptr, start_pg, start_gp, cnt_pg, m->tp_col, m->tp_row, m->tp_val);
        hipLaunchKernelGGL(tp_tables_kernel, dim3(kgrid), dim3(kBlock), 0, s, ngroups, P, start_pg, start_gp, m->tp_blk, bpos, m->tp_panel_ptr,
                           m->tp_group_ptr);
        hipLaunchKernelGGL(tp_pack_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, ceil_div((int64_t)nlines, kBlock))), dim3(kBlock), 0, s,
                           (int64_t)nlines, m->tp_blk, bpos);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SPMV_ERR_HIP;  // gstart (host) is done with
        m->tp_padded = padded;
    } while (0);
    for (int32_t* p : {cnt_pg, cnt_gp, start_pg, start_gp, bpos})
        if (p) (void)hipFree(p);
    if (rc != SPMV_OK)
    {
        (void)hipStreamSynchronize(s);
        for (double* p : early) (void)hipFree(p);
        csr_twophase_free(m);
        SPMV_FAIL(rc, "building the two-phase layout (%d groups x %d panels) failed: %s", ngroups, P, hipGetErrorString(hipGetLastError()));
    }
    m->tp_ngroups   = ngroups;
    m->tp_panels    = P;
    m->tp_pcols     = pcols;
    m->tp_max_rows  = per;
    // values 8 + columns 2 + rows 2 bytes per padded entry, the table, the pointers, and the product stream in whole pieces
    m->tp_bytes     = m->tp_padded * 12 + (m->tp_padded + kTpLine - 1) / kTpLine * 8 + (int64_t)(P + 1 + 2 * (ngroups + 1)) * 4 +
                      (int64_t)(m->tp_npieces - 1) * ((int64_t)16 << kTpPieceShift) + m->tp_last_piece_bytes;
    m->device_bytes += m->tp_bytes;
    rc = tp_choose_pieces(m, std::move(early));
    if (rc != SPMV_OK) csr_twophase_free(m);
    return rc;
}

int csr_twophase_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex)
{
    if (A->nrow == 0 || A->nnz == 0)
    {
        if (ex.overwrite && A->nrow > 0) SPMV_TRY(vec_fill(ctx, y, A->nrow, 0.0));
        return SPMV_OK;
    }
    // the kernels dereference exactly these: refuse on the host rather than fault on the GPU
    bool pieces_ok = A->tp_npieces > 0 && A->tp_npieces <= kTpMaxPieces && (int64_t)A->tp_npieces << (kTpPieceShift + 1) >= A->tp_padded;
    for (int i = 0; pieces_ok && i < A->tp_npieces; ++i) pieces_ok = A->tp_piece[i] != nullptr;
    if (!A->tp_val || !A->tp_col || !A->tp_row || !pieces_ok || !A->tp_blk || !A->tp_panel_ptr || !A->tp_group_ptr || !A->tp_gstart || !x || !y ||
        A->tp_panels <= 0 || A->tp_pcols <= 0 || A->tp_pcols > kTpPanelCols || A->tp_pcols % 2 != 0 || A->tp_max_rows > kTpGroupRows ||
        A->tp_padded % 2 != 0)
        SPMV_FAIL(SPMV_ERR_INVALID, "two-phase kernel selected but its layout was not built");
    tp_grant_lds(ctx);
    // tp_only = 1 / 2: one phase alone ("twophase_only", accepted by spmv_mat_set_param only under SPMV_EXPERIMENTS=1:
    // tools/tune_twophase.py times the phases; the RESULT IS THEN WRONG)
    const int only = A->tp_only;
    const tp_piece_tab tab = tp_table_of(A);
    if (only != 2) tp_launch_expand(ctx, A, x, tab);
    if (only != 1) tp_launch_reduce(ctx, A, y, ex, tab);
    SPMV_HIP(hipGetLastError());
    return SPMV_OK;
}
}  // namespace spmv
