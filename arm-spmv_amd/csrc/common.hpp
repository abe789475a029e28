// common.hpp — internal types shared by the translation units of libspmv_hip.so.
// Not part of the ABI (include/spmv_abi.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "spmv_abi.h"

namespace spmv
{
// ---- error plumbing -----------------------------------------------------------------------------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define SPMV_FAIL(code, ...)            \
    do                                  \
    {                                   \
        ::spmv::set_error(__VA_ARGS__); \
        return (code);                  \
    } while (0)

#define SPMV_HIP(call)                                                                     \
    do                                                                                     \
    {                                                                                      \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            SPMV_FAIL(SPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                                 \
    } while (0)

#define SPMV_TRY(call)          \
    do                          \
    {                           \
        int rc_ = (call);       \
        if (rc_ != 0) return rc_; \
    } while (0)

#define SPMV_REQUIRE(cond, ...) \
    do                          \
    {                           \
        if (!(cond)) SPMV_FAIL(SPMV_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// ---- chip constants (MI355X / gfx950) -------------------------------------------------------------
constexpr int kWave      = 64;   // wavefront width
constexpr int kBlock     = 256;  // default workgroup: 4 waves, one per SIMD
constexpr int kNumXcd    = 8;
constexpr int kNumCu     = 256;
constexpr int kMaxGrid   = kNumCu * 8;  // grid-stride cap for streaming kernels (8 blocks / CU)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- counter-based PRNG shared by the device generators and their numpy twin ------------------------
// splitmix64 finaliser; stream keys are derived on the host, the device adds the element index.
__host__ __device__ static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
enum : uint64_t
{
    kStreamCol = 1,
    kStreamVal = 2,
    kStreamVec = 3,
    kStreamLen = 4
};
static inline uint64_t stream_key(uint64_t seed, uint64_t stream)
{
    return splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (stream + 1)));
}
__host__ __device__ static inline double u64_to_unit(uint64_t r)  // [0,1)
{
    return (double)(r >> 11) * 0x1.0p-53;
}
__host__ __device__ static inline double u64_to_sym(uint64_t r)  // [-1,1)
{
    return (double)(r >> 11) * 0x1.0p-52 - 1.0;
}
__host__ __device__ static inline uint32_t u64_to_range(uint64_t r, uint32_t n)  // [0,n)
{
    return (uint32_t)(((r >> 32) * (uint64_t)n) >> 32);
}
}  // namespace spmv

// ---- the opaque ABI types ---------------------------------------------------------------------------
struct spmv_ctx
{
    int         device      = 0;
    hipStream_t stream      = nullptr;
    bool        owns_stream = false;
    hipEvent_t  ev_begin    = nullptr;
    hipEvent_t  ev_end      = nullptr;
    // small persistent scratch (dot partials, flags); grown on demand outside hot loops
    void*  scratch       = nullptr;
    size_t scratch_bytes = 0;
    double* host_pinned  = nullptr;  // 64 B of pinned host memory for scalar results
    double* dev_scalars  = nullptr;  // one slotted accumulator (kDotDoubles) for scalar results; never re-allocated
    // Start-up probe (abi.hip: xcd_probe): are the workgroups of a launch dealt round-robin over 8 XCDs, so that workgroups b
    // and b + 8 share an XCD (and its L2) and b, b + 1, ..., b + 7 sit on 8 different ones?  1 yes / 0 no (a partitioned
    // device, another dispatch order) / -1 the probe could not run.  Layouts that lean on it for SPEED (the COO scan over one
    // column bin per XCD) are not built when it is not 1; nothing leans on it for correctness.
    int32_t xcd_round_robin = -1;
    int32_t xcds_seen       = 0;
    // spmv_apply_host (abi.hip): the caller's HOST vectors staged through pinned, device-mapped host memory (small vectors: the
    // GPU copies them in and out itself, one stream, no hipMemcpy) and device buffers for x and y; grown on demand
    double* stage_pinned     = nullptr;  // host address
    double* stage_pinned_dev = nullptr;  // the same memory as the GPU sees it
    size_t  stage_pinned_n   = 0;        // doubles
    double* stage_x = nullptr;
    double* stage_y = nullptr;
    size_t  stage_x_n = 0, stage_y_n = 0;
    // spmv_ctx_set_plan: handles created on this context take this plan instead of selecting (plan.hip); `plan_armed` is set by
    // the public entry point that creates a handle and consumed by that handle's analysis (its children get theirs handed down)
    std::vector<unsigned char> plan_blob;
    bool                       plan_armed = false;
    // can the CPU store straight into device memory (large BAR: hipDeviceAttributeIsLargeBar)?  Then spmv_apply_host writes a
    // small x into its device buffer itself (80 KB in 2 us) instead of launching a kernel that pulls it over the host link;
    // hdp_flush: the HDP flush register (hipDeviceAttributeHdpMemFlushCntl), written after such stores
    int32_t            large_bar = 0;
    volatile unsigned* hdp_flush = nullptr;
    // the trial arena (select.hip): ONE allocation that the timing launches of AUTO carve their scratch vectors from, so that no
    // hipMalloc / hipFree falls between a candidate's build and its timing window; grown between trials, freed with the context
    void*  arena       = nullptr;
    size_t arena_bytes = 0;
    int    arena_users = 0;  // trials holding pointers into it (it is not re-allocated while > 0)
};

namespace spmv
{
struct symgs_plan;
// ---- plans (plan.hip): the decisions a handle's set-up made - by model or by timing - as plain data ------------------------------
// One node per handle, children by index (the row-grouped copy of a COO / CSC / ELL handle or the short-row copy of a split,
// the long rows' matrix of a split in virtual-row mode, the ELL copy of a CSR handle).  32 four-byte fields, no pointers:
// the blob of spmv_mat_get_plan is a plan_header followed by nnodes plan_node, node 0 the handle itself.
struct plan_node
{
    int32_t  format, kernel, lanes_per_row;
    uint32_t flags;
    int32_t  pb_group_rows, pb_width, pb_sort, pb_aos, pb_unroll, pb_pipe, pb_sync, pb_two_per_cu, pb_rounds;
    int32_t  split_threshold, split_mode;
    int32_t  tp_pcols, tp_rotate;
    int32_t  ell_variant, ell_tiled;
    int32_t  coo_bins_per_xcd;
    int32_t  child_rowgrouped, child_long, child_ell;  // node indices, -1: none
    int32_t  reserved[9];
};
static_assert(sizeof(plan_node) == 128, "plan_node is 32 four-byte fields");
struct plan_header
{
    uint32_t magic, version, bytes, nnodes;
};
constexpr uint32_t kPlanMagic = 0x4e4c5053u;  // "SPLN"
constexpr uint32_t kPlanVersion = 1;
}

struct spmv_vec
{
    spmv_ctx* ctx   = nullptr;
    int64_t   n     = 0;
    double*   d     = nullptr;
    bool      owned = false;
};

struct spmv_mat
{
    spmv_ctx* ctx       = nullptr;
    int32_t   format    = 0;
    int32_t   nrow      = 0;
    int32_t   ncol      = 0;
    int32_t   k         = 0;  // ELL slots / DIA ndiags
    int64_t   nnz       = 0;
    int64_t   row_begin = 0;
    // a: row_ptr | row_ind | -       | col_ptr | offsets
    // b: col_ind | col_ind | col_ind | row_ind | -
    const int32_t* a = nullptr;
    const int32_t* b = nullptr;
    const double*  v = nullptr;
    bool           owned        = false;
    int64_t        device_bytes = 0;

    // ---- analysis results (filled by analyse_*) ----
    bool    dia_off_known = false;  // DIA: smallest / largest diagonal offset (set where the offsets pass through the host)
    int32_t dia_off_min = 0, dia_off_max = 0;
    int32_t dia_col_bound = 0;  // DIA: columns >= this are skipped (0 = min(nrow, ncol)); row shards keep the global bound
    int32_t max_row_nnz   = 0;
    int32_t kernel        = SPMV_CSR_AUTO;
    int32_t lanes_per_row = 0;
    int32_t sorted_rows   = 0;
    bool    kernel_forced = false;
    uint32_t flags        = 0;  // SPMV_FLAG_* tuning bits

    // CSR LDS-window kernel: per row-block [lo, hi) column window
    int32_t  win_rows  = 0;        // rows per workgroup
    int32_t* win_lo    = nullptr;  // [nblocks] first column touched by the block
    int32_t* win_span  = nullptr;  // [nblocks] hi - lo
    int32_t  win_max_span = 0;

    double   win_avg_span = 0.0;   // mean over row blocks: how local the columns are
    double   contig_frac  = 0.0;   // CSR: fraction of the entries whose column is the previous entry's + 1 (dense blocks, bands)

    // AUTO selection by measurement (select.hip): candidates timed when the handle was analysed, microseconds per product
    // by spmv_csr_kernel id (COO / ELL: [1] the format's own kernel, [4] the row-grouped copy); 0 = not timed
    int32_t  sel_candidates = 0;
    float    sel_us[10]     = {0};
    int32_t  sel_rounds     = 0;  // rounds the last trial went through until its minima stood still ("select_rounds")

    // CSR panel kernel (kernels_csr_panel.hip): entries re-ordered per row group by column panel / x line
    int32_t*  pb_col         = nullptr;  // [nnz] global column
    uint16_t* pb_row         = nullptr;  // [nnz] row inside its group
    double*   pb_val         = nullptr;  // [nnz]
    int32_t*  pb_gstart      = nullptr;  // [ngroups + 1] first row of every group (entry-balanced unless G was requested)
    int32_t   pb_group_rows  = 0;        // G (0 = choose)
    int32_t   pb_panel_width = 0;        // W (0 = default)
    int32_t   pb_sort        = 1;        // bucket tile entries by 128-byte line of x
    int32_t   pb_unroll      = 0;        // entries in flight per lane (0 = default)
    int32_t   pb_aos         = 4;        // layout: 4 = 12-byte packed entries, slices in interleaved pairs (falls back to 0), 3 = the same
                                         // without the pairing, 0 = three arrays (14 bytes)
    uint32_t* pb_pack        = nullptr;  // [nnz] layout 3: (column - slice base) << rowbits | local row
    int32_t*  pb_sbase       = nullptr;  // [slices] layout 3: line-aligned first column of every 1024-entry slice
    int32_t*  pb_soff        = nullptr;  // [ngroups + 1] layout 3: first slice of every group
    int32_t   pb_rowbits     = 0;
    bool      pb_pair        = false;    // layout 4: slices stored in interleaved pairs
    int32_t   pb_slices      = 0;        // layout 3: 1024-entry slices in all (padded entries / 1024)
    int32_t   pb_built_layout = -1;      // pb_aos the layout in memory was built for
    int32_t   pb_pipe        = -1;      // chunk pipeline: 0 off, 1 stream-first, 2 gather-first, -1 = 1 or 2 by trial
    int32_t   pb_pipe_tuned  = 0;       // the order found by trying (in effect while pb_pipe == -1; 0 = not tried: 1)
    int32_t   pb_tuned_key   = 0;        // what the build-time trial was made for (requested unroll / order / sync); 0 = not tried
    int32_t   pb_unroll_tuned = 0;       // chunk size found by trying (in effect while pb_unroll == 0)
    int64_t   pb_max_group_nnz = 0;      // entries of the fullest row group
    int32_t   pb_sync        = -1;       // keep a workgroup's wavefronts together: 0 no, 1 barrier per chunk, 2 priority to late ones,
                                         // 3 barrier between a chunk's loads and its LDS adds; -1 = by trial
    int32_t   pb_sync_tuned  = 0;        // what the trial found (in effect while pb_sync == -1)
    int32_t   pb_trial       = -1;       // timing launches when the layout is built: 1 yes, 0 no, -1 = SPMV_PANEL_TRIAL (default yes)
    int32_t   pb_two_per_cu  = 1;        // allow two workgroups per CU when the accumulators fit twice
    int32_t   pb_ngroups     = 0;
    int32_t   pb_rounds_req  = 0;        // groups for k rounds of 256 workgroups instead of the fewest the LDS cap allows (0: one round's worth unless
                                         // skewed rows make a finer cut worth a timing, 1: never)
    int32_t   pb_built_rounds = 0;       // what the layout in memory was cut for
    float     pb_rounds_us[2] = {0.f, 0.f};  // the timing that decided: the single round, the finer cut (microseconds per product; 0 = not timed)
    int32_t   pb_max_rows    = 0;        // rows of the fullest group (sizes the LDS accumulators)
    int32_t   pb_built_rows = 0, pb_built_width = 0, pb_built_sort = -1;  // parameters of the layout in memory
    int64_t   pb_bytes       = 0;

    // CSR two-phase kernel (kernels_csr_twophase.hip): entries in (column panel, row group) order
    double*   tp_val       = nullptr;  // [padded] (panel, group) order, runs padded to 16 entries
    uint16_t* tp_col       = nullptr;  // [padded] column - panel base, same order
    uint16_t* tp_row       = nullptr;  // [padded] row - group base, (group, panel) order; 0xFFFF = padding
    double*   tp_piece[4] = {};        // the stream between the two phases, (group, panel) order, in pieces of 2^26 pairs (1 GB)
    bool      tp_piece_carved[4] = {}; // the piece lies inside tp_parent[] (an allocation taken over from the released CSR copy), not its own
    void*     tp_parent[2] = {};       // allocations of the CSR copy (values, column indices) the product stream took over at panel_keep_csr = 0
    int32_t   tp_carve_taken = 0;      // bit k: the last search took allocation k of those on offer
    int32_t   tp_offer_csr = 1;        // panel_keep_csr = 0 offers the CSR copy's gigabytes to the piece search first (0: A/B, just release them)
    int32_t   tp_npieces   = 0;
    void*     tp_pool      = nullptr;  // experiments: a pool of pieces (kernels_csr_twophase.hip: tp_pool)
    int64_t   tp_last_piece_bytes = 0;  // the last piece: a whole one when the piece search ran, else what the stream needs of it
    int32_t*  tp_blk       = nullptr;  // [2 * ceil(padded / 16)] two table words per source line: source pair -> destination pair
    int32_t*  tp_panel_ptr = nullptr;  // [panels + 1]
    int32_t*  tp_group_ptr = nullptr;  // [groups + 1]
    int32_t*  tp_gstart    = nullptr;  // [groups + 1] first row of every group
    int32_t   tp_ngroups = 0, tp_panels = 0, tp_pcols = 0, tp_max_rows = 0;
    int32_t   tp_pcols_req = 0;  // requested panel width (0 = default)
    int32_t   tp_place_budget_mb = -1;          // memory the piece search may hold beyond the stream (-1: SPMV_TP_PLACEMENT_BUDGET_MB or 8192; 0: no search)
    int32_t   tp_pieces_exchanged = 0;               // pieces of the stream the search exchanged for others
    int32_t   tp_rotate = 256;                  // expand kernel: distinct starting points of the workgroups inside their panels (0 / 1: none)
    int32_t   tp_only = 0;                      // experiment (SPMV_EXPERIMENTS=1): 1 / 2 = run phase A / B alone - wrong results
    int32_t   tp_place_seen = 0, tp_place_gain = 0;  // placements of the product stream timed at build; slowest / kept, in 1/1000
    int64_t   tp_padded = 0;
    int64_t   tp_bytes   = 0;

    // CSR segmented-scan kernel (kernels_coo.hip: csr_segscan_build): the row of every entry
    int32_t* seg_row = nullptr;  // [nnz]
    bool     sel_no_segscan = false;  // the handle is the row-grouped copy of a COO handle: that handle's own kernel IS this scan

    // CSR handle of (nearly) equal rows: an ELL copy of it with a handle - and kernels, diagonal slots included - of its own
    // (kernels_ell.hip: csr_ell_copy_build; kernel SPMV_CSR_ELL); owned
    spmv_mat* ell_copy = nullptr;
    int32_t   min_row_nnz = 0;
    bool      sel_no_ell = false;  // the handle is itself somebody's copy whose source is an ELL handle
    bool      sel_no_rowgrouped = false;  // ELL: the handle is the ELL copy of a CSR handle (no row-grouped copy of the copy)

    // CSR long-row split (kernels_csr_split.hip): the long rows in chunks over the handle's own arrays, the others in `coo_csr`
    int32_t* split_chunks = nullptr;      // [3 * nchunks] row | first entry | end, per chunk of a long row
    int32_t  split_nchunks = 0;
    int32_t  split_threshold = 0;         // rows of this many entries and more are long (0: max(4096, longest / 16))
    int32_t  split_built_threshold = 0;   // what the split in memory was built with
    bool     split_auto_low = false;      // AUTO timed a threshold of 256 faster than the default on this handle (select.hip)
    int32_t  split_long_rows = 0;
    int64_t  split_long_nnz = 0;
    bool     sel_no_split = false;        // the handle is the short-row part of a split
    int32_t  split_mode = 0;              // how the long rows run: 1 chunks of the handle's own arrays, 2 virtual rows in a matrix of their own, 0 = by their density
    int32_t  split_built_mode = 0;         // the mode in effect (1 or 2)
    int32_t  split_built_for_mode = 0;     // the "split_mode" request the split in memory was built under
    spmv_mat* split_long = nullptr;       // mode 2: the virtual rows (a CSR handle with a kernel of its own); owned
    double*  split_yl = nullptr;          // mode 2: [split_vrows] the virtual rows' sums of one product
    int32_t* split_rows = nullptr;        // mode 2: [4 * long rows] row | first entry | virtual rows V | first virtual row
    int32_t  split_vrows = 0;

    // ELL whose slots are diagonals (kernels_ell.hip: ell_detect_diagonals): slot s holds column i + off[s] in the row
    // pairs whose bit is set; the product reads no column index there.  ell_diag = off[K] | xbase[K] | clusters
    // (kernels_ell.hip: stage_x_windows); mask[wavefront * K + s] = 64 row pairs
    int32_t* ell_diag      = nullptr;
    void*    ell_diag_mask = nullptr;
    int32_t  ell_diag_lds = 0;  // doubles of LDS the x stretches of a block take (0: none, x from global memory)
    int32_t  ell_variant  = 0;  // which of the format's own kernels AUTO timed fastest: 0 two rows per lane (diagonal slots where found),
                                // 1 one row per lane, 2 two rows per lane reading every column index
    double*  ell_tval = nullptr;  // the values in tiles of 512 rows, (tile * k + slot) * 512 + row (ell_build_tiles); owned
    // ELL whose slots are diagonals, DIA-ORDER copy (kernels_ell.hip: ell_build_dia_order; ell_variant 3): the values once more
    // ROW-major, row * k + slot, multiplied by the DIA kernel (a workgroup streams one contiguous stretch and x goes through an
    // LDS window); rows in which any slot is not its diagonal are skipped there (ell_skip: a bit per row) and done by a side
    // kernel over the handle's own arrays (ell_nc_rows: their indices), in the same slot order: the same bits.  All owned.
    double*             ell_rval     = nullptr;
    unsigned long long* ell_skip     = nullptr;
    int32_t*            ell_nc_rows  = nullptr;
    int32_t             ell_nc_count = 0;
    int32_t             ell_off_min = 0, ell_off_max = 0;
    int32_t             ell_dia_order_req = -1;  // "ell_dia_order": -1 a candidate of the trial, 0 never, 1 built and used
    bool                ell_pad_marked = false;  // the ELL COPY of a CSR handle: padding slots carry a negative column and take no part in the sums (kernels_ell.hip: MASKED)

    // COO / CSC / ELL: internal row-grouped copy in the panel layout (coo_build_panel, csc_analyse, ell_build_panel); owned.
    // CSR with kernel SPLIT: the copy without the long rows
    spmv_mat* coo_csr = nullptr;

    // COO: copy of the entries in column bins, one run of bins per XCD, for the segmented scan (kernels_coo.hip:
    // coo_build_bins); bins start on workgroup chunks, the padding carries row INT32_MAX
    int32_t* cb_row = nullptr;
    int32_t* cb_col = nullptr;
    double*  cb_val = nullptr;
    int32_t  cb_bins = 0;             // bins in all (a multiple of 8), 0: no copy
    int32_t  cb_region[8] = {0};      // first chunk of the XCD's bins
    int32_t  cb_chunks[8] = {0};      // chunks of the XCD's bins
    int64_t  cb_padded = 0;           // entries of the copy with its padding

    // a plan being applied (plan.hip): the node array and this handle's node in it; set only while spmv_mat_set_plan or the
    // analysis of a handle created under spmv_ctx_set_plan runs, handed down to the copies that are built meanwhile
    const spmv::plan_node* plan_base = nullptr;
    int32_t                plan_at   = -1;

    // symmetric Gauss-Seidel (symgs.hip): L / D / U copies, rows by level, launch schedule; built by symgs_setup
    struct spmv::symgs_plan* gs = nullptr;
    int32_t                  gs_order = 1;  // sweep order: 1 multicolour (default), 0 the matrix's own row order
};

namespace spmv
{
// a launch holds fewer than 2^32 work-items: beyond that the grid wraps around without an error
inline bool launch_fits(int64_t items, int lanes_per_item) { return items * lanes_per_item < ((int64_t)1 << 32) - 4096; }
int ensure_scratch(spmv_ctx* ctx, size_t bytes);

// kernels_csr.hip
int csr_analyse(spmv_mat* m);
void csr_choose_kernel(spmv_mat* m);  // the model's pick (no launches)
int  csr_ldswin_capacity();           // columns of x the LDS-window kernel's tile holds
// select.hip: AUTO by measurement
bool select_trials_enabled(const spmv_mat* m);
int  csr_select_kernel(spmv_mat* m);
// zeroed x (ncol) and y (nrow) for timing launches; freed with the object
struct select_scratch
{
    double *x = nullptr, *y = nullptr;
    spmv_ctx* ctx        = nullptr;
    bool      from_arena = false;  // x / y lie in the context's trial arena (select.hip), not in allocations of their own
    int  alloc(spmv_ctx* ctx, int64_t ncol, int64_t nrow);
    void release();
    ~select_scratch() { release(); }
};
// candidates 0 .. n-1 timed in rounds until no minimum moves (select.hip); t[i] < 0: not timed yet
int  select_rounds(spmv_ctx* ctx, int n, const std::function<int(int)>& launch, float* t, int* rounds_run);
// ms per product of `launch`: 1 warm-up + 1 product, and 2 x 4 more (the minimum) unless that one was 3x behind best_so_far
int  select_time(spmv_ctx* ctx, const std::function<int()>& launch, float best_so_far, float* ms);
void select_note(spmv_mat* m, int slot, float ms);  // records a timed candidate (sel_us[slot], sel_candidates)
void select_reset(spmv_mat* m);
constexpr int64_t kSelectMinNnz = (int64_t)64 << 10;  // below: every kernel takes a launch latency, nothing to choose
constexpr int64_t kSelectMaxNnz = (int64_t)8 << 20;   // CSR above: the model's pick unless a statistic casts doubt on it
// kernels_csr_panel.hip
int  csr_panel_build(spmv_mat* m);
int  panel_choose_pace(spmv_mat* m);
void csr_panel_free(spmv_mat* m);
int  csr_panel_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
// kernels_csr_twophase.hip
int  csr_twophase_build(spmv_mat* m);
void csr_twophase_free(spmv_mat* m);
bool csr_twophase_worth(const spmv_mat* m);
int  csr_twophase_choose_again(spmv_mat* m);
int  csr_twophase_offer_csr_copy(spmv_mat* m, bool* keep_b, bool* keep_v);
int  csr_twophase_pool_alloc(spmv_mat* m, int extra);
int  csr_twophase_pool_config(spmv_mat* m, int64_t code);


// A sum that thousands of wavefronts add into is kept as kDotSlots partial sums on different 128-byte lines (an
// atomic on ONE word costs ~12 ns each at the L2, serialised: 8192 of them are 100 us); readers add the slots up.
constexpr int kDotSlots   = 32;
constexpr int kDotStride  = 16;                       // doubles between slots: one 128-byte line each
constexpr int kDotDoubles = kDotSlots * kDotStride;   // size of one slotted accumulator

// what a solver step wants on top of y += A*x (solver.hip)
struct apply_extra
{
    bool          overwrite = false;    // y = A*x instead of y += A*x
    const double* dot_w     = nullptr;  // if set: accumulate sum_i dot_w[i] * y_new[i] into dot_out
    double*       dot_out   = nullptr;  // slotted accumulator on the device (kDotDoubles doubles)
};
int  csr_panel_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex);
int  csr_twophase_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex);
bool csr_vector_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex, int* rc);
// solver.hip
int mat_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int mat_apply_ex(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y, const apply_extra& ex);
int cg_solve(spmv_ctx* ctx, const spmv_mat* A, const double* b, double* x, int max_iter, double rel_tol, int check_every,
             int precond, int* iters, double* rel_resid);
int vec_dot_accumulate(spmv_ctx* ctx, const double* x, const double* y, int64_t n, double* device_out);
int csr_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
// kernels_ell.hip
int ell_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int ell_analyse(spmv_mat* m);
int ell_build_panel(spmv_mat* m, bool only_if_worth);  // the row-grouped copy with the PANEL kernel forced on it
int ell_select_kernel(spmv_mat* m);                    // AUTO: the format's own variants and (where a candidate) the row-grouped copy, timed
int  csr_ell_copy_build(spmv_mat* m);  // SPMV_CSR_ELL: the ELL copy of a CSR handle with (nearly) equal rows
void csr_ell_copy_free(spmv_mat* m);
bool csr_ell_copy_worth(const spmv_mat* m);
// kernels_coo.hip
int  ell_build_tiles(spmv_mat* m, bool only_if_worth);
void ell_free_tiles(spmv_mat* m);
int  ell_build_dia_order(spmv_mat* m, bool only_if_worth);  // the DIA-order copy of the values (ell_variant 3)
void ell_free_dia_order(spmv_mat* m);
int coo_analyse(spmv_mat* m);
int coo_build_panel(spmv_mat* m, bool only_if_worth);  // the row-grouped copy with the PANEL kernel forced on it
int coo_select_kernel(spmv_mat* m);                    // AUTO: the segmented scan or the row-grouped copy (which picks its own kernel), timed
void coo_drop_rowgrouped(spmv_mat* m);
int  coo_build_bins(spmv_mat* m, int bins_per_xcd, bool only_if_worth);  // bins_per_xcd 0: as many as keep a slice of x inside an XCD's L2
void coo_free_bins(spmv_mat* m);
int coo_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int  csr_segscan_build(spmv_mat* m);  // SPMV_CSR_SEGSCAN: the row index per entry the scan runs over
void csr_segscan_free(spmv_mat* m);
int  csr_segscan_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
// kernels_csr_split.hip
int  csr_split_build(spmv_mat* m);
void csr_split_free(spmv_mat* m);
int  csr_split_threshold(const spmv_mat* m);
int  csr_split_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int  csr_split_long_rows_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
// kernels_misc.hip (CSC, DIA, BLAS-1, fill)
int csc_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int csc_analyse(spmv_mat* m);
int csc_select_kernel(spmv_mat* m);                       // AUTO: the scatter or the row-grouped copy (which picks its own kernel), timed
int csc_build_rowgrouped(spmv_mat* m, int32_t force_kernel);
void csc_drop_rowgrouped(spmv_mat* m);
int dia_apply(spmv_ctx* ctx, const spmv_mat* A, const double* x, double* y);
int dia_rows_apply(spmv_ctx* ctx, int nrow, int jmax, int k, const int32_t* offsets, const double* values, const double* x, double* y, bool off_known,
                   int off_min, int off_max, uint32_t flags, const unsigned long long* skip_rows, int stride);
int vec_fill(spmv_ctx* ctx, double* d, int64_t n, double a);
int vec_copy2(spmv_ctx* ctx, double* dst0, const double* src0, int64_t n0, double* dst1, const double* src1, int64_t n1);  // two copies, one launch
int vec_dot(spmv_ctx* ctx, const double* x, const double* y, int64_t n, double* result);
int vec_axpby(spmv_ctx* ctx, double alpha, const double* x, double beta, const double* y, double* w,
              int64_t n);
// validate.hip
int mat_validate(const spmv_mat* m);
// convert.hip
int exclusive_scan_i32(spmv_ctx* ctx, const int32_t* in, int32_t* out, int64_t n);
constexpr int32_t kCsrAutoNoSegscan = -1;  // coo_to_csr's force_kernel: AUTO among the kernels a COO handle does not have itself
int coo_to_csr(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out, int32_t force_kernel = 0 /* SPMV_CSR_AUTO: select */);
// symgs.hip
int  symgs_setup(spmv_mat* m);
void symgs_free(spmv_mat* m);
int  symgs_sweep(spmv_ctx* ctx, const spmv_mat* A, const double* b, double* x, bool zero_guess);
int  symgs_info(const spmv_mat* m, const char* what, int64_t* value);
int  symgs_sequence(const spmv_mat* m, int32_t* out);
// convert_sort.hip
int sort_ids_by_key(spmv_ctx* ctx, const int32_t* keys, int64_t n, int bits, int32_t* out_ids);
int coo_place_by_stable_sort(spmv_ctx* ctx, int64_t nnz, int32_t nrow, const int32_t* row, const int32_t* col, const double* val,
                             int32_t* out_col, double* out_val);
int csr_to_ell(spmv_ctx* ctx, const spmv_mat* csr, spmv_mat** out, bool pad_own_column = false);
int csr_split_columns(spmv_ctx* ctx, const spmv_mat* csr, int32_t c0, int32_t c1, spmv_mat** out_in, spmv_mat** out_out);
int csr_extract_rows(spmv_ctx* dst, const spmv_mat* csr, int64_t r0, int64_t r1, spmv_mat** out);
int coo_row_offsets(const spmv_mat* coo, int64_t* row_ptr64 /* host, nrow + 1 */);
int reduce_max_i32(spmv_ctx* ctx, const int32_t* in, int64_t n, int32_t* result);
// generate.hip
int gen_csr_uniform(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol, int32_t k,
                    int32_t band, uint64_t seed, spmv_mat** out);
int gen_ell_banded(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, uint64_t seed,
                   spmv_mat** out);
int gen_dia_banded(spmv_ctx* ctx, int32_t nrow, int32_t k, uint64_t seed, spmv_mat** out);
int gen_coo_powerlaw(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed, bool sorted_by_length,
                     spmv_mat** out);
int gen_vec_uniform(spmv_ctx* ctx, double* d, int64_t n, int64_t index_offset, uint64_t seed);

// plan.hip
inline const plan_node* plan_of(const spmv_mat* m) { return m->plan_base && m->plan_at >= 0 ? m->plan_base + m->plan_at : nullptr; }
enum plan_child { kPlanChildRowgrouped, kPlanChildLong, kPlanChildEll };
inline void plan_hand_down(const spmv_mat* parent, spmv_mat* child, plan_child which)
{
    const plan_node* p = plan_of(parent);
    if (!p) return;
    const int32_t at = which == kPlanChildRowgrouped ? p->child_rowgrouped : (which == kPlanChildLong ? p->child_long : p->child_ell);
    child->plan_base = at >= 0 ? parent->plan_base : nullptr;
    child->plan_at   = at;
}
bool plan_take_armed(spmv_mat* m);   // a handle under analysis takes the context's armed plan (if its format is the plan's); true: it did
void plan_clear(spmv_mat* m);
int  csr_apply_plan(spmv_mat* m);    // select.hip
void plan_reset_requests(spmv_mat* m);  // the parameters a plan sets explicitly, back to "choose" (a plan that did not fit)
int  ell_apply_plan(spmv_mat* m);    // kernels_ell.hip
int  coo_apply_plan(spmv_mat* m);    // kernels_coo.hip
int  csc_apply_plan(spmv_mat* m);    // kernels_misc.hip
// abi.hip helpers used by the other units
int  mat_alloc(spmv_ctx* ctx, int32_t format, int32_t nrow, int32_t ncol, int64_t nnz, int32_t k,
               size_t a_count, size_t b_count, size_t v_count, spmv_mat** out);
void mat_free(spmv_mat* m);
bool adds_into_y_with_atomics(const spmv_mat* A);  // global_atomic_add_f64 on y, by the kernel (of the handle or its copies) that runs
}  // namespace spmv
