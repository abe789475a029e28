// select.hip - AUTO kernel selection by MEASUREMENT ("adaptive format / kernel selection").
//
// Rounds 1-4 chose the kernel of a handle from a model (csr_choose_kernel: entry count, mean row length, column windows)
// whose thresholds were measured on BASELINE's own shapes - exactly 32 entries per row, uniform or band-random columns.
// Round 5 audited that policy on matrices that look like Matrix Market files (tools/sweep_structures.py: stencils,
// dense-block diagonals, R-MAT graphs, tall and wide rectangles, permutations; profiles/r05_sweep_structures_*.txt): 29 of 55
// (matrix, format) rows ran below 0.97 of their best forced kernel, some far below - a hub row of 8436 entries under the
// row-parallel kernel (0.18 ms against 0.016), an ELL handle of 5000 long rows (0.058 against 0.008), small COO handles
// (segmented scan 2x slower than the row-grouped copy), contiguous 64-entry rows (panel 0.116 against 0.093).  No set of
// thresholds covers that range.  What the handle does instead, when it is created (one-off, outside every timed region,
// like the reference's shard construction, src/mat_vec.cpp:240-268):
//
//   1. the model picks as before (improved where the audit showed a plain error: the row-grouped layouts from 1.5M entries
//      on, not 2M; no lower bound of 2 on the mean row length; skewed rows never go to the row-parallel kernel; long
//      contiguous rows do);
//   2. handles between 64K and 8M entries - where the candidates lie within a factor of a few of each other and a product
//      takes microseconds - TIME the candidates: 1 warm-up + 2 x 4 products each on zeroed scratch vectors (the gather
//      addresses, not the values, set the time), a candidate that is 3x behind after its first product is dropped at once;
//      the candidates go round twice and keep their minimum; the fastest wins, a later candidate has to win by 2 %.  Larger handles keep the model's pick (a trial of the
//      row-parallel kernel on C2 would cost 60 ms for a kernel that loses 5x) unless a statistic says the model may be
//      wrong: long contiguous rows (dense blocks) also time the row-parallel kernel;
//   3. layouts built for candidates that lost are freed before the call returns.
// Later in round 5 three more CSR kernels joined the candidates at ANY size where a statistic asks for them - the segmented
// scan and the long-row split for rows that dwarf the others (kernels_coo.hip: csr_segscan_build, kernels_csr_split.hip), the
// ELL copy for (nearly) equal rows with local columns (kernels_ell.hip: csr_ell_copy_build) - and the two-phase model is timed
// against the panel layout below 64M entries.
// "panel_trial" 0 / SPMV_PANEL_TRIAL=0 (no timing launches at all) leaves step 1 alone.  What was timed is reported:
// spmv_mat_get_param "select_candidates" and "select_us_<kernel>" (include/spmv_abi.h).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace spmv
{
bool select_trials_enabled(const spmv_mat* m)
{
    if (m->pb_trial == 0) return false;
    if (m->pb_trial > 0) return true;
    const char* e = getenv("SPMV_PANEL_TRIAL");  // (read when a handle is built, never on a product's path)
    return !(e && e[0] == '0');
}

// ---- the trial arena ----------------------------------------------------------------------------------------------
// Round 5 found products launched within a millisecond or two of a hipMalloc / hipFree running up to 2x slower, whichever
// kernel they are (the driver is still mapping / unmapping), and answered with a 2 ms sleep in front of the ELL trial.  Round 6
// removes the allocations it can and stops trusting a single moment for the rest:
//   * the zeroed x and y that every trial multiplies with come from ONE allocation of the context (ctx->arena: x in its first
//     half, y in its second - nothing ever writes the first half, and y += A * 0 leaves the second what it was), made before the
//     first trial of a context and kept: no hipMalloc / hipFree per trial, nested trials (a handle's copy selecting its own
//     kernel) share it.  Up to 4M entries per vector (64 MB in all) - the sizes where a product takes microseconds and a
//     transient decides; larger vectors get allocations of their own as before (their products take tens of microseconds up);
//   * the candidates of a small handle are BUILT first and timed afterwards (csr_select_kernel), and every trial goes round
//     until no candidate's minimum moves any more (select_rounds): whatever the allocations of a build stirred up has died
//     down by the time the minima agree, without anybody sleeping.
constexpr size_t kArenaHalfMax = (size_t)4 << 20;  // doubles per half

int select_scratch::alloc(spmv_ctx* c, int64_t ncol, int64_t nrow)
{
    const size_t nx = (size_t)std::max<int64_t>(ncol, 1), ny = (size_t)std::max<int64_t>(nrow, 1);
    const size_t half = std::max(nx, ny);
    if (half <= kArenaHalfMax)
    {
        if (c->arena_bytes < 2 * half * sizeof(double) && c->arena_users == 0)
        {
            size_t want = (size_t)1 << 17;  // doubles per half: 1 MB at least, powers of two
            while (want < half) want <<= 1;
            (void)hipStreamSynchronize(c->stream);
            if (c->arena) (void)hipFree(c->arena);
            c->arena       = nullptr;
            c->arena_bytes = 0;
            if (hipMalloc(&c->arena, 2 * want * sizeof(double)) == hipSuccess && hipMemsetAsync(c->arena, 0, 2 * want * sizeof(double), c->stream) == hipSuccess)
                c->arena_bytes = 2 * want * sizeof(double);
            else
            {
                (void)hipGetLastError();
                if (c->arena) (void)hipFree(c->arena);
                c->arena = nullptr;
            }
        }
        if (c->arena_bytes >= 2 * half * sizeof(double))
        {
            x          = (double*)c->arena;
            y          = (double*)c->arena + c->arena_bytes / sizeof(double) / 2;
            ctx        = c;
            from_arena = true;
            ++c->arena_users;
            return SPMV_OK;
        }
    }
    // beyond the arena (or the arena is in use and too small: a nested trial of another shape): allocations of its own
    if (hipMalloc(&x, sizeof(double) * nx) != hipSuccess || hipMalloc(&y, sizeof(double) * ny) != hipSuccess)
    {
        (void)hipGetLastError();
        release();
        return SPMV_ERR_ALLOC;
    }
    (void)hipMemsetAsync(x, 0, sizeof(double) * nx, c->stream);
    (void)hipMemsetAsync(y, 0, sizeof(double) * ny, c->stream);
    return SPMV_OK;
}
void select_scratch::release()
{
    if (from_arena)
    {
        if (ctx && ctx->arena_users > 0) --ctx->arena_users;
        from_arena = false;
        ctx        = nullptr;
    }
    else
    {
        if (x) (void)hipFree(x);
        if (y) (void)hipFree(y);
    }
    x = y = nullptr;
}

int select_time(spmv_ctx* ctx, const std::function<int()>& launch, float best_so_far, float* out_ms)
{
    auto run = [&](int n, float* ms) -> int {
        if (hipEventRecord(ctx->ev_begin, ctx->stream) != hipSuccess) return SPMV_ERR_HIP;
        for (int i = 0; i < n; ++i)
        {
            const int rc = launch();
            if (rc != SPMV_OK) return rc;
        }
        if (hipEventRecord(ctx->ev_end, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev_end) != hipSuccess ||
            hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end) != hipSuccess)
        {
            set_error("kernel selection: a timing launch failed: %s", hipGetErrorString(hipGetLastError()));
            return SPMV_ERR_HIP;
        }
        *ms /= (float)n;
        return SPMV_OK;
    };
    int rc = launch();  // warm-up (first-use set-up of the kernel, LDS grants)
    if (rc != SPMV_OK) return rc;
    float first = 0.f;
    if ((rc = run(1, &first)) != SPMV_OK) return rc;
    *out_ms = first;
    // hopeless: no more launches for it.  Only above 50 us: a single launch after an idle moment can take 25 us by itself, which
    // is "3x behind" any candidate at launch-latency scale (seen: a 3.6 us kernel recorded at 26 us in both passes)
    if (first > 3.0f * best_so_far && first > 0.05f) return SPMV_OK;
    // two runs of 4 products - more where a product takes microseconds, so that a run lasts ~0.1 ms and the events' own
    // resolution and the launch jitter stay below the 2 % a candidate has to win by
    const int n = std::min(32, std::max(4, (int)(0.06f / std::max(first, 1e-4f))));  // (round 6: runs of 0.06 ms, not 0.1 - there are more rounds now)
    float     a = 0.f, b = 0.f;
    if ((rc = run(n, &a)) != SPMV_OK || (rc = run(n, &b)) != SPMV_OK) return rc;
    *out_ms = std::min(a, b);
    return SPMV_OK;
}

// The candidates 0 .. n-1 timed in rounds, the minimum per candidate, until a round moves no minimum by 3 % or more (at least
// two rounds, at most six; two where a product takes 50 us and more - a transient of a millisecond is noise there and a round
// costs milliseconds).  t[i] < 0 on entry: not timed yet; t[i] >= 0: a timing from before, kept as a minimum to improve on.
// A candidate 8x behind the fastest is left out of later rounds.  This replaces "time twice and hope" and round 5's sleep: what a
// build's allocations stirred up a moment ago shows as a minimum that still moves, and the rounds go on until it does not.
int select_rounds(spmv_ctx* ctx, int n, const std::function<int(int)>& launch, float* t, int* rounds_run)
{
    constexpr int kMaxRounds = 6;
    float fastest = 1e30f;
    for (int i = 0; i < n; ++i)
        if (t[i] >= 0.f) fastest = std::min(fastest, t[i]);
    // ... and a budget: a trial must not cost more than the handle can be expected to win back.  Behind the reference's call
    // shape a small matrix is multiplied 50 times (main.cpp:16) at ~20 us a call - a millisecond in all - and its set-up falls
    // inside the reference's timed loop wherever a container reaches the device with its first product.  Beyond the two rounds
    // every trial gets, rounds go on only while the whole trial has taken less than 3 ms.
    const auto t_begin = std::chrono::steady_clock::now();
    int r = 0;
    for (; r < kMaxRounds; ++r)
    {
        if (r >= 2 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() > 3.0) break;
        bool moved = false;
        for (int i = 0; i < n; ++i)
        {
            if (t[i] >= 0.f && t[i] > 8.0f * fastest) continue;
            float ms = 0.f;
            SPMV_TRY(select_time(ctx, [&] { return launch(i); }, fastest, &ms));
            if (t[i] < 0.f || ms < 0.97f * t[i]) moved = true;
            t[i]    = t[i] < 0.f ? ms : std::min(t[i], ms);
            fastest = std::min(fastest, t[i]);
        }
        if (r >= 1 && (!moved || fastest > 0.05f))
        {
            ++r;
            break;
        }
    }
    if (rounds_run) *rounds_run = r;
    return SPMV_OK;
}

void select_note(spmv_mat* m, int slot, float ms)
{
    if (slot >= 0 && slot < 10) m->sel_us[slot] = ms * 1000.f;
    ++m->sel_candidates;
}
void select_reset(spmv_mat* m)
{
    m->sel_candidates = 0;
    m->sel_rounds     = 0;
    for (float& v : m->sel_us) v = 0.f;
}

// ---- a plan on a CSR handle (plan.hip) ------------------------------------------------------------------------------
// The kernel and every parameter a trial would have chosen come from the node; nothing is timed (pb_trial = 0 for the duration:
// no pace trial, no rounds trial, no piece search of the two-phase layout - "twophase_choose_pieces" runs that afterwards, it is
// about where the pieces lie in THIS device's memory and no part of a plan).  Layouts of other kernels are released.
int csr_apply_plan(spmv_mat* m)
{
    const plan_node& p = *plan_of(m);
    SPMV_REQUIRE(p.kernel >= SPMV_CSR_VECTOR && p.kernel <= SPMV_CSR_ELL, "plan: CSR kernel id %d", p.kernel);
    SPMV_REQUIRE((m->b && m->v) || m->nnz == 0, "plan: this handle gave up its CSR arrays (panel_keep_csr = 0)");
    select_reset(m);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (p.kernel != SPMV_CSR_PANEL) csr_panel_free(m);
    if (p.kernel != SPMV_CSR_TWOPHASE) csr_twophase_free(m);
    if (p.kernel != SPMV_CSR_SEGSCAN) csr_segscan_free(m);
    csr_split_free(m);     // (rebuilt from the plan: its parts carry decisions of their own)
    csr_ell_copy_free(m);  // (idem)
    const int32_t trial_before = m->pb_trial;
    m->pb_trial       = 0;
    m->kernel         = p.kernel;
    m->kernel_forced  = false;
    m->split_auto_low = false;
    if (p.lanes_per_row > 0) m->lanes_per_row = p.lanes_per_row;
    m->flags = p.flags;
    int rc   = SPMV_OK;
    if (m->nrow > 0 && m->nnz > 0) switch (p.kernel)
        {
            case SPMV_CSR_PANEL:
                m->pb_group_rows  = p.pb_group_rows;
                m->pb_panel_width = p.pb_width;
                m->pb_sort        = p.pb_sort;
                m->pb_aos         = p.pb_aos;
                m->pb_unroll      = p.pb_unroll;
                m->pb_pipe        = p.pb_pipe;
                m->pb_sync        = p.pb_sync;
                m->pb_two_per_cu  = p.pb_two_per_cu;
                m->pb_rounds_req  = std::max(1, p.pb_rounds);
                rc                = csr_panel_build(m);
                break;
            case SPMV_CSR_TWOPHASE:
                m->tp_pcols_req = p.tp_pcols;
                m->tp_rotate    = p.tp_rotate;
                rc              = csr_twophase_build(m);
                break;
            case SPMV_CSR_SEGSCAN: rc = csr_segscan_build(m); break;
            case SPMV_CSR_SPLIT:
                m->split_threshold = p.split_threshold;
                m->split_mode      = p.split_mode;
                rc                 = csr_split_build(m);  // (hands the parts' nodes down)
                break;
            case SPMV_CSR_ELL: rc = csr_ell_copy_build(m); break;
            case SPMV_CSR_LDSWIN:
                if (m->win_max_span <= 0 || m->win_max_span > csr_ldswin_capacity())
                {
                    set_error("plan: the LDS-window kernel does not fit this matrix (widest block window %d columns, the tile holds %d)", m->win_max_span,
                              csr_ldswin_capacity());
                    rc = SPMV_ERR_UNSUPPORTED;
                }
                break;
            default: break;  // VECTOR, SCALAR: the CSR arrays as they are
        }
    m->pb_trial = trial_before;
    return rc;
}

void plan_reset_requests(spmv_mat* m)
{
    m->pb_group_rows = m->pb_panel_width = m->pb_unroll = m->pb_rounds_req = 0;
    m->pb_sort       = 1;
    m->pb_aos        = 4;
    m->pb_pipe = m->pb_sync = -1;
    m->pb_two_per_cu        = 1;
    m->split_threshold = m->split_mode = m->tp_pcols_req = 0;
    m->tp_rotate                                       = 256;
    m->kernel_forced                                   = false;
}

// ---- CSR ----------------------------------------------------------------------------------------------------------
// Leaves m->kernel chosen and its layout built; layouts of candidates that lost are freed.
int csr_select_kernel(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    select_reset(m);
    csr_choose_kernel(m);  // the model (no launches)
    const int model = m->kernel;
    constexpr int kSplitLow = 100;  // a candidate of this function only (not a kernel id): kernel SPLIT with every row of 256 entries and more split off
    m->split_auto_low = false;
    auto build = [&](int kernel) -> int {
        if (kernel == kSplitLow || kernel == SPMV_CSR_SPLIT)
        {
            m->split_auto_low = kernel == kSplitLow;
            m->kernel         = SPMV_CSR_SPLIT;
            return csr_split_build(m);
        }
        m->kernel = kernel;
        if (kernel == SPMV_CSR_PANEL) return csr_panel_build(m);
        if (kernel == SPMV_CSR_TWOPHASE) return csr_twophase_build(m);
        if (kernel == SPMV_CSR_SEGSCAN) return csr_segscan_build(m);
        if (kernel == SPMV_CSR_ELL) return csr_ell_copy_build(m);
        return SPMV_OK;
    };
    if (m->nrow == 0 || m->nnz == 0 || !m->b || !m->v) return build(model);
    const double mean = (double)m->nnz / (double)m->nrow;
    std::vector<int> cand{model};
    auto             add = [&](int k) {
        if (std::find(cand.begin(), cand.end(), k) == cand.end()) cand.push_back(k);
    };
    // a row with 1/128 of the entries and more: under every row-wise kernel that row is one workgroup's work at best
    // (the panel kernel: 1.25 us per 1024 entries of it) - the scan over the entries in equal pieces is a candidate
    const bool long_row = !m->sel_no_segscan && m->max_row_nnz >= 4096 && (int64_t)m->max_row_nnz * 128 >= m->nnz;
    // a few rows far longer than the others (hubs of a graph, dense constraint rows): those rows in chunks, the rest through
    // a copy with a kernel of its own (kernels_csr_split.hip).  R-MAT scale 20, longest row 69348 of 16.8M entries (1/242):
    // panel 0.094 ms, split 0.076; an arrow of 60000 rows: scan 0.0092, split 0.0056
    const bool hub_rows = !m->sel_no_split && m->max_row_nnz >= 4096 && (double)m->max_row_nnz >= 32.0 * std::max(mean, 1.0) &&
                          (m->nnz < kSelectMaxNnz || (int64_t)m->max_row_nnz * 512 >= m->nnz);
    if (select_trials_enabled(m) && m->nnz >= kSelectMinNnz)
    {
        if (long_row)
        {
            add(SPMV_CSR_SEGSCAN);
            add(SPMV_CSR_PANEL);  // (also beyond 8M entries, where the model is otherwise taken at its word)
        }
        if (hub_rows) add(SPMV_CSR_SPLIT);
        // (nearly) equal rows - stencils, bands, block diagonals: an ELL copy streams values and indices coalesced and reads no
        // index at all where its slots are diagonals (kernels_ell.hip: csr_ell_copy_build; 1.15x to 1.57x over the best CSR
        // kernel on those shapes, at any size)
        if (csr_ell_copy_worth(m)) add(SPMV_CSR_ELL);
        // large graphs whose long rows also share their hub COLUMNS (R-MAT): neighbouring lanes of the panel kernel then add
        // into the same LDS accumulator, and it pays to turn every row of 256 entries and more into virtual rows (scale 22, 67M
        // entries: 0.327 -> 0.250 ms) - where two launches instead of one are not what decides (scale 20, 16.8M: 0.074 ->
        // 0.085).  No statistic at hand tells the two apart: both thresholds are timed.
        if (hub_rows && m->nnz >= kSelectMaxNnz && m->split_threshold == 0 && csr_split_threshold(m) > 256) add(kSplitLow);
        if (m->nnz < kSelectMaxNnz)
        {
            add(SPMV_CSR_PANEL);
            add(SPMV_CSR_VECTOR);
            if (m->win_max_span > 0 && m->win_max_span <= csr_ldswin_capacity()) add(SPMV_CSR_LDSWIN);
            if (mean <= 8.0 && m->max_row_nnz <= 64) add(SPMV_CSR_SCALAR);  // one lane per row: short, even rows only
            // an x beyond L2 under a handle of 1M entries and more: the gather-free two phases may win where the model's
            // run-length estimate says no (a permutation of 8M rows: 0.093 ms against the panel kernel's 0.137)
            if ((double)m->ncol * 8.0 > 4.0 * 1048576.0 && m->nnz >= ((int64_t)1 << 20)) add(SPMV_CSR_TWOPHASE);
        }
        else if (m->contig_frac >= 0.5 && mean >= 16.0)
        {
            add(SPMV_CSR_VECTOR);  // long contiguous rows (dense blocks): the row-parallel kernel reads x coalesced
            add(SPMV_CSR_PANEL);
            if (m->win_max_span > 0 && m->win_max_span <= csr_ldswin_capacity()) add(SPMV_CSR_LDSWIN);  // (64 x 64 blocks: 0.092 against 0.094 / 0.113)
        }
        // the two-phase model was measured on 10M rows x 32 under an x of 80 .. 640 MB (C5's shards); on other shapes below that
        // size it has been wrong by 4x (12.7M entries in 200000 rows of 64 over 4M columns: 0.80 ms, panel 0.06): time the panel
        // layout beside it.  From 64M entries on its word stands (a shard of C5 keeps the set-up - and the allocation history
        // its piece search starts from - it was measured with).
        if (m->nnz >= kSelectMaxNnz && model == SPMV_CSR_TWOPHASE && m->nnz < ((int64_t)64 << 20)) add(SPMV_CSR_PANEL);
    }
    if (cand.size() == 1) return build(model);

    select_scratch sv;
    if (sv.alloc(ctx, m->ncol, m->nrow) != SPMV_OK) return build(model);  // no room to try: the model's pick
    int                rc = SPMV_OK;
    float              fastest = 1e30f;
    std::vector<float> t(cand.size(), -1.f);
    // (a transient - the driver still unmapping what the caller freed a moment ago, clocks ramping - hits whoever is being timed
    // at that moment; a disturbed first measurement was seen 6x off, tools/probe_ell_trial.py: hence rounds, and minima)
    if (m->nnz < kSelectMaxNnz)
    {
        // every candidate's layout is built first (at most 8M entries: a few layouts of 100 MB side by side), then they are timed
        // in rounds with no allocation in between, until their minima stand still (select_rounds)
        std::vector<int> built;
        for (size_t i = 0; i < cand.size(); ++i)
        {
            if ((rc = build(cand[i])) != SPMV_OK)
            {
                if (cand[i] == model) return rc;
                (void)hipGetLastError();
                rc = SPMV_OK;  // a candidate that cannot be built (no memory for another layout) is not a candidate
                continue;
            }
            built.push_back((int)i);
        }
        std::vector<float> tb(built.size(), -1.f);
        rc = select_rounds(ctx, (int)built.size(),
                           [&](int j) {
                               m->kernel = cand[(size_t)built[(size_t)j]];  // (no split at two thresholds below 8M entries: the kernel id says which layout runs)
                               return csr_apply(ctx, m, sv.x, sv.y);
                           },
                           tb.data(), &m->sel_rounds);
        for (size_t j = 0; j < built.size(); ++j) t[(size_t)built[j]] = tb[j];
    }
    else
        // (from 8M entries on: one pass, build and time in turn - a product takes tens of microseconds there, the layouts are
        // hundreds of megabytes each, and the split in its two variants is built anew for every timing)
        for (size_t i = 0; i < cand.size() && rc == SPMV_OK; ++i)
        {
            const int k = cand[i];
            if ((rc = build(k)) != SPMV_OK)
            {
                if (k == model) return rc;
                (void)hipGetLastError();
                rc = SPMV_OK;  // a candidate that cannot be built (no memory for a second layout) is not a candidate
                continue;
            }
            float ms = 0.f;
            rc       = select_time(ctx, [&] { return csr_apply(ctx, m, sv.x, sv.y); }, fastest, &ms);
            if (rc != SPMV_OK) break;
            t[i]    = ms;
            fastest = std::min(fastest, t[i]);
        }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != SPMV_OK) return rc;
    int   best    = -1;
    float best_ms = 1e30f;
    for (size_t i = 0; i < cand.size(); ++i)  // the model's pick first; a later candidate has to win by 2 %
        if (t[i] >= 0.f)
        {
            select_note(m, cand[i] == kSplitLow ? 0 : cand[i], t[i]);
            if (t[i] < best_ms * (best >= 0 ? 0.98f : 1.0f))
            {
                best    = cand[i];
                best_ms = t[i];
            }
        }
    if (best < 0) best = model;
    if (best != SPMV_CSR_PANEL) csr_panel_free(m);
    if (best != SPMV_CSR_TWOPHASE) csr_twophase_free(m);
    if (best != SPMV_CSR_SEGSCAN) csr_segscan_free(m);
    if (best != SPMV_CSR_SPLIT && best != kSplitLow) csr_split_free(m);
    if (best != SPMV_CSR_ELL) csr_ell_copy_free(m);
    return build(best);  // (a layout that is already in memory with the current parameters is kept as it is)
}

}  // namespace spmv
