// plan.hip - a handle's set-up decisions as plain data: export, import, share.
//
// Round 5 made AUTO a measurement (select.hip): which kernel a handle runs, in which layout, with which chunk size, barrier
// placement, split threshold or ELL variant is found by timing candidates when the handle is created.  That has three costs the
// reference's shard construction (src/mat_vec.cpp:240-268: built once, the same way every time) does not have:
//   * two handles of one matrix - or two ranks holding statistically identical shards - may end on different kernels, and with
//     them on different last bits of y and different step times;
//   * the timing launches cost set-up time (C2: 0.195 s with, 0.048 s without; a C5 shard 0.4-1.0 s) on every handle, also
//     where the answer is known;
//   * a container re-uploaded after an edit of its values goes through all of it again.
// A PLAN is what those decisions come to: spmv_mat_get_plan writes it out as a small POD blob (a header and one 128-byte
// node per handle: the handle itself, then its copies - common.hpp: plan_node), spmv_mat_set_plan builds EXACTLY that kernel and
// layout on another handle of the same format with no timing launch, and spmv_ctx_set_plan makes every handle created on a
// context afterwards take the plan instead of selecting.  What a plan does NOT hold: anything about the matrix (row cuts,
// column panels and slot descriptors are re-derived from the handle's own arrays, so a plan made on one shard fits a shard of
// another size), and where the two-phase product stream lies in a device's physical memory (the piece search,
// "twophase_choose_pieces": a property of the device's allocator history, run separately where wanted).
#include <cstring>
#include <vector>

#include "common.hpp"

namespace spmv
{
namespace
{
int collect(const spmv_mat* m, std::vector<plan_node>& out)
{
    const int at = (int)out.size();
    out.emplace_back();
    plan_node n;
    memset(&n, 0, sizeof(n));
    n.format        = m->format;
    n.kernel        = m->kernel;
    n.lanes_per_row = m->lanes_per_row;
    n.flags         = m->flags;
    n.child_rowgrouped = n.child_long = n.child_ell = -1;
    // A plan is CANONICAL: a field is written only where the kernel that runs reads it (everything else stays 0), so that two
    // handles in the same effective state give the same bytes whatever layouts of other kernels they still hold or once held
    // (a handle forced to the row-parallel kernel keeps its panel layout in memory; a handle built from that plan has none).
    const bool csr = m->format == SPMV_FMT_CSR;
    if (csr && m->kernel == SPMV_CSR_PANEL)
    {
        // the panel layout: what it was built with, and what a launch reads (requests first, else what the trial found, else
        // defaults: the same precedence as panel_launch)
        n.pb_group_rows = m->pb_group_rows;
        n.pb_width      = m->pb_val ? m->pb_built_width : m->pb_panel_width;
        n.pb_sort       = m->pb_val ? m->pb_built_sort : m->pb_sort;
        n.pb_aos        = m->pb_val ? m->pb_built_layout : m->pb_aos;
        const int unroll = m->pb_unroll > 0 ? m->pb_unroll : (m->pb_unroll_tuned > 0 ? m->pb_unroll_tuned : 8);
        n.pb_unroll      = unroll >= 8 ? 8 : (unroll >= 4 ? 4 : 2);
        n.pb_pipe        = std::max(0, std::min(m->pb_pipe >= 0 ? m->pb_pipe : (m->pb_pipe_tuned > 0 ? m->pb_pipe_tuned : 1), 2));
        const int sync   = (m->pb_sync >= 0 ? m->pb_sync : m->pb_sync_tuned) & 3;
        n.pb_sync        = sync == 2 ? 3 : sync;
        n.pb_two_per_cu  = m->pb_two_per_cu;
        n.pb_rounds      = m->pb_val ? std::max(1, m->pb_built_rounds) : std::max(1, m->pb_rounds_req);
    }
    if (csr && m->kernel == SPMV_CSR_SPLIT)
    {
        n.split_threshold = m->split_built_threshold > 0 ? m->split_built_threshold : csr_split_threshold(m);
        n.split_mode      = m->split_built_mode ? m->split_built_mode : m->split_mode;
    }
    if (csr && m->kernel == SPMV_CSR_TWOPHASE)
    {
        n.tp_pcols  = m->tp_val ? m->tp_pcols : m->tp_pcols_req;
        n.tp_rotate = m->tp_rotate;
    }
    if (csr && m->kernel != SPMV_CSR_VECTOR && m->kernel != SPMV_CSR_LDSWIN && m->kernel != SPMV_CSR_AUTO) n.lanes_per_row = 0;  // (only the lane-group kernels read it)
    if (csr && m->kernel != SPMV_CSR_VECTOR && m->kernel != SPMV_CSR_AUTO) n.flags = 0;  // (the tuning bits of the row-parallel kernel)
    if (m->format == SPMV_FMT_ELL)
    {
        const bool own   = !(m->coo_csr && m->kernel == SPMV_CSR_PANEL);  // the format's own kernels run (else: the row-grouped copy)
        n.ell_variant    = own ? m->ell_variant : 0;
        n.ell_tiled      = own && m->ell_tval ? 1 : 0;
        if (!own) n.lanes_per_row = 0;
    }
    if (m->format == SPMV_FMT_COO) n.coo_bins_per_xcd = m->kernel == SPMV_CSR_PANEL ? 0 : m->cb_bins / 8;
    if (m->format == SPMV_FMT_COO || m->format == SPMV_FMT_CSC || m->format == SPMV_FMT_DIA) n.lanes_per_row = 0, n.flags = m->format == SPMV_FMT_DIA ? m->flags : 0;
    const bool from_copy = m->format != SPMV_FMT_CSR && m->coo_csr && m->kernel == SPMV_CSR_PANEL;
    const bool split     = m->format == SPMV_FMT_CSR && m->kernel == SPMV_CSR_SPLIT;
    if ((from_copy || split) && m->coo_csr) n.child_rowgrouped = collect(m->coo_csr, out);
    if (split && m->split_long) n.child_long = collect(m->split_long, out);
    if (m->format == SPMV_FMT_CSR && m->kernel == SPMV_CSR_ELL && m->ell_copy) n.child_ell = collect(m->ell_copy, out);
    out[(size_t)at] = n;
    return at;
}

// a blob is trusted with nothing: sizes, ids and child indices are checked before a node is read by anybody
int check_blob(const void* buf, int64_t len, std::vector<plan_node>* out)
{
    SPMV_REQUIRE(buf && len >= (int64_t)sizeof(plan_header), "plan: %lld bytes are not a plan", (long long)len);
    plan_header h;
    memcpy(&h, buf, sizeof(h));
    SPMV_REQUIRE(h.magic == kPlanMagic, "plan: bad magic 0x%08x", h.magic);
    SPMV_REQUIRE(h.version == kPlanVersion, "plan: version %u, this library reads version %u", h.version, kPlanVersion);
    SPMV_REQUIRE(h.nnodes >= 1 && h.nnodes <= 64 && h.bytes == sizeof(plan_header) + (size_t)h.nnodes * sizeof(plan_node) && (int64_t)h.bytes <= len,
                 "plan: %u nodes in %u bytes (%lld given)", h.nnodes, h.bytes, (long long)len);
    // (copied out before anything is read: the caller's buffer need not be aligned for 4-byte fields)
    std::vector<plan_node> n(h.nnodes);
    memcpy(n.data(), (const unsigned char*)buf + sizeof(plan_header), (size_t)h.nnodes * sizeof(plan_node));
    for (uint32_t i = 0; i < h.nnodes; ++i)
    {
        SPMV_REQUIRE(n[i].format >= SPMV_FMT_COO && n[i].format <= SPMV_FMT_DIA, "plan: node %u has format %d", i, n[i].format);
        SPMV_REQUIRE(n[i].kernel >= SPMV_CSR_AUTO && n[i].kernel <= SPMV_CSR_ELL, "plan: node %u has kernel %d", i, n[i].kernel);
        SPMV_REQUIRE(n[i].lanes_per_row >= 0 && n[i].lanes_per_row <= 64 && (n[i].lanes_per_row & (n[i].lanes_per_row - 1)) == 0, "plan: node %u has %d lanes per row", i,
                     n[i].lanes_per_row);
        for (const int32_t c : {n[i].child_rowgrouped, n[i].child_long, n[i].child_ell})
            // children come after their parent (collect writes them so): no cycles
            SPMV_REQUIRE(c == -1 || (c > (int32_t)i && c < (int32_t)h.nnodes), "plan: node %u names child %d of %u nodes", i, c, h.nnodes);
        if (n[i].child_rowgrouped >= 0) SPMV_REQUIRE(n[n[i].child_rowgrouped].format == SPMV_FMT_CSR, "plan: node %u: a row-grouped copy is a CSR handle", i);
        if (n[i].child_long >= 0) SPMV_REQUIRE(n[n[i].child_long].format == SPMV_FMT_CSR, "plan: node %u: the long rows' matrix is a CSR handle", i);
        if (n[i].child_ell >= 0) SPMV_REQUIRE(n[n[i].child_ell].format == SPMV_FMT_ELL, "plan: node %u: an ELL copy is an ELL handle", i);
        SPMV_REQUIRE(n[i].pb_rounds >= 0 && n[i].pb_rounds <= 16 && n[i].split_mode >= 0 && n[i].split_mode <= 2 && n[i].split_threshold >= 0 &&
                         n[i].coo_bins_per_xcd >= 0 && n[i].coo_bins_per_xcd <= 8 && n[i].ell_variant >= 0 && n[i].ell_variant <= 3 && n[i].tp_pcols >= 0,
                     "plan: node %u holds a parameter out of range", i);
    }
    *out = std::move(n);
    return SPMV_OK;
}
}  // namespace

void plan_clear(spmv_mat* m)
{
    m->plan_base = nullptr;
    m->plan_at   = -1;
}

bool plan_take_armed(spmv_mat* m)
{
    spmv_ctx* ctx = m->ctx;
    if (!ctx || !ctx->plan_armed) return false;
    ctx->plan_armed = false;  // the first handle analysed after a creating entry point began is the one that entry point creates
    if (plan_of(m) || ctx->plan_blob.empty()) return false;
    const plan_node* n = (const plan_node*)(ctx->plan_blob.data() + sizeof(plan_header));
    if (n[0].format != m->format) return false;  // a plan for another format: this handle selects as usual
    m->plan_base = n;
    m->plan_at   = 0;
    return true;
}
}  // namespace spmv

using namespace spmv;

extern "C" {

int spmv_mat_get_plan(const spmv_mat* m, void* buf, int64_t* len)
{
    SPMV_REQUIRE(m && len, "spmv_mat_get_plan: null argument");
    std::vector<plan_node> nodes;
    (void)collect(m, nodes);
    const int64_t need = (int64_t)sizeof(plan_header) + (int64_t)nodes.size() * (int64_t)sizeof(plan_node);
    if (!buf)
    {
        *len = need;
        return SPMV_OK;
    }
    SPMV_REQUIRE(*len >= need, "spmv_mat_get_plan: the plan takes %lld bytes, the buffer holds %lld", (long long)need, (long long)*len);
    plan_header h{kPlanMagic, kPlanVersion, (uint32_t)need, (uint32_t)nodes.size()};
    memcpy(buf, &h, sizeof(h));
    memcpy((unsigned char*)buf + sizeof(h), nodes.data(), nodes.size() * sizeof(plan_node));
    *len = need;
    return SPMV_OK;
}

int spmv_mat_set_plan(spmv_mat* m, const void* buf, int64_t len)
{
    SPMV_REQUIRE(m, "spmv_mat_set_plan: null matrix");
    std::vector<plan_node> own;  // (a private copy: the caller's buffer may go away while copies are being built)
    SPMV_TRY(check_blob(buf, len, &own));
    SPMV_REQUIRE(own[0].format == m->format, "spmv_mat_set_plan: the plan is for format %d, the handle holds format %d", own[0].format, m->format);
    SPMV_HIP(hipSetDevice(m->ctx->device));
    m->plan_base = own.data();
    m->plan_at   = 0;
    int rc       = SPMV_OK;
    switch (m->format)
    {
        case SPMV_FMT_CSR: rc = csr_apply_plan(m); break;
        case SPMV_FMT_COO: rc = coo_apply_plan(m); break;
        case SPMV_FMT_CSC: rc = csc_apply_plan(m); break;
        case SPMV_FMT_ELL: rc = ell_apply_plan(m); break;
        default: break;  // DIA: one kernel, nothing to plan
    }
    plan_clear(m);
    if (hipStreamSynchronize(m->ctx->stream) != hipSuccess && rc == SPMV_OK) SPMV_FAIL(SPMV_ERR_HIP, "spmv_mat_set_plan: %s", hipGetErrorString(hipGetLastError()));
    if (rc != SPMV_OK)
    {
        // a plan that does not fit this matrix (an LDS window too wide, an ELL copy of a matrix with an empty row, no memory for a
        // layout): the handle goes back to what AUTO makes of it, and the caller gets the error
        char why[512];
        snprintf(why, sizeof(why), "%s", spmv_last_error());
        (void)hipGetLastError();
        if (m->format == SPMV_FMT_CSR) plan_reset_requests(m);
        (void)spmv_mat_set_kernel(m, SPMV_CSR_AUTO, 0);
        set_error("spmv_mat_set_plan: %s (the handle selected its kernel by itself instead)", why);
    }
    return rc;
}

int spmv_plan_check(const void* buf, int64_t len, int32_t* format, int32_t* kernel, int32_t* nodes)
{
    std::vector<plan_node> n;
    SPMV_TRY(check_blob(buf, len, &n));
    if (format) *format = n[0].format;
    if (kernel) *kernel = n[0].kernel;
    if (nodes) *nodes = (int32_t)n.size();
    return SPMV_OK;
}

int spmv_ctx_set_plan(spmv_ctx* ctx, const void* buf, int64_t len)
{
    SPMV_REQUIRE(ctx, "spmv_ctx_set_plan: null context");
    if (!buf || len == 0)
    {
        ctx->plan_blob.clear();
        ctx->plan_armed = false;
        return SPMV_OK;
    }
    std::vector<plan_node> nodes;
    SPMV_TRY(check_blob(buf, len, &nodes));
    // stored behind its 16-byte header in the context's own (aligned) buffer: the nodes are read in place from there
    ctx->plan_blob.assign((const unsigned char*)buf, (const unsigned char*)buf + sizeof(plan_header) + nodes.size() * sizeof(plan_node));
    return SPMV_OK;
}

}  // extern "C"
