// Pure host arithmetic of the panel layout's row groups (kernels_csr_panel.hip: csr_panel_build): no HIP, no state, so that
// tests/test_abi_and_host.py can compile it with g++ (tests/panel_groups_check.cpp).
//
// A row group is a stretch of consecutive rows whose y accumulators fit one CU's LDS (at most `cap` rows); a 1024-lane
// workgroup walks all entries of its group, so a product lasts as long as the busiest workgroup.  Workgroup b of the launch
// takes the groups b, b + cus, b + 2 cus, ... (csr_panel_pp_kernel's loop).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace spmv
{
// Greedy cut: every group takes rows while its entries stay <= T and its rows <= cap (at least one row).  Returns the number
// of groups; `out` (optional) receives the first row of every group and nrow at the end.
inline int panel_cut(const std::vector<int32_t>& rp, int nrow, int64_t T, int cap, std::vector<int32_t>* out)
{
    int groups = 0, r = 0;
    if (out) out->assign(1, 0);
    while (r < nrow)
    {
        const int     r_cap = std::min(nrow, r + cap);
        const int64_t limit = (int64_t)rp[(size_t)r] + T;
        // last row index e in (r, r_cap] with rp[e] <= limit; at least one row
        int e = (int)(std::upper_bound(rp.begin() + r + 1, rp.begin() + r_cap + 1, limit, [](int64_t v, int32_t x) { return v < (int64_t)x; }) - rp.begin()) - 1;
        if (e <= r) e = r + 1;
        r = e;
        ++groups;
        if (out) out->push_back(r);
    }
    return groups;
}

// The cut into at most `groups` groups with the smallest entry bound T.  Returns what the busiest of `cus` CUs gets over the
// mean (1.0 = perfectly even); `out` receives the group boundaries.
inline double panel_balanced_cut(const std::vector<int32_t>& rp, int nrow, int groups, int cap, int cus, std::vector<int32_t>* out)
{
    const int64_t nnz = (int64_t)rp[(size_t)nrow] - (int64_t)rp[0];
    int64_t       lo = std::max<int64_t>(1, (nnz + groups - 1) / groups), hi = std::max<int64_t>(nnz, 1);
    while (lo < hi)
    {
        const int64_t mid = lo + (hi - lo) / 2;
        if (panel_cut(rp, nrow, mid, cap, nullptr) <= groups)
            hi = mid;
        else
            lo = mid + 1;
    }
    panel_cut(rp, nrow, lo, cap, out);
    if (nnz <= 0) return 1.0;
    std::vector<int64_t> load((size_t)cus, 0);
    for (size_t g = 0; g + 1 < out->size(); ++g) load[g % (size_t)cus] += (int64_t)rp[(size_t)(*out)[g + 1]] - (int64_t)rp[(size_t)(*out)[g]];
    return (double)*std::max_element(load.begin(), load.end()) * (double)cus / (double)nnz;
}

// Which multiple of `want` groups is worth a timing against `want` itself: 0 = none.  A finer cut has to bring the busiest CU
// down by 10 % per step to be considered (every further round of workgroups is another sweep of x).
inline int panel_rounds_worth_a_trial(const std::vector<int32_t>& rp, int nrow, int want, int cap, int cus, double busiest_one_round)
{
    if (!(busiest_one_round > 1.15)) return 0;
    int    pick   = 0;
    double q_best = 0.9 * busiest_one_round;
    for (int rounds = 2; rounds <= 4; ++rounds)
    {
        std::vector<int32_t> alt;
        const double         q = panel_balanced_cut(rp, nrow, want * rounds, cap, cus, &alt);
        if (q < q_best)
        {
            pick   = rounds;
            q_best = 0.9 * q;
        }
    }
    return pick;
}
}  // namespace spmv
