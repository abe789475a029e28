// Properties of arm-spmv_amd/csrc/panel_groups.hpp (the panel layout's row groups, csr_panel_build) on row-length profiles the
// GPU tests are too small or too slow for.  Compiled and run by tests/test_abi_and_host.py; prints one summary line.
//   * every cut covers all rows once, in order; no group exceeds the row cap; no group but a single over-long row exceeds the bound;
//   * the balanced cut stays within the requested number of groups (or the number the row cap forces);
//   * the busiest-CU figure is what a direct count gives;
//   * a profile of many light rows and a heavy stretch (the shape of an R-MAT graph's row lengths) leaves the single round far
//     from even and a finer cut close to it - and a uniform profile asks for no trial at all.
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
#include "panel_groups.hpp"

using spmv::panel_balanced_cut;
using spmv::panel_cut;
using spmv::panel_rounds_worth_a_trial;

static std::vector<int32_t> offsets(const std::vector<int32_t>& len)
{
    std::vector<int32_t> rp(len.size() + 1, 0);
    for (size_t i = 0; i < len.size(); ++i) rp[i + 1] = rp[i] + len[i];
    return rp;
}

static int check_cut(const std::vector<int32_t>& rp, int nrow, const std::vector<int32_t>& gs, int64_t T, int cap)
{
    int bad = 0;
    if (gs.empty() || gs.front() != 0 || gs.back() != nrow) ++bad;
    for (size_t g = 0; g + 1 < gs.size(); ++g)
    {
        const int     rows = gs[g + 1] - gs[g];
        const int64_t ent  = (int64_t)rp[(size_t)gs[g + 1]] - rp[(size_t)gs[g]];
        if (rows < 1 || rows > cap) ++bad;
        if (ent > T && rows != 1) ++bad;  // (a single row longer than the bound is a group of its own)
    }
    return bad;
}

int main()
{
    int          bad = 0, cases = 0;
    std::mt19937 rng(12345);
    const int    cap = 20000, cus = 256;
    // random profiles: uniform, power law, blocks of heavy rows, empty stretches
    for (int trial = 0; trial < 60; ++trial)
    {
        const int            nrow = 1 + (int)(rng() % 400000);
        std::vector<int32_t> len((size_t)nrow);
        const int            kind = trial % 4;
        for (int i = 0; i < nrow; ++i)
        {
            const double u = (rng() + 1.0) / 4294967297.0;
            len[(size_t)i] = kind == 0 ? 16 : kind == 1 ? (int32_t)std::min(50000.0, 8.0 / u) : kind == 2 ? ((i / 5000) % 7 == 0 ? 200 : 2) : (i % 3 ? 0 : (int32_t)(rng() % 40));
        }
        const std::vector<int32_t> rp = offsets(len);
        for (int want : {1, 7, 256, 512, 1000})
        {
            std::vector<int32_t> gs;
            const double         busiest = panel_balanced_cut(rp, nrow, want, cap, cus, &gs);
            ++cases;
            const int groups = (int)gs.size() - 1;
            // the number of groups the row cap forces whatever the bound
            const int least = (nrow + cap - 1) / cap;
            if (groups > std::max(want, least) && rp[(size_t)nrow] > 0) ++bad;
            // the bound in effect is at most the fullest multi-row group's entries
            int64_t T = 1;
            for (size_t g = 0; g + 1 < gs.size(); ++g)
                if (gs[g + 1] - gs[g] > 1) T = std::max<int64_t>(T, (int64_t)rp[(size_t)gs[g + 1]] - rp[(size_t)gs[g]]);
            bad += check_cut(rp, nrow, gs, T, cap);
            // direct count of the busiest CU
            std::vector<int64_t> load((size_t)cus, 0);
            for (size_t g = 0; g + 1 < gs.size(); ++g) load[g % cus] += (int64_t)rp[(size_t)gs[g + 1]] - rp[(size_t)gs[g]];
            int64_t mx = 0;
            for (int64_t v : load) mx = std::max(mx, v);
            const double direct = rp[(size_t)nrow] > 0 ? (double)mx * cus / (double)rp[(size_t)nrow] : 1.0;
            if (std::abs(direct - busiest) > 1e-12 * std::max(1.0, direct)) ++bad;
            if (busiest < 1.0 - 1e-12) ++bad;
        }
        // fixed group sizes: exact
        std::vector<int32_t> gs;
        const int            G = 1 + (int)(rng() % cap);
        panel_cut(rp, nrow, INT32_MAX, G, &gs);
        for (size_t g = 0; g + 2 < gs.size(); ++g)
            if (gs[g + 1] - gs[g] != G) ++bad;
        bad += check_cut(rp, nrow, gs, INT32_MAX, G);
    }
    // the R-MAT shape: 4M rows, the first 1/16 of them carry 3/4 of the entries
    double one = 0, two = 0;
    int    pick = -1, pick_uniform = -1;
    {
        const int            nrow = 4 << 20;
        std::vector<int32_t> len((size_t)nrow);
        for (int i = 0; i < nrow; ++i) len[(size_t)i] = i < nrow / 16 ? 180 + (int)(rng() % 40) : 3 + (int)(rng() % 3);
        const std::vector<int32_t> rp = offsets(len);
        std::vector<int32_t>       gs, gs2;
        one  = panel_balanced_cut(rp, nrow, 256, cap, cus, &gs);
        two  = panel_balanced_cut(rp, nrow, 512, cap, cus, &gs2);
        pick = panel_rounds_worth_a_trial(rp, nrow, 256, cap, cus, one);
        if (!(one > 1.5) || !(two < 0.9 * one) || pick < 2) ++bad;
        std::vector<int32_t> flat((size_t)nrow, 16);
        const std::vector<int32_t> rpf = offsets(flat);
        const double          even = panel_balanced_cut(rpf, nrow, 256, cap, cus, &gs);
        pick_uniform               = panel_rounds_worth_a_trial(rpf, nrow, 256, cap, cus, even);
        if (even > 1.01 || pick_uniform != 0) ++bad;
    }
    std::printf("panel_groups: %d cuts checked, %d violations; skewed 4M rows: busiest CU %.3f (one round) / %.3f (two), trial of %d rounds; uniform: trial of %d\n",
                cases, bad, one, two, pick, pick_uniform);
    return bad ? 1 : 0;
}
