// Pure host arithmetic of the two-phase piece search (kernels_csr_twophase.hip: tp_choose_pieces): no HIP, no state, so that
// tests/test_abi_and_host.py can compile it with g++ and walk it over every fp32 time (tests/placement_math_check.cpp).
#pragma once
#include <cmath>
#include <cstdint>

namespace spmv
{
// "twophase_placement_spread": time of the configuration as built / time of the configuration kept, in 1/1000.
//   same_configuration: the search kept (or fell back to) the pieces the layout was built with -> exactly 1000, whatever the
//   two timings say (they are then the same measurement; round 4 computed (int32_t)(1000.0f * t / t) here, which is 999 for
//   one fp32 t in nine between 0.3 and 2 ms).
//   otherwise the caller has made sure kept < built by its last measurement; the quotient is formed in double and rounded to
//   nearest, so it is >= 1000 whenever t_kept <= t_built.
inline int32_t tp_spread_permille(float t_built_ms, float t_kept_ms, bool same_configuration)
{
    if (same_configuration) return 1000;
    if (!(t_kept_ms > 0.f) || !(t_built_ms > 0.f)) return 0;
    const double q = 1000.0 * (double)t_built_ms / (double)t_kept_ms;
    return q >= 2.0e9 ? INT32_MAX : (int32_t)std::lround(q);
}
}  // namespace spmv
