// Walks arm-spmv_amd/csrc/placement_math.hpp over EVERY positive finite fp32 time (and a band of pairs around each):
// the reported spread of the two-phase piece search is exactly 1000 when the pieces stay as built, and never below 1000
// when the kept configuration is the faster one by the last measurement.  Compiled and run by tests/test_abi_and_host.py.
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include "placement_math.hpp"

int main()
{
    long long checked = 0, bad_same = 0, bad_pair = 0, old_form_999 = 0;
    for (uint32_t bits = 0x00800000u; bits < 0x7f800000u; ++bits)  // every positive normal float
    {
        float t;
        std::memcpy(&t, &bits, 4);
        if (spmv::tp_spread_permille(t, t, true) != 1000) ++bad_same;
        // round 4's arithmetic, for the record: how often did it say 999 for equal times?
        if (t > 0.3f && t < 2.0f && (int32_t)(1000.0f * t / t) < 1000) ++old_form_999;
        // kept faster than built by one ulp up to a factor of two
        if ((bits & 0xFFu) == 0)
        {
            for (uint32_t d : {1u, 2u, 3u, 1000u, 1u << 20, 1u << 23})
            {
                if (bits + d >= 0x7f800000u) continue;
                uint32_t bb = bits + d;
                float    tb;
                std::memcpy(&tb, &bb, 4);
                if (spmv::tp_spread_permille(tb, t, false) < 1000) ++bad_pair;
                if (spmv::tp_spread_permille(t, t, false) != 1000) ++bad_pair;  // equal times, different pieces
            }
        }
        ++checked;
    }
    if (spmv::tp_spread_permille(0.f, 1.f, false) != 0 || spmv::tp_spread_permille(1.f, 0.f, false) != 0) ++bad_pair;
    if (spmv::tp_spread_permille(1e30f, 1e-30f, false) != INT32_MAX) ++bad_pair;
    std::printf("checked %lld same %lld pair %lld old999 %lld\n", checked, bad_same, bad_pair, old_form_999);
    return bad_same || bad_pair ? 1 : 0;
}
