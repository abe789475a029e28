// wave.hpp — cross-lane primitives for 64-wide CDNA4 wavefronts (device code only).
//
// fp64 values travel as two 32-bit halves.  Three data paths are used:
//   ds_swizzle_b32  (bit-mode: lane' = ((lane & and) | or) ^ xor inside each 32-lane half) for the
//                   xor-butterfly steps 1..16 — no LDS memory is touched, only the LDS crossbar;
//   ds_bpermute_b32 (arbitrary gather across all 64 lanes) for the step that crosses the two
//                   32-lane halves and for segmented scans with a run-time distance;
//   DPP row_shr     (VALU, 16-lane rows) as an alternative for the <=16-lane steps.
#pragma once

#include <hip/hip_runtime.h>

namespace spmv
{
// clang ext-vector types: unlike HIP's int2/double2 structs they are accepted by
// __builtin_nontemporal_load and lower to one dwordx2 / dwordx4 instruction
using i32x2 = int __attribute__((ext_vector_type(2)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using f64x2 = double __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// ---- ds_swizzle xor butterfly inside a 32-lane half --------------------------------------------------
template <int XOR_MASK>
__device__ __forceinline__ double swizzle_xor(double v)
{
    static_assert(XOR_MASK >= 1 && XOR_MASK <= 31, "ds_swizzle bit-mode reaches 32 lanes");
    // offset[15]=0 selects bit-mode; and_mask=offset[4:0], or_mask=offset[9:5], xor_mask=offset[14:10]
    constexpr int pattern = (XOR_MASK << 10) | 0x1F;
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo     = __builtin_amdgcn_ds_swizzle(lo, pattern);
    hi     = __builtin_amdgcn_ds_swizzle(hi, pattern);
    return __hiloint2double(hi, lo);
}

// ---- ds_bpermute: read `v` from lane `src` (0..63) ------------------------------------------------------
__device__ __forceinline__ double bpermute(double v, int src_lane)
{
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo     = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
    hi     = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bpermute(int v, int src_lane)
{
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}

// ---- DPP row shift right by N inside 16-lane rows; lanes shifted in read 0 ------------------------------
template <int N>
__device__ __forceinline__ double dpp_row_shr(double v)
{
    constexpr int ctrl = 0x110 + N;  // DPP_ROW_SR0 + N
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo     = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi     = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// lane i reads lane i+N of its 16-lane row (row_shl); lanes past the row end read 0
template <int N>
__device__ __forceinline__ double dpp_row_shl(double v)
{
    constexpr int ctrl = 0x100 + N;  // DPP_ROW_SL0 + N
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo     = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi     = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ---- sum over aligned groups of LANES lanes; the result is valid (at least) in the group's lane 0 ----------
// SWIZZLE flavour: xor butterfly, every lane of the group ends with the full sum.
template <int LANES>
__device__ __forceinline__ double group_sum_swizzle(double v)
{
    if constexpr (LANES >= 2) v += swizzle_xor<1>(v);
    if constexpr (LANES >= 4) v += swizzle_xor<2>(v);
    if constexpr (LANES >= 8) v += swizzle_xor<4>(v);
    if constexpr (LANES >= 16) v += swizzle_xor<8>(v);
    if constexpr (LANES >= 32) v += swizzle_xor<16>(v);
    if constexpr (LANES >= 64) v += bpermute(v, lane_id() ^ 32);
    return v;
}
// DPP flavour: shift-down tree, only lane 0 of the group holds the full sum.
template <int LANES>
__device__ __forceinline__ double group_sum_dpp(double v)
{
    if constexpr (LANES >= 2) v += dpp_row_shl<1>(v);
    if constexpr (LANES >= 4) v += dpp_row_shl<2>(v);
    if constexpr (LANES >= 8) v += dpp_row_shl<4>(v);
    if constexpr (LANES >= 16) v += dpp_row_shl<8>(v);
    if constexpr (LANES >= 32) v += bpermute(v, lane_id() + 16);  // wraps mod 64; lane 0/32 read 16/48
    if constexpr (LANES >= 64) v += bpermute(v, lane_id() ^ 32);
    return v;
}
template <int LANES, bool USE_DPP>
__device__ __forceinline__ double group_sum(double v)
{
    if constexpr (USE_DPP)
        return group_sum_dpp<LANES>(v);
    else
        return group_sum_swizzle<LANES>(v);
}

// full-wave sum, valid in every lane
__device__ __forceinline__ double wave_sum(double v) { return group_sum_swizzle<64>(v); }

// ---- slotted accumulators (common.hpp: kDotSlots) -----------------------------------------------------------
__device__ __forceinline__ void slot_add(double* acc, double v)
{
    unsafeAtomicAdd(acc + (blockIdx.x & (kDotSlots - 1)) * kDotStride, v);
}
__device__ __forceinline__ double slot_sum(const double* acc)
{
    double t = 0.0;
#pragma unroll 8
    for (int i = 0; i < kDotSlots; ++i) t += acc[i * kDotStride];
    return t;
}

// ---- streaming (read-once) loads: nontemporal so the matrix stream does not displace x in L2/MALL -------
template <typename T>
__device__ __forceinline__ T load_stream(const T* p)
{
    return __builtin_nontemporal_load(p);
}
}  // namespace spmv
