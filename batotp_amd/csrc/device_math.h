// device_math.h -- lane-group primitives and exact fp64 helpers for the gfx950 kernels.
//
// Arithmetic contract of every kernel in this directory: IEEE-754 binary64, no FMA contraction
// (the translation unit is compiled with -ffp-contract=off), operations in the order the
// reference performs them.  fp64 '/' and sqrt lower to the correctly rounded sequences on gfx950
// (checked on the device by batotp_hip_fp64_kat / tests/test_gpu_kat.py).
#pragma once
#include <hip/hip_runtime.h>

namespace bk
{

// std::min / std::max of libstdc++: min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ int sgn(double v) { return (0.0 < v) - (v < 0.0); }

constexpr double kInf = __builtin_huge_val();

// DPP controls (CDNA ISA): quad_perm = sel0 | sel1<<2 | sel2<<4 | sel3<<6
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8-lane half row
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i inside each 16-lane row

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
   hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
   return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
   return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// All-reduce over the G lanes of a path group (G = 1, 8 or 16; groups are aligned to G lanes, so
// every DPP source lane belongs to the same group and shares its control flow).
template <int G>
__device__ __forceinline__ double grp_min(double v)
{
   if (G == 1) return v;
   v = dmin(v, dpp_mov<DPP_QUAD_XOR1>(v));
   v = dmin(v, dpp_mov<DPP_QUAD_XOR2>(v));
   v = dmin(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
   if (G == 16) v = dmin(v, dpp_mov<DPP_ROW_MIRROR>(v));
   return v;
}
template <int G>
__device__ __forceinline__ double grp_max(double v)
{
   if (G == 1) return v;
   v = dmax(v, dpp_mov<DPP_QUAD_XOR1>(v));
   v = dmax(v, dpp_mov<DPP_QUAD_XOR2>(v));
   v = dmax(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
   if (G == 16) v = dmax(v, dpp_mov<DPP_ROW_MIRROR>(v));
   return v;
}
template <int G>
__device__ __forceinline__ int grp_or(int v)
{
   if (G == 1) return v;
   v |= dpp_mov<DPP_QUAD_XOR1>(v);
   v |= dpp_mov<DPP_QUAD_XOR2>(v);
   v |= dpp_mov<DPP_ROW_HALF_MIRROR>(v);
   if (G == 16) v |= dpp_mov<DPP_ROW_MIRROR>(v);
   return v;
}

} // namespace bk
