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
constexpr int DPP_ROW_ROR8 = 0x128;        // lane i <-> i+-8 inside each 16-lane row

// value of lane perm(i) of every lane i (all source lanes are active group members)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
   const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
   const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
   return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
   return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}

// One-instruction min / max for the cross-lane reductions.  For the non-NaN values reduced here they
// return the same number as the compare-and-select forms dmin / dmax; the only representational
// difference is the sign of a zero result (v_min treats -0 < +0), which cannot propagate: the reduced
// quantities are only compared (+0 == -0) or enter sums a + h*(+-0) whose result does not depend on it.
__device__ __forceinline__ double vmin_f64(double a, double b)
{
   double r;
   // no trailing s_nop: the wait states a DPP read of the result needs are padded by hipcc's hazard recogniser, which
   // treats the asm statement as a VALU write of %0 (checked in the ISA of every reduction); only hazards between
   // instructions INSIDE one asm statement are left to the author (cdna_hip_programming.md 5.7 item 2)
   asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b)
{
   double r;
   asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
}

// All-reduce over the G lanes of a path group (G = 1, 2, 4, 8 or 16; groups are aligned to G lanes, so
// every DPP source lane belongs to the same group and shares its control flow).
template <int G>
__device__ __forceinline__ double grp_min(double v)
{
   if (G == 1) return v;
   v = vmin_f64(v, dpp_mov<DPP_QUAD_XOR1>(v));
   if (G >= 4) v = vmin_f64(v, dpp_mov<DPP_QUAD_XOR2>(v));
   if (G >= 8) v = vmin_f64(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
   if (G == 16) v = vmin_f64(v, dpp_mov<DPP_ROW_MIRROR>(v));
   return v;
}
template <int G>
__device__ __forceinline__ double grp_max(double v)
{
   if (G == 1) return v;
   v = vmax_f64(v, dpp_mov<DPP_QUAD_XOR1>(v));
   if (G >= 4) v = vmax_f64(v, dpp_mov<DPP_QUAD_XOR2>(v));
   if (G >= 8) v = vmax_f64(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
   if (G == 16) v = vmax_f64(v, dpp_mov<DPP_ROW_MIRROR>(v));
   return v;
}
// min over the group of h and max over the group of l in one interleaved sequence: the two chains are independent,
// so alternating them halves the number of hazard stalls: the DPP move that reads a v_min / v_max result needs two wait
// states after it; hipcc's hazard recogniser pads what is missing AFTER an asm statement (checked in the ISA: it
// inserts the s_nop itself), only hazards INSIDE one asm statement are the author's business.
__device__ __forceinline__ double vmin_f64_raw(double a, double b)
{
   double r;
   asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
}
__device__ __forceinline__ double vmax_f64_raw(double a, double b)
{
   double r;
   asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
}
template <int G>
__device__ __forceinline__ void grp_min_max(double &h, double &l)
{
   if (G == 1) return;
   double hd = dpp_mov<DPP_QUAD_XOR1>(h);
   double ld = dpp_mov<DPP_QUAD_XOR1>(l);
   h = vmin_f64_raw(h, hd);
   l = vmax_f64_raw(l, ld);
   if (G >= 4)
   {
      hd = dpp_mov<DPP_QUAD_XOR2>(h);
      ld = dpp_mov<DPP_QUAD_XOR2>(l);
      h = vmin_f64_raw(h, hd);
      l = vmax_f64_raw(l, ld);
   }
   if (G >= 8)
   {
      hd = dpp_mov<DPP_ROW_HALF_MIRROR>(h);
      ld = dpp_mov<DPP_ROW_HALF_MIRROR>(l);
      h = vmin_f64_raw(h, hd);
      l = vmax_f64_raw(l, ld);
   }
   if (G == 16)
   {
         hd = dpp_mov<DPP_ROW_MIRROR>(h);
      ld = dpp_mov<DPP_ROW_MIRROR>(l);
      h = vmin_f64_raw(h, hd);
      l = vmax_f64_raw(l, ld);
   }
}

// ---------------------------------------------------------------------------------------------------------------------
// Division by a value that stays the same for several quotients (theta' of the evaluation point divides the two bounds of
// every constraint check of a stage and the velocity limit of the next stage).  hipcc lowers an fp64 `a / b` to
//    ds = v_div_scale(b, b, a); ns = v_div_scale(a, b, a); r = v_rcp(ds); two Newton steps on r (four FMAs);
//    q = ns * r; rem = fma(-ds, q, ns); v_div_fmas(rem, r, q); v_div_fixup(., b, a)
// where both v_div_scale return their operand unchanged, v_div_fmas is a plain FMA and v_div_fixup returns its first
// operand unless an operand is zero / infinite / NaN, the quotient leaves the normal range or the exponents are extreme
// (ISA: scaling when the numerator's biased exponent is <= 53, the denominator or its reciprocal is denormal, the
// exponents differ by >= 768, or the quotient is denormal).  For |a|, |b| in [2^-350, 2^350] none of that applies, the
// refined reciprocal depends on b alone, and a quotient is the last three operations of the SAME sequence: the same bits
// as `a / b` by construction (checked on the device against `/` for 2^22 operand pairs incl. the edges of the window:
// batotp_hip_fp64_kat, tests/test_gpu_parity.py::test_shared_reciprocal_division...).  Outside the window the callers use `/`.
// Used by k_sweep8 (sweep8.hip.h) and k_sweep1 (sweep1.hip.h) for the quotients by theta' (ba.cpp:1223, 1526-1531) and by a1 (ba.cpp:1495-1509).
// ---------------------------------------------------------------------------------------------------------------------
constexpr double SDIV_LO = 0x1p-350, SDIV_HI = 0x1p350;
__device__ __forceinline__ bool sdiv_window(double x) { return (fabs(x) >= SDIV_LO) & (fabs(x) <= SDIV_HI); }
__device__ __forceinline__ double sdiv_rcp(double den)
{
   double r = __builtin_amdgcn_rcp(den);
   double e = __builtin_fma(-den, r, 1.0);
   r = __builtin_fma(r, e, r);
   e = __builtin_fma(-den, r, 1.0);
   r = __builtin_fma(r, e, r);
   return r;
}
__device__ __forceinline__ double sdiv_by(double num, double den, double r)
{
   const double q = num * r;
   const double rem = __builtin_fma(-den, q, num);
   return __builtin_fma(rem, r, q);
}


template <int G>
__device__ __forceinline__ int grp_or(int v)
{
   if (G == 1) return v;
   v |= dpp_mov<DPP_QUAD_XOR1>(v);
   if (G >= 4) v |= dpp_mov<DPP_QUAD_XOR2>(v);
   if (G >= 8) v |= dpp_mov<DPP_ROW_HALF_MIRROR>(v);
   if (G == 16) v |= dpp_mov<DPP_ROW_MIRROR>(v);
   return v;
}

} // namespace bk
