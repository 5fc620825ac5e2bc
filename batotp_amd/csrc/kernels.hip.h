// kernels.hip.h -- the gfx950 kernels of the batotp hot path (included by batotp_hip.hip).
//
//   k_sites / k_spline / k_samples / k_dynamics   per-knot precompute (K1, K2)
//   k_spline_pairs / k_spline_sol / k_spline_series  the same Thomas solve leaving second derivatives (compact
//                                                  splines of the hot path; resampler; output stage)
//   k_pointwise, k_pointwise_grp                   per-knot max admissible sdot (K3)
//   k_sweep<G, FEAT, UNI>                          reverse / forward sweep, G lanes per path (K4)
// (resample.hip.h and output.hip.h hold the kernels of the stages before and after the hot path)
//
// Reference routines restated here (file:line under /root/reference/batotp):
//   Spline::getSplineCoeffs spline.cpp:168-211, solveTriDiagNatural spline.cpp:252-276,
//   findInterpSegs spline.cpp:56-99, interp1spline spline.cpp:129-155,
//   BA::evalSplineFullTraj ba.cpp:790-863, BA::findDynModel ba.cpp:873-949,
//   Robot::dynRR robot.cpp:377-431, dynCSPR3DOF robot.cpp:487-517, setA robot.cpp:534-558,
//   solveLinSys util.cpp:413-442 (Eigen PartialPivLU), solveQuadratic util.cpp:361-383,
//   BA::sweep ba.cpp:979-1195, sdotLim ba.cpp:1204-1236, applyAccelConstraintsBisectionPt
//   ba.cpp:1248-1332, evalSplinePartials ba.cpp:1341-1413, evalCartQuadCoeffs ba.cpp:1423-1439,
//   verifySecondOrderConstraints ba.cpp:1449-1581, evalsdot ba.cpp:1590-1607,
//   updateCurSeg ba.cpp:1617-1652.
//
// Data layout in HBM (all fp64), per path p with N knots starting at knot offset koff:
//   yin   [Cin][N]          uploaded knot values, channel-major
//   sC    [N]               knot sites
//   coef  [N][C][4]         spline coefficients c0..c3, knot-major / channel-interleaved: the lanes
//                           of a path group read 32 contiguous bytes each, C*32 contiguous bytes per
//                           group and segment.  Device channel order: theta[nJ], cart[nC], then per
//                           dynamics row r the four channels (a1_r, a2_r, a3_r, a4_r).
//   samp  [Cin][3][N]       value, d/ds, d2/ds2 at the output sites
//   dyn   [4][d][N]         a1..a4 at the knots
//   km    [N][Cin][2]       compact splines (BATOTP_F_COMPACT_SPLINES, FEAT == -1): (knot value, second derivative)
//                           pairs, knot-major; replaces yin, coef and the Thomas scratch
//   curve [cap] double2     (s, sdot) of a sweep; the reverse sweep fills its curve from the end so
//                           that it is ascending in s, as the reference leaves it after std::reverse
#pragma once
#include <stdint.h>
#include "batotp_hip.h"
#include "device_math.h"

// diagnostic build (-DBK_PROFILE_SECTIONS): cycle attribution inside k_sweep; no effect otherwise
#ifdef BK_PROFILE_SECTIONS
#define BK_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define BK_ACC(acc, a, b) acc += (b) - (a)
#else
#define BK_TICK(var)
#define BK_ACC(acc, a, b)
#endif

namespace bk
{

struct DevProblem
{
   int nJ, nC, d, robot;
   unsigned flags;
   int C;      // channels per knot in coef
   int Cin;    // nJ + nC
   int pad;
   double vmax[8], amax[8], tmax[8], tmin[8];
   double cart_vel_max, cart_acc_max, jnt_thresh, quad_thresh, integ_res, max_integ_time;
   double pmat[9];
};

struct PathInfo
{
   int64_t koff;  // first knot of this path in the per-knot arrays
   int64_t n;     // knots
   double sres_c; // traj.sresC
   double sres;   // traj.sres after evalSplineFullTraj
   double vfact, afact;
   int32_t parallel_now; // BA::_isParallelMech as seen by the sweep
   int32_t uniform;      // sC[i] == sres_c * i exactly (sites can be computed instead of loaded)
   double integ_res;     // BA::_integRes of THIS path (batotp_problem.integ_res unless batotp_hip_set_path_integ_res changed it:
                         // the automatic integration resolution of ba.cpp:493-556 derives it from the path)
};

struct alignas(32) Coef4
{
   double c0, c1, c2, c3;
};

__constant__ double c_ctab[64]; // Thomas super-diagonal c[i] of the (1,4,1) system; constant from c_conv on

// ---------------------------------------------------------------------------------------------
// K0: knot sites sC[i] = sres * i  (ba.cpp:800-806)
// ---------------------------------------------------------------------------------------------
__global__ void k_sites(const PathInfo *__restrict__ pinfo, int B, double *__restrict__ sC, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   // locate the path by binary search over koff
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const int64_t i = g - pinfo[lo].koff;
   sC[g] = pinfo[lo].sres_c * (double)i;
}

// ---------------------------------------------------------------------------------------------
// K1: natural-spline coefficients of one channel per thread (Thomas recurrence is sequential in
// the knot index; all channels of all paths run in parallel).
// src: channel-major values; mode 0: yin channels (dev channel = c), mode 1: dyn channels
// (c = k*d + r  ->  dev channel Cin + r*4 + k).
// ---------------------------------------------------------------------------------------------
// Correctly rounded num/den for a loop-invariant divisor whose correctly rounded reciprocal rcp is
// known (Markstein: q0 = RN(num*rcp); r = num - den*q0 exactly (one FMA); q = RN(q0 + r*rcp) is
// RN(num/den) when rcp = RN(1/den), the mantissa of den is not all ones and nothing under/overflows).
// This takes the divide's Newton iteration off the dependent chain of the Thomas recurrence
// (4 dependent operations instead of 11).  The FMAs are explicit: they are the algorithm, not a
// contraction.  Outside the safe magnitude window, and for zero (sign), the true division is used.
__device__ __forceinline__ double div_by_const(double num, double den, double rcp)
{
   const double an = fabs(num);
   if (an > 1e-280 && an < 1e280)
   {
      const double q0 = num * rcp;
      const double r = __builtin_fma(-den, q0, num);
      return __builtin_fma(r, rcp, q0);
   }
   return num / den;
}

// (x / 6.0) correctly rounded through the reciprocal (div_by_const: RN(1/6) is a compile-time constant, the significand of 6 is
// not all ones): the bits of the IEEE quotient in 4 dependent operations instead of the ~30 instructions hipcc emits for `/`
// (checked on the device for 1.4e6 operands incl. neighbours of exact multiples: test_division_by_six_...).  Every coefficient row
// formed from (value, second derivative) pairs divides by 6 twice -- since round 5 all of them go through this (k_dynamics and the
// per-knot kernel of the cable robot form 12-36 rows per knot; the resampler one per channel and output site).
__device__ __forceinline__ double div6(double x) { return div_by_const(x, 6.0, 1.0 / 6.0); }

__device__ __forceinline__ void emit_segment(double *__restrict__ cf, int64_t i, int C, int dc, double solL, double solR,
                                             double yL, double yR)
{
   Coef4 o;
   o.c3 = div6(solR - solL);
   o.c2 = solL / 2.0;
   o.c1 = yR - yL - div6(solR + 2 * solL);
   o.c0 = yL;
   *reinterpret_cast<Coef4 *>(cf + (i * C + dc) * 4) = o;
}

// The solve kernels PARK the eliminated right-hand sides in global memory and load them back in the back substitution, tens of thousands
// of steps later in the same launch.  BK_PARK_RELOAD_AGENT = 1 makes that reload an agent-scope load (served by L2, never by a CU's vector
// L1): bit-identical by construction, prepared as the first mitigation to measure against the transient wrong-knots event of rounds 4 and 6
// (profiles/r06_i_*, last section).  Default 0: the shipped code is the code every measurement and parity run of round 6 used.
#ifndef BK_PARK_RELOAD_AGENT
#define BK_PARK_RELOAD_AGENT 0
#endif
#if BK_PARK_RELOAD_AGENT
#define BK_PARK_RELOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define BK_PARK_RELOAD(ptr) (*(ptr))
#endif

// The Thomas solve of one series of N values (spline.cpp:252-276) and what follows it: SOL = false emits the
// coefficient rows (emit_segment), SOL = true leaves the second derivatives sol[0..N-1] in dpark.
// STRIDED = false: y and dpark are contiguous; true: element i of y is y[i*ys], of dpark dpark[i*ds].
template <bool SOL, bool STRIDED>
__device__ __forceinline__ void thomas_series(int N, const double *__restrict__ y, int ysIn, double *__restrict__ dpark, int dsIn,
                                              double *__restrict__ cf, int C, int dc)
{
   const int ys = STRIDED ? ysIn : 1, ds = STRIDED ? dsIn : 1;
   const int n = N - 1;
   constexpr int CONV = 63;                    // c_ctab is constant from here on (checked by the host)
   const double cInf = c_ctab[CONV];
   const double denInf = 4.0 - 1.0 * cInf;
   const double rcpInf = c_ctab[0];            // RN(1/denInf), computed (and checked) by the host
   constexpr int CH = 32;                      // knots per batch of independent loads

   // forward elimination (spline.cpp:259-269).  The bulk runs in branch-free batches of CH knots
   // whose loads are issued together (one memory round trip per CH dependent divide steps) and
   // whose pivot is the converged constant; the first rows (table pivots) and the remainder take
   // the simple loop.
   double dprev = (6 * (y[(0) * ys] - 2 * y[(1) * ys] + y[(2) * ys])) / 4.0;
   dpark[(1) * ds] = dprev;
   double ym = y[(1) * ys], y0 = y[(2) * ys];
   int i = 2;
   auto step_fwd = [&](int ii, double yp) {
      const double rhs = 6 * (ym - 2 * y0 + yp);
      const double den = (ii - 1) < CONV ? (4.0 - 1.0 * c_ctab[ii - 1]) : denInf;
      const double di = (rhs - 1.0 * dprev) / den;
      dpark[(ii) * ds] = di;
      dprev = di;
      ym = y0; y0 = yp;
   };
   for (; i < n && i <= CONV + 1; ++i) step_fwd(i, y[(i + 1) * ys]);
   for (; i + CH <= n; i += CH)
   {
      double yy[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) yy[k] = y[(i + 1 + k) * ys];
#pragma unroll
      for (int k = 0; k < CH; ++k)
      {
         const double rhs = 6 * (ym - 2 * y0 + yy[k]);
         const double di = div_by_const(rhs - 1.0 * dprev, denInf, rcpInf);
         dpark[(i + k) * ds] = di;
         dprev = di;
         ym = y0; y0 = yy[k];
      }
   }
   for (; i < n; ++i) step_fwd(i, y[(i + 1) * ys]);
   const double cl = (n - 1) < CONV ? c_ctab[n - 1] : cInf;
   double solR = (0.0 - 1.0 * dprev) / (4.0 - 1.0 * cl); // spline.cpp:269 (not forced to zero)

   if (SOL) dpark[(n) * ds] = solR;
   else
   {
      // row of the last knot stays zero (spline.cpp:203-209 never writes it)
      Coef4 z; z.c0 = 0; z.c1 = 0; z.c2 = 0; z.c3 = 0;
      *reinterpret_cast<Coef4 *>(cf + ((unsigned)n * (unsigned)C * 4 + dc * 4)) = z;
   }

   // back substitution fused with the coefficient formulas (spline.cpp:271-274, 203-209):
   // reference loop index ii runs n .. 2 with d[ii-1] -= c[ii-1]*d[ii]
   double yR = y[(n) * ys];
   i = n;
   for (; i - CH >= CONV + 1; i -= CH)
   {
      double dd[CH], yy[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k)
      {
         dd[k] = BK_PARK_RELOAD(&dpark[(i - 1 - k) * ds]);
         yy[k] = y[(i - 1 - k) * ys];
      }
#pragma unroll
      for (int k = 0; k < CH; ++k)
      {
         const double solL = dd[k] - cInf * solR; // ii - 1 = i - k - 1 >= CONV
         if (SOL) dpark[(i - k - 1) * ds] = solL;
         else emit_segment(cf, i - k - 1, C, dc, solL, solR, yy[k], yR);
         solR = solL;
         yR = yy[k];
      }
   }
   for (; i > 1; --i)
   {
      const double ci = (i - 1) < CONV ? c_ctab[i - 1] : cInf;
      const double yL = y[(i - 1) * ys];
      const double solL = BK_PARK_RELOAD(&dpark[(i - 1) * ds]) - ci * solR;
      if (SOL) dpark[(i - 1) * ds] = solL;
      else emit_segment(cf, i - 1, C, dc, solL, solR, yL, yR);
      solR = solL;
      yR = yL;
   }
   if (SOL) dpark[(0) * ds] = 0.0;
   else emit_segment(cf, 0, C, dc, 0.0, solR, y[(0) * ys], yR);
}

// SOL = true (the resampler's spline builds): instead of the coefficient rows, the second
// derivatives sol[0..N-1] of the channel are left in `scratch` (same layout as src); consumers
// form c0..c3 of a segment from sol and y with emit_segment's formulas (coeffs_from_sol).
template <bool SOL, bool PAIRS>
__device__ __forceinline__ void spline_channel(const PathInfo *__restrict__ pinfo, int B, int nch, int mode, int C, int Cin, int d,
                                               const double *__restrict__ src, int64_t src_stride_per_knot,
                                               double *__restrict__ scratch, double *__restrict__ coef, const int *__restrict__ only = nullptr)
{
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= B * nch) return;
   if (only && !only[t]) return;   // series the tiled kernel (spline_tile.hip.h) has already solved
   const int p = t / nch, c = t - p * nch;
   const PathInfo pi = pinfo[p];
   const int N = (int)pi.n;
   // PAIRS (compact splines): src = scratch = the knot-major array of (value, second derivative) pairs,
   // [N][C][2]; the value of knot i of channel c is element (i*C + c)*2, its scratch slot the next one
   // (PAIRS: Cin carries the first channel of the nch series inside the C channels of a knot)
   const int64_t streamOff = PAIRS ? pi.koff * C * 2 + (int64_t)(Cin + c) * 2 : pi.koff * src_stride_per_knot + (int64_t)c * N;
   const double *__restrict__ y = src + streamOff;
   double *__restrict__ dpark = scratch + streamOff + (PAIRS ? 1 : 0); // eliminated right-hand sides d[i]
   double *__restrict__ cf = coef + pi.koff * C * 4;
   const int dc = (mode == 0) ? c : (Cin + (c % d) * 4 + (c / d));
   thomas_series<SOL, PAIRS>(N, y, 2 * C, dpark, 2 * C, cf, C, dc);
}

__global__ void __launch_bounds__(64) k_spline(const PathInfo *__restrict__ pinfo, int B, int nch, int mode, int C, int Cin, int d,
                                               const double *__restrict__ src, int64_t src_stride_per_knot,
                                               double *__restrict__ scratch, double *__restrict__ coef, const int *__restrict__ only)
{
   spline_channel<false, false>(pinfo, B, nch, mode, C, Cin, d, src, src_stride_per_knot, scratch, coef, only);
}

__global__ void __launch_bounds__(64) k_spline_sol(const PathInfo *__restrict__ pinfo, int B, int nch, int C, const double *__restrict__ src,
                                                   double *__restrict__ sol, const int *__restrict__ only)
{
   spline_channel<true, false>(pinfo, B, nch, 0, C, C, 1, src, (int64_t)C, sol, nullptr, only);
}

// compact splines of the hot path: km = [N][C][2] (value, second derivative) pairs per path, solved in place
// (series c0 .. c0 + nch - 1 of the C channels a knot holds: the input channels of a velocity / acceleration-only batch, or -- all
// channels as pairs -- the input channels and, in a second launch, the dynamics channels behind them)
__global__ void __launch_bounds__(64) k_spline_pairs(const PathInfo *__restrict__ pinfo, int B, int nch, int c0, int C, double *__restrict__ km,
                                                     const int *__restrict__ only)
{
   spline_channel<true, true>(pinfo, B, nch, 0, C, c0, 1, km, (int64_t)C, km, nullptr, only);
}

// natural-spline second derivatives of arbitrary series: series k has n[k] values y[yOff[k] + i*ys] and leaves its
// second derivatives at sol[solOff[k] + i]; one lane per series (output stage: s(t) of a path, channels to re-sample -- the
// series the wavefront-per-series kernel of spline_lanes.hip.h leaves: short ones)
__global__ void __launch_bounds__(64) k_spline_series(int count, const int64_t *__restrict__ yOff, const int64_t *__restrict__ solOff,
                                                      const int *__restrict__ n, const double *__restrict__ y, int ys,
                                                      double *__restrict__ sol, const int *__restrict__ only)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= count || n[k] < 4) return;
   if (only && !only[k]) return; // series that k_spline_series_lanes (spline_lanes.hip.h) has solved
   thomas_series<true, true>(n[k], y + yOff[k], ys, sol + solOff[k], 1, nullptr, 1, 0);
}

// coefficient row of segment i from the second derivatives and the values at its two ends
// (emit_segment's formulas, spline.cpp:203-209)
__device__ __forceinline__ Coef4 coeffs_from_sol(double solL, double solR, double yL, double yR)
{
   Coef4 o;
   o.c3 = div6(solR - solL);
   o.c2 = solL / 2.0;
   o.c1 = yR - yL - div6(solR + 2 * solL);
   o.c0 = yL;
   return o;
}

// ---------------------------------------------------------------------------------------------
// K2a: value / first / second derivative of every input channel at the output sites
// sMVC[i] = sScale*i (ba.cpp:809-813, spline.cpp:56-99,129-155).  One thread per knot.
// ---------------------------------------------------------------------------------------------
__global__ void k_samples(const PathInfo *__restrict__ pinfo, int B, int C, int Cin, const double *__restrict__ sC,
                          const double *__restrict__ coef, double *__restrict__ samp, batotp_path_result *__restrict__ res,
                          int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int64_t N = pi.n, i = g - pi.koff;
   const double *__restrict__ s = sC + pi.koff;

   // findInterpSegs aborts when two knots coincide (spline.cpp:81-89)
   if (i < N - 1 && (s[i + 1] - s[i]) < 1e-20) atomicOr(&res[lo].status_rev, (unsigned)BATOTP_ST_SEG_ERROR);

   const double sScale = s[N - 1] / (double)(N - 1);
   const double site = sScale * (double)i;
   // segment = first k with site < s[k+1], clipped to N-2 (cursor search of spline.cpp:70-79)
   int64_t seg = (i < N - 1) ? i : N - 2;
   while (seg > 0 && site < s[seg]) --seg;
   while (seg < N - 2 && !(site < s[seg + 1])) ++seg;
   const double tau = (site - s[seg]) / (s[seg + 1] - s[seg]);
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   const double vfact = 1.0 / pi.sres_c;
   const double afact = vfact * vfact;

   const double *__restrict__ cf = coef + (pi.koff + seg) * C * 4;
   double *__restrict__ o = samp + pi.koff * Cin * 3;
   for (int c = 0; c < Cin; ++c)
   {
      const Coef4 k = *reinterpret_cast<const Coef4 *>(cf + c * 4);
      o[((int64_t)c * 3 + 0) * N + i] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
      o[((int64_t)c * 3 + 1) * N + i] = (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * vfact;
      o[((int64_t)c * 3 + 2) * N + i] = (6 * k.c3 * tau + 2 * k.c2) * afact;
   }
}

// ---------------------------------------------------------------------------------------------
// 3x3 dense solve as Eigen::PartialPivLU does it for util.cpp:432-438 (unblocked LU, first-max
// pivot, true division of the sub-column, column-oriented substitutions).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap_d(double &a, double &b) { const double t = a; a = b; b = t; }

__device__ void lu3_solve(const double *A /* row-major 3x3 */, const double *b, double *x)
{
   double m00 = A[0], m01 = A[1], m02 = A[2], m10 = A[3], m11 = A[4], m12 = A[5], m20 = A[6], m21 = A[7], m22 = A[8];
   double r0 = b[0], r1 = b[1], r2 = b[2];
   // k = 0
   {
      int piv = 0;
      double best = fabs(m00);
      if (fabs(m10) > best) { best = fabs(m10); piv = 1; }
      if (fabs(m20) > best) { best = fabs(m20); piv = 2; }
      if (best != 0.0)
      {
         if (piv == 1) { swap_d(m00, m10); swap_d(m01, m11); swap_d(m02, m12); }
         if (piv == 2) { swap_d(m00, m20); swap_d(m01, m21); swap_d(m02, m22); }
         m10 /= m00; m20 /= m00;
      }
      if (piv == 1) swap_d(r0, r1);
      if (piv == 2) swap_d(r0, r2);
      m11 -= m10 * m01; m12 -= m10 * m02;
      m21 -= m20 * m01; m22 -= m20 * m02;
   }
   // k = 1
   {
      int piv = 1;
      double best = fabs(m11);
      if (fabs(m21) > best) { best = fabs(m21); piv = 2; }
      if (best != 0.0)
      {
         if (piv == 2) { swap_d(m10, m20); swap_d(m11, m21); swap_d(m12, m22); }
         m21 /= m11;
      }
      if (piv == 2) swap_d(r1, r2);
      m22 -= m21 * m12;
   }
   // unit-lower forward substitution, column oriented
   if (r0 != 0.0) { r1 -= r0 * m10; r2 -= r0 * m20; }
   if (r1 != 0.0) { r2 -= r1 * m21; }
   // upper back substitution, column oriented
   if (r2 != 0.0) { r2 /= m22; r0 -= r2 * m02; r1 -= r2 * m12; }
   if (r1 != 0.0) { r1 /= m11; r0 -= r1 * m01; }
   if (r0 != 0.0) { r0 /= m00; }
   x[0] = r0; x[1] = r1; x[2] = r2;
}

// ---------------------------------------------------------------------------------------------
// The isSVD = 1 branch of solveLinSys (util.cpp:421-438) for the 3x3 wrench systems: Eigen's two-sided Jacobi SVD
// (JacobiSVD<MatrixXd>(A, ComputeThinU | ComputeThinV), Eigen 3.3) and its solve, written out: work matrix = A scaled by its
// largest entry; sweeps over the pairs q < p; an off-diagonal pair above max(DBL_MIN, 2 eps max|diag|) is removed by a rotation
// that makes the 2x2 block symmetric followed by the symmetric Jacobi rotation (JacobiRotation::makeJacobi), both accumulated
// into U / V; singular values = |diag| x scale, descending; x = V diag(1/sigma) U^T b over the numerical rank (three-term sums
// left to right).  Pinned by the reference binary's isSVD = 1 outputs (tests/golden/CSPR3DOF_svd, CSPR3DOF_par_svd).
// ---------------------------------------------------------------------------------------------
struct Rot2 { double c, s; };
__device__ __forceinline__ void rot_pair(double &x, double &y, const Rot2 g)
{
   const double a = x, b = y;
   x = g.c * a + g.s * b;       // apply_rotation_in_the_plane
   y = -g.s * a + g.c * b;
}
__device__ inline void svd3_solve(const double *A /* row-major 3x3 */, const double *b, double *x)
{
   const double tiny = 2.2250738585072014e-308, eps = 2.220446049250313e-16;
   double W[3][3], U[3][3], V[3][3];
   double scale = 0.0;
#pragma unroll
   for (int k = 0; k < 9; ++k) scale = fmax(scale, fabs(A[k]));
   if (scale == 0.0) scale = 1.0;
#pragma unroll
   for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
         W[i][j] = A[i * 3 + j] / scale;
         U[i][j] = (i == j) ? 1.0 : 0.0;
         V[i][j] = (i == j) ? 1.0 : 0.0;
      }
   double diagMax = fmax(fmax(fabs(W[0][0]), fabs(W[1][1])), fabs(W[2][2]));
   for (int sweep = 0; sweep < 64; ++sweep)      // (Eigen loops until a sweep finds nothing; a 3x3 matrix needs a handful)
   {
      bool clean = true;
#pragma unroll
      for (int pq = 0; pq < 3; ++pq)
      {
         const int p = pq == 0 ? 1 : 2, q = pq == 2 ? 1 : 0;    // (1,0), (2,0), (2,1)
         const double limit = fmax(tiny, 2.0 * eps * diagMax);
         if (!(fabs(W[p][q]) > limit || fabs(W[q][p]) > limit)) continue;
         clean = false;
         double b00 = W[p][p], b01 = W[p][q], b10 = W[q][p], b11 = W[q][q];
         Rot2 sym = {1.0, 0.0};
         const double trace = b00 + b11, skew = b10 - b01;
         if (!(fabs(skew) < tiny))
         {
            const double u = trace / skew;
            const double h = sqrt(1.0 + u * u);
            sym.s = 1.0 / h;
            sym.c = u / h;
         }
         if (!(sym.c == 1.0 && sym.s == 0.0)) { rot_pair(b00, b10, sym); rot_pair(b01, b11, sym); }
         Rot2 right = {1.0, 0.0};
         {
            const double twice = 2.0 * fabs(b01);
            if (!(twice < tiny))
            {
               const double tau = (b00 - b11) / twice;
               const double w = sqrt(tau * tau + 1.0);
               const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
               const double sgn = t > 0.0 ? 1.0 : -1.0;
               const double n = 1.0 / sqrt(t * t + 1.0);
               right.s = -sgn * (b01 / fabs(b01)) * fabs(t) * n;
               right.c = n;
            }
         }
         const Rot2 rightT = {right.c, -right.s};
         const Rot2 left = {sym.c * rightT.c - sym.s * rightT.s, sym.c * rightT.s + sym.s * rightT.c};
         if (!(left.c == 1.0 && left.s == 0.0))
         {
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_pair(W[p][k], W[q][k], left);   // rows p, q of W
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_pair(U[k][p], U[k][q], left);   // columns p, q of U
         }
         if (!(rightT.c == 1.0 && rightT.s == 0.0))
         {
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_pair(W[k][p], W[k][q], rightT); // columns p, q of W
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_pair(V[k][p], V[k][q], rightT); // columns p, q of V
         }
         diagMax = fmax(diagMax, fmax(fabs(W[p][p]), fabs(W[q][q])));
      }
      if (clean) break;
   }
   double sg[3];
#pragma unroll
   for (int i = 0; i < 3; ++i)
   {
      const double d = W[i][i];
      sg[i] = fabs(d);
      if (d < 0.0) { U[0][i] = -U[0][i]; U[1][i] = -U[1][i]; U[2][i] = -U[2][i]; }
   }
#pragma unroll
   for (int i = 0; i < 3; ++i) sg[i] *= scale;
   int nonzero = 3;
#pragma unroll
   for (int i = 0; i < 3; ++i)
   {
      if (i >= nonzero) continue;
      int at = i;
#pragma unroll
      for (int k = 1; k < 3; ++k)
         if (k > i && sg[k] > sg[at]) at = k;
      if (sg[at] == 0.0) { nonzero = i; continue; }
      if (at != i)
      {
#pragma unroll
         for (int k = 0; k < 3; ++k)
            if (k == at)
            {
               swap_d(sg[i], sg[k]);
#pragma unroll
               for (int r = 0; r < 3; ++r) { swap_d(U[r][i], U[r][k]); swap_d(V[r][i], V[r][k]); }
            }
      }
   }
   if (sg[0] / sg[2] < 100.0 * eps) return;                       // util.cpp:424-426: reported ill-conditioned, x untouched
   int rank = nonzero;
   {
      const double cut = fmax(sg[0] * (3.0 * eps), tiny);
#pragma unroll
      for (int k = 2; k >= 0; --k)
         if (rank == k + 1 && sg[k] < cut) rank = k;
   }
   double y[3] = {0.0, 0.0, 0.0};
#pragma unroll
   for (int k = 0; k < 3; ++k)
      if (k < rank) y[k] = (1.0 / sg[k]) * ((U[0][k] * b[0] + U[1][k] * b[1]) + U[2][k] * b[2]);
#pragma unroll
   for (int r = 0; r < 3; ++r)
   {
      double acc = 0.0;
      if (rank > 0) acc = V[r][0] * y[0];
      if (rank > 1) acc = acc + V[r][1] * y[1];
      if (rank > 2) acc = acc + V[r][2] * y[2];
      x[r] = acc;
   }
}

// solveLinSys (util.cpp:413-442): Eigen's partial-pivot LU, or its Jacobi SVD when the problem says so (BATOTP_F_SVD)
__device__ __forceinline__ void solve3(unsigned flags, const double *A, const double *b, double *x)
{
   if (flags & BATOTP_F_SVD) svd3_solve(A, b, x);
   else lu3_solve(A, b, x);
}

// Robot::setA (robot.cpp:534-558): A[i][j] = (cart[i] - pmat[i][j]) / theta[j]
__device__ __forceinline__ void cspr_setA(const double *pmat, const double *th, const double *ca, double *A)
{
#pragma unroll
   for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) A[i * 3 + j] = (ca[i] - pmat[i * 3 + j]) / th[j];
}

// All channels as (value, second derivative) pairs (uniform sites): there is no sample array.  What k_samples would have stored
// for site i of a path is formed from the pairs of the site's segment -- the same search, the same row (coeffs_from_sol =
// emit_segment's formulas), the same three expressions.
struct PairSite
{
   const double2 *kmP; // the path's pairs, kmC channels per knot
   int kmC;
   int64_t seg;
   double tau, tau2, tau3, vf, af;
   __device__ __forceinline__ void init(const PathInfo &pi, int64_t i, const double *km, int kmC_)
   {
      kmP = reinterpret_cast<const double2 *>(km) + pi.koff * kmC_;
      kmC = kmC_;
      const int64_t N = pi.n;
      const double sScale = (pi.sres_c * (double)(N - 1)) / (double)(N - 1);
      const double site = sScale * (double)i;
      seg = (i < N - 1) ? i : N - 2;
      while (seg > 0 && site < pi.sres_c * (double)seg) --seg;
      while (seg < N - 2 && !(site < pi.sres_c * (double)(seg + 1))) ++seg;
      tau = (site - pi.sres_c * (double)seg) / (pi.sres_c * (double)(seg + 1) - pi.sres_c * (double)seg);
      tau2 = tau * tau; tau3 = tau2 * tau;
      vf = 1.0 / pi.sres_c; af = vf * vf;
   }
   // findInterpSegs aborts when two knots coincide (spline.cpp:81-89)
   __device__ __forceinline__ static bool coincide(const PathInfo &pi, int64_t i)
   {
      return i < pi.n - 1 && (pi.sres_c * (double)(i + 1) - pi.sres_c * (double)i) < 1e-20;
   }
   // sample `ord` (0 value, 1 d/ds, 2 d2/ds2) of input channel c
   __device__ __forceinline__ double sample(int c, int ord) const
   {
      const double2 a = kmP[seg * kmC + c], b = kmP[(seg + 1) * kmC + c];
      const Coef4 k = coeffs_from_sol(a.y, b.y, a.x, b.x);
      if (ord == 0) return k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
      if (ord == 1) return (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * vf;
      return (6 * k.c3 * tau + 2 * k.c2) * af;
   }
};

// ---------------------------------------------------------------------------------------------
// K2b: dynamics coefficients a1..a4 at the knots (ba.cpp:873-938).  One thread per knot.
// ---------------------------------------------------------------------------------------------
__global__ void k_dynamics(DevProblem P, const DevProblem *__restrict__ dP, const PathInfo *__restrict__ pinfo, int B, const double *__restrict__ samp,
                           const double *__restrict__ trig, double *__restrict__ dyn, int64_t total, double *km = nullptr, int kmC = 0,
                           batotp_path_result *__restrict__ res = nullptr)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int64_t N = pi.n, i = g - pi.koff;
   const double *__restrict__ sp = samp + pi.koff * P.Cin * 3;
   double *__restrict__ dy = dyn + pi.koff * 4 * P.d;
   const int d = P.d;
   // All channels as pairs (km != nullptr): samples through PairSite, and a1..a4 go straight into the value slots of their
   // channels (device channel Cin + 4 r + k), where K1 then solves them in place -- no sample array, no dynamics array.
   const bool fused = km != nullptr;
   PairSite ps;
   if (fused)
   {
      ps.init(pi, i, km, kmC);
      if (PairSite::coincide(pi, i) && res) atomicOr(&res[lo].status_rev, (unsigned)BATOTP_ST_SEG_ERROR);
   }
   auto S = [&](int c, int ord) -> double { return fused ? ps.sample(c, ord) : sp[((int64_t)c * 3 + ord) * N + i]; };
   // a_k of dynamics row r
   auto D = [&](int k, int r, double v) {
      if (fused) km[((pi.koff + i) * kmC + P.Cin + r * 4 + k) * 2] = v;
      else dy[((int64_t)k * d + r) * N + i] = v;
   };

   if (P.flags & BATOTP_F_PARALLEL)
   {
      // Robot::dynCSPR3DOF (robot.cpp:487-517)
      double a1[3], a2[3], a3[3], a4[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
         a1[j] = -S(P.nJ + j, 1);
         a2[j] = -S(P.nJ + j, 2);
         a3[j] = 0.0;
         a4[j] = 0.0;
      }
      a4[2] = 9.81;
      if (P.flags & BATOTP_F_PAR2SER)
      {
         // ba.cpp:916-936: A^-1 a_k through four LU solves
         double A[9], th[3], ca[3], xs[3];
#pragma unroll
         for (int j = 0; j < 3; ++j)
         {
            ca[j] = S(P.nJ + j, 0);
            th[j] = S(j, 0);
         }
         cspr_setA(dP->pmat, th, ca, A);
         solve3(P.flags, A, a1, xs); a1[0] = xs[0]; a1[1] = xs[1]; a1[2] = xs[2];
         solve3(P.flags, A, a2, xs); a2[0] = xs[0]; a2[1] = xs[1]; a2[2] = xs[2];
         solve3(P.flags, A, a3, xs); a3[0] = xs[0]; a3[1] = xs[1]; a3[2] = xs[2];
         solve3(P.flags, A, a4, xs); a4[0] = xs[0]; a4[1] = xs[1]; a4[2] = xs[2];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
         D(0, j, a1[j]);
         D(1, j, a2[j]);
         D(2, j, a3[j]);
         D(3, j, a4[j]);
      }
      return;
   }

   // Robot::dynRR (robot.cpp:377-431), joint values in degrees
   const double kDeg2Rad = 3.14159265358979323846 / 180.0;
   const double kG = 9.81;
   double A1 = .4, A2 = .6, m1 = 4, m2 = 8;
   const double th1 = kDeg2Rad * S(0, 0);
   const double th2 = kDeg2Rad * S(1, 0);
   const double dth1 = kDeg2Rad * S(0, 1);
   const double dth2 = kDeg2Rad * S(1, 1);
   const double ddth1 = kDeg2Rad * S(0, 2);
   const double ddth2 = kDeg2Rad * S(1, 2);
   double c1, c2, c12, s2;
   if (trig != nullptr)
   {
      const double *__restrict__ tg = trig + pi.koff * 4;
      c1 = tg[i]; c2 = tg[N + i]; c12 = tg[2 * N + i]; s2 = tg[3 * N + i];
   }
   else
   {
      // device libm: within an ulp of glibc but not bit-identical (RR trig policy)
      c1 = cos(th1); c2 = cos(th2); c12 = cos(th1 + th2); s2 = sin(th2);
   }
   const double A11 = .25 * m1 * A1 * A1 + m2 * (A1 * A1 + .25 * A2 * A2 + A1 * A2 * c2);
   const double A12 = .5 * m2 * (.5 * A2 * A2 + A1 * A2 * c2);
   const double A22 = .25 * m2 * A2 * A2;
   const double ccFact = m2 * A1 * A2 * s2;
   D(0, 0, A11 * dth1 + A12 * dth2);
   D(0, 1, A12 * dth1 + A22 * dth2);
   D(1, 0, A11 * ddth1 + A12 * ddth2 - ccFact * dth2 * (dth1 + .5 * dth2));
   D(1, 1, A12 * ddth1 + A22 * ddth2 - .5 * ccFact * dth1 * dth1);
   D(2, 0, 10 * dth1);
   D(2, 1, 10 * dth2);
   D(3, 0, .5 * kG * (m1 * A1 * c1 + m2 * (2.0 * A1 * c1 + A2 * c12)));
   D(3, 1, .5 * kG * m2 * A2 * c12);
}

// ---------------------------------------------------------------------------------------------
// K2c: dynamics coefficients of a serial chain of revolute joints (BASELINE config 3: 7-DOF arm with torque
// limits) -- one more case of Robot::dynSerial's switch (robot.cpp:349-360), table-driven (batotp_serial_model).
// tau = a1 sddot + a2 sdot^2 + a3 sdot + a4 (robot.cpp:368-372) with q = q(s):
//   a1 = M q' = RNEA(q, 0, q'), a2 = M q'' + C(q, q') q' = RNEA(q, q', q''), both without gravity;
//   a3 = fv .* q';  a4 = g(q) = RNEA(q, 0, 0) with the base accelerating at -gravity.
// Recursive Newton-Euler in link coordinates, zero-aligned frames, Rodrigues rotation about each joint's axis.
// One lane per knot, the model staged in LDS (every lane reads the same word: broadcast), the link loops unrolled
// so that the link forces of the outward pass stay in registers.  The expression order is the one of the oracle's
// bo_rnea (oracle/batotp_oracle_dyn.c); there is no reference model for this robot (parity unpinned).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o)
{
   o[0] = a[1] * b[2] - a[2] * b[1];
   o[1] = a[2] * b[0] - a[0] * b[2];
   o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void rot_axis(const double *a, double c, double s, double *R)
{
   const double omc = 1.0 - c;
   R[0] = omc * a[0] * a[0] + c;
   R[1] = omc * a[0] * a[1] - s * a[2];
   R[2] = omc * a[0] * a[2] + s * a[1];
   R[3] = omc * a[1] * a[0] + s * a[2];
   R[4] = omc * a[1] * a[1] + c;
   R[5] = omc * a[1] * a[2] - s * a[0];
   R[6] = omc * a[2] * a[0] - s * a[1];
   R[7] = omc * a[2] * a[1] + s * a[0];
   R[8] = omc * a[2] * a[2] + c;
}
__device__ __forceinline__ void mat_v(const double *R, const double *v, double *o)
{
   o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
   o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
   o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
__device__ __forceinline__ void matT_v(const double *R, const double *v, double *o)
{
   o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
   o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
   o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}
__device__ __forceinline__ void inertia_v(const double *I, const double *v, double *o)
{
   o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
   o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
   o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
}

// one pass of the recursion at one configuration (cosines / sines of the joint angles, rates, accelerations, base
// acceleration); tau[n_links]
__device__ __forceinline__ void rnea_pass(const batotp_serial_model &m, const double *cq, const double *sq, const double *qd,
                                          const double *qdd, const double *a0, double *tau)
{
   const int n = m.n_links;
   double F[BATOTP_MAX_LINKS][3], Nn[BATOTP_MAX_LINKS][3];
   double w[3] = {0, 0, 0}, wd[3] = {0, 0, 0}, a[3] = {a0[0], a0[1], a0[2]};
#pragma unroll
   for (int i = 0; i < BATOTP_MAX_LINKS; ++i)
   {
      if (i < n)
      {
         const batotp_serial_link &L = m.link[i];
         const double ax[3] = {L.axis[0], L.axis[1], L.axis[2]}, off[3] = {L.off[0], L.off[1], L.off[2]};
         const double com[3] = {L.com[0], L.com[1], L.com[2]};
         const double In[6] = {L.inertia[0], L.inertia[1], L.inertia[2], L.inertia[3], L.inertia[4], L.inertia[5]};
         double R[9], t1[3], t2[3], t3[3], wp[3], wdp[3], ap[3], zq[3], ac[3], Iw[3], Iwd[3];
         rot_axis(ax, cq[i], sq[i], R);
         cross3(wd, off, t1);
         cross3(w, off, t2);
         cross3(w, t2, t3);
#pragma unroll
         for (int k = 0; k < 3; ++k) t1[k] = a[k] + t1[k] + t3[k];
         matT_v(R, t1, ap);
         matT_v(R, w, wp);
         matT_v(R, wd, wdp);
#pragma unroll
         for (int k = 0; k < 3; ++k) zq[k] = ax[k] * qd[i];
         cross3(wp, zq, t2);
#pragma unroll
         for (int k = 0; k < 3; ++k)
         {
            w[k] = wp[k] + zq[k];
            wd[k] = wdp[k] + ax[k] * qdd[i] + t2[k];
            a[k] = ap[k];
         }
         cross3(wd, com, t1);
         cross3(w, com, t2);
         cross3(w, t2, t3);
#pragma unroll
         for (int k = 0; k < 3; ++k) ac[k] = a[k] + t1[k] + t3[k];
#pragma unroll
         for (int k = 0; k < 3; ++k) F[i][k] = L.mass * ac[k];
         inertia_v(In, wd, Iwd);
         inertia_v(In, w, Iw);
         cross3(w, Iw, t1);
#pragma unroll
         for (int k = 0; k < 3; ++k) Nn[i][k] = Iwd[k] + t1[k];
      }
   }
   double fc[3] = {0, 0, 0}, nc[3] = {0, 0, 0};
#pragma unroll
   for (int i = BATOTP_MAX_LINKS - 1; i >= 0; --i)
   {
      if (i < n)
      {
         const batotp_serial_link &L = m.link[i];
         const double ax[3] = {L.axis[0], L.axis[1], L.axis[2]}, com[3] = {L.com[0], L.com[1], L.com[2]};
         double f[3], nn[3], t1[3], t2[3] = {0, 0, 0};
         cross3(com, F[i], t1);
         if (i + 1 < n)
         {
            const batotp_serial_link &Lc = m.link[(i + 1) & (BATOTP_MAX_LINKS - 1)];
            const double offc[3] = {Lc.off[0], Lc.off[1], Lc.off[2]};
            cross3(offc, fc, t2);
         }
#pragma unroll
         for (int k = 0; k < 3; ++k)
         {
            f[k] = F[i][k] + fc[k];
            nn[k] = Nn[i][k] + nc[k] + t1[k] + t2[k];
         }
         tau[i] = ax[0] * nn[0] + ax[1] * nn[1] + ax[2] * nn[2];
         double R[9];
         rot_axis(ax, cq[i], sq[i], R);
         mat_v(R, f, fc);
         mat_v(R, nn, nc);
      }
   }
}

constexpr int KDS_BLOCK = 128;
__global__ void __launch_bounds__(KDS_BLOCK) k_dyn_serial(const batotp_serial_model *__restrict__ model, int Cin,
                                                          const PathInfo *__restrict__ pinfo, int B, const double *__restrict__ samp,
                                                          const double *__restrict__ trig, double *__restrict__ dyn, int64_t total,
                                                          double *km = nullptr, int kmC = 0, batotp_path_result *__restrict__ res = nullptr)
{
   __shared__ batotp_serial_model sm;
   {
      // cooperative copy of the table (8-byte words)
      const double *src = reinterpret_cast<const double *>(model);
      double *dst = reinterpret_cast<double *>(&sm);
      for (int k = threadIdx.x; k < (int)(sizeof(batotp_serial_model) / 8); k += KDS_BLOCK) dst[k] = src[k];
   }
   __syncthreads();
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int64_t N = pi.n, i = g - pi.koff;
   const int nl = sm.n_links;
   const double *__restrict__ sp = samp + pi.koff * Cin * 3;
   const double *__restrict__ tg = trig ? trig + pi.koff * 2 * nl : nullptr;
   double *__restrict__ dy = dyn + pi.koff * 4 * nl;
   const double kDeg2Rad = 3.14159265358979323846 / 180.0;
   const double unit = sm.degrees ? kDeg2Rad : 1.0;
   // all channels as pairs (km != nullptr): as in k_dynamics
   const bool fused = km != nullptr;
   PairSite ps;
   if (fused)
   {
      ps.init(pi, i, km, kmC);
      if (PairSite::coincide(pi, i) && res) atomicOr(&res[lo].status_rev, (unsigned)BATOTP_ST_SEG_ERROR);
   }
   auto S = [&](int c, int ord) -> double { return fused ? ps.sample(c, ord) : sp[((int64_t)c * 3 + ord) * N + i]; };
   auto D = [&](int k, int r, double v) {
      if (fused) km[((pi.koff + i) * kmC + Cin + r * 4 + k) * 2] = v;
      else dy[((int64_t)k * nl + r) * N + i] = v;
   };

   double cq[BATOTP_MAX_LINKS], sq[BATOTP_MAX_LINKS], q1[BATOTP_MAX_LINKS], q2[BATOTP_MAX_LINKS], z[BATOTP_MAX_LINKS];
#pragma unroll
   for (int j = 0; j < BATOTP_MAX_LINKS; ++j)
   {
      cq[j] = 1; sq[j] = 0; q1[j] = 0; q2[j] = 0; z[j] = 0;
      if (j < nl)
      {
         const double q = unit * S(j, 0);
         q1[j] = unit * S(j, 1);
         q2[j] = unit * S(j, 2);
         if (tg) { cq[j] = tg[(int64_t)j * N + i]; sq[j] = tg[(int64_t)(nl + j) * N + i]; }
         else { cq[j] = cos(q); sq[j] = sin(q); } // device libm: not bit-identical to glibc (BATOTP_F_HOST_TRIG)
      }
   }
   const double zero3[3] = {0, 0, 0};
   const double g0[3] = {-sm.gravity[0], -sm.gravity[1], -sm.gravity[2]};
   double t1[BATOTP_MAX_LINKS], t2[BATOTP_MAX_LINKS], t4[BATOTP_MAX_LINKS];
   rnea_pass(sm, cq, sq, z, q1, zero3, t1);
   rnea_pass(sm, cq, sq, q1, q2, zero3, t2);
   rnea_pass(sm, cq, sq, z, z, g0, t4);
#pragma unroll
   for (int j = 0; j < BATOTP_MAX_LINKS; ++j)
   {
      if (j < nl)
      {
         D(0, j, t1[j]);
         D(1, j, t2[j]);
         D(2, j, sm.link[j].fv * q1[j]);
         D(3, j, t4[j]);
      }
   }
}

// ---------------------------------------------------------------------------------------------
// Point evaluation shared by K3 and K4.  A path is handled by a group of G lanes: lane j of the
// group owns joint (and dynamics row) j; with G == 1 one lane loops over all joints.  Everything
// that is one value per path (cursor, sdot, bisection bracket ...) is kept redundantly in every
// lane of the group, so control flow is uniform inside a group and the only cross-lane traffic is
// the min / max over joints (DPP butterflies, device_math.h).
//   FEAT: which constraint families are compiled in (0 joint velocity/acceleration only,
//         1 + Cartesian, 2 + torque in serial form, 3 + torque of a parallel mechanism, isPar2Ser = 0);
//         the narrow variants carry fewer registers and instructions.
//   UNI: knot sites are sres*k exactly (every path built by batotp_hip_precompute): they are
//        computed instead of loaded, which removes the dependent loads of the segment search.
// ---------------------------------------------------------------------------------------------
template <int G, int FEAT, bool UNI>
struct Pt
{
   static constexpr int PER = (G >= BATOTP_MAX_JOINTS) ? 1 : BATOTP_MAX_JOINTS / G; // joints per lane: G = 1: 8, 2: 4, 4: 2, >= 8: 1
   // G == 16 (except for the parallel-mechanism variant): the group is two halves of 8 lanes; lane j and
   // lane j+8 own the same joint, the lower half evaluates the upper bounds of the sddot interval, the
   // upper half the lower bounds: one divide per lane and one reduction per constraint check.
   static constexpr bool SPLIT = (G == 16 && FEAT != 3);
   // G == 32: a path owns half a wavefront = 4 candidate slots x 8 joint lanes; the bisection evaluates
   // four candidate sdot values per pass (speculation on its own, fully determined, candidate sequence)
   static constexpr bool SPEC = (G == 32);
   static constexpr int RED = (SPLIT || SPEC) ? 8 : G; // lanes one joint reduction spans
   static constexpr bool PAR = (FEAT == 3); // FEAT: 0 joint vel/acc only, 1 + Cartesian, 2 + torque (serial form), 3 + torque (parallel mechanism)

   // path constants
   const double *__restrict__ sC;
   const double *__restrict__ coef; // this path's [N][C][4]   (FEAT >= 0)
   const double2 *__restrict__ km;  // FEAT == -1 (compact splines): this path's [N][Cin] (value, second derivative) pairs
   int n;
   int C, nJ, nC, nIn;
   unsigned flags;
   int parallel_now;
   double sresC, vfact, afact, thrV, thrA, quadA, quadA2, sdotCap, sddotMax, cartAccMaxSQ, cartVelMax;
   const double *pmat;
   // lane constants (limits of this lane's joints)
   double vmax[PER], amax[PER], tmax[PER], tmin[PER];
   // reverse curve (forward sweep only)
   const double *mvc; // (s, sdot) pairs; no __restrict__: with BATOTP_F_CURVES_IN_PLACE it points into the buffer `out` writes
   int nMvc;
   int dir;
   // cursor state (ba.h:94-103,144-145)
   int segC, segMVC;
   double tauC, tauMVC, sCur, sdotCur, sddotL, sddotH, sdotMin;
   // point buffers of this lane's joints
   double thD[PER], thD2[PER], a1[PER], a2[PER], a3[PER], a4[PER];
   double cq0, cq1, cq2; // CartAccCoeffs
   // parallel mechanism (isPar2Ser = 0): everything, in every lane
   double thp[PAR ? 3 : 1], cap[PAR ? 3 : 1], Am[PAR ? 9 : 1], pa1[PAR ? 3 : 1], pa2[PAR ? 3 : 1], pa3[PAR ? 3 : 1], pa4[PAR ? 3 : 1];
   unsigned status;
   int nfail;
   int sink; // consumer of the prefetch touches (keeps them alive; written to the result row)
   int side; // SPLIT: 0 = upper-bound half, 1 = lower-bound half
   int cslot, pbase; // SPEC: candidate slot 0..3 of this lane, first lane of the path in the wavefront
#ifdef BK_PROFILE_SECTIONS
   unsigned long long cycA, cycB, cycC, cycD; // diagnostic build: cycles in sdot_lim / eval_partials / bisection / rest
#endif
   // register cache of the last spline row / reverse-curve segment read (dense-step paths stay on
   // one segment for many consecutive evaluations)
   int rowSeg, mvcSeg;
   Coef4 rowTh[PER];
   double mvcS0, mvcS1, mvcD0, mvcD1;
};

// BA::updateCurSeg (ba.cpp:1617-1652).  KIND 0: sites computed as sres*k; 1: sites loaded from s[k];
// 2: the (s, sdot) pairs of the reverse curve.  A NaN position, for which the reference never
// terminates, sets BATOTP_ST_NONFINITE instead.
template <int KIND>
__device__ __forceinline__ double site_at(const double *__restrict__ s, double sres, int k)
{
   if (KIND == 0) return sres * (double)k;
   if (KIND == 1) return s[k];
   return s[2 * k];
}

template <int KIND>
__device__ __forceinline__ void update_cur_seg(const double *__restrict__ s, double sres, int n, double sCur, int &curSeg,
                                               double &tau, unsigned &status)
{
   const int lastSeg = n - 2;
   double sSeg, sNext;
   for (;;)
   {
      sSeg = site_at<KIND>(s, sres, curSeg);
      sNext = site_at<KIND>(s, sres, curSeg + 1);
      if (sCur >= sSeg && sCur <= sNext) break;
      bool moved = false;
      if (sCur > sSeg)
      {
         if (curSeg >= lastSeg) { curSeg = lastSeg; break; }
         ++curSeg; moved = true;
      }
      if (sCur < sSeg)
      {
         if (curSeg <= 0) { curSeg = 0; break; }
         --curSeg; moved = true;
      }
      if (!moved) { status |= BATOTP_ST_NONFINITE; break; }
   }
   tau = (sCur - sSeg) / (sNext - sSeg);
}

// BA::evalSplinePartials + evalCartQuadCoeffs (ba.cpp:1341-1439)
// SRC: where the coefficient row of the cursor's segment comes from -- src(c) returns c0..c3 of device channel c (a row in
// memory: RowAt; formed on the fly from (value, second derivative) pairs: PairRowAt)
struct RowAt
{
   const double *__restrict__ row;
   __device__ __forceinline__ Coef4 operator()(int c) const { return *reinterpret_cast<const Coef4 *>(row + c * 4); }
};
struct PairRowAt
{
   const double2 *__restrict__ kp; // knot of the segment's left end, C channels per knot
   int C;
   __device__ __forceinline__ Coef4 operator()(int c) const
   {
      const double2 a = kp[c], b = kp[C + c];
      return coeffs_from_sol(a.y, b.y, a.x, b.x);
   }
};
template <int G, int FEAT, bool UNI, typename SRC>
__device__ __forceinline__ void eval_partials_src(Pt<G, FEAT, UNI> &t, int j, const SRC &src)
{
   constexpr bool PAR = (FEAT == 3);
   const double tau = t.tauC, tau2 = tau * tau, tau3 = tau2 * tau;

#pragma unroll
   for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
   {
      const int jj = j + q * G;
      if (jj < t.nJ)
      {
         const Coef4 k = src(jj);
         t.thD[q] = (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * t.vfact;
         t.thD2[q] = (6 * k.c3 * tau + 2 * k.c2) * t.afact;
      }
   }
   if (FEAT >= 1 && (t.flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)))
   {
      double v[3], a[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
      {
         const Coef4 k = src(t.nJ + i);
         if (PAR) t.cap[PAR ? i : 0] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
         v[i] = (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * t.vfact;
         a[i] = (6 * k.c3 * tau + 2 * k.c2) * t.afact;
      }
      t.cq0 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      t.cq1 = 2 * (v[0] * a[0] + v[1] * a[1] + v[2] * a[2]);
      t.cq2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
   }
   if (FEAT >= 2 && (t.flags & BATOTP_F_TRQ_ON))
   {
      if (PAR && t.parallel_now)
      {
#pragma unroll
         for (int r = 0; r < 3; ++r)
         {
            const Coef4 kt = src(r);
            t.thp[PAR ? r : 0] = kt.c3 * tau3 + kt.c2 * tau2 + kt.c1 * tau + kt.c0;
            const Coef4 kd[4] = {src(t.nIn + r * 4), src(t.nIn + r * 4 + 1), src(t.nIn + r * 4 + 2), src(t.nIn + r * 4 + 3)};
            t.pa1[PAR ? r : 0] = kd[0].c3 * tau3 + kd[0].c2 * tau2 + kd[0].c1 * tau + kd[0].c0;
            t.pa2[PAR ? r : 0] = kd[1].c3 * tau3 + kd[1].c2 * tau2 + kd[1].c1 * tau + kd[1].c0;
            t.pa3[PAR ? r : 0] = kd[2].c3 * tau3 + kd[2].c2 * tau2 + kd[2].c1 * tau + kd[2].c0;
            t.pa4[PAR ? r : 0] = kd[3].c3 * tau3 + kd[3].c2 * tau2 + kd[3].c1 * tau + kd[3].c0;
         }
         cspr_setA(t.pmat, t.thp, t.cap, t.Am); // ba.cpp:1407-1410
      }
      else
      {
#pragma unroll
         for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
         {
            const int jj = j + q * G;
            if (jj < t.nJ)
            {
               const Coef4 k1 = src(t.nIn + jj * 4), k2 = src(t.nIn + jj * 4 + 1), k3 = src(t.nIn + jj * 4 + 2), k4 = src(t.nIn + jj * 4 + 3);
               t.a1[q] = k1.c3 * tau3 + k1.c2 * tau2 + k1.c1 * tau + k1.c0;
               t.a2[q] = k2.c3 * tau3 + k2.c2 * tau2 + k2.c1 * tau + k2.c0;
               t.a3[q] = k3.c3 * tau3 + k3.c2 * tau2 + k3.c1 * tau + k3.c0;
               t.a4[q] = k4.c3 * tau3 + k4.c2 * tau2 + k4.c1 * tau + k4.c0;
            }
         }
      }
   }
}

template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void eval_partials_row(Pt<G, FEAT, UNI> &t, int j, const double *__restrict__ row)
{
   eval_partials_src(t, j, RowAt{row});
}


// FEAT <= 0 (joint velocity / acceleration limits only): this lane's coefficients c1..c3 stay in
// registers while the cursor stays on the segment (group-uniform test).  FEAT == 0 reads them from the
// coefficient rows; FEAT == -1 forms them from the knot values and second derivatives with
// emit_segment's formulas (spline.cpp:203-209) -- half the bytes per knot, no coefficient array.
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void eval_partials_cached(Pt<G, FEAT, UNI> &t, int j)
{
   if (t.segC != t.rowSeg)
   {
#pragma unroll
      for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
      {
         const int jj = j + q * G;
         if (jj < t.nJ)
         {
            if (FEAT < 0)
            {
               const unsigned at = (unsigned)(t.segC * t.nIn + jj);
               const double2 kl = t.km[at], kr = t.km[at + t.nIn]; // knots segC and segC + 1 of this joint
               const double solL = kl.y, solR = kr.y, yL = kl.x, yR = kr.x;
               Coef4 k;
               k.c3 = div6(solR - solL);
               k.c2 = solL / 2.0;
               k.c1 = yR - yL - div6(solR + 2 * solL);
               k.c0 = yL;
               t.rowTh[q] = k;
            }
            else t.rowTh[q] = *reinterpret_cast<const Coef4 *>(t.coef + (unsigned)(t.segC * t.C * 4) + jj * 4);
         }
      }
      t.rowSeg = t.segC;
   }
   const double tau = t.tauC, tau2 = tau * tau;
#pragma unroll
   for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
   {
      const int jj = j + q * G;
      if (jj < t.nJ)
      {
         const Coef4 k = t.rowTh[q];
         t.thD[q] = (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * t.vfact;
         t.thD2[q] = (6 * k.c3 * tau + 2 * k.c2) * t.afact;
      }
   }
}

template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void eval_partials(Pt<G, FEAT, UNI> &t, int j)
{
   update_cur_seg<UNI ? 0 : 1>(t.sC, t.sresC, t.n, t.sCur, t.segC, t.tauC, t.status);
#ifndef BK_NO_ROWCACHE
   if (FEAT <= 0)
#else
   if (FEAT < 0)
#endif
   {
      eval_partials_cached(t, j);
      return;
   }
   eval_partials_row(t, j, t.coef + (unsigned)(t.segC * t.C * 4));
}

// BA::updateCurSeg on the reverse curve (ba.cpp:1592) with the current segment's two (s, sdot)
// pairs cached in registers: the common case "still inside the cached segment" (dense-step paths)
// costs no load; otherwise the literal walk runs and the cache is refilled.
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void mvc_walk(Pt<G, FEAT, UNI> &t)
{
   const double sCur = t.sCur;
   if (t.mvcSeg == t.segMVC && sCur >= t.mvcS0 && sCur <= t.mvcS1)
   {
      t.tauMVC = (sCur - t.mvcS0) / (t.mvcS1 - t.mvcS0);
      return;
   }
   update_cur_seg<2>(t.mvc, 0.0, t.nMvc, sCur, t.segMVC, t.tauMVC, t.status);
   const double2 a = *reinterpret_cast<const double2 *>(t.mvc + 2 * t.segMVC);
   const double2 b = *reinterpret_cast<const double2 *>(t.mvc + 2 * t.segMVC + 2);
   t.mvcS0 = a.x; t.mvcD0 = a.y; t.mvcS1 = b.x; t.mvcD1 = b.y;
   t.mvcSeg = t.segMVC;
}

// BA::evalsdot, "linear" (ba.cpp:1590-1607)
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ double eval_sdot(Pt<G, FEAT, UNI> &t)
{
   mvc_walk(t);
   const double v = t.mvcD0 + t.tauMVC * (t.mvcD1 - t.mvcD0);
   return dmax(v, t.sdotMin);
}

// BA::sdotLim (ba.cpp:1204-1236).  The joint-velocity limits use the theta' of the previous
// evalSplinePartials call, exactly as the reference does.
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void sdot_lim(Pt<G, FEAT, UNI> &t, int j, double &sdot)
{
   if (t.dir == 1)
   {
      const double sdotMVC = eval_sdot(t);
      if (sdot > sdotMVC) sdot = sdotMVC;
   }
   sdot = dmin(sdot, t.sdotCap);
   sdot = dmax(sdot, t.sdotMin);
   double lim = kInf;
#pragma unroll
   for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
   {
      const int jj = j + q * G;
      if (jj < t.nJ && fabs(t.thD[q]) > t.thrV) lim = dmin(lim, fabs(t.vmax[q] / t.thD[q]));
   }
   lim = grp_min<Pt<G, FEAT, UNI>::RED>(lim); // SPLIT / SPEC: every 8-lane part holds the same joints
   sdot = dmin(sdot, lim);
   if (FEAT >= 1 && (t.flags & BATOTP_F_CART_VEL_ON) && t.cq0 > t.quadA) sdot = dmin(sdot, t.cartVelMax / sqrt(t.cq0));
}

// solveQuadratic (util.cpp:361-383)
__device__ __forceinline__ int solve_quadratic(double A, double B, double C, double &sol1, double &sol2)
{
   if (fabs(A) < 1e-308)
   {
      if (fabs(B) < 1e-308) return -2;
      sol1 = -C / B; sol2 = sol1;
      return 0;
   }
   const double rad = B * B - 4 * A * C;
   if (rad < 0) return -1;
   const double den = 2 * A;
   const double F1 = -B / den, F2 = sqrt(rad) / den;
   sol1 = F1 + F2;
   sol2 = F1 - F2;
   return 0;
}

// BA::verifySecondOrderConstraints (ba.cpp:1449-1581).  Each lane intersects the sddot intervals
// of its own joints; the group then takes min(H) / max(L).  The reference leaves its joint loops
// early as soon as L > H; since H only shrinks and L only grows along the loops, that is the same
// predicate as L > H after the full reduction, and whenever the point is admissible both agree on
// sddotL / sddotH bit for bit.  The unconditional "violated" exit of the zero-velocity case
// (ba.cpp:1522) is folded into the H reduction as -inf (sddotL/H are never read after a violation).
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ bool verify_second_order(Pt<G, FEAT, UNI> &t, int j, double sdotCur)
{
   constexpr bool PAR = (FEAT == 3);
   const double sdotSQ = sdotCur * sdotCur;
   double H = t.sddotMax, L = -t.sddotMax;

   if (FEAT >= 2 && (t.flags & BATOTP_F_TRQ_ON))
   {
      if (PAR && t.parallel_now)
      {
         // ba.cpp:1463-1491: two 3x3 solves per joint
         double cStar[3];
#pragma unroll
         for (int i = 0; i < 3; ++i) cStar[i] = sdotSQ * t.pa2[PAR ? i : 0] + sdotCur * t.pa3[PAR ? i : 0] + t.pa4[PAR ? i : 0];
#pragma unroll
         for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
         {
            const int jj = j + q * G;
            if (jj < t.nJ)
            {
               double sol[2];
#pragma unroll
               for (int ii = 0; ii < 2; ++ii)
               {
                  const double lim = ii == 0 ? t.tmin[q] : t.tmax[q];
                  double As[9], bs[3], xs[3];
#pragma unroll
                  for (int k = 0; k < 9; ++k) As[k] = t.Am[PAR ? k : 0];
#pragma unroll
                  for (int k = 0; k < 3; ++k)
                  {
                     // column jj of A replaced by -a1, rhs = cStar - A[:,jj]*limit
                     const double akj = (jj == 0) ? t.Am[PAR ? k * 3 + 0 : 0] : (jj == 1) ? t.Am[PAR ? k * 3 + 1 : 0] : t.Am[PAR ? k * 3 + 2 : 0];
                     bs[k] = cStar[k] - akj * lim;
                     const double na1 = -t.pa1[PAR ? k : 0];
                     if (jj == 0) As[k * 3 + 0] = na1;
                     if (jj == 1) As[k * 3 + 1] = na1;
                     if (jj == 2) As[k * 3 + 2] = na1;
                  }
                  solve3(t.flags, As, bs, xs);
                  sol[ii] = (jj == 0) ? xs[0] : (jj == 1) ? xs[1] : xs[2];
               }
               H = dmin(H, dmax(sol[0], sol[1]));
               L = dmax(L, dmin(sol[0], sol[1]));
            }
         }
      }
      else
      {
         // ba.cpp:1495-1509
#pragma unroll
         for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
         {
            const int jj = j + q * G;
            if (jj < t.nJ)
            {
               const double a1pt = t.a1[q];
               const double tmp1 = t.a3[q] * sdotCur + t.a4[q];
               if (!(fabs(a1pt) < t.thrV))
               {
                  const double tmp2 = t.a2[q] * sdotSQ + tmp1;
                  if (Pt<G, FEAT, UNI>::SPLIT)
                  {
                     // max(s0, s1) and min(s0, s1) are each ONE of the two quotients: correctly rounded
                     // subtraction and division are monotone, so the larger numerator over a positive a1
                     // (the smaller one over a negative a1) is the maximum.  One divide per half.
                     const double n0 = t.tmax[q] - tmp2, n1 = t.tmin[q] - tmp2;
                     const bool firstIsMax = (n0 >= n1) == (a1pt > 0);
                     if (t.side == 0) H = dmin(H, (firstIsMax ? n0 : n1) / a1pt);
                     else L = dmax(L, (firstIsMax ? n1 : n0) / a1pt);
                  }
                  else
                  {
                     const double s0 = (t.tmax[q] - tmp2) / a1pt;
                     const double s1 = (t.tmin[q] - tmp2) / a1pt;
                     H = dmin(H, dmax(s0, s1));
                     L = dmax(L, dmin(s0, s1));
                  }
               }
            }
         }
      }
   }
   bool force = false;
   if (t.flags & BATOTP_F_JNT_ACC_ON)
   {
      // ba.cpp:1514-1534
#pragma unroll
      for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
      {
         const int jj = j + q * G;
         if (jj < t.nJ)
         {
            const double vpt = t.thD[q];
            if (fabs(vpt) < t.thrV)
            {
               if (!(fabs(t.thD2[q]) < t.thrA))
               {
                  if (sdotSQ > t.amax[q] / fabs(t.thD2[q])) force = true;
               }
            }
            else
            {
               const int svpt = sgn(vpt);
               const double vTerm = t.thD2[q] * sdotSQ;
               if (Pt<G, FEAT, UNI>::SPLIT)
               {
                  if (t.side == 0) H = dmin(H, (svpt * t.amax[q] - vTerm) / vpt);
                  else L = dmax(L, (-svpt * t.amax[q] - vTerm) / vpt);
               }
               else
               {
                  H = dmin(H, (svpt * t.amax[q] - vTerm) / vpt);
                  L = dmax(L, (-svpt * t.amax[q] - vTerm) / vpt);
               }
            }
         }
      }
   }
   double Hred;
   if (Pt<G, FEAT, UNI>::SPLIT)
   {
      // lower half reduces min(H), upper half min(-L) = -max(L) in the same three DPP steps, then the
      // halves exchange their results (lane i <-> i +- 8 owns the same joint)
      const double mine = grp_min<8>(t.side == 0 ? (force ? -kInf : H) : -L);
      const double other = dpp_mov<DPP_ROW_ROR8>(mine);
      Hred = t.side == 0 ? mine : other;
      L = -(t.side == 0 ? other : mine);
   }
   else
   {
      Hred = force ? -kInf : H;
      grp_min_max<Pt<G, FEAT, UNI>::RED>(Hred, L);
   }
   t.sddotH = Hred;
   t.sddotL = L;
   if (L > Hred) return true; // also the folded "force" exit: L >= -sddotMax > -inf

   if (FEAT >= 1 && (t.flags & BATOTP_F_CART_ACC_ON))
   {
      // ba.cpp:1535-1579
      const double A = t.cq0;
      if (A > t.quadA)
      {
         const double Bq = t.cq1 * sdotSQ;
         const double Cq = t.cq2 * sdotSQ * sdotSQ - t.cartAccMaxSQ;
         double sol1 = 0, sol2 = 0;
         const int ef = solve_quadratic(A, Bq, Cq, sol1, sol2);
         if (ef == -1) return true;
         const double cmax = dmax(sol1, sol2), cmin = dmin(sol1, sol2);
         t.sddotH = dmin(t.sddotH, cmax);
         t.sddotL = dmax(t.sddotL, cmin);
         if (t.sddotL > t.sddotH) return true;
      }
      else
      {
         const double Cq = t.cq2;
         if (Cq < t.quadA2) return false;
         if (sdotSQ * sdotSQ > t.cartAccMaxSQ / Cq) return true;
         return false;
      }
   }
   return false;
}

// (num / den) < thr, decided exactly as the correctly rounded division would decide it, without
// performing the division unless num/den lies within 1e-14 (relative) of the threshold: the quotient
// is a monotone function of num, so far from the threshold the comparison of num with thr*den is
// decisive.  Takes two divisions off the dependent chain of every bisection iteration.
__device__ __forceinline__ bool ratio_lt(double num, double den, double thr)
{
   const double p = thr * den;
   if (den > 0.0 && num >= 0.0 && p > 1e-290 && p < 1e290)
   {
      if (num < p * (1.0 - 1e-14)) return true;
      if (num > p * (1.0 + 1e-14)) return false;
   }
   return num / den < thr;
}

// BA::applyAccelConstraintsBisectionPt (ba.cpp:1248-1332).  Returns 0, or -1 on the failure exits of
// ba.cpp:1307-1319, in which case sddot is left untouched (the reference's caller ignores the code).
template <int G, int FEAT, bool UNI, bool DOEVAL = true>
__device__ __forceinline__ int apply_accel_bisection(Pt<G, FEAT, UNI> &t, int j, double &sddot, int &nIter)
{
   const double sdotErrThresh = .001;
   double lowFact = .01;
   const double sdotMin = 0;
   double sdotGood = sdotMin, sdotGoodLast;
   bool anyGoodIter = false;
   double sdotL = sdotGood;
   double sdotH = t.sdotCur;
   double sdotCur = sdotH;
   nIter = 0;

   BK_TICK(tb0);
   if (DOEVAL) eval_partials(t, j); // ba.cpp:1265
   BK_TICK(tb1);
   BK_ACC(t.cycB, tb0, tb1);

   for (;;)
   {
      const bool isViol = verify_second_order(t, j, sdotCur);
      if (isViol)
      {
         sdotH = sdotCur;
         if (!anyGoodIter)
         {
            lowFact *= 2.0;
            sdotL = dmax(.999 * sdotMin, (1.0 - lowFact) * sdotH);
         }
      }
      else
      {
         if (nIter == 0) break;
         anyGoodIter = true;
         sdotGoodLast = sdotGood;
         sdotGood = sdotCur;
         if (ratio_lt(fabs(sdotGood - sdotGoodLast), sdotGood, sdotErrThresh) || sdotCur < sdotMin)
         {
            t.sdotCur = sdotCur;
            break;
         }
         sdotL = sdotCur;
      }
      nIter++;
      if (nIter > 100) return -1;
      if (sdotCur < 0) return -1;
      if (!anyGoodIter)
      {
         if (ratio_lt(sdotH - sdotL, sdotH, 1e-20)) return -1; // only evaluated while no feasible point is known
      }
      sdotCur = .5 * (sdotH + sdotL);
   }
   sddot = (t.dir == 1) ? t.sddotH : t.sddotL;
   return 0;
}

// The same bisection with four candidates evaluated per pass (SPEC lane layout).  The reference's loop
// is replayed literally; the only thing speculated is WHICH sdot values it will ask for next, and a
// speculated value is used only if it is bit-identical to the value the replay then really asks for,
// so a wrong guess costs a pass, never a different result.  Guesses: while no feasible point is known
// the next candidates follow from assuming "violated" (geometric shrink, ba.cpp:1281-1285,1320);
// afterwards both children of the current midpoint.
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ int apply_accel_bisection_spec(Pt<G, FEAT, UNI> &t, int j, double &sddot, int &nIter)
{
   const double sdotErrThresh = .001;
   double lowFact = .01;
   const double sdotMin = 0;
   double sdotGood = sdotMin, sdotGoodLast;
   bool anyGoodIter = false;
   double sdotL = sdotGood;
   double sdotH = t.sdotCur;
   double sdotCur = sdotH;
   nIter = 0;

   BK_TICK(tb0);
   eval_partials(t, j); // ba.cpp:1265
   BK_TICK(tb1);
   BK_ACC(t.cycB, tb0, tb1);

   int rc = 0, lastSlot = 0;
   bool finished = false;
   while (!finished)
   {
      const double c0 = sdotCur;
      double c1, c2, c3;
      if (!anyGoodIter)
      {
         double lf = lowFact * 2.0;
         c1 = .5 * (c0 + dmax(.999 * sdotMin, (1.0 - lf) * c0));
         lf *= 2.0;
         c2 = .5 * (c1 + dmax(.999 * sdotMin, (1.0 - lf) * c1));
         lf *= 2.0;
         c3 = .5 * (c2 + dmax(.999 * sdotMin, (1.0 - lf) * c2));
      }
      else
      {
         c1 = .5 * (c0 + sdotL); // next midpoint if c0 is violated (sdotH <- c0)
         c2 = .5 * (sdotH + c0); // next midpoint if c0 is feasible (sdotL <- c0)
         c3 = c0;
      }
      const double mine = (t.cslot == 0) ? c0 : (t.cslot == 1) ? c1 : (t.cslot == 2) ? c2 : c3;
      const bool violMine = verify_second_order(t, j, mine); // this slot's sddotL / sddotH stay in t
      const unsigned long long ballot = __ballot(violMine);

      int k = 0;
      for (int consumed = 0; consumed < 4; ++consumed)
      {
         const bool isViol = (ballot >> (t.pbase + 8 * k)) & 1ull;
         lastSlot = k;
         if (isViol)
         {
            sdotH = sdotCur;
            if (!anyGoodIter)
            {
               lowFact *= 2.0;
               sdotL = dmax(.999 * sdotMin, (1.0 - lowFact) * sdotH);
            }
         }
         else
         {
            if (nIter == 0) { finished = true; break; }
            anyGoodIter = true;
            sdotGoodLast = sdotGood;
            sdotGood = sdotCur;
            if (ratio_lt(fabs(sdotGood - sdotGoodLast), sdotGood, sdotErrThresh) || sdotCur < sdotMin)
            {
               t.sdotCur = sdotCur;
               finished = true;
               break;
            }
            sdotL = sdotCur;
         }
         nIter++;
         if (nIter > 100) { rc = -1; finished = true; break; }
         if (sdotCur < 0) { rc = -1; finished = true; break; }
         if (!anyGoodIter)
         {
            if (ratio_lt(sdotH - sdotL, sdotH, 1e-20)) { rc = -1; finished = true; break; }
         }
         sdotCur = .5 * (sdotH + sdotL);
         // was this value evaluated in this pass?
         if (sdotCur == c1) k = 1;
         else if (sdotCur == c2) k = 2;
         else if (sdotCur == c3 && !anyGoodIter) k = 3;
         else break;
      }
   }
   if (rc != 0) return -1;
   // sddotL / sddotH of the last check the sequential loop performed live in the lanes of slot lastSlot
   const int src = t.pbase + 8 * lastSlot;
   t.sddotH = __shfl(t.sddotH, src);
   t.sddotL = __shfl(t.sddotL, src);
   sddot = (t.dir == 1) ? t.sddotH : t.sddotL;
   return 0;
}

template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void accel_pt(Pt<G, FEAT, UNI> &t, int j, double &sddot)
{
   int nIter;
   const int rc = Pt<G, FEAT, UNI>::SPEC ? apply_accel_bisection_spec(t, j, sddot, nIter) : apply_accel_bisection(t, j, sddot, nIter);
   if (rc != 0)
   {
      t.status |= BATOTP_ST_BISECT_FAIL;
      t.nfail++;
   }
}

// fill the constants of a path group
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ void pt_init(Pt<G, FEAT, UNI> &t, const DevProblem &P, const PathInfo &pi, const double *sC,
                                        const double *coef, const double *km, const double (*lim)[8], int j, int dir)
{
   constexpr bool PAR = (FEAT == 3);
   t.sC = sC + pi.koff;
   t.coef = (FEAT < 0) ? nullptr : coef + pi.koff * P.C * 4;
   t.km = (FEAT < 0) ? reinterpret_cast<const double2 *>(km) + pi.koff * P.Cin : nullptr;
   t.n = (int)pi.n;
   t.C = P.C; t.nJ = P.nJ; t.nC = P.nC; t.nIn = P.Cin;
   t.flags = P.flags;
   t.parallel_now = pi.parallel_now;
   t.sresC = pi.sres_c;
   t.vfact = pi.vfact; t.afact = pi.afact;
   t.thrV = P.jnt_thresh * pi.vfact;
   t.thrA = P.jnt_thresh * pi.afact;
   t.quadA = P.quad_thresh * pi.afact;
   t.quadA2 = P.quad_thresh * P.quad_thresh * pi.afact * pi.afact;
   const double sLastKnot = (UNI || pi.uniform) ? pi.sres_c * (double)(t.n - 1) : t.sC[t.n - 1]; // uniform: as k_sites computes it
   t.sdotCap = sLastKnot / pi.integ_res;                       // ba.cpp:1216
   t.sddotMax = 2 * sLastKnot / (pi.integ_res * pi.integ_res); // ba.cpp:1257
   t.cartAccMaxSQ = P.cart_acc_max * P.cart_acc_max;
   t.cartVelMax = P.cart_vel_max;
   t.pmat = &lim[4][0];
#pragma unroll
   for (int q = 0; q < Pt<G, FEAT, UNI>::PER; ++q)
   {
      const int jj = (j + q * G) & 7;
      t.vmax[q] = lim[0][jj]; t.amax[q] = lim[1][jj]; t.tmax[q] = lim[2][jj]; t.tmin[q] = lim[3][jj];
      t.thD[q] = 0; t.thD2[q] = 0; t.a1[q] = 0; t.a2[q] = 0; t.a3[q] = 0; t.a4[q] = 0;
   }
   t.cq0 = 0; t.cq1 = 0; t.cq2 = 0;
   if (PAR)
   {
      for (int k = 0; k < 3; ++k) { t.thp[PAR ? k : 0] = 0; t.cap[PAR ? k : 0] = 0; t.pa1[PAR ? k : 0] = 0; t.pa2[PAR ? k : 0] = 0; t.pa3[PAR ? k : 0] = 0; t.pa4[PAR ? k : 0] = 0; }
      for (int k = 0; k < 9; ++k) t.Am[PAR ? k : 0] = 0;
   }
   t.mvc = nullptr; t.nMvc = 0;
   t.dir = dir;
   t.segC = 0; t.segMVC = 0; t.tauC = 0; t.tauMVC = 0;
   t.sCur = 0; t.sdotCur = 0; t.sddotL = 0; t.sddotH = 0; t.sdotMin = 0;
   t.status = 0; t.nfail = 0; t.sink = 0; t.side = 0; t.cslot = 0; t.pbase = 0;
#ifdef BK_PROFILE_SECTIONS
   t.cycA = 0; t.cycB = 0; t.cycC = 0; t.cycD = 0;
#endif
   t.rowSeg = -1; t.mvcSeg = -1;
   t.mvcS0 = 0; t.mvcS1 = 0; t.mvcD0 = 0; t.mvcD1 = 0;
}

// stage the limit tables of the problem in LDS (one copy per workgroup); row 4/5 = cable anchors
__device__ __forceinline__ void stage_limits(const DevProblem *__restrict__ dP, double (*lim)[8])
{
   if (threadIdx.x < 32)
   {
      const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
      const double *src = (r == 0) ? dP->vmax : (r == 1) ? dP->amax : (r == 2) ? dP->tmax : dP->tmin;
      lim[r][c] = src[c];
   }
   else if (threadIdx.x < 41)
   {
      const int c = threadIdx.x - 32;
      lim[4 + c / 8][c % 8] = dP->pmat[c];
   }
   __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// K3: max admissible sdot and the sddot interval at every knot; one lane per (path, knot).
// Definition: cursor on segment min(i, N-2); sdot starts from the clamp of ba.cpp:1216,
// is cut by the velocity limits of ba.cpp:1219-1229 with THIS knot's derivatives, then by the
// bisection of ba.cpp:1248-1332.
// ---------------------------------------------------------------------------------------------
constexpr int K3_BLOCK = 128;

template <int FEAT>
__global__ void __launch_bounds__(K3_BLOCK) k_pointwise(DevProblem P, const PathInfo *__restrict__ pinfo, int B,
                                                        const DevProblem *__restrict__ dP,
                                                        const double *__restrict__ sC, const double *__restrict__ coef,
                                                        const double *__restrict__ km,
                                                        double *__restrict__ mvc, int64_t total, int useTile, int64_t mvcSlot)
{
   __shared__ double lim[6][8];
   extern __shared__ double tile[]; // K3_BLOCK rows of C*4 doubles, padded by 2 doubles per row
   stage_limits(dP, lim);

   // The coefficient rows of consecutive knots are contiguous in HBM (also across path boundaries):
   // the workgroup copies its K3_BLOCK rows with fully coalesced 32-byte-per-lane loads into LDS,
   // then every lane evaluates its own knot from LDS (row stride padded by 16 B: conflict-free b128 reads).
   const int64_t g0 = (int64_t)blockIdx.x * K3_BLOCK;
   const int rowD = P.C * 4, rowPad = rowD + 2;
   const int rowsHere = (total - g0) < K3_BLOCK ? (int)(total - g0) : K3_BLOCK;
   // (compact splines read their pair rows directly; so do wide rows -- torque problems -- whose tile would leave
   // room for one wavefront per SIMD only: useTile is the host's choice)
   if (FEAT >= 0 && useTile)
   {
      const Coef4 *__restrict__ srcp = reinterpret_cast<const Coef4 *>(coef + g0 * rowD);
      const int chunks = rowsHere * P.C;
      for (int k = threadIdx.x; k < chunks; k += K3_BLOCK)
      {
         const Coef4 v = srcp[k];
         const int r = k / P.C, c = k - r * P.C;
         double *dst = tile + r * rowPad + c * 4;
         dst[0] = v.c0; dst[1] = v.c1; dst[2] = v.c2; dst[3] = v.c3;
      }
   }
   __syncthreads();

   const int64_t g = g0 + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int N = (int)pi.n, i = (int)(g - pi.koff);

   Pt<1, FEAT, false> t;
   pt_init(t, P, pi, sC, coef, km, lim, 0, -1);
   t.segC = (i < N - 1) ? i : N - 2;
   // uniform sites are computed as k_sites computes them (the array is not kept for compact batches)
   t.sCur = pi.uniform ? pi.sres_c * (double)i : t.sC[i];
   // cursor already on its segment: updateCurSeg only computes tau (0 at a knot, 1 at the last knot)
   {
      const double sSeg = pi.uniform ? pi.sres_c * (double)t.segC : t.sC[t.segC];
      const double sNext = pi.uniform ? pi.sres_c * (double)(t.segC + 1) : t.sC[t.segC + 1];
      t.tauC = (t.sCur - sSeg) / (sNext - sSeg);
   }
   if (FEAT < 0) eval_partials_cached(t, 0);
   else
   {
      const int rowLocal = (int)threadIdx.x - (i - t.segC); // the last knot of a path uses the previous row
      if (km != nullptr)
      {
         // all channels as pairs: the row of the knot's segment is formed channel by channel where it is read (emit_segment's formulas)
         eval_partials_src(t, 0, PairRowAt{reinterpret_cast<const double2 *>(km) + (pi.koff + t.segC) * P.C, P.C});
      }
      else
      {
         const double *row = (useTile && rowLocal >= 0) ? (tile + rowLocal * rowPad) : (t.coef + (unsigned)(t.segC * rowD));
         eval_partials_row(t, 0, row);
      }
   }
   double sdot = t.sdotCap;
   sdot_lim(t, 0, sdot);
   t.sdotCur = sdot;
   double sddot = 0;
   int nIter;
   if (apply_accel_bisection<1, FEAT, false, false>(t, 0, sddot, nIter) != 0)
   {
      // no admissible sdot at this knot (ba.cpp:1307-1319): the K3 definition publishes NaN bounds (what the reference's
      // early loop exits last wrote is an artefact of its loop order)
      t.sddotL = __longlong_as_double(0x7ff8000000000000LL);
      t.sddotH = t.sddotL;
   }
   // mvcSlot != 0 (BATOTP_F_MVC_IN_CURVES): the values go to the front of the path's curve slot (mvcSlot doubles per path)
   double *__restrict__ o = mvcSlot ? mvc + (int64_t)lo * mvcSlot : mvc + pi.koff * 3;
   o[i] = t.sdotCur;
   o[N + i] = t.sddotL;
   o[2 * (int64_t)N + i] = t.sddotH;
}

// K3 with a lane group per knot (lane = joint, as in the sweep): one joint per lane keeps the register
// footprint small (the lane-per-knot form carries all eight joints of a knot in one lane and runs at
// low occupancy), the group reductions are DPP butterflies, and only the 8 knots of a wavefront share
// a bisection loop instead of 64.  Same definition, same results as k_pointwise.
constexpr int K3G_BLOCK = 256;

template <int FEAT>
__global__ void __launch_bounds__(K3G_BLOCK) k_pointwise_grp(DevProblem P, const PathInfo *__restrict__ pinfo, int B,
                                                             const DevProblem *__restrict__ dP, const double *__restrict__ sC,
                                                             const double *__restrict__ coef, const double *__restrict__ km,
                                                             double *__restrict__ mvc, int64_t first, int64_t total, int64_t mvcSlot)
{
   __shared__ double lim[6][8];
   stage_limits(dP, lim);
   const int lane = threadIdx.x & 63, j = lane & 7;
   // `first`: a launch covers at most 2^31 lanes (HIP limits grid x block to 32 bits), the host slices the knots
   const int64_t g = first + ((int64_t)blockIdx.x * (K3G_BLOCK / 64) + (threadIdx.x >> 6)) * 8 + (lane >> 3);
   if (g >= total) return; // whole groups leave together
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int N = (int)pi.n, i = (int)(g - pi.koff);

   Pt<8, FEAT, false> t;
   pt_init(t, P, pi, sC, coef, km, lim, j, -1);
   t.segC = (i < N - 1) ? i : N - 2;
   t.sCur = pi.uniform ? pi.sres_c * (double)i : t.sC[i];
   {
      const double sSeg = pi.uniform ? pi.sres_c * (double)t.segC : t.sC[t.segC];
      const double sNext = pi.uniform ? pi.sres_c * (double)(t.segC + 1) : t.sC[t.segC + 1];
      t.tauC = (t.sCur - sSeg) / (sNext - sSeg);
   }
   if (FEAT < 0) eval_partials_cached(t, j);
   else eval_partials_row(t, j, t.coef + (unsigned)(t.segC * t.C * 4));
   double sdot = t.sdotCap;
   sdot_lim(t, j, sdot);
   t.sdotCur = sdot;
   double sddot = 0;
   int nIter;
   if (apply_accel_bisection<8, FEAT, false, false>(t, j, sddot, nIter) != 0)
   {
      t.sddotL = __longlong_as_double(0x7ff8000000000000LL); // see k_pointwise
      t.sddotH = t.sddotL;
   }
   if (j == 0)
   {
      double *__restrict__ o = mvcSlot ? mvc + (int64_t)lo * mvcSlot : mvc + pi.koff * 3;
      o[i] = t.sdotCur;
      o[N + i] = t.sddotL;
      o[2 * (int64_t)N + i] = t.sddotH;
   }
}

// ---------------------------------------------------------------------------------------------
// K4: the sweep.  Wavefront = 64 lanes = up to 64/G paths (a.ppw of them are used: with few paths
// in the batch it is faster to spread them over more wavefronts than to fill every lane).
// ---------------------------------------------------------------------------------------------
struct SweepArgs
{
   DevProblem P;
   const DevProblem *dP;
   const PathInfo *pinfo;
   const double *sC;
   const double *coef;
   const double *km; // compact splines (FEAT == -1): [N][Cin][2] (value, second derivative) per path
   double2 *rev;  // [B][cap]
   double2 *fwd;  // [B][cap]
   batotp_path_result *res;
   int *sink;     // [B] consumer of the prefetch touches
   double *prof;  // diagnostic build (BK_PROFILE_SECTIONS): 4 doubles per path
   int64_t cap;
   int B;
   int dir;
   int ppw;       // paths per wavefront, 1 .. 64/G
   int hold;      // FLAT kernels: a stage is started once hold/8 of the wavefront's live paths wait for one
   int touch;     // software prefetch: bit 0 = spline rows ahead of the cursor, bit 1 = reverse curve ahead of its cursor (forward sweep)
   int ff;        // k_sweep1: certified fast-forward of the bisection (sweep1.hip.h), batotp_hip_set_fast_forward; k_sweep8: bit 0 forward, bit 1 reverse sweep
   int holdc;     // k_sweep8, reverse sweep: the certificate block serves the paths waiting for it once holdc/8 of the wavefront's live paths do
   // Ragged batches: launch slot k of the sweep processes path order[k] -- the paths sorted by knot count, longest
   // first.  The hardware hands workgroups to the SIMDs in launch order as slots free up, so this is longest-processing-time-first
   // scheduling for the kernels with a wavefront per path, and it puts paths of similar length into the same wavefront of the
   // kernels that carry several (a wavefront lasts as long as its longest path).  nullptr: paths in the order given (all equally
   // long, or batotp_hip_set_path_order(ctx, 0)).  Nothing in a path's own arithmetic or memory depends on its slot.
   const int *order;
};

// Butcher tableau of ba.cpp:58-63 (_B[k][j]; stage j+1 uses column j)
#define BK_B00 (1. / 5)
#define BK_B01 (3. / 40)
#define BK_B02 (44. / 45)
#define BK_B03 (19372. / 6561)
#define BK_B04 (9017. / 3168)
#define BK_B05 (35. / 384)
#define BK_B11 (9. / 40)
#define BK_B12 (-56. / 15)
#define BK_B13 (-25360. / 2187)
#define BK_B14 (-355. / 33)
#define BK_B15 (0.)
#define BK_B22 (32. / 9)
#define BK_B23 (64448. / 6561)
#define BK_B24 (46732. / 5247)
#define BK_B25 (500. / 1113)
#define BK_B33 (-212. / 729)
#define BK_B34 (49. / 176)
#define BK_B35 (125. / 192)
#define BK_B44 (-5103. / 18656)
#define BK_B45 (-2187. / 6784)
#define BK_B55 (11. / 84)

// Software prefetch: the walk over the spline rows (and over the reverse curve) is monotone, so
// every stage of the reverse sweep each lane of the group touches one 128-byte line of the next
// kilobyte ahead of (below) the cursor.  The loaded word is consumed one stage later (t.sink), which keeps the load alive without
// ever waiting on it, and by then the line sits in the vector L1 / L2 instead of HBM.
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ int touch_ahead(const Pt<G, FEAT, UNI> &t, int j)
{
   const int linesAhead = 1 + (j & 7);
   const int rowDoubles = t.C * 4;
   int v = 0;
   if (FEAT < 0)
   {
      // compact splines: rows of Cin pairs; same scheme as below on the pair stream
      const int rowD = t.nIn * 2;
      int off = t.segC * rowD + t.dir * linesAhead * 16;
      const int hi = (t.n - 1) * rowD;
      off = off < 0 ? 0 : (off > hi ? hi : off);
      v = reinterpret_cast<const int *>(t.km)[2 * (unsigned)off];
   }
   else
   {
      // spline rows: byte offset of the current row, then +/- (1..8) lines
      const int lastRow = t.n - 1;
      int off = t.segC * rowDoubles + t.dir * linesAhead * 16; // in doubles (16 doubles = 128 B)
      const int hi = lastRow * rowDoubles;
      off = off < 0 ? 0 : (off > hi ? hi : off);
      v = reinterpret_cast<const int *>(t.coef)[2 * (unsigned)off];
   }
   return v;
}

// the same for the reverse curve the forward sweep follows: (s, sdot) pairs, 8 points per 128-byte line, ascending walk
template <int G, int FEAT, bool UNI>
__device__ __forceinline__ int touch_curve_ahead(const Pt<G, FEAT, UNI> &t, int j)
{
   int off = t.segMVC * 2 + (1 + (j & 7)) * 16; // in doubles
   const int hi = (t.nMvc - 1) * 2;
   off = off > hi ? hi : off;
   return reinterpret_cast<const int *>(t.mvc)[2 * (unsigned)off];
}

// A workgroup is 4 wavefronts (one per SIMD of a CU): with one-wavefront workgroups the dispatcher was
// observed to stack several of them on the SIMDs of one CU while other CUs stayed empty.
constexpr int K4_BLOCK = 256;
#ifndef BK_SWEEP_WPE
#define BK_SWEEP_WPE 2
#endif
// waves per SIMD the register allocation of the narrow (FEAT <= 1) sweep kernels is tuned for
// FLAT: the stage loop and the bisection loop are one loop in which every path of the wavefront is either
// waiting for its next stage or inside a constraint check (see the comment at the loop).
template <int G, int FEAT, bool UNI, bool FLAT = false>
__global__ void __launch_bounds__(K4_BLOCK, (G > 1 && G < 8) ? 1 : (FEAT <= 1 && G > 1) ? BK_SWEEP_WPE : 2) k_sweep(SweepArgs a)
{
   __shared__ double lim[6][8];
   __shared__ double rk[7][6]; // FLAT: rk[st][k] = weight of stage value k in stage st (column st-1 of ba.cpp:58-63), 0 for k >= st
   if (FLAT && threadIdx.x < 42)
   {
      const double tab[42] = {0, 0, 0, 0, 0, 0,
                              BK_B00, 0, 0, 0, 0, 0,
                              BK_B01, BK_B11, 0, 0, 0, 0,
                              BK_B02, BK_B12, BK_B22, 0, 0, 0,
                              BK_B03, BK_B13, BK_B23, BK_B33, 0, 0,
                              BK_B04, BK_B14, BK_B24, BK_B34, BK_B44, 0,
                              BK_B05, BK_B15, BK_B25, BK_B35, BK_B45, BK_B55};
      (&rk[0][0])[threadIdx.x] = tab[threadIdx.x];
   }
   stage_limits(a.dP, lim);

   const int lane = threadIdx.x & 63;
   const int wave = blockIdx.x * (K4_BLOCK / 64) + (threadIdx.x >> 6);
   constexpr bool SPLIT = Pt<G, FEAT, UNI>::SPLIT;
   constexpr bool SPEC = Pt<G, FEAT, UNI>::SPEC;
   const int j = (SPLIT || SPEC) ? (lane & 7) : lane % G;   // joint owned by this lane
   const int slot = lane / G;
   const int pslot = wave * a.ppw + slot;
   if (slot >= a.ppw || pslot >= a.B) return; // whole groups leave together; DPP never crosses groups
   const int p = a.order ? a.order[pslot] : pslot;
   const bool writer = (lane % G == 0);
   const int dir = a.dir;
   const PathInfo pi = a.pinfo[p];
   const int n = (int)pi.n;
   const int64_t cap = a.cap;

   Pt<G, FEAT, UNI> t;
   pt_init(t, a.P, pi, a.sC, a.coef, a.km, lim, j, dir);
   t.side = SPLIT ? ((lane >> 3) & 1) : 0;
   t.cslot = SPEC ? ((lane >> 3) & 3) : 0;
   t.pbase = lane & ~(G - 1);

   double2 *out = (dir == 1 ? a.fwd : a.rev) + (int64_t)p * cap; // may alias t.mvc (curves in place): no __restrict__
   batotp_path_result *__restrict__ r = a.res + p;
   if (dir == 1)
   {
      const int64_t nRev = r->n_rev;
      if (nRev < 2)
      {
         // no reverse curve to follow (its sweep ended with an error status): nothing to integrate
         if (writer) { r->n_fwd = 0; r->steps_fwd = 0; r->t_total = 0; r->status_fwd = r->status_rev | BATOTP_ST_CAPACITY; r->n_bisect_fail_fwd = 0; }
         return;
      }
      t.mvc = reinterpret_cast<const double *>(a.rev + (int64_t)p * cap + (cap - nRev));
      t.nMvc = (int)nRev;
   }
   // BATOTP_F_CURVES_IN_PLACE (forward sweep, a.fwd == a.rev): point i of the forward curve goes to slot i, reverse point m sits in
   // slot cap - nRev + m.  The forward curve has at least as many points up to any s as the reverse curve, so slot i has been
   // left behind by the reverse-curve cursor -- checked every step with 64 points of margin (the cursor's back-steps, the
   // prefetch and the windows of k_sweep1 stay within 32); a path that would run into live reverse points ends as if out of capacity.
   const int64_t revStart = (dir == 1 && a.fwd == a.rev) ? cap - (int64_t)t.nMvc : ((int64_t)1 << 62);
#define BK_CURVE_FULL(i) ((i) >= cap || (i) + 64 >= revStart + (int64_t)t.segMVC)

   const double absh = pi.integ_res;
   const double h = dir * absh;
   const int64_t maxIntegSteps = (int64_t)floor(a.P.max_integ_time / pi.integ_res) + 1;
   const double sEnd = UNI ? pi.sres_c * (double)(n - 1) : t.sC[n - 1];
   double sLast;
   double s0v, s6v = 0;                             // sArr[0], sArr[6] (the other stage positions are not reused)
   double v0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;   // sdotArr
   double w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0; // sddotArr

   if (dir == 1) { t.segC = 0; t.tauC = 0; s0v = 0; t.segMVC = 0; t.tauMVC = 0; sLast = sEnd; }
   else { t.segC = n - 2; t.tauC = 1; s0v = sEnd; t.segMVC = n - 2; t.tauMVC = 1; sLast = 0; }
   t.sCur = s0v;
   t.sdotCur = 0;

   // bootstrap, ba.cpp:1024-1041 (two passes of [bisection, velocity limit]; kept as a loop so
   // that the point evaluation is instantiated once here and once in the stage loop)
#pragma unroll 1
   for (int pass = 0; pass < 2; ++pass)
   {
      accel_pt(t, j, w0);
      if (pass == 0) { v0 = .1 * h * w0; t.sdotMin = v0; }
      else v0 = t.sdotCur;
      sdot_lim(t, j, v0);
      if (pass == 0) { t.sdotMin = v0; t.sdotCur = v0; }
   }

   // point 0
   double sPrev = s0v, sdPrev = v0; // last two published points, for the end snap
   double sCurPt = s0v, sdCurPt = v0;
   if (writer) out[dir == 1 ? 0 : cap - 1] = make_double2(s0v, v0);

   // ba.cpp:1050-1051: dsMin uses sArr.back() == 0, hence every dsMinV[j]/absh is +0
   const double floorV = 0.0 / absh;

#ifdef BK_PROFILE_SECTIONS
   const unsigned long long tstart = __builtin_readcyclecounter();
#endif
   int64_t nPts = 0, i = 1;
   unsigned endStatus = 0;
   int pf = 0, pf2 = 0;
   if constexpr (FLAT)
   {
      // One loop instead of {stages {bisection iterations}}.  In the nested form a wavefront stays in the bisection
      // loop of a stage as long as ANY of its paths does (24 % of the path-stages of a reverse sweep bisect, for
      // 5-15 iterations; with 8 paths per wavefront nearly every stage has such a path) while the paths whose first
      // check passed idle.  Here a path is either waiting for its next stage or inside a check, every pass of the
      // loop runs one check for all paths that are inside one, and the stage prologue (tableau combination, velocity
      // limit, spline evaluation) runs when a.hold/8 of the live paths are waiting for it, so that its cost is shared.
      // Paths of a wavefront drift apart in stage and step; nothing in a path's own arithmetic or order changes.
      int st = (dir == 1) ? 0 : 1;
      // state of this lane's path: one integer in a vector register on purpose (separate booleans become lane masks in
      // scalar registers that are merged under ever-changing exec masks across this loop)
      constexpr int PH_FIRST = 0, PH_ENDED = 1, PH_CHECK = 2, PH_DEAD = 3; // waiting for its first stage / a stage has ended / inside a check / finished
      int phase = PH_FIRST;
      if (BK_CURVE_FULL(i)) { endStatus = BATOTP_ST_CAPACITY; phase = PH_DEAD; }
      double sN = 0, wN = 0;
      double lowFact = .01, sdotGood = 0, sdotL = 0, sdotH = 0, sdotTry = 0;
      int nGood = 0; // feasible points seen by the current bisection (anyGoodIter of ba.cpp:1254 == nGood > 0)
      int nIter = 0;
      for (;;)
      {
         const unsigned long long mAlive = __ballot(phase != PH_DEAD);
         if (mAlive == 0) break;
         const unsigned long long mWait = __ballot(phase < PH_CHECK);
         const bool startNow = (mWait == mAlive) || (__popcll(mWait) * 8 >= __popcll(mAlive) * a.hold);
         if (startNow && phase < PH_CHECK)
         {
            if (phase == PH_ENDED)
            {
               // the stage that just ended: keep its values (done here, once for all paths that wait, rather than
               // path by path at the pass in which each one's check ended)
               phase = PH_FIRST;
               const double vN = t.sdotCur;
               v1 = (st == 1) ? vN : v1; w1 = (st == 1) ? wN : w1;
               v2 = (st == 2) ? vN : v2; w2 = (st == 2) ? wN : w2;
               v3 = (st == 3) ? vN : v3; w3 = (st == 3) ? wN : w3;
               v4 = (st == 4) ? vN : v4; w4 = (st == 4) ? wN : w4;
               v5 = (st == 5) ? vN : v5; w5 = (st == 5) ? wN : w5;
               if (a.touch & 1)
               {
                  t.sink += pf;
                  pf = touch_ahead(t, j);
               }
               if (st < 6) ++st;
               else
               {
                  // FSAL shift and publish, ba.cpp:1096-1100 (stage 6: position sN, values vN, wN)
                  s0v = sN; v0 = vN; w0 = wN; w6 = wN;
                  sPrev = sCurPt; sdPrev = sdCurPt;
                  sCurPt = s0v; sdCurPt = v0;
                  if (writer) out[dir == 1 ? i : cap - 1 - i] = make_double2(s0v, v0);
                  st = (dir == 1) ? 0 : 1;
                  if (t.sCur * dir > sLast) { nPts = i + 1; phase = PH_DEAD; } // ba.cpp:1109-1115
                  else if (i > maxIntegSteps) { endStatus = BATOTP_ST_MAX_INTEG_TIME; phase = PH_DEAD; } // ba.cpp:1117-1122
                  else if (++i, BK_CURVE_FULL(i)) { endStatus = BATOTP_ST_CAPACITY; phase = PH_DEAD; }
               }
            }
            if (phase != PH_DEAD)
            {
               if (st == 0)
               {
                  // forward predictor (ba.cpp:1055-1065): only the move of the reverse-curve cursor survives
                  t.sCur = s0v + h * v0;
                  mvc_walk(t);
                  st = 1;
               }
               const double *bc = rk[st];
               double sdotT = 0, sddotT = 0;
               // stage st adds the terms k < st only (a stale stage value may be infinite: 0 * inf must not enter the sum)
               sdotT += bc[0] * v0; sddotT += bc[0] * w0;
               { const double a1 = sdotT + bc[1] * v1, b1 = sddotT + bc[1] * w1; sdotT = (st > 1) ? a1 : sdotT; sddotT = (st > 1) ? b1 : sddotT; }
               { const double a2 = sdotT + bc[2] * v2, b2 = sddotT + bc[2] * w2; sdotT = (st > 2) ? a2 : sdotT; sddotT = (st > 2) ? b2 : sddotT; }
               { const double a3 = sdotT + bc[3] * v3, b3 = sddotT + bc[3] * w3; sdotT = (st > 3) ? a3 : sdotT; sddotT = (st > 3) ? b3 : sddotT; }
               { const double a4 = sdotT + bc[4] * v4, b4 = sddotT + bc[4] * w4; sdotT = (st > 4) ? a4 : sdotT; sddotT = (st > 4) ? b4 : sddotT; }
               { const double a5 = sdotT + bc[5] * v5, b5 = sddotT + bc[5] * w5; sdotT = (st > 5) ? a5 : sdotT; sddotT = (st > 5) ? b5 : sddotT; }
               sN = s0v + h * sdotT;
               double vN = v0 + h * sddotT;
               vN = dmax(vN, floorV); // ba.cpp:1085
               t.sCur = sN;
               sdot_lim(t, j, vN);
               t.sdotCur = vN;
               // sddotArr[st] keeps its previous value when the bisection fails (ba.cpp:1091 ignores the code)
               wN = (st == 1) ? w1 : (st == 2) ? w2 : (st == 3) ? w3 : (st == 4) ? w4 : (st == 5) ? w5 : w6;
               // applyAccelConstraintsBisectionPt, ba.cpp:1250-1265
               lowFact = .01; sdotGood = 0; nGood = 0; sdotL = 0; sdotH = vN; sdotTry = vN; nIter = 0;
               eval_partials(t, j);
               phase = PH_CHECK;
            }
         }
         if (phase == PH_CHECK)
         {
            // one pass of the loop of ba.cpp:1267-1321: the statements of apply_accel_bisection above, written as selects
            // (no divergent branches around the few operations of the update)
            const bool isViol = verify_second_order(t, j, sdotTry);
            const bool first = (nIter == 0);
            const bool good = !isViol && !first;                 // a feasible point after at least one violated one
            const bool shrink = isViol && nGood == 0;            // ba.cpp:1281-1285: no feasible point known yet
            const double lowFact2 = lowFact * 2.0;
            const double sdotLShrunk = dmax(.999 * 0.0, (1.0 - lowFact2) * sdotTry);
            // ba.cpp:1294-1303: two successive feasible points closer than 1e-3 (relative), or a negative one
            const bool conv = good && (ratio_lt(fabs(sdotTry - sdotGood), sdotTry, .001) || sdotTry < 0.0);
            const bool fin = (!isViol && first) || conv;
            lowFact = shrink ? lowFact2 : lowFact;
            sdotH = isViol ? sdotTry : sdotH;
            sdotL = shrink ? sdotLShrunk : ((good && !conv) ? sdotTry : sdotL);
            sdotGood = good ? sdotTry : sdotGood;
            nGood += good ? 1 : 0;
            t.sdotCur = conv ? sdotTry : t.sdotCur;
            // ba.cpp:1305-1320
            const bool collapsed = (nGood == 0) && ratio_lt(sdotH - sdotL, sdotH, 1e-20);
            const bool failed = !fin && (nIter + 1 > 100 || sdotTry < 0.0 || collapsed);
            nIter += fin ? 0 : 1;
            sdotTry = (fin || failed) ? sdotTry : .5 * (sdotH + sdotL);
            wN = fin ? ((dir == 1) ? t.sddotH : t.sddotL) : wN;
            t.status |= failed ? (unsigned)BATOTP_ST_BISECT_FAIL : 0u;
            t.nfail += failed ? 1 : 0;
            phase = (fin || failed) ? PH_ENDED : phase;
         }
      }
   }
   else
   {
      // the nested form: stages, and inside every stage the bisection loop of accel_pt
      bool done = false;
      while (!done)
      {
         if (BK_CURVE_FULL(i)) { endStatus = BATOTP_ST_CAPACITY; break; }
         const double sStart = t.sCur;
         // st == 0: Euler predictor (ba.cpp:1055-1065).  Its sdot is overwritten by stage 6; all that
         // survives is the move of the reverse-curve cursor inside evalsdot (forward sweep only).
         // st == 1..6: the six stages of ba.cpp:1068-1094.  A real loop (not unrolled) keeps the
         // kernel inside the instruction cache.
#pragma unroll 1
         for (int st = (dir == 1 ? 0 : 1); st < 7; ++st)
         {
            double sN, vN, sdotT = 0, sddotT = 0;
            switch (st)
            {
            case 0: break;
            case 1: sdotT += BK_B00 * v0; sddotT += BK_B00 * w0; break;
            case 2: sdotT += BK_B01 * v0; sdotT += BK_B11 * v1; sddotT += BK_B01 * w0; sddotT += BK_B11 * w1; break;
            case 3:
               sdotT += BK_B02 * v0; sdotT += BK_B12 * v1; sdotT += BK_B22 * v2;
               sddotT += BK_B02 * w0; sddotT += BK_B12 * w1; sddotT += BK_B22 * w2;
               break;
            case 4:
               sdotT += BK_B03 * v0; sdotT += BK_B13 * v1; sdotT += BK_B23 * v2; sdotT += BK_B33 * v3;
               sddotT += BK_B03 * w0; sddotT += BK_B13 * w1; sddotT += BK_B23 * w2; sddotT += BK_B33 * w3;
               break;
            case 5:
               sdotT += BK_B04 * v0; sdotT += BK_B14 * v1; sdotT += BK_B24 * v2; sdotT += BK_B34 * v3; sdotT += BK_B44 * v4;
               sddotT += BK_B04 * w0; sddotT += BK_B14 * w1; sddotT += BK_B24 * w2; sddotT += BK_B34 * w3; sddotT += BK_B44 * w4;
               break;
            default:
               sdotT += BK_B05 * v0; sdotT += BK_B15 * v1; sdotT += BK_B25 * v2; sdotT += BK_B35 * v3; sdotT += BK_B45 * v4; sdotT += BK_B55 * v5;
               sddotT += BK_B05 * w0; sddotT += BK_B15 * w1; sddotT += BK_B25 * w2; sddotT += BK_B35 * w3; sddotT += BK_B45 * w4; sddotT += BK_B55 * w5;
               break;
            }
            if (st == 0)
            {
               // forward predictor: evalsdot's cursor walk at s0 + h*sdot0, nothing else is kept
               t.sCur = s0v + h * v0;
               mvc_walk(t);
               t.sCur = sStart;
               continue;
            }
            sN = s0v + h * sdotT;
            vN = v0 + h * sddotT;
            vN = dmax(vN, floorV); // ba.cpp:1085
            t.sCur = sN;
            BK_TICK(ta0);
            sdot_lim(t, j, vN);
            BK_TICK(ta1);
            BK_ACC(t.cycA, ta0, ta1);
            t.sdotCur = vN;
            // sddotArr[st] keeps its previous value when the bisection fails (ba.cpp:1091 ignores the code)
            double wN = (st == 1) ? w1 : (st == 2) ? w2 : (st == 3) ? w3 : (st == 4) ? w4 : (st == 5) ? w5 : w6;
            BK_TICK(tc0);
            accel_pt(t, j, wN);
            BK_TICK(tc1);
            BK_ACC(t.cycC, tc0, tc1);
            vN = t.sdotCur;
            switch (st)
            {
            case 1: v1 = vN; w1 = wN; break;
            case 2: v2 = vN; w2 = wN; break;
            case 3: v3 = vN; w3 = wN; break;
            case 4: v4 = vN; w4 = wN; break;
            case 5: v5 = vN; w5 = wN; break;
            default: s6v = sN; v6 = vN; w6 = wN; break;
            }
            // reverse sweep only (descending addresses; measured: -17 % there, +5 % on the forward sweep,
            // whose ascending walk finds the next lines already on their way): consume last stage's touch,
            // issue the next one
            if (a.touch & 1)
            {
               t.sink += pf;
               pf = touch_ahead(t, j);
            }
            if (dir == 1 && (a.touch & 2))
            {
               t.sink += pf2;
               pf2 = touch_curve_ahead(t, j);
            }
         }

         // FSAL shift and publish, ba.cpp:1096-1100
         s0v = s6v; v0 = v6; w0 = w6;
         sPrev = sCurPt; sdPrev = sdCurPt;
         sCurPt = s0v; sdCurPt = v0;
         if (writer) out[dir == 1 ? i : cap - 1 - i] = make_double2(s0v, v0);

         if (t.sCur * dir > sLast) { nPts = i + 1; done = true; } // ba.cpp:1109-1115
         else if (i > maxIntegSteps) { endStatus = BATOTP_ST_MAX_INTEG_TIME; break; } // ba.cpp:1117-1122
         else ++i;
      }
   }
   if (writer) a.sink[p] = t.sink + pf + pf2;
#ifdef BK_PROFILE_SECTIONS
   if (writer) { const unsigned long long tend = __builtin_readcyclecounter(); a.prof[4 * p + 0] = (double)t.cycA; a.prof[4 * p + 1] = (double)t.cycB; a.prof[4 * p + 2] = (double)(t.cycC - t.cycB); a.prof[4 * p + 3] = (double)(tend - tstart); }
#endif

   unsigned status = t.status | endStatus;
   if (endStatus != 0)
   {
      if (writer)
      {
         if (dir == 1) { r->n_fwd = 0; r->steps_fwd = i; r->t_total = 0; r->status_fwd = status; r->n_bisect_fail_fwd = t.nfail; }
         else { r->n_rev = 0; r->steps_rev = i; r->t_rev = 0; r->status_rev = (r->status_rev & BATOTP_ST_SEG_ERROR) | status; r->n_bisect_fail_rev = t.nfail; }
      }
      return;
   }

   // end snap onto sLast, ba.cpp:1132-1134; forward: last sdot <- reverse curve's last sdot, ba.cpp:1140
   {
      const double sRat = (sLast - sPrev) / (sCurPt - sPrev);
      sdCurPt = sdPrev + sRat * (sdCurPt - sdPrev);
      sCurPt = sLast;
      if (dir == 1) sdCurPt = t.mvc[(t.nMvc - 1) * 2 + 1];
      if (writer) out[dir == 1 ? nPts - 1 : cap - nPts] = make_double2(sCurPt, sdCurPt);
   }
   const double tElapsed = absh * (double)(nPts - 1); // ba.cpp:1112
   int64_t nOut = nPts;

   if (nPts < 4 && writer)
   {
      // ba.cpp:1171-1184: re-interpolate linearly in time to four points
      double ps[3], pd[3], tIn[3];
      for (int k = 0; k < (int)nPts; ++k)
      {
         // ascending-in-time order == integration order
         const double2 q = (k == (int)nPts - 1) ? make_double2(sCurPt, sdCurPt) : out[dir == 1 ? k : cap - 1 - k];
         ps[k] = q.x; pd[k] = q.y;
         tIn[k] = absh * (double)k;
      }
      if (dir != 1)
      {
         // the reference reversed the arrays before this step (ba.cpp:1145-1146)
         for (int k = 0; k < (int)nPts / 2; ++k)
         {
            swap_d(ps[k], ps[nPts - 1 - k]);
            swap_d(pd[k], pd[nPts - 1 - k]);
         }
      }
      const double tResNew = tIn[nPts - 1] / 3.;
      double ns[4], nd[4];
      int cur = 0;
      for (int k = 0; k < 4; ++k)
      {
         const double tn = tResNew * (double)k;
         while (!(tn < tIn[cur + 1] || cur == (int)nPts - 2)) ++cur;
         const double tau = (tn - tIn[cur]) / (tIn[cur + 1] - tIn[cur]);
         ns[k] = ps[cur] + (ps[cur + 1] - ps[cur]) * tau;
         nd[k] = pd[cur] + (pd[cur + 1] - pd[cur]) * tau;
      }
      for (int k = 0; k < 4; ++k) out[dir == 1 ? k : cap - 4 + k] = make_double2(ns[k], nd[k]);
   }
   if (nPts < 4) { status |= BATOTP_ST_SHORT; nOut = 4; }

   if (writer)
   {
      if (dir == 1)
      {
         r->n_fwd = nOut; r->steps_fwd = nPts - 1; r->t_total = tElapsed; r->status_fwd = status; r->n_bisect_fail_fwd = t.nfail;
      }
      else
      {
         r->n_rev = nOut; r->steps_rev = nPts - 1; r->t_rev = tElapsed;
         r->status_rev = (r->status_rev & BATOTP_ST_SEG_ERROR) | status; r->n_bisect_fail_rev = t.nfail;
      }
   }
}
#undef BK_CURVE_FULL

// ---------------------------------------------------------------------------------------------
// marshalling helpers: one channel between [4][N] (C-ABI) and the interleaved device layout
// ---------------------------------------------------------------------------------------------
__global__ void k_coef_gather(const double *__restrict__ coefPath, int C, int dc, int64_t N, double *__restrict__ out)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= N) return;
   const Coef4 k = *reinterpret_cast<const Coef4 *>(coefPath + (i * C + dc) * 4);
   out[i] = k.c0; out[N + i] = k.c1; out[2 * N + i] = k.c2; out[3 * N + i] = k.c3;
}
// compact splines: the coefficient rows of one channel formed from (value, second derivative); the
// row of the last knot is zero as in the coefficient layout (spline.cpp:203-209 never writes it)
__global__ void k_coef_from_sol(const double *__restrict__ kmPath, int C, int dc, int64_t N, double *__restrict__ out)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= N) return;
   Coef4 k;
   k.c0 = 0; k.c1 = 0; k.c2 = 0; k.c3 = 0;
   const double *a = kmPath + (i * C + dc) * 2, *b = a + 2 * C;
   if (i < N - 1) k = coeffs_from_sol(a[1], b[1], a[0], b[0]);
   out[i] = k.c0; out[N + i] = k.c1; out[2 * N + i] = k.c2; out[3 * N + i] = k.c3;
}
// knot values of paths [path0, path0+n) from the C-ABI layout (path after path, [srcC][N] each, of which the first C rows are taken)
// into the value slots of the pair array; one lane per knot
__global__ void k_pairs_from_rows(const PathInfo *__restrict__ pinfo, int path0, int nPaths, int C, int srcC, int kmC, const double *__restrict__ rows,
                                  double *__restrict__ km, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int64_t base = pinfo[path0].koff;
   int lo = path0, hi = path0 + nPaths - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff - base <= g) lo = mid; else hi = mid - 1;
   }
   const int64_t off = pinfo[lo].koff - base, N = pinfo[lo].n, i = g - off;
   const double *src = rows + off * srcC + i;
   double *dst = km + ((pinfo[lo].koff + i) * kmC) * 2; // kmC channels per knot in the pair array (C of them are input channels)
   for (int c = 0; c < C; ++c) dst[2 * c] = src[(int64_t)c * N];
}
// the same into the row layout ([C][N] per path): the first C of srcC rows of every path
__global__ void k_rows_take(const PathInfo *__restrict__ pinfo, int path0, int nPaths, int C, int srcC, const double *__restrict__ rows,
                            double *__restrict__ y, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int64_t base = pinfo[path0].koff;
   int lo = path0, hi = path0 + nPaths - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff - base <= g) lo = mid; else hi = mid - 1;
   }
   const int64_t off = pinfo[lo].koff - base, N = pinfo[lo].n, i = g - off;
   const double *src = rows + off * srcC + i;
   double *dst = y + pinfo[lo].koff * C + i;
   for (int c = 0; c < C; ++c) dst[(int64_t)c * N] = src[(int64_t)c * N];
}
__global__ void k_coef_scatter(double *__restrict__ coefPath, int C, int dc, int64_t N, const double *__restrict__ in)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= N) return;
   Coef4 k;
   k.c0 = in[i]; k.c1 = in[N + i]; k.c2 = in[2 * N + i]; k.c3 = in[3 * N + i];
   *reinterpret_cast<Coef4 *>(coefPath + (i * C + dc) * 4) = k;
}
__global__ void k_curve_pack(double2 *__restrict__ dst, const double *__restrict__ s, const double *__restrict__ sd, int64_t n)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) dst[i] = make_double2(s[i], sd[i]);
}
__global__ void k_curve_unpack(const double2 *__restrict__ src, double *__restrict__ s, double *__restrict__ sd, int64_t n)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { const double2 q = src[i]; s[i] = q.x; sd[i] = q.y; }
}

// curves of a range of paths packed path after path: off[k] = first output point of path k (off[n] = total),
// start[k] = first point of path k's curve inside its slot of `cap` points
__global__ void k_curves_pack(const double2 *__restrict__ src, int64_t cap, int path0, int n, const int64_t *__restrict__ off,
                              const int64_t *__restrict__ start, double2 *__restrict__ dst, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = n - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (off[mid] <= g) lo = mid; else hi = mid - 1;
   }
   dst[g] = src[(int64_t)(path0 + lo) * cap + start[lo] + (g - off[lo])];
}

// fp64 known-answer test: q = a/b, r = sqrt(a), p = a*b + q (must NOT be contracted)
__global__ void k_kat(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ q,
                      double *__restrict__ r, double *__restrict__ p)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   q[i] = a[i] / b[i];
   r[i] = sqrt(a[i]);
   const double prod = a[i] * b[i];
   p[i] = prod + q[i];
}

// known-answer test of the reciprocal-based division by 6 the compact spline form uses
__global__ void k_kat_div6(int64_t n, const double *__restrict__ a, double *__restrict__ q)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   q[i] = div6(a[i]);
}

} // namespace bk
