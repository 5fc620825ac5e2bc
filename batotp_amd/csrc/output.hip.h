// output.hip.h -- GPU port of the output stage behind the hot path: BA::interpOutputData
// (ba.cpp:1661-1931) for JOINT paths of a robot without kinematic model and without torque constraints.
// Included by batotp_hip.hip.  Arithmetic contract as everywhere: fp64, no contraction, the reference's
// operation order; the oracle's bo_output, pinned by the reference binary's traj_out.dat, is the checker
// (tests/test_gpu_output.py).
//
// Everything but the two spline builds is independent per output point (one lane per point); the spline
// builds reuse K1's Thomas code (k_spline_series, one lane per series).
#pragma once
#include "kernels.hip.h"

namespace bk
{

struct OutPath
{
   int64_t off1;    // first point of this path in the stage-1 arrays (sOut, seg, th1)
   int64_t off2;    // ... in the down-sampled arrays (th2, sol2)
   int64_t offF;    // ... in the final array
   int64_t offS;    // first element of this path's s(t) second derivatives
   int32_t p;       // path of the batch
   int32_t nFwd;    // points of the forward curve
   int32_t n1;      // nOut of ba.cpp:1683
   int32_t n2;      // points after smoothing + down-sampling (= n1 without smoothing)
   int32_t nF;      // final points
   int32_t pad;
   double tStep;    // time step of the forward curve (tMVC[i] = tStep*i)
};

struct OutParams
{
   int nJ;          // joint rows
   int R;           // rows per output point: joints (+ 3 Cartesian + 3 torque rows for the cable robot)
   int window;      // (int)_outSmoothFact when smoothing, else 0
   int reinterp;
   int compact;     // the batch keeps (value, second derivative) pairs instead of coefficient rows
   int C, Cin;
   double outRes;
   double vfactT, afactT; // cable robot: 1/tfact and its square, tfact = outRes/smoothFact (ba.cpp:1754)
   double pmat[9];
};

__device__ __forceinline__ int out_find_path(const OutPath *__restrict__ paths, int K, int64_t g, int which)
{
   int lo = 0, hi = K - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      const int64_t o = which == 1 ? paths[mid].off1 : (which == 2 ? paths[mid].off2 : paths[mid].offF);
      if (o <= g) lo = mid; else hi = mid - 1;
   }
   return lo;
}

// Spline::findInterpSegs (spline.cpp:56-99) for ONE site over the uniform sites a*k, k = 0..n-1: the segment the
// reference's cursor stops on when it starts at 0 -- the first k with x < a*(k+1), clipped to n-2.  The sites are
// formed exactly as the reference's arrays are (a*(double)k); the estimate from the division is only a starting point.
__device__ __forceinline__ int seg_uniform(double x, double a, int n)
{
   if (!(x == x)) return n - 2; // NaN compares false with everything: the cursor runs to the end
   double q = floor(x / a);
   int k = q < 0 ? 0 : (q > (double)(n - 2) ? n - 2 : (int)q);
   while (k > 0 && x < a * (double)k) --k;
   while (k < n - 2 && !(x < a * (double)(k + 1))) ++k;
   return k;
}
// the same over an ascending array of sites
__device__ __forceinline__ int seg_array(double x, const double *__restrict__ s, int n)
{
   if (!(x == x)) return n - 2;
   int a = 0, b = n - 2;
   while (a < b)
   {
      const int m = (a + b) >> 1;
      if (x < s[m + 1]) b = m; else a = m + 1;
   }
   return a;
}

// s at the output times (ba.cpp:1683-1707) and the segment of the path spline each s falls into (before the
// running maximum that makes it the reference's monotone cursor)
__global__ void k_out_s(const OutPath *__restrict__ paths, int K, const double2 *__restrict__ fwd, int64_t cap, const double *__restrict__ solS,
                        const PathInfo *__restrict__ pinfo, const double *__restrict__ sC, double *__restrict__ sOut, int *__restrict__ segK,
                        int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), nOut = op.n1;
   if (i >= nOut) return;
   const double tLast = op.tStep * (double)(op.nFwd - 1);
   // output times: uniform, with one extra site a third of a step from each end (ba.cpp:1691-1699)
   const double lastBase = (double)(nOut - 3);
   double base = (double)(i - 1);
   if (i == 0) base = 0;
   else if (i == 1) base = 1.0 / 3.0;
   if (i == nOut - 1) base = lastBase;
   else if (i == nOut - 2) base = lastBase - 1.0 / 3.0;
   const double tOut = (tLast / lastBase) * base;
   const int seg = seg_uniform(tOut, op.tStep, op.nFwd);
   const double t0 = op.tStep * (double)seg, t1 = op.tStep * (double)(seg + 1);
   const double tau = (tOut - t0) / (t1 - t0);
   const double2 *__restrict__ cv = fwd + (int64_t)op.p * cap;
   const double *__restrict__ m = solS + op.offS;
   const Coef4 k = coeffs_from_sol(m[seg], m[seg + 1], cv[seg].x, cv[seg + 1].x);
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   const double s = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
   sOut[g] = s;
   const PathInfo pi = pinfo[op.p];
   segK[g] = pi.uniform ? seg_uniform(s, pi.sres_c, (int)pi.n) : seg_array(s, sC + pi.koff, (int)pi.n);
}

// findInterpSegs' cursor never moves back: segment of site i = max over the sites up to i.  One lane per path
// (integer running maximum, loads in batches).
__global__ void k_out_segmax(const OutPath *__restrict__ paths, int K, int *__restrict__ segK)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= K) return;
   int *__restrict__ s = segK + paths[k].off1;
   const int n = paths[k].n1;
   constexpr int CH = 16;
   int run = 0, i = 0;
   for (; i + CH <= n; i += CH)
   {
      int v[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) v[q] = s[i + q];
      bool changed = false;
#pragma unroll
      for (int q = 0; q < CH; ++q)
      {
         if (v[q] < run) { v[q] = run; changed = true; }
         run = v[q];
      }
      if (changed)
      {
#pragma unroll
         for (int q = 0; q < CH; ++q) s[i + q] = v[q];
      }
   }
   for (; i < n; ++i)
   {
      if (s[i] < run) s[i] = run;
      run = s[i];
   }
}

// path values at the output sites (ba.cpp:1709-1742): channels [c0, c0+cn) of the path splines go to rows
// [r0, r0+cn) of th1[R][n1] (joint rows of a JOINT path; Cartesian rows of a CART path)
__global__ void k_out_eval(OutParams P, const OutPath *__restrict__ paths, int K, const PathInfo *__restrict__ pinfo,
                           const double *__restrict__ sC, const double *__restrict__ coef, const double *__restrict__ km,
                           const double *__restrict__ sOut, const int *__restrict__ segK, double *__restrict__ th1, int c0, int cn, int r0,
                           int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   const PathInfo pi = pinfo[op.p];
   const int seg = segK[g];
   const double s = sOut[g];
   double s0, s1;
   if (pi.uniform) { s0 = pi.sres_c * (double)seg; s1 = pi.sres_c * (double)(seg + 1); }
   else { s0 = sC[pi.koff + seg]; s1 = sC[pi.koff + seg + 1]; }
   const double tau = (s - s0) / (s1 - s0);
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   double *__restrict__ o = th1 + op.off1 * P.R + (int64_t)r0 * n1 + i;
   for (int cc = 0; cc < cn; ++cc)
   {
      const int c = c0 + cc;
      Coef4 k;
      if (P.compact)
      {
         const double *a = km + ((pi.koff + seg) * P.Cin + c) * 2, *b = a + 2 * P.Cin;
         k = coeffs_from_sol(a[1], b[1], a[0], b[0]);
      }
      else k = *reinterpret_cast<const Coef4 *>(coef + ((pi.koff + seg) * P.C + c) * 4);
      o[(int64_t)cc * n1] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
   }
}

// CART path of the cable robot: cable lengths from the platform position (Robot::invKinCSPR3DOF, robot.cpp:243-278):
// rows 0..2 from rows 3..5
__global__ void k_out_invkin(OutParams P, const OutPath *__restrict__ paths, int K, double *__restrict__ th1, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   double *__restrict__ x = th1 + op.off1 * P.R + i;
   const double px = x[(int64_t)3 * n1], py = x[(int64_t)4 * n1], pz = x[(int64_t)5 * n1];
#pragma unroll
   for (int k = 0; k < 3; ++k)
   {
      const double dx = px - P.pmat[0 * 3 + k], dy = py - P.pmat[1 * 3 + k], dz = pz - P.pmat[2 * 3 + k];
      double sq = 0.0;
      sq += dx * dx;
      sq += dy * dy;
      sq += dz * dz;
      x[(int64_t)k * n1] = sqrt(sq);
   }
}

// torque recomputation of the cable robot (ba.cpp:1744-1790): natural splines through the output samples themselves
// (second derivatives in sol1), every site evaluated at the END of the previous segment (site 0: start of segment 0) --
// the VALUES of the joint and Cartesian rows are replaced by that evaluation too -- then the cable tensions from
// Robot::dynCSPR3DOF (a2 = -cart'', a3 = 0, a4 = (0,0,g)), Robot::setA and the 3x3 LU solve.  src -> dst (all 9 rows).
__global__ void k_out_trq(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ src, const double *__restrict__ sol1,
                          double *__restrict__ dst, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   const int seg = i == 0 ? 0 : i - 1;
   const double tau = i == 0 ? 0.0 : 1.0;
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   double val[6], d2c[3];
#pragma unroll
   for (int r = 0; r < 6; ++r)
   {
      const int64_t at = op.off1 * P.R + (int64_t)r * n1 + seg;
      const Coef4 k = coeffs_from_sol(sol1[at], sol1[at + 1], src[at], src[at + 1]);
      val[r] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
      if (r >= 3) d2c[r - 3] = (6 * k.c3 * tau + 2 * k.c2) * P.afactT;
   }
   double b[3], A[9], xs[3];
#pragma unroll
   for (int j = 0; j < 3; ++j)
   {
      const double a2 = -d2c[j], a3 = 0.0, a4 = (j == 2) ? 9.81 : 0.0;
      b[j] = a2 + a3 + a4;
   }
   cspr_setA(P.pmat, val, val + 3, A);
   lu3_solve(A, b, xs);
   double *__restrict__ o = dst + op.off1 * P.R + i;
#pragma unroll
   for (int r = 0; r < 6; ++r) o[(int64_t)r * n1] = val[r];
#pragma unroll
   for (int j = 0; j < 3; ++j) o[(int64_t)(6 + j) * n1] = xs[j];
}

// (smooth_at: resample.hip.h)

// moving average + linear down-sampling by the smoothing factor (ba.cpp:1838-1871): th2[R][n2] per path
__global__ void k_out_down(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ th1, double *__restrict__ th2,
                           int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 2)];
   const int i = (int)(g - op.off2), nIn = op.n1, nDown = op.n2;
   if (i >= nDown) return;
   int w = P.window < nIn ? P.window : nIn;
   const int half = w / 2 + w % 2 - 1;
   w = 2 * half + 1;
   const double site = ((double)(nIn - 1) / (double)(nDown - 1)) * (double)i;
   const int cur = seg_uniform(site, 1.0, nIn);
   const double width = (double)(cur + 1) - (double)cur;
   const double t = (site - (double)cur) / width;
   for (int c = 0; c < P.R; ++c)
   {
      const double *__restrict__ x = th1 + op.off1 * P.R + (int64_t)c * nIn;
      const double b0 = smooth_at(x, nIn, half, w, cur), b1 = smooth_at(x, nIn, half, w, cur + 1);
      th2[op.off2 * P.R + (int64_t)c * nDown + i] = b0 + (b1 - b0) * t; // Spline::interp1linear, spline.cpp:108-120
   }
}

// back to the resolution the user asked for (ba.cpp:1873-1919): natural splines of the down-sampled channels
// (second derivatives in sol2) evaluated at nF uniform sites of the unit interval
__global__ void k_out_user(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ th2, const double *__restrict__ sol2,
                           double *__restrict__ thF, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 3)];
   const int i = (int)(g - op.offF), n = op.n2, nUser = op.nF;
   if (i >= nUser) return;
   const double c1 = 1. / (double)(n - 1), c2 = 1. / (double)(nUser - 1);
   const double site = c2 * (double)i;
   const int seg = seg_uniform(site, c1, n);
   const double g0 = c1 * (double)seg, g1 = c1 * (double)(seg + 1);
   const double tau = (site - g0) / (g1 - g0);
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   for (int c = 0; c < P.R; ++c)
   {
      const int64_t at = op.off2 * P.R + (int64_t)c * n + seg;
      const Coef4 k = coeffs_from_sol(sol2[at], sol2[at + 1], th2[at], th2[at + 1]);
      thF[op.offF * P.R + (int64_t)c * nUser + i] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
   }
}

} // namespace bk
