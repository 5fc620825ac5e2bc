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
   int kmC;         // ... channels per knot in the pair array (Cin, or C when all channels are pairs)
   int svd;         // BATOTP_F_SVD: the cable tensions through the Jacobi SVD instead of the LU
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

// findInterpSegs' cursor never moves back: segment of site i = max over the sites up to i -- a running maximum of integers, which
// (unlike the floating-point recurrences of this file) IS associative: one wavefront per path, 256 sites per round (four
// consecutive ones per lane, a shuffle scan over the lanes' maxima, the carry of the rounds before), the next round's loads under
// way while this one scans.  Round 5; the lane-per-path loop it replaces took 38.7 ms per launch for 256 paths of 2.2e5 sites,
// 43 % of the output stage.
__global__ void __launch_bounds__(64) k_out_segmax(const OutPath *__restrict__ paths, int K, int *__restrict__ segK)
{
   const int k = blockIdx.x, lane = threadIdx.x;
   if (k >= K) return;
   int *__restrict__ s = segK + paths[k].off1;
   const int n = paths[k].n1;
   int run = 0; // the maximum of everything before this round (segments are >= 0)
   int v0, v1, v2, v3;
   auto load = [&](int at, int &w0, int &w1, int &w2, int &w3) {
      w0 = at < n ? s[at] : 0;
      w1 = at + 1 < n ? s[at + 1] : 0;
      w2 = at + 2 < n ? s[at + 2] : 0;
      w3 = at + 3 < n ? s[at + 3] : 0;
   };
   load(4 * lane, v0, v1, v2, v3);
   for (int base = 0; base < n; base += 256)
   {
      const int at = base + 4 * lane;
      int w0, w1, w2, w3;
      load(at + 256, w0, w1, w2, w3);
      const int m0 = v0, m1 = max(m0, v1), m2 = max(m1, v2), m3 = max(m2, v3);
      int incl = m3; // inclusive maximum over the lanes up to this one
#pragma unroll
      for (int d = 1; d < 64; d <<= 1)
      {
         const int o = __shfl_up(incl, d);
         if (lane >= d) incl = max(incl, o);
      }
      int before = __shfl_up(incl, 1);
      before = max(run, lane ? before : 0);
      const int r0 = max(m0, before), r1 = max(m1, before), r2 = max(m2, before), r3 = max(m3, before);
      if (at < n && r0 != v0) s[at] = r0;
      if (at + 1 < n && r1 != v1) s[at + 1] = r1;
      if (at + 2 < n && r2 != v2) s[at + 2] = r2;
      if (at + 3 < n && r3 != v3) s[at + 3] = r3;
      run = max(run, __shfl(incl, 63));
      v0 = w0; v1 = w1; v2 = w2; v3 = w3;
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
         const double *a = km + ((pi.koff + seg) * P.kmC + c) * 2, *b = a + 2 * P.kmC;
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
   solve3(P.svd ? (unsigned)BATOTP_F_SVD : 0u, A, b, xs);
   double *__restrict__ o = dst + op.off1 * P.R + i;
#pragma unroll
   for (int r = 0; r < 6; ++r) o[(int64_t)r * n1] = val[r];
#pragma unroll
   for (int j = 0; j < 3; ++j) o[(int64_t)(6 + j) * n1] = xs[j];
}

// ---------------------------------------------------------------------------------------------
// JOINT paths of the robots with forward kinematics and the serial-robot torque recomputation
// (ba.cpp:1722-1725, 1744-1750, 1791-1827)
// ---------------------------------------------------------------------------------------------
// Cartesian rows nJ..nJ+2 of a stage array x[R][n1] from its joint rows (fwdkin_point: resample.hip.h).  The two-link
// arm's routine leaves the third row alone (robot.cpp:192-193): it keeps what the Traj held -- the knot samples of that
// channel (zsamp: value rows of the batch's sample array) -- cut or zero-extended to the new length.
__global__ void k_out_fwdkin(OutParams P, int robot, const OutPath *__restrict__ paths, int K, double *__restrict__ x, const double *__restrict__ trig,
                             int trigRows, const PathInfo *__restrict__ pinfo, const double *__restrict__ samp, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   double *__restrict__ xb = x + op.off1 * P.R;
   if (robot == BATOTP_ROBOT_RR)
   {
      const PathInfo pi = pinfo[op.p];
      const double *__restrict__ z = samp + pi.koff * P.Cin * 3 + (int64_t)(P.nJ + 2) * 3 * pi.n;
      xb[(int64_t)(P.nJ + 2) * n1 + i] = i < pi.n ? z[i] : 0.0;
   }
   fwdkin_point(robot, xb, xb + (int64_t)P.nJ * n1, trig ? trig + op.off1 * trigRows : nullptr, n1, i);
}

// joint rows of a stage array x[R][n1], packed [nJ][n1] per path at off1*nJ (what the host needs for trig tables)
__global__ void k_out_pack_theta(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ x, double *__restrict__ out, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   for (int j = 0; j < P.nJ; ++j) out[op.off1 * P.nJ + (int64_t)j * n1 + i] = x[op.off1 * P.R + (int64_t)j * n1 + i];
}

// Spline::solveTriDiagClamped (spline.cpp:225-243) on the right-hand sides 6(y[i-1] - 2y[i] + y[i+1]) of
// Spline::getSplineCoeffs (spline.cpp:186-190): series k has n[k] values y[yOff[k] + i] and leaves its second derivatives at
// sol[solOff[k] + i].  One lane per series.  Pivots: b = 2 in the first and last row, 4 elsewhere; the super-diagonal
// c[i] = 1/(b[i] - c[i-1]) depends on i alone and is a table (c_ctab_cl, constant from index 63 on; checked by the host).  The
// back substitution starts at row n-3 ("for (i = n-2; i-- > 0;)", spline.cpp:240): row n-2 keeps its eliminated value.
__constant__ double c_ctab_cl[64];
__global__ void __launch_bounds__(64) k_spline_series_clamped(int count, const int64_t *__restrict__ yOff, const int64_t *__restrict__ solOff,
                                                              const int *__restrict__ n, const double *__restrict__ y, double *__restrict__ sol)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= count || n[k] < 4) return;
   const int N = n[k];
   const double *__restrict__ yy = y + yOff[k];
   double *__restrict__ d = sol + solOff[k];
   double dprev = 0.0 / 2.0;                       // d[0] /= b[0] with sol[0] = 0
   d[0] = dprev;
   double ym = yy[0], y0 = yy[1];
   for (int i = 1; i < N; ++i)
   {
      const double cprev = (i - 1) < 63 ? c_ctab_cl[i - 1] : c_ctab_cl[63];
      double rhs = 0.0, bi = 2.0;
      if (i < N - 1)
      {
         const double yp = yy[i + 1];
         rhs = 6 * (ym - 2 * y0 + yp);
         ym = y0; y0 = yp;
         bi = 4.0;
      }
      const double di = (rhs - 1.0 * dprev) / (bi - 1.0 * cprev);
      d[i] = di;
      dprev = di;
   }
   double next = d[N - 2];
   for (int i = N - 3; i >= 0; --i)
   {
      const double ci = i < 63 ? c_ctab_cl[i] : c_ctab_cl[63];
      const double v = d[i] - ci * next;
      d[i] = v;
      next = v;
   }
}

// ba.cpp:1791-1800: every site evaluated at the END of the previous segment of the clamped spline through the joint samples
// (site 0: start of segment 0): value, first and second derivative per joint -> a sample array laid out like the batch's
// ([Cin][3][n1] per path at off1*Cin*3: what k_dynamics / k_dyn_serial read), a packed copy of the values ([nJ][n1] at
// off1*nJ: what the host needs for the trig tables) and the joint rows of dst; the Cartesian rows are carried over.
__global__ void k_out_serial_eval(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ src, const double *__restrict__ solC,
                                  double *__restrict__ dst, double *__restrict__ sampT, double *__restrict__ pack, int nCartRows, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   const int seg = i == 0 ? 0 : i - 1;
   const double tau = i == 0 ? 0.0 : 1.0;
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   for (int j = 0; j < P.nJ; ++j)
   {
      const int64_t at = op.off1 * P.R + (int64_t)j * n1 + seg;
      const Coef4 k = coeffs_from_sol(solC[at], solC[at + 1], src[at], src[at + 1]);
      const double v = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;          // spline.cpp:149-151
      const double v1 = (3 * k.c3 * tau2 + 2 * k.c2 * tau + k.c1) * P.vfactT;
      const double v2 = (6 * k.c3 * tau + 2 * k.c2) * P.afactT;
      double *__restrict__ sp = sampT + op.off1 * P.Cin * 3 + (int64_t)j * 3 * n1;
      sp[i] = v; sp[n1 + i] = v1; sp[2 * (int64_t)n1 + i] = v2;
      pack[op.off1 * P.nJ + (int64_t)j * n1 + i] = v;
      dst[op.off1 * P.R + (int64_t)j * n1 + i] = v;
   }
   for (int c = 0; c < nCartRows; ++c)
      dst[op.off1 * P.R + (int64_t)(P.nJ + c) * n1 + i] = src[op.off1 * P.R + (int64_t)(P.nJ + c) * n1 + i];
}

// torque rows = a2 + a3 + a4 (ba.cpp:1819-1825) from the dynamics array [4][nJ][n1] per path at off1*4*nJ
__global__ void k_out_trq_sum(OutParams P, const OutPath *__restrict__ paths, int K, const double *__restrict__ dyn, double *__restrict__ dst, int r0,
                              int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 1)];
   const int i = (int)(g - op.off1), n1 = op.n1;
   if (i >= n1) return;
   const double *__restrict__ dy = dyn + op.off1 * 4 * P.nJ;
   for (int j = 0; j < P.nJ; ++j)
      dst[op.off1 * P.R + (int64_t)(r0 + j) * n1 + i] =
         dy[((int64_t)1 * P.nJ + j) * n1 + i] + dy[((int64_t)2 * P.nJ + j) * n1 + i] + dy[((int64_t)3 * P.nJ + j) * n1 + i];
}

// ---------------------------------------------------------------------------------------------
// Pose rows of a BOTH path at the very end of the stage: BA::q2aaVect (ba.cpp:384-403) with q2aa (util.cpp:562-581).  The
// working arrays carry position + quaternion (7 Cartesian rows), the result position + axis-angle (6).
//   k_out_q_norm  the vector part's norm and q0 of every final point, packed [2][nF] per path at offF*2 (host atan2 table)
//   k_out_q2aa    src [Rw][nF] -> dst [Rw - 1][nF]; at = atan2(norm, q0) per point from the host, or nullptr (device libm)
// ---------------------------------------------------------------------------------------------
__global__ void k_out_q_norm(int nJ, int Rw, const OutPath *__restrict__ paths, int K, const double *__restrict__ src, double *__restrict__ pack,
                             int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 3)];
   const int i = (int)(g - op.offF), n = op.nF;
   if (i >= n) return;
   const double *__restrict__ q = src + op.offF * Rw + (int64_t)(nJ + 3) * n + i;
   const double q1 = q[n], q2 = q[2 * (int64_t)n], q3 = q[3 * (int64_t)n];
   pack[op.offF * 2 + i] = sqrt(q1 * q1 + q2 * q2 + q3 * q3);     // util.cpp:565
   pack[op.offF * 2 + n + i] = q[0];
}

__global__ void k_out_q2aa(int nJ, int Rw, const OutPath *__restrict__ paths, int K, const double *__restrict__ src, const double *__restrict__ at,
                           double *__restrict__ dst, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const OutPath op = paths[out_find_path(paths, K, g, 3)];
   const int i = (int)(g - op.offF), n = op.nF;
   if (i >= n) return;
   const double *__restrict__ s = src + op.offF * Rw + i;
   double *__restrict__ o = dst + op.offF * (Rw - 1) + i;
   for (int r = 0; r < nJ + 3; ++r) o[(int64_t)r * n] = s[(int64_t)r * n];
   const double q0 = s[(int64_t)(nJ + 3) * n], q1 = s[(int64_t)(nJ + 4) * n], q2 = s[(int64_t)(nJ + 5) * n], q3 = s[(int64_t)(nJ + 6) * n];
   const double norme = sqrt(q1 * q1 + q2 * q2 + q3 * q3);
   double a0 = 0.0, a1 = 0.0, a2 = 0.0;
   if (!(norme < 1e-6))
   {
      const double ang = at ? at[op.offF + i] : atan2(norme, q0);
      const double theta = 2.0 * ang / norme;                        // util.cpp:573
      a0 = theta * q1; a1 = theta * q2; a2 = theta * q3;
   }
   o[(int64_t)(nJ + 3) * n] = a0; o[(int64_t)(nJ + 4) * n] = a1; o[(int64_t)(nJ + 5) * n] = a2;
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
