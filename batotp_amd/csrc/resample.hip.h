// resample.hip.h -- GPU port of the path resampling that precedes the hot path:
// remClosePts (util.cpp:452-524), the two adjust_s passes (ba.cpp:412-638), interpSpecial
// (ba.cpp:651-781) and the N_old -> N_new evalSplineFullTraj (ba.cpp:790-863), for the path kinds of
// BASELINE configs 2, 4 and 5: JOINT paths of a robot without kinematic model (GENJNT) and CART
// paths of the cable robot (cable lengths by Robot::invKinCSPR3DOF, robot.cpp:243-278).
// Included by batotp_hip.hip.  Arithmetic contract as everywhere: fp64, no contraction, the
// reference's operation order; the host resampler (batotp_amd/host/ba_input.cpp), itself pinned by
// the reference binary's outputs, is the checker (tests/test_gpu_resample.py).
//
// The walks are sequential per path (cumulative sums, data-dependent emission): the spline builds reuse the hot path's Thomas kernel (one lane per channel), the uniform
// re-evaluation runs one lane per output site.
//
// Stage arrays are channel-major per path: x[path base + c*n + i], path base = off*C.
#pragma once
#include "kernels.hip.h"

namespace bk
{

struct RsParams
{
   int nJ, nC, C;
   int pathType;  // 1 JOINT, 2 CART
   int scaleType; // 0 input nodes, 1 joint-space arc length, 2 Cartesian arc length
   int robot;
   int cartEval;  // a Cartesian constraint is on: interpSpecial re-evaluates the Cartesian channels (ba.cpp:1362)
   int pad;
   double sW[3];
   double thetaRes, cartRes; // of the current pass
   double thresh;            // remClosePts threshold of the driving channel set
   double pmat[9];
   // automatic integration resolution (reference ba.cpp:462-470, 493-556): inputs of the rule
   int autoOn, degrees;
   double vmax[8], amax[8], cartVelMax, cartAccMax, quadThresh;
};

struct RsPath
{
   int64_t off;   // first point of this path in the stage arrays (points, not doubles)
   int32_t n;     // points in the stage
   int32_t status;
   double sres;   // traj.sres entering the pass
   double sLast, sResNew, tTeachFact, thetaFact, cartFact, sresNew; // set by k_rs_arclen
   int32_t nOut;  // points the pass produces (special: emitted; regular: nPtsNew)
   int32_t cap;   // capacity of the special pass' output rows
   int64_t offOut;
   int32_t n0;    // points the stage arrays reserve for this path (n <= n0; remClosePts shrinks n)
   int32_t pad;
   // what the reference keeps in the BA object across the two adjust_s passes and the automatic integration resolution
   // rewrites PER PATH: _sWeights[1], _sWeights[2], _scaleType, _integRes
   double sw1, sw2, integRes;
   int32_t scaleType, pad2;
};

// status bits of the resampler
constexpr int RS_TOO_SHORT = 1;   // fewer than 4 points somewhere: the host path handles it (interpTrajLinear)
constexpr int RS_IDENTICAL = 2;   // "all points identical" exit of adjust_s (ba.cpp:484-488)
constexpr int RS_CAPACITY = 4;    // interpSpecial produced more points than planned for
constexpr int RS_SMALL_STEP = 8;  // s-resolution too small between two points (ba.cpp:607-611)
constexpr int RS_SEG_ERROR = 16;  // findInterpSegs division by zero (spline.cpp:84-88)

// ---------------------------------------------------------------------------------------------
// remClosePts (util.cpp:452-524): one lane per path, in place; repacks the channels when points
// were dropped.  Driving set = theta channels (JOINT) or Cartesian channels (CART).
// ---------------------------------------------------------------------------------------------
// Does a path have ANY pair of consecutive points closer than the threshold?  One lane per point; sets flag[path].  Taught paths
// rarely do, and k_rs_remclose -- sequential per path, one lane each: 41 ms for 256 paths of 3.4e4 points, whatever they contain --
// then returns at once for every path whose flag is 0 (round 5).  The predicate is the first pass of remClosePts with the `!drop[i-1]`
// condition left out: a superset, so a path without a flag has nothing to drop.
__global__ void k_rs_close_any(RsParams P, const RsPath *__restrict__ paths, int B, const double *__restrict__ x, int *__restrict__ flag, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (paths[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i < 1 || i >= n) return;
   const double *__restrict__ xb = x + pp.off * P.C;
   const int c0 = (P.pathType == 2) ? P.nJ : 0;
   const int cN = (P.pathType == 2) ? P.nC : P.nJ;
   double sum = 0;
   for (int j = 0; j < cN; ++j)
   {
      const double d = xb[(int64_t)(c0 + j) * n + i] - xb[(int64_t)(c0 + j) * n + i - 1];
      sum += d * d;
   }
   if (sum < P.thresh * P.thresh) atomicOr(&flag[lo], 1);
}

__global__ void k_rs_remclose(RsParams P, RsPath *__restrict__ paths, int B, double *__restrict__ x, unsigned char *__restrict__ drop,
                              const int *__restrict__ flag)
{
   const int p = blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= B) return;
   RsPath &pp = paths[p];
   const int n0 = pp.n;
   if (flag && !flag[p])
   {
      // no two consecutive points within the threshold (k_rs_close_any): nothing to drop, only the length test remains
      if (n0 < 4) pp.status |= RS_TOO_SHORT;
      return;
   }
   double *__restrict__ xb = x + pp.off * P.C;
   unsigned char *__restrict__ dr = drop + pp.off;
   const int c0 = (P.pathType == 2) ? P.nJ : 0;
   const int cN = (P.pathType == 2) ? P.nC : P.nJ;
   const double thrSq = P.thresh * P.thresh;
   int n = n0;
   for (int i = 0; i < n; ++i) dr[i] = 0;
   for (;;)
   {
      bool any = false;
      for (int i = 1; i < n; ++i)
      {
         double sum = 0;
         for (int j = 0; j < cN; ++j)
         {
            const double d = xb[(int64_t)(c0 + j) * n0 + i] - xb[(int64_t)(c0 + j) * n0 + i - 1];
            sum += d * d;
         }
         if (sum < thrSq && !dr[i - 1]) { dr[i] = 1; any = true; }
      }
      if (dr[n - 1] && n > 2) { dr[n - 1] = 0; dr[n - 2] = 1; dr[n - 3] = 0; }
      if (!any) break;
      int keep = 0;
      for (int i = 0; i < n; ++i)
      {
         if (dr[i]) continue;
         for (int c = 0; c < P.C; ++c) xb[(int64_t)c * n0 + keep] = xb[(int64_t)c * n0 + i];
         ++keep;
      }
      n = keep;
      for (int i = 0; i < n; ++i) dr[i] = 0;
   }
   if (n != n0)
   {
      for (int c = 1; c < P.C; ++c)
         for (int i = 0; i < n; ++i) xb[(int64_t)c * n + i] = xb[(int64_t)c * n0 + i];
      pp.n = n;
   }
   if (n < 4) pp.status |= RS_TOO_SHORT;
}

// smooth() (util.cpp:263-290) at one index: centred moving average of width w = 2*half + 1, shrinking windows
// next to the ends, the end points themselves unchanged
__device__ __forceinline__ double smooth_at(const double *__restrict__ x, int n, int half, int w, int i)
{
   if (i == 0 || i == n - 1) return x[i];
   if (i < half)
   {
      const int span = 2 * i + 1;
      double head = 0;
      for (int j = 0; j < span; ++j) head += x[j];
      return head / span;
   }
   if (i >= n - half)
   {
      const int r = n - 1 - i, span = 2 * r + 1;
      double tail = 0;
      for (int j = 0; j < span; ++j) tail += x[n - j - 1];
      return tail / span;
   }
   double acc = 0;
   for (int j = i - half; j < i + half + 1; ++j) acc += x[j];
   return acc / w;
}

// input smoothing of the driving rows [c0, c0+cN) (ba.cpp:195-242 calls smooth() with window w on them); the other
// rows are copied.  src -> dst, same layout; one lane per point
__global__ void k_rs_smooth(RsParams P, const RsPath *__restrict__ paths, int B, const double *__restrict__ src, double *__restrict__ dst,
                            int c0, int cN, int window, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (paths[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n || pp.status) return;
   int w = window < n ? window : n;
   const int half = w / 2 + w % 2 - 1;
   w = 2 * half + 1;
   for (int c = 0; c < P.C; ++c)
   {
      const double *__restrict__ x = src + pp.off * P.C + (int64_t)c * n;
      dst[pp.off * P.C + (int64_t)c * n + i] = (c >= c0 && c < c0 + cN) ? smooth_at(x, n, half, w, i) : x[i];
   }
}

// decimate() (util.cpp:347-356) of every row: each w-th sample, always keeping the last one.  src (layout of `from`) ->
// dst (layout of `to`, to[p].n = (from[p].n - 1)/w + 1); one lane per kept point
__global__ void k_rs_decimate(RsParams P, const RsPath *__restrict__ from, const RsPath *__restrict__ to, int B, const double *__restrict__ src,
                              double *__restrict__ dst, int w, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (to[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const int i = (int)(g - to[lo].off), nOut = to[lo].n, nIn = from[lo].n;
   if (i >= nOut || to[lo].status) return;
   const int pick = (i == nOut - 1 && w * (nOut - 1) + 1 != nIn) ? nIn - 1 : w * i;
   for (int c = 0; c < P.C; ++c) dst[to[lo].off * P.C + (int64_t)c * nOut + i] = src[from[lo].off * P.C + (int64_t)c * nIn + pick];
}

// Robot::invKinCSPR3DOF (robot.cpp:243-278): cable lengths from the platform position; one lane per point
__global__ void k_rs_invkin_cspr(RsParams P, const RsPath *__restrict__ paths, int B, double *__restrict__ x, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (paths[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off);
   if (i >= pp.n) return;
   double *__restrict__ xb = x + pp.off * P.C;
   const int n = pp.n;
   const double px = xb[(int64_t)(P.nJ + 0) * n + i], py = xb[(int64_t)(P.nJ + 1) * n + i], pz = xb[(int64_t)(P.nJ + 2) * n + i];
#pragma unroll
   for (int k = 0; k < 3; ++k)
   {
      const double dx = px - P.pmat[0 * 3 + k], dy = py - P.pmat[1 * 3 + k], dz = pz - P.pmat[2 * 3 + k];
      double sq = 0.0;
      sq += dx * dx;
      sq += dy * dy;
      sq += dz * dz;
      xb[(int64_t)k * n + i] = sqrt(sq);
   }
}

// ---------------------------------------------------------------------------------------------
// adjust_s up to the sC array (ba.cpp:430-590, _isAutoIntegRes = false) in three kernels:
//   k_rs_seglen  one lane per point: joint-space and Cartesian length of the step to each point
//   k_rs_scan    one lane per path: the two running sums (sequential: fp addition order is the
//                reference's), the scale factors, the size of the pass' output
//   k_rs_sites   one lane per point: s of every point and, in the second pass, the spacing checks
// special != 0: first pass (interpSpecial follows); 0: second pass (evalSplineFullTraj follows).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int rs_find_path(const RsPath *__restrict__ paths, int B, int64_t g)
{
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (paths[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   return lo;
}

__global__ void k_rs_seglen(RsParams P, const RsPath *__restrict__ paths, int B, const double *__restrict__ x, double *__restrict__ thetaArc,
                            double *__restrict__ cartArc, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n || pp.status) return;
   if (i == 0) { thetaArc[g] = 0; cartArc[g] = 0; return; }
   const double *__restrict__ xb = x + pp.off * P.C;
   double sq = 0;
   for (int j = 0; j < P.nJ; ++j)
   {
      const double d = xb[(int64_t)j * n + i] - xb[(int64_t)j * n + i - 1];
      sq += d * d;
   }
   thetaArc[g] = sqrt(sq);
   sq = 0;
   for (int j = 0; j < 3; ++j)
   {
      const double d = xb[(int64_t)(P.nJ + j) * n + i] - xb[(int64_t)(P.nJ + j) * n + i - 1];
      sq += d * d;
   }
   cartArc[g] = sqrt(sq);
}

// ba.cpp:493-556 once the arc lengths of a pass are known: the path's integration step, s weights, scale type and the Cartesian
// resolution of the pass.  std::min / std::max in the reference's argument order (their NaN behaviour is part of the result: a
// robot without Cartesian limits gets 0/0 here, and keeps it).
__device__ __forceinline__ void rs_auto_rule(const RsParams &P, RsPath &pp, double thetaLast, double cartLast, double minCartPerTheta, double &cartRes)
{
   if (cartLast < cartRes && pp.scaleType == 2)
   {
      pp.sw1 = pp.sw1 + pp.sw2;
      pp.sw2 = 0;
      pp.scaleType = 1;
   }
   const double weightIn = pp.sw1 + pp.sw2;
   double cartRat = 500.0 * cartLast;
   double thetaRat = thetaLast;
   if (!P.degrees) thetaRat *= 180.0 / 3.14159265358979323846;
   const double lo = 0.004, hi = 0.2, K = 0.0003;
   double step = K * P.cartAccMax / P.cartVelMax;
   for (int j = 0; j < P.nJ; ++j) step = dmax(step, K * P.amax[j] / P.vmax[j]);
   step = dmin(step, hi);
   const double ratio = cartRat / thetaRat;
   double byJoints = hi * ratio * ratio;
   double byWindow = hi * minCartPerTheta * minCartPerTheta;
   byWindow = dmax(byWindow, 0.016);
   byJoints = dmin(byJoints, byWindow);
   if (byJoints < step) step = byJoints;
   step = dmax(step, lo);
   pp.integRes = step;
   const double rescale = weightIn / (cartRat + thetaRat);
   cartRat *= rescale;
   thetaRat *= rescale;
   if (thetaRat > pp.sw1) { pp.sw1 = thetaRat; pp.sw2 = cartRat; }
   if (pp.sw2 > 0) cartRes = dmin(cartRes, cartRes * pp.sw2 / pp.sw1);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
   const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
   return __hiloint2double(hi, lo);
}

// lane L of the own row (16 lanes) in every lane: v_mov_b32_dpp row_newbcast:L on both halves
template <int L> __device__ __forceinline__ double row_bcast_f64(double v)
{
   const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + L, 0xf, 0xf, false);
   const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + L, 0xf, 0xf, false);
   return __hiloint2double(hi, lo);
}

// what adjust_s derives from the two arc lengths of a path (ba.cpp:472-590): the scale factors and the size of the pass' output
__device__ __forceinline__ void rs_scan_finish(const RsParams &P, RsPath &pp, int n, double tacc, double cacc, double minCartPerTheta, int special)
{
   const bool autoOn = P.autoOn != 0;
   if (tacc < P.thetaRes) { pp.status |= RS_IDENTICAL; return; }

   double cartRes = P.cartRes;
   if (autoOn) rs_auto_rule(P, pp, tacc, cacc, minCartPerTheta, cartRes);
   const double sResi = pp.sres;
   const double ptsLast = (double)(n - 1); // traj.ptsOrig is 0,1,2,.. at both call sites
   double sLast = 0, sResNew = 0;
   switch (pp.scaleType)
   {
   case 0: sLast = sResi * ptsLast; sResNew = sResi; break;
   case 1: sLast = tacc; sResNew = P.thetaRes; break;
   default: sLast = cacc; sResNew = cartRes; break;
   }
   double cartFact = 0;
   if (cacc >= cartRes) cartFact = pp.sw2 * sLast / cacc;
   const double teachFact = P.sW[0] * sLast / (sResi * ptsLast);
   const double thetaFact = pp.sw1 * sLast / tacc;
   pp.sLast = sLast; pp.sResNew = sResNew; pp.tTeachFact = teachFact; pp.thetaFact = thetaFact; pp.cartFact = cartFact;
   pp.sresNew = sLast / (n - 1); // traj.sres = sLast/(nPts-1), ba.cpp:585
   if (special)
   {
      int nPts2 = (int)ceil(sLast / sResNew) + 1; // ba.cpp:666-667
      if (nPts2 < 4) nPts2 = 4;
      pp.nOut = nPts2; // planning figure; the walk reports the real count
   }
   else
   {
      // evalSplineFullTraj(traj, traj.sres, sResNew): ba.cpp:796-798
      int nNew = (int)ceil(pp.sresNew / sResNew * (n - 1)) + 1;
      if (nNew < 4) nNew = 4;
      pp.nOut = nNew;
   }
}

__global__ void k_rs_scan(RsParams P, RsPath *__restrict__ paths, int B, double *__restrict__ thetaArc, double *__restrict__ cartArc, int special)
{
   const int p = blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= B) return;
   RsPath &pp = paths[p];
   if (pp.status) return;
   const int n = pp.n;
   double *__restrict__ ta = thetaArc + pp.off, *__restrict__ ca = cartArc + pp.off;
   // thetaNorm[i+1] = thetaNorm[i] + dtheta (ba.cpp:452-470); step lengths are loaded a batch at a time
   constexpr int CH = 16;
   double tacc = 0, cacc = 0;
   // automatic integration resolution, ba.cpp:441-446, 462-470: the smallest Cartesian advance per joint-space advance over
   // windows of 5 degrees (a sequential scan with marks: it rides on the running sums)
   const bool autoOn = P.autoOn != 0;
   double minCartPerTheta = 1.0 / P.quadThresh, window = 5.0, tMark = 0, cMark = 0;
   if (!P.degrees) window *= 3.14159265358979323846 / 180.0;
   int i = 1;
   for (; i + CH <= n; i += CH)
   {
      double dt[CH], dc[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) { dt[k] = ta[i + k]; dc[k] = ca[i + k]; }
#pragma unroll
      for (int k = 0; k < CH; ++k)
      {
         tacc = tacc + dt[k]; ta[i + k] = tacc;
         cacc = cacc + dc[k]; ca[i + k] = cacc;
         if (autoOn)
         {
            const double dTh = tacc - tMark, dCa = cacc - cMark;
            if (dTh > window) { minCartPerTheta = dmin(minCartPerTheta, 3.0 * dCa / dTh); tMark = tacc; cMark = cacc; }
         }
      }
   }
   for (; i < n; ++i)
   {
      tacc = tacc + ta[i]; ta[i] = tacc;
      cacc = cacc + ca[i]; ca[i] = cacc;
      if (autoOn)
      {
         const double dTh = tacc - tMark, dCa = cacc - cMark;
         if (dTh > window) { minCartPerTheta = dmin(minCartPerTheta, 3.0 * dCa / dTh); tMark = tacc; cMark = cacc; }
      }
   }
   rs_scan_finish(P, pp, n, tacc, cacc, minCartPerTheta, special);
}

// The same running sums with a ROW (16 lanes) per chain: a wavefront takes two paths, rows 0 / 1 the joint-space and the
// Cartesian chain of the first, rows 2 / 3 those of the second.  A row loads 16 consecutive step lengths at once (one 128-byte
// line), lane k forms "carry + v0 + v1 + ... + vk" in exactly that order -- step j adds lane j's value, by DPP row broadcast,
// in the lanes >= j and +0 in the others (arc lengths are >= +0: adding +0 leaves no trace) --, lane 15 hands its sum on as
// the next carry.  Five instructions per step for four chains, against a lone lane per path that waits for its own 16 loads:
// 36 -> 8.5 ms for the four launches of two 1024-path GEN7DOF calls of tools/bench_resample.py.  Without the automatic integration resolution (its window marks are
// a scan of their own over both sums: k_rs_scan keeps those calls).
#define RS_SCAN_STEP(J) acc = acc + (rl >= (J) ? row_bcast_f64<(J)>(v) : 0.0)
__global__ void __launch_bounds__(64) k_rs_scan_rows(RsParams P, RsPath *__restrict__ paths, int B, double *__restrict__ thetaArc,
                                                    double *__restrict__ cartArc, int special)
{
   const int lane = threadIdx.x, row = lane >> 4, rl = lane & 15;
   const int pA = 2 * blockIdx.x, pB = pA + 1;
   const int nA = (pA < B && !paths[pA].status) ? paths[pA].n : 0, nB = (pB < B && !paths[pB].status) ? paths[pB].n : 0;
   const int p = row < 2 ? pA : pB, n = row < 2 ? nA : nB, nMax = max(nA, nB);
   double *__restrict__ a = ((row & 1) ? cartArc : thetaArc) + (n ? paths[p].off : 0);
   double acc = 0; // the carry of the row's chain on entry of a block, the lane's own sum on exit
   double v = 1 + rl < n ? a[1 + rl] : 0.0;
   for (int i = 1; i < nMax; i += 16)
   {
      const int ahead = i + 16 + rl;
      const double vNext = ahead < n ? a[ahead] : 0.0; // under way while this block adds
      RS_SCAN_STEP(0); RS_SCAN_STEP(1); RS_SCAN_STEP(2); RS_SCAN_STEP(3); RS_SCAN_STEP(4); RS_SCAN_STEP(5); RS_SCAN_STEP(6); RS_SCAN_STEP(7);
      RS_SCAN_STEP(8); RS_SCAN_STEP(9); RS_SCAN_STEP(10); RS_SCAN_STEP(11); RS_SCAN_STEP(12); RS_SCAN_STEP(13); RS_SCAN_STEP(14); RS_SCAN_STEP(15);
      if (i + rl < n) a[i + rl] = acc;
      acc = row_bcast_f64<15>(acc); // (points past the end added +0: the carry stays the path's total)
      v = vNext;
   }
   // the Cartesian total sits one row up
   const int hiC = __shfl(__double2hiint(acc), lane + 16), loC = __shfl(__double2loint(acc), lane + 16);
   if (rl == 0 && !(row & 1) && n) rs_scan_finish(P, paths[p], n, acc, __hiloint2double(hiC, loC), 0.0, special);
}
#undef RS_SCAN_STEP

// RsPath::pad collects the spacing findings of the second pass: bit 0 "s-resolution is too small"
// (ba.cpp:607-611), bit 1 findInterpSegs' zero width (spline.cpp:84-88); the host turns them into
// the status the sequential code would have returned (the first check wins).
__global__ void k_rs_sites(RsParams P, RsPath *__restrict__ paths, int B, const double *__restrict__ thetaArc, const double *__restrict__ cartArc,
                           double *__restrict__ sC, int special, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n || pp.status) return;
   const double a = pp.tTeachFact * pp.sres;
   const double si = a * (double)i + pp.thetaFact * thetaArc[g] + pp.cartFact * cartArc[g];
   sC[g] = si;
   if (!special && i > 0)
   {
      const double sm = a * (double)(i - 1) + pp.thetaFact * thetaArc[g - 1] + pp.cartFact * cartArc[g - 1];
      int f = 0;
      if (si - sm < 1e-12 * pp.sresNew) f |= 1;
      if (si - sm < 1e-20) f |= 2;
      if (f) atomicOr(&paths[lo].pad, f);
   }
}

// ---------------------------------------------------------------------------------------------
// interpSpecial (ba.cpp:651-781): walk along the original points and emit a new point every sResNew
// of (weighted) distance from the last emitted one, by evaluating the splines of the original
// points.  Output rows are point-major [i][C] (the count is only known at the end).
//
// One WAVEFRONT per path, a lane per channel (each of the four 16-lane rows carries all channels).  The walk is a serial
// chain (every emitted point depends on the previous one) with data-dependent control flow; with a lane per path the 64
// paths of a wavefront drag each other through both branches and through every memory wait.  With a wavefront per path
// the control flow is uniform, the channel work of one step runs across lanes, and what is left is the LATENCY of the
// chain (at <= 1024 paths a SIMD holds one wavefront: nothing hides anything), so: the taught points come from an LDS
// window (round 5), the sums of squares take their operands by DPP row broadcast instead of v_readlane round trips through
// the scalar unit, and values that all lanes share (distances, s, tau) are computed redundantly in every lane, in the
// reference's order of operations.
// ---------------------------------------------------------------------------------------------
constexpr int RS_WIN = 64; // taught points per LDS window (one per lane in a refill)
// k_rs_special's lane layout: a 16-lane row holds the joints in its lanes 0..7 and the Cartesian channels in 8..15 (x, y, z first)
static_assert(BATOTP_MAX_JOINTS <= 8 && BATOTP_MAX_CART <= 8, "k_rs_special carries at most 8 joints and 8 Cartesian channels per 16-lane row");
constexpr int RS_BACK = 4; // points kept behind the one that caused a refill

__global__ void __launch_bounds__(64) k_rs_special(RsParams P, RsPath *__restrict__ paths, int B, const double *__restrict__ x,
                                                   const double *__restrict__ sC, const double *__restrict__ sol, double *__restrict__ rows)
{
   const int p = blockIdx.x;
   if (p >= B) return;
   RsPath &pp = paths[p];
   if (pp.status) return;
   const int lane = threadIdx.x;
   const int n = pp.n, C = P.C, nJ = P.nJ;
   // lanes of a ROW (16 lanes) of the lower half-wave: 0..7 the joints, 8..15 the Cartesian channels (row 0 stores); of the
   // upper half-wave: 0..7 the Cartesian channels again.  The sum of squares of a step then is ONE chain of eight DPP row
   // broadcasts -- the joint-space sum in the lower half, the Cartesian one (x, y, z) in the upper half -- and one square
   // root; lanes whose channel does not enter their half's sum contribute +0 (a partial sum is never -0, so adding +0
   // leaves no trace), spare lanes shadow the last channel of their kind
   const int rl = lane & 15, nC = C - nJ;
   const bool upper = lane >= 32;
   const int c = (rl < 8 && !upper) ? min(rl, nJ - 1) : nJ + min(rl & 7, nC - 1);
   const bool owner = lane < 16 && (rl < 8 ? rl < nJ : rl - 8 < nC);
   const bool inSum = rl < (upper ? 3 : nJ); // (rl < 8 follows: nJ <= 8)
   const double *__restrict__ xc = x + pp.off * C + (int64_t)c * n; // this lane's channel of the taught points
   const double *__restrict__ s = sC + pp.off;
   double *__restrict__ out = rows + pp.offOut * C + c;
   const int cap = pp.cap;
   const double sResNew = pp.sResNew, teach = pp.tTeachFact * pp.sres, thF = pp.thetaFact, caF = pp.cartFact;
   const double sEnd = s[n - 1];
   const bool cartCh = c >= nJ;
   const bool evalCh = !cartCh || P.cartEval != 0; // Cartesian channels keep traj.cartpt (zero) unless a Cartesian constraint is on

   // A window of RS_WIN consecutive taught points (values, second derivatives, sites) in LDS: the walk reads a point, a site
   // or a coefficient row as soon as the previous step has decided which -- from global memory every such read is a full
   // memory latency on the serial chain (one wavefront per SIMD at <= 1024 paths: nothing hides it).  A refill is 2C+1
   // coalesced 512-byte loads issued back to back; the walk moves forward (the cursor steps back only after a rounding
   // surprise), so a window serves ~RS_WIN points.
   __shared__ double wX[BATOTP_MAX_JOINTS + BATOTP_MAX_CART][RS_WIN], wM[BATOTP_MAX_JOINTS + BATOTP_MAX_CART][RS_WIN], wS[RS_WIN];
   const double *__restrict__ xp = x + pp.off * C, *__restrict__ mp = sol + pp.off * C;
   int w0 = 0;
   auto refill = [&](int first) {
      w0 = max(0, min(first, n - RS_WIN));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int k = w0 + lane;
      if (k < n)
      {
#pragma unroll 4
         for (int ch = 0; ch < C; ++ch)
         {
            wX[ch][lane] = xp[(int64_t)ch * n + k];
            wM[ch][lane] = mp[(int64_t)ch * n + k];
         }
         wS[lane] = s[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   };
   // index of point i in the window (i is the same in every lane: the refill is taken by the whole wavefront)
   auto at = [&](int i) {
      if (i < w0 || i >= w0 + RS_WIN) refill(i - RS_BACK);
      return i - w0;
   };
   refill(0);

   double prev = wX[c][0], xo = wX[c][1];
   Coef4 kk = coeffs_from_sol(wM[c][0], wM[c][1], prev, xo);
   double cartpt = 0.0;
   if (owner) out[0] = prev;
   double sPrv = 0, prvDs = 0;
   int newPt = 1, oldPt = 1, seg = 0, xoAt = 1, kAt = 0;
   double sA = wS[0], sB = wS[1]; // s[seg], s[seg+1]
   const int lastSeg = n - 2;
   bool done = false;
   while (!done)
   {
      if (xoAt != oldPt)
      {
         xo = wX[c][at(oldPt)];
         xoAt = oldPt;
      }
      const double d = xo - prev;
      const double d2 = d * d;
      // both sums of squares in the reference's order (0 + d2[0] + d2[1] + ...), joint space in the lower half-wave, x, y, z in the
      // upper one; then the two square roots side by side
      const double t2 = inSum ? d2 : 0.0;
      double sq = row_bcast_f64<0>(t2);
      sq += row_bcast_f64<1>(t2);
      sq += row_bcast_f64<2>(t2);
      sq += row_bcast_f64<3>(t2);
      sq += row_bcast_f64<4>(t2);
      sq += row_bcast_f64<5>(t2);
      sq += row_bcast_f64<6>(t2);
      sq += row_bcast_f64<7>(t2);
      const double root = sqrt(sq);
      const double ds = teach * (double)oldPt + thF * readlane_f64(root, 0) + caF * readlane_f64(root, 32);
      if (ds > sResNew)
      {
         const double sNew = sPrv + sResNew - prvDs;
         prvDs = 0;
         sPrv = sNew;
         if (sNew > sEnd) done = true;
         if (!done)
         {
            // evalSplinePartials: segment walk from the cached segment (ba.cpp:1617-1652), then the cubics
            for (;;)
            {
               if (sNew >= sA && sNew <= sB) break;
               bool moved = false;
               if (sNew > sA)
               {
                  if (seg >= lastSeg) break;
                  ++seg; sA = sB; sB = wS[at(seg + 1)];
                  moved = true;
               }
               else if (sNew < sA)
               {
                  if (seg <= 0) break;
                  --seg; sB = sA; sA = wS[at(seg)];
                  moved = true;
               }
               if (!moved) break;
            }
            if (kAt != seg)
            {
               at(seg + 1);
               const int k0 = at(seg), k1 = k0 + 1; // (whichever of the two refilled, RS_BACK / RS_WIN keep the neighbour inside)
               kk = coeffs_from_sol(wM[c][k0], wM[c][k1], wX[c][k0], wX[c][k1]);
               kAt = seg;
            }
            const double tau = (sNew - sA) / (sB - sA);
            const double tau2 = tau * tau, tau3 = tau2 * tau;
            if (newPt >= cap) { if (lane == 0) pp.status |= RS_CAPACITY; return; }
            const double v = kk.c3 * tau3 + kk.c2 * tau2 + kk.c1 * tau + kk.c0;
            if (evalCh) cartpt = v; // (joint channels: simply the new value)
            prev = cartpt;
            if (owner) out[(int64_t)newPt * C] = prev;
            oldPt = seg + 1;
            ++newPt;
         }
      }
      else if (oldPt == n - 1)
      {
         done = true;
      }
      else
      {
         prvDs = ds;
         sPrv = wS[at(oldPt)];
         ++oldPt;
      }
   }
   if (newPt >= cap) { if (lane == 0) pp.status |= RS_CAPACITY; return; }
   if (owner) out[(int64_t)newPt * C] = xc[n - 1]; // the original end point closes the path
   if (lane == 0)
   {
      pp.nOut = newPt + 1;
      if (pp.nOut < 4) pp.status |= RS_TOO_SHORT;
   }
}

// point-major rows [i][C] -> channel-major [C][n] of the next stage
__global__ void k_rs_transpose(int C, const RsPath *__restrict__ src, const RsPath *__restrict__ dst, int B, const double *__restrict__ rows,
                               double *__restrict__ x, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (dst[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const int i = (int)(g - dst[lo].off), n = dst[lo].n;
   if (i >= n) return;
   const double *r = rows + (src[lo].offOut + i) * C;
   double *o = x + dst[lo].off * C;
   for (int c = 0; c < C; ++c) o[(int64_t)c * n + i] = r[c];
}

// ---------------------------------------------------------------------------------------------
// evalSplineFullTraj N_old -> N_new (ba.cpp:790-863): values of every channel at the uniform sites
// sMVC[i] = sScale*i.  One lane per output site; the segment is the one the reference's monotone
// cursor ends on (first segment with site < sC[seg+1], clipped), found by bisection on sC.
// ---------------------------------------------------------------------------------------------
__global__ void k_rs_regular(RsParams P, const RsPath *__restrict__ src, const RsPath *__restrict__ dst, int B, const double *__restrict__ sC,
                             const double *__restrict__ x, const double *__restrict__ sol, double *__restrict__ y, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (dst[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const RsPath sp = src[lo];
   const int i = (int)(g - dst[lo].off), nNew = dst[lo].n, nOld = sp.n, C = P.C;
   if (i >= nNew || sp.status) return;
   const double *__restrict__ s = sC + sp.off;
   const double sScale = s[nOld - 1] / (double)(nNew - 1);
   const double site = sScale * (double)i;
   // seg = number of interior knots <= site, clipped to nOld-2
   int a = 0, b = nOld - 2; // invariant: answer in [a, b]
   while (a < b)
   {
      const int m = (a + b) >> 1;
      if (site < s[m + 1]) b = m; else a = m + 1;
   }
   const int seg = a;
   const double tau = (site - s[seg]) / (s[seg + 1] - s[seg]);
   const double tau2 = tau * tau, tau3 = tau2 * tau;
   const double *xs = x + sp.off * C + seg, *ms = sol + sp.off * C + seg;
   double *o = y + dst[lo].off * C;
   for (int c = 0; c < C; ++c)
   {
      const int64_t at = (int64_t)c * nOld;
      const Coef4 k = coeffs_from_sol(ms[at], ms[at + 1], xs[at], xs[at + 1]);
      o[(int64_t)c * nNew + i] = k.c3 * tau3 + k.c2 * tau2 + k.c1 * tau + k.c0;
   }
}

// ---------------------------------------------------------------------------------------------
// Forward kinematics: Robot::fwdKinKuka robot.cpp:105-174, Robot::fwdKinRR robot.cpp:188-202, called
// by the resampler after each of its passes (ba.cpp:626-628), before them when a Cartesian constraint is on (ba.cpp:247-256)
// and by the output stage (ba.cpp:1722-1725).  One lane per point.
//
// The trigonometry is a table when the caller wants bit parity with the host (BATOTP_F_HOST_TRIG: glibc sincos() of
// every joint angle, computed by the host side of the library between two kernels), else the device libm's sincos (last-bit
// differences against glibc: the knots then differ from the reference's in the last bits and the chaotic sweep may take a
// different number of steps -- documented tolerance mode).  Table layout per path: [rows][n] at trigBase + off*rows with
// KUKA: cos(t_k), k = 0..6, then sin(t_k); RR: cos(th1), cos(th1+th2), sin(th1), sin(th1+th2).
//
// 3x3 products in the summation order of the reference binary's Eigen (host robot.cpp): rows 0 and 1 of a product left to
// right, row 2 and the row-times-vector products a0 + (a1 + a2).
// ---------------------------------------------------------------------------------------------
constexpr double KIN_DEG2RAD = 3.14159265358979323846 / 180.0; // config.h:28

__host__ __device__ inline int fwdkin_trig_rows(int robot, int nJ)
{
   if (robot == BATOTP_ROBOT_KUKA && nJ == 7) return 14;
   if (robot == BATOTP_ROBOT_RR && nJ == 2) return 4;
   return 0;
}

__device__ __forceinline__ double kin_sum_seq(double a0, double a1, double a2) { return (a0 + a1) + a2; }
__device__ __forceinline__ double kin_sum_tree(double a0, double a1, double a2) { return a0 + (a1 + a2); }

__device__ __forceinline__ void kin_mul3(const double (&L)[3][3], const double (&R)[3][3], double (&out)[3][3])
{
#pragma unroll
   for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
      {
         const double t0 = L[r][0] * R[0][c], t1 = L[r][1] * R[1][c], t2 = L[r][2] * R[2][c];
         out[r][c] = (r == 2) ? kin_sum_tree(t0, t1, t2) : kin_sum_seq(t0, t1, t2);
      }
}

// tool point of the KUKA LWR IV+ from the cosines / sines of its seven joint angles
__device__ __forceinline__ void kuka_tool_point(const double (&c)[7], const double (&s)[7], double (&p)[3])
{
   const double c1 = c[0], c2 = c[1], c3 = c[2], c4 = c[3], c5 = c[4], c6 = c[5], c7 = c[6];
   const double s1 = s[0], s2 = s[1], s3 = s[2], s4 = s[3], s5 = s[4], s6 = s[5], s7 = s[6];
   const double Q12[3][3] = {{c1 * c2, -s1, -c1 * s2}, {c2 * s1, c1, -s1 * s2}, {s2, 0, c2}};
   const double Q34[3][3] = {{c3 * c4, -s3, c3 * s4}, {c4 * s3, c3, s3 * s4}, {-s4, 0, c4}};
   const double Q567[3][3] = {{c5 * c6 * c7 - s5 * s7, -c7 * s5 - c5 * c6 * s7, -c5 * s6},
                              {c5 * s7 + c6 * c7 * s5, c5 * c7 - c6 * s5 * s7, -s5 * s6},
                              {c7 * s6, -s6 * s7, c6}};
   double Q1234[3][3], Q[3][3];
   kin_mul3(Q12, Q34, Q1234);
   kin_mul3(Q1234, Q567, Q);
   const double tool0 = 0, tool1 = -.08, tool2 = .545;   // robot.cpp:113
   const double a0 = .3105, a1 = .4, a2 = .39;           // robot.cpp:117
   const double x1 = a1 * Q12[0][2], y1 = a1 * Q12[1][2], z1 = a1 * Q12[2][2] + a0;
   const double x2 = x1 + a2 * Q1234[0][2], y2 = y1 + a2 * Q1234[1][2], z2 = z1 + a2 * Q1234[2][2];
   p[0] = x2 + kin_sum_tree(Q[0][0] * tool0, Q[0][1] * tool1, Q[0][2] * tool2);
   p[1] = y2 + kin_sum_tree(Q[1][0] * tool0, Q[1][1] * tool1, Q[1][2] * tool2);
   p[2] = z2 + kin_sum_tree(Q[2][0] * tool0, Q[2][1] * tool1, Q[2][2] * tool2);
}

// Cartesian rows of point i of a path from its joint rows: th / ca point at row 0 of the joint / Cartesian rows, both with
// row stride n; trig = this path's table (row stride n) or nullptr for the device libm
__device__ __forceinline__ void fwdkin_point(int robot, const double *__restrict__ th, double *__restrict__ ca, const double *__restrict__ trig,
                                             int64_t n, int64_t i)
{
   if (robot == BATOTP_ROBOT_RR)
   {
      double c1, c12, s1, s12;
      if (trig) { c1 = trig[i]; c12 = trig[n + i]; s1 = trig[2 * n + i]; s12 = trig[3 * n + i]; }
      else
      {
         const double th1 = KIN_DEG2RAD * th[i], th2 = KIN_DEG2RAD * th[n + i];
         sincos(th1, &s1, &c1);
         sincos(th1 + th2, &s12, &c12);
      }
      const double a1 = .4, a2 = .6;
      ca[i] = a1 * c1 + a2 * c12;           // robot.cpp:198-199; the third row is only sized
      ca[n + i] = a1 * s1 + a2 * s12;
      return;
   }
   double c[7], s[7], p[3];
#pragma unroll
   for (int k = 0; k < 7; ++k)
   {
      if (trig) { c[k] = trig[k * n + i]; s[k] = trig[(7 + k) * n + i]; }
      else sincos(KIN_DEG2RAD * th[k * n + i], &s[k], &c[k]);
   }
   kuka_tool_point(c, s, p);
   ca[i] = p[0]; ca[n + i] = p[1]; ca[2 * n + i] = p[2];
}

__global__ void k_rs_fwdkin(RsParams P, const RsPath *__restrict__ paths, int B, double *__restrict__ x, const double *__restrict__ trig, int trigRows,
                            int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n || pp.status) return;
   double *__restrict__ xb = x + pp.off * P.C;
   fwdkin_point(P.robot, xb, xb + (int64_t)P.nJ * n, trig ? trig + pp.off * trigRows : nullptr, n, i);
}

// joint rows of every path of a stage, packed [nJ][n] per path at off*nJ (what the host needs for the trig tables)
__global__ void k_rs_pack_theta(RsParams P, const RsPath *__restrict__ paths, int B, const double *__restrict__ x, double *__restrict__ out, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n) return;
   for (int j = 0; j < P.nJ; ++j) out[pp.off * P.nJ + (int64_t)j * n + i] = x[pp.off * P.C + (int64_t)j * n + i];
}

// ---------------------------------------------------------------------------------------------
// Tool poses taught with the joints (path type BOTH, the UR5 example): BA::aa2qVect (ba.cpp:327-369) with aa2q
// (util.cpp:534-555) -- Cartesian rows 3..5 hold the orientation as axis-angle on entry, rows 3..6 the quaternion on return,
// successive quaternions kept on one hemisphere -- and, in the output stage, BA::q2aaVect (ba.cpp:384-403) with q2aa
// (util.cpp:562-581).  Trigonometry by the same policy as the forward kinematics: tables from the host libm
// (BATOTP_F_HOST_TRIG: sincos of the half angle, atan2) or the device libm.
//   k_rs_aa_norm   one lane per point: the rotation angle |aa| (what the host needs for its table), packed [n] per path at off
//   k_rs_aa2q      one lane per point: the quaternion of the point (before the hemisphere alignment)
//   k_rs_qalign    one lane per path: q_i <- -q_i where it points away from its (aligned) predecessor; sequential, exact
// ---------------------------------------------------------------------------------------------
__global__ void k_rs_aa_norm(RsParams P, const RsPath *__restrict__ paths, int B, const double *__restrict__ x, double *__restrict__ ang, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n) return;
   const double *__restrict__ r = x + pp.off * P.C + (int64_t)(P.nJ + 3) * n + i;
   const double a0 = r[0], a1 = r[n], a2 = r[2 * (int64_t)n];
   ang[g] = sqrt(a0 * a0 + a1 * a1 + a2 * a2);     // util.cpp:537
}

// trig: [2][n] per path at off*2 (sin, cos of half the angle) or nullptr
__global__ void k_rs_aa2q(RsParams P, const RsPath *__restrict__ paths, int B, double *__restrict__ x, const double *__restrict__ trig, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   const int lo = rs_find_path(paths, B, g);
   const RsPath pp = paths[lo];
   const int i = (int)(g - pp.off), n = pp.n;
   if (i >= n || pp.status) return;
   double *__restrict__ r = x + pp.off * P.C + (int64_t)(P.nJ + 3) * n + i;
   const double a0 = r[0], a1 = r[n], a2 = r[2 * (int64_t)n];
   const double theta = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
   double q0 = 1.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
   if (!(theta < 1e-6))
   {
      double sh, ch;
      if (trig) { sh = trig[pp.off * 2 + i]; ch = trig[pp.off * 2 + n + i]; }
      else sincos(0.5 * theta, &sh, &ch);
      q0 = ch;
      q1 = a0 * sh / theta; q2 = a1 * sh / theta; q3 = a2 * sh / theta;   // util.cpp:551: aa[i]*sin_half_theta/theta
   }
   r[0] = q0; r[n] = q1; r[2 * (int64_t)n] = q2; r[3 * (int64_t)n] = q3;
}

__global__ void k_rs_qalign(RsParams P, const RsPath *__restrict__ paths, int B, double *__restrict__ x)
{
   const int p = blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= B) return;
   const RsPath pp = paths[p];
   if (pp.status) return;
   const int n = pp.n;
   double *__restrict__ r = x + pp.off * P.C + (int64_t)(P.nJ + 3) * n;
   // qprev starts as the first point's own quaternion (ba.cpp:339-340): the first point is never flipped unless q.q < 0
   double p0 = r[0], p1 = r[n], p2 = r[2 * (int64_t)n], p3 = r[3 * (int64_t)n];
   for (int i = 0; i < n; ++i)
   {
      double q0 = r[i], q1 = r[n + i], q2 = r[2 * (int64_t)n + i], q3 = r[3 * (int64_t)n + i];
      double qdir = 0;
      qdir += q0 * p0; qdir += q1 * p1; qdir += q2 * p2; qdir += q3 * p3;   // ba.cpp:347-351
      if (qdir < 0.0)
      {
         q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3;
         r[i] = q0; r[n + i] = q1; r[2 * (int64_t)n + i] = q2; r[3 * (int64_t)n + i] = q3;
      }
      p0 = q0; p1 = q1; p2 = q2; p3 = q3;
   }
}

// zero the Cartesian channels of a stage (robot without kinematic model: ba.cpp:618-625)
__global__ void k_rs_zero_cart(RsParams P, const RsPath *__restrict__ paths, int B, double *__restrict__ x, int64_t total)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (paths[mid].off <= g) lo = mid; else hi = mid - 1;
   }
   const int i = (int)(g - paths[lo].off), n = paths[lo].n;
   if (i >= n) return;
   double *o = x + paths[lo].off * P.C;
   for (int c = 0; c < P.nC; ++c) o[(int64_t)(P.nJ + c) * n + i] = 0.0;
}

// Order-independent 64-bit checksum of a path's knots (batotp_hip_resampled_checksums): the wrap-around sum over all values of
// mix(bits of the value XOR a multiple of its index) -- addition commutes, so every decomposition gives the same number (the CPU
// checker computes it serially: tests compare the two).  One block column per path (blockIdx.y), a grid-stride loop, one atomic add
// per wavefront.
__device__ __forceinline__ unsigned long long rs_mix64(unsigned long long x)
{
   x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
   x ^= x >> 27; x *= 0x94D049BB133111EBull;
   x ^= x >> 31;
   return x;
}
__global__ void k_rs_checksum(const int64_t *__restrict__ off, const int64_t *__restrict__ n, int C, const double *__restrict__ y,
                              unsigned long long *__restrict__ out)
{
   const int p = blockIdx.y;
   const int64_t total = n[p] * C;
   const unsigned long long *__restrict__ v = reinterpret_cast<const unsigned long long *>(y + off[p] * C);
   unsigned long long h = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
      h += rs_mix64(v[i] ^ ((unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull));
   for (int o = 32; o; o >>= 1)
   {
      const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)h, o), hi = (unsigned)__shfl_xor((int)(unsigned)(h >> 32), o);
      h += ((unsigned long long)hi << 32) | lo;
   }
   if ((threadIdx.x & 63) == 0 && h) atomicAdd(out + p, h);
}

} // namespace bk
