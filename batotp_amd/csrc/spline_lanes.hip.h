// spline_lanes.hip.h -- the Thomas solve of Spline::solveTriDiagNatural (reference batotp/spline.cpp:252-276) of ONE long series
// on the 64 lanes of a wavefront, bit for bit (round 5; output stage: s(t) of a path and the channels to re-sample, a few hundred
// series of ~2e5 values that the lane-per-series kernel walked with a handful of lone wavefronts: 16.7 ms per launch).
//
// The principle is spline_tile.hip.h's: both recurrences of the (1,4,1) system are contractions (an error in d[i-1] reaches d[i]
// divided by the pivot, an error in sol[i+1] reaches sol[i] multiplied by c: x 0.268 either way), so a lane that starts SL_K = 64
// knots outside its chunk from the guess 0 arrives at the chunk with the true value's bits (0.268^64 = 2.6e-37 of what it started
// from) and from there on IS the sequential computation.  That is not assumed but CHECKED: the value a lane's warm-up arrives at
// for the knot before (behind) its chunk is compared bit for bit with the value the neighbouring lane computed for the same knot
// (registers, one shuffle); lane 0 starts from the true left boundary with the reference's not-yet-converged pivots, lane 63 from
// the true right one, so by induction a series whose 2 x 63 comparisons all agree is the sequential result exactly.  A series with
// a disagreement, or one shorter than SL_MIN, is flagged and solved by the sequential kernel afterwards (k_spline_series with the
// flags as a mask): the result is the sequential kernel's in every case.
//
// Decomposition: the eliminated right-hand sides d[1 .. n-1] are cut into 64 chunks of equal length (the last one takes the
// remainder), lane l owns chunk l.  As in the sequential kernel the d[i] are parked in the output array and overwritten by the
// second derivatives on the way back; a lane reads its neighbour's d (the warm-up of its back substitution) in its first SL_K
// steps and the neighbour overwrites them in its last ones -- chunks are at least 4 SL_K long.
#pragma once
#include "kernels.hip.h"

namespace bk
{

constexpr int SL_K = 64;                     // warm-up knots
constexpr int SL_CH = 8;                     // loads per batch and lane
constexpr int SL_MIN = 64 * 4 * SL_K + 2;    // shortest series (values) the lanes take

__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// y[i * ys], i = 0 .. N-1, are the values; sol[0 .. N-1] receives the second derivatives.  Called by all 64 lanes of a wavefront
// with the same arguments; N >= SL_MIN.  Returns (in every lane) whether all boundary comparisons agreed.
__device__ __forceinline__ bool thomas_series_lanes(int N, const double *__restrict__ y, int ys, double *sol, int lane)
{
   constexpr int CONV = 63;                    // c_ctab is constant from here on (checked by the host)
   const double cInf = c_ctab[CONV];
   const double denInf = 4.0 - 1.0 * cInf;
   const double rcpInf = c_ctab[0];            // RN(1/denInf), computed (and checked) by the host
   const int n = N - 1;
   const int Lc = (n - 1) / 64;                // d[1 .. n-1] in 64 chunks
   const int a = 1 + lane * Lc, b = lane == 63 ? n : a + Lc; // this lane stores d[a .. b) and sol[a .. b)
   const int a63 = 1 + 63 * Lc;

   // ---- forward elimination (spline.cpp:257-268) ----
   int i;
   double dprev, ym, y0, warm = 0;
   if (lane == 0)
   {
      dprev = (6 * (y[0] - 2 * y[ys] + y[2 * ys])) / 4.0;
      sol[1] = dprev;
      ym = y[ys]; y0 = y[2 * ys];
      for (i = 2; i <= CONV + 1; ++i) // the pivots that have not converged yet
      {
         const double yp = y[(int64_t)(i + 1) * ys];
         const double rhs = 6 * (ym - 2 * y0 + yp);
         const double den = (i - 1) < CONV ? (4.0 - 1.0 * c_ctab[i - 1]) : denInf;
         const double di = (rhs - 1.0 * dprev) / den;
         sol[i] = di;
         dprev = di;
         ym = y0; y0 = yp;
      }
   }
   else
   {
      i = a - SL_K;
      dprev = 0.0; // the guess for d[i - 1]
      ym = y[(int64_t)(i - 1) * ys]; y0 = y[(int64_t)i * ys];
   }
   const int stepsF = (n - a63) + SL_K; // the longest lane's (lane 63: the remainder is its)
   for (int base = 0; base < stepsF; base += SL_CH)
   {
      double yy[SL_CH];
#pragma unroll
      for (int k = 0; k < SL_CH; ++k)
      {
         int idx = i + 1 + k;
         idx = idx <= n ? idx : n;
         yy[k] = y[(int64_t)idx * ys];
      }
#pragma unroll
      for (int k = 0; k < SL_CH; ++k)
      {
         if (i < b)
         {
            const double rhs = 6 * (ym - 2 * y0 + yy[k]);
            const double di = div_by_const(rhs - 1.0 * dprev, denInf, rcpInf);
            if (i == a - 1) warm = di;
            if (i >= a) sol[i] = di;
            dprev = di;
            ym = y0; y0 = yy[k];
            ++i;
         }
      }
   }
   // d[a - 1] as this lane's warm-up found it against the left neighbour's own last value
   const int hiL = __shfl_up(__double2hiint(dprev), 1), loL = __shfl_up(__double2loint(dprev), 1);
   bool ok = lane == 0 || same_bits(warm, __hiloint2double(hiL, loL));

   double sNext; // sol[j + 1]
   int j;
   if (lane == 63)
   {
      const double cl = (n - 1) < CONV ? c_ctab[n - 1] : cInf;
      sNext = (0.0 - 1.0 * dprev) / (4.0 - 1.0 * cl); // spline.cpp:269 (not forced to zero)
      sol[n] = sNext;
      j = n - 1;
   }
   else
   {
      j = b - 1 + SL_K;
      sNext = 0.0; // the guess for sol[j + 1]
   }
   __threadfence_block(); // every lane's d before any lane's read of them

   // ---- back substitution (spline.cpp:271-274): sol[j] = d[j] - c[j] * sol[j + 1] ----
   const int jLow = lane == 0 ? CONV : a;
   const int stepsB = max(n - a63, Lc + SL_K);
   double warmB = 0;
   for (int base = 0; base < stepsB; base += SL_CH)
   {
      double dd[SL_CH];
#pragma unroll
      for (int k = 0; k < SL_CH; ++k)
      {
         int idx = j - k;
         idx = idx >= 1 ? idx : 1;
         dd[k] = BK_PARK_RELOAD(&sol[idx]); // (kernels.hip.h: optionally an agent-scope load)
      }
#pragma unroll
      for (int k = 0; k < SL_CH; ++k)
      {
         if (j >= jLow)
         {
            const double s = dd[k] - cInf * sNext;
            if (j == b) warmB = s;
            if (j < b) sol[j] = s;
            sNext = s;
            --j;
         }
      }
   }
   // sol[b] as this lane's warm-up found it against the right neighbour's own last value
   const int hiR = __shfl_down(__double2hiint(sNext), 1), loR = __shfl_down(__double2loint(sNext), 1);
   ok = ok && (lane == 63 || same_bits(warmB, __hiloint2double(hiR, loR)));
   if (lane == 0)
   {
      for (j = CONV - 1; j >= 1; --j) // the rows whose c[j] has not converged yet
      {
         const double s = BK_PARK_RELOAD(&sol[j]) - c_ctab[j] * sNext;
         sol[j] = s;
         sNext = s;
      }
      sol[0] = 0.0;
   }
   return __all(ok);
}

// natural-spline second derivatives of arbitrary series (k_spline_series' interface), one wavefront per series; flag[k] = 1:
// series k is left to k_spline_series (shorter than SL_MIN, or a boundary comparison failed -- never observed)
__global__ void __launch_bounds__(64) k_spline_series_lanes(int count, const int64_t *__restrict__ yOff, const int64_t *__restrict__ solOff,
                                                            const int *__restrict__ n, const double *__restrict__ y, int ys, double *sol,
                                                            int *__restrict__ flag)
{
   const int k = blockIdx.x;
   if (k >= count) return;
   const int N = n[k];
   bool redo = N >= 4; // (k_spline_series skips shorter ones)
   if (N >= SL_MIN) redo = !thomas_series_lanes(N, y + yOff[k], ys, sol + solOff[k], threadIdx.x);
   if (threadIdx.x == 0) flag[k] = redo ? 1 : 0;
}

// the resampler's spline builds (k_spline_sol's layout: channel-major values of path p at src + koff * C, series t = p * C + c),
// one wavefront per series; flag[t] = 1: series t is left to k_spline_sol
__global__ void __launch_bounds__(64) k_spline_sol_lanes(const PathInfo *__restrict__ pinfo, int B, int C, const double *__restrict__ src, double *sol,
                                                         int *__restrict__ flag)
{
   const int t = blockIdx.x;
   if (t >= B * C) return;
   const int p = t / C, c = t - p * C;
   const PathInfo pi = pinfo[p];
   const int N = (int)pi.n;
   const int64_t off = pi.koff * C + (int64_t)c * N;
   bool redo = true;
   if (N >= SL_MIN) redo = !thomas_series_lanes(N, src + off, 1, sol + off, threadIdx.x);
   if (threadIdx.x == 0) flag[t] = redo ? 1 : 0;
}

} // namespace bk
