// spline_stream.hip.h -- K1 for large batches of (value, second derivative) pairs in ONE pass over HBM (round 4).
//
// k_spline_pairs (kernels.hip.h) runs the Thomas solve of Spline::solveTriDiagNatural (reference batotp/spline.cpp:252-276) as two
// sweeps over a series: the forward elimination parks its values d[i] in the pairs' second slot, the back substitution reads them
// again -- every line of the pair array crosses HBM four times (791 GB for 182 GB of pairs at the headline batch, 150 ms at the
// memory system's rate).  Here a lane still owns one series and runs the forward elimination sequentially and exactly, but it keeps
// only the last SS_R = 64 values d[i] -- in an LDS ring -- and back-substitutes in blocks of SS_T = 16 knots that start SS_W = 48 knots
// AHEAD of the block, from the guess 0: the back substitution is a contraction (an error in sol[i+1] reaches sol[i] multiplied by
// c = 0.268), so after 48 steps the guess has shrunk by 3e-28 of the value's own size -- eleven orders of magnitude below half an
// ulp -- and the recurrence continues with the true value's bits.  As in spline_tile.hip.h that is not assumed but CHECKED: the
// value a block arrives at for its upper boundary is compared bit for bit with the value the next block computes for the same knot
// (which has 16 more steps of contraction behind it; the last block starts from the true end condition, so by induction a series
// whose comparisons all agree is the sequential result exactly).  A series with a disagreement (or shorter than ST_MIN_KNOTS) is
// marked in `dirty` and solved by k_spline_pairs afterwards.  Values are read once (the .x slots) and second derivatives written
// once (the .y slots): two crossings per line instead of four.
#pragma once
#include "kernels.hip.h"
#include "spline_tile.hip.h"

namespace bk
{

constexpr int SS_T = 16;          // knots per block of the back substitution
constexpr int SS_W = 48;          // warm-up knots
constexpr int SS_R = SS_T + SS_W; // ring entries per lane (a power of two)
static_assert((SS_R & (SS_R - 1)) == 0, "ring index by mask");

// series c0 .. c0 + nch - 1 of the C channels per knot; dirty[p * nch + c]: 1 on entry = leave to the sequential kernel
__global__ void __launch_bounds__(64) k_spline_pairs_stream(const PathInfo *__restrict__ pinfo, int B, int nch, int c0, int C, double *km,
                                                            int *__restrict__ dirty)
{
   __shared__ double ring[SS_R][64]; // d[i] of lane l at ring[i & (SS_R - 1)][l]: conflict-free, 32 KB per wavefront
   const int lane = threadIdx.x;
   const int t = blockIdx.x * 64 + lane;
   if (t >= B * nch) return;
   if (dirty[t]) return;
   const int p = t / nch, c = t - p * nch;
   const PathInfo pi = pinfo[p];
   const int N = (int)pi.n, n = N - 1;
   // (8-byte accesses on purpose: a value slot is only ever read, a second-derivative slot only ever written, by this lane alone)
   const double *yv = km + (pi.koff * C + (c0 + c)) * 2; // knot i: yv[i * 2 * C]
   double *sol = km + (pi.koff * C + (c0 + c)) * 2 + 1;  // knot i: sol[i * 2 * C]
   const int ys = 2 * C, ss = 2 * C;

   constexpr int CONV = 63;                    // c_ctab is constant from here on (checked by the host)
   const double cInf = c_ctab[CONV];
   const double denInf = 4.0 - 1.0 * cInf;
   const double rcpInf = c_ctab[0];

   // forward elimination state (spline.cpp:257-268): d[1 .. F] are done, ym = y[F], y0 = y[F + 1]
   double dprev = (6 * (yv[0] - 2 * yv[ys] + yv[2 * ys])) / 4.0; // d[1] with c[1] = 1/4
   ring[1][lane] = dprev;
   double ym = yv[ys], y0 = yv[2 * ys];
   int F = 1;
   // advance the elimination to d[upto] (upto <= n - 1), SS_T values per round with their loads issued together
   auto forwardTo = [&](int upto) {
      while (F < upto)
      {
         const int cnt = (upto - F) < SS_T ? (upto - F) : SS_T;
         double yy[SS_T];
#pragma unroll
         for (int k = 0; k < SS_T; ++k)
         {
            int i = F + 2 + k;          // d[F + 1 + k] needs y[F + 2 + k]
            i = i <= n ? i : n;
            yy[k] = yv[(unsigned)(i * ys)];
         }
         if (cnt == SS_T && F >= CONV)
         {
            // the common round: sixteen steps with the converged pivot, nothing to test
#pragma unroll
            for (int k = 0; k < SS_T; ++k)
            {
               const double rhs = 6 * (ym - 2 * y0 + yy[k]);
               const double di = div_by_const(rhs - 1.0 * dprev, denInf, rcpInf);
               ring[(F + 1 + k) & (SS_R - 1)][lane] = di;
               dprev = di;
               ym = y0; y0 = yy[k];
            }
            F += SS_T;
            continue;
         }
#pragma unroll
         for (int k = 0; k < SS_T; ++k)
         {
            if (k < cnt)
            {
               const int i = F + 1 + k;
               const double rhs = 6 * (ym - 2 * y0 + yy[k]);
               const double num = rhs - 1.0 * dprev;
               double di;
               if (i - 1 < CONV) di = num / (4.0 - 1.0 * c_ctab[i - 1]);
               else di = div_by_const(num, denInf, rcpInf);
               ring[i & (SS_R - 1)][lane] = di;
               dprev = di;
               ym = y0; y0 = yy[k];
            }
         }
         F += cnt;
      }
   };

   const double cl1 = (n - 1) < CONV ? c_ctab[n - 1] : cInf;
   double warmPrev = 0.0; // what the previous block arrived at for sol[a] of this block
   bool havePrev = false, bad = false;
   for (int a = 0; a < N; a += SS_T)
   {
      const int b = (a + SS_T) < N ? (a + SS_T) : N;   // this block stores sol[a .. b)
      const bool last = (b + SS_W) >= n;                // the back substitution starts from the true end
      const int e = last ? n : (b + SS_W);              // ... at sol[e]
      forwardTo(e - 1);                                 // d[.. e - 1] (d exists for 1 <= i <= n - 1, and e <= n)
      if (!last && a > CONV && b - a == SS_T)
      {
         // the common block: SS_W warm-up steps and SS_T stored ones with the converged multiplier, the ring read sixteen values
         // at a time (the reads do not depend on the recurrence: one LDS round trip per sixteen steps)
         double s = 0.0, warmHere = 0.0, first = 0.0;
#pragma unroll
         for (int g = 0; g < SS_R / 16; ++g)
         {
            double dd[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) dd[k] = ring[(e - 1 - g * 16 - k) & (SS_R - 1)][lane];
#pragma unroll
            for (int k = 0; k < 16; ++k)
            {
               s = dd[k] - cInf * s;                                  // sol[e - 1 - g * 16 - k]
               if (g * 16 + k == SS_W - 1) warmHere = s;              // sol[b]
               if (g * 16 + k >= SS_W) sol[(unsigned)((e - 1 - g * 16 - k) * ss)] = s;
            }
         }
         first = s;                                                   // sol[a]
         if (havePrev && __double_as_longlong(warmPrev) != __double_as_longlong(first)) bad = true;
         warmPrev = warmHere;
         havePrev = true;
         continue;
      }
      double s;
      if (last) s = (0.0 - 1.0 * ring[(n - 1) & (SS_R - 1)][lane]) / (4.0 - 1.0 * cl1); // spline.cpp:269: not forced to zero
      else s = 0.0;                                                                      // the guess for sol[e]
      int i = e;                                        // s holds sol[i]
      if (last && n >= a && n < b) sol[(unsigned)(n * ss)] = s;
      // warm-up: down to sol[b] (nothing to do when the block holds the last knot)
      for (; i > b; --i)
      {
         const double ci = (i - 1) < CONV ? c_ctab[i - 1] : cInf;
         s = ring[(i - 1) & (SS_R - 1)][lane] - ci * s;
      }
      // (i == b, or i == n < b for the last block)
      const double warmHere = s; // sol[b] as this block knows it: the NEXT block's value for the same knot is compared with it
      // the block's values, highest index first
      double first = 0.0; // sol[a]
      for (; i > a; --i)
      {
         double v;
         if (i - 1 >= 1)
         {
            const double ci = (i - 1) < CONV ? c_ctab[i - 1] : cInf;
            v = ring[(i - 1) & (SS_R - 1)][lane] - ci * s;
         }
         else v = 0.0; // sol[0] = 0 (natural left end)
         sol[(unsigned)((i - 1) * ss)] = v;
         s = v;
         first = v;
      }
      // the previous block's warm value for sol[a] against this block's value
      if (havePrev && __double_as_longlong(warmPrev) != __double_as_longlong(first)) bad = true;
      warmPrev = warmHere;
      havePrev = !last; // (a block that started from the true end leaves nothing to compare: there is no next block past the end)
      if (last && b >= N) break;
   }
   if (bad) dirty[t] = 1; // (never observed) the sequential kernel solves this series again, in place
}

} // namespace bk
