// spline_tile.hip.h -- K1 in tiles of knots: the Thomas solve of Spline::solveTriDiagNatural (reference batotp/spline.cpp:252-276)
// parallel ALONG a series, bit for bit.
//
// The recurrences of the (1,4,1) system are sequential in the knot index and floating-point addition / division are not
// associative, so they cannot become a scan.  But both are contractions: an error in d[i-1] reaches d[i] divided by the pivot
// (x 0.268), an error in sol[i+1] reaches sol[i] multiplied by c (x 0.268).  A lane that starts the forward elimination K = 64
// knots BEFORE the chunk it is responsible for, from the guess 0, arrives at the chunk with a value whose distance from the true
// one has shrunk by 0.268^64 = 2.6e-37 -- twenty orders of magnitude below half an ulp of the value it started from -- i.e.,
// unless the series' curvature drops by more than those twenty orders within 64 knots (a kink followed by an exactly straight
// stretch does that: the tail of the kink's influence then IS the value, and a warm-up from 0 misses it), with the true
// value's bits; from there on it IS the sequential computation.  "With the true value's bits" is not assumed but CHECKED: every warm-up
// value is compared bit for bit with the value the neighbouring chunk computed for the same knot (inside a tile through LDS,
// between tiles through two doubles per series and tile in `edge`); the first chunk of a series starts from the true
// boundary, so by induction a series whose comparisons all agree is the sequential result exactly.  A series with a
// disagreement (none has been observed) is marked dirty and recomputed by the sequential kernel (k_spline / k_spline_pairs
// with a series mask) -- the result is identical to the sequential kernel's in every case, which is what the parity tests
// compare.  The same holds for the back substitution, run from K knots beyond the chunk.
//
// Decomposition: block = one tile of ST_T knots of one path x up to 8 channels.  The tile's values (plus K + 1 knots of halo
// on either side) are loaded into LDS by one coalesced copy, the forward chunks (ST_L knots per lane, all channels side by
// side) leave the eliminated right-hand sides d in LDS, the backward chunks the second derivatives, and one coalesced copy
// writes the result: (value, second derivative) pairs in place (compact layout: the pair array is read once and written
// once, whole lines both ways), or the coefficient rows c0..c3 (spline.cpp:203-209).  Against the lane-per-series kernel this
// removes the elimination scratch from HBM (it was written and read back as half-used lines: 4.2x the algorithmic bytes in
// round 2's PMC run) and gives a single long series N / ST_L lanes instead of one.
#pragma once
#include "kernels.hip.h"

namespace bk
{

constexpr int ST_T = 256;                       // knots per tile (the last tile of a series also takes the last knot: up to ST_T + 1)
constexpr int ST_K = 64;                        // warm-up knots
constexpr int ST_L = 16;                        // knots per lane and pass
constexpr int ST_CH = 8;                        // channels per block
constexpr int ST_SPAN = ST_T + 2 * ST_K + 3;    // knots in LDS: [t0 - K - 1, t1 + K + 1)
constexpr int ST_SPANP = ST_SPAN + 1;           // (odd row length: the channels of a knot fall into different banks)
constexpr int ST_BLOCK = 256;
constexpr int ST_MIN_KNOTS = 1024;              // shorter series take the sequential kernel

struct TileArgs
{
   const PathInfo *pinfo;
   const int *tileOff;      // [B + 1] prefix sums of the tiles per path (0 tiles for paths below ST_MIN_KNOTS)
   int B;
   int mode;                // 0: input channels (device channel = c), 1: dynamics channels (device channel Cin + r*4 + k)
   int pairs;               // 1: compact layout (km in place), 0: channel-major rows in, coefficient rows out
   int nch;                 // series per path in this launch
   int C, Cin, d;
   int c0;                  // pairs: first of the launch's nch series among the C channels of a knot
   const double *src;       // rows: channel-major values, path at koff*srcStride
   int64_t srcStride;
   double *km;              // pairs: [N][C][2] per path at koff*C*2
   double *coef;            // rows: [N][C][4] per path at koff*C*4
   double *edge;            // [(tileOff[p] + tile) * nch + e][4]: warm d, true d, warm sol, true sol at the tile's two boundaries
   int *dirty;              // [B * nch] series the sequential kernel has to redo
};

// enumeration of a launch's series such that a block's channels are neighbours in the coefficient rows: e -> source row
__device__ __forceinline__ int tile_src_channel(const TileArgs &a, int e) { return a.mode == 0 ? e : (e & 3) * a.d + (e >> 2); }
__device__ __forceinline__ int tile_dev_channel(const TileArgs &a, int e) { return a.mode == 0 ? e : a.Cin + e; }

__global__ void __launch_bounds__(ST_BLOCK) k_spline_tile(TileArgs a)
{
   __shared__ double Y[ST_CH][ST_SPANP];     // knot values
   __shared__ double D[ST_CH][ST_SPANP];     // eliminated right-hand sides d[i]
   __shared__ double S[ST_CH][ST_T + 3];     // second derivatives sol[t0 .. t1]
   __shared__ double warmF[ST_CH][(ST_T + ST_K) / ST_L + 3], warmB[ST_CH][ST_T / ST_L + 3];
   __shared__ int bad;
   // ~69 KB of static LDS: this translation unit is for gfx950 (160 KB of LDS per CU: two tiles per CU) and does not fit an
   // architecture with 64 KB per workgroup
   static_assert(sizeof(double) * (2 * ST_CH * ST_SPANP + ST_CH * (ST_T + 3) + ST_CH * ((ST_T + ST_K) / ST_L + 3) + ST_CH * (ST_T / ST_L + 3)) <= 96 * 1024,
                 "k_spline_tile: LDS budget of gfx950");

   // which tile
   int lo = 0, hi = a.B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (a.tileOff[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
   }
   const int p = lo, tile = (int)blockIdx.x - a.tileOff[p];
   const PathInfo pi = a.pinfo[p];
   const int N = (int)pi.n, n = N - 1;
   const int e0 = blockIdx.y * ST_CH;                           // first series of this block
   const int nc = (a.nch - e0) < ST_CH ? (a.nch - e0) : ST_CH;  // series of this block
   // knots [t0, t1) are this tile's; the last tile of the series ends at N (ceil((N - 1) / ST_T) tiles: the last knot never
   // gets a tile of its own -- its second derivative needs d[n - 1], which is its predecessor's)
   const int t0 = tile * ST_T, t1 = (tile + 1 == a.tileOff[p + 1] - a.tileOff[p]) ? N : (t0 + ST_T);
   const int k0 = (t0 - ST_K - 1) > 0 ? (t0 - ST_K - 1) : 0;
   const int k1 = (t1 + ST_K + 1) < N ? (t1 + ST_K + 1) : N;    // knots [k0, k1) are in LDS
   const int tid = threadIdx.x;
   if (tid == 0) bad = 0;

   // ---- load ----
   if (a.pairs)
   {
      const double2 *__restrict__ g = reinterpret_cast<const double2 *>(a.km) + pi.koff * a.C;
      const int cnt = (k1 - k0) * a.C;
      for (int x = tid; x < cnt; x += ST_BLOCK)
      {
         const int kk = x / a.C, c = x - kk * a.C;
         if (c >= a.c0 + e0 && c < a.c0 + e0 + nc) Y[c - a.c0 - e0][kk] = g[(int64_t)(k0 + kk) * a.C + c].x;
      }
   }
   else
   {
      const int len = k1 - k0;
      for (int x = tid; x < len * nc; x += ST_BLOCK)
      {
         const int cl = x / len, kk = x - cl * len;
         Y[cl][kk] = a.src[pi.koff * a.srcStride + (int64_t)tile_src_channel(a, e0 + cl) * N + k0 + kk];
      }
   }
   __syncthreads();

   constexpr int CONV = 63;                    // c_ctab is constant from here on (checked by the host)
   const double cInf = c_ctab[CONV];
   const double denInf = 4.0 - 1.0 * cInf;
   const double rcpInf = c_ctab[0];

   // ---- forward elimination (spline.cpp:259-268): chunks of ST_L knots over [t0, min(t1 + K, n)) ----
   const int fEnd = (t1 + ST_K) < n ? (t1 + ST_K) : n;    // d[i] exists for 1 <= i <= n - 1
   const int nF = (fEnd - t0 + ST_L - 1) / ST_L;
   {
      const int cl = tid % ST_CH, q = tid / ST_CH;
      if (cl < nc && q < nF)
      {
         const int ca = t0 + q * ST_L, cb = (ca + ST_L) < fEnd ? (ca + ST_L) : fEnd;   // this lane stores d[ca .. cb)
         const double *__restrict__ y = Y[cl] - k0;                                      // y[i] = value of knot i
         double *__restrict__ dd = D[cl] - k0;
         int i = ca - ST_K;
         double dprev;
         bool exact = false;
         if (i <= 2)
         {
            dprev = (6 * (y[0] - 2 * y[1] + y[2])) / 4.0;       // d[1], spline.cpp:257-258 with c[1] = 1/4
            if (ca <= 1 && 1 < cb) dd[1] = dprev;
            i = 2;
            exact = true;
         }
         else dprev = 0.0;                                       // guess for d[i - 1]; K steps of contraction follow
         for (; i < cb; ++i)
         {
            if (i == ca) warmF[cl][q] = exact ? 0.0 : dprev;     // the value the warm-up arrived at for d[ca - 1]
            const double rhs = 6 * (y[i - 1] - 2 * y[i] + y[i + 1]);
            const double num = rhs - 1.0 * dprev;
            double di;
            if (i - 1 < CONV) di = num / (4.0 - 1.0 * c_ctab[i - 1]);
            else di = div_by_const(num, denInf, rcpInf);
            if (i >= ca) dd[i] = di;
            dprev = di;
         }
      }
   }
   __syncthreads();
   // the warm-up values against the neighbours' true values (chunk q against chunk q - 1); chunk 0 against the previous tile
   {
      const int cl = tid % ST_CH, q = tid / ST_CH;
      if (cl < nc && q < nF)
      {
         const int ca = t0 + q * ST_L;
         if (ca - ST_K > 2)
         {
            const double w = warmF[cl][q];
            if (q > 0) { if (__double_as_longlong(w) != __double_as_longlong(D[cl][ca - 1 - k0])) bad = 1; }
            else a.edge[((int64_t)(a.tileOff[p] + tile) * a.nch + e0 + cl) * 4 + 0] = w;
         }
         // this tile's true d at the next tile's boundary
         if (q == 0 && t1 < N && t1 - 1 >= 1 && t1 - 1 < fEnd)
            a.edge[((int64_t)(a.tileOff[p] + tile + 1) * a.nch + e0 + cl) * 4 + 1] = D[cl][t1 - 1 - k0];
      }
   }

   // ---- back substitution (spline.cpp:269-274): chunks of ST_L second derivatives over [t0, t1) ----
   const int nB = (t1 - t0 + ST_L - 1) / ST_L;
   {
      const int cl = tid % ST_CH, q = tid / ST_CH;
      if (cl < nc && q < nB)
      {
         const int ca = t0 + q * ST_L, cb = (ca + ST_L) < t1 ? (ca + ST_L) : t1;     // this lane stores sol[ca .. cb)
         const double *__restrict__ dd = D[cl] - k0;
         double *__restrict__ ss = S[cl] - t0;
         // entry value: sol[cb] (or, for the chunk that holds the last knot, sol[n] itself)
         int i;          // s holds sol[i]
         double s;
         const double cl1 = (n - 1) < CONV ? c_ctab[n - 1] : cInf;
         const bool last = cb > n;                              // the chunk contains knot n
         const int e = (cb + ST_K) < n ? (cb + ST_K) : n;
         bool exact = false;
         if (last || e == n)
         {
            s = (0.0 - 1.0 * dd[n - 1]) / (4.0 - 1.0 * cl1);    // spline.cpp:269: eliminated once more, not forced to zero
            i = n;
            exact = true;
            if (last) ss[n] = s;
         }
         else { s = 0.0; i = e; }                               // guess for sol[e]
         const int stop = last ? n : cb;
         for (; i > stop; --i)                                  // warm-up: down to sol[cb]
         {
            const double ci = (i - 1) < CONV ? c_ctab[i - 1] : cInf;
            s = dd[i - 1] - ci * s;
         }
         if (!last)
         {
            warmB[cl][q] = exact ? 0.0 : s;
            if (cb == t1) ss[t1] = s;                           // the next tile's first value (its true value is compared below)
         }
         for (; i > ca; --i)                                    // sol[i - 1] from sol[i]
         {
            double v;
            if (i - 1 >= 1)
            {
               const double ci = (i - 1) < CONV ? c_ctab[i - 1] : cInf;
               v = dd[i - 1] - ci * s;
            }
            else v = 0.0;                                       // sol[0] = 0 (natural left end)
            ss[i - 1] = v;
            s = v;
         }
      }
   }
   __syncthreads();
   {
      const int cl = tid % ST_CH, q = tid / ST_CH;
      if (cl < nc && q < nB)
      {
         const int ca = t0 + q * ST_L, cb = (ca + ST_L) < t1 ? (ca + ST_L) : t1;
         const bool last = cb > n;
         const int e = (cb + ST_K) < n ? (cb + ST_K) : n;
         if (!last && e != n)
         {
            const double w = warmB[cl][q];
            if (cb < t1) { if (__double_as_longlong(w) != __double_as_longlong(S[cl][cb - t0])) bad = 1; }
            else a.edge[((int64_t)(a.tileOff[p] + tile) * a.nch + e0 + cl) * 4 + 2] = w;
         }
         else if (!last && cb == t1) a.edge[((int64_t)(a.tileOff[p] + tile) * a.nch + e0 + cl) * 4 + 2] = S[cl][t1 - t0];
         if (q == 0 && tile > 0) a.edge[((int64_t)(a.tileOff[p] + tile - 1) * a.nch + e0 + cl) * 4 + 3] = S[cl][0];
      }
   }
   __syncthreads();
   if (bad)
   {
      // (never observed) a chunk inside this tile did not arrive at its neighbour's value: every series of the block goes to
      // the sequential kernel; nothing is stored
      if (tid < nc) a.dirty[p * a.nch + tile_src_channel(a, e0 + tid)] = 1;
      return;
   }

   // ---- store ----
   if (a.pairs)
   {
      double2 *__restrict__ g = reinterpret_cast<double2 *>(a.km) + pi.koff * a.C;
      const int cnt = (t1 - t0) * a.C;
      for (int x = tid; x < cnt; x += ST_BLOCK)
      {
         const int kk = x / a.C, c = x - kk * a.C;
         // In place: a neighbouring block of this launch may still be LOADING the .x of these knots as its halo while this
         // store runs.  Invariant that makes that harmless: the stored .x is bit-identical to the value loaded (Y is never
         // modified), so a reader sees the same 8 bytes before, during and after the store; .y is only read by later kernels.
         if (c >= a.c0 + e0 && c < a.c0 + e0 + nc) g[(int64_t)(t0 + kk) * a.C + c] = make_double2(Y[c - a.c0 - e0][t0 + kk - k0], S[c - a.c0 - e0][kk]);
      }
   }
   else
   {
      double *__restrict__ cf = a.coef + pi.koff * a.C * 4;
      const int cnt = (t1 - t0) * nc;
      for (int x = tid; x < cnt; x += ST_BLOCK)
      {
         const int kk = x / nc, cl = x - kk * nc, k = t0 + kk;
         const int dc = tile_dev_channel(a, e0 + cl);
         Coef4 o;
         if (k < n)
         {
            const double solL = S[cl][kk], solR = S[cl][kk + 1], yL = Y[cl][k - k0], yR = Y[cl][k + 1 - k0];
            o.c3 = div6(solR - solL);            // spline.cpp:203-209 (x / 6 through the exact reciprocal form, kernels.hip.h)
            o.c2 = solL / 2.0;
            o.c1 = yR - yL - div6(solR + 2 * solL);
            o.c0 = yL;
         }
         else { o.c0 = 0; o.c1 = 0; o.c2 = 0; o.c3 = 0; }       // the row of the last knot is never written (stays zero)
         *reinterpret_cast<Coef4 *>(cf + ((int64_t)k * a.C + dc) * 4) = o;
      }
   }
}

// series the sequential kernel has to take from the start: those of paths below ST_MIN_KNOTS knots
__global__ void k_tile_dirty_init(const PathInfo *__restrict__ pinfo, int B, int nch, int *__restrict__ dirty)
{
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= B * nch) return;
   dirty[t] = pinfo[t / nch].n < ST_MIN_KNOTS ? 1 : 0;
}

// the boundary values between tiles: slot (tile, series) holds [0] the warm-up d the tile arrived at for knot t0 - 1, [1] the
// previous tile's true d there, [2] the warm-up sol the tile arrived at for knot t1, [3] the next tile's true sol there
__global__ void k_spline_tile_check(TileArgs a, int totalTiles)
{
   const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= (int64_t)totalTiles * a.nch) return;
   const int gt = (int)(g / a.nch), e = (int)(g - (int64_t)gt * a.nch);
   int lo = 0, hi = a.B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (a.tileOff[mid] <= gt) lo = mid; else hi = mid - 1;
   }
   const int p = lo, tile = gt - a.tileOff[p], tiles = a.tileOff[p + 1] - a.tileOff[p];
   const double *__restrict__ ed = a.edge + g * 4;
   bool ok = true;
   if (tile > 0) ok = ok && __double_as_longlong(ed[0]) == __double_as_longlong(ed[1]);
   if (tile + 1 < tiles) ok = ok && __double_as_longlong(ed[2]) == __double_as_longlong(ed[3]);
   if (!ok) a.dirty[p * a.nch + tile_src_channel(a, e)] = 1;
}

} // namespace bk
