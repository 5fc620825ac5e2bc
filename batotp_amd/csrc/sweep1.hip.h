// sweep1.hip.h -- the sweep of a path that has a wavefront to itself (the latency-bound regime: BASELINE configs 2, 3 and 4
// as worded -- one trajectory, or up to ~2 k paths per GPU, which is also one GPU's share of config 5), on uniform knot sites,
// for every constraint family except the torque limits of a parallel mechanism that was not converted (isPar2Ser = 0).
// Same arithmetic, same order, same results as k_sweep (kernels.hip.h) -- BA::sweep and everything it calls, reference
// batotp/ba.cpp:979-1195, 1204-1236, 1248-1332, 1341-1439, 1449-1581, 1590-1652 -- written for the fewest instructions per
// stage instead of for generality:
//   * a lone wavefront issues a vector instruction every 4-8 cycles whatever its number of active lanes (MI355X_MICROARCH.md;
//     tools/micro/half_wave_issue.hip: an fp64 FMA 5.4-5.7 cycles with 8, 32 or 64 active lanes), and the scalar instructions
//     of divergent control flow cost as much: k_sweep's 405 VALU + 250 SALU instructions per stage evaluation are the 5170
//     cycles measured at B = 1 (profiles/r02_c_*).  So: the six stages are straight-line code with their tableau column as
//     literals (no selects on the stage number, no table look-ups), one joint per lane without per-lane loops, the constraint
//     families that are off compiled out, the stage's spline row and the knot sites of its segment kept in registers,
//     conditions on path-level values as scalar branches (S1_UNI), and the bisection -- a quarter of the stages need it --
//     either fast-forwarded (S1_PREDICT) or evaluating four candidates of the reference's (deterministic) candidate sequence
//     per pass, as the 32-lane layout of k_sweep does;
//   * lanes: lane = slot * 8 + joint, 4 candidate slots x 8 joint lanes; the other 32 lanes of the wavefront exit.
#pragma once
#include "kernels.hip.h"

namespace bk
{

// ratio_lt (kernels.hip.h) for values that are the same in every lane of the wavefront: (num / den) < thr decided exactly as
// the correctly rounded division would decide it; the division itself is only performed -- behind a scalar branch -- when
// num / den lies within 1e-14 (relative) of the threshold or the shortcut's preconditions do not hold
__device__ __forceinline__ bool ratio_lt_uniform(double num, double den, double thr)
{
   const double p = thr * den;
   const bool pre = den > 0.0 && num >= 0.0 && p > 1e-290 && p < 1e290;
   const bool sureLess = pre && num < p * (1.0 - 1e-14);
   const bool sureNotLess = pre && num > p * (1.0 + 1e-14);
   bool res = sureLess;
   if (__builtin_amdgcn_readfirstlane((int)!(sureLess || sureNotLess))) res = num / den < thr;
   return res;
}

// Build variants measured in round 3 and not taken (profiles/r03_a_sweep1_code_size_ab.txt, r03_i_*; the code is in the history):
// the six stages of a step as one loop body (kernels of 23-38 KB instead of 54-92 KB: 14-17 % slower -- instruction fetch is not
// what the lone wavefront waits for); the per-lane conditions of the checks as selects (1-5 % slower); all 64 lanes with the
// upper bounds of a check in one half of a slot and the lower bounds in the other (one division per lane, k_sweep<16>'s layout:
// -2 % on the cable robot whose check has four divisions per lane, +3 % on the vel/acc problems -- LDS and memory instructions
// of 64 lanes take longer than those of 32).
// 1: a violated first check of a problem whose constraints are lines in sdot^2 goes through the CERTIFIED FAST-FORWARD of the
// bisection first (see accelPt): the speed at which the sddot intervals stop intersecting has a closed form there, and the
// iterations of the reference's loop whose outcome is certain given the check's rounding-error bound are taken without their
// checks (batotp_hip_set_fast_forward switches it off at run time; 0 here compiles it out)
#ifndef S1_PREDICT
#define S1_PREDICT 1
#endif
// tableau of ba.cpp:58-63 as a table: entry [6*k + m] = _B[k][m] (stage m + 1 combines the stage values k = 0..m)
__constant__ double c_s1B[36] = {BK_B00, BK_B01, BK_B02, BK_B03, BK_B04, BK_B05,
                                 0, BK_B11, BK_B12, BK_B13, BK_B14, BK_B15,
                                 0, 0, BK_B22, BK_B23, BK_B24, BK_B25,
                                 0, 0, 0, BK_B33, BK_B34, BK_B35,
                                 0, 0, 0, 0, BK_B44, BK_B45,
                                 0, 0, 0, 0, 0, BK_B55};

// A condition on path-level values, which are the same in every lane of the wavefront (one path per wavefront): evaluated as a
// ballot it becomes a scalar branch.  Written as a plain `if`, the compiler cannot know that the lanes agree and wraps the branch
// and every loop around it in exec-mask bookkeeping (a dozen scalar instructions per cursor-walk iteration) for a divergence that
// never happens.  NOT for conditions that differ between joints (lanes) or candidate slots.  Measured per use on BASELINE
// config 4 (1024 wavefronts, repeatable to 0.1 %): the reverse-curve walk gains most (forward sweep 438 -> 403 ms); the knot-cursor
// walk walkC is the exception -- its loop leaves at the first test nearly always, and there the ballot's round trip through the
// scalar unit costs more than the exec-mask form (449 / 403 ms plain against 469 / 420 ms): it keeps plain conditions.
#define S1_UNI(c) (NP == 1 ? (__ballot(c) != 0) : (bool)(c))
// NP == 2 (two paths per wavefront, one per half: see k_sweep1's template parameter): the ballot of a path's own 32 lanes, and
// "the value of the path's first lane" -- which every lane of the half holds anyway
#define S1_BALLOT(c) (NP == 1 ? (unsigned long long)__ballot(c) : ((unsigned long long)__ballot(c) >> (lane & 32)) & 0xffffffffull)
#define S1_FIRST(x) (NP == 1 ? __builtin_amdgcn_readfirstlane(x) : (x))
#define S1_RATIO_LT(n, d, t) (NP == 1 ? ratio_lt_uniform(n, d, t) : ratio_lt(n, d, t))

constexpr int S1_BLOCK = 256;
constexpr int S1_WK = 64;   // knots per spline window (compact splines: 64 knots x 8 joint slots x 16 B = 8 KB per path)
constexpr int S1_WM = 256;  // points per reverse-curve window (4 KB per path)
#ifndef S1_WR_BYTES
#define S1_WR_BYTES 12288   // coefficient rows (FEAT >= 0): LDS bytes per path for the window of consecutive rows (C x 32 B each: 21 rows of the
                            // cable robot's 18 channels, 10 of the 7-DOF arm's 38).  With the reverse-curve window of the forward sweep a
                            // block of four paths takes 64.4 KB: two blocks per CU (two wavefronts per SIMD) still fit the 160 KB
#endif

// FEAT: -1 = compact splines ((value, second derivative) pairs), 0 = coefficient rows, 1 = coefficient rows + the Cartesian
// speed / acceleration limits (ba.cpp:1225-1229, 1423-1439, 1535-1579), 2 = those + torque limits of a
// serial robot (a1..a4 splines, ba.cpp:1387-1405, 1495-1509; BASELINE config 3).  DIR: -1 reverse, +1 forward.
// FF (FEAT == 2 only): which form of the certified fast-forward of the bisection the instantiation carries -- 0: constraints that are
// lines in sdot^2 (the cable robot in serial form, whose a3 vanishes), 1: the general form for serial chains (a3 != 0).  One form
// per kernel: with both in one instantiation the cable robot's sweeps were 2 % slower for code they never run.
// PAIRS (FEAT >= 0 only): the batch keeps ALL its channels as (value, second derivative) pairs and coefficient rows exist in this
// kernel's LDS window only.  A template parameter, not a run-time test: the window's code and registers cost the lone wavefront of a
// single trajectory 10 % when they are merely present (BASELINE config 3: 1129 against 1018 ms).
// NP (round 5): paths per wavefront.  1: the path owns lanes 0..31, the other half exits.  2: a second path in lanes 32..63 -- the
// same code, with every condition on path-level values a per-lane condition (uniform inside a half, not across the wavefront: the
// compiler's exec-mask branches instead of scalar ones), ballots cut to the path's own half, and LDS windows of half the size so
// that a CU still holds two workgroups: where the two paths agree on the control flow (three quarters of the stages pass their
// first check) one instruction stream serves both; where one bisects the other waits.  For batches that put MORE than two paths
// on a SIMD (BASELINE config 5 on one GPU: 4096 cable-robot paths).
template <int FEAT, int DIR, int FF = 0, bool PAIRS = false, int NP = 1>
// (the cable robot's instantiation is held to 256 registers: with its channels as pairs a chunk holds two paths per SIMD, and the
//  forward kernel's 279 registers would leave the second one waiting)
__global__ void __launch_bounds__(S1_BLOCK, ((FEAT == 2 && FF == 0 && PAIRS) || NP == 2) ? 2 : 1) k_sweep1(SweepArgs a)
{
   static_assert(NP == 1 || NP == 2, "paths per wavefront");
   constexpr int NW = (S1_BLOCK / 64) * NP;            // paths (LDS windows) per workgroup
   constexpr int WM = (NP == 1) ? S1_WM : S1_WM / 2;   // points per reverse-curve window
   constexpr int WRB = S1_WR_BYTES / NP;               // bytes per coefficient-row window
   constexpr int WK = (NP == 1) ? S1_WK : S1_WK / 2;   // knots per spline window (compact splines)
   __shared__ double lim[6][8];
   // Sliding windows in LDS, one per wavefront (= per path): the (value, second derivative) pairs of S1_WK consecutive knots
   // and, for the forward sweep, S1_WM consecutive points of the reverse curve.  Both cursors move monotonically (up to small
   // back-steps), so a window is refilled once per ~S1_WK knots by one coalesced copy -- one memory round trip -- and a
   // segment change reads LDS instead of waiting ~1500 cycles for a dependent HBM access (39 % of the cycles of the lone
   // wavefront were s_waitcnt without the windows: profiles/r02_e_*).
   __shared__ double2 winKAll[(FEAT < 0) ? NW : 1][(FEAT < 0) ? WK * BATOTP_MAX_JOINTS : 1];
   __shared__ double2 winMAll[(DIR == 1) ? NW : 1][(DIR == 1) ? WM : 1];
   static_assert(!PAIRS || FEAT >= 0, "FEAT -1 reads its joint pairs through winK");
   // rows through an LDS window: always when the batch keeps pairs only; for rows in HBM where it measured faster (the cable robot's
   // sweeps 4-6 %, not the 7-DOF arm's, whose cursor needs a new row every other step)
   constexpr bool ROWWIN = FEAT >= 0 && S1_WR_BYTES > 0 && (PAIRS || (FEAT == 2 && FF == 0));
   __shared__ double2 winRAll[ROWWIN ? NW : 1][ROWWIN ? WRB / 16 : 1];
   stage_limits(a.dP, lim);
   const int lane = threadIdx.x & 63;
   const int hl = lane & 31;                                    // lane inside the path's half
   const int widx = (threadIdx.x >> 6) * NP + (NP == 2 ? (lane >> 5) : 0);   // this path's LDS windows
   const int pslot = blockIdx.x * NW + widx;
   if (pslot >= a.B || (NP == 1 && lane >= 32)) return;
   const int p = a.order ? a.order[pslot] : pslot;   // ragged batches: longest paths first (SweepArgs::order)
   const int j = lane & 7, cslot = hl >> 3;
   const bool writer = (hl == 0);
   const PathInfo pi = a.pinfo[p];
   const int n = (int)pi.n;
   const int64_t cap = a.cap;
   const int nJ = a.P.nJ, nIn = a.P.Cin, C = a.P.C;
   const bool jv = j < nJ;
   const int jr = jv ? j : 0; // row slot this lane reads (lanes beyond the last joint read joint 0 and discard it)
   const bool accOn = (a.P.flags & BATOTP_F_JNT_ACC_ON) != 0;

   const double sres = pi.sres_c, vfact = pi.vfact, afact = pi.afact;
   const double thrV = a.P.jnt_thresh * vfact, thrA = a.P.jnt_thresh * afact;
   const double absh = pi.integ_res; // BA::_integRes of this path
   const double h = DIR * absh;
   const double sEnd = sres * (double)(n - 1);
   const double sdotCap = sEnd / absh;                 // ba.cpp:1216
   const double sddotMax = 2 * sEnd / (absh * absh);   // ba.cpp:1257
   const double vmaxj = lim[0][j], amaxj = lim[1][j];
   const double tmaxj = lim[2][j], tminj = lim[3][j];
   const int64_t maxIntegSteps = (int64_t)floor(a.P.max_integ_time / absh) + 1;

   const double2 *__restrict__ km = (FEAT < 0) ? reinterpret_cast<const double2 *>(a.km) + pi.koff * nIn : nullptr;
   const double *__restrict__ coef = (FEAT < 0) ? nullptr : a.coef + pi.koff * C * 4;
   // FEAT >= 0 on a batch that keeps ALL channels as pairs (a.km set, C channels per knot): rows exist in the LDS window only
   const double2 *__restrict__ kmAll = PAIRS ? reinterpret_cast<const double2 *>(a.km) + pi.koff * C : nullptr;
   double2 *out = (DIR == 1 ? a.fwd : a.rev) + (int64_t)p * cap; // forward, curves in place: the same buffer as mvc (no __restrict__)
   batotp_path_result *__restrict__ r = a.res + p;

   // reverse curve the forward sweep follows
   const double *mvc = nullptr;
   int nMvc = 0;
   if (DIR == 1)
   {
      const int64_t nRev = r->n_rev;
      if (nRev < 2)
      {
         if (writer) { r->n_fwd = 0; r->steps_fwd = 0; r->t_total = 0; r->status_fwd = r->status_rev | BATOTP_ST_CAPACITY; r->n_bisect_fail_fwd = 0; }
         return;
      }
      mvc = reinterpret_cast<const double *>(a.rev + (int64_t)p * cap + (cap - nRev));
      nMvc = (int)nRev;
   }

   // ---- cursor state -------------------------------------------------------------------------------------------------
   int segC = (DIR == 1) ? 0 : n - 2;
   double sSeg = sres * (double)segC, sNext = sres * (double)(segC + 1); // sites of the cursor's segment (ba.cpp:800-806)
   double tauC = (DIR == 1) ? 0.0 : 1.0;
   int rowSeg = -1;
   double A3 = 0, B2 = 0, A6 = 0, c1 = 0; // 3*c3, 2*c2, 6*c3, c1 of this lane's joint on segment rowSeg
   double thD = 0, thD2 = 0;              // theta', theta'' of this lane's joint at the last evaluated position
   // FEAT >= 1: the three Cartesian channels on segment rowSeg (3*c3, 2*c2, 6*c3, c1 each; the same in every lane) and the
   // quadratic's coefficients at the last evaluated position (BA::evalCartQuadCoeffs, ba.cpp:1423-1439)
   const bool cartVelOn = FEAT >= 1 && (a.P.flags & BATOTP_F_CART_VEL_ON) != 0;
   const bool cartAccOn = FEAT >= 1 && (a.P.flags & BATOTP_F_CART_ACC_ON) != 0;
   const bool cartAny = cartVelOn || cartAccOn;
   double cA3[(FEAT >= 1) ? 3 : 1], cB2[(FEAT >= 1) ? 3 : 1], cA6[(FEAT >= 1) ? 3 : 1], cC1[(FEAT >= 1) ? 3 : 1];
#pragma unroll
   for (int q = 0; q < ((FEAT >= 1) ? 3 : 1); ++q) { cA3[q] = 0; cB2[q] = 0; cA6[q] = 0; cC1[q] = 0; }
   double cq0 = 0, cq1 = 0, cq2 = 0;
   const double quadA = a.P.quad_thresh * afact, quadA2 = a.P.quad_thresh * a.P.quad_thresh * afact * afact;
   const double cartAccMaxSQ = a.P.cart_acc_max * a.P.cart_acc_max, cartVelMax = a.P.cart_vel_max;
   Coef4 dynK[(FEAT == 2) ? 4 : 1];       // FEAT == 2: coefficient rows of a1..a4 of this lane's dynamics row on segment rowSeg
   double a1pt = 0, a2pt = 0, a3pt = 0, a4pt = 0;
   int segMVC = (DIR == 1) ? 0 : n - 2, mvcSeg = -1;
   double tauMVC = (DIR == 1) ? 0.0 : 1.0, mS0 = 0, mS1 = 0, mD0 = 0, mD1 = 0;
   double sdotMin = 0, sdotCur = 0, sddotH = 0, sddotL = 0;
   unsigned status = 0;
   int nfail = 0;
#ifdef BK_PROFILE_SECTIONS
   // diagnostic build: cycles in velocity limit / spline evaluation / first check / bisection passes, stages that bisect, passes
   unsigned long long cyA = 0, cyB = 0, cyC = 0, cyD = 0, nBis = 0, nPass = 0, nStage = 0, cyP1 = 0, cyP2 = 0, cyP3 = 0, nAcc = 0;
#endif

   double2 *winK = winKAll[(FEAT < 0) ? widx : 0];
   double2 *winM = winMAll[(DIR == 1) ? widx : 0];
   int wK0 = 0, wKn = 0; // knots [wK0, wK0 + wKn) are in winK, layout [knot][8 joint slots]
   int wM0 = 0, wMn = 0; // curve points [wM0, wM0 + wMn) are in winM
   double2 *winR = winRAll[ROWWIN ? widx : 0];
   constexpr bool rowWin = ROWWIN;
   int wR0 = 0, wRn = 0; // coefficient rows [wR0, wR0 + wRn) are in winR, C x 4 doubles each

   // make knots seg and seg + 1 available in winK: a coalesced copy of the window that extends from seg in the direction of
   // travel (two knots of slack behind it).  Fixed stride of 8 slots per knot: Cartesian channels the batch may carry are not
   // copied; slots beyond the last joint hold a copy of joint 0.
   auto needK = [&](int seg) __attribute__((always_inline)) {
      if (S1_UNI(seg >= wK0 && seg + 1 < wK0 + wKn)) return;
      int w = (DIR == 1) ? seg - 2 : seg + 4 - WK;
      const int wmax = n - WK;
      w = w > wmax ? wmax : w;
      w = w < 0 ? 0 : w;
      const int cntK = (n - w) < WK ? (n - w) : WK;
      const int groups = (cntK + 3) >> 2;  // 4 knots (32 lanes) per group
      const int kLane = hl >> 3;           // knot offset of this lane inside a group
      const double2 *__restrict__ src = km + jr;
      // Eight loads in flight per lane and round trip, then eight unconditional LDS writes (slots past the last knot of
      // the window receive a copy of the path's last knot and are never read).  Written with named temporaries and
      // unconditional stores: with a temporary array the compiler kept it in scratch memory, and with stores under a
      // condition it sank every load into its store's block -- load / s_waitcnt vmcnt(0) / ds_write per group, one
      // memory round trip per 4 knots.
#define S1_LD(T, I) { int ki = w + 4 * (g0 + I) + kLane; ki = ki < n ? ki : n - 1; T = src[(unsigned)(ki * nIn)]; }
#define S1_ST(T, I) winK[hl + 32 * (g0 + I)] = T;
#pragma unroll 1
      for (int g0 = 0; g0 < groups; g0 += 8)
      {
         double2 t0, t1, t2, t3, t4, t5, t6, t7;
         S1_LD(t0, 0) S1_LD(t1, 1) S1_LD(t2, 2) S1_LD(t3, 3) S1_LD(t4, 4) S1_LD(t5, 5) S1_LD(t6, 6) S1_LD(t7, 7)
         __builtin_amdgcn_sched_barrier(0);
         S1_ST(t0, 0) S1_ST(t1, 1) S1_ST(t2, 2) S1_ST(t3, 3) S1_ST(t4, 4) S1_ST(t5, 5) S1_ST(t6, 6) S1_ST(t7, 7)
      }
#undef S1_LD
#undef S1_ST
      wK0 = w; wKn = cntK;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   };
   // The same for coefficient rows (FEAT >= 0): a segment change then reads LDS instead of waiting for a dependent HBM access (the
   // cable robot changes segment at nearly every stage -- 0.4 steps per knot -- and every change cost the lone wavefront a memory
   // round trip).  Rows are knot-major and contiguous: the window is ONE contiguous block of HBM.
   auto needR = [&](int seg) __attribute__((always_inline)) {
      if (S1_UNI(seg >= wR0 && seg < wR0 + wRn)) return;
      const int rowD2 = C * 2;                        // 16-byte units per row
      const int WR = (WRB / 16) / rowD2;              // rows per window (C <= 8 + 3 + 32 channels: at least 11; two paths per wavefront: 5)
      const int nRows = n - 1;                        // rows 0 .. n - 2, one per segment
      int w = (DIR == 1) ? seg - 1 : seg + 2 - WR;    // a row of slack behind the direction of travel
      const int wmax = nRows - WR;
      w = w > wmax ? wmax : w;
      w = w < 0 ? 0 : w;
      const int cntR = (nRows - w) < WR ? (nRows - w) : WR;
      const int total = cntR * rowD2;
      if (PAIRS)
      {
         // all channels as (value, second derivative) pairs: the rows of the window are FORMED here, one (row, channel) per lane
         // and round -- emit_segment's formulas (spline.cpp:203-209) with x / 6 through div6 --, and nothing but the pairs
         // (half the bytes of the rows) is resident in HBM
         const int cells = cntR * C;
         const double2 *__restrict__ kp = kmAll + (int64_t)w * C;
#pragma unroll 1
         for (int e0 = 0; e0 < cells; e0 += 128)
         {
            double2 la[4], lb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
               int e = e0 + hl + 32 * u;
               e = e < cells ? e : cells - 1;
               la[u] = kp[e];
               lb[u] = kp[e + C];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
               const int e = e0 + hl + 32 * u;
               const double solL = la[u].y, solR = lb[u].y, yL = la[u].x, yR = lb[u].x;
               const double k3 = div6(solR - solL);
               const double k2 = solL / 2.0;
               const double k1 = yR - yL - div6(solR + 2 * solL);
               if (e < cells) { winR[2 * e] = make_double2(yL, k1); winR[2 * e + 1] = make_double2(k2, k3); }
            }
         }
         wR0 = w; wRn = cntR;
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
         return;
      }
      const double2 *__restrict__ src = reinterpret_cast<const double2 *>(coef + (int64_t)w * C * 4);
#define S1_LD(T, I) { const int e = e0 + hl + 32 * I; T = src[e < total ? e : total - 1]; }
#define S1_ST(T, I) { const int e = e0 + hl + 32 * I; if (e < total) winR[e] = T; }
#pragma unroll 1
      for (int e0 = 0; e0 < total; e0 += 256)
      {
         double2 t0, t1, t2, t3, t4, t5, t6, t7;
         S1_LD(t0, 0) S1_LD(t1, 1) S1_LD(t2, 2) S1_LD(t3, 3) S1_LD(t4, 4) S1_LD(t5, 5) S1_LD(t6, 6) S1_LD(t7, 7)
         __builtin_amdgcn_sched_barrier(0);
         S1_ST(t0, 0) S1_ST(t1, 1) S1_ST(t2, 2) S1_ST(t3, 3) S1_ST(t4, 4) S1_ST(t5, 5) S1_ST(t6, 6) S1_ST(t7, 7)
      }
#undef S1_LD
#undef S1_ST
      wR0 = w; wRn = cntR;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   };
   // the same for points k and k + 1 of the reverse curve (a few points behind k stay in the window for the back-steps)
   auto needM = [&](int k) __attribute__((always_inline)) {
      if (S1_UNI(k >= wM0 && k + 1 < wM0 + wMn)) return;
      int w = k - 16;
      const int wmax = nMvc - WM;
      w = w > wmax ? wmax : w;
      w = w < 0 ? 0 : w;
      const int cnt = (nMvc - w) < WM ? (nMvc - w) : WM;
      const double2 *src = reinterpret_cast<const double2 *>(mvc) + w;
      {
         // the whole window in one round trip: 8 loads in flight per lane (indices clamped to the last point), 8 LDS writes
         static_assert(S1_WM == 256, "needM moves 8 (one path per wavefront) or 4 (two) groups of 32 points");
#define S1_LD(T, I) { const int e = hl + 32 * I; T = src[e < cnt ? e : cnt - 1]; }
#define S1_ST(T, I) winM[hl + 32 * I] = T;
         double2 t0, t1, t2, t3, t4, t5, t6, t7;
         S1_LD(t0, 0) S1_LD(t1, 1) S1_LD(t2, 2) S1_LD(t3, 3)
         if (NP == 1) { S1_LD(t4, 4) S1_LD(t5, 5) S1_LD(t6, 6) S1_LD(t7, 7) }
         __builtin_amdgcn_sched_barrier(0);
         S1_ST(t0, 0) S1_ST(t1, 1) S1_ST(t2, 2) S1_ST(t3, 3)
         if (NP == 1) { S1_ST(t4, 4) S1_ST(t5, 5) S1_ST(t6, 6) S1_ST(t7, 7) }
#undef S1_LD
#undef S1_ST
      }
      wM0 = w; wMn = cnt;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   };

   // BA::updateCurSeg on the knot sites (ba.cpp:1617-1652): the literal walk, sites sres*k recomputed only when the cursor moves
   auto walkC = [&](double sCur) __attribute__((always_inline)) {
      const int lastSeg = n - 2;
      for (;;)
      {
         if (sCur >= sSeg && sCur <= sNext) break;
         bool moved = false;
         if (sCur > sSeg)
         {
            if (segC >= lastSeg) { segC = lastSeg; break; }
            ++segC; moved = true;
         }
         if (sCur < sSeg)
         {
            if (segC <= 0) { segC = 0; break; }
            --segC; moved = true;
         }
         if (!moved) { status |= BATOTP_ST_NONFINITE; break; }
         sSeg = sres * (double)segC;
         sNext = sres * (double)(segC + 1);
      }
      tauC = (sCur - sSeg) / (sNext - sSeg);
   };

   // BA::evalSplinePartials for joint velocity / acceleration limits only (ba.cpp:1341-1366)
   auto evalPartials = [&](double sCur) __attribute__((always_inline)) {
      walkC(sCur);
      if (S1_UNI(segC != rowSeg))
      {
         double k3, k2, k1;
         if (FEAT < 0)
         {
            needK(segC);
            const int at = (segC - wK0) * BATOTP_MAX_JOINTS + j;
            const double2 kl = winK[at], kr = winK[at + BATOTP_MAX_JOINTS]; // knots segC and segC + 1 of this joint
            k3 = div6(kr.y - kl.y);                        // spline.cpp:203-209
            k2 = kl.y / 2.0;
            k1 = kr.x - kl.x - div6(kr.y + 2 * kl.y);
         }
         else
         {
            auto readRow = [&](auto row) __attribute__((always_inline)) {
               const Coef4 k = *reinterpret_cast<const Coef4 *>(row + jr * 4);
               k3 = k.c3; k2 = k.c2; k1 = k.c1;
               if (FEAT >= 1 && cartAny)
               {
#pragma unroll
                  for (int q = 0; q < ((FEAT >= 1) ? 3 : 1); ++q)
                  {
                     const Coef4 kc = *reinterpret_cast<const Coef4 *>(row + (nJ + q) * 4);
                     cA3[q] = 3 * kc.c3; cB2[q] = 2 * kc.c2; cA6[q] = 6 * kc.c3; cC1[q] = kc.c1;
                  }
               }
               if (FEAT == 2)
               {
                  // device channel order: theta[nJ], cart[nC], then per dynamics row r the four channels (a1_r, a2_r, a3_r, a4_r)
                  const Coef4 *kd = reinterpret_cast<const Coef4 *>(row + (nIn + jr * 4) * 4);
#pragma unroll
                  for (int q = 0; q < 4; ++q) dynK[(FEAT == 2) ? q : 0] = kd[q];
               }
            };
            // rows through the LDS window, or from HBM
            if (ROWWIN && rowWin)
            {
               needR(segC);
               readRow(reinterpret_cast<const double *>(winR) + (unsigned)((segC - wR0) * C * 4));
            }
            else readRow(coef + (unsigned)(segC * C * 4));
         }
         // lanes beyond the last joint carry joint 0's numbers; every use of thD / thD2 is behind `jv`
         A3 = 3 * k3;
         B2 = 2 * k2;
         A6 = 6 * k3;
         c1 = k1;
         rowSeg = segC;
      }
      const double tau = tauC, tau2 = tau * tau;
      thD = (A3 * tau2 + B2 * tau + c1) * vfact; // (3*c3*tau2 + 2*c2*tau + c1)*vFact, ba.cpp:1359
      thD2 = (A6 * tau + B2) * afact;            // (6*c3*tau + 2*c2)*aFact, ba.cpp:1360
      if (FEAT >= 1 && cartAny)
      {
         // Cartesian velocity / acceleration along s and the quadratic's coefficients, ba.cpp:1368-1385, 1423-1439
         double v[3];
#pragma unroll
         for (int q = 0; q < 3; ++q)
            v[q] = (cA3[(FEAT >= 1) ? q : 0] * tau2 + cB2[(FEAT >= 1) ? q : 0] * tau + cC1[(FEAT >= 1) ? q : 0]) * vfact;
         cq0 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
         if (cartAccOn)
         {
            // the acceleration along s and the other two coefficients are read by the Cartesian acceleration limit only
            double ac[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) ac[q] = (cA6[(FEAT >= 1) ? q : 0] * tau + cB2[(FEAT >= 1) ? q : 0]) * afact;
            cq1 = 2 * (v[0] * ac[0] + v[1] * ac[1] + v[2] * ac[2]);
            cq2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
         }
      }
      if (FEAT == 2)
      {
         // a1..a4 at the cursor, ba.cpp:1387-1405
         const double tau3 = tau2 * tau;
         const Coef4 q1 = dynK[0], q2 = dynK[(FEAT == 2) ? 1 : 0], q3 = dynK[(FEAT == 2) ? 2 : 0], q4 = dynK[(FEAT == 2) ? 3 : 0];
         a1pt = q1.c3 * tau3 + q1.c2 * tau2 + q1.c1 * tau + q1.c0;
         a2pt = q2.c3 * tau3 + q2.c2 * tau2 + q2.c1 * tau + q2.c0;
         a3pt = q3.c3 * tau3 + q3.c2 * tau2 + q3.c1 * tau + q3.c0;
         a4pt = q4.c3 * tau3 + q4.c2 * tau2 + q4.c1 * tau + q4.c0;
      }
   };

   // BA::updateCurSeg on the reverse curve (ba.cpp:1592, 1617-1652) with the segment's two points cached in registers:
   // the literal walk of update_cur_seg<2>
   auto mvcWalk = [&](double sCur) __attribute__((always_inline)) {
      if (S1_UNI(mvcSeg == segMVC && sCur >= mS0 && sCur <= mS1))
      {
         tauMVC = (sCur - mS0) / (mS1 - mS0);
         return;
      }
      const int lastSeg = nMvc - 2;
      double2 pa, pb;
      for (;;)
      {
         needM(segMVC);
         pa = winM[segMVC - wM0];
         pb = winM[segMVC + 1 - wM0];
         if (S1_UNI(sCur >= pa.x && sCur <= pb.x)) break;
         bool moved = false;
         if (S1_UNI(sCur > pa.x))
         {
            if (S1_UNI(segMVC >= lastSeg)) { segMVC = lastSeg; break; }
            ++segMVC; moved = true;
         }
         if (S1_UNI(sCur < pa.x))
         {
            if (S1_UNI(segMVC <= 0)) { segMVC = 0; break; }
            --segMVC; moved = true;
         }
         if (!moved) { status |= BATOTP_ST_NONFINITE; break; }
      }
      tauMVC = (sCur - pa.x) / (pb.x - pa.x);
      mS0 = pa.x; mD0 = pa.y; mS1 = pb.x; mD1 = pb.y;
      mvcSeg = segMVC;
   };

   // BA::sdotLim (ba.cpp:1204-1236); theta' is the one of the previous evalSplinePartials call, as in the reference
   auto sdotLim = [&](double sCur, double &sdot) __attribute__((always_inline)) {
      if (DIR == 1)
      {
         mvcWalk(sCur);
         const double sdotMVC = dmax(mD0 + tauMVC * (mD1 - mD0), sdotMin); // evalsdot, ba.cpp:1590-1607
         if (sdot > sdotMVC) sdot = sdotMVC;
      }
      sdot = dmin(sdot, sdotCap);
      sdot = dmax(sdot, sdotMin);
      double l = kInf;
      if (jv && fabs(thD) > thrV) l = dmin(l, fabs(vmaxj / thD));
      l = grp_min<8>(l);
      sdot = dmin(sdot, l);
      if (FEAT >= 1 && cartVelOn && S1_UNI(cq0 > quadA))
      {
         // ba.cpp:1225-1229.  The quotient only matters when it is below sdot: cartVelMax^2 > sdot^2 cq0 (1 + 1e-12) =>
         // the correctly rounded cartVelMax / sqrt(cq0) > sdot (three products, a root and a quotient are off by 5 eps together)
         // => min(sdot, .) = sdot: root and division are skipped.  Overflow or a NaN make the test false or leave it right.
         // (The same shortcut for the joint velocity limits -- a division per lane and a reduction -- measured slower: its
         // ballot and branch in every stage cost more than the skipped quotients save, cfg 4 reverse 451 -> 482 ms.)
         if (!S1_UNI(cartVelMax > 0.0 && cartVelMax * cartVelMax > ((sdot * sdot) * cq0) * (1.0 + 1e-12)))
            sdot = dmin(sdot, cartVelMax / sqrt(cq0));
      }
   };

   // BA::verifySecondOrderConstraints, joint acceleration family (ba.cpp:1514-1534); see verify_second_order for why the
   // reference's early exits are the reduced predicate
   auto verify = [&](double sdotTry) __attribute__((always_inline)) -> bool {
      const double sdotSQ = sdotTry * sdotTry;
      double H = sddotMax, L = -sddotMax;
      bool force = false;
      if (FEAT == 2 && jv)
      {
         // torque limits of a serial robot, ba.cpp:1495-1509
         const double tmp1 = a3pt * sdotTry + a4pt;
         if (!(fabs(a1pt) < thrV))
         {
            const double tmp2 = a2pt * sdotSQ + tmp1;
            const double s0 = (tmaxj - tmp2) / a1pt;
            const double s1 = (tminj - tmp2) / a1pt;
            H = dmin(H, dmax(s0, s1));
            L = dmax(L, dmin(s0, s1));
         }
      }
      if (accOn && jv)
      {
         const double vpt = thD;
         if (fabs(vpt) < thrV)
         {
            if (!(fabs(thD2) < thrA))
            {
               if (sdotSQ > amaxj / fabs(thD2)) force = true;
            }
         }
         else
         {
            const int svpt = sgn(vpt);
            const double vTerm = thD2 * sdotSQ;
            H = dmin(H, (svpt * amaxj - vTerm) / vpt);
            L = dmax(L, (-svpt * amaxj - vTerm) / vpt);
         }
      }
      double Hred = force ? -kInf : H;
      grp_min_max<8>(Hred, L);
      sddotH = Hred;
      sddotL = L;
      if (L > Hred) return true;
      if (FEAT >= 1 && cartAccOn)
      {
         // ba.cpp:1535-1579 (the values are the same in the 8 lanes of a candidate slot)
         if (cq0 > quadA)
         {
            const double Bq = cq1 * sdotSQ;
            const double Cq = cq2 * sdotSQ * sdotSQ - cartAccMaxSQ;
            double sol1 = 0, sol2 = 0;
            const int ef = solve_quadratic(cq0, Bq, Cq, sol1, sol2);
            if (ef == -1) return true;
            const double cmax = dmax(sol1, sol2), cmin = dmin(sol1, sol2);
            sddotH = dmin(sddotH, cmax);
            sddotL = dmax(sddotL, cmin);
            if (sddotL > sddotH) return true;
         }
         else
         {
            if (cq2 < quadA2) return false;
            if (sdotSQ * sdotSQ > cartAccMaxSQ / cq2) return true;
            return false;
         }
      }
      return false;
   };

   // BA::applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) with four candidates per pass: the reference's loop is replayed
   // literally; a speculated candidate is used only if it is bit-identical to the value the replay asks for (see
   // apply_accel_bisection_spec).  Returns 0, or -1 on the failure exits (sddot untouched).
   auto accelPt = [&](double sCur, double &sddot) __attribute__((always_inline)) {
      const double sdotErrThresh = .001;
      double lowFact = .01;
      double sdotGood = 0;
      double sdotL = 0;
      double sdotH = sdotCur;
      double sdotTry = sdotH;
      int nIter = 0;
      BK_TICK(tp0);
      evalPartials(sCur); // ba.cpp:1265
      BK_TICK(tp1);
      BK_ACC(cyB, tp0, tp1);

      // the common case first: the speed the velocity limits left is admissible (three quarters of the stages)
      const bool firstViol = verify(sdotTry);
      BK_TICK(tp2);
      BK_ACC(cyC, tp1, tp2);
#ifdef BK_PROFILE_SECTIONS
      ++nStage;
#endif
      if (!S1_UNI(firstViol)) // every slot has checked the same speed
      {
         sddot = (DIR == 1) ? sddotH : sddotL;
         return;
      }
#ifdef BK_PROFILE_SECTIONS
      ++nBis;
#endif
      // first check violated: replay of ba.cpp:1276-1321 from its first iteration, four candidates per pass.  The check of
      // the first candidate has just been done: it is folded into the first pass below (slot 0 re-evaluates it, same bits).
      // One replayed iteration is written as selects (the form of the flat loop in k_sweep: no divergent branches around the few
      // operations of the update); the loop conditions are wavefront-uniform and made scalar with readfirstlane.
      int lastSlot = 0;
      int nGood = 0; // feasible points seen (anyGoodIter of ba.cpp:1254 == nGood > 0)
      bool fin = false, failed = false;
      // one iteration of the loop of ba.cpp:1267-1321 given the outcome of the check of sdotTry; leaves the next value to check
      // in sdotTry; returns true when the loop has ended (fin or failed)
      auto iterate = [&](bool isViol) __attribute__((always_inline)) -> bool {
         const bool first = (nIter == 0);
         const bool good = !isViol && !first;      // a feasible point after at least one violated one
         const bool shrink = isViol && nGood == 0; // ba.cpp:1281-1285: no feasible point known yet
         const double lowFact2 = lowFact * 2.0;
         const double sdotLShrunk = dmax(.999 * 0.0, (1.0 - lowFact2) * sdotTry);
         // ba.cpp:1294-1303: two successive feasible points closer than 1e-3 (relative), or a negative one
         const bool conv = good && (S1_RATIO_LT(fabs(sdotTry - sdotGood), sdotTry, sdotErrThresh) || sdotTry < 0.0);
         fin = (!isViol && first) || conv;
         lowFact = shrink ? lowFact2 : lowFact;
         sdotH = isViol ? sdotTry : sdotH;
         sdotL = shrink ? sdotLShrunk : ((good && !conv) ? sdotTry : sdotL);
         sdotGood = good ? sdotTry : sdotGood;
         nGood += good ? 1 : 0;
         sdotCur = conv ? sdotTry : sdotCur;
         // ba.cpp:1305-1320
         const bool collapsed = (nGood == 0) && S1_RATIO_LT(sdotH - sdotL, sdotH, 1e-20);
         failed = !fin && (nIter + 1 > 100 || sdotTry < 0.0 || collapsed);
         nIter += fin ? 0 : 1;
         const bool stop = fin || failed;
         sdotTry = stop ? sdotTry : .5 * (sdotH + sdotL);
         return stop;
      };
      // the first check (violated) is the loop's first iteration; the passes below start with its successor
      bool over = S1_FIRST((int)iterate(true)) != 0;
#if S1_PREDICT
      // where it applies: the check consists of constraints that are LINES in x = sdot^2 -- joint acceleration limits, and
      // the torque limits of a mechanism whose a3 (the term in sdot) vanishes identically, i.e. the cable robot in serial form
      // (robot.cpp:487-517: a3 = 0 at every knot, so its spline is 0 and tmp1 of ba.cpp:1497 is a4 exactly); no Cartesian
      // acceleration limit (a quadratic in sddot, ba.cpp:1535-1579).  The torque lines take the joint lanes 4..7 of a slot:
      // at most 4 joints.
      // (not in the forward kernel of a pair batch: it is held to 256 registers -- two paths per SIMD --, the forward sweep bisects in
      //  0.6 % of its stages, and without the block 22 instead of 35 registers spill: cfg 5 forward 3194 -> 3021 ms)
#ifndef S1_FWD_PAIRS_FF
#define S1_FWD_PAIRS_FF 0
#endif
#ifndef S1_FAIL_FF
#define S1_FAIL_FF 1   // the certificate of a stage whose bisection cannot succeed (0: A/B, the kernels of round 5)
#endif
      constexpr bool FF0 = FF == 0 && (S1_FWD_PAIRS_FF || !(PAIRS && DIR == 1));
      // (round 6) the certificate of a stage that CANNOT succeed -- see below -- is a small part of the block and is compiled into every
      // instantiation of this form, the forward kernel of a pair batch included
      constexpr bool FFAIL0 = FF == 0 && (S1_FAIL_FF != 0);
      bool ffApplies = (FF0 || FFAIL0) && a.ff && !over && !cartAccOn && (FEAT == 2 || accOn);
      if (FEAT == 2) ffApplies = ffApplies && nJ <= 4 && !S1_BALLOT(jv && !(a3pt == 0.0));
      if ((FF0 || FFAIL0) && ffApplies)
      {
         // CERTIFIED FAST-FORWARD.  In x = sdot^2 every constraint of the check is an interval [l_q(x), u_q(x)] for sddot with
         //      u_q = au_q - m_q x,   l_q = al_q - m_q x:
         //   joint acceleration, moving joint (ba.cpp:1526-1531):  au = -al = amax_q / |theta'_q|,  m = theta''_q / theta'_q;
         //   torque, |a1_q| >= thresh (ba.cpp:1495-1509, a3 = 0):  au, al = max, min of (tmax_q - a4_q) / a1_q, (tmin_q - a4_q) / a1_q,
         //                                                         m = a2_q / a1_q;
         //   [-sddotMax, sddotMax] (ba.cpp:1257):                  au = -al = sddotMax, m = 0;
         // and a joint that stands still allows x <= amax_q / |theta''_q| (ba.cpp:1519-1524).
         // g(x) = min u - max l = min over pairs (i, j) of (au_j - al_i) - (m_j - m_i) x is concave and vanishes at
         //      x* = min over pairs with m_j > m_i of (au_j - al_i) / (m_j - m_i),
         // so a check is a comparison of x with x* -- except within rounding of x*:
         //  * the check's bounds differ from u_q, l_q by at most 4 eps e_q(x), e_q = amax_q / |theta'_q| + |m_q| x resp.
         //    (|tmax_q| + |tmin_q| + 2 |a4_q| + |a2_q| x) / |a1_q| (two or three correctly rounded operations and a quotient,
         //    eps = 2^-53): its decision is the exact one when |g(x)| > 8 eps E, E = max_q e_q(x);
         //  * g concave with g(0) = Smin = min au - max al > 0 (demanded) and g(x*) = 0:  |g(x)| >= Smin |x - x*| / x* on both
         //    sides of x*;
         //  * x* is computed with approximate reciprocals (2 eps each): a pair whose bound is below 2 x* has m_j - m_i >= Smin / (2 x*),
         //    so its computed bound is off by at most 32 eps E / Smin relatively, and no other pair can come out below x* as long
         //    as 32 eps E / Smin < 1/2.
         // With R = 2 E(first candidate) / Smin the outcome of a check is certain when |x - x*| > 20 eps R x*; demanded here:
         // |x - x*| > 2^-40 R x* (400 times that) and R < 2^30.  The standing joints' thresholds are the check's own quotients and
         // comparisons.  The loop of ba.cpp:1267-1321 is then advanced, with its own update statements, through every iteration
         // whose outcome is certain and that neither ends it nor can take a failure exit: speeds in the normal range are
         // positive, the bracket of the search phase is [(1 - lowFact) c, c] with lowFact >= 0.02 (never collapsed), and the
         // iteration count is kept below 90.  It stops in front of the first candidate that is within the band (the generic
         // passes below go on from this state with real checks) or that would end the loop -- a feasible speed within 1e-3 of
         // the previous one: it gets the real check, whose sddot bounds are the result.
         // reciprocal to ~2 eps whatever the accuracy of v_rcp_f64's seed beyond 14 bits: two Newton steps
         auto fastRcp = [](double d) {
            double r = __builtin_amdgcn_rcp(d);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
         };
         const double xTop = sdotH * sdotH;                        // the first candidate: no later one is larger
         // this lane's line: the acceleration line of its joint ...
         const bool use = accOn && jv && !(fabs(thD) < thrV);
         bool valid = use;
         double au, al, mj, ej;
         {
            const double rv = fastRcp(use ? thD : 1.0);
            au = amaxj * fabs(rv);
            al = -au;
            mj = thD2 * rv;
            ej = au + fabs(mj) * xTop;
         }
         if (FEAT == 2)
         {
            // ... or, in the joint lanes 4..7 of the slot, the torque line of joint j - 4 (computed in that joint's lane)
            const bool useT = jv && !(fabs(a1pt) < thrV);
            const double r1 = fastRcp(useT ? a1pt : 1.0);
            const double q0 = (tmaxj - a4pt) * r1, q1 = (tminj - a4pt) * r1;
            const double tu = dmax(q0, q1), tl = dmin(q0, q1), tm = a2pt * r1;
            const double te = (fabs(tmaxj) + fabs(tminj) + 2.0 * fabs(a4pt) + fabs(a2pt) * xTop) * fabs(r1);
            const int src = (lane & 56) | (j & 3);
            const double su = __shfl(tu, src), sl = __shfl(tl, src), sm = __shfl(tm, src), se = __shfl(te, src);
            const int sv = __shfl((int)useT, src);
            if (j >= 4) { au = su; al = sl; mj = sm; ej = se; valid = sv != 0; }
         }
         // a line with a non-finite coefficient (theta' = 0 with a zero threshold gives rcp(0) here) must stop the fast-forward:
         // the min / max reductions below would silently drop its NaN
         const bool lineFinite = !valid || ((au == au) & (mj == mj) & (ej == ej) & (fabs(au) < kInf) & (fabs(mj) < kInf) & (ej < kInf));
         const bool allFinite = S1_BALLOT(!lineFinite) == 0;
         au = valid ? au : kInf;
         al = valid ? al : -kInf;
         mj = valid ? mj : 0.0;
         ej = valid ? ej : 0.0;
         double uMin = au, lMax = al;
         grp_min_max<8>(uMin, lMax);
         const double eMax = grp_max<8>(ej);
         const double sMin2 = .5 * (dmin(uMin, sddotMax) - dmax(lMax, -sddotMax));
         if (FFAIL0 && S1_UNI(sMin2 < 0.0))
         {
            // CERTAIN FAILURE (round 6).  The sddot interval is empty already at x = 0: the path asks for something the limits do not
            // admit at any speed (random cable-robot paths do: tensions outside [tmin, tmax] at rest).  The reference's
            // loop (ba.cpp:1267-1321) then halves the speed a hundred times, every check violated, and returns -1 without touching
            // sddot (:1307-1319; the caller ignores the code, :1091).  Whatever exit ends it -- the iteration count, a collapsed
            // bracket, a negative speed -- what it leaves is the same: the status bit, the counter, sddot and sdotCur as they were.  So
            // the hundred checks can be skipped when NO speed in [0, first candidate] can pass one: g(x) = min u - max l is at most the
            // gap of any ONE pair of lines (i above, j below), (au_i - al_j) - (m_i - m_j) x, which is affine in x -- negative on the
            // whole interval when it is negative at both ends.  The pair: the lines that bind at x = 0 (ties: the slopes that open the
            // gap fastest; any tie is a valid witness).  The check's computed bounds are off by at most 4 eps e_q(x), e_q(x) <= e_q of
            // the first candidate, these line coefficients by a few eps: demanded, at both ends, gap < -2^-40 E (the margin of the
            // certificate below).  Every candidate of the loop lies in [0, first candidate]: each is violated for certain.
            const double uB = dmin(uMin, sddotMax), lB = dmax(lMax, -sddotMax);
            const double mU = (uMin < sddotMax) ? grp_max<8>((valid && au == uMin) ? mj : -kInf) : 0.0;   // (the clamp is a line of slope 0)
            const double mL = (lMax > -sddotMax) ? grp_min<8>((valid && al == lMax) ? mj : kInf) : 0.0;
            const double gap0 = uB - lB, gapTop = gap0 - (mU - mL) * xTop;
            const double tolF = eMax * 0x1p-40;
            const bool certain = allFinite & (eMax == eMax) & (eMax < 1e100) & (xTop < 1e100) & (sddotMax == sddotMax) & (fabs(mU) < 1e100) & (fabs(mL) < 1e100) &
                                 (gap0 < -tolF) & (gapTop < -tolF);
            if (S1_UNI(certain))
            {
               status |= BATOTP_ST_BISECT_FAIL;
               nfail++;
               BK_TICK(tpf);
               BK_ACC(cyD, tp2, tpf);
               return;
            }
         }
         if (FF0)
         {
         const bool standing = accOn && jv && !use && !(fabs(thD2) < thrA);
         double xForce = kInf;
         if (S1_BALLOT(standing)) xForce = grp_min<8>(standing ? amaxj / fabs(thD2) : kInf);
         // the pairs of this lane's line with (u_0, l_0): as the upper line when m > 0, as the lower line when m < 0
         double xs = kInf;
         if (valid && mj > 0.0) xs = (au + sddotMax) * fastRcp(mj);
         if (valid && mj < 0.0) xs = (sddotMax - al) * fastRcp(-mj);
#pragma unroll
         for (int rr = 0; rr < 2; ++rr)
         {
            // this lane's line as the upper one, line cslot + 4 rr of its own slot as the lower one
            const int srcLane = (lane & 56) | (cslot + 4 * rr);
            const double ali = __shfl(al, srcLane), mi = __shfl(mj, srcLane);
            const double dm = mj - mi;
            const double bnd = (au - ali) * fastRcp(dm > 0.0 ? dm : 1.0);
            xs = dmin(xs, dm > 0.0 ? bnd : kInf);
         }
         xs = grp_min<8>(xs);
         xs = vmin_f64(xs, dpp_mov<DPP_ROW_ROR8>(xs));
         xs = vmin_f64(xs, __shfl_xor(xs, 16));
         const double xstar = dmin(xs, 4.0 * xTop);               // beyond 4 xTop: "never violated by the lines" just as well
         const double R = eMax * fastRcp(sMin2 > 0.0 ? sMin2 : 1.0);
         const double band = (R * 0x1p-40) * xstar;
         // One threshold for the loops below.  A standing joint's threshold below the band around x* decides alone, and exactly
         // (xh > xForce <=> xh - xForce > 0; the candidates it lets pass are more than the band below x*): threshold xForce, no
         // band.  One above the band never matters (what exceeds it is violated by the lines for certain): threshold x*.
         // One inside the band: no fast-forward.
         const bool forceFirst = xForce < xstar - band;
         const double xThr = forceFirst ? xForce : xstar;
         const double bandThr = forceFirst ? -1.0 : band;
         // magnitudes far inside the normal range (no overflow, no gradual underflow in the check or here); NaNs fail every test
         // (every quantity of the certificate must be a finite number: a NaN operand can be dropped by the min / max reductions that
         //  formed eMax / xstar, so the comparisons below are made on values that are tested for finiteness explicitly)
         const bool sane = allFinite & (eMax == eMax) & (xstar == xstar) & (R == R) & (R < 0x1p30) & (sMin2 > 1e-100) & (eMax < 1e100) & (xTop > 1e-100) & (xTop < 1e100) & (xstar > 1e-100) &
                           (forceFirst | (xForce > xstar + band));
         BK_TICK(tq1);
         BK_ACC(cyP1, tp2, tq1);
         bool expectEnd = false;
         if (S1_UNI(sane))
         {
            // every value below is the same in all lanes: the loop conditions are scalar branches on ballots
            // (bitwise operators on purpose: '||' and '&&' become exec-mask branches around single compares)
            int it = S1_FIRST(nIter);
            bool inBand = false;
            // the search for a first feasible speed (ba.cpp:1281-1285): the bracket shrinks below every violated candidate
#pragma unroll 1
            for (; it < 90; ++it)
            {
               const double c = sdotTry, d = c * c - xThr;           // c * c: sdotSQ of the check
               inBand = !((fabs(d) > bandThr) & (c > 1e-100));
               if (S1_UNI(inBand | !(d > 0.0))) break;
               lowFact *= 2.0;
               sdotH = c;
               sdotL = dmax(.999 * 0.0, (1.0 - lowFact) * c);
               sdotTry = .5 * (sdotH + sdotL);
            }
            if (!S1_UNI(inBand) && it < 90)
            {
               // sdotTry is feasible for certain and the first such speed: ba.cpp:1294 compares it with sdotGood = 0 and goes on
               // (|c - 0| > 1e-3 c).  From here the plain bisection (ba.cpp:1286-1303): sdotGood == sdotL throughout, every
               // candidate lies between this speed (> 1e-100) and the first candidate (< 1e50).
               sdotGood = sdotTry; nGood = 1; sdotL = sdotTry;
               ++it;
               sdotTry = .5 * (sdotH + sdotL);
               bool goesOn = true;
               // ba.cpp:1294 is false for certain when |c - sdotGood| > 1e-3 c (1 + 3e-14): ratio_lt(., c, 1e-3) then answers "no"
               // without dividing (its own margin is 1e-14 on the twice-rounded product) -- otherwise this candidate is, or may
               // be, the last one and gets the real check and the real test
               const double convThr = 1e-3 * (1.0 + 3e-14);
#pragma unroll 1
               for (; it < 90; ++it)
               {
                  const double c = sdotTry, d = c * c - xThr;
                  const bool viol = d > 0.0;
                  inBand = !(fabs(d) > bandThr);
                  goesOn = viol | (fabs(c - sdotL) > convThr * c);
                  if (S1_UNI(inBand | !goesOn)) break;
                  sdotH = viol ? c : sdotH;
                  sdotL = viol ? sdotL : c;
                  sdotTry = .5 * (sdotH + sdotL);
               }
               sdotGood = sdotL;
               expectEnd = !S1_UNI(inBand | goesOn);
            }
            nIter = it;
         }
         BK_TICK(tq2);
         BK_ACC(cyP2, tq1, tq2);
         if (expectEnd)
         {
            // the candidate that should end the loop: the real check and the real test (every slot evaluates the same speed)
            const double c = sdotTry;
            const bool violC = verify(c);
            const bool convC = ratio_lt(fabs(c - sdotGood), c, sdotErrThresh);
            BK_TICK(tq3);
            BK_ACC(cyP3, tq2, tq3);
            if (S1_UNI(!violC && convC))
            {
#ifdef BK_PROFILE_SECTIONS
               ++nAcc;
#endif
               sdotCur = c;                                        // ba.cpp:1296-1302
               sddot = (DIR == 1) ? sddotH : sddotL;
               BK_TICK(tpp);
               BK_ACC(cyD, tp2, tpp);
               return;
            }
         }
               }
}
      if (FEAT == 2 && FF == 1 && a.ff && !over && !cartAccOn)
      {
         // CERTIFIED FAST-FORWARD, general form for serial torque limits (a3 != 0: the KUKA chain's friction, the two-link arm; more
         // than 4 joints).  The bounds of the check are then quadratics in sdot,
         //      torque, |a1_q| >= thresh (ba.cpp:1495-1509):  (t_q - a4_q - a3_q c - a2_q c^2) / a1_q  for t_q = tmax_q, tmin_q,
         //      joint acceleration (ba.cpp:1526-1531):        +-amax_q / |theta'_q| - (theta''_q / theta'_q) c^2,
         // and g(c) = min u - max l has no closed-form first zero worth its price (15 x 15 pairs, a root each) and is not concave.
         // Instead every candidate gets an APPROXIMATE check: the bounds from coefficients divided once by a1_q resp. theta'_q
         // (reciprocals to 2 eps), six rounded operations per bound and no division, then the same min / max reduction as the
         // real check.  The real check's bounds differ from the exact ones by at most 6 eps e_q (ba.cpp's three or four rounded
         // operations and a quotient), e_q = (|tmax_q| + |tmin_q| + 2 |a4_q| + |a3_q| c + |a2_q| c^2) / |a1_q| resp.
         // amax_q / |theta'_q| + |theta''_q / theta'_q| c^2, the approximate ones by at most 8 eps e_q: where the approximate
         // g(c) is farther from 0 than 28 eps E, E = max_q e_q at the first candidate, the real check decides the same way.
         // Demanded: |g(c)| > 2^-44 E (18 times that).  Standing joints: the check's own quotient, exactly.  The loop of
         // ba.cpp:1267-1321 is advanced through the certain iterations as above; the first uncertain candidate goes to the generic
         // passes below, the candidate that would end the loop gets the real check (its sddot bounds are the result).
         auto fastRcp = [](double d) {
            double r = __builtin_amdgcn_rcp(d);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
         };
         const double cTop = sdotH, xTop = cTop * cTop;           // the first candidate: no later one is larger
         // acceleration line of this lane's joint: u = aa - ma x, l = -aa - ma x
         const bool useA = accOn && jv && !(fabs(thD) < thrV);
         const double ra = fastRcp(useA ? thD : 1.0);
         const double aa = useA ? amaxj * fabs(ra) : kInf;
         const double ma = useA ? thD2 * ra : 0.0;
         // torque bounds of this lane's joint: u = tu - tb c - tm x, l = tl - tb c - tm x
         const bool useT = jv && !(fabs(a1pt) < thrV);
         const double r1 = fastRcp(useT ? a1pt : 1.0);
         const double q0 = (tmaxj - a4pt) * r1, q1 = (tminj - a4pt) * r1;
         const double tu = useT ? dmax(q0, q1) : kInf, tl = useT ? dmin(q0, q1) : -kInf;
         const double tb = useT ? a3pt * r1 : 0.0, tm = useT ? a2pt * r1 : 0.0;
         double eq = useA ? aa + fabs(ma) * xTop : 0.0;
         const double eT = (fabs(tmaxj) + fabs(tminj) + 2.0 * fabs(a4pt) + fabs(a3pt) * cTop + fabs(a2pt) * xTop) * fabs(r1);
         eq = useT ? dmax(eq, eT) : eq;
         // a bound with a non-finite coefficient (a divisor of exactly 0 under a zero threshold) must stop the fast-forward: the
         // max reduction below would drop its NaN
         const bool finA = !useA || ((aa == aa) & (ma == ma) & (aa < kInf) & (fabs(ma) < kInf));
         const bool finT = !useT || ((tu == tu) & (tl == tl) & (tb == tb) & (tm == tm) & (fabs(tu) < kInf) & (fabs(tl) < kInf) & (fabs(tb) < kInf) & (fabs(tm) < kInf));
         const bool allFinite = S1_BALLOT(!(finA & finT & (eq == eq))) == 0;
         const double eMax = grp_max<8>(eq);
         const double bandG = eMax * 0x1p-44;
         const bool standing = accOn && jv && !useA && !(fabs(thD2) < thrA);
         double xForce = kInf;
         if (S1_BALLOT(standing)) xForce = grp_min<8>(standing ? amaxj / fabs(thD2) : kInf);
         // magnitudes far inside the normal range; NaNs fail every test
         const bool sane = allFinite & (eMax == eMax) & (cTop == cTop) & (eMax > 1e-100) & (eMax < 1e100) & (cTop > 1e-100) & (cTop < 1e50);
         // d(c) > 0 <=> violated, |d(c)| > bandG <=> certain: -g(c) of the approximate check, +inf where a standing joint forbids c
         auto dOf = [&](double c) __attribute__((always_inline)) -> double {
            const double x = c * c;
            const double tt = tb * c + tm * x;
            double U = vmin_f64(aa - ma * x, tu - tt), Lw = vmax_f64(-aa - ma * x, tl - tt);
            grp_min_max<8>(U, Lw);
            const double g = vmin_f64(U, sddotMax) - vmax_f64(Lw, -sddotMax);
            return (x > xForce) ? kInf : -g;
         };
         BK_TICK(tq1);
         BK_ACC(cyP1, tp2, tq1);
         bool expectEnd = false;
         if (S1_UNI(sane))
         {
            int it = S1_FIRST(nIter);
            bool inBand = false;
            // the search for a first feasible speed (ba.cpp:1281-1285)
#pragma unroll 1
            for (; it < 90; ++it)
            {
               const double c = sdotTry, d = dOf(c);
               inBand = !((fabs(d) > bandG) & (c > 1e-100));
               if (S1_UNI(inBand | !(d > 0.0))) break;
               lowFact *= 2.0;
               sdotH = c;
               sdotL = dmax(.999 * 0.0, (1.0 - lowFact) * c);
               sdotTry = .5 * (sdotH + sdotL);
            }
            if (!S1_UNI(inBand) && it < 90)
            {
               // the first feasible speed: ba.cpp:1294 compares it with sdotGood = 0 and goes on; then the plain bisection
               sdotGood = sdotTry; nGood = 1; sdotL = sdotTry;
               ++it;
               sdotTry = .5 * (sdotH + sdotL);
               bool goesOn = true;
               const double convThr = 1e-3 * (1.0 + 3e-14);
#pragma unroll 1
               for (; it < 90; ++it)
               {
                  const double c = sdotTry, d = dOf(c);
                  const bool viol = d > 0.0;
                  inBand = !(fabs(d) > bandG);
                  goesOn = viol | (fabs(c - sdotL) > convThr * c);
                  if (S1_UNI(inBand | !goesOn)) break;
                  sdotH = viol ? c : sdotH;
                  sdotL = viol ? sdotL : c;
                  sdotTry = .5 * (sdotH + sdotL);
               }
               sdotGood = sdotL;
               expectEnd = !S1_UNI(inBand | goesOn);
            }
            nIter = it;
         }
         BK_TICK(tq2);
         BK_ACC(cyP2, tq1, tq2);
         if (expectEnd)
         {
            const double c = sdotTry;
            const bool violC = verify(c);
            const bool convC = ratio_lt(fabs(c - sdotGood), c, sdotErrThresh);
            BK_TICK(tq3);
            BK_ACC(cyP3, tq2, tq3);
            if (S1_UNI(!violC && convC))
            {
#ifdef BK_PROFILE_SECTIONS
               ++nAcc;
#endif
               sdotCur = c;                                        // ba.cpp:1296-1302
               sddot = (DIR == 1) ? sddotH : sddotL;
               BK_TICK(tpp);
               BK_ACC(cyD, tp2, tpp);
               return;
            }
         }
      }
#endif
      while (!over)
      {
#ifdef BK_PROFILE_SECTIONS
         ++nPass;
#endif
         // the loop's counters are the same in every lane: say so, and every branch on them is a scalar branch
         nGood = S1_FIRST(nGood);
         nIter = S1_FIRST(nIter);
         const double c0 = sdotTry;
         double cand1, cand2, cand3;
         if (nGood == 0)
         {
            double lf = lowFact * 2.0;
            cand1 = .5 * (c0 + dmax(.999 * 0.0, (1.0 - lf) * c0));
            lf *= 2.0;
            cand2 = .5 * (cand1 + dmax(.999 * 0.0, (1.0 - lf) * cand1));
            lf *= 2.0;
            cand3 = .5 * (cand2 + dmax(.999 * 0.0, (1.0 - lf) * cand2));
         }
         else
         {
            cand1 = .5 * (c0 + sdotL); // next midpoint if c0 is violated
            cand2 = .5 * (sdotH + c0); // next midpoint if c0 is feasible
            cand3 = c0;
         }
         const double mine = (cslot == 0) ? c0 : (cslot == 1) ? cand1 : (cslot == 2) ? cand2 : cand3;
         const bool violMine = verify(mine); // this slot's sddotL / sddotH stay in its lanes
         const unsigned ballot = (unsigned)S1_BALLOT(violMine);

         // The replay of the reference's loop for this pass as a walk over ballot bits instead of runs of iterate(): whatever an
         // iteration tests about its candidate -- the convergence test of a feasible one (ba.cpp:1294), the collapsed-bracket
         // test of a violated one while no feasible point is known (ba.cpp:1311-1315) -- is evaluated by the candidate's own
         // slot beside the constraint check, all four at once.  Same updates, same order, same values; the generic replay
         // below keeps what can end in a failure exit (iteration count near 100, a negative candidate, a collapsed bracket).
         if (S1_FIRST((int)(nIter <= 96 && !(c0 < 0.0) && !(sdotL < 0.0))))
         {
            if (nGood != 0)
            {
               // plain bisection (ba.cpp:1286-1303): c0, then cand1 (c0 violated) or cand2 (c0 feasible); a feasible candidate is
               // compared with the last feasible point before it: sdotGood for c0 and cand1, c0 for cand2
               const double prevGood = (cslot == 2) ? c0 : sdotGood;
               const bool convMine = ratio_lt(fabs(mine - prevGood), mine, sdotErrThresh) || mine < 0.0;
               const unsigned conv = (unsigned)S1_BALLOT(convMine);
               int k2;
               if (ballot & 1u) { sdotH = c0; k2 = 1; }                       // c0 violated: ba.cpp:1278-1280
               else
               {
                  sdotGood = c0; ++nGood;
                  if (conv & 1u) { sdotCur = c0; lastSlot = 0; fin = true; over = true; break; } // ba.cpp:1294-1303
                  sdotL = c0; k2 = 2;
               }
               ++nIter;
               const double m = (k2 == 1) ? cand1 : cand2;                  // == .5 * (sdotH + sdotL), the value iterate() would ask for
               const unsigned bit = 1u << (8 * k2);
               lastSlot = k2;
               if (ballot & bit) sdotH = m;
               else
               {
                  sdotGood = m; ++nGood;
                  if (conv & bit) { sdotCur = m; fin = true; over = true; break; }
                  sdotL = m;
               }
               ++nIter;
               sdotTry = .5 * (sdotH + sdotL);
               continue;
            }
            // the search for a first feasible point (ba.cpp:1281-1285): c0, cand1, cand2, cand3 as long as they are violated,
            // the bracket shrinking below each; the first feasible one starts the plain bisection
            const double lfMine = lowFact * (double)(2 << cslot);
            const double shrunkMine = dmax(.999 * 0.0, (1.0 - lfMine) * mine);
            const bool convMine = ratio_lt(fabs(mine - sdotGood), mine, sdotErrThresh) || mine < 0.0;
            const bool badMine = mine < 0.0 || ratio_lt(mine - shrunkMine, mine, 1e-20);  // a failure exit if this one is violated
            const unsigned conv = (unsigned)S1_BALLOT(convMine);
            if ((((unsigned)S1_BALLOT(badMine)) & ballot & 0x01010101u) == 0u)
            {
               bool ended = false;
#pragma unroll
               for (int d = 0; d < 4; ++d)
               {
                  if (ended) break;
                  const double m = (d == 0) ? c0 : (d == 1) ? cand1 : (d == 2) ? cand2 : cand3;
                  const unsigned bit = 1u << (8 * d);
                  lastSlot = d;
                  if (ballot & bit)
                  {
                     lowFact *= 2.0;                                   // ba.cpp:1281-1285
                     sdotH = m;
                     sdotL = dmax(.999 * 0.0, (1.0 - lowFact) * m);
                     ++nIter;
                  }
                  else
                  {
                     sdotGood = m; nGood = 1;
                     if (conv & bit) { sdotCur = m; fin = true; over = true; }
                     else { sdotL = m; ++nIter; }
                     ended = true;
                  }
               }
               if (over) break;
               sdotTry = .5 * (sdotH + sdotL);
               continue;
            }
         }

         int k = 0;
#pragma unroll 1
         for (int consumed = 0; consumed < 4; ++consumed)
         {
            lastSlot = k;
            const bool stop = iterate((ballot >> (8 * k)) & 1u);
            // was the next value evaluated in this pass?
            k = (sdotTry == cand1) ? 1 : (sdotTry == cand2) ? 2 : (sdotTry == cand3 && nGood == 0) ? 3 : -1;
            over = S1_FIRST((int)stop) != 0;
            if (over || S1_FIRST((int)(k < 0))) break;
         }
      }
      fin = S1_FIRST((int)fin) != 0;
      failed = !fin;
      BK_TICK(tp3);
      BK_ACC(cyD, tp2, tp3);
      if (failed)
      {
         status |= BATOTP_ST_BISECT_FAIL;
         nfail++;
         return;
      }
      const int src = (lane & 32) + 8 * lastSlot;
      sddotH = __shfl(sddotH, src);
      sddotL = __shfl(sddotL, src);
      sddot = (DIR == 1) ? sddotH : sddotL;
   };

   // ---- bootstrap, ba.cpp:1021-1041 ------------------------------------------------------------------------------------
   const double sLast = (DIR == 1) ? sEnd : 0.0;
   double s0v = (DIR == 1) ? 0.0 : sEnd, s6v = 0;
   double v0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
   double w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0;
   double sCur = s0v;
   sdotCur = 0;
   accelPt(sCur, w0);
   v0 = .1 * h * w0;
   sdotMin = v0;
   sdotLim(sCur, v0);
   sdotMin = v0;
   sdotCur = v0;
   accelPt(sCur, w0);
   v0 = sdotCur;
   sdotLim(sCur, v0);

   double sPrev = s0v, sdPrev = v0;
   double sCurPt = s0v, sdCurPt = v0;
   if (writer) out[DIR == 1 ? 0 : cap - 1] = make_double2(s0v, v0);
   const double floorV = 0.0 / absh; // ba.cpp:1050-1051: dsMinV == 0

   // one stage of ba.cpp:1068-1094 with its tableau column written out (ba.cpp:58-63); SDOT / SDDOT are the partial sums
#define S1_STAGE(SDOT, SDDOT, VST, WST)                                   \
   {                                                                      \
      const double sN = s0v + h * (SDOT);                                 \
      double vN = v0 + h * (SDDOT);                                       \
      vN = dmax(vN, floorV);                                              \
      sCur = sN;                                                          \
      BK_TICK(ta0);                                                       \
      sdotLim(sN, vN);                                                    \
      BK_TICK(ta1);                                                       \
      BK_ACC(cyA, ta0, ta1);                                              \
      sdotCur = vN;                                                       \
      double wN = WST; /* kept when the bisection fails, ba.cpp:1091 */   \
      accelPt(sN, wN);                                                    \
      VST = sdotCur;                                                      \
      WST = wN;                                                           \
   }

   int64_t nPts = 0, i = 1;
   // BATOTP_F_CURVES_IN_PLACE: see k_sweep (the forward curve overwrites reverse points its cursor has left behind)
   const int64_t revStart = (DIR == 1 && a.fwd == a.rev) ? cap - (int64_t)nMvc : ((int64_t)1 << 62);
   unsigned endStatus = 0;
   bool done = false;
#ifdef BK_PROFILE_SECTIONS
   const unsigned long long tstart = __builtin_readcyclecounter();
#endif
   while (!done)
   {
      if (S1_UNI(i >= cap || i + 64 >= revStart + (int64_t)segMVC)) { endStatus = BATOTP_ST_CAPACITY; break; }
      if (DIR == 1) mvcWalk(s0v + h * v0); // Euler predictor, ba.cpp:1055-1065: only the move of the reverse-curve cursor survives

      S1_STAGE(BK_B00 * v0, BK_B00 * w0, v1, w1)
      S1_STAGE(BK_B01 * v0 + BK_B11 * v1, BK_B01 * w0 + BK_B11 * w1, v2, w2)
      S1_STAGE(BK_B02 * v0 + BK_B12 * v1 + BK_B22 * v2, BK_B02 * w0 + BK_B12 * w1 + BK_B22 * w2, v3, w3)
      S1_STAGE(BK_B03 * v0 + BK_B13 * v1 + BK_B23 * v2 + BK_B33 * v3, BK_B03 * w0 + BK_B13 * w1 + BK_B23 * w2 + BK_B33 * w3, v4, w4)
      S1_STAGE(BK_B04 * v0 + BK_B14 * v1 + BK_B24 * v2 + BK_B34 * v3 + BK_B44 * v4,
               BK_B04 * w0 + BK_B14 * w1 + BK_B24 * w2 + BK_B34 * w3 + BK_B44 * w4, v5, w5)
      S1_STAGE(BK_B05 * v0 + BK_B15 * v1 + BK_B25 * v2 + BK_B35 * v3 + BK_B45 * v4 + BK_B55 * v5,
               BK_B05 * w0 + BK_B15 * w1 + BK_B25 * w2 + BK_B35 * w3 + BK_B45 * w4 + BK_B55 * w5, v6, w6)
      s6v = sCur;

      // FSAL shift and publish, ba.cpp:1096-1100
      s0v = s6v; v0 = v6; w0 = w6;
      sPrev = sCurPt; sdPrev = sdCurPt;
      sCurPt = s0v; sdCurPt = v0;
      if (writer) out[DIR == 1 ? i : cap - 1 - i] = make_double2(s0v, v0);

      if (S1_UNI(sCur * DIR > sLast)) { nPts = i + 1; done = true; }                        // ba.cpp:1109-1115
      else if (S1_UNI(i > maxIntegSteps)) { endStatus = BATOTP_ST_MAX_INTEG_TIME; break; }    // ba.cpp:1117-1122
      else ++i;
   }
#undef S1_STAGE
#ifdef BK_PROFILE_SECTIONS
   if (writer)
   {
      const unsigned long long tend = __builtin_readcyclecounter();
      double *q = a.prof + 8 * p;
      q[0] = (double)cyA; q[1] = (double)cyB; q[2] = (double)cyC; q[3] = (double)cyD; q[4] = (double)(tend - tstart);
      q[5] = (double)nStage; q[6] = (double)nBis; q[7] = (double)nPass;
      if (a.B == 1) { q[8] = (double)cyP1; q[9] = (double)cyP2; q[10] = (double)cyP3; q[11] = (double)nAcc; }
   }
#endif

   status |= endStatus;
   if (endStatus != 0)
   {
      if (writer)
      {
         if (DIR == 1) { r->n_fwd = 0; r->steps_fwd = i; r->t_total = 0; r->status_fwd = status; r->n_bisect_fail_fwd = nfail; }
         else { r->n_rev = 0; r->steps_rev = i; r->t_rev = 0; r->status_rev = (r->status_rev & BATOTP_ST_SEG_ERROR) | status; r->n_bisect_fail_rev = nfail; }
      }
      return;
   }

   // end snap onto sLast, ba.cpp:1132-1134; forward: last sdot <- reverse curve's last sdot, ba.cpp:1140
   {
      const double sRat = (sLast - sPrev) / (sCurPt - sPrev);
      sdCurPt = sdPrev + sRat * (sdCurPt - sdPrev);
      sCurPt = sLast;
      if (DIR == 1) sdCurPt = mvc[(nMvc - 1) * 2 + 1];
      if (writer) out[DIR == 1 ? nPts - 1 : cap - nPts] = make_double2(sCurPt, sdCurPt);
   }
   const double tElapsed = absh * (double)(nPts - 1); // ba.cpp:1112
   int64_t nOut = nPts;

   if (nPts < 4 && writer)
   {
      // ba.cpp:1171-1184: re-interpolate linearly in time to four points
      double ps[3], pd[3], tIn[3];
      for (int k = 0; k < (int)nPts; ++k)
      {
         const double2 q = (k == (int)nPts - 1) ? make_double2(sCurPt, sdCurPt) : out[DIR == 1 ? k : cap - 1 - k];
         ps[k] = q.x; pd[k] = q.y;
         tIn[k] = absh * (double)k;
      }
      if (DIR != 1)
      {
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
      for (int k = 0; k < 4; ++k) out[DIR == 1 ? k : cap - 4 + k] = make_double2(ns[k], nd[k]);
   }
   if (nPts < 4) { status |= BATOTP_ST_SHORT; nOut = 4; }

   if (writer)
   {
      if (DIR == 1)
      {
         r->n_fwd = nOut; r->steps_fwd = nPts - 1; r->t_total = tElapsed; r->status_fwd = status; r->n_bisect_fail_fwd = nfail;
      }
      else
      {
         r->n_rev = nOut; r->steps_rev = nPts - 1; r->t_rev = tElapsed;
         r->status_rev = (r->status_rev & BATOTP_ST_SEG_ERROR) | status; r->n_bisect_fail_rev = nfail;
      }
   }
}

} // namespace bk
