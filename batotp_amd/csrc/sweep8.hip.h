// sweep8.hip.h -- K4 for large batches of velocity / acceleration-only problems (round 4): the flat stage / bisection loop of
// k_sweep<8, FEAT <= 0, true, true> (kernels.hip.h) written for the instruction count.
//
// Same decomposition: a path = 8 lanes (lane j = joint j), up to 8 paths per wavefront, every path of a wavefront is either
// waiting for its next stage or inside a constraint check; a pass of the loop runs the stage prologue for the waiting paths
// once `hold`/8 of the live paths wait, then one constraint check + bisection update for every path inside one.  Same
// arithmetic: every fp64 operation of BA::sweep (ba.cpp:979-1195), sdotLim (:1204-1236), evalsdot (:1590-1607), updateCurSeg
// (:1617-1652), applyAccelConstraintsBisectionPt (:1248-1332), evalSplinePartials (:1341-1413) and
// verifySecondOrderConstraints (:1514-1534, joint acceleration family) in the reference's order, uncontracted.
//
// What differs from the general kernel is the code around the arithmetic (profiles/r03_c: a third of its loop was exec-mask
// bookkeeping the compiler generated for nested divergent branches and loops with several exits):
//   * one level of divergence per block (waiting paths / checking paths); everything inside is straight-line select code;
//   * the segment walks are wavefront-uniform loops ("while any lane moved") over select-form steps;
//   * rare cases (a joint that stands still, a threshold test that needs its true quotient, a segment change, the end of a
//     step) sit behind wavefront-uniform branches on ballots and cost nothing when no lane needs them;
//   * the reverse-curve segment of the forward sweep is kept in registers and updated by ONE 16-byte load per move;
//   * curve points leave through LDS, four at a time: one 64-byte store per path and four steps instead of four 16-byte
//     stores (the reverse sweep's 16 384 descending streams lost their partly written lines from the L2 between the stores:
//     3.0x the curve bytes written, profiles/r03_e).
// Results are bit-identical to k_sweep's (every sweep test runs both; tests/test_gpu_parity.py, tests/test_gpu_fuzz.py).
#pragma once
#include "kernels.hip.h"

// diagnostic build (-DS8_PROFILE): what a wavefront of k_sweep8 spends its passes and cycles on, 16 doubles per wavefront in
// SweepArgs::prof (tools/sweep8_sections.py); no effect otherwise
#ifdef S8_PROFILE
#define S8_CNT(k, v) pc[k] += (double)(v)
#define S8_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define S8_CYC(k, a, b) pc[k] += (double)((b) - (a))
#else
#define S8_CNT(k, v)
#define S8_TICK(var)
#define S8_CYC(k, a, b)
#endif

namespace bk
{

constexpr int S8_BLOCK = 256;
__device__ __forceinline__ bool s8_div_window_fwd(double x) { return sdiv_window(x); }
// "does any active lane ...": the i1 ballot intrinsic (HIP's __ballot goes through an integer compare: a v_cndmask + v_cmp per use)
#define S8_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0)
#define S8_RARE(x) __builtin_expect(S8_ANY(x), 0) // a guard whose block is off the straight-line code
#ifndef S8_TAB_FAST
#define S8_TAB_FAST 1   // tableau combination without selects while every stage value is finite
#endif
#ifndef S8_LDS_STAGES
#define S8_LDS_STAGES 1 // the stage values (v_k, w_k), k = 1..5, of a path live in LDS instead of registers + 20 selects per stage
#endif
#ifndef S8_WALK_PRECHECK
#define S8_WALK_PRECHECK 1 // skip the knot-cursor walk when every path is still inside its segment
#endif
#ifndef S8_PREFETCH
#define S8_PREFETCH 1   // the knot the cursor will need next is loaded one segment change ahead (compact pairs).  (The same for the next
                        // point of the reverse curve in the forward sweep measured 5 % slower -- 440 against 419 ms -- and is not done.)
#endif
#ifndef S8_TAU_RCP
#define S8_TAU_RCP 0 // tau = (sCur - sSeg) / (sNext - sSeg) and the reverse-curve tau of the forward sweep through the refined reciprocal of
                     // their segment's width (device_math.h: sdiv_rcp / sdiv_by, the bits of `/` inside the window and for a numerator of
                     // +0), kept while the cursor stays on the segment: 3 instead of ~30 instructions per stage on the dependent chain.
                     // Round 5 gave the literal quotient (an operand outside the window) a wavefront-uniform guard of its own and measured
                     // 5-7 % SLOWER; round 6 let it share the guard of the segment change, which a third of the prologues enter anyway:
                     // SLOWER again (reduced batch, same box: reverse 626 against 601 ms, forward 417 against 386 ms; profiles/r06_b_*).
                     // The division is not what these wavefronts wait for.  Kept as an option, bit-identical.
#endif
#ifndef S8_SPEC_MID
#define S8_SPEC_MID 0  // s8_certify's bisection replay forms both possible next candidates beside the test: a shorter dependent chain, four more
                       // fp64 instructions per iteration -- bit-identical, reverse sweep 2.3 % SLOWER (profiles/r06_p_*): the block is issue-bound
#endif
#ifndef S8_FF
#define S8_FF 1      // the certified fast-forward of the bisection (s8_certify), both directions (batotp_hip_set_fast_forward: bit 0 forward, bit 1 reverse)
#endif
#ifndef S8_FF_REV
#define S8_FF_REV 1  // 0: the reverse kernel carries no certificate code at all (the kernel of rounds 4 and 5, for A/B runs)
#endif
#ifndef S8_CERT_PHASE
#define S8_CERT_PHASE 1 // reverse sweep: the certificate as a phase of its own, served in batches (0: in the check block, per arriving path)
#endif

// (num / den) < thr as ratio_lt (kernels.hip.h) decides it, in two parts: the product form, and whether it was decisive
__device__ __forceinline__ bool s8_ratio_lt_fast(double num, double den, double thr, bool &decided)
{
   const double p = thr * den;
   // thr > 0: p > 1e-290 implies den > 0, and for a positive divisor the product form is right for a numerator of either sign
   const bool ok = (p > 1e-290) & (p < 1e290);
   const bool lt = num < p * (1.0 - 1e-14), gt = num > p * (1.0 + 1e-14);
   decided = ok & (lt | gt);
   return lt;
}
// The two threshold tests of one bisection pass divide by the same den: num1 / den < THR1 and num2 / den < THR2 with the
// thresholds of ba.cpp:1294 and 1313 (1e-3, 1e-20).  A test is decided when the numerator lies outside the band
// den * thr * (1 -+ 1e-14): the quotient's rounding error (2^-53) cannot carry it across thr from there.  (The bounds of the band
// are single products with constants; where exactly the band ends is immaterial, inside it the literal quotient decides.)
// den in (1e-260, 1e260): positive, and every product is a normal number.
__device__ __forceinline__ void s8_ratio_lt_pair(double num1, double num2, double den, bool &lt1, bool &dec1, bool &lt2, bool &dec2)
{
   constexpr double T1 = .001, T2 = 1e-20;
   const bool ok = (den > 1e-260) & (den < 1e260);
   lt1 = num1 < den * (T1 * (1.0 - 1e-14));
   lt2 = num2 < den * (T2 * (1.0 - 1e-14));
   dec1 = ok & (lt1 | (num1 > den * (T1 * (1.0 + 1e-14))));
   dec2 = ok & (lt2 | (num2 > den * (T2 * (1.0 + 1e-14))));
}

// numerator of a quotient by a cached reciprocal: inside the window, or +0 (0 * r = +0, the remainder fma(-den, +0, +0) = +0, the result
// fma(+0, r, +0) = +0 = (+0) / den for the positive divisors this is used with; a numerator of -0 would come out as +0 and takes the
// literal form)
__device__ __forceinline__ bool s8_num_ok(double x) { return s8_div_window_fwd(x) | (__double_as_longlong(x) == 0ll); }
// (the shared refined reciprocal of theta' -- sdiv_window / sdiv_rcp / sdiv_by -- lives in device_math.h.  In k_sweep1, the
// one-path-per-wavefront kernel, the same technique measured 3-10 % SLOWER -- cfg 4: 961 against 874 ms -- and is not used there:
// its window tests and ballot guards cost a lone wavefront more than the shorter quotients save; profiles/r04_b_*)
constexpr double S8_DIV_LO = SDIV_LO, S8_DIV_HI = SDIV_HI;
__device__ __forceinline__ bool s8_div_window(double x) { return sdiv_window(x); }
__device__ __forceinline__ double s8_rcp_refined(double den) { return sdiv_rcp(den); }
__device__ __forceinline__ double s8_div_by(double num, double den, double r) { return sdiv_by(num, den, r); }

// known-answer test of the shared-reciprocal division against `/` (batotp_hip_fp64_kat)
__global__ void k_kat_sdiv(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ q, int *__restrict__ inWindow)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const bool ok = s8_div_window(a[i]) & s8_div_window(b[i]);
   inWindow[i] = ok ? 1 : 0;
   q[i] = ok ? s8_div_by(a[i], b[i], s8_rcp_refined(b[i])) : a[i] / b[i];
}

// ---- CERTIFIED FAST-FORWARD of the bisection on the 8 lanes of a path (the certificate of k_sweep1, sweep1.hip.h: its derivation
// and error analysis are there) ---------------------------------------------------------------------------------------------
// Called for paths whose first check of the stage was violated, after the loop's first update (ba.cpp:1281-1285: nIter = 1,
// sdotH = the first candidate, sdotTry = the second one).  In x = sdot^2 every bound of the check is a line: joint q allows
// sddot in [-au_q - m_q x, au_q - m_q x] with au = amax_q / |theta'_q|, m = theta''_q / theta'_q, and [-sddotMax, sddotMax] is the
// line au = sddotMax, m = 0.  They stop intersecting at
//    x* = min over the pairs of lines (i, j) of (au_i + au_j) / |m_i - m_j|
// (of the two orientations of a pair -- i above, j below, or the other way round -- the one with a positive slope difference
// crosses), and a candidate c of the loop is violated iff c^2 > x* -- for certain when |c^2 - x*| exceeds the band 2^-40 R x*
// (R = E / Smin: the rounding errors of the check against the width of the interval at x = 0).  The loop of ba.cpp:1267-1321 is
// advanced with its own update statements through every iteration whose outcome is certain and that neither ends it nor can take
// a failure exit; it stops in front of the first candidate inside the band, or the one that would end the loop: that one gets
// the real check in the next pass.  Anything unusual (a standing joint's threshold inside the band, non-finite or extreme values)
// leaves the state as it is.
// Round 6: written for the instruction count -- theta' already has its refined reciprocal (rD: two Newton steps, 2 eps, the
// accuracy the analysis asks for), the partner lines arrive by DPP moves (quad permutations, the half-row mirror and their
// products reach every lane of the group) instead of LDS-crossbar shuffles, a pair costs one reciprocal whatever its orientation
// (|m_i - m_j|), lines that do not exist are the line u = 1e300, m = 0 (every quantity stays finite without a select per pair),
// and the standing joints' true quotient sits behind a guard.  All lanes of a group are active together (the caller's condition
// is a path-level value), and the permutations never leave the group.
constexpr int DPP_QUAD_XOR3 = 0x1B; // quad_perm [3,2,1,0]
__device__ __forceinline__ double s8_rcp2(double d)
{
   double r = __builtin_amdgcn_rcp(d);
   r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
   return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
// max(|x|, floor): one instruction
__device__ __forceinline__ double s8_absmax(double x, double floorv)
{
   double r;
   asm("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(x), "v"(floorv));
   return r;
}
__device__ __forceinline__ void s8_certify(bool jOn, double thD, double thD2, double rD, bool rOk, double amaxq, double thrV, double thrA,
                                           double sddotMax, double &lowFact, double &sdotH, double &sdotL, double &sdotTry,
                                           double &sdotGood, int &nGood, int &nIter)
{
   constexpr double BIG = 1e300, TINY = 1e-300;
   const double xTop = sdotH * sdotH;                 // the first candidate: no later one is larger
   const bool use = jOn && !(fabs(thD) < thrV);       // a moving joint (ba.cpp:1526-1531)
   const double aur = amaxq * fabs(rD), mr = thD2 * rD;
   const double er = aur + fabs(mr) * xTop;
   // a line with a coefficient that is not an ordinary number stops the fast-forward (eMax = inf fails `sane`); NaNs fail every test
   const bool lineOk = rOk & (aur < 1e100) & (fabs(mr) < 1e100) & (er < 1e100);
   const bool have = use & lineOk;
   const double au = have ? aur : BIG;
   const double mj = have ? mr : 0.0;
   const double ej = use ? (lineOk ? er : kInf) : 0.0;
   const double uMin = grp_min<8>(au);
   const double eMax = grp_max<8>(ej);
   const double sMin2 = dmin(uMin, sddotMax);         // half the width of the interval at x = 0
   const bool standing = jOn && !use && !(fabs(thD2) < thrA);
   double xForce = kInf;
   if (S8_RARE(standing)) xForce = grp_min<8>(standing ? amaxq / fabs(thD2) : kInf);   // the check's own quotient (ba.cpp:1519-1524)
   // this lane's line against [-sddotMax, sddotMax] ...
   double xs = (au + sddotMax) * s8_rcp2(s8_absmax(mj, TINY));
   // ... and against the line of every other lane of the group
#define S8_PAIR(AUI, MI) xs = vmin_f64(xs, (au + (AUI)) * s8_rcp2(s8_absmax(mj - (MI), TINY)))
   {
      const double a1 = dpp_mov<DPP_QUAD_XOR1>(au), m1 = dpp_mov<DPP_QUAD_XOR1>(mj);
      S8_PAIR(a1, m1);
      const double a2 = dpp_mov<DPP_QUAD_XOR2>(au), m2 = dpp_mov<DPP_QUAD_XOR2>(mj);
      S8_PAIR(a2, m2);
      const double a3 = dpp_mov<DPP_QUAD_XOR3>(au), m3 = dpp_mov<DPP_QUAD_XOR3>(mj);
      S8_PAIR(a3, m3);
      const double a7 = dpp_mov<DPP_ROW_HALF_MIRROR>(au), m7 = dpp_mov<DPP_ROW_HALF_MIRROR>(mj);   // lane ^ 7
      S8_PAIR(a7, m7);
      const double a6 = dpp_mov<DPP_QUAD_XOR1>(a7), m6 = dpp_mov<DPP_QUAD_XOR1>(m7);               // lane ^ 6
      S8_PAIR(a6, m6);
      const double a5 = dpp_mov<DPP_QUAD_XOR2>(a7), m5 = dpp_mov<DPP_QUAD_XOR2>(m7);               // lane ^ 5
      S8_PAIR(a5, m5);
      const double a4 = dpp_mov<DPP_QUAD_XOR3>(a7), m4 = dpp_mov<DPP_QUAD_XOR3>(m7);               // lane ^ 4
      S8_PAIR(a4, m4);
   }
#undef S8_PAIR
   xs = grp_min<8>(xs);
   const double xstar = dmin(xs, 4.0 * xTop);          // beyond 4 xTop: "never violated by the lines" just as well
   const double R = eMax * s8_rcp2(sMin2 > 0.0 ? sMin2 : 1.0);
   const double band = (R * 0x1p-40) * xstar;
   // a standing joint's threshold below the band around x* decides alone, and exactly; one above the band never matters; one
   // inside the band: no fast-forward
   const bool forceFirst = xForce < xstar - band;
   const double xThr = forceFirst ? xForce : xstar;
   const double bandThr = forceFirst ? -1.0 : band;
   const bool sane = (sddotMax == sddotMax) & (eMax == eMax) & (xstar == xstar) & (R == R) & (R < 0x1p30) & (sMin2 > 1e-100) & (eMax < 1e100) &
                     (xTop > 1e-100) & (xTop < 1e100) & (xstar > 1e-100) & (forceFirst | (xForce > xstar + band));
   if (sane)
   {
      // Both loops of the replay are wavefront-uniform loops ("while any lane goes on") with BRANCH-FREE bodies: a per-lane loop with a
      // break, or an exec-masked update inside a uniform loop, makes the compiler structurize the whole loop as divergent and spend
      // twice as many scalar instructions on mask bookkeeping as the loop has vector instructions (18 against 7 per iteration in the
      // first round-6 build).  A lane that has stopped keeps its state through selects; its tests are re-evaluated on that unchanged
      // state and give what they gave.  Invariant used below: sdotTry == .5 * (sdotH + sdotL) in every lane at every point (the check
      // block's update, ba.cpp:1320, leaves it so, and every update here ends with that statement), so the midpoint needs no select.
      int it = nIter;
      bool inBand = false, more = true;
      // the search for a first feasible speed (ba.cpp:1281-1285): the bracket shrinks below every violated candidate
      for (;;)
      {
         const double c = sdotTry, d = c * c - xThr;   // c * c: sdotSQ of the check
         inBand = !((fabs(d) > bandThr) & (c > 1e-100));
         more = more & !inBand & (d > 0.0) & (it < 90);
         if (!S8_ANY(more)) break;
         const double lf2 = lowFact * 2.0;
         const double lo = dmax(.999 * 0.0, (1.0 - lf2) * c);
         lowFact = more ? lf2 : lowFact;
         sdotH = more ? c : sdotH;
         sdotL = more ? lo : sdotL;
         sdotTry = .5 * (sdotH + sdotL);
         it += more ? 1 : 0;
      }
      // sdotTry is feasible for certain and the first such speed: ba.cpp:1294 compares it with sdotGood = 0 and goes on.  From
      // here the plain bisection (ba.cpp:1286-1303): sdotGood == sdotL throughout
      const bool bisect = !inBand & (it < 90);
      sdotGood = bisect ? sdotTry : sdotGood;
      nGood = bisect ? 1 : nGood;
      sdotL = bisect ? sdotTry : sdotL;
      it += bisect ? 1 : 0;
      sdotTry = .5 * (sdotH + sdotL);
      // ba.cpp:1294 is false for certain when |c - sdotGood| > 1e-3 c (1 + 3e-14); otherwise this candidate is, or may be, the
      // last one and gets the real check and the real test
      const double convThr = 1e-3 * (1.0 + 3e-14);
      more = bisect;
      for (;;)
      {
         const double c = sdotTry, d = c * c - xThr;
#if S8_SPEC_MID
         // both possible next candidates, computed beside the test instead of behind it (the same sums as ba.cpp:1320 forms after the
         // update: .5 * (c + sdotL) when c becomes sdotH, .5 * (sdotH + c) when it becomes sdotL; a lane that has stopped keeps c, which
         // IS .5 * (sdotH + sdotL) by the invariant): the dependent chain of an iteration is square - subtract - compare - select
         // instead of square - subtract - compare - select - add - halve
         const double nextV = .5 * (c + sdotL), nextG = .5 * (sdotH + c);
#endif
         const bool viol = d > 0.0;
         const bool goesOn = viol | (fabs(c - sdotL) > convThr * c);
         more = more & (fabs(d) > bandThr) & goesOn & (it < 90);
         if (!S8_ANY(more)) break;
         sdotH = (more & viol) ? c : sdotH;
         sdotL = (more & !viol) ? c : sdotL;
#if S8_SPEC_MID
         sdotTry = more ? (viol ? nextV : nextG) : c;
#else
         sdotTry = .5 * (sdotH + sdotL);
#endif
         it += more ? 1 : 0;
      }
      sdotGood = bisect ? sdotL : sdotGood;
      nIter = it;
   }
}

// G lanes per path (8: one joint per lane, 4: joints j and j + 4 in lane j), up to 64 / G paths per wavefront
template <int G, int FEAT, int DIR>
__global__ void __launch_bounds__(S8_BLOCK, G == 8 ? 2 : 1) k_sweep8(SweepArgs a)
{
   static_assert(FEAT == -1 || FEAT == 0, "velocity / acceleration-only problems");
   static_assert(G == 8 || G == 4, "lanes per path");
   constexpr int PER = 8 / G;
   __shared__ double lim[6][8];
   __shared__ double rk[7][6];                 // rk[st][k] = weight of stage value k in stage st (column st-1 of ba.cpp:58-63), 0 for k >= st
   __shared__ double2 pts[S8_BLOCK / G][4];    // curve points of a path waiting for their 64-byte store
#if S8_LDS_STAGES
   __shared__ double2 vw[S8_BLOCK / G][8];     // (v_k, w_k) of stage k = 1..5 of every path; slot 0 and 7: writes that must not land
#endif
   if (threadIdx.x < 42)
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
   const int wave = blockIdx.x * (S8_BLOCK / 64) + (threadIdx.x >> 6);
   const int j = lane % G;
   const int slot = lane / G;
   const int pslot = wave * a.ppw + slot;
   if (slot >= a.ppw || pslot >= a.B) return; // whole groups leave together; DPP never crosses groups
   const int p = a.order ? a.order[pslot] : pslot;   // ragged batches: paths of similar length share a wavefront (SweepArgs::order)
   const bool writer = (j == 0);
   const PathInfo pi = a.pinfo[p];
   const int n = (int)pi.n;
   const int64_t cap = a.cap;
   double2 *mypts = pts[threadIdx.x / G];
#if S8_LDS_STAGES
   double2 *myvw = vw[threadIdx.x / G];
   for (int k = j; k < 8; k += G) myvw[k] = make_double2(0.0, 0.0);
#endif

   // bootstrap (ba.cpp:1021-1041) through the general kernel's device functions; the loop below carries its own state
   Pt<G, FEAT, true> t;
   pt_init(t, a.P, pi, a.sC, a.coef, a.km, lim, j, DIR);

   double2 *out = (DIR == 1 ? a.fwd : a.rev) + (int64_t)p * cap; // may alias the reverse curve (curves in place)
   batotp_path_result *__restrict__ r = a.res + p;
   const double2 *mvc = nullptr; // the reverse curve the forward sweep follows
   int nMvc = 0;
   if (DIR == 1)
   {
      const int64_t nRev = r->n_rev;
      if (nRev < 2)
      {
         if (writer) { r->n_fwd = 0; r->steps_fwd = 0; r->t_total = 0; r->status_fwd = r->status_rev | BATOTP_ST_CAPACITY; r->n_bisect_fail_fwd = 0; }
         return;
      }
      mvc = a.rev + (int64_t)p * cap + (cap - nRev);
      nMvc = (int)nRev;
      t.mvc = reinterpret_cast<const double *>(mvc);
      t.nMvc = nMvc;
   }
   // curves in place (kernels.hip.h, BK_CURVE_FULL): slot i must have been left behind by the reverse-curve cursor
   const int64_t revStart = (DIR == 1 && a.fwd == a.rev) ? cap - (int64_t)nMvc : ((int64_t)1 << 62);

   const double absh = pi.integ_res;
   const double h = DIR * absh;
   const int64_t maxIntegSteps = (int64_t)floor(a.P.max_integ_time / pi.integ_res) + 1;
   const double sres = pi.sres_c;
   const double sEnd = sres * (double)(n - 1);
   const double sLast = (DIR == 1) ? sEnd : 0.0;
   double s0v = (DIR == 1) ? 0.0 : sEnd;
   double v0, w0 = 0;
   if (DIR == 1) { t.segC = 0; t.tauC = 0; t.segMVC = 0; t.tauMVC = 0; }
   else { t.segC = n - 2; t.tauC = 1; t.segMVC = n - 2; t.tauMVC = 1; }
   t.sCur = s0v;
   t.sdotCur = 0;
#pragma unroll 1
   for (int pass = 0; pass < 2; ++pass)
   {
      accel_pt(t, j, w0);
      if (pass == 0) { v0 = .1 * h * w0; t.sdotMin = v0; }
      else v0 = t.sdotCur;
      sdot_lim(t, j, v0);
      if (pass == 0) { t.sdotMin = v0; t.sdotCur = v0; }
   }
   const double vBoot = v0;

   // ---- the loop's own state -------------------------------------------------------------------
   // lane constants: joints jj[q] = j + q G of this lane
   bool jOn[PER];
   int jAt[PER];
   double vmax[PER], amax[PER];
   // this lane's coefficients on the cursor's segment: c1, 2 c2, 3 c3, 6 c3 (the products the reference forms first:
   // (3*c3)*tau2, (2*c2)*tau, (6*c3)*tau, ba.cpp:1358-1360); theta', theta'' of the last evaluation point; the refined
   // reciprocal of theta' (s8_rcp_refined) and whether theta' lies in the window in which it may be used
   double c1[PER], c2x2[PER], c3x3[PER], c3x6[PER], thD[PER], thD2[PER], rD[PER];
   double sa[PER]; // sgn(theta') * amax of ba.cpp:1526-1531 where theta' != 0: amax with the sign of theta' (1.0 * amax and -1.0 * amax are exact)
   bool rOk[PER];
#pragma unroll
   for (int q = 0; q < PER; ++q)
   {
      jOn[q] = (j + q * G) < t.nJ;
      jAt[q] = jOn[q] ? j + q * G : 0;
      vmax[q] = t.vmax[q]; amax[q] = t.amax[q];
      c1[q] = jOn[q] ? t.rowTh[q].c1 : 0.0; c2x2[q] = jOn[q] ? 2 * t.rowTh[q].c2 : 0.0;
      c3x3[q] = jOn[q] ? 3 * t.rowTh[q].c3 : 0.0; c3x6[q] = jOn[q] ? 6 * t.rowTh[q].c3 : 0.0;
      thD[q] = t.thD[q]; thD2[q] = t.thD2[q];
      rOk[q] = s8_div_window(thD[q]);
      rD[q] = s8_rcp_refined(thD[q]);
      sa[q] = (thD[q] < 0.0) ? -amax[q] : amax[q];
   }
   const bool accOn = (t.flags & BATOTP_F_JNT_ACC_ON) != 0;
   const double thrV = t.thrV, thrA = t.thrA, vfact = t.vfact, afact = t.afact;
   const double sdotCap = t.sdotCap, sddotMax = t.sddotMax, sdotMin = t.sdotMin;
   const bool capOk = (sddotMax == sddotMax); // the acceleration cap is not a NaN (otherwise every bound takes the literal form)
   const int lastSeg = n - 2;
   const int nIn = t.nIn;
   const double2 *__restrict__ km = t.km;
   const double *__restrict__ coef = t.coef;
   const int rowStride = t.C * 4;
   int seg = t.segC, rowSeg = t.rowSeg;
   // S8_PREFETCH (compact pairs): the knot on the side the cursor moves to (reverse: knot rowSeg, forward: knot rowSeg + 1)
   // and the one beyond it, loaded when the cursor entered the current segment -- a segment change one step further then
   // forms its coefficients from registers and issues the load for the change after it
   double2 kEdge[PER], kPre[PER];
   int preIdx = -(1 << 20); // knot index kPre holds (and kEdge holds preIdx - DIR): none yet
#pragma unroll
   for (int q = 0; q < PER; ++q) { kEdge[q] = make_double2(0, 0); kPre[q] = make_double2(0, 0); }
   double sSeg = sres * (double)seg, sNext = sres * (double)(seg + 1); // sites of the cursor's segment
#if S8_TAU_RCP
   double rSeg = s8_rcp_refined(sNext - sSeg); // refined reciprocal of the segment's width (renewed when the walk ran)
   bool segOk = s8_div_window(sNext - sSeg) & (sNext - sSeg > 0.0);
   double rM = 0;   // the same for the reverse-curve segment of the forward sweep
   bool mOk = false;
   // forward sweep: what the literal form of evalsdot's quotient needs when it is redone behind the guard of the segment change
   double numM = 0, denM = 1, vPre = 0;
   bool fastM = true;
#endif
   // reverse-curve cursor (forward sweep): segment and its two points
   int segM = 0;
   double mS0 = 0, mD0 = 0, mS1 = 0, mD1 = 0;
   if (DIR == 1)
   {
      segM = t.segMVC;
      const double2 qa = mvc[segM], qb = mvc[segM + 1];
      mS0 = qa.x; mD0 = qa.y; mS1 = qb.x; mD1 = qb.y;
#if S8_TAU_RCP
      rM = s8_rcp_refined(mS1 - mS0); mOk = s8_div_window(mS1 - mS0) & (mS1 - mS0 > 0.0);
#endif
   }
   unsigned status = t.status;
   int nfail = t.nfail;
   double sdotCur = t.sdotCur;
   double sddotL = t.sddotL, sddotH = t.sddotH;

   double v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
   double w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0;
   double sPrev = s0v, sdPrev = v0, sCurPt = s0v, sdCurPt = v0; // last two published points, for the end snap
   double sCur = s0v; // traj.sCur

   const int iMax = (maxIntegSteps > 0x3ffffff0) ? 0x3ffffff0 : (int)maxIntegSteps;
   const int capI = (int)cap;
   // first slot the forward curve must not reach: revStart + segM - 64 (in-place curves)
   const int revStartI = (revStart > 0x3fffffff) ? 0x3fffffff : (int)revStart; // (the launcher keeps cap below 2^30)
#define S8_CURVE_FULL(i) ((i) >= capI || (DIR == 1 && (i) + 64 >= revStartI + segM))

   // point 0: straight to HBM, and into the group of four it belongs to
   {
      const int idx0 = (DIR == 1) ? 0 : capI - 1;
      if (writer) { out[idx0] = make_double2(s0v, v0); mypts[idx0 & 3] = make_double2(s0v, v0); }
   }
   int flushedTo = (DIR == 1) ? 0 : capI; // forward: points [0, flushedTo) are in HBM; reverse: points [flushedTo, cap)

   const double floorV = 0.0 / absh; // ba.cpp:1050-1051,1085
   int nPts = 0, i = 1;
   unsigned endStatus = 0;
   // A path starts as if the stage before its first one had just ended (st one lower, PH_ENDED): the stores of that stage
   // go to slots nothing reads, and the loop needs no separate state for the first stage.
   int st = (DIR == 1) ? -1 : 0;
   constexpr int PH_FIRST = 0, PH_ENDED = 1, PH_CHECK = 2, PH_DEAD = 3, PH_CERT = 4;
   // reverse sweep: a path whose first check was violated waits in PH_CERT for the certificate block at the top of the loop, which
   // runs once holdC/8 of the live paths have gathered there (or nothing else can run in this pass)
   constexpr bool CERT = (DIR == -1) && (PER == 1) && (S8_FF != 0) && (S8_FF_REV != 0) && (S8_CERT_PHASE != 0);
   const int holdC = a.holdc;
   int phase = PH_ENDED;
   if (S8_CURVE_FULL(i)) { endStatus = BATOTP_ST_CAPACITY; phase = PH_DEAD; }
   double sN = 0, wN = 0;
   double lowFact = .01, sdotGood = 0, sdotL = 0, sdotH = 0, sdotTry = 0;
   int nGood = 0, nIter = 0;
   bool wild = !(fabs(v0) < kInf) | !(fabs(w0) < kInf); // a non-finite stage value has been kept (sticky; see the tableau combination)
   bool stageFailed = false; // the bisection of the stage that ended failed: sddotArr[st] keeps its previous value (ba.cpp:1091 ignores the code)
   const int hold = a.hold;
#ifdef S8_PROFILE
   // [0] passes of the loop, [1] prologue blocks, [2] paths they served, [3] check blocks, [4] paths inside them, [5] cycles in
   // prologue blocks, [6] cycles in check blocks, [7] cycles in the loop, [8] step-end blocks, [9] segment-change blocks,
   // [10] knot-cursor walks, [11] fast-forward blocks, [12] literal-form (rare) blocks of the check, [13] live paths summed over passes
   double pc[16];
   for (int k = 0; k < 16; ++k) pc[k] = 0;
   const unsigned long long tLoop0 = __builtin_readcyclecounter();
#endif

   for (;;)
   {
      const unsigned long long mAlive = __ballot(phase != PH_DEAD);
      if (mAlive == 0) break;
      const unsigned long long mWait = __ballot(phase < PH_CHECK);
      const bool startNow = (mWait == mAlive) || (__popcll(mWait) * 8 >= __popcll(mAlive) * hold);
      S8_CNT(0, 1); S8_CNT(13, __popcll(mAlive) / G);
      if (CERT)
      {
         const unsigned long long mCert = __ballot(phase == PH_CERT);
         if (mCert != 0)
         {
            // (will a prologue or a check run in this pass?  If not, the waiting certificates are all there is to do)
            const bool progress = (startNow && mWait != 0) || ((mAlive & ~mWait & ~mCert) != 0);
            if (__popcll(mCert) * 8 >= __popcll(mAlive) * holdC || !progress)
            {
               S8_TICK(tF0);
               S8_CNT(11, 1); S8_CNT(15, __popcll(mCert) / G);
               if (phase == PH_CERT)
               {
                  s8_certify(jOn[0], thD[0], thD2[0], rD[0], rOk[0], amax[0], thrV, thrA, sddotMax, lowFact, sdotH, sdotL, sdotTry, sdotGood,
                             nGood, nIter);
                  phase = PH_CHECK;
               }
               S8_TICK(tF1);
               S8_CYC(14, tF0, tF1);
            }
         }
      }
      S8_TICK(tA);
      if (startNow && mWait != 0)
      {
         S8_CNT(1, 1); S8_CNT(2, __popcll(mWait) / G);
         if (phase < PH_CHECK)
         {
            // ---- the stage that ended: keep its values --------------------------------------------
            {
               phase = PH_FIRST;
               const double vN = sdotCur;
               wild |= !(fabs(vN) < kInf) | !(fabs(wN) < kInf);
#if S8_LDS_STAGES
               // slot st keeps (vN, wN); a failed bisection leaves sddotArr[st] as it was (the w half goes to the spare slot 7);
               // stage 6 is kept in registers by the step end below (its writes go to the spare slot 7 as well)
               {
                  double *slotV = reinterpret_cast<double *>(myvw + ((unsigned)st < 6u ? st : 7));
                  double *slotW = reinterpret_cast<double *>(myvw + (((unsigned)st < 6u && !stageFailed) ? st : 7)) + 1;
                  *slotV = vN;
                  *slotW = wN;
               }
#else
               const int stW = stageFailed ? 0 : st; // a failed bisection leaves sddotArr[st] as it was
               v1 = (st == 1) ? vN : v1; w1 = (stW == 1) ? wN : w1;
               v2 = (st == 2) ? vN : v2; w2 = (stW == 2) ? wN : w2;
               v3 = (st == 3) ? vN : v3; w3 = (stW == 3) ? wN : w3;
               v4 = (st == 4) ? vN : v4; w4 = (stW == 4) ? wN : w4;
               v5 = (st == 5) ? vN : v5; w5 = (stW == 5) ? wN : w5;
#endif
               const bool stepEnd = (st == 6);
               st = stepEnd ? st : st + 1;
               if (S8_ANY(stepEnd))
               {
                  S8_CNT(8, 1);
                  if (stepEnd)
                  {
                     // FSAL shift and publish, ba.cpp:1096-1100 (stage 6: position sN, values vN, wN or the stale sddotArr[6])
                     const double wE = stageFailed ? w6 : wN;
                     s0v = sN; v0 = vN; w0 = wE; w6 = wE;
                     sPrev = sCurPt; sdPrev = sdCurPt;
                     sCurPt = s0v; sdCurPt = v0;
                     const int idx = (DIR == 1) ? i : capI - 1 - i;
                     mypts[idx & 3] = make_double2(s0v, v0); // every lane of the group holds the same pair
                     st = (DIR == 1) ? 0 : 1;
                     const bool fin = sCur * DIR > sLast;      // ba.cpp:1109-1115
                     const bool late = !fin && (i > iMax);     // ba.cpp:1117-1122
                     nPts = fin ? i + 1 : nPts;
                     i = (fin || late) ? i : i + 1;
                     const bool full = !fin && !late && S8_CURVE_FULL(i);
                     endStatus = late ? (unsigned)BATOTP_ST_MAX_INTEG_TIME : (full ? (unsigned)BATOTP_ST_CAPACITY : endStatus);
                     phase = (fin || late || full) ? PH_DEAD : phase;
                     // a complete group of four points: one 64-byte store (not for the step that ends the path: its last
                     // point is still to be snapped onto the path end)
                     const bool chunk = !fin && ((DIR == 1) ? ((idx & 3) == 3) : ((idx & 3) == 0));
                     if (S8_ANY(chunk))
                     {
                        if (chunk)
                        {
                           const int base = idx & ~3;
                           const int lo = (DIR == 1) ? base : idx, hi = (DIR == 1) ? idx : ((base + 3 < capI) ? base + 3 : capI - 1);
                           const int at = base + j;
                           if (j < 4 && at >= lo && at <= hi) out[at] = mypts[j];
                           flushedTo = (DIR == 1) ? idx + 1 : idx;
                        }
                     }
                  }
               }
            }
            if (phase != PH_DEAD)
            {
               if (DIR == 1)
               {
                  // forward predictor (ba.cpp:1055-1065): only the move of the reverse-curve cursor survives
                  const bool pred = (st == 0);
                  if (S8_ANY(pred))
                  {
                     if (pred)
                     {
                        sCur = s0v + h * v0;
#include "sweep8_mvcwalk.inc"
                        st = 1;
                     }
                  }
               }
               // ---- tableau combination, ba.cpp:1073-1085 ---------------------------------------------
               const double *bc = rk[st];
               double sdotT = 0, sddotT = 0;
#if S8_LDS_STAGES
               {
                  const double2 q1 = myvw[1], q2 = myvw[2], q3 = myvw[3], q4 = myvw[4], q5 = myvw[5];
                  v1 = q1.x; w1 = q1.y; v2 = q2.x; w2 = q2.y; v3 = q3.x; w3 = q3.y; v4 = q4.x; w4 = q4.y; v5 = q5.x; w5 = q5.y;
               }
#endif
               // Stage st adds the terms k < st only.  The weights of the others are +0 in the table, and adding their products
               // changes nothing as long as every stage value is finite (x + (+-0) = x: a partial sum starts as 0 + b0 v0 and
               // is therefore never -0); a path that ever kept a non-finite stage value (0 * inf = NaN) takes the literal form.
               // Two rare situations share ONE wavefront-uniform guard (a ballot costs three vector instructions and a branch that
               // waits for it): a non-finite stage value (below), and a velocity-limit quotient outside the window of the shared
               // reciprocal (further down: `odd` is then a scalar condition)
               bool slowDiv = false;
#pragma unroll
               for (int q = 0; q < PER; ++q) slowDiv |= jOn[q] && fabs(thD[q]) > thrV && !(rOk[q] & s8_div_window(vmax[q]));
               const bool odd = S8_RARE(wild | slowDiv);
               if (S8_TAB_FAST && !odd)
               {
                  sdotT += bc[0] * v0; sddotT += bc[0] * w0;
                  sdotT += bc[1] * v1; sddotT += bc[1] * w1;
                  sdotT += bc[2] * v2; sddotT += bc[2] * w2;
                  sdotT += bc[3] * v3; sddotT += bc[3] * w3;
                  sdotT += bc[4] * v4; sddotT += bc[4] * w4;
                  sdotT += bc[5] * v5; sddotT += bc[5] * w5;
               }
               else
               {
                  sdotT += bc[0] * v0; sddotT += bc[0] * w0;
                  { const double a1 = sdotT + bc[1] * v1, b1 = sddotT + bc[1] * w1; sdotT = (st > 1) ? a1 : sdotT; sddotT = (st > 1) ? b1 : sddotT; }
                  { const double a2 = sdotT + bc[2] * v2, b2 = sddotT + bc[2] * w2; sdotT = (st > 2) ? a2 : sdotT; sddotT = (st > 2) ? b2 : sddotT; }
                  { const double a3 = sdotT + bc[3] * v3, b3 = sddotT + bc[3] * w3; sdotT = (st > 3) ? a3 : sdotT; sddotT = (st > 3) ? b3 : sddotT; }
                  { const double a4 = sdotT + bc[4] * v4, b4 = sddotT + bc[4] * w4; sdotT = (st > 4) ? a4 : sdotT; sddotT = (st > 4) ? b4 : sddotT; }
                  { const double a5 = sdotT + bc[5] * v5, b5 = sddotT + bc[5] * w5; sdotT = (st > 5) ? a5 : sdotT; sddotT = (st > 5) ? b5 : sddotT; }
               }
               sN = s0v + h * sdotT;
               double vN = v0 + h * sddotT;
               vN = dmax(vN, floorV); // ba.cpp:1085
               sCur = sN;

               // ---- sdotLim, ba.cpp:1204-1236 (theta' of the PREVIOUS evaluation point) -----------------
               if (DIR == 1)
               {
                  // evalsdot, ba.cpp:1590-1607
#include "sweep8_mvcwalk.inc"
#if S8_TAU_RCP
                  numM = sCur - mS0; denM = mS1 - mS0;
                  fastM = mOk & s8_num_ok(numM);
                  vPre = vN;
                  const double tauM = s8_div_by(numM, denM, rM);   // (a lane with !fastM is redone behind the guard of the segment change)
#else
                  const double tauM = (sCur - mS0) / (mS1 - mS0);
#endif
                  const double sdotMVC = dmax(mD0 + tauM * (mD1 - mD0), sdotMin);
                  vN = (vN > sdotMVC) ? sdotMVC : vN;
               }
               vN = dmin(vN, sdotCap);
               vN = dmax(vN, sdotMin);
               double lim1 = kInf;
               {
#pragma unroll
                  for (int q = 0; q < PER; ++q)
                  {
                     const bool on = jOn[q] && fabs(thD[q]) > thrV;
                     const bool fast = rOk[q] & s8_div_window(vmax[q]);
                     const double qv = fabs(s8_div_by(vmax[q], thD[q], rD[q]));
                     lim1 = (on & fast) ? dmin(lim1, qv) : lim1;
                  }
                  if (odd)
                  {
#pragma unroll
                     for (int q = 0; q < PER; ++q)
                     {
                        const bool on = jOn[q] && fabs(thD[q]) > thrV;
                        const bool fast = rOk[q] & s8_div_window(vmax[q]);
                        if (on && !fast) lim1 = dmin(lim1, fabs(vmax[q] / thD[q]));
                     }
                  }
                  lim1 = grp_min<G>(lim1);
                  vN = dmin(vN, lim1);
               }
               sdotCur = vN;
               // applyAccelConstraintsBisectionPt, ba.cpp:1250-1265
               lowFact = .01; sdotGood = 0; nGood = 0; sdotL = 0; sdotH = vN; sdotTry = vN; nIter = 0; stageFailed = false;

               // ---- evalSplinePartials, ba.cpp:1341-1413: updateCurSeg (ba.cpp:1617-1652) on the sites sres*k ----
               if (!S8_WALK_PRECHECK || S8_ANY(!((sCur >= sSeg) & (sCur <= sNext))))
               {
                  S8_CNT(10, 1);
                  for (;;)
                  {
                     sSeg = sres * (double)seg;
                     sNext = sres * (double)(seg + 1);
                     const bool inside = (sCur >= sSeg) & (sCur <= sNext);
                     const bool up = !inside & (sCur > sSeg), down = !inside & (sCur < sSeg);
                     status |= (!inside & !up & !down) ? (unsigned)BATOTP_ST_NONFINITE : 0u;
                     const bool mvUp = up & (seg < lastSeg), mvDn = down & (seg > 0);
                     seg = mvUp ? seg + 1 : (mvDn ? seg - 1 : seg);
                     if (!S8_ANY(mvUp | mvDn)) break;
                  }
#if S8_TAU_RCP
                  rSeg = s8_rcp_refined(sNext - sSeg); segOk = s8_div_window(sNext - sSeg) & (sNext - sSeg > 0.0);
#endif
               }
#if S8_TAU_RCP
               const double numT = sCur - sSeg, denT = sNext - sSeg;
               const bool fastT = segOk & s8_num_ok(numT);
               double tau = s8_div_by(numT, denT, rSeg);
               const bool slowM = (DIR == 1) && !fastM;
#else
               double tau = (sCur - sSeg) / (sNext - sSeg);
               const bool fastT = true, slowM = false;
#endif
               const bool chg = (seg != rowSeg);
               // one guard for the segment change and for the quotients that need their literal form
               if (S8_ANY(chg | !fastT | slowM))
               {
#if S8_TAU_RCP
                  if (!fastT) tau = numT / denT;
                  if (DIR == 1)
                  {
                     if (slowM)
                     {
                        // evalsdot and the rest of sdotLim once more with the literal quotient (ba.cpp:1590-1607, 1216-1229)
                        const double tauM = numM / denM;
                        const double sdotMVC = dmax(mD0 + tauM * (mD1 - mD0), sdotMin);
                        double vR = (vPre > sdotMVC) ? sdotMVC : vPre;
                        vR = dmin(vR, sdotCap);
                        vR = dmax(vR, sdotMin);
                        vR = dmin(vR, lim1);
                        sdotCur = vR; sdotH = vR; sdotTry = vR;
                     }
                  }
#endif
                  S8_CNT(9, 1);
                  if (chg)
                  {
#pragma unroll
                     for (int q = 0; q < PER; ++q)
                     {
                        if (FEAT < 0)
                        {
                           const unsigned at = (unsigned)(seg * nIn + jAt[q]);
                           double2 kl, kr; // knots seg and seg + 1 of this joint
                           bool literal = false; // the knots are not the prefetched ones, or a sixth lies outside div6's window
                           if (S8_PREFETCH)
                           {
                              // one segment further in the direction of the sweep: both knots are in registers
                              const bool hit = (DIR == 1) ? (preIdx == seg + 1) : (preIdx == seg);
                              kl = (DIR == 1) ? kEdge[q] : kPre[q];
                              kr = (DIR == 1) ? kPre[q] : kEdge[q];
                              literal = !hit;
                           }
                           else { kl = km[at]; kr = km[at + nIn]; }
                           // emit_segment's formulas (spline.cpp:203-209); x / 6 as div6 computes it inside its window, and ONE
                           // wavefront-uniform guard for everything that is rare here (a missed prefetch, a sixth outside the window)
                           double solL = kl.y, solR = kr.y, yL = kl.x, yR = kr.x;
                           double xa = solR - solL, xb = solR + 2 * solL;
                           literal |= !((fabs(xa) > 1e-280) & (fabs(xa) < 1e280) & (fabs(xb) > 1e-280) & (fabs(xb) < 1e280));
                           double c3, sixthB;
                           {
                              const double qa = xa * (1.0 / 6.0), qb = xb * (1.0 / 6.0);
                              c3 = __builtin_fma(__builtin_fma(-6.0, qa, xa), 1.0 / 6.0, qa);
                              sixthB = __builtin_fma(__builtin_fma(-6.0, qb, xb), 1.0 / 6.0, qb);
                           }
                           if (S8_RARE(literal))
                           {
                              if (S8_PREFETCH)
                              {
                                 const bool hit = (DIR == 1) ? (preIdx == seg + 1) : (preIdx == seg);
                                 const double2 dl = km[at], dr = km[at + nIn];
                                 kl = hit ? kl : dl;
                                 kr = hit ? kr : dr;
                              }
                              solL = kl.y; solR = kr.y; yL = kl.x; yR = kr.x;
                              c3 = div6(solR - solL);
                              sixthB = div6(solR + 2 * solL);
                           }
                           if (S8_PREFETCH)
                           {
                              kEdge[q] = (DIR == 1) ? kr : kl;
                              int nxt = (DIR == 1) ? seg + 2 : seg - 1;
                              nxt = nxt < 0 ? 0 : (nxt > lastSeg + 1 ? lastSeg + 1 : nxt);
                              kPre[q] = km[(unsigned)(nxt * nIn + jAt[q])];
                           }
                           const double c2 = solL / 2.0;
                           c1[q] = yR - yL - sixthB;
                           c2x2[q] = 2 * c2; c3x3[q] = 3 * c3; c3x6[q] = 6 * c3;
                        }
                        else
                        {
                           const Coef4 k = *reinterpret_cast<const Coef4 *>(coef + (unsigned)(seg * rowStride) + jAt[q] * 4);
                           c1[q] = k.c1; c2x2[q] = 2 * k.c2; c3x3[q] = 3 * k.c3; c3x6[q] = 6 * k.c3;
                        }
                     }
                     if (S8_PREFETCH && FEAT < 0)
                     {
                        const int nxt = (DIR == 1) ? seg + 2 : seg - 1;
                        preIdx = nxt < 0 ? (-(1 << 20)) : (nxt > lastSeg + 1 ? (-(1 << 20)) : nxt); // a clamped prefetch holds no usable knot
                     }
                     rowSeg = seg;
                  }
               }
               {
                  const double tau2 = tau * tau;
#pragma unroll
                  for (int q = 0; q < PER; ++q)
                  {
                     thD[q] = (c3x3[q] * tau2 + c2x2[q] * tau + c1[q]) * vfact;
                     thD2[q] = (c3x6[q] * tau + c2x2[q]) * afact;
                     rOk[q] = s8_div_window(thD[q]);
                     rD[q] = s8_rcp_refined(thD[q]);
                     sa[q] = (thD[q] < 0.0) ? -amax[q] : amax[q];
                  }
               }
               phase = PH_CHECK;
            }
         }
      }
      S8_TICK(tB);
      S8_CYC(5, tA, tB);
#ifdef S8_PROFILE
      {
         const unsigned long long mChk = __ballot(phase == PH_CHECK);
         S8_CNT(3, mChk != 0 ? 1 : 0); S8_CNT(4, __popcll(mChk) / G);
      }
#endif
      if (phase == PH_CHECK)
      {
         // (what the bisection update below needs and the constraint check does not produce is computed first: it fills the
         //  wait of the check's wavefront-uniform guard for its ballot)
         const double lowFact2 = lowFact * 2.0;
         const double sdotLShrunk = dmax(.999 * 0.0, (1.0 - lowFact2) * sdotTry);
         bool dec1, dec2;
         const double num1 = fabs(sdotTry - sdotGood), num2 = sdotTry - sdotLShrunk;
         bool close, tiny;
         s8_ratio_lt_pair(num1, num2, sdotTry, close, dec1, tiny, dec2);
         // ---- verifySecondOrderConstraints, ba.cpp:1514-1534, at sdotTry ---------------------------------
         const double sdotSQ = sdotTry * sdotTry;
         double H = sddotMax, L = -sddotMax;
         bool force = false;
         if (accOn)
         {
            bool rare = false; // a joint that stands still, or a quotient outside the window of the shared reciprocal
#pragma unroll
            for (int q = 0; q < PER; ++q)
            {
               const bool slow = fabs(thD[q]) < thrV;
               const double vTerm = thD2[q] * sdotSQ;
               const double nH = sa[q] - vTerm, nL = -sa[q] - vTerm; // (theta' = 0 lies outside the window: the literal form below)
               const bool fast = capOk & rOk[q] & s8_div_window(nH) & s8_div_window(nL);
               const double qH = s8_div_by(nH, thD[q], rD[q]);
               const double qL = s8_div_by(nL, thD[q], rD[q]);
               const bool use = jOn[q] & !slow & fast;
               // inside the window both quotients are finite and the bound is not a NaN (capOk): the one-instruction min / max
               // returns what the compare-and-select form returns (device_math.h: up to the sign of a zero that nothing reads)
               const double Hm = vmin_f64(H, qH), Lm = vmax_f64(L, qL);
               H = use ? Hm : H;
               L = use ? Lm : L;
               rare |= jOn[q] & (slow | !fast);
            }
            if (S8_RARE(rare))
            {
               S8_CNT(12, 1);
#pragma unroll
               for (int q = 0; q < PER; ++q)
               {
                  const bool slow = fabs(thD[q]) < thrV;
                  const int svpt = sgn(thD[q]);
                  const double vTerm = thD2[q] * sdotSQ;
                  const double nH = svpt * amax[q] - vTerm, nL = -svpt * amax[q] - vTerm;
                  const bool fast = capOk & rOk[q] & s8_div_window(sa[q] - vTerm) & s8_div_window(-sa[q] - vTerm);
                  if (jOn[q] && !slow && !fast)
                  {
                     H = dmin(H, nH / thD[q]);
                     L = dmax(L, nL / thD[q]);
                  }
                  // a joint that stands still (ba.cpp:1519-1524)
                  if (jOn[q] && slow && !(fabs(thD2[q]) < thrA)) force |= sdotSQ > amax[q] / fabs(thD2[q]);
               }
            }
         }
         double Hred = force ? -kInf : H;
         // (Reducing only the side the stage keeps and reading the other side's verdict off the ballot of "my bound is crossed"
         //  saves six instructions and measured 4 % SLOWER -- 716 against 690 ms: the ballot's way through the scalar registers
         //  sits on the critical path of the pass.)
         grp_min_max<G>(Hred, L);
         sddotH = Hred; sddotL = L;
         const bool isViol = L > Hred;

         // (The same pass as four exec-masked blocks by situation -- first check passed / violated without a feasible point / violated with
         //  one / feasible after a violation -- measured 5 % SLOWER than this one block of selects: the guards of four blocks cost more than
         //  the selects they save, profiles/r04_b_*; that variant is not kept in the source.)
         // ---- one pass of the loop of ba.cpp:1267-1321, as selects -----------------------------------------
         const bool first = (nIter == 0);
         const bool fin0 = !isViol && first; // the first check passes: the stage is done, nothing else happens
         bool fin = fin0, failed = false, toCert = false;
         // forward sweep: 99 % of the checks end here and the block is skipped; reverse sweep: three quarters of the time a
         // path is inside a bisection, the guard would nearly always be taken and only cost its ballot
         if (DIR == -1 || S8_ANY(!fin0))
         {
            const bool good = !isViol && !first;      // a feasible point after at least one violated one
            const bool shrink = isViol && nGood == 0; // ba.cpp:1281-1285: no feasible point known yet
            // the two threshold tests of the loop (computed above), behind ONE guard for the quotients they may need:
            //   ba.cpp:1294-1303: two successive feasible points closer than 1e-3 (relative)   -- matters for `good` paths
            //   ba.cpp:1313: the bracket [sdotL, sdotH] has collapsed                          -- matters while no feasible point is
            //   known, i.e. for `shrink` paths, whose bracket after this pass is [sdotLShrunk, sdotTry] whatever the other test says
            if (S8_RARE((good & !dec1) | (shrink & !dec2)))
            {
               close = dec1 ? close : (num1 / sdotTry < .001);
               tiny = dec2 ? tiny : (num2 / sdotTry < 1e-20);
            }
            const bool conv = good && (close || sdotTry < 0.0);
            fin = fin0 || conv;
            lowFact = shrink ? lowFact2 : lowFact;
            sdotH = isViol ? sdotTry : sdotH;
            sdotL = shrink ? sdotLShrunk : ((good && !conv) ? sdotTry : sdotL);
            sdotGood = good ? sdotTry : sdotGood;
            nGood += good ? 1 : 0;
            sdotCur = conv ? sdotTry : sdotCur;
            // ba.cpp:1305-1320
            const bool collapsed = shrink && tiny;
            failed = !fin && (nIter + 1 > 100 || sdotTry < 0.0 || collapsed);
            nIter += fin ? 0 : 1;
            // (the next candidate computed for both verdicts ahead of the reduction, then one select: 1.5 % SLOWER, 691 against 681 ms)
            sdotTry = (fin || failed) ? sdotTry : .5 * (sdotH + sdotL);
            status |= failed ? (unsigned)BATOTP_ST_BISECT_FAIL : 0u;
            nfail += failed ? 1 : 0;
            stageFailed = failed;
#if S8_FF
            // ---- CERTIFIED FAST-FORWARD of the bisection (s8_certify above) -------------------------------------------------------
            // The first check of the stage was violated and the loop goes on.  Forward sweep: the 8 paths of a wavefront run in
            // lockstep and 0.6 % of the stages bisect -- a dozen passes in which seven paths wait for one: the certificate runs
            // right here.  Reverse sweep: 22 % of the stages bisect and the passes are shared by several paths; run per arriving
            // path the block costs more than the passes it removes (profiles/r03_g_*, r04_j_*: +27 %), so there the path moves to
            // the phase PH_CERT and the block at the top of the loop serves the paths that have gathered in it (round 6).
            if (PER == 1 && accOn && (DIR == 1 || S8_FF_REV))
            {
               const bool ffWant = (((DIR == 1) ? (a.ff & 1) : (a.ff & 2)) != 0) && first && isViol && !failed;
               if (CERT) toCert = ffWant;
               else if (S8_ANY(ffWant))
               {
                  S8_CNT(11, 1);
                  if (ffWant)
                     s8_certify(jOn[0], thD[0], thD2[0], rD[0], rOk[0], amax[0], thrV, thrA, sddotMax, lowFact, sdotH, sdotL, sdotTry, sdotGood,
                                nGood, nIter);
               }
            }
#endif
         }
         wN = fin ? ((DIR == 1) ? sddotH : sddotL) : wN;
         phase = (fin || failed) ? PH_ENDED : (toCert ? PH_CERT : phase);
      }
      S8_TICK(tC);
      S8_CYC(6, tB, tC);
   }
#ifdef S8_PROFILE
   pc[7] = (double)(__builtin_readcyclecounter() - tLoop0);
   if (lane == 0 && a.prof)
      for (int k = 0; k < 16; ++k) a.prof[(int64_t)wave * 16 + k] = pc[k];
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
      if (DIR == 1) sdCurPt = mvc[nMvc - 1].y;
   }
   const double tElapsed = absh * (double)(nPts - 1); // ba.cpp:1112
   int64_t nOut = nPts;

   if (nPts >= 4)
   {
      // the snapped point joins the points still waiting in LDS; what has not reached HBM yet goes now
      const int idxLast = (DIR == 1) ? nPts - 1 : capI - nPts;
      mypts[idxLast & 3] = make_double2(sCurPt, sdCurPt);
      const int base = idxLast & ~3;
      const int at = base + j;
      const int lo = (DIR == 1) ? (flushedTo > base ? flushedTo : base) : idxLast;
      const int hi = (DIR == 1) ? idxLast : ((flushedTo - 1 < base + 3) ? flushedTo - 1 : base + 3);
      if (j < 4 && at >= lo && at <= hi) out[at] = mypts[j];
   }
   else
   {
      // ba.cpp:1171-1184: re-interpolate linearly in time to four points.  The points of so short a curve are all still
      // here: point 0 = the start, point nPts-2 = (sPrev, sdPrev), point nPts-1 = the snapped end.
      status |= BATOTP_ST_SHORT; nOut = 4;
      if (writer)
      {
         double ps[3], pd[3], tIn[3];
         const double sInit = (DIR == 1) ? 0.0 : sEnd;
         for (int k = 0; k < nPts; ++k)
         {
            const double2 q = (k == nPts - 1) ? make_double2(sCurPt, sdCurPt) : (k == 0 ? make_double2(sInit, vBoot) : make_double2(sPrev, sdPrev));
            ps[k] = q.x; pd[k] = q.y;
            tIn[k] = absh * (double)k;
         }
         if (DIR != 1)
         {
            for (int k = 0; k < nPts / 2; ++k)
            {
               swap_d(ps[k], ps[nPts - 1 - k]);
               swap_d(pd[k], pd[nPts - 1 - k]);
            }
         }
         const double tResNew = tIn[nPts - 1] / 3.;
         int cur = 0;
         for (int k = 0; k < 4; ++k)
         {
            const double tn = tResNew * (double)k;
            while (!(tn < tIn[cur + 1] || cur == nPts - 2)) ++cur;
            const double tauR = (tn - tIn[cur]) / (tIn[cur + 1] - tIn[cur]);
            out[DIR == 1 ? k : cap - 4 + k] = make_double2(ps[cur] + (ps[cur + 1] - ps[cur]) * tauR, pd[cur] + (pd[cur + 1] - pd[cur]) * tauR);
         }
      }
   }

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
#undef S8_CURVE_FULL

} // namespace bk
