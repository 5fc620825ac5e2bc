// pointwise_va.hip.h -- K3 (max admissible sdot and the sddot interval at EVERY knot) for velocity / acceleration-only problems,
// written for the instruction count (round 5).  Same definition and the same bits as k_pointwise<FEAT <= 0> (kernels.hip.h):
// cursor on segment min(i, N-2); sdot starts from the clamp of BA::sdotLim (ba.cpp:1216), is cut by the joint-velocity limits
// (ba.cpp:1219-1223) with THIS knot's theta', then by applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) with
// verifySecondOrderConstraints' joint-acceleration family (ba.cpp:1514-1534); a knot without admissible sdot publishes NaN
// bounds.  One lane per knot, all joints of the knot in the lane (adjacent lanes = adjacent knots).
//
// Why a second kernel.  The general one runs the reference's loop literally: every iteration two IEEE divisions per joint
// (~30 instructions each way), and the 64 knots of a wavefront wait for the knot with the most iterations -- half of the knots
// of the bench paths bisect, 5-15 iterations each, so nearly every wavefront runs ~15 of them: 343 ms of the headline step,
// 8 % of the HBM peak in the one region the survey expected at the roof.  Here
//   * the quotients by theta' go through the shared refined reciprocal (device_math.h: sdiv_rcp / sdiv_by, the same bits as `/`
//     inside its window, the literal division behind a wavefront-uniform guard outside): one reciprocal per joint and knot
//     serves the velocity limit and both bounds of every check;
//   * a pass of the loop is one block of selects (the form of k_sweep8, sweep8.hip.h);
//   * the CERTIFIED FAST-FORWARD of sweep1.hip.h: every bound of the check is a line in x = sdot^2, the intervals stop
//     intersecting at a closed-form x*, and the iterations of the reference's loop whose outcome is certain given the check's
//     rounding-error bound are taken with the loop's own update statements but without their checks.  In K3 all lanes meet
//     their first check in the same instruction, so the whole wavefront runs the certificate ONCE, then a few cheap replay
//     iterations, then one or two real checks -- where the sweeps' 8 paths per wavefront arrive one by one.
// The derivation and error analysis of the certificate are in sweep1.hip.h (accelPt); tests/test_fast_forward_certificate.py
// checks the band on the CPU.  batotp_hip_set_fast_forward(ctx, 0) runs every iteration's real check.
#pragma once
#include "kernels.hip.h"

namespace bk
{

constexpr int K3V_BLOCK = 128;
#define K3V_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0)
#define K3V_RARE(x) __builtin_expect(K3V_ANY(x), 0)

// FEAT: -1 compact splines ((value, second derivative) pairs, Cin per knot), 0 coefficient rows (C x 4 doubles per knot)
template <int FEAT>
__global__ void __launch_bounds__(K3V_BLOCK) k_pointwise_va(DevProblem P, const PathInfo *__restrict__ pinfo, int B, const DevProblem *__restrict__ dP,
                                                           const double *__restrict__ sC, const double *__restrict__ coef, const double *__restrict__ km,
                                                           double *__restrict__ mvc, int64_t total, int64_t mvcSlot, int ff)
{
   static_assert(FEAT == -1 || FEAT == 0, "velocity / acceleration-only problems");
   __shared__ double lim[6][8];
   stage_limits(dP, lim);
   const int64_t g = (int64_t)blockIdx.x * K3V_BLOCK + threadIdx.x;
   if (g >= total) return;
   int lo = 0, hi = B - 1;
   while (lo < hi)
   {
      const int mid = (lo + hi + 1) >> 1;
      if (pinfo[mid].koff <= g) lo = mid; else hi = mid - 1;
   }
   const PathInfo pi = pinfo[lo];
   const int N = (int)pi.n, i = (int)(g - pi.koff);
   const int nJ = P.nJ;
   const bool accOn = (P.flags & BATOTP_F_JNT_ACC_ON) != 0;
   const double vfact = pi.vfact, afact = pi.afact;
   const double thrV = P.jnt_thresh * vfact, thrA = P.jnt_thresh * afact;

   // ---- the cursor at knot i (updateCurSeg only computes tau: 0 at a knot, 1 at the last knot) -------------------------------
   const int seg = (i < N - 1) ? i : N - 2;
   // (uniform sites are computed as k_sites computes them: compact batches keep no site array)
   double sCur = pi.sres_c * (double)i, sSeg = pi.sres_c * (double)seg, sNext = pi.sres_c * (double)(seg + 1);
   double sLastKnot = pi.sres_c * (double)(N - 1);
   if (!pi.uniform)
   {
      const double *__restrict__ s = sC + pi.koff;
      sCur = s[i]; sSeg = s[seg]; sNext = s[seg + 1]; sLastKnot = s[N - 1];
   }
   const double tau = (sCur - sSeg) / (sNext - sSeg);
   const double sdotCap = sLastKnot / pi.integ_res;                        // ba.cpp:1216
   const double sddotMax = 2 * sLastKnot / (pi.integ_res * pi.integ_res);  // ba.cpp:1257
   const bool capOk = (sddotMax == sddotMax);

   // ---- evalSplinePartials (ba.cpp:1341-1366): theta', theta'' of every joint at the knot ------------------------------------
   double thD[8], thD2[8], rD[8];
   bool rOk[8];
   {
      const double tau2 = tau * tau;
#pragma unroll
      for (int q = 0; q < 8; ++q)
      {
         thD[q] = 0; thD2[q] = 0; rD[q] = 0; rOk[q] = false;
         if (q < nJ)
         {
            double c1, c2, c3;
            if (FEAT < 0)
            {
               const double2 *__restrict__ kp = reinterpret_cast<const double2 *>(km) + (pi.koff + seg) * P.Cin + q;
               const double2 kl = kp[0], kr = kp[P.Cin];              // knots seg and seg + 1 of this joint
               c3 = div6(kr.y - kl.y);                                 // spline.cpp:203-209
               c2 = kl.y / 2.0;
               c1 = kr.x - kl.x - div6(kr.y + 2 * kl.y);
            }
            else
            {
               const Coef4 k = *reinterpret_cast<const Coef4 *>(coef + ((pi.koff + seg) * P.C + q) * 4);
               c3 = k.c3; c2 = k.c2; c1 = k.c1;
            }
            thD[q] = (3 * c3 * tau2 + 2 * c2 * tau + c1) * vfact;     // ba.cpp:1359
            thD2[q] = (6 * c3 * tau + 2 * c2) * afact;                 // ba.cpp:1360
            rOk[q] = sdiv_window(thD[q]);
            rD[q] = sdiv_rcp(thD[q]);
         }
      }
   }

   // ---- sdotLim, reverse direction (ba.cpp:1204-1236): no backward curve, _sdotMin = 0 ---------------------------------------
   double sdotCur = sdotCap;
   sdotCur = dmin(sdotCur, sdotCap);
   sdotCur = dmax(sdotCur, 0.0);
   {
      double l = kInf;
      bool odd = false;
#pragma unroll
      for (int q = 0; q < 8; ++q)
      {
         if (q < nJ)
         {
            const double vmax = lim[0][q];
            const bool on = fabs(thD[q]) > thrV;
            const bool fast = rOk[q] & sdiv_window(vmax);
            const double qv = fabs(sdiv_by(vmax, thD[q], rD[q]));
            l = (on & fast) ? dmin(l, qv) : l;
            odd |= on & !fast;
         }
      }
      if (K3V_RARE(odd))
      {
         // a quotient outside the window of the shared reciprocal: every joint in the literal form, in the reference's order
         l = kInf;
#pragma unroll
         for (int q = 0; q < 8; ++q)
            if (q < nJ && fabs(thD[q]) > thrV) l = dmin(l, fabs(lim[0][q] / thD[q]));
      }
      sdotCur = dmin(sdotCur, l);
   }

   // ---- verifySecondOrderConstraints at sdotTry (ba.cpp:1514-1534); leaves sddotH / sddotL, returns "violated" ---------------
   double sddotH = 0, sddotL = 0;
   auto check = [&](double sdotTry) __attribute__((always_inline)) -> bool {
      const double sdotSQ = sdotTry * sdotTry;
      double H = sddotMax, L = -sddotMax;
      bool force = false;
      if (accOn)
      {
         bool rare = false; // a joint that stands still, or a quotient outside the window of the shared reciprocal
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            if (q < nJ)
            {
               const double amax = lim[1][q];
               const bool slow = fabs(thD[q]) < thrV;
               const double sa = (thD[q] < 0.0) ? -amax : amax;   // sgn(theta') * amax where theta' != 0 (exact)
               const double vTerm = thD2[q] * sdotSQ;
               const double nH = sa - vTerm, nL = -sa - vTerm;
               const bool fast = capOk & rOk[q] & sdiv_window(nH) & sdiv_window(nL);
               const double qH = sdiv_by(nH, thD[q], rD[q]);
               const double qL = sdiv_by(nL, thD[q], rD[q]);
               const bool use = !slow & fast;
               // std::min / std::max of the reference as compare-and-select (the bounds are PUBLISHED here: not even the sign
               // of a zero may differ, so no v_min / v_max)
               H = (use & (qH < H)) ? qH : H;
               L = (use & (L < qL)) ? qL : L;
               rare |= slow | !fast;
            }
         }
         if (K3V_RARE(rare))
         {
            // the literal form for every joint of the lanes that need it, in the reference's order (min / max of the same set of
            // numbers: the order does not change the result, NaNs take the same way through dmin / dmax in both forms only when
            // the whole chain is literal -- so a lane with a rare joint redoes all its joints)
            if (rare)
            {
               H = sddotMax; L = -sddotMax;
#pragma unroll
               for (int q = 0; q < 8; ++q)
               {
                  if (q < nJ)
                  {
                     const double amax = lim[1][q];
                     const double vpt = thD[q];
                     if (fabs(vpt) < thrV)
                     {
                        if (!(fabs(thD2[q]) < thrA))
                           if (sdotSQ > amax / fabs(thD2[q])) force = true;    // ba.cpp:1519-1524
                     }
                     else
                     {
                        const int svpt = sgn(vpt);
                        const double vTerm = thD2[q] * sdotSQ;
                        H = dmin(H, (svpt * amax - vTerm) / vpt);
                        L = dmax(L, (-svpt * amax - vTerm) / vpt);
                     }
                  }
               }
            }
         }
      }
      sddotH = force ? -kInf : H;
      sddotL = L;
      return L > sddotH;
   };

   // ---- applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) -------------------------------------------------------------------
   double lowFact = .01, sdotGood = 0, sdotL = 0, sdotH = sdotCur, sdotTry = sdotCur;
   int nIter = 0, nGood = 0;
   bool done = false, failed = false;

   // one pass of the loop of ba.cpp:1267-1321 given the check's verdict on sdotTry, as selects (the block of k_sweep8)
   auto update = [&](bool isViol) __attribute__((always_inline)) {
      const bool first = (nIter == 0);
      const bool fin0 = !isViol && first;        // the first check passes: nothing else happens
      const bool good = !isViol && !first;       // a feasible point after at least one violated one
      const bool shrink = isViol && nGood == 0;  // ba.cpp:1281-1285: no feasible point known yet
      const double lowFact2 = lowFact * 2.0;
      const double sdotLShrunk = dmax(.999 * 0.0, (1.0 - lowFact2) * sdotTry);
      // the two threshold tests (ba.cpp:1294, 1313) decided as the correctly rounded quotient decides them; the quotient itself
      // only inside the band around the threshold (ratio_lt), behind one wavefront-uniform guard
      const double num1 = fabs(sdotTry - sdotGood), num2 = sdotTry - sdotLShrunk;
      constexpr double T1 = .001, T2 = 1e-20;
      const bool ok = (sdotTry > 1e-260) & (sdotTry < 1e260);
      bool close = num1 < sdotTry * (T1 * (1.0 - 1e-14));
      bool tiny = num2 < sdotTry * (T2 * (1.0 - 1e-14));
      const bool dec1 = ok & (close | (num1 > sdotTry * (T1 * (1.0 + 1e-14))));
      const bool dec2 = ok & (tiny | (num2 > sdotTry * (T2 * (1.0 + 1e-14))));
      if (K3V_RARE((good & !dec1) | (shrink & !dec2)))
      {
         close = dec1 ? close : (num1 / sdotTry < T1);
         tiny = dec2 ? tiny : (num2 / sdotTry < T2);
      }
      const bool conv = good && (close || sdotTry < 0.0);
      const bool fin = fin0 || conv;
      lowFact = shrink ? lowFact2 : lowFact;
      sdotH = isViol ? sdotTry : sdotH;
      sdotL = shrink ? sdotLShrunk : ((good && !conv) ? sdotTry : sdotL);
      sdotGood = good ? sdotTry : sdotGood;
      nGood += good ? 1 : 0;
      sdotCur = conv ? sdotTry : sdotCur;
      const bool collapsed = shrink && tiny;
      failed = !fin && (nIter + 1 > 100 || sdotTry < 0.0 || collapsed);   // ba.cpp:1305-1320
      nIter += fin ? 0 : 1;
      sdotTry = (fin || failed) ? sdotTry : .5 * (sdotH + sdotL);
      done = fin || failed;
   };

   update(check(sdotTry)); // the first check, all lanes at once

   if (K3V_ANY(!done))
   {
      // ---- CERTIFIED FAST-FORWARD (see the header; sweep1.hip.h has the derivation): lanes whose first check was violated --------
      if (ff && accOn && !done)
      {
         auto fastRcp = [](double d) {
            double r = __builtin_amdgcn_rcp(d);
            r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
            return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
         };
         const double xTop = sdotH * sdotH;                  // the first candidate: no later one is larger
         double au[8], mj[8];
         double uMin = kInf, eMax = 0.0, xForce = kInf, xs = kInf;
         bool allFinite = true;
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            au[q] = kInf; mj[q] = 0.0;
            if (q < nJ)
            {
               const double amax = lim[1][q];
               const bool use = !(fabs(thD[q]) < thrV);
               const double rv = fastRcp(use ? thD[q] : 1.0);
               const double a = amax * fabs(rv), m = thD2[q] * rv;
               const double e = a + fabs(m) * xTop;
               // a line with a non-finite coefficient must stop the fast-forward (min / max would drop its NaN)
               allFinite &= !use || ((a == a) & (m == m) & (e == e) & (fabs(a) < kInf) & (fabs(m) < kInf) & (e < kInf));
               au[q] = use ? a : kInf;
               mj[q] = use ? m : 0.0;
               uMin = dmin(uMin, au[q]);
               eMax = dmax(eMax, use ? e : 0.0);
               // a joint that stands still allows x <= amax / |theta''| (ba.cpp:1519-1524): the check's own quotient
               const bool standing = !use && !(fabs(thD2[q]) < thrA);
               xForce = standing ? dmin(xForce, amax / fabs(thD2[q])) : xForce;
               // this line against [-sddotMax, sddotMax]: as the upper line when m > 0, as the lower line when m < 0
               const double am = fabs(m);
               const double b0 = (a + sddotMax) * fastRcp(am > 0.0 ? am : 1.0);
               xs = (use && am > 0.0) ? dmin(xs, b0) : xs;
            }
         }
         // every pair of lines: the one with the larger slope as the upper line, the other as the lower line
#pragma unroll
         for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int k = q + 1; k < 8; ++k)
            {
               if (k < nJ)
               {
                  const double dm = fabs(mj[q] - mj[k]);
                  const double bnd = (au[q] + au[k]) * fastRcp(dm > 0.0 ? dm : 1.0);
                  xs = (dm > 0.0 && au[q] < kInf && au[k] < kInf) ? dmin(xs, bnd) : xs;
               }
            }
         const double sMin2 = dmin(uMin, sddotMax);          // .5 * (min(uMin, cap) - max(-uMin, -cap))
         const double xstar = dmin(xs, 4.0 * xTop);          // beyond 4 xTop: "never violated by the lines" just as well
         const double R = eMax * fastRcp(sMin2 > 0.0 ? sMin2 : 1.0);
         const double band = (R * 0x1p-40) * xstar;
         // a standing joint's threshold below the band around x* decides alone, and exactly; one above the band never matters;
         // one inside the band: no fast-forward
         const bool forceFirst = xForce < xstar - band;
         const double xThr = forceFirst ? xForce : xstar;
         const double bandThr = forceFirst ? -1.0 : band;
         const bool sane = allFinite & (eMax == eMax) & (xstar == xstar) & (R == R) & (R < 0x1p30) & (sMin2 > 1e-100) & (eMax < 1e100) &
                           (xTop > 1e-100) & (xTop < 1e100) & (xstar > 1e-100) & (forceFirst | (xForce > xstar + band));
         if (sane)
         {
            int it = nIter;
            bool inBand = false;
            // the search for a first feasible speed (ba.cpp:1281-1285): the bracket shrinks below every violated candidate
#pragma unroll 1
            for (; it < 90; ++it)
            {
               const double c = sdotTry, d = c * c - xThr;   // c * c: sdotSQ of the check
               inBand = !((fabs(d) > bandThr) & (c > 1e-100));
               if (inBand | !(d > 0.0)) break;
               lowFact *= 2.0;
               sdotH = c;
               sdotL = dmax(.999 * 0.0, (1.0 - lowFact) * c);
               sdotTry = .5 * (sdotH + sdotL);
            }
            if (!inBand && it < 90)
            {
               // sdotTry is feasible for certain and the first such speed: ba.cpp:1294 compares it with sdotGood = 0 and goes on.
               // From here the plain bisection (ba.cpp:1286-1303): sdotGood == sdotL throughout
               sdotGood = sdotTry; nGood = 1; sdotL = sdotTry;
               ++it;
               sdotTry = .5 * (sdotH + sdotL);
               // ba.cpp:1294 is false for certain when |c - sdotGood| > 1e-3 c (1 + 3e-14); otherwise this candidate is, or may
               // be, the last one and gets the real check and the real test
               const double convThr = 1e-3 * (1.0 + 3e-14);
#pragma unroll 1
               for (; it < 90; ++it)
               {
                  const double c = sdotTry, d = c * c - xThr;
                  const bool viol = d > 0.0;
                  inBand = !(fabs(d) > bandThr);
                  const bool goesOn = viol | (fabs(c - sdotL) > convThr * c);
                  if (inBand | !goesOn) break;
                  sdotH = viol ? c : sdotH;
                  sdotL = viol ? sdotL : c;
                  sdotTry = .5 * (sdotH + sdotL);
               }
               sdotGood = sdotL;
            }
            nIter = it;
         }
      }
      // ---- the rest of the loop with real checks ---------------------------------------------------------------------------
      while (K3V_ANY(!done))
      {
         if (!done) update(check(sdotTry));
      }
   }

   if (failed)
   {
      // no admissible sdot at this knot (ba.cpp:1307-1319): the K3 definition publishes NaN bounds
      sddotL = __longlong_as_double(0x7ff8000000000000LL);
      sddotH = sddotL;
   }
   // mvcSlot != 0 (BATOTP_F_MVC_IN_CURVES): the values go to the front of the path's curve slot (mvcSlot doubles per path)
   double *__restrict__ o = mvcSlot ? mvc + (int64_t)lo * mvcSlot : mvc + pi.koff * 3;
   o[i] = sdotCur;
   o[N + i] = sddotL;
   o[2 * (int64_t)N + i] = sddotH;
}

} // namespace bk
