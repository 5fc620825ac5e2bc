// ba_output.cpp -- resampling of the optimised s(t) to a constant-time output trajectory.
//
// Restates reference batotp/ba.cpp:1661-1931 (interpOutputData).  Host post-processing, the step
// AFTER the GPU hot path.
#include <cstdio>
#include <numeric>

#include "ba.h"
#include "util.h"

namespace BATOTP
{

namespace
{
void scaleInPlace(std::vector<double> &v, double c)
{
   for (size_t k = 0; k < v.size(); ++k) v[k] = c * v[k];
}
} // namespace

int BA::interpOutputData(Traj &traj)
{
   bool reinterp = false;
   const double outResUser = _outRes;
   if (_outRes < _integRes)
   {
      // never sample finer than the integration step; smooth + re-interpolate afterwards
      reinterp = true;
      _outRes = _integRes;
      _outSmoothFact *= std::max(outResUser / _outRes, 1.);
   }

   std::vector<double> sOut, tOut, unused1, unused2;
   Spline::splineSegs where;

   double tLast = traj.tMVC[(int)traj.tMVC.size() - 1];
   sOut = traj.sMVC;
   int nOut = (int)(_outSmoothFact * std::ceil(traj.tMVC[(int)traj.tMVC.size() - 1] / _outRes + 1.));
   nOut = std::max(nOut, 4);
   tOut.resize(nOut);

   // output times: uniform, except one extra site a third of a step from each end, which
   // suppresses acceleration spikes of the spline there (reference ba.cpp:1691-1699)
   std::iota(tOut.begin(), tOut.end(), -1);
   tOut[0] = 0;
   tOut[1] = 1.0 / 3.0;
   tOut[nOut - 1] = tOut[nOut - 2];
   tOut[nOut - 2] = tOut[nOut - 2] - 1.0 / 3.0;
   scaleInPlace(tOut, traj.tMVC[(int)traj.tMVC.size() - 1] / tOut[nOut - 1]);

   // s at the output times
   Spline::splineCoeffs sOfT;
   mySpline.findInterpSegs(traj.tMVC, tOut, where);
   mySpline.getSplineCoeffs(traj.sMVC, sOfT, "natural");
   mySpline.interp1spline(sOut, unused1, unused2, sOfT, where, traj.sres / _outSmoothFact);

   // path samples at those s
   mySpline.findInterpSegs(traj.sC, sOut, where);
   traj.nPts = nOut;
   traj.sres = _outRes;

   if (_pathType == JOINT || _pathType == BOTH)
   {
      for (unsigned int j = 0; j < _nJoints; ++j)
         mySpline.interp1spline(traj.theta[j], traj.thetaD[j], traj.thetaD2[j], traj.thetaC[j], where, traj.sres);
      if (_pathType == JOINT && _robotType != GENJNT) myRobot.call_fwdKin(traj.theta, traj.cart);
   }
   if (_pathType == CART || _pathType == BOTH)
   {
      for (unsigned int j = 0; j < _nCart; ++j)
         mySpline.interp1spline(traj.cart[j], traj.cartD[j], traj.cartD2[j], traj.cartC[j], where, traj.sres);
      if (_pathType == CART) myRobot.call_invKin(traj.theta, traj.cart);
   }

   if (_isTrqConOn)
   {
      // re-derive time derivatives on the output grid and recompute the torques
      // (reference ba.cpp:1744-1827): every site looks at the END of the previous segment
      where.seg.resize(traj.nPts);
      std::iota(where.seg.begin(), where.seg.end(), -1);
      where.tau.assign(traj.nPts, 1);
      where.seg[0] = 0;
      where.tau[0] = 0;
      const double tfact = traj.sres / _outSmoothFact;

      if (_isParallelMechOrig)
      {
         for (unsigned int j = 0; j < _nJoints; ++j)
         {
            mySpline.getSplineCoeffs(traj.theta[j], traj.thetaC[j], "natural");
            mySpline.interp1spline(traj.theta[j], traj.thetaD[j], traj.thetaD2[j], traj.thetaC[j], where, tfact);
         }
         for (unsigned int j = 0; j < _nCart; ++j)
         {
            mySpline.getSplineCoeffs(traj.cart[j], traj.cartC[j], "natural");
            mySpline.interp1spline(traj.cart[j], traj.cartD[j], traj.cartD2[j], traj.cartC[j], where, tfact);
         }
         traj.nPts = (int)traj.theta[0].size();
         myRobot.call_dynParallel(traj.a1, traj.a2, traj.a3, traj.a4, traj.cart, traj.cartD, traj.cartD2);

         std::vector<double> rhs(_nCart), sol(_nCart), cartpt(_nCart), thetapt(_nJoints);
         traj.trq.resize(_nJoints);
         for (unsigned int j = 0; j < _nJoints; ++j) traj.trq[j].resize(traj.nPts);
         for (unsigned int i = 0; i < traj.nPts; ++i)
         {
            for (unsigned int j = 0; j < _nCart; ++j) rhs[j] = traj.a2[j][i] + traj.a3[j][i] + traj.a4[j][i];
            for (unsigned int j = 0; j < _nCart; ++j) cartpt[j] = traj.cart[j][i];
            for (unsigned int j = 0; j < _nJoints; ++j) thetapt[j] = traj.theta[j][i];
            myRobot.call_setA(thetapt, cartpt, traj.Apt);
            solveLinSys(traj.Apt, rhs, sol, _isSVD);
            for (unsigned int j = 0; j < _nJoints; ++j) traj.trq[j][i] = sol[j];
         }
      }
      else
      {
         for (unsigned int j = 0; j < _nJoints; ++j)
         {
            mySpline.getSplineCoeffs(traj.theta[j], traj.thetaC[j], "clamped");
            mySpline.interp1spline(traj.theta[j], traj.thetaD[j], traj.thetaD2[j], traj.thetaC[j], where, tfact);
         }
         traj.nPts = (int)traj.theta[0].size();
         myRobot.call_dynSerial(traj.a1, traj.a2, traj.a3, traj.a4, traj.theta, traj.thetaD, traj.thetaD2);
         traj.trq.resize(_nJoints);
         for (unsigned int j = 0; j < _nJoints; ++j)
         {
            traj.trq[j].resize(traj.nPts);
            for (unsigned int i = 0; i < traj.nPts; ++i) traj.trq[j][i] = traj.a2[j][i] + traj.a3[j][i] + traj.a4[j][i];
         }
      }
   }

   if (traj.cart[0].size() != traj.theta[0].size())
   {
      for (int k = 0; k < 3; ++k) traj.cart[k].resize(traj.nPts);
   }

   if (_outSmoothFact > 1.5)
   {
      // moving average, then linear down-sampling by the smoothing factor (reference ba.cpp:1838-1871)
      const int nIn = traj.nPts;
      const int nDown = std::max((int)((nIn - 1) / _outSmoothFact) + 1, 4);
      std::vector<double> inSites(nIn), outSites(nDown);
      std::iota(inSites.begin(), inSites.end(), 0);
      std::iota(outSites.begin(), outSites.end(), 0);
      scaleInPlace(outSites, inSites[nIn - 1] / outSites[nDown - 1]);
      mySpline.findInterpSegs(inSites, outSites, where);

      const int window = (int)_outSmoothFact;
      traj.nPts = nDown;
      for (unsigned int j = 0; j < _nJoints; ++j)
      {
         smooth(traj.theta[j], window);
         mySpline.interp1linear(traj.theta[j], where);
      }
      if (_isTrqConOn)
      {
         for (unsigned int j = 0; j < _nJoints; ++j)
         {
            smooth(traj.trq[j], window);
            mySpline.interp1linear(traj.trq[j], where);
         }
      }
      for (unsigned int j = 0; j < _nCart; ++j)
      {
         smooth(traj.cart[j], window);
         mySpline.interp1linear(traj.cart[j], where);
      }
   }

   if (reinterp)
   {
      // back to the resolution the user asked for (reference ba.cpp:1873-1919)
      const int nUser = std::max((int)(std::ceil(tLast / outResUser)), 4);
      if (nUser == 2) tLast = outResUser;
      std::vector<double> g1(traj.nPts), g2(nUser);
      std::iota(g1.begin(), g1.end(), 0);
      std::iota(g2.begin(), g2.end(), 0);
      scaleInPlace(g1, 1. / g1[traj.nPts - 1]);
      scaleInPlace(g2, 1. / g2[nUser - 1]);
      mySpline.findInterpSegs(g1, g2, where);

      for (unsigned int j = 0; j < _nJoints; ++j)
      {
         mySpline.getSplineCoeffs(traj.theta[j], traj.thetaC[j], "natural");
         mySpline.interp1spline(traj.theta[j], traj.thetaD[j], traj.thetaD2[j], traj.thetaC[j], where, outResUser);
      }
      if (!_isGenericRobot)
      {
         for (unsigned int j = 0; j < _nCart; ++j)
         {
            mySpline.getSplineCoeffs(traj.cart[j], traj.cartC[j], "natural");
            mySpline.interp1spline(traj.cart[j], traj.cartD[j], traj.cartD2[j], traj.cartC[j], where, outResUser);
         }
      }
      if (_isTrqConOn)
      {
         for (unsigned int j = 0; j < _nJoints; ++j)
         {
            Spline::splineCoeffs tmpC;
            mySpline.getSplineCoeffs(traj.trq[j], tmpC, "natural");
            mySpline.interp1spline(traj.trq[j], unused1, unused2, tmpC, where, outResUser);
         }
      }
      traj.nPts = nUser;
      _outRes = outResUser;
   }
   traj.sres = _outRes;
   traj.nPts = (int)traj.theta[0].size();
   if (_nCart == 7)
   {
      q2aaVect(traj.cart);
      traj.cartC.resize(_nCart);
      traj.cartpt.resize(_nCart);
      traj.cartDpt.resize(_nCart);
      traj.cartD2pt.resize(_nCart);
   }
   return 0;
}

} // namespace BATOTP
