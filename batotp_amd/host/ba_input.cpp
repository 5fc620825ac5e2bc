// ba_input.cpp -- host-side path resampling of BA (the step BEFORE the GPU hot path).
//
// Restates, from scratch, reference batotp/ba.cpp:95-316 (interpInputData), 327-403 (aa2qVect /
// q2aaVect), 412-638 (adjust_s), 651-781 (interpSpecial), 790-863 (evalSplineFullTraj, host copy
// used while resampling) and 2768-2794 (interpTrajLinear).  These routines are sequential walks
// over one path and decide the knot count N; they stay on the host.  The last
// step of interpInputData -- the N->N spline build and the dynamics model, reference
// ba.cpp:299-305 -- is NOT done here: BA::interpInputData() sends it to the HIP kernels
// (ba_device.cpp).
#include <cassert>
#include <cstdio>
#include <numeric>

#include "ba.h"
#include "util.h"

namespace BATOTP
{

namespace
{
// v[k] = c * k  (the reference builds these with std::iota followed by an in-place scale)
void rampTimes(std::vector<double> &v, double c)
{
   for (size_t k = 0; k < v.size(); ++k) v[k] = c * (double)k;
}
} // namespace

// ---------------------------------------------------------------------------------------------
// interpInputData, reference ba.cpp:95-316
// ---------------------------------------------------------------------------------------------
int BA::interpInputData(Traj &traj)
{
   const int rc = prepareKnots(traj);
   if (rc != 0) return rc; // also the designed -1 of interp-only mode (reference ba.cpp:139-159)

   // reference ba.cpp:299-305: final spline build (+ dynamics model) -> GPU
   if (deviceBuildKnotModel(traj) != 0) return -1;

   if (is_sdotOut)
   {
      traj.myMVChist.s.resize(4, std::vector<double>(0));
      traj.myMVChist.sdot.resize(4, std::vector<double>(0));
   }
   printf("Number of points on MVC, theta, and cart arrays after splineFact: %d\n", traj.nPts);
   return 0;
}

// reference ba.cpp:100-127: samples whose timestamp repeats the previous one are dropped, the input resolution
// follows from the timestamps.  Idempotent (the device path of optimizeBatch calls it before the resampler).
void BA::dropRepeatedTimestamps(Traj &traj)
{
   if (traj.timestamp.size() > 0)
   {
      // drop samples whose timestamp repeats the previous one (reference ba.cpp:100-127; the
      // index list is 8-bit there as well)
      std::vector<uint8_t> dup;
      dup.reserve(traj.nPts);
      for (unsigned int i = 1; i < traj.nPts; ++i)
      {
         if (traj.timestamp[i] == traj.timestamp[i - 1]) dup.push_back((uint8_t)i);
      }
      for (int k = (int)dup.size() - 1; k >= 0; --k)
      {
         traj.timestamp.erase(traj.timestamp.begin() + dup[k]);
         for (unsigned int j = 0; j < _nJoints; ++j) traj.theta[j].erase(traj.theta[j].begin() + dup[k]);
         for (unsigned int j = 0; j < _nCart; ++j) traj.cart[j].erase(traj.cart[j].begin() + dup[k]);
      }
      traj.nPts = (int)traj.timestamp.size();
      traj.tresInput = traj.timestamp.back() / (traj.nPts - 1);
      traj.sres = traj.tresInput;
      traj.sC = traj.timestamp;
   }
}

// everything of interpInputData up to and including "traj.sC.clear()" (reference ba.cpp:97-297)
int BA::prepareKnots(Traj &traj)
{
   dropRepeatedTimestamps(traj);

   if (traj.nPts == 1)
   {
      printf("Input trajectory has less than one site initially so no optimization will be performed.\n");
      return -1;
   }
   if (traj.nPts < 4) interpTrajLinear(traj, 4);

   if (_isInterpOnly)
   {
      traj.nPts = (int)traj.theta[0].size();
      const double oldRes = traj.sres;
      traj.ptsOrig.resize(traj.nPts);
      std::iota(traj.ptsOrig.begin(), traj.ptsOrig.end(), 0);
      if ((_pathType == CART || _pathType == BOTH) && _nCart == 6) aa2qVect(traj.cart);
      evalSplineFullTraj(traj, oldRes, _outRes);
      if (_nCart == 7) q2aaVect(traj.cart);
      traj.sres = _outRes;
      return -1;
   }

   traj.sLastSec = -1;

   if (_pathType == CART)
   {
      remClosePts(traj.cart, traj.theta, _cartThresh);
      traj.nPts = (int)traj.cart[0].size();
   }
   else
   {
      remClosePts(traj.theta, traj.cart, _jntThresh);
      traj.nPts = (int)traj.theta[0].size();
   }
   if (traj.nPts == 1)
   {
      printf("Input trajectory has less than one site after remClosePts() so no optimization will be performed.\n");
      return -1;
   }
   if (traj.nPts < 4) interpTrajLinear(traj, 4);

   if ((_pathType == CART || _pathType == BOTH) && _nCart == 6)
   {
      aa2qVect(traj.cart);
      traj.cartC.resize(_nCart);
      traj.cartpt.resize(_nCart);
      traj.cartDpt.resize(_nCart);
      traj.cartD2pt.resize(_nCart);
   }

   const bool jointPath = (_pathType == JOINT || _pathType == BOTH);
   const bool cartPath = (_pathType == CART || _pathType == BOTH);

   // decimate + smooth (reference ba.cpp:195-242; note both blocks pass _inputDecimFact to smooth)
   if (_inputDecimFact > 1)
   {
      if (jointPath)
      {
         for (unsigned int j = 0; j < _nJoints; ++j) smooth(traj.theta[j], _inputDecimFact);
         for (unsigned int j = 0; j < _nJoints; ++j) decimate(traj.theta[j], _inputDecimFact);
         traj.nPts = (int)traj.theta[0].size();
      }
      if (cartPath)
      {
         for (unsigned int j = 0; j < _nCart; ++j) smooth(traj.cart[j], _inputDecimFact);
         for (unsigned int j = 0; j < _nCart; ++j) decimate(traj.cart[j], _inputDecimFact);
         traj.nPts = (int)traj.cart[0].size();
      }
      traj.tresInput *= _inputDecimFact;
      traj.sres *= _inputDecimFact;
      _isInterpolated = true;
   }
   if (_smoothWindow > 1)
   {
      if (jointPath)
         for (unsigned int j = 0; j < _nJoints; ++j) smooth(traj.theta[j], _inputDecimFact);
      if (cartPath)
         for (unsigned int j = 0; j < _nCart; ++j) smooth(traj.cart[j], _inputDecimFact);
   }

   // kinematics (reference ba.cpp:244-280)
   if (_pathType == JOINT)
   {
      if (_isCartVelConOn || _isCartAccConOn)
      {
         if (myRobot.call_fwdKin(traj.theta, traj.cart) == -1) return -1;
      }
      else
      {
         traj.cart.resize(_nCart);
         for (size_t j = 0; j < _nCart; ++j) traj.cart[j].resize(traj.nPts);
      }
   }
   if (_pathType == CART)
   {
      if (_isJntVelConOn || _isJntAccConOn || _isTrqConOn)
      {
         myRobot.call_invKin(traj.theta, traj.cart);
      }
      else
      {
         traj.theta.resize(3);
         for (unsigned int j = 0; j < _nJoints; ++j) traj.theta[j].resize(traj.nPts);
      }
   }

   traj.ptsOrig.resize(traj.nPts);
   std::iota(traj.ptsOrig.begin(), traj.ptsOrig.end(), 0);

   // two resampling passes to a constant s resolution (reference ba.cpp:285-297)
   if (adjust_s(traj, "specialInterp") == -1) return -1;
   if (adjust_s(traj, "regularInterp") == -1) return -1;
   traj.sC.clear();
   return 0;
}

// ---------------------------------------------------------------------------------------------
// axis-angle <-> quaternion for a pose history, reference ba.cpp:327-403
// ---------------------------------------------------------------------------------------------
int BA::aa2qVect(std::vector<std::vector<double>> &pose)
{
   const size_t n = pose[0].size();
   _nCart = 7;
   pose.resize(_nCart);
   pose[6].resize(n);

   std::array<double, 3> aa = {{pose[3][0], pose[4][0], pose[5][0]}};
   std::array<double, 4> prev = aa2q(aa);
   for (size_t i = 0; i < n; ++i)
   {
      aa[0] = pose[3][i]; aa[1] = pose[4][i]; aa[2] = pose[5][i];
      std::array<double, 4> q = aa2q(aa);
      // keep successive quaternions on the same hemisphere
      double align = 0;
      for (int k = 0; k < 4; ++k) align += q[k] * prev[k];
      if (align < 0.0)
         for (int k = 0; k < 4; ++k) q[k] = -q[k];
      prev = q;
      for (int k = 0; k < 4; ++k) pose[3 + k][i] = q[k];
   }
   return 0;
}

int BA::q2aaVect(std::vector<std::vector<double>> &pose)
{
   const int n = (int)pose[0].size();
   for (int i = 0; i < n; ++i)
   {
      const std::array<double, 4> q = {{pose[3][i], pose[4][i], pose[5][i], pose[6][i]}};
      const std::array<double, 3> aa = q2aa(q);
      for (int k = 0; k < 3; ++k) pose[3 + k][i] = aa[k];
   }
   _nCart = 6;
   pose.resize(_nCart);
   return 0;
}

// ---------------------------------------------------------------------------------------------
// adjust_s, reference ba.cpp:412-638
// ---------------------------------------------------------------------------------------------
int BA::adjust_s(Traj &traj, std::string interpType)
{
   if (_sWeights[1] + _sWeights[2] < 1e-8) return 0; // s is time / node index: nothing to do

   const bool special = (interpType == "specialInterp");
   double cartNormRes = special ? _cartNormRes : _cartNormRes2;
   const double thetaNormRes = special ? _thetaNormRes : _thetaNormRes2;

   const int nPts = traj.nPts;
   std::vector<double> thetaArc(nPts), cartArc(nPts);
   traj.sC.resize(nPts);
   traj.ptsOrig.resize(nPts);

   const double sResi = traj.sres;
   assert(_quadraticRadThresh > 0.0);
   double minCartPerTheta = 1.0 / _quadraticRadThresh;
   double thetaWindow = 5; // degrees
   if (!_areJointAnglesDegrees) thetaWindow *= _DEG2RAD;
   double thetaArcMark = 0, cartArcMark = 0;

   // cumulative joint-space and Cartesian arc lengths
   for (int i = 0; i < nPts - 1; ++i)
   {
      double sq = 0;
      for (unsigned j = 0; j < _nJoints; ++j)
      {
         const double dlt = traj.theta[j][i + 1] - traj.theta[j][i];
         sq += dlt * dlt;
      }
      thetaArc[i + 1] = thetaArc[i] + std::sqrt(sq);

      sq = 0;
      for (int j = 0; j < 3; ++j)
      {
         const double dlt = traj.cart[j][i + 1] - traj.cart[j][i];
         sq += dlt * dlt;
      }
      cartArc[i + 1] = cartArc[i] + std::sqrt(sq);

      if (_isAutoIntegRes)
      {
         const double dTheta = thetaArc[i + 1] - thetaArcMark;
         const double dCart = cartArc[i + 1] - cartArcMark;
         if (dTheta > thetaWindow)
         {
            minCartPerTheta = std::min(minCartPerTheta, 3.0 * dCart / dTheta);
            thetaArcMark = thetaArc[i + 1];
            cartArcMark = cartArc[i + 1];
         }
      }
   }

   if (thetaArc[nPts - 1] < thetaNormRes)
   {
      printf("Input trajectory points are all identical no optimization will be performed.\n");
      return -1;
   }

   double sLast = 0, sResNew = 0;
   if (_isAutoIntegRes)
   {
      // reference ba.cpp:493-556: derive _integRes and the s weights from the path itself
      if ((cartArc[nPts - 1] < cartNormRes) && _scaleType == 2)
      {
         _sWeights[1] = _sWeights[1] + _sWeights[2];
         _sWeights[2] = 0;
         _scaleType = 1;
      }
      const double weightIn = _sWeights[1] + _sWeights[2];
      double cartRat = 500.0 * cartArc[nPts - 1];
      double thetaRat = thetaArc[nPts - 1];
      if (!_areJointAnglesDegrees) thetaRat *= _RAD2DEG;

      const double minIntegRes = 0.004, maxIntegRes = 0.2, KintegRes = 0.0003;
      double newIntegRes = KintegRes * _CartAccMax / _CartVelMax;
      for (unsigned int j = 0; j < _nJoints; ++j)
         newIntegRes = std::max(newIntegRes, KintegRes * _JntAccMax[j] / _JntVelMax[j]);
      newIntegRes = std::min(newIntegRes, maxIntegRes);

      const double changeRat = cartRat / thetaRat;
      double jointIntegRes = maxIntegRes * changeRat * changeRat;
      double jointIntegResWindow = maxIntegRes * minCartPerTheta * minCartPerTheta;
      jointIntegResWindow = std::max(jointIntegResWindow, 0.016);
      jointIntegRes = std::min(jointIntegRes, jointIntegResWindow);
      if (jointIntegRes < newIntegRes)
      {
         printf("InterpInputData(): Path-resolution integration resolution was changed from %f\n", newIntegRes);
         printf("                   to %f because orientation movement is dominant.\n", jointIntegRes);
         newIntegRes = jointIntegRes;
      }
      newIntegRes = std::max(newIntegRes, minIntegRes);
      printf("InterpInputData(): Final integ. res is %0.6f s.\n", newIntegRes);
      _integRes = newIntegRes;

      const double weightOut = cartRat + thetaRat;
      const double rescale = weightIn / weightOut;
      cartRat *= rescale;
      thetaRat *= rescale;
      if (thetaRat > _sWeights[1])
      {
         _sWeights[1] = thetaRat;
         _sWeights[2] = cartRat;
      }
      if (_sWeights[2] > 0) cartNormRes = std::min(cartNormRes, cartNormRes * _sWeights[2] / _sWeights[1]);
   }

   switch (_scaleType)
   {
   case 0: // s follows the teach time
      sLast = sResi * traj.ptsOrig[nPts - 1];
      sResNew = sResi;
      break;
   case 1: // s follows joint-space arc length
      sLast = thetaArc[nPts - 1];
      sResNew = thetaNormRes;
      break;
   case 2: // s follows Cartesian arc length
      sLast = cartArc[nPts - 1];
      sResNew = cartNormRes;
      break;
   }
   printf("_sWeights: %f %f %f; ", _sWeights[0], _sWeights[1], _sWeights[2]);

   double cartFact = 0;
   if (cartArc[nPts - 1] >= cartNormRes) cartFact = _sWeights[2] * sLast / cartArc[nPts - 1];
   const double teachFact = _sWeights[0] * sLast / (sResi * traj.ptsOrig[nPts - 1]);
   const double thetaFact = _sWeights[1] * sLast / thetaArc[nPts - 1];

   traj.sres = sLast / (nPts - 1);
   for (int i = 0; i < nPts; ++i)
   {
      traj.sC[i] = teachFact * sResi * traj.ptsOrig[i] + thetaFact * thetaArc[i] + cartFact * cartArc[i];
   }

   if (special)
   {
      InterpVars iv;
      iv.tTeachFact = teachFact;
      iv.thetaNormFact = thetaFact;
      iv.cartPosNormFact = cartFact;
      iv.sLast = sLast;
      iv.sResNew = sResNew;
      iv.sResi = sResi;
      interpSpecial(traj, iv);
   }
   else
   {
      for (int i = 1; i < nPts; ++i)
      {
         if (traj.sC[i] - traj.sC[i - 1] < 1e-12 * traj.sres)
         {
            printf("adjust_s(): s-resolution is too small between two points. aborting... \n");
            return -1;
         }
      }
      evalSplineFullTraj(traj, traj.sres, sResNew);
   }

   if (_pathType == JOINT)
   {
      if (_robotType == GENJNT)
      {
         traj.cart.resize(_nCart);
         for (size_t j = 0; j < _nCart; ++j) traj.cart[j].resize(traj.nPts);
      }
      else
      {
         myRobot.call_fwdKin(traj.theta, traj.cart);
      }
   }
   if (_pathType == CART) myRobot.call_invKin(traj.theta, traj.cart);

   printf("adjust_s() %s: number of trajpts: before %d; after %d\n", interpType.c_str(), nPts, traj.nPts);
   return 0;
}

// Host spline evaluation at traj.sCur used by interpSpecial(): the position part of reference
// ba.cpp:1341-1381 (segment walk of ba.cpp:1617-1652 + cubic evaluation).  As in the reference,
// the Cartesian channels are only evaluated when a Cartesian constraint is switched on.
int BA::evalSplinePoint(Traj &traj)
{
   const std::vector<double> &s = traj.sC;
   const int lastSeg = (int)s.size() - 2;
   int seg = traj.curSegC;
   double segStart;
   for (;;)
   {
      segStart = s[seg];
      if (traj.sCur >= segStart && traj.sCur <= s[seg + 1]) break;
      bool moved = false;
      if (traj.sCur > segStart)
      {
         if (seg >= lastSeg) { seg = lastSeg; break; }
         ++seg;
         moved = true;
      }
      if (traj.sCur < segStart)
      {
         if (seg <= 0) { seg = 0; break; }
         --seg;
         moved = true;
      }
      if (!moved) break; // NaN guard (the reference would not terminate)
   }
   traj.curSegC = seg;
   traj.tauC = (traj.sCur - segStart) / (s[seg + 1] - segStart);

   const double tau = traj.tauC, tau2 = tau * tau, tau3 = tau2 * tau;
   for (size_t j = 0; j < _nJoints; ++j)
   {
      const Spline::splineCoeffs &C = traj.thetaC[j];
      traj.thetapt[j] = C.c3[seg] * tau3 + C.c2[seg] * tau2 + C.c1[seg] * tau + C.c0[seg];
   }
   if (_isCartVelConOn || _isCartAccConOn)
   {
      for (size_t j = 0; j < _nCart; ++j)
      {
         const Spline::splineCoeffs &C = traj.cartC[j];
         traj.cartpt[j] = C.c3[seg] * tau3 + C.c2[seg] * tau2 + C.c1[seg] * tau + C.c0[seg];
      }
   }
   return 0;
}

// ---------------------------------------------------------------------------------------------
// interpSpecial, reference ba.cpp:651-781
// ---------------------------------------------------------------------------------------------
int BA::interpSpecial(Traj &traj, const InterpVars &iv)
{
   traj.thetaC.resize(_nJoints);
   for (unsigned int j = 0; j < _nJoints; ++j) mySpline.getSplineCoeffs(traj.theta[j], traj.thetaC[j], "natural");
   traj.cartC.resize(_nCart);
   for (unsigned int j = 0; j < _nCart; ++j) mySpline.getSplineCoeffs(traj.cart[j], traj.cartC[j], "natural");
   traj.curSegC = 0;
   traj.tauC = 0;

   int chunk = (int)std::ceil(iv.sLast / iv.sResNew) + 1;
   chunk = std::max(chunk, 4);

   std::vector<double> sNew(chunk);
   std::vector<std::vector<double>> thetaNew(_nJoints, std::vector<double>(chunk));
   std::vector<std::vector<double>> cartNew(_nCart, std::vector<double>(chunk));
   for (unsigned int j = 0; j < _nJoints; ++j) thetaNew[j][0] = traj.theta[j][0];
   for (unsigned int j = 0; j < _nCart; ++j) cartNew[j][0] = traj.cart[j][0];

   double sPrev = 0, dsCarry = 0;
   unsigned int newPt = 1, oldPt = 1;
   bool done = false;

   while (!done)
   {
      // distance from the last emitted point to the next original point
      double thSq = 0;
      for (unsigned int j = 0; j < _nJoints; ++j)
      {
         const double dlt = traj.theta[j][oldPt] - thetaNew[j][newPt - 1];
         thSq += dlt * dlt;
      }
      double caSq = 0;
      for (unsigned int j = 0; j < 3; ++j)
      {
         const double dlt = traj.cart[j][oldPt] - cartNew[j][newPt - 1];
         caSq += dlt * dlt;
      }
      const double ds = iv.tTeachFact * iv.sResi * traj.ptsOrig[oldPt] + iv.thetaNormFact * std::sqrt(thSq) +
                        iv.cartPosNormFact * std::sqrt(caSq);

      if (ds > iv.sResNew)
      {
         sNew[newPt] = sPrev + iv.sResNew - dsCarry;
         dsCarry = 0;
         sPrev = sNew[newPt];
         traj.sCur = sPrev;
         if (traj.sCur > traj.sC[traj.nPts - 1]) done = true;
         if (!done)
         {
            evalSplinePoint(traj);
            for (unsigned int j = 0; j < _nJoints; ++j) thetaNew[j][newPt] = traj.thetapt[j];
            for (unsigned int j = 0; j < _nCart; ++j) cartNew[j][newPt] = traj.cartpt[j];
            oldPt = traj.curSegC + 1;
            ++newPt;
            if (newPt == thetaNew[0].size())
            {
               sNew.resize(newPt + chunk);
               for (unsigned int j = 0; j < _nJoints; ++j) thetaNew[j].resize(newPt + chunk);
               for (unsigned int j = 0; j < _nCart; ++j) cartNew[j].resize(newPt + chunk);
            }
         }
      }
      else if (oldPt == traj.nPts - 1)
      {
         done = true;
      }
      else
      {
         dsCarry = ds;
         sPrev = traj.sC[oldPt];
         ++oldPt;
      }
   }
   // the original end point closes the new path
   for (unsigned int j = 0; j < _nJoints; ++j)
   {
      thetaNew[j][newPt] = traj.theta[j][traj.nPts - 1];
      thetaNew[j].resize(newPt + 1);
   }
   for (unsigned int j = 0; j < _nCart; ++j)
   {
      cartNew[j][newPt] = traj.cart[j][traj.nPts - 1];
      cartNew[j].resize(newPt + 1);
   }
   traj.nPts = newPt + 1;
   traj.sres = iv.sResNew;
   traj.theta = thetaNew;
   traj.cart = cartNew;

   if (traj.nPts < 4) interpTrajLinear(traj, 4);

   traj.ptsOrig.resize(traj.nPts);
   std::iota(traj.ptsOrig.begin(), traj.ptsOrig.end(), 0);
   return 0;
}

// ---------------------------------------------------------------------------------------------
// evalSplineFullTraj (host copy for resampling nPtsOld -> nPtsNew), reference ba.cpp:790-863
// ---------------------------------------------------------------------------------------------
int BA::evalSplineFullTraj(Traj &traj, const double oldRes, double newRes)
{
   const int nOld = traj.nPts;
   traj.nPtsC = nOld;
   int nNew = (int)std::ceil(oldRes / newRes * (nOld - 1)) + 1;
   nNew = std::max(nNew, 4);
   newRes = oldRes * (nOld - 1) / (nNew - 1);

   if ((int)traj.sC.size() != nOld)
   {
      traj.sC.resize(nOld);
      rampTimes(traj.sC, traj.sres);
   }
   traj.sMVC.resize(nNew);
   const double sScale = traj.sC[nOld - 1] / (double)(nNew - 1);
   rampTimes(traj.sMVC, sScale);

   traj.sresC = traj.sres;
   traj.vFact = 1 / traj.sresC;
   traj.aFact = traj.vFact * traj.vFact;
   traj.sres = newRes;
   traj.nPts = nNew;

   traj.thetaC.resize(_nJoints);
   traj.cartC.resize(_nCart);
   for (unsigned int j = 0; j < _nJoints; ++j) mySpline.getSplineCoeffs(traj.theta[j], traj.thetaC[j], "natural");
   for (unsigned j = 0; j < _nCart; ++j) mySpline.getSplineCoeffs(traj.cart[j], traj.cartC[j], "natural");
   mySpline.getSplineCoeffs(traj.ptsOrig, traj.ptsOrigC, "natural");

   Spline::splineSegs where;
   if (mySpline.findInterpSegs(traj.sC, traj.sMVC, where) == -1) return -1;

   traj.thetaD.resize(_nJoints);
   traj.thetaD2.resize(_nJoints);
   for (unsigned j = 0; j < _nJoints; ++j)
      mySpline.interp1spline(traj.theta[j], traj.thetaD[j], traj.thetaD2[j], traj.thetaC[j], where, oldRes);
   traj.cartD.resize(_nCart);
   traj.cartD2.resize(_nCart);
   for (unsigned j = 0; j < _nCart; ++j)
      mySpline.interp1spline(traj.cart[j], traj.cartD[j], traj.cartD2[j], traj.cartC[j], where, oldRes);
   std::vector<double> d1, d2;
   mySpline.interp1spline(traj.ptsOrig, d1, d2, traj.ptsOrigC, where, oldRes);
   _isInterpolated = true;
   return 0;
}

// ---------------------------------------------------------------------------------------------
// interpTrajLinear, reference ba.cpp:2768-2794
// ---------------------------------------------------------------------------------------------
int BA::interpTrajLinear(Traj &traj, const int nPtsNew)
{
   const int nOld = traj.nPts;
   std::vector<double> gridOld(nOld), gridNew(nPtsNew);
   rampTimes(gridOld, 1.0 / (nOld - 1));
   rampTimes(gridNew, 1.0 / (nPtsNew - 1));

   Spline::splineSegs where;
   mySpline.findInterpSegs(gridOld, gridNew, where);
   for (unsigned int j = 0; j < _nJoints; ++j) mySpline.interp1linear(traj.theta[j], where);
   for (unsigned int j = 0; j < _nCart; ++j) mySpline.interp1linear(traj.cart[j], where);
   traj.sres = traj.sres * (nOld - 1) / (nPtsNew - 1);
   traj.nPts = nPtsNew;
   return 0;
}

} // namespace BATOTP
