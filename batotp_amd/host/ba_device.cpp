// ba_device.cpp -- the seam between the BA C++ API and the HIP kernels.
//
// Everything here talks to the GPU only through the C-ABI of include/batotp_hip.h
// (libbatotp_hip.so).  It covers the hot-path calls of the reference:
//   * the final N->N spline build and the dynamics model of interpInputData
//     (reference ba.cpp:299-305 -> evalSplineFullTraj ba.cpp:790-863, findDynModel ba.cpp:873-949),
//   * sweep() (reference ba.cpp:979-1195),
// plus the many-path extension optimizeBatch().  There is no host implementation to fall back to:
// if the device layer fails the call returns -1 after printing the reason.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>

#include "ba.h"
#include "batotp_hip.h"
#include "util.h"

namespace BATOTP
{

struct BA::Gpu
{
   batotp_ctx *ctx = nullptr;
   ~Gpu()
   {
      if (ctx) batotp_hip_ctx_destroy(ctx);
   }
};

namespace
{
// RAII for a device batch
struct BatchGuard
{
   batotp_batch *b = nullptr;
   ~BatchGuard()
   {
      if (b) batotp_hip_batch_destroy(b);
   }
};
struct OutputGuard
{
   batotp_output *o = nullptr;
   ~OutputGuard()
   {
      if (o) batotp_hip_output_destroy(o);
   }
};
struct ResampledGuard
{
   batotp_resampled *r = nullptr;
   ~ResampledGuard()
   {
      if (r) batotp_hip_resampled_destroy(r);
   }
};

int fail(const char *what, int rc)
{
   printf("batotp device layer: %s failed (code %d) %s\n", what, rc, batotp_hip_last_error());
   return -1;
}

void coeffsToVector(const Spline::splineCoeffs &C, size_t n, std::vector<double> &flat)
{
   flat.assign(4 * n, 0.0);
   const std::vector<double> *src[4] = {&C.c0, &C.c1, &C.c2, &C.c3};
   for (int k = 0; k < 4; ++k)
   {
      const size_t m = std::min(n, src[k]->size());
      std::copy(src[k]->begin(), src[k]->begin() + m, flat.begin() + k * n);
   }
}

void vectorToCoeffs(const std::vector<double> &flat, size_t n, Spline::splineCoeffs &C)
{
   C.c0.assign(flat.begin(), flat.begin() + n);
   C.c1.assign(flat.begin() + n, flat.begin() + 2 * n);
   C.c2.assign(flat.begin() + 2 * n, flat.begin() + 3 * n);
   C.c3.assign(flat.begin() + 3 * n, flat.begin() + 4 * n);
}
} // namespace

int BA::gpuAcquire()
{
   if (_gpu && _gpu->ctx) return 0;
   _gpu = std::make_shared<Gpu>();
   const int rc = batotp_hip_ctx_create(_deviceId, &_gpu->ctx);
   if (rc != BATOTP_OK)
   {
      _gpu.reset();
      printf("batotp: no usable HIP device (batotp_hip_ctx_create(%d) returned %d: %s).\n", _deviceId, rc,
             batotp_hip_last_error());
      printf("batotp: the sweep and the per-knot precompute run on the GPU only; there is no CPU fallback.\n");
      return -1;
   }
   return 0;
}

// BA configuration -> the POD the kernels read
void BA::fillProblem(void *out) const
{
   batotp_problem &P = *static_cast<batotp_problem *>(out);
   std::memset(&P, 0, sizeof(P));
   P.n_joints = (int32_t)_nJoints;
   P.n_cart = (int32_t)_nCart;
   P.robot_type = _robotType;
   uint32_t f = 0;
   if (_isJntAccConOn) f |= BATOTP_F_JNT_ACC_ON;
   if (_isTrqConOn) f |= BATOTP_F_TRQ_ON;
   if (_isCartVelConOn) f |= BATOTP_F_CART_VEL_ON;
   if (_isCartAccConOn) f |= BATOTP_F_CART_ACC_ON;
   if (_isParallelMechOrig) f |= BATOTP_F_PARALLEL;
   if (_isPar2Ser) f |= BATOTP_F_PAR2SER;
   if (_isSVD) f |= BATOTP_F_SVD;
   // cos/sin from the host libm: bit parity with the reference (RR), and with the host twin of the chain model
   // ... and with the forward kinematics of the device resampler / output stage (KUKA, RR)
   //     and with the pose conversions of a BOTH path (atan2 of the output stage)
   if (_robotType == RR || _robotType == KUKA || _pathType == BOTH || (_isTrqConOn && !_isParallelMechOrig)) f |= BATOTP_F_HOST_TRIG;
   P.flags = f;
   for (unsigned int j = 0; j < _nJoints && j < BATOTP_MAX_JOINTS; ++j)
   {
      if (j < _JntVelMax.size()) P.jnt_vel_max[j] = _JntVelMax[j];
      if (j < _JntAccMax.size()) P.jnt_acc_max[j] = _JntAccMax[j];
      if (j < _JntTrqMax.size()) P.jnt_trq_max[j] = _JntTrqMax[j];
      if (j < _JntTrqMin.size()) P.jnt_trq_min[j] = _JntTrqMin[j];
   }
   P.cart_vel_max = _CartVelMax;
   P.cart_acc_max = _CartAccMax;
   P.jnt_thresh = _jntThresh;
   P.quad_rad_thresh = _quadraticRadThresh;
   P.integ_res = _integRes;
   P.max_integ_time = _maxIntegTime;
   if (_robotType == CSPR3DOF)
   {
      const std::vector<std::vector<double>> &A = const_cast<Robot &>(myRobot).cableAnchors();
      for (int r = 0; r < 3; ++r)
         for (int c = 0; c < 3; ++c) P.pmat[r * 3 + c] = A[r][c];
   }
}

namespace
{
// cos and sin of the same angle through two separate libm calls: inlined side by side the compiler fuses them into one
// sincos(), whose results are not bit-identical to cos() / sin() in this glibc (SURVEY.md 8c observed the same on the
// reference's KUKA kinematics).  The trig tables of the chain model are DEFINED as libm cos() and libm sin().
double __attribute__((noinline)) libmCos(double x) { return std::cos(x); }
double __attribute__((noinline)) libmSin(double x) { return std::sin(x); }
} // namespace

void BA::setSerialModel(const void *model)
{
   if (model) myRobot.setSerialModel(*static_cast<const batotp_serial_model *>(model));
}

// Serial robots with a chain model (every serial robot but RR): hand the table to the batch once (pathIndex 0)
// and the host cosines / sines of the joint angles of one path
int BA::deviceSerialDynamicsInputs(void *batch, int pathIndex, const double *const *thetaRows, long long N)
{
   const batotp_serial_model *m = myRobot.serialModel();
   if (!m) return -1;
   batotp_batch *b = static_cast<batotp_batch *>(batch);
   int rc;
   if (pathIndex == 0)
   {
      rc = batotp_hip_set_serial_model(b, m);
      if (rc) return fail("set_serial_model", rc);
   }
   const double unit = m->degrees ? _DEG2RAD : 1.0;
   std::vector<double> trig(2 * (size_t)_nJoints * (size_t)N);
   for (unsigned int j = 0; j < _nJoints; ++j)
      for (long long i = 0; i < N; ++i)
      {
         const double q = unit * thetaRows[j][i];
         trig[(size_t)j * N + i] = libmCos(q);
         trig[(size_t)(_nJoints + j) * N + i] = libmSin(q);
      }
   rc = batotp_hip_upload_joint_trig(b, pathIndex, trig.data());
   if (rc) return fail("upload_joint_trig", rc);
   return 0;
}

// Which configurations batotp_hip_resample covers (what it does not, interpInputData refuses with a message).
int BA::exportResampleParams(const Traj &traj, void *out) const
{
   batotp_resample_params &R = *static_cast<batotp_resample_params *>(out);
   std::memset(&R, 0, sizeof(R));
   R.n_joints = (int32_t)_nJoints;
   R.n_cart = (int32_t)_nCart;
   R.robot_type = _robotType;
   R.path_type = (_pathType == JOINT) ? BATOTP_PATH_JOINT : (_pathType == CART ? BATOTP_PATH_CART : (_pathType == BOTH ? BATOTP_PATH_BOTH : 0));
   R.scale_type = _scaleType;
   uint32_t f = 0;
   if (_isCartVelConOn) f |= BATOTP_F_CART_VEL_ON;
   if (_isCartAccConOn) f |= BATOTP_F_CART_ACC_ON;
   R.flags = f;
   for (int k = 0; k < 3; ++k) R.s_weights[k] = _sWeights[k];
   R.theta_norm_res = _thetaNormRes; R.theta_norm_res2 = _thetaNormRes2;
   R.cart_norm_res = _cartNormRes; R.cart_norm_res2 = _cartNormRes2;
   R.jnt_thresh = _jntThresh; R.cart_thresh = _cartThresh;
   R.input_decim_fact = (int32_t)_inputDecimFact; R.smooth_window = (int32_t)_smoothWindow;
   if (_robotType == CSPR3DOF)
   {
      const std::vector<std::vector<double>> &A = const_cast<Robot &>(myRobot).cableAnchors();
      for (int r = 0; r < 3; ++r)
         for (int c = 0; c < 3; ++c) R.pmat[r * 3 + c] = A[r][c];
   }

   if (_isAutoIntegRes)
   {
      // the class default (reference ba.h:309): both adjust_s passes derive _integRes, _sWeights, _scaleType and the Cartesian
      // resolution from the path (ba.cpp:462-470, 493-556); these are the inputs of that rule
      R.flags |= BATOTP_RS_AUTO_INTEG_RES;
      for (unsigned int j = 0; j < _nJoints && j < BATOTP_MAX_JOINTS; ++j)
      {
         if (j < _JntVelMax.size()) R.jnt_vel_max[j] = _JntVelMax[j];
         if (j < _JntAccMax.size()) R.jnt_acc_max[j] = _JntAccMax[j];
      }
      R.cart_vel_max = _CartVelMax; R.cart_acc_max = _CartAccMax; R.quad_rad_thresh = _quadraticRadThresh;
      R.degrees = _areJointAnglesDegrees ? 1 : 0;
   }
   if (_isInterpOnly) return -1;
   if (traj.nPts < 4) return -1; // (timestamps: call dropRepeatedTimestamps first; they only set traj.sres)
   if (_sWeights[1] + _sWeights[2] < 1e-8) return -1;
   if (_nJoints > BATOTP_MAX_JOINTS || _nCart > BATOTP_MAX_CART || _nCart < 3) return -1;
   const bool joint = _pathType == JOINT && _robotType == GENJNT && !_isCartVelConOn && !_isCartAccConOn;
   const bool cable = _pathType == CART && _robotType == CSPR3DOF && _nJoints == 3 && _nCart == 3 &&
                      (_isJntVelConOn || _isJntAccConOn || _isTrqConOn);
   // JOINT paths of the robots with forward kinematics (reference robot.cpp:73-96): the tool point is recomputed after
   // each resampling pass; its cos / sin come from the host libm (bit parity with the host resampler and the reference)
   const bool kin = _pathType == JOINT && _nCart == 3 && ((_robotType == KUKA && _nJoints == 7) || (_robotType == RR && _nJoints == 2));
   // joints and Cartesian rows taught together (reference ba.cpp:184-192, 245-262: both channel sets are resampled as taught).  Six
   // Cartesian rows are tool poses (the UR5 example): orientations as axis-angle -> quaternions (BA::aa2qVect, ba.cpp:327-369; sine /
   // cosine of the half angle from the host libm); any other count (tool positions without orientations) goes through unchanged
   const bool both = _pathType == BOTH;
   if (kin || (both && _nCart == 6)) R.flags |= BATOTP_F_HOST_TRIG;
   return (joint || cable || kin || both) ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// reference ba.cpp:160-297 for ONE path behind the C-ABI: taught points in, uniform-s knots back in traj.theta / traj.cart
// ---------------------------------------------------------------------------------------------
int BA::deviceResampleOne(Traj &traj)
{
   batotp_resample_params rsp;
   if (exportResampleParams(traj, &rsp) != 0)
   {
      printf("interpInputData(): robotType=%s with pathType=%s is not covered by the device resampler.\n", _robotTypeStr.c_str(), _pathTypeStr.c_str());
      return -1;
   }
   if (gpuAcquire() != 0) return -1;
   const int64_t n = traj.nPts;
   const int rowsIn = (int)(_nJoints + _nCart);
   std::vector<double> x((size_t)n * rowsIn, 0.0);
   for (unsigned int j = 0; j < _nJoints && j < traj.theta.size(); ++j)
      if (traj.theta[j].size() >= (size_t)n) std::copy(traj.theta[j].begin(), traj.theta[j].begin() + n, x.begin() + (size_t)j * n);
   for (unsigned int j = 0; j < _nCart && j < traj.cart.size(); ++j)
      if (traj.cart[j].size() >= (size_t)n) std::copy(traj.cart[j].begin(), traj.cart[j].begin() + n, x.begin() + (size_t)(_nJoints + j) * n);
   const double sresIn = traj.sres;
   // The path is resampled TWICE and the two results must be identical to the bit before one of them is used.  Why: one of
   // ~11 000 one-path calls of round 4 (64 processes sharing a GPU) came back with different knots and status 0; 9 216 calls
   // under the same conditions since have not reproduced it, and the whole GPU suite passes with every workspace poisoned before
   // use (round 6: batotp_hip_set_poison, profiles/r06_f_*) -- no kernel reads memory nobody wrote.  Its cause is not known, and a
   // caller of interpInputData() must never be handed a wrong path silently.  Two evaluations that disagree are reported with the
   // place of the first difference and the path is evaluated again: the result is used once two CONSECUTIVE evaluations agree (four
   // evaluations at most).  A batch of one costs milliseconds; the many-path route (optimizeBatch) has the same protection through
   // checksums computed in HBM.
   std::vector<double> y, yAgain;
   int64_t nKnots = 0;
   double sres = 0;
   uint32_t status = 0;
   double integ = 0, sw[3] = {0, 0, 0};
   int32_t scale = 0;
   float ms = 0;
   const bool poses = rsp.path_type == BATOTP_PATH_BOTH && _nCart == 6;
   const int rowsOut = (int)(_nJoints + _nCart) + (poses ? 1 : 0); // aa2qVect leaves position + quaternion rows (reference ba.cpp:335)
   bool agreed = false;
   // every evaluation keeps a checksum of each of its intermediate stages (batotp_hip_set_resample_trace): two evaluations that
   // disagree then say in which stage -- which kernel's output -- they first differ
   static const char *const kStage[8] = {"taught points after close-point removal", "their sites (first adjust_s pass)", "their second derivatives",
                                         "points emitted by interpSpecial", "those points in the second pass", "their sites (second pass)",
                                         "their second derivatives", "knots"};
   uint64_t trace[8] = {0, 0, 0, 0, 0, 0, 0, 0}, traceAgain[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   bool haveTrace = false;
   std::vector<double> stageKept[2];       // stages 2 and 3 of the previous evaluation
   batotp_hip_set_resample_trace(_gpu->ctx, 1);
   struct TraceOff { batotp_ctx *c; ~TraceOff() { batotp_hip_set_resample_trace(c, 0); } } traceOff{_gpu->ctx};
   for (int pass = 0; pass < 4 && !agreed; ++pass)
   {
      ResampledGuard rs;
      int rc = batotp_hip_resample(_gpu->ctx, &rsp, 1, &n, x.data(), &sresIn, &rs.r);
      if (rc) return fail("resample", rc);
      const bool traced = batotp_hip_resampled_trace(rs.r, traceAgain) == BATOTP_OK;
      std::vector<double> stageAgain[2];   // stages 2 and 3 of this evaluation (a few MB)
      if (traced)
         for (int k = 0; k < 2; ++k)
         {
            int64_t cnt = 0;
            if (batotp_hip_resampled_trace_data(rs.r, 2 + k, nullptr, 0, &cnt) == BATOTP_OK && cnt > 0)
            {
               stageAgain[k].resize((size_t)cnt);
               batotp_hip_resampled_trace_data(rs.r, 2 + k, stageAgain[k].data(), cnt, &cnt);
            }
         }
      int64_t nK = 0;
      double sr = 0;
      uint32_t st = 0;
      rc = batotp_hip_resampled_info(rs.r, &nK, &sr, &st);
      if (rc) return fail("resampled_info", rc);
      yAgain.clear();
      if (!st)
      {
         yAgain.assign((size_t)nK * rowsOut, 0.0);
         rc = batotp_hip_resampled_download(rs.r, 0, yAgain.data());
         if (rc) return fail("resampled_download", rc);
      }
      double integP = 0, swP[3] = {0, 0, 0};
      int32_t scaleP = 0;
      if (_isAutoIntegRes)
      {
         rc = batotp_hip_resampled_auto(rs.r, &integP, swP, &scaleP);
         if (rc) return fail("resampled_auto", rc);
      }
      if (pass > 0)
      {
         agreed = nK == nKnots && st == status && std::memcmp(&sr, &sres, sizeof(double)) == 0 && std::memcmp(&integP, &integ, sizeof(double)) == 0 &&
                  scaleP == scale && std::memcmp(swP, sw, sizeof(sw)) == 0 && yAgain.size() == y.size() &&
                  (y.empty() || std::memcmp(yAgain.data(), y.data(), sizeof(double) * y.size()) == 0);
         if (!agreed)
         {
            size_t at = 0;
            while (at < y.size() && at < yAgain.size() && std::memcmp(&y[at], &yAgain[at], sizeof(double)) == 0) ++at;
            printf("interpInputData(): evaluations %d and %d of the device resampler disagree (knots %lld / %lld, status 0x%x / 0x%x, spacing %.17g / %.17g, "
                   "first different value at index %zu of %zu", pass, pass + 1, (long long)nKnots, (long long)nK, status, st, sres, sr, at, y.size());
            if (at < y.size() && at < yAgain.size()) printf(": %.17g / %.17g", y[at], yAgain[at]);
            printf(")%s\n", pass < 3 ? ": evaluating again." : ".");
            if (traced && haveTrace)
            {
               int firstStage = -1;
               for (int k = 0; k < 8 && firstStage < 0; ++k)
                  if (trace[k] != traceAgain[k]) firstStage = k;
               printf("interpInputData(): stage checksums of the two evaluations:");
               for (int k = 0; k < 8; ++k) printf(" [%d] %016llx/%016llx", k, (unsigned long long)trace[k], (unsigned long long)traceAgain[k]);
               printf("; first difference in stage %d (%s).\n", firstStage, firstStage >= 0 ? kStage[firstStage] : "none: the difference lies outside the traced arrays");
               if (firstStage == 2 || firstStage == 3)
               {
                  // both evaluations' arrays of that stage, for a look at WHERE they differ
                  const std::vector<double> &a = stageKept[firstStage - 2], &b = stageAgain[firstStage - 2];
                  size_t nd = 0, first = 0, last = 0;
                  for (size_t i = 0; i < a.size() && i < b.size(); ++i)
                     if (std::memcmp(&a[i], &b[i], sizeof(double)) != 0) { if (!nd) first = i; last = i; ++nd; }
                  printf("interpInputData(): stage %d arrays: %zu / %zu values, %zu differ, first at %zu, last at %zu", firstStage, a.size(), b.size(), nd, first, last);
                  char name[96];
                  for (int w = 0; w < 2; ++w)
                  {
                     std::snprintf(name, sizeof(name), "resampler_disagreement_stage%d_eval%d.bin", firstStage, pass + w);
                     FILE *f = std::fopen(name, "wb");
                     const std::vector<double> &v = w ? b : a;
                     if (f) { std::fwrite(v.data(), sizeof(double), v.size(), f); std::fclose(f); }
                  }
                  printf("; written to resampler_disagreement_stage%d_eval%d.bin / _eval%d.bin.\n", firstStage, pass, pass + 1);
               }
            }
         }
      }
      // the latest evaluation is the one the next is compared with (and the one that is used once two in a row agree)
      nKnots = nK; sres = sr; status = st; integ = integP; scale = scaleP;
      for (int k = 0; k < 3; ++k) sw[k] = swP[k];
      y.swap(yAgain);
      haveTrace = traced;
      for (int k = 0; k < 8; ++k) trace[k] = traceAgain[k];
      stageKept[0].swap(stageAgain[0]); stageKept[1].swap(stageAgain[1]);
      batotp_hip_resampled_ms(rs.r, &ms);
   }
   if (!agreed)
   {
      printf("interpInputData(): no two consecutive evaluations of the device resampler agree: refusing the result.\n");
      return -1;
   }
   if (status)
   {
      // the reference's own exits (ba.cpp:176-181, 484-488, 607-611; spline.cpp:84-88)
      if (status & BATOTP_RS_TOO_SHORT) printf("Input trajectory has less than one site after remClosePts() so no optimization will be performed.\n");
      else if (status & BATOTP_RS_IDENTICAL) printf("Input trajectory points are all identical no optimization will be performed.\n");
      else if (status & BATOTP_RS_SMALL_STEP) printf("adjust_s(): s-resolution is too small between two points. aborting... \n");
      else if (status & BATOTP_RS_SEG_ERROR) printf("Error: division by zero in findInterpSegs().\n");
      else printf("interpInputData(): the device resampler ended with status 0x%x.\n", status);
      return -1;
   }
   if (poses) _nCart = 7; // what aa2qVect leaves (reference ba.cpp:335): position + quaternion rows from here on
   traj.theta.assign(_nJoints, std::vector<double>());
   traj.cart.assign(_nCart, std::vector<double>());
   for (unsigned int j = 0; j < _nJoints; ++j) traj.theta[j].assign(y.begin() + (size_t)j * nKnots, y.begin() + (size_t)(j + 1) * nKnots);
   for (unsigned int j = 0; j < _nCart; ++j)
      traj.cart[j].assign(y.begin() + (size_t)(_nJoints + j) * nKnots, y.begin() + (size_t)(_nJoints + j + 1) * nKnots);
   traj.nPts = (unsigned int)nKnots;
   traj.sres = sres;
   if (_isAutoIntegRes)
   {
      // what the rule leaves in the BA object (reference ba.cpp:493-556)
      printf("InterpInputData(): Final integ. res is %0.6f s.\n", integ);
      _integRes = integ;
      _sWeights[1] = sw[1]; _sWeights[2] = sw[2];
      _scaleType = scale;
   }
   _lastResampleMs = ms;
   return 0;
}

// Which configurations the device output stage takes over: JOINT paths of a robot without kinematic model,
// no torque recomputation (ba.cpp:1744-1827 stays on the host).
int BA::exportOutputParams(void *out) const
{
   batotp_output_params &O = *static_cast<batotp_output_params *>(out);
   std::memset(&O, 0, sizeof(O));
   O.n_joints = (int32_t)_nJoints;
   O.path_type = (_pathType == JOINT) ? BATOTP_PATH_JOINT : (_pathType == CART ? BATOTP_PATH_CART : (_pathType == BOTH ? BATOTP_PATH_BOTH : 0));
   O.integ_res = _integRes;
   O.out_res = _outRes;
   O.out_smooth_fact = _outSmoothFact;
   if (_isInterpOnly) return -1;
   if (!(_outRes > 0) || !(_integRes > 0) || !(_outSmoothFact >= 1)) return -1;
   const bool joint = _pathType == JOINT && _robotType == GENJNT && !_isTrqConOn;
   const bool cable = _pathType == CART && _robotType == CSPR3DOF && _nJoints == 3 && _nCart == 3 && _isTrqConOn && _isParallelMechOrig;
   // JOINT paths of the robots with forward kinematics: Cartesian rows by Robot::fwdKin at the output points (reference
   // ba.cpp:1722-1725) and, with torque constraints, the serial-robot torque recomputation (ba.cpp:1791-1827) with the
   // two-link arm's closed form or the chain model
   const bool kinRobot = _pathType == JOINT && _nCart == 3 && ((_robotType == KUKA && _nJoints == 7) || (_robotType == RR && _nJoints == 2));
   const bool kin = kinRobot && (!_isTrqConOn || (!_isParallelMechOrig && (_robotType == RR || const_cast<Robot &>(myRobot).serialModel() != nullptr)));
   // joints and Cartesian rows together (ba.cpp:1726-1736: the Cartesian rows are evaluated like the joints); pose rows are quaternions
   // during the run (_nCart == 7) and leave as axis-angle (BA::q2aaVect, ba.cpp:384-403)
   const bool both = _pathType == BOTH && _nCart >= 3 && _nCart <= BATOTP_MAX_CART && !_isTrqConOn;
   return (joint || cable || kin || both) ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// reference ba.cpp:299-305 on the GPU: evalSplineFullTraj(traj, sres, sres), sdot = DBL_MAX,
// findDynModel
// ---------------------------------------------------------------------------------------------
int BA::deviceBuildKnotModel(Traj &traj)
{
   if (_nJoints > BATOTP_MAX_JOINTS || _nCart > BATOTP_MAX_CART)
   {
      printf("batotp: at most %d joints / %d Cartesian channels are supported on the device.\n",
             BATOTP_MAX_JOINTS, BATOTP_MAX_CART);
      return -1;
   }
   const bool chainModel = _isTrqConOn && !_isParallelMechOrig && _robotType != RR && myRobot.serialModel() != nullptr;
   if (_isTrqConOn && !(_robotType == RR || _robotType == CSPR3DOF || chainModel))
   {
      // the reference has no dynamics model for the other robots either (robot.cpp:349-360,452-463)
      printf("No dynamics model provided for robotType=%s.\n", _robotTypeStr.c_str());
      return -1;
   }
   if (gpuAcquire() != 0) return -1;

   const int64_t N = traj.nPts;
   if (N < 4)
   {
      printf("batotp: fewer than 4 knots after resampling.\n");
      return -1;
   }
   const int nIn = (int)(_nJoints + _nCart);
   batotp_problem prob;
   fillProblem(&prob);

   BatchGuard g;
   int rc = batotp_hip_batch_create(_gpu->ctx, &prob, 1, &N, 4, &g.b);
   if (rc) return fail("batch_create", rc);

   std::vector<double> y((size_t)nIn * N, 0.0);
   for (unsigned int j = 0; j < _nJoints; ++j) std::copy(traj.theta[j].begin(), traj.theta[j].begin() + N, y.begin() + (size_t)j * N);
   for (unsigned int j = 0; j < _nCart; ++j)
   {
      if (j < traj.cart.size() && traj.cart[j].size() >= (size_t)N)
         std::copy(traj.cart[j].begin(), traj.cart[j].begin() + N, y.begin() + (size_t)(_nJoints + j) * N);
   }
   const double sresIn = traj.sres;
   rc = batotp_hip_upload_knots(g.b, 0, 1, y.data(), &sresIn);
   if (rc) return fail("upload_knots", rc);
   rc = batotp_hip_precompute(g.b, 1);
   if (rc) return fail("precompute(kinematics)", rc);

   // --- unmarshal what evalSplineFullTraj leaves in the Traj (reference ba.cpp:794-860) ---
   traj.nPtsC = (int)N;
   traj.sC.resize(N);
   for (int64_t i = 0; i < N; ++i) traj.sC[i] = traj.sres * (double)i;
   traj.sMVC.resize(N);
   {
      const double sScale = traj.sC[N - 1] / (double)(N - 1);
      for (int64_t i = 0; i < N; ++i) traj.sMVC[i] = sScale * (double)i;
   }
   traj.sresC = traj.sres;
   traj.vFact = 1 / traj.sresC;
   traj.aFact = traj.vFact * traj.vFact;
   traj.sres = sresIn * (double)(N - 1) / (double)(N - 1);
   traj.nPts = (unsigned int)N;

   std::vector<double> flat(4 * (size_t)N), samp(3 * (size_t)N);
   traj.thetaC.resize(_nJoints);
   traj.cartC.resize(_nCart);
   traj.thetaD.resize(_nJoints);
   traj.thetaD2.resize(_nJoints);
   traj.cartD.resize(_nCart);
   traj.cartD2.resize(_nCart);
   traj.cart.resize(_nCart);
   for (int ch = 0; ch < nIn; ++ch)
   {
      rc = batotp_hip_download_coeffs(g.b, 0, ch, flat.data());
      if (rc) return fail("download_coeffs", rc);
      rc = batotp_hip_download_samples(g.b, 0, ch, samp.data());
      if (rc) return fail("download_samples", rc);
      const bool isTheta = ch < (int)_nJoints;
      const int j = isTheta ? ch : ch - (int)_nJoints;
      vectorToCoeffs(flat, (size_t)N, isTheta ? traj.thetaC[j] : traj.cartC[j]);
      std::vector<double> &v = isTheta ? traj.theta[j] : traj.cart[j];
      std::vector<double> &vD = isTheta ? traj.thetaD[j] : traj.cartD[j];
      std::vector<double> &vD2 = isTheta ? traj.thetaD2[j] : traj.cartD2[j];
      v.assign(samp.begin(), samp.begin() + N);
      vD.assign(samp.begin() + N, samp.begin() + 2 * N);
      vD2.assign(samp.begin() + 2 * N, samp.begin() + 3 * N);
   }
   {
      // the ptsOrig channel is bookkeeping of the resampler, nothing downstream reads it:
      // kept on the host (reference ba.cpp:833,857-859)
      mySpline.getSplineCoeffs(traj.ptsOrig, traj.ptsOrigC, "natural");
      Spline::splineSegs where;
      if (mySpline.findInterpSegs(traj.sC, traj.sMVC, where) == 0)
      {
         std::vector<double> d1, d2;
         mySpline.interp1spline(traj.ptsOrig, d1, d2, traj.ptsOrigC, where, sresIn);
      }
   }
   _isInterpolated = true;

   // reference ba.cpp:300
   traj.sdot.resize(traj.nPts, std::numeric_limits<double>::max());

   if (!_isTrqConOn) return 0;

   // --- findDynModel (reference ba.cpp:873-949) ---
   _dynDim = _isParallelMech ? (int)_nCart : (int)_nJoints;
   if (_isParallelMech)
   {
      traj.Apt.resize(_nCart);
      for (unsigned int r = 0; r < _nCart; ++r) traj.Apt[r].resize(_nJoints);
   }
   traj.a1.resize(_dynDim); traj.a2.resize(_dynDim); traj.a3.resize(_dynDim); traj.a4.resize(_dynDim);
   traj.a1C.resize(_dynDim); traj.a2C.resize(_dynDim); traj.a3C.resize(_dynDim); traj.a4C.resize(_dynDim);
   traj.a1pt.resize(_dynDim); traj.a2pt.resize(_dynDim); traj.a3pt.resize(_dynDim); traj.a4pt.resize(_dynDim);

   if (_robotType == RR)
   {
      // cos/sin of the knot samples with the host libm (reference robot.cpp:408-419), so that the
      // device dynamics reproduce the reference bit for bit
      std::vector<double> trig(4 * (size_t)N);
      for (int64_t i = 0; i < N; ++i)
      {
         double tr[4];
         Robot::planarRRDynTrig(_DEG2RAD * traj.theta[0][i], _DEG2RAD * traj.theta[1][i], tr);
         trig[i] = tr[0];
         trig[N + i] = tr[1];
         trig[2 * N + i] = tr[2];
         trig[3 * N + i] = tr[3];
      }
      rc = batotp_hip_upload_rr_trig(g.b, 0, trig.data());
      if (rc) return fail("upload_rr_trig", rc);
   }
   else if (chainModel)
   {
      std::vector<const double *> rows(_nJoints);
      for (unsigned int j = 0; j < _nJoints; ++j) rows[j] = traj.theta[j].data();
      if (deviceSerialDynamicsInputs(g.b, 0, rows.data(), (long long)N) != 0) return -1;
   }
   rc = batotp_hip_precompute(g.b, 2);
   if (rc) return fail("precompute(dynamics)", rc);

   std::vector<std::vector<double>> *ak[4] = {&traj.a1, &traj.a2, &traj.a3, &traj.a4};
   std::vector<Spline::splineCoeffs> *akC[4] = {&traj.a1C, &traj.a2C, &traj.a3C, &traj.a4C};
   for (int k = 0; k < 4; ++k)
   {
      for (int r = 0; r < _dynDim; ++r)
      {
         (*ak[k])[r].resize(N);
         rc = batotp_hip_download_dyn(g.b, 0, k + 1, r, (*ak[k])[r].data());
         if (rc) return fail("download_dyn", rc);
         rc = batotp_hip_download_coeffs(g.b, 0, nIn + k * _dynDim + r, flat.data());
         if (rc) return fail("download_coeffs(dyn)", rc);
         vectorToCoeffs(flat, (size_t)N, (*akC[k])[r]);
      }
   }
   if (_isParallelMech && _isPar2Ser) _isParallelMech = false; // reference ba.cpp:937
   return 0;
}

// the knot model of a Traj (sites, spline coefficient rows of joints / Cartesian channels / dynamics coefficients) into path 0
// of a batch of one -- what the step-by-step API carries from call to call in the public Traj arrays
int BA::uploadTrajSplines(void *batch, Traj &traj, int nCartChannels, bool withDynamics)
{
   batotp_batch *b = static_cast<batotp_batch *>(batch);
   const size_t N = (size_t)traj.nPtsC;
   int rc = batotp_hip_upload_path_sites(b, 0, traj.sC.data(), traj.vFact, traj.aFact, _isParallelMech ? 1 : 0);
   if (rc) return fail("upload_path_sites", rc);
   std::vector<double> flat;
   int ch = 0;
   for (unsigned int j = 0; j < _nJoints; ++j, ++ch)
   {
      coeffsToVector(traj.thetaC[j], N, flat);
      rc = batotp_hip_upload_coeffs(b, 0, ch, flat.data());
      if (rc) return fail("upload_coeffs(theta)", rc);
   }
   for (int j = 0; j < nCartChannels; ++j, ++ch)
   {
      coeffsToVector(traj.cartC[j], N, flat);
      rc = batotp_hip_upload_coeffs(b, 0, ch, flat.data());
      if (rc) return fail("upload_coeffs(cart)", rc);
   }
   if (withDynamics)
   {
      const int d = (int)traj.a1C.size();
      const std::vector<Spline::splineCoeffs> *akC[4] = {&traj.a1C, &traj.a2C, &traj.a3C, &traj.a4C};
      for (int k = 0; k < 4; ++k)
         for (int r = 0; r < d; ++r, ++ch)
         {
            coeffsToVector((*akC[k])[r], N, flat);
            rc = batotp_hip_upload_coeffs(b, 0, ch, flat.data());
            if (rc) return fail("upload_coeffs(dyn)", rc);
         }
   }
   return 0;
}

// ---------------------------------------------------------------------------------------------
// sweep (reference ba.cpp:979-1195) on the GPU, one path
// ---------------------------------------------------------------------------------------------
int BA::deviceSweep(Traj &traj)
{
   if (gpuAcquire() != 0) return -1;
   const int64_t N = traj.nPtsC;
   if (N < 2 || (int64_t)traj.sC.size() != N || traj.thetaC.size() < _nJoints)
   {
      printf("batotp: sweep() called on a trajectory without spline interpolants (call interpInputData first).\n");
      return -1;
   }
   if (_integDir != 1 && _integDir != -1)
   {
      printf("batotp: setIntegDir() must be +1 or -1.\n");
      return -1;
   }

   batotp_problem prob;
   fillProblem(&prob);
   const bool needCart = _isCartVelConOn || _isCartAccConOn;
   const bool haveDyn = _isTrqConOn && traj.a1C.size() > 0;
   prob.n_cart = needCart ? (int32_t)_nCart : 0;
   if (!haveDyn) prob.flags &= ~(uint32_t)BATOTP_F_TRQ_ON;
   if (needCart && traj.cartC.size() < _nCart)
   {
      printf("batotp: Cartesian constraints are on but the trajectory has no Cartesian interpolants.\n");
      return -1;
   }

   if (!(_integRes > 0) || std::isinf(_integRes) || !(_maxIntegTime >= 0))
   {
      // (the automatic integration resolution leaves NaN for a robot without Cartesian limits, reference ba.cpp:519-533; the
      // reference then loops on NaN state until its step budget -- itself a cast of NaN -- runs out)
      printf("Error in sweep(): the integration step (%g s) is not a positive finite number.\n", _integRes);
      setErrorOptimization(MAX_INTEGRATION_TIME);
      return -1;
   }
   const int64_t maxIntegSteps = (int64_t)std::floor(_maxIntegTime / _integRes) + 1;
   const int64_t cap = maxIntegSteps + 2;

   BatchGuard g;
   int rc = batotp_hip_batch_create(_gpu->ctx, &prob, 1, &N, cap, &g.b);
   if (rc) return fail("batch_create", rc);
   if (uploadTrajSplines(g.b, traj, prob.n_cart, haveDyn) != 0) return -1;
   if (_integDir == 1)
   {
      if (traj.sMVC.size() < 2 || traj.sMVC.size() != traj.sdot.size() || traj.nPts != traj.sMVC.size())
      {
         printf("batotp: forward sweep needs the curve of the reverse sweep in traj.sMVC / traj.sdot.\n");
         return -1;
      }
      rc = batotp_hip_upload_curve(g.b, 0, traj.sMVC.data(), traj.sdot.data(), (int64_t)traj.nPts);
      if (rc) return fail("upload_curve", rc);
   }

   rc = batotp_hip_sweep(g.b, _integDir);
   if (rc) return fail("sweep", rc);

   batotp_path_result res;
   rc = batotp_hip_get_results(g.b, &res);
   if (rc) return fail("get_results", rc);

   const bool fwd = (_integDir == 1);
   const uint32_t status = fwd ? res.status_fwd : res.status_rev;
   const int64_t nPts = fwd ? res.n_fwd : res.n_rev;
   const int64_t steps = fwd ? res.steps_fwd : res.steps_rev;
   const int nFailBisect = fwd ? res.n_bisect_fail_fwd : res.n_bisect_fail_rev;
   const double tElapsed = fwd ? res.t_total : res.t_rev;

   if (status & BATOTP_ST_MAX_INTEG_TIME)
   {
      printf("Error in sweep(): maxIntegTime of %.1f s was exceeded.\n", _maxIntegTime);
      setErrorOptimization(MAX_INTEGRATION_TIME);
      return -1;
   }
   if (status & (BATOTP_ST_CAPACITY | BATOTP_ST_NONFINITE))
   {
      printf("Error in sweep(): integration aborted on the device (status 0x%x).\n", status);
      return -1;
   }
   if (nFailBisect > 0)
   {
      // the reference prints one message per failure and keeps integrating (ba.cpp:1307-1319)
      printf("applyAccelConstraintsBisectionPt() error: %d point(s) did not respect accel constraints.\n", nFailBisect);
   }

   std::vector<double> sInteg((size_t)nPts), sdotInteg((size_t)nPts);
   int64_t got = 0;
   rc = batotp_hip_download_curve(g.b, 0, _integDir, sInteg.data(), sdotInteg.data(), nPts, &got);
   if (rc || got != nPts) return fail("download_curve", rc);

   printf("%s integ.: %4d steps; %5d ODE evals; %3d failed steps; traj time. %.3f sec.; avg. step size %f sec.\n",
          fwd ? "fwd." : "rev.", (int)(steps + 1), (int)(4 * steps), 0, tElapsed, tElapsed / (double)(steps + 1));

   // publish (reference ba.cpp:1154-1190)
   std::vector<double> tInteg((size_t)nPts);
   if (status & BATOTP_ST_SHORT)
   {
      const double tResNew = tElapsed / 3.;
      for (int64_t k = 0; k < nPts; ++k) tInteg[k] = tResNew * (double)k;
   }
   else
   {
      for (int64_t k = 0; k < nPts; ++k) tInteg[k] = _integRes * (double)k;
   }
   traj.tTotalTraj = tElapsed;
   if (is_sdotOut && traj.myMVChist.s.size() >= 2)
   {
      const int slot = _isLastSweep ? 1 : 0;
      traj.myMVChist.s[slot] = sInteg;
      traj.myMVChist.sdot[slot] = sdotInteg;
      if (status & BATOTP_ST_SHORT)
      {
         // the reference stores the curve before the 4-point fix-up; with <4 points the history is
         // not meaningful and is left as published
      }
   }
   if (_isLastSweep) traj.tMVC = tInteg;
   traj.sMVC = sInteg;
   traj.sdot = sdotInteg;
   traj.nPts = (unsigned int)nPts;
   return 0;
}

// ---------------------------------------------------------------------------------------------
// extension: many independent paths, one device batch
// ---------------------------------------------------------------------------------------------
int BA::useAllDevices()
{
   int n = 0;
   if (batotp_hip_device_count(&n) != BATOTP_OK || n < 1) return 0;
   _devices.resize((size_t)n);
   std::iota(_devices.begin(), _devices.end(), 0);
   return n;
}

// Many paths over several GPUs: contiguous blocks of paths, one host thread + one BA copy + one device context per GPU,
// no exchange between devices (the paths are independent); the blocks come back in place.
int BA::optimizeBatch(std::vector<Traj> &trajs)
{
   const size_t nDev = std::min(_devices.size(), trajs.size());
   if (nDev <= 1)
   {
      if (_devices.size() == 1) _deviceId = _devices[0];
      return optimizeBatchOnDevice(trajs);
   }
   const size_t base = trajs.size() / nDev, rem = trajs.size() % nDev;
   std::vector<std::vector<Traj>> part(nDev);
   std::vector<size_t> lo(nDev + 1, 0);
   for (size_t d = 0; d < nDev; ++d) lo[d + 1] = lo[d] + base + (d < rem ? 1 : 0);
   std::vector<BA> worker(nDev, *this);
   std::vector<int> rcs(nDev, -1);
   std::vector<std::thread> threads;
   for (size_t d = 0; d < nDev; ++d)
   {
      part[d].assign(std::make_move_iterator(trajs.begin() + lo[d]), std::make_move_iterator(trajs.begin() + lo[d + 1]));
      worker[d]._gpu.reset();            // every worker owns its context
      worker[d]._devices.clear();
      worker[d]._deviceId = _devices[d];
      threads.emplace_back([&, d]() { rcs[d] = worker[d].optimizeBatchOnDevice(part[d]); });
   }
   for (std::thread &t : threads) t.join();
   int failed = 0;
   bool broken = false;
   setErrorOptimization(NO_ERROR);
   _lastResampleMs = _lastOutputMs = _lastOutputKernelMs = 0;
   for (size_t d = 0; d < nDev; ++d)
   {
      std::move(part[d].begin(), part[d].end(), trajs.begin() + lo[d]);
      if (rcs[d] < 0) broken = true; else failed += rcs[d];
      if (worker[d].getErrorOptimization() == MAX_INTEGRATION_TIME) setErrorOptimization(MAX_INTEGRATION_TIME);
      _lastResampleMs = std::max(_lastResampleMs, worker[d]._lastResampleMs);
      _lastOutputMs = std::max(_lastOutputMs, worker[d]._lastOutputMs);
      _lastOutputKernelMs = std::max(_lastOutputKernelMs, worker[d]._lastOutputKernelMs);
   }
   // what a single-device run leaves in the object (run state a later writeOutputData / interpOutputData reads)
   _isInterpolated = worker[0]._isInterpolated;
   _isParallelMech = worker[0]._isParallelMech;
   _nCart = worker[0]._nCart;
   _outRes = worker[0]._outRes;
   _outSmoothFact = worker[0]._outSmoothFact;
   return broken ? -1 : failed;
}

// rows of one finished trajectory (joints, then Cartesian rows, then torques, each n points) into the Traj, the way
// interpOutputData leaves it (reference ba.cpp:1829-1836, 1920-1931)
void BA::unpackOutputRows(Traj &t, const double *rows, long long n, int nTheta, int nCartRows, int nTrq, double sresOut, long long nFwd, double tTotal,
                          bool shortCurve, double integRes)
{
   t.theta.assign(_nJoints, std::vector<double>());
   for (unsigned int j = 0; j < _nJoints; ++j) t.theta[j].assign(rows + (size_t)j * n, rows + (size_t)(j + 1) * n);
   t.trq.clear();
   if (nCartRows > 0)
   {
      // cable robot, robots with forward kinematics, pose paths: Cartesian rows and (torque constraints on) the recomputed
      // cable tensions / joint torques come with the joints
      if (_pathType == BOTH && nCartRows == 6) _nCart = 6;   // q2aaVect (reference ba.cpp:399): axis-angle again
      t.cart.assign(_nCart, std::vector<double>());
      for (int j = 0; j < nCartRows; ++j) t.cart[j].assign(rows + (size_t)(nTheta + j) * n, rows + (size_t)(nTheta + j + 1) * n);
      if (nTrq > 0)
      {
         t.trq.assign(_nJoints, std::vector<double>());
         for (int j = 0; j < nTrq; ++j) t.trq[j].assign(rows + (size_t)(nTheta + nCartRows + j) * n, rows + (size_t)(nTheta + nCartRows + j + 1) * n);
      }
   }
   else
   {
      // no kinematic model: the Cartesian rows are zeros that interpOutputData sizes (ba.cpp:1829-1836), smooths and
      // down-samples with the joints (ba.cpp:1861-1869) but does not re-interpolate (ba.cpp:1899); the writer keeps them
      // only if their length ends up equal to the joints'
      double outResEff = _outRes, smoothEff = _outSmoothFact;
      if (_outRes < integRes) { outResEff = integRes; smoothEff *= std::max(_outRes / outResEff, 1.); }
      const double tStep = shortCurve ? tTotal / 3. : integRes;
      const double tLast = tStep * (double)(nFwd - 1);
      int nCartPts = std::max((int)(smoothEff * std::ceil(tLast / outResEff + 1.)), 4);
      if (smoothEff > 1.5) nCartPts = std::max((int)((nCartPts - 1) / smoothEff) + 1, 4);
      t.cart.assign(_nCart, std::vector<double>((size_t)nCartPts, 0.0));
   }
   t.nPts = (unsigned int)n;
   t.sres = sresOut;
   t.tTotalTraj = tTotal;
}

// ---------------------------------------------------------------------------------------------
// reference ba.cpp:1661-1931 for ONE path behind the C-ABI: the knot model and the forward curve of the Traj go into a batch
// of one, batotp_hip_output runs, the finished rows come back
// ---------------------------------------------------------------------------------------------
int BA::deviceOutputOne(Traj &traj, const void *prmIn)
{
   const batotp_output_params &prm = *static_cast<const batotp_output_params *>(prmIn);
   if (gpuAcquire() != 0) return -1;
   const int64_t N = traj.nPtsC, nF = (int64_t)traj.sMVC.size();
   if (N < 4 || (int64_t)traj.sC.size() != N || traj.thetaC.size() < _nJoints)
   {
      printf("batotp: interpOutputData() called on a trajectory without spline interpolants (call interpInputData first).\n");
      return -1;
   }
   batotp_problem prob;
   fillProblem(&prob);
   const bool haveCart = traj.cartC.size() >= _nCart && _nCart > 0 && traj.cartC[0].c0.size() >= (size_t)N;
   const bool haveDyn = _isTrqConOn && traj.a1C.size() > 0;
   prob.n_cart = haveCart ? (int32_t)_nCart : 0;
   if (!haveDyn && !_isTrqConOn) prob.flags &= ~(uint32_t)BATOTP_F_TRQ_ON;
   BatchGuard g;
   int rc = batotp_hip_batch_create(_gpu->ctx, &prob, 1, &N, nF + 8, &g.b);
   if (rc) return fail("batch_create", rc);
   if (uploadTrajSplines(g.b, traj, prob.n_cart, haveDyn) != 0) return -1;
   if (_isTrqConOn && !_isParallelMechOrig && _robotType != RR)
   {
      if (!myRobot.serialModel()) { printf("No dynamics model provided for robotType=%s.\n", _robotTypeStr.c_str()); return -1; }
      rc = batotp_hip_set_serial_model(g.b, myRobot.serialModel());
      if (rc) return fail("set_serial_model", rc);
   }
   if (nF == 4 && traj.tMVC.size() == 4 && traj.tMVC[1] != _integRes)
   {
      // the four-point fix-up of a curve of fewer than four integration steps (ba.cpp:1171-1184) spaces its points by T/3:
      // the batch entry carries that in the sweep's status word, the step-by-step Traj does not
      printf("interpOutputData(): a curve of fewer than four integration steps goes through optimizeBatch().\n");
      return -1;
   }
   rc = batotp_hip_upload_forward_curve(g.b, 0, traj.sMVC.data(), traj.sdot.data(), nF, traj.tTotalTraj);
   if (rc) return fail("upload_forward_curve", rc);
   OutputGuard og;
   rc = batotp_hip_output(g.b, &prm, 0, 1, &og.o);
   if (rc) return fail("output", rc);
   int64_t nOut = 0;
   double sresOut = 0;
   rc = batotp_hip_output_info(og.o, &nOut, &sresOut);
   if (rc) return fail("output_info", rc);
   int32_t nTh = 0, nCa = 0, nTq = 0;
   rc = batotp_hip_output_channels(og.o, &nTh, &nCa, &nTq);
   if (rc) return fail("output_channels", rc);
   if (nOut < 1) { printf("interpOutputData(): the device output stage produced no trajectory.\n"); return -1; }
   std::vector<double> rows((size_t)nOut * (size_t)(nTh + nCa + nTq));
   rc = batotp_hip_output_download(og.o, 0, rows.data());
   if (rc) return fail("output_download", rc);
   float ms = 0;
   batotp_hip_output_ms(og.o, &ms);
   _lastOutputKernelMs = ms;
   unpackOutputRows(traj, rows.data(), nOut, nTh, nCa, nTq, sresOut, nF, traj.tTotalTraj, false, _integRes);
   if (_nCart == 7) _nCart = 6; // the pose rows left as axis-angle (reference ba.cpp:1920-1927)
   return 0;
}

int BA::optimizeBatchOnDevice(std::vector<Traj> &trajs)
{
   setErrorOptimization(NO_ERROR);
   if (trajs.empty()) return 0;
   if (_isInterpOnly || _sWeights[1] + _sWeights[2] < 1e-8)
   {
      printf("optimizeBatch(): the interpolate-only mode and paths whose s is the teach time go through optimize(), one path at a time.\n");
      return -1;
   }
   if (gpuAcquire() != 0) return -1;

   std::vector<int> ok(trajs.size(), 0);
   std::vector<int64_t> nKnots;
   std::vector<size_t> live;
   const unsigned int nCartTaught = _nCart;
   // whatever happens below, the object leaves with the number of Cartesian rows it came with (pose paths run with 7)
   struct CartGuard { unsigned int &ref; unsigned int keep; ~CartGuard() { ref = keep; } } cartGuard{_nCart, nCartTaught};
   const int nInTaught = (int)(_nJoints + _nCart);

   // 1) resampling to uniform-s knots behind the C-ABI (batotp_hip_resample); the knots never leave HBM
   ResampledGuard rs;
   batotp_resample_params rsp;
   for (size_t p = 0; p < trajs.size(); ++p)
   {
      dropRepeatedTimestamps(trajs[p]);
      if (trajs[p].nPts >= 2 && trajs[p].nPts < 4) stretchShortPath(trajs[p], 4);
      if (exportResampleParams(trajs[p], &rsp) != 0)
      {
         printf("optimizeBatch(): path %d (robotType=%s, pathType=%s, %u points) is not covered by the device resampler.\n", (int)p, _robotTypeStr.c_str(),
                _pathTypeStr.c_str(), trajs[p].nPts);
         return -1;
      }
   }
   std::vector<double> sresKnots, integOf;
   std::vector<int64_t> nAll(trajs.size());
   {
      std::vector<int64_t> nTaught(trajs.size());
      std::vector<double> sresTaught(trajs.size());
      size_t total = 0;
      for (size_t p = 0; p < trajs.size(); ++p) { nTaught[p] = trajs[p].nPts; sresTaught[p] = trajs[p].sres; total += (size_t)trajs[p].nPts; }
      std::vector<double> x(total * nInTaught, 0.0);
      size_t at = 0;
      for (size_t p = 0; p < trajs.size(); ++p)
      {
         const Traj &t = trajs[p];
         const size_t n = (size_t)t.nPts;
         for (unsigned int j = 0; j < _nJoints && j < t.theta.size(); ++j)
            if (t.theta[j].size() >= n) std::copy(t.theta[j].begin(), t.theta[j].begin() + n, x.begin() + at + (size_t)j * n);
         for (unsigned int j = 0; j < _nCart && j < t.cart.size(); ++j)
            if (t.cart[j].size() >= n) std::copy(t.cart[j].begin(), t.cart[j].begin() + n, x.begin() + at + (size_t)(_nJoints + j) * n);
         at += n * nInTaught;
      }
      // The batch is resampled TWICE and the knots are used only when the two evaluations agree -- counts, spacings, status words,
      // what the automatic rule left, and a 64-bit checksum of every path's knots computed where they lie (batotp_hip_resampled_checksums:
      // 8 bytes per path leave the device).  The same protection BA::interpInputData gives a single path, for the same reason (one
      // unexplained wrong result in ~11 000 one-path calls of round 4).  Evaluations that disagree are reported
      // path by path and the batch is evaluated again; two CONSECUTIVE evaluations must agree, four at most.  Resampling is a few
      // per cent of a batch's time.
      std::vector<double> sr(trajs.size()), integ(trajs.size());
      std::vector<uint32_t> st(trajs.size());
      std::vector<uint64_t> sums(trajs.size());
      int rcR = 0;
      bool agreed = false;
      for (int pass = 0; pass < 4 && !agreed; ++pass)
      {
         std::vector<int64_t> nP(trajs.size());
         std::vector<double> srP(trajs.size()), integP(trajs.size());
         std::vector<uint32_t> stP(trajs.size());
         std::vector<uint64_t> sumsP(trajs.size());
         if (rs.r) { batotp_hip_resampled_destroy(rs.r); rs.r = nullptr; }
         rcR = batotp_hip_resample(_gpu->ctx, &rsp, (int32_t)trajs.size(), nTaught.data(), x.data(), sresTaught.data(), &rs.r);
         if (rcR) return fail("resample", rcR);
         rcR = batotp_hip_resampled_info(rs.r, nP.data(), srP.data(), stP.data());
         if (rcR) return fail("resampled_info", rcR);
         rcR = batotp_hip_resampled_auto(rs.r, integP.data(), nullptr, nullptr);
         if (rcR) return fail("resampled_auto", rcR);
         rcR = batotp_hip_resampled_checksums(rs.r, sumsP.data());
         if (rcR) return fail("resampled_checksums", rcR);
         if (pass > 0)
         {
            size_t bad = 0, first = 0;
            for (size_t p = 0; p < trajs.size(); ++p)
            {
               const bool same = nP[p] == nAll[p] && stP[p] == st[p] && sumsP[p] == sums[p] && std::memcmp(&srP[p], &sr[p], sizeof(double)) == 0 &&
                                 std::memcmp(&integP[p], &integ[p], sizeof(double)) == 0;
               if (!same && bad++ == 0) first = p;
            }
            agreed = bad == 0;
            if (!agreed)
               printf("optimizeBatch(): evaluations %d and %d of the device resampler disagree on %zu of %zu paths (first: path %zu, knots %lld / %lld, status "
                      "0x%x / 0x%x, checksum %016llx / %016llx)%s\n", pass, pass + 1, bad, trajs.size(), first, (long long)nAll[first], (long long)nP[first],
                      st[first], stP[first], (unsigned long long)sums[first], (unsigned long long)sumsP[first], pass < 3 ? ": evaluating again." : ".");
         }
         nAll.swap(nP); sr.swap(srP); st.swap(stP); integ.swap(integP); sums.swap(sumsP);
      }
      if (!agreed)
      {
         printf("optimizeBatch(): no two consecutive evaluations of the device resampler agree: refusing the result.\n");
         return -1;
      }
      for (size_t p = 0; p < trajs.size(); ++p)
      {
         if (st[p]) continue; // the reference returns -1 for this path (identical points / degenerate s)
         ok[p] = 1;
         live.push_back(p);
         nKnots.push_back(nAll[p]);
         sresKnots.push_back(sr[p]);
         integOf.push_back(_isAutoIntegRes ? integ[p] : _integRes);
         trajs[p].sres = sr[p];
         trajs[p].nPts = (unsigned int)nAll[p];
         trajs[p].sLastSec = -1;
      }
      _isInterpolated = true;
      if (rsp.path_type == BATOTP_PATH_BOTH && _nCart == 6) _nCart = 7; // what aa2qVect leaves (reference ba.cpp:335)
      float ms = 0;
      batotp_hip_resampled_ms(rs.r, &ms);
      _lastResampleMs = ms;
   }
   if (live.empty()) return (int)trajs.size();

   batotp_problem prob;
   fillProblem(&prob);
   batotp_output_params outPrm;
   if (exportOutputParams(&outPrm) != 0)
   {
      printf("optimizeBatch(): robotType=%s with pathType=%s%s is not covered by the device output stage.\n", _robotTypeStr.c_str(), _pathTypeStr.c_str(),
             _isTrqConOn ? " and torque constraints" : "");
      return -1;
   }
   // nothing on the host reads knot samples or coefficient rows: joint velocity/acceleration-only problems keep their
   // splines as (value, second derivative) pairs (same results, less than half the memory per path)
   if (!_isTrqConOn && !_isCartVelConOn && !_isCartAccConOn) prob.flags |= BATOTP_F_NO_SAMPLES | BATOTP_F_COMPACT_SPLINES;
   // ... and so does the cable robot in serial form, whose dynamics the device evaluates itself: every channel (cable lengths,
   // platform position, a1..a4 of every row) as pairs -- 288 instead of 992 bytes per knot, twice the paths per chunk of a large batch
   else if (_isTrqConOn && _isParallelMechOrig && _isPar2Ser) prob.flags |= BATOTP_F_NO_SAMPLES | BATOTP_F_COMPACT_SPLINES;
   // the reverse curves are only read back for s-sdot.dat: without it one curve buffer per path is enough (the forward curve
   // replaces the reverse curve, as traj.sMVC / traj.sdot do in the reference)
   if (!is_sdotOut) prob.flags |= BATOTP_F_CURVES_IN_PLACE;
   const int nIn = (int)(_nJoints + _nCart);
   // integration steps a path may take (ba.cpp:984): per path with the automatic integration resolution (a NaN step --
   // the rule's result for a robot without Cartesian limits -- allows none)
   int64_t maxIntegSteps = 1;
   for (size_t k = 0; k < live.size(); ++k)
   {
      const double h = integOf[k];
      if (h > 0) maxIntegSteps = std::max<int64_t>(maxIntegSteps, (int64_t)std::floor(_maxIntegTime / h) + 1);
   }
   int64_t nMax = 0;
   for (size_t k = 0; k < nKnots.size(); ++k) nMax = std::max(nMax, nKnots[k]);
   // Capacity of a curve in points.  The single-path API allows maxIntegSteps + 2 like the reference's unbounded arrays
   // (ba.cpp:1053); a batch starts with what paths usually need (a few integration steps per knot) and, should a path run
   // out of room (BATOTP_ST_CAPACITY without BATOTP_ST_MAX_INTEG_TIME), the batch is run again with four times the room
   // -- up to maxIntegSteps + 2, where the path ends with the reference's MAX_INTEGRATION_TIME error instead.
   int64_t cap = std::min<int64_t>(maxIntegSteps + 2, 8 * nMax + 4096);
   BatchGuard g;
   std::vector<batotp_path_result> res(live.size());
   int rc = 0;
   const double *yDev = nullptr;
   std::vector<int64_t> offAll(trajs.size(), 0);
   rc = batotp_hip_resampled_knots_device(rs.r, &yDev, nullptr);
   if (rc) return fail("resampled_knots_device", rc);
   for (size_t p = 1; p < trajs.size(); ++p) offAll[p] = offAll[p - 1] + nAll[p - 1];
   for (;;)
   {
      if (g.b) { batotp_hip_batch_destroy(g.b); g.b = nullptr; }
      // this call sequence never asks for the pointwise values (K3): without the flag their array would only occupy HBM
      if (2 * cap >= 3 * nMax) prob.flags |= BATOTP_F_MVC_IN_CURVES; else prob.flags &= ~(uint32_t)BATOTP_F_MVC_IN_CURVES;
      rc = batotp_hip_batch_create(_gpu->ctx, &prob, (int32_t)live.size(), nKnots.data(), cap, &g.b);
      if (rc) return fail("batch_create", rc);
      if (_isAutoIntegRes)
      {
         rc = batotp_hip_set_path_integ_res(g.b, 0, (int32_t)live.size(), integOf.data());
         if (rc) return fail("set_path_integ_res", rc);
      }
      // runs of consecutive surviving paths are contiguous in the resampler's output
      size_t k = 0;
      while (k < live.size())
      {
         size_t e = k + 1;
         while (e < live.size() && live[e] == live[e - 1] + 1) ++e;
         rc = batotp_hip_upload_knots_device(g.b, (int32_t)k, (int32_t)(e - k), yDev + offAll[live[k]] * nIn, sresKnots.data() + k);
         if (rc) return fail("upload_knots_device", rc);
         k = e;
      }
      rc = batotp_hip_precompute(g.b, 1);
      if (rc) return fail("precompute(kinematics)", rc);
      if (_isTrqConOn)
      {
         if (_robotType == RR)
         {
            std::vector<double> samp, trig;
            for (size_t q = 0; q < live.size(); ++q)
            {
               const int64_t N = nKnots[q];
               samp.resize(6 * (size_t)N);
               trig.resize(4 * (size_t)N);
               rc = batotp_hip_download_samples(g.b, (int32_t)q, 0, samp.data());
               if (rc) return fail("download_samples", rc);
               rc = batotp_hip_download_samples(g.b, (int32_t)q, 1, samp.data() + 3 * N);
               if (rc) return fail("download_samples", rc);
               for (int64_t i = 0; i < N; ++i)
               {
                  double tr[4];
                  Robot::planarRRDynTrig(_DEG2RAD * samp[i], _DEG2RAD * samp[3 * N + i], tr);
                  trig[i] = tr[0];
                  trig[N + i] = tr[1];
                  trig[2 * N + i] = tr[2];
                  trig[3 * N + i] = tr[3];
               }
               rc = batotp_hip_upload_rr_trig(g.b, (int32_t)q, trig.data());
               if (rc) return fail("upload_rr_trig", rc);
            }
         }
         else if (!_isParallelMechOrig)
         {
            if (!myRobot.serialModel())
            {
               printf("No dynamics model provided for robotType=%s.\n", _robotTypeStr.c_str());
               return -1;
            }
            std::vector<double> samp;
            std::vector<const double *> rows(_nJoints);
            for (size_t q = 0; q < live.size(); ++q)
            {
               const int64_t N = nKnots[q];
               samp.resize(3 * (size_t)N * _nJoints);
               for (unsigned int j = 0; j < _nJoints; ++j)
               {
                  rc = batotp_hip_download_samples(g.b, (int32_t)q, (int32_t)j, samp.data() + 3 * (size_t)N * j);
                  if (rc) return fail("download_samples", rc);
                  rows[j] = samp.data() + 3 * (size_t)N * j;
               }
               if (deviceSerialDynamicsInputs(g.b, (int)q, rows.data(), (long long)N) != 0) return -1;
            }
         }
         rc = batotp_hip_precompute(g.b, 2);
         if (rc) return fail("precompute(dynamics)", rc);
      }
      rc = batotp_hip_sweep(g.b, -1);
      if (rc) return fail("sweep(-1)", rc);
      rc = batotp_hip_sweep(g.b, +1);
      if (rc) return fail("sweep(+1)", rc);

      rc = batotp_hip_get_results(g.b, res.data());
      if (rc) return fail("get_results", rc);

      size_t outOfRoom = 0, firstOut = 0;
      for (size_t q = 0; q < res.size(); ++q)
      {
         const uint32_t st = res[q].status_rev | res[q].status_fwd;
         // with one curve buffer per path (BATOTP_F_CURVES_IN_PLACE) the forward sweep gives up as soon as its curve comes
         // within 64 points of the reverse points still to be read -- long before steps_fwd reaches cap
         const bool inPlaceFull = (prob.flags & BATOTP_F_CURVES_IN_PLACE) && res[q].n_rev > 0 && (res[q].status_fwd & BATOTP_ST_CAPACITY) &&
                                  !(res[q].status_fwd & BATOTP_ST_NONFINITE);
         if ((st & BATOTP_ST_CAPACITY) && !(st & BATOTP_ST_MAX_INTEG_TIME) && !(res[q].status_rev & BATOTP_ST_NONFINITE) &&
             (res[q].n_rev == 0 || res[q].n_fwd == 0) && (res[q].steps_rev + 1 >= cap || res[q].steps_fwd + 1 >= cap || inPlaceFull))
         {
            if (outOfRoom == 0) firstOut = q;
            ++outOfRoom;
         }
      }
      if (outOfRoom == 0 || cap >= maxIntegSteps + 2) break;
      const int64_t capNext = std::min<int64_t>(maxIntegSteps + 2, 4 * cap);
      printf("optimizeBatch(): %d path(s) (first: path %d) need more than %lld integration steps (%.1f per knot): running the batch "
             "again with room for %lld.\n", (int)outOfRoom, (int)live[firstOut], (long long)cap, (double)cap / (double)nKnots[firstOut],
             (long long)capNext);
      cap = capNext;
   }
   batotp_hip_resampled_destroy(rs.r);
   rs.r = nullptr;

   // 3) output stage behind the C-ABI (batotp_hip_output): only the finished trajectories come back, in ranges of paths
   //    that integrate with the same step (one step for the whole batch unless the automatic integration resolution is
   //    on) and of at most 1024 paths, so that their device copy stays small
   const std::chrono::steady_clock::time_point tOut0 = std::chrono::steady_clock::now();
   double kernelMs = 0;
   const size_t range = 1024;
   std::vector<double> flatAll;
   std::vector<int64_t> nPtsOut;
   std::vector<double> sresOut;
   for (size_t k0 = 0; k0 < live.size();)
   {
      size_t cnt = 1;
      while (k0 + cnt < live.size() && cnt < range && integOf[k0 + cnt] == integOf[k0]) ++cnt;
      const double h = integOf[k0];
      if (!(h > 0))
      {
         // no integration step, no trajectory (the rule gave NaN: a robot without Cartesian limits)
         for (size_t q = 0; q < cnt; ++q) ok[live[k0 + q]] = 0;
         k0 += cnt;
         continue;
      }
      outPrm.integ_res = h;
      OutputGuard og;
      rc = batotp_hip_output(g.b, &outPrm, (int32_t)k0, (int32_t)cnt, &og.o);
      if (rc) return fail("output", rc);
      nPtsOut.assign(cnt, 0);
      sresOut.assign(cnt, 0.0);
      rc = batotp_hip_output_info(og.o, nPtsOut.data(), sresOut.data());
      if (rc) return fail("output_info", rc);
      float ms = 0;
      batotp_hip_output_ms(og.o, &ms); // timing only
      kernelMs += ms;
      int32_t nTh = 0, nCa = 0, nTq = 0;
      rc = batotp_hip_output_channels(og.o, &nTh, &nCa, &nTq);
      if (rc) return fail("output_channels", rc);
      const size_t rows = (size_t)(nTh + nCa + nTq);
      size_t rangePts = 0;
      for (size_t q = 0; q < cnt; ++q) rangePts += (size_t)nPtsOut[q];
      flatAll.resize(rangePts * rows);
      rc = batotp_hip_output_download_all(og.o, flatAll.data()); // one copy for the whole range of paths
      if (rc) return fail("output_download_all", rc);
      size_t at = 0;
      for (size_t q = 0; q < cnt; ++q)
      {
         const size_t k = k0 + q;
         Traj &t = trajs[live[k]];
         const batotp_path_result &r = res[k];
         if ((r.status_rev | r.status_fwd) & BATOTP_ST_MAX_INTEG_TIME) setErrorOptimization(MAX_INTEGRATION_TIME);
         const int64_t n = nPtsOut[q];
         if (n == 0) { ok[live[k]] = 0; continue; }
         _nCart = (rsp.path_type == BATOTP_PATH_BOTH && nCartTaught == 6) ? 7 : nCartTaught;
         unpackOutputRows(t, flatAll.data() + at, n, nTh, nCa, nTq, sresOut[q], r.n_fwd, r.t_total, (r.status_fwd & BATOTP_ST_SHORT) != 0, h);
         at += (size_t)n * rows;
         if (is_sdotOut)
         {
            int64_t got = 0;
            t.myMVChist.s.assign(4, std::vector<double>());
            t.myMVChist.sdot.assign(4, std::vector<double>());
            t.myMVChist.s[0].resize((size_t)r.n_rev); t.myMVChist.sdot[0].resize((size_t)r.n_rev);
            t.myMVChist.s[1].resize((size_t)r.n_fwd); t.myMVChist.sdot[1].resize((size_t)r.n_fwd);
            rc = batotp_hip_download_curve(g.b, (int32_t)k, -1, t.myMVChist.s[0].data(), t.myMVChist.sdot[0].data(), r.n_rev, &got);
            if (rc || got != r.n_rev) return fail("download_curve(rev)", rc);
            rc = batotp_hip_download_curve(g.b, (int32_t)k, +1, t.myMVChist.s[1].data(), t.myMVChist.sdot[1].data(), r.n_fwd, &got);
            if (rc || got != r.n_fwd) return fail("download_curve(fwd)", rc);
         }
      }
      k0 += cnt;
   }
   _lastOutputKernelMs = kernelMs;
   _lastOutputMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tOut0).count();
   if (_isParallelMechOrig && _isPar2Ser && _isTrqConOn) _isParallelMech = false; // reference ba.cpp:937
   int failed = 0;
   for (size_t p = 0; p < trajs.size(); ++p) failed += ok[p] ? 0 : 1;
   return failed;
}

} // namespace BATOTP
