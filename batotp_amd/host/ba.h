// ba.h -- BATOTP::BA / BATOTP::Traj, the drop-in boundary of the MI355X implementation.
//
// Source-compatible with reference batotp/ba.h:59-255: same namespace, the same public aggregate
// Traj (field names and types), the same public methods, setters, getters, Config defaults and
// ErrorOptimization enum, so that test/main.cpp (batest) and any other user of the reference
// header recompile unchanged.  What differs is underneath:
//   * interpInputData() does the sequential path resampling on the host and then hands the
//     final per-knot spline / dynamics build (reference ba.cpp:299-305) to the HIP kernels;
//   * sweep() marshals the Traj coefficient arrays to the GPU, runs the sweep kernel through the
//     C-ABI of include/batotp_hip.h and unmarshals the integrated curve.
// There is no CPU fallback: without a usable HIP device these calls fail (return -1 and print
// the reason).
// Extension (not in the reference): optimizeBatch() runs many independent paths as one device
// batch; see also batotp_amd/host/README.md.
#ifndef BATOTP_AMD_BA_H
#define BATOTP_AMD_BA_H

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "config.h"
#include "robot.h"
#include "spline.h"

namespace BATOTP
{

// scale factors handed from adjust_s() to interpSpecial() (reference ba.h:49-57)
struct InterpVars
{
   double tTeachFact;
   double thetaNormFact;
   double cartPosNormFact;
   double sLast;
   double sResNew;
   double sResi;
};

// One trajectory: inputs, spline interpolants, the (s, sdot) curve and scratch of the sweep.
// All-public aggregate, mutated in place by BA (reference ba.h:59-153).
struct Traj
{
   // ---- input / bookkeeping -------------------------------------------------------------
   double tresInput;        // [s] resolution of the recorded trajectory
   double sres;             // resolution of sMVC
   unsigned int nPts;       // number of points on the trajectory
   double tTotalTraj;       // traversal time after the last sweep
   std::string trajFileName;
   std::vector<std::string> trajFileHeader; // header of a CSV input
   std::vector<double> timestamp;           // nPts input timestamps

   // number of original (taught) points before each current point, and its spline
   std::vector<double> ptsOrig;
   Spline::splineCoeffs ptsOrigC;

   // ---- path samples vs. s ---------------------------------------------------------------
   std::vector<std::vector<double>> theta;   // joint positions
   std::vector<std::vector<double>> thetaD;  // d/ds
   std::vector<std::vector<double>> thetaD2; // d2/ds2
   std::vector<std::vector<double>> cart;    // Cartesian positions
   std::vector<std::vector<double>> cartD;
   std::vector<std::vector<double>> cartD2;

   // ---- dynamic model: tau = a1 sddot + a2 sdot^2 + a3 sdot + a4 --------------------------
   std::vector<std::vector<double>> trq;
   std::vector<std::vector<double>> a1;
   std::vector<std::vector<double>> a2;
   std::vector<std::vector<double>> a3;
   std::vector<std::vector<double>> a4;

   // ---- cursor on the maximum-velocity curve (MVC) ---------------------------------------
   int curSegMVC;     // current MVC segment
   double tauMVC;     // normalised position on it
   double sCur;       // current s
   int i;             // current point
   double sdotCur;
   bool sdotLimTypeT; // sdot was limited by the MVC / velocity limits
   double sddotH;     // admissible sddot interval from the bisection
   double sddotL;

   // ---- the integrated curve (variable s resolution after a sweep) -----------------------
   std::vector<double> sMVC;
   std::vector<double> tMVC;
   std::vector<double> sdot;
   std::vector<double> sddot;
   Spline::splineCoeffs sdotC;

   // history of the two integrated curves (written to s-sdot.dat)
   struct MVChist
   {
      std::vector<std::vector<double>> s;
      std::vector<std::vector<double>> sdot;
   } myMVChist;

   // ---- point buffers of the sweep ---------------------------------------------------------
   std::vector<double> thetapt;
   std::vector<double> thetaDpt;
   std::vector<double> thetaD2pt;
   std::vector<double> a1pt;
   std::vector<double> a2pt;
   std::vector<double> a3pt;
   std::vector<double> a4pt;
   std::vector<double> cartpt;
   std::vector<double> cartDpt;
   std::vector<double> cartD2pt;
   std::array<double, 3> CartAccCoeffs;
   std::vector<std::vector<double>> Apt;

   double sLastSec;
   bool isOn_sdot = false;

   // ---- spline interpolants of the path ------------------------------------------------------
   int nPtsC;      // number of knots
   double sresC;   // knot spacing
   double vFact;   // first-derivative scale (1/sresC)
   double aFact;   // second-derivative scale
   int curSegC;    // current spline segment
   double tauC;    // normalised position on it
   std::vector<double> sC;
   std::vector<Spline::splineCoeffs> thetaC;
   std::vector<Spline::splineCoeffs> a1C;
   std::vector<Spline::splineCoeffs> a2C;
   std::vector<Spline::splineCoeffs> a3C;
   std::vector<Spline::splineCoeffs> a4C;
   std::vector<Spline::splineCoeffs> cartC;
};

// defaults of BA::Config (reference ba.h:157-161)
static const std::vector<double> jntVelLims_init(6, 190);
static const std::vector<double> jntAccLims_init(6, 500);
static const std::vector<double> zeros6(6, 0);
static const std::array<double, 3> sWeights_initArr = {{0, 0.1, 1}};
static const std::vector<double> sWeights_init(sWeights_initArr.begin(), sWeights_initArr.end());

class BA
{
public:
   BA(void);
   ~BA(void);

   struct Config;
   enum ErrorOptimization { NO_ERROR, MAX_INTEGRATION_TIME };

   int readConfigData(const char *filename);
   int loadConfigData(const Config &conf);
   int loadTrajectoryData(Traj &traj);
   int interpInputData(Traj &myTraj);
   int sweep(Traj &myTraj);
   int interpOutputData(Traj &myTraj);
   int writeOutputData(Traj &myTraj);
   int optimize(Traj &myTraj);

   // Extension: resample every path, run precompute + both sweeps and the output stage for all of them as ONE device
   // batch.  With the automatic integration resolution (the class default) every path integrates with the step the rule
   // derives from it (batotp_hip_set_path_integ_res); the BA object's own _integRes / _sWeights / _scaleType are left
   // as configured.  Returns the number of
   // paths that failed (0 = all good), -1 if the batch could not be run at all.
   int optimizeBatch(std::vector<Traj> &trajs);
   // Extension: select the HIP device used by this object (default 0).
   void setDevice(int device) { _deviceId = device; }
   // Extension: optimizeBatch() over several GPUs.  The paths are independent, so the batch is cut into contiguous blocks
   // (first devices take the remainder, as batotp_amd/dist.py shard_range), every block runs on its own device from its own
   // host thread with its own context, and nothing is exchanged between devices.  An empty list (default)
   // keeps the single device of setDevice(); useAllDevices() lists every visible device and returns their number.
   void setDevices(const std::vector<int> &devices) { _devices = devices; }
   int useAllDevices();
   // (round 3 switch between a host and a device resampler; since round 4 the resampling always runs behind the C-ABI --
   // batotp_hip_resample -- and the call is kept for source compatibility only)
   void setDeviceResample(bool on) { _deviceResample = on; }
   // milliseconds the last optimizeBatch() spent resampling (device kernels, or host wall clock)
   double getLastResampleMs() const { return _lastResampleMs; }
   // (likewise for the output stage, batotp_hip_output; kept for source compatibility)
   void setDeviceOutput(bool on) { _deviceOutput = on; }
   // milliseconds the last optimizeBatch() spent in the output stage: wall clock incl. downloads, and device kernels
   double getLastOutputMs() const { return _lastOutputMs; }
   double getLastOutputKernelMs() const { return _lastOutputKernelMs; }
   // Extension: the host half of interpInputData() only (everything before reference
   // ba.cpp:299): leaves the final knot values in traj.theta / traj.cart, the knot spacing in
   // traj.sres and the knot count in traj.nPts.  No device call.
   int resampleToKnots(Traj &traj) { return prepareKnots(traj); }
   // Extension: the POD description of the current configuration that is handed to the device
   // layer (struct batotp_problem of include/batotp_hip.h).
   void exportProblem(void *batotp_problem_out) const { fillProblem(batotp_problem_out); }
   // Extension: the resampling parameters (struct batotp_resample_params) of the current
   // configuration.  Returns 0 when batotp_hip_resample (include/batotp_hip.h) covers this configuration and `traj`,
   // -1 otherwise (interpInputData then fails with a message: there is no host resampler any more).
   int exportResampleParams(const Traj &traj, void *batotp_resample_params_out) const;
   // Extension: the first step of interpInputData (ba.cpp:100-127): drop samples with a repeated timestamp and take the
   // input resolution from the timestamps.  To be called before exportResampleParams on freshly loaded data.
   void dropRepeatedTimestamps(Traj &traj);
   // Extension: the output-stage parameters (struct batotp_output_params).  Returns 0 when the device output
   // stage covers this configuration (batotp_hip_output, include/batotp_hip.h), -1 when interpOutputData has
   // to run on the host.
   int exportOutputParams(void *batotp_output_params_out) const;
   // Extension: torque limits on a serial robot other than the two-link arm (the reference's Robot::dynSerial,
   // robot.cpp:349-360, knows RR only; KUKA with isTrqConOn segfaults there).  model = struct batotp_serial_model of
   // include/batotp_hip.h (link table of a chain of revolute joints).  Robot type KUKA has a built-in table
   // (include/batotp_models.h: LWR IV+, nominal inertial parameters) that is used when none is set.
   void setSerialModel(const void *batotp_serial_model_in);
   unsigned int getNumJoints() const { return _nJoints; }
   unsigned int getNumCart() const { return _nCart; }

   // setters
   inline void setIsLastSweep(bool isLastSweep) { _isLastSweep = isLastSweep; }
   inline void setIntegDir(int integDir) { _integDir = integDir; }
   inline void setIsInterpOnly(bool isInterpOnly) { _isInterpOnly = isInterpOnly; }
   inline void setCartesianMaximalVelocity(const double &velocity) { _CartVelMax = velocity; }
   inline void setCartesianMaximalAcceleration(const double &acceleration) { _CartAccMax = acceleration; }
   inline void setJointMaximalVelocity(const std::vector<double> &velocity) { _JntVelMax = velocity; }
   inline void setJointMaximalAcceleration(const std::vector<double> &acceleration) { _JntAccMax = acceleration; }
   inline void setIsAutoIntegRes(const bool isAutoIntegRes) { _isAutoIntegRes = isAutoIntegRes; }
   inline void setHomeFolder(const std::string &HomeFolder) { _HomeFolder = HomeFolder; }
   inline void setInputFolder(const std::string &InputFolder) { _InputFolder = InputFolder; }
   inline void setOutputFolder(const std::string &OutputFolder) { _OutputFolder = OutputFolder; }

   // getters
   inline double getCartesianMaximalVelocity() const { return _CartVelMax; }
   inline double getCartesianMaximalAcceleration() const { return _CartAccMax; }
   inline std::vector<double> getJointMaximalVelocity() const { return _JntVelMax; }
   inline std::vector<double> getJointMaximalAcceleration() const { return _JntAccMax; }
   inline ErrorOptimization getErrorOptimization() const { return _errorOptimization; }
   inline double getOutTimeRes() const { return _outRes; }
   inline std::string getHomeFolder() const { return _HomeFolder; }
   inline std::string getInputFolder() const { return _InputFolder; }
   inline std::string getOutputFolder() const { return _OutputFolder; }

   // programmatic alternative to config.dat (reference ba.h:213-255, same defaults)
   struct Config
   {
      std::string robotTypeStr = "UR";
      bool isParallelMech = false;
      int nJoints = 6;
      int nCart = 6;
      std::string trajFileName = "urtraj.csv";
      bool isBinFile = false;
      std::string pathType = "BOTH";

      // constraints
      bool isJntVelConon = true;
      std::vector<double> jntVelLims = jntVelLims_init;
      bool isJntAccConOn = true;
      std::vector<double> jntAccLims = jntAccLims_init;
      bool isTrqConOn = false;
      std::vector<double> jntTrqMax = zeros6;
      std::vector<double> jntTrqMin = zeros6;
      bool isCartVelConOn = true;
      double cartVelMax = 0.4;
      bool isCarAccConOn = true;
      double cartAccMax = 5.0;

      // integration
      double integRes = 0.016;
      double maxIntegTime = 60000;

      // other controls
      int inputDecimFact = 1;
      int smoothWindow = 1;
      bool is_sdotOut = false;
      double jntThresh = 1e-6;
      double cartThresh = 1e-6;
      std::vector<double> sWeights = sWeights_init;
      int scaleType = 2;
      double thetaNormRes = 0.01;
      double thetaNormRes2 = 0.01;
      double cartNormRes = 0.002;
      double cartNormRes2 = 0.002;
      double outRes = 0.008;
      int outSmoothFact = 1;
      bool isSVD = false;
      bool isPar2Ser = false;
   };

private:
   // ---- configuration (config.dat order; input/README_for_config_file.txt) ------------------
   std::string _robotTypeStr;
   bool _isParallelMech = false;
   unsigned int _nJoints = 0;
   unsigned int _nCart = 0;
   std::string _trajFileName;
   bool _isBINfile = false;
   int _pathType = 0;
   std::string _pathTypeStr;

   bool _areJointAnglesDegrees = false;
   bool _isJntVelConOn = false;
   std::vector<double> _JntVelMax;
   bool _isJntAccConOn = false;
   std::vector<double> _JntAccMax;
   bool _isTrqConOn = false;
   std::vector<double> _JntTrqMax;
   std::vector<double> _JntTrqMin;
   bool _isCartVelConOn = false;
   double _CartVelMax = 0;
   bool _isCartAccConOn = false;
   double _CartAccMax = 0;

   double _integRes = 0;
   double _maxIntegTime = 0;

   int _inputDecimFact = 1;
   int _smoothWindow = 1;
   bool is_sdotOut = false;
   double _jntThresh = 0;
   double _cartThresh = 0;
   std::vector<double> _sWeights;
   int _scaleType = 0;
   double _thetaNormRes = 0;
   double _thetaNormRes2 = 0;
   double _cartNormRes = 0;
   double _cartNormRes2 = 0;
   double _outRes = 0;
   double _outSmoothFact = 1;
   bool _isSVD = false;
   bool _isPar2Ser = false;

   // ---- run state ------------------------------------------------------------------------------
   bool _isInterpOnly = false;
   bool _isParallelMechOrig = false;
   bool _isGenericRobot = false;
   bool _isAutoIntegRes = true;
   bool _isInterpolated = false;
   bool _isLastSweep = false;
   int _robotType = 0;
   int _integDir = 1;
   int _dynDim = 0;
   double _quadraticRadThresh = 0;

   std::string _HomeFolder;
   std::string _InputFolder;
   std::string _OutputFolder;
   ErrorOptimization _errorOptimization = NO_ERROR;

   Spline mySpline;
   Robot myRobot;

   // ---- device seam ------------------------------------------------------------------------------
   struct Gpu; // owns the batotp_ctx (ba_device.cpp)
   std::shared_ptr<Gpu> _gpu;
   bool _deviceResample = true;
   double _lastResampleMs = 0;
   bool _deviceOutput = true;
   double _lastOutputMs = 0, _lastOutputKernelMs = 0;
   int _deviceId = 0;
   std::vector<int> _devices;              // optimizeBatch over several GPUs (empty: _deviceId only)
   int optimizeBatchOnDevice(std::vector<Traj> &trajs);
   int gpuAcquire();                       // create the context on first use; -1 + message on failure
   void fillProblem(void *prob) const;     // BA configuration -> batotp_problem
   int deviceBuildKnotModel(Traj &traj);   // ba.cpp:299-305 on the GPU (B = 1)
   int deviceSweep(Traj &traj);            // ba.cpp:979-1195 on the GPU (B = 1)
   int uploadTrajSplines(void *batch, Traj &traj, int nCartChannels, bool withDynamics); // knot model of a Traj -> path 0 of a batch
   void unpackOutputRows(Traj &t, const double *rows, long long n, int nTheta, int nCartRows, int nTrq, double sresOut, long long nFwd,
                         double tTotal, bool shortCurve, double integRes); // finished rows -> Traj (reference ba.cpp:1829-1836, 1920-1931)
   // serial-chain dynamics on the device: set the model on the batch and upload the host cosines / sines of the
   // joint angles of path k (samples = knot values [nJoints][N], row stride N)
   int deviceSerialDynamicsInputs(void *batch, int pathIndex, const double *const *thetaRows, long long N);

   inline void setErrorOptimization(const ErrorOptimization &e) { _errorOptimization = e; }

   // ---- before the hot path (ba_input.cpp: bookkeeping; the resampling itself runs behind the C-ABI) -------------
   int prepareKnots(Traj &traj);           // interpInputData up to the final knot grid
   void stretchShortPath(Traj &traj, unsigned int nNew); // fewer than four points (reference ba.cpp:2768-2794)
   int interpolateOnly(Traj &traj);        // interpolate-only mode (reference ba.cpp:139-159)
   int keepTaughtSpacing(Traj &traj);      // s = teach time: adjust_s has nothing to do (reference ba.cpp:416)
   int deviceResampleOne(Traj &traj);      // reference ba.cpp:160-297 through batotp_hip_resample (B = 1)
   int deviceOutputOne(Traj &traj, const void *batotp_output_params_in); // reference ba.cpp:1661-1931 through batotp_hip_output (B = 1)

   // ---- file IO (ba_io.cpp) -------------------------------------------------------------------------
   int finishConfig();                     // derived settings + checks shared by readConfigData / loadConfigData
   int trajReadBIN(Traj &myTraj, const char *filename);
   int trajReadCSV(Traj &myTraj, const char *filename);
   int printInputData(const Traj &myTraj);
   int trajWriteBIN(Traj &myTraj, const char *fname);
   int trajWriteCSV(Traj &myTraj, const char *fname);
   int sdotWrite(Traj &myTraj, const char *fname);
};

} // namespace BATOTP

#endif // BATOTP_AMD_BA_H
