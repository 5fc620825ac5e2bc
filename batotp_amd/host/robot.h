// robot.h -- the five built-in robot models of BA on the host side.
//
// Interface parity with reference batotp/robot.h:33-132: the robot/path id constants and the
// public call_* methods of class Robot.  Host use: kinematics during path resampling
// (BA::interpInputData / adjust_s) and torque recomputation in BA::interpOutputData.  The per-knot
// dynamics of the hot path (dynRR, dynCSPR3DOF, setA) run on the GPU; this class only supplies
// them to the host post-processing and hands the CSPR anchor matrix to the device problem.
#ifndef BATOTP_AMD_ROBOT_H
#define BATOTP_AMD_ROBOT_H

#include <string>
#include <vector>

#include "batotp_hip.h" // POD batotp_serial_model only

namespace BATOTP
{

// robot type
static const int KUKA = 1;
static const int UR = 2;
static const int RR = 3;
static const int CSPR3DOF = 4;
static const int GENJNT = 5;

// path type
static const int JOINT = 1;
static const int CART = 2;
static const int BOTH = 3;

class Robot
{
public:
   typedef std::vector<std::vector<double>> Channels; // [channel][sample]

   int call_set_robotType(const std::string &robotTypeStr);
   int call_fwdKin(const Channels &theta, Channels &cart);
   int call_invKin(Channels &theta, const Channels &cart);
   int call_dynSerial(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                      const Channels &theta, const Channels &thetaD, const Channels &thetaD2);
   int call_dynParallel(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                        const Channels &cart, const Channels &cartD, const Channels &cartD2);
   int call_setA(const std::vector<double> &theta, const std::vector<double> &cart, Channels &A);

   // extension: CSPR cable anchor matrix, row-major [3][3] (built on first use)
   const std::vector<std::vector<double>> &cableAnchors();
   // extension: dynamics of a serial chain other than the two-link arm (call_dynSerial has the RR case only,
   // reference robot.cpp:349-360).  Without a user-supplied table the built-in one of the robot type is used
   // (include/batotp_models.h: KUKA LWR IV+).  serialModel() returns nullptr when there is neither.
   void setSerialModel(const batotp_serial_model &m) { _serial = m; _hasSerial = true; }
   const batotp_serial_model *serialModel();
   // extension: the trigonometry of the built-in models, spelled out call by call -- it is what the reference's optimised
   // build calls, and the tables the device layer uploads (BATOTP_F_HOST_TRIG) are made with the same helpers.
   //   kinSinCos        one glibc sincos(): "c=cos(t); s=sin(t);" of the forward kinematics (reference robot.cpp:130-136,198-199)
   //   planarRRDynTrig  out = cos(th1), cos(th2), cos(th1+th2), sin(th2) of Robot::dynRR (robot.cpp:408-419): th2's cosine and
   //                    sine as one sincos(), the other two as cos()
   static void kinSinCos(double angle, double *sinOut, double *cosOut);
   static void planarRRDynTrig(double th1, double th2, double out[4]);

private:
   int _kind = 0;
   std::string _kindName;
   Channels _anchors;
   batotp_serial_model _serial;
   bool _hasSerial = false, _serialProbed = false;
   void serialChainDynamics(const batotp_serial_model &m, Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                            const Channels &theta, const Channels &thetaD, const Channels &thetaD2) const;

   void kukaToolPoint(const Channels &theta, Channels &cart) const;
   void planarRRToolPoint(const Channels &theta, Channels &cart) const;
   void csprCableLengths(Channels &theta, const Channels &cart);
   void buildCsprAnchors();
   void planarRRDynamics(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                         const Channels &theta, const Channels &thetaD,
                         const Channels &thetaD2) const;
   void csprDynamics(Channels &a1, Channels &a2, Channels &a3, Channels &a4, const Channels &cartD,
                     const Channels &cartD2) const;
};

} // namespace BATOTP

#endif // BATOTP_AMD_ROBOT_H
