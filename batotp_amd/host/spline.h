// spline.h -- uniform-knot cubic spline used by the host pre/post-processing of BA.
//
// Interface parity with reference batotp/spline.h:39-70 (class Spline, splineCoeffs, splineSegs,
// the four public methods).  The final per-knot coefficient build of the hot path does NOT run
// here: BA::interpInputData hands it to the HIP kernels through include/batotp_hip.h.  This class
// serves the sequential host resampling (adjust_s / interpSpecial) and interpOutputData.
#ifndef BATOTP_AMD_SPLINE_H
#define BATOTP_AMD_SPLINE_H

#include <cstdio>
#include <string>
#include <vector>

class Spline
{
public:
   // per-segment cubic  c3 t^3 + c2 t^2 + c1 t + c0,  t in [0,1] over unit knot spacing
   struct splineCoeffs
   {
      std::vector<double> c0;
      std::vector<double> c1;
      std::vector<double> c2;
      std::vector<double> c3;
   };
   // segment index + normalised position for each output site
   struct splineSegs
   {
      std::vector<int> seg;
      std::vector<double> tau;
   };

   Spline(void) {}
   ~Spline(void) {}

   int getSplineCoeffs(const std::vector<double> &y, splineCoeffs &yC, const std::string endCond);
   int findInterpSegs(const std::vector<double> &aIn, const std::vector<double> &aOut,
                      splineSegs &mySegs);
   int interp1linear(std::vector<double> &b, const splineSegs &mySegs);
   int interp1spline(std::vector<double> &b, std::vector<double> &bD, std::vector<double> &bD2,
                     splineCoeffs &bC, const splineSegs &mySegs, const double tfact);

private:
   // second-derivative solves of the (1,4,1) system
   static void secondDerivsNatural(std::vector<double> &rhs);
   static void secondDerivsClamped(std::vector<double> &rhs);
};

#endif // BATOTP_AMD_SPLINE_H
