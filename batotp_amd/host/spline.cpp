// spline.cpp -- host cubic-spline helper (see spline.h).
//
// Arithmetic follows reference batotp/spline.cpp operation for operation, including its two
// non-textbook end treatments, because the resampled knots it produces decide N and every later
// bit of the sweep:
//   * "natural": the last second derivative is obtained by one more elimination step with zero
//     right-hand side rather than being forced to zero (reference spline.cpp:269);
//   * "clamped": the back substitution starts at row n-3 (reference spline.cpp:240).
#include "spline.h"

// reference spline.cpp:252-276
void Spline::secondDerivsNatural(std::vector<double> &rhs)
{
   const size_t last = rhs.size() - 1;
   std::vector<double> up(last, 1.0); // super-diagonal after elimination
   const double diag = 4.0, sub = 1.0;

   up[1] /= diag;
   rhs[1] /= diag;
   for (size_t i = 2; i < last; ++i)
   {
      const double piv = diag - sub * up[i - 1];
      up[i] /= piv;
      rhs[i] = (rhs[i] - sub * rhs[i - 1]) / piv;
   }
   rhs[last] = (rhs[last] - sub * rhs[last - 1]) / (diag - sub * up[last - 1]);
   for (size_t i = last; i > 1; --i) rhs[i - 1] -= up[i - 1] * rhs[i];
}

// reference spline.cpp:225-243
void Spline::secondDerivsClamped(std::vector<double> &rhs)
{
   const int n = (int)rhs.size();
   std::vector<double> up(n, 1.0);
   std::vector<double> diag(n, 4.0);
   const double sub = 1.0;
   diag[0] = 2.0;
   diag[n - 1] = 2.0;
   up[0] /= diag[0];
   rhs[0] /= diag[0];
   for (int i = 1; i < n; ++i)
   {
      const double piv = diag[i] - sub * up[i - 1];
      up[i] /= piv;
      rhs[i] = (rhs[i] - sub * rhs[i - 1]) / piv;
   }
   for (int i = n - 3; i >= 0; --i) rhs[i] -= up[i] * rhs[i + 1];
}

// reference spline.cpp:168-211
int Spline::getSplineCoeffs(const std::vector<double> &y, splineCoeffs &yC,
                            const std::string endCond)
{
   const size_t n = y.size();
   if (endCond != "clamped" && endCond != "natural")
   {
      printf("getSplineCoeffs() error: endCond was not \"clamped\" or \"natural\".");
      return -1;
   }
   if (n != yC.c0.size())
   {
      yC.c0.resize(n);
      yC.c1.resize(n);
      yC.c2.resize(n);
      yC.c3.resize(n);
   }
   std::vector<double> m(n); // second derivatives at the knots
   for (size_t i = 1; i + 1 < n; ++i) m[i] = 6 * (y[i - 1] - 2 * y[i] + y[i + 1]);

   if (endCond == "clamped") secondDerivsClamped(m);
   else secondDerivsNatural(m);

   // the row of the last knot is deliberately left untouched, as in the reference
   for (size_t i = 0; i + 1 < n; ++i)
   {
      yC.c3[i] = (m[i + 1] - m[i]) / 6.0;
      yC.c2[i] = m[i] / 2.0;
      yC.c1[i] = y[i + 1] - y[i] - (m[i + 1] + 2 * m[i]) / 6.0;
      yC.c0[i] = y[i];
   }
   return 0;
}

// reference spline.cpp:56-99
int Spline::findInterpSegs(const std::vector<double> &aIn, const std::vector<double> &aOut,
                           splineSegs &mySegs)
{
   const int nIn = (int)aIn.size();
   const int nOut = (int)aOut.size();
   mySegs.seg.resize(nOut);
   mySegs.tau.resize(nOut);

   int cursor = 0;
   for (int i = 0; i < nOut; ++i)
   {
      while (!(aOut[i] < aIn[cursor + 1] || cursor == nIn - 2)) ++cursor;
      mySegs.seg[i] = cursor;
   }
   std::vector<double> width(nIn);
   for (int i = 0; i < nIn - 1; ++i)
   {
      width[i] = aIn[i + 1] - aIn[i];
      if (width[i] < 1e-20)
      {
         printf("Error: division by zero in findInterpSegs().\n");
         return -1;
      }
   }
   for (int i = 0; i < nOut; ++i)
   {
      const int s = mySegs.seg[i];
      mySegs.tau[i] = (aOut[i] - aIn[s]) / width[s];
   }
   return 0;
}

// reference spline.cpp:108-120
int Spline::interp1linear(std::vector<double> &b, const splineSegs &mySegs)
{
   const int nOut = (int)mySegs.seg.size();
   std::vector<double> out(nOut);
   for (int i = 0; i < nOut; ++i)
   {
      const int s = mySegs.seg[i];
      out[i] = b[s] + (b[s + 1] - b[s]) * mySegs.tau[i];
   }
   b = out;
   return 0;
}

// reference spline.cpp:129-155
int Spline::interp1spline(std::vector<double> &b, std::vector<double> &bD,
                          std::vector<double> &bD2, splineCoeffs &bC, const splineSegs &mySegs,
                          const double tfact)
{
   const int nOut = (int)mySegs.seg.size();
   b.resize(nOut);
   bD.resize(nOut);
   bD2.resize(nOut);
   const double vfact = 1.0 / tfact;
   const double afact = vfact * vfact;
   for (int i = 0; i < nOut; ++i)
   {
      const int j = mySegs.seg[i];
      const double tau = mySegs.tau[i];
      const double tau2 = tau * tau, tau3 = tau2 * tau;
      const double c3 = bC.c3[j], c2 = bC.c2[j], c1 = bC.c1[j], c0 = bC.c0[j];
      b[i] = c3 * tau3 + c2 * tau2 + c1 * tau + c0;
      bD[i] = (3 * c3 * tau2 + 2 * c2 * tau + c1) * vfact;
      bD2[i] = (6 * c3 * tau + 2 * c2) * afact;
   }
   return 0;
}
