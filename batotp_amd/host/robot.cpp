// robot.cpp -- built-in robot models (host side), written without Eigen.
//
// Behavioural restatement of reference batotp/robot.cpp; each routine names the reference lines
// it follows.  Expression order is kept wherever the result feeds the sweep.
#include "robot.h"

#include <array>
#include <cassert>
#include <cmath>
#include <cstdio>

#include "batotp_models.h"
#include "config.h"

namespace BATOTP
{

namespace
{
// 3x3 products as the reference's build of Eigen evaluates them for fixed-size double matrices
// (reference robot.cpp:153), read off the instruction stream of the reference's prebuilt bin/batest:
// per destination column, rows 0 and 1 are computed with one SSE2 packet -- terms added left to
// right, (a0+a1)+a2 -- and row 2 through the scalar coefficient path, whose unrolled reduction adds
// a0+(a1+a2).  The row-times-vector products of robot.cpp:170-172 use the same scalar reduction.
// The KUKA example is sensitive to this at the ulp level.
// (Robot::kinSinCos) Trigonometry of the forward kinematics: cos and sin of one angle as ONE glibc sincos() call.  That is what the reference's
// optimised build makes of "c1=cos(t1); s1=sin(t1);" (robot.cpp:130-136) and of the two-link arm's cos(th1) / sin(th1)
// (robot.cpp:198-199), and sincos() is not bit-identical to separate cos() / sin() calls in this glibc --
// so it is spelled out instead of being left to the optimiser.  The device resampler / output stage take these values as
// tables (BATOTP_F_HOST_TRIG), the checker computes them the same way.
inline double sumSeq(double a0, double a1, double a2) { return (a0 + a1) + a2; }
inline double sumTree(double a0, double a1, double a2) { return a0 + (a1 + a2); }

struct Mat3
{
   double m[3][3];
};

inline Mat3 mul(const Mat3 &L, const Mat3 &R)
{
   Mat3 out;
   for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
      {
         const double t0 = L.m[r][0] * R.m[0][c], t1 = L.m[r][1] * R.m[1][c], t2 = L.m[r][2] * R.m[2][c];
         out.m[r][c] = (r == 2) ? sumTree(t0, t1, t2) : sumSeq(t0, t1, t2);
      }
   return out;
}
} // namespace

void __attribute__((noinline)) Robot::kinSinCos(double angle, double *sinOut, double *cosOut) { ::sincos(angle, sinOut, cosOut); }

namespace
{
double __attribute__((noinline)) plainCos(double x) { return std::cos(x); }
} // namespace

void Robot::planarRRDynTrig(double th1, double th2, double out[4])
{
   out[0] = plainCos(th1);
   kinSinCos(th2, &out[3], &out[1]);
   out[2] = plainCos(th1 + th2);
}

// reference robot.cpp:46-62
int Robot::call_set_robotType(const std::string &robotTypeStr)
{
   static const struct { const char *name; int id; } table[] = {
      {"KUKA", KUKA}, {"UR", UR}, {"RR", RR}, {"CSPR3DOF", CSPR3DOF}, {"GENJNT", GENJNT}};
   _kind = 0;
   _kindName = robotTypeStr;
   for (size_t k = 0; k < sizeof(table) / sizeof(table[0]); ++k)
   {
      if (_kindName == table[k].name) _kind = table[k].id;
   }
   return _kind;
}

// ---------------------------------------------------------------------------------------------
// forward kinematics (reference robot.cpp:73-202)
// ---------------------------------------------------------------------------------------------
int Robot::call_fwdKin(const Channels &theta, Channels &cart)
{
   if (_kind == KUKA) { kukaToolPoint(theta, cart); return 0; }
   if (_kind == RR) { planarRRToolPoint(theta, cart); return 0; }
   printf("No forward Kinematics model provided for robotType=%s.\n", _kindName.c_str());
   return -1;
}

// KUKA LWR IV+ tool point (reference robot.cpp:105-174)
void Robot::kukaToolPoint(const Channels &theta, Channels &cart) const
{
   const double tool[3] = {0, -.08, .545};
   const double a0 = .3105, a1 = .4, a2 = .39;
   const int n = (int)theta[0].size();
   cart.resize(3);
   for (int k = 0; k < 3; ++k) cart[k].resize(n);

   for (int i = 0; i < n; ++i)
   {
      double c[7], s[7];
      for (int k = 0; k < 7; ++k)
      {
         kinSinCos(_DEG2RAD * theta[k][i], &s[k], &c[k]);
      }
      const double c1 = c[0], c2 = c[1], c3 = c[2], c4 = c[3], c5 = c[4], c6 = c[5], c7 = c[6];
      const double s1 = s[0], s2 = s[1], s3 = s[2], s4 = s[3], s5 = s[4], s6 = s[5], s7 = s[6];

      const Mat3 Q12 = {{{c1 * c2, -s1, -c1 * s2}, {c2 * s1, c1, -s1 * s2}, {s2, 0, c2}}};
      const Mat3 Q34 = {{{c3 * c4, -s3, c3 * s4}, {c4 * s3, c3, s3 * s4}, {-s4, 0, c4}}};
      const Mat3 Q567 = {{{c5 * c6 * c7 - s5 * s7, -c7 * s5 - c5 * c6 * s7, -c5 * s6},
                          {c5 * s7 + c6 * c7 * s5, c5 * c7 - c6 * s5 * s7, -s5 * s6},
                          {c7 * s6, -s6 * s7, c6}}};
      const Mat3 Q1234 = mul(Q12, Q34);
      const Mat3 Q = mul(Q1234, Q567);

      // elbow, wrist, tool point
      const double x1 = a1 * Q12.m[0][2];
      const double y1 = a1 * Q12.m[1][2];
      const double z1 = a1 * Q12.m[2][2] + a0;
      const double x2 = x1 + a2 * Q1234.m[0][2];
      const double y2 = y1 + a2 * Q1234.m[1][2];
      const double z2 = z1 + a2 * Q1234.m[2][2];
      cart[0][i] = x2 + sumTree(Q.m[0][0] * tool[0], Q.m[0][1] * tool[1], Q.m[0][2] * tool[2]);
      cart[1][i] = y2 + sumTree(Q.m[1][0] * tool[0], Q.m[1][1] * tool[1], Q.m[1][2] * tool[2]);
      cart[2][i] = z2 + sumTree(Q.m[2][0] * tool[0], Q.m[2][1] * tool[1], Q.m[2][2] * tool[2]);
   }
}

// planar 2R arm; the z channel is only sized (reference robot.cpp:183-202)
void Robot::planarRRToolPoint(const Channels &theta, Channels &cart) const
{
   const double L1 = .4, L2 = .6;
   const int n = (int)theta[0].size();
   cart.resize(3);
   for (int k = 0; k < 3; ++k) cart[k].resize(n);
   for (int i = 0; i < n; ++i)
   {
      const double q1 = _DEG2RAD * theta[0][i];
      const double q2 = _DEG2RAD * theta[1][i];
      double s1, c1, s12, c12;
      kinSinCos(q1, &s1, &c1);
      kinSinCos(q1 + q2, &s12, &c12);
      cart[0][i] = L1 * c1 + L2 * c12;
      cart[1][i] = L1 * s1 + L2 * s12;
   }
}

// ---------------------------------------------------------------------------------------------
// inverse kinematics (reference robot.cpp:213-322)
// ---------------------------------------------------------------------------------------------
int Robot::call_invKin(Channels &theta, const Channels &cart)
{
   if (_kind == CSPR3DOF) { csprCableLengths(theta, cart); return 0; }
   printf("No inverse Kinematics model provided for robotType=%s.\n", _kindName.c_str());
   return -1;
}

const std::vector<std::vector<double>> &Robot::cableAnchors()
{
   if (_anchors.empty()) buildCsprAnchors();
   return _anchors;
}

// cable lengths = distance from the platform point to each anchor (reference robot.cpp:243-278)
void Robot::csprCableLengths(Channels &theta, const Channels &cart)
{
   const int dims = (int)cart.size();
   const int n = (int)cart[0].size();
   theta.resize(3);
   for (int k = 0; k < dims; ++k) theta[k].resize(n);
   const Channels &P = cableAnchors();
   for (int i = 0; i < n; ++i)
   {
      const double x = cart[0][i], y = cart[1][i], z = cart[2][i];
      for (int cable = 0; cable < 3; ++cable)
      {
         const double dx = x - P[0][cable], dy = y - P[1][cable], dz = z - P[2][cable];
         double sq = 0.0;
         sq += dx * dx;
         sq += dy * dy;
         sq += dz * dz;
         theta[cable][i] = std::sqrt(sq);
      }
   }
}

// anchor points of the Laval 3-DOF cable robot, centred on their centroid
// (reference robot.cpp:291-322)
void Robot::buildCsprAnchors()
{
   const double target1[3] = {1.0941, -4.9074, 2.5542};
   const double shift1[3] = {-0.765, 0.112, 3.74};
   const double target3[3] = {0.2098, 5.3409, 2.6236};
   const double shift2[3] = {0.43, 0.125, 3.615};
   const double p3[3] = {-5.9751, 0.1399, 6.1543};
   double p1[3], p2[3];
   for (int k = 0; k < 3; ++k)
   {
      p1[k] = target1[k] + shift1[k];
      p2[k] = target3[k] + shift2[k];
   }
   const int axis[3] = {1, 0, 2};
   _anchors.assign(3, std::vector<double>(3));
   for (int r = 0; r < 3; ++r)
   {
      _anchors[r][0] = -p1[axis[r]];
      _anchors[r][1] = -p2[axis[r]];
      _anchors[r][2] = -p3[axis[r]];
   }
   for (int r = 0; r < 3; ++r)
   {
      const double centre = 1 / 3.0 * (_anchors[r][0] + _anchors[r][1] + _anchors[r][2]);
      for (int c = 0; c < 3; ++c) _anchors[r][c] -= centre;
   }
}

// ---------------------------------------------------------------------------------------------
// dynamics: tau (or A tau) = a1 sddot + a2 sdot^2 + a3 sdot + a4
// ---------------------------------------------------------------------------------------------
int Robot::call_dynSerial(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                          const Channels &theta, const Channels &thetaD, const Channels &thetaD2)
{
   if (_kind == RR && !_hasSerial) { planarRRDynamics(a1, a2, a3, a4, theta, thetaD, thetaD2); return 0; }
   if (const batotp_serial_model *m = serialModel())
   {
      if ((size_t)m->n_links != theta.size())
      {
         printf("Serial-chain model has %d links but the path has %d joints.\n", (int)m->n_links, (int)theta.size());
         return -1;
      }
      serialChainDynamics(*m, a1, a2, a3, a4, theta, thetaD, thetaD2);
      return 0;
   }
   printf("No dynamics model provided for serial robotType=%s.\n", _kindName.c_str());
   return -1;
}

const batotp_serial_model *Robot::serialModel()
{
   if (!_hasSerial && !_serialProbed)
   {
      _serialProbed = true;
      // the closed-form two-link arm stays the RR model (bit parity with the reference); other robot types take
      // the table of include/batotp_models.h if there is one
      if (_kind != RR && batotp_builtin_serial_model(_kind, &_serial) == 0) _hasSerial = true;
   }
   return _hasSerial ? &_serial : nullptr;
}

// ---------------------------------------------------------------------------------------------
// Table-driven serial chain (BASELINE config 3).  a1 = M q', a2 = M q'' + C(q, q') q', a3 = fv q',
// a4 = g(q) by three passes of the recursive Newton-Euler algorithm in link coordinates (zero-aligned
// frames, Rodrigues rotation about each joint axis).  Host twin of the device kernel k_dyn_serial
// (same expression order), used by the torque recomputation of interpOutputData (reference ba.cpp:1791-1827).
// ---------------------------------------------------------------------------------------------
namespace
{
inline void crossV(const double *a, const double *b, double *o)
{
   o[0] = a[1] * b[2] - a[2] * b[1];
   o[1] = a[2] * b[0] - a[0] * b[2];
   o[2] = a[0] * b[1] - a[1] * b[0];
}
inline void axisRotation(const double *a, double c, double s, double *R)
{
   const double omc = 1.0 - c;
   R[0] = omc * a[0] * a[0] + c;
   R[1] = omc * a[0] * a[1] - s * a[2];
   R[2] = omc * a[0] * a[2] + s * a[1];
   R[3] = omc * a[1] * a[0] + s * a[2];
   R[4] = omc * a[1] * a[1] + c;
   R[5] = omc * a[1] * a[2] - s * a[0];
   R[6] = omc * a[2] * a[0] - s * a[1];
   R[7] = omc * a[2] * a[1] + s * a[0];
   R[8] = omc * a[2] * a[2] + c;
}
inline void rotate(const double *R, const double *v, double *o)
{
   o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
   o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
   o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
inline void rotateBack(const double *R, const double *v, double *o)
{
   o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
   o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
   o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}
inline void inertiaTimes(const double *I, const double *v, double *o)
{
   o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
   o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
   o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
}

// separate libm calls (see ba_device.cpp: a fused sincos() is not bit-identical to cos() / sin())
double __attribute__((noinline)) libmCos(double x) { return std::cos(x); }
double __attribute__((noinline)) libmSin(double x) { return std::sin(x); }

void newtonEulerPass(const batotp_serial_model &m, const double *cq, const double *sq, const double *qd, const double *qdd,
                     const double *a0, double *tau)
{
   const int n = m.n_links;
   double F[BATOTP_MAX_LINKS][3], Nn[BATOTP_MAX_LINKS][3];
   double w[3] = {0, 0, 0}, wd[3] = {0, 0, 0}, a[3] = {a0[0], a0[1], a0[2]};
   for (int i = 0; i < n; ++i)
   {
      const batotp_serial_link &L = m.link[i];
      double R[9], t1[3], t2[3], t3[3], wp[3], wdp[3], ap[3], zq[3], ac[3], Iw[3], Iwd[3];
      axisRotation(L.axis, cq[i], sq[i], R);
      crossV(wd, L.off, t1);
      crossV(w, L.off, t2);
      crossV(w, t2, t3);
      for (int k = 0; k < 3; ++k) t1[k] = a[k] + t1[k] + t3[k];
      rotateBack(R, t1, ap);
      rotateBack(R, w, wp);
      rotateBack(R, wd, wdp);
      for (int k = 0; k < 3; ++k) zq[k] = L.axis[k] * qd[i];
      crossV(wp, zq, t2);
      for (int k = 0; k < 3; ++k)
      {
         w[k] = wp[k] + zq[k];
         wd[k] = wdp[k] + L.axis[k] * qdd[i] + t2[k];
         a[k] = ap[k];
      }
      crossV(wd, L.com, t1);
      crossV(w, L.com, t2);
      crossV(w, t2, t3);
      for (int k = 0; k < 3; ++k) ac[k] = a[k] + t1[k] + t3[k];
      for (int k = 0; k < 3; ++k) F[i][k] = L.mass * ac[k];
      inertiaTimes(L.inertia, wd, Iwd);
      inertiaTimes(L.inertia, w, Iw);
      crossV(w, Iw, t1);
      for (int k = 0; k < 3; ++k) Nn[i][k] = Iwd[k] + t1[k];
   }
   double fc[3] = {0, 0, 0}, nc[3] = {0, 0, 0};
   for (int i = n - 1; i >= 0; --i)
   {
      const batotp_serial_link &L = m.link[i];
      double f[3], nn[3], t1[3], t2[3] = {0, 0, 0}, R[9];
      crossV(L.com, F[i], t1);
      if (i + 1 < n) crossV(m.link[i + 1].off, fc, t2);
      for (int k = 0; k < 3; ++k)
      {
         f[k] = F[i][k] + fc[k];
         nn[k] = Nn[i][k] + nc[k] + t1[k] + t2[k];
      }
      tau[i] = L.axis[0] * nn[0] + L.axis[1] * nn[1] + L.axis[2] * nn[2];
      axisRotation(L.axis, cq[i], sq[i], R);
      rotate(R, f, fc);
      rotate(R, nn, nc);
   }
}
} // namespace

void Robot::serialChainDynamics(const batotp_serial_model &m, Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                                const Channels &theta, const Channels &thetaD, const Channels &thetaD2) const
{
   const int nl = m.n_links;
   const int n = (int)theta[0].size();
   const double unit = m.degrees ? _DEG2RAD : 1.0;
   for (int j = 0; j < nl; ++j)
   {
      a1[j].resize(n); a2[j].resize(n); a3[j].resize(n); a4[j].resize(n);
   }
   const double zero3[3] = {0, 0, 0};
   const double g0[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
   for (int i = 0; i < n; ++i)
   {
      double cq[BATOTP_MAX_LINKS], sq[BATOTP_MAX_LINKS], q1[BATOTP_MAX_LINKS], q2[BATOTP_MAX_LINKS], z[BATOTP_MAX_LINKS];
      double t1[BATOTP_MAX_LINKS], t2[BATOTP_MAX_LINKS], t4[BATOTP_MAX_LINKS];
      for (int j = 0; j < nl; ++j)
      {
         const double q = unit * theta[j][i];
         q1[j] = unit * thetaD[j][i];
         q2[j] = unit * thetaD2[j][i];
         z[j] = 0.0;
         cq[j] = libmCos(q);
         sq[j] = libmSin(q);
      }
      newtonEulerPass(m, cq, sq, z, q1, zero3, t1);
      newtonEulerPass(m, cq, sq, q1, q2, zero3, t2);
      newtonEulerPass(m, cq, sq, z, z, g0, t4);
      for (int j = 0; j < nl; ++j)
      {
         a1[j][i] = t1[j];
         a2[j][i] = t2[j];
         a3[j][i] = m.link[j].fv * q1[j];
         a4[j][i] = t4[j];
      }
   }
}

// point-mass 2R arm, joint values in degrees (reference robot.cpp:377-431)
void Robot::planarRRDynamics(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                             const Channels &theta, const Channels &thetaD,
                             const Channels &thetaD2) const
{
   const int n = (int)theta[0].size();
   double A1 = .4, A2 = .6, m1 = 4, m2 = 8;
   for (int k = 0; k < 2; ++k)
   {
      a1[k].resize(n); a2[k].resize(n); a3[k].resize(n); a4[k].resize(n);
   }
   for (int i = 0; i < n; ++i)
   {
      const double th1 = _DEG2RAD * theta[0][i];
      const double th2 = _DEG2RAD * theta[1][i];
      const double dth1 = _DEG2RAD * thetaD[0][i];
      const double dth2 = _DEG2RAD * thetaD[1][i];
      const double ddth1 = _DEG2RAD * thetaD2[0][i];
      const double ddth2 = _DEG2RAD * thetaD2[1][i];

      double tr[4];
      planarRRDynTrig(th1, th2, tr);
      const double c1 = tr[0], c2 = tr[1], c12 = tr[2], s2 = tr[3];

      const double A11 = .25 * m1 * A1 * A1 + m2 * (A1 * A1 + .25 * A2 * A2 + A1 * A2 * c2);
      const double A12 = .5 * m2 * (.5 * A2 * A2 + A1 * A2 * c2);
      const double A22 = .25 * m2 * A2 * A2;

      a1[0][i] = A11 * dth1 + A12 * dth2;
      a1[1][i] = A12 * dth1 + A22 * dth2;

      const double ccFact = m2 * A1 * A2 * s2;
      a2[0][i] = A11 * ddth1 + A12 * ddth2 - ccFact * dth2 * (dth1 + .5 * dth2);
      a2[1][i] = A12 * ddth1 + A22 * ddth2 - .5 * ccFact * dth1 * dth1;

      a3[0][i] = 10 * dth1;
      a3[1][i] = 10 * dth2;

      a4[0][i] = .5 * _g * (m1 * A1 * c1 + m2 * (2.0 * A1 * c1 + A2 * c12));
      a4[1][i] = .5 * _g * m2 * A2 * c12;
   }
}

int Robot::call_dynParallel(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                            const Channels &cart, const Channels &cartD, const Channels &cartD2)
{
   (void)cart;
   if (_kind == CSPR3DOF) { csprDynamics(a1, a2, a3, a4, cartD, cartD2); return 0; }
   printf("No dynamics model provided for parallel robotType=%s.\n", _kindName.c_str());
   return -1;
}

// point-mass platform: a1 = -p', a2 = -p'', a4 = (0,0,g) (reference robot.cpp:487-517)
void Robot::csprDynamics(Channels &a1, Channels &a2, Channels &a3, Channels &a4,
                         const Channels &cartD, const Channels &cartD2) const
{
   const int n = (int)cartD[0].size();
   for (int k = 0; k < 3; ++k)
   {
      a1[k].resize(n);
      a2[k].resize(n);
      a3[k].assign(n, 0.0);
      a4[k].assign(n, 0.0);
   }
   for (int i = 0; i < n; ++i)
   {
      for (int k = 0; k < 3; ++k)
      {
         a1[k][i] = -cartD[k][i];
         a2[k][i] = -cartD2[k][i];
      }
      a4[2][i] = _g;
   }
}

// wrench matrix of the cable robot: unit cable directions (reference robot.cpp:534-558)
int Robot::call_setA(const std::vector<double> &theta, const std::vector<double> &cart, Channels &A)
{
   if (_kind != CSPR3DOF)
   {
      printf("isParallel=True and no code was provided to find the A matrix");
      printf("for robotType=%s.\n", _kindName.c_str());
      return -1;
   }
   const Channels &P = cableAnchors();
   for (int r = 0; r < 3; ++r)
      for (int cable = 0; cable < 3; ++cable) A[r][cable] = (cart[r] - P[r][cable]) / theta[cable];
   return 0;
}

} // namespace BATOTP
