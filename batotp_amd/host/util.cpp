// util.cpp -- host helpers of the BA drop-in library.
//
// Behavioural restatement (from scratch, no Eigen) of reference batotp/util.cpp; each function
// names the reference lines it follows.  Host pre/post-processing only -- nothing here is on the
// GPU hot path.
#include "util.h"

#include <time.h>
#include <unistd.h>

#include <limits>

// ---------------------------------------------------------------------------------------------
// timing (reference util.cpp:52-88)
// ---------------------------------------------------------------------------------------------
Time getTime(void)
{
   struct timespec now;
   clock_gettime(CLOCK_REALTIME, &now);
   Time t;
   t.a = (int64_t)now.tv_sec;
   t.b = (int64_t)now.tv_nsec;
   return t;
}

double diffTime(const Time &endTime, const Time &startTime)
{
   const double whole = (double)(endTime.a - startTime.a);
   if (endTime.b < startTime.b)
   {
      return whole - 1 + (double)(endTime.b - startTime.b + 1.0e9) / 1.0e9;
   }
   return whole + (double)(endTime.b - startTime.b) / 1.0e9;
}

// ---------------------------------------------------------------------------------------------
// file helpers (reference util.cpp:98-235)
// ---------------------------------------------------------------------------------------------
int NextLine(FILE *fid)
{
   int ch = 0;
   do
   {
      ch = fgetc(fid);
   } while (ch != '\n' && ch != EOF);
   return ch;
}

int doesFileExist(const char *fname)
{
   return (access(fname, F_OK) != -1) ? 0 : -1;
}

int mkDirIfNec(const char *dirname)
{
   return (mkdir(dirname, S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH) == -1) ? -1 : 0;
}

std::string readChar(FILE *fid, int &readCountTotal)
{
   char word[80];
   word[0] = '\0';
   readCountTotal += fscanf(fid, "%79s", word);
   NextLine(fid);
   return std::string(word);
}

int readInt(FILE *fid, int &readCountTotal)
{
   int v = 0;
   readCountTotal += fscanf(fid, "%d", &v);
   NextLine(fid);
   return v;
}

double readDouble(FILE *fid, int &readCountTotal)
{
   double v = 0;
   readCountTotal += fscanf(fid, "%lf", &v);
   NextLine(fid);
   return v;
}

std::vector<double> readDoubleVector(FILE *fid, int &readCountTotal, const int vectorLen)
{
   std::vector<double> v(vectorLen);
   for (int k = 0; k < vectorLen; ++k)
   {
      readCountTotal += fscanf(fid, "%lf", &v[k]);
   }
   NextLine(fid);
   return v;
}

bool readBool(FILE *fid, int &readCountTotal)
{
   return readInt(fid, readCountTotal) == 1;
}

// ---------------------------------------------------------------------------------------------
// vector helpers
// ---------------------------------------------------------------------------------------------

// reference util.cpp:244-248
int normalizeArcLength(std::vector<double> &s)
{
   const double scale = 1.0 / s[s.size() - 1];
   for (size_t k = 0; k < s.size(); ++k) s[k] = scale * s[k];
   return 0;
}

// Centred moving average with shrinking windows at both ends (reference util.cpp:257-290).
int smooth(std::vector<double> &x, int w)
{
   const int n = (int)x.size();
   w = std::min(w, n);
   const int half = w / 2 + w % 2 - 1;
   w = 2 * half + 1;

   std::vector<double> y(n);
   y[0] = x[0];
   y[n - 1] = x[n - 1];

   for (int i = 1; i < half; ++i)
   {
      const int span = 2 * i + 1;
      double head = 0, tail = 0;
      for (int j = 0; j < span; ++j)
      {
         head += x[j];
         tail += x[n - j - 1];
      }
      y[i] = head / span;
      y[n - i - 1] = tail / span;
   }
   for (int i = half; i < n - half; ++i)
   {
      double acc = 0;
      for (int j = i - half; j < i + half + 1; ++j) acc += x[j];
      y[i] = acc / w;
   }
   x = y;
   return 0;
}

// Moving-window minimum followed by smooth(), then pointwise min with the input
// (reference util.cpp:299-338).
int minsmooth(std::vector<double> &x, int w)
{
   const int n = (int)x.size();
   w = std::min(w, n);
   const int half = w / 2 + w % 2 - 1;
   w = 2 * half + 1;

   std::vector<double> y(n);
   y[0] = x[0];
   y[n - 1] = x[n - 1];

   for (int i = 1; i < half; ++i)
   {
      const int span = 2 * i + 1;
      double head = x[0], tail = x[n - 1];
      for (int j = 1; j < span; ++j)
      {
         head = std::min(head, x[j]);
         tail = std::min(tail, x[n - j - 1]);
      }
      y[i] = head;
      y[n - i - 1] = tail;
   }
   for (int i = half; i < n - half; ++i)
   {
      double m = x[i - half];
      for (int j = i - half + 1; j < i + half + 1; ++j) m = std::min(m, x[j]);
      y[i] = m;
   }
   smooth(y, w);
   for (int i = 0; i < n; ++i) x[i] = std::min(x[i], y[i]);
   return 0;
}

// keep every w-th sample, always keeping the last one (reference util.cpp:347-356)
int decimate(std::vector<double> &x, int w)
{
   const int nIn = (int)x.size();
   const int nOut = (nIn - 1) / w + 1;
   for (int i = 0; i < nOut; ++i) x[i] = x[w * i];
   if (w * (nOut - 1) + 1 != nIn) x[nOut - 1] = x[nIn - 1];
   x.resize(nOut);
   return 0;
}

// roots of A x^2 + B x + C; -1 complex, -2 degenerate (reference util.cpp:361-383)
int solveQuadratic(const double A, const double B, const double C, double &sol1, double &sol2)
{
   if (std::abs(A) < 1e-308)
   {
      if (std::abs(B) < 1e-308) return -2;
      sol1 = -C / B;
      sol2 = sol1;
      return 0;
   }
   const double rad = B * B - 4 * A * C;
   if (rad < 0) return -1;
   const double den = 2 * A;
   const double F1 = -B / den;
   const double F2 = std::sqrt(rad) / den;
   sol1 = F1 + F2;
   sol2 = F1 - F2;
   return 0;
}

// reference util.cpp:392-404
double findMedian(std::vector<double> &x)
{
   std::vector<double> sorted(x);
   std::sort(sorted.begin(), sorted.end());
   const size_t n = sorted.size();
   const size_t mid = n / 2;
   if (n % 2 == 1) return sorted[mid];
   return .5 * (sorted[mid - 1] + sorted[mid]);
}

// The isSVD = 1 branch of solveLinSys (reference util.cpp:421-438): Eigen's JacobiSVD<MatrixXd>(A, ComputeThinU | ComputeThinV)
// and its solve(), written out for a square real matrix (Eigen 3.3: JacobiSVD.h, Jacobi.h, SVDBase.h).  Work matrix = A scaled by
// its largest entry; sweeps over the index pairs q < p, each off-diagonal pair above max(DBL_MIN, 2 eps max|diag|) removed by a
// left and a right plane rotation that are accumulated into U and V; singular values = |diagonal| x scale in descending order;
// x = V diag(1/sigma) U^T b over the numerical rank.  The three-term sums of the last step are added left to right, which is what
// reproduces the reference binary's isSVD = 1 outputs (tests/golden/CSPR3DOF_svd, CSPR3DOF_par_svd).
namespace
{
struct PlaneRot { double c, s; };

// x <- c x + s y,  y <- -s x + c y  over two strided vectors (Eigen: apply_rotation_in_the_plane)
void rotatePair(double *x, int strideX, double *y, int strideY, int count, const PlaneRot &g)
{
   if (g.c == 1.0 && g.s == 0.0) return;
   for (int k = 0; k < count; ++k)
   {
      const double a = x[k * strideX], b = y[k * strideY];
      x[k * strideX] = g.c * a + g.s * b;
      y[k * strideY] = -g.s * a + g.c * b;
   }
}

// the rotation that diagonalises the symmetric 2x2 block [x y; y z] (Eigen: JacobiRotation::makeJacobi)
PlaneRot symmetricJacobi(double x, double y, double z)
{
   PlaneRot g = {1.0, 0.0};
   const double twice = 2.0 * std::abs(y);
   if (twice < std::numeric_limits<double>::min()) return g;
   const double tau = (x - z) / twice;
   const double w = std::sqrt(tau * tau + 1.0);
   const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
   const double sgn = t > 0.0 ? 1.0 : -1.0;
   const double n = 1.0 / std::sqrt(t * t + 1.0);
   g.s = -sgn * (y / std::abs(y)) * std::abs(t) * n;
   g.c = n;
   return g;
}

bool jacobiSvdSolve(const std::vector<std::vector<double>> &Av, const std::vector<double> &bv, std::vector<double> &xv)
{
   const int n = (int)bv.size();
   const double tiny = std::numeric_limits<double>::min(), eps = std::numeric_limits<double>::epsilon();
   std::vector<double> W((size_t)n * n), U((size_t)n * n, 0.0), V((size_t)n * n, 0.0), sigma(n);
   double scale = 0.0;
   for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) scale = std::max(scale, std::abs(Av[i][j]));
   if (scale == 0.0) scale = 1.0;
   double diagMax = 0.0;
   for (int i = 0; i < n; ++i)
   {
      for (int j = 0; j < n; ++j) W[(size_t)i * n + j] = Av[i][j] / scale;
      U[(size_t)i * n + i] = 1.0;
      V[(size_t)i * n + i] = 1.0;
      diagMax = std::max(diagMax, std::abs(W[(size_t)i * n + i]));
   }
   for (bool clean = false; !clean;)
   {
      clean = true;
      for (int p = 1; p < n; ++p)
         for (int q = 0; q < p; ++q)
         {
            const double limit = std::max(tiny, 2.0 * eps * diagMax);
            if (!(std::abs(W[(size_t)p * n + q]) > limit || std::abs(W[(size_t)q * n + p]) > limit)) continue;
            clean = false;
            // 2x2 block (p,p) (p,q) / (q,p) (q,q): first a rotation that makes it symmetric, then the symmetric Jacobi rotation
            double b00 = W[(size_t)p * n + p], b01 = W[(size_t)p * n + q], b10 = W[(size_t)q * n + p], b11 = W[(size_t)q * n + q];
            PlaneRot sym = {1.0, 0.0};
            const double trace = b00 + b11, skew = b10 - b01;
            if (!(std::abs(skew) < tiny))
            {
               const double u = trace / skew;
               const double h = std::sqrt(1.0 + u * u);
               sym.s = 1.0 / h;
               sym.c = u / h;
            }
            {
               double row0[2] = {b00, b01}, row1[2] = {b10, b11};
               rotatePair(row0, 1, row1, 1, 2, sym);
               b00 = row0[0]; b01 = row0[1]; b11 = row1[1];
            }
            const PlaneRot right = symmetricJacobi(b00, b01, b11);
            const PlaneRot rightT = {right.c, -right.s};
            const PlaneRot left = {sym.c * rightT.c - sym.s * rightT.s, sym.c * rightT.s + sym.s * rightT.c};
            rotatePair(&W[(size_t)p * n], 1, &W[(size_t)q * n], 1, n, left);          // rows p, q of W
            rotatePair(&U[p], n, &U[q], n, n, left);                                    // columns p, q of U
            rotatePair(&W[p], n, &W[q], n, n, rightT);                                  // columns p, q of W
            rotatePair(&V[p], n, &V[q], n, n, rightT);                                  // columns p, q of V
            diagMax = std::max(diagMax, std::max(std::abs(W[(size_t)p * n + p]), std::abs(W[(size_t)q * n + q])));
         }
   }
   for (int i = 0; i < n; ++i)
   {
      const double d = W[(size_t)i * n + i];
      sigma[i] = std::abs(d);
      if (d < 0.0)
         for (int r = 0; r < n; ++r) U[(size_t)r * n + i] = -U[(size_t)r * n + i];
   }
   for (int i = 0; i < n; ++i) sigma[i] *= scale;
   int nonzero = n;
   for (int i = 0; i < n; ++i)
   {
      int at = i;
      for (int k = i + 1; k < n; ++k)
         if (sigma[k] > sigma[at]) at = k;
      if (sigma[at] == 0.0) { nonzero = i; break; }
      if (at != i)
      {
         std::swap(sigma[i], sigma[at]);
         for (int r = 0; r < n; ++r)
         {
            std::swap(U[(size_t)r * n + i], U[(size_t)r * n + at]);
            std::swap(V[(size_t)r * n + i], V[(size_t)r * n + at]);
         }
      }
   }
   // reference util.cpp:424-426
   if (sigma[0] / sigma[n - 1] < 100.0 * eps) return true;
   int rank = nonzero;
   {
      const double cut = std::max(sigma[0] * ((double)n * eps), tiny);
      while (rank > 0 && sigma[rank - 1] < cut) --rank;
   }
   std::vector<double> y(rank);
   for (int k = 0; k < rank; ++k)
   {
      double acc = U[k] * bv[0];
      for (int r = 1; r < n; ++r) acc += U[(size_t)r * n + k] * bv[r];
      y[k] = (1.0 / sigma[k]) * acc;
   }
   for (int r = 0; r < n; ++r)
   {
      double acc = 0.0;
      if (rank > 0)
      {
         acc = V[(size_t)r * n] * y[0];
         for (int k = 1; k < rank; ++k) acc += V[(size_t)r * n + k] * y[k];
      }
      xv[r] = acc;
   }
   return false;
}
} // namespace

// Dense solve A x = b (reference util.cpp:413-442 calls Eigen's PartialPivLU, or its JacobiSVD when isSVD is set).  Same elimination order as Eigen's unblocked LU:
// first-max partial pivoting, column scaling by true division, rank-1 trailing update, then a
// column-oriented unit-lower / upper substitution.  Returns isIllCond (always false here).
bool solveLinSys(const std::vector<std::vector<double>> &Av, const std::vector<double> &bv,
                 std::vector<double> &xv, const bool isSVD)
{
   if (isSVD) return jacobiSvdSolve(Av, bv, xv);
   const int dim = (int)bv.size();
   std::vector<double> lu((size_t)dim * dim);
   std::vector<double> rhs(bv);
   std::vector<int> swapWith(dim);
   for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j) lu[(size_t)i * dim + j] = Av[i][j];

   for (int k = 0; k < dim; ++k)
   {
      int pivotRow = k;
      double pivotMag = std::abs(lu[(size_t)k * dim + k]);
      for (int i = k + 1; i < dim; ++i)
      {
         const double mag = std::abs(lu[(size_t)i * dim + k]);
         if (mag > pivotMag)
         {
            pivotMag = mag;
            pivotRow = i;
         }
      }
      swapWith[k] = pivotRow;
      if (pivotMag != 0.0)
      {
         if (pivotRow != k)
         {
            for (int j = 0; j < dim; ++j) std::swap(lu[(size_t)k * dim + j], lu[(size_t)pivotRow * dim + j]);
         }
         for (int i = k + 1; i < dim; ++i) lu[(size_t)i * dim + k] /= lu[(size_t)k * dim + k];
      }
      for (int i = k + 1; i < dim; ++i)
         for (int j = k + 1; j < dim; ++j)
            lu[(size_t)i * dim + j] -= lu[(size_t)i * dim + k] * lu[(size_t)k * dim + j];
   }
   for (int k = 0; k < dim; ++k)
   {
      if (swapWith[k] != k) std::swap(rhs[k], rhs[swapWith[k]]);
   }
   for (int i = 0; i < dim; ++i)
   {
      if (rhs[i] != 0.0)
         for (int j = i + 1; j < dim; ++j) rhs[j] -= rhs[i] * lu[(size_t)j * dim + i];
   }
   for (int i = dim - 1; i >= 0; --i)
   {
      if (rhs[i] != 0.0)
      {
         rhs[i] /= lu[(size_t)i * dim + i];
         for (int j = 0; j < i; ++j) rhs[j] -= rhs[i] * lu[(size_t)j * dim + i];
      }
   }
   for (int i = 0; i < dim; ++i) xv[i] = rhs[i];
   return false;
}

// Iteratively drop samples closer than xThresh to their predecessor; within one pass only
// non-adjacent samples are dropped and the last sample always survives
// (reference util.cpp:452-524).  y is filtered with the same mask.
int remClosePts(std::vector<std::vector<double>> &x, std::vector<std::vector<double>> &y,
                double xThresh)
{
   printf("remClosePts():  ||dtheta||_min=%f deg imposed. ", xThresh);

   const double threshSq = xThresh * xThresh;
   const int nx = (int)x.size();
   const int ny = (int)y.size();
   int n = (int)x[0].size();
   const int nStart = n;
   std::vector<char> drop(n, 0);

   for (;;)
   {
      bool any = false;
      for (int i = 1; i < n; ++i)
      {
         double distSq = 0;
         for (int j = 0; j < nx; ++j)
         {
            const double dlt = x[j][i] - x[j][i - 1];
            distSq += dlt * dlt;
         }
         if (distSq < threshSq && !drop[i - 1])
         {
            drop[i] = 1;
            any = true;
         }
      }
      if (drop[n - 1] && n > 2)
      {
         drop[n - 1] = 0;
         drop[n - 2] = 1;
         drop[n - 3] = 0;
      }
      if (!any) break;

      int keep = 0;
      for (int i = 0; i < n; ++i)
      {
         if (drop[i]) continue;
         for (int j = 0; j < nx; ++j) x[j][keep] = x[j][i];
         for (int j = 0; j < ny; ++j) y[j][keep] = y[j][i];
         ++keep;
      }
      n = keep;
      for (int j = 0; j < nx; ++j) x[j].resize(n);
      for (int j = 0; j < ny; ++j) y[j].resize(n);
      drop.assign(n, 0);
   }
   printf(" before %d points; after %d points\n", nStart, n);
   return 0;
}

// axis-angle -> unit quaternion (reference util.cpp:534-553)
std::array<double, 4> aa2q(std::array<double, 3> aa)
{
   std::array<double, 4> q;
   const double angle = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
   if (angle < 1e-6)
   {
      q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0;
      return q;
   }
   // sine and cosine of the half angle as ONE glibc sincos(): what the reference's optimised build calls (util.cpp:547-548),
   // spelled out because sincos() is not bit-identical to separate sin() / cos() calls in this glibc
   double sinHalf, cosHalf;
   ::sincos(0.5 * angle, &sinHalf, &cosHalf);
   q[0] = cosHalf;
   for (int k = 0; k < 3; ++k) q[k + 1] = aa[k] * sinHalf / angle;
   return q;
}

// unit quaternion -> axis-angle (reference util.cpp:562-581)
std::array<double, 3> q2aa(std::array<double, 4> q)
{
   std::array<double, 3> aa;
   const double vecNorm = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
   if (vecNorm < 1e-6)
   {
      aa[0] = 0.0; aa[1] = 0.0; aa[2] = 0.0;
      return aa;
   }
   const double scale = 2.0 * std::atan2(vecNorm, q[0]) / vecNorm;
   for (int k = 0; k < 3; ++k) aa[k] = scale * q[k + 1];
   return aa;
}
