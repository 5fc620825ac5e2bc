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

// ---- moving-window filters (reference util.cpp:257-338) -----------------------------------------
// Both filters of the reference look at sample i through a window that is centred on i, `half` samples wide on either side in
// the interior and shrinking to what fits next to an end; the two end samples pass through.  What makes the result
// reproducible to the bit is the ORDER in which a window is folded: windows that touch the front of the series, and interior
// windows, are folded front to back; windows that touch the back are folded from the last sample backwards.  Here that is one
// description (Window) and one traversal (foldWindows); the filters only differ in the fold.  The device kernels
// k_rs_smooth / k_out_down and the oracle state the same rule per sample.
namespace
{
struct Window
{
   int first; // first sample folded
   int count; // samples in the window (always odd)
   int step;  // +1 front to back, -1 back to front
};

// effective half-width of a filter of nominal width w on n samples: the largest odd window not wider than min(w, n)
int halfWidth(int w, int n)
{
   const int width = std::min(w, n);
   return width / 2 + width % 2 - 1;
}

Window windowAt(int i, int n, int half)
{
   if (i < half) return Window{0, 2 * i + 1, +1};
   if (i >= n - half) return Window{n - 1, 2 * (n - 1 - i) + 1, -1};
   return Window{i - half, 2 * half + 1, +1};
}

// out[i] = finish(fold over the window of i, number of samples folded).  The two ends are copied -- except by a filter of
// width one (half == 0), whose single-sample windows the reference also runs over the ends (util.cpp:281-287)
template <class Fold, class Finish>
std::vector<double> foldWindows(const std::vector<double> &in, int half, Fold fold, Finish finish)
{
   const int n = (int)in.size();
   std::vector<double> out(in);
   const int skip = half > 0 ? 1 : 0;
   for (int i = skip; i + skip < n; ++i)
   {
      const Window win = windowAt(i, n, half);
      double acc = in[win.first];
      for (int k = 1, at = win.first + win.step; k < win.count; ++k, at += win.step) acc = fold(acc, in[at]);
      out[i] = finish(acc, win.count);
   }
   return out;
}

double addTo(double acc, double v) { return acc + v; }
double lesserOf(double acc, double v) { return std::min(acc, v); }
double meanOf(double sum, int count) { return sum / count; }
double asIs(double v, int) { return v; }
} // namespace

// centred moving average (reference util.cpp:257-290)
int smooth(std::vector<double> &x, int w)
{
   // the first term of a window sum is 0 + x in the reference; adding to +0.0 changes no bit except -0.0 -> +0.0, which
   // the explicit zero start below keeps
   x = foldWindows(x, halfWidth(w, (int)x.size()), addTo, [](double sum, int count) { return meanOf(0.0 + sum, count); });
   return 0;
}

// moving minimum, smoothed, and never above the input (reference util.cpp:299-338)
int minsmooth(std::vector<double> &x, int w)
{
   const int half = halfWidth(w, (int)x.size());
   std::vector<double> floorLine = foldWindows(x, half, lesserOf, asIs);
   smooth(floorLine, 2 * half + 1);
   std::transform(x.begin(), x.end(), floorLine.begin(), x.begin(), lesserOf);
   return 0;
}

// every w-th sample, and always the final one (reference util.cpp:347-356)
int decimate(std::vector<double> &x, int w)
{
   const size_t last = x.size() - 1, kept = last / (size_t)w + 1;
   std::vector<double> thin;
   thin.reserve(kept);
   for (size_t at = 0; thin.size() + 1 < kept; at += (size_t)w) thin.push_back(x[at]);
   thin.push_back(x[last]);
   x.swap(thin);
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

// ---- close-point removal (reference util.cpp:452-524) --------------------------------------------
// A sample goes when it lies within the threshold of its predecessor and that predecessor stays; the final sample always
// stays -- when it was marked, the sample before it goes instead and the one before that is kept.  Passes repeat on the
// thinned series until one removes nothing.  A pass is stated as the list of surviving indices, which is then applied to
// every row of both channel sets.
namespace
{
typedef std::vector<std::vector<double>> Rows;

std::vector<int> survivorsOfPass(const Rows &driving, int n, double threshSq)
{
   std::vector<int> stay(1, 0);
   stay.reserve((size_t)n);
   bool previousGone = false;
   for (int i = 1; i < n; ++i)
   {
      double distSq = 0;
      for (size_t r = 0; r < driving.size(); ++r)
      {
         const double step = driving[r][i] - driving[r][i - 1];
         distSq += step * step;
      }
      const bool gone = !previousGone && distSq < threshSq;
      if (!gone) stay.push_back(i);
      previousGone = gone;
   }
   if (previousGone && n > 2)
   {
      stay.pop_back();                              // n-2 (it stayed, or n-1 could not have been marked)
      if (stay.back() != n - 3) stay.push_back(n - 3);
      stay.push_back(n - 1);
   }
   return stay;
}

void keepOnly(Rows &rows, const std::vector<int> &stay, size_t nBefore)
{
   for (size_t r = 0; r < rows.size(); ++r)
   {
      if (rows[r].size() < nBefore) continue;
      for (size_t k = 0; k < stay.size(); ++k) rows[r][k] = rows[r][(size_t)stay[k]];
      rows[r].resize(stay.size());
   }
}
} // namespace

int remClosePts(std::vector<std::vector<double>> &x, std::vector<std::vector<double>> &y,
                double xThresh)
{
   printf("remClosePts():  ||dtheta||_min=%f deg imposed. ", xThresh);
   const int nStart = (int)x[0].size();
   int n = nStart;
   for (;;)
   {
      const std::vector<int> stay = survivorsOfPass(x, n, xThresh * xThresh);
      if ((int)stay.size() == n) break;
      keepOnly(x, stay, (size_t)n);
      keepOnly(y, stay, (size_t)n);
      n = (int)stay.size();
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
