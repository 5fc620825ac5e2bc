// util.h -- host helpers of the BA drop-in library (timing, config-file scanners, path filters).
//
// Interface parity with reference batotp/util.h:46-96 (same free-function names, argument meaning
// and return conventions) so that test/main.cpp and other users of the reference headers compile
// unchanged.  None of this is on the GPU hot path; it serves BA::interpInputData /
// interpOutputData / file IO on the host.
#ifndef BATOTP_AMD_UTIL_H
#define BATOTP_AMD_UTIL_H

#include <sys/stat.h>
#include <stdint.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

// wall-clock stamp: a = seconds, b = nanoseconds (reference util.h:46-50)
struct Time
{
   int64_t a;
   int64_t b;
};

Time getTime(void);
double diffTime(const Time &endTime, const Time &startTime);

// file helpers
int NextLine(FILE *fid);
int doesFileExist(const char *fname);
int mkDirIfNec(const char *dirname);
std::string readChar(FILE *fid, int &readCountTotal);
int readInt(FILE *fid, int &readCountTotal);
double readDouble(FILE *fid, int &readCountTotal);
std::vector<double> readDoubleVector(FILE *fid, int &readCountTotal, const int vectorLen);
bool readBool(FILE *fid, int &readCountTotal);

// vector helpers
int normalizeArcLength(std::vector<double> &s);
int smooth(std::vector<double> &x, int w);
int minsmooth(std::vector<double> &x, int w);
int decimate(std::vector<double> &x, int w);
int solveQuadratic(const double A, const double B, const double C, double &sol1, double &sol2);
double findMedian(std::vector<double> &x);
bool solveLinSys(const std::vector<std::vector<double>> &Av, const std::vector<double> &bv,
                 std::vector<double> &xv, const bool isSVD);
int remClosePts(std::vector<std::vector<double>> &x, std::vector<std::vector<double>> &y,
                double xThresh);
std::array<double, 4> aa2q(std::array<double, 3> aa);
std::array<double, 3> q2aa(std::array<double, 4> q);

// sign of a value, sgn(0) = 0 (reference util.h:94-96)
template <typename T>
int sgn(T val)
{
   return (T(0) < val) - (val < T(0));
}

// Euclidean norm of a fixed-size array (reference util.h:141-148)
template <typename T, size_t N>
double norm(const std::array<T, N> &a)
{
   double acc = 0.0;
   for (size_t k = 0; k < N; ++k) acc += a[k] * a[k];
   return std::sqrt(acc);
}

#endif // BATOTP_AMD_UTIL_H
