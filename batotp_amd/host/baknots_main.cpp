// baknots_main.cpp -- "baknots": the host half of BA::interpInputData as a command-line tool.
// Reads config.dat and the taught path from the current directory, runs the host resampler (config + trajectory IO +
// path resampling; no device call is made) and writes
//   knots.bin    : int64 N, int64 nJ, int64 nCart, double sres, double y[nJ+nCart][N]   (the uniform-s knots)
//   problem.bin  : raw struct batotp_problem (include/batotp_hip.h)
//   taught.bin   : int64 n, int64 nJ, int64 nCart, double sres, double x[nJ+nCart][n]   (the taught points, absent rows zero)
//   resample.bin : int32 covered by the device resampler (1/0), raw struct batotp_resample_params
//   output.bin   : int32 covered by the device output stage (1/0), raw struct batotp_output_params
// bench.py and the tools use it to prepare synthetic inputs; oracle/Makefile builds the same source against the oracle
// shim (dump_knots) for the fixture generators.
// Usage: baknots config.dat
#include <cstdio>
#include <vector>

#include "ba.h"
#include "batotp_hip.h"

using namespace BATOTP;

int main(int argc, char **argv)
{
   if (argc < 2) { fprintf(stderr, "usage: baknots config.dat\n"); return 2; }
   BA ba;
   Traj tr;
   ba.setHomeFolder("./"); ba.setInputFolder("./"); ba.setOutputFolder("./");
   ba.setIsAutoIntegRes(false);
   if (ba.readConfigData((std::string("./") + argv[1]).c_str()) == -1) return 1;
   if (ba.loadTrajectoryData(tr) == -1) return 1;
   {
      // the taught points and the parameters of the device stages
      batotp_resample_params R;
      ba.dropRepeatedTimestamps(tr);
      const int supported = ba.exportResampleParams(tr, &R) == 0 ? 1 : 0;
      FILE *g = fopen("resample.bin", "wb");
      fwrite(&supported, 4, 1, g); fwrite(&R, sizeof(R), 1, g);
      fclose(g);
      batotp_output_params O;
      const int outSupported = ba.exportOutputParams(&O) == 0 ? 1 : 0;
      g = fopen("output.bin", "wb");
      fwrite(&outSupported, 4, 1, g); fwrite(&O, sizeof(O), 1, g);
      fclose(g);
      const long long n = tr.nPts, nJ = ba.getNumJoints(), nC = ba.getNumCart();
      g = fopen("taught.bin", "wb");
      fwrite(&n, 8, 1, g); fwrite(&nJ, 8, 1, g); fwrite(&nC, 8, 1, g); fwrite(&tr.sres, 8, 1, g);
      std::vector<double> zeros(n, 0.0);
      for (long long j = 0; j < nJ; ++j)
         fwrite(((size_t)j < tr.theta.size() && tr.theta[j].size() >= (size_t)n) ? tr.theta[j].data() : zeros.data(), 8, n, g);
      for (long long j = 0; j < nC; ++j)
         fwrite(((size_t)j < tr.cart.size() && tr.cart[j].size() >= (size_t)n) ? tr.cart[j].data() : zeros.data(), 8, n, g);
      fclose(g);
   }
   if (ba.resampleToKnots(tr) != 0) return 1;
   const long long N = tr.nPts, nJ = ba.getNumJoints(), nC = ba.getNumCart();
   FILE *f = fopen("knots.bin", "wb");
   fwrite(&N, 8, 1, f); fwrite(&nJ, 8, 1, f); fwrite(&nC, 8, 1, f); fwrite(&tr.sres, 8, 1, f);
   std::vector<double> zeros(N, 0.0);
   for (long long j = 0; j < nJ; ++j) fwrite(tr.theta[j].data(), 8, N, f);
   for (long long j = 0; j < nC; ++j)
   {
      const bool have = (size_t)j < tr.cart.size() && tr.cart[j].size() >= (size_t)N;
      fwrite(have ? tr.cart[j].data() : zeros.data(), 8, N, f);
   }
   fclose(f);
   batotp_problem P;
   ba.exportProblem(&P);
   f = fopen("problem.bin", "wb");
   fwrite(&P, sizeof(P), 1, f);
   fclose(f);
   printf("baknots: N=%lld nJ=%lld nCart=%lld sres=%.17g\n", N, nJ, nC, tr.sres);
   return 0;
}
