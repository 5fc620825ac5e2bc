// dump_knots.cpp -- TEST INFRASTRUCTURE (golden-fixture generation, build container only).
// Runs the host half of the BA pipeline (config + trajectory IO + path resampling, no device
// call) on a config.dat in the current directory and writes
//   knots.bin   : int64 N, int64 nJ, int64 nCart, double sres, double y[nJ+nCart][N]
//   problem.bin : raw struct batotp_problem (include/batotp_hip.h)
// Usage: dump_knots config.dat
#include <cstdio>
#include <vector>

#include "ba.h"
#include "batotp_hip.h"

using namespace BATOTP;

int main(int argc, char **argv)
{
   if (argc < 2) { fprintf(stderr, "usage: dump_knots config.dat\n"); return 2; }
   BA ba;
   Traj tr;
   ba.setHomeFolder("./"); ba.setInputFolder("./"); ba.setOutputFolder("./");
   ba.setIsAutoIntegRes(false);
   if (ba.readConfigData((std::string("./") + argv[1]).c_str()) == -1) return 1;
   if (ba.loadTrajectoryData(tr) == -1) return 1;
   {
      // the taught points and the resampling parameters, for the device-resampler fixtures:
      //   taught.bin   : int64 n, int64 nJ, int64 nCart, double sres, double x[nJ+nCart][n] (absent rows zero)
      //   resample.bin : int32 supported (1/0), raw struct batotp_resample_params
      batotp_resample_params R;
      ba.dropRepeatedTimestamps(tr);
      const int supported = ba.exportResampleParams(tr, &R) == 0 ? 1 : 0;
      FILE *g = fopen("resample.bin", "wb");
      fwrite(&supported, 4, 1, g); fwrite(&R, sizeof(R), 1, g);
      fclose(g);
      //   output.bin   : int32 supported (1/0), raw struct batotp_output_params
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
   printf("dump_knots: N=%lld nJ=%lld nCart=%lld sres=%.17g\n", N, nJ, nC, tr.sres);
   return 0;
}
