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
