// oracle/dump_autores.cpp -- TEST TOOLING, one-off fixture generator (round 4; see oracle/make_autores_fixtures.py).
// Runs the ROUND-3 host resampler of this repository (batotp_amd/host/ba_input.cpp at commit HEAD~ of the commit that
// deleted it: the statement-level restatement of reference ba.cpp:95-863 incl. the automatic integration resolution of
// ba.cpp:493-556) with _isAutoIntegRes left at the class default (true, reference ba.h:309) and dumps the knots, the
// integration step and the s weights / scale type it leaves.  The reference binary cannot produce these vectors: batest
// forces the switch off (test/main.cpp:53).  Kept for provenance; it no longer builds once that file is gone.
#include <cstdio>
#include <vector>
#include "ba.h"
#include "batotp_hip.h"
using namespace BATOTP;
int main(int argc, char **argv)
{
   BA ba; Traj tr;
   ba.setHomeFolder("./"); ba.setInputFolder("./"); ba.setOutputFolder("./");
   if (ba.readConfigData((std::string("./") + argv[1]).c_str()) == -1) return 1;
   ba.setIsAutoIntegRes(true);
   if (ba.loadTrajectoryData(tr) == -1) return 1;
   if (ba.resampleToKnots(tr) != 0) return 1;
   const long long N = tr.nPts, nJ = ba.getNumJoints(), nC = ba.getNumCart();
   batotp_problem P; ba.exportProblem(&P);
   batotp_resample_params R; ba.exportResampleParams(tr, &R);
   FILE *f = fopen("autores.bin", "wb");
   fwrite(&N, 8, 1, f); fwrite(&nJ, 8, 1, f); fwrite(&nC, 8, 1, f); fwrite(&tr.sres, 8, 1, f);
   fwrite(&P.integ_res, 8, 1, f); fwrite(R.s_weights, 8, 3, f); long long st = R.scale_type; fwrite(&st, 8, 1, f);
   std::vector<double> zeros(N, 0.0);
   for (long long j = 0; j < nJ; ++j) fwrite(tr.theta[j].data(), 8, N, f);
   for (long long j = 0; j < nC; ++j) { const bool have = (size_t)j < tr.cart.size() && tr.cart[j].size() >= (size_t)N; fwrite(have ? tr.cart[j].data() : zeros.data(), 8, N, f); }
   fclose(f);
   printf("autores: N=%lld integ_res=%.17g sw=%.17g %.17g %.17g scale=%lld nC=%lld\n", N, P.integ_res, R.s_weights[0], R.s_weights[1], R.s_weights[2], st, nC);
   return 0;
}
