// oracle/traj_offsets.cpp -- TEST TOOLING (build container only): byte offsets of the BATOTP::Traj members that
// oracle/make_golden_f64.py reads out of the running reference binary.  They are taken from this repository's OWN
// batotp_amd/host/ba.h, which is member-for-member the aggregate of the reference (ba.h:59-153; the reference's test/main.cpp
// compiles unchanged against it, tests/test_host_api.py); the generator cross-checks them at run time against what the
// reference binary itself prints (point counts, traversal time) before it trusts a single value.
#include <cstddef>
#include <cstdio>
#include "ba.h"
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
using BATOTP::Traj;
#define OFF(m) printf("  \"%s\": %zu,\n", #m, offsetof(Traj, m))
int main()
{
   printf("{\n");
   OFF(nPts); OFF(tTotalTraj); OFF(curSegMVC); OFF(tauMVC); OFF(sCur); OFF(sdotCur); OFF(sddotH); OFF(sddotL);
   OFF(sMVC); OFF(tMVC); OFF(sdot); OFF(thetaDpt); OFF(CartAccCoeffs); OFF(nPtsC); OFF(sresC); OFF(vFact); OFF(curSegC); OFF(tauC); OFF(sC);
   printf("  \"sizeof_Traj\": %zu,\n  \"sizeof_vector\": %zu\n}\n", sizeof(Traj), sizeof(std::vector<double>));
   return 0;
}
