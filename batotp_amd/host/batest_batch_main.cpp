// batest_batch_main.cpp -- command-line driver of the many-path extension BA::optimizeBatch().
//
//   batest_batch config.dat nPaths [--auto-integ-res] [--devices N | --all-devices]
//
// --auto-integ-res leaves BA's class default on (reference ba.h:309; the reference's own driver switches it off,
// test/main.cpp:53): every path then integrates with the step the rule of ba.cpp:493-556 derives from it.
// (--host-resample / --host-output are accepted and ignored: since round 4 both stages always run behind the C-ABI.)
//
// Loads the trajectory named by the configuration nPaths times, optimises all copies as one device
// batch and writes, for the first and the last path, the same files the single-path driver writes
// (traj_out.dat / s-sdot.dat) into ./out_first/ and ./out_last/.  Configuration, input and
// outputs live in the current directory.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ba.h"
#include "util.h"

using namespace BATOTP;

int main(int argc, char *argv[])
{
   if (argc < 3)
   {
      fprintf(stderr, "usage: batest_batch config.dat nPaths [--auto-integ-res] [--devices N | --all-devices]\n");
      return 2;
   }
   const int nPaths = atoi(argv[2]);
   if (nPaths < 1) return 2;
   bool hostResample = false, hostOutput = false, allDevices = false, autoIntegRes = false;
   int nDevices = 0;
   for (int k = 3; k < argc; ++k)
   {
      if (std::string(argv[k]) == "--host-resample") hostResample = true;
      if (std::string(argv[k]) == "--host-output") hostOutput = true;
      if (std::string(argv[k]) == "--all-devices") allDevices = true;
      if (std::string(argv[k]) == "--auto-integ-res") autoIntegRes = true;
      if (std::string(argv[k]) == "--devices" && k + 1 < argc) nDevices = atoi(argv[++k]);
   }

   BA planner;
   planner.setHomeFolder("./");
   planner.setInputFolder("./");
   planner.setOutputFolder("./");
   planner.setIsAutoIntegRes(autoIntegRes);
   planner.setDeviceResample(!hostResample);
   planner.setDeviceOutput(!hostOutput);
   if (planner.readConfigData((std::string("./") + argv[1]).c_str()) == -1) return 1;
   if (allDevices) nDevices = planner.useAllDevices();
   else if (nDevices > 0)
   {
      std::vector<int> devices((size_t)nDevices);
      for (int d = 0; d < nDevices; ++d) devices[d] = d;
      planner.setDevices(devices);
   }

   std::vector<Traj> paths(nPaths);
   if (planner.loadTrajectoryData(paths[0]) == -1) return 1;
   for (int p = 1; p < nPaths; ++p) paths[p] = paths[0];

   const Time t0 = getTime();
   const int failed = planner.optimizeBatch(paths);
   const Time t1 = getTime();
   if (nDevices > 1) printf("\noptimizeBatch: paths sharded over %d devices\n", nDevices);
   printf("\noptimizeBatch: %d paths, %d failed, %.3f s (resampling %s: %.3f ms; output stage %s: %.3f ms, kernels %.3f ms)\n", nPaths, failed,
          diffTime(t1, t0), "device", planner.getLastResampleMs(), "device",
          planner.getLastOutputMs(), planner.getLastOutputKernelMs());
   if (failed < 0 || failed == nPaths) return 1;

   const char *dirs[2] = {"./out_first/", "./out_last/"};
   const int which[2] = {0, nPaths - 1};
   for (int k = 0; k < 2; ++k)
   {
      mkDirIfNec(dirs[k]);
      planner.setOutputFolder(dirs[k]);
      planner.writeOutputData(paths[which[k]]);
   }
   return failed == 0 ? 0 : 3;
}
