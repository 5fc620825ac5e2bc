// batest_main.cpp -- command-line driver of the BA library ("batest").
//
// Same call order, console report and compTimes.dat as the reference demo driver
// (reference test/main.cpp:57-113): read config -> load trajectory -> interpInputData ->
// reverse sweep -> forward sweep -> interpOutputData -> writeOutputData.  With an argument the
// configuration file, inputs and outputs all live in the current directory; without one the
// ../input and ../output folders are used.
#include <cstdio>
#include <string>

#include "ba.h"
#include "util.h"

using namespace BATOTP;

namespace
{
struct Stopwatch
{
   Time mark[6];
   void lap(int k) { mark[k] = getTime(); }
   double span(int to, int from) const { return diffTime(mark[to], mark[from]); }
};
} // namespace

int main(int argc, char *argv[])
{
   BA planner;
   Traj path;
   Stopwatch clock;

   std::string configName = "config.dat";
   if (argc > 1)
   {
      configName = argv[1];
      planner.setHomeFolder("./");
      planner.setInputFolder("./");
      planner.setOutputFolder("./");
   }
   else
   {
      mkDirIfNec(planner.getOutputFolder().c_str());
   }
   const std::string configPath = planner.getInputFolder() + configName;
   // the reference's driver switches the automatic integration resolution off (test/main.cpp:53); a second argument
   // "--auto-integ-res" keeps the class default (reference ba.h:309)
   planner.setIsAutoIntegRes(argc > 2 && std::string(argv[2]) == "--auto-integ-res");

   clock.lap(0);
   if (planner.readConfigData(configPath.c_str()) == -1) return -1;
   if (planner.loadTrajectoryData(path) == -1) return -1;

   clock.lap(1);
   printf("-----Interpolation of input data-----------\n");
   if (planner.interpInputData(path) == -1) return -1;

   clock.lap(2);
   printf("\n--Constant-step accel. constraint integ.--\n");
   planner.setIntegDir(-1);
   planner.setIsLastSweep(false);
   if (planner.sweep(path) == -1) return -1;
   planner.setIntegDir(1);
   planner.setIsLastSweep(true);
   if (planner.sweep(path) == -1) return -1;

   clock.lap(3);
   printf("---------------------------------------\n");
   planner.interpOutputData(path);
   clock.lap(4);
   planner.writeOutputData(path);
   clock.lap(5);

   printf("\nComputational times (sec):\n");
   printf("Input data interp.      : %f\n", clock.span(2, 1));
   printf("Accel. constraint integ.: %f\n", clock.span(3, 2));
   printf("Reading input data      : %f\n", clock.span(1, 0));
   printf("Writing Output data     : %f\n", clock.span(5, 4));
   printf("Total,     with file IO : %f\n", clock.span(5, 0));
   printf("Total,  without file IO : %f\n", clock.span(4, 1));
   printf("\n");

   // compTimes.dat: sweep time, total without IO, total with IO (float32 each)
   const std::string timesPath = planner.getOutputFolder() + "compTimes.dat";
   FILE *fid = fopen(timesPath.c_str(), "wb");
   if (fid != nullptr)
   {
      const float t3[3] = {(float)clock.span(3, 2), (float)clock.span(4, 1), (float)clock.span(5, 0)};
      fwrite(t3, 4, 3, fid);
      fclose(fid);
   }
   return 0;
}
