// ba.cpp -- BATOTP::BA: construction, the public sweep()/optimize() entry points.
//
// sweep() keeps the reference contract (reference batotp/ba.cpp:979-1195: reads the spline
// coefficient arrays of the Traj, integrates (s, sdot) in the direction set by setIntegDir() and
// publishes sMVC / sdot / nPts / tMVC / tTotalTraj) but the integration itself runs in the HIP
// sweep kernel: see ba_device.cpp and include/batotp_hip.h.  optimize() is the reference call
// sequence of ba.cpp:2538-2573.
#include "ba.h"

#include <cstdio>

namespace BATOTP
{

BA::BA(void)
{
   // same default folder layout as the reference driver expects (reference ba.cpp:67-74)
   _HomeFolder = "../";
   _InputFolder = _HomeFolder + "input/";
   _OutputFolder = _HomeFolder + "output/";
   setErrorOptimization(NO_ERROR);
}

BA::~BA(void) {}

int BA::sweep(Traj &traj)
{
   return deviceSweep(traj);
}

int BA::optimize(Traj &traj)
{
   setErrorOptimization(NO_ERROR);

   if (interpInputData(traj) == -1) return -1;
   if (traj.nPts < 4) return -1;

   setIntegDir(-1);
   setIsLastSweep(false);
   if (sweep(traj) == -1) return -1;

   setIntegDir(1);
   setIsLastSweep(true);
   if (sweep(traj) == -1) return -1;

   interpOutputData(traj);
   return 0;
}

} // namespace BATOTP
