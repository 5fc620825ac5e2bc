// ba_output.cpp -- BA::interpOutputData: the optimised s(t) of one path turned into a constant-time trajectory.
//
// Since round 4 this is not host arithmetic any more: the stage of reference batotp/ba.cpp:1661-1931 -- s(t) spline over the
// integration steps, re-sampling at the output resolution, evaluation of the path splines there, kinematics, torque
// recomputation, smoothing / down-sampling, re-interpolation when the output is finer than the integration step -- runs
// behind the C-ABI (batotp_hip_output: the HIP kernels of batotp_amd/csrc/output.hip.h) on a batch of one path that
// BA::deviceOutputOne (ba_device.cpp) builds from the public Traj arrays.  This file keeps the decision and the messages.
#include <cstdio>

#include "ba.h"
#include "batotp_hip.h"

namespace BATOTP
{

int BA::interpOutputData(Traj &traj)
{
   if (traj.tMVC.size() < 2 || traj.sMVC.size() != traj.tMVC.size() || traj.sdot.size() != traj.tMVC.size())
   {
      printf("interpOutputData(): the trajectory holds no forward curve (run sweep(-1) and sweep(+1) first).\n");
      return -1;
   }
   batotp_output_params prm;
   if (exportOutputParams(&prm) != 0)
   {
      // what the device output stage does not cover (include/batotp_hip.h): torque recomputation for a robot without a
      // dynamics model on the device, path / robot kinds without kinematics there
      printf("interpOutputData(): robotType=%s with pathType=%s%s is not covered by the device output stage.\n", _robotTypeStr.c_str(),
             _pathTypeStr.c_str(), _isTrqConOn ? " and torque constraints" : "");
      return -1;
   }
   return deviceOutputOne(traj, &prm);
}

} // namespace BATOTP
