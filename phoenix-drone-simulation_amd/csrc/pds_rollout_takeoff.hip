// pds_rollout_takeoff.hip -- the fused rollout (pds_rollout.h) for TakeOff (control_mode PWM only, envs/takeoff.py:225):
// every noise setting, the latency ring, the Kalman hold.
#include "pds_rollout.h"

namespace pds {
bool launch_rollout_takeoff(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (!rollout_supported(PDS_TASK_TAKEOFF, f)) return false;
  if (f.hold) return launch_rollout_pid_hold_family<PDS_TASK_TAKEOFF>(f, grid, s, ra);
  if (f.lat) return launch_rollout_lat_family<PDS_TASK_TAKEOFF>(f, grid, s, ra);
  return launch_rollout_pwm_family<PDS_TASK_TAKEOFF>(f, grid, s, ra);
}
}  // namespace pds
