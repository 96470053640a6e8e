// pds_rollout_circle.hip -- the fused rollout (pds_rollout.h) for one task: dispatcher + the PID control modes and the
// Kalman hold; control_mode PWM with every noise setting: pds_rollout_circle_pwm.hip, the latency ring: pds_rollout_circle_lat.hip.
#include "pds_rollout.h"

namespace pds {
bool launch_rollout_circle(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (!rollout_supported(PDS_TASK_CIRCLE, f)) return false;
  if (f.hold) return launch_rollout_pid_hold_family<PDS_TASK_CIRCLE>(f, grid, s, ra);
  if (f.lat) return launch_rollout_circle_lat(f, grid, s, ra);
  if (f.ctrl == 0) return launch_rollout_circle_pwm(f, grid, s, ra);
  return launch_rollout_pid_hold_family<PDS_TASK_CIRCLE>(f, grid, s, ra);
}
}  // namespace pds
