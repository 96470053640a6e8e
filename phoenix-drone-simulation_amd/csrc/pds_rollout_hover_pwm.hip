// pds_rollout_hover_pwm.hip -- the fused rollout kernels of control_mode PWM without latency ring / Kalman hold: all eight
// settings of domain randomisation x thrust noise x observation noise, with and without motor dynamics.
#include "pds_rollout.h"

namespace pds {
bool launch_rollout_hover_pwm(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) { return launch_rollout_pwm_family<PDS_TASK_HOVER>(f, grid, s, ra); }
}  // namespace pds
