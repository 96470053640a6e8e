// pds_task_hover_hold.hip -- observation noise with observation_frequency below the simulation frequency
// (obs_rate > 1: the Kalman-hold branch of compute_observation, envs/hover.py:150-156 and the Circle / TakeOff
// equivalents): 8 variants per task (motor dynamics x domain randomisation x thrust noise), control_mode PWM.
#include "pds_step.h"

namespace pds {
void launch_hover_hold(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) { launch_hold<PDS_TASK_HOVER>(kind, f, grid, s, a); }
}  // namespace pds
