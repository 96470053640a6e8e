// pds_task_hover_pid_ge.hip -- the PID control modes (envs/control.py:120-287) together with the ground-effect extension
// (envs/physics.py:27-58) for one task: 2 x 16 step variants (round 5; no latency ring, no Kalman hold).
#include "pds_step.h"

namespace pds {
void launch_hover_pid_ge(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) { launch_pid_ge<PDS_TASK_HOVER>(kind, f, grid, s, a); }
}  // namespace pds
