// pds_task_circle_pid.hip -- the PID control modes (AttitudeRate / Attitude, envs/control.py:120-287) of the
// fused step / K-step kernels for one task: 2 x 16 variants (no ground effect).
#include "pds_step.h"

namespace pds {
void launch_circle_pid(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) { launch_pid<PDS_TASK_CIRCLE>(kind, f, grid, s, a); }
}  // namespace pds
