// pds_task_hover.hip -- instantiates the fused step / K-step / reset kernels of pds_step.h for one task,
// control_mode PWM, no latency (32 variants: motor dynamics x domain randomisation x ground effect x
// thrust noise x observation noise; the variants without observation noise twice: full and half tile).
#include "pds_step.h"

namespace pds {
void launch_hover(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) { launch_base<PDS_TASK_HOVER>(kind, f, grid, s, a); }
}  // namespace pds
