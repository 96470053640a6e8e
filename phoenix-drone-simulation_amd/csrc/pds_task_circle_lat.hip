// pds_task_circle_lat.hip -- the use_latency variants (delayed-action ring, envs/agents.py:267-276) of the
// fused step / K-step / reset kernels for one task: 16 per control mode (no ground effect), full tile.
#include "pds_step.h"

namespace pds {
void launch_circle_lat(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) { launch_lat<PDS_TASK_CIRCLE>(kind, f, grid, s, a); }
}  // namespace pds
