// pds_rollout_circle_lat.hip -- the fused rollout kernels of the latency ring (envs/agents.py:267-276) with control_mode PWM and
// with the PID modes: {lean, reference default} x {with, without motor dynamics}.
#include "pds_rollout.h"

namespace pds {
bool launch_rollout_circle_lat(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) { return launch_rollout_lat_family<PDS_TASK_CIRCLE>(f, grid, s, ra); }
}  // namespace pds
