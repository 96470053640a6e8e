// pds_rollout_circle.hip -- instantiates the fused rollout kernel of pds_rollout.h for one task
// ({lean, reference default} x {with, without motor dynamics}; control_mode PWM, no latency / hold / ground effect).
#include "pds_rollout.h"

namespace pds {
bool launch_rollout_circle(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) { return launch_rollout_task<PDS_TASK_CIRCLE>(f, grid, s, ra); }
}  // namespace pds
