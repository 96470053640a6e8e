// pds_rollout_hist_hover.hip -- instantiates the one-launch rollout for observation histories other than 2
// (csrc/pds_rollout_hist.h) for one task: {lean, reference-default noise} x {with, without motor dynamics} x 4 input widths.
#include "pds_rollout_hist.h"

namespace pds {
bool launch_rollout_hist_hover(const LaunchFlags &f, int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra) {
  return launch_rollout_hist_task<PDS_TASK_HOVER>(f, hn, grid, s, ra);
}
}  // namespace pds
