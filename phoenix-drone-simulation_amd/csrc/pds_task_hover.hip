// pds_task_hover.hip -- instantiates the fused step / reset kernels of pds_step.h for one task
// (32 step variants: motor dynamics x domain randomisation x ground effect x thrust noise x
// observation noise; 8 reset variants).
#include "pds_step.h"

namespace pds {
PDS_DEFINE_TASK_LAUNCHERS(hover, PDS_TASK_HOVER)
}  // namespace pds
