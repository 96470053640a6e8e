// pds_task_circle.hip -- instantiates the fused step / reset kernels of pds_step.h for one task
// (32 step variants: motor dynamics x domain randomisation x ground effect x thrust noise x
// observation noise; 8 reset variants).
#include "pds_step.h"

namespace pds {
PDS_DEFINE_TASK_LAUNCHERS(circle, PDS_TASK_CIRCLE)
}  // namespace pds
