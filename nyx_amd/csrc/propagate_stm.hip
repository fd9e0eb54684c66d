// propagate_stm.hip - one workgroup shape of the propagation kernel in its own translation unit (see NYX_KERNEL in
// propagate_kernel.hip): its own __launch_bounds__, hence its own register budget; compiled in parallel with the others.
#define NYX_EMIT 4 /* NYX_EMIT_STM */
#include "propagate_kernel.hip"
