// propagate_p2.hip - the sixteen-wave plain kernel for cooperative launches whose hand-off has TWO parts (two helper workgroups per
// owner and evaluation: large fields with more idle CUs than owners, see DevBatch.coop_parts) in its own translation unit: the
// two-part mailbox calls are compiled into THIS kernel only (NYX_COOP_TWO_PARTS), the default kernel keeps the role code it had.
#define NYX_EMIT 32 /* NYX_EMIT_PLAIN16_P2 */
#define NYX_COOP_TWO_PARTS 1
#include "propagate_kernel.hip"
