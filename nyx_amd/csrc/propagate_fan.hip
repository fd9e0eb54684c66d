// propagate_fan.hip - the sixteen-wave plain kernel for cooperative launches in the FAN-OUT mode (small shards of an ensemble: the idle CUs
// outnumber the trajectory-owning workgroups at least two to one, so every owner gets several DEDICATED helper workgroups and the
// columns of an evaluation are dealt over them; DevBatch.coop_fan, helper_body) in its own translation unit: the dedicated producer,
// the lead's collection of the other parts and the many-part fallback are compiled into THIS kernel only (NYX_COOP_FAN), the default
// kernel keeps the role code it had.  The owner's side is the single-part protocol of the default kernel.
#define NYX_EMIT 128 /* NYX_EMIT_PLAIN16_FAN */
#define NYX_COOP_FAN 1
#include "propagate_kernel.hip"
