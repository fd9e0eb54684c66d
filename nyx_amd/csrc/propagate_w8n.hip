// propagate_w8n.hip - the eight-wave plain kernel for dynamics WITHOUT a body-fixed model (no gravity field, no drag, no tides: point
// masses and SRP around the two-body term, BASELINE config 3) in its own translation unit: `has_grav`, `has_drag`, `has_tides` and
// `has_grav2` are compile-time constants here (NYX_ASSUME_SMALL, see role_loop), so the role code carries none of those models.
// nyx_launch_propagate picks it on the host's word (`no_body_fixed`); same source, same arithmetic, same bits as propagate_w8.hip.
#define NYX_EMIT 64 /* NYX_EMIT_PLAIN8N */
#define NYX_ASSUME_SMALL 1
#include "propagate_kernel.hip"
