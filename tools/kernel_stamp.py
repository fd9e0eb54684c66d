#!/usr/bin/env python3
"""Identity of the device code a measurement belongs to: sha256 over the sources every kernel and its launch logic are built
from (first 16 hex digits).  tools/make_profiles.py stamps it into profiles/*_hbm_traffic.json, bench.py refuses a traffic
figure whose stamp is not the one of the tree it runs from (`roofline.traffic` = null, `traffic_source` says why): a counter
pass is only quoted for the build it was taken on.  Standard library only."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = [
    "nyx_amd/csrc/propagate_kernel.hip", "nyx_amd/csrc/pk_epoch_data.h", "nyx_amd/csrc/pk_force_models.h", "nyx_amd/csrc/pk_harmonics.h", "nyx_amd/csrc/pk_cooperative.h", "nyx_amd/csrc/pk_error_stm.h", "nyx_amd/csrc/pk_state_lds.h", "nyx_amd/csrc/pk_integrator_ool.h", "nyx_amd/csrc/pk_segment_update.h",
    "nyx_amd/csrc/propagate_w8.hip", "nyx_amd/csrc/propagate_stm.hip", "nyx_amd/csrc/propagate_stmq.hip",
    "nyx_amd/csrc/propagate_stmq_w8.hip", "nyx_amd/csrc/propagate_p2.hip", "nyx_amd/csrc/propagate_w8n.hip", "nyx_amd/csrc/propagate_fan.hip", "nyx_amd/csrc/harm_stream_asm.h", "nyx_amd/csrc/devcfg.h", "nyx_amd/csrc/butcher.h",
    "nyx_amd/csrc/hifitime_dev.h", "nyx_amd/csrc/event_dev.h", "nyx_amd/csrc/predict_kernel.hip", "nyx_amd/csrc/predict_args.h",
    "nyx_amd/csrc/moments_kernel.hip", "nyx_amd/csrc/abi.cpp", "nyx_amd/csrc/col_partition.h", "include/nyx_hip.h",
]


def kernel_source_stamp(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        p = os.path.join(root, rel)
        if not os.path.exists(p):
            continue
        h.update(rel.encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_stamp())
