"""The Rust binding text of INTEGRATION.md is generated from include/nyx_hip.h; this keeps the two from drifting and checks
the layout the generator assumes (repr(C)) against the built library.  (No rustc in this image: this is the machine check.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_rust_sys as g  # noqa: E402
from nyx_amd import _abi  # noqa: E402


def test_integration_md_holds_the_generated_block():
    text, _ = g.generate()
    doc = open(g.DOC).read()
    i, j = doc.index(g.BEGIN) + len(g.BEGIN), doc.index(g.END)
    assert doc[i:j] == "\n```rust\n" + text + "```\n", "run: python tools/gen_rust_sys.py --update"


def test_every_entry_point_and_struct_is_bound():
    text, sizes = g.generate()
    for name in _abi.EXPORTS:
        assert f"pub fn {name}(" in text, name
    # every struct the library reports a size for is in the binding, with the size the C compiler gave it
    lib = _abi.load_library()
    order = ["nyx_hip_integ_opts_t", "nyx_hip_cheby_segment_t", "nyx_hip_body_t", "nyx_hip_rotation_t", "nyx_hip_gravity_field_t",
             "nyx_hip_srp_t", "nyx_hip_drag_t", "nyx_hip_config_t", "nyx_hip_states_t", "nyx_hip_step_stats_t", "nyx_hip_traj_t",
             "nyx_hip_solid_tides_t", "nyx_hip_predict_t", "nyx_hip_predict_history_t", "nyx_hip_process_noise_t", "nyx_hip_tuning_t"]
    for which, name in enumerate(order):
        assert lib.nyx_hip_abi_sizeof(which) == sizes[name][0], name
    assert lib.nyx_hip_abi_sizeof(len(order)) in (-1, sizes.get("nyx_hip_event_t", (0,))[0], sizes.get("nyx_hip_estimates_t", (0,))[0])


def test_stm_order_is_stated_once_and_right():
    doc = open(g.DOC).read()
    assert "row-major 9" not in doc and "column-major" in doc
