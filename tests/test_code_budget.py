"""CPU-side fence around the built code objects (Weak 6 / item 4).

The seven propagation kernels are translation units of ONE 3 300-line source selected by macros; their speed depends on what the
register allocator does with code that is never executed (profiles/HISTORY.md, round 4 log 12: the mere presence of the two-part
hand-off calls cost the default kernel 3-5 %).  A maintainer who adds a force model can lose that much on a configuration that does
not use it, and only a GPU A/B run would notice.  This test reads the metadata of every kernel in nyx_amd/libnyx_hip.so - VGPRs,
scratch bytes per lane, static spill counts, .text bytes (tools/kernel_meta.py) - and the per-ROLE census of the headline kernel's
stage loops (scratch loads / stores and SGPR-spill lane moves between a role's first and last barrier, tools/kernel_roles.py) and
holds them to the budgets committed in tests/golden/code_budget.json: a figure above its budget fails with the number, and a
figure more than 25 % BELOW its budget fails too (the budget is then stale: tighten it with `python tools/code_budget.py --update`).
The figures are register-allocation outcomes of ONE compiler: the budget file records the hipcc it was written under and the tests
skip under any other (a ROCm bump is not a regression of this source; re-run the tool and read the diff).
No GPU needed: hipcc cross-compiles, the objects are inspected with the LLVM binutils of the ROCm image."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "nyx_amd", "libnyx_hip.so")
BUDGET = os.path.join(ROOT, "tests", "golden", "code_budget.json")

pytestmark = pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("objcopy") is None,
                                reason="no built library or no LLVM binutils here")


_CACHE = {}


def _same_toolchain():
    import code_budget
    want = json.load(open(BUDGET)).get("hipcc")
    have = code_budget.toolchain()
    if want and have and want != have:
        pytest.skip(f"budgets were written under hipcc {want}, this is {have}: python tools/code_budget.py --update")


def measured():
    import code_budget
    _same_toolchain()
    if "m" not in _CACHE:
        _CACHE["m"] = code_budget.measure(LIB)
    return _CACHE["m"]


def test_every_kernel_is_inside_its_budget():
    budget = json.load(open(BUDGET))
    got = measured()
    problems = []
    for kernel, b in budget["kernels"].items():
        assert kernel in got["kernels"], f"{kernel}: not in the library any more (python tools/code_budget.py --update)"
        g = got["kernels"][kernel]
        for key, limit in b.items():
            v = g[key]
            if v > limit:
                problems.append(f"{kernel}.{key}: {v} > budget {limit}")
            elif key in ("scratch_bytes", "vgpr_spills", "text_bytes") and limit > 64 and v < 0.75 * limit:
                problems.append(f"{kernel}.{key}: {v} is more than 25 % under its budget {limit} - tighten it")
    new = sorted(set(got["kernels"]) - set(budget["kernels"]))
    assert not new, f"kernels without a budget: {new}"
    assert not problems, "\n".join(problems)


def test_the_headline_roles_stage_loops_are_inside_their_budget():
    budget = json.load(open(BUDGET))
    got = measured()
    problems = []
    for key, b in budget["roles"].items():
        assert key in got["roles"], f"{key}: role not found in the code object (markers of role_loop, tools/kernel_roles.py)"
        g = got["roles"][key]
        for field, limit in b.items():
            if g[field] > limit:
                problems.append(f"{key}.{field}: {g[field]} > budget {limit}")
    assert not problems, "\n".join(problems)
