"""C++ twin of the Rust host's collective step (tests/cxx/rccl_twin.cpp; INTEGRATION.md "the covariance reduction"): libnyx_hip.so's
device entry points and librccl's ncclAllReduce / ncclAllGather on one launch stream, no PyTorch in the loop.  CPU: it compiles and
links against both libraries (the C-ABI symbols and the RCCL ones resolve).  GPU: one rank per visible device, results equal to the
single-context run."""
import os
import shutil
import subprocess

import pytest

from nyx_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    if shutil.which("hipcc") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no hipcc / rccl.h in this image")
    _abi.load_library()
    exe = str(tmp_path / "rccl_twin")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cxx", "rccl_twin.cpp"), "-L" + os.path.join(ROOT, "nyx_amd"), "-lnyx_hip", "-lrccl",
                    "-Wl,-rpath," + os.path.join(ROOT, "nyx_amd"), "-o", exe], check=True)
    return exe


def test_rccl_twin_compiles_and_links(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "link check only" in r.stdout or "rccl twin:" in r.stdout, r.stdout


@pytest.mark.gpu
def test_rccl_twin_on_gpu(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([_build(tmp_path), "1000"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rccl twin:" in r.stdout and ": ok" in r.stdout and "FAILED" not in r.stdout, r.stdout
    # a ragged shard size too (the gather blocks are padded to the largest shard)
    r = subprocess.run([_build(tmp_path), "333"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and ": ok" in r.stdout, r.stdout + r.stderr
