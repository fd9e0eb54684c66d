"""CPU: the free-order placement of the owner's column runs (nyx_amd/csrc/col_partition.h, used by fill_schedule in abi.cpp) as a
stand-alone C++ check - g++ only, no HIP, no GPU: every column dealt exactly once, contiguous runs in list order, every wave inside
the tolerance the search reports, the 70x70 cooperative shape placed at a tolerance the linear partition of round 4 misses by more
than 2x, random shapes, refused inputs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_col_partition_check(tmp_path):
    exe = str(tmp_path / "col_partition_check")
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cxx", "col_partition_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert r.stdout.strip().endswith("ok")
    assert "70x70 owner, frozen table" in r.stdout
