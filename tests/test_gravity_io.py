"""The host-side coefficient loaders (C-ABI nyx_hip_load_cof / _shadr) against an independent Python restatement of
GravityFieldData::from_cof / ::load (reference io/gravity.rs:150-501) and the committed JGM3 fixture."""
import gzip
import os
import sys

import numpy as np
import pytest

import nyx_amd as nx
from scenarios import JGM3_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from convert_cof import parse_cof  # noqa: E402

COF = """COMMENT test
CCCCCCCCCCCCCC
POTFIELD 4 4  1 3.98600441500000e+14 6.37813630000000e+06 1.00000000000000e+00
RECOEF    2  0   -4.84165374886470e-04
RECOEF    2  1   -1.86987640000000e-10 1.19528010000000e-09
RECOEF    2  2    2.43926074865630e-06-1.40026639758800e-06
RECOEF    3  0    9.57170590888000e-07
RECOEF    3  1    2.03013720555300e-06 2.48130798255610e-07
RECOEF    3  3    7.21144939823090e-07 1.41420398473540e-06
RECOEF    4  1   -5.36243554298510e-07-4.73772370615970e-07
RECOEF    4  4   -1.88481367425270e-07 3.08848036903550e-07
RECOEF    5  0    6.86589879865430e-08
END
"""
FRAME = nx.Frame(nx.EARTH, 398600.4415, 6378.1363, nx.IAU_EARTH_ROTATION)


def idx(n, m):
    return n * (n + 1) // 2 + m


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("gz", [False, True])
def test_cof_loader_quirks(tmp_path, gz, crlf):
    text = COF.replace("\n", "\r\n") if crlf else COF
    path = tmp_path / ("t.cof.gz" if gz else "t.cof")
    (gzip.open if gz else open)(path, "wb").write(text.encode())
    g = nx.GravityFieldData.from_cof(str(path), 4, 4, gz, FRAME)
    assert (g.degree, g.order) == (4, 4)  # degree-5 line stops the read (io/gravity.rs:324-328)
    assert g.c_nm[idx(2, 0)] == -4.84165374886470e-04 and g.s_nm[idx(2, 0)] == 0.0
    assert g.s_nm[idx(2, 1)] == 1.19528010000000e-09                    # separate positive S
    assert (g.c_nm[idx(2, 2)], g.s_nm[idx(2, 2)]) == (2.43926074865630e-06, -1.40026639758800e-06)  # glued, C>0 S<0
    assert (g.c_nm[idx(4, 1)], g.s_nm[idx(4, 1)]) == (-5.36243554298510e-07, -4.73772370615970e-07)  # glued, both <0
    d, o, c, s = parse_cof(str(path), 4, 4)
    np.testing.assert_array_equal(g.c_nm, c[: g.c_nm.size])
    np.testing.assert_array_equal(g.s_nm, s[: g.s_nm.size])
    # order filter: skipped coefficients still count for the reported max order (io/gravity.rs:330-366)
    g2 = nx.GravityFieldData.from_cof(str(path), 3, 1, gz, FRAME)
    assert (g2.degree, g2.order) == (3, 3) and g2.c_nm[idx(3, 3)] == 0.0 and g2.c_nm[idx(3, 1)] == 2.03013720555300e-06


def test_cof_errors(tmp_path):
    with pytest.raises(IOError, match="FileUnreadable"):
        nx.GravityFieldData.from_cof(str(tmp_path / "missing.cof"), 4, 4, False, FRAME)
    bad = tmp_path / "bad.cof"
    bad.write_text("RECOEF    2  x   1.0e-3\n")
    with pytest.raises(IOError, match="could not parse order"):
        nx.GravityFieldData.from_cof(str(bad), 4, 4, False, FRAME)


def test_shadr_loader(tmp_path):
    text = "6378.1, 398600.4, 0, 3, 3, 1\n2, 0, -4.84165D-04, 0.0\n 2 1 -1.8e-10 1.2e-09\n3,3, 7.2e-07,1.4e-06\n4,0,5.0e-07,0\n"
    p = tmp_path / "m.shadr"
    p.write_text(text)
    g = nx.GravityFieldData.from_shadr(str(p), 3, 3, False, FRAME)
    assert (g.degree, g.order) == (3, 3)
    assert g.c_nm[idx(2, 0)] == -4.84165e-04 and g.s_nm[idx(2, 1)] == 1.2e-09 and g.c_nm[idx(3, 3)] == 7.2e-07


def test_fixture_matches_reference_file_when_present():
    """nyx_amd/data/jgm3_70x70.f64 == what the C++ loader reads from the reference's own JGM3.cof.gz
    (only runnable where /root/reference exists; the GPU box uses the committed fixture)."""
    src = "/root/reference/data/01_planetary/JGM3.cof.gz"
    if not os.path.exists(src):
        pytest.skip("reference checkout not present")
    g = nx.GravityFieldData.from_cof(src, 70, 70, True, FRAME)  # reference io/gravity.rs:533-567 loads 50x50 of the same file
    f = nx.GravityFieldData.from_packed_file(JGM3_PATH, FRAME)
    assert (g.degree, g.order) == (70, 70) == (f.degree, f.order)
    np.testing.assert_array_equal(g.c_nm, f.c_nm)
    np.testing.assert_array_equal(g.s_nm, f.s_nm)
    g50 = nx.GravityFieldData.from_cof(src, 50, 50, True, FRAME)
    assert (g50.degree, g50.order) == (50, 50)
