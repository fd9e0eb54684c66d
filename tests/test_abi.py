"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/nyx_hip.h declares,
and its struct layouts match the ctypes mirror.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nyx_amd as nx
from nyx_amd import _abi
from scenarios import leo_full_setup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _abi.load_library()
    header = open(os.path.join(ROOT, "include", "nyx_hip.h")).read()
    declared = set(re.findall(r"^(?:int32_t|void|double|const char \*)\s*(nyx_hip_[a-z_0-9]+)\(", header, flags=re.M))
    assert len(declared) == 25
    assert declared >= set(_abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nyx_hip.h but not exported"


def test_struct_layouts_match_the_header():
    lib = _abi.load_library()
    mirror = [_abi.IntegOpts, _abi.ChebySegment, _abi.Body, _abi.Rotation, _abi.GravityField, _abi.Srp, _abi.Drag,
              _abi.Config, _abi.States, _abi.StepStats, _abi.Traj, _abi.SolidTidesC, _abi.Predict, _abi.PredictHistory,
              _abi.ProcessNoiseC, _abi.Tuning]
    for which, cls in enumerate(mirror):
        assert lib.nyx_hip_abi_sizeof(which) == C.sizeof(cls), cls.__name__


def test_no_device_is_reported_not_papered_over():
    """On a box without a GPU the product path must fail loudly (no CPU fallback)."""
    lib = _abi.load_library()
    if lib.nyx_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    prop, almanac, central = leo_full_setup(degree=4)
    with pytest.raises(RuntimeError, match="nyx_hip_ctx_create failed"):
        nx.GpuContext(prop.compile(almanac, central))


def test_config_flattening():
    prop, almanac, central = leo_full_setup(degree=21, order=10)
    cc = prop.compile(almanac, central)
    cfg = cc.cfg
    assert cfg.abi_version == _abi.ABI_VERSION and cfg.opts.method == _abi.RK89 and cfg.opts.error_ctrl == _abi.RSS_CARTESIAN_STEP
    assert cfg.opts.init_step_ns == 60 * nx.NS_PER_S and cfg.opts.min_step_ns == 1_000_000 and cfg.opts.max_step_ns == 2700 * nx.NS_PER_S
    ids = [cfg.bodies[i].naif_id for i in range(cfg.n_bodies)]
    assert ids[0] == nx.EARTH and set(ids) == {nx.EARTH, nx.SUN, nx.MOON}
    assert cfg.bodies[0].n_chain == 0 and cfg.n_point_masses == 2
    g = cfg.gravity.contents
    assert (g.degree, g.order) == (21, 10)
    # coefficients above the requested order are never stored by the loaders
    assert g.c_nm[21 * 22 // 2 + 11] == 0.0 and g.c_nm[21 * 22 // 2 + 10] != 0.0
    assert g.c_nm[3] == pytest.approx(-4.84165374886470e-04)  # JGM3 C20 (data/01_planetary/JGM3.cof.gz)
    srp = cfg.srp.contents
    assert srp.phi_w_m2 == 1367.0 and srp.estimate == 1 and cfg.bodies[srp.sun_body].naif_id == nx.SUN
    assert srp.n_shadow_bodies == 1 and cfg.bodies[srp.shadow_body[0]].naif_id == nx.EARTH


def test_integrator_options_mirror_the_reference():
    # reference: propagators/options.rs ut_integr_opts::test_options (:216-248)
    o = nx.IntegratorOptions.with_fixed_step_s(1e-1)
    assert o.min_step == nx.seconds(1e-1) == o.max_step and o.tolerance == 0.0 and o.fixed_step
    o = nx.IntegratorOptions.with_adaptive_step_s(1e-2, 10.0, 1e-12, nx.ErrorControl.RSSStep)
    assert o.min_step == nx.seconds(1e-2) and o.max_step == nx.seconds(10.0) and not o.fixed_step and o.init_step == o.max_step
    o = nx.IntegratorOptions()
    assert (o.init_step, o.min_step, o.max_step, o.tolerance, o.attempts, o.fixed_step) == (60 * 10**9, 10**6, 2700 * 10**9, 1e-12, 50, False)
    o = nx.IntegratorOptions.with_max_step(nx.seconds(1.0))
    assert o.init_step == nx.seconds(1.0) == o.max_step and o.min_step == 10**6
    # rk_methods/mod.rs ut_propagator::from_str_ok
    for m in nx.IntegratorMethod:
        assert nx.IntegratorMethod.from_str(m.name.upper()) == m
    with pytest.raises(ValueError):
        nx.IntegratorMethod.from_str("blah")
    # Duration conversions (hifitime): truncation toward zero, to_seconds = whole + sub * 1e-9
    assert nx.seconds(0.1) == 100_000_000 and nx.seconds(29.9999999999) == 29_999_999_999
    assert nx.to_seconds(1_500_000_000) == 1.5 and nx.to_seconds(-30 * 10**9) == -30.0


def _build_cxx_check(tmp_path):
    import subprocess
    exe = str(tmp_path / "host_mirror_check")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "host_mirror_check.cpp"),
                    "-L" + os.path.join(ROOT, "nyx_amd"), "-lnyx_hip", "-Wl,-rpath," + os.path.join(ROOT, "nyx_amd"), "-o", exe], check=True)
    return exe


def test_cxx_host_mirror_compiles_and_links(tmp_path):
    """Also runs the host-only part: the reference's seeded known-answer counts of the dispersions
    (mc/multivariate.rs:420-556) through include/nyx_hip_mc.hpp."""
    import subprocess
    _abi.load_library()
    r = subprocess.run([_build_cxx_check(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dispersion known answers: 312 (want 312), 6 (want 6), resume ok" in r.stdout


@pytest.mark.gpu
def test_cxx_host_mirror_golden_on_gpu(tmp_path):
    import subprocess
    r = subprocess.run([_build_cxx_check(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "max |delta|" in r.stdout, r.stdout + r.stderr
    assert "MultiGpuPropagator (2 contexts" in r.stdout and "FAILED" not in r.stdout, r.stdout


def _create_rc(cc):
    lib = _abi.load_library()
    h = C.c_void_p()
    rc = lib.nyx_hip_ctx_create(C.byref(cc.cfg), 0, C.byref(h))
    if rc == 0:
        lib.nyx_hip_ctx_destroy(h)
    return rc, _abi.last_error()


def test_ctx_create_refuses_what_the_device_code_cannot_represent():
    """Plain-data validation happens before any device is touched: a segment the 16-wide Clenshaw window would truncate,
    degenerate segments and over-long body chains are errors, never silently clipped."""
    from nyx_amd import ephem
    day = 86400.0
    prop, _, central = leo_full_setup(degree=4)

    def almanac_with(n_coeffs, interval=16 * day, n_chain=1):
        al = nx.Almanac()
        s = al.add_segment(ephem.fit_segment(ephem.sun_geocentric, 7.6e8, interval, 3, n_coeffs))
        al.add_body(nx.EARTH, ephem.MU_EARTH, ephem.R_EARTH, [])
        al.add_body(nx.SUN, ephem.MU_SUN, ephem.R_SUN, [(s, +1)] * n_chain)
        al.add_body(nx.MOON, ephem.MU_MOON, ephem.R_MOON, [(s, +1)])
        return al

    rc, msg = _create_rc(prop.compile(almanac_with(33), central))
    assert rc == _abi.RC_UNSUPPORTED and "33 Chebyshev coefficients" in msg
    cc = prop.compile(almanac_with(13), central)
    cc.cfg.segments[0].interval_s = 0.0
    assert _create_rc(cc)[0] == _abi.RC_BAD_ARG
    cc = prop.compile(almanac_with(13), central)
    cc.cfg.segments[0].n_records = 0
    assert _create_rc(cc)[0] == _abi.RC_BAD_ARG
    cc = prop.compile(almanac_with(13), central)
    cc.cfg.bodies[1].n_chain = 5
    rc, msg = _create_rc(cc)
    assert rc == _abi.RC_BAD_ARG and "n_chain" in msg
    # valid 16- and 32-coefficient segments get past validation (and then stop at "no device" on this box)
    for nc in (16, 32):
        rc, msg = _create_rc(prop.compile(almanac_with(nc), central))
        assert rc in (0, _abi.RC_NO_DEVICE), msg


@pytest.mark.gpu
@pytest.mark.parametrize("ncoef", [16, 26, 32])
def test_wide_coefficient_segments_device_vs_oracle(ncoef):
    """The widest segment of the register window (16) and of the rolled path (up to NYX_HIP_MAX_CHEBY_COEFFS = 32, what DE440's
    inner-planet segments and the binary PCKs carry): every coefficient must enter the Clenshaw sum."""
    import oracle_lib
    from nyx_amd import ephem
    from scenarios import EPOCH0_NS, dispersed_leo_batch, pos_vel_errors
    day = 86400.0
    et0 = nx.to_seconds(EPOCH0_NS)
    al = nx.Almanac()
    s_sun = al.add_segment(ephem.fit_segment(ephem.sun_geocentric, et0 - 2 * day, 16 * day, 2, 16))
    moon = ephem.fit_segment(ephem.moon_geocentric, et0 - 2 * day, 8 * day, 3, ncoef)
    for c in range(3):
        moon.records[:, 2 + ncoef * c + ncoef - 1] = 500.0   # a LARGE last coefficient (synthetic table): dropping it moves the Moon by up to 500 km
    s_moon = al.add_segment(moon)
    al.add_body(nx.EARTH, ephem.MU_EARTH, ephem.R_EARTH, [])
    al.add_body(nx.SUN, ephem.MU_SUN, ephem.R_SUN, [(s_sun, +1)])
    al.add_body(nx.MOON, ephem.MU_MOON, ephem.R_MOON, [(s_moon, +1)])
    prop, _, central = leo_full_setup(degree=4)
    compiled = prop.compile(al, central)
    b = dispersed_leo_batch(70, seed=16)
    ctx = nx.GpuContext(compiled)
    dur = 2 * 3600 * nx.NS_PER_S
    out, st = ctx.propagate(b, dur)
    ref, rst = oracle_lib.propagate(compiled, b, dur, n_threads=os.cpu_count() or 1)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = pos_vel_errors(out, ref)
    assert dr.max() < 1e-6 and dv.max() < 1e-9, (dr.max(), dv.max())
    # ... and it matters at the level of this comparison: without it the oracle itself lands > 1 mm elsewhere
    for c in range(3):
        al.segments[s_moon].records[:, 2 + ncoef * c + ncoef - 1] = 0.0
    ref0, _ = oracle_lib.propagate(prop.compile(al, central), b, dur, n_threads=os.cpu_count() or 1)
    assert pos_vel_errors(ref0, ref)[0].max() > 1e-6
