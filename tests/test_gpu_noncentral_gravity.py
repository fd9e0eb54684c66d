"""-m gpu: the gravity field of a non-central body on the device (`nyx_hip_gravity_field_t.offset_body`, gravity_field.rs:150-154,
258-265) against the oracle, the field's effect against the central formulation of the same system, and the STM forms (round 4)."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
import noncentral_cases as nc
from scenarios import pos_vel_errors
from test_oracle_noncentral_gravity import check_field_effects, field_effects

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree,n,waves", [(20, 70, 0), (8, 5, 1), (70, 200, 16)])
def test_device_vs_oracle(degree, n, waves):
    """Earth-centred integration, Moon / Sun point masses, the Moon's field at r - r_moon(t); states Moon-centred (integration-frame
    swap) and Earth-centred; one wave, the default shape and sixteen column waves (plain two-barrier loop: the field's inputs need
    the Moon's position of the stage)."""
    prop, almanac, earth = nc.earth_centred(degree)
    b = nc.batch(n, seed=2)
    dur = 2 * 3600 * nx.NS_PER_S
    for state_frame in (nc.MOON_FRAME, None):
        compiled = prop.compile(almanac, earth, state_frame=state_frame)
        bb = b
        if state_frame is None:  # the same orbits given Earth-centred: translated by hand (numpy's Chebyshev value and derivative)
            import frame_swap_cases as fs
            rv = b.rv().copy()
            for i in range(b.n):
                r, v = fs.chain_state_numpy(almanac, nx.MOON, int(b.epoch_ns[i]))
                rv[i, :3] += r
                rv[i, 3:] += v
            bb = b.copy()
            bb.set_rv(rv)
        ctx = nx.GpuContext(compiled)
        if waves:
            ctx.set_column_waves(waves)
        out, st = ctx.propagate(bb, dur)
        ref, rst = oracle_lib.propagate(compiled, bb, dur, n_threads=8)
        assert (st.status == 0).all() and (rst.status == 0).all()
        # (no step-for-step comparison here: a 50 km lunar orbit integrated at 384 000 km from the origin has its error estimate in the
        #  rounding noise of the position - 600 steps where the Moon-centred formulation takes 130 - and two correct implementations
        #  accept different steps; both stay within the tolerance of the same solution)
        dr, dv = pos_vel_errors(out, ref)
        print(f"deg {degree} n {n} waves {waves} state frame {'Moon' if state_frame else 'Earth'}: dr {dr.max() * 1e3:.3e} m dv {dv.max() * 1e6:.3e} mm/s, "
              f"helpers {ctx.last_coop_helpers()}")
        assert dr.max() < 1e-6 and dv.max() < 1e-9 and ctx.last_coop_helpers() == 0  # (measured: 0.08-0.18 mm, 0.08-0.15 um/s)
        ctx.close()


def test_same_field_effect_as_the_moon_centred_formulation_on_the_device():
    def run(compiled, b, dur):
        ctx = nx.GpuContext(compiled)
        out, st = ctx.propagate(b, dur)
        ctx.close()
        return out, st
    check_field_effects(*field_effects(run, n=70))


def _stm_batch(b):
    b = b.copy()
    b.stm = np.zeros((b.n, 81))
    b.reset_stm()
    return b


def _earth_centred(b, almanac):
    import frame_swap_cases as fs
    rv = b.rv().copy()
    for i in range(b.n):
        r, v = fs.chain_state_numpy(almanac, nx.MOON, int(b.epoch_ns[i]))
        rv[i, :3] += r
        rv[i, 3:] += v
    b = b.copy()
    b.set_rv(rv)
    return b


def _phi_err(out, ref):
    scale = np.maximum(np.abs(ref.stm), 1e-6 * np.abs(ref.stm).max(axis=1, keepdims=True))
    return (np.abs(out.stm - ref.stm) / scale).max()


@pytest.mark.parametrize("layout", [0, 1])
def test_stm_with_a_non_central_field(layout):
    """Round 4: GravityField::gradient is as frame-agnostic as eom (gravity_field.rs:279-283: transform_to
    translates to the field's body, the duals are seeded on that radius, the gradient is rotated back) - the Moon's 20x20 field in an
    EARTH-centred STM propagation, both STM layouts (64-lane duals, quad), in the OD pattern (1-minute segments from an identity
    Phi: one attempt at the same step on both sides, so Phi is comparable to 1e-9) and over ten minutes with fixed steps."""
    prop, almanac, earth = nc.earth_centred(20)
    b = _stm_batch(_earth_centred(nc.batch(37, seed=6), almanac))
    compiled = prop.compile(almanac, earth, stm=True)
    ctx = nx.GpuContext(compiled)
    ctx.set_stm_layout(layout)
    out, st = ctx.propagate(b, 60 * nx.NS_PER_S)
    ref, rst = oracle_lib.propagate(compiled, b, 60 * nx.NS_PER_S, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = pos_vel_errors(out, ref)
    e = _phi_err(out, ref)
    print(f"non-central field, STM layout {layout}: 1-min segment dr {dr.max() * 1e3:.3e} m, Phi rel err {e:.3e}")
    assert dr.max() < 1e-6 and dv.max() < 1e-9 and e < 1e-9
    np.testing.assert_array_equal(st.n_accepted, rst.n_accepted)
    ctx.close()
    # the field's gradient is in Phi: the same segment without the field differs in the velocity rows by far more than the agreement
    bare, _, _ = nc.earth_centred(0)
    cb = bare.compile(almanac, earth, stm=True)
    ref0, _ = oracle_lib.propagate(cb, b, 60 * nx.NS_PER_S, n_threads=8)
    assert np.abs(ref.stm - ref0.stm).max() > 1e3 * np.abs(out.stm - ref.stm).max()
    # fixed 30 s steps, ten minutes
    fixed = nx.Propagator(prop.dynamics, prop.method, nx.IntegratorOptions.with_fixed_step_s(30.0))
    cf = fixed.compile(almanac, earth, stm=True)
    ctx = nx.GpuContext(cf)
    ctx.set_stm_layout(layout)
    out, st = ctx.propagate(b, 600 * nx.NS_PER_S)
    ref, rst = oracle_lib.propagate(cf, b, 600 * nx.NS_PER_S, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all() and _phi_err(out, ref) < 1e-9
    # frame-independent check: the STM run follows the orbit of the plain run
    plain = nx.GpuContext(fixed.compile(almanac, earth))
    pout, _ = plain.propagate(b, 600 * nx.NS_PER_S)
    dr, dv = pos_vel_errors(out, pout)
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    plain.close()
    ctx.close()


@pytest.mark.parametrize("centre,layout", [("earth", 0), ("moon", 0), ("earth", 1), ("moon", 1)])
def test_stm_with_two_stacked_fields(centre, layout):
    """Earth 21x21 + Moon 20x20 in one OrbitalDynamics with the STM, around either body (the cislunar OD set-up): sixty 1-minute
    segments with Phi reset (od/process/mod.rs:466-483) against the oracle's twin - states to 1 mm, Phi element-wise to 1e-9 - and the
    second field's gradient visibly in Phi.  Both layouts: 64 lanes x three-partial duals, and (round 5) the quad layout."""
    prop, almanac, frame = nc.two_fields(centre, 21, 20)
    b = nc.batch(70, seed=8)
    if centre == "earth":
        b = _earth_centred(b, almanac)
    b = _stm_batch(b)
    compiled = prop.compile(almanac, frame, stm=True)
    ctx = nx.GpuContext(compiled)
    ctx.set_stm_layout(layout)
    cur_d, cur_o = b, b
    worst = 0.0
    for seg in range(60):
        out, st = ctx.propagate(cur_d, 60 * nx.NS_PER_S)
        ref, rst = oracle_lib.propagate(compiled, cur_o, 60 * nx.NS_PER_S, n_threads=8)
        assert (st.status == 0).all() and (rst.status == 0).all()
        worst = max(worst, _phi_err(out, ref))
        cur_d, cur_o = out.copy(), ref.copy()
        cur_d.reset_stm()
        cur_o.reset_stm()
    dr, dv = pos_vel_errors(cur_d, cur_o)
    print(f"{centre}-centred, two fields, STM, 60 one-minute segments: dr {dr.max() * 1e3:.3e} m dv {dv.max() * 1e6:.3e} mm/s, worst Phi rel err {worst:.3e}")
    assert dr.max() < 1e-6 and dv.max() < 1e-9 and worst < 1e-9
    # what the second field contributes to Phi: the same segment with the larger field alone
    fields = [m for m in prop.dynamics.orbital_dyn.accel_models if isinstance(m, nx.GravityFieldData)]
    small = min(fields, key=lambda f: f.degree)
    one = nx.Propagator(nx.SpacecraftDynamics(nx.OrbitalDynamics([m for m in prop.dynamics.orbital_dyn.accel_models if m is not small]), []), prop.method, prop.opts)
    seg1, _ = ctx.propagate(b, 60 * nx.NS_PER_S)
    ref1, _ = oracle_lib.propagate(compiled, b, 60 * nx.NS_PER_S, n_threads=8)
    alone, _ = oracle_lib.propagate(one.compile(almanac, frame, stm=True), b, 60 * nx.NS_PER_S, n_threads=8)
    assert np.abs(ref1.stm - alone.stm).max() > 30 * np.abs(seg1.stm - ref1.stm).max()
    # and the orbit of the STM run is the plain run's
    plain = nx.GpuContext(prop.compile(almanac, frame))
    pout, _ = plain.propagate(b, 60 * nx.NS_PER_S)
    dr, dv = pos_vel_errors(seg1, pout)
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    plain.close()
    ctx.close()


@pytest.mark.parametrize("centre,deg_earth,deg_moon,n,waves", [("earth", 21, 20, 70, 0), ("moon", 8, 70, 130, 16), ("moon", 6, 70, 1100, 0),
                                                               ("earth", 12, 10, 5, 1)])
def test_two_stacked_fields_vs_oracle(centre, deg_earth, deg_moon, n, waves):
    """The Earth's and the Moon's field in one OrbitalDynamics (`config.gravity2`): the larger one on the column waves, the other walked
    by the perturbation wave; around either body, one wave to sixteen, stand-alone and cooperative (1 100 low lunar orbits, 70x70 +
    the Earth's 6x6 as the non-central second field)."""
    prop, almanac, frame = nc.two_fields(centre, deg_earth, deg_moon)
    compiled = prop.compile(almanac, frame)
    b = nc.batch(n, seed=4)
    if centre == "earth":  # the lunar orbits, Earth-centred
        import frame_swap_cases as fs
        rv = b.rv().copy()
        for i in range(b.n):
            r, v = fs.chain_state_numpy(almanac, nx.MOON, int(b.epoch_ns[i]))
            rv[i, :3] += r
            rv[i, 3:] += v
        b.set_rv(rv)
    dur = 2 * 3600 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled)
    if waves:
        ctx.set_column_waves(waves)
    out, st = ctx.propagate(b, dur)
    helpers = ctx.last_coop_helpers()
    ctx.close()
    sample = np.arange(0, n, max(1, n // 64))
    ref, rst = oracle_lib.propagate(compiled, b.take(sample), dur, n_threads=16)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = pos_vel_errors(out.take(sample), ref)
    # what the second field does: the same run without it
    fields = [m for m in prop.dynamics.orbital_dyn.accel_models if isinstance(m, nx.GravityFieldData)]
    small = min(fields, key=lambda f: f.degree)
    models = [m for m in prop.dynamics.orbital_dyn.accel_models if m is not small]
    one = nx.Propagator(nx.SpacecraftDynamics(nx.OrbitalDynamics(models), []), prop.method, prop.opts)
    without, _ = oracle_lib.propagate(one.compile(almanac, frame), b.take(sample), dur, n_threads=16)
    effect = np.linalg.norm((ref.rv() - without.rv())[:, :3], axis=1).max()
    print(f"{centre}-centred, Earth {deg_earth} + Moon {deg_moon}, n {n}, waves {waves or 'auto'}, helpers {helpers}: dr {dr.max() * 1e3:.3e} m "
          f"dv {dv.max() * 1e6:.3e} mm/s; the second field moves the orbit by {effect * 1e3:.3e} m")
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    assert effect > 20 * dr.max()          # the term is there (and far above the agreement)
    # cooperative launches: sixteen-wave workgroups of a central first field that leave CUs idle - since round 6 (fan-out mode: dedicated
    # helpers for small batches) whatever the batch size; the eight-wave shapes and non-central first fields never
    assert (helpers > 0) == (n >= 1000 or (waves == 16 and centre == "moon"))
    if centre == "moon":
        # step for step (round 4): around the Moon the error estimate is not rounding noise (130 steps per 2 h), so the device and the
        # oracle must make the SAME accept / reject decisions - equal accepted / rejected / evaluation counts (the step sizes differ in the
        # fourth digit: the two sum the harmonics in different orders and the error estimate is a difference of nearly equal numbers)
        same = (st.n_accepted[sample] == rst.n_accepted) & (st.n_rejected[sample] == rst.n_rejected) & (st.n_evals[sample] == rst.n_evals)
        rel = np.abs(out.take(sample).step_ns - ref.step_ns) / np.abs(ref.step_ns)
        print(f"   step decisions equal for {same.sum()} of {len(sample)} sampled trajectories; accepted {int(rst.n_accepted.min())}-{int(rst.n_accepted.max())}, "
              f"rejected up to {int(rst.n_rejected.max())}; last step sizes agree to {rel.max():.1e}")
        assert same.all() and rel.max() < 5e-3   # (the NEXT step is 0.9 h (tol / err)^(1/9): the error estimate, ~1e-13, carries the summation order in its third digit)


@pytest.mark.parametrize("stm", [False, True])
def test_second_field_orientation_leaving_its_coverage_is_reported(stm):
    """the second field's orientation is evaluated by the perturbation wave, and a binary-PCK orientation whose
    coverage the epoch leaves used to be clamped silently there (the first field and the bodies need not share that segment).  The
    Moon's field as the SECOND field of an Earth-centred run, its orientation a Chebyshev Euler-angle segment covering ONE day: inside
    the coverage device == oracle, beyond it every trajectory ends with NYX_HIP_ERR_EPHEM_RANGE on both sides."""
    from nyx_amd import _abi
    from rotation_cases import euler_rotation_like
    from scenarios import EPOCH0_NS, JGM3_PATH, iau_earth_frame, almanac_earth, earth_frame, kaula_field
    from nyx_amd import ephem
    moon_bpc = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, euler_rotation_like(nx.IAU_MOON_ROTATION_POLY, nx.to_seconds(EPOCH0_NS) - 600.0, 1.0))
    earth_field = nx.GravityFieldData.from_packed_file(JGM3_PATH, iau_earth_frame(), 12, 12)
    moon_field = kaula_field(8, seed=1, frame=moon_bpc)
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses([nx.MOON, nx.SUN]), earth_field, moon_field]), [])
    prop = nx.Propagator(dyn, nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions())
    almanac, frame = almanac_earth(), earth_frame(ephem.MU_EARTH)
    compiled = prop.compile(almanac, frame, stm=stm)
    b = _earth_centred(nc.batch(5, seed=3), almanac)
    if stm:
        b = _stm_batch(b)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(b, 3600 * nx.NS_PER_S)
    ref, rst = oracle_lib.propagate(compiled, b, 3600 * nx.NS_PER_S, n_threads=4)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = pos_vel_errors(out, ref)
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    out2, st2 = ctx.propagate(b, 25 * 3600 * nx.NS_PER_S)
    _, rst2 = oracle_lib.propagate(compiled, b, 25 * 3600 * nx.NS_PER_S, n_threads=8)
    assert (rst2.status == _abi.ERR_EPHEM_RANGE).all()
    assert (st2.status == _abi.ERR_EPHEM_RANGE).all()
    ctx.close()


def test_quad_and_64_lane_layouts_agree_bit_for_bit_with_a_second_field():
    """The quad layout's second-field gradient is the 64-lane layout's expression per partial slot (second_field_into_pert_q): with
    the column split of the first field fixed the two layouts return the same bits."""
    prop, almanac, frame = nc.two_fields("moon", 8, 12)
    b = _stm_batch(nc.batch(40, seed=3))
    compiled = prop.compile(almanac, frame, stm=True)
    outs = []
    for layout in (0, 1):
        ctx = nx.GpuContext(compiled)
        ctx.set_stm_layout(layout)
        ctx.set_column_waves(4)
        out, st = ctx.propagate(b, 600 * nx.NS_PER_S)
        assert (st.status == 0).all()
        outs.append(out)
        ctx.close()
    np.testing.assert_array_equal(outs[0].rv(), outs[1].rv())
    np.testing.assert_array_equal(outs[0].stm, outs[1].stm)
