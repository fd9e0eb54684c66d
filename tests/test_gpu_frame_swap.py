"""opts.integration_frame on the device (nyx_hip_config_t.state_frame_body): Moon-centred states integrated in the Earth frame,
translated in and out by nyx_frame_shift_kernel - against the oracle twin, and the entry points that refuse a swap."""
import dataclasses

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from frame_swap_cases import MOON_FRAME, moon_batch, setup

pytestmark = pytest.mark.gpu


def test_swap_vs_oracle_and_refusals():
    prop, almanac, earth = setup()
    b = moon_batch(100, seed=3)
    dur = 2 * 3600 * nx.NS_PER_S
    compiled = prop.compile(almanac, earth, state_frame=MOON_FRAME)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(b, dur)
    ref, rst = oracle_lib.propagate(compiled, b, dur, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    np.testing.assert_array_equal(out.epoch_ns, ref.epoch_ns)
    d = out.rv() - ref.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()
    print(f"swap: dr {dr*1e3:.2e} m dv {dv*1e6:.2e} mm/s")
    assert dr < 1e-3 and dv < 1e-6                      # the path's parity bar: 1 m, 1 mm/s
    assert np.all(np.linalg.norm(out.rv()[:, :3], axis=1) < 2200.0)      # still Moon-centred
    # per-trajectory end epochs, same translation
    end = int(b.epoch_ns.max()) + 1800 * nx.NS_PER_S
    o2, s2 = ctx.propagate_until_epoch(b, end)
    assert (s2.status == 0).all() and (o2.epoch_ns == end).all()
    # what records or searches trajectories refuses the swap (the reference's trajectory there is in the integration frame)
    with pytest.raises(RuntimeError, match="integration-frame swap"):
        ctx.propagate_with_traj(b, dur, capacity=64)
    ctx.close()
    # outside the ephemeris: reported per run by the translation
    late = b.copy()
    late.epoch_ns[:] += 400 * 86400 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled)
    _, s3 = ctx.propagate(late, 600 * nx.NS_PER_S)
    assert (s3.status == nx._abi.ERR_EPHEM_RANGE).all()
    ctx.close()


def test_mirror_many_for_duration_with_integration_frame():
    prop, almanac, earth = setup()
    prop.opts = dataclasses.replace(prop.opts, integration_frame=earth)
    b = moon_batch(5, seed=8)
    scs = [nx.Spacecraft(int(b.epoch_ns[i]), b.rv()[i], MOON_FRAME, cr=1.5, dry_mass_kg=200.0, srp_area_m2=2.0) for i in range(5)]
    res = prop.many_for_duration(scs, almanac, 1800 * nx.NS_PER_S)
    assert len(res) == 5 and all(r.frame.naif_id == nx.MOON for r in res)
    ref, _ = oracle_lib.propagate(prop.compile(almanac, earth, state_frame=MOON_FRAME), b, 1800 * nx.NS_PER_S)
    got = np.array([np.asarray(r.rv) for r in res])
    assert np.abs(got - ref.rv()).max() < 1e-6
