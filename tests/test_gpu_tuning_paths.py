"""-m gpu: tuning fields that re-deal the columns must never lose one.  Round 5 found `tuning.role_duties` (explicit duties of the role
waves) handing the INTEGRATOR wave of a pipelined workgroup a share of the columns - which that wave never walks: the run finished
with status OK and 57 of 2 556 rows of the 70x70 field missing from every acceleration."""
import os

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from scenarios import dispersed_leo_batch, leo_full_setup, pos_vel_errors

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S


@pytest.mark.parametrize("tuning", [dict(role_duties=[60.0, 600.0, 52.0]), dict(role_duties=[60.0, 600.0, 500.0], cooperative=0),
                                    dict(coop_fraction=0.40), dict(coop_fraction=0.25, harmonics_feed=1),
                                    # the owner's column runs: free wave order (default since round 5) and the linear partition of round 4,
                                    # at the default share and at shares that move every target
                                    dict(), dict(debug_flags=0x2000000), dict(coop_fraction=0.33), dict(coop_fraction=0.36, debug_flags=0x2000000),
                                    dict(schedule=nx.SCHED_EXPLICIT, wave_weights=[1.0, 1.2, 0.7, 1.7, 1.9, 1.3, 1.5, 1.1, 1.0, 0.9, 0.8, 0.7, 0.5, 0.4, 0.3, 0.2])])
def test_every_column_is_walked_whatever_the_tuning(tuning):
    prop, almanac, central = leo_full_setup(degree=70)
    compiled = prop.compile(almanac, central)
    batch = dispersed_leo_batch(640, seed=17)       # ten workgroups: cooperative mode where the tuning leaves it on
    dur = 1200 * S
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(**tuning))
    out, st = ctx.propagate(batch, dur)
    ctx.close()
    assert (st.status == 0).all()
    sub = batch.take(np.arange(0, 640, 10))
    ref, rst = oracle_lib.propagate(compiled, sub, dur, n_threads=os.cpu_count() or 1)
    assert (rst.status == 0).all()
    d = out.rv()[::10] - ref.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()
    assert dr < 1e-3 and dv < 1e-6, (tuning, dr, dv)   # (a dropped column is tens of metres after twenty minutes)


def test_free_order_partition_deals_every_row_once():
    """The rows the owner's schedule holds (nyx_hip_debug_schedule_rows) are the same total under both partitions of its column runs,
    and owner + helper rows are the whole 70x70 table (2 556 rows)."""
    import ctypes as C
    prop, almanac, central = leo_full_setup(degree=70)
    compiled = prop.compile(almanac, central)
    batch = dispersed_leo_batch(640, seed=17)
    totals = []
    for flags in (0x8000000, 0x8000000 | 0x2000000):   # (0x8000000: the claim mode, whose owner share the two partitions deal; the fan-out mode below)
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(debug_flags=flags))
        ctx.propagate(batch, 60 * S)
        assert ctx.last_coop_helpers() > 0
        rows = (C.c_int32 * 16)()
        ctx._lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        per = []
        for sched in (1, 2):
            assert ctx._lib.nyx_hip_debug_schedule_rows(ctx._h, sched, rows, None) == 0
            per.append(list(rows[:]))
        ctx.close()
        assert per[0][0] == 0                       # the integrator wave of a pipelined workgroup walks no columns
        assert sum(per[0]) + sum(per[1]) == 2556, per
        totals.append(per[0])
    assert totals[0] != totals[1]                   # (the two partitions do differ on this shape)


@pytest.mark.parametrize("degree", [40, 47, 56, 64, 83, 95])
def test_free_order_partition_at_other_degrees(degree):
    """Every cooperative shape that streams the table in its owners (degree 40 ... 95) goes through the depth-first placement: the
    context is built in bounded time, both partitions deal the same rows in total, and the run stays on the oracle."""
    import ctypes as C
    import time
    prop, almanac, central = leo_full_setup(degree=degree)
    compiled = prop.compile(almanac, central)
    batch = dispersed_leo_batch(640, seed=23)
    dur = 600 * S
    sums, outs = [], []
    for flags in (0, 0x2000000):
        t0 = time.time()
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(debug_flags=flags))
        out, st = ctx.propagate(batch, dur)
        assert time.time() - t0 < 20.0
        assert (st.status == 0).all()
        rows = (C.c_int32 * 16)()
        ctx._lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        tot = 0
        for sched in (1, 2):
            assert ctx._lib.nyx_hip_debug_schedule_rows(ctx._h, sched, rows, None) == 0
            tot += sum(rows[:])
        helpers = ctx.last_coop_helpers()
        ctx.close()
        sums.append((tot, helpers))
        outs.append(out.rv())
    assert sums[0] == sums[1], sums
    sub = batch.take(np.arange(0, 640, 40))
    ref, rst = oracle_lib.propagate(compiled, sub, dur, n_threads=os.cpu_count() or 1)
    for o in outs:
        d = o[::40] - ref.rv()
        assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-3 and np.linalg.norm(d[:, 3:], axis=1).max() < 1e-6


@pytest.mark.parametrize("degree, n", [(70, 640), (30, 70), (24, 200), (95, 1280)])
def test_fan_out_schedules_deal_every_row_once(degree, n):
    """Fan-out mode (round 6): the owner's rows (schedule 1) and those of the dedicated helpers' parts (schedules 5 ...) are the whole
    table whatever the field's size - a field below degree 40 leaves the owner two or three rows, which the two-ended fill once handed to
    the integrator wave (which walks none) - and no part is dealt to the integrator wave."""
    import ctypes as C
    from scenarios import kaula_field
    prop, almanac, central = leo_full_setup(degree=degree if degree == 70 else 0)
    if degree != 70:
        from scenarios import iau_earth_frame
        prop = nx.Propagator(nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses([nx.SUN, nx.MOON]), kaula_field(degree, seed=3, frame=iau_earth_frame())]), []),
                             prop.method, prop.opts)
    compiled = prop.compile(almanac, central)
    batch = dispersed_leo_batch(n, seed=17)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(batch, 600 * S)
    assert (st.status == 0).all() and ctx.last_coop_helpers() > 0
    rows = (C.c_int32 * 16)()
    ctx._lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    tot = 0
    for sched in [1] + list(range(5, 13)):
        assert ctx._lib.nyx_hip_debug_schedule_rows(ctx._h, sched, rows, None) == 0
        assert rows[0] == 0 and (sched == 1 or rows[15] == 0)      # the integrator wave / a helper's answering wave walk nothing
        tot += sum(rows[:])
    ctx.close()
    assert tot == (degree + 1) * (degree + 2) // 2, (degree, tot)
    sub = batch.take(np.arange(0, n, max(1, n // 16))[:16])
    ref, rst = oracle_lib.propagate(compiled, sub, 600 * S, n_threads=os.cpu_count() or 1)
    d = out.rv()[np.arange(0, n, max(1, n // 16))[:16]] - ref.rv()
    assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-3
