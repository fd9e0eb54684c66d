"""Loader of the CPU oracle (oracle/libnyx_oracle.so).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

from nyx_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libnyx_oracle.so")
    src = os.path.join(ORACLE_DIR, "nyx_oracle.c")
    if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        build()
    lib = C.CDLL(path)
    lib.nyx_oracle_propagate_batch.argtypes = [C.POINTER(_abi.Config), C.POINTER(_abi.States), C.c_int64,
                                               C.POINTER(_abi.States), C.POINTER(_abi.StepStats), C.c_int32]
    lib.nyx_oracle_propagate_batch.restype = C.c_int32
    lib.nyx_oracle_propagate_batch_traj.argtypes = [C.POINTER(_abi.Config), C.POINTER(_abi.States), C.c_int64, C.POINTER(_abi.States),
                                                    C.POINTER(_abi.StepStats), C.POINTER(_abi.Traj), C.c_int32]
    lib.nyx_oracle_propagate_batch_traj.restype = C.c_int32
    lib.nyx_oracle_eom.argtypes = [C.POINTER(_abi.Config), C.c_int64, C.c_double, _abi.c_double_p, _abi.c_double_p,
                                   C.c_double, C.c_double, C.c_double, C.c_double, _abi.c_double_p]
    lib.nyx_oracle_eom.restype = C.c_int32
    lib.nyx_oracle_dual_eom.argtypes = [C.POINTER(_abi.Config), C.c_int64, _abi.c_double_p, C.c_double, C.c_double,
                                        C.c_double, _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_dual_eom.restype = C.c_int32
    lib.nyx_oracle_body_position.argtypes = [C.POINTER(_abi.Config), C.c_int32, C.c_int64, _abi.c_double_p, _abi.c_int32_p]
    lib.nyx_oracle_rotation_dcm.argtypes = [C.POINTER(_abi.Rotation), C.POINTER(_abi.ChebySegment), C.c_int64, _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_rotation_dcm.restype = C.c_int32
    lib.nyx_oracle_gravity_accel.argtypes = [C.POINTER(_abi.GravityField), C.c_int64, _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_occultation_factor.argtypes = [C.POINTER(_abi.Config), C.c_int32, C.c_int32, C.c_int64, _abi.c_double_p, _abi.c_int32_p]
    lib.nyx_oracle_occultation_factor.restype = C.c_double
    lib.nyx_oracle_error_estimate.argtypes = [C.c_int32, C.c_int32, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_error_estimate.restype = C.c_double
    lib.nyx_oracle_seconds_to_ns.argtypes = [C.c_double]
    lib.nyx_oracle_seconds_to_ns.restype = C.c_int64
    lib.nyx_oracle_ns_to_seconds.argtypes = [C.c_int64]
    lib.nyx_oracle_ns_to_seconds.restype = C.c_double
    lib.nyx_oracle_set_ns_rounding.argtypes = [C.c_int32]
    lib.nyx_oracle_predict_until.argtypes = [C.POINTER(_abi.Config), C.POINTER(_abi.States), C.POINTER(_abi.Predict), C.POINTER(_abi.Estimates),
                                             C.POINTER(_abi.States), C.POINTER(_abi.StepStats), C.POINTER(_abi.PredictHistory)]
    lib.nyx_oracle_predict_until.restype = C.c_int32
    lib.nyx_oracle_tides_accel.argtypes = [C.POINTER(_abi.Config), C.c_int64, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p,
                                           _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_tides_accel.restype = C.c_int32
    lib.nyx_oracle_until_event.argtypes = [C.POINTER(_abi.Config), C.POINTER(_abi.States), C.c_int64, C.POINTER(_abi.EventC), C.POINTER(_abi.States),
                                           C.POINTER(_abi.StepStats), C.POINTER(_abi.Traj), _abi.c_int32_p]
    lib.nyx_oracle_until_event.restype = C.c_int32
    lib.nyx_oracle_hermite_eval.argtypes = [_abi.c_double_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_double,
                                            _abi.c_double_p, _abi.c_double_p]
    lib.nyx_oracle_hermite_eval.restype = C.c_int32
    lib.nyx_oracle_traj_at.argtypes = [C.POINTER(_abi.Traj), C.c_int64, C.c_int64, C.c_int64, _abi.c_double_p]
    lib.nyx_oracle_traj_at.restype = C.c_int32
    lib.nyx_oracle_traj_window_ill.argtypes = [C.POINTER(_abi.Traj), C.c_int64, C.c_int64, C.c_int64]
    lib.nyx_oracle_traj_window_ill.restype = C.c_int32
    lib.nyx_oracle_traj_every.argtypes = [C.POINTER(_abi.Traj), C.c_int64, C.c_int64, C.POINTER(_abi.Traj)]
    lib.nyx_oracle_traj_every.restype = C.c_int32
    _LIB = lib
    return lib


def propagate(compiled, batch, duration_ns, n_threads=1):
    """Oracle twin of GpuContext.propagate."""
    lib = load()
    out = batch.copy()
    stats = _abi.StatsBatch(batch.n)
    cin, cout, cst = batch.as_c(), out.as_c(), stats.as_c()
    rc = lib.nyx_oracle_propagate_batch(C.byref(compiled.cfg), C.byref(cin), int(duration_ns), C.byref(cout), C.byref(cst), n_threads)
    assert rc == 0
    return out, stats


def propagate_with_traj(compiled, batch, duration_ns, capacity, n_threads=1):
    lib = load()
    out = batch.copy()
    stats = _abi.StatsBatch(batch.n)
    traj = _abi.TrajBatch(batch.n, capacity)
    cin, cout, cst, ctr = batch.as_c(), out.as_c(), stats.as_c(), traj.as_c()
    rc = lib.nyx_oracle_propagate_batch_traj(C.byref(compiled.cfg), C.byref(cin), int(duration_ns), C.byref(cout), C.byref(cst), C.byref(ctr), n_threads)
    assert rc == 0
    return out, stats, traj


def eom(compiled, epoch_ns, dt_s, y, ctx_stm=None, dry=0.0, extra=0.0, srp_area=0.0, drag_area=0.0):
    lib = load()
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros_like(y)
    stm_p = _abi.c_double_p() if ctx_stm is None else np.ascontiguousarray(ctx_stm, dtype=np.float64).ctypes.data_as(_abi.c_double_p)
    st = lib.nyx_oracle_eom(C.byref(compiled.cfg), int(epoch_ns), float(dt_s), y.ctypes.data_as(_abi.c_double_p), stm_p,
                            dry, extra, srp_area, drag_area, out.ctypes.data_as(_abi.c_double_p))
    return st, out


def dual_eom(compiled, epoch_ns, y9, dry=0.0, extra=0.0, srp_area=0.0):
    lib = load()
    y9 = np.ascontiguousarray(y9, dtype=np.float64)
    fx, grad = np.zeros(9), np.zeros(81)
    st = lib.nyx_oracle_dual_eom(C.byref(compiled.cfg), int(epoch_ns), y9.ctypes.data_as(_abi.c_double_p), dry, extra, srp_area,
                                 fx.ctypes.data_as(_abi.c_double_p), grad.ctypes.data_as(_abi.c_double_p))
    return st, fx, grad.reshape(9, 9).T  # column-major -> [i, j]


def hermite_eval(xs, ys, ydots, x_eval):
    lib = load()
    xs, ys, ydots = (np.ascontiguousarray(a, dtype=np.float64) for a in (xs, ys, ydots))
    f, df = C.c_double(), C.c_double()
    st = lib.nyx_oracle_hermite_eval(xs.ctypes.data_as(_abi.c_double_p), ys.ctypes.data_as(_abi.c_double_p),
                                     ydots.ctypes.data_as(_abi.c_double_p), len(xs), float(x_eval), C.byref(f), C.byref(df))
    return st, f.value, df.value


def traj_at(traj, epochs_ns):
    """Oracle twin of GpuContext.traj_at: (states[m, n, 6], status[m, n])."""
    lib = load()
    ctr = traj.as_c()
    m = len(epochs_ns)
    out = np.full((m, traj.n, 6), np.nan)
    status = np.zeros((m, traj.n), dtype=np.int32)
    s6 = np.zeros(6)
    for q, e in enumerate(epochs_ns):
        for i in range(traj.n):
            status[q, i] = lib.nyx_oracle_traj_at(C.byref(ctr), traj.n, i, int(e), s6.ctypes.data_as(_abi.c_double_p))
            if status[q, i] == _abi.INTERP_OK and lib.nyx_oracle_traj_window_ill(C.byref(ctr), traj.n, i, int(e)):
                status[q, i] = _abi.INTERP_ILL_CONDITIONED   # (the C-ABI's warning on a produced sample)
            out[q, i] = s6
    return out, status


def traj_every(traj, step_ns, capacity):
    lib = load()
    out = _abi.TrajBatch(traj.n, capacity)
    cin, cout = traj.as_c(), out.as_c()
    rc = lib.nyx_oracle_traj_every(C.byref(cin), traj.n, int(step_ns), C.byref(cout))
    assert rc == 0
    return out


def predict_until(compiled, batch, covar, end_epoch_ns, max_step_ns, **kw):
    """Oracle twin of nyx_amd.od.predict_until (same arguments, same result object)."""
    from nyx_amd import od
    lib = load()
    return od.predict_until(None, batch, covar, end_epoch_ns, max_step_ns,
                            _call=lambda *a: lib.nyx_oracle_predict_until(C.byref(compiled.cfg), *a), **kw)


def tides_accel(compiled, epoch_ns, r3):
    """(accel[3], gradient[3, 3], delta_c[4, 4], delta_s[4, 4]) of the solid-tides model of `compiled`."""
    lib = load()
    r3 = np.ascontiguousarray(r3, dtype=np.float64)
    a, g, dc, ds = np.zeros(3), np.zeros(9), np.zeros(16), np.zeros(16)
    p = lambda x: x.ctypes.data_as(_abi.c_double_p)
    st = lib.nyx_oracle_tides_accel(C.byref(compiled.cfg), int(epoch_ns), p(r3), p(a), p(g), p(dc), p(ds))
    assert st == 0, st
    return a, g.reshape(3, 3), dc.reshape(4, 4), ds.reshape(4, 4)


def propagate_until_event(compiled, batch, max_duration_ns, event, trigger=1, capacity=4096):
    """Oracle twin of GpuContext.propagate_until_event."""
    lib = load()
    out = batch.copy()
    stats = _abi.StatsBatch(batch.n)
    traj = _abi.TrajBatch(batch.n, capacity)
    crossings = np.zeros(batch.n, dtype=np.int32)
    cin, cout, cst, ctr, cev = batch.as_c(), out.as_c(), stats.as_c(), traj.as_c(), event.as_c(trigger)
    rc = lib.nyx_oracle_until_event(C.byref(compiled.cfg), C.byref(cin), int(max_duration_ns), C.byref(cev), C.byref(cout), C.byref(cst),
                                    C.byref(ctr), crossings.ctypes.data_as(_abi.c_int32_p))
    assert rc == 0
    return out, stats, traj, crossings
