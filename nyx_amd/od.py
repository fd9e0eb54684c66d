"""Covariance mapping front end: the prediction-only part of the reference's Kalman OD process.

Mirror of `KalmanODProcess::predict_until / predict_for` (nyx-core/src/od/process/mod.rs:440-499) with
`KalmanFilter::time_update` (od/kalman/filtering.rs:59-99) and `ProcessNoise3D` (od/snc.rs:40-309), batched over
many estimates: every 1-step segment, its time update and the STM reset run on the device
(`nyx_hip_predict_until`); the host only stages the inputs and collects the estimates.

Not mirrored (host-side OD machinery, out of scope): measurement updates, smoothing, residual rejection.  Process-noise
decay and the local-frame (RIC / VNC) definition of the noise (ProcessNoise::with_decay, local_frame) are on the device.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi
from .propagator import Almanac, GpuContext, Propagator, Spacecraft, pack_spacecraft

INT64_MAX = (1 << 63) - 1


@dataclass
class ProcessNoise3D:
    """od/snc.rs:40-59 (inertial, constant)."""

    diag: Sequence[float]
    disable_time_ns: int
    start_time_ns: Optional[int] = None
    local_frame: Optional[str] = None          # None (inertial), "RIC" or "VNC" (snc.rs:46-47, 219-239)
    decay_s: Optional[Sequence[float]] = None  # with_decay (snc.rs:145-160)
    init_epoch_ns: Optional[int] = None        # None = each estimate's own initial epoch (kalman/initializers.rs:75-101)

    @classmethod
    def with_decay(cls, values: Sequence[float], disable_time_ns: int, decay_constants_s: Sequence[float], local_frame: Optional[str] = None):
        """snc.rs:145-160."""
        assert len(decay_constants_s) == 3, "Not enough decay constants for the size of the SNC matrix"
        me = cls.from_diagonal(values, disable_time_ns)
        me.decay_s, me.local_frame = [float(v) for v in decay_constants_s], local_frame
        return me

    @classmethod
    def from_diagonal(cls, values: Sequence[float], disable_time_ns: int, local_frame: Optional[str] = None):
        """snc.rs:108-135."""
        assert len(values) == 3, "Not enough values for the size of the SNC matrix"
        return cls([float(v) for v in values], int(disable_time_ns), local_frame=local_frame)

    @classmethod
    def with_start_time(cls, disable_time_ns: int, values: Sequence[float], start_time_ns: int):
        """snc.rs:138-142."""
        me = cls.from_diagonal(values, disable_time_ns)
        me.start_time_ns = int(start_time_ns)
        return me

    @classmethod
    def from_velocity_km_s(cls, velocity_noise: Sequence[float], noise_duration_ns: int, disable_time_ns: int):
        """snc.rs:288-309: diag = velocity noise / noise duration (seconds)."""
        from .propagator import to_seconds
        return cls([float(v) / to_seconds(int(noise_duration_ns)) for v in velocity_noise], int(disable_time_ns))


@dataclass
class Predicted:
    """What `ODSolution.estimates` holds after a `predict_until` (time updates only), for the whole batch."""

    states: _abi.StateBatch              # last nominal states
    stats: _abi.StatsBatch               # status = first failure per trajectory; counters summed over the segments
    covar: np.ndarray                    # [n, 9, 9] covar_bar of the last time update
    state_deviation: np.ndarray          # [n, 9]
    n_updates: np.ndarray                # [n]
    epochs_ns: Optional[np.ndarray] = None   # history, [updates, n]
    nominal: Optional[np.ndarray] = None     # [updates, n, 9]
    stm: Optional[np.ndarray] = None         # [updates, n, 9, 9]
    covar_history: Optional[np.ndarray] = None  # [updates, n, 9, 9]
    deviation_history: Optional[np.ndarray] = None
    kernel_ms: float = -1.0


def _col_major(m: np.ndarray) -> np.ndarray:
    """[n, 9, 9] (row, col) -> the ABI's per-trajectory column-major 81-vectors."""
    return np.ascontiguousarray(np.transpose(m, (0, 2, 1)).reshape(m.shape[0], 81))


def _from_col_major(v: np.ndarray) -> np.ndarray:
    return np.transpose(v.reshape(v.shape[:-1] + (9, 9)), tuple(range(v.ndim - 1)) + (v.ndim, v.ndim - 1))


def build_predict(max_step_ns: int, end_epoch_ns: int, process_noise: Sequence[ProcessNoise3D] = (),
                  deviation_tracking: bool = False) -> _abi.Predict:
    if len(process_noise) > _abi.MAX_PROCESS_NOISE:
        raise ValueError(f"at most {_abi.MAX_PROCESS_NOISE} process noises")
    pc = _abi.Predict()
    pc.max_step_ns, pc.end_epoch_ns = int(max_step_ns), int(end_epoch_ns)
    pc.deviation_tracking = 1 if deviation_tracking else 0
    pc.n_process_noise = len(process_noise)
    for k, pn in enumerate(process_noise):
        for j in range(3):
            pc.process_noise[k].diag[j] = float(pn.diag[j])
        pc.process_noise[k].disable_time_ns = int(pn.disable_time_ns)
        pc.process_noise[k].has_start_time = 0 if pn.start_time_ns is None else 1
        pc.process_noise[k].start_time_ns = 0 if pn.start_time_ns is None else int(pn.start_time_ns)
        lf = {None: _abi.FRAME_INERTIAL, "RIC": _abi.FRAME_RIC, "VNC": _abi.FRAME_VNC}.get(pn.local_frame, -1)
        if lf < 0:
            raise NotImplementedError(f"process noise in the {pn.local_frame} frame is not on the device path")
        pc.process_noise[k].local_frame = lf
        pc.process_noise[k].has_decay = 0 if pn.decay_s is None else 1
        for j in range(3):
            pc.process_noise[k].decay_s[j] = 0.0 if pn.decay_s is None else float(pn.decay_s[j])
        pc.process_noise[k].init_epoch_ns = -(1 << 63) if pn.init_epoch_ns is None else int(pn.init_epoch_ns)
    return pc


def predict_until(ctx: GpuContext, batch: _abi.StateBatch, covar: np.ndarray, end_epoch_ns: int, max_step_ns: int,
                  process_noise: Sequence[ProcessNoise3D] = (), deviation_tracking: bool = False,
                  state_deviation: Optional[np.ndarray] = None, history: int = 0, keep_stm: bool = True,
                  _call=None) -> Predicted:
    """`predict_until` for every (nominal state, covariance) pair of the batch.  `ctx` must be an STM context.
    `history` = number of time updates to keep per trajectory (0: only the final estimate)."""
    n = batch.n
    pc = build_predict(max_step_ns, end_epoch_ns, process_noise, deviation_tracking)
    cv = _col_major(np.asarray(covar, dtype=np.float64).reshape(n, 9, 9))
    dev = np.zeros((n, 9)) if state_deviation is None else np.ascontiguousarray(state_deviation, dtype=np.float64).reshape(n, 9).copy()
    est = _abi.Estimates(cv.ctypes.data_as(_abi.c_double_p), dev.ctypes.data_as(_abi.c_double_p))
    out = batch.copy()
    if out.stm is None:
        out.stm = np.zeros((n, 81))
    stats = _abi.StatsBatch(n)
    n_up = np.zeros(n, dtype=np.int32)
    h = _abi.PredictHistory()
    h.capacity = int(history)
    h.n_updates = n_up.ctypes.data_as(_abi.c_int32_p)
    arrays = {}
    if history > 0:
        arrays = dict(epoch=np.zeros((history, n), dtype=np.int64), state=np.zeros((history, n, 9)),
                      covar=np.zeros((history, n, 81)), dev=np.zeros((history, n, 9)))
        h.epoch_ns = arrays["epoch"].ctypes.data_as(_abi.c_int64_p)
        h.state = arrays["state"].ctypes.data_as(_abi.c_double_p)
        h.covar = arrays["covar"].ctypes.data_as(_abi.c_double_p)
        h.state_dev = arrays["dev"].ctypes.data_as(_abi.c_double_p)
        if keep_stm:
            arrays["stm"] = np.zeros((history, n, 81))
            h.stm = arrays["stm"].ctypes.data_as(_abi.c_double_p)
    cin, cout, cst = batch.as_c(), out.as_c(), stats.as_c()
    if _call is None:
        rc = ctx._lib.nyx_hip_predict_until(ctx._h, C.byref(cin), C.byref(pc), C.byref(est), C.byref(cout), C.byref(cst), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_predict_until failed (rc={rc}): {_abi.last_error()}")
        ms = ctx.last_kernel_ms()
    else:  # the oracle twin (tests)
        rc = _call(C.byref(cin), C.byref(pc), C.byref(est), C.byref(cout), C.byref(cst), C.byref(h))
        assert rc == 0
        ms = -1.0
    res = Predicted(out, stats, _from_col_major(cv), dev, n_up, kernel_ms=ms)
    if history > 0:
        res.epochs_ns, res.nominal = arrays["epoch"], arrays["state"]
        res.covar_history, res.deviation_history = _from_col_major(arrays["covar"]), arrays["dev"]
        if keep_stm:
            res.stm = _from_col_major(arrays["stm"])
    return res


@dataclass
class KalmanODProcess:
    """The prediction-only face of `KalmanODProcess` (od/process/mod.rs:60-140 builder fields used by predict_until)."""

    prop: Propagator
    almanac: Almanac
    max_step_ns: int = 60 * 10**9
    process_noise: List[ProcessNoise3D] = field(default_factory=list)
    deviation_tracking: bool = False

    def predict_until(self, spacecraft: Sequence[Spacecraft], covar: np.ndarray, end_epoch_ns: int, history: int = 0) -> Predicted:
        ctx = self.prop._context(self.almanac, spacecraft[0].frame, True)
        batch = pack_spacecraft(list(spacecraft), True)
        return predict_until(ctx, batch, covar, end_epoch_ns, self.max_step_ns, self.process_noise, self.deviation_tracking,
                             history=history)

    def predict_for(self, spacecraft: Sequence[Spacecraft], covar: np.ndarray, duration_ns: int, history: int = 0) -> Predicted:
        """mod.rs:489-498: end epoch = epoch of the (first) nominal state + duration."""
        return self.predict_until(spacecraft, covar, int(spacecraft[0].epoch_ns) + int(duration_ns), history)
