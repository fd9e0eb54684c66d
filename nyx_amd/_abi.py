"""ctypes mirror of ``include/nyx_hip.h`` (the C-ABI of the batched propagation path).

Plain data only.  Field order/types must match the header exactly; ``tests/test_abi.py``
checks the struct sizes against the compiled library (``nyx_hip_abi_sizeof``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ABI_VERSION = 4
MAX_CHAIN = 4
MAX_BODIES = 8
MAX_SEGMENTS = 16

# enum nyx_hip_method  (reference: propagators/rk_methods/mod.rs:65-79)
RK89, DP78, DP45, RK4, CASHKARP45, VERNER56 = range(6)
# enum nyx_hip_error_ctrl  (reference: propagators/error_ctrl.rs:30-76)
RSS_CARTESIAN_STATE, RSS_CARTESIAN_STEP, RSS_STATE, RSS_STEP, LARGEST_ERROR, LARGEST_STATE, LARGEST_STEP = range(7)
# enum nyx_hip_status
OK, ERR_NAN, ERR_MASSLESS, ERR_FUEL_EXHAUSTED, ERR_EPHEM_RANGE, ERR_UNSUPPORTED, ERR_EVENT_NOT_FOUND, ERR_EVENT_SEARCH = range(8)
# return codes of the entry points (enum nyx_hip_rc)
RC_OK, RC_BAD_ARG, RC_NO_DEVICE, RC_HIP_ERROR, RC_UNSUPPORTED = range(5)
STATUS_NAMES = ["Ok", "PropMathError(NaN)", "MasslessSpacecraft", "FuelExhausted", "EphemerisOutOfRange", "Unsupported",
                "NthEventError", "EventSearchFailed"]
# enum nyx_hip_interp_status
INTERP_OK, INTERP_NO_DATA, INTERP_MATH, INTERP_ILL_CONDITIONED = range(4)


def interp_failed(status):
    """True where a `traj_at` sample was not produced (ILL_CONDITIONED is a warning on a produced sample)."""
    return (status != INTERP_OK) & (status != INTERP_ILL_CONDITIONED)
# flags
FLAG_STM = 0x1
FLAG_STM_TEXTBOOK = 0x2
# density
RHO_CONSTANT, RHO_EXPONENTIAL, RHO_STDATM = range(3)

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)


class IntegOpts(C.Structure):
    _fields_ = [
        ("init_step_ns", C.c_int64),
        ("min_step_ns", C.c_int64),
        ("max_step_ns", C.c_int64),
        ("tolerance", C.c_double),
        ("attempts", C.c_int32),
        ("fixed_step", C.c_int32),
        ("error_ctrl", C.c_int32),
        ("method", C.c_int32),
    ]


class ChebySegment(C.Structure):
    _fields_ = [
        ("init_et_s", C.c_double),
        ("interval_s", C.c_double),
        ("n_records", C.c_int32),
        ("n_coeffs", C.c_int32),
        ("records", c_double_p),
    ]


class Body(C.Structure):
    _fields_ = [
        ("naif_id", C.c_int32),
        ("n_chain", C.c_int32),
        ("chain_segment", C.c_int32 * MAX_CHAIN),
        ("chain_sign", C.c_int32 * MAX_CHAIN),
        ("mu_km3_s2", C.c_double),
        ("mean_radius_km", C.c_double),
    ]


MAX_NUT_PREC = 16
ROT_IAU, ROT_EULER_CHEBY = 0, 1


class Rotation(C.Structure):
    _fields_ = [("ra_deg", C.c_double * 3), ("dec_deg", C.c_double * 3), ("w_deg", C.c_double * 3),
                ("kind", C.c_int32), ("n_nut_prec", C.c_int32),
                ("nut_prec_angle_deg", (C.c_double * 2) * MAX_NUT_PREC),
                ("nut_prec_ra", C.c_double * MAX_NUT_PREC), ("nut_prec_dec", C.c_double * MAX_NUT_PREC), ("nut_prec_w", C.c_double * MAX_NUT_PREC),
                ("euler_segment", C.c_int32), ("_pad", C.c_int32), ("base_dcm", C.c_double * 9)]


class GravityField(C.Structure):
    _fields_ = [
        ("degree", C.c_int32),
        ("order", C.c_int32),
        ("offset_body", C.c_int32),   # 0 = the integration centre's field; k > 0 = the field of bodies[k - 1]
        ("_pad", C.c_int32),
        ("mu_km3_s2", C.c_double),
        ("eq_radius_km", C.c_double),
        ("c_nm", c_double_p),
        ("s_nm", c_double_p),
        ("rotation", Rotation),
    ]


class Srp(C.Structure):
    _fields_ = [
        ("phi_w_m2", C.c_double),
        ("estimate", C.c_int32),
        ("sun_body", C.c_int32),
        ("n_shadow_bodies", C.c_int32),
        ("shadow_body", C.c_int32 * MAX_BODIES),
    ]


class Drag(C.Structure):
    _fields_ = [
        ("density", C.c_int32),
        ("_pad", C.c_int32),
        ("rho0", C.c_double),
        ("r0", C.c_double),
        ("ref_alt_m", C.c_double),
        ("max_alt_m", C.c_double),
        ("eq_radius_km", C.c_double),
        ("rotation", Rotation),
    ]


class SolidTidesC(C.Structure):
    _fields_ = [
        ("k2", C.c_double),
        ("k3", C.c_double),
        ("mu_km3_s2", C.c_double),
        ("eq_radius_km", C.c_double),
        ("rotation", Rotation),
        ("n_perturbers", C.c_int32),
        ("perturber_body", C.c_int32 * MAX_BODIES),
        ("compute_degree_3", C.c_int32 * MAX_BODIES),
        ("_pad", C.c_int32),
    ]


SCHED_MODEL, SCHED_CALIBRATED, SCHED_EXPLICIT = 0, 1, 2


class Tuning(C.Structure):
    """``nyx_hip_tuning_t`` (ABI v4): every execution switch of the library; defaults = all "auto" (``Tuning()``).
    ``schedule``: SCHED_MODEL (default, process-independent), SCHED_CALIBRATED (measured at the first launch) or
    SCHED_EXPLICIT (``wave_weights``); ``deterministic=1``: bits per trajectory independent of the batch."""
    _fields_ = [
        ("schedule", C.c_int32), ("deterministic", C.c_int32), ("cooperative", C.c_int32), ("pipelined", C.c_int32),
        ("chained_attempts", C.c_int32), ("epoch_data_reuse", C.c_int32), ("role_fanout", C.c_int32), ("merge_roles", C.c_int32),
        ("stm_quad", C.c_int32), ("harmonics_feed", C.c_int32), ("coop_max_columns", C.c_int32), ("coop_mute", C.c_int32),
        ("profile", C.c_int32), ("debug_flags", C.c_int32),
        ("coop_fraction", C.c_double), ("coop_helper_ratio", C.c_double), ("column_start_cost", C.c_double),
        ("role_duties", C.c_double * 3), ("age_weights", C.c_double * 4), ("wave_weights", C.c_double * 16),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.cooperative = self.pipelined = self.chained_attempts = self.epoch_data_reuse = self.role_fanout = -1
        self.stm_quad = self.harmonics_feed = -1
        self.column_start_cost = -1.0
        for k, v in kw.items():
            if k in ("role_duties", "age_weights", "wave_weights"):
                arr = getattr(self, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                if not hasattr(self, k):
                    raise AttributeError(f"nyx_hip_tuning_t has no field {k!r}")
                setattr(self, k, v)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("flags", C.c_uint32),
        ("opts", IntegOpts),
        ("central_mu_km3_s2", C.c_double),
        ("n_segments", C.c_int32),
        ("n_bodies", C.c_int32),
        ("segments", C.POINTER(ChebySegment)),
        ("bodies", C.POINTER(Body)),
        ("n_point_masses", C.c_int32),
        ("point_mass_body", C.c_int32 * MAX_BODIES),
        ("gravity", C.POINTER(GravityField)),
        ("srp", C.POINTER(Srp)),
        ("drag", C.POINTER(Drag)),
        ("speed_of_light_km_s", C.c_double),
        ("tides", C.POINTER(SolidTidesC)),
        ("state_frame_body", C.c_int32),   # opts.integration_frame: the body the states of a batch are centred on (0: no swap)
        ("_pad_cfg", C.c_int32),
        ("tuning", C.POINTER(Tuning)),     # NULL => defaults (ABI v4)
        ("gravity2", C.POINTER(GravityField)),  # a second field of the same OrbitalDynamics (NULL => none)
    ]


class States(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("epoch_ns", c_int64_p),
        ("x_km", c_double_p),
        ("y_km", c_double_p),
        ("z_km", c_double_p),
        ("vx_km_s", c_double_p),
        ("vy_km_s", c_double_p),
        ("vz_km_s", c_double_p),
        ("cr", c_double_p),
        ("cd", c_double_p),
        ("prop_mass_kg", c_double_p),
        ("dry_mass_kg", c_double_p),
        ("extra_mass_kg", c_double_p),
        ("srp_area_m2", c_double_p),
        ("drag_area_m2", c_double_p),
        ("stm", c_double_p),
        ("step_ns", c_int64_p),
    ]


class StepStats(C.Structure):
    _fields_ = [
        ("status", c_int32_p),
        ("last_step_ns", c_int64_p),
        ("last_error", c_double_p),
        ("last_attempts", c_int32_p),
        ("n_accepted", c_int64_p),
        ("n_rejected", c_int64_p),
        ("n_evals", c_int64_p),
    ]


class Traj(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("epoch_ns", c_int64_p),
        ("x_km", c_double_p), ("y_km", c_double_p), ("z_km", c_double_p),
        ("vx_km_s", c_double_p), ("vy_km_s", c_double_p), ("vz_km_s", c_double_p),
        ("len", c_int32_p),
    ]


MAX_PROCESS_NOISE = 4
# enum nyx_hip_event_scalar
(EV_TRUE_ANOMALY_DEG, EV_RMAG_KM, EV_VMAG_KM_S, EV_SMA_KM, EV_ECC, EV_X_KM, EV_Y_KM, EV_Z_KM, EV_VX_KM_S, EV_VY_KM_S,
 EV_VZ_KM_S, EV_LONGITUDE_DEG, EV_DECLINATION_DEG, EV_LATITUDE_DEG, EV_HEIGHT_KM) = range(15)


class EventC(C.Structure):
    _fields_ = [("scalar", C.c_int32), ("trigger", C.c_int32), ("desired", C.c_double), ("value_precision", C.c_double),
                ("epoch_precision_ns", C.c_int64), ("has_frame", C.c_int32), ("_pad", C.c_int32),
                ("frame_eq_radius_km", C.c_double), ("frame_flattening", C.c_double), ("frame", Rotation)]



FRAME_INERTIAL, FRAME_RIC, FRAME_VNC = 0, 1, 2


class ProcessNoiseC(C.Structure):
    _fields_ = [("diag", C.c_double * 3), ("disable_time_ns", C.c_int64), ("start_time_ns", C.c_int64),
                ("has_start_time", C.c_int32), ("local_frame", C.c_int32), ("has_decay", C.c_int32), ("_pad", C.c_int32),
                ("decay_s", C.c_double * 3), ("init_epoch_ns", C.c_int64)]


class Predict(C.Structure):
    _fields_ = [("max_step_ns", C.c_int64), ("end_epoch_ns", C.c_int64), ("deviation_tracking", C.c_int32),
                ("n_process_noise", C.c_int32), ("process_noise", ProcessNoiseC * MAX_PROCESS_NOISE)]


class Estimates(C.Structure):
    _fields_ = [("covar", c_double_p), ("state_dev", c_double_p)]


class PredictHistory(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("epoch_ns", c_int64_p), ("state", c_double_p), ("stm", c_double_p),
                ("covar", c_double_p), ("state_dev", c_double_p), ("n_updates", c_int32_p)]


class TrajBatch:
    """Dense output of a batch: entry k of trajectory i at [k, i]; k = 0 is the start state (step-major, as the ABI)."""

    def __init__(self, n: int, capacity: int):
        self.n, self.capacity = n, capacity
        self.epoch_ns = np.zeros((capacity, n), dtype=np.int64)
        self.state = np.zeros((6, capacity, n), dtype=np.float64)
        self.len = np.zeros(n, dtype=np.int32)

    def as_c(self) -> "Traj":
        t = Traj()
        t.capacity = self.capacity
        t.epoch_ns = self.epoch_ns.ctypes.data_as(c_int64_p)
        for k, f in enumerate(["x_km", "y_km", "z_km", "vx_km_s", "vy_km_s", "vz_km_s"]):
            setattr(t, f, self.state[k].ctypes.data_as(c_double_p))
        t.len = self.len.ctypes.data_as(c_int32_p)
        return t

    def trajectory(self, i: int):
        """(epochs, states[len, 6]) of run i in propagation order; `finalize()` of the reference = sort by epoch."""
        m = min(int(self.len[i]), self.capacity)
        return self.epoch_ns[:m, i].copy(), self.state[:, :m, i].T.copy()


F64_FIELDS = ["x_km", "y_km", "z_km", "vx_km_s", "vy_km_s", "vz_km_s", "cr", "cd", "prop_mass_kg",
              "dry_mass_kg", "extra_mass_kg", "srp_area_m2", "drag_area_m2"]


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


class StateBatch:
    """Host-side SoA batch of ``Spacecraft`` (owns numpy arrays, exposes a ``States`` view).

    Mirrors ``Spacecraft::to_vector``/``set`` (reference cosmic/spacecraft.rs:451-497).
    """

    def __init__(self, n: int, with_stm: bool = False):
        self.n = int(n)
        self.epoch_ns = np.zeros(n, dtype=np.int64)
        for f in F64_FIELDS:
            setattr(self, f, np.zeros(n, dtype=np.float64))
        self.stm = np.zeros((n, 81), dtype=np.float64) if with_stm else None
        self.step_ns = np.zeros(n, dtype=np.int64)

    def reset_stm(self):
        """``Spacecraft::reset_stm`` (cosmic/spacecraft.rs:441-443): Phi := I9 (column-major)."""
        if self.stm is None:
            self.stm = np.zeros((self.n, 81), dtype=np.float64)
        self.stm[:] = np.eye(9).reshape(-1)

    def rv(self) -> np.ndarray:
        return np.stack([self.x_km, self.y_km, self.z_km, self.vx_km_s, self.vy_km_s, self.vz_km_s], axis=1)

    def set_rv(self, rv: np.ndarray):
        rv = np.asarray(rv, dtype=np.float64).reshape(self.n, 6)
        for i, f in enumerate(F64_FIELDS[:6]):
            getattr(self, f)[:] = rv[:, i]

    def copy(self) -> "StateBatch":
        o = StateBatch(self.n, self.stm is not None)
        o.epoch_ns[:] = self.epoch_ns
        for f in F64_FIELDS:
            getattr(o, f)[:] = getattr(self, f)
        if self.stm is not None:
            o.stm[:] = self.stm
        o.step_ns[:] = self.step_ns
        return o

    def slice(self, lo: int, hi: int) -> "StateBatch":
        o = StateBatch(hi - lo, self.stm is not None)
        o.epoch_ns[:] = self.epoch_ns[lo:hi]
        for f in F64_FIELDS:
            getattr(o, f)[:] = getattr(self, f)[lo:hi]
        if self.stm is not None:
            o.stm[:] = self.stm[lo:hi]
        o.step_ns[:] = self.step_ns[lo:hi]
        return o

    def take(self, index) -> "StateBatch":
        """The trajectories `index` (an integer array), in that order."""
        index = np.asarray(index, dtype=np.int64)
        o = StateBatch(len(index), self.stm is not None)
        o.epoch_ns[:] = self.epoch_ns[index]
        for f in F64_FIELDS:
            getattr(o, f)[:] = getattr(self, f)[index]
        if self.stm is not None:
            o.stm[:] = self.stm[index]
        o.step_ns[:] = self.step_ns[index]
        return o

    def as_c(self) -> States:
        s = States()
        s.n = self.n
        s.epoch_ns = self.epoch_ns.ctypes.data_as(c_int64_p)
        for f in F64_FIELDS:
            setattr(s, f, _dp(getattr(self, f)))
        s.stm = _dp(self.stm) if self.stm is not None else c_double_p()
        s.step_ns = self.step_ns.ctypes.data_as(c_int64_p)
        return s


class StatsBatch:
    """Per-trajectory ``IntegrationDetails`` + counters (reference propagators/mod.rs:49-56)."""

    def __init__(self, n: int):
        self.n = n
        self.status = np.zeros(n, dtype=np.int32)
        self.last_step_ns = np.zeros(n, dtype=np.int64)
        self.last_error = np.zeros(n, dtype=np.float64)
        self.last_attempts = np.zeros(n, dtype=np.int32)
        self.n_accepted = np.zeros(n, dtype=np.int64)
        self.n_rejected = np.zeros(n, dtype=np.int64)
        self.n_evals = np.zeros(n, dtype=np.int64)

    def as_c(self) -> StepStats:
        t = StepStats()
        t.status = self.status.ctypes.data_as(c_int32_p)
        t.last_step_ns = self.last_step_ns.ctypes.data_as(c_int64_p)
        t.last_error = _dp(self.last_error)
        t.last_attempts = self.last_attempts.ctypes.data_as(c_int32_p)
        t.n_accepted = self.n_accepted.ctypes.data_as(c_int64_p)
        t.n_rejected = self.n_rejected.ctypes.data_as(c_int64_p)
        t.n_evals = self.n_evals.ctypes.data_as(c_int64_p)
        return t


# ---------------------------------------------------------------------------------------------
# library loading — the product path fails loudly when the HIP extension is missing
# ---------------------------------------------------------------------------------------------

_LIB = None
LIB_NAME = "libnyx_hip.so"


def lib_path() -> str:
    """In-tree build; NYX_HIP_LIB selects another build of the same library (kernel tuning experiments)."""
    return os.environ.get("NYX_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


EXPORTS = [
    "nyx_hip_device_count", "nyx_hip_ctx_create", "nyx_hip_ctx_destroy", "nyx_hip_propagate_batch",
    "nyx_hip_propagate_batch_device", "nyx_hip_propagate_until_epoch", "nyx_hip_ctx_set_column_waves",
    "nyx_hip_last_kernel_ms", "nyx_hip_last_error", "nyx_hip_load_cof", "nyx_hip_load_shadr", "nyx_hip_free",
    "nyx_hip_propagate_batch_with_traj", "nyx_hip_propagate_batch_with_traj_device",
    "nyx_hip_traj_at", "nyx_hip_traj_every", "nyx_hip_traj_at_device", "nyx_hip_traj_every_device",
    "nyx_hip_predict_until", "nyx_hip_propagate_until_event", "nyx_hip_last_coop_helpers", "nyx_hip_ctx_set_tuning",
    "nyx_hip_propagate_batch_sharded", "nyx_hip_ensemble_moments", "nyx_hip_ensemble_moments_device",
]


def load_library():
    """dlopen the in-tree HIP extension.  No CPU fallback: a missing build is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "nyx_amd has no CPU fallback."
        )
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.nyx_hip_device_count.restype = C.c_int32
    lib.nyx_hip_ctx_create.argtypes = [C.POINTER(Config), C.c_int32, C.POINTER(C.c_void_p)]
    lib.nyx_hip_ctx_create.restype = C.c_int32
    lib.nyx_hip_ctx_destroy.argtypes = [C.c_void_p]
    lib.nyx_hip_ctx_destroy.restype = None
    lib.nyx_hip_propagate_batch.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(States), C.POINTER(StepStats)]
    lib.nyx_hip_propagate_batch.restype = C.c_int32
    lib.nyx_hip_propagate_batch_device.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(States),
                                                   C.POINTER(StepStats), C.c_void_p]
    lib.nyx_hip_propagate_batch_device.restype = C.c_int32
    lib.nyx_hip_propagate_batch_with_traj.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(States), C.POINTER(StepStats), C.POINTER(Traj)]
    lib.nyx_hip_propagate_batch_with_traj.restype = C.c_int32
    lib.nyx_hip_propagate_batch_with_traj_device.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(States), C.POINTER(StepStats),
                                                             C.POINTER(Traj), C.c_void_p]
    lib.nyx_hip_propagate_batch_with_traj_device.restype = C.c_int32
    lib.nyx_hip_traj_at.argtypes = [C.c_void_p, C.POINTER(Traj), C.c_int64, c_int64_p, C.c_int64, C.POINTER(Traj), c_int32_p]
    lib.nyx_hip_traj_at.restype = C.c_int32
    lib.nyx_hip_traj_every.argtypes = [C.c_void_p, C.POINTER(Traj), C.c_int64, C.c_int64, C.POINTER(Traj)]
    lib.nyx_hip_traj_every.restype = C.c_int32
    lib.nyx_hip_traj_at_device.argtypes = [C.c_void_p, C.POINTER(Traj), C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Traj), C.c_void_p, C.c_void_p]
    lib.nyx_hip_traj_at_device.restype = C.c_int32
    lib.nyx_hip_traj_every_device.argtypes = [C.c_void_p, C.POINTER(Traj), C.c_int64, C.c_int64, C.POINTER(Traj), C.c_void_p]
    lib.nyx_hip_traj_every_device.restype = C.c_int32
    lib.nyx_hip_propagate_until_event.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(EventC), C.POINTER(States),
                                                  C.POINTER(StepStats), C.POINTER(Traj), c_int32_p]
    lib.nyx_hip_propagate_until_event.restype = C.c_int32
    lib.nyx_hip_predict_until.argtypes = [C.c_void_p, C.POINTER(States), C.POINTER(Predict), C.POINTER(Estimates), C.POINTER(States),
                                          C.POINTER(StepStats), C.POINTER(PredictHistory)]
    lib.nyx_hip_predict_until.restype = C.c_int32
    lib.nyx_hip_propagate_until_epoch.argtypes = [C.c_void_p, C.POINTER(States), C.c_int64, C.POINTER(States), C.POINTER(StepStats)]
    lib.nyx_hip_propagate_until_epoch.restype = C.c_int32
    lib.nyx_hip_ctx_set_column_waves.argtypes = [C.c_void_p, C.c_int32]
    lib.nyx_hip_ctx_set_column_waves.restype = C.c_int32
    lib.nyx_hip_last_coop_helpers.argtypes = [C.c_void_p]
    lib.nyx_hip_last_coop_helpers.restype = C.c_int32
    lib.nyx_hip_last_kernel_ms.argtypes = [C.c_void_p]
    lib.nyx_hip_last_kernel_ms.restype = C.c_double
    lib.nyx_hip_last_error.restype = C.c_char_p
    lib.nyx_hip_load_cof.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_int32_p,
                                     C.POINTER(c_double_p), C.POINTER(c_double_p)]
    lib.nyx_hip_load_cof.restype = C.c_int32
    lib.nyx_hip_load_shadr.argtypes = lib.nyx_hip_load_cof.argtypes
    lib.nyx_hip_load_shadr.restype = C.c_int32
    lib.nyx_hip_free.argtypes = [C.c_void_p]
    lib.nyx_hip_free.restype = None
    lib.nyx_hip_propagate_batch_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(States), C.c_int64, C.POINTER(States),
                                                    C.POINTER(StepStats), C.POINTER(Traj)]
    lib.nyx_hip_propagate_batch_sharded.restype = C.c_int32
    lib.nyx_hip_ensemble_moments.argtypes = [C.c_void_p, C.POINTER(States), c_int32_p, c_double_p, c_double_p]
    lib.nyx_hip_ensemble_moments.restype = C.c_int32
    lib.nyx_hip_ensemble_moments_device.argtypes = [C.c_void_p, C.POINTER(States), C.c_void_p, c_double_p, C.c_void_p, C.c_void_p]
    lib.nyx_hip_ensemble_moments_device.restype = C.c_int32
    lib.nyx_hip_ctx_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
    lib.nyx_hip_ctx_set_tuning.restype = C.c_int32
    lib.nyx_hip_abi_sizeof.argtypes = [C.c_int32]
    lib.nyx_hip_abi_sizeof.restype = C.c_int64
    _LIB = lib
    return lib


def last_error() -> str:
    lib = load_library()
    s = lib.nyx_hip_last_error()
    return s.decode() if s else ""
