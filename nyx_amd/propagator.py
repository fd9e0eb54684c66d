"""Host-side mirror of the reference's operator interface for the batched propagation path.

Same names, argument meaning and error behaviour as the reference so that the parity tests
read like the reference's own tests:

* ``IntegratorOptions``     -> nyx-core/src/propagators/options.rs:42-186
* ``IntegratorMethod``      -> propagators/rk_methods/mod.rs:65-79
* ``ErrorControl``          -> propagators/error_ctrl.rs:30-76
* ``PointMasses``           -> dynamics/orbital.rs:174-198
* ``GravityFieldData``      -> io/gravity.rs:90-128, 514-516
* ``SolarPressure``         -> dynamics/solarpressure.rs:40-128
* ``Drag``                  -> dynamics/drag.rs:115-160
* ``OrbitalDynamics`` / ``SpacecraftDynamics`` -> dynamics/orbital.rs:43-71, dynamics/spacecraft.rs:44-107
* ``Propagator`` / ``PropInstance``            -> propagators/propagator.rs:34-121, instance.rs:62-352
* ``Propagator.many_for_duration``             -> nyx-py/src/py_md.rs:275-321

Everything numeric happens behind the C-ABI (include/nyx_hip.h) on the GPU.  There is no CPU
fallback in this module.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi

NS_PER_S = 1_000_000_000
SPEED_OF_LIGHT_KM_S = 299_792.458  # anise::constants::SPEED_OF_LIGHT_KM_S (reference cosmic/mod.rs:179-180)

# NAIF ids (anise::constants::celestial_objects)
SUN, MOON, EARTH, EARTH_MOON_BARYCENTER, JUPITER_BARYCENTER, SSB = 10, 301, 399, 3, 5, 0


def seconds(x: float) -> int:
    """``x * Unit::Second`` -> integer-ns ``Duration`` (hifitime: f64*1e9 then `as i64`)."""
    t = float(x) * 1e9
    if t != t:
        return 0
    return int(t)  # truncation toward zero


def to_seconds(ns: int) -> float:
    """``Duration::to_seconds``."""
    ns = int(ns)
    npc = 3155760000 * NS_PER_S
    cent, rem = divmod(ns, npc)
    whole, sub = divmod(rem, NS_PER_S)
    if cent == 0:
        return float(whole) + float(sub) * 1e-9
    return float(cent) * 3155760000.0 + float(whole) + float(sub) * 1e-9


class IntegratorMethod(enum.IntEnum):
    RungeKutta89 = _abi.RK89
    DormandPrince78 = _abi.DP78
    DormandPrince45 = _abi.DP45
    RungeKutta4 = _abi.RK4
    CashKarp45 = _abi.CASHKARP45
    Verner56 = _abi.VERNER56

    @classmethod
    def from_str(cls, s: str) -> "IntegratorMethod":
        """rk_methods/mod.rs:135-164 (case-insensitive)."""
        for m in cls:
            if m.name.lower() == s.lower():
                return m
        raise ValueError(f"unknow integration method `{s}`, must be one of " + ",".join(m.name for m in cls))


class ErrorControl(enum.IntEnum):
    RSSCartesianState = _abi.RSS_CARTESIAN_STATE
    RSSCartesianStep = _abi.RSS_CARTESIAN_STEP
    RSSState = _abi.RSS_STATE
    RSSStep = _abi.RSS_STEP
    LargestError = _abi.LARGEST_ERROR
    LargestState = _abi.LARGEST_STATE
    LargestStep = _abi.LARGEST_STEP


@dataclass
class IntegratorOptions:
    """options.rs:42-61; defaults are GMAT's (options.rs:172-186). Durations are integer ns."""

    init_step: int = 60 * NS_PER_S
    min_step: int = NS_PER_S // 1000
    max_step: int = 2700 * NS_PER_S
    tolerance: float = 1e-12
    attempts: int = 50
    fixed_step: bool = False
    error_ctrl: ErrorControl = ErrorControl.RSSCartesianStep
    integration_frame: Optional["Frame"] = None   # options.rs:60: integrate in this frame, hand the states back in their own

    @classmethod
    def with_adaptive_step(cls, min_step: int, max_step: int, tolerance: float, error_ctrl: ErrorControl):
        return cls(init_step=max_step, min_step=min_step, max_step=max_step, tolerance=tolerance, attempts=50,
                   fixed_step=False, error_ctrl=error_ctrl)

    @classmethod
    def with_adaptive_step_s(cls, min_step: float, max_step: float, tolerance: float, error_ctrl: ErrorControl):
        return cls.with_adaptive_step(seconds(min_step), seconds(max_step), tolerance, error_ctrl)

    @classmethod
    def with_fixed_step(cls, step: int):
        return cls(init_step=step, min_step=step, max_step=step, tolerance=0.0, fixed_step=True, attempts=0,
                   error_ctrl=ErrorControl.RSSCartesianStep)

    @classmethod
    def with_fixed_step_s(cls, step: float):
        return cls.with_fixed_step(seconds(step))

    @classmethod
    def with_tolerance(cls, tolerance: float):
        o = cls()
        o.tolerance = tolerance
        return o

    @classmethod
    def with_max_step(cls, max_step: int):
        o = cls()
        o.set_max_step(max_step)
        return o

    def set_max_step(self, max_step: int):
        if self.init_step > max_step:
            self.init_step = max_step
        self.max_step = max_step

    def set_min_step(self, min_step: int):
        if self.init_step < min_step:
            self.init_step = min_step
        self.min_step = min_step


# --------------------------------------------------------------------------------------------
# "Almanac": the plain data the path needs from ANISE (ephemeris segments, body constants,
# body-fixed orientation).  The host flattens ANISE's ephemeris tree into signed chains.
# --------------------------------------------------------------------------------------------


@dataclass
class ChebySegment:
    """SPK-type-2-style segment: records[r] = [mid, radius, X.., Y.., Z..]."""

    init_et_s: float
    interval_s: float
    records: np.ndarray  # (n_records, 2 + 3*n_coeffs)

    @property
    def n_coeffs(self) -> int:
        return (self.records.shape[1] - 2) // 3


@dataclass
class Rotation:
    """Body-fixed orientation as ANISE's planetary data hold it (`nyx_hip_rotation_t`): the IAU phase angles (PCK: alpha /
    delta in deg + deg/century, W in deg + deg/day) with their trigonometric nutation-precession series, or - `euler` set -
    Chebyshev Euler angles of a binary PCK (ITRF93, MOON_PA): a `ChebySegment` of [A1, A2, A3] in radians with
    DCM(base->fixed) = R3(A3) R1(A2) R3(A1) and `base_dcm` = integration frame -> base frame."""

    ra_deg: Sequence[float] = (0.0, 0.0, 0.0)
    dec_deg: Sequence[float] = (90.0, 0.0, 0.0)
    w_deg: Sequence[float] = (0.0, 0.0, 0.0)
    nut_prec_angles_deg: Sequence[Sequence[float]] = ()   # [(theta0, theta1 per century), ...]
    nut_prec_ra: Sequence[float] = ()
    nut_prec_dec: Sequence[float] = ()
    nut_prec_w: Sequence[float] = ()
    euler: Optional["ChebySegment"] = None
    base_dcm: Sequence[float] = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


# pck00008 Earth (IAU 2000): alpha0 = 0 - 0.641 T, delta0 = 90 - 0.557 T, W = 190.147 + 360.9856235 d
IAU_EARTH_ROTATION = Rotation((0.0, -0.641, 0.0), (90.0, -0.557, 0.0), (190.147, 360.9856235, 0.0))
# pck00008 Moon, polynomial part only
IAU_MOON_ROTATION_POLY = Rotation((269.9949, 0.0031, 0.0), (66.5392, 0.0130, 0.0), (38.3213, 13.17635815, -1.4e-12))
# pck00008 Moon with its 13-term series (BODY3_NUT_PREC_ANGLES E1..E13, BODY301_NUT_PREC_RA / _DEC / _PM; IAU/IAG 2000
# report, restated from the published table - the descriptor is data: a caller with ANISE passes its PlanetaryData)
_E_MOON = ((125.045, -1935.5364525000), (250.089, -3871.0729050000), (260.008, 475263.3328725000), (176.625, 487269.6299850000),
           (357.529, 35999.0509575000), (311.589, 964468.4993100000), (134.963, 477198.8675605000), (276.617, 12006.3007650000),
           (34.226, 63863.5132425000), (15.134, -5806.6093575000), (119.743, 131.8406400000), (239.961, 6003.1503825000),
           (25.053, 473327.2793700000))
IAU_MOON_ROTATION = Rotation((269.9949, 0.0031, 0.0), (66.5392, 0.0130, 0.0), (38.3213, 13.17635815, -1.4e-12), _E_MOON,
                             (-3.8787, -0.1204, 0.0700, -0.0172, 0.0, 0.0072, 0.0, 0.0, 0.0, -0.0052, 0.0, 0.0, 0.0043),
                             (1.5419, 0.0239, -0.0278, 0.0068, 0.0, -0.0029, 0.0009, 0.0, 0.0, 0.0008, 0.0, 0.0, -0.0009),
                             (3.5610, 0.1208, -0.0642, 0.0158, 0.0252, -0.0066, -0.0047, -0.0046, 0.0028, 0.0052, 0.0040, 0.0019, -0.0044))


@dataclass
class Frame:
    """What the path reads from an ANISE ``Frame``: mu, mean equatorial radius, orientation."""

    naif_id: int
    mu_km3_s2: float
    mean_equatorial_radius_km: float = 0.0
    rotation: Optional[Rotation] = None  # None => inertial (J2000 orientation)
    flattening: float = 0.0              # (a - c) / a of the frame's ellipsoid (geodetic event scalars)

    def with_mu_km3_s2(self, mu: float) -> "Frame":
        return Frame(self.naif_id, mu, self.mean_equatorial_radius_km, self.rotation, self.flattening)


class Almanac:
    """Container of segments and body constants (values of data/02_config/full_seq.dhall:150-182)."""

    def __init__(self):
        self.segments: List[ChebySegment] = []
        self.bodies = {}  # naif_id -> dict(mu, radius, chain=[(seg_idx, sign)])

    def add_segment(self, seg: ChebySegment) -> int:
        self.segments.append(seg)
        return len(self.segments) - 1

    def add_body(self, naif_id: int, mu_km3_s2: float, mean_radius_km: float, chain):
        self.bodies[naif_id] = dict(mu=mu_km3_s2, radius=mean_radius_km, chain=list(chain))

    def frame_info(self, naif_id: int, rotation: Optional[Rotation] = None) -> Frame:
        b = self.bodies[naif_id]
        return Frame(naif_id, b["mu"], b["radius"], rotation)


@dataclass
class PointMasses:
    celestial_objects: List[int]


@dataclass
class GravityFieldData:
    """io/gravity.rs:90-96. ``c_nm``/``s_nm`` packed lower-triangular, idx = n(n+1)/2+m."""

    degree: int
    order: int
    c_nm: np.ndarray
    s_nm: np.ndarray
    frame: Frame

    @classmethod
    def from_j2(cls, j2: float, frame: Frame) -> "GravityFieldData":
        """io/gravity.rs:117-128 — stores the (normalised) C20 as given."""
        c = np.zeros(6)
        c[3] = j2
        return cls(2, 0, c, np.zeros(6), frame)

    @classmethod
    def from_cof(cls, path: str, degree: int, order: int, gunzipped: bool, frame: Frame) -> "GravityFieldData":
        return cls._load(path, degree, order, gunzipped, frame, "nyx_hip_load_cof")

    @classmethod
    def from_shadr(cls, path: str, degree: int, order: int, gunzipped: bool, frame: Frame) -> "GravityFieldData":
        return cls._load(path, degree, order, gunzipped, frame, "nyx_hip_load_shadr")

    @classmethod
    def _load(cls, path, degree, order, gunzipped, frame, fn):
        lib = _abi.load_library()
        od, oo = C.c_int32(), C.c_int32()
        pc, ps = _abi.c_double_p(), _abi.c_double_p()
        rc = getattr(lib, fn)(str(path).encode(), degree, order, int(gunzipped), C.byref(od), C.byref(oo), C.byref(pc), C.byref(ps))
        if rc != 0:
            raise IOError(f"FileUnreadable: {_abi.last_error()}")
        n = (degree + 1) * (degree + 2) // 2
        c = np.ctypeslib.as_array(pc, shape=(n,)).copy()
        s = np.ctypeslib.as_array(ps, shape=(n,)).copy()
        lib.nyx_hip_free(pc)
        lib.nyx_hip_free(ps)
        # from_cof keeps the max degree/order seen (io/gravity.rs:330-366); arrays stay sized for `degree`
        d = od.value
        nn = (d + 1) * (d + 2) // 2
        return cls(d, oo.value, c[:nn].copy(), s[:nn].copy(), frame)

    @classmethod
    def from_packed_file(cls, path: str, frame: Frame, degree: Optional[int] = None, order: Optional[int] = None):
        """Loads the converted fixture written by tools/convert_cof.py (float64 little-endian:
        [degree, order, C packed..., S packed...])."""
        raw = np.fromfile(path, dtype="<f8")
        d0, o0 = int(raw[0]), int(raw[1])
        n0 = (d0 + 1) * (d0 + 2) // 2
        c0, s0 = raw[2:2 + n0], raw[2 + n0:2 + 2 * n0]
        d = d0 if degree is None else min(degree, d0)
        o = o0 if order is None else min(order, o0, d)
        n = (d + 1) * (d + 2) // 2
        c, s = c0[:n].copy(), s0[:n].copy()
        for nn in range(d + 1):  # entries with m > order are never stored by the loaders
            for m in range(o + 1, nn + 1):
                c[nn * (nn + 1) // 2 + m] = 0.0
                s[nn * (nn + 1) // 2 + m] = 0.0
        return cls(d, o, c, s, frame)


@dataclass
class SolarPressure:
    """solarpressure.rs:40-128 (`estimate` defaults to True for default_flux/new, :88-93)."""

    shadow_bodies: List[int]
    phi: float = 1367.0
    estimate: bool = True
    light_source: int = SUN

    @classmethod
    def default_flux(cls, shadow_body: int):
        return cls([shadow_body])

    @classmethod
    def default_no_estimation(cls, shadow_bodies: List[int]):
        return cls(list(shadow_bodies), estimate=False)

    @classmethod
    def with_flux(cls, flux_w_m2: float, shadow_bodies: List[int]):
        return cls(list(shadow_bodies), phi=flux_w_m2)


@dataclass
class Drag:
    """drag.rs:115-160. density: ('constant', rho) | ('exponential', rho0, r0, ref_alt_m) | ('stdatm', max_alt_m)."""

    density: tuple
    frame: Frame
    estimate: bool = False

    @classmethod
    def earth_exp(cls, iau_earth: Frame):
        return cls(("exponential", 3.614e-13, 700_000.0, 88_667.0), iau_earth)

    @classmethod
    def std_atm1976(cls, iau_earth: Frame):
        return cls(("stdatm", 1_000_000.0), iau_earth)


@dataclass
class TidalPerturber:
    """solid_tides.rs:65-72: the body raising the tide (NAIF id; its GM comes from the almanac's planetary data)."""

    naif_id: int
    compute_degree_3: bool


@dataclass
class SolidTides:
    """solid_tides.rs:43-63: IERS-2010 solid tides of the central body; `frame` is its body-fixed frame."""

    frame: Frame
    k2: float
    k3: float
    perturbers: List[TidalPerturber]

    @classmethod
    def earth_moon_system(cls, earth_frame: Frame, moon_id: int = 301, sun_id: int = 10):
        """solid_tides.rs:181-219: Moon (degrees 2 and 3) and Sun (degree 2), k2 = 0.3019, k3 = 0.093."""
        return cls(earth_frame, 0.3019, 0.093, [TidalPerturber(moon_id, True), TidalPerturber(sun_id, False)])


@dataclass
class OrbitalDynamics:
    accel_models: list = field(default_factory=list)

    @classmethod
    def two_body(cls):
        return cls([])

    @classmethod
    def point_masses(cls, celestial_objects: List[int]):
        return cls([PointMasses(list(celestial_objects))])

    @classmethod
    def from_model(cls, model):
        return cls([model])


@dataclass
class SpacecraftDynamics:
    orbital_dyn: OrbitalDynamics
    force_models: list = field(default_factory=list)

    @classmethod
    def new(cls, orbital_dyn: OrbitalDynamics):
        return cls(orbital_dyn, [])

    @classmethod
    def from_model(cls, orbital_dyn: OrbitalDynamics, force_model):
        return cls(orbital_dyn, [force_model])

    @classmethod
    def from_models(cls, orbital_dyn: OrbitalDynamics, force_models):
        return cls(orbital_dyn, list(force_models))


@dataclass
class Spacecraft:
    """cosmic/spacecraft.rs:115-143, the fields this path reads."""

    epoch_ns: int
    rv: Sequence[float]  # x,y,z km, vx,vy,vz km/s in `frame`
    frame: Frame
    dry_mass_kg: float = 0.0
    prop_mass_kg: float = 0.0
    extra_mass_kg: float = 0.0
    srp_area_m2: float = 0.0
    cr: float = 1.8  # SRPData default (cosmic/spacecraft.rs:210-217 from_srp_defaults)
    drag_area_m2: float = 0.0
    cd: float = 2.2
    stm: Optional[np.ndarray] = None  # 9x9

    def with_stm(self) -> "Spacecraft":
        s = Spacecraft(**{**self.__dict__})
        s.stm = np.eye(9)
        return s


class PropagationError(RuntimeError):
    def __init__(self, status: int, index: int = 0):
        self.status = status
        self.index = index
        super().__init__(f"run {index}: {_abi.STATUS_NAMES[status]}")


class CompiledConfig:
    """C descriptor + the numpy arrays it points into (kept alive here)."""

    def __init__(self, cfg: _abi.Config, keep: list, central: Frame):
        self.cfg = cfg
        self._keep = keep
        self.central = central


def compile_config(dynamics: SpacecraftDynamics, method: IntegratorMethod, opts: IntegratorOptions, almanac: Almanac,
                   central: Frame, stm: bool = False, stm_textbook: bool = False, state_frame: Optional[Frame] = None) -> CompiledConfig:
    """Flattens (dynamics, method, options, almanac) into ``nyx_hip_config_t`` — the analogue of
    ``PropagatorConfig::build`` (dynamics/sequence/config.rs:145-151) run in reverse."""
    keep: list = []
    cfg = _abi.Config()
    cfg.abi_version = _abi.ABI_VERSION
    cfg.flags = (_abi.FLAG_STM if stm else 0) | (_abi.FLAG_STM_TEXTBOOK if stm_textbook else 0)
    o = cfg.opts
    o.init_step_ns, o.min_step_ns, o.max_step_ns = int(opts.init_step), int(opts.min_step), int(opts.max_step)
    o.tolerance, o.attempts, o.fixed_step = float(opts.tolerance), int(opts.attempts), int(bool(opts.fixed_step))
    o.error_ctrl, o.method = int(opts.error_ctrl), int(method)
    cfg.central_mu_km3_s2 = float(central.mu_km3_s2)
    cfg.speed_of_light_km_s = SPEED_OF_LIGHT_KM_S

    # segments: the almanac's ephemeris segments, then the Euler-angle segments of the orientations that have one
    all_segments = list(almanac.segments)
    euler_index = {}

    def euler_segment_of(rot: Optional[Rotation]) -> int:
        if rot is None or rot.euler is None:
            return -1
        if id(rot.euler) not in euler_index:
            euler_index[id(rot.euler)] = len(all_segments)
            all_segments.append(rot.euler)
        return euler_index[id(rot.euler)]

    for m_ in list(dynamics.orbital_dyn.accel_models) + list(dynamics.force_models):
        euler_segment_of(getattr(getattr(m_, "frame", None), "rotation", None))
    segs = (_abi.ChebySegment * max(1, len(all_segments)))()
    for i, s in enumerate(all_segments):
        rec = np.ascontiguousarray(s.records, dtype=np.float64)
        keep.append(rec)
        segs[i].init_et_s, segs[i].interval_s = float(s.init_et_s), float(s.interval_s)
        segs[i].n_records, segs[i].n_coeffs = rec.shape[0], s.n_coeffs
        segs[i].records = rec.ctypes.data_as(_abi.c_double_p)
    keep.append(segs)
    cfg.n_segments = len(all_segments)
    cfg.segments = C.cast(segs, C.POINTER(_abi.ChebySegment))

    # bodies actually used, central body first
    body_index = {}
    body_list = []

    def body_of(naif_id: int) -> int:
        if naif_id in body_index:
            return body_index[naif_id]
        if naif_id == central.naif_id:
            b = dict(mu=central.mu_km3_s2, radius=central.mean_equatorial_radius_km, chain=[])
        else:
            if naif_id not in almanac.bodies:
                raise KeyError(f"planetary data from third body not loaded: {naif_id}")
            b = almanac.bodies[naif_id]
        body_index[naif_id] = len(body_list)
        body_list.append((naif_id, b))
        return body_index[naif_id]

    body_of(central.naif_id)
    # opts.integration_frame (instance.rs:117-142): `central` is the integration frame, the states come centred on `state_frame`
    cfg.state_frame_body = 0
    if state_frame is not None and state_frame.naif_id != central.naif_id:
        # the device swap is a pure translation (two frames of one orientation): a rotated pair needs almanac.transform_to's
        # rotation as well, which this path does not do - refused rather than silently treated as a translation
        if (state_frame.rotation is None) != (central.rotation is None) or (state_frame.rotation is not None and state_frame.rotation != central.rotation):
            raise NotImplementedError("opts.integration_frame: the state frame and the integration frame must have the same orientation")
        cfg.state_frame_body = body_of(state_frame.naif_id)
    pm = [m for m in dynamics.orbital_dyn.accel_models if isinstance(m, PointMasses)]
    gf = [m for m in dynamics.orbital_dyn.accel_models if isinstance(m, GravityFieldData)]
    srp = [m for m in dynamics.force_models if isinstance(m, SolarPressure)]
    drag = [m for m in dynamics.force_models if isinstance(m, Drag)]
    tides = [m for m in dynamics.orbital_dyn.accel_models if isinstance(m, SolidTides)]
    if len(pm) > 1 or len(gf) > 2 or len(srp) > 1 or len(drag) > 1 or len(tides) > 1:
        raise NotImplementedError("the device path takes at most one model of each kind (two gravity fields)")
    known = len(pm) + len(gf) + len(tides)
    if known != len(dynamics.orbital_dyn.accel_models) or len(srp) + len(drag) != len(dynamics.force_models):
        raise NotImplementedError("unsupported model on the device path (guidance laws fall back to the CPU reference)")

    cfg.n_point_masses = 0
    if pm:
        objs = [c for c in pm[0].celestial_objects]
        if len(objs) > _abi.MAX_BODIES:
            raise ValueError("too many point masses")
        for k, nid in enumerate(objs):
            cfg.point_mass_body[k] = body_of(nid)
        cfg.n_point_masses = len(objs)

    if srp:
        s = _abi.Srp()
        s.phi_w_m2, s.estimate = float(srp[0].phi), int(bool(srp[0].estimate))
        s.sun_body = body_of(srp[0].light_source)
        s.n_shadow_bodies = len(srp[0].shadow_bodies)
        for k, nid in enumerate(srp[0].shadow_bodies):
            s.shadow_body[k] = body_of(nid)
        keep.append(s)
        cfg.srp = C.pointer(s)

    def fill_rot(dst, rot: Optional[Rotation]):
        rot = rot or Rotation()
        for k in range(3):
            dst.ra_deg[k], dst.dec_deg[k], dst.w_deg[k] = float(rot.ra_deg[k]), float(rot.dec_deg[k]), float(rot.w_deg[k])
        n = len(rot.nut_prec_angles_deg)
        if n > _abi.MAX_NUT_PREC or max(len(rot.nut_prec_ra), len(rot.nut_prec_dec), len(rot.nut_prec_w)) > n:
            raise ValueError(f"at most {_abi.MAX_NUT_PREC} nutation-precession angles, and no more coefficients than angles")
        dst.n_nut_prec = n
        for k in range(n):
            dst.nut_prec_angle_deg[k][0], dst.nut_prec_angle_deg[k][1] = float(rot.nut_prec_angles_deg[k][0]), float(rot.nut_prec_angles_deg[k][1])
            dst.nut_prec_ra[k] = float(rot.nut_prec_ra[k]) if k < len(rot.nut_prec_ra) else 0.0
            dst.nut_prec_dec[k] = float(rot.nut_prec_dec[k]) if k < len(rot.nut_prec_dec) else 0.0
            dst.nut_prec_w[k] = float(rot.nut_prec_w[k]) if k < len(rot.nut_prec_w) else 0.0
        dst.kind = _abi.ROT_EULER_CHEBY if rot.euler is not None else _abi.ROT_IAU
        dst.euler_segment = euler_segment_of(rot)
        for k in range(9):
            dst.base_dcm[k] = float(rot.base_dcm[k])

    # One or two GravityFields (accel_models is a list: orbital.rs:100-108).  The larger one gets the column waves (`gravity`), the
    # other is walked in one piece by the perturbation wave (`gravity2`); the order of two terms of a sum is not what parity looks at.
    for which, g in enumerate(sorted(gf, key=lambda f: -int(f.degree))):
        gs = _abi.GravityField()
        gs.degree, gs.order = int(g.degree), int(g.order)
        if g.frame.naif_id != central.naif_id:
            # gravity_field.rs:150-154: the orbit is transformed into the field's frame WHATEVER its centre (translation to that body,
            # then its rotation), the acceleration is rotated back (:258-265): that body's harmonics at r - r_body(t)
            gs.offset_body = body_of(g.frame.naif_id) + 1
        gs.mu_km3_s2, gs.eq_radius_km = float(g.frame.mu_km3_s2), float(g.frame.mean_equatorial_radius_km)
        cn = np.ascontiguousarray(g.c_nm, dtype=np.float64)
        sn = np.ascontiguousarray(g.s_nm, dtype=np.float64)
        need = (g.degree + 1) * (g.degree + 2) // 2
        assert cn.size >= need and sn.size >= need
        keep += [cn, sn]
        gs.c_nm, gs.s_nm = cn.ctypes.data_as(_abi.c_double_p), sn.ctypes.data_as(_abi.c_double_p)
        fill_rot(gs.rotation, g.frame.rotation)
        keep.append(gs)
        if which == 0:
            cfg.gravity = C.pointer(gs)
        else:
            cfg.gravity2 = C.pointer(gs)

    if tides:
        t = tides[0]
        if t.frame.naif_id != central.naif_id:
            raise NotImplementedError("solid tides of a non-central body are not on the device path")
        ts = _abi.SolidTidesC()
        ts.k2, ts.k3 = float(t.k2), float(t.k3)
        ts.mu_km3_s2, ts.eq_radius_km = float(t.frame.mu_km3_s2), float(t.frame.mean_equatorial_radius_km)
        fill_rot(ts.rotation, t.frame.rotation)
        if len(t.perturbers) > _abi.MAX_BODIES:
            raise ValueError("too many tidal perturbers")
        ts.n_perturbers = len(t.perturbers)
        for k, pert in enumerate(t.perturbers):
            ts.perturber_body[k] = body_of(pert.naif_id)
            ts.compute_degree_3[k] = int(bool(pert.compute_degree_3))
        keep.append(ts)
        cfg.tides = C.pointer(ts)

    if drag:
        d = drag[0]
        ds = _abi.Drag()
        kind = d.density[0]
        if kind == "constant":
            ds.density, ds.rho0 = _abi.RHO_CONSTANT, float(d.density[1])
        elif kind == "exponential":
            ds.density = _abi.RHO_EXPONENTIAL
            ds.rho0, ds.r0, ds.ref_alt_m = (float(x) for x in d.density[1:4])
        elif kind == "stdatm":
            ds.density, ds.max_alt_m = _abi.RHO_STDATM, float(d.density[1])
        else:
            raise ValueError(kind)
        ds.eq_radius_km = float(d.frame.mean_equatorial_radius_km)
        fill_rot(ds.rotation, d.frame.rotation)
        keep.append(ds)
        cfg.drag = C.pointer(ds)

    if len(body_list) > _abi.MAX_BODIES:
        raise ValueError("too many bodies")
    bodies = (_abi.Body * len(body_list))()
    for i, (nid, b) in enumerate(body_list):
        bodies[i].naif_id = nid
        bodies[i].mu_km3_s2 = float(b["mu"])
        bodies[i].mean_radius_km = float(b["radius"])
        ch = b["chain"]
        if len(ch) > _abi.MAX_CHAIN:
            raise ValueError("ephemeris chain too long")
        bodies[i].n_chain = len(ch)
        for k, (si, sg) in enumerate(ch):
            bodies[i].chain_segment[k], bodies[i].chain_sign[k] = int(si), int(sg)
    keep.append(bodies)
    cfg.n_bodies = len(body_list)
    cfg.bodies = C.cast(bodies, C.POINTER(_abi.Body))
    return CompiledConfig(cfg, keep, central)


def pack_spacecraft(states: Sequence[Spacecraft], with_stm: bool) -> _abi.StateBatch:
    b = _abi.StateBatch(len(states), with_stm)
    for i, s in enumerate(states):
        b.epoch_ns[i] = int(s.epoch_ns)
        rv = np.asarray(s.rv, dtype=np.float64)
        b.x_km[i], b.y_km[i], b.z_km[i], b.vx_km_s[i], b.vy_km_s[i], b.vz_km_s[i] = rv
        b.cr[i], b.cd[i] = s.cr, s.cd
        b.prop_mass_kg[i], b.dry_mass_kg[i], b.extra_mass_kg[i] = s.prop_mass_kg, s.dry_mass_kg, s.extra_mass_kg
        b.srp_area_m2[i], b.drag_area_m2[i] = s.srp_area_m2, s.drag_area_m2
        if with_stm:
            m = np.eye(9) if s.stm is None else np.asarray(s.stm, dtype=np.float64)
            b.stm[i] = m.reshape(9, 9).T.reshape(-1)  # column-major (cosmic/spacecraft.rs:467-471)
    return b


def unpack_spacecraft(batch: _abi.StateBatch, template: Sequence[Spacecraft]) -> List[Spacecraft]:
    out = []
    rv = batch.rv()
    for i, t in enumerate(template):
        s = Spacecraft(**{**t.__dict__})
        s.epoch_ns = int(batch.epoch_ns[i])
        s.rv = rv[i].copy()
        s.cr, s.cd, s.prop_mass_kg = float(batch.cr[i]), float(batch.cd[i]), float(batch.prop_mass_kg[i])
        if batch.stm is not None:
            s.stm = batch.stm[i].reshape(9, 9).T.copy()
        out.append(s)
    return out


def moments_to_mean_cov(mom: np.ndarray, x0: Optional[np.ndarray] = None):
    """(mean[9], unbiased cov[9, 9]) from the 55 moments of `nyx_hip_ensemble_moments` (summed over the ranks of a sharded run):
    mean = x0 + s / n, cov = (S - n m m^T) / (n - 1).  No successful run: NaNs; one: the covariance is NaN."""
    d = 9
    n, sx = float(mom[0]), np.asarray(mom[1:1 + d], dtype=np.float64)
    if n < 1:
        return np.full(d, np.nan), np.full((d, d), np.nan)
    iu = np.triu_indices(d)
    sxx = np.zeros((d, d))
    sxx[iu] = mom[1 + d:]
    sxx = sxx + np.triu(sxx, 1).T
    m = sx / n
    cov = (sxx - n * np.outer(m, m)) / (n - 1.0) if n > 1 else np.full((d, d), np.nan)
    return m + (np.zeros(d) if x0 is None else np.asarray(x0, dtype=np.float64)), cov


def propagate_sharded(contexts, batch: "_abi.StateBatch", duration_ns: int):
    """``nyx_hip_propagate_batch_sharded``: one batch over several contexts (one per device of the node, driven by this process):
    contiguous index shards, concurrent devices, results in place.  Returns (out, stats)."""
    lib = _abi.load_library()
    out = batch.copy()
    stats = _abi.StatsBatch(batch.n)
    cin, cout, cst = batch.as_c(), out.as_c(), stats.as_c()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib.nyx_hip_propagate_batch_sharded(arr, len(contexts), C.byref(cin), int(duration_ns), C.byref(cout), C.byref(cst), None)
    if rc != 0:
        raise RuntimeError(f"nyx_hip_propagate_batch_sharded failed (rc={rc}): {_abi.last_error()}")
    return out, stats


class GpuContext:
    """RAII wrapper of ``nyx_hip_ctx`` (immutable after creation => shareable like `Arc<Propagator>`)."""

    def __init__(self, compiled: CompiledConfig, device: int = 0, tuning: Optional[_abi.Tuning] = None):
        self._lib = _abi.load_library()
        self.compiled = compiled
        self.tuning = tuning
        h = C.c_void_p()
        cfg = compiled.cfg
        if tuning is not None:  # (a private copy of the descriptor: the compiled config may be shared by several contexts)
            cfg = _abi.Config()
            C.memmove(C.byref(cfg), C.byref(compiled.cfg), C.sizeof(_abi.Config))
            cfg.tuning = C.pointer(tuning)
        rc = self._lib.nyx_hip_ctx_create(C.byref(cfg), device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_ctx_create failed (rc={rc}): {_abi.last_error()}")
        self._h = h

    def set_tuning(self, tuning: Optional[_abi.Tuning]):
        """Launch-time part of the tuning for the following launches (``nyx_hip_ctx_set_tuning``)."""
        rc = self._lib.nyx_hip_ctx_set_tuning(self._h, C.byref(tuning) if tuning is not None else None)
        if rc != 0:
            raise RuntimeError(f"nyx_hip_ctx_set_tuning failed (rc={rc}): {_abi.last_error()}")
        self.tuning = tuning

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nyx_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_column_waves(self, waves: int):
        self._lib.nyx_hip_ctx_set_column_waves(self._h, int(waves))

    def set_stm_layout(self, quad: int):
        """STM kernel layout of the following launches: -1 = by ensemble size (default), 0 = 64 trajectories per workgroup
        with three-partial duals (D3), 1 = quad layout (16 trajectories x 4 lanes, one partial per lane)."""
        self._lib.nyx_hip_debug_set_stm_layout(self._h, int(quad))

    def last_coop_helpers(self) -> int:
        """Helper workgroups of the last launch (0 = no cooperative mode)."""
        return int(self._lib.nyx_hip_last_coop_helpers(self._h))

    def last_kernel_ms(self) -> float:
        return float(self._lib.nyx_hip_last_kernel_ms(self._h))

    def ensemble_moments(self, batch: _abi.StateBatch, status: Optional[np.ndarray] = None, x0: Optional[np.ndarray] = None) -> np.ndarray:
        """``nyx_hip_ensemble_moments``: [count, sum(x - x0) (9), upper triangle of sum((x - x0)(x - x0)^T) (45)] of the runs whose
        status is 0, reduced on the device (what the consumers of mc/results.rs:60-245 need of an ensemble; a sharded run adds the
        55 numbers of its ranks with one all-reduce).  `moments_to_mean_cov` turns them into mean and covariance."""
        out = np.zeros(55)
        cin = batch.as_c()
        st = None if status is None else np.ascontiguousarray(status, dtype=np.int32)
        x = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
        rc = self._lib.nyx_hip_ensemble_moments(self._h, C.byref(cin), None if st is None else st.ctypes.data_as(_abi.c_int32_p),
                                                None if x is None else x.ctypes.data_as(_abi.c_double_p), out.ctypes.data_as(_abi.c_double_p))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_ensemble_moments failed (rc={rc}): {_abi.last_error()}")
        return out

    def propagate(self, batch: _abi.StateBatch, duration_ns: int, out: Optional[_abi.StateBatch] = None):
        out = out if out is not None else batch.copy()
        stats = _abi.StatsBatch(batch.n)
        cin, cout, cst = batch.as_c(), out.as_c(), stats.as_c()
        rc = self._lib.nyx_hip_propagate_batch(self._h, C.byref(cin), int(duration_ns), C.byref(cout), C.byref(cst))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_propagate_batch failed (rc={rc}): {_abi.last_error()}")
        return out, stats

    def propagate_with_traj(self, batch: _abi.StateBatch, duration_ns: int, capacity: int, grow: bool = True):
        """Batch form of `for_duration_with_traj` (instance.rs:297-326): final states, stats and the accepted states.
        The kernel stops STORING at `capacity` accepted states per run but keeps counting (`traj.len`): with `grow` (default)
        the launch is repeated with the needed capacity, so no caller ever sees a clipped trajectory; `grow=False` raises."""
        cap = max(int(capacity), 1)
        while True:
            out = batch.copy()
            stats = _abi.StatsBatch(batch.n)
            traj = _abi.TrajBatch(batch.n, cap)
            cin, cout, cst, ctr = batch.as_c(), out.as_c(), stats.as_c(), traj.as_c()
            rc = self._lib.nyx_hip_propagate_batch_with_traj(self._h, C.byref(cin), int(duration_ns), C.byref(cout), C.byref(cst), C.byref(ctr))
            if rc != 0:
                raise RuntimeError(f"nyx_hip_propagate_batch_with_traj failed (rc={rc}): {_abi.last_error()}")
            need = int(traj.len.max()) if batch.n else 0
            if need <= cap:
                return out, stats, traj
            if not grow:
                raise OverflowError(f"dense output needs {need} states per run, capacity is {cap}")
            cap = max(2 * cap, need)

    def propagate_until_event(self, batch: _abi.StateBatch, max_duration_ns: int, event: "Event", trigger: int = 1, capacity: int = 4096):
        """Batch form of `until_nth_event` (propagators/event.rs:88-211): (states at the event, stats, TrajBatch, crossings).
        stats.status is ERR_EVENT_NOT_FOUND where `max_duration_ns` elapsed first (the reference's NthEventError)."""
        cap = max(int(capacity), 2)
        while True:
            out = batch.copy()
            stats = _abi.StatsBatch(batch.n)
            traj = _abi.TrajBatch(batch.n, cap)
            crossings = np.zeros(batch.n, dtype=np.int32)
            cin, cout, cst, ctr, cev = batch.as_c(), out.as_c(), stats.as_c(), traj.as_c(), event.as_c(trigger)
            rc = self._lib.nyx_hip_propagate_until_event(self._h, C.byref(cin), int(max_duration_ns), C.byref(cev), C.byref(cout), C.byref(cst),
                                                         C.byref(ctr), crossings.ctypes.data_as(_abi.c_int32_p))
            if rc != 0:
                raise RuntimeError(f"nyx_hip_propagate_until_event failed (rc={rc}): {_abi.last_error()}")
            need = int(traj.len.max()) if batch.n else 0
            # the search appends the end state to the run's trajectory (event.rs:179): it needs len < capacity; a buffer
            # that is too small would have turned the root search into ERR_EVENT_SEARCH: grow and repeat instead
            if need < cap:
                return out, stats, traj, crossings
            cap = max(2 * cap, need + 1)

    def traj_at(self, traj: _abi.TrajBatch, epochs_ns):
        """`Traj::at(epoch)` (traj.rs:82-127) of every trajectory of the batch at the shared epochs:
        (states[m, n, 6], status[m, n]) with status = nyx_hip_interp_status; failed samples are NaN."""
        q = np.ascontiguousarray(epochs_ns, dtype=np.int64)
        m = len(q)
        out = _abi.TrajBatch(traj.n, max(m, 1))
        status = np.zeros((max(m, 1), traj.n), dtype=np.int32)
        cin, cout = traj.as_c(), out.as_c()
        rc = self._lib.nyx_hip_traj_at(self._h, C.byref(cin), traj.n, q.ctypes.data_as(_abi.c_int64_p), m, C.byref(cout),
                                       status.ctypes.data_as(_abi.c_int32_p))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_traj_at failed (rc={rc}): {_abi.last_error()}")
        return np.ascontiguousarray(out.state[:, :m, :].transpose(1, 2, 0)), status[:m]

    def traj_every(self, traj: _abi.TrajBatch, step_ns: int, capacity: int) -> _abi.TrajBatch:
        """`Traj::every(step)` (traj.rs:148-162) of every trajectory: a TrajBatch of the resampled states."""
        out = _abi.TrajBatch(traj.n, int(capacity))
        cin, cout = traj.as_c(), out.as_c()
        rc = self._lib.nyx_hip_traj_every(self._h, C.byref(cin), traj.n, int(step_ns), C.byref(cout))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_traj_every failed (rc={rc}): {_abi.last_error()}")
        return out

    def propagate_until_epoch(self, batch: _abi.StateBatch, end_epoch_ns: int, out: Optional[_abi.StateBatch] = None):
        out = out if out is not None else batch.copy()
        stats = _abi.StatsBatch(batch.n)
        cin, cout, cst = batch.as_c(), out.as_c(), stats.as_c()
        rc = self._lib.nyx_hip_propagate_until_epoch(self._h, C.byref(cin), int(end_epoch_ns), C.byref(cout), C.byref(cst))
        if rc != 0:
            raise RuntimeError(f"nyx_hip_propagate_until_epoch failed (rc={rc}): {_abi.last_error()}")
        return out, stats


class PropInstance:
    """instance.rs:62-352 for one state, executed as a batch of one on the device."""

    def __init__(self, prop: "Propagator", state: Spacecraft, almanac: Almanac):
        self.prop = prop
        self.state = state
        self.almanac = almanac
        self.step_size = prop.opts.init_step  # propagator.rs:105
        self.details = dict(step=prop.opts.init_step, error=0.0, attempts=1)
        self._ctx = prop._context(almanac, state.frame, state.stm is not None)

    def quiet(self):
        return self

    def for_duration(self, duration_ns: int) -> Spacecraft:
        batch = pack_spacecraft([self.state], self.state.stm is not None)
        batch.step_ns[0] = self.step_size
        out, st = self._ctx.propagate(batch, int(duration_ns))
        if st.status[0] != _abi.OK:
            raise PropagationError(int(st.status[0]))
        self.step_size = int(out.step_ns[0])
        self.details = dict(step=int(st.last_step_ns[0]), error=float(st.last_error[0]), attempts=int(st.last_attempts[0]))
        self.state = unpack_spacecraft(out, [self.state])[0]
        return self.state

    def until_epoch(self, end_epoch_ns: int) -> Spacecraft:
        return self.for_duration(int(end_epoch_ns) - int(self.state.epoch_ns))

    def for_duration_with_traj(self, duration_ns: int, capacity: int = 1 << 16):
        """instance.rs:297-326: (end state, Traj) — the trajectory is (epochs, states) sorted by epoch (`finalize`)."""
        batch = pack_spacecraft([self.state], False)
        batch.step_ns[0] = self.step_size
        out, st, traj = self._ctx.propagate_with_traj(batch, int(duration_ns), capacity)
        if st.status[0] != _abi.OK:
            raise PropagationError(int(st.status[0]))
        self.step_size = int(out.step_ns[0])
        self.state = unpack_spacecraft(out, [self.state])[0]
        return self.state, Traj(self._ctx, traj, 0)

    def until_nth_event(self, max_duration_ns: int, event: "Event", trigger: int, capacity: int = 1 << 16):
        """event.rs:88-211: (state at the event, Traj up to the end of the step where it occurred)."""
        batch = pack_spacecraft([self.state], False)
        batch.step_ns[0] = self.step_size
        out, st, traj, _ = self._ctx.propagate_until_event(batch, int(max_duration_ns), event, trigger, capacity)
        if st.status[0] != _abi.OK:
            raise PropagationError(int(st.status[0]))
        self.state = unpack_spacecraft(out, [self.state])[0]
        return self.state, Traj(self._ctx, traj, 0)

    def until_event(self, max_duration_ns: int, event: "Event", capacity: int = 1 << 16):
        return self.until_nth_event(max_duration_ns, event, 1, capacity)

    def latest_details(self):
        return dict(self.details)


@dataclass
class Event:
    """anise `Event{scalar, Condition::Equals(desired)}` as the stop condition of `until_nth_event`
    (propagators/event.rs:88-211).  The precisions of the root search are explicit (anise keeps them in the Event)."""

    scalar: int                       # _abi.EV_*
    desired: float
    value_precision: float = 1e-7           # deg / km / km/s: tight enough for the reference's own 1e-6 deg assertions
    epoch_precision_ns: int = 1_000         # 1 us: a bracket narrower than this without a hit is "not found"
    frame: Optional[Frame] = None           # until_nth_event's `event_frame` (event.rs:104-117): an IAU-oriented frame of the same centre

    @classmethod
    def less_than(cls, scalar: int, value: float, **kw):
        """`Condition::LessThan(value)`.  The closure of `until_nth_event` (event.rs:124-141) looks at SIGN CHANGES of the event's
        value only (`y_prev * y_next < 0.0`, either direction) and the root search then finds where it vanishes: for a non-angle
        scalar the propagation stops at the same crossing of `value` as `Equals(value)` - which is what this builds.
        UNVERIFIED against the reference: `Event::eval` for this condition is in the anise crate (not in the reference tree); see
        INTEGRATION.md, "Stop conditions"."""
        if scalar in (_abi.EV_TRUE_ANOMALY_DEG, _abi.EV_LONGITUDE_DEG):
            raise NotImplementedError("LessThan / GreaterThan on an angle: the wrapped difference has no such reading")
        return cls(scalar, value, **kw)

    greater_than = less_than   # (the same crossings: the counter does not see a direction)

    @classmethod
    def between(cls, scalar: int, lo: float, hi: float, **kw):
        raise NotImplementedError("Condition::Between: two boundaries, two crossings per pass - refused on the device path (INTEGRATION.md)")

    @classmethod
    def apoapsis(cls):
        return cls(_abi.EV_TRUE_ANOMALY_DEG, 180.0)

    @classmethod
    def periapsis(cls):
        return cls(_abi.EV_TRUE_ANOMALY_DEG, 0.0)

    def as_c(self, trigger: int) -> "_abi.EventC":
        c = _abi.EventC()
        c.scalar, c.trigger, c.desired = int(self.scalar), int(trigger), float(self.desired)
        c.value_precision, c.epoch_precision_ns = float(self.value_precision), int(self.epoch_precision_ns)
        if self.frame is not None:
            rot = self.frame.rotation or Rotation()
            if rot.euler is not None:
                raise NotImplementedError("event frames are IAU-oriented frames on the device path")
            c.has_frame = 1 if self.frame.rotation is not None else 0
            c.frame_eq_radius_km, c.frame_flattening = float(self.frame.mean_equatorial_radius_km), float(self.frame.flattening)
            for k in range(3):
                c.frame.ra_deg[k], c.frame.dec_deg[k], c.frame.w_deg[k] = float(rot.ra_deg[k]), float(rot.dec_deg[k]), float(rot.w_deg[k])
            c.frame.n_nut_prec = len(rot.nut_prec_angles_deg)
            for k in range(c.frame.n_nut_prec):
                c.frame.nut_prec_angle_deg[k][0], c.frame.nut_prec_angle_deg[k][1] = float(rot.nut_prec_angles_deg[k][0]), float(rot.nut_prec_angles_deg[k][1])
                c.frame.nut_prec_ra[k] = float(rot.nut_prec_ra[k]) if k < len(rot.nut_prec_ra) else 0.0
                c.frame.nut_prec_dec[k] = float(rot.nut_prec_dec[k]) if k < len(rot.nut_prec_dec) else 0.0
                c.frame.nut_prec_w[k] = float(rot.nut_prec_w[k]) if k < len(rot.nut_prec_w) else 0.0
        return c


class TrajError(Exception):
    """TrajError::NoInterpolationData / InterpolationError (md/trajectory/mod.rs)."""

    def __init__(self, status: int, epoch_ns: int):
        super().__init__(f"{['Ok', 'NoInterpolationData', 'InterpMath'][status]} at epoch {epoch_ns} ns")
        self.status, self.epoch_ns = status, epoch_ns


class Traj:
    """One run's `Traj<Spacecraft>` (md/trajectory/traj.rs:40-162): the stored states sorted by epoch, evaluated on
    the device.  Unpacks as `(epochs_ns, states[len, 6])` for callers that only want the stored samples."""

    def __init__(self, ctx: "GpuContext", batch: _abi.TrajBatch, index: int = 0):
        self._ctx, self._batch, self._i = ctx, batch, index
        ep, xs = batch.trajectory(index)
        self._finalize(ep, xs)

    def _finalize(self, ep, xs):
        order = np.argsort(ep, kind="stable")                       # finalize(): sort_by_key(epoch) ...
        keep = np.concatenate([[True], np.diff(ep[order]) != 0]) if len(ep) else np.zeros(0, dtype=bool)
        self.epochs_ns, self.states = ep[order][keep], xs[order][keep]  # ... after dedup_by epoch (traj.rs:76-79)

    @classmethod
    def from_arrays(cls, ctx, epochs_ns, states) -> "Traj":
        """A trajectory from stored samples (`Traj::new()` + pushes + `finalize()`); `ctx` evaluates it (traj_at / traj_every)."""
        t = cls.__new__(cls)
        t._ctx, t._batch, t._i = ctx, None, 0
        t._finalize(np.asarray(epochs_ns, dtype=np.int64).reshape(-1), np.asarray(states, dtype=np.float64).reshape(-1, 6))
        return t

    def __iter__(self):
        return iter((self.epochs_ns, self.states))

    def __len__(self):
        return len(self.epochs_ns)

    def first(self):
        return self.states[0]

    def last(self):
        return self.states[-1]

    def _single(self) -> _abi.TrajBatch:
        one = _abi.TrajBatch(1, max(len(self.epochs_ns), 1))
        one.len[0] = len(self.epochs_ns)
        one.epoch_ns[: len(self), 0] = self.epochs_ns
        one.state[:, : len(self), 0] = self.states.T
        return one

    def at(self, epoch_ns: int) -> np.ndarray:
        """traj.rs:82-127: the state [x, y, z, vx, vy, vz] at `epoch_ns`; raises TrajError outside the trajectory."""
        states, status = self._ctx.traj_at(self._single(), [int(epoch_ns)])
        if _abi.interp_failed(status[0, 0]):
            raise TrajError(int(status[0, 0]), int(epoch_ns))
        return states[0, 0]

    def every(self, step_ns: int):
        """traj.rs:148-150: (epochs, states) every `step_ns` from the first to the last epoch, inclusive."""
        if len(self) == 0:
            return np.zeros(0, dtype=np.int64), np.zeros((0, 6))
        count = int((self.epochs_ns[-1] - self.epochs_ns[0]) // int(step_ns)) + 1
        out = self._ctx.traj_every(self._single(), int(step_ns), count)
        return out.trajectory(0)

    def start_epoch(self) -> int:
        return int(self.epochs_ns[0])

    def end_epoch(self) -> int:
        return int(self.epochs_ns[-1])

    def every_between(self, step_ns: int, start_ns: int, end_ns: int):
        """traj.rs:153-162: the inclusive series from max(start, first epoch) to min(end, last epoch); it stops at the first
        epoch that cannot be interpolated (traj_it.rs:39-61)."""
        if len(self) == 0:
            return np.zeros(0, dtype=np.int64), np.zeros((0, 6))
        lo, hi = max(int(start_ns), self.start_epoch()), min(int(end_ns), self.end_epoch())
        if hi < lo:
            return np.zeros(0, dtype=np.int64), np.zeros((0, 6))
        q = lo + int(step_ns) * np.arange((hi - lo) // int(step_ns) + 1, dtype=np.int64)
        states, status = self._ctx.traj_at(self._single(), q)
        bad = np.nonzero(_abi.interp_failed(status[:, 0]))[0]
        n = int(bad[0]) if len(bad) else len(q)
        return q[:n], states[:n, 0]

    def filter_by_epoch(self, start_ns: Optional[int] = None, end_ns: Optional[int] = None, end_inclusive: bool = True) -> "Traj":
        """traj.rs:165-173: the stored states whose epoch lies in the range (a Rust RangeBounds: `a..b`, `a..=b`, `..`)."""
        keep = np.ones(len(self), dtype=bool)
        if start_ns is not None:
            keep &= self.epochs_ns >= int(start_ns)
        if end_ns is not None:
            keep &= (self.epochs_ns <= int(end_ns)) if end_inclusive else (self.epochs_ns < int(end_ns))
        return Traj.from_arrays(self._ctx, self.epochs_ns[keep], self.states[keep])

    def filter_by_offset(self, start_offset_ns: Optional[int] = None, end_offset_ns: Optional[int] = None) -> "Traj":
        """traj.rs:177-193: offsets from the FIRST epoch; both bounds end up inclusive whatever the range type was
        (`filter_by_epoch(start..=end)`)."""
        if len(self) == 0:
            return self
        start = self.start_epoch() + (0 if start_offset_ns is None else int(start_offset_ns))
        end = self.end_epoch() if end_offset_ns is None else self.start_epoch() + int(end_offset_ns)
        return self.filter_by_epoch(start, end, True)

    def resample(self, step_ns: int) -> "Traj":
        """traj.rs:367-384: a new trajectory of the states `every(step)`."""
        if len(self) == 0:
            raise TrajError(1, 0)   # CreationError: "No trajectory to convert"
        ep, xs = self.every(step_ns)
        return Traj.from_arrays(self._ctx, ep, xs)

    def rebuild(self, epochs_ns) -> "Traj":
        """traj.rs:388-407: a new trajectory of `at(epoch)` for the given epochs; the first failure is the error."""
        if len(self) == 0:
            raise TrajError(1, 0)
        q = np.asarray(epochs_ns, dtype=np.int64).reshape(-1)
        if len(q) == 0:
            return Traj.from_arrays(self._ctx, q, np.zeros((0, 6)))
        states, status = self._ctx.traj_at(self._single(), q)
        bad = np.nonzero(_abi.interp_failed(status[:, 0]))[0]
        if len(bad):
            raise TrajError(int(status[bad[0], 0]), int(q[bad[0]]))
        return Traj.from_arrays(self._ctx, q, states[:, 0])


_ARRAY_DIGESTS: dict = {}


def _array_sample(a: np.ndarray) -> bytes:
    """A cheap fingerprint of a large array - its first and last KiB and a comb of 256 elements - to tell a buffer whose ADDRESS is
    being reused by another table from the one a memoised digest belongs to."""
    flat = a.reshape(-1) if a.flags.c_contiguous else np.ascontiguousarray(a).reshape(-1)
    k = max(1, 1024 // max(flat.itemsize, 1))
    comb = flat[:: max(1, flat.size // 256)][:256]
    return hashlib.blake2b(flat[:k].tobytes() + flat[-k:].tobytes() + np.ascontiguousarray(comb).tobytes(), digest_size=8).digest()


def _array_digest(a: np.ndarray) -> bytes:
    """Digest of an array's content (blake2b over the whole buffer: ~1 ms per megabyte).  The digest is memoised ONLY for
    arrays that cannot change under it - `a.flags.writeable == False` all the way down to the buffer's owner (freeze a table
    with `arr.setflags(write=False)` to get the cached path) - keyed on (address, shape, strides, type) AND tied to the buffer's
    owner: the entry holds a weak reference to the owning object and is dropped when that object dies, so a later table that the
    allocator places at the same address (free JGM3, load another 70x70 field: the same shape, quite possibly the same address)
    can never inherit it; where the owner cannot be weakly referenced, and as a second line in any case, a hit is revalidated
    against a cheap sample of the content (first and last KiB + a comb of 256 elements).  A writable array is hashed in full every
    time: an in-place edit of one Stokes coefficient must give a new context, never a stale one."""
    import weakref

    def owner_of(x):
        frozen = True
        while isinstance(x, np.ndarray):
            if x.flags.writeable:
                frozen = False
            if x.base is None:
                break
            x = x.base
        return x, frozen   # (owner read-only: nobody holds a writable view through numpy)
    key = (a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)
    owner, frozen = owner_of(a)
    if a.nbytes <= 4096 or not frozen:
        _ARRAY_DIGESTS.pop(key, None)   # (a buffer seen writable may be edited before it is frozen again: forget what was known of it)
        return hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=16).digest()
    hit = _ARRAY_DIGESTS.get(key)
    if hit is not None:
        ref, digest, sample = hit
        alive = ref is None or ref() is owner
        if alive and sample == _array_sample(a):
            return digest
        _ARRAY_DIGESTS.pop(key, None)
    d = hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=16).digest()
    if len(_ARRAY_DIGESTS) > 256:
        _ARRAY_DIGESTS.clear()
    try:
        ref = weakref.ref(owner, lambda _r, k=key: _ARRAY_DIGESTS.pop(k, None))
    except TypeError:
        ref = None   # (bytes, mmap, ...: the sample check alone guards the entry)
    _ARRAY_DIGESTS[key] = (ref, d, _array_sample(a))
    return d


def _feed(h, obj) -> None:
    """Content fingerprint of a model tree (dataclasses, containers, numpy arrays, scalars): what the context cache is
    keyed on, so that ANY change of the dynamics, the options, the frame or the almanac's tables builds a new context."""
    if isinstance(obj, np.ndarray):
        h.update(str((obj.dtype.str, obj.shape)).encode())
        h.update(_array_digest(obj))
    elif dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        h.update(type(obj).__name__.encode())
        for f in dataclasses.fields(obj):
            h.update(f.name.encode())
            _feed(h, getattr(obj, f.name))
    elif isinstance(obj, Almanac):
        _feed(h, obj.segments)
        _feed(h, sorted(obj.bodies.items()))
    elif isinstance(obj, dict):
        _feed(h, sorted(obj.items()))
    elif isinstance(obj, (list, tuple)):
        h.update(b"[")
        for x in obj:
            _feed(h, x)
        h.update(b"]")
    elif isinstance(obj, float):
        h.update(np.float64(obj).tobytes())
    else:
        h.update(repr(obj).encode())
    h.update(b";")


class Propagator:
    """propagator.rs:34-121."""

    def __init__(self, dynamics: SpacecraftDynamics, method: IntegratorMethod, opts: IntegratorOptions, device: int = 0):
        self.dynamics, self.method, self.opts, self.device = dynamics, method, opts, device
        self._ctx_cache = {}

    new = classmethod(lambda cls, dynamics, method, opts: cls(dynamics, method, opts))

    @classmethod
    def rk89(cls, dynamics, opts):
        return cls(dynamics, IntegratorMethod.RungeKutta89, opts)

    @classmethod
    def dp78(cls, dynamics, opts):
        return cls(dynamics, IntegratorMethod.DormandPrince78, opts)

    @classmethod
    def default(cls, dynamics):
        return cls.rk89(dynamics, IntegratorOptions())

    @classmethod
    def default_dp78(cls, dynamics):
        return cls.dp78(dynamics, IntegratorOptions())

    def set_tolerance(self, tol: float):
        self.opts.tolerance = tol
        self._ctx_cache.clear()

    def set_max_step(self, step: int):
        self.opts.set_max_step(step)
        self._ctx_cache.clear()

    def set_min_step(self, step: int):
        self.opts.set_min_step(step)
        self._ctx_cache.clear()

    def compile(self, almanac: Almanac, central: Frame, stm: bool = False, state_frame: Optional[Frame] = None,
                stm_textbook: bool = False) -> CompiledConfig:
        """`central`: the frame the dynamics integrate in; `state_frame`: the frame the states come in when it is another one
        (`opts.integration_frame`, instance.rs:117-142: translated in at the start, back at the end).  `stm_textbook` (with `stm`):
        the variational equations dPhi/dt = A Phi instead of the reference's Phi_ctx * A (NYX_HIP_FLAG_STM_TEXTBOOK)."""
        return compile_config(self.dynamics, self.method, self.opts, almanac, central, stm=stm, stm_textbook=stm_textbook, state_frame=state_frame)

    def _context(self, almanac: Almanac, central: Frame, stm: bool) -> GpuContext:
        """Cached device context.  The key is a CONTENT fingerprint (dynamics, method, options, the full central frame, the
        almanac's bodies and segment tables): `prop.opts.tolerance = ...`, a mutated almanac or a new one at a recycled
        address all miss the cache instead of silently reusing a stale compiled configuration."""
        # `central` is the frame of the states; with opts.integration_frame set to another body the dynamics run there
        state_frame = None
        integ = self.opts.integration_frame
        if integ is not None and integ.naif_id != central.naif_id:
            state_frame, central = central, integ
        h = hashlib.blake2b(digest_size=16)
        _feed(h, (self.dynamics, int(self.method), self.opts, central, state_frame, bool(stm), int(self.device)))
        _feed(h, almanac)
        key = h.digest()
        if key not in self._ctx_cache:
            if len(self._ctx_cache) >= 8:   # bounded: forget the oldest context (device tables are a few MB each).  NOT closed here:
                # PropInstance and Traj objects may still hold it; GpuContext.__del__ destroys it with its last user
                self._ctx_cache.pop(next(iter(self._ctx_cache)))
            self._ctx_cache[key] = GpuContext(self.compile(almanac, central, stm, state_frame), self.device)
        return self._ctx_cache[key]

    def with_(self, state: Spacecraft, almanac: Almanac) -> PropInstance:
        """``Propagator::with`` (propagator.rs:88-108)."""
        return PropInstance(self, state, almanac)

    def many_until_event(self, spacecraft: Sequence[Spacecraft], almanac: Almanac, max_duration_ns: int, event: "Event", trigger: int = 1,
                         drop_failed: bool = True, capacity: int = 4096):
        """``Propagator.many_until_event`` (nyx-py/src/py_md.rs:324-370): the states at the n-th event; like the reference,
        runs that fail (event not found included) are dropped unless ``drop_failed=False``."""
        if len(spacecraft) == 0:
            return []
        ctx = self._context(almanac, spacecraft[0].frame, False)
        batch = pack_spacecraft(list(spacecraft), False)
        out, st, _, _ = ctx.propagate_until_event(batch, int(max_duration_ns), event, trigger, capacity)
        res = unpack_spacecraft(out, spacecraft)
        if drop_failed:
            return [r for r, s in zip(res, st.status) if s == _abi.OK]
        return [r if s == _abi.OK else PropagationError(int(s), i) for i, (r, s) in enumerate(zip(res, st.status))]

    def many_for_duration(self, spacecraft: Sequence[Spacecraft], almanac: Almanac, duration_ns: int, drop_failed: bool = True):
        """``Propagator.many_for_duration`` (nyx-py/src/py_md.rs:275-321): like the reference, failed runs
        are dropped (py_md.rs:312-315) unless ``drop_failed=False`` in which case a PropagationError is put in place."""
        if len(spacecraft) == 0:
            return []
        stm = spacecraft[0].stm is not None
        ctx = self._context(almanac, spacecraft[0].frame, stm)
        batch = pack_spacecraft(spacecraft, stm)
        out, st = ctx.propagate(batch, int(duration_ns))
        res = unpack_spacecraft(out, spacecraft)
        outl = []
        for i, s in enumerate(res):
            if st.status[i] == _abi.OK:
                outl.append(s)
            elif not drop_failed:
                outl.append(PropagationError(int(st.status[i]), i))
        return outl

    def many_until_epoch(self, spacecraft: Sequence[Spacecraft], almanac: Almanac, end_epoch_ns: int, drop_failed: bool = True):
        """py_md.rs:225-273."""
        if len(spacecraft) == 0:
            return []
        stm = spacecraft[0].stm is not None
        ctx = self._context(almanac, spacecraft[0].frame, stm)
        batch = pack_spacecraft(spacecraft, stm)
        out, st = ctx.propagate_until_epoch(batch, int(end_epoch_ns))
        res = unpack_spacecraft(out, spacecraft)
        return [s if st.status[i] == _abi.OK else PropagationError(int(st.status[i]), i)
                for i, s in enumerate(res) if st.status[i] == _abi.OK or not drop_failed]
