"""Synthetic DE-like planetary ephemeris for the tests and the benchmark.

The reference reads `de440s.bsp` through ANISE; in this repository that file is a git-LFS
pointer, so the harness builds Chebyshev segments with the SAME
structure the path sees with DE440 — SPK type 2 records, an ephemeris tree
Sun->SSB, EMB->SSB, Earth->EMB, Moon->EMB (+ Jupiter barycentre->SSB) that the host flattens
into signed chains — from low-precision analytic Sun/Moon series (Montenbruck & Gill,
"Satellite Orbits", section 3.3.2).  Record lengths / coefficient counts follow DE44x:
Moon & Earth w.r.t. EMB 4-day x 13, EMB 16-day x 13, Sun 16-day x 11, Jupiter 32-day x 8.

The Chebyshev records ARE the ephemeris: oracle and GPU are fed these same tables.
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import chebyshev as _cheb

from .propagator import Almanac, ChebySegment, EARTH, EARTH_MOON_BARYCENTER, JUPITER_BARYCENTER, MOON, SSB, SUN

# GM values printed by the reference's own configs (data/02_config/full_seq.dhall:156,175; README of example 04)
MU_SUN = 132712440041.93938
MU_MOON = 4902.800066163796
MU_EARTH = 398600.435436096
MU_JUPITER_BARY = 126712764.1
R_SUN, R_MOON, R_EARTH = 696000.0, 1737.4, 6378.14  # pck08 radii (full_seq.dhall:159-180)

_ARC = np.pi / (180.0 * 3600.0)
_DEG = np.pi / 180.0
_EPS = 23.43929111 * _DEG


def _ecl_to_eq(v):
    c, s = np.cos(_EPS), np.sin(_EPS)
    x, y, z = v
    return np.array([x, c * y - s * z, s * y + c * z])


def sun_geocentric(et_s):
    """Geocentric Sun, km, equatorial J2000 (M&G eq. 3.43-3.46)."""
    T = np.asarray(et_s, dtype=np.float64) / (86400.0 * 36525.0)
    M = (357.5256 + 35999.049 * T) * _DEG
    lam = (282.94 * _DEG) + M + (6892.0 * np.sin(M) + 72.0 * np.sin(2 * M)) * _ARC
    r = (149.619 - 2.499 * np.cos(M) - 0.021 * np.cos(2 * M)) * 1e6
    return _ecl_to_eq(np.array([r * np.cos(lam), r * np.sin(lam), np.zeros_like(r)]))


def moon_geocentric(et_s):
    """Geocentric Moon, km, equatorial J2000 (M&G eq. 3.47-3.52)."""
    T = np.asarray(et_s, dtype=np.float64) / (86400.0 * 36525.0)
    L0 = (218.31617 + 481267.88088 * T - 1.3972 * T) * _DEG
    l = (134.96292 + 477198.86753 * T) * _DEG
    lp = (357.52543 + 35999.04944 * T) * _DEG
    F = (93.27283 + 483202.01873 * T) * _DEG
    D = (297.85027 + 445267.11135 * T) * _DEG
    dlam = (22640 * np.sin(l) + 769 * np.sin(2 * l) - 4586 * np.sin(l - 2 * D) + 2370 * np.sin(2 * D) - 668 * np.sin(lp)
            - 412 * np.sin(2 * F) - 212 * np.sin(2 * l - 2 * D) - 206 * np.sin(l + lp - 2 * D) + 192 * np.sin(l + 2 * D)
            - 165 * np.sin(lp - 2 * D) + 148 * np.sin(l - lp) - 125 * np.sin(D) - 110 * np.sin(l + lp) - 55 * np.sin(2 * F - 2 * D))
    lam = L0 + dlam * _ARC
    beta = (18520 * np.sin(F + lam - L0 + (412 * np.sin(2 * F) + 541 * np.sin(lp)) * _ARC) - 526 * np.sin(F - 2 * D)
            + 44 * np.sin(l + F - 2 * D) - 31 * np.sin(-l + F - 2 * D) - 25 * np.sin(-2 * l + F) - 23 * np.sin(lp + F - 2 * D)
            + 21 * np.sin(-l + F) + 11 * np.sin(-lp + F - 2 * D)) * _ARC
    r = (385000 - 20905 * np.cos(l) - 3699 * np.cos(2 * D - l) - 2956 * np.cos(2 * D) - 570 * np.cos(2 * l)
         + 246 * np.cos(2 * l - 2 * D) - 205 * np.cos(lp - 2 * D) - 171 * np.cos(l + 2 * D) - 152 * np.cos(l + lp - 2 * D))
    return _ecl_to_eq(np.array([r * np.cos(lam) * np.cos(beta), r * np.sin(lam) * np.cos(beta), r * np.sin(beta)]))


def jupiter_heliocentric(et_s):
    """Circular, ecliptic Jupiter-barycentre stand-in (a = 5.2044 AU, P = 4332.59 d)."""
    t = np.asarray(et_s, dtype=np.float64)
    a = 5.2044 * 149597870.7
    th = (34.35 * _DEG) + 2 * np.pi * t / (4332.59 * 86400.0)
    return _ecl_to_eq(np.array([a * np.cos(th), a * np.sin(th), np.zeros_like(th)]))


def fit_segment(func, t0: float, interval_s: float, n_records: int, n_coeffs: int) -> ChebySegment:
    """Chebyshev interpolation of `func(et)->(3,n)` record by record, SPK type 2 layout."""
    recs = np.zeros((n_records, 2 + 3 * n_coeffs))
    k = np.arange(n_coeffs)
    nodes = np.cos(np.pi * (k + 0.5) / n_coeffs)  # Chebyshev points of the first kind
    radius = interval_s / 2.0
    for r in range(n_records):
        mid = t0 + (r + 0.5) * interval_s
        vals = func(mid + radius * nodes)  # (3, n_coeffs)
        recs[r, 0], recs[r, 1] = mid, radius
        for c in range(3):
            recs[r, 2 + c * n_coeffs: 2 + (c + 1) * n_coeffs] = _cheb.chebfit(nodes, vals[c], n_coeffs - 1)
    return ChebySegment(t0, interval_s, recs)


def build_almanac(et0_s: float, span_days: float = 40.0, with_jupiter: bool = True) -> Almanac:
    """Earth-centred almanac covering [et0 - 8 d, et0 + span]."""
    day = 86400.0
    start = np.floor((et0_s - 8 * day) / (32 * day)) * 32 * day  # aligned like DE record boundaries
    end = et0_s + span_days * day
    k = MU_MOON / (MU_EARTH + MU_MOON)

    def earth_wrt_emb(t):
        return -k * moon_geocentric(t)

    def moon_wrt_emb(t):
        return (1.0 - k) * moon_geocentric(t)

    def sun_wrt_ssb(t):
        return -(MU_JUPITER_BARY / MU_SUN) * jupiter_heliocentric(t)

    def emb_wrt_ssb(t):
        return sun_wrt_ssb(t) - earth_wrt_emb(t) - sun_geocentric(t)

    def jup_wrt_ssb(t):
        return sun_wrt_ssb(t) + jupiter_heliocentric(t)

    def nrec(interval_days):
        return int(np.ceil((end - start) / (interval_days * day)))

    al = Almanac()
    s_sun = al.add_segment(fit_segment(sun_wrt_ssb, start, 16 * day, nrec(16), 11))
    s_emb = al.add_segment(fit_segment(emb_wrt_ssb, start, 16 * day, nrec(16), 13))
    s_earth = al.add_segment(fit_segment(earth_wrt_emb, start, 4 * day, nrec(4), 13))
    s_moon = al.add_segment(fit_segment(moon_wrt_emb, start, 4 * day, nrec(4), 13))
    # chains: position of the body w.r.t. Earth (the integration centre)
    al.add_body(SUN, MU_SUN, R_SUN, [(s_sun, +1), (s_emb, -1), (s_earth, -1)])
    al.add_body(MOON, MU_MOON, R_MOON, [(s_moon, +1), (s_earth, -1)])
    al.add_body(EARTH, MU_EARTH, R_EARTH, [])
    if with_jupiter:
        s_jup = al.add_segment(fit_segment(jup_wrt_ssb, start, 32 * day, nrec(32), 8))
        al.add_body(JUPITER_BARYCENTER, MU_JUPITER_BARY, 71492.0, [(s_jup, +1), (s_emb, -1), (s_earth, -1)])
    return al


def build_moon_centered_almanac(et0_s: float, span_days: float = 40.0) -> Almanac:
    """Same tables, chains expressed w.r.t. the Moon (cislunar / LRO-like configs)."""
    al = build_almanac(et0_s, span_days, with_jupiter=False)
    sun_chain = al.bodies[SUN]["chain"]
    moon_chain = al.bodies[MOON]["chain"]
    s_sun, s_emb, s_earth = sun_chain[0][0], sun_chain[1][0], sun_chain[2][0]
    s_moon = moon_chain[0][0]
    al.bodies = {}
    al.add_body(MOON, MU_MOON, R_MOON, [])
    al.add_body(EARTH, MU_EARTH, R_EARTH, [(s_earth, +1), (s_moon, -1)])
    al.add_body(SUN, MU_SUN, R_SUN, [(s_sun, +1), (s_emb, -1), (s_moon, -1)])
    return al


# 2024-02-29T12:13:14 UTC (examples/01_orbit_prop/main.rs:43) in TDB seconds past J2000:
# 8825 d + 794 s + 69.184 s (TT-UTC; TDB-TT periodic term dropped)
EPOCH_2024_02_29_NS = 762_480_863_184_000_000
