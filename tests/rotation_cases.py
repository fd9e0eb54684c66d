"""Builders for the orientation tests: a binary-PCK-like Euler-angle segment fitted to a known rotation (host logic only)."""
import numpy as np
from numpy.polynomial import chebyshev as _cheb

import nyx_amd as nx

DEG = np.pi / 180.0
OBLIQUITY = 84381.448 / 3600.0 * DEG   # ECLIPJ2000 <-> J2000 (IAU 1976), the base frame of the Earth high-precision BPCs


def iau_angles_rad(rot: nx.Rotation, et_s):
    """(phi, delta, w) = (pi/2 + alpha, pi/2 - delta0, W) of an IAU orientation, SPICE TISBOD."""
    et_s = np.asarray(et_s, dtype=np.float64)
    d, T = et_s / 86400.0, et_s / (86400.0 * 36525.0)
    ra = rot.ra_deg[0] + rot.ra_deg[1] * T + rot.ra_deg[2] * T * T
    dec = rot.dec_deg[0] + rot.dec_deg[1] * T + rot.dec_deg[2] * T * T
    w = rot.w_deg[0] + rot.w_deg[1] * d + rot.w_deg[2] * d * d
    for k, (t0, t1) in enumerate(rot.nut_prec_angles_deg):
        th = (t0 + t1 * T) * DEG
        ra = ra + (rot.nut_prec_ra[k] if k < len(rot.nut_prec_ra) else 0.0) * np.sin(th)
        dec = dec + (rot.nut_prec_dec[k] if k < len(rot.nut_prec_dec) else 0.0) * np.cos(th)
        w = w + (rot.nut_prec_w[k] if k < len(rot.nut_prec_w) else 0.0) * np.sin(th)
    return np.array([np.pi / 2 + ra * DEG, np.pi / 2 - dec * DEG, w * DEG])


def r3(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])


def r1(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, s], [0.0, -s, c]])


def dcm_from_angles(a):
    return r3(a[2]) @ r1(a[1]) @ r3(a[0])


def euler_segment_like(rot: nx.Rotation, t0_s: float, days: float, interval_days: float = 1.0, n_coeffs: int = 10) -> nx.ChebySegment:
    """A PCK-type-2-style segment ([mid, radius, A1.., A2.., A3..], radians) of the Euler angles of `rot`: what a BPC holds
    for ITRF93 / MOON_PA, here generated from an orientation whose DCM is known in closed form."""
    n_rec = int(np.ceil(days / interval_days))
    interval = interval_days * 86400.0
    recs = np.zeros((n_rec, 2 + 3 * n_coeffs))
    k = np.arange(n_coeffs)
    nodes = np.cos(np.pi * (k + 0.5) / n_coeffs)
    for r in range(n_rec):
        mid, radius = t0_s + (r + 0.5) * interval, interval / 2.0
        vals = iau_angles_rad(rot, mid + radius * nodes)
        recs[r, 0], recs[r, 1] = mid, radius
        for c in range(3):
            recs[r, 2 + c * n_coeffs: 2 + (c + 1) * n_coeffs] = _cheb.chebfit(nodes, vals[c], n_coeffs - 1)
    return nx.ChebySegment(t0_s, interval, recs)


def euler_rotation_like(rot: nx.Rotation, t0_s: float, days: float, base=None) -> nx.Rotation:
    """The same orientation as `rot`, expressed the way a BPC does; with `base` (3x3, integration frame -> base frame) the
    segment is that of base->fixed, i.e. the composition gives `rot` again only if base is the identity."""
    b = np.eye(3) if base is None else np.asarray(base)
    return nx.Rotation(euler=euler_segment_like(rot, t0_s, days), base_dcm=tuple(b.ravel()))
