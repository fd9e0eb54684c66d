"""The reference's random stream: `Pcg64Mcg` + ziggurat `StandardNormal`.

`MonteCarlo::generate_states` seeds `rand_pcg::Pcg64Mcg::new(seed)` and every dispersed state draws nine
`Normal(0, 1)` samples from it (nyx-core/src/mc/montecarlo.rs:277-296, mc/multivariate.rs:298-303).  Neither `rand_pcg`
nor `rand_distr` is part of the reference tree (crates.io dependencies, no Cargo.lock in the tree), so their published
algorithms are restated here:

* PCG XSL-RR 128/64 (MCG): state <- state * 0x2360ED051FC65DA44385DF649FCCF645 mod 2^128, output = rotr64(hi ^ lo, state >> 122)
  (O'Neill, "PCG: A Family of Simple Fast Space-Efficient Statistically Good Algorithms", 2014; `Mcg128Xsl64`), seeded with
  `state | 1`.
* Ziggurat (Marsaglia & Tsang 2000; Doornik 2005 layout) with 256 layers, R = 3.654152885361008796, V = 0.00492867323399,
  the tables computed by the recurrence of rand's `ziggurat_tables.py` (x[0] = V / f(R), x[1] = R,
  x[i] = f^-1(V / x[i-1] + f(x[i-1])), x[256] = 0): symmetric variant, u in [-1, 1) from the top 52 bits, layer from the
  low 8 bits; the tail by Marsaglia's exponential rejection with two `Open01` draws per trial.

Pinned by the reference's own known-answer tests, which fix seed 0 and assert exact counts
(`disperse_r_mag`: 6 of 1 000 samples beyond 3 sigma, multivariate.rs:470-475; `disperse_full_cartesian`: 312 = floor(count of
components beyond 1 sigma / 6), :551-556): tests/test_rng_parity.py.
"""
from __future__ import annotations

import math

_MASK64 = (1 << 64) - 1
_MASK128 = (1 << 128) - 1
_MULT = 0x2360ED051FC65DA44385DF649FCCF645

ZIG_NORM_R = 3.654152885361008796
_ZIG_NORM_V = 0.00492867323399


def _tables():
    f = lambda x: math.exp(-x * x / 2.0)          # noqa: E731
    finv = lambda y: math.sqrt(-2.0 * math.log(y))  # noqa: E731
    x = [0.0] * 257
    x[0] = _ZIG_NORM_V / f(ZIG_NORM_R)
    x[1] = ZIG_NORM_R
    for i in range(2, 256):
        last = x[i - 1]
        x[i] = finv(_ZIG_NORM_V / last + f(last))
    x[256] = 0.0
    return x, [f(v) for v in x]


ZIG_NORM_X, ZIG_NORM_F = _tables()


class Pcg64Mcg:
    """rand_pcg::Pcg64Mcg (`Mcg128Xsl64`)."""

    def __init__(self, seed: int):
        self.state = (int(seed) | 1) & _MASK128

    def next_u64(self) -> int:
        self.state = (self.state * _MULT) & _MASK128
        rot = self.state >> 122
        xsl = ((self.state >> 64) ^ self.state) & _MASK64
        return ((xsl >> rot) | (xsl << ((64 - rot) & 63))) & _MASK64

    # ---- rand's float conversions
    def random_f64(self) -> float:
        """`rng.random::<f64>()`: 53 random bits in [0, 1)."""
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))

    def open01(self) -> float:
        """`Open01`: (0, 1): the top 52 bits as the mantissa of a float in [1, 2), minus (1 - eps/2)."""
        frac = self.next_u64() >> 12
        return (1.0 + frac * 2.0 ** -52) - (1.0 - 2.0 ** -53)

    def standard_normal(self) -> float:
        """`StandardNormal` (rand_distr, ziggurat)."""
        while True:
            bits = self.next_u64()
            i = bits & 0xFF
            # a value in [2, 4) from the top 52 bits, minus 3: u in [-1, 1)
            u = (2.0 + (bits >> 12) * 2.0 ** -51) - 3.0
            x = u * ZIG_NORM_X[i]
            if abs(x) < ZIG_NORM_X[i + 1]:
                return x
            if i == 0:
                return self._tail(u)
            if ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * self.random_f64() < math.exp(-x * x / 2.0):
                return x

    def _tail(self, u: float) -> float:
        x, y = 1.0, 0.0
        while -2.0 * y < x * x:
            x_ = self.open01()
            y_ = self.open01()
            x = math.log(x_) / ZIG_NORM_R
            y = math.log(y_)
        return x - ZIG_NORM_R if u < 0.0 else ZIG_NORM_R - x

    def normal_vector(self, n: int):
        return [self.standard_normal() for _ in range(n)]
