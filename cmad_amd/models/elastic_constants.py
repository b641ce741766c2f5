"""Isotropic elastic constants, any two of {E, nu, mu, kappa, lambda} -> Lame pair
(mirror of /root/reference/cmad/models/elastic_constants.py:23-104).  Host-side only: the kernels
take (lambda, mu); `lame_jacobian` supplies d(lambda, mu)/d(given pair) for the chain rule that maps
kernel sensitivities back onto the user's parameters."""
from dataclasses import dataclass
from typing import Any

import numpy as np

_CONSTANT_NAMES = ("E", "nu", "mu", "kappa", "lambda")


def compute_mu(E, nu):
    return E / (2. * (1. + nu))


def compute_kappa(E, nu):
    return E / (3. * (1. - 2. * nu))


def compute_lambda(E, nu):
    return E * nu / ((1. + nu) * (1. - 2. * nu))


def _sqrt(x):
    return x.sqrt() if hasattr(x, "sqrt") else np.sqrt(x)


@dataclass(frozen=True)
class ElasticConstants:
    lmbda: Any
    mu: Any

    @property
    def kappa(self):
        return self.lmbda + 2. * self.mu / 3.

    @property
    def E(self):
        return self.mu * (3. * self.lmbda + 2. * self.mu) / (self.lmbda + self.mu)

    @property
    def nu(self):
        return self.lmbda / (2. * (self.lmbda + self.mu))

    @classmethod
    def from_params(cls, elastic: dict) -> "ElasticConstants":
        """Any two of {E, nu, mu, kappa, lambda} -> the Lame pair, by table lookup (`_TO_LAME`)."""
        names = sorted(n for n in _CONSTANT_NAMES if n in elastic)
        if len(names) != 2:
            raise ValueError(f"ElasticConstants needs exactly two of {_CONSTANT_NAMES}; got {tuple(names)}")
        lmbda, mu = _TO_LAME[tuple(names)](elastic[names[0]], elastic[names[1]])
        return cls(lmbda=lmbda, mu=mu)


def _from_E_lambda(E, lam):
    root = _sqrt(E * E + 9. * lam * lam + 2. * E * lam)         # the one pair that needs a quadratic root
    return lam, (E - 3. * lam + root) / 4.


# (first, second) in alphabetical order of the names -> (lambda, mu); the textbook conversions between isotropic constants
_TO_LAME = {
    ("E", "nu"): lambda E, nu: (compute_lambda(E, nu), compute_mu(E, nu)),
    ("E", "mu"): lambda E, mu: (mu * (E - 2. * mu) / (3. * mu - E), mu),
    ("E", "kappa"): lambda E, k: (3. * k * (3. * k - E) / (9. * k - E), 3. * k * E / (9. * k - E)),
    ("E", "lambda"): _from_E_lambda,
    ("lambda", "mu"): lambda lam, mu: (lam, mu),
    ("kappa", "mu"): lambda k, mu: (k - 2. * mu / 3., mu),
    ("mu", "nu"): lambda mu, nu: (2. * mu * nu / (1. - 2. * nu), mu),
    ("kappa", "nu"): lambda k, nu: (3. * k * nu / (1. + nu), 3. * k * (1. - 2. * nu) / (2. * (1. + nu))),
    ("lambda", "nu"): lambda lam, nu: (lam, lam * (1. - 2. * nu) / (2. * nu)),
    ("kappa", "lambda"): lambda k, lam: (lam, 3. * (k - lam) / 2.),
}


class _D2:
    """Minimal forward-mode dual with two tangent slots (host only, for lame_jacobian)."""
    __slots__ = ("v", "d")

    def __init__(self, v, d=(0., 0.)):
        self.v, self.d = float(v), np.asarray(d, dtype=float)

    @staticmethod
    def _c(o):
        return o if isinstance(o, _D2) else _D2(o)

    def __add__(self, o): o = self._c(o); return _D2(self.v + o.v, self.d + o.d)
    __radd__ = __add__
    def __sub__(self, o): o = self._c(o); return _D2(self.v - o.v, self.d - o.d)
    def __rsub__(self, o): return self._c(o) - self
    def __mul__(self, o): o = self._c(o); return _D2(self.v * o.v, self.d * o.v + self.v * o.d)
    __rmul__ = __mul__
    def __truediv__(self, o): o = self._c(o); q = self.v / o.v; return _D2(q, (self.d - q * o.d) / o.v)
    def __rtruediv__(self, o): return self._c(o) / self
    def __neg__(self): return _D2(-self.v, -self.d)
    def __pow__(self, n): return _D2(self.v ** n, n * self.v ** (n - 1) * self.d)
    def sqrt(self): r = np.sqrt(self.v); return _D2(r, self.d / (2. * r))


def lame_jacobian(elastic: dict):
    """Return (names, lmbda, mu, J) with J[i, j] = d (lmbda, mu)[i] / d elastic[names[j]]."""
    names = tuple(n for n in _CONSTANT_NAMES if n in elastic)
    if len(names) != 2:
        raise ValueError(f"ElasticConstants needs exactly two of {_CONSTANT_NAMES}; got {names}")
    seeded = {names[0]: _D2(elastic[names[0]], (1., 0.)), names[1]: _D2(elastic[names[1]], (0., 1.))}
    ec = ElasticConstants.from_params(seeded)
    lm, mu = _D2._c(ec.lmbda), _D2._c(ec.mu)
    return names, lm.v, mu.v, np.array([lm.d, mu.d])


class _HD:
    """Minimal hyper-dual (value, d/da, d/db, d2/dadb) for the second-order chain rule of the elastic pair."""
    __slots__ = ("v", "a", "b", "ab")

    def __init__(self, v, a=0., b=0., ab=0.):
        self.v, self.a, self.b, self.ab = float(v), float(a), float(b), float(ab)

    @staticmethod
    def _c(o):
        return o if isinstance(o, _HD) else _HD(o)

    def __add__(self, o): o = self._c(o); return _HD(self.v + o.v, self.a + o.a, self.b + o.b, self.ab + o.ab)
    __radd__ = __add__
    def __neg__(self): return _HD(-self.v, -self.a, -self.b, -self.ab)
    def __sub__(self, o): return self + (-self._c(o))
    def __rsub__(self, o): return self._c(o) - self
    def __mul__(self, o):
        o = self._c(o)
        return _HD(self.v * o.v, self.a * o.v + self.v * o.a, self.b * o.v + self.v * o.b,
                   self.ab * o.v + self.a * o.b + self.b * o.a + self.v * o.ab)
    __rmul__ = __mul__
    def _chain(self, g0, g1, g2): return _HD(g0, g1 * self.a, g1 * self.b, g1 * self.ab + g2 * self.a * self.b)
    def __truediv__(self, o): o = self._c(o); i = 1. / o.v; return self * o._chain(i, -i * i, 2. * i ** 3)
    def __rtruediv__(self, o): return self._c(o) / self
    def __pow__(self, n): return self._chain(self.v ** n, n * self.v ** (n - 1), n * (n - 1) * self.v ** (n - 2))
    def sqrt(self): r = np.sqrt(self.v); return self._chain(r, 0.5 / r, -0.25 / (r * self.v))


def lame_second_derivs(elastic: dict):
    """H[i, j, k] = d2 (lmbda, mu)[i] / d elastic[names[j]] d elastic[names[k]]  (names as in lame_jacobian)."""
    names = tuple(n for n in _CONSTANT_NAMES if n in elastic)
    H = np.zeros((2, 2, 2))
    for j in range(2):
        for k in range(j, 2):
            seeded = {n: _HD(elastic[n]) for n in names}
            seeded[names[j]].a = 1.0
            seeded[names[k]].b = 1.0
            ec = ElasticConstants.from_params(seeded)
            for i, val in enumerate((_HD._c(ec.lmbda), _HD._c(ec.mu))):
                H[i, j, k] = H[i, k, j] = val.ab
    return names, H
