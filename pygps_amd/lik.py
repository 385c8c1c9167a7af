"""Likelihoods on the path (reference: pyGPs/Core/lik.py -- Gauss :123-198, Erf :236-366).

``Gauss`` is the type gate and noise source of Exact inference (Core/inf.py:354,360) and supplies
the predictive moments (Core/gp.py:422-427); ``Erf`` supplies the probit predictive and the EP site
moments.  All of it is O(N) scalar host work."""
import numpy as np
from scipy.special import erf as _erf


class Likelihood(object):
    def __init__(self):
        self.hyp = []

    def evaluate(self, y=None, mu=None, s2=None, inffunc=None, der=None, nargout=1):
        raise NotImplementedError


def _take(values, nargout):
    return values[0] if nargout <= 1 else tuple(values[:nargout])


class Gauss(Likelihood):
    """hyp = [log_sigma]"""

    def __init__(self, log_sigma=np.log(0.1)):
        self.hyp = [log_sigma]

    def evaluate(self, y=None, mu=None, s2=None, inffunc=None, der=None, nargout=1):
        from . import inf
        sn2 = np.exp(2. * self.hyp[0])
        if inffunc is None:                                   # prediction mode (lik.py:134-158)
            if y is None:
                y = np.zeros_like(mu)
            if s2 is not None and np.linalg.norm(s2) > 0:
                lp = self.evaluate(y, mu, s2, inf.EP())
            else:
                lp = -(y - mu) ** 2 / sn2 / 2 - np.log(2. * np.pi * sn2) / 2.
                s2 = np.zeros_like(s2) if s2 is not None else 0.0
            return _take((lp, mu, s2 + sn2), nargout)
        if isinstance(inffunc, inf.EP):                       # lik.py:160-176
            if der is None:
                v = sn2 + s2
                return _take((-(y - mu) ** 2 / v / 2. - np.log(2 * np.pi * v) / 2., (y - mu) / v, -1 / v), nargout)
            return ((y - mu) ** 2 / (sn2 + s2) - 1) / (1 + s2 / sn2)
        raise Exception("Incorrect inference in lik.Gauss\n")


class Erf(Likelihood):
    """Cumulative Gaussian (probit) likelihood for labels in {+1,-1}; no hyper-parameters."""

    def __init__(self):
        self.hyp = []

    @staticmethod
    def _logphi(z, p):
        """log Phi(z) with the asymptotic branch below -6.2 and a blend on [-6.2,-5.5] (lik.py:354-366)."""
        z = np.asarray(z, dtype=float)
        lp = np.zeros_like(z)
        lo, hi = -6.2, -5.5
        safe = z > hi
        far = z < lo
        rest = ~safe
        mid = rest & ~far
        lam = 1. / (1. + np.exp(25. * (0.5 - (z[mid] - lo) / (hi - lo))))
        lp[safe] = np.log(p[safe])
        zr = z[rest]
        lp[rest] = -np.log(np.pi) / 2. - zr ** 2 / 2. - np.log(np.sqrt(zr ** 2 / 2. + 2.) - zr / np.sqrt(2.))
        lp[mid] = (1 - lam) * lp[mid] + lam * np.log(p[mid])
        return lp

    @staticmethod
    def _ratio(f, p):
        """N(f)/Phi(f), switched to its tight upper bound below -6, blended on [-6,-5] (lik.py:341-352)."""
        f = np.asarray(f, dtype=float)
        out = np.zeros_like(f)
        ok = f > -5
        out[ok] = (np.exp(-f[ok] ** 2 / 2) / np.sqrt(2 * np.pi)) / p[ok]
        far = f < -6
        out[far] = np.sqrt(f[far] ** 2 / 4 + 1) - f[far] / 2
        mid = ~ok & ~far
        t = f[mid]
        lam = -5. - t
        out[mid] = (1 - lam) * (np.exp(-t ** 2 / 2) / np.sqrt(2 * np.pi)) / p[mid] + lam * (np.sqrt(t ** 2 / 4 + 1) - t / 2)
        return out

    def cumGauss(self, y=None, f=None, nargout=1):
        yf = f if y is None else y * f
        p = (1. + _erf(yf / np.sqrt(2.))) / 2.
        return (p, self._logphi(yf, p)) if nargout > 1 else p

    def evaluate(self, y=None, mu=None, s2=None, inffunc=None, der=None, nargout=1):
        from . import inf
        if y is not None:
            y = np.sign(y)
            y = np.where(y == 0, 1.0, y)
        else:
            y = 1
        if inffunc is None:                                   # prediction mode (lik.py:251-269)
            y = y * np.ones_like(mu)
            if s2 is not None and np.linalg.norm(s2) > 0:
                lp = self.evaluate(y, mu, s2, inf.EP())
                p = np.exp(lp)
            else:
                p, lp = self.cumGauss(y, mu, 2)
            return _take((lp, 2 * p - 1, 4 * p * (1 - p)), nargout)
        if isinstance(inffunc, inf.EP):                       # lik.py:295-313
            if der is not None:
                return []
            z = mu / np.sqrt(1 + s2)
            lZ = self.cumGauss(y, z, 2)[1]
            if nargout <= 1:
                return lZ
            z = z * y
            n_p = self._ratio(z, np.exp(lZ))
            dlZ = y * n_p / np.sqrt(1. + s2)
            d2lZ = -n_p * (z + n_p) / (1. + s2)
            return _take((lZ, dlZ, d2lZ), nargout)
        raise Exception("Incorrect inference in lik.Erf\n")
