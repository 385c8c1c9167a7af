"""Host-side nonlinear conjugate-gradient minimiser (Polack-Ribiere directions, Wolfe-Powell line
search with cubic/quadratic inter- and extrapolation) -- the algorithm of C. E. Rasmussen's
``minimize.m`` that pyGPs drives its hyper-parameter search with
(reference: pyGPs/Optimization/minimize.py:41-172; constants :49-54).

This is scalar control logic (4 .. 67 parameters); it stays on the host and every objective
evaluation ``f(X) -> (value, gradient)`` is one GPU fit.  The iterates follow the reference's
arithmetic so that a run started from the same point visits the same points:

* ``length > 0``: at most ``length`` line searches; ``length < 0``: at most ``-length`` evaluations;
* at most ``MAX`` = 20 evaluations per line search;
* an evaluation that raises inside the extrapolation phase bisects the step and retries
  (:88-97); a NaN/Inf value or gradient there makes ``run`` return ``None`` (:93-94);
* two consecutive failed line searches terminate the run.

Returns ``(X, fX, i)``: the best point, the list of accepted function values, the count of line
searches (or evaluations).
"""
import numpy as np

INT = 0.1     # do not re-evaluate within 0.1 of the limit of the current bracket
EXT = 3.0     # extrapolate at most 3 times the current step size
MAX = 20      # at most 20 function evaluations per line search
RATIO = 10.0  # maximum allowed slope ratio
SIG = 0.1     # Wolfe-Powell: max |new slope / old slope|
RHO = SIG / 2.0   # Wolfe-Powell: min fraction of the expected decrease
TINY = np.finfo(float).tiny


class _Pt(object):
    """A point on the current search ray: step x, value f, slope d (and gradient g)."""
    __slots__ = ("x", "f", "d", "g")

    def __init__(self, x, f, d, g=None):
        self.x, self.f, self.d, self.g = x, f, d, g


def _bad(value, grad):
    return bool(np.isnan(value) or np.isinf(value) or np.any(np.isnan(grad) + np.isinf(grad)))


def _cubic_extrapolate(p1, p2):
    """Minimiser of the cubic through (p1, p2) beyond p2, clipped to the allowed range."""
    dx = p2.x - p1.x
    A = 6.0 * (p1.f - p2.f) + 3.0 * (p2.d + p1.d) * dx
    B = 3.0 * (p2.f - p1.f) - (2.0 * p1.d + p2.d) * dx
    Z = B + np.sqrt(complex(B * B - A * p1.d * dx))
    x3 = p1.x - p1.d * dx ** 2 / Z if Z != 0.0 else np.inf
    if (not np.isreal(x3)) or np.isnan(x3) or np.isinf(x3) or (x3 < 0):
        x3 = p2.x * EXT                       # numerical trouble or wrong sign: extrapolate maximally
    elif x3 > p2.x * EXT:
        x3 = p2.x * EXT                       # beyond the extrapolation limit
    elif x3 < p2.x + INT * dx:
        x3 = p2.x + INT * dx                  # too close to the previous point
    return np.real(x3)


def _interpolate(p2, p4, f0):
    """New trial step inside the bracket [p2, p4]."""
    w = p4.x - p2.x
    if p4.f > f0:                             # quadratic through p2 (value+slope) and p4 (value)
        x3 = p2.x - (0.5 * p2.d * w ** 2) / (p4.f - p2.f - p2.d * w)
    else:                                     # cubic through both
        A = 6.0 * (p2.f - p4.f) / w + 3.0 * (p4.d + p2.d)
        B = 3.0 * (p4.f - p2.f) - (2.0 * p2.d + p4.d) * w
        x3 = p2.x + (np.sqrt(B * B - A * p2.d * w ** 2) - B) / A if A != 0 else np.inf
    if np.isnan(x3) or np.isinf(x3):
        x3 = (p2.x + p4.x) / 2                # numerical problem: bisect
    return max(min(x3, p4.x - INT * w), p2.x + INT * w)


def run(f, X, args=(), length=None, red=1.0, verbose=False):
    count_evals = length < 0
    budget = abs(length)
    i = 0
    ls_failed = False
    f0, df0 = f(X, *args)[:2]
    fX = [f0]
    i += int(count_evals)
    s = -df0
    d0 = -np.dot(s, s)                        # steepest descent to start with
    x3 = red / (1.0 - d0)                     # initial step red / (|s|^2 + 1)

    while i < budget:
        i += int(not count_evals)
        best_X, best_f, best_g = X, f0, df0   # best point seen in this line search
        M = MAX if not count_evals else min(MAX, -length - i)
        f3, df3 = f0, df0
        # ---- extrapolation: walk out until the minimum is bracketed or Wolfe-Powell holds ----
        p2 = _Pt(0.0, f0, d0)
        while True:
            p2 = _Pt(0.0, f0, d0)
            f3, df3 = f0, df0
            ok = False
            while (not ok) and M > 0:
                try:
                    M -= 1
                    i += int(count_evals)
                    f3, df3 = f(X + x3 * s, *args)[:2]
                    if _bad(f3, df3):
                        return None
                    ok = True
                except Exception:             # any failure inside f: bisect and try again
                    x3 = (p2.x + x3) / 2.0
            if f3 < best_f:
                best_X, best_f, best_g = X + x3 * s, f3, df3
            d3 = np.dot(df3, s)
            if d3 > SIG * d0 or f3 > f0 + x3 * RHO * d0 or M == 0:
                break
            p1, p2 = p2, _Pt(x3, f3, d3)
            x3 = _cubic_extrapolate(p1, p2)
            # the reference re-initialises point 2 at the top of this loop (minimize.py:83), so the
            # next extrapolation is again taken from the origin of the ray
        # ---- interpolation inside the bracket -------------------------------------------------
        p4 = None
        while (abs(d3) > -SIG * d0 or f3 > f0 + x3 * RHO * d0) and M > 0:
            if d3 > 0 or f3 > f0 + x3 * RHO * d0:
                p4 = _Pt(x3, f3, d3)
            else:
                p2 = _Pt(x3, f3, d3)
            x3 = _interpolate(p2, p4, f0)
            f3, df3 = f(X + x3 * s, *args)[:2]
            if f3 < best_f:
                best_X, best_f, best_g = X + x3 * s, f3, df3
            M -= 1
            i += int(count_evals)
            d3 = np.dot(df3, s)

        if abs(d3) < -SIG * d0 and f3 < f0 + x3 * RHO * d0:      # line search succeeded
            X = X + x3 * s
            f0 = f3
            fX.append(f0)
            s = (np.dot(df3, df3) - np.dot(df0, df3)) / np.dot(df0, df0) * s - df3   # Polack-Ribiere
            df0 = df3
            d3 = d0
            d0 = np.dot(df0, s)
            if d0 > 0:                        # not a descent direction: restart with steepest descent
                s = -df0
                d0 = -np.dot(s, s)
            x3 = x3 * min(RATIO, d3 / (d0 - TINY))
            ls_failed = False
        else:
            X, f0, df0 = best_X, best_f, best_g
            if ls_failed or i > budget:       # failed twice in a row, or out of budget
                break
            s = -df0
            d0 = -np.dot(s, s)
            x3 = 1.0 / (1.0 - d0)
            ls_failed = True
    if verbose:
        import logging
        logging.getLogger(__name__).info(str(fX))
    return X, fX, i
