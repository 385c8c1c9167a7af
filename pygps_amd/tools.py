"""Numeric helpers (reference: pyGPs/Core/tools.py -- jitchol :31-77, solve_chol :81-97) on the device."""
import numpy as np

from . import _lib


def jitchol(A, maxtries=5):
    """Lower Cholesky factor of the symmetric positive-definite A.  As in the reference the jitter
    branch never rescues a non-PD matrix (it is dead code there): non-PD input raises
    numpy.linalg.LinAlgError."""
    A = _lib.f64(A)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise Exception("jitchol needs a square matrix")
    n = A.shape[0]
    L = np.empty((n, n))
    rc = _lib.load().pgp_potrf(_lib.ctx(), _lib.ptr(A), n, _lib.ptr(L))
    if rc > 0 and np.any(np.diag(A) <= 0.):
        raise np.linalg.LinAlgError("kernel matrix not positive definite: non-positive diagonal elements")
    _lib.check(rc, "pgp_potrf")
    return L


def solve_chol(L, B):
    """X = (L'L)^-1 B for the UPPER factor L."""
    if not (L.shape[0] == L.shape[1] and L.shape[0] == B.shape[0]):
        raise Exception("Wrong sizes of matrix arguments in solve_chol.py")
    L = _lib.f64(L)
    B2 = _lib.f64(B.reshape(B.shape[0], -1))
    X = np.empty_like(B2)
    rc = _lib.load().pgp_potrs(_lib.ctx(), _lib.ptr(L), L.shape[0], _lib.ptr(B2), B2.shape[1], _lib.ptr(X))
    if rc == -99:
        raise NotImplementedError("pygps_amd: device potrs is not built in this version")
    _lib.check(rc, "pgp_potrs")
    return X.reshape(B.shape)
