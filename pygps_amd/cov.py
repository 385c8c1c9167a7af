"""Covariance functions: the drop-in classes for the three kernels on the hot path
(reference: pyGPs/Core/cov.py -- Kernel :61-226, RBF :786-828, RBFard :872-938, Matern :1078-1182).

Same constructor signatures, same ``hyp`` (log-space list, read fresh on every call: the optimiser
overwrites it, Core/opt.py:88) and ``para`` attributes, same ``getCovMatrix(x, z, mode)`` /
``getDerMatrix(x, z, mode, der)`` contract and the same exceptions.  The arithmetic runs in the HIP
tile kernel behind ``pgp_cov`` (csrc/assemble.hip); there is no numpy fallback.
"""
import logging

import numpy as np

from . import _lib

_MODES = {"train": _lib.MODE_TRAIN, "cross": _lib.MODE_CROSS, "self_test": _lib.MODE_SELF_TEST}


class Kernel(object):
    """Base class: argument checks (Core/cov.py:115-154) and operator overloading (:160-202)."""
    _kind = None            # PGP_COV_* for the kernels that have a device functor
    #: Matern only -- reproduce the reference's derivative-of-K quirk (Core/cov.py:1173-1177).
    reference_compat = False

    def __init__(self):
        self.hyp = []
        self.para = []
        self.logger = logging.getLogger(__name__)

    def __repr__(self):
        return (str(type(self)) + ": to get the kernel matrix or kernel derviatives use: \n"
                "model.covfunc.getCovMatrix()\nmodel.covfunc.getDerMatrix()")

    # -- contract ------------------------------------------------------------------------------
    def getCovMatrix(self, x=None, z=None, mode=None):
        raise NotImplementedError

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        raise NotImplementedError

    def checkInputGetCovMatrix(self, x, z, mode):
        if mode is None:
            raise Exception("Specify the mode: 'train' or 'cross'")
        if x is None and z is None:
            raise Exception("Specify at least one: training input (x) or test input (z) or both.")
        if mode == "cross" and (x is None or z is None):
            raise Exception("Specify both: training input (x) and test input (z) for cross covariance.")

    def checkInputGetDerMatrix(self, x, z, mode, der):
        self.checkInputGetCovMatrix(x, z, mode)
        if der is None:
            raise Exception("Specify the index of parameters of the derivatives.")

    # -- composition (host-side sums/products of device-built matrices) -----------------------------
    def __add__(self, other):
        return SumOfKernel(self, other)

    def __mul__(self, other):
        if isinstance(other, (int, float)):
            return ScaleOfKernel(self, other)
        if isinstance(other, Kernel):
            return ProductOfKernel(self, other)
        logging.getLogger(__name__).error("only numbers and Kernels are supported operand types for *")

    __rmul__ = __mul__

    # -- device dispatch ---------------------------------------------------------------------------
    def _device_params(self):
        """(kind, para, flags) of the device functor."""
        return self._kind, 0, 0

    _WRONG_DER = "Wrong derivative index"

    def _device_eval(self, x, z, mode, der):
        if mode not in _MODES:
            raise Exception("Specify the mode: 'train' or 'cross'")
        kind, para, flags = self._device_params()
        xa = None if x is None else _lib.f64(x)
        za = None if z is None else _lib.f64(z)
        if mode == "self_test":
            xa = None
        if mode == "train":
            za = None
        ref = xa if xa is not None else za
        n = 0 if xa is None else xa.shape[0]
        m = 0 if za is None else za.shape[0]
        d = ref.shape[1]
        hyp = _lib.f64(np.asarray(self.hyp, dtype=float))
        shape = {"train": (n, n), "cross": (n, m), "self_test": (m, 1)}[mode]
        out = np.empty(shape)
        rc = _lib.load().pgp_cov(_lib.ctx(), kind, _MODES[mode], -1 if der is None else int(der), _lib.ptr(xa), n,
                                 _lib.ptr(za), m, d, _lib.ptr(hyp), len(hyp), int(para), int(flags), _lib.ptr(out))
        _lib.check(rc, "pgp_cov", {-4: self._WRONG_DER, -11: "number of hyperparameters does not match the input dimension"})
        return out


class RBF(Kernel):
    """Squared exponential, isotropic.  hyp = [log_ell, log_sigma]   (Core/cov.py:786-828)"""
    _kind = _lib.COV_RBF
    _WRONG_DER = "Calling for a derivative in RBF that does not exist"

    def __init__(self, log_ell=0., log_sigma=0.):
        self.hyp = [log_ell, log_sigma]
        self.para = []

    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        return self._device_eval(x, z, mode, None)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        return self._device_eval(x, z, mode, der)


class RBFard(Kernel):
    """Squared exponential with ARD.  hyp = log_ell_list + [log_sigma]   (Core/cov.py:872-938)"""
    _kind = _lib.COV_RBFARD
    _WRONG_DER = "Wrong derivative index in RDFard"

    def __init__(self, D=None, log_ell_list=None, log_sigma=0.):
        if log_ell_list is None:
            self.hyp = [0. for _ in range(D)] + [log_sigma]
        else:
            self.hyp = list(log_ell_list) + [log_sigma]
        self.para = []

    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        return self._device_eval(x, z, mode, None)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        return self._device_eval(x, z, mode, der)


class Matern(Kernel):
    """Matern, nu = d/2, d in {1,3,5,7}.  hyp = [log_ell, log_sigma], para = [d]  (Core/cov.py:1078-1182)

    ``getDerMatrix`` returns the mathematically correct derivative by default; set
    ``reference_compat = True`` to reproduce the reference's result bit-for-bit in structure
    (it applies dmfunc/mfunc to K instead of t, Core/cov.py:1173-1177 -- SURVEY Q4)."""
    _kind = _lib.COV_MATERN
    _WRONG_DER = "Wrong derivative value in Matern"

    def __init__(self, log_ell=0., d=3, log_sigma=0.):
        self.hyp = [log_ell, log_sigma]
        self.para = [d]
        self.logger = logging.getLogger(__name__)

    def _d(self):
        d = self.para[0]
        if np.abs(d - np.round(d)) < 1e-8:
            d = int(round(d))
        d = int(d)
        if d not in (1, 3, 5, 7):
            logging.getLogger(__name__).warning("d is neither 1,3,5 nor 7. We set it to d=3. ")
            d = 3
        return d

    def _device_params(self):
        return self._kind, self._d(), (_lib.FLAG_MATERN_REFERENCE_DER if self.reference_compat else 0)

    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        return self._device_eval(x, z, mode, None)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        return self._device_eval(x, z, mode, der)


class _DeviceKernel(Kernel):
    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        return self._device_eval(x, z, mode, None)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        return self._device_eval(x, z, mode, der)


class RBFunit(_DeviceKernel):
    """Squared exponential with unit magnitude.  hyp = [log_ell]   (Core/cov.py:832-869)"""
    _kind = _lib.COV_RBFUNIT
    _WRONG_DER = "Wrong derivative index in RDFunit"

    def __init__(self, log_ell=0.):
        self.hyp = [log_ell]
        self.para = []


class RQ(_DeviceKernel):
    """Rational quadratic, isotropic.  hyp = [log_ell, log_sigma, log_alpha]   (Core/cov.py:1304-1347)"""
    _kind = _lib.COV_RQ
    _WRONG_DER = "Wrong derivative index in covRQ"

    def __init__(self, log_ell=0., log_sigma=0., log_alpha=0.):
        self.hyp = [log_ell, log_sigma, log_alpha]
        self.para = []


class PiecePoly(_DeviceKernel):
    """Piecewise polynomial kernel with compact support.  hyp = [log_ell, log_sigma], para = [v], v in {0,1,2,3}
    (Core/cov.py:683-782)"""
    _kind = _lib.COV_PIECEPOLY
    _WRONG_DER = "Wrong derivative entry in PiecePoly"

    def __init__(self, log_ell=0., v=2, log_sigma=0.):
        self.hyp = [log_ell, log_sigma]
        self.para = [v]

    def _device_params(self):
        v = self.para[0]
        if np.abs(v - np.round(v)) < 1e-8:
            v = int(round(v))
        assert int(v) in range(4)                      # only degrees 0,1,2,3 (Core/cov.py:737)
        return self._kind, int(v), 0


# ---- composites: children are evaluated on the device, combined on the host (Core/cov.py:230-328) ----
class _Pair(Kernel):
    def __init__(self, cov1, cov2):
        self.cov1, self.cov2 = cov1, cov2
        self.para = []

    @property
    def hyp(self):
        return list(self.cov1.hyp) + list(self.cov2.hyp)

    @hyp.setter
    def hyp(self, value):
        n1 = len(self.cov1.hyp)
        assert len(value) == n1 + len(self.cov2.hyp)
        self.cov1.hyp = list(value[:n1])
        self.cov2.hyp = list(value[n1:])


class SumOfKernel(_Pair):
    def getCovMatrix(self, x=None, z=None, mode=None):
        return self.cov1.getCovMatrix(x, z, mode) + self.cov2.getCovMatrix(x, z, mode)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        n1 = len(self.cov1.hyp)
        if der < n1:
            return self.cov1.getDerMatrix(x, z, mode, der)
        if der < n1 + len(self.cov2.hyp):
            return self.cov2.getDerMatrix(x, z, mode, der - n1)
        raise Exception("Error: der out of range for covSum")


class ProductOfKernel(_Pair):
    def getCovMatrix(self, x=None, z=None, mode=None):
        return self.cov1.getCovMatrix(x, z, mode) * self.cov2.getCovMatrix(x, z, mode)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        n1 = len(self.cov1.hyp)
        if der < n1:
            return self.cov1.getDerMatrix(x, z, mode, der) * self.cov2.getCovMatrix(x, z, mode)
        if der < n1 + len(self.cov2.hyp):
            return self.cov2.getDerMatrix(x, z, mode, der - n1) * self.cov1.getCovMatrix(x, z, mode)
        raise Exception("Error: der out of range for covProduct")


class ScaleOfKernel(Kernel):
    def __init__(self, cov, scalar):
        self.cov = cov
        self.para = []
        self._scale = [np.log(scalar)] if scalar else [-np.inf]

    @property
    def hyp(self):
        return self._scale + list(self.cov.hyp)

    @hyp.setter
    def hyp(self, value):
        assert len(value) == 1 + len(self.cov.hyp)
        self._scale = [value[0]]
        self.cov.hyp = list(value[1:])

    def getCovMatrix(self, x=None, z=None, mode=None):
        return np.exp(self._scale[0]) * self.cov.getCovMatrix(x, z, mode)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        if der == 0:
            return np.exp(self._scale[0]) * self.cov.getCovMatrix(x, z, mode)
        return np.exp(self._scale[0]) * self.cov.getDerMatrix(x, z, mode, der - 1)
