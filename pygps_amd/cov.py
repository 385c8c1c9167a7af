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

    def fitc(self, inducingInput):
        """Covariance function for the FITC approximation (Core/cov.py:195-202)."""
        return FITCOfKernel(self, inducingInput)

    # -- device dispatch ---------------------------------------------------------------------------
    def _device_params(self):
        """(kind, para, flags) of the device functor."""
        return self._kind, 0, 0

    def _program(self, h0):
        """Postfix device program of this (sub)tree whose first hyper has flat index ``h0``:
        ``(tokens, n_leaves, n_products, n_scales)`` -- or None when a leaf cannot be part of a program."""
        if self._kind is None:
            return None
        kind, para, flags = self._device_params()
        ard = 1 if self._kind in (_lib.COV_RBFARD, _lib.COV_RQARD) else 0
        if ard and len(self.hyp) - (1 if self._kind == _lib.COV_RBFARD else 2) > _lib.PROG_MAX_ARD_DIM:
            return None
        return [_lib.PROG_LEAF, int(kind), int(para), int(flags), int(h0)], 1, 1, ard * 1000

    def _bind(self, ctx):
        """Select this kernel on context ``ctx``: returns (kind, para, flags) for the C entry points."""
        if self._kind is None:
            raise NotImplementedError(
                "pygps_amd: %s has no device covariance functor; there is no CPU fallback" % type(self).__name__)
        return self._device_params()

    _WRONG_DER = "Wrong derivative index"
    _BAD_PARA = "invalid kernel parameter"

    def _device_eval(self, x, z, mode, der):
        if mode not in _MODES:
            raise Exception("Specify the mode: 'train' or 'cross'")
        kind, para, flags = self._bind(_lib.ctx())
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
        _lib.check(rc, "pgp_cov", {-4: self._WRONG_DER, -11: "number of hyperparameters does not match the input dimension",
                                   -12: self._BAD_PARA})
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


class RQard(_DeviceKernel):
    """Rational quadratic with ARD.  hyp = log_ell_list + [log_sigma, log_alpha]   (Core/cov.py:1356-1425)

    The length-scale derivatives are the mathematically correct ones by default.  ``reference_compat = True``
    reproduces what the reference returns: all-zero matrices on 'train' (a missing transpose makes cdist see one
    1 x n point, :1415-1416) and coordinates multiplied instead of divided by ell_k on 'cross' (:1418)."""
    _kind = _lib.COV_RQARD
    _WRONG_DER = "Wrong derivative index in covRQard"

    def _device_params(self):
        return self._kind, 0, (_lib.FLAG_MATERN_REFERENCE_DER if self.reference_compat else 0)

    def __init__(self, D=None, log_ell_list=None, log_sigma=0., log_alpha=0.):
        if log_ell_list is None:
            self.hyp = [0. for _ in range(D)] + [log_sigma, log_alpha]
        else:
            self.hyp = list(log_ell_list) + [log_sigma, log_alpha]
        self.para = []


class Gabor(_DeviceKernel):
    """Gabor kernel h(t) = exp(-t^2/(2 ell^2)) cos(2 pi t / p).  hyp = [log_ell, log_p]   (Core/cov.py:392-450).
    Like the reference, the period is p = exp(2 * log_p) (:416) and the "derivatives" are dp*K and tan(dp)*dp*K
    (:441-445)."""
    _kind = _lib.COV_GABOR
    _WRONG_DER = "Wrong derivative entry in Gabor"

    def __init__(self, log_ell=0., log_p=0.):
        self.hyp = [log_ell, log_p]
        self.para = []


class Periodic(_DeviceKernel):
    """Smooth periodic kernel for 1-d inputs.  hyp = [log_ell, log_p, log_sigma]   (Core/cov.py:1186-1250)"""
    _kind = _lib.COV_PERIODIC
    _WRONG_DER = "Wrong derivative index in covPeriodic"
    _BAD_PARA = "periodic covariance can only be used for 1d data"

    def __init__(self, log_ell=0., log_p=0., log_sigma=0.):
        self.hyp = [log_ell, log_p, log_sigma]
        self.para = []

    def _device_eval(self, x, z, mode, der):
        for a in (x, z):                                   # Core/cov.py:1201-1204
            if a is not None:
                assert np.shape(a)[1] == 1, 'periodic covariance can only be used for 1d data'
        return super(Periodic, self)._device_eval(x, z, mode, der)


class Noise(_DeviceKernel):
    """White noise.  hyp = [log_sigma]   (Core/cov.py:1254-1300): s2*I on 'train', s2 where |x-z|^2 < 1e-9 on
    'cross', zeros on 'self_test'."""
    _kind = _lib.COV_NOISE
    _WRONG_DER = "Wrong derivative index in covNoise"

    def __init__(self, log_sigma=0.):
        self.hyp = [log_sigma]
        self.para = []


class Const(_DeviceKernel):
    """Constant kernel.  hyp = [log_sigma]   (Core/cov.py:941-982).  As in the reference the variance is
    exp(log_sigma) (:951, not squared), the training matrix carries a 1e-10 diagonal jitter (:957) and the
    derivative is 2*sf2 (:979)."""
    _kind = _lib.COV_CONST
    _WRONG_DER = "Wrong derivative entry in covConst"

    def __init__(self, log_sigma=0.):
        self.hyp = [log_sigma]
        self.para = []


# ---- composites (Core/cov.py:230-328) ---------------------------------------------------------------------
# A tree whose leaves all have isotropic device functors is evaluated in ONE pass of the tile kernel as a device
# program (sum of products of leaf functors, csrc/sqdist_tile.h CovProgram) -- in getCovMatrix/getDerMatrix and,
# more importantly, inside Exact/EP fits and predict.  Up to TWO leaves may be ARD kernels (RBFard / RQard, D <= 64): each
# gets its own weighted distance, accumulated beside the shared one.  Trees with three ARD leaves (or more than 8 leaves /
# products) still offer getCovMatrix/getDerMatrix by combining the children's device-built matrices.
class _Composite(Kernel):
    _kind = _lib.COV_COMPOSITE

    def _tokens(self):
        pr = self._program(0)
        if pr is None or pr[1] > _lib.PROG_MAX or pr[2] > _lib.PROG_MAX or pr[3] % 1000 > _lib.PROG_MAX or pr[3] >= 3000:
            return None                                   # too many leaves / products / Scale nodes, or more than two ARD leaves
        return pr[0]

    def _bind(self, ctx):
        tok = self._tokens()
        if tok is None:
            raise NotImplementedError(
                "pygps_amd: this composite kernel cannot run as a device program (more than two ARD leaves, an unsupported "
                "leaf, or more than %d leaves/products); there is no CPU fallback" % _lib.PROG_MAX)
        arr = (_lib.C.c_int32 * len(tok))(*tok)
        _lib.check(_lib.load().pgp_set_composite(ctx, arr, len(tok)), "pgp_set_composite")
        return _lib.COV_COMPOSITE, 0, 0

    def _on_device(self):
        return self._tokens() is not None

    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        if self._on_device():
            return self._device_eval(x, z, mode, None)
        return self._host_cov(x, z, mode)

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        if der >= len(self.hyp):
            raise Exception(self._WRONG_DER)
        if self._on_device():
            return self._device_eval(x, z, mode, der)
        return self._host_der(x, z, mode, der)


class _Pair(_Composite):
    _op = None

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

    def _program(self, h0):
        a = self.cov1._program(h0)
        b = self.cov2._program(h0 + len(self.cov1.hyp))
        if a is None or b is None:
            return None
        nprod = a[2] + b[2] if self._op == _lib.PROG_SUM else a[2] * b[2]
        return a[0] + b[0] + [self._op], a[1] + b[1], nprod, a[3] + b[3]     # [3]: Scale nodes + 1000 per ARD leaf


class SumOfKernel(_Pair):
    """Sum of two kernels (Core/cov.py:265-296)."""
    _op = _lib.PROG_SUM
    _WRONG_DER = "Error: der out of range for covSum"

    def _host_cov(self, x, z, mode):
        return self.cov1.getCovMatrix(x, z, mode) + self.cov2.getCovMatrix(x, z, mode)

    def _host_der(self, x, z, mode, der):
        n1 = len(self.cov1.hyp)
        if der < n1:
            return self.cov1.getDerMatrix(x, z, mode, der)
        return self.cov2.getDerMatrix(x, z, mode, der - n1)


class ProductOfKernel(_Pair):
    """Product of two kernels (Core/cov.py:230-261)."""
    _op = _lib.PROG_PRODUCT
    _WRONG_DER = "Error: der out of range for covProduct"

    def _host_cov(self, x, z, mode):
        return self.cov1.getCovMatrix(x, z, mode) * self.cov2.getCovMatrix(x, z, mode)

    def _host_der(self, x, z, mode, der):
        n1 = len(self.cov1.hyp)
        if der < n1:
            return self.cov1.getDerMatrix(x, z, mode, der) * self.cov2.getCovMatrix(x, z, mode)
        return self.cov2.getDerMatrix(x, z, mode, der - n1) * self.cov1.getCovMatrix(x, z, mode)


class ScaleOfKernel(_Composite):
    """Scaled kernel exp(h) * k (Core/cov.py:299-328).  As in the reference, the number given to ``k * number`` IS the
    log-space hyper h (:303), and the derivative w.r.t. h is returned as 2 * exp(h) * k (:324)."""
    _WRONG_DER = "Error: der out of range for covScale"

    def __init__(self, cov, scalar):
        self.cov = cov
        self.para = []
        self._scale = [scalar]

    @property
    def hyp(self):
        return self._scale + list(self.cov.hyp)

    @hyp.setter
    def hyp(self, value):
        assert len(value) == 1 + len(self.cov.hyp)
        self._scale = [value[0]]
        self.cov.hyp = list(value[1:])

    def _program(self, h0):
        a = self.cov._program(h0 + 1)
        if a is None:
            return None
        return a[0] + [_lib.PROG_SCALE, int(h0)], a[1], a[2], a[3] + 1

    def _host_cov(self, x, z, mode):
        return np.exp(self._scale[0]) * self.cov.getCovMatrix(x, z, mode)

    def _host_der(self, x, z, mode, der):
        if der == 0:
            return 2. * np.exp(self._scale[0]) * self.cov.getCovMatrix(x, z, mode)
        return np.exp(self._scale[0]) * self.cov.getDerMatrix(x, z, mode, der - 1)


class FITCOfKernel(Kernel):
    """Covariance "function" of the FITC approximation (Core/cov.py:332-390): instead of a full matrix it returns the
    (cross-)covariances with the inducing inputs xu -- 'train': (diag K, Kuu, Ku), 'cross': k(xu, z), 'self_test':
    k(z, z).  inf.FITC_Exact does not call these (the whole fit is one device call); they exist for API parity."""

    def __init__(self, cov, inducingInput):
        self.inducingInput = np.asarray(inducingInput, dtype=float)
        self.covfunc = cov
        self.para = []

    @property
    def hyp(self):
        return self.covfunc.hyp

    @hyp.setter
    def hyp(self, value):
        self.covfunc.hyp = value

    def _bind(self, ctx):
        return self.covfunc._bind(ctx)

    def _check_dim(self, x):
        if x is not None and self.inducingInput.shape[1] != np.shape(x)[1]:
            raise Exception('Dimensionality of inducing inputs must match training inputs')

    def getCovMatrix(self, x=None, z=None, mode=None):
        self.checkInputGetCovMatrix(x, z, mode)
        xu = self.inducingInput
        self._check_dim(x)
        if mode == 'self_test':
            return self.covfunc.getCovMatrix(z=z, mode='self_test')
        if mode == 'train':
            return (self.covfunc.getCovMatrix(z=x, mode='self_test'), self.covfunc.getCovMatrix(x=xu, mode='train'),
                    self.covfunc.getCovMatrix(x=xu, z=x, mode='cross'))
        return self.covfunc.getCovMatrix(x=xu, z=z, mode='cross')

    def getDerMatrix(self, x=None, z=None, mode=None, der=None):
        self.checkInputGetDerMatrix(x, z, mode, der)
        xu = self.inducingInput
        self._check_dim(x)
        if mode == 'self_test':
            return self.covfunc.getDerMatrix(z=z, mode='self_test', der=der)
        if mode == 'train':
            return (self.covfunc.getDerMatrix(z=x, mode='self_test', der=der),
                    self.covfunc.getDerMatrix(x=xu, mode='train', der=der),
                    self.covfunc.getDerMatrix(x=xu, z=x, mode='cross', der=der))
        return self.covfunc.getDerMatrix(x=xu, z=z, mode='cross', der=der)
