"""Inference engines on the hot path (reference: pyGPs/Core/inf.py -- postStruct :59-89,
dnlZStruct :93-128, Inference :133-172, Exact :345-384, EP :723-806).

``Exact.evaluate`` keeps the reference's signature and result types but the whole body -- kernel
assembly, Cholesky, solves, nlZ and all hyper-gradients -- is ONE call into the device pipeline
(``pgp_exact_fit``).  x and y stay resident on the GPU between calls (the optimiser only changes
hyp, Core/opt.py:70-75); ``post.L`` is a lazy host view of the device-resident factor so that a fit
is not a PCIe benchmark.  No CPU fallback.
"""
import ctypes as C
import hashlib
import logging
import weakref

import numpy as np

from . import _lib, cov as _cov, lik as _lik

try:                                    # fast content hash for the residency check
    import xxhash

    def _digest(a):
        return xxhash.xxh3_64_intdigest(memoryview(a).cast("B"))
except Exception:                       # pragma: no cover
    def _digest(a):
        return hashlib.blake2b(memoryview(a).cast("B"), digest_size=8).digest()


class postStruct(object):
    """Posterior parametrisation N(m + K alpha, (K^-1 + W)^-1): alpha (n,1), sW (n,1), L (n,n) =
    chol(I + sW sW' o K) as the UPPER factor (Core/inf.py:59-89)."""

    def __init__(self):
        self.alpha = np.array([])
        self.L = np.array([])
        self.sW = np.array([])

    def __repr__(self):
        return ("posterior: to get the parameters of the posterior distribution use:\n"
                "model.posterior.alpha\nmodel.posterior.L\nmodel.posterior.sW\n"
                "See documentation and gpml book chapter 2.3 and chapter 3.4.3 for these parameters.")

    def __str__(self):
        return ("posterior distribution described by alpha, sW and L\n"
                "See documentation and gpml book chapter 2.3 and chapter 3.4.3 for these parameters\n"
                "alpha:\n" + str(self.alpha) + "\nL:\n" + str(np.asarray(self.L)) + "\nsW:\n" + str(self.sW))


class dnlZStruct(object):
    """Partial derivatives of nlZ w.r.t. mean / cov / lik hyper-parameters (Core/inf.py:93-128)."""

    def __init__(self, m, c, l):
        self.mean = [0 for _ in range(len(m.hyp))] if m.hyp is not None else []
        self.cov = [0 for _ in range(len(c.hyp))] if c.hyp is not None else []
        self.lik = [0 for _ in range(len(l.hyp))] if l.hyp is not None else []

    def __str__(self):
        return ("Derivatives of mean, cov and lik functions:\nmean:" + str(self.mean) + "\ncov:" + str(self.cov)
                + "\nlik:" + str(self.lik))

    def __repr__(self):
        return ("dnlZ: to get the derivatives of mean, cov and lik functions use:\n"
                "model.dnlZ.mean\nmodel.dnlZ.cov\nmodel.dnlZ.lik")

    def accumulateDnlZ(self, other):
        self.mean = [a + b for a, b in zip(self.mean, other.mean)]
        self.cov = [a + b for a, b in zip(self.cov, other.cov)]
        self.lik = [a + b for a, b in zip(self.lik, other.lik)]
        return self


class DeviceFactor(object):
    """``post.L``: the (n,n) upper Cholesky factor R (R'R = I + sW sW' o K), resident on the GPU.

    Behaves like a read-only numpy array -- shape/len/indexing/np.asarray/.T all work and trigger ONE
    device->host copy (2 GiB at n=16384) on first touch; exact zeros below the diagonal, as
    Core/gp.py:393 requires.  ``copy.deepcopy`` shares the handle (the buffer is never rewritten).
    ``predict`` uses the handle directly and never materialises it."""
    ndim = 2
    dtype = np.dtype(np.float64)
    #: True on the factors of the dense path (inf.Exact._evaluate_dense sets it on the instance).  A CLASS attribute so that
    #: `predict`'s test never reaches __getattr__, which forwards unknown names to the host copy (an n x n D2H transfer).
    dense = False
    #: bytes of factor buffers ((n + 128) n doubles each: factor + rhs rows) held by live DeviceFactor objects.
    #: Models sit in reference cycles (model <-> optimizer), so a dropped model frees its factor only when the cyclic
    #: garbage collector runs; `reserve` forces a collection before the device fills up with unreachable factors.
    live_bytes = 0
    # RLock: the finalizer below can run from the cyclic GC on the very thread that is inside `with _live_lock`
    _live_lock = __import__("threading").RLock()

    def __init__(self, handle, n, device, slot=0):
        self._h = handle
        self._n = int(n)
        self._dev = device
        self._slot = slot
        self._host = None
        nbytes = DeviceFactor.nbytes_for(n)
        with DeviceFactor._live_lock:
            DeviceFactor.live_bytes += nbytes
        # the context handle and the entry point are captured NOW: a finalizer may fire (cyclic GC) while this thread
        # holds _lib's module lock inside ctx()/load(), so it must not look either of them up again
        self._ctx = _lib.ctx(device, slot)
        self._fin = weakref.finalize(self, DeviceFactor._release, _lib.load().pgp_factor_free, self._ctx, handle, nbytes)

    @staticmethod
    def nbytes_for(n):
        np_ = (int(n) + 127) // 128 * 128
        return (np_ + 128) * np_ * 8

    @staticmethod
    def reserve(n, device):
        """Called before a fit: collect unreachable models when their factors hold more than a quarter of the HBM."""
        if DeviceFactor.live_bytes + DeviceFactor.nbytes_for(n) > 0.25 * _lib.device_memory_bytes(device):
            import gc
            gc.collect()

    @staticmethod
    def _release(free_fn, ctx_handle, handle, nbytes=0):
        with DeviceFactor._live_lock:
            DeviceFactor.live_bytes -= nbytes
        try:
            free_fn(ctx_handle, handle)     # pgp_factor_free: takes the context's own pool mutex, no Python lock
        except Exception:       # interpreter shutdown
            pass

    @property
    def ctx(self):
        """The context that owns the device buffers of this factor."""
        return self._ctx

    @property
    def handle(self):
        return self._h

    @property
    def shape(self):
        return (self._n, self._n)

    @property
    def size(self):
        return self._n * self._n

    def __len__(self):
        return self._n

    def host(self):
        if self._host is None:
            out = np.empty((self._n, self._n))
            _lib.check(_lib.load().pgp_factor_to_host(self.ctx, self._h, _lib.ptr(out)), "pgp_factor_to_host")
            out.setflags(write=False)
            self._host = out
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        return self.host()[idx]

    @property
    def T(self):
        return self.host().T

    def __deepcopy__(self, memo):
        return self

    def __copy__(self):
        return self

    def __getattr__(self, name):            # any other ndarray attribute/method
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.host(), name)

    def __repr__(self):
        return "DeviceFactor(n=%d, on device %d%s)" % (self._n, self._dev, ", materialised" if self._host is not None else "")


class Inference(object):
    """Base class (Core/inf.py:133-172)."""

    def __init__(self):
        self.logger = logging.getLogger(__name__)

    def evaluate(self, meanfunc, covfunc, likfunc, x, y, nargout=1):
        raise NotImplementedError


class _Resident(object):
    """Tracks which (x, y) the device context currently holds."""
    key = {}

    @classmethod
    def ensure(cls, x, y, device):
        k = (x.shape, _digest(x), _digest(y))
        where = (device, _lib.current_slot())
        if cls.key.get(where) != k:
            _lib.check(_lib.load().pgp_set_data(_lib.ctx(device), _lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(y)),
                       "pgp_set_data")
            cls.key[where] = k

    @classmethod
    def invalidate(cls, device=None):
        for k in [k for k in cls.key if device is None or k[0] == device]:
            cls.key.pop(k, None)


def _device_kernel(covfunc, ctx):
    """(kind, para, flags) of ``covfunc`` on context ``ctx`` (composites register their device program there)."""
    if not isinstance(covfunc, _cov.Kernel):
        raise NotImplementedError("pygps_amd: covfunc must be a pygps_amd.cov.Kernel (got %s); there is no CPU fallback"
                                  % type(covfunc).__name__)
    return covfunc._bind(ctx)


def _mean_inputs(meanfunc, x):
    n = x.shape[0]
    m = _lib.f64(meanfunc.getMean(x)).reshape(n)
    nm = len(meanfunc.hyp) if meanfunc.hyp else 0
    dm = None
    if nm:
        dm = np.empty((nm, n))
        for i in range(nm):
            dm[i] = np.asarray(meanfunc.getDerMatrix(x, i), dtype=float).reshape(n)
    return m, dm, nm


class Exact(Inference):
    """Exact inference for a Gaussian likelihood (Core/inf.py:345-384)."""

    def __init__(self, sharded=False, gather_factor=False):
        """sharded: False = one GPU (the reference's unit of work); True or a ``sharded.Comm`` = ONE fit over every rank of
        the process group (one process per GPU, all ranks call with the same model; pygps_amd/sharded.py).  gather_factor:
        a sharded fit also returns post.L as a host array on every rank (n^2 doubles; moderate n)."""
        self.name = "Exact inference"
        self.device = None
        self.sharded = sharded
        self.gather_factor = gather_factor

    def _evaluate_sharded(self, meanfunc, covfunc, likfunc, x, y, nargout):
        from . import sharded as _sh
        comm = self.sharded if isinstance(self.sharded, _sh.Comm) else _sh.default_comm(self.device)
        dev = comm.device
        kind, para, flags = _device_kernel(covfunc, comm.ctx)
        x = _lib.f64(x)
        n = x.shape[0]
        y = _lib.f64(y).reshape(n)
        _Resident.ensure(x, y, dev)
        m, dm, nm = _mean_inputs(meanfunc, x)
        nc = len(covfunc.hyp)
        log_sn = float(likfunc.hyp[0])
        alpha, nlz, g, ms6, Lh, fh = _sh.exact_fit(comm, kind, para, flags, covfunc.hyp, log_sn, m, dm, nm, n, nargout,
                                                   gather_factor=self.gather_factor, keep_factor=not self.gather_factor)
        self.last_ms = ms6[:4]
        self.last_bytes = {"peak_device_bytes": int(ms6[4]), "factor_device_bytes": int(ms6[5])}
        # the multi-rank timers of this rank: stall of the compute stream waiting for panels, time inside the broadcasts
        self.last_comm = {"wait_panel_ms": float(ms6[6]), "bcast_ms": float(ms6[7]), "bcast_bytes": float(ms6[8]),
                          "bcast_max_ms": float(ms6[9]),
                          "bcast_GBs": float(ms6[8]) / max(float(ms6[7]), 1e-9) / 1e6 if ms6[7] > 0 else 0.0,
                          "wait_share": float(ms6[6]) / max(float(ms6[1]), 1e-9) if ms6[1] > 0 else 0.0}
        post = postStruct()
        post.alpha = alpha.reshape(n, 1)
        post.sW = np.ones((n, 1)) / np.sqrt(np.exp(2 * log_sn))
        post.L = Lh if Lh is not None else _sh.DistributedFactor(n, comm, fh)
        if nargout > 1:
            if nargout > 2:
                dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
                dnlZ.mean = [np.float64(v) for v in g[:nm]]
                dnlZ.cov = [np.float64(v) for v in g[nm:nm + nc]]
                dnlZ.lik = [np.float64(g[nm + nc])]
                return post, np.float64(nlz), dnlZ
            return post, np.float64(nlz)
        return post

    def _evaluate_dense(self, meanfunc, covfunc, likfunc, x, y, nargout):
        """Covariance functions that are not device programs (a tree with more than two ARD leaves or more than 8 leaves /
        products, Core/cov.py:230-328): K and the derivative matrices come from getCovMatrix / getDerMatrix (the children's
        device-built matrices combined on the host), the factorisation with the fused inverse, alpha, nlZ and every Hadamard
        sum sum(Q o dK_h) / 2 run on the device (csrc/dense.hip).  Costs PCIe traffic (n^2 doubles per matrix), never raises."""
        dev = _lib.default_device() if self.device is None else self.device
        ctx = _lib.ctx(dev)
        lib = _lib.load()
        x = _lib.f64(x)
        n = x.shape[0]
        y = _lib.f64(y).reshape(n)
        m, dm, nm = _mean_inputs(meanfunc, x)
        nc = len(covfunc.hyp)
        log_sn = float(likfunc.hyp[0])
        K = _lib.f64(covfunc.getCovMatrix(x=x, mode="train"))
        r = _lib.f64(y - m)
        alpha = np.empty(n)
        nlZ = np.zeros(1)
        glik = np.zeros(1)
        fh = C.c_void_p()
        DeviceFactor.reserve(n, dev)
        want = int(min(max(nargout, 1), 3))
        _lib.check(lib.pgp_exact_fit_dense(ctx, _lib.ptr(K), n, _lib.ptr(r), log_sn, want, _lib.ptr(alpha), _lib.ptr(nlZ),
                                           _lib.ptr(glik), C.byref(fh)), "pgp_exact_fit_dense")
        del K
        post = postStruct()
        post.alpha = alpha.reshape(n, 1)
        post.sW = np.ones((n, 1)) / np.sqrt(np.exp(2 * log_sn))
        post.L = DeviceFactor(fh, n, dev, _lib.current_slot())
        post.L.dense = True                          # predict hands the cross-covariance block in (GP._latent)
        if nargout > 1:
            if nargout > 2:
                g = np.zeros(1)
                dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
                dnlZ.cov = []
                for h in range(nc):                                               # Core/inf.py:376-377, one matrix at a time
                    dK = _lib.f64(covfunc.getDerMatrix(x=x, mode="train", der=h))
                    _lib.check(lib.pgp_dense_grad_term(ctx, _lib.ptr(dK), n, log_sn, _lib.ptr(g)), "pgp_dense_grad_term")
                    dnlZ.cov.append(np.float64(g[0]))
                dnlZ.mean = [np.float64(-(dm[i] @ alpha)) for i in range(nm)]     # Core/inf.py:378-381
                dnlZ.lik = [np.float64(glik[0])]
                return post, np.float64(nlZ[0]), dnlZ
            return post, np.float64(nlZ[0])
        return post

    def evaluate(self, meanfunc, covfunc, likfunc, x, y, nargout=1):
        if not isinstance(likfunc, _lik.Gauss):
            raise Exception("Exact inference only possible with Gaussian likelihood")
        if isinstance(covfunc, _cov._Composite) and not covfunc._on_device():
            if self.sharded:
                raise NotImplementedError("pygps_amd: a sharded fit needs a covariance function that runs as a device program")
            return self._evaluate_dense(meanfunc, covfunc, likfunc, x, y, nargout)
        if self.sharded:
            return self._evaluate_sharded(meanfunc, covfunc, likfunc, x, y, nargout)
        dev = _lib.default_device() if self.device is None else self.device
        kind, para, flags = _device_kernel(covfunc, _lib.ctx(dev))
        x = _lib.f64(x)
        n, D = x.shape
        y = _lib.f64(y).reshape(n)
        DeviceFactor.reserve(n, dev)
        _Resident.ensure(x, y, dev)
        m, dm, nm = _mean_inputs(meanfunc, x)
        hyp = _lib.f64(np.asarray(covfunc.hyp, dtype=float))
        nc = len(hyp)
        log_sn = float(likfunc.hyp[0])
        alpha = np.empty(n)
        nlZ = np.zeros(1)
        g = np.zeros(nm + nc + 1)
        fh = C.c_void_p()
        rc = _lib.load().pgp_exact_fit(_lib.ctx(dev), kind, _lib.ptr(hyp), nc, int(para), int(flags), log_sn,
                                       _lib.ptr(m), _lib.ptr(dm), nm, int(min(max(nargout, 1), 3)), _lib.ptr(alpha),
                                       _lib.ptr(nlZ), _lib.ptr(g), C.byref(fh))
        _lib.check(rc, "pgp_exact_fit")
        sn2 = np.exp(2 * log_sn)
        post = postStruct()
        post.alpha = alpha.reshape(n, 1)
        post.sW = np.ones((n, 1)) / np.sqrt(sn2)
        post.L = DeviceFactor(fh, n, dev, _lib.current_slot())
        if nargout > 1:
            nlz = np.float64(nlZ[0])
            if nargout > 2:
                dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
                dnlZ.mean = [np.float64(v) for v in g[:nm]]
                dnlZ.cov = [np.float64(v) for v in g[nm:nm + nc]]
                dnlZ.lik = [np.float64(g[nm + nc])]
                return post, nlz, dnlZ
            return post, nlz
        return post


class EP(Inference):
    """Expectation propagation with the probit likelihood (Core/inf.py:723-806).  The site-parameter
    state (last_ttau / last_tnu) persists on the object across calls and warm-starts the next one,
    exactly like the reference."""

    def __init__(self):
        self.name = "Expectation Propagation"
        self.last_ttau = None
        self.last_tnu = None
        self.device = None
        self.sweeps = 0

    def evaluate(self, meanfunc, covfunc, likfunc, x, y, nargout=1):
        if not isinstance(likfunc, _lik.Erf):
            raise NotImplementedError("pygps_amd: EP runs on the device for lik.Erf only (no CPU fallback)")
        dev = _lib.default_device() if self.device is None else self.device
        dense = isinstance(covfunc, _cov._Composite) and not covfunc._on_device()
        if not dense:
            kind, para, flags = _device_kernel(covfunc, _lib.ctx(dev))
        x = _lib.f64(x)
        n, D = x.shape
        y = _lib.f64(y).reshape(n)
        DeviceFactor.reserve(n, dev)
        _Resident.ensure(x, y, dev)
        m, dm, nm = _mean_inputs(meanfunc, x)
        hyp = _lib.f64(np.asarray(covfunc.hyp, dtype=float))
        nc = len(hyp)
        if dense:
            return self._evaluate_dense(meanfunc, covfunc, likfunc, x, n, m, dm, nm, nc, dev, nargout)
        warm = self.last_ttau is not None
        ttau = _lib.f64(self.last_ttau).reshape(n).copy() if warm else np.zeros(n)
        tnu = _lib.f64(self.last_tnu).reshape(n).copy() if warm else np.zeros(n)
        alpha = np.empty(n)
        sW = np.empty(n)
        nlZ = np.zeros(1)
        g = np.zeros(nm + nc + 1)
        sweeps = C.c_int()
        fh = C.c_void_p()
        rc = _lib.load().pgp_ep_fit(_lib.ctx(dev), kind, _lib.ptr(hyp), nc, int(para), int(flags), _lib.ptr(m),
                                    _lib.ptr(dm), nm, int(min(max(nargout, 1), 3)), int(warm), _lib.ptr(ttau),
                                    _lib.ptr(tnu), _lib.ptr(alpha), _lib.ptr(sW), _lib.ptr(nlZ), _lib.ptr(g),
                                    C.byref(sweeps), C.byref(fh))
        if rc == -99:
            raise NotImplementedError("pygps_amd: the device EP path is not built in this version")
        _lib.check(rc, "pgp_ep_fit")
        self.sweeps = sweeps.value
        if self.sweeps == 10:
            logging.getLogger(__name__).warning("maximum number of sweeps reached in function infEP")
        self.last_ttau = ttau.reshape(n, 1)
        self.last_tnu = tnu.reshape(n, 1)
        post = postStruct()
        post.alpha = alpha.reshape(n, 1)
        post.sW = sW.reshape(n, 1)
        post.L = DeviceFactor(fh, n, dev, _lib.current_slot())
        if nargout > 2:
            dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
            dnlZ.mean = [np.float64(v) for v in g[:nm]]
            dnlZ.cov = [np.float64(v) for v in g[nm:nm + nc]]
            dnlZ.lik = []
            return post, np.float64(nlZ[0]), dnlZ
        return post, np.float64(nlZ[0])


    def _evaluate_dense(self, meanfunc, covfunc, likfunc, x, n, m, dm, nm, nc, dev, nargout):
        """Covariance functions that are not device programs (Core/cov.py:230-328 composes anything): K and the derivative
        matrices come from getCovMatrix / getDerMatrix, the site sweeps, the posterior, alpha, nlZ and every Hadamard sum
        1/2 sum((sW sW' o B^-1 - alpha alpha') o dK_h) (Core/inf.py:780-786) run on the device."""
        lib = _lib.load()
        ctx = _lib.ctx(dev)
        K = _lib.f64(covfunc.getCovMatrix(x=x, mode="train"))
        warm = self.last_ttau is not None
        ttau = _lib.f64(self.last_ttau).reshape(n).copy() if warm else np.zeros(n)
        tnu = _lib.f64(self.last_tnu).reshape(n).copy() if warm else np.zeros(n)
        alpha = np.empty(n)
        sW = np.empty(n)
        nlZ = np.zeros(1)
        g = np.zeros(nm + 1)
        sweeps = C.c_int()
        fh = C.c_void_p()
        want = int(min(max(nargout, 1), 3))
        _lib.check(lib.pgp_ep_fit_dense(ctx, _lib.ptr(K), _lib.ptr(m), _lib.ptr(dm), nm, want, int(warm), _lib.ptr(ttau),
                                        _lib.ptr(tnu), _lib.ptr(alpha), _lib.ptr(sW), _lib.ptr(nlZ), _lib.ptr(g),
                                        C.byref(sweeps), C.byref(fh)), "pgp_ep_fit_dense")
        del K
        self.sweeps = sweeps.value
        if self.sweeps == 10:
            logging.getLogger(__name__).warning("maximum number of sweeps reached in function infEP")
        self.last_ttau = ttau.reshape(n, 1)
        self.last_tnu = tnu.reshape(n, 1)
        post = postStruct()
        post.alpha = alpha.reshape(n, 1)
        post.sW = sW.reshape(n, 1)
        post.L = DeviceFactor(fh, n, dev, _lib.current_slot())
        post.L.dense = True                          # predict hands the cross-covariance block in (GP._latent)
        if nargout > 2:
            dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
            gh = np.zeros(1)
            dnlZ.cov = []
            for h in range(nc):                                                   # one derivative matrix at a time
                dK = _lib.f64(covfunc.getDerMatrix(x=x, mode="train", der=h))
                _lib.check(lib.pgp_dense_grad_term(ctx, _lib.ptr(dK), n, 0.0, _lib.ptr(gh)), "pgp_dense_grad_term")
                dnlZ.cov.append(np.float64(gh[0]))
            dnlZ.mean = [np.float64(v) for v in g[:nm]]
            dnlZ.lik = []
            return post, np.float64(nlZ[0]), dnlZ
        return post, np.float64(nlZ[0])


class FITCPosterior(object):
    """Device handle of a FITC posterior (alpha, dense L, inducing coordinates) for predict; freed with the object."""

    def __init__(self, handle, device, slot):
        self.handle = handle
        self._dev = device
        self._slot = slot
        self._ctx = _lib.ctx(device, slot)                # captured now: the finalizer must not take _lib's lock
        self._fin = weakref.finalize(self, FITCPosterior._free, _lib.load().pgp_fitc_free, self._ctx, handle.value)

    @property
    def ctx(self):
        return self._ctx

    @staticmethod
    def _free(free_fn, ctx_handle, h):
        try:
            free_fn(ctx_handle, C.c_void_p(h))
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return self                                      # the device object is immutable; share it


class FITC_Exact(Inference):
    """FITC approximation to the posterior GP (Core/inf.py:386-455): exact inference with
    Kt = Q + diag(K - Q), Q = Ku' inv(Kuu + snu2 I) Ku, snu2 = sn2 / 1e6.  One device call (csrc/fitc.hip)."""

    def __init__(self):
        self.name = 'FICT exact inference'
        self.device = None

    def evaluate(self, meanfunc, covfunc, likfunc, x, y, nargout=1):
        if not isinstance(likfunc, _lik.Gauss):
            raise Exception('Exact inference only possible with Gaussian likelihood')
        if not isinstance(covfunc, _cov.FITCOfKernel):
            raise Exception('Only covFITC supported.')
        dev = _lib.default_device() if self.device is None else self.device
        kind, para, flags = _device_kernel(covfunc.covfunc, _lib.ctx(dev))
        x = _lib.f64(x)
        n, D = x.shape
        xu = _lib.f64(covfunc.inducingInput)
        if xu.shape[1] != D:
            raise Exception('Dimensionality of inducing inputs must match training inputs')
        nu = xu.shape[0]
        y = _lib.f64(y).reshape(n)
        _Resident.ensure(x, y, dev)
        m, dm, nm = _mean_inputs(meanfunc, x)
        hyp = _lib.f64(np.asarray(covfunc.hyp, dtype=float))
        nc = len(hyp)
        log_sn = float(likfunc.hyp[0])
        alpha = np.empty(nu)
        Lm = np.empty((nu, nu))
        nlZ = np.zeros(1)
        g = np.zeros(nm + nc + 1)
        fh = C.c_void_p()
        rc = _lib.load().pgp_fitc_fit(_lib.ctx(dev), kind, _lib.ptr(hyp), nc, int(para), int(flags), log_sn, _lib.ptr(xu), nu,
                                      _lib.ptr(m), _lib.ptr(dm), nm, int(min(max(nargout, 1), 3)), _lib.ptr(alpha),
                                      _lib.ptr(Lm), _lib.ptr(nlZ), _lib.ptr(g), C.byref(fh))
        _lib.check(rc, "pgp_fitc_fit")
        post = postStruct()
        post.alpha = alpha.reshape(nu, 1)
        post.L = Lm                                       # Sigma - inv(Kuu): dense, not triangular (inf.py:424)
        post.sW = np.ones((n, 1)) / np.sqrt(np.exp(2 * log_sn))
        post.fitc = FITCPosterior(fh, dev, _lib.current_slot())
        if nargout > 1:
            nlz = np.float64(nlZ[0])
            if nargout > 2:
                dnlZ = dnlZStruct(meanfunc, covfunc, likfunc)
                dnlZ.mean = [np.float64(v) for v in g[:nm]]
                dnlZ.cov = [np.float64(v) for v in g[nm:nm + nc]]
                dnlZ.lik = [np.float64(g[nm + nc])]
                return post, nlz, dnlZ
            return post, nlz
        return post
