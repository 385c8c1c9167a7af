"""Model facade (reference: pyGPs/Core/gp.py -- GP :62-527, GPR :533-635, GPC :641-732).

Same public surface for the pieces that call the hot path: ``setData``, ``setPrior``, ``setNoise``,
``setOptimizer``, ``optimize``, ``getPosterior``, ``predict``, ``predict_with_posterior`` and the
result attributes ``nlZ, dnlZ, posterior, ym, ys2, fm, fs2, lp``.  Plotting, FITC and multi-class
wrappers are out of scope.
"""
import logging
from copy import deepcopy

import numpy as np

from . import _lib, conf, cov, inf, lik, mean, opt


def _col(a):
    a = np.asarray(a)
    return a.reshape(a.shape[0], 1) if a.ndim == 1 else a


class GP(object):
    """Base class for GP models."""

    def __init__(self):
        super(GP, self).__init__()
        self.usingDefaultMean = True
        self.meanfunc = None
        self.covfunc = None
        self.likfunc = None
        self.inffunc = None
        self.optimizer = None
        self.nlZ = None
        self.dnlZ = None
        self.posterior = None
        self.x = None
        self.y = None
        self.xs = None
        self.ys = None
        self.ym = None
        self.ys2 = None
        self.fm = None
        self.fs2 = None
        self.lp = None
        self.logger = logging.getLogger(__name__)

    def __repr__(self):
        return str(type(self)) + ": model.nlZ, model.dnlZ, model.posterior, model.{mean,cov,lik}func.hyp, model.ym/ys2/fm/fs2/lp"

    # ---- data / prior -------------------------------------------------------------------------------
    def setData(self, x, y):
        """Set training inputs (n,D) and targets (n,1); 1-d arrays are reshaped.  While the default
        mean is in use it is replaced by Const(mean(y))  (Core/gp.py:133-156)."""
        assert x.shape[0] == y.shape[0], "number of inputs and labels does not match"
        self.x = _col(x)
        self.y = _col(y)
        if self.usingDefaultMean:
            self.meanfunc = mean.Const(np.mean(y))

    def setPrior(self, mean=None, kernel=None):
        from . import mean as _mean
        if mean is not None:
            assert isinstance(mean, _mean.Mean), "mean function is not an instance of pygps_amd.mean.Mean"
            self.meanfunc = mean
            self.usingDefaultMean = False
        if kernel is not None:
            assert isinstance(kernel, cov.Kernel), "cov function is not an instance of pygps_amd.cov.Kernel"
            self.covfunc = kernel

    def setOptimizer(self, method, num_restarts=None, min_threshold=None, meanRange=None, covRange=None, likRange=None):
        conf_ = None
        if (num_restarts is not None) or (min_threshold is not None):
            conf_ = conf.random_init_conf(self.meanfunc, self.covfunc, self.likfunc)
            conf_.num_restarts = num_restarts
            conf_.min_threshold = min_threshold
            if meanRange is not None:
                conf_.meanRange = meanRange
            if covRange is not None:
                conf_.covRange = covRange
            if likRange is not None:
                conf_.likRange = likRange
        if method == "Minimize":
            self.optimizer = opt.Minimize(self, conf_)
        elif method == "ShardedMinimize":
            self.optimizer = opt.ShardedMinimize(self, conf_)
        else:
            raise Exception("Optimization method is not set correctly in setOptimizer")

    # ---- training -----------------------------------------------------------------------------------
    def _take_xy(self, x, y):
        if x is not None and y is not None:
            assert x.shape[0] == y.shape[0], "number of inputs and labels does not match"
        if x is not None:
            self.x = _col(x)
        if y is not None:
            self.y = _col(y)
        if self.usingDefaultMean and self.meanfunc is None:
            self.meanfunc = mean.Const(np.mean(y))

    def optimize(self, x=None, y=None, numIterations=40):
        """Learn the hyper-parameters; then refresh the posterior (Core/gp.py:251-285)."""
        self._take_xy(x, y)
        optimalHyp, optimalNlZ = self.optimizer.findMin(self.x, self.y, numIters=numIterations)
        self.nlZ = optimalNlZ
        self.optimizer._apply_in_objects(optimalHyp)
        self.getPosterior()

    def getPosterior(self, x=None, y=None, der=True):
        """nlZ, dnlZ, post = getPosterior(x, y) ; nlZ, post = getPosterior(x, y, der=False)
        (Core/gp.py:289-345)."""
        self._take_xy(x, y)
        if isinstance(self.likfunc, lik.Erf):
            labels = np.unique(np.asarray(self.y))
            if np.any((labels != 1) & (labels != -1)):
                raise Exception("You attempt classification using labels different from {+1,-1}")
        if not der:
            post, nlZ = self.inffunc.evaluate(self.meanfunc, self.covfunc, self.likfunc, self.x, self.y, 2)
            self.nlZ = nlZ
            self.posterior = deepcopy(post)
            return nlZ, post
        post, nlZ, dnlZ = self.inffunc.evaluate(self.meanfunc, self.covfunc, self.likfunc, self.x, self.y, 3)
        self.nlZ = nlZ
        self.dnlZ = deepcopy(dnlZ)
        self.posterior = deepcopy(post)
        return nlZ, dnlZ, post

    # ---- prediction ---------------------------------------------------------------------------------
    def _latent(self, post, xs):
        """fmu, fs2 for the (alpha, sW, L) parametrisation, on the device (Core/gp.py:395-417)."""
        L = post.L
        fitc = getattr(post, "fitc", None)
        if fitc is not None:                               # dense-L parametrisation of FITC (Core/gp.py:404-417)
            xs = _lib.f64(xs)
            ns = xs.shape[0]
            ms = _lib.f64(self.meanfunc.getMean(xs)).reshape(ns)
            fmu = np.empty(ns)
            fs2 = np.empty(ns)
            _lib.check(_lib.load().pgp_fitc_predict(fitc.ctx, fitc.handle, _lib.ptr(xs), ns, _lib.ptr(ms), _lib.ptr(fmu),
                                                    _lib.ptr(fs2)), "pgp_fitc_predict")
            return fmu.reshape(ns, 1), fs2.reshape(ns, 1)
        if type(L).__name__ == "DistributedFactor":            # a sharded fit: collective predict on the distributed posterior
            xs = _lib.f64(xs)
            return L.predict(xs, self.meanfunc.getMean(xs))
        if not isinstance(L, inf.DeviceFactor):
            raise NotImplementedError("pygps_amd: predict needs a posterior produced by pygps_amd inference "
                                      "(device-resident factor); there is no CPU fallback")
        xs = _lib.f64(xs)
        ns = xs.shape[0]
        ms = _lib.f64(self.meanfunc.getMean(xs)).reshape(ns)
        fmu = np.empty(ns)
        fs2 = np.empty(ns)
        if getattr(L, "dense", False):
            # a covariance function that is not a device program (inf.Exact._evaluate_dense): the cross-covariance block is
            # built by getCovMatrix and handed in; solve and reductions on the device (pgp_predict_dense)
            Ks = _lib.f64(self.covfunc.getCovMatrix(x=self.x, z=xs, mode="cross"))
            kss = _lib.f64(self.covfunc.getCovMatrix(z=xs, mode="self_test")).reshape(ns)
            _lib.check(_lib.load().pgp_predict_dense(L.ctx, L.handle, _lib.ptr(Ks), ns, _lib.ptr(kss), _lib.ptr(ms), _lib.ptr(fmu),
                                                     _lib.ptr(fs2)), "pgp_predict_dense")
            return fmu.reshape(ns, 1), fs2.reshape(ns, 1)
        rc = _lib.load().pgp_predict(L.ctx, L.handle, _lib.ptr(xs), ns, _lib.ptr(ms), _lib.ptr(fmu), _lib.ptr(fs2))
        if rc == -99:
            raise NotImplementedError("pygps_amd: the device predict path is not built in this version")
        _lib.check(rc, "pgp_predict")
        return fmu.reshape(ns, 1), fs2.reshape(ns, 1)

    def _predict(self, post, xs, ys):
        xs = _col(xs)
        self.xs = xs
        if ys is not None:
            ys = _col(ys)
            self.ys = ys
        fmu, fs2 = self._latent(post, xs)
        lp, ymu, ys2 = self.likfunc.evaluate(ys, fmu, fs2, None, None, 3)
        self.ym, self.ys2, self.lp, self.fm, self.fs2 = ymu, ys2, lp, fmu, fs2
        return (ymu, ys2, fmu, fs2, None) if ys is None else (ymu, ys2, fmu, fs2, lp)

    def predict(self, xs, ys=None):
        """ym, ys2, fm, fs2, lp = predict(xs[, ys])   (Core/gp.py:349-437)"""
        if self.posterior is None:
            self.getPosterior()
        return self._predict(self.posterior, xs, ys)

    def predict_with_posterior(self, post, xs, ys=None):
        """Same with an explicitly given posterior (Core/gp.py:441-527)."""
        return self._predict(post, xs, ys)


class GPR(GP):
    """Gaussian-process regression: Zero mean, RBF, Gauss likelihood, Exact inference, Minimize."""

    def __init__(self):
        super(GPR, self).__init__()
        self.meanfunc = mean.Zero()
        self.covfunc = cov.RBF()
        self.likfunc = lik.Gauss()
        self.inffunc = inf.Exact()
        self.optimizer = opt.Minimize(self)

    def setNoise(self, log_sigma):
        self.likfunc = lik.Gauss(log_sigma)

    def useInference(self, newInf):
        if newInf == "EP":
            self.inffunc = inf.EP()
        else:
            raise Exception('Possible inf values are "EP" (Laplace is out of scope of pygps_amd).')


class GPC(GP):
    """Binary GP classification: Zero mean, RBF, Erf likelihood, EP inference, Minimize."""

    def __init__(self):
        super(GPC, self).__init__()
        self.meanfunc = mean.Zero()
        self.covfunc = cov.RBF()
        self.likfunc = lik.Erf()
        self.inffunc = inf.EP()
        self.optimizer = opt.Minimize(self)

    def useInference(self, newInf):
        if newInf == "EP":
            self.inffunc = inf.EP()
        else:
            raise Exception('Possible inf values are "EP" (Laplace is out of scope of pygps_amd).')


class GP_FITC(GP):
    """Base class of the FITC models (Core/gp.py:934-1008)."""

    def __init__(self):
        super(GP_FITC, self).__init__()
        self.u = None                                      # inducing points

    def setData(self, x, y, value_per_axis=5):
        """Training data; without user-given inducing points a regular grid with ``value_per_axis`` values per input
        dimension is used (Core/gp.py:944-983)."""
        import itertools
        assert x.shape[0] == y.shape[0], "number of inputs and labels does not match"
        self.x = _col(x)
        self.y = _col(y)
        if self.usingDefaultMean:
            self.meanfunc = mean.Const(np.mean(y))
        axes = [np.linspace(np.min(self.x[:, k]), np.max(self.x[:, k]), value_per_axis) for k in range(self.x.shape[1])]
        if self.u is None:
            self.u = np.array(list(itertools.product(*axes)))
            self.covfunc = self.covfunc.fitc(self.u)

    def setPrior(self, mean=None, kernel=None, inducing_points=None):
        if kernel is not None:
            if inducing_points is not None:
                self.covfunc = kernel.fitc(inducing_points)
                self.u = inducing_points
            elif self.u is not None:
                self.covfunc = kernel.fitc(self.u)
            else:
                raise Exception("To use default inducing points, please call setData() first!")
        if mean is not None:
            self.meanfunc = mean
            self.usingDefaultMean = False


class GPR_FITC(GP_FITC):
    """Sparse GP regression with the FITC approximation (Core/gp.py:1010-1100)."""

    def __init__(self):
        super(GPR_FITC, self).__init__()
        self.meanfunc = mean.Zero()
        self.covfunc = cov.RBF()
        self.likfunc = lik.Gauss()
        self.inffunc = inf.FITC_Exact()
        self.optimizer = opt.Minimize(self)
        self.u = None

    def setNoise(self, log_sigma):
        self.likfunc = lik.Gauss(log_sigma)

    def useInference(self, newInf):
        raise Exception('FITC_Laplace / FITC_EP are out of scope of pygps_amd.')
