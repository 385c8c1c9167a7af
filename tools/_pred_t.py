import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
n, d, ns = 8192, 16, 65536
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
lib = _lib.load()
post = m.posterior
L = post.L
for i in range(6):
    t0 = time.perf_counter()
    xs_ = _lib.f64(xs); ms = _lib.f64(m.meanfunc.getMean(xs_)).reshape(ns); fmu = np.empty(ns); fs2 = np.empty(ns)
    t1 = time.perf_counter()
    rc = lib.pgp_predict(L.ctx, L.handle, _lib.ptr(xs_), ns, _lib.ptr(ms), _lib.ptr(fmu), _lib.ptr(fs2))
    t2 = time.perf_counter()
    lp, ymu, ys2 = m.likfunc.evaluate(None, fmu.reshape(ns, 1), fs2.reshape(ns, 1), None, None, 3)
    t3 = time.perf_counter()
    print("call %d: prep %.2f ms, pgp_predict %.2f ms, likelihood %.2f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), file=sys.stderr, flush=True)
