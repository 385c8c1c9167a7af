"""first predict after a fit (N = 8192): ms by number of test points, product form (W from the fit's inverse rows / from a trtri) and blocked solve"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
n, d = 8192, 16
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
xs = np.random.RandomState(1).randn(8192, d)
lib = _lib.load(); ctx = _lib.ctx()
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.setData(x, y)
for keep, mode, tag in ((1, 2, "product, W = E'"), (0, 2, "product, W by trtri"), (1, 0, "blocked solve")):
    lib.pgp_set_option(ctx, b"keep_inverse", keep); lib.pgp_set_option(ctx, b"predict_inverse", mode)
    for pts in (100, 1000, 8192):
        ts = []
        for rep in range(4):
            m.covfunc.hyp = [np.log(np.sqrt(d)) + 1e-3 * rep, 0.0]
            m.getPosterior()
            t = time.perf_counter(); m.predict(xs[:pts]); ts.append((time.perf_counter() - t) * 1e3)
        print("%-22s first predict of %5d points after a fit: %s ms" % (tag, pts, " ".join("%.2f" % v for v in ts)), flush=True)
