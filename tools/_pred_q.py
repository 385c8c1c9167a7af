import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
n, d, ns = 8192, 16, 98304
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
lib = _lib.load(); ctx = _lib.ctx()
lib.pgp_set_option(ctx, b"predict_batch", 131072)
for mode in (1, 0):
    lib.pgp_set_option(ctx, b"predict_inverse", mode)
    for pts in (32768, 49152, 57344, 65536, 81920, 98304):
        ts = []
        for i in range(7):
            t = time.perf_counter(); m.predict(xs[:pts]); wall = (time.perf_counter() - t) * 1e3
            ts.append("%.1f/%.1f" % (wall, _lib.last_timings()["total"]))
        print("predict_inverse %d, %6d points x 7 (wall/device ms): %s" % (mode, pts, " ".join(ts)), flush=True)
