import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
n, d, ns = 8192, 16, 65536
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
lib = _lib.load(); ctx = _lib.ctx()
for mode in (1, 0):
    lib.pgp_set_option(ctx, b"predict_inverse", mode)
    for pts in (8192, 65536, 1024):
        ts = []
        for i in range(8):
            t = time.perf_counter(); m.predict(xs[:pts]); ts.append((time.perf_counter() - t) * 1e3)
        print("predict_inverse %d, %5d points x 8: %s ms" % (mode, pts, " ".join("%.1f" % v for v in ts)), flush=True)
        ts = []
        for i in range(5):
            time.sleep(0.2)
            t = time.perf_counter(); m.predict(xs[:pts]); ts.append((time.perf_counter() - t) * 1e3)
        print("   ... with 0.2 s pauses: %s ms" % " ".join("%.1f" % v for v in ts), flush=True)
