"""GP.predict throughput on an N=8192 posterior under library options (predict_batch=...)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
for o in sys.argv[1:]:
    k, v = o.split('='); _lib.load().pgp_set_option(_lib.ctx(), k.encode(), int(v))
n, d, ns = 8192, 16, 65536
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
m.predict(xs)
m.predict(xs)
t = time.perf_counter(); out = m.predict(xs); dt = time.perf_counter() - t
print(sys.argv[1:], "%.1f ms, %.0f points/s, %.1f TF, fs2[0] %.12f" % (dt * 1e3, ns / dt, 1.0 * n * n * ns / dt / 1e12, out[3][0, 0]))
