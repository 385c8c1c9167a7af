"""Timing of one FITC fit (nlZ + gradients) through GPR_FITC for a few (n, nu); CPU oracle beside it for the small one."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import pygps_amd as pyGPs
from oracle import gp_oracle as O

for n, nu, d in ((16384, 512, 16), (131072, 1024, 16), (262144, 2048, 16)):
    rng = np.random.RandomState(0)
    x = rng.randn(n, d); w = rng.randn(d, 1)
    y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
    u = x[rng.choice(n, nu, replace=False)] + 0.01 * rng.randn(nu, d)
    m = pyGPs.GPR_FITC()
    m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0), inducing_points=u)
    m.setNoise(np.log(0.1))
    m.getPosterior(x, y)
    t = time.perf_counter()
    for it in range(3):
        m.covfunc.hyp = [np.log(np.sqrt(d)) + 1e-3 * it, 0.0]
        nlZ, dnlZ, post = m.getPosterior(x, y)
    dt = (time.perf_counter() - t) / 3
    fl = 2.0 * nu * nu * n * (4 + 2 * 2)          # V, VsVs', B, W, BW' + per hyper (R, RW') x 2 hypers
    print("n=%d nu=%d d=%d: %.1f ms per fit (nlZ + 3 gradients), ~%.1f TFLOP/s on the nu^2 n products, nlZ %.6f" % (
        n, nu, d, dt * 1e3, fl / dt / 1e12, nlZ))
    if n <= 16384:
        t = time.perf_counter()
        out = O.fitc_fit(O.RBF, np.array(m.covfunc.hyp), 0, np.log(0.1), x, u, y, np.zeros_like(y), None)
        print("   CPU oracle (reference algorithm, numpy/scipy): %.2f s, nlZ %.6f" % (time.perf_counter() - t, out["nlZ"]))
