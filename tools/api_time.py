"""Per-evaluation wall time through the Python API (GPR.getPosterior, what the optimiser calls) vs the bare C call."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import pygps_amd as pyGPs
for N in (8192, 2048, 256, 20):
    d = 16 if N > 20 else 1
    rng = np.random.RandomState(0)
    x = rng.randn(N, d); w = rng.randn(d, 1)
    y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
    m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.setData(x, y)
    m.getPosterior(); m.getPosterior()
    reps = 20
    t = time.perf_counter()
    for it in range(reps):
        m.covfunc.hyp = [np.log(np.sqrt(d)) + 1e-4 * it, 0.0]
        m.getPosterior()
    print("N=%5d: %.3f ms per getPosterior (nlZ + gradients) through the API" % (N, (time.perf_counter() - t) / reps * 1e3))
