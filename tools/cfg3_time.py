"""cfg 3: SEard N=16384 d=64 (65 cov hypers): stage timings of one fit with all gradients."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygps_amd as pyGPs
from pygps_amd import _lib
N, d = 16384, 64
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
m = pyGPs.GPR()
m.setPrior(kernel=pyGPs.cov.RBFard(log_ell_list=[float(np.log(np.sqrt(d)))] * d, log_sigma=0.0))
m.setNoise(np.log(0.1)); m.setData(x, y)
m.getPosterior()
for it in range(3):
    t = time.perf_counter(); m.getPosterior(); dt = time.perf_counter() - t
print("wall %.2f ms, %.1f TFLOP/s (N^3)" % (dt * 1e3, N ** 3 / dt / 1e12), _lib.last_timings())
