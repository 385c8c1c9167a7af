"""Per-launch profile of one FITC fit (PGP_PROF_DUMP) + wall time of the call."""
import os, sys, time
import numpy as np
os.environ["PGP_PROF_DUMP"] = "1"
sys.path.insert(0, "/root/repo")
import pygps_amd as pyGPs
from pygps_amd import _lib
n, nu, d = 131072, 1024, 16
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1)
y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
u = x[rng.choice(n, nu, replace=False)] + 0.01 * rng.randn(nu, d)
m = pyGPs.GPR_FITC()
m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0), inducing_points=u)
m.setNoise(np.log(0.1))
m.getPosterior(x, y); m.getPosterior(x, y)
lib = _lib.load()
lib.pgp_set_profiling(_lib.ctx(), 1)
t = time.perf_counter(); m.getPosterior(x, y); print("wall ms (profiled)", (time.perf_counter() - t) * 1e3)
