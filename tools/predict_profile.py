import sys, time, os, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, pygps_amd as pyGPs
n, d, ns = 8192, 16, 65536
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
m.predict(xs); m.predict(xs)
for _ in range(12):
    t = time.perf_counter(); m.predict(xs); print("predict %.1f ms" % ((time.perf_counter() - t) * 1e3))
pr = cProfile.Profile(); pr.enable(); m.predict(xs); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
lib.pgp_profile_reset(ctx); lib.pgp_set_profiling(ctx, 1)
t = time.perf_counter(); m.predict(xs); dt = time.perf_counter() - t
lib.pgp_set_profiling(ctx, 0)
print("profiled predict %.1f ms" % (dt * 1e3))
for k, v in _lib.profile().items():
    if v["launches"]:
        print("  %-60s %4d launches %8.2f ms  %6.1f TF" % (k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9))
for pb in (4096, 16384, 32768, 65536):
    lib.pgp_set_option(ctx, b"predict_batch", pb)
    m.predict(xs)
    t = time.perf_counter(); m.predict(xs); print("predict_batch %d: %.1f ms" % (pb, (time.perf_counter() - t) * 1e3))
