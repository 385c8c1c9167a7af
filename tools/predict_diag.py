"""GP.predict at N = 8192, 65536 test points (Core/gp.py:395-417): where a call's wall time goes -- scratch allocation (host), device
time, the rest of the host side -- for the FIRST call at a batch shape and for later ones, by batch size.
    python tools/predict_diag.py"""
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


def one(tag, pts=ns):
    t = time.perf_counter(); m.predict(xs[:pts]); wall = (time.perf_counter() - t) * 1e3
    lt = _lib.last_timings()
    print("%-44s wall %7.1f ms | library call %7.1f  scratch alloc %7.1f  device %7.1f | python side %6.1f"
          % (tag, wall, lt["solve"], lt["assemble"], lt["total"], wall - lt["solve"]), flush=True)


one("warm-up, 4096 points (what bench.py r5 did)", 4096)
one("FIRST call, 65536 points, batch 65536")
for i in range(4):
    one("call %d" % (i + 2))
for pts in (1024, 8192):
    one("%d points" % pts, pts); one("%d points again" % pts, pts)
lib.pgp_set_option(ctx, b"predict_inverse", 0)
print("predict_inverse = 0: the blocked triangular solve")
one("65536 points"); one("65536 points again")
for pts in (1024, 8192):
    one("%d points" % pts, pts); one("%d points again" % pts, pts)
for pb in (8192, 16384, 32768, 65536):
    lib.pgp_set_option(ctx, b"predict_batch", pb)
    one("batch %d: first" % pb)
    one("batch %d: second" % pb)
    one("batch %d: third" % pb)
