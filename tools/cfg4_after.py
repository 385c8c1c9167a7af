"""cfg 4 (two fit streams, one GPU) after another workload ran in the same process: python tools/cfg4_after.py [ep|api|predict|none]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygps_amd as pyGPs
from pygps_amd import opt, _lib
what = sys.argv[1] if len(sys.argv) > 1 else "none"
x, y = bench.synth_reg(8192, 16)
if what == "ep":
    n5, d5 = 4096, 32
    rng = np.random.RandomState(0)
    x5 = rng.randn(n5, d5); w5 = rng.randn(d5, 1)
    y5 = np.sign(x5 @ w5 / np.sqrt(d5) + 0.3 * rng.randn(n5, 1)); y5[y5 == 0] = 1
    for it in range(2):
        m5 = pyGPs.GPC(); m5.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d5)), 0.0))
        m5.getPosterior(x5, y5)
elif what.startswith("api"):
    if what == "api3":
        _lib.check(_lib.load().pgp_set_option(_lib.ctx(), b"eet_overlap", 3))
    print(bench.api_rate(8192, 16, x, y, 20))
    if what == "api3":
        _lib.check(_lib.load().pgp_set_option(_lib.ctx(), b"eet_overlap", 4))
opt.ShardedMinimize.streams_per_gpu = 2
m = pyGPs.GPR()
m.setPrior(kernel=pyGPs.cov.RBF(np.log(4.0), 0.0)); m.setNoise(np.log(0.1))
m.setData(x, y)
m.setOptimizer("ShardedMinimize", num_restarts=8)
np.random.seed(7); m.optimize(x, y, numIterations=2)
np.random.seed(7)
t = time.perf_counter(); m.optimize(x, y, numIterations=10); dt = time.perf_counter() - t
print("after %s: cfg 4 in %.2f s" % (what, dt), flush=True)
