"""cfg 5 timing: GPC + RBF, infEP, N=4096 d=32 through the API (cold start each fit)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
for o in sys.argv[1:]:
    k, v = o.split('='); _lib.load().pgp_set_option(_lib.ctx(), k.encode(), int(v))
n5, d5 = 4096, 32
rng = np.random.RandomState(0)
x5 = rng.randn(n5, d5); w5 = rng.randn(d5, 1)
y5 = np.sign(x5 @ w5 / np.sqrt(d5) + 0.3 * rng.randn(n5, 1)); y5[y5 == 0] = 1
for it in range(4):
    m5 = pyGPs.GPC(); m5.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d5)), 0.0))
    t = time.perf_counter(); nlz = m5.getPosterior(x5, y5)[0]
    print("EP fit %.1f ms sweeps %d nlZ %.10f" % ((time.perf_counter() - t) * 1e3, m5.inffunc.sweeps, nlz))
