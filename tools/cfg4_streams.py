"""cfg 4 on one GPU through GPR.optimize: fits/s against the number of concurrent restarts (fit streams) per GPU.
usage: python tools/cfg4_streams.py 2 4"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygps_amd as pyGPs
from pygps_amd import opt
x, y = bench.synth_reg(8192, 16)
from pygps_amd import _lib
OPTS = [a.split("=") for a in sys.argv[1:] if "=" in a]
for S in [int(a) for a in sys.argv[1:] if "=" not in a] or [2, 4]:
    opt.ShardedMinimize.streams_per_gpu = S
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(4.0), 0.0)); m.setNoise(np.log(0.1))
    m.setData(x, y)
    m.setOptimizer("ShardedMinimize", num_restarts=8)
    calls, lock, orig = [0], threading.Lock(), pyGPs.inf.Exact.evaluate
    def counted(self_, *a, **k):
        with lock:
            calls[0] += 1
        return orig(self_, *a, **k)
    pyGPs.inf.Exact.evaluate = counted
    np.random.seed(7); m.optimize(x, y, numIterations=2)
    for h in list(_lib._ctx.values()):
        for k_, v_ in OPTS:
            _lib.check(_lib.load().pgp_set_option(h, k_.encode(), int(v_)))
    for rep in range(2):
        calls[0] = 0
        np.random.seed(7)
        t = time.perf_counter(); m.optimize(x, y, numIterations=10); dt = time.perf_counter() - t
        print("streams %d: %d fits in %.2f s = %.1f fits/s, nlZ %.6f" % (S, calls[0], dt, calls[0] / dt, m.nlZ), flush=True)
    pyGPs.inf.Exact.evaluate = orig
