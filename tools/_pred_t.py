"""Where the wall time of GP.predict goes when the BLAS pool is NOT capped at the container's CPU quota (PYGPS_AMD_KEEP_THREADS=1):
the library call, then the reference's host code behind it (lik.Gauss.evaluate: numpy.linalg.norm, Core/lik.py:134-158), with the
cgroup's throttle counters before and after.     python tools/_pred_t.py        PYGPS_AMD_KEEP_THREADS=1 python tools/_pred_t.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib, _threads


def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return "nr_throttled %s, throttled %.1f s" % (d.get("nr_throttled"), int(d.get("throttled_usec", 0)) / 1e6)
    except Exception as e:
        return "cpu.stat unreadable (%r)" % (e,)


n, d, ns = 8192, 16, 32768
rng = np.random.RandomState(0)
x = rng.randn(n, d); w = rng.randn(d, 1); y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(n, 1)
m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.getPosterior(x, y)
xs = np.random.RandomState(1).randn(ns, d)
lib = _lib.load()
L = m.posterior.L
print("CPU quota %s cores, visible %d, pools capped: %s; %s" % (_threads.cpu_quota(), len(os.sched_getaffinity(0)), _threads._LIMIT is not None or "OPENBLAS_NUM_THREADS" in os.environ, throttled()))
for i in range(6):
    t0 = time.perf_counter()
    xs_ = _lib.f64(xs); ms = _lib.f64(m.meanfunc.getMean(xs_)).reshape(ns); fmu = np.empty(ns); fs2 = np.empty(ns)
    t1 = time.perf_counter()
    rc = lib.pgp_predict(L.ctx, L.handle, _lib.ptr(xs_), ns, _lib.ptr(ms), _lib.ptr(fmu), _lib.ptr(fs2))
    t2 = time.perf_counter()
    lp, ymu, ys2 = m.likfunc.evaluate(None, fmu.reshape(ns, 1), fs2.reshape(ns, 1), None, None, 3)
    t3 = time.perf_counter()
    print("call %d, %d points: pgp_predict %.2f ms (device %.2f), lik.Gauss.evaluate on the host %.2f ms" % (i, ns, (t2 - t1) * 1e3, _lib.last_timings()["total"], (t3 - t2) * 1e3), flush=True)
print(throttled())
