"""Mixed soak: one host thread runs cold EP fits (cfg 5 shape), another exact fits (cfg 2 shape) on a second fit stream of the
same GPU, for SECONDS (argv[1]; further k=v arguments are library options of the EP context).  Both results must stay bit-identical from call to call: the EP chain kernel hands data between its
waves through LDS sequence counters, and a co-running bulk workload changes every timing in it."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
import bench
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n5, d5 = 4096, 32
rng = np.random.RandomState(0)
x5 = rng.randn(n5, d5); w5 = rng.randn(d5, 1)
y5 = np.sign(x5 @ w5 / np.sqrt(d5) + 0.3 * rng.randn(n5, 1)); y5[y5 == 0] = 1
x2, y2 = bench.synth_reg(8192, 16)
stop = time.perf_counter() + SECONDS
out = {}

def ep_loop():
    with _lib.fit_stream(0):
        for o in sys.argv[2:]:                     # k=v library options for the EP context
            k_, v_ = o.split('='); _lib.load().pgp_set_option(_lib.ctx(), k_.encode(), int(v_))
        ref, k = None, 0
        while time.perf_counter() < stop:
            m = pyGPs.GPC(); m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d5)), 0.0))
            nlz, dnlz, post = m.getPosterior(x5, y5)
            sig = (float(nlz), float(np.sum(post.alpha)), float(np.sum(dnlz.cov)))
            if ref is None: ref = sig
            assert sig == ref, ("EP result moved", k, sig, ref)
            k += 1
        out["ep"] = (k, ref)

def exact_loop():
    with _lib.fit_stream(1):
        ref, k = None, 0
        m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(4.0), 0.0)); m.setNoise(np.log(0.1))
        while time.perf_counter() < stop:
            nlz, dnlz, post = m.getPosterior(x2, y2)
            sig = (float(nlz), float(np.sum(post.alpha)), float(np.sum(dnlz.cov)))
            if ref is None: ref = sig
            assert sig == ref, ("exact result moved", k, sig, ref)
            k += 1
        out["exact"] = (k, ref)

ths = [threading.Thread(target=ep_loop), threading.Thread(target=exact_loop)]
[t.start() for t in ths]; [t.join() for t in ths]
print("EP fits %d (nlZ %.10f), exact fits %d (nlZ %.10f) in %.0f s, all bit-identical" % (out["ep"][0], out["ep"][1][0], out["exact"][0], out["exact"][1][0], SECONDS))
