"""Soak of the posterior handle's buffers (F, the fit's inverse rows E, W = L^-1) and predict's two forms under two fit streams:
each host thread runs fit -> predict (a few batch sizes, alternating predict_inverse 1 / 0) -> drop the posterior, for SECONDS;
every output must stay bit-identical from round to round within its thread and form.      python tools/predict_soak.py [SECONDS=30]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
import bench
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
lib = _lib.load()
stop = time.perf_counter() + SECONDS
out = {}


def loop(slot, n, d):
    with _lib.fit_stream(slot):
        x, y = bench.synth_reg(n, d)
        xs = np.random.RandomState(slot).randn(3000, d)
        ref, k = {}, 0
        while time.perf_counter() < stop:
            m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.setData(x, y)
            m.getPosterior()
            for mode in (1, 0, 1):
                lib.pgp_set_option(_lib.ctx(), b"predict_inverse", mode)
                for pts in (100, 1000, 3000):
                    ym, ys2, fm, fs2, lp = m.predict(xs[:pts])
                    sig = (float(np.sum(fm)), float(np.sum(fs2)), float(fm[pts // 2, 0]))
                    assert ref.setdefault((mode, pts), sig) == sig, ("predict moved", slot, k, mode, pts, sig, ref[(mode, pts)])
            lib.pgp_set_option(_lib.ctx(), b"predict_inverse", 1)
            k += 1
        out[slot] = k


ths = [threading.Thread(target=loop, args=(0, 4096, 16)), threading.Thread(target=loop, args=(1, 6000, 8))]
[t.start() for t in ths]; [t.join() for t in ths]
print("fit + 9 predicts per round: %d rounds on stream 0 (N = 4096), %d on stream 1 (N = 6000) in %.0f s, every output bit-identical" % (out[0], out[1], SECONDS))
