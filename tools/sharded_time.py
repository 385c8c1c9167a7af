"""World-size-1 timing of the sharded fit (pgp_sharded_exact_fit) against the single-GPU fit, by size:
python tools/sharded_time.py [N ...]   (options as name=value are passed to pgp_set_option)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pygps_amd as pyGPs
from pygps_amd import _lib, sharded

lib = _lib.load()
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 16384, 32768]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
for k, v in opts:
    assert lib.pgp_set_option(_lib.ctx(), k.encode(), int(v)) == 0
comm = sharded.Comm()
for N in sizes:
    d = 16
    rng = np.random.RandomState(0)
    x = rng.randn(N, d); w = rng.randn(d, 1)
    y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
    res = {}
    for mode in ("sharded", "single"):
        m = pyGPs.GPR()
        m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1))
        m.setData(x, y)
        m.inffunc = pyGPs.inf.Exact(sharded=comm if mode == "sharded" else False)
        best = 1e9
        for rep in range(3):
            m.covfunc.hyp[0] = float(np.log(np.sqrt(d)) + 1e-4 * rep)
            t = time.perf_counter()
            nlZ, dnlZ, post = m.getPosterior()
            best = min(best, time.perf_counter() - t)
        res[mode] = (best, nlZ, getattr(m.inffunc, "last_ms", None))
        del m, post
    npad = -(-N // (1024 if N >= 12288 else 512)) * (1024 if N >= 12288 else 512)
    ms = res["sharded"][2]
    print("N=%6d  sharded %.2f ms (%.1f TF, %.3f of peak; stages ms %s; sweep+EEt alone %.3f of peak)   single %.2f ms (%.1f TF)   nlZ diff %.1e"
          % (N, res["sharded"][0] * 1e3, N ** 3 / res["sharded"][0] / 1e12, N ** 3 / res["sharded"][0] / 78.6e12, np.round(ms, 2),
             float(npad) ** 3 / (ms[1] * 1e-3) / 78.6e12, res["single"][0] * 1e3, N ** 3 / res["single"][0] / 1e12,
             abs(res["sharded"][1] - res["single"][1]) / abs(res["single"][1])), flush=True)
