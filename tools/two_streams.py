"""Throughput with S independent fit streams on ONE GPU (one pgp_ctx per stream, one host thread each)."""
import ctypes as C
import sys
import threading
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
N, d = 8192, 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()


def worker(ctx, steps, out, k):
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    for s in range(steps):
        hyp[0] = np.log(np.sqrt(d)) + 1e-4 * (s + k)
        rc = lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                               _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        assert rc == 0
    out[k] = nlZ[0]


import os
opts = [tuple(o.split("=")) for o in sys.argv[1:]]
for S in tuple(int(v) for v in os.environ.get("NSTREAMS", "1,2").split(",")):
    ctxs = []
    for k in range(S):
        h = C.c_void_p()
        assert lib.pgp_init(0, C.byref(h)) == 0
        assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
        for k_, v_ in opts:
            assert lib.pgp_set_option(h, k_.encode(), int(v_)) == 0
        ctxs.append(h)
    out = [0] * S
    for steps in (2, 12):
        ths = [threading.Thread(target=worker, args=(ctxs[k], steps, out, k)) for k in range(S)]
        t = time.time()
        [th.start() for th in ths]
        [th.join() for th in ths]
        dt = time.time() - t
    print(opts, "streams %d: %d fits in %.1f ms -> %.2f ms/fit, %.1f fits/s  nlZ %s" % (S, S * steps, dt * 1e3, dt * 1e3 / (S * steps), S * steps / dt, out[0]))
    for h in ctxs:
        lib.pgp_destroy(h)
