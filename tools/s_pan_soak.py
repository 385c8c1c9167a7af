"""Soak of the s_pan schedule (S(p) on the panel stream, D's stage-out on the main stream, double-buffered scratch): the SAME fit
repeated, every output compared bit for bit with the first one -- a missing ordering between the streams shows up as a result that
differs from run to run.  python tools/s_pan_soak.py [N=8192] [REPS=300] [THREADS=1|2]"""
import ctypes as C
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

kv = dict(a.split("=") for a in sys.argv[1:])
N, REPS, T = int(kv.get("N", 8192)), int(kv.get("REPS", 300)), int(kv.get("THREADS", 1))
d = 16
lib = _lib.load()
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
bad = [0] * T


def worker(k):
    h = C.c_void_p()
    assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    assert lib.pgp_set_option(h, b"sched", 2) == 0                    # forced also when two threads run side by side
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    first = None
    for r in range(REPS):
        alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
        rc = lib.pgp_exact_fit(h, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                               _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        assert rc == 0, rc
        cur = (nlZ.copy(), alpha.copy(), g.copy())
        if first is None:
            first = cur
        elif not all(np.array_equal(a, b) for a, b in zip(first, cur)):
            bad[k] += 1
    lib.pgp_destroy(h)


ths = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print("N=%d reps=%d threads=%d: fits that differ from the first: %s" % (N, REPS, T, bad))
sys.exit(1 if any(bad) else 0)
