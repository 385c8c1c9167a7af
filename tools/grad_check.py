"""dnlZ of one exact fit under library options, for several N (A/B of schedule variants)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from pygps_amd import _lib

lib = _lib.load()
opts = [tuple(o.split("=")) for o in sys.argv[1:] if "=" in o]
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [2048, 8192]
for N in sizes:
    d = 16
    rng = np.random.RandomState(0)
    x = rng.randn(N, d); w = rng.randn(d, 1)
    y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
    res = []
    for use in (False, True):
        h = C.c_void_p()
        assert lib.pgp_init(0, C.byref(h)) == 0
        assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
        if use:
            for k_, v_ in opts:
                assert lib.pgp_set_option(h, k_.encode(), int(v_)) == 0
        hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
        alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
        for rep in range(2):
            rc = lib.pgp_exact_fit(h, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                                   _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
            assert rc == 0
            res.append((nlZ[0], g.copy()))
        lib.pgp_destroy(h)
    print("N", N, "default", res[0][1], res[1][1])
    print("N", N, opts, res[2][1], res[3][1])
