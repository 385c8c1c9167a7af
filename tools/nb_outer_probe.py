import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from pygps_amd import _lib
lib = _lib.load()
for N in (2048, 4096, 8192):
    d = 16
    rng = np.random.RandomState(0); x = rng.randn(N, d); y = rng.randn(N)
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    hyp = np.array([np.log(4.0), 0.0]); m = np.zeros(N); dm = np.ones((1, N)); alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    for want in (2, 3):
        for nb in (4, 8, 2, 6):
            lib.pgp_set_option(h, b"nb_outer", nb)
            ts = []
            for it in range(12):
                t = time.perf_counter()
                rc = lib.pgp_exact_fit(h, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, want, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
                assert rc == 0
                ts.append(time.perf_counter() - t)
            print("N=%d want=%d nb_outer=%d: %.3f ms  nlZ %.6f" % (N, want, nb, np.median(ts[2:]) * 1e3, nlZ[0]))
    lib.pgp_destroy(h)
# stage times of the value-only fit against the full one
for N in (4096, 8192):
    d = 16
    rng = np.random.RandomState(0); x = rng.randn(N, d); y = rng.randn(N)
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    hyp = np.array([np.log(4.0), 0.0]); m = np.zeros(N); dm = np.ones((1, N)); alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    for want in (1, 2, 3):
        for it in range(4):
            lib.pgp_exact_fit(h, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, want, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        st = np.zeros(len(_lib.STAGES)); lib.pgp_last_timings(h, _lib.ptr(st))
        print("N=%d want=%d stages" % (N, want), dict(zip(_lib.STAGES, np.round(st, 3))))
    lib.pgp_destroy(h)
