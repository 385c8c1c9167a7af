"""One exact fit (nlZ + gradients) per size on one GPU: ms per fit and N^3-flop rate, with a residual check of alpha
(|| (K + sn2 I) alpha - (y - m) || / ||y - m||, K rebuilt block-wise on the host) at the sizes the host can afford."""
import ctypes as C
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygps_amd import _lib

lib = _lib.load()
sizes = [int(a) for a in sys.argv[1:] if '=' not in a] or [1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768]
OPTS = [a.split('=') for a in sys.argv[1:] if '=' in a]
d = 16
for N in sizes:
    rng = np.random.RandomState(0)
    x = rng.randn(N, d); w = rng.randn(d, 1)
    y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
    h = C.c_void_p()
    assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    for k_, v_ in OPTS:
        assert lib.pgp_set_option(h, k_.encode(), int(v_)) == 0
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        rc = lib.pgp_exact_fit(h, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                               _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        ts.append(time.perf_counter() - t)
        assert rc == 0, rc
    dt = min(ts)
    res = ""
    if N <= 32768:
        ell, sf2, sn2 = np.exp(hyp[0]), np.exp(2 * hyp[1]), 0.01
        r = np.zeros(N)
        xs = x / ell
        sq = (xs * xs).sum(1)
        for i0 in range(0, N, 4096):
            D2 = sq[i0:i0 + 4096, None] + sq[None, :] - 2.0 * xs[i0:i0 + 4096] @ xs.T
            r[i0:i0 + 4096] = (sf2 * np.exp(-0.5 * np.maximum(D2, 0.0))) @ alpha
        r += sn2 * alpha - (y - m)
        res = "  residual %.1e" % (np.linalg.norm(r) / np.linalg.norm(y - m))
    print("N %6d: %9.2f ms per fit  %6.1f TF (N^3 / t)  %.3f of fp64-MFMA peak  nlZ %.6f%s" % (
        N, dt * 1e3, float(N) ** 3 / dt / 1e12, float(N) ** 3 / dt / 1e12 / 78.6, nlZ[0], res), flush=True)
    lib.pgp_destroy(h)
