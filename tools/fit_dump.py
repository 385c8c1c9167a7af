"""One profiled N=8192 fit with every launch record dumped (PGP_PROF_DUMP=1): per-launch ms / TFLOP/s by class."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["PGP_PROF_DUMP"] = "1"
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
for o in sys.argv[1:]:
    k, v = o.split('='); lib.pgp_set_option(ctx, k.encode(), int(v))
N, d = 8192, 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
assert lib.pgp_set_data(ctx, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
for it in range(3):
    lib.pgp_set_profiling(ctx, 1 if it == 2 else 0)
    assert lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                             _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None) == 0
