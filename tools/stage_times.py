"""Stage times of N=8192 (or N=<n>) fits under library options."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
N = 8192
for o in sys.argv[1:]:
    k, v = o.split('=')
    if k == "N": N = int(v)
    else: lib.pgp_set_option(ctx, k.encode(), int(v))
d = 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
assert lib.pgp_set_data(ctx, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
for it in range(4):
    rc = lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                           _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
    print(sys.argv[1:], "rc", rc, {k: round(v, 3) for k, v in _lib.last_timings().items()})
