"""Debug: exact fit factor vs numpy Cholesky, block-wise (512) error map."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
want = 3
for o in sys.argv[2:]:
    k, v = o.split('=')
    if k == "want": want = int(v)
    else: lib.pgp_set_option(ctx, k.encode(), int(v))
d = 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
assert lib.pgp_set_data(ctx, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
h = C.c_void_p()
rc = lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, want,
                       _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), C.byref(h))
print("rc", rc, "nlZ", nlZ[0])
if rc == 0:
    R = np.zeros((N, N))
    assert lib.pgp_factor_to_host(ctx, h, _lib.ptr(R)) == 0
    from scipy.spatial.distance import cdist
    K = np.exp(-0.5 * cdist(x / np.sqrt(d), x / np.sqrt(d), 'sqeuclidean'))
    Lr = np.linalg.cholesky(K / 0.01 + np.eye(N))
    L = R.T
    nb = (N + 511) // 512
    for i in range(nb):
        print(" ".join("%9.2e" % np.abs(L[i*512:(i+1)*512, j*512:(j+1)*512] - Lr[i*512:(i+1)*512, j*512:(j+1)*512]).max() for j in range(i + 1)))
