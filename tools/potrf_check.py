"""Debug: pgp_potrf on a random SPD matrix vs numpy, block-wise error map (512-panels)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for o in sys.argv[2:]:
    k, v = o.split('='); lib.pgp_set_option(ctx, k.encode(), int(v))
rng = np.random.RandomState(1)
G = rng.randn(n, n)
A = G @ G.T / n + np.eye(n)
L = np.zeros((n, n))
rc = lib.pgp_potrf(ctx, _lib.ptr(A), n, _lib.ptr(L))
print("rc", rc)
Lr = np.linalg.cholesky(A)
nb = (n + 511) // 512
for i in range(nb):
    print(" ".join("%9.2e" % np.abs(L[i*512:(i+1)*512, j*512:(j+1)*512] - Lr[i*512:(i+1)*512, j*512:(j+1)*512]).max() for j in range(i + 1)))
