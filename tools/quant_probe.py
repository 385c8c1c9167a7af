"""Round quantisation of the bulk GEMM: K = 512 launches whose tile counts straddle the 512 workgroup slots of the chip
(2 per CU): time per launch against the number of 128 x 128 tiles."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
ctx = _lib.ctx()
K = 512
rng = np.random.RandomState(0)
for mt, nt in ((12, 32), (16, 32), (17, 32), (20, 32), (24, 32), (28, 32), (32, 32), (33, 32), (40, 32), (48, 32), (64, 32)):
    M, N = 128 * mt, 128 * nt
    A = np.asfortranarray(rng.randn(M, K))
    B = np.asfortranarray(rng.randn(N, K))
    Cm = np.asfortranarray(rng.randn(M, N))
    ms = C.c_double()
    rc = lib.pgp_test_gemm(ctx, 128, 0, 0, 0, 0, 0, 0, -1.0, 1.0, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Cm), M, M, N, K, 20,
                           C.byref(ms))
    t = mt * nt
    print("tiles %5d (%.2f rounds of 512)  %.1f us per launch  %.1f us per round-up  %.1f TF" % (
        t, t / 512.0, ms.value * 1e3, ms.value * 1e3 / -(-t // 512), 2.0 * M * N * K / ms.value / 1e9), flush=True)
