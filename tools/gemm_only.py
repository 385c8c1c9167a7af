"""A few launches of the fp64 MFMA GEMM at the Cholesky trailing-update shape, for rocprofv3 counter passes."""
import ctypes as C
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
ctx = _lib.ctx()
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tri = int(sys.argv[3]) if len(sys.argv) > 3 else 0
beta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 128
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
if len(sys.argv) > 7:
    lib.pgp_set_option(ctx, b"gemm_dbg", int(sys.argv[7]))
rng = np.random.RandomState(0)
A = np.asfortranarray(rng.randn(M, K))
B = np.asfortranarray(rng.randn(N, K))
Cm = np.asfortranarray(rng.randn(M, N))
ms = C.c_double()
rc = lib.pgp_test_gemm(ctx, tile, 0, 0, tri, 1 if tri else 0, 0, 0, -1.0, beta, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Cm),
                       M, M, N, K, iters, C.byref(ms))
print(sys.argv[1:], "rc", rc, "ms", ms.value, "TF", 2.0 * M * N * K * (0.5 if tri else 1) / ms.value / 1e9)
