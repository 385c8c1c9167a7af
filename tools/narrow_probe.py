"""The bulk GEMM on a NARROW output (M x 512: the columns of one panel, one workgroup per CU) against the depth of K:
what a left-looking update of one panel by all earlier ones would run at."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
ctx = _lib.ctx()
rng = np.random.RandomState(0)
for M, N in ((8192, 512), (8192, 1024), (16384, 512)):
    for K in (512, 1024, 2048, 4096, 8192):
        A = np.asfortranarray(rng.randn(M, K))
        B = np.asfortranarray(rng.randn(N, K))
        Cm = np.asfortranarray(rng.randn(M, N))
        ms = C.c_double()
        rc = lib.pgp_test_gemm(ctx, 128, 0, 0, 0, 0, 0, 0, -1.0, 1.0, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Cm), M, M, N, K, 10, C.byref(ms))
        print("M=%d N=%d K=%5d  tiles %4d  %.1f us  %.1f TF" % (M, N, K, M // 128 * (N // 128), ms.value * 1e3, 2.0 * M * N * K / ms.value / 1e9), flush=True)
