"""The rectangle TU_r of the sweep's trailing update (244 128 x 128 tiles at N = 8192: fewer than one per CU) alone on the chip,
as 128 x 128 LDS-DMA tiles (one workgroup per CU) and as 128 x 64 ones (two per CU): ms per launch, warm.
    python tools/tur_probe.py [M=7808] [N=512] [K=512]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

kv = dict(a.split("=") for a in sys.argv[1:])
M, N, K = int(kv.get("M", 7808)), int(kv.get("N", 512)), int(kv.get("K", 512))
lib = _lib.load()
ctx = _lib.ctx()
rng = np.random.RandomState(0)
A = np.asfortranarray(rng.randn(M, K) * 0.01)
B = np.asfortranarray(rng.randn(N, K) * 0.01)
for rnd in range(3):
    for tile in (128, 1264, 64):
        Cm = np.asfortranarray(rng.randn(M, N))
        ms = C.c_double()
        rc = lib.pgp_test_gemm(ctx, tile, 0, 0, 0, 0, 0, 0, -1.0, 1.0, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Cm), M, M, N, K, 200, C.byref(ms))
        assert rc == 0, rc
        print("M=%d N=%d K=%d tile %4d: %.1f us per launch, %.1f TF" % (M, N, K, tile, ms.value * 1e3, 2.0 * M * N * K / ms.value / 1e9), flush=True)
