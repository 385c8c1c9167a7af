import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
rng = np.random.RandomState(0)
def run(M, N, K, kmode, lda, ldb, beta, iters=6):
    A = np.asfortranarray(rng.randn(lda, K) * 0.01); B = np.asfortranarray(rng.randn(ldb, K) * 0.01); Cm = np.asfortranarray(rng.randn(M, N))
    ms = C.c_double()
    rc = lib.pgp_test_gemm(ctx, 128, 0, 0, 0, 0, kmode, 0, 1.0, beta, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(Cm), M, M, N, K, iters, C.byref(ms))
    fl = 2.0 * M * N * K * (0.5 if kmode else 1.0)
    print("M=%d N=%d K=%d kmode=%d lda=%d ldb=%d beta=%g: rc %d  %.3f ms  %.1f TF (algorithmic)" % (M, N, K, kmode, lda, ldb, beta, rc, ms.value, fl / ms.value / 1e9), flush=True)
KM = int(sys.argv[1]) if len(sys.argv) > 1 else 3
run(8192, 8192, 8192, 0, 8192, 8192, 1.0)
run(8192, 8192, 8192, 0, 8320, 8192, 0.0)
run(8192, 8192, 8192, KM, 8320, 8192, 0.0)
run(8192, 8192, 8192, KM, 8320, 8320, 0.0)
run(8192, 1024, 8192, KM, 8320, 1024, 0.0, 20)
run(8192, 1024, 8192, 0, 8320, 1024, 0.0, 20)
run(8192, 1024, 8192, 0, 8320, 1152, 0.0, 20)
run(8192, 2048, 8192, 0, 8320, 2048, 0.0, 10)
run(8192, 2048, 8192, 0, 8320, 2176, 0.0, 10)
