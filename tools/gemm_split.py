"""One K=512 GEMM (M=N=8192) vs the same work as two half-width GEMMs issued concurrently from two contexts."""
import ctypes as C, sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pygps_amd import _lib
lib = _lib.load()
M = 8192; K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = 20
rng = np.random.RandomState(0)
A = np.asfortranarray(rng.randn(M, K))
ctxs = []
for k in range(2):
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0; ctxs.append(h)
def run(h, N, out, k):
    ms = C.c_double()
    B = np.asfortranarray(rng.randn(N, K)); Ck = np.asfortranarray(np.zeros((M, N)))
    rc = lib.pgp_test_gemm(h, 128, 0, 0, 0, 0, 0, 0, -1.0, 1.0, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Ck), M, M, N, K, iters, C.byref(ms))
    out[k] = ms.value
out = [0, 0]
run(ctxs[0], 8192, out, 0)
print("one GEMM N=8192: %.3f ms, %.1f TF" % (out[0], 2.0 * M * 8192 * K / out[0] / 1e9))
run(ctxs[0], 4096, out, 0)
print("one GEMM N=4096 alone: %.3f ms, %.1f TF" % (out[0], 2.0 * M * 4096 * K / out[0] / 1e9))
for rep in range(2):
    ths = [threading.Thread(target=run, args=(ctxs[k], 4096, out, k)) for k in range(2)]
    [th.start() for th in ths]; [th.join() for th in ths]
    print("two concurrent N=4096 halves: %.3f / %.3f ms each -> the pair in ~%.3f ms = %.1f TF" % (out[0], out[1], max(out), 2.0 * M * 8192 * K / max(out) / 1e9))
