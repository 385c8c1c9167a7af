"""Two identical K=512 GEMM loops on two contexts (two HIP streams) at once vs one alone: is the per-tile C traffic /
pipeline fill hidden when co-resident workgroups are out of phase?"""
import ctypes as C, sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo")
from pygps_amd import _lib
lib = _lib.load()
M = N = 8192; K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = 20
rng = np.random.RandomState(0)
A = np.asfortranarray(rng.randn(M, K)); B = np.asfortranarray(rng.randn(N, K)); Cm = np.asfortranarray(rng.randn(M, N))
ctxs = []
for k in range(2):
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0; ctxs.append(h)
def run(h, out, k):
    ms = C.c_double()
    Ck = Cm.copy(order="F")
    rc = lib.pgp_test_gemm(h, 128, 0, 0, 0, 0, 0, 0, -1.0, 1.0, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(Ck), M, M, N, K, iters, C.byref(ms))
    out[k] = ms.value
out = [0, 0]
run(ctxs[0], out, 0)
print("alone: %.3f ms per GEMM, %.1f TF" % (out[0], 2.0 * M * N * K / out[0] / 1e9))
ths = [threading.Thread(target=run, args=(ctxs[k], out, k)) for k in range(2)]
t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
print("two at once: %.3f / %.3f ms per GEMM each -> combined %.1f TF (if fully overlapped)" % (out[0], out[1], 2.0 * M * N * K * (1 / out[0] + 1 / out[1]) / 1e9))
