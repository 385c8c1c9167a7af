import sys, ctypes as C
sys.path.insert(0, "/root/repo")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx(0)
def asm(n, d, iters, kind=0):
    ms = C.c_double(); assert lib.pgp_test_assemble(ctx, kind, 0, n, d, iters, C.byref(ms)) == 0
    return ms.value
seq = sys.argv[1:]
for a in seq:
    n, it = a.split(":")
    print("n=%s iters=%s: %.4f ms" % (n, it, asm(int(n), 16, int(it))), flush=True)
