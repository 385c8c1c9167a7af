"""Device-only timing of the kernel-assembly tile kernel (pgp_test_assemble) for a few iteration counts."""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
ms = C.c_double()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for iters in (3, 10, 100, 1000, 10):
    for mode in (0, 2):
        assert lib.pgp_test_assemble(ctx, 0, mode, n, 16, iters, C.byref(ms)) == 0
        by = 8.0 * n * n * (1.0 if mode == 0 else 0.5)
        print("iters %4d mode %d: %.4f ms  %.0f GB/s" % (iters, mode, ms.value, by / ms.value / 1e6))
