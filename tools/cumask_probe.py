"""What does a CU-masked stream cost the bulk GEMM?  (pgp_test_cumask_gemm; csrc/testhooks.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
out = np.zeros(2)
for M, K in ((8192, 512), (8192, 2048)):
    for stride in (8, 1):
        for res in (0, 1, 2, 4):
            kept = lib.pgp_test_cumask_gemm(ctx, M, K, res, stride, 20, _lib.ptr(out))
            fl = 2.0 * M * M * K
            print("M=%d K=%d enumeration %s reserve %d per XCD (kept %3d CUs): masked %.3f ms = %.1f TF | plain stream %.3f ms = %.1f TF | masked/plain time %.3f (CU ratio %.3f)"
                  % (M, K, "interleaved" if stride == 8 else "blocked", res, kept, out[0], fl / out[0] / 1e9, out[1], fl / out[1] / 1e9, out[0] / out[1], 256.0 / max(kept, 1)))
