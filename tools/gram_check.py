"""Gram-form getCovMatrix('train') (csrc/assemble.hip cov_gram_fast_kernel) against the difference form, d = 32 / 48 / 64, RBF and RBFard.
    python tools/gram_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
for d in (32, 48, 64):
    for n in (4096, 4160):
        x = np.random.RandomState(d).randn(n, d)
        k = pyGPs.cov.RBFard(log_ell_list=[float(np.log(np.sqrt(d)) + 0.01 * (i % 5)) for i in range(d)], log_sigma=0.2)
        lib.pgp_set_option(ctx, b"gram_assembly", 0); K0 = k.getCovMatrix(x=x, mode="train")
        lib.pgp_set_option(ctx, b"gram_assembly", 1); K1 = k.getCovMatrix(x=x, mode="train")
        e = np.abs(K1 - K0) / K0
        i, j = np.unravel_index(np.argmax(e), e.shape)
        bad = np.argwhere(e > 1e-12)
        print("d=%d n=%d  max rel err %.3e at (%d,%d): %.17g vs %.17g; #bad %d; sym %s diag %s; bad tiles %s"
              % (d, n, e.max(), i, j, K1[i, j], K0[i, j], len(bad), np.array_equal(K1, K1.T), np.all(np.diag(K1) == np.exp(0.4)),
                 sorted(set((int(a) // 64, int(b) // 64) for a, b in bad[:2000]))[:12]), flush=True)
