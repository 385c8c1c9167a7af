"""Gram-form assembly (csrc/assemble.hip cov_gram_kernel / cov_gram_fast_kernel): full symmetric SEard K at N = 16384, d = 64
(pgp_test_assemble, HIP-event time over 100 launches after a warm-up), option sets alternated in one process.
    python tools/gram_probe.py [N=16384] [d=64] -- "gram_fast=0" "gram_fast=1" "gram_fast=1,gram_grid=4096" """
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
kv = dict(a.split("=") for a in args[:cut])
sets = args[cut + 1:] or ["gram_fast=0", "gram_fast=1"]
N, d = int(kv.get("N", 16384)), int(kv.get("d", 64))
lib = _lib.load()
ctx = _lib.ctx(0)
res = {s: [] for s in sets}
for r in range(int(kv.get("ROUNDS", 3))):
    for s in sets:
        for o in (sets[0] + "," + s).split(","):
            k_, v_ = o.split("=")
            assert lib.pgp_set_option(ctx, k_.encode(), int(v_)) == 0, o
        ms = C.c_double()
        assert lib.pgp_test_assemble(ctx, _lib.COV_RBFARD, 0, N, d, 100, C.byref(ms)) == 0
        res[s].append(ms.value)
b = 8.0 * N * N + 8.0 * N * d
for s in sets:
    m = float(np.median(res[s]))
    print("%-36s %s ms  median %.4f ms = %.3f of 8 TB/s" % (s, " ".join("%.4f" % v for v in res[s]), m, b / m / 1e6 / 8000.0))
