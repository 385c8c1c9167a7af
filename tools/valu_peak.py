"""fp64 VALU issue rate (v_add_f64 / v_fma_f64 mix of the distance loop) for 1..4 waves per SIMD."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
o = np.zeros(2)
for w in (1, 2, 4, 8):
    for it in (20000, 200000):
        assert lib.pgp_test_valu_peak(ctx, it, w, _lib.ptr(o)) == 0
        print("waves/SIMD %d iters %6d: %.1f wave-instr/ns chip-wide, %.2f cycles/instr @2.4GHz -> %.1f TFLOP/s if all were FMA" % (
            w, it, o[0], o[1], o[0] * 128 * 1e9 / 1e12))
