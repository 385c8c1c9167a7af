"""Cost of single instructions for ONE workgroup on an idle chip (s_memtime ticks; see pgp_test_wave_costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygps_amd import _lib
lib = _lib.load()
out = np.zeros(16)
_lib.check(lib.pgp_test_wave_costs(_lib.ctx(), _lib.ptr(out)))
names = ["dependent v_fma_f64", "(100 MHz ticks of test 0)", "8 independent fma chains, per fma", "dependent LDS read", "20 LDS reads + use",
         "s_barrier", "LDS write -> barrier -> read", "dependent v_rcp_f64 (+add)", "dependent v_rsq_f64 (+add)", "(s_memtime ticks of test 0)"]
for n_, v in zip(names, out):
    print("%-40s %10.2f" % (n_, v))
print("s_memtime rate: %.1f MHz" % (out[9] / out[1] * 100.0))
