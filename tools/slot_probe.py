"""Dispatcher experiment: how long does a one-workgroup kernel with a big LDS request wait beside a resident grid?
(pgp_test_slot_probe; see csrc/testhooks.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
out = np.zeros(4)
print("holder nwg lds_kb hold_us reserve | probe lds_kb -> probe started X us after its launch (stayed, left)")
for nwg, lds, hold, res, plds in ((512, 74, 400, 0, 76), (496, 74, 400, 0, 76), (544, 74, 400, 1, 76), (560, 74, 400, 1, 76),
                                  (640, 74, 400, 1, 76), (544, 74, 400, 1, 40), (512, 74, 400, 0, 8), (256, 74, 400, 0, 76),
                                  (544, 74, 400, 1, 100), (1024, 74, 400, 1, 76)):
    for rep in range(2):
        rc = lib.pgp_test_slot_probe(ctx, nwg, lds, hold, res, plds, 60, _lib.ptr(out))
        print("%5d %3d %4d %d | %3d -> rc %d  waited %7.1f us  (stayed %d, left %d; probe start %.1f us after holder start)" % (
            nwg, lds, hold, res, plds, rc, out[0], out[1], out[2], out[3]))
