"""Reproduce the K-fold EP scenario of tests/test_gpu_r5.py outside pytest and probe the streams when a sweep's wait gives up."""
import os, sys, time
os.environ.setdefault("PYGPS_AMD_TORCH_FIRST", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if os.environ.get("WITH_TORCH", "1") == "1":
    import torch
import pygps_amd as pyGPs
from pygps_amd import _lib, valid
from conftest import synth_cls, synth_reg
lib = _lib.load()


def probe(tag):
    for slot in (0, 1):
        o = np.zeros(2)
        rc = lib.pgp_test_stream_concurrency(_lib.ctx(0, slot), 20000, _lib.ptr(o))
        print("  probe %s slot %d: rc %d concurrent=%d waited %.0f us" % (tag, slot, rc, int(o[0]), o[1]), flush=True)


if os.environ.get("REG_FIRST", "1") == "1":
    x, y = synth_reg(2048, 16)
    def mk():
        m = pyGPs.GPR(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(4.0), 0.0)); m.setNoise(np.log(0.1)); return m
    valid.sharded_k_fold(mk, x, y, K=10)
probe("start")
N, d, K = 600, 8, 5
x, y = synth_cls(N, d)
def make():
    m = pyGPs.GPC(); m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); return m
S = int(os.environ.get("STREAMS", "2"))
for rep in range(int(os.environ.get("REPS", "6"))):
    t = time.time()
    res = valid.sharded_k_fold(make, x, y, K=K, metrics=("ACC",), streams_per_gpu=S)
    bad = np.isnan(res["nlZ"]).sum()
    print("rep %d: %.2f s, failed folds %d" % (rep, time.time() - t, bad), flush=True)
    if bad:
        probe("after failure")
