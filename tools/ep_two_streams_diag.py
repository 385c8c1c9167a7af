import os, sys, threading, time
if os.environ.get("WITH_TORCH"):
    import torch
    torch.zeros(4, device="cuda").sum().item()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pygps_amd as pyGPs
from pygps_amd import _lib
from conftest import synth_cls
x, y = synth_cls(1024, 8)
if os.environ.get("WITH_COMM"):
    from pygps_amd import sharded
    cm = sharded.search_comm(None)
    print("comm", cm.transport, cm.world)
def fit():
    m = pyGPs.GPC()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(8.0)), 0.0))
    t = time.time()
    nlZ, dnlZ, post = m.getPosterior(x, y)
    return nlZ, time.time() - t
print("slot0", [fit() for _ in range(3)])
for k in (1, 2):
    with _lib.fit_stream(k):
        try:
            print("slot%d alone" % k, [fit() for _ in range(3)])
        except Exception as e:
            print("slot%d alone FAILED" % k, e)
def work(k, out):
    with _lib.fit_stream(k):
        for i in range(int(os.environ.get("REPS", "12"))):
            try:
                out.append((k, i) + fit())
            except Exception as e:
                out.append((k, i, "FAIL", str(e)[-60:]))
out = []
ths = [threading.Thread(target=work, args=(k, out)) for k in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
for o in out: print(o)
