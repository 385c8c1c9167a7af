"""Leak check of the drop-in path: a long loop of model.getPosterior() (posterior handles created and dropped every call,
finalizers return their device buffers to the context pools) with the free device memory sampled along the way.
usage: python tools/api_soak.py [fits] [N]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import pygps_amd as pyGPs
fits = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
x, y = bench.synth_reg(N, 16)
m = pyGPs.GPR()
m.setPrior(kernel=pyGPs.cov.RBF(np.log(4.0), 0.0)); m.setNoise(np.log(0.1))
m.setData(x, y)
free = []
t0 = time.perf_counter()
keep = []
for s in range(fits):
    hyp, log_sn = bench.hyp_for(s, 0, 16)
    m.covfunc.hyp = [float(hyp[0]), float(hyp[1])]
    m.likfunc.hyp = [log_sn]
    nlZ, dnlZ, post = m.getPosterior()
    if s % 7 == 0:
        keep.append(post)                       # some posteriors live longer than others
        if len(keep) > 5:
            keep.pop(0)
    if s % (fits // 10) == 0:
        gc.collect()
        free.append(torch.cuda.mem_get_info()[0] / 2 ** 30)
dt = time.perf_counter() - t0
print("%d fits in %.1f s = %.1f fits/s; free GiB along the way: %s" % (fits, dt, fits / dt, " ".join("%.2f" % f for f in free)))
drift = free[2] - free[-1]
print("drift after warm-up: %.3f GiB" % drift)
assert drift < 0.5, "device memory keeps shrinking"
