"""Soak: restart search with two fit streams on one GPU (ShardedMinimize), repeated; watches device memory."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import pygps_amd as pyGPs
N, d = 4096, 8
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
for rep in range(10):
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0)); m.setNoise(np.log(0.1)); m.setData(x, y)
    m.setOptimizer("ShardedMinimize", num_restarts=6)
    np.random.seed(100 + rep)
    t = time.perf_counter()
    m.optimize(x, y, numIterations=15)
    free, total = torch.cuda.mem_get_info()
    print("rep %d: %.2f s, nlZ %.6f, hyp %s, device memory in use %.2f GiB" % (
        rep, time.perf_counter() - t, m.nlZ, np.round(m.covfunc.hyp + m.likfunc.hyp, 4), (total - free) / 2 ** 30))
