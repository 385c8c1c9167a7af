"""EP soak: K cold-start EP fits (cfg 5 shape) in one process; time of every 50th fit, free HBM along the way, nlZ must not move."""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pygps_amd as pyGPs
from pygps_amd import _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n5, d5 = 4096, 32
rng = np.random.RandomState(0)
x5 = rng.randn(n5, d5); w5 = rng.randn(d5, 1)
y5 = np.sign(x5 @ w5 / np.sqrt(d5) + 0.3 * rng.randn(n5, 1)); y5[y5 == 0] = 1
hip = C.CDLL(_lib.hip_runtime_path()) if hasattr(_lib, "hip_runtime_path") and _lib.hip_runtime_path() else None
ref = None
t0 = time.perf_counter()
for it in range(K):
    m5 = pyGPs.GPC(); m5.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d5)), 0.0))
    t = time.perf_counter(); nlz = float(m5.getPosterior(x5, y5)[0]); dt = time.perf_counter() - t
    if ref is None: ref = nlz
    assert nlz == ref, (it, nlz, ref)
    if it % 50 == 0:
        free = C.c_size_t(); tot = C.c_size_t()
        if hip is not None: hip.hipMemGetInfo(C.byref(free), C.byref(tot))
        print("fit %4d: %.1f ms, nlZ %.10f, free %.2f GiB" % (it, dt * 1e3, nlz, free.value / 2.0 ** 30), flush=True)
print("%d fits in %.1f s = %.1f ms each, bit-identical nlZ" % (K, time.perf_counter() - t0, (time.perf_counter() - t0) / K * 1e3))
