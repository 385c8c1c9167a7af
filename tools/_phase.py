import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib
kv = dict(a.split("=") for a in sys.argv[1:])
N = int(kv.get("N", 8192)); STEPS = int(kv.get("STEPS", 100))
lib = _lib.load()
rng = np.random.RandomState(0); d = 16
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
ctxs = []
for k in range(2):
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    assert lib.pgp_set_option(h, b"concurrent_streams", int(kv.get("hint", 1))) == 0
    ctxs.append(h)
def worker(ctx, steps, k, delay):
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    if k == 1 and delay > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < delay: pass
    for s in range(steps):
        hyp[0] = np.log(np.sqrt(d)) + 1e-4 * (s + k)
        rc = lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        assert rc == 0, rc
for rnd in range(2):
    for delay_ms in (0.0, 0.15, 0.3, 0.45, 2.3, 4.7, 7.0):
        ths = [threading.Thread(target=worker, args=(ctxs[k], STEPS, k, delay_ms * 1e-3)) for k in range(2)]
        t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
        dt = time.perf_counter() - t - delay_ms * 1e-3
        print("N=%d delay %.2f ms: %.1f fits/s" % (N, delay_ms, 2 * STEPS / dt), flush=True)
