import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib
kv = dict(a.split("=") for a in sys.argv[1:])
N = int(kv.get("N", 4096)); tile = int(kv.get("tile", 1264)); S = int(kv.get("S", 1)); NB = int(kv.get("NB", 20))
lib = _lib.load()
rng = np.random.RandomState(0); d = 16
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
ctxs = []
for k in range(S):
    h = C.c_void_p(); assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    assert lib.pgp_set_option(h, b"tur_tile", tile) == 0
    for o in kv.get("opts", "").split(","):
        if o:
            a, b = o.split(":"); assert lib.pgp_set_option(h, a.encode(), int(b)) == 0
    ctxs.append(h)
def worker(ctx, steps, k):
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    for s in range(steps):
        hyp[0] = np.log(np.sqrt(d)) + 1e-4 * (s + k)
        rc = lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        assert rc == 0, rc
for b in range(NB):
    ths = [threading.Thread(target=worker, args=(ctxs[k], 30, k)) for k in range(S)]
    t = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
    print("N=%d tile=%d S=%d batch %2d: %.3f ms per fit" % (N, tile, S, b, (time.perf_counter() - t) / (30 * S) * 1e3), flush=True)
