"""Phase stamps of the resident diagonal-panel server over one N=8192 fit (us relative to the first stamp)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from pygps_amd import _lib
lib = _lib.load(); ctx = _lib.ctx()
for o in sys.argv[1:]:
    k, v = o.split('='); lib.pgp_set_option(ctx, k.encode(), int(v))
lib.pgp_set_option(ctx, b"ds_ticks", 1)
N, d = 8192, 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
assert lib.pgp_set_data(ctx, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
for it in range(3):
    assert lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                             _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None) == 0
npan = N // 512
t = np.zeros(16 * npan)
assert lib.pgp_test_ds_ticks(ctx, _lib.ptr(t), npan) == 0
t = t.reshape(npan, 16) / 100.0      # us
t0 = t[0, 0]
print("panel: wait_go  stage_in  leaf0 leaf1 leaf2 leaf3  stage_out+done | start(us)  total")
for p in range(npan):
    r = t[p]
    print("%3d: %8.1f %8.1f  %6.1f %6.1f %6.1f %6.1f  %8.1f | %9.1f %8.1f" % (
        p, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], r[15] - r[6], r[0] - t0, r[15] - r[1]))
print("panel: in_loops in_barrier | leaf0: leaf bar trsm bar update bar")
for p in range(npan):
    r = t[p]
    print("%3d: %8.1f %8.1f | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f" % (p, r[7] - r[1], r[2] - r[7], r[8] - r[2], r[9] - r[8], r[10] - r[9], r[11] - r[10], r[12] - r[11], r[3] - r[12]))
print(_lib.last_timings())
