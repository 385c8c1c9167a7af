"""What the bulk GEMM workgroups of a loop of FITS do, from their own stamps (option gemm_trace: every bulk 128-tile launch records,
per workgroup, the 100 MHz wall clock at entry / k-loop end / stores issued / stores acknowledged, the CU, and s_memtime at both ends).

    python tools/fit_clock.py [streams=1] [fits per stream=12]

Prints the shader clock the workgroups ran at, their life in cycles against the cycles of their MFMAs, and -- over the steady part of
the loop -- how much of the CUs' time had 2, 1 or 0 bulk workgroups resident (the chip holds 2 per CU)."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N, d = 8192, 16
rng = np.random.RandomState(0)
x = rng.randn(N, d); w = rng.randn(d, 1)
y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()


def worker(ctx, steps, k):
    hyp = np.array([np.log(np.sqrt(d)), 0.0]); m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); g = np.zeros(4)
    for s in range(steps):
        hyp[0] = np.log(np.sqrt(d)) + 1e-4 * (s + k)
        assert lib.pgp_exact_fit(ctx, 0, _lib.ptr(hyp), 2, 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1, 3,
                                 _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None) == 0


def run(st):
    ths = [threading.Thread(target=worker, args=(ctxs[k], st, k)) for k in range(S)]
    t = time.time()
    [th.start() for th in ths]
    [th.join() for th in ths]
    return time.time() - t


ctxs = []
for k in range(S):
    h = C.c_void_p()
    assert lib.pgp_init(0, C.byref(h)) == 0
    assert lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(y)) == 0
    ctxs.append(h)
run(3)
dt = run(steps)
print("streams %d, untraced: %.2f ms per fit (%.1f fits/s)" % (S, dt * 1e3 / (S * steps), S * steps / dt))
CAP = 40 * steps + 64                      # thousands of workgroups per context: ~31 k per fit
for h in ctxs:
    assert lib.pgp_set_option(h, b"gemm_trace", CAP) == 0
dt = run(steps)
print("streams %d, traced  : %.2f ms per fit (%.1f fits/s)" % (S, dt * 1e3 / (S * steps), S * steps / dt))
rec = []
buf = (C.c_longlong * (8 * CAP * 1024))()
for k, h in enumerate(ctxs):
    n = C.c_int64()
    assert lib.pgp_test_read_gemm_trace(h, buf, 8 * CAP * 1024, C.byref(n)) == 0
    t = np.frombuffer(buf, dtype=np.int64)[: 8 * n.value].reshape(n.value, 8).astype(np.float64)
    rec.append(t[t[:, 4] > 0].copy())
    lib.pgp_set_option(h, b"gemm_trace", 0)
t = np.concatenate(rec)
t0 = t[:, 0].min()
ent, kend, end = (t[:, 0] - t0) / 100.0, (t[:, 2] - t0) / 100.0, (t[:, 4] - t0) / 100.0
cyc = t[:, 7] - t[:, 6]
mhz = cyc / (end - ent)
print("%d bulk workgroups recorded over %.1f ms; shader clock median %.0f MHz (p10 %.0f, p90 %.0f)" % (
    len(t), end.max() / 1e3, np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90)))
print("workgroup life: median %.1f us = %.0f cycles = %.2f x the 131072 cycles of a K = 512 tile's MFMAs (2.0 = two workgroups sharing a CU at the pipe's rate)"
      % (np.median(end - ent), np.median(cyc), np.median(cyc) / 131072.0))
# steady window: the middle 60 % of the recorded time
lo, hi = 0.2 * end.max(), 0.8 * end.max()
cu = t[:, 5].astype(np.int64)
b = np.zeros(4)
for key in np.unique(cu):
    idx = np.where(cu == key)[0]
    ev = sorted([(max(lo, min(hi, ent[i])), -1, +1) for i in idx] + [(max(lo, min(hi, end[i])), +1, -1) for i in idx])
    ev = [(a, c_) for (a, _, c_) in ev]           # at equal (clamped) times the entries come first
    cur, last = 0, lo
    for (tt, dlt) in ev:
        b[min(cur, 3)] += tt - last
        cur += dlt; last = tt
    b[min(cur, 3)] += hi - last
tot = b.sum()
print("steady window %.1f - %.1f ms, %d CUs: time with 0 / 1 / 2 / more bulk workgroups resident: %.1f %% / %.1f %% / %.1f %% / %.1f %%" % (
    lo / 1e3, hi / 1e3, len(np.unique(cu)), 100 * b[0] / tot, 100 * b[1] / tot, 100 * b[2] / tot, 100 * b[3] / tot))
inwin = (ent >= lo) & (end <= hi)
fl = 2.0 * 128 * 128 * 512 * inwin.sum()
print("tiles inside the window: %d -> %.1f TF of K = 512 tile work over the window (if every tile were a full K = 512 one)" % (
    inwin.sum(), fl / ((hi - lo) * 1e-6) / 1e12))
