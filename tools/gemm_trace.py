"""Where the time of a K = 512 trailing-update launch goes, workgroup by workgroup (GemmArgs::trace, 100 MHz stamps).

    python tools/gemm_trace.py [M=8192] [K=512] [tri=0] [conc=0] [warm-up launches=5]

Per workgroup: entry -> k-loop end (prologue + k-loop; the lazy-C prologue is inside the first k-steps), epilogue issue, store drain;
per CU slot: the gap between one workgroup's end and the next one's entry (dispatch), and how the two workgroups of a CU overlap."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygps_amd import _lib

lib = _lib.load()
ctx = _lib.ctx()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tri = int(sys.argv[3]) if len(sys.argv) > 3 else 0
conc = int(sys.argv[4]) if len(sys.argv) > 4 else 0
warm = int(sys.argv[5]) if len(sys.argv) > 5 else 5
mt = M // 128
nmax = mt * mt
buf = (C.c_longlong * (8 * nmax))()
nblk = C.c_int64()
rc = lib.pgp_test_gemm_trace(ctx, M, K, tri, warm, conc, buf, 8 * nmax, C.byref(nblk))
assert rc == 0, rc
n = nblk.value
t = np.frombuffer(buf, dtype=np.int64)[: 8 * n].reshape(n, 8).astype(np.float64)
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0                      # 100 MHz -> us
cu = t[:, 5].astype(np.int64)
total = us[:, 4].max()
print("M=%d K=%d tri=%d conc=%d warm=%d: %d workgroups, launch %.1f us (first entry -> last store acknowledged), %.1f TF" % (
    M, K, tri, conc, warm, n, total, 2.0 * 128 * 128 * K * n / total / 1e6))
loop = us[:, 2] - us[:, 0]
epi = us[:, 3] - us[:, 2]
drain = us[:, 4] - us[:, 3]
life = us[:, 4] - us[:, 0]
def q(v):
    return "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
mhz = (t[:, 7] - t[:, 6]) / (us[:, 4] - us[:, 0])
print("shader clock while the workgroups ran (s_memtime / wall clock): median %.0f MHz (p10 %.0f, p90 %.0f)" % (
    np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90)))
cyc = t[:, 7] - t[:, 6]
print("workgroup life in shader cycles: median %.0f = %.2f x the %d cycles of its MFMAs alone (two workgroups share a CU: 2 x)" % (
    np.median(cyc), np.median(cyc) / (K / 4 * 64 * 64 / 4 * 4 / 4), K / 4 * 64 * 64 / 4))
print("entry -> k-loop end   :", q(loop))
print("epilogue (issue)      :", q(epi))
print("stores acknowledged   :", q(drain))
print("workgroup life        :", q(life))
print("pure MFMA time of a tile at 2.4 GHz: %.1f us (alone on its CU), %.1f us when two share the pipe" % (
    K / 4 * 64 * 64 / 4 / 2400.0, 2 * K / 4 * 64 * 64 / 4 / 2400.0))
# per CU: sort its workgroups by entry, pair them into two slots greedily
gaps, busy2, busy1, busy0 = [], 0.0, 0.0, 0.0
ncu = 0
for key in np.unique(cu):
    idx = np.where(cu == key)[0]
    ncu += 1
    ev = sorted([(us[i, 0], +1) for i in idx] + [(us[i, 2], -1) for i in idx])      # in "prologue + k-loop" = may issue MFMAs
    cur, last = 0, 0.0
    for (tt, d) in ev:
        if cur >= 2: busy2 += tt - last
        elif cur == 1: busy1 += tt - last
        else: busy0 += tt - last
        cur += d; last = tt
    busy0 += total - last
    ends = sorted(us[i, 4] for i in idx)
    starts = sorted(us[i, 0] for i in idx)
    # dispatch gap: for every start after the first two, the time since the most recent earlier end
    for s in starts[2:]:
        prev = [e for e in ends if e <= s + 0.005]
        if prev: gaps.append(s - max(prev))
print("CUs seen: %d, workgroups per CU: %.2f" % (ncu, n / ncu))
tot = ncu * total
print("per-CU time with 2 / 1 / 0 workgroups inside [entry, k-loop end): %.1f %% / %.1f %% / %.1f %%" % (
    100 * busy2 / tot, 100 * busy1 / tot, 100 * busy0 / tot))
if gaps:
    print("slot hand-over (previous workgroup's stores acknowledged -> next workgroup's entry):", q(np.array(gaps)))
first = np.sort(us[:, 0])
print("entries: first 512 within %.1f us; last entry at %.1f us; first k-loop end at %.1f us" % (first[min(511, n - 1)], first[-1], us[:, 2].min()))
