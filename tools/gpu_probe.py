"""Ad-hoc measurements on the GPU box (not part of the test-suite)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pygps_amd import _lib

lib = _lib.load()
ctx = _lib.ctx()
print(_lib.device_info())


def mfma():
    for wps in (-1, 1, 2, 4):
        for nacc in (8, -8, -4):
            out = np.zeros(3)
            iters = 200000 // abs(nacc)
            lib.pgp_test_mfma_cycles(ctx, iters, nacc, wps, _lib.ptr(out))
            n = iters * abs(nacc)
            cyc = out[0] / n
            mhz = out[0] / (out[1] / 100e6) / 1e6
            nb = 256 * wps if wps > 0 else -wps
            tf = nb * 4 * n * 2048 / (out[2] * 1e-3) / 1e12
            print("waves/SIMD(or -blocks) %d nacc %d: %.1f shader-cycles/MFMA, shader clock %.0f MHz, kernel %.2f ms -> %.1f TF"
                  % (wps, nacc, cyc, mhz, out[2], tf))


def gemm(M, N, K, tile=128, a_kc=0, b_kc=0, tri=0, iters=5, fill=None):
    rng = np.random.RandomState(0)
    A = np.asfortranarray(rng.randn(M if not a_kc else K, K if not a_kc else M))
    B = np.asfortranarray(rng.randn(N if not b_kc else K, K if not b_kc else N))
    Cm = np.asfortranarray(rng.randn(M, N))
    if fill is not None:
        A[:] = fill; B[:] = fill; Cm[:] = fill
        print("  (operands filled with %g)" % fill, end=" ")
    ms = C.c_double()
    rc = lib.pgp_test_gemm(ctx, tile, a_kc, b_kc, tri, 1 if tri else 0, 0, 0, -1.0, 1.0, _lib.ptr(A), A.shape[0],
                           _lib.ptr(B), B.shape[0], _lib.ptr(Cm), M, M, N, K, iters, C.byref(ms))
    assert rc == 0
    fl = 2.0 * M * N * K * (0.5 if tri else 1.0)
    print("gemm M=%d N=%d K=%d tile=%d kc=(%d,%d) tri=%d: %.3f ms  %.1f TF" % (M, N, K, tile, a_kc, b_kc, tri, ms.value,
                                                                               fl / ms.value / 1e9))


def fit(N, d, kind=0, want=3, reps=3, prof=True):
    rng = np.random.RandomState(0)
    x = rng.randn(N, d); w = rng.randn(d, 1)
    y = (np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)).ravel()
    _lib.check(lib.pgp_set_data(ctx, _lib.ptr(x), N, d, _lib.ptr(y)))
    hyp = np.array([np.log(np.sqrt(d)), 0.0]) if kind != 1 else np.array([np.log(np.sqrt(d))] * d + [0.0])
    m = np.full(N, y.mean()); dm = np.ones((1, N))
    alpha = np.zeros(N); nlZ = np.zeros(1); dn = np.zeros(1 + len(hyp) + 1)
    for r in range(reps + 1):
        if r == reps and prof:
            lib.pgp_profile_reset(ctx); lib.pgp_set_profiling(ctx, 1)
        t = time.time()
        rc = lib.pgp_exact_fit(ctx, kind, _lib.ptr(hyp), len(hyp), 0, 0, float(np.log(0.1)), _lib.ptr(m), _lib.ptr(dm), 1,
                               want, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(dn), None)
        dt = time.time() - t
        assert rc == 0, rc
        print("N=%d d=%d kind=%d want=%d: wall %.2f ms nlZ=%.10g  stages(ms)=%s" % (
            N, d, kind, want, dt * 1e3, nlZ[0], {k: round(v, 3) for k, v in _lib.last_timings().items()}))
    if prof:
        lib.pgp_set_profiling(ctx, 0)
        for k, v in _lib.profile().items():
            if v["launches"]:
                print("   %-34s launches %4d  %8.3f ms  %7.1f TF  %7.1f GB/s" % (
                    k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6))


if __name__ == "__main__":
    what = sys.argv[1:] or ["mfma", "gemm", "fit"]
    if "mfma" in what:
        mfma()
    if "gemm" in what:
        gemm(8192, 8192, 512)
        gemm(8192, 8192, 512, tri=1)
        gemm(8192, 8192, 128, tri=1)
        gemm(8192, 8192, 2048, a_kc=1, b_kc=1)
        gemm(8192, 8192, 2048, a_kc=0, b_kc=1)
        gemm(8192, 384, 128, tile=64)
        gemm(8192, 384, 128, tile=128)
        gemm(4096, 4096, 4096)
    if "leaf" in what:
      for pv in (0, 1, 2):
        assert lib.pgp_set_option(ctx, b"leaf_pivot", pv) == 0
        print("leaf_pivot =", pv, "(0: LDS leaf, lane-per-row pivot blocks; 1: LDS leaf, pivot blocks on the matrix cores; 2: register-resident leaf)")
        tk = np.zeros(24)
        assert lib.pgp_test_leaf_ticks(ctx, _lib.ptr(tk)) == 0
        names = ["start", "loaded", "pivot0"] + ["b%d" % (i // 2) if i % 2 == 0 else "c%d" % (i // 2) for i in range(16)]
        base = tk[0]
        seq = [(names[i], tk[i]) for i in range(19)] + [("writeback", tk[20]), ("end", tk[19])]
        prev = base
        for nm, v in seq:
            print("  %-10s +%7.0f cycles (%6.2f us)  total %8.0f" % (nm, v - prev, (v - prev) / 2400.0, v - base))
            prev = v
      lib.pgp_set_option(ctx, b"leaf_pivot", 2)
    if "dvfs" in what:
        for v in (0, 7):
            lib.pgp_set_option(ctx, b"gemm_dbg", v)
            print("gemm_dbg =", v)
            for fill in (None, 0.0, 1.0):
                gemm(8192, 8192, 2048, a_kc=1, b_kc=1, fill=fill, iters=8)
        lib.pgp_set_option(ctx, b"gemm_dbg", 0)
    if "dbg" in what:
        for v in (0, 8, 7, 15, 0):
            lib.pgp_set_option(ctx, b"gemm_dbg", v)
            print("gemm_dbg =", v)
            gemm(8192, 8192, 512)
            gemm(8192, 8192, 2048, a_kc=1, b_kc=1)
        lib.pgp_set_option(ctx, b"gemm_dbg", 0)
    if "sweep" in what:
        for la in (1, 0):
            for q in (2, 3, 4, 6, 8):
                for stb in (128, 256, 512):
                    lib.pgp_set_option(ctx, b"lookahead", la); lib.pgp_set_option(ctx, b"nb_outer", q)
                    lib.pgp_set_option(ctx, b"small_tile_below", stb)
                    print("lookahead", la, "nb_outer", q, "small_tile_below", stb)
                    fit(8192, 16, reps=2, prof=False)
        lib.pgp_set_option(ctx, b"lookahead", 1); lib.pgp_set_option(ctx, b"nb_outer", 4)
        lib.pgp_set_option(ctx, b"small_tile_below", 256)
    if "qsweep" in what:
        for q in (2, 3, 4, 6):
            for stb in (256, 1024):
                lib.pgp_set_option(ctx, b"nb_outer", q); lib.pgp_set_option(ctx, b"small_tile_below", stb)
                print("nb_outer", q, "small_tile_below", stb)
                fit(8192, 16, reps=3, prof=False)
                fit(16384, 16, reps=2, prof=False)
        lib.pgp_set_option(ctx, b"nb_outer", 4); lib.pgp_set_option(ctx, b"small_tile_below", 256)
    if "trtri" in what:
        for v in (256, 513, 1025, 4097, 256):
            lib.pgp_set_option(ctx, b"trtri_small_tile_below", v)
            print("trtri_small_tile_below =", v)
            fit(8192, 16, reps=3, prof=False)
        lib.pgp_set_option(ctx, b"trtri_small_tile_below", 256)
    if "ab" in what:
        for v in (0, 1, 0, 1):
            lib.pgp_set_option(ctx, b"lookahead", v)
            print("lookahead =", v)
            fit(8192, 16, reps=3, prof=False)
    if "fit" in what:
        fit(8192, 16)
        fit(8192, 16, want=2, prof=False)
        fit(2048, 16, prof=False)
    if "asm" in what:
     for ntv in [int(a[3:]) for a in what if a.startswith("nt=")] or [-1]:
      lib.pgp_set_option(ctx, b"asm_nt", ntv)
      for grid in [int(a[5:]) for a in what if a.startswith("grid=")] or [4096]:
        lib.pgp_set_option(ctx, b"asm_grid", grid)
        print("asm_grid", grid, "asm_nt", ntv)
        for (kind, n, d) in ((0, 8192, 16), (0, 16384, 16), (1, 16384, 64), (2, 16384, 16), (0, 16384, 4)):
            for mode in (0, 2):
                ms = C.c_double()
                assert lib.pgp_test_assemble(ctx, kind, mode, n, d, 10, C.byref(ms)) == 0
                by = 8.0 * n * n * (1.0 if mode == 0 else 0.5) + 8.0 * n * d
                print("assemble kind=%d n=%d d=%d mode=%d: %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
                    kind, n, d, mode, ms.value, by / ms.value / 1e6, by / ms.value / 1e6 / 80.0))
    if "asm" in what or "stores" in what:
        o3 = (C.c_double * 3)()
        for n in (8192, 16384):
            assert lib.pgp_test_store_roof(ctx, n, 0, 20, o3) == 0
            print("stores alone n=%d: 'train' tile pattern %.3f ms (%.1f%% of 8 TB/s), hipMemsetAsync %.3f ms (%.1f%%), linear fill %.3f ms (%.1f%%)" % (
                n, o3[0], 8.0 * n * n / o3[0] / 1e6 / 80.0, o3[1], 8.0 * n * n / o3[1] / 1e6 / 80.0, o3[2], 8.0 * n * n / o3[2] / 1e6 / 80.0))
    if "ep" in what:
        import pygps_amd as pyGPs
        for N in (1024, 4096):
            rng = np.random.RandomState(0)
            d = 32
            x = rng.randn(N, d); w = rng.randn(d, 1)
            y = np.sign(x @ w / np.sqrt(d) + 0.3 * rng.randn(N, 1)); y[y == 0] = 1
            for blk, gr in ((0, 0), (1, 0)):
                lib.pgp_set_option(ctx, b"ep_block", blk)
                m = pyGPs.GPC()
                m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0))
                t = time.time()
                nlZ, dnlZ, post = m.getPosterior(x, y)
                print("EP N=%d d=%d ep_block=%d (0 = per-site reference form, 1 = block sweep) %d: %.3f s, %d sweeps, nlZ=%.12g dnlZ.cov=%s alpha[:2]=%s" % (
                    N, d, blk, gr, time.time() - t, m.inffunc.sweeps, nlZ, dnlZ.cov, post.alpha[:2, 0]))
    if "fit16k" in what:
        fit(16384, 64, kind=1, reps=1)
