"""GPU: round 6.

* value-only fits (Exact.evaluate nargout 1 / 2, Core/inf.py:382-384) beyond `fused_value_max_np` -- and whenever the
  inverse rows' scratch is not to be had -- take the plain factorisation + blocked back-substitution; same numbers either way.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import golden, relerr, synth_reg
from test_gpu_core import _fit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.mark.parametrize("opts", [dict(fused_value_max_np=1024), dict(fused_inverse=0), dict(fused_value_max_np=-1), {}])
def test_value_only_fit_with_and_without_the_fused_inverse_rows(lib, opts):
    """want = 1 / 2 at N = 2048 and 4096 against G6 (the reference's numbers): the fused-inverse sweep (default below
    fused_value_max_np), the plain sweep forced by the size bound, and fused_inverse = 0."""
    from pygps_amd import _lib
    ctx = _lib.ctx()
    try:
        for k, v in opts.items():
            _lib.check(lib.pgp_set_option(ctx, k.encode(), v))
        for N in (2048, 4096):
            g = golden("G6_rbf_d16_N%d" % N)
            x, y = synth_reg(N, 16)
            for want in (2, 1):
                got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, g["mean_hyp"][0] * np.ones(N), np.ones((1, N)), want=want)
                if want == 2:
                    assert relerr(got["nlZ"], g["nlZ"]) < 1e-9
                assert relerr(got["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7
                assert relerr(got["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8
    finally:
        lib.pgp_set_option(ctx, b"fused_value_max_np", 12288)
        lib.pgp_set_option(ctx, b"fused_inverse", 1)


def test_predict_wall_time_is_device_time(lib):
    """VERDICT r5 item 4 (GP.predict, Core/gp.py:395-417): a predict call's wall time is its device time -- no hidden host cost (pageable
    staging, a scratch allocation per call): warm calls at one shape spend < 1 ms getting scratch and wall <= 1.5 x device + 5 ms;
    the library reports the split (pgp_last_timings after pgp_predict)."""
    import time
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    n, d, ns = 4096, 16, 32768
    x, y = synth_reg(n, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0))
    m.setNoise(np.log(0.1))
    m.getPosterior(x, y)
    xs = np.random.RandomState(1).randn(ns, d)
    ref = m.predict(xs)[0].copy()
    seen = []
    for it in range(4):
        t = time.perf_counter()
        ym = m.predict(xs)[0]
        wall = (time.perf_counter() - t) * 1e3
        lt = _lib.last_timings()
        assert np.array_equal(ym, ref)
        assert lt["total"] > 0 and lt["assemble"] < 1.0, lt
        seen.append((wall, lt["total"]))
    # the best of the four warm calls (a neighbour's burst on the host must not fail the suite; a hidden per-call host cost shows in all)
    assert min(w - 1.5 * dev for w, dev in seen) <= 5.0, seen


@pytest.mark.parametrize("N", [2048, 4608, 8192])
def test_lean_panel_solve_is_bit_identical_to_the_lds_form(lib, N):
    """trsm_rows_lean_kernel (round 6: the panel solve of the diagonal-block chain without LDS, on < 96 registers, so that it is
    resident beside two bulk workgroups per CU) does the same products in the same order as trsm_rows_kernel: every output of a
    fit -- nlZ, alpha, dnlZ, the factor -- is bit-identical with the lean form off (0), on for the chain (1, default) and everywhere (2);
    and each reproduces G6 (Core/inf.py:353-384)."""
    from pygps_amd import _lib
    ctx = _lib.ctx()
    x, y = synth_reg(N, 16)
    hyp = np.array([np.log(4.0), 0.0])
    out = {}
    try:
        for v in (0, 1, 2):
            _lib.check(lib.pgp_set_option(ctx, b"trsm_lean", v))
            out[v] = _fit(lib, 0, hyp, 0, np.log(0.1), x, y, float(y.mean()) * np.ones(N), np.ones((1, N)))
    finally:
        lib.pgp_set_option(ctx, b"trsm_lean", 1)
    for v in (1, 2):
        assert out[v]["nlZ"] == out[0]["nlZ"] and np.array_equal(out[v]["alpha"], out[0]["alpha"])
        assert np.array_equal(out[v]["dnlZ"], out[0]["dnlZ"]) and np.array_equal(out[v]["L"], out[0]["L"])
    if N in (2048, 8192):
        g = golden("G6_rbf_d16_N%d" % N)
        assert relerr(out[1]["nlZ"], g["nlZ"]) < 1e-9 and relerr(out[1]["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8


@pytest.mark.parametrize("N", [128, 2048, 4096])
def test_ep_first_two_sweeps_back_to_back_change_nothing(lib, N):
    """EP.evaluate (Core/inf.py:731-806): the first two sweeps are unconditional (inf.py:732), so sweep 2 is queued behind sweep 1
    without a host round trip (round 6, option ep_merge12) and sweep 1's nlZ comes back with sweep 2's.  Same sweeps, same
    arithmetic: sweep count, nlZ, alpha, sW, the gradients and the factor are bit-identical with the option off, and both
    reproduce the reference's fixture (G8ii)."""
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    from conftest import synth_cls
    ctx = _lib.ctx()
    g = golden("G8ii_ep_d32_N%d" % N)
    x, y = synth_cls(N, 32)
    res = {}
    try:
        for v in (1, 0):
            _lib.check(lib.pgp_set_option(ctx, b"ep_merge12", v))
            m = pyGPs.GPC()
            m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(32.0)), 0.0))
            nlZ, dnlZ, post = m.getPosterior(x, y)
            res[v] = (nlZ, np.array(dnlZ.cov), post.alpha.copy(), post.sW.copy(), np.diag(np.asarray(post.L)).copy(), int(m.inffunc.sweeps))
    finally:
        lib.pgp_set_option(ctx, b"ep_merge12", 1)
    a, b = res[1], res[0]
    assert a[5] == b[5] and a[0] == b[0]
    if "n_sweeps" in g.files:
        assert a[5] == int(g["n_sweeps"])
    for u, v in zip(a[1:5], b[1:5]):
        assert np.array_equal(u, v)
    assert relerr(a[0], g["nlZ"]) < 1e-8 and relerr(a[2], g["alpha"]) < 1e-6 and relerr(a[4], g["L_diag"]) < 1e-7


@pytest.mark.parametrize("M,N,K,pad", [(128, 64, 160, 0), (1024, 512, 512, 0), (896, 192, 528, 136), (256, 64, 16, 0), (384, 128, 48, 8), (512, 256, 288, 0),
                                         (640, 128, 304, 24)])
def test_gemm_128x64_lds_dma_tile_against_numpy(lib, M, N, K, pad):
    """The 128 x 64 LDS-DMA tile (GemmArgs::tile 1264: the rectangle TU_r of the sweep's trailing update, Core/tools.py:31-62 is the
    factorisation it serves) on plain rectangles: C <- beta C + alpha A B' against numpy, short K (no lazy C), K not a multiple of
    the lazy-C prologue, leading dimensions beyond the operand heights; and the 128 x 128 tile on the same inputs."""
    from pygps_amd import _lib
    rng = np.random.RandomState(M + N + K)
    lda, ldb, ldc = M + pad, N + pad, M + 2 * pad
    A = np.asfortranarray(rng.randn(lda, K))
    B = np.asfortranarray(rng.randn(ldb, K))
    C0 = np.asfortranarray(rng.randn(ldc, N))
    ref = C0.copy()
    ref[:M] = 0.5 * C0[:M] - 1.0 * (A[:M] @ B[:N].T)
    outs = {}
    for tile in (1264, 128) if N % 128 == 0 else (1264,):
        Cw = C0.copy(order="F")
        ms = C.c_double()
        rc = lib.pgp_test_gemm(_lib.ctx(), tile, 0, 0, 0, 0, 0, 0, -1.0, 0.5, A.ctypes.data_as(_lib._dp), lda, B.ctypes.data_as(_lib._dp), ldb,
                               Cw.ctypes.data_as(_lib._dp), ldc, M, N, K, 0, C.byref(ms))
        assert rc == 0, _lib.strerror(rc)
        assert np.max(np.abs(Cw - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), (tile, np.max(np.abs(Cw - ref)))
        assert np.array_equal(Cw[M:], C0[M:])                       # rows beyond M untouched
        outs[tile] = Cw
    if 128 in outs:                                                 # both tile shapes fold C in at the same k-step: bit for bit
        assert np.array_equal(outs[1264], outs[128])


@pytest.mark.parametrize("N", [4096, 8192])
def test_fit_with_the_128x64_rectangle_against_G6(lib, N):
    """sched 2's rectangle TU_r as 128 x 64 LDS-DMA tiles (tur_tile = 1264; the default up to N = 5120) and as 128 x 128 tiles (tur_tile = 128): both
    reproduce the reference's G6 numbers (Exact.evaluate, Core/inf.py:353-384), and each other bit for bit (an element's C value is
    folded into its accumulator at the same k-step in both tile shapes)."""
    from pygps_amd import _lib
    g = golden("G6_rbf_d16_N%d" % N)
    x, y = synth_reg(N, 16)
    ctx = _lib.ctx()
    res = {}
    try:
        for tile in (1264, 128):
            _lib.check(lib.pgp_set_option(ctx, b"tur_tile", tile))
            got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, g["mean_hyp"][0] * np.ones(N), np.ones((1, N)), want=3)
            assert relerr(got["nlZ"], g["nlZ"]) < 1e-9
            assert relerr(got["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7
            assert relerr(got["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8
            assert relerr(got["dnlZ"], np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-7
            res[tile] = got
    finally:
        lib.pgp_set_option(ctx, b"tur_tile", 0)
    assert res[1264]["nlZ"] == res[128]["nlZ"]
    for k in ("alpha", "dnlZ", "L"):
        assert np.array_equal(res[1264][k], res[128][k]), k


def test_predict_product_form_equals_the_blocked_solve(lib):
    """GP.predict (Core/gp.py:395-417): V = L^-1 (sW o Ks) as one MFMA product with the cached W = L^-1 (predict_inverse 2; the default for
    batches of >= 1024 points) against the blocked triangular solve (predict_inverse 0) and the oracle -- GPR at a ragged size, test
    points that include near-duplicates of training points (fs2 -> small), a count that is not a multiple of anything; and GPC + EP
    (per-point sW)."""
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    from oracle import gp_oracle as O
    ctx = _lib.ctx()
    n, d, ns = 3000, 7, 2501
    x, y = synth_reg(n, d, seed=11)
    rng = np.random.RandomState(5)
    xs = rng.randn(ns, d)
    xs[:300] = x[rng.randint(0, n, 300)] + 1e-3 * rng.randn(300, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.2))
    m.setNoise(np.log(0.05))
    m.setData(x, y)
    out = {}
    # keep_inverse 1 (default): W = L^-1 is the transpose of the fit's own fused inverse rows; 0: a trtri at the first predict
    for keep in (1, 0):
        try:
            _lib.check(lib.pgp_set_option(ctx, b"keep_inverse", keep))
            m.getPosterior()
            for mode in (0, 2, 1):
                _lib.check(lib.pgp_set_option(ctx, b"predict_inverse", mode))
                out[mode] = [np.array(v) for v in m.predict(xs)[:4]]
                out[(mode, "small")] = [np.array(v) for v in m.predict(xs[:37])[:4]]
        finally:
            lib.pgp_set_option(ctx, b"predict_inverse", 1)
            lib.pgp_set_option(ctx, b"keep_inverse", 1)
        for a, b in ((0, 2), (0, 1), ((0, "small"), (2, "small"))):
            for u, v, tol in zip(out[a], out[b], (1e-11, 1e-9, 1e-11, 1e-9)):
                assert np.max(np.abs(u - v)) <= tol * max(1.0, float(np.max(np.abs(u)))), (keep, a, b, np.max(np.abs(u - v)))
        assert np.array_equal(out[(1, "small")][2], out[(2, "small")][2])   # W exists by then: the default takes the product form for 37 points too
    # several device batches (predict_batch 1024: 2501 points = 1024 + 1024 + 453) give what one batch gives, bit for bit, in both forms
    try:
        _lib.check(lib.pgp_set_option(ctx, b"predict_batch", 1024))
        for mode in (2, 0):
            _lib.check(lib.pgp_set_option(ctx, b"predict_inverse", mode))
            got = [np.array(v) for v in m.predict(xs)[:4]]
            for u, v in zip(got, out[mode]):
                assert np.array_equal(u, v), mode
    finally:
        lib.pgp_set_option(ctx, b"predict_batch", 65536)
        lib.pgp_set_option(ctx, b"predict_inverse", 1)
    c = m.meanfunc.hyp[0]
    ref = O.exact_fit(O.RBF, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, y, c * np.ones((n, 1)), np.ones((n, 1)), nargout=2, faithful=False)
    rym, rys2, rfm, rfs2 = O.predict(O.RBF, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, ref["alpha"], ref["L"], ref["sW"], xs,
                                     c * np.ones((ns, 1)), faithful=False)
    assert np.max(np.abs(out[2][2] - rfm)) < 1e-8 * float(np.max(np.abs(rfm))) and np.max(np.abs(out[2][3] - rfs2)) < 1e-7 * float(np.max(rfs2))
    # GPC + EP: per-point sW scales the cross-covariances
    nc, dc = 1500, 5
    rng = np.random.RandomState(3)
    xc = rng.randn(nc, dc); wc = rng.randn(dc, 1)
    yc = np.sign(xc @ wc / np.sqrt(dc) + 0.3 * rng.randn(nc, 1)); yc[yc == 0] = 1
    mc = pyGPs.GPC()
    mc.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(dc)), 0.3))
    mc.getPosterior(xc, yc)
    xcs = rng.randn(1100, dc)
    outc = {}
    try:
        for mode in (0, 2):
            _lib.check(lib.pgp_set_option(ctx, b"predict_inverse", mode))
            outc[mode] = [np.array(v) for v in mc.predict(xcs, ys=np.ones((1100, 1)))]
    finally:
        lib.pgp_set_option(ctx, b"predict_inverse", 1)
    for u, v in zip(outc[0], outc[2]):
        assert np.max(np.abs(u - v)) <= 1e-9 * max(1.0, float(np.max(np.abs(u))))


@pytest.mark.parametrize("M,N,K", [(1024, 256, 1024), (896, 128, 896), (128, 128, 128), (1152, 384, 1152)])
def test_gemm_fold_rows_kernel_against_numpy(lib, M, N, K, monkeypatch):
    """gemm_f64_fold_kernel (GemmArgs::fold_rows: tile rows mt-1-r and r in one workgroup -- GP.predict's V = L^-1 Ks, Core/gp.py:395-417)
    on a product clipped to a lower-triangular A (KM_LT_I: k < i0 + 128): against numpy with the same clipping, an odd number of tile
    rows (the middle row is its own pair), a single tile; and bit for bit against the plain kernel on the same arguments."""
    from pygps_amd import _lib
    rng = np.random.RandomState(M + N)
    A = np.asfortranarray(np.tril(rng.randn(M, K)))
    B = np.asfortranarray(rng.randn(N, K))
    C0 = np.asfortranarray(rng.randn(M, N))
    ref = np.empty_like(C0)
    for i0 in range(0, M, 128):
        k1 = min(K, i0 + 128)
        ref[i0:i0 + 128] = A[i0:i0 + 128, :k1] @ B[:, :k1].T
    outs = []
    for fold in (True, False):
        if fold:
            monkeypatch.setenv("PGP_TEST_GEMM_FOLD", "1")
        else:
            monkeypatch.delenv("PGP_TEST_GEMM_FOLD", raising=False)
        Cw = C0.copy(order="F")
        ms = C.c_double()
        rc = lib.pgp_test_gemm(_lib.ctx(), 128, 0, 0, 0, 0, 3, 0, 1.0, 0.0, A.ctypes.data_as(_lib._dp), M, B.ctypes.data_as(_lib._dp), N,
                               Cw.ctypes.data_as(_lib._dp), M, M, N, K, 0, C.byref(ms))
        assert rc == 0, _lib.strerror(rc)
        assert np.max(np.abs(Cw - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), (fold, np.max(np.abs(Cw - ref)))
        outs.append(Cw)
    assert np.array_equal(outs[0], outs[1])
