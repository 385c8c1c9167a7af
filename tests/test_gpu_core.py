"""GPU: C-ABI level parity of the HIP kernels against numpy / the oracle (bit layouts, edge sizes)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, relerr, synth_reg
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _gemm(lib, tile, a_kc, b_kc, tri, mask, kmode, alpha, beta, M, N, K, seed=0):
    from pygps_amd import _lib
    rng = np.random.RandomState(seed)
    A = rng.randn(M, K)            # A(m,k)
    B = rng.randn(N, K)            # B(n,k)
    C0 = rng.randn(M, N)
    Ast = np.asfortranarray(A) if not a_kc else np.asfortranarray(A.T)     # column-major M x K or K x M
    Bst = np.asfortranarray(B) if not b_kc else np.asfortranarray(B.T)
    Cst = np.asfortranarray(C0.copy())
    lda = Ast.shape[0]
    ldb = Bst.shape[0]
    ms = C.c_double()
    rc = lib.pgp_test_gemm(_lib.ctx(), tile, a_kc, b_kc, tri, mask, kmode, 0, alpha, beta,
                           Ast.ctypes.data_as(_lib._dp), lda, Bst.ctypes.data_as(_lib._dp), ldb,
                           Cst.ctypes.data_as(_lib._dp), M, M, N, K, 0, C.byref(ms))
    assert rc == 0, _lib.strerror(rc)
    # numpy reference with the same k-range clipping per tile
    ref = C0.copy()
    T = tile
    for i0 in range(0, M, T):
        for j0 in range(0, N, T):
            if tri and i0 < j0:
                continue
            k0, k1 = 0, K
            if kmode == 1:
                k0 = i0
            elif kmode == 2:
                k0 = j0
            elif kmode == 3:
                k1 = min(K, i0 + T)
            blk = alpha * (A[i0:i0 + T, k0:k1] @ B[j0:j0 + T, k0:k1].T) + beta * C0[i0:i0 + T, j0:j0 + T]
            if tri and mask and i0 == j0:
                keep = np.tril(np.ones((T, T), dtype=bool))
                blk = np.where(keep, blk, C0[i0:i0 + T, j0:j0 + T])
            ref[i0:i0 + T, j0:j0 + T] = blk
    return np.array(Cst), ref


@pytest.mark.parametrize("tile", [128, 64])
@pytest.mark.parametrize("a_kc,b_kc", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts_asymmetric(lib, tile, a_kc, b_kc):
    got, ref = _gemm(lib, tile, a_kc, b_kc, 0, 0, 0, 1.0, 0.0, 256, 384, 160)
    assert relerr(got, ref) < 1e-13
    got, ref = _gemm(lib, tile, a_kc, b_kc, 0, 0, 0, -1.0, 1.0, 384, 256, 48, seed=1)
    assert relerr(got, ref) < 1e-13


@pytest.mark.parametrize("tile", [128, 64])
def test_gemm_tri_and_kmodes(lib, tile):
    for tri in (1, 2):
        got, ref = _gemm(lib, tile, 0, 0, tri, 1, 0, -1.0, 1.0, 512, 512, 128, seed=2)
        assert relerr(got, ref) < 1e-13
    got, ref = _gemm(lib, tile, 1, 1, 2, 1, 1, 1.0, 0.0, 512, 512, 512, seed=3)     # lauum shape
    assert relerr(got, ref) < 1e-13
    got, ref = _gemm(lib, tile, 0, 1, 0, 0, 2, 1.0, 0.0, 256, 256, 256, seed=4)     # T = L21 W11
    assert relerr(got, ref) < 1e-13
    got, ref = _gemm(lib, tile, 0, 1, 0, 0, 3, -1.0, 0.0, 256, 256, 256, seed=5)    # W21 = -W22 T
    assert relerr(got, ref) < 1e-13


@pytest.mark.parametrize("n", [1, 20, 128, 129, 300, 1024])
def test_potrf_helper_matches_lapack(lib, n):
    from pygps_amd import _lib
    rng = np.random.RandomState(n)
    G = rng.randn(n, n)
    A = G @ G.T / n + np.eye(n)
    L = np.zeros((n, n))
    rc = lib.pgp_potrf(_lib.ctx(), _lib.ptr(A), n, _lib.ptr(L))
    assert rc == 0
    ref = np.linalg.cholesky(A)
    assert relerr(L, ref) < 1e-12
    assert np.all(np.triu(L, 1) == 0)


def test_potrf_reports_first_bad_pivot(lib):
    from pygps_amd import _lib
    n = 200
    A = np.eye(n)
    A[150, 150] = -1.0
    L = np.zeros((n, n))
    rc = lib.pgp_potrf(_lib.ctx(), _lib.ptr(A), n, _lib.ptr(L))
    assert rc == 151


@pytest.mark.parametrize("leaf_pivot", [0, 1, 2])
def test_leaf_kernels_agree_with_lapack_and_on_the_first_bad_pivot(lib, leaf_pivot):
    """The three leaf kernels (Core/tools.py:31-77 jitchol's dpotrf on a 128 x 128 diagonal block): 0 = LDS leaf with lane-per-row
    pivot blocks (rounds 2-4), 1 = LDS leaf with the pivot blocks on the matrix cores, 2 = the register-resident leaf (default).
    Badly scaled and correlated blocks against LAPACK; the FIRST non-positive pivot wherever it sits inside a 16-column pivot
    block, a 4-column panel or a leaf; the triangular part above the diagonal stays exactly zero."""
    from pygps_amd import _lib
    ctx = _lib.ctx()
    _lib.check(lib.pgp_set_option(ctx, b"leaf_pivot", leaf_pivot))
    try:
        # (2304 / 4096: the look-ahead sweep WITHOUT inverse rows -- jitchol's shape -- under the default schedule: sched 2's two
        #  pieces of TU_a with a single row piece, a last panel of two leaves, no rows below the last panel)
        for n, seed, scale in ((128, 1, 1.0), (257, 2, 1e6), (640, 3, 1e-6), (1024, 4, 1.0), (2304, 5, 1.0), (4096, 6, 1.0)):
            rng = np.random.RandomState(seed)
            t = np.sort(rng.rand(n))[:, None]
            A = scale * (np.exp(-0.5 * (t - t.T) ** 2 / 0.05 ** 2) * 50.0 + np.eye(n))      # strongly correlated neighbours
            L = np.zeros((n, n))
            assert lib.pgp_potrf(ctx, _lib.ptr(A), n, _lib.ptr(L)) == 0
            ref = np.linalg.cholesky(A)
            assert relerr(L, ref) < 1e-11, (n, relerr(L, ref))
            assert np.abs(L @ L.T - A).max() / np.abs(A).max() < 1e-14
            assert np.all(np.triu(L, 1) == 0)
        for bad in (0, 1, 3, 4, 15, 16, 17, 127, 128, 129, 150, 255, 299):
            n = 300
            rng = np.random.RandomState(bad)
            G = rng.randn(n, n)
            A = G @ G.T / n + np.eye(n)
            A[bad, bad] = -1.0 if bad % 2 else 0.0                                        # zero counts as non-positive, too
            if bad % 2 == 0:
                A[bad, :] = 0.0; A[:, bad] = 0.0
            L = np.zeros((n, n))
            assert lib.pgp_potrf(ctx, _lib.ptr(A), n, _lib.ptr(L)) == bad + 1, bad
    finally:
        lib.pgp_set_option(ctx, b"leaf_pivot", 2)


def _fit(lib, kind, cov_hyp, para, log_sn, x, y, m, dm, want=3, flags=0, factor=True):
    from pygps_amd import _lib
    n, d = x.shape
    ctx = _lib.ctx()
    x = _lib.f64(x); y = _lib.f64(y).ravel()
    _lib.check(lib.pgp_set_data(ctx, _lib.ptr(x), n, d, _lib.ptr(y)))
    cov_hyp = _lib.f64(cov_hyp)
    nmean = 0 if dm is None else dm.shape[0]
    alpha = np.zeros(n); nlZ = np.zeros(1); dn = np.zeros(nmean + len(cov_hyp) + 1)
    fh = C.c_void_p()
    mv = _lib.f64(m).ravel()
    dmv = None if dm is None else _lib.f64(dm)
    rc = lib.pgp_exact_fit(ctx, kind, _lib.ptr(cov_hyp), len(cov_hyp), para, flags, float(log_sn), _lib.ptr(mv),
                           _lib.ptr(dmv), nmean, want, _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(dn),
                           C.byref(fh) if factor else None)
    _lib.check(rc)
    out = dict(alpha=alpha.reshape(n, 1), nlZ=float(nlZ[0]), dnlZ=dn)
    if factor:
        L = np.zeros((n, n))
        _lib.check(lib.pgp_factor_to_host(ctx, fh, _lib.ptr(L)))
        lib.pgp_factor_free(ctx, fh)
        out["L"] = L
    return out


@pytest.mark.parametrize("N,d", [(20, 3), (129, 2), (700, 5), (2048, 16)])
@pytest.mark.parametrize("want", [3, 2])
def test_exact_fit_rbf_vs_oracle(lib, N, d, want):
    x, y = synth_reg(N, d, seed=N)
    c = float(y.mean())
    hyp = np.array([np.log(np.sqrt(d)), 0.1])
    ref = O.exact_fit(O.RBF, hyp, 0, np.log(0.1), x, y, c * np.ones_like(y), np.ones_like(y), nargout=want,
                      faithful=False)
    got = _fit(lib, 0, hyp, 0, np.log(0.1), x, y, c * np.ones(N), np.ones((1, N)), want=want)
    assert relerr(got["nlZ"], ref["nlZ"]) < 1e-10                      # north_star: 1e-8
    assert relerr(got["alpha"], ref["alpha"]) < 1e-8                   # north_star: 1e-6
    assert relerr(got["L"], ref["L"]) < 1e-10                          # north_star: 1e-6
    assert np.all(np.tril(got["L"], -1) == 0)
    if want == 3:
        rg = np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]])
        assert relerr(got["dnlZ"], rg) < 1e-8


def test_exact_fit_golden_G1_G2_G3(lib):
    g = golden("G1_regression_default")
    n = g["x"].shape[0]
    got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], g["x"], g["y"], g["mean_hyp"][0] * np.ones(n), np.ones((1, n)))
    assert relerr(got["nlZ"], g["nlZ"]) < 1e-10 and relerr(got["alpha"], g["alpha"]) < 1e-8
    assert relerr(got["L"], g["L"]) < 1e-10
    assert relerr(got["dnlZ"], np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-8
    g = golden("G2_seed0_rbf_zero_mean")
    got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], g["x"], g["y"], np.zeros(20), None)
    assert relerr(got["nlZ"], g["nlZ"]) < 1e-10 and relerr(got["dnlZ"], np.concatenate([g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-8
    g = golden("G3_seed0_rbfard_zero_mean")
    got = _fit(lib, 1, g["cov_hyp"], 0, g["lik_hyp"][0], g["x"], g["y"], np.zeros(20), None)
    assert relerr(got["nlZ"], g["nlZ"]) < 1e-10 and relerr(got["dnlZ"], np.concatenate([g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-8
    assert relerr(got["L"], g["L"]) < 1e-10


def test_exact_fit_golden_G6_cfg2_scale(lib):
    for N in (2048, 8192):
        g = golden("G6_rbf_d16_N%d" % N)
        x, y = synth_reg(N, 16)
        got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, g["mean_hyp"][0] * np.ones(N), np.ones((1, N)))
        assert relerr(got["nlZ"], g["nlZ"]) < 1e-9                     # north_star: 1e-8
        assert relerr(got["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7
        assert relerr(np.diag(got["L"]), g["L_diag"]) < 1e-9
        assert relerr(got["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8
        assert relerr(got["dnlZ"], np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-7


def test_probit_hazard_of_the_site_update(lib):
    """N(z) / Phi(z) of the latency-trimmed EP site update (erfcx polynomial in Estrin form, csrc/erfcx_poly.h) against
    50-digit mpmath over the whole branch z > -5 of Core/lik.py:340-343 -- including z << 0, where the reference's own
    0.5 (1 + erf) loses digits."""
    mp = pytest.importorskip("mpmath")
    from pygps_amd import _lib
    mp.mp.dps = 40
    ctx = _lib.ctx()
    rng = np.random.RandomState(2)
    z = np.concatenate([np.linspace(-4.999, 9.0, 3001), rng.uniform(-4.999, 0.0, 2000), [0.0, -0.0, 1e-300, 8.4852, 8.49, 30.0]])
    out = np.empty_like(z)
    _lib.check(lib.pgp_test_probit_hazard(ctx, _lib.ptr(z), _lib.ptr(out), len(z)))
    worst, zw = 0.0, None
    for zi, oi in zip(z, out):
        ex = mp.npdf(zi) / mp.ncdf(zi)
        err = float(abs((mp.mpf(float(oi)) - ex) / ex))
        if err > worst:
            worst, zw = err, float(zi)
    assert worst < 2e-13, (worst, zw)          # v_rcp_f64 + one Newton step (fast_rcp) is good to ~6e-14


def test_shrinking_batched_trailing_update(lib):
    """GemmArgs::batch_dm: ONE launch updates several column panels whose row counts shrink with the batch index (the owned
    panels of the block-cyclic sweep, csrc/sharded.hip), each with its own first-touch row, against numpy."""
    from pygps_amd import _lib
    ctx = _lib.ctx()
    rng = np.random.RandomState(5)
    M, K, w, nb, dm = 2560, 512, 512, 3, 1024            # panels at rows 0, 1024, 2048 of the broadcast panel Y
    zf = M - 384                                           # rows >= zf of product 0 are first-touch (garbage on input)
    Y = np.asfortranarray(rng.randn(M, K))
    ldc, sC = M, M * w
    C = rng.randn(nb * sC)
    want = C.copy()
    for z in range(nb):
        Mz = M - z * dm
        Cz = want[z * sC:z * sC + ldc * w].reshape(w, ldc).T[:Mz]          # column-major (ldc x w) view, first Mz rows
        old = Cz.copy()
        old[zf - z * dm:] = 0.0
        upd = old - Y[z * dm:] @ Y[z * dm:z * dm + w].T
        lower = np.tril(np.ones((Mz, w), dtype=bool))
        Cz[lower] = upd[lower]
    got = C.copy()
    _lib.check(lib.pgp_test_gemm_shrink(ctx, _lib.ptr(Y), M, M, K, w, nb, dm, zf, _lib.ptr(got), ldc, sC))
    for z in range(nb):
        Mz = M - z * dm
        g = got[z * sC:z * sC + ldc * w].reshape(w, ldc).T
        wv = want[z * sC:z * sC + ldc * w].reshape(w, ldc).T
        lower = np.tril(np.ones((Mz, w), dtype=bool))
        assert np.abs(g[:Mz][lower] - wv[:Mz][lower]).max() < 1e-11, z
        strict_upper = np.triu(np.ones((w, w), dtype=bool), 129)           # beyond the masked diagonal tiles: untouched
        assert np.array_equal(g[:w][strict_upper], wv[:w][strict_upper])
        assert np.array_equal(g[Mz:], wv[Mz:])                             # rows beyond the product: untouched


@pytest.mark.parametrize("opts", [{'yield': 0}, dict(leaf_first=1), {'leaf_first': 3, 'yield': 0}, dict(lookahead=0), dict(eet_overlap=0), dict(eet_overlap=2, eet_tile=64),
                                  dict(s_tile=128), dict(eet_first=0), dict(small_tile_below=256), dict(gemm_dbg=0),
                                  dict(xcd_order=1), dict(xcd_order=1, xcd_super=4, xcd_min_tiles=64),
                                  dict(pair_launch=0), dict(pair_launch=1, gemm_trace=64), dict(sched=1), dict(sched=0),
                                  dict(sched=1, pair_launch=0, eet_overlap=2), dict(sched=1, eet_overlap=0),
                                  dict(sched=2), dict(sched=2, tud_tile=128, tud_mark=1), dict(sched=2, nb_outer=6, leaf_pivot=0),
                                  dict(sched=2, pair_launch=0, eet_overlap=0, nb_outer=8), dict(leaf_pivot=0), dict(leaf_pivot=1),
                                  dict(s_pan=0), dict(s_pan=1), dict(s_pan=2), dict(sched=2, s_pan=1, nb_outer=6), dict(s_pan=1, pair_launch=0, eet_overlap=0), dict(s_pan=1, s_pan_direct=0), dict(s_pan=1, s_pan_out=0), dict(s_pan=2, s_pan_out=1, pair_launch=0)])
def test_cholesky_sweep_variants_agree_with_the_reference(lib, opts):
    """The kept schedule options of the Cholesky sweep -- without the cooperative yield of the bulk workgroups, the trailing
    update held back until D(p+1)'s stage-in / third chain kernel, the serial order, B^-1 = E E^T as one product after the sweep or as panel products behind every trailing update, tile
    choices, the plain (register-staged) GEMM form, the XCD-aware tile order, TU_b(p) and panel p's share of E E' as TWO launches
    instead of one (gemm_f64_pair_kernel is the default), the workgroups' phase stamps switched on -- against the reference's own numbers
    (G6: Core/inf.py:353-384 at N=2048 and at the benchmark size N=8192).  The variants that were measured and lost in
    rounds 1-2 (resident server, depth-2 look-ahead, side streams, CU reservation, merged grids, the round-1 sweep) are
    gone from the library."""
    from pygps_amd import _lib
    ctx = _lib.ctx()
    try:
        for k, v in opts.items():
            _lib.check(lib.pgp_set_option(ctx, k.encode(), v))
        for N in (2048, 8192):
            g = golden("G6_rbf_d16_N%d" % N)
            x, y = synth_reg(N, 16)
            for rep in range(2):
                got = _fit(lib, 0, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, g["mean_hyp"][0] * np.ones(N), np.ones((1, N)),
                           factor=rep == 0)
                assert relerr(got["nlZ"], g["nlZ"]) < 1e-9
                assert relerr(got["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7
                assert relerr(got["dnlZ"], np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-7
                if rep == 0:
                    assert relerr(got["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8
    finally:
        for k in opts:
            lib.pgp_set_option(ctx, k.encode(), {"lookahead": 1, "leaf_first": 0, "yield": 1, "eet_overlap": 3, "eet_tile": 128, "s_tile": 0,
                                                 "eet_first": -1, "small_tile_below": 200, "xcd_order": 0, "xcd_super": 8,
                                                 "xcd_min_tiles": 256, "gemm_dbg": 64 | 256 | 512, "pair_launch": 1, "leaf_pivot": 2, "sched": -1,
                                                 "tud_tile": 64, "s_pan": -1, "s_pan_direct": 1, "s_pan_out": 1}.get(k, 0))


def test_exact_fit_golden_G7_ard_d64(lib):
    g = golden("G7_rbfard_d64_N1024")
    x, y = synth_reg(1024, 64)
    got = _fit(lib, 1, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, g["mean_hyp"][0] * np.ones(1024), np.ones((1, 1024)))
    assert relerr(got["nlZ"], g["nlZ"]) < 1e-9
    assert relerr(got["dnlZ"], np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])) < 1e-7


def test_exact_fit_matern_and_nonpd(lib):
    g = golden("G4b_matern5_N256")
    got = _fit(lib, 2, g["cov_hyp"], 5, g["lik_hyp"][0], g["x"], g["y"], g["mean_hyp"][0] * np.ones(256), None, want=2)
    assert relerr(got["nlZ"], g["nlZ"]) < 1e-10 and relerr(got["L"], g["L"]) < 1e-10
    # correct Matern gradient vs the oracle with matern_reference_compat=False, and the compat flag
    x, y = g["x"], g["y"]
    for flags, compat in ((0, False), (1, True)):
        ref = O.exact_fit(O.MATERN, g["cov_hyp"], 5, g["lik_hyp"][0], x, y, np.zeros_like(y), None, faithful=False,
                          matern_reference_compat=compat)
        got = _fit(lib, 2, g["cov_hyp"], 5, g["lik_hyp"][0], x, y, np.zeros(256), None, flags=flags)
        assert relerr(got["dnlZ"], np.concatenate([ref["dnlZ_cov"], ref["dnlZ_lik"]])) < 1e-8
    # non-PD: a huge signal / tiny noise ratio with duplicated points
    xx = np.zeros((40, 2)); yy = np.zeros((40, 1))
    with pytest.raises(np.linalg.LinAlgError):
        _fit(lib, 0, np.array([0.0, 40.0]), 0, -40.0, xx, yy, np.zeros(40), None)


def test_cov_all_modes_vs_golden(lib):
    from pygps_amd import _lib
    kinds = {"rbf": (0, 0), "rbfard": (1, 0), "matern1": (2, 1), "matern3": (2, 3), "matern5": (2, 5), "matern7": (2, 7)}
    for fix, names in (("G4_kernels_seed0", list(kinds)), ("G5_kernels_unit_test_setup", ["rbf", "rbfard", "matern3"])):
        g = golden(fix)
        x, z = _lib.f64(g["x"]), _lib.f64(g["z"])
        n, d = x.shape
        m = z.shape[0]
        for nm in names:
            kind, para = kinds[nm]
            hyp = _lib.f64(g[nm + "_hyp"])
            for der in [-1] + list(range(len(hyp))):
                key = "K" if der < 0 else "dK%d" % der
                for mode, mname, shape in ((0, "train", (n, n)), (1, "cross", (n, m)), (2, "self", (m, 1))):
                    out = np.zeros(shape)
                    rc = lib.pgp_cov(_lib.ctx(), kind, mode, der, _lib.ptr(x), n, _lib.ptr(z), m, d, _lib.ptr(hyp),
                                     len(hyp), para, _lib.FLAG_MATERN_REFERENCE_DER, _lib.ptr(out))
                    assert rc == 0
                    ref = g["%s_%s_%s" % (nm, key, mname)]
                    assert np.max(np.abs(out - ref)) <= 1e-13 * max(1.0, np.max(np.abs(ref))), (fix, nm, key, mname)


def test_mfma_peak_reported(lib):
    from pygps_amd import _lib
    tf = C.c_double()
    assert lib.pgp_test_mfma_peak(_lib.ctx(), 20000, C.byref(tf)) == 0
    print("fp64 MFMA issue-rate peak: %.1f TFLOP/s" % tf.value, _lib.device_info())
    assert tf.value > 20


def test_exp_nonpos_accuracy_over_the_whole_range(lib):
    """The device exp for non-positive arguments (Cody-Waite + degree-13 Taylor, csrc/sqdist_tile.h) against numpy over
    [-745, 0]: RBF values from 1 down to the denormals, and exact zeros beyond."""
    import pygps_amd as pyGPs
    n = 4000
    x = np.zeros((n, 1))
    x[:, 0] = np.sqrt(np.linspace(0.0, 1600.0, n))          # s = x_i^2 against the origin: 0 .. 1600
    z = np.zeros((1, 1))
    K = pyGPs.cov.RBF(0.0, 0.0).getCovMatrix(x=x, z=z, mode="cross")[:, 0]
    s = x[:, 0] ** 2
    ref = np.exp(-0.5 * s)
    normal = ref > 1e-300
    assert np.max(np.abs(K[normal] / ref[normal] - 1.0)) < 1e-15
    assert np.all(K[~normal] >= 0.0) and np.all(np.abs(K[~normal] - ref[~normal]) < 1e-300)
    assert K[0] == 1.0 and K[-1] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("tile,n,skip,wait_ms", [(128, 1024, (256, 384), 0), (64, 512, (128, 256), 0), (128, 1024, (0, 0), 0),
                                                  (128, 1024, (512, 640), 20), (64, 512, (0, 0), 20)])
def test_gemm_skip_block_and_in_kernel_wait(lib, tile, n, skip, wait_ms):
    """GemmArgs::skip_lo / skip_hi (the diagonal block another kernel owns inside a whole-matrix update) and GemmArgs::wait_flag
    (every workgroup waits for a device counter that another stream raises): the two pieces of the EP block sweep's fold / U
    launches, here on their own.  Lower tiles (masked diagonal) of C -= A B'; the skipped block must come back untouched and a
    launch that waited 20 ms for its counter must not have timed out."""
    from pygps_amd import _lib
    rng = np.random.RandomState(5)
    K = 128
    A = np.asfortranarray(rng.randn(n, K)); B = np.asfortranarray(rng.randn(n, K)); C0 = np.asfortranarray(rng.randn(n, n))
    Cst = C0.copy(order="F")
    to = C.c_int(-1)
    _lib.check(lib.pgp_test_gemm_skip_wait(_lib.ctx(), tile, _lib.ptr(A), _lib.ptr(B), _lib.ptr(Cst), n, K, skip[0], skip[1], wait_ms,
                                           C.byref(to)))
    assert to.value == 0
    ref = C0 - A @ B.T
    low = np.tril(np.ones((n, n), bool))
    touched = low.copy()
    # tiles strictly above the diagonal are not part of the launch; within diagonal tiles only i >= j is written
    if skip[1] > skip[0]:
        touched[skip[0]:skip[1], skip[0]:skip[1]] = False
    assert np.max(np.abs(Cst[touched] - ref[touched])) < 1e-10
    assert np.array_equal(Cst[~touched], C0[~touched])
