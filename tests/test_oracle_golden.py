"""CPU: pin the oracle (oracle/gp_oracle.py) to the golden vectors recorded from the reference."""
import numpy as np
import pytest

from conftest import golden, relerr, synth_reg, synth_cls, g11_trees, g14_trees, g15_trees, G11_1D
from oracle import gp_oracle as O

KINDS = {"rbf": (O.RBF, 0), "rbfard": (O.RBFARD, 0), "matern1": (O.MATERN, 1), "matern3": (O.MATERN, 3),
         "matern5": (O.MATERN, 5), "matern7": (O.MATERN, 7)}


@pytest.mark.parametrize("fix,names", [("G4_kernels_seed0", list(KINDS)),
                                       ("G5_kernels_unit_test_setup", ["rbf", "rbfard", "matern3"])])
def test_kernels_all_modes(fix, names):
    g = golden(fix)
    x, z = g["x"], g["z"]
    for nm in names:
        kind, para = KINDS[nm]
        hyp = g[nm + "_hyp"]
        for mode, kw in (("train", dict(x=x)), ("cross", dict(x=x, z=z)), ("self", dict(z=z))):
            m = "self_test" if mode == "self" else mode
            K = O.cov_matrix(kind, hyp, para, mode=m, **kw)
            assert K.shape == g["%s_K_%s" % (nm, mode)].shape
            np.testing.assert_allclose(K, g["%s_K_%s" % (nm, mode)], rtol=1e-13, atol=1e-300)
            for i in range(len(hyp)):
                dK = O.der_matrix(kind, hyp, para, mode=m, der=i, **kw)
                np.testing.assert_allclose(dK, g["%s_dK%d_%s" % (nm, i, mode)], rtol=1e-12, atol=1e-300)


def _check_fit(g, out, tol=1e-11):
    assert relerr(out["nlZ"], g["nlZ"]) < tol
    assert relerr(out["alpha"], g["alpha"]) < 1e-9
    assert relerr(out["L"], g["L"]) < 1e-11
    assert np.all(np.tril(out["L"], -1) == 0)
    assert relerr(out["sW"], g["sW"]) < 1e-14
    for k in ("dnlZ_cov", "dnlZ_lik", "dnlZ_mean"):
        if k in g.files and g[k].size:
            assert relerr(out[k], g[k]) < 1e-9, k


@pytest.mark.parametrize("faithful", [True, False])
def test_G1_regression_default(faithful):
    g = golden("G1_regression_default")
    x, y = g["x"], g["y"]
    n = x.shape[0]
    c = g["mean_hyp"][0]
    assert abs(c - y.mean()) < 1e-15                    # Q8: setData -> Const(mean(y))
    out = O.exact_fit(O.RBF, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones((n, 1)), np.ones((n, 1)),
                      faithful=faithful)
    _check_fit(g, out)
    ym, ys2, fm, fs2 = O.predict(O.RBF, g["cov_hyp"], 0, g["lik_hyp"][0], x, out["alpha"], out["L"], out["sW"],
                                 g["xstar"][:3], c * np.ones((3, 1)))
    assert relerr(ym, g["pred3_ym"]) < 1e-11 and relerr(ys2, g["pred3_ys2"]) < 1e-10


@pytest.mark.parametrize("fix,kind", [("G2_seed0_rbf_zero_mean", O.RBF), ("G3_seed0_rbfard_zero_mean", O.RBFARD)])
@pytest.mark.parametrize("faithful", [True, False])
def test_G2_G3(fix, kind, faithful):
    g = golden(fix)
    x, y = g["x"], g["y"].reshape(-1, 1)
    out = O.exact_fit(kind, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, np.zeros_like(y), None, faithful=faithful)
    _check_fit(g, out)


def test_G4b_matern5_value_path():
    g = golden("G4b_matern5_N256")
    x, y = g["x"], g["y"]
    c = g["mean_hyp"][0]
    out = O.exact_fit(O.MATERN, g["cov_hyp"], 5, g["lik_hyp"][0], x, y, c * np.ones_like(y), None, nargout=2)
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-12
    assert relerr(out["alpha"], g["alpha"]) < 1e-9 and relerr(out["L"], g["L"]) < 1e-11


def test_G6_N2048_scale_and_recipe():
    g = golden("G6_rbf_d16_N2048")
    x, y = synth_reg(2048, 16)
    c = g["mean_hyp"][0]
    assert abs(c - y.mean()) < 1e-15
    out = O.exact_fit(O.RBF, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y),
                      faithful=False)
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-11
    assert relerr(out["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-8
    assert relerr(np.diag(out["L"]), g["L_diag"]) < 1e-11
    assert relerr(out["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-10
    for k in ("dnlZ_cov", "dnlZ_lik", "dnlZ_mean"):
        assert relerr(out[k], g[k]) < 1e-8, k


def test_G7_N1024_ard_both_gradient_forms():
    g = golden("G7_rbfard_d64_N1024")
    x, y = synth_reg(1024, 64)
    c = g["mean_hyp"][0]
    for faithful in (True, False):          # False uses the W@X contraction the GPU path also uses
        out = O.exact_fit(O.RBFARD, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y),
                          faithful=faithful)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11
        assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8
        assert relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8


def test_G8_ep_demo_and_synth():
    g = golden("G8i_classification_demo_ep")
    x, y = g["x"], g["y"]
    out = O.ep_fit(O.RBF, g["cov_hyp"], 0, x, y, np.zeros_like(y, dtype=float))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-10
    assert relerr(out["alpha"], g["alpha"]) < 1e-8 and relerr(out["sW"], g["sW"]) < 1e-9
    assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8
    assert relerr(out["L"], g["L"]) < 1e-9
    ym, ys2, fm, fs2 = O.predict(O.RBF, g["cov_hyp"], 0, 0.0, x, out["alpha"], out["L"], out["sW"], g["xstar5"],
                                 np.zeros((5, 1)), gauss=False)
    assert relerr(ym, g["pred_ym"]) < 1e-9 and relerr(fs2, g["pred_fs2"]) < 1e-9
    g = golden("G8ii_ep_d32_N128")
    x, y = synth_cls(128, 32)
    out = O.ep_fit(O.RBF, g["cov_hyp"], 0, x, y, np.zeros_like(y))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-10 and relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8
    assert relerr(out["ttau"], g["ttau"]) < 1e-9


def test_matern_correct_derivative_differs_and_matches_fd():
    g = golden("G4_kernels_seed0")
    x = g["x"]
    hyp = np.array(g["matern3_hyp"])
    good = O.der_matrix(O.MATERN, hyp, 3, x=x, mode="train", der=0, matern_reference_compat=False)
    h = 1e-6
    hp, hm = hyp.copy(), hyp.copy()
    hp[0] += h; hm[0] -= h
    fd = (O.cov_matrix(O.MATERN, hp, 3, x=x, mode="train") - O.cov_matrix(O.MATERN, hm, 3, x=x, mode="train")) / (2 * h)
    assert np.max(np.abs(good - fd)) < 1e-8
    assert np.max(np.abs(g["matern3_dK0_train"] - fd)) > 1e-2      # the reference's is wrong


def test_non_pd_raises_linalgerror():
    with pytest.raises(np.linalg.LinAlgError):
        O.jitchol(np.ones((4, 4)))
    with pytest.raises(Exception, match="Wrong sizes"):
        O.solve_chol(np.eye(3), np.ones((4, 1)))


def test_G10_rbfunit_rq_piecepoly_kernels_and_fits():
    """the next stationary kernels (Core/cov.py:683-782, 832-869, 1304-1347)."""
    g = golden("G10_kernels_rbfunit_rq_piecepoly")
    x, z = g["x"], g["z"]
    kinds = {"rbfunit": (O.RBFUNIT, 0), "rq": (O.RQ, 0), "pp0": (O.PIECEPOLY, 0), "pp1": (O.PIECEPOLY, 1),
             "pp2": (O.PIECEPOLY, 2), "pp3": (O.PIECEPOLY, 3)}
    for nm, (kind, para) in kinds.items():
        hyp = g[nm + "_hyp"]
        for mode, kw in (("train", dict(x=x)), ("cross", dict(x=x, z=z)), ("self", dict(z=z))):
            m = "self_test" if mode == "self" else mode
            np.testing.assert_allclose(O.cov_matrix(kind, hyp, para, mode=m, **kw), g["%s_K_%s" % (nm, mode)], rtol=1e-13, atol=1e-300)
            for i in range(len(hyp)):
                np.testing.assert_allclose(O.der_matrix(kind, hyp, para, mode=m, der=i, **kw), g["%s_dK%d_%s" % (nm, i, mode)],
                                           rtol=1e-12, atol=1e-300)
    x, y = synth_reg(300, 4)
    for nm, kind in (("rbfunit", O.RBFUNIT), ("rq", O.RQ), ("pp2", O.PIECEPOLY)):
        g = golden("G10_fit_%s_N300" % nm)
        c = g["mean_hyp"][0]
        out = O.exact_fit(kind, g["cov_hyp"], int(g["para"][0]), g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y),
                          faithful=False)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9
        assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8


def test_G11_remaining_kernels_and_composites():
    """RQard, Gabor, Periodic, Noise, Const and Sum/Product/Scale trees (Core/cov.py:230-328)."""
    g = golden("G11_kernels_rqard_gabor_periodic_noise_const_composites")
    trees = g11_trees()
    for nm in ("rqard", "gabor", "noise", "const", "periodic", "sum", "prod", "scale", "tree", "ardsum", "maunaloa"):
        x, z = (g["x1"], g["z1"]) if nm in G11_1D else (g["x"], g["z"])
        tree = trees[nm]
        kind, para = (tree[1], tree[2]) if tree[0] == "leaf" else (tree, 0)
        hyp = g[nm + "_hyp"]
        assert O.n_cov_hyp(kind, x.shape[1]) == len(hyp)
        for mode, kw in (("train", dict(x=x)), ("cross", dict(x=x, z=z)), ("self", dict(z=z))):
            m = "self_test" if mode == "self" else mode
            np.testing.assert_allclose(O.cov_matrix(kind, hyp, para, mode=m, **kw), g["%s_K_%s" % (nm, mode)], rtol=1e-13,
                                       atol=1e-300, err_msg=nm + mode)
            for i in range(len(hyp)):
                np.testing.assert_allclose(O.der_matrix(kind, hyp, para, mode=m, der=i, **kw), g["%s_dK%d_%s" % (nm, i, mode)],
                                           rtol=1e-12, atol=1e-300, err_msg="%s d%d %s" % (nm, i, mode))


def test_G11_fits_with_composites():
    trees = g11_trees()
    g = golden("G11_fit_maunaloa_N300")
    x, y = g["x"], g["y"]
    c = g["mean_hyp"][0]
    out = O.exact_fit(trees["maunaloa"], g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False)
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-10 and relerr(out["alpha"], g["alpha"]) < 1e-8
    assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8
    x, y = synth_reg(300, 4)
    for nm, kind in (("rqard", O.RQARD), ("scaled_sum", trees["scaled_sum"])):
        g = golden("G11_fit_%s_N300" % nm)
        c = g["mean_hyp"][0]
        out = O.exact_fit(kind, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9
        assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8
    g = golden("G11_ep_composite_N200")
    out = O.ep_fit(trees["ep_composite"], g["cov_hyp"], 0, g["x"], g["y"], np.zeros_like(g["y"]))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-9 and relerr(out["alpha"], g["alpha"]) < 1e-7
    assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-7


def test_G12_fitc_sparse_regression():
    """FITC_Exact (Core/inf.py:386-455) + FITCOfKernel (Core/cov.py:332-390) + predict."""
    g = golden("G12_fitc_demo_default_u")
    x, y = g["x"], g["y"]
    c = g["mean_hyp"][0]
    out = O.fitc_fit(O.RBF, g["cov_hyp"], 0, g["lik_hyp"][0], x, g["u"], y, c * np.ones_like(y), np.ones_like(y))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9 and relerr(out["L"], g["L"]) < 1e-9
    assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8
    assert relerr(out["dnlZ_mean"], g["dnlZ_mean"]) < 1e-8
    fm, fs2 = O.fitc_predict(O.RBF, g["cov_hyp"], 0, g["u"], out["alpha"], out["L"], g["pred_xs"], c * np.ones(7))
    assert relerr(fm, g["pred_fm"]) < 1e-9 and relerr(fs2, g["pred_fs2"]) < 1e-8
    x, y = synth_reg(1500, 4)
    sum_tree = \
        ("sum", ("leaf", O.RBF, 0), ("scale", ("leaf", O.RQ, 0)))
    for nm, kind, para in (("rbf", O.RBF, 0), ("matern5", O.MATERN, 5), ("rbfard", O.RBFARD, 0), ("sum", sum_tree, 0)):
        g = golden("G12_fitc_%s_N1500_nu160" % nm)
        c = g["mean_hyp"][0]
        out = O.fitc_fit(kind, g["cov_hyp"], para, g["lik_hyp"][0], x, g["u"], y, c * np.ones_like(y), np.ones_like(y))
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-10, nm
        assert relerr(out["alpha"], g["alpha"]) < 1e-7 and relerr(out["L"], g["L"]) < 1e-7, nm
        assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-7 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-7, nm


def test_G14_composites_with_an_ard_leaf():
    trees = g14_trees()
    x, y = synth_reg(300, 4)
    for nm in ("ard_noise", "ard_scaled_prod", "rqard_sum"):
        g = golden("G14_fit_%s_N300" % nm)
        c = g["mean_hyp"][0]
        hyp = g["cov_hyp"]
        out = O.exact_fit(trees[nm], hyp, 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9, nm
        assert np.allclose(out["dnlZ_cov"], g["dnlZ_cov"], rtol=1e-8, atol=1e-8 * np.max(np.abs(g["dnlZ_cov"]))), nm
        kx, kz = g["kx"], g["kz"]
        for mode, kw in (("train", dict(x=kx)), ("cross", dict(x=kx, z=kz)), ("self", dict(z=kz))):
            mm = "self_test" if mode == "self" else mode
            np.testing.assert_allclose(O.cov_matrix(trees[nm], hyp, 0, mode=mm, **kw), g["k_K_%s" % mode], rtol=1e-13, atol=1e-300)
            for i in range(len(hyp)):
                np.testing.assert_allclose(O.der_matrix(trees[nm], hyp, 0, mode=mm, der=i, **kw), g["k_dK%d_%s" % (i, mode)],
                                           rtol=1e-12, atol=1e-300)
    g = golden("G14_ep_ard_const_N200")
    out = O.ep_fit(trees["ep_ard_const"], g["cov_hyp"], 0, g["x"], g["y"], np.zeros_like(g["y"]))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-9 and relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-7


def test_G15_composites_with_two_ard_leaves():
    trees = g15_trees()
    x, y = synth_reg(300, 4)
    for nm in ("ard_plus_rqard", "ard_times_ard", "scaled_ard_rq_ard"):
        g = golden("G15_fit_%s_N300" % nm)
        c = g["mean_hyp"][0]
        hyp = g["cov_hyp"]
        out = O.exact_fit(trees[nm], hyp, 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9, nm
        assert np.allclose(out["dnlZ_cov"], g["dnlZ_cov"], rtol=1e-8, atol=1e-8 * np.max(np.abs(g["dnlZ_cov"]))), nm
        kx, kz = g["kx"], g["kz"]
        for mode, kw in (("train", dict(x=kx)), ("cross", dict(x=kx, z=kz)), ("self", dict(z=kz))):
            mm = "self_test" if mode == "self" else mode
            np.testing.assert_allclose(O.cov_matrix(trees[nm], hyp, 0, mode=mm, **kw), g["k_K_%s" % mode], rtol=1e-13, atol=1e-300)
            for i in range(len(hyp)):
                np.testing.assert_allclose(O.der_matrix(trees[nm], hyp, 0, mode=mm, der=i, **kw), g["k_dK%d_%s" % (i, mode)],
                                           rtol=1e-12, atol=1e-300)
    g = golden("G15_ep_ard_times_ard_N200")
    out = O.ep_fit(trees["ep_ard_times_ard"], g["cov_hyp"], 0, g["x"], g["y"], np.zeros_like(g["y"]))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-9 and relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-7


@pytest.mark.parametrize("md", [1, 3, 5, 7])
def test_G16_matern_fits_with_the_reference_gradient_N2048(md):
    """Round 3: Matern at N = 2048 with the reference's own gradient (the quirk of Core/cov.py:1173-1177 included) and
    64 predictions; pins the oracle's Matern path beyond the value-only G4b."""
    g = golden("G16_matern%d_N2048" % md)
    x, y = synth_reg(2048, 16)
    c = g["mean_hyp"][0]
    out = O.exact_fit(O.MATERN, g["cov_hyp"], md, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y),
                      faithful=False, matern_reference_compat=True)
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-11
    assert relerr(out["alpha"][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-8
    assert relerr(np.diag(out["L"]), g["L_diag"]) < 1e-11
    assert relerr(out["L"].ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-10
    for k in ("dnlZ_cov", "dnlZ_lik", "dnlZ_mean"):
        assert relerr(out[k], g[k]) < 1e-8, k
    xs = g["pred_xs"]
    ym, ys2, fm, fs2 = O.predict(O.MATERN, g["cov_hyp"], md, g["lik_hyp"][0], x, out["alpha"], out["L"], out["sW"], xs,
                                 c * np.ones((xs.shape[0], 1)))
    assert relerr(fm, g["pred_fm"]) < 1e-9 and relerr(fs2, g["pred_fs2"]) < 1e-8
    assert relerr(ym, g["pred_ym"]) < 1e-9 and relerr(ys2, g["pred_ys2"]) < 1e-8


def test_G17_predict_at_bench_scale_first_batch():
    """The reference's GP.predict on the N = 8192 posterior: the oracle (own fit, general-LU solve like gp.py:415 and the
    triangular form the GPU tests use) on the first 500 of the 16384 recorded points."""
    g = golden("G17_predict_N8192_ns16384")
    N, d, ns = 8192, 16, 16384
    x, y = synth_reg(N, d)
    rng = np.random.RandomState(7)
    xs = rng.randn(ns, d)
    xs[: ns // 4] = x[rng.randint(0, N, ns // 4)] + 0.05 * rng.randn(ns // 4, d)
    xs = xs[:500]
    hyp = np.array([np.log(np.sqrt(d)), 0.0])
    c = float(y.mean())
    out = O.exact_fit(O.RBF, hyp, 0, np.log(0.1), x, y, c * np.ones_like(y), None, nargout=1, faithful=False)
    for faithful in (True, False):
        ym, ys2, fm, fs2 = O.predict(O.RBF, hyp, 0, np.log(0.1), x, out["alpha"], out["L"], out["sW"], xs,
                                     c * np.ones((500, 1)), faithful=faithful)
        assert relerr(fm, g["pred_fm"][:500]) < 1e-9 and relerr(fs2, g["pred_fs2"][:500]) < 1e-8
        assert relerr(ys2, g["pred_ys2"][:500]) < 1e-8


def g18_inputs(N, d, seed, cls=False):
    """make_golden.py g18_inputs: the 8(d) recipe with a constant offset on every third coordinate and ragged length scales."""
    x, y = (synth_cls if cls else synth_reg)(N, d, seed)
    x = x.copy()
    x[:, ::3] += 40.0
    return x, y


@pytest.mark.parametrize("tag,kind,N,d,seed", [("rbfard_d100_N1500", O.RBFARD, 1500, 100, 3), ("rbfard_d65_N700", O.RBFARD, 700, 65, 4),
                                                ("rqard_d70_N700", O.RQARD, 700, 70, 5)])
def test_G18_ard_fits_beyond_64_dimensions(tag, kind, N, d, seed):
    """ARD kernels with D > 64 (Core/cov.py:872-938, :1356-1425 take any D) -- recorded from the reference in round 4."""
    g = golden("G18_fit_" + tag)
    x, y = g18_inputs(N, d, seed)
    c = g["mean_hyp"][0]
    for faithful in ((True, False) if N <= 700 else (False,)):
        out = O.exact_fit(kind, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=faithful)
        assert relerr(out["nlZ"], g["nlZ"]) < 1e-11 and relerr(out["alpha"], g["alpha"]) < 1e-9
        assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-8 and relerr(out["dnlZ_lik"], g["dnlZ_lik"]) < 1e-8
        assert relerr(np.diag(out["L"]), g["L_diag"]) < 1e-11


def test_G18_ep_ard_d80():
    g = golden("G18_ep_rbfard_d80_N300")
    x, y = g18_inputs(300, 80, 6, cls=True)
    out = O.ep_fit(O.RBFARD, g["cov_hyp"], 0, x, y, np.zeros_like(y))
    assert relerr(out["nlZ"], g["nlZ"]) < 1e-10 and relerr(out["alpha"], g["alpha"]) < 1e-8
    assert relerr(out["dnlZ_cov"], g["dnlZ_cov"]) < 1e-7 and relerr(out["ttau"], g["ttau"]) < 1e-9
