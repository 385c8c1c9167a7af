"""GPU: the remaining stationary kernels (RQard, Gabor, Periodic, Noise, Const) and Sum / Product / Scale trees
running as ONE device program (Core/cov.py:230-328, 392-450, 941-982, 1186-1300, 1356-1425).
Everything goes through the C ABI (pgp_set_composite + pgp_cov / pgp_exact_fit / pgp_predict / pgp_ep_fit) and is
compared with the golden vectors recorded from the reference (G11) and with the oracle."""
import numpy as np
import pytest

from conftest import golden, relerr, synth_cls, synth_reg, g11_trees, g14_trees, g15_trees, G11_1D
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def build(tree, hyp, D, compat=True):
    """pygps_amd kernel object for an oracle `kind` tree with flattened hypers `hyp`."""
    from pygps_amd import cov
    hyp = [float(h) for h in hyp]
    if tree[0] == "leaf":
        kind, para = tree[1], tree[2]
        k = {O.RBF: cov.RBF, O.RBFUNIT: cov.RBFunit, O.RQ: cov.RQ, O.GABOR: cov.Gabor, O.PERIODIC: cov.Periodic,
             O.NOISE: cov.Noise, O.CONST: cov.Const}.get(kind)
        if k is not None:
            k = k()
        elif kind == O.RBFARD:
            k = cov.RBFard(D=D)
        elif kind == O.RQARD:
            k = cov.RQard(D=D)
        elif kind == O.MATERN:
            k = cov.Matern(d=para)
        elif kind == O.PIECEPOLY:
            k = cov.PiecePoly(v=para)
        k.reference_compat = compat
        assert len(k.hyp) == len(hyp)
        k.hyp = hyp
        return k
    if tree[0] == "scale":
        return build(tree[1], hyp[1:], D, compat) * hyp[0]
    n1 = O.n_cov_hyp(tree[1] if tree[1][0] != "leaf" else tree[1][1], D)
    a, b = build(tree[1], hyp[:n1], D, compat), build(tree[2], hyp[n1:], D, compat)
    return a + b if tree[0] == "sum" else a * b


def _close(a, b, tol=2e-12):
    scale = max(1.0, float(np.max(np.abs(b))))
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize("nm", ["rqard", "gabor", "noise", "const", "periodic", "sum", "prod", "scale", "tree", "ardsum",
                                "maunaloa"])
def test_G11_kernel_matrices_all_modes(nm):
    g = golden("G11_kernels_rqard_gabor_periodic_noise_const_composites")
    x, z = (g["x1"], g["z1"]) if nm in G11_1D else (g["x"], g["z"])
    hyp = g[nm + "_hyp"]
    k = build(g11_trees()[nm], hyp, x.shape[1])
    assert list(np.asarray(k.hyp, float)) == list(hyp)                  # reference flatten order
    if hasattr(k, "_on_device"):
        assert k._on_device()                                            # runs as one device program (ardsum: one ARD leaf)
    for mode, kw in (("train", dict(x=x)), ("cross", dict(x=x, z=z)), ("self", dict(z=z))):
        m = "self_test" if mode == "self" else mode
        K = k.getCovMatrix(mode=m, **kw)
        assert K.shape == g["%s_K_%s" % (nm, mode)].shape
        _close(K, g["%s_K_%s" % (nm, mode)])
        for i in range(len(hyp)):
            _close(k.getDerMatrix(mode=m, der=i, **kw), g["%s_dK%d_%s" % (nm, i, mode)], 1e-11)
    with pytest.raises(Exception):
        k.getDerMatrix(x=x, mode="train", der=len(hyp))


def test_rqard_default_derivative_is_the_correct_one():
    """reference_compat=False: analytic length-scale derivatives agree with central differences of K."""
    from pygps_amd import cov
    rng = np.random.RandomState(1)
    x = rng.randn(40, 3)
    k = cov.RQard(log_ell_list=[0.1, 0.4, -0.2], log_sigma=0.2, log_alpha=-0.3)
    h0 = list(k.hyp)
    for i in range(5):
        dK = k.getDerMatrix(x=x, mode="train", der=i)
        e = 1e-6
        k.hyp = [h + (e if j == i else 0) for j, h in enumerate(h0)]
        Kp = k.getCovMatrix(x=x, mode="train")
        k.hyp = [h - (e if j == i else 0) for j, h in enumerate(h0)]
        Km = k.getCovMatrix(x=x, mode="train")
        k.hyp = h0
        assert np.max(np.abs(dK - (Kp - Km) / (2 * e))) < 1e-8
        _close(dK, O.der_matrix(O.RQARD, np.array(h0), 0, x=x, mode="train", der=i, matern_reference_compat=False), 1e-11)


def test_G11_maunaloa_fit_predict_optimize():
    """doc/source/demoMaunaLoa.rst:104-108 -- RBF + Periodic*RBF + RQ + (RBF + Noise), 11 hypers, in one program."""
    import pygps_amd as pyGPs
    g = golden("G11_fit_maunaloa_N300")
    x, y = g["x"], g["y"]
    m = pyGPs.GPR()
    m.setPrior(kernel=build(g11_trees()["maunaloa"], g["cov_hyp"], 1))
    m.setNoise(g["lik_hyp"][0])
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    assert relerr(nlZ, g["nlZ"]) < 1e-9
    assert relerr(post.alpha, g["alpha"]) < 1e-6
    assert relerr(np.diag(post.L), g["L_diag"]) < 1e-8
    assert relerr(dnlZ.cov, g["dnlZ_cov"]) < 1e-7 and relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-7
    assert relerr(dnlZ.mean, g["dnlZ_mean"]) < 1e-7
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fs2, g["pred_fs2"]) < 1e-6 and relerr(ys2, g["pred_ys2"]) < 1e-6
    m.optimize(x, y, numIterations=15)
    assert abs(m.nlZ - g["opt_nlZ"]) < 1e-4 * abs(g["opt_nlZ"])           # optimiser path amplifies rounding
    ym2 = m.predict(g["pred_xs"])[0]
    assert relerr(ym2, g["opt_ym"]) < 1e-3


@pytest.mark.parametrize("nm", ["rqard", "scaled_sum"])
def test_G11_fits(nm):
    import pygps_amd as pyGPs
    g = golden("G11_fit_%s_N300" % nm)
    x, y = synth_reg(300, 4)
    tree = g11_trees()[nm]
    m = pyGPs.GPR()
    m.setPrior(kernel=build(tree, g["cov_hyp"], 4))
    m.setNoise(g["lik_hyp"][0])
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    assert relerr(nlZ, g["nlZ"]) < 1e-9 and relerr(post.alpha, g["alpha"]) < 1e-6
    assert np.allclose(dnlZ.cov, g["dnlZ_cov"], rtol=1e-7, atol=1e-7 * np.max(np.abs(g["dnlZ_cov"])))
    assert relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-7
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fs2, g["pred_fs2"]) < 1e-6
    if nm == "rqard":        # default (correct) length-scale gradient against the oracle without the quirk + FD
        k = build(tree, g["cov_hyp"], 4, compat=False)
        m.setPrior(kernel=k)
        m.setData(x, y)
        nlZ2, dn2, _ = m.getPosterior()
        c = m.meanfunc.hyp[0]
        out = O.exact_fit(O.RQARD, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False,
                          matern_reference_compat=False)
        assert relerr(nlZ2, out["nlZ"]) < 1e-9 and relerr(dn2.cov, out["dnlZ_cov"]) < 1e-7
        assert abs(dn2.cov[0]) > 1.0


def test_G11_ep_with_a_composite_kernel():
    import pygps_amd as pyGPs
    g = golden("G11_ep_composite_N200")
    m = pyGPs.GPC()
    m.setPrior(kernel=build(g11_trees()["ep_composite"], g["cov_hyp"], 3))
    nlZ, dnlZ, post = m.getPosterior(g["x"], g["y"])
    assert relerr(nlZ, g["nlZ"]) < 1e-8 and relerr(post.alpha, g["alpha"]) < 1e-6 and relerr(post.sW, g["sW"]) < 1e-6
    assert relerr(dnlZ.cov, g["dnlZ_cov"]) < 1e-6
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"], ys=np.ones((5, 1)))
    assert relerr(ym, g["pred_ym"]) < 1e-7 and relerr(lp, g["pred_lp"]) < 1e-7


def test_composite_limits_and_ard_leaves():
    """More than 8 leaves, or more than two ARD leaves: not a device program -- getCovMatrix / getDerMatrix combine the
    children's device-built matrices, and fits / predictions take the dense path (csrc/dense.hip: factorisation, alpha, nlZ, the
    Hadamard sums and the predictive solve on the device from caller-built matrices).  Core/cov.py:230-328 composes anything;
    nothing raises."""
    import pygps_amd as pyGPs
    from pygps_amd import cov
    rng = np.random.RandomState(0)
    x = rng.randn(30, 2)
    y = rng.randn(30, 1)
    big = cov.RBF(0.1, 0.0)
    for i in range(8):
        big = big + cov.RBF(0.1 * i, -0.2)
    assert not big._on_device()
    ref = sum(O.cov_matrix(O.RBF, [0.1 * i, -0.2], 0, x=x, mode="train") for i in range(8)) + O.cov_matrix(O.RBF, [0.1, 0.0], 0, x=x, mode="train")
    _close(big.getCovMatrix(x=x, mode="train"), ref)
    ard = cov.RBFard(D=2) * cov.RQard(D=2) + cov.RBFard(D=2)      # three ARD leaves: no device program
    assert not ard._on_device() and (cov.RBFard(D=2) * cov.RBF())._on_device() and (cov.RBFard(D=2) * cov.RQard(D=2))._on_device()
    t3 = ("sum", ("prod", ("leaf", O.RBFARD, 0), ("leaf", O.RQARD, 0)), ("leaf", O.RBFARD, 0))
    _close(ard.getCovMatrix(x=x, mode="train"), O.cov_matrix(t3, np.array(ard.hyp, float), 0, x=x, mode="train"))
    tbig = ("leaf", O.RBF, 0)
    for i in range(8):
        tbig = ("sum", tbig, ("leaf", O.RBF, 0))
    xs = rng.randn(9, 2)
    for k, tree in ((big, tbig), (ard, t3)):
        m = pyGPs.GPR()
        m.setPrior(kernel=k)
        m.setNoise(np.log(0.3))
        m.setData(x, y)
        nlZ, dnlZ, post = m.getPosterior()
        c = m.meanfunc.hyp[0]
        hyp = np.array(m.covfunc.hyp, float)
        ref = O.exact_fit(tree, hyp, 0, m.likfunc.hyp[0], x, y, c * np.ones((30, 1)), np.ones((30, 1)), faithful=False,
                          matern_reference_compat=False)
        assert abs(nlZ - ref["nlZ"]) < 1e-10 * abs(ref["nlZ"])
        _close(post.alpha, ref["alpha"], 1e-9)
        _close(np.array(dnlZ.mean + dnlZ.cov + dnlZ.lik), np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]]), 1e-8)
        _close(np.asarray(post.L), ref["L"], 1e-10)
        nlZ2, post2 = m.getPosterior(der=False)                       # value-only call: no inverse
        assert abs(nlZ2 - nlZ) < 1e-12 * abs(nlZ)
        ym, ys2, fm, fs2, lp = m.predict(xs)
        rym, rys2, rfm, rfs2 = O.predict(tree, hyp, 0, m.likfunc.hyp[0], x, ref["alpha"], ref["L"], ref["sW"], xs, c * np.ones((9, 1)))
        _close(fm, rfm, 1e-9)
        _close(fs2, rfs2, 1e-8)
    # exactly at the limit: 8 products from (a+b)*(c+d)*(e+f)
    k8 = (cov.RBF(0.1, 0.) + cov.RQ(0.2, 0., 0.1)) * (cov.RBF(0.3, 0.) + cov.Matern(0.2, 3, 0.)) * (cov.RBFunit(0.5) + cov.Const(-1.))
    assert k8._on_device()
    t = ("prod", ("prod", ("sum", ("leaf", O.RBF, 0), ("leaf", O.RQ, 0)), ("sum", ("leaf", O.RBF, 0), ("leaf", O.MATERN, 3))),
         ("sum", ("leaf", O.RBFUNIT, 0), ("leaf", O.CONST, 0)))
    h = np.array(k8.hyp, float)
    _close(k8.getCovMatrix(x=x, mode="train"), O.cov_matrix(t, h, 0, x=x, mode="train"))
    for i in range(len(h)):
        _close(k8.getDerMatrix(x=x, mode="train", der=i), O.der_matrix(t, h, 0, x=x, mode="train", der=i, matern_reference_compat=False), 1e-11)
    m = pyGPs.GPR()
    m.setPrior(kernel=k8)
    nlZ, dnlZ, post = m.getPosterior(x, y)
    out = O.exact_fit(t, h, 0, m.likfunc.hyp[0], x, y, np.zeros_like(y), None, faithful=False, matern_reference_compat=False)
    assert relerr(nlZ, out["nlZ"]) < 1e-10 and relerr(dnlZ.cov, out["dnlZ_cov"]) < 1e-8


def test_ep_on_a_tree_that_is_no_device_program():
    """GPC + EP with a covariance tree the device programs cannot hold (three ARD leaves; nine leaves): K and the derivative
    matrices are handed in (pgp_ep_fit_dense, pgp_dense_grad_term), sweeps, posterior, alpha, nlZ, the gradient sums and the
    predictive solve run on the device -- against the oracle's EP (Core/inf.py:731-806), ragged n, with a warm start."""
    import pygps_amd as pyGPs
    from pygps_amd import cov
    rng = np.random.RandomState(3)
    for n in (150, 333):
        x = rng.randn(n, 2)
        y = np.sign(np.sin(1.7 * x[:, :1]) + 0.3 * x[:, 1:2] + 0.2 * rng.randn(n, 1)); y[y == 0] = 1
        ard = cov.RBFard(D=2) * cov.RQard(D=2) + cov.RBFard(D=2)
        ard.hyp = list(0.3 * rng.randn(len(ard.hyp)))
        t3 = ("sum", ("prod", ("leaf", O.RBFARD, 0), ("leaf", O.RQARD, 0)), ("leaf", O.RBFARD, 0))
        big = cov.RBF(0.1, 0.0)
        tbig = ("leaf", O.RBF, 0)
        for i in range(8):
            big = big + cov.RBF(0.1 * i, -0.2)
            tbig = ("sum", tbig, ("leaf", O.RBF, 0))
        xs = rng.randn(11, 2)
        for k, tree in ((ard, t3), (big, tbig)):
            assert not k._on_device()
            m = pyGPs.GPC()
            m.setPrior(mean=pyGPs.mean.Zero(), kernel=k)
            nlZ, dnlZ, post = m.getPosterior(x, y)
            hyp = np.array(k.hyp, float)
            ref = O.ep_fit(tree, hyp, 0, x, y, np.zeros_like(y), matern_reference_compat=False)
            assert m.inffunc.sweeps == ref["sweeps"]
            assert relerr(nlZ, ref["nlZ"]) < 1e-8, (n, nlZ, ref["nlZ"])
            assert relerr(post.alpha, ref["alpha"]) < 1e-6 and relerr(post.sW, ref["sW"]) < 1e-6
            assert relerr(dnlZ.cov, ref["dnlZ_cov"]) < 1e-6
            ym, ys2, fm, fs2, lp = m.predict(xs)
            rym, rys2, rfm, rfs2 = O.predict(tree, hyp, 0, None, x, ref["alpha"], ref["L"], ref["sW"], xs, np.zeros((11, 1)), gauss=False,
                                             faithful=False)
            assert relerr(fm, rfm) < 1e-7 and relerr(fs2, rfs2) < 1e-6
            nlZ2, dnlZ2, post2 = m.getPosterior(x, y)                      # warm start from the converged sites: the same optimum
            assert relerr(nlZ2, nlZ) < 1e-6


def test_periodic_needs_1d_inputs():
    from pygps_amd import cov
    with pytest.raises(AssertionError):
        cov.Periodic().getCovMatrix(x=np.zeros((4, 2)), mode="train")


@pytest.mark.parametrize("nm", ["ard_noise", "ard_scaled_prod", "rqard_sum"])
def test_G14_fit_with_an_ard_leaf_inside_the_program(nm):
    """One ARD leaf (own weighted distance) inside a Sum/Product/Scale tree: kernel matrices in all modes with every
    derivative, fit with all gradients (per-dimension length-scale sums in the second pass), predict."""
    import pygps_amd as pyGPs
    g = golden("G14_fit_%s_N300" % nm)
    tree = g14_trees()[nm]
    hyp = g["cov_hyp"]
    k = build(tree, hyp, 4)
    assert k._on_device() and list(np.asarray(k.hyp, float)) == list(hyp)
    kx, kz = g["kx"], g["kz"]
    for mode, kw in (("train", dict(x=kx)), ("cross", dict(x=kx, z=kz)), ("self", dict(z=kz))):
        mm = "self_test" if mode == "self" else mode
        _close(k.getCovMatrix(mode=mm, **kw), g["k_K_%s" % mode])
        for i in range(len(hyp)):
            _close(k.getDerMatrix(mode=mm, der=i, **kw), g["k_dK%d_%s" % (i, mode)], 1e-11)
    x, y = synth_reg(300, 4)
    m = pyGPs.GPR()
    m.setPrior(kernel=k)
    m.setNoise(g["lik_hyp"][0])
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    assert relerr(nlZ, g["nlZ"]) < 1e-9 and relerr(post.alpha, g["alpha"]) < 1e-6
    assert np.allclose(dnlZ.cov, g["dnlZ_cov"], rtol=1e-7, atol=1e-7 * np.max(np.abs(g["dnlZ_cov"])))
    assert relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-7 and relerr(dnlZ.mean, g["dnlZ_mean"]) < 1e-7
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fs2, g["pred_fs2"]) < 1e-6
    if nm == "rqard_sum":                # default (correct) RQard length-scale gradient vs the oracle without the quirk
        m.setPrior(kernel=build(tree, hyp, 4, compat=False))
        m.setData(x, y)
        nlZ2, dn2, _ = m.getPosterior()
        c = m.meanfunc.hyp[0]
        out = O.exact_fit(tree, hyp, 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False,
                          matern_reference_compat=False)
        assert relerr(dn2.cov, out["dnlZ_cov"]) < 1e-7 and abs(dn2.cov[0]) > 1e-3


def test_G14_ep_with_an_ard_leaf():
    import pygps_amd as pyGPs
    g = golden("G14_ep_ard_const_N200")
    m = pyGPs.GPC()
    m.setPrior(kernel=build(g14_trees()["ep_ard_const"], g["cov_hyp"], 3))
    nlZ, dnlZ, post = m.getPosterior(g["x"], g["y"])
    assert relerr(nlZ, g["nlZ"]) < 1e-8 and relerr(post.alpha, g["alpha"]) < 1e-6 and relerr(dnlZ.cov, g["dnlZ_cov"]) < 1e-6
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"], ys=np.ones((5, 1)))
    assert relerr(ym, g["pred_ym"]) < 1e-7 and relerr(lp, g["pred_lp"]) < 1e-7


@pytest.mark.parametrize("nm", ["ard_plus_rqard", "ard_times_ard", "scaled_ard_rq_ard"])
def test_G15_fit_with_two_ard_leaves_inside_the_program(nm):
    """Two ARD leaves (each with its own weighted distance) inside a Sum/Product/Scale tree, against the reference's
    composition (Core/cov.py:230-296): kernel matrices in all modes with every derivative, fit with all gradients (one
    per-dimension length-scale pass per ARD leaf), predict."""
    import pygps_amd as pyGPs
    g = golden("G15_fit_%s_N300" % nm)
    tree = g15_trees()[nm]
    hyp = g["cov_hyp"]
    k = build(tree, hyp, 4)
    assert k._on_device() and list(np.asarray(k.hyp, float)) == list(hyp)
    kx, kz = g["kx"], g["kz"]
    for mode, kw in (("train", dict(x=kx)), ("cross", dict(x=kx, z=kz)), ("self", dict(z=kz))):
        mm = "self_test" if mode == "self" else mode
        _close(k.getCovMatrix(mode=mm, **kw), g["k_K_%s" % mode])
        for i in range(len(hyp)):
            _close(k.getDerMatrix(mode=mm, der=i, **kw), g["k_dK%d_%s" % (i, mode)], 1e-11)
    x, y = synth_reg(300, 4)
    m = pyGPs.GPR()
    m.setPrior(kernel=k)
    m.setNoise(g["lik_hyp"][0])
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    assert relerr(nlZ, g["nlZ"]) < 1e-9 and relerr(post.alpha, g["alpha"]) < 1e-6
    assert np.allclose(dnlZ.cov, g["dnlZ_cov"], rtol=1e-7, atol=1e-7 * np.max(np.abs(g["dnlZ_cov"])))
    assert relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-7 and relerr(dnlZ.mean, g["dnlZ_mean"]) < 1e-7
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fs2, g["pred_fs2"]) < 1e-6
    # the default (mathematically correct) RQard length-scale gradient against the oracle without the reference's quirk
    m.setPrior(kernel=build(tree, hyp, 4, compat=False))
    m.setData(x, y)
    nlZ2, dn2, _ = m.getPosterior()
    c = m.meanfunc.hyp[0]
    out = O.exact_fit(tree, hyp, 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False,
                      matern_reference_compat=False)
    assert relerr(dn2.cov, out["dnlZ_cov"]) < 1e-7


def test_G15_ep_with_two_ard_leaves_and_larger_sizes_vs_oracle():
    import pygps_amd as pyGPs
    g = golden("G15_ep_ard_times_ard_N200")
    tree = g15_trees()["ep_ard_times_ard"]
    m = pyGPs.GPC()
    m.setPrior(kernel=build(tree, g["cov_hyp"], 3))
    nlZ, dnlZ, post = m.getPosterior(g["x"], g["y"])
    assert relerr(nlZ, g["nlZ"]) < 1e-8 and relerr(post.alpha, g["alpha"]) < 1e-6 and relerr(dnlZ.cov, g["dnlZ_cov"]) < 1e-6
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"], ys=np.ones((5, 1)))
    assert relerr(ym, g["pred_ym"]) < 1e-7 and relerr(lp, g["pred_lp"]) < 1e-7
    # d > 16 (several coordinate slabs) and a ragged n, against the oracle
    n, d = 700, 37
    x, y = synth_reg(n, d, seed=3)
    tr = g15_trees()["ard_plus_rqard"]
    rng = np.random.RandomState(1)
    hyp = np.concatenate([np.log(np.sqrt(d)) + 0.3 * rng.randn(d), [0.1], np.log(np.sqrt(d)) + 0.3 * rng.randn(d), [-0.2, 0.3]])
    mm = pyGPs.GPR()
    mm.setPrior(kernel=build(tr, hyp, d, compat=False))
    mm.setNoise(np.log(0.1))
    mm.setData(x, y)
    nlZ, dnlZ, post = mm.getPosterior()
    c = mm.meanfunc.hyp[0]
    out = O.exact_fit(tr, hyp, 0, np.log(0.1), x, y, c * np.ones_like(y), np.ones_like(y), faithful=False,
                      matern_reference_compat=False)
    assert relerr(nlZ, out["nlZ"]) < 1e-9 and relerr(post.alpha, out["alpha"]) < 1e-7
    assert np.allclose(dnlZ.cov, out["dnlZ_cov"], rtol=1e-6, atol=1e-7 * np.max(np.abs(out["dnlZ_cov"])))


def test_three_ard_leaves_fit_and_predict_at_N1500_against_the_oracle():
    """A tree the device programs cannot hold (three ARD leaves, 16 hypers) at a size where the dense path's linear algebra
    runs through the panel sweep: fit, all gradients, 200 predictions and a few line searches of the optimiser."""
    import pygps_amd as pyGPs
    from pygps_amd import cov
    rng = np.random.RandomState(2)
    n, d = 1500, 4
    x = rng.randn(n, d)
    y = np.sin(x[:, :1] * 1.3) + 0.5 * np.cos(x[:, 1:2]) + 0.1 * rng.randn(n, 1)
    k = cov.RBFard(log_ell_list=[0.2, 0.4, 0.1, 0.6], log_sigma=0.0) * cov.RQard(log_ell_list=[0.5, 0.3, 0.7, 0.2], log_sigma=-0.1, log_alpha=0.3) \
        + cov.RBFard(log_ell_list=[1.0, 0.8, 1.2, 0.9], log_sigma=-0.5)
    assert not k._on_device()
    tree = ("sum", ("prod", ("leaf", O.RBFARD, 0), ("leaf", O.RQARD, 0)), ("leaf", O.RBFARD, 0))
    m = pyGPs.GPR()
    m.setPrior(kernel=k)
    m.setNoise(np.log(0.15))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    c = m.meanfunc.hyp[0]
    hyp = np.array(m.covfunc.hyp, float)
    ref = O.exact_fit(tree, hyp, 0, m.likfunc.hyp[0], x, y, c * np.ones((n, 1)), np.ones((n, 1)), faithful=False,
                      matern_reference_compat=False)
    assert abs(nlZ - ref["nlZ"]) < 1e-9 * abs(ref["nlZ"])
    _close(post.alpha, ref["alpha"], 1e-7)
    _close(np.array(dnlZ.mean + dnlZ.cov + dnlZ.lik), np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]]), 1e-7)
    xs = rng.randn(200, d)
    ym, ys2, fm, fs2, lp = m.predict(xs)
    rym, rys2, rfm, rfs2 = O.predict(tree, hyp, 0, m.likfunc.hyp[0], x, ref["alpha"], ref["L"], ref["sW"], xs, c * np.ones((200, 1)))
    _close(fm, rfm, 1e-8)
    _close(fs2, rfs2, 1e-7)
    m.optimize(x, y, numIterations=3)
    assert m.nlZ < nlZ
