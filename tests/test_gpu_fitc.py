"""GPU: FITC sparse regression -- FITC_Exact.evaluate (Core/inf.py:386-455), FITCOfKernel
(Core/cov.py:332-390), GPR_FITC (Core/gp.py:934-1100) and the dense-L predict branch (Core/gp.py:404-417), through
pgp_fitc_fit / pgp_fitc_predict, against golden vectors recorded from the reference (G12) and the oracle."""
import numpy as np
import pytest

from conftest import golden, relerr, synth_reg
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def test_G12_fitc_demo_default_inducing_grid():
    """Demo/Regression/demo_GPR_FITC.py example 1: default 5-point grid, default hypers."""
    import pygps_amd as pyGPs
    g = golden("G12_fitc_demo_default_u")
    m = pyGPs.GPR_FITC()
    m.setData(g["x"], g["y"])
    assert np.allclose(m.u, g["u"]) and isinstance(m.covfunc, pyGPs.cov.FITCOfKernel)
    nlZ, dnlZ, post = m.getPosterior()
    assert type(nlZ) is np.float64 and relerr(nlZ, g["nlZ"]) < 1e-9
    assert post.alpha.shape == g["alpha"].shape and post.L.shape == g["L"].shape
    assert relerr(post.alpha, g["alpha"]) < 1e-6 and relerr(post.L, g["L"]) < 1e-6 and relerr(post.sW, g["sW"]) < 1e-14
    assert relerr(dnlZ.cov, g["dnlZ_cov"]) < 1e-7 and relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-7
    assert relerr(dnlZ.mean, g["dnlZ_mean"]) < 1e-7
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fs2, g["pred_fs2"]) < 1e-6 and relerr(ys2, g["pred_ys2"]) < 1e-6
    # the FITC covariance triple, API parity (Core/cov.py:352-369)
    K, Kuu, Ku = m.covfunc.getCovMatrix(x=g["x"], mode="train")
    assert K.shape == (20, 1) and Kuu.shape == (5, 5) and Ku.shape == (5, 20)
    assert relerr(Ku, O.cov_matrix(O.RBF, g["cov_hyp"], 0, x=g["u"], z=g["x"], mode="cross")) < 1e-13


def _tree(nm):
    if nm == "sum":
        return ("sum", ("leaf", O.RBF, 0), ("scale", ("leaf", O.RQ, 0)))
    return {"rbf": O.RBF, "matern5": O.MATERN, "rbfard": O.RBFARD}[nm]


def _kernel(nm, hyp):
    from pygps_amd import cov
    h = [float(v) for v in hyp]
    if nm == "rbf":
        return cov.RBF(h[0], h[1])
    if nm == "matern5":
        return cov.Matern(h[0], 5, h[1])
    if nm == "rbfard":
        return cov.RBFard(log_ell_list=h[:4], log_sigma=h[4])
    return cov.RBF(h[0], h[1]) + cov.RQ(h[3], h[4], h[5]) * h[2]


@pytest.mark.parametrize("nm", ["rbf", "matern5", "rbfard", "sum"])
def test_G12_fitc_N1500_nu160(nm):
    import pygps_amd as pyGPs
    g = golden("G12_fitc_%s_N1500_nu160" % nm)
    x, y = synth_reg(1500, 4)
    m = pyGPs.GPR_FITC()
    m.setData(x, y)
    k = _kernel(nm, g["cov_hyp"])
    m.setPrior(kernel=k, inducing_points=g["u"])
    if nm == "matern5":
        m.covfunc.covfunc.reference_compat = True            # golden gradients hold the reference's Matern quirk (Q4)
    m.setNoise(g["lik_hyp"][0])
    assert list(np.asarray(m.covfunc.hyp, float)) == list(g["cov_hyp"])
    nlZ, dnlZ, post = m.getPosterior()
    assert relerr(nlZ, g["nlZ"]) < 1e-8
    # alpha and post.L go through inv(Kuu + 1e-8 I): compare what they are used for (predictions) tightly, themselves loosely
    assert relerr(post.alpha, g["alpha"]) < 1e-3 and relerr(post.L, g["L"]) < 1e-3
    scale = np.max(np.abs(g["dnlZ_cov"]))
    assert np.max(np.abs(np.array(dnlZ.cov) - g["dnlZ_cov"])) < 1e-6 * scale and relerr(dnlZ.lik, g["dnlZ_lik"]) < 1e-6
    assert relerr(dnlZ.mean, g["dnlZ_mean"]) < 1e-6
    ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
    assert relerr(ym, g["pred_ym"]) < 1e-7 and relerr(fs2, g["pred_fs2"]) < 1e-5
    if nm == "rbf":
        m.optimize(x, y, numIterations=10)
        assert abs(m.nlZ - g["opt_nlZ"]) < 1e-5 * abs(g["opt_nlZ"])
        assert relerr(m.predict(g["pred_xs"])[0], g["opt_ym"]) < 1e-4


def test_fitc_against_oracle_ragged_sizes_and_gradient_fd():
    """n and nu not multiples of 128; directional finite difference of nlZ along the analytic gradient."""
    import pygps_amd as pyGPs
    rng = np.random.RandomState(5)
    n, nu, d = 700, 77, 3
    x = rng.randn(n, d)
    y = np.sin(x.sum(1, keepdims=True)) + 0.1 * rng.randn(n, 1)
    u = x[rng.choice(n, nu, replace=False)] + 0.05 * rng.randn(nu, d)
    hyp = np.array([0.4, 0.2])
    out = O.fitc_fit(O.RBF, hyp, 0, np.log(0.2), x, u, y, np.zeros_like(y), None)
    m = pyGPs.GPR_FITC()
    m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(0.4, 0.2), inducing_points=u)
    m.setNoise(np.log(0.2))
    nlZ, dnlZ, post = m.getPosterior(x, y)
    assert relerr(nlZ, out["nlZ"]) < 1e-9 and relerr(dnlZ.cov, out["dnlZ_cov"]) < 1e-6 and relerr(dnlZ.lik, out["dnlZ_lik"]) < 1e-6
    fm, fs2 = O.fitc_predict(O.RBF, hyp, 0, u, out["alpha"], out["L"], x[:9] + 0.1, np.zeros(9))
    _, _, fm2, fs22, _ = m.predict(x[:9] + 0.1)
    assert relerr(fm2, fm) < 1e-7 and relerr(fs22, fs2) < 1e-5
    g = np.array(dnlZ.cov + dnlZ.lik)
    e = 1e-5
    vals = []
    for sgn in (1, -1):
        m.covfunc.hyp = list(hyp + sgn * e * g[:2] / np.linalg.norm(g))
        m.setNoise(np.log(0.2) + sgn * e * g[2] / np.linalg.norm(g))
        vals.append(m.getPosterior(x, y, der=False)[0])
    assert abs((vals[0] - vals[1]) / (2 * e) - np.linalg.norm(g)) < 1e-4 * np.linalg.norm(g)


def test_fitc_contract_errors():
    import pygps_amd as pyGPs
    x = np.random.RandomState(0).randn(30, 2)
    y = np.sign(x[:, :1])
    with pytest.raises(Exception, match="Only covFITC supported"):
        pyGPs.inf.FITC_Exact().evaluate(pyGPs.mean.Zero(), pyGPs.cov.RBF(), pyGPs.lik.Gauss(), x, y, 2)
    with pytest.raises(Exception, match="Exact inference only possible with Gaussian likelihood"):
        pyGPs.inf.FITC_Exact().evaluate(pyGPs.mean.Zero(), pyGPs.cov.RBF().fitc(x[:5]), pyGPs.lik.Erf(), x, y, 2)
    with pytest.raises(Exception, match="Dimensionality of inducing inputs"):
        pyGPs.inf.FITC_Exact().evaluate(pyGPs.mean.Zero(), pyGPs.cov.RBF().fitc(np.zeros((4, 3))), pyGPs.lik.Gauss(), x, y, 2)
    m = pyGPs.GPR_FITC()
    with pytest.raises(Exception, match="please call setData"):
        m.setPrior(kernel=pyGPs.cov.RBF())
