"""GPU: the parity holes the round-2 verdict listed, closed with fixtures recorded from the reference itself.

* Matern d in {1, 3, 5, 7} fits at N = 2048 (Core/cov.py:1124-1182 through Core/inf.py:353-384): G16 fixtures carry the
  reference's nlZ, gradients (its derivative quirk of :1173-1177 included), alpha / L samples and 64 predictions.
* GP.predict at the bench scale (Core/gp.py:349-437): the cfg-2 posterior (N = 8192) and 16384 test points in ONE call,
  every point against the G17 fixture AND against oracle.predict.
* The randomised sweep that used to live in tools/fuzz_exact.py, with a fixed seed (random n <= 2600, d <= 19,
  RBF / RBFard / Matern / RQ through the C ABI against the oracle).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, relerr, synth_reg
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _flat(d):
    return np.array(list(d.mean) + list(d.cov) + list(d.lik), dtype=float)


@pytest.mark.parametrize("md", [1, 3, 5, 7])
def test_matern_fit_reference_fixture_N2048(lib, md):
    import pygps_amd as pyGPs
    g = golden("G16_matern%d_N2048" % md)
    N, d = 2048, 16
    x, y = synth_reg(N, d)
    gref = np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])
    for compat in (True, False):
        k = pyGPs.cov.Matern(np.log(np.sqrt(16.0)), md, 0.0)
        k.reference_compat = compat
        m = pyGPs.GPR()
        m.setPrior(kernel=k)
        m.setNoise(np.log(0.1))
        m.setData(x, y)
        nlZ, dnlZ, post = m.getPosterior()
        assert relerr(nlZ, g["nlZ"]) < 1e-9                                       # north star: 1e-8
        assert relerr(post.alpha[g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7   # 1e-6
        L = np.asarray(post.L)
        assert relerr(np.diag(L), g["L_diag"]) < 1e-9 and relerr(L.ravel()[g["L_flat_idx"]], g["L_sample"]) < 1e-8
        if compat:
            # the reference's own gradient vector (its derivative w.r.t. log ell is the derivative of K, not of t)
            assert relerr(_flat(dnlZ), gref) < 1e-7
            ym, ys2, fm, fs2, lp = m.predict(g["pred_xs"])
            assert relerr(ym, g["pred_ym"]) < 1e-8 and relerr(fm, g["pred_fm"]) < 1e-8
            assert relerr(fs2, g["pred_fs2"]) < 1e-7 and relerr(ys2, g["pred_ys2"]) < 1e-7
        else:
            # the default (mathematically correct) derivative: the oracle with the quirk switched off
            c = m.meanfunc.hyp[0]
            ref = O.exact_fit(O.MATERN, np.array(m.covfunc.hyp), md, m.likfunc.hyp[0], x, y, c * np.ones((N, 1)),
                              np.ones((N, 1)), faithful=False, matern_reference_compat=False)
            want = np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]])
            assert relerr(_flat(dnlZ), want) < 1e-7
            assert relerr(nlZ, ref["nlZ"]) < 1e-9
            # the mean and noise derivatives do not depend on the quirk (both covariance derivatives do: the reference
            # overwrites the distance matrix with K before the der branches, Core/cov.py:1173-1176)
            assert abs(_flat(dnlZ)[0] - gref[0]) < 1e-7 * abs(gref[0]) and abs(_flat(dnlZ)[3] - gref[3]) < 1e-7 * abs(gref[3])


def test_matern3_fit_vs_oracle_N4096(lib):
    import pygps_amd as pyGPs
    N, d = 4096, 16
    x, y = synth_reg(N, d, seed=3)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.Matern(np.log(np.sqrt(16.0)) + 0.2, 3, 0.1))
    m.setNoise(np.log(0.15))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    c = m.meanfunc.hyp[0]
    ref = O.exact_fit(O.MATERN, np.array(m.covfunc.hyp), 3, m.likfunc.hyp[0], x, y, c * np.ones((N, 1)), np.ones((N, 1)),
                      faithful=False, matern_reference_compat=False)
    assert relerr(nlZ, ref["nlZ"]) < 1e-9
    assert relerr(post.alpha, ref["alpha"]) < 1e-7
    assert relerr(_flat(dnlZ), np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]])) < 1e-7
    assert relerr(np.diag(np.asarray(post.L)), np.diag(ref["L"])) < 1e-9


def test_predict_bench_scale_every_point(lib):
    """N = 8192 posterior, 16384 test points in one call (one batch of the two-level blocked multi-rhs solve of
    csrc/predict.hip): fm / fs2 / ym / ys2 of EVERY point against the reference's own output (G17) and the oracle."""
    import pygps_amd as pyGPs
    g = golden("G17_predict_N8192_ns16384")
    N, d, ns = 8192, 16, 16384
    x, y = synth_reg(N, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0))
    m.setNoise(np.log(0.1))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    rng = np.random.RandomState(7)                                               # make_golden.py g17: same draws
    xs = rng.randn(ns, d)
    xs[: ns // 4] = x[rng.randint(0, N, ns // 4)] + 0.05 * rng.randn(ns // 4, d)
    ym, ys2, fm, fs2, lp = m.predict(xs)
    assert ym.shape == (ns, 1) and fs2.shape == (ns, 1)
    scale = float(np.max(np.abs(g["pred_fm"])))
    assert np.max(np.abs(fm - g["pred_fm"])) < 1e-8 * scale and np.max(np.abs(ym - g["pred_ym"])) < 1e-8 * scale
    assert np.max(np.abs(fs2 - g["pred_fs2"])) < 1e-7 * float(np.max(g["pred_fs2"]))
    assert np.max(np.abs(ys2 - g["pred_ys2"])) < 1e-7 * float(np.max(g["pred_ys2"]))
    # the oracle on the same points, from its own fit (independent of the device's alpha / L)
    c = m.meanfunc.hyp[0]
    ref = O.exact_fit(O.RBF, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, y, c * np.ones((N, 1)), np.ones((N, 1)),
                      nargout=2, faithful=False)
    rym, rys2, rfm, rfs2 = O.predict(O.RBF, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, ref["alpha"], ref["L"],
                                     ref["sW"], xs, c * np.ones((ns, 1)), faithful=False)
    assert np.max(np.abs(fm - rfm)) < 1e-8 * scale and np.max(np.abs(fs2 - rfs2)) < 1e-7 * float(np.max(rfs2))
    # a second call with a ragged count that spans two device batches
    ym2, ys22, fm2, fs22, lp2 = m.predict(xs[:16384 - 77])
    assert np.array_equal(fm2, fm[:16384 - 77]) and np.max(np.abs(fs22 - fs2[:16384 - 77])) < 1e-12


def test_randomised_exact_fits_fixed_seed(lib):
    """Random sizes (ragged against the 128 padding and the 512 panels), dimensions, kernels and hypers through the C ABI
    against the oracle -- nlZ, alpha and every gradient."""
    from pygps_amd import _lib
    rng = np.random.RandomState(20260928)
    worst = dict(nlZ=0.0, alpha=0.0, grad=0.0)
    for case in range(24):
        n = int(rng.choice([rng.randint(1, 200), rng.randint(200, 1700), rng.randint(1536, 2600)]))
        d = int(rng.randint(1, 20))
        kind = int(rng.choice([O.RBF, O.RBFARD, O.MATERN, O.MATERN, O.RQ]))
        para = int(rng.choice([1, 3, 5, 7])) if kind == O.MATERN else 0
        x = rng.randn(n, d) * rng.uniform(0.5, 2.0)
        y = np.sin(x.sum(1, keepdims=True)) + 0.2 * rng.randn(n, 1)
        nh = {O.RBF: 2, O.RBFARD: d + 1, O.MATERN: 2, O.RQ: 3}[kind]
        hyp = rng.uniform(-0.5, 1.0, nh)
        log_sn = float(rng.uniform(-2.5, -0.5))
        mvec = np.full((n, 1), float(y.mean()))
        dm = np.ones((1, n))
        ref = O.exact_fit(kind, hyp, para, log_sn, x, y, mvec, dm=dm.T, faithful=False, matern_reference_compat=False)
        h = C.c_void_p()
        assert lib.pgp_init(0, C.byref(h)) == 0
        try:
            xx = np.ascontiguousarray(x)
            yy = np.ascontiguousarray(y).ravel()
            assert lib.pgp_set_data(h, _lib.ptr(xx), n, d, _lib.ptr(yy)) == 0
            alpha, nlZ, gvec = np.zeros(n), np.zeros(1), np.zeros(1 + nh + 1)
            mv, dmv, hv = np.ascontiguousarray(mvec).ravel(), np.ascontiguousarray(dm), np.ascontiguousarray(hyp)
            rc = lib.pgp_exact_fit(h, kind, _lib.ptr(hv), nh, para, 0, log_sn, _lib.ptr(mv), _lib.ptr(dmv), 1, 3,
                                   _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(gvec), None)
            assert rc == 0, (rc, n, d, kind)
        finally:
            lib.pgp_destroy(h)
        gref = np.concatenate([np.ravel(ref["dnlZ_mean"]), np.ravel(ref["dnlZ_cov"]), np.ravel(ref["dnlZ_lik"])])
        e1 = abs(nlZ[0] - ref["nlZ"]) / max(1.0, abs(ref["nlZ"]))
        e2 = np.abs(alpha - ref["alpha"].ravel()).max() / max(1e-300, np.abs(ref["alpha"]).max())
        e3 = np.abs(gvec - gref).max() / max(1.0, np.abs(gref).max())
        assert e1 < 1e-9 and e2 < 1e-7 and e3 < 1e-7, (case, n, d, kind, para, e1, e2, e3)
        worst = dict(nlZ=max(worst["nlZ"], e1), alpha=max(worst["alpha"], e2), grad=max(worst["grad"], e3))
    print("worst", worst)
