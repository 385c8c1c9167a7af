"""GPU: round 4 -- ARD kernels with more than 64 input dimensions.

The reference takes any D (Core/cov.py:872-938 RBFard, :1356-1425 RQard: one getDerMatrix per length scale).  Up to round 3
the device's per-coordinate gradient sums stopped at coordinate 63 (and left the rest of the gradient vector unwritten);
they now run on the matrix cores in a product form over chunks of 64 coordinates (csrc/grad.hip ard_dim_reduce).

* G18 fixtures recorded from the reference: RBFard d = 100 (N = 1500), d = 65 (N = 700), RQard d = 70 (N = 700) through
  Exact.evaluate, RBFard d = 80 through EP (N = 300) -- every one of the D + 1 (+1) gradients.
* d in {65, 100, 200} at N ~ 1500 and an EP fit at d = 80 against the oracle.
* the product form's conditioning: coordinates with a large common offset (the sums are shift-invariant; the kernel centres).
* the randomised sweep of round 3 with d up to 130.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, relerr, synth_cls, synth_reg
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _flat(d):
    return np.array(list(d.mean) + list(d.cov) + list(d.lik), dtype=float)


def g18_inputs(N, d, seed, cls=False):
    x, y = (synth_cls if cls else synth_reg)(N, d, seed)
    x = x.copy()
    x[:, ::3] += 40.0
    return x, y


@pytest.mark.parametrize("tag,N,d,seed", [("rbfard_d100_N1500", 1500, 100, 3), ("rbfard_d65_N700", 700, 65, 4),
                                           ("rqard_d70_N700", 700, 70, 5)])
def test_G18_ard_fits_beyond_64_dimensions(lib, tag, N, d, seed):
    import pygps_amd as pyGPs
    g = golden("G18_fit_" + tag)
    x, y = g18_inputs(N, d, seed)
    hyp = [float(v) for v in g["cov_hyp"]]
    rq = tag.startswith("rqard")
    k = pyGPs.cov.RQard(D=d) if rq else pyGPs.cov.RBFard(D=d)
    k.hyp = hyp
    k.reference_compat = True          # RQard: the reference's length-scale derivative (Core/cov.py:1412-1418) for the fixture
    m = pyGPs.GPR()
    m.setPrior(kernel=k)
    m.setNoise(float(g["lik_hyp"][0]))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    assert len(dnlZ.cov) == len(hyp)
    assert relerr(nlZ, g["nlZ"]) < 1e-9 and relerr(post.alpha, g["alpha"]) < 1e-7
    assert relerr(np.diag(np.asarray(post.L)), g["L_diag"]) < 1e-9
    gref = np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])
    assert np.max(np.abs(_flat(dnlZ) - gref)) < 1e-7 * np.max(np.abs(gref))
    # every single length-scale gradient, not only the largest (the unwritten ones of round 3 were garbage of any size)
    assert np.allclose(dnlZ.cov, g["dnlZ_cov"], rtol=1e-6, atol=1e-7 * np.max(np.abs(g["dnlZ_cov"])))
    if rq:                             # the default (mathematically correct) RQard derivative vs the oracle without the quirk
        k2 = pyGPs.cov.RQard(D=d)
        k2.hyp = hyp
        m.setPrior(kernel=k2)
        m.setData(x, y)
        nlZ2, dn2, _ = m.getPosterior()
        c = m.meanfunc.hyp[0]
        out = O.exact_fit(O.RQARD, g["cov_hyp"], 0, g["lik_hyp"][0], x, y, c * np.ones_like(y), np.ones_like(y), faithful=False,
                          matern_reference_compat=False)
        assert relerr(nlZ2, out["nlZ"]) < 1e-9 and relerr(dn2.cov, out["dnlZ_cov"]) < 1e-7
        assert np.max(np.abs(np.array(dn2.cov[:d]))) > 1e-3


def test_G18_ep_ard_d80(lib):
    import pygps_amd as pyGPs
    g = golden("G18_ep_rbfard_d80_N300")
    x, y = g18_inputs(300, 80, 6, cls=True)
    k = pyGPs.cov.RBFard(D=80)
    k.hyp = [float(v) for v in g["cov_hyp"]]
    m = pyGPs.GPC()
    m.setPrior(mean=pyGPs.mean.Zero(), kernel=k)
    nlZ, dnlZ, post = m.getPosterior(x, y)
    assert relerr(nlZ, g["nlZ"]) < 1e-8 and relerr(post.alpha, g["alpha"]) < 1e-6 and relerr(post.sW, g["sW"]) < 1e-6
    assert len(dnlZ.cov) == 81
    assert np.allclose(dnlZ.cov, g["dnlZ_cov"], rtol=1e-5, atol=1e-6 * np.max(np.abs(g["dnlZ_cov"])))


@pytest.mark.parametrize("d,N", [(65, 1500), (100, 1411), (200, 1537), (254, 700)])
def test_rbfard_fit_vs_oracle_high_dim(lib, d, N):
    """nlZ <= 1e-9, ALL D + 1 covariance gradients (and mean, noise) <= 1e-7 against oracle.exact_fit; d = 254 is the largest
    the C ABI takes (ncov <= 255)."""
    import pygps_amd as pyGPs
    rng = np.random.RandomState(d)
    x = rng.randn(N, d) * rng.uniform(0.5, 2.0, d) + rng.uniform(-3.0, 3.0, d)
    w = rng.randn(d, 1)
    y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
    log_ell = np.log(np.sqrt(d)) + rng.uniform(-0.5, 0.7, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBFard(log_ell_list=[float(v) for v in log_ell], log_sigma=0.15))
    m.setNoise(np.log(0.12))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    c = m.meanfunc.hyp[0]
    ref = O.exact_fit(O.RBFARD, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, y, c * np.ones((N, 1)), np.ones((N, 1)),
                      faithful=False)
    want = np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]])
    assert relerr(nlZ, ref["nlZ"]) < 1e-9 and relerr(post.alpha, ref["alpha"]) < 1e-7
    assert len(dnlZ.cov) == d + 1
    assert np.max(np.abs(_flat(dnlZ) - want)) < 1e-7 * np.max(np.abs(want))
    assert np.allclose(dnlZ.cov, ref["dnlZ_cov"], rtol=1e-6, atol=1e-7 * np.max(np.abs(ref["dnlZ_cov"])))


def test_ep_rbfard_d80_vs_oracle(lib):
    import pygps_amd as pyGPs
    N, d = 900, 80
    x, y = synth_cls(N, d, seed=11)
    rng = np.random.RandomState(5)
    log_ell = np.log(np.sqrt(d)) + rng.uniform(-0.3, 0.5, d)
    k = pyGPs.cov.RBFard(log_ell_list=[float(v) for v in log_ell], log_sigma=0.3)
    m = pyGPs.GPC()
    m.setPrior(mean=pyGPs.mean.Zero(), kernel=k)
    nlZ, dnlZ, post = m.getPosterior(x, y)
    out = O.ep_fit(O.RBFARD, np.array(k.hyp), 0, x, y, np.zeros_like(y))
    assert relerr(nlZ, out["nlZ"]) < 1e-8 and relerr(post.alpha, out["alpha"]) < 1e-6
    assert np.max(np.abs(np.array(dnlZ.cov) - out["dnlZ_cov"])) < 1e-7 * np.max(np.abs(out["dnlZ_cov"]))


@pytest.mark.parametrize("offset", [0.0, 1e3, 1e6])
def test_ard_gradient_sums_with_a_common_offset(lib, offset):
    """The per-coordinate sums run as  sum_r R_r x_r^2 + sum_c x_c (C_c x_c - 2 (W'X)_c)  on centred coordinates: a common offset
    of the inputs (a year, a temperature in kelvin) must not cost digits -- the difference form of the reference
    (Core/cov.py:929-930, cdist on one column) does not see it either."""
    import pygps_amd as pyGPs
    N, d = 1100, 24
    x, y = synth_reg(N, d, seed=2)
    x = x + offset
    log_ell = np.log(np.sqrt(d)) + np.linspace(-0.3, 0.3, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBFard(log_ell_list=[float(v) for v in log_ell], log_sigma=0.0))
    m.setNoise(np.log(0.1))
    m.setData(x, y)
    nlZ, dnlZ, post = m.getPosterior()
    c = m.meanfunc.hyp[0]
    ref = O.exact_fit(O.RBFARD, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, y, c * np.ones((N, 1)), np.ones((N, 1)),
                      faithful=True)
    tol = 1e-7 if offset <= 1e3 else 1e-6          # at 1e6 the INPUTS carry 1e-10 relative rounding of their own
    assert relerr(nlZ, ref["nlZ"]) < tol * 1e-2
    assert np.max(np.abs(np.array(dnlZ.cov) - ref["dnlZ_cov"])) < tol * np.max(np.abs(ref["dnlZ_cov"]))


def test_randomised_ard_fits_up_to_130_dimensions(lib):
    """Round 3's fixed-seed sweep, ARD kinds only, d drawn up to 130 (ragged against the slabs of 16 and the chunks of 64)."""
    from pygps_amd import _lib
    rng = np.random.RandomState(20260929)
    for case in range(14):
        n = int(rng.choice([rng.randint(2, 200), rng.randint(200, 1500), rng.randint(1536, 2300)]))
        d = int(rng.choice([rng.randint(1, 64), rng.randint(64, 131), rng.randint(64, 131)]))
        kind = int(rng.choice([O.RBFARD, O.RBFARD, O.RQARD]))
        x = rng.randn(n, d) * rng.uniform(0.5, 2.0) + rng.uniform(-2, 2)
        y = np.sin(x.sum(1, keepdims=True) / np.sqrt(d)) + 0.2 * rng.randn(n, 1)
        nh = d + 1 if kind == O.RBFARD else d + 2
        hyp = np.concatenate([np.log(np.sqrt(d)) + rng.uniform(-0.5, 1.0, d), rng.uniform(-0.5, 0.5, nh - d)])
        log_sn = float(rng.uniform(-2.5, -0.5))
        mvec = np.full((n, 1), float(y.mean()))
        dm = np.ones((1, n))
        ref = O.exact_fit(kind, hyp, 0, log_sn, x, y, mvec, dm=dm.T, faithful=False, matern_reference_compat=False)
        h = C.c_void_p()
        assert lib.pgp_init(0, C.byref(h)) == 0
        try:
            xx = np.ascontiguousarray(x)
            yy = np.ascontiguousarray(y).ravel()
            assert lib.pgp_set_data(h, _lib.ptr(xx), n, d, _lib.ptr(yy)) == 0
            alpha, nlZ, gvec = np.zeros(n), np.zeros(1), np.full(1 + nh + 1, np.nan)
            mv, dmv, hv = np.ascontiguousarray(mvec).ravel(), np.ascontiguousarray(dm), np.ascontiguousarray(hyp)
            rc = lib.pgp_exact_fit(h, kind, _lib.ptr(hv), nh, 0, 0, log_sn, _lib.ptr(mv), _lib.ptr(dmv), 1, 3,
                                   _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(gvec), None)
            assert rc == 0, (rc, n, d, kind)
        finally:
            lib.pgp_destroy(h)
        gref = np.concatenate([np.ravel(ref["dnlZ_mean"]), np.ravel(ref["dnlZ_cov"]), np.ravel(ref["dnlZ_lik"])])
        e1 = abs(nlZ[0] - ref["nlZ"]) / max(1.0, abs(ref["nlZ"]))
        e2 = np.abs(alpha - ref["alpha"].ravel()).max() / max(1e-300, np.abs(ref["alpha"]).max())
        e3 = np.abs(gvec - gref).max() / max(1.0, np.abs(gref).max())
        assert e1 < 1e-9 and e2 < 1e-7 and e3 < 1e-7, (case, n, d, kind, e1, e2, e3)


def _fit_with(lib, option, x, y, kern, log_sn):
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    ctx = _lib.ctx()
    _lib.check(lib.pgp_set_option(ctx, b"gram_assembly", option))
    try:
        m = pyGPs.GPR()
        m.setPrior(kernel=kern())
        m.setNoise(log_sn)
        m.setData(x, y)
        nlZ, dnlZ, post = m.getPosterior()
        return nlZ, _flat(dnlZ), np.array(post.alpha), np.asarray(post.L).copy()
    finally:
        lib.pgp_set_option(ctx, b"gram_assembly", 1)


def test_gram_form_assembly_against_the_difference_form(lib):
    """RBF / RBFard fits at d >= 32 assemble K in the Gram form on the matrix cores (csrc/assemble.hip cov_gram_kernel) when the host's
    bound on the centred, scaled points' squared norms allows it; the reference's cdist is the difference form (Core/cov.py:804,
    :899-901).  The measured parity argument: the same fit with the Gram form forced (2), forbidden (0) and chosen (1) --
    * cfg-3 recipe (SEard, d = 64) and RBF d = 32: forced vs forbidden agree far inside the parity tolerances, the default picks
      the Gram form (bit-identical to forced) and reproduces the reference's fixture;
    * data whose norms break the bound (a cluster 1e3 length scales from the mean): the default must pick the difference form
      (bit-identical to forbidden), and the forced Gram form shows why: it loses digits there."""
    import pygps_amd as pyGPs
    g = golden("G7_rbfard_d64_N2048")
    N, d = 2048, 64
    x, y = synth_reg(N, d)
    kern = lambda: pyGPs.cov.RBFard(log_ell_list=[float(np.log(np.sqrt(d)))] * d, log_sigma=0.0)
    r0, r1, r2 = (_fit_with(lib, o, x, y, kern, np.log(0.1)) for o in (0, 1, 2))
    assert r1[0] == r2[0] and np.array_equal(r1[2], r2[2])                         # the default chose the Gram form
    assert relerr(r2[0], r0[0]) < 1e-12 and relerr(r2[2], r0[2]) < 1e-10 and relerr(r2[1], r0[1]) < 1e-9
    assert relerr(np.diag(r2[3]), np.diag(r0[3])) < 1e-12 and relerr(r2[3], r0[3]) < 1e-11
    assert relerr(r1[0], g["nlZ"]) < 1e-9 and relerr(r1[2][g["alpha_idx"], 0], g["alpha_sample"]) < 1e-7
    gref = np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])
    assert np.max(np.abs(r1[1] - gref)) < 1e-7 * np.max(np.abs(gref))
    # plain RBF at d = 32 (the Gram form starts at 32 coordinates)
    x3, y3 = synth_reg(1500, 32, seed=3)
    k3 = lambda: pyGPs.cov.RBF(np.log(np.sqrt(32.0)) + 0.1, 0.2)
    a0, a1, a2 = (_fit_with(lib, o, x3, y3, k3, np.log(0.15)) for o in (0, 1, 2))
    assert a1[0] == a2[0] and relerr(a2[0], a0[0]) < 1e-12 and relerr(a2[2], a0[2]) < 1e-10 and relerr(a2[1], a0[1]) < 1e-9
    # norms far beyond the bound: a second cluster 1e3 length scales away
    xb = x3.copy()
    xb[::2, 0] += 6.0e3
    b0, b1, b2 = (_fit_with(lib, o, xb, y3, k3, np.log(0.15)) for o in (0, 1, 2))
    assert b1[0] == b0[0] and np.array_equal(b1[2], b0[2]) and np.array_equal(b1[1], b0[1])     # the default chose the difference form
    c = float(y3.mean())
    ref = O.exact_fit(O.RBF, np.array([np.log(np.sqrt(32.0)) + 0.1, 0.2]), 0, np.log(0.15), xb, y3, c * np.ones_like(y3), np.ones_like(y3),
                      faithful=False)
    assert relerr(b1[0], ref["nlZ"]) < 1e-9 and relerr(b1[2], ref["alpha"]) < 1e-7
    assert relerr(b2[2], ref["alpha"]) > 10 * relerr(b1[2], ref["alpha"])        # ... and the forced Gram form is visibly worse here


def test_gram_form_assembly_in_ep_cfg5_recipe(lib):
    """EP builds the full symmetric K (csrc/ep.hip) -- for RBF at d = 32 (cfg 5's recipe) in the Gram form too (the kernel's
    MODE_SYM: mirrored stores, ragged n).  Forbidden vs forced vs chosen at N = 300 (ragged against the 64-tiles) and against the
    reference's fixture at N = 512."""
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    ctx = _lib.ctx()
    res = {}
    try:
        for o in (0, 1, 2):
            _lib.check(lib.pgp_set_option(ctx, b"gram_assembly", o))
            for N in (300, 512):
                x, y = synth_cls(N, 32)
                m = pyGPs.GPC()
                m.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(32.0)), 0.0))
                nlZ, dnlZ, post = m.getPosterior(x, y)
                res[o, N] = (nlZ, np.array(post.alpha), np.array(dnlZ.cov), np.asarray(post.L).copy(), int(m.inffunc.sweeps))
    finally:
        lib.pgp_set_option(ctx, b"gram_assembly", 1)
    for N in (300, 512):
        a0, a1, a2 = res[0, N], res[1, N], res[2, N]
        assert a1[0] == a2[0] and np.array_equal(a1[1], a2[1])                     # the default chose the Gram form
        assert a0[4] == a2[4] and relerr(a2[0], a0[0]) < 1e-11 and relerr(a2[1], a0[1]) < 1e-9 and relerr(a2[2], a0[2]) < 1e-8
        assert relerr(a2[3], a0[3]) < 1e-10
    g = golden("G8ii_ep_d32_N512")
    assert relerr(res[1, 512][0], g["nlZ"]) < 1e-8 and relerr(res[1, 512][1], g["alpha"]) < 1e-6


def _ard_fit_with(lib, form, x, y, log_ell, log_sn=np.log(0.12), log_sf=0.1, composite=False):
    import pygps_amd as pyGPs
    from pygps_amd import _lib
    ctx = _lib.ctx()
    _lib.check(lib.pgp_set_option(ctx, b"ard_grad_form", form))
    try:
        m = pyGPs.GPR()
        k = pyGPs.cov.RBFard(log_ell_list=[float(v) for v in log_ell], log_sigma=float(log_sf))
        if composite:                  # a device program with an ARD leaf (csrc/grad.hip hadamard_prog_kernel<1>)
            k = k + pyGPs.cov.Matern(np.log(3.0), 3, -0.5)
        m.setPrior(kernel=k)
        m.setNoise(float(log_sn))
        m.setData(x, y)
        nlZ, dnlZ, post = m.getPosterior()
        return nlZ, _flat(dnlZ), np.array(post.alpha), m
    finally:
        lib.pgp_set_option(ctx, b"ard_grad_form", 0)


def test_ard_gradient_weights_gram_vs_difference_form(lib):
    """The ARD gradient pass runs on the matrix cores in forms that cancel: K recomputed in the Gram form and the per-coordinate sums
    as R x^2 + x (C x - 2 W'x) on centred coordinates (csrc/grad.hip hadamard_ard_kernel, ard_dim_reduce).  Their error is
    eps |a|^2 of the scaled, centred points instead of the reference's eps (a - b)^2 (Core/cov.py:899-901, :924-931), so the host
    (make_spec) switches to the difference form throughout -- hadamard_reduce_kernel<1> with ard_dim_reduce_diff -- beyond
    |a|^2 = 1e8.
    * ordinary data (d = 100, the G18 recipe): both forms agree <= 1e-10, the default IS the matrix-core form (bit-identical), and
      the difference form alone reproduces the reference's fixture too; the same for a device program with an ARD leaf;
    * points spread over ~2e4 length scales with near-duplicate pairs (the only off-diagonal K entries that are not 0): the default
      must pick the difference form (bit-identical to forced) and match the oracle; the forced matrix-core form is measurably worse."""
    g = golden("G18_fit_rbfard_d100_N1500")
    x, y = g18_inputs(1500, 100, 3)
    r0, r1, r2 = (_ard_fit_with(lib, f, x, y, g["cov_hyp"][:100], g["lik_hyp"][0], g["cov_hyp"][100]) for f in (0, 1, 2))
    assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1])                         # the default chose the Gram-form weights
    assert np.max(np.abs(r2[1] - r1[1])) < 1e-10 * np.max(np.abs(r1[1]))
    gref = np.concatenate([g["dnlZ_mean"], g["dnlZ_cov"], g["dnlZ_lik"]])
    assert np.max(np.abs(r2[1] - gref)) < 1e-7 * np.max(np.abs(gref))
    # far-out data: 40 coordinates, points on a huge scale, every second point a near-copy of its neighbour
    rng = np.random.RandomState(11)
    N, d = 600, 40
    x = rng.randn(N, d) * 2.0e4
    x[1::2] = x[0::2] + 0.7 * rng.randn(N // 2, d)
    y = np.sin(x[:, :1] / 2.0e4) + 0.1 * rng.randn(N, 1)
    log_ell = np.log(np.sqrt(d)) + rng.uniform(-0.3, 0.3, d)
    b0, b1, b2 = (_ard_fit_with(lib, f, x, y, log_ell) for f in (0, 1, 2))
    assert b0[0] == b2[0] and np.array_equal(b0[1], b2[1])                         # the default chose the difference form
    m = b0[3]
    # faithful=True: one getDerMatrix per length scale like the reference (the oracle's fast branch is a product form itself)
    ref = O.exact_fit(O.RBFARD, np.array(m.covfunc.hyp), 0, m.likfunc.hyp[0], x, y, m.meanfunc.hyp[0] * np.ones((N, 1)),
                      np.ones((N, 1)), faithful=True)
    want = np.concatenate([ref["dnlZ_mean"], ref["dnlZ_cov"], ref["dnlZ_lik"]])
    e_diff = np.max(np.abs(b0[1] - want)) / np.max(np.abs(want))
    e_gram = np.max(np.abs(b1[1] - want)) / np.max(np.abs(want))
    assert relerr(b0[0], ref["nlZ"]) < 1e-9 and e_diff < 1e-10
    assert e_gram > 100 * e_diff
    # a device program with an ARD leaf: the same switch for its per-coordinate sums (K is the difference form there anyway)
    xo, yo = g18_inputs(700, 40, 9)
    lo = np.log(np.sqrt(40.0)) + rng.uniform(-0.3, 0.3, 40)
    p0, p1, p2 = (_ard_fit_with(lib, f, xo, yo, lo, composite=True) for f in (0, 1, 2))
    assert np.array_equal(p0[1], p1[1]) and np.max(np.abs(p2[1] - p1[1])) < 1e-10 * np.max(np.abs(p1[1]))
    q0, q1, q2 = (_ard_fit_with(lib, f, x, y, log_ell, composite=True) for f in (0, 1, 2))
    assert q0[0] == q2[0] and np.array_equal(q0[1], q2[1])
    ard = slice(1, 41)                 # [mean | 40 length scales, ...]: the product form is visibly off out there, on its own entries
    assert np.max(np.abs(q1[1][ard] - q2[1][ard])) > 1e-9 * np.max(np.abs(q2[1][ard]))
