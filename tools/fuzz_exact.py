"""Randomised parity sweep (not a test: uses oracle/ as the checker): exact fits at random sizes / dimensions / kernels
through the C ABI against the CPU oracle -- nlZ, alpha, all gradients."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pygps_amd import _lib
from oracle import gp_oracle as O

lib = _lib.load()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = dict(nlZ=0.0, alpha=0.0, grad=0.0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(ncase):
    n = int(rng.choice([rng.randint(1, 200), rng.randint(200, 1700), rng.randint(1536, 2600)]))
    d = int(rng.randint(1, 20))
    kind = int(rng.choice([O.RBF, O.RBFARD, O.MATERN, O.RQ]))
    para = int(rng.choice([1, 3, 5])) if kind == O.MATERN else 0
    x = rng.randn(n, d) * rng.uniform(0.5, 2.0)
    y = np.sin(x.sum(1, keepdims=True)) + 0.2 * rng.randn(n, 1)
    nh = {O.RBF: 2, O.RBFARD: d + 1, O.MATERN: 2, O.RQ: 3}[kind]
    hyp = rng.uniform(-0.5, 1.0, nh)
    log_sn = float(rng.uniform(-2.5, -0.5))
    m = np.full((n, 1), float(y.mean())); dm = np.ones((1, n))
    ref = O.exact_fit(kind, hyp, para, log_sn, x, y, m, dm=dm.T, faithful=False, matern_reference_compat=False)
    h = C.c_void_p()
    assert lib.pgp_init(0, C.byref(h)) == 0
    xx = np.ascontiguousarray(x); yy = np.ascontiguousarray(y).ravel()
    assert lib.pgp_set_data(h, _lib.ptr(xx), n, d, _lib.ptr(yy)) == 0
    alpha = np.zeros(n); nlZ = np.zeros(1); g = np.zeros(1 + nh + 1)
    mv = np.ascontiguousarray(m).ravel(); dmv = np.ascontiguousarray(dm)
    hv = np.ascontiguousarray(hyp)
    rc = lib.pgp_exact_fit(h, kind, _lib.ptr(hv), nh, para, 0, log_sn, _lib.ptr(mv), _lib.ptr(dmv), 1, 3,
                           _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
    assert rc == 0, (rc, n, d, kind)
    lib.pgp_destroy(h)
    gref = np.concatenate([np.ravel(ref["dnlZ_mean"]), np.ravel(ref["dnlZ_cov"]), np.ravel(ref["dnlZ_lik"])])
    e1 = abs(nlZ[0] - ref["nlZ"]) / max(1.0, abs(ref["nlZ"]))
    e2 = np.abs(alpha - ref["alpha"].ravel()).max() / max(1e-300, np.abs(ref["alpha"]).max())
    e3 = np.abs(g - gref).max() / max(1.0, np.abs(gref).max())
    worst["nlZ"] = max(worst["nlZ"], e1); worst["alpha"] = max(worst["alpha"], e2); worst["grad"] = max(worst["grad"], e3)
    flag = "" if (e1 < 1e-9 and e2 < 1e-7 and e3 < 1e-7) else "   <-- CHECK"
    print("case %2d n %4d d %2d kind %d para %d: nlZ %.1e alpha %.1e grad %.1e%s" % (case, n, d, kind, para, e1, e2, e3, flag), flush=True)
print("worst", worst)
