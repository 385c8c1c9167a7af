"""CPU: host-side logic of the drop-in (no GPU): the CG minimiser against trajectories recorded from the
reference's minimize.py, the optimiser bookkeeping, likelihood scalar maps against the oracle, the
argument checks, and the loud failure when no GPU / library is available."""
import numpy as np
import pytest

from conftest import golden
from oracle import gp_oracle as O


def rosen(v):
    a, b = v[:-1], v[1:]
    f = np.sum(100.0 * (b - a ** 2) ** 2 + (1 - a) ** 2)
    g = np.zeros_like(v)
    g[:-1] += -400.0 * a * (b - a ** 2) - 2 * (1 - a)
    g[1:] += 200.0 * (b - a ** 2)
    return f, g


def test_minimize_reproduces_reference_trajectories_bit_for_bit():
    from pygps_amd import minimize
    g = golden("Gmin_minimize_trajectories")
    for tag in "abc":
        calls = [0]

        def f(v):
            calls[0] += 1
            return rosen(v)
        X, fX, i = minimize.run(f, g[tag + "_x0"].copy(), length=int(g[tag + "_length"]))
        assert i == int(g[tag + "_i"]) and calls[0] == int(g[tag + "_calls"])
        assert np.array_equal(X, g[tag + "_X"]) and np.array_equal(np.array(fX), g[tag + "_fX"])
    calls = [0]

    def flaky(v):
        calls[0] += 1
        if np.abs(v).max() > 2.2:            # fails far out: only ever hit while extrapolating
            raise ValueError("boom")
        return rosen(v)
    X, fX, i = minimize.run(flaky, np.array([2.0, -1.5, 0.7]), length=-40)
    assert np.array_equal(X, g["d_X"]) and np.array_equal(np.array(fX), g["d_fX"]) and i == int(g["d_i"])
    assert calls[0] == int(g["d_calls"])
    calls = [0]

    def nanny(v):
        calls[0] += 1
        f, gr = rosen(v)
        return (np.nan, gr) if calls[0] == 4 else (f, gr)
    assert bool(g["e_is_none"]) and minimize.run(nanny, np.array([-1.2, 1.0]), length=20) is None


class _FakeModel(object):
    """A model whose objective is a cheap numpy function: exercises Minimize / ShardedMinimize
    bookkeeping (restart order, RNG order, failure accounting) without a GPU."""

    class _H(object):
        def __init__(self, hyp):
            self.hyp = list(hyp)

    class _D(object):
        def __init__(self, g):
            self.mean, self.cov, self.lik = [g[0]], list(g[1:3]), [g[3]]

    def __init__(self, fail_at=()):
        self.meanfunc, self.covfunc, self.likfunc = self._H([0.1]), self._H([0.2, -0.3]), self._H([0.4])
        self.x = np.zeros((4, 1)); self.y = np.zeros((4, 1))
        self.calls = 0
        self.fail_at = set(fail_at)
        self.starts = []

    def getPosterior(self, der=True):
        self.calls += 1
        v = np.array(self.meanfunc.hyp + self.covfunc.hyp + self.likfunc.hyp)
        if any(abs(v[0] - s) < 1e-12 for s in self.fail_at):
            raise np.linalg.LinAlgError("not PD")
        f, g = rosen(v)
        return f + np.sum(np.cos(3 * v)), self._D(g - 3 * np.sin(3 * v)), None


def _conf(model, R):
    from pygps_amd import conf
    c = conf.random_init_conf(model.meanfunc, model.covfunc, model.likfunc)
    c.num_restarts = R
    return c


def test_minimize_restarts_rng_order_and_selection():
    from pygps_amd import opt
    m1 = _FakeModel()
    o1 = opt.Minimize(m1, _conf(m1, 6))
    np.random.seed(7)
    h1, f1 = o1.findMin(m1.x, m1.y, numIters=15)
    assert o1.trailsCounter == 6 and o1.errorCounter == 0                # total runs = num_restarts
    # the sharded optimiser (no process group -> single rank) must visit the same starts and pick the same optimum
    m2 = _FakeModel()
    o2 = opt.ShardedMinimize(m2, _conf(m2, 6))
    np.random.seed(7)
    h2, f2 = o2.findMin(m2.x, m2.y, numIters=15)
    assert f1 == f2 and np.array_equal(h1, h2)
    assert len(o2.runs) == 6 and min(r.f for r in o2.runs) == f2
    # replay the RNG: restart-major, hyp-minor uniform(-5, 5) draws
    np.random.seed(7)
    table = np.array([[np.random.uniform(-5, 5) for _ in range(4)] for _ in range(5)])
    assert table.shape == (5, 4)


def test_sharded_minimize_threshold_only_equals_the_sequential_loop():
    """min_threshold without num_restarts (Core/opt.py:322-327): the sharded optimiser runs waves of restarts and must
    return the optimum the sequential loop stops at, from the same RNG stream."""
    from pygps_amd import opt
    for thr in (40.0, 8.0, 3.2):
        m1 = _FakeModel()
        c1 = _conf(m1, None)
        c1.min_threshold = thr
        o1 = opt.Minimize(m1, c1)
        np.random.seed(11)
        h1, f1 = o1.findMin(m1.x, m1.y, numIters=15)
        assert f1 <= thr
        for streams in (1, 2, 3):
            m2 = _FakeModel()
            c2 = _conf(m2, None)
            c2.min_threshold = thr
            o2 = opt.ShardedMinimize(m2, c2, streams_per_gpu=streams)
            np.random.seed(11)
            h2, f2 = o2.findMin(m2.x, m2.y, numIters=15)
            assert f2 == f1 and np.array_equal(h1, h2), (thr, streams)
            assert len(o2.runs) == o1.trailsCounter and o2.init_table.shape == (o1.trailsCounter, 4)
            assert c2.num_restarts is None and c2.min_threshold == thr           # the configuration is left as it was
    m = _FakeModel()
    with pytest.raises(Exception, match="at least one of the stop conditions"):
        opt.ShardedMinimize(m, _conf(m, None)).findMin(m.x, m.y, numIters=5)


def test_sharded_minimize_restarts_start_cold():
    """A restart's result must not depend on which restart ran before it on the same model object: warm-start
    state of the inference method is dropped at the beginning of every restart."""
    from pygps_amd import opt

    class _Inf(object):
        last_ttau = "warm"
        last_tnu = "warm"
    m = _FakeModel()
    m.inffunc = _Inf()
    seen = []
    orig = m.getPosterior

    def spy(der=True):
        seen.append(m.inffunc.last_ttau)
        m.inffunc.last_ttau = m.inffunc.last_tnu = "warm"                        # what EP.evaluate leaves behind
        return orig(der)
    m.getPosterior = spy
    o = opt.ShardedMinimize(m, _conf(m, 3), streams_per_gpu=1)
    np.random.seed(3)
    o.findMin(m.x, m.y, numIters=4)
    assert seen.count(None) == 3 and seen[0] is None                              # one cold start per restart


def test_minimize_failure_accounting():
    from pygps_amd import opt
    m = _FakeModel()
    c = _conf(m, 4)
    c.meanRange = [(2.0, 2.0)]                                           # every random restart starts at hyp[0] == 2.0 ...
    m.fail_at = {2.0}                                                     # ... where the objective raises
    o = opt.Minimize(m, c)
    with pytest.raises(Exception, match="Over half of the trails failed"):
        o.findMin(m.x, m.y, numIters=5)
    with pytest.raises(Exception, match="not consistent"):
        c.covRange = [(-1, 1)]
    o = opt.Minimize(_FakeModel(fail_at={0.1}), None)                    # no searchConfig: a failing first run is fatal
    with pytest.raises(Exception, match="Can not learn hyperparamters"):
        o.findMin(None, None, numIters=5)


def test_likelihood_scalar_maps_match_oracle():
    from pygps_amd import inf, lik
    rng = np.random.RandomState(0)
    y = np.sign(rng.randn(200, 1)); mu = 4 * rng.randn(200, 1); s2 = rng.rand(200, 1) * 3
    mu[:5] = [[-40.], [-8.], [-6.1], [-5.7], [-5.2]]; y[:5] = 1; s2[:5] = 0.01         # asymptotic branches
    got = lik.Erf().evaluate(y, mu, s2, inf.EP(), None, 3)
    ref = O.erf_ep_moments(y, mu, s2, 3)
    for a, b in zip(got, ref):
        assert np.allclose(a, b, rtol=1e-14, atol=0)
    lp, ymu, ys2 = lik.Erf().evaluate(None, mu, s2, None, None, 3)
    p = np.exp(O.erf_ep_moments(np.ones_like(mu), mu, s2, 1)[0])
    assert np.allclose(ymu, 2 * p - 1) and np.allclose(ys2, 4 * p * (1 - p))
    g = lik.Gauss(np.log(0.3))
    lp, ymu, ys2 = g.evaluate(None, mu, s2, None, None, 3)
    assert np.allclose(ys2, s2 + 0.09) and np.array_equal(ymu, mu)
    lp0 = g.evaluate(y, mu, np.zeros_like(mu), None, None, 1)
    assert np.allclose(lp0, -(y - mu) ** 2 / 0.09 / 2 - np.log(2 * np.pi * 0.09) / 2)


def test_model_facade_host_rules():
    import pygps_amd as pyGPs
    m = pyGPs.GPR()
    assert isinstance(m.meanfunc, pyGPs.mean.Zero) and m.covfunc.hyp == [0., 0.]
    assert abs(m.likfunc.hyp[0] - np.log(0.1)) < 1e-15
    x = np.arange(6.0); y = np.array([1., 2., 3., 4., 5., 9.])
    m.setData(x, y)                                                      # 1-d -> (n,1); default mean -> Const(mean y) (Q8)
    assert m.x.shape == (6, 1) and m.y.shape == (6, 1) and isinstance(m.meanfunc, pyGPs.mean.Const)
    assert m.meanfunc.hyp == [4.0]
    m.setPrior(mean=pyGPs.mean.Zero())
    m.setData(x, y)
    assert isinstance(m.meanfunc, pyGPs.mean.Zero)                       # a user-set mean is kept
    with pytest.raises(AssertionError):
        m.setPrior(kernel="rbf")
    with pytest.raises(Exception, match="not set correctly"):
        m.setOptimizer("SCG")
    m.setOptimizer("Minimize", num_restarts=3, covRange=[(-1, 1), (-2, 2)])
    assert m.optimizer.searchConfig.covRange == [(-1, 1), (-2, 2)] and m.optimizer.searchConfig.num_restarts == 3
    k = pyGPs.cov.RBFard(D=3)
    assert k.hyp == [0., 0., 0., 0.]
    c = pyGPs.GPC()
    with pytest.raises(Exception, match="labels different from"):
        c.getPosterior(np.zeros((3, 1)), np.array([0., 1., 2.]))
    d = pyGPs.inf.dnlZStruct(pyGPs.mean.Const(1.), pyGPs.cov.RBF(), pyGPs.lik.Gauss())
    d.mean, d.cov, d.lik = [1.], [2., 3.], [4.]
    d.accumulateDnlZ(d)
    assert d.cov == [4., 6.]


def test_no_gpu_fails_loudly_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pygps_amd as pyGPs
    with pytest.raises(RuntimeError, match="no CPU fallback|cannot initialise"):
        pyGPs.GPR().getPosterior(np.zeros((3, 1)), np.zeros(3))
    with pytest.raises(RuntimeError):
        pyGPs.cov.RBF().getCovMatrix(x=np.zeros((3, 1)), mode="train")


def test_G13_mean_composites_match_the_reference():
    """Sum / Product / Scale / Power of mean functions (Core/mean.py:140-276), values recorded from the reference."""
    from conftest import golden
    from pygps_amd import mean
    g = golden("G13_mean_composites")
    x = g["x"]
    ms = {"sum": mean.Linear(alpha_list=[0.3, -0.2, 0.7]) + mean.Const(1.5),
          "prod": mean.Linear(alpha_list=[0.3, 0.2, 0.7]) * mean.Const(1.5),
          "scale": mean.Linear(alpha_list=[0.3, 0.2, 0.7]) * 2.5,
          "power": (mean.Linear(alpha_list=[0.3, 0.2, 0.7]) + mean.One()) ** 3,
          "tree": (mean.Linear(alpha_list=[0.3, 0.2, 0.7]) * 0.5 + mean.Const(0.4)) * mean.One() + mean.Zero()}
    for nm, m in ms.items():
        assert list(np.asarray(m.hyp, float)) == list(g[nm + "_hyp"]), nm
        np.testing.assert_allclose(m.getMean(x), g[nm + "_m"], rtol=1e-14)
        for i in range(len(m.hyp)):
            np.testing.assert_allclose(m.getDerMatrix(x, i), g["%s_dm%d" % (nm, i)], rtol=1e-13, atol=1e-15)
        h = [v + 0.01 for v in m.hyp]
        m.hyp = h
        assert list(m.hyp) == h                       # setter reaches the children


def test_bench_cpu_baselines_of_the_other_configs_are_labelled_extrapolations():
    """bench.py's CPU baselines for cfg 3 / 4 / 5: bounded oracle samples scaled by a stated model -- each object
    says so (`extrapolated`, the sample, the reference's own recorded wall time) and carries the contract's keys."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline_other_configs(0.021)
    assert set(out) == {"cfg3", "cfg4", "cfg5"}
    for k, v in out.items():
        assert "error" not in v, (k, v)
        assert v["unit"] == "fits/s" and v["kind"] == "port" and v["extrapolated"] is True and v["value"] > 0
        assert "EXTRAPOLATED" in v["sample"] or "derived" in v["sample"]
    assert "2969 s" in out["cfg3"]["reference_recorded"] and "43 min" in out["cfg5"]["reference_recorded"]
    assert out["cfg4"]["value"] == 0.021


def test_bench_roofline_head_order_is_what_the_driver_record_keeps():
    """bench.flatten_roofline: the contract members and the north-star scalars lead `roofline` in ROOFLINE_HEAD's order (the driver's
    record keeps the leading members only); long strings, `*_what`, the sched-0 diagnostics and nested objects go to the detail."""
    import bench
    roof = {"kernel": "k" * 300, "bound": "mfma", "achieved": 55.0, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.7, "traffic": 8.1e8,
            "traffic_source": "t" * 200, "how": "h" * 200, "launches_per_fit": 31.0, "cholesky_sweep_what": "short", "cholesky_sweep_frac_of_peak": 0.6,
            "frac_sched0": 0.71, "timed_window": {"frac_of_peak": 0.76, "streams": 2}, "factor_inverse_EEt": {"ms": 9.0, "frac_of_peak": 0.7},
            "assembly_full_N16384": {"ms": 0.43, "frac_of_hbm_peak": 0.63, "stores_alone_frac_of_hbm_peak": 0.66},
            "assembly_full_N16384_SEard_d64": {"ms": 0.5, "frac_of_hbm_peak": 0.5, "frac_of_fp64_pipe": 0.8},
            "cholesky_sweep_N16384": {"ms": 47.0, "frac_of_peak": 0.79, "fit_ms": 67.0}, "assembly_fused_frac_of_hbm_peak": 0.4}
    out = {"roofline": roof, "single_stream_ms_per_fit": 10.5, "single_stream_fits_per_s": 95.0, "per_gpu_fits_per_s": 108.0,
           "cfg4_as_written_fits_per_s_per_gpu": 95.0, "cfg4_as_written_fits_per_s": 95.0, "cfg3_seard_N16384_d64": {"fit_ms": 70.0},
           "cfg5_ep_N4096_d32": {"fit_ms": 17.0, "sweeps": 4}, "predict_N8192_ns65536": {"ms": 78.0, "device_ms": 60.0},
           "sharded_fit": {"wait_share": 0.01, "n": 65536}}
    flat, detail = bench.flatten_roofline(out)
    keys = list(flat)
    assert keys[:len(bench.ROOFLINE_HEAD)] == list(bench.ROOFLINE_HEAD) and len(bench.ROOFLINE_HEAD) <= 24
    assert all(flat[k] is not None for k in bench.ROOFLINE_HEAD), [k for k in bench.ROOFLINE_HEAD if flat[k] is None]
    assert flat["frac"] == 0.7 and flat["predict_ns65536_device_ms"] == 60.0 and flat["two_stream_fits_per_s_per_gpu"] == 108.0
    assert all(not isinstance(v, (dict, list)) for v in flat.values()) and all(len(v) <= 120 for v in flat.values() if isinstance(v, str))
    assert "how" not in flat and "frac_sched0" not in flat and "cholesky_sweep_what" not in flat
    assert detail["kernel"] == "k" * 300 and detail["how"] and detail["frac_sched0"] == 0.71 and isinstance(detail["timed_window"], dict)


def test_thread_pools_follow_the_cpu_quota(monkeypatch, tmp_path):
    import os
    """pygps_amd/_threads.py: a container CPU quota below the visible core count caps the BLAS / OpenMP pools (the reference's host code
    around a device call -- numpy.linalg.norm in lik.Gauss.evaluate, Core/lik.py:134-158 -- otherwise wakes one thread per VISIBLE
    core and the cgroup throttles the process); no quota, or PYGPS_AMD_KEEP_THREADS, changes nothing."""
    import builtins
    from pygps_amd import _threads
    real_open = builtins.open

    def fake_open(quota):
        def _open(path, *a, **k):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(quota)
                return real_open(f, *a, **k)
            return real_open(path, *a, **k)
        return _open
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.delenv("PYGPS_AMD_KEEP_THREADS", raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake_open("1600000 100000"))
    assert _threads.cpu_quota() == 16
    assert _threads.respect_cpu_quota() == 16 and os.environ["OPENBLAS_NUM_THREADS"] == "16"
    monkeypatch.setattr(builtins, "open", fake_open("150000 100000"))
    assert _threads.cpu_quota() == 2
    monkeypatch.setattr(builtins, "open", fake_open("max 100000"))
    assert _threads.cpu_quota() is None and _threads.respect_cpu_quota() is None
    monkeypatch.setattr(builtins, "open", fake_open("1600000 100000"))
    monkeypatch.setenv("PYGPS_AMD_KEEP_THREADS", "1")
    assert _threads.respect_cpu_quota() is None
    if _threads._LIMIT is not None:                     # (leave this test process as it was)
        _threads._LIMIT.restore_original_limits()
        _threads._LIMIT = None
