import sys, os, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, pygps_amd as pyGPs
n5, d5 = 4096, 32
rng = np.random.RandomState(0)
x5 = rng.randn(n5, d5); w5 = rng.randn(d5, 1)
y5 = np.sign(x5 @ w5 / np.sqrt(d5) + 0.3 * rng.randn(n5, 1)); y5[y5 == 0] = 1
def one():
    m5 = pyGPs.GPC(); m5.setPrior(mean=pyGPs.mean.Zero(), kernel=pyGPs.cov.RBF(np.log(np.sqrt(d5)), 0.0))
    return m5.getPosterior(x5, y5)[0]
one(); one()
pr = cProfile.Profile(); pr.enable(); one(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18); print(st.getvalue()[:3500])
