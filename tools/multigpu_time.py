"""World-size-1 timing of the block-cyclic executor (pygps_amd/multigpu.py): the per-step overhead of driving the sweep
panel by panel from Python with synchronous steps, against the library's own sweep (pgp_potrf, host matrix in / out)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygps_amd import _lib
from pygps_amd.multigpu import ShardedCholesky
torch.cuda.set_device(0)
for n in [int(a) for a in sys.argv[1:]] or [8192]:
    rng = np.random.RandomState(0)
    G = rng.randn(n, 64)
    A = G @ G.T / 64 + np.eye(n)
    sc = ShardedCholesky(n)
    for rep in range(3):
        sc.load_host(A)
        torch.cuda.synchronize()
        t = time.perf_counter(); sc.factor(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    L = sc.gather_host()
    err = np.abs(L @ L.T - A).max()
    print("N %d: block-cyclic executor (world 1) %.2f ms = %.1f TF, |LL'-A| %.2e" % (n, dt * 1e3, n ** 3 / 3 / dt / 1e12, err))
