"""Worker of the torch-free multi-process tests (tests/test_hostgroup.py, tests/test_gpu_r5.py).  Launched as
    RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p PYGPS_AMD_NO_TORCH=1 python search_worker.py <case> <out_dir> [args]
-- the environment any launcher exports; the collectives go through pygps_amd.hostgroup + the library's pgp_comm_*_host."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                     # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))    # repo root


def fake_search(out, R, streams, deal):
    from test_host_logic import _FakeModel, _conf
    from pygps_amd import opt
    rank = int(os.environ["RANK"])
    m = _FakeModel()
    if rank != 0:
        m.x = np.full_like(m.x, -7.0)
        m.y = np.full_like(m.y, -7.0)
        np.random.seed(999)
    else:
        m.x = np.arange(4.0).reshape(4, 1)
        m.y = np.arange(4.0).reshape(4, 1) * 2
        np.random.seed(7)
    o = opt.ShardedMinimize(m, _conf(m, R), streams_per_gpu=streams, deal=deal)
    h, f = o.findMin(m.x, m.y, numIters=15)
    np.savez(os.path.join(out, "r%d.npz" % rank), h=h, f=f, x=m.x, y=m.y, calls=m.calls, owner=o.owner,
             runs_f=np.array([r.f for r in o.runs]), transport=np.array(o.comm.transport))


def group_ops(out):
    from pygps_amd.hostgroup import HostGroup
    g = HostGroup.from_env()
    r, w = g.rank, g.world
    a = np.arange(5.0) + 10 * r
    b = g.bcast(a.copy(), root=w - 1)
    s = g.allreduce(np.array([1.0 * r, 2.0, np.inf if r == 1 else 0.0]), "sum")
    mx = g.allreduce(np.array([1.0 * r, -3.0]), "max")
    ga = g.allgather(np.array([r, r * r], dtype=float))
    t = [g.ticket("a") for _ in range(3)]
    g.barrier()
    t2 = g.ticket("b")
    allt = g.allgather(np.array(t, dtype=float))
    g.barrier()
    np.savez(os.path.join(out, "r%d.npz" % r), b=b, s=s, mx=mx, ga=ga, allt=allt, t2=t2)
    g.close()


def comm_ops(out):
    """the library's host collectives (pgp_comm_*_host) on a host-only communicator"""
    from pygps_amd import sharded
    c = sharded.search_comm(None)
    r, w = c.rank, c.world
    b = c.bcast(np.arange(6.0).reshape(2, 3) + 100 * r, 0)
    ga = c.allgather(np.array([[r, np.inf if r == 0 else np.nan]]))
    s = c.allreduce(np.array([r + 1.0, 1.0]), "sum")
    mx = c.allreduce(np.array([r + 1.0, -r]), "max")
    np.savez(os.path.join(out, "r%d.npz" % r), b=b, ga=ga, s=s, mx=mx, transport=np.array(c.transport))


def kfold_fake(out, K):
    """sharded_k_fold with a numpy stand-in model: fold -> rank bookkeeping, the one all-gather, metrics"""
    from pygps_amd import valid

    class M(object):
        x = None
        usingDefaultMean = True
        inffunc = None

        def setData(self, x, y):
            self.x, self.y = x, y

        def getPosterior(self, *a, **k):
            return (float(np.sum(self.y)), None, None)

        def predict(self, xs, ys=None):
            ym = xs[:, :1] * 2.0 + np.mean(self.y)
            return ym, np.full_like(ym, 0.5), ym, np.full_like(ym, 0.4), None
    rng = np.random.RandomState(3)
    x = rng.randn(53, 2)
    y = x[:, :1] * 2 + 0.1 * rng.randn(53, 1)
    rank = int(os.environ["RANK"])
    res = valid.sharded_k_fold(lambda: M(), x if rank == 0 else np.zeros_like(x), y if rank == 0 else np.zeros_like(y), K=K,
                               metrics=("RMSE", "NLPD"), streams_per_gpu=2)
    np.savez(os.path.join(out, "r%d.npz" % rank), **res)


def multi_fake(out):
    """opt.multi_dataset_objective with a numpy stand-in model: data set i on rank i % world, one all-reduce"""
    from pygps_amd import opt, inf

    class H(object):
        def __init__(self, hyp):
            self.hyp = list(hyp)

    class M(object):
        def __init__(self):
            self.meanfunc, self.covfunc, self.likfunc = H([0.5]), H([0.1, 0.2]), H([-1.0])
            self.calls = 0

        def setData(self, x, y):
            self.x, self.y = x, y

        def getPosterior(self, der=True):
            self.calls += 1
            nlZ = float(np.sum(self.y ** 2) * (1 + self.covfunc.hyp[0]))
            if not der:
                return nlZ, None
            d = inf.dnlZStruct(self.meanfunc, self.covfunc, self.likfunc)
            d.mean = [np.float64(np.sum(self.y))]; d.cov = [np.float64(np.sum(self.x)), np.float64(nlZ)]; d.lik = [np.float64(1.0)]
            return nlZ, d, None
    rng = np.random.RandomState(4)
    xs = [rng.randn(5 + i, 2) for i in range(5)]
    ys = [rng.randn(5 + i, 1) for i in range(5)]
    m = M()
    tot, dn, each = opt.multi_dataset_objective(np.array([0.3, 0.2]), m, xs, ys, der=True)
    np.savez(os.path.join(out, "r%d.npz" % int(os.environ["RANK"])), tot=tot, g=np.array(dn.mean + dn.cov + dn.lik), each=each, calls=m.calls)


def g9_search(out, N, streams):
    """cfg 4 on the device without torch: G9 data, 8 restarts; rank 0 alone holds the data and the RNG state"""
    import pygps_amd as pyGPs
    from conftest import synth_reg
    rank = int(os.environ["RANK"])
    d = 16
    x, y = synth_reg(N, d)
    m = pyGPs.GPR()
    m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0))
    m.setNoise(np.log(0.1))
    if rank == 0:
        m.setData(x, y)
        np.random.seed(123)
    else:
        m.setData(np.zeros_like(x), np.ones_like(y))
        np.random.seed(999 + rank)
    m.setOptimizer("ShardedMinimize", num_restarts=8)
    m.optimizer.streams_per_gpu = streams
    m.optimize(m.x, m.y, numIterations=40)
    runs = m.optimizer.runs
    np.savez(os.path.join(out, "r%d.npz" % rank), f=np.array([r.f for r in runs]), nls=np.array([r.nls for r in runs]),
             X0=m.optimizer.init_table, hyp=np.array(m.optimizer._convert_to_array()), nlZ=m.nlZ, owner=m.optimizer.owner,
             transport=np.array(m.optimizer.comm.transport))


def kfold_gpu(out):
    """G19: the reference's 10-fold loop on the G6 N = 2048 data, folds sharded over the ranks"""
    import pygps_amd as pyGPs
    from pygps_amd import valid
    from conftest import synth_reg
    rank = int(os.environ["RANK"])
    N, d, K = 2048, 16, 10
    x, y = synth_reg(N, d)

    def make():
        m = pyGPs.GPR()
        m.setPrior(kernel=pyGPs.cov.RBF(np.log(np.sqrt(d)), 0.0))
        m.setNoise(np.log(0.1))
        return m
    res = valid.sharded_k_fold(make, x if rank == 0 else np.zeros_like(x), y if rank == 0 else np.zeros_like(y), K=K)
    np.savez(os.path.join(out, "r%d.npz" % rank), **res)


if __name__ == "__main__":
    case, out = sys.argv[1], sys.argv[2]
    args = [json.loads(a) for a in sys.argv[3:]]
    {"fake_search": fake_search, "group_ops": group_ops, "comm_ops": comm_ops, "kfold_fake": kfold_fake, "g9_search": g9_search,
     "kfold_gpu": kfold_gpu, "multi_fake": multi_fake}[case](out, *args)
    if os.environ.get("PYGPS_AMD_NO_TORCH"):
        assert "torch" not in sys.modules, "torch was imported in a PYGPS_AMD_NO_TORCH process"
    print("worker ok", flush=True)
