"""CPU, 2 processes over gloo: the restart sharding of ShardedMinimize (broadcast of the init table and data,
per-rank share, all-gather, selection rule) must give every rank exactly what the sequential Minimize gives.
The objective here is a numpy stand-in (tests/test_host_logic._FakeModel); the collective code path is the
one that runs over RCCL (backend nccl) on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


from conftest import spawn_with_port  # noqa: E402  (a rendezvous port below the ephemeral range; a lost race is retried)


def _worker(rank, world, port, R, out_dir, streams):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_host_logic import _FakeModel, _conf
        from pygps_amd import opt
        m = _FakeModel()
        if rank != 0:                       # only rank 0 holds the real data / RNG state; the others get it by broadcast
            m.x = np.full_like(m.x, -7.0)
            m.y = np.full_like(m.y, -7.0)
            np.random.seed(999)
        else:
            m.x = np.arange(4.0).reshape(4, 1)
            m.y = np.arange(4.0).reshape(4, 1) * 2
            np.random.seed(7)
        o = opt.ShardedMinimize(m, _conf(m, R), streams_per_gpu=streams)
        h, f = o.findMin(m.x, m.y, numIters=15)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), h=h, f=f, x=m.x, y=m.y, calls=m.calls,
                 runs_f=np.array([r.f for r in o.runs]), runs_nls=np.array([r.nls for r in o.runs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,streams", [(6, 1), (5, 1), (7, 2)])
def test_sharded_minimize_two_ranks_equals_sequential(tmp_path, R, streams):
    from test_host_logic import _FakeModel, _conf
    from pygps_amd import opt
    m = _FakeModel()
    o = opt.Minimize(m, _conf(m, R))
    np.random.seed(7)
    h_seq, f_seq = o.findMin(m.x, m.y, numIters=15)
    spawn_with_port(_worker, lambda port: (2, port, R, str(tmp_path), streams), 2)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    for r in (r0, r1):
        assert float(r["f"]) == f_seq and np.array_equal(r["h"], h_seq)          # same optimum on every rank
        assert np.array_equal(r["x"], np.arange(4.0).reshape(4, 1))               # data arrived by broadcast
        assert np.array_equal(r["runs_f"], r0["runs_f"]) and len(r["runs_f"]) == R
    # the work really was shared: each rank evaluated only its restarts (with several fit streams per rank the
    # evaluations happen on per-thread deep copies of the model, so they are not counted on the original)
    if streams == 1:
        assert int(r0["calls"]) + int(r1["calls"]) == m.calls
        assert 0 < int(r1["calls"]) < m.calls


def test_sharded_minimize_world_8_one_restart_per_rank(tmp_path):
    """BASELINE configs[3] as written: 8 restarts, 8 ranks, restart r on rank r (t % world with world == R)."""
    from test_host_logic import _FakeModel, _conf
    from pygps_amd import opt
    R = 8
    m = _FakeModel()
    o = opt.Minimize(m, _conf(m, R))
    np.random.seed(7)
    h_seq, f_seq = o.findMin(m.x, m.y, numIters=15)
    spawn_with_port(_worker, lambda port: (8, port, R, str(tmp_path), 1), 8)
    rs = [np.load(tmp_path / ("r%d.npz" % k)) for k in range(8)]
    for r in rs:
        assert float(r["f"]) == f_seq and np.array_equal(r["h"], h_seq)
        assert np.array_equal(r["runs_f"], rs[0]["runs_f"]) and len(r["runs_f"]) == R
        assert np.array_equal(r["x"], np.arange(4.0).reshape(4, 1))
    assert sum(int(r["calls"]) for r in rs) == m.calls and all(int(r["calls"]) > 0 for r in rs)
