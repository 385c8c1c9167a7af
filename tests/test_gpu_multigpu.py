"""SURVEY 8(f) row 4 on the device: the 1-D block-cyclic multi-GPU Cholesky executor (pygps_amd/multigpu.py) against
LAPACK -- world size 1 over RCCL, and world size 2 with both ranks sharing GPU 0 and the panel broadcasts staged through
host memory over gloo (this box has one GPU; the 8-GPU run is the driver's)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spd(n, seed=3):
    rng = np.random.RandomState(seed)
    G = rng.randn(n, n)
    return G @ G.T / n + np.eye(n)


def _worker(rank, world, port, backend, n, w, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from pygps_amd.multigpu import ShardedCholesky
        A = _spd(n)
        sc = ShardedCholesky(n, w=w).load_host(A)
        assert sc.world == world and sc.plan.owned(rank) == list(range(rank, sc.plan.npanel, world))
        sc.factor()
        L = sc.gather_host()
        np.save(os.path.join(out_dir, "L%d.npy" % rank), L)
        # a non-PD matrix: the owner of the failing panel marks the panel it broadcasts (or, for the last panel, the final
        # status all-reduce), so EVERY rank raises at the same step instead of waiting in a broadcast
        for bad in (700, n - 3):
            B = A.copy()
            B[bad, bad] = -1.0
            with pytest.raises(np.linalg.LinAlgError) as ei:
                ShardedCholesky(n, w=w).load_host(B).factor()
            assert "pivot %d" % (bad + 1) in str(ei.value)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,n,w", [(1, "nccl", 2048, 512), (2, "gloo", 2048, 512), (2, "gloo", 1900, 256),
                                                (3, "gloo", 2300, 256)])
def test_block_cyclic_cholesky_on_device(tmp_path, world, backend, n, w):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), backend, n, w, str(tmp_path)), nprocs=world, join=True)
    ref = np.linalg.cholesky(_spd(n))
    for r in range(world):
        L = np.load(os.path.join(str(tmp_path), "L%d.npy" % r))
        assert np.abs(L - ref).max() < 1e-11 * np.abs(ref).max() * n ** 0.5
        assert np.array_equal(np.triu(L, 1), np.zeros_like(L))


def _extra_worker(rank, world, port, backend, n, out_dir):
    import json
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import bench
        out = bench.sharded_cholesky_extra(torch, dist, n, w=256)
        if rank == 0:
            json.dump(out, open(os.path.join(out_dir, "extra.json"), "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo")])
def test_bench_extra_one_factorisation_over_all_ranks(tmp_path, world, backend):
    """bench.py's `sharded_cholesky` extra: device-generated kernel matrix, factor, distributed residual check."""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_extra_worker, args=(world, _free_port(), backend, 2048, str(tmp_path)), nprocs=world, join=True)
    out = json.load(open(os.path.join(str(tmp_path), "extra.json")))
    assert out["world"] == world and out["n"] == 2048 and out["panels"] == 8
    assert out["residual_LLtv_vs_Av"] < 1e-13 and out["seconds"] > 0
