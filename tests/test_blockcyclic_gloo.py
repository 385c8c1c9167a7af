"""CPU, 2 processes over gloo: the 1-D block-cyclic plan of tests/blockcyclic_plan.py (SURVEY 8(f) row 4 groundwork)
executed with numpy tiles -- owner factors + solves its panel, broadcasts it, every rank updates the panels it owns --
must reproduce LAPACK's Cholesky, and the plan's ownership / message / balance figures must be self-consistent.
The numpy executor below is test infrastructure, not a product path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from blockcyclic_plan import BlockCyclic1D  # noqa: E402


def test_plan_ownership_messages_and_balance():
    plan = BlockCyclic1D(8192, 512, 8)
    assert plan.npanel == 16 and [plan.owner(p) for p in range(16)] == [p % 8 for p in range(16)]
    assert plan.owned(3) == [3, 11] and plan.local_index(11) == 1 and plan.local_cols(3) == 1024
    assert sorted(sum((plan.owned(r) for r in range(8)), [])) == list(range(16))
    st = plan.steps()
    assert st[0].bcast_bytes == (8192 - 512) * 512 * 8 and st[-1].bcast_rows == 0
    for s in st:                                   # every panel right of p is updated exactly once, by its owner
        upd = sorted(j for js in s.updates.values() for j in js)
        assert upd == list(range(s.p + 1, 16))
        assert all(plan.owner(j) == r for r, js in s.updates.items() for j in js)
    tot = sum(plan.flops_per_rank())
    ref = sum(2.0 * 512 * ((8192 - j * 512) * 512 - 0.5 * 512 * 512) * j for j in range(16))   # panel j gets j updates
    assert abs(tot - ref) < 1e-6 * ref
    assert plan.imbalance() < 1.35 and BlockCyclic1D(65536, 512, 8).imbalance() < 1.05
    # a ring broadcast at a realistic 64 GB/s per direction is hidden behind the trailing update alone only for the first half of
    # the sweep of a matrix that needs more than one GPU (updates shrink quadratically, panels linearly); the device code also
    # runs E_p E_p' per step, which grows with p -- the library's wait timers (timings_out[6..9]) are what a hardware run reports
    big = BlockCyclic1D(131072, 512, 8).wire_model()
    assert all(tw < tu for tw, tu in big[: len(big) // 2]) and not all(tw < tu for tw, tu in big[: len(big) * 3 // 4])
    with pytest.raises(ValueError):
        BlockCyclic1D(1000, 512, 2)


from conftest import spawn_with_port  # noqa: E402  (a rendezvous port below the ephemeral range; a lost race is retried)


def _worker(rank, world, port, n, w, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = BlockCyclic1D(n, w, world)
        rng = np.random.RandomState(5)
        G = rng.randn(n, n)
        A = G @ G.T / n + np.eye(n)                            # every rank can build A; it only KEEPS its own panels
        mine = {p: np.tril(A)[:, p * w:(p + 1) * w].copy() for p in plan.owned(rank)}
        sent = 0
        for s in plan.steps():
            p, r0 = s.p, (s.p + 1) * w
            Y = torch.zeros((s.bcast_rows, w), dtype=torch.float64)
            if rank == s.owner:
                blk = mine[p]
                Ld = np.linalg.cholesky(blk[p * w:r0] + np.tril(blk[p * w:r0], -1).T)           # D(p)
                blk[p * w:r0] = Ld
                blk[r0:] = np.linalg.solve(Ld, blk[r0:].T).T                                    # S(p): X L^-T
                Y = torch.from_numpy(blk[r0:].copy())
            if s.bcast_rows:
                dist.broadcast(Y, src=s.owner)                                                  # the panel broadcast
                sent += s.bcast_bytes if rank == s.owner else 0
            Yn = Y.numpy()
            for j in s.updates[rank]:                                                           # TU(p) on owned panels
                rows = slice(j * w - r0, None)
                cols = slice(j * w - r0, (j + 1) * w - r0)
                mine[j][j * w:] -= Yn[rows] @ Yn[cols].T
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), sent=sent, **{"p%d" % p: v for p, v in mine.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,w", [(96, 16), (80, 16)])
def test_blockcyclic_sweep_two_ranks_matches_lapack(tmp_path, n, w):
    spawn_with_port(_worker, lambda port: (2, port, n, w, str(tmp_path)), 2)
    rng = np.random.RandomState(5)
    G = rng.randn(n, n)
    A = G @ G.T / n + np.eye(n)
    L = np.linalg.cholesky(A)
    plan = BlockCyclic1D(n, w, 2)
    got = np.zeros((n, n))
    sent = 0
    for r in range(2):
        z = np.load(tmp_path / ("r%d.npz" % r))
        sent += int(z["sent"])
        for p in plan.owned(r):
            got[:, p * w:(p + 1) * w] = z["p%d" % p]
    got = np.tril(got)
    assert np.max(np.abs(got - L)) < 1e-12
    assert sent == sum(s.bcast_bytes for s in plan.steps())
