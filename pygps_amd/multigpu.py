"""Device executor of SURVEY 8(f) row 4: ONE Cholesky factorisation spread over the GPUs of a node.

No reference counterpart (pyGPs factors on one host: Core/tools.py:31-77 `jitchol`).  The layout, ownership and schedule
are `multigpu_plan.BlockCyclic1D` (1-D block-cyclic over column panels of w = 512 columns, one process per GPU); this module
executes that plan on the device:

    owner(p):   pgp_dev_panel_factor   -- D(p) + S(p) of the single-GPU sweep on the owner's column panel (csrc/capi.hip)
                broadcast of the solved panel Y_p (rows >= (p + 1) w) with torch.distributed: backend "nccl" = RCCL over xGMI;
                "gloo" stages through host memory (self-test on a box with fewer GPUs than ranks)
    all ranks:  pgp_dev_panel_update   -- C_j -= Y_p Y_p[j]' on the fp64-MFMA GEMM for every OWNED panel j > p,
                the next panel first (it is the next owner's look-ahead target)

Status: correct by construction and tested (world 1 over RCCL, world 2 sharing one GPU over gloo, against LAPACK); the
steps are synchronous (the updates of a step are queued without waiting, but the broadcast of step p+1 is not yet
overlapped with the trailing update of step p -- the plan's look-ahead) and it has never been timed on more than one GPU.  Product rule as everywhere: no CPU fallback, the
primitives raise without the HIP library.
"""
import numpy as np

from . import _lib
from .multigpu_plan import BlockCyclic1D


class ShardedCholesky(object):
    """Lower Cholesky factor of a symmetric positive definite matrix, column panels distributed block-cyclically.

    Every rank passes the same `n`; rank r stores panels r, r + world, ... as one (local_cols x np) float64 torch tensor on
    its GPU (row c of the tensor = column c of the local storage, i.e. column-major with leading dimension np)."""

    def __init__(self, n, w=512, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise RuntimeError("ShardedCholesky needs an MI355X (no CPU fallback)")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.n = int(n)
        self.w = int(w)
        self.np = -(-self.n // self.w) * self.w                  # padded with an identity block, like the single-GPU path
        self.plan = BlockCyclic1D(self.np, self.w, self.world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = _lib.load()
        self.ctx = _lib.ctx(self.device.index or 0)
        self.local = torch.zeros((max(self.plan.local_cols(self.rank), 1), self.np), dtype=torch.float64, device=self.device)

    # ---- data in / out (tests, small problems): the full matrix on every rank's host ---------------------------------
    def load_host(self, A):
        """Scatter the lower triangle of the host matrix A (n x n) into the owned panels (identity on the padding)."""
        A = np.asarray(A, dtype=np.float64)
        assert A.shape == (self.n, self.n)
        Ap = np.eye(self.np)
        Ap[:self.n, :self.n] = np.tril(A)
        w = self.w
        for p in self.plan.owned(self.rank):
            c = self.plan.local_index(p) * w
            blk = np.ascontiguousarray(Ap[:, p * w:(p + 1) * w].T)              # (w x np): row = column of the panel
            self.local[c:c + w].copy_(self.torch.from_numpy(blk))
        return self

    def gather_host(self):
        """The factor L (n x n, lower) on every rank's host."""
        torch, dist = self.torch, self.dist
        L = np.zeros((self.np, self.np))
        w = self.w
        for p in range(self.plan.npanel):
            buf = torch.empty((w, self.np), dtype=torch.float64, device=self.device)
            if self.plan.owner(p) == self.rank:
                c = self.plan.local_index(p) * w
                buf.copy_(self.local[c:c + w])
            self._bcast(buf, self.plan.owner(p))
            L[:, p * w:(p + 1) * w] = buf.cpu().numpy().T
        return np.tril(L)[:self.n, :self.n]

    # ---- the sweep ------------------------------------------------------------------------------------------------
    def _bcast(self, t, src):
        if self.world == 1 and self.backend is None:
            return
        if self.backend == "gloo":                         # self-test transport: through host memory
            h = t.cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:                                              # RCCL (xGMI ring for world > 1)
            self.dist.broadcast(t, src=src, group=self.group)

    def factor(self):
        torch, lib, plan, w, npd = self.torch, self.lib, self.plan, self.w, self.np
        pending = []
        for s in plan.steps():
            p = s.p
            rows = npd - p * w                              # panel p from its diagonal block down
            below = s.bcast_rows                            # = rows - w: what the other panels need of it
            ybuf = torch.empty((w, max(below, 1)), dtype=torch.float64, device=self.device)   # column-major below x w, ld = below
            if s.owner == self.rank:
                c = plan.local_index(p) * w
                torch.cuda.synchronize(self.device)         # torch's copies into `local` precede the library's stream
                ptr = self.local.data_ptr() + 8 * (c * npd + p * w)
                rc = lib.pgp_dev_panel_factor(self.ctx, ptr, npd, rows, w)
                if rc > 0:
                    raise np.linalg.LinAlgError("Matrix is not positive definite (pivot %d)" % (p * w + rc))
                _lib.check(rc, "pgp_dev_panel_factor")
                if below:
                    ybuf.copy_(self.local[c:c + w, (p + 1) * w:])
            if not below:
                break
            self._bcast(ybuf, s.owner)
            torch.cuda.synchronize(self.device)
            for j in s.updates[self.rank]:                  # ascending: the next panel (look-ahead target) first
                cj = plan.local_index(j) * w
                m = npd - j * w
                cptr = self.local.data_ptr() + 8 * (cj * npd + j * w)
                yptr = ybuf.data_ptr() + 8 * ((j - p - 1) * w)
                _lib.check(lib.pgp_dev_panel_update(self.ctx, cptr, npd, m, w, yptr, below, w), "pgp_dev_panel_update")
            # the queued updates read ybuf: it must outlive them (torch's caching allocator would hand the block out again)
            pending.append(ybuf)
            if len(pending) > 1:
                _lib.check(lib.pgp_dev_sync(self.ctx), "pgp_dev_sync")
                del pending[:-1]
        _lib.check(lib.pgp_dev_sync(self.ctx), "pgp_dev_sync")
        return self
