"""Device executor of SURVEY 8(f) row 4: ONE Cholesky factorisation spread over the GPUs of a node.

No reference counterpart (pyGPs factors on one host: Core/tools.py:31-77 `jitchol`).  The layout, ownership and schedule
are `multigpu_plan.BlockCyclic1D` (1-D block-cyclic over column panels of w = 512 columns, one process per GPU); this module
executes that plan on the device:

    owner(p):   pgp_dev_panel_factor   -- D(p) + S(p) of the single-GPU sweep on the owner's column panel (csrc/capi.hip)
                broadcast of the solved panel Y_p (rows >= (p + 1) w) with torch.distributed: backend "nccl" = RCCL over xGMI;
                "gloo" stages through host memory (self-test on a box with fewer GPUs than ranks)
    all ranks:  pgp_dev_panel_update   -- C_j -= Y_p Y_p[j]' on the fp64-MFMA GEMM for every OWNED panel j > p,
                the next panel first (it is the next owner's look-ahead target)

Look-ahead (depth 1): in step p the owner of panel p+1 updates that panel first, factors it and posts its broadcast
(`async_op`), so Y_{p+1} travels while every rank works through the rest of step p's trailing updates; two receive
buffers alternate.  A non-positive pivot is marked in the broadcast panel (the last panel: a final status all-reduce), so
every rank raises `LinAlgError` at the same step instead of waiting in a collective.

Status: tested against LAPACK (world 1 over RCCL, world 2 sharing one GPU over gloo, non-PD input at both); timed on ONE
GPU only (N = 32768: 0.27 s = 43.5 TF, the Python-driven panel loop included) -- `bench.py` runs it over all ranks as the
`sharded_cholesky` extra, which is where the first multi-GPU figure comes from.  Product rule as everywhere: no CPU
fallback, the primitives raise without the HIP library.
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
        self._bcast_start(t, src)()

    def _bcast_start(self, t, src):
        """Post the broadcast of `t`; the returned callable blocks the host until `t` holds the data."""
        if self.world == 1 and self.backend is None:
            return lambda: None
        if self.backend == "gloo":                         # self-test transport: through host memory, at wait time
            def wait():
                h = t.cpu()
                self.dist.broadcast(h, src=src, group=self.group)
                t.copy_(h)
                self.torch.cuda.current_stream(self.device).synchronize()
            return wait
        work = self.dist.broadcast(t, src=src, group=self.group, async_op=True)    # RCCL (xGMI ring for world > 1)

        def wait():
            work.wait()                                    # torch's stream waits for the collective ...
            self.torch.cuda.current_stream(self.device).synchronize()              # ... and the host for torch's stream
        return wait

    def _panel_ptr(self, p):
        """Device address of panel p's diagonal block inside the local storage."""
        return self.local.data_ptr() + 8 * (self.plan.local_index(p) * self.w * self.np + p * self.w)

    def _factor_owned(self, p, ybuf):
        """D(p) + S(p) on the owned panel p; the solved rows below go to `ybuf` (contiguous, what the broadcast sends).
        A non-positive pivot does not raise here: it is marked in the buffer (NaN, pivot index) so that EVERY rank sees it
        after the broadcast and leaves the sweep at the same step."""
        torch, w, npd = self.torch, self.w, self.np
        rc = self.lib.pgp_dev_panel_factor(self.ctx, self._panel_ptr(p), npd, npd - p * w, w)
        if rc < 0:
            _lib.check(rc, "pgp_dev_panel_factor")
        c = self.plan.local_index(p) * w
        if ybuf is not None:
            ybuf.copy_(self.local[c:c + w, (p + 1) * w:])
            if rc > 0:
                ybuf.view(-1)[:2] = torch.tensor([float("nan"), float(p * w + rc)], dtype=torch.float64, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
        return rc

    def factor(self):
        """Right-looking sweep with a depth-1 look-ahead: in step p the owner of panel p+1 updates that panel first,
        factors it and posts its broadcast; the broadcast of Y_{p+1} then travels while every rank (the owner included)
        works through the rest of step p's trailing updates (multigpu_plan: "Look-ahead")."""
        torch, lib, plan, w, npd = self.torch, self.lib, self.plan, self.w, self.np
        steps = plan.steps()
        store = [torch.empty(w * max(npd - w, 1), dtype=torch.float64, device=self.device) for _ in range(2)]

        def ybuf_of(p):                                    # Y_p: column-major below x w, ld = below
            below = steps[p].bcast_rows
            return store[p & 1][:w * below].view(w, below) if below else None

        def update(j, p, y):
            m = npd - j * w
            yptr = y.data_ptr() + 8 * ((j - p - 1) * w)
            _lib.check(lib.pgp_dev_panel_update(self.ctx, self._panel_ptr(j), npd, m, w, yptr, steps[p].bcast_rows, w),
                       "pgp_dev_panel_update")

        def bad_pivot(piv):
            raise np.linalg.LinAlgError("Matrix is not positive definite (pivot %d)" % piv)

        torch.cuda.synchronize(self.device)                 # torch's copies into `local` precede the library's stream
        y = ybuf_of(0)
        rc = self._factor_owned(0, y) if steps[0].owner == self.rank else 0
        if y is None:                                       # a single panel
            if rc > 0:
                bad_pivot(rc)
            return self
        arrived = self._bcast_start(y, steps[0].owner)
        for s in steps:
            p = s.p
            if not s.bcast_rows:
                break
            arrived()                                       # Y_p is here
            head = y.view(-1)[:2].cpu().numpy()
            if head[0] != head[0]:
                _lib.check(lib.pgp_dev_sync(self.ctx), "pgp_dev_sync")
                bad_pivot(int(head[1]))
            mine = list(s.updates[self.rank])
            nxt = p + 1
            ynext = ybuf_of(nxt)
            rc = 0
            if steps[nxt].owner == self.rank:
                update(nxt, p, y)                           # look-ahead target first ...
                mine.remove(nxt)
                rc = self._factor_owned(nxt, ynext)         # ... waits for the stream (my step p-1 updates are behind it)
            else:
                # the buffer Y_{p+1} lands in was Y_{p-1}: the updates that read it must have left the stream
                _lib.check(lib.pgp_dev_sync(self.ctx), "pgp_dev_sync")
            if ynext is not None:
                arrived = self._bcast_start(ynext, steps[nxt].owner)
            elif rc > 0:                                    # the last panel has nothing to send: status by all-reduce below
                pass
            for j in mine:                                  # the rest of step p, queued; it runs under the broadcast
                update(j, p, y)
            y = ynext
            last_rc = rc
        _lib.check(lib.pgp_dev_sync(self.ctx), "pgp_dev_sync")
        # the last panel's pivot status reaches everybody through one small all-reduce
        piv = torch.tensor([float((plan.npanel - 1) * w + last_rc) if last_rc > 0 else 0.0], dtype=torch.float64,
                           device=self.device)
        if not (self.world == 1 and self.backend is None):
            t = piv.cpu() if self.backend == "gloo" else piv
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            piv = t
        if float(piv.item()) > 0:
            bad_pivot(int(piv.item()))
        return self
