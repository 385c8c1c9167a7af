"""ONE exact-GP fit over the GPUs of a node: the host side of csrc/sharded.hip.

No reference counterpart -- ``Exact.evaluate`` (pyGPs/Core/inf.py:353-384) factors on one host.  One process per GPU;
every rank calls with the same model; the factorisation is 1-D block-cyclic over column panels, panels travel by broadcast
and everything else a fit returns (alpha, nlZ, dnlZ) needs two small all-reduces (see the header of csrc/sharded.hip).

``Comm`` binds a ``torch.distributed`` process group to the library's transport (``pgp_comm``):

* backend ``nccl``: the library talks to RCCL itself (it dlopens the librccl the process already uses -- the one bundled
  with torch -- and enqueues ``ncclBroadcast`` / ``ncclAllReduce`` on its own streams: no host synchronisation inside the
  sweep).  ``torch.distributed`` only carries the 128-byte communicator id from rank 0 to the others.
* backend ``gloo`` (or any other): host call-backs; the library stages device memory through pinned host memory and this
  module broadcasts / reduces the host buffers with ``torch.distributed``.  That is the self-test transport: several ranks can
  share the ONE GPU of a test box (RCCL refuses two ranks on one device).
* no process group: world size 1 over the host transport (nothing moves).

``inf.Exact(sharded=True)`` (or ``sharded=Comm(...)``) routes ``evaluate`` through it; ``post.L`` is then a
``DistributedFactor``: the factor stays on the ranks (every rank keeps its column panels of L and of L^-T, O(n^2 / world)
bytes), ``GP.predict`` runs on it collectively (``pgp_sharded_predict``: every rank calls with the same test points), and
touching it as an array raises unless ``gather_factor=True`` was asked for.  No CPU fallback.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib

_BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
_ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int)


class Comm(object):
    """The transport of a sharded fit: a ``pgp_comm`` bound to a torch.distributed process group (or to nothing: world 1)."""

    def __init__(self, group=None, device=None, transport=None):
        self.lib = _lib.load()
        self.group = group
        self.device = _lib.default_device() if device is None else int(device)
        self.ctx = _lib.ctx(self.device)
        dist = None
        try:
            import torch.distributed as dist_
            if dist_.is_available() and dist_.is_initialized():
                dist = dist_
        except Exception:
            dist = None
        self.dist = dist
        self.rank = dist.get_rank(group) if dist else 0
        self.world = dist.get_world_size(group) if dist else 1
        backend = dist.get_backend(group) if dist else None
        if transport is None:
            transport = "rccl" if backend == "nccl" else "host"
        self.transport = transport
        h = C.c_void_p()
        if transport == "rccl":
            torch = _lib.want_torch()
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self._rccl_path = (path if os.path.exists(path) else "").encode()
            ident = C.create_string_buffer(128)
            if self.rank == 0:
                _lib.check(self.lib.pgp_comm_unique_id(self._rccl_path, ident), "pgp_comm_unique_id")
            if dist and self.world > 1:                                 # the id to everybody (the only use of torch's collectives)
                dev = torch.device("cuda", self.device) if backend == "nccl" else torch.device("cpu")
                t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).to(dev)
                dist.broadcast(t, src=0, group=group)
                ident = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
            _lib.check(self.lib.pgp_comm_init_rccl(self.ctx, self.world, self.rank, ident, self._rccl_path, C.byref(h)),
                       "pgp_comm_init_rccl")
        elif transport == "host":
            self._cb = (_BCAST(self._host_bcast), _ALLRED(self._host_allreduce))      # kept alive with the object
            _lib.check(self.lib.pgp_comm_init_host(self.ctx, self.world, self.rank, self._cb[0], self._cb[1], None, C.byref(h)),
                       "pgp_comm_init_host")
        else:
            raise ValueError("transport must be 'rccl' or 'host'")
        self.handle = h
        self._free = self.lib.pgp_comm_free

    # ---- host transport: the library hands over pinned host buffers ---------------------------------------------------
    def _host_bcast(self, user, buf, nbytes, root):
        try:
            import torch
            a = np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(buf))
            t = torch.from_numpy(a)
            self.dist.broadcast(t, src=self._global(root), group=self.group)
            return 0
        except Exception:                                              # no exception may cross the C ABI
            import traceback
            traceback.print_exc()
            return 1

    def _host_allreduce(self, user, buf, count, op):
        try:
            import torch
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op else self.dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _global(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self._free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_comm = {}


def default_comm(device=None):
    """One Comm per (process, device) over the default process group, created on first use."""
    device = _lib.default_device() if device is None else int(device)
    if device not in _default_comm:
        _default_comm[device] = Comm(device=device)
    return _default_comm[device]


def exact_fit(comm, kind, para, flags, cov_hyp, log_sn, m, dm, nm, n, nargout=3, gather_factor=False, keep_factor=True):
    """pgp_sharded_exact_fit on the data the context of ``comm`` holds.  Returns (alpha (n,), nlZ, g (nm+nc+1,), ms (6,), L, h).
    ms: stage times (assembly, sweep, epilogue, total) in ms, then the device bytes the call held at its peak and the bytes the
    posterior handle keeps.  gather_factor: L = the (n,n) upper factor post.L on every rank (each rank fetches its own columns,
    the host arrays are summed over the ranks) -- for moderate n only: it is n^2 doubles on every host; None otherwise.
    h: the rank's part of the distributed posterior (a ``pgp_sfactor`` handle) when keep_factor, else None."""
    hyp = _lib.f64(np.asarray(cov_hyp, dtype=float))
    nc = len(hyp)
    alpha = np.empty(n)
    nlZ = np.zeros(1)
    g = np.zeros(nm + nc + 1)
    ms = np.zeros(6)
    L = np.zeros((n, n)) if gather_factor else None
    h = C.c_void_p()
    rc = comm.lib.pgp_sharded_exact_fit(comm.ctx, comm.handle, int(kind), _lib.ptr(hyp), nc, int(para), int(flags),
                                        float(log_sn), _lib.ptr(m), _lib.ptr(dm), int(nm), int(min(max(nargout, 1), 3)),
                                        _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), _lib.ptr(ms), _lib.ptr(L),
                                        C.byref(h) if keep_factor else None)
    _lib.check(rc, "pgp_sharded_exact_fit")
    if L is not None and comm.world > 1:
        import torch
        t = torch.from_numpy(L)
        if comm.dist.get_backend(comm.group) == "nccl":
            t = t.cuda(comm.device)
            comm.dist.all_reduce(t, group=comm.group)
            L = t.cpu().numpy()
        else:
            comm.dist.all_reduce(t, group=comm.group)
    return alpha, float(nlZ[0]), g, ms, L, (h if keep_factor else None)


class DistributedFactor(object):
    """``post.L`` of a sharded fit: the factor stays distributed over the ranks -- this rank's column panels of L and of
    L^-T, alpha and the scaled coordinates live behind a ``pgp_sfactor`` handle on its GPU.  ``GP.predict`` uses it
    collectively (``predict`` below); touching it as an array raises (``Exact(gather_factor=True)`` gathers it instead)."""

    def __init__(self, n, comm, handle):
        self.shape = (int(n), int(n))
        self.world = int(comm.world)
        self.comm = comm
        self.handle = handle
        self.nbytes_device = int(comm.lib.pgp_sfactor_bytes(handle)) if handle else 0
        if handle:
            self._fin = weakref.finalize(self, DistributedFactor._release, comm.lib.pgp_sfactor_free, comm.ctx, handle)

    @staticmethod
    def _release(free_fn, ctx, handle):
        try:
            free_fn(ctx, handle)
        except Exception:           # interpreter shutdown
            pass

    def predict(self, xs, ms):
        """fmu, fs2 of GP.predict (Core/gp.py:395-417) for test points xs (ns, d) with prior mean ms (ns,): collective, every
        rank calls with the same arguments and receives the same result."""
        if not self.handle:
            raise NotImplementedError("this sharded posterior was computed without a factor handle")
        xs = _lib.f64(xs)
        ns = xs.shape[0]
        ms = _lib.f64(ms).reshape(ns)
        fmu = np.empty(ns)
        fs2 = np.empty(ns)
        _lib.check(self.comm.lib.pgp_sharded_predict(self.comm.ctx, self.comm.handle, self.handle, _lib.ptr(xs), ns, _lib.ptr(ms),
                                                    _lib.ptr(fmu), _lib.ptr(fs2)), "pgp_sharded_predict")
        return fmu.reshape(ns, 1), fs2.reshape(ns, 1)

    def _no(self, *a, **k):
        raise NotImplementedError("the Cholesky factor of a sharded fit is distributed over %d ranks and is not gathered "
                                  "(predict() works on it as it is; Exact(gather_factor=True) returns post.L as an array)"
                                  % self.world)

    __array__ = __getitem__ = _no

    def __deepcopy__(self, memo):
        return self

    def __repr__(self):
        return "DistributedFactor(n=%d over %d ranks, %d device bytes on this rank)" % (self.shape[0], self.world, self.nbytes_device)
