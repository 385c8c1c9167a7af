"""ONE exact-GP fit over the GPUs of a node: the host side of csrc/sharded.hip.

No reference counterpart -- ``Exact.evaluate`` (pyGPs/Core/inf.py:353-384) factors on one host.  One process per GPU;
every rank calls with the same model; the factorisation is 1-D block-cyclic over column panels, panels travel by broadcast
and everything else a fit returns (alpha, nlZ, dnlZ) needs two small all-reduces (see the header of csrc/sharded.hip).

``Comm`` binds the library's transport (``pgp_comm``) to a SIDE CHANNEL that only ever carries kilobytes:

* a ``torch.distributed`` process group (backend ``nccl``: the library talks to RCCL itself -- it dlopens the librccl the
  process already uses and enqueues ``ncclBroadcast`` / ``ncclAllReduce`` on its own streams, no host synchronisation
  inside the sweep; torch only carries the 128-byte communicator id.  Backend ``gloo``: the library's host transport, the
  call-backs broadcast / reduce pinned host buffers through torch -- several ranks can share the ONE GPU of a test box), or
* a ``hostgroup.HostGroup`` (sockets, no torch anywhere: ``PYGPS_AMD_NO_TORCH=1`` or simply no initialised process group
  while RANK / WORLD_SIZE are set): the same two transports -- RCCL with the id handed around over the sockets, or the host
  call-backs served by the group, or
* nothing: world size 1.

Besides the sharded fit the communicator serves the HOST collectives of the restart search, the K-fold loop and the
multi-dataset objective (``bcast`` / ``allgather`` / ``allreduce`` below -> ``pgp_comm_bcast_host`` ...: on the RCCL
transport they are ncclBroadcast / ncclAllGather / ncclAllReduce over xGMI) and a group-wide ticket counter.

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


class _TorchSide(object):
    """The side channel over an initialised torch.distributed process group."""

    def __init__(self, dist, group):
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # tickets live in the DEFAULT store whatever the group: two disjoint sub-groups searching at the same sequence number must
        # not share a counter -- the key carries the group's member list
        members = list(range(self.world)) if group is None else [dist.get_global_rank(group, r) for r in range(self.world)]
        self.group_key = "all" if group is None else "g" + "-".join(str(m) for m in members)

    def _global(self, group_rank):
        return group_rank if self.group is None else self.dist.get_global_rank(self.group, group_rank)

    def _dev(self, device):
        import torch
        return torch.device("cuda", device) if self.backend == "nccl" else torch.device("cpu")

    def bcast(self, a, root=0, device=0):
        """a: writable numpy array, in place."""
        import torch
        t = torch.from_numpy(a)
        if self.backend == "nccl":
            t = t.to(self._dev(device))
        self.dist.broadcast(t, src=self._global(root), group=self.group)
        if self.backend == "nccl":
            a[...] = t.cpu().numpy()
        return a

    def allreduce(self, a, op="sum", device=0):
        import torch
        t = torch.from_numpy(a)
        if self.backend == "nccl":
            t = t.to(self._dev(device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op in ("max", 1) else self.dist.ReduceOp.SUM, group=self.group)
        if self.backend == "nccl":
            a[...] = t.cpu().numpy()
        return a

    def ticket(self, name):
        # the default store of the process group is a key-value server with an atomic add
        from torch.distributed import distributed_c10d as c10d
        return int(c10d._get_default_store().add("pygps_amd/ticket/%s/%s" % (self.group_key, name), 1)) - 1


def _find_side(group):
    """group: None (auto), a torch process group, or a hostgroup.HostGroup."""
    from .hostgroup import HostGroup
    if isinstance(group, HostGroup):
        return group
    no_torch = bool(os.environ.get("PYGPS_AMD_NO_TORCH"))
    if not no_torch:
        import sys
        if group is not None or "torch" in sys.modules:
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    return _TorchSide(dist, group)
            except Exception:
                pass
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        global _env_group
        if _env_group is None:
            _env_group = HostGroup.from_env()
        return _env_group
    return None


_env_group = None


class Comm(object):
    """A ``pgp_comm`` bound to a side channel (module docstring).  ``device="host"``: a communicator for the host
    collectives only -- no GPU is touched (CPU-side tests of the sharded searches)."""

    def __init__(self, group=None, device=None, transport=None):
        self.lib = _lib.load()
        self.group = group
        self.side = _find_side(group)
        self.rank = self.side.rank if self.side else 0
        self.world = self.side.world if self.side else 1
        self.host_only = device == "host"
        if self.host_only:
            self.device, self.ctx = None, None
        else:
            self.device = _lib.default_device() if device is None else int(device)
            self.ctx = _lib.ctx(self.device)
        torch_backend = getattr(self.side, "backend", None)
        self.dist = getattr(self.side, "dist", None)
        if transport is None:
            transport = os.environ.get("PYGPS_AMD_TRANSPORT") or None
        if transport is None:
            if self.host_only or torch_backend not in (None, "nccl"):
                transport = "host"
            elif torch_backend == "nccl":
                transport = "rccl"
            else:                                                       # HostGroup / nothing: RCCL when every rank has its own GPU
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
                transport = "rccl" if self.lib.pgp_device_count() >= local_world else "host"
        if transport == "rccl" and self.host_only:
            raise ValueError("the RCCL transport needs a device")
        self.transport = transport
        h = C.c_void_p()
        if transport == "rccl":
            path = os.environ.get("PYGPS_AMD_RCCL_PATH", "")            # an explicit librccl (tests: tests/stub_rccl, the RCCL branch on one GPU)
            if path:
                pass
            elif self.dist is not None or (not os.environ.get("PYGPS_AMD_NO_TORCH") and "torch" in __import__("sys").modules):
                torch = _lib.want_torch()                                   # one copy of librccl per process: the one torch loaded
                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                path = cand if os.path.exists(cand) else ""
            self._rccl_path = path.encode()
            ident = np.zeros(128, dtype=np.uint8)
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                _lib.check(self.lib.pgp_comm_unique_id(self._rccl_path, buf), "pgp_comm_unique_id")
                ident[:] = np.frombuffer(buf.raw, dtype=np.uint8)
            if self.side is not None and self.world > 1:                # the id to everybody: the side channel's only job here
                if isinstance(self.side, _TorchSide):
                    self.side.bcast(ident, 0, self.device)
                else:
                    ident = np.asarray(self.side.bcast(ident, 0), dtype=np.uint8)
            ident = C.create_string_buffer(ident.tobytes(), 128)
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

    # ---- host transport: the library hands over (pinned) host buffers -------------------------------------------------
    def _host_bcast(self, user, buf, nbytes, root):
        try:
            a = np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(buf))
            if isinstance(self.side, _TorchSide):
                import torch
                self.side.dist.broadcast(torch.from_numpy(a), src=self.side._global(root), group=self.side.group)
            else:
                a[...] = self.side.bcast(a, root)
            return 0
        except Exception:                                              # no exception may cross the C ABI
            import traceback
            traceback.print_exc()
            return 1

    def _host_allreduce(self, user, buf, count, op):
        try:
            a = np.ctypeslib.as_array(buf, shape=(count,))
            if isinstance(self.side, _TorchSide):
                import torch
                d = self.side.dist
                d.all_reduce(torch.from_numpy(a), op=d.ReduceOp.MAX if op else d.ReduceOp.SUM, group=self.side.group)
            else:
                self.side.allreduce(a, "max" if op else "sum")
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    # ---- host collectives through the library (pgp_comm_*_host) -------------------------------------------------------
    def bcast(self, arr, root=0):
        """Broadcast a float64 array from ``root``; returns the array (a C-contiguous copy when the input was not one)."""
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if a is arr and not a.flags.writeable:
            a = a.copy()
        _lib.check(self.lib.pgp_comm_bcast_host(self.handle, _lib.ptr(a), a.size, int(root)), "pgp_comm_bcast_host")
        return a

    def allreduce(self, arr, op="sum"):
        a = np.array(arr, dtype=np.float64, order="C")
        _lib.check(self.lib.pgp_comm_allreduce_host(self.handle, _lib.ptr(a), a.size, 1 if op in ("max", 1) else 0),
                   "pgp_comm_allreduce_host")
        return a

    def allgather(self, arr):
        """(world,) + arr.shape: rank r's array at [r]."""
        a = np.ascontiguousarray(arr, dtype=np.float64)
        out = np.empty((self.world,) + a.shape)
        _lib.check(self.lib.pgp_comm_allgather_host(self.handle, _lib.ptr(a), a.size, _lib.ptr(out)), "pgp_comm_allgather_host")
        return out

    def ticket(self, name="default"):
        """Next value of a group-wide counter, or None when the side channel has none (then work is dealt statically)."""
        if self.side is None:
            if not hasattr(self, "_tickets"):
                self._tickets = {}
            v = self._tickets.get(name, 0)
            self._tickets[name] = v + 1
            return v
        try:
            return self.side.ticket(name)
        except Exception:
            return None

    def __deepcopy__(self, memo):
        return self                     # a communicator is shared, never copied (models that reference one are deep-copied per fit stream)

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


_search_comms = {}


def search_comm(group=None):
    """The communicator of the restart / fold searches for ``group`` (None: the initialised torch.distributed default group,
    else the launcher's environment, else world size 1): one per (process, group), created on first use.  Without a visible GPU
    (the CPU-side tests of the search logic) it is a host-only communicator."""
    side = _find_side(group)
    if isinstance(side, _TorchSide):
        from torch.distributed import distributed_c10d as c10d
        key = ("torch", id(group if group is not None else c10d._get_default_group()))
    elif side is not None:
        key = ("hostgroup", id(side))
    else:
        key = ("single",)
    c = _search_comms.get(key)
    if c is None or not c.handle:
        lib = _lib.load()
        c = Comm(group=side if key[0] == "hostgroup" else group, device=None if lib.pgp_device_count() > 0 else "host")
        _search_comms[key] = c
    return c


def exact_fit(comm, kind, para, flags, cov_hyp, log_sn, m, dm, nm, n, nargout=3, gather_factor=False, keep_factor=True):
    """pgp_sharded_exact_fit on the data the context of ``comm`` holds.  Returns (alpha (n,), nlZ, g (nm+nc+1,), ms (10,), L, h).
    ms: stage times (assembly, sweep, epilogue, total) in ms, the device bytes the call held at its peak and the bytes the
    posterior handle keeps, then the multi-rank timers: ms the compute stream stalled for a panel, ms of the broadcasts
    (enqueue -> complete, summed), bytes moved in them, the slowest single broadcast (ms).  gather_factor: L = the (n,n) upper factor post.L on every rank (each rank fetches its own columns,
    the host arrays are summed over the ranks) -- for moderate n only: it is n^2 doubles on every host; None otherwise.
    h: the rank's part of the distributed posterior (a ``pgp_sfactor`` handle) when keep_factor, else None."""
    hyp = _lib.f64(np.asarray(cov_hyp, dtype=float))
    nc = len(hyp)
    alpha = np.empty(n)
    nlZ = np.zeros(1)
    g = np.zeros(nm + nc + 1)
    ms = np.zeros(10)
    L = np.zeros((n, n)) if gather_factor else None
    h = C.c_void_p()
    rc = comm.lib.pgp_sharded_exact_fit(comm.ctx, comm.handle, int(kind), _lib.ptr(hyp), nc, int(para), int(flags),
                                        float(log_sn), _lib.ptr(m), _lib.ptr(dm), int(nm), int(min(max(nargout, 1), 3)),
                                        _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), _lib.ptr(ms), _lib.ptr(L),
                                        C.byref(h) if keep_factor else None)
    _lib.check(rc, "pgp_sharded_exact_fit")
    if L is not None and comm.world > 1:
        L = comm.allreduce(L)                        # every rank holds its own columns, zeros elsewhere
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
