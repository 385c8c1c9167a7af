"""Optimiser plug-ins (reference: pyGPs/Core/opt.py -- Optimizer :35-89, Minimize :273-328).

``Minimize`` is the reference's sequential restart loop around the CG minimiser; every objective
evaluation is one device fit.  ``ShardedMinimize`` runs the SAME restarts one-per-GPU: one process
per GPU, rank 0 draws the random initial points in the reference's RNG order and broadcasts them,
every rank optimises its share with its own GPU, a single all-gather returns (nlZ, hyp,
#line-searches, failed) per restart and every rank applies the reference's selection rule.  No
collective touches the data path.  The three broadcasts and the all-gather go through the library's
own communicator (``sharded.Comm`` -> ``pgp_comm_bcast_host`` / ``pgp_comm_allgather_host``: RCCL
over xGMI, bound by the library itself); what hands the 128-byte RCCL id around is either an
initialised ``torch.distributed`` group or -- no torch anywhere -- the socket group of
``pygps_amd.hostgroup`` (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as any launcher exports them).
"""
import logging
from copy import deepcopy

import numpy as np

from . import minimize


class Optimizer(object):
    def __init__(self, model=None, searchConfig=None):
        self.model = model
        self.logger = logging.getLogger(__name__)

    def findMin(self, x, y, numIters):
        raise NotImplementedError

    # hyper-parameter vector order is the contract: mean | cov | lik   (Core/opt.py:77-89)
    def _convert_to_array(self):
        m = self.model
        return np.array(list(m.meanfunc.hyp) + list(m.covfunc.hyp) + list(m.likfunc.hyp), dtype=float)

    def _apply_in_objects(self, hypInArray):
        m = self.model
        Lm, Lc = len(m.meanfunc.hyp), len(m.covfunc.hyp)
        v = np.asarray(hypInArray).tolist()
        m.meanfunc.hyp = v[:Lm]
        m.covfunc.hyp = v[Lm:Lm + Lc]
        m.likfunc.hyp = v[Lm + Lc:]

    def _nlml(self, hypInArray):
        self._apply_in_objects(hypInArray)
        return self.model.getPosterior(der=False)[0]

    def _dnlml(self, hypInArray):
        return self._nlzAnddnlz(hypInArray)[1]

    def _nlzAnddnlz(self, hypInArray):
        self._apply_in_objects(hypInArray)
        nlZ, dnlZ, post = self.model.getPosterior()
        return nlZ, np.array(dnlZ.mean + dnlZ.cov + dnlZ.lik)


class _Run(object):
    """Outcome of one minimiser run."""
    __slots__ = ("ok", "f", "hyp", "nls")

    def __init__(self, ok=False, f=np.inf, hyp=None, nls=-1):
        self.ok, self.f, self.hyp, self.nls = ok, f, hyp, nls


def _select(runs, num_restarts, logger, trails):
    """The reference's bookkeeping over a sequence of runs in restart order (Core/opt.py:289-327):
    run 0 seeds the incumbent; a later run replaces it on strict '<'; a failed run counts as an
    error -- and so does every later run if run 0 failed, because the incumbent is then undefined
    and the comparison itself raises inside the reference's try-block; more than num_restarts/2
    errors abort."""
    errors = 0
    best = None
    for k, r in enumerate(runs):
        if k == 0:
            if r.ok:
                best = r
            else:
                errors += 1
        else:
            if r.ok and best is not None:
                if r.f < best.f:
                    best = r
            else:
                errors += 1
            if num_restarts and errors > num_restarts / 2:
                logger.warning("[Minimize] %d out of %d trails failed during optimization", errors, trails + k + 1)
                raise Exception("Over half of the trails failed for minimize")
    return best, errors


class Minimize(Optimizer):
    """CG minimiser with optional random restarts, sequential (Core/opt.py:273-328)."""

    def __init__(self, model, searchConfig=None):
        super(Minimize, self).__init__()
        self.model = model
        self.searchConfig = searchConfig
        self.trailsCounter = 0
        self.errorCounter = 0

    def _one(self, hyp0, numIters):
        try:
            out = minimize.run(self._nlzAnddnlz, hyp0, length=numIters)
            run = _Run(True, out[1][-1], deepcopy(out[0]), out[2])       # TypeError if out is None
            self.logger.warning("Number of line searches %g", out[2])
            return run
        except Exception:
            return _Run()

    def findMin(self, x, y, numIters=200):
        hyp = self._convert_to_array()
        first = self._one(hyp, numIters)
        if not first.ok:
            self.errorCounter += 1
            if not self.searchConfig:
                raise Exception("Can not learn hyperparamters using minimize.")
        self.trailsCounter += 1
        if not self.searchConfig:
            return first.hyp, first.f
        cfg = self.searchConfig
        ranges = cfg.meanRange + cfg.covRange + cfg.likRange
        if not (cfg.num_restarts or cfg.min_threshold):
            raise Exception("Specify at least one of the stop conditions")
        best = first if first.ok else None
        while True:
            self.trailsCounter += 1
            for i in range(hyp.shape[0]):                                # global numpy RNG, hyp order
                hyp[i] = np.random.uniform(low=ranges[i][0], high=ranges[i][1])
            r = self._one(hyp, numIters)
            if r.ok and best is not None:
                if r.f < best.f:
                    best = r
            else:
                self.errorCounter += 1
            if cfg.num_restarts and self.errorCounter > cfg.num_restarts / 2:
                self.logger.warning("[Minimize] %d out of %d trails failed during optimization", self.errorCounter,
                                    self.trailsCounter)
                raise Exception("Over half of the trails failed for minimize")
            done = cfg.num_restarts and self.trailsCounter > cfg.num_restarts - 1
            done = done or (cfg.min_threshold and best is not None and best.f <= cfg.min_threshold)
            if done:
                self.logger.warning("[Minimize] %d out of %d trails failed during optimization", self.errorCounter,
                                    self.trailsCounter)
                return best.hyp, best.f


class ShardedMinimize(Minimize):
    """The restart loop of ``Minimize`` sharded over the GPUs of one node, restart r -> rank r % world.

    With ``num_restarts`` the whole table is drawn up front and sharded; with ``min_threshold`` also set, the
    lowest-index restart that reaches it wins, which is what the sequential loop would have returned.  With
    ``min_threshold`` alone the search runs in waves of world x streams_per_gpu restarts (_find_by_threshold).  Without an initialised process group (or world size 1) it
    degenerates to running all restarts locally from the same pre-drawn table, so results do not
    depend on the number of GPUs.
    """

    #: concurrent fit streams per GPU.  A single fit leaves the GPU under-used while its Cholesky panel chain
    #: runs; a second, independent restart on the same GPU (own context, own host thread) fills those gaps:
    #: 10.0 instead of 12.6 ms per fit at N=8192 on MI355X (round 2).  Results do not depend on this number.
    streams_per_gpu = 2

    def __init__(self, model, searchConfig=None, group=None, streams_per_gpu=None, deal="auto"):
        """group: None (the initialised torch.distributed default group if there is one, else the launcher's environment
        through ``hostgroup.HostGroup``, else world size 1), a torch process group, a ``HostGroup`` or a ready
        ``sharded.Comm``.  deal: "static" = restart t on rank t % world; "dynamic" = ranks take restarts from a group-wide
        ticket counter; "auto" = static when there is at most one restart per rank (cfg 4: 8 restarts on 8 GPUs), else
        dynamic -- line searches differ in length and the slowest rank sets the wall time.  Results do not depend on it."""
        super(ShardedMinimize, self).__init__(model, searchConfig)
        self.group = group
        self.comm = None
        self.deal = deal
        self.runs = None                # per-restart records of the last findMin (all ranks)
        self.owner = None               # rank that ran each restart of the last findMin
        if streams_per_gpu is not None:
            self.streams_per_gpu = int(streams_per_gpu)

    def _get_comm(self):
        """The communicator of the search, created on first use and shared by every ShardedMinimize of the process that
        names the same group."""
        if self.comm is None:
            from . import sharded
            self.comm = self.group if isinstance(self.group, sharded.Comm) else sharded.search_comm(self.group)
        return self.comm

    @staticmethod
    def _cold_start(model):
        """A restart must not depend on which restart ran before it on the same model object: inference methods that
        keep warm-start state between evaluations (EP: last_ttau / last_tnu, Core/inf.py:735-741) begin every restart
        cold.  The sequential reference carries that state from one restart into the next; a sharded search cannot
        (the predecessor runs on another GPU), so it defines the per-restart result as the cold-started one -- the same
        on every world size, fit-stream count and timing."""
        inf = getattr(model, "inffunc", None)
        for name in ("last_ttau", "last_tnu"):
            if hasattr(inf, name):
                setattr(inf, name, None)

    def _one(self, hyp0, numIters):
        self._cold_start(self.model)
        return super(ShardedMinimize, self)._one(hyp0, numIters)

    def _run_share(self, mine, table, numIters, take=None):
        """Optimise restarts (indices into table); returns {t: _Run}.  `mine`: this rank's list (static deal); `take`: a
        thread-safe callable that returns the next restart index of the whole group or None (dynamic deal).  With more
        than one fit stream the restarts are dealt to host threads, each with a private deep copy of the model and its
        own device context (pygps_amd._lib.fit_stream); ctypes releases the GIL while a fit runs on the GPU."""
        import threading
        lock = threading.Lock()
        todo = list(mine) if mine is not None else None

        def next_item():
            if take is not None:
                return take()
            with lock:                            # restarts are taken from a shared queue: line searches differ in length, a
                return todo.pop(0) if todo else None   # static deal can leave one stream idle at the end
        S = max(1, int(self.streams_per_gpu))
        if todo is not None:
            S = min(S, max(1, len(todo)))
        out = {}
        if S == 1:
            while True:
                t = next_item()
                if t is None:
                    return out
                out[t] = self._one(table[t].copy(), numIters)
        from copy import deepcopy as _dc
        from . import _lib

        def work(k):
            with _lib.fit_stream(k), _lib.concurrent_fit_streams():
                clone = Minimize(_dc(self.model), None)
                clone.model.optimizer = clone
                clone.logger = self.logger
                while True:                       # results do not depend on the deal: every restart starts cold (_cold_start)
                    t = next_item()
                    if t is None:
                        return
                    self._cold_start(clone.model)
                    out[t] = clone._one(table[t].copy(), numIters)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        [th.start() for th in ths]
        [th.join() for th in ths]
        return out

    #: threshold-only search (Core/opt.py:322-327 with num_restarts unset): give up after this many waves
    max_waves = 64

    def _find_by_threshold(self, x, y, numIters):
        """``min_threshold`` alone (Core/opt.py:301-327 without num_restarts): the reference keeps drawing restarts until
        the incumbent is at or below the threshold.  Here the restarts run in WAVES of world x streams_per_gpu, each wave
        one sharded search (same draw order: restart-major, hyp-minor, continuing the global numpy stream on rank 0);
        the reference's sequential bookkeeping is replayed over the concatenated runs, so the optimum returned is the one
        the sequential loop would have stopped at -- only the restarts after it inside the last wave are extra work."""
        cfg = self.searchConfig
        world = self._get_comm().world
        wave = max(2, world * max(1, int(self.streams_per_gpu)))
        runs_all, tables = [], []
        hyp_keep = self._convert_to_array()
        saved = (cfg.num_restarts, cfg.min_threshold)
        try:
            for w in range(self.max_waves):
                cfg.num_restarts, cfg.min_threshold = wave, None
                self._apply_in_objects(hyp_keep)                # the minimiser leaves the model at its last evaluation
                self._wave_cont = w > 0                         # only the first wave starts with the model's own hypers
                self.runs = None
                try:
                    ShardedMinimize.findMin(self, x, y, numIters)
                except Exception:
                    if self.runs is None:
                        raise
                    # "over half failed" inside one wave: the replay below decides, like the sequential bookkeeping
                finally:
                    self._wave_cont = False
                tables.append(self.init_table)
                runs_all.extend(self.runs)
                inc = None
                for k, r in enumerate(runs_all):                # Core/opt.py:289-327 in restart order
                    if k == 0:
                        inc = r if r.ok else None
                    elif r.ok and inc is not None and r.f < inc.f:
                        inc = r
                    if k >= 1 and inc is not None and inc.f <= saved[1]:
                        self.runs, self.init_table = runs_all[:k + 1], np.concatenate(tables)[:k + 1]
                        return inc.hyp, inc.f
            raise Exception("ShardedMinimize: min_threshold %g not reached in %d restarts" % (saved[1], len(runs_all)))
        finally:
            cfg.num_restarts, cfg.min_threshold = saved

    def findMin(self, x, y, numIters=200):
        cfg = self.searchConfig
        if not cfg or not (cfg.num_restarts or cfg.min_threshold):
            raise Exception("Specify at least one of the stop conditions")       # Core/opt.py:303-304
        if not cfg.num_restarts:
            return self._find_by_threshold(x, y, numIters)
        comm = self._get_comm()
        rank, world = comm.rank, comm.world
        hyp0 = self._convert_to_array()
        nh = hyp0.shape[0]
        R = int(cfg.num_restarts)
        ranges = cfg.meanRange + cfg.covRange + cfg.likRange
        # rank 0 draws the table in the reference's order: restart-major, hyp-minor (Core/opt.py:307-308)
        table = np.empty((R, nh))
        table[0] = hyp0
        cont = bool(getattr(self, "_wave_cont", False))       # a later wave of a threshold-only search: every row is drawn
        if rank == 0:
            for t in range(0 if cont else 1, R):
                for i in range(nh):
                    table[t, i] = np.random.uniform(low=ranges[i][0], high=ranges[i][1])
        # the collectives run at world size 1 too: the RCCL path of a one-GPU job is then the same code that runs on eight
        # (and is exercised by the one-GPU test tier)
        table = comm.bcast(table, 0)                                     # broadcast #1: init table
        xs = np.asarray(self.model.x).shape
        ys = np.asarray(self.model.y).shape
        self.model.x = comm.bcast(np.array(self.model.x, dtype=np.float64), 0).reshape(xs)     # broadcast #2, #3: X, y (~1 MB)
        self.model.y = comm.bcast(np.array(self.model.y, dtype=np.float64), 0).reshape(ys)
        self.init_table = table.copy()                                   # per-restart initial points (row 0 = current hyps)
        # local share
        rec = np.zeros((R, nh + 4))
        rec[:, 0] = np.inf
        rec[:, nh + 2] = 1.0                                             # failed unless proven otherwise
        seq = comm.search_seq = getattr(comm, "search_seq", 0) + 1       # the same on every rank: findMin is collective
        dynamic = self.deal == "dynamic" or (self.deal == "auto" and R > world and world > 1)
        take = None
        if dynamic and comm.ticket("probe/%d" % seq) is not None:
            import threading
            tlock = threading.Lock()
            name = "restart/%d" % seq

            def take():
                with tlock:
                    t = comm.ticket(name)
                return t if t is not None and t < R else None
        mine = None if take else [t for t in range(R) if t % world == rank]
        for t, r in sorted(self._run_share(mine, table, numIters, take).items()):
            rec[t, nh + 3] = 1.0                                         # this rank ran it
            if r.ok:
                rec[t, 0] = r.f
                rec[t, 1:1 + nh] = r.hyp
                rec[t, nh + 1] = r.nls
                rec[t, nh + 2] = 0.0
        # each restart was run by exactly one rank and the others hold (inf, 0.., failed, not run): gather all shares
        # (ONE all-gather of R x (nh+4) doubles), then pick the runner's row
        full = comm.allgather(rec)
        ran = full[:, :, nh + 3]
        if not np.all(ran.sum(axis=0) == 1.0):
            raise RuntimeError("ShardedMinimize: restarts %s were run by %s ranks" % (np.nonzero(ran.sum(axis=0) != 1.0)[0].tolist(),
                                                                                      ran.sum(axis=0)[ran.sum(axis=0) != 1.0].tolist()))
        self.owner = np.argmax(ran, axis=0)
        rec = np.stack([full[self.owner[t], t] for t in range(R)])
        runs = [_Run(rec[t, nh + 2] == 0.0, rec[t, 0], rec[t, 1:1 + nh].copy(), int(rec[t, nh + 1])) for t in range(R)]
        self.runs = runs
        self.trailsCounter += R
        best, errors = _select(runs, R, self.logger, 0)
        self.errorCounter += errors
        if cfg.min_threshold:
            inc = None
            for k, r in enumerate(runs):                                 # replay the sequential early exit
                if k == 0:
                    inc = r if r.ok else None
                elif r.ok and inc is not None and r.f < inc.f:
                    inc = r
                if k >= 1 and inc is not None and inc.f <= cfg.min_threshold:
                    best = inc
                    break
        if best is None:
            raise Exception("Over half of the trails failed for minimize")
        self.logger.warning("[Minimize] %d out of %d trails failed during optimization", self.errorCounter,
                            self.trailsCounter)
        return best.hyp, best.f


def multi_dataset_objective(hyperparams, model, xs, ys, der=False, group=None):
    """Sum of the negative log marginal likelihoods (and gradients) of ONE model over independent data sets -- the objective of
    pyGPs/Demo/Clustering/pyGP_extension.py:27-76 (``gp_likelihood_independent``: covfunc.hyp <- hyperparams, then per
    data set setData + getPosterior, ``all_nlZ += nlZ``, ``all_dnlZ.accumulateDnlZ``) -- with data set i on rank i % world
    and ONE all-reduce of 1 + nhyp + len(xs) doubles.  Every rank passes the same lists.

    Returns (all_nlZ, all_dnlZ or None, per-data-set nlZ array), identical on every rank; the reference's scalar return value
    is ``all_nlZ`` (der False) or ``all_nlZ + sum(all_dnlZ.cov) + sum(all_dnlZ.mean)`` (der True)."""
    from . import inf, sharded
    comm = group if isinstance(group, sharded.Comm) else sharded.search_comm(group)
    model.covfunc.hyp = np.asarray(hyperparams, dtype=float).tolist()
    D = len(xs)
    model.setData(xs[0], ys[0])                      # fixes the mean function's size (the default mean becomes Const on setData)
    nm, nc, nl = len(model.meanfunc.hyp), len(model.covfunc.hyp), len(model.likfunc.hyp)
    acc = np.zeros(1 + nm + nc + nl + D)
    for i in range(comm.rank, D, comm.world):
        model.setData(xs[i], ys[i])
        if der:
            nlZ, dnlZ, post = model.getPosterior(der=True)
            acc[1:1 + nm + nc + nl] += np.array(list(dnlZ.mean) + list(dnlZ.cov) + list(dnlZ.lik), dtype=float)
        else:
            nlZ, post = model.getPosterior(der=False)
        acc[0] += nlZ
        acc[1 + nm + nc + nl + i] = nlZ
    acc = comm.allreduce(acc, "sum")
    all_dnlZ = None
    if der:
        all_dnlZ = inf.dnlZStruct(model.meanfunc, model.covfunc, model.likfunc)
        g = acc[1:1 + nm + nc + nl]
        all_dnlZ.mean = [np.float64(v) for v in g[:nm]]
        all_dnlZ.cov = [np.float64(v) for v in g[nm:nm + nc]]
        all_dnlZ.lik = [np.float64(v) for v in g[nm + nc:]]
    return float(acc[0]), all_dnlZ, acc[1 + nm + nc + nl:].copy()
