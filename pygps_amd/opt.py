"""Optimiser plug-ins (reference: pyGPs/Core/opt.py -- Optimizer :35-89, Minimize :273-328).

``Minimize`` is the reference's sequential restart loop around the CG minimiser; every objective
evaluation is one device fit.  ``ShardedMinimize`` runs the SAME restarts one-per-GPU: one process
per GPU (torch.distributed, backend nccl = RCCL over xGMI; gloo in the CPU tests), rank 0 draws the
random initial points in the reference's RNG order and broadcasts them, every rank optimises its
share with its own GPU, a single all-gather returns (nlZ, hyp, #line-searches, failed) per restart
and every rank applies the reference's selection rule.  No collective touches the data path.
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

    def __init__(self, model, searchConfig=None, group=None, streams_per_gpu=None):
        super(ShardedMinimize, self).__init__(model, searchConfig)
        from . import _lib
        _lib.want_torch()               # torch.distributed carries the collectives; torch goes in before the first context
        self.group = group
        self.runs = None                # per-restart records of the last findMin (all ranks)
        if streams_per_gpu is not None:
            self.streams_per_gpu = int(streams_per_gpu)

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

    def _run_share(self, mine, table, numIters):
        """Optimise the restarts `mine` (indices into table); returns {t: _Run}.  With more than one fit stream
        the restarts are dealt to host threads, each with a private deep copy of the model and its own device
        context (pygps_amd._lib.fit_stream); ctypes releases the GIL while a fit runs on the GPU."""
        S = max(1, min(int(self.streams_per_gpu), len(mine)))
        if S == 1:
            return {t: self._one(table[t].copy(), numIters) for t in mine}
        import threading
        from copy import deepcopy as _dc
        from . import _lib
        out = {}

        todo = list(mine)
        lock = threading.Lock()

        def work(k):
            with _lib.fit_stream(k):
                clone = Minimize(_dc(self.model), None)
                clone.model.optimizer = clone
                clone.logger = self.logger
                while True:                       # restarts are taken from a shared queue: line searches differ in length, a
                    with lock:                    # static deal can leave one stream idle at the end.  Results do not depend
                        if not todo:              # on the deal: every restart starts cold (_cold_start)
                            return
                        t = todo.pop(0)
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
        dist = self._dist()
        world = dist.get_world_size(self.group) if dist else 1
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

    @staticmethod
    def _dist():
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist
        except Exception:
            pass
        return None

    def findMin(self, x, y, numIters=200):
        import torch
        cfg = self.searchConfig
        if not cfg or not (cfg.num_restarts or cfg.min_threshold):
            raise Exception("Specify at least one of the stop conditions")       # Core/opt.py:303-304
        if not cfg.num_restarts:
            return self._find_by_threshold(x, y, numIters)
        dist = self._dist()
        rank = dist.get_rank(self.group) if dist else 0
        world = dist.get_world_size(self.group) if dist else 1
        use_cuda = bool(dist) and dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
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
        # the collectives run whenever a process group exists, world size 1 included: the RCCL path of a one-GPU job is
        # then the same code that runs on eight (and is exercised by the one-GPU test tier)
        if dist:
            tt = torch.from_numpy(table).to(dev)
            dist.broadcast(tt, src=0, group=self.group)                  # RCCL broadcast #1: init table
            table = tt.cpu().numpy()
            xt = torch.from_numpy(np.ascontiguousarray(self.model.x, dtype=np.float64)).to(dev)
            yt = torch.from_numpy(np.ascontiguousarray(self.model.y, dtype=np.float64)).to(dev)
            dist.broadcast(xt, src=0, group=self.group)                  # RCCL broadcast #2: X, y (~1 MB)
            dist.broadcast(yt, src=0, group=self.group)
            self.model.x, self.model.y = xt.cpu().numpy(), yt.cpu().numpy()
        self.init_table = table.copy()                                   # per-restart initial points (row 0 = current hyps)
        # local share
        rec = np.zeros((R, nh + 3))
        rec[:, 0] = np.inf
        rec[:, nh + 2] = 1.0                                             # failed unless proven otherwise
        mine = [t for t in range(R) if t % world == rank]
        for t, r in sorted(self._run_share(mine, table, numIters).items()):
            if r.ok:
                rec[t, 0] = r.f
                rec[t, 1:1 + nh] = r.hyp
                rec[t, nh + 1] = r.nls
                rec[t, nh + 2] = 0.0
        if dist:
            mt = torch.from_numpy(rec).to(dev)
            # each restart is owned by exactly one rank and the others hold (inf, 0.., failed):
            # gather all shares, then pick the owner's row
            parts = [torch.empty_like(mt) for _ in range(world)]
            dist.all_gather(parts, mt, group=self.group)                 # RCCL all-gather: R x (nh+3) doubles
            full = np.stack([p.cpu().numpy() for p in parts])
            rec = np.stack([full[t % world, t] for t in range(R)])
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
