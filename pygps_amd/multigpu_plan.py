"""Groundwork for SURVEY 8(f) row 4 -- ONE exact-GP fit spread over the GPUs of a node (no reference counterpart).

Nothing here computes: this module is the partition / ownership / message plan of a 1-D block-cyclic right-looking
Cholesky over `world` ranks (one process per GPU, RCCL over xGMI), i.e. the part of the multi-GPU fit that can be made
correct by construction before an 8-GPU node is available.  `tests/test_blockcyclic_gloo.py` executes the plan with
numpy tiles on 2 gloo ranks and checks it against LAPACK; the device executor (the single-GPU sweep of
csrc/capi.hip:potrf_blocked_v2 restricted to the owned column panels, panel broadcast over RCCL) is a later round.

Layout.  The (np x np) lower matrix is cut into column panels of `w` columns (w = 512 = the single-GPU sweep's outer
panel: K = 512 trailing updates).  Panel p lives on rank p % world -- 1-D block-cyclic over COLUMNS, every rank holds
full-height column panels.  Per step p:
    owner(p):  D(p)  factor the diagonal block, S(p) solve the rows below  (exactly the single-GPU kernels)
               broadcast  Y_p = the solved panel, rows >= (p+1) w                    [ (np - (p+1) w) x w doubles ]
    all ranks: TU(p)  C_j -= Y_p[rows of j..] Y_p[rows j]'   for every OWNED panel j > p
Look-ahead: owner(p+1) updates panel p+1 first, factors it and starts its broadcast while the others are still in
TU(p) -- the broadcast of step p+1 overlaps the trailing update of step p on every rank.

Why 1-D: xGMI is point-to-point (7 links x ~153 GB/s per GPU).  A panel broadcast as a pipelined ring moves each byte
once over every link of the ring: t_bcast ~ bytes / 153 GB/s, independent of world.  At np = 65536 the first panel is
256 MiB -> 1.75 ms, against a trailing update of 2 np^2 w / world / 50 TF = 11 ms per rank at world = 8: the wire is
hidden by a depth-1 look-ahead.  A 2-D layout would cut the panel into sqrt(world) pieces but needs two collectives
per step and row-wise reductions; it only pays when world >> 8.  The fused inverse (E rows) and the rhs rows ride
along as extra rows of every column panel, as on one GPU; E E' afterwards is a reduce-scatter over K.
"""
from collections import namedtuple

Step = namedtuple("Step", "p owner bcast_rows bcast_bytes updates")


class BlockCyclic1D(object):
    """Ownership and schedule of an (np_ x np_) lower-triangular sweep in column panels of w, over `world` ranks."""

    def __init__(self, np_, w, world):
        if np_ <= 0 or w <= 0 or world <= 0:
            raise ValueError("np_, w and world must be positive")
        if np_ % w:
            raise ValueError("np_ must be a multiple of the panel width")
        self.np, self.w, self.world = int(np_), int(w), int(world)
        self.npanel = self.np // self.w

    def owner(self, p):
        if not 0 <= p < self.npanel:
            raise IndexError(p)
        return p % self.world

    def owned(self, rank):
        """Global panel indices stored on `rank`, in storage order."""
        return list(range(rank, self.npanel, self.world))

    def local_index(self, p):
        """Position of global panel p inside its owner's storage."""
        return p // self.world

    def local_cols(self, rank):
        return len(self.owned(rank)) * self.w

    def steps(self):
        """The sweep: one Step per panel; updates[rank] = the owned panels rank updates with Y_p (ascending, so the
        next panel -- the look-ahead target -- comes first on its owner)."""
        out = []
        for p in range(self.npanel):
            rows = self.np - (p + 1) * self.w
            upd = {r: [j for j in self.owned(r) if j > p] for r in range(self.world)}
            out.append(Step(p, self.owner(p), rows, rows * self.w * 8, upd))
        return out

    def flops_per_rank(self):
        """Trailing-update flops per rank (lower tiles only) -- the load balance of the cyclic layout."""
        f = [0.0] * self.world
        for s in self.steps():
            for r, js in s.updates.items():
                for j in js:
                    m = self.np - j * self.w                       # rows of panel j at and below its diagonal block
                    f[r] += 2.0 * self.w * (m * self.w - 0.5 * self.w * self.w)
        return f

    def imbalance(self):
        f = self.flops_per_rank()
        return max(f) / (sum(f) / len(f)) if sum(f) else 1.0

    def wire_model(self, link_GBs=153.0, tflops=50.0):
        """(seconds on the wire, seconds of trailing update on the busiest rank) per step: the look-ahead hides the
        wire while the first stays below the second."""
        out = []
        for s in self.steps():
            t_w = s.bcast_bytes / (link_GBs * 1e9) if self.world > 1 else 0.0
            t_u = 0.0
            for r, js in s.updates.items():
                fl = sum(2.0 * self.w * ((self.np - j * self.w) * self.w - 0.5 * self.w * self.w) for j in js)
                t_u = max(t_u, fl / (tflops * 1e12))
            out.append((t_w, t_u))
        return out
