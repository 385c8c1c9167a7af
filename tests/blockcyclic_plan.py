"""TEST HELPER (not product code): the partition / ownership / message plan of the 1-D block-cyclic right-looking Cholesky
that csrc/sharded.hip executes on the device (no reference counterpart).  `tests/test_blockcyclic_gloo.py`
runs this plan with numpy tiles on 2 gloo ranks against LAPACK: the ownership arithmetic and the look-ahead order are checked
independently of the HIP code.

Layout.  The (np x np) lower matrix is cut into column panels of `w` columns.  Panel p lives on rank p % world -- 1-D
block-cyclic over COLUMNS, every rank holds full-height column panels.  Per step p:
    owner(p):  D(p)  factor the diagonal block, S(p) solve the rows below  (exactly the single-GPU kernels)
               broadcast  Y_p = the solved panel
    all ranks: TU(p)  C_j -= Y_p[rows of j..] Y_p[rows j]'   for every OWNED panel j > p
Look-ahead: owner(p+1) updates panel p+1 first, factors it and starts its broadcast while the others are still in TU(p).

Wire model.  xGMI is point-to-point: 7 links per GPU, ~153 GB/s per link counted in BOTH directions, i.e. ~77 GB/s peak and
60-75 GB/s realistic per direction.  A pipelined ring broadcast moves every byte once over every link of the ring in ONE
direction: t_bcast ~ bytes / 64 GB/s, independent of world (round 4 budgeted 153 GB/s here: a bidirectional figure, twice
too optimistic).  `wire_model` therefore defaults to 64 GB/s; RCCL may stripe a broadcast over several rings of the full mesh,
which the first hardware run will show in the library's own timers (pgp_sharded_exact_fit timings_out[6..9]).
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

    def wire_model(self, link_GBs=64.0, tflops=60.0):
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
