"""pygps_amd -- MI355X-native exact-GP core behind the pyGPs API.

Mirrors the module layout a pyGPs user imports (``pyGPs.cov``, ``pyGPs.inf``, ``pyGPs.GPR`` ...) for the
hot path only: kernel-matrix construction (RBF / RBFard / Matern), Exact (and EP) inference, the
Minimize restart loop and predict.  All arithmetic above O(N) runs in hand-written HIP kernels for
gfx950 behind the C ABI in include/pygps_amd.h; importing the package does not need a GPU, calling
into it does (no CPU fallback).
"""
from ._threads import respect_cpu_quota as _respect_cpu_quota

_respect_cpu_quota()            # a container CPU quota below the visible core count: cap the BLAS / OpenMP pools (see _threads.py)
from . import conf, cov, inf, lik, mean, minimize, opt, tools  # noqa: F401,E402
from .gp import GP, GPC, GPR, GP_FITC, GPR_FITC  # noqa: F401,E402

__version__ = "0.1"
