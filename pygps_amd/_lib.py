"""ctypes binding of libpygps_amd.so (C ABI: include/pygps_amd.h).

This is the stub a pyGPs maintainer would add to call the MI355X core from
Core/cov.py, Core/inf.py and Core/tools.py.  There is NO
CPU fallback: if the shared library is missing, or no GPU is visible, every
entry point raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYGPS_AMD_LIB: another build of the same library (the host-side AddressSanitizer build, csrc/Makefile target `asan`)
LIB_PATH = os.environ.get("PYGPS_AMD_LIB") or os.path.join(_HERE, "libpygps_amd.so")

COV_RBF, COV_RBFARD, COV_MATERN, COV_RBFUNIT, COV_RQ, COV_PIECEPOLY = 0, 1, 2, 3, 4, 5
COV_RQARD, COV_GABOR, COV_PERIODIC, COV_NOISE, COV_CONST = 6, 7, 8, 9, 10
COV_COMPOSITE = 100
PROG_LEAF, PROG_SUM, PROG_PRODUCT, PROG_SCALE = 1, 2, 3, 4
PROG_MAX_ARD_DIM = 64             # input dimensions of the (single) ARD leaf of a device program
PROG_MAX = 8                      # leaves / Scale nodes / products per device program (csrc/sqdist_tile.h)
MODE_TRAIN, MODE_CROSS, MODE_SELF_TEST = 0, 1, 2
FLAG_MATERN_REFERENCE_DER = 1
STAGES = ("assemble", "potrf", "solve", "trtri", "lauum", "grad", "total")

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); also used by tests/test_capi_symbols.py against include/pygps_amd.h
SIGNATURES = {
    "pgp_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pgp_destroy": (None, [_vp]),
    "pgp_strerror": (C.c_char_p, [C.c_int]),
    "pgp_version": (C.c_char_p, []),
    "pgp_device_count": (C.c_int, []),
    "pgp_device_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_char_p, C.c_int]),
    "pgp_cov": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _i64, _dp, _i64, _i64, _dp, C.c_int, C.c_int,
                          C.c_int, _dp]),
    "pgp_set_composite": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int]),
    "pgp_set_data": (C.c_int, [_vp, _dp, _i64, _i64, _dp]),
    "pgp_exact_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int,
                                C.c_int, _dp, _dp, _dp, C.POINTER(_vp)]),
    "pgp_factor_to_host": (C.c_int, [_vp, _vp, _dp]),
    "pgp_factor_n": (_i64, [_vp]),
    "pgp_factor_free": (None, [_vp, _vp]),
    "pgp_predict": (C.c_int, [_vp, _vp, _dp, _i64, _dp, _dp, _dp]),
    "pgp_exact_fit_dense": (C.c_int, [_vp, _dp, _i64, _dp, C.c_double, C.c_int, _dp, _dp, _dp, C.POINTER(_vp)]),
    "pgp_dense_grad_term": (C.c_int, [_vp, _dp, _i64, C.c_double, _dp]),
    "pgp_predict_dense": (C.c_int, [_vp, _vp, _dp, _i64, _dp, _dp, _dp, _dp]),
    "pgp_ep_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int,
                             _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "pgp_ep_fit_dense": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                   C.POINTER(C.c_int), C.POINTER(_vp)]),
    "pgp_fitc_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _i64, _dp, _dp, C.c_int,
                               C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    "pgp_fitc_predict": (C.c_int, [_vp, _vp, _dp, _i64, _dp, _dp, _dp]),
    "pgp_fitc_free": (None, [_vp, _vp]),
    "pgp_potrf": (C.c_int, [_vp, _dp, _i64, _dp]),
    "pgp_potrs": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _dp]),
    "pgp_last_timings": (C.c_int, [_vp, _dp]),
    "pgp_set_profiling": (C.c_int, [_vp, C.c_int]),
    "pgp_profile_classes": (C.c_int, []),
    "pgp_profile_class_name": (C.c_char_p, [C.c_int]),
    "pgp_profile_read": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), _dp, _dp, _dp]),
    "pgp_profile_reset": (C.c_int, [_vp]),
    "pgp_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "pgp_comm_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "pgp_comm_init_rccl": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "pgp_comm_init_host": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "pgp_comm_free": (None, [_vp]),
    "pgp_comm_world": (C.c_int, [_vp]),
    "pgp_comm_rank": (C.c_int, [_vp]),
    "pgp_comm_bcast_host": (C.c_int, [_vp, _dp, _i64, C.c_int]),
    "pgp_comm_allreduce_host": (C.c_int, [_vp, _dp, _i64, C.c_int]),
    "pgp_comm_allgather_host": (C.c_int, [_vp, _dp, _i64, _dp]),
    "pgp_sharded_exact_fit": (C.c_int, [_vp, _vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int,
                                        C.c_int, _dp, _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    "pgp_sharded_predict": (C.c_int, [_vp, _vp, _vp, _dp, _i64, _dp, _dp, _dp]),
    "pgp_sfactor_free": (None, [_vp, _vp]),
    "pgp_sfactor_bytes": (_i64, [_vp]),
}

# self-test / calibration hooks (csrc/testhooks.h): exported by the library, NOT part of the drop-in boundary
TEST_SIGNATURES = {
    "pgp_test_gemm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.c_double, _dp, _i64, _dp, _i64, _dp, _i64, C.c_int, C.c_int, C.c_int, C.c_int,
                                _dp]),
    "pgp_test_gemm_shrink": (C.c_int, [_vp, _dp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _i64,
                                       _i64]),
    "pgp_test_gemm_skip_wait": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int)]),
    "pgp_test_probit_hazard": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "pgp_test_valu_peak": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "pgp_test_mfma_peak": (C.c_int, [_vp, C.c_int, _dp]),
    "pgp_test_mfma_cycles": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "pgp_test_leaf_ticks": (C.c_int, [_vp, _dp]),
    "pgp_test_assemble": (C.c_int, [_vp, C.c_int, C.c_int, _i64, _i64, C.c_int, _dp]),
    "pgp_test_gemm_trace": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong), _i64, C.POINTER(_i64)]),
    "pgp_test_read_gemm_trace": (C.c_int, [_vp, C.POINTER(C.c_longlong), _i64, C.POINTER(_i64)]),
    "pgp_test_store_roof": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _dp]),
    "pgp_test_cumask_gemm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "pgp_test_wave_costs": (C.c_int, [_vp, _dp]),
    "pgp_test_stream_concurrency": (C.c_int, [_vp, C.c_int, _dp]),
    "pgp_test_slot_probe": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
}

_lock = threading.RLock()
_dll = None
_ctx = {}


def _torch_first():
    """PyTorch's ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 (same sonames as /opt/rocm's).  Whichever
    copy is loaded first serves the whole process.  Measured on the round-2 GPU boxes: with THIS library loaded first
    (runtime from /opt/rocm) and torch imported afterwards, one of the two then fails to see the GPU ("No HIP GPUs are
    available" / "no ROCm-capable device is detected", node dependent); with torch imported first both run on torch's
    copy and every order of initialisation works.  So torch goes first -- but only in processes that use it: when it is
    already imported, or when PYGPS_AMD_TORCH_FIRST=1 asks for it (tests/conftest.py and bench.py do; ShardedMinimize /
    the sharded fit import torch at module level, before the first context exists).  A plain single-GPU fit never pays
    the multi-second torch import nor swaps the HIP runtime it was built against."""
    import sys
    if os.environ.get("PYGPS_AMD_NO_TORCH"):
        return
    if "torch" in sys.modules or os.environ.get("PYGPS_AMD_TORCH_FIRST") == "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def want_torch():
    """Called by the parts of the package that need torch.distributed (ShardedMinimize, the sharded fit): import torch,
    before the library if that is still possible; if the library came first, say which HIP runtime serves the process."""
    import sys
    if "torch" not in sys.modules and _dll is not None:
        import logging
        logging.getLogger(__name__).warning(
            "pygps_amd: torch is imported AFTER libpygps_amd.so (HIP runtime in use: %s); on some nodes one of the two then "
            "sees no GPU -- import torch first or set PYGPS_AMD_TORCH_FIRST=1", hip_runtime_path())
    import torch
    return torch


def hip_runtime_path():
    """Which libamdhip64 this process ended up with (diagnostics: /opt/rocm's or the torch wheel's copy)."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


def load():
    """dlopen the library and attach the prototypes.  Raises if it is not built."""
    global _dll
    if _dll is not None:                    # lock-free fast path
        return _dll
    with _lock:
        if _dll is None:
            _torch_first()
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "pygps_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
            dll = C.CDLL(LIB_PATH)
            for table in (SIGNATURES, TEST_SIGNATURES):
                for name, (res, args) in table.items():
                    fn = getattr(dll, name)      # AttributeError if the .so lacks a declared symbol
                    fn.restype = res
                    fn.argtypes = args
            _dll = dll
    return _dll


def strerror(rc):
    return load().pgp_strerror(rc).decode()


def check(rc, what="pygps_amd", arg_messages=None):
    """Map the C status convention onto the reference's exception behaviour
    (Core/tools.py:66-77 LinAlgError; plain Exception for argument errors)."""
    if rc == 0:
        return
    if rc > 0:
        raise np.linalg.LinAlgError("kernel matrix not positive definite, even with jitter. (first bad pivot %d)" % rc)
    if rc <= -100:
        raise RuntimeError("%s: %s" % (what, strerror(rc)))
    if arg_messages and rc in arg_messages:
        raise Exception(arg_messages[rc])
    raise Exception("%s: bad argument (code %d)" % (what, rc))


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def f64(a):
    """C-contiguous float64 view/copy of a."""
    return np.ascontiguousarray(a, dtype=np.float64)


def default_device():
    return int(os.environ.get("LOCAL_RANK", os.environ.get("PYGPS_AMD_DEVICE", "0")))


_tls = threading.local()


def current_slot():
    """Index of the fit stream the calling thread works on (0 unless inside `fit_stream(k)`)."""
    return getattr(_tls, "slot", 0)


class fit_stream(object):
    """`with fit_stream(k): ...` -- everything the calling thread does inside runs on context k of the device: its
    own HIP streams, workspace pool and resident data.  A context is NOT thread-safe, so concurrent host threads
    (e.g. two restarts optimised at once on one GPU) must each use their own slot."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        self.prev = current_slot()
        _tls.slot = self.k
        return self

    def __exit__(self, *exc):
        _tls.slot = self.prev
        return False


class concurrent_fit_streams(object):
    """`with concurrent_fit_streams(): ...` inside `fit_stream(k)` when SEVERAL fit streams of one GPU run at the same time: the
    context's Cholesky sweep then keeps its critical path (D -> S -> TU_a) on the panel stream beside the bulk updates (the hint
    `concurrent_streams`: the sweep then runs as under `sched` = 1 UNLESS the caller has set `sched` explicitly -- an explicit
    setting is neither overridden nor lost on exit; bit-identical results).  Measured, alternated in one process (profiles/r05_sched_ab_and_leaf_ticks.txt and three
    more boxes): two fit streams +0.6 ... +2 % fits/s, a lone chain -3 % -- hence only while streams run side by side."""

    def __enter__(self):
        # nothing touches the device here: the option is set on the contexts the thread actually uses inside the scope (`ctx`),
        # so host-only callers (a model without device work, the CPU tests of the restart bookkeeping) need no GPU
        self.prev = getattr(_tls, "concurrent", None)
        self.used = {}
        _tls.concurrent = self
        return self

    def _use(self, key, h):
        if key not in self.used:
            load().pgp_set_option(h, b"concurrent_streams", 1)
            self.used[key] = h

    def __exit__(self, *exc):
        _tls.concurrent = self.prev
        for key, h in self.used.items():
            if self.prev is not None and key in self.prev.used:
                continue                                     # an enclosing scope still runs streams side by side on this context
            load().pgp_set_option(h, b"concurrent_streams", 0)  # the hint only: a user's own `sched` setting stays as it was
        self.used = {}
        return False


def ctx(device=None, slot=None):
    """Context (one device + its HIP streams + one workspace pool) of (device, fit-stream slot); created on first use."""
    if device is None:
        device = default_device()
    if slot is None:
        slot = current_slot()
    h = _ctx.get((device, slot))            # lock-free fast path (dict reads are atomic under the GIL)
    scope = getattr(_tls, "concurrent", None)
    if h is not None:
        if scope is not None:
            scope._use((device, slot), h)
        return h
    dll = load()
    with _lock:
        h = _ctx.get((device, slot))
        if h is None:
            out = _vp()
            rc = dll.pgp_init(device, C.byref(out))
            if rc != 0:
                raise RuntimeError("pygps_amd: cannot initialise HIP device %d: %s (no CPU fallback)"
                                   % (device, dll.pgp_strerror(rc).decode()))
            h = out
            _ctx[(device, slot)] = h
    if scope is not None:
        scope._use((device, slot), h)
    return h


def device_info(device=None):
    dll = load()
    ncu, clk, gib = C.c_int(), C.c_int(), C.c_double()
    name = C.create_string_buffer(256)
    check(dll.pgp_device_info(ctx(device), C.byref(ncu), C.byref(clk), C.byref(gib), name, 256))
    return dict(name=name.value.decode(), n_cu=ncu.value, sclk_mhz=clk.value, hbm_gib=gib.value)


_devmem = {}


def device_memory_bytes(device=None):
    """HBM capacity of the device (cached)."""
    if device is None:
        device = default_device()
    if device not in _devmem:
        _devmem[device] = device_info(device)["hbm_gib"] * 2.0 ** 30
    return _devmem[device]


def last_timings(device=None):
    out = np.zeros(len(STAGES))
    check(load().pgp_last_timings(ctx(device), ptr(out)))
    return dict(zip(STAGES, out.tolist()))


def profile(device=None):
    """{class name: dict(launches, ms, flops, bytes)} accumulated since the last reset."""
    dll = load()
    res = {}
    for i in range(dll.pgp_profile_classes()):
        n, ms, fl, by = _i64(), C.c_double(), C.c_double(), C.c_double()
        check(dll.pgp_profile_read(ctx(device), i, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        res[dll.pgp_profile_class_name(i).decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value,
                                                           bytes=by.value)
    return res
