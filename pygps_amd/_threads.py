"""Host thread pools under a container CPU quota.

The reference's host code (kept in pygps_amd/lik.py, mean.py ...) calls small numpy / BLAS routines around every device call --
`numpy.linalg.norm(s2)` in lik.Gauss.evaluate's prediction mode (Core/lik.py:134-158), for one.  OpenBLAS / OpenMP size their
pools by the VISIBLE cores; in a container that sees 256 cores and is granted 16 (cgroup cpu.max: the GPU boxes of this
project) one such call wakes 256 spinning threads, the group's CFS quota for the 100 ms period is gone, and the whole process --
the threads that enqueue GPU work included -- is throttled until the next period: `GP.predict` of 32768 points took 100.0 ms
wall for 34 ms of device time, every call (rounds 3-5 read that as "a hot chip"; tools/_pred_t.py, /sys/fs/cgroup/cpu.stat
nr_throttled).  So: when the quota is smaller than the visible core count, cap the pools at the quota -- environment defaults for
libraries not yet loaded, threadpoolctl for those that are.  PYGPS_AMD_KEEP_THREADS=1 leaves everything alone."""
import os

_LIMIT = None          # the threadpoolctl limiter, kept alive for the life of the process


def cpu_quota():
    """Cores the container's CFS quota grants (cgroup v2 cpu.max, v1 cfs_quota_us), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(p)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, -(-q // p)) if q > 0 else None
    except Exception:
        return None


def respect_cpu_quota():
    """Cap OpenMP / BLAS pools at the CPU quota when it is below the visible core count.  Returns the cap or None."""
    global _LIMIT
    if os.environ.get("PYGPS_AMD_KEEP_THREADS"):
        return None
    quota = cpu_quota()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if quota is None or quota >= visible:
        return None
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, str(quota))
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        if any(p.get("num_threads", 1) > quota for p in threadpool_info()):
            _LIMIT = threadpool_limits(limits=quota)
    except Exception:                       # pragma: no cover  (threadpoolctl missing: the environment defaults still hold for later loads)
        pass
    return quota
