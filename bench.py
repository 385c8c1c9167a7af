#!/usr/bin/env python3
"""Benchmark of the exact-GP hot path on MI355X (metric of BASELINE.json).

One "step" = one GP fit = Exact.evaluate(..., nargout=3): kernel assembly -> Cholesky -> alpha -> nlZ
-> all hyper-gradients, with x, y resident in HBM and only the hyper-parameters changing (what the
optimiser does, Core/opt.py:70-75); outputs returned to the host per step: nlZ, dnlZ, alpha.
Workload at N GPUs: BASELINE configs[1] (GPR + RBF, N=8192, d=16, fp64, synthetic recipe of SURVEY
8(d)) on EVERY rank with rank-specific hyper-parameters -- i.e. independent objective evaluations of
the restart search (configs[3]) sharded one stream of fits per GPU; RCCL is used only for the
broadcast of (x, y) before and the all-gather of results after the timed region ("weak" scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (gemm_f64_kernel, the fp64 MFMA
GEMM behind the Cholesky trailing updates, the triangular inverse and W'W): algorithmic flops of all
its launches in a fit / their summed duration, measured with HIP events on the library's stream in a
profiled pass over the same steps.  `cpu_baseline` (N=1 only) times the oracle's reference-faithful
CPU path (scipy cdist + LAPACK dpotrf + general-LU solve_chol + one derivative matrix per hyper,
call-for-call what pyGPs does) on this host's cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TF = 78.6      # MI355X fp64 matrix peak: 128 flop/clk/CU x 256 CU x 2.4 GHz (measured issue rate 77.6)
PEAK_HBM_GBS = 8000.0


def synth_reg(N, d, seed=0):
    rng = np.random.RandomState(seed)
    x = rng.randn(N, d)
    w = rng.randn(d, 1)
    y = np.sin(x @ w / np.sqrt(d)) + 0.1 * rng.randn(N, 1)
    return x, y


def hyp_for(step, rank, d):
    """Hyper-parameters of one objective evaluation: the cfg-2 point, nudged so that no two steps repeat."""
    eps = 1e-3 * ((step * 7 + rank * 13) % 101) / 101.0
    return np.array([np.log(np.sqrt(d)) + eps, 0.0 - eps]), float(np.log(0.1) + 0.5 * eps)


def cpu_baseline(N, d, budget_s=40.0):
    """Reference-faithful CPU path (oracle) on the host cores.  Returns the cpu_baseline object."""
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1

    def one(n, faithful=True):
        x, y = synth_reg(n, d)
        c = float(y.mean())
        t = time.perf_counter()
        O.exact_fit(O.RBF, np.array([np.log(np.sqrt(d)), 0.0]), 0, np.log(0.1), x, y, c * np.ones_like(y),
                    np.ones_like(y), nargout=3, faithful=faithful)
        return time.perf_counter() - t

    one(512)                                            # warm the BLAS threads
    t2k = one(2048)
    if t2k * 64 <= budget_s:
        t = one(N)
        sample = "1 full fit at N=%d d=%d (%.1f s), oracle reference-faithful path" % (N, d, t)
        t_sane = one(N, faithful=False) if t * 0.4 <= budget_s else None
    else:
        n_s = 4096 if t2k * 8 <= budget_s else 2048
        ts = one(n_s) if n_s != 2048 else t2k
        t = ts * (N / n_s) ** 3
        sample = ("1 fit at N=%d d=%d took %.1f s; scaled by (N/%d)^3 to N=%d (the fit is O(N^3): LU/Cholesky "
                  "dominated)" % (n_s, d, ts, n_s, N))
        t_sane = one(n_s, faithful=False) * (N / n_s) ** 3
    out = {"value": 1.0 / t, "unit": "fits/s", "cores": int(thr), "kind": "port", "sample": sample,
           "host_cpu_count": os.cpu_count()}
    if t_sane:
        out["value_sane_linear_algebra"] = 1.0 / t_sane       # same maths with triangular solves + potri
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=2,
                    help="independent fit streams per GPU (one pgp_ctx + one host thread each); the K timed steps are "
                         "split over them.  2 overlaps one fit's latency-bound panel phases with the other's GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the N=16384 assembly / Cholesky figures (used by the PMC passes)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # self-test hook only: PYGPS_BENCH_BACKEND=gloo lets N ranks share fewer GPUs (collectives through host memory)
    backend = os.environ.get("PYGPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    cdev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))    # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"

    from pygps_amd import _lib
    lib = _lib.load()
    ctx = _lib.ctx(local)
    N, d = args.n, args.d

    # ---- data: rank 0 generates, RCCL broadcast (configs[3]: "trivial RCCL broadcast/gather") ------
    if rank == 0:
        x, y = synth_reg(N, d)
    else:
        x, y = np.zeros((N, d)), np.zeros((N, 1))
    if dist:
        xt = torch.from_numpy(x).to(cdev)
        yt = torch.from_numpy(y).to(cdev)
        dist.broadcast(xt, src=0)
        dist.broadcast(yt, src=0)
        x, y = xt.cpu().numpy(), yt.cpu().numpy()
    x = np.ascontiguousarray(x)
    yv = np.ascontiguousarray(y).ravel()
    m = np.full(N, yv.mean())
    dm = np.ones((1, N))
    # one context (own HIP streams + workspace) per fit stream; x, y resident in HBM in each
    import ctypes
    import threading
    S = max(1, args.streams)
    ctxs = [ctx]
    for _ in range(1, S):
        h = ctypes.c_void_p()
        _lib.check(lib.pgp_init(local, ctypes.byref(h)), "pgp_init")
        ctxs.append(h)
    for h in ctxs:
        _lib.check(lib.pgp_set_data(h, _lib.ptr(x), N, d, _lib.ptr(yv)))
    bufs = [(np.empty(N), np.zeros(1), np.zeros(4)) for _ in range(S)]

    def fit(step, k=0):
        hyp, log_sn = hyp_for(step, rank, d)
        alpha, nlZ, g = bufs[k]
        rc = lib.pgp_exact_fit(ctxs[k], _lib.COV_RBF, _lib.ptr(hyp), 2, 0, 0, log_sn, _lib.ptr(m), _lib.ptr(dm), 1, 3,
                               _lib.ptr(alpha), _lib.ptr(nlZ), _lib.ptr(g), None)
        _lib.check(rc, "pgp_exact_fit")
        return float(nlZ[0])

    def run_steps(first, count):
        """`count` fits, split round-robin over the S fit streams (ctypes releases the GIL during a fit)."""
        res = [None] * count
        if S == 1:
            for s in range(count):
                res[s] = fit(first + s)
            return res

        def work(k):
            for s in range(k, count, S):
                res[s] = fit(first + s, k)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        return res

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(0, max(args.warmup, S))
    fence()
    t0 = time.perf_counter()
    vals = run_steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    # single-stream latency of one fit (not the headline: reported alongside)
    t1 = time.perf_counter()
    for s in range(3):
        fit(args.warmup + s)
    lat_ms = (time.perf_counter() - t1) / 3 * 1e3
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        res = torch.tensor(vals, dtype=torch.float64, device=cdev)
        parts = [torch.empty_like(res) for _ in range(world)]
        dist.all_gather(parts, res)                                              # RCCL gather of results
        vals_all = torch.stack(parts).cpu().numpy()
        assert np.all(np.isfinite(vals_all))
    stages = _lib.last_timings(local)

    # ---- roofline of the dominant kernel: profiled pass over the same steps (HIP events per launch) ----
    roof = None
    classes = {}
    if rank == 0:
        lib.pgp_profile_reset(ctx)
        lib.pgp_set_profiling(ctx, 1)
        for s in range(args.prof_steps):
            fit(args.warmup + s)
        lib.pgp_set_profiling(ctx, 0)
        prof = _lib.profile(local)
        gl = gm = gf = 0.0
        for name, v in prof.items():
            if v["launches"]:
                classes[name] = {"launches_per_fit": v["launches"] / args.prof_steps,
                                 "ms_per_fit": v["ms"] / args.prof_steps,
                                 "TFLOPs": v["flops"] / max(v["ms"], 1e-12) / 1e9 if v["flops"] else None,
                                 "GBs": v["bytes"] / max(v["ms"], 1e-12) / 1e6 if v["bytes"] else None}
            if name.startswith("gemm_f64"):
                gl += v["launches"]; gm += v["ms"]; gf += v["flops"]
        achieved = gf / max(gm, 1e-12) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "gemm_f64_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"kernel": "gemm_f64_kernel (fp64 MFMA: Cholesky trailing/inner updates incl. the fused inverse, E E^T)",
                "bound": "mfma",
                "achieved": achieved, "peak": PEAK_FP64_MFMA_TF, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TF,
                "traffic": traffic, "launches_per_fit": gl / args.prof_steps,
                "flops_per_launch": gf / max(gl, 1), "avg_launch_ms": gm / max(gl, 1),
                # the Cholesky sweep also produces L^-T (fused triangular inverse): 2 N^3 / 3 flops in that stage
                "cholesky_sweep_TFLOPs": (2.0 * N ** 3 / 3.0) / (stages["potrf"] * 1e-3) / 1e12,
                "cholesky_sweep_frac_of_peak": (2.0 * N ** 3 / 3.0) / (stages["potrf"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TF}
        asm = prof.get("cov_tile_kernel(assemble)")
        if asm and asm["launches"]:
            roof["assembly_GBs"] = asm["bytes"] / asm["ms"] / 1e6
            roof["assembly_frac_of_hbm_peak"] = asm["bytes"] / asm["ms"] / 1e6 / PEAK_HBM_GBS
        # the north-star assembly figure: full symmetric K (getCovMatrix 'train'), RBF, N=16384 d=16, device-resident
        # (both N=16384 extras are skipped with --no-extras)
        # coordinates, algorithmic bytes 8 N^2 + 8 N d, HIP-event time over 100 launches
        try:
            if args.no_extras:
                raise RuntimeError("skipped")
            ms_a = ctypes.c_double()
            na = 16384
            if lib.pgp_test_assemble(ctx, _lib.COV_RBF, 0, na, 16, 100, ctypes.byref(ms_a)) == 0 and ms_a.value > 0:
                ba = 8.0 * na * na + 8.0 * na * 16
                roof["assembly_full_N16384"] = {"ms": ms_a.value, "GBs": ba / ms_a.value / 1e6, "bound": "hbm",
                                                "frac_of_hbm_peak": ba / ms_a.value / 1e6 / PEAK_HBM_GBS}
        except Exception:
            pass
        # the north-star Cholesky figure at N=16384: one extra context, RBF d=16, the factorisation stage of a full fit
        # (with the fused triangular inverse riding along: 2 N^3 / 3 flops in that stage), HIP-event stage time
        try:
            if args.no_extras:
                raise RuntimeError("skipped")
            hb = ctypes.c_void_p()
            nb_ = 16384
            if lib.pgp_init(local, ctypes.byref(hb)) == 0:
                xb, yb = synth_reg(nb_, d)
                ybv = np.ascontiguousarray(yb).ravel()
                mb, dmb = np.full(nb_, ybv.mean()), np.ones((1, nb_))
                ab, nzb, gb = np.empty(nb_), np.zeros(1), np.zeros(4)
                if lib.pgp_set_data(hb, _lib.ptr(np.ascontiguousarray(xb)), nb_, d, _lib.ptr(ybv)) == 0:
                    tb = []
                    for it in range(3):
                        hyp_b, lsn_b = hyp_for(it, 0, d)
                        if lib.pgp_exact_fit(hb, _lib.COV_RBF, _lib.ptr(hyp_b), 2, 0, 0, lsn_b, _lib.ptr(mb), _lib.ptr(dmb), 1, 3,
                                             _lib.ptr(ab), _lib.ptr(nzb), _lib.ptr(gb), None) != 0:
                            break
                        st_b = np.zeros(len(_lib.STAGES))
                        lib.pgp_last_timings(hb, _lib.ptr(st_b))
                        tb.append(dict(zip(_lib.STAGES, st_b.tolist())))
                    if len(tb) == 3:
                        pm = tb[-1]["potrf"]
                        roof["cholesky_sweep_N16384"] = {
                            "ms": pm, "TFLOPs": (2.0 * nb_ ** 3 / 3.0) / (pm * 1e-3) / 1e12, "bound": "mfma",
                            "frac_of_peak": (2.0 * nb_ ** 3 / 3.0) / (pm * 1e-3) / 1e12 / PEAK_FP64_MFMA_TF,
                            "fit_ms": tb[-1]["total"], "fit_TFLOPs": float(nb_) ** 3 / (tb[-1]["total"] * 1e-3) / 1e12}
                lib.pgp_destroy(hb)
        except Exception:
            pass

    if rank == 0:
        total_fits = world * args.steps
        out = {
            "metric": "GP fits/sec (nlZ+grad, RBF, N=%d d=%d)" % (N, d),
            "value": total_fits / dt, "unit": "fits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "GPR+RBF, N=%d d=%d fp64 synthetic (SURVEY 8d recipe, seed 0), infExact nlZ + dnlZ "
                                   "(BASELINE configs[1]); x,y resident in HBM, hyp changes every step; outputs "
                                   "nlZ, dnlZ(4), alpha(N) to host per step" % (N, d),
                       "fits_per_rank": args.steps, "fit_streams_per_gpu": S,
                       "parallelism": "independent fits (restart evaluations) per GPU, %d concurrent fit streams per "
                                      "GPU; RCCL broadcast+gather only" % S},
            "single_stream_ms_per_fit": lat_ms,
            "stage_ms_last_fit": stages,
            "flops_per_fit": float(N) ** 3,
            "fit_TFLOPs": float(N) ** 3 / (dt / args.steps) / 1e12,
            "roofline": roof, "kernel_classes": classes,
            "device": _lib.device_info(local),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, d)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
