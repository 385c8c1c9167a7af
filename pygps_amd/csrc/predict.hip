// GP.predict on the device (reference: Core/gp.py:349-437, Cholesky parametrisation) and the
// solve_chol helper (Core/tools.py:81-97), both built on one blocked multi-right-hand-side
// triangular solve that runs on the fp64 MFMA GEMM kernel:
//     forward  L V = Y :  for kb = 0..nb-1 :  V_kb = W_kk Y_kb ;  Y_rest -= L(rest,kb) V_kb
//     backward L'X = Y :  for kb = nb-1..0 :  X_kb = W_kk' Y_kb ;  Y_(<kb) -= L(kb,<kb)' X_kb
// with W_kk = inv(L_kk) from leaf_inv_kernel (the reference instead LU-factorises the triangular
// L for every 1000-point batch, gp.py:415).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"

// Y (np x nrhs, column-major, ldy; nrhs multiple of 128: the in-place diagonal solve needs ONE row tile) is overwritten by the solution.
int solve_lower_multi(pgp_ctx* c, const double* L, long ldl, const double* Wd /*128 x np, ld 128*/, double* Y,
                             long ldy, long np, int nrhs, bool trans) {
    const int nb = (int)(np / 128);
    const int tile = (nrhs % 128 == 0 && (long)nrhs / 128 * (np / 128) >= c->small_tile_below) ? 128 : 64;
    // Two-level blocking: inside an outer panel of Q leaves the updates have K = 128 and stay inside the panel; the rows
    // outside the panel are updated ONCE per panel with K = 128 Q (C is read and written once per panel instead of once
    // per leaf, and the k-loop is four times longer: the product that carries ~all the flops runs at the K = 512 rate)
    const int Q = std::max(1, std::min(c->solve_outer, 8));
    auto update = [&](long o, long kw, long r0, long rows) -> int {        // Y[r0 .. r0+rows) -= L-part(.., o..o+kw) V[o..o+kw)
        if (rows <= 0) return PGP_OK;
        GemmArgs h{};
        if (!trans) { h.A = L + r0 + o * ldl; h.lda = ldl; h.a_kc = 0; }
        else        { h.A = L + o + r0 * ldl; h.lda = ldl; h.a_kc = 1; }   // A(m,k) = L(o+k, r0+m)
        h.C = Y + r0;
        h.B = Y + o; h.ldb = ldy; h.b_kc = 1;
        h.ldc = ldy; h.M = (int)rows; h.N = nrhs; h.K = (int)kw; h.alpha = -1.0; h.beta = 1.0; h.tile = tile;
        h.flops = 2.0 * (double)kw * (double)rows * nrhs;
        return gemm_prof(c, PC_GEMM_INNER, h);
    };
    for (int p0 = 0; p0 < nb; p0 += Q) {
        const int p1 = std::min(p0 + Q, nb);
        // leaves of this panel in solve order: forward p0 .. p1-1 (global blocks kb), backward the mirrored blocks
        for (int s = p0; s < p1; ++s) {
            const int kb = trans ? nb - 1 - s : s;
            const long o = (long)kb * 128;
            GemmArgs g{};                               // diagonal solve, in place (one WG column-tile owns its rows)
            g.A = Wd + o * 128; g.lda = 128; g.a_kc = trans ? 1 : 0;
            g.B = Y + o; g.ldb = ldy; g.b_kc = 1;
            g.C = Y + o; g.ldc = ldy;
            g.M = 128; g.N = nrhs; g.K = 128; g.alpha = 1.0; g.beta = 0.0; g.tile = 128;
            g.flops = 128.0 * 128.0 * nrhs;
            CHK(gemm_prof(c, PC_GEMM_INNER, g));
            // the leaves still to come INSIDE the panel
            const long in_rows = (long)(p1 - 1 - s) * 128;
            if (!trans) CHK(update(o, 128, o + 128, in_rows));
            else        CHK(update(o, 128, o - in_rows, in_rows));
        }
        // everything outside the panel, once, with K = 128 (p1 - p0)
        const long kw = (long)(p1 - p0) * 128;
        if (!trans) { const long o = (long)p0 * 128; CHK(update(o, kw, o + kw, np - o - kw)); }
        else        { const long o = (long)(nb - p1) * 128; CHK(update(o, kw, 0, o)); }
    }
    return PGP_OK;
}

static int ensure_wd(pgp_ctx* c, pgp_factor* f) {
    if (f->Wd) return PGP_OK;
    CHK(spool_take(c, (size_t)128 * f->np * sizeof(double), (void**)&f->Wd));
    return leaf_inv_launch(f->F, f->ldf, f->Wd, 128, 128L * 128L, (int)(f->np / 128), c->st);
}

extern "C" {

int pgp_predict(pgp_ctx* c, pgp_factor* f, const double* xs, int64_t ns, const double* ms, double* fmu, double* fs2) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!f) return -2;
    if (!xs) return -3;
    if (ns <= 0) return -4;
    if (!fmu) return -6;
    if (!fs2) return -7;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    CHK(ensure_wd(c, f));
    const long np = f->np, n = f->n;
    const int d = f->d, dpad = f->dpad;
    // test points per batch (the reference uses 1000): up to predict_batch = 65536 within 16 GiB of scratch, so that the K = 128
    // updates of the blocked solve are many waves of tiles and the per-batch host round trip is rare (round 2, N = 8192, 65536 test
    // points: batch 1024 -> 318 ms, 4096 -> 220, 8192 -> 192, 16384 -> 171; round 4 on one box: 16384 -> 120 ms, 65536 -> 101 ms)
    const long NSB = predict_batch_points(c->predict_batch, ns, np);
    const long ldc = NSB;
    // pgp_last_timings after a predict: [ASSEMBLE] = host ms spent getting the scratch (a first call at a new batch shape pays a
    // multi-GiB hipMalloc here: 4 GiB at N = 8192 and 65536 points -- tens of ms on a fast host, hundreds on a slow one; later calls
    // take it from the pool), [SOLVE] = host wall ms of the whole call, [TOTAL] = device ms between the first and the last event.
    const auto t_call = std::chrono::steady_clock::now();
    PoolScratch tmp(c);
    double *xd = nullptr, *XcT = nullptr, *scd = nullptr, *Ks = nullptr, *msd = nullptr, *o1 = nullptr, *o2 = nullptr;
    CHK(tmp.alloc(&xd, NSB * d * sizeof(double)));
    CHK(tmp.alloc(&XcT, (size_t)dpad * ldc * sizeof(double)));
    CHK(tmp.alloc(&scd, dpad * sizeof(double)));
    CHK(tmp.alloc(&Ks, (size_t)np * NSB * sizeof(double)));
    CHK(tmp.alloc(&msd, NSB * sizeof(double)));
    CHK(tmp.alloc(&o1, NSB * sizeof(double)));
    CHK(tmp.alloc(&o2, NSB * sizeof(double)));
    const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    HIP_TRY(hipEventRecord(c->ev[0], st));
    HIP_TRY(hipMemcpyAsync(scd, f->cs.scale.data(), d * sizeof(double), hipMemcpyHostToDevice, st));
    CovSpec cp = f->cs;
    cp.cp.der = -1; cp.pg.der = -1;
    for (long a = 0; a < ns; a += NSB) {
        const long nb_ = std::min<long>(NSB, ns - a);
        const int nrhs = (int)round_up(nb_, 128);
        HIP_TRY(hipMemcpyAsync(xd, xs + a * d, nb_ * d * sizeof(double), hipMemcpyHostToDevice, st));
        if (ms) HIP_TRY(hipMemcpyAsync(msd, ms + a, nb_ * sizeof(double), hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemsetAsync(msd, 0, nb_ * sizeof(double), st));
        CHK(scale_transpose_launch(xd, nb_, d, scd, XcT, ldc, dpad, st));
        // Ks as column-major (np x nrhs): rows = test points, columns = training points in the tile kernel's view
        HIP_TRY(hipMemsetAsync(Ks, 0, (size_t)np * nrhs * sizeof(double), st));
        CHK(cov_rect_launch(XcT, ldc, nb_, f->XsT, np, n, dpad, cp, Ks, np, st));
        CHK(col_dot_full_launch(Ks, np, n, nb_, f->alpha, msd, o1, st));                  // fmu = ms + Ks' alpha
        if (f->sWv) CHK(row_scale_launch(Ks, np, n, nrhs, f->sWv, st));                   // EP: sW o Ks
        CHK(solve_lower_multi(c, f->F, f->ldf, f->Wd, Ks, np, np, nrhs, false));
        CHK(col_sumsq_launch(Ks, np, n, nb_, f->kss, f->sWv ? 1.0 : f->sw * f->sw, o2, st));
        HIP_TRY(hipMemcpyAsync(fmu + a, o1, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(fs2 + a, o2, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        if (a + NSB >= ns) HIP_TRY(hipEventRecord(c->ev[1], st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    {
        float dev_ms = 0.f;
        (void)hipEventElapsedTime(&dev_ms, c->ev[0], c->ev[1]);
        for (double& v : c->last_ms) v = 0.0;
        c->last_ms[PGP_STAGE_ASSEMBLE] = alloc_ms;
        c->last_ms[PGP_STAGE_SOLVE] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        c->last_ms[PGP_STAGE_TOTAL] = dev_ms;
    }
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

// GP.predict (Core/gp.py:395-417) for a posterior whose covariance function is not a device program (csrc/dense.hip): the
// caller hands in the cross-covariance block Ks (n, ns) row-major = getCovMatrix(x, xs, 'cross') and kss (ns) =
// getCovMatrix(z = xs, 'self_test'); fmu = ms + Ks' alpha, fs2 = max(kss - colsum((R'^-1 (sW o Ks))^2), 0).
int pgp_predict_dense(pgp_ctx* c, pgp_factor* f, const double* Ks_host, int64_t ns, const double* kss, const double* ms,
                      double* fmu, double* fs2) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!f) return -2;
    if (!Ks_host) return -3;
    if (ns <= 0) return -4;
    if (!kss) return -5;
    if (!fmu) return -7;
    if (!fs2) return -8;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    CHK(ensure_wd(c, f));
    const long np = f->np, n = f->n;
    const long NSB = predict_batch_points(std::min(c->predict_batch, 16384), ns, np);   // a HOST transpose block of n x batch doubles per step
    PoolScratch tmp(c);
    double *Ks = nullptr, *msd = nullptr, *o1 = nullptr, *o2 = nullptr;
    CHK(tmp.alloc(&Ks, (size_t)np * NSB * sizeof(double)));
    CHK(tmp.alloc(&msd, NSB * sizeof(double)));
    CHK(tmp.alloc(&o1, NSB * sizeof(double)));
    CHK(tmp.alloc(&o2, NSB * sizeof(double)));
    std::vector<double> blk;
    for (long a = 0; a < ns; a += NSB) {
        const long nb_ = std::min<long>(NSB, ns - a);
        const int nrhs = (int)round_up(nb_, 128);
        blk.resize((size_t)nb_ * n);
        for (long i = 0; i < n; ++i)                               // column j of the device block = test point a + j
            for (long j = 0; j < nb_; ++j) blk[(size_t)j * n + i] = Ks_host[(size_t)i * ns + a + j];
        HIP_TRY(hipMemsetAsync(Ks, 0, (size_t)np * nrhs * sizeof(double), st));
        HIP_TRY(hipMemcpy2DAsync(Ks, np * sizeof(double), blk.data(), n * sizeof(double), n * sizeof(double), nb_,
                                 hipMemcpyHostToDevice, st));
        if (ms) HIP_TRY(hipMemcpyAsync(msd, ms + a, nb_ * sizeof(double), hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemsetAsync(msd, 0, nb_ * sizeof(double), st));
        CHK(col_dot_full_launch(Ks, np, n, nb_, f->alpha, msd, o1, st));                  // fmu = ms + Ks' alpha
        if (f->sWv) CHK(row_scale_launch(Ks, np, n, nrhs, f->sWv, st));
        CHK(solve_lower_multi(c, f->F, f->ldf, f->Wd, Ks, np, np, nrhs, false));
        // the raw column sums of squares (kss differs per test point here): out = max(BIG - scale * s, 0) would clip, so take
        // them through kss = 0 with a NEGATIVE scale: out = max(scale' * s, 0) = scale' * s
        CHK(col_sumsq_launch(Ks, np, n, nb_, 0.0, -(f->sWv ? 1.0 : f->sw * f->sw), o2, st));
        HIP_TRY(hipMemcpyAsync(fmu + a, o1, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(fs2 + a, o2, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));                          // blk is reused by the next batch
        for (long j = 0; j < nb_; ++j) fs2[a + j] = std::max(kss[a + j] - fs2[a + j], 0.0);
    }
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

int pgp_potrs(pgp_ctx* c, const double* R, int64_t n, const double* Bm, int64_t nrhs, double* X_out) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!R) return -2;
    if (n <= 0) return -3;
    if (!Bm) return -4;
    if (nrhs <= 0) return -5;
    if (!X_out) return -6;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    const long np = round_up(n, 128);
    const int nr = (int)round_up(nrhs, 128);
    PoolScratch tmp(c);
    double *L = nullptr, *Wd = nullptr, *Y = nullptr;
    CHK(tmp.alloc(&L, (size_t)np * np * sizeof(double)));
    CHK(tmp.alloc(&Wd, (size_t)128 * np * sizeof(double)));
    CHK(tmp.alloc(&Y, (size_t)np * nr * sizeof(double)));
    HIP_TRY(hipMemsetAsync(L, 0, (size_t)np * np * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(Y, 0, (size_t)np * nr * sizeof(double), st));
    // row-major upper R == column-major lower L (same bytes); identity on the padding
    HIP_TRY(hipMemcpy2DAsync(L, np * sizeof(double), R, n * sizeof(double), n * sizeof(double), n, hipMemcpyHostToDevice, st));
    std::vector<double> ones(np - n + 1, 1.0);
    if (np > n)
        HIP_TRY(hipMemcpy2DAsync(L + n + n * np, (np + 1) * sizeof(double), ones.data(), sizeof(double), sizeof(double),
                                 np - n, hipMemcpyHostToDevice, st));
    std::vector<double> yt((size_t)n * nrhs);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < nrhs; ++j) yt[(size_t)j * n + i] = Bm[i * nrhs + j];
    HIP_TRY(hipMemcpy2DAsync(Y, np * sizeof(double), yt.data(), n * sizeof(double), n * sizeof(double), nrhs,
                             hipMemcpyHostToDevice, st));
    CHK(leaf_inv_launch(L, np, Wd, 128, 128L * 128L, (int)(np / 128), st));
    CHK(solve_lower_multi(c, L, np, Wd, Y, np, np, nr, false));      // (R')^-1 = L^-1
    CHK(solve_lower_multi(c, L, np, Wd, Y, np, np, nr, true));       // R^-1 = L^-T
    HIP_TRY(hipMemcpy2DAsync(yt.data(), n * sizeof(double), Y, np * sizeof(double), n * sizeof(double), nrhs,
                             hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < nrhs; ++j) X_out[i * nrhs + j] = yt[(size_t)j * n + i];
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

}  // extern "C"
