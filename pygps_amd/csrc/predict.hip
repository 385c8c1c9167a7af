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
#include <cstdio>
#include <cstdlib>
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

// ---- the product form (round 6): V = L^-1 (sW o Ks) as ONE fp64 MFMA GEMM ----------------------------------------------------------
// The blocked solve above is a chain of np / 128 dependent steps per batch (three launches each): 65536 points at N = 8192 run
// at 45 TF, 8192 points take ~18 ms.  With W = L^-1 at hand the same V is the NT product  V(m, j) = sum_{k <= m} W(m, k) KsT(j, k)
// on the LDS-DMA 128 x 128 tile (k clipped to the triangle), if the cross-covariances are laid out test-point-contiguous
// (KsT: nrhs x np, column = training point) -- the tile kernel writes them that way when the two point sets swap roles.
// W costs one trtri (N^3 / 3 flops) at the first predict that wants it and stays with the posterior handle (ldf x np doubles, from
// the factor pool) until the handle is freed -- or one transpose when the fit handed its fused inverse rows E = L^-T to the handle
// (option keep_inverse, the default up to np = 16384: W = E').  Option predict_inverse: 0 = never (the blocked solve), 1 = whenever
// W or E is at hand, else for batches of >= 1024 points (default), 2 = always.  Reference: Core/gp.py:395-417 (V = solve(L', sW o Ks)).
namespace {
// KsT lives in SLABS of SW test points: slab s is an (SW x np) column-major matrix (leading dimension SW) at KsT + s SW np, so that
// the k-rows a GEMM tile streams through are 8 SW bytes apart, not 8 nrhs (nrhs = 65536: 512 KiB between k-rows, one TLB entry
// per four of them: the product ran at the blocked solve's 46 TF); element (test j, training k) at (j / SW) SW np + j % SW + k SW.
// part[kc * ldp + j] = sum_{k in chunk kc} KsT(j, k) v[k]      (thread j: coalesced within its slab)
__global__ __launch_bounds__(256) void kst_dot_part_kernel(const double* __restrict__ KsT, long SW, long np, long ncols, long ntrain, long chunk,
                                                           const double* __restrict__ v, double* __restrict__ part, long ldp) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    const double* col = KsT + (j / SW) * SW * np + j % SW;
    const long k0 = (long)blockIdx.y * chunk, k1 = k0 + chunk < ntrain ? k0 + chunk : ntrain;
    double s0 = 0.0, s1 = 0.0;
    long k = k0;
    for (; k + 1 < k1; k += 2) { s0 = fma(col[k * SW], v[k], s0); s1 = fma(col[(k + 1) * SW], v[k + 1], s1); }
    if (k < k1) s0 = fma(col[k * SW], v[k], s0);
    part[(long)blockIdx.y * ldp + j] = s0 + s1;
}
// out[j] = add[j] + sum_kc part[kc][j]   (fixed order)
__global__ __launch_bounds__(256) void kst_dot_finish_kernel(const double* __restrict__ part, long ldp, int nchunk, long ncols,
                                                             const double* __restrict__ add, double* __restrict__ out) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double s = 0.0;
    for (int kc = 0; kc < nchunk; ++kc) s += part[(long)kc * ldp + j];
    out[j] = (add ? add[j] : 0.0) + s;
}
// W (lower) <- E' for an upper-triangular E: 64 x 64 tiles through LDS; blockIdx = (tile row of W, tile column of W), row >= column
__global__ __launch_bounds__(256) void linv_from_e_kernel(const double* __restrict__ E, long lde, double* __restrict__ W, long ldw) {
    if (blockIdx.x < blockIdx.y) return;
    __shared__ double tile[64][65];
    const long m0 = 64L * blockIdx.x, k0 = 64L * blockIdx.y;         // W(m0 + a, k0 + q) = E(k0 + q, m0 + a)
    const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
    for (int q = b; q < 64; q += 4) tile[q][a] = E[k0 + a + (m0 + q) * lde];      // tile[q][a] = E(k0 + a, m0 + q)
    __syncthreads();
    const bool diag = blockIdx.x == blockIdx.y;
    for (int q = b; q < 64; q += 4)
        if (!diag || a >= q) W[m0 + a + (k0 + q) * ldw] = tile[a][q];             // W(m0 + a, k0 + q) = E(k0 + q, m0 + a)
}
// KsT(:, k) *= s[k]   (EP: sW o Ks)
__global__ __launch_bounds__(256) void kst_col_scale_kernel(double* __restrict__ KsT, long SW, long np, long ncols, long ntrain, const double* __restrict__ s) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    double* col = KsT + (j / SW) * SW * np + j % SW;
    for (long k = blockIdx.y; k < ntrain; k += gridDim.y) col[k * SW] *= s[k];
}
}  // namespace

static int ensure_linv(pgp_ctx* c, pgp_factor* f) {
    if (f->Linv) return PGP_OK;
    hipStream_t st = c->st;
    double* W = nullptr;
    CHK(alloc_factor_buffer(c, f->np, f->ldf, &W));
    const size_t bytes = (size_t)f->ldf * f->np * sizeof(double);
    PoolScratch tmp(c);
    double* T = nullptr;
    int rc = tmp.alloc(&T, std::max<size_t>((size_t)f->np * f->np / 4, (size_t)128 * 128) * sizeof(double));
    if (rc == PGP_OK && hipMemsetAsync(W, 0, bytes, st) != hipSuccess) rc = PGP_ERR_HIP;     // (a pooled buffer: only its strict-upper tiles are known to be zero)
    if (rc == PGP_OK && f->Eraw) {
        // the fit left E = L^-T (upper triangular, np x np; whatever lies below its diagonal was never written): W(m, k) = E(k, m), k <= m
        hipLaunchKernelGGL(linv_from_e_kernel, dim3((unsigned)(f->np / 64), (unsigned)(f->np / 64)), dim3(256), 0, st, f->Eraw, f->np, W, f->ldf);
        if (hipGetLastError() != hipSuccess) rc = PGP_ERR_HIP;
    } else
    if (rc == PGP_OK) rc = trtri_lower(c, f->F, f->ldf, W, f->ldf, T, f->np);
    if (rc == PGP_OK && hipStreamSynchronize(st) != hipSuccess) rc = PGP_ERR_HIP;            // (T goes back to the pool)
    if (rc != PGP_OK) {
        (void)hipStreamSynchronize(st);
        (void)hipMemsetAsync(W, 0, bytes, st);                                                // the pool contract: zero strict-upper tiles and augmented rows
        (void)hipStreamSynchronize(st);
        pool_free(c, bytes, W);
        return rc;
    }
    f->Linv = W;
    if (f->Eraw) { spool_give(c, f->Eraw_bytes, f->Eraw); f->Eraw = nullptr; f->Eraw_bytes = 0; }     // (the transpose is done: synchronised above)
    return PGP_OK;
}

// one batch through the product form: xd (nb_ points, scaled + transposed into XcT), results in o1 (fmu) and o2 (fs2)
static int predict_batch_product(pgp_ctx* c, pgp_factor* f, const CovSpec& cp, const double* XcT, long ldc, long nb_, int nrhs, const double* msd,
                                 double* KsT, double* V, double* part, double* o1, double* o2, hipStream_t st) {
    const long np = f->np, n = f->n;
    long SW = 1024;                                                                          // slab width: the largest of 1024 .. 128 that divides nrhs
    while (nrhs % SW) SW >>= 1;
    const int nslab = (int)(nrhs / SW);
    HIP_TRY(hipMemsetAsync(KsT, 0, (size_t)np * nrhs * sizeof(double), st));               // padding rows / columns: exact zeros
    // rows = training points, columns = the slab's test points in the tile kernel's view: out[test + train * SW]
    for (int sl = 0; sl < nslab; ++sl) {
        const long cnt = std::min<long>(SW, nb_ - (long)sl * SW);
        if (cnt <= 0) break;
        CHK(cov_rect_launch(f->XsT, np, n, XcT + (long)sl * SW, ldc, cnt, f->dpad, cp, KsT + (long)sl * SW * np, SW, st));
    }
    constexpr int NCH = 16;
    const long chunk = (n + NCH - 1) / NCH;
    hipLaunchKernelGGL(kst_dot_part_kernel, dim3((unsigned)((nb_ + 255) / 256), NCH), dim3(256), 0, st, KsT, SW, np, nb_, n, chunk, f->alpha, part, (long)nrhs);
    hipLaunchKernelGGL(kst_dot_finish_kernel, dim3((unsigned)((nb_ + 255) / 256)), dim3(256), 0, st, part, (long)nrhs, NCH, nb_, msd, o1);   // fmu = ms + Ks' alpha
    if (f->sWv) hipLaunchKernelGGL(kst_col_scale_kernel, dim3((unsigned)((nb_ + 255) / 256), (unsigned)std::min<long>(n, 32768)), dim3(256), 0, st, KsT, SW, np, nb_, n, f->sWv);
    if (hipGetLastError() != hipSuccess) return PGP_ERR_HIP;
    GemmArgs g{};                                                                            // one product per slab, ONE launch
    g.A = f->Linv; g.lda = f->ldf; g.a_kc = 0;
    g.B = KsT; g.ldb = SW; g.b_kc = 0;
    g.C = V; g.ldc = np;
    g.M = (int)np; g.N = (int)SW; g.K = (int)np; g.alpha = 1.0; g.beta = 0.0;
    g.batch = nslab; g.sA = 0; g.sB = SW * np; g.sC = SW * np;
    g.kmode = KM_LT_I; g.koff = 0; g.tile = 128; g.fold_rows = 1;                           // (tile rows r and mt - 1 - r in one workgroup: uniform work)
    g.flops = (double)np * (double)np * nrhs;
    CHK(gemm_prof(c, PC_GEMM_INNER, g, st));
    return col_sumsq_launch(V, np, n, nb_, f->kss, f->sWv ? 1.0 : f->sw * f->sw, o2, st);
}

// Pinned staging of a batch's inputs (test points, prior mean) and outputs (fmu, fs2).  The callers' arrays are pageable: an
// "asynchronous" copy from / to them is staged by the runtime's host threads, i.e. the DEVICE timeline of a predict call hangs on
// the host's scheduling.  Seen on the GPU boxes (256 cores visible, 16 granted): the reference's host code around the call woke a
// 256-thread BLAS pool, the container's CPU quota throttled the whole process to the next 100 ms period, and every predict call --
// 32768 ... 98304 points, either form -- took 99 ms on the device timeline (34 ... 98 ms of kernels); rounds 3-5 read that as "a
// hot chip" (tools/_pred_q.py, _pred_t.py; pygps_amd/_threads.py caps the pools).  One memcpy through a pinned buffer of our own
// (grow-only, per context) keeps the copies true DMA whatever the host is doing.
static int pred_stage(pgp_ctx* c, size_t doubles) {
    const size_t bytes = doubles * sizeof(double);
    if (c->pred_cap >= bytes) return PGP_OK;
    if (c->pred_host) { (void)hipStreamSynchronize(c->st); (void)hipHostFree(c->pred_host); c->pred_host = nullptr; c->pred_cap = 0; }
    HIP_TRY(hipHostMalloc((void**)&c->pred_host, bytes, hipHostMallocDefault));
    c->pred_cap = bytes;
    return PGP_OK;
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
    const long np = f->np, n = f->n;
    const int d = f->d, dpad = f->dpad;
    // the product form (W = L^-1, one GEMM per batch) or the blocked solve: see predict_batch_product
    const bool product = np >= 1024 && (c->predict_inverse == 2 || (c->predict_inverse == 1 && (f->Linv != nullptr || f->Eraw != nullptr || ns >= 1024)));
    if (product) CHK(ensure_linv(c, f));
    else CHK(ensure_wd(c, f));
    // test points per batch (the reference uses 1000): up to predict_batch = 65536 within 16 GiB of scratch, so that the K = 128
    // updates of the blocked solve are many waves of tiles and the per-batch host round trip is rare (round 2, N = 8192, 65536 test
    // points: batch 1024 -> 318 ms, 4096 -> 220, 8192 -> 192, 16384 -> 171; round 4 on one box: 16384 -> 120 ms, 65536 -> 101 ms)
    const long NSB = predict_batch_points(c->predict_batch, ns, np);
    const long ldc = NSB;
    // pgp_last_timings after a predict: [ASSEMBLE] = host ms spent getting the scratch (a first call at a new batch shape pays a
    // multi-GiB hipMalloc here: 4 GiB at N = 8192 and 65536 points -- tens of ms on a fast host, hundreds on a slow one; later calls
    // take it from the pool), [SOLVE] = host wall ms of the whole call, [TOTAL] = device ms between the first and the last event.
    const auto t_call = std::chrono::steady_clock::now();
    static const bool pred_timing = getenv("PGP_PRED_TIMING") != nullptr;
    auto stamp = [&](const char* what) {
        if (pred_timing) fprintf(stderr, "[predict] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    };
    PoolScratch tmp(c);
    double *xd = nullptr, *XcT = nullptr, *scd = nullptr, *Ks = nullptr, *msd = nullptr, *o1 = nullptr, *o2 = nullptr;
    CHK(tmp.alloc(&xd, NSB * d * sizeof(double)));
    CHK(tmp.alloc(&XcT, (size_t)dpad * ldc * sizeof(double)));
    CHK(tmp.alloc(&scd, dpad * sizeof(double)));
    CHK(tmp.alloc(&Ks, (size_t)np * NSB * sizeof(double)));
    double *Vp = nullptr, *part = nullptr;
    if (product) {
        CHK(tmp.alloc(&Vp, (size_t)np * NSB * sizeof(double)));
        CHK(tmp.alloc(&part, (size_t)16 * NSB * sizeof(double)));
    }
    CHK(tmp.alloc(&msd, NSB * sizeof(double)));
    CHK(tmp.alloc(&o1, NSB * sizeof(double)));
    CHK(tmp.alloc(&o2, NSB * sizeof(double)));
    stamp("device scratch");
    CHK(pred_stage(c, (size_t)NSB * (d + 3) + d));
    stamp("pinned staging");
    double* const h_x = c->pred_host, *const h_ms = h_x + (size_t)NSB * d, *const h_o1 = h_ms + NSB, *const h_o2 = h_o1 + NSB, *const h_sc = h_o2 + NSB;
    const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    HIP_TRY(hipEventRecord(c->ev[0], st));
    memcpy(h_sc, f->cs.scale.data(), d * sizeof(double));
    HIP_TRY(hipMemcpyAsync(scd, h_sc, d * sizeof(double), hipMemcpyHostToDevice, st));
    CovSpec cp = f->cs;
    cp.cp.der = -1; cp.pg.der = -1;
    for (long a = 0; a < ns; a += NSB) {
        const long nb_ = std::min<long>(NSB, ns - a);
        const int nrhs = (int)round_up(nb_, 128);
        stamp("batch start");
        memcpy(h_x, xs + a * d, (size_t)nb_ * d * sizeof(double));             // (the previous batch ended with a stream synchronisation)
        stamp("inputs staged");
        HIP_TRY(hipMemcpyAsync(xd, h_x, nb_ * d * sizeof(double), hipMemcpyHostToDevice, st));
        if (ms) {
            memcpy(h_ms, ms + a, nb_ * sizeof(double));
            HIP_TRY(hipMemcpyAsync(msd, h_ms, nb_ * sizeof(double), hipMemcpyHostToDevice, st));
        } else HIP_TRY(hipMemsetAsync(msd, 0, nb_ * sizeof(double), st));
        CHK(scale_transpose_launch(xd, nb_, d, scd, XcT, ldc, dpad, st));
        if (product) CHK(predict_batch_product(c, f, cp, XcT, ldc, nb_, nrhs, msd, Ks, Vp, part, o1, o2, st));
        else {
        // Ks as column-major (np x nrhs): rows = test points, columns = training points in the tile kernel's view
        HIP_TRY(hipMemsetAsync(Ks, 0, (size_t)np * nrhs * sizeof(double), st));
        CHK(cov_rect_launch(XcT, ldc, nb_, f->XsT, np, n, dpad, cp, Ks, np, st));
        CHK(col_dot_full_launch(Ks, np, n, nb_, f->alpha, msd, o1, st));                  // fmu = ms + Ks' alpha
        if (f->sWv) CHK(row_scale_launch(Ks, np, n, nrhs, f->sWv, st));                   // EP: sW o Ks
        CHK(solve_lower_multi(c, f->F, f->ldf, f->Wd, Ks, np, np, nrhs, false));
        CHK(col_sumsq_launch(Ks, np, n, nb_, f->kss, f->sWv ? 1.0 : f->sw * f->sw, o2, st));
        }
        HIP_TRY(hipMemcpyAsync(h_o1, o1, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_o2, o2, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        if (a + NSB >= ns) HIP_TRY(hipEventRecord(c->ev[1], st));
        stamp("batch queued");
        HIP_TRY(hipStreamSynchronize(st));
        stamp("batch done on the device");
        memcpy(fmu + a, h_o1, nb_ * sizeof(double));
        memcpy(fs2 + a, h_o2, nb_ * sizeof(double));
    }
    {
        float dev_ms = 0.f;
        (void)hipEventElapsedTime(&dev_ms, c->ev[0], c->ev[1]);
        for (double& v : c->last_ms) v = 0.0;
        c->last_ms[PGP_STAGE_ASSEMBLE] = alloc_ms;
        c->last_ms[PGP_STAGE_SOLVE] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        c->last_ms[PGP_STAGE_TOTAL] = dev_ms;
    }
    stamp("timings read");
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
