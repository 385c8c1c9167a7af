// Self-test / calibration hooks (tests/ and bench.py only): host-buffer GEMM through the MFMA kernel,
// and an fp64 MFMA issue-rate micro-benchmark used to restate the roofline peak from measurement.
#include <cmath>
#include <vector>

#include "ctx.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters, double a0, double b0) {
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;      // keep the chain alive
}
// one wave per SIMD (or two), NACC independent accumulators, timed with s_memtime by lane 0
template <int NACC, bool VACC>
__global__ __launch_bounds__(256) void mfma_cycles_kernel(long long* out, int iters, double a0) {
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = 1.0;
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (VACC) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (s == 123.456) out[2] = (long long)s;
}
}  // namespace

extern "C" {

// out[0] = shader cycles (s_memtime) for iters*nacc MFMAs on one wave, out[1] = 100 MHz wall ticks,
// out[2] = kernel ms (events).  waves_per_simd in {1,2}.
int pgp_test_mfma_cycles(pgp_ctx* c, int iters, int nacc, int waves_per_simd, double* out3) {
    if (!c || !out3) return -1;
    HIP_TRY(hipSetDevice(c->device));
    long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 32));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int blocks = waves_per_simd > 0 ? c->prop.multiProcessorCount * waves_per_simd : -waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(hipEventRecord(e0, c->st));
        if (nacc == 1) hipLaunchKernelGGL((mfma_cycles_kernel<1, false>), dim3(blocks), dim3(256), 0, c->st, d, iters, 1.0);
        else if (nacc == 4) hipLaunchKernelGGL((mfma_cycles_kernel<4, false>), dim3(blocks), dim3(256), 0, c->st, d, iters, 1.0);
        else if (nacc == 8) hipLaunchKernelGGL((mfma_cycles_kernel<8, false>), dim3(blocks), dim3(256), 0, c->st, d, iters, 1.0);
        else if (nacc == -4) hipLaunchKernelGGL((mfma_cycles_kernel<4, true>), dim3(blocks), dim3(256), 0, c->st, d, iters, 1.0);
        else hipLaunchKernelGGL((mfma_cycles_kernel<8, true>), dim3(blocks), dim3(256), 0, c->st, d, iters, 1.0);
        HIP_TRY(hipEventRecord(e1, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    long long h[2];
    HIP_TRY(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    out3[0] = (double)h[0]; out3[1] = (double)h[1]; out3[2] = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
    return PGP_OK;
}

// fp64 VALU issue rate with the instruction mix of the distance loop: 16 x (v_add_f64, v_fma_f64) per step
__global__ __launch_bounds__(256) void valu_peak_kernel(double* out, int iters, double a0, double b0) {
    double s[16], a[4], b[4];
    for (int i = 0; i < 16; ++i) s[i] = 0.0;
    for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 1e-3 + i; b[i] = b0 + i * 0.5; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = a[i] - b[j];
                s[4 * i + j] = fma(d, d, s[4 * i + j]);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] += 1e-9; }
    }
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += s[i];
    if (t == 12345.678) out[0] = t;
}

// instr_rate_out[0] = fp64 VALU wave-instructions per ns over the chip; [1] = implied cycles per instruction at 2.4 GHz
int pgp_test_valu_peak(pgp_ctx* c, int iters, int waves_per_simd, double* out2) {
    if (!c || !out2) return -1;
    HIP_TRY(hipSetDevice(c->device));
    double* out = nullptr;
    HIP_TRY(hipMalloc((void**)&out, 8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int blocks = c->prop.multiProcessorCount * waves_per_simd;
    hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, 64, 1.0, 1.0);
    HIP_TRY(hipEventRecord(e0, c->st));
    hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, iters, 1.0, 1.0);
    HIP_TRY(hipEventRecord(e1, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4.0 * (double)iters * 36.0;           // 16 add + 16 fma + 4 add per step
    const double per_simd = (double)waves_per_simd * (double)iters * 36.0;       // wave-instructions issued per SIMD
    out2[0] = winstr / (ms * 1e6);
    out2[1] = (ms * 1e-3) * 2.4e9 / per_simd;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return PGP_OK;
}

int pgp_test_mfma_peak(pgp_ctx* ctx, int iters, double* tflops_out) {
    if (!ctx || !tflops_out) return -1;
    pgp_ctx* c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    double* out = nullptr;
    HIP_TRY(hipMalloc((void**)&out, 8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int blocks = prop.multiProcessorCount * 2;     // 8 waves / CU = 2 per SIMD
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, 16, 1.0, 1.0);   // warm-up
    HIP_TRY(hipEventRecord(e0, c->st));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->st, out, iters, 1.0, 1.0);
    HIP_TRY(hipEventRecord(e1, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2048.0;
    *tflops_out = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return PGP_OK;
}

// phase time-stamps (s_memtime) of one leaf_potrf_kernel launch on an SPD 128x128 block: ticks_out[24]
int pgp_test_leaf_ticks(pgp_ctx* c, double* ticks_out) {
    if (!c || !ticks_out) return -1;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double> A(128 * 128, 0.0);
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) A[i + j * 128] = (i == j ? 3.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *Ad, *pk; int* info; long long* tk;
    HIP_TRY(hipMalloc((void**)&Ad, A.size() * 8)); HIP_TRY(hipMalloc((void**)&pk, PACK_DOUBLES * 8));
    HIP_TRY(hipMalloc((void**)&info, 4)); HIP_TRY(hipMalloc((void**)&tk, 24 * 8));
    HIP_TRY(hipMemset(info, 0, 4)); HIP_TRY(hipMemset(tk, 0, 24 * 8));
    for (int rep = 0; rep < 3; ++rep) {
        HIP_TRY(hipMemcpy(Ad, A.data(), A.size() * 8, hipMemcpyHostToDevice));
        int rc = leaf_potrf_launch(Ad, 128, pk, info, 0, c->st, tk);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->st));
    }
    long long h[24];
    HIP_TRY(hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 24; ++i) ticks_out[i] = (double)h[i];
    (void)hipFree(Ad); (void)hipFree(pk); (void)hipFree(info); (void)hipFree(tk);
    return PGP_OK;
}

// wall-clock stamps (100 MHz) of the resident diagonal-panel server's last sweep (option ds_ticks=1): 16 per panel
int pgp_test_ds_ticks(pgp_ctx* c, double* ticks_out, int npanel) {
    if (!c || !ticks_out || !c->ds_ticks) return -1;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<long long> h((size_t)npanel * 16);
    HIP_TRY(hipMemcpy(h.data(), c->ds_ticks, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) ticks_out[i] = (double)h[i];
    return PGP_OK;
}

// device-only timing of the kernel-assembly tile kernel on synthetic resident coordinates:
// mode 0 = full symmetric (n,n) output ('train'), 2 = fused lower-triangle B = K/sn2 + I.  ms_out = avg per launch.
int pgp_test_assemble(pgp_ctx* c, int kind, int mode, int64_t n, int64_t d, int iters, double* ms_out) {
    if (!c || !ms_out) return -1;
    HIP_TRY(hipSetDevice(c->device));
    const long np = round_up(n, 128);
    const int dpad = (int)round_up(d, SKC);
    std::vector<double> x((size_t)n * d);
    unsigned long long s = 88172645463325252ULL;
    for (auto& v : x) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 3.4; }
    std::vector<double> hyp(kind == PGP_COV_RBFARD ? d + 1 : 2, 0.0);
    for (size_t i = 0; i + 1 < hyp.size(); ++i) hyp[i] = 0.5 * log((double)d);
    CovSpec cp;
    CHK(make_spec(c, kind, hyp.data(), (int)hyp.size(), 3, 0, -1, d, cp));
    const std::vector<double>& sc = cp.scale;
    double *xd, *XT, *scd, *out;
    const long ldo = mode == 2 ? np + 128 : n;
    HIP_TRY(hipMalloc((void**)&xd, x.size() * 8)); HIP_TRY(hipMalloc((void**)&XT, (size_t)dpad * np * 8));
    HIP_TRY(hipMalloc((void**)&scd, dpad * 8)); HIP_TRY(hipMalloc((void**)&out, (size_t)ldo * np * 8));
    HIP_TRY(hipMemcpy(xd, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(scd, sc.data(), d * 8, hipMemcpyHostToDevice));
    CHK(scale_transpose_launch(xd, n, (int)d, scd, XT, np, dpad, c->st));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = PGP_OK;
    for (int it = -1; it < iters && rc == PGP_OK; ++it) {
        if (it == 0) HIP_TRY(hipEventRecord(e0, c->st));
        rc = mode == 2 ? cov_factor_launch(XT, np, n, np, dpad, cp, 100.0, out, ldo, c->st)
                       : cov_sym_launch(XT, np, n, dpad, cp, out, c->st, 0);
    }
    HIP_TRY(hipEventRecord(e1, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(xd); (void)hipFree(XT); (void)hipFree(scd); (void)hipFree(out);
    return rc;
}

// Does a small panel kernel on the high-priority stream overlap a big trailing GEMM on the main stream?
// out[0] = ms until the GEMM is done, out[1] = ms until the leaf kernel (issued after it, other stream) is done,
// out[2] = same for a trsm over 4096 rows, out[3] = leaf alone.
int pgp_test_overlap(pgp_ctx* c, double* out) {
    if (!c || !out) return -1;
    HIP_TRY(hipSetDevice(c->device));
    const int N = 8192, K = 512;
    double *A, *Cm, *L, *pk, *X; int* info;
    HIP_TRY(hipMalloc((void**)&A, (size_t)N * K * 8)); HIP_TRY(hipMalloc((void**)&Cm, (size_t)N * N * 8));
    HIP_TRY(hipMalloc((void**)&L, 128 * 128 * 8)); HIP_TRY(hipMalloc((void**)&pk, PACK_DOUBLES * 8));
    HIP_TRY(hipMalloc((void**)&X, (size_t)4096 * 128 * 8)); HIP_TRY(hipMalloc((void**)&info, 4));
    HIP_TRY(hipMemset(A, 0, (size_t)N * K * 8)); HIP_TRY(hipMemset(Cm, 0, (size_t)N * N * 8));
    HIP_TRY(hipMemset(X, 0, (size_t)4096 * 128 * 8)); HIP_TRY(hipMemset(info, 0, 4));
    std::vector<double> Lh(128 * 128, 0.0);
    for (int i = 0; i < 128; ++i) Lh[i + i * 128] = 2.0;
    HIP_TRY(hipMemcpy(L, Lh.data(), Lh.size() * 8, hipMemcpyHostToDevice));
    GemmArgs g{};
    g.A = A; g.lda = N; g.B = A; g.ldb = N; g.C = Cm; g.ldc = N; g.M = N; g.N = N; g.K = K;
    g.alpha = -1.0; g.beta = 1.0; g.tile = 128; g.batch = 1;
    hipEvent_t e0, eg, el, et;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&eg)); HIP_TRY(hipEventCreate(&el)); HIP_TRY(hipEventCreate(&et));
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipEventRecord(e0, c->st));
        HIP_TRY(hipStreamWaitEvent(c->st2, e0, 0));
        CHK(gemm_f64_launch(g, c->st));
        HIP_TRY(hipEventRecord(eg, c->st));
        CHK(leaf_potrf_launch(L, 128, pk, info, 0, c->st2));
        HIP_TRY(hipEventRecord(el, c->st2));
        CHK(trsm_rows_launch(X, 4096, 4096, L, 128, pk, c->st2));
        HIP_TRY(hipEventRecord(et, c->st2));
        HIP_TRY(hipDeviceSynchronize());
    }
    float a, b, d;
    HIP_TRY(hipEventElapsedTime(&a, e0, eg)); HIP_TRY(hipEventElapsedTime(&b, e0, el)); HIP_TRY(hipEventElapsedTime(&d, e0, et));
    out[0] = a; out[1] = b; out[2] = d;
    HIP_TRY(hipEventRecord(e0, c->st2));
    CHK(leaf_potrf_launch(L, 128, pk, info, 0, c->st2));
    HIP_TRY(hipEventRecord(el, c->st2));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventElapsedTime(&a, e0, el));
    out[3] = a;
    (void)hipFree(A); (void)hipFree(Cm); (void)hipFree(L); (void)hipFree(pk); (void)hipFree(X); (void)hipFree(info);
    return PGP_OK;
}

int pgp_test_gemm(pgp_ctx* ctx, int tile, int a_kc, int b_kc, int tri, int mask_diag, int kmode, int koff,
                  double alpha, double beta, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                  int64_t ldc, int M, int N, int K, int iters, double* ms_out) {
    if (!ctx) return -1;
    pgp_ctx* c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t an = (size_t)lda * (a_kc ? M : K), bn = (size_t)ldb * (b_kc ? N : K), cn = (size_t)ldc * N;
    double *Ad, *Bd, *Cd;
    HIP_TRY(hipMalloc((void**)&Ad, an * 8)); HIP_TRY(hipMalloc((void**)&Bd, bn * 8)); HIP_TRY(hipMalloc((void**)&Cd, cn * 8));
    HIP_TRY(hipMemcpy(Ad, A, an * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(Bd, B, bn * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(Cd, C, cn * 8, hipMemcpyHostToDevice));
    GemmArgs g{};
    g.A = Ad; g.lda = lda; g.a_kc = a_kc; g.B = Bd; g.ldb = ldb; g.b_kc = b_kc; g.C = Cd; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.tri = tri; g.tri_off = 0; g.mask_diag = mask_diag;
    g.kmode = kmode; g.koff = koff; g.batch = 1; g.tile = tile; g.dbg = c->gemm_dbg;
    hipStream_t ts = c->st_masked ? c->st_masked : c->st;         // option cu_reserve: time the CU-masked stream
    long long* stamps = nullptr;
    const long nwg = (long)(M / tile) * (N / tile);
    if (c->gemm_dbg & 128) { HIP_TRY(hipMalloc((void**)&stamps, nwg * 4 * sizeof(long long))); g.stamps = stamps; }
    int rc = gemm_f64_launch(g, ts);
    HIP_TRY(hipStreamSynchronize(ts));
    if (rc == PGP_OK) HIP_TRY(hipMemcpy(C, Cd, cn * 8, hipMemcpyDeviceToHost));
    if (rc == PGP_OK && iters > 0 && ms_out) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, ts));
        for (int i = 0; i < iters; ++i) rc = gemm_f64_launch(g, ts);
        HIP_TRY(hipEventRecord(e1, ts));
        HIP_TRY(hipStreamSynchronize(ts));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (stamps) {                 // per-tile phase times of the LAST launch (wall clock, 100 MHz)
        std::vector<long long> h((size_t)nwg * 4);
        HIP_TRY(hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double a = 0, b = 0, d = 0, e = 0; long long tmin = h[0], tmax = 0;
        for (long w = 0; w < nwg; ++w) {
            a += (double)(h[4 * w + 1] - h[4 * w]); b += (double)(h[4 * w + 2] - h[4 * w + 1]); d += (double)(h[4 * w + 3] - h[4 * w + 2]);
            e += (double)(h[4 * w + 3] - h[4 * w]);
            tmin = std::min(tmin, h[4 * w]); tmax = std::max(tmax, h[4 * w + 3]);
        }
        fprintf(stderr, "gemm stamps: %ld tiles; per tile us: prologue (C pre-load + first stage) %.1f, k-loop %.1f, stores %.1f, total %.1f; "
                        "kernel span %.1f us, sum(tile time)/512 slots = %.1f us\n", nwg, a / nwg / 100, b / nwg / 100, d / nwg / 100,
                e / nwg / 100, (double)(tmax - tmin) / 100, e / 100 / 512);
        (void)hipFree(stamps);
    }
    (void)hipFree(Ad); (void)hipFree(Bd); (void)hipFree(Cd);
    return rc;
}

// Two products in one grid (gemm_f64_dual_launch): a = trailing-update shape (C1 -= A1 A1^T, lower trapezoid), b = the
// E E^T filler shape (C2 = [rows < zero_from: C2] + A2 A2^T, packed lower tiles, k >= row + koff).  Results back to the host.
int pgp_test_gemm_dual(pgp_ctx* ctx, const double* A1, int64_t lda1, double* C1, int64_t ldc1, int M1, int K1,
                       const double* A2, int64_t lda2, double* C2, int64_t ldc2, int M2, int K2, int koff2, int zero_from2) {
    if (!ctx) return -1;
    pgp_ctx* c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    DevScratch scr;
    double *a1, *c1, *a2, *c2;
    CHK(scr.alloc(&a1, (size_t)lda1 * K1 * 8)); CHK(scr.alloc(&c1, (size_t)ldc1 * M1 * 8));
    CHK(scr.alloc(&a2, (size_t)lda2 * K2 * 8)); CHK(scr.alloc(&c2, (size_t)ldc2 * M2 * 8));
    HIP_TRY(hipMemcpy(a1, A1, (size_t)lda1 * K1 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c1, C1, (size_t)ldc1 * M1 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(a2, A2, (size_t)lda2 * K2 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c2, C2, (size_t)ldc2 * M2 * 8, hipMemcpyHostToDevice));
    GemmArgs a{}, b{};
    a.A = a1; a.lda = lda1; a.B = a1; a.ldb = lda1; a.C = c1; a.ldc = ldc1; a.M = M1; a.N = M1; a.K = K1;
    a.alpha = -1.0; a.beta = 1.0; a.tri = 1; a.mask_diag = 1; a.kmode = KM_FULL; a.batch = 1; a.tile = 128; a.dbg = c->gemm_dbg;
    b.A = a2; b.lda = lda2; b.B = a2; b.ldb = lda2; b.C = c2; b.ldc = ldc2; b.M = M2; b.N = M2; b.K = K2;
    b.alpha = 1.0; b.beta = 1.0; b.tri = 2; b.mask_diag = 1; b.kmode = KM_GE_I; b.koff = koff2; b.zero_from = zero_from2;
    b.batch = 1; b.tile = 128; b.dbg = c->gemm_dbg;
    if (!gemm_f64_dual_ok(a, b)) return -2;
    CHK(gemm_f64_dual_launch(a, b, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    HIP_TRY(hipMemcpy(C1, c1, (size_t)ldc1 * M1 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(C2, c2, (size_t)ldc2 * M2 * 8, hipMemcpyDeviceToHost));
    return PGP_OK;
}

}  // extern "C"
