// FITC sparse regression on the device (reference: FITC_Exact.evaluate Core/inf.py:398-455 with FITCOfKernel's
// (diagK, Kuu, Ku) triple Core/cov.py:352-369, and the dense-L branch of GP.predict Core/gp.py:395-417).
//
// Same building blocks as the exact fit, at (nu x nu) and (nu x n) shapes.  With nu inducing points, n data points:
//   Kuu + snu2 I = Luu' Luu          blocked MFMA Cholesky with the fused inverse:  E1 = Luu^-1  (upper)
//   V = Luu'^-1 Ku = E1' Ku                                   GEMM  nu x n x nu      (reference: LU solve, inf.py:413)
//   g = diagK + sn2 - colsum(V o V)                           column reduction
//   I + (V/g) V' = Lu' Lu            GEMM (lower tiles, K = n) + Cholesky, E2 = Lu^-1; the rhs V r/sqrt(g) rides along
//   alpha = E1 E2 be ;  post.L = E1 (E2 E2' - I) E1'          matvecs / nu^3 GEMMs
//   gradients: B = E1 V, W = E2' (V/g), B W' once; per hyper  R = 2 dKu - dKuu B  and  R W'   (two nu x n x nu GEMMs)
// The reference forms iKuu = inv(Kuu + snu2 I) and multiplies with it; E1 E1' is the same matrix, and every product
// with it is written as two triangular-factor GEMMs here.  O(n) and O(nu) reductions run on the host, like dnlZ.mean
// in the exact fit.  Kuu/snu2 + I is what is factored (the fused assembly kernel adds the identity): Luu = sqrt(snu2) L'.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"

struct pgp_fitc {
    long nu = 0, nup = 0;
    int d = 0, dpad = 0;
    double* XuT = nullptr;      // dpad x nup scaled inducing coordinates
    double* alpha = nullptr;    // nup
    double* Lpost = nullptr;    // nup x nup, symmetric, ld nup
    CovSpec cs;
    double kss = 0.0;
};

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// out[j] = sum_i A(i,j)^2  (one wave per column)
__global__ __launch_bounds__(256) void colsumsq_kernel(const double* __restrict__ A, long lda, long nrows, long ncols,
                                                       double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double* col = A + j * lda;
    double s = 0.0;
    for (long i = lane; i < nrows; i += 64) s = fma(col[i], col[i], s);
    s = wsum(s);
    if (lane == 0) out[j] = s;
}

// out[j] = c0 - sum_i A(i,j) B(i,j)
__global__ __launch_bounds__(256) void coldot2_kernel(const double* __restrict__ A, const double* __restrict__ B, long ld,
                                                      long nrows, long ncols, double c0, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double* a = A + j * ld;
    const double* b = B + j * ld;
    double s = 0.0;
    for (long i = lane; i < nrows; i += 64) s = fma(a[i], b[i], s);
    s = wsum(s);
    if (lane == 0) out[j] = c0 - s;
}

// dst(:,j) = src(:,j) * s[j]   (s == nullptr or j >= nvalid: zero column)
__global__ __launch_bounds__(256) void colscale_kernel(const double* __restrict__ src, double* __restrict__ dst, long ld,
                                                       long nrows, long ncols, long nvalid, const double* __restrict__ s) {
    const long j = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows || j >= ncols) return;
    dst[i + j * ld] = j < nvalid ? src[i + j * ld] * s[j] : 0.0;
}

// part[chunk * nrows + i] = sum_{j in chunk} A(i,j) x[j]   (row-wise product: 64 rows per block, the chunk's columns
// strided over the 4 waves; lanes read consecutive rows = contiguous memory)
constexpr int MV_CHUNK = 1024;
__global__ __launch_bounds__(256) void matvec_rows_kernel(const double* __restrict__ A, long lda, long nrows, long ncols,
                                                          const double* __restrict__ x, double* __restrict__ part) {
    __shared__ double sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const long j0 = (long)blockIdx.y * MV_CHUNK, j1 = j0 + MV_CHUNK < ncols ? j0 + MV_CHUNK : ncols;
    double s = 0.0;
    if (i < nrows)
        for (long j = j0 + w; j < j1; j += 4) s = fma(A[i + j * lda], x[j], s);
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < nrows) part[(long)blockIdx.y * nrows + i] = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
}
// y[i * ystride] = scale * sum_chunk part[chunk * nrows + i]   (fixed order)
__global__ void matvec_rows_finish_kernel(const double* __restrict__ part, long nrows, int nchunk, double scale,
                                          double* __restrict__ y, long ystride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(long)k * nrows + i];
    y[i * ystride] = scale * s;
}

// partial[b] = sum over a grid-strided slice of a[k] * b[k]
__global__ __launch_bounds__(256) void dot_partial_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                          long n, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) s = fma(a[k], b[k], s);
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void set_diag_kernel(double* __restrict__ F, long ld, long n, double v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) F[i + i * ld] = v;
}

int colsumsq(const double* A, long lda, long nrows, long ncols, double* out, hipStream_t st) {
    hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, A, lda, nrows, ncols, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int coldot2(const double* A, const double* B, long ld, long nrows, long ncols, double c0, double* out, hipStream_t st) {
    hipLaunchKernelGGL(coldot2_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, A, B, ld, nrows, ncols, c0, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int colscale(const double* src, double* dst, long ld, long nrows, long ncols, long nvalid, const double* s, hipStream_t st) {
    hipLaunchKernelGGL(colscale_kernel, dim3((unsigned)((nrows + 255) / 256), (unsigned)ncols), dim3(256), 0, st, src, dst,
                       ld, nrows, ncols, nvalid, s);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int matvec_rows(const double* A, long lda, long nrows, long ncols, const double* x, double scale, double* y, long ystride,
                double* part /* ceil(ncols / MV_CHUNK) * nrows */, hipStream_t st) {
    const int nchunk = (int)((ncols + MV_CHUNK - 1) / MV_CHUNK);
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)((nrows + 63) / 64), (unsigned)nchunk), dim3(256), 0, st, A, lda,
                       nrows, ncols, x, part);
    hipLaunchKernelGGL(matvec_rows_finish_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, part, nrows,
                       nchunk, scale, y, ystride);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
// sum_k a[k] b[k], fixed-order (deterministic) two-stage reduction; result on the host
int dot_host(pgp_ctx* c, const double* a, const double* b, long n, double* partial_dev, double* out) {
    const int nb = 512;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, c->st, a, b, n, partial_dev);
    std::vector<double> h(nb);
    HIP_TRY(hipMemcpyAsync(h.data(), partial_dev, nb * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    double s = 0.0;
    for (double v : h) s += v;
    *out = s;
    return PGP_OK;
}

// C(:, j) = beta C(:, j) + alpha sum_s P_s(:, j)   (split-K partial sums, fixed order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ P, long pstride, int S, long M,
                                                            double* __restrict__ C, long ldc, double alpha, double beta) {
    const long j = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double s = 0.0;
    for (int k = 0; k < S; ++k) s += P[(long)k * pstride + i + j * M];
    C[i + j * ldc] = (beta != 0.0 ? beta * C[i + j * ldc] : 0.0) + alpha * s;
}

// C = alpha * op(A) op(B)' + beta * C through the MFMA GEMM; all dimensions multiples of 128.
// (nu x nu) outputs with K = n have only (nu/64)^2 tiles: the k-range is then split over the batch dimension into
// partial products (full chip instead of a third of it) that a small kernel sums in a fixed order.
int gemm(pgp_ctx* c, const double* A, long lda, int a_kc, const double* B, long ldb, int b_kc, double* C, long ldc, long M,
         long N, long K, double alpha, double beta, int tri = 0) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_kc = a_kc; g.B = B; g.ldb = ldb; g.b_kc = b_kc; g.C = C; g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.alpha = alpha; g.beta = beta;
    g.tri = tri; g.tri_off = 0; g.mask_diag = tri ? 1 : 0; g.kmode = KM_FULL;
    const long t128 = (M / 128) * (N / 128) / (tri ? 2 : 1);
    g.tile = t128 < c->small_tile_below ? 64 : 128;
    g.flops = 2.0 * (double)M * N * K * (tri ? 0.5 : 1.0);
    const long tiles = (M / g.tile) * (N / g.tile) / (tri ? 2 : 1);
    int S = 1;
    while (S < 32 && tiles * S < 1536 && K / (2 * S) >= 2048 && (K / (2 * S)) % 16 == 0) S *= 2;
    if (S == 1 || a_kc || b_kc) return gemm_prof(c, PC_GEMM_TRAIL, g, c->st);
    PoolScratch tmp(c);
    double* P = nullptr;
    CHK(tmp.alloc(&P, (size_t)S * M * N * sizeof(double)));
    if (tri) HIP_TRY(hipMemsetAsync(P, 0, (size_t)S * M * N * sizeof(double), c->st));    // upper tiles are not written
    const long kc = K / S;
    g.C = P; g.ldc = M; g.K = (int)kc; g.alpha = 1.0; g.beta = 0.0;
    g.batch = S; g.sA = kc * lda; g.sB = kc * ldb; g.sC = M * N;
    CHK(gemm_prof(c, PC_GEMM_TRAIL, g, c->st));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)N), dim3(256), 0, c->st, P, M * N, S,
                       M, C, ldc, alpha, beta);
    HIP_TRY(hipStreamSynchronize(c->st));                 // P goes back to the pool on return
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// factor the (nup x nup) matrix already assembled in F (ld = 2 nup + 128) with the fused inverse; E = F + nup + 128
int factor_with_inverse(pgp_ctx* c, double* F, long ld, long nup, double* pack) {
    double* E = F + nup + 128;
    CHK(identity_upper_launch(E, ld, nup, c->st));
    double* save = c->inv16;
    c->inv16 = pack;
    const int rc = potrf_blocked(c, F, ld, nup, nup + 128, true);
    c->inv16 = save;
    return rc;
}

}  // namespace

extern "C" {

int pgp_fitc_fit(pgp_ctx* c, int kind, const double* covhyp, int ncov, int para, int flags, double log_sn,
                 const double* xu, int64_t nu, const double* mvec, const double* dm, int nmean, int want,
                 double* alpha_out, double* L_out, double* nlZ_out, double* dnlZ_out, pgp_fitc** handle_out) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (c->n <= 0) return -1;
    if (!covhyp) return -3;
    if (!xu || nu <= 0) return -8;
    if (want < 1 || want > 3) return -11;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    const long n = c->n, d = c->d, np = c->np;
    const int dpad = c->dpad;
    const long nup = round_up(nu, 128);
    CovSpec cs;
    { const int rc = make_spec(c, kind, covhyp, ncov, para, flags, -1, d, cs); if (rc != PGP_OK) return rc == -11 ? -10 : rc; }
    // no N x N workspace here: everything is (nu x n) or (nu x nu)
    double kss = 0.0;
    CHK(cov_point_value(c, cs, 2, &kss));                   // diagK = getCovMatrix(z=x, 'self_test') (cov.py:365)
    const double sn2 = exp(2.0 * log_sn), snu2 = 1e-6 * sn2, isnu = 1.0 / sqrt(snu2);
    const long ldf = 2 * nup + 128;
    const size_t big = (size_t)nup * np * sizeof(double), sq = (size_t)nup * nup * sizeof(double);

    PoolScratch tmp(c);
    double *xud = nullptr, *XuT = nullptr, *Ku = nullptr, *V = nullptr, *Vs = nullptr, *F1 = nullptr, *F2 = nullptr, *pack = nullptr,
           *vecs = nullptr, *nuv = nullptr, *part = nullptr;
    CHK(tmp.alloc(&xud, (size_t)nu * d * sizeof(double)));
    CHK(tmp.alloc(&XuT, (size_t)dpad * nup * sizeof(double)));
    CHK(tmp.alloc(&Ku, big)); CHK(tmp.alloc(&V, big)); CHK(tmp.alloc(&Vs, big));
    CHK(tmp.alloc(&F1, (size_t)ldf * nup * sizeof(double)));
    CHK(tmp.alloc(&F2, (size_t)ldf * nup * sizeof(double)));
    CHK(tmp.alloc(&pack, (size_t)(nup / 128) * PACK_DOUBLES * sizeof(double)));
    CHK(tmp.alloc(&vecs, (size_t)8 * np * sizeof(double)));
    CHK(tmp.alloc(&nuv, (size_t)8 * nup * sizeof(double)));
    CHK(tmp.alloc(&part, (size_t)std::max<long>(std::max<long>(32, (np + MV_CHUNK - 1) / MV_CHUNK) * nup, 1024) * sizeof(double)));
    double *cs_d = vecs, *isg_d = vecs + np, *r_d = vecs + 2 * np, *al_d = vecs + 3 * np, *tmp_d = vecs + 4 * np;
    double *be_d = nuv + nup, *t_d = nuv + 2 * nup, *alpha_d = nuv + 3 * nup, *w_d = nuv + 4 * nup, *q_d = nuv + 5 * nup;
    HIP_TRY(hipMemsetAsync(vecs, 0, (size_t)8 * np * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(nuv, 0, (size_t)8 * nup * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(F1, 0, (size_t)ldf * nup * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(F2, 0, (size_t)ldf * nup * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(Ku, 0, big, st));
    HIP_TRY(hipMemsetAsync(c->info_dev, 0, sizeof(int), st));

    // ---- coordinates, Ku, Kuu ----------------------------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(xud, xu, (size_t)nu * d * sizeof(double), hipMemcpyHostToDevice, st));
    CHK(upload_scaled(c, c->x_dev, n, d, cs.scale, c->XsT, np, dpad, c->scale_dev));
    CHK(scale_transpose_launch(xud, nu, (int)d, c->scale_dev, XuT, nup, dpad, st));
    // Ku (nup x np, column-major, ld nup) = row-major (n x nu) view of the tile kernel: rows = data, columns = inducing
    CHK(cov_rect_launch(c->XsT, np, n, XuT, nup, nu, dpad, cs, Ku, nup, st));
    CHK(cov_factor_launch(XuT, nup, nu, nup, dpad, cs, 1.0 / snu2, F1, ldf, st));       // Kuu / snu2 + I (inf.py:412)
    CHK(factor_with_inverse(c, F1, ldf, nup, pack));
    double* E1 = F1 + nup + 128;                             // L'^-T, upper; Luu^-1 = E1 / sqrt(snu2)
    int info = 0;
    HIP_TRY(hipMemcpyAsync(&info, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    // ---- V, g --------------------------------------------------------------------------------------------
    CHK(gemm(c, E1, ldf, 1, Ku, nup, 1, V, nup, nup, np, nup, isnu, 0.0));               // V = Luu'^-1 Ku
    CHK(colsumsq(V, nup, nup, np, cs_d, st));
    std::vector<double> csum(n), yh(n), mh(n, 0.0);
    HIP_TRY(hipMemcpyAsync(csum.data(), cs_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(yh.data(), c->y_dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (info != 0) return info > (int)nu ? (int)nu : info;
    if (mvec) memcpy(mh.data(), mvec, n * sizeof(double));
    std::vector<double> g(n), isg(n), r(n);
    double sum_logg = 0.0, rtr = 0.0, sum_ig = 0.0;
    for (long j = 0; j < n; ++j) {
        g[j] = kss + sn2 - csum[j];                                                      // inf.py:415
        if (!(g[j] > 0.0)) return (int)std::min<long>(j + 1, nu);                        // not positive definite
        isg[j] = 1.0 / sqrt(g[j]);
        r[j] = (yh[j] - mh[j]) * isg[j];                                                 // inf.py:418
        sum_logg += log(g[j]); rtr += r[j] * r[j]; sum_ig += 1.0 / g[j];
    }
    HIP_TRY(hipMemcpyAsync(isg_d, isg.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(r_d, r.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    CHK(colscale(V, Vs, nup, nup, np, n, isg_d, st));                                    // Vs = V diag(1/sqrt g)
    // ---- I + Vs Vs' = Lu'Lu, be = Lu'^-1 (Vs r) ----------------------------------------------------------
    hipLaunchKernelGGL(set_diag_kernel, dim3((unsigned)((nup + 255) / 256)), dim3(256), 0, st, F2, ldf, nup, 1.0);
    CHK(gemm(c, Vs, nup, 0, Vs, nup, 0, F2, ldf, nup, nup, np, 1.0, 1.0, 1));
    CHK(matvec_rows(Vs, nup, nup, np, r_d, 1.0, F2 + nup, ldf, part, st));                     // rhs row: u = V (r / sqrt g)
    CHK(factor_with_inverse(c, F2, ldf, nup, pack));
    double* E2 = F2 + nup + 128;                             // Lu^-1 (upper)
    HIP_TRY(hipMemcpyAsync(&info, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    CHK(gather_strided_launch(F2 + nup, ldf, nup, be_d, st));
    CHK(logdet_ztz_launch(F2, ldf, nup, F2 + nup, ldf, c->scal, st));                    // sum log diag(Lu), be'be
    CHK(upper_matvec_launch(E2, ldf, nup, be_d, 1.0, part, t_d, st));                    // t = Lu^-1 be
    CHK(upper_matvec_launch(E1, ldf, nup, t_d, isnu, part, alpha_d, st));                // alpha = Luu^-1 t   (inf.py:423)
    // ---- post.L = Luu^-1 ((Lu'Lu)^-1 - I) Luu'^-1 = (E1 E2)(E1 E2)' / snu2 - E1 E1' / snu2     (inf.py:424) ----
    double *M1 = nullptr, *Lp = nullptr;
    CHK(tmp.alloc(&M1, sq)); CHK(tmp.alloc(&Lp, sq));
    CHK(gemm(c, E1, ldf, 0, E2, ldf, 1, M1, nup, nup, nup, nup, 1.0, 0.0));
    CHK(gemm(c, M1, nup, 0, M1, nup, 0, Lp, nup, nup, nup, nup, 1.0 / snu2, 0.0));
    CHK(gemm(c, E1, ldf, 0, E1, ldf, 0, Lp, nup, nup, nup, nup, -1.0 / snu2, 1.0));
    double sc2[2] = {0.0, 0.0};
    std::vector<double> alpha_h(nu), t_h(nup);
    HIP_TRY(hipMemcpyAsync(sc2, c->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(alpha_h.data(), alpha_d, nu * sizeof(double), hipMemcpyDeviceToHost, st));
    if (L_out) HIP_TRY(hipMemcpy2DAsync(L_out, nu * sizeof(double), Lp, nup * sizeof(double), nu * sizeof(double), nu,
                                        hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (info != 0) return info > (int)nu ? (int)nu : info;
    if (alpha_out) memcpy(alpha_out, alpha_h.data(), nu * sizeof(double));
    if (want >= 2 && nlZ_out)
        *nlZ_out = sc2[0] + 0.5 * (sum_logg + (double)n * log(2.0 * M_PI) + rtr - sc2[1]);   // inf.py:428

    // ---- gradients (inf.py:429-452) ------------------------------------------------------------------------
    if (want >= 3 && dnlZ_out) {
        double *Bm = nullptr, *W = nullptr, *R = nullptr, *dKuu = nullptr, *BW = nullptr, *RW = nullptr;
        CHK(tmp.alloc(&Bm, big)); CHK(tmp.alloc(&W, big)); CHK(tmp.alloc(&R, big));
        CHK(tmp.alloc(&dKuu, sq)); CHK(tmp.alloc(&BW, sq)); CHK(tmp.alloc(&RW, sq));
        // al = r/sqrt(g) - V'(Lu^-1 be)/g = isg o (r - Vs' t)
        CHK(col_dot_full_launch(Vs, nup, nup, np, t_d, nullptr, tmp_d, st));
        std::vector<double> vt(n), al(n);
        HIP_TRY(hipMemcpyAsync(vt.data(), tmp_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        double ata = 0.0;
        for (long j = 0; j < n; ++j) { al[j] = isg[j] * (r[j] - vt[j]); ata += al[j] * al[j]; }
        HIP_TRY(hipMemcpyAsync(al_d, al.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
        CHK(gemm(c, E1, ldf, 0, V, nup, 1, Bm, nup, nup, np, nup, isnu, 0.0));           // B = iKuu Ku = Luu^-1 V
        CHK(matvec_rows(Bm, nup, nup, np, al_d, 1.0, w_d, 1, part, st));                       // w = B al
        double* Vg = Ku;                                                                 // Ku is no longer needed
        CHK(colscale(Vs, Vg, nup, nup, np, n, isg_d, st));                               // V / g
        CHK(gemm(c, E2, ldf, 1, Vg, nup, 1, W, nup, nup, np, nup, 1.0, 0.0));            // W = Lu'^-1 (V / g)
        CHK(gemm(c, Bm, nup, 0, W, nup, 0, BW, nup, nup, nup, np, 1.0, 0.0));            // B W'
        CHK(colsumsq(W, nup, nup, np, cs_d, st));
        CHK(colsumsq(Bm, nup, nup, np, tmp_d, st));
        std::vector<double> cw(n), cb(n), wv(nup), qv(nup), vcol(n);
        HIP_TRY(hipMemcpyAsync(cw.data(), cs_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(cb.data(), tmp_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(wv.data(), w_d, nup * sizeof(double), hipMemcpyDeviceToHost, st));
        double bwbw = 0.0;
        CHK(dot_host(c, BW, BW, nup * nup, part, &bwbw));                                // syncs
        double sum_cw = 0.0, wtw = 0.0, s_cb_al2 = 0.0, s_cw_cb = 0.0;
        for (long j = 0; j < n; ++j) { sum_cw += cw[j]; s_cb_al2 += cb[j] * al[j] * al[j]; s_cw_cb += cw[j] * cb[j]; }
        for (long i = 0; i < nu; ++i) wtw += wv[i] * wv[i];
        for (int h = 0; h < ncov; ++h) {
            CovSpec ch;
            CHK(make_spec(c, kind, covhyp, ncov, para, flags, h, d, ch));
            double dk0 = 0.0;
            CHK(cov_point_value(c, ch, 2, &dk0));                                        // ddiagK (cov.py:380)
            HIP_TRY(hipMemsetAsync(R, 0, big, st));
            HIP_TRY(hipMemsetAsync(dKuu, 0, sq, st));
            CHK(cov_rect_launch(c->XsT, np, n, XuT, nup, nu, dpad, ch, R, nup, st));     // dKu
            CHK(cov_sym_launch(XuT, nup, nu, dpad, ch, dKuu, st, nup));                  // dKuu
            CHK(gemm(c, dKuu, nup, 0, Bm, nup, 1, R, nup, nup, np, nup, -1.0, 2.0));     // R = 2 dKu - dKuu B
            CHK(matvec_rows(R, nup, nup, np, al_d, 1.0, q_d, 1, part, st));                    // R al
            CHK(coldot2(R, Bm, nup, nup, np, dk0, tmp_d, st));                           // v = ddiagK - colsum(R o B)
            CHK(gemm(c, R, nup, 0, W, nup, 0, RW, nup, nup, nup, np, 1.0, 0.0));         // R W'
            HIP_TRY(hipMemcpyAsync(qv.data(), q_d, nup * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(vcol.data(), tmp_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
            double rwbw = 0.0;
            CHK(dot_host(c, RW, BW, nup * nup, part, &rwbw));
            // w'(dKuu w - 2 dKu al) = -w'(R al)   because 2 dKu al = R al + dKuu B al = R al + dKuu w
            double wq = 0.0, s3 = 0.0, s4 = 0.0;
            for (long i = 0; i < nu; ++i) wq += wv[i] * qv[i];
            for (long j = 0; j < n; ++j) { s3 += vcol[j] * al[j] * al[j]; s4 += cw[j] * vcol[j]; }
            dnlZ_out[nmean + h] = 0.5 * (dk0 * sum_ig - wq - s3 - s4 - rwbw);            // inf.py:438-439
        }
        // noise: sn2 part + the snu2 = 1e-6 sn2 part with dKuu = 2 snu2 I, R = -2 snu2 B  (inf.py:441-446)
        double lik = sn2 * (sum_ig - sum_cw - ata);
        lik += 0.5 * (2.0 * snu2 * wtw - 2.0 * snu2 * s_cb_al2 - 2.0 * snu2 * s_cw_cb + 2.0 * snu2 * bwbw);
        dnlZ_out[nmean + ncov] = lik;
        for (int i = 0; i < nmean; ++i) {                                                // inf.py:448-450
            double s = 0.0;
            for (long j = 0; j < n; ++j) s += dm[(long)i * n + j] * al[j];
            dnlZ_out[i] = -s;
        }
    }
    if (handle_out) {
        pgp_fitc* f = new pgp_fitc();
        f->nu = nu; f->nup = nup; f->d = (int)d; f->dpad = dpad; f->cs = cs; f->kss = kss;
        CHK(spool_take(c, (size_t)dpad * nup * sizeof(double), (void**)&f->XuT));      // pooled: hipFree would synchronise the device
        CHK(spool_take(c, nup * sizeof(double), (void**)&f->alpha));
        CHK(spool_take(c, (size_t)nup * nup * sizeof(double), (void**)&f->Lpost));
        HIP_TRY(hipMemcpyAsync(f->XuT, XuT, (size_t)dpad * nup * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(f->alpha, alpha_d, nup * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(f->Lpost, Lp, sq, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        *handle_out = f;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

// GP.predict with the FITC posterior (Core/gp.py:395-417, dense-L branch :415):  fmu = ms + Ks' alpha,
// fs2 = max(kss + colsum(Ks o (L Ks)), 0) with Ks = k(xu, xs)
int pgp_fitc_predict(pgp_ctx* c, pgp_fitc* f, const double* xs, int64_t ns, const double* ms, double* fmu, double* fs2) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c || !f) return -1;
    if (!xs || ns <= 0) return -3;
    if (!fmu || !fs2) return -6;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->st;
    const long nup = f->nup, nu = f->nu;
    const int d = f->d, dpad = f->dpad;
    const long NSB = 1024;
    PoolScratch tmp(c);
    double *xd = nullptr, *XcT = nullptr, *scd = nullptr, *Ks = nullptr, *LKs = nullptr, *msd = nullptr, *o1 = nullptr, *o2 = nullptr;
    CHK(tmp.alloc(&xd, NSB * d * sizeof(double)));
    CHK(tmp.alloc(&XcT, (size_t)dpad * NSB * sizeof(double)));
    CHK(tmp.alloc(&scd, dpad * sizeof(double)));
    CHK(tmp.alloc(&Ks, (size_t)nup * NSB * sizeof(double)));
    CHK(tmp.alloc(&LKs, (size_t)nup * NSB * sizeof(double)));
    CHK(tmp.alloc(&msd, NSB * sizeof(double)));
    CHK(tmp.alloc(&o1, NSB * sizeof(double)));
    CHK(tmp.alloc(&o2, NSB * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(scd, f->cs.scale.data(), d * sizeof(double), hipMemcpyHostToDevice, st));
    CovSpec cs = f->cs;
    cs.cp.der = -1; cs.pg.der = -1;
    std::vector<double> h2(NSB);
    for (long a = 0; a < ns; a += NSB) {
        const long nb_ = std::min<long>(NSB, ns - a);
        const long nrhs = round_up(nb_, 128);
        HIP_TRY(hipMemcpyAsync(xd, xs + a * d, nb_ * d * sizeof(double), hipMemcpyHostToDevice, st));
        if (ms) HIP_TRY(hipMemcpyAsync(msd, ms + a, nb_ * sizeof(double), hipMemcpyHostToDevice, st));
        else HIP_TRY(hipMemsetAsync(msd, 0, nb_ * sizeof(double), st));
        CHK(scale_transpose_launch(xd, nb_, d, scd, XcT, NSB, dpad, st));
        HIP_TRY(hipMemsetAsync(Ks, 0, (size_t)nup * nrhs * sizeof(double), st));
        CHK(cov_rect_launch(XcT, NSB, nb_, f->XuT, nup, nu, dpad, cs, Ks, nup, st));     // (nup x nrhs) column-major
        CHK(col_dot_full_launch(Ks, nup, nu, nb_, f->alpha, msd, o1, st));               // fmu
        CHK(gemm(c, f->Lpost, nup, 0, Ks, nup, 1, LKs, nup, nup, nrhs, nup, 1.0, 0.0));  // L Ks
        CHK(coldot2(Ks, LKs, nup, nu, nb_, 0.0, o2, st));                                // -colsum(Ks o L Ks)
        HIP_TRY(hipMemcpyAsync(fmu + a, o1, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h2.data(), o2, nb_ * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (long j = 0; j < nb_; ++j) fs2[a + j] = std::max(f->kss - h2[j], 0.0);       // gp.py:415-416
    }
    if (c->prof) prof_collect(c);
    return PGP_OK;
}

void pgp_fitc_free(pgp_ctx* c, pgp_fitc* f) { GateShared device_gate_hold(c);
    if (!f) return;
    if (c) (void)hipSetDevice(c->device);
    spool_give(c, (size_t)f->dpad * f->nup * sizeof(double), f->XuT);
    spool_give(c, (size_t)f->nup * sizeof(double), f->alpha);
    spool_give(c, (size_t)f->nup * f->nup * sizeof(double), f->Lpost);
    delete f;
}

}  // extern "C"
