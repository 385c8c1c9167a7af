// Exact.evaluate (Core/inf.py:353-384) and GP.predict (Core/gp.py:395-417) from CALLER-BUILT covariance matrices.
//
// The reference composes kernels freely (ProductOfKernel / SumOfKernel / ScaleOfKernel, Core/cov.py:230-328).  Trees that
// fit a device program (<= 8 leaves / products, <= 2 ARD leaves) run fused inside the tile kernels; every OTHER tree still has
// getCovMatrix / getDerMatrix -- the children's device-built matrices combined by the Python layer -- and this file gives such
// a kernel the rest of the hot path: the factorisation with the fused inverse, alpha, nlZ, tr Q, the Hadamard sums
// sum(Q o dK_h) / 2 against derivative matrices handed in one at a time, and the predictive solve against a cross-covariance
// block handed in.  Everything O(n^3) and the O(n^2) reductions run on the device; what the generic path costs is PCIe
// traffic (n^2 doubles per matrix).  No reference counterpart beyond the call sites above.
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"

namespace {

// F (column-major lower, ld) <- K / sn2 + I from the symmetric n x n matrix Kd (row-major == column-major), identity on the
// padding; only entries on or below the diagonal are written (the strict upper part of a factor buffer stays zero)
__global__ __launch_bounds__(256) void dense_to_factor_kernel(const double* __restrict__ Kd, long n, double inv_sn2,
                                                              double* __restrict__ F, long ld, long np) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    for (long j = blockIdx.y; j <= i; j += gridDim.y) {          // (grid.y is capped at 65535: columns by stride)
        double v = (i == j) ? 1.0 : 0.0;
        if (i < n && j < n) v = fma(Kd[i + j * n], inv_sn2, v);
        F[i + j * ld] = v;
    }
}

// partial[j] = sum_{i >= j} w_ij (Binv(i,j) inv_sn2 - a_i a_j) D(i,j),  w = 1 on the diagonal, 2 below (symmetric matrices)
__global__ __launch_bounds__(256) void dense_hadamard_kernel(const double* __restrict__ Binv, long ldb, const double* __restrict__ alpha,
                                                             const double* __restrict__ D, long n, double inv_sn2,
                                                             double* __restrict__ partial) {
    __shared__ double red[4];
    const long j = blockIdx.x;
    const double aj = alpha[j];
    double s = 0.0;
    for (long i = j + threadIdx.x; i < n; i += 256) {
        const double q = fma(Binv[i + j * ldb], inv_sn2, -alpha[i] * aj);
        s = fma((i == j ? 1.0 : 2.0) * q, D[i + j * n], s);
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[j] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[0] = sum_j partial[j] in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void dense_sum_kernel(const double* __restrict__ partial, long n, double scale, double* __restrict__ out) {
    __shared__ double r[256];
    double s = 0.0;
    for (long j = threadIdx.x; j < n; j += 256) s += partial[j];
    r[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) r[threadIdx.x] += r[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = scale * r[0];
}
// partial[j] = Binv(j,j) - sn2 a_j^2   ->  sn2 tr Q  (Core/inf.py:374)
__global__ __launch_bounds__(256) void dense_trace_kernel(const double* __restrict__ Binv, long ldb, const double* __restrict__ alpha,
                                                          long n, double sn2, double* __restrict__ partial) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j < n) partial[j] = Binv[j + j * ldb] - sn2 * alpha[j] * alpha[j];
}

}  // namespace

extern "C" {

// r = y - m (n).  K (n,n) symmetric, row-major host.  want as pgp_exact_fit.  dnlZ_lik_out: sn2 tr Q (want = 3).  The inverse
// B^-1 and alpha stay in the context's workspace for the pgp_dense_grad_term calls that follow directly.
int pgp_exact_fit_dense(pgp_ctx* c, const double* K, int64_t n, const double* r, double log_sn, int want, double* alpha_out,
                        double* nlZ_out, double* dnlZ_lik_out, pgp_factor** factor_out) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!K) return -2;
    if (n <= 0) return -3;
    if (!r) return -4;
    if (want < 1 || want > 3) return -6;
    HIP_TRY(hipSetDevice(c->device));
    c->dense_ready = false;
    const long np = round_up(n, 128), ldf = np + 128;
    const bool fused = want >= 3;
    CHK(ensure_workspace(c, np));
    const long need = std::max<long>(32L * np, np);
    if (c->partial_cap < need) {
        if (c->partial) (void)hipFree(c->partial);
        c->partial = nullptr; c->partial_cap = 0;
        HIP_TRY(hipMalloc((void**)&c->partial, need * sizeof(double)));
        c->partial_cap = need;
    }
    const double sn2 = exp(2.0 * log_sn);
    double* F = nullptr;
    CHK(alloc_factor_buffer(c, np, ldf, &F));
    FactorGuard fguard(c, F, (size_t)ldf * np * sizeof(double), /*scrub=*/true);
    PoolScratch scr(c);
    double *Kd = nullptr, *E = nullptr, *zero = nullptr;
    CHK(scr.alloc(&Kd, (size_t)n * n * sizeof(double)));
    CHK(scr.alloc(&zero, (size_t)np * sizeof(double)));
    if (fused) CHK(scr.alloc(&E, (size_t)np * np * sizeof(double)));
    hipStream_t st = c->st;
    HIP_TRY(hipMemsetAsync(c->info_dev, 0, sizeof(int), st));
    HIP_TRY(hipMemcpyAsync(Kd, K, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(zero, 0, np * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(c->zvec, 0, np * sizeof(double), st));
    HIP_TRY(hipMemcpyAsync(c->zvec, r, n * sizeof(double), hipMemcpyHostToDevice, st));      // zvec doubles as the upload of r
    hipLaunchKernelGGL(dense_to_factor_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, Kd, n, 1.0 / sn2, F,
                       ldf, np);
    CHK(aug_rhs_launch(c->zvec, zero, n, F, ldf, np, c->rvec, st));
    if (fused) { c->eet_out = c->Binv; c->eet_ld = np; }
    c->eet_join = nullptr;
    const int prc = potrf_blocked(c, F, ldf, np, np + 128, fused, E, np);
    c->eet_out = nullptr;
    if (prc != PGP_OK) (void)hipDeviceSynchronize();
    CHK(prc);
    CHK(logdet_ztz_launch(F, ldf, n, F + np, ldf, c->scal, st));
    CHK(gather_strided_launch(F + np, ldf, np, c->zvec, st));
    if (fused) {
        CHK(upper_matvec_launch(E, np, np, c->zvec, 1.0 / sn2, c->partial, c->alpha_dev, st));
        if (c->eet_join) HIP_TRY(hipStreamWaitEvent(st, c->eet_join, 0));
        else CHK(eet_lower(c, E, np, c->Binv, np, np));
        hipLaunchKernelGGL(dense_trace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->Binv, np, c->alpha_dev, n, sn2,
                           c->partial);
        hipLaunchKernelGGL(dense_sum_kernel, dim3(1), dim3(256), 0, st, c->partial, n, 1.0, c->scal + 8);
    } else {
        CHK(leaf_inv_launch(F, ldf, c->W, np, 128L * (1 + np), (int)(np / 128), st));
        CHK(trsv_bwd_launch(F, ldf, c->W, np, c->zvec, c->alpha_dev, (int)(np / 128), st));
    }
    if (hipGetLastError() != hipSuccess) return PGP_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(c->res_host, c->res_dev, (size_t)(272 + n) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c->prof) prof_collect(c);
    int info = 0;
    memcpy(&info, c->res_host + 264, sizeof(int));
    if (info != 0) return info > (int)n ? (int)n : info;
    double* alpha_h = c->res_host + 272;
    if (!fused) for (long j = 0; j < n; ++j) alpha_h[j] /= sn2;
    if (alpha_out) memcpy(alpha_out, alpha_h, n * sizeof(double));
    if (want >= 2 && nlZ_out) *nlZ_out = 0.5 * c->res_host[1] / sn2 + c->res_host[0] + 0.5 * (double)n * log(2.0 * M_PI * sn2);
    if (want >= 3 && dnlZ_lik_out) *dnlZ_lik_out = c->res_host[8];
    if (factor_out) {
        FactorHandleGuard hg(c, new pgp_factor());
        pgp_factor* f = hg.f;
        f->n = n; f->np = np; f->ldf = ldf; f->F = fguard.release(); f->dpad = 0; f->d = 0; f->kss = 0.0;
        f->sn2 = sn2; f->sw = 1.0 / sqrt(sn2); f->Wd = nullptr; f->XsT = nullptr;
        CHK(spool_take(c, np * sizeof(double), (void**)&f->alpha));
        HIP_TRY(hipMemsetAsync(f->alpha, 0, np * sizeof(double), st));
        HIP_TRY(hipMemcpyAsync(f->alpha, alpha_h, n * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        *factor_out = hg.release();
    }
    fguard.scrub = false;
    if (want >= 3) { c->dense_ready = true; c->dense_n = n; }
    return PGP_OK;
}

// out = 1/2 sum_ij Q_ij dK_ij with Q = B^-1 / sn2 - alpha alpha' of the pgp_exact_fit_dense call (want = 3, same n, same log_sn)
// that directly precedes it on this context; dK (n,n) symmetric host (Core/inf.py:376-377).
int pgp_dense_grad_term(pgp_ctx* c, const double* dK, int64_t n, double log_sn, double* out) {
    if (!c) return -1;
    GateShared device_gate_hold(c);
    if (!c) return -1;
    if (!dK) return -2;
    if (n <= 0 || round_up(n, 128) != c->ws_np) return -3;
    if (!out) return -5;
    // the Q this term is summed against is what the LAST dense fit with want = 3 left in the workspace: any other fit on this
    // context since then (or none at all) would give a plausible but wrong number
    if (!c->dense_ready || c->dense_n != n) return -6;
    HIP_TRY(hipSetDevice(c->device));
    const long np = c->ws_np;
    hipStream_t st = c->st;
    PoolScratch scr(c);
    double* Dd = nullptr;
    CHK(scr.alloc(&Dd, (size_t)n * n * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(Dd, dK, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dense_hadamard_kernel, dim3((unsigned)n), dim3(256), 0, st, c->Binv, np, c->alpha_dev, Dd, n, exp(-2.0 * log_sn),
                       c->partial);
    hipLaunchKernelGGL(dense_sum_kernel, dim3(1), dim3(256), 0, st, c->partial, n, 0.5, c->scal + 9);
    if (hipGetLastError() != hipSuccess) return PGP_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(out, c->scal + 9, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PGP_OK;
}

}  // extern "C"
