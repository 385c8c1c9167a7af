// Self-test / calibration hooks (tests/ and bench.py only): host-buffer GEMM through the MFMA kernel,
// and an fp64 MFMA issue-rate micro-benchmark used to restate the roofline peak from measurement.
#include <time.h>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "ctx.h"
#include <chrono>
#include <thread>

#include "testhooks.h"

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
        int rc = leaf_potrf_launch(Ad, 128, pk, info, 0, c->st, tk, nullptr, c->leaf_pivot);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->st));
    }
    long long h[24];
    HIP_TRY(hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 24; ++i) ticks_out[i] = (double)h[i];
    (void)hipFree(Ad); (void)hipFree(pk); (void)hipFree(info); (void)hipFree(tk);
    return PGP_OK;
}

// What single instructions cost ONE workgroup of 4 waves on an otherwise idle chip (the situation of the latency-bound chain
// kernels): s_memtime ticks per step of (0) a dependent v_fma_f64 chain, (2) eight independent chains, (3) a dependent LDS
// read chain, (4) 20 independent LDS reads + one use, (5) s_barrier, (6) LDS write -> barrier -> read, (7) a dependent
// v_rcp_f64 chain, (8) dependent v_rsq_f64; out[1] = s_memrealtime (100 MHz) ticks of test 0, i.e. the s_memtime rate.
namespace {
__global__ __launch_bounds__(256) void wave_costs_kernel(double* __restrict__ out, double seed) {
    __shared__ double lds[512];
    __shared__ int nxt[256];
    const int t = threadIdx.x;
    lds[t] = seed + t; lds[t + 256] = seed - t; nxt[t] = (t * 7 + 1) & 255;
    __syncthreads();
    long long a, b, ra, rb;
    double x = seed, acc = 0.0;
    a = __builtin_amdgcn_s_memtime(); ra = __builtin_amdgcn_s_memrealtime();
#pragma unroll 16
    for (int i = 0; i < 2048; ++i) x = fma(x, 0.999999, 1e-9);
    acc += x;
    asm volatile("" :: "v"(x));
    b = __builtin_amdgcn_s_memtime(); rb = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { out[0] = (double)(b - a) / 2048.0; out[1] = (double)(rb - ra); out[9] = (double)(b - a); }
    double y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = seed + q;
    a = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) y[q] = fma(y[q], 0.999999, 1e-9);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += y[q];
    asm volatile("" :: "v"(acc));
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[2] = (double)(b - a) / 2048.0;
    int idx = t;
    a = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) idx = nxt[idx];
    asm volatile("" :: "v"(idx));
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[3] = (double)(b - a) / 512.0;
    acc += idx;
    a = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) {
        double sum = 0.0;
        const int o = (i & 1) * 256;
#pragma unroll
        for (int q = 0; q < 20; ++q) sum += lds[o + ((t >> 4) + 16 * q) % 256];
        acc = fma(sum, 1e-9, acc);
        asm volatile("" :: "v"(acc));
    }
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[4] = (double)(b - a) / 512.0;
    a = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) __syncthreads();
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[5] = (double)(b - a) / 512.0;
    a = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512; ++i) {
        if ((t & 15) == (i & 15)) lds[(i & 1) * 256 + (t >> 4)] = acc + i;
        __syncthreads();
        acc += lds[(i & 1) * 256 + (i & 15)];
        asm volatile("" :: "v"(acc));
    }
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[6] = (double)(b - a) / 512.0;
    x = seed + 2.0;
    a = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < 512; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
    asm volatile("" :: "v"(x));
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[7] = (double)(b - a) / 512.0;
    acc += x;
    x = seed + 2.0;
    a = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < 512; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
    asm volatile("" :: "v"(x));
    b = __builtin_amdgcn_s_memtime();
    if (t == 0) out[8] = (double)(b - a) / 512.0;
    acc += x;
    if (acc == 12345.678 && t == 0) out[15] = acc;
}
}  // namespace
int pgp_test_wave_costs(pgp_ctx* c, double* out16) {
    if (!c || !out16) return -1;
    HIP_TRY(hipSetDevice(c->device));
    double* od;
    HIP_TRY(hipMalloc((void**)&od, 16 * 8));
    HIP_TRY(hipMemset(od, 0, 16 * 8));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(wave_costs_kernel, dim3(1), dim3(256), 0, c->st, od, 1.0 + 1e-3 * rep);
    HIP_TRY(hipStreamSynchronize(c->st));
    HIP_TRY(hipMemcpy(out16, od, 16 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(od);
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
    // the form a fit would pick for this data (uniform in +-1.7, unit length scales sqrt(d): squared norms ~1): the Gram form on the
    // matrix cores for RBF / RBFard at d >= 32 (option gram_assembly 0: the difference form).  The means / norms are part of the cost.
    const bool gram = c->gram_assembly && cov_gram_applies(cp, dpad);
    double* prep = nullptr;
    if (gram) HIP_TRY(hipMalloc((void**)&prep, (size_t)hadamard_prep_count(np) * 8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = PGP_OK;
    // warm-up = at least 100 launches: the first 30-50 ms of device work after an idle spell run up to 25 % slower (clocks ramping:
    // 10 launches right after process start read 0.55 ms, the next tens 0.47 / 0.45, steady state 0.43: tools/first_call.py)
    for (int it = -std::max(100, iters); it < iters && rc == PGP_OK; ++it) {
        if (it == 0) HIP_TRY(hipEventRecord(e0, c->st));
        if (gram) {
            rc = hadamard_prepare_launch(XT, np, n, np, dpad, cp, prep, c->st, true);
            if (rc == PGP_OK) rc = mode == 2 ? cov_factor_gram_launch(XT, np, n, np, dpad, cp, 100.0, out, ldo, prep, c->st)
                                             : cov_sym_gram_launch(XT, np, n, dpad, cp, out, 0, prep, c->st);
        } else
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
    if (prep) (void)hipFree(prep);
    return rc;
}

// What the box gives a kernel that does NOTHING but the stores of the 'train' assembly: the same 64 x 64 tiles of the upper
// triangle in the same super-tile order, each written twice (tile + mirror, 512-byte runs), persistent workgroups -- no
// coordinates, no distances, no exp.  out3[0] = ms per launch of that kernel, out3[1] = ms of hipMemsetAsync over the same
// 8 n^2 bytes, out3[2] = ms of a linear fill with one 16-byte store per thread.  The assembly kernel's HBM fraction is read
// against out3[0]: no symmetric tile pattern measured in tools/store_roof.hip writes faster (0.65-0.69 of 8 TB/s box to box).
namespace {
__global__ __launch_bounds__(256) void store_only_sym_kernel(double* out, long n, const int2* tab, long ntiles, double v) {
    const int t = threadIdx.x;
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int2 ij = tab[ti];
        double* base = out + (long)ij.x * 64 * n + (long)ij.y * 64;
        double* mir = out + (long)ij.y * 64 * n + (long)ij.x * 64;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int e = t + 256 * p, r = e >> 5, cpair = e & 31;
            *(double2_t*)(base + (long)r * n + 2 * cpair) = double2_t{v, v + 1.0};
        }
        if (ij.x != ij.y) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int e = t + 256 * p, r = e >> 5, cpair = e & 31;
                *(double2_t*)(mir + (long)r * n + 2 * cpair) = double2_t{v, v + 1.0};
            }
        }
    }
}
__global__ __launch_bounds__(256) void store_only_linear_kernel(double* out, long n2, double v) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n2) *(double2_t*)(out + i) = double2_t{v, v + 1.0};
}
}  // namespace

int pgp_test_store_roof(pgp_ctx* c, int64_t n, int grid, int iters, double* out3) {
    if (!c || !out3 || n <= 0 || n % 64 || iters <= 0) return -1;
    HIP_TRY(hipSetDevice(c->device));
    const long nt = n / 64, S = 8, nst = (nt + S - 1) / S;
    std::vector<int2> h;
    for (long SI = 0; SI < nst; ++SI)
        for (long SJ = SI; SJ < nst; ++SJ)
            for (long i = SI * S; i < std::min(nt, SI * S + S); ++i)
                for (long j = std::max(i, SJ * S); j < std::min(nt, SJ * S + S); ++j) h.push_back(make_int2((int)i, (int)j));
    int2* tab; double* out;
    HIP_TRY(hipMalloc((void**)&tab, h.size() * sizeof(int2)));
    HIP_TRY(hipMalloc((void**)&out, (size_t)n * n * 8));
    HIP_TRY(hipMemcpy(tab, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const long ntl = (long)h.size(), n2 = (long)n * n;
    const unsigned g = (unsigned)std::min<long>(ntl, grid > 0 ? grid : 4096);
    for (int which = 0; which < 3; ++which) {
        for (int it = -std::max(50, iters / 2); it < iters; ++it) {
            if (it == 0) HIP_TRY(hipEventRecord(e0, c->st));
            if (which == 0) hipLaunchKernelGGL(store_only_sym_kernel, dim3(g), dim3(256), 0, c->st, out, (long)n, tab, ntl, 1.0);
            else if (which == 1) HIP_TRY(hipMemsetAsync(out, 0, (size_t)n2 * 8, c->st));
            else hipLaunchKernelGGL(store_only_linear_kernel, dim3((unsigned)((n2 / 2 + 255) / 256)), dim3(256), 0, c->st, out, n2, 1.0);
        }
        HIP_TRY(hipEventRecord(e1, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        out3[which] = ms / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(tab); (void)hipFree(out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// Does a small panel kernel on the high-priority stream overlap a big trailing GEMM on the main stream?
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
    if (getenv("PGP_TEST_GEMM_FOLD")) g.fold_rows = 1;                 // (tools: the two-tile-rows-per-workgroup kernel on the same arguments)
    hipStream_t ts = c->st;
    int rc = gemm_f64_launch(g, ts);
    HIP_TRY(hipStreamSynchronize(ts));
    if (rc == PGP_OK) HIP_TRY(hipMemcpy(C, Cd, cn * 8, hipMemcpyDeviceToHost));
    if (rc == PGP_OK && iters > 0 && ms_out) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        // warm-up: ~80 ms of the same launches.  The shader clock of an idle chip starts near 2.0 GHz and takes tens of milliseconds
        // of load to reach 2.37 (tools/gemm_trace.py: the K = 512 product 57 TF after 5 launches, 68 TF after 100); round 4's first
        // stand-alone tables were taken cold
        {
            HIP_TRY(hipEventRecord(e0, ts));
            rc = gemm_f64_launch(g, ts);
            HIP_TRY(hipEventRecord(e1, ts));
            HIP_TRY(hipStreamSynchronize(ts));
            float m1 = 0.f;
            HIP_TRY(hipEventElapsedTime(&m1, e0, e1));
            const int nwarm = (int)std::min(4000.0, std::max(1.0, 80.0 / std::max(1e-3, (double)m1)));
            for (int i = 0; i < nwarm && rc == PGP_OK; ++i) rc = gemm_f64_launch(g, ts);
        }
        HIP_TRY(hipEventRecord(e0, ts));
        for (int i = 0; i < iters; ++i) rc = gemm_f64_launch(g, ts);
        HIP_TRY(hipEventRecord(e1, ts));
        HIP_TRY(hipStreamSynchronize(ts));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(Ad); (void)hipFree(Bd); (void)hipFree(Cd);
    return rc;
}

// Phase stamps of every workgroup of ONE launch of the trailing-update product C -= A A' (M x M, depth K, synthetic operands, after
// `warm` untraced launches; `conc` > 0: a second, untraced launch of the same product runs beside it on the other stream).
// out: 8 words per workgroup (GemmArgs::trace), *nblk_out workgroups.
int pgp_test_gemm_trace(pgp_ctx* c, int M, int K, int tri, int warm, int conc, long long* out, int64_t out_words, int64_t* nblk_out) {
    if (!c || !out || !nblk_out || M % 128 || K % 16) return -1;
    HIP_TRY(hipSetDevice(c->device));
    const long mt = M / 128;
    const long nblk = tri ? mt * (mt + 1) / 2 : mt * mt;
    if (out_words < 8 * nblk) return -2;
    double *A, *Cd, *C2; long long* tr;
    HIP_TRY(hipMalloc((void**)&A, (size_t)M * K * 8)); HIP_TRY(hipMalloc((void**)&Cd, (size_t)M * M * 8));
    HIP_TRY(hipMalloc((void**)&C2, (size_t)M * M * 8)); HIP_TRY(hipMalloc((void**)&tr, (size_t)nblk * 64));
    HIP_TRY(hipMemset(tr, 0, (size_t)nblk * 64));
    std::vector<double> h((size_t)M * K);
    unsigned long long s = 88172645463325252ULL;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.01; }
    HIP_TRY(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(Cd, 0, (size_t)M * M * 8)); HIP_TRY(hipMemset(C2, 0, (size_t)M * M * 8));
    GemmArgs g{};
    g.A = A; g.lda = M; g.B = A; g.ldb = M; g.C = Cd; g.ldc = M; g.M = M; g.N = M; g.K = K; g.alpha = -1.0; g.beta = 1.0;
    g.tri = tri ? 2 : 0; g.mask_diag = tri ? 1 : 0; g.batch = 1; g.tile = 128; g.dbg = c->gemm_dbg;
    int rc = PGP_OK;
    for (int i = 0; i < warm && rc == PGP_OK; ++i) rc = gemm_f64_launch(g, c->st);
    HIP_TRY(hipStreamSynchronize(c->st));
    GemmArgs g2 = g; g2.C = C2;
    if (conc > 0 && rc == PGP_OK) rc = gemm_f64_launch(g2, c->st2);
    g.trace = tr;
    if (rc == PGP_OK) rc = gemm_f64_launch(g, c->st);
    HIP_TRY(hipStreamSynchronize(c->st)); HIP_TRY(hipStreamSynchronize(c->st2));
    HIP_TRY(hipMemcpy(out, tr, (size_t)nblk * 64, hipMemcpyDeviceToHost));
    *nblk_out = nblk;
    (void)hipFree(A); (void)hipFree(Cd); (void)hipFree(C2); (void)hipFree(tr);
    return rc;
}

// the stamps recorded since the last call (option "gemm_trace"): 8 words per workgroup, launch after launch; returns the number of
// workgroups through *nwg and rewinds the recorder
int pgp_test_read_gemm_trace(pgp_ctx* c, long long* out, int64_t words, int64_t* nwg) {
    if (!c || !out || !nwg || !c->gemm_trace) return -1;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    const long n = std::min<long>(c->gemm_trace_pos, words / 8);
    HIP_TRY(hipMemcpy(out, c->gemm_trace, (size_t)n * 64, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->gemm_trace, 0, (size_t)c->gemm_trace_cap * 64));
    c->gemm_trace_pos = 0;
    *nwg = n;
    return PGP_OK;
}

// GemmArgs::skip_lo / skip_hi and GemmArgs::wait_flag (EP's block sweep): n x n lower-triangular update, column-major host buffers
int pgp_test_gemm_skip_wait(pgp_ctx* ctx, int tile, const double* A, const double* B, double* C, int n, int K, int skip_lo,
                            int skip_hi, int wait_ms, int* timed_out) {
    if (!ctx) return -1;
    pgp_ctx* c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    DevScratch scr;
    double *Ad = nullptr, *Bd = nullptr, *Cd = nullptr;
    unsigned* fl = nullptr;
    const size_t an = (size_t)n * K * 8, cn = (size_t)n * n * 8;
    CHK(scr.alloc(&Ad, an)); CHK(scr.alloc(&Bd, an)); CHK(scr.alloc(&Cd, cn));
    CHK(scr.alloc(&fl, 16));
    HIP_TRY(hipMemcpy(Ad, A, an, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(Bd, B, an, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(Cd, C, cn, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(fl, 0, 16));
    GemmArgs g{};
    g.A = Ad; g.lda = n; g.B = Bd; g.ldb = n; g.C = Cd; g.ldc = n; g.M = n; g.N = n; g.K = K; g.alpha = -1.0; g.beta = 1.0;
    g.tri = 2; g.mask_diag = 1; g.batch = 1; g.tile = tile; g.dbg = c->gemm_dbg; g.skip_lo = skip_lo; g.skip_hi = skip_hi;
    if (wait_ms > 0) { g.wait_flag = fl; g.wait_target = 7u; g.wait_err = fl + 1; }
    int rc = gemm_f64_launch(g, c->st);
    if (rc == PGP_OK && wait_ms > 0) {
        std::this_thread::sleep_for(std::chrono::milliseconds(wait_ms));
        const unsigned seven = 7u;
        HIP_TRY(hipMemcpyAsync(fl, &seven, sizeof(seven), hipMemcpyHostToDevice, c->st2));     // the other stream raises the counter
        HIP_TRY(hipStreamSynchronize(c->st2));
    }
    HIP_TRY(hipStreamSynchronize(c->st));
    unsigned flh[2] = {0u, 0u};
    HIP_TRY(hipMemcpy(flh, fl, sizeof(flh), hipMemcpyDeviceToHost));
    if (timed_out) *timed_out = (int)flh[1];
    if (rc == PGP_OK) HIP_TRY(hipMemcpy(C, Cd, cn, hipMemcpyDeviceToHost));
    return rc;
}

// Batched trailing update whose products shrink with the batch index (GemmArgs::batch_dm; the owned column panels of the
// block-cyclic sweep in csrc/sharded.hip): for z < nb
//     C_z (M - z dm rows x w, at C + z sC)  -=  Y[z dm : M, :] Y[z dm : z dm + w, :]'     lower trapezoid, diagonal tile masked,
// rows >= zero_from - z dm of C_z taken as zero on input (first touch).  Host buffers, column-major.
int pgp_test_gemm_shrink(pgp_ctx* ctx, const double* Y, int64_t ldy, int M, int K, int w, int nb, int dm, int zero_from,
                         double* C, int64_t ldc, int64_t sC) {
    if (!ctx) return -1;
    pgp_ctx* c = ctx;
    if (!Y || !C || M <= 0 || M % 128 || w % 128 || K % 16 || dm % 128 || nb < 1) return -2;
    HIP_TRY(hipSetDevice(c->device));
    DevScratch scr;
    double *yd, *cd;
    const size_t cbytes = (size_t)((nb - 1) * sC + ldc * w) * 8;
    CHK(scr.alloc(&yd, (size_t)ldy * K * 8)); CHK(scr.alloc(&cd, cbytes));
    HIP_TRY(hipMemcpy(yd, Y, (size_t)ldy * K * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(cd, C, cbytes, hipMemcpyHostToDevice));
    GemmArgs g{};
    g.A = yd; g.lda = ldy; g.B = yd; g.ldb = ldy; g.C = cd; g.ldc = ldc; g.M = M; g.N = w; g.K = K;
    g.alpha = -1.0; g.beta = 1.0; g.tri = 1; g.mask_diag = 1; g.kmode = KM_FULL; g.tile = 128;
    g.batch = nb; g.sA = dm; g.sB = dm; g.sC = sC; g.batch_dm = dm; g.zero_from = zero_from;
    CHK(gemm_prof(c, PC_GEMM_TRAIL, g, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    HIP_TRY(hipMemcpy(C, cd, cbytes, hipMemcpyDeviceToHost));
    return PGP_OK;
}

}  // extern "C"

// ---- dispatcher experiment (tools/slot_probe.py): does a small high-LDS kernel on a second stream find the CUs that a resident
// grid leaves empty?  `holder`: nwg workgroups of 256 threads and lds_kb KB of LDS spin for hold_us; with reserve > 0 every
// workgroup that finds itself on the first-claimed CU of its XCD leaves at once.  `probe`: one workgroup (probe_lds_kb) launched
// on the panel stream delay_us later; it records when it actually started.  out: [0] probe start - probe launch (us, device
// wall clock), [1] holder workgroups that stayed, [2] holder workgroups that left, [3] probe start - holder start (us).
namespace {
__device__ unsigned g_probe_claim[16];
__device__ unsigned g_probe_cnt[4];
__device__ long long g_probe_t[4];
__device__ unsigned g_probe_log[4 * 2048];
__global__ void holder_kernel(long long hold_ticks, int reserve) {
    extern __shared__ double hs[];
    __shared__ int leave;
    if (threadIdx.x == 0) {
        int mine = 0;
        if (reserve > 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;
            const unsigned key = 1u + ((hw >> 8) & 0xffu);
            const unsigned prev = atomicCAS(&g_probe_claim[xcc], 0u, key);
            mine = prev == 0u || prev == key;
        }
        leave = mine;
        atomicAdd(&g_probe_cnt[mine ? 1 : 0], 1u);
        if (blockIdx.x < 2048) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
            g_probe_log[4 * blockIdx.x] = hw; g_probe_log[4 * blockIdx.x + 1] = xcc; g_probe_log[4 * blockIdx.x + 2] = mine;
            g_probe_log[4 * blockIdx.x + 3] = (unsigned)(wall_clock64() & 0xffffffffu);
        }
        if (blockIdx.x == 0) g_probe_t[0] = wall_clock64();
    }
    __syncthreads();
    if (leave) return;
    hs[threadIdx.x] = 1.0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(20);
}
__global__ void probe_kernel() {
    extern __shared__ double ps[];
    if (threadIdx.x == 0) { g_probe_t[1] = wall_clock64(); ps[0] = 1.0; }
}
__global__ void probe_mark_kernel() { if (threadIdx.x == 0) g_probe_t[2] = wall_clock64(); }
}  // namespace

extern "C" int pgp_test_slot_probe(pgp_ctx* c, int nwg, int lds_kb, int hold_us, int reserve, int probe_lds_kb, int delay_us,
                                   double* out4) {
    if (!c || !out4) return -1;
    HIP_TRY(hipSetDevice(c->device));
    unsigned zero16[16] = {0}; unsigned zero4[4] = {0}; long long zt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_claim), zero16, sizeof(zero16)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_cnt), zero4, sizeof(zero4)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_t), zt, sizeof(zt)));
    (void)hipFuncSetAttribute((const void*)holder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(holder_kernel, dim3(nwg), dim3(256), (size_t)lds_kb * 1024, c->st, (long long)hold_us * 100, reserve);
    // a marker kernel on the panel stream `delay_us` later (host sleep), then the probe right behind it
    struct timespec ts = {0, (long)delay_us * 1000};
    nanosleep(&ts, nullptr);
    hipLaunchKernelGGL(probe_mark_kernel, dim3(1), dim3(64), 0, c->st2);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), (size_t)probe_lds_kb * 1024, c->st2);
    HIP_TRY(hipDeviceSynchronize());
    unsigned cnt[4]; long long t[4];
    HIP_TRY(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_probe_cnt), sizeof(cnt)));
    HIP_TRY(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_probe_t), sizeof(t)));
    out4[0] = (double)(t[1] - t[2]) / 100.0; out4[1] = cnt[0]; out4[2] = cnt[1]; out4[3] = (double)(t[1] - t[0]) / 100.0;
    if (getenv("PGP_PROBE_DUMP")) {
        static unsigned lg[4 * 2048];
        HIP_TRY(hipMemcpyFromSymbol(lg, HIP_SYMBOL(g_probe_log), sizeof(lg)));
        for (int i = 0; i < nwg && i < 2048; ++i)
            fprintf(stderr, "wg %4d hw %08x cu %2u sh %u se %u xcc %08x leave %u t %u\n", i, lg[4 * i], (lg[4 * i] >> 8) & 15u, (lg[4 * i] >> 12) & 1u,
                    (lg[4 * i] >> 13) & 7u, lg[4 * i + 1], lg[4 * i + 2], lg[4 * i + 3] - lg[3]);
    }
    return PGP_OK;
}

// ---- CU-mask experiment (tools/cumask_probe.py): the K-deep trailing-update-shaped GEMM (M = N, lower trapezoid off) on a
// stream created with hipExtStreamCreateWithCUMask.  reserve_per_xcd CUs of every XCD are masked OUT (0 = a full mask); the
// mask word order is probed by the caller through `stride` (bit i of the mask = CU i in the runtime's enumeration; CUs of one
// XCD are `stride` apart).  out: ms per launch on the masked stream, ms per launch on the plain stream.
extern "C" int pgp_test_cumask_gemm(pgp_ctx* c, int M, int K, int reserve_per_xcd, int stride, int iters, double* out2) {
    if (!c || !out2) return -1;
    HIP_TRY(hipSetDevice(c->device));
    const int ncu = c->prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    int kept = 0;
    for (int i = 0; i < ncu; ++i) {
        // XCD of CU i = i % stride' ... two enumerations are tried by the caller: stride = 8 (interleaved: CU i on XCD i % 8,
        // index within the XCD i / 8) or stride = 1 (blocked: XCD i / 32, index i % 32)
        const int within = stride == 8 ? i / 8 : i % (ncu / 8);
        const bool keep = within >= reserve_per_xcd;
        if (keep) { mask[i / 32] |= 1u << (i % 32); ++kept; }
    }
    hipStream_t ms = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&ms, (uint32_t)mask.size(), mask.data()));
    DevScratch scr;
    double *A, *Cm;
    CHK(scr.alloc(&A, (size_t)M * K * 8)); CHK(scr.alloc(&Cm, (size_t)M * M * 8));
    HIP_TRY(hipMemset(A, 0, (size_t)M * K * 8)); HIP_TRY(hipMemset(Cm, 0, (size_t)M * M * 8));
    GemmArgs g{};
    g.A = A; g.lda = M; g.B = A; g.ldb = M; g.C = Cm; g.ldc = M; g.M = M; g.N = M; g.K = K;
    g.alpha = -1.0; g.beta = 1.0; g.tile = 128; g.batch = 1; g.dbg = c->gemm_dbg;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipStream_t sts[2] = {ms, c->st};
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 3; ++i) CHK(gemm_f64_launch(g, sts[s]));
        HIP_TRY(hipStreamSynchronize(sts[s]));
        HIP_TRY(hipEventRecord(e0, sts[s]));
        for (int i = 0; i < iters; ++i) CHK(gemm_f64_launch(g, sts[s]));
        HIP_TRY(hipEventRecord(e1, sts[s]));
        HIP_TRY(hipStreamSynchronize(sts[s]));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        out2[s] = t / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(ms);
    return kept;
}

// Do the two streams of a context run CONCURRENTLY right now?  A one-wave kernel on the panel stream spins (bounded: wait_us) until a
// one-thread kernel on the main stream, launched after it, has set a flag.  out2[0] = 1 if the flag arrived, out2[1] = microseconds the
// spinner waited.  (EP's block sweep depends on exactly this: a resident kernel on st2 meets bulk launches on st.)
__global__ void pgp_probe_spin_kernel(unsigned* flag, long long wait_ticks, long long* out) {
    const long long t0 = (long long)wall_clock64();
    long long t = t0;
    unsigned seen = 0u;
    while ((t = (long long)wall_clock64()) - t0 < wait_ticks) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { seen = 1u; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    if (threadIdx.x == 0) { out[0] = (long long)seen; out[1] = t - t0; }
}
__global__ void pgp_probe_set_kernel(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
extern "C" int pgp_test_stream_concurrency(pgp_ctx* c, int wait_us, double* out2) {
    if (!c || !out2) return -1;
    HIP_TRY(hipSetDevice(c->device));
    DevScratch scr;
    double* buf;
    CHK(scr.alloc(&buf, 64));
    HIP_TRY(hipMemset(buf, 0, 64));
    unsigned* flag = (unsigned*)buf;
    long long* res = (long long*)(buf + 2);
    hipLaunchKernelGGL(pgp_probe_spin_kernel, dim3(1), dim3(64), 0, c->st2, flag, (long long)wait_us * 100LL, res);   // 100 MHz wall clock
    hipLaunchKernelGGL(pgp_probe_set_kernel, dim3(1), dim3(1), 0, c->st, flag);
    HIP_TRY(hipStreamSynchronize(c->st2));
    HIP_TRY(hipStreamSynchronize(c->st));
    long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost));
    out2[0] = (double)h[0]; out2[1] = (double)h[1] / 100.0;
    return PGP_OK;
}
