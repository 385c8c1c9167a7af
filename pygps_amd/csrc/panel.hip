// Latency-critical panel kernels of the blocked right-looking Cholesky (gfx950):
//
//   leaf_potrf_kernel   one workgroup factors a 128x128 diagonal block held in LDS, 16 columns at
//                       a time: a single wave factors the 16x16 pivot block in registers with
//                       v_readlane broadcasts (the "wavefront shuffle" diagonal panel), the rows
//                       below are solved lane-per-row, the in-LDS trailing block is updated with
//                       v_mfma_f64_16x16x4_f64.  Also emits the eight inverted 16x16 pivot blocks.
//   trsm_rows_kernel    X <- X L^-T for the rows below the leaf.  One wave owns 16 rows for all 128
//                       columns and keeps them in MFMA accumulator registers for the whole solve:
//                       the f64 16x16x4 D layout (row = lane/16 + 4*reg) is exactly the B-operand
//                       layout (k = lane/16 + 4*kstep), so the chain
//                           Y_t^T = inv(L_tt) X_t^T ;  X_c^T -= L_ct Y_t^T
//                       never leaves registers: no LDS, no barriers, 144 MFMAs per wave.
//   leaf_inv_kernel     batched inverse of every 128x128 diagonal block of L (level 0 of trtri and
//                       the diagonal solves of trsv), off the critical path.
//
// Storage: column-major lower ("L(i,j) at i + j*ld"), which is the same memory as numpy's
// row-major UPPER factor R = L^T that pyGPs stores in post.L (Core/inf.py:362,367).
#include "common.h"
#include "kernels.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int NB = 128;   // leaf size
constexpr int LS = 144;   // LDS column stride (doubles): 2*LS = 32 mod 64 banks

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Factor the 16x16 block whose row `lane` is held in a[0..15] (lanes 0..15 of the calling wave).
// Returns 0 or the 1-based index of the first non-positive pivot.  On return a[] holds row `lane`
// of the lower factor (entries right of the diagonal are junk) and dinv = 1 / L(lane,lane).
__device__ __forceinline__ int potf2_16_rows(double (&a)[16], double& dinv, int lane) {
    int info = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = readlane_f64(a[j], j);
        if (!(d > 0.0) && info == 0) info = j + 1;
        const double s = sqrt(d);
        const double rinv = 1.0 / s;
        a[j] = (lane == j) ? s : a[j] * rinv;
        if (lane == j) dinv = rinv;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
            const double lkj = readlane_f64(a[j], k);
            a[k] = fma(-a[j], lkj, a[k]);
        }
    }
    return info;
}

__global__ __launch_bounds__(256, 2) void leaf_potrf_kernel(double* __restrict__ A, long lda,
                                                            double* __restrict__ inv16, int* __restrict__ info,
                                                            int info_base) {
    extern __shared__ __attribute__((aligned(16))) double s[];   // s[c*LS + r], + dinv[128]
    double* dinv = s + NB * LS;
    int& s_info = *(int*)(dinv + NB);                             // keep ALL LDS in the dynamic region
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_info = 0;
    // load the lower triangle (upper part of the block is never referenced)
    for (int c = wave; c < NB; c += 4)
        for (int r = lane; r < NB; r += 64) s[c * LS + r] = (r >= c) ? A[(long)r + (long)c * lda] : 0.0;
    __syncthreads();

    for (int tb = 0; tb < 8; ++tb) {
        const int o = tb * 16;
        // (a) pivot block: wave 0, lanes 0..15 hold one row each
        if (wave == 0) {
            double a[16];
            double di = 0.0;
            const int rl = lane & 15;
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = s[(o + c) * LS + o + rl];
            const int inf = potf2_16_rows(a, di, lane);     // lanes >= 16 compute junk on row copies
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c <= lane) s[(o + c) * LS + o + lane] = a[c];
                dinv[o + lane] = di;
            }
            if (lane == 0 && inf != 0 && s_info == 0) s_info = o + inf;
        }
        __syncthreads();
        // (b) rows below the pivot block: x <- x L16^-T, one lane per row
        const int nrow = NB - o - 16;
        if (t < nrow) {
            const int r = o + 16 + t;
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = s[(o + c) * LS + r];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double v = x[j];
#pragma unroll
                for (int k = 0; k < j; ++k) v = fma(-x[k], s[(o + k) * LS + o + j], v);
                x[j] = v * dinv[o + j];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) s[(o + c) * LS + r] = x[c];
        }
        __syncthreads();
        // (c) trailing update inside the leaf: C(bi,bj) -= P(bi) P(bj)^T for tb < bj <= bi
        const int nt = 7 - tb;                               // remaining block rows
        const int ntile = nt * (nt + 1) / 2;
        const int l15 = lane & 15, l4 = lane >> 4;
        for (int tile = wave; tile < ntile; tile += 4) {
            int bi = 0, rem = tile;
            while (rem > bi) { rem -= bi + 1; ++bi; }
            const int bj = rem;                              // 0 <= bj <= bi < nt
            const int ri = o + 16 + 16 * bi, rj = o + 16 + 16 * bj;
            double4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = s[(rj + l4 + 4 * q) * LS + ri + l15];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = o + 4 * ks + l4;
                const double fa = -s[k * LS + rj + l15];
                const double fb = s[k * LS + ri + l15];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s[(rj + l4 + 4 * q) * LS + ri + l15] = acc[q];
        }
        __syncthreads();
    }
    // write back the factor (lower part; the strict upper part of the block stays untouched = 0)
    for (int c = wave; c < NB; c += 4)
        for (int r = lane; r < NB; r += 64)
            if (r >= c) A[(long)r + (long)c * lda] = s[c * LS + r];
    // inverted 16x16 pivot blocks: thread (blk = t/16, col = t%16) for t < 128 solves L16 x = e_col
    if (t < 128) {
        const int blk = t >> 4, c = t & 15, o = blk * 16;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) v = fma(-s[(o + k) * LS + o + r], x[k], v);
            x[r] = (r < c) ? 0.0 : v * dinv[o + r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) inv16[blk * 256 + c * 16 + r] = x[r];   // column-major 16x16
    }
    if (t == 0 && s_info != 0) atomicCAS(info, 0, info_base + s_info);
}

// X (nrows x 128, column-major, ld) <- X * L^-T, L = 128x128 lower at Ld (ld), inv16 = 8 inverted
// pivot blocks (column-major 16x16 each).  One wave per 16 rows.
__global__ __launch_bounds__(256, 2) void trsm_rows_kernel(double* __restrict__ X, long ldx, long nrows,
                                                           const double* __restrict__ Ld, long ldl,
                                                           const double* __restrict__ inv16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 16;
    if (r0 >= nrows) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    // acc[c][q] = X[r0 + l15][16c + l4 + 4q]
    double4_t acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[c][q] = X[r0 + l15 + (long)(16 * c + l4 + 4 * q) * ldx];
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
        // Y^T = inv16_tb * X_tb^T          (A operand: inv16[j = l15][k = 4ks + l4])
        double4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const double fa = inv16[tb * 256 + (4 * ks + l4) * 16 + l15];
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, acc[tb][ks], y, 0, 0, 0);
        }
        acc[tb] = y;
        // X_c^T -= L(c,tb) * Y^T  for c > tb   (A operand: L[16c + l15][16tb + 4ks + l4])
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = tb + 1; c < 8; ++c) {
                const double fa = -Ld[(long)(16 * c + l15) + (long)(16 * tb + 4 * ks + l4) * ldl];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, y[ks], acc[c], 0, 0, 0);
            }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[r0 + l15 + (long)(16 * c + l4 + 4 * q) * ldx] = acc[c][q];
}

// W_kk = inv(L_kk) for every 128x128 diagonal block k (blockIdx.x), written into W (same layout).
// Column c of the inverse is the forward substitution L x = e_c; 128 independent columns ->
// threads 0..127, coefficients broadcast from LDS.
__global__ __launch_bounds__(128, 1) void leaf_inv_kernel(const double* __restrict__ L, long ldl,
                                                          double* __restrict__ Wb, long ldw, long wstride) {
    extern __shared__ __attribute__((aligned(16))) double sx[];  // sx[r*NB + c]: column c owned by thread c
    const long o = (long)blockIdx.x * NB;
    const int c = threadIdx.x;
    const double* __restrict__ Lb = L + o + o * ldl;
    for (int r = 0; r < NB; ++r) {
        double v = (r == c) ? 1.0 : 0.0;
        // uniform loop bounds keep the waves converged; L(r,k) is wave-uniform (scalar load)
        for (int k = 0; k < r; ++k) v = fma(-Lb[r + (long)k * ldl], sx[k * NB + c], v);
        sx[r * NB + c] = (r < c) ? 0.0 : v / Lb[r + (long)r * ldl];
    }
    __syncthreads();
    // transpose through LDS so the global stores run along r (contiguous)
    double* __restrict__ W = Wb + (long)blockIdx.x * wstride;
    for (int cc = 0; cc < NB; ++cc) W[c + (long)cc * ldw] = sx[c * NB + cc];
}

}  // namespace

int leaf_potrf_launch(double* A, long lda, double* inv16, int* info, int info_base, hipStream_t st) {
    const size_t shm = (NB * LS + NB + 2) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)leaf_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    hipLaunchKernelGGL(leaf_potrf_kernel, dim3(1), dim3(256), shm, st, A, lda, inv16, info, info_base);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int trsm_rows_launch(double* X, long ldx, long nrows, const double* Ld, long ldl, const double* inv16,
                     hipStream_t st) {
    if (nrows <= 0) return PGP_OK;
    const unsigned nblk = (unsigned)((nrows + 63) / 64);
    hipLaunchKernelGGL(trsm_rows_kernel, dim3(nblk), dim3(256), 0, st, X, ldx, nrows, Ld, ldl, inv16);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int leaf_inv_launch(const double* L, long ldl, double* W, long ldw, long wstride, int nblocks, hipStream_t st) {
    const size_t shm = (size_t)NB * NB * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)leaf_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = true;
    }
    hipLaunchKernelGGL(leaf_inv_kernel, dim3(nblocks), dim3(128), shm, st, L, ldl, W, ldw, wstride);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

namespace {
// Backward substitution step kb of  L^T a = z  (L column-major lower, W_kk = inv(L_kk) on W's diagonal):
//   a_kb = W_kk^T z_kb ;  z_j -= L(kb, j)^T a_kb  for every column block j < kb  (blockIdx.x = j <= kb).
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(const double* __restrict__ L, long ldl,
                                                            const double* __restrict__ W, long ldw,
                                                            double* __restrict__ z, double* __restrict__ a_out,
                                                            int kb) {
    __shared__ double a[NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long o = (long)kb * NB;
    const int j = blockIdx.x;
    const double z0 = z[o + lane], z1 = z[o + 64 + lane];
    for (int c = wave; c < NB; c += 4) {
        const double* col = W + o + (o + c) * ldw;
        double v = (lane >= c ? col[lane] * z0 : 0.0) + (lane + 64 >= c ? col[lane + 64] * z1 : 0.0);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (lane == 0) a[c] = v;
    }
    __syncthreads();
    if (j == kb) {
        if (threadIdx.x < NB) a_out[o + threadIdx.x] = a[threadIdx.x];
        return;
    }
    const double a0 = a[lane], a1 = a[lane + 64];
    const long oj = (long)j * NB;
    for (int c = wave; c < NB; c += 4) {
        const double* col = L + o + (oj + c) * ldl;
        double v = col[lane] * a0 + col[lane + 64] * a1;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (lane == 0) z[oj + c] -= v;
    }
}

__global__ void gather_strided_kernel(const double* __restrict__ src, long stride, long n, double* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i * stride];
}
}  // namespace

int trsv_bwd_launch(const double* L, long ldl, const double* W, long ldw, double* z, double* a_out, int nblk,
                    hipStream_t st) {
    for (int kb = nblk - 1; kb >= 0; --kb)
        hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(kb + 1), dim3(256), 0, st, L, ldl, W, ldw, z, a_out, kb);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int gather_strided_launch(const double* src, long stride, long n, double* dst, hipStream_t st) {
    hipLaunchKernelGGL(gather_strided_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, stride, n, dst);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
