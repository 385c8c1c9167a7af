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
#include <atomic>

#include "common.h"
#include "gemm_tile.h"
#include "kernels.h"

namespace {

constexpr int NB = 128;   // leaf size


__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) and sqrt(d) from v_rsq_f64 + two Newton steps (pure FMA chains, ~1 ulp): the IEEE sqrt and
// divide sequences hipcc emits are ~35 dependent instructions per pivot, the dominant cost of the 16
// sequential pivots.  Pivots of B = K/sn2 + I lie in [1, 1 + sf2/sn2]: no scaling needed.
__device__ __forceinline__ void rsqrt_sqrt(double d, double& rinv, double& s) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    double q = d * r;
    q = fma(fma(-q, q, d), 0.5 * r, q);
    rinv = r;
    s = q;
}

// broadcast lane `K` of each 16-lane row (DPP row_newbcast, 64-bit form on gfx90a+: one VALU op, no SGPR trip)
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    double r;       // asm (not the builtin) so that the wait states stay glued to the DPP read
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}

// acc += (lane K's src) * mul  in ONE instruction: v_fmac_f64 with a DPP row_newbcast source.  hipcc pads no
// hazards inside asm: a DPP read of a VGPR written by the previous VALU op needs 2 wait states, so the FIRST
// use after `src` was produced carries an s_nop 1.
template <int K, bool NOP>
__device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul) {
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}

template <int J, int K>
struct Rank1 {          // a[K] -= a[J] * (lane K's a[J])   for K = J+1 .. 15
    static __device__ __forceinline__ void run(double (&a)[16], double naj) {
        fmac_bcast<K, K == J + 1>(a[K], a[J], naj);
        Rank1<J, K + 1>::run(a, naj);
    }
};
template <int J>
struct Rank1<J, 16> {
    static __device__ __forceinline__ void run(double (&)[16], double) {}
};

template <int J>
struct PivotCol {
    static __device__ __forceinline__ void run(double (&a)[16], double& dinv, int& info, int rl) {
        const double d = row_bcast<J>(a[J]);
        if (!(d > 0.0) && info == 0) info = J + 1;
        double rinv, sq;
        rsqrt_sqrt(d, rinv, sq);
        a[J] = (rl == J) ? sq : a[J] * rinv;
        if (rl == J) dinv = rinv;
        Rank1<J, J + 1>::run(a, -a[J]);
        PivotCol<J + 1>::run(a, dinv, info, rl);
    }
};
template <>
struct PivotCol<16> {
    static __device__ __forceinline__ void run(double (&)[16], double&, int&, int) {}
};

// column `rl` of W = L16^-1 by forward substitution, L16(r,k) fetched from lane r's a[k] by DPP broadcast
template <int R, int Kk>
struct InvDot {
    static __device__ __forceinline__ void run(double& v, const double (&a)[16], const double (&nx)[16]) {
        fmac_bcast<R, Kk == 0>(v, a[Kk], nx[Kk]);
        InvDot<R, Kk + 1>::run(v, a, nx);
    }
};
template <int R>
struct InvDot<R, R> {
    static __device__ __forceinline__ void run(double&, const double (&)[16], const double (&)[16]) {}
};
template <int R>
struct InvRow {
    static __device__ __forceinline__ void run(const double (&a)[16], double dinv, double (&x)[16], double (&nx)[16],
                                               int rl) {
        double v = (rl == R) ? 1.0 : 0.0;
        InvDot<R, 0>::run(v, a, nx);
        x[R] = v * row_bcast<R>(dinv);
        nx[R] = -x[R];
        InvRow<R + 1>::run(a, dinv, x, nx, rl);
    }
};
template <>
struct InvRow<16> {
    static __device__ __forceinline__ void run(const double (&)[16], double, double (&)[16], double (&)[16], int) {}
};

// LDS storage of the leaf: the 36 lower 16x16 blocks, each stored [k][j] (column k, row j: k*16 + j) -- which is
// exactly the MFMA A-operand order, so a fragment read is blockbase[kstep*64 + lane]: one contiguous 512-byte,
// bank-conflict-free run.  Order: 28 strictly-lower blocks (bi > bj: bi(bi-1)/2 + bj), then the 8 diagonal blocks.
// 74 KB: the leaf fits next to one resident gemm_f64 workgroup on a CU (look-ahead overlap).
__device__ __forceinline__ int lblk(int bi, int bj) { return (bi == bj) ? 28 + bi : bi * (bi - 1) / 2 + bj; }

// Wave 0: factor the pivot block tb (each 16-lane row works on its own copy; lane & 15 = row), write the factor
// back to LDS, and its inverse both to LDS (sinv) and to the packed global image (ginv), [k][j] order.
__device__ __forceinline__ void pivot_block(double* __restrict__ s, double* __restrict__ sinv, int* s_info, int tb,
                                            int lane, double* __restrict__ ginv) {
    double a[16];
    double di = 0.0;
    const int rl = lane & 15;
    double* blk = s + lblk(tb, tb) * 256;
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = blk[c * 16 + rl];
    int inf = 0;
    PivotCol<0>::run(a, di, inf, rl);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= lane) blk[c * 16 + lane] = a[c];
    }
    double x[16], nx[16];
    InvRow<0>::run(a, di, x, nx, rl);
    if (lane < 16) {                                   // lane = column k of W: W(j,k) = x[j] -> [k][j]
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const double2_t v = {x[j], x[j + 1]};
            *(double2_t*)(sinv + lane * 16 + j) = v;
            *(double2_t*)(ginv + lane * 16 + j) = v;
        }
    }
    if (lane == 0 && inf != 0 && *s_info == 0) *s_info = tb * 16 + inf;
}

// C(bi,bj) -= P(bi) P(bj)^T with P = column block tb (one 16x16 tile, 4 MFMAs, two accumulation chains)
__device__ __forceinline__ void leaf_tile_update(double* __restrict__ s, int tb, int bi, int bj, int lane) {
    double* cblk = s + lblk(bi, bj) * 256;
    const double* pa = s + lblk(bj, tb) * 256;
    const double* pb = s + lblk(bi, tb) * 256;
    double4_t acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc0[q] = cblk[q * 64 + lane];
    double fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = -pa[ks * 64 + lane];
        fb[ks] = pb[ks * 64 + lane];
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], fb[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], fb[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], fb[3], acc1, 0, 0, 0);
    acc0 += acc1;
#pragma unroll
    for (int q = 0; q < 4; ++q) cblk[q * 64 + lane] = acc0[q];
}

// block (bi, tb):  X <- X W^T  (W = inverted pivot block in sinv), in place
__device__ __forceinline__ void leaf_tile_trsm(double* __restrict__ s, const double* __restrict__ sinv, int tb, int bi,
                                               int lane) {
    double* xb = s + lblk(bi, tb) * 256;
    double fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = sinv[ks * 64 + lane];
        fb[ks] = xb[ks * 64 + lane];
    }
    double4_t y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0], y0, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], fb[1], y1, 0, 0, 0);
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], fb[2], y0, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], fb[3], y1, 0, 0, 0);
    y0 += y1;
#pragma unroll
    for (int q = 0; q < 4; ++q) xb[q * 64 + lane] = y0[q];
}

// thread t, piece i (0..17) of the 36-block image: block, column, row pair
__device__ __forceinline__ void leaf_piece(int t, int i, int& blk, int& bi, int& bj, int& col, int& pr) {
    const int v = t + 256 * i;
    blk = v >> 7;
    col = (v >> 3) & 15;
    pr = v & 7;
    if (blk >= 28) { bi = bj = blk - 28; }
    else {
        bi = (int)((sqrtf(8.0f * blk + 1.0f) + 1.0f) * 0.5f);       // bi(bi-1)/2 <= blk < bi(bi+1)/2
        bj = blk - bi * (bi - 1) / 2;
    }
}

__device__ __forceinline__ void leaf_potrf_body(double* __restrict__ A, long lda, double* __restrict__ pack,
                                                int* __restrict__ info, int info_base, long long* __restrict__ tick,
                                                double* __restrict__ s /* LDS: 36 blocks | sinv[256] | info */) {
#define TICK(i) do { if (tick && threadIdx.x == 0) tick[i] = __builtin_readcyclecounter(); } while (0)
    TICK(0);
    double* sinv = s + 36 * 256;
    int* s_info = (int*)(sinv + 256);                             // keep ALL LDS in the dynamic region
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) *s_info = 0;
    // load the 36 lower blocks in 16-byte pieces; every load is in flight before the first LDS store
    {
        double2_t regs[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            int blk, bi, bj, col, pr;
            leaf_piece(t, i, blk, bi, bj, col, pr);
            regs[i] = *(const double2_t*)(A + (long)(16 * bi + 2 * pr) + (long)(16 * bj + col) * lda);
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) *(double2_t*)(s + 2 * (t + 256 * i)) = regs[i];
    }
    __syncthreads();
    TICK(1);
    if (wave == 0) pivot_block(s, sinv, s_info, 0, lane, pack + 28 * 256);
    __syncthreads();
    TICK(2);

    for (int tb = 0; tb < 8; ++tb) {
        const int nt = 7 - tb;                               // block rows below the pivot block
        // (b) rows below the pivot block: X <- X W^T on the MFMA, one 16-row tile per wave at a time
        for (int i = wave; i < nt; i += 4) leaf_tile_trsm(s, sinv, tb, tb + 1 + i, lane);
        __syncthreads();
        TICK(3 + 2 * tb);
        // (c) trailing update inside the leaf, with look-ahead: wave 0 updates the next pivot block first and
        // factors + inverts it right away while waves 1..3 update the remaining tiles
        if (nt > 0) {
            if (wave == 0) {
                leaf_tile_update(s, tb, tb + 1, tb + 1, lane);
                pivot_block(s, sinv, s_info, tb + 1, lane, pack + (28 + tb + 1) * 256);
            } else {
                const int ntile = nt * (nt + 1) / 2;
                for (int tile = wave; tile < ntile; tile += 3) {      // tile 0 = (0,0) is wave 0's
                    int bi = 0, rem = tile;
                    while (rem > bi) { rem -= bi + 1; ++bi; }
                    leaf_tile_update(s, tb, tb + 1 + bi, tb + 1 + rem, lane);
                }
            }
        }
        __syncthreads();
        TICK(4 + 2 * tb);
    }
    if (tick && t == 0) tick[20] = __builtin_readcyclecounter();
    // write back: the factor into A (lower part only: the strict upper part of the diagonal blocks keeps its
    // exact zeros) and the 28 strictly-lower blocks into the packed operand image for trsm_rows_kernel
    {
        double2_t regs[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) regs[i] = *(const double2_t*)(s + 2 * (t + 256 * i));
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            int blk, bi, bj, col, pr;
            leaf_piece(t, i, blk, bi, bj, col, pr);
            double* dst = A + (long)(16 * bi + 2 * pr) + (long)(16 * bj + col) * lda;
            if (blk < 28) {
                *(double2_t*)dst = regs[i];
                *(double2_t*)(pack + 2 * (t + 256 * i)) = regs[i];
            } else if (2 * pr >= col) {
                *(double2_t*)dst = regs[i];
            } else if (2 * pr + 1 >= col) {
                dst[1] = regs[i][1];
            }
        }
    }
    if (t == 0 && *s_info != 0) atomicCAS(info, 0, info_base + *s_info);
    TICK(19);
#undef TICK
}

__global__ __launch_bounds__(256, 2) void leaf_potrf_kernel(double* __restrict__ A, long lda,
                                                            double* __restrict__ pack, int* __restrict__ info,
                                                            int info_base, long long* __restrict__ tick) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    leaf_potrf_body(A, lda, pack, info, info_base, tick, s);
}

// X (nrows x 128, column-major, ld) <- X * L^-T, L = 128x128 lower at Ld (ld), inv16 = 8 inverted
// pivot blocks (column-major 16x16 each).  One wave per 16 rows.  The 36 lower 16x16 blocks of L and the
// 8 inverted pivot blocks are staged ONCE per workgroup into LDS in MFMA A-operand order
// ([k][j], 16 contiguous doubles per k: a wave's fragment read is one contiguous 512-byte run,
// bank-conflict free), so the solve itself never waits on global memory.
__device__ __forceinline__ int tri_blk(int c, int t) { return c * (c - 1) / 2 + t; }   // strictly lower: c > t

// wg = index of this workgroup among those sharing the rows (4 waves x 16 rows each)
__device__ __forceinline__ void trsm_rows_body(double* __restrict__ X, long ldx, long nrows,
                                               const double* __restrict__ inv16 /* packed image */, long wg,
                                               double* __restrict__ sl /* LDS: 28 strictly-lower L blocks + 8 inverted pivot blocks */) {
    double* si = sl + 28 * 256;
    const int t = threadIdx.x;
    // stage the packed operand image (36 x 256 doubles = 72 KB, written by leaf_potrf_kernel): 18 coalesced
    // 16-byte loads per thread, all in flight together.  72 KB: fits next to one resident gemm_f64 workgroup.
    {
        double2_t regs[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) regs[i] = *(const double2_t*)(inv16 + 2 * (t + 256 * i));
#pragma unroll
        for (int i = 0; i < 18; ++i) *(double2_t*)(sl + 2 * (t + 256 * i)) = regs[i];
    }
    const int lane = t & 63, wave = t >> 6;
    const long r0 = (wg * 4 + wave) * 16;
    const bool active = r0 < nrows;
    const int l15 = lane & 15, l4 = lane >> 4;
    // acc[c][q] = X[r0 + l15][16c + l4 + 4q]
    double4_t acc[8];
    if (active) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[c][q] = X[r0 + l15 + (long)(16 * c + l4 + 4 * q) * ldx];
    }
    __syncthreads();
    if (active) {
    const int fo = l4 * 16 + l15;                 // fragment offset inside a block for kstep 0
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
        // Y^T = inv16_tb * X_tb^T          (A operand: inv16[j = l15][k = 4ks + l4]); two partial chains
        double4_t y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(si[tb * 256 + fo], acc[tb][0], y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(si[tb * 256 + 64 + fo], acc[tb][1], y1, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(si[tb * 256 + 128 + fo], acc[tb][2], y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(si[tb * 256 + 192 + fo], acc[tb][3], y1, 0, 0, 0);
        const double4_t y = y0 + y1;
        acc[tb] = y;
        // X_c^T -= L(c,tb) * Y^T  for c > tb   (A operand: L[16c + l15][16tb + 4ks + l4])
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int c = tb + 1; c < 8; ++c) {
                const double fa = -sl[tri_blk(c, tb) * 256 + ks * 64 + fo];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, y[ks], acc[c], 0, 0, 0);
            }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) X[r0 + l15 + (long)(16 * c + l4 + 4 * q) * ldx] = acc[c][q];
    }
}

__global__ __launch_bounds__(256, 2) void trsm_rows_kernel(double* __restrict__ X, long ldx, long nrows,
                                                           const double* __restrict__ Ld, long ldl,
                                                           const double* __restrict__ inv16 /* packed image */) {
    extern __shared__ __attribute__((aligned(16))) double sl[];
    trsm_rows_body(X, ldx, nrows, inv16, blockIdx.x, sl);
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

int leaf_potrf_launch(double* A, long lda, double* inv16, int* info, int info_base, hipStream_t st,
                      long long* tick) {
    const size_t shm = (36 * 256 + 256 + 2) * sizeof(double);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)leaf_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(leaf_potrf_kernel, dim3(1), dim3(256), shm, st, A, lda, inv16, info, info_base, tick);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int trsm_rows_launch(double* X, long ldx, long nrows, const double* Ld, long ldl, const double* inv16,
                     hipStream_t st) {
    if (nrows <= 0) return PGP_OK;
    const unsigned nblk = (unsigned)((nrows + 63) / 64);
    const size_t shm = 36 * 256 * sizeof(double);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)trsm_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(trsm_rows_kernel, dim3(nblk), dim3(256), shm, st, X, ldx, nrows, Ld, ldl, inv16);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int leaf_inv_launch(const double* L, long ldl, double* W, long ldw, long wstride, int nblocks, hipStream_t st) {
    const size_t shm = (size_t)NB * NB * sizeof(double);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)leaf_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(leaf_inv_kernel, dim3(nblocks), dim3(128), shm, st, L, ldl, W, ldw, wstride);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

namespace {
// Diagonal-panel staging (see potrf_blocked in capi.hip).  The w x w diagonal block of an outer panel is factored on
// its own in a small (2w x w) scratch D: top half = the block (lower part; strict upper zeroed), bottom half = identity,
// so the leaf-level sweep over D leaves  L_D on top and  E_D = L_D^-T (upper triangular) below -- the operand that turns
// the solve of ALL rows below the block into one MFMA GEMM  X <- X E_D.
__global__ __launch_bounds__(256) void diag_in_kernel(const double* __restrict__ src, long lds, double* __restrict__ D,
                                                      long ldd, int w) {
    const int j = blockIdx.x;                          // column
    for (int i = threadIdx.x; i < 2 * w; i += 256) {
        double v;
        if (i < w) v = (i >= j) ? src[i + (long)j * lds] : 0.0;
        else v = (i - w == j) ? 1.0 : 0.0;
        D[i + (long)j * ldd] = v;
    }
}
// L_D (lower part) -> the factor's diagonal block; E_D (whole w x w block, zeros below its diagonal included) -> the
// fused-inverse rows of this panel (E may be null: no global inverse wanted)
// Blocks >= w (present when Dt != null): 64 x 64 tiles of E_D transposed through LDS into Dt(n, k) = E_D(k, n), ld ldt --
// the panel solve Y = X E_D then reads its B operand n-contiguous like every other operand of the LDS-DMA GEMM.
__global__ __launch_bounds__(256) void diag_out_kernel(const double* __restrict__ D, long ldd, int w,
                                                       double* __restrict__ Fd, long ldf, double* __restrict__ Ed,
                                                       long lde, double* __restrict__ Dt, long ldt) {
    if ((int)blockIdx.x >= w) {
        __shared__ double tile[64][65];
        const int tb = blockIdx.x - w, nt = w / 64;
        const int k0 = (tb % nt) * 64, n0 = (tb / nt) * 64;
        const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
        for (int r = b; r < 64; r += 4) tile[r][a] = D[w + k0 + a + (long)(n0 + r) * ldd];      // tile[n][k], k contiguous
        __syncthreads();
        for (int r = b; r < 64; r += 4) Dt[n0 + a + (long)(k0 + r) * ldt] = tile[a][r];         // n contiguous
        return;
    }
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * w; i += 256) {
        const double v = D[i + (long)j * ldd];
        if (i < w) { if (i >= j) Fd[i + (long)j * ldf] = v; }
        else if (Ed) Ed[(i - w) + (long)j * lde] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Resident diagonal-panel server.
//
// The leaf chain of the Cholesky sweep (13 small dependent launches per 512-column panel) loses its CU slots to the
// 150-250 us workgroups of the concurrent trailing update: measured, the first leaf of a panel waited ~160 us for a
// slot, its trsm ~70 us, its inner update ~160 us.  So the chain does not queue at all: ONE kernel of DS_WG workgroups
// is launched per sweep BEFORE any bulk work, stays resident, and factors every diagonal panel block when the main
// stream says it is ready:
//     go[p]   (set by a 1-thread kernel on the main stream after TU_a(p-1))  ->  D(p)  ->  done[p]  (main stream waits)
// Inside, the phases of D(p) (stage in | per leaf: potrf, trsm of the 2w..w rows, K=128 inner update | stage out) are
// separated by a counter barrier over the DS_WG workgroups (agent-scope release / acquire, cdna guide G16).
// Every spin is bounded by a wall-clock timeout: on expiry the server posts an error code and exits.
constexpr int DS_WG = 16;
constexpr unsigned DS_BAR = 0, DS_ERR = 1, DS_GO = 16, DS_MAXP = 1024, DS_DONE = DS_GO + DS_MAXP, DS_CNT = DS_DONE + DS_MAXP,
                   DS_STG = DS_CNT + DS_MAXP, DS_CNT2 = DS_STG + DS_MAXP;

struct DiagServerArgs {
    double* Dk; long dk_stride;       // two scratch images (2w x w each), alternating by panel parity
    double* dpack;
    double* F; long ldf; double* E; long lde;
    const double* Xs; long ldx; long xs_stride;    // two staging buffers (panel p's columns live in buffer p & 1)
    double* Yn;                       // w x w scratch of the next-diagonal-block update
    int nblk, q;
    int fake;                         // experiment: post done[p] at once (no factorisation): times the bulk schedule alone
    unsigned* flags; int* info;
    long long timeout_ticks;          // wall_clock64 ticks (100 MHz)
    long long* ticks;                 // optional: per panel, per phase wall-clock stamps (16 per panel)
};

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 spins until *p >= target (relaxed agent-scope loads), then ONE acquire; false on timeout / posted error
__device__ __forceinline__ bool ds_wait(unsigned* flags, unsigned idx, unsigned target, long long timeout, int* s_ok) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        while (ld_flag(flags + idx) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (ld_flag(flags + DS_ERR) != 0) { ok = 0; break; }
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(flags + DS_ERR, 1000u + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

// barrier over the DS_WG workgroups: every wave drains its stores, one lane releases, arrives, polls, acquires
__device__ __forceinline__ bool ds_barrier(unsigned* flags, unsigned& epoch, long long timeout, int* s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(flags + DS_BAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return ds_wait(flags, DS_BAR, epoch * DS_WG, timeout, s_ok);
}

// The three phase bodies are real calls (noinline): inlined together they need > 256 VGPRs and spill; as separate
// functions each keeps the allocation of its stand-alone kernel (leaf 225, trsm, gemm tile < 256).
__device__ __noinline__ void ds_leaf(double* A, long lda, double* pack, int* info, int info_base, double* smem) {
    leaf_potrf_body(A, lda, pack, info, info_base, nullptr, smem);
}
__device__ __noinline__ void ds_trsm(double* X, long ldx, long nrows, const double* pack, long wg, double* smem) {
    trsm_rows_body(X, ldx, nrows, pack, wg, smem);
}
// K = 128 update of the scratch columns right of a leaf, one wave per 16 rows (no LDS, no workgroup barrier):
//     C[rows, c] -= Y[rows, :] Y[c, :]'      Y = the w solved rows of this leaf (Xw, 128 columns), c = window rows 0..ncols
// in the transposed MFMA form of trsm_rows (D = C^T tile: row = column of C, col = row of C): the B operand is the
// wave's own 16 rows of Y (32 doubles per lane, loaded once), the A operand the 16 rows of Y that belong to the target
// columns -- L2-resident lines shared by every wave.  Window rows < ncols are the symmetric part: only tiles on or
// below the diagonal are touched there.
__device__ __noinline__ void ds_update_rows(const double* __restrict__ Xw, double* __restrict__ C, long ldd, int w,
                                            int ncols, int grp, int part, int nparts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (grp * 4 + wave) * 16;
    if (r0 >= w) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    double yb[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) yb[ks] = Xw[(r0 + l15) + (long)(4 * ks + l4) * ldd];
    const int ngrp = ncols >> 4;
    const int jend = (r0 < ncols) ? (r0 >> 4) + 1 : ngrp;               // symmetric part: column groups up to the diagonal
    // the column groups are dealt round-robin to `nparts` waves; the A fragments (and the C tile) of the NEXT group are
    // loaded while the 32 MFMAs of the current one issue (software pipeline: the loop is latency-bound otherwise)
    double ya[32], yn[32];
    double4_t cn;
    auto fetch = [&](int jg, double (&dst)[32], double4_t& c) {
        const int c0 = 16 * jg;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) dst[ks] = Xw[(c0 + l15) + (long)(4 * ks + l4) * ldd];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = C[(r0 + l15) + (long)(c0 + l4 + 4 * q) * ldd];
    };
    if (part < jend) fetch(part, yn, cn);
    for (int jg = part; jg < jend; jg += nparts) {
        const int c0 = 16 * jg;
        double4_t a0 = cn, b0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) ya[ks] = -yn[ks];
        if (jg + nparts < jend) fetch(jg + nparts, yn, cn);
#pragma unroll
        for (int ks = 0; ks < 32; ks += 2) {            // two independent accumulation chains
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ks], yb[ks], a0, 0, 0, 0);
            b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ks + 1], yb[ks + 1], b0, 0, 0, 0);
        }
        a0 += b0;
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(r0 + l15) + (long)(c0 + l4 + 4 * q) * ldd] = a0[q];
    }
}

// stage in: top = lower part of the (updated) diagonal block, bottom = identity.  Workgroup -> columns wg, wg + DS_WG,
// ...; 8 columns (= 8 independent 16-byte loads per thread) per round: the loop is latency-bound.  (Own functions, like
// the phase bodies: inlined into the kernel next to the calls they were compiled with a handful of registers and spilled.)
__device__ __noinline__ void ds_stage_in(const double* __restrict__ src, long lds, double* __restrict__ Dk, long ldd, int w,
                                         int wg) {
    const int t = threadIdx.x;
    for (int jb = wg; jb < w; jb += 8 * DS_WG)
        for (int i = 2 * t; i < w; i += 512) {
            double2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jb + u * DS_WG;
                v[u] = double2_t{0.0, 0.0};
                if (j < w && i + 1 >= j) v[u] = *(const double2_t*)(src + i + (long)j * lds);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jb + u * DS_WG;
                if (j >= w) continue;
                double2_t o = v[u];
                if (i < j) o[0] = 0.0;
                *(double2_t*)(Dk + i + (long)j * ldd) = o;
                *(double2_t*)(Dk + w + i + (long)j * ldd) = double2_t{i == j ? 1.0 : 0.0, i + 1 == j ? 1.0 : 0.0};
            }
        }
}
// stage out: L_D (lower part) -> factor, E_D (whole block, zeros below its diagonal included) -> inverse rows
__device__ __noinline__ void ds_stage_out(const double* __restrict__ Dk, long ldd, int w, double* __restrict__ Fd, long ldf,
                                          double* __restrict__ Ed, long lde, int wg) {
    const int t = threadIdx.x;
    for (int jb = wg; jb < w; jb += 4 * DS_WG)
        for (int i = 2 * t; i < w; i += 512) {
            double2_t lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + u * DS_WG;
                lo[u] = hi[u] = double2_t{0.0, 0.0};
                if (j < w) {
                    if (i + 1 >= j) lo[u] = *(const double2_t*)(Dk + i + (long)j * ldd);
                    if (Ed && i <= j) hi[u] = *(const double2_t*)(Dk + w + i + (long)j * ldd);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + u * DS_WG;
                if (j >= w) continue;
                if (i >= j) *(double2_t*)(Fd + i + (long)j * ldf) = lo[u];
                else if (i + 1 >= j) Fd[i + 1 + (long)j * ldf] = lo[u][1];
                if (Ed) {
                    double2_t o = hi[u];
                    if (i + 1 > j) o[1] = 0.0;
                    *(double2_t*)(Ed + i + (long)j * lde) = o;
                }
            }
        }
}

// The server's own GEMM tiles (64 x 64, the latency-friendly shape): the diagonal block of the NEXT panel is brought up
// to date here, on the server's CUs, instead of waiting for the bulk trailing update to get to it:
//   (a) Yn = X E_D(p-1)            X = rows of panel p in panel p-1's (staged) columns          [K clipped: k < j0 + 64]
//   (b) D_top = F_diag(p) - Yn Yn'  lower tiles only; F_diag(p) carries every older panel's update already
__device__ __noinline__ void ds_tile_nk(const GemmArgs* g, int ti, int tj, double* smem) {
    gemm_tile_ns::gemm_tile<128, 128, false, true>(*g, ti, tj, 0, smem);
}
__device__ __noinline__ void ds_tile_nt(const GemmArgs* g, int ti, int tj, double* smem) {
    gemm_tile_ns::gemm_tile<128, 128, false, false>(*g, ti, tj, 0, smem);
}
__device__ __noinline__ void ds_identity_bottom(double* __restrict__ Dk, long ldd, int w, int wg) {
    const int t = threadIdx.x;
    for (int j = wg; j < w; j += DS_WG)
        for (int i = 2 * t; i < w; i += 512)
            *(double2_t*)(Dk + w + i + (long)j * ldd) = double2_t{i == j ? 1.0 : 0.0, i + 1 == j ? 1.0 : 0.0};
}

__global__ __launch_bounds__(256, 2) void diag_server_kernel(DiagServerArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int* s_ok = (int*)(smem + 36 * 256 + 256 + 1);            // beside the leaf's info word (all LDS stays dynamic)
    const int wg = blockIdx.x, t = threadIdx.x;
    const int npanel = (a.nblk + a.q - 1) / a.q;
    unsigned epoch = 0;
    // the chain is the critical path and shares its CUs with trailing-update workgroups: win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
#define STAMP(i) do { if (a.ticks && wg == 0 && t == 0) a.ticks[(long)p * 16 + (i)] = wall_clock64(); } while (0)
    for (int p = 0; p < npanel; ++p) {
        const int s0 = p * a.q, s1 = min(s0 + a.q, a.nblk), qq = s1 - s0, w = qq * 128;
        const long ldd = 2L * w;
        double* Dc = a.Dk + (long)(p & 1) * a.dk_stride;
        STAMP(0);
        if (a.fake) {
            if (p == 0 && !ds_wait(a.flags, DS_GO, 1u, a.timeout_ticks, s_ok)) return;
            if (p >= 2 && !ds_wait(a.flags, DS_GO + p, 1u, a.timeout_ticks, s_ok)) return;
            if (wg == 0 && t == 0) __hip_atomic_store(a.flags + DS_DONE + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (p == 0) {
            if (!ds_wait(a.flags, DS_GO, 1u, a.timeout_ticks, s_ok)) return;
            STAMP(1);
            ds_stage_in(a.F, a.ldf, Dc, ldd, w, wg);
            STAMP(7);
            if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
        } else {
            // go[p] (p >= 2) is released from INSIDE trailing update TU(p-2) when its first two column panels are done:
            // rows p x columns p-1 (staged) and the diagonal block p (in place) then carry every update up to panel p-2
            if (p >= 2 && !ds_wait(a.flags, DS_GO + p, 1u, a.timeout_ticks, s_ok)) return;
            STAMP(1);
            const int wp = a.q * 128;                           // panel p-1 is a full panel
            const double* Dp = a.Dk + (long)((p - 1) & 1) * a.dk_stride;
            {
                GemmArgs g{};
                g.A = a.Xs + (long)((p - 1) & 1) * a.xs_stride + (long)s0 * 128; g.lda = a.ldx; g.a_kc = 0;
                g.B = Dp + wp; g.ldb = 2L * wp; g.b_kc = 1;
                g.C = a.Yn; g.ldc = w;
                g.M = w; g.N = wp; g.K = wp; g.alpha = 1.0; g.beta = 0.0; g.kmode = KM_LT_J;
                // 128 x 128 tiles (a 64 x 64 tile needs ~1.3 us per 16-deep k-step on a CU of its own: latency, not MFMA),
                // longest k-range first: one tile per workgroup when w = 512
                const int mt = w / 128, nt = wp / 128;
                for (int tile = wg; tile < mt * nt; tile += DS_WG) ds_tile_nk(&g, tile % mt, nt - 1 - tile / mt, smem);
            }
            ds_identity_bottom(Dc, ldd, w, wg);
            STAMP(7);
            if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
            {
                GemmArgs g{};
                g.A = a.Yn; g.lda = w; g.a_kc = 0;
                g.B = a.Yn; g.ldb = w; g.b_kc = 0;
                g.Cin = a.F + (long)s0 * 128 * (1 + a.ldf); g.ldcin = a.ldf;
                g.C = Dc; g.ldc = ldd;
                g.M = w; g.N = w; g.K = wp; g.alpha = -1.0; g.beta = 1.0;
                g.tri = 1; g.tri_off = 0; g.mask_diag = 1; g.kmode = KM_FULL;
                const int mt = w / 128;
                int act = 0;
                for (int tj = 0; tj < mt; ++tj)
                    for (int ti = tj; ti < mt; ++ti, ++act)
                        if (act % DS_WG == wg) ds_tile_nt(&g, ti, tj, smem);
            }
            if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
        }
        STAMP(2);
        // (Measured and rejected: in-panel look-ahead -- workgroup 0 factoring leaf cb+1 while the others apply the
        //  update of leaf cb -- 320 us per panel instead of 296: the deferred update on 15 workgroups is slower than a leaf.)
        for (int cb = 0; cb < qq; ++cb) {
            double* Acc = Dc + (long)cb * 128 * (1 + ldd);
            double* pack = a.dpack + (long)cb * PACK_DOUBLES;
            if (wg == 0) ds_leaf(Acc, ldd, pack, a.info, (s0 + cb) * 128, smem);
            if (cb == 0) STAMP(8);
            if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
            if (cb == 0) STAMP(9);
            // rows below the leaf inside the scratch: (qq-1-cb) 128 block rows + (cb+1) 128 identity-born rows = w rows
            for (long g0 = wg; g0 * 64 < w; g0 += DS_WG) {
                ds_trsm(Acc + 128, ldd, w, pack, g0, smem);
                __syncthreads();
            }
            if (cb == 0) STAMP(10);
            if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
            if (cb == 0) STAMP(11);
            if (cb + 1 < qq) {                               // K = 128 update of the scratch columns right of the leaf
                // row groups of 64 (4 waves x 16 rows); the column groups of one row group are shared by `np_` workgroups
                const int ngr = w / 64, np_ = max(1, DS_WG / ngr);
                for (int u = wg; u < ngr * np_; u += DS_WG)
                    ds_update_rows(Acc + 128, Dc + (long)(cb + 1) * 128 * (1 + ldd), ldd, w, (qq - 1 - cb) * 128, u % ngr,
                                   u / ngr, np_);
                if (cb == 0) STAMP(12);
                if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;
            }
            STAMP(3 + cb);
        }
        ds_stage_out(Dc, ldd, w, a.F + (long)s0 * 128 * (1 + a.ldf), a.ldf,
                     a.E ? a.E + (long)s0 * 128 * (1 + a.lde) : nullptr, a.lde, wg);
        if (!ds_barrier(a.flags, epoch, a.timeout_ticks, s_ok)) return;      // every store released and arrived
        if (wg == 0 && t == 0) __hip_atomic_store(a.flags + DS_DONE + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        STAMP(15);
    }
#undef STAMP
}

__global__ void ds_signal_kernel(unsigned* flag) {
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// main stream: hold the stream until the server has finished D(p) (or posted an error / timed out)
__global__ void ds_wait_kernel(unsigned* flags, unsigned idx, long long timeout) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (ld_flag(flags + idx) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (ld_flag(flags + DS_ERR) != 0) break;
        if (wall_clock64() - t0 > timeout) {
            __hip_atomic_store(flags + DS_ERR, 2000u + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
}  // namespace

size_t diag_server_flag_bytes() { return (size_t)(DS_CNT2 + DS_MAXP) * sizeof(unsigned); }
unsigned* diag_server_counter2(unsigned* flags, int p) { return flags + DS_CNT2 + p; }
unsigned* diag_server_stage_flag(unsigned* flags, int p) { return flags + DS_STG + p; }
int diag_server_stage_index(int p) { return (int)(DS_STG + p); }
unsigned* diag_server_counter(unsigned* flags, int p) { return flags + DS_CNT + p; }
unsigned* diag_server_go_flag(unsigned* flags, int p) { return flags + DS_GO + p; }
int diag_server_max_panels() { return (int)DS_MAXP; }

int diag_server_launch(double* Dk, long dk_stride, double* dpack, double* F, long ldf, double* E, long lde,
                       const double* Xs, long ldx, long xs_stride, double* Yn, int nblk, int q, unsigned* flags, int* info,
                       double timeout_s, long long* ticks, hipStream_t st, bool exclusive, int fake) {
    DiagServerArgs a{Dk, dk_stride, dpack, F, ldf, E, lde, Xs, ldx, xs_stride, Yn, nblk, q, fake, flags, info,
                     (long long)(timeout_s * 1e8), ticks};
    // 96 KB of LDS although the phases need 74: a server workgroup then has its CU to itself (a 74 KB trailing-update
    // workgroup no longer fits beside it).  Sharing the CU slowed every phase 2-2.7x (leaf 34 -> 90 us): LDS + MFMA pipe.
    const size_t shm = exclusive ? 96 * 1024 : (36 * 256 + 256 + 2) * sizeof(double);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute((const void*)diag_server_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(diag_server_kernel, dim3(DS_WG), dim3(256), shm, st, a);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_server_post(unsigned* flag, hipStream_t st) {
    hipLaunchKernelGGL(ds_signal_kernel, dim3(1), dim3(1), 0, st, flag);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_server_go(unsigned* flags, int p, hipStream_t st) {
    hipLaunchKernelGGL(ds_signal_kernel, dim3(1), dim3(1), 0, st, flags + DS_GO + p);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_server_wait(unsigned* flags, int p, double timeout_s, hipStream_t st) {
    hipLaunchKernelGGL(ds_wait_kernel, dim3(1), dim3(64), 0, st, flags, DS_DONE + (unsigned)p, (long long)(timeout_s * 1e8));
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
// hold the stream until the staged columns of panel p are complete (released from inside trailing update TU(p-1))
int diag_server_wait_staged(unsigned* flags, int p, double timeout_s, hipStream_t st) {
    hipLaunchKernelGGL(ds_wait_kernel, dim3(1), dim3(64), 0, st, flags, DS_STG + (unsigned)p, (long long)(timeout_s * 1e8));
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_server_err_index() { return (int)DS_ERR; }

namespace {
}  // namespace

int diag_in_launch(const double* src, long lds, double* D, long ldd, int w, hipStream_t st) {
    hipLaunchKernelGGL(diag_in_kernel, dim3(w), dim3(256), 0, st, src, lds, D, ldd, w);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_out_launch(const double* D, long ldd, int w, double* Fd, long ldf, double* Ed, long lde, hipStream_t st,
                    double* Dt, long ldt) {
    const int extra = Dt ? (w / 64) * (w / 64) : 0;
    hipLaunchKernelGGL(diag_out_kernel, dim3(w + extra), dim3(256), 0, st, D, ldd, w, Fd, ldf, Ed, lde, Dt, ldt);
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
