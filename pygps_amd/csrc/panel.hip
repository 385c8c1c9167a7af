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
#include <type_traits>

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

// The same pivot block on the matrix cores (round 5).  The lane-per-row version above is bound by the issue rate of ONE
// wave's 240 v_fmac_f64_dpp + 250 other fp64 instructions; this one needs 18 MFMAs and ~400 fp64 instructions.
// V[q](lane) = M(row l15, col 4q + l4) is the block in the f64 16x16x4 accumulator layout (valid lower triangle; the
// updates are symmetric, the upper triangle is never read).  Per panel t of 4 columns:
//   * the 4x4 diagonal block comes out of reg t by v_readlane (10 values), its factor L_tt and inverse W_tt are computed
//     redundantly by every lane (uniform values);
//   * the UNSCALED panel Praw[r][k] = M(r, 4t + k) is reg t AS IT STANDS in the B-operand layout (lane (l15, l4) holds
//     row l15, panel column l4), so P = Praw W_tt^T is ONE mfma with A = W_tt repeated down the 16 rows -- and its
//     output lands in the A-operand layout P[l15][l4];
//   * the rank-4 update of all 16 columns is ONE mfma with A = -P, B = P (the same register).
// The inverse rides along as the rows of an appended identity (like the 2w x w scratch of the outer sweep): U holds
// G^T (G -> L^-T) in the same layout, two more MFMAs per panel; four MFMAs against identity slices turn it into the
// [k][j] image of W = L^-1 that leaf_tile_trsm / trsm_rows_kernel read.
// 1/sqrt(d) from v_rsq_f64 (24 bits) and ONE third-order step  r (1 + e/2 + 3 e^2/8),  e = 1 - d r^2  (measured on gfx950,
// tools/rsq_probe.hip: 1.24 ulp against 2.14 for two Newton steps, four dependent operations instead of six); sqrt(d) = d r
// with one correction.
__device__ __forceinline__ void rsqrt_sqrt3(double d, double& rinv, double& s) {
    const double r0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * r0), r0, 1.0);
    const double r = fma(r0 * e, fma(0.375, e, 0.5), r0);
    const double q = d * r;
    rinv = r;
    s = fma(fma(-q, q, d), 0.5 * r, q);
}

// The core: V (the block in accumulator layout, by value) -> info (0, or 1 + the first column whose pivot is not positive),
// Wo = the [k][j] image of W = L^-1 (Wo[q](lane) = W(row l15, col 4q + l4)); store_panel(t, v) receives the factor's columns
// 4t .. 4t+3 in the A-operand layout (lane (l15, l4): L(l15, 4t + l4), exact zeros above the diagonal).
template <class StoreL>
__device__ __forceinline__ int pivot16_mfma(double4_t V, int lane, double4_t& Wo, StoreL&& store_panel) {
    const int l15 = lane & 15, l4 = lane >> 4, m4 = l15 & 3, g4 = l15 >> 2;
    double4_t U;
#pragma unroll
    for (int q = 0; q < 4; ++q) U[q] = (l15 == 4 * q + l4) ? 1.0 : 0.0;
    const bool r0m = m4 == 0, r1m = m4 == 1, r2m = m4 == 2;
    const bool c0m = l4 == 0, c1m = l4 == 1, c2m = l4 == 2;
    int inf = 0;
    const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    // Panels of FOUR columns: a dependent mfma -> VALU round trip costs ~300 cycles on this chip, so the fewer the better.  The
    // column-by-column form (one rank-1 mfma per column, no 4 x 4 arithmetic, no scaling product) was built and measured:
    // 7070 cycles per block against 5340 (EXPERIMENTS.md).
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        // M(4t + a, 4t + b), a >= b: reg t of lane 16 b + 4t + a
        const double s00 = readlane_f64(V[t], 4 * t + 0), s10 = readlane_f64(V[t], 4 * t + 1);
        const double s20 = readlane_f64(V[t], 4 * t + 2), s30 = readlane_f64(V[t], 4 * t + 3);
        const double s11 = readlane_f64(V[t], 16 + 4 * t + 1), s21 = readlane_f64(V[t], 16 + 4 * t + 2);
        const double s31 = readlane_f64(V[t], 16 + 4 * t + 3);
        const double s22 = readlane_f64(V[t], 32 + 4 * t + 2), s32 = readlane_f64(V[t], 32 + 4 * t + 3);
        const double s33 = readlane_f64(V[t], 48 + 4 * t + 3);
        double rr0, rr1, rr2, rr3, l00, l11, l22, l33;
        if (!(s00 > 0.0) && inf == 0) inf = 4 * t + 1;
        rsqrt_sqrt3(s00, rr0, l00);
        const double l10 = s10 * rr0, l20 = s20 * rr0, l30 = s30 * rr0;
        const double d1 = fma(-l10, l10, s11);
        if (!(d1 > 0.0) && inf == 0) inf = 4 * t + 2;
        rsqrt_sqrt3(d1, rr1, l11);
        const double l21 = fma(-l20, l10, s21) * rr1, l31 = fma(-l30, l10, s31) * rr1;
        const double d2 = fma(-l21, l21, fma(-l20, l20, s22));
        if (!(d2 > 0.0) && inf == 0) inf = 4 * t + 3;
        rsqrt_sqrt3(d2, rr2, l22);
        const double l32 = fma(-l31, l21, fma(-l30, l20, s32)) * rr2;
        const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33)));
        if (!(d3 > 0.0) && inf == 0) inf = 4 * t + 4;
        rsqrt_sqrt3(d3, rr3, l33);
        // W_tt = L_tt^-1
        const double w10 = -(l10 * rr0) * rr1, w21 = -(l21 * rr1) * rr2, w32 = -(l32 * rr2) * rr3;
        const double w20 = -fma(l21, w10, l20 * rr0) * rr2;
        const double w31 = -fma(l32, w21, l31 * rr1) * rr3;
        const double w30 = -fma(l32, w20, fma(l31, w10, l30 * rr0)) * rr3;
        // A operand of the scaling products: W_tt[l15 & 3][l4] (repeated down the rows: every output register is a copy)
        const double wc0 = r0m ? rr0 : (r1m ? w10 : (r2m ? w20 : w30));
        const double wc1 = r0m ? 0.0 : (r1m ? rr1 : (r2m ? w21 : w31));
        const double wc2 = r2m ? rr2 : ((m4 == 3) ? w32 : 0.0);
        const double wc3 = (m4 == 3) ? rr3 : 0.0;
        const double wrep = c0m ? wc0 : (c1m ? wc1 : (c2m ? wc2 : wc3));
        const double4_t op = __builtin_amdgcn_mfma_f64_16x16x4f64(wrep, V[t], zero4, 0, 0, 0);   // P[l15][l4]
        const double4_t og = __builtin_amdgcn_mfma_f64_16x16x4f64(wrep, U[t], zero4, 0, 0, 0);   // (G W_tt^T)[l15][l4]
        U[t] = og[0];
        // rows of the diagonal block: the factor's own entries (diagonal from the uniform values, exact zeros above)
        const double ldiag = r0m ? l00 : (r1m ? l11 : (r2m ? l22 : l33));
        const double lrow = (l4 < m4) ? op[0] : ((l4 == m4) ? ldiag : 0.0);
        const bool below = g4 > t;
        const double pupd = below ? op[0] : 0.0;
        store_panel(t, below ? op[0] : ((g4 == t) ? lrow : 0.0));          // column 4t + l4, row l15
        if (t < 3) {
            V = __builtin_amdgcn_mfma_f64_16x16x4f64(-pupd, pupd, V, 0, 0, 0);
            U = __builtin_amdgcn_mfma_f64_16x16x4f64(-pupd, og[0], U, 0, 0, 0);
        }
    }
    // U[q](l15, l4) = G[l15][4q + l4] = W[4q + l4][l15]  ->  [k][j] image of W: Wo[q](l15, l4) = W(row l15, col 4q + l4)
    Wo = zero4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double ik = (l15 == 4 * ks + l4) ? 1.0 : 0.0;
        Wo = __builtin_amdgcn_mfma_f64_16x16x4f64(U[ks], ik, Wo, 0, 0, 0);
    }
    return inf;
}

__device__ __forceinline__ void pivot_block_mfma(double* __restrict__ s, double* __restrict__ sinv, int* s_info, int tb,
                                                 int lane, double* __restrict__ ginv) {
    double* blk = s + lblk(tb, tb) * 256;
    double4_t V, Wo;
#pragma unroll
    for (int q = 0; q < 4; ++q) V[q] = blk[q * 64 + lane];
    const int inf = pivot16_mfma(V, lane, Wo, [&](int t, double v) { blk[64 * t + lane] = v; });
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sinv[q * 64 + lane] = Wo[q];
        ginv[q * 64 + lane] = Wo[q];
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

template <bool MP>     // MP: the pivot blocks on the matrix cores (pivot_block_mfma)
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
    if (wave == 0) {
        if (MP) pivot_block_mfma(s, sinv, s_info, 0, lane, pack + 28 * 256);
        else pivot_block(s, sinv, s_info, 0, lane, pack + 28 * 256);
    }
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
                if (MP) pivot_block_mfma(s, sinv, s_info, tb + 1, lane, pack + (28 + tb + 1) * 256);
                else pivot_block(s, sinv, s_info, tb + 1, lane, pack + (28 + tb + 1) * 256);
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

template <bool MP>
__global__ __launch_bounds__(256, 2) void leaf_potrf_kernel(double* __restrict__ A, long lda,
                                                            double* __restrict__ pack, int* __restrict__ info,
                                                            int info_base, long long* __restrict__ tick, unsigned* yield_flags) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    pgp_yield_mark(yield_flags, +1);                 // the bulk workgroups on this CU give way while the leaf runs (gemm_tile.h)
    leaf_potrf_body<MP>(A, lda, pack, info, info_base, tick, s);
    __syncthreads();
    pgp_yield_mark(yield_flags, -1);
}

// ---------------------------------------------------------------------------------------------------------------------
// leaf_potrf_reg_kernel (round 5): the same factorisation with the 128 x 128 block REGISTER-RESIDENT.  Every 16 x 16 tile
// lives in the MFMA accumulator registers of the wave that owns it, from the global load to the global store:
//   wave 0        the 8 diagonal tiles -- and the pivot blocks (pivot16_mfma works on the registers as they stand);
//   waves 1 .. 3  the 28 tiles below the diagonal, tile (bi, bj) on wave 1 + (bi + bj) % 3 (2-3 panel tiles and an equal
//                 share of the trailing tiles per wave in every step).
// LDS holds only FINAL blocks of L (the operand images the updates read) and the current inverted pivot block: no staging
// of the matrix, no round trip of C through LDS per update (the LDS version spends ~1100 cycles per tile update on them),
// 59 KB instead of 76 KB.  The solve X <- X W^T of a panel tile uses the accumulator registers directly as B operand
// (the trsm_rows_kernel trick).  Wave 0 brings only the NEXT diagonal tile up to date before it factors it; the other
// diagonal tiles receive a column's update one phase later, while waves 1 .. 3 solve the next panel.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
constexpr int rl_owner(int bi, int bj) { return bi == bj ? 0 : 1 + (bi + bj) % 3; }
constexpr int rl_slot(int bi, int bj) {                // position of (bi, bj) among its owner's tiles
    const int w = rl_owner(bi, bj);
    int n = 0;
    for (int j = 0; j < 8; ++j)
        for (int i = j; i < 8; ++i) {
            if (i == bi && j == bj) return n;
            if (rl_owner(i, j) == w) ++n;
        }
    return -1;
}
constexpr int rl_tile(int w, int slot) {                // 8 bi + bj of wave w's slot (slots beyond its last tile repeat it)
    int last = 0;
    for (int j = 0; j < 8; ++j)
        for (int i = j; i < 8; ++i)
            if (rl_owner(i, j) == w) {
                last = 8 * i + j;
                if (rl_slot(i, j) == slot) return last;
            }
    return last;
}
constexpr int RL_NT = 10;                              // most tiles on one wave
constexpr int rl_blk(int bi, int bj) { return bi * (bi - 1) / 2 + bj; }     // strictly lower blocks: the packed image's order

template <class F>
__device__ __forceinline__ void role_dispatch(int wave, F&& f) {
    if (wave == 0) f(std::integral_constant<int, 0>{});
    else if (wave == 1) f(std::integral_constant<int, 1>{});
    else if (wave == 2) f(std::integral_constant<int, 2>{});
    else f(std::integral_constant<int, 3>{});
}

__global__ __launch_bounds__(256, 2) void leaf_potrf_reg_kernel(double* __restrict__ A, long lda, double* __restrict__ pack,
                                                                int* __restrict__ info, int info_base,
                                                                long long* __restrict__ tick, unsigned* yield_flags) {
    extern __shared__ __attribute__((aligned(16))) double s[];          // 28 final blocks | sinv[256]
    pgp_yield_mark(yield_flags, +1);
#define TICK(i) do { if (tick && threadIdx.x == 0) tick[i] = __builtin_readcyclecounter(); } while (0)
    TICK(0);
    double* sinv = s + 28 * 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double4_t acc[RL_NT];
    int binfo = 0;                                                       // wave 0: first bad pivot (1-based column of the leaf)
    // ---- load: every wave its own tiles, straight into the accumulator layout (reg q: row l15, column 4q + l4).  ONE code path
    // for the four roles, the tile of (wave, slot) picked at run time: stores into acc[] from four role branches are sunk
    // into one store through a pointer phi by the optimiser, and the array then stays in scratch
    static_for<0, RL_NT>([&](auto slc) __attribute__((always_inline)) {
        constexpr int sl = decltype(slc)::value;
        constexpr int c0 = rl_tile(0, sl), c1 = rl_tile(1, sl), c2 = rl_tile(2, sl), c3 = rl_tile(3, sl);
        const int code = wave == 0 ? c0 : (wave == 1 ? c1 : (wave == 2 ? c2 : c3));
        const int bi = code >> 3, bj = code & 7;
        const double* src = A + (long)(16 * bi + l15) + (long)(16 * bj + l4) * lda;
        const double4_t v = {src[0], src[4 * lda], src[8 * lda], src[12 * lda]};
        acc[sl] = v;
    });
    TICK(1);
    // the pivot block of column block `tb` on wave 0's registers: factor -> global (lower part), inverse -> LDS + packed image
    auto pivot = [&](auto tbc) __attribute__((always_inline)) {
        constexpr int tb = decltype(tbc)::value;
        double* Ad = A + (long)(16 * tb) * (1 + lda);
        double4_t Wo;
        const int inf = pivot16_mfma(acc[tb], lane, Wo, [&](int tt, double v) __attribute__((always_inline)) {
            if (l15 >= 4 * tt + l4) Ad[l15 + (long)(4 * tt + l4) * lda] = v;
        });
        double* ginv = pack + (28 + tb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sinv[q * 64 + lane] = Wo[q];
            ginv[q * 64 + lane] = Wo[q];
        }
        if (inf != 0 && binfo == 0) binfo = tb * 16 + inf;
    };
    // acc -= P(bi, c) P(bj, c)^T from the final blocks of column block c in LDS (bi == bj: one operand image)
    auto update = [&](double4_t& a, auto bic, auto bjc, auto cc) __attribute__((always_inline)) {
        constexpr int bi = decltype(bic)::value, bj = decltype(bjc)::value, c = decltype(cc)::value;
        const double* pa = s + rl_blk(bj, c) * 256;
        const double* pb = s + rl_blk(bi, c) * 256;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[ks * 64 + lane], pb[ks * 64 + lane], a, 0, 0, 0);
    };
    if (wave == 0) pivot(std::integral_constant<int, 0>{});
    __syncthreads();                                                     // B1(0): the inverted pivot block is in LDS
    TICK(2);
    static_for<0, 7>([&](auto tbc) __attribute__((always_inline)) {
        constexpr int tb = decltype(tbc)::value;
        // ---- b: waves 1..3 solve their tiles of panel column tb; wave 0 applies column tb-1 to the diagonal tiles behind the next
        role_dispatch(wave, [&](auto Wc) __attribute__((always_inline)) {
            constexpr int W = decltype(Wc)::value;
            if constexpr (W == 0) {
                if constexpr (tb >= 1)
                    static_for<tb + 1, 8>([&](auto kc) __attribute__((always_inline)) { update(acc[decltype(kc)::value], kc, kc, std::integral_constant<int, tb - 1>{}); });
            } else {
                double fa[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[ks] = sinv[ks * 64 + lane];
                static_for<tb + 1, 8>([&](auto bic) __attribute__((always_inline)) {
                    constexpr int bi = decltype(bic)::value;
                    if constexpr (rl_owner(bi, tb) == W) {
                        constexpr int sl = rl_slot(bi, tb);
                        double4_t y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
                        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], acc[sl][0], y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], acc[sl][1], y1, 0, 0, 0);
                        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[2], acc[sl][2], y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[3], acc[sl][3], y1, 0, 0, 0);
                        y0 += y1;
                        double* lb = s + rl_blk(bi, tb) * 256;
                        double* gp = pack + rl_blk(bi, tb) * 256;
                        double* ga = A + (long)(16 * bi + l15) + (long)(16 * tb + l4) * lda;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            lb[q * 64 + lane] = y0[q];
                            gp[q * 64 + lane] = y0[q];
                            ga[(long)(4 * q) * lda] = y0[q];
                        }
                    }
                });
            }
        });
        __syncthreads();                                                 // B2(tb): panel column tb is final in LDS
        TICK(3 + 2 * tb);
        // ---- c: waves 1..3 update their trailing tiles with column tb; wave 0 the next diagonal tile, then its pivot block
        role_dispatch(wave, [&](auto Wc) __attribute__((always_inline)) {
            constexpr int W = decltype(Wc)::value;
            if constexpr (W == 0) {
                update(acc[tb + 1], std::integral_constant<int, tb + 1>{}, std::integral_constant<int, tb + 1>{}, tbc);
                pivot(std::integral_constant<int, tb + 1>{});
            } else {
                static_for<tb + 1, 8>([&](auto bjc) __attribute__((always_inline)) {
                    static_for<tb + 2, 8>([&](auto bic) __attribute__((always_inline)) {
                        constexpr int bj = decltype(bjc)::value, bi = decltype(bic)::value;
                        if constexpr (bi > bj && rl_owner(bi, bj) == W) {
                            constexpr int sl = rl_slot(bi, bj);
                            update(acc[sl], bic, bjc, tbc);
                        }
                    });
                });
            }
        });
        __syncthreads();                                                 // B1(tb + 1)
        TICK(4 + 2 * tb);
    });
    if (t == 0 && binfo != 0) atomicCAS(info, 0, info_base + binfo);
    TICK(19);
    if (tick && t == 0) tick[20] = tick[19];
#undef TICK
    __syncthreads();
    pgp_yield_mark(yield_flags, -1);
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

// ---- the same solve without LDS and on < 64 VGPRs (round 6) --------------------------------------------------------------------
// Beside a trailing update every CU holds two bulk workgroups (2 x 74 KB of LDS, 2 x 208 VGPRs per SIMD): what is left is 12 KB and
// 96 registers per SIMD lane.  trsm_rows_kernel (72 KB) therefore waits for a bulk workgroup to leave -- ~100 us when the bulk launch
// has just filled the chip (kernel trace: 98 us in flight against 10 alone, once per panel).  This form fits the gap: LEFT-looking
// over the eight column blocks, one 16 x 16 accumulator in flight,
//     X_c <- (X_c - sum_{tb < c} Y_tb L(c, tb)') inv(L_cc)',
// every operand fragment straight from the L2 (the packed image is 72 KB and hot; the finished Y_tb are re-read from X itself through
// agent-scope loads -- the newest one is kept in registers).  Same products in the same order as the LDS form: bit-identical.
__global__ __launch_bounds__(256, 5) void trsm_rows_lean_kernel(double* __restrict__ X, long ldx, long nrows,
                                                             const double* __restrict__ img /* packed image */,
                                                             unsigned* yield_flags) {
    pgp_yield_mark(yield_flags, +1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 16;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (r0 < nrows) {
        // addresses = wave-uniform 64-bit base (SGPRs: the kernel arguments and the loop counters) + one 32-bit byte offset per lane
        const char* ib = (const char*)img;                                  // fragment (block b, k-step ks) at ib + (b * 256 + ks * 64) * 8 + ioff
        const unsigned ioff = (unsigned)(l4 * 16 + l15) * 8u;
        char* xb = (char*)X;                                                // element (column j) at xb + j * ldx * 8 + xoff
        const unsigned xoff = (unsigned)((r0 + l15 + (long)l4 * ldx) * 8);
        const size_t cstep = (size_t)ldx * 8;
        double4_t ylast = {0.0, 0.0, 0.0, 0.0};
        // rolled loops; the operands of step tb + 1 are in flight while step tb multiplies
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            double4_t acc, iv;
            char* xc = xb + (size_t)(16 * c) * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = *(const double*)(xc + (size_t)(4 * q) * cstep + xoff);
            const char* ic = ib + (size_t)((28 + c) * 256) * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) iv[ks] = *(const double*)(ic + ks * 512 + ioff);
            if (c >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores of Y_0 .. Y_{c-2} have left this wave
            double4_t ya = {0.0, 0.0, 0.0, 0.0}, la = ya;
            const char* lc = ib + (size_t)(c * (c - 1) / 2 * 256) * 8;     // tri_blk(c, 0)
            auto fetch = [&](int tb, double4_t& yv, double4_t& lv) {
                const char* lp = lc + (size_t)tb * 2048;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) lv[ks] = *(const double*)(lp + ks * 512 + ioff);
                const char* yp = xb + (size_t)(16 * tb) * cstep;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    yv[q] = __hip_atomic_load((const double*)(yp + (size_t)(4 * q) * cstep + xoff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            if (c > 0) fetch(0, ya, la);
#pragma unroll 1
            for (int tb = 0; tb < c; ++tb) {
                double4_t yn = ya, ln = la;
                if (tb + 1 < c) fetch(tb + 1, yn, ln);
                const double4_t y = tb == c - 1 ? ylast : ya;              // the newest Y never went to memory and back
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[ks], y[ks], acc, 0, 0, 0);
                ya = yn; la = ln;
            }
            double4_t y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
            y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[0], acc[0], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[1], acc[1], y1, 0, 0, 0);
            y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[2], acc[2], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[3], acc[3], y1, 0, 0, 0);
            ylast = y0 + y1;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(double*)(xc + (size_t)(4 * q) * cstep + xoff) = ylast[q];
        }
    }
    __syncthreads();
    pgp_yield_mark(yield_flags, -1);
}

__global__ __launch_bounds__(256, 2) void trsm_rows_kernel(double* __restrict__ X, long ldx, long nrows,
                                                           const double* __restrict__ Ld, long ldl,
                                                           const double* __restrict__ inv16 /* packed image */,
                                                           unsigned* yield_flags) {
    extern __shared__ __attribute__((aligned(16))) double sl[];
    pgp_yield_mark(yield_flags, +1);
    trsm_rows_body(X, ldx, nrows, inv16, blockIdx.x, sl);
    __syncthreads();
    pgp_yield_mark(yield_flags, -1);
}

// dst (pinned host memory, mapped into the device's address space) <- src: 16 bytes per thread; visible to the host when the kernel
// has completed (coherent host memory, the end of a kernel is a system-scope release)
__global__ __launch_bounds__(256) void publish_kernel(const double* __restrict__ src, double* __restrict__ dst, long count) {
    const long i = 2 * ((long)blockIdx.x * 256 + threadIdx.x);
    if (i + 1 < count) *(double2_t*)(dst + i) = *(const double2_t*)(src + i);
    else if (i < count) dst[i] = src[i];
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

// pivot: 1 = the pivot blocks on the matrix cores (pivot_block_mfma), 0 = lane per row (context option `leaf_pivot`)
int leaf_potrf_launch(double* A, long lda, double* inv16, int* info, int info_base, hipStream_t st,
                      long long* tick, unsigned* yield_flags, int pivot) {
    const size_t shm = (36 * 256 + 256 + 2) * sizeof(double);
    func_max_dynamic_lds(pivot == 2 ? (const void*)leaf_potrf_reg_kernel : pivot ? (const void*)leaf_potrf_kernel<true> : (const void*)leaf_potrf_kernel<false>, shm);
    if (pivot == 2) {
        const size_t shm2 = (28 * 256 + 256) * sizeof(double);
        hipLaunchKernelGGL(leaf_potrf_reg_kernel, dim3(1), dim3(256), shm2, st, A, lda, inv16, info, info_base, tick, yield_flags);
    } else if (pivot)
        hipLaunchKernelGGL(leaf_potrf_kernel<true>, dim3(1), dim3(256), shm, st, A, lda, inv16, info, info_base, tick, yield_flags);
    else
        hipLaunchKernelGGL(leaf_potrf_kernel<false>, dim3(1), dim3(256), shm, st, A, lda, inv16, info, info_base, tick, yield_flags);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int trsm_rows_launch(double* X, long ldx, long nrows, const double* Ld, long ldl, const double* inv16,
                     hipStream_t st, unsigned* yield_flags, bool lean) {
    if (nrows <= 0) return PGP_OK;
    const unsigned nblk = (unsigned)((nrows + 63) / 64);
    if (lean) {
        hipLaunchKernelGGL(trsm_rows_lean_kernel, dim3(nblk), dim3(256), 0, st, X, ldx, nrows, inv16, yield_flags);
        return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
    }
    const size_t shm = 36 * 256 * sizeof(double);
    func_max_dynamic_lds((const void*)trsm_rows_kernel, shm);
    hipLaunchKernelGGL(trsm_rows_kernel, dim3(nblk), dim3(256), shm, st, X, ldx, nrows, Ld, ldl, inv16, yield_flags);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int publish_launch(const double* src_dev, double* dst_host_mapped, long count, hipStream_t st) {
    hipLaunchKernelGGL(publish_kernel, dim3((unsigned)((count + 511) / 512)), dim3(256), 0, st, src_dev, dst_host_mapped, count);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int leaf_inv_launch(const double* L, long ldl, double* W, long ldw, long wstride, int nblocks, hipStream_t st) {
    const size_t shm = (size_t)NB * NB * sizeof(double);
    func_max_dynamic_lds((const void*)leaf_inv_kernel, shm);
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
__global__ __launch_bounds__(256) void diag_out_kernel(const double* __restrict__ D, long ldd, int w,
                                                       double* __restrict__ Fd, long ldf, double* __restrict__ Ed,
                                                       long lde) {
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * w; i += 256) {
        const double v = D[i + (long)j * ldd];
        if (i < w) { if (i >= j) Fd[i + (long)j * ldf] = v; }
        else if (Ed) Ed[(i - w) + (long)j * lde] = v;
    }
}

}  // namespace

int diag_in_launch(const double* src, long lds, double* D, long ldd, int w, hipStream_t st) {
    hipLaunchKernelGGL(diag_in_kernel, dim3(w), dim3(256), 0, st, src, lds, D, ldd, w);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
int diag_out_launch(const double* D, long ldd, int w, double* Fd, long ldf, double* Ed, long lde, hipStream_t st) {
    hipLaunchKernelGGL(diag_out_kernel, dim3(w), dim3(256), 0, st, D, ldd, w, Fd, ldf, Ed, lde);
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
