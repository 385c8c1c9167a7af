// fp64 MFMA GEMM / SYRK / TRMM-shaped kernel family for gfx950 (MI355X, CDNA4).
//
// One kernel template serves every dense contraction of the exact-GP pipeline:
//   * Cholesky trailing updates        C -= P P^T          (lower tiles only, NT)
//   * triangular-inverse combine steps T = L21 W11, W21 = -W22 T   (k-range clipped to the triangle)
//   * B^-1 = W^T W                     (lower tiles only, TN, k >= i0)
// Design (see DESIGN.md "gemm_f64"):
//   - v_mfma_f64_16x16x4_f64: 2048 flop / instruction, one f64 A and one f64 B operand per lane.
//   - workgroup = 256 threads = 4 waves (2x2); tile TMxTN = 128x128 (wave 64x64, 16 accumulators
//     = 128 VGPRs) or 64x64 (wave 32x32) for latency-bound panel work; BK = 16.
//   - operands staged global -> registers -> LDS, double-buffered, one barrier per BK step.
//   - LDS layouts chosen so that every ds_read_b64 fragment read is bank-conflict free:
//       M-contiguous source: [k][m] with row stride TM+16 doubles  (2*stride = 32 mod 64 banks)
//       K-contiguous source: [m][k] with row stride BK+2 = 18 doubles (36m+2k distinct mod 64)
//   - the MFMA "row" index is mapped to our N (column) index and the MFMA "col" index to our M
//     (contiguous) index, so each accumulator register is a 16-element contiguous run of C
//     in memory (128-byte segments) for the C load / store.
//   - C is pre-loaded into the accumulators (acc = (beta/alpha)*C) before the k-loop so its HBM
//     latency overlaps the first operand tiles; out = alpha*acc = beta*C + alpha*A*B'.
#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16;

template <int TM, int TN, bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int SA = TM + 16, SB = TN + 16, SK = BK + 2;
    constexpr int ASZ = AKC ? TM * SK : BK * SA;
    constexpr int BSZ = BKC ? TN * SK : BK * SB;
    constexpr int STAGE = ASZ + BSZ;
    constexpr int FM = TM / 32, FN = TN / 32;      // 16x16 fragments per wave in M and N
    constexpr int AV = TM * BK / 2 / 256;          // double2 vectors staged per thread
    constexpr int BV = TN * BK / 2 / 256;

    const int mt = g.M / TM;
    int ti, tj;
    if (g.order) {                                  // host-built tile order (see tile_order() in capi.hip)
        ti = g.order[2 * blockIdx.x];
        tj = g.order[2 * blockIdx.x + 1];
        if (ti < 0) return;                         // padding entry
    } else if (g.tri == 2) {                               // packed lower-triangular tile index
        const int b = blockIdx.x;
        int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((long)(r + 1) * (r + 2) / 2 <= b) ++r;
        while ((long)r * (r + 1) / 2 > b) --r;
        ti = r;
        tj = b - r * (r + 1) / 2;
    } else {
        ti = blockIdx.x % mt;
        tj = blockIdx.x / mt;
    }
    const int i0 = ti * TM, j0 = tj * TN;
    bool diag = false;
    if (g.tri) {
        if (i0 + g.tri_off < j0) return;
        diag = g.mask_diag && (i0 + g.tri_off == j0);
    }
    int k0 = 0, k1 = g.K;
    if (g.kmode == KM_GE_I) k0 = i0 + g.koff;
    else if (g.kmode == KM_GE_J) k0 = j0 + g.koff;
    else if (g.kmode == KM_LT_I) k1 = i0 + TM + g.koff;
    if (k0 < 0) k0 = 0;
    if (k1 > g.K) k1 = g.K;
    k0 &= ~(BK - 1);

    const long bz = blockIdx.z;
    const double* __restrict__ A = g.A + bz * g.sA;
    const double* __restrict__ B = g.B + bz * g.sB;
    double* __restrict__ C = g.C + bz * g.sC;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (wave & 1) * (TM / 2), wn = (wave >> 1) * (TN / 2);
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- accumulators, pre-loaded with (beta/alpha)*C (== alpha*beta*C for alpha = +-1) ------
    double4_t acc[FM][FN];
    const double ab = g.beta / g.alpha;
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int in = 0; in < FN; ++in) {
            if (g.beta != 0.0 && !(g.dbg & 4)) {
                const double* cp = C + (long)(i0 + wm + im * 16 + l15) + (long)(j0 + wn + in * 16 + l4) * g.ldc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[im][in][r] = ab * cp[(long)(4 * r) * g.ldc];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[im][in][r] = 0.0;
            }
        }

    // ---- staging helpers ----------------------------------------------------------------
    double2_t ra[AV], rb[BV];
    // per-thread source pointers of the staging loads, bumped by one k-tile per iteration (no 64-bit address
    // arithmetic inside the k-loop: those VALU ops would sit un-overlapped at the loop head)
    const double* pa[AV];
    const double* pb[BV];
    long astep, bstep;
    if (!AKC) {
        constexpr int VPR = TM / 2, RPP = 256 / VPR;
#pragma unroll
        for (int p = 0; p < AV; ++p) pa[p] = A + (long)(i0 + 2 * (t % VPR)) + (long)(k0 + t / VPR + p * RPP) * g.lda;
        astep = (long)BK * g.lda;
    } else {
#pragma unroll
        for (int p = 0; p < AV; ++p) pa[p] = A + (long)(k0 + 2 * (t & 7)) + (long)(i0 + (t >> 3) + p * 32) * g.lda;
        astep = BK;
    }
    if (!BKC) {
        constexpr int VPR = TN / 2, RPP = 256 / VPR;
#pragma unroll
        for (int p = 0; p < BV; ++p) pb[p] = B + (long)(j0 + 2 * (t % VPR)) + (long)(k0 + t / VPR + p * RPP) * g.ldb;
        bstep = (long)BK * g.ldb;
    } else {
#pragma unroll
        for (int p = 0; p < BV; ++p) pb[p] = B + (long)(k0 + 2 * (t & 7)) + (long)(j0 + (t >> 3) + p * 32) * g.ldb;
        bstep = BK;
    }
    auto gload = [&](int) {
#pragma unroll
        for (int p = 0; p < AV; ++p) { ra[p] = *(const double2_t*)pa[p]; pa[p] += astep; }
#pragma unroll
        for (int p = 0; p < BV; ++p) { rb[p] = *(const double2_t*)pb[p]; pb[p] += bstep; }
    };
    auto sstore = [&](int buf) {
        double* sa = smem + buf * STAGE;
        double* sb = sa + ASZ;
        if (!AKC) {
            constexpr int VPR = TM / 2;
            constexpr int RPP = 256 / VPR;
#pragma unroll
            for (int p = 0; p < AV; ++p) {
                const int mv = t % VPR, kr = t / VPR + p * RPP;
                *(double2_t*)(sa + kr * SA + 2 * mv) = ra[p];
            }
        } else {
#pragma unroll
            for (int p = 0; p < AV; ++p) {
                const int kp = t & 7, m = (t >> 3) + p * 32;
                *(double2_t*)(sa + m * SK + 2 * kp) = ra[p];
            }
        }
        if (!BKC) {
            constexpr int VPR = TN / 2;
            constexpr int RPP = 256 / VPR;
#pragma unroll
            for (int p = 0; p < BV; ++p) {
                const int nv = t % VPR, kr = t / VPR + p * RPP;
                *(double2_t*)(sb + kr * SB + 2 * nv) = rb[p];
            }
        } else {
#pragma unroll
            for (int p = 0; p < BV; ++p) {
                const int kp = t & 7, n = (t >> 3) + p * 32;
                *(double2_t*)(sb + n * SK + 2 * kp) = rb[p];
            }
        }
    };

    if (k0 < k1) {
        gload(k0);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int kt = k0; kt < k1; kt += BK) {
            const bool more = kt + BK < k1;
            if (more && !(g.dbg & 1)) gload(kt + BK);
            const double* sa = smem + buf * STAGE;
            const double* sb = sa + ASZ;
            // fragments are double-buffered in registers: the LDS reads of k-step ks+1 are issued BEFORE the 16
            // MFMAs of k-step ks, so their latency hides behind 1024 cycles of matrix work instead of stalling
            // the (in-order) wave once per k-step
            double fa[2][FM], fb[2][FN];
            auto ldfrag = [&](int ks, int slot) {
                const int k = ks * 4 + l4;
#pragma unroll
                for (int im = 0; im < FM; ++im) {
                    const int m = wm + im * 16 + l15;
                    fa[slot][im] = AKC ? sa[m * SK + k] : sa[k * SA + m];
                }
#pragma unroll
                for (int in = 0; in < FN; ++in) {
                    const int n = wn + in * 16 + l15;
                    fb[slot][in] = BKC ? sb[n * SK + k] : sb[k * SB + n];
                }
            };
            ldfrag(0, 0);
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                if (ks + 1 < BK / 4) ldfrag(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int in = 0; in < FN; ++in)
#pragma unroll
                    for (int im = 0; im < FM; ++im)
                        acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ks & 1][in], fa[ks & 1][im], acc[im][in], 0, 0, 0);
            }
            if (more && !(g.dbg & 1)) sstore(buf ^ 1);
            if (!(g.dbg & 2)) __syncthreads();
            if (!(g.dbg & 1)) buf ^= 1;
        }
    }

    // ---- epilogue: C = alpha * acc ---------------------------------------------------------
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int in = 0; in < FN; ++in) {
            const int m = wm + im * 16 + l15;
            const int nb = wn + in * 16 + l4;
            double* cp = C + (long)(i0 + m) + (long)(j0 + nb) * g.ldc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 4 * r;
                if ((!diag || m >= n) && (!(g.dbg & 4) || acc[im][in][r] == 12345.678)) cp[(long)(4 * r) * g.ldc] = g.alpha * acc[im][in][r];
            }
        }
}

template <int T, bool AKC, bool BKC>
int launch_t(const GemmArgs& g, hipStream_t st) {
    constexpr int SK = BK + 2;
    constexpr int ASZ = AKC ? T * SK : BK * (T + 16);
    constexpr int BSZ = BKC ? T * SK : BK * (T + 16);
    const size_t shm = 2 * (ASZ + BSZ) * sizeof(double) + ((g.dbg & 8) ? 20000 : 0);   // dbg 8: force 1 workgroup / CU
    const int mt = g.M / T, nt = g.N / T;
    unsigned nblk = (g.tri == 2) ? (unsigned)((long)mt * (mt + 1) / 2) : (unsigned)(mt * nt);
    if (g.order) nblk = (unsigned)g.norder;
    dim3 grid(nblk, 1, g.batch > 0 ? g.batch : 1);
    static size_t attr_set = 0;
    if (attr_set < shm) {
        (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<T, T, AKC, BKC>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set = shm;
    }
    hipLaunchKernelGGL((gemm_f64_kernel<T, T, AKC, BKC>), grid, dim3(256), shm, st, g);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

template <int T>
int launch_l(const GemmArgs& g, hipStream_t st) {
    if (!g.a_kc && !g.b_kc) return launch_t<T, false, false>(g, st);
    if (!g.a_kc && g.b_kc) return launch_t<T, false, true>(g, st);
    if (g.a_kc && !g.b_kc) return launch_t<T, true, false>(g, st);
    return launch_t<T, true, true>(g, st);
}

}  // namespace

int gemm_f64_launch(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return PGP_OK;
    if (g.tile == 64) return launch_l<64>(g, st);
    return launch_l<128>(g, st);
}
