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
#include <atomic>

#include "gemm_tile.h"

namespace {

using gemm_tile_ns::BK;

__device__ int g_cu_arrivals[4096];       // de-phasing experiment (gemm_dbg & 32): arrivals per CU

// tile (ti, tj) of workgroup number b of a launch described by g; false: padding entry, nothing to do
__device__ __forceinline__ bool decode_tile(const GemmArgs& g, int b, int T, int& ti, int& tj) {
    if (g.order) {                                  // host-built tile order (see tile_order() in capi.hip)
        ti = g.order[2 * b];
        tj = g.order[2 * b + 1];
        return ti >= 0;
    }
    if (g.tri == 2) {                               // packed lower-triangular tile index
        int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((long)(r + 1) * (r + 2) / 2 <= b) ++r;
        while ((long)r * (r + 1) / 2 > b) --r;
        ti = r;
        tj = b - r * (r + 1) / 2;
        return true;
    }
    const int mt = g.M / T;
    ti = b % mt;
    tj = b / mt;
    if (g.rev_cols) tj = g.N / T - 1 - tj;
    return true;
}

// Two independent products in ONE grid: workgroups [0, na) work on `a`, the rest on `b`.  Used by the Cholesky sweep to
// append panel p's share of B^-1 = E E^T to the trailing update TU_b(p): consecutive launches on one stream drain the
// chip at every boundary (the partial last wave of one kernel runs alone), a merged grid has one tail instead of two.
struct GemmArgsPair { GemmArgs g[2]; int na; };
template <int TM, int TN, bool AKC, bool BKC, bool DMA>
__global__ __launch_bounds__(256, 2) void gemm_f64_dual_kernel(GemmArgsPair p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int na = p.na;
    const bool second = (int)blockIdx.x >= na;
    const GemmArgs& g = p.g[second ? 1 : 0];        // uniform index into the kernarg segment: scalar loads, no private copy
    int ti, tj;
    if (!decode_tile(g, second ? (int)blockIdx.x - na : (int)blockIdx.x, TM, ti, tj)) return;
    gemm_tile_ns::gemm_tile<TM, TN, AKC, BKC, DMA>(g, ti, tj, 0, smem);
}

template <int TM, int TN, bool AKC, bool BKC, bool DMA = false, bool EXP = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int ti, tj;
    if (!decode_tile(g, (int)blockIdx.x, TM, ti, tj)) return;
    if ((g.dbg & 32) && blockIdx.x < 512) {          // experiment: de-phase the two workgroups that share a CU (first wave only)
        __shared__ int s_par;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, 32 bits
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u; // HW_REG_XCC_ID
            const unsigned key = (xcc << 8) | ((hw >> 8) & 0xffu);                                  // cu_id, sh_id, se_id
            s_par = atomicAdd(&g_cu_arrivals[key & 4095u], 1) & 1;
        }
        __syncthreads();
        if (s_par) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 7000) __builtin_amdgcn_s_sleep(10);                        // ~70 us = half a K=512 tile
        }
    }
    gemm_tile_ns::gemm_tile<TM, TN, AKC, BKC, DMA, EXP>(g, ti, tj, blockIdx.z, smem);
}

template <int T, bool AKC, bool BKC, bool DMA = false, bool EXP = false>
int launch_t(const GemmArgs& g, hipStream_t st) {
    constexpr int SK = BK + 2;
    constexpr int ASZ = AKC ? T * SK : BK * (T + 16);
    constexpr int BSZ = BKC ? T * SK : BK * (T + 16);
    const size_t shm = 2 * (ASZ + BSZ) * sizeof(double) + ((g.dbg & 8) ? 20000 : 0);   // dbg 8: force 1 workgroup / CU
    const int mt = g.M / T, nt = g.N / T;
    unsigned nblk = (g.tri == 2) ? (unsigned)((long)mt * (mt + 1) / 2) : (unsigned)(mt * nt);
    if (g.order) nblk = (unsigned)g.norder;
    dim3 grid(nblk, 1, g.batch > 0 ? g.batch : 1);
    static std::atomic<size_t> attr_set{0};          // two fit streams (host threads) launch concurrently
    if (attr_set.load(std::memory_order_acquire) < shm) {
        (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<T, T, AKC, BKC, DMA, EXP>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set.store(shm, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_f64_kernel<T, T, AKC, BKC, DMA, EXP>), grid, dim3(256), shm, st, g);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

static unsigned grid_blocks(const GemmArgs& g, int T) {
    const int mt = g.M / T, nt = g.N / T;
    if (g.order) return (unsigned)g.norder;
    return (g.tri == 2) ? (unsigned)((long)mt * (mt + 1) / 2) : (unsigned)(mt * nt);
}

static bool dma_ok(const GemmArgs& g) {
    return g.tile != 64 && (g.dbg & 64) && !g.a_kc && !g.b_kc && (g.K % 16) == 0 && (g.koff % 16) == 0;
}

template <int T>
int launch_l(const GemmArgs& g, hipStream_t st) {
    if (!g.a_kc && !g.b_kc) return launch_t<T, false, false>(g, st);
    if (!g.a_kc && g.b_kc) return launch_t<T, false, true>(g, st);
    if (g.a_kc && !g.b_kc) return launch_t<T, true, false>(g, st);
    return launch_t<T, true, true>(g, st);
}

}  // namespace

bool gemm_f64_uses_dma128(const GemmArgs& g) { return dma_ok(g); }

bool gemm_f64_dual_ok(const GemmArgs& a, const GemmArgs& b) {
    return dma_ok(a) && dma_ok(b) && a.batch <= 1 && b.batch <= 1 && a.M > 0 && a.N > 0 && b.M > 0 && b.N > 0 &&
           !b.sig_counter && !b.sig2_counter && !a.stamps && !b.stamps;
}

int gemm_f64_dual_launch(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    if (!gemm_f64_dual_ok(a, b)) return -1;
    constexpr int T = 128;
    const size_t shm = 2 * 2 * BK * (T + 16) * sizeof(double);
    const unsigned na = grid_blocks(a, T), nb = grid_blocks(b, T);
    static std::atomic<size_t> attr_set{0};
    if (attr_set.load(std::memory_order_acquire) < shm) {
        (void)hipFuncSetAttribute((const void*)gemm_f64_dual_kernel<T, T, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_set.store(shm, std::memory_order_release);
    }
    GemmArgsPair pr;
    pr.g[0] = a; pr.g[1] = b; pr.na = (int)na;
    hipLaunchKernelGGL((gemm_f64_dual_kernel<T, T, false, false, true>), dim3(na + nb), dim3(256), shm, st, pr);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int gemm_f64_launch(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return PGP_OK;
    if (g.tile == 64) return launch_l<64>(g, st);
    // LDS-DMA staging (dbg bit 64): 128 x 128 tiles of M-contiguous operands, whole 16-deep k-tiles
    const bool ablate = (g.dbg & (1 | 2 | 1024)) != 0;        // timing experiments: the instantiations that test those bits
    if (dma_ok(g)) return ablate ? launch_t<128, false, false, true, true>(g, st) : launch_t<128, false, false, true>(g, st);
    if (ablate && !g.a_kc && !g.b_kc) return launch_t<128, false, false, false, true>(g, st);
    return launch_l<128>(g, st);
}
