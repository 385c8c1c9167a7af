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
#include <algorithm>
#include <atomic>

#include "gemm_tile.h"

void func_max_dynamic_lds(const void* fn, size_t bytes) {
    constexpr int MAXF = 64, MAXD = 16;
    static std::atomic<const void*> fns[MAXF];
    static std::atomic<size_t> done[MAXF][MAXD];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXD) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); return; }
    int slot = -1;
    for (int i = 0; i < MAXF; ++i) {
        const void* cur = fns[i].load(std::memory_order_acquire);
        if (cur == fn) { slot = i; break; }
        if (cur == nullptr) {
            const void* expect = nullptr;
            if (fns[i].compare_exchange_strong(expect, fn, std::memory_order_acq_rel) || expect == fn) { slot = i; break; }
        }
    }
    if (slot < 0) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); return; }   // table full: every time
    if (done[slot][dev].load(std::memory_order_acquire) >= bytes) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    size_t prev = done[slot][dev].load(std::memory_order_relaxed);
    while (prev < bytes && !done[slot][dev].compare_exchange_weak(prev, bytes, std::memory_order_release)) {}
}


namespace {

using gemm_tile_ns::BK;

// tile (ti, tj) of workgroup number b of a launch described by g; false: padding entry, nothing to do
__device__ __forceinline__ bool decode_tile(const GemmArgs& g, int b, int T, int TNc, int& ti, int& tj) {
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
    if (g.rev_cols) tj = g.N / TNc - 1 - tj;
    if (g.rev_rows) ti = mt - 1 - ti;
    return true;
}

template <int TM, int TN, bool AKC, bool BKC, bool DMA = false, bool YIELD = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (g.trace) {                                   // every lane the same words: no divergent branch in here
        long long* tr0 = g.trace + 8L * (blockIdx.x + (long)gridDim.x * blockIdx.z);
        tr0[0] = (long long)wall_clock64();
        tr0[6] = (long long)__builtin_readcyclecounter();     // shader clock: with [7] the frequency this workgroup ran at
    }
    int ti, tj;
    if (!decode_tile(g, (int)blockIdx.x, TM, TN, ti, tj)) return;
    long bz = blockIdx.z;
    if (g.order_z) { bz = ti >> 16; ti &= 0xffff; }
    if (g.wait_flag) {
        if (threadIdx.x == 0) {
            for (unsigned it = 0; (int)(__hip_atomic_load(g.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - g.wait_target) < 0; ++it) {
                if (it > (1u << 22) || ((it & 1023u) == 1023u && __hip_atomic_load(g.wait_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(g.wait_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
    }
    if (!YIELD && g.yield_role == 2) {               // one of the chain's own small products: its CU's bulk workgroups give way
        pgp_yield_mark(g.yield_flags, +1);
        gemm_tile_ns::gemm_tile<TM, TN, AKC, BKC, DMA, false>(g, ti, tj, bz, smem, (int)(blockIdx.x + gridDim.x * blockIdx.z));
        __syncthreads();
        pgp_yield_mark(g.yield_flags, -1);
        return;
    }
    gemm_tile_ns::gemm_tile<TM, TN, AKC, BKC, DMA, YIELD>(g, ti, tj, bz, smem, (int)(blockIdx.x + gridDim.x * blockIdx.z));
}

// TWO independent bulk products in ONE launch (round 4: the trailing update TU_b(p) and panel p's share of E E'): workgroups
// [0, na) run a's tiles, the rest b's.  Each launch of a few hundred tiles ends in a tail during which the chip drains (the next
// launch of an in-order stream starts when the last workgroup has gone); merging the pair halves the number of tails per panel.
// Both argument sets live in the kernel-argument segment; the workgroup picks one by a uniform pointer select.
template <bool YIELD>
__global__ __launch_bounds__(256, 2) void gemm_f64_pair_kernel(GemmArgs a, GemmArgs b, int na) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // two copies of the tile code, each reading ITS argument set straight from the kernel-argument segment: a pointer select
    // between the two makes the compiler treat every field as divergent (the LDS-DMA bases must live in SGPRs)
    auto run = [&](const GemmArgs& g, int bid) {
        if (g.trace) {                               // every lane the same words: no divergent branch in here
            long long* tr0 = g.trace + 8L * bid;
            tr0[0] = (long long)wall_clock64();
            tr0[6] = (long long)__builtin_readcyclecounter();
        }
        int ti, tj;
        if (!decode_tile(g, bid, 128, 128, ti, tj)) return;
        gemm_tile_ns::gemm_tile<128, 128, false, false, true, YIELD>(g, ti, tj, 0, smem, bid);
    };
    if ((int)blockIdx.x < na) run(a, (int)blockIdx.x);
    else run(b, (int)blockIdx.x - na);
}

// TWO tile rows per workgroup (GemmArgs::fold_rows): rows mt - 1 - r and r of the same tile column, one after the other.  For a
// product whose k-range grows with the tile row (KM_LT_I: a lower-triangular A, GP.predict's V = L^-1 Ks) the pair's work is the
// same for every workgroup.  It matters on this chip because the workgroup distributor hands out slots in a fixed rotation and
// BLOCKS on the engine whose turn it is (EXPERIMENTS.md, round 3): a grid of tiles that last 1x .. 64x leaves the slots of the
// short ones idle until the rotation comes round -- M = N = K = 8192 clipped to the triangle ran at 47 TF of algorithmic flops
// against 75 TF unclipped (tools/_gemm_clip.py).
template <bool YIELD>
__global__ __launch_bounds__(256, 2) void gemm_f64_fold_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int mt = g.M / 128, half = (mt + 1) / 2;
    const int r = (int)blockIdx.x % half, tj = (int)blockIdx.x / half;
    const int hi = mt - 1 - r;
    const long bz = blockIdx.z;
    gemm_tile_ns::gemm_tile<128, 128, false, false, true, YIELD>(g, hi, tj, bz, smem, 0);
    if (hi != r) {
        __syncthreads();                             // (the first tile's last LDS reads are over before the second tile stages)
        gemm_tile_ns::gemm_tile<128, 128, false, false, true, YIELD>(g, r, tj, bz, smem, 0);
    }
}

template <int T, bool AKC, bool BKC, bool DMA = false, bool YIELD = false>
int launch_t(const GemmArgs& g, hipStream_t st) {
    constexpr int SK = BK + 2;
    constexpr int ASZ = AKC ? T * SK : BK * (T + 16);
    constexpr int BSZ = BKC ? T * SK : BK * (T + 16);
    const size_t shm = 2 * (ASZ + BSZ) * sizeof(double);
    const int mt = g.M / T, nt = g.N / T;
    unsigned nblk = (g.tri == 2) ? (unsigned)((long)mt * (mt + 1) / 2) : (unsigned)(mt * nt);
    if (g.order) nblk = (unsigned)g.norder;
    dim3 grid(nblk, 1, (g.batch > 0 && !g.order_z) ? g.batch : 1);
    func_max_dynamic_lds((const void*)gemm_f64_kernel<T, T, AKC, BKC, DMA, YIELD>, shm);
    hipLaunchKernelGGL((gemm_f64_kernel<T, T, AKC, BKC, DMA, YIELD>), grid, dim3(256), shm, st, g);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// the 128 x 64 LDS-DMA tile (GemmArgs::tile == 1264): plain rectangles only -- no triangular tile set, no host-built order, no batch
static bool dma_ok(const GemmArgs& g);
static bool dma1264_ok(const GemmArgs& g) {
    return g.tile == 1264 && dma_ok(g) && !g.tri && !g.order && g.batch <= 1 && !g.order_z && (g.N % 64) == 0 && (g.M % 128) == 0 && g.kmode == KM_FULL;
}
template <bool YIELD>
int launch_1264(const GemmArgs& g, hipStream_t st) {
    constexpr int SA_ = 128 + 16;
    const size_t shm = 2 * (BK * SA_ + (BK / 2) * SA_) * sizeof(double);
    dim3 grid((unsigned)((g.M / 128) * (g.N / 64)), 1, 1);
    func_max_dynamic_lds((const void*)gemm_f64_kernel<128, 64, false, false, true, YIELD>, shm);
    hipLaunchKernelGGL((gemm_f64_kernel<128, 64, false, false, true, YIELD>), grid, dim3(256), shm, st, g);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

static bool dma_ok(const GemmArgs& g) {
    return g.tile != 64 && (g.dbg & 64) && !g.a_kc && !g.b_kc && (g.K % 16) == 0 && (g.koff % 16) == 0 && (g.batch_dk % 16) == 0;
}

template <int T>
int launch_l(const GemmArgs& g, hipStream_t st) {
    if (!g.a_kc && !g.b_kc) return launch_t<T, false, false>(g, st);
    if (!g.a_kc && g.b_kc) return launch_t<T, false, true>(g, st);
    if (g.a_kc && !g.b_kc) return launch_t<T, true, false>(g, st);
    return launch_t<T, true, true>(g, st);
}

}  // namespace

bool gemm_f64_uses_dma128(const GemmArgs& g) { return dma_ok(g) && !dma1264_ok(g); }
bool gemm_f64_uses_dma(const GemmArgs& g) { return dma_ok(g); }           // any LDS-DMA tile (128 x 128 or 128 x 64): the yield poll's side

static unsigned launch_blocks(const GemmArgs& g) {
    const int mt = g.M / 128, nt = g.N / 128;
    if (g.order) return (unsigned)g.norder;
    return (g.tri == 2) ? (unsigned)((long)mt * (mt + 1) / 2) : (unsigned)(mt * nt);
}
// can the two go out as one launch?  Both on the LDS-DMA 128-tile path, unbatched, no device-side waits, the same yield role
bool gemm_f64_pair_ok(const GemmArgs& a, const GemmArgs& b) {
    auto plain = [](const GemmArgs& g) {
        return g.M > 0 && g.N > 0 && g.tile != 64 && dma_ok(g) && g.batch <= 1 && !g.order_z && !g.wait_flag && g.yield_role != 2;
    };
    return plain(a) && plain(b) && (a.yield_role == 1 && a.yield_flags) == (b.yield_role == 1 && b.yield_flags);
}
int gemm_f64_launch_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    if (!gemm_f64_pair_ok(a, b)) return -2;
    constexpr int SA_ = 128 + 16;
    const size_t shm = 2 * (2 * BK * SA_) * sizeof(double);
    const unsigned na = launch_blocks(a), nb = launch_blocks(b);
    const bool yield = a.yield_role == 1 && a.yield_flags;
    func_max_dynamic_lds((const void*)(yield ? gemm_f64_pair_kernel<true> : gemm_f64_pair_kernel<false>), shm);
    if (yield) hipLaunchKernelGGL(gemm_f64_pair_kernel<true>, dim3(na + nb), dim3(256), shm, st, a, b, (int)na);
    else hipLaunchKernelGGL(gemm_f64_pair_kernel<false>, dim3(na + nb), dim3(256), shm, st, a, b, (int)na);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

static int launch_fold(const GemmArgs& g, hipStream_t st) {
    constexpr int SA_ = 128 + 16;
    const size_t shm = 2 * (2 * BK * SA_) * sizeof(double);
    const int mt = g.M / 128, nt = g.N / 128;
    dim3 grid((unsigned)(((mt + 1) / 2) * nt), 1, g.batch > 0 ? g.batch : 1);
    const bool yield = g.yield_role == 1 && g.yield_flags;
    func_max_dynamic_lds((const void*)(yield ? gemm_f64_fold_kernel<true> : gemm_f64_fold_kernel<false>), shm);
    if (yield) hipLaunchKernelGGL(gemm_f64_fold_kernel<true>, grid, dim3(256), shm, st, g);
    else hipLaunchKernelGGL(gemm_f64_fold_kernel<false>, grid, dim3(256), shm, st, g);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int gemm_f64_launch(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return PGP_OK;
    if (g.tile == 64) return launch_l<64>(g, st);
    if (g.fold_rows && g.tile == 128 && dma_ok(g) && !g.tri && !g.order && !g.order_z && !g.wait_flag && g.yield_role != 2 && !g.trace)
        return launch_fold(g, st);
    if (dma1264_ok(g)) return (g.yield_role == 1 && g.yield_flags) ? launch_1264<true>(g, st) : launch_1264<false>(g, st);
    // LDS-DMA staging (dbg bit 64): 128 x 128 tiles of M-contiguous operands, whole 16-deep k-tiles
    if (dma_ok(g)) return (g.yield_role == 1 && g.yield_flags) ? launch_t<128, false, false, true, true>(g, st)
                                                               : launch_t<128, false, false, true>(g, st);
    return launch_l<128>(g, st);
}
