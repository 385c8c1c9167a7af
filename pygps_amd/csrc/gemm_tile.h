// fp64 MFMA GEMM tile body shared by gemm_f64_kernel (gemm_f64.hip) and the resident diagonal-panel server
// (panel.hip): one workgroup (256 threads) computes one TM x TN tile of  C = alpha * A B' + beta * Cin.
#pragma once
#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace gemm_tile_ns {

constexpr int BK = 16;

// One 1 KiB LDS-DMA piece: every lane of the wave names 16 bytes of global memory, the 64 x 16 B land contiguously at
// the (wave-uniform) LDS byte address in M0 -- no staging VGPRs, no ds_write.  Issued through asm so that hipcc does not
// serialise it against the LDS reads of the other buffer (it cannot prove the two stages disjoint); completion is the
// issuing wave's vmcnt, visibility to the other waves the barrier that follows (cdna guide, LDS-DMA).
// Everything wave-uniform is kept OUT of the vector ALU: the source row is a 64-bit SGPR base (saddr form) plus
// one loop-invariant per-lane byte offset, the LDS address is an SGPR base plus a compile-time constant.  A stage of the k-loop
// then issues its eight pieces with scalar instructions only -- VALU instructions of one wave compete with the MFMA issue of
// the other wave on the same SIMD (measured: the k-loop with loop-invariant addresses runs 8 % faster).
template <int LDS_OFF>
__device__ __forceinline__ void lds_dma_1k_s(unsigned voff, const double* sbase, unsigned lds_wave_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_wave_base), "n"(LDS_OFF) : "memory", "scc");   // s_add_u32 writes SCC
}
__device__ __forceinline__ const double* uniform_ptr(const double* p) {
    const unsigned long a = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const double*)(((unsigned long)hi << 32) | lo);
}
template <int V> struct IC { static constexpr int value = V; };

// value of the neighbouring lane (l ^ 1): DPP quad_perm [1,0,3,2] on both halves
__device__ __forceinline__ double swap_adjacent(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// A bulk workgroup gives way: while a workgroup of the diagonal-panel chain is resident on this CU (its word in the yield table
// is non-zero) the wave sleeps instead of issuing MFMAs -- the chain's pivot wave runs a dependent fp64 VALU chain on the same
// pipe and is 3-4x slower beside a saturating bulk wave.  Bounded: gives up after ~350 us.
__device__ __noinline__ void gemm_yield_wait(const unsigned* yf) {
    for (int i = 0; i < 400; ++i) {
        __builtin_amdgcn_s_sleep(32);
        if (__hip_atomic_load(yf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
    }
}

// A stage of the k-loop is one basic block: nothing run-time selectable is tested inside it (YIELD adds one load per stage,
// issued beside the LDS-DMA pieces and consumed after the stage's barrier, and a never-taken branch).
template <int TM, int TN, bool AKC, bool BKC, bool DMA = false, bool YIELD = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int ti, int tj, long bz, double* __restrict__ smem, int wg = 0) {
    static_assert(!DMA || (!AKC && !BKC && TM == 128 && (TN == 128 || TN == 64)), "LDS-DMA staging: M-contiguous operands, 128 x 128 or 128 x 64 tiles");
    // BH (round 6, the 128 x 64 LDS-DMA tile: TWO workgroups per CU for a launch of fewer 128 x 128 tiles than CUs): a k-row of the B
    // operand is 64 doubles = half a 1 KiB LDS-DMA piece, so one piece carries the k-rows (k, k + 4) -- lanes 0-31 fetch row k,
    // lanes 32-63 row k + 4, landing 512 bytes apart -- and the four k-rows a fragment read touches (4 ks + l4) sit in four
    // DIFFERENT pieces, SA doubles apart like the A operand's rows (the same conflict-free bank pattern):
    //   row k -> piece (k & 3) + 4 (k >> 3), half (k >> 2) & 1;  8 pieces per stage, wave w stages pieces w and w + 4
    constexpr bool BH = DMA && TN == 64;
    constexpr int SA = TM + 16, SB = BH ? TM + 16 : TN + 16, SK = BK + 2;
    constexpr int ASZ = AKC ? TM * SK : BK * SA;
    constexpr int BSZ = BKC ? TN * SK : (BH ? (BK / 2) * SB : BK * SB);
    constexpr int STAGE = ASZ + BSZ;
    constexpr int FM = TM / 32, FN = TN / 32;      // 16x16 fragments per wave in M and N
    constexpr int AV = TM * BK / 2 / 256;          // double2 vectors staged per thread
    constexpr int BV = TN * BK / 2 / 256;

    const int i0 = ti * TM, j0 = tj * TN;
    // shrinking batch (see GemmArgs::batch_dm): product bz has fewer rows, its first-touch row moves with them
    if (g.batch_dm && i0 >= g.M - (int)bz * g.batch_dm) return;
    const int zero_from = g.zero_from > 0 ? g.zero_from - (int)bz * g.batch_dm : 0;
    if (i0 < g.skip_hi && j0 < g.skip_hi && i0 >= g.skip_lo && j0 >= g.skip_lo) return;
    bool diag = false;
    if (g.tri) {
        if (i0 + g.tri_off < j0) return;
        diag = g.mask_diag && (i0 + g.tri_off == j0);
    }
    int k0 = 0, k1 = g.K;
    const int koff = g.koff + (int)bz * g.batch_dk;
    if (g.kmode == KM_GE_I) k0 = i0 + koff;
    else if (g.kmode == KM_GE_J) k0 = j0 + koff;
    else if (g.kmode == KM_LT_I) k1 = i0 + TM + koff;
    else if (g.kmode == KM_LT_J) k1 = j0 + TN + koff;
    if (k0 < 0) k0 = 0;
    if (k1 > g.K) k1 = g.K;
    k0 &= ~(BK - 1);

    // two-piece row spaces: pick the piece this tile's rows live in (workgroup-uniform)
    const bool a_hi = g.a_split && i0 >= g.a_split;
    const bool c_hi = g.c_split && i0 >= g.c_split;
    const double* __restrict__ A = (a_hi ? g.A2 - g.a_split : g.A) + bz * g.sA;
    const long lda = a_hi ? g.lda2 : g.lda;
    const double* __restrict__ B = g.B + bz * g.sB;
    double* __restrict__ C = (c_hi ? g.C2 - g.c_split : g.C) + bz * g.sC - (bz * (bz - 1) / 2) * g.batch_sC2;
    const long ldc = (c_hi ? g.ldc2 : g.ldc) - bz * g.batch_dldc;
    const double* __restrict__ Cin = g.Cin ? (c_hi ? g.Cin2 - g.c_split : g.Cin) : C;
    const long ldcin = g.Cin ? (c_hi ? g.ldcin2 : g.ldcin) : ldc;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (wave & 1) * (TM / 2), wn = (wave >> 1) * (TN / 2);
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- accumulators, pre-loaded with (beta/alpha)*C (== alpha*beta*C for alpha = +-1) ------
    double4_t acc[FM][FN];
    const double ab = g.beta / g.alpha;
    const bool wantc = g.beta != 0.0 && !(zero_from > 0 && i0 >= zero_from);
    // LDS-DMA variant: the staging registers it frees hold one quarter of the C tile at a time, fetched DURING the k-loop
    // and folded into the accumulators two k-steps later ("lazy C": the pre-load no longer delays the first MFMA)
    // (the prologue is LZ = 16 k-steps for BOTH LDS-DMA tile shapes, and an element's C value is folded in at the same k-step in
    //  both -- step 4 q + 2 (im / 2) + 1 for the element's 16-column group q of its 64-column slab: the 128 x 64 tile reproduces the
    //  128 x 128 tile bit for bit)
    constexpr int LZ = 16;
    const bool lazyc = DMA && wantc && (g.dbg & 256) && (k1 - k0) >= (LZ + 2) * BK;
    const bool preload = wantc && !lazyc;
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int in = 0; in < FN; ++in) {
            if (preload) {
                const double* cp = Cin + (long)(i0 + wm + im * 16 + l15) + (long)(j0 + wn + in * 16 + l4) * ldcin;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[im][in][r] = ab * cp[(long)(4 * r) * ldcin];
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
        for (int p = 0; p < AV; ++p) pa[p] = A + (long)(i0 + 2 * (t % VPR)) + (long)(k0 + t / VPR + p * RPP) * lda;
        astep = (long)BK * lda;
    } else {
#pragma unroll
        for (int p = 0; p < AV; ++p) pa[p] = A + (long)(k0 + 2 * (t & 7)) + (long)(i0 + (t >> 3) + p * 32) * lda;
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
    // LDS-DMA variant: wave w stages k-rows w, w+4, w+8, w+12 of both operands (one 1 KiB piece = one k-row of 128 doubles)
    const double* da[4];                           // wave-uniform (SGPR) row bases: k-row wave + 4 p of this stage, column i0 / j0
    const double* db[4];
    unsigned lds_w = 0;                            // LDS byte address of k-row `wave` of stage buffer 0 (wave-uniform)
    const unsigned dvoff = (unsigned)lane * 16u;   // this lane's 16 bytes inside the 1 KiB row piece
    unsigned dvoff_b = dvoff;                      // BH: lanes 32-63 fetch the k-row four further down
    if constexpr (BH) dvoff_b = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * (unsigned)(4L * g.ldb * 8L);
    const unsigned smem_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
    if constexpr (DMA) {
        static_assert(SA == SB, "one LDS row stride for both operands");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            da[p] = uniform_ptr(A + (long)i0 + (long)(k0 + wave + 4 * p) * lda);
            // BH: two pieces per wave, k-rows (wave, wave + 4) and (wave + 8, wave + 12)
            db[p] = uniform_ptr(B + (long)j0 + (long)(k0 + wave + (BH ? 8 * (p & 1) : 4 * p)) * g.ldb);
        }
        lds_w = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)wave * (unsigned)(SA * 8));
    }
    const long dstep_a = (long)BK * lda, dstep_b = (long)BK * g.ldb;
    auto dma_stage = [&](auto bufc) {
        constexpr int O = decltype(bufc)::value * STAGE * 8;
        if constexpr (DMA && BH) {
            lds_dma_1k_s<O + 0 * SA * 8>(dvoff, da[0], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 0 * SB * 8>(dvoff_b, db[0], lds_w);
            lds_dma_1k_s<O + 4 * SA * 8>(dvoff, da[1], lds_w);
            lds_dma_1k_s<O + 8 * SA * 8>(dvoff, da[2], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 4 * SB * 8>(dvoff_b, db[1], lds_w);
            lds_dma_1k_s<O + 12 * SA * 8>(dvoff, da[3], lds_w);
#pragma unroll
            for (int p = 0; p < 4; ++p) da[p] += dstep_a;
            db[0] += dstep_b; db[1] += dstep_b;
        } else if constexpr (DMA) {
            lds_dma_1k_s<O + 0 * SA * 8>(dvoff, da[0], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 0 * SB * 8>(dvoff, db[0], lds_w);
            lds_dma_1k_s<O + 4 * SA * 8>(dvoff, da[1], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 4 * SB * 8>(dvoff, db[1], lds_w);
            lds_dma_1k_s<O + 8 * SA * 8>(dvoff, da[2], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 8 * SB * 8>(dvoff, db[2], lds_w);
            lds_dma_1k_s<O + 12 * SA * 8>(dvoff, da[3], lds_w);
            lds_dma_1k_s<O + ASZ * 8 + 12 * SB * 8>(dvoff, db[3], lds_w);
#pragma unroll
            for (int p = 0; p < 4; ++p) { da[p] += dstep_a; db[p] += dstep_b; }
        }
    };
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
        if constexpr (DMA) {
            dma_stage(IC<0>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            gload(k0);
            sstore(0);
        }
        __syncthreads();
        // fragments are double-buffered in registers: the LDS reads of k-substep ks+1 are issued BEFORE the 16 MFMAs of
        // substep ks, so their latency hides behind 1024 cycles of matrix work.  The same holds ACROSS the stage barrier:
        // the last substep's MFMAs of a stage are held back until after the barrier and issued behind the first fragment
        // reads of the next stage (they only need registers), so the wave does not restart each stage with an exposed
        // LDS round trip (measured: the per-stage barrier + restart cost 5.7 % of the K = 512 kernel)
        double fa[2][FM], fb[2][FN];
        auto ldfrag = [&](int b, int ks, int slot) {
            const double* sa = smem + b * STAGE;
            const double* sb = sa + ASZ;
            const int k = ks * 4 + l4;
#pragma unroll
            for (int im = 0; im < FM; ++im) {
                const int m = wm + im * 16 + l15;
                fa[slot][im] = AKC ? sa[m * SK + k] : sa[k * SA + m];
            }
#pragma unroll
            for (int in = 0; in < FN; ++in) {
                const int n = wn + in * 16 + l15;
                if constexpr (BH) fb[slot][in] = sb[(l4 + 4 * (ks >> 1)) * SB + (ks & 1) * 64 + n];
                else fb[slot][in] = BKC ? sb[n * SK + k] : sb[k * SB + n];
            }
        };
        auto mfmas = [&](int slot) {
#pragma unroll
            for (int in = 0; in < FN; ++in)
#pragma unroll
                for (int im = 0; im < FM; ++im)
                    acc[im][in] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[slot][in], fa[slot][im], acc[im][in], 0, 0, 0);
        };
        constexpr int NS = BK / 4;                // k-substeps per stage
        ldfrag(0, 0, 0);
        // the stage buffer is a COMPILE-TIME constant of a k-step (the loop below alternates two instantiations): every LDS
        // address of the step is then the loop-invariant per-lane base plus an immediate, no address arithmetic in the loop
        const unsigned* yf = nullptr;
        if constexpr (YIELD) yf = g.yield_flags + pgp_cu_key();
        auto kstep = [&](int kt, auto bufc, auto pollc) {
            constexpr int buf = decltype(bufc)::value;
            constexpr bool POLL = YIELD && decltype(pollc)::value != 0;       // the yield word is looked at every fourth stage
            const bool more = kt + BK < k1;
            if constexpr (DMA) { if (more) dma_stage(IC<(buf ^ 1)>{}); }     // the other stage was last read before the previous barrier
            else if (more) gload(kt + BK);
            unsigned yv = 0u;
            if constexpr (POLL) yv = __hip_atomic_load(yf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int ks = 0; ks + 1 < NS; ++ks) {
                ldfrag(buf, ks + 1, (ks + 1) & 1);
                mfmas(ks & 1);
            }
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (more) sstore(buf ^ 1);
            __syncthreads();
            if constexpr (POLL) { if (__builtin_amdgcn_readfirstlane(yv) != 0u) gemm_yield_wait(yf); }
            if (more) ldfrag(buf ^ 1, 0, NS & 1); // first fragments of the next stage, behind ...
            __builtin_amdgcn_sched_barrier(0);    // (keep the reads AHEAD of the MFMAs: the scheduler sinks them behind otherwise)
            mfmas((NS - 1) & 1);                  // ... the held-back last substep of this one
        };
        int kt = k0;
        if constexpr (DMA) {
            if (lazyc) {          // first 2 FM FN / 2 k-steps, fully unrolled: chunk c = two accumulator tiles, fetched at step 2c,
                double4_t creg[2];   // folded in at step 2c + 1 (16 VGPRs in flight; every accumulator index is a constant)
                static_assert(FM == 4 && (FN == 4 || FN == 2), "lazy C: 64-row wave tiles, 64 or 32 columns");
                // BH: a wave owns 32 columns = two of the four 16-column groups of a 64-column slab; the waves of the slab's first half
                // fold theirs during steps 0-7, those of the second half during steps 8-15 (wave-uniform predicate)
                const int my_half = __builtin_amdgcn_readfirstlane(wave >> 1);
#pragma unroll
                for (int st = 0; st < LZ; ++st) {
                    const int chunk = (BH ? (st & 7) : st) >> 1, in = chunk / (FM / 2), hf = chunk % (FM / 2);
                    const bool act = !BH || my_half == (st >> 3);
                    if (!(st & 1)) {
                        if (act) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const double* cp = Cin + (long)(i0 + wm + (2 * hf + q) * 16 + l15) + (long)(j0 + wn + in * 16 + l4) * ldcin;
#pragma unroll
                            for (int r = 0; r < 4; ++r) creg[q][r] = cp[(long)(4 * r) * ldcin];
                        }
                        }
                    } else if (act) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[2 * hf + q][in][r] = fma(ab, creg[q][r], acc[2 * hf + q][in][r]);
                    }
                    if (st & 1) kstep(kt, IC<1>{}, IC<0>{});                     // st is an unrolled constant
                    else if ((st & 3) == 0) kstep(kt, IC<0>{}, IC<1>{});
                    else kstep(kt, IC<0>{}, IC<0>{});
                    kt += BK;
                }
            }
        }
        while (kt < k1) {                         // FN * FM is even: the stage parity is 0 here on both paths
            kstep(kt, IC<0>{}, IC<1>{});
            kt += BK;
            if (kt >= k1) break;
            kstep(kt, IC<1>{}, IC<0>{});
            kt += BK;
            if constexpr (YIELD) {                // four stages per trip: one poll
                if (kt >= k1) break;
                kstep(kt, IC<0>{}, IC<0>{});
                kt += BK;
                if (kt >= k1) break;
                kstep(kt, IC<1>{}, IC<0>{});
                kt += BK;
            }
        }
    }

    // (derived HERE, not ahead of the k-loop: two more live SGPRs in there and the LDS-DMA bases no longer fit the scalar file)
    long long* const trc = g.trace ? g.trace + 8L * wg : nullptr;   // workgroup-uniform; wg = the workgroup's number within ITS product
    if (trc) trc[2] = (long long)wall_clock64();             // (every lane the same word: no divergent branch)
    // ---- epilogue: C = alpha * acc ---------------------------------------------------------
    if (!diag && (g.dbg & 512) && !(ldc & 1) && !((unsigned long)C & 15ul)) {
        // 16-byte stores: adjacent lanes (rows m, m+1) trade one value each by a DPP quad_perm, then the even lane stores
        // rows {m, m+1} of column n(r), the odd lane rows {m-1, m} of column n(r+1): half the store instructions (the
        // epilogue is store-ISSUE bound: 64 global_store_dwordx2 per lane otherwise).  C is 16-byte aligned at even rows.
        const bool odd = l15 & 1;
#pragma unroll
        for (int im = 0; im < FM; ++im)
#pragma unroll
            for (int in = 0; in < FN; ++in) {
                const int m = wm + im * 16 + l15;
                const int nb = wn + in * 16 + l4;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const double x = g.alpha * acc[im][in][r], yv = g.alpha * acc[im][in][r + 1];
                    const double nx = swap_adjacent(x), ny = swap_adjacent(yv);
                    const double2_t val = odd ? double2_t{ny, yv} : double2_t{x, nx};
                    double* cp = C + (long)(i0 + (odd ? m - 1 : m)) + (long)(j0 + nb + 4 * (odd ? r + 1 : r)) * ldc;
                    *(double2_t*)cp = val;
                }
            }
    } else {
#pragma unroll
    for (int im = 0; im < FM; ++im)
#pragma unroll
        for (int in = 0; in < FN; ++in) {
            const int m = wm + im * 16 + l15;
            const int nb = wn + in * 16 + l4;
            double* cp = C + (long)(i0 + m) + (long)(j0 + nb) * ldc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 4 * r;
                if (!diag || m >= n) cp[(long)(4 * r) * ldc] = g.alpha * acc[im][in][r];
            }
        }
    }
    if (trc) {
        trc[3] = (long long)wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        trc[4] = (long long)wall_clock64(); trc[5] = (long long)pgp_cu_key(); trc[7] = (long long)__builtin_readcyclecounter();
    }
}

}  // namespace gemm_tile_ns
