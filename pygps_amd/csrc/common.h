// Shared declarations for the MI355X (gfx950) exact-GP core.  Internal header: the public
// C ABI is include/pygps_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define PGP_OK 0
#define PGP_ERR_HIP (-100)
#define PGP_ERR_ARG (-1)

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            pgp_set_last_hip_error(e__, #expr, __FILE__, __LINE__);                         \
            return PGP_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

void pgp_set_last_hip_error(hipError_t e, const char* what, const char* file, int line);

// ------------------------------------------------------------------------------------------
// fp64 MFMA GEMM family (gemm_f64.hip).  All matrices are column-major ("Fortran") views:
//   C(m,n) at C[m + n*ldc].
// Operand storage is described per operand:
//   *_kc == 0 : "M-contiguous"  A(m,k) at A[m + k*lda]   (column-major M x K)
//   *_kc == 1 : "K-contiguous"  A(m,k) at A[k + m*lda]   (column-major K x M, i.e. A^T stored)
// The kernel computes, per 128x128 (or 64x64) tile,
//   C = alpha * sum_{k in range(tile)} A(m,k) B(n,k)  + beta * C        alpha in {+1,-1}, beta in {0,1}
// ------------------------------------------------------------------------------------------
enum GemmKMode {
    KM_FULL = 0,     // k in [0, K)
    KM_GE_I = 1,     // k in [i0 + koff, K)        (i0 = first row of the tile)
    KM_GE_J = 2,     // k in [j0 + koff, K)        (j0 = first column of the tile)
    KM_LT_I = 3,     // k in [0, i0 + TM + koff)
    KM_LT_J = 4,     // k in [0, j0 + TN + koff)
};

struct GemmArgs {
    const double* A; long lda; int a_kc;
    const double* B; long ldb; int b_kc;
    double* C; long ldc;
    int M, N, K;            // multiples of the tile size / 16
    double alpha, beta;
    int tri;                // 1: only tiles with (i0 + tri_off >= j0); diagonal tiles (==) masked to i>=j if mask_diag
    int tri_off;            // row offset of C's row 0 relative to its column 0 in the parent matrix
    int mask_diag;
    int kmode; int koff;
    int batch; long sA, sB, sC;   // batch strides in elements
    int tile;               // 128 or 64; 1264 = 128 x 64 LDS-DMA tiles (plain rectangles only; anything else falls back to 128)
    const int* order;       // optional (ti,tj) pairs per block (XCD-aware / LPT tile order built on the host); grid = norder
    int norder;
    int order_z;            // 1: the list spans a whole batch -- ti carries the batch index in its bits 16.. (grid.z = 1): a shrinking
                            // batch dispatches exactly its tiles instead of batch x (tiles of the largest product)
    int dbg;                // variant bits (pgp_ctx::gemm_dbg): 64 LDS-DMA staging, 256 lazy C, 512 16-byte epilogue stores
    double flops;           // algorithmic flops of this launch (for profiling; filled by caller)
    // Two-piece row spaces (the factor rows and the fused-inverse rows of a Cholesky sweep live in separate buffers so
    // that a posterior handle only keeps the factor).  Tile rows i0 >= *_split (relative to the operand's row 0, a
    // multiple of the tile size; 0 = single piece) are addressed at  X2[(i - split) + k*ldx2].  M-contiguous A only.
    const double* A2; long lda2; int a_split;
    double* C2; long ldc2; int c_split;
    // Optional separate source of the beta*C term (same shape and split point as C): out-of-place updates
    // C = beta*Cin + alpha*A*B' (the look-ahead Cholesky redirects the next panel's columns into a staging buffer).
    const double* Cin; long ldcin; const double* Cin2; long ldcin2;
    int rev_cols;           // rectangular grids: tile columns enumerated last-to-first (KM_LT_J: the long-k tiles start first)
    int rev_rows;           // ... tile rows last-to-first (KM_LT_I: the long-k tiles start first)
    int fold_rows;          // plain LDS-DMA 128-tile rectangles: one workgroup runs the tile rows mt - 1 - r AND r (gemm_f64_fold_kernel)
    int zero_from;          // > 0: tile rows i0 >= zero_from take beta = 0 (rows touched for the first time: never read)
    // Batched launches whose products shrink with the batch index z (the owned column panels of a block-cyclic sweep):
    // product z has M - z * batch_dm rows (tiles beyond them exit at once) and its first-touch row moves up with it.
    int batch_dm;
    int batch_dk;           // ... and the k clip of product z is koff + z * batch_dk (strips of a triangular product: sharded.hip)
    // ... whose C pieces are stored COMPACTLY one behind the other: piece z has leading dimension ldc - z * batch_dldc and
    // starts at C + z * sC - z (z - 1) / 2 * batch_sC2  (column strips of a lower triangle, each as tall as it needs to be)
    int batch_dldc; long batch_sC2;
    // Cooperative yield (see gemm_tile.h): a per-CU word counts the workgroups of the latency-critical diagonal-panel chain that
    // are resident on that CU.  yield_role 1 (bulk launches): the k-loop polls its CU's word once per stage and sleeps while it
    // is non-zero; yield_role 2 (the chain's own small GEMMs): the workgroup increments the word while it runs.
    unsigned* yield_flags; int yield_role;
    // Tiles whose first row AND first column lie in [skip_lo, skip_hi) are left alone (0, 0: none): the diagonal block another
    // kernel owns inside a whole-matrix update (EP's block sweep: the next block's tile belongs to its prep workgroups).
    int skip_lo, skip_hi;
    // Device-side dependency (0: none): every workgroup waits until the counter *wait_flag has reached wait_target before it reads
    // anything -- an operand another RESIDENT kernel publishes (EP's sweep kernel: W of the block).  Bounded; a timeout sets *wait_err.
    unsigned* wait_flag; unsigned wait_target; unsigned* wait_err;
    // phase stamps of every workgroup (pgp_test_gemm_trace; nullptr otherwise): 8 words per block -- 100 MHz wall clock at kernel
    // entry, (unused), behind the k-loop, with the epilogue's stores issued, with them acknowledged; the CU key; the shader-clock
    // counter at entry and at the end
    long long* trace;
};

// index of the calling workgroup's CU in a yield-flag table (XCC id | shader engine, array, CU of HW_ID): < 4096
__device__ __forceinline__ unsigned pgp_cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u; // HW_REG_XCC_ID
    return (xcc << 8) | ((hw >> 8) & 0xffu);
}
// the chain's kernels: +1 on entry, -1 on exit (thread 0; the caller puts a workgroup barrier before the exit mark)
__device__ __forceinline__ void pgp_yield_mark(unsigned* flags, int delta) {
    if (flags && threadIdx.x == 0) {
        if (delta > 0) __hip_atomic_fetch_add(flags + pgp_cu_key(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_sub(flags + pgp_cu_key(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int gemm_f64_launch(const GemmArgs& g, hipStream_t st);
bool gemm_f64_pair_ok(const GemmArgs& a, const GemmArgs& b);     // may the two go out as ONE launch (gemm_f64_launch_pair)?
int gemm_f64_launch_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t st);
bool gemm_f64_uses_dma(const GemmArgs& g);          // ... or any LDS-DMA tile (128 x 128, or 128 x 64 = GemmArgs::tile 1264)?
bool gemm_f64_uses_dma128(const GemmArgs& g);       // would gemm_f64_launch pick the LDS-DMA 128-tile instantiation?

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that needs more than 64 KB of dynamic LDS: applied once per
// (kernel, DEVICE, size) -- the attribute belongs to the device's code object, a process may hold contexts on several devices
// (_lib.ctx(device)), and two fit streams (host threads) launch concurrently (capi.hip).
void func_max_dynamic_lds(const void* fn, size_t bytes);
