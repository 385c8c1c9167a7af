/* Self-test / calibration hooks of libpygps_amd.so: NOT part of the drop-in boundary (include/pygps_amd.h).  They are
 * exported for tests/, tools/ and bench.py's calibration legs only (pygps_amd/_lib.py: TEST_SIGNATURES). */
#pragma once
#include <stdint.h>

#include "../../include/pygps_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Column-major GEMM on host buffers through the fp64 MFMA kernel. */
int pgp_test_gemm(pgp_ctx* ctx, int tile, int a_kc, int b_kc, int tri, int mask_diag, int kmode, int koff,
                  double alpha, double beta, const double* A, int64_t lda, const double* B, int64_t ldb,
                  double* C, int64_t ldc, int M, int N, int K, int iters, double* ms_out);
int pgp_test_gemm_shrink(pgp_ctx* ctx, const double* Y, int64_t ldy, int M, int K, int w, int nb, int dm, int zero_from,
                         double* C, int64_t ldc, int64_t sC);
/* C -= A B' on the lower tiles (packed, masked diagonal tiles) but those whose first row and column lie in [skip_lo, skip_hi); with
   wait_ms > 0 every workgroup waits inside the kernel for a device counter that the second stream raises wait_ms later. */
int pgp_test_gemm_skip_wait(pgp_ctx* ctx, int tile, const double* A, const double* B, double* C, int n, int K, int skip_lo,
                            int skip_hi, int wait_ms, int* timed_out);
int pgp_test_probit_hazard(pgp_ctx* ctx, const double* z, double* out, int n);
int pgp_test_valu_peak(pgp_ctx* ctx, int iters, int waves_per_simd, double* out2);
int pgp_test_mfma_peak(pgp_ctx* ctx, int iters, double* tflops_out);
int pgp_test_mfma_cycles(pgp_ctx* ctx, int iters, int nacc, int waves_per_simd, double* out3);
int pgp_test_leaf_ticks(pgp_ctx* ctx, double* ticks_out /* 24 */);
int pgp_test_wave_costs(pgp_ctx* ctx, double* out16);
int pgp_test_slot_probe(pgp_ctx* ctx, int nwg, int lds_kb, int hold_us, int reserve, int probe_lds_kb, int delay_us, double* out4);
int pgp_test_cumask_gemm(pgp_ctx* ctx, int M, int K, int reserve_per_xcd, int stride, int iters, double* out2);
int pgp_test_assemble(pgp_ctx* ctx, int kind, int mode, int64_t n, int64_t d, int iters, double* ms_out);
/* the stores of the 'train' assembly alone (out3[0], ms), hipMemsetAsync (out3[1]) and a linear fill (out3[2]) over 8 n^2 bytes */
/* phase stamps (100 MHz) of every workgroup of one trailing-update launch: 8 words per workgroup, see GemmArgs::trace */
int pgp_test_gemm_trace(pgp_ctx* ctx, int M, int K, int tri, int warm, int conc, long long* out, int64_t out_words, int64_t* nblk_out);
int pgp_test_read_gemm_trace(pgp_ctx* ctx, long long* out, int64_t words, int64_t* nwg);
int pgp_test_store_roof(pgp_ctx* ctx, int64_t n, int grid, int iters, double* out3);
/* out2[0] = 1 if the context's two streams ran concurrently (a spinner on the panel stream saw a flag set from the main stream), out2[1] = us waited */
int pgp_test_stream_concurrency(pgp_ctx* ctx, int wait_us, double* out2);
#ifdef __cplusplus
}
#endif
