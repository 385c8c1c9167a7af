// Launch wrappers of the device kernels (internal).
#pragma once
#include "common.h"

struct CovParams;
struct CovSpec;

// panel.hip
constexpr long PACK_DOUBLES = 36 * 256;   // per-leaf packed operand image: 28 strictly-lower L blocks + 8 inverted pivot blocks
int leaf_potrf_launch(double* A, long lda, double* inv16, int* info, int info_base, hipStream_t st,
                      long long* tick = nullptr, unsigned* yield_flags = nullptr, int pivot = 1);
int trsm_rows_launch(double* X, long ldx, long nrows, const double* Ld, long ldl, const double* inv16,
                     hipStream_t st, unsigned* yield_flags = nullptr, bool lean = false);   // lean: no LDS, < 64 VGPRs (beside bulk work)
int leaf_inv_launch(const double* L, long ldl, double* W, long ldw, long wstride, int nblocks, hipStream_t st);
int trsv_bwd_launch(const double* L, long ldl, const double* W, long ldw, double* z, double* a_out, int nblk,
                    hipStream_t st);
int diag_in_launch(const double* src, long lds, double* D, long ldd, int w, hipStream_t st);
int diag_out_launch(const double* D, long ldd, int w, double* Fd, long ldf, double* Ed, long lde, hipStream_t st);
int gather_strided_launch(const double* src, long stride, long n, double* dst, hipStream_t st);
int publish_launch(const double* src_dev, double* dst_host_mapped, long count, hipStream_t st);    // results -> pinned host memory, by a kernel

// assemble.hip
int scale_transpose_launch(const double* x, long n, int d, const double* scale_dev, double* XsT, long ldp, int dpad,
                           hipStream_t st);
int cov_sym_launch(const double* XT, long ldp, long n, int dpad, const CovSpec& cs, double* out, hipStream_t st,
                   long ldo = 0);
int cov_rect_launch(const double* XrT, long ldr, long n, const double* XcT, long ldc, long m, int dpad,
                    const CovSpec& cs, double* out, long ldo, hipStream_t st);
int cov_factor_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double inv_sn2,
                      double* Bf, long ldf, hipStream_t st);
int cov_factor_panel_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double inv_sn2,
                            long col0, long ncols, double* panel, long ldpanel, hipStream_t st);
int cov_self_launch(const CovSpec& cs, int train, double* out_dev, hipStream_t st);
int self_fill_launch(double* out, long m, double val, hipStream_t st);

// grad.hip
int hadamard_reduce_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, int ncov,
                           double sn2, const double* Binv, long ldb, const double* alpha, double* partial,
                           double* out_dev, hipStream_t st, const double* wv = nullptr, const double* prep = nullptr);
long hadamard_partial_count(long np, int ncov);
// the same reduce in pieces, for a B^-1 that exists as column strips only (csrc/sharded.hip)
long hadamard_block_count(long np, long tr0, long trn);
constexpr long HADAMARD_PREP_MU = 272;               // prep buffer: [means of up to 256 coordinates + slack | np squared norms]
long hadamard_prep_count(long np);
int hadamard_prepare_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double* mu, hipStream_t st,
                            bool force = false);
// Gram-form (MFMA) assembly of RBF / RBFard values from centred coordinates; prep as written by hadamard_prepare_launch(force)
bool cov_gram_applies(const CovSpec& cs, int dpad);
int cov_factor_gram_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double inv_sn2, double* Bf,
                           long ldf, const double* prep, hipStream_t st);
int cov_sym_gram_launch(const double* XT, long ldp, long n, int dpad, const CovSpec& cs, double* out, long ldo, const double* prep,
                        hipStream_t st);
int hadamard_partial_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, int ncov, double sn2,
                            const double* Binv, long ldb, const double* alpha, const double* wv, double* partial,
                            const double* mu, long tr0, long trn, hipStream_t st);
int hadamard_final_launch(const double* partial, long nblk, int ncov, double* out_dev, hipStream_t st);
int colsumsq_acc_launch(const double* A, long lda, long nrows, long ncols, double* acc, hipStream_t st);
int col_dot_launch(const double* W, long ldw, long n, const double* z, long zs, double scale, double* y,
                   hipStream_t st);
int logdet_ztz_launch(const double* L, long ldl, long n, const double* z, long zs, double* out, hipStream_t st);
int dot2_launch(const double* u, const double* v, long n, double* out, hipStream_t st);
int aug_rhs_launch(const double* y, const double* m, long n, double* F, long ldf, long row, double* rvec,
                   hipStream_t st);
int zero_strip_launch(double* F, long ldf, long np, long row0, long nrows, hipStream_t st);

// grad.hip (column reductions used by predict)
int col_dot_full_launch(const double* A, long lda, long nrows, long ncols, const double* v, const double* add,
                        double* out, hipStream_t st);
int col_sumsq_launch(const double* A, long lda, long nrows, long ncols, double kss, double scale, double* out,
                     hipStream_t st);
int row_scale_launch(double* A, long lda, long nrows, long ncols, const double* s, hipStream_t st);

// grad.hip (fused-inverse helpers)
int identity_upper_launch(double* E, long lde, long np, hipStream_t st);
int upper_matvec_launch(const double* E, long lde, long np, const double* z, double scale, double* partial, double* y,
                        hipStream_t st);
