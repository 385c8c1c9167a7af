/* pygps_amd -- C ABI of the MI355X-native exact-GP core (drop-in for the hot path of
 * marionmari/pyGPs: Core/cov.py kernel-matrix construction + Core/inf.py Exact/EP inference +
 * Core/tools.py jitchol/solve_chol + Core/gp.py predict).
 *
 * Plain C, caller-owned host buffers, fp64, numpy row-major unless stated.  No exception crosses
 * the ABI.  Every function returns an int status:
 *      0            ok
 *     >0            LAPACK-style info: 1-based index of the first non-positive pivot
 *                   (the shim raises numpy.linalg.LinAlgError, as Core/tools.py:66-77 does)
 *     -1 .. -99     bad argument (index of the offending argument; the shim raises the reference's
 *                   plain Exception with the reference's message)
 *     <= -100       HIP runtime failure (pgp_strerror gives the text; the shim raises RuntimeError)
 *
 * One pgp_ctx = one device + one HIP stream + one workspace pool; NOT thread-safe; calls are
 * synchronous (stream-synchronised before return).  Multi-GPU = one process / one ctx per GPU.
 *
 * The Python binding a pyGPs maintainer would add is pygps_amd/_lib.py (ctypes); see INTEGRATION.md.
 */
#ifndef PYGPS_AMD_H
#define PYGPS_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgp_ctx pgp_ctx;
typedef struct pgp_factor pgp_factor;
/* device-resident posterior: factor R, alpha, sW, inputs */
typedef struct pgp_fitc pgp_fitc;     /* FITC posterior on the device (alpha, dense L, inducing coordinates) */

/* covariance kinds            reference class                      */
#define PGP_COV_RBF 0     /* Core/cov.py:786-828   hyp=[log ell, log sf]               */
#define PGP_COV_RBFARD 1  /* Core/cov.py:872-938   hyp=[log ell_1..log ell_D, log sf]   (any D with nhyp <= 255) */
#define PGP_COV_MATERN 2  /* Core/cov.py:1078-1182 hyp=[log ell, log sf], para=d in {1,3,5,7} */
#define PGP_COV_RBFUNIT 3 /* Core/cov.py:832-869   hyp=[log ell]                                  */
#define PGP_COV_RQ 4      /* Core/cov.py:1304-1347 hyp=[log ell, log sf, log alpha]               */
#define PGP_COV_PIECEPOLY 5 /* Core/cov.py:683-782 hyp=[log ell, log sf], para=v in {0,1,2,3}      */
#define PGP_COV_RQARD 6   /* Core/cov.py:1356-1425 hyp=[log ell_1..log ell_D, log sf, log alpha]     */
#define PGP_COV_GABOR 7   /* Core/cov.py:392-450   hyp=[log ell, log p]  (p = exp(2 hyp1), as the reference) */
#define PGP_COV_PERIODIC 8 /* Core/cov.py:1186-1250 hyp=[log ell, log p, log sf], 1-d inputs only    */
#define PGP_COV_NOISE 9   /* Core/cov.py:1254-1300 hyp=[log sf]                                      */
#define PGP_COV_CONST 10  /* Core/cov.py:941-982   hyp=[log sf]  (sf2 = exp(hyp0), as the reference) */
#define PGP_COV_NKIND 11
/* Sum / Product / Scale tree over primitives (Core/cov.py:230-328; up to two ARD leaves), registered with
 * pgp_set_composite and selected by kind = PGP_COV_COMPOSITE in pgp_cov / pgp_exact_fit / pgp_ep_fit.
 * hyp is the composite's flattened list in the reference's order (cov1.hyp + cov2.hyp; [scalar] + cov.hyp). */
#define PGP_COV_COMPOSITE 100
/* postfix program tokens */
#define PGP_PROG_LEAF 1    /* LEAF kind para flags first_hyp_index */
#define PGP_PROG_SUM 2
#define PGP_PROG_PRODUCT 3
#define PGP_PROG_SCALE 4   /* SCALE hyp_index */
/* modes of getCovMatrix / getDerMatrix (Core/cov.py:81-111) */
#define PGP_MODE_TRAIN 0
#define PGP_MODE_CROSS 1
#define PGP_MODE_SELF_TEST 2
/* flags */
#define PGP_FLAG_MATERN_REFERENCE_DER 1 /* reproduce the reference's derivative quirks: Matern Core/cov.py:1173-1177
                                         * (derivative of K, not t); RQard :1412-1418 (length-scale derivatives) */

/* ---- lifetime ---------------------------------------------------------------------------- */
int pgp_init(int device, pgp_ctx** ctx_out);
void pgp_destroy(pgp_ctx* ctx);
const char* pgp_strerror(int status);
const char* pgp_version(void);
int pgp_device_count(void);              /* visible HIP devices (0 when there is none or no driver) */
int pgp_device_info(pgp_ctx* ctx, int* n_cu, int* sclk_mhz, double* hbm_gib, char* name, int name_len);

/* ---- kernel plug-in: Kernel.getCovMatrix / getDerMatrix (Core/cov.py:81-111) ---------------
 * der < 0 : covariance value; der >= 0 : derivative w.r.t. hyper `der` (log-space).
 * TRAIN: x (n,d) -> out (n,n).  CROSS: x (n,d), z (m,d) -> out (n,m).  SELF_TEST: z (m,d) -> out (m,1).
 * Error codes: -3 unknown mode, -4 derivative index does not exist, -5/-7 missing x / z.        */
int pgp_cov(pgp_ctx* ctx, int kind, int mode, int der, const double* x, int64_t n, const double* z, int64_t m,
            int64_t d, const double* hyp, int nhyp, int para, int flags, double* out);

/* Composite kernels: ProductOfKernel / SumOfKernel / ScaleOfKernel (Core/cov.py:230-328).  prog is the tree in
 * postfix order (PGP_PROG_* tokens); it stays registered in the context until replaced.  Limits: 8 leaves, 8 Scale
 * nodes, 8 products after distributing products over sums; at most TWO ARD leaves (RBFard / RQard), each with D <= 64
 * (their 1 / ell_k^2 weights travel as kernel arguments); anything beyond returns -13 from the calls that use the program
 * -- the Python layer then takes the dense path (pgp_exact_fit_dense / pgp_ep_fit_dense).  The PLAIN kinds
 * PGP_COV_RBFARD / PGP_COV_RQARD take any D (the scales are folded into the coordinates). */
int pgp_set_composite(pgp_ctx* ctx, const int32_t* prog, int nprog);

/* ---- data residency -------------------------------------------------------------------------
 * The optimiser calls the fit hundreds of times with identical (x, y) and only hyp changing
 * (Core/opt.py:70-75), so x and y are uploaded once.                                            */
int pgp_set_data(pgp_ctx* ctx, const double* x, int64_t n, int64_t d, const double* y);

/* ---- inference plug-in: Exact.evaluate (Core/inf.py:353-384) ---------------------------------
 * Uses the data of the last pgp_set_data.  mvec: prior mean m (n); dm: (nmean, n) rows = derivative
 * of m w.r.t. each mean hyper (NULL if nmean == 0).  want = nargout (1: post, 2: +nlZ, 3: +dnlZ).
 * alpha_out (n), nlZ_out (1), dnlZ_out (nmean + ncov + 1, order mean|cov|lik as Core/opt.py:77-80).
 * factor_out (optional): handle keeping R (upper, R'R = K/sn2 + I), alpha, sW on the device.   */
int pgp_exact_fit(pgp_ctx* ctx, int kind, const double* covhyp, int ncov, int para, int flags, double log_sn,
                  const double* mvec, const double* dm, int nmean, int want, double* alpha_out, double* nlZ_out,
                  double* dnlZ_out, pgp_factor** factor_out);

/* post.L as numpy expects it: (n,n) row-major UPPER factor, exact zeros below the diagonal
 * (Core/gp.py:393 branches on that).                                                           */
int pgp_factor_to_host(pgp_ctx* ctx, pgp_factor* f, double* L_out);
int64_t pgp_factor_n(pgp_factor* f);
void pgp_factor_free(pgp_ctx* ctx, pgp_factor* f);

/* ---- GP.predict (Core/gp.py:349-437), Cholesky parametrisation -------------------------------
 * fmu = ms + Ks' alpha ; fs2 = max(kss - colsum((R'^-1 (sW o Ks))^2), 0).                       */
int pgp_predict(pgp_ctx* ctx, pgp_factor* f, const double* xs, int64_t ns, const double* ms, double* fmu,
                double* fs2);

/* ---- the same path from CALLER-BUILT covariance matrices (csrc/dense.hip) ---------------------------------
 * For covariance functions that are not device programs (Core/cov.py:230-328 composes anything; a tree with more than two
 * ARD leaves or more than 8 leaves has getCovMatrix / getDerMatrix only): K (n,n) = getCovMatrix(x, 'train'), r = y - m.
 * pgp_exact_fit_dense: want as pgp_exact_fit; dnlZ_lik_out = sn2 tr Q.  pgp_dense_grad_term: 1/2 sum(Q o dK) for one
 * derivative matrix dK = getDerMatrix(x, 'train', h), with the Q of the dense fit that directly precedes it on this context
 * (Core/inf.py:373-377).  pgp_predict_dense: Ks (n,ns) = getCovMatrix(x, xs, 'cross'), kss (ns) = 'self_test'.        */
int pgp_exact_fit_dense(pgp_ctx* ctx, const double* K, int64_t n, const double* r, double log_sn, int want,
                        double* alpha_out, double* nlZ_out, double* dnlZ_lik_out, pgp_factor** factor_out);
int pgp_dense_grad_term(pgp_ctx* ctx, const double* dK, int64_t n, double log_sn, double* out);
int pgp_predict_dense(pgp_ctx* ctx, pgp_factor* f, const double* Ks, int64_t ns, const double* kss, const double* ms,
                      double* fmu, double* fs2);

/* ---- EP.evaluate with lik.Erf (Core/inf.py:731-806, 174-189; Core/lik.py:295-366) ----------
 * ttau/tnu (n): in = warm start (ignored if warm == 0), out = final site parameters.           */
int pgp_ep_fit(pgp_ctx* ctx, int kind, const double* covhyp, int ncov, int para, int flags, const double* mvec,
               const double* dm, int nmean, int want, int warm, double* ttau, double* tnu, double* alpha_out,
               double* sW_out, double* nlZ_out, double* dnlZ_out, int* sweeps_out, pgp_factor** factor_out);

/* EP.evaluate (Core/inf.py:723-806) from a CALLER-BUILT covariance matrix: the covariance trees of Core/cov.py:230-328 that are
 * not device programs.  K (n, n) symmetric row-major host; n, y as set by pgp_set_data (x is not used).  dnlZ_mean_out: nmean + 1
 * entries (mean gradients, then 0: lik.Erf has no hyper-parameter).  want = 3 leaves sW sW' o B^-1 and alpha in the context's
 * workspace: pgp_dense_grad_term(ctx, dK_h, n, 0.0, &g) directly afterwards gives dnlZ.cov[h] (Core/inf.py:780-786).
 * Other arguments and status codes as pgp_ep_fit. */
int pgp_ep_fit_dense(pgp_ctx* ctx, const double* K, const double* mvec, const double* dm, int nmean, int want, int warm,
                     double* ttau_io, double* tnu_io, double* alpha_out, double* sW_out, double* nlZ_out,
                     double* dnlZ_mean_out, int* sweeps_out, pgp_factor** factor_out);

/* ---- FITC sparse regression: FITC_Exact.evaluate (Core/inf.py:398-455) with FITCOfKernel (Core/cov.py:332-390)
 * x, y of the last pgp_set_data; xu (nu,d) inducing inputs.  alpha_out (nu), L_out (nu,nu) = post.L (dense,
 * symmetric; NULL to skip), dnlZ_out = [mean.., cov.., lik].  snu2 = 1e-6 sn2 like the reference (inf.py:410).
 * Returns >0 when a Cholesky pivot is not positive.  handle_out (optional) feeds pgp_fitc_predict.               */
int pgp_fitc_fit(pgp_ctx* ctx, int kind, const double* covhyp, int ncov, int para, int flags, double log_sn,
                 const double* xu, int64_t nu, const double* mvec, const double* dm, int nmean, int want,
                 double* alpha_out, double* L_out, double* nlZ_out, double* dnlZ_out, pgp_fitc** handle_out);
/* the dense-L branch of GP.predict (Core/gp.py:404-417): fmu = ms + Ks'alpha, fs2 = max(kss + colsum(Ks o (L Ks)), 0) */
int pgp_fitc_predict(pgp_ctx* ctx, pgp_fitc* f, const double* xs, int64_t ns, const double* ms, double* fmu,
                     double* fs2);
void pgp_fitc_free(pgp_ctx* ctx, pgp_fitc* f);

/* ---- ONE fit over the GPUs of a node (no reference counterpart: Core/inf.py:353-384 runs on one host) ---
 * One process per GPU, every rank calls with the same data (pgp_set_data) and arguments.  The factorisation is 1-D
 * block-cyclic over column panels; panels are broadcast over `comm`, alpha / nlZ / dnlZ come back identical on every
 * rank (csrc/sharded.hip).  Transports:
 *   pgp_comm_init_rccl: RCCL bound at run time (rccl_path: the librccl to dlopen, NULL = the system's); rank 0 makes the
 *                       128-byte id with pgp_comm_unique_id and hands it to the other ranks (any side channel).
 *   pgp_comm_init_host: call-backs that broadcast / all-reduce HOST buffers (op: 0 sum, 1 max; return 0 on success); the
 *                       library stages device memory through pinned host memory.  For MPI / gloo style transports and for
 *                       tests in which several ranks share one GPU. */
typedef struct pgp_comm pgp_comm;
typedef int (*pgp_host_bcast_fn)(void* user, void* buf, int64_t bytes, int root);
typedef int (*pgp_host_allreduce_fn)(void* user, double* buf, int64_t count, int op);
int pgp_comm_unique_id(const char* rccl_path, char* id_out /* 128 bytes */);
int pgp_comm_init_rccl(pgp_ctx* ctx, int world, int rank, const char* id /* 128 bytes */, const char* rccl_path,
                       pgp_comm** comm_out);
int pgp_comm_init_host(pgp_ctx* ctx, int world, int rank, pgp_host_bcast_fn bcast, pgp_host_allreduce_fn allreduce,
                       void* user, pgp_comm** comm_out);
void pgp_comm_free(pgp_comm* comm);
int pgp_comm_world(pgp_comm* comm);
int pgp_comm_rank(pgp_comm* comm);
/* Host collectives on the communicator: what the sharded restart search (Core/opt.py:301-327: init table + data out, one record
 * per restart back), the sharded K-fold loop (Validation/valid.py:20-66) and a sum of nlZ / dnlZ over independent data sets
 * (Demo/Clustering/pyGP_extension.py:54-64) exchange.  Buffers are HOST doubles; every rank calls with the same count.
 * RCCL transport: staged through a device buffer on the communicator's stream (ncclBroadcast / ncclAllReduce / ncclAllGather);
 * host transport: the call-backs.  pgp_comm_init_host accepts ctx = NULL for a communicator that only ever serves these
 * three (no device is touched).  allgather: recv holds world * count doubles, rank r's at recv + r * count.  op: 0 sum, 1 max. */
int pgp_comm_bcast_host(pgp_comm* comm, double* buf, int64_t count, int root);
int pgp_comm_allreduce_host(pgp_comm* comm, double* buf, int64_t count, int op);
int pgp_comm_allgather_host(pgp_comm* comm, const double* send, int64_t count, double* recv);
/* Arguments and results as pgp_exact_fit.  timings_out (optional, 10): ms of assembly, sweep, epilogue, total; device bytes
 * this call held at its peak; device bytes the posterior handle keeps; [6] ms the compute stream stalled waiting for a panel
 * (sum over the panels), [7] ms of the panel broadcasts from enqueue to complete on this rank (sum), [8] bytes this rank moved
 * in them, [9] the slowest single broadcast (ms) -- [6..9] are 0 at world size 1.  L_out (optional, (n,n) row-major, zero-filled by the
 * caller): THIS rank's columns of the factor in post.L's form (upper R, R'R = K/sn2 + I); the sum over the ranks is the whole
 * factor.  factor_out (optional): this rank's part of the distributed posterior (its column panels of L and of L^-T, alpha, the
 * coordinates) for pgp_sharded_predict.  Per-rank memory is O(n^2 / world): the panels, and for want = 3 the rank's column
 * strips of B^-1.  A rank that fails (out of memory, a launch error) makes EVERY rank return an error: its flag rides in the
 * collectives the others wait in.  (A sticky device fault, after which no HIP call succeeds, cannot enqueue that collective: the
 * failing rank then aborts the RCCL communicator so that the peers' collectives return with an error.) */
typedef struct pgp_sfactor pgp_sfactor;
int pgp_sharded_exact_fit(pgp_ctx* ctx, pgp_comm* comm, int kind, const double* covhyp, int ncov, int para, int flags,
                          double log_sn, const double* mvec, const double* dm, int nmean, int want, double* alpha_out,
                          double* nlZ_out, double* dnlZ_out, double* timings_out, double* L_out, pgp_sfactor** factor_out);
/* GP.predict (Core/gp.py:395-417) on the distributed posterior; every rank calls with the same test points (ns, d) and
 * receives the same fmu / fs2.  One all-reduce of ns doubles per batch of test points. */
int pgp_sharded_predict(pgp_ctx* ctx, pgp_comm* comm, pgp_sfactor* f, const double* xs, int64_t ns, const double* ms,
                        double* fmu, double* fs2);
void pgp_sfactor_free(pgp_ctx* ctx, pgp_sfactor* f);
int64_t pgp_sfactor_bytes(pgp_sfactor* f);

/* ---- helper functions: tools.jitchol / tools.solve_chol (Core/tools.py:31-97) ----------------
 * pgp_potrf: A (n,n) symmetric row-major in -> lower Cholesky factor (row-major, zeros above) out.
 * pgp_potrs: R (n,n) UPPER factor row-major, Bm (n,nrhs) row-major in -> X = (R'R)^-1 Bm out.   */
int pgp_potrf(pgp_ctx* ctx, const double* A, int64_t n, double* L_out);
int pgp_potrs(pgp_ctx* ctx, const double* R, int64_t n, const double* Bm, int64_t nrhs, double* X_out);

/* ---- measurement ------------------------------------------------------------------------------
 * Stage timings (ms, HIP events on the ctx stream) of the last fit, and -- when profiling is on --
 * per-kernel-class totals (launches, ms, algorithmic flops, algorithmic bytes).                 */
#define PGP_STAGE_ASSEMBLE 0
#define PGP_STAGE_POTRF 1
#define PGP_STAGE_SOLVE 2
#define PGP_STAGE_TRTRI 3
#define PGP_STAGE_LAUUM 4
#define PGP_STAGE_GRAD 5
#define PGP_STAGE_TOTAL 6
#define PGP_NSTAGE 7
int pgp_last_timings(pgp_ctx* ctx, double* ms_out /* PGP_NSTAGE */);
int pgp_set_profiling(pgp_ctx* ctx, int on);
int pgp_profile_classes(void);
const char* pgp_profile_class_name(int cls);
int pgp_profile_read(pgp_ctx* ctx, int cls, int64_t* launches, double* ms, double* flops, double* bytes);
int pgp_profile_reset(pgp_ctx* ctx);
/* tuning knobs (outer panel width of the blocked Cholesky, look-ahead on/off ...) */
int pgp_set_option(pgp_ctx* ctx, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
