// Internal: context / factor structs shared by the host-side drivers (capi.hip, predict.hip, ep.hip).
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <mutex>
#include <vector>

#include "../../include/pygps_amd.h"
#include "kernels.h"
#include "sqdist_tile.h"

enum ProfClass { PC_ASSEMBLE = 0, PC_GEMM_TRAIL, PC_GEMM_INNER, PC_LEAF, PC_TRSM, PC_LEAFINV, PC_GEMM_TRTRI,
                 PC_GEMM_LAUUM, PC_HADAMARD, PC_SMALL, PC_GEMM_SOLVE, PC_DIAG, PC_KERNEL_DMA128, PC_COUNT };
static const char* const kProfNames[PC_COUNT] = {
    "cov_tile_kernel(assemble)", "gemm_f64(potrf trailing syrk)", "gemm_f64(potrf inner update)",
    "leaf_potrf_kernel", "trsm_rows_kernel", "leaf_inv_kernel", "gemm_f64(trtri)", "gemm_f64(lauum W^T W)",
    "hadamard_reduce_kernel", "small/O(N) kernels", "gemm_f64(panel solve X E_D)", "diag_in/out staging",
    // shadow class: every launch of the dominant kernel INSTANTIATION (one row of a rocprofv3 kernel-stats CSV), whatever
    // its purpose class above -- each such launch is counted here AND in its purpose class
    "kernel gemm_f64_kernel<128,128,false,false,true,*> + gemm_f64_pair_kernel (LDS-DMA 128-tile)"};

struct ProfRec { int cls; hipEvent_t e0, e1; double flops, bytes; int shadow; };

struct pgp_factor {
    long n, np, ldf;
    double* F;            // (ldf x np) column-major lower factor == row-major upper R (+ augmented rows)
    double* alpha;        // n
    double* XsT;          // dpad x np scaled coordinates used for this fit
    double* Wd;           // np x 128 : inverted diagonal blocks (lazy, for predict)
    double* Linv = nullptr; // ldf x np : W = L^-1 (lazy: the product form of predict, predict.hip ensure_linv; from the factor pool)
    double* Eraw = nullptr; // np x np  : the fit's fused inverse rows E = L^-T as the sweep left them (option keep_inverse; scratch-pool buffer of
    size_t Eraw_bytes = 0;  //            Eraw_bytes): the first predict transposes them into Linv instead of running a trtri
    int dpad, d;
    CovSpec cs;           // covariance functor / program of this fit (predict re-evaluates it in 'cross' mode)
    double kss = 0.0;     // k(z,z) of 'self_test' mode
    double sn2;
    double sw;            // sW entries (1/sqrt(sn2)) for Exact
    double* sWv = nullptr; // per-point sW (EP); nullptr for Exact
};

constexpr int PGP_SCHED_DEFAULT = 2;

struct pgp_ctx {
    int device = 0;
    std::vector<int> composite;         // postfix program of kind PGP_COV_COMPOSITE (pgp_set_composite)
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr;          // panel stream of the look-ahead Cholesky (high priority)
    std::vector<hipEvent_t> fill_ev;
    std::vector<hipEvent_t> ep_ev;      // EP block sweep: the two events around a sweep (bulk stream -> chain stream and back)
    double* eet_out = nullptr;          // set by the fit for the duration of one sweep: where the filler accumulates B^-1
    long eet_ld = 0;
    double* fill2_C = nullptr;          // potrf_blocked_rhs: symmetric matrix (lower tiles) that receives -= V' V'^T panel by panel, or null
    long fill2_ld = 0;
    hipEvent_t eet_join = nullptr;      // non-null: the sweep queued every panel product; the fit joins on this event
    int eet_tile = 128;                 // tile size of the filler products (64: shorter workgroups in the way of the chain)
    int eet_first = -1;                 // inline filler: panels 0 .. eet_first are folded into one product (-1: a sixth of the panels)
    int eet_overlap = 3;                // B^-1 = sum_p E_p E_p^T accumulated under the sweep: 0 off (one product after it), 2 behind
                                        // every TU_b on the main stream, 3 the same when npanel <= eet_max_panels
    int eet_max_panels = 32;
    std::vector<hipEvent_t> la_ev;      // look-ahead hand-off events
    std::vector<hipEvent_t> tm_ev;      // timing events of the sharded fit's wait / broadcast timers (4 per panel)
    int lookahead = 1;
    bool sched_explicit = false;        // "sched" was set to a value >= 0 by the caller: the concurrent-streams hint does not override it
    bool concurrent_streams = false;    // option "concurrent_streams" (set by _lib.concurrent_fit_streams while fit streams run side by side)
    int sched = PGP_SCHED_DEFAULT;      // option "sched" (-1 = this default).  0 = rounds 2-4: S, TU_a, TU_b on the main stream, only D on the
                                        // panel stream; 1 = the critical path D -> S -> TU_a on the panel stream, the bulk updates on the main
                                        // stream (what fit streams that run side by side select); 2 = like 0, but the piece of TU_a that D(p+1)
                                        // needs -- the next panel's diagonal block -- runs on the panel stream right behind S(p)
                                        // (potrf_blocked_v2; lone chain at N = 8192: 11.13 -> 10.90 ms)
    int tur_tile = 0;                   // sched 2: tiles of the rectangle below it (TU_r, main stream): 128, or 1264 = 128 x 64 LDS-DMA tiles (two workgroups per
                                        // CU: 78 -> 70 us alone on the chip at N = 8192); 0 = by size: 1264 up to N = 5120 (N = 4096: two streams +2 ... 3 %,
                                        // single chain equal), 128 beyond (N = 8192: the chain's kernels find no free slot beside two workgroups per CU:
                                        // single chain 10.52 -> 10.63 ms)
    int tud_tile = 64;                  // sched 2: tile size of the diagonal-block piece of TU_a on the panel stream
    int sched2_wide = 0;                // sched 2 also with panels wider than 512 columns (measured slower: N = 16384 +2 %)
    int s_pan = -1;                     // sched 2: S(p), p >= 1, on the panel stream right behind D(p)'s leaf chain: it runs in the tail of the
                                        // previous paired launch instead of after it (-1 / 1 = on; 2 = on and marked like the chain's products;
                                        // 0 = on the main stream).  With s_pan_direct and s_pan_out: N = 8192 10.93 -> 10.53 ms, 7168 7.91 -> 7.80,
                                        // 6144 5.71 -> 5.61, 5120 neutral, 4096 2.84 -> 2.78
    int s_pan_direct = 1;               // s_pan: TU_d(p) waits for the paired launch of panel p - 1 by its own event (0: through the main stream)
    int s_pan_out = 1;                  // s_pan: D(p)'s stage-out on the main stream beside S(p) instead of on the chain (scratch double-buffered)
    int tud_mark = 1;                   // sched 2: that piece marks its CUs like the chain's own products (yield role 2)
    int leaf_pivot = 2;                 // 2: register-resident leaf (panel.hip leaf_potrf_reg_kernel); 1: the LDS leaf with its 16 x 16 pivot blocks on
                                        // the matrix cores (pivot_block_mfma); 0: the LDS leaf, pivot blocks lane per row
    int leaf_first = 0;                 // 1: TU_b(p) is launched only after D(p+1)'s stage-in kernel, so that the first leaf is
                                        // dispatched BEFORE the update's first wave takes every workgroup slot (a leaf dispatched
                                        // into that wave waited ~140 us for it): 12.08 -> 11.72 ms per N = 8192 fit, two fit
                                        // streams 102.6 -> 105 fits/s.  k > 1: after the (k-1)-th chain kernel (measured worse)
    int yield = 1;                      // cooperative yield: bulk GEMM workgroups sleep while a workgroup of the diagonal-panel
                                        // chain is resident on their CU (csrc/gemm_tile.h); 0 = off
    long long* gemm_trace = nullptr;    // option "gemm_trace" = v > 0: the bulk 128-tile launches stamp their workgroups' phases and the shader
    long gemm_trace_cap = 0;            // clock (GemmArgs::trace), one behind the other, until v * 1024 workgroups are recorded --
    long gemm_trace_pos = 0;            // pgp_test_read_gemm_trace reads and rewinds; 0 frees the buffer
    unsigned* yield_flags = nullptr;    // the device's per-CU table (shared by every context on the device)
    int chain_now = 0;                  // set by the sweep while it queues the chain's kernels (factor_panel)
    int ep_fused = 2;                   // EP parameter recomputation: 0 blocked multi-rhs solve, 1 through the fused inverse (V' = K diag(sW)
                                        // L^-T as one product), 2 K diag(sW) as dense right-hand-side rows of the sweep
    int ep_r_direct = 1;                // EP gradient: sW sW' o B^-1 = S - S Sigma S from the rebuilt Sigma; 0 = triangular inverse + W'W
    int ep_alpha_direct = 1;            // EP: alpha = tnu - ttau o mu (identity, no solve); 0 = the reference's two triangular solves
    int ep_recompute = 0;               // EP: 1 = rebuild Sigma, mu, L after EVERY sweep like the reference (inf.py:772); 0 = carry them by exact
                                        // identities and rebuild once, from the converged site parameters
    int ep_final_rebuild = 0;           // EP, carried posterior: 1 = _epComputeParams once more on the converged site parameters and everything
                                        // returned comes from it (round 3); 0 = alpha / nlZ / gradients from the carried Sigma, mu, log det B
                                        // and ONE plain Cholesky for post.L
    int ep_sigma_under = 1;             // EP: Sigma = K - V'V'^T accumulated under the sweep of the parameter recomputation (ep_fused 2)
    int ep_wait_kernel = 1;             // EP block sweep: the bulk stream waits for the chain in a one-wave kernel of its own (1) or inside
                                        // every workgroup of U = strip W (0: GemmArgs::wait_flag; 128 spinning workgroups cost 2.5 % of a fit)
    int ep_sym = 1;                     // EP: Sigma kept current in its lower triangle only (folds and K - V'V on the lower tiles)
    int ep_block = 1;                   // EP site sweep: 1 = one chain launch per 128 sites + Woodbury fold beside the next chain (round 3),
                                        // 0 = the reference's arithmetic literally: Sigma updated per site
    int fused_value_max_np = 12288;     // value-only fits (want < 3) take the fused-inverse sweep up to this padded size; < 0: always
    int fused_inverse = 1;              // 1: L^-T falls out of the Cholesky sweep (appended identity rows); 0: recursive trtri
    hipDeviceProp_t prop;
    // pooled device buffers, keyed by byte size
    std::mutex pool_mu;                 // pgp_factor_free may run on another thread (Python finalizers / cyclic GC)
    std::multimap<size_t, void*> pool;  // factor buffers only: strict-upper tiles and spare rhs rows are zero by contract
    size_t pool_bytes = 0;
    std::multimap<size_t, void*> spool; // general scratch (arbitrary contents)
    size_t spool_bytes = 0;
    // data
    long n = 0, d = 0, np = 0, ldf = 0;
    int dpad = 0;
    double *x_dev = nullptr, *y_dev = nullptr, *XsT = nullptr, *scale_dev = nullptr;
    // fit workspace (sized for np)
    long ws_np = 0;
    double *W = nullptr, *T = nullptr, *Binv = nullptr, *inv16 = nullptr, *alpha_dev = nullptr, *m_dev = nullptr,
           *rvec = nullptr, *zvec = nullptr, *partial = nullptr, *scal = nullptr;
    long partial_cap = 0;
    double* prep = nullptr;             // [coordinate means | squared norms of the centred points] of the current XsT (hadamard_prep_count(np))
    int pair_launch = 1;                // option "pair_launch": TU_b(p) and panel p's share of E E' go out as ONE launch (gemm_f64_pair_kernel, the same
                                        // tile code behind a second entry point): +0.8 ... 2.4 % on two fit streams at N = 8192, -1.2 % time at N = 16384,
                                        // neutral for a lone N = 8192 chain; bit-identical results (EXPERIMENTS.md)
    int ard_grad_form = 0;              // option "ard_grad_form": 0 = by the norm bound, 1 = always the Gram-form weights, 2 = always the difference form
    std::vector<double> xdev2;          // per coordinate: max_p (x_pk - mean_k)^2 of the resident x (host, pgp_set_data)
    int gram_assembly = 1;              // RBF / RBFard assembly of a fit in the Gram form on the matrix cores: 1 when the host's bound on
                                        // the centred, scaled points' squared norms allows it (csrc/assemble.hip), 0 never, 2 always
    bool dense_ready = false;           // the workspace holds the Q of a dense fit with want = 3 (pgp_dense_grad_term sums against it);
    long dense_n = 0;                   // cleared by every other entry point that rewrites B^-1 / alpha
    int* info_dev = nullptr;
    // results of one fit, gathered on the device and fetched with ONE copy into pinned host memory (three separate copies
    // into pageable memory cost ~270 us per N = 8192 fit: each is staged and synchronised by the runtime)
    double* res_dev = nullptr;          // [info | 8 scalars + ncov + 1 gradient sums | alpha (np)]
    double* res_host = nullptr;         // pinned
    double* res_host_dev = nullptr;     // the same pinned buffer in the device's address space (option "publish": the last kernel of a fit
                                        // writes the results straight into it; no copy command behind the fit)
    int ep_merge12 = 1;                 // option "ep_merge12": EP's first two (unconditional) sweeps queued back to back, one host round trip for both
    int trsm_lean = 0;                  // option "trsm_lean": 1 = the LDS-free panel solve for the diagonal-block chain beside bulk work, 0 never, 2 always
    int publish = 1;                    // option "publish" 1 / 0
    double* in_host = nullptr;          // pinned staging of the per-fit inputs (prior mean, scales)
    double* pred_host = nullptr;        // pinned staging of pgp_predict's inputs and outputs (grow-only: predict.hip pred_stage)
    size_t pred_cap = 0;
    size_t res_cap = 0, in_cap = 0;
    // Cholesky sweep: diagonal-panel scratch (2w x w, w <= 1024), its leaf operand images, column staging buffer
    double *Dk = nullptr, *dpack = nullptr, *Xs = nullptr;
    int xcd_max_k = 512;                // xcd_order applies up to this k extent ...
    int xcd_min_tiles = 256;            // ... to launches with at least this many 128-tiles
    int xcd_super = 8;                  // xcd_order: super-tile edge in tiles
    int solve_outer = 8;                // leaves per outer panel of the blocked multi-rhs triangular solve (K = 128 solve_outer)
    int keep_inverse = 1;               // a fit with the fused inverse rows hands E = L^-T to its posterior handle (np <= 16384): predict's W without a trtri
    int predict_inverse = 1;            // pgp_predict: 0 = blocked solve, 1 = product form with W = L^-1 for batches >= 1024 points (or once W exists), 2 = always
    int predict_batch = 65536;          // test points per batch of pgp_predict (scratch: np x batch doubles, capped at 16 GiB -- predict_batch_points)
    int s_tile = 0;                     // tile size of the panel solves: 0 = automatic
    int gram_fast = 2, gram_grid = 32768; // Gram-form assembly: restructured kernel on / off, persistent workgroups (CovSpec)
    int asm_grid = 4096, asm_nt = 0;    // assembly kernels: persistent workgroups per launch; non-temporal stores (CovSpec::asm_grid / asm_nt)
    size_t Xs_bytes = 0;
    hipEvent_t ev[PGP_NSTAGE + 2];
    double last_ms[PGP_NSTAGE];
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    double pc_ms[PC_COUNT], pc_flops[PC_COUNT], pc_bytes[PC_COUNT];
    int64_t pc_launch[PC_COUNT];
    // cached tile-order tables (device), keyed by (mt, nt, tri, tri_off_tiles)
    std::map<std::vector<int>, std::pair<int*, int>> orders;
    // GEMM variant bits: 64 LDS-DMA operand staging, 256 lazy C (fetched during the k-loop into the registers the DMA
    // frees), 512 16-byte epilogue stores -- the measured-best set; clearing a bit selects the plain form (tests)
    int gemm_dbg = 64 | 256 | 512;
    int xcd_order = 0;    // 1: XCD-aware super-tile order for bulk launches (see gemm_prof): -40 % FETCH per launch, 1-3 % slower
    // options
    int nb_outer = 0;     // leaves (128 columns each) per outer panel -> trailing update K = 128 nb_outer; 0 = automatic (4, or 8 from N = 12288)
    int trtri_small_tile_below = 2049;   // measured: 64x64 tiles win on every recursion level at N=8192 (more, shorter tiles)
    int small_tile_below = 200;   // use 64x64 tiles when a GEMM has fewer 128-tiles than this
};

#define CHK(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != PGP_OK) return rc__; \
    } while (0)

// Every live context, so that an out-of-memory allocation can drop the idle pools of ALL contexts on the device (two fit
// streams per GPU = two contexts, each with its own pools) and so that the idle caps are per device, not per context.
int pgp_ctx_count_on_device(int device);
void pgp_drop_idle_pools(int device);

// PGP_POOL_TRACE=1: every hipMalloc / hipFree the pools fall through to, with its size and duration (stderr)
static inline bool pool_trace_on() { static const bool on = getenv("PGP_POOL_TRACE") != nullptr; return on; }
static inline hipError_t pool_hip_malloc(void** out, size_t bytes, const char* who) {
    if (!pool_trace_on()) return hipMalloc(out, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMalloc(out, bytes);
    fprintf(stderr, "[pool] hipMalloc %8.1f MiB  %8.3f ms  (%s)%s\n", bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), who, e == hipSuccess ? "" : " FAILED");
    return e;
}
static inline void pool_hip_free(void* p, size_t bytes, const char* who) {
    if (!pool_trace_on()) { (void)hipFree(p); return; }
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipFree(p);
    fprintf(stderr, "[pool] hipFree   %8.1f MiB  %8.3f ms  (%s)\n", bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), who);
}

static inline size_t pool_idle_cap(pgp_ctx* c) {
    return (size_t)c->prop.totalGlobalMem / 3 / (size_t)std::max(1, pgp_ctx_count_on_device(c->device));
}

// factor buffers: strict-upper tiles and spare rhs rows are zero by contract; *fresh = true when the caller must zero it
static inline int pool_alloc(pgp_ctx* c, size_t bytes, void** out, bool* fresh = nullptr) {
    if (fresh) *fresh = false;
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        auto it = c->pool.find(bytes);
        if (it != c->pool.end()) {
            *out = it->second;
            c->pool.erase(it);
            c->pool_bytes -= bytes;
            return PGP_OK;
        }
    }
    if (fresh) *fresh = true;
    if (pool_hip_malloc(out, bytes, "factor pool") != hipSuccess) {           // out of memory: drop every idle pool on this device, retry once
        (void)hipGetLastError();
        pgp_drop_idle_pools(c->device);
        HIP_TRY(pool_hip_malloc(out, bytes, "factor pool, retry"));
    }
    return PGP_OK;
}
// idle buffers are kept for the next call with the same shape, up to a third of the device memory over all contexts
static inline void pool_free(pgp_ctx* c, size_t bytes, void* p) {
    if (!p) return;
    const size_t cap = pool_idle_cap(c);
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (c->pool_bytes + bytes > cap) { pool_hip_free(p, bytes, "factor pool over its cap"); return; }
    c->pool.insert({bytes, p});
    c->pool_bytes += bytes;
}

// Small device buffers owned by a posterior handle (alpha, scaled coordinates, leaf inverses): through the scratch pool, not
// hipMalloc / hipFree -- hipFree synchronises the whole device, so releasing a posterior on one fit stream stalled the
// other (API restart search on two streams: 88 instead of 100 fits/s).  No content contract: the owner writes every byte.
static inline int spool_take(pgp_ctx* c, size_t bytes, void** out) {
    {
        std::lock_guard<std::mutex> lk(c->pool_mu);
        auto it = c->spool.find(bytes);
        if (it != c->spool.end()) {
            *out = it->second;
            c->spool.erase(it);
            c->spool_bytes -= bytes;
            return PGP_OK;
        }
    }
    hipError_t e = pool_hip_malloc(out, bytes ? bytes : 8, "handle buffer");
    if (e != hipSuccess) { pgp_set_last_hip_error(e, "hipMalloc(handle buffer)", __FILE__, __LINE__); return PGP_ERR_HIP; }
    return PGP_OK;
}
static inline void spool_give(pgp_ctx* c, size_t bytes, void* p) {
    if (!p) return;
    if (!c) { (void)hipFree(p); return; }
    const size_t cap = pool_idle_cap(c);
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (c->spool_bytes + bytes > cap) { pool_hip_free(p, bytes, "scratch pool over its cap (handle buffer)"); return; }
    c->spool.insert({bytes, p});
    c->spool_bytes += bytes;
}

struct ProfScope {
    pgp_ctx* c; int cls; double flops, bytes; hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t s; int shadow;
    ProfScope(pgp_ctx* c_, int cls_, double f, double b, hipStream_t s_ = nullptr, int shadow_ = -1)
        : c(c_), cls(cls_), flops(f), bytes(b), s(s_ ? s_ : c_->st), shadow(shadow_) {
        if (!c->prof) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        e0 = get(); e1 = get();
        (void)hipEventRecord(e0, s);
    }
    ~ProfScope() {
        if (!c->prof) return;
        (void)hipEventRecord(e1, s);
        c->recs.push_back({cls, e0, e1, flops, bytes, shadow});
    }
};


// Owns hipMalloc'ed scratch for the duration of one API call: every early `return rc` frees it.
struct DevScratch {
    std::vector<void*> ptrs;
    ~DevScratch() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    template <typename T>
    int alloc(T** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) { pgp_set_last_hip_error(e, "hipMalloc(scratch)", __FILE__, __LINE__); return PGP_ERR_HIP; }
        ptrs.push_back(p);
        *out = (T*)p;
        return PGP_OK;
    }
};

// Scratch from the context's size-keyed pool: an optimiser calls with identical shapes hundreds of times, so after the
// first call no hipMalloc / hipFree (both device-synchronising) remain on the path.  Buffers go back on scope exit.
struct PoolScratch {
    pgp_ctx* c;
    std::vector<std::pair<size_t, void*>> held;
    explicit PoolScratch(pgp_ctx* c_) : c(c_) {}
    ~PoolScratch() {
        const size_t cap = pool_idle_cap(c);
        std::lock_guard<std::mutex> lk(c->pool_mu);
        for (auto& h : held) {
            if (c->spool_bytes + h.first > cap) { pool_hip_free(h.second, h.first, "scratch pool over its cap"); continue; }
            c->spool.insert({h.first, h.second});
            c->spool_bytes += h.first;
        }
    }
    template <typename T>
    int alloc(T** out, size_t bytes) {
        void* p = nullptr;
        if (!bytes) bytes = 8;
        {
            std::lock_guard<std::mutex> lk(c->pool_mu);
            auto it = c->spool.find(bytes);
            if (it != c->spool.end()) { p = it->second; c->spool.erase(it); c->spool_bytes -= bytes; }
        }
        if (!p && pool_hip_malloc(&p, bytes, "scratch pool") != hipSuccess) {       // out of memory: drop every idle pool on this device, retry once
            (void)hipGetLastError();
            pgp_drop_idle_pools(c->device);
            hipError_t e = pool_hip_malloc(&p, bytes, "scratch pool, retry");
            if (e != hipSuccess) { pgp_set_last_hip_error(e, "hipMalloc(pool scratch)", __FILE__, __LINE__); return PGP_ERR_HIP; }
        }
        held.push_back({bytes, p});
        *out = (T*)p;
        return PGP_OK;
    }
    // hand a buffer on to another owner (a posterior handle): it no longer goes back to the pool with this scope; returns its size
    size_t release(void* p) {
        for (size_t i = 0; i < held.size(); ++i)
            if (held[i].second == p) { const size_t b = held[i].first; held.erase(held.begin() + (long)i); return b; }
        return 0;
    }
};

// Returns a pooled factor buffer to the context unless ownership was handed on (release()).  The pool's contract is
// "strict-upper tiles and spare rhs rows are zero": with scrub set, a buffer abandoned half-way (non-PD pivot -> NaNs,
// HIP failure) is zeroed before it goes back.
struct FactorGuard {
    pgp_ctx* c; double* F; size_t bytes; bool scrub;
    FactorGuard(pgp_ctx* c_, double* F_, size_t b, bool scrub_ = false) : c(c_), F(F_), bytes(b), scrub(scrub_) {}
    ~FactorGuard() {
        if (!F) return;
        if (scrub) { (void)hipMemsetAsync(F, 0, bytes, c->st); (void)hipStreamSynchronize(c->st); }
        pool_free(c, bytes, F);
    }
    double* release() { double* p = F; F = nullptr; return p; }
};

// A factor handle under construction: freed (with everything it owns) on an early return.
struct FactorHandleGuard {
    pgp_ctx* c; pgp_factor* f;
    FactorHandleGuard(pgp_ctx* c_, pgp_factor* f_) : c(c_), f(f_) {}
    ~FactorHandleGuard() { if (f) pgp_factor_free(c, f); }
    pgp_factor* release() { pgp_factor* p = f; f = nullptr; return p; }
};

void prof_collect(pgp_ctx* c);
static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }
int make_spec(pgp_ctx* c, int kind, const double* hyp, int nhyp, int para, int flags, int der, long d, CovSpec& cs);
int cov_point_value(pgp_ctx* c, const CovSpec& cs, int train, double* out);
int potrf_blocked(pgp_ctx* c, double* F, long ld, long np, long mrows, bool with_inverse = false, double* E = nullptr,
                  long lde = 0);
int gemm_prof(pgp_ctx* c, int cls, GemmArgs g, hipStream_t st = nullptr);
int gemm_prof_pair(pgp_ctx* c, int cls_a, GemmArgs a, int cls_b, GemmArgs b, hipStream_t st = nullptr);
int batch_tile_list(pgp_ctx* c, int mt0, int nt, int nb, int dmt, const int** out, int* n);
// test points per batch: the option, the number of points, and a 16 GiB cap on the np x batch cross-covariance block (never below 16384)
inline long predict_batch_points(int option, long ns, long np) {
    const long cap = std::max<long>(16384, ((long)1 << 34) / (8 * std::max<long>(np, 1)) / 128 * 128);
    const long want = std::min<long>(option, (ns + 127) / 128 * 128);
    return std::max<long>(128, std::min<long>(want, cap));
}
// Per-device gate between the contexts of ONE process.  Every entry point that may reach the HIP runtime holds it SHARED for the
// duration of its call (calls are synchronous: when nobody holds it, none of this process's work is on the device and no thread
// sits in a device-synchronising runtime call).  An EP block sweep holds it EXCLUSIVELY: the
// sweep is a resident kernel on one stream that meets bulk launches on another through device counters, and the runtime may map
// streams of DIFFERENT contexts onto one hardware queue -- a foreign launch that waits for an event of its own context's other
// stream, queued between the sweep's bulk launches while its producer sits behind the resident kernel, closes a cycle (seen as "a
// device-side wait gave up" when a K-fold search ran two EP fits beside each other).  The cycle that was actually caught (round 5,
// tools/ep_kfold_diag.py): the OTHER host thread frees a buffer (pgp_set_data of the next fold, a posterior handle that dies) --
// hipFree waits for the device to drain, i.e. for the resident sweep kernel, and holds the runtime's lock meanwhile, so the sweeping
// thread cannot enqueue the strip launch the resident kernel is waiting for.  Writers are preferred (sweeps are short).
struct DeviceGate {
    std::mutex m;
    std::condition_variable cv;
    int readers = 0, writers_waiting = 0;
    bool writer = false;
    void lock_shared() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !writer && writers_waiting == 0; }); ++readers; }
    void unlock_shared() { std::lock_guard<std::mutex> l(m); if (--readers == 0) cv.notify_all(); }
    void lock() { std::unique_lock<std::mutex> l(m); ++writers_waiting; cv.wait(l, [&] { return !writer && readers == 0; }); --writers_waiting; writer = true; }
    void unlock() { std::lock_guard<std::mutex> l(m); writer = false; cv.notify_all(); }
};
DeviceGate& device_gate(int device);
int& device_gate_depth(int device);      // per thread and device: entry points may nest (a host-transport call-back that runs a Python
                                         // finalizer -> pgp_factor_free inside pgp_sharded_exact_fit): the inner hold is a no-op
struct GateShared {                     // RAII: shared for a whole entry point; exclusive() / shared_again() around an EP sweep
    DeviceGate& g; int& depth; int state = 1;       // 1 shared, 2 exclusive, 0 nested (the outer hold covers it)
    explicit GateShared(pgp_ctx* c) : GateShared(c ? c->device : 0) {}
    explicit GateShared(int device) : g(device_gate(device)), depth(device_gate_depth(device)) {
        if (depth++ > 0) state = 0; else g.lock_shared();
    }
    void exclusive() { if (state == 1) { g.unlock_shared(); g.lock(); state = 2; } }
    void shared_again() { if (state == 2) { g.unlock(); g.lock_shared(); state = 1; } }
    ~GateShared() { --depth; if (state == 1) g.unlock_shared(); else if (state == 2) g.unlock(); }
    GateShared(const GateShared&) = delete;
    GateShared& operator=(const GateShared&) = delete;
};
constexpr double ARD_GRAM_GRAD_BOUND = 1.0e6;     // max squared norm of a scaled, centred point for the Gram-form gradient weights
bool gram_assembly_applies(pgp_ctx* c, const CovSpec& cs);
int diag_block_factor(pgp_ctx* c, const double* src, long lds, int w, double* Fd, long ldf, double* Ed, long lde,
                      int info_base, hipStream_t st, hipEvent_t staged = nullptr);
int trtri_lower(pgp_ctx* c, const double* L, long ldl, double* W, long ldw, double* T, long np);
int lauum_lower(pgp_ctx* c, const double* W, long ldw, double* Binv, long ldb, long np);
int upload_scaled(pgp_ctx* c, const double* x_dev, long n, long d, const std::vector<double>& sc, double* XsT, long ldp,
                  int dpad, double* scale_dev);
int alloc_factor_buffer(pgp_ctx* c, long np, long ldf, double** F);
int ensure_workspace(pgp_ctx* c, long np);
int solve_lower_multi(pgp_ctx* c, const double* L, long ldl, const double* Wd, double* Y, long ldy, long np, int nrhs,
                      bool trans);
int potrf_blocked_rhs(pgp_ctx* c, double* F, long ld, long np, long mrows, double* R, long ldr, long nrhs2);
int eet_lower(pgp_ctx* c, const double* E, long lde, double* Binv, long ldb, long np);
