// Expectation propagation for the probit likelihood on the device -- reference: Core/inf.py EP.evaluate
// :731-806, Inference._epComputeParams :174-189, Core/lik.py Erf (EP mode) :295-366.
//
// The site loop is inherently sequential (site i+1 needs Sigma_ii, mu_i after site i's update, fixed order 0..n-1,
// inf.py:757).  Two forms:
//   ep_block 0   the reference's arithmetic literally, two launches per site: ep_site_kernel (cavity, probit moments, new
//                (ttau_i, tnu_i), rank-1 coefficient) and ep_rank1_mu_kernel (Sigma -= c s_i s_i' fused with mu = Sigma tnu:
//                16 N^2 bytes per site).  Kept as the parity anchor of the variants test.
//   ep_block 1   (default) the BLOCK sweep: the 128 sites of a block only read Sigma_BB and mu_B, so one workgroup runs them
//                with Sigma_BB in registers (ep_chain_kernel, one launch per SWEEP) and the rest of Sigma gets the block's effect
//                all at once by the matrix inversion lemma, folded in beside the next block's chain -- see "block sweep" below.
// The posterior -- Sigma, mu, log det B -- is CARRIED through the sweeps (every step of the block sweep is an exact identity; the
// determinant lemma per site) and rebuilt from scratch once, from the converged site parameters, with the SAME kernels as
// exact inference; option ep_recompute 1 rebuilds it after every sweep like the reference (inf.py:772):
//   B = I + sW sW' o K (fused build) -> blocked MFMA Cholesky with K diag(sW) riding along as right-hand-side rows
//   (V' = K sW L^-T) -> Sigma = K - V'V'^T accumulated panel by panel under that sweep -> mu = Sigma tnu.
// alpha and sW sW' o B^-1 follow from Sigma and mu by identities (no further solve); the gradients reuse the Hadamard-reduce
// kernel of the exact fit.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "ctx.h"
#include "erf_lik.h"
#include "erfcx_poly.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double* red /* 4 doubles */) {      // 256 threads, fixed order
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ep_site_kernel(const double* __restrict__ Sig, long ld, long np, long i,
                                                      const double* __restrict__ mu, const double* __restrict__ m,
                                                      const double* __restrict__ y, double* __restrict__ ttau,
                                                      double* __restrict__ tnu, double* __restrict__ sbuf,
                                                      double* __restrict__ coef) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r < np) sbuf[r] = Sig[r + i * ld];
    if (r == 0) {
        const double sii = Sig[i + i * ld];
        const double tau_ni = 1.0 / sii - ttau[i];                         // cavity (inf.py:759-760)
        const double nu_ni = mu[i] / sii + m[i] * tau_ni - tnu[i];
        double lZ, dlZ, d2lZ;
        erf_ep_moments(y[i], nu_ni / tau_ni, 1.0 / tau_ni, &lZ, &dlZ, &d2lZ);
        const double ttau_old = ttau[i];
        double t_new = -d2lZ / (1.0 + d2lZ / tau_ni);
        t_new = fmax(t_new, 0.0);                                          // inf.py:765
        const double nu_new = (dlZ + (m[i] - nu_ni / tau_ni) * d2lZ) / (1.0 + d2lZ / tau_ni);
        ttau[i] = t_new;
        tnu[i] = nu_new;
        const double ds2 = t_new - ttau_old;
        coef[0] = ds2 / (1.0 + ds2 * sii);                                 // inf.py:769
    }
}

// one wave per row: Sigma[r,:] -= coef * s_r * s ; mu_r = Sigma_new[r,:] . tnu
__global__ __launch_bounds__(256) void ep_rank1_mu_kernel(double* __restrict__ Sig, long ld, long np,
                                                          const double* __restrict__ sbuf,
                                                          const double* __restrict__ coef,
                                                          const double* __restrict__ tnu, double* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= np) return;
    const double cs = coef[0] * sbuf[r];
    double* row = Sig + r * ld;                                            // symmetric: row r == column r
    double acc = 0.0;
    for (long c = 2 * lane; c < np; c += 128) {
        double2_t v = *(double2_t*)(row + c);
        const double2_t s = *(const double2_t*)(sbuf + c);
        const double2_t t = *(const double2_t*)(tnu + c);
        v[0] = fma(-cs, s[0], v[0]);
        v[1] = fma(-cs, s[1], v[1]);
        *(double2_t*)(row + c) = v;
        acc = fma(v[0], t[0], acc);
        acc = fma(v[1], t[1], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) mu[r] = acc;
}

constexpr int EPB = 128;                       // sites per block of the sweep

// Sigma_blk is kept current in its LOWER triangle only (the folds and Sigma = K - V'V are symmetric rank-k updates: half
// the tiles); element (r, c) of the symmetric matrix:
__device__ __forceinline__ double sym_at(const double* __restrict__ Sig, long ld, long r, long c) {
    return r >= c ? Sig[r + c * ld] : Sig[c + r * ld];
}

// upper triangle <- transpose of the lower one, 64 x 64 tiles through LDS (blockIdx = (tile row, tile column), row >= column)
__global__ __launch_bounds__(256) void ep_mirror_kernel(double* __restrict__ A, long ld) {
    if (blockIdx.x < blockIdx.y) return;
    __shared__ double tile[64][65];
    const long r0 = 64L * blockIdx.x, c0 = 64L * blockIdx.y;
    const int a = threadIdx.x & 63, b = threadIdx.x >> 6;
    const bool diag = blockIdx.x == blockIdx.y;      // diagonal tiles: only the entries above the diagonal are rewritten
    for (int q = b; q < 64; q += 4) tile[q][a] = A[r0 + a + (c0 + q) * ld];
    __syncthreads();
    for (int q = b; q < 64; q += 4)
        if (!diag || a < q) A[c0 + a + (r0 + q) * ld] = tile[a][q];
}

// Latency-trimmed site update for the replay chain (4096 sequentially dependent evaluations per sweep: every cycle of
// this function is on the critical path).  Same formulas as Core/inf.py:759-769 + lik.py:295-311, but: reciprocals by
// v_rcp_f64 + one Newton step instead of IEEE division sequences (10 divisions per site), 1/sqrt by v_rsq_f64 + Newton, and
// for z > -5 (the branch without asymptotics) N(z)/Phi(z) straight from Phi -- the reference's exp(log Phi) round trip and
// log Phi itself are not needed for the derivatives.  Differences to the scalar path: a few ulp (parity tests: 1e-8).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}
// erfcx(u) = exp(u^2) erfc(u) for 0 <= u <= 6: ONE polynomial of degree 20 in the mapped variable y = 4 / (4 + u)
// (csrc/erfcx_poly.h, generated with mpmath; 1.2e-15 relative) in Estrin form -- depth ~12 from u including the reciprocal,
// branch-free (piecewise versions were built: six degree-16 pieces in u get if-converted into all six evaluations, and a
// scalar switch on the piece stops the unroller of the site recurrence), against the library's erf + exp + division
__device__ __forceinline__ double erfcx_0_6(double u) {
    using namespace erfcx_poly;
    const double y = C0 * fast_rcp(C0 + u);
    const double t = (y - MID) * INV_HALF;
    const double t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
    double p[NC / 2];
#pragma unroll
    for (int i = 0; i < NC / 2; ++i) p[i] = fma(COEF[2 * i + 1], t, COEF[2 * i]);
    double q[NC / 4];
#pragma unroll
    for (int i = 0; i < NC / 4; ++i) q[i] = fma(p[2 * i + 1], t2, p[2 * i]);
    const double r0 = fma(q[1], t4, q[0]), r1 = fma(q[3], t4, q[2]), r2 = fma(q[5], t4, q[4]);
    return fma(r2, t16, fma(r1, t8, r0));
}
// exp(x) for -800 <= x <= 0: x = n ln 2 + r (Cody-Waite in two pieces), Taylor polynomial of degree 13 on |r| <= ln 2 / 2
// (remainder < 4e-18), v_ldexp_f64.  ~20 instructions against ~40 of the library's exp: every instruction of the site update
// is on the chain of 4096 dependent updates.
__device__ __forceinline__ double exp_neg(double x) {
    const double n = rint(x * 1.44269504088896340736);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
// N(z) / Phi(z) for z > -5 (the branch of lik.py:340-343 without asymptotics).  z <= 0: Phi = exp(-z^2/2) erfcx(-z/sqrt 2) / 2,
// so the ratio is sqrt(2/pi) / erfcx(-z/sqrt 2) -- no exponential at all; z > 0: Phi = 1 - exp(-u^2) erfcx(u) / 2, u = z/sqrt 2
// (beyond u = 6 exp(-u^2) < 2.4e-16 makes Phi = 1 whatever the clamped polynomial returns).  The callers evaluate it on
// wave-uniform arguments, so the sign test is a real branch: z <= 0 never pays for the exponential.
__device__ __forceinline__ void probit_hazard_frac(double z, double& P, double& Q) {      // N(z) / Phi(z) = P / Q
    const double u = fabs(z) * 0.70710678118654752440;
    const double gx = erfcx_0_6(fmin(u, erfcx_poly::UMAX));
    if (z <= 0.0) { P = 0.79788456080286535588; Q = gx; return; }
    const double e = exp_neg(fmax(-u * u, -800.0));
    P = e * 0.39894228040143267794;
    Q = fma(-0.5 * e, gx, 1.0);
}
__device__ __forceinline__ double probit_hazard(double z) {
    double P, Q;
    probit_hazard_frac(z, P, Q);
    return P * fast_rcp(Q);
}

// One site update (inf.py:759-769 + lik.py:295-311) on the chain of 4096 dependent updates.  The likelihood ratio
// n_p = N(z) / Phi(z) stays a FRACTION P / Q all the way to the rank-1 coefficient, so the chain from (Sigma_ii, mu_i) to c_j
// has three reciprocals (cavity, the polynomial's argument map, c_j) and one reciprocal square root instead of six:
//   a = -d2lZ = n_p (z + n_p) / (1 + s2) = A / Q^2,  A = P (z Q + P) rden^2        den = 1 + d2lZ s2 = D / Q^2,  D = Q^2 - A s2
//   ttau_new = a / den = A / D            ds2 = (A - tp D) / D            c_j = ds2 / (1 + ds2 Sigma_ii) = N / (D + N Sigma_ii),  N = A - tp D
//   tnu_new = (dlZ + (m - mu_c) d2lZ) / den = (ys P Q rden - (m - mu_c) A) / D   (its reciprocal runs beside c_j's)
// ttau_new < 0 (inf.py:765 clamps it to 0) <=> A D < 0: then ds2 = -tp and c_j = -tp / (1 - tp Sigma_ii) = -tp r1.
struct EpSiteMid { double r1, s2, mu_c, ys, rden, z, P, Q; };
__device__ __forceinline__ void ep_site_update_a(double sii, double mui, double tp, double np_, double mi, double yi, EpSiteMid& h) {
    // cavity (inf.py:759-760) with ONE reciprocal: tau_ni = 1/sii - tp = d1/sii, d1 = 1 - tp sii
    const double d1 = fma(-tp, sii, 1.0);
    h.r1 = fast_rcp(d1);
    h.s2 = sii * h.r1;                                                         // 1 / tau_ni
    h.mu_c = fma(mi, d1, fma(-np_, sii, mui)) * h.r1;                          // nu_ni / tau_ni
    h.ys = (yi < 0.0) ? -1.0 : 1.0;
    h.rden = fast_rsqrt(1.0 + h.s2);
    h.z = h.ys * h.mu_c * h.rden;
    if (h.z > -5.0) probit_hazard_frac(h.z, h.P, h.Q);                        // lik.py:340-343 (naive ratio)
    else { h.P = erf_ratio(h.z, exp(erf_logphi(h.z))); h.Q = 1.0; }
}
__device__ __forceinline__ void ep_site_update_b(double sii, double mui, double tp, double np_, double mi, const EpSiteMid& h,
                                                 double& t_new, double& nu_new, double& cj, double& qj) {
    const double r1 = h.r1, s2 = h.s2, mu_c = h.mu_c, ys = h.ys, rden = h.rden, z = h.z, P = h.P, Q = h.Q;
    const double A = P * fma(z, Q, P) * (rden * rden);
    const double Q2 = Q * Q;
    const double D = fma(-A, s2, Q2);
    const double N = fma(-tp, D, A);
    const double rc = fast_rcp(fma(N, sii, D));
    const double wq = fast_rcp(D);
    const bool clamp = A * D < 0.0;                                            // inf.py:764-765
    t_new = clamp ? 0.0 : A * wq;
    cj = clamp ? -tp * r1 : N * rc;
    nu_new = fma(ys * P * rden, Q, -(mi - mu_c) * A) * wq;
    const double dnu = nu_new - np_;
    qj = dnu - cj * fma(dnu, sii, mui);
}
__device__ __forceinline__ void ep_site_update(double sii, double mui, double tp, double np_, double mi, double yi,
                                               double& t_new, double& nu_new, double& cj, double& qj) {
    EpSiteMid h;
    ep_site_update_a(sii, mui, tp, np_, mi, yi, h);
    ep_site_update_b(sii, mui, tp, np_, mi, h, t_new, nu_new, cj, qj);
}

// ---- block sweep (round 3) ---------------------------------------------------------------------------------
// The sites of one block B of 128 consecutive sites only ever read Sigma_BB and mu_B: the sequential site loop of the block is
// EP on a 128-point problem, run by ONE workgroup with Sigma_BB in registers (ep_chain_kernel, 128 dependent site updates in
// one launch, no launch boundary and no global memory between two sites).  What the block did to the rest of Sigma follows
// from the matrix inversion lemma, exactly and all at once: with dT = diag(ttau_new - ttau_old) on B,
//     Sigma_new = (Sigma^-1 + E_B dT E_B')^-1 = Sigma - Sigma(:,B) W Sigma(B,:),   W = (dT^-1 + Sigma_BB)^-1 = dT - dT Sigma_BB,new dT
// (Sigma_BB,new is what the chain ends with; no inverse, and dT = 0 rows are fine), and for mu = Sigma tnu with
// h = dnu - dT o mu_B,old:                 mu_new = mu + Sigma(:,B) g,    g = h - dT o (Sigma_BB,new h).
// Same sites in the same order with the same scalar update as inf.py:757-770; only the order in which the rank-1 terms are
// summed differs.  Two streams: the chain stream runs ONE resident kernel per sweep -- one workgroup the chain, 36 workgroups
// prep (they bring the NEXT diagonal block and its mu up to date: one 128 x 128 tile) -- the bulk stream copies the strip
// Sigma(:,B), forms U = strip W and folds U strip' into all of Sigma in one launch, beside the next block's chain.  The two
// meet through device counters, not events (ep_chain_kernel).
// The waves are SPECIALISED.  A first version in which every wave did everything (site update evaluated redundantly, 8 x 8
// entries of Sigma_BB per thread, one barrier per site) took 2850 s_memtime ticks per site, of which the scalar site update
// is ~1050 and everything around it ~1800: a lone wave issues one fp64 instruction per 5.4 ticks, and LDS is bandwidth-bound
// per LANE (a ds_read_b64 costs the 4 waves 17 ticks whether the lanes share an address or not: tools/wave_costs.py) -- the 16
// broadcast reads, the 64 + 16 FMAs of the rank-1 share and the column hand-over all sat in series with the site update.
// Here wave 0 does nothing but the site updates, back to back: at the end of
// site k it needs only three numbers -- Sigma(k+1,k), Sigma(k+1,k+1) and mu(k+1) as they were BEFORE site k -- which the three
// update waves (Sigma_BB in their registers, 48 row slots each) published one site earlier, and applies site k's own rank-1
// term to them itself.  The update waves run one site behind: they wait for (c_k, q_k), pass column k+1 / diagonal / mu on
// (double-buffered by the parity of k), then do their share of the rank-1 update while wave 0 is already inside site k+1.
// Hand-over through LDS sequence counters, no workgroup barrier inside the loop.
constexpr int EPCP = 144;
template <int V> using IntC = std::integral_constant<int, V>;
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(IntC<Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ int lds_seq(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct EpChainLds {
    __attribute__((aligned(16))) double colb[8][EPCP];     // column k of Sigma_BB before site k: ring over k mod 8
    double diagb[2][EPB], mub[2][EPB];                     // diagonal and mu before site k, by parity of k
    __attribute__((aligned(16))) double cq[16][2];         // (c_k, q_k): ring over k mod 16
    double gpart[2][EPCP];                                 // epilogue: the two column halves of Sigma_BB,new h
    __attribute__((aligned(32))) double prm[EPB][4];       // the sites' (ttau, tnu) of the previous sweep, m, y
    double s_mu0[EPB], s_dt[EPB], s_dn[EPB], s_tn[EPB], s_nn[EPB];
    int seqC, seqP, seqG;             // sites wave 0 has finished ; 3 x sites the update waves have passed on ; 6 x groups read
};
struct EpPrepLds {
    double Xr[16][EPB + 1], Xc[16][EPB + 1], T[16][EPB + 1], red[4][16][17], gl[EPB];
};
constexpr size_t EP_BLOCK_LDS = sizeof(EpPrepLds) > sizeof(EpChainLds) ? sizeof(EpPrepLds) : sizeof(EpChainLds);

// ---- hand-overs of the resident sweep kernel ----------------------------------------------------------------------------
// Device counters that only grow within a fit (flags[EPF_*], see ep_chain_kernel) and data read / written at AGENT scope: the
// sc1 forms of global_load / global_store are coherent across the XCDs' L2s by themselves.  A hand-over is then "stores, the
// s_waitcnt of a barrier, counter" on one side and "counter, loads" on the other -- no release / acquire fence: those write back
// and invalidate a WHOLE L2, which the bulk stream's folds keep full of dirty tiles (measured: 21 us per block with fences, 15
// without; a launch boundary costs the same flush plus the dispatch).
enum { EPF_CHAIN = 0, EPF_PREP = 1, EPF_STRIP = 2, EPF_ERR = 3 };
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ld_dev with a wave-uniform base (SGPR pair), a 32-bit per-lane byte offset and an immediate: the 48 tile loads of an update wave
// then cost two address registers, not 96 (as compiler-generated atomic loads they pushed the accumulators into scratch).  The
// compiler does not count asm loads: ep_loads_done() waits for them and is the only place their results become usable.
template <int OFF>
__device__ __forceinline__ void ld_dev_issue(double& x, const double* sbase, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 sc1" : "=v"(x) : "v"(voff), "s"(sbase), "n"(OFF));
}
__device__ __forceinline__ void ep_loads_done(double (&v)[12]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]));
}
// wait until the counter *f has reached `target`.  Bounded (~1 s, at once when another wait has already given up): a launch that
// never came must not hang the device; the fit then fails through flags[EPF_ERR]
// *err keeps the FIRST site that gave up (diagnostics): site | block << 8 | 1 << 31
__device__ __forceinline__ void ep_wait_ge(unsigned* f, unsigned target, unsigned* err, unsigned site = 0u) {
    for (unsigned it = 0; (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0; ++it) {
        if (it > (1u << 22) || ((it & 1023u) == 1023u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            unsigned expected = 0u;
            (void)__hip_atomic_compare_exchange_strong(err, &expected, site | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

// prep(b): the diagonal tile and the mu entries of block b, brought up to date in place from block b-1's (W, g):
//   Sigma(B, B) -= X W X',  mu(B) += X g,   X = strip(B, :) = Sigma(B, B_prev) before the previous block.   36 workgroups: the
// 16 x 16 output tiles on or below the diagonal (the lower triangle is the one kept current); T = X_rows W (16 x 128, K = 128: each
// wave two column tiles, W straight from global memory as the MFMA operand), then T X_cols' with K split over the four waves.
// The result goes to three places: Sigma itself, the rows B of the strip buffer of block b (Snext(r, k) = Sigma(r, r0 + k): the
// bulk stream's strip kernel skips them and has nothing to wait for) and the tile buffer the chain reads.  It sits between two
// blocks of the chain, on the critical path of the sweep: workgroups 1 .. 36 of the sweep kernel, 256 of their 512 threads.
// Index of Sigma_BB(R, C) in the tile buffer: the chain workgroup's accumulator layout (update wave pair u = R / 48, tile row
// I, tile column J = C / 16 of wave `half` = J / 4, component r, lane), so that its 96 loads per lane are contiguous per wave.
constexpr int EP_TILE_N = 3 * 2 * 3 * 4 * 4 * 64;        // (rows 128 .. 143 of the layout are padding, never written or read)
__device__ __forceinline__ int ep_tile_index(int R, int C) {
    const int u = R / 48, I = (R % 48) >> 4, l15 = R & 15, J = C >> 4, half = J >> 2, jj = J & 3, c16 = C & 15, l4 = c16 & 3, r = c16 >> 2;
    return ((((u * 2 + half) * 3 + I) * 4 + jj) * 4 + r) * 64 + l15 + 16 * l4;
}
// copy_only (the first block of a sweep: nothing to fold in): the tile, its diagonal and mu_B as they are.
__device__ __forceinline__ void ep_prep_body(int blk, double* Sig, long ld, long r0, const double* S, double* Snext, const double* W,
                                             const double* g, double* mu, double* Tile, bool copy_only, EpPrepLds& L) {
    int bi = 0, rem = blk;
    while (rem > bi) { rem -= bi + 1; ++bi; }
    const int bj = rem;                                  // bi >= bj
    auto& Xr = L.Xr; auto& Xc = L.Xc; auto& T = L.T; auto& red = L.red; auto& gl = L.gl;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l15 = lane & 15, l4 = lane >> 4;
    const long orow = r0 + 16 * bi + (t >> 4), ocol = r0 + 16 * bj + (t & 15);
    const int R = 16 * bi + (t >> 4), C = 16 * bj + (t & 15);
    if (copy_only) {
        if (R >= C) {
            const double v = ld_dev(Sig + orow + ocol * ld);
            st_dev(Tile + ep_tile_index(R, C), v);
            st_dev(Tile + ep_tile_index(C, R), v);
            if (R == C) st_dev(Tile + EP_TILE_N + EPB + R, v);
        }
        if (bi == bj && t < 16) st_dev(Tile + EP_TILE_N + 16 * bi + t, ld_dev(mu + r0 + 16 * bi + t));
        return;
    }
    // every global read is issued up front (a chain of memory round trips otherwise): the 64 W operands of this lane, its 16
    // strip entries, the output entry it will update
    const double* w0 = W + 16 * (2 * wv) + l15 + (long)EPB * l4;              // W(q, kk) at W[kk + 128 q]
    double wx[EPB / 4][2];
#pragma unroll
    for (int ks = 0; ks < EPB / 4; ++ks) { wx[ks][0] = ld_dev(w0 + (long)EPB * 4 * ks); wx[ks][1] = ld_dev(w0 + (long)EPB * 4 * ks + 16); }
    double xr[8], xc[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int e = t + 256 * v, ii = e & 15, q = e >> 4;
        xr[v] = ld_dev(S + r0 + 16 * bi + ii + (long)q * ld);
        xc[v] = ld_dev(S + r0 + 16 * bj + ii + (long)q * ld);
    }
    const double old = ld_dev(Sig + orow + ocol * ld);
    const double gk = t < EPB ? ld_dev(g + t) : 0.0;
    double muold = 0.0;
    if (bi == bj && t < 16) muold = ld_dev(mu + r0 + 16 * bi + t);
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int e = t + 256 * v, ii = e & 15, q = e >> 4;
        Xr[ii][q] = xr[v]; Xc[ii][q] = xc[v];
    }
    if (t < EPB) gl[t] = gk;
    __syncthreads();
    {
        double4_t acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int ks = 0; ks < EPB / 4; ++ks) {
            const double y = Xr[l15][4 * ks + l4];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wx[ks][0], y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wx[ks][1], y, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[l15][16 * (2 * wv + q) + l4 + 4 * r] = acc[q][r];
    }
    __syncthreads();
    {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int k = 32 * wv + 4 * ks + l4;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xc[l15][k], T[l15][k], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][l15][l4 + 4 * r] = acc[r];
    }
    __syncthreads();
    {
        const int ii = t >> 4, jj = t & 15;
        const double out = ((red[0][ii][jj] + red[1][ii][jj]) + red[2][ii][jj]) + red[3][ii][jj];
        if (orow >= ocol) {
            const double nv = old - out;
            st_dev(Tile + ep_tile_index(R, C), nv);
            st_dev(Tile + ep_tile_index(C, R), nv);
            if (R == C) st_dev(Tile + EP_TILE_N + EPB + R, nv);
            st_dev(Sig + orow + ocol * ld, nv);
            st_dev(Snext + orow + (ocol - r0) * ld, nv);
            st_dev(Snext + ocol + (orow - r0) * ld, nv);
        }
    }
    if (bi == bj && t < 16) {
        double acc = 0.0;
        for (int k = 0; k < EPB; ++k) acc = fma(Xr[t][k], gl[k], acc);
        st_dev(Tile + EP_TILE_N + 16 * bi + t, muold + acc);
        st_dev(mu + r0 + 16 * bi + t, muold + acc);
    }
}

// ONE launch per sweep, resident for all its blocks.  Workgroup 0 = the chain: the sites of block 0, 1, ... back to back (below);
// workgroups 1 .. 36 = prep: the diagonal tile and mu of block b from block b-1's (W, g) as soon as the chain has published them
// (block 0: a plain copy into the tile buffer).  No launch boundary on the sweep's critical path; the hand-overs are counters --
//   flags[EPF_CHAIN]  blocks the chain has finished (W, g of the block are in memory): prep(b+1) and, on the bulk stream,
//                     a one-wave ep_wait_kernel ahead of U(b) = strip W (or U's own workgroups: GemmArgs::wait_flag) wait for it;
//   flags[EPF_PREP]   prep workgroups that have finished: chain(b) waits for 36 more;
//   flags[EPF_STRIP]  workgroups of the bulk stream's strip kernels that have finished: prep(b+1) reads strip(b).
// cbase / pbase / sbase: the counters' values before this sweep's first block (sbase in strip LAUNCHES of swg workgroups each).
// Buffers by block parity: W (128 x 128), g (128), strip (np x 128).  Measured on the way here (cfg 5, same box): prep and chain as
// two launches per block 22.4 ms; prep workgroups inside the chain's launch 21.7; + one fold launch per block 21.2; events
// replaced by these counters but still one launch per block 21.2; resident with release / acquire fences 21.1; with agent-scope
// loads and stores 20.7.  (The body of a block as a non-inlined function: callee-saved registers in scratch, site loop 14 % slower:
// 24.0.  One loop over the blocks AROUND the roles instead of one inside each: 636 bytes of spills.)
template <bool TIMED>                            // TIMED: s_memtime stamps and spin counts of block 5 (PGP_EP_TIMING); costs registers
__global__ __launch_bounds__(512) void ep_chain_kernel(double* Sig, long ld, long n, int nbl, double* mu, const double* __restrict__ m,
                                                        const double* __restrict__ y, double* ttau, double* tnu, double* Wbuf,
                                                        double* gbuf, double* ldbuf, unsigned* yield_flags, long long* stamps_b5,
                                                        double* Sbuf, double* Tile, unsigned* flags, unsigned cbase, unsigned pbase,
                                                        unsigned sbase, unsigned swg) {
    extern __shared__ __attribute__((aligned(32))) double ep_smem[];
    if (blockIdx.x > 0) {
        if (threadIdx.x < 256) {
            for (int b = 0; b < nbl; ++b) {
                if (b > 0 && threadIdx.x == 0) {
                    // (the strip is there long before the chain is through: its poll -- one agent-scope round trip even when satisfied --
                    //  comes first, so that nothing stands between the chain's counter and the loads)
                    ep_wait_ge(flags + EPF_STRIP, (sbase + (unsigned)b) * swg, flags + EPF_ERR, 2u | ((unsigned)b << 8));
                    ep_wait_ge(flags + EPF_CHAIN, cbase + (unsigned)b, flags + EPF_ERR, 1u | ((unsigned)b << 8));
                }
                __syncthreads();
                ep_prep_body((int)blockIdx.x - 1, Sig, ld, (long)b * EPB, Sbuf + (long)((b + 1) & 1) * EPB * ld, Sbuf + (long)(b & 1) * EPB * ld,
                             Wbuf + ((b + 1) & 1) * EPB * EPB, gbuf + ((b + 1) & 1) * EPB, mu, Tile, b == 0, *reinterpret_cast<EpPrepLds*>(ep_smem));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a barrier only waits for LDS: every wave sees its stores through first)
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + EPF_PREP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    EpChainLds& L = *reinterpret_cast<EpChainLds*>(ep_smem);
    auto& colb = L.colb; auto& diagb = L.diagb; auto& mub = L.mub; auto& cq = L.cq; auto& gpart = L.gpart; auto& prm = L.prm;
    auto& s_mu0 = L.s_mu0; auto& s_dt = L.s_dt; auto& s_dn = L.s_dn; auto& s_tn = L.s_tn; auto& s_nn = L.s_nn;
    int& seqC = L.seqC; int& seqP = L.seqP; int& seqG = L.seqG;
    pgp_yield_mark(yield_flags, +1);
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // eight waves, two per SIMD: wave 0 = the chain (alone on SIMD 0: wave 4 only keeps the barriers company); waves 1-3 and
    // 5-7 = the update waves, pair u on SIMD u + 1: both hold the rows 48 u .. 48 u + 47, `half` 0 the tile columns 0-3, 1 the
    // tile columns 4-7.  The wave that owns the current tile column does the per-site work; the other one only has the
    // rank-4 MFMAs of its 12 tiles, which it issues one site late -- in the shadow of its partner's hand-overs
    const int u = (wv & 3) - 1, half = wv >> 2;
    const int l15 = lane & 15, l4 = lane >> 4;
    // what every wave does at the head of a block
    auto prologue = [&](const int b, const long i0, const int nb, long long* const stamps) {
        if (stamps && t == 0) stamps[0] = __builtin_amdgcn_s_memtime();
        if (t == 0) { seqC = 0; seqP = 0; seqG = 0; }
        if (t < EPB) {
            const bool live = t < nb;
            prm[t][0] = live ? ttau[i0 + t] : 0.0; prm[t][1] = live ? tnu[i0 + t] : 0.0;
            prm[t][2] = live ? m[i0 + t] : 0.0; prm[t][3] = live ? y[i0 + t] : 1.0;
            s_dt[t] = 0.0; s_dn[t] = 0.0; s_tn[t] = 0.0; s_nn[t] = 0.0;
        }
        // Sigma_BB and mu_B are the prep workgroups' to finish
        if (t == 0) ep_wait_ge(flags + EPF_PREP, pbase + 36u * ((unsigned)b + 1u), flags + EPF_ERR, 3u | ((unsigned)b << 8));
        __syncthreads();
        if (t < EPB) {
            s_mu0[t] = ld_dev(Tile + EP_TILE_N + t);
            diagb[0][t] = ld_dev(Tile + EP_TILE_N + EPB + t); mub[0][t] = s_mu0[t];
        }
        if (t < 8 * (EPCP - EPB)) colb[t / (EPCP - EPB)][EPB + t % (EPCP - EPB)] = 0.0;
    };
    // (the roles share nothing but LDS: everything of a role -- its loop over the blocks, its barriers -- sits inside its branch, so
    //  that the 96 accumulator registers of the update waves are not live across the chain's code)
    if (wv == 0) {
#pragma unroll 1
        for (int b = 0; b < nbl; ++b) {
            const long i0 = (long)b * EPB;
            const int nb = (int)(n - i0 < EPB ? n - i0 : EPB);
            double* const Wout = Wbuf + (b & 1) * EPB * EPB;
            double* const gout = gbuf + (b & 1) * EPB;
            double* const ldout = ldbuf + b;
            long long* const stamps = (TIMED && b == 5) ? stamps_b5 : nullptr;
            long long spinC = 0, spinP = 0, spinQ = 0;
            (void)Wout; (void)gout; (void)ldout; (void)spinC; (void)spinP; (void)spinQ;
            prologue(b, i0, nb, stamps);
            __syncthreads();
            if (stamps && t == 0) stamps[1] = __builtin_amdgcn_s_memtime();
            // ---- the chain: inf.py:757-770 for the sites of the block, nothing else -------------------------------------------
            __builtin_amdgcn_s_setprio(3);
            double dkk = diagb[0][0], muk = mub[0][0];
            double pc0 = prm[0][0], pc1 = prm[0][1], pc2 = prm[0][2], pc3 = prm[0][3];
            // log det B moves with every site by the matrix determinant lemma: det(Sigma^-1 + dtau e e') = det(Sigma^-1) (1 + dtau Sigma_ii)
            double fprod = 1.0, lsum = 0.0;
            for (int k = 0; k < nb; ++k) {
                const int kn = k + 1 < EPB ? k + 1 : EPB - 1;
                const double pn0 = prm[kn][0], pn1 = prm[kn][1], pn2 = prm[kn][2], pn3 = prm[kn][3];
                double t_new, nu_new, cj, qj;
                ep_site_update(dkk, muk, pc0, pc1, pc2, pc3, t_new, nu_new, cj, qj);
                fprod *= fma(t_new - pc0, dkk, 1.0);
                if ((k & 31) == 31) { lsum += log(fprod); fprod = 1.0; }
                if (lane == 0) {
                    cq[k & 15][0] = cj; cq[k & 15][1] = qj;
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    __hip_atomic_store(&seqC, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // LDS stores of one wave land in order
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    s_dt[k] = t_new - pc0; s_dn[k] = nu_new - pc1; s_tn[k] = t_new; s_nn[k] = nu_new;
                }
                if (k + 1 < nb) {
                    // Sigma(k+1,k), Sigma(k+1,k+1), mu(k+1) before site k: passed on by the update waves during their step k-1
                    // (the sequence number and the three values in one round trip, see the update waves)
                    double e1, d1, m1;
                    for (;;) {
                        const int sp = __hip_atomic_load(&seqP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __atomic_signal_fence(__ATOMIC_SEQ_CST);
                        e1 = colb[k & 7][k + 1]; d1 = diagb[k & 1][k + 1];
                        m1 = mub[k & 1][k + 1];
                        if (sp >= 3 * k) break;
                        if (TIMED) ++spinC;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    dkk = fma(-cj * e1, e1, d1);
                    muk = fma(qj, e1, m1);
                }
                pc0 = pn0; pc1 = pn1; pc2 = pn2; pc3 = pn3;
            }
            __syncthreads();
            if (stamps && t == 0) { stamps[2] = __builtin_amdgcn_s_memtime(); stamps[4] = stamps[1] + spinC; }
            for (int k = lane; k < nb; k += 64) { ttau[i0 + k] = s_tn[k]; tnu[i0 + k] = s_nn[k]; }
            if (lane == 0) ldout[0] = lsum + log(fprod);
            __syncthreads();
            __syncthreads();
            if (stamps && t == 0) stamps[3] = __builtin_amdgcn_s_memtime();
            // (the update waves saw W and g through before that last barrier)
            if (t == 0) __hip_atomic_store(flags + EPF_CHAIN, cbase + (unsigned)b + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (u < 0) {                          // wave 4
#pragma unroll 1
        for (int b = 0; b < nbl; ++b) {
            const long i0 = (long)b * EPB;
            const int nb = (int)(n - i0 < EPB ? n - i0 : EPB);
            double* const Wout = Wbuf + (b & 1) * EPB * EPB;
            double* const gout = gbuf + (b & 1) * EPB;
            double* const ldout = ldbuf + b;
            long long* const stamps = (TIMED && b == 5) ? stamps_b5 : nullptr;
            long long spinC = 0, spinP = 0, spinQ = 0;
            (void)Wout; (void)gout; (void)ldout; (void)spinC; (void)spinP; (void)spinQ;
            prologue(b, i0, nb, stamps);
            __syncthreads();
            __syncthreads();
            __syncthreads();
            __syncthreads();
        }
    } else {
#pragma unroll 1
        for (int b = 0; b < nbl; ++b) {
            const long i0 = (long)b * EPB;
            const int nb = (int)(n - i0 < EPB ? n - i0 : EPB);
            double* const Wout = Wbuf + (b & 1) * EPB * EPB;
            double* const gout = gbuf + (b & 1) * EPB;
            double* const ldout = ldbuf + b;
            long long* const stamps = (TIMED && b == 5) ? stamps_b5 : nullptr;
            long long spinC = 0, spinP = 0, spinQ = 0;
            (void)Wout; (void)gout; (void)ldout; (void)spinC; (void)spinP; (void)spinQ;
            prologue(b, i0, nb, stamps);
            // update waves: Sigma_BB in the MFMA accumulator layout -- tile (I, J) of wave u: rows 48 u + 16 I + l15, columns
            // 16 J + l4 + 4 r in component r (rows >= 128 are padding and stay zero)
            double4_t A[3][4];                       // tile (I, 4 half + jj)
            {
                // 48 values per lane, 512 bytes apart (ep_tile_index): windows of eight loads share one offset register
                const double* const tbase = Tile + __builtin_amdgcn_readfirstlane((u * 2 + half) * (3 * 4 * 4 * 64));
                double tv[4][12];
                static_for<6>([&](auto Wt) {
                    constexpr int wdw = decltype(Wt)::value;
                    const unsigned voff = (unsigned)lane * 8u + (unsigned)wdw * 4096u;
                    static_for<8>([&](auto Et) {
                        constexpr int e = 8 * wdw + decltype(Et)::value;
                        ld_dev_issue<512 * decltype(Et)::value>(tv[e / 12][e % 12], tbase, voff);
                    });
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) ep_loads_done(tv[q]);
                const bool pad_wave = u == 2;            // its tile row 2 = rows 128 .. 143: padding, zero
#pragma unroll
                for (int I = 0; I < 3; ++I)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int e = (I * 4 + jj) * 4 + r;
                            A[I][jj][r] = (I == 2 && pad_wave) ? 0.0 : tv[e / 12][e % 12];
                        }
#pragma unroll
                for (int I = 0; I < 3; ++I) {
                    const int row = 48 * u + 16 * I + l15;
                    if (half == 0 && l4 == 0 && row < EPB) colb[0][row] = A[I][0][0];
                }
            }
            __syncthreads();
            // ---- the update waves, one site behind ------------------------------------------------------------------------------
            // Per site the VALU keeps only the tile column(s) current from which the NEXT sites' columns leave (J = k / 16, 12 FMAs;
            // during the last four sites of a tile column also its successor) -- in "panel" registers Pa / Pb of the wave that
            // owns that tile column; the other wave of the pair skips the site.  Everything gets the FOUR rank-1 terms of the
            // sites 4 g .. 4 g + 3 at once, as one 16 x 16 x 4 MFMA per tile whose k index is the site: lane group l4 feeds column
            // 4 g + l4 (the last eight columns and sixteen (c, q) pairs stay in LDS rings) -- 12 matrix instructions per wave and
            // four sites instead of 4 x (96 + 8) VALU instructions.  (One MFMA per SITE with the k = 1..3 lanes zeroed was measured
            // first: fp64 MFMA and fp64 VALU both do 16 FMA per cycle and SIMD here, so three quarters of each were wasted.)
            const int di_ = 64 * u + lane;           // the diagonal / mu entry the publishing wave of the pair keeps up to date (< 128)
            double4_t Pa[3], Pb[3];
    #pragma unroll
            for (int I = 0; I < 3; ++I) { Pa[I] = A[I][0]; Pb[I] = A[I][0]; }
            // one site of tile column kc for the wave(s) that own Pa (mine_a) / Pb (mine_b, WB sites only); `pub`: this wave passes
            // column k+1, the diagonal and mu on.  The early column is component (second ? R1 : R0) of Pa in the lanes l4 == bn, or
            // (from_b) component 0 of Pb in the lanes l4 == 0.  Only R0 / R1 / WB are compile-time: four code bodies.
            auto u_site = [&](const int k, const int kc, auto R0t, auto R1t, auto WBt, const bool mine_a, const bool mine_b, const bool pub,
                              const bool second, const bool from_b, const int bn) {
                constexpr int r0 = decltype(R0t)::value, r1 = decltype(R1t)::value;
                constexpr bool wb = decltype(WBt)::value;
                const int b = k & 1;
                // ONE LDS round trip per try: the sequence numbers and everything they guard are read in the same batch -- LDS
                // executes a wave's instructions in order and the writers store their data before they raise the number, so data
                // read AFTER a number that has arrived has arrived too
                const double* cb = colb[k & 7];
                const int need_g = k >= 7 ? 6 * (((k - 7) >> 2) + 1) : 0;         // the column this step overwrites has been read by all
                double crow[3], cca[4], ccb[4], dI = 0.0, mI = 0.0, cI = 0.0, cj, qj;
                for (;;) {
                    const int sp = __hip_atomic_load(&seqP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int sc = __hip_atomic_load(&seqC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int sg = __hip_atomic_load(&seqG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    #pragma unroll
                    for (int I = 0; I < 3; ++I) crow[I] = cb[48 * u + 16 * I + l15];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cca[r] = cb[16 * kc + l4 + 4 * r];
                        ccb[r] = wb ? cb[16 * kc + 16 + l4 + 4 * r] : 0.0;        // (kc = 7: the zero padding of the column)
                    }
                    if (pub && di_ < EPB) { dI = diagb[b][di_]; mI = mub[b][di_]; cI = cb[di_]; }
                    cj = cq[k & 15][0]; qj = cq[k & 15][1];
                    if (sp >= 3 * k && sc >= k + 1 && (!pub || sg >= need_g)) break;
                    if (TIMED) { if (sp < 3 * k) ++spinP; else ++spinQ; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const double ncj = -cj;
    #pragma unroll
                for (int I = 0; I < 3; ++I) {
                    const double sr = ncj * crow[I];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mine_a) Pa[I][r] = fma(sr, cca[r], Pa[I][r]);
                        if (wb && mine_b) Pb[I][r] = fma(sr, ccb[r], Pb[I][r]);
                    }
                }
                if (pub) {
                    if (k + 1 < EPB) {
                        if (l4 == bn) {
    #pragma unroll
                            for (int I = 0; I < 3; ++I) {
                                const int row = 48 * u + 16 * I + l15;
                                const double nv = (wb && from_b) ? Pb[I][0] : (second ? Pa[I][r1] : Pa[I][r0]);
                                if (row < EPB) colb[(k + 1) & 7][row] = nv;
                            }
                        }
                        if (di_ < EPB) { diagb[b ^ 1][di_] = fma(ncj * cI, cI, dI); mub[b ^ 1][di_] = fma(qj, cI, mI); }
                    }
                    // (no wait for the stores: LDS executes this wave's add after them)
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    if (lane == 0) __hip_atomic_fetch_add(&seqP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                }
            };
            // the four sites k0 .. k0+3 as ONE rank-4 update of this wave's 12 tiles (the panels' own tiles too: they are overwritten
            // from the panels when their tile column is done).  `late`: the wave that skipped the sites waits one hand-over longer,
            // so that its MFMAs do not share the matrix pipe with its partner's batch
            auto u_group = [&](const int k0, const bool late) {
                const int ks = k0 + l4;
                const double* cb = colb[ks & 7];
                const int last = k0 + 3 < nb ? k0 + 3 : nb - 1;
                const int wait_p = late ? (last + 2 < nb ? 3 * (last + 2) : 3 * last) : 3 * last;
                double ncl, yop[3], xop[4];
                for (;;) {
                    const int sp = __hip_atomic_load(&seqP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int sc = __hip_atomic_load(&seqC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    ncl = ks < nb ? -cq[ks & 15][0] : 0.0;
    #pragma unroll
                    for (int I = 0; I < 3; ++I) yop[I] = cb[48 * u + 16 * I + l15];
    #pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xop[jj] = cb[16 * (4 * half + jj) + l15];
                    if (sp >= wait_p && sc >= last + 1) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                if (lane == 0) __hip_atomic_fetch_add(&seqG, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (late) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    #pragma unroll
                for (int I = 0; I < 3; ++I) yop[I] *= ncl;
    #pragma unroll
                for (int jj = 0; jj < 4; ++jj)
    #pragma unroll
                    for (int I = 0; I < 3; ++I) A[I][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[jj], yop[I], A[I][jj], 0, 0, 0);
            };
    #pragma unroll 1
            for (int kc = 0; kc < 8; ++kc) {
                if (16 * kc >= nb) break;
                const bool own_a = (kc >> 2) == half;                    // tile column kc is mine
                const bool own_b = kc < 7 && ((kc + 1) >> 2) == half;    // its successor is mine
                static_for<4>([&](auto KQt) {
                    // sites 16 kc + 4 kq + kb: column k+1 is component kq (kb < 3) or kq + 1 (kb == 3) of tile column kc, lanes l4 == kb + 1 mod 4
                    constexpr int kq = decltype(KQt)::value, kbn = kq == 3 ? 3 : 4;
                    if constexpr (kq == 3) {          // the successor joins: its accumulators have every group before this one
                        if (own_b) {
                            switch ((kc + 1) & 3) {
    #define PGP_EP_LOADB(J) case J: { _Pragma("unroll") for (int I = 0; I < 3; ++I) Pb[I] = A[I][J]; } break;
                                PGP_EP_LOADB(0) PGP_EP_LOADB(1) PGP_EP_LOADB(2) PGP_EP_LOADB(3)
    #undef PGP_EP_LOADB
                                default: break;
                            }
                        }
                    }
                    if (16 * kc + 4 * kq < nb) {
                        const bool act = own_a || (kq == 3 && own_b);
                        if (act) {
    #pragma unroll 1
                            for (int kb = 0; kb < kbn; ++kb) {
                                const int k = 16 * kc + 4 * kq + kb;
                                if (k < nb) u_site(k, kc, IntC<kq>{}, IntC<(kq + 1) & 3>{}, std::integral_constant<bool, kq == 3>{}, own_a, own_b, own_a,
                                                   kb == 3, false, (kb + 1) & 3);
                            }
                            if constexpr (kq == 3) {  // site 16 kc + 15: the next column is column 0 of the NEXT tile column
                                const int k = 16 * kc + 15;
                                if (k < nb) u_site(k, kc, IntC<0>{}, IntC<0>{}, std::true_type{}, own_a, own_b, kc < 7 ? own_b : own_a, false, true, 0);
                            }
                        }
                        u_group(16 * kc + 4 * kq, !act);
                    }
                });
                // tile column kc goes back to the accumulators (the later groups update it there); its successor becomes the panel
                if (own_a) {
                    switch (kc & 3) {
    #define PGP_EP_STOREA(J) case J: { _Pragma("unroll") for (int I = 0; I < 3; ++I) A[I][J] = Pa[I]; } break;
                        PGP_EP_STOREA(0) PGP_EP_STOREA(1) PGP_EP_STOREA(2) PGP_EP_STOREA(3)
    #undef PGP_EP_STOREA
                        default: break;
                    }
                }
    #pragma unroll
                for (int I = 0; I < 3; ++I) Pa[I] = Pb[I];
            }
            __syncthreads();
            if (stamps && t == 64) { stamps[5] = spinP; stamps[6] = spinQ; }
            // W(i, j) = dT_i [i == j] - dT_i dT_j Sigma_BB,new(i, j) ;  g = h - dT o (Sigma_BB,new h)
            // (W is stored as W(row, col) at [row + 128 col]: 16 consecutive rows per lane group; it is symmetric up to rounding)
    #pragma unroll
            for (int I = 0; I < 3; ++I) {
                const int i = 48 * u + 16 * I + l15;
                const bool live = i < EPB;
                const double di = live ? s_dt[i] : 0.0;
                double acc = 0.0;
    #pragma unroll
                for (int jj = 0; jj < 4; ++jj)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * (4 * half + jj) + l4 + 4 * r;
                        const double hj = fma(-s_dt[j], s_mu0[j], s_dn[j]);
                        acc = fma(A[I][jj][r], hj, acc);
                        if (live) st_dev(Wout + i + (long)EPB * j, (i == j ? di : 0.0) - di * s_dt[j] * A[I][jj][r]);
                    }
                acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);
                if (l4 == 0) gpart[half][48 * u + 16 * I + l15] = acc;
            }
            __syncthreads();
            if (half == 0 && l4 == 0) {
    #pragma unroll
                for (int I = 0; I < 3; ++I) {
                    const int i = 48 * u + 16 * I + l15;
                    if (i < EPB) {
                        const double di = s_dt[i];
                        st_dev(gout + i, fma(-di, gpart[0][i] + gpart[1][i], fma(-di, s_mu0[i], s_dn[i])));
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // W and g are through before wave 0 says so (a barrier only waits for LDS)
            __syncthreads();
        }
    }
    pgp_yield_mark(yield_flags, -1);
}

// The bulk stream waits for the chain's counter in a one-wave kernel of its own; stream order holds U, the fold and the strips back
// behind it (128 workgroups of U spinning on the counter themselves: 20.2 instead of 19.7 ms per fit, option ep_wait_kernel 0)
__global__ __launch_bounds__(64) void ep_wait_kernel(unsigned* f, unsigned target, unsigned* err) {
    if (threadIdx.x == 0) {
        ep_wait_ge(f, target, err, 4u | ((target & 0xffffu) << 8));
        __threadfence();
    }
}

// strip(:, k) = column i0 + k of the symmetric Sigma (kept in its lower triangle), all np rows but [skip0, skip1) (the block's own
// rows when its prep workgroups have written them).  Grid (np / 256, 8).
__global__ __launch_bounds__(256) void ep_strip_kernel(const double* __restrict__ Sig, long ld, long np, long i0,
                                                       double* __restrict__ S, long skip0, long skip1, unsigned* done) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    const int k0 = 16 * blockIdx.y;
    if (r < np && !(r >= skip0 && r < skip1)) {
        double v[16];
        if (r >= i0 + EPB) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = Sig[r + (i0 + k0 + k) * ld];
        } else if (r < i0) {
            const double2_t* src = reinterpret_cast<const double2_t*>(Sig + i0 + k0 + r * ld);
#pragma unroll
            for (int k = 0; k < 16; k += 2) { const double2_t x = src[k / 2]; v[k] = x[0]; v[k + 1] = x[1]; }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = sym_at(Sig, ld, r, i0 + k0 + k);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) st_dev(S + r + (long)(k0 + k) * ld, v[k]);
    }
    // the prep workgroups of the resident sweep kernel count the workgroups that are through (EPF_STRIP).  The strip is written at
    // agent scope like every other hand-over: a release fence per workgroup writes back a whole L2, 128 times per launch (13 us
    // instead of 5 for the kernel)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mu_r += sum_k S(r, k) g_k for the rows rlo <= r < rhi outside [skip0, skip1) (64 rows per workgroup, the columns split over the
// 4 waves, fixed order)
__global__ __launch_bounds__(256) void ep_mu_strip_kernel(const double* __restrict__ S, long ld, long rhi, long rlo,
                                                          const double* __restrict__ g, double* __restrict__ mu, long skip0, long skip1) {
    const long np = rhi;
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long r = rlo + (long)blockIdx.x * 64 + lane;
    const bool mine = r < np && !(r >= skip0 && r < skip1);
    double acc = 0.0;
    if (mine) {
#pragma unroll 4
        for (int k = grp * (EPB / 4); k < (grp + 1) * (EPB / 4); ++k) acc = fma(g[k], S[r + (long)k * ld], acc);
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && mine) mu[r] += ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// F (column-major lower, ldf) = I + s s' o K ; Y (column-major, ld np) = diag(s) K     (K symmetric, ld np)
__global__ __launch_bounds__(256) void ep_build_kernel(const double* __restrict__ K, long np,
                                                       const double* __restrict__ s, double* __restrict__ F, long ldf,
                                                       double* __restrict__ Y, int colscale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const double si = s[i];
    for (long j = blockIdx.y; j < np; j += gridDim.y) {      // (grid.y is capped at 65535: columns by stride)
        const double k = K[i + j * np];
        if (Y) Y[i + j * np] = (colscale ? s[j] : si) * k;    // colscale: Y = K diag(s) (= (diag(s) K)' : the rhs ROWS of the sweep)
        if (i >= j) F[i + j * ldf] = (i == j ? 1.0 : 0.0) + si * s[j] * k;
    }
}

// Per-site terms of the EP marginal likelihood (inf.py:184-188) and, optionally, d lZ_i / d mu (inf.py:788-790) on the device:
// the host loop over n sites (probit moments in double, ~0.3 us each) sat between every sweep and the next one -- 1.2 ms of
// idle GPU per parameter recomputation at N = 4096.  Block partials [blk][5] = (sum lZ, t3, t4, t5, t6) in a fixed order.
__global__ __launch_bounds__(256) void ep_site_terms_kernel(long n, const double* __restrict__ y, const double* __restrict__ m,
                                                            const double* __restrict__ mu, const double* __restrict__ dsig,
                                                            double dsig_const, const double* __restrict__ ttau,
                                                            const double* __restrict__ tnu, int with_m,
                                                            double* __restrict__ partial, double* __restrict__ dlz) {
    __shared__ double red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (i < n) {
        const double ds = dsig ? dsig[i] : dsig_const, mui = mu ? mu[i] : 0.0;
        const double tt = ttau ? ttau[i] : 0.0, tn = tnu ? tnu[i] : 0.0;
        const double tau_n = 1.0 / ds - tt;
        const double nu_n = mui / ds - tn + (with_m ? m[i] * tau_n : 0.0);
        double lZ, dl;
        erf_ep_moments(y[i], nu_n / tau_n, 1.0 / tau_n, &lZ, &dl, nullptr);
        if (dlz) dlz[i] = dl;
        const double a = nu_n - (with_m ? m[i] * tau_n : 0.0);
        v[0] = lZ;
        v[1] = tn * mui;
        v[2] = a * ((tt / tau_n * a - 2.0 * tn) / (tt + tau_n));
        v[3] = tn * tn / (tau_n + tt);
        v[4] = log(1.0 + tt / tau_n);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double sres = block_sum(v[q], red);
        if (threadIdx.x == 0) partial[5L * blockIdx.x + q] = sres;
    }
}

// R = sW sW' o B^-1 without B^-1: sW (I + sW K sW)^-1 sW = (S^-1 + K)^-1 = S - S Sigma S with S = diag(ttau) and the
// Sigma = (K^-1 + S)^-1 that _epComputeParams has just rebuilt (matrix inversion lemma); lower triangle, column-major
__global__ __launch_bounds__(256) void ep_r_from_sigma_kernel(const double* __restrict__ Sig, long ld, long np,
                                                              const double* __restrict__ ttau, double* __restrict__ R, long ldr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const double ti = ttau[i];
    for (long j = blockIdx.y; j <= i; j += gridDim.y) R[i + j * ldr] = (i == j ? ti : 0.0) - ti * ttau[j] * Sig[i + j * ld];
}

// E'(k, m) = s_k E(k, m) on the block-upper part of E = L^-T (everything the sweep wrote: k < 128 (m / 128 + 1))
__global__ __launch_bounds__(256) void ep_rowscale_upper_kernel(double* __restrict__ E, long lde, const double* __restrict__ s, long np) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    for (long m = blockIdx.y; m < np; m += gridDim.y)
        if (k < (m / 128 + 1) * 128) E[k + m * lde] *= s[k];
}

struct EpWork {
    long n, np, ldf;
    double* Ed;                          // fused path: E = L^-T from the sweep, then diag(sW) E
    double *Kd, *Sig, *Vd, *F, *Wd, *rhs;
    double *ttau_d, *tnu_d, *mu_d, *m_d, *s_d, *sbuf, *coef, *diag_d, *tmp_d;
    double *S, *Sc;                      // block sweep: the strips Sigma(:, B) of the last two blocks and U = strip W
    double *Wb, *gb, *ldb;               // block sweep: W and g of the last two blocks, log of the blocks' determinant factors
    double* tile;                        // block sweep: Sigma_BB, mu_B, diag Sigma_BB as the prep workgroups hand them to the chain
    unsigned* flags;                     // EPF_*: the device counters through which the resident sweep kernel and the bulk stream meet
    unsigned chain_total, prep_total, strip_total;   // their values once everything launched so far has run (strips in launches)
};

}  // namespace

// recompute Sigma, mu, L from (ttau, tnu) and return nlZ (inf.py:174-189).  Host vectors in/out.
static int ep_compute_params(pgp_ctx* c, EpWork& w, const std::vector<double>& y, const std::vector<double>& m,
                             const std::vector<double>& ttau, const std::vector<double>& tnu, double* nlZ_out,
                             std::vector<double>& mu_h, std::vector<double>& dsig_h, double* half_logdet_out = nullptr) {
    hipStream_t st = c->st;
    const long n = w.n, np = w.np;
    std::vector<double> s_h(np, 0.0);
    for (long i = 0; i < n; ++i) s_h[i] = sqrt(ttau[i]);
    HIP_TRY(hipMemcpyAsync(w.s_d, s_h.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w.ttau_d, ttau.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w.tnu_d, tnu.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->info_dev, 0, sizeof(int), st));
    const bool fusedp = c->ep_fused == 1 && w.Ed;
    const bool rhsp = c->ep_fused == 2;             // V' = (K diag(sW)) L^-T as dense right-hand-side ROWS of the sweep itself
    bool sigma_done = false;
    hipLaunchKernelGGL(ep_build_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, w.Kd, np,
                       w.s_d, w.F, w.ldf, fusedp ? nullptr : w.Vd, rhsp ? 1 : 0);
    // fused path: the sweep also yields E = L^-T, so V' = (K diag(sW)) E is ONE clipped MFMA product (no blocked multi-rhs
    // solve, no leaf inverses) and Sigma = K - V'V'^T an NT product in the LDS-DMA form
    // (the sweep's two-piece row space wants the factor's 128 spare rows between the factor and the inverse rows, like the
    //  exact fit's rhs rows: they are zeroed and ride along)
    if (fusedp) {
        CHK(zero_strip_launch(w.F, w.ldf, np, np, 128, st));
        CHK(potrf_blocked(c, w.F, w.ldf, np, np + 128, true, w.Ed, np));
    } else if (rhsp) {
        // the np rows of K diag(sW) ride along in the panel solves and trailing updates (np^3 flops inside the bulk MFMA
        // launches, which at N = 4096 also gives the chain of diagonal blocks enough work to hide behind): no inverse
        // rows (np^3 / 3 less) and no separate product
        CHK(zero_strip_launch(w.F, w.ldf, np, np, 128, st));
        // Sigma = K - V'V'^T accumulated under the sweep, panel by panel (lower tiles; mirrored below)
        const bool under = c->ep_sym && c->ep_sigma_under;
        if (under) {
            HIP_TRY(hipMemcpyAsync(w.Sig, w.Kd, (size_t)np * np * sizeof(double), hipMemcpyDeviceToDevice, st));
            c->fill2_C = w.Sig; c->fill2_ld = np;
        }
        const int prc = potrf_blocked_rhs(c, w.F, w.ldf, np, np + 128, w.Vd, np, np);
        c->fill2_C = nullptr;
        CHK(prc);
        sigma_done = under;
    }
    else CHK(potrf_blocked(c, w.F, w.ldf, np, np));
    int info = 0;
    HIP_TRY(hipMemcpyAsync(&info, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (info != 0) return info > (int)n ? (int)n : info;
    if (fusedp) {
        hipLaunchKernelGGL(ep_rowscale_upper_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, w.Ed, np, w.s_d, np);
        GemmArgs g{};                                                                   // V'(n, m) = sum_{k <= m} K(n, k) sW_k E(k, m)
        g.A = w.Kd; g.lda = np; g.a_kc = 0;
        g.B = w.Ed; g.ldb = np; g.b_kc = 1;
        g.C = w.Vd; g.ldc = np; g.M = (int)np; g.N = (int)np; g.K = (int)np;
        g.alpha = 1.0; g.beta = 0.0; g.kmode = KM_LT_J; g.koff = 0; g.rev_cols = 1;
        g.tile = (np / 128) * (np / 128) < c->small_tile_below ? 64 : 128;
        g.flops = (double)np * np * np;
        CHK(gemm_prof(c, PC_GEMM_SOLVE, g));
    } else if (!rhsp) {
        CHK(leaf_inv_launch(w.F, w.ldf, w.Wd, 128, 128L * 128L, (int)(np / 128), st));
        CHK(solve_lower_multi(c, w.F, w.ldf, w.Wd, w.Vd, np, np, (int)np, false));    // V = L^-1 (sW o K)
    }
    if (sigma_done) hipLaunchKernelGGL(ep_mirror_kernel, dim3((unsigned)(np / 64), (unsigned)(np / 64)), dim3(256), 0, st, w.Sig, np);
    else {
    HIP_TRY(hipMemcpyAsync(w.Sig, w.Kd, (size_t)np * np * sizeof(double), hipMemcpyDeviceToDevice, st));
    {
        GemmArgs g{};                                                                   // Sigma = K - V'V
        g.A = w.Vd; g.lda = np; g.a_kc = (fusedp || rhsp) ? 0 : 1;                      // fused paths: Vd holds V' (n-contiguous)
        g.B = w.Vd; g.ldb = np; g.b_kc = (fusedp || rhsp) ? 0 : 1;
        g.C = w.Sig; g.ldc = np; g.M = (int)np; g.N = (int)np; g.K = (int)np;
        g.alpha = -1.0; g.beta = 1.0; g.tile = (np / 128) * (np / 128) < c->small_tile_below ? 64 : 128;
        g.flops = 2.0 * (double)np * np * np;
        if (c->ep_sym) { g.tri = 2; g.mask_diag = 1; g.flops *= 0.5; }                  // lower tiles only, then mirrored
        CHK(gemm_prof(c, PC_GEMM_INNER, g));
        if (c->ep_sym)
            hipLaunchKernelGGL(ep_mirror_kernel, dim3((unsigned)(np / 64), (unsigned)(np / 64)), dim3(256), 0, st, w.Sig, np);
    }
    }
    CHK(col_dot_full_launch(w.Sig, np, np, np, w.tnu_d, nullptr, w.mu_d, st));           // mu = Sigma tnu
    CHK(gather_strided_launch(w.Sig, np + 1, np, w.diag_d, st));
    CHK(logdet_ztz_launch(w.F, w.ldf, n, w.F, 0, c->scal, st));
    const long nblk_terms = (n + 255) / 256;
    std::vector<double> part_h(5 * nblk_terms);
    hipLaunchKernelGGL(ep_site_terms_kernel, dim3((unsigned)nblk_terms), dim3(256), 0, st, n, c->y_dev, w.m_d, w.mu_d, w.diag_d, 0.0,
                       w.ttau_d, w.tnu_d, 1, w.tmp_d, (double*)nullptr);
    HIP_TRY(hipMemcpyAsync(part_h.data(), w.tmp_d, part_h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    double sc[2];
    HIP_TRY(hipMemcpyAsync(mu_h.data(), w.mu_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dsig_h.data(), w.diag_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(sc, c->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // -log marginal likelihood (inf.py:184-188): the per-site terms were reduced on the device (ep_site_terms_kernel, queued
    // before the copies above); the block partials are added here in block order
    double slZ = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0, t6 = 0.0;
    for (long b = 0; b < nblk_terms; ++b) {
        slZ += part_h[5 * b]; t3 += part_h[5 * b + 1]; t4 += part_h[5 * b + 2]; t5 += part_h[5 * b + 3]; t6 += part_h[5 * b + 4];
    }
    *nlZ_out = sc[0] - slZ - 0.5 * t3 - 0.5 * t4 + 0.5 * t5 - 0.5 * t6;
    if (half_logdet_out) *half_logdet_out = sc[0];
    return PGP_OK;
}

// L = chol(I + sW sW' o K) of the given site parameters into w.F -- and nothing else (no V, no Sigma, no mu): what is left to
// do when Sigma, mu and log det B were carried through the sweeps (ep_fit_core, track) and only post.L is missing.
static int ep_factor_only(pgp_ctx* c, EpWork& w, const std::vector<double>& ttau) {
    hipStream_t st = c->st;
    const long n = w.n, np = w.np;
    std::vector<double> s_h(np, 0.0);
    for (long i = 0; i < n; ++i) s_h[i] = sqrt(ttau[i]);
    HIP_TRY(hipMemcpyAsync(w.s_d, s_h.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->info_dev, 0, sizeof(int), st));
    hipLaunchKernelGGL(ep_build_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, w.Kd, np, w.s_d, w.F, w.ldf,
                       (double*)nullptr, 0);
    {
        // a plain factorisation (no inverse rows, no E E' to fill the main stream) is bound by the chain of diagonal blocks: 1024-wide
        // panels halve the number of panel steps (cfg 5, N = 4096: 17.8 -> 17.4 ms per fit); an explicit nb_outer option wins
        const int keep = c->nb_outer;
        if (keep == 0 && np >= 2048) c->nb_outer = 8;
        const int prc = potrf_blocked(c, w.F, w.ldf, np, np);
        c->nb_outer = keep;
        CHK(prc);
    }
    int info = 0;
    HIP_TRY(hipMemcpyAsync(&info, c->info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));               // (s_h goes out of scope)
    if (info != 0) return info > (int)n ? (int)n : info;
    return PGP_OK;
}

namespace {
__global__ void probit_hazard_test_kernel(const double* __restrict__ z, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = probit_hazard(z[i]);
}
}  // namespace

// self-test hook: N(z) / Phi(z) of the latency-trimmed site update for host values z > -5 (tests compare with mpmath)
extern "C" int pgp_test_probit_hazard(pgp_ctx* c, const double* z, double* out, int n) {
    if (!c || !z || !out || n <= 0) return -1;
    HIP_TRY(hipSetDevice(c->device));
    DevScratch scr;
    double *zd, *od;
    CHK(scr.alloc(&zd, (size_t)n * 8)); CHK(scr.alloc(&od, (size_t)n * 8));
    HIP_TRY(hipMemcpy(zd, z, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probit_hazard_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->st, zd, od, n);
    HIP_TRY(hipStreamSynchronize(c->st));
    HIP_TRY(hipMemcpy(out, od, (size_t)n * 8, hipMemcpyDeviceToHost));
    return PGP_OK;
}


// The EP fit.  Kdense == nullptr: K is a device program (kind, covhyp, ...) assembled on the device, the gradients come back in
// dnlZ_out.  Kdense != nullptr (pgp_ep_fit_dense): K (n x n, symmetric, host) is handed in -- a covariance tree that is not
// a device program --; dnlZ_out receives the mean gradients only, and R = sW sW' o B^-1 and alpha stay in the context's
// workspace for the pgp_dense_grad_term calls that follow (1/2 sum (R - alpha alpha') o dK_h, inf.py:780-786).
// a bounded device-side wait of the block sweep gave up: say which one (site 1 prep<-chain, 2 prep<-strip, 3 chain<-prep, 4 bulk<-chain)
static int ep_wait_failed(const unsigned* eflags, unsigned chain_total, unsigned prep_total, unsigned strip_total, int line) {
    char msg[256];
    snprintf(msg, sizeof(msg), "EP block sweep: a device-side wait gave up (site %u, block/target %u; counters chain %u/%u prep %u/%u strip-wgs %u, strips launched %u)",
             eflags[EPF_ERR] & 0xffu, (eflags[EPF_ERR] >> 8) & 0x7fffffu, eflags[EPF_CHAIN], chain_total, eflags[EPF_PREP], prep_total,
             eflags[EPF_STRIP], strip_total);
    pgp_set_last_hip_error(hipErrorLaunchTimeOut, msg, __FILE__, line);
    return PGP_ERR_HIP;
}

static int ep_fit_core(pgp_ctx* c, const double* Kdense, int kind, const double* covhyp, int ncov, int para, int flags,
                       const double* mvec, const double* dm, int nmean, int want, int warm, double* ttau_io, double* tnu_io,
                       double* alpha_out, double* sW_out, double* nlZ_out, double* dnlZ_out, int* sweeps_out,
                       pgp_factor** factor_out) {
    if (!c) return -1;
    if (c->n <= 0) return -1;
    if (!covhyp && !Kdense) return -3;
    if (!ttau_io || !tnu_io) return -12;
    GateShared gate(c);                              // shared for the fit, exclusive during each block sweep (ctx.h DeviceGate)
    HIP_TRY(hipSetDevice(c->device));
    c->dense_ready = false;                          // the workspace (B^-1, alpha) is about to be rewritten
    hipStream_t st = c->st;
    const long n = c->n, d = c->d, np = c->np, ldf = c->ldf;
    const bool dense = Kdense != nullptr;
    CovSpec cp;
    if (dense) ncov = 0;
    else { const int rc = make_spec(c, kind, covhyp, ncov, para, flags, -1, d, cp); if (rc != PGP_OK) return rc == -11 ? -10 : rc; }
    const std::vector<double>& sc = cp.scale;
    CHK(ensure_workspace(c, np));
    double kdiag = 0.0;                               // K_ii, identical for every training point (stationary kernels)
    double kss = 0.0;
    if (!dense) {
        CHK(cov_point_value(c, cp, 1, &kdiag));
        CHK(cov_point_value(c, cp, 2, &kss));
    }
    const long need = std::max<long>(hadamard_partial_count(np, ncov), np);
    if (want >= 3 && c->partial_cap < need) {
        if (c->partial) (void)hipFree(c->partial);
        c->partial = nullptr; c->partial_cap = 0;      // a failed realloc must not leave a dangling pointer behind
        HIP_TRY(hipMalloc((void**)&c->partial, need * sizeof(double)));
        c->partial_cap = need;
    }
    static const bool ep_timing = getenv("PGP_EP_TIMING") != nullptr;      // host wall-clock stamps of the phases (stderr)
    const auto tp0 = std::chrono::steady_clock::now();
    // phase times of this fit, host wall clock (every phase ends synchronised): pgp_last_timings reports them as
    // assemble = K + first parameters, solve = the site sweeps, potrf = the parameter recomputations, grad = alpha + gradients
    double ph_ms[4] = {0.0, 0.0, 0.0, 0.0};
    auto tlast = tp0;
    auto stamp = [&](const char* what, int phase = -1) {
        const auto now = std::chrono::steady_clock::now();
        if (phase >= 0) ph_ms[phase] += std::chrono::duration<double, std::milli>(now - tlast).count();
        tlast = now;
        if (ep_timing) fprintf(stderr, "[ep] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tp0).count());
    };
    EpWork w{};
    w.n = n; w.np = np; w.ldf = ldf;
    const size_t nn = (size_t)np * np * sizeof(double);
    // RAII: every early return (HIP_TRY / EP_TRY / CHK) gives the scratch back to the context's pool, scrubs + returns the
    // factor buffer and frees a half-built handle -- no hipMalloc / hipFree (device-synchronising) on the steady-state path
    PoolScratch pscr(c);
    auto dalloc = [&](double** p, size_t bytes) -> int { return pscr.alloc(p, bytes); };
#define EP_TRY(x) CHK(x)
    EP_TRY(dalloc(&w.Kd, nn)); EP_TRY(dalloc(&w.Sig, nn)); EP_TRY(dalloc(&w.Vd, nn));
    if (c->ep_fused == 1) EP_TRY(dalloc(&w.Ed, nn));
    EP_TRY(dalloc(&w.Wd, (size_t)128 * np * sizeof(double)));
    EP_TRY(dalloc(&w.rhs, (size_t)128 * np * sizeof(double)));
    double* vecs = nullptr;
    EP_TRY(dalloc(&vecs, (size_t)10 * np * sizeof(double)));
    w.ttau_d = vecs; w.tnu_d = vecs + np; w.mu_d = vecs + 2 * np; w.m_d = vecs + 3 * np; w.s_d = vecs + 4 * np;
    w.sbuf = vecs + 5 * np; w.coef = vecs + 6 * np; w.diag_d = vecs + 7 * np; w.tmp_d = vecs + 8 * np;
    HIP_TRY(hipMemsetAsync(vecs, 0, (size_t)10 * np * sizeof(double), st));
    EP_TRY(dalloc(&w.S, (size_t)2 * EPB * np * sizeof(double)));
    EP_TRY(dalloc(&w.Sc, (size_t)EPB * np * sizeof(double)));
    EP_TRY(dalloc(&w.Wb, (size_t)2 * EPB * EPB * sizeof(double)));
    EP_TRY(dalloc(&w.tile, (size_t)(EP_TILE_N + 2 * EPB) * sizeof(double)));
    EP_TRY(dalloc(&w.gb, (size_t)(2 * EPB + 16) * sizeof(double)));
    w.ldb = c->alpha_dev;                // [log det factors (np / 128) | . | counters (2 doubles) | per-site partial sums]: the head of the
                                         // result buffer (alpha is written there when the sweeps are over), one pinned copy per sweep
    w.flags = (unsigned*)(w.ldb + np / EPB + 1);
    w.chain_total = w.prep_total = w.strip_total = 0u;
    HIP_TRY(hipMemsetAsync(w.flags, 0, 2 * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(w.Kd, 0, nn, st));
    EP_TRY(alloc_factor_buffer(c, np, ldf, &w.F));
    FactorGuard fguard(c, w.F, (size_t)ldf * np * sizeof(double), /*scrub=*/true);
    stamp("scratch acquired");
    // ---- K (full symmetric, padded with zeros) --------------------------------------------------------
    if (dense) HIP_TRY(hipMemcpy2DAsync(w.Kd, np * sizeof(double), Kdense, n * sizeof(double), n * sizeof(double), n, hipMemcpyHostToDevice, st));
    else {
        EP_TRY(upload_scaled(c, c->x_dev, n, d, sc, c->XsT, np, c->dpad, c->scale_dev));
        if (gram_assembly_applies(c, cp)) {            // RBF / RBFard at d >= 32 (cfg 5: d = 32): the Gram form on the matrix cores
            EP_TRY(hadamard_prepare_launch(c->XsT, np, n, np, c->dpad, cp, c->prep, st, /*force=*/true));
            EP_TRY(cov_sym_gram_launch(c->XsT, np, n, c->dpad, cp, w.Kd, np, c->prep, st));
        } else
        EP_TRY(cov_sym_launch(c->XsT, np, n, c->dpad, cp, w.Kd, st, np));
    }
    std::vector<double> m(n, 0.0), y(n), ttau(n, 0.0), tnu(n, 0.0), mu(n, 0.0), dsig(n, kdiag);
    if (mvec) memcpy(m.data(), mvec, n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(y.data(), c->y_dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(w.m_d, m.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    // nlZ0 = -sum lik(y, m, diag K)  (inf.py:737)
    double nlZ0 = 0.0;
    {   // the same per-site kernel with mu = 0, Sigma_ii = K_ii, zero site parameters: lZ_i = lik(y_i, m_i, K_ii)
        const long nbt = (n + 255) / 256;
        std::vector<double> ph(5 * nbt);
        if (dense) EP_TRY(gather_strided_launch(w.Kd, np + 1, np, w.diag_d, st));          // K_ii differs from point to point
        hipLaunchKernelGGL(ep_site_terms_kernel, dim3((unsigned)nbt), dim3(256), 0, st, n, c->y_dev, w.m_d, (const double*)nullptr,
                           dense ? (const double*)w.diag_d : (const double*)nullptr, kdiag, (const double*)nullptr, (const double*)nullptr,
                           1, w.tmp_d, (double*)nullptr);
        HIP_TRY(hipMemcpyAsync(ph.data(), w.tmp_d, ph.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (long b = 0; b < nbt; ++b) nlZ0 -= ph[5 * b];
    }
    double nlZ = nlZ0;
    bool fresh = true;
    int rc = PGP_OK;
    // track (default): Sigma, mu and log det B are carried through the sweeps by exact identities (Woodbury folds of whole
    // blocks, the determinant lemma per site) and the posterior is rebuilt from scratch ONCE, from the converged site
    // parameters (everything returned comes from that rebuild).  ep_recompute 1: the reference's schedule, a rebuild after
    // every sweep (inf.py:772).
    const bool track = c->ep_block && !c->ep_recompute;
    double half_logdet = 0.0;                          // sum log diag chol(B): B = I at the cold start
    if (warm) {                                                                       // inf.py:744-753
        memcpy(ttau.data(), ttau_io, n * sizeof(double));
        memcpy(tnu.data(), tnu_io, n * sizeof(double));
        rc = ep_compute_params(c, w, y, m, ttau, tnu, &nlZ, mu, dsig, &half_logdet);
        if (rc == PGP_OK && !(nlZ > nlZ0)) fresh = false;
        if (rc > 0) rc = PGP_OK;                                                      // bad warm start: fall back to zeros
        if (rc != PGP_OK) return rc;
    }
    if (fresh) {
        std::fill(ttau.begin(), ttau.end(), 0.0);
        std::fill(tnu.begin(), tnu.end(), 0.0);
        nlZ = nlZ0;
        half_logdet = 0.0;
        HIP_TRY(hipMemcpyAsync(w.Sig, w.Kd, nn, hipMemcpyDeviceToDevice, st));        // Sigma = K, mu = 0
        HIP_TRY(hipMemsetAsync(w.mu_d, 0, np * sizeof(double), st));
        HIP_TRY(hipMemsetAsync(w.ttau_d, 0, np * sizeof(double), st));
        HIP_TRY(hipMemsetAsync(w.tnu_d, 0, np * sizeof(double), st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    stamp("K built, nlZ0", 0);
    const double tol = 1e-4;
    const int max_sweep = 10, min_sweep = 2;
    double nlZ_old = INFINITY;
    int sweep = 0;
    // ONE block sweep per device at a time, and nothing else of this process beside it (ctx.h DeviceGate: the sweep is a resident
    // kernel on one stream that meets bulk launches on another through device counters; a foreign launch with a cross-stream wait
    // on a shared hardware queue can close a cycle).  Exclusive from a sweep's first launch until the host synchronisation that
    // follows it; K, the final Cholesky, alpha and the gradients of an EP fit run under the shared gate beside other contexts' work.
    struct SweepExclusive {
        GateShared& g; bool on, held = false;
        SweepExclusive(GateShared& g_, bool on_) : g(g_), on(on_) {}
        void acquire() { if (on && !held) { g.exclusive(); held = true; } }
        void release() { if (held) { g.shared_again(); held = false; } }
        ~SweepExclusive() { release(); }
    } sweep_excl(gate, c->ep_block != 0);
    // (round 6) The first two sweeps are unconditional (min_sweep = 2, inf.py:732): sweep 2 is queued straight behind sweep 1 -- no
    // host round trip between them.  What the host needs of sweep 1 (its blocks' log det factors and the per-site partial sums:
    // nlZ after sweep 1 is what sweep 2's convergence test compares with) waits in a second slot of the result buffer and comes
    // back with sweep 2's in one copy.  Option ep_merge12 = 0: a synchronisation after every sweep, as before.
    const bool merge12 = track && c->ep_block && c->ep_merge12 && min_sweep >= 2 && !ep_timing;
    const long res_len = np / EPB + 3 + 5 * ((n + 255) / 256), res_slot = (res_len + 15) & ~15L;   // [log det factors | counters | partial sums]
    bool deferred = false;                             // sweep 1's results are in the second slot
    while ((fabs(nlZ - nlZ_old) > tol && sweep < max_sweep) || sweep < min_sweep) {
        nlZ_old = nlZ;
        ++sweep;
        sweep_excl.acquire();
        if (c->ep_block) {
            // block sweep: the chain stream (the high-priority panel stream) runs ONE resident kernel per sweep (chain + prep workgroups);
            // the bulk stream (main) strip(b), U(b), fold(b), mu(b) per block.  The two meet through device counters (ep_chain_kernel).
            const long nbl = (n + EPB - 1) / EPB;
            hipStream_t sa = c->st2 ? c->st2 : st, sb = st;
            while ((long)c->ep_ev.size() < 2) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                c->ep_ev.push_back(e);
            }
            unsigned* yfl = (c->yield && c->yield_flags) ? c->yield_flags : nullptr;
            const unsigned swg = (unsigned)((np + 255) / 256) * 8u;       // workgroups of one strip launch
            HIP_TRY(hipEventRecord(c->ep_ev[0], st));
            if (sa != st) HIP_TRY(hipStreamWaitEvent(sa, c->ep_ev[0], 0));
            if (sa == st) return PGP_ERR_HIP;          // (the resident kernel needs a stream of its own beside the bulk launches)
            if (ep_timing)
                hipLaunchKernelGGL(ep_chain_kernel<true>, dim3(37), dim3(512), EP_BLOCK_LDS, sa, w.Sig, np, n, (int)nbl, w.mu_d, w.m_d,
                                   c->y_dev, w.ttau_d, w.tnu_d, w.Wb, w.gb, w.ldb, yfl, (long long*)(w.gb + 2 * EPB), w.S, w.tile, w.flags,
                                   w.chain_total, w.prep_total, w.strip_total, swg);
            else
                hipLaunchKernelGGL(ep_chain_kernel<false>, dim3(37), dim3(512), EP_BLOCK_LDS, sa, w.Sig, np, n, (int)nbl, w.mu_d, w.m_d,
                                   c->y_dev, w.ttau_d, w.tnu_d, w.Wb, w.gb, w.ldb, yfl, (long long*)nullptr, w.S, w.tile, w.flags,
                                   w.chain_total, w.prep_total, w.strip_total, swg);
            HIP_TRY(hipEventRecord(c->ep_ev[1], sa));
            // from here on the resident kernel is spinning on device counters: an early return (a failed launch on the bulk stream)
            // must stop it and wait for it before the scratch it reads and writes goes back to the pool
            struct ChainGuard {
                unsigned* err; hipStream_t sa, sb; bool armed = true;
                ~ChainGuard() {
                    if (!armed) return;
                    const unsigned one = 1u;
                    (void)hipMemcpy(err, &one, sizeof(one), hipMemcpyHostToDevice);   // every device-side wait gives up at once
                    (void)hipStreamSynchronize(sa);
                    (void)hipStreamSynchronize(sb);
                }
            } chain_guard{w.flags + EPF_ERR, sa, sb};
            const unsigned cbase = w.chain_total;
            w.chain_total += (unsigned)nbl;
            w.prep_total += 36u * (unsigned)nbl;
            for (long b = 0; b < nbl; ++b) {
                const long i0 = b * EPB;
                double* Wb = w.Wb + (b & 1) * EPB * EPB;
                double* gb = w.gb + (b & 1) * EPB;
                double* Sb = w.S + (b & 1) * EPB * np;  // strip(b); its rows B come from the prep workgroups
                const bool last = b + 1 >= nbl;
                if (last && !track) break;             // nothing of this sweep reads what the last block does to the rest
                hipLaunchKernelGGL(ep_strip_kernel, dim3((unsigned)((np + 255) / 256), 8), dim3(256), 0, sb, w.Sig, np, np, i0, Sb,
                                   b > 0 ? i0 : 0L, b > 0 ? i0 + EPB : 0L, w.flags + EPF_STRIP);
                ++w.strip_total;
                const long r0 = std::min<long>(i0 + EPB, np);   // first row of the sites still to come
                const long u0 = track ? 0 : r0;        // track: Sigma stays complete (every row), otherwise only the rows still to be read
                {
                    GemmArgs g{};                       // U = strip W
                    g.A = Sb + u0; g.lda = np; g.a_kc = 0; g.B = Wb; g.ldb = EPB; g.b_kc = 0;
                    g.C = w.Sc + u0; g.ldc = np; g.M = (int)(np - u0); g.N = EPB; g.K = EPB;
                    g.alpha = 1.0; g.beta = 0.0; g.tile = 64; g.flops = 2.0 * (double)(np - u0) * EPB * EPB;
                    if (c->ep_wait_kernel) hipLaunchKernelGGL(ep_wait_kernel, dim3(1), dim3(64), 0, sb, w.flags + EPF_CHAIN, cbase + (unsigned)b + 1u, w.flags + EPF_ERR);
                    else { g.wait_flag = w.flags + EPF_CHAIN; g.wait_target = cbase + (unsigned)b + 1u; g.wait_err = w.flags + EPF_ERR; }
                    EP_TRY(gemm_prof(c, PC_GEMM_INNER, g, sb));
                }
                const long r1 = r0 + EPB;
                if (track) {
                    // the whole lower triangle in ONE launch (but the next block's diagonal tile: its prep workgroups own it), mu likewise
                    GemmArgs g{};
                    g.A = w.Sc; g.lda = np; g.a_kc = 0; g.B = Sb; g.ldb = np; g.b_kc = 0;
                    g.C = w.Sig; g.ldc = np; g.M = (int)np; g.N = (int)np; g.K = EPB;
                    g.alpha = -1.0; g.beta = 1.0; g.tile = np >= 1024 ? 128 : 64; g.tri = 2; g.mask_diag = 1;
                    if (!last) { g.skip_lo = (int)r0; g.skip_hi = (int)r1; }
                    g.flops = (double)EPB * ((double)np * np - (last ? 0.0 : (double)EPB * EPB));
                    EP_TRY(gemm_prof(c, PC_GEMM_INNER, g, sb));
                    hipLaunchKernelGGL(ep_mu_strip_kernel, dim3((unsigned)((np + 63) / 64)), dim3(256), 0, sb, Sb, np, np, 0L, gb, w.mu_d,
                                       last ? 0L : r0, last ? 0L : r1);
                } else {
                    if (!last) {
                        GemmArgs g{};                   // the next block's rows left of its diagonal tile (prep owns that tile)
                        g.A = w.Sc + r0; g.lda = np; g.a_kc = 0; g.B = Sb; g.ldb = np; g.b_kc = 0;
                        g.C = w.Sig + r0; g.ldc = np; g.M = EPB; g.N = (int)r0; g.K = EPB;
                        g.alpha = -1.0; g.beta = 1.0; g.tile = 64; g.flops = 2.0 * (double)EPB * r0 * EPB;
                        EP_TRY(gemm_prof(c, PC_GEMM_INNER, g, sb));
                    }
                    if (!last && r1 < np) {
                        GemmArgs g{};                   // rows >= r1: lower trapezoid
                        g.A = w.Sc + r1; g.lda = np; g.a_kc = 0; g.B = Sb; g.ldb = np; g.b_kc = 0;
                        g.C = w.Sig + r1; g.ldc = np; g.M = (int)(np - r1); g.N = (int)np; g.K = EPB;
                        g.alpha = -1.0; g.beta = 1.0; g.tile = 128; g.tri = 1; g.tri_off = (int)r1;
                        g.flops = (double)EPB * ((double)np * np - (double)r1 * r1);
                        EP_TRY(gemm_prof(c, PC_GEMM_INNER, g, sb));
                        hipLaunchKernelGGL(ep_mu_strip_kernel, dim3((unsigned)((np - r1 + 63) / 64)), dim3(256), 0, sb, Sb, np, np, r1, gb,
                                           w.mu_d, 0L, 0L);
                    }
                }
            }
            HIP_TRY(hipStreamWaitEvent(st, c->ep_ev[1], 0));
            if (hipGetLastError() != hipSuccess) return PGP_ERR_HIP;
            chain_guard.armed = false;                 // everything of this sweep is queued: the ordinary joins below take over
            if (ep_timing) {
                long long sp[16];
                HIP_TRY(hipMemcpyAsync(sp, w.gb + 2 * EPB, sizeof(sp), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                fprintf(stderr, "[ep] chain(5) stamps (s_memtime ticks): load %lld  loop %lld (first half %lld)  epilogue %lld\n", sp[1] - sp[0], sp[2] - sp[1], sp[4] - sp[1], sp[3] - sp[2]);
                fprintf(stderr, "[ep]   spins: chain waiting for the update waves %lld, update wave 1 waiting for its peers %lld, for the chain %lld\n", sp[4] - sp[1], sp[5], sp[6]);
            }
        } else
        for (long i = 0; i < n; ++i) {
            hipLaunchKernelGGL(ep_site_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, w.Sig, np, np, i,
                               w.mu_d, w.m_d, c->y_dev, w.ttau_d, w.tnu_d, w.sbuf, w.coef);
            hipLaunchKernelGGL(ep_rank1_mu_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, w.Sig, np, np,
                               w.sbuf, w.coef, w.tnu_d, w.mu_d);
        }
        auto sites_to_host = [&]() -> int {
            HIP_TRY(hipMemcpyAsync(ttau.data(), w.ttau_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(tnu.data(), w.tnu_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
            return PGP_OK;
        };
        unsigned eflags[4] = {0u, 0u, 0u, 0u};          // EPF_ERR: a device-side wait of the block sweep gave up
        if (track) {
            // nlZ (inf.py:184-188) from the carried state: the per-site terms on diag Sigma and mu as the sweep left them, log det B
            // from the sites' determinant factors.  What the host needs to decide about the next sweep -- [log det factors of the
            // blocks | counters | the per-site partial sums] -- sits in ONE device buffer and comes back in ONE copy into pinned
            // memory (an asynchronous copy into pageable memory is staged and synchronised by the runtime: five of them per sweep
            // were most of the 0.25 ms between two sweeps); the site parameters themselves are fetched when the sweeps are over
            const long nbl = (n + EPB - 1) / EPB, nbt = (n + 255) / 256;
            double* const part_d = w.ldb + np / EPB + 3;
            const double* const res_h = c->res_host + (w.ldb - c->res_dev);
            EP_TRY(gather_strided_launch(w.Sig, np + 1, np, w.diag_d, st));
            hipLaunchKernelGGL(ep_site_terms_kernel, dim3((unsigned)nbt), dim3(256), 0, st, n, c->y_dev, w.m_d, w.mu_d, w.diag_d, 0.0,
                               w.ttau_d, w.tnu_d, 1, part_d, (double*)nullptr);
            if (merge12 && sweep == 1 && 2 * res_slot <= np) {
                HIP_TRY(hipMemcpyAsync(w.ldb + res_slot, w.ldb, (size_t)res_len * sizeof(double), hipMemcpyDeviceToDevice, st));
                deferred = true;
                continue;                              // sweep 2 follows at once; the gate stays exclusive
            }
            HIP_TRY(hipMemcpyAsync((void*)res_h, w.ldb, (size_t)(deferred ? res_slot + res_len : res_len) * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            sweep_excl.release();
            memcpy(eflags, res_h + np / EPB + 1, sizeof(eflags));
            if (eflags[EPF_ERR]) return ep_wait_failed(eflags, w.chain_total, w.prep_total, w.strip_total, __LINE__);
            auto nlz_from = [&](const double* r) -> double {      // inf.py:184-188 from one slot; half_logdet accumulates over the sweeps
                const double* const ph = r + np / EPB + 3;
                for (long b = 0; b < nbl; ++b) half_logdet += 0.5 * r[b];
                double slZ = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0, t6 = 0.0;
                for (long b = 0; b < nbt; ++b) { slZ += ph[5 * b]; t3 += ph[5 * b + 1]; t4 += ph[5 * b + 2]; t5 += ph[5 * b + 3]; t6 += ph[5 * b + 4]; }
                return half_logdet - slZ - 0.5 * t3 - 0.5 * t4 + 0.5 * t5 - 0.5 * t6;
            };
            if (deferred) {                            // sweep 1 first: its nlZ is what this sweep's convergence test compares with
                const double nlZ1 = nlz_from(res_h + res_slot);
                nlZ_old = std::isfinite(nlZ1) ? nlZ1 : INFINITY;
                deferred = false;
            }
            nlZ = nlz_from(res_h);
            if (!std::isfinite(nlZ)) {                 // let the rebuild say what is wrong (first bad pivot)
                EP_TRY(sites_to_host());
                HIP_TRY(hipStreamSynchronize(st));
                rc = ep_compute_params(c, w, y, m, ttau, tnu, &nlZ, mu, dsig, &half_logdet);
                if (rc != PGP_OK) return rc;
            }
            stamp("sweep done, nlZ from the carried state", 1);
            continue;
        }
        EP_TRY(sites_to_host());
        if (c->ep_block) HIP_TRY(hipMemcpyAsync(eflags, w.flags, sizeof(eflags), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        sweep_excl.release();
        if (eflags[EPF_ERR]) return ep_wait_failed(eflags, w.chain_total, w.prep_total, w.strip_total, __LINE__);
        stamp("sweep done (synced)", 1);
        rc = ep_compute_params(c, w, y, m, ttau, tnu, &nlZ, mu, dsig);                // inf.py:772
        HIP_TRY(hipStreamSynchronize(st));
        stamp("params recomputed", 2);
        if (rc != PGP_OK) return rc;
    }
    if (track && sweep > 0) {
        HIP_TRY(hipMemcpyAsync(ttau.data(), w.ttau_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(tnu.data(), w.tnu_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (c->ep_final_rebuild) {
            rc = ep_compute_params(c, w, y, m, ttau, tnu, &nlZ, mu, dsig);            // the posterior rebuilt from the converged site parameters
            HIP_TRY(hipStreamSynchronize(st));
            stamp("params rebuilt from the converged sites", 2);
        } else {
            // Sigma, mu, log det B and with them nlZ are CURRENT (carried by exact identities; nlZ above is the value the
            // convergence test just used): alpha, the gradients and the mean derivatives follow from them as they do after a
            // rebuild.  Only post.L is missing: one plain Cholesky of I + sW sW' o K (N^3 / 3 flops instead of the 8 N^3 / 3 of
            // _epComputeParams).  The carried Sigma is current in its lower triangle: mirror it for the gradient's kernels.
            HIP_TRY(hipMemcpyAsync(mu.data(), w.mu_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
            if (c->ep_sym) hipLaunchKernelGGL(ep_mirror_kernel, dim3((unsigned)(np / 64), (unsigned)(np / 64)), dim3(256), 0, st, w.Sig, np);
            rc = ep_factor_only(c, w, ttau);
            stamp("factor of the converged sites", 2);
        }
        if (rc != PGP_OK) return rc;
    }
    if (sweeps_out) *sweeps_out = sweep;
    memcpy(ttau_io, ttau.data(), n * sizeof(double));
    memcpy(tnu_io, tnu.data(), n * sizeof(double));
    // ---- alpha = tnu - sW o B^-1 (sW o K tnu)   (inf.py:777) ----------------------------------------------
    std::vector<double> sW(np, 0.0), b(n), alpha(n);
    for (long i = 0; i < n; ++i) sW[i] = sqrt(ttau[i]);
    if (c->ep_alpha_direct) {
        // The same vector without any solve: Sigma = (K^-1 + S)^-1, S = diag(ttau), and mu = Sigma tnu (inf.py:183), so
        // K^-1 mu = tnu - S mu; and K alpha = mu for the reference's alpha (Sigma tnu = K (tnu - sW B^-1 sW K tnu), because
        // sW K sW = B - I).  mu is the one _epComputeParams just produced from the final site parameters.
        for (long i = 0; i < n; ++i) alpha[i] = tnu[i] - ttau[i] * mu[i];
    } else {
    EP_TRY(col_dot_full_launch(w.Kd, np, np, np, w.tnu_d, nullptr, w.tmp_d, st));
    HIP_TRY(hipMemcpyAsync(b.data(), w.tmp_d, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (long i = 0; i < n; ++i) b[i] *= sW[i];
    HIP_TRY(hipMemsetAsync(w.rhs, 0, (size_t)128 * np * sizeof(double), st));
    HIP_TRY(hipMemcpyAsync(w.rhs, b.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    // (the fused parameter recomputations and the factor-only last step do not produce the leaf inverses of the blocked solve)
    if (c->ep_fused || (track && sweep > 0 && !c->ep_final_rebuild))
        EP_TRY(leaf_inv_launch(w.F, ldf, w.Wd, 128, 128L * 128L, (int)(np / 128), st));
    EP_TRY(solve_lower_multi(c, w.F, ldf, w.Wd, w.rhs, np, np, 128, false));
    EP_TRY(solve_lower_multi(c, w.F, ldf, w.Wd, w.rhs, np, np, 128, true));
    HIP_TRY(hipMemcpyAsync(b.data(), w.rhs, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (long i = 0; i < n; ++i) alpha[i] = tnu[i] - sW[i] * b[i];
    }
    if (alpha_out) memcpy(alpha_out, alpha.data(), n * sizeof(double));
    if (sW_out) memcpy(sW_out, sW.data(), n * sizeof(double));
    if (nlZ_out) *nlZ_out = nlZ;
    HIP_TRY(hipMemcpyAsync(w.s_d, sW.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
    stamp("alpha", 3);
    // ---- derivatives (inf.py:780-803) ------------------------------------------------------------------------
    if (want >= 3 && dnlZ_out) {
        HIP_TRY(hipMemsetAsync(c->alpha_dev, 0, np * sizeof(double), st));
        HIP_TRY(hipMemcpyAsync(c->alpha_dev, alpha.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
        // F = alpha alpha' - sW sW' o B^-1 ; dnlZ.cov[j] = -sum(F o dK_j)/2 = sum((sW sW' o B^-1 - alpha alpha') o dK_j)/2
        if (dense) {
            // R = sW sW' o B^-1 stays in the workspace: the caller hands the derivative matrices in one at a time
            hipLaunchKernelGGL(ep_r_from_sigma_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, w.Sig, np, np,
                               w.ttau_d, c->Binv, np);
        } else if (c->ep_fused == 1 && w.Ed && !(track && sweep > 0 && !c->ep_final_rebuild)) {
            // Ed holds diag(sW) L^-T of the final parameters (not after a factor-only last step: Sigma is what is current then): (diag(sW) E)(diag(sW) E)' = sW sW' o B^-1 in one product
            EP_TRY(eet_lower(c, w.Ed, np, c->Binv, np, np));
            EP_TRY(hadamard_reduce_launch(c->XsT, np, n, np, c->dpad, cp, ncov, 1.0, c->Binv, np, c->alpha_dev, c->partial,
                                          c->scal + 8, st, nullptr));
        } else if (c->ep_r_direct || c->ep_fused == 1) {
            hipLaunchKernelGGL(ep_r_from_sigma_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)std::min<long>(np, 65535)), dim3(256), 0, st, w.Sig, np, np,
                               w.ttau_d, c->Binv, np);
            EP_TRY(hadamard_reduce_launch(c->XsT, np, n, np, c->dpad, cp, ncov, 1.0, c->Binv, np, c->alpha_dev, c->partial,
                                          c->scal + 8, st, nullptr));
        } else {
            EP_TRY(trtri_lower(c, w.F, ldf, c->W, np, c->T, np));
            EP_TRY(lauum_lower(c, c->W, np, c->Binv, np, np));
            EP_TRY(hadamard_reduce_launch(c->XsT, np, n, np, c->dpad, cp, ncov, 1.0, c->Binv, np, c->alpha_dev, c->partial,
                                          c->scal + 8, st, w.s_d));
        }
        std::vector<double> g(ncov + 1, 0.0);
        if (!dense) HIP_TRY(hipMemcpyAsync(g.data(), c->scal + 8, (ncov + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (nmean > 0) {                              // d lZ_j / d mu on the device (inf.py:788-790: no m term in nu_n), dot with dm on the host
            std::vector<double> dl(n);
            hipLaunchKernelGGL(ep_site_terms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, c->y_dev, w.m_d, w.mu_d,
                               w.diag_d, 0.0, w.ttau_d, w.tnu_d, 0, w.tmp_d, w.sbuf);
            HIP_TRY(hipMemcpyAsync(dl.data(), w.sbuf, n * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int i = 0; i < nmean; ++i) {
                double sacc = 0.0;
                for (long j = 0; j < n; ++j) sacc += dl[j] * dm[(long)i * n + j];
                dnlZ_out[i] = -sacc;
            }
        }
        for (int h = 0; h < ncov; ++h) dnlZ_out[nmean + h] = 0.5 * g[h];
        dnlZ_out[nmean + ncov] = 0.0;                                                   // lik.Erf has no hyper
    }
    if (c->prof) prof_collect(c);
    HIP_TRY(hipStreamSynchronize(st));
    stamp("gradients", 3);
    c->last_ms[PGP_STAGE_ASSEMBLE] = ph_ms[0]; c->last_ms[PGP_STAGE_SOLVE] = ph_ms[1]; c->last_ms[PGP_STAGE_POTRF] = ph_ms[2];
    c->last_ms[PGP_STAGE_GRAD] = ph_ms[3]; c->last_ms[PGP_STAGE_TRTRI] = 0.0; c->last_ms[PGP_STAGE_LAUUM] = 0.0;
    c->last_ms[PGP_STAGE_TOTAL] = ph_ms[0] + ph_ms[1] + ph_ms[2] + ph_ms[3];
    if (factor_out) {
        FactorHandleGuard hg(c, new pgp_factor());
        pgp_factor* f = hg.f;
        f->n = n; f->np = np; f->ldf = ldf; f->F = fguard.release(); f->dpad = dense ? 0 : c->dpad; f->d = dense ? 0 : (int)d; f->kss = kss;
        if (!dense) f->cs = cp;
        f->sn2 = 1.0; f->sw = 1.0; f->Wd = nullptr; f->XsT = nullptr;
        CHK(spool_take(c, np * sizeof(double), (void**)&f->alpha));
        HIP_TRY(hipMemsetAsync(f->alpha, 0, np * sizeof(double), st));
        HIP_TRY(hipMemcpyAsync(f->alpha, alpha.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
        CHK(spool_take(c, np * sizeof(double), (void**)&f->sWv));
        HIP_TRY(hipMemcpyAsync(f->sWv, sW.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
        if (!dense) {
            CHK(spool_take(c, (size_t)c->dpad * np * sizeof(double), (void**)&f->XsT));
            HIP_TRY(hipMemcpyAsync(f->XsT, c->XsT, (size_t)c->dpad * np * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        *factor_out = hg.release();
    } else {
        HIP_TRY(hipStreamSynchronize(st));
        fguard.scrub = false;                         // a finished factor honours the pool contract (zeros above the diagonal)
    }
    if (dense && want >= 3) { c->dense_ready = true; c->dense_n = n; }
    return PGP_OK;
#undef EP_TRY
}

extern "C" int pgp_ep_fit(pgp_ctx* c, int kind, const double* covhyp, int ncov, int para, int flags, const double* mvec,
                          const double* dm, int nmean, int want, int warm, double* ttau_io, double* tnu_io,
                          double* alpha_out, double* sW_out, double* nlZ_out, double* dnlZ_out, int* sweeps_out,
                          pgp_factor** factor_out) {
    if (!covhyp) return -3;
    return ep_fit_core(c, nullptr, kind, covhyp, ncov, para, flags, mvec, dm, nmean, want, warm, ttau_io, tnu_io, alpha_out, sW_out,
                       nlZ_out, dnlZ_out, sweeps_out, factor_out);
}

// EP.evaluate from a CALLER-BUILT covariance matrix K (n x n, symmetric, row-major host; n and y as set by pgp_set_data): the
// covariance trees that are not device programs.  dnlZ_mean_out (nmean + 1 entries: the mean gradients, then 0 for lik.Erf);
// the covariance gradients follow from pgp_dense_grad_term(ctx, dK_h, n, 0.0, &g) calls made directly afterwards
// (want = 3): g = 1/2 sum (sW sW' o B^-1 - alpha alpha') o dK_h  (Core/inf.py:780-786).
extern "C" int pgp_ep_fit_dense(pgp_ctx* c, const double* K, const double* mvec, const double* dm, int nmean, int want, int warm,
                                double* ttau_io, double* tnu_io, double* alpha_out, double* sW_out, double* nlZ_out,
                                double* dnlZ_mean_out, int* sweeps_out, pgp_factor** factor_out) {
    if (!K) return -2;
    return ep_fit_core(c, K, 0, nullptr, 0, 0, 0, mvec, dm, nmean, want, warm, ttau_io, tnu_io, alpha_out, sW_out, nlZ_out,
                       dnlZ_mean_out, sweeps_out, factor_out);
}
