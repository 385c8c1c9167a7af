// Shared device code: the 64x64 pairwise squared-distance tile and the covariance scalar maps.
// Used by assemble.hip (kernel-matrix construction) and grad.hip (Hadamard-reduce that recomputes
// K tile-wise instead of re-reading it).
//
// Inputs are the *scaled, transposed* coordinates XsT[k*ldp + p] = x[p][k] * scale_k (k-major, so a
// 64-point slab of one coordinate is one contiguous 512-byte run; coalesced loads, LDS-staged 16
// coordinates at a time).  The distance is accumulated in the difference form sum_k (a_k - b_k)^2,
// like scipy's cdist('sqeuclidean') that the reference calls (Core/cov.py:804) -- no |a|^2+|b|^2-2ab
// cancellation, K(x,x) = sf2 exactly.
#pragma once
#include <vector>

#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

struct CovParams {
    int kind;        // PGP_COV_*
    int der;         // -1 value, >=0 derivative index
    int md;          // Matern d (1,3,5,7)
    int ref_der;     // Matern: reproduce the reference's derivative-of-K quirk (Core/cov.py:1173-1177)
    int D;           // input dimension (ARD: der < D selects a length-scale)
    int train;       // 1: x against itself ('train'), 0: 'cross', 2: 'self_test' -- Noise and Const depend on it
    double sf2;      // exp(2 log sf)   (1 for RBFunit and Gabor; exp(log sf) for Const, cov.py:951)
    double alpha;    // RQ / RQard shape parameter exp(log alpha)
    int ppv;         // PiecePoly degree v (0..3)
    double ppj;      // PiecePoly j = floor(D/2) + v + 1
    double ga;       // Gabor: 2 pi ell / p with p = exp(2 hyp1) (cov.py:416,425); Periodic: pi / p (cov.py:1212)
    double gb;       // Periodic: 1 / ell (cov.py:1213); RQard: factor on the length-scale derivative (1, or the
                     // reference's 0 on 'train' / ell_k^4 on 'cross' with PGP_FLAG_REFERENCE_DER, cov.py:1412-1418)
};

// A Sum / Product / Scale tree over non-ARD leaves (Core/cov.py:230-328), expanded on the host into a sum of
// products:  K = sum_t coef_t * prod_{l in term t} k_l(r^2 * is2_l).  Every leaf is a function of the *raw*
// squared distance r^2 (XsT is uploaded unscaled for programs), so one distance tile serves all leaves.
constexpr int CP_MAXLEAF = 8, CP_MAXTERM = 8, CP_MAXSCALE = 8, CP_MAXARD = 64;
struct CovProgram {
    int nleaf, nterm, nscale;
    int der;                         // flat hyper index for derivative matrices (-1: value)
    int der_leaf, der_j, der_scale;  // der resolved: hyper j of leaf l, or Scale node s (other = -1)
    CovParams leaf[CP_MAXLEAF];
    double is2[CP_MAXLEAF];          // leaf's scaled squared distance = r^2 * is2
    int hyp0[CP_MAXLEAF];            // flat index of the leaf's first hyper
    int nh[CP_MAXLEAF];              // number of hypers of the leaf (<= 3)
    double coef[CP_MAXTERM];         // product of exp(scale hyper) over the Scale nodes above the term (cov.py:315)
    unsigned tl[CP_MAXTERM];         // leaves multiplied in term t (bit mask)
    unsigned ts[CP_MAXTERM];         // Scale nodes above term t (bit mask)
    int shyp[CP_MAXSCALE];           // flat hyper index of each Scale node
    int ard_leaf;                    // index of the first ARD leaf (RBFard / RQard), -1: none.  Its distance is the
    double ardw[CP_MAXARD];          // weighted sum_k ardw[k] (x_k - z_k)^2, ardw[k] = 1 / ell_k^2, accumulated beside r^2
    int ard_leaf2;                   // a second ARD leaf with its own weights (-1: none): at most two per program
    double ardw2[CP_MAXARD];
};

constexpr int ST = 64;      // tile edge
constexpr int SKC = 16;     // coordinates staged per step

// thread t of 256 owns rows r = 4*(t/16) + a (a<4) and columns c = 2*(t%16) + e + 32*b (e<2, b<2);
// s[a][2*b + e] accumulates |x_r - z_c|^2.
__device__ __forceinline__ void sqdist_tile(const double* __restrict__ XrT, long ldr, long r0,
                                            const double* __restrict__ XcT, long ldc, long c0, int dpad,
                                            double* __restrict__ sm /* 2*SKC*ST doubles */, double (&s)[4][4]) {
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double* xr = sm;
    double* xc = sm + SKC * ST;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
    for (int k0 = 0; k0 < dpad; k0 += SKC) {
        // 2 * 16 * 64 doubles = 1024 double2; 4 per thread
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = t + p * 256;            // 0..511 : k = v / 32, pair = v % 32
            const int k = v >> 5, pr = v & 31;
            *(double2_t*)(xr + k * ST + 2 * pr) = *(const double2_t*)(XrT + (long)(k0 + k) * ldr + r0 + 2 * pr);
            *(double2_t*)(xc + k * ST + 2 * pr) = *(const double2_t*)(XcT + (long)(k0 + k) * ldc + c0 + 2 * pr);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SKC; ++k) {
            const double2_t r01 = *(const double2_t*)(xr + k * ST + 4 * tr);
            const double2_t r23 = *(const double2_t*)(xr + k * ST + 4 * tr + 2);
            const double2_t c01 = *(const double2_t*)(xc + k * ST + 2 * tc);
            const double2_t c23 = *(const double2_t*)(xc + k * ST + 2 * tc + 32);
            const double rv[4] = {r01[0], r01[1], r23[0], r23[1]};
            const double cv[4] = {c01[0], c01[1], c23[0], c23[1]};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = rv[a] - cv[b];
                    s[a][b] = fma(df, df, s[a][b]);
                }
        }
        __syncthreads();
    }
}

// the same tile with a second, per-coordinate weighted distance s1 = sum_k w[k] (a_k - b_k)^2 (ARD leaf of a program)
__device__ __forceinline__ void sqdist_tile2(const double* __restrict__ XrT, long ldr, long r0,
                                             const double* __restrict__ XcT, long ldc, long c0, int dpad,
                                             const double* __restrict__ w /* uniform, >= dpad entries */,
                                             double* __restrict__ sm, double (&s)[4][4], double (&s1)[4][4]) {
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double* xr = sm;
    double* xc = sm + SKC * ST;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { s[a][b] = 0.0; s1[a][b] = 0.0; }
    for (int k0 = 0; k0 < dpad; k0 += SKC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = t + p * 256;
            const int k = v >> 5, pr = v & 31;
            *(double2_t*)(xr + k * ST + 2 * pr) = *(const double2_t*)(XrT + (long)(k0 + k) * ldr + r0 + 2 * pr);
            *(double2_t*)(xc + k * ST + 2 * pr) = *(const double2_t*)(XcT + (long)(k0 + k) * ldc + c0 + 2 * pr);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SKC; ++k) {
            const double wk = (k0 + k) < CP_MAXARD ? w[k0 + k] : 0.0;
            const double2_t r01 = *(const double2_t*)(xr + k * ST + 4 * tr);
            const double2_t r23 = *(const double2_t*)(xr + k * ST + 4 * tr + 2);
            const double2_t c01 = *(const double2_t*)(xc + k * ST + 2 * tc);
            const double2_t c23 = *(const double2_t*)(xc + k * ST + 2 * tc + 32);
            const double rv[4] = {r01[0], r01[1], r23[0], r23[1]};
            const double cv[4] = {c01[0], c01[1], c23[0], c23[1]};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = rv[a] - cv[b];
                    const double d2 = df * df;
                    s[a][b] += d2;
                    s1[a][b] = fma(wk, d2, s1[a][b]);
                }
        }
        __syncthreads();
    }
}

// and with two weighted distances (programs with two ARD leaves)
__device__ __forceinline__ void sqdist_tile3(const double* __restrict__ XrT, long ldr, long r0,
                                             const double* __restrict__ XcT, long ldc, long c0, int dpad,
                                             const double* __restrict__ w, const double* __restrict__ w2,
                                             double* __restrict__ sm, double (&s)[4][4], double (&s1)[4][4],
                                             double (&s2)[4][4]) {
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double* xr = sm;
    double* xc = sm + SKC * ST;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { s[a][b] = 0.0; s1[a][b] = 0.0; s2[a][b] = 0.0; }
    for (int k0 = 0; k0 < dpad; k0 += SKC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = t + p * 256;
            const int k = v >> 5, pr = v & 31;
            *(double2_t*)(xr + k * ST + 2 * pr) = *(const double2_t*)(XrT + (long)(k0 + k) * ldr + r0 + 2 * pr);
            *(double2_t*)(xc + k * ST + 2 * pr) = *(const double2_t*)(XcT + (long)(k0 + k) * ldc + c0 + 2 * pr);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SKC; ++k) {
            const double wk = (k0 + k) < CP_MAXARD ? w[k0 + k] : 0.0;
            const double wk2 = (k0 + k) < CP_MAXARD ? w2[k0 + k] : 0.0;
            const double2_t r01 = *(const double2_t*)(xr + k * ST + 4 * tr);
            const double2_t r23 = *(const double2_t*)(xr + k * ST + 4 * tr + 2);
            const double2_t c01 = *(const double2_t*)(xc + k * ST + 2 * tc);
            const double2_t c23 = *(const double2_t*)(xc + k * ST + 2 * tc + 32);
            const double rv[4] = {r01[0], r01[1], r23[0], r23[1]};
            const double cv[4] = {c01[0], c01[1], c23[0], c23[1]};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = rv[a] - cv[b];
                    const double d2 = df * df;
                    s[a][b] += d2;
                    s1[a][b] = fma(wk, d2, s1[a][b]);
                    s2[a][b] = fma(wk2, d2, s2[a][b]);
                }
        }
        __syncthreads();
    }
}

// s[e >> 2][e & 3] for a run-time e without dynamic register indexing (the second ARD distance stays in registers:
// three 32 KB staging arrays would not fit the 64 KB of static LDS)
__device__ __forceinline__ double sel16(const double (&s)[4][4], int e) {
    double v = s[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) v = (e == i) ? s[i >> 2][i & 3] : v;
    return v;
}
__device__ __forceinline__ void put16(double (&s)[4][4], int e, double val) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i >> 2][i & 3] = (e == i) ? val : s[i >> 2][i & 3];
}

__device__ __forceinline__ double matern_poly(int d, double t) {
    switch (d) {
        case 1: return 1.0;
        case 3: return 1.0 + t;
        case 5: return 1.0 + t + t * t / 3.0;
        default: return 1.0 + t + 2.0 * t * t / 5.0 + t * t * t / 15.0;
    }
}
__device__ __forceinline__ double matern_dpoly(int d, double t) {
    switch (d) {
        case 1: return 1.0;
        case 3: return t;
        case 5: return (t + t * t) / 3.0;
        default: return (t + 3.0 * t * t + t * t * t) / 15.0;
    }
}

// PiecePoly polynomial f(v, r, j) and "f - f'" companion (Core/cov.py:698-720)
__device__ __forceinline__ double pp_func(int v, double r, double j) {
    switch (v) {
        case 0: return 1.0;
        case 1: return 1.0 + (j + 1.0) * r;
        case 2: return 1.0 + (j + 2.0) * r + (j * j + 4.0 * j + 3.0) / 3.0 * r * r;
        default: return 1.0 + (j + 3.0) * r + (6.0 * j * j + 36.0 * j + 45.0) / 15.0 * r * r
                        + (j * j * j + 9.0 * j * j + 23.0 * j + 15.0) / 15.0 * r * r * r;
    }
}
__device__ __forceinline__ double pp_dfunc(int v, double r, double j) {
    switch (v) {
        case 0: return 0.0;
        case 1: return j + 1.0;
        case 2: return (j + 2.0) + 2.0 * (j * j + 4.0 * j + 3.0) / 3.0 * r;
        default: return (j + 3.0) + 2.0 * (6.0 * j * j + 36.0 * j + 45.0) / 15.0 * r
                        + (j * j * j + 9.0 * j * j + 23.0 * j + 15.0) / 5.0 * r * r;
    }
}
__device__ __forceinline__ double ipow(double b, int n) {      // b^n, n >= 0, 0^0 = 1 like numpy
    double r = 1.0;
    while (n > 0) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}

// exp(x) for x <= 0 (the squared-exponential map).  Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, Taylor
// polynomial of degree 13 (remainder < 5e-18 relative), 2^k applied by v_ldexp (underflows to 0 by itself): 19 VALU
// instructions against ~26 for the library exp with its overflow / underflow selects; within 1-2 ulp of it.
__device__ __forceinline__ double exp_nonpos(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                 // 1/13!
    p = fma(p, r, 2.08767569878681e-09);               // 1/12!
    p = fma(p, r, 2.505210838544172e-08);              // 1/11!
    p = fma(p, r, 2.755731922398589e-07);              // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);             // 1/9!
    p = fma(p, r, 2.48015873015873e-05);               // 1/8!
    p = fma(p, r, 1.984126984126984e-04);              // 1/7!
    p = fma(p, r, 1.3888888888888889e-03);             // 1/6!
    p = fma(p, r, 8.333333333333333e-03);              // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);             // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);             // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double kc = fmax(k, -1100.0);                // keep the int conversion in range; 2^-1100 is already 0
    return ldexp(p, (int)kc);
}

// exp by table (round 6; the compute-bound kernels -- Gram-form assembly at d = 64, the ARD gradient pass): x = (64 k + j) ln2/64 + r,
// |r| <= ln2/128, exp(x) = 2^k * T[j] * (1 + q(r)), q = expm1 by a degree-5 Taylor polynomial (remainder < 3.5e-17 relative).
// The reduction uses the "magic number" rounding: u = x * 64/ln2 + 1.5 * 2^52 holds round(x * 64/ln2) as a two's-complement integer
// in its low dword (no v_rndne / v_cvt), T comes from a 64-entry table in LDS that the caller has PRE-SCALED by `c` (sf2, or
// sf2 / sn2: the product with the prefactor costs nothing), the 2^k by v_ldexp (gradual underflow and exact zeros by itself).
// 14 VALU instructions + one ds_read_b64 against 20 for exp_nonpos + the prefactor's multiplication; <= 1 ulp of c * exp(x).
// Valid for -745 <= x <= 0 (the caller clamps: beyond, the low dword no longer holds the integer).
static __device__ const double EXP2_TAB64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
// min(max(x, -745), 0).  Plain fmin / fmax on purpose: the argument comes straight out of MFMA accumulators, and inline assembly
// that reads them is invisible to the compiler's hazard recogniser (an asm v_min_f64 right behind the last v_mfma_f64 read stale
// registers in one instantiation: the DGEMM -> VALU read needs 18 wait states that only the compiler inserts).  The price is one
// canonicalising v_max per value.
__device__ __forceinline__ double clamp_exp_arg(double x) { return fmax(fmin(x, 0.0), -745.0); }
// every thread t < 64 of the workgroup writes one entry; the caller's next barrier publishes the table
__device__ __forceinline__ void exp_tab_fill(double* tab, double c, int t) { if (t < 64) tab[t] = c * EXP2_TAB64[t]; }
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* tab) {      // = c * exp(x), -745 <= x <= 0
    const double u = fma(x, 92.33248261689366, 6755399441055744.0);                   // 64 / ln2, 1.5 * 2^52
    const double m = u - 6755399441055744.0;
    double r = fma(m, -0x1.62e42fee00000p-7, x);                                      // ln2 / 64: 32 leading bits (m * hi is exact)
    r = fma(m, -2.9815858269852933e-12, r);
    const int mi = __double2loint(u);
    const double T = tab[mi & 63];
    double p = fma(r, 8.333333333333333e-03, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    const double q = fma(p, r * r, r);
    return ldexp(fma(T, q, T), mi >> 6);
}

// covariance value k(x,z) from the scaled squared distance s; same = "row and column are the same training point"
// EXT = false leaves out the trigonometric / index-dependent kinds (7..10): the primitive hot-path kernels are
// instantiated without them (they run as one-leaf programs), which keeps their register count where it was.
template <bool EXT = false>
__device__ __forceinline__ double cov_value(const CovParams& p, double s, bool same = false) {
    if (p.kind == 4 || p.kind == 6) {     // RQ: Core/cov.py:1323, RQard :1394
        return p.sf2 *  exp_nonpos(-p.alpha * log(1.0 + 0.5 * s / p.alpha));
    }
    if (EXT && p.kind == 7) {                    // Gabor: cov.py:424-426
        return  exp_nonpos(-0.5 * s) * cos(sqrt(s) * p.ga);
    }
    if (EXT && p.kind == 8) {             // Periodic: cov.py:1212-1216 (s = raw |x-z|^2, 1-d inputs)
        const double R = sin(sqrt(s) * p.ga) * p.gb;
        return p.sf2 *  exp_nonpos(-2.0 * R * R);
    }
    if (EXT && p.kind == 9) {             // Noise: cov.py:1269-1280 (identity on 'train', |x-z|^2 < 1e-9 on 'cross', 0 on 'self_test')
        const bool one = p.train == 1 ? same : (p.train == 0 ? s < 1e-9 : false);
        return one ? p.sf2 : 0.0;
    }
    if (EXT && p.kind == 10) {            // Const: cov.py:951-962 (1e-10 jitter on the training diagonal)
        return p.sf2 + ((p.train == 1 && same) ? 1e-10 : 0.0);
    }
    if (p.kind == 5) {                    // PiecePoly: Core/cov.py:746 (compact support: 0 beyond r = 1)
        const double r = sqrt(s);
        const double pm = fmax(1.0 - r, 0.0);
        return p.sf2 * pp_func(p.ppv, r, p.ppj) * ipow(pm, (int)p.ppj + p.ppv);
    }
    if (p.kind == 2) {                    // Matern: t = sqrt(d) |x-z| / ell (scale folded into XsT)
        const double t = sqrt(s);
        return p.sf2 * matern_poly(p.md, t) *  exp_nonpos(-t);
    }
    return p.sf2 * exp_nonpos(-0.5 * s);  // RBF / RBFard
}

// derivative w.r.t. hyper p.der; dk2 = scaled squared difference in coordinate p.der (ARD only)
template <bool EXT = false>
__device__ __forceinline__ double cov_deriv(const CovParams& p, double s, double dk2, bool same = false) {
    if (p.kind == 3) return  exp_nonpos(-0.5 * s) * s;        // RBFunit: Core/cov.py:865
    if (p.kind == 6) {                    // RQard: cov.py:1412-1425
        const double Kp = 1.0 + 0.5 * s / p.alpha;
        const double lk = log(Kp);
        if (p.der < p.D) return p.sf2 *  exp_nonpos((-p.alpha - 1.0) * lk) * dk2 * p.gb;   // gb = 1 (see make_leaf for compat)
        if (p.der == p.D) return 2.0 * p.sf2 *  exp_nonpos(-p.alpha * lk);
        return p.sf2 *  exp_nonpos(-p.alpha * lk) * (0.5 * s / Kp - p.alpha * lk);
    }
    if (EXT && p.kind == 7) {             // Gabor: cov.py:440-448 (as the reference returns them)
        const double dp = sqrt(s) * p.ga;
        const double K =  exp_nonpos(-0.5 * s) * cos(dp);
        return p.der == 0 ? dp * K : tan(dp) * dp * K;
    }
    if (EXT && p.kind == 8) {             // Periodic: cov.py:1236-1249
        const double A = sqrt(s) * p.ga;
        const double R = sin(A) * p.gb;
        const double e = p.sf2 *  exp_nonpos(-2.0 * R * R);
        if (p.der == 0) return 4.0 * e * R * R;
        if (p.der == 1) return 4.0 * p.gb * e * R * cos(A) * A;
        return 2.0 * e;
    }
    if (EXT && p.kind == 9) {             // Noise: cov.py:1296-1297
        const bool one = p.train == 1 ? same : (p.train == 0 ? s < 1e-9 : false);
        return one ? 2.0 * p.sf2 : 0.0;
    }
    if (EXT && p.kind == 10) return 2.0 * p.sf2; // Const: cov.py:978-979 (no jitter in the derivative)
    if (p.kind == 4) {                    // RQ: Core/cov.py:1337-1345
        const double Kp = 1.0 + 0.5 * s / p.alpha;
        const double lk = log(Kp);
        if (p.der == 0) return p.sf2 *  exp_nonpos((-p.alpha - 1.0) * lk) * s;
        if (p.der == 1) return 2.0 * p.sf2 *  exp_nonpos(-p.alpha * lk);
        return p.sf2 *  exp_nonpos(-p.alpha * lk) * (0.5 * s / Kp - p.alpha * lk);
    }
    if (p.kind == 5) {                    // PiecePoly: Core/cov.py:774-780
        if (p.der == 2) return 0.0;
        const double r = sqrt(s);
        const double pm = fmax(1.0 - r, 0.0);
        const int e = (int)p.ppj + p.ppv;
        if (p.der == 1) return 2.0 * p.sf2 * pp_func(p.ppv, r, p.ppj) * ipow(pm, e);
        return p.sf2 * ipow(pm, e - 1) * r * ((double)e * pp_func(p.ppv, r, p.ppj) - pm * pp_dfunc(p.ppv, r, p.ppj));
    }
    if (p.kind == 0) {                    // RBF: Core/cov.py:823-825
        const double K = p.sf2 *  exp_nonpos(-0.5 * s);
        return p.der == 0 ? K * s : 2.0 * K;
    }
    if (p.kind == 1) {                    // RBFard: Core/cov.py:922-936
        const double K = p.sf2 *  exp_nonpos(-0.5 * s);
        return p.der < p.D ? K * dk2 : 2.0 * K;
    }
    const double t = sqrt(s);             // Matern
    if (p.der == 2) return 0.0;
    if (p.ref_der) {                      // Core/cov.py:1173-1177: dmfunc / mfunc applied to K, not t
        const double K = p.sf2 * matern_poly(p.md, t) *  exp_nonpos(-t);
        return p.der == 0 ? p.sf2 * matern_dpoly(p.md, K) * K *  exp_nonpos(-K)
                          : 2.0 * p.sf2 * matern_poly(p.md, K) *  exp_nonpos(-K);
    }
    return p.der == 0 ? p.sf2 * matern_dpoly(p.md, t) * t *  exp_nonpos(-t)
                      : 2.0 * p.sf2 * matern_poly(p.md, t) *  exp_nonpos(-t);
}

// every derivative (up to three hypers) of the non-ARD kernels, transcendental functions evaluated once
template <bool EXT = false>
__device__ __forceinline__ void cov_deriv_all(const CovParams& p, double s, double& d0, double& d1, double& d2,
                                              bool same = false) {
    d2 = 0.0;
    if (p.kind == 3) { d0 =  exp_nonpos(-0.5 * s) * s; d1 = 0.0; return; }
    if (EXT && p.kind == 7) {
        const double dp = sqrt(s) * p.ga;
        const double K =  exp_nonpos(-0.5 * s) * cos(dp);
        d0 = dp * K;
        d1 = tan(dp) * dp * K;
        return;
    }
    if (EXT && p.kind == 8) {
        const double A = sqrt(s) * p.ga;
        const double R = sin(A) * p.gb;
        const double e = p.sf2 *  exp_nonpos(-2.0 * R * R);
        d0 = 4.0 * e * R * R;
        d1 = 4.0 * p.gb * e * R * cos(A) * A;
        d2 = 2.0 * e;
        return;
    }
    if (EXT && p.kind == 9) {
        const bool one = p.train == 1 ? same : (p.train == 0 ? s < 1e-9 : false);
        d0 = one ? 2.0 * p.sf2 : 0.0; d1 = 0.0;
        return;
    }
    if (EXT && p.kind == 10) { d0 = 2.0 * p.sf2; d1 = 0.0; return; }
    if (p.kind == 4) {
        const double Kp = 1.0 + 0.5 * s / p.alpha;
        const double lk = log(Kp);
        const double Ka = p.sf2 *  exp_nonpos(-p.alpha * lk);
        d0 = Ka / Kp * s;
        d1 = 2.0 * Ka;
        d2 = Ka * (0.5 * s / Kp - p.alpha * lk);
        return;
    }
    if (p.kind == 5) {
        const double r = sqrt(s);
        const double pm = fmax(1.0 - r, 0.0);
        const int e = (int)p.ppj + p.ppv;
        const double f = pp_func(p.ppv, r, p.ppj);
        const double pe1 = ipow(pm, e - 1);
        d0 = p.sf2 * pe1 * r * ((double)e * f - pm * pp_dfunc(p.ppv, r, p.ppj));
        d1 = 2.0 * p.sf2 * f * pe1 * pm;
        return;
    }
    if (p.kind == 0) {
        const double K = p.sf2 *  exp_nonpos(-0.5 * s);
        d0 = K * s;
        d1 = 2.0 * K;
        return;
    }
    const double t = sqrt(s);                          // Matern
    const double e =  exp_nonpos(-t);
    if (p.ref_der) {
        const double K = p.sf2 * matern_poly(p.md, t) * e;
        const double eK =  exp_nonpos(-K);
        d0 = p.sf2 * matern_dpoly(p.md, K) * K * eK;
        d1 = 2.0 * p.sf2 * matern_poly(p.md, K) * eK;
    } else {
        d0 = p.sf2 * matern_dpoly(p.md, t) * t * e;
        d1 = 2.0 * p.sf2 * matern_poly(p.md, t) * e;
    }
}

// host-side description of the covariance function of one call: a primitive functor or a program
struct CovSpec {
    bool prog = false;
    CovParams cp;
    CovProgram pg;
    std::vector<double> scale;      // per-coordinate scale folded into XsT (all ones for programs)
    int ncov = 0;                   // number of hyperparameters (= gradient entries)
    int nder = 0;                   // getDerMatrix accepts der in [0, nder)
    double ell4 = 1.0;              // RQard reference-compat 'cross' derivative factor ell_der^4
    bool ard_grad_diff = false;     // plain ARD kinds: the gradient pass must weight with the difference-form K (make_spec: the scaled,
                                    // centred points of the resident x are too far out for the Gram form's eps |a|^2 -- grad.hip)
    // launch tuning of the assembly kernels, copied from the context that made the spec (options "asm_grid", "asm_nt")
    int asm_grid = 4096;            // persistent workgroups (4 resident per CU, the rest queue: dynamic balance)
    int asm_nt = 0;                 // non-temporal stores: 0 never (default), 1 always, -1 for outputs >= 1 GB
    int gram_fast = 2;              // Gram-form assembly: 0 the general kernel, 1 the restructured one, 2 + its four-workgroups-per-CU form at d = 64
    int gram_grid = 32768;          // Gram-form assembly: persistent workgroups per launch (option "gram_grid")
    double sf2() const { return cp.sf2; }
};

// ---- composite programs ----------------------------------------------------------------------------
__device__ __forceinline__ bool cov_is_ard(const CovParams& p) { return p.kind == 1 || p.kind == 6; }

// term products T_t = coef_t prod_{l in t} v_l
__device__ __forceinline__ void prog_terms(const CovProgram& P, const double (&v)[CP_MAXLEAF], double (&T)[CP_MAXTERM]) {
#pragma unroll
    for (int t = 0; t < CP_MAXTERM; ++t) {
        double pr = 0.0;
        if (t < P.nterm) {
            pr = P.coef[t];
#pragma unroll
            for (int l = 0; l < CP_MAXLEAF; ++l)
                if ((P.tl[t] >> l) & 1u) pr *= v[l];
        }
        T[t] = pr;
    }
}

// dK / d(leaf l's value) = sum_{t containing l} coef_t prod_{l' in t, l' != l} v_l'
__device__ __forceinline__ double prog_leaf_weight(const CovProgram& P, const double (&v)[CP_MAXLEAF], int l) {
    double w = 0.0;
#pragma unroll
    for (int t = 0; t < CP_MAXTERM; ++t) {
        if (t < P.nterm && ((P.tl[t] >> l) & 1u)) {
            double pr = P.coef[t];
#pragma unroll
            for (int l2 = 0; l2 < CP_MAXLEAF; ++l2)
                if (l2 != l && ((P.tl[t] >> l2) & 1u)) pr *= v[l2];
            w += pr;
        }
    }
    return w;
}

// scaled squared distance of leaf l: the shared raw distance times the leaf's 1/ell^2, or the ARD-weighted one
__device__ __forceinline__ double prog_leaf_dist(const CovProgram& P, int l, double r2, double s1, double s2 = 0.0) {
    return l == P.ard_leaf ? s1 : (l == P.ard_leaf2 ? s2 : r2 * P.is2[l]);
}

__device__ __forceinline__ double prog_value(const CovProgram& P, double r2, bool same, double s1 = 0.0, double s2 = 0.0) {
    double v[CP_MAXLEAF], T[CP_MAXTERM];
#pragma unroll
    for (int l = 0; l < CP_MAXLEAF; ++l) v[l] = l < P.nleaf ? cov_value<true>(P.leaf[l], prog_leaf_dist(P, l, r2, s1, s2), same) : 1.0;
    prog_terms(P, v, T);
    double K = 0.0;
#pragma unroll
    for (int t = 0; t < CP_MAXTERM; ++t) K += T[t];
    return K;
}

// derivative matrix entry for the flat hyper index P.der (Product :246-256, Sum :281-291, Scale :320-328)
// dk2: ARD-weighted squared difference in coordinate der_j (only when the derivative is w.r.t. an ARD length-scale)
__device__ __forceinline__ double prog_deriv(const CovProgram& P, double r2, bool same, double s1 = 0.0, double dk2 = 0.0,
                                             double s2 = 0.0) {
    double v[CP_MAXLEAF], T[CP_MAXTERM];
#pragma unroll
    for (int l = 0; l < CP_MAXLEAF; ++l) v[l] = l < P.nleaf ? cov_value<true>(P.leaf[l], prog_leaf_dist(P, l, r2, s1, s2), same) : 1.0;
    if (P.der_scale >= 0) {               // 2 * exp(h) * child, through whatever sits above the Scale node
        prog_terms(P, v, T);
        double K = 0.0;
#pragma unroll
        for (int t = 0; t < CP_MAXTERM; ++t)
            if ((P.ts[t] >> P.der_scale) & 1u) K += T[t];
        return 2.0 * K;
    }
    double out = 0.0;
#pragma unroll
    for (int l = 0; l < CP_MAXLEAF; ++l) {
        if (l == P.der_leaf) {
            CovParams lp = P.leaf[l];
            lp.der = P.der_j;
            out = prog_leaf_weight(P, v, l) * cov_deriv<true>(lp, prog_leaf_dist(P, l, r2, s1, s2), dk2, same);
        }
    }
    return out;
}

// one element seen by the tile kernels, primitive functor or program
__device__ __forceinline__ double cov_elem(const CovParams& p, double s, double dk2, bool same) {
    return p.der < 0 ? cov_value(p, s, same) : cov_deriv(p, s, dk2, same);
}
__device__ __forceinline__ double cov_elem(const CovProgram& P, double s, double dk2, bool same, double s1 = 0.0,
                                           double s2 = 0.0) {
    return P.der < 0 ? prog_value(P, s, same, s1, s2) : prog_deriv(P, s, same, s1, dk2, s2);
}
__device__ __forceinline__ int cov_ard_der(const CovParams& p) { return (cov_is_ard(p) && p.der >= 0 && p.der < p.D) ? p.der : -1; }
__device__ __forceinline__ int cov_ard_der(const CovProgram& P) {
    const bool ard = P.der_leaf >= 0 && (P.der_leaf == P.ard_leaf || P.der_leaf == P.ard_leaf2);
    return (ard && P.der_j >= 0 && P.der_j < P.leaf[P.der_leaf & 7].D) ? P.der_j : -1;
}
// 1 / ell_k^2 of the ARD leaf the derivative is taken in
__device__ __forceinline__ const double* cov_ard_der_w(const CovProgram& P) { return P.der_leaf == P.ard_leaf2 ? P.ardw2 : P.ardw; }
