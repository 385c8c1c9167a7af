// Gradient Hadamard-reduce and the O(N)/O(N^2) HBM-bound helpers of Exact.evaluate
// (reference: Core/inf.py:370-381).
//
//   hadamard_reduce_kernel   one pass over B^-1 (upper row-major view of the column-major lower
//        result of W^T W) that recomputes K tile-wise from the LDS-staged coordinates and accumulates,
//        for Q = B^-1/sn2 - alpha alpha^T, the sums  sum_ij Q_ij dK_h,ij  for EVERY covariance hyper at
//        once plus sn2*tr(Q) -- the reference re-builds a full N x N derivative matrix per hyper
//        (65 times for SEard d=64).  Algorithmic bytes: 8 N(N+1)/2 (B^-1 triangle) + 8 N d.
//        Deterministic: per-block partials, fixed-order final reduction (no float atomics).
//   col_dot_kernel           alpha' = W^T z           (W = L^-1 column-major lower)
//   logdet_ztz_kernel        sum log L_jj, z'z, alpha'alpha
#include "kernels.h"
#include <atomic>

#include "sqdist_tile.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHK_RC(x) do { int rc__ = (x); if (rc__ != PGP_OK) return rc__; } while (0)

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block of 256 threads: reduce `v`, result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* red /* 4 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// four block sums with ONE barrier pair (block_sum x 4 costs eight barriers: the 64 x 64-tile reduce kernels do one tile per
// workgroup and were bound by them); fixed order: waves 0..3
__device__ __forceinline__ void block_sum4(double (&v)[4], double* red16 /* 16 doubles */) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red16[4 * wave + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = red16[q] + red16[4 + q] + red16[8 + q] + red16[12 + q];
}

// value of lane DPP(l) inside the 16-lane row of the wave (both halves of the double)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, every lane ends with the same bits: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}

constexpr int STP = ST + 2;     // slab row stride of the ARD reduce (the MFMA fragment reads walk ACROSS the slab rows)

// ARD length-scale sums of one 64x64 tile: out[k] = wk[k] * sum_rc w_rc (x_rk - x_ck)^2 for every k < D (wk == nullptr: the
// weights are already folded into the scaled coordinates) -- Core/cov.py:924-931 summed against Q, all D derivatives at once.
//
// On the matrix cores, and for ANY D (round 4; the VALU form walked 3 instructions per element and coordinate and stopped at
// D = 64).  With row sums R_r = sum_c w_rc and column sums C_c = sum_r w_rc of the tile,
//     sum_rc w_rc (x_rk - x_ck)^2 = sum_r R_r x_rk^2 + sum_c x_ck (C_c x_ck - 2 P_ck),      P = W' Xr   (64 x 64 by 64 x D)
// and P needs no LDS round trip for W: thread (tr, tc) holds w[a][q] at row 4 tr + a, column 2 tc + (q & 1) + 32 (q >> 1), so
// for fixed (a, q) the wave's 64 lanes hold a 4 (rows: lane / 16) x 16 (columns: lane % 16) block -- the B operand of
// v_mfma_f64_16x16x4 with the contraction running over the wave's rows; the A operand is the slab of row coordinates
// (16 coordinates x those 4 rows).  Each wave contracts over ITS 16 rows (16 MFMAs per 16 coordinates) and keeps a partial
// P; everything after it is linear, so the partials are folded with the wave's own column sums and only the D results meet
// in LDS (fixed order).  The three terms cancel where a difference would not, so the coordinates are CENTRED first
// (mu[k] = column mean; the sums are shift-invariant): the rounding error is then c eps sum |w_rc| (|x_rk| + |x_ck|)^2
// against the difference form's c' eps sum |w_rc| (x_rk - x_ck)^2 -- the same order for centred data.
// sm: 2 * SKC * STP doubles.
__device__ __forceinline__ void ard_dim_reduce(const double* __restrict__ XT, long ldp, long r0, long c0, int dpad,
                                               double* __restrict__ sm, const double (&w)[4][4],
                                               const double* __restrict__ wk, const double* __restrict__ mu, int D,
                                               double* __restrict__ out) {
    __shared__ double ardA[4][64], ardB[4][64];   // per-wave sums of a chunk of 64 coordinates: the waves meet once per chunk
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l4 = lane >> 4, l15 = lane & 15;
    double* xr = sm;
    double* xc = sm + SKC * STP;
    double Cw[4], Rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                  // column sums over the wave's 16 rows, at the lane's four columns
        double v = (w[0][q] + w[1][q]) + (w[2][q] + w[3][q]);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        Cw[q] = v;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) Rr[a] = row16_sum((w[a][0] + w[a][1]) + (w[a][2] + w[a][3]));   // full row sums of the lane's rows
    const int sk = t >> 5, spr = t & 31;           // staging share: coordinates sk and sk + 8 of the slab, points 2 spr, 2 spr + 1
    double2_t gr[2], gc[2];
    double gm[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long k = k0 + sk + 8 * p;
            gr[p] = *(const double2_t*)(XT + k * ldp + r0 + 2 * spr);
            gc[p] = *(const double2_t*)(XT + k * ldp + c0 + 2 * spr);
            gm[p] = mu[k];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < dpad; k0 += SKC) {
        __syncthreads();                            // the previous slab (or the caller's use of sm) is consumed
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = sk + 8 * p;
            *(double2_t*)(xr + k * STP + 2 * spr) = double2_t{gr[p][0] - gm[p], gr[p][1] - gm[p]};
            *(double2_t*)(xc + k * STP + 2 * spr) = double2_t{gc[p][0] - gm[p], gc[p][1] - gm[p]};
        }
        __syncthreads();
        if (k0 + SKC < dpad) fetch(k0 + SKC);
        // A fragments: coordinate l15 of the slab at the wave's rows 16 wave + 4 l4 + a
        const double2_t f01 = *(const double2_t*)(xr + l15 * STP + 16 * wave + 4 * l4);
        const double2_t f23 = *(const double2_t*)(xr + l15 * STP + 16 * wave + 4 * l4 + 2);
        const double fa[4] = {f01[0], f01[1], f23[0], f23[1]};
        double4_t acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], w[a][q], acc[q], 0, 0, 0);
        // acc[q][r4] = P(coordinate 4 r4 + l4, column 2 l15 + (q & 1) + 32 (q >> 1)), partial over the wave's rows
        const int kc = k0 & 63;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const double* row = xc + (4 * r4 + l4) * STP + 2 * l15;
            const double2_t c01 = *(const double2_t*)row;
            const double2_t c23 = *(const double2_t*)(row + 32);
            const double xv[4] = {c01[0], c01[1], c23[0], c23[1]};
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fma(xv[q], fma(Cw[q], xv[q], -2.0 * acc[q][r4]), s);
            s = row16_sum(s);
            if (l15 == 0) ardA[wave][kc + 4 * r4 + l4] = s;
        }
        double rt = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) rt = fma(Rr[a] * fa[a], fa[a], rt);
        rt += __shfl_xor(rt, 16, 64);
        rt += __shfl_xor(rt, 32, 64);
        if (l4 == 0) ardB[wave][kc + l15] = rt;
        if (kc == 64 - SKC || k0 + SKC >= dpad) {  // a chunk of 64 coordinates (or the last one) is complete
            __syncthreads();
            const int kb = k0 - kc;
            if (t <= kc + SKC - 1 && kb + t < D) {
                const double v = ((ardA[0][t] + ardB[0][t]) + (ardA[1][t] + ardB[1][t])) +
                                 ((ardA[2][t] + ardB[2][t]) + (ardA[3][t] + ardB[3][t]));
                out[kb + t] = wk ? wk[kb + t] * v : v;
            }
            // the next chunk's first writes to ardA / ardB come after the next slab's two barriers
        }
    }
    __syncthreads();                                // a second call (second ARD leaf) reuses sm and ardA / ardB
}

// The same sums in the DIFFERENCE form -- sum_rc w_rc (x_rk - x_ck)^2 term by term, what the reference's per-coordinate
// getDerMatrix computes (Core/cov.py:924-931): three VALU instructions per element and coordinate, any D.  The fallback for data
// whose scaled, centred points lie too far out for the product form above (its error is eps sum |w| |x|^2, this one's
// eps sum |w| (x_r - x_c)^2): make_spec sets CovSpec::ard_grad_diff beyond |x|^2 = 1e6.  sm: 2 * SKC * STP doubles.
__device__ __forceinline__ void ard_dim_reduce_diff(const double* __restrict__ XT, long ldp, long r0, long c0, int dpad,
                                                    double* __restrict__ sm, const double (&w)[4][4],
                                                    const double* __restrict__ wk, int D, double* __restrict__ out) {
    __shared__ double dred[4][SKC];
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15, lane = t & 63, wave = t >> 6;
    double* xr = sm;
    double* xc = sm + SKC * STP;
    for (int k0 = 0; k0 < dpad; k0 += SKC) {
        __syncthreads();                            // the previous slab, its sums and the caller's use of sm are consumed
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = t + p * 256, k = v >> 5, pr = v & 31;
            *(double2_t*)(xr + k * STP + 2 * pr) = *(const double2_t*)(XT + (long)(k0 + k) * ldp + r0 + 2 * pr);
            *(double2_t*)(xc + k * STP + 2 * pr) = *(const double2_t*)(XT + (long)(k0 + k) * ldp + c0 + 2 * pr);
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < SKC; ++k) {
            const double2_t r01 = *(const double2_t*)(xr + k * STP + 4 * tr);
            const double2_t r23 = *(const double2_t*)(xr + k * STP + 4 * tr + 2);
            const double2_t c01 = *(const double2_t*)(xc + k * STP + 2 * tc);
            const double2_t c23 = *(const double2_t*)(xc + k * STP + 2 * tc + 32);
            const double rv[4] = {r01[0], r01[1], r23[0], r23[1]};
            const double cv[4] = {c01[0], c01[1], c23[0], c23[1]};
            double sacc = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = rv[a] - cv[b];
                    sacc = fma(w[a][b], df * df, sacc);
                }
            sacc = wave_sum(sacc);
            if (lane == 0) dred[wave][k] = sacc;
        }
        __syncthreads();
        if (t < SKC && k0 + t < D) {
            const double v = (dred[0][t] + dred[1][t]) + (dred[2][t] + dred[3][t]);
            out[k0 + t] = wk ? wk[k0 + t] * v : v;
        }
    }
    __syncthreads();                                // a second call (second ARD leaf) reuses sm and dred
}

// mu[k] = mean over the n points of coordinate k (fixed order); one block per coordinate
__global__ __launch_bounds__(256) void coord_mean_kernel(const double* __restrict__ XT, long ldp, long n, double* __restrict__ mu) {
    __shared__ double red[4];
    const double* row = XT + (long)blockIdx.x * ldp;
    double v = 0.0;
    for (long p = threadIdx.x; p < n; p += 256) v += row[p];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) mu[blockIdx.x] = tot / (double)n;
}

// partial[blk * nacc + h]: h < ncov -> sum Q dK_h ; h == ncov -> sn2 * trace(Q)
// KIND: the kernel family, a compile-time parameter (one functor's code and constants per instantiation)
template <int KIND>
__global__ __launch_bounds__(256) void hadamard_reduce_kernel(const double* __restrict__ XT, long ldp, long n, int dpad,
                                                              CovParams cp0, int ncov, double inv_sn2, double sn2,
                                                              const double* __restrict__ Binv, long ldb,
                                                              const double* __restrict__ alpha,
                                                              const double* __restrict__ wv,
                                                              double* __restrict__ partial, long nt,
                                                              const double* __restrict__ mu, long b0) {
    __shared__ __attribute__((aligned(16))) double sm[2 * SKC * STP];
    CovParams cp = cp0;
    cp.kind = KIND;
    const long b = blockIdx.x + b0;                 // b0: first block of a range of tile rows (sharded fit: one strip of B^-1)
    long r = (long)(((2.0 * nt + 1.0) - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)b)) * 0.5);
    if (r < 0) r = 0;
    while (r > 0 && r * nt - r * (r - 1) / 2 > b) --r;
    while ((r + 1) * nt - (r + 1) * r / 2 <= b) ++r;
    const long ti = r, tj = ti + (b - (r * nt - r * (r - 1) / 2));
    const long r0 = ti * ST, c0 = tj * ST;
    double s[4][4];
    sqdist_tile(XT, ldp, r0, XT, ldp, c0, dpad, sm, s);

    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double w[4][4];                 // weight * Q_rc * K_rc   (ARD) -- or per-hyper accumulators below
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, tq = 0.0;
    double ar[4], ac[4], wr[4], wc[4];        // Q_rc = Binv_rc * w_r w_c - alpha_r alpha_c  (Exact: w = 1/sn)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long rr = r0 + 4 * tr + a;
        ar[a] = rr < n ? alpha[rr] : 0.0;
        wr[a] = wv ? (rr < n ? wv[rr] : 0.0) : inv_sn2;
    }
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
        const long cc = c0 + 2 * tc + (bq & 1) + 32 * (bq >> 1);
        ac[bq] = cc < n ? alpha[cc] : 0.0;
        wc[bq] = wv ? (cc < n ? wv[cc] : 0.0) : 1.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long rr = r0 + 4 * tr + a;
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            const long cb = c0 + 2 * tc + 32 * bh;
            const double2_t bv = *(const double2_t*)(Binv + rr * ldb + cb);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int bq = 2 * bh + e;
                const long cc = cb + e;
                double wt = (cc > rr) ? 2.0 : (cc == rr ? 1.0 : 0.0);     // symmetric: count the mirror
                if (rr >= n || cc >= n) wt = 0.0;                          // padding
                const double q = bv[e] * (wr[a] * wc[bq]) - ar[a] * ac[bq];
                const double wq = (wt != 0.0) ? wt * q : 0.0;             // never let unused entries in
                if (cc == rr && rr < n) tq += sn2 * q;
                if (cp.kind == 1) {
                    const double K = cov_value(cp, s[a][bq]);
                    w[a][bq] = wq * K;
                    g1 += 2.0 * wq * K;                                    // d/d log sf
                } else if (cp.kind == 6) {                                 // RQard (Core/cov.py:1412-1425)
                    const double Kp = 1.0 + 0.5 * s[a][bq] / cp.alpha;
                    const double lk = log(Kp);
                    const double Ka = cp.sf2 * exp(-cp.alpha * lk);
                    w[a][bq] = cp.ref_der ? 0.0 : wq * Ka / Kp;         // compat: zero length-scale gradient (cov.py:1415)
                    g1 = fma(wq, 2.0 * Ka, g1);
                    g2 = fma(wq, Ka * (0.5 * s[a][bq] / Kp - cp.alpha * lk), g2);
                } else {
                    double d0, d1, d2;
                    cov_deriv_all(cp, s[a][bq], d0, d1, d2, cc == rr);
                    g0 = fma(wq, d0, g0);
                    g1 = fma(wq, d1, g1);
                    g2 = fma(wq, d2, g2);
                    w[a][bq] = 0.0;
                }
            }
        }
    }
    double* out = partial + (long)blockIdx.x * (long)(ncov + 1);
    __shared__ double red16[16];
    if (cov_is_ard(cp)) {
        // this kernel serves the plain ARD kinds only when CovSpec::ard_grad_diff is set (hadamard_partial_launch): the
        // difference form throughout -- K above (sqdist_tile) and the per-coordinate sums here
        ard_dim_reduce_diff(XT, ldp, r0, c0, dpad, sm, w, nullptr, cp.D, out);
        double v4[4] = {g1, tq, g2, 0.0};
        block_sum4(v4, red16);
        if (t == 0) {
            out[cp.D] = v4[0];
            if (cp.kind == 6) out[cp.D + 1] = v4[2];
            out[ncov] = v4[1];
        }
    } else {
        double v4[4] = {g0, g1, g2, tq};
        block_sum4(v4, red16);
        if (t == 0) {
            out[0] = v4[0];
            if (ncov > 1) out[1] = v4[1];
            if (ncov > 2) out[2] = v4[2];
            out[ncov] = v4[3];
        }
    }
}

// The Hadamard reduce of the plain ARD kinds (RBFard = SEard, RQard) -- cfg 3 -- with BOTH contractions on the matrix cores:
//   * the squared distances of the tile in the Gram form  r^2 = |a|^2 + |b|^2 - 2 a.b  on CENTRED, scaled coordinates (one
//     v_mfma_f64_16x16x4 per 4 coordinates and 16 x 16 outputs: a third of the VALU difference form's issue slots).  This K
//     only WEIGHTS the gradient sums (tolerance 1e-7 on dnlZ): its absolute error in r^2 is eps (|a|^2 + |b|^2), i.e. a relative
//     error of that size in K -- 1e-16 at ell ~ sqrt(D), 3e-10 at the smallest length scale the optimiser's range allows
//     (ell = e^-5, D = 64).  K itself (the factor, nlZ, alpha) keeps the difference form of the reference's cdist (assemble.hip).
//     The diagonal is forced to r^2 = 0, negative round-off is clamped.
//   * the per-coordinate sums through W' Xr as in ard_dim_reduce.
// Thread layout = the MFMA accumulator layout, so nothing is transposed between the two: wave w, lane (l4, l15) holds rows
// 16 w + 4 a + l4 (a < 4) and columns 16 q + l15 (q < 4) of the 64 x 64 tile.  The coordinates of both point sets are staged
// ONCE per chunk of 64 coordinates (dynamic LDS, 2 x 64 x 66 doubles): two barriers per tile for D <= 64 instead of sixteen.
// NP double2 pieces of one thread's share of a staged chunk: piece i = coordinate 4 i of the wave's residue class
template <int NP>
__device__ __forceinline__ void stage_pieces(const double* __restrict__ gp, long ldp, const double* __restrict__ mup,
                                             double* __restrict__ slp) {
    double2_t g[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) g[i] = *(const double2_t*)(gp + (long)(4 * i) * ldp);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const double m_ = mup[4 * i];
        *(double2_t*)(slp + 4 * i * STP) = double2_t{g[i][0] - m_, g[i][1] - m_};
    }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void hadamard_ard_kernel(const double* __restrict__ XT, long ldp, long n, int dpad,
                                                              CovParams cp0, int ncov, double inv_sn2, double sn2,
                                                              const double* __restrict__ Binv, long ldb,
                                                              const double* __restrict__ alpha,
                                                              const double* __restrict__ wv,
                                                              double* __restrict__ partial, long nt,
                                                              const double* __restrict__ mu,
                                                              const double* __restrict__ nrm, long b0) {
    extern __shared__ __attribute__((aligned(16))) double smx[];
    __shared__ double ardA[4][64], ardB[4][64];
    __shared__ double red16[16];
    CovParams cp = cp0;
    cp.kind = KIND;
    const long b = blockIdx.x + b0;
    long r = (long)(((2.0 * nt + 1.0) - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)b)) * 0.5);
    if (r < 0) r = 0;
    while (r > 0 && r * nt - r * (r - 1) / 2 > b) --r;
    while ((r + 1) * nt - (r + 1) * r / 2 <= b) ++r;
    const long ti = r, tj = ti + (b - (r * nt - r * (r - 1) / 2));
    const long r0 = ti * ST, c0 = tj * ST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l4 = lane >> 4, l15 = lane & 15;
    const int CH = dpad < 64 ? dpad : 64;            // coordinates per staged chunk
    double* xr = smx;
    double* xc = smx + CH * STP;
    // coordinates k0 .. k0 + kk of both point sets, centred.  kk / 4 double2 pieces per thread (piece i: coordinate wave + 4 i,
    // 32 lanes the rows' points, 32 the columns'); ALL pieces of a chunk are in flight before the first is used (one round
    // trip per chunk, not one per piece), the means come through the scalar unit (the coordinate index is wave-uniform)
    const int spr = lane & 31, sside = lane >> 5;
    const int wvu = __builtin_amdgcn_readfirstlane(wave);
    const double* sgp = XT + (sside ? c0 : r0) + 2 * spr + (long)wvu * ldp;
    double* slp = (sside ? xc : xr) + 2 * spr + wvu * STP;
    const double* mup = mu + wvu;
    auto stage = [&](int k0, int kk) {                // kk is a multiple of 16: 4, 8, 12 or 16 pieces, branch-free inside
        const double* gp = sgp + (long)k0 * ldp;
        switch (kk >> 4) {
            case 1: stage_pieces<4>(gp, ldp, mup + k0, slp); break;
            case 2: stage_pieces<8>(gp, ldp, mup + k0, slp); break;
            case 3: stage_pieces<12>(gp, ldp, mup + k0, slp); break;
            default: stage_pieces<16>(gp, ldp, mup + k0, slp); break;
        }
    };
    // the tile of B^-1, alpha, the per-point weights and norms: requested NOW, used after the Gram products
    double bvv[4][4];
    double ar[4], wr[4], nr[4], ac[4], wc[4], nc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) bvv[a][q] = Binv[(r0 + 16 * wave + 4 * a + l4) * ldb + c0 + 16 * q + l15];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long rr = r0 + 16 * wave + 4 * a + l4, rl = rr < n ? rr : n - 1;      // loads without a branch: clamped index,
        const double av = alpha[rl], wvv = wv ? wv[rl] : inv_sn2;                      // the value selected afterwards
        ar[a] = rr < n ? av : 0.0;
        wr[a] = wv ? (rr < n ? wvv : 0.0) : inv_sn2;
        nr[a] = nrm[rr];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long cc = c0 + 16 * q + l15, cl = cc < n ? cc : n - 1;
        const double av = alpha[cl], wvv = wv ? wv[cl] : 1.0;
        ac[q] = cc < n ? av : 0.0;
        wc[q] = wv ? (cc < n ? wvv : 0.0) : 1.0;
        nc[q] = nrm[cc];
    }
    // ---- Gram products: S(row, col) = sum_k xr(k, row) xc(k, col) ------------------------------------------------------
    double4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < dpad; k0 += 64) {
        const int kk = (dpad - k0) < 64 ? (dpad - k0) : 64;
        if (k0) __syncthreads();
        stage(k0, kk);
        __syncthreads();
        // fragments of step ks + 4 are read before the MFMAs of step ks are issued
        const double* ap = xr + l4 * STP + 16 * wave + l15;
        const double* bp = xc + l4 * STP + l15;
        double fa0 = ap[0], fb0[4] = {bp[0], bp[16], bp[32], bp[48]};
        for (int ks = 0; ks < kk; ks += 8) {
            const double* a1 = ap + (ks + 4) * STP;
            const double* b1 = bp + (ks + 4) * STP;
            const double fa1 = a1[0], fb1[4] = {b1[0], b1[16], b1[32], b1[48]};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fb0[q], acc[q], 0, 0, 0);
            if (ks + 8 < kk) {
                const double* a2 = ap + (ks + 8) * STP;
                const double* b2 = bp + (ks + 8) * STP;
                fa0 = a2[0]; fb0[0] = b2[0]; fb0[1] = b2[16]; fb0[2] = b2[32]; fb0[3] = b2[48];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fb1[q], acc[q], 0, 0, 0);
        }
    }
    // ---- element-wise: Q, K, the weights of the per-coordinate sums ------------------------------------------------------
    double w[4][4];
    double g1 = 0.0, g2 = 0.0, tq = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long rr = r0 + 16 * wave + 4 * a + l4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long cc = c0 + 16 * q + l15;
            const double bv = bvv[a][q];
            double wt = (cc > rr) ? 2.0 : (cc == rr ? 1.0 : 0.0);     // symmetric: count the mirror
            if (rr >= n || cc >= n) wt = 0.0;                          // padding
            const double qv = bv * (wr[a] * wc[q]) - ar[a] * ac[q];
            const double wq = (wt != 0.0) ? wt * qv : 0.0;             // never let unused entries in
            if (cc == rr && rr < n) tq += sn2 * qv;
            double s = fmax(fma(-2.0, acc[q][a], nr[a] + nc[q]), 0.0);
            if (cc == rr) s = 0.0;
            if (KIND == 1) {
                const double K = cp.sf2 * exp_nonpos(-0.5 * s);
                w[a][q] = wq * K;
                g1 += 2.0 * wq * K;                                    // d/d log sf
            } else {                                                   // RQard (Core/cov.py:1412-1425)
                const double Kp = 1.0 + 0.5 * s / cp.alpha;
                const double lk = log(Kp);
                const double Ka = cp.sf2 * exp(-cp.alpha * lk);
                w[a][q] = cp.ref_der ? 0.0 : wq * Ka / Kp;             // compat: zero length-scale gradient (cov.py:1415)
                g1 = fma(wq, 2.0 * Ka, g1);
                g2 = fma(wq, Ka * (0.5 * s / Kp - cp.alpha * lk), g2);
            }
        }
    }
    // ---- per-coordinate sums:  sum_rc w_rc (x_rk - x_ck)^2 = sum_r R_r x_rk^2 + sum_c x_ck (C_c x_ck - 2 (W' Xr)_ck) ----------
    double Cw[4], Rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // column sums over the wave's 16 rows, at the lane's four columns
        double v = (w[0][q] + w[1][q]) + (w[2][q] + w[3][q]);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        Cw[q] = v;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) Rr[a] = row16_sum((w[a][0] + w[a][1]) + (w[a][2] + w[a][3]));   // full row sums
    double* out = partial + (long)blockIdx.x * (long)(ncov + 1);
    {                                                 // the three scalar sums: per wave now, met behind the barrier of the loop below
        const double s1 = wave_sum(g1), s2 = wave_sum(tq), s3 = wave_sum(g2);
        if (lane == 0) { red16[4 * wave] = s1; red16[4 * wave + 1] = s2; red16[4 * wave + 2] = s3; }
    }
    for (int k0 = 0; k0 < dpad; k0 += 64) {
        const int kk = (dpad - k0) < 64 ? (dpad - k0) : 64;
        if (dpad > 64) {                              // more than one chunk: the Gram pass left the LAST one in LDS
            __syncthreads();
            stage(k0, kk);
            __syncthreads();
        }
        for (int jb = 0; jb < kk; jb += 16) {
            double fa[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = xr[(jb + l15) * STP + 16 * wave + 4 * a + l4];
            double4_t p2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) p2[q] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], w[a][q], p2[q], 0, 0, 0);
            // p2[q][r4] = (W' Xr)(coordinate jb + 4 r4 + l4, column 16 q + l15), partial over the wave's rows
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const double* row = xc + (jb + 4 * r4 + l4) * STP + l15;
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double xv = row[16 * q];
                    s = fma(xv, fma(Cw[q], xv, -2.0 * p2[q][r4]), s);
                }
                s = row16_sum(s);
                if (l15 == 0) ardA[wave][jb + 4 * r4 + l4] = s;
            }
            double rt = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) rt = fma(Rr[a] * fa[a], fa[a], rt);
            rt += __shfl_xor(rt, 16, 64);
            rt += __shfl_xor(rt, 32, 64);
            if (l4 == 0) ardB[wave][jb + l15] = rt;
        }
        __syncthreads();
        if (t < kk && k0 + t < cp.D)
            out[k0 + t] = ((ardA[0][t] + ardB[0][t]) + (ardA[1][t] + ardB[1][t])) + ((ardA[2][t] + ardB[2][t]) + (ardA[3][t] + ardB[3][t]));
    }
    if (t == 0) {                                     // (the last barrier of the loop above made red16 visible)
        out[cp.D] = (red16[0] + red16[4]) + (red16[8] + red16[12]);
        if (KIND == 6) out[cp.D + 1] = (red16[2] + red16[6]) + (red16[10] + red16[14]);
        out[ncov] = (red16[1] + red16[5]) + (red16[9] + red16[13]);
    }
}

// nrm[p] = sum_k (x_pk - mu_k)^2 of the scaled coordinates, p < np (padding points included: they carry weight 0)
__global__ __launch_bounds__(256) void point_norm_kernel(const double* __restrict__ XT, long ldp, long np, int dpad,
                                                         const double* __restrict__ mu, double* __restrict__ nrm) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    double s = 0.0;
    for (int k = 0; k < dpad; ++k) {
        const double v = XT[(long)k * ldp + p] - mu[k];
        s = fma(v, v, s);
    }
    nrm[p] = s;
}

// Same reduction for a composite program: per element the leaf values, their (up to three) derivatives and the
// chain-rule weights through the Sum/Product/Scale tree.  Elements run in a rolled loop over LDS-staged distances
// so that the leaf functors are instantiated once.
// NARD: number of ARD leaves of the program (own weighted distances; per-dimension length-scale sums in a second pass
// per leaf).  The second leaf's distance and weights stay in registers (sel16 / put16): static LDS ends at 64 KB.
template <int NARD>
__global__ __launch_bounds__(256) void hadamard_prog_kernel(const double* __restrict__ XT, long ldp, long n, int dpad,
                                                            CovProgram P, int ncov, double inv_sn2, double sn2,
                                                            const double* __restrict__ Binv, long ldb,
                                                            const double* __restrict__ alpha,
                                                            const double* __restrict__ wv,
                                                            double* __restrict__ partial, long nt,
                                                            const double* __restrict__ mu, long b0) {
    constexpr bool PARD = NARD >= 1, PARD2 = NARD == 2;
    __shared__ __attribute__((aligned(16))) double sm[(PARD ? 2 : 1) * 16 * 256];
    __shared__ double red[4];
    const long b = blockIdx.x + b0;
    long r = (long)(((2.0 * nt + 1.0) - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)b)) * 0.5);
    if (r < 0) r = 0;
    while (r > 0 && r * nt - r * (r - 1) / 2 > b) --r;
    while ((r + 1) * nt - (r + 1) * r / 2 <= b) ++r;
    const long ti = r, tj = ti + (b - (r * nt - r * (r - 1) / 2));
    const long r0 = ti * ST, c0 = tj * ST;
    double s[4][4], s1[PARD ? 4 : 1][4], s2[PARD2 ? 4 : 1][4], w2[PARD2 ? 4 : 1][4];
    if constexpr (PARD2) sqdist_tile3(XT, ldp, r0, XT, ldp, c0, dpad, P.ardw, P.ardw2, sm, s, s1, s2);
    else if constexpr (PARD) sqdist_tile2(XT, ldp, r0, XT, ldp, c0, dpad, P.ardw, sm, s, s1);
    else sqdist_tile(XT, ldp, r0, XT, ldp, c0, dpad, sm, s);
    if constexpr (PARD2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) w2[e >> 2][e & 3] = 0.0;
    }
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double* sv = sm + t;
    double* sv1 = sm + 16 * 256 + t;       // PARD: the ARD distance going in, the length-scale weight omega coming out
#pragma unroll
    for (int e = 0; e < 16; ++e) { sv[e * 256] = s[e >> 2][e & 3]; if (PARD) sv1[e * 256] = s1[e >> 2][e & 3]; }

    double gl[CP_MAXLEAF][3], gs[CP_MAXSCALE], tq = 0.0;
#pragma unroll
    for (int l = 0; l < CP_MAXLEAF; ++l) gl[l][0] = gl[l][1] = gl[l][2] = 0.0;
#pragma unroll
    for (int k = 0; k < CP_MAXSCALE; ++k) gs[k] = 0.0;

#pragma unroll 1
    for (int e = 0; e < 16; ++e) {
        const int a = e >> 2, bq = e & 3;
        const long rr = r0 + 4 * tr + a;
        const long cc = c0 + 2 * tc + (bq & 1) + 32 * (bq >> 1);
        double wt = (cc > rr) ? 2.0 : (cc == rr ? 1.0 : 0.0);
        if (rr >= n || cc >= n) wt = 0.0;
        if (wt == 0.0) { if (PARD) sv1[e * 256] = 0.0; continue; }
        const double wr = wv ? wv[rr] : inv_sn2, wc = wv ? wv[cc] : 1.0;
        const double q = Binv[rr * ldb + cc] * (wr * wc) - alpha[rr] * alpha[cc];
        const double wq = wt * q;
        if (cc == rr) tq += sn2 * q;
        const double r2 = sv[e * 256];
        const bool same = cc == rr;
        double v[CP_MAXLEAF], d[CP_MAXLEAF][3], T[CP_MAXTERM];
        double ardfac = 0.0, ardfac2 = 0.0;   // dK_l / d log ell_k = ardfac * (scaled squared difference in coordinate k)
#pragma unroll
        for (int l = 0; l < CP_MAXLEAF; ++l) {
            v[l] = 1.0; d[l][0] = d[l][1] = d[l][2] = 0.0;
            if (l < P.nleaf) {
                const bool a1 = PARD && l == P.ard_leaf, a2 = PARD2 && l == P.ard_leaf2;
                if (a1 || a2) {
                    // d[l][0]: magnitude hyper, d[l][1]: RQard shape hyper (Core/cov.py:922-936, 1412-1425)
                    double sl = sv1[e * 256];
                    if constexpr (PARD2) { if (a2) sl = sel16(s2, e); }
                    const CovParams& lp = P.leaf[l];
                    double af;
                    if (lp.kind == 1) {
                        v[l] = lp.sf2 * exp_nonpos(-0.5 * sl);
                        d[l][0] = 2.0 * v[l];
                        af = v[l];
                    } else {
                        const double Kp = 1.0 + 0.5 * sl / lp.alpha;
                        const double lk = log(Kp);
                        v[l] = lp.sf2 * exp_nonpos(-lp.alpha * lk);
                        d[l][0] = 2.0 * v[l];
                        d[l][1] = v[l] * (0.5 * sl / Kp - lp.alpha * lk);
                        af = lp.ref_der ? 0.0 : v[l] / Kp;
                    }
                    if (a2) ardfac2 = af; else ardfac = af;
                } else {
                    const double sl = r2 * P.is2[l];
                    v[l] = cov_value<true>(P.leaf[l], sl, same);
                    cov_deriv_all<true>(P.leaf[l], sl, d[l][0], d[l][1], d[l][2], same);
                }
            }
        }
        prog_terms(P, v, T);
#pragma unroll
        for (int l = 0; l < CP_MAXLEAF; ++l) {
            if (l < P.nleaf) {
                const double wl = wq * prog_leaf_weight(P, v, l);
                gl[l][0] = fma(wl, d[l][0], gl[l][0]);
                gl[l][1] = fma(wl, d[l][1], gl[l][1]);
                gl[l][2] = fma(wl, d[l][2], gl[l][2]);
                if (PARD && l == P.ard_leaf) sv1[e * 256] = wl * ardfac;
                if constexpr (PARD2) { if (l == P.ard_leaf2) put16(w2, e, wl * ardfac2); }
            }
        }
#pragma unroll
        for (int k = 0; k < CP_MAXSCALE; ++k) {
            if (k < P.nscale) {
                double acc = 0.0;
#pragma unroll
                for (int tt = 0; tt < CP_MAXTERM; ++tt)
                    if ((P.ts[tt] >> k) & 1u) acc += T[tt];
                gs[k] = fma(wq, 2.0 * acc, gs[k]);
            }
        }
    }
    double* out = partial + (long)blockIdx.x * (long)(ncov + 1);
#pragma unroll
    for (int l = 0; l < CP_MAXLEAF; ++l) {
        if (l < P.nleaf) {
            if ((PARD && l == P.ard_leaf) || (PARD2 && l == P.ard_leaf2)) {      // ARD leaf: D length-scales (below), magnitude, [shape]
                const int D = P.leaf[l].D;
                const double t0 = block_sum(gl[l][0], red);
                const double t1 = block_sum(gl[l][1], red);
                if (t == 0) { out[P.hyp0[l] + D] = t0; if (P.leaf[l].kind == 6) out[P.hyp0[l] + D + 1] = t1; }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < P.nh[l]) {
                    const double tot = block_sum(gl[l][j], red);
                    if (t == 0) out[P.hyp0[l] + j] = tot;
                }
            }
        }
    }
    if constexpr (PARD) {
        double w[4][4];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e >> 2][e & 3] = sv1[e * 256];
        const int la = P.ard_leaf & 7;
        // mu == nullptr: the host asks for the difference form (CovSpec::ard_grad_diff, far-out data)
        if (mu) ard_dim_reduce(XT, ldp, r0, c0, dpad, sm, w, P.ardw, mu, P.leaf[la].D, out + P.hyp0[la]);
        else ard_dim_reduce_diff(XT, ldp, r0, c0, dpad, sm, w, P.ardw, P.leaf[la].D, out + P.hyp0[la]);
        if constexpr (PARD2) {
            const int lb = P.ard_leaf2 & 7;
            __syncthreads();
            if (mu) ard_dim_reduce(XT, ldp, r0, c0, dpad, sm, w2, P.ardw2, mu, P.leaf[lb].D, out + P.hyp0[lb]);
            else ard_dim_reduce_diff(XT, ldp, r0, c0, dpad, sm, w2, P.ardw2, P.leaf[lb].D, out + P.hyp0[lb]);
        }
    }
#pragma unroll
    for (int k = 0; k < CP_MAXSCALE; ++k) {
        if (k < P.nscale) {
            const double tot = block_sum(gs[k], red);
            if (t == 0) out[P.shyp[k]] = tot;
        }
    }
    const double tt = block_sum(tq, red);
    if (t == 0) out[ncov] = tt;
}

// out[h] = sum_b partial[b*nacc + h], fixed order; one block per h
__global__ __launch_bounds__(256) void final_reduce_kernel(const double* __restrict__ partial, long nblk, int nacc,
                                                           double* __restrict__ out) {
    __shared__ double red[4];
    const int h = blockIdx.x;
    double v = 0.0;
    for (long b = threadIdx.x; b < nblk; b += 256) v += partial[b * nacc + h];
    const double tot = block_sum(v, red);
    if (threadIdx.x == 0) out[h] = tot;
}

// y[j] = scale * sum_{i >= j} W(i,j) z[i*zs]   (W column-major lower, n x n); one wave per column
__global__ __launch_bounds__(256) void col_dot_kernel(const double* __restrict__ W, long ldw, long n,
                                                      const double* __restrict__ z, long zs, double scale,
                                                      double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double* col = W + j * ldw;
    double v = 0.0;
    for (long i = (j & ~63L) + lane; i < n; i += 64)
        if (i >= j) v = fma(col[i], z[i * zs], v);
    v = wave_sum(v);
    if (lane == 0) y[j] = scale * v;
}

// out[0] = sum_{j<n} log L(j,j) ; out[1] = sum_{j<n} z_j^2 ; (z strided by zs)
__global__ __launch_bounds__(256) void logdet_ztz_kernel(const double* __restrict__ L, long ldl, long n,
                                                         const double* __restrict__ z, long zs,
                                                         double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0, b = 0.0;
    for (long j = threadIdx.x; j < n; j += 256) {
        a += log(L[j + j * ldl]);
        const double zj = z[j * zs];
        b = fma(zj, zj, b);
    }
    const double ta = block_sum(a, red);
    const double tb = block_sum(b, red);
    if (threadIdx.x == 0) { out[0] = ta; out[1] = tb; }
}

// out[0] = sum v_i^2, out[1] = sum u_i v_i
__global__ __launch_bounds__(256) void dot2_kernel(const double* __restrict__ u, const double* __restrict__ v, long n,
                                                   double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0, b = 0.0;
    for (long j = threadIdx.x; j < n; j += 256) {
        a = fma(v[j], v[j], a);
        b = fma(u[j], v[j], b);
    }
    const double ta = block_sum(a, red);
    const double tb = block_sum(b, red);
    if (threadIdx.x == 0) { out[0] = ta; out[1] = tb; }
}

// augmented right-hand-side row of the factor buffer: F(np, j) = y_j - m_j (j < n)
__global__ void aug_rhs_kernel(const double* __restrict__ y, const double* __restrict__ m, long n, double* __restrict__ F,
                               long ldf, long row, double* __restrict__ rvec) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const double r = y[j] - m[j];
        F[row + j * ldf] = r;
        rvec[j] = r;
    }
}

__global__ void zero_upper_strip_kernel(double* __restrict__ F, long ldf, long np, long row0, long nrows) {
    // zero F(row0 .. row0+nrows-1, 0..np-1)
    const long j = (long)blockIdx.x;
    for (long i = threadIdx.x; i < nrows; i += blockDim.x) F[row0 + i + j * ldf] = 0.0;
}

// out[j] = add[j] + sum_i A(i,j) v[i]    (A column-major nrows x ncols); one wave per column
__global__ __launch_bounds__(256) void col_dot_full_kernel(const double* __restrict__ A, long lda, long nrows,
                                                           long ncols, const double* __restrict__ v,
                                                           const double* __restrict__ add, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double* col = A + j * lda;
    double s = 0.0;
    for (long i = lane; i < nrows; i += 64) s = fma(col[i], v[i], s);
    s = wave_sum(s);
    if (lane == 0) out[j] = (add ? add[j] : 0.0) + s;
}

// out[j] = max(kss - scale * sum_i A(i,j)^2, 0)
__global__ __launch_bounds__(256) void col_sumsq_kernel(const double* __restrict__ A, long lda, long nrows,
                                                        long ncols, double kss, double scale,
                                                        double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double* col = A + j * lda;
    double s = 0.0;
    for (long i = lane; i < nrows; i += 64) s = fma(col[i], col[i], s);
    s = wave_sum(s);
    if (lane == 0) out[j] = fmax(kss - scale * s, 0.0);
}

// acc[j] += sum_i A(i,j)^2   (one wave per column; the panels of a distributed factor add up in stream order)
__global__ __launch_bounds__(256) void colsumsq_acc_kernel(const double* __restrict__ A, long lda, long nrows, long ncols,
                                                           double* __restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const double* col = A + j * lda;
    double s = 0.0;
    for (long i = lane; i < nrows; i += 64) s = fma(col[i], col[i], s);
    s = wave_sum(s);
    if (lane == 0) acc[j] += s;
}

// A(i,j) *= s[i]
__global__ __launch_bounds__(256) void row_scale_kernel(double* __restrict__ A, long lda, long nrows, long ncols,
                                                        const double* __restrict__ s) {
    const long j = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nrows && j < ncols) A[i + j * lda] *= s[i];
}

// E(i,j) = (i == j) for j >= i (the upper trapezoid is the only part of the E region a fit ever dirties)
__global__ __launch_bounds__(256) void identity_upper_kernel(double* __restrict__ E, long lde, long np) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    for (long j = blockIdx.y; j < np; j += gridDim.y)
        if (i <= j) E[i + j * lde] = (i == j) ? 1.0 : 0.0;
}

// partial[c*np + i] = sum_{k in chunk c, k >= i} E(i,k) z[k]   (E column-major: coalesced over i)
__global__ __launch_bounds__(256) void upper_matvec_kernel(const double* __restrict__ E, long lde, long np,
                                                           const double* __restrict__ z, double* __restrict__ partial,
                                                           int nchunk) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const long kc = (np + nchunk - 1) / nchunk;
    const long k0 = c * kc, k1 = k0 + kc < np ? k0 + kc : np;
    if (i >= np) return;
    // eight independent loads in flight per thread (one load per iteration left the kernel latency-bound at 2 TB/s);
    // the eight partial sums are combined in a fixed order
    double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    long k = (k0 > i ? k0 : i);
    for (; k + 8 <= k1; k += 8) {
        double e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = E[i + (k + q) * lde];
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] = fma(e[q], z[k + q], a8[q]);
    }
    for (; k < k1; ++k) a8[0] = fma(E[i + k * lde], z[k], a8[0]);
    partial[(long)c * np + i] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}
__global__ __launch_bounds__(256) void upper_matvec_reduce_kernel(const double* __restrict__ partial, long np, int nchunk,
                                                                  double scale, double* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    double acc = 0.0;
    for (int c = 0; c < nchunk; ++c) acc += partial[(long)c * np + i];
    y[i] = scale * acc;
}

}  // namespace

int identity_upper_launch(double* E, long lde, long np, hipStream_t st) {
    hipLaunchKernelGGL(identity_upper_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)(np < 65535 ? np : 65535)), dim3(256), 0, st, E, lde,
                       np);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int upper_matvec_launch(const double* E, long lde, long np, const double* z, double scale, double* partial, double* y,
                        hipStream_t st) {
    const int nchunk = 32;
    hipLaunchKernelGGL(upper_matvec_kernel, dim3((unsigned)((np + 255) / 256), nchunk), dim3(256), 0, st, E, lde, np, z,
                       partial, nchunk);
    hipLaunchKernelGGL(upper_matvec_reduce_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, partial, np,
                       nchunk, scale, y);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int col_dot_full_launch(const double* A, long lda, long nrows, long ncols, const double* v, const double* add,
                        double* out, hipStream_t st) {
    hipLaunchKernelGGL(col_dot_full_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, A, lda, nrows, ncols, v,
                       add, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int col_sumsq_launch(const double* A, long lda, long nrows, long ncols, double kss, double scale, double* out,
                     hipStream_t st) {
    hipLaunchKernelGGL(col_sumsq_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, A, lda, nrows, ncols, kss,
                       scale, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int colsumsq_acc_launch(const double* A, long lda, long nrows, long ncols, double* acc, hipStream_t st) {
    hipLaunchKernelGGL(colsumsq_acc_kernel, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st, A, lda, nrows, ncols, acc);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int row_scale_launch(double* A, long lda, long nrows, long ncols, const double* s, hipStream_t st) {
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((nrows + 255) / 256), (unsigned)ncols), dim3(256), 0, st, A, lda,
                       nrows, ncols, s);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// blocks of the tile rows [tr0, tr0 + trn) of the upper-triangular 64-tile grid of an np x np matrix
static long tri_blocks_before(long nt, long r) { return r * nt - r * (r - 1) / 2; }
long hadamard_block_count(long np, long tr0, long trn) {
    const long nt = np / ST;
    return tri_blocks_before(nt, tr0 + trn) - tri_blocks_before(nt, tr0);
}

// the coordinate means the ARD reduce centres with (no-op for covariance functions without an ARD leaf)
// mu: HADAMARD_PREP_MU doubles of means, then np squared norms of the centred points (plain ARD kinds: the Gram form of
// hadamard_ard_kernel) -- hadamard_prep_count(np) doubles in all
long hadamard_prep_count(long np) { return HADAMARD_PREP_MU + np; }
int hadamard_prepare_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double* mu, hipStream_t st,
                            bool force) {
    const bool ard = force || (cs.prog ? cs.pg.ard_leaf >= 0 : (cs.cp.kind == 1 || cs.cp.kind == 6));
    if (ard && dpad > HADAMARD_PREP_MU) return PGP_ERR_HIP;            // the mean region of the prep buffer (callers gate on it)
    if (ard) hipLaunchKernelGGL(coord_mean_kernel, dim3((unsigned)dpad), dim3(256), 0, st, XT, ldp, n, mu);
    if (ard && !cs.prog)
        hipLaunchKernelGGL(point_norm_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, XT, ldp, np, dpad, mu,
                           mu + HADAMARD_PREP_MU);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// Per-block partial sums of the tile rows [tr0, tr0 + trn) (64 rows each): `Binv` is addressed as Binv[r * ldb + c] for
// c >= r (GLOBAL indices: the caller of a strip passes a pointer shifted accordingly), partial receives
// hadamard_block_count(np, tr0, trn) * (ncov + 1) doubles.
int hadamard_partial_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, int ncov, double sn2,
                            const double* Binv, long ldb, const double* alpha, const double* wv, double* partial,
                            const double* mu, long tr0, long trn, hipStream_t st) {
    const long nt = np / ST;
    const long b0 = tri_blocks_before(nt, tr0);
    const long nblk = tri_blocks_before(nt, tr0 + trn) - b0;
    if (nblk <= 0) return PGP_OK;
    if (cs.prog) {
        CovProgram pg = cs.pg;
        for (int l = 0; l < pg.nleaf; ++l) pg.leaf[l].train = 1;
        if (cs.ard_grad_diff) mu = nullptr;           // the kernels' switch to the difference-form per-coordinate sums
        if (pg.ard_leaf2 >= 0)
            hipLaunchKernelGGL(hadamard_prog_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, dpad, pg, ncov,
                               1.0 / sn2, sn2, Binv, ldb, alpha, wv, partial, nt, mu, b0);
        else if (pg.ard_leaf >= 0)
            hipLaunchKernelGGL(hadamard_prog_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, dpad, pg, ncov,
                               1.0 / sn2, sn2, Binv, ldb, alpha, wv, partial, nt, mu, b0);
        else
            hipLaunchKernelGGL(hadamard_prog_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, dpad, pg, ncov,
                               1.0 / sn2, sn2, Binv, ldb, alpha, wv, partial, nt, mu, b0);
    } else if ((cs.cp.kind == 1 || cs.cp.kind == 6) && !cs.ard_grad_diff) {
        CovParams cp = cs.cp;
        cp.train = 1;
        const int CH = dpad < 64 ? dpad : 64;
        const size_t shm = (size_t)2 * CH * STP * sizeof(double);
        if (cp.kind == 1) {
            func_max_dynamic_lds((const void*)hadamard_ard_kernel<1>, shm);
            hipLaunchKernelGGL(hadamard_ard_kernel<1>, dim3((unsigned)nblk), dim3(256), shm, st, XT, ldp, n, dpad, cp, ncov, 1.0 / sn2,
                               sn2, Binv, ldb, alpha, wv, partial, nt, mu, mu + HADAMARD_PREP_MU, b0);
        } else {
            func_max_dynamic_lds((const void*)hadamard_ard_kernel<6>, shm);
            hipLaunchKernelGGL(hadamard_ard_kernel<6>, dim3((unsigned)nblk), dim3(256), shm, st, XT, ldp, n, dpad, cp, ncov, 1.0 / sn2,
                               sn2, Binv, ldb, alpha, wv, partial, nt, mu, mu + HADAMARD_PREP_MU, b0);
        }
    } else {
        CovParams cp = cs.cp;
        cp.train = 1;
#define HLAUNCH(K) hipLaunchKernelGGL(hadamard_reduce_kernel<K>, dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, dpad, cp, \
                                      ncov, 1.0 / sn2, sn2, Binv, ldb, alpha, wv, partial, nt, mu, b0)
        switch (cp.kind) {
            case 0: HLAUNCH(0); break;
            case 1: HLAUNCH(1); break;                // the plain ARD kinds with cs.ard_grad_diff: K in the difference form, the
            case 6: HLAUNCH(6); break;                // per-coordinate sums on the matrix cores all the same (ard_dim_reduce)
            case 2: HLAUNCH(2); break;
            case 3: HLAUNCH(3); break;
            case 4: HLAUNCH(4); break;
            case 5: HLAUNCH(5); break;
            default: return -2;
        }
#undef HLAUNCH
    }
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

// out_dev[h] = sum over the nblk blocks, fixed order
int hadamard_final_launch(const double* partial, long nblk, int ncov, double* out_dev, hipStream_t st) {
    hipLaunchKernelGGL(final_reduce_kernel, dim3((unsigned)(ncov + 1)), dim3(256), 0, st, partial, nblk, ncov + 1, out_dev);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int hadamard_reduce_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, int ncov,
                           double sn2, const double* Binv, long ldb, const double* alpha, double* partial,
                           double* out_dev, hipStream_t st, const double* wv, const double* prep) {
    const long nt = np / ST;
    const long nblk = nt * (nt + 1) / 2;
    if (prep) {                                       // the caller already ran hadamard_prepare_launch on this XT (the Gram-form assembly)
        CHK_RC(hadamard_partial_launch(XT, ldp, n, np, dpad, cs, ncov, sn2, Binv, ldb, alpha, wv, partial, prep, 0, nt, st));
        return hadamard_final_launch(partial, nblk, ncov, out_dev, st);
    }
    // ARD leaves: the per-coordinate sums run in the product form on centred coordinates (ard_dim_reduce); the means live
    // behind the per-block partials (hadamard_partial_count leaves room for them)
    double* mu = partial + nblk * (long)(ncov + 1);
    CHK_RC(hadamard_prepare_launch(XT, ldp, n, np, dpad, cs, mu, st));
    CHK_RC(hadamard_partial_launch(XT, ldp, n, np, dpad, cs, ncov, sn2, Binv, ldb, alpha, wv, partial, mu, 0, nt, st));
    return hadamard_final_launch(partial, nblk, ncov, out_dev, st);
}

long hadamard_partial_count(long np, int ncov) {
    const long nt = np / ST;
    return nt * (nt + 1) / 2 * (long)(ncov + 1) + hadamard_prep_count(np);   // + the coordinate means / point norms of the ARD reduce
}

int col_dot_launch(const double* W, long ldw, long n, const double* z, long zs, double scale, double* y,
                   hipStream_t st) {
    hipLaunchKernelGGL(col_dot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, W, ldw, n, z, zs, scale, y);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int logdet_ztz_launch(const double* L, long ldl, long n, const double* z, long zs, double* out, hipStream_t st) {
    hipLaunchKernelGGL(logdet_ztz_kernel, dim3(1), dim3(256), 0, st, L, ldl, n, z, zs, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int dot2_launch(const double* u, const double* v, long n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(dot2_kernel, dim3(1), dim3(256), 0, st, u, v, n, out);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int aug_rhs_launch(const double* y, const double* m, long n, double* F, long ldf, long row, double* rvec,
                   hipStream_t st) {
    hipLaunchKernelGGL(aug_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, m, n, F, ldf, row, rvec);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int zero_strip_launch(double* F, long ldf, long np, long row0, long nrows, hipStream_t st) {
    hipLaunchKernelGGL(zero_upper_strip_kernel, dim3((unsigned)np), dim3(128), 0, st, F, ldf, np, row0, nrows);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
