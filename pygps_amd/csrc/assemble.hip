// Kernel-matrix construction (reference: Core/cov.py RBF :796-828, RBFard :887-938, Matern
// :1124-1182 -- scipy cdist('sqeuclidean') + numpy exp/sqrt passes, replaced by ONE fused tile kernel).
//
//   scale_transpose_kernel  x (n,d) row-major  ->  XsT (dpad, ldp) k-major, scaled by the kernel's
//                           length-scales (x/ell, x*(1/ell_k), sqrt(d) x/ell), zero padded.
//   cov_tile_kernel<MODE>   64x64 output tile per workgroup; X slabs staged through LDS with
//                           coalesced 512-byte runs; distance in difference form; scalar map
//                           (exp / Matern polynomial / derivative) fused; HBM-write bound:
//                           algorithmic bytes = 8 n m (+ 8 (n+m) d).
//        MODE_SYM    'train': only tiles on/above the diagonal are computed, each is stored twice
//                    (direct + mirrored) -> full symmetric (n,n) numpy array.
//        MODE_RECT   'cross': (n,m).
//        MODE_FACTOR fused assembly of B = K/sn2 + I straight into the (padded) factor buffer that
//                    the Cholesky overwrites: only the row-major upper triangle (= column-major
//                    lower) is written, padding rows/cols get the identity.
#include "kernels.h"
#include "sqdist_tile.h"

namespace {

__global__ void scale_transpose_kernel(const double* __restrict__ x, long n, int d, const double* __restrict__ scale,
                                       double* __restrict__ XsT, long ldp, int dpad) {
    // one thread per (k, p); p fastest for coalesced stores.  x reads are strided but x is tiny (n*d).
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= ldp) return;
    double v = 0.0;
    if (p < n && k < d) v = x[p * d + k] * scale[k];
    XsT[(long)k * ldp + p] = v;
    (void)dpad;
}

enum { MODE_SYM = 0, MODE_RECT = 1, MODE_FACTOR = 2 };

template <class COV> struct is_program { static constexpr bool value = false; };
template <> struct is_program<CovProgram> { static constexpr bool value = true; };

// COV = CovParams (one functor, the hot path) or CovProgram (Sum/Product/Scale tree; elements are evaluated in a
// rolled loop over LDS-staged distances so that the eight leaf functors are instantiated once, not 16 times)
template <int MODE, class COV>
__global__ __launch_bounds__(256) void cov_tile_kernel(const double* __restrict__ XrT, long ldr, long n,
                                                       const double* __restrict__ XcT, long ldc, long m, int dpad,
                                                       COV cp, double inv_sn2, double* __restrict__ out,
                                                       long ldo, long ntile_c) {
    constexpr int TS = ST + 2;                      // transpose-tile row stride (16-byte aligned rows)
    constexpr bool PROG = is_program<COV>::value;
    constexpr int SMN = MODE == MODE_SYM ? ST * TS : (PROG ? 16 * 256 : 2 * SKC * ST);
    __shared__ __attribute__((aligned(16))) double sm[SMN];
    long ti, tj;
    if (MODE == MODE_RECT) {
        ti = blockIdx.x / ntile_c;
        tj = blockIdx.x % ntile_c;
    } else {                                       // packed upper-triangular tile index (tj >= ti)
        const long b = blockIdx.x;
        const long nt = ntile_c;
        // row ti holds (nt - ti) tiles; invert the prefix sum b = ti*nt - ti(ti-1)/2 + (tj - ti)
        long r = (long)(((2.0 * nt + 1.0) - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)b)) * 0.5);
        if (r < 0) r = 0;
        while (r > 0 && r * nt - r * (r - 1) / 2 > b) --r;
        while ((r + 1) * nt - (r + 1) * r / 2 <= b) ++r;
        ti = r;
        tj = ti + (b - (r * nt - r * (r - 1) / 2));
    }
    const long r0 = ti * ST, c0 = tj * ST;
    double s[4][4];
    sqdist_tile(XrT, ldr, r0, XcT, ldc, c0, dpad, sm, s);

    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    double v[4][4];
    if (PROG) {                                     // sm is free: sqdist_tile ends with a barrier
        double* sv = sm + t;
#pragma unroll
        for (int e = 0; e < 16; ++e) sv[e * 256] = s[e >> 2][e & 3];
#pragma unroll 1
        for (int e = 0; e < 16; ++e) {
            const int a = e >> 2, b = e & 3;
            const long r = r0 + 4 * tr + a;
            const long c = c0 + 2 * tc + (b & 1) + 32 * (b >> 1);
            sv[e * 256] = cov_elem(cp, sv[e * 256], 0.0, MODE != MODE_RECT && r == c);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e >> 2][e & 3] = sv[e * 256];
    }
    const int ard_der = cov_ard_der(cp);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long r = r0 + 4 * tr + a;
            const long c = c0 + 2 * tc + (b & 1) + 32 * (b >> 1);
            double val;
            if (PROG) {
                val = s[a][b];
            } else {
                double dk2 = 0.0;
                if (ard_der >= 0) {
                    const double dd = XrT[(long)ard_der * ldr + r] - XcT[(long)ard_der * ldc + c];
                    dk2 = dd * dd;
                }
                val = cov_elem(cp, s[a][b], dk2, MODE != MODE_RECT && r == c);
            }
            if (MODE == MODE_FACTOR) {
                if (r < n && c < n) val = val * inv_sn2 + (r == c ? 1.0 : 0.0);
                else val = (r == c) ? 1.0 : 0.0;
            }
            v[a][b] = val;
        }
    // direct store: row-major, double2 along c
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long r = r0 + 4 * tr + a;
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            const long c = c0 + 2 * tc + 32 * bh;
            if (MODE == MODE_FACTOR) {
                if (ti != tj) {
                    *(double2_t*)(out + r * ldo + c) = double2_t{v[a][2 * bh], v[a][2 * bh + 1]};
                } else {                            // diagonal tile: keep exact zeros below the diagonal
                    if (c >= r) out[r * ldo + c] = v[a][2 * bh];
                    if (c + 1 >= r) out[r * ldo + c + 1] = v[a][2 * bh + 1];
                }
            } else {
                if (r < n) {
                    if (c + 1 < m && ((ldo & 1) == 0)) {
                        *(double2_t*)(out + r * ldo + c) = double2_t{v[a][2 * bh], v[a][2 * bh + 1]};
                    } else {
                        if (c < m) out[r * ldo + c] = v[a][2 * bh];
                        if (c + 1 < m) out[r * ldo + c + 1] = v[a][2 * bh + 1];
                    }
                }
            }
        }
    }
    if (MODE == MODE_SYM && ti != tj) {
        // mirrored store out[c][r]: transpose the tile through LDS so that the global stores are 512-byte
        // contiguous runs (32 lanes x 16 B) instead of 32-byte pieces with a row stride between lanes
        __syncthreads();                            // sm is free again (sqdist_tile ends with a barrier)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int cl = 2 * tc + (b & 1) + 32 * (b >> 1);
            *(double2_t*)(sm + cl * TS + 4 * tr) = double2_t{v[0][b], v[1][b]};
            *(double2_t*)(sm + cl * TS + 4 * tr + 2) = double2_t{v[2][b], v[3][b]};
        }
        __syncthreads();
        const int pr = t & 31, rw = t >> 5;         // 8 tile rows per pass, 32 double2 per row
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int cl = p * 8 + rw;
            const long c = c0 + cl, r = r0 + 2 * pr;
            if (c < m) {
                const double2_t val = *(const double2_t*)(sm + cl * TS + 2 * pr);
                if (r + 1 < n && ((ldo & 1) == 0)) *(double2_t*)(out + c * ldo + r) = val;
                else {
                    if (r < n) out[c * ldo + r] = val[0];
                    if (r + 1 < n) out[c * ldo + r + 1] = val[1];
                }
            }
        }
    }
}

__global__ void self_fill_kernel(double* out, long m, double val) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = val;
}

}  // namespace

int scale_transpose_launch(const double* x, long n, int d, const double* scale_dev, double* XsT, long ldp, int dpad,
                           hipStream_t st) {
    dim3 grid((unsigned)((ldp + 255) / 256), (unsigned)dpad);
    hipLaunchKernelGGL(scale_transpose_kernel, grid, dim3(256), 0, st, x, n, d, scale_dev, XsT, ldp, dpad);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

template <int MODE>
static int cov_tile_dispatch(const CovSpec& cs, int train, unsigned nblk, hipStream_t st, const double* XrT, long ldr,
                             long n, const double* XcT, long ldc, long m, int dpad, double inv_sn2, double* out,
                             long ldo, long ntile_c) {
    if (cs.prog) {
        CovProgram pg = cs.pg;
        for (int l = 0; l < pg.nleaf; ++l) pg.leaf[l].train = train;
        hipLaunchKernelGGL((cov_tile_kernel<MODE, CovProgram>), dim3(nblk), dim3(256), 0, st, XrT, ldr, n, XcT, ldc, m,
                           dpad, pg, inv_sn2, out, ldo, ntile_c);
    } else {
        CovParams cp = cs.cp;
        cp.train = train;
        if (cp.kind == 6 && cp.ref_der) cp.gb = train == 1 ? 0.0 : cs.ell4;       // Core/cov.py:1415-1418 as returned
        hipLaunchKernelGGL((cov_tile_kernel<MODE, CovParams>), dim3(nblk), dim3(256), 0, st, XrT, ldr, n, XcT, ldc, m,
                           dpad, cp, inv_sn2, out, ldo, ntile_c);
    }
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int cov_sym_launch(const double* XT, long ldp, long n, int dpad, const CovSpec& cs, double* out, hipStream_t st,
                   long ldo) {
    const long nt = (n + ST - 1) / ST;
    const long nblk = nt * (nt + 1) / 2;
    return cov_tile_dispatch<MODE_SYM>(cs, 1, (unsigned)nblk, st, XT, ldp, n, XT, ldp, n, dpad, 0.0, out,
                                       ldo > 0 ? ldo : n, nt);
}

int cov_rect_launch(const double* XrT, long ldr, long n, const double* XcT, long ldc, long m, int dpad,
                    const CovSpec& cs, double* out, long ldo, hipStream_t st) {
    const long ntr = (n + ST - 1) / ST, ntc = (m + ST - 1) / ST;
    return cov_tile_dispatch<MODE_RECT>(cs, 0, (unsigned)(ntr * ntc), st, XrT, ldr, n, XcT, ldc, m, dpad, 0.0, out, ldo,
                                        ntc);
}

int cov_factor_launch(const double* XT, long ldp, long n, long np, int dpad, const CovSpec& cs, double inv_sn2,
                      double* Bf, long ldf, hipStream_t st) {
    const long nt = np / ST;
    const long nblk = nt * (nt + 1) / 2;
    return cov_tile_dispatch<MODE_FACTOR>(cs, 1, (unsigned)nblk, st, XT, ldp, n, XT, ldp, n, dpad, inv_sn2, Bf, ldf, nt);
}

// k(z,z) and its derivatives at zero distance ('self_test' mode): one evaluation of the functor
template <class COV>
__global__ void cov_self_kernel(COV cp, int same, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = cov_elem(cp, 0.0, 0.0, same != 0);
}

// train = 1: a diagonal entry of the training matrix; train = 2: 'self_test'
int cov_self_launch(const CovSpec& cs, int train, double* out_dev, hipStream_t st) {
    if (cs.prog) {
        CovProgram pg = cs.pg;
        for (int l = 0; l < pg.nleaf; ++l) pg.leaf[l].train = train;
        hipLaunchKernelGGL((cov_self_kernel<CovProgram>), dim3(1), dim3(64), 0, st, pg, train == 1, out_dev);
    } else {
        CovParams cp = cs.cp;
        cp.train = train;
        hipLaunchKernelGGL((cov_self_kernel<CovParams>), dim3(1), dim3(64), 0, st, cp, train == 1, out_dev);
    }
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int self_fill_launch(double* out, long m, double val, hipStream_t st) {
    hipLaunchKernelGGL(self_fill_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, out, m, val);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
