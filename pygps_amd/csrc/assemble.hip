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

template <int MODE>
__global__ __launch_bounds__(256) void cov_tile_kernel(const double* __restrict__ XrT, long ldr, long n,
                                                       const double* __restrict__ XcT, long ldc, long m, int dpad,
                                                       CovParams cp, double inv_sn2, double* __restrict__ out,
                                                       long ldo, long ntile_c) {
    constexpr int TS = ST + 2;                      // transpose-tile row stride (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) double sm[MODE == MODE_SYM ? ST * TS : 2 * SKC * ST];
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
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long r = r0 + 4 * tr + a;
            const long c = c0 + 2 * tc + (b & 1) + 32 * (b >> 1);
            double val;
            if (cp.der < 0) {
                val = cov_value(cp, s[a][b]);
            } else {
                double dk2 = 0.0;
                if (cp.kind == 1 && cp.der < cp.D) {
                    const double dd = XrT[(long)cp.der * ldr + r] - XcT[(long)cp.der * ldc + c];
                    dk2 = dd * dd;
                }
                val = cov_deriv(cp, s[a][b], dk2);
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

int cov_sym_launch(const double* XT, long ldp, long n, int dpad, const CovParams& cp, double* out, hipStream_t st,
                   long ldo) {
    const long nt = (n + ST - 1) / ST;
    const long nblk = nt * (nt + 1) / 2;
    hipLaunchKernelGGL((cov_tile_kernel<MODE_SYM>), dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, XT, ldp, n,
                       dpad, cp, 0.0, out, ldo > 0 ? ldo : n, nt);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int cov_rect_launch(const double* XrT, long ldr, long n, const double* XcT, long ldc, long m, int dpad,
                    const CovParams& cp, double* out, long ldo, hipStream_t st) {
    const long ntr = (n + ST - 1) / ST, ntc = (m + ST - 1) / ST;
    hipLaunchKernelGGL((cov_tile_kernel<MODE_RECT>), dim3((unsigned)(ntr * ntc)), dim3(256), 0, st, XrT, ldr, n, XcT,
                       ldc, m, dpad, cp, 0.0, out, ldo, ntc);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int cov_factor_launch(const double* XT, long ldp, long n, long np, int dpad, const CovParams& cp, double inv_sn2,
                      double* Bf, long ldf, hipStream_t st) {
    const long nt = np / ST;
    const long nblk = nt * (nt + 1) / 2;
    hipLaunchKernelGGL((cov_tile_kernel<MODE_FACTOR>), dim3((unsigned)nblk), dim3(256), 0, st, XT, ldp, n, XT, ldp, n,
                       dpad, cp, inv_sn2, Bf, ldf, nt);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}

int self_fill_launch(double* out, long m, double val, hipStream_t st) {
    hipLaunchKernelGGL(self_fill_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, out, m, val);
    return hipGetLastError() == hipSuccess ? PGP_OK : PGP_ERR_HIP;
}
