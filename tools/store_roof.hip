// Store-bandwidth ceiling of the box, for reading the assembly kernel's HBM fraction against what the
// chip delivers to a kernel that does NOTHING but write an N x N fp64 matrix.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_roof tools/store_roof.hip && /tmp/store_roof 16384
// (results of round 4: EXPERIMENTS.md "What the stores of the assembly cost by themselves"; the product's own pattern is also
// timed inside bench.py through pgp_test_store_roof)
//   linear   : grid-stride 16-byte stores over the whole buffer
//   tiles    : 64 x 64 tiles of a row-major N x N matrix (512-byte runs, stride N*8), persistent workgroups —
//              the assembly kernel's store pattern without its arithmetic
//   tiles_sym: upper-triangle tiles, each stored twice (tile and mirror tile), like MODE_SYM
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st16(double* p, double a, double b) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v = {a, b};
    if (NT) __builtin_nontemporal_store(v, (d2*)p);
    else *(d2*)p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void linear_kernel(double* out, long n2, double v) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    long step = (long)gridDim.x * 512;
    for (; i < n2; i += step) st16<NT>(out + i, v, v + 1.0);
}

// one 64 x 64 tile per iteration; thread t: row 4*(t>>4)+r (r = 0..3), columns 4*(t&15) .. +3 (two 16-byte stores)
template <bool NT, bool SYM>
__global__ __launch_bounds__(256) void tile_kernel(double* out, long n, long ntiles, double v) {
    const long nb = n / 64;
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        long bi, bj;
        if (SYM) {
            // row-major walk of the upper triangle
            long rem = ti; bi = 0;
            while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
            bj = bi + rem;
        } else { bi = ti / nb; bj = ti % nb; }
        double* p = out + (bi * 64 + tr * 4) * n + bj * 64 + tc * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st16<NT>(p + r * n, v, v + 1.0);
            st16<NT>(p + r * n + 2, v + 2.0, v + 3.0);
        }
        if (SYM && bi != bj) {
            double* q = out + (bj * 64 + tr * 4) * n + bi * 64 + tc * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st16<NT>(q + r * n, v, v + 1.0);
                st16<NT>(q + r * n + 2, v + 2.0, v + 3.0);
            }
        }
    }
}

// generic run-shaped tiles: a workgroup writes R rows x BD doubles (BD*8-byte runs, row stride n*8), lanes along the run,
// 16 bytes each; persistent over tiles in row-major tile order
template <int BS>
__global__ __launch_bounds__(BS) void runs_kernel(double* out, long n, int R, int BD, long ntiles, double v) {
    const int lpr = BD / 2;                       // lanes per row
    const int t = threadIdx.x;
    const long tpr = n / BD;                      // tiles per tile row
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const long bi = ti / tpr, bj = ti % tpr;
        double* base = out + bi * R * n + bj * BD;
        for (int e = t; e < R * lpr; e += BS) {
            const int r = e / lpr, c = e % lpr;
            st16<false>(base + (long)r * n + 2 * c, v, v + 1.0);
        }
    }
}

// square T x T tiles of the upper triangle, each stored twice (tile + mirror), lanes along the run (T*8-byte runs)
template <int BS>
__global__ __launch_bounds__(BS) void runs_sym_kernel(double* out, long n, int T, long ntiles, double v) {
    const int lpr = T / 2;
    const long nb = n / T;
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        long rem = ti, bi = 0;
        while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
        const long bj = bi + rem;
        double* base = out + bi * T * n + bj * T;
        double* mir = out + bj * T * n + bi * T;
        for (int e = threadIdx.x; e < T * lpr; e += BS) {
            const int r = e / lpr, c = e % lpr;
            st16<false>(base + (long)r * n + 2 * c, v, v + 1.0);
        }
        if (bi != bj)
            for (int e = threadIdx.x; e < T * lpr; e += BS) {
                const int r = e / lpr, c = e % lpr;
                st16<false>(mir + (long)r * n + 2 * c, v, v + 1.0);
            }
    }
}

// table-driven symmetric tiles (T x T, runs mapping): the order of the table is the experiment
template <int BS>
__global__ __launch_bounds__(BS) void tab_sym_kernel(double* out, long n, int T, const int2* tab, long ntiles, double v) {
    const int lpr = T / 2;
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int2 ij = tab[ti];
        const long bi = ij.x, bj = ij.y;
        double* base = out + bi * T * n + bj * T;
        double* mir = out + bj * T * n + bi * T;
        for (int e = threadIdx.x; e < T * lpr; e += BS) {
            const int r = e / lpr, c = e % lpr;
            st16<false>(base + (long)r * n + 2 * c, v, v + 1.0);
        }
        if (bi != bj)
            for (int e = threadIdx.x; e < T * lpr; e += BS) {
                const int r = e / lpr, c = e % lpr;
                st16<false>(mir + (long)r * n + 2 * c, v, v + 1.0);
            }
    }
}

// rectangular symmetric tiles: R x C direct (C*8-byte runs) + C x R mirror (R*8-byte runs); upper block-triangle in units
// of max(R, C) squares, table-driven
template <int BS>
__global__ __launch_bounds__(BS) void rect_sym_kernel(double* out, long n, int R, int C, const int2* tab, long ntiles, double v) {
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int2 ij = tab[ti];
        const long r0 = ij.x, c0 = ij.y;             // element offsets
        double* base = out + r0 * n + c0;
        double* mir = out + c0 * n + r0;
        const int lc = C / 2, lr = R / 2;
        for (int e = threadIdx.x; e < R * lc; e += BS) {
            const int r = e / lc, c = e % lc;
            st16<false>(base + (long)r * n + 2 * c, v, v + 1.0);
        }
        if (r0 + R <= c0)
            for (int e = threadIdx.x; e < C * lr; e += BS) {
                const int r = e / lr, c = e % lr;
                st16<false>(mir + (long)r * n + 2 * c, v, v + 1.0);
            }
    }
}

// tab_sym with a leading dimension ld >= n (is the power-of-two row stride the problem?)
template <int BS>
__global__ __launch_bounds__(BS) void tab_sym_ld_kernel(double* out, long ld, int T, const int2* tab, long ntiles, double v) {
    const int lpr = T / 2;
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int2 ij = tab[ti];
        const long bi = ij.x, bj = ij.y;
        double* base = out + bi * T * ld + bj * T;
        double* mir = out + bj * T * ld + bi * T;
        for (int e = threadIdx.x; e < T * lpr; e += BS) {
            const int r = e / lpr, c = e % lpr;
            st16<false>(base + (long)r * ld + 2 * c, v, v + 1.0);
        }
        if (bi != bj)
            for (int e = threadIdx.x; e < T * lpr; e += BS) {
                const int r = e / lpr, c = e % lpr;
                st16<false>(mir + (long)r * ld + 2 * c, v, v + 1.0);
            }
    }
}

// a workgroup writes a contiguous chunk of CH doubles, persistent over chunks
template <int BS>
__global__ __launch_bounds__(BS) void chunk_kernel(double* out, long n2, long CH, double v) {
    const long nch = n2 / CH;
    for (long ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        double* base = out + ch * CH;
        for (long e = threadIdx.x * 2; e < CH; e += BS * 2) st16<false>(base + e, v, v + 1.0);
    }
}

// each lane writes 64 contiguous bytes (four 16-byte stores): a wave covers 4 KB per "row" of stores
__global__ __launch_bounds__(256) void lane64_kernel(double* out, long n2, double v) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    long step = (long)gridDim.x * 2048;
    for (; i < n2; i += step) {
        st16<false>(out + i, v, v + 1.0); st16<false>(out + i + 2, v, v + 1.0);
        st16<false>(out + i + 4, v, v + 1.0); st16<false>(out + i + 6, v, v + 1.0);
    }
}

// 8-byte stores, grid-stride
__global__ __launch_bounds__(256) void linear8_kernel(double* out, long n2, double v) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long step = (long)gridDim.x * 256;
    for (; i < n2; i += step) out[i] = v;
}

template <int BS>
__global__ __launch_bounds__(BS) void linear_bs_kernel(double* out, long n2, double v) {
    long i = ((long)blockIdx.x * BS + threadIdx.x) * 2;
    long step = (long)gridDim.x * BS * 2;
    for (; i < n2; i += step) st16<false>(out + i, v, v + 1.0);
}

template <class F>
static double time_ms(F launch, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[i], a, b));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 16384;
    long n2 = n * n; double gb = n2 * 8.0 / 1e9;
    double* out; CK(hipMalloc(&out, n2 * 8));
    printf("N = %ld, %.3f GB per pass; fractions of 8 TB/s\n", n, gb);
    auto rep = [&](const char* name, double ms) { printf("%-34s %.4f ms  %.2f TB/s  %.3f\n", name, ms, gb / ms, gb / ms / 8.0); };
    rep("hipMemsetAsync", time_ms([&] { CK(hipMemsetAsync(out, 0, n2 * 8)); }));
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "linear plain grid %d", grid);
        rep(nm, time_ms([&] { linear_kernel<false><<<grid, 256>>>(out, n2, 1.0); }));
        snprintf(nm, 64, "linear nt    grid %d", grid);
        rep(nm, time_ms([&] { linear_kernel<true><<<grid, 256>>>(out, n2, 1.0); }));
    }
    long nb = n / 64;
    for (int grid : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "tiles plain grid %d", grid);
        rep(nm, time_ms([&] { tile_kernel<false, false><<<grid, 256>>>(out, n, nb * nb, 1.0); }));
        snprintf(nm, 64, "tiles nt    grid %d", grid);
        rep(nm, time_ms([&] { tile_kernel<true, false><<<grid, 256>>>(out, n, nb * nb, 1.0); }));
        snprintf(nm, 64, "tiles_sym plain grid %d", grid);
        rep(nm, time_ms([&] { tile_kernel<false, true><<<grid, 256>>>(out, n, nb * (nb + 1) / 2, 1.0); }));
        snprintf(nm, 64, "tiles_sym nt    grid %d", grid);
        rep(nm, time_ms([&] { tile_kernel<true, true><<<grid, 256>>>(out, n, nb * (nb + 1) / 2, 1.0); }));
    }
    rep("hipMemsetD32Async value 0x3ff00000", time_ms([&] { CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x3ff00000, n2 * 2)); }));
    for (long grid : {32768L, 65536L, 131072L, n2 / 512}) {
        char nm[64];
        snprintf(nm, 64, "linear plain grid %ld", grid);
        rep(nm, time_ms([&] { linear_kernel<false><<<grid, 256>>>(out, n2, 1.0); }));
    }
    for (long grid : {512L, 1024L, 2048L, 8192L, n2 / 2048}) {
        char nm[64];
        snprintf(nm, 64, "linear bs1024 grid %ld", grid);
        rep(nm, time_ms([&] { linear_bs_kernel<1024><<<grid, 1024>>>(out, n2, 1.0); }));
    }
    for (long grid : {2048L, 4096L, 16384L, n2 / 128}) {
        char nm[64];
        snprintf(nm, 64, "linear bs64 grid %ld", grid);
        rep(nm, time_ms([&] { linear_bs_kernel<64><<<grid, 64>>>(out, n2, 1.0); }));
    }
    for (long grid : {2048L, 16384L, 65536L}) {
        char nm[64];
        snprintf(nm, 64, "linear 8-byte grid %ld", grid);
        rep(nm, time_ms([&] { linear8_kernel<<<grid, 256>>>(out, n2, 1.0); }));
        snprintf(nm, 64, "lane64 grid %ld", grid);
        rep(nm, time_ms([&] { lane64_kernel<<<grid, 256>>>(out, n2, 1.0); }));
    }
    for (long chkb : {16L, 64L, 256L, 1024L, 4096L})
        for (long grid : {2048L, 8192L, 1L << 30}) {
            long CH = chkb * 128, nch = n2 / CH;
            long g = grid < nch ? grid : nch;
            char nm[64];
            snprintf(nm, 64, "chunk %ld KB grid %ld", chkb, g);
            rep(nm, time_ms([&] { chunk_kernel<256><<<g, 256>>>(out, n2, CH, 1.0); }));
        }
    struct { int R, BD; } shapes[] = {{64, 64}, {64, 128}, {128, 128}, {32, 256}, {16, 512}, {64, 256}, {8, 2048}, {1, 16384}, {256, 64}, {128, 32}, {64, 32}};
    for (auto sh : shapes)
        for (long grid : {2048L, 8192L, 1L << 30}) {
            if (n % sh.BD || n % sh.R) continue;
            long ntl = (n / sh.R) * (n / sh.BD);
            long g = grid < ntl ? grid : ntl;
            char nm[64];
            snprintf(nm, 64, "runs %d rows x %d B grid %ld", sh.R, sh.BD * 8, g);
            rep(nm, time_ms([&] { runs_kernel<256><<<g, 256>>>(out, n, sh.R, sh.BD, ntl, 1.0); }));
        }
    for (int T : {64, 128, 256})
        for (long grid : {2048L, 4096L, 8192L, 1L << 30}) {
            long nbt = n / T, ntl = nbt * (nbt + 1) / 2;
            long g = grid < ntl ? grid : ntl;
            char nm[64];
            snprintf(nm, 64, "runs_sym T=%d grid %ld", T, g);
            rep(nm, time_ms([&] { runs_sym_kernel<256><<<g, 256>>>(out, n, T, ntl, 1.0); }));
        }
    for (int T : {64, 128})
        for (long S : {1L, 2L, 4L, 8L, 16L, 32L, 64L}) {
            const long ntr = n / T, nst = (ntr + S - 1) / S;
            std::vector<int2> h;
            for (long SI = 0; SI < nst; ++SI)
                for (long SJ = SI; SJ < nst; ++SJ)
                    for (long i = SI * S; i < std::min(ntr, SI * S + S); ++i)
                        for (long j = std::max(i, SJ * S); j < std::min(ntr, SJ * S + S); ++j) h.push_back(make_int2((int)i, (int)j));
            int2* d; CK(hipMalloc(&d, h.size() * sizeof(int2)));
            CK(hipMemcpy(d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
            for (long grid : {2048L, 4096L, 1L << 30}) {
                long ntl = (long)h.size();
                long g = grid < ntl ? grid : ntl;
                char nm[64];
                snprintf(nm, 64, "tab_sym T=%d super %ld grid %ld", T, S, g);
                rep(nm, time_ms([&] { tab_sym_kernel<256><<<g, 256>>>(out, n, T, d, ntl, 1.0); }));
            }
            CK(hipFree(d));
        }
    {
        struct { int R, C; } sh[] = {{64, 128}, {64, 256}, {32, 128}, {128, 64}, {32, 256}, {16, 256}, {64, 512}};
        for (auto q : sh) {
            // rows in steps of R, columns in steps of C, tiles with c0 + C > r0 (on / above the diagonal band); diagonal-crossing
            // tiles write their direct part only (the probe slightly under-writes the strictly-lower part there: < 1 %)
            std::vector<int2> h;
            for (long r0 = 0; r0 < n; r0 += q.R)
                for (long c0 = (r0 / q.C) * q.C; c0 < n; c0 += q.C) h.push_back(make_int2((int)r0, (int)c0));
            int2* d; CK(hipMalloc(&d, h.size() * sizeof(int2)));
            CK(hipMemcpy(d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
            for (long grid : {2048L, 4096L, 8192L, 1L << 30}) {
                long ntl = (long)h.size();
                long g = grid < ntl ? grid : ntl;
                char nm[64];
                snprintf(nm, 64, "rect_sym %dx%d grid %ld", q.R, q.C, g);
                rep(nm, time_ms([&] { rect_sym_kernel<256><<<g, 256>>>(out, n, q.R, q.C, d, ntl, 1.0); }));
            }
            CK(hipFree(d));
        }
    }
    {
        double* big; CK(hipMalloc(&big, (size_t)n * (n + 4096) * 8));
        const long T = 64, S = 8, ntr = n / T, nst = (ntr + S - 1) / S;
        std::vector<int2> h;
        for (long SI = 0; SI < nst; ++SI)
            for (long SJ = SI; SJ < nst; ++SJ)
                for (long i = SI * S; i < std::min(ntr, SI * S + S); ++i)
                    for (long j = std::max(i, SJ * S); j < std::min(ntr, SJ * S + S); ++j) h.push_back(make_int2((int)i, (int)j));
        int2* d; CK(hipMalloc(&d, h.size() * sizeof(int2)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
        for (int rep_ = 0; rep_ < 2; ++rep_)
        for (long pad : {0L, 16L, 32L, 64L, 128L, 256L, 512L, 1024L, 2048L, 48L, 80L, 4096L - 64L}) {
            char nm[64];
            snprintf(nm, 64, "tab_sym T=64 ld = n + %ld grid 4096", pad);
            rep(nm, time_ms([&] { tab_sym_ld_kernel<256><<<4096, 256>>>(big, n + pad, 64, d, (long)h.size(), 1.0); }));
        }
        CK(hipFree(d)); CK(hipFree(big));
    }
    CK(hipFree(out));
    return 0;
}
