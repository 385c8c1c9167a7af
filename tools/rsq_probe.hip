// Accuracy of v_rsq_f64 / v_rcp_f64 and of their Newton refinements on gfx950 (standalone: hipcc --offload-arch=gfx950 tools/rsq_probe.hip -o /tmp/rsq_probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    double r1 = r0 * fma(-h * r0, r0, 1.5);
    double r2 = r1 * fma(-h * r1, r1, 1.5);
    // one step in the "residual" form: e = 1 - d r^2 ; r += r * e/2
    double e = fma(-d * r0, r0, 1.0);
    double r1b = fma(r0 * 0.5, e, r0);
    // second-order (Halley-like) single step: r (1 + e/2 + 3 e^2/8)
    double r1c = fma(r0 * e, fma(0.375, e, 0.5), r0);
    double c0 = __builtin_amdgcn_rcp(d);
    double c1 = fma(c0, fma(-d, c0, 1.0), c0);
    double c2 = fma(c1, fma(-d, c1, 1.0), c1);
    double ec = fma(-d, c0, 1.0);
    double c1c = fma(c0 * ec, 1.0 + ec, c0);      // second order in one go
    out[i * 9 + 0] = r0; out[i * 9 + 1] = r1; out[i * 9 + 2] = r2; out[i * 9 + 3] = r1b; out[i * 9 + 4] = r1c;
    out[i * 9 + 5] = c0; out[i * 9 + 6] = c1; out[i * 9 + 7] = c2; out[i * 9 + 8] = c1c;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 40.0); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 9 * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    std::vector<double> o(n * 9); hipMemcpy(o.data(), dout, n * 9 * 8, hipMemcpyDeviceToHost);
    const char* nm[9] = {"rsq raw", "rsq 1 Newton", "rsq 2 Newton", "rsq 1 step residual form", "rsq 1 step 2nd order", "rcp raw", "rcp 1 Newton", "rcp 2 Newton", "rcp 1 step 2nd order"};
    for (int j = 0; j < 9; ++j) {
        long double mx = 0;
        for (int i = 0; i < n; ++i) {
            long double ex = j < 5 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
            long double e = fabsl(((long double)o[i * 9 + j] - ex) / ex);
            if (e > mx) mx = e;
        }
        printf("%-28s max rel err %.3Le  (%.2Lf ulp of 2^-53)\n", nm[j], mx, mx / 1.1102230246251565e-16L);
    }
    return 0;
}
