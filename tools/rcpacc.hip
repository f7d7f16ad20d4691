// rcpacc.hip - accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the estimate + 1 or 2 Newton steps (hm_stats.hip: rcp_newton,
// rsq_newton), in ulp of the correctly rounded result, over 2^22 random operands. Reference: long double on the host.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/rcpacc.hip -o tools/bin/rcpacc && tools/bin/rcpacc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>

__global__ void k(const double* x, double* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double r0 = __builtin_amdgcn_rcp(v);
    const double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    const double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    const double y0 = __builtin_amdgcn_rsq(v);
    const double h = 0.5 * v;
    const double y1 = y0 * fma(-h * y0, y0, 1.5);
    const double y2 = y1 * fma(-h * y1, y1, 1.5);
    double* p = o + 6 * static_cast<int64_t>(i);
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = y0; p[4] = y1; p[5] = y2;
}

static double ulps(double got, long double exact) {
    const double e = static_cast<double>(exact);
    int ex; std::frexp(e, &ex);
    const long double ulp = std::ldexp(1.0L, ex - 53);
    return static_cast<double>(fabsl(static_cast<long double>(got) - exact) / ulp);
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> m(1.0, 4.0);
    std::uniform_int_distribution<int> e(-40, 40);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), e(g));
    double *dx, *dout;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, n * 48ll);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> o(6ll * n);
    (void)hipMemcpy(o.data(), dout, n * 48ll, hipMemcpyDeviceToHost);
    double mx[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / x[i], y = 1.0L / sqrtl(static_cast<long double>(x[i]));
        for (int q = 0; q < 3; ++q) { mx[q] = fmax(mx[q], ulps(o[6ll * i + q], r)); mx[3 + q] = fmax(mx[3 + q], ulps(o[6ll * i + 3 + q], y)); }
    }
    printf("max error in ulp over %d operands:\n rcp estimate %.3g, +1 Newton %.3g, +2 Newton %.3g\n rsq estimate %.3g, +1 Newton %.3g, +2 Newton %.3g\n",
           n, mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]);
    return 0;
}
