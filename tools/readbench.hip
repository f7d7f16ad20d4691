// readbench.hip - read-only streaming ceiling for the reduction kernels (hm_stats.hip): S float64 streams summed, VEC doubles per
// lane and load, UN wave-chunks in flight per stream, nontemporal or plain loads, G workgroups. Prints TB/s per shape.
//   hipcc -O3 --offload-arch=gfx950 tools/readbench.hip -o tools/bin/readbench && tools/bin/readbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct P { const double* s[4]; int64_t n; double* out; };
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int S, int VEC, bool NT, int UN>
__global__ __launch_bounds__(256) void k(const P a) {
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t chunk = 64 * VEC;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * chunk;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4 * chunk;
    double acc = 0.0;
    for (int64_t b = wave0; b + (UN - 1) * stride + chunk <= a.n; b += UN * stride) {
        double v[S][UN][VEC];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double* p = a.s[s] + b + u * stride + lane * VEC;
                if constexpr (VEC == 2) {
                    const f64x2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(p)) : *reinterpret_cast<const f64x2*>(p);
                    v[s][u][0] = t.x; v[s][u][1] = t.y;
                } else {
                    v[s][u][0] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc += v[s][u][e];
    }
    a.out[static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x] = acc;
}

template <int S, int VEC, bool NT, int UN>
static void run(const P& p, int grid, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<S, VEC, NT, UN>), dim3(grid), dim3(256), 0, 0, p);
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<S, VEC, NT, UN>), dim3(grid), dim3(256), 0, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / 20);
    }
    std::sort(t.begin(), t.end());
    const double bytes = static_cast<double>(S) * p.n * 8;
    printf("%-34s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, t[2] * 1e3, bytes / (t[2] * 1e-3) / 1e12);
    fflush(stdout);
}

__global__ __launch_bounds__(256) void k_fill(double* o, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n / 2; q += stride) {
        f64x2 v; v.x = 1.0; v.y = 2.0;
        __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(o) + q);
    }
}

__global__ __launch_bounds__(256) void k_copy(const double* in, double* o, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n / 2; q += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f64x2*>(in) + q), reinterpret_cast<f64x2*>(o) + q);
}

// copy with B x 16 bytes per lane read before the B stores (a wave moves B KB in, then B KB out), nontemporal or plain
template <int B, bool NT>
__global__ __launch_bounds__(256) void k_copy_b(const double* in, double* o, int64_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t chunk = 128ll * B;                                   // doubles per wave iteration
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * chunk;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4 * chunk;
    for (int64_t b = wave0; b + chunk <= n; b += stride) {
        f64x2 v[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const f64x2* p = reinterpret_cast<const f64x2*>(in + b + 128 * k) + lane;
            v[k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            f64x2* p = reinterpret_cast<f64x2*>(o + b + 128 * k) + lane;
            if (NT) __builtin_nontemporal_store(v[k], p); else *p = v[k];
        }
    }
}
template <int B, bool NT>
static void run_copy(const double* in, double* out, int64_t n, int grid, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_copy_b<B, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_copy_b<B, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / 20);
    }
    std::sort(t.begin(), t.end());
    printf("copy %-26s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, t[2] * 1e3, 2.0 * n * 8 / (t[2] * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t n = 4096ll * 4096 * 3;
    if (argc > 1 && argv[1][0] == 'c') {            // readbench copysweep
        double *a, *b;
        (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8);
        (void)hipMemset(a, 0, n * 8);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL((k_copy_b<1, true>), dim3(2048), dim3(256), 0, 0, a, b, n);
        (void)hipDeviceSynchronize();
        for (int grid : {768, 1024, 2048, 4096}) {
            run_copy<1, true>(a, b, n, grid, "1 KB/wave nt");
            run_copy<2, true>(a, b, n, grid, "2 KB/wave nt");
            run_copy<4, true>(a, b, n, grid, "4 KB/wave nt");
            run_copy<8, true>(a, b, n, grid, "8 KB/wave nt");
            run_copy<16, true>(a, b, n, grid, "16 KB/wave nt");
            run_copy<4, false>(a, b, n, grid, "4 KB/wave plain");
        }
        return 0;
    }
    if (argc > 2) {            // readbench hold <read|fill> : keep one shape running for ~8 s (power / clock polling from outside)
        P q{};
        q.n = n;
        for (int s = 0; s < 4; ++s) { (void)hipMalloc(const_cast<double**>(&q.s[s]), n * 8); (void)hipMemset(const_cast<double*>(q.s[s]), 0, n * 8); }
        (void)hipMalloc(&q.out, 8192 * 256 * 8);
        const bool fill = argv[2][0] == 'f', copy = argv[2][0] == 'c';
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        const int reps = copy ? 60000 : 120000;
        for (int i = 0; i < reps; ++i) {
            if (copy) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, q.s[i & 1], const_cast<double*>(q.s[2 + (i & 1)]), n);
            else if (fill) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, const_cast<double*>(q.s[i & 3]), n);
            else hipLaunchKernelGGL((k<1, 1, true, 4>), dim3(1024), dim3(256), 0, 0, P{{q.s[i & 3], nullptr, nullptr, nullptr}, n, q.out});
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.2f TB/s over %.1f s\n", copy ? "copy (403 MB in + 403 MB out)" : fill ? "fill (403 MB)" : "read (403 MB)", ms / reps * 1e3,
               (copy ? 2 : 1) * n * 8.0 / (ms / reps * 1e-3) / 1e12, ms / 1e3);
        return 0;
    }
    P p{};
    p.n = n;
    for (int s = 0; s < 4; ++s) { hipMalloc(const_cast<double**>(&p.s[s]), n * 8); hipMemset(const_cast<double*>(p.s[s]), 0, n * 8); }
    hipMalloc(&p.out, 8192 * 256 * 8);
    // warm the clock
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL((k<1, 1, true, 4>), dim3(2048), dim3(256), 0, 0, p);
    hipDeviceSynchronize();
    for (int grid : {1024, 2040, 4096, 8192}) {
        run<1, 1, true, 4>(p, grid, "1 stream  8B nt UN4");
        run<1, 1, true, 8>(p, grid, "1 stream  8B nt UN8");
        run<1, 2, true, 4>(p, grid, "1 stream 16B nt UN4");
        run<1, 2, true, 8>(p, grid, "1 stream 16B nt UN8");
        run<1, 2, false, 4>(p, grid, "1 stream 16B plain UN4");
        run<2, 1, true, 4>(p, grid, "2 streams 8B nt UN4");
        run<2, 2, true, 4>(p, grid, "2 streams 16B nt UN4");
        run<4, 1, true, 2>(p, grid, "4 streams 8B nt UN2");
        run<4, 1, true, 4>(p, grid, "4 streams 8B nt UN4");
        run<4, 2, true, 2>(p, grid, "4 streams 16B nt UN2");
        run<4, 2, false, 2>(p, grid, "4 streams 16B plain UN2");
    }
    return 0;
}
