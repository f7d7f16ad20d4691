// linebench.hip - how many scattered 128-byte lines per second does the memory system of this box deliver? The hot-pixel patch kernel
// (merge_patch_hot) touches ~44 lines of 128 bytes per hot element, 8-byte pieces of each: this probe does nothing else - every lane
// reads 8 bytes of a DIFFERENT line of a 3.2 GB buffer (the size of config 3's inputs), lines visited in a full-period pseudo-random
// order (or in order, for the streaming reference), UN independent loads in flight per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/linebench.hip -o tools/bin/linebench && tools/bin/linebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// a bijection of [0, 2^25) with no structure left (multiply / xor-shift rounds, each one invertible on 25 bits)
__device__ __forceinline__ uint64_t scramble25(uint64_t x) {
    const uint64_t m = (1ull << 25) - 1;
    x = (x * 2654435761ull) & m; x ^= x >> 12;
    x = (x * 0x9E3779B1ull) & m; x ^= x >> 13;
    x = (x * 0x85EBCA6Bull) & m; x ^= x >> 11;
    return x & m;
}

template <int UN, int RANDOM>      // 0: lines in order, 1: a constant odd stride (i * mul mod n), 2: scrambled
__global__ __launch_bounds__(256) void k(const double* buf, uint64_t n_lines, uint64_t mul, uint64_t per_thread, double* out) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    const uint64_t nthreads = static_cast<uint64_t>(gridDim.x) * 256;
    double acc = 0.0;
    for (uint64_t it = 0; it < per_thread; it += UN) {
        double v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint64_t i = (it + u) * nthreads + tid;                       // consecutive lanes: consecutive i
            const uint64_t line = RANDOM == 2 ? scramble25(i % n_lines) : RANDOM == 1 ? (i * mul) % n_lines : i % n_lines;   // permutations of the lines
            v[u] = __builtin_nontemporal_load(buf + line * 16);                 // 8 bytes of a 128-byte line
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u];
    }
    if (acc == 123.456) out[tid] = acc;
}

template <int UN, int RANDOM>
static void run(const char* name, const double* buf, uint64_t n_lines, double* out, int wg_per_cu) {
    const uint64_t grid = 256ull * wg_per_cu, nthreads = grid * 256;
    const uint64_t per_thread = ((n_lines + nthreads - 1) / nthreads + UN - 1) / UN * UN;   // every line about once
    const uint64_t mul = 2654435761ull;                                                      // odd; n_lines is a power of two below
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<UN, RANDOM>), dim3(grid), dim3(256), 0, 0, buf, n_lines, mul, per_thread, out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<UN, RANDOM>), dim3(grid), dim3(256), 0, 0, buf, n_lines, mul, per_thread, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lines = static_cast<double>(per_thread) * nthreads;
    const double us = ms * 1e3 / reps;
    printf("%-44s UN=%d wg/cu=%2d: %8.1f us for %.1f M lines = %6.1f G lines/s = %5.2f TB/s of 128-byte lines (%5.2f TB/s of the 8 bytes used)\n",
           name, UN, wg_per_cu, us, lines / 1e6, lines / us / 1e3, lines * 128 / us / 1e6, lines * 8 / us / 1e6);
}

int main() {
    const uint64_t n_lines = 1ull << 25;                       // 2^25 lines x 128 B = 4.3 GB
    double *buf, *out;
    hipMalloc(&buf, n_lines * 128);
    hipMalloc(&out, 256ull * 32 * 256 * 8);
    hipMemset(buf, 0, n_lines * 128);
    for (int wg : {8, 16, 32}) {
        run<4, 2>("scrambled lines (one 8-byte load per line)", buf, n_lines, out, wg);
        run<16, 2>("scrambled lines (one 8-byte load per line)", buf, n_lines, out, wg);
    }
    run<16, 1>("constant odd stride (one 8-byte load per line)", buf, n_lines, out, 16);
    run<16, 0>("lines in order (one 8-byte load per line)", buf, n_lines, out, 16);
    hipFree(buf); hipFree(out);
    return 0;
}
