// burstbench.hip - traffic-shape probe (round 4): does a merge whose waves each read ONE stream in 4 KB bursts (the shape of the box's best
// plain copy) move more bytes than the shipped shape, in which every wave reads a little of every stream?
//   shape 0 "interleaved" (today): a wave owns U x 128 consecutive elements and reads 2 bytes per lane from each of the N frames per
//           sub-unit, then stores 16 bytes per lane per sub-unit (1 KB contiguous per store instruction).
//   shape 1 "wave-per-stream burst": a workgroup of 8 waves owns a tile of T = 4096 elements; wave i < N reads frame i's 4 KB of the tile
//           (four 16-byte loads per lane, back to back) [for N = 15: waves take two frames each], a barrier stands for the LDS hand-off,
//           then every wave stores its 512 elements' 4 KB of float64 output in four back-to-back 16-byte stores per lane.
// No arithmetic, no LDS tables: outputs are meaningless (a sum of the loaded bytes keeps the loads alive). N frames of E bytes, one
// float64 output of E elements; R rotating sets so that no launch finds its inputs in the 256 MB Infinity Cache. Prints us per launch
// and the fraction of 8 TB/s on (N + 8) E bytes.
// build: hipcc -O3 --offload-arch=gfx950 tools/burstbench.hip -o tools/bin/burstbench ; run: tools/bin/burstbench [N=7] [E=50331648]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int MAXF = 16;
struct Ptrs { const uint8_t* in[MAXF]; double* out; uint32_t n_units; };

template <int NF, int U>
__global__ __launch_bounds__(256) void k_interleaved(const Ptrs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * 4u;
    for (uint32_t u = blockIdx.x * 4u + wave; u < a.n_units; u += stride) {
        const size_t base = static_cast<size_t>(u) * (U * 128);
        uint32_t r[NF][U];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int s = 0; s < U; ++s) r[i][s] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(a.in[i] + base + 128 * s + 2 * lane));
#pragma unroll
        for (int s = 0; s < U; ++s) {
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < NF; ++i) acc += r[i][s];
            f64x2 v; v.x = static_cast<double>(acc & 255u); v.y = static_cast<double>(acc >> 8);
            __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(a.out + base + 128 * s) + lane);
        }
    }
}

// tile = 4096 elements; 8 waves; wave w reads frames w, w + 8 (if < NF): 4 KB each = 4 x 16 B per lane at 1 KB pitch per instruction
template <int NF>
__global__ __launch_bounds__(512) void k_burst(const Ptrs a) {
    __shared__ uint32_t hand[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_tiles = a.n_units;                       // tiles of 4096 elements
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = static_cast<size_t>(t) * 4096;
        uint32_t acc = lane;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int i = static_cast<int>(wave) + 8 * f;
            if (i < NF) {                                      // wave-uniform
                u32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.in[i] + base + 1024 * k) + lane);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
            }
        }
        if (lane == 0) hand[wave] = acc;                       // stands for the LDS hand-off of the tile's bytes
        __syncthreads();
        const uint32_t all = hand[0] + hand[(wave + 1) & 7];
        f64x2 o; o.x = static_cast<double>(all & 255u); o.y = static_cast<double>(acc & 255u);
        double* ob = a.out + base + 512 * wave;                // this wave's 512 elements = 4 KB of output
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(ob + 128 * k) + lane);
        __syncthreads();
    }
}

template <typename F> static double time_us(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 7;
    const size_t E = argc > 2 ? static_cast<size_t>(atoll(argv[2])) : 50331648ull;
    const int R = 4;
    if ((N != 7 && N != 15) || E % 4096) { printf("N must be 7 or 15, E a multiple of 4096\n"); return 1; }
    std::vector<Ptrs> sets(R);
    for (int r = 0; r < R; ++r) {
        for (int i = 0; i < N; ++i) { void* p; CK(hipMalloc(&p, E)); CK(hipMemset(p, 17 * i + r, E)); sets[r].in[i] = static_cast<const uint8_t*>(p); }
        void* o; CK(hipMalloc(&o, E * 8)); CK(hipMemset(o, 0, E * 8)); sets[r].out = static_cast<double*>(o);
    }
    int cus = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); cus = prop.multiProcessorCount;
    const double bytes = static_cast<double>(E) * (N + 8);
    auto report = [&](const char* name, double us) { printf("%-44s %8.2f us  %7.1f GB/s  %.4f of 8 TB/s\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000); };
    // warm the clocks
    for (int rep = 0; rep < 3; ++rep) {
        for (int wg : {8, 12}) {
            char nm[96];
            auto inter = [&](int i) {
                Ptrs p = sets[i % R];
                if (N == 7) { p.n_units = static_cast<uint32_t>(E / 512); hipLaunchKernelGGL((k_interleaved<7, 4>), dim3(cus * wg), dim3(256), 0, 0, p); }
                else { p.n_units = static_cast<uint32_t>(E / 384); hipLaunchKernelGGL((k_interleaved<15, 3>), dim3(cus * wg), dim3(256), 0, 0, p); }
            };
            snprintf(nm, sizeof nm, "interleaved N=%d, %d wg/CU", N, wg);
            report(nm, time_us(inter, 100));
        }
        for (int wg : {2, 3, 4, 6}) {
            char nm[96];
            auto burst = [&](int i) {
                Ptrs p = sets[i % R];
                p.n_units = static_cast<uint32_t>(E / 4096);
                if (N == 7) hipLaunchKernelGGL((k_burst<7>), dim3(cus * wg), dim3(512), 0, 0, p);
                else hipLaunchKernelGGL((k_burst<15>), dim3(cus * wg), dim3(512), 0, 0, p);
            };
            snprintf(nm, sizeof nm, "wave-per-stream burst N=%d, %d wg/CU", N, wg);
            report(nm, time_us(burst, 100));
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
