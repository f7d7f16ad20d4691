// membench.hip - HBM access-pattern microbenchmark for the merge kernel's traffic shape on MI355X:
// N uint8 input streams (E bytes each) -> one float64 output stream (8E bytes). No LUT work: the
// point is to find which load/store shapes reach the HBM ceiling for this 7:8 read:write byte mix.
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int NF = 7;
struct Ptrs { const uint8_t* in[NF]; double* out; uint32_t n_units; };

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(double* p, double a, double b) {
    f64x2 v; v.x = a; v.y = b;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(p)); else *reinterpret_cast<f64x2*>(p) = v;
}

// LOADW: bytes per lane per stream load (4, 8, 16). STORE: 0 none, 1 natural (each lane writes its LOADW*8 contiguous
// bytes), 2 coalesced (same bytes per wave, lane l writes 16 B at 16*l + 1024*k: every instruction 1 KB contiguous).
// READ: false = no loads (write-only). A "unit" is one wave-iteration: 64*LOADW elements.
template <int LOADW, int STORE, bool READ, bool NT>
__global__ __launch_bounds__(256) void k_pattern(const Ptrs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * 4;
    constexpr int NW = LOADW / 4;
    constexpr uint32_t UNIT = 64 * LOADW;                   // elements per wave iteration
    for (uint32_t u = blockIdx.x * 4 + wave; u < a.n_units; u += stride) {
        const size_t base = static_cast<size_t>(u) * UNIT;
        uint32_t acc[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) acc[k] = lane + k;
        if (READ) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const uint8_t* p = a.in[i] + base + lane * LOADW;
                if constexpr (LOADW == 4) { acc[0] += ld<NT>(reinterpret_cast<const uint32_t*>(p)); }
                if constexpr (LOADW == 8) { u32x2 v = ld<NT>(reinterpret_cast<const u32x2*>(p)); acc[0] += v.x; acc[1] += v.y; }
                if constexpr (LOADW == 16) { u32x4 v = ld<NT>(reinterpret_cast<const u32x4*>(p)); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
            }
        }
        double vals[LOADW];
#pragma unroll
        for (int j = 0; j < LOADW; ++j) vals[j] = static_cast<double>((acc[j >> 2] >> (8 * (j & 3))) & 255u);
        double* o = a.out + base;
        if (STORE == 1) {
#pragma unroll
            for (int j = 0; j < LOADW; j += 2) st<NT>(o + lane * LOADW + j, vals[j], vals[j + 1]);
        } else if (STORE == 2) {
#pragma unroll
            for (int j = 0; j < LOADW; j += 2) st<NT>(o + (j / 2) * 128 + lane * 2, vals[j], vals[j + 1]);
        } else {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) x ^= acc[k];
            if (x == 0x12345678u) o[lane] = 1.0;            // never true in practice; keeps the loads alive
        }
    }
}

// plain copy, 16 B per lane in and out
__global__ __launch_bounds__(256) void k_copy16(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}

struct Result { const char* name; double us, gbps; };

template <typename F>
static double time_us(F&& launch, int iters = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const size_t E = argc > 1 ? strtoull(argv[1], nullptr, 10) : size_t(4096) * 4096 * 3;
    const int bpc = argc > 2 ? atoi(argv[2]) : 8;            // blocks per CU
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d, E = %zu elements, %d frames\n", prop.gcnArchName, prop.multiProcessorCount, E, NF);
    Ptrs a{};
    for (int i = 0; i < NF; ++i) { void* p; CK(hipMalloc(&p, E)); CK(hipMemset(p, 17 * i + 3, E)); a.in[i] = static_cast<uint8_t*>(p); }
    void* out; CK(hipMalloc(&out, E * 8)); CK(hipMemset(out, 0, E * 8)); a.out = static_cast<double*>(out);
    const int grid = prop.multiProcessorCount * bpc;
    const double rb = double(E) * NF, wb = double(E) * 8;
#define RUN(NAME, LOADW, STORE, READ, NT, BYTES) { \
        Ptrs b = a; b.n_units = static_cast<uint32_t>(E / (64 * LOADW)); \
        double us = time_us([&] { hipLaunchKernelGGL((k_pattern<LOADW, STORE, READ, NT>), dim3(grid), dim3(256), 0, 0, b); }); \
        printf("%-58s %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", NAME, us, (BYTES) / us / 1e3, (BYTES) / us / 1e3 / 8000.0); }
    printf("--- read only (7 x uint8 streams)\n");
    RUN("read  dword   plain", 4, 0, true, false, rb);
    RUN("read  dword   nt", 4, 0, true, true, rb);
    RUN("read  dwordx2 nt", 8, 0, true, true, rb);
    RUN("read  dwordx4 plain", 16, 0, true, false, rb);
    RUN("read  dwordx4 nt", 16, 0, true, true, rb);
    printf("--- write only (float64 stream)\n");
    RUN("write natural 32B/lane (2 x dwordx4, 32 B lane stride) nt", 4, 1, false, true, wb);
    RUN("write natural 32B/lane plain", 4, 1, false, false, wb);
    RUN("write coalesced (1 KB per instruction) nt", 4, 2, false, true, wb);
    RUN("write coalesced plain", 4, 2, false, false, wb);
    RUN("write natural 128B/lane (8 x dwordx4, 128 B stride) nt", 16, 1, false, true, wb);
    RUN("write coalesced, 8 instr per wave-unit nt", 16, 2, false, true, wb);
    printf("--- read + write (merge traffic shape, 15 B/element)\n");
    RUN("dword   loads + natural 32B stores   nt  [kernel today]", 4, 1, true, true, rb + wb);
    RUN("dword   loads + natural 32B stores   plain", 4, 1, true, false, rb + wb);
    RUN("dword   loads + coalesced stores     nt", 4, 2, true, true, rb + wb);
    RUN("dword   loads + coalesced stores     plain", 4, 2, true, false, rb + wb);
    RUN("dwordx2 loads + natural 64B stores   nt", 8, 1, true, true, rb + wb);
    RUN("dwordx2 loads + coalesced stores     nt", 8, 2, true, true, rb + wb);
    RUN("dwordx4 loads + natural 128B stores  nt", 16, 1, true, true, rb + wb);
    RUN("dwordx4 loads + coalesced stores     nt", 16, 2, true, true, rb + wb);
    RUN("dwordx4 loads + coalesced stores     plain", 16, 2, true, false, rb + wb);
    {
        const size_t n16 = (E * 15 / 2) / 16;               // same total traffic, half read half write
        void *ci, *co; CK(hipMalloc(&ci, n16 * 16)); CK(hipMalloc(&co, n16 * 16)); CK(hipMemset(ci, 1, n16 * 16));
        double us = time_us([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, static_cast<const u32x4*>(ci), static_cast<u32x4*>(co), n16); });
        printf("%-58s %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", "copy dwordx4 (same total bytes, 1:1 read:write)", us, 2.0 * n16 * 16 / us / 1e3, 2.0 * n16 * 16 / us / 1e3 / 8000.0);
    }
    return 0;
}
