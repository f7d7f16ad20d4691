// membench4.hip - cache-policy sweep for the merge kernel's traffic shape (7 ushort-per-lane input streams, one 16-byte-per-lane
// float64 output stream, 128-element sub-units, 2 sub-units per wave iteration): loads and stores issued through inline asm with
// every combination of the gfx950 cache-policy bits (sc0, sc1, nt). Prints the launch time per policy pair.
//   hipcc -O3 --offload-arch=gfx950 tools/membench4.hip -o tools/bin/membench4 && tools/bin/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NF = 7, U = 2, SUB = 128;
struct Ptrs { const uint8_t* f[NF]; double* out; uint32_t n_groups; };

template <int LP> __device__ __forceinline__ uint32_t ld16(const uint8_t* base, uint32_t off) {
    uint32_t v;
    if constexpr (LP == 0) asm volatile("global_load_ushort %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 1) asm volatile("global_load_ushort %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 2) asm volatile("global_load_ushort %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 3) asm volatile("global_load_ushort %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 4) asm volatile("global_load_ushort %0, %1, %2 sc1 nt" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 5) asm volatile("global_load_ushort %0, %1, %2 sc0 sc1 nt" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 6) asm volatile("global_load_ushort %0, %1, %2 sc0" : "=v"(v) : "v"(off), "s"(base) : "memory");
    if constexpr (LP == 7) asm volatile("global_load_ushort %0, %1, %2 sc0 nt" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int SP> __device__ __forceinline__ void st128(double* base, uint32_t off, u32x4 v) {
    if constexpr (SP == 0) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 5) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 6) asm volatile("global_store_dwordx4 %0, %1, %2 sc0" :: "v"(off), "v"(v), "s"(base) : "memory");
    if constexpr (SP == 7) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
}

template <int LP, int SP>
__global__ __launch_bounds__(256) void k(const Ptrs a) {
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gstride = gridDim.x * 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < a.n_groups; g += gstride) {
        const int64_t base = static_cast<int64_t>(g) * (U * SUB);
        uint32_t r[NF][U];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int s = 0; s < U; ++s) r[i][s] = ld16<LP>(a.f[i] + base + s * SUB, lane * 2u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < U; ++s) {
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < NF; ++i) acc += r[i][s];
            u32x4 v; v.x = acc; v.y = acc >> 8; v.z = acc + 1; v.w = 0x3ff00000u;
            st128<SP>(a.out + base + s * SUB, lane * 16u, v);
        }
    }
}

template <int LP, int SP>
static float run(const Ptrs& p, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<LP, SP>), dim3(2048), dim3(256), 0, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<LP, SP>), dim3(2048), dim3(256), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    const int64_t E = 4096LL * 4096 * 3;
    Ptrs p{};
    for (int i = 0; i < NF; ++i) { void* q; hipMalloc(&q, E); hipMemset(q, i + 1, E); p.f[i] = static_cast<const uint8_t*>(q); }
    void* o; hipMalloc(&o, E * 8); p.out = static_cast<double*>(o);
    p.n_groups = static_cast<uint32_t>(E / (U * SUB));
    const char* names[8] = {"-", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0", "sc0 nt"};
    // pre-warm
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL((k<1, 1>), dim3(2048), dim3(256), 0, 0, p);
    hipDeviceSynchronize();
    float best[8][8];
    for (int round = 0; round < 3; ++round) {
#define RUN(L, S) { float t = run<L, S>(p, 100); best[L][S] = round == 0 ? t : std::min(best[L][S], t); }
#define ROW(L) RUN(L, 0) RUN(L, 1) RUN(L, 2) RUN(L, 3) RUN(L, 4) RUN(L, 5) RUN(L, 6) RUN(L, 7)
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
    }
    printf("launch us (min of 3 rounds x 100 launches); rows = load policy, columns = store policy; traffic 755 MB\n%-12s", "");
    for (int s = 0; s < 8; ++s) printf("%12s", names[s]);
    printf("\n");
    for (int l = 0; l < 8; ++l) {
        printf("%-12s", names[l]);
        for (int s = 0; s < 8; ++s) printf("%12.1f", best[l][s]);
        printf("\n");
    }
    return 0;
}
