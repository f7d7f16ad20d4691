// membench2.hip - HBM traffic-shape study for the merge (7 uint8 streams in, one float64 stream out),
// pre-warmed and measured in interleaved rounds (medians). Probes only: outputs are not meaningful.
// Knobs: LOADW (bytes/lane/load), U (sub-units per wave iteration = contiguous span per stream),
//        MAP 0: group g -> wave (g strided by total waves)           [kernel today]
//            1: each workgroup owns a contiguous run of groups (chunked), workgroups dealt round-robin
//            2: XCD-contiguous: workgroups with the same blockIdx%8 own one contiguous eighth of the image
//        NT loads/stores.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int NF = 7;
struct Ptrs { const uint8_t* in[NF]; double* out; uint32_t n_groups; };

template <int LOADW, int U, int MAP, bool NTL, bool NTS, int BLOCK, int STP = 0>
__global__ __launch_bounds__(BLOCK) void k(const Ptrs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = BLOCK / 64;
    constexpr uint32_t SUB = 64 * LOADW, GROUP = U * SUB;
    constexpr int NW = LOADW / 4;
    const uint32_t total_waves = gridDim.x * WPB;
    uint32_t g, gend, gstep;
    if (MAP == 0) { g = blockIdx.x * WPB + wave; gend = a.n_groups; gstep = total_waves; }
    else {
        uint32_t vb = blockIdx.x;                                         // virtual block index
        if (MAP == 2) { const uint32_t per = gridDim.x / 8; vb = (blockIdx.x % 8) * per + blockIdx.x / 8; }
        const uint32_t per_block = (a.n_groups + gridDim.x - 1) / gridDim.x;
        g = vb * per_block + wave; gend = min(a.n_groups, (vb + 1) * per_block); gstep = WPB;
    }
    for (; g < gend; g += gstep) {
        const size_t base = static_cast<size_t>(g) * GROUP;
        uint32_t acc[U][NW];
#pragma unroll
        for (int s = 0; s < U; ++s)
#pragma unroll
            for (int k2 = 0; k2 < NW; ++k2) acc[s][k2] = lane;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint8_t* p = a.in[i] + base + lane * LOADW;
#pragma unroll
            for (int s = 0; s < U; ++s) {
                if constexpr (LOADW == 4) { const uint32_t* q = reinterpret_cast<const uint32_t*>(p + s * SUB); acc[s][0] += NTL ? __builtin_nontemporal_load(q) : *q; }
                if constexpr (LOADW == 8) { const u32x2* q = reinterpret_cast<const u32x2*>(p + s * SUB); u32x2 v = NTL ? __builtin_nontemporal_load(q) : *q; acc[s][0] += v.x; acc[s][1] += v.y; }
                if constexpr (LOADW == 16) { const u32x4* q = reinterpret_cast<const u32x4*>(p + s * SUB); u32x4 v = NTL ? __builtin_nontemporal_load(q) : *q; acc[s][0] += v.x; acc[s][1] += v.y; acc[s][2] += v.z; acc[s][3] += v.w; }
            }
        }
#pragma unroll
        for (int s = 0; s < U; ++s) {
            double* o = a.out + base + s * SUB;
#pragma unroll
            for (int j = 0; j < LOADW; j += 2) {
                f64x2 v; v.x = static_cast<double>((acc[s][j >> 2] >> (8 * (j & 3))) & 255u); v.y = static_cast<double>((acc[s][(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 255u);
                f64x2* q;
                if (STP == 0) q = reinterpret_cast<f64x2*>(o + (j / 2) * 128 + lane * 2);                       // 1 KB contiguous per instruction
                else if (STP == 1) q = reinterpret_cast<f64x2*>(o + lane * LOADW + j);                         // natural: LOADW*8 B per lane
                else if (STP == 2) q = reinterpret_cast<f64x2*>(o + (lane >> 2) * (4 * LOADW) + (j / 2) * 8 + (lane & 3) * 2);   // 64-B chunks per quad
                else q = reinterpret_cast<f64x2*>(o + (lane >> 3) * (8 * LOADW) + (j / 2) * 16 + (lane & 7) * 2);               // 128-B chunks per 8 lanes
                if (NTS) __builtin_nontemporal_store(v, q); else *q = v;
            }
        }
    }
}

struct V { std::string name; std::function<void()> launch; std::vector<double> us; };
static std::vector<V> vs;
static Ptrs P; static size_t E;
template <int LOADW, int U, int MAP, bool NTL, bool NTS, int BLOCK, int STP = 0>
static void add(const char* name, int bpc) {
    Ptrs b = P; b.n_groups = static_cast<uint32_t>(E / (64 * LOADW * U));
    const int grid = 256 * bpc;
    vs.push_back(V{name, [=] { hipLaunchKernelGGL((k<LOADW, U, MAP, NTL, NTS, BLOCK, STP>), dim3(grid), dim3(BLOCK), 0, 0, b); }, {}});
}
int main(int argc, char** argv) {
    E = size_t(4096) * 4096 * 3;
    for (int i = 0; i < NF; ++i) { void* p; CK(hipMalloc(&p, E)); CK(hipMemset(p, 17 * i + 3, E)); P.in[i] = static_cast<uint8_t*>(p); }
    void* out; CK(hipMalloc(&out, E * 8)); CK(hipMemset(out, 0, E * 8)); P.out = static_cast<double*>(out);
    //   LOADW U MAP NTL   NTS   BLOCK          name                                      blocks/CU
    add<4, 2, 0, true, true, 256>("dword   U2 strided   nt/nt  x8   [kernel today]", 8);
    add<4, 2, 0, true, true, 256, 1>("dword   U2 stores natural 32B/lane   nt", 8);
    add<4, 2, 0, true, false, 256, 1>("dword   U2 stores natural 32B/lane   plain", 8);
    add<4, 2, 0, true, true, 256, 2>("dword   U2 stores 64-B quad chunks   nt", 8);
    add<4, 2, 0, true, false, 256, 2>("dword   U2 stores 64-B quad chunks   plain", 8);
    add<4, 2, 0, true, true, 256, 3>("dword   U2 stores 128-B 8-lane chunks nt", 8);
    add<4, 2, 0, true, false, 256, 3>("dword   U2 stores 128-B 8-lane chunks plain", 8);
    add<8, 1, 0, true, true, 256, 2>("dwordx2 U1 stores 128-B quad chunks  nt", 8);
    add<8, 1, 0, true, true, 256, 3>("dwordx2 U1 stores 256-B 8-lane chunks nt", 8);
    add<8, 1, 0, true, true, 256, 1>("dwordx2 U1 stores natural 64B/lane   nt", 8);
    add<16, 1, 0, true, true, 256, 2>("dwordx4 U1 stores 256-B quad chunks  nt", 8);
    add<16, 1, 0, true, true, 256, 1>("dwordx4 U1 stores natural 128B/lane  nt", 8);
    add<8, 1, 0, true, true, 256>("dwordx2 U1 1KB stores nt", 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 6000; ++i) vs[0].launch();
    CK(hipDeviceSynchronize());
    const int rounds = 9, iters = 40;
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            for (int i = 0; i < 5; ++i) v.launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v.launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.us.push_back(ms * 1e3 / iters);
        }
    CK(hipGetLastError());
    const double bytes = double(E) * 15;
    for (auto& v : vs) { std::sort(v.us.begin(), v.us.end()); double m = v.us[v.us.size() / 2];
        printf("%-48s med %6.1f min %6.1f max %6.1f us  %6.1f GB/s  %.3f of 8TB/s\n", v.name.c_str(), m, v.us.front(), v.us.back(), bytes / m / 1e3, bytes / m / 8e6); }
    return 0;
}
