// gatherbench.hip - how fast are random 16-byte table gathers on MI355X through LDS vs through the
// vector L1 (global loads from an L1-resident 12 KB table)? Decides whether part of the merge kernel's
// gathers could be moved off the LDS pipe.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int ENTRY>   // MODE 0: LDS, 1: global (L1). ENTRY: bytes per gather (8 or 16)
__global__ __launch_bounds__(256) void k(const double* __restrict__ table, double* out, int iters, uint32_t mask_same) {
    __shared__ double lds[1536];
    for (int i = threadIdx.x; i < 1536; i += 256) lds[i] = table[i];
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc0 = 0, acc1 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t q = ((x >> 10) % 768u) & mask_same;            // mask_same = 0 -> all lanes same entry (broadcast)
            if (ENTRY == 16) {
                const double2 v = MODE == 0 ? *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(lds) + q * 16u)
                                            : *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(table) + q * 16u);
                acc0 += v.x; acc1 += v.y;
            } else {
                const double v = MODE == 0 ? lds[q] : table[q];
                acc0 += v;
            }
        }
    }
    if (acc0 + acc1 == 1.2345) out[0] = acc0;
}

template <int MODE, int ENTRY>
static void run(const char* name, const double* tab, double* out, uint32_t mask) {
    const int iters = 2000, grid = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, ENTRY>), dim3(grid), dim3(256), 0, 0, tab, out, iters, mask);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, ENTRY>), dim3(grid), dim3(256), 0, 0, tab, out, iters, mask);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instrs_per_cu = double(grid) * 4 * iters * 8 / 256.0;
    printf("%-44s %8.1f us   %.2f ns per wave-gather per CU  (= %.1f cycles at 2.1 GHz)\n", name, ms * 1e3, ms * 1e6 / wave_instrs_per_cu,
           ms * 1e6 / wave_instrs_per_cu * 2.1);
}

int main() {
    double* tab; double* out;
    CK(hipMalloc(&tab, 1536 * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(tab, 0, 1536 * 8));
    run<0, 16>("LDS    16-B random gathers", tab, out, 0xffffffffu);
    run<1, 16>("global 16-B random gathers (12 KB, L1)", tab, out, 0xffffffffu);
    run<0, 8>("LDS     8-B random gathers", tab, out, 0xffffffffu);
    run<1, 8>("global  8-B random gathers (6 KB, L1)", tab, out, 0xffffffffu);
    run<0, 16>("LDS    16-B broadcast (same entry)", tab, out, 0u);
    run<1, 16>("global 16-B broadcast (same entry)", tab, out, 0u);
    return 0;
}
