// hm_common.h - shared device/host helpers of libhdrmerge (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdrmerge.h"

namespace hm {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kCUs = 256;            // MI355X: 8 XCD x 32 CU (grid sizing only; queried at runtime too)
constexpr int kMaxLds = 160 * 1024;  // bytes of LDS per CU / per workgroup on gfx950

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HM_OK : HM_ELAUNCH;
}

__host__ __device__ inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
__device__ __forceinline__ bool aligned_dev(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// number of CUs of the current device (cached)
int cu_count();

// grid for a bandwidth-bound grid-stride kernel: enough blocks to fill the chip, capped (guide G11)
inline unsigned stream_grid(int64_t work_items, int block, int blocks_per_cu) {
    int64_t need = (work_items + block - 1) / block;
    int64_t cap = static_cast<int64_t>(cu_count()) * blocks_per_cu;
    if (need < 1) need = 1;
    return static_cast<unsigned>(need < cap ? need : cap);
}

// scipy.ndimage 'reflect' (d c b a | a b c d) index fold, valid for any offset
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n) {
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - i - 1;
    }
    return i;
}

}  // namespace hm
