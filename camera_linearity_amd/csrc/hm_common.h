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

// Gaussian weight of the merge, measurand.py:615: w = e ** (-30 (v - 0.5)^2) with dv = v - 0.5. One definition for every
// kernel that evaluates it analytically (float64 frames), so they agree bit for bit.
#ifdef HM_FAKE_EXP          /* timing experiment only: removes the cost of exp() */
__device__ __forceinline__ double gauss_weight(double dv) { return 1.0 + -30.0 * (dv * dv); }
#else
__device__ __forceinline__ double gauss_weight(double dv) { return exp(-30.0 * (dv * dv)); }
#endif

// scipy.ndimage 'reflect' (d c b a | a b c d) index fold, valid for any offset
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n) {
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - i - 1;
    }
    return i;
}

// ------------------------------------------------------------------------------------------------
// k x k median of one channel around one pixel, 'reflect' at the true image edges, computed by the
// WHOLE WAVE for one wave-uniform pixel: lane p < k*k loads neighbour p (one parallel load instead of
// k*k dependent ones), then every lane ranks its value against the others with k*k readlane
// broadcasts; a lane whose value v satisfies #(x < v) <= m < #(x <= v), m = k*k/2, holds the median.
// Must be called with all 64 lanes active and identical arguments.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long bits = __double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(bits), l);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<unsigned long long>(bits) >> 32), l);
    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

template <typename T>
__device__ __noinline__ T wave_median(const T* __restrict__ buf, int64_t H, int64_t W, int C,
                                      int64_t buf_row0, int64_t row, int64_t col, int c, int k) {
    const int lane = threadIdx.x & 63;
    const int n = k * k, r = k / 2, m = n / 2;
    const bool valid = lane < n;
    const int p = valid ? lane : 0;
    const int64_t yy = reflect_index(row + (p / k - r), H) - buf_row0;
    const int64_t xx = reflect_index(col + (p % k - r), W);
    const double v = static_cast<double>(buf[(yy * W + xx) * C + c]);   // uint8 -> double: exact, order-preserving
    int less = 0, leq = 0;
    for (int q = 0; q < n; ++q) {
        const double u = readlane_f64(v, q);
        less += (u < v);
        leq += (u <= v);
    }
    const unsigned long long is_med = __ballot(valid && less <= m && m < leq);
    const int src = __ffsll(static_cast<long long>(is_med)) - 1;        // non-empty for totally ordered input
    return static_cast<T>(readlane_f64(v, src < 0 ? 0 : src));
}


}  // namespace hm
