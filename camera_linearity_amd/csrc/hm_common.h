// hm_common.h - shared device/host helpers of libhdrmerge (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdrmerge.h"
#include <string>

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

// hm_merge_chunk.hip: stacks of more than HM_MAX_FRAMES frames, HM_MAX_FRAMES (or chunk_frames) per launch
int merge_chunked(const hm_merge_args* g, int chunk_frames, std::string* describe, hipStream_t st);

// grid for a bandwidth-bound grid-stride kernel: enough blocks to fill the chip, capped (guide G11)
inline unsigned stream_grid(int64_t work_items, int block, int blocks_per_cu) {
    int64_t need = (work_items + block - 1) / block;
    int64_t cap = static_cast<int64_t>(cu_count()) * blocks_per_cu;
    if (need < 1) need = 1;
    return static_cast<unsigned>(need < cap ? need : cap);
}

// grid for a kernel whose WAVES stride over `wave_items` equal work items: the cap of stream_grid(), then as few workgroups as keep the
// number of items per wave the same for every wave (12 288 items on 8 192 waves = half of the waves do two and the others wait for them)
inline unsigned balanced_wave_grid(int64_t wave_items, int waves_per_block, int blocks_per_cu) {
    const int64_t cap_waves = static_cast<int64_t>(cu_count()) * blocks_per_cu * waves_per_block;
    if (wave_items <= 0) return 1;
    const int64_t iters = (wave_items + cap_waves - 1) / cap_waves;
    const int64_t waves = (wave_items + iters - 1) / iters;
    return static_cast<unsigned>((waves + waves_per_block - 1) / waves_per_block);
}

// Gaussian weight of the merge, measurand.py:615: w = e ** (-30 (v - 0.5)^2) with dv = v - 0.5. One definition for every
// kernel that evaluates it analytically (float64 frames), so they agree bit for bit.
#ifdef HM_FAKE_EXP          /* timing experiment only: removes the cost of exp() */
__device__ __forceinline__ double gauss_weight(double dv) { return 1.0 + -30.0 * (dv * dv); }
#else
__device__ __forceinline__ double gauss_weight(double dv) { return exp(-30.0 * (dv * dv)); }
#endif

// flat-field epilogue on loaded operands, modules/measurand.py:585-602. The value keeps the reference's
// operations, (val / F) * m (:602). The three variance terms (:586-596) each divide by F**2 or F**4 in the
// reference; here 1/F**2 is formed once and multiplied (3 float64 divisions fewer per element; the std
// moves by <= 2 ulp, its test tolerance is 1e-9).
__device__ __forceinline__ void flat_field_math(double F, double iF2 /* 1 / (F*F) */, double sF, double m, double s,
                                                bool with_std, double& val, double& sd) {
    if (with_std) {
        const double v2 = val * val;
        const double u_acq = ((sd * sd) * iF2) * (m * m);
        const double u_ff = ((v2 * (iF2 * iF2)) * (sF * sF)) * (m * m);
        const double u_ffm = (v2 * iF2) * (s * s);
        sd = sqrt(u_acq + u_ff + u_ffm);
    }
    val = (val / F) * m;
}

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

// ------------------------------------------------------------------------------------------------
// k x k median of one channel around one pixel computed by ONE LANE for its own pixel (every lane of a wave may
// be at a different pixel): the shape of the queue-driven hot-pixel kernels, where every lane patches one hot
// element. k = 3: nine loads and the 19-exchange median-of-9 network (min / max pairs; checked exhaustively on
// 0/1 inputs). k = 5, 7: rank counting with the neighbourhood re-read from the cache for every candidate
// (k^4 loads worst case - rare kernel sizes; no 49-entry register array). Same order statistic as wave_median().
// ------------------------------------------------------------------------------------------------
template <typename T> struct MedianKey;
template <> struct MedianKey<uint8_t> {
    using type = uint32_t;
    static __device__ __forceinline__ uint32_t lo(uint32_t a, uint32_t b) { return a < b ? a : b; }
    static __device__ __forceinline__ uint32_t hi(uint32_t a, uint32_t b) { return a < b ? b : a; }
};
template <> struct MedianKey<double> {
    using type = double;
    static __device__ __forceinline__ double lo(double a, double b) { return a < b ? a : b; }
    static __device__ __forceinline__ double hi(double a, double b) { return a < b ? b : a; }
};

// median of nine values in registers: the 19-exchange network
template <typename T>
__device__ __forceinline__ typename MedianKey<T>::type median9(typename MedianKey<T>::type (&p)[9]) {
    using K = MedianKey<T>;
    using V = typename K::type;
#define HM_CX(a, b) { const V lo_ = K::lo(p[a], p[b]); const V hi_ = K::hi(p[a], p[b]); p[a] = lo_; p[b] = hi_; }
    HM_CX(1, 2) HM_CX(4, 5) HM_CX(7, 8) HM_CX(0, 1) HM_CX(3, 4) HM_CX(6, 7) HM_CX(1, 2) HM_CX(4, 5) HM_CX(7, 8)
    HM_CX(0, 3) HM_CX(5, 8) HM_CX(4, 7) HM_CX(3, 6) HM_CX(1, 4) HM_CX(2, 5) HM_CX(4, 7) HM_CX(4, 2) HM_CX(6, 4) HM_CX(4, 2)
#undef HM_CX
    return p[4];
}

template <typename T>
__device__ __forceinline__ T lane_median(const T* __restrict__ buf, int64_t H, int64_t W, int C,
                                         int64_t buf_row0, int64_t row, int64_t col, int c, int k) {
    using K = MedianKey<T>;
    using V = typename K::type;
    if (k == 3) {
        const int64_t y[3] = {(reflect_index(row - 1, H) - buf_row0) * W, (row - buf_row0) * W, (reflect_index(row + 1, H) - buf_row0) * W};
        const int64_t x[3] = {reflect_index(col - 1, W), col, reflect_index(col + 1, W)};
        V p[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) p[q] = static_cast<V>(buf[(y[q / 3] + x[q % 3]) * C + c]);
        return static_cast<T>(median9<T>(p));
    }
    const int n = k * k, r = k / 2, m = n / 2;
    auto at = [&](int q) -> V {
        const int64_t yy = reflect_index(row + (q / k - r), H) - buf_row0;
        const int64_t xx = reflect_index(col + (q % k - r), W);
        return static_cast<V>(buf[(yy * W + xx) * C + c]);
    };
    V med = at(m);
    for (int q = 0; q < n; ++q) {
        const V v = at(q);
        int less = 0, leq = 0;
        for (int s = 0; s < n; ++s) {
            const V u = at(s);
            less += (u < v);
            leq += (u <= v);
        }
        if (less <= m && m < leq) { med = v; break; }
    }
    return static_cast<T>(med);
}

}  // namespace hm
