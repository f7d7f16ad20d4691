// hm_corrections.hip - dark-frame hot-pixel filter and flat-field correction as standalone launches.
//   hm_hot_pixel_filter_*   AbstractMeasurand.filter_larger_than_by_map, modules/measurand.py:543-557
//                           (intended semantics, SURVEY.md 3.4-F: masked pixels <- k x k median, 'reflect')
//   hm_roi_mean_*           flat_field_mean, modules/measurand.py:561-579 (integer ROI)
//   hm_normalize_by_map     modules/measurand.py:585-604
#include "hm_common.h"
#include <algorithm>
#include <cstdlib>

namespace hm {

#ifndef HM_HOT_UN
#define HM_HOT_UN 2      // batches of loads in flight per lane: 1 -> 0.56, 2 -> 0.60, 6 (a wave's whole share of a 4096 x 4096 x 3 frame at once, 137 VGPRs) -> 0.49
#endif

// Streaming copy with the rare hot elements patched: every lane owns EL consecutive elements (16 bytes of x:
// 16 uint8 or 2 float64), loads them and the matching map entries with vector loads, and a wave ballot finds the
// lanes that hold hot elements; each hot element's k x k median is taken cooperatively by the whole wave
// (wave_median, hm_common.h) and patched into the owning lane's registers before the single 16-byte store.
template <typename T> struct Chunk;
template <> struct Chunk<uint8_t> { static constexpr int EL = 16; };
template <> struct Chunk<double>  { static constexpr int EL = 2; };

typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void k_hot_filter(const T* __restrict__ x, const uint8_t* __restrict__ map_u8,
                                                    const double* __restrict__ map_f64, int min_dn, double thr, int k,
                                                    T* __restrict__ out, int64_t H, int64_t W, int C, int64_t first) {
    constexpr int EL = Chunk<T>::EL;
    const int64_t n = H * W * C;
    const int lane = threadIdx.x & 63;
    const int64_t n_chunks = (n - first + EL - 1) / EL;                  // chunks of [first, n): `first` is a multiple of 16 elements
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    const bool vec_x = aligned_dev(x, 16) && aligned_dev(out, 16);
    // UN batches of 64 chunks per iteration: the 16-byte loads of x and of the map for all of them are issued before the first
    // ballot (one batch per iteration left a single pair of loads in flight per lane: 0.56 of the roofline, r01e_bench_ops.json).
    constexpr int UN = HM_HOT_UN;
    const int64_t wstride = n_waves * 64;
    for (int64_t cb0 = wave0 * 64; cb0 < n_chunks; cb0 += UN * wstride) {       // wave-uniform trip count
        cu32x4 xr[UN];                                                           // the lane's 16 bytes of x per batch, kept packed
        uint32_t hotbits[UN];
        int cnts[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t chunk = cb0 + q * wstride + lane;
            const int64_t e0 = first + chunk * EL;
            const int cnt = chunk < n_chunks ? static_cast<int>(n - e0 < EL ? n - e0 : EL) : 0;
            cnts[q] = cnt;
            hotbits[q] = 0;
            xr[q] = cu32x4{0u, 0u, 0u, 0u};
            if (cnt == EL && vec_x) {
                xr[q] = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(x + e0));
            } else if (cnt > 0) {
                T tmp[EL] = {};
                for (int j = 0; j < cnt; ++j) tmp[j] = x[e0 + j];
                __builtin_memcpy(&xr[q], tmp, 16);
            }
            if (map_u8 && cnt == EL && EL == 16 && aligned_dev(map_u8 + e0, 16)) {
                const cu32x4 mr = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(map_u8 + e0));
                const uint32_t w4[4] = {mr.x, mr.y, mr.z, mr.w};
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    hotbits[q] |= (static_cast<int>((w4[j >> 2] >> (8 * (j & 3))) & 255u) >= min_dn) ? (1u << j) : 0u;   // measurand.py:545
            } else {
                for (int j = 0; j < cnt; ++j) {
                    const bool hot = map_u8 ? (static_cast<int>(map_u8[e0 + j]) >= min_dn) : (map_f64[e0 + j] > thr);
                    hotbits[q] |= hot ? (1u << j) : 0u;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int64_t cb = cb0 + q * wstride;
            unsigned long long pending = __ballot(hotbits[q] != 0);
            while (pending) {                                                        // rare
                const int src = __ffsll(static_cast<long long>(pending)) - 1;
                pending &= pending - 1;
                uint32_t bits = __builtin_amdgcn_readlane(hotbits[q], src);
                const int64_t base = first + (cb + src) * EL;
                while (bits) {
                    const int j = __ffs(static_cast<int>(bits)) - 1;
                    bits &= bits - 1;
                    const int64_t e = base + j;
                    const int64_t wc = W * C;
                    const int64_t row = e / wc, rem = e % wc;
                    const T med = wave_median(x, H, W, C, 0, row, rem / C, static_cast<int>(rem % C), k);
                    if (lane == src) {                                               // patch element j of the packed 16 bytes
                        T tmp[EL];
                        __builtin_memcpy(tmp, &xr[q], 16);
#pragma unroll
                        for (int p = 0; p < EL; ++p) tmp[p] = (p == j) ? med : tmp[p];
                        __builtin_memcpy(&xr[q], tmp, 16);
                    }
                }
            }
            const int64_t e0 = first + (cb + lane) * EL;
            if (cnts[q] == EL && vec_x) {
                __builtin_nontemporal_store(xr[q], reinterpret_cast<cu32x4*>(out + e0));
            } else if (cnts[q] > 0) {
                T tmp[EL];
                __builtin_memcpy(tmp, &xr[q], 16);
                for (int j = 0; j < cnts[q]; ++j) out[e0 + j] = tmp[j];
            }
        }
    }
}

// k_hot_filter_burst: the same filter in the burst access shape (a wave that owns 4 KB contiguous chunks and issues its
// four 16-byte loads per lane back to back, then its four stores, copies at 6.0-6.7 TB/s against 5.3-5.9 TB/s for one load + store per
// lane and step). uint8 maps only, 16-byte aligned x / out / map, whole spans of UN * 64 chunks [span0, span0 + n_spans); no load sits
// behind a branch. Hot elements are patched by the lane that owns them (lane_median: nine loads + the median-of-9 network for k = 3), so
// a dense map costs its hot elements' neighbourhoods, not one wave-serialised median each (k_hot_filter above does the ragged rest).
constexpr int kHotBurst = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_hot_filter_burst(const T* __restrict__ x, const uint8_t* __restrict__ map_u8, int min_dn, int k,
                                                          T* __restrict__ out, int64_t H, int64_t W, int C, int64_t n_spans) {
    constexpr int EL = Chunk<T>::EL;
    constexpr int UN = kHotBurst;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    const int64_t wc = W * C;
    for (int64_t span = wave0; span < n_spans; span += n_waves) {
        const int64_t c0 = span * (UN * 64) + lane;                      // the lane's first chunk; its others follow at + 64, + 128, ...
        cu32x4 xr[UN];
        uint32_t hotbits[UN];
        if constexpr (EL == 16) {
            cu32x4 mr[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) xr[q] = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(x + (c0 + q * 64) * EL));
#pragma unroll
            for (int q = 0; q < UN; ++q) mr[q] = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(map_u8 + (c0 + q * 64) * EL));
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const uint32_t w4[4] = {mr[q].x, mr[q].y, mr[q].z, mr[q].w};
                hotbits[q] = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    hotbits[q] |= (static_cast<int>((w4[j >> 2] >> (8 * (j & 3))) & 255u) >= min_dn) ? (1u << j) : 0u;   // measurand.py:545
            }
        } else {
            uint32_t mr[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) xr[q] = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(x + (c0 + q * 64) * EL));
#pragma unroll
            for (int q = 0; q < UN; ++q) mr[q] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(map_u8 + (c0 + q * 64) * EL));
#pragma unroll
            for (int q = 0; q < UN; ++q)
                hotbits[q] = (static_cast<int>(mr[q] & 255u) >= min_dn ? 1u : 0u) | (static_cast<int>(mr[q] >> 8) >= min_dn ? 2u : 0u);
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            uint32_t bits = hotbits[q];
            if (bits) {                                                  // rare: the lane patches its own hot elements
                T tmp[EL];
                __builtin_memcpy(tmp, &xr[q], 16);
                const int64_t e0 = (c0 + q * 64) * EL;
                while (bits) {
                    const int j = __ffs(static_cast<int>(bits)) - 1;
                    bits &= bits - 1;
                    const int64_t e = e0 + j;
                    const int64_t row = e / wc, rem = e % wc;
                    const T med = lane_median(x, H, W, C, 0, row, rem / C, static_cast<int>(rem % C), k);
#pragma unroll
                    for (int p = 0; p < EL; ++p) tmp[p] = (p == j) ? med : tmp[p];
                }
                __builtin_memcpy(&xr[q], tmp, 16);
            }
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) __builtin_nontemporal_store(xr[q], reinterpret_cast<cu32x4*>(out + (c0 + q * 64) * EL));
    }
}

// ---- ROI mean: stage 1 = per-block partial sums per channel (wave shuffles + LDS), stage 2 = one block ----
constexpr int kRoiBlocks = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_roi_partial(const T* __restrict__ img, int64_t W, int C,
                                                     int64_t x0, int64_t x1, int64_t y0, int64_t y1,
                                                     double* __restrict__ partial /*[gridDim.x][HM_MAX_CHANNELS]*/) {
    __shared__ double red[4][HM_MAX_CHANNELS];
    const int64_t roi_w = (y1 - y0) * C;                 // contiguous elements per ROI row
    const int64_t total = (x1 - x0) * roi_w;
    double acc[HM_MAX_CHANNELS] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t r = q / roi_w, o = q % roi_w;
        const double v = static_cast<double>(img[(x0 + r) * W * C + y0 * C + o]);
        const int c = static_cast<int>(o % C);
#pragma unroll
        for (int k = 0; k < HM_MAX_CHANNELS; ++k) acc[k] += (k == c) ? v : 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < HM_MAX_CHANNELS; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < HM_MAX_CHANNELS)
        partial[blockIdx.x * HM_MAX_CHANNELS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_roi_final(const double* __restrict__ partial, int nblocks, int C,
                                                   double count, double div, double* __restrict__ out) {
    __shared__ double red[4][HM_MAX_CHANNELS];
    double acc[HM_MAX_CHANNELS] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
        for (int k = 0; k < HM_MAX_CHANNELS; ++k) acc[k] += partial[b * HM_MAX_CHANNELS + k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < HM_MAX_CHANNELS; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < C)
        out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / div) / count;
}

// ---- normalize_by_map ----
struct FFMeans { double m[HM_MAX_CHANNELS]; double s[HM_MAX_CHANNELS]; };

// one element of normalize_by_map, operation for operation (measurand.py:585-602)
__device__ __forceinline__ void normalize_one(double v, double s0, double F, double sF, double m, double s, bool with_std,
                                              double& ov, double& os) {
    if (with_std) {
        const double F2 = F * F;
        double u_acq = (s0 * s0) / F2;        // measurand.py:586-587
        u_acq *= m * m;
        double u_ff = (v * v) / (F2 * F2);    // :590-592
        u_ff *= sF * sF;
        u_ff *= m * m;
        double u_ffm = (v * v) / F2;          // :595-596
        u_ffm *= s * s;
        os = sqrt(u_acq + u_ff + u_ffm);      // :599
    }
    ov = (v / F) * m;                         // :602
}

__device__ __forceinline__ double pick(const double (&a)[HM_MAX_CHANNELS], uint32_t c) {      // no per-lane kernarg indexing
    double r = a[0];
#pragma unroll
    for (int k = 1; k < HM_MAX_CHANNELS; ++k) r = (c == static_cast<uint32_t>(k)) ? a[k] : r;
    return r;
}

// two elements per lane: 16-byte accesses on every float64 stream (1 KB contiguous per wave instruction), one ushort for a
// uint8 flat; the channel of the lane's first element is a running counter along the grid-stride loop.
__global__ __launch_bounds__(256) void k_normalize(const double* __restrict__ val, const double* __restrict__ sd,
                                                   const uint8_t* __restrict__ flat_u8, const double* __restrict__ flat_f64,
                                                   const double* __restrict__ flat_std, const FFMeans ff,
                                                   double* __restrict__ out_val, double* __restrict__ out_std,
                                                   int64_t n, int C) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t units = n / 2;
    const bool with_std = out_std != nullptr;
    const bool vec_ok = aligned_dev(val, 16) && aligned_dev(out_val, 16) && (!flat_u8 || aligned_dev(flat_u8, 2)) &&
                        (!flat_f64 || aligned_dev(flat_f64, 16)) &&
                        (!with_std || (aligned_dev(sd, 16) && aligned_dev(flat_std, 16) && aligned_dev(out_std, 16)));
    const uint32_t uC = static_cast<uint32_t>(C);
    const int64_t u0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t c0 = static_cast<uint32_t>((2 * u0) % C);
    const uint32_t cstep = static_cast<uint32_t>((2 * stride) % C);
    for (int64_t u = u0; u < units; u += stride) {
        const int64_t e = 2 * u;
        const uint32_t c1 = c0 + 1u == uC ? 0u : c0 + 1u;
        double v[2], s0[2] = {0.0, 0.0}, F[2], sF[2] = {0.0, 0.0};
        if (vec_ok) {
            const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(val + e));
            v[0] = a.x; v[1] = a.y;
            if (flat_u8) {
                const uint32_t r = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(flat_u8 + e));
                F[0] = static_cast<double>(r & 255u) / 255.0; F[1] = static_cast<double>(r >> 8) / 255.0;
            } else {
                const f64x2 f = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(flat_f64 + e));
                F[0] = f.x; F[1] = f.y;
            }
            if (with_std) {
                const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sd + e));
                const f64x2 g = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(flat_std + e));
                s0[0] = b.x; s0[1] = b.y; sF[0] = g.x; sF[1] = g.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                v[j] = val[e + j];
                F[j] = flat_u8 ? static_cast<double>(flat_u8[e + j]) / 255.0 : flat_f64[e + j];
                if (with_std) { s0[j] = sd[e + j]; sF[j] = flat_std[e + j]; }
            }
        }
        double ov[2], os[2] = {0.0, 0.0};
        normalize_one(v[0], s0[0], F[0], sF[0], pick(ff.m, c0), pick(ff.s, c0), with_std, ov[0], os[0]);
        normalize_one(v[1], s0[1], F[1], sF[1], pick(ff.m, c1), pick(ff.s, c1), with_std, ov[1], os[1]);
        if (vec_ok) {
            f64x2 o; o.x = ov[0]; o.y = ov[1];
            __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(out_val + e));
            if (with_std) { f64x2 q; q.x = os[0]; q.y = os[1]; __builtin_nontemporal_store(q, reinterpret_cast<f64x2*>(out_std + e)); }
        } else {
            out_val[e] = ov[0]; out_val[e + 1] = ov[1];
            if (with_std) { out_std[e] = os[0]; out_std[e + 1] = os[1]; }
        }
        c0 += cstep;
        if (c0 >= uC) c0 -= uC;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        const int64_t e = n - 1;
        const uint32_t c = static_cast<uint32_t>(e % C);
        const double F = flat_u8 ? static_cast<double>(flat_u8[e]) / 255.0 : flat_f64[e];
        double ov, os = 0.0;
        normalize_one(val[e], with_std ? sd[e] : 0.0, F, with_std ? flat_std[e] : 0.0, pick(ff.m, c), pick(ff.s, c), with_std, ov, os);
        out_val[e] = ov;
        if (with_std) out_std[e] = os;
    }
}

template <typename T>
static int hot_filter_common(const T* x, const uint8_t* map_u8, const double* map_f64, int min_dn, double thr, int k,
                             T* out, int64_t H, int64_t W, int C, void* stream) {
    if (H < 0 || W < 0 || C < 1) return HM_EINVAL;
    const int64_t n = H * W * C;
    if (n == 0) return HM_OK;
    if (!x || !out || (!map_u8 && !map_f64) || x == out) return HM_EINVAL;
    if (k < 3 || k > 7 || (k % 2) == 0) return HM_EINVAL;
    if (sizeof(T) == 8 && (!aligned(x, 8) || !aligned(out, 8))) return HM_EALIGN;
    // whole 4 KB-per-wave spans in the burst shape (uint8 map, 16-byte aligned buffers), the ragged rest - and every other case - chunk by chunk
    constexpr int EL = Chunk<T>::EL;
    const int64_t span_elems = static_cast<int64_t>(EL) * 64 * kHotBurst;
    int64_t n_spans = 0;
    if (map_u8 && aligned(x, 16) && aligned(out, 16) && aligned(map_u8, 16)) n_spans = n / span_elems;
    // 12 workgroups per CU and every wave the same number of spans: 25.3-25.6 us for a 4096 x 4096 x 3 uint8 frame (0.74-0.75 of 8 TB/s, the box's
    // copy rate) against 27.3 us with 8 (12 288 spans on 8 192 waves) - profiles/r03_hot_filter_grid.log
    if (n_spans > 0)
        hipLaunchKernelGGL(k_hot_filter_burst<T>, dim3(balanced_wave_grid(n_spans, 4, 12)), dim3(256), 0, as_stream(stream), x, map_u8, min_dn, k, out, H, W, C, n_spans);
    const int64_t done = n_spans * span_elems;
    if (done < n) {
        const int64_t rest = n - done;
        hipLaunchKernelGGL(k_hot_filter<T>, dim3(stream_grid((rest + EL - 1) / EL, 256, 8)), dim3(256), 0, as_stream(stream),
                           x, map_u8, map_f64, min_dn, thr, k, out, H, W, C, done);
    }
    return launch_status();
}

template <typename T>
static int roi_mean_common(const T* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0, int64_t y1,
                           double div, double* out_mean, void* workspace, void* stream) {
    if (!img || !out_mean || !workspace || C < 1 || C > HM_MAX_CHANNELS) return HM_EINVAL;
    if (x0 < 0 || y0 < 0 || x1 > H || y1 > W || x1 <= x0 || y1 <= y0) return HM_ESHAPE;
    const int64_t total = (x1 - x0) * (y1 - y0) * C;
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(kRoiBlocks, (total + 255) / 256));
    double* partial = static_cast<double*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_roi_partial<T>, dim3(grid), dim3(256), 0, st, img, W, C, x0, x1, y0, y1, partial);
    hipLaunchKernelGGL(k_roi_final, dim3(1), dim3(256), 0, st, partial, static_cast<int>(grid), C,
                       static_cast<double>((x1 - x0) * (y1 - y0)), div, out_mean);
    return launch_status();
}

}  // namespace hm

using namespace hm;

extern "C" int hm_hot_pixel_filter_u8(const uint8_t* x, const uint8_t* map_u8, const double* map_f64, int min_dn,
                                      double thr, int median_k, uint8_t* out, int64_t H, int64_t W, int C, void* stream) {
    return hot_filter_common<uint8_t>(x, map_u8, map_f64, min_dn, thr, median_k, out, H, W, C, stream);
}
extern "C" int hm_hot_pixel_filter_f64(const double* x, const uint8_t* map_u8, const double* map_f64, int min_dn,
                                       double thr, int median_k, double* out, int64_t H, int64_t W, int C, void* stream) {
    return hot_filter_common<double>(x, map_u8, map_f64, min_dn, thr, median_k, out, H, W, C, stream);
}

extern "C" size_t hm_roi_mean_workspace_bytes(void) { return sizeof(double) * kRoiBlocks * HM_MAX_CHANNELS; }

extern "C" int hm_roi_mean_u8(const uint8_t* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0,
                              int64_t y1, double* out_mean, void* workspace, void* stream) {
    // mean of DN/255 values: sum the integer DNs exactly, scale once
    return roi_mean_common<uint8_t>(img, H, W, C, x0, x1, y0, y1, 255.0, out_mean, workspace, stream);
}
extern "C" int hm_roi_mean_f64(const double* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0,
                               int64_t y1, double* out_mean, void* workspace, void* stream) {
    if (img && !aligned(img, 8)) return HM_EALIGN;
    return roi_mean_common<double>(img, H, W, C, x0, x1, y0, y1, 1.0, out_mean, workspace, stream);
}

extern "C" int hm_normalize_by_map(const double* val, const double* std, const uint8_t* flat_u8, const double* flat_f64,
                                   const double* flat_std, const double* ff_mean, const double* ff_std_mean,
                                   double* out_val, double* out_std, int64_t n, int C, void* stream) {
    if (n < 0 || C < 1 || C > HM_MAX_CHANNELS) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!val || !out_val || !ff_mean || (!flat_u8 == !flat_f64)) return HM_EINVAL;
    if (out_std && (!std || !flat_std || !ff_std_mean)) return HM_EINVAL;
    FFMeans ff{};
    for (int c = 0; c < C; ++c) { ff.m[c] = ff_mean[c]; ff.s[c] = ff_std_mean ? ff_std_mean[c] : 0.0; }
    hipLaunchKernelGGL(k_normalize, dim3(stream_grid((n + 1) / 2, 256, 8)), dim3(256), 0, as_stream(stream),
                       val, std, flat_u8, flat_f64, flat_std, ff, out_val, out_std, n, C);
    return launch_status();
}
