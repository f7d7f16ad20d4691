// hm_linearize.hip - the per-frame pieces of the hot path as standalone launches (gfx950):
//   hm_u8_to_unit_f64       modules/image_set.py:223        value = DN / 255
//   hm_gaussian_weight_*    modules/measurand.py:606-618    w, dw
//   hm_linearize_*          modules/measurand.py:471-541    ICRF LUT gather (+ ICRF_diff * std)
// All are streaming kernels: 2 elements per lane for uint8 input (ushort load, one 16-byte store that makes
// every store instruction a contiguous 1 KB), LUTs staged in LDS, grid-stride loops.
#include "hm_common.h"

namespace hm {

typedef double f64x2 __attribute__((ext_vector_type(2)));


// ---------------- u8 -> DN/255 ----------------
// 2 elements per lane: one ushort load (128 contiguous bytes per wave instruction) and one 16-byte store
// (1 KB contiguous per wave instruction) - store instructions must cover whole 128-byte lines.
__global__ __launch_bounds__(256) void k_u8_to_unit(const uint8_t* __restrict__ dn, double* __restrict__ out, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t units = n / 2;
    const bool vec_ok = aligned_dev(dn, 2) && aligned_dev(out, 16);
    for (int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u < units; u += stride) {
        const uint32_t r = vec_ok ? static_cast<uint32_t>(__builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(dn + 2 * u)))
                                  : (dn[2 * u] | (static_cast<uint32_t>(dn[2 * u + 1]) << 8));
        const double a = static_cast<double>(r & 255u) / 255.0, b = static_cast<double>(r >> 8) / 255.0;
        if (vec_ok) { f64x2 v; v.x = a; v.y = b; __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(out + 2 * u)); }
        else { out[2 * u] = a; out[2 * u + 1] = b; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) out[n - 1] = static_cast<double>(dn[n - 1]) / 255.0;
}

// ---------------- gaussian weight ----------------
__global__ __launch_bounds__(256) void k_weight_f64(const double* __restrict__ v, double* __restrict__ w,
                                                    double* __restrict__ dw, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const double d = v[e] - 0.5;
        const double y = gauss_weight(d);          // np.e ** (-30 (v - 0.5)^2), measurand.py:615
        if (w) w[e] = y;
        if (dw) dw[e] = (-60.0 * d) * y;                // -2 * 30 * (v - 0.5) * y, measurand.py:616
    }
}

// the same in the burst access shape: a wave owns 512-element chunks, issues its four 16-byte loads per lane back to back,
// evaluates its 8 elements and writes w and dw stream by stream. profiles/r03i_weightf64_rocprof_summary.md is the kernel above on a
// 4096 x 4096 x 3 frame: 16 VGPRs, one 8-byte load in flight per lane, VALU active 2.7 % of the wave cycles, 0.59 of 8 TB/s - bound by its
// access shape, not by exp().
constexpr int kWBurst = 4;
constexpr int kWChunk = 128 * kWBurst;
template <bool HAS_W, bool HAS_DW>
__global__ __launch_bounds__(256) void k_weight_f64_burst(const double* __restrict__ v, double* __restrict__ w, double* __restrict__ dw, int64_t n_chunks) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cstride = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); c < n_chunks; c += cstride) {
        const int64_t b = c * kWChunk;
        f64x2 x[kWBurst], y[kWBurst], dy[kWBurst];
#pragma unroll
        for (int k = 0; k < kWBurst; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(v + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < kWBurst; ++k) {
            const double d0 = x[k].x - 0.5, d1 = x[k].y - 0.5;
            const double y0 = gauss_weight(d0), y1 = gauss_weight(d1);      // measurand.py:615
            y[k] = f64x2{y0, y1};
            dy[k] = f64x2{(-60.0 * d0) * y0, (-60.0 * d1) * y1};           // measurand.py:616
        }
        if constexpr (HAS_W) {
#pragma unroll
            for (int k = 0; k < kWBurst; ++k) __builtin_nontemporal_store(y[k], reinterpret_cast<f64x2*>(w + b + 128 * k) + lane);
        }
        if constexpr (HAS_DW) {
#pragma unroll
            for (int k = 0; k < kWBurst; ++k) __builtin_nontemporal_store(dy[k], reinterpret_cast<f64x2*>(dw + b + 128 * k) + lane);
        }
    }
}

__global__ __launch_bounds__(256) void k_weight_u8(const uint8_t* __restrict__ dn, const double* __restrict__ w_lut,
                                                   const double* __restrict__ dw_lut, double* __restrict__ w,
                                                   double* __restrict__ dw, int64_t n) {
    __shared__ double2 t[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) t[i] = double2{w_lut[i], dw_lut ? dw_lut[i] : 0.0};
    __syncthreads();
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t units = n / 2;
    const bool vec_ok = aligned_dev(dn, 2) && (!w || aligned_dev(w, 16)) && (!dw || aligned_dev(dw, 16));
    for (int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u < units; u += stride) {
        const uint32_t r = vec_ok ? static_cast<uint32_t>(__builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(dn + 2 * u)))
                                  : (dn[2 * u] | (static_cast<uint32_t>(dn[2 * u + 1]) << 8));
        const double2 a = t[r & 255u], b = t[r >> 8];
        if (vec_ok) {
            if (w) { f64x2 v; v.x = a.x; v.y = b.x; __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(w + 2 * u)); }
            if (dw) { f64x2 v; v.x = a.y; v.y = b.y; __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(dw + 2 * u)); }
        } else {
            if (w) { w[2 * u] = a.x; w[2 * u + 1] = b.x; }
            if (dw) { dw[2 * u] = a.y; dw[2 * u + 1] = b.y; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        const double2 a = t[dn[n - 1]];
        if (w) w[n - 1] = a.x;
        if (dw) dw[n - 1] = a.y;
    }
}

// ---------------- linearize ----------------
// LUT in LDS: {g, d} 16-byte entries with std, plain 8-byte g entries without (half the LDS bytes per gather);
// index idx * lut_stride + (e % C) * (lut_stride > 1). The channel of a thread's element is carried along the
// grid-stride loop as a running counter (one add and one conditional subtract per step instead of a 64-bit modulo).
template <bool STD> struct LinEntry { typedef double type; };
template <> struct LinEntry<true> { typedef double2 type; };
template <bool STD> __device__ __forceinline__ typename LinEntry<STD>::type lin_entry(double g, double d);
template <> __device__ __forceinline__ double lin_entry<false>(double g, double) { return g; }
template <> __device__ __forceinline__ double2 lin_entry<true>(double g, double d) { return double2{g, d}; }
__device__ __forceinline__ double lin_g(double e) { return e; }
__device__ __forceinline__ double lin_g(double2 e) { return e.x; }
__device__ __forceinline__ double lin_d(double) { return 0.0; }
__device__ __forceinline__ double lin_d(double2 e) { return e.y; }

template <bool F64IN, bool STD>
__global__ __launch_bounds__(256) void k_linearize(const void* __restrict__ in, const double* __restrict__ sd,
                                                   const double* __restrict__ icrf, const double* __restrict__ icrf_diff,
                                                   double* __restrict__ out_val, double* __restrict__ out_std,
                                                   uint8_t* __restrict__ out_idx, int64_t n, int C, int lut_stride) {
    typedef typename LinEntry<STD>::type Entry;
    __shared__ Entry t[256 * HM_MAX_CHANNELS];
    const int entries = 256 * lut_stride;
    for (int i = threadIdx.x; i < entries; i += blockDim.x) t[i] = lin_entry<STD>(icrf[i], STD ? icrf_diff[i] : 0.0);
    __syncthreads();
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const bool per_ch = lut_stride > 1;
    const uint32_t uC = static_cast<uint32_t>(C), ls = static_cast<uint32_t>(lut_stride);
    if (!F64IN) {
        const uint8_t* dn = static_cast<const uint8_t*>(in);
        const int64_t units = n / 2;
        const bool vec_ok = aligned_dev(dn, 2) && aligned_dev(out_val, 16) && (!STD || (aligned_dev(sd, 16) && aligned_dev(out_std, 16)));
        const int64_t u0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
        uint32_t c0 = per_ch ? static_cast<uint32_t>((2 * u0) % C) : 0u;               // channel of element 2u
        const uint32_t cstep = per_ch ? static_cast<uint32_t>((2 * stride) % C) : 0u;
        for (int64_t u = u0; u < units; u += stride) {
            const int64_t e0 = 2 * u;
            const uint32_t r = vec_ok ? static_cast<uint32_t>(__builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(dn + e0)))
                                      : (dn[e0] | (static_cast<uint32_t>(dn[e0 + 1]) << 8));
            const uint32_t c1 = per_ch ? (c0 + 1u == uC ? 0u : c0 + 1u) : 0u;
            const Entry g0 = t[(r & 255u) * ls + c0], g1 = t[(r >> 8) * ls + c1];
            double s0 = 0.0, s1 = 0.0;
            if (STD) {
                if (vec_ok) { const f64x2 sv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sd + e0)); s0 = sv.x; s1 = sv.y; }
                else { s0 = sd[e0]; s1 = sd[e0 + 1]; }
                s0 = lin_d(g0) * s0; s1 = lin_d(g1) * s1;                                // measurand.py:512
            }
            if (vec_ok) {
                f64x2 v; v.x = lin_g(g0); v.y = lin_g(g1);
                __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(out_val + e0));
                if (STD) { f64x2 q; q.x = s0; q.y = s1; __builtin_nontemporal_store(q, reinterpret_cast<f64x2*>(out_std + e0)); }
            } else {
                out_val[e0] = lin_g(g0); out_val[e0 + 1] = lin_g(g1);
                if (STD) { out_std[e0] = s0; out_std[e0 + 1] = s1; }
            }
            c0 += cstep;
            if (c0 >= uC) c0 -= uC;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
            const int64_t e = n - 1;
            const Entry gd = t[dn[e] * ls + (per_ch ? static_cast<uint32_t>(e % C) : 0u)];
            out_val[e] = lin_g(gd);
            if (STD) out_std[e] = lin_d(gd) * sd[e];
        }
    } else {
        const double* v = static_cast<const double*>(in);
        const int64_t e_0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
        uint32_t c = per_ch ? static_cast<uint32_t>(e_0 % C) : 0u;
        const uint32_t cstep = per_ch ? static_cast<uint32_t>(stride % C) : 0u;
        for (int64_t e = e_0; e < n; e += stride) {
            // around(val * MAX_DN).astype(uint8): round half to even, then wrap (measurand.py:503)
            const uint32_t k = static_cast<uint32_t>(static_cast<int64_t>(rint(v[e] * 255.0))) & 255u;
            const Entry gd = t[k * ls + c];
            out_val[e] = lin_g(gd);
            if (STD) out_std[e] = lin_d(gd) * sd[e];
            if (out_idx) out_idx[e] = static_cast<uint8_t>(k);
            c += cstep;
            if (c >= uC) c -= uC;
        }
    }
}


// ---------------- linearize, streaming form ----------------
// The shape of the merge kernels (hm_merge.hip, merge_u8_val3) for one frame: a wave owns groups of 3 x 128 = 384 consecutive
// elements (128 whole pixels for C = 3, so the channel of element j of sub-unit s is a compile-time function of (s, j) on top of a
// lane constant), lane l handles elements 2l, 2l + 1 of a sub-unit - one ushort (uint8 input) or one 16-byte (float64 input) load
// per sub-unit and 16-byte stores that make every store instruction a contiguous 1 KB; the NEXT group's loads are issued before
// the current group's gathers. The general kernel below issues one load per lane and loop iteration and waits for it: a wave then
// spends more of its life in HBM round trips when there are several input streams (float64 values, float64 std). Takes C == 3 with a
// per-channel table, or any C with a single table (lut_stride == 1), on aligned buffers; hm_linearize_* sends the body here where this
// kernel is the faster one (measured: see linearize_common) and the tail (n mod 384) and everything else to k_linearize.
constexpr uint32_t kLinSub = 128, kLinU = 3, kLinGroup = kLinSub * kLinU;

template <bool F64IN, bool STD, bool PERCH>
__global__ __launch_bounds__(256) void k_linearize_stream(const void* __restrict__ in, const double* __restrict__ sd,
                                                          const double* __restrict__ icrf, const double* __restrict__ icrf_diff,
                                                          double* __restrict__ out_val, double* __restrict__ out_std,
                                                          uint8_t* __restrict__ out_idx, uint32_t n_groups) {
    typedef typename LinEntry<STD>::type Entry;
    constexpr int ENT = sizeof(Entry);
    __shared__ __attribute__((aligned(16))) Entry t[256 * (PERCH ? 3 : 1)];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gstride = gridDim.x * 4u;
    uint32_t g = blockIdx.x * 4u + wave;                                            // wave-uniform
    const uint8_t* dn = static_cast<const uint8_t*>(in);
    const double* vin = static_cast<const double*>(in);

    uint32_t rA[kLinU], rB[kLinU];                                                  // uint8 input: two DNs per register
    f64x2 xA[F64IN ? kLinU : 1], xB[F64IN ? kLinU : 1];                             // float64 input
    f64x2 sA[STD ? kLinU : 1], sB[STD ? kLinU : 1];
    auto load = [&](uint32_t grp, uint32_t (&r)[kLinU], f64x2 (&x)[F64IN ? kLinU : 1], f64x2 (&sv)[STD ? kLinU : 1]) {
        const int64_t base = static_cast<int64_t>(grp) * kLinGroup;                 // scalar
#pragma unroll
        for (int s_ = 0; s_ < static_cast<int>(kLinU); ++s_) {
            if constexpr (F64IN) x[s_] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(vin + base + kLinSub * s_) + lane);
            else r[s_] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(dn + base + kLinSub * s_) + lane);
            if constexpr (STD) sv[s_] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sd + base + kLinSub * s_) + lane);
        }
    };
    if (g < n_groups) load(g, rA, xA, sA);
    for (int i = threadIdx.x; i < 256 * (PERCH ? 3 : 1); i += 256) t[i] = lin_entry<STD>(icrf[i], STD ? icrf_diff[i] : 0.0);
    __syncthreads();
    if (g >= n_groups) return;
    const uint32_t k = PERCH ? (lane * 2u) % 3u : 0u;                               // channel of the lane's first element in sub-unit 0
    const uint32_t off[3] = {k * ENT, ((k + 1u) % 3u) * ENT, ((k + 2u) % 3u) * ENT};

    auto process = [&](uint32_t grp, const uint32_t (&r)[kLinU], const f64x2 (&x)[F64IN ? kLinU : 1], const f64x2 (&sv)[STD ? kLinU : 1]) {
        const int64_t base = static_cast<int64_t>(grp) * kLinGroup;
#pragma unroll
        for (int s_ = 0; s_ < static_cast<int>(kLinU); ++s_) {
            uint32_t k0, k1;
            if constexpr (F64IN) {                                                   // around(val * MAX_DN).astype(uint8): half to even, wrap (measurand.py:503)
                k0 = static_cast<uint32_t>(static_cast<int64_t>(rint(x[s_].x * 255.0))) & 255u;
                k1 = static_cast<uint32_t>(static_cast<int64_t>(rint(x[s_].y * 255.0))) & 255u;
            } else { k0 = r[s_] & 255u; k1 = r[s_] >> 8; }
            const char* tb = reinterpret_cast<const char*>(t);
            const uint32_t a0 = PERCH ? __umul24(k0, 3u * ENT) + off[(2 * s_) % 3] : k0 * ENT;
            const uint32_t a1 = PERCH ? __umul24(k1, 3u * ENT) + off[(2 * s_ + 1) % 3] : k1 * ENT;
            const Entry g0 = *reinterpret_cast<const Entry*>(tb + a0), g1 = *reinterpret_cast<const Entry*>(tb + a1);
            f64x2 o; o.x = lin_g(g0); o.y = lin_g(g1);
            __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(out_val + base + kLinSub * s_) + lane);
            if constexpr (STD) {
                f64x2 w; w.x = lin_d(g0) * sv[s_].x; w.y = lin_d(g1) * sv[s_].y;     // measurand.py:512
                __builtin_nontemporal_store(w, reinterpret_cast<f64x2*>(out_std + base + kLinSub * s_) + lane);
            }
            if constexpr (F64IN) {
                if (out_idx) reinterpret_cast<uint16_t*>(out_idx + base + kLinSub * s_)[lane] = static_cast<uint16_t>(k0 | (k1 << 8));
            }
        }
    };
    while (true) {                                                                   // A holds group g
        if (g + gstride >= n_groups) { process(g, rA, xA, sA); break; }
        load(g + gstride, rB, xB, sB);
        process(g, rA, xA, sA);
        g += gstride;                                                                // B holds group g
        if (g + gstride >= n_groups) { process(g, rB, xB, sB); break; }
        load(g + gstride, rA, xA, sA);
        process(g, rB, xB, sB);
        g += gstride;
    }
}

}  // namespace hm

using namespace hm;

extern "C" int hm_u8_to_unit_f64(const uint8_t* dn, double* out, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!dn || !out))) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!aligned(out, 8)) return HM_EALIGN;
    hipLaunchKernelGGL(k_u8_to_unit, dim3(stream_grid((n + 1) / 2, 256, 8)), dim3(256), 0, as_stream(stream), dn, out, n);
    return launch_status();
}

extern "C" int hm_gaussian_weight_f64(const double* v, double* w, double* dw, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!v || (!w && !dw)))) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!aligned(v, 8) || (w && !aligned(w, 8)) || (dw && !aligned(dw, 8))) return HM_EALIGN;
    // whole 512-element chunks of 16-byte aligned buffers in the burst shape, the rest element by element
    const int64_t n_chunks = (aligned(v, 16) && (!w || aligned(w, 16)) && (!dw || aligned(dw, 16))) ? n / kWChunk : 0;
    if (n_chunks > 0) {
        const unsigned bgrid = balanced_wave_grid(n_chunks, 4, 12);
        if (w && dw) hipLaunchKernelGGL((k_weight_f64_burst<true, true>), dim3(bgrid), dim3(256), 0, as_stream(stream), v, w, dw, n_chunks);
        else if (w) hipLaunchKernelGGL((k_weight_f64_burst<true, false>), dim3(bgrid), dim3(256), 0, as_stream(stream), v, w, dw, n_chunks);
        else hipLaunchKernelGGL((k_weight_f64_burst<false, true>), dim3(bgrid), dim3(256), 0, as_stream(stream), v, w, dw, n_chunks);
        const int64_t done = n_chunks * kWChunk;
        if (done == n) return launch_status();
        v += done; n -= done;
        if (w) w += done;
        if (dw) dw += done;
    }
    hipLaunchKernelGGL(k_weight_f64, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream), v, w, dw, n);
    return launch_status();
}

extern "C" int hm_gaussian_weight_u8(const uint8_t* dn, const double* w_lut, const double* dw_lut,
                                     double* w, double* dw, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!dn || !w_lut || (!w && !dw) || (dw && !dw_lut)))) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if ((w && !aligned(w, 8)) || (dw && !aligned(dw, 8))) return HM_EALIGN;
    hipLaunchKernelGGL(k_weight_u8, dim3(stream_grid((n + 1) / 2, 256, 8)), dim3(256), 0, as_stream(stream), dn, w_lut, dw_lut, w, dw, n);
    return launch_status();
}

static int linearize_common(bool f64in, const void* in, const double* sd, const double* icrf, const double* icrf_diff,
                            double* out_val, double* out_std, uint8_t* out_idx, int64_t n, int C, int lut_stride, void* stream) {
    if (n < 0 || C < 1 || !icrf || (n > 0 && (!in || !out_val))) return HM_EINVAL;
    if (lut_stride != 1 && lut_stride != C) return HM_EINVAL;
    if (C > HM_MAX_CHANNELS && lut_stride != 1) return HM_EUNSUPPORTED;
    if (n == 0) return HM_OK;
    if (!aligned(out_val, 8) || (out_std && !aligned(out_std, 8)) || (sd && !aligned(sd, 8)) || (f64in && !aligned(in, 8)))
        return HM_EALIGN;
    const bool with_std = sd && icrf_diff && out_std;                       // measurand.py:498-500
    hipStream_t st = as_stream(stream);
    // streaming kernel for the body (whole groups of 384 elements) when the shape and the alignment allow it
    int64_t body = 0;
    const bool perch = lut_stride > 1;
    const bool shape_ok = (perch && C == 3) || !perch;
    const bool align_ok = aligned(out_val, 16) && (!with_std || (aligned(sd, 16) && aligned(out_std, 16))) &&
                          (f64in ? aligned(in, 16) : aligned(in, 2)) && (!out_idx || aligned(out_idx, 2));
    // ... and when it is the faster of the two (tools/lin_ab.py, one box, tables resident): float64 input 131 vs 144 us, + std 272 vs 307 us,
    // uint8 + std 220 vs 239 us - but uint8 WITHOUT std (9 B/element, 8 of them written) 95 vs 69 us: that case stays on the general kernel,
    // whose plain thread-strided loop (1 KB per wave and step over a window of 8 MB) reaches 0.81 of the roofline, like hm_u8_to_unit_f64.
    const bool stream_wins = f64in || with_std;
    if (stream_wins && shape_ok && align_ok && n / kLinGroup > 0 && n / kLinGroup < (int64_t{1} << 31)) {
        const uint32_t groups = static_cast<uint32_t>(n / kLinGroup);
        body = static_cast<int64_t>(groups) * kLinGroup;
        const unsigned sgrid = stream_grid(groups, 4, 8);
#define HM_LINS(F, S, P) hipLaunchKernelGGL((k_linearize_stream<F, S, P>), dim3(sgrid), dim3(256), 0, st, in, sd, icrf, icrf_diff, out_val, out_std, out_idx, groups)
        if (f64in) {
            if (with_std) { if (perch) HM_LINS(true, true, true); else HM_LINS(true, true, false); }
            else          { if (perch) HM_LINS(true, false, true); else HM_LINS(true, false, false); }
        } else {
            if (with_std) { if (perch) HM_LINS(false, true, true); else HM_LINS(false, true, false); }
            else          { if (perch) HM_LINS(false, false, true); else HM_LINS(false, false, false); }
        }
#undef HM_LINS
        if (body == n) return launch_status();
        // tail: fewer than 384 elements, through the general kernel (body is a multiple of 3, so element % C is unchanged for C = 3;
        // a single table does not look at the channel)
        in = f64in ? static_cast<const void*>(static_cast<const double*>(in) + body) : static_cast<const void*>(static_cast<const uint8_t*>(in) + body);
        if (sd) sd += body;
        out_val += body;
        if (out_std) out_std += body;
        if (out_idx) out_idx += body;
        n -= body;
    }
    const unsigned grid = stream_grid(f64in ? n : (n + 1) / 2, 256, 8);
#define HM_LIN(F, S) hipLaunchKernelGGL((k_linearize<F, S>), dim3(grid), dim3(256), 0, st, in, sd, icrf, icrf_diff, \
                                        out_val, out_std, out_idx, n, C, lut_stride)
    if (f64in) { if (with_std) HM_LIN(true, true); else HM_LIN(true, false); }
    else       { if (with_std) HM_LIN(false, true); else HM_LIN(false, false); }
#undef HM_LIN
    return launch_status();
}

extern "C" int hm_linearize_u8(const uint8_t* dn, const double* std, const double* icrf, const double* icrf_diff,
                               double* out_val, double* out_std, int64_t n, int C, int lut_stride, void* stream) {
    return linearize_common(false, dn, std, icrf, icrf_diff, out_val, out_std, nullptr, n, C, lut_stride, stream);
}

extern "C" int hm_linearize_f64(const double* v, const double* std, const double* icrf, const double* icrf_diff,
                                double* out_val, double* out_std, uint8_t* out_idx,
                                int64_t n, int C, int lut_stride, void* stream) {
    return linearize_common(true, v, std, icrf, icrf_diff, out_val, out_std, out_idx, n, C, lut_stride, stream);
}
