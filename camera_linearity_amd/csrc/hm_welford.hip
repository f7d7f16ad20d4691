// hm_welford.hip - streaming per-pixel mean / M2 over video frames (gfx950), the producer of the mean and
// STD frames the merge consumes: welford_algorithm, modules/video_processing.py:161-219.
//
//   per frame k (count n = count_before + k + 1):
//       f     = ICRF[dn, c]            (:200-201)      or   dn / 255   (:203)
//       delta = f - mean                                 (:205)
//       mean  = mean + delta / n                         (:206)
//       m2    = m2 + delta * (f - mean)                  (:208)
//   finalize:  mean -> around(mean * 255) as uint8       (:210-211)
//              std  -> around(sqrt(m2 / (n - 1)) / sqrt(n)) as uint8     (:214-215, as written: no * 255)
//
// One launch consumes K frames: the float64 state (16 or 8 B per element) is read and written once per launch
// instead of once per frame, so the traffic per element-frame is 1 + (16 or 32)/K bytes. The frame value comes
// from a 256 x C table in LDS ({dn/255} or the ICRF), two elements per lane (one ushort load per frame, 16-byte
// state accesses: every wave instruction covers whole 128-byte lines).
//
// delta / n must carry the bits of NumPy's IEEE division, but the divisor is the same for every element of a
// frame, so the correctly rounded reciprocal r = RN(1 / n) comes from the host per frame and the quotient is
//     q0 = RN(delta * r);   e = delta - q0 * n  (one FMA, exact);   q = RN(q0 + e * r)  (one FMA)
// which is the correctly rounded delta / n whenever r = RN(1 / n) and nothing underflows (Markstein's final
// division step; Brisebarre, Muller, Raina, IEEE TC 53(8) 2004, "division when the divisor is known in
// advance"). 3 dependent ops instead of the ~14-instruction IEEE sequence (FP64 VALU: 4 cycles per wave op).
// The range condition is checked, not assumed - once per element and launch, not per frame (round 4: the per-frame test was 2 of the 13.8
// VALU instructions of an element-frame in a kernel that is FP64-VALU bound): the fast path needs every non-zero delta >= 2^-900.
// With table entries that are 0 or in [2^-500, 2^500] (checked while the table is built), an incoming mean that is 0 or in
// [2^-540, 2^500] (checked at load) and n <= 2^40 (checked on the host), every later mean is 0 or >= 2^-632 in magnitude:
// m' = m (1 - 1/n) + f/n can lose 52 bits to ONE cancellation (two doubles >= 2^-541 differ by >= 2^-593 when they differ), after which
// a non-zero f/n >= 2^-540 dwarfs it and a run of f = 0 shrinks it by n0/n >= 2^-40 at most - so a non-zero delta = f - m is >= 2^-684.
// A lane whose state or table is outside those ranges (or not finite) redoes its elements with the full division (welford_exact).
#include "hm_common.h"

namespace hm {

typedef double f64x2 __attribute__((ext_vector_type(2)));

struct WelfordK {
    const uint8_t* frame[HM_MAX_FRAMES];
    const double* icrf;          // (256, C) or null
    double* mean;
    double* m2;                  // nullable
    int64_t n;
    int32_t n_frames, C;
    double count0;               // frames already folded into mean / m2
    double rcp[HM_MAX_FRAMES];   // RN(1 / (count0 + k + 1)), formed on the host
};

constexpr int kWelfordGroup = 8;             // frames whose loads / table gathers are issued together

// reference arithmetic with the IEEE division for `count` consecutive elements, all frames (rare path and odd tail)
template <bool M2>
__device__ __forceinline__ void welford_exact(const WelfordK& a, const double* t, int64_t e0, int count, double* mean, double* m2) {
    for (int j = 0; j < count; ++j) {
        const int64_t e = e0 + j;
        const int c = static_cast<int>(e % a.C);
        double m = a.mean[e], q = M2 ? a.m2[e] : 0.0, cnt = a.count0;
        for (int k = 0; k < a.n_frames; ++k) {
            cnt += 1.0;
            const double f = t[a.frame[k][e] + c * 256];
            const double delta = f - m;                                   // :205
            m = m + delta / cnt;                                          // :206
            if (M2) q = q + delta * (f - m);                              // :208
        }
        mean[j] = m; m2[j] = q;
    }
}

// G frames starting at k0 for the lane's two elements (fast division: the caller has checked its range, see the file header)
template <bool M2, int G, bool VEC>
__device__ __forceinline__ bool welford_group(const WelfordK& a, const double* t, int k0, int64_t e, uint32_t c0, uint32_t c1,
                                              double& cnt, double (&m)[2], double (&q)[2]) {
    uint32_t raw[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint8_t* p = a.frame[k0 + k] + e;
        raw[k] = VEC ? static_cast<uint32_t>(__builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p)))
                     : (p[0] | (static_cast<uint32_t>(p[1]) << 8));
    }
    double f[G][2];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        f[k][0] = t[(raw[k] & 255u) + c0 * 256u];
        f[k][1] = t[(raw[k] >> 8) + c1 * 256u];
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        cnt += 1.0;
        const double r = a.rcp[k0 + k];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double delta = f[k][j] - m[j];                                  // :205
            const double q0 = delta * r;
            const double quot = fma(fma(-q0, cnt, delta), r, q0);                 // == delta / cnt
            m[j] = m[j] + quot;                                                   // :206
            if (M2) q[j] = q[j] + delta * (f[k][j] - m[j]);                       // :208
        }
    }
    return ok;
}

template <bool M2>
__global__ __launch_bounds__(256) void k_welford(const WelfordK a) {
    __shared__ double t[256 * HM_MAX_CHANNELS];
    const int C = a.C;
    int odd_table = 0;
    for (int i = threadIdx.x; i < 256 * C; i += blockDim.x) {           // t[c][dn]: the gather index is dn + 256 c, no multiply
        const double v = a.icrf ? a.icrf[i] : static_cast<double>(i / C) / 255.0;
        t[(i % C) * 256 + i / C] = v;
        odd_table |= !(fabs(v) <= 0x1p500) || (v != 0.0 && fabs(v) < 0x1p-500);       // non-finite, huge or tiny entries
    }
    const bool table_ok = !__syncthreads_or(odd_table);

    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t units = a.n / 2;
    bool vec_ok = aligned_dev(a.mean, 16) && (!M2 || aligned_dev(a.m2, 16));
    for (int k = 0; k < a.n_frames; ++k) vec_ok = vec_ok && aligned_dev(a.frame[k], 2);

    for (int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u < units; u += stride) {
        const int64_t e = 2 * u;
        const uint32_t c0 = static_cast<uint32_t>(e % C), c1 = (c0 + 1u) % static_cast<uint32_t>(C);
        double m[2], q[2] = {0.0, 0.0};
        if (vec_ok) {
            const f64x2 mv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(a.mean + e));
            m[0] = mv.x; m[1] = mv.y;
            if (M2) { const f64x2 qv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(a.m2 + e)); q[0] = qv.x; q[1] = qv.y; }
        } else {
            m[0] = a.mean[e]; m[1] = a.mean[e + 1];
            if (M2) { q[0] = a.m2[e]; q[1] = a.m2[e + 1]; }
        }
        auto mean_ok = [](double x) { const double ax = fabs(x); return x == 0.0 || (ax >= 0x1p-540 && ax <= 0x1p500); };
        bool ok = table_ok && mean_ok(m[0]) && mean_ok(m[1]) && fabs(q[0]) <= 0x1p900 && fabs(q[1]) <= 0x1p900;
        double cnt = a.count0;
        int k0 = 0;
        if (vec_ok) {
            for (; k0 + kWelfordGroup <= a.n_frames; k0 += kWelfordGroup)
                ok = welford_group<M2, kWelfordGroup, true>(a, t, k0, e, c0, c1, cnt, m, q) && ok;
            for (; k0 < a.n_frames; ++k0) ok = welford_group<M2, 1, true>(a, t, k0, e, c0, c1, cnt, m, q) && ok;
        } else {
            for (; k0 < a.n_frames; ++k0) ok = welford_group<M2, 1, false>(a, t, k0, e, c0, c1, cnt, m, q) && ok;
        }
        if (__builtin_expect(!ok, 0)) {                // the state in memory is still the launch's input
            welford_exact<M2>(a, t, e, 2, m, q);
        }
        if (vec_ok) {
            f64x2 mv; mv.x = m[0]; mv.y = m[1];
            __builtin_nontemporal_store(mv, reinterpret_cast<f64x2*>(a.mean + e));
            if (M2) { f64x2 qv; qv.x = q[0]; qv.y = q[1]; __builtin_nontemporal_store(qv, reinterpret_cast<f64x2*>(a.m2 + e)); }
        } else {
            a.mean[e] = m[0]; a.mean[e + 1] = m[1];
            if (M2) { a.m2[e] = q[0]; a.m2[e + 1] = q[1]; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (a.n & 1)) {
        const int64_t e = a.n - 1;
        double m[1], q[1];
        welford_exact<M2>(a, t, e, 1, m, q);
        a.mean[e] = m[0];
        if (M2) a.m2[e] = q[0];
    }
}

// around() = round half to even; astype(uint8) of an in-range value. Out-of-range / NaN inputs follow the
// convention of hm_linearize_f64's index: convert to a 64-bit integer and keep the low byte (NaN -> 0).
__device__ __forceinline__ uint8_t round_to_u8(double x) {
    const double r = rint(x);
    if (!(r == r)) return 0;
    if (r >= 9.2e18 || r <= -9.2e18) return 0;
    return static_cast<uint8_t>(static_cast<uint64_t>(static_cast<int64_t>(r)) & 255u);
}

__global__ __launch_bounds__(256) void k_welford_finalize(const double* __restrict__ mean, const double* __restrict__ m2,
                                                          double count, uint8_t* __restrict__ out_mean,
                                                          uint8_t* __restrict__ out_std, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const double denom = count - 1.0, root_n = sqrt(count);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        if (out_mean) out_mean[e] = round_to_u8(mean[e] * 255.0);                        // :210-211
        if (out_std) out_std[e] = round_to_u8(sqrt(m2[e] / denom) / root_n);           // :214-215
    }
}

}  // namespace hm

using namespace hm;

extern "C" int hm_welford_update(const void* const* frames, int n_frames, int64_t count_before, const double* icrf,
                                 double* mean, double* m2, int64_t n_elems, int C, void* stream) {
    if (n_frames < 0 || n_frames > HM_MAX_FRAMES || count_before < 0 || n_elems < 0) return HM_EINVAL;
    if (count_before > (int64_t{1} << 40)) return HM_EUNSUPPORTED;       // div_by_count's acceptance test assumes n << 2^53
    if (C < 1 || C > HM_MAX_CHANNELS) return HM_ESHAPE;
    if (n_elems % C != 0) return HM_ESHAPE;
    if (n_frames == 0 || n_elems == 0) return HM_OK;
    if (!frames || !mean) return HM_EINVAL;
    WelfordK k{};
    for (int i = 0; i < n_frames; ++i) {
        if (!frames[i]) return HM_EINVAL;
        k.frame[i] = static_cast<const uint8_t*>(frames[i]);
    }
    k.icrf = icrf; k.mean = mean; k.m2 = m2; k.n = n_elems; k.n_frames = n_frames; k.C = C;
    k.count0 = static_cast<double>(count_before);
    for (int i = 0; i < n_frames; ++i) k.rcp[i] = 1.0 / (k.count0 + static_cast<double>(i + 1));
    const unsigned grid = stream_grid((n_elems + 1) / 2, 256, 8);
    if (m2) hipLaunchKernelGGL(k_welford<true>, dim3(grid), dim3(256), 0, as_stream(stream), k);
    else    hipLaunchKernelGGL(k_welford<false>, dim3(grid), dim3(256), 0, as_stream(stream), k);
    return launch_status();
}

extern "C" int hm_welford_finalize(const double* mean, const double* m2, int64_t count, uint8_t* out_mean,
                                   uint8_t* out_std, int64_t n_elems, void* stream) {
    if (n_elems < 0 || count < 1) return HM_EINVAL;
    if (n_elems == 0) return HM_OK;
    if ((out_mean && !mean) || (out_std && !m2)) return HM_EINVAL;
    if (out_std && count < 2) return HM_EINVAL;            // m2 / (n - 1) needs two frames (:214)
    const unsigned grid = stream_grid(n_elems, 256, 8);
    hipLaunchKernelGGL(k_welford_finalize, dim3(grid), dim3(256), 0, as_stream(stream), mean, m2,
                       static_cast<double>(count), out_mean, out_std, n_elems);
    return launch_status();
}

extern "C" int64_t hm_welford_algorithmic_bytes(int n_frames, int with_m2, int64_t n_elems) {
    return n_elems * (static_cast<int64_t>(n_frames) + (with_m2 ? 32 : 16));
}
