// hm_energy.hip - the ICRF-calibration energy function on the device (gfx950), for a whole population of
// candidate ICRFs per launch: _energy_function + analyze_linearity,
// modules/ICRF_calibration_exposure.py:66-145,148-201.
//
//   per candidate b (a 256-entry ICRF of one channel), frame pair i < j, pixel p:
//       v_i = ICRF_b[dn[p, i]], NaN if v_i < ICRF_b[lower] or v_i > ICRF_b[upper]        (:96-97,:186-194)
//       ratio = t_i / t_j                                                                  (:100)
//       scaled = v_j * ratio                                                               (:111)
//       d = v_i - scaled;  relative: d /= scaled;  a = |d|                                (:114-120)
//       with std:  sigma = sqrt((s_i / scaled)^2 + ((v_i * s_j) / (ratio * v_j^2))^2)     (:127)
//                  (absolute: sqrt(s_i^2 + (ratio * s_j)^2)                                :129)
//                  w = 1 / sigma where a is finite and sigma != 0, else excluded          (:133-134)
//                  result[i, j] = sum(a * w) / sum(w), NaN when sum(w) == 0               (general_functions.py:164-174)
//       without:   result[i, j] = nanmean(a)                                              (:138)
//   energy_b = nanmean over the pairs, +inf if that is NaN                                (:196-198)
//
// The stack is the reference's (X, Y, N) layout flattened to (P, N): a pixel's N samples are adjacent.
// Two launch geometries (chosen in hm_linearity_energy): pair-major, grid = (pixel chunks, pairs, candidates), every
// workgroup reduces one pair of one candidate over one pixel chunk (any N; a lone candidate still fills the chip
// with pairs x chunks workgroups); pixel-major for N <= 8, grid = (pixel chunks, candidates), all pairs of a pixel
// from registers (see k_energy_pixel). The reference evaluates one candidate at a time on the host. The sums are
// formed in a fixed order (lane-strided, shuffle tree, chunk order), so results are reproducible run to run;
// they differ from NumPy's pairwise summation in the last bits (tests: 1e-12 relative).
#include "hm_common.h"

namespace hm {

struct EnergyK {
    const uint8_t* dn;        // (P, N)
    const double* sd;         // (P, N) or null
    const double* icrf;       // (B, 256)
    const uint8_t* valid;     // (B) or null: 0 = candidate rejected on the host, energy preset to +inf
    double* partial;          // (B, pairs, chunks, 2)
    int64_t P;
    int32_t N, lower, upper, relative, chunks;
    double t[HM_MAX_FRAMES];
};

// pixel-major variant (N <= kEnergyRegFrames): per-pair ratio t_i / t_j and its reciprocal from the host
constexpr int kEnergyRegFrames = 8;
constexpr int kEnergyRegPairs = kEnergyRegFrames * (kEnergyRegFrames - 1) / 2;
struct EnergyPairs {
    double ratio[kEnergyRegPairs];
    double inv_ratio[kEnergyRegPairs];
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// pair index p (row-major upper triangle, np.triu_indices(N, 1) order) -> (i, j)
__device__ __forceinline__ void pair_of(int p, int N, int& i, int& j) {
    int row = 0, left = p;
    while (left >= N - 1 - row) { left -= N - 1 - row; ++row; }
    i = row; j = row + 1 + left;
}

template <bool STD>
__global__ __launch_bounds__(256) void k_energy_partial(const EnergyK a) {
    __shared__ double lut[256];
    __shared__ double red[4][2];
    const int b = blockIdx.z;
    double* out = a.partial + ((static_cast<int64_t>(b) * gridDim.y + blockIdx.y) * a.chunks + blockIdx.x) * 2;
    if (a.valid && !a.valid[b]) {                     // uniform per workgroup
        if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 0.0; }
        return;
    }
    lut[threadIdx.x] = a.icrf[static_cast<int64_t>(b) * 256 + threadIdx.x];
    __syncthreads();
    int i, j;
    pair_of(blockIdx.y, a.N, i, j);
    const double lo = lut[a.lower], hi = lut[a.upper];
    const double ratio = a.t[i] / a.t[j];
    const int N = a.N;

    double num = 0.0, den = 0.0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < a.P; p += stride) {
        const uint8_t* px = a.dn + p * N;
        double vi = lut[px[i]], vj = lut[px[j]];
        if (vi < lo || vi > hi) vi = __builtin_nan("");
        if (vj < lo || vj > hi) vj = __builtin_nan("");
        const double scaled = vj * ratio;
        double d = vi - scaled;
        if (a.relative) d = d / scaled;
        const double ad = fabs(d);
        if (STD) {
            const double si = a.sd[p * N + i], sj = a.sd[p * N + j];
            double sigma;
            if (a.relative) {
                const double u = si / scaled;
                const double v = (vi * sj) / (ratio * (vj * vj));
                sigma = sqrt(u * u + v * v);
            } else {
                const double v = ratio * sj;
                sigma = sqrt(si * si + v * v);
            }
            const bool finite = isfinite(ad) && sigma != 0.0;
            const double w = 1.0 / sigma;
            if (finite && w == w) { num += ad * w; den += w; }
        } else {
            if (ad == ad) { num += ad; den += 1.0; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double s0 = wave_sum(num), s1 = wave_sum(den);
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        out[1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
}

// Pixel-major evaluation for N <= 8 frames: a thread owns pixels, reads their N samples (and stds) once, maps them
// through the candidate's ICRF and evaluates ALL N(N-1)/2 pairs from registers, with the per-pair sums in registers
// too (2 x 28 float64 at most). Against the pair-major kernel above this removes the N-fold re-reads of the stack
// (27 GB of HBM traffic per 75-candidate launch on a 1024 x 1024 x 7 stack, profiles/r01e_producers_pmc.json) and
// the four IEEE divisions + square root per pair-pixel: 1 / v_j is formed once per (pixel, frame), 1 / (v_j ratio) is
// a product, and the weight 1 / sigma is one reciprocal square root. Those products differ from the reference's
// quotients by an ulp or two per term (tests: 1e-12 relative on the pair results).
// 1 / sqrt(q) for the weights of the pixel-major kernel: the hardware estimate (about 2^-24) and ONE third-order step,
// y0 (1 + e/2 + 3 e^2 / 8) with e = 1 - q y0^2 - 8 instructions and at most 1.25 ulp (checked on 2 M operands against long-double
// references, as for the pair statistics) where the library's rsqrt() costs about twenty; q = 0 / inf keep the estimate (inf / 0).
__device__ __forceinline__ double rsqrt_third_order(double q) {
    const double y0 = __builtin_amdgcn_rsq(q);
    const double e = fma(-(q * y0), y0, 1.0);
    const double y = fma(y0, fma(0.375, e, 0.5) * e, y0);
    return __builtin_amdgcn_class(y0, 0x264) ? y0 : y;                  // -inf | -0 | +0 | +inf
}

// 1 / x for the per-(pixel, frame) reciprocals of the pixel-major kernel: the hardware estimate and ONE second-order step, r0 (1 + e + e^2)
// with e = 1 - x r0 - 4 instructions and at most 1.00 ulp (hm_stats.hip: rcp_newton, checked on 2 M operands) where the IEEE division
// expansion costs eleven, two of them quarter-rate; x = 0 / inf / NaN keep the estimate (inf / 0 / NaN, the IEEE answers).
#ifndef HM_ENERGY_RCP
#define HM_ENERGY_RCP 1
#endif
__device__ __forceinline__ double rcp_second_order(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    const double r = fma(r0, fma(e, e, e), r0);
    return __builtin_amdgcn_class(r0, 0x264) ? r0 : r;                   // -inf | -0 | +0 | +inf
}

template <int N, bool STD>
__global__ __launch_bounds__(256) void k_energy_pixel(const EnergyK a, const EnergyPairs pr) {
    constexpr int P = N * (N - 1) / 2;
    __shared__ double lut[256];
    __shared__ double red[4][2 * P];
    const int b = blockIdx.y;
    const int pairs = P;
    auto out_of = [&](int p) { return a.partial + ((static_cast<int64_t>(b) * pairs + p) * a.chunks + blockIdx.x) * 2; };
    if (a.valid && !a.valid[b]) {
        if (threadIdx.x < P) { double* o = out_of(threadIdx.x); o[0] = 0.0; o[1] = 0.0; }
        return;
    }
    lut[threadIdx.x] = a.icrf[static_cast<int64_t>(b) * 256 + threadIdx.x];
    __syncthreads();
    const double lo = lut[a.lower], hi = lut[a.upper];
    const bool rel = a.relative != 0;
    double num[P], den[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { num[p] = 0.0; den[p] = 0.0; }

    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t px = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; px < a.P; px += stride) {
        const uint8_t* q = a.dn + px * N;
        double v[N], rv[N], s[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double x = lut[q[i]];
            if (x < lo || x > hi) x = __builtin_nan("");                       // :96-97
            v[i] = x;
            rv[i] = HM_ENERGY_RCP ? rcp_second_order(x) : 1.0 / x;
            s[i] = STD ? a.sd[px * N + i] : 0.0;
        }
        int p = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = i + 1; j < N; ++j, ++p) {
                const double ratio = pr.ratio[p];
                const double scaled = v[j] * ratio;                             // :111
                double d = v[i] - scaled;                                        // :114
                double inv_s = 0.0;
                if (rel) { inv_s = rv[j] * pr.inv_ratio[p]; d = d * inv_s; }     // :117  d / scaled
                const double ad = fabs(d);                                       // :120
                if (STD) {
                    double qq;
                    if (rel) {
                        const double u = s[i] * inv_s;                           // s_i / scaled
                        const double w2 = ((v[i] * s[j]) * inv_s) * rv[j];       // (v_i s_j) / (ratio v_j^2)        :127
                        qq = u * u + w2 * w2;
                    } else {
                        const double w2 = ratio * s[j];
                        qq = s[i] * s[i] + w2 * w2;                              // :129
                    }
                    const double w = rsqrt_third_order(qq);                      // 1 / sigma
                    const bool ok = isfinite(ad) && qq != 0.0 && w == w;         // :133-134, general_functions.py:164
                    if (ok) { num[p] += ad * w; den[p] += w; }
                } else {
                    if (ad == ad) { num[p] += ad; den[p] += 1.0; }               // :138
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const double s0 = wave_sum(num[p]), s1 = wave_sum(den[p]);
        if (lane == 0) { red[wave][2 * p] = s0; red[wave][2 * p + 1] = s1; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * P) {
        const int p = threadIdx.x >> 1, k = threadIdx.x & 1;
        out_of(p)[k] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// one wave per candidate: pair results (chunks summed in order) and the NaN-ignoring mean over pairs
__global__ __launch_bounds__(64) void k_energy_final(const double* __restrict__ partial, const uint8_t* __restrict__ valid,
                                                     int pairs, int chunks, double* __restrict__ out_pairs,
                                                     double* __restrict__ out_energy) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool ok = !valid || valid[b];
    double sum = 0.0, cnt = 0.0;
    for (int p = lane; p < pairs; p += 64) {
        const double* q = partial + (static_cast<int64_t>(b) * pairs + p) * chunks * 2;
        double num = 0.0, den = 0.0;
        for (int c = 0; c < chunks; ++c) { num += q[2 * c]; den += q[2 * c + 1]; }
        const double r = ok ? num / den : __builtin_nan("");           // 0 / 0 = NaN: no contributing pixel
        if (out_pairs) out_pairs[static_cast<int64_t>(b) * pairs + p] = r;
        if (r == r) { sum += r; cnt += 1.0; }
    }
    sum = wave_sum(sum); cnt = wave_sum(cnt);
    if (lane == 0) {
        const double e = sum / cnt;
        out_energy[b] = (ok && e == e) ? e : __builtin_inf();
    }
}

static int energy_chunks(int64_t P) {
    int64_t c = (P + 1023) / 1024;                  // >= 4 pixels per thread before splitting further
    if (c < 1) c = 1;
    return static_cast<int>(c > 64 ? 64 : c);
}

}  // namespace hm

using namespace hm;

extern "C" size_t hm_linearity_energy_workspace_bytes(int64_t n_pixels, int n_frames, int n_candidates) {
    if (n_pixels < 0 || n_frames < 2 || n_candidates < 1) return 0;
    const int64_t pairs = static_cast<int64_t>(n_frames) * (n_frames - 1) / 2;
    return static_cast<size_t>(n_candidates) * pairs * energy_chunks(n_pixels) * 2 * sizeof(double);
}

extern "C" int hm_linearity_energy(const uint8_t* dn, const double* std, const double* exposures, const double* icrf,
                                   const uint8_t* valid, int n_candidates, int lower, int upper, int use_relative,
                                   int64_t n_pixels, int n_frames, double* out_pairs, double* out_energy,
                                   void* workspace, void* stream) {
    if (n_candidates < 0 || n_pixels < 0) return HM_EINVAL;
    if (n_frames < 2 || n_frames > HM_MAX_FRAMES) return HM_ESHAPE;
    if (lower < 0 || lower > 255 || upper < 0 || upper > 255) return HM_EINVAL;
    if (n_candidates == 0) return HM_OK;
    if (n_candidates > 65535) return HM_EUNSUPPORTED;
    if (!dn || !exposures || !icrf || !out_energy || !workspace) return HM_EINVAL;
    const int pairs = n_frames * (n_frames - 1) / 2;
    EnergyK k{};
    k.dn = dn; k.sd = std; k.icrf = icrf; k.valid = valid; k.partial = static_cast<double*>(workspace);
    k.P = n_pixels; k.N = n_frames; k.lower = lower; k.upper = upper; k.relative = use_relative ? 1 : 0;
    k.chunks = energy_chunks(n_pixels);
    for (int i = 0; i < n_frames; ++i) k.t[i] = exposures[i];
    // pixel-major kernel when there are enough candidates (or few enough pixels) for chunks x candidates workgroups to
    // fill the chip; a lone candidate on a large stack keeps the pair-major kernel's pairs x chunks workgroups
    if (n_frames <= kEnergyRegFrames && (n_candidates >= 8 || n_pixels <= 16384)) {
        EnergyPairs pr{};
        int p = 0;
        for (int i = 0; i < n_frames; ++i)
            for (int j = i + 1; j < n_frames; ++j, ++p) {
                pr.ratio[p] = exposures[i] / exposures[j];                       // :100
                pr.inv_ratio[p] = 1.0 / pr.ratio[p];
            }
        const dim3 grid(k.chunks, n_candidates);
#define HM_EPX(n) case n: if (std) hipLaunchKernelGGL((k_energy_pixel<n, true>), grid, dim3(256), 0, as_stream(stream), k, pr); \
                          else hipLaunchKernelGGL((k_energy_pixel<n, false>), grid, dim3(256), 0, as_stream(stream), k, pr); break;
        switch (n_frames) { HM_EPX(2) HM_EPX(3) HM_EPX(4) HM_EPX(5) HM_EPX(6) HM_EPX(7) HM_EPX(8) default: return HM_EUNSUPPORTED; }
#undef HM_EPX
    } else {
        const dim3 grid(k.chunks, pairs, n_candidates);
        if (std) hipLaunchKernelGGL(k_energy_partial<true>, grid, dim3(256), 0, as_stream(stream), k);
        else     hipLaunchKernelGGL(k_energy_partial<false>, grid, dim3(256), 0, as_stream(stream), k);
    }
    int rc = launch_status();
    if (rc != HM_OK) return rc;
    hipLaunchKernelGGL(k_energy_final, dim3(n_candidates), dim3(64), 0, as_stream(stream),
                       static_cast<const double*>(workspace), valid, pairs, k.chunks, out_pairs, out_energy);
    return launch_status();
}
