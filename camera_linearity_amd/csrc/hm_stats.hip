// hm_stats.hip - the linearity-statistics row of SURVEY.md 8(f)-1 as HIP kernels (gfx950):
//   hm_apply_thresholds      AbstractMeasurand.apply_thresholds          modules/measurand.py:375-428
//   hm_compute_difference    AbstractMeasurand.compute_difference        modules/measurand.py:620-655
//   hm_interpolate           AbstractMeasurand.interpolate               modules/measurand.py:657-681
//   hm_channel_statistics    compute_dimension_statistics(axis = all but the last)   modules/measurand.py:318-350
// Streaming, HBM-bound; the statistics are two deterministic reduction passes (per-workgroup partials by wave
// shuffles + LDS, then one workgroup), NaNs ignored exactly as np.nansum / np.nanmean / np.nanstd do.
#include "hm_common.h"
#include <algorithm>

namespace hm {

struct ChanLimits { double lo[HM_MAX_CHANNELS]; double hi[HM_MAX_CHANNELS]; };

// In place; two elements per lane (16-byte loads; a store only where something changes - thresholds usually clip a
// minority), running channel counter.
__global__ __launch_bounds__(256) void k_thresholds(double* __restrict__ val, double* __restrict__ sd, const ChanLimits lim,
                                                    int64_t n, int C) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int64_t units = n / 2;
    const bool vec_ok = aligned_dev(val, 16) && (!sd || aligned_dev(sd, 16));
    const uint32_t uC = static_cast<uint32_t>(C);
    const int64_t u0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t c0 = static_cast<uint32_t>((2 * u0) % C);
    const uint32_t cstep = static_cast<uint32_t>((2 * stride) % C);
    auto lo_of = [&](uint32_t c) { double r = lim.lo[0]; for (int k = 1; k < HM_MAX_CHANNELS; ++k) r = c == static_cast<uint32_t>(k) ? lim.lo[k] : r; return r; };
    auto hi_of = [&](uint32_t c) { double r = lim.hi[0]; for (int k = 1; k < HM_MAX_CHANNELS; ++k) r = c == static_cast<uint32_t>(k) ? lim.hi[k] : r; return r; };
    for (int64_t u = u0; u < units; u += stride) {
        const int64_t e = 2 * u;
        const uint32_t c1 = c0 + 1u == uC ? 0u : c0 + 1u;
        double v0, v1;
        if (vec_ok) { const f64x2 a = *reinterpret_cast<const f64x2*>(val + e); v0 = a.x; v1 = a.y; }
        else { v0 = val[e]; v1 = val[e + 1]; }
        const bool k0 = (v0 < lo_of(c0)) || (v0 > hi_of(c0));           // measurand.py:418 (NaN compares false: stays NaN)
        const bool k1 = (v1 < lo_of(c1)) || (v1 > hi_of(c1));
        if (k0) { val[e] = nan; if (sd) sd[e] = nan; }
        if (k1) { val[e + 1] = nan; if (sd) sd[e + 1] = nan; }
        c0 += cstep;
        if (c0 >= uC) c0 -= uC;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        const int64_t e = n - 1;
        const uint32_t c = static_cast<uint32_t>(e % C);
        const double v = val[e];
        if ((v < lo_of(c)) || (v > hi_of(c))) { val[e] = nan; if (sd) sd[e] = nan; }
    }
}

__global__ __launch_bounds__(256) void k_difference(const double* __restrict__ x, const double* __restrict__ sx,
                                                    const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                    double* __restrict__ ad, double* __restrict__ ads,
                                                    double* __restrict__ rd, double* __restrict__ rds, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const double xv = x[e], yv = y[e];
        const double scale = mult * yv;                     // :634
        const double a = xv - scale;                        // :635
        ad[e] = a;
        rd[e] = a / scale;                                  // :636
        if (ads) {
            const double xs = sx ? sx[e] : 0.0, ys = sy ? sy[e] : 0.0;
            const double m1 = mult * ys;
            ads[e] = sqrt(xs * xs + m1 * m1);               // :652
            const double u1 = xs / (mult * yv);
            const double u2 = (ys * xv) / (mult * (yv * yv));
            rds[e] = sqrt(u1 * u1 + u2 * u2);               // :653
        }
    }
}

__global__ __launch_bounds__(256) void k_interpolate(const double* __restrict__ x0, const double* __restrict__ s0,
                                                     const double* __restrict__ x1, const double* __restrict__ s1,
                                                     double y0, double y1, double y, double* __restrict__ out,
                                                     double* __restrict__ out_std, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const double a = y1 - y, b = y - y0, d = y1 - y0;
    const double ca = (a / d) * (a / d), cb = (b / d) * (b / d);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        out[e] = (x0[e] * a + x1[e] * b) / d;               // :665
        if (out_std) out_std[e] = sqrt((s0 ? s0[e] : 0.0) * ca + (s1 ? s1[e] : 0.0) * cb);   // :679 as written
    }
}

// ---- statistics -------------------------------------------------------------------------------
constexpr int kStatBlocks = 2040;                           // a multiple of 12: total threads divisible by C = 1..4

static int stat_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    g = ((g + 11) / 12) * 12;
    return static_cast<int>(g < kStatBlocks ? g : kStatBlocks);
}
constexpr int kStatVals = 4;                                // per channel: 4 partial sums

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-level reduction of acc[C][4] -> partial[block][C][4]
__device__ __forceinline__ void block_reduce_store(double (&acc)[HM_MAX_CHANNELS][kStatVals], double* partial) {
    __shared__ double red[4][HM_MAX_CHANNELS * kStatVals];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < HM_MAX_CHANNELS; ++c)
#pragma unroll
        for (int k = 0; k < kStatVals; ++k) {
            const double s = wave_sum_d(acc[c][k]);
            if (lane == 0) red[wave][c * kStatVals + k] = s;
        }
    __syncthreads();
    if (threadIdx.x < HM_MAX_CHANNELS * kStatVals)
        partial[blockIdx.x * HM_MAX_CHANNELS * kStatVals + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// pass 1: weighted: [sum w, sum v*w, sum std (non-nan), count std non-nan]; unweighted: [sum v, count, 0, 0]
// pass 2: weighted: [sum w (v-mean)^2, 0,0,0];                              unweighted: [sum (v-mean)^2, 0,0,0]
// The launch uses a total thread count that is a multiple of C (grid rounded to a multiple of 12 workgroups),
// so a thread's elements all have the same channel c_t = first_element % C and it keeps 4 running sums, not 4*C.
template <int PASS>
__global__ __launch_bounds__(256) void k_stats(const double* __restrict__ val, const double* __restrict__ sd, int64_t n, int C,
                                               const double* __restrict__ mean /*pass 2*/, double* __restrict__ partial) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int ct = static_cast<int>(first % C);
    const double mu = PASS == 2 ? mean[ct] : 0.0;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    constexpr int UN = 4;                                   // independent loads in flight per lane
    for (int64_t e = first; e < n; e += UN * stride) {
        double vv[UN], sv[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t q = e + u * stride;
            ok[u] = q < n;
            vv[u] = ok[u] ? __builtin_nontemporal_load(val + q) : 0.0;
            sv[u] = (ok[u] && sd) ? __builtin_nontemporal_load(sd + q) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ok[u]) continue;
            const double v = vv[u];
            if (sd) {
                const double s = sv[u];
                const double w = 1.0 / s;                                    // :342
                if (PASS == 1) {
                    if (w == w) t0 += w;                                    // nansum(weights)
                    const double vw = v * w;
                    if (vw == vw) t1 += vw;                                 // nansum(values * weights)
                    if (s == s) { t2 += s; t3 += 1.0; }                     // nanmean(stds)
                } else {
                    const double d = v - mu;
                    const double q2 = w * (d * d);                          // :345
                    if (q2 == q2) t0 += q2;
                }
            } else {
                if (PASS == 1) { if (v == v) { t0 += v; t1 += 1.0; } }
                else { const double d = v - mu; const double q2 = d * d; if (q2 == q2) t0 += q2; }
            }
        }
    }
    double acc[HM_MAX_CHANNELS][kStatVals];
#pragma unroll
    for (int k = 0; k < HM_MAX_CHANNELS; ++k) {
        const bool me = k == ct;
        acc[k][0] = me ? t0 : 0.0; acc[k][1] = me ? t1 : 0.0; acc[k][2] = me ? t2 : 0.0; acc[k][3] = me ? t3 : 0.0;
    }
    block_reduce_store(acc, partial);
}

// Sums `ncols` columns of partial[nblocks][ncols] with 256 threads: thread t adds rows t/ncols, t/ncols + R, ...
// (R = 256/ncols row groups), then the R partial sums of a column are added in a fixed order -> deterministic.
template <int NCOLS>
__device__ __forceinline__ void column_sums(const double* __restrict__ partial, int nblocks, double* sums /*LDS [NCOLS]*/) {
    __shared__ double part[256];
    constexpr int R = 256 / NCOLS;
    const int col = threadIdx.x % NCOLS, rg = threadIdx.x / NCOLS;
    double s = 0.0;
    for (int b = rg; b < nblocks; b += R) s += partial[b * NCOLS + col];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < NCOLS) {
        double tot = 0.0;
        for (int r = 0; r < R; ++r) tot += part[r * NCOLS + threadIdx.x];
        sums[threadIdx.x] = tot;
    }
    __syncthreads();
}

// final: sums the partials in block order; stage 1 -> out[0..C) = mean, scratch keeps the denominators;
//        stage 2 -> out[C..2C) = std, out[2C..3C) = error (nanmean of stds, weighted case only; NaN otherwise)
template <int PASS>
__global__ __launch_bounds__(256) void k_stats_final(const double* __restrict__ partial, int nblocks, int C, int weighted,
                                                     double* __restrict__ out, double* __restrict__ denom) {
    const int t = threadIdx.x;
    __shared__ double sums[HM_MAX_CHANNELS * kStatVals];
    column_sums<HM_MAX_CHANNELS * kStatVals>(partial, nblocks, sums);
    if (t < C) {
        const double* q = sums + t * kStatVals;
        if (PASS == 1) {
            if (weighted) { out[t] = q[1] / q[0]; denom[t] = q[0]; out[2 * C + t] = q[2] / q[3]; }
            else { out[t] = q[0] / q[1]; denom[t] = q[1]; out[2 * C + t] = __longlong_as_double(0x7ff8000000000000ll); }
        } else {
            out[C + t] = sqrt(q[0] / denom[t]);
        }
    }
}

// ---- pair-fused linearity statistics ----------------------------------------------------------
// ExposurePair.compute_difference + compute_stats(axis=(0,1)) (modules/exposure_series.py:33-54, called per pair
// by process_linearity :443-446) without materialising the two difference images: each pass reads x, y (and
// their stds) once and reduces both the absolute and the relative difference. out: 6*C doubles
// [abs mean | abs std | abs error | rel mean | rel std | rel error].
constexpr int kPairVals = 8;          // per channel: 4 sums for the absolute, 4 for the relative difference

__device__ __forceinline__ void pair_terms(double xv, double xs, double yv, double ys, double mult, bool with_std,
                                           double& a, double& as, double& r, double& rs) {
    const double scale = mult * yv;                     // measurand.py:634
    a = xv - scale;                                     // :635
    r = a / scale;                                      // :636
    if (with_std) {
        const double m1 = mult * ys;
        as = sqrt(xs * xs + m1 * m1);                   // :652
        const double u1 = xs / (mult * yv);
        const double u2 = (ys * xv) / (mult * (yv * yv));
        rs = sqrt(u1 * u1 + u2 * u2);                   // :653
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_pair_stats(const double* __restrict__ x, const double* __restrict__ sx,
                                                    const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                    int64_t n, int C, const double* __restrict__ means /*pass 2: out[]*/,
                                                    double* __restrict__ partial) {
    __shared__ double red[4][HM_MAX_CHANNELS * kPairVals];
    const bool with_std = sx || sy;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;       // a multiple of C (see k_stats)
    const int64_t first = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int ct = static_cast<int>(first % C);
    const double mu[2] = {PASS == 2 ? means[ct] : 0.0, PASS == 2 ? means[3 * C + ct] : 0.0};
    double t[kPairVals] = {};
    constexpr int UN = 2;
    for (int64_t e0 = first; e0 < n; e0 += UN * stride) {
        double xv[UN], yv[UN], xs[UN], ys[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t q = e0 + u * stride;
            ok[u] = q < n;
            xv[u] = ok[u] ? __builtin_nontemporal_load(x + q) : 0.0;
            yv[u] = ok[u] ? __builtin_nontemporal_load(y + q) : 1.0;
            xs[u] = (ok[u] && sx) ? __builtin_nontemporal_load(sx + q) : 0.0;
            ys[u] = (ok[u] && sy) ? __builtin_nontemporal_load(sy + q) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ok[u]) continue;
            double a, as = 0.0, r, rs = 0.0;
            pair_terms(xv[u], xs[u], yv[u], ys[u], mult, with_std, a, as, r, rs);
            const double vv[2] = {a, r}, ss[2] = {as, rs};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double v = vv[h];
                if (with_std) {
                    const double w = 1.0 / ss[h];
                    if (PASS == 1) {
                        if (w == w) t[4 * h] += w;
                        const double vw = v * w;
                        if (vw == vw) t[4 * h + 1] += vw;
                        if (ss[h] == ss[h]) { t[4 * h + 2] += ss[h]; t[4 * h + 3] += 1.0; }
                    } else {
                        const double d = v - mu[h];
                        const double q2 = w * (d * d);
                        if (q2 == q2) t[4 * h] += q2;
                    }
                } else {
                    if (PASS == 1) { if (v == v) { t[4 * h] += v; t[4 * h + 1] += 1.0; } }
                    else { const double d = v - mu[h]; const double q2 = d * d; if (q2 == q2) t[4 * h] += q2; }
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < HM_MAX_CHANNELS; ++c)
#pragma unroll
        for (int k = 0; k < kPairVals; ++k) {
            const double s = wave_sum_d(c == ct ? t[k] : 0.0);
            if (lane == 0) red[wave][c * kPairVals + k] = s;
        }
    __syncthreads();
    if (threadIdx.x < HM_MAX_CHANNELS * kPairVals)
        partial[blockIdx.x * HM_MAX_CHANNELS * kPairVals + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <int PASS>
__global__ __launch_bounds__(256) void k_pair_final(const double* __restrict__ partial, int nblocks, int C, int weighted,
                                                    double* __restrict__ out, double* __restrict__ denom) {
    const int t = threadIdx.x;
    __shared__ double sums[HM_MAX_CHANNELS * kPairVals];
    column_sums<HM_MAX_CHANNELS * kPairVals>(partial, nblocks, sums);
    if (t < 2 * C) {
        const int h = t / C, c = t % C;
        const double* q = sums + c * kPairVals + 4 * h;
        double* o = out + 3 * C * h;
        if (PASS == 1) {
            if (weighted) { o[c] = q[1] / q[0]; denom[t] = q[0]; o[2 * C + c] = q[2] / q[3]; }
            else { o[c] = q[0] / q[1]; denom[t] = q[1]; o[2 * C + c] = __longlong_as_double(0x7ff8000000000000ll); }
        } else {
            o[C + c] = sqrt(q[0] / denom[t]);
        }
    }
}

// ---- per-channel histogram (compute_channel_histogram, modules/measurand.py:430-469) ----------------
// np.histogram with `bins` equal-width bins on [lo, hi]: the bin of x is int((x - lo) * bins / (hi - lo)),
// corrected against the actual edges (np.linspace(lo, hi, bins + 1), passed in by the caller) exactly as
// numpy/lib/_histograms_impl.py does, the right edge inclusive. Non-finite values are skipped
// (measurand.py:453); with weights, elements whose std is 0 are skipped and the weight is 1/std (:457-460).
// Per-workgroup histograms in LDS (ds_add_f64), written as partials and column-summed: counts are exact,
// weighted sums reproducible up to the order of the LDS atomics inside a workgroup.
constexpr int kHistBlocks = 256;
constexpr int kHistMaxBins = 2048;        // bins * C doubles of LDS per workgroup (<= 64 KB)

__global__ __launch_bounds__(256) void k_hist(const double* __restrict__ val, const double* __restrict__ sd, int64_t n, int C,
                                              int chan_mask, const double* __restrict__ edges, int bins, double lo, double hi,
                                              double* __restrict__ partial /*[grid][C*bins]*/) {
    extern __shared__ double h[];
    const int nb = C * bins;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) h[i] = 0.0;
    __syncthreads();
    const double norm = static_cast<double>(bins) / (hi - lo);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int c = static_cast<int>(e % C);
        if (!((chan_mask >> c) & 1)) continue;
        const double x = val[e];
        if (!(fabs(x) <= 1.79769313486231570e308)) continue;              // isfinite
        double w = 1.0;
        if (sd) {
            const double s = sd[e];
            if (s == 0.0) continue;                                         // :457
            w = 1.0 / s;                                                    // :460
        }
        if (!(x >= lo && x <= hi)) continue;                                // outside the range: not counted
        int idx = static_cast<int>((x - lo) * norm);
        if (idx == bins) idx -= 1;
        if (x < edges[idx]) idx -= 1;
        else if (x >= edges[idx + 1] && idx != bins - 1) idx += 1;
        atomicAdd(&h[c * bins + idx], w);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) partial[static_cast<int64_t>(blockIdx.x) * nb + i] = h[i];
}

__global__ __launch_bounds__(256) void k_hist_final(const double* __restrict__ partial, int nblocks, int nb, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partial[static_cast<int64_t>(b) * nb + i];
        out[i] = s;
    }
}

// per-channel min / max of the finite (and, with std, non-zero-std) values: the default range of np.histogram
__global__ __launch_bounds__(256) void k_minmax(const double* __restrict__ val, const double* __restrict__ sd, int64_t n, int C,
                                                double* __restrict__ partial /*[grid][C][2]*/) {
    __shared__ double red[4][HM_MAX_CHANNELS * 2];
    double mn[HM_MAX_CHANNELS], mx[HM_MAX_CHANNELS];
#pragma unroll
    for (int k = 0; k < HM_MAX_CHANNELS; ++k) { mn[k] = 1.0 / 0.0; mx[k] = -1.0 / 0.0; }
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int c = static_cast<int>(e % C);
        const double x = val[e];
        if (!(fabs(x) <= 1.79769313486231570e308)) continue;
        if (sd && sd[e] == 0.0) continue;
#pragma unroll
        for (int k = 0; k < HM_MAX_CHANNELS; ++k) if (k == c) { mn[k] = fmin(mn[k], x); mx[k] = fmax(mx[k], x); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < HM_MAX_CHANNELS; ++k) {
        double a = mn[k], b = mx[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a = fmin(a, __shfl_down(a, off, 64)); b = fmax(b, __shfl_down(b, off, 64)); }
        if (lane == 0) { red[wave][2 * k] = a; red[wave][2 * k + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < HM_MAX_CHANNELS) {
        const int k = threadIdx.x;
        partial[(blockIdx.x * HM_MAX_CHANNELS + k) * 2] = fmin(fmin(red[0][2 * k], red[1][2 * k]), fmin(red[2][2 * k], red[3][2 * k]));
        partial[(blockIdx.x * HM_MAX_CHANNELS + k) * 2 + 1] = fmax(fmax(red[0][2 * k + 1], red[1][2 * k + 1]), fmax(red[2][2 * k + 1], red[3][2 * k + 1]));
    }
}

__global__ __launch_bounds__(64) void k_minmax_final(const double* __restrict__ partial, int nblocks, int C, double* __restrict__ out /*[C][2]*/) {
    const int k = threadIdx.x;
    if (k >= C) return;
    double a = 1.0 / 0.0, b = -1.0 / 0.0;
    for (int i = 0; i < nblocks; ++i) { a = fmin(a, partial[(i * HM_MAX_CHANNELS + k) * 2]); b = fmax(b, partial[(i * HM_MAX_CHANNELS + k) * 2 + 1]); }
    out[2 * k] = a; out[2 * k + 1] = b;
}

// apply_thresholds for more than HM_MAX_CHANNELS channels on the last axis (the reference's property tests draw up to 10):
// limits staged in LDS (uniform loop over the kernarg arrays), one element per thread.
struct ChanLimitsWide { double lo[HM_THRESHOLD_MAX_CHANNELS]; double hi[HM_THRESHOLD_MAX_CHANNELS]; };
__global__ __launch_bounds__(256) void k_thresholds_wide(double* __restrict__ val, double* __restrict__ sd, const ChanLimitsWide lim,
                                                         int64_t n, int C) {
    __shared__ double lo[HM_THRESHOLD_MAX_CHANNELS], hi[HM_THRESHOLD_MAX_CHANNELS];
    for (int k = 0; k < C; ++k)
        if (threadIdx.x == 0) { lo[k] = lim.lo[k]; hi[k] = lim.hi[k]; }
    __syncthreads();
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int c = static_cast<int>(e % C);
        const double v = val[e];
        if ((v < lo[c]) || (v > hi[c])) { val[e] = nan; if (sd) sd[e] = nan; }       // measurand.py:418
    }
}

}  // namespace hm

using namespace hm;

extern "C" int hm_apply_thresholds(double* val, double* std, const double* lower, const double* upper,
                                   int64_t n, int C, void* stream) {
    if (n < 0 || C < 1 || !lower || !upper) return HM_EINVAL;
    if (C > HM_THRESHOLD_MAX_CHANNELS) return HM_EUNSUPPORTED;
    if (n == 0) return HM_OK;
    if (!val) return HM_EINVAL;
    if (!aligned(val, 8) || (std && !aligned(std, 8))) return HM_EALIGN;
    if (C > HM_MAX_CHANNELS) {
        ChanLimitsWide w{};
        for (int c = 0; c < C; ++c) { w.lo[c] = lower[c]; w.hi[c] = upper[c]; }
        hipLaunchKernelGGL(k_thresholds_wide, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream), val, std, w, n, C);
        return launch_status();
    }
    ChanLimits lim{};
    for (int c = 0; c < C; ++c) { lim.lo[c] = lower[c]; lim.hi[c] = upper[c]; }
    hipLaunchKernelGGL(k_thresholds, dim3(stream_grid((n + 1) / 2, 256, 8)), dim3(256), 0, as_stream(stream), val, std, lim, n, C);
    return launch_status();
}

extern "C" int hm_compute_difference(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                                     double* out_abs, double* out_abs_std, double* out_rel, double* out_rel_std,
                                     int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !y || !out_abs || !out_rel) return HM_EINVAL;
    const bool with_std = sx || sy;
    if (with_std != (out_abs_std != nullptr) || with_std != (out_rel_std != nullptr)) return HM_EINVAL;
    hipLaunchKernelGGL(k_difference, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream),
                       x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, n);
    return launch_status();
}

extern "C" int hm_interpolate(const double* x0, const double* s0, const double* x1, const double* s1,
                              double y0, double y1, double y, double* out, double* out_std, int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x0 || !x1 || !out || ((s0 || s1) != (out_std != nullptr))) return HM_EINVAL;
    hipLaunchKernelGGL(k_interpolate, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream),
                       x0, s0, x1, s1, y0, y1, y, out, out_std, n);
    return launch_status();
}

extern "C" size_t hm_channel_statistics_workspace_bytes(void) {
    return sizeof(double) * (kStatBlocks * HM_MAX_CHANNELS * kStatVals + HM_MAX_CHANNELS);
}

extern "C" int hm_channel_statistics(const double* val, const double* std, int64_t n, int C,
                                     double* out /*3*C: mean, std, error*/, void* workspace, void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !val || !out || !workspace) return HM_EINVAL;
    if (!aligned(val, 8) || (std && !aligned(std, 8))) return HM_EALIGN;
    double* partial = static_cast<double*>(workspace);
    double* denom = partial + kStatBlocks * HM_MAX_CHANNELS * kStatVals;
    const int grid = stat_grid(n);
    hipStream_t st = as_stream(stream);
    const int weighted = std ? 1 : 0;
    hipLaunchKernelGGL(k_stats<1>, dim3(grid), dim3(256), 0, st, val, std, n, C, static_cast<const double*>(nullptr), partial);
    hipLaunchKernelGGL(k_stats_final<1>, dim3(1), dim3(256), 0, st, partial, grid, C, weighted, out, denom);
    hipLaunchKernelGGL(k_stats<2>, dim3(grid), dim3(256), 0, st, val, std, n, C, static_cast<const double*>(out), partial);
    hipLaunchKernelGGL(k_stats_final<2>, dim3(1), dim3(256), 0, st, partial, grid, C, weighted, out, denom);
    return launch_status();
}

extern "C" size_t hm_pair_statistics_workspace_bytes(void) {
    return sizeof(double) * (kStatBlocks * HM_MAX_CHANNELS * kPairVals + 2 * HM_MAX_CHANNELS);
}

extern "C" int hm_pair_statistics(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                                  int64_t n, int C, double* out /*6*C*/, void* workspace, void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !x || !y || !out || !workspace) return HM_EINVAL;
    double* partial = static_cast<double*>(workspace);
    double* denom = partial + kStatBlocks * HM_MAX_CHANNELS * kPairVals;
    const int grid = stat_grid(n);
    hipStream_t st = as_stream(stream);
    const int weighted = (sx || sy) ? 1 : 0;
    hipLaunchKernelGGL(k_pair_stats<1>, dim3(grid), dim3(256), 0, st, x, sx, y, sy, multiplier, n, C, static_cast<const double*>(nullptr), partial);
    hipLaunchKernelGGL(k_pair_final<1>, dim3(1), dim3(256), 0, st, partial, grid, C, weighted, out, denom);
    hipLaunchKernelGGL(k_pair_stats<2>, dim3(grid), dim3(256), 0, st, x, sx, y, sy, multiplier, n, C, static_cast<const double*>(out), partial);
    hipLaunchKernelGGL(k_pair_final<2>, dim3(1), dim3(256), 0, st, partial, grid, C, weighted, out, denom);
    return launch_status();
}

extern "C" size_t hm_histogram_workspace_bytes(int bins, int C) {
    const size_t a = sizeof(double) * static_cast<size_t>(kHistBlocks) * static_cast<size_t>(bins) * static_cast<size_t>(C);
    const size_t b = sizeof(double) * kHistBlocks * HM_MAX_CHANNELS * 2;
    return a > b ? a : b;
}

extern "C" int hm_channel_minmax(const double* val, const double* std, int64_t n, int C, double* out /*2*C*/, void* workspace,
                                 void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !val || !out || !workspace) return HM_EINVAL;
    const int grid = static_cast<int>(std::min<int64_t>(kHistBlocks, (n + 255) / 256));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_minmax, dim3(grid), dim3(256), 0, st, val, std, n, C, static_cast<double*>(workspace));
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), grid, C, out);
    return launch_status();
}

extern "C" int hm_channel_histogram(const double* val, const double* std, int64_t n, int C, int channel_mask,
                                    const double* edges /*device, bins + 1*/, int bins, double lo, double hi,
                                    double* out /*C * bins*/, void* workspace, void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || bins < 1 || !val || !edges || !out || !workspace || !(hi > lo)) return HM_EINVAL;
    if (bins * C > kHistMaxBins * 4) return HM_EUNSUPPORTED;
    const int nb = bins * C;
    const int lds = nb * static_cast<int>(sizeof(double));
    if (lds > 64 * 1024) return HM_EUNSUPPORTED;
    const int grid = static_cast<int>(std::min<int64_t>(kHistBlocks, (n + 255) / 256));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), lds, st, val, std, n, C, channel_mask, edges, bins, lo, hi, static_cast<double*>(workspace));
    hipLaunchKernelGGL(k_hist_final, dim3((nb + 255) / 256), dim3(256), 0, st, static_cast<const double*>(workspace), grid, nb, out);
    return launch_status();
}
