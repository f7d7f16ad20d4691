// hm_stats.hip - the linearity-statistics row of SURVEY.md 8(f)-1 as HIP kernels (gfx950):
//   hm_apply_thresholds      AbstractMeasurand.apply_thresholds          modules/measurand.py:375-428
//   hm_compute_difference    AbstractMeasurand.compute_difference        modules/measurand.py:620-655
//   hm_interpolate           AbstractMeasurand.interpolate               modules/measurand.py:657-681
//   hm_channel_statistics    compute_dimension_statistics(axis = all but the last)   modules/measurand.py:318-350
// Streaming, HBM-bound; the statistics are ONE deterministic reduction pass (per-thread Welford states combined with Chan's
// formula by wave shuffles + LDS, then one workgroup), NaNs ignored exactly as np.nansum / np.nanmean / np.nanstd do.
#include "hm_common.h"
#include <initializer_list>
#include <algorithm>

namespace hm {

typedef double f64x2 __attribute__((ext_vector_type(2)));
struct ChanLimits { double lo[HM_MAX_CHANNELS]; double hi[HM_MAX_CHANNELS]; };

// In place; two elements per lane (16-byte loads; a store only where something changes - thresholds usually clip a
// minority), running channel counter.
__global__ __launch_bounds__(256) void k_thresholds(double* __restrict__ val, double* __restrict__ sd, const ChanLimits lim,
                                                    int64_t n, int C) {
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

// dense, 16-byte-aligned buffers: two elements per lane, 16-byte nontemporal accesses, std presence at compile time (the
// per-element kernel above measured 0.63 of 8 TB/s on its eight streams)
template <bool SX, bool SY>
__global__ __launch_bounds__(256) void k_difference_dense(const double* __restrict__ x, const double* __restrict__ sx,
                                                          const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                          double* __restrict__ ad, double* __restrict__ ads,
                                                          double* __restrict__ rd, double* __restrict__ rds, int64_t n) {
    constexpr bool STD = SX || SY;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t pairs = n / 2;
    auto one = [&](double xv, double yv, double xs, double ys, double& a, double& r, double& as, double& rs) {
        const double scale = mult * yv;                     // :634
        a = xv - scale;                                     // :635
        r = a / scale;                                      // :636
        if constexpr (STD) {
            const double m1 = mult * ys;
            as = sqrt(xs * xs + m1 * m1);                   // :652
            const double u1 = xs / (mult * yv);
            const double u2 = (ys * xv) / (mult * (yv * yv));
            rs = sqrt(u1 * u1 + u2 * u2);                   // :653
        }
    };
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < pairs; q += stride) {
        const f64x2 xv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x) + q);
        const f64x2 yv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(y) + q);
        f64x2 xs = {0.0, 0.0}, ys = {0.0, 0.0};
        if constexpr (SX) xs = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sx) + q);
        if constexpr (SY) ys = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sy) + q);
        double a0, a1, r0, r1, as0 = 0.0, as1 = 0.0, rs0 = 0.0, rs1 = 0.0;
        one(xv.x, yv.x, xs.x, ys.x, a0, r0, as0, rs0);
        one(xv.y, yv.y, xs.y, ys.y, a1, r1, as1, rs1);
        f64x2 o;
        o.x = a0; o.y = a1; __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(ad) + q);
        o.x = r0; o.y = r1; __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(rd) + q);
        if constexpr (STD) {
            o.x = as0; o.y = as1; __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(ads) + q);
            o.x = rs0; o.y = rs1; __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(rds) + q);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1;
        double a, r, as = 0.0, rs = 0.0;
        one(x[e], y[e], SX ? sx[e] : 0.0, SY ? sy[e] : 0.0, a, r, as, rs);
        ad[e] = a; rd[e] = r;
        if constexpr (STD) { ads[e] = as; rds[e] = rs; }
    }
}

// the burst access shape (hm_ops.hip: k_binary_burst; tools/readbench.hip copysweep): a wave owns chunks of 512 elements, issues the four
// 16-byte loads per lane of every input stream back to back, then the stores of every output stream. Whole chunks only.
constexpr int kDiffBurst = 4;
constexpr int64_t kDiffChunk = 128 * kDiffBurst;
template <bool SX, bool SY>
__global__ __launch_bounds__(256) void k_difference_burst(const double* __restrict__ x, const double* __restrict__ sx,
                                                          const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                          double* __restrict__ ad, double* __restrict__ ads,
                                                          double* __restrict__ rd, double* __restrict__ rds, int64_t n_chunks) {
    constexpr bool STD = SX || SY;
    constexpr int B = kDiffBurst;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cstride = static_cast<int64_t>(gridDim.x) * 4;
    auto one = [&](double xv, double yv, double xs, double ys, double& a, double& r, double& as, double& rs) {
        const double scale = mult * yv;                     // :634
        a = xv - scale;                                     // :635
        r = a / scale;                                      // :636
        if constexpr (STD) {
            const double m1 = mult * ys;
            as = sqrt(xs * xs + m1 * m1);                   // :652
            const double u1 = xs / (mult * yv);
            const double u2 = (ys * xv) / (mult * (yv * yv));
            rs = sqrt(u1 * u1 + u2 * u2);                   // :653
        }
    };
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); c < n_chunks; c += cstride) {
        const int64_t b = c * kDiffChunk;
        f64x2 xv[B], yv[B], xs[B], ys[B];
#pragma unroll
        for (int k = 0; k < B; ++k) xv[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < B; ++k) yv[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(y + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < B; ++k) {
            xs[k] = f64x2{0.0, 0.0};
            if constexpr (SX) xs[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sx + b + 128 * k) + lane);
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            ys[k] = f64x2{0.0, 0.0};
            if constexpr (SY) ys[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sy + b + 128 * k) + lane);
        }
        f64x2 a[B], r[B], as[B], rs[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            double a0, a1, r0, r1, as0 = 0.0, as1 = 0.0, rs0 = 0.0, rs1 = 0.0;
            one(xv[k].x, yv[k].x, xs[k].x, ys[k].x, a0, r0, as0, rs0);
            one(xv[k].y, yv[k].y, xs[k].y, ys[k].y, a1, r1, as1, rs1);
            a[k] = f64x2{a0, a1}; r[k] = f64x2{r0, r1}; as[k] = f64x2{as0, as1}; rs[k] = f64x2{rs0, rs1};
        }
#pragma unroll
        for (int k = 0; k < B; ++k) __builtin_nontemporal_store(a[k], reinterpret_cast<f64x2*>(ad + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < B; ++k) __builtin_nontemporal_store(r[k], reinterpret_cast<f64x2*>(rd + b + 128 * k) + lane);
        if constexpr (STD) {
#pragma unroll
            for (int k = 0; k < B; ++k) __builtin_nontemporal_store(as[k], reinterpret_cast<f64x2*>(ads + b + 128 * k) + lane);
#pragma unroll
            for (int k = 0; k < B; ++k) __builtin_nontemporal_store(rs[k], reinterpret_cast<f64x2*>(rds + b + 128 * k) + lane);
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

// interpolate in the burst access shape (k_difference_burst): whole 512-element chunks of 16-byte aligned, dense buffers
template <bool S0, bool S1>
__global__ __launch_bounds__(256) void k_interpolate_burst(const double* __restrict__ x0, const double* __restrict__ s0,
                                                           const double* __restrict__ x1, const double* __restrict__ s1,
                                                           double y0, double y1, double y, double* __restrict__ out,
                                                           double* __restrict__ out_std, int64_t n_chunks) {
    constexpr bool STD = S0 || S1;
    constexpr int B = kDiffBurst;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cstride = static_cast<int64_t>(gridDim.x) * 4;
    const double a = y1 - y, b = y - y0, d = y1 - y0;
    const double ca = (a / d) * (a / d), cb = (b / d) * (b / d);
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); c < n_chunks; c += cstride) {
        const int64_t e = c * kDiffChunk;
        f64x2 v0[B], v1[B], u0[B], u1[B];
#pragma unroll
        for (int k = 0; k < B; ++k) v0[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x0 + e + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < B; ++k) v1[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x1 + e + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < B; ++k) {
            u0[k] = f64x2{0.0, 0.0};
            if constexpr (S0) u0[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s0 + e + 128 * k) + lane);
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            u1[k] = f64x2{0.0, 0.0};
            if constexpr (S1) u1[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s1 + e + 128 * k) + lane);
        }
        f64x2 o[B], os[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            o[k] = f64x2{(v0[k].x * a + v1[k].x * b) / d, (v0[k].y * a + v1[k].y * b) / d};             // :665
            os[k] = f64x2{sqrt(u0[k].x * ca + u1[k].x * cb), sqrt(u0[k].y * ca + u1[k].y * cb)};       // :679 as written
        }
#pragma unroll
        for (int k = 0; k < B; ++k) __builtin_nontemporal_store(o[k], reinterpret_cast<f64x2*>(out + e + 128 * k) + lane);
        if constexpr (STD) {
#pragma unroll
            for (int k = 0; k < B; ++k) __builtin_nontemporal_store(os[k], reinterpret_cast<f64x2*>(out_std + e + 128 * k) + lane);
        }
    }
}

// ---- statistics -------------------------------------------------------------------------------
// compute_dimension_statistics (modules/measurand.py:318-350) in ONE pass over the data. The reference's two NumPy passes
// (mean, then sum of squared deviations from it) would read every byte twice; here every thread keeps a running (W, mean, M2)
// of its elements (short blocks of shifted sums folded in with Chan's formula, see MomAcc) and thread, wave, workgroup and grid
// partials are combined with the same pairwise formula in a fixed tree order (bit-reproducible; as stable as the two-pass form:
// moments are always taken about a mean of the data they cover). NaNs are skipped exactly as np.nansum / np.nanmean do, term by
// term: `Wall` sums every non-NaN weight (the reference's denominators), the moments take the elements whose v * w is FINITE.
// Infinite terms (a relative difference against y = 0 is one) never enter the moments - inf - K would poison them with NaN where NumPy
// still answers: their SUM (+inf, -inf, or NaN when both signs occurred) is added into M2, which has no other meaning once a term was
// infinite, survives every fold and merge there (inf + finite = inf), and mom_finish() then gives NumPy's answers (mean = +-inf / NaN by sign,
// nanstd = NaN; the weighted formulas' inf or 0) - see there.
// workgroups of the reduction kernels: 3 per CU. A/B on one box (tools/ab_ops.sh, tools/ab_lin.sh, profiles/r02h_ab_stat_blocks.log): 2040
// -> 768 takes channel statistics from 0.65 / 0.59 to 0.73 / 0.69 of 8 TB/s and the all-pairs kernel from 1.53 / 2.91 to 1.37 / 2.77 ms
#ifndef HM_STAT_BLOCKS
#define HM_STAT_BLOCKS 768
#endif
constexpr int kStatBlocks = HM_STAT_BLOCKS;                           // a multiple of 12: total threads divisible by C = 1..4

static int stat_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    g = ((g + 11) / 12) * 12;
    return static_cast<int>(g < kStatBlocks ? g : kStatBlocks);
}

struct Mom {
    double Wall;        // sum of the non-NaN weights (unweighted: count of non-NaN values)
    double W;           // weight of the elements in (mean, M2)
    double mean, M2;    // weighted mean and sum of w (v - mean)^2 of those elements
    double ss, cs;      // sum and count of the non-NaN stds (`error` = nanmean(std)); unused without std
};
constexpr int kMomVals = 6;
__device__ __forceinline__ Mom mom_zero() { return Mom{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; }

// Per-thread accumulation, branch-free: kMomBlock (64) elements are summed as shifted moments about K = the state's current mean
// (S0 = sum w, S1 = sum w (v - K), S2 = sum w (v - K)^2: three FMAs per element, no division), an element that NumPy's nan-functions
// would skip enters with weight 0 and deviation 0 (selects, no branch), and the block is folded into the running (W, mean, M2) with
// Chan's formula when the LOOP COUNTER says so (wave-uniform) - two divisions per block instead of two per element. K tracks the data
// (until the first fold it is the first valid element), so S2 - S1^2 / S0 differences nothing large. The FIRST fold comes early - after a
// lane's first iteration (2-4 elements) - because that first element may be an outlier that barely counts (a relative difference against
// y ~ 0 with its huge std: weight ~ 0): as the shift of a whole 64-element block it cost eps (K - mean)^2 / sigma^2 = 1e-7 on the std
// (tools/fuzz_backends.py --scale 6); after two elements the weighted mean is already the ordinary one.
#ifndef HM_MOM_BLOCK
#define HM_MOM_BLOCK 64      // round 4: 8 -> 64. A fold is two reciprocals, a dozen multiplies and a branch per state; the all-pairs kernel (FP64-VALU
#endif                       // bound, two states per wave) went 2 673 -> 2 440 us with std and 1 429 -> 1 210 us without on one box (16: 2 548 / 1 305,
                             // 32: 2 485 / 1 245, 128: 2 425 / 1 190, 256: 2 407 / 1 260; tools/gpu_r04d.sh, profiles/r04d_*). K is refreshed with the
                             // running mean at every fold and starts as a data element, so 64 shifted terms difference nothing large.
constexpr int kMomBlock = HM_MOM_BLOCK;
struct MomAcc {
    Mom m;
    double K, S0, S1, S2;
    bool haveK;
    int cnt;            // acc_add_pair: count of the non-NaN stds as an integer (folded into m.cs by acc_finish)
};
__device__ __forceinline__ MomAcc acc_zero() { return MomAcc{mom_zero(), 0.0, 0.0, 0.0, 0.0, false, 0}; }
// the shift K of a block must be finite (inf - K has to keep its sign): an infinite first element gives +-1e300, and the block it poisons
// (S1 = +-inf / NaN) is then set aside by acc_fold()
__device__ __forceinline__ double finite_shift(double v) { return fmin(fmax(v, -1e300), 1e300); }
__device__ __forceinline__ bool is_finite(double x) { return fabs(x) < __longlong_as_double(0x7ff0000000000000ll); }   // false for NaN and +-inf

__device__ __forceinline__ double rcp_nr(double x);
// RCP: the two quotients of a fold as products with rcp_nr() (~1 ulp; 14 instructions instead of two IEEE division expansions) - the
// pair kernels, which are FP64-VALU bound and fold two states per 8 element-pairs
template <bool RCP = false>
__device__ __forceinline__ void acc_fold(MomAcc& a) {
    if (!a.haveK && a.S0 != 0.0) { a.K = a.S2; a.S2 = 0.0; }        // a lone parked element (weighted_first below): it is its own shift, d = 0
    if (!is_finite(a.S1)) {                                         // acc_add_pair let infinite terms in: their sum's class goes to M2, the block is not folded
        a.m.M2 += a.S1;
        a.K = 0.0; a.haveK = false;                                 // (K may be the clamped +-1e300 of an infinite first element: take a new one)
    } else if (a.S0 != 0.0) {                                       // (a lane whose whole block was skipped: rare)
        const double q = RCP ? a.S1 * rcp_nr(a.S0) : a.S1 / a.S0;   // block mean - K
        const double mb = a.K + q;
        const double M2b = a.S2 - a.S1 * q;
        if (a.m.W == 0.0) { a.m.W = a.S0; a.m.mean = mb; a.m.M2 += M2b; }      // (+= : M2 is 0 here - or already holds infinite terms)
        else {
            const double W = a.m.W + a.S0;
            const double d = mb - a.m.mean;
            const double f = RCP ? a.S0 * rcp_nr(W) : a.S0 / W;
            a.m.mean = a.m.mean + d * f;
            a.m.M2 = (a.m.M2 + M2b) + (d * d) * (a.m.W * f);
            a.m.W = W;
        }
        a.K = a.m.mean;
    }
    a.S0 = 0.0; a.S1 = 0.0; a.S2 = 0.0;
}

// WEIGHTED statistics, the first elements of a lane (no shift yet): the shift must not be an element that barely counts. A relative
// difference against y ~ 0 is huge AND has a huge std, i.e. a weight ~ 0: as the shift K of a block it makes every ordinary element
// contribute w (v - K)^2 ~ 1e14 to S2, and M2 = S2 - S1^2 / S0 then carries eps x that - 2.8e-6 on a std in one case of
// tools/fuzz_backends.py (seed 1760803113997506811: K = 1.4e6 with weight 4e-12 beside values of 0.05 with weight 54). So the first element
// is PARKED - its weight in S0, its value in S2, nothing accumulated - and when the second arrives the heavier of the two becomes K and both
// are accumulated about it (exactly: one of the two deviations is 0). Elements of weight 0 add nothing to the moments and are passed over.
// Returns true when the element has been dealt with here. (acc_fold turns a parked element that stayed alone into a block of its own.)
// Used by the HBM-bound kernels (acc_add: channel and axis statistics); the FP64-bound pair kernels keep the first element as the shift
// (see acc_add_pair) and with it the limit on unthresholded heavy-tailed data described in DESIGN.md 4.4.
__device__ __forceinline__ bool weighted_first(MomAcc& a, double v, double w) {
    if (w == 0.0) return true;
    if (a.S0 == 0.0) { a.S0 = w; a.S2 = v; return true; }          // park
    const double vp = a.S2, wp = a.S0;
    a.K = w > wp ? v : vp;
    const double dp = vp - a.K, dn = v - a.K;
    const double tp = wp * dp, tn = w * dn;
    a.S0 = wp + w; a.S1 = tp + tn; a.S2 = fma(tn, dn, tp * dp);
    a.haveK = true;
    return true;
}

// one element: value v, weight w (already formed; 1 when unweighted), std s for the `error` sum; in_range = the element exists (tail lanes)
__device__ __forceinline__ void acc_add(MomAcc& a, double v, double w, double s, bool weighted, bool in_range) {
    bool use;
    if (weighted) {
        const bool okw = in_range && (w == w);
        const bool oks = in_range && (s == s);
        a.m.Wall += okw ? w : 0.0;                                          // nansum(weights)
        a.m.ss += oks ? s : 0.0;                                            // nanmean(stds)
        a.m.cs += oks ? 1.0 : 0.0;
        const double vw = v * w;
        use = in_range && is_finite(vw);                                    // nansum(values * weights), nansum(weights * (values - mean)**2)
        a.m.M2 += (in_range && !use && vw == vw) ? vw : 0.0;                // an infinite term
    } else {
        use = in_range && is_finite(v);
        a.m.M2 += (in_range && !use && v == v) ? v : 0.0;
        w = 1.0;
    }
    if (weighted && use && !a.haveK) { weighted_first(a, v, w); return; }   // (rare: a lane's first two elements; v is finite here - v w is)
    a.K = (!a.haveK && use) ? v : a.K;
    a.haveK = a.haveK || use;
    const double we = use ? w : 0.0;
    const double d = use ? v - a.K : 0.0;
    const double t = we * d;
    a.S0 += we; a.S1 += t; a.S2 = fma(t, d, a.S2);
}

template <bool RCP = false>
__device__ __forceinline__ Mom acc_finish(MomAcc& a, bool weighted) {
    acc_fold<RCP>(a);
    if (!weighted) a.m.Wall = a.m.W;
    a.m.cs += static_cast<double>(a.cnt);
    return a.m;
}

// acc_add for the pair kernels. Weighted: w = 1 / sqrt(q) >= 0 and s = sqrt(q) >= 0 are NaN together (q NaN), so max(., 0) replaces
// the NaN selects of nansum(weights) / nansum(stds) and ONE comparison counts the non-NaN stds in an integer. Unweighted: w = 1.
// Elements that do not exist (tail lanes) must arrive as NaNs (value, and std when weighted).
// max(x, 0) with NaN -> 0 as ONE v_max_f64: fmax() puts a canonicalising v_max_f64 x, x in front because it cannot rule out a
// signalling NaN; x is the result of arithmetic here, never one
__device__ __forceinline__ double max0_nan_to_zero(double x) {
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
// LEAN: every lane of the wave already has its shift K (haveK set), so the two selects that look for the first valid element are
// dead - the caller checks that with one ballot per iteration (pair_process_any). Same bits either way.
#ifndef HM_PAIR_EXEC
#define HM_PAIR_EXEC 1       // round 4: all-pairs statistics 2 456-2 469 -> 2 337-2 341 us with std, 1 208-1 213 -> 1 154-1 161 us without (one box,
#endif                       // two runs each, profiles/r04k_pair_exec_ab.log); 0 = the select form
template <bool WEIGHTED, bool LEAN = false>
__device__ __forceinline__ void acc_add_pair(MomAcc& a, double v, double w, double s) {
    bool use;
    if constexpr (WEIGHTED) {
        a.m.Wall += max0_nan_to_zero(w);
        a.m.ss += max0_nan_to_zero(s);
        a.cnt += (s == s) ? 1 : 0;
        const double vw = v * w;
        use = vw == vw;
    } else {
        use = v == v;
    }
    // Infinite terms (unweighted: a relative difference against y = 0) are NOT filtered per element here - these kernels are FP64-issue
    // bound and a second branch per state and element cost 11-21 %, a finiteness test in the first-element path 4 % (profiles/r04n_*,
    // r04o_*): they enter the block sums (K stays finite - finite_shift() - so S1 becomes +inf / -inf / NaN by their signs) and
    // acc_fold() moves a non-finite S1 into M2 instead of folding the block. The finite
    // elements of such a block (at most 63 per lane) are dropped: the mean and the unweighted std are NumPy's regardless; the weighted
    // std's "inf or 0" (mom_finish) then rests on the finite elements of all other blocks.
    if constexpr (HM_PAIR_EXEC != 0) {
        // EXEC-masked form: the lanes whose element NumPy's nan-functions skip sit out the accumulation (a real branch - the empty volatile asm
        // keeps the compiler from turning it back into selects, which cost 4-6 of the 19 VALU instructions of a state and element; the branch
        // itself is scalar-unit work). Same arithmetic on the participating lanes: the same bits.
        if (use) {
            asm volatile("" ::: "memory");
            if constexpr (!LEAN) {
                if (!a.haveK) {                       // (elements of weight 0 - std = inf - have added nothing: K is re-taken until one counts)
                    asm volatile("" ::: "memory");
                    a.K = finite_shift(v);            // (weighted_first() here costs the std kernel 12 bytes of scratch and 11 %: 2 410 -> 2 670 us)
                    a.haveK = WEIGHTED ? (w != 0.0) : true;
                }
            }
            const double d = v - a.K;
            if constexpr (WEIGHTED) {
                const double t = w * d;
                a.S0 += w; a.S1 += t; a.S2 = fma(t, d, a.S2);
            } else {
                a.S0 += 1.0; a.S1 += d; a.S2 = fma(d, d, a.S2);
            }
        }
        return;
    }
    if constexpr (!LEAN) {
        a.K = (!a.haveK && use) ? finite_shift(v) : a.K;
        a.haveK = a.haveK || (use && (!WEIGHTED || w != 0.0));
    }
    const double d = use ? v - a.K : 0.0;
    if constexpr (WEIGHTED) {
        const double we = use ? w : 0.0;
        const double t = we * d;
        a.S0 += we; a.S1 += t; a.S2 = fma(t, d, a.S2);
    } else {
        a.S0 += use ? 1.0 : 0.0; a.S1 += d; a.S2 = fma(d, d, a.S2);
    }
}

// Chan / Golub / LeVeque pairwise combination; an empty side (W == 0) leaves the other's moments untouched
__device__ __forceinline__ Mom mom_merge(const Mom& a, const Mom& b) {
    Mom r;
    r.Wall = a.Wall + b.Wall; r.ss = a.ss + b.ss; r.cs = a.cs + b.cs;
    if (b.W == 0.0) { r.W = a.W; r.mean = a.mean; r.M2 = a.M2 + b.M2; return r; }      // (an empty side's M2 is 0 - or the sum of its infinite terms)
    if (a.W == 0.0) { r.W = b.W; r.mean = b.mean; r.M2 = b.M2 + a.M2; return r; }
    const double W = a.W + b.W;
    const double d = b.mean - a.mean;
    const double f = b.W / W;
    r.W = W;
    r.mean = a.mean + d * f;
    r.M2 = (a.M2 + b.M2) + (d * d) * (a.W * f);
    return r;
}

__device__ __forceinline__ Mom mom_shfl_down(const Mom& m, int off) {
    Mom r;
    r.Wall = __shfl_down(m.Wall, off, 64); r.W = __shfl_down(m.W, off, 64); r.mean = __shfl_down(m.mean, off, 64);
    r.M2 = __shfl_down(m.M2, off, 64); r.ss = __shfl_down(m.ss, off, 64); r.cs = __shfl_down(m.cs, off, 64);
    return r;
}
__device__ __forceinline__ void mom_store(double* p, const Mom& m) { p[0] = m.Wall; p[1] = m.W; p[2] = m.mean; p[3] = m.M2; p[4] = m.ss; p[5] = m.cs; }
__device__ __forceinline__ Mom mom_load(const double* p) { return Mom{p[0], p[1], p[2], p[3], p[4], p[5]}; }

// workgroup-level combination of one state per thread, where thread t's state belongs to channel `ct`:
// per channel a shuffle tree inside each wave (lanes of other channels contribute the empty state), then the four waves in order.
// Result for channel c is written to partial[(block * HM_MAX_CHANNELS + c) * nq + q] by thread c (q = quantity index).
template <int NQ>
__device__ __forceinline__ void block_merge_store(const Mom (&mine)[NQ], int ct, double* partial) {
    __shared__ double red[4][HM_MAX_CHANNELS][NQ][kMomVals];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            Mom m = c == ct ? mine[q] : mom_zero();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = mom_merge(m, mom_shfl_down(m, off));
            if (lane == 0) mom_store(red[wave][c][q], m);
        }
    }
    __syncthreads();
    if (threadIdx.x < HM_MAX_CHANNELS * NQ) {
        const int c = threadIdx.x / NQ, q = threadIdx.x % NQ;
        Mom m = mom_load(red[0][c][q]);
        for (int w = 1; w < 4; ++w) m = mom_merge(m, mom_load(red[w][c][q]));
        mom_store(partial + ((static_cast<int64_t>(blockIdx.x) * HM_MAX_CHANNELS + c) * NQ + q) * kMomVals, m);
    }
}

// grid-level combination (one workgroup): thread t folds blocks t, t + 256, ... in order, then a fixed LDS tree over the 256 threads
template <int NQ>
__device__ __forceinline__ Mom grid_merge(const double* __restrict__ partial, int nblocks, int c, int q) {
    __shared__ double tree[256][kMomVals];
    Mom m = mom_zero();
    for (int b = threadIdx.x; b < nblocks; b += 256)
        m = mom_merge(m, mom_load(partial + ((static_cast<int64_t>(b) * HM_MAX_CHANNELS + c) * NQ + q) * kMomVals));
    __syncthreads();                                        // (the previous (c, q) round is done with `tree`)
    mom_store(tree[threadIdx.x], m);
    __syncthreads();
    for (int span = 128; span > 0; span >>= 1) {
        if (static_cast<int>(threadIdx.x) < span) {
            const Mom r = mom_merge(mom_load(tree[threadIdx.x]), mom_load(tree[threadIdx.x + span]));
            mom_store(tree[threadIdx.x], r);
        }
        __syncthreads();
    }
    return mom_load(tree[0]);
}

// mean / std / error of modules/measurand.py:339-349 from the combined state:
//   mean = nansum(v w) / nansum(w) = W mean_W / Wall;  std = sqrt(nansum(w (v - mean)^2) / nansum(w));  error = nanmean(std)
__device__ __forceinline__ void mom_finish(const Mom& m, bool weighted, double& mean, double& sd, double& err) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (!is_finite(m.M2)) {
        // a term v * w was infinite; m.M2 is the sum of those terms (+inf, -inf, NaN for both signs). What NumPy then computes:
        //   nanmean / nansum(v w) / nansum(w): the infinite sum over the count or the weights;  nanstd: (v - mean) is inf - inf = NaN for the
        //   infinite element and the reduction there is a plain sum -> NaN;  weighted: nansum(w (v - mean)^2) SKIPS the NaN terms - with
        //   mean = +-inf every finite element with w > 0 contributes inf (sum inf, or 0 when there is none), with mean = NaN every term is
        //   skipped (sum 0) - over nansum(w).
        mean = m.M2 / m.Wall;
        if (!weighted) sd = nan;
        else sd = sqrt(((mean == mean && m.W > 0.0) ? __longlong_as_double(0x7ff0000000000000ll) : 0.0) / m.Wall);
        err = weighted ? m.ss / m.cs : nan;
        return;
    }
    const double M2 = fmax(m.M2, 0.0);                      // a sum of squares: rounding in S2 - S1^2 / S0 may leave -1e-13 where the data have no spread
    if (m.Wall == m.W) { mean = m.W == 0.0 ? nan : m.mean; sd = sqrt(M2 / m.Wall); }
    else {                                                  // some weights belong to NaN values: the reference's denominators still count them
        mean = (m.W * m.mean) / m.Wall;
        const double d = m.mean - mean;
        sd = sqrt((M2 + m.W * (d * d)) / m.Wall);
    }
    err = weighted ? m.ss / m.cs : nan;
}

// 8-byte load at (wave-uniform base) + (32-bit byte offset in a VGPR): one 64-bit add per load, no per-lane index arithmetic
template <bool NT>
__device__ __forceinline__ double ld_f64_sv(const double* sbase, uint32_t byte_off) {
    typedef const __attribute__((address_space(1))) double* gdp;
    typedef const __attribute__((address_space(1))) char* gcp;
    gdp p = (gdp)((gcp)sbase + byte_off);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// 1 / s as the IEEE division gives it, without the division's operand scaling and fix-up instructions: the estimate, two Newton
// steps and Markstein's correction - instruction for instruction what hipcc emits for 1.0 / s minus v_div_scale_f64 (x 2) and
// v_div_fixup_f64, which are the identity for 2^-500 <= |s| <= 2^500 (hm_merge.hip: div_inrange). `special` reports a lane outside
// that range (zero, denormal, huge, infinite; a NaN needs no help); the caller then divides properly for the whole wave.
__device__ __forceinline__ double recip_inrange(double s, bool& special) {
    const double as = fabs(s);
    special = as < 0x1p-500 || as > 0x1p500;
    double r = __builtin_amdgcn_rcp(s);
    double e = fma(-s, r, 1.0);
    r = fma(r, e, r);
    e = fma(-s, r, 1.0);
    r = fma(r, e, r);
    const double q = 1.0 * r;
    const double res = fma(-s, q, 1.0);
    return fma(res, r, q);
}

// The launch uses a total thread count that is a multiple of C (grid rounded to a multiple of 12 workgroups), so a lane's
// elements all have the same channel ct = first_element % C. A wave walks over 64-element chunks (sb, sb + stride, ...: four per
// iteration), whole chunks on a select-free path with the next iteration's loads issued before the current one's arithmetic (two
// register sets, unconditional prefetch: see pair_loop), the last partial ones with their missing elements at weight 0.
#ifndef HM_STATS_UN_NOSTD
#define HM_STATS_UN_NOSTD 8
#endif
template <bool WEIGHTED>
__device__ __forceinline__ Mom stats_loop(const double* val, const double* sd, int64_t n, int64_t sb0, int64_t stride, uint32_t lane) {
    constexpr int UN = WEIGHTED ? 4 : HM_STATS_UN_NOSTD;     // 64-element chunks per iteration (one stream: more of them in flight; same fold points)
    MomAcc st = acc_zero();
    const uint32_t lo = lane * 8u;
    int it = 0;
    int64_t sb = sb0;
    auto add = [&](double v, double s, bool ok) {
        double w = 1.0;
        if constexpr (WEIGHTED) {                           // w = 1 / std, measurand.py:342
            bool special;
            w = recip_inrange(s, special);
            if (__builtin_amdgcn_ballot_w64(special) != 0) w = 1.0 / s;
        }
        acc_add(st, v, w, s, WEIGHTED, ok);
    };
    auto fold = [&]() { if ((it & (kMomBlock / UN - 1)) == kMomBlock / UN - 1 || it == 0) acc_fold<false>(st); };   // (it == 0: early_fold, see MomAcc)
    auto whole = [&](int64_t b) { return b + (UN - 1) * stride + 64 <= n; };
    auto load = [&](int64_t b, double (&vv)[UN], double (&sv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t cb = b + u * stride;                                   // scalar
            vv[u] = ld_f64_sv<true>(val + cb, lo);
            sv[u] = WEIGHTED ? ld_f64_sv<true>(sd + cb, lo) : 1.0;
        }
    };
    auto process = [&](const double (&vv)[UN], const double (&sv)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) add(vv[u], sv[u], true);
        fold();
    };
    const int64_t step = UN * stride;
    double va[UN], sa[UN], vb[UN], sbv[UN];
    if (whole(sb)) load(sb, va, sa);
    while (whole(sb)) {
        load(whole(sb + step) ? sb + step : sb, vb, sbv);
        process(va, sa);
        ++it; sb += step;
        if (!whole(sb)) break;
        load(whole(sb + step) ? sb + step : sb, va, sa);
        process(vb, sbv);
        ++it; sb += step;
    }
    for (; sb < n; sb += step, ++it) {                                           // the wave's last, partial chunks
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t q = sb + u * stride + lane;
            const bool ok = q < n;
            const int64_t qc = ok ? q : n - 1;                                   // a valid address, weight 0
            add(val[qc], WEIGHTED ? sd[qc] : 1.0, ok);
        }
        fold();
    }
    return acc_finish<false>(st, WEIGHTED);
}

template <bool WEIGHTED>          // one kernel per case: each gets the registers (occupancy) and the code size of its own loop
__global__ __launch_bounds__(256) void k_stats(const double* __restrict__ val, const double* __restrict__ sd, int64_t n, int C,
                                               double* __restrict__ partial) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t sb0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    const int ct = static_cast<int>((sb0 + lane) % C);
    const Mom mine[1] = {stats_loop<WEIGHTED>(val, sd, n, sb0, stride, lane)};
    block_merge_store<1>(mine, ct, partial);
}

__global__ __launch_bounds__(256) void k_stats_final(const double* __restrict__ partial, int nblocks, int C, int weighted,
                                                     double* __restrict__ out) {
    const int c = blockIdx.x;                               // one workgroup per channel (they ran one after the other in one workgroup: 19 us)
    const Mom m = grid_merge<1>(partial, nblocks, c, 0);
    if (threadIdx.x == 0) {
        double mean, sdv, err;
        mom_finish(m, weighted != 0, mean, sdv, err);
        out[c] = mean; out[C + c] = sdv; out[2 * C + c] = err;
    }
}

// ---- pair-fused linearity statistics ----------------------------------------------------------
// ExposurePair.compute_difference + compute_stats(axis=(0,1)) (modules/exposure_series.py:33-54, called per pair
// by process_linearity :443-446) without materialising the two difference images and in one pass: x, y (and their stds) are read
// once and both the absolute and the relative difference are reduced. out: 6*C doubles
// [abs mean | abs std | abs error | rel mean | rel std | rel error].
// 1 / x and 1 / sqrt(q) to within ~1 ulp without the IEEE expansions and without branches: the hardware estimate (v_rcp_f64 /
// v_rsq_f64, about 26 bits; already the IEEE answer for 0, infinities and NaN) and two Newton steps, kept only when the estimate is a
// finite non-zero number - 7 / 10 instructions instead of 11 (divide) / ~26 (sqrt + divide).
__device__ __forceinline__ bool finite_nonzero(double x) { const double ax = fabs(x); return ax > 0.0 && ax < __builtin_huge_val(); }
// STEPS == 2 (the default) is ONE higher-order step from the 2^-24 hardware estimate instead of two Newton steps - the same or better
// accuracy in fewer instructions (checked on 2 M operands against long-double references: reciprocal 1 + e + e^2, e = 1 - x r0: 3 instead
// of 4 instructions, max 1.00 ulp either way; reciprocal square root y0 (1 + e/2 + 3 e^2 / 8), e = 1 - q y0^2: 5 instead of 7 instructions,
// max 1.25 ulp against 2.19 ulp). Round 3: -5 of 104 VALU instructions per element-pair of the weighted pair statistics.
template <int STEPS = 2>
__device__ __forceinline__ double rcp_newton(double x, double r0) {
    const double e = fma(-x, r0, 1.0);
    if constexpr (STEPS == 1) return fma(e, r0, r0);
    return fma(r0, fma(e, e, e), r0);
}
template <int STEPS = 2>
__device__ __forceinline__ double rsq_newton(double q, double y0) {
    if constexpr (STEPS == 1) {
        const double h = 0.5 * q;
        return y0 * fma(-h * y0, y0, 1.5);
    }
    const double e = fma(-(q * y0), y0, 1.0);
    return fma(y0, fma(0.375, e, 0.5) * e, y0);
}
#ifndef HM_PAIR_NEWTON
#define HM_PAIR_NEWTON 2
#endif
__device__ __forceinline__ double rcp_nr(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double r = rcp_newton(x, r0);
    return finite_nonzero(r0) ? r : r0;
}
__device__ __forceinline__ double rsqrt_nr(double q, double& root) {       // also sqrt(q) = q * rsqrt(q)
    const double y0 = __builtin_amdgcn_rsq(q);
    const double y = rsq_newton(q, y0);
    const bool ok = finite_nonzero(y0);
    root = ok ? q * y : q;                                                   // sqrt(0) = 0, sqrt(inf) = inf, NaN stays NaN (q is a sum of squares)
    return ok ? y : y0;
}

// difference terms of one element (measurand.py:634-653) and the statistics weights 1 / std of both differences. One reciprocal
// (1 / scale) and two reciprocal square roots per element: x / (m y) = x r, (ys x) / (m y^2) = ys x m r^2 with r = 1 / (m y), and
// w = 1 / sqrt(q), std = q w - a few ulp from the reference's three divisions, two square roots and two reciprocals (the pair
// kernels are FP64-VALU bound); the statistics agree with the oracle to 1e-12.
// GUARD = false leaves out the selects that keep the hardware estimate where it is 0 or infinite (scale, q = 0 or inf) and reports in
// `special` whether this lane had such an estimate: the caller redoes the element with GUARD = true when any lane of the wave did
// (wave-uniform branch; both forms give the same bits everywhere else, NaNs propagate through the Newton steps by themselves).
constexpr int kClassZeroInf = 0x264;                    // v_cmp_class_f64 mask: -inf | -0 | +0 | +inf
template <bool STD, bool GUARD>
__device__ __forceinline__ void pair_terms(double xv, double xs, double yv, double ys, double mult,
                                           double& a, double& as, double& wa, double& r, double& rs, double& wr, bool& special) {
    const double scale = mult * yv;                     // measurand.py:634
    a = xv - scale;                                     // :635
    double inv;
    if constexpr (GUARD) inv = rcp_nr(scale);
    else {
        const double r0 = __builtin_amdgcn_rcp(scale);
        inv = rcp_newton<HM_PAIR_NEWTON>(scale, r0);
        special = __builtin_amdgcn_class(r0, kClassZeroInf);
    }
    r = a * inv;                                        // :636
    wa = 1.0; wr = 1.0;
    if constexpr (STD) {
        const double m1 = mult * ys;
        const double qa = xs * xs + m1 * m1;            // :652
        const double u1 = xs * inv;
        const double u2 = ((ys * xv) * mult) * (inv * inv);
        const double qr = u1 * u1 + u2 * u2;            // :653
        if constexpr (GUARD) {
            wa = rsqrt_nr(qa, as);
            wr = rsqrt_nr(qr, rs);
        } else {
            const double ya0 = __builtin_amdgcn_rsq(qa), yr0 = __builtin_amdgcn_rsq(qr);
            wa = rsq_newton<HM_PAIR_NEWTON>(qa, ya0); as = qa * wa;
            wr = rsq_newton<HM_PAIR_NEWTON>(qr, yr0); rs = qr * wr;
            special = special || __builtin_amdgcn_class(ya0, kClassZeroInf) || __builtin_amdgcn_class(yr0, kClassZeroInf);
        }
    }
}

// The element loop of both pair kernels for ONE wave: chunks of 64 consecutive elements starting at sb0, sb0 + stride, ... (sb0 and
// stride wave-uniform, stride a multiple of C so that a lane's elements share a channel). Whole chunks take the select-free path;
// the last, partial ones hand their missing elements to the accumulators as NaNs. Both Welford states fold every kMomBlock elements.
// SX / SY: the first / second operand has a std image (compile-time, so that no load sits behind a branch: the waits in front of an
// iteration's arithmetic can then leave the NEXT iteration's loads in flight); statistics are weighted when either has one.
constexpr int kPairUN = 2;                     // 64-element chunks per wave iteration (sb and sb + stride)
template <bool STD, int UN = kPairUN, bool LEAN = false>
__device__ __forceinline__ void pair_process(MomAcc (&st)[2], int it, double mult, const double (&xv)[UN], const double (&xs)[UN],
                                             const double (&yv)[UN], const double (&ys)[UN]) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        double av, as = 0.0, wa, rv, rs = 0.0, wr;
        bool special = false;
        pair_terms<STD, false>(xv[u], xs[u], yv[u], ys[u], mult, av, as, wa, rv, rs, wr, special);
        if (__builtin_amdgcn_ballot_w64(special) != 0)                           // a zero / infinite scale or variance somewhere in the wave
            pair_terms<STD, true>(xv[u], xs[u], yv[u], ys[u], mult, av, as, wa, rv, rs, wr, special);
        acc_add_pair<STD, LEAN>(st[0], av, wa, as);
        acc_add_pair<STD, LEAN>(st[1], rv, wr, rs);
    }
    // every kMomBlock elements, whatever UN; weighted: + the early fold (unweighted an outlier counts in full and sets the scale of the
    // variance itself - and the extra condition cost the unweighted all-pairs kernel 3.5 %: 1 160 -> 1 203 us)
    if ((it & (kMomBlock / UN - 1)) == kMomBlock / UN - 1 || (STD && it == 0)) { acc_fold<true>(st[0]); acc_fold<true>(st[1]); }
}

// whole chunks only (all 64 lanes active): the lean body once every lane of the wave has seen a valid element of both differences
// (normally from the first iteration on) - 8 of ~106 VALU instructions per two element-pairs less in a kernel that is VALU-bound
#ifndef HM_PAIR_LEAN
#define HM_PAIR_LEAN 1
#endif
template <bool STD, int UN = kPairUN>
__device__ __forceinline__ void pair_process_any(MomAcc (&st)[2], int it, double mult, const double (&xv)[UN], const double (&xs)[UN],
                                                 const double (&yv)[UN], const double (&ys)[UN]) {
    // (without std only: with std the all-pairs kernel sits at the 128 VGPRs a 1024-thread workgroup may use and two bodies make it spill)
#ifndef HM_PAIR_LEAN_STD
#define HM_PAIR_LEAN_STD 0
#endif
    if constexpr (HM_PAIR_LEAN && (!STD || HM_PAIR_LEAN_STD)) {
        if (__builtin_amdgcn_ballot_w64(st[0].haveK && st[1].haveK) == ~0ull) { pair_process<STD, UN, true>(st, it, mult, xv, xs, yv, ys); return; }
    }
    pair_process<STD, UN, false>(st, it, mult, xv, xs, yv, ys);
}

// the wave's last, partial chunks (from iteration `it` at element sb on), then the final fold
template <bool SX, bool SY, int UN = kPairUN>
__device__ __forceinline__ void pair_tail(MomAcc (&st)[2], int it, int64_t sb, const double* x, const double* sx, const double* y,
                                          const double* sy, double mult, int64_t n, int64_t stride, uint32_t lane, Mom (&mine)[2]) {
    constexpr bool STD = SX || SY;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (; sb < n; sb += UN * stride, ++it) {
        double xv[UN], yv[UN], xs[UN], ys[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t q = sb + u * stride + lane;
            const bool ok = q < n;
            const int64_t qc = ok ? q : n - 1;                                   // a valid address; the element enters as NaN
            xv[u] = ok ? x[qc] : nan;
            yv[u] = y[qc];
            xs[u] = SX ? sx[qc] : 0.0;
            ys[u] = SY ? sy[qc] : 0.0;
            if (STD) xs[u] = ok ? xs[u] : nan;
        }
        pair_process<STD, UN>(st, it, mult, xv, xs, yv, ys);
    }
    mine[0] = acc_finish<true>(st[0], STD);
    mine[1] = acc_finish<true>(st[1], STD);
}

// UN = 64-element chunks per wave iteration. Without stds a lane has only two 8-byte loads per chunk in flight: the per-pair kernel then
// takes four chunks per iteration (same element order per lane, same fold points: the same bits as UN = 2)
template <bool SX, bool SY, bool NT, int UN = kPairUN>
__device__ __forceinline__ void pair_loop(const double* x, const double* sx, const double* y, const double* sy, double mult, int64_t n,
                                          int64_t sb0, int64_t stride, uint32_t lane, Mom (&mine)[2]) {
    constexpr bool STD = SX || SY;
    MomAcc st[2] = {acc_zero(), acc_zero()};                                   // absolute | relative difference
    const uint32_t lo = lane * 8u;
    int it = 0;
    int64_t sb = sb0;
    // whole chunks (every lane exists): the next iteration's loads are issued before the current one's arithmetic (two register sets,
    // the loop written out twice) - a wave that waits for its loads at the top of every iteration leaves the VALU idle for the latency
    auto whole = [&](int64_t b) { return b + (UN - 1) * stride + 64 <= n; };
    auto load = [&](int64_t b, double (&xv)[UN], double (&xs)[UN], double (&yv)[UN], double (&ys)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t cb = b + u * stride;                                   // scalar
            xv[u] = ld_f64_sv<NT>(x + cb, lo);
            yv[u] = ld_f64_sv<NT>(y + cb, lo);
            xs[u] = SX ? ld_f64_sv<NT>(sx + cb, lo) : 0.0;
            ys[u] = SY ? ld_f64_sv<NT>(sy + cb, lo) : 0.0;
        }
    };
    const int64_t step = UN * stride;
    double xa[UN], ya[UN], sxa[UN], sya[UN], xb[UN], yb[UN], sxb[UN], syb[UN];
    // (the prefetch is unconditional - past the last whole iteration it re-reads the current one and the values are dropped: a branch
    // around the loads would make the compiler's wait counts assume they were skipped and wait for them before the arithmetic)
    if (whole(sb)) load(sb, xa, sxa, ya, sya);
    while (whole(sb)) {
        load(whole(sb + step) ? sb + step : sb, xb, sxb, yb, syb);
        pair_process_any<STD, UN>(st, it, mult, xa, sxa, ya, sya);
        ++it; sb += step;
        if (!whole(sb)) break;
        load(whole(sb + step) ? sb + step : sb, xa, sxa, ya, sya);
        pair_process_any<STD, UN>(st, it, mult, xb, sxb, yb, syb);
        ++it; sb += step;
    }
    pair_tail<SX, SY, UN>(st, it, sb, x, sx, y, sy, mult, n, stride, lane, mine);
}

template <bool SX, bool SY>
__global__ __launch_bounds__(256) void k_pair_stats(const double* __restrict__ x, const double* __restrict__ sx,
                                                    const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                    int64_t n, int C, double* __restrict__ partial) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;       // a multiple of C (see k_stats)
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t sb0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    const int ct = static_cast<int>((sb0 + lane) % C);
    Mom mine[2];
#ifndef HM_PAIR_UN_NOSTD
#define HM_PAIR_UN_NOSTD 4
#endif
    pair_loop<SX, SY, true, (SX || SY) ? kPairUN : HM_PAIR_UN_NOSTD>(x, sx, y, sy, mult, n, sb0, stride, lane, mine);
    block_merge_store<2>(mine, ct, partial);
}

__global__ __launch_bounds__(256) void k_pair_final(const double* __restrict__ partial, int nblocks, int C, int weighted,
                                                    double* __restrict__ out) {
    const int h = blockIdx.x / C, c = blockIdx.x % C;       // one workgroup per (difference kind, channel)
    const Mom m = grid_merge<2>(partial, nblocks, c, h);
    if (threadIdx.x == 0) {
        double mean, sdv, err;
        mom_finish(m, weighted != 0, mean, sdv, err);
        double* o = out + 3 * C * h;
        o[c] = mean; o[C + c] = sdv; o[2 * C + c] = err;
    }
}

// ---- all exposure pairs of a stack in one launch ------------------------------------------------
// ExposureSeries.process_linearity (modules/exposure_series.py:421-446) evaluates compute_difference + compute_stats for every
// exposure pair (i, j) of the series: N (N - 1) / 2 pairs, each reading two frames (+ stds). Pair by pair every frame is read
// N - 1 times from HBM. Here a workgroup owns a run of elements and each of its waves owns ONE pair: the waves of a workgroup
// read the same 64-element chunks of the frames their pairs need at about the same time, so a chunk comes from HBM once and
// from the CU's L1 / the XCD's L2 for the other waves; every wave keeps the Welford states of its pair (absolute and relative
// difference) and its partial goes to partial[block][pair]. Up to HM_PAIRS_MAX pairs per launch (1024-thread workgroups).
struct PairsK {
    const double* val[HM_MAX_FRAMES];
    const double* sd[HM_MAX_FRAMES];
    int32_t pi[HM_PAIRS_MAX], pj[HM_PAIRS_MAX];
    double mult[HM_PAIRS_MAX];
    int64_t n;
    int32_t C, n_pairs, with_std;
    double lo[HM_MAX_CHANNELS], hi[HM_MAX_CHANNELS];        // apply_thresholds limits per channel (THR instantiations of the LDS kernel)
};

__global__ __launch_bounds__(1024) void k_pairs_stats(const PairsK a, double* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));       // = pair index, wave-uniform
    const double* x = a.val[a.pi[wave]];
    const double* y = a.val[a.pj[wave]];
    const double* sx = a.with_std ? a.sd[a.pi[wave]] : nullptr;
    const double* sy = a.with_std ? a.sd[a.pj[wave]] : nullptr;
    const double mult = a.mult[wave];
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 64;                               // a multiple of C (grid: multiple of 12)
    const int64_t sb0 = static_cast<int64_t>(blockIdx.x) * 64;
    const int ct = static_cast<int>((sb0 + lane) % a.C);
    Mom mine[2];
    // (default cache policy on the loads: the other waves of the workgroup re-read these lines)
    if (a.with_std) pair_loop<true, true, false>(x, sx, y, sy, mult, a.n, sb0, stride, lane, mine);
    else pair_loop<false, false, false>(x, sx, y, sy, mult, a.n, sb0, stride, lane, mine);
#pragma unroll
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            Mom m = c == ct ? mine[q] : mom_zero();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = mom_merge(m, mom_shfl_down(m, off));
            if (lane == 0)
                mom_store(partial + (((static_cast<int64_t>(blockIdx.x) * a.n_pairs + wave) * HM_MAX_CHANNELS + c) * 2 + q) * kMomVals, m);
        }
    }
}

// The same statistics with the frames staged through LDS (the launch's usual shape: N >= 3 frames, all their pairs). In k_pairs_stats
// every wave loads its own operands: 4 global loads per wave and chunk, 60 per workgroup for 14 distinct streams, and only the waves'
// rough lock-step makes the other 46 hit in L1 / L2. Here the workgroup's threads copy each stream's two 512-byte chunks of an
// iteration ONCE into LDS (16-byte items; item t of an iteration is frame t / 64, chunk (t % 64) / 32, piece t % 32 of the value image and,
// with stds, the same piece of the std image - NI = 1 or 2 items per thread) one iteration ahead of its use into one of three LDS stages, the loads in flight during the arithmetic before;
// after one barrier per iteration every wave reads its pair's operands with conflict-free ds_read_b64. HBM sees the read-once
// traffic, 1 KB per stream in flight per CU for a whole iteration, instead of 60 wave loads that mostly hit in cache.
// THR: AbstractMeasurand.apply_thresholds (modules/measurand.py:375-428) fused into the loader, which is the one place where every element of
// every frame passes exactly once per launch: a value outside [lo, hi] of its channel becomes NaN together with its std, in the LDS copy AND in
// the frame itself (a 16-byte store where something changed) - ExposureSeries.process_linearity (modules/exposure_series.py:437-441) leaves
// its image sets thresholded. One thread stages a value item AND its std item, so the decision needs no second reader. Idempotent, so
// the dropped re-read past the last whole iteration and a second launch over the same frames change nothing.
template <bool STD, int NI, bool THR>
__global__ __launch_bounds__(1024) void k_pairs_stats_lds(const PairsK a, int n_frames, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char stage_mem[];
    const uint32_t lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));       // = pair index, wave-uniform
    const int pi = a.pi[wave], pj = a.pj[wave];
    const double mult = a.mult[wave];
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 64;                               // a multiple of C (grid: multiple of 12)
    const int64_t step = kPairUN * stride;
    const int64_t sb0 = static_cast<int64_t>(blockIdx.x) * 64;
    const int ct = static_cast<int>((sb0 + lane) % a.C);
    const int n_streams = n_frames * (STD ? 2 : 1);
    const uint32_t stage_bytes = static_cast<uint32_t>(n_streams) * 1024u;
    const int n_items = n_frames * 64;                       // an item = 16 bytes (two elements) of one frame's value chunk - and of its std chunk
    const uint32_t std_off = static_cast<uint32_t>(n_frames) * 1024u;
    // this thread's items: source pointers at iteration 0; the LDS byte offset inside a stage is item * 16 (+ std_off for the std)
    const double* src[NI];
    const double* ssrc[NI];
    bool have[NI];
    double tlo[NI][2], thi[NI][2];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int item = static_cast<int>(threadIdx.x) + k * static_cast<int>(blockDim.x);
        have[k] = item < n_items;
        const int it_ = have[k] ? item : 0;
        const int frame = it_ >> 6, u = (it_ >> 5) & 1, piece = it_ & 31;
        const int64_t e0 = sb0 + u * stride + 2 * piece;
        src[k] = a.val[frame] + e0;
        ssrc[k] = STD ? a.sd[frame] + e0 : nullptr;
        if constexpr (THR) {                                  // channels of the item's two elements: constant along the loop (step % C == 0)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = static_cast<int>((e0 + j) % a.C);
                tlo[k][j] = c == 0 ? a.lo[0] : c == 1 ? a.lo[1] : c == 2 ? a.lo[2] : a.lo[3];
                thi[k][j] = c == 0 ? a.hi[0] : c == 1 ? a.hi[1] : c == 2 ? a.hi[2] : a.hi[3];
            }
        }
    }
    auto whole = [&](int64_t b) { return b + (kPairUN - 1) * stride + 64 <= a.n; };
    struct Fetched { f64x2 v[NI]; f64x2 s[STD ? NI : 1]; int64_t delta; };
    auto fetch = [&](int64_t delta, Fetched& f) {                                               // delta = element offset of the iteration from sb0
        f.delta = delta;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            f.v[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(src[k] + delta));   // read once
            if constexpr (STD) f.s[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(ssrc[k] + delta));
        }
    };
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    auto stash = [&](int stage, const Fetched& f) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            f64x2 v = f.v[k], sd = STD ? f.s[k] : f64x2{0.0, 0.0};
            if constexpr (THR) {                              // the value decides for itself and for its std (measurand.py:421-426; a NaN compares false)
                const bool m0 = v.x < tlo[k][0] || v.x > thi[k][0], m1 = v.y < tlo[k][1] || v.y > thi[k][1];
                v.x = m0 ? qnan : v.x; v.y = m1 ? qnan : v.y;
                if constexpr (STD) { sd.x = m0 ? qnan : sd.x; sd.y = m1 ? qnan : sd.y; }
                if (have[k] && (m0 || m1)) {                  // the frame itself, where it changed
                    *reinterpret_cast<f64x2*>(const_cast<double*>(src[k]) + f.delta) = v;
                    if constexpr (STD) *reinterpret_cast<f64x2*>(const_cast<double*>(ssrc[k]) + f.delta) = sd;
                }
            }
            if (have[k]) {
                char* dst = stage_mem + stage * stage_bytes + (threadIdx.x + k * blockDim.x) * 16u;
                *reinterpret_cast<f64x2*>(dst) = v;
                if constexpr (STD) *reinterpret_cast<f64x2*>(dst + std_off) = sd;
            }
        }
    };
    // a wave none of whose threads has an item (15 waves, 7 frames: items = 448 = waves 0..6) skips the loader altogether - its loads,
    // threshold selects and LDS stores were issued for nothing before: 180 of the ~3 000 VALU wave-instructions of a workgroup iteration
    const bool stager = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x & ~63u)) < n_items;          // wave-uniform
    MomAcc st[2] = {acc_zero(), acc_zero()};
    const uint32_t ox = static_cast<uint32_t>(pi) * 1024u + lane * 8u, oy = static_cast<uint32_t>(pj) * 1024u + lane * 8u;
    const uint32_t osx = ox + std_off, osy = oy + std_off;
    auto compute = [&](int stage, int it) {
        const char* sm = stage_mem + stage * stage_bytes;
        double xv[kPairUN], yv[kPairUN], xs[kPairUN], ys[kPairUN];
#pragma unroll
        for (int u = 0; u < kPairUN; ++u) {
            xv[u] = *reinterpret_cast<const double*>(sm + ox + u * 512);
            yv[u] = *reinterpret_cast<const double*>(sm + oy + u * 512);
            xs[u] = STD ? *reinterpret_cast<const double*>(sm + osx + u * 512) : 0.0;
            ys[u] = STD ? *reinterpret_cast<const double*>(sm + osy + u * 512) : 0.0;
        }
        pair_process_any<STD>(st, it, mult, xv, xs, yv, ys);
    };
    // Whole iterations (sb is the same for every wave of the workgroup, so the trip count is too). Iteration k: the registers hold
    // iteration k + 1 (loaded during iteration k - 1's arithmetic) -> LDS stage (k + 1) % 3; load iteration k + 2; barrier; compute
    // iteration k from stage k % 3. Three stages: a fast wave writes stage k + 1 while a slow one may still read stage k - 1.
    // One register set and one loop body: the only loads outstanding at the wait are the ones it is for. Past the last whole
    // iteration the prefetch re-reads the current one (dropped) - no branch around a load, see pair_loop.
    int64_t sb = sb0;
    int it = 0, stage = 0;
    Fetched r;
    auto clamp = [&](int64_t b) { return (whole(b) ? b : sb) - sb0; };
    if (whole(sb) && stager) { fetch(0, r); stash(0, r); fetch(clamp(sb + step), r); }
    while (whole(sb)) {
        const int nxt = stage == 2 ? 0 : stage + 1;
        if (stager) {
            stash(nxt, r);
            fetch(clamp(sb + 2 * step), r);
        }
        __syncthreads();
        compute(stage, it);
        ++it; sb += step; stage = nxt;
    }
    Mom mine[2];
    const double* x = a.val[pi];
    const double* y = a.val[pj];
    pair_tail<STD, STD>(st, it, sb, x, STD ? a.sd[pi] : nullptr, y, STD ? a.sd[pj] : nullptr, mult, a.n, stride, lane, mine);
#pragma unroll
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            Mom m = c == ct ? mine[q] : mom_zero();
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = mom_merge(m, mom_shfl_down(m, off));
            if (lane == 0)
                mom_store(partial + (((static_cast<int64_t>(blockIdx.x) * a.n_pairs + wave) * HM_MAX_CHANNELS + c) * 2 + q) * kMomVals, m);
        }
    }
}

// one workgroup per pair: folds the per-block partials of its pair in a fixed order
__global__ __launch_bounds__(256) void k_pairs_final(const double* __restrict__ partial, int nblocks, int n_pairs, int C, int weighted,
                                                     double* __restrict__ out) {
    __shared__ double tree[256][kMomVals];
    const int pair = blockIdx.x;
    const int h = blockIdx.y / C, c = blockIdx.y % C;       // one workgroup per (pair, difference kind, channel)
    Mom m = mom_zero();
    for (int b = threadIdx.x; b < nblocks; b += 256)
        m = mom_merge(m, mom_load(partial + (((static_cast<int64_t>(b) * n_pairs + pair) * HM_MAX_CHANNELS + c) * 2 + h) * kMomVals));
    mom_store(tree[threadIdx.x], m);
    __syncthreads();
    for (int span = 128; span > 0; span >>= 1) {
        if (static_cast<int>(threadIdx.x) < span) {
            const Mom r = mom_merge(mom_load(tree[threadIdx.x]), mom_load(tree[threadIdx.x + span]));
            mom_store(tree[threadIdx.x], r);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean, sdv, err;
        mom_finish(mom_load(tree[0]), weighted != 0, mean, sdv, err);
        double* o = out + static_cast<int64_t>(pair) * 6 * C + 3 * C * h;
        o[c] = mean; o[C + c] = sdv; o[2 * C + c] = err;
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
    const bool al16 = aligned(x, 16) && aligned(y, 16) && aligned(out_abs, 16) && aligned(out_rel, 16) && (!sx || aligned(sx, 16)) &&
                      (!sy || aligned(sy, 16)) && (!with_std || (aligned(out_abs_std, 16) && aligned(out_rel_std, 16)));
    if (al16) {
        const int64_t n_chunks = n / kDiffChunk;
        if (n_chunks > 0) {
            const unsigned bgrid = stream_grid((n_chunks + 3) / 4, 1, 16);
#define HM_DIFFB(A, B) hipLaunchKernelGGL((k_difference_burst<A, B>), dim3(bgrid), dim3(256), 0, as_stream(stream), \
                                          x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, n_chunks)
            if (sx && sy) HM_DIFFB(true, true); else if (sx) HM_DIFFB(true, false); else if (sy) HM_DIFFB(false, true); else HM_DIFFB(false, false);
#undef HM_DIFFB
            const int64_t done = n_chunks * kDiffChunk;                  // the remainder (< 512 elements) through the per-lane kernel
            if (done == n) return launch_status();
            x += done; y += done; out_abs += done; out_rel += done; n -= done;
            if (sx) sx += done;
            if (sy) sy += done;
            if (out_abs_std) { out_abs_std += done; out_rel_std += done; }
        }
        const unsigned dgrid = stream_grid((n + 1) / 2, 256, 8);
#define HM_DIFF(A, B) hipLaunchKernelGGL((k_difference_dense<A, B>), dim3(dgrid), dim3(256), 0, as_stream(stream), \
                                         x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, n)
        if (sx && sy) HM_DIFF(true, true); else if (sx) HM_DIFF(true, false); else if (sy) HM_DIFF(false, true); else HM_DIFF(false, false);
#undef HM_DIFF
        return launch_status();
    }
    hipLaunchKernelGGL(k_difference, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream),
                       x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, n);
    return launch_status();
}

extern "C" int hm_interpolate(const double* x0, const double* s0, const double* x1, const double* s1,
                              double y0, double y1, double y, double* out, double* out_std, int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x0 || !x1 || !out || ((s0 || s1) != (out_std != nullptr))) return HM_EINVAL;
    const bool al16 = aligned(x0, 16) && aligned(x1, 16) && aligned(out, 16) && (!s0 || aligned(s0, 16)) && (!s1 || aligned(s1, 16)) &&
                      (!out_std || aligned(out_std, 16));
    const int64_t n_chunks = n / kDiffChunk;
    if (al16 && n_chunks > 0) {
        const unsigned bgrid = stream_grid((n_chunks + 3) / 4, 1, 16);
#define HM_INTB(A, B) hipLaunchKernelGGL((k_interpolate_burst<A, B>), dim3(bgrid), dim3(256), 0, as_stream(stream), \
                                         x0, s0, x1, s1, y0, y1, y, out, out_std, n_chunks)
        if (s0 && s1) HM_INTB(true, true); else if (s0) HM_INTB(true, false); else if (s1) HM_INTB(false, true); else HM_INTB(false, false);
#undef HM_INTB
        const int64_t done = n_chunks * kDiffChunk;                      // the remainder (< 512 elements) through the per-lane kernel
        if (done == n) return launch_status();
        x0 += done; x1 += done; out += done; n -= done;
        if (s0) s0 += done;
        if (s1) s1 += done;
        if (out_std) out_std += done;
    }
    hipLaunchKernelGGL(k_interpolate, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream),
                       x0, s0, x1, s1, y0, y1, y, out, out_std, n);
    return launch_status();
}

extern "C" size_t hm_channel_statistics_workspace_bytes(void) {
    return sizeof(double) * (kStatBlocks * HM_MAX_CHANNELS * kMomVals);
}

extern "C" int hm_channel_statistics(const double* val, const double* std, int64_t n, int C,
                                     double* out /*3*C: mean, std, error*/, void* workspace, void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !val || !out || !workspace) return HM_EINVAL;
    if (!aligned(val, 8) || (std && !aligned(std, 8))) return HM_EALIGN;
    double* partial = static_cast<double*>(workspace);
    const int grid = stat_grid(n);
    hipStream_t st = as_stream(stream);
    if (std) hipLaunchKernelGGL(k_stats<true>, dim3(grid), dim3(256), 0, st, val, std, n, C, partial);
    else hipLaunchKernelGGL(k_stats<false>, dim3(grid), dim3(256), 0, st, val, std, n, C, partial);
    hipLaunchKernelGGL(k_stats_final, dim3(C), dim3(256), 0, st, partial, grid, C, std ? 1 : 0, out);
    return launch_status();
}

extern "C" size_t hm_pair_statistics_workspace_bytes(void) {
    return sizeof(double) * (kStatBlocks * HM_MAX_CHANNELS * 2 * kMomVals);
}

extern "C" int hm_pair_statistics(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                                  int64_t n, int C, double* out /*6*C*/, void* workspace, void* stream) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !x || !y || !out || !workspace) return HM_EINVAL;
    double* partial = static_cast<double*>(workspace);
    const int grid = stat_grid(n);
    hipStream_t st = as_stream(stream);
#define HM_PAIR(A, B) hipLaunchKernelGGL((k_pair_stats<A, B>), dim3(grid), dim3(256), 0, st, x, sx, y, sy, multiplier, n, C, partial)
    if (sx && sy) HM_PAIR(true, true); else if (sx) HM_PAIR(true, false); else if (sy) HM_PAIR(false, true); else HM_PAIR(false, false);
#undef HM_PAIR
    hipLaunchKernelGGL(k_pair_final, dim3(2 * C), dim3(256), 0, st, partial, grid, C, (sx || sy) ? 1 : 0, out);
    return launch_status();
}

extern "C" size_t hm_pairs_statistics_workspace_bytes(int n_pairs) {
    const int p = n_pairs < 1 ? 1 : (n_pairs > HM_PAIRS_MAX ? HM_PAIRS_MAX : n_pairs);
    return sizeof(double) * static_cast<size_t>(kStatBlocks) * static_cast<size_t>(p) * HM_MAX_CHANNELS * 2 * kMomVals;
}

extern "C" int hm_pairs_statistics(const double* const* vals, const double* const* stds, int n_frames, const int32_t* pair_i,
                                   const int32_t* pair_j, const double* multipliers, int n_pairs, int64_t n, int C,
                                   const double* lower, const double* upper,
                                   double* out /*n_pairs * 6C*/, void* workspace, void* stream) {
    if (!vals || !pair_i || !pair_j || !multipliers || !out || !workspace || n < 1 || C < 1 || C > HM_MAX_CHANNELS) return HM_EINVAL;
    if (n_frames < 1 || n_frames > HM_MAX_FRAMES || n_pairs < 1 || ((lower != nullptr) != (upper != nullptr))) return HM_EINVAL;
    const bool thr = lower != nullptr;
    PairsK k{};
    k.n = n; k.C = C; k.with_std = stds ? 1 : 0;
    ChanLimits lim{};
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) {
        k.lo[c] = lim.lo[c] = (thr && c < C) ? lower[c] : -__builtin_huge_val();
        k.hi[c] = lim.hi[c] = (thr && c < C) ? upper[c] : __builtin_huge_val();
    }
    for (int i = 0; i < n_frames; ++i) {
        if (!vals[i] || !aligned(vals[i], 8)) return HM_EINVAL;
        k.val[i] = vals[i];
        if (stds) { if (!stds[i]) return HM_EINVAL; k.sd[i] = stds[i]; }
    }
    for (int p = 0; p < n_pairs; ++p)
        if (pair_i[p] < 0 || pair_i[p] >= n_frames || pair_j[p] < 0 || pair_j[p] >= n_frames) return HM_EINVAL;
    hipStream_t st = as_stream(stream);
    double* partial = static_cast<double*>(workspace);
    int64_t g = (n + 63) / 64;
    g = ((g + 11) / 12) * 12;
    const int grid = static_cast<int>(g < kStatBlocks ? g : kStatBlocks);
    const int n_streams = n_frames * (stds ? 2 : 1);
    bool al16 = true;
    for (int i = 0; i < n_frames; ++i) al16 = al16 && aligned(vals[i], 16) && (!stds || aligned(stds[i], 16));
    // the thresholds (in place, as apply_thresholds does) ride on the first launch's loader when that launch stages the frames through LDS:
    // every element below `fused_end` lies in an iteration that is whole for every workgroup. The rest - and everything when the first
    // launch cannot use the LDS kernel - goes through k_thresholds first.
    auto lds_ok = [&](int np) { return al16 && n_frames * 64 <= 2 * 64 * np && n_streams <= 21; };
    const int np_first = n_pairs < HM_PAIRS_MAX ? n_pairs : HM_PAIRS_MAX;
    bool fuse_thr = false;
    if (thr) {
        const int64_t stride = static_cast<int64_t>(grid) * 64;
        int64_t whole_all = 0;                                               // iterations that are whole for the LAST workgroup (hence for all)
        while ((static_cast<int64_t>(grid) - 1) * 64 + (2 * whole_all + 1) * stride + 64 <= n) ++whole_all;
        const int64_t fused_end = lds_ok(np_first) ? 2 * stride * whole_all : 0;         // a multiple of C (stride is)
        fuse_thr = fused_end > 0;
        if (fused_end < n) {
            const int64_t m = n - fused_end;
            for (int i = 0; i < n_frames; ++i) {
                double* v = const_cast<double*>(vals[i]) + fused_end;
                double* sdp = stds ? const_cast<double*>(stds[i]) + fused_end : nullptr;
                hipLaunchKernelGGL(k_thresholds, dim3(stream_grid((m + 1) / 2, 256, 8)), dim3(256), 0, st, v, sdp, lim, m, C);
            }
        }
    }
    for (int p0 = 0; p0 < n_pairs; p0 += HM_PAIRS_MAX) {                 // HM_PAIRS_MAX pairs (waves of a workgroup) per launch
        const int np = n_pairs - p0 < HM_PAIRS_MAX ? n_pairs - p0 : HM_PAIRS_MAX;
        k.n_pairs = np;
        for (int p = 0; p < np; ++p) { k.pi[p] = pair_i[p0 + p]; k.pj[p] = pair_j[p0 + p]; k.mult[p] = multipliers[p0 + p]; }
        // frames through LDS when a workgroup's threads can copy an iteration's 64 * n_streams 16-byte items with at most two each
        const int items = n_frames * 64, threads = 64 * np;          // one item = 16 bytes of a frame's value chunk (+ the same of its std)
        if (lds_ok(np)) {                                                                   // 3 stages x n_streams KB <= 64 KB of LDS
            const size_t lds = 3u * static_cast<size_t>(n_streams) * 1024u;
            const bool t_ = fuse_thr && p0 == 0;
#define HM_PLDS(S, N, T) hipLaunchKernelGGL((k_pairs_stats_lds<S, N, T>), dim3(grid), dim3(threads), lds, st, k, n_frames, partial)
            if (stds) {
                if (items <= threads) { if (t_) HM_PLDS(true, 1, true); else HM_PLDS(true, 1, false); }
                else { if (t_) HM_PLDS(true, 2, true); else HM_PLDS(true, 2, false); }
            } else {
                if (items <= threads) { if (t_) HM_PLDS(false, 1, true); else HM_PLDS(false, 1, false); }
                else { if (t_) HM_PLDS(false, 2, true); else HM_PLDS(false, 2, false); }
            }
#undef HM_PLDS
        } else
            hipLaunchKernelGGL(k_pairs_stats, dim3(grid), dim3(64 * np), 0, st, k, partial);
        hipLaunchKernelGGL(k_pairs_final, dim3(np, 2 * C), dim3(256), 0, st, partial, grid, np, C, k.with_std, out + static_cast<int64_t>(p0) * 6 * C);
    }
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

// ================================================================================================
// compute_dimension_statistics over ANY axis (modules/measurand.py:318-350 take a NumPy axis argument): the array is seen as a dense
// (outer, A, inner) block and reduced over A - out[o, i] from x[(o A + k) inner + i], k = 0..A-1. (Two separate groups of reduced axes:
// hm_axis_statistics2 / k_axis_final2 below; three or more are brought together by the caller with a layout copy.) Same one-pass moments and NaN rules as the channel statistics above
// (MomAcc / mom_merge / mom_finish). Two shapes:
//   k_axis_thread   one thread per output (o, i): lanes run along i (contiguous) - inner >= 16, or short axes (A <= 32: e.g. the
//                   channel axis of an image, where a lane reads its A consecutive values itself);
//   k_axis_row      inner < 16 and a long axis: the A x inner values of one o are contiguous; a workgroup strides over them with a
//                   thread count that is a multiple of inner (every thread keeps ONE channel), then folds threads of equal channel.
// Both split A into KS segments (grid dimension) when there are too few outputs to fill the chip; k_axis_final folds the segments.
// ================================================================================================
namespace hm {

__device__ __forceinline__ void axis_finish_store(const Mom& m, bool weighted, int64_t j, double* out_mean, double* out_std, double* out_err) {
    double mean, sdv, err;
    mom_finish(m, weighted, mean, sdv, err);
    out_mean[j] = mean; out_std[j] = sdv;
    if (out_err) out_err[j] = err;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_axis_thread(const double* __restrict__ val, const double* __restrict__ sd, int64_t outer, int64_t A,
                                                     int64_t inner, int KS, double* __restrict__ partial, double* __restrict__ out_mean,
                                                     double* __restrict__ out_std, double* __restrict__ out_err, int keep_partial) {
    const int64_t n_out = outer * inner;
    const int seg = blockIdx.y;
    const int64_t jstride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; j < n_out; j += jstride) {
        const int64_t o = j / inner, i = j - o * inner;
        const double* pv = val + (o * A) * inner + i;
        const double* ps = WEIGHTED ? sd + (o * A) * inner + i : nullptr;
        MomAcc st = acc_zero();
        int it = 0;
        // four axis positions per iteration, their loads issued together (one dependent load per iteration left the lanes waiting on HBM
        // latency: 0.36 of the roofline on axis 0 of a 4096 x 4096 x 3 image)
        constexpr int UNR = 4;
        int64_t k = seg;
        for (; k + static_cast<int64_t>(UNR - 1) * KS < A; k += static_cast<int64_t>(UNR) * KS) {
            double v[UNR], sv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                v[u] = __builtin_nontemporal_load(pv + (k + static_cast<int64_t>(u) * KS) * inner);
                sv[u] = WEIGHTED ? __builtin_nontemporal_load(ps + (k + static_cast<int64_t>(u) * KS) * inner) : 1.0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const double w = WEIGHTED ? 1.0 / sv[u] : 1.0;                  // measurand.py:342
                acc_add(st, v[u], w, sv[u], WEIGHTED, true);
            }
            it += UNR;
            if ((it & (kMomBlock - 1)) == 0 || it == UNR) acc_fold<false>(st);
        }
        if (k < A) {                                                            // the last 1-3 positions: loads together, missing ones at weight 0
            double v[UNR - 1], sv[UNR - 1];
            bool ok[UNR - 1];
#pragma unroll
            for (int u = 0; u < UNR - 1; ++u) {
                const int64_t kk = k + static_cast<int64_t>(u) * KS;
                ok[u] = kk < A;
                const int64_t kc = ok[u] ? kk : k;
                v[u] = pv[kc * inner];
                sv[u] = WEIGHTED ? ps[kc * inner] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < UNR - 1; ++u) {
                const double w = WEIGHTED ? 1.0 / sv[u] : 1.0;
                acc_add(st, v[u], w, sv[u], WEIGHTED, ok[u]);
            }
        }
        const Mom m = acc_finish<false>(st, WEIGHTED);
        if (KS == 1 && !keep_partial) axis_finish_store(m, WEIGHTED, j, out_mean, out_std, out_err);
        else mom_store(partial + (static_cast<int64_t>(seg) * n_out + j) * kMomVals, m);
    }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_axis_row(const double* __restrict__ val, const double* __restrict__ sd, int64_t outer, int64_t A,
                                                  int64_t inner, int KS, double* __restrict__ partial, double* __restrict__ out_mean,
                                                  double* __restrict__ out_std, double* __restrict__ out_err, int keep_partial) {
    __shared__ double red[256][kMomVals];
    const int in = static_cast<int>(inner);
    const int Tm = (256 / in) * in;                                             // active threads: a multiple of inner
    const int seg = blockIdx.x;
    const int64_t seg_k = (A + KS - 1) / KS;                                    // axis positions per segment
    const int64_t e_lo = static_cast<int64_t>(seg) * seg_k * inner;
    int64_t e_hi = e_lo + seg_k * inner;
    if (e_hi > A * inner) e_hi = A * inner;
    const int64_t n_out = outer * inner;
    for (int64_t o = blockIdx.y; o < outer; o += gridDim.y) {
        MomAcc st = acc_zero();
        if (static_cast<int>(threadIdx.x) < Tm) {
            const double* pv = val + o * A * inner;
            const double* ps = WEIGHTED ? sd + o * A * inner : nullptr;
            int it = 0;
            constexpr int UNR = 4;
            int64_t e = e_lo + threadIdx.x;
            for (; e + static_cast<int64_t>(UNR - 1) * Tm < e_hi; e += static_cast<int64_t>(UNR) * Tm) {
                double v[UNR], sv[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    v[u] = __builtin_nontemporal_load(pv + e + static_cast<int64_t>(u) * Tm);
                    sv[u] = WEIGHTED ? __builtin_nontemporal_load(ps + e + static_cast<int64_t>(u) * Tm) : 1.0;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const double w = WEIGHTED ? 1.0 / sv[u] : 1.0;
                    acc_add(st, v[u], w, sv[u], WEIGHTED, true);
                }
                it += UNR;
                if ((it & (kMomBlock - 1)) == 0 || it == UNR) acc_fold<false>(st);
            }
            for (; e < e_hi; e += Tm) {
                const double v = pv[e];
                const double s = WEIGHTED ? ps[e] : 1.0;
                const double w = WEIGHTED ? 1.0 / s : 1.0;
                acc_add(st, v, w, s, WEIGHTED, true);
                if ((++it & (kMomBlock - 1)) == 0) acc_fold<false>(st);
            }
        }
        const Mom mine = acc_finish<false>(st, WEIGHTED);
        __syncthreads();                                                        // (the previous o is done with `red`)
        mom_store(red[threadIdx.x], mine);
        __syncthreads();
        // thread t = p * inner + c holds partial p of channel c: a halving tree over p (fixed order; 7 steps for 85 partials - the serial fold
        // by `inner` threads it replaces was 85 dependent divisions per row)
        for (int P = Tm / in; P > 1; P = (P + 1) / 2) {
            const int half = (P + 1) / 2;
            const int pidx = static_cast<int>(threadIdx.x) / in;
            const bool act = static_cast<int>(threadIdx.x) < Tm && pidx < half && pidx + half < P;
            Mom r = mom_zero();
            if (act) r = mom_merge(mom_load(red[threadIdx.x]), mom_load(red[threadIdx.x + half * in]));
            __syncthreads();
            if (act) mom_store(red[threadIdx.x], r);
            __syncthreads();
        }
        if (static_cast<int>(threadIdx.x) < in) {
            const Mom m = mom_load(red[threadIdx.x]);
            const int64_t j = o * inner + threadIdx.x;
            if (KS == 1 && !keep_partial) axis_finish_store(m, WEIGHTED, j, out_mean, out_std, out_err);
            else mom_store(partial + (static_cast<int64_t>(seg) * n_out + j) * kMomVals, m);
        }
    }
}

__global__ __launch_bounds__(256) void k_axis_final(const double* __restrict__ partial, int KS, int64_t n_out, int weighted,
                                                    double* __restrict__ out_mean, double* __restrict__ out_std, double* __restrict__ out_err) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; j < n_out; j += stride) {
        Mom m = mom_load(partial + j * kMomVals);
        for (int s = 1; s < KS; ++s) m = mom_merge(m, mom_load(partial + (static_cast<int64_t>(s) * n_out + j) * kMomVals));
        axis_finish_store(m, weighted != 0, j, out_mean, out_std, out_err);
    }
}

// few outputs, many segments: one workgroup per output, thread t folds segments t, t + 256, ... and an LDS tree finishes (a single thread
// folding 1024 segments one after the other took 50 us)
__global__ __launch_bounds__(256) void k_axis_final_tree(const double* __restrict__ partial, int KS, int64_t n_out, int weighted,
                                                         double* __restrict__ out_mean, double* __restrict__ out_std, double* __restrict__ out_err) {
    __shared__ double tree[256][kMomVals];
    const int64_t j = blockIdx.x;
    Mom m = mom_zero();
    for (int s = threadIdx.x; s < KS; s += 256) m = mom_merge(m, mom_load(partial + (static_cast<int64_t>(s) * n_out + j) * kMomVals));
    mom_store(tree[threadIdx.x], m);
    __syncthreads();
    for (int span = 128; span > 0; span >>= 1) {
        if (static_cast<int>(threadIdx.x) < span) {
            const Mom r = mom_merge(mom_load(tree[threadIdx.x]), mom_load(tree[threadIdx.x + span]));
            mom_store(tree[threadIdx.x], r);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) axis_finish_store(mom_load(tree[0]), weighted != 0, j, out_mean, out_std, out_err);
}

// Two groups of reduced axes with kept axes between them - the array as (outer, A1, mid, A2, inner), reduced over A1 and A2 with no layout
// copy of the input. Stage 1 is the one-group reduction over the LONGER of the two (A1: inner' = mid * A2 * inner; A2: outer' = outer * A1 *
// mid), its states kept as partials; stage 2 folds, per output (o, m, i), the R positions of the other group and the KS segments in a fixed
// order: state index o * o_stride + m * m_stride + r * r_stride + i.
struct Final2K { int64_t n_out1, n_out2, mid, inner, o_stride, m_stride, r_stride, R; int KS, weighted; };
__device__ __forceinline__ int64_t final2_base(const Final2K& f, int64_t j) {
    const int64_t i = j % f.inner, om = j / f.inner;
    return (om / f.mid) * f.o_stride + (om % f.mid) * f.m_stride + i;
}
__global__ __launch_bounds__(256) void k_axis_final2(const double* __restrict__ partial, const Final2K f, double* __restrict__ out_mean,
                                                     double* __restrict__ out_std, double* __restrict__ out_err) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; j < f.n_out2; j += stride) {
        const int64_t base = final2_base(f, j);
        Mom m = mom_zero();
        for (int64_t r = 0; r < f.R; ++r)
            for (int s = 0; s < f.KS; ++s) m = mom_merge(m, mom_load(partial + (static_cast<int64_t>(s) * f.n_out1 + base + r * f.r_stride) * kMomVals));
        axis_finish_store(m, f.weighted != 0, j, out_mean, out_std, out_err);
    }
}
// few outputs, many states each: one workgroup per output, thread t folds states t, t + 256, ... of the R x KS, then an LDS tree
__global__ __launch_bounds__(256) void k_axis_final2_tree(const double* __restrict__ partial, const Final2K f, double* __restrict__ out_mean,
                                                          double* __restrict__ out_std, double* __restrict__ out_err) {
    __shared__ double tree[256][kMomVals];
    const int64_t j = blockIdx.x, base = final2_base(f, j), total = f.R * f.KS;
    Mom m = mom_zero();
    for (int64_t q = threadIdx.x; q < total; q += 256) {
        const int64_t r = q / f.KS, sgm = q % f.KS;
        m = mom_merge(m, mom_load(partial + (sgm * f.n_out1 + base + r * f.r_stride) * kMomVals));
    }
    mom_store(tree[threadIdx.x], m);
    __syncthreads();
    for (int span = 128; span > 0; span >>= 1) {
        if (static_cast<int>(threadIdx.x) < span) {
            const Mom r = mom_merge(mom_load(tree[threadIdx.x]), mom_load(tree[threadIdx.x + span]));
            mom_store(tree[threadIdx.x], r);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) axis_finish_store(mom_load(tree[0]), f.weighted != 0, j, out_mean, out_std, out_err);
}

// the element count of a dense block of the given extents, or -1 when an extent is < 1 or the product does not fit (absurd arguments must
// come back as HM_EINVAL / 0, not as an integer overflow inside the launch arithmetic)
static int64_t dense_elems(std::initializer_list<int64_t> dims) {
    int64_t n = 1;
    for (int64_t d : dims) {
        if (d < 1 || __builtin_mul_overflow(n, d, &n) || n > (int64_t{1} << 48)) return -1;
    }
    return n;
}

struct AxisPlan { bool row; int KS; };
static AxisPlan axis_plan(int64_t outer, int64_t A, int64_t inner) {
    AxisPlan p;
    p.row = inner < 16 && A > 32;
    int64_t ks = 1;
    if (!p.row) {
        const int64_t blocks = (outer * inner + 255) / 256;
        if (blocks < 1024) { ks = (1024 + blocks - 1) / blocks; const int64_t cap = A / 16 > 1 ? A / 16 : 1; if (ks > cap) ks = cap; }
        if (ks > 256) ks = 256;
    } else {
        const int64_t tm = (256 / inner) * inner;
        if (outer < 1024) { ks = (1024 + outer - 1) / outer; const int64_t cap = (A * inner) / (tm * 8) > 1 ? (A * inner) / (tm * 8) : 1; if (ks > cap) ks = cap; }
        if (ks > 1024) ks = 1024;
    }
    p.KS = static_cast<int>(ks);
    return p;
}

// compute_difference / interpolate on BROADCAST operands (modules/measurand.py:621-681 apply NumPy broadcasting to x and y): element
// strides with 0 on broadcast axes, as hm_binary_op; x and its std share strides (one Measurand), y and its std likewise.
struct Bcast2K { int64_t shape[HM_MAX_DIMS], st1[HM_MAX_DIMS], st2[HM_MAX_DIMS]; int ndim; };
__device__ __forceinline__ void bcast_offsets(const Bcast2K& b, int64_t e, int64_t& o1, int64_t& o2) {
    o1 = 0; o2 = 0;
    int64_t rem = e;
    for (int d = b.ndim - 1; d >= 0; --d) {
        const int64_t i = rem % b.shape[d];
        rem /= b.shape[d];
        o1 += i * b.st1[d];
        o2 += i * b.st2[d];
    }
}
__global__ __launch_bounds__(256) void k_difference_bcast(const double* __restrict__ x, const double* __restrict__ sx,
                                                          const double* __restrict__ y, const double* __restrict__ sy, double mult,
                                                          double* __restrict__ ad, double* __restrict__ ads,
                                                          double* __restrict__ rd, double* __restrict__ rds, int64_t n, const Bcast2K b) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        int64_t ox, oy;
        bcast_offsets(b, e, ox, oy);
        const double xv = x[ox], yv = y[oy];
        const double scale = mult * yv;                     // :634
        const double a = xv - scale;                        // :635
        ad[e] = a;
        rd[e] = a / scale;                                  // :636
        if (ads) {
            const double xs = sx ? sx[ox] : 0.0, ys = sy ? sy[oy] : 0.0;
            const double m1 = mult * ys;
            ads[e] = sqrt(xs * xs + m1 * m1);               // :652
            const double u1 = xs / (mult * yv);
            const double u2 = (ys * xv) / (mult * (yv * yv));
            rds[e] = sqrt(u1 * u1 + u2 * u2);               // :653
        }
    }
}
__global__ __launch_bounds__(256) void k_interpolate_bcast(const double* __restrict__ x0, const double* __restrict__ s0,
                                                           const double* __restrict__ x1, const double* __restrict__ s1,
                                                           double y0, double y1, double y, double* __restrict__ out,
                                                           double* __restrict__ out_std, int64_t n, const Bcast2K b) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const double a = y1 - y, bb = y - y0, d = y1 - y0;
    const double ca = (a / d) * (a / d), cb = (bb / d) * (bb / d);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        int64_t o0, o1;
        bcast_offsets(b, e, o0, o1);
        out[e] = (x0[o0] * a + x1[o1] * bb) / d;            // :665
        if (out_std) out_std[e] = sqrt((s0 ? s0[o0] : 0.0) * ca + (s1 ? s1[o1] : 0.0) * cb);   // :679 as written
    }
}

static bool fill_bcast(Bcast2K& b, int ndim, const int64_t* shape, const int64_t* st1, const int64_t* st2, int64_t& n) {
    if (ndim < 1 || ndim > HM_MAX_DIMS || !shape || !st1 || !st2) return false;
    n = 1;
    b.ndim = ndim;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] < 0 || st1[d] < 0 || st2[d] < 0) return false;
        b.shape[d] = shape[d]; b.st1[d] = st1[d]; b.st2[d] = st2[d];
        n *= shape[d];
    }
    return true;
}

}  // namespace hm

extern "C" size_t hm_axis_statistics_workspace_bytes(int64_t outer, int64_t axis_len, int64_t inner) {
    if (hm::dense_elems({outer, axis_len, inner}) < 0) return 0;
    const hm::AxisPlan p = hm::axis_plan(outer, axis_len, inner);
    return p.KS > 1 ? static_cast<size_t>(p.KS) * static_cast<size_t>(outer * inner) * hm::kMomVals * sizeof(double) : 0;
}

namespace hm {
// stage 1 of both entry points: the one-group reduction; keep_partial = leave the states in `partial` even when there is one segment
static void launch_axis_stage1(const double* val, const double* std, int64_t outer, int64_t axis_len, int64_t inner, const AxisPlan& p,
                               double* partial, double* out_mean, double* out_std, double* out_err, int keep_partial, hipStream_t st) {
    const int64_t n_out = outer * inner;
    if (!p.row) {
        int64_t gx = (n_out + 255) / 256;
        if (gx > (int64_t{1} << 20)) gx = int64_t{1} << 20;
        const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(p.KS));
        if (std) hipLaunchKernelGGL(k_axis_thread<true>, grid, dim3(256), 0, st, val, std, outer, axis_len, inner, p.KS, partial, out_mean, out_std, out_err, keep_partial);
        else hipLaunchKernelGGL(k_axis_thread<false>, grid, dim3(256), 0, st, val, std, outer, axis_len, inner, p.KS, partial, out_mean, out_std, out_err, keep_partial);
    } else {
        const dim3 grid(static_cast<unsigned>(p.KS), static_cast<unsigned>(outer < 65535 ? outer : 65535));
        if (std) hipLaunchKernelGGL(k_axis_row<true>, grid, dim3(256), 0, st, val, std, outer, axis_len, inner, p.KS, partial, out_mean, out_std, out_err, keep_partial);
        else hipLaunchKernelGGL(k_axis_row<false>, grid, dim3(256), 0, st, val, std, outer, axis_len, inner, p.KS, partial, out_mean, out_std, out_err, keep_partial);
    }
}
}  // namespace hm

extern "C" int hm_axis_statistics(const double* val, const double* std, int64_t outer, int64_t axis_len, int64_t inner,
                                  double* out_mean, double* out_std, double* out_err, void* workspace, void* stream) {
    using namespace hm;
    if (dense_elems({outer, axis_len, inner}) < 0 || !val || !out_mean || !out_std) return HM_EINVAL;
    if (!aligned(val, 8) || (std && !aligned(std, 8))) return HM_EALIGN;
    const AxisPlan p = axis_plan(outer, axis_len, inner);
    if (p.KS > 1 && !workspace) return HM_EINVAL;
    double* partial = static_cast<double*>(workspace);
    hipStream_t st = as_stream(stream);
    const int64_t n_out = outer * inner;
    launch_axis_stage1(val, std, outer, axis_len, inner, p, partial, out_mean, out_std, out_err, 0, st);
    if (p.KS > 1) {
        if (p.KS >= 32 && n_out <= 4096)
            hipLaunchKernelGGL(k_axis_final_tree, dim3(static_cast<unsigned>(n_out)), dim3(256), 0, st, partial, p.KS, n_out, std ? 1 : 0, out_mean, out_std, out_err);
        else
            hipLaunchKernelGGL(k_axis_final, dim3(stream_grid(n_out, 256, 8)), dim3(256), 0, st, partial, p.KS, n_out, std ? 1 : 0, out_mean, out_std, out_err);
    }
    return launch_status();
}

namespace hm {
// which group stage 1 reduces (the longer one), the (outer, A, inner) view it sees and the index map of stage 2
struct Axis2Plan { int64_t outer1, A, inner1; AxisPlan p; Final2K f; };
static Axis2Plan axis2_plan(int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner) {
    Axis2Plan q{};
    if (a1 >= a2) {            // stage 1 over A1: state index ((o mid + m) A2 + r) inner + i
        q.outer1 = outer; q.A = a1; q.inner1 = mid * a2 * inner;
        q.f.o_stride = mid * a2 * inner; q.f.m_stride = a2 * inner; q.f.r_stride = inner; q.f.R = a2;
    } else {                   // stage 1 over A2: state index ((o A1 + r) mid + m) inner + i
        q.outer1 = outer * a1 * mid; q.A = a2; q.inner1 = inner;
        q.f.o_stride = a1 * mid * inner; q.f.m_stride = inner; q.f.r_stride = mid * inner; q.f.R = a1;
    }
    q.p = axis_plan(q.outer1, q.A, q.inner1);
    q.f.n_out1 = q.outer1 * q.inner1; q.f.n_out2 = outer * mid * inner; q.f.mid = mid; q.f.inner = inner; q.f.KS = q.p.KS;
    return q;
}
}  // namespace hm

extern "C" size_t hm_axis_statistics2_workspace_bytes(int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner) {
    if (hm::dense_elems({outer, a1, mid, a2, inner}) < 0) return 0;
    const hm::Axis2Plan q = hm::axis2_plan(outer, a1, mid, a2, inner);
    return static_cast<size_t>(q.p.KS) * static_cast<size_t>(q.f.n_out1) * hm::kMomVals * sizeof(double);
}

extern "C" int hm_axis_statistics2(const double* val, const double* std, int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner,
                                   double* out_mean, double* out_std, double* out_err, void* workspace, void* stream) {
    using namespace hm;
    if (dense_elems({outer, a1, mid, a2, inner}) < 0 || !val || !out_mean || !out_std || !workspace) return HM_EINVAL;
    if (!aligned(val, 8) || (std && !aligned(std, 8))) return HM_EALIGN;
    Axis2Plan q = axis2_plan(outer, a1, mid, a2, inner);
    q.f.weighted = std ? 1 : 0;
    double* partial = static_cast<double*>(workspace);
    hipStream_t st = as_stream(stream);
    launch_axis_stage1(val, std, q.outer1, q.A, q.inner1, q.p, partial, nullptr, nullptr, nullptr, 1, st);
    if (q.f.R * q.f.KS >= 64 && q.f.n_out2 <= 4096)
        hipLaunchKernelGGL(k_axis_final2_tree, dim3(static_cast<unsigned>(q.f.n_out2)), dim3(256), 0, st, partial, q.f, out_mean, out_std, out_err);
    else
        hipLaunchKernelGGL(k_axis_final2, dim3(stream_grid(q.f.n_out2, 256, 8)), dim3(256), 0, st, partial, q.f, out_mean, out_std, out_err);
    return launch_status();
}

extern "C" int hm_compute_difference_bcast(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                                           double* out_abs, double* out_abs_std, double* out_rel, double* out_rel_std,
                                           int ndim, const int64_t* shape, const int64_t* strides_x, const int64_t* strides_y, void* stream) {
    hm::Bcast2K b;
    int64_t n = 0;
    if (!hm::fill_bcast(b, ndim, shape, strides_x, strides_y, n)) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !y || !out_abs || !out_rel) return HM_EINVAL;
    const bool with_std = sx || sy;
    if (with_std != (out_abs_std != nullptr) || with_std != (out_rel_std != nullptr)) return HM_EINVAL;
    hipLaunchKernelGGL(hm::k_difference_bcast, dim3(hm::stream_grid(n, 256, 8)), dim3(256), 0, hm::as_stream(stream),
                       x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, n, b);
    return hm::launch_status();
}

extern "C" int hm_interpolate_bcast(const double* x0, const double* s0, const double* x1, const double* s1, double y0, double y1, double y,
                                    double* out, double* out_std, int ndim, const int64_t* shape, const int64_t* strides0,
                                    const int64_t* strides1, void* stream) {
    hm::Bcast2K b;
    int64_t n = 0;
    if (!hm::fill_bcast(b, ndim, shape, strides0, strides1, n)) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x0 || !x1 || !out || ((s0 || s1) != (out_std != nullptr))) return HM_EINVAL;
    hipLaunchKernelGGL(hm::k_interpolate_bcast, dim3(hm::stream_grid(n, 256, 8)), dim3(256), 0, hm::as_stream(stream),
                       x0, s0, x1, s1, y0, y1, y, out, out_std, n, b);
    return hm::launch_status();
}
