// hm_merge.hip - fused HDR merge for gfx950 (MI355X).
//
// Replaces, in one call, the two passes of the reference's merge loop
// (modules/exposure_series.py:317-345 `_precalculate_sum_of_weights` and :347-397
// `_compute_HDR_image_set`) together with the per-frame arithmetic they call:
// apply_gaussian_weight (modules/measurand.py:606-618), linearize (:471-541), the optional hot-pixel
// median (:543-557) and the optional flat-field normalisation (:559-604).
//
// The path is HBM-bandwidth bound (no MFMA): every input byte is read once with coalesced loads, every
// output byte written once. Per (element, frame) the only arithmetic is a table gather, because for
// 8-bit frames both the Gaussian weight and the ICRF depend on the DN alone; the 256-entry tables
// (w, w*g per channel, ...) live in LDS.
//
// Kernels (hm_merge dispatches; all share one operation sequence per output element, so a result does not depend on
// which kernel or tiling produced it):
//   merge_u8_val3                       uint8 frames, C == 3, N <= 20 (compile-time), val-only, no extras: the bench kernel (config 2).
//   merge_u8_fast / merge_u8_fast_std   the same frames with std, flat field or sum-of-weights output (config 3).
//   merge_u8_loop / merge_u8_loop_std   uint8 frames, run-time N (17..32) and C (1..4): same decomposition, frames in chunks.
//   merge_f64_val / merge_f64_std       float64 frames (64-bit mode): analytic weight, computed LUT index.
//   merge_generic     anything else (tails shorter than a group, unaligned tiles, forced by variant < 0):
//                     one element per thread.
//   merge_scan_hot + merge_patch_hot    dark-frame hot pixels: the streaming kernels never look at the dark maps; a scan reads every
//                     distinct map once (the d term of the algorithmic byte count) and queues the hot elements, a second kernel
//                     recomputes one queued element per lane with the k x k medians substituted (needs the caller's workspace).
//   merge_fixup_hot   the same pass without a workspace: one hot element per wave at a time (sparse maps only).
#include "hm_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>

#ifndef HM_TUNE_NF
#define HM_TUNE_NF 0      /* build with -DHM_TUNE_NF=7 to get the variant matrix for N = 7 */
#endif

namespace hm {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// kernel arguments (passed by value in the kernarg segment; all loads from it are scalar)
// ------------------------------------------------------------------------------------------------
struct MergeK {
    const void*    frame[HM_MAX_FRAMES];   // uint8 or float64 frames, at image row buf_row0
    const double*  sd[HM_MAX_FRAMES];      // float64 std frames (or null)
    const uint8_t* dark[HM_MAX_FRAMES];    // per-frame dark DN map (or null)
    double  inv_t[HM_MAX_FRAMES];          // 1 / exposure
    int32_t dark_min[HM_MAX_FRAMES];       // hot iff dark >= dark_min
    const double* icrf;
    const double* icrf_diff;
    const double* w_lut;
    const double* dw_lut;
    const uint8_t* flat_u8;                // at image row row0
    const double*  flat_f64;
    const double*  flat_std;
    double ff_mean[HM_MAX_CHANNELS];
    double ff_std_mean[HM_MAX_CHANNELS];
    double* out_val;                       // at image row row0
    double* out_std;
    double* out_sum_w;
    int64_t n_elems;                       // elements this launch covers (starting at elem0)
    int64_t elem0;                         // first element, relative to row0 (tail launches)
    int64_t in_off;                        // (row0 - buf_row0) * W * C: offset of row0 inside the input buffers
    int64_t H, W, row0, buf_row0, buf_rows;
    int32_t n_frames, C, median_k, has_flat;
    int32_t inv_t_inrange;                 // every 1/exposure in [2^-300, 2^300] (host check; see div_inrange)
    int32_t variant;
    uint32_t* hot_reset;                   // the hot-pixel queue's four counter words, zeroed by the streaming kernel that runs before the scan
};

// The queue's counters must be zero when merge_scan_hot starts. A hipMemsetAsync costs two fill kernels (~10 us per call, 1 % of
// config 3); the streaming kernel that precedes the scan in stream order does it in passing instead (one wave-uniform test per workgroup).
__device__ __forceinline__ void reset_hot_counters(const MergeK& a) {
    if (a.hot_reset && blockIdx.x == 0 && threadIdx.x < 4) a.hot_reset[threadIdx.x] = 0u;
}

__device__ __forceinline__ void elem_to_pixel(const MergeK& a, int64_t e, int64_t& row, int64_t& col, int& c) {
    const int64_t wc = a.W * a.C;
    row = a.row0 + e / wc;
    const int64_t rem = e % wc;
    col = rem / a.C;
    c = static_cast<int>(rem % a.C);
}

__device__ __forceinline__ void flat_field_apply(const MergeK& a, int64_t e, int c, bool with_std,
                                                 double& val, double& sd) {
    const double F = a.flat_u8 ? static_cast<double>(a.flat_u8[e]) / 255.0 : a.flat_f64[e];
    flat_field_math(F, with_std ? 1.0 / (F * F) : 1.0, with_std ? a.flat_std[e] : 0.0, a.ff_mean[c], a.ff_std_mean[c], with_std, val, sd);
}

// ------------------------------------------------------------------------------------------------
// One output element: the reference arithmetic (exposure_series.py:340-394) in the operation sequence
// shared by every kernel of this file, so a result does not depend on which kernel (or tiling) produced
// it: S = sum_i w_i in frame order; 1/S and 1/S**2 formed once; the numerator of :388 accumulated with
// fma and divided by S at the end; variance terms of :389 accumulated with fma.
// HOT = HOT_NONE: plain per-thread evaluation, dark maps ignored (streaming kernels).
// HOT = HOT_WAVE: `e` is wave-uniform, every lane evaluates the same element and hot frames take their value
//                 from wave_median(); `store` selects the one lane that writes.
// HOT = HOT_LANE: every lane evaluates its OWN element and takes the medians of its hot frames itself
//                 (lane_median(); the queue-driven patch kernel).
// Tables: t_w/t_dw [256], t_g/t_d [256*C] in LDS.
// ------------------------------------------------------------------------------------------------
enum { HOT_NONE = 0, HOT_WAVE = 1, HOT_LANE = 2 };
template <bool F64IN, bool STD, int HOT>
__device__ __forceinline__ void merge_one_element(const MergeK& a, const double* t_w, const double* t_dw,
                                                  const double* t_g, const double* t_d, int64_t e, bool store) {
    const int C = a.C, N = a.n_frames;
    const int64_t ei = a.in_off + e;
    const int c = static_cast<int>(e % C);
    int64_t row = 0, col = 0; int cc = 0;
    if (HOT != HOT_NONE) elem_to_pixel(a, e, row, col, cc);
    auto is_hot = [&](int i) -> bool { return HOT != HOT_NONE && a.dark[i] && static_cast<int>(a.dark[i][ei]) >= a.dark_min[i]; };
    auto median_of = [&](auto* f) {
        if constexpr (HOT == HOT_WAVE) return wave_median(f, a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
        else return lane_median(f, a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
    };
    auto value_f64 = [&](int i, bool hot) -> double {
        const double* f = static_cast<const double*>(a.frame[i]);
        if (HOT != HOT_NONE && hot) return median_of(f);
        return f[ei];
    };
    auto value_u8 = [&](int i, bool hot) -> uint32_t {
        const uint8_t* f = static_cast<const uint8_t*>(a.frame[i]);
        if (HOT != HOT_NONE && hot) return median_of(f);
        return f[ei];
    };
    // ---- pass 1: S = sum_i w_i (exposure_series.py:340) ----
    double S = 0.0;
    for (int i = 0; i < N; ++i) {
        const bool hot = is_hot(i);
        double w;
        if (F64IN) {
            const double dv = value_f64(i, hot) - 0.5;
            w = gauss_weight(dv);                                 // measurand.py:615
        } else {
            w = t_w[value_u8(i, hot)];
        }
        S = (i == 0) ? w : S + w;
    }
    if (a.out_sum_w && store) a.out_sum_w[e] = S;
    if (!a.out_val) return;
    const double invS = 1.0 / S;
    const double invS2 = 1.0 / (S * S);                                 // 1 / S**2, exposure_series.py:343
    // ---- pass 2: exposure_series.py:382-389 ----
    double acc = 0.0, var = 0.0;
    for (int i = 0; i < N; ++i) {
        const bool hot = is_hot(i);
        double w, dw;
        uint32_t idx;
        if (F64IN) {
            const double v = value_f64(i, hot);
            const double dv = v - 0.5;
            w = gauss_weight(dv);
            dw = (-60.0 * dv) * w;                                      // measurand.py:616
            idx = static_cast<uint32_t>(static_cast<int64_t>(rint(v * 255.0))) & 255u;   // measurand.py:503
        } else {
            idx = value_u8(i, hot);
            w = t_w[idx];
            dw = STD ? t_dw[idx] : 0.0;
        }
        const double g = t_g[idx * C + c];
        const double it = a.inv_t[i];
        const double wg = w * g;
        acc = (i == 0) ? wg * it : fma(wg, it, acc);                    // :388 numerator
        if (STD) {
            double s;
            if (HOT != HOT_NONE && hot) s = median_of(a.sd[i]);
            else s = a.sd[i][ei];
            const double dg = t_d[idx * C + c] * s;                     // measurand.py:512
            const double A = (dw * g + w * dg) * invS - ((dw * w) * g) * invS2;   // :389
            const double term = (A * dg) * it;
            var = (i == 0) ? term * term : fma(term, term, var);
        }
    }
    double val = acc / S;
    double sd = STD ? sqrt(var) : 0.0;                                  // :394
    if (a.has_flat) flat_field_apply(a, e, c, STD, val, sd);
    if (store) {
        a.out_val[e] = val;
        if (STD) a.out_std[e] = sd;
    }
}

template <bool F64IN, bool STD>
__device__ __forceinline__ void fill_plain_tables(const MergeK& a, double* t_w, double* t_dw, double* t_g, double* t_d) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        t_w[i] = (!F64IN) ? a.w_lut[i] : 0.0;
        t_dw[i] = (!F64IN && STD) ? a.dw_lut[i] : 0.0;
    }
    for (int i = threadIdx.x; i < 256 * a.C; i += blockDim.x) {
        t_g[i] = a.icrf[i];
        t_d[i] = STD ? a.icrf_diff[i] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// generic kernel: one element per thread, runtime N and C, uint8 or float64 frames.
// ------------------------------------------------------------------------------------------------
template <bool F64IN, bool STD>
__global__ __launch_bounds__(256) void merge_generic(const MergeK a) {
    __shared__ double t_w[256], t_dw[256], t_g[256 * HM_MAX_CHANNELS], t_d[256 * HM_MAX_CHANNELS];
    reset_hot_counters(a);
    fill_plain_tables<F64IN, STD>(a, t_w, t_dw, t_g, t_d);
    __syncthreads();
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < a.n_elems; q += stride)
        merge_one_element<F64IN, STD, HOT_NONE>(a, t_w, t_dw, t_g, t_d, a.elem0 + q, true);
}

// ------------------------------------------------------------------------------------------------
// hot-pixel fix-up.
// fixup_element(): one wave recomputes one output element `e` (wave-uniform) with the frames spread over
// the lanes - lane i < N owns frame i: its dark byte, its value and std are loaded in parallel (one memory
// latency instead of N); the k x k medians of ALL hot frames are taken at once by sub-groups of k*k lanes
// (floor(64 / k^2) frames per batch; ranks by ds_bpermute shuffles inside the sub-group); the per-frame
// terms are then formed in parallel and accumulated with readlane broadcasts IN FRAME ORDER, i.e. with
// exactly the operation sequence of merge_one_element(), so the fix-up is bit-identical to what a
// streaming kernel would have produced had it been given the filtered frames.
// ------------------------------------------------------------------------------------------------
template <bool F64IN, bool STD>
__device__ __forceinline__ void fixup_element(const MergeK& a, const double* t_w, const double* t_dw,
                                           const double* t_g, const double* t_d, int64_t e) {
    const int lane = threadIdx.x & 63;
    const int C = a.C, N = a.n_frames;
    const int64_t ei = a.in_off + e;
    const int c = static_cast<int>(e % C);
    int64_t row, col; int cc;
    elem_to_pixel(a, e, row, col, cc);
    const bool owner = lane < N;
    const int fi = owner ? lane : 0;
    // the lane's own frame: pointers picked with wave-uniform indices (a per-lane index into the kernarg
    // struct would make hipcc spill the whole struct to scratch for every thread of the kernel)
    const void* fr = a.frame[0];
    const double* sp = a.sd[0];
    const uint8_t* dk = a.dark[0];
    int dmin = a.dark_min[0];
    double it = a.inv_t[0];
    for (int i = 1; i < N; ++i) {
        const bool me = lane == i;
        fr = me ? a.frame[i] : fr;
        sp = me ? a.sd[i] : sp;
        dk = me ? a.dark[i] : dk;
        dmin = me ? a.dark_min[i] : dmin;
        it = me ? a.inv_t[i] : it;
    }
    // --- parallel loads of the lane's own frame
    const bool hot = owner && dk && static_cast<int>(dk[ei]) >= dmin;
    double v = F64IN ? static_cast<const double*>(fr)[ei] : static_cast<double>(static_cast<const uint8_t*>(fr)[ei]);
    double sdv = STD ? sp[ei] : 0.0;
    // --- medians of the hot frames, floor(64 / k^2) frames per batch
    const unsigned long long hotmask = __ballot(hot);
    const int k = a.median_k, kk = k * k, r = k / 2, m = kk / 2;
    const int B = 64 / kk;
    const int sub = lane / kk, p = lane % kk;
    for (int i0 = 0; i0 < N; i0 += B) {
        if (((hotmask >> i0) & ((B >= 64 ? ~0ull : ((1ull << B) - 1)))) == 0) continue;          // wave-uniform
        const int f = i0 + sub;
        const bool part = sub < B && f < N && ((hotmask >> f) & 1ull);
        const void* pf = a.frame[i0];
        const double* ps = a.sd[i0];
        for (int j = 1; j < B && i0 + j < N; ++j) {
            pf = sub == j ? a.frame[i0 + j] : pf;
            ps = sub == j ? a.sd[i0 + j] : ps;
        }
        const int64_t yy = reflect_index(row + (p / k - r), a.H) - a.buf_row0;
        const int64_t xx = reflect_index(col + (p % k - r), a.W);
        const int64_t ni = (yy * a.W + xx) * C + cc;
        double nv = 0.0, ns = 0.0;
        if (part) {
            nv = F64IN ? static_cast<const double*>(pf)[ni] : static_cast<double>(static_cast<const uint8_t*>(pf)[ni]);
            if (STD) ns = ps[ni];
        }
        int lv = 0, qv = 0, ls = 0, qs = 0;
        for (int q = 0; q < kk; ++q) {
            const int src = sub * kk + q;
            const double uv = __shfl(nv, src, 64);
            lv += (uv < nv); qv += (uv <= nv);
            if (STD) { const double us = __shfl(ns, src, 64); ls += (us < ns); qs += (us <= ns); }
        }
        const unsigned long long medv = __ballot(part && lv <= m && m < qv);
        const unsigned long long meds = STD ? __ballot(part && ls <= m && m < qs) : 0ull;
        // owner lane of frame f picks the median out of its sub-group
        const int osub = fi - i0;
        const bool mine = hot && osub >= 0 && osub < B;
        const unsigned long long gmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1);
        const int sh = mine ? osub * kk : 0;
        const unsigned long long gv = (medv >> sh) & gmask, gs = (meds >> sh) & gmask;
        const int srcv = sh + (gv ? __ffsll(static_cast<long long>(gv)) - 1 : 0);
        const int srcs = sh + (gs ? __ffsll(static_cast<long long>(gs)) - 1 : 0);
        const double mv = __shfl(nv, srcv, 64);
        const double ms = STD ? __shfl(ns, srcs, 64) : 0.0;
        if (mine) { v = mv; if (STD) sdv = ms; }
    }
    // --- per-frame quantities, in parallel
    double w = 0.0, dw = 0.0;
    uint32_t idx = 0;
    if (owner) {
        if (F64IN) {
            const double dv = v - 0.5;
            w = gauss_weight(dv);
            dw = (-60.0 * dv) * w;
            idx = static_cast<uint32_t>(static_cast<int64_t>(rint(v * 255.0))) & 255u;
        } else {
            idx = static_cast<uint32_t>(v);
            w = t_w[idx];
            dw = STD ? t_dw[idx] : 0.0;
        }
    }
    // --- S in frame order
    double S = 0.0;
    for (int i = 0; i < N; ++i) {
        const double wi = readlane_f64(w, i);
        S = (i == 0) ? wi : S + wi;
    }
    if (a.out_sum_w && lane == 0) a.out_sum_w[e] = S;
    if (!a.out_val) return;
    const double invS = 1.0 / S;
    const double invS2 = 1.0 / (S * S);
    const double g = t_g[idx * C + c];
    const double wg = w * g;
    double term = 0.0;
    if (STD) {
        const double dg = t_d[idx * C + c] * sdv;
        const double A = (dw * g + w * dg) * invS - ((dw * w) * g) * invS2;
        term = (A * dg) * it;
    }
    double acc = 0.0, var = 0.0;
    for (int i = 0; i < N; ++i) {
        const double wgi = readlane_f64(wg, i), iti = readlane_f64(it, i);
        acc = (i == 0) ? wgi * iti : fma(wgi, iti, acc);
        if (STD) {
            const double ti = readlane_f64(term, i);
            var = (i == 0) ? ti * ti : fma(ti, ti, var);
        }
    }
    double val = acc / S;
    double sd = STD ? sqrt(var) : 0.0;
    if (a.has_flat) flat_field_apply(a, e, c, STD, val, sd);
    if (lane == 0) {
        a.out_val[e] = val;
        if (STD) a.out_std[e] = sd;
    }
}

// hot bits of the lane's 16-element chunk: bit b set iff element e0 + b is hot in at least one frame. Every DISTINCT
// (dark map, threshold) is read once - one 16-byte load when the address allows (the d * N term of the algorithmic bytes).
__device__ __forceinline__ uint32_t scan_chunk_hotbits(const MergeK& a, int64_t e0, int cnt) {
    uint32_t hotbits = 0;
    for (int i = 0; i < a.n_frames; ++i) {
        const uint8_t* d = a.dark[i];
        if (!d) continue;
        bool seen = false;                                          // same map + threshold as an earlier frame
        for (int k = 0; k < i; ++k) seen = seen || (a.dark[k] == d && a.dark_min[k] == a.dark_min[i]);
        if (seen) continue;
        const uint8_t* p = d + a.in_off + e0;
        const uint32_t thr = static_cast<uint32_t>(a.dark_min[i]);
        if (cnt == 16 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) hotbits |= (((w4[b >> 2] >> (8 * (b & 3))) & 255u) >= thr) ? (1u << b) : 0u;
        } else {
            for (int b = 0; b < cnt; ++b) hotbits |= (static_cast<uint32_t>(p[b]) >= thr) ? (1u << b) : 0u;
        }
    }
    return hotbits;
}

// Hot-element queue in the caller's workspace (hm_merge_args.hot_workspace), uint32 words:
//   ws[0] = number of queued elements, ws[1] = overflow flag (the queue was too small: merge_patch_hot goes over the whole tile),
//   ws[2] = number of pieces the scan wrote (its workgroups x rounds), ws[3] unused;
//   ws[4 .. 4 + 2 T)  piece table, T = hot_piece_slots(n_elems): {queue offset, count} of piece p - a piece is what one scan workgroup
//                     queued in one round, 65 536 consecutive elements of the image, and p counts them in IMAGE order; the pieces land in
//                     the queue in the order of their atomics, the table lets the patch kernel walk them in image order;
//   ws[4 + 2 T ...]   element indices relative to row0 (uint32: the queue path requires fewer than 2^32 elements per call).
constexpr int kHotQueueHeader = 4;       // uint32 words before the piece table (16 bytes)
constexpr int kHotPiecesLds = 2048;      // pieces the patch kernel can order in LDS (134 M elements per call); beyond that it walks the queue as it lies
__host__ __device__ inline uint32_t hot_piece_slots(int64_t n_elems) { return static_cast<uint32_t>(n_elems / 65536 + 1 + 1024); }

// merge_fixup_hot: every lane scans 16 consecutive elements of every distinct dark map; elements with at least one hot
// frame are recomputed, one at a time, by the wave. This is the path WITHOUT a workspace: one element per wave at a time
// makes it collapse on dense maps (15 us of one wave per hot element) - callers that can pass a workspace should.
template <bool F64IN, bool STD>
__global__ __launch_bounds__(256) void merge_fixup_hot(const MergeK a) {
    __shared__ double t_w[256], t_dw[256], t_g[256 * HM_MAX_CHANNELS], t_d[256 * HM_MAX_CHANNELS];
    fill_plain_tables<F64IN, STD>(a, t_w, t_dw, t_g, t_d);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t n_chunks = (a.n_elems + 15) / 16;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    // chunks are dealt 64 at a time to a wave, so the trip count is wave-uniform
    for (int64_t cb = wave0 * 64; cb < n_chunks; cb += n_waves * 64) {
        const int64_t chunk = cb + lane;
        const int64_t e0 = a.elem0 + chunk * 16;                            // relative to row0
        uint32_t hotbits = 0;
        if (chunk < n_chunks) {
            const int64_t left = a.elem0 + a.n_elems - e0;
            hotbits = scan_chunk_hotbits(a, e0, left < 16 ? static_cast<int>(left) : 16);
        }
        unsigned long long pending = __ballot(hotbits != 0);
        while (pending) {                                                    // rare
            const int src = __ffsll(static_cast<long long>(pending)) - 1;
            pending &= pending - 1;
            uint32_t bits = __builtin_amdgcn_readlane(hotbits, src);
            const int64_t base = a.elem0 + (cb + src) * 16;
            while (bits) {
                const int b = __ffs(static_cast<int>(bits)) - 1;
                bits &= bits - 1;
                fixup_element<F64IN, STD>(a, t_w, t_dw, t_g, t_d, base + b);
            }
        }
    }
}

// merge_scan_hot: the scan alone. A wave scans kScanRound consecutive spans of 1024 elements per round (16 bytes per lane, map and
// span, the round's loads of a map issued together), a workgroup's 16 waves 64 consecutive spans = 65 536 elements (a few image
// rows: what lands next to each other in the queue shares its neighbour rows, which the patch kernel then finds in the cache).
// The waves add up their counts through LDS and ONE atomic add per workgroup
// and round reserves its piece of the queue (an atomic per wave and span serialised on the counter: 49 000 returning atomics on
// one address took 365 us at a density of 1e-3, profiles/r03c_hot_trace.txt). Every wave runs the same number of rounds (barriers).
// The order of the pieces in the queue depends on the order of the atomics; the patched image does not (every queued element is
// recomputed independently).
constexpr int kScanRound = 4;
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void merge_scan_hot(const MergeK a, uint32_t* ws, uint32_t capacity) {
    uint32_t* const table = ws + kHotQueueHeader;
    uint32_t* const queue = table + 2u * hot_piece_slots(a.n_elems);
    __shared__ uint32_t s_tot[kScanBlock / 64];
    __shared__ uint32_t s_base, s_ok;
    constexpr uint32_t WPB = kScanBlock / 64;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n_chunks = (a.n_elems + 15) / 16;
    const int64_t n_spans = (n_chunks + 63) / 64;
    const int64_t n_waves = static_cast<int64_t>(gridDim.x) * WPB;
    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * WPB + wave;
    const int64_t rounds = (n_spans + n_waves * kScanRound - 1) / (n_waves * kScanRound);
    // distinct (map, threshold) pairs: bit i set = frame i's map is read (wave-uniform, once); `wide` = every such map can be read with
    // aligned 16-byte loads (its pointer + in_off is 16-byte aligned: elem0 and the chunk starts are multiples of 16)
    uint32_t distinct = 0;
    bool wide = (a.elem0 & 15) == 0;
    for (int i = 0; i < a.n_frames; ++i) {
        const uint8_t* d = a.dark[i];
        if (!d) continue;
        bool seen = false;
        for (int k = 0; k < i; ++k) seen = seen || (a.dark[k] == d && a.dark_min[k] == a.dark_min[i]);
        if (seen) continue;
        distinct |= 1u << i;
        wide = wide && (reinterpret_cast<uintptr_t>(d + a.in_off) & 15) == 0;
    }
    const int64_t n_full = a.n_elems / 16;                                  // whole 16-element chunks; a last partial one goes the slow way
    for (int64_t r = 0; r < rounds; ++r) {
        uint32_t hb[kScanRound];
        uint32_t mine = 0;
        if (wide) {
            // no load sits behind a lane-dependent branch: the kScanRound loads of a map are issued back to back (a chunk past the end
            // re-reads chunk 0 and is masked afterwards), then compared
            int64_t off[kScanRound];
            bool valid[kScanRound];
#pragma unroll
            for (int k = 0; k < kScanRound; ++k) {
                const int64_t chunk = (((r * n_waves + wave_global) * kScanRound) + k) * 64 + lane;
                valid[k] = chunk < n_full;
                off[k] = a.in_off + a.elem0 + (valid[k] ? chunk : 0) * 16;
                hb[k] = 0;
            }
            if (n_full > 0) {
                for (uint32_t left = distinct; left; left &= left - 1) {    // wave-uniform loop over the distinct maps
                    const int i = __builtin_amdgcn_readfirstlane(__ffs(static_cast<int>(left)) - 1);
                    const uint8_t* d = a.dark[i];
                    const uint32_t thr = static_cast<uint32_t>(a.dark_min[i]);
                    u32x4 v[kScanRound];
#pragma unroll
                    for (int k = 0; k < kScanRound; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(d + off[k]));
#pragma unroll
                    for (int k = 0; k < kScanRound; ++k) {
                        const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                        for (int b = 0; b < 16; ++b) hb[k] |= (((w4[b >> 2] >> (8 * (b & 3))) & 255u) >= thr) ? (1u << b) : 0u;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kScanRound; ++k) {
                const int64_t chunk = (((r * n_waves + wave_global) * kScanRound) + k) * 64 + lane;
                if (!valid[k]) hb[k] = 0;
                if (chunk == n_full && chunk < n_chunks)                    // the tile's last, partial chunk (one lane of one wave)
                    hb[k] = scan_chunk_hotbits(a, a.elem0 + chunk * 16, static_cast<int>(a.n_elems - chunk * 16));
                mine += static_cast<uint32_t>(__popc(hb[k]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < kScanRound; ++k) {
                const int64_t chunk = (((r * n_waves + wave_global) * kScanRound) + k) * 64 + lane;
                hb[k] = 0;
                if (chunk < n_chunks) {
                    const int64_t e0 = a.elem0 + chunk * 16;
                    const int64_t left = a.elem0 + a.n_elems - e0;
                    hb[k] = scan_chunk_hotbits(a, e0, left < 16 ? static_cast<int>(left) : 16);
                }
                mine += static_cast<uint32_t>(__popc(hb[k]));
            }
        }
        uint32_t incl = mine;                                               // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= static_cast<uint32_t>(d)) incl += up;
        }
        if (lane == 63u) s_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (uint32_t w = 0; w < WPB; ++w) total += s_tot[w];
            uint32_t base = 0, ok = 1;
            if (total) {
                base = atomicAdd(&ws[0], total);
                if (base + total > capacity || base + total < base) {       // queue full: flag it, merge_patch_hot goes over the whole tile instead
                    atomicOr(&ws[1], 1u);
                    ok = 0;
                }
            }
            s_base = base; s_ok = ok;
            const uint32_t piece = static_cast<uint32_t>(r * gridDim.x + blockIdx.x);          // image order
            table[2u * piece] = ok ? base : 0u;
            table[2u * piece + 1u] = ok ? total : 0u;
            if (blockIdx.x == 0 && r == 0) ws[2] = static_cast<uint32_t>(rounds * gridDim.x);
        }
        __syncthreads();
        if (s_ok && mine) {
            uint32_t off = s_base + (incl - mine);
            for (uint32_t w = 0; w < wave; ++w) off += s_tot[w];
            uint32_t* q = queue + off;
#pragma unroll
            for (int k = 0; k < kScanRound; ++k) {
                uint32_t bits = hb[k];
                const int64_t chunk = (((r * n_waves + wave_global) * kScanRound) + k) * 64 + lane;
                const uint32_t first = static_cast<uint32_t>(a.elem0 + chunk * 16);   // < 2^32 (host check)
                while (bits) {
                    const int b = __ffs(static_cast<int>(bits)) - 1;
                    bits &= bits - 1;
                    *q++ = first + static_cast<uint32_t>(b);
                }
            }
        }
        __syncthreads();                                                    // s_tot / s_base are rewritten in the next round
    }
}

// ------------------------------------------------------------------------------------------------
// Per-LANE hot-pixel patch: every lane recomputes its own hot element, so up to 64 elements are in flight per wave and their
// memory round trips overlap. patch_element_k3() is merge_one_element()'s operation sequence written so that the dependent
// rounds are few: the dark bytes of 8 frames at a time, then - for kPatchFB frames at a time - ALL nine neighbours of every frame
// with select-ed addresses (a frame that is not hot reads its own element nine times: no branch sits between the loads and
// the compiler's wait counts stay exact), the 19-exchange median network, and the frames' arithmetic in frame order.
// Pass 1 leaves every frame's (filtered) value in a per-thread LDS column for pass 2; the std medians are taken in pass 2.
// Other kernel sizes (5, 7) and float64 stacks of more than kPatchF64Frames frames go through merge_one_element<HOT_LANE>.
// ------------------------------------------------------------------------------------------------
#ifndef HM_PATCH_HOT_ONLY
#define HM_PATCH_HOT_ONLY 1             // 1: neighbour loads only in the lanes (and waves) where the frame is hot; 0: select-ed addresses, no branch
#endif
constexpr int kPatchFB = 4;             // frames whose nine neighbour loads are issued together (float64 values / stds)
constexpr int kPatchFBU8 = 8;           // the same for uint8 values (one VGPR each)
constexpr int kPatchF64Frames = 16;     // float64 stacks: frames kept in LDS between the passes (8 B x 256 threads each)

__device__ __forceinline__ uint32_t lane_hotmask(const MergeK& a, int64_t ei) {
    uint32_t hotmask = 0;
    const int N = a.n_frames;
    for (int i0 = 0; i0 < N; i0 += 8) {
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            d[k] = 0;
            if (i0 + k < N && a.dark[i0 + k]) d[k] = a.dark[i0 + k][ei];                 // wave-uniform condition
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < N && a.dark[i0 + k]) hotmask |= (static_cast<int>(d[k]) >= a.dark_min[i0 + k]) ? (1u << (i0 + k)) : 0u;
    }
    return hotmask;
}

template <bool F64IN, bool STD>
__device__ __forceinline__ void patch_element_k3(const MergeK& a, const double* t_w, const double* t_dw, const double* t_g, const double* t_d,
                                                 std::conditional_t<F64IN, double, uint8_t>* keep /* LDS, this thread's column, stride 256 */,
                                                 int64_t e, uint32_t hotmask) {
    using T = std::conditional_t<F64IN, double, uint8_t>;
    using V = typename MedianKey<T>::type;
    const int C = a.C, N = a.n_frames;
    const int64_t ei = a.in_off + e;
    const int c = static_cast<int>(e % C);
    int64_t row, col; int cc;
    elem_to_pixel(a, e, row, col, cc);
    const int64_t wc = a.W * C;
    const int64_t dy[3] = {(reflect_index(row - 1, a.H) - row) * wc, 0, (reflect_index(row + 1, a.H) - row) * wc};
    const int64_t dx[3] = {(reflect_index(col - 1, a.W) - col) * C, 0, (reflect_index(col + 1, a.W) - col) * C};
    // ---- pass 1: filtered values, S = sum_i w_i (exposure_series.py:340)
    constexpr int FBV = F64IN ? kPatchFB : kPatchFBU8;
    double S = 0.0;
    for (int i0 = 0; i0 < N; i0 += FBV) {
        V p[FBV][9];
#pragma unroll
        for (int f = 0; f < FBV; ++f) {
            if (i0 + f < N) {
                const T* fr = static_cast<const T*>(a.frame[i0 + f]) + ei;
                const bool hot = (hotmask >> (i0 + f)) & 1u;
#if HM_PATCH_HOT_ONLY
                // the frame's own element for every lane; its eight neighbours only in the lanes where THIS frame is hot (a wave in
                // which it is hot nowhere skips them): with one map per frame an element is typically hot in one frame of N
                const V centre = static_cast<V>(fr[0]);
#pragma unroll
                for (int q = 0; q < 9; ++q) p[f][q] = centre;
                if (hot) {
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        if (q != 4) p[f][q] = static_cast<V>(fr[dy[q / 3] + dx[q % 3]]);
                }
#else
#pragma unroll
                for (int q = 0; q < 9; ++q) p[f][q] = static_cast<V>(fr[hot ? dy[q / 3] + dx[q % 3] : int64_t{0}]);
#endif
            }
        }
#pragma unroll
        for (int f = 0; f < FBV; ++f) {
            if (i0 + f < N) {
                const V v = median9<T>(p[f]);
                keep[(i0 + f) * 256] = static_cast<T>(v);
                double w;
                if constexpr (F64IN) w = gauss_weight(v - 0.5);                          // measurand.py:615
                else w = t_w[v];
                S = (i0 + f == 0) ? w : S + w;
            }
        }
    }
    if (a.out_sum_w) a.out_sum_w[e] = S;
    if (!a.out_val) return;
    const double invS = 1.0 / S;
    const double invS2 = 1.0 / (S * S);                                                  // 1 / S**2, exposure_series.py:343
    // ---- pass 2: exposure_series.py:382-389
    double acc = 0.0, var = 0.0;
    for (int i0 = 0; i0 < N; i0 += kPatchFB) {
        double sp[STD ? kPatchFB : 1][9];
        if constexpr (STD) {
#pragma unroll
            for (int f = 0; f < kPatchFB; ++f) {
                if (i0 + f < N) {
                    const double* sr = a.sd[i0 + f] + ei;
                    const bool hot = (hotmask >> (i0 + f)) & 1u;
#if HM_PATCH_HOT_ONLY
                    const double centre = sr[0];
#pragma unroll
                    for (int q = 0; q < 9; ++q) sp[f][q] = centre;
                    if (hot) {
#pragma unroll
                        for (int q = 0; q < 9; ++q)
                            if (q != 4) sp[f][q] = sr[dy[q / 3] + dx[q % 3]];
                    }
#else
#pragma unroll
                    for (int q = 0; q < 9; ++q) sp[f][q] = sr[hot ? dy[q / 3] + dx[q % 3] : int64_t{0}];
#endif
                }
            }
        }
#pragma unroll
        for (int f = 0; f < kPatchFB; ++f) {
            if (i0 + f < N) {
                const int i = i0 + f;
                double w, dw;
                uint32_t idx;
                if constexpr (F64IN) {
                    const double v = keep[i * 256];
                    const double dv = v - 0.5;
                    w = gauss_weight(dv);
                    dw = (-60.0 * dv) * w;                                               // measurand.py:616
                    idx = static_cast<uint32_t>(static_cast<int64_t>(rint(v * 255.0))) & 255u;   // measurand.py:503
                } else {
                    idx = keep[i * 256];
                    w = t_w[idx];
                    dw = STD ? t_dw[idx] : 0.0;
                }
                const double g = t_g[idx * C + c];
                const double it = a.inv_t[i];
                const double wg = w * g;
                acc = (i == 0) ? wg * it : fma(wg, it, acc);                             // :388 numerator
                if constexpr (STD) {
                    const double sv = median9<double>(sp[f]);
                    const double dg = t_d[idx * C + c] * sv;                             // measurand.py:512
                    const double A = (dw * g + w * dg) * invS - ((dw * w) * g) * invS2;  // :389
                    const double term = (A * dg) * it;
                    var = (i == 0) ? term * term : fma(term, term, var);
                }
            }
        }
    }
    double val = acc / S;
    double sd = STD ? sqrt(var) : 0.0;                                                   // :394
    if (a.has_flat) flat_field_apply(a, e, c, STD, val, sd);
    a.out_val[e] = val;
    if (STD) a.out_std[e] = sd;
}

template <bool F64IN, bool STD>
__device__ __forceinline__ void patch_element(const MergeK& a, const double* t_w, const double* t_dw, const double* t_g, const double* t_d,
                                              char* keep_lds, int64_t e, uint32_t hotmask) {
    using T = std::conditional_t<F64IN, double, uint8_t>;
    if (a.median_k == 3 && (!F64IN || a.n_frames <= kPatchF64Frames))                    // wave-uniform
        patch_element_k3<F64IN, STD>(a, t_w, t_dw, t_g, t_d, reinterpret_cast<T*>(keep_lds) + threadIdx.x, e, hotmask);
    else
        merge_one_element<F64IN, STD, HOT_LANE>(a, t_w, t_dw, t_g, t_d, e, true);
}
static int patch_keep_bytes(bool f64in, int n_frames, int median_k) {
    if (median_k != 3 || (f64in && n_frames > kPatchF64Frames)) return 16;
    return n_frames * 256 * (f64in ? 8 : 1);
}

// merge_patch_hot: the queue's elements, one per lane. The grid is fixed (the host does not know the count), so the entries are
// dealt to ALL its waves in blocks of B = ceil(count / waves) <= 64 consecutive entries: a short queue is spread over the
// whole chip (one or a few active lanes per wave: the latency of one element, not of 64), a long one fills the waves (and
// consecutive lanes take consecutive entries, which a wave's span put there in element order, so neighbouring lanes touch
// neighbouring cache lines). Workgroups beyond the queue leave before they build their tables.
// If the scan raised the overflow flag (more hot elements than the workspace holds: a quarter of the image with the recommended
// size) the queue is ignored: every lane looks at its own elements' dark bytes and patches the hot ones.
template <bool F64IN, bool STD>
__global__ __launch_bounds__(256) void merge_patch_hot(const MergeK a, const uint32_t* ws, uint32_t bmax) {
    __shared__ double t_w[256], t_dw[256], t_g[256 * HM_MAX_CHANNELS], t_d[256 * HM_MAX_CHANNELS];
    extern __shared__ __attribute__((aligned(16))) char keep_lds[];
    const uint32_t count = __builtin_amdgcn_readfirstlane(ws[0]);
    const uint32_t overflow = __builtin_amdgcn_readfirstlane(ws[1]);
    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t B = (count + n_waves - 1u) / n_waves;
    B = B > bmax ? bmax : B;
    if (overflow == 0u && count == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // Blocks of B consecutive entries - consecutive in IMAGE order: the pieces are walked through the scan's piece table - are dealt
    // iteration by iteration: in iteration `it` the logical workgroup l takes blocks (it * G + l) * 4 + wave, and workgroups that run on
    // the same XCD (blockIdx % 8: round-robin dispatch) are neighbours in l. At any moment one XCD works on ONE run of image rows, and
    // the lines its hot elements share (rows r - 1, r, r + 1 of every frame) are requested from one L2 at about the same time: at a
    // density of 5e-2 a workgroup that owned a contiguous part of the queue over all its iterations fetched every line 2.3 times
    // (profiles/r03e_hot5e2_patch_rocprof_summary.md; 2 270 -> 1 820 us per merge with the interleaved order, r03m_ab_patch_order.log).
    // HM_PATCH_ORDER 1 = that first mapping.
#ifndef HM_PATCH_ORDER
#define HM_PATCH_ORDER 0
#endif
    const uint32_t* const table = ws + kHotQueueHeader;
    const uint32_t* const queue = table + 2u * hot_piece_slots(a.n_elems);
    const uint32_t iters = B ? (count + n_waves * B - 1u) / (n_waves * B) : 0u;
    const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
    const uint32_t logical = gridDim.x % 8u == 0u ? (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : blockIdx.x;
    auto entry_of = [&](uint32_t it) -> uint64_t {
        if (HM_PATCH_ORDER == 1) return static_cast<uint64_t>(logical) * 4u * B * iters + (static_cast<uint64_t>(it) * 4u + wave) * B + lane;
        return ((static_cast<uint64_t>(it) * gridDim.x + logical) * 4u + wave) * B + lane;
    };
    if (overflow == 0u && static_cast<uint64_t>(logical) * 4u * B * (HM_PATCH_ORDER == 1 ? iters : 1u) >= count) return;
    // image-order position -> queue slot: exclusive prefix of the piece counts in LDS (one piece = 8 bytes of table per 65 536 elements)
    __shared__ uint32_t s_pref[kHotPiecesLds + 1], s_off[kHotPiecesLds], s_wsum[4];
    const uint32_t n_pieces = __builtin_amdgcn_readfirstlane(ws[2]);
    const bool ordered = overflow == 0u && n_pieces <= static_cast<uint32_t>(kHotPiecesLds) && HM_PATCH_ORDER == 0;
    if (ordered) {
        constexpr uint32_t PER = kHotPiecesLds / 256;
        uint32_t cnt[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t pc = threadIdx.x * PER + k;
            cnt[k] = pc < n_pieces ? table[2u * pc + 1u] : 0u;
            if (pc < n_pieces) s_off[pc] = table[2u * pc];
            sum += cnt[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= static_cast<uint32_t>(d)) incl += up;
        }
        if (lane == 63u) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t pc = threadIdx.x * PER + k;
            if (pc <= n_pieces) s_pref[pc] = run;
            run += cnt[k];
        }
        if (threadIdx.x == 255) s_pref[kHotPiecesLds] = run;
        __syncthreads();
    }
    auto slot_of = [&](uint64_t pos) -> uint64_t {                  // pos < count
        if (!ordered) return pos;
        uint32_t lo = 0, hi = n_pieces;                               // the piece p with s_pref[p] <= pos < s_pref[p + 1]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pref[mid] <= static_cast<uint32_t>(pos)) lo = mid; else hi = mid;
        }
        return static_cast<uint64_t>(s_off[lo]) + (static_cast<uint32_t>(pos) - s_pref[lo]);
    };
    // the first entry and its dark bytes are fetched while the workgroup builds its tables (a short queue is one entry per lane:
    // its latency is the kernel's duration)
    int64_t e_first = 0;
    uint32_t hot_first = 0;
    if (overflow == 0u && lane < B && entry_of(0) < count) {
        e_first = static_cast<int64_t>(queue[slot_of(entry_of(0))]);
        hot_first = lane_hotmask(a, a.in_off + e_first);
    }
    fill_plain_tables<F64IN, STD>(a, t_w, t_dw, t_g, t_d);
    __syncthreads();
    if (overflow != 0u) {
        const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
        for (int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; q < a.n_elems; q += stride) {
            const int64_t e = a.elem0 + q;
            const uint32_t hotmask = lane_hotmask(a, a.in_off + e);
            if (hotmask) patch_element<F64IN, STD>(a, t_w, t_dw, t_g, t_d, keep_lds, e, hotmask);
        }
        return;
    }
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t q = entry_of(it);
        if (lane < B && q < count) {
            int64_t e = e_first;
            uint32_t hotmask = hot_first;
            if (it != 0u) {
                e = static_cast<int64_t>(queue[slot_of(q)]);
                hotmask = lane_hotmask(a, a.in_off + e);
            }
            patch_element<F64IN, STD>(a, t_w, t_dw, t_g, t_d, keep_lds, e, hotmask);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fast kernel: uint8 frames, C == 3, compile-time N.
// ------------------------------------------------------------------------------------------------
// LDS table layouts of the val-only kernel:
//   TAB_PLAIN   w[256] | wg[256*3]  (8-byte entries, two ds_read_b64 per (element, frame))        8 KB
//   TAB_FUSED   {w, wg}[256*3] as 16-byte entries, one ds_read_b128 per (element, frame)          12 KB
//   TAB_NONE    (tuning probe only, wrong results) no gather at all: the HBM ceiling of the access pattern
// Replicating the tables across banks was measured and dropped (docs/DESIGN_history_r01_r02.md 4.4): only a full private copy
// per lane of an LDS lane group removes the ~2.4-way conflicts random DNs cause, and that needs 64 KB per
// 256-entry float64 table, i.e. one workgroup per CU, which loses more latency hiding than it gains.
// The std kernel uses {w,dw}[256] and {g,d}[768] as 16-byte entries (16 KB).
enum { TAB_PLAIN = 0, TAB_FUSED = 1, TAB_NONE = 5 };

template <int TAB> struct TabInfo;
template <> struct TabInfo<TAB_PLAIN> { static constexpr int bytes = 8 * 256 * 4; };
template <> struct TabInfo<TAB_FUSED> { static constexpr int bytes = 16 * 768; };
template <> struct TabInfo<TAB_NONE>  { static constexpr int bytes = 16; };
constexpr int kStdTabBytes = 16 * 256 + 16 * 768;

__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p));
}
// the same load through an explicit global-address-space pointer: after an `asm volatile("+s")` pin the compiler no longer
// knows that a kernarg pointer is global and would fall back to flat_load (which also counts against lgkmcnt)
// 16-byte nontemporal load at (scalar base) + (32-bit byte offset in a VGPR) through an explicit global-address-space pointer:
// after an `asm volatile("+s")` pin hipcc no longer knows that a kernarg pointer is global and emits flat_load, which
// counts against lgkmcnt as well as vmcnt and so serialises with the LDS gathers around it.
typedef const __attribute__((address_space(1))) f64x2* global_f64x2_ptr;
typedef const __attribute__((address_space(1))) char* global_char_ptr;
__device__ __forceinline__ f64x2 ld_f64x2_global(const double* scalar_base, uint32_t byte_off) {
    global_char_ptr g = (global_char_ptr)(scalar_base);
    return __builtin_nontemporal_load((global_f64x2_ptr)(g + byte_off));
}

// Buffer addressing for the frame bytes: the frame pointer sits in a 4-SGPR resource descriptor, the group's byte offset
// in one SGPR shared by all frames (soffset), the lane's offset in one VGPR and the sub-unit in the immediate - a load
// costs no address arithmetic at all (global_load needs a 64-bit VGPR address or a 64-bit scalar base per frame and group).
// Raw buffer, stride 0, num_records = 2^32 - 1 bytes: offsets are below 2^32 because the fast kernels require E < 2^32.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffff, 0x00020000);
}
constexpr int kAuxNT = 2;                                             // cache-policy bits of the raw-buffer builtins on gfx94x/gfx950: nt
__device__ __forceinline__ uint16_t ld_u16_buf(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t group_off) {
    return __builtin_amdgcn_raw_buffer_load_b16(r, static_cast<int>(lane_off), static_cast<int>(group_off), kAuxNT);
}
// store two float64 at (scalar base) + (32-bit byte offset in a VGPR)
__device__ __forceinline__ void store2(double* base, uint32_t byte_off, double x, double y) {
    f64x2 v; v.x = x; v.y = y;
    __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(reinterpret_cast<char*>(base) + byte_off));
}

template <int TAB>
__device__ __forceinline__ void fill_val_tables(char* lds, const MergeK& a) {
    constexpr int C = 3;
    if constexpr (TAB == TAB_NONE) return;
    for (int q = threadIdx.x; q < 256 * C; q += blockDim.x) {
        const int dn = q / C;
        const double w = a.w_lut[dn];
        const double wg = w * a.icrf[q];             // (w * g), exposure_series.py:388
        if constexpr (TAB == TAB_PLAIN) {
            double* t = reinterpret_cast<double*>(lds);
            if (q % C == 0) t[dn] = w;
            t[256 + q] = wg;
        } else {
            reinterpret_cast<double2*>(lds)[q] = double2{w, wg};
        }
    }
}

template <int TAB>
__device__ __forceinline__ uint32_t chan_off(uint32_t c) {
    if constexpr (TAB == TAB_PLAIN) return 2048u + c * 8u;
    if constexpr (TAB == TAB_FUSED) return c * 16u;
    return c;
}
template <int TAB>
__device__ __forceinline__ void gather_val(const char* lds, uint32_t dn, uint32_t coff, double& w, double& wg) {
    if constexpr (TAB == TAB_NONE) {
        w = static_cast<double>(dn + 1u); wg = static_cast<double>(dn + coff);
    } else if constexpr (TAB == TAB_PLAIN) {
        w = *reinterpret_cast<const double*>(lds + dn * 8u);
        wg = *reinterpret_cast<const double*>(lds + dn * 24u + coff);
    } else {
        const double2 t = *reinterpret_cast<const double2*>(lds + dn * 48u + coff);
        w = t.x; wg = t.y;
    }
}

// Work decomposition of the fast kernel (tools/membench2.hip and tools/mergelab.hip are the experiments
// behind it): a wave owns "groups" of U * 128 consecutive elements. In a 128-element sub-unit lane l
// handles elements 2l and 2l+1:
//   * input: one global_load_ushort per (frame, sub-unit), 128 contiguous bytes per wave instruction,
//     immediate offsets 128*s off one scalar base;
//   * output: lane l owns the 16 contiguous bytes of elements 2l, 2l+1, so every store instruction is
//     one fully contiguous 1 KB global_store_dwordx4 (nontemporal) - no cross-lane transposition.
//     (Store instructions must cover whole 128-byte lines: 4 elements per lane stored directly is a
//     32-byte-stride pattern that costs 146 us instead of 128 us for the traffic alone; transposing
//     through LDS fixed that but added 13 % to an LDS pipe that bank conflicts already fill.)
// All group-level address arithmetic is scalar (the group index is wave-uniform). All N*U loads of a
// group are issued before the first gather; PREFETCH issues the next group's loads first.
// Channel of element 2l + j of sub-unit s of group g: (g*U*128 + 128*s + 2l + j) % 3 = (2*(g*U + s + l) + j) % 3.
//
// FLAT / SUMW = flat-field epilogue / sum-of-weights output compiled in (separate instantiations: a runtime
// branch around the flat-field loads cost config 3 ~100 us).

// keeps an accumulator chain where the source puts it (hipcc otherwise sinks the second element's chain
// below the first element's epilogue and keeps every gathered value live until then)
#define HM_PIN(x) asm volatile("" : "+v"(x))

constexpr uint32_t kSub = 128;     // elements per sub-unit (64 lanes x 2)
#ifndef HM_FB
#define HM_FB 4
#endif
#ifndef HM_STD_FB
#define HM_STD_FB 2      // std kernel, pass 2: frames per scheduling bundle
#endif
#ifndef HM_VAL3_FLAT_U
#define HM_VAL3_FLAT_U 2     // sub-units per iteration of merge_u8_val3's flat-field instantiation (N <= 8): 152 us at 134 VGPRs against 156 us at 176 VGPRs with U = 4
#endif
#ifndef HM_F64_KEEP_W
#define HM_F64_KEEP_W 1      // float64-frame std kernel: keep pass 1's weights in registers for pass 2 (N <= 8): one exp() per
#endif                       // element-frame instead of two; with 3 waves/SIMD 1 385 -> 1 296 us on 7 x 4096 x 4096 x 3 (profiles/r02c_ab_f64std.log)
#ifndef HM_FLAT_PREFETCH
#define HM_FLAT_PREFETCH 1   // flat-field operands fetched one group ahead, with the frame bytes
#endif
#ifndef HM_STD_EARLY
#define HM_STD_EARLY 1       // std + flat-field kernel: issue every std load of a sub-unit before pass 1 (N <= HM_PIN_NF). A/B on one
#endif                       // box (tools/ab3.sh, profiles/r02_ab_std1.log): config 3 890 -> 847 us; without the flat field it costs 10 %, so FLAT only
#ifndef HM_PIN_NF
#define HM_PIN_NF 8      // std kernel: above this N, pass 2 re-derives its per-frame addresses (registers)
#endif

#ifdef HM_WAVES          /* experiment knob (tools/build_alt.sh): ask the register allocator for this many waves per SIMD */
#define HM_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(HM_WAVES, HM_WAVES)))
#else
#define HM_WAVES_ATTR
#endif

template <int NF, int U, int TAB, bool STD, bool PREFETCH, bool FLAT, bool SUMW, int BLOCK, int CH = 3>
__device__ __forceinline__ void merge_u8_fast_body(const MergeK& a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    reset_hot_counters(a);
    static_assert(CH == 3 || (CH == 1 && STD && !SUMW), "monochrome: the std instantiations (with or without the flat field) only");
    constexpr int C = CH;
    // frames up to which pass 2 keeps its per-frame state pinned / its std loads issued early (flat-field instantiations). With the
    // sum-of-weights output on top, N = 8 needed 168 VGPRs + 20 bytes of scratch per lane: that one shape takes the N > 8 form (152 VGPRs)
    constexpr int PIN_NF = SUMW ? HM_PIN_NF - 1 : HM_PIN_NF;
    constexpr uint32_t GROUP = U * kSub;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane2 = lane * 2u;
    const uint32_t lane16 = lane * 16u;          // byte offset of the lane's two float64 (scalar base + 32-bit VGPR offset addressing)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t n_groups = static_cast<uint32_t>(a.n_elems / GROUP);
    const uint32_t gstride = gridDim.x * WPB;

    uint32_t g = blockIdx.x * WPB + wave;                                   // wave-uniform
    uint32_t raw[NF][U];
    auto load_group = [&](uint32_t grp, uint32_t (&dst)[NF][U]) {
        const int64_t off = a.in_off + static_cast<int64_t>(grp) * GROUP;   // scalar
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint8_t* p = static_cast<const uint8_t*>(a.frame[i]) + off;
#pragma unroll
            for (int s = 0; s < U; ++s) dst[i][s] = ld_u16(p + kSub * s + lane2);
        }
    };
    // flat-field operands of a group (FLAT only): with HM_FLAT_PREFETCH they are fetched one group ahead, with the frame bytes
    // (three waves per SIMD do not hide a load issued at the top of the sub-unit that consumes it)
    uint32_t fl_dn[U];
    f64x2 fl_v[U], fl_s[U];
    auto load_flat = [&](uint32_t grp, uint32_t (&dn)[U], f64x2 (&v)[U], f64x2 (&sd)[U]) {
        const int64_t fb = static_cast<int64_t>(grp) * GROUP;               // scalar, relative to row0
#pragma unroll
        for (int s = 0; s < U; ++s) {
            if (a.flat_u8) dn[s] = ld_u16(a.flat_u8 + fb + kSub * s + lane2);
            else v[s] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(a.flat_f64 + fb + kSub * s) + lane16));
            if (STD) sd[s] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(a.flat_std + fb + kSub * s) + lane16));
        }
    };
    constexpr bool FLAT_PF = FLAT && PREFETCH && HM_FLAT_PREFETCH;
    // the first group's HBM loads are in flight while the workgroup builds its LDS tables
    if (PREFETCH && g < n_groups) {
        load_group(g, raw);
        if constexpr (FLAT_PF) load_flat(g, fl_dn, fl_v, fl_s);
    }

    if constexpr (!STD) {
        fill_val_tables<TAB>(lds, a);
    } else {
        double2* t_wdw = reinterpret_cast<double2*>(lds);
        double2* t_gd = reinterpret_cast<double2*>(lds + 16 * 256);
        for (int i = threadIdx.x; i < 256; i += BLOCK) t_wdw[i] = double2{a.w_lut[i], a.dw_lut[i]};
        for (int i = threadIdx.x; i < 256 * C; i += BLOCK) t_gd[i] = double2{a.icrf[i], a.icrf_diff[i]};
    }
    // uint8 flat fields take only 256 values: {F = DN/255, 1/F**2} per DN in LDS saves two float64 divisions per element
    double2* t_flat = reinterpret_cast<double2*>(lds + (STD ? kStdTabBytes : TabInfo<TAB>::bytes));
    if constexpr (FLAT) {
        for (int i = threadIdx.x; i < 256; i += BLOCK) {
            const double F = static_cast<double>(i) / 255.0;
            t_flat[i] = double2{F, 1.0 / (F * F)};
        }
    }
    __syncthreads();

    for (; g < n_groups; g += gstride) {
        uint32_t cur[NF][U];
        uint32_t cfl_dn[U];
        f64x2 cfl_v[U], cfl_s[U];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int s = 0; s < U; ++s) cur[i][s] = raw[i][s];
            if constexpr (FLAT_PF) {
#pragma unroll
                for (int s = 0; s < U; ++s) { cfl_dn[s] = fl_dn[s]; cfl_v[s] = fl_v[s]; cfl_s[s] = fl_s[s]; }
            }
            if (g + gstride < n_groups) {
                load_group(g + gstride, raw);
                if constexpr (FLAT_PF) load_flat(g + gstride, fl_dn, fl_v, fl_s);
            }
        } else {
            load_group(g, cur);
        }
        if constexpr (FLAT && !FLAT_PF) load_flat(g, cfl_dn, cfl_v, cfl_s);
        const int64_t gbase = static_cast<int64_t>(g) * GROUP;            // relative to row0, scalar

#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int64_t sbase = gbase + kSub * s;                         // scalar
            const uint32_t c0 = C == 1 ? 0u : (2u * (g * U + s + lane)) % 3u;
            const uint32_t c1 = C == 1 ? 0u : (c0 + 1u) % 3u;
            const uint32_t coffs[2] = {STD ? c0 * 16u : chan_off<TAB>(c0), STD ? c1 * 16u : chan_off<TAB>(c1)};
            double* ov = a.out_val + sbase;                                   // scalar bases
            double* osw = SUMW ? a.out_sum_w + sbase : nullptr;

            // flat-field operands of the lane's two elements (one ushort / 16-byte loads)
            double F[2] = {1.0, 1.0}, sF[2] = {0.0, 0.0}, iF2[2] = {1.0, 1.0};
            if constexpr (FLAT) {
                {
                    if (a.flat_u8) {
                        const uint32_t f = cfl_dn[s];
                        const double2 f0 = t_flat[f & 255u], f1 = t_flat[f >> 8];
                        F[0] = f0.x; iF2[0] = f0.y; F[1] = f1.x; iF2[1] = f1.y;
                    } else {
                        const f64x2 f = cfl_v[s];
                        F[0] = f.x; F[1] = f.y;
                        if (STD) { iF2[0] = 1.0 / (F[0] * F[0]); iF2[1] = 1.0 / (F[1] * F[1]); }
                    }
                    if (STD) { sF[0] = cfl_s[s].x; sF[1] = cfl_s[s].y; }
                }
            }

            if constexpr (!STD) {
                double S[2], acc[2];
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const double it = a.inv_t[i];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t dn = j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8);
                        double w, wg;
                        gather_val<TAB>(lds, dn, coffs[j], w, wg);
                        if (i == 0) { S[j] = w; acc[j] = wg * it; }
                        else {
                            S[j] += w;                               // exposure_series.py:340
                            acc[j] = fma(wg, it, acc[j]);            // :388 numerator
                        }
                    }
                    // bound the gathers in flight (HM_FB frames = 2*HM_FB ds_read_b128, 8*HM_FB VGPRs) per scheduling
                    // bundle; HM_FB = 4 measured best (2: more VGPRs and 144 us; none: everything hoisted)
                    if (((i % HM_FB) == HM_FB - 1 || i == NF - 1)) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { HM_PIN(S[j]); HM_PIN(acc[j]); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                double val[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) val[j] = acc[j] / S[j];
                if constexpr (FLAT) {
                    double dummy = 0.0;
                    flat_field_math(F[0], iF2[0], 0.0, a.ff_mean[c0], 0.0, false, val[0], dummy);
                    flat_field_math(F[1], iF2[1], 0.0, a.ff_mean[c1], 0.0, false, val[1], dummy);
                }
                if constexpr (SUMW) store2(osw, lane16, S[0], S[1]);
                store2(ov, lane16, val[0], val[1]);
            } else {
                const double2* t_wdw = reinterpret_cast<const double2*>(lds);
                const char* t_gd = lds + 16 * 256;
                // HM_STD_EARLY (flat-field instantiations): all NF float64 std loads of the sub-unit are issued here, ahead of pass 1's
                // gathers (4 VGPRs per frame: stacks of up to HM_PIN_NF frames only), instead of HM_STD_FB frames at a time inside pass 2
                constexpr bool EARLY = HM_STD_EARLY && FLAT && NF <= PIN_NF;
                f64x2 sd_early[EARLY ? NF : 1];
                if constexpr (EARLY) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) sd_early[i] = ld_f64x2_global(a.sd[i] + a.in_off + sbase, lane16);
                }
                // pass 1: S = sum_i w_i
                double S[2];
#pragma unroll
                for (int i = 0; i < NF; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#ifdef HM_PROBE_STD   /* table-free traffic probe of the std kernels (wrong results; tuning builds only) */
                        const double w = static_cast<double>((j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8)) + 1u);
#else
                        const double w = t_wdw[j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8)].x;
#endif
                        if (i == 0) S[j] = w; else S[j] += w;
                    }
                    if ((i & 3) == 3 && i != NF - 1) {      // at most 8 weight gathers in flight (large N: registers)
                        HM_PIN(S[0]); HM_PIN(S[1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                double invS[2], invS2[2], acc[2], var[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    invS[j] = 1.0 / S[j];
                    invS2[j] = 1.0 / (S[j] * S[j]);                  // 1 / S**2, exposure_series.py:343
                    HM_PIN(invS[j]); HM_PIN(invS2[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // pass 2
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const double it = a.inv_t[i];
                    f64x2 sdv;
                    if constexpr (EARLY) sdv = sd_early[i];
                    else {
                        const double* sp = a.sd[i] + a.in_off + sbase;                           // scalar base
                        if (NF > PIN_NF || !FLAT) asm volatile("" : "+s"(sp));     // keep base + 32-bit lane offset addressing (no per-frame VGPR address pairs)
                        sdv = ld_f64x2_global(sp, lane16);
                    }
                    uint32_t packed = cur[i][s];
                    if (NF > PIN_NF || !FLAT) HM_PIN(packed);                  // re-extract the DNs here instead of keeping pass 1's indices alive
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t dn = j == 0 ? (packed & 255u) : (packed >> 8);
#ifdef HM_PROBE_STD
                        const double2 wdw = double2{static_cast<double>(dn + 1u), static_cast<double>(dn + 2u)};
                        const double2 gd = double2{static_cast<double>(dn + coffs[j]), static_cast<double>(dn + 3u)};
#else
                        const double2 wdw = t_wdw[dn];
                        const double2 gd = *reinterpret_cast<const double2*>(t_gd + dn * (16u * C) + coffs[j]);
#endif
                        const double w = wdw.x, dw = wdw.y, gg = gd.x;
                        const double dg = gd.y * (j == 0 ? sdv.x : sdv.y);                       // measurand.py:512
                        const double A = (dw * gg + w * dg) * invS[j] - ((dw * w) * gg) * invS2[j];   // :389
                        const double term = (A * dg) * it;
                        if (i == 0) { acc[j] = (w * gg) * it; var[j] = term * term; }
                        else {
                            acc[j] = fma(w * gg, it, acc[j]);                                    // :388
                            var[j] = fma(term, term, var[j]);
                        }
                    }
                    if ((i % HM_STD_FB) == HM_STD_FB - 1 || i == NF - 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { HM_PIN(acc[j]); HM_PIN(var[j]); }
                        __builtin_amdgcn_sched_barrier(0);   // HM_STD_FB frames' std loads + gathers at a time
                    }
                }
                double val[2], so[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    val[j] = acc[j] / S[j];
                    so[j] = sqrt(var[j]);                                                        // :394
                }
                if constexpr (FLAT) {
                    flat_field_math(F[0], iF2[0], sF[0], a.ff_mean[c0], a.ff_std_mean[c0], true, val[0], so[0]);
                    flat_field_math(F[1], iF2[1], sF[1], a.ff_mean[c1], a.ff_std_mean[c1], true, val[1], so[1]);
                }
                if constexpr (SUMW) store2(osw, lane16, S[0], S[1]);
                store2(ov, lane16, val[0], val[1]);
                store2(a.out_std + sbase, lane16, so[0], so[1]);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one sub-unit's gathers from piling onto the next one's
        }
    }
}

// Entry points. The std kernel is given an occupancy target of 3 waves/SIMD: with up to 168 VGPRs the scheduler keeps more
// loads in flight per wave (A/B on one box, tools/ab3.sh: 7 x 4096 x 4096 x 3 + std 716 -> 697 us, + flat field 869 -> 823 us;
// targets 4 and 5 are slower, the val-only kernel is insensitive).
#ifndef HM_STD_WAVES
#define HM_STD_WAVES 3
#endif
// Round 4: the flat-field instantiations (two more streams, the early std loads, the epilogue's operands) take a target of 2: config 3's
// std + flat stack 808-821 -> 790-792 us on one box, two runs each (tools/gpu_r04f.sh, profiles/r04f_*); without a flat field 2 costs 7 %.
#ifndef HM_STD_WAVES_FLAT
#define HM_STD_WAVES_FLAT 2
#endif
template <int NF, int U, int TAB, bool STD, bool PREFETCH, bool FLAT, bool SUMW, int BLOCK>
__global__ __launch_bounds__(BLOCK) HM_WAVES_ATTR void merge_u8_fast(const MergeK a) {
    static_assert(!STD, "std instantiations go through merge_u8_fast_std");
    merge_u8_fast_body<NF, U, TAB, false, PREFETCH, FLAT, SUMW, BLOCK>(a);
}
template <int NF, int U, int TAB, bool PREFETCH, bool FLAT, bool SUMW, int BLOCK, int CH = 3>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FLAT ? HM_STD_WAVES_FLAT : HM_STD_WAVES, FLAT ? HM_STD_WAVES_FLAT : HM_STD_WAVES)))
void merge_u8_fast_std(const MergeK a) {
    merge_u8_fast_body<NF, U, TAB, true, PREFETCH, FLAT, SUMW, BLOCK, CH>(a);
}

// ------------------------------------------------------------------------------------------------
// merge_u8_val3: val-only merge of uint8 frames, C == 3, compile-time N, no extras - the bench kernel (config 2).
//
// Element -> lane ownership is that of merge_u8_fast (lane l owns elements 2l, 2l+1 of a 128-element sub-unit: one
// global_load_ushort per frame, one fully contiguous 1 KB global_store_dwordx4 per wave), what changes is everything
// around the gathers, which rocprof showed to be the larger part of the VALU stream (profiles/r01g: 7.9 VALU
// instructions per element-frame, 2 of them the float64 accumulation):
//   * a wave's group is THREE sub-units = 384 elements = 128 whole pixels, so every group starts on channel 0 and the
//     channel of element j of sub-unit s is (2s + 2l + j) % 3 - a compile-time function of (s, j) on top of the lane
//     constant (2l) % 3. The three possible table offsets live in three VGPRs for the whole kernel; merge_u8_fast
//     recomputed its channel offsets per sub-unit with 32-bit multiplies-high (% 3 of a run-time group index).
//     The LDS address of a gather is then one v_mad_u32_u24 (dn * 48 + offset) after the byte extraction.
//   * the next group's loads go into a second register set and the loop body is written out twice (A/B), so the
//     prefetch costs no register-to-register copies (14 v_mov per 256 elements before).
//   * acc / S uses the correctly rounded reciprocal-Newton-Markstein sequence WITHOUT the operand scaling and fix-up
//     steps of the IEEE division expansion (v_div_scale x2, v_div_fixup: 3 of its 11 instructions) whenever the
//     workgroup has proven, while building its tables, that no operand can leave the range in which those steps are
//     the identity (div_inrange below); otherwise the plain division. Both give the same bits.
// ------------------------------------------------------------------------------------------------

// a / b, correctly rounded, for operands in the range the table check guarantees: b in [2^-64, 2^70] and a either +0 or
// |a| in [2^-660, 2^610]. This is instruction for instruction what hipcc emits for a float64 division minus the two
// v_div_scale_f64 (which return their operands unchanged in that range: no operand is denormal, the quotient and 1/b
// are normal, the numerator's biased exponent is above 53 and the exponent difference below 768) and the
// v_div_fixup_f64 (which only rewrites zero / infinite / NaN cases): r = rcp(b) refined twice, q = a r,
// q' = q + (a - b q) r (Markstein). a = +0 gives +0 on both paths; -0 never reaches it (the check rejects tables that hold one).
__device__ __forceinline__ double div_inrange(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    const double res = fma(-b, q, a);
    return fma(res, r, q);
}
template <bool FASTDIV>
__device__ __forceinline__ double merge_div(double a, double b) {
    if constexpr (FASTDIV) return div_inrange(a, b);
    else return a / b;
}
// table entry check behind div_inrange: w in [2^-64, 2^64]; w*g either +0 (bit pattern) or in [2^-300, 2^300] - POSITIVE: with
// entries of both signs the fma chain could cancel far below any single term (down to ~2^-706 for 16 frames), outside the range the
// argument above covers. Non-negative terms only grow the sum: acc is +0 or in [2^-600, 2^605] (times 1/t in [2^-300, 2^300], 16 terms).
__device__ __forceinline__ bool entry_inrange(double w, double wg) {
    const bool w_ok = w >= 0x1p-64 && w <= 0x1p64;
    const bool wg_ok = __double_as_longlong(wg) == 0 || (wg >= 0x1p-300 && wg <= 0x1p300);
    return w_ok && wg_ok;
}

// Template parameters (the defaults are chosen by launch_val3(); the others exist for A/B runs, tools/ab_val3.py):
//   U    sub-units of 128 elements per wave and iteration (2 or 3).
//   PF   0: in-place refill - each R register is reloaded with the next unit's bytes as soon as its DNs have been turned
//           into LDS addresses (one register set, loads issued in bundles of HM_FB frames between the gathers);
//        1: the next unit's loads are all issued at the top of the iteration into a second register set (A/B ping-pong).
//   MAP  0: a wave owns U*128 contiguous elements per iteration ("group");
//        1: the WORKGROUP owns U*512 contiguous elements ("chunk") and slot s of wave w sits at 512 s + 128 w, so the four
//           waves' loads of one time slot cover 512 contiguous bytes of every frame and their stores 4 contiguous KB.
// A unit (group or chunk) must start on channel 0 for the channel pattern to be a compile-time function of (s, j): its
// size is a multiple of 3 when U == 3; for U == 2 the unit index advances by a multiple of 3 per iteration (the host
// launches a multiple of 3 workgroups), so the phase is a per-wave constant folded into the lane's table offsets.
// FLAT (round 3; uint8 flat fields, PF == 1): the val-only flat-field epilogue (val / F) * m of measurand.py:602 - the flat's DNs travel as
// one more byte stream beside the frames', F = DN / 255 comes from a 2 KB LDS table, the channel means sit in three registers next to
// the table offsets. Before, a flat field sent the val-only merge to merge_u8_fast (186 us / 0.54 on config 2's stack).
// CH (round 3): 3 = colour (the channel pattern above), 1 = monochrome cameras - one table column, no channel bookkeeping at all; before,
// every C != 3 stack went to the run-time-N, run-time-C merge_u8_loop (0.62-0.68 val-only on 7 x 4096 x 4096 x 1).
#ifndef HM_VAL3_PROBE
#define HM_VAL3_PROBE 0      // measurement builds only (wrong results): 1 = no LDS gathers (the ceiling of the HBM access pattern); 2 = every unit's
#endif                       // frame bytes come from the first 64 units (L2 hits; the gathers and the stores are real); 3 = 2 without the stores
#if HM_VAL3_PROBE >= 2
#define HM_PROBE_UNIT(x) ((x) & 63u)
#else
#define HM_PROBE_UNIT(x) (x)
#endif
template <int NF, int U, int PF, int MAP, bool FLAT = false, int CH = 3>
__global__ __launch_bounds__(256) void merge_u8_val3(const MergeK a) {
    static_assert(CH == 3 || CH == 1, "colour or monochrome");
    constexpr uint32_t TS = 16u * CH;                              // bytes of table per DN
    __shared__ __attribute__((aligned(16))) char lds[16 * 256 * CH];
    __shared__ uint32_t s_bad[4];
    __shared__ double t_F[FLAT ? 256 : 1];
    static_assert(!FLAT || PF == 1, "the flat-field instantiation uses the two-register-set prefetch");
    constexpr int NS = NF + (FLAT ? 1 : 0);                        // byte streams: the frames (+ the flat field's DNs as stream NF)
    reset_hot_counters(a);
    constexpr bool DEFER = MAP == 2;                               // MAP 2: the element map of MAP 0, all U stores of a unit issued together at its end
    constexpr bool PIPE = MAP == 3;                                // MAP 3: the element map of MAP 0, the gathers of frame bundle b + 1 issued under the accumulation of bundle b
    constexpr bool WGMAP = MAP == 1;
    constexpr uint32_t SLOT = WGMAP ? 4u * kSub : kSub;            // element distance between a wave's consecutive slots
    constexpr uint32_t UNIT = U * SLOT;                            // elements per unit (group: U*128, chunk: U*512)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane16 = lane * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wave_el = WGMAP ? wave * kSub : 0u;             // scalar: the wave's element offset inside a chunk
    const uint32_t lane2 = lane * 2u + wave_el;                    // element (= byte) offset of the lane's first element inside the unit
    const uint32_t n_units = static_cast<uint32_t>(a.n_elems / UNIT);
    const uint32_t ustride = WGMAP ? gridDim.x : gridDim.x * 4u;
    uint32_t u = WGMAP ? blockIdx.x : blockIdx.x * 4u + wave;      // wave-uniform

    // R[i][s]: the two DNs of frame i, slot s (kept 16-bit: widening at the load would put a v_and - and a wait for the
    // load - right behind it)
    // (PF == 1 keeps 32-bit registers filled by zero-extending global loads: with two 16-bit register sets hipcc packs pairs
    // of them into one VGPR with v_perm_b32 at the loop back-edge, i.e. waits for the prefetch it has just issued)
    using reg_t = std::conditional_t<PF == 0, uint16_t, uint32_t>;
    reg_t RA[NS][U], RB[PF ? NS : 1][PF ? U : 1];
    auto load_unit = [&](uint32_t unit, auto& dst) {
        if constexpr (FLAT) {                                      // the flat field covers the OUTPUT rows: no in_off
            const uint8_t* p = a.flat_u8 + static_cast<int64_t>(unit) * UNIT;
#pragma unroll
            for (int s = 0; s < U; ++s) dst[NF][s] = ld_u16(p + SLOT * s + lane2);
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if constexpr (PF == 0) {
                const __amdgpu_buffer_rsrc_t fr = frame_rsrc(static_cast<const uint8_t*>(a.frame[i]) + a.in_off);
#pragma unroll
                for (int s = 0; s < U; ++s) dst[i][s] = ld_u16_buf(fr, lane2 + SLOT * s, HM_PROBE_UNIT(unit) * UNIT);
            } else {
                const uint8_t* p = static_cast<const uint8_t*>(a.frame[i]) + a.in_off + static_cast<int64_t>(HM_PROBE_UNIT(unit)) * UNIT;
#pragma unroll
                for (int s = 0; s < U; ++s) dst[i][s] = ld_u16(p + SLOT * s + lane2);
            }
        }
    };
    if (u < n_units) load_unit(u, RA);                                              // in flight while the tables are built

    bool bad = false;
    for (int q = threadIdx.x; q < 256 * CH; q += 256) {
        const double w = a.w_lut[q / CH];
        const double wg = w * a.icrf[q];                                            // (w * g), exposure_series.py:388
        reinterpret_cast<double2*>(lds)[q] = double2{w, wg};
        bad = bad || !entry_inrange(w, wg);
    }
    if constexpr (FLAT) t_F[threadIdx.x] = static_cast<double>(threadIdx.x) / 255.0;     // F = DN / 255 (image_set.py:223)
    const bool wave_bad = __ballot(bad) != 0ull;
    if (lane == 0) s_bad[wave] = wave_bad ? 1u : 0u;
    __syncthreads();
    const uint32_t any_bad = __builtin_amdgcn_readfirstlane(s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]);
    const bool fastdiv = any_bad == 0u && a.inv_t_inrange != 0;                     // scalar
    if (u >= n_units) return;

    // channel of the lane's first element in slot 0: (unit start + wave offset + 2 lane) % 3; the unit start is a
    // per-wave constant mod 3 (see above)
    const uint32_t p0 = CH == 1 ? 0u : static_cast<uint32_t>((static_cast<uint64_t>(u) * UNIT) % 3u);
    const uint32_t k = CH == 1 ? 0u : (p0 + lane2) % 3u;
    const uint32_t off[3] = {k * 16u, CH == 1 ? 0u : ((k + 1u) % 3u) * 16u, CH == 1 ? 0u : ((k + 2u) % 3u) * 16u};
    auto ci = [](int x) constexpr { return CH == 1 ? 0 : x % 3; };      // which of the three offsets / means element x of a unit uses
    // FLAT: the ROI means of the channels k, k + 1, k + 2 (no per-lane indexing of the kernarg array)
    double mm[3] = {1.0, 1.0, 1.0};
    if constexpr (FLAT) {
        const double m0 = a.ff_mean[0], m1 = CH == 1 ? m0 : a.ff_mean[1], m2 = CH == 1 ? m0 : a.ff_mean[2];
        mm[0] = k == 0u ? m0 : k == 1u ? m1 : m2;
        mm[1] = k == 0u ? m1 : k == 1u ? m2 : m0;
        mm[2] = k == 0u ? m2 : k == 1u ? m0 : m1;
    }
    auto flat_epilogue = [&](int s_, reg_t r, double& v0, double& v1) {          // measurand.py:602, as flat_field_math() does it
        if constexpr (FLAT) {
            v0 = (v0 / t_F[r & 255u]) * mm[ci(2 * s_)];
            v1 = (v1 / t_F[r >> 8]) * mm[ci(2 * s_ + 1)];
        }
    };

    // one unit out of `cur`; REFILL (PF == 0 only): fetch the next unit's bytes into the registers this one frees
    auto process = [&](auto refill_tag, uint32_t unit, auto& cur) {
        constexpr bool REFILL = decltype(refill_tag)::value;
        const uint32_t next_off = HM_PROBE_UNIT(unit + ustride) * UNIT;                                   // scalar: byte offset of the next unit in every frame
        double* og = a.out_val + static_cast<int64_t>(unit) * UNIT + wave_el;                // scalar base of the wave's output in this unit
        double held[DEFER ? U : 1][2];
        if constexpr (PIPE) {
            // software-pipelined gathers ("software-pipelined LDS gathers"): two sets of gathered {w, w g} pairs; the ds_read_b128 of the next
            // bundle of HM_FB frames (of this or the next sub-unit) are in flight while the current bundle's add / fma chain runs
            constexpr int NB = (NF + HM_FB - 1) / HM_FB;
            double2 T[2][HM_FB][2];
            auto issue = [&](int s_, int b_, double2 (&dst)[HM_FB][2]) {
#pragma unroll
                for (int f = 0; f < HM_FB; ++f) {
                    if (b_ * HM_FB + f < NF) {
                        const reg_t r = cur[b_ * HM_FB + f][s_];
                        dst[f][0] = *reinterpret_cast<const double2*>(lds + (__umul24(static_cast<uint32_t>(r & 255u), TS) + off[ci(2 * s_)]));
                        dst[f][1] = *reinterpret_cast<const double2*>(lds + (__umul24(static_cast<uint32_t>(r >> 8), TS) + off[ci(2 * s_ + 1)]));
                    }
                }
                if constexpr (REFILL) {                     // PF == 0: the bytes just turned into addresses make room for the next unit's
#pragma unroll
                    for (int f = 0; f < HM_FB; ++f)
                        if (b_ * HM_FB + f < NF)
                            cur[b_ * HM_FB + f][s_] = ld_u16_buf(frame_rsrc(static_cast<const uint8_t*>(a.frame[b_ * HM_FB + f]) + a.in_off), lane2 + SLOT * s_, next_off);
                }
            };
            issue(0, 0, T[0]);
#pragma unroll
            for (int s = 0; s < U; ++s) {
                double S[2], acc[2];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    constexpr int dummy = 0; (void)dummy;
                    const int gi = s * NB + b;
                    if (gi + 1 < U * NB) issue((gi + 1) / NB, (gi + 1) % NB, T[(gi + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < HM_FB; ++f) {
                        if (b * HM_FB + f < NF) {
                            const double it = a.inv_t[b * HM_FB + f];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const double2 t = T[gi & 1][f][j];
                                if (b * HM_FB + f == 0) { S[j] = t.x; acc[j] = t.y * it; }
                                else {
                                    S[j] += t.x;                               // exposure_series.py:340
                                    acc[j] = fma(t.y, it, acc[j]);             // :388 numerator
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) { HM_PIN(S[j]); HM_PIN(acc[j]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                double v0, v1;
                if (fastdiv) { v0 = div_inrange(acc[0], S[0]); v1 = div_inrange(acc[1], S[1]); }
                else { v0 = acc[0] / S[0]; v1 = acc[1] / S[1]; }
                flat_epilogue(s, cur[NS - 1][s], v0, v1);
                store2(og + SLOT * s, lane16, v0, v1);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < U; ++s) {
            double S[2], acc[2];
#pragma unroll
            for (int i0 = 0; i0 < NF; i0 += HM_FB) {                                // HM_FB frames per scheduling bundle
                uint32_t addr[HM_FB][2];
#pragma unroll
                for (int f = 0; f < HM_FB; ++f) {
                    if (i0 + f < NF) {
                        const reg_t r = cur[i0 + f][s];
                        addr[f][0] = __umul24(static_cast<uint32_t>(r & 255u), TS) + off[ci(2 * s)];
                        addr[f][1] = __umul24(static_cast<uint32_t>(r >> 8), TS) + off[ci(2 * s + 1)];
                    }
                }
                if constexpr (REFILL) {
                    __builtin_amdgcn_sched_barrier(0);      // the refill must not be hoisted above the last use of the old bytes (it would need a second register set)
#pragma unroll
                    for (int f = 0; f < HM_FB; ++f)
                        if (i0 + f < NF)
                            cur[i0 + f][s] = ld_u16_buf(frame_rsrc(static_cast<const uint8_t*>(a.frame[i0 + f]) + a.in_off), lane2 + SLOT * s, next_off);
                }
#pragma unroll
                for (int f = 0; f < HM_FB; ++f) {
                    if (i0 + f < NF) {
                        const double it = a.inv_t[i0 + f];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
#if HM_VAL3_PROBE == 1
                            const double2 t = double2{static_cast<double>(addr[f][j] + 1u), static_cast<double>(addr[f][j])};
#else
                            const double2 t = *reinterpret_cast<const double2*>(lds + addr[f][j]);
#endif
                            if (i0 + f == 0) { S[j] = t.x; acc[j] = t.y * it; }
                            else {
                                S[j] += t.x;                               // exposure_series.py:340
                                acc[j] = fma(t.y, it, acc[j]);             // :388 numerator
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { HM_PIN(S[j]); HM_PIN(acc[j]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            double v0, v1;
            if (fastdiv) { v0 = div_inrange(acc[0], S[0]); v1 = div_inrange(acc[1], S[1]); }
            else { v0 = acc[0] / S[0]; v1 = acc[1] / S[1]; }
            flat_epilogue(s, cur[NS - 1][s], v0, v1);
            if constexpr (DEFER) { held[s][0] = v0; held[s][1] = v1; }
#if HM_VAL3_PROBE == 3
            else { if (v0 != v0 && v1 == 12345.0) store2(og + SLOT * s, lane16, v0, v1); }
#else
            else store2(og + SLOT * s, lane16, v0, v1);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DEFER) {                      // U KB of contiguous output in U back-to-back store instructions
#pragma unroll
            for (int s = 0; s < U; ++s) store2(og + SLOT * s, lane16, held[s][0], held[s][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (PF == 0) {
        for (; u + ustride < n_units; u += ustride) process(std::true_type{}, u, RA);
        process(std::false_type{}, u, RA);
    } else {
        while (true) {                                                               // RA holds unit u
            if (u + ustride >= n_units) { process(std::false_type{}, u, RA); break; }
            load_unit(u + ustride, RB);
            __builtin_amdgcn_sched_barrier(0);
            process(std::false_type{}, u, RA);
            u += ustride;                                                            // RB holds unit u
            if (u + ustride >= n_units) { process(std::false_type{}, u, RB); break; }
            load_unit(u + ustride, RA);
            __builtin_amdgcn_sched_barrier(0);
            process(std::false_type{}, u, RB);
            u += ustride;
        }
    }
}

#if HM_TUNE_NF != 0
#include "../../tools/hm_merge_priv.inc"     // merge_u8_priv: the conflict-free-LDS experiment (tuning builds only, lives with the other experiments)
#endif

// ------------------------------------------------------------------------------------------------
// merge_u8_loop: the same work decomposition and arithmetic as merge_u8_fast with the frame count as a run-time
// value, for 16 < N <= HM_MAX_FRAMES (the templated kernel is instantiated for N <= 16; before this kernel larger
// stacks fell to merge_generic at 0.19-0.33 of the HBM roofline, tools/bench_n.py). Frames are consumed in chunks
// of kLoopChunk: the chunk's ushort loads are issued together (frame pointers and 1/t come from the kernarg segment
// through scalar loads with a uniform index), then its LDS gathers, then the accumulation in frame order. The std
// variant re-reads the frame bytes in its second pass (L2 hits, 1 byte per element-frame) instead of keeping N
// packed words in registers. No cross-group prefetch: the kernel stays under 64 VGPRs and relies on 8 waves/SIMD.
// ------------------------------------------------------------------------------------------------
constexpr int kLoopChunk = 8;
constexpr int kTemplatedN = 20;                    // frame counts with their own instantiation of the templated kernels (see hm_merge's dispatch)

template <int C, bool STD, bool FLAT, bool SUMW>
__device__ __forceinline__ void merge_u8_loop_body(const MergeK& a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    reset_hot_counters(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane2 = lane * 2u, lane16 = lane * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = 4;
    const uint32_t n_groups = static_cast<uint32_t>(a.n_elems / kSub);
    const uint32_t gstride = gridDim.x * WPB;
    const int N = a.n_frames;

    if constexpr (!STD) {
        for (int q = threadIdx.x; q < 256 * C; q += 256) {
            const double w = a.w_lut[q / C];
            reinterpret_cast<double2*>(lds)[q] = double2{w, w * a.icrf[q]};      // {w, w * g}, exposure_series.py:388
        }
    } else {
        double2* t_wdw = reinterpret_cast<double2*>(lds);
        double2* t_gd = reinterpret_cast<double2*>(lds + 16 * 256);
        for (int i = threadIdx.x; i < 256; i += 256) t_wdw[i] = double2{a.w_lut[i], a.dw_lut[i]};
        for (int i = threadIdx.x; i < 256 * C; i += 256) t_gd[i] = double2{a.icrf[i], a.icrf_diff[i]};
    }
    double2* t_flat = reinterpret_cast<double2*>(lds + (STD ? 16 * 256 + 16 * 256 * C : 16 * 256 * C));
    if constexpr (FLAT) {
        for (int i = threadIdx.x; i < 256; i += 256) {
            const double F = static_cast<double>(i) / 255.0;
            t_flat[i] = double2{F, 1.0 / (F * F)};
        }
    }
    __syncthreads();

    for (uint32_t g = blockIdx.x * WPB + wave; g < n_groups; g += gstride) {
        const int64_t sbase = static_cast<int64_t>(g) * kSub;                 // relative to row0, scalar
        const int64_t ibase = a.in_off + sbase;
        const uint32_t c0 = C == 1 ? 0u : static_cast<uint32_t>((static_cast<uint64_t>(g) * (kSub % C) + lane2) % C);   // element % C
        const uint32_t c1 = C == 1 ? 0u : (c0 + 1u) % C;
        const uint32_t coffs[2] = {c0 * 16u, c1 * 16u};

        double F[2] = {1.0, 1.0}, sF[2] = {0.0, 0.0}, iF2[2] = {1.0, 1.0};
        if constexpr (FLAT) {
            if (a.flat_u8) {
                const uint32_t f = ld_u16(a.flat_u8 + sbase + lane2);
                const double2 f0 = t_flat[f & 255u], f1 = t_flat[f >> 8];
                F[0] = f0.x; iF2[0] = f0.y; F[1] = f1.x; iF2[1] = f1.y;
            } else {
                const f64x2 f = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(a.flat_f64 + sbase) + lane16));
                F[0] = f.x; F[1] = f.y;
                if (STD) { iF2[0] = 1.0 / (F[0] * F[0]); iF2[1] = 1.0 / (F[1] * F[1]); }
            }
            if (STD) {
                const f64x2 f = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(a.flat_std + sbase) + lane16));
                sF[0] = f.x; sF[1] = f.y;
            }
        }

        double S[2] = {0.0, 0.0}, acc[2] = {0.0, 0.0}, var[2] = {0.0, 0.0};
        if constexpr (!STD) {
            for (int i0 = 0; i0 < N; i0 += kLoopChunk) {
                uint32_t r[kLoopChunk];
#pragma unroll
                for (int k = 0; k < kLoopChunk; ++k)
                    if (i0 + k < N) r[k] = ld_u16(static_cast<const uint8_t*>(a.frame[i0 + k]) + ibase + lane2);
#pragma unroll
                for (int k = 0; k < kLoopChunk; ++k) {
                    if (i0 + k < N) {
                        const double it = a.inv_t[i0 + k];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint32_t dn = j == 0 ? (r[k] & 255u) : (r[k] >> 8);
                            const double2 tw = *reinterpret_cast<const double2*>(lds + dn * (16u * C) + coffs[j]);
                            const double w = tw.x, wg = tw.y;
                            if (i0 + k == 0) { S[j] = w; acc[j] = wg * it; }
                            else { S[j] += w; acc[j] = fma(wg, it, acc[j]); }            // exposure_series.py:340, :388
                        }
                    }
                }
            }
            double val[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) val[j] = acc[j] / S[j];
            if constexpr (FLAT) {
                double dummy = 0.0;
                flat_field_math(F[0], iF2[0], 0.0, a.ff_mean[c0], 0.0, false, val[0], dummy);
                flat_field_math(F[1], iF2[1], 0.0, a.ff_mean[c1], 0.0, false, val[1], dummy);
            }
            if constexpr (SUMW) store2(a.out_sum_w + sbase, lane16, S[0], S[1]);
            store2(a.out_val + sbase, lane16, val[0], val[1]);
        } else {
            const double2* t_wdw = reinterpret_cast<const double2*>(lds);
            const char* t_gd = lds + 16 * 256;
            // pass 1: S = sum_i w_i
            for (int i0 = 0; i0 < N; i0 += kLoopChunk) {
                uint32_t r[kLoopChunk];
#pragma unroll
                for (int k = 0; k < kLoopChunk; ++k)
                    if (i0 + k < N) r[k] = ld_u16(static_cast<const uint8_t*>(a.frame[i0 + k]) + ibase + lane2);
#pragma unroll
                for (int k = 0; k < kLoopChunk; ++k) {
                    if (i0 + k < N) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const double w = t_wdw[j == 0 ? (r[k] & 255u) : (r[k] >> 8)].x;
                            if (i0 + k == 0) S[j] = w; else S[j] += w;
                        }
                    }
                }
            }
            double invS[2], invS2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                invS[j] = 1.0 / S[j];
                invS2[j] = 1.0 / (S[j] * S[j]);                          // 1 / S**2, exposure_series.py:343
            }
            // pass 2: the frame bytes again (cache hits) + the float64 std streams, two frames per step
            for (int i0 = 0; i0 < N; i0 += 2) {
                uint32_t r[2];
                f64x2 sdv[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (i0 + k < N) {
                        r[k] = ld_u16(static_cast<const uint8_t*>(a.frame[i0 + k]) + ibase + lane2);
                        const double* sp = a.sd[i0 + k] + ibase;
                        sdv[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(sp) + lane16));
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (i0 + k < N) {
                        const double it = a.inv_t[i0 + k];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint32_t dn = j == 0 ? (r[k] & 255u) : (r[k] >> 8);
                            const double2 wdw = t_wdw[dn];
                            const double2 gd = *reinterpret_cast<const double2*>(t_gd + dn * (16u * C) + coffs[j]);
                            const double w = wdw.x, dw = wdw.y, gg = gd.x;
                            const double dg = gd.y * (j == 0 ? sdv[k].x : sdv[k].y);                       // measurand.py:512
                            const double A = (dw * gg + w * dg) * invS[j] - ((dw * w) * gg) * invS2[j];   // :389
                            const double term = (A * dg) * it;
                            if (i0 + k == 0) { acc[j] = (w * gg) * it; var[j] = term * term; }
                            else {
                                acc[j] = fma(w * gg, it, acc[j]);                                          // :388
                                var[j] = fma(term, term, var[j]);
                            }
                        }
                    }
                }
            }
            double val[2], so[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                val[j] = acc[j] / S[j];
                so[j] = sqrt(var[j]);                                                                      // :394
            }
            if constexpr (FLAT) {
                flat_field_math(F[0], iF2[0], sF[0], a.ff_mean[c0], a.ff_std_mean[c0], true, val[0], so[0]);
                flat_field_math(F[1], iF2[1], sF[1], a.ff_mean[c1], a.ff_std_mean[c1], true, val[1], so[1]);
            }
            if constexpr (SUMW) store2(a.out_sum_w + sbase, lane16, S[0], S[1]);
            store2(a.out_val + sbase, lane16, val[0], val[1]);
            store2(a.out_std + sbase, lane16, so[0], so[1]);
        }
    }
}

#ifdef HM_LOOP_VAL_WAVES
#define HM_LOOP_VAL_ATTR __attribute__((amdgpu_waves_per_eu(HM_LOOP_VAL_WAVES, HM_LOOP_VAL_WAVES)))
#else
#define HM_LOOP_VAL_ATTR
#endif
#ifdef HM_LOOP_STD_WAVES
#define HM_LOOP_STD_ATTR __attribute__((amdgpu_waves_per_eu(HM_LOOP_STD_WAVES, HM_LOOP_STD_WAVES)))
#else
#define HM_LOOP_STD_ATTR
#endif
template <int C, bool FLAT, bool SUMW>
__global__ __launch_bounds__(256) HM_LOOP_VAL_ATTR void merge_u8_loop(const MergeK a) { merge_u8_loop_body<C, false, FLAT, SUMW>(a); }
template <int C, bool FLAT, bool SUMW>
__global__ __launch_bounds__(256) HM_LOOP_STD_ATTR void merge_u8_loop_std(const MergeK a) { merge_u8_loop_body<C, true, FLAT, SUMW>(a); }

// ------------------------------------------------------------------------------------------------
// merge_f64_val / merge_f64_std (body: merge_f64_body): float64 frames (the reference's 64-bit mode, image_set.py:225 / frames saved by save_64bit) with the
// streaming decomposition of merge_u8_loop: lane l owns elements 2l, 2l+1 of a 128-element group, every frame / std /
// output access is one 16-byte load or store per lane (1 KB contiguous per wave instruction). The weight is evaluated
// analytically (measurand.py:615-616) and the LUT index is computed (round-half-even, wrap, :503), exactly as
// merge_generic does; val-only needs one pass (S and the numerator accumulate together), std needs S first and
// re-evaluates the weights in its second pass. Stacks of up to 8 frames keep the frame VALUES in registers between
// the passes; larger ones re-read them (not the weights: with exp() stubbed out the kernel is 1.5 % faster, it is
// bound by memory traffic, and the re-read misses the caches). merge_generic evaluated exp() twice per
// element-frame even for val-only and moved 8-byte pieces: 1 307 -> see docs/DESIGN_history_r01_r02.md section 8 for the numbers.
// ------------------------------------------------------------------------------------------------
constexpr int kF64Chunk = 4;
constexpr int kF64Keep = 8;        // std mode: stacks up to this size keep their frame values in registers between the passes

__device__ __forceinline__ uint32_t lut_index_f64(double v) {
    return static_cast<uint32_t>(static_cast<int64_t>(rint(v * 255.0))) & 255u;          // measurand.py:503
}

template <int C, bool STD, bool FLAT, bool SUMW>
__device__ __forceinline__ void merge_f64_body(const MergeK& a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    reset_hot_counters(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane2 = lane * 2u, lane16 = lane * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = 4;
    const uint32_t n_groups = static_cast<uint32_t>(a.n_elems / kSub);
    const uint32_t gstride = gridDim.x * WPB;
    const int N = a.n_frames;
    double2* t_gd = reinterpret_cast<double2*>(lds);                              // {g, d} per (dn, c)
    for (int i = threadIdx.x; i < 256 * C; i += 256) t_gd[i] = double2{a.icrf[i], STD ? a.icrf_diff[i] : 0.0};
    double2* t_flat = reinterpret_cast<double2*>(lds + 16 * 256 * C);
    if constexpr (FLAT) {
        for (int i = threadIdx.x; i < 256; i += 256) {
            const double F = static_cast<double>(i) / 255.0;
            t_flat[i] = double2{F, 1.0 / (F * F)};
        }
    }
    __syncthreads();
    auto ld2 = [&](const double* base) -> f64x2 {
        return __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(base) + lane16));
    };

#ifndef HM_F64_PREFETCH
#define HM_F64_PREFETCH 0    // std mode, N <= kF64Keep: the next group's frame values loaded (second register set) before the current group's exp()s.
                             // Round 3, measured and left off: the kernel sits at its 3-waves/SIMD register budget (166 VGPRs) and the second set spills
                             // (1 890 us against 1 305 us on 7 x 4096 x 4096 x 3 + std); at 2 waves/SIMD 1 565 us, without the kept weights 1 456 us,
                             // that at 4 waves/SIMD 1 745 us (profiles/r03_ab_f64std_prefetch.log). As with the uint8 std kernels, more loads in
                             // flight per wave do not help a kernel that already covers bandwidth x latency across its waves.
#endif
    constexpr bool PRE = STD && HM_F64_PREFETCH;
    const bool keep_regs = STD && N <= kF64Keep;                                  // wave-uniform
    f64x2 vpre[PRE ? kF64Keep : 1];
    if constexpr (PRE) {
        const uint32_t g0 = blockIdx.x * WPB + wave;
        if (keep_regs && g0 < n_groups) {
#pragma unroll
            for (int i = 0; i < kF64Keep; ++i)
                if (i < N) vpre[i] = ld2(static_cast<const double*>(a.frame[i]) + a.in_off + static_cast<int64_t>(g0) * kSub);
        }
    }
    for (uint32_t g = blockIdx.x * WPB + wave; g < n_groups; g += gstride) {
        const int64_t sbase = static_cast<int64_t>(g) * kSub;
        const int64_t ibase = a.in_off + sbase;
        const uint32_t c0 = C == 1 ? 0u : static_cast<uint32_t>((static_cast<uint64_t>(g) * (kSub % C) + lane2) % C);
        const uint32_t c1 = C == 1 ? 0u : (c0 + 1u) % C;
        const uint32_t cs[2] = {c0, c1};

        double F[2] = {1.0, 1.0}, sF[2] = {0.0, 0.0}, iF2[2] = {1.0, 1.0};
        if constexpr (FLAT) {
            if (a.flat_u8) {
                const uint32_t f = ld_u16(a.flat_u8 + sbase + lane2);
                const double2 f0 = t_flat[f & 255u], f1 = t_flat[f >> 8];
                F[0] = f0.x; iF2[0] = f0.y; F[1] = f1.x; iF2[1] = f1.y;
            } else {
                const f64x2 f = ld2(a.flat_f64 + sbase);
                F[0] = f.x; F[1] = f.y;
                if (STD) { iF2[0] = 1.0 / (F[0] * F[0]); iF2[1] = 1.0 / (F[1] * F[1]); }
            }
            if (STD) { const f64x2 f = ld2(a.flat_std + sbase); sF[0] = f.x; sF[1] = f.y; }
        }

        double S[2] = {0.0, 0.0}, acc[2] = {0.0, 0.0}, var[2] = {0.0, 0.0};
        if constexpr (!STD) {
            for (int i0 = 0; i0 < N; i0 += kF64Chunk) {
                f64x2 v[kF64Chunk];
#pragma unroll
                for (int k = 0; k < kF64Chunk; ++k)
                    if (i0 + k < N) v[k] = ld2(static_cast<const double*>(a.frame[i0 + k]) + ibase);
#pragma unroll
                for (int k = 0; k < kF64Chunk; ++k) {
                    if (i0 + k < N) {
                        const double it = a.inv_t[i0 + k];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const double x = j == 0 ? v[k].x : v[k].y;
                            const double dv = x - 0.5;
                            const double w = gauss_weight(dv);                            // measurand.py:615
                            const double gg = t_gd[lut_index_f64(x) * C + cs[j]].x;
                            const double wg = w * gg;
                            if (i0 + k == 0) { S[j] = w; acc[j] = wg * it; }
                            else { S[j] += w; acc[j] = fma(wg, it, acc[j]); }                   // exposure_series.py:340, :388
                        }
                    }
                }
            }
            double val[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) val[j] = acc[j] / S[j];
            if constexpr (FLAT) {
                double dummy = 0.0;
                flat_field_math(F[0], iF2[0], 0.0, a.ff_mean[c0], 0.0, false, val[0], dummy);
                flat_field_math(F[1], iF2[1], 0.0, a.ff_mean[c1], 0.0, false, val[1], dummy);
            }
            if constexpr (SUMW) store2(a.out_sum_w + sbase, lane16, S[0], S[1]);
            store2(a.out_val + sbase, lane16, val[0], val[1]);
        } else {
            double invS[2], invS2[2];
            // one frame of pass 2 (measurand.py:512,616; exposure_series.py:388-389) given its weight
            auto pass2 = [&](int i, int j, double x, double w, double s, double it) {
                const double dw = (-60.0 * (x - 0.5)) * w;                                      // measurand.py:616
                const double2 gd = t_gd[lut_index_f64(x) * C + cs[j]];
                const double gg = gd.x;
                const double wg = w * gg;
                const double dg = gd.y * s;                                                     // measurand.py:512
                const double A = (dw * gg + w * dg) * invS[j] - ((dw * w) * gg) * invS2[j];     // :389
                const double term = (A * dg) * it;
                if (i == 0) { acc[j] = wg * it; var[j] = term * term; }
                else { acc[j] = fma(wg, it, acc[j]); var[j] = fma(term, term, var[j]); }
            };
            if (N <= kF64Keep) {
                // small stacks: the frame values stay in registers between the passes (re-reading them misses the
                // caches - 7 KB per wave-iteration, 20 waves per CU - and costs 56 B/element of extra traffic);
                // the weight is re-evaluated, exp() is not what bounds this kernel
                f64x2 v[kF64Keep];
                constexpr bool KEEPW = HM_F64_KEEP_W && !FLAT;   // (with the flat-field operands live as well the kept weights spill)
                double wk[KEEPW ? kF64Keep : 1][2];              // pass 1's weights, reused by pass 2 (one exp() per element-frame instead of two)
                if constexpr (PRE) {
                    // (unconditional: past the last group it re-reads the current one and the values are dropped - a branch around the loads
                    // would make the compiler wait for them before this group's arithmetic)
                    const int64_t nbase = a.in_off + static_cast<int64_t>(g + gstride < n_groups ? g + gstride : g) * kSub;
#pragma unroll
                    for (int i = 0; i < kF64Keep; ++i)
                        if (i < N) { v[i] = vpre[i]; vpre[i] = ld2(static_cast<const double*>(a.frame[i]) + nbase); }
                } else {
#pragma unroll
                    for (int i = 0; i < kF64Keep; ++i)
                        if (i < N) v[i] = ld2(static_cast<const double*>(a.frame[i]) + ibase);
                }
#pragma unroll
                for (int i = 0; i < kF64Keep; ++i) {
                    if (i < N) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const double w = gauss_weight((j == 0 ? v[i].x : v[i].y) - 0.5);
                            if constexpr (KEEPW) wk[i][j] = w;
                            if (i == 0) S[j] = w; else S[j] += w;
                        }
                        HM_PIN(S[0]); HM_PIN(S[1]);                 // one frame's exp() pair at a time: their temporaries,
                        __builtin_amdgcn_sched_barrier(0);          // interleaved across frames, are the register peak
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { invS[j] = 1.0 / S[j]; invS2[j] = 1.0 / (S[j] * S[j]); }
#pragma unroll
                for (int i = 0; i < kF64Keep; ++i) {
                    if (i < N) {
                        const f64x2 sdv = ld2(a.sd[i] + ibase);
                        const double it = a.inv_t[i];
                        if constexpr (KEEPW) {
                            pass2(i, 0, v[i].x, wk[i][0], sdv.x, it);
                            pass2(i, 1, v[i].y, wk[i][1], sdv.y, it);
                        } else {
                            pass2(i, 0, v[i].x, gauss_weight(v[i].x - 0.5), sdv.x, it);
                            pass2(i, 1, v[i].y, gauss_weight(v[i].y - 0.5), sdv.y, it);
                        }
                        HM_PIN(acc[0]); HM_PIN(acc[1]); HM_PIN(var[0]); HM_PIN(var[1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                for (int i0 = 0; i0 < N; i0 += kF64Chunk) {                                    // pass 1: S
                    f64x2 v[kF64Chunk];
#pragma unroll
                    for (int k = 0; k < kF64Chunk; ++k)
                        if (i0 + k < N) v[k] = ld2(static_cast<const double*>(a.frame[i0 + k]) + ibase);
#pragma unroll
                    for (int k = 0; k < kF64Chunk; ++k) {
                        if (i0 + k < N) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const double dv = (j == 0 ? v[k].x : v[k].y) - 0.5;
                                const double w = gauss_weight(dv);
                                if (i0 + k == 0) S[j] = w; else S[j] += w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) { invS[j] = 1.0 / S[j]; invS2[j] = 1.0 / (S[j] * S[j]); }
                for (int i0 = 0; i0 < N; i0 += 2) {                                            // pass 2: weights re-evaluated
                    f64x2 v[2], sdv[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (i0 + k < N) {
                            v[k] = ld2(static_cast<const double*>(a.frame[i0 + k]) + ibase);
                            sdv[k] = ld2(a.sd[i0 + k] + ibase);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (i0 + k < N) {
                            const double it = a.inv_t[i0 + k];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const double x = j == 0 ? v[k].x : v[k].y;
                                const double dv = x - 0.5;
                                pass2(i0 + k, j, x, gauss_weight(dv), j == 0 ? sdv[k].x : sdv[k].y, it);
                            }
                        }
                    }
                }
            }
            double val[2], so[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { val[j] = acc[j] / S[j]; so[j] = sqrt(var[j]); }
            if constexpr (FLAT) {
                flat_field_math(F[0], iF2[0], sF[0], a.ff_mean[c0], a.ff_std_mean[c0], true, val[0], so[0]);
                flat_field_math(F[1], iF2[1], sF[1], a.ff_mean[c1], a.ff_std_mean[c1], true, val[1], so[1]);
            }
            if constexpr (SUMW) store2(a.out_sum_w + sbase, lane16, S[0], S[1]);
            store2(a.out_val + sbase, lane16, val[0], val[1]);
            store2(a.out_std + sbase, lane16, so[0], so[1]);
        }
    }
}

// The two entry points differ in the occupancy the register allocator is asked for (A/B on one box, tools/ab3.sh):
// val-only runs best when it may use up to 168 VGPRs (3 waves/SIMD: more loads in flight per wave, 665 -> 599 us on
// 7 x 4096 x 4096 x 3); the std kernel ran best at 4 waves/SIMD while it re-evaluated its weights in pass 2 (127 VGPRs + 20 B
// scratch, 1 382 -> 1 343 us) and at 3 waves/SIMD now that it keeps them (HM_F64_KEEP_W; at 4 waves the kept weights spill: 1 566 us).
#ifndef HM_F64_VAL_WAVES
#define HM_F64_VAL_WAVES 3
#endif
#ifndef HM_F64_STD_WAVES
#define HM_F64_STD_WAVES 3
#endif
template <int C, bool FLAT, bool SUMW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HM_F64_VAL_WAVES, HM_F64_VAL_WAVES))) void merge_f64_val(const MergeK a) {
    merge_f64_body<C, false, FLAT, SUMW>(a);
}
template <int C, bool FLAT, bool SUMW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HM_F64_STD_WAVES, HM_F64_STD_WAVES))) void merge_f64_std(const MergeK a) {
    merge_f64_body<C, true, FLAT, SUMW>(a);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// hm_merge_describe(): the dispatch below runs with this pointer set and every launch site records its kernel's name
// instead of launching - so the description cannot drift from what hm_merge really dispatches to.
static thread_local std::string* g_describe = nullptr;
static bool describe_only(const char* fmt, int a = 0, int b = 0, int c = 0, int d = 0) {
    if (!g_describe) return false;
    char buf[160];
    snprintf(buf, sizeof buf, fmt, a, b, c, d);
    if (!g_describe->empty()) *g_describe += " + ";
    *g_describe += buf;
    return true;
}

// experiment knobs of the tools/ A/B scripts, read from the environment in TUNING BUILDS only (-DHM_TUNE_NF=<N> or -DHM_TUNE_ENV): the shipped
// library never calls getenv. 0 = not set; values are clamped to [lo, hi] (a bmax above 64 or a per-CU count of 0 must not reach a launch).
static int tune_env(const char* name, int lo, int hi) {
#if HM_TUNE_NF != 0 || defined(HM_TUNE_ENV)
    const char* v = getenv(name);
    const int e = v ? atoi(v) : 0;
    if (e == 0) return 0;
    return e < lo ? lo : (e > hi ? hi : e);
#else
    (void)name; (void)lo; (void)hi;
    return 0;
#endif
}

static int g_cu_count = 0;
int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            g_cu_count = p.multiProcessorCount;
        else
            g_cu_count = kCUs;
    }
    return g_cu_count;
}

// Variant encoding (args->variant; values other than 0 are meaningful in tuning builds only):
//   variant = 1000 * TAB + 100 * PREFETCH + 10 * U + BLOCK_CODE     U in {2,4,8}; BLOCK_CODE: 0 -> 256 threads, 1 -> 1024
struct FastCfg { int tab, u, prefetch, block; };

// Units per group (U sub-units of 128 elements) of the production configurations:
constexpr int kUVal = 2;     // val-only: 256 contiguous bytes per frame per wave iteration, next group prefetched
                             // (132-136 us on every box seen; without prefetch 135-146 us depending on the box)
constexpr int kUStd = 1;     // with std: one 128-element sub-unit per iteration + prefetch (tune: 684 us vs 753 at U = 2, no prefetch)

static FastCfg default_cfg(bool with_std) {
    if (with_std) return FastCfg{TAB_PLAIN, kUStd, 1, 256};
    return FastCfg{TAB_FUSED, kUVal, 1, 256};     // tools/tune_merge.py, tools/mergelab.hip, profiles/
}

static bool decode_variant(int variant, bool with_std, FastCfg& c) {
    c = default_cfg(with_std);
    if (variant <= 0) return true;
    const int tab = variant / 1000, pf = (variant / 100) % 10, u = (variant / 10) % 10, bc = variant % 10;
    if (with_std) {                                   // std kernel: only U (1, 2, 4) and PREFETCH are tunable
        if (pf > 1 || (u != 1 && u != 2 && u != 4)) return false;
        c.u = u; c.prefetch = pf;
        return true;
    }
#ifndef HM_PROBE
    if (tab == TAB_NONE) return false;                // the table-free traffic probe (wrong results) exists in -DHM_PROBE builds only
#endif
    if ((tab != TAB_PLAIN && tab != TAB_FUSED && tab != TAB_NONE) || pf > 1 || (u != 2 && u != 4 && u != 8) || bc > 1) return false;
    c.tab = tab; c.u = u; c.prefetch = pf; c.block = bc ? 1024 : 256;
    return true;
}

template <int NF, int U, int TAB, bool STD, bool PF, bool FLAT, bool SUMW, int BLOCK, int CH = 3>
static int launch_one(const MergeK& k, hipStream_t st) {
    constexpr int lds = (STD ? kStdTabBytes : TabInfo<TAB>::bytes) + (FLAT ? 16 * 256 : 0);
    void (*kernel)(const MergeK);
    if constexpr (STD) kernel = merge_u8_fast_std<NF, U, TAB, PF, FLAT, SUMW, BLOCK, CH>;
    else kernel = merge_u8_fast<NF, U, TAB, false, PF, FLAT, SUMW, BLOCK>;
    if (describe_only(CH == 1 ? "merge_u8_fast_std<N=%d,U=%d,flat=%d,sum_w=%d,C=1>"
                              : (STD ? "merge_u8_fast_std<N=%d,U=%d,flat=%d,sum_w=%d>" : "merge_u8_fast<N=%d,U=%d,flat=%d,sum_w=%d>"), NF, U, FLAT, SUMW)) return HM_OK;
    int per_cu = 2048 / BLOCK;                       // 32 waves per CU
    if (kMaxLds / lds < per_cu) per_cu = kMaxLds / lds;
    if (const int e = tune_env("HM_TUNE_WG_PER_CU", 1, 64)) per_cu = e;
    const int64_t groups = k.n_elems / (U * static_cast<int>(kSub));
    const unsigned grid = stream_grid(groups, BLOCK / 64, per_cu);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, st, k);
    return launch_status();
}

// the val-only, no-extras configuration runs merge_u8_val3 unless a variant asks for the older merge_u8_fast (A/B runs).
// variant 7UPM (tuning builds, N == HM_TUNE_NF only) selects merge_u8_val3<NF, U, PF, MAP>; 0 = the production choice.
struct Val3Cfg { int u, pf, map; };
// Production choice per frame count (A/B on one box, tools/ab_val3.py): up to 8 frames <U=2, PF=1> - 64 VGPRs, 125.5 us on config 2
// against 130.0 us for merge_u8_fast and 133-135 us for the in-place refill (profiles/r02_ab_val3_matrix2.json); above that the second
// register set costs occupancy (N = 15: 137 VGPRs) and the in-place refill with U = 3 wins (82 VGPRs; config-4 tile 105.0 us against
// 109.3 us for both <2, 1> and merge_u8_fast, profiles/r02_ab_val3_n15.json).
// Late round 2: U = 4 with the second register set (115 VGPRs at N = 7) beats U = 2 by 1.3-1.9 us on three boxes, fast and slow (same-process A/B,
// shared outputs: 130.2 / 130.3 / 136.4 against 132.1 / 132.0 / 137.9 us, profiles/r02h_ab_val3_u4.log) - 512 contiguous bytes per frame and 4 KB of
// output per wave iteration; U = 5, 6, 8 and issuing a unit's stores together at its end (MAP = 2) are slower.
// Round 3: software-pipelined gathers (MAP = 3: the ds_read_b128 of the next bundle of 4 frames in flight under the current bundle's add / fma
// chain, two sets of gathered values, 158 VGPRs = 3 waves/SIMD at N = 7) are 2.0 us faster than MAP = 0 in three same-process A/B runs with
// shared output buffers on two boxes (129.25 / 126.1 / 128.5 against 131.3 / 128.1 / 130.2 us, profiles/r03_ab_pipe_n7_box*.json) and make no
// difference for N = 15 (<3, 0, 3> 105.1 against 105.2 us at 133 instead of 82 VGPRs; profiles/r03_ab_pipe_n15_box1.json), which keeps MAP = 0.
// N = 8 takes three sub-units per wave instead of four: 73.0 against 77.9 us on 2048 x 4096 x 3 stacks (0.689 against 0.646; a sweep of every
// frame count with a dip in tools/bench_n.py - 6, 8, 10, 16 - late in round 4: profiles/r04x_sweep8.log, r04x_sweep6_10_16.log; the others
// are within 3 % of their best variant as they stand).
// N = 9 and N = 10 take two sub-units (with / without the second register set): 78.8 against 81.2 us and 84.8 against 89.4 us
// (profiles/r04x_sweep9_14.log); 11-14 are within 1 % of their best variant with the shape of N = 15.
constexpr Val3Cfg val3_default(int n_frames) {
    return n_frames <= 7 ? Val3Cfg{4, 1, 3} : n_frames == 8 ? Val3Cfg{3, 1, 3} : n_frames == 9 ? Val3Cfg{2, 1, 0} : n_frames == 10 ? Val3Cfg{2, 0, 0}
                                                                                                                               : Val3Cfg{3, 0, 0};
}
static bool val3_variant(int variant, int n_frames, Val3Cfg& c) {
    c = val3_default(n_frames);
    if (variant == 0) return true;
    variant %= 10000;                                     // (tuning builds: W * 10000 + 7UPM also sets the workgroups per CU, launch_val3_cfg)
    if (variant < 7000 || variant >= 8000 || n_frames != HM_TUNE_NF) return false;
    c.u = (variant / 100) % 10; c.pf = (variant / 10) % 10; c.map = variant % 10;
    return c.u >= 1 && c.u <= 8 && c.pf <= 1 && c.map <= 3;
}
#if HM_TUNE_NF != 0
#include "../../tools/hm_merge_priv_launch.inc"
#else
static bool priv_variant(int, int, int&) { return false; }       // merge_u8_priv exists in tuning builds only
#endif

static bool use_val3(int variant, int n_frames, bool with_std, bool extras) {
    Val3Cfg c;
    return !with_std && !extras && val3_variant(variant, n_frames, c);
}
// val-only WITH a uint8 flat field (no sum-of-weights output, library default variant): merge_u8_val3's FLAT instantiation
constexpr Val3Cfg val3_flat_default(int n_frames) { return n_frames <= 8 ? Val3Cfg{HM_VAL3_FLAT_U, 1, 3} : Val3Cfg{2, 1, 0}; }
// monochrome stacks (C == 1), val-only, at most a uint8 flat field, library default variant: merge_u8_val3's CH = 1 instantiations
static bool use_val3_mono(const MergeK& k, bool with_std, bool f64in) {
    return !f64in && k.C == 1 && !with_std && !k.out_sum_w && k.variant == 0 && k.n_frames <= 16 && (!k.has_flat || k.flat_u8);
}
// ... and with std (with or without a flat field): merge_u8_fast_std's CH = 1 instantiations
static bool use_fast_std_mono(const MergeK& k, bool with_std, bool f64in) {
    return !f64in && k.C == 1 && with_std && !k.out_sum_w && k.variant == 0 && k.n_frames <= 16;
}
static bool use_val3_flat(const MergeK& k, bool with_std) {
    return !with_std && k.has_flat && k.flat_u8 && !k.out_sum_w && k.variant == 0 && k.n_frames <= 16;
}
static int val3_unit_elems(const Val3Cfg& c) { return c.u * (c.map == 1 ? 4 : 1) * static_cast<int>(kSub); }

template <int NF, int U, int PF, int MAP, bool FLAT = false, int CH = 3>
static int launch_val3_cfg(const MergeK& k, hipStream_t st) {
    const int64_t units = k.n_elems / (U * (MAP == 1 ? 4 : 1) * static_cast<int>(kSub));
    // workgroups per CU (8 are resident). U = 2 (N <= 8): 12 - a grid of 3072 is a multiple of 3 as it stands (2048 had to become 2046) and
    // config 2's 196 608 units divide evenly among its waves; same-process A/B with shared output buffers (tools/ab_val3.py, variants
    // W * 10000 + 7210 of a tuning build, profiles/r02h_ab_val3_wg_per_cu.log): 134.9 against 136.7 us (8) on a slow box, 129.0 against 129.9 us
    // on a fast one; 2, 3, 4, 6, 16, 24, 32 are no better. U = 3 keeps 8 (config 4's tile: 65 536 units = 8 per wave).
#ifndef HM_VAL3_WG_PER_CU
#define HM_VAL3_WG_PER_CU (U % 3 != 0 ? 12 : 8)
#endif
    int wg_per_cu = HM_VAL3_WG_PER_CU;
    if (HM_TUNE_NF != 0 && k.variant >= 10000) wg_per_cu = k.variant / 10000;
    unsigned grid = MAP == 1 ? static_cast<unsigned>(units < cu_count() * 8 ? units : cu_count() * 8) : stream_grid(units, 4, wg_per_cu);   // 8 workgroups of 4 waves per CU
    if (U % 3 != 0 && grid >= 3) grid -= grid % 3;          // the unit index must advance by a multiple of 3 per iteration (see the kernel)
    if (grid == 0) grid = 1;
    if (describe_only(CH == 1 ? (FLAT ? "merge_u8_val3<N=%d,U=%d,PF=%d,MAP=%d,flat=1,C=1>" : "merge_u8_val3<N=%d,U=%d,PF=%d,MAP=%d,C=1>")
                              : (FLAT ? "merge_u8_val3<N=%d,U=%d,PF=%d,MAP=%d,flat=1>" : "merge_u8_val3<N=%d,U=%d,PF=%d,MAP=%d>"), NF, U, PF, MAP)) return HM_OK;
    hipLaunchKernelGGL((merge_u8_val3<NF, U, PF, MAP, FLAT, CH>), dim3(grid), dim3(256), 0, st, k);
    return launch_status();
}
template <int NF>
static int launch_val3(const MergeK& k, hipStream_t st) {
    Val3Cfg c;
    if (!val3_variant(k.variant, NF, c)) return HM_EINVAL;
    if constexpr (NF == HM_TUNE_NF) {
        switch (c.u * 100 + c.pf * 10 + c.map) {
            case 300: return launch_val3_cfg<NF, 3, 0, 0>(k, st);
            case 301: return launch_val3_cfg<NF, 3, 0, 1>(k, st);
            case 310: return launch_val3_cfg<NF, 3, 1, 0>(k, st);
            case 311: return launch_val3_cfg<NF, 3, 1, 1>(k, st);
            case 200: return launch_val3_cfg<NF, 2, 0, 0>(k, st);
            case 201: return launch_val3_cfg<NF, 2, 0, 1>(k, st);
            case 210: return launch_val3_cfg<NF, 2, 1, 0>(k, st);
            case 211: return launch_val3_cfg<NF, 2, 1, 1>(k, st);
            case 100: return launch_val3_cfg<NF, 1, 0, 0>(k, st);
            case 212: return launch_val3_cfg<NF, 2, 1, 2>(k, st);
            case 312: return launch_val3_cfg<NF, 3, 1, 2>(k, st);
            case 412: return launch_val3_cfg<NF, 4, 1, 2>(k, st);
            case 410: return launch_val3_cfg<NF, 4, 1, 0>(k, st);
            case 413: return launch_val3_cfg<NF, 4, 1, 3>(k, st);
            case 213: return launch_val3_cfg<NF, 2, 1, 3>(k, st);
            case 303: return launch_val3_cfg<NF, 3, 0, 3>(k, st);
            case 313: return launch_val3_cfg<NF, 3, 1, 3>(k, st);
            case 402: return launch_val3_cfg<NF, 4, 0, 2>(k, st);
            case 400: return launch_val3_cfg<NF, 4, 0, 0>(k, st);
            case 510: return launch_val3_cfg<NF, 5, 1, 0>(k, st);
            case 610: return launch_val3_cfg<NF, 6, 1, 0>(k, st);
            case 810: return launch_val3_cfg<NF, 8, 1, 0>(k, st);
            case 600: return launch_val3_cfg<NF, 6, 0, 0>(k, st);
            case 800: return launch_val3_cfg<NF, 8, 0, 0>(k, st);
            default:  return launch_val3_cfg<NF, 1, 1, 0>(k, st);
        }
    } else {
        return launch_val3_cfg<NF, val3_default(NF).u, val3_default(NF).pf, val3_default(NF).map>(k, st);
    }
}

template <int NF, int U, int TAB, bool PF>
static int launch_val_blk(const MergeK& k, const FastCfg& c, hipStream_t st) {
    if (c.block == 256) return launch_one<NF, U, TAB, false, PF, false, false, 256>(k, st);
    return launch_one<NF, U, TAB, false, PF, false, false, 1024>(k, st);
}

template <int NF, int U, bool PF>
static int launch_val_tab(const MergeK& k, const FastCfg& c, hipStream_t st) {
    switch (c.tab) {
        case TAB_PLAIN: return launch_val_blk<NF, U, TAB_PLAIN, PF>(k, c, st);
        case TAB_FUSED: return launch_val_blk<NF, U, TAB_FUSED, PF>(k, c, st);
#ifdef HM_PROBE
        default:        return launch_val_blk<NF, U, TAB_NONE, PF>(k, c, st);
#else
        default:        return HM_EINVAL;
#endif
    }
}

template <int NF, bool STD, int U, int TAB>
static int launch_extras(const MergeK& k, hipStream_t st) {
    const bool flat = k.has_flat != 0, sumw = k.out_sum_w != nullptr;
    if (flat && sumw) return launch_one<NF, U, TAB, STD, true, true, true, 256>(k, st);
    if (flat) return launch_one<NF, U, TAB, STD, true, true, false, 256>(k, st);
    return launch_one<NF, U, TAB, STD, true, false, true, 256>(k, st);
}

template <int NF>
static int launch_fast_nf(const MergeK& k, const FastCfg& c, bool with_std, hipStream_t st) {
    const bool extras = k.has_flat || k.out_sum_w;
    if (with_std) {
        if (use_fast_std_mono(k, true, false)) {
            if (k.has_flat) return launch_one<NF, kUStd, TAB_PLAIN, true, true, true, false, 256, 1>(k, st);
            return launch_one<NF, kUStd, TAB_PLAIN, true, true, false, false, 256, 1>(k, st);
        }
        if (extras) return launch_extras<NF, true, kUStd, TAB_PLAIN>(k, st);
        if constexpr (NF == HM_TUNE_NF) {
            if (c.prefetch) {
                if (c.u == 1) return launch_one<NF, 1, TAB_PLAIN, true, true, false, false, 256>(k, st);
                if (c.u == 2) return launch_one<NF, 2, TAB_PLAIN, true, true, false, false, 256>(k, st);
                return launch_one<NF, 4, TAB_PLAIN, true, true, false, false, 256>(k, st);
            }
            if (c.u == 1) return launch_one<NF, 1, TAB_PLAIN, true, false, false, false, 256>(k, st);
            if (c.u == 2) return launch_one<NF, 2, TAB_PLAIN, true, false, false, false, 256>(k, st);
            return launch_one<NF, 4, TAB_PLAIN, true, false, false, false, 256>(k, st);
        }
        return launch_one<NF, kUStd, TAB_PLAIN, true, true, false, false, 256>(k, st);
    }
    if (use_val3_mono(k, false, false)) {
        if (k.has_flat) return launch_val3_cfg<NF, val3_flat_default(NF).u, val3_flat_default(NF).pf, val3_flat_default(NF).map, true, 1>(k, st);
        return launch_val3_cfg<NF, val3_default(NF).u, val3_default(NF).pf, val3_default(NF).map, false, 1>(k, st);
    }
    if (use_val3_flat(k, false))
        return launch_val3_cfg<NF, val3_flat_default(NF).u, val3_flat_default(NF).pf, val3_flat_default(NF).map, true>(k, st);
    if (extras) return launch_extras<NF, false, kUVal, TAB_FUSED>(k, st);
#if HM_TUNE_NF != 0
    if constexpr (NF == HM_TUNE_NF) {
        int pu = 0;
        if (priv_variant(k.variant, NF, pu)) return launch_priv<NF>(k, st);
    }
#endif
    if (use_val3(k.variant, NF, false, false)) return launch_val3<NF>(k, st);
    if constexpr (NF == HM_TUNE_NF) {
        if (c.prefetch) {
            if (c.u == 2) return launch_val_tab<NF, 2, true>(k, c, st);
            if (c.u == 4) return launch_val_tab<NF, 4, true>(k, c, st);
            return launch_val_tab<NF, 8, true>(k, c, st);
        }
        if (c.u == 2) return launch_val_tab<NF, 2, false>(k, c, st);
        if (c.u == 4) return launch_val_tab<NF, 4, false>(k, c, st);
        return launch_val_tab<NF, 8, false>(k, c, st);
    } else {
        return launch_one<NF, kUVal, TAB_FUSED, false, true, false, false, 256>(k, st);
    }
}

// elements per group of the configuration launch_fast_nf() will really use
static int fast_group_elems(int n_frames, int variant, const FastCfg& c, bool with_std, bool extras, bool val3_flat) {
    if (val3_flat) return val3_unit_elems(val3_flat_default(n_frames));
    { int pu = 0; if (!with_std && !extras && priv_variant(variant, n_frames, pu)) return pu * 120; }     // merge_u8_priv's chunks (tuning builds)
    { Val3Cfg vc; if (!with_std && !extras && val3_variant(variant, n_frames, vc)) return val3_unit_elems(vc); }
    if (with_std) return ((extras || n_frames != HM_TUNE_NF) ? kUStd : c.u) * static_cast<int>(kSub);
    if (extras || n_frames != HM_TUNE_NF) return kUVal * static_cast<int>(kSub);
    return c.u * static_cast<int>(kSub);
}

template <int C>
static int launch_loop_c(const MergeK& k, bool with_std, hipStream_t st) {
    const bool flat = k.has_flat != 0, sumw = k.out_sum_w != nullptr;
    const int lds = (with_std ? 16 * 256 + 16 * 256 * C : 16 * 256 * C) + (flat ? 16 * 256 : 0);
    const unsigned grid = stream_grid(k.n_elems / static_cast<int>(kSub), 4, 8);
    if (describe_only(with_std ? "merge_u8_loop_std<C=%d,flat=%d,sum_w=%d>(N=%d)" : "merge_u8_loop<C=%d,flat=%d,sum_w=%d>(N=%d)", C, flat, sumw, k.n_frames)) return HM_OK;
#define HM_LOOP(K, F, W) hipLaunchKernelGGL((K<C, F, W>), dim3(grid), dim3(256), lds, st, k)
    if (with_std) {
        if (flat && sumw) HM_LOOP(merge_u8_loop_std, true, true); else if (flat) HM_LOOP(merge_u8_loop_std, true, false);
        else if (sumw) HM_LOOP(merge_u8_loop_std, false, true); else HM_LOOP(merge_u8_loop_std, false, false);
    } else {
        if (flat && sumw) HM_LOOP(merge_u8_loop, true, true); else if (flat) HM_LOOP(merge_u8_loop, true, false);
        else if (sumw) HM_LOOP(merge_u8_loop, false, true); else HM_LOOP(merge_u8_loop, false, false);
    }
#undef HM_LOOP
    return launch_status();
}

template <int C>
static int launch_f64_c(const MergeK& k, bool with_std, hipStream_t st) {
    const bool flat = k.has_flat != 0, sumw = k.out_sum_w != nullptr;
    const int lds = 16 * 256 * C + (flat ? 16 * 256 : 0);
    const unsigned grid = stream_grid(k.n_elems / static_cast<int>(kSub), 4, 8);
    if (describe_only(with_std ? "merge_f64_std<C=%d,flat=%d,sum_w=%d>(N=%d)" : "merge_f64_val<C=%d,flat=%d,sum_w=%d>(N=%d)", C, flat, sumw, k.n_frames)) return HM_OK;
#define HM_F64(K, F, W) hipLaunchKernelGGL((K<C, F, W>), dim3(grid), dim3(256), lds, st, k)
    if (with_std) {
        if (flat && sumw) HM_F64(merge_f64_std, true, true); else if (flat) HM_F64(merge_f64_std, true, false);
        else if (sumw) HM_F64(merge_f64_std, false, true); else HM_F64(merge_f64_std, false, false);
    } else {
        if (flat && sumw) HM_F64(merge_f64_val, true, true); else if (flat) HM_F64(merge_f64_val, true, false);
        else if (sumw) HM_F64(merge_f64_val, false, true); else HM_F64(merge_f64_val, false, false);
    }
#undef HM_F64
    return launch_status();
}

static int launch_f64(const MergeK& k, bool with_std, hipStream_t st) {
    switch (k.C) {
        case 1: return launch_f64_c<1>(k, with_std, st);
        case 2: return launch_f64_c<2>(k, with_std, st);
        case 3: return launch_f64_c<3>(k, with_std, st);
        default: return launch_f64_c<4>(k, with_std, st);
    }
}

static int launch_loop(const MergeK& k, bool with_std, hipStream_t st) {
    switch (k.C) {
        case 1: return launch_loop_c<1>(k, with_std, st);
        case 2: return launch_loop_c<2>(k, with_std, st);
        case 3: return launch_loop_c<3>(k, with_std, st);
        default: return launch_loop_c<4>(k, with_std, st);
    }
}

static int launch_generic(const MergeK& k, bool f64in, bool with_std, hipStream_t st) {
    const unsigned grid = stream_grid(k.n_elems, 256, 8);
    if (describe_only("merge_generic<f64in=%d,std=%d>", f64in, with_std)) return HM_OK;
#define HM_GEN(F, S) hipLaunchKernelGGL((merge_generic<F, S>), dim3(grid), dim3(256), 0, st, k)
    if (f64in) { if (with_std) HM_GEN(true, true); else HM_GEN(true, false); }
    else       { if (with_std) HM_GEN(false, true); else HM_GEN(false, false); }
#undef HM_GEN
    return launch_status();
}

static int launch_fixup(const MergeK& k, bool f64in, bool with_std, hipStream_t st) {
    const int64_t chunks = (k.n_elems + 15) / 16;
    const unsigned grid = stream_grid(chunks, 256, 8);
    if (describe_only("merge_fixup_hot<f64in=%d,std=%d>", f64in, with_std)) return HM_OK;
#define HM_FIX(F, S) hipLaunchKernelGGL((merge_fixup_hot<F, S>), dim3(grid), dim3(256), 0, st, k)
    if (f64in) { if (with_std) HM_FIX(true, true); else HM_FIX(true, false); }
    else       { if (with_std) HM_FIX(false, true); else HM_FIX(false, false); }
#undef HM_FIX
    return launch_status();
}

// the queue path of the hot-pixel pass: zero the counters, scan the dark maps into the queue, patch the queued elements;
// on overflow merge_patch_hot goes over the whole tile instead of the queue (a queue that was too small costs time, never correctness)
static int launch_hot_queue(const MergeK& k, bool f64in, bool with_std, uint32_t* ws, size_t ws_bytes, hipStream_t st) {
    const size_t head = static_cast<size_t>(kHotQueueHeader) + 2u * hot_piece_slots(k.n_elems);     // counters + piece table, in words
    const size_t words = ws_bytes / 4 - head;
    const uint32_t capacity = static_cast<uint32_t>(words > 0xffffffffull ? 0xffffffffull : words);
    if (describe_only("merge_scan_hot")) {
        describe_only("merge_patch_hot<f64in=%d,std=%d>", f64in, with_std);
        return HM_OK;
    }
    // (the counters were zeroed by the streaming kernel of this call - MergeK::hot_reset; the scan writes every table entry it owns)
    const int64_t chunks = (k.n_elems + 15) / 16;
    hipLaunchKernelGGL(merge_scan_hot, dim3(stream_grid(chunks, kScanBlock, 1)), dim3(kScanBlock), 0, st, k, ws, capacity);
    int rc = launch_status();
    if (rc != HM_OK) return rc;
    // The grid is what the chip holds AT ONCE (the kernel's occupancy: 3-6 workgroups per CU, by registers): every workgroup is then
    // resident from the start and a sparse queue costs the latency of one element. A grid of 8 per CU ran its workgroups in three
    // rounds at 3 resident per CU - 57-65 us for 35 000 queued elements, each round the same dependent chain of memory round trips.
    const int keep = patch_keep_bytes(f64in, k.n_frames, k.median_k);
    const void* fn;
    if (f64in) fn = with_std ? reinterpret_cast<const void*>(merge_patch_hot<true, true>) : reinterpret_cast<const void*>(merge_patch_hot<true, false>);
    else       fn = with_std ? reinterpret_cast<const void*>(merge_patch_hot<false, true>) : reinterpret_cast<const void*>(merge_patch_hot<false, false>);
    // (the query costs a few microseconds of host time: remembered per kernel, LDS size and device, under a mutex - hm_merge may be called
    // from several host threads)
    static struct { const void* fn; int keep, dev, per_cu; } seen[16];
    static int n_seen = 0;
    static std::mutex seen_lock;
    int per_cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> hold(seen_lock);
        for (int i = 0; i < n_seen; ++i)
            if (seen[i].fn == fn && seen[i].keep == keep && seen[i].dev == dev) per_cu = seen[i].per_cu;
    }
    if (per_cu == 0) {
        // (gfx950 has 160 KB of LDS per workgroup; the kernel's 36 KB of tables + up to 32 KB of kept float64 frames fit. Should a device
        // report that not even one workgroup fits, -1 is remembered and the workspace-free pass runs instead - never a failing launch.)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, static_cast<size_t>(keep)) != hipSuccess) { (void)hipGetLastError(); per_cu = 2; }
        if (per_cu < 1) per_cu = -1;
        if (per_cu > 8) per_cu = 8;
        std::lock_guard<std::mutex> hold(seen_lock);
        if (n_seen < 16) { seen[n_seen].fn = fn; seen[n_seen].keep = keep; seen[n_seen].dev = dev; seen[n_seen].per_cu = per_cu; ++n_seen; }
    }
    if (per_cu < 0) return launch_fixup(k, f64in, with_std, st);
    if (const int e = tune_env("HM_TUNE_PATCH_WG_PER_CU", 1, 16)) per_cu = e;
    const unsigned grid = stream_grid(k.n_elems, 256, per_cu);
    uint32_t bmax = 64;
    if (const int e = tune_env("HM_TUNE_PATCH_BMAX", 1, 64)) bmax = static_cast<uint32_t>(e);
#define HM_PATCH(K, F, S) hipLaunchKernelGGL((K<F, S>), dim3(grid), dim3(256), keep, st, k, static_cast<const uint32_t*>(ws), bmax)
#define HM_PATCH4(K) { if (f64in) { if (with_std) HM_PATCH(K, true, true); else HM_PATCH(K, true, false); } \
                       else       { if (with_std) HM_PATCH(K, false, true); else HM_PATCH(K, false, false); } }
    HM_PATCH4(merge_patch_hot)
#undef HM_PATCH4
#undef HM_PATCH
    return launch_status();
}

}  // namespace hm

extern "C" int64_t hm_merge_algorithmic_bytes(const hm_merge_args* g) {
    if (!g || g->n_frames <= 0 || g->rows <= 0 || g->width <= 0 || g->channels <= 0) return 0;
    const int64_t E = g->rows * g->width * g->channels;
    const int N = g->n_frames;
    const bool s = g->stds != nullptr;
    const int64_t in_b = g->frames_f64 ? 8 : 1;
    int64_t per = N * (in_b + (s ? 8 : 0));
    if (g->out_val) per += 8 * (1 + (s ? 1 : 0));
    if (g->out_sum_w) per += 8;
    if (g->flat_u8 || g->flat_f64) per += (g->flat_u8 ? 1 : 8) + ((s && g->flat_std) ? 8 : 0);
    if (g->darks_u8)                                       // every DISTINCT (map, threshold) is read once (scan_chunk_hotbits)
        for (int i = 0; i < N; ++i) {
            if (!g->darks_u8[i]) continue;
            bool seen = false;
            for (int k = 0; k < i; ++k)
                seen = seen || (g->darks_u8[k] == g->darks_u8[i] && (!g->dark_min_dn || g->dark_min_dn[k] == g->dark_min_dn[i]));
            per += seen ? 0 : 1;
        }
    return per * E;
}

// Workspace of the hot-pixel queue for a call that produces n_elems = rows * W * C output elements: 16 bytes of counters, the piece
// table (8 bytes per 65 536 elements + 8 KB) and one uint32 per queued element. The recommended size holds a quarter of the elements
// (a dark map with 25 % hot pixels); a workspace with room for at least one entry (hm_merge_hot_workspace_min_bytes) is accepted - a
// queue that overflows makes the patch kernel go over the whole tile; a smaller one selects the workspace-free pass.
extern "C" size_t hm_merge_hot_workspace_min_bytes(int64_t n_elems) {
    if (n_elems < 1) n_elems = 1;
    return (static_cast<size_t>(hm::kHotQueueHeader) + 2u * hm::hot_piece_slots(n_elems) + 1u) * 4;
}
extern "C" size_t hm_merge_hot_workspace_bytes(int64_t n_elems) {
    if (n_elems < 1) n_elems = 1;
    int64_t entries = n_elems / 4;
    if (entries < 4096) entries = n_elems < 4096 ? n_elems : 4096;
    return hm_merge_hot_workspace_min_bytes(n_elems) + static_cast<size_t>(entries - 1) * 4;
}

extern "C" int hm_merge_describe(const hm_merge_args* g, char* buf, int buf_len) {
    if (!buf || buf_len < 1) return HM_EINVAL;
    std::string names;
    hm::g_describe = &names;
    const int rc = hm_merge(g, nullptr);
    hm::g_describe = nullptr;
    snprintf(buf, static_cast<size_t>(buf_len), "%s", names.c_str());
    return rc;
}

// struct_size values hm_merge accepts: the current layout and the two older ones of ABI version 1 (without the hot-pixel queue
// workspace: 264 bytes; with it: 280 bytes) - the missing tail reads as zero (no workspace).
static bool widen_args(const hm_merge_args* g, hm_merge_args& full) {
    if (!g) return false;
    const uint32_t sz = g->struct_size;
    if (sz != sizeof(hm_merge_args) && sz != 264u && sz != 280u) return false;
    full = hm_merge_args{};
    memcpy(&full, g, sz < sizeof(hm_merge_args) ? sz : sizeof(hm_merge_args));
    full.struct_size = sizeof(hm_merge_args);
    return true;
}

extern "C" int hm_merge(const hm_merge_args* g_in, void* stream) {
    using namespace hm;
    hm_merge_args full;
    if (!widen_args(g_in, full)) return HM_EINVAL;
    const hm_merge_args* g = &full;
    const int N = g->n_frames, C = g->channels;
    if (N < 1 || C < 1 || g->height < 1 || g->width < 1 || g->rows < 0) return HM_EINVAL;
    if (C > HM_MAX_CHANNELS) return HM_EUNSUPPORTED;
    if (g->rows == 0) return g->row0 >= 0 && g->row0 <= g->height ? HM_OK : HM_ESHAPE;   // empty tile: nothing to do
    const bool f64in = g->frames_f64 != nullptr;
    if (f64in == (g->frames_u8 != nullptr)) return HM_EINVAL;          // exactly one input kind
    if (!g->exposures || !g->icrf) return HM_EINVAL;
    const bool with_std = g->stds != nullptr;
    if (!g->out_val && !g->out_sum_w) return HM_EINVAL;
    if (g->out_val) {
        if (with_std != (g->out_std != nullptr)) return HM_EINVAL;
        if (with_std && !g->icrf_diff) return HM_EINVAL;
    }
    if (!f64in && (!g->w_lut || (with_std && !g->dw_lut))) return HM_EINVAL;
    const bool flat = g->flat_u8 || g->flat_f64;
    if (g->flat_u8 && g->flat_f64) return HM_EINVAL;
    if (flat && with_std && !g->flat_std) return HM_EINVAL;
    // geometry
    if (g->row0 < 0 || g->row0 + g->rows > g->height) return HM_ESHAPE;
    if (g->buf_row0 < 0 || g->buf_row0 > g->row0 || g->buf_row0 + g->buf_rows > g->height ||
        g->buf_row0 + g->buf_rows < g->row0 + g->rows) return HM_ESHAPE;
    bool hot = false;
    if (g->darks_u8) {
        if (!g->dark_min_dn) return HM_EINVAL;
        for (int i = 0; i < N; ++i) hot = hot || (g->darks_u8[i] != nullptr);
    }
    if (hot) {
        const int k = g->median_k;
        if (k < 3 || k > 7 || (k % 2) == 0) return HM_EINVAL;
        const int64_t r = k / 2;
        const int64_t need_lo = g->row0 - r < 0 ? 0 : g->row0 - r;
        const int64_t need_hi = g->row0 + g->rows + r > g->height ? g->height : g->row0 + g->rows + r;
        if (g->buf_row0 > need_lo || g->buf_row0 + g->buf_rows < need_hi) return HM_ESHAPE;   // halo too small
    }
    // per-frame pointers and exposures (any N: the chunked path below has no frame limit)
    for (int i = 0; i < N; ++i) {
        const void* f = f64in ? static_cast<const void*>(g->frames_f64[i]) : static_cast<const void*>(g->frames_u8[i]);
        if (!f) return HM_EINVAL;
        if (f64in && !aligned(f, 8)) return HM_EALIGN;
        if (with_std) {
            if (!g->stds[i]) return HM_EINVAL;
            if (!aligned(g->stds[i], 8)) return HM_EALIGN;
        }
        if (!(g->exposures[i] > 0.0)) return HM_EINVAL;
    }
    if ((g->out_val && !aligned(g->out_val, 8)) || (g->out_std && !aligned(g->out_std, 8)) ||
        (g->out_sum_w && !aligned(g->out_sum_w, 8))) return HM_EALIGN;
    // More frames than one launch takes (modules/exposure_series.py:334,372 have no limit): HM_MAX_FRAMES per launch with the running
    // sums in memory (hm_merge_chunk.hip). variant <= -2 forces that path with -variant frames per chunk (tests: any chunking gives the
    // bits of the one-launch kernels).
    if (N > HM_MAX_FRAMES || g->variant <= -2) {
        int chunk = g->variant <= -2 ? -g->variant : HM_MAX_FRAMES;
        if (chunk > HM_MAX_FRAMES) chunk = HM_MAX_FRAMES;
        return merge_chunked(g, chunk, g_describe, as_stream(stream));
    }
    // The streaming kernels index groups and buffer offsets with 32 bits: a tile of 2^32 elements or more is merged as consecutive
    // row bands of fewer than 2^32 elements each (same buffers, row0 / rows / output pointers advanced; an even number of rows per band
    // keeps every band's first byte 2-byte aligned). Only a single row of >= 2^32 elements is left to merge_generic.
    {
        const int64_t per_row = g->width * static_cast<int64_t>(C);
        const int64_t limit = (int64_t{1} << 32) - 1;
        if (g->rows * per_row > limit && per_row <= limit / 2 && g->variant >= 0) {
            int64_t band_rows = limit / per_row;
            if (band_rows > 1) band_rows &= ~int64_t{1};
            for (int64_t r = 0; r < g->rows; r += band_rows) {
                hm_merge_args b = *g;
                b.row0 = g->row0 + r;
                b.rows = g->rows - r < band_rows ? g->rows - r : band_rows;
                const int64_t adv = r * per_row;                       // outputs and the flat field cover the OUTPUT rows
                if (g->out_val) b.out_val = g->out_val + adv;
                if (g->out_std) b.out_std = g->out_std + adv;
                if (g->out_sum_w) b.out_sum_w = g->out_sum_w + adv;
                if (g->flat_u8) b.flat_u8 = g->flat_u8 + adv;
                if (g->flat_f64) b.flat_f64 = g->flat_f64 + adv;
                if (g->flat_std) b.flat_std = g->flat_std + adv;
                const int rc_b = hm_merge(&b, stream);
                if (rc_b != HM_OK) return rc_b;
            }
            return HM_OK;
        }
    }
    MergeK k{};
    for (int i = 0; i < N; ++i) {
        k.frame[i] = f64in ? static_cast<const void*>(g->frames_f64[i]) : static_cast<const void*>(g->frames_u8[i]);
        if (with_std) k.sd[i] = g->stds[i];
        k.inv_t[i] = 1.0 / g->exposures[i];
        k.dark[i] = hot ? g->darks_u8[i] : nullptr;
        k.dark_min[i] = hot && g->dark_min_dn ? g->dark_min_dn[i] : 256;
    }
    k.icrf = g->icrf; k.icrf_diff = g->icrf_diff; k.w_lut = g->w_lut; k.dw_lut = g->dw_lut;
    k.flat_u8 = g->flat_u8; k.flat_f64 = g->flat_f64; k.flat_std = g->flat_std;
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) { k.ff_mean[c] = g->ff_mean[c]; k.ff_std_mean[c] = g->ff_std_mean[c]; }
    k.out_val = g->out_val; k.out_std = g->out_std; k.out_sum_w = g->out_sum_w;
    const int64_t E = g->rows * g->width * C;
    k.n_elems = E; k.elem0 = 0;
    k.in_off = (g->row0 - g->buf_row0) * g->width * C;
    k.H = g->height; k.W = g->width; k.row0 = g->row0; k.buf_row0 = g->buf_row0; k.buf_rows = g->buf_rows;
    k.n_frames = N; k.C = C; k.median_k = hot ? g->median_k : 3; k.has_flat = flat ? 1 : 0;
    k.variant = g->variant;
    k.inv_t_inrange = 1;
    for (int i = 0; i < N; ++i) k.inv_t_inrange = k.inv_t_inrange && k.inv_t[i] >= 0x1p-300 && k.inv_t[i] <= 0x1p300;
    hipStream_t st = as_stream(stream);
    const bool hot_queue = hot && g->hot_workspace && g->hot_workspace_bytes >= hm_merge_hot_workspace_min_bytes(g->rows * g->width * C) &&
                           aligned(g->hot_workspace, 16) && g->rows * g->width * C < (int64_t{1} << 32);
    k.hot_reset = hot_queue ? static_cast<uint32_t*>(g->hot_workspace) : nullptr;

    // ---- streaming pass: fast kernel where eligible, generic kernel otherwise (dark maps are not read here)
    FastCfg cfg;
    const int tune_variant = (HM_TUNE_NF != 0 && g->variant >= 10000) ? g->variant % 10000 : g->variant;   // W * 10000 + 7UPM (tuning builds)
    if (tune_variant >= 7000 && tune_variant < 9000) {        // merge_u8_val3 / merge_u8_priv A/B variants (tuning builds)
        Val3Cfg vc;
        int pu = 0;
        const bool ok = tune_variant < 8000 ? val3_variant(g->variant, N, vc) : priv_variant(g->variant, N, pu);
        if (!ok || with_std || flat || g->out_sum_w || f64in || C != 3) return HM_EINVAL;
        cfg = default_cfg(with_std);
    } else if (!decode_variant(g->variant, with_std, cfg)) return HM_EINVAL;
    bool fast = g->out_val && E < (int64_t{1} << 32) && g->variant >= 0;
    const bool mono_val3 = use_val3_mono(k, with_std, f64in);
    const bool mono_std = use_fast_std_mono(k, with_std, f64in);
    // Templates up to kTemplatedN = 20 frames (round 4, late: 16 before). On 2048 x 4096 x 3 stacks the templated kernels beat the run-time-N
    // kernel at N = 17 (val-only 115 against 130 us, with std 790 against 880 us) and N = 20 (132 against 150, 985 against 1 025 us) and lose from
    // N = 24 (with std) / 32 (val-only) on, where their per-frame register arrays no longer fit (profiles/r04z_n32_templates_ab.log).
    const bool loop_kernel = (f64in || N > kTemplatedN || C != 3) && !mono_val3 && !mono_std;   // run-time-N / any-C streaming kernel instead of the templates
    if (fast) {
        for (int i = 0; i < N && fast; ++i) {
            fast = f64in ? aligned(static_cast<const double*>(k.frame[i]) + k.in_off, 16)
                         : aligned(static_cast<const uint8_t*>(k.frame[i]) + k.in_off, 2);
            if (fast && with_std) fast = aligned(k.sd[i] + k.in_off, 16);
        }
        fast = fast && aligned(k.out_val, 16) && (!k.out_std || aligned(k.out_std, 16)) &&
               (!k.out_sum_w || aligned(k.out_sum_w, 16));
        if (fast && flat)
            fast = (k.flat_u8 ? aligned(k.flat_u8, 2) : aligned(k.flat_f64, 16)) && (!with_std || aligned(k.flat_std, 16));
    }
    int rc = HM_OK;
    if (!fast) {
        rc = launch_generic(k, f64in, with_std, st);
    } else {
        const int64_t grp = loop_kernel ? static_cast<int64_t>(kSub) : (mono_val3 ? val3_unit_elems(flat ? val3_flat_default(N) : val3_default(N))
                                                                                                 : fast_group_elems(N, g->variant, cfg, with_std, flat || g->out_sum_w, use_val3_flat(k, with_std)));
        const int64_t body = (E / grp) * grp;
        if (body > 0) {
            MergeK kb = k;
            kb.n_elems = body;
            switch (loop_kernel ? 0 : N) {
#define HM_CASE(n) case n: rc = launch_fast_nf<n>(kb, cfg, with_std, st); break;
                HM_CASE(1) HM_CASE(2) HM_CASE(3) HM_CASE(4) HM_CASE(5) HM_CASE(6) HM_CASE(7) HM_CASE(8)
                HM_CASE(9) HM_CASE(10) HM_CASE(11) HM_CASE(12) HM_CASE(13) HM_CASE(14) HM_CASE(15) HM_CASE(16)
                HM_CASE(17) HM_CASE(18) HM_CASE(19) HM_CASE(20)
                static_assert(kTemplatedN == 20, "one HM_CASE per templated frame count");
#undef HM_CASE
                default: rc = f64in ? launch_f64(kb, with_std, st) : launch_loop(kb, with_std, st); break;   // run-time frame count
            }
            if (rc != HM_OK) return rc;
        }
        if (body < E) {                                    // tail: less than one group
            MergeK kt = k;
            kt.elem0 = body; kt.n_elems = E - body;
            rc = launch_generic(kt, f64in, with_std, st);
        }
    }
    if (rc != HM_OK) return rc;
    // ---- hot-pixel fix-up pass (stream-ordered after the streaming pass: it overwrites the affected elements)
    if (hot) {
        rc = hot_queue ? launch_hot_queue(k, f64in, with_std, static_cast<uint32_t*>(g->hot_workspace), g->hot_workspace_bytes, st)
                   : launch_fixup(k, f64in, with_std, st);
    }
    return rc;
}
