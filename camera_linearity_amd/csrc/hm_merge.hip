// hm_merge.hip - fused HDR merge for gfx950 (MI355X).
//
// Replaces, in one launch, the two passes of the reference's merge loop
// (modules/exposure_series.py:317-345 `_precalculate_sum_of_weights` and :347-397
// `_compute_HDR_image_set`) together with the per-frame arithmetic they call:
// apply_gaussian_weight (modules/measurand.py:606-618), linearize (:471-541), the optional hot-pixel
// median (:543-557) and the optional flat-field normalisation (:559-604).
//
// The path is HBM-bandwidth bound (no MFMA): every input byte is read once with wide coalesced loads,
// every output byte written once. Per (element, frame) the only arithmetic is a table gather, because
// for 8-bit frames both the Gaussian weight and the ICRF depend on the DN alone; the 256-entry tables
// (w, w*g per channel, ...) live in LDS. With N = 7 val-only frames the LDS gather rate, not HBM, is
// the first ceiling, so the val-only kernel replicates the tables across LDS banks (TAB_* below).
//
// Two kernels:
//   merge_u8_fast   C == 3, N <= 16 (compile-time), 2-byte-aligned uint8 frames; the bench path.
//   merge_generic   anything else (float64 frames, other C, N <= 32, tails, unaligned tiles):
//                   one element per thread, same arithmetic.
#include "hm_common.h"

namespace hm {

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
};

// ------------------------------------------------------------------------------------------------
// rare path: k x k median of one channel around one pixel, 'reflect' at the true image edges.
// Selection by counting (no local arrays -> no scratch): the median is the sample v with
// #(x < v) <= m < #(x <= v), m = k*k/2.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __noinline__ T median_at(const T* __restrict__ buf, int64_t H, int64_t W, int C,
                                    int64_t buf_row0, int64_t row, int64_t col, int c, int k) {
    const int r = k / 2;
    const int m = (k * k) / 2;
    T best = buf[((row - buf_row0) * W + col) * C + c];
    for (int py = -r; py <= r; ++py) {
        const int64_t yy = reflect_index(row + py, H) - buf_row0;
        for (int px = -r; px <= r; ++px) {
            const int64_t xx = reflect_index(col + px, W);
            const T v = buf[(yy * W + xx) * C + c];
            int less = 0, leq = 0;
            for (int qy = -r; qy <= r; ++qy) {
                const int64_t y2 = reflect_index(row + qy, H) - buf_row0;
                for (int qx = -r; qx <= r; ++qx) {
                    const int64_t x2 = reflect_index(col + qx, W);
                    const T u = buf[(y2 * W + x2) * C + c];
                    less += (u < v);
                    leq += (u <= v);
                }
            }
            if (less <= m && m < leq) best = v;
        }
    }
    return best;
}

__device__ __forceinline__ void elem_to_pixel(const MergeK& a, int64_t e, int64_t& row, int64_t& col, int& c) {
    const int64_t wc = a.W * a.C;
    row = a.row0 + e / wc;
    const int64_t rem = e % wc;
    col = rem / a.C;
    c = static_cast<int>(rem % a.C);
}

// flat-field epilogue, modules/measurand.py:585-602, operation order kept
__device__ __forceinline__ void flat_field_apply(const MergeK& a, int64_t e, int c, bool with_std,
                                                 double& val, double& sd) {
    const double F = a.flat_u8 ? static_cast<double>(a.flat_u8[e]) / 255.0 : a.flat_f64[e];
    const double m = a.ff_mean[c];
    if (with_std) {
        const double sF = a.flat_std[e];
        const double s = a.ff_std_mean[c];
        const double F2 = F * F;
        double u_acq = (sd * sd) / F2;
        u_acq *= m * m;
        double u_ff = (val * val) / (F2 * F2);
        u_ff *= sF * sF;
        u_ff *= m * m;
        double u_ffm = (val * val) / F2;
        u_ffm *= s * s;
        sd = sqrt(u_acq + u_ff + u_ffm);
    }
    val = (val / F) * m;
}

// ------------------------------------------------------------------------------------------------
// generic kernel: one element per thread, runtime N and C, uint8 or float64 frames.
// LDS: w[256] dw[256] g[256*C] d[256*C]  (plain tables, <= 20 KB)
// ------------------------------------------------------------------------------------------------
template <bool F64IN, bool STD, bool HOT>
__global__ __launch_bounds__(256) void merge_generic(const MergeK a) {
    __shared__ double t_w[256], t_dw[256], t_g[256 * HM_MAX_CHANNELS], t_d[256 * HM_MAX_CHANNELS];
    const int C = a.C;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        t_w[i] = (!F64IN) ? a.w_lut[i] : 0.0;
        t_dw[i] = (!F64IN && STD) ? a.dw_lut[i] : 0.0;
    }
    for (int i = threadIdx.x; i < 256 * C; i += blockDim.x) {
        t_g[i] = a.icrf[i];
        t_d[i] = STD ? a.icrf_diff[i] : 0.0;
    }
    __syncthreads();
    const int N = a.n_frames;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < a.n_elems; q += stride) {
        const int64_t e = a.elem0 + q;          // element relative to row0
        const int64_t ei = a.in_off + e;        // element inside the input buffers
        const int c = static_cast<int>(e % C);
        // ---- pass 1: S = sum_i w_i (exposure_series.py:340) ----
        double S = 0.0;
        for (int i = 0; i < N; ++i) {
            bool hot = false;
            if (HOT && a.dark[i]) hot = a.dark[i][ei] >= a.dark_min[i];
            double w;
            if (F64IN) {
                double v = static_cast<const double*>(a.frame[i])[ei];
                if (HOT && hot) {
                    int64_t row, col; int cc; elem_to_pixel(a, e, row, col, cc);
                    v = median_at(static_cast<const double*>(a.frame[i]), a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                }
                const double dv = v - 0.5;
                w = exp(-30.0 * (dv * dv));
            } else {
                uint8_t dn = static_cast<const uint8_t*>(a.frame[i])[ei];
                if (HOT && hot) {
                    int64_t row, col; int cc; elem_to_pixel(a, e, row, col, cc);
                    dn = median_at(static_cast<const uint8_t*>(a.frame[i]), a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                }
                w = t_w[dn];
            }
            S = (i == 0) ? w : S + w;
        }
        if (a.out_sum_w) a.out_sum_w[e] = S;
        if (!a.out_val) continue;
        // Same operation sequence as merge_u8_fast, so a result does not depend on which kernel (or
        // which tiling) produced it: reciprocals of S and S**2 (exposure_series.py:343) are formed once,
        // the numerator of :388 is accumulated with fma and divided by S at the end.
        const double invS = 1.0 / S;
        const double invS2 = 1.0 / (S * S);
        // ---- pass 2: exposure_series.py:382-389 ----
        double acc = 0.0, var = 0.0;
        for (int i = 0; i < N; ++i) {
            bool hot = false;
            if (HOT && a.dark[i]) hot = a.dark[i][ei] >= a.dark_min[i];
            int64_t row = 0, col = 0; int cc = 0;
            if (HOT && hot) elem_to_pixel(a, e, row, col, cc);
            double w, dw;
            uint32_t idx;
            if (F64IN) {
                double v = static_cast<const double*>(a.frame[i])[ei];
                if (HOT && hot)
                    v = median_at(static_cast<const double*>(a.frame[i]), a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                const double dv = v - 0.5;
                w = exp(-30.0 * (dv * dv));
                dw = (-60.0 * dv) * w;
                idx = static_cast<uint32_t>(static_cast<int64_t>(rint(v * 255.0))) & 255u;   // measurand.py:503
            } else {
                uint8_t dn = static_cast<const uint8_t*>(a.frame[i])[ei];
                if (HOT && hot)
                    dn = median_at(static_cast<const uint8_t*>(a.frame[i]), a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                idx = dn;
                w = t_w[dn];
                dw = STD ? t_dw[dn] : 0.0;
            }
            const double g = t_g[idx * C + c];
            const double it = a.inv_t[i];
            const double wg = w * g;
            acc = (i == 0) ? wg * it : fma(wg, it, acc);               // :388 numerator
            if (STD) {
                double s = a.sd[i][ei];
                if (HOT && hot) s = median_at(a.sd[i], a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                const double dg = t_d[idx * C + c] * s;                // measurand.py:512
                const double A = (dw * g + w * dg) * invS - ((dw * w) * g) * invS2;   // :389
                const double term = (A * dg) * it;
                var = (i == 0) ? term * term : fma(term, term, var);
            }
        }
        double val = acc / S;
        double sd = STD ? sqrt(var) : 0.0;                             // :394
        if (a.has_flat) flat_field_apply(a, e, c, STD, val, sd);
        a.out_val[e] = val;
        if (STD) a.out_std[e] = sd;
    }
}

// ------------------------------------------------------------------------------------------------
// fast kernel: uint8 frames, C == 3, compile-time N, VEC consecutive elements per lane.
// ------------------------------------------------------------------------------------------------
// LDS table layouts of the val-only kernel (entry = 8-byte double unless noted):
//   TAB_PLAIN   w[256] | wg[256*3]                         8 KB   (random DNs: ~3.5-way bank conflicts)
//   TAB_FUSED   {w, wg}[256*3] as 16-byte entries          12 KB  one ds_read_b128 per (element, frame)
//   TAB_REP16   w and wg, 16 replicas each: entry q at q*128 + (lane&15)*8     32 + 96 = 128 KB
//               lanes l and l+16 of a 32-lane LDS group share a replica -> at most 2-way conflicts
//   TAB_REP32W  w private per lane of the group: dn*256 + (lane&31)*8 (64 KB, conflict-free),
//               wg as TAB_REP16 (96 KB)                    160 KB = all of the CU's LDS
//   TAB_FUSED8  {w, wg} 16-byte entries, 8 replicas: q*128 + (lane&7)*16       96 KB
// The std kernel uses {w,dw}[256] and {g,d}[768] as 16-byte entries (16 KB), unreplicated: with the
// float64 std streams it is HBM-bound by a wide margin.
//   TAB_NONE    (tuning probe only, wrong results) no LDS gather at all: w, wg derived from the DN by a
//               conversion - the HBM ceiling of this access pattern
enum { TAB_PLAIN = 0, TAB_FUSED = 1, TAB_REP16 = 2, TAB_REP32W = 3, TAB_FUSED8 = 4, TAB_NONE = 5 };

template <int TAB> struct TabInfo;
template <> struct TabInfo<TAB_PLAIN>  { static constexpr int bytes = 8 * 256 * 4; };
template <> struct TabInfo<TAB_FUSED>  { static constexpr int bytes = 16 * 768; };
template <> struct TabInfo<TAB_REP16>  { static constexpr int bytes = 128 * 256 + 128 * 768; };
template <> struct TabInfo<TAB_REP32W> { static constexpr int bytes = 256 * 256 + 128 * 768; };
template <> struct TabInfo<TAB_FUSED8> { static constexpr int bytes = 128 * 768; };
template <> struct TabInfo<TAB_NONE>   { static constexpr int bytes = 0; };
constexpr int kStdTabBytes = 16 * 256 + 16 * 768;

typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p));
}
__device__ __forceinline__ void store2(double* p, double x, double y) {
    f64x2 v; v.x = x; v.y = y;
    __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(p));
}

// fill the val-only tables; blockDim-agnostic
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
        } else if constexpr (TAB == TAB_FUSED) {
            double2* t = reinterpret_cast<double2*>(lds);
            t[q] = double2{w, wg};
        } else if constexpr (TAB == TAB_REP16) {
            double* tw = reinterpret_cast<double*>(lds);
            double* tg = reinterpret_cast<double*>(lds + 128 * 256);
            for (int r = 0; r < 16; ++r) {
                if (q % C == 0) tw[dn * 16 + r] = w;
                tg[q * 16 + r] = wg;
            }
        } else if constexpr (TAB == TAB_REP32W) {
            double* tw = reinterpret_cast<double*>(lds);
            double* tg = reinterpret_cast<double*>(lds + 256 * 256);
            for (int r = 0; r < 32; ++r)
                if (q % C == 0) tw[dn * 32 + r] = w;
            for (int r = 0; r < 16; ++r) tg[q * 16 + r] = wg;
        } else if constexpr (TAB == TAB_FUSED8) {
            double2* t = reinterpret_cast<double2*>(lds);
            for (int r = 0; r < 8; ++r) t[q * 8 + r] = double2{w, wg};
        }
    }
}

// gather {w, wg} for DN `dn`, channel byte offset prepared by lane_coff<TAB>()
template <int TAB>
__device__ __forceinline__ uint32_t lane_woff() {
    const uint32_t lane = threadIdx.x & 63u;
    if constexpr (TAB == TAB_REP16) return (lane & 15u) * 8u;
    if constexpr (TAB == TAB_REP32W) return (lane & 31u) * 8u;
    return 0u;
}
template <int TAB>
__device__ __forceinline__ uint32_t lane_coff(int c) {
    const uint32_t lane = threadIdx.x & 63u;
    if constexpr (TAB == TAB_PLAIN) return 2048u + c * 8u;
    if constexpr (TAB == TAB_FUSED) return c * 16u;
    if constexpr (TAB == TAB_REP16) return 128u * 256u + c * 128u + (lane & 15u) * 8u;
    if constexpr (TAB == TAB_REP32W) return 256u * 256u + c * 128u + (lane & 15u) * 8u;
    return c * 128u + (lane & 7u) * 16u;   // TAB_FUSED8
}
template <int TAB>
__device__ __forceinline__ void gather_val(const char* lds, uint32_t dn, uint32_t woff, uint32_t coff,
                                           double& w, double& wg) {
    if constexpr (TAB == TAB_NONE) {
        w = static_cast<double>(dn + 1u); wg = static_cast<double>(dn + coff);
    } else if constexpr (TAB == TAB_PLAIN) {
        w = *reinterpret_cast<const double*>(lds + dn * 8u);
        wg = *reinterpret_cast<const double*>(lds + dn * 24u + coff);
    } else if constexpr (TAB == TAB_FUSED) {
        const double2 t = *reinterpret_cast<const double2*>(lds + dn * 48u + coff);
        w = t.x; wg = t.y;
    } else if constexpr (TAB == TAB_REP16) {
        w = *reinterpret_cast<const double*>(lds + dn * 128u + woff);
        wg = *reinterpret_cast<const double*>(lds + dn * 384u + coff);
    } else if constexpr (TAB == TAB_REP32W) {
        w = *reinterpret_cast<const double*>(lds + dn * 256u + woff);
        wg = *reinterpret_cast<const double*>(lds + dn * 384u + coff);
    } else {
        const double2 t = *reinterpret_cast<const double2*>(lds + dn * 384u + coff);
        w = t.x; wg = t.y;
    }
}

// Work decomposition of the fast kernel (tools/membench2.hip and tools/mergelab.hip are the experiments
// behind it): a wave owns "groups" of U * 128 consecutive elements. In a 128-element sub-unit lane l
// handles elements 2l and 2l+1:
//   * input: one global_load_ushort per (frame, sub-unit), 128 contiguous bytes per wave instruction,
//     immediate offsets 128*s off one scalar base; with U = 4 a wave streams 512 contiguous bytes per
//     frame, the span that measured best for this 7-in / 1-out traffic shape;
//   * output: lane l owns the 16 contiguous bytes of elements 2l, 2l+1, so every store instruction is
//     one fully contiguous 1 KB global_store_dwordx4 (nontemporal) - no cross-lane transposition.
//     (Store instructions must cover whole 128-byte lines: 4 elements per lane stored directly is a
//     32-byte-stride pattern that costs 146 us instead of 128 us for the traffic alone; transposing
//     through LDS fixed that but added 13 % to an LDS pipe that bank conflicts already fill.)
// All group-level address arithmetic is scalar (the group index is wave-uniform). All N*U loads of a
// group are issued before the first gather; PREFETCH issues the next group's loads first.
// Channel of element 2l + j of sub-unit s of group g: (g*U*128 + 128*s + 2l + j) % 3 = (2*(g*U + s + l) + j) % 3.
//
// EXTRAS = flat-field epilogue and/or sum-of-weights output compiled in (runtime-selected inside);
// the plain instantiation has a branch-free epilogue.

// keeps an accumulator chain where the source puts it (hipcc otherwise sinks the second element's chain
// below the first element's epilogue and keeps every gathered value live until then)
#define HM_PIN(x) asm volatile("" : "+v"(x))

constexpr uint32_t kSub = 128;     // elements per sub-unit (64 lanes x 2)

template <int NF, int U, int TAB, bool STD, bool HOT, bool PREFETCH, bool EXTRAS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void merge_u8_fast(const MergeK a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int C = 3;
    constexpr uint32_t GROUP = U * kSub;
    if constexpr (!STD) {
        fill_val_tables<TAB>(lds, a);
    } else {
        double2* t_wdw = reinterpret_cast<double2*>(lds);
        double2* t_gd = reinterpret_cast<double2*>(lds + 16 * 256);
        for (int i = threadIdx.x; i < 256; i += BLOCK) t_wdw[i] = double2{a.w_lut[i], a.dw_lut[i]};
        for (int i = threadIdx.x; i < 256 * C; i += BLOCK) t_gd[i] = double2{a.icrf[i], a.icrf_diff[i]};
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lane2 = lane * 2u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = BLOCK / 64;
    const uint32_t n_groups = static_cast<uint32_t>(a.n_elems / GROUP);
    const uint32_t gstride = gridDim.x * WPB;
    const uint32_t woff = lane_woff<TAB>();

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
    if (PREFETCH && g < n_groups) load_group(g, raw);

    for (; g < n_groups; g += gstride) {
        uint32_t cur[NF][U];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int s = 0; s < U; ++s) cur[i][s] = raw[i][s];
            if (g + gstride < n_groups) load_group(g + gstride, raw);
        } else {
            load_group(g, cur);
        }
        const int64_t gbase = static_cast<int64_t>(g) * GROUP;            // relative to row0, scalar

#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int64_t sbase = gbase + kSub * s;                         // scalar
            const int64_t e0 = sbase + lane2;
            const uint32_t c0 = (2u * (g * U + s + lane)) % 3u;

            // hot-pixel prologue (rare): replace the DN by the k x k median of its frame
            uint32_t hotmask[NF];
            if constexpr (HOT) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    hotmask[i] = 0;
                    if (a.dark[i]) {
                        const uint32_t dr = ld_u16(a.dark[i] + a.in_off + sbase + lane2);
                        const uint32_t thr = static_cast<uint32_t>(a.dark_min[i]);
                        hotmask[i] = ((dr & 255u) >= thr ? 1u : 0u) | ((dr >> 8) >= thr ? 2u : 0u);
                        if (hotmask[i]) {
                            for (int j = 0; j < 2; ++j) {
                                if (hotmask[i] & (1u << j)) {
                                    int64_t row, col; int cc; elem_to_pixel(a, e0 + j, row, col, cc);
                                    const uint32_t md = median_at(static_cast<const uint8_t*>(a.frame[i]), a.H, a.W, C,
                                                                  a.buf_row0, row, col, cc, a.median_k);
                                    cur[i][s] = (cur[i][s] & ~(255u << (8 * j))) | (md << (8 * j));
                                }
                            }
                        }
                    }
                }
            }

            uint32_t coffs[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t c = (c0 + k) % 3u;
                coffs[k] = STD ? c * 16u : lane_coff<TAB>(static_cast<int>(c));
            }
            double* ov = a.out_val + sbase;                                   // scalar bases
            double* osw = EXTRAS && a.out_sum_w ? a.out_sum_w + sbase : nullptr;

            if constexpr (!STD) {
                double S[2], acc[2];
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const double it = a.inv_t[i];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t dn = j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8);
                        double w, wg;
                        gather_val<TAB>(lds, dn, woff, coffs[j], w, wg);
                        if (i == 0) { S[j] = w; acc[j] = wg * it; }
                        else {
                            S[j] += w;                               // exposure_series.py:340
                            acc[j] = fma(wg, it, acc[j]);            // :388 numerator
                        }
                    }
                    // bound the gathers in flight (4 frames = 8 ds_reads, 32 VGPRs) per scheduling bundle
                    if ((i & 3) == 3 || i == NF - 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { HM_PIN(S[j]); HM_PIN(acc[j]); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                double val[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) val[j] = acc[j] / S[j];
                if constexpr (EXTRAS) {
                    if (a.has_flat) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            double dummy = 0.0;
                            flat_field_apply(a, e0 + j, static_cast<int>((c0 + j) % 3u), false, val[j], dummy);
                        }
                    }
                    if (osw) store2(osw + lane2, S[0], S[1]);
                }
                store2(ov + lane2, val[0], val[1]);
            } else {
                const double2* t_wdw = reinterpret_cast<const double2*>(lds);
                const char* t_gd = lds + 16 * 256;
                // pass 1: S = sum_i w_i
                double S[2];
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const double w = t_wdw[j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8)].x;
                        if (i == 0) S[j] = w; else S[j] += w;
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
                    const double* sp = a.sd[i] + a.in_off + sbase;                               // scalar base
                    double sdv[2];
                    {
                        const f64x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(sp + lane2));
                        sdv[0] = v0.x; sdv[1] = v0.y;
                    }
                    if constexpr (HOT) {
                        if (hotmask[i]) {
                            for (int j = 0; j < 2; ++j) {
                                if (hotmask[i] & (1u << j)) {
                                    int64_t row, col; int cc; elem_to_pixel(a, e0 + j, row, col, cc);
                                    sdv[j] = median_at(a.sd[i], a.H, a.W, C, a.buf_row0, row, col, cc, a.median_k);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t dn = j == 0 ? (cur[i][s] & 255u) : (cur[i][s] >> 8);
                        const double2 wdw = t_wdw[dn];
                        const double2 gd = *reinterpret_cast<const double2*>(t_gd + dn * 48u + coffs[j]);
                        const double w = wdw.x, dw = wdw.y, gg = gd.x;
                        const double dg = gd.y * sdv[j];                                        // measurand.py:512
                        const double A = (dw * gg + w * dg) * invS[j] - ((dw * w) * gg) * invS2[j];   // :389
                        const double term = (A * dg) * it;
                        if (i == 0) { acc[j] = (w * gg) * it; var[j] = term * term; }
                        else {
                            acc[j] = fma(w * gg, it, acc[j]);                                    // :388
                            var[j] = fma(term, term, var[j]);
                        }
                    }
                    if ((i & 1) == 1 || i == NF - 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) { HM_PIN(acc[j]); HM_PIN(var[j]); }
                        __builtin_amdgcn_sched_barrier(0);   // two frames' std loads + gathers at a time
                    }
                }
                double val[2], so[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    val[j] = acc[j] / S[j];
                    so[j] = sqrt(var[j]);                                                        // :394
                }
                if constexpr (EXTRAS) {
                    if (a.has_flat) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            flat_field_apply(a, e0 + j, static_cast<int>((c0 + j) % 3u), true, val[j], so[j]);
                    }
                    if (osw) store2(osw + lane2, S[0], S[1]);
                }
                store2(ov + lane2, val[0], val[1]);
                store2(a.out_std + sbase + lane2, so[0], so[1]);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one sub-unit's gathers from piling onto the next one's
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
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

static FastCfg default_cfg(bool with_std) {
    if (with_std) return FastCfg{TAB_PLAIN, 2, 0, 256};
    return FastCfg{TAB_FUSED, 2, 0, 256};     // tools/tune_merge.py, tools/mergelab.hip, profiles/
}

static bool decode_variant(int variant, bool with_std, FastCfg& c) {
    c = default_cfg(with_std);
    if (variant <= 0 || with_std) return true;
    const int tab = variant / 1000, pf = (variant / 100) % 10, u = (variant / 10) % 10, bc = variant % 10;
    if (tab < 0 || tab > TAB_NONE || pf > 1 || (u != 2 && u != 4 && u != 8) || bc > 1) return false;
    c.tab = tab; c.u = u; c.prefetch = pf; c.block = bc ? 1024 : 256;
    return true;
}

template <int NF, int U, int TAB, bool STD, bool HOT, bool PF, bool EXTRAS, int BLOCK>
static int launch_one(const MergeK& k, hipStream_t st) {
    constexpr int lds = (STD ? kStdTabBytes : TabInfo<TAB>::bytes) > 0 ? (STD ? kStdTabBytes : TabInfo<TAB>::bytes) : 16;
    static_assert(lds <= kMaxLds, "LDS budget");
    auto kernel = merge_u8_fast<NF, U, TAB, STD, HOT, PF, EXTRAS, BLOCK>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds) != hipSuccess) {
            (void)hipGetLastError();
            return HM_ELAUNCH;
        }
    }
    int per_cu = 2048 / BLOCK;                       // 32 waves per CU
    if (kMaxLds / lds < per_cu) per_cu = kMaxLds / lds;
    if (per_cu < 1) per_cu = 1;
    const int64_t groups = k.n_elems / (U * static_cast<int>(kSub));
    const unsigned grid = stream_grid(groups, BLOCK / 64, per_cu);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, st, k);
    return launch_status();
}

#ifndef HM_TUNE_NF
#define HM_TUNE_NF 0      /* build with -DHM_TUNE_NF=7 to get the full variant matrix for N = 7 */
#endif

template <int NF, int U, int TAB, bool PF>
static int launch_val_blk(const MergeK& k, const FastCfg& c, hipStream_t st) {
    if (c.block == 256) return launch_one<NF, U, TAB, false, false, PF, false, 256>(k, st);
    return launch_one<NF, U, TAB, false, false, PF, false, 1024>(k, st);
}

template <int NF, int U, bool PF>
static int launch_val_tab(const MergeK& k, const FastCfg& c, hipStream_t st) {
    switch (c.tab) {
        case TAB_PLAIN:  return launch_val_blk<NF, U, TAB_PLAIN, PF>(k, c, st);
        case TAB_FUSED:  return launch_val_blk<NF, U, TAB_FUSED, PF>(k, c, st);
        case TAB_REP16:  return launch_val_blk<NF, U, TAB_REP16, PF>(k, c, st);
        case TAB_FUSED8: return launch_val_blk<NF, U, TAB_FUSED8, PF>(k, c, st);
        default:         return launch_val_blk<NF, U, TAB_NONE, PF>(k, c, st);
    }
}

// Units per group (U sub-units of 128 elements) of the production configurations:
constexpr int kUVal = 2;     // val-only: 256 contiguous bytes per frame per wave iteration (tune: 134.7 us vs 139.9 at U = 4)
constexpr int kUStd = 2;     // with std: the float64 std streams dominate; fewer registers

template <int NF>
static int launch_fast_nf(const MergeK& k, const FastCfg& c, bool with_std, bool hot, hipStream_t st) {
    const bool extras = k.has_flat || k.out_sum_w;
    if (with_std) {
        if (hot) return launch_one<NF, kUStd, TAB_PLAIN, true, true, false, true, 256>(k, st);
        if (extras) return launch_one<NF, kUStd, TAB_PLAIN, true, false, false, true, 256>(k, st);
        return launch_one<NF, kUStd, TAB_PLAIN, true, false, false, false, 256>(k, st);
    }
    if (hot) return launch_one<NF, kUVal, TAB_FUSED, false, true, false, true, 256>(k, st);
    if (extras) return launch_one<NF, kUVal, TAB_FUSED, false, false, false, true, 256>(k, st);
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
        return launch_one<NF, kUVal, TAB_FUSED, false, false, false, false, 256>(k, st);
    }
}

// elements per group of the configuration launch_fast_nf() will really use
static int fast_group_elems(int n_frames, const FastCfg& c, bool with_std, bool hot, bool extras) {
    if (with_std) return kUStd * static_cast<int>(kSub);
    if (hot || extras || n_frames != HM_TUNE_NF) return kUVal * static_cast<int>(kSub);
    return c.u * static_cast<int>(kSub);
}

static int launch_generic(const MergeK& k, bool f64in, bool with_std, bool hot, hipStream_t st) {
    const unsigned grid = stream_grid(k.n_elems, 256, 8);
#define HM_GEN(F, S, H) hipLaunchKernelGGL((merge_generic<F, S, H>), dim3(grid), dim3(256), 0, st, k)
    if (f64in) {
        if (with_std) { if (hot) HM_GEN(true, true, true); else HM_GEN(true, true, false); }
        else          { if (hot) HM_GEN(true, false, true); else HM_GEN(true, false, false); }
    } else {
        if (with_std) { if (hot) HM_GEN(false, true, true); else HM_GEN(false, true, false); }
        else          { if (hot) HM_GEN(false, false, true); else HM_GEN(false, false, false); }
    }
#undef HM_GEN
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
    if (g->darks_u8)
        for (int i = 0; i < N; ++i) per += g->darks_u8[i] ? 1 : 0;
    return per * E;
}

extern "C" int hm_merge(const hm_merge_args* g, void* stream) {
    using namespace hm;
    if (!g || g->struct_size != sizeof(hm_merge_args)) return HM_EINVAL;
    const int N = g->n_frames, C = g->channels;
    if (N < 1 || C < 1 || g->height < 1 || g->width < 1 || g->rows < 0) return HM_EINVAL;
    if (N > HM_MAX_FRAMES || C > HM_MAX_CHANNELS) return HM_EUNSUPPORTED;
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
    if (g->rows == 0) return HM_OK;

    MergeK k{};
    for (int i = 0; i < N; ++i) {
        const void* f = f64in ? static_cast<const void*>(g->frames_f64[i]) : static_cast<const void*>(g->frames_u8[i]);
        if (!f) return HM_EINVAL;
        if (f64in && !aligned(f, 8)) return HM_EALIGN;
        k.frame[i] = f;
        if (with_std) {
            if (!g->stds[i]) return HM_EINVAL;
            if (!aligned(g->stds[i], 8)) return HM_EALIGN;
            k.sd[i] = g->stds[i];
        }
        if (!(g->exposures[i] > 0.0)) return HM_EINVAL;
        k.inv_t[i] = 1.0 / g->exposures[i];
        k.dark[i] = hot ? g->darks_u8[i] : nullptr;
        k.dark_min[i] = hot && g->dark_min_dn ? g->dark_min_dn[i] : 256;
    }
    k.icrf = g->icrf; k.icrf_diff = g->icrf_diff; k.w_lut = g->w_lut; k.dw_lut = g->dw_lut;
    k.flat_u8 = g->flat_u8; k.flat_f64 = g->flat_f64; k.flat_std = g->flat_std;
    for (int c = 0; c < HM_MAX_CHANNELS; ++c) { k.ff_mean[c] = g->ff_mean[c]; k.ff_std_mean[c] = g->ff_std_mean[c]; }
    k.out_val = g->out_val; k.out_std = g->out_std; k.out_sum_w = g->out_sum_w;
    if ((k.out_val && !aligned(k.out_val, 8)) || (k.out_std && !aligned(k.out_std, 8)) ||
        (k.out_sum_w && !aligned(k.out_sum_w, 8))) return HM_EALIGN;
    const int64_t E = g->rows * g->width * C;
    k.n_elems = E; k.elem0 = 0;
    k.in_off = (g->row0 - g->buf_row0) * g->width * C;
    k.H = g->height; k.W = g->width; k.row0 = g->row0; k.buf_row0 = g->buf_row0; k.buf_rows = g->buf_rows;
    k.n_frames = N; k.C = C; k.median_k = hot ? g->median_k : 3; k.has_flat = flat ? 1 : 0;
    hipStream_t st = as_stream(stream);

    // ---- fast path eligibility ----
    FastCfg cfg;
    if (!decode_variant(g->variant, with_std, cfg)) return HM_EINVAL;
    bool fast = !f64in && C == 3 && N <= 16 && g->out_val && E < (int64_t{1} << 32) && g->variant >= 0;
    if (fast) {
        for (int i = 0; i < N && fast; ++i) {
            fast = aligned(static_cast<const uint8_t*>(k.frame[i]) + k.in_off, 2);
            if (fast && with_std) fast = aligned(k.sd[i] + k.in_off, 16);
            if (fast && k.dark[i]) fast = aligned(k.dark[i] + k.in_off, 2);
        }
        fast = fast && aligned(k.out_val, 16) && (!k.out_std || aligned(k.out_std, 16)) &&
               (!k.out_sum_w || aligned(k.out_sum_w, 16));
    }
    if (!fast) return launch_generic(k, f64in, with_std, hot, st);

    const int64_t grp = fast_group_elems(N, cfg, with_std, hot, flat || g->out_sum_w);
    const int64_t body = (E / grp) * grp;
    int rc = HM_OK;
    if (body > 0) {
        MergeK kb = k;
        kb.n_elems = body;
        switch (N) {
#define HM_CASE(n) case n: rc = launch_fast_nf<n>(kb, cfg, with_std, hot, st); break;
            HM_CASE(1) HM_CASE(2) HM_CASE(3) HM_CASE(4) HM_CASE(5) HM_CASE(6) HM_CASE(7) HM_CASE(8)
            HM_CASE(9) HM_CASE(10) HM_CASE(11) HM_CASE(12) HM_CASE(13) HM_CASE(14) HM_CASE(15) HM_CASE(16)
#undef HM_CASE
            default: return HM_EUNSUPPORTED;
        }
        if (rc != HM_OK) return rc;
    }
    if (body < E) {                                    // tail: less than one group
        MergeK kt = k;
        kt.elem0 = body; kt.n_elems = E - body;
        rc = launch_generic(kt, false, with_std, hot, st);
    }
    return rc;
}
