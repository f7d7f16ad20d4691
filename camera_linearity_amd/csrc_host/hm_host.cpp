// hm_host.cpp - HOST build of the C ABI of include/hdrmerge.h (libhdrmerge_host.so).
//
// The reference's factory returns a NumPy (host) Measurand for use_cupy=False (modules/measurand_factory.py:10-14,
// modules/measurand.py:684-714), which is also BASELINE.json's configs[0] ("NumPy Measurand CPU merge, plumbing, no GPU").
// This file backs that slot: the same entry points as libhdrmerge.so with HOST pointers - `stream` is ignored, calls are
// synchronous, workspaces may be NULL - written from scratch in plain C++ (g++ -O2 -fopenmp, no HIP, no NumPy, nothing from
// oracle/). It is selected EXPLICITLY (Measurand(use_cupy=False) / HostMeasurand); the HIP backend never falls back to it.
//
// Arithmetic: the operation sequence of the device kernels, element by element (merge_one_element of hm_merge.hip, the operator
// formulas of hm_ops.hip, ...), each citing the reference line it follows. Statistics use the reference's own two-pass form.
// Since round 4 also the upstream producers (hm_welford_*, hm_linearity_energy: SURVEY.md 8f-2/3). Not here:
// the TIFF strip decoders (host code already, in libhdrmerge.so) and the hm_debug_* probes.
#include "hdrmerge.h"

#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();
constexpr double kInf = std::numeric_limits<double>::infinity();

inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// measurand.py:615: w = e ** (-30 (v - 0.5)^2)
inline double gauss_weight(double dv) { return std::exp(-30.0 * (dv * dv)); }

// measurand.py:503: around(v * 255) (half to even) then astype(uint8) - wraps modulo 256, negatives included
inline uint32_t lut_index(double v) {
    const double r = std::nearbyint(v * 255.0);          // default rounding mode: to nearest even
    if (!(r == r) || r >= 9.2e18 || r <= -9.2e18) return 0u;
    return static_cast<uint32_t>(static_cast<uint64_t>(static_cast<int64_t>(r)) & 255u);
}

// scipy.ndimage 'reflect' (d c b a | a b c d)
inline int64_t reflect_index(int64_t i, int64_t n) {
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - i - 1;
    }
    return i;
}

// k x k median of one channel around one pixel; buf holds image rows [buf_row0, ...)
template <typename T>
inline T median_at(const T* buf, int64_t H, int64_t W, int C, int64_t buf_row0, int64_t row, int64_t col, int c, int k) {
    T v[49];
    const int r = k / 2;
    int n = 0;
    for (int dy = -r; dy <= r; ++dy) {
        const int64_t yy = reflect_index(row + dy, H) - buf_row0;
        for (int dx = -r; dx <= r; ++dx) {
            const int64_t xx = reflect_index(col + dx, W);
            v[n++] = buf[(yy * W + xx) * C + c];
        }
    }
    std::nth_element(v, v + n / 2, v + n);
    return v[n / 2];
}

// measurand.py:585-602 (1 / F**2 formed once, as the device kernels do; <= 2 ulp on the std)
inline void flat_field_math(double F, double sF, double m, double s, bool with_std, double& val, double& sd) {
    if (with_std) {
        const double iF2 = 1.0 / (F * F);
        const double v2 = val * val;
        const double u_acq = ((sd * sd) * iF2) * (m * m);
        const double u_ff = ((v2 * (iF2 * iF2)) * (sF * sF)) * (m * m);
        const double u_ffm = (v2 * iF2) * (s * s);
        sd = std::sqrt(u_acq + u_ff + u_ffm);
    }
    val = (val / F) * m;
}

struct Bcast {
    int ndim;
    int64_t shape[HM_MAX_DIMS], st1[HM_MAX_DIMS], st2[HM_MAX_DIMS];
    int64_t n;
};
bool fill_bcast(Bcast& b, int ndim, const int64_t* shape, const int64_t* s1, const int64_t* s2) {
    if (ndim < 1 || ndim > HM_MAX_DIMS || !shape || !s1 || !s2) return false;
    b.ndim = ndim; b.n = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] < 0 || s1[d] < 0 || s2[d] < 0) return false;
        b.shape[d] = shape[d]; b.st1[d] = s1[d]; b.st2[d] = s2[d];
        b.n *= shape[d];
    }
    return true;
}
inline void bcast_offsets(const Bcast& b, int64_t e, int64_t& o1, int64_t& o2) {
    o1 = 0; o2 = 0;
    for (int d = b.ndim - 1; d >= 0; --d) {
        const int64_t i = e % b.shape[d];
        e /= b.shape[d];
        o1 += i * b.st1[d];
        o2 += i * b.st2[d];
    }
}

// compute_dimension_statistics of ONE reduction line given by an accessor (measurand.py:337-347): NaN-ignoring, weights 1/std
template <typename Get>
inline void line_statistics(int64_t n, bool weighted, Get get, double& mean, double& sd, double& err) {
    if (!weighted) {
        double s = 0.0; int64_t cnt = 0;
        for (int64_t k = 0; k < n; ++k) { double v, u; get(k, v, u); if (v == v) { s += v; ++cnt; } }
        mean = cnt ? s / static_cast<double>(cnt) : kNaN;                          // nanmean
        double q = 0.0;
        for (int64_t k = 0; k < n; ++k) { double v, u; get(k, v, u); if (v == v) { const double d = v - mean; q += d * d; } }
        sd = cnt ? std::sqrt(q / static_cast<double>(cnt)) : kNaN;                 // nanstd
        err = kNaN;
        return;
    }
    double sw = 0.0, svw = 0.0, ss = 0.0; int64_t cs = 0;
    for (int64_t k = 0; k < n; ++k) {
        double v, u; get(k, v, u);
        const double w = 1.0 / u;                                                  // :342
        if (w == w) sw += w;                                                       // nansum(weights)
        const double vw = v * w;
        if (vw == vw) svw += vw;                                                   // nansum(values * weights)
        if (u == u) { ss += u; ++cs; }
    }
    mean = svw / sw;                                                               // :344
    double q = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        double v, u; get(k, v, u);
        const double w = 1.0 / u;
        const double d = v - mean;
        const double t = w * (d * d);                                              // :345
        if (t == t) q += t;
    }
    sd = std::sqrt(q / sw);
    err = cs ? ss / static_cast<double>(cs) : kNaN;                                // nanmean(stds)
}

// ExposurePair difference terms of one element (measurand.py:634-653)
inline void diff_terms(double xv, double xs, double yv, double ys, double mult, bool with_std,
                       double& a, double& as, double& r, double& rs) {
    const double scale = mult * yv;
    a = xv - scale;
    r = a / scale;
    as = 0.0; rs = 0.0;
    if (with_std) {
        const double m1 = mult * ys;
        as = std::sqrt(xs * xs + m1 * m1);
        const double u1 = xs / (mult * yv);
        const double u2 = (ys * xv) / (mult * (yv * yv));
        rs = std::sqrt(u1 * u1 + u2 * u2);
    }
}

}  // namespace

extern "C" {

int hm_version(void) { return HM_ABI_VERSION; }

const char* hm_strerror(int code) {
    switch (code) {
        case HM_OK: return "ok";
        case HM_EINVAL: return "invalid argument";
        case HM_EUNSUPPORTED: return "unsupported configuration (host build: channels > HM_MAX_CHANNELS, or a device-only entry point)";
        case HM_EALIGN: return "float64 buffer is not 8-byte aligned";
        case HM_ELAUNCH: return "launch failed";
        case HM_ENODEVICE: return "no usable gfx950 device";
        case HM_ESHAPE: return "inconsistent geometry (tile outside image or median halo missing)";
        default: return "unknown hdrmerge error";
    }
}

int hm_device_info(int* n_devices, int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    if (n_devices) *n_devices = 0;
    if (cu_count) *cu_count = 0;
    if (lds_bytes) *lds_bytes = 0;
    if (arch && arch_len > 0) { std::strncpy(arch, "host", static_cast<size_t>(arch_len) - 1); arch[arch_len - 1] = 0; }
    return HM_OK;
}

// ---- rows 1, 3, 4 -------------------------------------------------------------------------------------------------
int hm_gaussian_weight_f64(const double* v, double* w, double* dw, int64_t n, void*) {
    if (n < 0 || (n > 0 && !v)) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const double dv = v[e] - 0.5;
        const double y = gauss_weight(dv);
        if (w) w[e] = y;
        if (dw) dw[e] = (-60.0 * dv) * y;                                         // measurand.py:616
    }
    return HM_OK;
}

int hm_gaussian_weight_u8(const uint8_t* dn, const double* w_lut, const double* dw_lut, double* w, double* dw, int64_t n, void*) {
    if (n < 0 || (n > 0 && (!dn || (w && !w_lut) || (dw && !dw_lut)))) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        if (w) w[e] = w_lut[dn[e]];
        if (dw) dw[e] = dw_lut[dn[e]];
    }
    return HM_OK;
}

int hm_gaussian_weight_lut_host(double* w_lut, double* dw_lut) {
    if (!w_lut && !dw_lut) return HM_EINVAL;
    for (int k = 0; k < HM_BITS; ++k) {
        const double v = static_cast<double>(k) / 255.0;
        const double y = std::pow(M_E, -30.0 * ((v - 0.5) * (v - 0.5)));
        if (w_lut) w_lut[k] = y;
        if (dw_lut) dw_lut[k] = ((-2.0 * 30.0) * (v - 0.5)) * y;
    }
    return HM_OK;
}

int hm_u8_to_unit_f64(const uint8_t* dn, double* out, int64_t n, void*) {
    if (n < 0 || (n > 0 && (!dn || !out))) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) out[e] = static_cast<double>(dn[e]) / 255.0;   // image_set.py:223
    return HM_OK;
}

static int linearize_any(const uint8_t* dn, const double* v, const double* std_, const double* icrf, const double* icrf_diff,
                         double* out_val, double* out_std, uint8_t* out_idx, int64_t n, int C, int lut_stride) {
    if (n < 0 || C < 1 || (lut_stride != 1 && lut_stride != C)) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if ((!dn && !v) || !icrf || !out_val) return HM_EINVAL;
    const bool use_std = std_ && icrf_diff && out_std;                            // measurand.py:498-500
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const uint32_t k = dn ? dn[e] : lut_index(v[e]);                          // :505 / :503
        const int64_t at = lut_stride == 1 ? k : static_cast<int64_t>(k) * C + (e % C);
        out_val[e] = icrf[at];
        if (use_std) out_std[e] = icrf_diff[at] * std_[e];                        // :512
        if (out_idx) out_idx[e] = static_cast<uint8_t>(k);
    }
    return HM_OK;
}
int hm_linearize_u8(const uint8_t* dn, const double* std_, const double* icrf, const double* icrf_diff, double* out_val, double* out_std,
                    int64_t n, int C, int lut_stride, void*) {
    return linearize_any(dn, nullptr, std_, icrf, icrf_diff, out_val, out_std, nullptr, n, C, lut_stride);
}
int hm_linearize_f64(const double* v, const double* std_, const double* icrf, const double* icrf_diff, double* out_val, double* out_std,
                     uint8_t* out_idx, int64_t n, int C, int lut_stride, void*) {
    return linearize_any(nullptr, v, std_, icrf, icrf_diff, out_val, out_std, out_idx, n, C, lut_stride);
}

// ---- rows 5-9: the merge ------------------------------------------------------------------------------------------
size_t hm_merge_hot_workspace_bytes(int64_t) { return 16; }
size_t hm_merge_hot_workspace_min_bytes(int64_t) { return 16; }
size_t hm_merge_frames_workspace_bytes(int, int64_t, int) { return 0; }           // any number of frames in one pass on the host

int64_t hm_merge_algorithmic_bytes(const hm_merge_args* g) {
    if (!g || g->n_frames <= 0 || g->rows <= 0 || g->width <= 0 || g->channels <= 0) return 0;
    const int64_t E = g->rows * g->width * g->channels;
    const bool s = g->stds != nullptr;
    int64_t per = g->n_frames * ((g->frames_f64 ? 8 : 1) + (s ? 8 : 0));
    if (g->out_val) per += 8 * (1 + (s ? 1 : 0));
    if (g->out_sum_w) per += 8;
    if (g->flat_u8 || g->flat_f64) per += (g->flat_u8 ? 1 : 8) + ((s && g->flat_std) ? 8 : 0);
    if (g->darks_u8)
        for (int i = 0; i < g->n_frames; ++i) {
            if (!g->darks_u8[i]) continue;
            bool seen = false;
            for (int k = 0; k < i; ++k)
                seen = seen || (g->darks_u8[k] == g->darks_u8[i] && (!g->dark_min_dn || g->dark_min_dn[k] == g->dark_min_dn[i]));
            per += seen ? 0 : 1;
        }
    return per * E;
}

static int merge_check(const hm_merge_args* g_in, hm_merge_args& full, bool& hot) {
    if (!g_in) return HM_EINVAL;
    const uint32_t sz = g_in->struct_size;
    if (sz != sizeof(hm_merge_args) && sz != 264u && sz != 280u) return HM_EINVAL;
    full = hm_merge_args{};
    std::memcpy(&full, g_in, sz < sizeof(hm_merge_args) ? sz : sizeof(hm_merge_args));
    const hm_merge_args* g = &full;
    const int N = g->n_frames, C = g->channels;
    if (N < 1 || C < 1 || g->height < 1 || g->width < 1 || g->rows < 0) return HM_EINVAL;
    if (C > HM_MAX_CHANNELS) return HM_EUNSUPPORTED;
    const bool f64in = g->frames_f64 != nullptr;
    if (f64in == (g->frames_u8 != nullptr)) return HM_EINVAL;
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
    if (g->row0 < 0 || g->row0 + g->rows > g->height) return HM_ESHAPE;
    if (g->buf_row0 < 0 || g->buf_row0 > g->row0 || g->buf_row0 + g->buf_rows > g->height ||
        g->buf_row0 + g->buf_rows < g->row0 + g->rows) return HM_ESHAPE;
    hot = false;
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
        if (g->buf_row0 > need_lo || g->buf_row0 + g->buf_rows < need_hi) return HM_ESHAPE;
    }
    for (int i = 0; i < N; ++i) {
        const void* f = f64in ? static_cast<const void*>(g->frames_f64[i]) : static_cast<const void*>(g->frames_u8[i]);
        if (!f) return HM_EINVAL;
        if (f64in && !aligned8(f)) return HM_EALIGN;
        if (with_std) {
            if (!g->stds[i]) return HM_EINVAL;
            if (!aligned8(g->stds[i])) return HM_EALIGN;
        }
        if (!(g->exposures[i] > 0.0)) return HM_EINVAL;
    }
    if ((g->out_val && !aligned8(g->out_val)) || (g->out_std && !aligned8(g->out_std)) || (g->out_sum_w && !aligned8(g->out_sum_w)))
        return HM_EALIGN;
    return HM_OK;
}

int hm_merge_describe(const hm_merge_args* g, char* buf, int buf_len) {
    if (!buf || buf_len < 1) return HM_EINVAL;
    hm_merge_args full; bool hot = false;
    const int rc = merge_check(g, full, hot);
    if (rc == HM_OK && full.rows > 0)
        std::snprintf(buf, static_cast<size_t>(buf_len), "merge_host<f64in=%d,std=%d,hot=%d>(N=%d)", full.frames_f64 != nullptr, full.stds != nullptr, hot, full.n_frames);
    else buf[0] = 0;
    return rc;
}

// exposure_series.py:340-394 per output element, the operation sequence of merge_one_element() (hm_merge.hip): S in frame order;
// 1/S and 1/S**2 once; numerator and variance accumulated with fma in frame order; acc / S last; flat field last.
}  // extern "C" (templates below)

// One instantiation per (float64 frames, dark maps, std) combination: the per-element loop carries no run-time mode tests, and the row / column
// of an element are only worked out where a median needs them. Same operations in the same order in every instantiation.
template <bool F64IN, bool HOT, bool STD>
static void merge_elements(const hm_merge_args* g, const double* inv_t, int k) {
    const int N = g->n_frames, C = g->channels;
    const bool flat = g->flat_u8 || g->flat_f64;
    const int64_t W = g->width, wc = W * C, E = g->rows * wc;
    const int64_t in_off = (g->row0 - g->buf_row0) * wc;
    std::vector<double> wg_store;
    if (!F64IN && !STD && g->out_val) {
        wg_store.resize(static_cast<size_t>(256) * C);
        for (int q = 0; q < 256 * C; ++q) wg_store[static_cast<size_t>(q)] = g->w_lut[q / C] * g->icrf[q];
    }
    const double* const wg_tab = wg_store.data();
#pragma omp parallel
    {
        std::vector<double> vv(static_cast<size_t>(N)), ss(static_cast<size_t>(N));
        std::vector<uint32_t> dd(static_cast<size_t>(N));
        double* const v = vv.data(); double* const sdv = ss.data(); uint32_t* const dn = dd.data();
        // this thread's contiguous range of elements; the channel index runs along with e (a 64-bit remainder per element cost as much as
        // the rest of a val-only element)
        int64_t lo = 0, hi = E;
#ifdef _OPENMP
        {
            const int64_t nt = omp_get_num_threads(), tid = omp_get_thread_num();
            lo = E * tid / nt; hi = E * (tid + 1) / nt;
        }
#endif
        if constexpr (!F64IN && !STD && !HOT) {
            // val-only on DNs without dark maps (BASELINE config 2's arithmetic): four elements at a time - four independent sum / fma chains
            // per frame instead of one (the chains are latency-bound: 7 dependent adds and 7 dependent fmas per element) - from the w g
            // table; the same operations per element in the same order as the one-element loop below, which takes the tail.
            if (wg_tab) {
                constexpr int B = 4;
                int cb[B];
                for (; lo + B <= hi; lo += B) {
                    const int64_t ei = in_off + lo;
                    for (int j = 0; j < B; ++j) cb[j] = static_cast<int>((lo + j) % C);
                    double S4[B], a4[B];
                    for (int i = 0; i < N; ++i) {
                        const uint8_t* p = g->frames_u8[i] + ei;
                        const double it = inv_t[i];
                        for (int j = 0; j < B; ++j) {
                            const uint32_t d = p[j];
                            const double w = g->w_lut[d], wg = wg_tab[d * C + cb[j]];
                            S4[j] = i == 0 ? w : S4[j] + w;                          // exposure_series.py:340
                            a4[j] = i == 0 ? wg * it : std::fma(wg, it, a4[j]);      // :388 numerator
                        }
                    }
                    for (int j = 0; j < B; ++j) {
                        const int64_t e = lo + j;
                        if (g->out_sum_w) g->out_sum_w[e] = S4[j];
                        double val = a4[j] / S4[j], none = 0.0;
                        if (flat) {
                            const double F = g->flat_u8 ? static_cast<double>(g->flat_u8[e]) / 255.0 : g->flat_f64[e];
                            flat_field_math(F, 0.0, g->ff_mean[cb[j]], g->ff_std_mean[cb[j]], false, val, none);
                        }
                        g->out_val[e] = val;
                    }
                }
            }
        }
        int c = static_cast<int>(lo % C) - 1;
        for (int64_t e = lo; e < hi; ++e) {
            const int64_t ei = in_off + e;
            c = c + 1 == C ? 0 : c + 1;
            // the (filtered) frame values of this element
            if constexpr (HOT) {
                const int64_t row = g->row0 + e / wc, col = (e % wc) / C;
                for (int i = 0; i < N; ++i) {
                    const bool h = g->darks_u8[i] && static_cast<int>(g->darks_u8[i][ei]) >= g->dark_min_dn[i];   // measurand.py:545
                    if constexpr (F64IN) v[i] = h ? median_at(g->frames_f64[i], g->height, W, C, g->buf_row0, row, col, c, k) : g->frames_f64[i][ei];
                    else dn[i] = h ? median_at(g->frames_u8[i], g->height, W, C, g->buf_row0, row, col, c, k) : g->frames_u8[i][ei];
                    if constexpr (STD) sdv[i] = h ? median_at(g->stds[i], g->height, W, C, g->buf_row0, row, col, c, k) : g->stds[i][ei];
                }
            } else {
                for (int i = 0; i < N; ++i) {
                    if constexpr (F64IN) v[i] = g->frames_f64[i][ei]; else dn[i] = g->frames_u8[i][ei];
                    if constexpr (STD) sdv[i] = g->stds[i][ei];
                }
            }
            double S = 0.0;
            if (!F64IN && !STD && wg_tab) {
                // val-only on DNs: S and the numerator in ONE walk over the frames, the product w g from a table built once per call
                // (wg_tab[dn C + c] = w_lut[dn] * icrf[dn C + c]: the same product of the same operands, the same bits)
                double acc1 = 0.0;
                const double* tab = wg_tab + c;
                for (int i = 0; i < N; ++i) {
                    const uint32_t d = dn[i];
                    const double w = g->w_lut[d];
                    S = i == 0 ? w : S + w;                                          // exposure_series.py:340
                    acc1 = i == 0 ? tab[d * C] * inv_t[i] : std::fma(tab[d * C], inv_t[i], acc1);   // :388 numerator
                }
                if (g->out_sum_w) g->out_sum_w[e] = S;
                if (!g->out_val) continue;
                double val1 = acc1 / S;
                if (flat) {
                    double none = 0.0;
                    const double F = g->flat_u8 ? static_cast<double>(g->flat_u8[e]) / 255.0 : g->flat_f64[e];
                    flat_field_math(F, 0.0, g->ff_mean[c], g->ff_std_mean[c], false, val1, none);
                }
                g->out_val[e] = val1;
                continue;
            }
            for (int i = 0; i < N; ++i) {
                double w;
                if constexpr (F64IN) w = gauss_weight(v[i] - 0.5); else w = g->w_lut[dn[i]];
                S = i == 0 ? w : S + w;                                              // exposure_series.py:340
            }
            if (g->out_sum_w) g->out_sum_w[e] = S;
            if (!g->out_val) continue;
            const double invS = 1.0 / S, invS2 = 1.0 / (S * S);                     // :343
            double acc = 0.0, var = 0.0;
            for (int i = 0; i < N; ++i) {
                double w, dw = 0.0; uint32_t idx;
                if constexpr (F64IN) {
                    const double dv = v[i] - 0.5;
                    w = gauss_weight(dv);
                    dw = (-60.0 * dv) * w;
                    idx = lut_index(v[i]);
                } else {
                    idx = dn[i];
                    w = g->w_lut[idx];
                    if constexpr (STD) dw = g->dw_lut[idx];
                }
                const double gg = g->icrf[idx * C + c];
                const double it = inv_t[i];
                const double wg = w * gg;
                acc = i == 0 ? wg * it : std::fma(wg, it, acc);                     // :388 numerator
                if constexpr (STD) {
                    const double dg = g->icrf_diff[idx * C + c] * sdv[i];                                // measurand.py:512
                    const double A = (dw * gg + w * dg) * invS - ((dw * w) * gg) * invS2;                // :389
                    const double term = (A * dg) * it;
                    var = i == 0 ? term * term : std::fma(term, term, var);
                }
            }
            double val = acc / S;
            double sd = STD ? std::sqrt(var) : 0.0;                                 // :394
            if (flat) {
                const double F = g->flat_u8 ? static_cast<double>(g->flat_u8[e]) / 255.0 : g->flat_f64[e];
                flat_field_math(F, STD ? g->flat_std[e] : 0.0, g->ff_mean[c], g->ff_std_mean[c], STD, val, sd);
            }
            g->out_val[e] = val;
            if constexpr (STD) g->out_std[e] = sd;
        }
    }
}

extern "C" {

int hm_merge(const hm_merge_args* g_in, void*) {
    hm_merge_args full; bool hot = false;
    const int rc = merge_check(g_in, full, hot);
    if (rc != HM_OK) return rc;
    const hm_merge_args* g = &full;
    if (g->rows == 0) return HM_OK;
    const int N = g->n_frames;
    const bool f64in = g->frames_f64 != nullptr, with_std = g->stds != nullptr && g->out_val != nullptr;
    std::vector<double> inv_t(static_cast<size_t>(N));
    for (int i = 0; i < N; ++i) inv_t[static_cast<size_t>(i)] = 1.0 / g->exposures[i];
    const int k = hot ? g->median_k : 3;
    const int mode = (f64in ? 4 : 0) | (hot ? 2 : 0) | (with_std ? 1 : 0);
    switch (mode) {
        case 0: merge_elements<false, false, false>(g, inv_t.data(), k); break;
        case 1: merge_elements<false, false, true>(g, inv_t.data(), k); break;
        case 2: merge_elements<false, true, false>(g, inv_t.data(), k); break;
        case 3: merge_elements<false, true, true>(g, inv_t.data(), k); break;
        case 4: merge_elements<true, false, false>(g, inv_t.data(), k); break;
        case 5: merge_elements<true, false, true>(g, inv_t.data(), k); break;
        case 6: merge_elements<true, true, false>(g, inv_t.data(), k); break;
        default: merge_elements<true, true, true>(g, inv_t.data(), k); break;
    }
    return HM_OK;
}

}  // extern "C"

// ---- row 8 standalone ---------------------------------------------------------------------------------------------
template <typename T>
static int hot_filter(const T* x, const uint8_t* map_u8, const double* map_f64, int min_dn, double thr, int k, T* out,
                      int64_t H, int64_t W, int C) {
    if (H < 0 || W < 0 || C < 1) return HM_EINVAL;
    if (H * W * C == 0) return HM_OK;
    if (!x || !out || (!map_u8 == !map_f64) || k < 3 || k > 7 || (k % 2) == 0) return HM_EINVAL;
    const int64_t wc = W * C, n = H * wc;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const bool hot = map_u8 ? static_cast<int>(map_u8[e]) >= min_dn : map_f64[e] > thr;              // measurand.py:545
        out[e] = hot ? median_at(x, H, W, C, 0, e / wc, (e % wc) / C, static_cast<int>(e % C), k) : x[e];
    }
    return HM_OK;
}
extern "C" {
int hm_hot_pixel_filter_u8(const uint8_t* x, const uint8_t* map_u8, const double* map_f64, int min_dn, double thr, int median_k,
                           uint8_t* out, int64_t H, int64_t W, int C, void*) {
    return hot_filter(x, map_u8, map_f64, min_dn, thr, median_k, out, H, W, C);
}
int hm_hot_pixel_filter_f64(const double* x, const uint8_t* map_u8, const double* map_f64, int min_dn, double thr, int median_k,
                            double* out, int64_t H, int64_t W, int C, void*) {
    return hot_filter(x, map_u8, map_f64, min_dn, thr, median_k, out, H, W, C);
}

// ---- row 9 standalone ---------------------------------------------------------------------------------------------
size_t hm_roi_mean_workspace_bytes(void) { return 64; }
}  // extern "C"
template <typename T>
static int roi_mean(const T* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0, int64_t y1, double* out, double scale) {
    if (!img || !out || C < 1 || C > HM_MAX_CHANNELS) return HM_EINVAL;
    if (x0 < 0 || y0 < 0 || x1 > H || y1 > W || x1 <= x0 || y1 <= y0) return HM_ESHAPE;
    double acc[HM_MAX_CHANNELS] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t r = x0; r < x1; ++r)
        for (int64_t q = y0; q < y1; ++q)
            for (int c = 0; c < C; ++c) acc[c] += static_cast<double>(img[(r * W + q) * C + c]);
    const double cnt = static_cast<double>((x1 - x0) * (y1 - y0));
    for (int c = 0; c < C; ++c) out[c] = (acc[c] / cnt) / scale;                  // measurand.py:579 (uint8 images: DN / 255)
    return HM_OK;
}
extern "C" {
int hm_roi_mean_u8(const uint8_t* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0, int64_t y1, double* out, void*, void*) {
    return roi_mean(img, H, W, C, x0, x1, y0, y1, out, 255.0);
}
int hm_roi_mean_f64(const double* img, int64_t H, int64_t W, int C, int64_t x0, int64_t x1, int64_t y0, int64_t y1, double* out, void*, void*) {
    return roi_mean(img, H, W, C, x0, x1, y0, y1, out, 1.0);
}
int hm_normalize_by_map(const double* val, const double* std_, const uint8_t* flat_u8, const double* flat_f64, const double* flat_std,
                        const double* ff_mean, const double* ff_std_mean, double* out_val, double* out_std, int64_t n, int C, void*) {
    if (n < 0 || C < 1 || C > HM_MAX_CHANNELS) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!val || !out_val || !ff_mean || (!flat_u8 == !flat_f64)) return HM_EINVAL;
    if (out_std && (!std_ || !flat_std || !ff_std_mean)) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const int c = static_cast<int>(e % C);
        const double F = flat_u8 ? static_cast<double>(flat_u8[e]) / 255.0 : flat_f64[e];
        double v = val[e], s = out_std ? std_[e] : 0.0;
        flat_field_math(F, out_std ? flat_std[e] : 0.0, ff_mean[c], out_std ? ff_std_mean[c] : 0.0, out_std != nullptr, v, s);
        out_val[e] = v;
        if (out_std) out_std[e] = s;
    }
    return HM_OK;
}

// ---- row 10: operators (measurand.py:106-279) ----------------------------------------------------------------------
int hm_binary_op(int op, const double* x1, const double* s1, const double* x2, const double* s2, double* out_val, double* out_std,
                 int ndim, const int64_t* shape, const int64_t* strides1, const int64_t* strides2, void*) {
    Bcast b;
    if (op < HM_OP_ADD || op > HM_OP_POW || !fill_bcast(b, ndim, shape, strides1, strides2)) return HM_EINVAL;
    if (!x1 || !x2 || !out_val) return HM_EINVAL;
    if ((out_std != nullptr) != (s1 != nullptr || s2 != nullptr)) return HM_EINVAL;
    const bool with_std = out_std != nullptr;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < b.n; ++e) {
        int64_t o1, o2;
        bcast_offsets(b, e, o1, o2);
        const double a = x1[o1], c = x2[o2];
        const double sa = (with_std && s1) ? s1[o1] : 0.0, sc = (with_std && s2) ? s2[o2] : 0.0;       // missing std -> zeros (:121-124)
        double r, rs = 0.0;
        switch (op) {
            case HM_OP_ADD: r = a + c; if (with_std) rs = std::sqrt((sa * sa) + (sc * sc)); break;       // :114,126
            case HM_OP_SUB: r = a - c; if (with_std) rs = std::sqrt((sa * sa) + (sc * sc)); break;       // :138,149
            case HM_OP_MUL: r = a * c; if (with_std) { const double p = a * sc, q = c * sa; rs = std::sqrt(p * p + q * q); } break;   // :198,209
            case HM_OP_DIV: r = a / c; if (with_std) { const double u1 = sa / c, u2 = (a * sc) / (c * c); rs = std::sqrt(u1 * u1 + u2 * u2); } break;   // :173,184-186
            default:
                r = std::pow(a, c);                                                                      // :225,236-239
                if (with_std) { const double u1 = c * std::pow(a, c - 1.0), u2 = std::log(a) * r; const double p = u1 * sa, q = u2 * sc; rs = std::sqrt(p * p + q * q); }
        }
        out_val[e] = r;
        if (with_std) out_std[e] = rs;
    }
    return HM_OK;
}

int hm_unary_op(int op, const double* x, const double* s, double* out_val, double* out_std, int64_t n, void*) {
    if (op < HM_UOP_NEG || op > HM_UOP_LOG_10 || n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !out_val || ((out_std != nullptr) != (s != nullptr))) return HM_EINVAL;
    const double ln10 = std::log(10.0);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const double v = x[e];
        double r, rs = 0.0;
        if (op == HM_UOP_NEG) { r = -v; if (out_std) rs = s[e]; }                                        // :154-157
        else if (op == HM_UOP_LOG_E) { r = std::log(v); if (out_std) rs = s[e] / r; }                    // :251,258 (as written)
        else { r = std::log10(v); if (out_std) rs = s[e] / (v * ln10); }                                 // :270,277
        out_val[e] = r;
        if (out_std) out_std[e] = rs;
    }
    return HM_OK;
}

int hm_pow_scalar(const double* x, const double* s, double exponent, double* out_val, double* out_std, int64_t n, void*) {
    if (n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !out_val || ((out_std != nullptr) != (s != nullptr))) return HM_EINVAL;
    const double p = exponent;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const double v = x[e];
        double r, d;                                                                                     // d = x ** (p - 1)
        if (p == 2.0) { r = v * v; d = v; }
        else if (p == 0.5) { r = std::sqrt(v); d = 1.0 / r; }
        else if (p == 1.0) { r = v; d = 1.0; }
        else { r = std::pow(v, p); d = std::pow(v, p - 1.0); }
        out_val[e] = r;
        if (out_std) {
            const double a = (p * d) * s[e];                                                             // :236
            double bterm = 0.0;                                                                          // (log(x1) * x1**x2) * 0, :237-238
            if (!(v > 0.0) || !(std::fabs(v) < kInf) || !(std::fabs(r) < kInf)) bterm = (std::log(v) * r) * 0.0;
            out_std[e] = std::sqrt(a * a + bterm * bterm);
        }
    }
    return HM_OK;
}

int hm_take_axis(const double* x, const double* s, double* out_val, double* out_std, int64_t outer, int64_t axis_len, int64_t inner,
                 const int64_t* indices, int n_indices, void*) {
    if (!x || !out_val || !indices || outer < 0 || axis_len < 1 || inner < 1 || n_indices < 1) return HM_EINVAL;
    if ((out_std != nullptr) != (s != nullptr)) return HM_EINVAL;
    std::vector<int64_t> idx(static_cast<size_t>(n_indices));
    for (int q = 0; q < n_indices; ++q) {
        int64_t v = indices[q];
        if (v < 0) v += axis_len;
        if (v < 0 || v >= axis_len) return HM_ESHAPE;
        idx[static_cast<size_t>(q)] = v;
    }
    for (int64_t o = 0; o < outer; ++o)
        for (int q = 0; q < n_indices; ++q) {
            const int64_t src = (o * axis_len + idx[static_cast<size_t>(q)]) * inner, dst = (o * n_indices + q) * inner;
            std::memcpy(out_val + dst, x + src, static_cast<size_t>(inner) * 8u);
            if (out_std) std::memcpy(out_std + dst, s + src, static_cast<size_t>(inner) * 8u);
        }
    return HM_OK;
}

// ---- linearity statistics ----------------------------------------------------------------------------
int hm_apply_thresholds(double* val, double* std_, const double* lower, const double* upper, int64_t n, int C, void*) {
    if (n < 0 || C < 1 || C > HM_THRESHOLD_MAX_CHANNELS) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!val || !lower || !upper) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const int c = static_cast<int>(e % C);
        const double v = val[e];
        if (v < lower[c] || v > upper[c]) { val[e] = kNaN; if (std_) std_[e] = kNaN; }                   // measurand.py:421-426
    }
    return HM_OK;
}

int hm_compute_difference_bcast(const double* x, const double* sx, const double* y, const double* sy, double multiplier, double* out_abs,
                                double* out_abs_std, double* out_rel, double* out_rel_std, int ndim, const int64_t* shape,
                                const int64_t* strides_x, const int64_t* strides_y, void*) {
    Bcast b;
    if (!fill_bcast(b, ndim, shape, strides_x, strides_y)) return HM_EINVAL;
    if (b.n == 0) return HM_OK;
    if (!x || !y || !out_abs || !out_rel) return HM_EINVAL;
    const bool with_std = sx || sy;
    if (with_std != (out_abs_std != nullptr) || with_std != (out_rel_std != nullptr)) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < b.n; ++e) {
        int64_t ox, oy;
        bcast_offsets(b, e, ox, oy);
        double a, as, r, rs;
        diff_terms(x[ox], sx ? sx[ox] : 0.0, y[oy], sy ? sy[oy] : 0.0, multiplier, with_std, a, as, r, rs);
        out_abs[e] = a; out_rel[e] = r;
        if (with_std) { out_abs_std[e] = as; out_rel_std[e] = rs; }
    }
    return HM_OK;
}
int hm_compute_difference(const double* x, const double* sx, const double* y, const double* sy, double multiplier, double* out_abs,
                          double* out_abs_std, double* out_rel, double* out_rel_std, int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    const int64_t shape[1] = {n}, st[1] = {1};
    return hm_compute_difference_bcast(x, sx, y, sy, multiplier, out_abs, out_abs_std, out_rel, out_rel_std, 1, shape, st, st, stream);
}

int hm_interpolate_bcast(const double* x0, const double* s0, const double* x1, const double* s1, double y0, double y1, double y,
                         double* out, double* out_std, int ndim, const int64_t* shape, const int64_t* strides0, const int64_t* strides1, void*) {
    Bcast b;
    if (!fill_bcast(b, ndim, shape, strides0, strides1)) return HM_EINVAL;
    if (b.n == 0) return HM_OK;
    if (!x0 || !x1 || !out || ((s0 || s1) != (out_std != nullptr))) return HM_EINVAL;
    const double a = y1 - y, bb = y - y0, d = y1 - y0;
    const double ca = (a / d) * (a / d), cb = (bb / d) * (bb / d);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < b.n; ++e) {
        int64_t o0, o1;
        bcast_offsets(b, e, o0, o1);
        out[e] = (x0[o0] * a + x1[o1] * bb) / d;                                                         // :665
        if (out_std) out_std[e] = std::sqrt((s0 ? s0[o0] : 0.0) * ca + (s1 ? s1[o1] : 0.0) * cb);       // :679 as written
    }
    return HM_OK;
}
int hm_interpolate(const double* x0, const double* s0, const double* x1, const double* s1, double y0, double y1, double y, double* out,
                   double* out_std, int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    const int64_t shape[1] = {n}, st[1] = {1};
    return hm_interpolate_bcast(x0, s0, x1, s1, y0, y1, y, out, out_std, 1, shape, st, st, stream);
}

size_t hm_axis_statistics_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
int hm_axis_statistics(const double* val, const double* std_, int64_t outer, int64_t axis_len, int64_t inner, double* out_mean,
                       double* out_std, double* out_err, void*, void*) {
    if (outer < 1 || axis_len < 1 || inner < 1 || !val || !out_mean || !out_std) return HM_EINVAL;
    const int64_t n_out = outer * inner;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n_out; ++j) {
        const int64_t o = j / inner, i = j % inner;
        const double* pv = val + (o * axis_len) * inner + i;
        const double* ps = std_ ? std_ + (o * axis_len) * inner + i : nullptr;
        double mean, sd, err;
        line_statistics(axis_len, std_ != nullptr, [&](int64_t k, double& v, double& u) { v = pv[k * inner]; u = ps ? ps[k * inner] : 1.0; }, mean, sd, err);
        out_mean[j] = mean; out_std[j] = sd;
        if (out_err) out_err[j] = err;
    }
    return HM_OK;
}

size_t hm_axis_statistics2_workspace_bytes(int64_t, int64_t, int64_t, int64_t, int64_t) { return 0; }
int hm_axis_statistics2(const double* val, const double* std_, int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner,
                        double* out_mean, double* out_std, double* out_err, void*, void*) {
    if (outer < 1 || a1 < 1 || mid < 1 || a2 < 1 || inner < 1 || !val || !out_mean || !out_std) return HM_EINVAL;
    const int64_t n_out = outer * mid * inner, line = a1 * a2;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n_out; ++j) {
        const int64_t i = j % inner, m = (j / inner) % mid, o = j / (inner * mid);
        // element (o, k1, m, k2, i) of the dense (outer, a1, mid, a2, inner) block, line position k = k1 * a2 + k2 (NumPy's order of the reduced axes)
        auto at = [&](int64_t k) { return (((o * a1 + k / a2) * mid + m) * a2 + k % a2) * inner + i; };
        double mean, sd, err;
        line_statistics(line, std_ != nullptr, [&](int64_t k, double& v, double& u) { const int64_t e = at(k); v = val[e]; u = std_ ? std_[e] : 1.0; }, mean, sd, err);
        out_mean[j] = mean; out_std[j] = sd;
        if (out_err) out_err[j] = err;
    }
    return HM_OK;
}

size_t hm_channel_statistics_workspace_bytes(void) { return 64; }
int hm_channel_statistics(const double* val, const double* std_, int64_t n, int C, double* out, void*, void*) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !val || !out || n % C != 0) return HM_EINVAL;
    return hm_axis_statistics(val, std_, 1, n / C, C, out, out + C, out + 2 * C, nullptr, nullptr);
}

size_t hm_pair_statistics_workspace_bytes(void) { return 64; }
int hm_pair_statistics(const double* x, const double* sx, const double* y, const double* sy, double multiplier, int64_t n, int C,
                       double* out, void*, void*) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !x || !y || !out || n % C != 0) return HM_EINVAL;
    const bool with_std = sx || sy;
    const int64_t A = n / C;
    for (int h = 0; h < 2; ++h)
        for (int c = 0; c < C; ++c) {
            double mean, sd, err;
            line_statistics(A, with_std, [&](int64_t k, double& v, double& u) {
                const int64_t e = k * C + c;
                double a, as, r, rs;
                diff_terms(x[e], sx ? sx[e] : 0.0, y[e], sy ? sy[e] : 0.0, multiplier, with_std, a, as, r, rs);
                v = h == 0 ? a : r; u = h == 0 ? as : rs;
            }, mean, sd, err);
            double* o = out + 3 * C * h;
            o[c] = mean; o[C + c] = sd; o[2 * C + c] = err;
        }
    return HM_OK;
}

size_t hm_pairs_statistics_workspace_bytes(int) { return 64; }
int hm_pairs_statistics(const double* const* vals, const double* const* stds, int n_frames, const int32_t* pair_i, const int32_t* pair_j,
                        const double* multipliers, int n_pairs, int64_t n, int C, const double* lower, const double* upper, double* out,
                        void*, void*) {
    if (n_frames < 1 || n_pairs < 0 || n < 1 || C < 1 || C > HM_MAX_CHANNELS || !vals || !out || (n_pairs && (!pair_i || !pair_j || !multipliers)))
        return HM_EINVAL;
    if ((lower != nullptr) != (upper != nullptr)) return HM_EINVAL;
    for (int i = 0; i < n_frames; ++i) if (!vals[i] || (stds && !stds[i])) return HM_EINVAL;
    for (int p = 0; p < n_pairs; ++p)
        if (pair_i[p] < 0 || pair_i[p] >= n_frames || pair_j[p] < 0 || pair_j[p] >= n_frames) return HM_EINVAL;
    if (lower)                                                                                           // exposure_series.py:437-441, in place
        for (int i = 0; i < n_frames; ++i)
            hm_apply_thresholds(const_cast<double*>(vals[i]), stds ? const_cast<double*>(stds[i]) : nullptr, lower, upper, n, C, nullptr);
#pragma omp parallel for schedule(dynamic)
    for (int p = 0; p < n_pairs; ++p)
        hm_pair_statistics(vals[pair_i[p]], stds ? stds[pair_i[p]] : nullptr, vals[pair_j[p]], stds ? stds[pair_j[p]] : nullptr, multipliers[p],
                           n, C, out + static_cast<int64_t>(p) * 6 * C, nullptr, nullptr);
    return HM_OK;
}

size_t hm_histogram_workspace_bytes(int, int) { return 64; }
int hm_channel_minmax(const double* val, const double* std_, int64_t n, int C, double* out, void*, void*) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || !val || !out) return HM_EINVAL;
    for (int c = 0; c < C; ++c) { out[2 * c] = kInf; out[2 * c + 1] = -kInf; }
    for (int64_t e = 0; e < n; ++e) {
        const double x = val[e];
        if (!(std::fabs(x) <= std::numeric_limits<double>::max())) continue;
        if (std_ && std_[e] == 0.0) continue;
        const int c = static_cast<int>(e % C);
        out[2 * c] = std::fmin(out[2 * c], x); out[2 * c + 1] = std::fmax(out[2 * c + 1], x);
    }
    return HM_OK;
}
int hm_channel_histogram(const double* val, const double* std_, int64_t n, int C, int channel_mask, const double* edges, int bins, double lo,
                         double hi, double* out, void*, void*) {
    if (n < 1 || C < 1 || C > HM_MAX_CHANNELS || bins < 1 || !val || !edges || !out || !(hi > lo)) return HM_EINVAL;
    for (int i = 0; i < C * bins; ++i) out[i] = 0.0;
    const double norm = static_cast<double>(bins) / (hi - lo);
    for (int64_t e = 0; e < n; ++e) {                                                                     // np.histogram, measurand.py:430-469
        const int c = static_cast<int>(e % C);
        if (!((channel_mask >> c) & 1)) continue;
        const double x = val[e];
        if (!(std::fabs(x) <= std::numeric_limits<double>::max())) continue;
        double w = 1.0;
        if (std_) { const double s = std_[e]; if (s == 0.0) continue; w = 1.0 / s; }                     // :457,460
        if (!(x >= lo && x <= hi)) continue;
        int idx = static_cast<int>((x - lo) * norm);
        if (idx == bins) idx -= 1;
        if (x < edges[idx]) idx -= 1;
        else if (x >= edges[idx + 1] && idx != bins - 1) idx += 1;
        out[c * bins + idx] += w;
    }
    return HM_OK;
}

// ---- the producers ---------------------------------------------------------------------------------
// welford_algorithm, modules/video_processing.py:161-219: per element, frames in order - delta = f - mean; mean += delta / n;
// m2 += delta * (f - mean) (:205-208), f = icrf[dn, c] (:200-201) or dn / 255 (:203)
int hm_welford_update(const void* const* frames, int n_frames, int64_t count_before, const double* icrf, double* mean, double* m2,
                      int64_t n_elems, int C, void*) {
    if (n_frames < 0 || n_frames > HM_MAX_FRAMES || count_before < 0 || n_elems < 0) return HM_EINVAL;
    if (count_before > (int64_t{1} << 40)) return HM_EUNSUPPORTED;              // (the device build's limit: one ABI, one answer)
    if (C < 1 || C > HM_MAX_CHANNELS) return HM_ESHAPE;
    if (n_elems % C != 0) return HM_ESHAPE;
    if (n_frames == 0 || n_elems == 0) return HM_OK;
    if (!frames || !mean) return HM_EINVAL;
    for (int k = 0; k < n_frames; ++k) if (!frames[k]) return HM_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n_elems; ++e) {
        const int c = static_cast<int>(e % C);
        double m = mean[e], q = m2 ? m2[e] : 0.0, cnt = static_cast<double>(count_before);
        for (int k = 0; k < n_frames; ++k) {
            cnt += 1.0;
            const int dn = static_cast<const uint8_t*>(frames[k])[e];
            const double f = icrf ? icrf[dn * C + c] : static_cast<double>(dn) / 255.0;
            const double delta = f - m;
            m = m + delta / cnt;
            q = q + delta * (f - m);
        }
        mean[e] = m;
        if (m2) m2[e] = q;
    }
    return HM_OK;
}

// np.around(x).astype(uint8) as the device build defines it: round half even, NaN and |x| >= 2^63 -> 0, else the low byte
static inline uint8_t round_to_u8(double x) {
    const double r = std::nearbyint(x);
    if (!(r == r) || r >= 9.2e18 || r <= -9.2e18) return 0;
    return static_cast<uint8_t>(static_cast<uint64_t>(static_cast<int64_t>(r)) & 255u);
}

int hm_welford_finalize(const double* mean, const double* m2, int64_t count, uint8_t* out_mean, uint8_t* out_std, int64_t n_elems, void*) {
    if (n_elems < 0 || count < 1) return HM_EINVAL;
    if (n_elems == 0) return HM_OK;
    if ((out_mean && !mean) || (out_std && !m2)) return HM_EINVAL;
    if (out_std && count < 2) return HM_EINVAL;                                   // m2 / (n - 1) needs two frames (:214)
    const double denom = static_cast<double>(count) - 1.0, root_n = std::sqrt(static_cast<double>(count));
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n_elems; ++e) {
        if (out_mean) out_mean[e] = round_to_u8(mean[e] * 255.0);                 // :210-211
        if (out_std) out_std[e] = round_to_u8(std::sqrt(m2[e] / denom) / root_n); // :214-215, as written
    }
    return HM_OK;
}
int64_t hm_welford_algorithmic_bytes(int n_frames, int with_m2, int64_t n_elems) { return n_elems * (static_cast<int64_t>(n_frames) + (with_m2 ? 32 : 16)); }

// _energy_function + analyze_linearity, modules/ICRF_calibration_exposure.py:66-145,148-201, for n_candidates ICRFs of one channel:
// per candidate and exposure pair (i < j, np.triu_indices order) the (weighted) mean of |v_i - v_j t_i/t_j| (/ (v_j t_i/t_j) when
// relative) over the pixels whose values lie inside [icrf[lower], icrf[upper]]; the energy is the nanmean over the pairs, +inf for NaN
size_t hm_linearity_energy_workspace_bytes(int64_t, int, int) { return 0; }
int hm_linearity_energy(const uint8_t* dn, const double* std_, const double* exposures, const double* icrf, const uint8_t* valid,
                        int n_candidates, int lower, int upper, int use_relative, int64_t n_pixels, int n_frames,
                        double* out_pairs, double* out_energy, void*, void*) {
    if (n_candidates < 0 || n_pixels < 0) return HM_EINVAL;
    if (n_frames < 2 || n_frames > HM_MAX_FRAMES) return HM_ESHAPE;
    if (lower < 0 || lower > 255 || upper < 0 || upper > 255) return HM_EINVAL;
    if (n_candidates == 0) return HM_OK;
    if (n_candidates > 65535) return HM_EUNSUPPORTED;
    if (!dn || !exposures || !icrf || !out_energy) return HM_EINVAL;
    const int N = n_frames, pairs = N * (N - 1) / 2;
    std::vector<int> pi(pairs), pj(pairs);
    { int p = 0; for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j, ++p) { pi[p] = i; pj[p] = j; } }
    std::vector<double> res(static_cast<size_t>(n_candidates) * pairs);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < n_candidates; ++b)
        for (int p = 0; p < pairs; ++p) {
            double& r = res[static_cast<size_t>(b) * pairs + p];
            if (valid && !valid[b]) { r = kNaN; continue; }
            const double* lut = icrf + static_cast<int64_t>(b) * 256;
            const double lo = lut[lower], hi = lut[upper];
            const int i = pi[p], j = pj[p];
            const double ratio = exposures[i] / exposures[j];                            // :100
            double num = 0.0, den = 0.0, bnum = 0.0, bden = 0.0;                       // sums in blocks of 1024 pixels (a million-term running sum
            for (int64_t px = 0; px < n_pixels; ++px) {                                  // loses three more digits than NumPy's pairwise one)
                if ((px & 1023) == 0) { num += bnum; den += bden; bnum = 0.0; bden = 0.0; }
                double vi = lut[dn[px * N + i]], vj = lut[dn[px * N + j]];
                if (vi < lo || vi > hi) vi = kNaN;                                       // :96-97
                if (vj < lo || vj > hi) vj = kNaN;
                const double scaled = vj * ratio;                                        // :111
                double d = vi - scaled;                                                  // :114
                if (use_relative) d = d / scaled;                                        // :117
                const double ad = std::fabs(d);                                          // :120
                if (std_) {
                    const double si = std_[px * N + i], sj = std_[px * N + j];
                    double sigma;
                    if (use_relative) {
                        const double u = si / scaled, v = (vi * sj) / (ratio * (vj * vj));   // :127
                        sigma = std::sqrt(u * u + v * v);
                    } else {
                        const double v = ratio * sj;
                        sigma = std::sqrt(si * si + v * v);                              // :129
                    }
                    const double w = 1.0 / sigma;
                    if (std::isfinite(ad) && sigma != 0.0 && w == w) { bnum += ad * w; bden += w; }   // :133-134, general_functions.py:164-174
                } else if (ad == ad) { bnum += ad; bden += 1.0; }                        // :138
            }
            num += bnum; den += bden;
            r = num / den;                                                               // 0 / 0 = NaN: no contributing pixel
        }
    for (int b = 0; b < n_candidates; ++b) {
        const bool ok = !valid || valid[b];
        double sum = 0.0, cnt = 0.0;
        for (int p = 0; p < pairs; ++p) {
            const double r = res[static_cast<size_t>(b) * pairs + p];
            if (out_pairs) out_pairs[static_cast<int64_t>(b) * pairs + p] = r;
            if (r == r) { sum += r; cnt += 1.0; }
        }
        const double e = sum / cnt;                                                      // :196
        out_energy[b] = (ok && e == e) ? e : std::numeric_limits<double>::infinity();    // :197-200
    }
    return HM_OK;
}

}  // extern "C"
