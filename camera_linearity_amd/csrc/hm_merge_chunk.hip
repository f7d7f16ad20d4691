// hm_merge_chunk.hip - HDR merge of stacks of MORE than HM_MAX_FRAMES frames (gfx950).
//
// The reference's merge loop has no frame limit (modules/exposure_series.py:334,372 iterate over the whole series). The
// streaming kernels of hm_merge.hip take their frame pointers from the kernarg segment and therefore serve at most
// HM_MAX_FRAMES per launch; a longer stack is merged here, HM_MAX_FRAMES frames per launch, with the running sums kept in
// memory between the launches:
//
//   val-only   one launch per chunk:  S += w_i, acc = fma(w_i g_i, 1/t_i, acc)   (S in the sum buffer, acc in out_val);
//              the last chunk's launch finishes val = acc / S (+ flat field).
//   with std   launches A (one per chunk): S += w_i (reads the frame bytes only); then launches B (one per chunk):
//              acc and var of exposure_series.py:388-389 with the final 1/S and 1/S**2 (acc in out_val, var in out_std);
//              the last one finishes val = acc / S, std = sqrt(var) (+ flat field).
//
// The operation sequence per output element is merge_one_element()'s of hm_merge.hip - S in frame order, the numerator and
// the variance accumulated with fma in frame order, acc / S last; a float64 stored and re-loaded between two launches is the
// same float64 - so a stack merged in chunks of ANY size gives the bits of the one-launch kernels (tests force chunks of 2-5
// frames on short stacks and compare with merge_generic). Hot pixels (dark maps) are handled inline: a lane whose frame is
// hot takes the k x k median itself (lane_median), value and std, exactly as the patch kernel does.
//
// Element ownership: thread q owns elements 2q, 2q + 1 (one ushort / one 16-byte load per frame, one 16-byte store per
// array); unaligned tiles and odd tails fall back to one element per thread. Bound: HBM; traffic = the algorithmic bytes
// + 16 B (val-only) or 24-40 B (std) of running sums per element and chunk + the frame bytes a second time with std.
#include "hm_common.h"

namespace hm {

typedef double f64x2c __attribute__((ext_vector_type(2)));

struct ChunkK {
    const void*    frame[HM_MAX_FRAMES];
    const double*  sd[HM_MAX_FRAMES];
    const uint8_t* dark[HM_MAX_FRAMES];
    double  inv_t[HM_MAX_FRAMES];
    int32_t dark_min[HM_MAX_FRAMES];
    const double* icrf;
    const double* icrf_diff;
    const double* w_lut;
    const double* dw_lut;
    const uint8_t* flat_u8;
    const double*  flat_f64;
    const double*  flat_std;
    double ff_mean[HM_MAX_CHANNELS];
    double ff_std_mean[HM_MAX_CHANNELS];
    double* out_val;
    double* out_std;
    double* s_acc;                 // running / final sum of weights (the caller's out_sum_w or the workspace)
    int64_t n_elems, elem0, in_off, H, W, row0, buf_row0;
    int32_t n_frames, C, median_k, has_flat;
    int32_t first, last;           // first / last chunk of the stack
};

enum { PH_S = 0, PH_VAL = 1, PH_STD = 2 };

template <bool F64IN, int PH, bool HOT, int VEC>
__global__ __launch_bounds__(256) void merge_chunk(const ChunkK a) {
    __shared__ double t_w[256], t_dw[256], t_g[256 * HM_MAX_CHANNELS], t_d[256 * HM_MAX_CHANNELS];
    const int C = a.C, N = a.n_frames;
    for (int i = threadIdx.x; i < 256; i += 256) {
        t_w[i] = F64IN ? 0.0 : a.w_lut[i];
        t_dw[i] = (!F64IN && PH == PH_STD) ? a.dw_lut[i] : 0.0;
    }
    if (PH != PH_S) {
        for (int i = threadIdx.x; i < 256 * C; i += 256) {
            t_g[i] = a.icrf[i];
            t_d[i] = PH == PH_STD ? a.icrf_diff[i] : 0.0;
        }
    }
    __syncthreads();
    const int64_t n_items = a.n_elems / VEC;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; q < n_items; q += stride) {
        const int64_t e0 = a.elem0 + q * VEC;                 // relative to row0
        const int64_t ei0 = a.in_off + e0;                    // inside the input buffers
        int ch[VEC];
        int64_t row[VEC], col[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int64_t e = e0 + j;
            ch[j] = static_cast<int>(e % C);
            if (HOT) {
                const int64_t wc = a.W * C;
                row[j] = a.row0 + e / wc;
                col[j] = (e % wc) / C;
            } else { row[j] = 0; col[j] = 0; }
        }
        double S[VEC], acc[VEC], var[VEC], invS[VEC], invS2[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { S[j] = 0.0; acc[j] = 0.0; var[j] = 0.0; invS[j] = 0.0; invS2[j] = 0.0; }
        // ---- the running sums of the chunks before this one
        if (PH == PH_STD || !a.first) {
            if (VEC == 2) { const f64x2c s = *reinterpret_cast<const f64x2c*>(a.s_acc + e0); S[0] = s.x; S[VEC - 1] = s.y; }
            else S[0] = a.s_acc[e0];
        }
        if (PH != PH_S && !a.first) {
            if (VEC == 2) { const f64x2c v = *reinterpret_cast<const f64x2c*>(a.out_val + e0); acc[0] = v.x; acc[VEC - 1] = v.y; }
            else acc[0] = a.out_val[e0];
            if (PH == PH_STD) {
                if (VEC == 2) { const f64x2c v = *reinterpret_cast<const f64x2c*>(a.out_std + e0); var[0] = v.x; var[VEC - 1] = v.y; }
                else var[0] = a.out_std[e0];
            }
        }
        if (PH == PH_STD) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                invS[j] = 1.0 / S[j];
                invS2[j] = 1.0 / (S[j] * S[j]);               // 1 / S**2, exposure_series.py:343
            }
        }
        // ---- this chunk's frames, in order
        for (int i = 0; i < N; ++i) {
            const bool start = a.first && i == 0;             // the stack's first frame initialises instead of accumulating
            double v[VEC], sdv[VEC];
            uint32_t dn[VEC];
            bool hot[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { v[j] = 0.0; sdv[j] = 0.0; dn[j] = 0u; hot[j] = false; }
            if (HOT && a.dark[i]) {                           // wave-uniform pointer test
                if (VEC == 2) {
                    const uint32_t d = *reinterpret_cast<const uint16_t*>(a.dark[i] + ei0);
                    hot[0] = static_cast<int>(d & 255u) >= a.dark_min[i];
                    hot[VEC - 1] = static_cast<int>(d >> 8) >= a.dark_min[i];
                } else hot[0] = static_cast<int>(a.dark[i][ei0]) >= a.dark_min[i];
            }
            if (F64IN) {
                const double* f = static_cast<const double*>(a.frame[i]);
                if (VEC == 2) { const f64x2c x = __builtin_nontemporal_load(reinterpret_cast<const f64x2c*>(f + ei0)); v[0] = x.x; v[VEC - 1] = x.y; }
                else v[0] = f[ei0];
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (HOT && hot[j]) v[j] = lane_median(f, a.H, a.W, C, a.buf_row0, row[j], col[j], ch[j], a.median_k);
            } else {
                const uint8_t* f = static_cast<const uint8_t*>(a.frame[i]);
                if (VEC == 2) {
                    const uint32_t r = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(f + ei0));
                    dn[0] = r & 255u; dn[VEC - 1] = r >> 8;
                } else dn[0] = f[ei0];
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (HOT && hot[j]) dn[j] = lane_median(f, a.H, a.W, C, a.buf_row0, row[j], col[j], ch[j], a.median_k);
            }
            if (PH == PH_STD) {
                const double* sp = a.sd[i];
                if (VEC == 2) { const f64x2c x = __builtin_nontemporal_load(reinterpret_cast<const f64x2c*>(sp + ei0)); sdv[0] = x.x; sdv[VEC - 1] = x.y; }
                else sdv[0] = sp[ei0];
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (HOT && hot[j]) sdv[j] = lane_median(sp, a.H, a.W, C, a.buf_row0, row[j], col[j], ch[j], a.median_k);
            }
            const double it = a.inv_t[i];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double w, dw = 0.0;
                uint32_t idx;
                if (F64IN) {
                    const double dv = v[j] - 0.5;
                    w = gauss_weight(dv);                                            // measurand.py:615
                    if (PH == PH_STD) dw = (-60.0 * dv) * w;                         // :616
                    idx = static_cast<uint32_t>(static_cast<int64_t>(rint(v[j] * 255.0))) & 255u;   // :503
                } else {
                    idx = dn[j];
                    w = t_w[idx];
                    if (PH == PH_STD) dw = t_dw[idx];
                }
                if (PH != PH_STD) S[j] = start ? w : S[j] + w;                       // exposure_series.py:340
                if (PH != PH_S) {
                    const double g = t_g[idx * C + ch[j]];
                    const double wg = w * g;
                    acc[j] = start ? wg * it : fma(wg, it, acc[j]);                  // :388 numerator
                    if (PH == PH_STD) {
                        const double dg = t_d[idx * C + ch[j]] * sdv[j];             // measurand.py:512
                        const double A = (dw * g + w * dg) * invS[j] - ((dw * w) * g) * invS2[j];   // :389
                        const double term = (A * dg) * it;
                        var[j] = start ? term * term : fma(term, term, var[j]);
                    }
                }
            }
        }
        // ---- store the running sums, or finish
        if (PH != PH_STD) {
            if (VEC == 2) { f64x2c s; s.x = S[0]; s.y = S[VEC - 1]; *reinterpret_cast<f64x2c*>(a.s_acc + e0) = s; }
            else a.s_acc[e0] = S[0];
        }
        if (PH != PH_S) {
            double val[VEC], so[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                val[j] = acc[j];
                so[j] = var[j];
                if (a.last) {
                    val[j] = acc[j] / S[j];
                    so[j] = PH == PH_STD ? sqrt(var[j]) : 0.0;                       // :394
                    if (a.has_flat) {                                                // measurand.py:585-602
                        const int64_t e = e0 + j;
                        const double F = a.flat_u8 ? static_cast<double>(a.flat_u8[e]) / 255.0 : a.flat_f64[e];
                        flat_field_math(F, PH == PH_STD ? 1.0 / (F * F) : 1.0, PH == PH_STD ? a.flat_std[e] : 0.0, a.ff_mean[ch[j]],
                                        a.ff_std_mean[ch[j]], PH == PH_STD, val[j], so[j]);
                    }
                }
            }
            if (VEC == 2) {
                f64x2c o; o.x = val[0]; o.y = val[VEC - 1]; *reinterpret_cast<f64x2c*>(a.out_val + e0) = o;
                if (PH == PH_STD) { f64x2c p; p.x = so[0]; p.y = so[VEC - 1]; *reinterpret_cast<f64x2c*>(a.out_std + e0) = p; }
            } else {
                a.out_val[e0] = val[0];
                if (PH == PH_STD) a.out_std[e0] = so[0];
            }
        }
    }
}

template <bool F64IN, int PH>
static int launch_chunk_ph(const ChunkK& k, bool hot, bool vec, hipStream_t st) {
    auto go = [&](auto kernel, const ChunkK& kk, int vecn) {
        const unsigned grid = stream_grid(kk.n_elems / vecn, 256, 8);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, kk);
        return launch_status();
    };
    int rc = HM_OK;
    ChunkK body = k;
    if (vec) {
        body.n_elems = k.n_elems & ~int64_t{1};
        if (body.n_elems > 0)
            rc = hot ? go(merge_chunk<F64IN, PH, true, 2>, body, 2) : go(merge_chunk<F64IN, PH, false, 2>, body, 2);
        if (rc != HM_OK || body.n_elems == k.n_elems) return rc;
        body.elem0 = k.elem0 + body.n_elems;                  // the odd last element
        body.n_elems = 1;
    }
    return hot ? go(merge_chunk<F64IN, PH, true, 1>, body, 1) : go(merge_chunk<F64IN, PH, false, 1>, body, 1);
}

static int launch_chunk(const ChunkK& k, bool f64in, int phase, bool hot, bool vec, hipStream_t st) {
    if (f64in) {
        if (phase == PH_S) return launch_chunk_ph<true, PH_S>(k, hot, vec, st);
        if (phase == PH_VAL) return launch_chunk_ph<true, PH_VAL>(k, hot, vec, st);
        return launch_chunk_ph<true, PH_STD>(k, hot, vec, st);
    }
    if (phase == PH_S) return launch_chunk_ph<false, PH_S>(k, hot, vec, st);
    if (phase == PH_VAL) return launch_chunk_ph<false, PH_VAL>(k, hot, vec, st);
    return launch_chunk_ph<false, PH_STD>(k, hot, vec, st);
}

// Called by hm_merge() with VALIDATED arguments (geometry, pointers, alignment to 8 bytes, exposures > 0) for a stack of more than
// HM_MAX_FRAMES frames, or - testing - with chunk_frames < n_frames on a short one. `describe`: append the kernel names instead of launching.
int merge_chunked(const hm_merge_args* g, int chunk_frames, std::string* describe, hipStream_t st) {
    const int N = g->n_frames, C = g->channels;
    const bool f64in = g->frames_f64 != nullptr, with_std = g->stds != nullptr && g->out_val != nullptr;
    const int64_t E = g->rows * g->width * C;
    double* s_acc = g->out_sum_w ? g->out_sum_w : static_cast<double*>(g->frames_workspace);
    if (!s_acc || (!g->out_sum_w && g->frames_workspace_bytes < static_cast<size_t>(E) * 8u)) return HM_EUNSUPPORTED;
    if (!aligned(s_acc, 8)) return HM_EALIGN;
    bool hot = false;
    if (g->darks_u8) for (int i = 0; i < N; ++i) hot = hot || g->darks_u8[i] != nullptr;
    const int64_t in_off = (g->row0 - g->buf_row0) * g->width * C;
    // two elements per thread need 2-byte aligned byte streams and 16-byte aligned float64 arrays
    bool vec = aligned(s_acc, 16) && (!g->out_val || aligned(g->out_val, 16)) && (!g->out_std || aligned(g->out_std, 16));
    for (int i = 0; i < N && vec; ++i) {
        vec = f64in ? aligned(g->frames_f64[i] + in_off, 16) : aligned(g->frames_u8[i] + in_off, 2);
        if (vec && with_std) vec = aligned(g->stds[i] + in_off, 16);
        if (vec && hot && g->darks_u8[i]) vec = aligned(g->darks_u8[i] + in_off, 2);
    }
    if (describe) {
        char buf[160];
        snprintf(buf, sizeof(buf), "merge_chunk<f64in=%d,%s,hot=%d,vec=%d>(N=%d in chunks of %d)", f64in, with_std ? "S + std" : (g->out_val ? "val" : "S"),
                 hot, vec ? 2 : 1, N, chunk_frames);
        if (!describe->empty()) *describe += " + ";
        *describe += buf;
        return HM_OK;
    }
    const int n_pass = with_std ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int phase = with_std ? (pass == 0 ? PH_S : PH_STD) : (g->out_val ? PH_VAL : PH_S);
        for (int k0 = 0; k0 < N; k0 += chunk_frames) {
            ChunkK k{};
            const int n = N - k0 < chunk_frames ? N - k0 : chunk_frames;
            for (int i = 0; i < n; ++i) {
                k.frame[i] = f64in ? static_cast<const void*>(g->frames_f64[k0 + i]) : static_cast<const void*>(g->frames_u8[k0 + i]);
                k.sd[i] = with_std ? g->stds[k0 + i] : nullptr;
                k.inv_t[i] = 1.0 / g->exposures[k0 + i];
                k.dark[i] = hot ? g->darks_u8[k0 + i] : nullptr;
                k.dark_min[i] = hot && g->dark_min_dn ? g->dark_min_dn[k0 + i] : 256;
            }
            k.icrf = g->icrf; k.icrf_diff = g->icrf_diff; k.w_lut = g->w_lut; k.dw_lut = g->dw_lut;
            k.flat_u8 = g->flat_u8; k.flat_f64 = g->flat_f64; k.flat_std = g->flat_std;
            for (int c = 0; c < HM_MAX_CHANNELS; ++c) { k.ff_mean[c] = g->ff_mean[c]; k.ff_std_mean[c] = g->ff_std_mean[c]; }
            k.out_val = g->out_val; k.out_std = g->out_std; k.s_acc = s_acc;
            k.n_elems = E; k.elem0 = 0; k.in_off = in_off;
            k.H = g->height; k.W = g->width; k.row0 = g->row0; k.buf_row0 = g->buf_row0;
            k.n_frames = n; k.C = C; k.median_k = hot ? g->median_k : 3; k.has_flat = (g->flat_u8 || g->flat_f64) ? 1 : 0;
            k.first = k0 == 0; k.last = k0 + n == N;
            const int rc = launch_chunk(k, f64in, phase, hot, vec, st);
            if (rc != HM_OK) return rc;
        }
    }
    return HM_OK;
}

}  // namespace hm

// Bytes of hm_merge_args.frames_workspace a call needs: 0 up to HM_MAX_FRAMES frames or when the call has an out_sum_w
// (the sum of weights is then accumulated there), else one float64 per output element.
extern "C" size_t hm_merge_frames_workspace_bytes(int n_frames, int64_t n_elems, int has_out_sum_w) {
    if (n_frames <= HM_MAX_FRAMES || has_out_sum_w || n_elems <= 0) return 0;
    return static_cast<size_t>(n_elems) * 8u;
}
