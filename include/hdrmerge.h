/*
 * hdrmerge.h - C ABI of the MI355X-native HDR-merge / linearization engine (libhdrmerge.so).
 *
 * This is the drop-in boundary for ONE path of samivout/camera_linearity: per-pixel ICRF-LUT
 * linearization -> Gaussian-weighted HDR merge -> first-order uncertainty propagation, with
 * dark-frame hot-pixel filtering and flat-field correction. The reference has no FFI of its own;
 * its extension point is the array-backend protocol of AbstractMeasurand
 * (modules/measurand.py:26-32, modules/cupy_measurand.py:28-39, modules/measurand_factory.py:10-14).
 * A maintainer binds these entry points with ctypes from a third backend class next to
 * NumpyMeasurand / CupyMeasurand; each declaration below cites the reference
 * function whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) unless marked [host];
 *     buffers are caller-owned, contiguous, and never retained after the call returns;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call is
 *     asynchronous on that stream and performs no allocation and no host synchronisation, so it may
 *     be captured in a hipGraph;
 *   - images are row-major H x W x C with the channel innermost (OpenCV BGR order,
 *     modules/global_settings.py:32); "n" counts ELEMENTS (H*W*C), element e has channel e % C;
 *   - 8-bit frames are digital numbers (DN) 0..255; their value is DN / 255.0
 *     (modules/image_set.py:223); BITS = 256, MAX_DN = 255 (modules/global_settings.py:35-37);
 *   - return value: HM_OK (0) or a negative HM_E* code; nothing throws. hm_strerror() names it.
 *
 * Two libraries export this ABI. libhdrmerge.so (the .hip files under csrc/) is the MI355X build described above. libhdrmerge_host.so
 * (csrc_host/hm_host.cpp, plain C++) is the HOST build behind Measurand(use_cupy=False) - the slot of the reference's NumpyMeasurand
 * (modules/measurand_factory.py:10-14): every pointer is then a HOST pointer, `stream` is ignored, calls are synchronous and workspaces
 * may be NULL; the hm_tiff_* decoders are not exported (they are host code of libhdrmerge.so already).
 * The device build never calls or falls back to the host build.
 */
#ifndef HDRMERGE_H
#define HDRMERGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history. 1: first release (hm_merge_args 264 bytes), later grown by hot_workspace / hot_workspace_bytes (280 bytes) without a bump.
 * 2: frames_workspace / frames_workspace_bytes (296 bytes), hm_merge_frames_workspace_bytes(), stacks of more than HM_MAX_FRAMES
 *    frames; hm_merge still accepts the 264- and 280-byte layouts (the missing tail reads as "no workspace").            */
#define HM_ABI_VERSION 2
#define HM_BITS 256
#define HM_MAX_FRAMES 32     /* frames per merge LAUNCH; hm_merge takes any number of frames (32 per launch, see frames_workspace) */
#define HM_MAX_CHANNELS 4

enum {
    HM_OK = 0,
    HM_EINVAL = -1,        /* bad argument (null pointer, non-positive size, bad enum)        */
    HM_EUNSUPPORTED = -2,  /* valid request this build cannot serve (C > 4; N > HM_MAX_FRAMES without frames_workspace / out_sum_w) */
    HM_EALIGN = -3,        /* a float64 buffer is not 8-byte aligned                           */
    HM_ELAUNCH = -4,       /* the HIP runtime rejected the launch (hipGetLastError != success) */
    HM_ENODEVICE = -5,     /* no usable gfx950 device                                          */
    HM_ESHAPE = -6         /* geometry inconsistent (tile outside image, halo too small, ...)  */
};

int hm_version(void);                 /* HM_ABI_VERSION of the loaded library          */
const char* hm_strerror(int code);    /* static string, never NULL                     */
int hm_device_info(int* n_devices, int* cu_count, int* lds_bytes, char* arch, int arch_len);
/* (diagnostic probes - clock, copy rate, address translation - are declared in hdrmerge_debug.h: they replace nothing in the reference) */

/* ------------------------------------------------------------------------------------------
 * Row 3 - AbstractMeasurand.apply_gaussian_weight (modules/measurand.py:606-618)
 *   w = e**(-30 (v-0.5)^2), dw = -60 (v-0.5) w.
 * hm_gaussian_weight_f64 evaluates it analytically on float64 values; hm_gaussian_weight_u8 looks
 * the DN up in the caller's 256-entry tables (bit-identical to NumPy when the caller built the
 * tables with NumPy). Either output may be NULL.
 * ------------------------------------------------------------------------------------------ */
int hm_gaussian_weight_f64(const double* v, double* w, double* dw, int64_t n, void* stream);
int hm_gaussian_weight_u8(const uint8_t* dn, const double* w_lut, const double* dw_lut,
                          double* w, double* dw, int64_t n, void* stream);
/* [host] helper for C callers: fills w_lut/dw_lut (256 each) with libm pow(); may differ from
 * NumPy's `np.e ** x` in the last ulp. */
int hm_gaussian_weight_lut_host(double* w_lut /*[host]*/, double* dw_lut /*[host]*/);

/* ------------------------------------------------------------------------------------------
 * Row 1 - ImageSet.load_value_image arithmetic (modules/image_set.py:223): out = dn / 255.0
 * ------------------------------------------------------------------------------------------ */
int hm_u8_to_unit_f64(const uint8_t* dn, double* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row 4 - AbstractMeasurand.linearize / _linearize_channel / _linearize_single
 *         (modules/measurand.py:471-541)
 *   idx  = dn                       (integer input, :505/:533)
 *        = uint8(rint(v * 255))     (float input, round-half-even then wrap mod 256, :503/:531)
 *   out_val[e] = icrf[idx[e] * lut_stride + (e % C) * (lut_stride > 1)]
 *   out_std[e] = icrf_diff[...same...] * std[e]    only when std, icrf_diff and out_std are given
 * icrf / icrf_diff are (256, C) row-major (lut_stride == C) or 1-D (256,) applied to every channel
 * (lut_stride == 1, the `_linearize_single` form used by ImageSet.calculate_numerical_STD,
 * modules/image_set.py:383). out_idx (nullable) exposes the uint8 index for the bit-exact check.
 * ------------------------------------------------------------------------------------------ */
int hm_linearize_u8(const uint8_t* dn, const double* std, const double* icrf, const double* icrf_diff,
                    double* out_val, double* out_std, int64_t n, int C, int lut_stride, void* stream);
int hm_linearize_f64(const double* v, const double* std, const double* icrf, const double* icrf_diff,
                     double* out_val, double* out_std, uint8_t* out_idx,
                     int64_t n, int C, int lut_stride, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rows 5-9 fused - ExposureSeries._precalculate_sum_of_weights + _compute_HDR_image_set
 * (modules/exposure_series.py:317-397) with the optional hot-pixel prologue
 * (ImageSet.bad_pixel_filter -> filter_larger_than_by_map, modules/measurand.py:543-557) and the
 * optional flat-field epilogue (normalize_by_map, modules/measurand.py:559-604).
 *
 *   S      = sum_i w(v_i)                                                       (:340)
 *   val    = sum_i (w_i g_i) / (S t_i)                                          (:388)
 *   var    = sum_i ( ((dw_i g_i + w_i dg_i)/S - (dw_i w_i g_i)/S^2) dg_i/t_i )^2 (:389)
 *   std    = var ** (1/2)                                                       (:394)
 * with v_i the frame value BEFORE linearization, g_i = icrf[idx_i, c], dg_i = icrf_diff[idx_i, c] * std_i.
 * One streaming launch reads every frame / std / flat byte once and writes every output byte once; when dark
 * maps are given a second, stream-ordered launch reads each dark byte once and recomputes the (rare) elements
 * that have a hot frame with the k x k medians substituted. Results do not depend on how the image is tiled.
 *
 * Tiling: a call produces image rows [row0, row0 + rows) of an H x W x C image.
 * Input frame / std / dark pointers address image row `buf_row0` (<= row0) and hold `buf_rows`
 * rows, so a row-tile shard passes its slice plus the median halo; the median uses scipy 'reflect'
 * only at the true image edges (0 and H). Output and flat pointers address row0.
 * ------------------------------------------------------------------------------------------ */
typedef struct hm_merge_args {
    uint32_t struct_size;         /* sizeof(hm_merge_args), for ABI evolution                       */
    int32_t  n_frames;            /* N >= 1, ascending exposure (N > HM_MAX_FRAMES: see frames_workspace) */
    int32_t  channels;            /* C, 1..HM_MAX_CHANNELS                                          */
    int32_t  variant;             /* 0 = library default; >0 selects a tuning variant (tools/tune_merge.py), -1 forces the generic kernel,
                                     <= -2 forces the chunked path with -variant frames per launch (tests) */
    int64_t  height;              /* H of the full image                                            */
    int64_t  width;               /* W                                                              */
    int64_t  row0, rows;          /* output rows of this call                                       */
    int64_t  buf_row0, buf_rows;  /* image rows covered by the input buffers                        */
                                  /* (the streaming kernels read element pairs: they need the call's first input element,
                                     (row0 - buf_row0) * W * C elements into every frame buffer, at an EVEN offset - and float64
                                     frames / stds / outputs / flat buffers 16-byte aligned there; otherwise the call runs in the
                                     one-element-per-thread generic kernel: same bits, 3-4 x slower. Only an odd W * C with an odd
                                     number of halo rows above the tile breaks this: give such a tile one more row above.)          */

    const uint8_t* const* frames_u8;   /* [host] N device pointers to uint8 DN frames, or NULL       */
    const double*  const* frames_f64;  /* [host] N device pointers to float64 value frames, or NULL  */
    const double*  const* stds;        /* [host] N device pointers to float64 std frames, or NULL    */
    const double*  exposures;          /* [host] N exposure times in seconds (features['exposure'])  */

    const double* icrf;           /* (256, C) float64                                               */
    const double* icrf_diff;      /* (256, C) float64; required when stds != NULL                   */
    const double* w_lut;          /* (256,) weight table on the DN grid (u8 frames)                 */
    const double* dw_lut;         /* (256,) weight derivative table (u8 frames with stds)           */

    /* hot-pixel prologue: frame i is filtered where dark_u8[i][e] >= dark_min_dn[i]                 */
    const uint8_t* const* darks_u8;    /* [host] N device pointers (NULL entries = no filter), or NULL */
    const int32_t* dark_min_dn;        /* [host] N thresholds in DN (1..256)                         */
    int32_t  median_k;            /* odd kernel size (gs.MEDIAN_FILTER_KERNEL_SIZE), 3..7           */
    int32_t  _pad0;

    /* flat-field epilogue (all NULL = off)                                                          */
    const uint8_t* flat_u8;       /* flat value as DN (value = DN/255), or                          */
    const double*  flat_f64;      /* flat value as float64                                          */
    const double*  flat_std;      /* flat uncertainty, float64 (required when std output is on)     */
    double ff_mean[HM_MAX_CHANNELS];     /* per-channel ROI mean of flat value  (measurand.py:582)  */
    double ff_std_mean[HM_MAX_CHANNELS]; /* per-channel ROI mean of flat std    (measurand.py:583)  */

    double* out_val;              /* rows x W x C float64, or NULL (sum-of-weights only)            */
    double* out_std;              /* rows x W x C float64; required iff stds != NULL                */
    double* out_sum_w;            /* optional S (rows x W x C), NULL to skip                        */

    /* hot-pixel queue (optional; used when darks_u8 is set). With a workspace the dark maps are scanned into a queue of hot
     * element indices and a balanced second kernel patches one queued element per lane - its cost follows the number of
     * hot elements, dense maps included. NULL: the scan recomputes hot elements one at a time per wave (sparse maps only).
     * Device memory, 16-byte aligned, hm_merge_hot_workspace_bytes(rows * W * C) bytes recommended (down to
     * hm_merge_hot_workspace_min_bytes() is legal: a queue that overflows makes the patch kernel test every element itself;
     * anything smaller selects the NULL path); owned by this call until it has completed on `stream`.                      */
    void*    hot_workspace;
    size_t   hot_workspace_bytes;

    /* Stacks of more than HM_MAX_FRAMES frames (the reference's loop has no limit, modules/exposure_series.py:334,372) are merged
     * HM_MAX_FRAMES frames per launch with the running sums in memory: the numerator in out_val, the variance in out_std, and the sum
     * of weights in out_sum_w when the call has one, else HERE: device memory, 16-byte aligned, hm_merge_frames_workspace_bytes()
     * bytes (one float64 per output element), owned by the call until it has completed on `stream`. NULL with N > HM_MAX_FRAMES and
     * no out_sum_w: HM_EUNSUPPORTED. Same operation sequence per element as the one-launch kernels (bit-identical for any chunking).  */
    void*    frames_workspace;
    size_t   frames_workspace_bytes;
} hm_merge_args;

int hm_merge(const hm_merge_args* args /*[host]*/, void* stream);
/* Recommended / smallest accepted size of hm_merge_args.hot_workspace for a call with n_elems = rows * W * C output elements
 * (counters + a table of 8 bytes per 65 536 elements + the queue: a quarter of the elements / one entry). */
size_t hm_merge_hot_workspace_bytes(int64_t n_elems);
size_t hm_merge_hot_workspace_min_bytes(int64_t n_elems);
/* Bytes of hm_merge_args.frames_workspace for a call with n_frames frames and n_elems = rows * W * C output elements: 0 up to
 * HM_MAX_FRAMES frames or when the call has an out_sum_w, else 8 * n_elems. */
size_t hm_merge_frames_workspace_bytes(int n_frames, int64_t n_elems, int has_out_sum_w);

/* Algorithmic HBM bytes one hm_merge call moves: every input byte once, every output
 * byte once, LUTs excluded. Used by bench.py for roofline.achieved. */
int64_t hm_merge_algorithmic_bytes(const hm_merge_args* args /*[host]*/);
/* The kernels hm_merge() would launch for these arguments, in launch order, as text ("merge_u8_val3<N=7,...> +
 * merge_fixup_hot<...>"): the same dispatch code runs with launching switched off. Same status codes as hm_merge. */
int hm_merge_describe(const hm_merge_args* args, char* buf, int buf_len);

/* ------------------------------------------------------------------------------------------
 * Row 8 standalone - AbstractMeasurand.filter_larger_than_by_map (modules/measurand.py:543-557):
 * out = where(map >= min / map > thr, median_kxk(x, axes=(0,1), mode='reflect'), x), one call per array
 * (val: uint8 or float64; std: float64). map_f64 is compared with `> thr` in value units,
 * map_u8 with `>= min_dn` in DN.
 * ------------------------------------------------------------------------------------------ */
int hm_hot_pixel_filter_u8(const uint8_t* x, const uint8_t* map_u8, const double* map_f64,
                           int min_dn, double thr, int median_k, uint8_t* out,
                           int64_t H, int64_t W, int C, void* stream);
int hm_hot_pixel_filter_f64(const double* x, const uint8_t* map_u8, const double* map_f64,
                            int min_dn, double thr, int median_k, double* out,
                            int64_t H, int64_t W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row 9 standalone - flat-field ROI mean (modules/measurand.py:561-583) and normalize_by_map
 * (modules/measurand.py:585-604).
 * hm_roi_mean: per-channel mean over rows [x0,x1) x cols [y0,y1) of an H x W x C image, two-stage
 * deterministic reduction; `workspace` needs hm_roi_mean_workspace_bytes() bytes; out_mean is C
 * float64 on the device.
 * ------------------------------------------------------------------------------------------ */
size_t hm_roi_mean_workspace_bytes(void);
int hm_roi_mean_u8(const uint8_t* img, int64_t H, int64_t W, int C,
                   int64_t x0, int64_t x1, int64_t y0, int64_t y1,
                   double* out_mean, void* workspace, void* stream);
int hm_roi_mean_f64(const double* img, int64_t H, int64_t W, int C,
                    int64_t x0, int64_t x1, int64_t y0, int64_t y1,
                    double* out_mean, void* workspace, void* stream);
int hm_normalize_by_map(const double* val, const double* std,
                        const uint8_t* flat_u8, const double* flat_f64, const double* flat_std,
                        const double* ff_mean /*[host] C*/, const double* ff_std_mean /*[host] C*/,
                        double* out_val, double* out_std, int64_t n, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row 10 - Measurand operators with first-order uncertainty propagation
 * (modules/measurand.py:106-279). Operands broadcast NumPy-style: `shape` is the broadcast result
 * shape (ndim <= 6), strides are in ELEMENTS with 0 on broadcast axes. s1 / s2 may be NULL (treated
 * as zeros, :121-124); out_std must be NULL iff both are NULL.
 * ------------------------------------------------------------------------------------------ */
enum { HM_OP_ADD = 0, HM_OP_SUB = 1, HM_OP_MUL = 2, HM_OP_DIV = 3, HM_OP_POW = 4 };
enum { HM_UOP_NEG = 0, HM_UOP_LOG_E = 1, HM_UOP_LOG_10 = 2 };
#define HM_MAX_DIMS 6
int hm_binary_op(int op, const double* x1, const double* s1, const double* x2, const double* s2,
                 double* out_val, double* out_std, int ndim, const int64_t* shape /*[host]*/,
                 const int64_t* strides1 /*[host]*/, const int64_t* strides2 /*[host]*/, void* stream);
int hm_unary_op(int op, const double* x, const double* s, double* out_val, double* out_std,
                int64_t n, void* stream);

/* Measurand.__pow__ with a plain scalar exponent (modules/measurand.py:217-241; `S ** 2`, `** (1/2)` of the merge loop,
 * modules/exposure_series.py:343,394): out = x ** exponent, out_std = |exponent * x ** (exponent - 1)| * s (the exponent
 * has no uncertainty; its term of :237-238 is evaluated only where it is not +0). Exponents 2, 0.5, 1 and integers in
 * [-8, 8] avoid pow(). s / out_std are NULL together. */
int hm_pow_scalar(const double* x, const double* s /*nullable*/, double exponent, double* out_val, double* out_std /*nullable*/,
                  int64_t n, void* stream);

/* Measurand.extract (modules/measurand.py:352-373, `lib.take(val, dims, axis)`): out[o, k, i] = in[o, indices[k], i]
 * for the dense (outer, axis_len, inner) view of the array; axis=None in the reference is outer = inner = 1 on the
 * flattened array. std (nullable, with out_std) is taken with the same indices. Negative indices count from the end;
 * an index out of range returns HM_ESHAPE (np.take raises IndexError). At most HM_TAKE_MAX indices per call. */
#define HM_TAKE_MAX 16
int hm_take_axis(const double* x, const double* s, double* out_val, double* out_std, int64_t outer, int64_t axis_len,
                 int64_t inner, const int64_t* indices /*[host]*/, int n_indices, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY.md 8(f)-1 - linearity statistics (ExposureSeries.process_linearity, modules/exposure_series.py:421-446)
 *   hm_apply_thresholds     apply_thresholds (modules/measurand.py:375-428), in place: elements with
 *                           val < lower[c] or val > upper[c] become NaN in val and std; pass -inf / +inf for "none".
 *   hm_compute_difference   compute_difference (:620-655): abs = x - m*y, rel = abs / (m*y) and their std.
 *   hm_interpolate          interpolate (:657-681), std formula as written.
 *   hm_channel_statistics   compute_dimension_statistics (:318-350) over every axis but the last (the
 *                           axis=(0,1) call of exposure_series.py:446): NaN-ignoring mean / std per channel,
 *                           weighted by 1/std when std is given, plus error = nanmean(std). out is 3*C
 *                           float64 on the device: [mean | std | error]; error is NaN without std.
 * ------------------------------------------------------------------------------------------ */
#define HM_THRESHOLD_MAX_CHANNELS 32   /* hm_apply_thresholds: channels on the last axis (a slower one-element-per-thread kernel above HM_MAX_CHANNELS) */
int hm_apply_thresholds(double* val, double* std /*nullable*/, const double* lower /*[host] C*/,
                        const double* upper /*[host] C*/, int64_t n, int C, void* stream);
int hm_compute_difference(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                          double* out_abs, double* out_abs_std, double* out_rel, double* out_rel_std,
                          int64_t n, void* stream);
int hm_interpolate(const double* x0, const double* s0, const double* x1, const double* s1,
                   double y0, double y1, double y, double* out, double* out_std, int64_t n, void* stream);
 /* hm_pair_statistics: ExposurePair.compute_difference + compute_stats(axis=(0,1)) fused
 * (modules/exposure_series.py:33-54, the per-pair body of process_linearity :443-446): the statistics of
 * the absolute and relative difference of two frames without writing the difference images.
 * out is 6*C float64 on the device: [abs mean | abs std | abs error | rel mean | rel std | rel error]. */
size_t hm_pair_statistics_workspace_bytes(void);
int hm_pair_statistics(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                       int64_t n, int C, double* out, void* workspace, void* stream);

/* ExposureSeries.process_linearity (modules/exposure_series.py:421-446) for ALL pairs of a stack at once: pair p compares
 * frame pair_i[p] (short) with frame pair_j[p] (long) scaled by multipliers[p] (their exposure ratio) and gets the six
 * statistics of hm_pair_statistics at out[p * 6C ...]. One launch per HM_PAIRS_MAX pairs: a workgroup owns a run of elements,
 * each of its waves one pair, so a frame is read from HBM once per launch instead of once per pair it takes part in.
 * vals / stds: [host] arrays of n_frames device pointers (stds NULL = unweighted); pair_i / pair_j / multipliers: [host].
 * lower / upper: [host] C limits each, or both NULL. When given, hm_apply_thresholds(lower, upper) is applied to EVERY frame (and its
 * std) IN PLACE before it is compared - the apply_thresholds loop of process_linearity (modules/exposure_series.py:437-441), which leaves
 * the series' image sets thresholded - fused into the first launch's loads where the frames are staged through LDS (no separate pass
 * over the frames), by thresholding kernels otherwise; the result is the same either way. */
#define HM_PAIRS_MAX 16
size_t hm_pairs_statistics_workspace_bytes(int n_pairs);
int hm_pairs_statistics(const double* const* vals, const double* const* stds /*nullable*/, int n_frames,
                        const int32_t* pair_i, const int32_t* pair_j, const double* multipliers, int n_pairs,
                        int64_t n, int C, const double* lower /*nullable*/, const double* upper /*nullable*/,
                        double* out /*n_pairs * 6C*/, void* workspace, void* stream);
/* hm_channel_histogram: compute_channel_histogram (modules/measurand.py:430-469) = np.histogram per channel on
 * [lo, hi] with `bins` equal-width bins (edges = np.linspace(lo, hi, bins + 1) on the device), non-finite values
 * skipped, optional weights 1/std with zero stds skipped. out is (C, bins) float64; channels not in
 * channel_mask stay 0. hm_channel_minmax gives the default range (min / max of the counted values). */
size_t hm_histogram_workspace_bytes(int bins, int C);
int hm_channel_minmax(const double* val, const double* std /*nullable*/, int64_t n, int C, double* out /*2*C*/,
                      void* workspace, void* stream);
int hm_channel_histogram(const double* val, const double* std /*nullable*/, int64_t n, int C, int channel_mask,
                         const double* edges, int bins, double lo, double hi, double* out, void* workspace, void* stream);
size_t hm_channel_statistics_workspace_bytes(void);
int hm_channel_statistics(const double* val, const double* std /*nullable*/, int64_t n, int C,
                          double* out, void* workspace, void* stream);
/* compute_dimension_statistics over ANY axis (modules/measurand.py:318-350 pass `axis` to NumPy's nan-reductions): the array is the
 * dense (outer, axis_len, inner) block and is reduced over its middle dimension; out_mean / out_std / out_err are outer * inner
 * float64 each (out_err nullable; NaN without std). Several reduced axes = their product when adjacent (two separate groups:
 * hm_axis_statistics2; three or more: the caller brings them together with a layout copy). workspace: hm_axis_statistics_workspace_bytes() bytes (0 = none needed). */
size_t hm_axis_statistics_workspace_bytes(int64_t outer, int64_t axis_len, int64_t inner);
int hm_axis_statistics(const double* val, const double* std /*nullable*/, int64_t outer, int64_t axis_len, int64_t inner,
                       double* out_mean, double* out_std, double* out_err /*nullable*/, void* workspace, void* stream);
/* The same over TWO separate groups of reduced axes (e.g. axis = (0, 2) of an H x W x C image: statistics per column over rows and
 * channels): the array is the dense (outer, a1, mid, a2, inner) block, reduced over a1 AND a2 in place - no layout copy; outputs are
 * outer * mid * inner float64 each. workspace: hm_axis_statistics2_workspace_bytes() bytes, always required on the device build. */
size_t hm_axis_statistics2_workspace_bytes(int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner);
int hm_axis_statistics2(const double* val, const double* std /*nullable*/, int64_t outer, int64_t a1, int64_t mid, int64_t a2, int64_t inner,
                        double* out_mean, double* out_std, double* out_err /*nullable*/, void* workspace, void* stream);
/* compute_difference / interpolate on operands that BROADCAST against each other (the reference applies NumPy broadcasting,
 * modules/measurand.py:621-681): `shape` is the broadcast result shape (ndim <= HM_MAX_DIMS), strides are in ELEMENTS with 0 on
 * broadcast axes (as hm_binary_op); an operand and its std share strides. Outputs are dense arrays of that shape. */
int hm_compute_difference_bcast(const double* x, const double* sx, const double* y, const double* sy, double multiplier,
                                double* out_abs, double* out_abs_std, double* out_rel, double* out_rel_std,
                                int ndim, const int64_t* shape /*[host]*/, const int64_t* strides_x /*[host]*/,
                                const int64_t* strides_y /*[host]*/, void* stream);
int hm_interpolate_bcast(const double* x0, const double* s0, const double* x1, const double* s1, double y0, double y1, double y,
                         double* out, double* out_std, int ndim, const int64_t* shape /*[host]*/, const int64_t* strides0 /*[host]*/,
                         const int64_t* strides1 /*[host]*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * Upstream producer: welford_algorithm, modules/video_processing.py:161-219.
 *   hm_welford_update    folds n_frames uint8 frames (each n_elems = H*W*C, HWC) into the running float64
 *                        mean / m2 (m2 nullable = use_std False), in frame order:
 *                          f = icrf[dn, c] (:200-201; icrf (256, C) float64, nullable) or dn / 255 (:203);
 *                          delta = f - mean; mean += delta / n; m2 += delta * (f - mean)   (:205-208)
 *                        count_before = frames already folded (the caller keeps the count). mean / m2 must be
 *                        zero-initialised by the caller before the first call (:181-184).
 *   hm_welford_finalize  mean frame = around(mean * 255) as uint8 (:210-211); std frame =
 *                        around(sqrt(m2 / (count - 1)) / sqrt(count)) as uint8 (:214-215, as written).
 *                        out_std needs count >= 2 (HM_EINVAL otherwise).
 *   hm_welford_algorithmic_bytes   n_elems * (n_frames + 16 or 32): each frame byte read once, the state
 *                        read and written once per launch.
 * ------------------------------------------------------------------------------------------ */
int hm_welford_update(const void* const* frames /*[host] n_frames device ptrs*/, int n_frames, int64_t count_before,
                      const double* icrf /*nullable*/, double* mean, double* m2 /*nullable*/,
                      int64_t n_elems, int C, void* stream);
int hm_welford_finalize(const double* mean, const double* m2 /*nullable*/, int64_t count,
                        uint8_t* out_mean /*nullable*/, uint8_t* out_std /*nullable*/, int64_t n_elems, void* stream);
int64_t hm_welford_algorithmic_bytes(int n_frames, int with_m2, int64_t n_elems);

/* ------------------------------------------------------------------------------------------
 * ICRF-calibration energy function: _energy_function + analyze_linearity,
 * modules/ICRF_calibration_exposure.py:66-145,148-201, for n_candidates ICRFs of one channel per launch.
 *   dn         (n_pixels, n_frames) uint8: the reference's (X, Y, N) channel stack (:257-277), frames ascending
 *              in exposure; std (same shape, float64) nullable = use_std False
 *   exposures  [host] n_frames exposure values (:282)
 *   icrf       (n_candidates, 256) float64 on the device: each row already shifted to ICRF[255] = 1, ICRF[0] = 0
 *              (:166-167); valid (n_candidates uint8, nullable): 0 marks a candidate the host rejected
 *              (range / monotonicity tests :173-179): its energy is +inf and no pixel work is done
 *   lower, upper   DN limits (:181-182): values outside [icrf[lower], icrf[upper]] are ignored (:96-97)
 *   use_relative   analyze_linearity's flag (the energy function passes True, :192-193)
 *   out_pairs  (n_candidates, n_frames (n_frames-1)/2) float64, nullable: the per-pair results in
 *              np.triu_indices(N, 1) order (:141-143); out_energy (n_candidates): nanmean over the pairs,
 *              +inf where that is NaN (:196-198)
 * ------------------------------------------------------------------------------------------ */
size_t hm_linearity_energy_workspace_bytes(int64_t n_pixels, int n_frames, int n_candidates);
int hm_linearity_energy(const uint8_t* dn, const double* std /*nullable*/, const double* exposures /*[host]*/,
                        const double* icrf, const uint8_t* valid /*nullable*/, int n_candidates,
                        int lower, int upper, int use_relative, int64_t n_pixels, int n_frames,
                        double* out_pairs /*nullable*/, double* out_energy, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-disk formats: host-side strip decoders for the TIFF files the reference exchanges with
 * OpenCV (modules/image_set.py:214-243 cv.imread, :264-363 cv.imwrite). HOST pointers, no device work, re-entrant.
 * Return the number of bytes written to dst, HM_EINVAL for a corrupt stream, HM_ESHAPE if dst_cap is too small.
 *   hm_tiff_lzw_decode       TIFF 6.0 LZW (Compression 5), MSB-first 9..12-bit codes, libtiff's early change
 *   hm_tiff_packbits_decode  PackBits (Compression 32773)
 * ------------------------------------------------------------------------------------------ */
int64_t hm_tiff_lzw_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
int64_t hm_tiff_packbits_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);

#ifdef __cplusplus
}
#endif
#endif /* HDRMERGE_H */
