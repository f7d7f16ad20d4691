// hm_ops.hip - Measurand operators with first-order uncertainty propagation (gfx950).
// Formulas and operation order of modules/measurand.py:106-279 (add/sub/neg/div/mul/pow/log_e/log_10);
// operands broadcast NumPy-style through element strides (0 on broadcast axes).
#include "hm_common.h"

namespace hm {

struct BcastK {
    int64_t shape[HM_MAX_DIMS];
    int64_t st1[HM_MAX_DIMS];
    int64_t st2[HM_MAX_DIMS];
    int ndim;
    int contiguous;     // both operands dense and same shape: offsets == linear index
};

template <int OP>
__device__ __forceinline__ void binary_eval(double x1, double s1, double x2, double s2, bool with_std,
                                            double& r, double& rs) {
    if (OP == HM_OP_ADD) {                                  // measurand.py:114,126
        r = x1 + x2;
        if (with_std) rs = sqrt((s1 * s1) + (s2 * s2));
    } else if (OP == HM_OP_SUB) {                           // :138,149
        r = x1 - x2;
        if (with_std) rs = sqrt((s1 * s1) + (s2 * s2));
    } else if (OP == HM_OP_MUL) {                           // :198,209
        r = x1 * x2;
        if (with_std) { const double a = x1 * s2, b = x2 * s1; rs = sqrt(a * a + b * b); }
    } else if (OP == HM_OP_DIV) {                           // :173,184-186
        r = x1 / x2;
        if (with_std) { const double u1 = s1 / x2, u2 = (x1 * s2) / (x2 * x2); rs = sqrt(u1 * u1 + u2 * u2); }
    } else {                                                // pow :225,236-239
        r = pow(x1, x2);
        if (with_std) {
            const double u1 = x2 * pow(x1, x2 - 1.0);
            const double u2 = log(x1) * r;
            const double a = u1 * s1, b = u2 * s2;
            rs = sqrt(a * a + b * b);
        }
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_binary(const double* __restrict__ x1, const double* __restrict__ s1,
                                                const double* __restrict__ x2, const double* __restrict__ s2,
                                                double* __restrict__ out, double* __restrict__ out_std,
                                                int64_t n, const BcastK b) {
    const bool with_std = out_std != nullptr;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        int64_t o1 = e, o2 = e;
        if (!b.contiguous) {
            o1 = 0; o2 = 0;
            int64_t rem = e;
            for (int d = b.ndim - 1; d >= 0; --d) {
                const int64_t i = rem % b.shape[d];
                rem /= b.shape[d];
                o1 += i * b.st1[d];
                o2 += i * b.st2[d];
            }
        }
        const double a = x1[o1], c = x2[o2];
        const double sa = (with_std && s1) ? s1[o1] : 0.0;      // missing std -> zeros (:121-124)
        const double sc = (with_std && s2) ? s2[o2] : 0.0;
        double r, rs = 0.0;
        binary_eval<OP>(a, sa, c, sc, with_std, r, rs);
        out[e] = r;
        if (with_std) out_std[e] = rs;
    }
}

__global__ __launch_bounds__(256) void k_unary(int op, const double* __restrict__ x, const double* __restrict__ s,
                                               double* __restrict__ out, double* __restrict__ out_std, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const double ln10 = log(5.0) + log(2.0);                   // measurand.py:277
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const double v = x[e];
        double r, rs = 0.0;
        if (op == HM_UOP_NEG) { r = -v; if (out_std) rs = s[e]; }                              // :154-157
        else if (op == HM_UOP_LOG_E) { r = log(v); if (out_std) rs = s[e] / r; }               // :251,258 (as written)
        else { r = log10(v); if (out_std) rs = s[e] / (v * ln10); }                            // :270,277
        out[e] = r;
        if (out_std) out_std[e] = rs;
    }
}

}  // namespace hm

using namespace hm;

extern "C" int hm_binary_op(int op, const double* x1, const double* s1, const double* x2, const double* s2,
                            double* out_val, double* out_std, int ndim, const int64_t* shape,
                            const int64_t* strides1, const int64_t* strides2, void* stream) {
    if (op < HM_OP_ADD || op > HM_OP_POW || ndim < 1 || ndim > HM_MAX_DIMS || !shape || !strides1 || !strides2)
        return HM_EINVAL;
    if (!x1 || !x2 || !out_val) return HM_EINVAL;
    if ((out_std != nullptr) != (s1 != nullptr || s2 != nullptr)) return HM_EINVAL;
    if (!aligned(x1, 8) || !aligned(x2, 8) || !aligned(out_val, 8)) return HM_EALIGN;
    BcastK b{};
    b.ndim = ndim;
    int64_t n = 1;
    bool dense = true;
    for (int d = ndim - 1; d >= 0; --d) {
        if (shape[d] < 0) return HM_EINVAL;
        b.shape[d] = shape[d]; b.st1[d] = strides1[d]; b.st2[d] = strides2[d];
        if (shape[d] != 1 && (strides1[d] != n || strides2[d] != n)) dense = false;
        n *= shape[d];
    }
    b.contiguous = dense ? 1 : 0;
    if (n == 0) return HM_OK;
    const unsigned grid = stream_grid(n, 256, 8);
    hipStream_t st = as_stream(stream);
#define HM_BIN(O) hipLaunchKernelGGL(k_binary<O>, dim3(grid), dim3(256), 0, st, x1, s1, x2, s2, out_val, out_std, n, b)
    switch (op) {
        case HM_OP_ADD: HM_BIN(HM_OP_ADD); break;
        case HM_OP_SUB: HM_BIN(HM_OP_SUB); break;
        case HM_OP_MUL: HM_BIN(HM_OP_MUL); break;
        case HM_OP_DIV: HM_BIN(HM_OP_DIV); break;
        default:        HM_BIN(HM_OP_POW); break;
    }
#undef HM_BIN
    return launch_status();
}

extern "C" int hm_unary_op(int op, const double* x, const double* s, double* out_val, double* out_std,
                           int64_t n, void* stream) {
    if (op < HM_UOP_NEG || op > HM_UOP_LOG_10 || n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !out_val || ((out_std != nullptr) != (s != nullptr))) return HM_EINVAL;
    if (!aligned(x, 8) || !aligned(out_val, 8)) return HM_EALIGN;
    hipLaunchKernelGGL(k_unary, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream), op, x, s, out_val, out_std, n);
    return launch_status();
}
