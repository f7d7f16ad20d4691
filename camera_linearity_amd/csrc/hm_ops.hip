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

// Dense operands of one shape (what the reference's operators see on image stacks): two elements per lane, 16-byte nontemporal
// accesses, the presence of each operand's std a compile-time flag - no per-element index arithmetic, no load behind a branch.
// The general kernel above (one element per lane and step, strided offsets) measured 0.62-0.64 of 8 TB/s on six dense streams.
template <int OP, bool S1, bool S2>
__global__ __launch_bounds__(256) void k_binary_dense(const double* __restrict__ x1, const double* __restrict__ s1,
                                                      const double* __restrict__ x2, const double* __restrict__ s2,
                                                      double* __restrict__ out, double* __restrict__ out_std, int64_t n) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    constexpr bool STD = S1 || S2;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t pairs = n / 2;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < pairs; q += stride) {
        const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x1) + q);
        const f64x2 c = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x2) + q);
        f64x2 sa = {0.0, 0.0}, sc = {0.0, 0.0};
        if constexpr (S1) sa = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s1) + q);
        if constexpr (S2) sc = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s2) + q);
        f64x2 r, rs = {0.0, 0.0};
        double r0, r1, e0 = 0.0, e1 = 0.0;
        binary_eval<OP>(a.x, sa.x, c.x, sc.x, STD, r0, e0);
        binary_eval<OP>(a.y, sa.y, c.y, sc.y, STD, r1, e1);
        r.x = r0; r.y = r1; rs.x = e0; rs.y = e1;
        __builtin_nontemporal_store(r, reinterpret_cast<f64x2*>(out) + q);
        if constexpr (STD) __builtin_nontemporal_store(rs, reinterpret_cast<f64x2*>(out_std) + q);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double r, e = 0.0;
        binary_eval<OP>(x1[n - 1], S1 ? s1[n - 1] : 0.0, x2[n - 1], S2 ? s2[n - 1] : 0.0, STD, r, e);
        out[n - 1] = r;
        if constexpr (STD) out_std[n - 1] = e;
    }
}

// The same, in the access shape that measured fastest for plain copies on these boxes (tools/readbench.hip copysweep: 6.0-6.7 TB/s against
// 5.3-5.9 for one 16-byte access per lane and step): a wave owns chunks of 512 consecutive elements and issues the four 16-byte loads per
// lane of EVERY input stream back to back (4 KB per stream and wave), computes its 8 elements per lane, then issues the stores of every
// output stream back to back. Whole chunks only; the host sends the remainder through k_binary_dense.
constexpr int kBurst = 4;                                   // 16-byte accesses per lane, stream and chunk
constexpr int64_t kBurstChunk = 128 * kBurst;               // elements per wave chunk
template <int OP, bool S1, bool S2>
__global__ __launch_bounds__(256) void k_binary_burst(const double* __restrict__ x1, const double* __restrict__ s1,
                                                      const double* __restrict__ x2, const double* __restrict__ s2,
                                                      double* __restrict__ out, double* __restrict__ out_std, int64_t n_chunks) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    constexpr bool STD = S1 || S2;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cstride = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); c < n_chunks; c += cstride) {
        const int64_t b = c * kBurstChunk;
        f64x2 a[kBurst], d[kBurst], sa[kBurst], sd[kBurst];
#pragma unroll
        for (int k = 0; k < kBurst; ++k) a[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x1 + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < kBurst; ++k) d[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x2 + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < kBurst; ++k) {
            sa[k] = f64x2{0.0, 0.0};
            if constexpr (S1) sa[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s1 + b + 128 * k) + lane);
        }
#pragma unroll
        for (int k = 0; k < kBurst; ++k) {
            sd[k] = f64x2{0.0, 0.0};
            if constexpr (S2) sd[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s2 + b + 128 * k) + lane);
        }
        f64x2 r[kBurst], rs[kBurst];
#pragma unroll
        for (int k = 0; k < kBurst; ++k) {
            double r0, r1, e0 = 0.0, e1 = 0.0;
            binary_eval<OP>(a[k].x, sa[k].x, d[k].x, sd[k].x, STD, r0, e0);
            binary_eval<OP>(a[k].y, sa[k].y, d[k].y, sd[k].y, STD, r1, e1);
            r[k].x = r0; r[k].y = r1; rs[k].x = e0; rs[k].y = e1;
        }
#pragma unroll
        for (int k = 0; k < kBurst; ++k) __builtin_nontemporal_store(r[k], reinterpret_cast<f64x2*>(out + b + 128 * k) + lane);
        if constexpr (STD) {
#pragma unroll
            for (int k = 0; k < kBurst; ++k) __builtin_nontemporal_store(rs[k], reinterpret_cast<f64x2*>(out_std + b + 128 * k) + lane);
        }
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


// ------------------------------------------------------------------------------------------------
// x ** p for a plain scalar exponent p (no uncertainty of its own) - Measurand.__pow__, modules/measurand.py:217-241, as the
// merge loop uses it (S ** 2 at exposure_series.py:343, ** (1/2) at :394). The generic kernel above evaluates pow() twice and
// log() once per element (0.19 of the HBM roofline, profiles/r01e_bench_ops.json); with the exponent known on the host
//   value:  p == 2 -> x*x,  p == 0.5 -> sqrt(x),  p == 1 -> x,  small integer p -> repeated multiplication, else pow(x, p)
//   std  :  |dx**p/dx| * s = (p * x**(p-1)) * s, with x**(p-1) = x, 1/sqrt(x)... formed from the value; the exponent's term
//           (ln x * x**p * 0)**2 of :236-239 is +0 whenever ln x * x**p is finite and is evaluated as written otherwise
//           (x <= 0 or non-finite: NumPy gives NaN / inf there and so does this kernel).
// Results agree with NumPy's pow-based evaluation to a few ulp (tests hold 1e-13 against the reference's own outputs).
// Two elements per lane, 16-byte accesses.
// ------------------------------------------------------------------------------------------------
enum { POW_SQUARE = 0, POW_SQRT = 1, POW_ONE = 2, POW_INT = 3, POW_GENERAL = 4 };

template <int KIND>
__device__ __forceinline__ void pow_scalar_eval(double x, double s, double p, int ip, bool with_std, double& r, double& rs) {
    double d;                                               // x ** (p - 1)
    if (KIND == POW_SQUARE) { r = x * x; d = x; }
    else if (KIND == POW_SQRT) { r = sqrt(x); d = 1.0 / r; }
    else if (KIND == POW_ONE) { r = x; d = 1.0; }
    else if (KIND == POW_INT) {                             // |ip| <= 8: x**(|ip|-1) by multiplication, one more for the value
        const int n = ip < 0 ? -ip : ip;
        double acc = 1.0;
        for (int k = 1; k < n; ++k) acc *= x;
        if (ip > 0) { d = acc; r = acc * x; }
        else { r = 1.0 / (acc * x); d = r / x; }
    } else { r = pow(x, p); d = pow(x, p - 1.0); }
    if (with_std) {
        const double a = (p * d) * s;                        // measurand.py:236
        double b = 0.0;                                      // (log(x1) * x1**x2) * 0, :237-238
        const double lr = r;
        if (!(x > 0.0) || !(fabs(x) < __builtin_huge_val()) || !(fabs(lr) < __builtin_huge_val())) b = (log(x) * r) * 0.0;
        rs = sqrt(a * a + b * b);
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_pow_scalar(const double* __restrict__ x, const double* __restrict__ s, double* __restrict__ out,
                                                    double* __restrict__ out_s, int64_t n, double p, int ip) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const bool with_std = out_s != nullptr;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t pairs = n / 2;
    const bool vec = aligned_dev(x, 16) && aligned_dev(out, 16) && (!with_std || (aligned_dev(s, 16) && aligned_dev(out_s, 16)));
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < pairs; q += stride) {
        double x0, x1, s0 = 0.0, s1 = 0.0;
        if (vec) {
            const f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x) + q);
            x0 = v.x; x1 = v.y;
            if (with_std) { const f64x2 u = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s) + q); s0 = u.x; s1 = u.y; }
        } else {
            x0 = x[2 * q]; x1 = x[2 * q + 1];
            if (with_std) { s0 = s[2 * q]; s1 = s[2 * q + 1]; }
        }
        double r0, r1, e0 = 0.0, e1 = 0.0;
        pow_scalar_eval<KIND>(x0, s0, p, ip, with_std, r0, e0);
        pow_scalar_eval<KIND>(x1, s1, p, ip, with_std, r1, e1);
        if (vec) {
            f64x2 o; o.x = r0; o.y = r1;
            __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(out) + q);
            if (with_std) { f64x2 e; e.x = e0; e.y = e1; __builtin_nontemporal_store(e, reinterpret_cast<f64x2*>(out_s) + q); }
        } else {
            out[2 * q] = r0; out[2 * q + 1] = r1;
            if (with_std) { out_s[2 * q] = e0; out_s[2 * q + 1] = e1; }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double r, e = 0.0;
        pow_scalar_eval<KIND>(x[n - 1], with_std ? s[n - 1] : 0.0, p, ip, with_std, r, e);
        out[n - 1] = r;
        if (with_std) out_s[n - 1] = e;
    }
}

// pow with a scalar exponent in the burst access shape (k_binary_burst): whole 512-element chunks, 16-byte aligned buffers
template <int KIND, bool STD>
__global__ __launch_bounds__(256) void k_pow_scalar_burst(const double* __restrict__ x, const double* __restrict__ s, double* __restrict__ out,
                                                          double* __restrict__ out_s, int64_t n_chunks, double p, int ip) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t cstride = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); c < n_chunks; c += cstride) {
        const int64_t b = c * kBurstChunk;
        f64x2 xv[kBurst], sv[kBurst];
#pragma unroll
        for (int k = 0; k < kBurst; ++k) xv[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(x + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < kBurst; ++k) {
            sv[k] = f64x2{0.0, 0.0};
            if constexpr (STD) sv[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(s + b + 128 * k) + lane);
        }
        f64x2 r[kBurst], e[kBurst];
#pragma unroll
        for (int k = 0; k < kBurst; ++k) {
            double r0, r1, e0 = 0.0, e1 = 0.0;
            pow_scalar_eval<KIND>(xv[k].x, sv[k].x, p, ip, STD, r0, e0);
            pow_scalar_eval<KIND>(xv[k].y, sv[k].y, p, ip, STD, r1, e1);
            r[k] = f64x2{r0, r1}; e[k] = f64x2{e0, e1};
        }
#pragma unroll
        for (int k = 0; k < kBurst; ++k) __builtin_nontemporal_store(r[k], reinterpret_cast<f64x2*>(out + b + 128 * k) + lane);
        if constexpr (STD) {
#pragma unroll
            for (int k = 0; k < kBurst; ++k) __builtin_nontemporal_store(e[k], reinterpret_cast<f64x2*>(out_s + b + 128 * k) + lane);
        }
    }
}

// take along one axis: out[o, k, i] = in[o, idx[k], i] for the (outer, axis_len, inner) view of a dense array
// (modules/measurand.py:352-373, lib.take(val, dims, axis)). A pure gather copy: val and std in one launch.
struct TakeK {
    int64_t idx[HM_TAKE_MAX];
    int n_idx;
};
__global__ __launch_bounds__(256) void k_take(const double* __restrict__ x, const double* __restrict__ s, double* __restrict__ out,
                                              double* __restrict__ out_s, int64_t outer, int64_t axis_len, int64_t inner, const TakeK t) {
    const int64_t n = outer * t.n_idx * inner;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t i = e % inner;
        const int64_t r = e / inner;
        const int64_t kx = r % t.n_idx;
        const int64_t o = r / t.n_idx;
        int64_t src_k = t.idx[0];
        for (int q = 1; q < t.n_idx; ++q) src_k = kx == q ? t.idx[q] : src_k;       // uniform indexing of the kernarg array
        const int64_t src = (o * axis_len + src_k) * inner + i;
        out[e] = x[src];
        if (out_s) out_s[e] = s[src];
    }
}

}  // namespace hm

using namespace hm;

extern "C" int hm_take_axis(const double* x, const double* s, double* out_val, double* out_std, int64_t outer, int64_t axis_len,
                            int64_t inner, const int64_t* indices, int n_indices, void* stream) {
    if (!x || !out_val || !indices || outer < 0 || axis_len < 1 || inner < 1 || n_indices < 1) return HM_EINVAL;
    if (n_indices > HM_TAKE_MAX) return HM_EUNSUPPORTED;
    if ((out_std != nullptr) != (s != nullptr)) return HM_EINVAL;
    TakeK t{};
    t.n_idx = n_indices;
    for (int q = 0; q < n_indices; ++q) {
        int64_t v = indices[q];
        if (v < 0) v += axis_len;                          // NumPy's negative indices
        if (v < 0 || v >= axis_len) return HM_ESHAPE;      // np.take raises IndexError (mode='raise')
        t.idx[q] = v;
    }
    const int64_t n = outer * n_indices * inner;
    if (n == 0) return HM_OK;
    hipLaunchKernelGGL(k_take, dim3(stream_grid(n, 256, 8)), dim3(256), 0, as_stream(stream), x, s, out_val, out_std, outer, axis_len, inner, t);
    return launch_status();
}

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
    hipStream_t st = as_stream(stream);
    const bool al16 = aligned(x1, 16) && aligned(x2, 16) && aligned(out_val, 16) && (!s1 || aligned(s1, 16)) && (!s2 || aligned(s2, 16)) &&
                      (!out_std || aligned(out_std, 16));
    if (dense && al16 && op != HM_OP_POW) {
#ifndef HM_NO_BURST
        const int64_t n_chunks = n / kBurstChunk;
        if (n_chunks > 0) {
            const unsigned bgrid = stream_grid((n_chunks + 3) / 4, 1, 16);
#define HM_BINB(O, A, B) hipLaunchKernelGGL((k_binary_burst<O, A, B>), dim3(bgrid), dim3(256), 0, st, x1, s1, x2, s2, out_val, out_std, n_chunks)
#define HM_BINBS(O) do { if (s1 && s2) HM_BINB(O, true, true); else if (s1) HM_BINB(O, true, false); else if (s2) HM_BINB(O, false, true); \
                         else HM_BINB(O, false, false); } while (0)
            switch (op) {
                case HM_OP_ADD: HM_BINBS(HM_OP_ADD); break;
                case HM_OP_SUB: HM_BINBS(HM_OP_SUB); break;
                case HM_OP_MUL: HM_BINBS(HM_OP_MUL); break;
                default:        HM_BINBS(HM_OP_DIV); break;
            }
#undef HM_BINBS
#undef HM_BINB
            const int64_t done = n_chunks * kBurstChunk;                 // the remainder (< 512 elements) through the per-lane kernel below
            if (done == n) return launch_status();
            x1 += done; x2 += done; out_val += done; n -= done;
            if (s1) s1 += done;
            if (s2) s2 += done;
            if (out_std) out_std += done;
        }
#endif
        const unsigned dgrid = stream_grid((n + 1) / 2, 256, 8);
#define HM_BIND(O, A, B) hipLaunchKernelGGL((k_binary_dense<O, A, B>), dim3(dgrid), dim3(256), 0, st, x1, s1, x2, s2, out_val, out_std, n)
#define HM_BINS(O) do { if (s1 && s2) HM_BIND(O, true, true); else if (s1) HM_BIND(O, true, false); else if (s2) HM_BIND(O, false, true); \
                        else HM_BIND(O, false, false); } while (0)
        switch (op) {
            case HM_OP_ADD: HM_BINS(HM_OP_ADD); break;
            case HM_OP_SUB: HM_BINS(HM_OP_SUB); break;
            case HM_OP_MUL: HM_BINS(HM_OP_MUL); break;
            default:        HM_BINS(HM_OP_DIV); break;
        }
#undef HM_BINS
#undef HM_BIND
        return launch_status();
    }
    const unsigned grid = stream_grid(n, 256, 8);
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

extern "C" int hm_pow_scalar(const double* x, const double* s, double exponent, double* out_val, double* out_std, int64_t n, void* stream) {
    if (n < 0) return HM_EINVAL;
    if (n == 0) return HM_OK;
    if (!x || !out_val || ((out_std != nullptr) != (s != nullptr))) return HM_EINVAL;
    if (!aligned(x, 8) || !aligned(out_val, 8)) return HM_EALIGN;
    hipStream_t st = as_stream(stream);
    const double p = exponent;
    const bool small = p >= -8.0 && p <= 8.0;                       // range first: double -> int is undefined outside int's range and for NaN
    const int ip = (small && p == static_cast<double>(static_cast<int>(p))) ? static_cast<int>(p) : 0;
    // whole 512-element chunks of 16-byte aligned buffers in the burst access shape (the cheap exponents: a general pow() is compute-bound)
    const bool al16 = aligned(x, 16) && aligned(out_val, 16) && (!s || (aligned(s, 16) && aligned(out_std, 16)));
    const int64_t n_chunks = n / kBurstChunk;
    if (al16 && n_chunks > 0 && (p == 2.0 || p == 0.5 || p == 1.0 || ip != 0)) {
        const unsigned bgrid = stream_grid((n_chunks + 3) / 4, 1, 16);
#define HM_POWB(K) do { if (s) hipLaunchKernelGGL((k_pow_scalar_burst<K, true>), dim3(bgrid), dim3(256), 0, st, x, s, out_val, out_std, n_chunks, p, ip); \
                        else hipLaunchKernelGGL((k_pow_scalar_burst<K, false>), dim3(bgrid), dim3(256), 0, st, x, s, out_val, out_std, n_chunks, p, ip); } while (0)
        if (p == 2.0) HM_POWB(POW_SQUARE);
        else if (p == 0.5) HM_POWB(POW_SQRT);
        else if (p == 1.0) HM_POWB(POW_ONE);
        else HM_POWB(POW_INT);
#undef HM_POWB
        const int64_t done = n_chunks * kBurstChunk;
        if (done == n) return launch_status();
        x += done; out_val += done; n -= done;
        if (s) { s += done; out_std += done; }
    }
    const unsigned grid = stream_grid((n + 1) / 2, 256, 8);
#define HM_POW(K) hipLaunchKernelGGL(k_pow_scalar<K>, dim3(grid), dim3(256), 0, st, x, s, out_val, out_std, n, p, ip)
    if (p == 2.0) HM_POW(POW_SQUARE);
    else if (p == 0.5) HM_POW(POW_SQRT);
    else if (p == 1.0) HM_POW(POW_ONE);
    else if (ip != 0) HM_POW(POW_INT);
    else HM_POW(POW_GENERAL);
#undef HM_POW
    return launch_status();
}
