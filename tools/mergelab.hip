// mergelab.hip - kernel laboratory for the val-only fused merge (config 2) on MI355X.
// Standalone (no torch): compiles in seconds, runs in seconds. Each variant is the production kernel's
// structure with one knob changed; probe variants (marked *) compute wrong results on purpose and only
// price a component. Non-probe variants are verified against a host double-precision evaluation of
// the same formula on a sample of elements.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/mergelab.hip -o tools/bin/mergelab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <functional>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define PIN(x) asm volatile("" : "+v"(x))

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int NF = 7;
struct Args {
    const uint8_t* frame[NF];
    double inv_t[NF];
    const double* w_lut;     // 256
    const double* icrf;      // 256*3
    double* out;
    uint32_t n_elems;
};

enum { T_FUSED = 0, T_NONE = 1, T_SPLIT = 2, T_WPRIV = 3, T_BOTH = 4 };   // T_BOTH: w R=32 private + wg R=16 (160 KB)
enum { D_IEEE = 0, D_RCP = 1, D_NONE = 2, D_BATCH = 3 };   // D_BATCH: one division per 4 elements (Montgomery batch inverse) + one fma correction step
enum { TR_LDS = 0, TR_FAKE = 1 };

template <int TAB> constexpr int tab_bytes() {
    return TAB == T_FUSED ? 16 * 768 : TAB == T_SPLIT ? 8 * 1024 : TAB == T_WPRIV ? 256 * 256 + 8 * 768 : TAB == T_BOTH ? 256 * 256 + 128 * 768 : 0;
}

// acc / S without the scaling / fixup steps of the IEEE sequence: valid for normal, positive operands
// (S >= N * e^-7.5 > 0, |acc| far from overflow). rcp + 2 Newton steps + 1 correction of the quotient.
__device__ __forceinline__ double div_rcp(double a, double s) {
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-s, q, a), r, q);
}

// EPL = elements per lane per sub-unit (4: dword loads, 8: dwordx2 loads, 12: dwordx3 loads)
template <int EPL, int U, int TAB, int DIV, int TR, bool PF, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_merge(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RW = EPL >= 4 ? EPL / 4 : 1;
    constexpr uint32_t SUB = 64 * EPL;           // elements per sub-unit
    constexpr uint32_t GROUP = U * SUB;
    // ---- tables
    for (int q = threadIdx.x; q < 768; q += BLOCK) {
        const int dn = q / 3;
        const double w = a.w_lut[dn], wg = w * a.icrf[q];
        if constexpr (TAB == T_FUSED) { reinterpret_cast<double2*>(lds)[q] = double2{w, wg}; }
        if constexpr (TAB == T_SPLIT) { double* t = reinterpret_cast<double*>(lds); if (q % 3 == 0) t[dn] = w; t[256 + q] = wg; }
        if constexpr (TAB == T_WPRIV) {
            double* tw = reinterpret_cast<double*>(lds);
            double* tg = reinterpret_cast<double*>(lds + 256 * 256);
            if (q % 3 == 0) for (int r = 0; r < 32; ++r) tw[dn * 32 + r] = w;
            tg[q] = wg;
        }
        if constexpr (TAB == T_BOTH) {
            double* tw = reinterpret_cast<double*>(lds);
            double* tg = reinterpret_cast<double*>(lds + 256 * 256);
            if (q % 3 == 0) for (int r = 0; r < 32; ++r) tw[dn * 32 + r] = w;
            for (int r = 0; r < 16; ++r) tg[q * 16 + r] = wg;
        }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t WPB = BLOCK / 64;
    char* slab = lds + tab_bytes<TAB>() + wave * (SUB * 8);
    const uint32_t n_groups = a.n_elems / GROUP;
    const uint32_t gstride = gridDim.x * WPB;
    uint32_t g = blockIdx.x * WPB + wave;
    uint32_t raw[NF][U][RW];
    auto load_group = [&](uint32_t grp, uint32_t (&dst)[NF][U][RW]) {
        const size_t off = static_cast<size_t>(grp) * GROUP;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint8_t* p = a.frame[i] + off + lane * EPL;
#pragma unroll
            for (int s = 0; s < U; ++s) {
                if constexpr (EPL == 2) dst[i][s][0] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p + SUB * s));
                if constexpr (EPL == 4) dst[i][s][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p + SUB * s));
                if constexpr (EPL == 8) { u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p + SUB * s)); dst[i][s][0] = v.x; dst[i][s][1] = v.y; }
                if constexpr (EPL == 12) { u32x3 v = __builtin_nontemporal_load(reinterpret_cast<const u32x3*>(p + SUB * s)); dst[i][s][0] = v.x; dst[i][s][1] = v.y; dst[i][s][2] = v.z; }
            }
        }
    };
    if (PF && g < n_groups) load_group(g, raw);
    for (; g < n_groups; g += gstride) {
        uint32_t cur[NF][U][RW];
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int s = 0; s < U; ++s)
#pragma unroll
                    for (int k = 0; k < RW; ++k) cur[i][s][k] = raw[i][s][k];
            if (g + gstride < n_groups) load_group(g + gstride, raw);
        } else {
            load_group(g, cur);
        }
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const size_t sbase = static_cast<size_t>(g) * GROUP + SUB * s;
            // channel of the lane's element j: (sbase + EPL*lane + j) % 3; SUB % 3 == EPL*64 % 3
            const uint32_t c0 = (EPL % 3 == 0) ? 0u : ((g * U + s) * (SUB % 3) + lane * (EPL % 3)) % 3u;
            uint32_t coffs[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t c = (c0 + k) % 3u;
                coffs[k] = TAB == T_FUSED ? c * 16u : TAB == T_SPLIT ? 2048u + c * 8u : TAB == T_WPRIV ? 65536u + c * 8u : TAB == T_BOTH ? 65536u + c * 128u + (lane & 15u) * 8u : c;
            }
            const uint32_t woff = (TAB == T_WPRIV || TAB == T_BOTH) ? (lane & 31u) * 8u : 0u;
            double S[EPL], acc[EPL];
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const double it = a.inv_t[i];
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const uint32_t dn = (cur[i][s][j >> 2] >> (8 * (j & 3))) & 255u;
                    double w, wg;
                    if constexpr (TAB == T_FUSED) { const double2 t = *reinterpret_cast<const double2*>(lds + dn * 48u + coffs[j % 3]); w = t.x; wg = t.y; }
                    if constexpr (TAB == T_SPLIT) { w = *reinterpret_cast<const double*>(lds + dn * 8u); wg = *reinterpret_cast<const double*>(lds + dn * 24u + coffs[j % 3]); }
                    if constexpr (TAB == T_WPRIV) { w = *reinterpret_cast<const double*>(lds + dn * 256u + woff); wg = *reinterpret_cast<const double*>(lds + dn * 24u + coffs[j % 3]); }
                    if constexpr (TAB == T_BOTH) { w = *reinterpret_cast<const double*>(lds + dn * 256u + woff); wg = *reinterpret_cast<const double*>(lds + dn * 384u + coffs[j % 3]); }
                    if constexpr (TAB == T_NONE) { w = static_cast<double>(dn + 1u); wg = static_cast<double>(dn + coffs[j % 3]); }
                    if (i == 0) { S[j] = w; acc[j] = wg * it; } else { S[j] += w; acc[j] = fma(wg, it, acc[j]); }
                }
                if ((i & 1) == 1 || i == NF - 1) {
#pragma unroll
                    for (int j = 0; j < EPL; ++j) { PIN(S[j]); PIN(acc[j]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            double val[EPL];
            if constexpr (DIV == D_BATCH) {
#pragma unroll
                for (int j = 0; j < EPL; j += 4) {
                    const double p01 = S[j] * S[j + 1], p23 = S[j + 2] * S[j + 3];
                    const double r = 1.0 / (p01 * p23);
                    const double r01 = r * p23, r23 = r * p01;
                    const double inv[4] = {r01 * S[j + 1], r01 * S[j], r23 * S[j + 3], r23 * S[j + 2]};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double q = acc[j + k] * inv[k];
                        val[j + k] = fma(fma(-S[j + k], q, acc[j + k]), inv[k], q);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if constexpr (DIV == D_IEEE) val[j] = acc[j] / S[j];
                if constexpr (DIV == D_RCP) val[j] = div_rcp(acc[j], S[j]);
                if constexpr (DIV == D_NONE) val[j] = acc[j] + S[j];
            }
            double* o = a.out + sbase;
            if constexpr (EPL == 2) {
                f64x2 v; v.x = val[0]; v.y = val[1];
                __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(o + 2 * lane));
            } else if constexpr (TR == TR_LDS) {
#pragma unroll
                for (int j = 0; j < EPL; j += 2) { f64x2 v; v.x = val[j]; v.y = val[j + 1]; *reinterpret_cast<f64x2*>(slab + lane * (EPL * 8) + j * 8) = v; }
                __builtin_amdgcn_wave_barrier();
                f64x2 r[EPL / 2];
#pragma unroll
                for (int k = 0; k < EPL / 2; ++k) r[k] = *reinterpret_cast<const f64x2*>(slab + k * 1024 + lane * 16);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < EPL / 2; ++k) __builtin_nontemporal_store(r[k], reinterpret_cast<f64x2*>(o + k * 128 + 2 * lane));
            } else {
#pragma unroll
                for (int k = 0; k < EPL / 2; ++k) { f64x2 v; v.x = val[2 * k]; v.y = val[2 * k + 1]; __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(o + k * 128 + 2 * lane)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <typename F>
static double time_sustained(F&& launch, int iters = 100) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

static std::vector<std::vector<uint8_t>> h_frames;
static std::vector<double> h_w, h_icrf, h_invt;
static double* d_out;
static Args g_args;
static size_t E;

static double verify(size_t step = 4099) {
    std::vector<double> out(E);
    CK(hipMemcpy(out.data(), d_out, E * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t e = 0; e < E; e += step) {
        const int c = e % 3;
        double S = 0, acc = 0;
        for (int i = 0; i < NF; ++i) {
            const int dn = h_frames[i][e];
            const double w = h_w[dn], wg = w * h_icrf[dn * 3 + c];
            S = i == 0 ? w : S + w;
            acc = i == 0 ? wg * h_invt[i] : std::fma(wg, h_invt[i], acc);
        }
        const double ref = acc / S;
        worst = std::max(worst, std::fabs(out[e] - ref) / std::fabs(ref));
    }
    return worst;
}

struct Variant { std::string name; std::function<void()> launch; bool check; double err; std::vector<double> us; int lds, per_cu; };
static std::vector<Variant> g_variants;

template <int EPL, int U, int TAB, int DIV, int TR, bool PF, int BLOCK>
static void run(const char* name, bool check) {
    constexpr int lds = tab_bytes<TAB>() + (EPL == 2 ? 0 : (BLOCK / 64) * 64 * EPL * 8);
    auto kernel = k_merge<EPL, U, TAB, DIV, TR, PF, BLOCK>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int per_cu = 2048 / BLOCK;
    if (lds > 0 && 160 * 1024 / lds < per_cu) per_cu = 160 * 1024 / lds;
    const uint32_t groups = E / (U * 64 * EPL);
    uint32_t grid = std::min<uint32_t>((groups + BLOCK / 64 - 1) / (BLOCK / 64), 256 * per_cu);
    CK(hipMemset(d_out, 0, E * 8));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, 0, g_args);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    Variant v{name, [=] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, 0, g_args); }, check, check ? verify() : -1.0, {}, lds, per_cu};
    g_variants.push_back(v);
}

static void run_rounds(int rounds, int iters) {
    // pre-warm: ~1 s of back-to-back launches so the chip is at its sustained clock before anything is timed
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 6000; ++i) g_variants[0].launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto& v : g_variants) {
            for (int i = 0; i < 5; ++i) v.launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v.launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(ms * 1e3 / iters);
        }
    const double bytes = double(E) * 15;
    for (auto& v : g_variants) {
        std::sort(v.us.begin(), v.us.end());
        const double med = v.us[v.us.size() / 2];
        printf("%-52s med %6.1f  min %6.1f  max %6.1f us  %6.1f GB/s  %.3f of 8TB/s  lds %6d x%d %s", v.name.c_str(), med, v.us.front(), v.us.back(),
               bytes / med / 1e3, bytes / med / 1e3 / 8000, v.lds, v.per_cu, v.check ? "" : "*probe");
        if (v.check) printf(" err %.1e%s", v.err, v.err < 1e-13 ? "" : " <-- MISMATCH");
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const size_t H = 4096, W = 4096;
    E = H * W * 3;
    const bool uniform = argc > 1 && atoi(argv[1]) == 1;
    // synthetic stack of SURVEY 8(d): rad in [0,4), t_i = 1ms * 2^i, k = 255 / (4 t_3)
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> uni(0.0, 4.0);
    h_frames.assign(NF, std::vector<uint8_t>(E));
    h_invt.resize(NF);
    std::vector<double> t(NF);
    for (int i = 0; i < NF; ++i) { t[i] = 1e-3 * std::ldexp(1.0, i); h_invt[i] = 1.0 / t[i]; }
    const double k = 255.0 / (4.0 * t[NF / 2]);
    for (size_t e = 0; e < E; ++e) {
        const double rad = uni(rng);
        for (int i = 0; i < NF; ++i) {
            double v = uniform ? double(rng() & 255) : std::nearbyint(rad * t[i] * k);
            h_frames[i][e] = static_cast<uint8_t>(std::min(255.0, std::max(0.0, v)));
        }
    }
    h_w.resize(256); h_icrf.resize(768);
    const double gam[3] = {2.2, 2.0, 1.8};
    for (int d = 0; d < 256; ++d) {
        const double v = d / 255.0;
        h_w[d] = std::pow(M_E, -30.0 * (v - 0.5) * (v - 0.5));
        for (int c = 0; c < 3; ++c) h_icrf[d * 3 + c] = std::pow(v, gam[c]);
    }
    for (int i = 0; i < NF; ++i) {
        void* p; CK(hipMalloc(&p, E)); CK(hipMemcpy(p, h_frames[i].data(), E, hipMemcpyHostToDevice));
        g_args.frame[i] = static_cast<const uint8_t*>(p); g_args.inv_t[i] = h_invt[i];
    }
    void *pw, *pi; CK(hipMalloc(&pw, 2048)); CK(hipMalloc(&pi, 6144));
    CK(hipMemcpy(pw, h_w.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(pi, h_icrf.data(), 6144, hipMemcpyHostToDevice));
    g_args.w_lut = static_cast<const double*>(pw); g_args.icrf = static_cast<const double*>(pi);
    CK(hipMalloc(reinterpret_cast<void**>(&d_out), E * 8)); g_args.out = d_out; g_args.n_elems = static_cast<uint32_t>(E);
    printf("mergelab: %s DNs, 7x4096x4096x3, algorithmic bytes %.1f MB\n", uniform ? "uniform random" : "radiance-stack", E * 15 / 1e6);
    //           EPL U  TAB      DIV     TR      PF   BLOCK
    run<4, 2, T_FUSED, D_IEEE, TR_LDS, false, 256>("dword  U2 fused ieee   [production]", true);
    run<4, 2, T_FUSED, D_BATCH, TR_LDS, false, 256>("dword  U2 fused batch-div", true);
    run<2, 2, T_FUSED, D_IEEE, TR_LDS, false, 256>("ushort U2 fused ieee direct", true);
    run<2, 4, T_FUSED, D_IEEE, TR_LDS, false, 256>("ushort U4 fused ieee direct", true);
    run<2, 2, T_BOTH, D_IEEE, TR_LDS, false, 1024>("ushort U2 BOTH ieee blk1024", true);
    run<2, 4, T_BOTH, D_IEEE, TR_LDS, false, 1024>("ushort U4 BOTH ieee blk1024", true);
    run<2, 4, T_BOTH, D_IEEE, TR_LDS, true, 1024>("ushort U4 BOTH ieee blk1024 pf", true);
    run<2, 8, T_BOTH, D_IEEE, TR_LDS, false, 1024>("ushort U8 BOTH ieee blk1024", true);
    run<4, 1, T_WPRIV, D_IEEE, TR_LDS, false, 1024>("dword  U1 wpriv ieee blk1024", true);
    run<4, 2, T_WPRIV, D_BATCH, TR_LDS, false, 1024>("dword  U2 wpriv batch blk1024", true);
    run<4, 2, T_WPRIV, D_BATCH, TR_LDS, true, 1024>("dword  U2 wpriv batch blk1024 pf", true);
    run<4, 4, T_WPRIV, D_BATCH, TR_LDS, false, 1024>("dword  U4 wpriv batch blk1024", true);
    run<4, 2, T_NONE, D_NONE, TR_FAKE, false, 1024>("dword  U2 NONE no-div fake blk1024", false);
    run<4, 2, T_NONE, D_NONE, TR_FAKE, false, 256>("dword  U2 NONE no-div fake blk256", false);
    run_rounds(argc > 2 ? atoi(argv[2]) : 9, 40);
    return 0;
}
