/* merge_c_abi.c - libhdrmerge.so (or, with -DHM_HOST_BUILD, libhdrmerge_host.so) from plain C: no Python, no torch, only the HIP
 * runtime for device memory - and not even that for the host build, whose entry points take host pointers.
 *
 * Merges a small synthetic exposure stack (uint8 frames, val + std) with hm_merge and checks the result against a
 * straightforward host loop over the formulas of modules/exposure_series.py:340,388-389,394 (this file's own few
 * lines of C, not the oracle). Build and run (tests/test_gpu_api.py::test_c_abi_example_from_plain_c does both):
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ examples/merge_c_abi.c -Iinclude -I/opt/rocm/include \
 *       -Lcamera_linearity_amd/lib -lhdrmerge -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/camera_linearity_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/merge_c_abi && /tmp/merge_c_abi
 *
 * Host build of the same ABI (tests/test_host_backend.py::test_c_abi_example_host_build; runs without a GPU):
 *
 *   gcc -std=c11 -O2 -DHM_HOST_BUILD examples/merge_c_abi.c -Iinclude -Lcamera_linearity_amd/lib -lhdrmerge_host \
 *       -Wl,-rpath,$PWD/camera_linearity_amd/lib -lm -o /tmp/merge_c_abi_host && /tmp/merge_c_abi_host
 */
#ifndef HM_HOST_BUILD
#include <hip/hip_runtime_api.h>
#else                                   /* host build: "device" memory is host memory, the stream is NULL */
#include <stdlib.h>
#include <string.h>
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static const char* hipGetErrorString(hipError_t e) { (void)e; return "out of memory"; }
static hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
static hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) { (void)kind; memcpy(d, s, n); return hipSuccess; }
static hipError_t hipStreamCreate(hipStream_t* s) { *s = NULL; return hipSuccess; }
static hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }
#endif
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hdrmerge.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HM(x) do { int r_ = (x); if (r_ != HM_OK) { fprintf(stderr, "%s: %s\n", #x, hm_strerror(r_)); return 3; } } while (0)

enum { N = 5, H = 37, W = 53, C = 3 };

int main(void) {
    const int64_t E = (int64_t)H * W * C;
    int n_dev = 0, cus = 0, lds = 0;
    char arch[64];
    CHECK_HM(hm_device_info(&n_dev, &cus, &lds, arch, (int)sizeof arch));
    printf("libhdrmerge ABI %d on %s (%d CUs)\n", hm_version(), arch, cus);

    /* host data: frames, stds, tables */
    static uint8_t frames[N][H * W * C];
    static double stds[N][H * W * C], icrf[256 * C], icrf_diff[256 * C], w_lut[256], dw_lut[256];
    double exposures[N];
    uint32_t seed = 12345u;
    for (int i = 0; i < N; ++i) {
        exposures[i] = 0.001 * (double)(1 << i);
        for (int64_t e = 0; e < E; ++e) {
            seed = seed * 1664525u + 1013904223u;
            const double rad = (double)(seed >> 8) / 16777216.0 * 4.0;
            double dn = floor(rad * (1 << i) * 255.0 / 16.0 + 0.5);
            frames[i][e] = (uint8_t)(dn > 255.0 ? 255.0 : dn);
            seed = seed * 1664525u + 1013904223u;
            stds[i][e] = 0.004 * (1.0 + (double)(seed >> 8) / 16777216.0);
        }
    }
    for (int k = 0; k < 256; ++k)
        for (int c = 0; c < C; ++c) icrf[k * C + c] = pow((double)k / 255.0, 1.8 + 0.2 * c);
    for (int k = 0; k < 256; ++k)                                  /* np.gradient with dx = 2 / 255 */
        for (int c = 0; c < C; ++c) {
            const int lo = k == 0 ? 0 : k - 1, hi = k == 255 ? 255 : k + 1;
            icrf_diff[k * C + c] = (icrf[hi * C + c] - icrf[lo * C + c]) / ((hi - lo) * (2.0 / 255.0));
        }
    CHECK_HM(hm_gaussian_weight_lut_host(w_lut, dw_lut));

    /* device buffers */
    void *d_frames[N], *d_stds[N];
    double *d_icrf, *d_diff, *d_w, *d_dw, *d_val, *d_std;
    for (int i = 0; i < N; ++i) {
        CHECK_HIP(hipMalloc(&d_frames[i], E));
        CHECK_HIP(hipMalloc(&d_stds[i], E * 8));
        CHECK_HIP(hipMemcpy(d_frames[i], frames[i], E, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_stds[i], stds[i], E * 8, hipMemcpyHostToDevice));
    }
    CHECK_HIP(hipMalloc((void**)&d_icrf, sizeof icrf)); CHECK_HIP(hipMalloc((void**)&d_diff, sizeof icrf_diff));
    CHECK_HIP(hipMalloc((void**)&d_w, sizeof w_lut)); CHECK_HIP(hipMalloc((void**)&d_dw, sizeof dw_lut));
    CHECK_HIP(hipMalloc((void**)&d_val, E * 8)); CHECK_HIP(hipMalloc((void**)&d_std, E * 8));
    CHECK_HIP(hipMemcpy(d_icrf, icrf, sizeof icrf, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_diff, icrf_diff, sizeof icrf_diff, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_w, w_lut, sizeof w_lut, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_dw, dw_lut, sizeof dw_lut, hipMemcpyHostToDevice));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    hm_merge_args a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a; a.n_frames = N; a.channels = C;
    a.height = H; a.width = W; a.row0 = 0; a.rows = H; a.buf_row0 = 0; a.buf_rows = H;
    a.frames_u8 = (const uint8_t* const*)d_frames; a.stds = (const double* const*)d_stds; a.exposures = exposures;
    a.icrf = d_icrf; a.icrf_diff = d_diff; a.w_lut = d_w; a.dw_lut = d_dw;
    a.out_val = d_val; a.out_std = d_std;
    CHECK_HM(hm_merge(&a, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    double* val = (double*)malloc(E * 8);
    double* sd = (double*)malloc(E * 8);
    CHECK_HIP(hipMemcpy(val, d_val, E * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(sd, d_std, E * 8, hipMemcpyDeviceToHost));

    /* host check */
    double worst_val = 0.0, worst_std = 0.0;
    for (int64_t e = 0; e < E; ++e) {
        const int c = (int)(e % C);
        double S = 0.0;
        for (int i = 0; i < N; ++i) S += w_lut[frames[i][e]];
        double v = 0.0, var = 0.0;
        for (int i = 0; i < N; ++i) {
            const int k = frames[i][e];
            const double w = w_lut[k], dw = dw_lut[k], g = icrf[k * C + c], dg = icrf_diff[k * C + c] * stds[i][e];
            v += (w * g) / (S * exposures[i]);
            const double t = ((dw * g + w * dg) / S - (dw * w * g) / (S * S)) * dg / exposures[i];
            var += t * t;
        }
        const double s = sqrt(var);
        const double dv = fabs(val[e] - v) / (fabs(v) > 0 ? fabs(v) : 1.0), ds = fabs(sd[e] - s) / (s > 0 ? s : 1.0);
        if (dv > worst_val) worst_val = dv;
        if (ds > worst_std) worst_std = ds;
    }
    printf("max relative difference: val %.3g, std %.3g over %lld elements (%lld algorithmic bytes)\n",
           worst_val, worst_std, (long long)E, (long long)hm_merge_algorithmic_bytes(&a));
    int graph_ok = 1;
#ifndef HM_HOST_BUILD
    /* The library enqueues kernels and nothing else (no allocation, copy or synchronisation): the same call is captured into a hipGraph as
     * it stands and the replay must write the same bits. (The argument struct is read at call time only; the device buffers must live on.) */
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
        CHECK_HM(hm_merge(&a, stream));
        CHECK_HIP(hipStreamEndCapture(stream, &graph));
        CHECK_HIP(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
        CHECK_HIP(hipMemset(d_val, 0xff, E * 8)); CHECK_HIP(hipMemset(d_std, 0xff, E * 8));
        for (int rep = 0; rep < 3; ++rep) CHECK_HIP(hipGraphLaunch(exec, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        double* val2 = (double*)malloc(E * 8);
        double* sd2 = (double*)malloc(E * 8);
        CHECK_HIP(hipMemcpy(val2, d_val, E * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(sd2, d_std, E * 8, hipMemcpyDeviceToHost));
        graph_ok = memcmp(val, val2, E * 8) == 0 && memcmp(sd, sd2, E * 8) == 0;
        printf("hipGraph replay of the captured hm_merge: %s\n", graph_ok ? "identical bits" : "DIFFERENT");
        CHECK_HIP(hipGraphExecDestroy(exec)); CHECK_HIP(hipGraphDestroy(graph));
        free(val2); free(sd2);
    }
#endif
    const int ok = worst_val < 1e-12 && worst_std < 1e-9 && graph_ok;
    puts(ok ? "C ABI merge OK" : "C ABI merge MISMATCH");
    return ok ? 0 : 1;
}
