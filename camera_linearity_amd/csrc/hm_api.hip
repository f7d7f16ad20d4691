// hm_api.hip - version / error strings / device query / host-side weight table of libhdrmerge.
#include "hm_common.h"
#include "hdrmerge_debug.h"
#include <cmath>
#include <cstring>

extern "C" int hm_version(void) { return HM_ABI_VERSION; }

extern "C" const char* hm_strerror(int code) {
    switch (code) {
        case HM_OK: return "ok";
        case HM_EINVAL: return "invalid argument";
        case HM_EUNSUPPORTED: return "unsupported configuration (channels > HM_MAX_CHANNELS, or frames > HM_MAX_FRAMES without frames_workspace / out_sum_w)";
        case HM_EALIGN: return "float64 buffer is not 8-byte aligned";
        case HM_ELAUNCH: return "HIP kernel launch failed";
        case HM_ENODEVICE: return "no usable gfx950 device";
        case HM_ESHAPE: return "inconsistent geometry (tile outside image or median halo missing)";
        default: return "unknown hdrmerge error";
    }
}

extern "C" int hm_device_info(int* n_devices, int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); if (n_devices) *n_devices = 0; return HM_ENODEVICE; }
    if (n_devices) *n_devices = n;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return HM_ENODEVICE; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = static_cast<int>(p.maxSharedMemoryPerMultiProcessor);
    if (arch && arch_len > 0) { std::strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return HM_OK;
}

// modules/measurand.py:615-616 on the DN grid v = k/255, evaluated with libm
extern "C" int hm_gaussian_weight_lut_host(double* w_lut, double* dw_lut) {
    if (!w_lut && !dw_lut) return HM_EINVAL;
    for (int k = 0; k < HM_BITS; ++k) {
        const double v = static_cast<double>(k) / 255.0;
        const double y = std::pow(M_E, -30.0 * ((v - 0.5) * (v - 0.5)));
        if (w_lut) w_lut[k] = y;
        if (dw_lut) dw_lut[k] = ((-2.0 * 30.0) * (v - 0.5)) * y;
    }
    return HM_OK;
}


// Shader-clock probe (diagnostic; docs/DESIGN_history_r01_r02.md 4.4): ONE wave on a side stream samples s_memtime (shader cycles) and s_memrealtime (100 MHz)
// at its start and after `spins` sleep periods, while the caller's kernels run on another stream: out[0] = shader cycles, out[1] = 100 MHz
// ticks, so clock [GHz] = out[0] / out[1] / 10. The values go to a buffer of their own; nothing else reads them.
namespace hm {
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, int spins) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
// One dword per thread and pass from a different `stride`-sized page each time: with many more pages than TLB entries the pass time is
// the address-translation cost of the buffer's physical backing (fragment size), not its bandwidth (tools/tlb_probe.py).
__global__ __launch_bounds__(256) void k_stride_probe(const uint32_t* __restrict__ p, uint64_t n_pages, uint64_t stride_dw, int passes,
                                                      uint32_t* __restrict__ sink) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t nthreads = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t in_page = stride_dw < 1024 ? stride_dw : 1024;                      // stay inside the first 4 KB of a page
    uint32_t acc = 0;
    for (int q = 0; q < passes; ++q) {
        const uint64_t page = (tid * 7u + static_cast<uint64_t>(q) * nthreads * 7u + static_cast<uint64_t>(q) * 131u) % n_pages;
        const uint64_t off = page * stride_dw + (static_cast<uint64_t>(q) * 16u) % in_page;
        acc += __builtin_nontemporal_load(p + off);
    }
    if (acc == 0x12345678u) sink[0] = acc;                                             // keeps the loads alive
}
// dst[i] = src[i], the copy shape that measured fastest on these boxes (tools/readbench.hip copysweep, profiles/r02h_copy_sweep.log): a wave
// reads 4 KB (four 16-byte nontemporal loads per lane, 1 KB per instruction), then writes those 4 KB; 6.0-6.7 TB/s where one 16-byte
// load + store per lane and step gets 5.3-5.9. The merge's traffic (47 % reads in seven byte streams, 53 % writes in one float64
// stream) is held against this rate by bench.py (roofline.copy_GBps).
constexpr int kCopyB = 4;
__global__ __launch_bounds__(256) void k_copy_probe(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t chunk = 128ll * kCopyB;                              // doubles per wave iteration
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * chunk;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4 * chunk;
    int64_t b = wave0;
    for (; b + chunk <= n; b += stride) {
        f64x2 v[kCopyB];
#pragma unroll
        for (int k = 0; k < kCopyB; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(src + b + 128 * k) + lane);
#pragma unroll
        for (int k = 0; k < kCopyB; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f64x2*>(dst + b + 128 * k) + lane);
    }
    for (int64_t e = b + lane; e < n && e < b + chunk; e += 64) dst[e] = src[e];     // the last, partial chunk of the one wave that owns it
}
}  // namespace hm

extern "C" int hm_debug_copy_probe(const void* src, void* dst, unsigned long long bytes, void* stream) {
    if (!src || !dst || bytes < 16 || (bytes & 15) || !hm::aligned(src, 16) || !hm::aligned(dst, 16)) return HM_EINVAL;
    const int64_t n = static_cast<int64_t>(bytes / 8);
    hipLaunchKernelGGL(hm::k_copy_probe, dim3(hm::stream_grid((n + 512 * hm::kCopyB - 1) / (512 * hm::kCopyB), 1, 16)), dim3(256), 0, hm::as_stream(stream),
                       static_cast<const double*>(src), static_cast<double*>(dst), n);
    return hm::launch_status();
}

extern "C" int hm_debug_stride_probe(const void* buf, unsigned long long bytes, unsigned long long stride_bytes, int passes, int blocks,
                                     unsigned int* sink_device, void* stream) {
    if (!buf || !sink_device || stride_bytes < 64 || (stride_bytes & 3) || bytes < stride_bytes || passes < 1 || blocks < 1) return HM_EINVAL;
    hipLaunchKernelGGL(hm::k_stride_probe, dim3(blocks), dim3(256), 0, hm::as_stream(stream), static_cast<const uint32_t*>(buf),
                       bytes / stride_bytes, stride_bytes / 4, passes, sink_device);
    return hm::launch_status();
}

extern "C" int hm_debug_clock_probe(unsigned long long* out_device /*2 x uint64*/, int spins, void* stream) {
    if (!out_device || spins < 1) return HM_EINVAL;
    hipLaunchKernelGGL(hm::k_clock_probe, dim3(1), dim3(64), 0, hm::as_stream(stream), out_device, spins);
    return hm::launch_status();
}
