/* hdrmerge_debug.h - diagnostic entry points of libhdrmerge.so (MI355X / gfx950).
 *
 * NOT part of the drop-in boundary: none of these replaces a function of samivout/camera_linearity. They are measurement probes
 * used by bench.py (the box's copy rate printed beside the roofline) and by the experiment scripts under tools/ (shader clock under
 * load, address-translation cost of a buffer's backing). Same conventions as hdrmerge.h: device pointers, `stream` = hipStream_t,
 * HM_OK or a negative HM_E* code. A production build may drop this header and the three symbols.
 */
#ifndef HDRMERGE_DEBUG_H
#define HDRMERGE_DEBUG_H
#include "hdrmerge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: one wave samples the shader-cycle counter and the 100 MHz real-time counter over `spins` sleep periods on `stream` (launch it on a
 * side stream while the kernels of interest run): out_device[0] = shader cycles, out_device[1] = 100 MHz ticks -> clock [GHz] = [0] / [1] / 10. */
int hm_debug_clock_probe(unsigned long long* out_device, int spins, void* stream);
/* Debug: dst[0 .. bytes) = src[0 .. bytes) with 16-byte nontemporal loads and stores (16-byte aligned, bytes a multiple of 16): the
 * copy bandwidth of the box, against which bench.py holds the merge's read / write mix (roofline.copy_GBps). No reference counterpart. */
int hm_debug_copy_probe(const void* src, void* dst, unsigned long long bytes, void* stream);
/* Debug: `blocks` x 256 threads each read `passes` dwords, every one from a different `stride_bytes`-sized page of buf[0 .. bytes): the
 * launch time reflects the address-translation cost of the buffer's physical backing (tools/tlb_probe.py). No reference counterpart. */
int hm_debug_stride_probe(const void* buf, unsigned long long bytes, unsigned long long stride_bytes, int passes, int blocks,
                          unsigned int* sink_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDRMERGE_DEBUG_H */
