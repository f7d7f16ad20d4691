// hm_tiff.hip - host-side strip decoders for the TIFF files the reference exchanges with OpenCV
// (modules/image_set.py:214-243 reads, :264-363 writes through cv.imread / cv.imwrite; OpenCV's TIFF writer
// uses LZW with the horizontal predictor for 8/16-bit images). Host code only: no device work. The container
// format itself (IFD parsing, predictor, channel order) lives in camera_linearity_amd/tiff_io.py; these two
// byte-serial loops are the part that is too slow in Python. Both are re-entrant (no globals), so the Python
// side decodes the strips of one image on several threads.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "hdrmerge.h"

// TIFF 6.0 LZW (Compression = 5): MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, code width
// grows one code early ("early change", as libtiff writes). Returns the number of bytes produced, or HM_EINVAL
// for a corrupt stream / HM_ESHAPE if dst_cap is too small.
extern "C" int64_t hm_tiff_lzw_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    if (!src || !dst || src_len < 0 || dst_cap < 0) return HM_EINVAL;
    enum { kClear = 256, kEoi = 257, kFirst = 258, kMax = 4096 };
    uint16_t prefix[kMax];
    uint8_t suffix[kMax], first[kMax];
    uint16_t length[kMax];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0; suffix[i] = static_cast<uint8_t>(i); first[i] = static_cast<uint8_t>(i); length[i] = 1; }
    int nbits = 9, next = kFirst, old = -1;
    uint64_t acc = 0;
    int have = 0;
    int64_t ip = 0, op = 0;
    for (;;) {
        while (have < nbits && ip < src_len) { acc = (acc << 8) | src[ip++]; have += 8; }
        if (have < nbits) break;                                   // ran out of input without EOI: accept what we have
        const int code = static_cast<int>((acc >> (have - nbits)) & ((1u << nbits) - 1u));
        have -= nbits;
        if (code == kEoi) break;
        if (code == kClear) { nbits = 9; next = kFirst; old = -1; continue; }
        int emit;                                                  // the table entry to write out
        if (old < 0) {
            if (code >= 256) return HM_EINVAL;
            emit = code;
        } else {
            if (code > next || next >= kMax + 1) return HM_EINVAL;
            if (next < kMax) {                                     // new entry: string(old) + first char of the emitted string
                prefix[next] = static_cast<uint16_t>(old);
                first[next] = first[old];
                length[next] = static_cast<uint16_t>(length[old] + 1);
                suffix[next] = code < next ? first[code] : first[old];
                ++next;
            } else if (code >= kMax) {
                return HM_EINVAL;
            }
            emit = code;
        }
        const int len = length[emit];
        if (op + len > dst_cap) return HM_ESHAPE;
        uint8_t* p = dst + op + len;
        for (int c = emit, k = 0; k < len; ++k) { *--p = suffix[c]; c = prefix[c]; }
        op += len;
        old = code;
        if (next + 1 >= (1 << nbits) && nbits < 12) ++nbits;       // early change
    }
    return op;
}

// PackBits (Compression = 32773): n in [0,127] -> copy n+1 literal bytes; n in [-127,-1] -> repeat next byte 1-n times.
extern "C" int64_t hm_tiff_packbits_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    if (!src || !dst || src_len < 0 || dst_cap < 0) return HM_EINVAL;
    int64_t ip = 0, op = 0;
    while (ip < src_len) {
        const int n = static_cast<int8_t>(src[ip++]);
        if (n >= 0) {
            const int cnt = n + 1;
            if (ip + cnt > src_len) return HM_EINVAL;
            if (op + cnt > dst_cap) return HM_ESHAPE;
            memcpy(dst + op, src + ip, cnt);
            ip += cnt; op += cnt;
        } else if (n != -128) {
            const int cnt = 1 - n;
            if (ip >= src_len) return HM_EINVAL;
            if (op + cnt > dst_cap) return HM_ESHAPE;
            memset(dst + op, src[ip++], cnt);
            op += cnt;
        }
    }
    return op;
}
