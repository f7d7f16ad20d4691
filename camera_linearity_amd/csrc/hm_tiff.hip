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
//
// The string table holds no characters: every string a code stands for has ALREADY been written to the output, so an entry is
// (position, length) into dst. The entry made after emitting string(old) and reading `code` is string(old) + first byte of string(code) -
// and those are adjacent in the output: it is (position of old's emission, length(old) + 1). Emitting a code is then one forward copy
// inside dst (the KwKwK case - a code that is defined by the very emission in progress - is the same copy overlapping itself by one
// byte, as in LZ77). The first version walked a prefix chain backwards for every output byte (dependent loads): 135 MB/s on noise and
// 540 MB/s on smooth images per thread.
extern "C" int64_t hm_tiff_lzw_decode(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    if (!src || !dst || src_len < 0 || dst_cap < 0) return HM_EINVAL;
    enum { kClear = 256, kEoi = 257, kFirst = 258, kMax = 4096 };
    if (dst_cap > 0xFFFFFFFFll) dst_cap = 0xFFFFFFFFll;           // 32-bit positions (24 KB of tables, L1-resident); a strip is far smaller
    uint32_t pos[kMax];
    uint16_t len[kMax];
    int nbits = 9, next = kFirst, old = -1;
    uint32_t old_pos = 0, old_len = 0;
    uint64_t acc = 0;
    int have = 0;
    int64_t ip = 0, op = 0;
    for (;;) {
        if (have < nbits) {                                        // refill: four bytes at a time while the input lasts
            if (ip + 4 <= src_len) {
                acc = (acc << 32) | (static_cast<uint32_t>(src[ip]) << 24 | static_cast<uint32_t>(src[ip + 1]) << 16 |
                                     static_cast<uint32_t>(src[ip + 2]) << 8 | src[ip + 3]);
                ip += 4; have += 32;
            } else {
                while (have < nbits && ip < src_len) { acc = (acc << 8) | src[ip++]; have += 8; }
                if (have < nbits) break;                           // ran out of input without EOI: accept what we have
            }
        }
        const int code = static_cast<int>((acc >> (have - nbits)) & ((1u << nbits) - 1u));
        have -= nbits;
        if (code == kEoi) break;
        if (code == kClear) { nbits = 9; next = kFirst; old = -1; continue; }
        uint32_t from = 0, n = 1;                                  // the string of `code`: n bytes at dst + from (codes >= 258)
        if (old < 0) {
            if (code >= 256) return HM_EINVAL;
        } else {
            if (code > next || next >= kMax + 1) return HM_EINVAL;
            if (code >= kFirst) {
                if (code >= kMax) return HM_EINVAL;
                if (code < next) { from = pos[code]; n = len[code]; }
                else { from = old_pos; n = old_len + 1; }          // code == next: string(old) + its own first byte (overlaps by one)
            }
            if (next < kMax) {                                     // new entry: string(old) + first byte of the string being emitted now
                pos[next] = old_pos;
                len[next] = static_cast<uint16_t>(old_len + 1);
                ++next;
            }
        }
        if (op + n > dst_cap) return HM_ESHAPE;
        uint8_t* p = dst + op;
        if (code < 256) {
            *p = static_cast<uint8_t>(code);
        } else {
            const uint8_t* q = dst + from;
            if (n <= 16 && from + n <= op && op + 16 <= dst_cap) {
                // short strings (the common case on noisy images): sixteen bytes loaded, then stored - the bytes past n are overwritten by the
                // next emissions (reads stay inside dst: from < op and op + 16 <= dst_cap)
                uint64_t lo, hi;
                memcpy(&lo, q, 8); memcpy(&hi, q + 8, 8);
                memcpy(p, &lo, 8); memcpy(p + 8, &hi, 8);
            } else if (from + n <= op) {
                memcpy(p, q, n);
            } else {
                for (uint32_t k = 0; k < n; ++k) p[k] = q[k];      // forward: the one-byte overlap (KwKwK) reads what this loop wrote
            }
        }
        old = code; old_pos = static_cast<uint32_t>(op); old_len = n;
        op += n;
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
