// utf8_decode.h -- UTF-8 sequence decode shared by the staged decoder (aux_kernels.hip) and the byte-space tile kernel.
#ifndef LATOK_UTF8_DECODE_H
#define LATOK_UTF8_DECODE_H
#include <stdint.h>

namespace latok {

// Decode the sequence whose lead byte is byte I of the 19 bytes {w0..w3, 3 bytes of w4}; all indices are static.
// One code point per lead byte; a truncated sequence yields U+FFFD; "surrogatepass" forms decode as they are.
template <int I>
__device__ __forceinline__ uint32_t utf8_decode_at(const uint32_t (&w)[5]) {
    auto byte_at = [&](int j) -> uint32_t { return (w[j >> 2] >> (8 * (j & 3))) & 0xFFu; };
    const uint32_t b0 = byte_at(I);
    if (b0 < 0x80u) return b0;
    uint32_t cp = b0;
    int extra = 0;
    if (b0 >= 0xF0u) { cp = b0 & 0x07u; extra = 3; }
    else if (b0 >= 0xE0u) { cp = b0 & 0x0Fu; extra = 2; }
    else if (b0 >= 0xC0u) { cp = b0 & 0x1Fu; extra = 1; }
#pragma unroll
    for (int j = 1; j <= 3; ++j) {
        if (j <= extra) {
            const uint32_t b = byte_at(I + j);
            if ((b & 0xC0u) == 0x80u) cp = (cp << 6) | (b & 0x3Fu);
            else { cp = 0xFFFDu; extra = 0; }                          // truncated sequence
        }
    }
    return cp;
}

// same for a lead byte b0 followed by b1..b3 (pass 0xFF for bytes that do not exist)
__device__ __forceinline__ uint32_t utf8_decode_bytes(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    const uint32_t w[5] = {b0 | (b1 << 8) | (b2 << 16) | (b3 << 24), 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    return utf8_decode_at<0>(w);
}

// Code point of the multi-byte sequence whose lead byte (>= 0xC0) is the LOW byte of W (W = 4 bytes in memory order).
// Branch-free restatement of utf8_decode_at (utf8_decode.h): a truncated sequence yields U+FFFD, "surrogatepass" and
// overlong forms decode as they are, 0xF8..0xFF count as 4-byte leads with 3 payload bits.
__device__ __forceinline__ uint32_t utf8_cp_of(uint32_t W) {
    const uint32_t b0 = W & 0xFFu;
    const uint32_t X = (((W >> 8) & 0x3Fu) << 12) | (((W >> 16) & 0x3Fu) << 6) | ((W >> 24) & 0x3Fu);   // payload of bytes 1..3
    // pure arithmetic on the sequence length n = 2, 3, 4 (selects between three forms end up as divergent control flow,
    // and that serialises the table lookups of the slots of a row)
    const uint32_t n = 2u + (uint32_t)(b0 >= 0xE0u) + (uint32_t)(b0 >= 0xF0u);
    const uint32_t sh = 24u - 6u * n;                                   // payload bits of bytes 1..3 that are not used
    const uint32_t cp = ((b0 & (0x7Fu >> n)) << (18u - sh)) | (X >> sh);
    const uint32_t notc = (W ^ 0x80808000u) & 0xC0C0C000u;            // byte j != 0  <=>  byte j is not 10xxxxxx
    const uint32_t need = (0xC0C0C000u >> (32u - 8u * n)) & 0xFFFFFF00u;
    return (notc & need) ? 0xFFFDu : cp;
}
// bit i of the result = byte i of the dword is a lead byte ((b & 0xC0) != 0x80)
__device__ __forceinline__ uint32_t utf8_lead_nibble(uint32_t w) {
    const uint32_t cont = (w & 0x80808080u) & ~((w << 1) & 0x80808080u);   // top bits "10"
    const uint32_t lead = (~cont) & 0x80808080u;
    return (((lead >> 7) * 0x00204081u) >> 21) & 0xFu;
}

}  // namespace latok
#endif
