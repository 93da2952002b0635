// lane_math.h -- per-lane (one 64-char word) bit-sliced arithmetic of the fused split-mask kernel.
//
// One lane of a wavefront owns one 64-bit word of every feature plane: bit i of a plane = the feature of char
// (word_base + i).  Everything in this header is pure integer math on that word plus a few carry bits, so it compiles
// for the device (hipcc) and for the host (the CPU model under oracle/ that is used to debug the algorithm; the
// product never runs it).
//
// Reference semantics restated here (all citations are to the reference tree):
//   * base features + context columns .......... latok/core/src/latok/latok.c:87-134
//   * C_SPLIT / C_MASK / C_SYM rule tables ..... latok/core/default_tokenizer.py:39-110
//   * gen_block_mask queue semantics ........... latok/core/src/latok/latok.c:217-245
//   * splits = raw*mask + sym; splits[0]=1 ..... latok/core/default_tokenizer.py:121-132
#ifndef LATOK_LANE_MATH_H
#define LATOK_LANE_MATH_H
#include <stdint.h>

#include "split_code.h"

#if defined(__HIPCC__)
#define LATOK_HD __host__ __device__ inline
#define LATOK_HD_COLD __host__ __device__ __attribute__((noinline))   // rare paths: a real call keeps them out of the hot loop's registers
#else
#define LATOK_HD static inline
#define LATOK_HD_COLD static
#endif

typedef unsigned long long lk_u64;

LATOK_HD int lk_popc(lk_u64 x) { return __builtin_popcountll(x); }
LATOK_HD int lk_ctz(lk_u64 x) { return __builtin_ctzll(x); }  // x != 0
LATOK_HD lk_u64 lk_rev(lk_u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// A 64-position plane as two 32-bit halves.  gfx950 has v_bitop3_b32 -- any boolean function of three words in one
// instruction -- and the backend forms it from chains of 32-bit and / or / xor / not, but NOT from 64-bit ones (those are
// split into halves only after instruction selection).  The same boolean algebra written on lk_w costs ~20 % fewer VALU
// instructions than on lk_u64 (lk_ascii_code_planes: 117 -> 95 per word); shifts by 1 / 2 with a bit injected at the open
// end are one v_alignbit_b32 + one v_lshl_or_b32.  The CPU model compiles the same code (tests/test_fused_model.py).
// ---------------------------------------------------------------------------------------------------------------
struct lk_w {
    uint32_t lo, hi;
};
LATOK_HD lk_w lk_w_of(lk_u64 x) { lk_w r; r.lo = (uint32_t)x; r.hi = (uint32_t)(x >> 32); return r; }
LATOK_HD lk_u64 lk_u_of(lk_w x) { return (lk_u64)x.lo | ((lk_u64)x.hi << 32); }
LATOK_HD lk_w operator&(lk_w a, lk_w b) { lk_w r; r.lo = a.lo & b.lo; r.hi = a.hi & b.hi; return r; }
LATOK_HD lk_w operator|(lk_w a, lk_w b) { lk_w r; r.lo = a.lo | b.lo; r.hi = a.hi | b.hi; return r; }
LATOK_HD lk_w operator^(lk_w a, lk_w b) { lk_w r; r.lo = a.lo ^ b.lo; r.hi = a.hi ^ b.hi; return r; }
LATOK_HD lk_w operator~(lk_w a) { lk_w r; r.lo = ~a.lo; r.hi = ~a.hi; return r; }
// x << K with the low K bits of `in` entering at the bottom;  x >> K with the low K bits of `in` entering at the top
template <int K>
LATOK_HD lk_w lk_w_shl(lk_w x, uint32_t in) {
    lk_w r;
    r.lo = (x.lo << K) | in;
    r.hi = (x.hi << K) | (x.lo >> (32 - K));
    return r;
}
template <int K>
LATOK_HD lk_w lk_w_shr(lk_w x, uint32_t in) {
    lk_w r;
    r.lo = (x.lo >> K) | (x.hi << (32 - K));
    r.hi = (x.hi >> K) | (in << (32 - K));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Parallel bit extract (x86 pext; Hacker's Delight 7-4 "compress"): the bits of x at the set positions of m, packed at the
// bottom in order.  (This arithmetic form is the definition the table form below is tested against; the kernel uses the table.)  Code-point results from a byte-space mask: m = the lead bytes of a 64-byte word, x = boundary bits at lead
// bytes -> boundary bits of the word's chars (compact_kernels.hip: k_lead_compress).  On 32-bit halves: five rounds each.
// ---------------------------------------------------------------------------------------------------------------
LATOK_HD uint32_t lk_pext32(uint32_t x, uint32_t m) {
    x &= m;
    uint32_t mk = ~m << 1;                    // counts the 0s to the right of every position, one bit of the count per round
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 5; ++i) {
        uint32_t mp = mk ^ (mk << 1);         // parallel suffix: parity of the zeros below
        mp ^= mp << 2;
        mp ^= mp << 4;
        mp ^= mp << 8;
        mp ^= mp << 16;
        const uint32_t mv = mp & m;           // bits that move by 2^i in this round
        m = (m ^ mv) | (mv >> (1 << i));
        const uint32_t t = x & mv;
        x = (x ^ t) | (t >> (1 << i));
        mk &= ~mp;
    }
    return x;
}
LATOK_HD lk_u64 lk_pext64(lk_u64 x, lk_u64 m) {
    const uint32_t lo = lk_pext32((uint32_t)x, (uint32_t)m), hi = lk_pext32((uint32_t)(x >> 32), (uint32_t)(m >> 32));
    return (lk_u64)lo | ((lk_u64)hi << __builtin_popcount((uint32_t)m));   // (a shift by 32 is fine on 64 bits)
}
// The same through a 256-byte table of 4-bit pexts (k_lead_compress keeps it in LDS: 64 dwords = one per bank, so the byte reads of
// a wave never conflict): entry (m4 << 4 | x4) = pext4(x4, m4) | popcount(m4) << 4.  Per nibble one lookup, a shift-or and an add
// (the eight table indices of a 32-bit half come out of six mask / shift instructions): ~100 VALU per 64-bit word against ~256
// for the five-round compress -- the kernel is bound by exactly this.  A second word through the same mask shares the positions.
LATOK_HD uint8_t lk_pext4_entry(uint32_t i) {
    const uint32_t m = i >> 4, x = i & 15u;
    uint32_t out = 0, k = 0;
    for (uint32_t b = 0; b < 4; ++b)
        if ((m >> b) & 1u) { out |= ((x >> b) & 1u) << k; ++k; }
    return (uint8_t)(out | (k << 4));
}
template <bool TWO>
LATOK_HD void lk_pext32_lut(uint32_t x, uint32_t x2, uint32_t m, const uint8_t* tab, uint32_t* out, uint32_t* out2, uint32_t* n_out) {
    // byte k of ie / io = table index of nibble 2k / 2k + 1
    const uint32_t mh = (m & 0x0F0F0F0Fu) << 4, mo = m & 0xF0F0F0F0u;
    const uint32_t ie = mh | (x & 0x0F0F0F0Fu), io = mo | ((x >> 4) & 0x0F0F0F0Fu);
    const uint32_t ie2 = TWO ? (mh | (x2 & 0x0F0F0F0Fu)) : 0u, io2 = TWO ? (mo | ((x2 >> 4) & 0x0F0F0F0Fu)) : 0u;
    uint32_t acc = 0, acc2 = 0, pos = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
        const uint32_t sh = 8u * (uint32_t)(k >> 1);
        const uint32_t e = tab[((k & 1) ? io : ie) >> sh & 0xFFu];
        acc |= (e & 15u) << pos;
        if (TWO) acc2 |= ((uint32_t)tab[((k & 1) ? io2 : ie2) >> sh & 0xFFu] & 15u) << pos;
        pos += e >> 4;
    }
    *out = acc;
    if (TWO) *out2 = acc2;
    *n_out = pos;
}
template <bool TWO>
LATOK_HD void lk_pext64_lut(lk_u64* x1, lk_u64* x2, lk_u64 m, const uint8_t* tab) {
    uint32_t a0, a1, b0 = 0, b1 = 0, n0, n1;
    lk_pext32_lut<TWO>((uint32_t)*x1, TWO ? (uint32_t)*x2 : 0u, (uint32_t)m, tab, &a0, &b0, &n0);
    lk_pext32_lut<TWO>((uint32_t)(*x1 >> 32), TWO ? (uint32_t)(*x2 >> 32) : 0u, (uint32_t)(m >> 32), tab, &a1, &b1, &n1);
    *x1 = (lk_u64)a0 | ((lk_u64)a1 << n0);
    if (TWO) *x2 = (lk_u64)b0 | ((lk_u64)b1 << n0);
}

// ---------------------------------------------------------------------------------------------------------------
// Byte space: class-table indices of a char straight from its bytes.  The byte-space kernel has its own two-stage class table,
// cut where UTF-8 cuts: stage 1 is indexed by cp >> 6 -- everything but the last byte's payload -- and holds the offset of a
// 64-entry stage-2 block, which the last byte's six payload bits index (LK_B6_*; api.cpp derives both stages from the
// generated tables).  The code point itself is never assembled.  An 8-byte table entry per byte value b0 holds
//   .sel   a v_perm selector that brings the sequence into ONE order whatever its length: R = {last byte, the one before it
//          (3- and 4-byte sequences), the one before that (4-byte sequences), -}; every position the sequence does not fill
//          takes the constant 0x80, an empty continuation byte
//   .hi0   what the lead byte itself contributes to the stage-1 index, as the BYTE offset of the 16-bit entry (2 x (cp >> 6)),
//          less what the 10xxxxxx prefixes of R.byte1 and R.byte2 add to the sum below (the same for every length)
// so that the stage-1 offset is hi0 + 2 * R.byte1 + 128 * R.byte2 -- ONE v_dot4_u32_u8 with the entry as its addend --, the
// stage-2 index is the block offset | R & 0x3F, and the sequence is cut short exactly when some byte of R is not 10xxxxxx.
// lk_lead_entry_of(b0) = the entry of byte b0: 0xC0..0xFF are lead bytes (0xF8..0xFF count as 4-byte leads with 3 payload
// bits); a byte below 0xC0 starts no multi-byte char: its entry selects nothing (R = 0x80808080: never "cut short", index 0
// in its block) and points at stage 1's last entry, the block of cp >= 0x110000, whose codes are all 0 -- a decode slot that
// holds no lead adds nothing to the staging bytes without a select.
// lk_lead_index(entry, W) with W = the 4 bytes from b0 on: *off2 = byte offset into stage 1 (clamp it to the last entry:
// sequences that decode beyond U+10FFFF, and whatever a sequence that is cut short adds up to), *R_out = R (low six bits = the
// stage-2 index within the block); returns true when the sequence is cut short (the char is U+FFFD then).
// ---------------------------------------------------------------------------------------------------------------
struct lk_lead_entry {
    uint32_t sel;
    uint32_t hi0;
};
LATOK_HD lk_lead_entry lk_lead_entry_of(uint32_t b0) {
    lk_lead_entry e;
    const uint32_t prefixes = 0x80u + 64u * 0x80u;   // of R.byte1 / R.byte2 in the sum (a position without a byte holds 0x80 too)
    if (b0 < 0xC0u) {
        e.sel = 0x04040404u;
        e.hi0 = 2u * ((uint32_t)(LK_B6_STAGE1_LEN - 1) - prefixes);
        return e;
    }
    const int n = 2 + (b0 >= 0xE0u) + (b0 >= 0xF0u);
    e.sel = n == 2 ? 0x04040401u : (n == 3 ? 0x04040102u : 0x04010203u);   // (0..3: byte of W, 4: the constant)
    const uint32_t own = n == 2 ? (b0 & 31u) : (n == 3 ? (b0 & 15u) << 6 : (b0 & 7u) << 12);   // the lead's part of cp >> 6
    e.hi0 = 2u * (own - prefixes);                                                               // (mod 2^32: the sum is exact again)
    return e;
}
LATOK_HD bool lk_lead_index(lk_lead_entry e, uint32_t W, uint32_t* off2, uint32_t* R_out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t R = __builtin_amdgcn_perm(0x80808080u, W, e.sel);
    *off2 = __builtin_amdgcn_udot4(R, 0x00800200u, e.hi0, false);
#else
    uint32_t R = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t s = (e.sel >> (8 * k)) & 0xFFu;
        R |= (s < 4u ? (W >> (8 * s)) & 0xFFu : 0x80u) << (8 * k);
    }
    *off2 = e.hi0 + 2u * ((R >> 8) & 0xFFu) + 128u * ((R >> 16) & 0xFFu);
#endif
    *R_out = R;
    return ((R ^ 0x80808080u) & 0xC0C0C0C0u) != 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// 8x8 bit-matrix transpose: input byte r (bits 8r..8r+7) = row r; output byte c holds column c (bit r = row r).
// ---------------------------------------------------------------------------------------------------------------
LATOK_HD lk_u64 lk_transpose8(lk_u64 x) {
    lk_u64 t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// 64 code bytes (d[k] = chars 4k..4k+3, little endian) -> 8 planes; plane[b] bit i = bit b of char i's code.
#if defined(__HIP_DEVICE_COMPILE__)
// Second half of the bit-slicing: lo[g] / hi[g] hold, for chars 8g..8g+7, plane b in byte b (lo: planes 0..3, hi: 4..7;
// bit j of the byte = char 8g + j).  4x4 byte transposes regroup them into the 64-bit planes.
__device__ __forceinline__ void lk_planes_from_groups(const uint32_t lo[8], const uint32_t hi[8], lk_u64 plane[8]) {
    // 4x4 byte transposes: out[b] = {x0.byte b, x1.byte b, x2.byte b, x3.byte b}
#define LK_T4(x0, x1, x2, x3, o0, o1, o2, o3)                                   \
    {                                                                           \
        const uint32_t e01 = __builtin_amdgcn_perm(x1, x0, 0x06020400u);        \
        const uint32_t o01 = __builtin_amdgcn_perm(x1, x0, 0x07030501u);        \
        const uint32_t e23 = __builtin_amdgcn_perm(x3, x2, 0x06020400u);        \
        const uint32_t o23 = __builtin_amdgcn_perm(x3, x2, 0x07030501u);        \
        o0 = __builtin_amdgcn_perm(e23, e01, 0x05040100u);                      \
        o2 = __builtin_amdgcn_perm(e23, e01, 0x07060302u);                      \
        o1 = __builtin_amdgcn_perm(o23, o01, 0x05040100u);                      \
        o3 = __builtin_amdgcn_perm(o23, o01, 0x07060302u);                      \
    }
    uint32_t pl[8], ph[8];
    LK_T4(lo[0], lo[1], lo[2], lo[3], pl[0], pl[1], pl[2], pl[3]);
    LK_T4(lo[4], lo[5], lo[6], lo[7], ph[0], ph[1], ph[2], ph[3]);
    LK_T4(hi[0], hi[1], hi[2], hi[3], pl[4], pl[5], pl[6], pl[7]);
    LK_T4(hi[4], hi[5], hi[6], hi[7], ph[4], ph[5], ph[6], ph[7]);
#undef LK_T4
#pragma unroll
    for (int b = 0; b < 8; ++b) plane[b] = (lk_u64)pl[b] | ((lk_u64)ph[b] << 32);
}
// Device form: the same three delta-swap stages written on 32-bit halves, then the byte regrouping done with
// v_perm_b32 (one instruction per output dword half-pair) instead of shift/mask chains.
__device__ __forceinline__ void lk_bitslice64(const uint32_t d[16], lk_u64 plane[8]) {
    uint32_t lo[8], hi[8];   // after the transposes: byte b of lo[g] = plane b (b < 4), of hi[g] = plane 4 + b, chars 8g..8g+7
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        uint32_t a = d[2 * g], b = d[2 * g + 1], t;
        t = (a ^ (a >> 7)) & 0x00AA00AAu;  a ^= t ^ (t << 7);
        t = (b ^ (b >> 7)) & 0x00AA00AAu;  b ^= t ^ (t << 7);
        t = (a ^ (a >> 14)) & 0x0000CCCCu; a ^= t ^ (t << 14);
        t = (b ^ (b >> 14)) & 0x0000CCCCu; b ^= t ^ (t << 14);
        // stage 3 of the 64-bit form: t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0 ; x ^= t ^ (t << 28)
        t = (a ^ ((a >> 28) | (b << 4))) & 0xF0F0F0F0u;
        a ^= t ^ (t << 28);
        b ^= (t >> 4);
        lo[g] = a;
        hi[g] = b;
    }
    lk_planes_from_groups(lo, hi, plane);
}
#else
LATOK_HD void lk_planes_from_groups(const uint32_t lo[8], const uint32_t hi[8], lk_u64 plane[8]) {
    for (int b = 0; b < 8; ++b) {
        lk_u64 p = 0;
        for (int g = 0; g < 8; ++g) p |= (lk_u64)(((b < 4 ? lo[g] : hi[g]) >> (8 * (b & 3))) & 0xFFu) << (8 * g);
        plane[b] = p;
    }
}
LATOK_HD void lk_bitslice64(const uint32_t d[16], lk_u64 plane[8]) {
    lk_u64 y[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) y[g] = lk_transpose8((lk_u64)d[2 * g] | ((lk_u64)d[2 * g + 1] << 32));
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        lk_u64 p = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) p |= ((y[g] >> (8 * b)) & 0xFFull) << (8 * g);
        plane[b] = p;
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Features of one word, decoded from the split-code planes (split_code.h).
// ---------------------------------------------------------------------------------------------------------------
struct lk_feat {
    lk_u64 S, Y, L, U, AN, A, T, AT, CO, SL, PE;
};

template <class W>
struct lk_feat_t {
    W S, Y, L, U, AN, A, T, AT, CO, SL, PE;
};
template <class W>
LATOK_HD lk_feat_t<W> lk_decode_t(const W p[8]) {   // the same decode on any word type (uint32_t halves, lk_w)
    lk_feat_t<W> f;
    f.S = p[LK_BIT_SPACE];
    f.Y = p[LK_BIT_SYMBOL];
    f.L = p[LK_BIT_LOWER];
    f.U = p[LK_BIT_UPPER];
    f.AN = p[LK_BIT_ALNUM];
    f.A = p[5] & ~f.Y;
    f.T = p[5] & f.Y & ~p[7];
    f.AT = p[5] & p[6] & f.Y;
    f.CO = p[7] & ~p[6] & ~p[5];
    f.SL = p[7] & p[5];
    f.PE = p[7] & p[6];
    return f;
}

LATOK_HD lk_feat lk_decode(const lk_u64 p[8]) {
    lk_w q[8];
    for (int b = 0; b < 8; ++b) q[b] = lk_w_of(p[b]);
    const lk_feat_t<lk_w> g = lk_decode_t<lk_w>(q);    // on halves: see lk_w
    lk_feat f;
    f.S = lk_u_of(g.S); f.Y = lk_u_of(g.Y); f.L = lk_u_of(g.L); f.U = lk_u_of(g.U); f.AN = lk_u_of(g.AN); f.A = lk_u_of(g.A);
    f.T = lk_u_of(g.T); f.AT = lk_u_of(g.AT); f.CO = lk_u_of(g.CO); f.SL = lk_u_of(g.SL); f.PE = lk_u_of(g.PE);
    return f;
}

// ---------------------------------------------------------------------------------------------------------------
// ASCII without a table: the 8 split-code planes of a word as boolean functions of the 7 bit planes of its RAW bytes
// (r[b] bit i = bit b of byte i; every byte < 0x80).  The class of an ASCII char is a handful of range and equality
// tests on its bits (reference flag derivation, scripts/unicode/makeunicodedata.py:167-200,256-257, swept into
// unicode_tables.inc: SPACE = 09-0D 1C-20, digits, A-Z, a-z, SYMBOL = every other printable 21-7E, sub-types for
// # $ ^ @ : / .), so a word costs ~120 bit operations instead of 128 table lookups through the LDS pipe.
// tests/test_fused_model.py checks all 128 values in every position against the table.
// ---------------------------------------------------------------------------------------------------------------
template <bool RULE_CODES, class W>   // rule codes carry NUM in bit 6 of non-symbols (split_code.h); W: lk_u64 or one 32-bit half
LATOK_HD void lk_ascii_code_planes_t(const W r[8], W p[8]) {
    const W b0 = r[0], b1 = r[1], b2 = r[2], b3 = r[3], b4 = r[4], b5 = r[5], b6 = r[6];
    const W nz = b3 | b2 | b1 | b0;                      // low nibble != 0
    const W le10 = ~(b3 & (b2 | (b1 & b0)));             // low nibble <= 10
    const W le9 = ~(b3 & (b2 | b1));                     // low nibble <= 9
    const W g2 = ~b6 & b5 & ~b4, g3 = ~b6 & b5 & b4;     // 0x2_, 0x3_
    const W letter = b6 & ((~b4 & nz) | (b4 & le10));    // 41-5A, 61-7A
    const W digit = g3 & le9;
    const W alnum = letter | digit;
    const W sp20 = g2 & ~nz;                             // 0x20
    const W space = (~b6 & ~b5 & b3 & ((~b4 & ((~b2 & (b1 | b0)) | (b2 & ~b1))) | (b4 & b2))) | sp20;   // 09-0D, 1C-1F, 20
    const W del = b6 & b5 & b4 & b3 & b2 & b1 & b0;      // 0x7F
    const W sym = (b6 | b5) & ~sp20 & ~del & ~alnum;
    const W n_e = b3 & b2 & b1 & ~b0, n_f = b3 & b2 & b1 & b0;
    const W c_hash = g2 & ~b3 & ~b2 & b1 & b0, c_dollar = g2 & ~b3 & b2 & ~b1 & ~b0, c_caret = b6 & ~b5 & b4 & n_e;
    const W c_at = b6 & ~b5 & ~b4 & ~nz, c_colon = g3 & b3 & ~b2 & b1 & ~b0, c_slash = g2 & n_f, c_dot = g2 & n_e;
    p[LK_BIT_SPACE] = space;
    p[LK_BIT_SYMBOL] = sym;
    p[LK_BIT_LOWER] = letter & b5;
    p[LK_BIT_UPPER] = letter & ~b5;
    p[LK_BIT_ALNUM] = alnum;
    p[5] = letter | c_hash | c_dollar | c_caret | c_at | c_slash;
    p[6] = c_at | c_dot | (RULE_CODES ? digit : (W)0);
    p[7] = c_colon | c_slash | c_dot;
}

template <bool RULE_CODES = false>
LATOK_HD void lk_ascii_code_planes(const lk_u64 r[8], lk_u64 p[8]) {
    // on 32-bit halves: the backend fuses the chains into v_bitop3_b32 (see lk_w)
    uint32_t rl[8], rh[8], pl[8], ph[8];
    for (int b = 0; b < 7; ++b) {
        rl[b] = (uint32_t)r[b];
        rh[b] = (uint32_t)(r[b] >> 32);
    }
    lk_ascii_code_planes_t<RULE_CODES, uint32_t>(rl, pl);
    lk_ascii_code_planes_t<RULE_CODES, uint32_t>(rh, ph);
    for (int b = 0; b < 8; ++b) p[b] = (lk_u64)pl[b] | ((lk_u64)ph[b] << 32);
}

// neighbour characters outside the word: codes of char (base-1), (base+64), (base+65); 0 when they do not exist
struct lk_halo {
    uint32_t prev, next0, next1;
};

// outputs of the purely local rules for one word
struct lk_local {
    lk_u64 t_space, t_sym, t_prevsym, t_camel_next, t_camel_prev;  // the five C_SPLIT terms (for 0..5 values)
    lk_u64 raw;    // OR of the five terms  (C_SPLIT != 0)
    lk_u64 start;  // C_MASK term: URL / e-mail / twitter starts
    lk_u64 sym;    // C_SYM term: SYMBOL & NEXT_SPACE
    lk_u64 S;      // space plane (block delimiters)
};

// B  = string-start bits of this word; Bn = bits 0..1 = string-start bits of the next two chars after the word.
// Edge conventions (latok.c:69-73,114-134): at a string start PREV_SPACE=1 and the other PREV_* are 0; at a string
// end NEXT_SPACE=1 and the other NEXT_* are 0; AFTER_NEXT_* are 0 for the last two chars.
// single-character decode of a split code (same layout as lk_decode) for the three halo chars: 32-bit, bit 0 only
struct lk_feat1 {
    uint32_t S, Y, L, AN, A, AT, SL;
};
LATOK_HD lk_feat1 lk_decode1(uint32_t c) {
    lk_feat1 f;
    f.S = c & 1u;
    f.Y = (c >> 1) & 1u;
    f.L = (c >> 2) & 1u;
    f.AN = (c >> 4) & 1u;
    f.A = (c >> 5) & ~(c >> 1) & 1u;
    f.AT = (c >> 5) & (c >> 6) & (c >> 1) & 1u;
    f.SL = (c >> 7) & (c >> 5) & 1u;
    return f;
}

LATOK_HD lk_local lk_rules(const lk_feat& f64, lk_halo h, lk_u64 B64, lk_u64 Bn) {
    const lk_feat1 fp = lk_decode1(h.prev), f0 = lk_decode1(h.next0), f1 = lk_decode1(h.next1);
    // the planes as 32-bit halves (lk_w): the and / or chains below become v_bitop3_b32
    const lk_w B = lk_w_of(B64);
    const lk_w S = lk_w_of(f64.S), Y = lk_w_of(f64.Y), L = lk_w_of(f64.L), U = lk_w_of(f64.U), AN = lk_w_of(f64.AN), A = lk_w_of(f64.A),
               T = lk_w_of(f64.T), AT = lk_w_of(f64.AT), CO = lk_w_of(f64.CO), SL = lk_w_of(f64.SL), PE = lk_w_of(f64.PE);

    const lk_w E = lk_w_shr<1>(B, (uint32_t)(Bn & 1ull));                   // last char of a string
    const lk_w E2 = E | lk_w_shr<2>(B, (uint32_t)(Bn & 3ull));              // last or second-to-last char
    const lk_w nB = ~B, nE = ~E, nE2 = ~E2;

#define LK_PREV(X) (lk_w_shl<1>(X, fp.X) & nB)
#define LK_NEXT(X) (lk_w_shr<1>(X, f0.X) & nE)
#define LK_ANEXT(X) (lk_w_shr<2>(X, f0.X | (f1.X << 1)) & nE2)
    const lk_w prevS = lk_w_shl<1>(S, fp.S) | B;
    const lk_w nextS = lk_w_shr<1>(S, f0.S) | E;
    const lk_w prevY = LK_PREV(Y), prevL = LK_PREV(L), prevAN = LK_PREV(AN), prevA = LK_PREV(A);
    const lk_w nextL = LK_NEXT(L), nextA = LK_NEXT(A), nextAN = LK_NEXT(AN), nextAT = LK_NEXT(AT), nextSL = LK_NEXT(SL);
    const lk_w anA = LK_ANEXT(A), anSL = LK_ANEXT(SL);
#undef LK_PREV
#undef LK_NEXT
#undef LK_ANEXT

    lk_local r;
    r.S = f64.S;
    r.t_space = f64.S;                // [SPACE]
    r.t_sym = f64.Y;                  // [SYMBOL]
    const lk_w camel_next = U & nextL, camel_prev = U & prevL;
    r.t_prevsym = lk_u_of(prevY);     // [PREV_SYMBOL]
    r.t_camel_next = lk_u_of(camel_next);   // [UPPER, NEXT_LOWER]
    r.t_camel_prev = lk_u_of(camel_prev);   // [UPPER, PREV_LOWER]
    r.raw = lk_u_of(S | Y | prevY | camel_next | camel_prev);
    r.start = lk_u_of((T & prevS & nextA)                 // [TWITTER, PREV_SPACE, NEXT_ALPHA]
                      | (PE & prevS & nextAT & anA)       // [CHAR_PERIOD, PREV_SPACE, NEXT_AT, AFTER_NEXT_ALPHA]
                      | (AT & prevAN & nextAN)            // [CHAR_AT, PREV_ALPHA_NUM, NEXT_ALPHA_NUM]
                      | (CO & nextSL & anSL & prevA));    // [CHAR_COLON, NEXT_SLASH, AFTER_NEXT_SLASH, PREV_ALPHA]
    r.sym = lk_u_of(Y & nextS);       // [SYMBOL, NEXT_SPACE]
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// UTF-8 in byte space (fused UTF-8 ingest).  Positions are BYTES: a char lives at its lead byte; its continuation
// bytes carry the same ("smeared") code in the code planes and are marked in the plane C.  String starts, boundaries
// and the block mask are all expressed at lead-byte positions, so the rest of the pipeline is unchanged and the
// results are byte offsets into the UTF-8 buffer.  What changes is "the next char": it is no longer one position away.
//   PREV_X[i]  = X of the char that owns byte i-1            -> one shift of the smeared planes, as before
//   NL(X)[j]   = X at the first lead byte at or after j (a char has <= 3 continuation bytes)
//   NEXT_X[i]  = NL(X)[i+1],  AFTER_NEXT_X[i] = NL(NEXT_X at leads)[i+1]
//   E[i]       = lead i whose next lead starts a string (or is the end sentinel); E2 = E or "next lead is in E"
// For a word without continuation bytes every NL is the identity and the formulas collapse to lk_rules.
// ---------------------------------------------------------------------------------------------------------------
struct lk_halo_bytes {
    uint32_t prev;         // smeared code of byte base-1 (0 when it does not exist)
    lk_u64 next_codes;     // staging bytes of bytes base+64 .. base+71 (byte k of the word = byte base+64+k): the code at a lead
                           // byte, LK_CODE_CONT at a continuation byte
    uint32_t next_B;       // bit k: a string starts at byte base+64+k (k = 0..15)
};

// NL over one word: m0 = C, m1 = C & C>>1, m2 = m1 & C>>2 (shifts pulling from the next word); Xn = X of the next word
LATOK_HD lk_u64 lk_nl(lk_u64 X, lk_u64 Xn, lk_u64 m0, lk_u64 m1, lk_u64 m2) {
    return X | (m0 & ((X >> 1) | (Xn << 63))) | (m1 & ((X >> 2) | (Xn << 62))) | (m2 & ((X >> 3) | (Xn << 61)));
}

// Byte space, "smearing": a continuation byte carries the code of the char that owns it -- the nearest non-continuation
// byte at most 3 positions back; a 4th continuation byte in a row (malformed input) carries nothing.  Phase 1 of the tile
// kernel leaves codes at LEAD bytes only (0 at continuation bytes); the smear is mask arithmetic on the word's planes.
//
// lk_take_cont_plane: the continuation bytes of the word from the bit-sliced staging bytes (LK_CODE_CONT); the planes are
// left with the lead-only codes.
LATOK_HD lk_u64 lk_take_cont_plane(lk_u64 p[8]) {
    const lk_u64 C = p[7] & ~p[LK_BIT_SYMBOL];
    p[7] &= ~C;
    return C;
}
// lk_owner_before: the owner state in front of a word from the four staging bytes before it (codes4: byte 3 = position -1
// .. byte 0 = position -4; LK_CODE_CONT at continuation bytes):  *code = smeared code of byte -1, *left = how many more
// continuation bytes that char may still take.
LATOK_HD void lk_owner_before(uint32_t codes4, uint32_t* code, int* left) {
    const uint32_t b1 = codes4 >> 24, b2 = (codes4 >> 16) & 0xFFu, b3 = (codes4 >> 8) & 0xFFu, b4 = codes4 & 0xFFu;
    const bool c1 = b1 == LK_CODE_CONT, c2 = b2 == LK_CODE_CONT, c3 = b3 == LK_CODE_CONT, c4 = b4 == LK_CODE_CONT;
    *code = !c1 ? b1 : (!c2 ? b2 : (!c3 ? b3 : (!c4 ? b4 : 0u)));
    *left = !c1 ? 3 : (!c2 ? 2 : (!c3 ? 1 : 0));
}
// lk_smear_planes: planes PLANES (bit b = plane b) of p, lead-only on entry, smeared on return.  C = continuation bits of
// the word, (cin_code, cin_left) = owner state in front of it.
template <unsigned PLANES>
LATOK_HD void lk_smear_planes(lk_u64 p[8], lk_u64 C, uint32_t cin_code, int cin_left) {
    // leading continuation bytes of the word that the entering char still takes
    const lk_u64 nC = ~C;
    const int run = nC ? lk_ctz(nC) : 64;
    const int take = run < cin_left ? run : cin_left;              // 0..3
    const lk_u64 in_mask = (1ull << take) - 1ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; b < 8; ++b) {
        if (!((PLANES >> b) & 1u)) continue;
        lk_w x = lk_w_of(p[b]);               // the leads inside the word reach at most 3 continuation bytes each
        const lk_w Cw = lk_w_of(C);           // (on halves: see lk_w)
        x = x | (lk_w_shl<1>(x, 0u) & Cw);
        x = x | (lk_w_shl<1>(x, 0u) & Cw);
        x = x | (lk_w_shl<1>(x, 0u) & Cw);
        // the entering char: only the word's leading continuation bytes (which no lead of the word reaches), `take` of them
        p[b] = lk_u_of(x) | (((cin_code >> b) & 1u) ? in_mask : 0ull);
    }
}

// p = bit-sliced SMEARED code planes of the word, C = its continuation bytes.  *Ss_out = smeared SPACE plane (a byte of
// a SPACE char), which is what token stripping needs in byte space.
LATOK_HD lk_local lk_rules_bytes_general(const lk_u64 p[8], lk_u64 C, lk_halo_bytes h, lk_u64 B, lk_u64* Ss_out) {
    const lk_feat fs = lk_decode(p);                 // smeared features
    const lk_u64 Lead = ~C;
    // the next 8 byte positions as a mini word (only its low bits matter)
    const lk_u64 Tn = lk_transpose8(h.next_codes);
    const lk_u64 Cn = (Tn >> 56) & ~(Tn >> (8 * LK_BIT_SYMBOL)) & 0xFFull;      // LK_CODE_CONT: bit 7 without SYMBOL
    const lk_u64 Ln = ~Cn & 0xFFull;
#define LK_PN(b) ((Tn >> (8 * (b))) & 0xFFull)
    const lk_u64 Sn = LK_PN(0) & Ln, Lwn = LK_PN(2) & Ln, ANn = LK_PN(4) & Ln, An = LK_PN(5) & ~LK_PN(1) & Ln,
                 ATn = LK_PN(5) & LK_PN(6) & Ln, SLn = LK_PN(7) & LK_PN(5) & Ln;
#undef LK_PN
    const lk_u64 Bn = (lk_u64)(h.next_B & 0xFFFFu);
    const lk_u64 n0 = Cn, n1 = Cn & (Cn >> 1), n2 = n1 & (Cn >> 2);
    // masks of the main word
    const lk_u64 C1 = (C >> 1) | (Cn << 63), C2 = (C >> 2) | (Cn << 62);
    const lk_u64 m0 = C, m1 = C & C1, m2 = m1 & C2;
    // lead-only features of the word
    const lk_u64 S = fs.S & Lead, Lw = fs.L & Lead, AN = fs.AN & Lead, A = fs.A & Lead, AT = fs.AT & Lead, SL = fs.SL & Lead;
    const lk_u64 Y = fs.Y & Lead, U = fs.U & Lead, T = fs.T & Lead, PE = fs.PE & Lead, CO = fs.CO & Lead;

#define LK_NEXTB(X, Xn_) ((lk_nl(X, Xn_, m0, m1, m2) >> 1) | ((lk_nl(Xn_, 0, n0, n1, n2) & 1ull) << 63))
    // string ends: E = lead whose next lead is a string start
    const lk_u64 E = Lead & LK_NEXTB(B, Bn);
    const lk_u64 En = Ln & (lk_nl(Bn, 0, n0, n1, n2) >> 1);                  // valid for the low bits
    const lk_u64 E2 = E | (Lead & LK_NEXTB(E, En));
    const lk_u64 nB = ~B, nE = ~E, nE2 = ~E2;

    const lk_u64 nextS = LK_NEXTB(S, Sn) | E;
    const lk_u64 nextL = LK_NEXTB(Lw, Lwn) & nE, nextAN = LK_NEXTB(AN, ANn) & nE, nextAT = LK_NEXTB(AT, ATn) & nE;
    const lk_u64 nextA_raw = LK_NEXTB(A, An), nextSL_raw = LK_NEXTB(SL, SLn);
    const lk_u64 nextA = nextA_raw & nE, nextSL = nextSL_raw & nE;
    // AFTER_NEXT: the NEXT feature of the next char
    const lk_u64 nAn = Ln & (lk_nl(An, 0, n0, n1, n2) >> 1), nSLn = Ln & (lk_nl(SLn, 0, n0, n1, n2) >> 1);
    const lk_u64 anA = LK_NEXTB(nextA_raw & Lead, nAn) & nE2, anSL = LK_NEXTB(nextSL_raw & Lead, nSLn) & nE2;
#undef LK_NEXTB

    const lk_feat1 fp = lk_decode1(h.prev);
#define LK_PREVB(X) ((((fs.X) << 1) | (lk_u64)fp.X) & nB)
    const lk_u64 prevS = ((fs.S << 1) | (lk_u64)fp.S) | B;
    const lk_u64 prevY = LK_PREVB(Y), prevL = LK_PREVB(L), prevAN = LK_PREVB(AN), prevA = LK_PREVB(A);
#undef LK_PREVB

    lk_local r;
    r.S = S;
    r.t_space = S;
    r.t_sym = Y;
    r.t_prevsym = prevY & Lead;
    r.t_camel_next = U & nextL;
    r.t_camel_prev = U & prevL;
    r.raw = r.t_space | r.t_sym | r.t_prevsym | r.t_camel_next | r.t_camel_prev;
    r.start = (T & prevS & nextA) | (PE & prevS & nextAT & anA) | (AT & prevAN & nextAN) | (CO & nextSL & anSL & prevA);
    r.sym = Y & nextS;
    *Ss_out = fs.S;
    return r;
}

// The same rules, cheaper on everything but stray continuation bytes.  The four C_MASK terms start at '#' '$' '^' '@' '.'
// ':' and continue over '@' and '/': all ASCII (the reference's TWITTER / AT / COLON / SLASH / PERIOD flags are set for
// those single code points only, latok.h:15-22), and in well-formed UTF-8 the byte after an ASCII byte is a lead byte.
// So unless a continuation byte directly follows one of those chars ("weird": malformed input), their NEXT_* and
// AFTER_NEXT_* columns are plain shifts by one and two positions, exactly as in char space; only NEXT_LOWER (camel case:
// the upper-case char may be Cyrillic, Greek, ...), NEXT_SPACE (C_SYM: the symbol may be CJK punctuation, an emoji, ...)
// and the string ends need the "next lead byte" operator.  A word with a weird byte takes lk_rules_bytes_general; the
// CPU model checks that both agree wherever the fast form is taken.
LATOK_HD int lk_rules_bytes_weird(const lk_u64 p[8], lk_u64 C, lk_u64 next_codes) {
    const lk_u64 Tn = lk_transpose8(next_codes);
    const lk_u64 Cn = (Tn >> 56) & ~(Tn >> (8 * LK_BIT_SYMBOL)) & 0xFFull;
    // P = chars that start or continue a C_MASK term: SYMBOL with a sub-type (bits 5..7 != 0): # $ ^ @ : / .
    const lk_u64 P = p[LK_BIT_SYMBOL] & ~C & (p[5] | p[6] | p[7]);
    const lk_u64 Pn = (Tn >> (8 * LK_BIT_SYMBOL)) & ((Tn >> 40) | (Tn >> 48) | (Tn >> 56)) & ~Cn & 0xFFull;
    return ((P & ((C >> 1) | (Cn << 63))) | (Pn & (Cn >> 1) & 1ull)) != 0ull;
}
LATOK_HD lk_local lk_rules_bytes_fast(const lk_u64 p[8], lk_u64 C64, lk_halo_bytes h, lk_u64 B64, lk_u64* Ss_out) {
    // (on 32-bit halves throughout: see lk_w)
    lk_w q[8];
    for (int b = 0; b < 8; ++b) q[b] = lk_w_of(p[b]);
    const lk_feat_t<lk_w> fs = lk_decode_t<lk_w>(q);   // smeared features (planes 0, 1, 2, 4, 5 are smeared)
    const lk_w C = lk_w_of(C64), B = lk_w_of(B64);
    const lk_w Lead = ~C;
    const lk_u64 Tn = lk_transpose8(h.next_codes);
    const uint32_t Cn = (uint32_t)((Tn >> 56) & ~(Tn >> (8 * LK_BIT_SYMBOL)) & 0xFFull);
    const uint32_t Ln = ~Cn & 0xFFu;
#define LK_PN(b) ((uint32_t)(Tn >> (8 * (b))) & 0xFFu)
    const uint32_t Sn = LK_PN(0) & Ln, Lwn = LK_PN(2) & Ln, ANn = LK_PN(4) & Ln, An = LK_PN(5) & ~LK_PN(1) & Ln,
                   ATn = LK_PN(5) & LK_PN(6) & LK_PN(1) & Ln, SLn = LK_PN(7) & LK_PN(5) & Ln;
#undef LK_PN
    const uint32_t Bn = h.next_B & 0xFFFFu;
    const lk_w S = fs.S & Lead, Lw = fs.L & Lead, AN = fs.AN & Lead, A = fs.A & Lead, AT = fs.AT & Lead, SL = fs.SL & Lead;
    const lk_w Y = fs.Y & Lead, U = fs.U & Lead, T = fs.T & Lead, PE = fs.PE & Lead, CO = fs.CO & Lead;
    // X >> k with the first k positions of the next word (Xn_) entering at the top
#define LK_SH(X, Xn_, k) lk_w_shr<k>(X, Xn_)
    // "next lead byte" operator for lead-only planes: the value at the first lead after position i (at most 4 bytes on)
    const lk_w Q1 = LK_SH(C, Cn, 1), Q2 = Q1 & LK_SH(C, Cn, 2), Q3 = Q2 & LK_SH(C, Cn, 3);
#define LK_NEXTQ(X, Xn_) (LK_SH(X, Xn_, 1) | (Q1 & LK_SH(X, Xn_, 2)) | (Q2 & LK_SH(X, Xn_, 3)) | (Q3 & LK_SH(X, Xn_, 4)))
    const lk_w E = Lead & LK_NEXTQ(B, Bn);                          // string ends: the next lead is a string start
    const lk_w nextS = LK_NEXTQ(S, Sn) | E;
    const lk_w nextL = LK_NEXTQ(Lw, Lwn) & ~E;
#undef LK_NEXTQ
    // the ASCII-started terms: the next char is the next byte, the one after it the byte after that
    const lk_w nB1 = ~LK_SH(B, Bn, 1), nB2 = nB1 & ~LK_SH(B, Bn, 2);
    const lk_w nextA = LK_SH(A, An, 1) & nB1, nextAN = LK_SH(AN, ANn, 1) & nB1, nextAT = LK_SH(AT, ATn, 1) & nB1,
               nextSL = LK_SH(SL, SLn, 1) & nB1;
    const lk_w anA = LK_SH(A, An, 2) & nB2, anSL = LK_SH(SL, SLn, 2) & nB2;
#undef LK_SH
    const lk_w nB = ~B;
    const lk_feat1 fp = lk_decode1(h.prev);
#define LK_PREVB(X) (lk_w_shl<1>(fs.X, fp.X) & nB)
    const lk_w prevS = lk_w_shl<1>(fs.S, fp.S) | B;
    const lk_w prevY = LK_PREVB(Y), prevL = LK_PREVB(L), prevAN = LK_PREVB(AN), prevA = LK_PREVB(A);
#undef LK_PREVB
    const lk_w t_prevsym = prevY & Lead, camel_next = U & nextL, camel_prev = U & prevL;
    lk_local r;
    r.S = lk_u_of(S);
    r.t_space = r.S;
    r.t_sym = lk_u_of(Y);
    r.t_prevsym = lk_u_of(t_prevsym);
    r.t_camel_next = lk_u_of(camel_next);
    r.t_camel_prev = lk_u_of(camel_prev);
    r.raw = lk_u_of(S | Y | t_prevsym | camel_next | camel_prev);
    r.start = lk_u_of((T & prevS & nextA) | (PE & prevS & nextAT & anA) | (AT & prevAN & nextAN) | (CO & nextSL & anSL & prevA));
    r.sym = lk_u_of(Y & nextS);
    *Ss_out = lk_u_of(fs.S);
    return r;
}
LATOK_HD lk_local lk_rules_bytes(const lk_u64 p[8], lk_u64 C, lk_halo_bytes h, lk_u64 B, lk_u64* Ss_out) {
    if (lk_rules_bytes_weird(p, C, h.next_codes)) return lk_rules_bytes_general(p, C, h, B, Ss_out);
    return lk_rules_bytes_fast(p, C, h, B, Ss_out);
}

// ---------------------------------------------------------------------------------------------------------------
// Runtime rule tables: the reference's extension point (default_tokenizer.py:9-30) lets a user build other
// C_SPLIT / C_MASK / C_SYM matrices with build_combo_matrix (latok_utils.py:27-56) over the 25 feature columns.
// _combine_matrix_rows (latok.c:318-354) is "sum over rows of the product over the row's columns"; on a 0/1 feature
// matrix with fewer than 256 rows, "!= 0" of that sum is the OR over rows of the AND over the row's columns, which is
// what is evaluated here on whole 64-char planes.  A row is stored as the SET of its columns (bit k = column k of
// offsets.py:24-49); -1 padding simply is not in the set.
// ---------------------------------------------------------------------------------------------------------------
// (struct lk_rule_tables lives in split_code.h)

// the 25 planes of one word, selectable by a (wave-uniform) column id
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t lk_v32x32 __attribute__((ext_vector_type(32)));
struct lk_planes {
    lk_v32x32 lo, hi;   // kept as two register vectors so that a uniform dynamic index becomes an indexed register move
};
#define LK_PLANE_SET(P, k, v) do { const lk_u64 v_ = (v); (P).lo[k] = (uint32_t)v_; (P).hi[k] = (uint32_t)(v_ >> 32); } while (0)
#define LK_PLANE_GET(P, k) ((lk_u64)(P).lo[k] | ((lk_u64)(P).hi[k] << 32))
#define LK_PLANE_HALF(P, k, upper) ((upper) ? (P).hi[k] : (P).lo[k])
#else
struct lk_planes {
    lk_u64 v[32];
};
#define LK_PLANE_SET(P, k, val) ((P).v[k] = (val))
#define LK_PLANE_GET(P, k) ((P).v[k])
#define LK_PLANE_HALF(P, k, upper) ((uint32_t)((P).v[k] >> ((upper) ? 32 : 0)))
#endif

// 12-bit base feature word (bit i = reference column i, offsets.py:24-35) of ONE rule code (gen_unicode_tables.py:
// rule_code): bit0 S, bit1 Y, bit2 L, bit3 U, bit4 AN; Y=0: bit5 A, bit6 N; Y=1: bits 5..7 = symbol sub-type
LATOK_HD uint32_t lk_base_word1(uint32_t c) {
    const uint32_t y = (c >> 1) & 1u, b5 = (c >> 5) & 1u, b6 = (c >> 6) & 1u, b7 = (c >> 7) & 1u;
    return (b5 & ~y & 1u) | (((c >> 4) & 1u) << 1) | ((b6 & ~y & 1u) << 2) | (((c >> 2) & 1u) << 3) |
           (((c >> 3) & 1u) << 4) | ((c & 1u) << 5) | (y << 6) | ((b5 & y & ~b7 & 1u) << 7) | ((b5 & b6 & y) << 8) |
           ((b7 & ~b6 & ~b5 & 1u) << 9) | ((b7 & b5) << 10) | ((b7 & b6) << 11);
}

// all 25 columns of _gen_parse_matrix (latok.c:87-134) for one word, from the bit-sliced rule codes
LATOK_HD void lk_feature_planes(const lk_u64 p[8], lk_halo h, lk_u64 B, lk_u64 Bn, lk_planes& F) {
    const lk_u64 Y = p[LK_BIT_SYMBOL], nY = ~Y;
    lk_u64 b[12];
    b[0] = p[5] & nY;                 // ALPHA
    b[1] = p[LK_BIT_ALNUM];           // ALPHA_NUM
    b[2] = p[6] & nY;                 // NUM
    b[3] = p[LK_BIT_LOWER];
    b[4] = p[LK_BIT_UPPER];
    b[5] = p[LK_BIT_SPACE];
    b[6] = Y;                         // SYMBOL
    b[7] = p[5] & Y & ~p[7];          // TWITTER
    b[8] = p[5] & p[6] & Y;           // CHAR_AT
    b[9] = p[7] & ~p[6] & ~p[5];      // CHAR_COLON
    b[10] = p[7] & p[5];              // CHAR_SLASH
    b[11] = p[7] & p[6];              // CHAR_PERIOD
    const uint32_t wp = lk_base_word1(h.prev), w0 = lk_base_word1(h.next0), w1 = lk_base_word1(h.next1);
    const lk_u64 E = (B >> 1) | ((Bn & 1ull) << 63);
    const lk_u64 E2 = E | (B >> 2) | ((Bn & 3ull) << 62);
    const lk_u64 nB = ~B, nE = ~E, nE2 = ~E2;
#define LK_GPREV(i) ((((b[i]) << 1) | (lk_u64)((wp >> (i)) & 1u)) & nB)
#define LK_GNEXT(i) ((((b[i]) >> 1) | ((lk_u64)((w0 >> (i)) & 1u) << 63)) & nE)
#define LK_GANEXT(i) ((((b[i]) >> 2) | ((lk_u64)((w0 >> (i)) & 1u) << 62) | ((lk_u64)((w1 >> (i)) & 1u) << 63)) & nE2)
#pragma unroll
    for (int i = 0; i < 12; ++i) LK_PLANE_SET(F, i, b[i]);
    LK_PLANE_SET(F, 12, LK_GPREV(0));          // PREV_ALPHA
    LK_PLANE_SET(F, 13, LK_GNEXT(0));          // NEXT_ALPHA
    LK_PLANE_SET(F, 14, LK_GPREV(1));          // PREV_ALPHA_NUM
    LK_PLANE_SET(F, 15, LK_GNEXT(1));          // NEXT_ALPHA_NUM
    LK_PLANE_SET(F, 16, LK_GPREV(3));          // PREV_LOWER
    LK_PLANE_SET(F, 17, LK_GNEXT(3));          // NEXT_LOWER
    LK_PLANE_SET(F, 18, LK_GPREV(5) | B);      // PREV_SPACE: 1 at a string start (latok.c:116)
    LK_PLANE_SET(F, 19, LK_GNEXT(5) | E);      // NEXT_SPACE: 1 at a string end (latok.c:122-130)
    LK_PLANE_SET(F, 20, LK_GPREV(6));          // PREV_SYMBOL
    LK_PLANE_SET(F, 21, LK_GNEXT(8));          // NEXT_AT
    LK_PLANE_SET(F, 22, LK_GNEXT(10));         // NEXT_SLASH
    LK_PLANE_SET(F, 23, LK_GANEXT(0));         // AFTER_NEXT_ALPHA
    LK_PLANE_SET(F, 24, LK_GANEXT(10));        // AFTER_NEXT_SLASH
#undef LK_GPREV
#undef LK_GNEXT
#undef LK_GANEXT
}

// "_combine_matrix_rows(m.T, table) != 0" on one word
LATOK_HD lk_u64 lk_combine_rows(const lk_planes& F, const uint32_t* rows, int n_rows) {
    lk_u64 acc = 0;
    for (int r = 0; r < n_rows; ++r) {
        uint32_t m = rows[r];
        lk_u64 x = ~0ull;
        while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            x &= LK_PLANE_GET(F, k);
        }
        acc |= x;
    }
    return acc;
}

// number of rows of a table that hold at a position, bit-sliced: cnt[b] bit i = bit b of the count at char i.  This is the
// VALUE _combine_matrix_rows returns on a 0/1 matrix (latok.c:329-338: sum over rows of the product over the row's
// columns; uint8 wrap needs 256 rows, LK_MAX_RULE_ROWS is far below that).
#define LK_COUNT_BITS 6
LATOK_HD void lk_count_rows(const lk_planes& F, const uint32_t* rows, int n_rows, lk_u64 cnt[LK_COUNT_BITS]) {
    for (int b = 0; b < LK_COUNT_BITS; ++b) cnt[b] = 0;
    for (int r = 0; r < n_rows; ++r) {
        uint32_t m = rows[r];
        lk_u64 x = ~0ull;
        while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            x &= LK_PLANE_GET(F, k);
        }
        for (int b = 0; b < LK_COUNT_BITS; ++b) {   // ripple add of the row's plane
            const lk_u64 c = cnt[b] & x;
            cnt[b] ^= x;
            x = c;
        }
    }
}
// counts of C_SPLIT and C_SYM rows per position (split VALUES under run-time tables, default_tokenizer.py:121-132)
struct lk_rule_counts {
    lk_u64 split[LK_COUNT_BITS], sym[LK_COUNT_BITS];
};

LATOK_HD lk_local lk_rules_generic(const lk_u64 p[8], lk_halo h, lk_u64 B, lk_u64 Bn, const lk_rule_tables& R,
                                   lk_rule_counts* counts = nullptr) {
    lk_planes F;
    lk_feature_planes(p, h, B, Bn, F);
    lk_local r;
    r.S = p[LK_BIT_SPACE];
    r.raw = lk_combine_rows(F, R.row[0], R.n_rows[0]);
    r.start = lk_combine_rows(F, R.row[1], R.n_rows[1]);
    r.sym = lk_combine_rows(F, R.row[2], R.n_rows[2]);
    r.t_space = r.t_sym = r.t_prevsym = r.t_camel_next = r.t_camel_prev = 0;   // per-term planes exist for the default tables only
    if (counts) {
        lk_count_rows(F, R.row[0], R.n_rows[0], counts->split);
        lk_count_rows(F, R.row[2], R.n_rows[2], counts->sym);
    }
    return r;
}

// Run-time rule tables in BYTE space (UTF-8 input, positions are bytes): the 25 columns of a char live at its lead byte.
// p = rule-code planes with planes 0, 1, 2, 4, 5 smeared over the continuation bytes (lk_smear_planes), the others lead
// only; C = continuation bytes; h as for lk_rules_bytes.  PREV_* columns read the smeared planes one byte back, NEXT_* /
// AFTER_NEXT_* use the "next lead byte" operator -- for every column: a caller's row may put any column beside any char.
LATOK_HD void lk_feature_planes_bytes(const lk_u64 p[8], lk_u64 C, lk_halo_bytes h, lk_u64 B, lk_planes& F, lk_u64* E_out) {
    const lk_u64 Lead = ~C;
    const lk_u64 Y = p[LK_BIT_SYMBOL], nY = ~Y;
    lk_u64 b[12];
    b[0] = p[5] & nY & Lead;                 // ALPHA
    b[1] = p[LK_BIT_ALNUM] & Lead;           // ALPHA_NUM
    b[2] = p[6] & nY & Lead;                 // NUM
    b[3] = p[LK_BIT_LOWER] & Lead;
    b[4] = p[LK_BIT_UPPER] & Lead;
    b[5] = p[LK_BIT_SPACE] & Lead;
    b[6] = Y & Lead;                         // SYMBOL
    b[7] = p[5] & Y & ~p[7] & Lead;          // TWITTER
    b[8] = p[5] & p[6] & Y & Lead;           // CHAR_AT
    b[9] = p[7] & ~p[6] & ~p[5] & Lead;      // CHAR_COLON
    b[10] = p[7] & p[5] & Lead;              // CHAR_SLASH
    b[11] = p[7] & p[6] & Lead;              // CHAR_PERIOD
    // the next 8 byte positions as a mini word
    const lk_u64 Tn = lk_transpose8(h.next_codes);
    const lk_u64 Cn = (Tn >> 56) & ~(Tn >> (8 * LK_BIT_SYMBOL)) & 0xFFull;
    const lk_u64 Ln = ~Cn & 0xFFull;
#define LK_PN(k) ((Tn >> (8 * (k))) & 0xFFull)
    const lk_u64 Yn = LK_PN(1);
    const lk_u64 An = LK_PN(5) & ~Yn & Ln, ANn = LK_PN(4) & Ln, Lwn = LK_PN(2) & Ln, Sn = LK_PN(0) & Ln,
                 ATn = LK_PN(5) & LK_PN(6) & Yn & Ln, SLn = LK_PN(7) & LK_PN(5) & Ln;
#undef LK_PN
    const lk_u64 Bn = (lk_u64)(h.next_B & 0xFFFFu);
#define LK_SH(X, Xn_, k) (((X) >> (k)) | ((Xn_) << (64 - (k))))
    const lk_u64 Q1 = LK_SH(C, Cn, 1), Q2 = Q1 & LK_SH(C, Cn, 2), Q3 = Q2 & LK_SH(C, Cn, 3);
#define LK_NEXTQ(X, Xn_) (LK_SH(X, Xn_, 1) | (Q1 & LK_SH(X, Xn_, 2)) | (Q2 & LK_SH(X, Xn_, 3)) | (Q3 & LK_SH(X, Xn_, 4)))
    // the same operator inside the mini word (what lies behind its 8 bytes does not reach the main word's columns)
    const lk_u64 n1 = Cn >> 1, n2 = n1 & (Cn >> 2), n3 = n2 & (Cn >> 3);
#define LK_NEXTQ_MINI(Xn_) (Ln & (((Xn_) >> 1) | (n1 & ((Xn_) >> 2)) | (n2 & ((Xn_) >> 3)) | (n3 & ((Xn_) >> 4))))
    const lk_u64 E = Lead & LK_NEXTQ(B, Bn);                 // string ends: the next lead is a string start
    const lk_u64 En = LK_NEXTQ_MINI(Bn);
    const lk_u64 E2 = E | (Lead & LK_NEXTQ(E, En));
    const lk_u64 nB = ~B, nE = ~E, nE2 = ~E2;
    const lk_u64 nextA_raw = LK_NEXTQ(b[0], An), nextSL_raw = LK_NEXTQ(b[10], SLn);
    const lk_u64 anA = LK_NEXTQ(nextA_raw & Lead, LK_NEXTQ_MINI(An)) & nE2, anSL = LK_NEXTQ(nextSL_raw & Lead, LK_NEXTQ_MINI(SLn)) & nE2;
    const uint32_t wp = lk_base_word1(h.prev);
    const lk_u64 sA = p[5] & nY;                              // smeared ALPHA (planes 5 and 1 are smeared)
#define LK_PREVQ(S_, i) (((((S_)) << 1) | (lk_u64)((wp >> (i)) & 1u)) & nB & Lead)
#pragma unroll
    for (int i = 0; i < 12; ++i) LK_PLANE_SET(F, i, b[i]);
    LK_PLANE_SET(F, 12, LK_PREVQ(sA, 0));                               // PREV_ALPHA
    LK_PLANE_SET(F, 13, nextA_raw & nE & Lead);                         // NEXT_ALPHA
    LK_PLANE_SET(F, 14, LK_PREVQ(p[LK_BIT_ALNUM], 1));                  // PREV_ALPHA_NUM
    LK_PLANE_SET(F, 15, LK_NEXTQ(b[1], ANn) & nE & Lead);               // NEXT_ALPHA_NUM
    LK_PLANE_SET(F, 16, LK_PREVQ(p[LK_BIT_LOWER], 3));                  // PREV_LOWER
    LK_PLANE_SET(F, 17, LK_NEXTQ(b[3], Lwn) & nE & Lead);               // NEXT_LOWER
    LK_PLANE_SET(F, 18, (LK_PREVQ(p[LK_BIT_SPACE], 5) | B) & Lead);     // PREV_SPACE: 1 at a string start
    LK_PLANE_SET(F, 19, ((LK_NEXTQ(b[5], Sn) & nE) | E) & Lead);        // NEXT_SPACE: 1 at a string end
    LK_PLANE_SET(F, 20, LK_PREVQ(Y, 6));                                // PREV_SYMBOL
    LK_PLANE_SET(F, 21, LK_NEXTQ(b[8], ATn) & nE & Lead);               // NEXT_AT
    LK_PLANE_SET(F, 22, nextSL_raw & nE & Lead);                        // NEXT_SLASH
    LK_PLANE_SET(F, 23, anA & Lead);                                    // AFTER_NEXT_ALPHA
    LK_PLANE_SET(F, 24, anSL & Lead);                                   // AFTER_NEXT_SLASH
#undef LK_PREVQ
#undef LK_NEXTQ_MINI
#undef LK_NEXTQ
#undef LK_SH
    *E_out = E;
}
LATOK_HD lk_local lk_rules_generic_bytes(const lk_u64 p[8], lk_u64 C, lk_halo_bytes h, lk_u64 B, const lk_rule_tables& R,
                                         lk_u64* Ss_out) {
    lk_planes F;
    lk_u64 E;
    lk_feature_planes_bytes(p, C, h, B, F, &E);
    const lk_u64 Lead = ~C;
    lk_local r;
    r.S = p[LK_BIT_SPACE] & Lead;
    // an empty row set is "all ones" per row, and a table may have rows that no lead byte... keep everything at lead bytes
    r.raw = lk_combine_rows(F, R.row[0], R.n_rows[0]) & Lead;
    r.start = lk_combine_rows(F, R.row[1], R.n_rows[1]) & Lead;
    r.sym = lk_combine_rows(F, R.row[2], R.n_rows[2]) & Lead;
    r.t_space = r.t_sym = r.t_prevsym = r.t_camel_next = r.t_camel_prev = 0;
    *Ss_out = p[LK_BIT_SPACE];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Block mask, forward half: which block-closing events zero their block.
//
// Events inside a word, in position order; at one position: (1) B: the previous string ends (a virtual space that
// closes its last block, latok.c:240-244) and the pending-start queue q resets; (2) St: q++ ; (3) S: the space closes
// the block before it.  A closing event with q > 0 zeroes its block and consumes one start (latok.c:227-236).
// The word is first simulated with q_in = 0; `idle` are the closings (up to and including the first B) that found
// q == 0: a non-zero q_in = r is consumed by exactly the first r of them, so Z(r) = Z(0) | first_r(idle) and
// q_out(r) = hasB ? q0 : q0 + max(r - popc(idle), 0).
// ---------------------------------------------------------------------------------------------------------------
struct lk_fwd {
    lk_u64 zs, zb;     // zeroing closings with q_in = 0: spaces / string-ends (bit at the *next* string's start)
    lk_u64 idle;       // idle closings before-or-at the first B (bit at first B position = that B closing)
    lk_u64 firstB;     // lowest B bit (0 if none)
    int q0;            // q_out with q_in = 0
    int n_idle;        // popc(idle)
    int head_starts;   // starts strictly before the first closing event of the word
    int has_closing;   // (S | B) != 0
};

LATOK_HD lk_fwd lk_forward(lk_u64 St, lk_u64 S, lk_u64 B) {
    lk_fwd r;
    const lk_u64 closing = S | B;
    const lk_u64 Pf = ~closing;
    r.firstB = B & (~B + 1ull);
    r.has_closing = closing != 0;
    const lk_u64 first_closing = closing & (~closing + 1ull);
    // a start sitting ON a space is consumed by that very space (only possible through the generic
    // _gen_block_mask surface; the tokenizer's own starts are never spaces)
    r.head_starts = lk_popc(first_closing ? (St & (first_closing - 1ull)) : St) +
                    (int)((St & S & ~B & first_closing) != 0);

    // fast path: a start's carry ripples (binary add) through the non-closing positions above it and lands on the
    // first closing.  Valid iff no two starts share a block, checked by counting (a merged ripple loses a start).
    // A start ON a space generates at its own position (its space consumes it); start+space+string-start on one
    // char is left to the exact loop.
    const lk_u64 on_space = St & S;
    const lk_u64 gen = ((St & ~S) << 1) | on_space;
    const int g63 = (int)((St & ~S) >> 63);
    const lk_u64 sum = Pf + gen;
    const int cout = sum < Pf;
    const lk_u64 reached = (sum ^ Pf) & closing;
    if ((on_space & B) == 0 && lk_popc(St) == lk_popc(reached) + cout + g63) {
        r.zb = reached & B;
        r.zs = reached & ~B;
        const lk_u64 below = r.firstB ? (r.firstB - 1ull) : ~0ull;
        r.idle = ((S & ~B & below) | r.firstB) & ~reached;
        r.q0 = cout + g63;
    } else {
        // exact sequential simulation (a block with >= 2 starts: latok's "spill-over" case)
        lk_u64 zs = 0, zb = 0, idle = 0, M = St | closing;
        int q = 0, seenB = 0;
        while (M) {
            const lk_u64 bit = M & (~M + 1ull);
            M ^= bit;
            if (B & bit) {
                if (q > 0) zb |= bit; else if (!seenB) idle |= bit;
                q = 0;
                seenB = 1;
            }
            if (St & bit) ++q;
            if (S & bit) {
                // (a space that is also a string's first char closes an EMPTY block: it consumes, nothing to clear)
                if (q > 0) { if (!(B & bit)) zs |= bit; --q; } else if (!seenB) idle |= bit;
            }
        }
        r.zs = zs; r.zb = zb; r.idle = idle; r.q0 = q;
    }
    r.n_idle = lk_popc(r.idle);
    return r;
}

// q transfer function of a word / tile: f(q) = max(q + a, b).  a == LK_NEG_INF encodes "constant b" (a string start
// inside resets the queue).  Composition (first f1 then f2) stays in the family.
#define LK_NEG_INF (-(1 << 29))
struct lk_qfn {
    int a, b;
};
LATOK_HD lk_qfn lk_qfn_of(const lk_fwd& w) {
    lk_qfn f;
    f.a = w.firstB ? LK_NEG_INF : (w.q0 - w.n_idle);
    f.b = w.q0;
    return f;
}
LATOK_HD lk_qfn lk_qfn_then(lk_qfn f1, lk_qfn f2) {
    lk_qfn f;
    int a = f1.a + f2.a;
    f.a = a < LK_NEG_INF ? LK_NEG_INF : a;
    int c = f1.b + f2.a;
    f.b = c > f2.b ? c : f2.b;
    return f;
}
LATOK_HD int lk_qfn_apply(lk_qfn f, int q) {
    int c = q + f.a;
    return c > f.b ? c : f.b;
}

// add the closings consumed by r extra pending starts (r > 0): the first r idle closings
LATOK_HD void lk_apply_extra(lk_fwd& w, int r) {
    lk_u64 idle = w.idle, take = 0;
    while (r > 0 && idle) {
        const lk_u64 bit = idle & (~idle + 1ull);
        idle ^= bit;
        take |= bit;
        --r;
    }
    w.zb |= take & w.firstB;
    w.zs |= take & ~w.firstB;
}

// ---------------------------------------------------------------------------------------------------------------
// Block mask, backward half: every zeroing closing at position i clears the run of non-delimiter chars below it
// (down to, and including, a string's first char).  Done as one binary add on the bit-reversed word; the carry that
// leaves the word at the bottom continues in the previous word.
//   gen_top : bit 63 generator = the NEXT word's closing at its position 0 zeroes (Zall_next & 1)
//   cin     : a fill that started in a later word arrives at position 63
// returns the cleared positions (B positions may be included: the caller ORs B back, splits[0] = 1).
// ---------------------------------------------------------------------------------------------------------------
struct lk_bwd {
    lk_u64 xr, gr;  // reversed propagate mask / generators
    int g, p;       // carry-out with cin = 0; word is all-propagate
};
LATOK_HD lk_bwd lk_backward_prepare(lk_u64 zall, lk_u64 S, lk_u64 B, int gen_top) {
    lk_bwd r;
    const lk_u64 G = (zall >> 1) | ((lk_u64)(gen_top & 1) << 63);
    r.xr = lk_rev(~(S | B));
    r.gr = lk_rev(G);
    const lk_u64 sum = r.xr + r.gr;
    r.g = sum < r.xr;
    r.p = r.xr == ~0ull;
    return r;
}
LATOK_HD lk_u64 lk_backward_fill(const lk_bwd& w, int cin, lk_u64 S) {
    const lk_u64 sum = w.xr + w.gr + (lk_u64)(cin & 1);
    return lk_rev(sum ^ w.xr) & ~S;
}

#endif
