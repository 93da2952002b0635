// split_code.h -- the 8-bit per-character "split code" that the fused kernel bit-slices.
//
// The reference derives 12 base features per character from a Unicode flag word
// (reference latok/core/src/latok/latok.c:87-98).  Only 17 distinct feature combinations exist over all code points,
// and the split rules (reference latok/core/default_tokenizer.py:39-102) never look at NUM except through ALPHA_NUM,
// which leaves 16 behaviours.  They are encoded in one byte so that the planes the rules need most are single bits:
//
//   bit0 SPACE   bit1 SYMBOL   bit2 LOWER   bit3 UPPER   bit4 ALPHA_NUM
//   bit5 ALPHA                      (when SYMBOL = 0; ALPHA implies ALPHA_NUM, and SYMBOL excludes both)
//   bit5..7 = sub-type of a SYMBOL  (when SYMBOL = 1):  0 plain, 1 twitter-only (# $ ^), 3 '@' (twitter + at),
//                                                       4 ':', 5 '/', 6 '.'
//
// tools/gen_unicode_tables.py emits kClassCode[] with exactly this layout; lane_math.h:lk_decode() undoes it.
#ifndef LATOK_SPLIT_CODE_H
#define LATOK_SPLIT_CODE_H
#include <stdint.h>

#define LK_BIT_SPACE 0
#define LK_BIT_SYMBOL 1
#define LK_BIT_LOWER 2
#define LK_BIT_UPPER 3
#define LK_BIT_ALNUM 4

// Byte space (UTF-8 input): the staging byte of a CONTINUATION byte holds this marker -- bit 7 without SYMBOL, which no
// split code and no rule code has -- so that the continuation plane of a word falls out of the bit-slicing (plane 7 & ~plane 1)
#define LK_CODE_CONT 0x80u
// Byte space has its own two-stage class table, cut where UTF-8 cuts a char: stage 1 by cp >> 6 (uint16: offset of a 64-entry
// stage-2 block), stage 2 by the last byte's payload (lane_math.h: lk_lead_index; api.cpp builds it from the generated tables)
#define LK_B6_SHIFT 6
#define LK_B6_STAGE1_LEN ((0x110000 >> LK_B6_SHIFT) + 1)   /* 17409; the last entry = the block of cp >= 0x110000 (all codes 0) */

// Rule code (runtime rule tables): the split code with NUM added in bit 6 for non-symbols, so that all 12 base
// features can be decoded from the byte (tools/gen_unicode_tables.py:rule_code, lane_math.h:lk_feature_planes).
// Runtime rule tables: each row of C_SPLIT / C_MASK / C_SYM is the SET of feature columns it multiplies
// (bit k = column k of reference latok/core/offsets.py:24-49).
#define LK_N_FEATURES 25
#define LK_MAX_RULE_ROWS 32    /* rows per table (the reference's own tables have 5 / 4 / 1); sets the size of the kernel argument only */
struct lk_rule_tables {
    uint32_t row[3][LK_MAX_RULE_ROWS];   // [0] C_SPLIT, [1] C_MASK, [2] C_SYM
    int32_t n_rows[3];
};

#endif
