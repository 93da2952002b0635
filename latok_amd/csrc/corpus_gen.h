// corpus_gen.h -- deterministic synthetic corpora (C2 ASCII word soup, C3 mixed Unicode).
//
// Counter-based: string `sid` of corpus (`seed`, `model`) is a pure function of (seed, model, sid, length), so the
// host (gcc / g++) and the device (hipcc) produce identical code points without any PCIe traffic.  The same header is
// compiled three ways: into the HIP library (device kernel + host twin) and into tests' host helper.
//
// There is no reference counterpart: latok ships no corpus (its timing script reads a private CSV,
// reference scripts/timing/time_tokenizer.py:25-40).  The char model is built to hit every split rule of
// reference latok/core/default_tokenizer.py:39-102 (camelCase, symbols, URL / e-mail / twitter starts, whitespace).
#ifndef LATOK_CORPUS_GEN_H
#define LATOK_CORPUS_GEN_H
#include <stdint.h>

#if defined(__HIPCC__)
#define LATOK_HD __host__ __device__ inline
#else
#define LATOK_HD static inline
#endif

#ifndef LATOK_CORPUS_ASCII
#define LATOK_CORPUS_ASCII 0
#define LATOK_CORPUS_UNICODE 1
#endif

LATOK_HD uint64_t latok_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// length of string `sid`: uniform in [lo, hi]
LATOK_HD int64_t latok_corpus_length(uint64_t seed, uint64_t sid, int64_t lo, int64_t hi) {
    uint64_t h = latok_mix64(seed ^ latok_mix64(sid * 2 + 1));
    return lo + (int64_t)(h % (uint64_t)(hi - lo + 1));
}

typedef struct {
    uint64_t state;
    uint32_t* out;
    int64_t pos, len;
} latok_gen_t;

LATOK_HD uint32_t latok_gen_next(latok_gen_t* g) {  // 32 fresh bits
    g->state += 0x9E3779B97F4A7C15ull;
    uint64_t z = g->state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
LATOK_HD uint32_t latok_gen_below(latok_gen_t* g, uint32_t n) { return (uint32_t)(((uint64_t)latok_gen_next(g) * n) >> 32); }
LATOK_HD void latok_gen_emit(latok_gen_t* g, uint32_t cp) {
    if (g->pos < g->len) g->out[g->pos] = cp;
    g->pos++;
}
LATOK_HD int latok_gen_wordlen(latok_gen_t* g) {  // geometric, mean ~5, capped at 14
    int n = 1;
    while (n < 14 && latok_gen_below(g, 5) != 0) ++n;
    return n;
}
LATOK_HD void latok_gen_lower(latok_gen_t* g, int n) {
    for (int i = 0; i < n; ++i) latok_gen_emit(g, 'a' + latok_gen_below(g, 26));
}
LATOK_HD void latok_gen_str(latok_gen_t* g, const char* s) {
    for (; *s; ++s) latok_gen_emit(g, (uint32_t)(unsigned char)*s);
}

LATOK_HD void latok_gen_ascii_word(latok_gen_t* g) {
    uint32_t r = latok_gen_below(g, 100);
    int n = latok_gen_wordlen(g);
    if (r < 2) {  // 2 %: special tokens (URL, e-mail, twitter forms, digits)
        switch (latok_gen_below(g, 6)) {
            case 0: latok_gen_str(g, "http://"); latok_gen_lower(g, 2 + n % 4); latok_gen_emit(g, '.');
                    latok_gen_lower(g, 2); latok_gen_emit(g, '/'); latok_gen_lower(g, n); break;
            case 1: latok_gen_lower(g, n); latok_gen_emit(g, '@'); latok_gen_lower(g, 3 + n % 3);
                    latok_gen_str(g, ".com"); break;
            case 2: latok_gen_emit(g, '#'); latok_gen_lower(g, n); break;
            case 3: latok_gen_emit(g, '@'); latok_gen_lower(g, n); break;
            case 4: latok_gen_str(g, ".@"); latok_gen_lower(g, n); break;
            default: for (int i = 0, k = 1 + n % 4; i < k; ++i) latok_gen_emit(g, '0' + latok_gen_below(g, 10)); break;
        }
    } else if (r < 14) {  // 12 %: Capitalised
        latok_gen_emit(g, 'A' + latok_gen_below(g, 26)); latok_gen_lower(g, n - 1);
    } else if (r < 18) {  // 4 %: camelCase
        latok_gen_lower(g, n); latok_gen_emit(g, 'A' + latok_gen_below(g, 26)); latok_gen_lower(g, 1 + n % 5);
    } else if (r < 21) {  // 3 %: ALLCAPS
        for (int i = 0; i < n; ++i) latok_gen_emit(g, 'A' + latok_gen_below(g, 26));
    } else {
        latok_gen_lower(g, n);
    }
}
LATOK_HD void latok_gen_separator(latok_gen_t* g) {
    uint32_t r = latok_gen_below(g, 100);
    if (r < 85) latok_gen_emit(g, ' ');
    else if (r < 91) latok_gen_str(g, ", ");
    else if (r < 96) latok_gen_str(g, ". ");
    else if (r < 97) latok_gen_emit(g, '\n');
    else if (r < 98) latok_gen_emit(g, '\t');
    else if (r < 99) latok_gen_str(g, "! ");
    else latok_gen_str(g, "? ");
}

LATOK_HD void latok_gen_unicode_word(latok_gen_t* g) {
    uint32_t r = latok_gen_below(g, 100);
    int n = latok_gen_wordlen(g);
    if (r < 35) {  // CJK ideographs / kana, sometimes closed by an ideographic comma / full stop
        uint32_t kind = latok_gen_below(g, 4);
        for (int i = 0; i < n; ++i) {
            if (kind < 2) latok_gen_emit(g, 0x4E00 + latok_gen_below(g, 0x9FEF - 0x4E00 + 1));
            else if (kind == 2) latok_gen_emit(g, 0x3041 + latok_gen_below(g, 0x3096 - 0x3041 + 1));
            else latok_gen_emit(g, 0x30A1 + latok_gen_below(g, 0x30FA - 0x30A1 + 1));
        }
        if (latok_gen_below(g, 4) == 0) latok_gen_emit(g, 0x3001 + latok_gen_below(g, 2));
    } else if (r < 65) {
        latok_gen_ascii_word(g);
    } else if (r < 75) {  // emoji
        int k = 1 + n % 3;
        for (int i = 0; i < k; ++i)
            latok_gen_emit(g, latok_gen_below(g, 2) ? 0x1F300 + latok_gen_below(g, 0x1F64F - 0x1F300 + 1)
                                                     : 0x1F900 + latok_gen_below(g, 0x1F9FF - 0x1F900 + 1));
    } else if (r < 85) {  // Latin letters each possibly followed by a combining mark
        for (int i = 0; i < n; ++i) {
            latok_gen_emit(g, 'a' + latok_gen_below(g, 26));
            if (latok_gen_below(g, 3) == 0) latok_gen_emit(g, 0x0300 + latok_gen_below(g, 0x70));
        }
    } else if (r < 95) {  // Cyrillic / Greek cased words (non-ASCII UPPER/LOWER, camelCase rule)
        int greek = (int)latok_gen_below(g, 2);
        uint32_t up = greek ? 0x0391 : 0x0410, lo = greek ? 0x03B1 : 0x0430, span = greek ? 17 : 32;
        int style = (int)latok_gen_below(g, 4);
        for (int i = 0; i < n; ++i) {
            int upper = (style == 0 && i == 0) || (style == 1 && i == n / 2 && i > 0) || style == 2;
            latok_gen_emit(g, (upper ? up : lo) + latok_gen_below(g, span));
        }
    } else {  // digits: ASCII, Arabic-Indic, circled numbers
        uint32_t kind = latok_gen_below(g, 3);
        int k = 1 + n % 4;
        for (int i = 0; i < k; ++i) {
            if (kind == 0) latok_gen_emit(g, '0' + latok_gen_below(g, 10));
            else if (kind == 1) latok_gen_emit(g, 0x0660 + latok_gen_below(g, 10));
            else latok_gen_emit(g, 0x2460 + latok_gen_below(g, 20));
        }
    }
}

// Fill out[0..len) with string `sid`.
LATOK_HD void latok_corpus_string(uint64_t seed, int model, uint64_t sid, uint32_t* out, int64_t len) {
    latok_gen_t g;
    g.state = latok_mix64(seed ^ latok_mix64(sid * 2));
    g.out = out;
    g.pos = 0;
    g.len = len;
    while (g.pos < len) {
        if (model == LATOK_CORPUS_UNICODE) latok_gen_unicode_word(&g); else latok_gen_ascii_word(&g);
        if (g.pos < len) latok_gen_separator(&g);
    }
}

// UTF-8 encoded size of one code point (surrogates / out-of-range counted as 3 bytes, like U+FFFD)
LATOK_HD int latok_utf8_len(uint32_t cp) {
    if (cp < 0x80) return 1;
    if (cp < 0x800) return 2;
    if (cp < 0x10000) return 3;
    if (cp < 0x110000) return 4;
    return 3;
}
#endif
