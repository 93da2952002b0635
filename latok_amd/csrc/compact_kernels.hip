// compact_kernels.hip -- everything tokenize() / featurize() do AFTER the split mask, on the device and parallel over
// the packed buffer (not over strings, so one 1 M-char document is as parallel as 8 000 tweets):
//
//   boundary offsets   np.nonzero(splits)[0] per string                      reference default_tokenizer.py:148
//   token spans        slice between consecutive boundaries, strip, drop ''   reference default_tokenizer.py:149-158
//   token features     per-token sums of the 25 matrix columns (featurize)    reference default_tokenizer.py:163-191
//
// Inputs are the two bitmasks the tile kernel writes (boundary bits, SPACE bits; bit i = packed char i) and row_off.
// Because strings are contiguous and ordered in the packed buffer, "all offsets of string 0, then string 1, ..." is
// simply position order, so the output index of an item is the global rank of its bit:
//   pass 1  one thread per 64-bit word: how many items start in the word (-> device-wide exclusive scan = word ranks)
//   pass 2  one thread per string: count = rank(end) - rank(start), O(1)
//   pass 3  one thread per word: scatter its items at word rank + position inside the word
// A token is kept iff it contains a non-SPACE char; its extent needs the next boundary bit, which normally sits in the
// same or the next word (a thread follows the mask forward only for its own items).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace latok {

__device__ __forceinline__ uint64_t low_mask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// first set bit of `bits` at a position in [from, limit), or `limit`
__device__ __forceinline__ int64_t next_set_bit(const uint64_t* __restrict__ bits, int64_t from, int64_t limit) {
    for (int64_t w = from >> 6; from < limit; ++w) {
        uint64_t x = bits[w];
        const int64_t base = w << 6;
        if (from > base) x &= ~0ull << (from - base);
        if (x) {
            const int64_t p = base + __builtin_ctzll(x);
            return p < limit ? p : limit;
        }
        from = base + 64;
    }
    return limit;
}
// first position in [from, to) whose bit is 0, or `to`
__device__ __forceinline__ int64_t next_zero_bit(const uint64_t* __restrict__ bits, int64_t from, int64_t to) {
    for (int64_t w = from >> 6; from < to; ++w) {
        uint64_t x = ~bits[w];
        const int64_t base = w << 6;
        if (from > base) x &= ~0ull << (from - base);
        if (x) {
            const int64_t p = base + __builtin_ctzll(x);
            return p < to ? p : to;
        }
        from = base + 64;
    }
    return to;
}
// last position in [from, to) whose bit is 0, plus one; `from` if none
__device__ __forceinline__ int64_t prev_zero_end(const uint64_t* __restrict__ bits, int64_t from, int64_t to) {
    for (int64_t w = (to - 1) >> 6; to > from; --w) {
        uint64_t x = ~bits[w];
        const int64_t base = w << 6;
        if (to < base + 64) x &= (1ull << (to - base)) - 1ull;
        if (x) {
            const int64_t p = base + 63 - __builtin_clzll(x);
            return p >= from ? p + 1 : from;
        }
        to = base;
    }
    return from;
}

// the token that starts at boundary p: [p, e) with e = next boundary (every string start is one) or the end of the
// batch; stripped extent [a2, e2); kept iff a2 < e
struct Token {
    int64_t e, a2, e2;
    bool kept;
};
__device__ __forceinline__ Token token_at(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space,
                                          int64_t p, int64_t total) {
    Token t;
    t.e = next_set_bit(bits, p + 1, total);
    t.a2 = next_zero_bit(space, p, t.e);
    t.kept = t.a2 < t.e;
    t.e2 = t.kept ? prev_zero_end(space, t.a2, t.e) : t.a2;
    return t;
}

// ---- pass 1 --------------------------------------------------------------------------------------------------------
// SPANS = false: items = boundary bits.  SPANS = true: items = boundaries whose token is kept; the kept-mask word is
// stored for the later passes.
template <bool SPANS>
__global__ void k_word_counts(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space, int64_t n_words,
                              int64_t total, uint64_t* __restrict__ kept_out, int64_t* __restrict__ cnt) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t x = bits[w];
    if (!SPANS) {
        cnt[w] = __popcll(x);
        return;
    }
    uint64_t kept = 0;
    while (x) {
        const int b = __builtin_ctzll(x);
        x &= x - 1;
        if (token_at(bits, space, (w << 6) + b, total).kept) kept |= 1ull << b;
    }
    kept_out[w] = kept;
    cnt[w] = __popcll(kept);
}

// ---- pass 2: items per string = rank(row_off[s+1]) - rank(row_off[s]) ------------------------------------------------
__device__ __forceinline__ int64_t rank_at(const uint64_t* __restrict__ mask, const int64_t* __restrict__ word_rank,
                                           int64_t x, int64_t total, int64_t n_items) {
    if (x >= total) return n_items;
    return word_rank[x >> 6] + __popcll(mask[x >> 6] & low_mask((int)(x & 63)));
}
__global__ void k_string_counts(const uint64_t* __restrict__ mask, const int64_t* __restrict__ word_rank,
                                const int64_t* __restrict__ row_off, int64_t n_str, int64_t total,
                                const int64_t* __restrict__ n_items, int64_t* __restrict__ counts) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_str) return;
    const int64_t n = *n_items;
    counts[s] = rank_at(mask, word_rank, row_off[s + 1], total, n) - rank_at(mask, word_rank, row_off[s], total, n);
}

// ---- pass 3 --------------------------------------------------------------------------------------------------------
// index of the string that contains packed position p (the last string that starts at or before p)
__device__ __forceinline__ int64_t string_of(const int64_t* __restrict__ row_off, int64_t n_str, int64_t p) {
    int64_t lo = 0, hi = n_str;   // answer s in [lo, hi): row_off[s] <= p < row_off[s+1] (skipping empty strings)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// KIND 0: offsets[k] = p - start of its string.   KIND 1: spans[2k..] = stripped extent.
// KIND 2: spans[4k..] = {raw start, raw end, stripped start, stripped end}, tok_sid[k] = string id (featurize pass).
template <int KIND>
__global__ void k_word_scatter(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space,
                               const uint64_t* __restrict__ item_mask, const int64_t* __restrict__ word_rank,
                               int64_t n_words, int64_t total, const int64_t* __restrict__ row_off, int64_t n_str,
                               int64_t* __restrict__ out, int64_t* __restrict__ tok_sid) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t x = item_mask[w];
    if (!x) return;
    int64_t k = word_rank[w];
    int64_t s = string_of(row_off, n_str, (w << 6) + __builtin_ctzll(x));
    int64_t lo = row_off[s], hi = row_off[s + 1];
    while (x) {
        const int64_t p = (w << 6) + __builtin_ctzll(x);
        x &= x - 1;
        while (p >= hi) { ++s; lo = hi; hi = row_off[s + 1]; }   // next (non-empty) string
        if (KIND == 0) {
            out[k] = p - lo;
        } else {
            const Token t = token_at(bits, space, p, total);
            if (KIND == 1) {
                out[2 * k] = t.a2 - lo;
                out[2 * k + 1] = t.e2 - lo;
            } else {
                out[4 * k] = p - lo;
                out[4 * k + 1] = t.e - lo;
                out[4 * k + 2] = t.a2 - lo;
                out[4 * k + 3] = t.e2 - lo;
                tok_sid[k] = s;
            }
        }
        ++k;
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
hipError_t launch_word_counts(bool spans, const uint64_t* bits, const uint64_t* space, int64_t n_words, int64_t total,
                              uint64_t* kept, int64_t* cnt, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_words + 255) / 256)), block(256);
    if (spans) hipLaunchKernelGGL((k_word_counts<true>), grid, block, 0, st, bits, space, n_words, total, kept, cnt);
    else hipLaunchKernelGGL((k_word_counts<false>), grid, block, 0, st, bits, space, n_words, total, kept, cnt);
    return hipGetLastError();
}

hipError_t launch_string_counts(const uint64_t* mask, const int64_t* word_rank, const int64_t* row_off, int64_t n_str,
                                int64_t total, const int64_t* n_items, int64_t* counts, hipStream_t st) {
    if (n_str <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_string_counts, dim3((unsigned)((n_str + 255) / 256)), dim3(256), 0, st, mask, word_rank, row_off,
                       n_str, total, n_items, counts);
    return hipGetLastError();
}

hipError_t launch_word_scatter(int kind, const uint64_t* bits, const uint64_t* space, const uint64_t* item_mask,
                               const int64_t* word_rank, int64_t n_words, int64_t total, const int64_t* row_off,
                               int64_t n_str, int64_t* out, int64_t* tok_sid, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_words + 255) / 256)), block(256);
    if (kind == 0)
        hipLaunchKernelGGL((k_word_scatter<0>), grid, block, 0, st, bits, space, item_mask, word_rank, n_words, total,
                           row_off, n_str, out, tok_sid);
    else if (kind == 1)
        hipLaunchKernelGGL((k_word_scatter<1>), grid, block, 0, st, bits, space, item_mask, word_rank, n_words, total,
                           row_off, n_str, out, tok_sid);
    else
        hipLaunchKernelGGL((k_word_scatter<2>), grid, block, 0, st, bits, space, item_mask, word_rank, n_words, total,
                           row_off, n_str, out, tok_sid);
    return hipGetLastError();
}

}  // namespace latok
