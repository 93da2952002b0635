// bitscan.h -- walking a device bitmask (bit i = packed position i) forwards / backwards; shared by the compaction
// kernels and the featurize kernel.
#ifndef LATOK_BITSCAN_H
#define LATOK_BITSCAN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// bits of word w that lie inside the batch
__device__ __forceinline__ uint64_t valid_mask(int64_t w, int64_t total) {
    const int64_t remain = total - (w << 6);
    return remain >= 64 ? ~0ull : (remain <= 0 ? 0ull : ((1ull << remain) - 1ull));
}

__device__ __forceinline__ bool tail_has_nonspace(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space,
                                                  int64_t w, int64_t n_words, int64_t total) {
    for (int64_t v = w + 1; v < n_words; ++v) {
        const uint64_t xb = bits[v];
        const uint64_t nn = ~space[v] & valid_mask(v, total);
        if (xb) return (nn & ((xb & (~xb + 1ull)) - 1ull)) != 0;
        if (nn) return true;
    }
    return false;
}

}  // namespace latok
#endif
