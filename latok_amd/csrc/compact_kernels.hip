// compact_kernels.hip -- everything tokenize() / featurize() do AFTER the split mask, on the device and parallel over
// the packed buffer (not over strings, so one 1 M-char document is as parallel as 8 000 tweets):
//
//   boundary offsets   np.nonzero(splits)[0] per string                      reference default_tokenizer.py:148
//   token spans        slice between consecutive boundaries, strip, drop ''   reference default_tokenizer.py:149-158
//   token features     per-token sums of the 25 matrix columns (featurize)    reference default_tokenizer.py:163-191
//
// Inputs are the two bitmasks the tile kernel writes (boundary bits, SPACE bits; bit i = packed char i) and row_off.
// Because strings are contiguous and ordered in the packed buffer, "all offsets of string 0, then string 1, ..." is
// simply position order, so the output index of an item is the global rank of its bit.  Three launches:
//   k_word_counts        one wave per 4096-char tile: items per word (uint16 prefix inside the tile), items per tile
//   k_scan_chained       exclusive scan of the tile counts over the whole batch in one launch (single-pass chained scan
//                        with look-back over workgroup totals): every tile's rank and, for the host, the item total
//   k_counts_scatter     workgroups split into two roles: one thread per string: count = rank(end) - rank(start), O(1);
//                        one wave per tile: scatter its items at tile rank + position inside the tile
// A token is kept iff it contains a non-SPACE char; its extent needs the next boundary bit, which normally sits in the
// same or the next word (a lane follows the mask forward only for its own items).  Counts and records are written as
// int64 or, on request (LATOK_OUT_INT32), as int32: the records are most of the traffic of these paths.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bitscan.h"
#include "kernels.h"
#include "lane_math.h"

namespace latok {

// ---- pass 1 --------------------------------------------------------------------------------------------------------
// SPANS = false: items = boundary bits.  SPANS = true: items = boundaries whose token is kept; the kept-mask word is
// stored for the later passes.  A token is kept iff it holds a non-SPACE char; for all but the last boundary of a
// word that is a mask test inside the word, the last one looks ahead until the next boundary (normally the next word).
// One wave per 4096-char tile, lane = word: the tile's item count and every word's exclusive prefix inside its tile
// (uint16); rank(word) = tile_rank[tile] + word_pref[word], tile_rank = exclusive scan of the tile counts (k_scan_chained).
template <bool SPANS>
__device__ __forceinline__ void word_counts_tile(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space,
                                                 int64_t n_words, int64_t total, uint64_t* __restrict__ kept_out,
                                                 int64_t* __restrict__ tile_cnt, uint16_t* __restrict__ word_pref, int64_t t, int lane) {
    const int64_t w = t * 64 + lane;
    if (t * 64 >= n_words) return;   // whole wave
    int cnt = 0;
    const uint64_t x = w < n_words ? bits[w] : 0ull;
    if (!SPANS) {
        cnt = __popcll(x);
    } else {
        const uint64_t nn = w < n_words ? (~space[w] & valid_mask(w, total)) : 0ull;   // non-SPACE chars of the word
        // the next word's masks from the neighbour lane (lane 63: from memory): the word's last token normally ends there
        uint64_t x1 = __shfl_down(x, 1), nn1 = __shfl_down(nn, 1);
        if (lane == 63) {
            const bool has = w + 1 < n_words;
            x1 = has ? bits[w + 1] : 0ull;
            nn1 = has ? (~space[w + 1] & valid_mask(w + 1, total)) : 0ull;
        }
        // Which boundaries start a token with a non-SPACE char, for all boundaries of the word at once: on the bit-reversed
        // word a boundary is the TOP of its token, so "some non-SPACE below me in my token" is a carry chain -- one add.
        //   generate = non-SPACE chars that are not boundaries, propagate = non-boundaries, carry-in = the token that
        //   continues into the next word(s) has a non-SPACE char there
        bool cin;
        if (x1) cin = (nn1 & ((x1 & (~x1 + 1ull)) - 1ull)) != 0;
        else cin = nn1 != 0 || (x != 0 && w + 1 < n_words && tail_has_nonspace(bits, space, w + 1, n_words, total));
        const uint64_t xr = __builtin_bitreverse64(x), nr = __builtin_bitreverse64(nn);
        const uint64_t g = nr & ~xr, pr = ~xr;
        const uint64_t a = pr | g;
        const uint64_t carries = (a + g + (cin ? 1ull : 0ull)) ^ a ^ g;      // carry INTO every position
        const uint64_t kept = __builtin_bitreverse64(xr & (nr | carries));
        if (w < n_words) kept_out[w] = kept;
        cnt = __popcll(kept);
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (w < n_words) word_pref[w] = (uint16_t)(inc - cnt);
    if (lane == 63) tile_cnt[t] = inc;
}

template <bool SPANS>
__global__ __launch_bounds__(256) void k_word_counts(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space,
                                                     int64_t n_words, int64_t total, uint64_t* __restrict__ kept_out,
                                                     int64_t* __restrict__ tile_cnt, uint16_t* __restrict__ word_pref) {
    word_counts_tile<SPANS>(bits, space, n_words, total, kept_out, tile_cnt, word_pref,
                            ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, threadIdx.x & 63);
}

// Exclusive scan of the tile counts over the whole batch in ONE launch (it used to be three: local scans, scan of the
// block totals, fix-up): single-pass chained scan.  A workgroup takes a ticket -- its position in the scan order is the
// order in which workgroups START, so a workgroup only ever waits for workgroups that are already running --, scans its
// chunk of 4096 counts, publishes the chunk total, and looks back over its predecessors' published words, 64 at a time
// (one per lane), summing totals until it meets a predecessor that already knows its inclusive prefix.  The words carry
// the launch's epoch, so nothing has to be cleared between launches (api.cpp: next_scan_epoch).
constexpr int kChainBlock = 1024, kChainItems = 4, kChainChunk = kChainBlock * kChainItems;
constexpr int kChainValueBits = 44, kChainFlagShift = 44, kChainEpochShift = 46;
constexpr unsigned long long kChainValueMask = (1ull << kChainValueBits) - 1ull;
constexpr unsigned kChainAggregate = 1u, kChainPrefix = 2u;

__device__ __forceinline__ void chain_publish(unsigned long long* slot, unsigned epoch, unsigned flag, long long value) {
    const unsigned long long w = ((unsigned long long)epoch << kChainEpochShift) | ((unsigned long long)flag << kChainFlagShift) |
                                 ((unsigned long long)value & kChainValueMask);
    // relaxed, agent scope: the word IS the message (value + flag + epoch in one 64-bit store), nothing else has to be
    // visible with it -- a release here writes back the whole L2 of the XCD for every workgroup (measured: 10x slower)
    __hip_atomic_store(slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kChainBlock) void k_scan_chained(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                                              unsigned long long* __restrict__ chain, unsigned* __restrict__ ticket,
                                                              unsigned epoch, unsigned n_blocks, int64_t* __restrict__ total_out,
                                                              int64_t* __restrict__ total_host, int* __restrict__ err) {
    __shared__ unsigned s_ticket;
    __shared__ long long s_wave_tot[kChainBlock / 64];
    __shared__ long long s_prefix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const unsigned b = s_ticket;
    const int64_t base = (int64_t)b * kChainChunk + (int64_t)threadIdx.x * kChainItems;
    long long v[kChainItems], sum = 0;
#pragma unroll
    for (int j = 0; j < kChainItems; ++j) {
        v[j] = base + j < n ? in[base + j] : 0;
        sum += v[j];
    }
    long long inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave_tot[wave] = inc;
    __syncthreads();
    long long wave_excl = 0, agg = 0;
#pragma unroll
    for (int k = 0; k < kChainBlock / 64; ++k) {
        const long long x = s_wave_tot[k];
        if (k < wave) wave_excl += x;
        agg += x;
    }
    if (wave == 0) {
        if (lane == 0) chain_publish(&chain[b], epoch, b == 0 ? kChainPrefix : kChainAggregate, agg);
        long long prefix = 0;
        for (long long j0 = (long long)b - 1; j0 >= 0; j0 -= 64) {   // wave-uniform
            const long long j = j0 - lane;
            unsigned long long sv = 0;
            if (j >= 0) {
                // (bounded: every predecessor holds a ticket, i.e. is running, and publishes before it waits for anything,
                // so this never spins for long; should the state ever be corrupt, the launch ends with an error flag
                // instead of hanging the device)
                int spins = 0;
                do {
                    sv = __hip_atomic_load(&chain[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (((unsigned)(sv >> kChainEpochShift) != epoch || ((sv >> kChainFlagShift) & 3ull) == 0ull) && ++spins < (1 << 22));
                if (spins >= (1 << 22)) {
                    *err = 2;
                    sv = ((unsigned long long)epoch << kChainEpochShift) | ((unsigned long long)kChainPrefix << kChainFlagShift);
                }
            }
            const bool is_prefix = j < 0 || ((sv >> kChainFlagShift) & 3ull) == kChainPrefix;   // before workgroup 0: prefix 0
            const long long val = j >= 0 ? (long long)(sv & kChainValueMask) : 0;
            const unsigned long long pm = __ballot(is_prefix);
            const int first = pm ? __builtin_ctzll(pm) : 64;      // the nearest predecessor that knows its inclusive prefix
            long long part = lane <= first ? val : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
            prefix += part;
            if (pm) break;
        }
        if (lane == 0) {
            s_prefix = prefix;
            if (b > 0) chain_publish(&chain[b], epoch, kChainPrefix, prefix + agg);
            if (b == n_blocks - 1) {          // the last ticket: every workgroup has taken its own by now
                *total_out = prefix + agg;
                if (total_host) *total_host = prefix + agg;   // pinned, device-mapped: read by the host after the stream drains
                *ticket = 0u;
            }
        }
    }
    __syncthreads();
    long long run = s_prefix + wave_excl + inc - sum;
#pragma unroll
    for (int j = 0; j < kChainItems; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

// ---- pass 2: items per string = rank(row_off[s+1]) - rank(row_off[s]) ------------------------------------------------
__device__ __forceinline__ int64_t rank_at(const uint64_t* __restrict__ mask, const int64_t* __restrict__ tile_rank,
                                           const uint16_t* __restrict__ word_pref, int64_t x, int64_t total, int64_t n_items) {
    if (x >= total) return n_items;
    const int64_t w = x >> 6;
    return tile_rank[w >> 6] + word_pref[w] + __popcll(mask[w] & low_mask((int)(x & 63)));
}
// OUT = int64_t or int32_t.  A string of 2^31 chars or more cannot be reported in int32: *err is raised (the host turns
// it into LATOK_ERR_INVALID).
template <typename OUT>
__device__ __forceinline__ void string_counts_role(int64_t s, const uint64_t* __restrict__ mask, const int64_t* __restrict__ tile_rank,
                                                   const uint16_t* __restrict__ word_pref, const int64_t* __restrict__ row_off,
                                                   int64_t n_str, int64_t total, int64_t n, OUT* __restrict__ counts,
                                                   int* __restrict__ err) {
    if (s >= n_str) return;
    const int64_t r0 = row_off[s], r1 = row_off[s + 1];
    if (sizeof(OUT) == 4 && r1 - r0 > 0x7FFFFFFFll) *err = 1;   // (plain store: the flag may live in pinned host memory)
    counts[s] = (OUT)(rank_at(mask, tile_rank, word_pref, r1, total, n) - rank_at(mask, tile_rank, word_pref, r0, total, n));
}
template <typename OUT>
__global__ __launch_bounds__(256) void k_string_counts(const uint64_t* __restrict__ mask, const int64_t* __restrict__ tile_rank,
                                                       const uint16_t* __restrict__ word_pref, const int64_t* __restrict__ row_off,
                                                       int64_t n_str, int64_t total, const int64_t* __restrict__ n_items,
                                                       OUT* __restrict__ counts, int* __restrict__ err) {
    string_counts_role<OUT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, mask, tile_rank, word_pref, row_off, n_str, total, *n_items,
                            counts, err);
}

// ---- pass 3 --------------------------------------------------------------------------------------------------------
// One wave per 64 consecutive words (4096 chars).  The wave
//   (a) finds where the owning string of every position begins: the string starts inside the tile become bits in LDS,
//       the last start before each word is a prefix max over the lanes, and the string that was open when the tile began
//       comes from the per-tile string index the tile kernel publishes;
//   (b) lane = word: walks the items of its word (all mask arithmetic inside the word; only a token that runs past the
//       word's end follows the masks further) and puts the records into an LDS window at their rank inside the wave;
//   (c) streams the window to the output: consecutive lanes write consecutive 8-byte words.
// KIND 0: offsets[k] = p - start of its string.   KIND 1: spans[2k..] = stripped extent.
// (featurize writes its 4-value span records from k_features_tiles, together with the sums)
constexpr int scatter_waves(int kind) { return 4; }   // waves per workgroup

__device__ __forceinline__ int64_t scatter_lower_bound(const int64_t* __restrict__ row_off, int64_t n_entries, int64_t c,
                                                       int lane) {
    int64_t lo = 0, hi = n_entries;   // smallest s with row_off[s] >= c, 64 probes per round
    while (hi > lo) {
        const int64_t len = hi - lo;
        const int64_t step = (len + 63) / 64;
        const int64_t p = lo + (int64_t)lane * step;
        const bool pred = p < hi && row_off[p] >= c;
        const unsigned long long m = __ballot(pred);
        if (!m) {
            const int64_t n_valid = (len + step - 1) / step;
            lo = min(lo + (n_valid - 1) * step + 1, hi);
        } else {
            const int f = __builtin_ctzll(m);
            hi = lo + (int64_t)f * step;
            if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
        }
    }
    return lo;
}

template <int KIND, typename OUT>
__device__ __forceinline__ void counts_scatter_block(
    const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space, const uint64_t* __restrict__ item_mask,
    const int64_t* __restrict__ tile_rank, const int64_t* __restrict__ tile_cnt, const uint16_t* __restrict__ word_pref,
    int64_t n_words, int64_t total, const int64_t* __restrict__ row_off, int64_t n_str,
    const int64_t* __restrict__ tile_first, OUT* __restrict__ out, const int64_t* __restrict__ n_items_dev, int64_t cap,
    OUT* __restrict__ counts, unsigned n_scatter_blocks, int* __restrict__ err, unsigned vb) {   // vb: (virtual) workgroup index
    static_assert(scatter_waves(KIND) * 64 == 256, "both roles use 256-thread workgroups");
    if (vb >= n_scatter_blocks) {   // role 2: one thread per string
        if (counts)
            string_counts_role<OUT>((int64_t)(vb - n_scatter_blocks) * 256 + threadIdx.x, item_mask, tile_rank, word_pref,
                                    row_off, n_str, total, *n_items_dev, counts, err);
        return;
    }
    // role 1: one wave per tile.  The caller's buffer holds `cap` items; when the batch has more, nothing is written
    // (the host reports the needed size) -- the launch does not have to wait for the host to learn the total.
    if (*n_items_dev > cap) return;
    constexpr int kScatterWaves = scatter_waves(KIND);
    constexpr int kCodes = 1024;                                    // items per round
    // KIND 0: window of values (OUT); KIND 1: the item codes (2 B each) + the 48-byte mask rows of the 64 words.  Sized for
    // what the form needs (it was 8 KB per wave for every form: 4 workgroups per CU).  Measured on C2, same box: int32 offsets
    // 0.174 -> 0.165 ms and int32 spans 0.233 -> 0.222 at 7 - 8 workgroups per CU, but int64 spans 0.247 -> 0.256 (twice
    // the store stream per token): that form keeps the footprint that holds it at 4.
    constexpr int kSpan64Buf = 6656;   // int64 spans: 5 workgroups per CU (4: C3 +2.5 %; 7: C2 +4 %)
    constexpr int kBufBytes = KIND == 0 ? kCodes * (int)sizeof(OUT) : (sizeof(OUT) == 8 ? kSpan64Buf : kCodes * 2 + 64 * 48);
    __shared__ __attribute__((aligned(16))) uint8_t buf_s[kScatterWaves][kBufBytes];
    __shared__ long long smax_s[kScatterWaves][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w0 = ((int64_t)vb * kScatterWaves + wave) * 64;
    if (w0 >= n_words) return;                                      // whole wave (no block-wide barrier is used below)
    const int64_t w = w0 + lane;
    const int n_wave = (int)tile_cnt[w0 >> 6];
    if (n_wave == 0) return;
    const uint64_t x = w < n_words ? item_mask[w] : 0ull;
    const int off = w < n_words ? (int)word_pref[w] : 0;            // rank of my first item inside the wave
    const int64_t base_out = tile_rank[w0 >> 6];
    uint16_t* codes = reinterpret_cast<uint16_t*>(buf_s[wave]);
    OUT* win = reinterpret_cast<OUT*>(buf_s[wave]);
    long long* smax = smax_s[wave];

    // (a) where the string that owns a position begins: string-start bits of the tile in LDS (one atomicOr per string
    //     that starts in it), per lane the last start before its word (prefix max), and for everything before the first
    //     start the string that was open when the tile began.  No per-item loads: a lane that had to fetch row_off at
    //     every string change stalled the whole wave on every step of its item loop.
    const int64_t t0 = w0 << 6;
    // first string starting at or after t0: published by the tile kernel (one wave = one tile), else searched
    int64_t idx0 = tile_first ? tile_first[w0 >> 6] : scatter_lower_bound(row_off, n_str, t0, lane);
    idx0 = idx0 < 0 ? 0 : (idx0 > n_str ? n_str : idx0);
    if (idx0 > n_str) idx0 = n_str;
    const int64_t start_before = idx0 > 0 ? row_off[idx0 - 1] : 0;
    unsigned long long* bw = reinterpret_cast<unsigned long long*>(smax);
    bw[lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t i0 = idx0; i0 < n_str; i0 += 64) {
        const int64_t sidx = i0 + lane;
        const int64_t ro = sidx < n_str ? row_off[sidx] : INT64_MAX;
        const int64_t rel = ro - t0;
        if (rel >= 0 && rel < 4096) atomicOr(&bw[rel >> 6], 1ull << (rel & 63));
        if (__shfl(ro, 63) >= t0 + 4096) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint64_t Bw = bw[lane];
    int carry = Bw ? 64 * lane + 63 - __builtin_clzll(Bw) : -1;     // tile-relative position of my word's last string start
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(carry, d);
        if (lane >= d && o > carry) carry = o;
    }
    carry = __shfl_up(carry, 1);
    if (lane == 0) carry = -1;
    const int64_t lo_in = carry >= 0 ? t0 + carry : start_before;

    // (b) + (c), window by window
    const uint64_t xb = w < n_words ? bits[w] : 0ull;               // all boundaries of the word (item_mask is a subset)
    const uint64_t nn = (KIND != 0 && w < n_words) ? (~space[w] & valid_mask(w, total)) : 0ull;
    // the next word's masks (a token that crosses the word's end normally ends there): from the neighbour lane
    uint64_t xb1 = 0, nn1 = 0;
    if (KIND != 0) {
        xb1 = __shfl_down(xb, 1);
        nn1 = __shfl_down(nn, 1);
        if (lane == 63) {
            const bool has = w + 1 < n_words;
            xb1 = has ? bits[w + 1] : 0ull;
            nn1 = has ? (~space[w + 1] & valid_mask(w + 1, total)) : 0ull;
        }
    }
    // (b) token spans: every lane lists its items as (lane, bit) codes at their rank inside the wave ...
    // (c) ... and the wave then takes the items in rank order, lane j the j-th: the owner word's masks arrive through
    //     shuffles, so all 64 lanes are busy whatever the spread of items over the words is (a word-major loop runs as
    //     long as the fullest word, typically 2x the mean), and the records go straight to consecutive addresses.
    uint64_t rest = x;
    int k = off;                                                    // wave rank of my next item
    if (KIND == 0) {
        // offsets are one subtraction per item: here the word-major loop through an LDS window is the faster form
        // Every string start is itself a boundary (splits[0] = 1), so the walk over a word's items meets the string starts
        // in passing: `cur` = offset of the word's bit 0 relative to the string the walk is in -- base - lo_in on entry,
        // -b once the item at bit b is a string start.  32-bit halves: no 64-bit shifts, masks or clz in the loop (the
        // former form -- mask the string starts below the item, clz -- was 3x the instructions).
        const int64_t base = w << 6;
        uint32_t r0 = (uint32_t)rest, r1 = (uint32_t)(rest >> 32);
        const uint32_t B0 = (uint32_t)Bw, B1 = (uint32_t)(Bw >> 32);
        OUT cur = (OUT)(base - lo_in);
        for (int win0 = 0; win0 < n_wave; win0 += kCodes) {
            const int lim = win0 + kCodes;
            while (r0 && k < lim) {
                const int b = __builtin_ctz(r0);
                r0 &= r0 - 1u;
                if ((B0 >> b) & 1u) cur = (OUT)(-b);
                win[k - win0] = (OUT)(cur + b);
                ++k;
            }
            while (!r0 && r1 && k < lim) {
                const int b = __builtin_ctz(r1);
                r1 &= r1 - 1u;
                if ((B1 >> b) & 1u) cur = (OUT)(-(32 + b));
                win[k - win0] = (OUT)(cur + 32 + b);
                ++k;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // stream the window out: single elements up to the first 16-byte boundary of the output, then 16-byte stores
            const int n_val = min(kCodes, n_wave - win0);
            OUT* dst = out + base_out + win0;
            constexpr int kPer = 16 / (int)sizeof(OUT);                   // elements per 16-byte store
            const int head = min(n_val, (int)(((16u - ((uintptr_t)dst & 15u)) & 15u) / sizeof(OUT)));
            if (lane < head) __builtin_nontemporal_store(win[lane], dst + lane);
            const int n_vec = (n_val - head) / kPer;
            typedef OUT vec_t __attribute__((ext_vector_type(16 / sizeof(OUT))));
            for (int i = lane; i < n_vec; i += 64) {
                vec_t v;
#pragma unroll
                for (int e = 0; e < kPer; ++e) v[e] = win[head + kPer * i + e];
                __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(dst + head) + i);
            }
            const int tail0 = head + kPer * n_vec;
            if (lane < n_val - tail0) __builtin_nontemporal_store(win[tail0 + lane], dst + tail0 + lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // the lanes' masks as 48-byte rows in LDS (behind the codes): in the token-major loop lane j reads the row of the
    // word that owns its item with three 16-byte reads (six 64-bit shuffles = twelve ds_bpermute before)
    uint64_t* rows = reinterpret_cast<uint64_t*>(buf_s[wave] + kCodes * 2);            // codes take kCodes * 2 bytes
    {
        uint64_t* r = rows + 6 * lane;
        r[0] = xb; r[1] = nn; r[2] = xb1; r[3] = nn1; r[4] = Bw; r[5] = (uint64_t)lo_in;
    }
    for (int win0 = 0; win0 < n_wave; win0 += kCodes) {
        while (rest && k < win0 + kCodes) {
            const int b = __builtin_ctzll(rest);
            rest &= rest - 1;
            codes[k - win0] = (uint16_t)((lane << 6) | b);
            ++k;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_here = min(kCodes, n_wave - win0);
        for (int j0 = 0; j0 < n_here; j0 += 64) {
            const int j = j0 + lane;
            const bool active = j < n_here;
            const int code = active ? (int)codes[j] : 0;
            const int owner = code >> 6, b = code & 63;
            const uint64_t* orow = rows + 6 * owner;
            const uint64_t o_xb = orow[0], o_nn = orow[1], o_xb1 = orow[2], o_nn1 = orow[3], o_Bw = orow[4];
            const int64_t o_lo_in = (int64_t)orow[5];
            if (active) {
                const int64_t obase = (w0 + owner) << 6;
                const uint64_t bl = o_Bw & ((2ull << b) - 1ull);      // string starts at or before the item (b = 63: all)
                const int64_t lo = bl ? obase + 63 - __builtin_clzll(bl) : o_lo_in;
                {
                    // token [p, e): e = next boundary; stripped extent [a2, e2).  The 64 positions from the item on, taken out
                    // of the 128-bit pair (owner word, next word): one form whether the token ends in its own word or in the
                    // next one (two divergent branches before, and some lane of 64 nearly always crosses a word)
                    const uint64_t X = (o_xb >> b) | ((o_xb1 << 1) << (63 - b));     // boundaries at p, p+1, ..
                    const uint64_t N = (o_nn >> b) | ((o_nn1 << 1) << (63 - b));     // non-SPACE chars at p, p+1, ..
                    const uint64_t after = X & ~1ull;
                    int64_t a2, e2;
                    if (after) {                                    // the token ends within 64 chars (kept => seg != 0)
                        const uint64_t seg = N & (((after & (~after + 1ull)) - 1ull));
                        const int64_t p = obase + b;
                        a2 = p + __builtin_ctzll(seg);
                        e2 = p + 64 - __builtin_clzll(seg);
                    } else if (o_xb1 & ~((1ull << b) - 1ull)) {     // (rare) longer: ends at a boundary of the next word beyond p + 63
                        const int eb = __builtin_ctzll(o_xb1 & ~((1ull << b) - 1ull));
                        const uint64_t seg0 = o_nn & (~0ull << b);
                        const uint64_t seg1 = o_nn1 & ((1ull << eb) - 1ull);
                        a2 = seg0 ? obase + __builtin_ctzll(seg0) : obase + 64 + __builtin_ctzll(seg1);
                        e2 = seg1 ? obase + 128 - __builtin_clzll(seg1) : obase + 64 - __builtin_clzll(seg0);
                    } else {                                        // (rare) no boundary up to the end of the next word
                        const int64_t e = next_set_bit(bits, obase + 128, total);
                        const uint64_t seg = o_nn & (~0ull << b);
                        a2 = seg ? obase + __builtin_ctzll(seg) : (o_nn1 ? obase + 64 + __builtin_ctzll(o_nn1) : next_zero_bit(space, obase + 128, e));
                        e2 = prev_zero_end(space, a2, e);
                    }
                    typedef OUT out2 __attribute__((ext_vector_type(2)));
                    out2 v;
                    v.x = (OUT)(a2 - lo);
                    v.y = (OUT)(e2 - lo);
                    __builtin_nontemporal_store(v, reinterpret_cast<out2*>(out) + base_out + win0 + j);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KIND, typename OUT>
__global__ __launch_bounds__(scatter_waves(KIND) * 64) void k_counts_scatter(
    const uint64_t* __restrict__ bits, const uint64_t* __restrict__ space, const uint64_t* __restrict__ item_mask,
    const int64_t* __restrict__ tile_rank, const int64_t* __restrict__ tile_cnt, const uint16_t* __restrict__ word_pref,
    int64_t n_words, int64_t total, const int64_t* __restrict__ row_off, int64_t n_str,
    const int64_t* __restrict__ tile_first, OUT* __restrict__ out, const int64_t* __restrict__ n_items_dev, int64_t cap,
    OUT* __restrict__ counts, unsigned n_scatter_blocks, int* __restrict__ err, DoneSignal done) {
    counts_scatter_block<KIND, OUT>(bits, space, item_mask, tile_rank, tile_cnt, word_pref, n_words, total, row_off, n_str, tile_first,
                                    out, n_items_dev, cap, counts, n_scatter_blocks, err, blockIdx.x);
    signal_block_done(done);   // (pinned outputs of a small host batch: the host polls the completion word)
}

// ---- code-point results of a UTF-8 batch from its BYTE-space results ------------------------------------------------------
// The reference reads a str as code points (latok.c:53-55,79) and reports boundaries as code-point indices.  A UTF-8 batch in
// code-point units used to be decoded into a UTF-32 copy first (1 B read + 4 B written + 4 B read again per char); instead the
// byte-space tile kernel runs on the bytes themselves and leaves two bitmasks over the BYTES -- boundaries (set at lead bytes)
// and lead bytes -- plus, per 64-byte word, the number of leads of its tile before it and the leads per tile.  Code point k of
// the batch is the k-th lead byte, so
//   cp mask        = the boundary bits at the lead positions, packed: per word pext(boundaries, leads), appended at the word's
//                    rank = leads before it (tile_rank from k_scan_chained over the tile counts + the word's prefix);
//   cp_row_off[s]  = number of leads before byte_off[s].
// One workgroup per 16 tiles of 64 words; the packed chunks of its words meet in one LDS window (a chunk straddles at most two
// output words) and leave as whole words.  Every output word is written exactly once, by the workgroup that holds the lead of
// its LAST bit (the batch's final, partial word: by the last workgroup with a lead): the bits of a workgroup's first word that
// belong to earlier ones are recomputed by its first wave from the words in front of it.  No atomics on global memory, no
// cleared output (two global atomics per tile cost 40 of 110 us on C3); the kernel is bound by the pext: through a 256-byte table
// of 4-bit pexts in LDS (~100 VALU per word; the five-round arithmetic compress: ~256, 80 us on C3; one wave per tile with a
// look-back pext of its own: 106 us).  Role 2 (the workgroups behind): one thread
// per row offset.
// *odd is raised when the byte-space model and the decoder's model of MALFORMED input differ: a continuation byte with no lead
// byte within the 3 bytes before it, or at the start of a string (the host then takes the staged decoder instead).
constexpr int kCompressWaves = 16;                         // tiles per workgroup
constexpr int kCompressWords = kCompressWaves * 64;        // input words per workgroup = output words it can own (+ 1)
template <bool TWO>   // TWO: a second byte-space mask (the SPACE plane, for token spans) is packed the same way into out_mask2
__global__ __launch_bounds__(kCompressWaves * 64) void k_lead_compress(
    const uint64_t* __restrict__ bmask, const uint64_t* __restrict__ bmask2, uint64_t* __restrict__ out_mask2,
    const uint64_t* __restrict__ lead, const int64_t* __restrict__ tile_rank, const int64_t* __restrict__ tile_cnt,
    const uint16_t* __restrict__ word_pref, int64_t n_words, int64_t total_bytes, const int64_t* __restrict__ byte_off, int64_t n_str,
    const int64_t* __restrict__ total_cps_dev, uint64_t* __restrict__ out_mask, int64_t cap_words, int64_t* __restrict__ cp_row_off,
    int* __restrict__ odd, unsigned n_tile_blocks) {
    if (blockIdx.x >= n_tile_blocks) {   // role 2: code-point offset of every string (and of the end of the batch)
        const int64_t total_cps = *total_cps_dev;
        const int64_t s = (int64_t)(blockIdx.x - n_tile_blocks) * (kCompressWaves * 64) + threadIdx.x;
        if (s > n_str) return;
        const int64_t b = byte_off[s];
        if (b >= total_bytes) { cp_row_off[s] = total_cps; return; }
        const int64_t bw = b >> 6;
        const uint64_t mw = lead[bw];
        cp_row_off[s] = tile_rank[bw >> 6] + word_pref[bw] + __popcll(mw & low_mask((int)(b & 63)));
        if (s < n_str && byte_off[s + 1] > b && !((mw >> (b & 63)) & 1ull)) *odd = 1;   // a string begins with a continuation byte
        return;
    }
    // The workgroup takes kCompressWaves consecutive tiles; their packed chunks meet in ONE window in LDS, so only the
    // workgroup's first output word needs bits from in front of it (and only its last, partial one is left to the next).
    __shared__ unsigned long long win_s[TWO ? 2 : 1][kCompressWords + 2];
    __shared__ __attribute__((aligned(16))) uint8_t pext_tab[256];       // lk_pext4_entry: 4-bit pexts (lane_math.h)
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) pext_tab[tid] = lk_pext4_entry((uint32_t)tid);
    const int64_t n_tiles = (n_words + 63) >> 6;
    const int64_t T0 = (int64_t)blockIdx.x * kCompressWaves;
    const int64_t T1 = min(T0 + kCompressWaves, n_tiles) - 1;      // the workgroup's last tile
    const int64_t t = min(T0 + (tid >> 6), T1);
    const int64_t w = T0 * 64 + tid;
    const bool in = w < n_words;
    // everything the thread needs from memory is requested here, in one round trip, by unconditional loads at clamped addresses
    // (a predicated load is a branch, and hipcc waits for everything in flight at it)
    const int64_t total_cps = *total_cps_dev;
    const int64_t wc = in ? w : n_words - 1;
    lk_u64 m = lead[wc];
    lk_u64 x = bmask[wc];
    lk_u64 x2 = TWO ? bmask2[wc] : 0ull;
    const uint64_t m_before = lead[wc > 0 ? wc - 1 : 0];           // (the word before mine: every thread loads its own, coalesced)
    const int64_t my_pos0 = tile_rank[t];                          // code-point index of my tile's first lead
    const int pref = (int)word_pref[wc];
    const int64_t pos0 = tile_rank[T0];                            // ... of the workgroup's
    const int64_t end = tile_rank[T1] + tile_cnt[T1];              // one past the workgroup's last code point
    // the look-back (first wave): lane k takes the k-th word in front of the workgroup
    const int64_t wk = T0 * 64 - 1 - lane > 0 ? T0 * 64 - 1 - lane : 0;
    lk_u64 mb = 0, xb = 0, xb2 = 0;
    if (tid < 64) { mb = lead[wk]; xb = bmask[wk]; xb2 = TWO ? bmask2[wk] : 0ull; }
    unsigned long long* win = win_s[0];
    unsigned long long* win2 = win_s[TWO ? 1 : 0];
    win[tid] = 0ull;
    if (tid < 2) win[kCompressWords + tid] = 0ull;
    if (TWO) {
        win2[tid] = 0ull;
        if (tid < 2) win2[kCompressWords + tid] = 0ull;
    }
    if (!in) { m = 0ull; x = 0ull; x2 = 0ull; }
    // continuation bytes without a lead byte in the 3 bytes before them (malformed input)
    {
        const uint64_t C = in ? (~m & valid_mask(w, total_bytes)) : 0ull;
        const uint64_t Cp = w > 0 ? ~m_before : 0ull;
        const uint64_t run = C & ((C << 1) | (Cp >> 63)) & ((C << 2) | (Cp >> 62)) & ((C << 3) | (Cp >> 61));
        if (run) *odd = 1;
    }
    if ((total_cps + 63) / 64 > cap_words) return;                 // the caller's mask is too small: nothing is written (uniform)
    const int64_t ow0 = pos0 >> 6;                                 // first output word the workgroup has bits in
    // words the workgroup owns: those whose last bit is its own, + the batch's final partial word if its last lead is
    const int64_t own_end = end == total_cps ? (end + 63) >> 6 : end >> 6;
    if (end == pos0 || own_end <= ow0) return;                     // no lead at all / all bits lie in a word a later one owns (uniform)
    // Every byte of the workgroup's tiles is a lead (ASCII text) and its first code point opens an output word: code point k of the
    // range IS byte k, the packed words are the input words (uniform; nothing to pack, nothing shared with a neighbour).
    {
        const int64_t b0 = T0 * kTile, b1 = min((T1 + 1) * (int64_t)kTile, total_bytes);
        if ((pos0 & 63) == 0 && end - pos0 == b1 - b0) {
            if (in) {
                out_mask[ow0 + tid] = x;
                if (TWO) out_mask2[ow0 + tid] = x2;
            }
            return;
        }
    }
    __syncthreads();
    {
        const int64_t pos = my_pos0 + pref;
        const int rel = (int)((pos >> 6) - ow0), sh = (int)(pos & 63);
        if (TWO) {
            if (x | x2) {
                lk_pext64_lut<true>(&x, &x2, m, pext_tab);
                if (x) {
                    atomicOr(&win[rel], x << sh);
                    if (sh && (x >> (64 - sh))) atomicOr(&win[rel + 1], x >> (64 - sh));
                }
                if (x2) {
                    atomicOr(&win2[rel], x2 << sh);
                    if (sh && (x2 >> (64 - sh))) atomicOr(&win2[rel + 1], x2 >> (64 - sh));
                }
            }
        } else if (x) {                                            // (a word without boundaries adds nothing)
            lk_u64 c = x, unused = 0;
            lk_pext64_lut<false>(&c, &unused, m, pext_tab);
            atomicOr(&win[rel], c << sh);
            if (sh && (c >> (64 - sh))) atomicOr(&win[rel + 1], c >> (64 - sh));
        }
    }
    if (tid < 64) {
        // the bits of the first word that belong to earlier workgroups: the `need` leads in front of this one, nearest word first
        // (the 63 leads before a tile lie in its previous 4 words in well-formed text; the look-back goes on for as long as
        // malformed input makes it)
        int need = (int)(pos0 & 63);
        for (int64_t back = 0; need > 0; back += 64) {             // wave-uniform; one round unless the input is malformed
            const int64_t wj = T0 * 64 - 1 - back - lane;
            if (back > 0) {
                mb = wj >= 0 ? lead[wj] : 0ull;
                xb = wj >= 0 ? bmask[wj] : 0ull;
                xb2 = (TWO && wj >= 0) ? bmask2[wj] : 0ull;
            } else if (wj < 0) {
                mb = 0ull; xb = 0ull; xb2 = 0ull;
            }
            const int cnt = __popcll(mb);
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            const int before = inc - cnt;                           // leads between my word and the workgroup's first
            if (cnt > 0 && before < need) {
                const int take = min(cnt, need - before);           // my top `take` leads
                lk_u64 c = xb, c2 = xb2;
                lk_pext64_lut<TWO>(&c, &c2, mb, pext_tab);
                c >>= (cnt - take);
                if (c) atomicOr(&win[0], c << (need - before - take));
                if (TWO) {
                    c2 >>= (cnt - take);
                    if (c2) atomicOr(&win2[0], c2 << (need - before - take));
                }
            }
            need -= __shfl(inc, 63);
            if (T0 * 64 - 1 - back - 63 <= 0) break;                // the batch begins here (cannot happen with need > 0: ranks are exact)
        }
    }
    __syncthreads();
    const int n_out = (int)(own_end - ow0);                         // 1 .. kCompressWords + 1
    for (int j = tid; j < n_out; j += kCompressWaves * 64) {
        out_mask[ow0 + j] = win[j];
        if (TWO) out_mask2[ow0 + j] = win2[j];
    }
}

hipError_t launch_lead_compress(const uint64_t* bmask, const uint64_t* bmask2, const uint64_t* lead, const int64_t* tile_rank,
                                const int64_t* tile_cnt, const uint16_t* word_pref, int64_t n_words, int64_t total_bytes,
                                const int64_t* byte_off, int64_t n_str, const int64_t* total_cps_dev, uint64_t* out_mask,
                                uint64_t* out_mask2, int64_t cap_words, int64_t* cp_row_off, int* odd, hipStream_t st) {
    const int64_t n_tiles = (n_words + 63) / 64;
    const unsigned nb_tiles = (unsigned)((n_tiles + kCompressWaves - 1) / kCompressWaves);
    const unsigned nb_rows = (unsigned)((n_str + 1 + kCompressWaves * 64 - 1) / (kCompressWaves * 64));
    const dim3 grid(nb_tiles + nb_rows), block(kCompressWaves * 64);
    if (bmask2)
        hipLaunchKernelGGL(k_lead_compress<true>, grid, block, 0, st, bmask, bmask2, out_mask2, lead, tile_rank, tile_cnt, word_pref, n_words,
                           total_bytes, byte_off, n_str, total_cps_dev, out_mask, cap_words, cp_row_off, odd, nb_tiles);
    else
        hipLaunchKernelGGL(k_lead_compress<false>, grid, block, 0, st, bmask, bmask2, out_mask2, lead, tile_rank, tile_cnt, word_pref, n_words,
                           total_bytes, byte_off, n_str, total_cps_dev, out_mask, cap_words, cp_row_off, odd, nb_tiles);
    return hipGetLastError();
}

// exclusive scan of per-tile counts that some other kernel left (the byte-space tile kernel: leads per tile): k_scan_chained alone
hipError_t launch_tile_scan(const int64_t* tile_cnt, int64_t n_tiles, int64_t* tile_rank, unsigned long long* chain, unsigned* ticket,
                            unsigned epoch, int64_t* total_dev, int64_t* total_host, int* err, hipStream_t st) {
    if (n_tiles <= 0) return hipSuccess;
    const unsigned n_blocks = (unsigned)((n_tiles + kChainChunk - 1) / kChainChunk);
    hipLaunchKernelGGL(k_scan_chained, dim3(n_blocks), dim3(kChainBlock), 0, st, tile_cnt, n_tiles, tile_rank, chain, ticket, epoch,
                       n_blocks, total_dev, total_host, err);
    return hipGetLastError();
}

// ---- launchers -----------------------------------------------------------------------------------------------------
int64_t count_blocks(int64_t n_words) { return n_words > 0 ? ((n_words + 63) / 64 + kChainChunk - 1) / kChainChunk : 0; }

hipError_t launch_word_counts_scan(bool spans, const uint64_t* bits, const uint64_t* space, int64_t n_words, int64_t total,
                                   uint64_t* kept, int64_t* tile_cnt, uint16_t* word_pref, int64_t* tile_rank,
                                   unsigned long long* chain, unsigned* ticket, unsigned epoch, int64_t* total_dev,
                                   int64_t* total_host, int* err, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
    const int64_t n_tiles = (n_words + 63) / 64;
    const dim3 grid((unsigned)((n_tiles + 3) / 4)), block(256);
    if (spans) hipLaunchKernelGGL((k_word_counts<true>), grid, block, 0, st, bits, space, n_words, total, kept, tile_cnt, word_pref);
    else hipLaunchKernelGGL((k_word_counts<false>), grid, block, 0, st, bits, space, n_words, total, kept, tile_cnt, word_pref);
    const unsigned n_blocks = (unsigned)count_blocks(n_words);
    hipLaunchKernelGGL(k_scan_chained, dim3(n_blocks), dim3(kChainBlock), 0, st, tile_cnt, n_tiles, tile_rank, chain, ticket, epoch,
                       n_blocks, total_dev, total_host, err);
    return hipGetLastError();
}

hipError_t launch_string_counts(bool out32, const uint64_t* mask, const int64_t* tile_rank, const uint16_t* word_pref,
                                const int64_t* row_off, int64_t n_str, int64_t total, const int64_t* n_items, void* counts, int* err,
                                hipStream_t st) {
    if (n_str <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_str + 255) / 256)), block(256);
    if (out32)
        hipLaunchKernelGGL((k_string_counts<int32_t>), grid, block, 0, st, mask, tile_rank, word_pref, row_off, n_str, total, n_items,
                           (int32_t*)counts, err);
    else
        hipLaunchKernelGGL((k_string_counts<int64_t>), grid, block, 0, st, mask, tile_rank, word_pref, row_off, n_str, total, n_items,
                           (int64_t*)counts, err);
    return hipGetLastError();
}

template <int KIND, typename OUT>
static hipError_t launch_counts_scatter_t(const uint64_t* bits, const uint64_t* space, const uint64_t* item_mask,
                                          const int64_t* tile_rank, const int64_t* tile_cnt, const uint16_t* word_pref,
                                          int64_t n_words, int64_t total, const int64_t* row_off, int64_t n_str,
                                          const int64_t* tile_first, void* out, const int64_t* n_items_dev, int64_t cap,
                                          void* counts, int* err, hipStream_t st, DoneSignal done) {
    const int64_t per_block = (int64_t)scatter_waves(KIND) * 64;
    const unsigned nb_scatter = out ? (unsigned)((n_words + per_block - 1) / per_block) : 0u;
    const unsigned nb_counts = counts ? (unsigned)((n_str + 255) / 256) : 0u;
    if (nb_scatter + nb_counts == 0) return hipSuccess;
    hipLaunchKernelGGL((k_counts_scatter<KIND, OUT>), dim3(nb_scatter + nb_counts), dim3(scatter_waves(KIND) * 64), 0, st, bits, space,
                       item_mask, tile_rank, tile_cnt, word_pref, n_words, total, row_off, n_str, tile_first, (OUT*)out, n_items_dev,
                       cap, (OUT*)counts, nb_scatter, err, done);
    return hipGetLastError();
}

// counts (may be NULL: already written) and / or items (out may be NULL: counts only) in one launch
hipError_t launch_counts_scatter(int kind, bool out32, const uint64_t* bits, const uint64_t* space, const uint64_t* item_mask,
                                 const int64_t* tile_rank, const int64_t* tile_cnt, const uint16_t* word_pref, int64_t n_words,
                                 int64_t total, const int64_t* row_off, int64_t n_str, const int64_t* tile_first, void* out,
                                 const int64_t* n_items_dev, int64_t cap, void* counts, int* err, hipStream_t st, DoneSignal done) {
    if (n_words <= 0) return hipSuccess;
#define LATOK_CS(K, T) launch_counts_scatter_t<K, T>(bits, space, item_mask, tile_rank, tile_cnt, word_pref, n_words, total, row_off, \
                                                     n_str, tile_first, out, n_items_dev, cap, counts, err, st, done)
    if (kind == 0) return out32 ? LATOK_CS(0, int32_t) : LATOK_CS(0, int64_t);
    return out32 ? LATOK_CS(1, int32_t) : LATOK_CS(1, int64_t);
#undef LATOK_CS
}

}  // namespace latok
