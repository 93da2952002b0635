// kernels.h -- constants, parameter blocks and launcher prototypes shared by the .hip kernel files and api.cpp.
#ifndef LATOK_KERNELS_H
#define LATOK_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_code.h"

namespace latok {

// ---- geometry ------------------------------------------------------------------------------------------------
constexpr int kTile = 4096;              // chars per tile = 64 lanes x 64-bit words (== LATOK_TILE_CHARS)
constexpr int kTblShift = 7;             // stage-2 block = 128 code points
constexpr int kStage1Len = 8705;         // 0x110000 >> 7, + 1 entry for cp >= 0x110000
constexpr int kStage1Pad = 8720;         // padded to 16 B
constexpr int kStage2Len = 255 * 128;    // 32640, multiple of 16
constexpr int kTablesLdsBytes = kStage1Pad + kStage2Len;   // 41360
// byte space (kModeBytes): [stage 1: uint16 block offsets, by cp >> 6 | stage 2: 64-entry blocks] (split_code.h: LK_B6_*)
constexpr int kB6Stage1Len = LK_B6_STAGE1_LEN;                       // 17409
constexpr int kB6Stage1Bytes = (kB6Stage1Len * 2 + 1023) / 1024 * 1024;   // 35840: whole 1 KiB pieces (one wave instruction of the on-demand copy each)
constexpr int kB6MaxBlocks = 400;                                    // distinct 64-char blocks: 354 (split codes) / 394 (rule codes)
constexpr int kB6Stage2Bytes = kB6MaxBlocks * 64;                    // 25600, whole 1 KiB pieces as well
constexpr int kB6TablesBytes = kB6Stage1Bytes + kB6Stage2Bytes;
static_assert(kB6Stage1Bytes % 1024 == 0 && kB6Stage2Bytes % 1024 == 0 && kB6Stage1Bytes <= kTablesLdsBytes, "byte space keeps its stage 1 where the other modes keep both stages");
constexpr int kStageBytes = 64 * 80;     // 64 rows of 64 code bytes + 16 B pad (conflict-free ds_read_b128)
constexpr int kWaveLdsBytes = kStageBytes + 16 + 65 * 8 + 8;  // staging + halo + string-start words = 5664

constexpr int kModeBits = 0;
constexpr int kModeValues = 1;
constexpr int kModeBlockMask = 2;        // compat _gen_block_mask: planes from caller byte arrays, byte mask out
constexpr int kModeRules = 3;            // kModeBits with caller-supplied C_SPLIT / C_MASK / C_SYM (SplitParams::rules)
constexpr int kModeBytes = 4;            // kModeBits in BYTE space: input = UTF-8 bytes (SplitParams::u8), row_off = byte
                                         // offsets, bit i of the mask = byte i (set at the lead byte of a boundary char)
constexpr int kModeLatin1 = 5;           // kModeBits on PEP 393 kind-1 input: SplitParams::u8 = one BYTE per char (U+0000..U+00FF)
constexpr int kModeUcs2 = 6;             // kModeBits on PEP 393 kind-2 input: SplitParams::u8 = one uint16 per char
// modes whose input is SplitParams::u8 and whose tiles use the byte-space LDS layout (pads, 16-byte halo)
// the same four input forms / outputs with caller-supplied C_SPLIT / C_MASK / C_SYM (SplitParams::rules; t2 = rule codes):
// kModeRules is kModeBits + tables; these are the others
constexpr int kModeBytesRules = 7;       // kModeBytes + run-time rule tables
constexpr int kModeLatin1Rules = 8;      // kModeLatin1 + run-time rule tables
constexpr int kModeUcs2Rules = 9;        // kModeUcs2 + run-time rule tables
constexpr int kModeValuesRules = 10;     // kModeValues + run-time rule tables: what gen_split_mask returns for any tables
                                         // (default_tokenizer.py:121-132): rows(C_SPLIT) * mask + rows(C_SYM), string start = 1
// input / output form of a mode, and whether the rules are interpreted at run time
constexpr int mode_base(int mode) {
    return mode == kModeRules ? kModeBits : mode == kModeBytesRules ? kModeBytes : mode == kModeLatin1Rules ? kModeLatin1
         : mode == kModeUcs2Rules ? kModeUcs2 : mode == kModeValuesRules ? kModeValues : mode;
}
constexpr bool mode_rules(int mode) { return mode != mode_base(mode); }
constexpr int mode_with_rules(int base) {
    return base == kModeBits ? kModeRules : base == kModeBytes ? kModeBytesRules : base == kModeLatin1 ? kModeLatin1Rules
         : base == kModeUcs2 ? kModeUcs2Rules : base == kModeValues ? kModeValuesRules : base;
}
constexpr bool mode_is_bytes(int mode) { return mode_base(mode) == kModeBytes || mode_base(mode) == kModeLatin1 || mode_base(mode) == kModeUcs2; }
constexpr bool mode_is_units(int mode) { return mode_base(mode) == kModeLatin1 || mode_base(mode) == kModeUcs2; }
constexpr bool mode_writes_bits(int mode) { return mode_base(mode) == kModeBits || mode_is_bytes(mode); }

constexpr long long kNegInf64 = -(1ll << 60);
constexpr int kWPB = 12;                 // waves per workgroup: 768 threads -> 168 VGPRs per lane, one workgroup per CU
constexpr int kSegMax = kWPB * 64;       // tiles per segment = threads of the block-wide scans

struct Fn64 {        // q transfer function of a run of tiles: f(q) = max(q + a, b); a <= kNegInf64 means constant b
    long long a, b;
};
struct Hd64 {        // head descriptor of a run of tiles: starts before its first closing event, and whether it has one
    long long h;
    int c;
};

// Completion word of a multi-workgroup launch whose outputs land in pinned host memory (small / mid-size host batches): every
// workgroup drains its stores to system scope and counts itself in; the last one stores `seq` into `word` (which the host
// polls instead of waiting for the stream) and resets the counter.  word == NULL: off.
struct DoneSignal {
    unsigned long long* word;
    unsigned long long seq;
    unsigned* counter;     // device memory, 0 between launches
};
#if defined(__HIPCC__)
__device__ __forceinline__ void signal_block_done(const DoneSignal& d) {
    if (!d.word) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = gridDim.x;
        // acq_rel at agent scope: the last workgroup's increment synchronises with every earlier workgroup's, so their
        // (system-fenced) output stores happen-before the word store below -- not only by posted-write ordering
        if (__hip_atomic_fetch_add(d.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n - 1u) {
            *d.counter = 0u;
            __threadfence_system();
            __hip_atomic_store(d.word, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif

struct SplitParams {
    const uint32_t* cps;        // packed UTF-32 code points (16-byte aligned)
    const uint8_t* u8;          // kModeBytes: packed UTF-8 bytes instead (16-byte aligned); kModeLatin1 / kModeUcs2: the code units; cps is unused
    const int64_t* row_off;     // [n_str + 1]
    int64_t n_str, total, n_tiles;
    int seg_tiles;              // tiles per segment (kWPB..kSegMax, see plan_segments)
    int64_t n_segs;             // ceil(n_tiles / seg_tiles)
    const uint8_t* t1;          // stage-1 table in global memory (kStage1Pad bytes; kModeBytes: kB6Stage1Bytes of uint16 block offsets)
    const uint8_t* t2;          // stage-2 split codes in global memory (kStage2Len bytes; kModeBytes: kB6Stage2Bytes), behind t1
    uint64_t* bits_out;         // kModeBits
    uint8_t* values_out;        // kModeValues / kModeBlockMask
    uint64_t* space_out;        // optional (kModeBits): SPACE plane as a bitmask, same layout as bits_out (token spans)
    uint64_t* lead_out;         // optional (kModeBytes): bit i = byte i is a LEAD byte (not 10xxxxxx), same layout as bits_out
                                // (code-point results from the byte-space mask: compact_kernels.hip, k_lead_compress)
    uint16_t* lead_pref_out;    // with lead_out: [words] leads of the tile before each word, and
    int64_t* lead_cnt_out;      //                [n_tiles] leads per tile (what k_word_counts would compute from lead_out)
    uint8_t* codes_out;         // optional (kModeBits / kModeRules on UTF-32 input, t2 = rule codes): the code byte of every char
                                // (featurize: k_features_tiles reads 1 B/char instead of classifying 4 B/char again)
    int64_t* tile_first;        // [n_tiles]: first string that starts at or after each tile's first char (k_tile_index; also read by the compaction passes)
    int4* summ;                 // [n_tiles] {a, b, head_starts, has_closing | edge-block geometry}
    Fn64* seg_fn;               // [n_segs] segment aggregates
    Hd64* seg_hd;               // [n_segs]
    int64_t* fix_count;         // [1] statistics: tiles recomputed by the resolve stage
    // kModeBlockMask only
    const int8_t* bm_a1;        // "starts" bytes [total]
    const int8_t* bm_a2;        // "spaces" bytes [total]
    const int* bm_flags;        // {any(a1), any(a2)}
    DoneSignal done;            // last launch of a host-pointer call on pinned memory (k_one_segment / k_resolve_fix): completion word
    // kModeRules only (t2 then points at the rule-code table)
    lk_rule_tables rules;
};


// featurize on the tile grid (split_kernels.hip: k_features_tiles)
struct FeatParams {
    const uint8_t* codes;         // rule code of every char (SplitParams::codes_out of the tile kernel), padded by one tile + 256 B
    const int64_t* row_off;
    int64_t n_str, total, n_tiles;
    const uint64_t* bits;         // final boundary bitmask
    const uint64_t* kept;         // boundaries whose token is kept (k_word_counts<true>)
    const int64_t* tile_rank;     // index of a tile's first token (exclusive scan of the per-tile token counts)
    const int64_t* tile_cnt;      // tokens per tile
    const uint16_t* word_pref;    // tokens of the tile before each word
    const int64_t* tile_first;    // per tile: first string that starts at or after its first char
    const uint64_t* space;        // SPACE bitmask (only walked for tokens that span more than two words)
    int8_t* features;             // [n_tokens][25]
    void* spans4;                 // [n_tokens][4] = {raw start, raw end, stripped start, stripped end}, string relative; int64 or int32
    bool out32;                   // spans4 holds int32
    const int64_t* n_tokens_dev;  // device: total tokens of the batch (k_word_counts_scan) ...
    int64_t cap;                  // ... nothing is written when it exceeds the caller's capacity
    DoneSignal done;
};
hipError_t launch_features_tiles(const FeatParams& P, int n_cu, hipStream_t st);

void plan_segments(int64_t n_tiles, int n_cu, int* seg_tiles, int64_t* n_segs);
hipError_t launch_tile_index(const SplitParams& P, hipStream_t st);   // stage 0: P.tile_first (must be set)
hipError_t launch_split_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st, bool in_flow = false);   // in_flow: a batch of a flow (deeper prefetch)
hipError_t launch_resolve_fix(const SplitParams& P, int mode, int n_cu, hipStream_t st);
// the three stages in one launch of one workgroup: batches of at most kOneSegTiles tiles planned as ONE segment
// (P.n_segs = 1, P.seg_tiles >= P.n_tiles), UTF-32 bitmask modes (kModeBits / kModeRules)
constexpr int64_t kOneSegTiles = 24;
hipError_t launch_one_segment(const SplitParams& P, int mode, hipStream_t st);
// batches of at most one tile, everything in one launch (split_kernels.hip: k_small_batch); P.t1 / P.t2 / P.rules / P.cps /
// P.row_off / P.n_str / P.total must be set, kind 0 = offsets, 1 = token spans.  done (or NULL): a pinned host word that
// receives seq once every output is visible to the host.
// kind 2 = featurize: items = [n_items][4] span records, features = [n_items][25] sums
hipError_t launch_small_batch(const SplitParams& P, bool rules, int kind, bool out32, void* counts, void* items, int8_t* features,
                              int64_t* n_items, unsigned long long* done, unsigned long long seq, hipStream_t st);
hipError_t launch_any_nonzero(const int8_t* a1, const int8_t* a2, int64_t n, int* flags, hipStream_t st);

// aux_kernels.hip
hipError_t launch_parse_matrix(const uint32_t* cps, int64_t n, const uint8_t* t1, const uint8_t* t2cls,
                               const uint16_t* cw, int8_t* out, hipStream_t st);
// one string of at most kSmallMatrixChars chars in pinned memory: one workgroup; done (or NULL) receives seq after the last store
constexpr int kSmallMatrixChars = 4096;
hipError_t launch_parse_matrix_small(const uint32_t* cps, int n, const uint8_t* t1, const uint8_t* t2cls, const uint16_t* cw,
                                     int8_t* out, unsigned long long* done, unsigned long long seq, hipStream_t st);
// done != NULL: one workgroup, which stores seq into *done after its last output (small arrays in pinned memory)
hipError_t launch_combine_rows(const uint8_t* m, int64_t stride_r, int64_t stride_c, int64_t cols, const int8_t* idx,
                               int idx_ndim, int irows, int icols, int8_t* out, hipStream_t st,
                               unsigned long long* done = nullptr, unsigned long long seq = 0);
// _gen_block_mask of one array pair of at most kTile elements (P.bm_a1 / bm_a2 / values_out / row_off = {0, n} in pinned
// memory, P.total = n): one single-wave launch
hipError_t launch_small_block_mask(const SplitParams& P, unsigned long long* done, unsigned long long seq, hipStream_t st);
hipError_t launch_rebase_rows(int64_t* row, int64_t n, int64_t base, hipStream_t st);
int64_t scan_blocks(int64_t n);   // entries the caller must provide in `block_tot`
hipError_t launch_exclusive_scan(const int64_t* in, int64_t n, int64_t* out, int64_t* total, int64_t* block_tot,
                                 hipStream_t st, int64_t* total_host = nullptr);
// compact_kernels.hip: word-parallel compaction (offsets / token spans / featurize spans)
int64_t count_blocks(int64_t n_words);   // workgroups of launch_word_counts_scan = entries of its `chain` state
hipError_t launch_word_counts_scan(bool spans, const uint64_t* bits, const uint64_t* space, int64_t n_words, int64_t total,
                                   uint64_t* kept, int64_t* tile_cnt, uint16_t* word_pref, int64_t* tile_rank,
                                   unsigned long long* chain, unsigned* ticket, unsigned epoch, int64_t* total_dev,
                                   int64_t* total_host, int* err, hipStream_t st);
hipError_t launch_string_counts(bool out32, const uint64_t* mask, const int64_t* tile_rank, const uint16_t* word_pref,
                                const int64_t* row_off, int64_t n_str, int64_t total, const int64_t* n_items, void* counts, int* err,
                                hipStream_t st);
hipError_t launch_counts_scatter(int kind, bool out32, const uint64_t* bits, const uint64_t* space, const uint64_t* item_mask,
                                 const int64_t* tile_rank, const int64_t* tile_cnt, const uint16_t* word_pref, int64_t n_words,
                                 int64_t total, const int64_t* row_off, int64_t n_str, const int64_t* tile_first, void* out,
                                 const int64_t* n_items_dev, int64_t cap, void* counts, int* err, hipStream_t st, DoneSignal done = DoneSignal{nullptr, 0, nullptr});
// code-point boundary mask + code-point row offsets from the byte-space mask and the lead-byte mask of a UTF-8 batch
// (bmask2 / out_mask2: optionally a second mask -- the SPACE plane -- packed the same way, for token spans in code-point units)
hipError_t launch_lead_compress(const uint64_t* bmask, const uint64_t* bmask2, const uint64_t* lead, const int64_t* tile_rank,
                                const int64_t* tile_cnt, const uint16_t* word_pref, int64_t n_words, int64_t total_bytes,
                                const int64_t* byte_off, int64_t n_str, const int64_t* total_cps_dev, uint64_t* out_mask,
                                uint64_t* out_mask2, int64_t cap_words, int64_t* cp_row_off, int* odd, hipStream_t st);
hipError_t launch_tile_scan(const int64_t* tile_cnt, int64_t n_tiles, int64_t* tile_rank, unsigned long long* chain, unsigned* ticket,
                            unsigned epoch, int64_t* total_dev, int64_t* total_host, int* err, hipStream_t st);
int64_t utf8_blocks(int64_t total_bytes);   // 4 KiB blocks of the chunk-parallel UTF-8 decoder
hipError_t launch_utf8_block_counts(const uint8_t* u8, int64_t total, int64_t* block_cnt, hipStream_t st);
hipError_t launch_utf8_decode(const uint8_t* u8, int64_t total, const int64_t* byte_off, int64_t n_str,
                              const int64_t* block_base, uint16_t* chunk_pref, int64_t total_cps, uint32_t* cps,
                              int64_t* cp_off, hipStream_t st);
hipError_t launch_widen_units(const void* units, int kind, int64_t n, uint32_t* cps, hipStream_t st);   // kind 1 / 2 -> UTF-32
hipError_t launch_corpus_fill(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off,
                              uint32_t* cps, hipStream_t st);
hipError_t launch_stream_read(const void* src, int64_t bytes, uint32_t* sink, int n_cu, hipStream_t st);
hipError_t launch_utf8_bytes(const uint32_t* cps, int64_t n, unsigned long long* total, hipStream_t st);

}  // namespace latok
#endif
