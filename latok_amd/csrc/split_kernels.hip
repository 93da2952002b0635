// split_kernels.hip -- the fused character-feature + split-mask kernels for gfx950 (MI355X, CDNA4).
//
// What the reference does per string with an n x 25 int8 matrix and five passes over it
// (reference latok/core/src/latok/latok.c:31-138 gen_parse_matrix, :275-370 combine_matrix_rows x3, :140-258
// gen_block_mask, glued by latok/core/default_tokenizer.py:113-134) is done here in ONE pass over the packed
// UTF-32 batch, without ever materialising the matrix: 4 B read and 1 bit written per character.
//
// Work decomposition: the packed code-point buffer is cut into fixed tiles of 4096 chars = 64 words of 64 chars.
// One wavefront (64 lanes) owns one tile at a time:
//   phase 1 (lane = 4 consecutive chars, coalesced):  16 x global_load_dwordx4 (1 KiB per wave instruction) ->
//            two-stage Unicode class lookup in LDS -> one 8-bit "split code" per char -> wave-private LDS staging.
//   phase 2 (lane = one 64-char word):  lane reads its 64 code bytes back (4 x ds_read_b128, 80-byte padded rows,
//            conflict-free), bit-slices them into 8 feature planes, and evaluates all rules as 64-bit boolean algebra
//            (lane_math.h).  PREV/NEXT/AFTER_NEXT features are word shifts plus three neighbour bytes from LDS.
//   block mask: exact queue semantics of gen_block_mask via carry-propagating adds; cross-lane state is a (max,+)
//            scan over lanes (forward) and a carry chain over lane ballots (backward).
// Tiles are not string-aligned, so a block (whitespace-delimited span) may straddle tiles.  Each tile is first
// computed assuming no pending start enters it and with a provisional decision for its open tail block, and
// publishes a 16-byte summary (k_tiles_main); k_resolve_fix then resolves the two unknowns per tile exactly (block-wide
// scans over the tile summaries of a segment + the aggregates of the other segments) and repairs the few tiles whose
// assumption was wrong: a patch of the bitmask in place for the common cases, else a recomputation by the same tile
// code with the exact inputs.  Stage 0 (k_tile_index) gives every tile the first string that starts in it.
//
// No MFMA: this is integer/bit work bounded by HBM reads (4 B/char), not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "bitscan.h"
#include "kernels.h"
#include "lane_math.h"
#include "utf8_decode.h"

namespace latok {

// Diagnostic build only (-DLATOK_STAMPS): s_memtime stamps at phase boundaries of the tile function, summed per wave and
// added to a global array at the end of the kernel.  Never compiled into the shipped library.
#ifdef LATOK_STAMPS
__device__ unsigned long long g_stamp_sum[16];
__device__ unsigned long long g_stamp_cnt;
#define LATOK_STAMP(k)                                                                           \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((k) > 0) stamp_acc[(k)] += t_ - stamp_prev;                                          \
        stamp_prev = t_;                                                                         \
    } while (0)
#else
#define LATOK_STAMP(k) do { } while (0)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// staging layout: 64-byte rows padded to 80 bytes: byte p of the tile lives at p + 16 * (p / 64)
__device__ __forceinline__ uint32_t stage_addr(uint32_t p) { return p + ((p >> 6) << 4); }

__device__ __forceinline__ uint32_t classify1(const uint8_t* t1, const uint8_t* t2, uint32_t cp) {
    const uint32_t hi = min(cp >> kTblShift, (uint32_t)(kStage1Len - 1));
    const uint32_t blk = t1[hi];
    return t2[(blk << kTblShift) | (cp & ((1u << kTblShift) - 1u))];
}

// byte space: the same through its own table (stage 1 by cp >> 6 as uint16 block offsets; kernels.h: kB6*)
__device__ __forceinline__ uint32_t classify1_b6(const uint8_t* t1b, const uint8_t* t2b, uint32_t cp) {
    const uint32_t hi = min(cp >> LK_B6_SHIFT, (uint32_t)(kB6Stage1Len - 1));
    const uint32_t off = *reinterpret_cast<const uint16_t*>(t1b + 2u * hi);
    return t2b[off | (cp & 63u)];
}

__device__ __forceinline__ uint32_t classify4(const uint8_t* t1, const uint8_t* t2, u32x4 v, bool* not_ascii = nullptr) {
    // wave-uniform fast path: all 256 chars of this wave instruction are ASCII -> stage-2 block 0, no stage-1 lookup.
    // Either way the four lookups of a table level are requested together (the empty asm pins them): left alone hipcc shares the
    // fourth lookup between the two branches and strings the others along -- two LDS round trips per ASCII row instead of one,
    // five per non-ASCII row instead of two.
    const bool ascii = __all((v.x | v.y | v.z | v.w) < 128u);
    if (not_ascii && !ascii) *not_ascii = true;
    uint32_t c0, c1, c2, c3;
    if (ascii) {
        c0 = t2[v.x]; c1 = t2[v.y]; c2 = t2[v.z]; c3 = t2[v.w];
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    } else {
        const uint32_t last = (uint32_t)(kStage1Len - 1), low = (1u << kTblShift) - 1u;
        uint32_t b0 = t1[min(v.x >> kTblShift, last)], b1 = t1[min(v.y >> kTblShift, last)],
                 b2 = t1[min(v.z >> kTblShift, last)], b3 = t1[min(v.w >> kTblShift, last)];
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        c0 = t2[(b0 << kTblShift) | (v.x & low)]; c1 = t2[(b1 << kTblShift) | (v.y & low)];
        c2 = t2[(b2 << kTblShift) | (v.z & low)]; c3 = t2[(b3 << kTblShift) | (v.w & low)];
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    }
    return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS traffic of one wave is executed in issue order; this only stops the compiler from reordering across it
    // fences restricted to the LDS address space: a plain wavefront fence makes hipcc drain vmcnt(0) as well, i.e.
    // wait for this tile's output store (and any load in flight) before the next tile's loads can be issued
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// ---- cross-lane helpers on DPP / readlane (no LDS round trip, unlike ds_bpermute-based __shfl) -----------------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl=false): lanes without a valid source keep `old`.
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int lane_read(int v, int uniform_lane) { return __builtin_amdgcn_readlane(v, uniform_lane); }
// values that are wave-uniform by construction but live in VGPRs: move them to SGPRs
__device__ __forceinline__ int to_scalar(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t to_scalar64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t lane_read64(int64_t v, int uniform_lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, uniform_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), uniform_lane);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_mov<kDppRowShr1, 0xF>(0, v);
    v += dpp_mov<kDppRowShr2, 0xF>(0, v);
    v += dpp_mov<kDppRowShr4, 0xF>(0, v);
    v += dpp_mov<kDppRowShr8, 0xF>(0, v);   // lane 15 of every row now holds its row's sum
    return lane_read(v, 15) + lane_read(v, 31) + lane_read(v, 47) + lane_read(v, 63);
}

// inclusive prefix maximum over the 64 lanes of values >= `floor` (six DPP steps; the __shfl_up form is six dependent
// ds_bpermute round trips through the LDS pipe, ~0.7 us per call at the occupancy of the featurize kernel)
__device__ __forceinline__ int wave_scan_max(int v, int floor) {
    v = max(v, dpp_mov<kDppRowShr1, 0xF>(floor, v));
    v = max(v, dpp_mov<kDppRowShr2, 0xF>(floor, v));
    v = max(v, dpp_mov<kDppRowShr4, 0xF>(floor, v));
    v = max(v, dpp_mov<kDppRowShr8, 0xF>(floor, v));
    v = max(v, dpp_mov<kDppRowBcast15, 0xA>(floor, v));
    v = max(v, dpp_mov<kDppRowBcast31, 0xC>(floor, v));
    return v;
}
// maximum over the wave, in every lane (wave-uniform)
__device__ __forceinline__ int wave_max(int v, int floor) { return lane_read(wave_scan_max(v, floor), 63); }

// inclusive scan over the 64 lanes of the queue transfer functions, earlier lanes applied first; (0,0) is neutral
// for the functions that occur here (a <= b, b >= 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ lk_qfn qfn_scan_step(lk_qfn inc) {
    lk_qfn o;
    o.a = dpp_mov<CTRL, ROW_MASK>(0, inc.a);
    o.b = dpp_mov<CTRL, ROW_MASK>(0, inc.b);
    return lk_qfn_then(o, inc);
}
__device__ __forceinline__ lk_qfn qfn_wave_scan(lk_qfn f) {
    f = qfn_scan_step<kDppRowShr1, 0xF>(f);
    f = qfn_scan_step<kDppRowShr2, 0xF>(f);
    f = qfn_scan_step<kDppRowShr4, 0xF>(f);
    f = qfn_scan_step<kDppRowShr8, 0xF>(f);
    f = qfn_scan_step<kDppRowBcast15, 0xA>(f);   // rows 1 and 3 take the total of the row before them
    f = qfn_scan_step<kDppRowBcast31, 0xC>(f);   // rows 2 and 3 take the total of rows 0..1
    return f;
}

// ---------------------------------------------------------------------------------------------------------------
// the tile function
// ---------------------------------------------------------------------------------------------------------------
struct TileLds {
    const uint8_t* t1;     // stage-1 table (LDS)
    const uint8_t* t2;     // stage-2 table of split codes (LDS)
    uint8_t* stage;        // kStageBytes, wave private
    uint8_t* halo;         // 16 bytes, wave private
    lk_u64* bw;            // 65 words of string-start bits, wave private
    const uint8_t* lut;    // kModeLatin1: slice LUT (kSliceLutBytes) in place of the Unicode tables
    const uint8_t* ctab;   // kModeLatin1: split code of each of the 256 Latin-1 chars (kModeBytes: the stage-2 block of U+0000, for its ASCII tiles)
    const uint8_t* ltab;   // kModeBytes: decode table of the multi-byte lead bytes (kLeadTabBytes, build_lead_table)
    const uint8_t* t1b;    // kModeBytes: its own class table (kernels.h: kB6*) -- stage 1, uint16 block offsets by cp >> 6 ...
    const uint8_t* t2b;    //             ... and stage 2, 64-entry blocks (t1 / t2 are unused there; ctab = t2b: ASCII is blocks 0, 1)
    uint8_t* tables;       // the workgroup's table area (LDS offset 0) and the two counters of the on-demand table load
    int* ctl;              // (tables_ensure); ctl == nullptr: the kernel reads its tables from global memory
    uint64_t* small_bits;  // k_small_batch: where the tile's boundary / SPACE words go (LDS) in place of P.bits_out /
    uint64_t* small_space; // P.space_out -- the kernel arguments stay where they are (no private copy of the rule tables)
};

// ---------------------------------------------------------------------------------------------------------------
// Latin-1 input: classification and bit-slicing by ONE table.  A lane holds 64 raw bytes; what phase 2 needs from
// them are the 8 planes of their split codes.  Looking a byte up in the code table and then transposing 8 x 64 bits
// with shifts and masks costs ~7 VALU instructions per char; instead the table itself holds the code already spread
// out -- entry c = {lo, hi}: bit 8 b of lo = bit b of code(c) (b < 4), of hi = bit 4 + b -- so that OR-ing the entries
// of 8 consecutive chars, each shifted by its position, gives exactly the byte-per-plane words lk_bitslice64 has
// after its three delta-swap stages.  The shift is folded into the table: 8 pre-shifted copies (2 KiB each), the
// copy is selected by the immediate offset of the LDS instruction -> per char one address computation, two
// ds_read_b32 and one v_or3 shared by two values.  (The narrow-input kernels are VALU-bound, not HBM-bound: 1 B/char
// in, and the rule algebra per char is the same as for UTF-32.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSliceHiOff = 1024 + 64;                   // hi[c] sits 16 banks away from lo[c]: the two reads of a char never collide
constexpr int kSliceCopyBytes = kSliceHiOff + 1024;      // {lo[256], pad, hi[256]} as uint32
constexpr int kSliceLutBytes = 8 * kSliceCopyBytes;      // copy j = entries << j
static_assert(kSliceLutBytes + 256 <= kTablesLdsBytes, "the Latin-1 tables live where the Unicode tables would");

constexpr int kSlicePin = 2;   // groups of 8 chars whose lookups are requested together (slice_lut64)
__device__ __forceinline__ void slice_lut64(const uint32_t (&d)[16], const uint8_t* lut, lk_u64 (&plane)[8]) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        uint32_t l = 0, h = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = d[2 * g + (j >> 2)];
            const uint32_t off = ((w >> (8 * (j & 3))) & 0xFFu) << 2;
            const uint8_t* e = lut + j * kSliceCopyBytes + off;
            l |= *reinterpret_cast<const uint32_t*>(e);
            h |= *reinterpret_cast<const uint32_t*>(e + kSliceHiOff);
        }
        lo[g] = l;
        hi[g] = h;
        // pin the words every kSlicePin groups: without it the compiler requests all 128 lookups first (one result
        // register each, spills) and ORs them afterwards
        if ((g + 1) % kSlicePin == 0) {
#pragma unroll
            for (int q = g + 1 - kSlicePin; q <= g; ++q) asm volatile("" : "+v"(lo[q]), "+v"(hi[q]));
        }
    }
    lk_planes_from_groups(lo, hi, plane);
}

// split code of Latin-1 char c from the global tables (U+0000..U+00FF live in the first two stage-2 blocks)
__device__ __forceinline__ uint32_t latin1_code_global(const SplitParams& P, uint32_t c) {
    const uint32_t blk = P.t1[c >> kTblShift];
    return P.t2[(blk << kTblShift) | (c & ((1u << kTblShift) - 1u))];
}

// build [slice LUT | code table] in LDS (NT threads)
template <int NT>
__device__ __forceinline__ void build_latin1_tables(uint8_t* lds, const SplitParams& P) {   // lds = where the LUT goes
    for (int c = threadIdx.x; c < 256; c += NT) {
        const uint32_t code = latin1_code_global(P, (uint32_t)c);
        lds[kSliceLutBytes + c] = (uint8_t)code;
        const uint32_t lo = ((code & 15u) * 0x00204081u) & 0x01010101u;
        const uint32_t hi = ((code >> 4) * 0x00204081u) & 0x01010101u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t* copy = reinterpret_cast<uint32_t*>(lds + j * kSliceCopyBytes);
            copy[c] = lo << j;
            copy[kSliceHiOff / 4 + c] = hi << j;
        }
    }
}

// One tile = 4096 chars, one wave.  (Register prefetch of the next tile -- full, half, quarter; 8/10/12/16 waves per
// CU -- was measured and gives nothing: see DESIGN.md, so the tile function stays simple.)
// idx0 = index of the first string that starts at or after the tile's first char.
// With write_summary the tile summary is written to *summ_l (LDS copy of the segment).
// Returns this lane's 64-bit boundary word (kModeBits); with DEFER the caller stores it later (write combining).
// ---------------------------------------------------------------------------------------------------------------
// kModeBytes: the tile is 4096 BYTES of UTF-8; a char lives at its lead byte.  An all-ASCII tile (the common case) is one
// table lookup per byte.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool u8_is_cont(uint32_t b) { return (b & 0xC0u) == 0x80u; }

// bit i = byte i of the dword has its top bit set (is not ASCII)
__device__ __forceinline__ uint32_t u8_high_nibble(uint32_t w) { return ((((w >> 7) & 0x01010101u) * 0x00204081u) >> 21) & 0xFu; }

// kModeLatin1 / kModeUcs2 (PEP 393 kinds 1 / 2: Latin-1 / UCS-2 code units), phase 1: the tile is 4096 CHARS of 1 or
// 2 bytes each.  Nothing is decoded and there are no continuation bytes, so the staging buffer receives plain codes and
// phase 2 evaluates the char-space rules (lk_rules), exactly like a UTF-32 tile; only the LDS layout is the byte-space one.
// Halo: halo[0] = code of unit t0-1, halo[8], halo[9] = codes of units t0+4096, t0+4097 (0 where there is no such char).
// Returns (KIND 1 only) whether every byte of the tile is ASCII (wave-uniform): phase 2 then classifies without a table.
template <int KIND>
__device__ __forceinline__ bool units_phase1(const SplitParams& P, const TileLds& L, int64_t t0, int lane) {
    const int64_t total = P.total;
    if (lane < 2) *reinterpret_cast<lk_u64*>(L.halo + 8u * lane) = 0ull;
    uint32_t halo_u = 0xFFFFFFFFu;                                              // out of range -> class 0
    if (lane < 3) {
        const int64_t hp = lane == 0 ? t0 - 1 : t0 + kTile + (lane - 1);
        if (hp >= 0 && hp < total)
            halo_u = KIND == 1 ? (uint32_t)P.u8[hp] : (uint32_t)reinterpret_cast<const uint16_t*>(P.u8)[hp];
    }
    if (KIND == 1) {
        // Latin-1: the RAW bytes go to the staging buffer (phase 2 classifies and bit-slices them with one table,
        // slice_lut64); only the three halo chars are classified here
        u32x4 v[4];
        if (t0 + kTile <= total) {
            const u32x4* src = reinterpret_cast<const u32x4*>(P.u8 + t0) + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
        } else {
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                uint32_t d[4] = {0, 0, 0, 0};
                const int64_t p = t0 + 1024 * i + 16 * lane;
                for (int j = 0; j < 16; ++j)
                    if (p + j < total) d[j >> 2] |= (uint32_t)P.u8[p + j] << (8 * (j & 3));
                v[i].x = d[0]; v[i].y = d[1]; v[i].z = d[2]; v[i].w = d[3];
            }
        }
        wave_lds_sync();   // the zero stores to the halo are ordered before the halo stores below
        // positions past the end of the batch hold unit 0 here; their codes are masked by `valid` in phase 2
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<uint4*>(L.stage + stage_addr(1024u * i + 16u * lane)) = make_uint4(v[i].x, v[i].y, v[i].z, v[i].w);
        if (lane < 3 && halo_u != 0xFFFFFFFFu) L.halo[lane == 0 ? 0 : 7 + lane] = L.ctab[halo_u & 0xFFu];
        uint32_t hi_bits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) hi_bits |= (v[i].x | v[i].y | v[i].z | v[i].w) & 0x80808080u;
        return __all(hi_bits == 0u);
    } else {
        // UCS-2: 8 units per 16-byte load, row i of the tile = units 512 i + 8 lane ..
        u32x4 v[8];
        if (t0 + kTile <= total) {
            const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(P.u8) + t0) + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
        } else {
            const uint16_t* __restrict__ u16 = reinterpret_cast<const uint16_t*>(P.u8);
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                uint32_t d[4] = {0, 0, 0, 0};   // units past the end read as 0, like the bytes of a UTF-8 tail tile
                const int64_t p = t0 + 512 * i + 8 * lane;
                for (int j = 0; j < 8; ++j)
                    if (p + j < total) d[j >> 1] |= (uint32_t)u16[p + j] << (16 * (j & 1));
                v[i].x = d[0]; v[i].y = d[1]; v[i].z = d[2]; v[i].w = d[3];
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t d[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            uint32_t lo, hi;
            if (__all(((d[0] | d[1] | d[2] | d[3]) & 0xFF80FF80u) == 0u)) {      // 512 ASCII chars: stage-2 block of U+0000
                const uint32_t off0 = (uint32_t)L.t1[0] << kTblShift;
                lo = (uint32_t)L.t2[off0 + (d[0] & 0xFFFFu)] | ((uint32_t)L.t2[off0 + (d[0] >> 16)] << 8) |
                     ((uint32_t)L.t2[off0 + (d[1] & 0xFFFFu)] << 16) | ((uint32_t)L.t2[off0 + (d[1] >> 16)] << 24);
                hi = (uint32_t)L.t2[off0 + (d[2] & 0xFFFFu)] | ((uint32_t)L.t2[off0 + (d[2] >> 16)] << 8) |
                     ((uint32_t)L.t2[off0 + (d[3] & 0xFFFFu)] << 16) | ((uint32_t)L.t2[off0 + (d[3] >> 16)] << 24);
            } else {
                lo = classify1(L.t1, L.t2, d[0] & 0xFFFFu) | (classify1(L.t1, L.t2, d[0] >> 16) << 8) |
                     (classify1(L.t1, L.t2, d[1] & 0xFFFFu) << 16) | (classify1(L.t1, L.t2, d[1] >> 16) << 24);
                hi = classify1(L.t1, L.t2, d[2] & 0xFFFFu) | (classify1(L.t1, L.t2, d[2] >> 16) << 8) |
                     (classify1(L.t1, L.t2, d[3] & 0xFFFFu) << 16) | (classify1(L.t1, L.t2, d[3] >> 16) << 24);
            }
            *reinterpret_cast<uint2*>(L.stage + stage_addr(512u * i + 8u * lane)) = make_uint2(lo, hi);
        }
        if (lane < 3 && halo_u != 0xFFFFFFFFu) L.halo[lane == 0 ? 0 : 7 + lane] = (uint8_t)classify1(L.t1, L.t2, halo_u);
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// Class tables on demand (k_tiles_main, byte space).  A workgroup used to copy its tables to LDS before its first tile: 60 KB
// per CU from L2 with nothing else in flight, ~2 us at the head of every launch -- for tables that a batch of ASCII text never
// reads beyond the 128 codes of U+0000..U+007F.  Now the launch copies those 128 bytes, and the first wave that meets a
// multi-byte char brings in the rest: the copy is cut into kLazyGrabs pieces handed out by an LDS counter, so every wave that
// arrives while it is under way takes a share (text that is not ASCII anywhere: all twelve at once, as fast as before), and a
// second counter says when the last piece is in place.  Waves that find both counters full pay one LDS read per tile.
// (Both counters count LANES, 64 per piece: every lane of the wave executes the same atomic -- hipcc folds them into one
// ds_add of 64 per wave -- so no lane-0 branch surrounds the wave-wide copy instructions.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLazyPer = 4;                                    // 1 KiB rows (one global_load_lds_dwordx4 of the wave each) per piece

constexpr int kLazyRows = kB6TablesBytes / 1024;                 // byte space: [stage 1 | stage 2], then (computed, the last piece) the byte decode table
constexpr int kLazyGrabs = (kLazyRows + kLazyPer - 1) / kLazyPer + 1;
__device__ __attribute__((noinline, cold)) void tables_fetch_bytes(const uint8_t* t1, const uint8_t* t2, uint8_t* tables, int* ctl);
__device__ __forceinline__ void tables_ensure_bytes(const SplitParams& P, const TileLds& L, int lane) {
    if (L.ctl == nullptr) return;
    if (__hip_atomic_load(&L.ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= 64 * kLazyGrabs) return;
    tables_fetch_bytes(P.t1, P.t2, L.tables, L.ctl);
}

// kModeBytes, phase 1 (lane = 16 consecutive bytes per 1 KiB row).  What reaches the staging buffer: the split code of
// its char at every LEAD byte (any non-continuation byte), the marker LK_CODE_CONT at continuation bytes (the continuation
// plane of a word then falls out of phase 2's bit-slicing).  Nothing is carried from lane to lane: the continuation bytes take their owner's code in phase 2, as
// mask arithmetic on the word's planes (lane_math.h: lk_smear_planes).  An ASCII byte is one table lookup; only the
// NON-ASCII LEAD bytes are decoded and classified through the two-stage table -- two slots per dword (well-formed UTF-8
// has at most two multi-byte leads in 4 bytes), all eight slots of a row independent and branch-free so that their LDS
// lookups overlap; a dword with more (malformed input) takes a wave-uniform loop afterwards.
// Halo: halo[0] = code of the char that owns byte t0-1, halo[4] = how many more continuation bytes it may take,
// halo[8..15] = staging bytes of the 8 bytes after the tile.
// the window of slot (dword Q, lead mask m within the dword): its 4 bytes and where in the dword the lead sits
// the window of a slot of dword Q: m = lead mask within the dword in "bit 7 of the byte" form; the slot takes its lowest
// lead: *r8_out = the bit position of that byte in the dword, returns the 4 bytes from there on
// ---------------------------------------------------------------------------------------------------------------
// kModeBytes: class of a multi-byte char straight from its bytes (lane_math.h: lk_lead_index has the scheme).  The code point is
// never assembled: byte space has its own two-stage class table cut at 6 bits, so stage 1 wants "every byte but the last" and
// stage 2 the last byte's payload.  Per decode slot: one ds_read_b64 of the 8-byte entry of the window's first byte, v_perm,
// v_dot4_u32_u8 (the stage-1 offset), a clamp, the "cut short" test (2), ds_read_u16, v_and_or (stage-2 index), ds_read_u8 --
// 7 VALU instructions where utf8_cp_of + classify1 took 27 and the 7-bit table (hi = cp >> 7, lo = cp & 127 by shifts and masks)
// 15.  Same results as utf8_cp_of + classify1: a sequence cut short is U+FFFD, overlong / surrogate forms decode as they are,
// 0xF8..0xFF are 4-byte leads with 3 payload bits.  The table has an entry for EVERY byte value: a slot that holds no lead
// decodes whatever byte its window starts at, and the entries below 0xC0 yield code 0 (nothing to OR into the staging bytes).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLeadTabBytes = 256 * 8;
template <int NT>
__device__ __forceinline__ void build_lead_table(uint8_t* lds) {
    for (int i = threadIdx.x; i < 256; i += NT) {
        const lk_lead_entry e = lk_lead_entry_of((uint32_t)i);
        reinterpret_cast<uint2*>(lds)[i] = make_uint2(e.sel, e.hi0);
    }
}
// the entry of the byte a slot's window W starts with
__device__ __forceinline__ uint2 lead_entry(const uint8_t* ltab, uint32_t W) {
    return *reinterpret_cast<const uint2*>(ltab + ((W & 0xFFu) << 3));
}
// W = the 4 bytes from a slot's first byte on (memory order), q = its entry: *off2 = byte offset of the char's stage-1 entry
// (clamped), *R = the sequence in lk_lead_index's order (low 6 bits = stage-2 index); returns whether the sequence is cut short
__device__ __forceinline__ bool lead_index(uint2 q, uint32_t W, uint32_t* off2, uint32_t* R) {
    lk_lead_entry e;
    e.sel = q.x; e.hi0 = q.y;
    uint32_t o;
    const bool bad = lk_lead_index(e, W, &o, R);
    *off2 = min(o, 2u * (uint32_t)(kB6Stage1Len - 1));
    return bad;
}

template <int Q>
__device__ __forceinline__ uint32_t bytes_slot_window(const uint32_t (&w)[5], uint32_t m, uint32_t* r8_out) {
    const uint32_t r8 = (uint32_t)__builtin_ctz(m | 0x80000000u) & 24u;   // bit position of the byte; m == 0: byte 3, whatever it is
    *r8_out = r8;
    return __builtin_amdgcn_alignbit(w[Q + 1], w[Q], r8);
}

// byte space: the halo bytes of the tile at t0 -- lane 0: the dword before the tile, lanes 1..11: the 11 bytes after it
__device__ __forceinline__ uint32_t bytes_halo_load(const uint8_t* __restrict__ u8, int64_t t0, int64_t total, int lane) {
    uint32_t hb = 0;
    if (lane == 0) {
        if (t0 > 0) hb = *reinterpret_cast<const uint32_t*>(u8 + t0 - 4);
    } else if (lane < 12) {
        const int64_t q = t0 + kTile + (lane - 1);
        if (q < total) hb = u8[q];
    }
    return hb;
}

constexpr int kCpsPrefetchRows = 2;   // rows (1 KiB) of the wave's next UTF-32 tile requested before phase 2 of the current one
// ... and in the tile kernel of a FLOW batch (two batches in flight, each planned for 7/8 of the CUs): six.  Same box, R = 2 / 4 / 5 / 6:
// C2 through the flow 1 424 / 1 434 / 1 439 / 1 448 GB/s (sustained 1 472 / 1 487 / 1 489 / 1 500), C3 2 450 -> 2 585 (+5 %) -- but
// one batch at a time 1 213 -> 1 205 on C2, 2 306 -> 2 266 on C3, the isolated kernel 94.5 -> 95.2 us: the depth that pays while another
// kernel shares the memory system costs a little when the kernel is alone, so the launch scheme picks the instantiation.
constexpr int kCpsPrefetchRowsFlow = 6;
constexpr int kCpsPrefetchMax = kCpsPrefetchRowsFlow;
struct CpsPrefetch {
    u32x4 v[kCpsPrefetchMax];
    bool valid;
};

__device__ __forceinline__ bool bytes_phase1(const SplitParams& P, const TileLds& L, int64_t t0, int lane
#ifdef LATOK_STAMPS
                                             , unsigned long long* stamp_acc, unsigned long long& stamp_prev
#endif
                                             ) {
    const int64_t total = P.total;
    const uint8_t* __restrict__ u8 = P.u8;
    // halo bytes: lane 0 holds the dword before the tile (t0 is a multiple of 4096 and u8 is 16-byte aligned; byte j of it
    // = byte t0 - 4 + j), lanes 1..11 the 11 bytes after the tile (0 where the batch has ended).  Requested BEFORE the rows: loads
    // come back in order, and the decision below wants the halo and row 0 only.
    uint32_t hb = bytes_halo_load(u8, t0, total, lane);
    // the tile: 4 x 16 bytes per lane (row i covers bytes 1024 i + 16 lane ..)
    u32x4 v[4];
    if (t0 + kTile <= total) {
        const u32x4* src = reinterpret_cast<const u32x4*>(u8 + t0) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
    } else {
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            uint32_t d[4] = {0, 0, 0, 0};
            const int64_t p = t0 + 1024 * i + 16 * lane;
            for (int j = 0; j < 16; ++j)
                if (p + j < total) d[j >> 2] |= (uint32_t)u8[p + j] << (8 * (j & 3));
            v[i].x = d[0]; v[i].y = d[1]; v[i].z = d[2]; v[i].w = d[3];
        }
    }
    // All-ASCII tiles take their own road, which needs all four rows; a tile whose row 0 already holds a multi-byte char does not
    // wait for the others to find that out.  (Requesting row 0 and the halo bytes of the wave's NEXT tile during phase 2 was
    // measured on top of this: C3 0.505 -> 0.515 ms, C2 0.073 -> 0.077; the 16 extra bytes of scratch cost more than the wait.)
    uint32_t hi_bits = (hb | v[0].x | v[0].y | v[0].z | v[0].w) & 0x80808080u;
    if (__all(hi_bits == 0u)) {
#pragma unroll
        for (int i = 1; i < 4; ++i) hi_bits |= (v[i].x | v[i].y | v[i].z | v[i].w) & 0x80808080u;
    }
    if (lane < 2) *reinterpret_cast<lk_u64*>(L.halo + 8u * lane) = 0ull;
    const bool all_ascii = __all(hi_bits == 0u);
    wave_lds_sync();   // the zero stores are ordered before everything below
    LATOK_STAMP(11);   // (share of stamp 2: the tile's bytes have arrived)

    if (all_ascii) {
        // no multi-byte char in or around the tile (the common case): the RAW bytes go to the staging buffer and phase 2
        // classifies and bit-slices them with one table (slice_lut64), exactly like a Latin-1 tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<uint4*>(L.stage + stage_addr(1024u * i + 16u * lane)) = make_uint4(v[i].x, v[i].y, v[i].z, v[i].w);
        if (lane == 0 && t0 > 0) L.halo[0] = L.ctab[hb >> 24];
        if (lane >= 1 && lane < 9 && t0 + kTile + (lane - 1) < total) L.halo[8 + (lane - 1)] = L.ctab[hb & 0xFFu];
        return true;
    }

    tables_ensure_bytes(P, L, lane);   // from here on the whole class table is read, not just its ASCII part
    // The char that owns byte t0-1: its lead is byte t0-k, k = 1..4 (further back: nobody owns it).  The 8 bytes
    // t0-4 .. t0+3 sit in lane 0's registers; every lane computes (no divergence), lane 0 stores.
    {
        const bool c1 = u8_is_cont(hb >> 24), c2 = u8_is_cont((hb >> 16) & 0xFFu), c3 = u8_is_cont((hb >> 8) & 0xFFu);
        const uint32_t k = !c1 ? 1u : (!c2 ? 2u : (!c3 ? 3u : 4u));
        const lk_u64 Z = (lk_u64)hb | ((lk_u64)v[0].x << 32);
        const uint32_t W = (uint32_t)(Z >> (8u * (4u - k)));
        const uint32_t b0 = W & 0xFFu;
        const uint32_t code = classify1_b6(L.t1b, L.t2b, b0 < 0x80u ? b0 : utf8_cp_of(W));
        if (lane == 0 && t0 > 0 && !u8_is_cont(b0)) {
            L.halo[0] = (uint8_t)code;
            L.halo[4] = (uint8_t)(4u - k);
        }
    }

    const uint32_t code_fffd = classify1_b6(L.t1b, L.t2b, 0xFFFDu);   // a sequence that is cut short
    LATOK_STAMP(12);   // (share of stamp 2: owner of the byte before the tile)
    // R rows (1 KiB each) per round, every stage over all of them: the table lookups of a round -- R x 16 ASCII, then R x 8 per
    // level of the multi-byte decode (byte entry, stage 1, stage 2) -- are in flight together, so a round is four trips to the
    // LDS whatever R is.  (Stamped build on C3: the rows were 3.7 K clocks each for ~250 VALU instructions -- the wave sat in the
    // LDS latency of one row at a time.)
    constexpr int R = 1;   // (2 rows per round: 128 B more scratch, 0.505 -> 0.56 ms on C3; 4: 0.78 KB of scratch, 1.4 ms)
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += R) {
        uint32_t d[R][4], out[R][4];
#pragma unroll
        for (int a = 0; a < R; ++a) { d[a][0] = v[i0 + a].x; d[a][1] = v[i0 + a].y; d[a][2] = v[i0 + a].z; d[a][3] = v[i0 + a].w; }
        // every byte as if it were a char of its own: one lookup each in code[256] (the ASCII codes; 0 from 0x80 on, so the
        // positions of multi-byte chars come back empty)
#pragma unroll
        for (int a = 0; a < R; ++a) {
            // all 16 lookups of the row requested before the first one is used: left to itself hipcc keeps two or three in flight
            // (a register each), and the row waits for the LDS eight times instead of once
            uint32_t c[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = L.ctab[(d[a][k >> 2] >> (8 * (k & 3))) & 0xFFu];
            asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                              "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]));
#pragma unroll
            for (int j = 0; j < 4; ++j) out[a][j] = c[4 * j] | (c[4 * j + 1] << 8) | (c[4 * j + 2] << 16) | (c[4 * j + 3] << 24);
        }
        uint32_t any_hi = 0;
#pragma unroll
        for (int a = 0; a < R; ++a) any_hi |= (d[a][0] | d[a][1] | d[a][2] | d[a][3]) & 0x80808080u;
        if (__all(any_hi == 0u)) {        // these rows are pure ASCII
#pragma unroll
            for (int a = 0; a < R; ++a)
                *reinterpret_cast<uint4*>(L.stage + stage_addr(1024u * (i0 + a) + 16u * lane)) = make_uint4(out[a][0], out[a][1], out[a][2], out[a][3]);
            continue;
        }
        // bytes 16..18 after my chunk: the next lane's first dword; lane 63: lane 0's next row, or the bytes after the tile
        uint32_t w[R][5];
#pragma unroll
        for (int a = 0; a < R; ++a) {
            const int i = i0 + a;
            uint32_t nx = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)d[a][0]);
            const uint32_t wrap = i < 3 ? (uint32_t)lane_read((int)v[(i + 1) & 3].x, 0)
                                        : ((uint32_t)lane_read((int)hb, 1) | ((uint32_t)lane_read((int)hb, 2) << 8) |
                                           ((uint32_t)lane_read((int)hb, 3) << 16));
            if (lane == 63) nx = wrap;
            w[a][0] = d[a][0]; w[a][1] = d[a][1]; w[a][2] = d[a][2]; w[a][3] = d[a][3]; w[a][4] = nx;
        }
        // Per dword, all masks in "bit 7 of the byte" form (no bit gathering): hi = not ASCII, cont = 10xxxxxx, nl = the
        // leads that need a decode.  (Bytes past the end of the batch were loaded as 0: ASCII, never a continuation.)
        uint32_t m1[R][4], m2[R][4], rest[R][4];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t hi = d[a][q] & 0x80808080u;
                const uint32_t cont = hi & ~(d[a][q] << 1);
                out[a][q] |= cont;                                              // ASCII codes | LK_CODE_CONT at continuation bytes
                m1[a][q] = hi ^ cont;
                m2[a][q] = m1[a][q] & (m1[a][q] - 1u);
                rest[a][q] = m2[a][q] & (m2[a][q] - 1u);                        // leads beyond two per dword (malformed input)
            }
        {
            // stage by stage over the 8 R slots (slot s < 4: the first lead of dword s, else the second of dword s - 4)
            uint32_t r8[R][8], W[R][8], off2[R][8], Rs[R][8], blk[R][8], code[R][8];
            bool bad[R][8];
#pragma unroll
            for (int a = 0; a < R; ++a) {
                W[a][0] = bytes_slot_window<0>(w[a], m1[a][0], &r8[a][0]);
                W[a][1] = bytes_slot_window<1>(w[a], m1[a][1], &r8[a][1]);
                W[a][2] = bytes_slot_window<2>(w[a], m1[a][2], &r8[a][2]);
                W[a][3] = bytes_slot_window<3>(w[a], m1[a][3], &r8[a][3]);
                W[a][4] = bytes_slot_window<0>(w[a], m2[a][0], &r8[a][4]);
                W[a][5] = bytes_slot_window<1>(w[a], m2[a][1], &r8[a][5]);
                W[a][6] = bytes_slot_window<2>(w[a], m2[a][2], &r8[a][6]);
                W[a][7] = bytes_slot_window<3>(w[a], m2[a][3], &r8[a][7]);
            }
            // (the eight entries requested together, like the ASCII lookups above: left alone hipcc reads, waits and decodes slot by slot)
            static_assert(R == 1, "the pin below names the entries of one row");
            uint2 ent[R][8];
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) ent[a][s] = lead_entry(L.ltab, W[a][s]);
            asm volatile("" : "+v"(ent[0][0].x), "+v"(ent[0][0].y), "+v"(ent[0][1].x), "+v"(ent[0][1].y), "+v"(ent[0][2].x), "+v"(ent[0][2].y),
                              "+v"(ent[0][3].x), "+v"(ent[0][3].y), "+v"(ent[0][4].x), "+v"(ent[0][4].y), "+v"(ent[0][5].x), "+v"(ent[0][5].y),
                              "+v"(ent[0][6].x), "+v"(ent[0][6].y), "+v"(ent[0][7].x), "+v"(ent[0][7].y));
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    bad[a][s] = lead_index(ent[a][s], W[a][s], &off2[a][s], &Rs[a][s]);
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) blk[a][s] = *reinterpret_cast<const uint16_t*>(L.t1b + off2[a][s]);
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) code[a][s] = L.t2b[blk[a][s] | (Rs[a][s] & 0x3Fu)];
            // A slot without a lead decoded the dword's last byte: an ASCII or continuation byte gives code 0 (its table entry),
            // a lead -- then the dword's first or second lead -- its own code once more, at its own place: OR-ing is right either way.
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) out[a][s & 3] |= code[a][s] << r8[a][s];
            // Sequences that are cut short (malformed input) are U+FFFD: looked for once per row, wave-wide, instead of a compare
            // and a select per slot; the rare row that holds one puts U+FFFD's code in place of what the slot looked up.
            bool any_bad = false;
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int s = 0; s < 8; ++s) any_bad = any_bad || bad[a][s];
            if (__any(any_bad)) {
#pragma unroll
                for (int a = 0; a < R; ++a)
#pragma unroll
                    for (int s = 0; s < 8; ++s)
                        if (bad[a][s]) out[a][s & 3] = (out[a][s & 3] & ~(0xFFu << r8[a][s])) | (code_fffd << r8[a][s]);
            }
        }
#pragma unroll
        for (int a = 0; a < R; ++a) {
            while (__any((rest[a][0] | rest[a][1] | rest[a][2] | rest[a][3]) != 0u)) {   // wave-uniform; never taken on well-formed UTF-8
                uint32_t r8[4], cp[4];
                cp[0] = utf8_cp_of(bytes_slot_window<0>(w[a], rest[a][0], &r8[0]));
                cp[1] = utf8_cp_of(bytes_slot_window<1>(w[a], rest[a][1], &r8[1]));
                cp[2] = utf8_cp_of(bytes_slot_window<2>(w[a], rest[a][2], &r8[2]));
                cp[3] = utf8_cp_of(bytes_slot_window<3>(w[a], rest[a][3], &r8[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    out[a][q] |= (rest[a][q] ? classify1_b6(L.t1b, L.t2b, cp[q]) : 0u) << r8[q];
                    rest[a][q] &= rest[a][q] - 1u;
                }
            }
            *reinterpret_cast<uint4*>(L.stage + stage_addr(1024u * (i0 + a) + 16u * lane)) = make_uint4(out[a][0], out[a][1], out[a][2], out[a][3]);
        }
    }
    LATOK_STAMP(13);   // (share of stamp 2: the four rows)
    // the 8 bytes after the tile -> halo[8..15] as staging bytes (code at a lead, LK_CODE_CONT at a continuation byte).
    // Lane k+1 owns byte k.
    {
        const int k = lane - 1;
        const bool in_win = lane >= 1 && lane < 9 && t0 + kTile + k < total;
        // the 3 bytes after mine (lanes 2..11 hold them; 0 = "no such byte", which is not a continuation byte)
        const uint32_t b1 = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)hb) & 0xFFu;
        const uint32_t b2 = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)b1) & 0xFFu;
        const uint32_t b3 = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)b2) & 0xFFu;
        const uint32_t W = (hb & 0xFFu) | (b1 << 8) | (b2 << 16) | (b3 << 24);
        const uint32_t b0 = W & 0xFFu;
        const uint32_t my_code = classify1_b6(L.t1b, L.t2b, b0 < 0x80u ? b0 : utf8_cp_of(W));
        if (in_win) L.halo[8 + k] = (uint8_t)(u8_is_cont(b0) ? LK_CODE_CONT : my_code);
    }
    return false;
}

// byte space, phase 2: what a word needs from its surroundings -- the staging bytes of the 8 bytes after it (lane 63: the
// halo), whether one of them is a continuation byte, the string starts after it, and the owner state in front of it (the
// last four staging bytes of the row before; lane 0: the halo)
__device__ __forceinline__ void bytes_word_context(const TileLds& L, int lane, lk_halo_bytes* hb, bool* next_has_cont, uint32_t* own_code,
                                                   int* own_left) {
    hb->next_codes = lane < 63 ? *reinterpret_cast<const lk_u64*>(L.stage + 80u * lane + 80u) : *reinterpret_cast<const lk_u64*>(L.halo + 8);
    hb->next_B = (uint32_t)(L.bw[lane + 1] & 0xFFFFull);
    const lk_u64 y = hb->next_codes ^ 0x8080808080808080ull;          // a byte equals LK_CODE_CONT
    *next_has_cont = (~(((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y) & 0x8080808080808080ull) != 0ull;
    lk_owner_before(lane > 0 ? *reinterpret_cast<const uint32_t*>(L.stage + 80u * lane - 20u) : 0u, own_code, own_left);
    if (lane == 0) {
        *own_code = L.halo[0];
        *own_left = L.halo[4];
    }
    hb->prev = *own_code;
}
// Phase 2 of a tile (lane = one 64-char word): everything after the code bytes, the halo codes and the string-start
// words are in the wave's LDS buffer L.  (A separate function because a producer / consumer variant of the kernel ran
// the two phases in different waves; see DESIGN.md, negative results.)
template <int MODE, bool DEFER = false, bool SMALL = false>
__device__ __forceinline__ lk_u64 tile_phase2(const SplitParams& P, const TileLds& L, int64_t t, int q_in, int tail_zero,
                                            bool write_summary, int4* summ_l, int lane, bool raw_stage, bool ascii_tile
#ifdef LATOK_STAMPS
                                            , unsigned long long* stamp_acc, unsigned long long& stamp_prev
#endif
                                            ) {
    const int64_t t0 = t * kTile;
    const int64_t total = P.total;
    // ---- phase 2: lane = one 64-char word ---------------------------------------------------------------------
    const lk_u64 B = L.bw[lane];
    const int64_t base = t0 + 64 * (int64_t)lane;
    lk_local loc;
    lk_rule_counts counts;    // kModeValuesRules only: rows of C_SPLIT / C_SYM that hold at each char
    lk_u64 space_plane = 0;   // SPACE plane for the token-span passes (byte mode: smeared over continuation bytes)
    lk_u64 cont_plane = 0;    // byte mode: continuation bytes of my word (P.lead_out)
    int no_patch = 0;         // byte mode: the tile holds multi-byte chars: the one bit a patch of the resolve stage keeps (the C_SYM
                              // bit of a block's last char) sits at that char's LEAD byte, which the stage finds in the bytes
    if (MODE == kModeBlockMask) {
        // a1 -> start plane, a2 -> space plane; 64 bytes each, non-zero = set (PyArray_Nonzero, latok.c:178,198)
        lk_u64 st = 0, sp = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t p = base + 4 * k;
            uint32_t w1 = 0, w2 = 0;
            if (p + 4 <= total) {
                w1 = *reinterpret_cast<const uint32_t*>(P.bm_a1 + p);
                w2 = *reinterpret_cast<const uint32_t*>(P.bm_a2 + p);
            } else {
                for (int b = 0; b < 4; ++b)
                    if (p + b < total) {
                        w1 |= (uint32_t)(uint8_t)P.bm_a1[p + b] << (8 * b);
                        w2 |= (uint32_t)(uint8_t)P.bm_a2[p + b] << (8 * b);
                    }
            }
            // byte != 0 -> one bit per byte -> 4-bit nibble
            const uint32_t n1 = ((((w1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w1) & 0x80808080u) >> 7;
            const uint32_t n2 = ((((w2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w2) & 0x80808080u) >> 7;
            st |= (lk_u64)(((n1 * 0x00204081u) >> 21) & 0xFu) << (4 * k);
            sp |= (lk_u64)(((n2 * 0x00204081u) >> 21) & 0xFu) << (4 * k);
        }
        loc.start = st;
        loc.S = sp;
        loc.raw = ~0ull;
        loc.sym = 0;
        loc.t_space = loc.t_sym = loc.t_prevsym = loc.t_camel_next = loc.t_camel_prev = 0;
    } else {
        uint32_t d[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 q = *reinterpret_cast<const uint4*>(L.stage + 80u * lane + 16u * k);   // == stage_addr(64 lane + 16 k)
            d[4 * k + 0] = q.x; d[4 * k + 1] = q.y; d[4 * k + 2] = q.z; d[4 * k + 3] = q.w;
        }
        lk_halo h;
        // neighbours of the word in the padded layout: stage_addr(64 lane - 1) = 80 lane - 17, (64 lane + 64) = 80 lane + 80
        h.prev = lane > 0 ? L.stage[80u * lane - 17u] : L.halo[0];
        h.next0 = lane < 63 ? L.stage[80u * lane + 80u] : L.halo[1];
        h.next1 = lane < 63 ? L.stage[80u * lane + 81u] : L.halo[2];
        const lk_u64 Bn = L.bw[lane + 1] & 3ull;
        lk_u64 plane[8];
        if (mode_base(MODE) == kModeLatin1 || (mode_base(MODE) == kModeBytes && raw_stage)) {
            // d = raw bytes: classify + slice through the LUT; the neighbour bytes become codes through the code table
            LATOK_STAMP(9);    // (share of stamp 4: the four ds_read_b128 + neighbour bytes)
            if (ascii_tile) {
                // all 4096 bytes are ASCII: bit-slice the raw bytes and derive the code planes as boolean functions of the
                // raw planes -- no table, nothing through the LDS pipe (128 ds_read_b32 per word otherwise, which hit a bank
                // twice in ~90 % of the passes and kept the pipe busy for about half of a tile round)
                lk_u64 rawp[8];
                lk_bitslice64(d, rawp);
                lk_ascii_code_planes<mode_rules(MODE)>(rawp, plane);
            } else {
                slice_lut64(d, L.lut, plane);   // (Latin-1 tiles with chars >= 0x80; a raw byte-space tile is always ASCII)
            }
            LATOK_STAMP(10);   // (share of stamp 4: LUT slicing)
            h.prev = lane > 0 ? L.ctab[h.prev] : h.prev;
            h.next0 = lane < 63 ? L.ctab[h.next0] : h.next0;
            h.next1 = lane < 63 ? L.ctab[h.next1] : h.next1;
        } else {
            lk_bitslice64(d, plane);
        }
        if (mode_is_units(MODE)) {
            // one code per char like a UTF-32 tile, in the byte-space layout: the two chars after my row are the next row's
            // first codes (lane 63: halo[8], halo[9])
            lk_halo ha;
            ha.prev = h.prev;
            ha.next0 = lane < 63 ? h.next0 : L.halo[8];
            ha.next1 = lane < 63 ? h.next1 : L.halo[9];
            if (mode_rules(MODE)) loc = lk_rules_generic(plane, ha, B, Bn, P.rules);   // (the staging bytes are rule codes then)
            else loc = lk_rules(lk_decode(plane), ha, B, Bn);
            space_plane = loc.S;
        } else if (mode_base(MODE) == kModeBytes && raw_stage) {
            // all-ASCII tile: byte positions are char positions, the plain rules apply (codes of the neighbours: above)
            lk_halo ha;
            ha.prev = h.prev;
            ha.next0 = lane < 63 ? h.next0 : L.halo[8];
            ha.next1 = lane < 63 ? h.next1 : L.halo[9];
            if (mode_rules(MODE)) loc = lk_rules_generic(plane, ha, B, Bn, P.rules);
            else loc = lk_rules(lk_decode(plane), ha, B, Bn);
            space_plane = loc.S;
        } else if (mode_base(MODE) == kModeBytes) {
            // byte space: the continuation bytes carry LK_CODE_CONT -> continuation plane of my word
            const lk_u64 C = lk_take_cont_plane(plane);
            cont_plane = C;
            lk_halo_bytes hb;
            bool next_has_cont;
            uint32_t own_code;
            int own_left;
            bytes_word_context(L, lane, &hb, &next_has_cont, &own_code, &own_left);
            no_patch = __ballot(C != 0ull || next_has_cont) != 0ull;
            if (!no_patch) {
                // no multi-byte char in or right after the tile: positions are chars, the plain rules apply
                lk_halo ha;
                ha.prev = h.prev;
                ha.next0 = (uint32_t)(hb.next_codes & 0xFFull);
                ha.next1 = (uint32_t)((hb.next_codes >> 8) & 0xFFull);
                if (mode_rules(MODE)) loc = lk_rules_generic(plane, ha, B, Bn, P.rules);
                else loc = lk_rules(lk_decode(plane), ha, B, Bn);
                space_plane = loc.S;
            } else if (mode_rules(MODE)) {
                // run-time rule tables in byte space: every NEXT_* / AFTER_NEXT_* column through the next-lead operator
                lk_smear_planes<0x37u>(plane, C, own_code, own_left);
                hb.prev = own_code;
                loc = lk_rules_generic_bytes(plane, C, hb, B, P.rules, &space_plane);
            } else if (__ballot(lk_rules_bytes_weird(plane, C, hb.next_codes)) == 0ull) {
                // codes sit at lead bytes only: give the continuation bytes their owner's code in the planes the PREV_*
                // columns and the token stripping read (SPACE, SYMBOL, LOWER, ALPHA_NUM, ALPHA = bits 0, 1, 2, 4, 5)
                lk_smear_planes<0x37u>(plane, C, own_code, own_left);
                hb.prev = own_code;
                loc = lk_rules_bytes_fast(plane, C, hb, B, &space_plane);
            } else {
                // a continuation byte right after '#' '$' '^' '@' ':' '/' '.' somewhere in the tile (malformed UTF-8): the
                // general form.  (Inlined: round 3 had it as a __noinline__ call for the sake of the hot loop's registers, which
                // cost a 240-byte scratch frame; without the write-combining buffer the kernel holds both forms in 168 VGPRs, 0 B scratch.)
                lk_smear_planes<0x37u>(plane, C, own_code, own_left);
                hb.prev = own_code;
                loc = lk_rules_bytes_general(plane, C, hb, B, &space_plane);
            }
        } else if (mode_rules(MODE)) {
            loc = lk_rules_generic(plane, h, B, Bn, P.rules, MODE == kModeValuesRules ? &counts : nullptr);
        } else {
            const lk_feat f = lk_decode(plane);
            loc = lk_rules(f, h, B, Bn);
        }
    }
    LATOK_STAMP(4);
    lk_fwd fw = lk_forward(loc.start, loc.S, B);

    // forward: inclusive (max,+) scan of the per-word queue transfer functions over the 64 lanes
    const lk_qfn inc = qfn_wave_scan(lk_qfn_of(fw));
    lk_qfn exc;   // exclusive: the function of lanes 0..lane-1 (identity for lane 0)
    exc.a = dpp_mov<kDppWaveShr1, 0xF>(0, inc.a);
    exc.b = dpp_mov<kDppWaveShr1, 0xF>(0, inc.b);
    const int r = lk_qfn_apply(exc, q_in);
    lk_qfn tile_fn;
    tile_fn.a = lane_read(inc.a, 63);
    tile_fn.b = lane_read(inc.b, 63);
    if (r > 0) lk_apply_extra(fw, r);
    LATOK_STAMP(5);

    if (write_summary) {
        const lk_u64 cl = loc.S | B;                       // closing events of my word
        const lk_u64 closing_lanes = __ballot(cl != 0);
        const int first_lane = closing_lanes ? lk_ctz(closing_lanes) : 64;
        const int contrib = lane < first_lane ? lk_popc(loc.start) : (lane == first_lane ? fw.head_starts : 0);
        const int head = __ballot(contrib != 0) ? wave_sum(contrib) : 0;   // starts are rare: usually no sum needed
        // geometry of the two blocks that straddle the tile edges, so that the scan stage can patch the common cases
        // in place instead of recomputing the tile:
        //   c_rel : first closing event (4096 if none)       -> head block = [0, c_rel)
        //   p_rel : first char of the open tail block          -> tail block = [p_rel, 4096)
        //   head_sym / tail_sym : the one position of each block that can carry a C_SYM bit (its last char)
        //   tail_keep : the tail block begins with a string start (its bit stays 1)
        int c_rel = kTile, p_rel = 0, tail_keep = 0;
        if (closing_lanes) {
            const int last_lane = 63 - __builtin_clzll(closing_lanes);
            const int my_first = cl ? 64 * lane + lk_ctz(cl) : 0;
            const int top = cl ? 63 - __builtin_clzll(cl) : 0;
            const int s_top = (int)((loc.S >> top) & 1ull);
            c_rel = lane_read(my_first, first_lane);
            p_rel = lane_read(64 * lane + top + s_top, last_lane);
            tail_keep = lane_read(1 - s_top, last_lane);
        }
        int head_sym, tail_sym;
        if (mode_base(MODE) == kModeBytes && no_patch) {   // (wave-uniform; tiles without multi-byte chars take the char form below)
            // A block holds no closing event, and C_SYM = SYMBOL & NEXT_SPACE is set only in front of one: the only C_SYM bit a block
            // can hold is its last char's, wherever that char's lead byte is -- "any C_SYM bit in the block" is the flag.
            const int64_t lo_w = 64 * (int64_t)lane;
            const lk_u64 in_head = c_rel >= lo_w + 64 ? ~0ull : (c_rel <= lo_w ? 0ull : ((1ull << (c_rel - lo_w)) - 1ull));
            const lk_u64 in_tail = p_rel <= lo_w ? ~0ull : (p_rel >= lo_w + 64 ? 0ull : (~0ull << (p_rel - lo_w)));
            head_sym = __ballot((loc.sym & in_head) != 0ull) != 0ull;
            tail_sym = __ballot((loc.sym & in_tail) != 0ull) != 0ull;
        } else {
            const int hs_pos = c_rel > 0 ? c_rel - 1 : 0;
            head_sym = c_rel > 0 ? lane_read((int)((loc.sym >> (hs_pos & 63)) & 1ull), hs_pos >> 6) : 0;
            tail_sym = lane_read((int)(loc.sym >> 63), 63);
        }
        if (lane == 0) {
            const int geom = (closing_lanes != 0) | (c_rel << 1) | (p_rel << 14) | (head_sym << 27) | (tail_keep << 28) |
                             (tail_sym << 29) | (no_patch << 30);
            *summ_l = make_int4(tile_fn.a, tile_fn.b, head, geom);   // LDS; the segment publishes them in one burst
        }
    }

    // backward: zeroing closings clear the block below them; the carry chain over lanes is one 64-bit add on ballots
    LATOK_STAMP(6);
    const lk_u64 zall = fw.zs | fw.zb;
    const int z0_next = dpp_mov<kDppWaveShl1, 0xF>(0, (int)(zall & 1ull));   // lane 63 gets 0
    const lk_bwd bw = lk_backward_prepare(zall, loc.S, B, z0_next);
    const int tz = tail_zero >= 0 ? tail_zero : (lk_qfn_apply(tile_fn, q_in) > 0);
    // chain order is lane 63 -> 0, so reverse the ballots: bit i' = lane 63 - i'
    const lk_u64 G = lk_rev(__ballot(bw.g)), Pm = lk_rev(__ballot(bw.p));
    const lk_u64 X = G | Pm, Y = G;
    const lk_u64 carries_in = (X + Y + (lk_u64)tz) ^ X ^ Y;
    const int cin = (int)((carries_in >> (63 - lane)) & 1ull);
    const lk_u64 cleared = lk_backward_fill(bw, cin, loc.S);

    LATOK_STAMP(7);
    lk_u64 out_word = 0;
    if (base < total) {
        const int64_t remain = total - base;
        const lk_u64 valid = remain >= 64 ? ~0ull : ((1ull << remain) - 1ull);
        const lk_u64 keep = ~cleared;
        if (mode_writes_bits(MODE)) {
            out_word = ((loc.raw & keep) | loc.sym | B) & valid;
            uint64_t* const bits_out = SMALL ? L.small_bits : P.bits_out;
            uint64_t* const space_out = SMALL ? L.small_space : P.space_out;
            if (!DEFER) bits_out[base >> 6] = out_word;
            if (space_out) space_out[base >> 6] = (mode_is_bytes(MODE) ? space_plane : loc.S) & valid;   // token-span mode only
        } else {
            // kModeValues: split VALUES 0..5 = (sum of the five C_SPLIT terms) * mask + C_SYM term; string start = 1
            // kModeBlockMask: the 1/0 block mask itself; element 0 follows the reference's quirk (never zeroed on the
            //   general path because "previous space" starts at 0, latok.c:224; zero only when there is no space at all)
            uint8_t* dst = P.values_out + base;
            const int n = remain >= 64 ? 64 : (int)remain;
#pragma unroll 1
            for (int w = 0; w < 16; ++w) {
                uint32_t packed = 0;
                if (MODE == kModeBlockMask) {
                    packed = ((uint32_t)((keep >> (4 * w)) & 0xFull) * 0x00204081u) & 0x01010101u;
                    if (base == 0 && w == 0) {
                        const int first = (P.bm_flags[0] != 0 && P.bm_flags[1] == 0) ? 0 : 1;
                        packed = (packed & ~0xFFu) | (uint32_t)first;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int i = 4 * w + b;
                        int v, vy;
                        if (MODE == kModeValuesRules) {
                            // what the reference returns for ANY tables: (number of C_SPLIT rows that hold) * mask + (number
                            // of C_SYM rows that hold), default_tokenizer.py:121-132 over latok.c:329-338
                            v = vy = 0;
#pragma unroll
                            for (int c = 0; c < LK_COUNT_BITS; ++c) {
                                v |= (int)((counts.split[c] >> i) & 1) << c;
                                vy |= (int)((counts.sym[c] >> i) & 1) << c;
                            }
                        } else {
                            v = (int)((loc.t_space >> i) & 1) + (int)((loc.t_sym >> i) & 1) + (int)((loc.t_prevsym >> i) & 1) +
                                (int)((loc.t_camel_next >> i) & 1) + (int)((loc.t_camel_prev >> i) & 1);
                            vy = (int)((loc.sym >> i) & 1);
                        }
                        v = ((keep >> i) & 1) ? v : 0;
                        v += vy;
                        if ((B >> i) & 1) v = 1;
                        packed |= (uint32_t)v << (8 * b);
                    }
                }
                if (4 * w + 4 <= n) {
                    *reinterpret_cast<uint32_t*>(dst + 4 * w) = packed;
                } else {
                    for (int b = 0; b < 4 && 4 * w + b < n; ++b) dst[4 * w + b] = (uint8_t)(packed >> (8 * b));
                }
            }
        }
    }
    if (mode_base(MODE) == kModeBytes && !SMALL && P.lead_out) {
        // code-point results (k_lead_compress): the lead bytes of my word, how many leads the tile has before it, leads per tile
        const int64_t remain = total - base;
        const lk_u64 valid = remain >= 64 ? ~0ull : (remain <= 0 ? 0ull : ((1ull << remain) - 1ull));
        const lk_u64 leadw = ~cont_plane & valid;
        const int cnt = lk_popc(leadw);
        int inc = cnt;
        inc += dpp_mov<kDppRowShr1, 0xF>(0, inc);
        inc += dpp_mov<kDppRowShr2, 0xF>(0, inc);
        inc += dpp_mov<kDppRowShr4, 0xF>(0, inc);
        inc += dpp_mov<kDppRowShr8, 0xF>(0, inc);
        inc += dpp_mov<kDppRowBcast15, 0xA>(0, inc);
        inc += dpp_mov<kDppRowBcast31, 0xC>(0, inc);
        if (base < total) {
            P.lead_out[base >> 6] = leadw;
            P.lead_pref_out[base >> 6] = (uint16_t)(inc - cnt);
        }
        if (lane == 63) P.lead_cnt_out[t] = inc;
    }
    LATOK_STAMP(8);
    wave_lds_sync();  // staging buffer is reused by this wave's next tile
    return out_word;
}

// UTF-32 bitmask mode: the first R rows (1 KiB each) of the wave's NEXT tile are requested into registers right
// before phase 2 of the current one, so that the wave has loads in flight while it computes.  The wait counts are explicit
// (a vmcnt(0) on every path in front of the requests; the pass that inserts waits otherwise drains the queue at the first
// register it loses track of, and the prefetch silently does nothing).  Measured on C2, same box, twice: kernel 96.1-96.3 ->
// 94.4-94.5 us with R = 2 (R = 1: 93.9-94.1, R = 4: 95-99), step 106.6-107.3 -> 105.0-105.1; C3 / C4 / C5 within their noise
// (profiles/r03_ab_headline_prefetch.txt).

template <int MODE, bool DEFER = false, bool SMALL = false, bool FAST_TAIL = false, int PF = kCpsPrefetchRows>
__device__ __forceinline__ lk_u64 process_tile(const SplitParams& P, const TileLds& L, int64_t t, int64_t idx0, int q_in,
                                             int tail_zero, bool write_summary, int4* summ_l, int lane
#ifdef LATOK_STAMPS
                                             , unsigned long long* stamp_acc = nullptr
#endif
                                             , CpsPrefetch* pf = nullptr, int64_t t_next = -1
                                             ) {
#ifdef LATOK_STAMPS
    unsigned long long stamp_prev = 0, stamp_dummy[16];
    if (!stamp_acc) stamp_acc = stamp_dummy;
#endif
    const int64_t t0 = t * kTile;
    const int64_t total = P.total;
    const uint32_t st_lane = 4u * lane + 16u * ((uint32_t)lane >> 4);   // stage_addr(4 lane); row i adds 320 i
    bool raw_stage = mode_base(MODE) == kModeLatin1;   // the staging buffer holds raw bytes, not codes (Latin-1; all-ASCII tiles of byte mode)
    bool ascii_tile = false;                // ... and every one of them is ASCII (wave-uniform)
    bool tile_not_ascii = false;            // UTF-32: some row of the tile took the two-stage lookup (wave-uniform)
    LATOK_STAMP(0);

    // small loads first, so that their latency flies together with the 16 KiB of code points: the start offsets of
    // the next 64 strings and the three halo characters
    int64_t ro = idx0 + lane <= P.n_str ? P.row_off[idx0 + lane] : INT64_MAX;
    uint32_t halo_cp = 0xFFFFFFFFu;   // out of range -> class 0
    if (MODE != kModeBlockMask && !mode_is_bytes(MODE) && lane < 3) {
        const int64_t hp = lane == 0 ? t0 - 1 : t0 + kTile + (lane - 1);
        if (hp >= 0 && hp < total) halo_cp = P.cps[hp];
    }

    // ---- phase 1: classify 4096 chars, 4 per lane per step, into the staging buffer --------------------------
    if (MODE == kModeBlockMask) {
        // planes come straight from the caller's byte arrays (compat _gen_block_mask): nothing to classify
    } else if (mode_base(MODE) == kModeLatin1) {
        ascii_tile = units_phase1<1>(P, L, t0, lane);
    } else if (mode_base(MODE) == kModeUcs2) {
        units_phase1<2>(P, L, t0, lane);
    } else if (mode_base(MODE) == kModeBytes) {
        raw_stage = bytes_phase1(P, L, t0, lane
#ifdef LATOK_STAMPS
                                 , stamp_acc, stamp_prev
#endif
                                 );
        ascii_tile = raw_stage;
    } else if (FAST_TAIL) {
        // Small batches: the batch's last, partial tile takes the same road as a full one -- only the rows of 256 chars that
        // exist are requested, all of them before the first table lookup; chars that do not exist read as 0 and their
        // codes are masked to 0 ("nothing") -- because its latency is a visible share of the call: a one-tile batch with
        // its chars in host memory (k_small_batch) spent most of its 13 us in the 16 serial load -> lookup rounds of the
        // general form below, a 4-tile batch on the pinned path 15 of its 31 us.  (Not for large batches: the shared
        // loop costs the full-tile path 16 VGPRs and 4 % on C2.)
        const bool full = t0 + kTile <= total;
        u32x4 v[16];
        const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t0) + lane;
        const int64_t remain0 = total - t0 - 4 * (int64_t)lane;       // chars that exist from my first char on (row 0)
        int n_rows = 16;
        if (full) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
        } else {
            // One 16-byte load per existing row and lane, nothing divergent around it and no zero-fill before it (a divergent
            // x4 / scalar-tail choice, or a register write the compiler cannot order against loads in flight, makes it wait
            // for outstanding loads between the rows: they would arrive one by one again).  A lane beyond the end re-reads
            // the 16-byte block that holds the last char; a lane whose 4 chars straddle the end reads up to 12 bytes past
            // the last char inside that block (cps is 16-byte aligned: the same page).  The codes of chars that do not
            // exist are masked below.
            const int64_t last_blk = (total - 1) & ~(int64_t)3;
            n_rows = (int)((total - t0 + 255) >> 8);                  // wave-uniform, 1..16
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < n_rows) {
                    int64_t p = t0 + 256 * i + 4 * (int64_t)lane;
                    p = p < last_blk ? p : last_blk;
                    v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(P.cps + p));
                }
            }
        }
        uint32_t* codes = P.codes_out ? reinterpret_cast<uint32_t*>(P.codes_out + t0) + lane : nullptr;   // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t c = 0;
            if (i < n_rows) {
                c = classify4(L.t1, L.t2, v[i]);
                if (!full) {
                    const int64_t remain = remain0 - 256 * i;
                    c &= remain >= 4 ? 0xFFFFFFFFu : (remain <= 0 ? 0u : ((1u << (8 * (int)remain)) - 1u));
                }
            }
            *reinterpret_cast<uint32_t*>(L.stage + st_lane + 320u * i) = c;   // == stage_addr(256 i + 4 lane)
            if (codes) codes[64 * i] = c;
        }
    } else if (t0 + kTile <= total) {
        u32x4 v[16];
        const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t0) + lane;
        constexpr int R = PF;
        const bool pre = R > 0 && pf && pf->valid;      // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < R && pre) v[i] = pf->v[i < R ? i : 0];
            else v[i] = __builtin_nontemporal_load(src + 64 * i);
        }
        LATOK_STAMP(1);
        uint32_t* codes = P.codes_out ? reinterpret_cast<uint32_t*>(P.codes_out + t0) + lane : nullptr;   // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = classify4(L.t1, L.t2, v[i], &tile_not_ascii);
            *reinterpret_cast<uint32_t*>(L.stage + st_lane + 320u * i) = c;   // == stage_addr(256 i + 4 lane)
            if (codes) codes[64 * i] = c;                                     // 256 contiguous bytes per wave instruction
        }
    } else {
        // the batch's last, partial tile of a large batch (its code bytes are written up to the end of the tile: 0 behind
        // the last char)
        uint32_t* codes = P.codes_out ? reinterpret_cast<uint32_t*>(P.codes_out + t0) + lane : nullptr;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int64_t p = t0 + 256 * i + 4 * lane;
            u32x4 v;
            v.x = p + 0 < total ? P.cps[p + 0] : 0xFFFFFFFFu;   // out of range -> class 0 ("nothing")
            v.y = p + 1 < total ? P.cps[p + 1] : 0xFFFFFFFFu;
            v.z = p + 2 < total ? P.cps[p + 2] : 0xFFFFFFFFu;
            v.w = p + 3 < total ? P.cps[p + 3] : 0xFFFFFFFFu;
            const uint32_t c = classify4(L.t1, L.t2, v);
            *reinterpret_cast<uint32_t*>(L.stage + st_lane + 320u * i) = c;   // == stage_addr(256 i + 4 lane)
            if (codes) codes[64 * i] = c;
        }
    }
    // halo chars t0-1, t0+4096, t0+4097 (lanes 0..2) and the string-start words
    if (MODE != kModeBlockMask && !mode_is_bytes(MODE) && lane < 3) L.halo[lane] = (uint8_t)classify1(L.t1, L.t2, halo_cp);
    L.bw[lane] = 0;
    if (lane == 0) L.bw[64] = 0;
    LATOK_STAMP(2);
    wave_lds_sync();
    for (;;) {
        const int64_t rel = ro - t0;
        if (rel >= 0 && rel < kTile + 64) atomicOr(&L.bw[rel >> 6], 1ull << (rel & 63));
        const int64_t last = lane_read64(ro, 63);
        if (last >= t0 + kTile + 64) break;
        idx0 += 64;
        ro = idx0 + lane <= P.n_str ? P.row_off[idx0 + lane] : INT64_MAX;
    }
    wave_lds_sync();
    LATOK_STAMP(3);
    if (MODE == kModeBits && pf) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every load of this tile has been consumed -- said on every path
        pf->valid = false;
        const bool want = true;
        if (want && t_next >= 0 && (t_next + 1) * kTile <= total) {
            const u32x4* nsrc = reinterpret_cast<const u32x4*>(P.cps + t_next * kTile) + lane;
#pragma unroll
            for (int i = 0; i < PF; ++i) pf->v[i] = __builtin_nontemporal_load(nsrc + 64 * i);
            pf->valid = true;
        }
    }

    return tile_phase2<MODE, DEFER, SMALL>(P, L, t, q_in, tail_zero, write_summary, summ_l, lane, raw_stage, ascii_tile
#ifdef LATOK_STAMPS
                                    , stamp_acc, stamp_prev
#endif
                                    );
}

// ---------------------------------------------------------------------------------------------------------------
// stage 2: resolve, for every tile, (a) the number of pending starts entering it (forward scan of the tile transfer
// functions) and (b) whether the block that is open at its end gets zeroed (needs the starts that follow before the
// next closing event: backward scan).  Tiles whose provisional assumptions (q_in == 0, tail = "pending at end") do
// not hold are appended to the fix list.  One workgroup; each thread owns a contiguous chunk of tiles.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ Fn64 fn_then(Fn64 f1, Fn64 f2) {
    Fn64 f;
    const bool c1 = f1.a <= kNegInf64, c2 = f2.a <= kNegInf64;
    if (c2) { f.a = kNegInf64; f.b = f2.b; return f; }
    f.a = c1 ? kNegInf64 : f1.a + f2.a;
    const long long c = f1.b + f2.a;
    f.b = c > f2.b ? c : f2.b;
    return f;
}
__device__ __forceinline__ long long fn_apply(Fn64 f, long long q) {
    if (f.a <= kNegInf64) return f.b;
    const long long c = q + f.a;
    return c > f.b ? c : f.b;
}
__device__ __forceinline__ Fn64 fn_of(int4 s) {
    Fn64 f;
    f.a = s.x <= LK_NEG_INF / 2 ? kNegInf64 : (long long)s.x;
    f.b = s.y;
    return f;
}
// backward element: (has_closing, head_starts); (c1,h1) followed by (c2,h2) = (c1|c2, c1 ? h1 : h1+h2)
__device__ __forceinline__ Hd64 hd_then(Hd64 x, Hd64 y) {
    Hd64 r;
    r.c = x.c | y.c;
    r.h = x.c ? x.h : x.h + y.h;
    return r;
}

constexpr int kScanWaves = kWPB;

__device__ __forceinline__ Fn64 fn_identity() { Fn64 f; f.a = 0; f.b = 0; return f; }   // identity on q >= 0
__device__ __forceinline__ Hd64 hd_identity() { Hd64 h; h.h = 0; h.c = 0; return h; }

// Ordered block-wide scans over kScanThreads elements (one per thread).  Returns, for this thread, the composition of
// all EARLIER elements (fn: exclusive prefix) and of all LATER elements (hd: exclusive suffix); *tot_* get the
// composition of the whole block.  Wave-level shuffles + 16 wave aggregates in LDS.
template <int NW>
struct ScanLdsT {
    Fn64 fn_w[NW];
    Hd64 hd_w[NW];
};
typedef ScanLdsT<kScanWaves> ScanLds;
template <int NW = kScanWaves>
__device__ __forceinline__ void block_scan(Fn64 f, Hd64 h, ScanLdsT<NW>& L, Fn64* excl_fn, Hd64* excl_hd, Fn64* tot_fn,
                                           Hd64* tot_hd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Fn64 fi = f;
    Hd64 hi = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Fn64 o; o.a = __shfl_up(fi.a, d); o.b = __shfl_up(fi.b, d);
        if (lane >= d) fi = fn_then(o, fi);
        Hd64 oh; oh.h = __shfl_down(hi.h, d); oh.c = __shfl_down(hi.c, d);
        if (lane + d < 64) hi = hd_then(hi, oh);
    }
    __syncthreads();  // protects L against the previous use
    if (lane == 63) L.fn_w[wave] = fi;
    if (lane == 0) L.hd_w[wave] = hi;
    __syncthreads();
    Fn64 ef; ef.a = __shfl_up(fi.a, 1); ef.b = __shfl_up(fi.b, 1);
    if (lane == 0) ef = fn_identity();
    Hd64 eh; eh.h = __shfl_down(hi.h, 1); eh.c = __shfl_down(hi.c, 1);
    if (lane == 63) eh = hd_identity();
    // second level: every wave scans the 16 wave aggregates with shuffles (lanes 0..15), then picks its own entry
    Fn64 wf = lane < NW ? L.fn_w[lane] : fn_identity();
    Hd64 wh = lane < NW ? L.hd_w[lane] : hd_identity();
#pragma unroll
    for (int d = 1; d < NW; d <<= 1) {
        Fn64 o; o.a = __shfl_up(wf.a, d); o.b = __shfl_up(wf.b, d);
        if (lane >= d) wf = fn_then(o, wf);
        Hd64 oh; oh.h = __shfl_down(wh.h, d); oh.c = __shfl_down(wh.c, d);
        if (lane + d < NW) wh = hd_then(wh, oh);
    }
    // inclusive prefix of waves 0..lane in wf, inclusive suffix of waves lane..15 in wh
    Fn64 before; before.a = __shfl(wf.a, wave > 0 ? wave - 1 : 0); before.b = __shfl(wf.b, wave > 0 ? wave - 1 : 0);
    if (wave == 0) before = fn_identity();
    Hd64 after; after.h = __shfl(wh.h, wave < NW - 1 ? wave + 1 : 0); after.c = __shfl(wh.c, wave < NW - 1 ? wave + 1 : 0);
    if (wave == NW - 1) after = hd_identity();
    Fn64 all_f; all_f.a = __shfl(wf.a, NW - 1); all_f.b = __shfl(wf.b, NW - 1);
    Hd64 all_h; all_h.h = __shfl(wh.h, 0); all_h.c = __shfl(wh.c, 0);
    *excl_fn = fn_then(before, ef);
    *excl_hd = hd_then(eh, after);
    *tot_fn = all_f;
    *tot_hd = all_h;
}


// clear mask bits [lo, hi) (clamped to limit); afterwards re-set the first / last bit of the range on request
// (last_back: the kept last bit sits that many positions before hi - 1 -- byte space: the lead byte of the block's last char)
__device__ __forceinline__ void clear_range(uint64_t* bits, int64_t lo, int64_t hi, int64_t limit, int keep_first,
                                            int keep_last, int last_back = 0) {
    if (hi > limit) hi = limit;
    if (lo >= hi) return;
    const int64_t last = hi - 1 - last_back;
    for (int64_t w = lo >> 6; w <= (hi - 1) >> 6; ++w) {
        const int64_t base = w << 6;
        uint64_t m = ~0ull;
        if (lo > base) m &= ~0ull << (lo - base);
        if (hi < base + 64) m &= (1ull << (hi - base)) - 1ull;
        uint64_t v = bits[w] & ~m;
        if (keep_first && (lo >> 6) == w) v |= 1ull << (lo & 63);
        if (keep_last && last >= lo && (last >> 6) == w) v |= 1ull << (last & 63);
        bits[w] = v;
    }
}

// lower_bound over the row offsets: smallest s in [0, n_entries] with row_off[s] >= c (n_entries if none).  64-ary
// search by one wave: every lane probes one pivot, the ballot picks the sub-range (the resolve stage's rare recomputations).
__device__ __forceinline__ int64_t wave_lower_bound(const int64_t* __restrict__ row_off, int64_t n_entries, int64_t c,
                                                    int lane) {
    int64_t lo = 0, hi = n_entries;
    while (hi > lo) {
        const int64_t len = hi - lo;
        const int64_t step = (len + 63) / 64;
        const int64_t p = lo + (int64_t)lane * step;
        const bool pred = p < hi && row_off[p] >= c;
        const lk_u64 m = __ballot(pred);
        if (!m) {
            const int64_t n_valid = (len + step - 1) / step;
            lo = min(lo + (n_valid - 1) * step + 1, hi);
        } else {
            const int f = lk_ctz(m);
            hi = lo + (int64_t)f * step;
            if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
        }
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------------------------
// LDS map of both kernels (one workgroup of kWPB waves per CU)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLdsWaves = kTablesLdsBytes;                         // 16 x kWaveLdsBytes
constexpr int kLdsTf = kLdsWaves + kWPB * kWaveLdsBytes;           // int32[kSegMax]: k_resolve_fix's list of tiles to recompute
constexpr int kLdsSumm = kLdsTf + kSegMax * 4;                     // int4[kSegMax]: tile summaries of the segment
constexpr int kLdsScan = kLdsSumm + kSegMax * 16;                  // ScanLds
constexpr int kLdsMisc = kLdsScan + 512;                           // 64 ints of scratch
constexpr int kLdsSlice = kLdsMisc + 256;                          // end of the map (Latin-1 has its [slice LUT | code table] at 0)
// Byte space's tables are larger than the other modes' ([stage 1, uint16 | stage 2 | byte decode table] at 0, all of it below
// 64 KiB, so that a lookup's table base fits the 16-bit offset field of its ds_read); everything behind the tables moves up by
// the difference.  The kernels add lds_shift(MODE) to every offset above but the tables'.
constexpr int kLdsLeadTab = kB6TablesBytes;                       // kModeBytes: the byte decode table, behind the class table,
constexpr int kLdsByteCodes = kLdsLeadTab + kLeadTabBytes;        //   then code[256] of a byte taken as a char of its own: the ASCII codes, 0 from 0x80 on
constexpr int kByteCodesBytes = 256;
constexpr int lds_shift(int mode) { return mode_base(mode) == kModeBytes ? kLdsByteCodes + kByteCodesBytes - kTablesLdsBytes : 0; }
static_assert(lds_shift(kModeBytes) % 16 == 0 && kLdsByteCodes + kByteCodesBytes < 65536, "alignment / immediate offsets");
constexpr int kLdsTotalBase = kLdsSlice;
constexpr int kLdsTotalBytes = kLdsSlice + lds_shift(kModeBytes);
constexpr int lds_total(int mode) { return mode_base(mode) == kModeBytes ? kLdsTotalBytes : kLdsTotalBase; }
// Waves per workgroup of the TILE kernel.  The Latin-1 kernel needs <= 128 VGPRs and its LDS map has room, so it runs 16 waves
// per CU (4 per SIMD): its waves spend half their life in s_waitcnt, a fourth wave per SIMD fills part of that.  The
// buffers of waves 12..15 sit behind the rest of the map, so that every other offset is the same for all kernels.
constexpr int kNarrowWPB = 16;   // Latin-1 / UCS-2 tile kernels: 4 waves per SIMD (<= 128 VGPRs)
constexpr int tile_wpb(int mode) {
    return mode_base(mode) == kModeLatin1 && !mode_rules(mode) ? kNarrowWPB
         : (mode_base(mode) == kModeUcs2 && !mode_rules(mode) ? kNarrowWPB : kWPB);   // (the rule interpreter needs > 128 VGPRs)
}
constexpr int lds_total_tiles(int mode) { return lds_total(mode) + (tile_wpb(mode) > kWPB ? (tile_wpb(mode) - kWPB) * kWaveLdsBytes : 0); }
static_assert(lds_total_tiles(kModeLatin1) <= 160 * 1024 && lds_total_tiles(kModeUcs2) <= 160 * 1024, "LDS budget of one CU");
constexpr int kLdsTotal = kLdsTotalBytes;
static_assert(sizeof(ScanLds) <= 512, "scan scratch");
static_assert(kSegMax == kWPB * 64, "one tile per thread in the block-wide scans");
static_assert(kLdsTotal <= 160 * 1024, "LDS budget of one CU");
static_assert(kLdsSumm % 16 == 0 && kLdsScan % 16 == 0 && kLdsSlice % 16 == 0, "alignment");

// byte space: code[256] of a byte taken as a char of its own -- the 128 ASCII codes (stage-2 blocks 0 and 1), 0 from 0x80 on: the
// lookups of phase 1 then need no masking of the non-ASCII positions (threads 0..15)
__device__ __forceinline__ void load_byte_codes(uint8_t* lds, const SplitParams& P) {
    if (threadIdx.x < 8) reinterpret_cast<uint4*>(lds + kLdsByteCodes)[threadIdx.x] = reinterpret_cast<const uint4*>(P.t2)[threadIdx.x];
    else if (threadIdx.x < 16) reinterpret_cast<uint4*>(lds + kLdsByteCodes)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
}

// Both tables are contiguous in LDS ([stage1 | stage2]) and in global memory (api.cpp uploads them back to back), so the
// copy is one stream of kTablesLdsBytes / 16 vectors; all of a thread's loads are issued before its first LDS write.
template <int NT = kWPB * 64, int MODE = kModeBits>
__device__ __forceinline__ void load_tables(uint8_t* lds, const SplitParams& P) {
    if (mode_base(MODE) == kModeLatin1) {
        build_latin1_tables<NT>(lds, P);
        return;
    }
    if (mode_base(MODE) == kModeBytes) {
        // byte space: its own class table -- stage 1 (uint16 offsets) to 0, stage 2 behind it
        build_lead_table<NT>(lds + kLdsLeadTab);
        load_byte_codes(lds, P);
        const uint4* src1 = reinterpret_cast<const uint4*>(P.t1);
        const uint4* src2 = reinterpret_cast<const uint4*>(P.t2);
        uint4* dst1 = reinterpret_cast<uint4*>(lds);
        uint4* dst2 = reinterpret_cast<uint4*>(lds + kB6Stage1Bytes);
        constexpr int kVec1 = kB6Stage1Bytes / 16, kVecB = kVec1 + kB6Stage2Bytes / 16;   // 2240 + 1600
        constexpr int kPerB = (kVecB + NT - 1) / NT;                                       // 5 with 768 threads
        if (kPerB <= 5) {
            uint4 tmp[kPerB <= 5 ? kPerB : 1];   // all of a thread's loads are issued before its first LDS write
#pragma unroll
            for (int j = 0; j < (kPerB <= 5 ? kPerB : 1); ++j) {
                const int i = threadIdx.x + j * NT;
                if (i < kVecB) tmp[j] = i < kVec1 ? src1[i] : src2[i - kVec1];
            }
#pragma unroll
            for (int j = 0; j < (kPerB <= 5 ? kPerB : 1); ++j) {
                const int i = threadIdx.x + j * NT;
                if (i < kVecB) { if (i < kVec1) dst1[i] = tmp[j]; else dst2[i - kVec1] = tmp[j]; }
            }
        } else {
            for (int i = threadIdx.x; i < kVec1; i += NT) dst1[i] = src1[i];
            for (int i = threadIdx.x; i < kVecB - kVec1; i += NT) dst2[i] = src2[i];
        }
        if (threadIdx.x == 0) {   // (tables_ensure_bytes: nothing left to fetch)
            int* ctl = reinterpret_cast<int*>(lds + lds_shift(kModeBytes) + kLdsMisc) + 16;
            ctl[0] = ctl[1] = 64 * kLazyGrabs;
        }
        return;
    }
    constexpr int kVec = kTablesLdsBytes / 16;                    // 2585
    constexpr int kPer = (kVec + NT - 1) / NT;                    // 4 with 768 threads
    const uint4* src = reinterpret_cast<const uint4*>(P.t1);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    if (kPer <= 4) {
        uint4 tmp[kPer <= 4 ? kPer : 1];
#pragma unroll
        for (int j = 0; j < (kPer <= 4 ? kPer : 1); ++j) {
            const int i = threadIdx.x + j * NT;
            if (i < kVec) tmp[j] = src[i];
        }
#pragma unroll
        for (int j = 0; j < (kPer <= 4 ? kPer : 1); ++j) {
            const int i = threadIdx.x + j * NT;
            if (i < kVec) dst[i] = tmp[j];
        }
    } else {
        for (int i = threadIdx.x; i < kVec; i += NT) dst[i] = src[i];
    }
}

// k_tiles_main in byte space: the ASCII part of the class table now, the rest when a tile asks for it (tables_ensure_bytes)
__device__ __forceinline__ void load_tables_ascii_bytes(uint8_t* lds, const SplitParams& P) {
    load_byte_codes(lds, P);
    if (threadIdx.x == 16) {
        int* ctl = reinterpret_cast<int*>(lds + lds_shift(kModeBytes) + kLdsMisc) + 16;
        ctl[0] = ctl[1] = 0;
    }
}
// the rest of it: pieces of kLazyPer rows of 1 KiB, handed out by ctl[0]; ctl[1] counts the pieces that are in place.  The rows go
// from global memory straight to LDS (global_load_lds_dwordx4: wave-uniform LDS base + 16 x lane, no registers in between -- the
// wave that fetches holds its tile's bytes in registers), the last piece is the byte decode table, which is computed.
__device__ __attribute__((noinline, cold)) void tables_fetch_bytes(const uint8_t* t1, const uint8_t* t2, uint8_t* tables, int* ctl) {
    const int lane = (int)(threadIdx.x & 63u);
    struct { const uint8_t* t1; const uint8_t* t2; } P = {t1, t2};
    struct { uint8_t* tables; int* ctl; } L = {tables, ctl};
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;
    for (;;) {
        const int ticket = __hip_atomic_fetch_add(&L.ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int g = __builtin_amdgcn_readfirstlane(ticket) >> 6;                 // wave-uniform: 64 tickets per piece
        if (g >= kLazyGrabs) break;
        if (g == kLazyGrabs - 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const lk_lead_entry e = lk_lead_entry_of((uint32_t)(64 * j + lane));
                reinterpret_cast<uint2*>(L.tables + kLdsLeadTab)[64 * j + lane] = make_uint2(e.sel, e.hi0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kLazyPer; ++j) {
                const int row = g * kLazyPer + j;                                  // wave-uniform
                if (row < kLazyRows)   // (both stages are back to back, in global memory as in LDS)
                    __builtin_amdgcn_global_load_lds((gptr_t)(P.t1 + 1024 * row + 16 * lane), (lptr_t)(L.tables + 1024 * row), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the rows have landed
        }
        // (the LDS executes one wave's instructions in order: the piece is in place before the count says so)
        __hip_atomic_fetch_add(&L.ctl[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    while (__hip_atomic_load(&L.ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 64 * kLazyGrabs) __builtin_amdgcn_s_sleep(2);
}

template <int MODE = kModeBits>
__device__ __forceinline__ TileLds wave_lds(uint8_t* lds, int wave) {
    TileLds L;
    L.t1 = lds;
    L.t2 = lds + kStage1Pad;
    uint8_t* const w = lds + lds_shift(MODE);
    uint8_t* mine = wave < kWPB ? w + kLdsWaves + wave * kWaveLdsBytes : lds + lds_total(MODE) + (wave - kWPB) * kWaveLdsBytes;
    L.stage = mine;
    L.halo = mine + kStageBytes;
    L.bw = reinterpret_cast<lk_u64*>(mine + kStageBytes + 16);
    L.lut = lds;                                                     // build_latin1_tables (kModeLatin1: in place of the Unicode tables)
    L.ltab = lds + kLdsLeadTab;
    L.tables = lds;
    L.ctl = reinterpret_cast<int*>(w + kLdsMisc) + 16;
    L.t1b = lds;
    L.t2b = lds + kB6Stage1Bytes;
    L.ctab = mode_base(MODE) == kModeBytes ? lds + kLdsByteCodes : L.lut + kSliceLutBytes;   // (byte space: code of a byte taken as a char, 0 from 0x80 on)
    L.small_bits = L.small_space = nullptr;
    return L;
}

// ---------------------------------------------------------------------------------------------------------------
// stage 1: the tiles.  The tile range is cut into segments of P.seg_tiles (<= 1024) consecutive tiles; a workgroup
// owns whole segments (grid-stride), its kWPB waves take the segment's tiles round-robin.  Per segment the workgroup
//   (a) fetches the first string of each of its tiles from the index stage 0 (k_tile_index) left in P.tile_first,
//   (b) runs the tiles, each publishing its 16-byte summary to LDS and to global memory,
//   (c) composes the segment's transfer function / head descriptor with a block-wide scan -> one aggregate per segment.
// ---------------------------------------------------------------------------------------------------------------
#ifdef LATOK_STAMPS
#define LATOK_STAMP_ARG , stamp_acc
#define LATOK_STAMP_PARAM , unsigned long long* stamp_acc
#define LATOK_STAMP_NULL , nullptr
#else
#define LATOK_STAMP_ARG
#define LATOK_STAMP_PARAM
#define LATOK_STAMP_NULL
#endif

template <int MODE, bool FAST_TAIL = false, int WPB = kWPB, int PF = kCpsPrefetchRows>
__device__ __forceinline__ void run_segment(const SplitParams& P, uint8_t* lds, int64_t seg, int tid, int lane,
                                            int wave, bool tables LATOK_STAMP_PARAM) {
    const int S = P.seg_tiles;
    const TileLds L = wave_lds<MODE>(lds, wave);
    int4* sm = reinterpret_cast<int4*>(lds + lds_shift(MODE) + kLdsSumm);
    ScanLdsT<WPB>& scan = *reinterpret_cast<ScanLdsT<WPB>*>(lds + lds_shift(MODE) + kLdsScan);

    const int64_t T0 = seg * S;
    const int64_t T1 = min(T0 + S, P.n_tiles);
    const int n_seg = (int)(T1 - T0);
    // (a) first string of each tile: P.tile_first, written by k_tile_index before this kernel.  A wave takes the tiles
    //     wave, wave + kWPB, ... of the segment -- at most 64 -- so one load per lane holds the whole segment's worth.
    int64_t tfv = 0;
    if (wave + WPB * lane < n_seg) tfv = P.tile_first[T0 + wave + WPB * lane];
    tfv = tfv < 0 ? 0 : (tfv > P.n_str ? P.n_str : tfv);   // (row offsets that are not non-decreasing leave holes in the index)
    if (tables) __syncthreads();   // the class tables (first segment of the workgroup) are in LDS
    // (b) the tiles
    // Output write combining (bitmask mode): the words of up to 8 tiles stay in registers and are stored together.
    // One 512-byte store per 16 KiB tile, interleaved with the read stream, costs ~5 % of HBM throughput.
    // UTF-32 input only: that kernel is HBM bound.  The narrow-input kernels are bound by instruction issue and registers: without
    // the 16 VGPRs of the buffer Latin-1 104 VGPRs (was 121), UCS-2 123 + 64 B scratch (128 + 96), byte space 128 B scratch (192);
    // C3 byte space 0.538 -> 0.506 ms, UCS-2 0.157 -> 0.142, C2 byte space 0.085 -> 0.073, Latin-1 0.069 -> 0.067.
    constexpr bool kDefer = mode_writes_bits(MODE) && !mode_is_bytes(MODE);
    lk_u64 obuf[8];
    int slot = 0, k_first = wave;   // buffered tiles are k_first, k_first + kWPB, ...
    const int64_t n_words = (P.total + 63) >> 6;
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < slot) {
                const int64_t w = (T0 + k_first + j * WPB) * 64 + lane;
                if (w < n_words) P.bits_out[w] = obuf[j];
            }
        }
        slot = 0;
    };
    auto put = [&](lk_u64 w, int k) {
        if (slot == 0) k_first = k;
#pragma unroll
        for (int j = 0; j < 8; ++j) obuf[j] = (j == slot) ? w : obuf[j];
        if (++slot == 8) flush();
    };
    CpsPrefetch pf;
    pf.valid = false;
    for (int k = wave, j = 0; k < n_seg; k += WPB, ++j) {
        const lk_u64 w = process_tile<MODE, kDefer, false, FAST_TAIL, PF>(P, L, T0 + k, lane_read64(tfv, j), 0, -1, true, &sm[k], lane LATOK_STAMP_ARG
                                                                      , MODE == kModeBits && !FAST_TAIL ? &pf : nullptr, k + WPB < n_seg ? T0 + k + WPB : (int64_t)-1
                                                                      );
        if (kDefer) put(w, k);
#ifdef LATOK_STAMPS
        stamp_acc[0] += 1;
#endif
    }
    if (kDefer) flush();
    __syncthreads();
    // (c) segment aggregate
    {
        Fn64 f = fn_identity();
        Hd64 h = hd_identity();
        if (tid < n_seg) {
            const int4 s = sm[tid];
            P.summ[T0 + tid] = s;            // coalesced: 16 B per thread, one burst per segment
            f = fn_of(s);
            h.h = s.z; h.c = s.w & 1;
        }
        Fn64 ef, tfn; Hd64 eh, th;
        block_scan<WPB>(f, h, scan, &ef, &eh, &tfn, &th);
        if (tid == 0) { P.seg_fn[seg] = tfn; P.seg_hd[seg] = th; }
    }
    __syncthreads();
}

// FAST_TAIL: the batch's last, partial tile requests all its rows before the first lookup (process_tile).  Chosen for small
// batches, where that tile's latency is a visible share of the call (a 4-tile batch in pinned host memory: 31 -> 16 us);
// the large-batch instantiation keeps the serial tail: the unified form costs its full-tile loop 16 VGPRs and 4 % on C2.
template <int MODE, bool FAST_TAIL = false, int PF = kCpsPrefetchRows>
__global__ __launch_bounds__(tile_wpb(MODE) * 64) void k_tiles_main(SplitParams P) {
    constexpr int WPB = tile_wpb(MODE);
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_total_tiles(MODE)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> scalar tile arithmetic
    if (blockIdx.x == 0 && tid == 0) *P.fix_count = 0;            // statistics counter of the resolve stage

    bool tables = MODE != kModeBlockMask;
    // (published by the barrier at the top of the workgroup's first segment)
    if (tables) {
        if (mode_base(MODE) == kModeBytes) load_tables_ascii_bytes(lds, P);
        else load_tables<WPB * 64, MODE>(lds, P);
    }
#ifdef LATOK_STAMPS
    unsigned long long stamp_acc[16];
    for (int i = 0; i < 16; ++i) stamp_acc[i] = 0;
#endif
    for (int64_t seg = blockIdx.x; seg < P.n_segs; seg += gridDim.x) {
        run_segment<MODE, FAST_TAIL, WPB, PF>(P, lds, seg, tid, lane, wave, tables LATOK_STAMP_ARG);
        tables = false;
    }
#ifdef LATOK_STAMPS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_stamp_sum[i], stamp_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// stage 2: resolve + repair.  Same segment ownership.  Per segment the workgroup composes the aggregates of the
// segments before it (pending starts entering the segment) and after it (starts before the next closing event),
// scans its own tiles, and for every tile whose provisional assumptions (no pending start enters, tail block decided
// by "pending at the tile end") were wrong: patches the bitmask in place for the common cases, or recomputes the tile
// with the exact inputs (the same tile code; Unicode tables are only copied to LDS if that ever happens).
// ---------------------------------------------------------------------------------------------------------------
// NW = waves per workgroup: NW * 64 threads must cover a segment's tiles (one tile per thread).  Batches whose segments
// are short (C2: 122 tiles) run it with 2 or 4 waves instead of 12 -- the stage is all latency, fewer waves start faster.
// Stage 2.  ONE = false: k_resolve_fix, every segment of the batch, a workgroup per segment at a time.  ONE = true: the only
// segment of a small batch, inside the launch that computed it (k_one_segment; the class tables are still in LDS).
template <int MODE, int NW, bool ONE>
__device__ __forceinline__ void resolve_segments(const SplitParams& P, uint8_t* lds, int tid, int lane, int wave) {
    const int S = P.seg_tiles;
    ScanLdsT<NW>& scan = *reinterpret_cast<ScanLdsT<NW>*>(lds + lds_shift(MODE) + kLdsScan);
    int* misc = reinterpret_cast<int*>(lds + lds_shift(MODE) + kLdsMisc);          // misc[0] = number of tiles to recompute
    int* fix_t = reinterpret_cast<int*>(lds + lds_shift(MODE) + kLdsTf);           // tile index inside the segment (tiles to recompute)
    int2* fix_in = reinterpret_cast<int2*>(lds + lds_shift(MODE) + kLdsSumm);      // {q_in, tail_zero}
    bool tables_loaded = ONE;

    for (int64_t seg = ONE ? 0 : (int64_t)blockIdx.x; seg < (ONE ? 1 : P.n_segs); seg += ONE ? 1 : (int64_t)gridDim.x) {
        const int64_t T0 = seg * S;
        const int64_t T1 = min(T0 + S, P.n_tiles);
        const int n_seg = (int)(T1 - T0);
        // my tile's summary is requested first: its latency overlaps the scan of the segment aggregates
        if (tid == 0) misc[0] = 0;
        const int64_t t = T0 + tid;
        int4 s = make_int4(0, 0, 0, 0);
        if (tid < n_seg) s = P.summ[t];
        // (1) aggregates of the other segments: contiguous range per thread, ordered
        Fn64 pf = fn_identity();
        Hd64 sh = hd_identity();
        if (P.n_segs > 1) {
            const int64_t per = (P.n_segs + NW * 64 - 1) / (NW * 64);
            const int64_t lo = min((int64_t)tid * per, P.n_segs), hi = min(lo + per, P.n_segs);
            Fn64 f = fn_identity();
            Hd64 h = hd_identity();
            for (int64_t j = lo; j < hi; ++j) {
                if (j < seg) f = fn_then(f, P.seg_fn[j]);
                if (j > seg) h = hd_then(h, P.seg_hd[j]);
            }
            Fn64 ef; Hd64 eh;
            block_scan<NW>(f, h, scan, &ef, &eh, &pf, &sh);
        }
        const long long q_seg_in = fn_apply(pf, 0);
        Hd64 rest; rest.h = sh.h; rest.c = 1;   // what follows the segment: sh.h starts before the next closing

        // (2) my tile
        Fn64 f = fn_identity();
        Hd64 h = hd_identity();
        if (tid < n_seg) {
            f = fn_of(s);
            h.h = s.z; h.c = s.w & 1;
        }
        Fn64 ef, tfn; Hd64 eh, th;
        block_scan<NW>(f, h, scan, &ef, &eh, &tfn, &th);   // (its barriers also publish misc[0] = 0)
        if (tid < n_seg) {
            const long long q_in = fn_apply(ef, q_seg_in);
            const long long q_end = fn_apply(f, q_in);
            const long long h_next = hd_then(eh, rest).h;
            const int tz = (q_end + h_next) > 0;
            const int tz0 = s.y > 0;
            if (q_in != 0 || tz != tz0) {
                const int geom = s.w;
                if (!mode_rules(MODE) && (MODE == kModeBits || mode_is_bytes(MODE)) && (geom & 1) && q_in <= 1 &&
                    (q_in == 0 || s.z == 0)) {
                    // Patch in place: one pending start entering a tile whose head block has no start of its own
                    // zeroes that head block; a tail block that turns out to be zeroed is cleared.  What stays in a
                    // cleared block: the C_SYM bit of its last char and the bit of a string start.
                    // (byte space, tiles with multi-byte chars: the last char's bit is at its lead byte, up to 3 bytes before the
                    // block's last position -- found in the bytes themselves)
                    const int64_t t0 = t * kTile;
                    const int64_t t_end = min(t0 + kTile, P.total);
                    auto lead_back = [&](int64_t lo, int64_t hi) -> int {
                        int back = 0;
                        if (mode_base(MODE) == kModeBytes && ((geom >> 30) & 1)) {
                            if (hi > t_end) hi = t_end;
                            while (back < 3 && hi - 1 - back > lo && (P.u8[hi - 1 - back] & 0xC0u) == 0x80u) ++back;
                        }
                        return back;
                    };
                    if (q_in == 1) {
                        const int64_t hi = t0 + ((geom >> 1) & 0x1FFF);
                        const int keep = (geom >> 27) & 1;
                        clear_range(P.bits_out, t0, hi, t_end, 0, keep, keep ? lead_back(t0, hi) : 0);
                    }
                    if (tz != tz0) {
                        const int64_t lo = t0 + ((geom >> 14) & 0x1FFF);
                        const int keep = (geom >> 29) & 1;
                        clear_range(P.bits_out, lo, t_end, t_end, (geom >> 28) & 1, keep, keep ? lead_back(lo, t_end) : 0);
                    }
                } else {
                    const int slot = atomicAdd(&misc[0], 1);
                    fix_t[slot] = tid;
                    fix_in[slot] = make_int2((int)(q_in < (1 << 20) ? q_in : (1 << 20)), tz);   // <= 4096 closings/tile
                }
            }
        }
        __syncthreads();
        // (3) recompute what could not be patched (rare)
        const int n_fix = misc[0];
        if (n_fix > 0) {
            if (!tables_loaded && MODE != kModeBlockMask) {
                load_tables<NW * 64, MODE>(lds, P);
                tables_loaded = true;
                __syncthreads();
            }
            const TileLds L = wave_lds<MODE>(lds, wave);
            for (int i = wave; i < n_fix; i += NW) {
                const int64_t tt = T0 + fix_t[i];
                const int2 in = fix_in[i];
                const int64_t idx0 = wave_lower_bound(P.row_off, P.n_str + 1, tt * kTile, lane);
                process_tile<MODE>(P, L, tt, idx0, in.x, in.y, false, nullptr, lane);
            }
            if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(P.fix_count), (unsigned long long)n_fix);
        }
        __syncthreads();
    }
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_resolve_fix(SplitParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_total(MODE)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    resolve_segments<MODE, NW, false>(P, lds, tid, lane, wave);
    signal_block_done(P.done);
}

// stage 0 (k_tile_index below, k_one_segment): entry s of row_off (s > n_str: nothing); must be called by whole waves (the wide form below uses wave operations)
__device__ __forceinline__ void tile_index_entry(const int64_t* __restrict__ row_off, int64_t n_str, int64_t n_tiles,
                                                 int64_t* __restrict__ tile_first, int64_t s, int lane) {
    int64_t w0 = 0, w1 = -1;
    if (s <= n_str) {
        const int64_t p = row_off[s];
        const int64_t prev = s > 0 ? row_off[s - 1] : -1;
        w0 = prev < 0 ? 0 : prev / kTile + 1;
        w1 = p / kTile;
        if (w1 > n_tiles - 1) w1 = n_tiles - 1;
    }
    const bool wide = w1 - w0 >= 16;
    if (!wide)
        for (int64_t w = w0; w <= w1; ++w) tile_first[w] = s;
    unsigned long long m = __ballot(wide);
    while (m) {   // wave-uniform
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int64_t a = lane_read64(w0, src), b = lane_read64(w1, src), v = lane_read64(s, src);
        for (int64_t w = a + lane; w <= b; w += 64) tile_first[w] = v;
    }
}


// A batch of at most kOneSegTiles tiles (P.n_segs == 1, P.seg_tiles >= P.n_tiles): the three stages in ONE launch of one
// workgroup -- the per-tile string index, the tiles, the resolve stage -- with workgroup barriers where the stream order
// of the three launches was.  Three dependent launches of a few microseconds of work each cost ~5 us apiece in dispatch
// and drain; a host batch of 40 ... 1000 short strings is nothing but that.
template <int MODE>
__global__ __launch_bounds__(kWPB * 64) void k_one_segment(SplitParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_total(MODE)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *P.fix_count = 0;
    load_tables<kWPB * 64, MODE>(lds, P);                       // (published by the barriers below)
    for (int64_t base = 0; base <= P.n_str; base += kWPB * 64)  // stage 0
        tile_index_entry(P.row_off, P.n_str, P.n_tiles, P.tile_first, base + tid, lane);
    __threadfence_block();
    __syncthreads();
    run_segment<MODE, true>(P, lds, 0, tid, lane, wave, true LATOK_STAMP_NULL);   // stage 1 (ends with a barrier)
    __threadfence_block();
    __syncthreads();
    resolve_segments<MODE, kWPB, true>(P, lds, tid, lane, wave);                  // stage 2
    signal_block_done(P.done);
}

// flags[0] = any(a1 != 0), flags[1] = any(a2 != 0) (flags zeroed by the caller)
__global__ void k_any_nonzero(const int8_t* __restrict__ a1, const int8_t* __restrict__ a2, int64_t n,
                              int* __restrict__ flags) {
    int f1 = 0, f2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f1 |= a1[i] != 0;
        f2 |= a2[i] != 0;
    }
    if (__any(f1) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
    if (__any(f2) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1);
}

hipError_t launch_any_nonzero(const int8_t* a1, const int8_t* a2, int64_t n, int* flags, hipStream_t st) {
    hipError_t e = hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
    if (e != hipSuccess || n <= 0) return e;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_any_nonzero, dim3((unsigned)blocks), dim3(256), 0, st, a1, a2, n, flags);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// featurize: per-token sums of the 25 feature columns (reference default_tokenizer.py:163-191), one wave
// per tile like the split kernel.  The tile's chars are classified into rule codes and bit-sliced exactly as above,
// all 25 feature planes of a word are built in registers (lk_feature_planes), and the sum of column c over a token is
// popcount(plane_c & token_span): the work per word is proportional to its tokens, not its chars, and the n x 25
// matrix never exists.  A token that runs past its word takes the "head" sums (chars before the first boundary) of the
// following words from the neighbour lanes; one that runs past the tile is finished char by char (rare).  The 25
// bytes of a token are packed in 7 dwords (byte-wise wrap-around adds = the reference's uint8 arithmetic) and leave
// through the wave's staging buffer as one contiguous stream.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t swar_add_u8(uint32_t a, uint32_t b) {
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}
struct FeatSums {
    uint32_t v[7];   // byte c of the 28 = column c (bytes 25..27 unused)
};
__device__ __forceinline__ FeatSums feat_popc(const lk_planes& F, lk_u64 m) {
    FeatSums r;
#pragma unroll
    for (int j = 0; j < 7; ++j) r.v[j] = 0;
#pragma unroll
    for (int c = 0; c < LK_N_FEATURES; ++c)
        r.v[c >> 2] |= (uint32_t)__popcll(LK_PLANE_GET(F, c) & m) << (8 * (c & 3));   // <= 64: no byte overflow
    return r;
}
// the same over one 32-bit half of the word (HI = 0: chars 0..31, 1: chars 32..63): a token of a few chars lies in one half,
// so its 25 sums cost one and + one popcount per column instead of two
template <int HI>
__device__ __forceinline__ FeatSums feat_popc_half(const lk_planes& F, uint32_t m) {
    FeatSums r;
#pragma unroll
    for (int j = 0; j < 7; ++j) r.v[j] = 0;
#pragma unroll
    for (int c = 0; c < LK_N_FEATURES; ++c)
        r.v[c >> 2] |= (uint32_t)__popc(LK_PLANE_HALF(F, c, HI) & m) << (8 * (c & 3));   // <= 32: no byte overflow
    return r;
}
__device__ __forceinline__ uint32_t feat_row_bits1(uint32_t w, uint32_t p, uint32_t x, uint32_t y, bool first, bool last) {
    // 25 columns of one char from base words (aux_kernels.hip:feature_row_bits, same bit layout)
    uint32_t r = w & 0xFFFu;
    r |= ((p >> 0) & 1u) << 12; r |= ((x >> 0) & 1u) << 13; r |= ((p >> 1) & 1u) << 14; r |= ((x >> 1) & 1u) << 15;
    r |= ((p >> 3) & 1u) << 16; r |= ((x >> 3) & 1u) << 17;
    r |= (first ? 1u : (p >> 5) & 1u) << 18; r |= (last ? 1u : (x >> 5) & 1u) << 19;
    r |= ((p >> 6) & 1u) << 20; r |= ((x >> 8) & 1u) << 21; r |= ((x >> 10) & 1u) << 22;
    r |= ((y >> 0) & 1u) << 23; r |= ((y >> 10) & 1u) << 24;
    return r;
}

// Lane = word while the sums are computed, but the output is token-major (all tokens of lane 0, then lane 1, ...), so
// a tile's records have to meet in LDS before they can leave as a stream.  A window that only holds part of a tile
// forces rounds in which most lanes idle (192-token rounds: 5x the instructions); writing the 25-byte records straight
// to global memory costs +0.4 ms in scattered stores.  So this kernel trades waves for LDS: kFeatWaves waves per CU,
// each with a window for kFeatRound tokens (a 4096-char tile of word-soup text has ~830), which doubles as the
// code-byte staging buffer before the planes are built.
constexpr int kFeatWaves = 7;                                         // (6 -> 7: C2 -4.5 %, C3 -6 %; 8 would need rounds of < 800 tokens: two rounds per C2 tile)
constexpr int kFeatRound = 896;                                       // tokens per round (word-major form)
constexpr int kFeatRec = 25;                                          // packed records in the window, as in the output
constexpr int kFeatRoundTm = 768;                                     // token-major form: records + 2-byte (lane, bit) codes share the window
constexpr int kFeatWinBytes = kFeatRound * kFeatRec + 16;             // (+ 16: the records start at record_shift(dst); token-major rounds cost nothing extra)
static_assert(kFeatRoundTm * (kFeatRec + 2) + 16 <= kFeatWinBytes, "token-major round fits the window");
constexpr int kFeatWaveLds = kFeatWinBytes + 16 + 66 * 8;             // window | (unused) | string-start words
constexpr int kFeatLdsTotal = kFeatWaves * kFeatWaveLds;
static_assert(kFeatLdsTotal <= 160 * 1024, "LDS budget of one CU");
static_assert(kFeatWinBytes % 16 == 0 && kFeatWaveLds % 16 == 0, "alignment");

// A token's 25 sums enter the window packed (25-byte stride: the dword stores become byte stores).  Padding the records
// to 28 or 32 bytes for aligned LDS stores was measured: the LDS pipe's busy time drops 2.5x (PMC), but the un-padding
// on the way out costs more instructions than it saves and only 832 tokens fit a round at six waves: 796 vs 691 us on
// C2 (32-byte records: every slot lands in one of four bank groups, 913 us).  The kernel is bound by the dependent
// chain of ~7 K instructions per tile at 1.5 waves per SIMD, not by a pipe.
__device__ __forceinline__ void put_record(uint8_t* win, int slot, const FeatSums& s) {
    uint8_t* rec = win + slot * kFeatRec;
#pragma unroll
    for (int q = 0; q < 6; ++q) __builtin_memcpy(rec + 4 * q, &s.v[q], 4);
    rec[24] = (uint8_t)s.v[6];
}
// Stream n_rec records out as n_rec * 25 contiguous bytes at dst (any alignment).  The records were put at
// win + record_shift(dst): LDS and global address then agree modulo 16, so after at most 15 head bytes the stream leaves as
// aligned 16-byte vectors (one ds_read_b128 + one global_store_dwordx4 per lane and step).  With the records at the
// window's start the LDS side was unaligned whenever dst was: four byte reads + shifts per dword, ~800 of the kernel's
// ~7 K instructions per tile.
__device__ __forceinline__ int record_shift(const void* dst) { return (int)((uintptr_t)dst & 15u); }
__device__ __forceinline__ void flush_records(const uint8_t* win, int n_rec, uint8_t* dst, int lane) {
    const int n_bytes = n_rec * 25;
    const uint8_t* src = win + record_shift(dst);
    const int head = min((16 - record_shift(dst)) & 15, n_bytes);
    if (lane < head) dst[lane] = src[lane];
    const int n_vec = (n_bytes - head) >> 4;
    for (int i = lane; i < n_vec; i += 64) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + head + 16 * i);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + head + 16 * i));
    }
    const int tail0 = head + 16 * n_vec;
    if (lane < n_bytes - tail0) dst[tail0 + lane] = src[tail0 + lane];
}
// span records (4 x OUT per token) go through the same window in rounds of what fits
template <typename OUT>
constexpr int span_round() { return kFeatWinBytes / (4 * (int)sizeof(OUT)) < kFeatRound ? kFeatWinBytes / (4 * (int)sizeof(OUT)) : kFeatRound; }

// 64 rule codes of one word (16-byte aligned) -> d[16]
__device__ __forceinline__ void load_codes64(const uint8_t* __restrict__ p, uint32_t (&d)[16]) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x4 v = q[k];
        d[4 * k + 0] = v.x; d[4 * k + 1] = v.y; d[4 * k + 2] = v.z; d[4 * k + 3] = v.w;
    }
}

template <typename OUT>
__device__ __forceinline__ void feature_tile(const FeatParams& P, const TileLds& L, int64_t t, int lane) {
    OUT* const spans4 = reinterpret_cast<OUT*>(P.spans4);   // int64 or int32 records (LATOK_OUT_INT32)
    const int64_t t0 = t * kTile;
    const int64_t total = P.total;
    const int64_t n_words = (total + 63) >> 6;
    const int64_t w = t * 64 + lane;
    const int n_wave = (int)P.tile_cnt[t];
    if (n_wave == 0) return;
    const lk_u64 x = w < n_words ? P.kept[w] : 0ull;          // kept tokens that start in my word
    const int off = w < n_words ? (int)P.word_pref[w] : 0;
    const int64_t base_out = P.tile_rank[t];
    const lk_u64 xb = w < n_words ? P.bits[w] : 0ull;         // all boundaries of my word

    // ---- string starts of the tile (+ the two words behind it) as bits in LDS, from the per-tile string index --------
    int64_t idx0 = P.tile_first[t];
    idx0 = idx0 < 0 ? 0 : (idx0 > P.n_str ? P.n_str : idx0);
    // start of the string that is open when the tile begins (spans are string relative)
    const int64_t start_before = idx0 > 0 ? P.row_off[idx0 - 1] : 0;
    int64_t ro = idx0 + lane <= P.n_str ? P.row_off[idx0 + lane] : INT64_MAX;
    // ---- my word: 64 rule codes (1 B/char, left by the tile kernel: P.codes; padded behind `total`), the three
    //      neighbour codes from the neighbour lanes, and the 25 planes ------------------------------------------------
    const int64_t base = t0 + 64 * (int64_t)lane;
    const int64_t remain = total - base;
    const lk_u64 valid = remain >= 64 ? ~0ull : (remain <= 0 ? 0ull : ((1ull << remain) - 1ull));
    uint32_t d[16];
    load_codes64(P.codes + base, d);
    uint32_t edge = 0;                                    // lane 0: code of char t0-1; lane 63: codes of t0+4096, t0+4097
    if (lane == 0 && t0 > 0) edge = P.codes[t0 - 1];
    if (lane == 63) {
        if (t0 + kTile < total) edge = P.codes[t0 + kTile];
        if (t0 + kTile + 1 < total) edge |= (uint32_t)P.codes[t0 + kTile + 1] << 8;
    }
    L.bw[lane] = 0;
    if (lane < 2) L.bw[64 + lane] = 0;   // one word more than the split kernel: the first word of the next tile is needed
    wave_lds_sync();
    for (;;) {
        const int64_t rel = ro - t0;
        if (rel >= 0 && rel < kTile + 128) atomicOr(&L.bw[rel >> 6], 1ull << (rel & 63));
        const int64_t last = lane_read64(ro, 63);
        if (last >= t0 + kTile + 128) break;
        idx0 += 64;
        ro = idx0 + lane <= P.n_str ? P.row_off[idx0 + lane] : INT64_MAX;
    }
    wave_lds_sync();
    const lk_u64 B = L.bw[lane];
    const lk_u64 Bn = L.bw[lane + 1] & 3ull;
    lk_planes F;
    {
        lk_halo h;
        const uint32_t up = (uint32_t)dpp_mov<kDppWaveShr1, 0xF>(0, (int)(d[15] >> 24));      // from lane - 1
        const uint32_t dn = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)(d[0] & 0xFFFFu));   // from lane + 1
        h.prev = lane > 0 ? up : edge;
        h.next0 = lane < 63 ? (dn & 0xFFu) : (edge & 0xFFu);
        h.next1 = lane < 63 ? (dn >> 8) : (edge >> 8);
        lk_u64 plane[8];
        lk_bitslice64(d, plane);
        lk_feature_planes(plane, h, B, Bn, F);
    }
    const uint32_t prev65 = (uint32_t)lane_read((int)(d[15] >> 24), 63);   // code of the tile's last char

    // ---- what a token that leaves my word collects from the following words -----------------------------------------
    const lk_u64 head_mask = (xb ? ((xb & (~xb + 1ull)) - 1ull) : ~0ull) & valid;
    const FeatSums H = feat_popc(F, head_mask);
    const int full = xb == 0;
    const int top = xb ? 63 - __builtin_clzll(xb) : 0;
    const bool need_tail = xb != 0 && ((x >> top) & 1ull);     // my last boundary starts a kept token: it continues
    FeatSums C;
#pragma unroll
    for (int j = 0; j < 7; ++j) C.v[j] = 0;
    bool open = need_tail;                                     // still collecting
    lk_u64 xb_next_tile = 0, nn_next_tile = 0;                 // boundary / non-SPACE masks of the next tile's first word
    FeatSums Hs = H;                                           // H / full of lane + d: one more DPP shift per step
    int fs = full;
    for (int d = 1; d < 64; ++d) {
        if (!__ballot(open && lane + d < 64)) break;
#pragma unroll
        for (int j = 0; j < 7; ++j) Hs.v[j] = (uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)Hs.v[j]);
        fs = dpp_mov<kDppWaveShl1, 0xF>(0, fs);
        if (open && lane + d < 64) {
#pragma unroll
            for (int j = 0; j < 7; ++j) C.v[j] = swar_add_u8(C.v[j], Hs.v[j]);
            open = fs != 0;
        }
    }
    if (__ballot(open)) {
        // A token runs past the tile (the usual case for its last token): every lane builds the planes of the next tile's
        // first word (the same 64 codes, a broadcast load) and the open lanes take its head sums.
        const int64_t q0 = t0 + kTile;
        uint32_t d65[16];
        load_codes64(P.codes + q0, d65);
        lk_halo h65;
        h65.prev = prev65;                         // code of the tile's last char
        h65.next0 = q0 + 64 < total ? (uint32_t)P.codes[q0 + 64] : 0u;
        h65.next1 = q0 + 65 < total ? (uint32_t)P.codes[q0 + 65] : 0u;
        lk_u64 plane65[8];
        lk_bitslice64(d65, plane65);
        lk_planes F65;
        const lk_u64 B65 = L.bw[64], Bn65 = L.bw[65] & 3ull;
        lk_feature_planes(plane65, h65, B65, Bn65, F65);
        const int64_t rem65 = total - q0;
        const lk_u64 valid65 = rem65 >= 64 ? ~0ull : (rem65 <= 0 ? 0ull : ((1ull << rem65) - 1ull));
        const lk_u64 xb65 = (q0 >> 6) < n_words ? P.bits[q0 >> 6] : 0ull;
        const lk_u64 hm65 = (xb65 ? ((xb65 & (~xb65 + 1ull)) - 1ull) : ~0ull) & valid65;
        const FeatSums H65 = feat_popc(F65, hm65);
        xb_next_tile = xb65;
        nn_next_tile = ~LK_PLANE_GET(F65, 5) & valid65;
        if (open) {
#pragma unroll
            for (int j = 0; j < 7; ++j) C.v[j] = swar_add_u8(C.v[j], H65.v[j]);
            open = xb65 == 0 && q0 + 64 < total;
        }
    }
    const lk_u64 open_m = __ballot(open);
    if (open_m) {
        // Still open after the next tile's first word: a token of more than 64 chars that leaves the tile (at most one
        // lane: the owner of the tile's last kept token -- a masked URL, a long run of letters; a 1 M-char document
        // without whitespace is ONE such token).  The whole wave continues it, 64 words per step, lane = word: the same
        // planes + popcount as above, until the word that holds the next boundary; the partial sums meet in a
        // butterfly and go to the owner.  (The token lies inside one string: the only string start that matters is
        // the string's end.)
        const int owner = lk_ctz(open_m);
        const int64_t from = t0 + kTile + 64;
        // the string that holds the token ends at the first row offset > from - 1 (tokens never cross strings)
        int64_t lo_s = 0, hi_s = P.n_str;
        while (hi_s - lo_s > 1) {
            const int64_t mid = (lo_s + hi_s) >> 1;
            if (P.row_off[mid] <= from - 1) lo_s = mid; else hi_s = mid;
        }
        const int64_t s_end = P.row_off[lo_s + 1];
        FeatSums acc;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc.v[j] = 0;
        for (int64_t c0 = from; c0 < total; c0 += kTile) {
            const int64_t wb = c0 + 64 * (int64_t)lane;
            const lk_u64 xbw = (wb >> 6) < n_words ? P.bits[wb >> 6] : 0ull;
            const lk_u64 hasb = __ballot(xbw != 0ull);
            const int fl = hasb ? lk_ctz(hasb) : 64;                  // lane of the word that holds the next boundary
            uint32_t dw[16];
            load_codes64(P.codes + wb, dw);                            // (in bounds: the code array is padded by a tile)
            uint32_t e2 = 0;
            if (lane == 0) e2 = P.codes[c0 - 1];
            if (lane == 63) {
                if (c0 + kTile < total) e2 = P.codes[c0 + kTile];
                if (c0 + kTile + 1 < total) e2 |= (uint32_t)P.codes[c0 + kTile + 1] << 8;
            }
            const uint32_t up = (uint32_t)__shfl_up((int)(dw[15] >> 24), 1);
            const uint32_t dn = (uint32_t)__shfl_down((int)(dw[0] & 0xFFFFu), 1);
            lk_halo hw;
            hw.prev = lane > 0 ? up : e2;
            hw.next0 = lane < 63 ? (dn & 0xFFu) : (e2 & 0xFFu);
            hw.next1 = lane < 63 ? (dn >> 8) : (e2 >> 8);
            const int64_t rel = s_end - wb;                            // the string's end as a "string start" bit
            const lk_u64 Bw = (rel >= 0 && rel < 64) ? (1ull << rel) : 0ull;
            const lk_u64 Bnw = (rel == 64) ? 1ull : (rel == 65 ? 2ull : 0ull);
            lk_u64 pw[8];
            lk_bitslice64(dw, pw);
            lk_planes Fw;
            lk_feature_planes(pw, hw, Bw, Bnw, Fw);
            const int64_t remw = total - wb;
            const lk_u64 validw = remw >= 64 ? ~0ull : (remw <= 0 ? 0ull : ((1ull << remw) - 1ull));
            lk_u64 m = 0ull;
            if (lane < fl) m = validw;
            else if (lane == fl) m = ((xbw & (~xbw + 1ull)) - 1ull) & validw;
            const FeatSums part = feat_popc(Fw, m);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc.v[j] = swar_add_u8(acc.v[j], part.v[j]);
            if (hasb) break;
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
#pragma unroll
            for (int j = 0; j < 7; ++j) acc.v[j] = swar_add_u8(acc.v[j], (uint32_t)__shfl_xor((int)acc.v[j], sh));
        }
        if (lane == owner) {
#pragma unroll
            for (int j = 0; j < 7; ++j) C.v[j] = swar_add_u8(C.v[j], acc.v[j]);
        }
    }

    // ---- per-word values both forms below need --------------------------------------------------------------------
    const lk_u64 nn = ~LK_PLANE_GET(F, 5) & valid;             // non-SPACE chars of my word
    // the next word's masks (from lane + 1)
    lk_u64 xb1 = (lk_u64)(uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)(uint32_t)xb) |
                 ((lk_u64)(uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)(uint32_t)(xb >> 32)) << 32);
    lk_u64 nn1 = (lk_u64)(uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)(uint32_t)nn) |
                 ((lk_u64)(uint32_t)dpp_mov<kDppWaveShl1, 0xF>(0, (int)(uint32_t)(nn >> 32)) << 32);
    if (lane == 63) { xb1 = xb_next_tile; nn1 = nn_next_tile; }
    // start of the string that is open at my word's first char: the last string start before my word inside the tile,
    // else the one that was open when the tile began
    int last_b = B ? 64 * lane + 63 - __builtin_clzll(B) : -1;  // tile-relative position of my word's last string start
    const int carry = dpp_mov<kDppWaveShr1, 0xF>(-1, wave_scan_max(last_b, -1));   // exclusive: lane 0 gets -1
    const int64_t lo_in = carry >= 0 ? t0 + carry : start_before;

    // ---- skewed tiles (some word holds many more tokens than the mean, e.g. CJK text where every char is a token):
    // token-major form.  Every lane lists its tokens as (lane, bit) codes at their rank inside the tile, then lane j
    // takes the j-th token and pulls the owner word's 25 planes (and masks) through shuffles, so all lanes stay busy.
    // ~70 64-bit shuffles per token make it the slower form for evenly filled tiles, hence the choice per tile.
    const int maxc = wave_max(lk_popc(x), 0);
    // threshold swept on C2 (word-major 9 % faster) and C3 (token-major 15 % faster): fullest word > 1.5 x steps of 64 tokens
    constexpr int kFeatFormThresh = 5;
    // The word-major walk below runs as long as the fullest word (maxc steps of ~160 instructions with the span records), once
    // per window round; the token-major form takes ceil(n_wave / 64) steps of ~300 whatever the spread.
    const int wm_rounds = (n_wave + kFeatRound - 1) / kFeatRound;
    if (maxc * 2 * wm_rounds > ((n_wave + 63) >> 6) * kFeatFormThresh) {
        uint8_t* fwin = L.stage;
        uint16_t* codes = reinterpret_cast<uint16_t*>(L.stage + kFeatRoundTm * kFeatRec + 16);   // behind the feature records
        lk_u64 trest = x;
        int tk = off;
        for (int win0 = 0; win0 < n_wave; win0 += kFeatRoundTm) {
            uint8_t* const fdst = reinterpret_cast<uint8_t*>(P.features) + (base_out + win0) * 25;
            const int shift = record_shift(fdst);
            while (trest && tk < win0 + kFeatRoundTm) {
                const int b = lk_ctz(trest);
                trest &= trest - 1;
                codes[tk - win0] = (uint16_t)((lane << 6) | b);
                ++tk;
            }
            wave_lds_sync();
            const int n_here = min(kFeatRoundTm, n_wave - win0);
            for (int j0 = 0; j0 < n_here; j0 += 64) {
                const int j = j0 + lane;
                const bool active = j < n_here;
                const int code = active ? (int)codes[j] : 0;
                const int owner = code >> 6, b = code & 63;
                const int64_t obase = t0 + 64 * (int64_t)owner;
                const int64_t orem = total - obase;
                const lk_u64 ovalid = orem >= 64 ? ~0ull : (orem <= 0 ? 0ull : ((1ull << orem) - 1ull));
                const lk_u64 o_xb = (lk_u64)__shfl((long long)xb, owner), o_xb1 = (lk_u64)__shfl((long long)xb1, owner),
                             o_nn1 = (lk_u64)__shfl((long long)nn1, owner), o_B = (lk_u64)__shfl((long long)B, owner);
                const int64_t o_lo_in = __shfl((long long)lo_in, owner);
                const lk_u64 above = o_xb & (~1ull << b);
                lk_u64 seg = (~0ull << b) & ovalid;
                if (above) seg &= (above & (~above + 1ull)) - 1ull;
                FeatSums sum;
#pragma unroll
                for (int q = 0; q < 7; ++q) sum.v[q] = 0;
                lk_u64 o_S = 0;
#pragma unroll
                for (int c = 0; c < LK_N_FEATURES; ++c) {
                    const lk_u64 pc = (lk_u64)__shfl((long long)LK_PLANE_GET(F, c), owner);
                    if (c == 5) o_S = pc;
                    sum.v[c >> 2] |= (uint32_t)__popcll(pc & seg) << (8 * (c & 3));
                }
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const uint32_t cq = (uint32_t)__shfl((int)C.v[q], owner);
                    if (!above) sum.v[q] = swar_add_u8(sum.v[q], cq);
                }
                if (active) {
                    put_record(fwin + shift, j, sum);
                    const lk_u64 o_nn = ~o_S & ovalid;
                    const int64_t p = obase + b;
                    const lk_u64 bl = o_B & ((2ull << b) - 1ull);
                    const int64_t lo = bl ? obase + 63 - __builtin_clzll(bl) : o_lo_in;
                    int64_t e, a2, e2;
                    if (above) {
                        const int eb = lk_ctz(above);
                        const lk_u64 sg = o_nn & (~0ull << b) & ((1ull << eb) - 1ull);
                        e = obase + eb;
                        a2 = obase + lk_ctz(sg);
                        e2 = obase + 64 - __builtin_clzll(sg);
                    } else if (o_xb1) {
                        const int eb = lk_ctz(o_xb1);
                        const lk_u64 sg0 = o_nn & (~0ull << b);
                        const lk_u64 sg1 = o_nn1 & ((1ull << eb) - 1ull);
                        e = obase + 64 + eb;
                        a2 = sg0 ? obase + lk_ctz(sg0) : obase + 64 + lk_ctz(sg1);
                        e2 = sg1 ? obase + 128 - __builtin_clzll(sg1) : obase + 64 - __builtin_clzll(sg0);
                    } else {
                        e = next_set_bit(P.bits, obase + 64, total);
                        const lk_u64 sg = o_nn & (~0ull << b);
                        a2 = sg ? obase + lk_ctz(sg) : next_zero_bit(P.space, obase + 64, e);
                        e2 = prev_zero_end(P.space, a2, e);
                    }
                    typedef OUT out2 __attribute__((ext_vector_type(2)));
                    out2* sp = reinterpret_cast<out2*>(spans4 + (base_out + win0 + j) * 4);
                    out2 v0, v1;
                    v0.x = (OUT)(p - lo);
                    v0.y = (OUT)(e - lo);
                    v1.x = (OUT)(a2 - lo);
                    v1.y = (OUT)(e2 - lo);
                    __builtin_nontemporal_store(v0, sp);
                    __builtin_nontemporal_store(v1, sp + 1);
                }
            }
            wave_lds_sync();
            flush_records(fwin, n_here, fdst, lane);
            wave_lds_sync();
        }
        return;
    }

    // ---- tokens of my word, round by round through the staging buffer -------------------------------------------------
    // The walk is split by 32-bit halves: first the tokens that start in chars 0..31 (popcounts on the low halves of the
    // planes only), then the part of the one token that may reach from the low half into the high half, then the tokens that
    // start in chars 32..63 (high halves only).  Ranks grow in that order, so the records land at consecutive slots.
    uint8_t* win = L.stage;
    uint32_t rest_lo = (uint32_t)x, rest_hi = (uint32_t)(x >> 32);
    const uint32_t xb_lo = (uint32_t)xb, xb_hi = (uint32_t)(xb >> 32);
    const uint32_t valid_lo = (uint32_t)valid, valid_hi = (uint32_t)(valid >> 32);
    FeatSums S_str;                       // low-half sums of the straddling token
#pragma unroll
    for (int j = 0; j < 7; ++j) S_str.v[j] = 0;
    int str_slot = -1;                    // its slot (>= 0: the upper part is still to be added)
    int k = off;
    for (int win0 = 0; win0 < n_wave; win0 += kFeatRound) {
        const int lim = win0 + kFeatRound;
        uint8_t* const fdst = reinterpret_cast<uint8_t*>(P.features) + (base_out + win0) * 25;
        uint8_t* const rwin = win + record_shift(fdst);   // where this round's records go (flush_records)
        while (rest_lo && k < lim) {
            const int b = __builtin_ctz(rest_lo);
            rest_lo &= rest_lo - 1u;
            const uint32_t above = xb_lo & (~1u << b);
            uint32_t seg = (~0u << b) & valid_lo;
            if (above) seg &= (above & (0u - above)) - 1u;
            const FeatSums sum = feat_popc_half<0>(F, seg);
            if (above) {
                put_record(rwin, k - win0, sum);
            } else {                      // no boundary up to char 31: the token goes on in the high half (the last low token)
                S_str = sum;
                str_slot = k;
            }
            ++k;
        }
        if (__ballot(str_slot >= 0 && rest_lo == 0u)) {
            if (str_slot >= 0 && rest_lo == 0u) {
                uint32_t seg = valid_hi;
                if (xb_hi) seg &= (xb_hi & (0u - xb_hi)) - 1u;          // chars 32.. up to the first boundary there
                const FeatSums s2 = feat_popc_half<1>(F, seg);
#pragma unroll
                for (int j = 0; j < 7; ++j) S_str.v[j] = swar_add_u8(S_str.v[j], s2.v[j]);
                if (!xb_hi) {                                             // ... and on into the following words
#pragma unroll
                    for (int j = 0; j < 7; ++j) S_str.v[j] = swar_add_u8(S_str.v[j], C.v[j]);
                }
                put_record(rwin, str_slot - win0, S_str);
                str_slot = -1;
            }
        }
        while (rest_lo == 0u && rest_hi && k < lim) {
            const int b = __builtin_ctz(rest_hi);
            rest_hi &= rest_hi - 1u;
            const uint32_t above = xb_hi & (~1u << b);
            uint32_t seg = (~0u << b) & valid_hi;
            if (above) seg &= (above & (0u - above)) - 1u;
            FeatSums sum = feat_popc_half<1>(F, seg);
            if (!above) {
#pragma unroll
                for (int j = 0; j < 7; ++j) sum.v[j] = swar_add_u8(sum.v[j], C.v[j]);
            }
            put_record(rwin, k - win0, sum);
            ++k;
        }
        wave_lds_sync();
        flush_records(win, min(kFeatRound, n_wave - win0), fdst, lane);
        wave_lds_sync();
    }

    // ---- the span records of the same tokens: {raw start, raw end, stripped start, stripped end}, string relative --------
    // (reference featurize: LaToken.start_idx / end_idx = the raw span, .text = text[stripped]; default_tokenizer.py:173-191)
    OUT* swin = reinterpret_cast<OUT*>(L.stage);
    constexpr int kSpanRound = span_round<OUT>();
    lk_u64 rest = x;
    k = off;
    for (int win0 = 0; win0 < n_wave; win0 += kSpanRound) {
        while (rest && k < win0 + kSpanRound) {
            const int b = lk_ctz(rest);
            rest &= rest - 1;
            const int64_t p = base + b;
            const lk_u64 bl = B & ((2ull << b) - 1ull);         // string starts at or before the token (b = 63: all)
            const int64_t lo = bl ? base + 63 - __builtin_clzll(bl) : lo_in;
            const lk_u64 above = xb & (~1ull << b);
            int64_t e, a2, e2;
            if (above) {
                const int eb = lk_ctz(above);
                const lk_u64 seg = nn & (~0ull << b) & ((1ull << eb) - 1ull);
                e = base + eb;
                a2 = base + lk_ctz(seg);
                e2 = base + 64 - __builtin_clzll(seg);
            } else if (xb1) {
                const int eb = lk_ctz(xb1);
                const lk_u64 seg0 = nn & (~0ull << b);
                const lk_u64 seg1 = nn1 & ((1ull << eb) - 1ull);
                e = base + 64 + eb;
                a2 = seg0 ? base + lk_ctz(seg0) : base + 64 + lk_ctz(seg1);
                e2 = seg1 ? base + 128 - __builtin_clzll(seg1) : base + 64 - __builtin_clzll(seg0);
            } else {
                e = next_set_bit(P.bits, base + 64, total);
                const lk_u64 seg = nn & (~0ull << b);
                a2 = seg ? base + lk_ctz(seg) : next_zero_bit(P.space, base + 64, e);
                e2 = prev_zero_end(P.space, a2, e);
            }
            OUT* rec = swin + (k - win0) * 4;
            rec[0] = (OUT)(p - lo);
            rec[1] = (OUT)(e - lo);
            rec[2] = (OUT)(a2 - lo);
            rec[3] = (OUT)(e2 - lo);
            ++k;
        }
        wave_lds_sync();
        {
            // 16-byte stores: two (int64) or four (int32) values each; a record is 32 or 16 bytes, so the stream is aligned
            constexpr int kPer = 16 / (int)sizeof(OUT);
            typedef OUT vec_t __attribute__((ext_vector_type(16 / sizeof(OUT))));
            const int n_vec = min(kSpanRound, n_wave - win0) * 4 / kPer;
            vec_t* dst = reinterpret_cast<vec_t*>(spans4 + (base_out + win0) * 4);
            for (int i = lane; i < n_vec; i += 64) {
                vec_t v;
#pragma unroll
                for (int e = 0; e < kPer; ++e) v[e] = swin[kPer * i + e];
                __builtin_nontemporal_store(v, dst + i);
            }
        }
        wave_lds_sync();
    }
}

template <typename OUT>
__global__ __launch_bounds__(kFeatWaves * 64) void k_features_tiles(FeatParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFeatLdsTotal];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (the caller's buffers are too small: nothing is written)
    if (!(P.n_tokens_dev && *P.n_tokens_dev > P.cap)) {
        TileLds L;
        L.small_bits = L.small_space = nullptr;
        L.t1 = L.t2 = L.lut = L.ctab = L.ltab = L.t1b = L.t2b = nullptr; L.tables = nullptr; L.ctl = nullptr;   // nothing is classified here: the tile kernel left the rule codes (P.codes)
        uint8_t* mine = lds + wave * kFeatWaveLds;
        L.stage = mine;
        L.halo = mine + kFeatWinBytes;
        L.bw = reinterpret_cast<lk_u64*>(mine + kFeatWinBytes + 16);
        for (int64_t t = (int64_t)blockIdx.x * kFeatWaves + wave; t < P.n_tiles; t += (int64_t)gridDim.x * kFeatWaves)
            feature_tile<OUT>(P, L, t, lane);
    }
    signal_block_done(P.done);   // (pinned outputs of a small host batch: the host polls the completion word)
}

hipError_t launch_features_tiles(const FeatParams& P, int n_cu, hipStream_t st) {
    if (P.n_tiles <= 0) return hipSuccess;
    int64_t blocks = (P.n_tiles + kFeatWaves - 1) / kFeatWaves;
    if (blocks > n_cu) blocks = n_cu;
    if (P.out32) hipLaunchKernelGGL((k_features_tiles<int32_t>), dim3((unsigned)blocks), dim3(kFeatWaves * 64), 0, st, P);
    else hipLaunchKernelGGL((k_features_tiles<int64_t>), dim3((unsigned)blocks), dim3(kFeatWaves * 64), 0, st, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches in ONE launch (the drop-in tokenize(text) surface: one string per call).  A batch of at most one tile
// (4096 chars) needs no tile index, no resolve stage and no device-wide scan: a single wave classifies it, runs the
// tile function -- whose two provisional assumptions are exact here: nothing enters the batch's first tile, and the
// batch ends inside it or at its end --, counts, ranks and scatters.  What was six dependent launches (~45 us of launch
// latency for ~2 us of work) is one.  Inputs and outputs live in pinned host memory the kernel reads and writes over
// the bus (api.cpp), the class tables are read from global memory (they sit in L2; the 41 KB LDS copy would cost more
// than the few hundred lookups).
// ---------------------------------------------------------------------------------------------------------------
struct SmallParams {
    SplitParams P;          // cps, row_off, n_str, total (<= kTile), t1 / t2 (global memory), rules
    void* counts;           // OUT[n_str]
    void* items;            // KIND 0: OUT[n_items] offsets; KIND 1: OUT[n_items][2] stripped token spans
    int64_t* n_items;       // [1]
    unsigned long long* done;   // pinned host word that receives `seq` after every output has been stored (or NULL)
    unsigned long long seq;
    int8_t* features;       // KIND 2 (featurize): [n_items][25] sums; items = [n_items][4] {raw start, raw end, stripped start, stripped end}
};

template <int MODE, int KIND, typename OUT>
__global__ __launch_bounds__(64) void k_small_batch(SmallParams S) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWaveLdsBytes];
    __shared__ uint64_t s_bits[66], s_space[66], s_items[66];
    __shared__ int s_pref[66];
    const int lane = threadIdx.x;
    TileLds L;
    L.t1 = S.P.t1;          // global memory
    L.t2 = S.P.t2;
    L.lut = L.ctab = L.ltab = L.t1b = L.t2b = nullptr; L.tables = nullptr; L.ctl = nullptr;
    L.stage = lds;
    L.halo = lds + kStageBytes;
    L.bw = reinterpret_cast<lk_u64*>(lds + kStageBytes + 16);
    s_bits[lane] = 0ull;
    s_space[lane] = ~0ull;   // positions behind the batch read as SPACE
    if (lane < 2) { s_bits[64 + lane] = 0ull; s_space[64 + lane] = ~0ull; }
    const SplitParams& P = S.P;
    L.small_bits = s_bits;
    L.small_space = KIND >= 1 ? s_space : nullptr;
    wave_lds_sync();
    const int64_t total = P.total, n_str = P.n_str;
    const int64_t n_words = (total + 63) >> 6;
    // the bounds of the first 64 strings, requested now: they travel over the bus together with the tile's chars
    int64_t ro_a = 0, ro_b = 0;
    if (lane < n_str) { ro_a = P.row_off[lane]; ro_b = P.row_off[lane + 1]; }
    const lk_u64 xb = process_tile<MODE, false, true, true>(P, L, 0, 0, 0, -1, false, nullptr, lane);   // boundaries of my word
    wave_lds_sync();
    const int64_t base = 64 * (int64_t)lane;
    // ---- which boundaries are items (spans: those whose token holds a non-SPACE char), like k_word_counts -------------
    lk_u64 x = xb;
    const lk_u64 nn = KIND >= 1 ? (~s_space[lane] & valid_mask(lane, total)) : 0ull;
    lk_u64 xb1 = 0, nn1 = 0;
    if (KIND >= 1) {
        xb1 = __shfl_down(xb, 1);
        nn1 = __shfl_down(nn, 1);
        if (lane == 63) { xb1 = 0; nn1 = 0; }
        bool cin;
        if (xb1) cin = (nn1 & ((xb1 & (~xb1 + 1ull)) - 1ull)) != 0;
        else cin = nn1 != 0 || (xb != 0 && lane + 1 < n_words && tail_has_nonspace(s_bits, s_space, lane + 1, n_words, total));
        const lk_u64 xr = __builtin_bitreverse64(xb), nr = __builtin_bitreverse64(nn);
        const lk_u64 g = nr & ~xr, pr = ~xr;
        const lk_u64 a = pr | g;
        const lk_u64 carries = (a + g + (cin ? 1ull : 0ull)) ^ a ^ g;
        x = __builtin_bitreverse64(xr & (nr | carries));
    }
    const int cnt = lk_popc(x);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    const int n_items = __shfl(inc, 63);
    s_items[lane] = x;
    s_pref[lane] = inc - cnt;
    if (lane == 0) { s_items[64] = 0ull; s_pref[64] = n_items; *S.n_items = n_items; }
    wave_lds_sync();
    // ---- per-string counts: rank(end) - rank(start) ----------------------------------------------------------------
    OUT* counts = reinterpret_cast<OUT*>(S.counts);
    auto rank_of = [&](int64_t p) -> int {
        if (p >= total) return n_items;
        return s_pref[p >> 6] + lk_popc(s_items[p >> 6] & low_mask((int)(p & 63)));
    };
    if (lane < n_str) counts[lane] = (OUT)(rank_of(ro_b) - rank_of(ro_a));
    for (int64_t s = 64 + lane; s < n_str; s += 64) counts[s] = (OUT)(rank_of(P.row_off[s + 1]) - rank_of(P.row_off[s]));
    // ---- where the string that owns a position begins: last string start at or before it (L.bw: the tile's string starts)
    const lk_u64 Bw = L.bw[lane];
    int carry = Bw ? 64 * lane + 63 - __builtin_clzll(Bw) : -1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(carry, d);
        if (lane >= d && o > carry) carry = o;
    }
    carry = __shfl_up(carry, 1);
    if (lane == 0) carry = -1;
    const int64_t lo_in = carry >= 0 ? carry : 0;
    if (KIND == 2) {
        // ---- featurize: the 25 feature planes of every word go to LDS, then lane = TOKEN: its span from the bitmasks, its
        //      sums = popcounts of the planes over the span.  (k_features_tiles is built for throughput -- ~7 K dependent
        //      instructions per tile, 26 us when a single tile is all there is; this is ~1 K.)
        __shared__ lk_u64 s_planes[64 * LK_N_FEATURES];
        __shared__ int s_lo[64];
        __shared__ __attribute__((aligned(16))) uint8_t s_win[64 * kFeatRec + 16];
        {
            uint32_t d[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 q = *reinterpret_cast<const uint4*>(L.stage + 80u * lane + 16u * k);
                d[4 * k + 0] = q.x; d[4 * k + 1] = q.y; d[4 * k + 2] = q.z; d[4 * k + 3] = q.w;
            }
            lk_halo h;
            h.prev = lane > 0 ? L.stage[80u * lane - 17u] : L.halo[0];
            h.next0 = lane < 63 ? L.stage[80u * lane + 80u] : L.halo[1];
            h.next1 = lane < 63 ? L.stage[80u * lane + 81u] : L.halo[2];
            lk_u64 plane[8];
            lk_bitslice64(d, plane);
            lk_planes F;
            lk_feature_planes(plane, h, Bw, L.bw[lane + 1] & 3ull, F);
#pragma unroll
            for (int c = 0; c < LK_N_FEATURES; ++c) s_planes[lane * LK_N_FEATURES + c] = LK_PLANE_GET(F, c);
        }
        s_lo[lane] = (int)lo_in;
        wave_lds_sync();
        OUT* spans4 = reinterpret_cast<OUT*>(S.items);
        for (int k0 = 0; k0 < n_items; k0 += 64) {
            const int k = k0 + lane;
            FeatSums sum;
#pragma unroll
            for (int j = 0; j < 7; ++j) sum.v[j] = 0;
            if (k < n_items) {
                // the word that holds kept token k: the last w with s_pref[w] <= k (s_pref[64] = n_items)
                int wl = 0, wh = 64;
                while (wh - wl > 1) {
                    const int mid = (wl + wh) >> 1;
                    if (s_pref[mid] <= k) wl = mid; else wh = mid;
                }
                lk_u64 m = s_items[wl];
                for (int r = k - s_pref[wl]; r > 0; --r) m &= m - 1ull;
                const int b = lk_ctz(m);
                const int64_t a = 64 * (int64_t)wl + b;                       // raw start
                const int64_t e = next_set_bit(s_bits, a + 1, total);         // raw end: the next boundary
                const int64_t a2 = next_zero_bit(s_space, a, e);              // stripped span (a kept token has a non-SPACE char)
                const int64_t e2 = prev_zero_end(s_space, a2, e);
                const lk_u64 bl = L.bw[wl] & ((2ull << b) - 1ull);            // string starts at or before the token
                const int64_t lo = bl ? 64 * (int64_t)wl + 63 - __builtin_clzll(bl) : (int64_t)s_lo[wl];
                OUT* rec = spans4 + 4 * (int64_t)k;
                rec[0] = (OUT)(a - lo); rec[1] = (OUT)(e - lo); rec[2] = (OUT)(a2 - lo); rec[3] = (OUT)(e2 - lo);
                uint32_t acc[LK_N_FEATURES];
#pragma unroll
                for (int c = 0; c < LK_N_FEATURES; ++c) acc[c] = 0;
                for (int64_t w = wl; 64 * w < e; ++w) {
                    lk_u64 msk = ~0ull;
                    if (w == wl) msk &= ~0ull << b;
                    if (e < 64 * w + 64) msk &= (1ull << (e - 64 * w)) - 1ull;
#pragma unroll
                    for (int c = 0; c < LK_N_FEATURES; ++c) acc[c] += (uint32_t)__popcll(s_planes[w * LK_N_FEATURES + c] & msk);
                }
#pragma unroll
                for (int c = 0; c < LK_N_FEATURES; ++c) sum.v[c >> 2] |= (acc[c] & 0xFFu) << (8 * (c & 3));   // uint8 wrap-around (latok.c:342-354)
            }
            const int n_here = min(64, n_items - k0);
            uint8_t* const fdst = reinterpret_cast<uint8_t*>(S.features) + (int64_t)k0 * kFeatRec;
            if (k < n_items) put_record(s_win + record_shift(fdst), lane, sum);
            wave_lds_sync();
            flush_records(s_win, n_here, fdst, lane);
            wave_lds_sync();
        }
    }
    // ---- the records, word-major: lane = word walks its items -------------------------------------------------------
    OUT* out = reinterpret_cast<OUT*>(S.items);
    lk_u64 rest = KIND == 2 ? 0ull : x;
    int k = inc - cnt;
    while (rest) {
        const int b = lk_ctz(rest);
        rest &= rest - 1;
        const lk_u64 bl = Bw & ((2ull << b) - 1ull);      // string starts at or before the item (b = 63: all)
        const int64_t lo = bl ? base + 63 - __builtin_clzll(bl) : lo_in;
        if (KIND == 0) {
            out[k] = (OUT)(base + b - lo);
        } else {
            const lk_u64 above = xb & (~1ull << b);
            int64_t a2, e2;
            if (above) {
                const int eb = lk_ctz(above);
                const lk_u64 seg = nn & (~0ull << b) & ((1ull << eb) - 1ull);
                a2 = base + lk_ctz(seg);
                e2 = base + 64 - __builtin_clzll(seg);
            } else if (xb1) {
                const int eb = lk_ctz(xb1);
                const lk_u64 seg0 = nn & (~0ull << b);
                const lk_u64 seg1 = nn1 & ((1ull << eb) - 1ull);
                a2 = seg0 ? base + lk_ctz(seg0) : base + 64 + lk_ctz(seg1);
                e2 = seg1 ? base + 128 - __builtin_clzll(seg1) : base + 64 - __builtin_clzll(seg0);
            } else {
                const int64_t e = next_set_bit(s_bits, base + 64, total);
                const lk_u64 seg = nn & (~0ull << b);
                a2 = seg ? base + lk_ctz(seg) : next_zero_bit(s_space, base + 64, e);
                e2 = prev_zero_end(s_space, a2, e);
            }
            out[2 * k] = (OUT)(a2 - lo);
            out[2 * k + 1] = (OUT)(e2 - lo);
        }
        ++k;
    }
    // completion word: the host polls it instead of waiting for the end-of-kernel signal to travel through the runtime.
    // One wave wrote everything, its stores leave in order, the fence drains them to system scope before the word follows.
    if (S.done) {
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(S.done, S.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_small_batch(const SplitParams& P, bool rules, int kind, bool out32, void* counts, void* items, int8_t* features,
                              int64_t* n_items, unsigned long long* done, unsigned long long seq, hipStream_t st) {
    SmallParams S;
    S.P = P;
    S.counts = counts;
    S.items = items;
    S.features = features;
    S.n_items = n_items;
    S.done = done;
    S.seq = seq;
#define LATOK_SB(M, K, T) hipLaunchKernelGGL((k_small_batch<M, K, T>), dim3(1), dim3(64), 0, st, S)
#define LATOK_SB_K(M, T) do { if (kind == 0) LATOK_SB(M, 0, T); else if (kind == 1) LATOK_SB(M, 1, T); else LATOK_SB(M, 2, T); } while (0)
    if (rules) {
        if (out32) LATOK_SB_K(kModeRules, int32_t); else LATOK_SB_K(kModeRules, int64_t);
    } else {
        if (out32) LATOK_SB_K(kModeBits, int32_t); else LATOK_SB_K(kModeBits, int64_t);
    }
#undef LATOK_SB_K
#undef LATOK_SB
    return hipGetLastError();
}

// _gen_block_mask (latok.c:150-270) of ONE small array pair: the tile function in block-mask mode on a single tile, the
// {any(a1), any(a2)} flags of the reference's element-0 quirk computed by the same wave, the completion word at the end.
__global__ __launch_bounds__(64) void k_small_block_mask(SmallParams S) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWaveLdsBytes];
    __shared__ int s_flags[2];
    const int lane = threadIdx.x;
    TileLds L;
    L.t1 = L.t2 = L.lut = L.ctab = L.ltab = L.t1b = L.t2b = nullptr; L.tables = nullptr; L.ctl = nullptr;
    L.small_bits = L.small_space = nullptr;
    L.stage = lds;
    L.halo = lds + kStageBytes;
    L.bw = reinterpret_cast<lk_u64*>(lds + kStageBytes + 16);
    SplitParams P = S.P;
    const int64_t n = P.total;
    uint32_t f1 = 0, f2 = 0;
    for (int64_t i = 4 * (int64_t)lane; i < n; i += 256) {
        if (i + 4 <= n) {
            f1 |= *reinterpret_cast<const uint32_t*>(P.bm_a1 + i);
            f2 |= *reinterpret_cast<const uint32_t*>(P.bm_a2 + i);
        } else {
            for (int64_t j = i; j < n; ++j) { f1 |= (uint8_t)P.bm_a1[j]; f2 |= (uint8_t)P.bm_a2[j]; }
        }
    }
    const int any1 = __any(f1 != 0u), any2 = __any(f2 != 0u);
    if (lane == 0) { s_flags[0] = any1; s_flags[1] = any2; }
    wave_lds_sync();
    P.bm_flags = s_flags;
    process_tile<kModeBlockMask, false, true, true>(P, L, 0, 0, 0, -1, false, nullptr, lane);
    if (S.done) {
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(S.done, S.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_small_block_mask(const SplitParams& P, unsigned long long* done, unsigned long long seq, hipStream_t st) {
    SmallParams S;
    S.P = P;
    S.counts = S.items = nullptr;
    S.features = nullptr;
    S.n_items = nullptr;
    S.done = done;
    S.seq = seq;
    hipLaunchKernelGGL(k_small_block_mask, dim3(1), dim3(64), 0, st, S);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// stage 0: the per-tile string index.  tile_first[t] = first s in [0, n_str] with row_off[s] >= t * kTile (row_off[n_str]
// = total closes the list).  One thread per entry: entry s is the first string of every tile that begins in
// (row_off[s-1], row_off[s]].  Entries that cover many tiles (a long document, the tail of the batch) are written by
// the whole wave.  ~3 us for 1 M strings; inside k_tiles_main the same work was a ~9 us serial prologue of every segment
// (tables -> row_off window -> barriers -> first tile: three dependent trips to memory before the stream started).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_index(const int64_t* __restrict__ row_off, int64_t n_str, int64_t n_tiles,
                                                   int64_t* __restrict__ tile_first) {
    tile_index_entry(row_off, n_str, n_tiles, tile_first, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, threadIdx.x & 63);
}

hipError_t launch_tile_index(const SplitParams& P, hipStream_t st) {
    if (P.n_tiles <= 0) return hipSuccess;
    const int64_t entries = P.n_str + 1;
    hipLaunchKernelGGL(k_tile_index, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, P.row_off, P.n_str, P.n_tiles,
                       P.tile_first);
    return hipGetLastError();
}

void plan_segments(int64_t n_tiles, int n_cu, int* seg_tiles, int64_t* n_segs) {
    // every workgroup gets the same number of (almost) equally sized segments: `rounds` segments of
    // ceil(n_tiles / (n_cu * rounds)) tiles, rounds = smallest count that keeps a segment within kSegMax tiles
    const int64_t per_cu = (n_tiles + n_cu - 1) / n_cu;
    const int64_t rounds = per_cu > kSegMax ? (per_cu + kSegMax - 1) / kSegMax : 1;
    int64_t s = (n_tiles + (int64_t)n_cu * rounds - 1) / ((int64_t)n_cu * rounds);
    // (A segment's tiles go round-robin over the workgroup's kWPB waves, so a length that is not a multiple of kWPB ends in a
    // round with most waves idle.  Rounding long segments to whole rounds was measured for the blocking calls and not kept;
    // the flow picks, among three CU shares, one whose last round is at least half full: api.cpp run_pipeline.)
    if (s < kWPB) s = kWPB;
    if (s > kSegMax) s = kSegMax;
    *seg_tiles = (int)s;
    *n_segs = (n_tiles + s - 1) / s;
}

constexpr int64_t kFastTailTiles = 256;   // ~1 M chars: beyond that one tile's latency is noise
static inline int grid_for(const SplitParams& P, int n_cu) {
    return (int)(P.n_segs < n_cu ? (P.n_segs < 1 ? 1 : P.n_segs) : n_cu);
}

hipError_t launch_split_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st, bool in_flow) {
    const dim3 grid(grid_for(P, n_cu)), block(kWPB * 64);
    const bool fast_tail = P.n_tiles <= kFastTailTiles;
    if (mode == kModeBits && in_flow && !fast_tail) hipLaunchKernelGGL((k_tiles_main<kModeBits, false, kCpsPrefetchRowsFlow>), grid, block, 0, st, P);
    else if (mode == kModeBits && fast_tail) hipLaunchKernelGGL((k_tiles_main<kModeBits, true>), grid, block, 0, st, P);
    else if (mode == kModeRules && fast_tail) hipLaunchKernelGGL((k_tiles_main<kModeRules, true>), grid, block, 0, st, P);
    else if (mode == kModeBits) hipLaunchKernelGGL((k_tiles_main<kModeBits>), grid, block, 0, st, P);
    else if (mode == kModeValues) hipLaunchKernelGGL((k_tiles_main<kModeValues>), grid, block, 0, st, P);
    else if (mode == kModeRules) hipLaunchKernelGGL((k_tiles_main<kModeRules>), grid, block, 0, st, P);
    else if (mode == kModeBytes) hipLaunchKernelGGL((k_tiles_main<kModeBytes>), grid, dim3(tile_wpb(kModeBytes) * 64), 0, st, P);
    else if (mode == kModeLatin1) hipLaunchKernelGGL((k_tiles_main<kModeLatin1>), grid, dim3(tile_wpb(kModeLatin1) * 64), 0, st, P);
    else if (mode == kModeUcs2) hipLaunchKernelGGL((k_tiles_main<kModeUcs2>), grid, dim3(tile_wpb(kModeUcs2) * 64), 0, st, P);
    else if (mode == kModeBytesRules) hipLaunchKernelGGL((k_tiles_main<kModeBytesRules>), grid, block, 0, st, P);
    else if (mode == kModeLatin1Rules) hipLaunchKernelGGL((k_tiles_main<kModeLatin1Rules>), grid, block, 0, st, P);
    else if (mode == kModeUcs2Rules) hipLaunchKernelGGL((k_tiles_main<kModeUcs2Rules>), grid, block, 0, st, P);
    else if (mode == kModeValuesRules) hipLaunchKernelGGL((k_tiles_main<kModeValuesRules>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((k_tiles_main<kModeBlockMask>), grid, block, 0, st, P);
    return hipGetLastError();
}

hipError_t launch_one_segment(const SplitParams& P, int mode, hipStream_t st) {
    if (P.n_segs != 1 || P.seg_tiles < P.n_tiles || P.n_tiles > kOneSegTiles || (mode != kModeBits && mode != kModeRules))
        return hipErrorInvalidValue;
    if (mode == kModeBits) hipLaunchKernelGGL((k_one_segment<kModeBits>), dim3(1), dim3(kWPB * 64), 0, st, P);
    else hipLaunchKernelGGL((k_one_segment<kModeRules>), dim3(1), dim3(kWPB * 64), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_resolve_fix(const SplitParams& P, int mode, int n_cu, hipStream_t st) {
    const dim3 grid(grid_for(P, n_cu)), block(kWPB * 64);
    // The bitmask mode repairs almost everything in place, so its resolve stage is pure latency and runs with as few
    // waves as cover a segment; the modes that recompute tiles keep all 12 waves for that.
    if (mode == kModeBits) {
        if (P.seg_tiles <= 128) hipLaunchKernelGGL((k_resolve_fix<kModeBits, 2>), grid, dim3(128), 0, st, P);
        else if (P.seg_tiles <= 256) hipLaunchKernelGGL((k_resolve_fix<kModeBits, 4>), grid, dim3(256), 0, st, P);
        else hipLaunchKernelGGL((k_resolve_fix<kModeBits, kWPB>), grid, block, 0, st, P);
    } else if (mode == kModeValues) hipLaunchKernelGGL((k_resolve_fix<kModeValues, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeRules) hipLaunchKernelGGL((k_resolve_fix<kModeRules, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeBytesRules) hipLaunchKernelGGL((k_resolve_fix<kModeBytesRules, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeLatin1Rules) hipLaunchKernelGGL((k_resolve_fix<kModeLatin1Rules, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeUcs2Rules) hipLaunchKernelGGL((k_resolve_fix<kModeUcs2Rules, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeValuesRules) hipLaunchKernelGGL((k_resolve_fix<kModeValuesRules, kWPB>), grid, block, 0, st, P);
    else if (mode == kModeBytes) {   // (ASCII tiles are repaired in place here too; measured better with few waves on C2 and C3)
        if (P.seg_tiles <= 128) hipLaunchKernelGGL((k_resolve_fix<kModeBytes, 2>), grid, dim3(128), 0, st, P);
        else if (P.seg_tiles <= 256) hipLaunchKernelGGL((k_resolve_fix<kModeBytes, 4>), grid, dim3(256), 0, st, P);
        else hipLaunchKernelGGL((k_resolve_fix<kModeBytes, kWPB>), grid, block, 0, st, P);
    } else if (mode == kModeLatin1) {
        if (P.seg_tiles <= 128) hipLaunchKernelGGL((k_resolve_fix<kModeLatin1, 2>), grid, dim3(128), 0, st, P);
        else if (P.seg_tiles <= 256) hipLaunchKernelGGL((k_resolve_fix<kModeLatin1, 4>), grid, dim3(256), 0, st, P);
        else hipLaunchKernelGGL((k_resolve_fix<kModeLatin1, kWPB>), grid, block, 0, st, P);
    } else if (mode == kModeUcs2) {
        if (P.seg_tiles <= 128) hipLaunchKernelGGL((k_resolve_fix<kModeUcs2, 2>), grid, dim3(128), 0, st, P);
        else if (P.seg_tiles <= 256) hipLaunchKernelGGL((k_resolve_fix<kModeUcs2, 4>), grid, dim3(256), 0, st, P);
        else hipLaunchKernelGGL((k_resolve_fix<kModeUcs2, kWPB>), grid, block, 0, st, P);
    } else hipLaunchKernelGGL((k_resolve_fix<kModeBlockMask, kWPB>), grid, block, 0, st, P);
    return hipGetLastError();
}

#ifdef LATOK_STAMPS
extern "C" int latok_diag_stamps(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_sum), 16 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), z, sizeof(z));
    }
    return (int)e;
}
#endif

}  // namespace latok
