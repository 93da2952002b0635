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
// publishes a 16-byte summary; k_scan_summaries resolves the two unknowns per tile exactly and lists the (few) tiles
// whose assumption was wrong; those are recomputed by the same tile code with the exact inputs.
//
// No MFMA: this is integer/bit work bounded by HBM reads (4 B/char), not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "lane_math.h"

namespace latok {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t stage_addr(uint32_t p) { return p + ((p >> 6) << 4); }  // 64-byte rows, 16 B pad

__device__ __forceinline__ uint32_t classify1(const uint8_t* t1, const uint8_t* t2, uint32_t cp) {
    const uint32_t hi = min(cp >> kTblShift, (uint32_t)(kStage1Len - 1));
    const uint32_t blk = t1[hi];
    return t2[(blk << kTblShift) | (cp & ((1u << kTblShift) - 1u))];
}

__device__ __forceinline__ uint32_t classify4(const uint8_t* t1, const uint8_t* t2, u32x4 v) {
    uint32_t c;
    // wave-uniform fast path: all 256 chars of this wave instruction are ASCII -> stage-2 block 0, no stage-1 lookup
    if (__all((v.x | v.y | v.z | v.w) < 128u)) {
        c = (uint32_t)t2[v.x] | ((uint32_t)t2[v.y] << 8) | ((uint32_t)t2[v.z] << 16) | ((uint32_t)t2[v.w] << 24);
    } else {
        c = classify1(t1, t2, v.x) | (classify1(t1, t2, v.y) << 8) | (classify1(t1, t2, v.z) << 16) |
            (classify1(t1, t2, v.w) << 24);
    }
    return c;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS traffic of one wave is executed in issue order; this only stops the compiler from reordering across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
// stage 0: tile_first[t] = index of the first string whose start offset is >= t * kTile
// (one thread per row_off entry, including the end sentinel row_off[n_str])
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_tile_index(const int64_t* __restrict__ row_off, int64_t n_str, int64_t n_tiles,
                             int64_t* __restrict__ tile_first, int64_t* __restrict__ fix_count) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) *fix_count = 0;   // consumed two launches later by the summary scan
    if (s > n_str) return;
    const int64_t cur = row_off[s] / kTile;
    const int64_t prev = s > 0 ? row_off[s - 1] / kTile : -1;
    for (int64_t w = prev + 1; w <= cur && w < n_tiles; ++w) tile_first[w] = s;
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
};

// PF: software prefetch.  `v` carries the 16 KiB of code points of a full tile in registers; when `v_valid` the loads
// for THIS tile were issued while the previous tile was in phase 2.  After classifying, the loads of tile `t_next` are
// issued so that they fly under this tile's phase 2.  Returns whether `v` now holds tile `t_next`.
// VMODE 0: the tile's loads are issued here.  1: `v` was loaded by the caller (first tile of a wave, requested before
// the table copy).  2 (= PF): runtime `v_valid`, and the next tile's loads are issued after classification.
template <int MODE, int VMODE>
__device__ __forceinline__ bool process_tile(const SplitParams& P, const TileLds& L, int64_t t, int q_in, int tail_zero,
                                             bool write_summary, int lane, u32x4 (&v)[16], bool v_valid,
                                             int64_t t_next) {
    const int64_t t0 = t * kTile;
    const int64_t total = P.total;
    bool next_valid = false;

    // small loads first, so that their latency flies together with the 16 KiB of code points: the first string that
    // starts in this tile, the start offsets of the next 64 strings, and the three halo characters
    int64_t idx0 = P.tile_first[t];
    int64_t ro = idx0 + lane <= P.n_str ? P.row_off[idx0 + lane] : INT64_MAX;
    uint32_t halo_cp = 0xFFFFFFFFu;   // out of range -> class 0
    if (MODE != kModeBlockMask && lane < 3) {
        const int64_t hp = lane == 0 ? t0 - 1 : t0 + kTile + (lane - 1);
        if (hp >= 0 && hp < total) halo_cp = P.cps[hp];
    }

    // ---- phase 1: classify 4096 chars, 4 per lane per step, into the staging buffer --------------------------
    if (MODE == kModeBlockMask) {
        // planes come straight from the caller's byte arrays (compat _gen_block_mask): nothing to classify
    } else if (t0 + kTile <= total) {
        if (VMODE == 3) {
            // half prefetch: v[0..7] (first 8 KiB) may already be here; the second half is requested now and is
            // classified after the first
            const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t0) + lane;
            u32x4 w[8];
            if (!v_valid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = __builtin_nontemporal_load(src + 64 * (8 + i));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t c = classify4(L.t1, L.t2, v[i]);
                *reinterpret_cast<uint32_t*>(L.stage + stage_addr(256u * i + 4u * lane)) = c;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t c = classify4(L.t1, L.t2, w[i]);
                *reinterpret_cast<uint32_t*>(L.stage + stage_addr(256u * (8 + i) + 4u * lane)) = c;
            }
            if (t_next >= 0 && (t_next + 1) * kTile <= total) {
                const u32x4* nsrc = reinterpret_cast<const u32x4*>(P.cps + t_next * kTile) + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(nsrc + 64 * i);
                next_valid = true;
            }
        } else {
        if (VMODE == 0 || (VMODE == 2 && !v_valid)) {
            const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t0) + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = classify4(L.t1, L.t2, v[i]);
            *reinterpret_cast<uint32_t*>(L.stage + stage_addr(256u * i + 4u * lane)) = c;
        }
        if (VMODE == 2 && t_next >= 0 && (t_next + 1) * kTile <= total) {
            const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t_next * kTile) + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
            next_valid = true;
        }
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int64_t p = t0 + 256 * i + 4 * lane;
            u32x4 v;
            v.x = p + 0 < total ? P.cps[p + 0] : 0xFFFFFFFFu;   // out of range -> class 0 ("nothing")
            v.y = p + 1 < total ? P.cps[p + 1] : 0xFFFFFFFFu;
            v.z = p + 2 < total ? P.cps[p + 2] : 0xFFFFFFFFu;
            v.w = p + 3 < total ? P.cps[p + 3] : 0xFFFFFFFFu;
            const uint32_t c = classify4(L.t1, L.t2, v);
            *reinterpret_cast<uint32_t*>(L.stage + stage_addr(256u * i + 4u * lane)) = c;
        }
    }
    // halo chars t0-1, t0+4096, t0+4097 (lanes 0..2) and the string-start words
    if (MODE != kModeBlockMask && lane < 3) L.halo[lane] = (uint8_t)classify1(L.t1, L.t2, halo_cp);
    L.bw[lane] = 0;
    if (lane == 0) L.bw[64] = 0;
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

    // ---- phase 2: lane = one 64-char word ---------------------------------------------------------------------
    const lk_u64 B = L.bw[lane];
    const int64_t base = t0 + 64 * (int64_t)lane;
    lk_local loc;
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
            const uint4 q = *reinterpret_cast<const uint4*>(L.stage + stage_addr(64u * lane + 16u * k));
            d[4 * k + 0] = q.x; d[4 * k + 1] = q.y; d[4 * k + 2] = q.z; d[4 * k + 3] = q.w;
        }
        lk_halo h;
        h.prev = lane > 0 ? L.stage[stage_addr(64u * lane - 1u)] : L.halo[0];
        h.next0 = lane < 63 ? L.stage[stage_addr(64u * lane + 64u)] : L.halo[1];
        h.next1 = lane < 63 ? L.stage[stage_addr(64u * lane + 65u)] : L.halo[2];
        const lk_u64 Bn = L.bw[lane + 1] & 3ull;
        lk_u64 plane[8];
        lk_bitslice64(d, plane);
        const lk_feat f = lk_decode(plane);
        loc = lk_rules(f, h, B, Bn);
    }
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
        const int hs_pos = c_rel > 0 ? c_rel - 1 : 0;
        const int head_sym = c_rel > 0 ? lane_read((int)((loc.sym >> (hs_pos & 63)) & 1ull), hs_pos >> 6) : 0;
        const int tail_sym = lane_read((int)(loc.sym >> 63), 63);
        if (lane == 0) {
            const int geom = (closing_lanes != 0) | (c_rel << 1) | (p_rel << 14) | (head_sym << 27) | (tail_keep << 28) |
                             (tail_sym << 29);
            P.summ[t] = make_int4(tile_fn.a, tile_fn.b, head, geom);
        }
    }

    // backward: zeroing closings clear the block below them; the carry chain over lanes is one 64-bit add on ballots
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

    if (base < total) {
        const int64_t remain = total - base;
        const lk_u64 valid = remain >= 64 ? ~0ull : ((1ull << remain) - 1ull);
        const lk_u64 keep = ~cleared;
        if (MODE == kModeBits) {
            P.bits_out[base >> 6] = ((loc.raw & keep) | loc.sym | B) & valid;
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
                        int v = (int)((loc.t_space >> i) & 1) + (int)((loc.t_sym >> i) & 1) +
                                (int)((loc.t_prevsym >> i) & 1) + (int)((loc.t_camel_next >> i) & 1) +
                                (int)((loc.t_camel_prev >> i) & 1);
                        v = ((keep >> i) & 1) ? v : 0;
                        v += (int)((loc.sym >> i) & 1);
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
    wave_lds_sync();  // staging buffer is reused by this wave's next tile
    return next_valid;
}

// ---------------------------------------------------------------------------------------------------------------
// stage 1 / stage 3 kernel.  FIX = false: all tiles, grid-stride, q_in = 0, provisional tail, writes summaries.
//                            FIX = true : only the tiles listed by k_scan_summaries, with their exact inputs.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool FIX, int WPB, int PREFETCH>
__global__ __launch_bounds__(WPB * 64) void k_split_tiles(SplitParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kTablesLdsBytes + WPB * kWaveLdsBytes];

    int64_t n_items = P.n_tiles;
    if (FIX) {
        n_items = *P.fix_count;
        if (n_items == 0) return;  // uniform: nothing to repair, skip the table load
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> scalar tile arithmetic
    const int64_t wave_gid = (int64_t)blockIdx.x * WPB + wave;
    const int64_t n_waves = (int64_t)gridDim.x * WPB;

    // the first tile's 16 KiB of code points are requested before anything else, so that HBM latency overlaps the
    // table copy and the barrier below
    u32x4 v[16];
    bool v_valid = false;
    if (!FIX && MODE != kModeBlockMask && wave_gid < n_items && (wave_gid + 1) * kTile <= P.total) {
        const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + wave_gid * kTile) + lane;
#pragma unroll
        for (int i = 0; i < (PREFETCH == 3 ? 8 : 16); ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
        v_valid = true;
    }
    // cooperative table load (global/L2 -> LDS), 16 B per thread per step
    if (MODE != kModeBlockMask) {
        const uint4* s1 = reinterpret_cast<const uint4*>(P.t1);
        uint4* d1 = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < kStage1Pad / 16; i += WPB * 64) d1[i] = s1[i];
        const uint4* s2 = reinterpret_cast<const uint4*>(P.t2);
        uint4* d2 = reinterpret_cast<uint4*>(lds + kStage1Pad);
        for (int i = threadIdx.x; i < kStage2Len / 16; i += WPB * 64) d2[i] = s2[i];
    }
    __syncthreads();

    TileLds L;
    L.t1 = lds;
    L.t2 = lds + kStage1Pad;
    uint8_t* mine = lds + kTablesLdsBytes + wave * kWaveLdsBytes;
    L.stage = mine;
    L.halo = mine + kStageBytes;
    L.bw = reinterpret_cast<lk_u64*>(mine + kStageBytes + 16);

    int64_t i = wave_gid;
    if (FIX) {
        for (; i < n_items; i += n_waves)
            process_tile<MODE, 0>(P, L, P.fix_list[i], P.fix_q[i], P.fix_tz[i], false, lane, v, false, -1);
    } else if (PREFETCH != 0) {
        for (; i < n_items; i += n_waves) {
            const int64_t nxt = i + n_waves < n_items ? i + n_waves : -1;
            v_valid = process_tile<MODE, PREFETCH>(P, L, i, 0, -1, true, lane, v, v_valid, nxt);
        }
    } else {
        if (v_valid) {   // first tile: its loads were requested before the table copy
            process_tile<MODE, 1>(P, L, i, 0, -1, true, lane, v, true, -1);
            i += n_waves;
        }
        for (; i < n_items; i += n_waves) process_tile<MODE, 0>(P, L, i, 0, -1, true, lane, v, false, -1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stage 2: resolve, for every tile, (a) the number of pending starts entering it (forward scan of the tile transfer
// functions) and (b) whether the block that is open at its end gets zeroed (needs the starts that follow before the
// next closing event: backward scan).  Tiles whose provisional assumptions (q_in == 0, tail = "pending at end") do
// not hold are appended to the fix list.  One workgroup; each thread owns a contiguous chunk of tiles.
// ---------------------------------------------------------------------------------------------------------------
struct Fn64 {
    long long a, b;  // f(q) = max(q + a, b); a <= kNegInf64 means constant b
};
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
struct Hd64 {
    long long h;
    int c;
};
__device__ __forceinline__ Hd64 hd_then(Hd64 x, Hd64 y) {
    Hd64 r;
    r.c = x.c | y.c;
    r.h = x.c ? x.h : x.h + y.h;
    return r;
}

constexpr int kScanThreads = 1024;   // tiles per scan block
constexpr int kScanWaves = kScanThreads / 64;

__device__ __forceinline__ Fn64 fn_identity() { Fn64 f; f.a = 0; f.b = 0; return f; }   // identity on q >= 0
__device__ __forceinline__ Hd64 hd_identity() { Hd64 h; h.h = 0; h.c = 0; return h; }

// Ordered block-wide scans over kScanThreads elements (one per thread).  Returns, for this thread, the composition of
// all EARLIER elements (fn: exclusive prefix) and of all LATER elements (hd: exclusive suffix); *tot_* get the
// composition of the whole block.  Wave-level shuffles + 16 wave aggregates in LDS.
struct ScanLds {
    Fn64 fn_w[kScanWaves];
    Hd64 hd_w[kScanWaves];
};
__device__ __forceinline__ void block_scan(Fn64 f, Hd64 h, ScanLds& L, Fn64* excl_fn, Hd64* excl_hd, Fn64* tot_fn,
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
    Fn64 wf = lane < kScanWaves ? L.fn_w[lane] : fn_identity();
    Hd64 wh = lane < kScanWaves ? L.hd_w[lane] : hd_identity();
#pragma unroll
    for (int d = 1; d < kScanWaves; d <<= 1) {
        Fn64 o; o.a = __shfl_up(wf.a, d); o.b = __shfl_up(wf.b, d);
        if (lane >= d) wf = fn_then(o, wf);
        Hd64 oh; oh.h = __shfl_down(wh.h, d); oh.c = __shfl_down(wh.c, d);
        if (lane + d < kScanWaves) wh = hd_then(wh, oh);
    }
    // inclusive prefix of waves 0..lane in wf, inclusive suffix of waves lane..15 in wh
    Fn64 before; before.a = __shfl(wf.a, wave > 0 ? wave - 1 : 0); before.b = __shfl(wf.b, wave > 0 ? wave - 1 : 0);
    if (wave == 0) before = fn_identity();
    Hd64 after; after.h = __shfl(wh.h, wave < kScanWaves - 1 ? wave + 1 : 0); after.c = __shfl(wh.c, wave < kScanWaves - 1 ? wave + 1 : 0);
    if (wave == kScanWaves - 1) after = hd_identity();
    Fn64 all_f; all_f.a = __shfl(wf.a, kScanWaves - 1); all_f.b = __shfl(wf.b, kScanWaves - 1);
    Hd64 all_h; all_h.h = __shfl(wh.h, 0); all_h.c = __shfl(wh.c, 0);
    *excl_fn = fn_then(before, ef);
    *excl_hd = hd_then(eh, after);
    *tot_fn = all_f;
    *tot_hd = all_h;
}

// stage 2a: one block per 1024 tiles -> block aggregates (transfer function of the block, head descriptor of the block)
__global__ __launch_bounds__(kScanThreads) void k_scan_aggregate(const int4* __restrict__ summ, int64_t n_tiles,
                                                                 Fn64* __restrict__ agg_fn, Hd64* __restrict__ agg_hd) {
    __shared__ ScanLds L;
    const int64_t t = (int64_t)blockIdx.x * kScanThreads + threadIdx.x;
    Fn64 f = fn_identity();
    Hd64 h = hd_identity();
    if (t < n_tiles) {
        const int4 s = summ[t];
        f = fn_of(s);
        h.h = s.z; h.c = s.w & 1;
    }
    Fn64 ef, tf; Hd64 eh, th;
    block_scan(f, h, L, &ef, &eh, &tf, &th);
    if (threadIdx.x == 0) { agg_fn[blockIdx.x] = tf; agg_hd[blockIdx.x] = th; }
}

// clear mask bits [lo, hi) (clamped to limit); afterwards re-set the first / last bit of the range on request
__device__ __forceinline__ void clear_range(uint64_t* bits, int64_t lo, int64_t hi, int64_t limit, int keep_first,
                                            int keep_last) {
    if (hi > limit) hi = limit;
    if (lo >= hi) return;
    for (int64_t w = lo >> 6; w <= (hi - 1) >> 6; ++w) {
        const int64_t base = w << 6;
        uint64_t m = ~0ull;
        if (lo > base) m &= ~0ull << (lo - base);
        if (hi < base + 64) m &= (1ull << (hi - base)) - 1ull;
        uint64_t v = bits[w] & ~m;
        if (keep_first && (lo >> 6) == w) v |= 1ull << (lo & 63);
        if (keep_last && ((hi - 1) >> 6) == w) v |= 1ull << ((hi - 1) & 63);
        bits[w] = v;
    }
}

// stage 2b: one block per 1024 tiles.  Every block first composes the aggregates of the blocks before it (pending
// starts entering the block) and after it (starts before the next closing after the block), then resolves its own
// tiles and appends the ones whose provisional assumptions were wrong to the fix list.
__global__ __launch_bounds__(kScanThreads) void k_scan_resolve(const int4* __restrict__ summ, int64_t n_tiles,
                                                               const Fn64* __restrict__ agg_fn,
                                                               const Hd64* __restrict__ agg_hd, int n_blocks,
                                                               uint64_t* __restrict__ patch_bits, int64_t total,
                                                               int64_t* __restrict__ fix_list, int* __restrict__ fix_q,
                                                               int* __restrict__ fix_tz, int64_t* __restrict__ fix_count) {
    __shared__ ScanLds L;
    const int b = blockIdx.x;
    // (1) aggregates of the other blocks, contiguous range per thread, ordered
    Fn64 pf = fn_identity();
    Hd64 sh = hd_identity();
    if (n_blocks > 1) {
        const int per = (n_blocks + kScanThreads - 1) / kScanThreads;
        const int lo = min((int)threadIdx.x * per, n_blocks), hi = min(lo + per, n_blocks);
        Fn64 f = fn_identity();
        Hd64 h = hd_identity();
        for (int j = lo; j < hi; ++j) {
            if (j < b) f = fn_then(f, agg_fn[j]);
            if (j > b) h = hd_then(h, agg_hd[j]);
        }
        Fn64 ef; Hd64 eh;
        block_scan(f, h, L, &ef, &eh, &pf, &sh);
    }
    const long long q_block_in = fn_apply(pf, 0);
    Hd64 rest; rest.h = sh.h; rest.c = 1;   // what follows the block: sh.h starts before the next closing

    // (2) my tile
    const int64_t t = (int64_t)b * kScanThreads + threadIdx.x;
    int4 s = make_int4(0, 0, 0, 0);
    Fn64 f = fn_identity();
    Hd64 h = hd_identity();
    if (t < n_tiles) {
        s = summ[t];
        f = fn_of(s);
        h.h = s.z; h.c = s.w & 1;
    }
    Fn64 ef, tf; Hd64 eh, th;
    block_scan(f, h, L, &ef, &eh, &tf, &th);
    long long q_in = 0;
    int tz = 0, need = 0;
    if (t < n_tiles) {
        q_in = fn_apply(ef, q_block_in);
        const long long q_end = fn_apply(f, q_in);
        const long long h_next = hd_then(eh, rest).h;
        tz = (q_end + h_next) > 0;
        const int tz0 = s.y > 0;
        need = q_in != 0 || tz != tz0;
        // Common cases are patched in place (bitmask mode): exactly one pending start entering a tile whose head
        // block has no start of its own zeroes that head block; a tail block that turns out to be zeroed is cleared.
        // What stays in a cleared block: the C_SYM bit of its last char and the bit of a string start.
        const int geom = s.w;
        if (need && patch_bits && (geom & 1) && q_in <= 1 && (q_in == 0 || s.z == 0)) {
            const int64_t t0 = t * kTile;
            const int64_t t_end = min(t0 + kTile, total);
            if (q_in == 1) clear_range(patch_bits, t0, t0 + ((geom >> 1) & 0x1FFF), t_end, 0, (geom >> 27) & 1);
            if (tz != tz0) clear_range(patch_bits, t0 + ((geom >> 14) & 0x1FFF), t_end, t_end, (geom >> 28) & 1, (geom >> 29) & 1);
            need = 0;
        }
    }
    // one atomic per wave (a single counter word saturates at a few dozen atomics per microsecond)
    const lk_u64 m = __ballot(need);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(reinterpret_cast<unsigned long long*>(fix_count), (unsigned long long)lk_popc(m));
        base = __shfl(base, 0);
        if (need) {
            const unsigned long long slot = base + (unsigned long long)lk_popc(m & ((1ull << lane) - 1ull));
            fix_list[slot] = t;
            fix_q[slot] = (int)(q_in < (1 << 20) ? q_in : (1 << 20));   // a tile has <= 4096 closings: clamp is exact
            fix_tz[slot] = tz;
        }
    }
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
// host-callable launchers
// ---------------------------------------------------------------------------------------------------------------
static inline int blocks_for(int64_t n_items, int wpb, int n_cu, int max_blocks_per_cu) {
    int64_t b = (n_items + wpb - 1) / wpb;
    const int64_t cap = (int64_t)n_cu * max_blocks_per_cu;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

hipError_t launch_tile_index(const int64_t* row_off, int64_t n_str, int64_t n_tiles, int64_t* tile_first,
                             int64_t* fix_count, hipStream_t st) {
    const int64_t n = n_str + 1;
    const int threads = 256;
    const int64_t blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(k_tile_index, dim3((unsigned)blocks), dim3(threads), 0, st, row_off, n_str, n_tiles, tile_first,
                       fix_count);
    return hipGetLastError();
}

template <int WPB, int PF>
static hipError_t launch_main_bits(const SplitParams& P, int n_cu, hipStream_t st) {
    const int blocks = blocks_for(P.n_tiles, WPB, n_cu, 1);
    hipLaunchKernelGGL((k_split_tiles<kModeBits, false, WPB, PF>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_split_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st) {
    constexpr int WPB = kWavesPerBlockMain;
    if (mode == kModeBits) {
        static const int variant = [] { const char* e = getenv("LATOK_VARIANT"); return e ? atoi(e) : 0; }();
        switch (variant) {   // experiment switch; 0 = shipped configuration
            case 1: return launch_main_bits<16, 0>(P, n_cu, st);
            case 2: return launch_main_bits<12, 2>(P, n_cu, st);
            case 3: return launch_main_bits<8, 2>(P, n_cu, st);
            case 4: return launch_main_bits<12, 0>(P, n_cu, st);
            case 5: return launch_main_bits<12, 3>(P, n_cu, st);
            default: return launch_main_bits<WPB, kPrefetchMain>(P, n_cu, st);
        }
    }
    const int blocks = blocks_for(P.n_tiles, WPB, n_cu, 1);
    if (mode == kModeValues)
        hipLaunchKernelGGL((k_split_tiles<kModeValues, false, WPB, 0>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    else
        hipLaunchKernelGGL((k_split_tiles<kModeBlockMask, false, WPB, 0>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_scan_summaries(const SplitParams& P, int mode, hipStream_t st) {
    const int n_blocks = (int)((P.n_tiles + kScanThreads - 1) / kScanThreads);
    Fn64* agg_fn = reinterpret_cast<Fn64*>(P.scan_agg);
    Hd64* agg_hd = reinterpret_cast<Hd64*>(P.scan_agg + (size_t)n_blocks * sizeof(Fn64));
    if (n_blocks > 1)
        hipLaunchKernelGGL(k_scan_aggregate, dim3(n_blocks), dim3(kScanThreads), 0, st, P.summ, P.n_tiles, agg_fn, agg_hd);
    hipLaunchKernelGGL(k_scan_resolve, dim3(n_blocks), dim3(kScanThreads), 0, st, P.summ, P.n_tiles, agg_fn, agg_hd,
                       n_blocks, mode == kModeBits ? P.bits_out : nullptr, P.total, P.fix_list, P.fix_q, P.fix_tz,
                       P.fix_count);
    return hipGetLastError();
}

hipError_t launch_fix_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st) {
    constexpr int WPB = kWavesPerBlockFix;
    // the number of tiles to repair is only known on the device; a modest fixed grid loops over the list
    int blocks = blocks_for(P.n_tiles / 8 + 1, WPB, n_cu, 2);
    if (blocks > 2 * n_cu) blocks = 2 * n_cu;
    if (mode == kModeBits)
        hipLaunchKernelGGL((k_split_tiles<kModeBits, true, WPB, 0>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    else if (mode == kModeValues)
        hipLaunchKernelGGL((k_split_tiles<kModeValues, true, WPB, 0>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    else
        hipLaunchKernelGGL((k_split_tiles<kModeBlockMask, true, WPB, 0>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    return hipGetLastError();
}

}  // namespace latok
