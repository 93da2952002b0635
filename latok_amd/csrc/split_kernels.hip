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

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// stage 0: tile_first[t] = index of the first string whose start offset is >= t * kTile
// (one thread per row_off entry, including the end sentinel row_off[n_str])
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_tile_index(const int64_t* __restrict__ row_off, int64_t n_str, int64_t n_tiles,
                             int64_t* __restrict__ tile_first) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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

template <int MODE>
__device__ __forceinline__ void process_tile(const SplitParams& P, const TileLds& L, int64_t t, int q_in, int tail_zero,
                                             bool write_summary, int lane) {
    const int64_t t0 = t * kTile;
    const int64_t total = P.total;

    // ---- phase 1: classify 4096 chars, 4 per lane per step, into the staging buffer --------------------------
    if (t0 + kTile <= total) {
        const u32x4* src = reinterpret_cast<const u32x4*>(P.cps + t0) + lane;
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + 64 * i);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = classify4(L.t1, L.t2, v[i]);
            *reinterpret_cast<uint32_t*>(L.stage + stage_addr(256u * i + 4u * lane)) = c;
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
    if (lane < 3) {
        const int64_t hp = lane == 0 ? t0 - 1 : t0 + kTile + (lane - 1);
        L.halo[lane] = (hp >= 0 && hp < total) ? (uint8_t)classify1(L.t1, L.t2, P.cps[hp]) : (uint8_t)0;
    }
    L.bw[lane] = 0;
    if (lane == 0) L.bw[64] = 0;
    wave_lds_sync();
    {
        int64_t idx0 = P.tile_first[t];
        for (;;) {
            const int64_t idx = idx0 + lane;
            const int64_t ro = idx <= P.n_str ? P.row_off[idx] : INT64_MAX;
            const int64_t rel = ro - t0;
            if (rel >= 0 && rel < kTile + 64) atomicOr(&L.bw[rel >> 6], 1ull << (rel & 63));
            const int64_t last = __shfl(ro, 63);
            if (last >= t0 + kTile + 64) break;
            idx0 += 64;
        }
    }
    wave_lds_sync();

    // ---- phase 2: lane = one 64-char word ---------------------------------------------------------------------
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
    const lk_u64 B = L.bw[lane];
    const lk_u64 Bn = L.bw[lane + 1] & 3ull;

    lk_u64 plane[8];
    lk_bitslice64(d, plane);
    const lk_feat f = lk_decode(plane);
    const lk_local loc = lk_rules(f, h, B, Bn);
    lk_fwd fw = lk_forward(loc.start, loc.S, B);

    // forward: inclusive (max,+) scan of the per-word queue transfer functions over the 64 lanes
    lk_qfn inc = lk_qfn_of(fw);
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        lk_qfn o;
        o.a = __shfl_up(inc.a, dlt);
        o.b = __shfl_up(inc.b, dlt);
        if (lane >= dlt) inc = lk_qfn_then(o, inc);
    }
    lk_qfn exc;
    exc.a = __shfl_up(inc.a, 1);
    exc.b = __shfl_up(inc.b, 1);
    const int r = lane > 0 ? lk_qfn_apply(exc, q_in) : q_in;
    lk_qfn tile_fn;
    tile_fn.a = __shfl(inc.a, 63);
    tile_fn.b = __shfl(inc.b, 63);
    if (r > 0) lk_apply_extra(fw, r);

    if (write_summary) {
        const lk_u64 closing_lanes = __ballot(fw.has_closing);
        const int first_lane = closing_lanes ? lk_ctz(closing_lanes) : 64;
        const int contrib = lane < first_lane ? lk_popc(loc.start) : (lane == first_lane ? fw.head_starts : 0);
        const int head = wave_sum(contrib);
        if (lane == 0) P.summ[t] = make_int4(tile_fn.a, tile_fn.b, head, closing_lanes != 0);
    }

    // backward: zeroing closings clear the block below them; the carry chain over lanes is one 64-bit add on ballots
    const lk_u64 zall = fw.zs | fw.zb;
    const int z0_next = __shfl_down((int)(zall & 1ull), 1);
    const lk_bwd bw = lk_backward_prepare(zall, loc.S, B, lane < 63 ? z0_next : 0);
    const int tz = tail_zero >= 0 ? tail_zero : (lk_qfn_apply(tile_fn, q_in) > 0);
    // chain order is lane 63 -> 0, so reverse the ballots: bit i' = lane 63 - i'
    const lk_u64 G = lk_rev(__ballot(bw.g)), Pm = lk_rev(__ballot(bw.p));
    const lk_u64 X = G | Pm, Y = G;
    const lk_u64 carries_in = (X + Y + (lk_u64)tz) ^ X ^ Y;
    const int cin = (int)((carries_in >> (63 - lane)) & 1ull);
    const lk_u64 cleared = lk_backward_fill(bw, cin, loc.S);

    const int64_t base = t0 + 64 * (int64_t)lane;
    if (base < total) {
        const int64_t remain = total - base;
        const lk_u64 valid = remain >= 64 ? ~0ull : ((1ull << remain) - 1ull);
        const lk_u64 keep = ~cleared;
        if (MODE == kModeBits) {
            P.bits_out[base >> 6] = ((loc.raw & keep) | loc.sym | B) & valid;
        } else {
            // split VALUES 0..5: (sum of the five C_SPLIT terms) * mask + C_SYM term; first char of a string = 1
            uint8_t* dst = P.values_out + base;
            const int n = remain >= 64 ? 64 : (int)remain;
#pragma unroll 1
            for (int w = 0; w < 16; ++w) {
                uint32_t packed = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * w + b;
                    int v = (int)((loc.t_space >> i) & 1) + (int)((loc.t_sym >> i) & 1) + (int)((loc.t_prevsym >> i) & 1) +
                            (int)((loc.t_camel_next >> i) & 1) + (int)((loc.t_camel_prev >> i) & 1);
                    v = ((keep >> i) & 1) ? v : 0;
                    v += (int)((loc.sym >> i) & 1);
                    if ((B >> i) & 1) v = 1;
                    packed |= (uint32_t)v << (8 * b);
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
}

// ---------------------------------------------------------------------------------------------------------------
// stage 1 / stage 3 kernel.  FIX = false: all tiles, grid-stride, q_in = 0, provisional tail, writes summaries.
//                            FIX = true : only the tiles listed by k_scan_summaries, with their exact inputs.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool FIX, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_split_tiles(SplitParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kTablesLdsBytes + WPB * kWaveLdsBytes];

    int64_t n_items = P.n_tiles;
    if (FIX) {
        n_items = *P.fix_count;
        if (n_items == 0) return;  // uniform: nothing to repair, skip the table load
    }
    // cooperative table load (global/L2 -> LDS), 16 B per thread per step
    {
        const uint4* s1 = reinterpret_cast<const uint4*>(P.t1);
        uint4* d1 = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < kStage1Pad / 16; i += WPB * 64) d1[i] = s1[i];
        const uint4* s2 = reinterpret_cast<const uint4*>(P.t2);
        uint4* d2 = reinterpret_cast<uint4*>(lds + kStage1Pad);
        for (int i = threadIdx.x; i < kStage2Len / 16; i += WPB * 64) d2[i] = s2[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    TileLds L;
    L.t1 = lds;
    L.t2 = lds + kStage1Pad;
    uint8_t* mine = lds + kTablesLdsBytes + wave * kWaveLdsBytes;
    L.stage = mine;
    L.halo = mine + kStageBytes;
    L.bw = reinterpret_cast<lk_u64*>(mine + kStageBytes + 16);

    const int64_t wave_gid = (int64_t)blockIdx.x * WPB + wave;
    const int64_t n_waves = (int64_t)gridDim.x * WPB;
    for (int64_t i = wave_gid; i < n_items; i += n_waves) {
        if (FIX) {
            const int64_t t = P.fix_list[i];
            process_tile<MODE>(P, L, t, P.fix_q[i], P.fix_tz[i], false, lane);
        } else {
            process_tile<MODE>(P, L, i, 0, -1, true, lane);
        }
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

constexpr int kScanThreads = 1024;

__global__ __launch_bounds__(kScanThreads) void k_scan_summaries(const int4* __restrict__ summ, int64_t n_tiles,
                                                                 int* __restrict__ tile_q,
                                                                 int64_t* __restrict__ fix_list,
                                                                 int* __restrict__ fix_q, int* __restrict__ fix_tz,
                                                                 int64_t* __restrict__ fix_count) {
    __shared__ Fn64 s_fn[kScanThreads];
    __shared__ Hd64 s_hd[kScanThreads];
    const int tid = threadIdx.x;
    const int64_t chunk = (n_tiles + kScanThreads - 1) / kScanThreads;
    const int64_t lo = min((int64_t)tid * chunk, n_tiles), hi = min(lo + chunk, n_tiles);

    // pass 1: compose my chunk of tile functions (forward) and head descriptors (backward)
    Fn64 F; F.a = 0; F.b = 0;
    Hd64 H; H.h = 0; H.c = 0;
    for (int64_t t = lo; t < hi; ++t) {
        const int4 s = summ[t];
        F = fn_then(F, fn_of(s));
        Hd64 e; e.h = s.z; e.c = s.w;
        H = hd_then(H, e);
    }
    s_fn[tid] = F;
    s_hd[tid] = H;
    __syncthreads();
    // inclusive scans across the 1024 chunks: prefix for the functions, suffix for the head descriptors
    for (int dlt = 1; dlt < kScanThreads; dlt <<= 1) {
        Fn64 o; Hd64 oh;
        const bool hf = tid >= dlt, hb = tid + dlt < kScanThreads;
        if (hf) o = s_fn[tid - dlt];
        if (hb) oh = s_hd[tid + dlt];
        __syncthreads();
        if (hf) s_fn[tid] = fn_then(o, s_fn[tid]);
        if (hb) s_hd[tid] = hd_then(s_hd[tid], oh);
        __syncthreads();
    }
    long long q = tid > 0 ? fn_apply(s_fn[tid - 1], 0) : 0;                   // pending starts entering my chunk
    long long h_next = tid + 1 < kScanThreads ? s_hd[tid + 1].h : 0;          // starts after my chunk before a closing

    // pass 2a (forward): pending starts entering every tile; a tile has <= 4096 closings, so clamping is exact
    for (int64_t t = lo; t < hi; ++t) {
        tile_q[t] = (int)(q < (1 << 20) ? q : (1 << 20));
        q = fn_apply(fn_of(summ[t]), q);
    }
    // pass 2b (backward): tail decision per tile, and the list of tiles whose provisional assumptions were wrong
    for (int64_t t = hi - 1; t >= lo; --t) {
        const int4 s = summ[t];
        const int qin = tile_q[t];
        const long long q_end = fn_apply(fn_of(s), qin);
        const int tz = (q_end + h_next) > 0;
        const int tz0 = s.y > 0;
        if (qin != 0 || tz != tz0) {
            const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long*>(fix_count), 1ull);
            fix_list[slot] = t;
            fix_q[slot] = qin;
            fix_tz[slot] = tz;
        }
        h_next = (long long)s.z + (s.w ? 0 : h_next);
    }
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
    hipError_t e = hipMemsetAsync(fix_count, 0, sizeof(int64_t), st);
    if (e != hipSuccess) return e;
    const int64_t n = n_str + 1;
    const int threads = 256;
    const int64_t blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(k_tile_index, dim3((unsigned)blocks), dim3(threads), 0, st, row_off, n_str, n_tiles, tile_first);
    return hipGetLastError();
}

hipError_t launch_split_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st) {
    constexpr int WPB = kWavesPerBlockMain;
    const int blocks = blocks_for(P.n_tiles, WPB, n_cu, 1);
    if (mode == kModeBits)
        hipLaunchKernelGGL((k_split_tiles<kModeBits, false, WPB>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    else
        hipLaunchKernelGGL((k_split_tiles<kModeValues, false, WPB>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_scan_summaries(const SplitParams& P, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_summaries, dim3(1), dim3(kScanThreads), 0, st, P.summ, P.n_tiles, P.tile_q,
                       P.fix_list, P.fix_q, P.fix_tz, P.fix_count);
    return hipGetLastError();
}

hipError_t launch_fix_tiles(const SplitParams& P, int mode, int n_cu, hipStream_t st) {
    constexpr int WPB = kWavesPerBlockFix;
    // the number of tiles to repair is only known on the device; a modest fixed grid loops over the list
    int blocks = blocks_for(P.n_tiles, WPB, n_cu, 1);
    if (blocks > 128) blocks = 128;
    if (mode == kModeBits)
        hipLaunchKernelGGL((k_split_tiles<kModeBits, true, WPB>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    else
        hipLaunchKernelGGL((k_split_tiles<kModeValues, true, WPB>), dim3(blocks), dim3(WPB * 64), 0, st, P);
    return hipGetLastError();
}

}  // namespace latok
