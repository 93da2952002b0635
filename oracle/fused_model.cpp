// fused_model.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A CPU model of the HIP pipeline in latok_amd/csrc/split_kernels.hip: same tiling (4096-char tiles, 64 "lanes" of
// one 64-bit word each), same per-lane math (it includes the product's lane_math.h and unicode_tables.inc), same
// four stages (tile index / tiles / summary scan / fix-up), with the wavefront written as a plain loop over 64
// lanes.  It exists to debug the *algorithm* against the reference-shaped oracle (latok_oracle.c) in a container
// without a GPU, and doubles as the multi-core "fused" CPU number.  The product never links or loads it.
//
//   g++ -O2 -shared -fPIC -I../latok_amd/csrc fused_model.cpp -o libfused_model.so
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "lane_math.h"
#include "unicode_tables.inc"

namespace {

constexpr int kTile = 4096;
constexpr int kLanes = 64;

struct TileSummary {
    int a, b;          // q transfer function of the tile
    int head_starts;   // starts before the first closing event of the tile
    int has_closing;
    // geometry of the edge blocks (mirrors the packed `geom` word of the HIP kernel)
    int c_rel, p_rel, head_sym, tail_keep, tail_sym;
};

inline uint32_t classify(uint32_t cp, bool rule_codes = false) {
    uint32_t hi = cp >> LATOK_TBL_SHIFT;
    if (hi > LATOK_TBL_STAGE1_LEN - 1) hi = LATOK_TBL_STAGE1_LEN - 1;
    uint32_t blk = kStage1[hi];
    const unsigned cls = kStage2[(blk << LATOK_TBL_SHIFT) | (cp & ((1u << LATOK_TBL_SHIFT) - 1))];
    return rule_codes ? kClassRuleCode[cls] : kClassCode[cls];
}

struct Model {
    const uint32_t* cps;
    const int64_t* row_off;
    int64_t n_str, total, n_tiles;
    std::vector<int64_t> tile_first;
    std::vector<TileSummary> summ;
    uint8_t* values;   // may be null
    uint64_t* bits;    // may be null
    const int8_t* bm_a1 = nullptr;   // block-mask mode (compat _gen_block_mask): planes from byte arrays
    const int8_t* bm_a2 = nullptr;
    int8_t* bm_out = nullptr;
    const lk_rule_tables* rules = nullptr;   // runtime rule tables (kModeRules of the HIP kernel)
    const uint8_t* u8 = nullptr;             // byte-space mode (kModeBytes): positions are UTF-8 bytes, row_off byte offsets
    uint64_t* space_bits = nullptr;          // optional: smeared SPACE plane (byte mode)
    int64_t n_fix = 0, n_patch = 0;

    void clear_range(int64_t lo, int64_t hi, int64_t limit, int keep_first, int keep_last) {
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

    bool cont_at(int64_t p) const { return u8 && p >= 0 && p < total && (u8[p] & 0xC0u) == 0x80u; }
    // byte mode: code of the char that owns byte p (a lead byte and up to 3 continuation bytes after it); 0 for stray
    // continuation bytes.  Decoding mirrors aux_kernels.hip:utf8_decode_at (truncated sequence -> U+FFFD).
    uint32_t byte_code_at(int64_t p) const {
        if (p < 0 || p >= total) return 0u;
        int64_t q = p;
        for (int k = 0; k < 3 && cont_at(q); ++k) --q;
        if (q < 0 || cont_at(q)) return 0u;
        const uint32_t b0 = u8[q];
        uint32_t cp = b0;
        int extra = 0;
        if (b0 >= 0xF0u) { cp = b0 & 0x07u; extra = 3; }
        else if (b0 >= 0xE0u) { cp = b0 & 0x0Fu; extra = 2; }
        else if (b0 >= 0xC0u) { cp = b0 & 0x1Fu; extra = 1; }
        for (int j = 1; j <= extra; ++j) {
            if (cont_at(q + j)) cp = (cp << 6) | (u8[q + j] & 0x3Fu);
            else { cp = 0xFFFDu; break; }
        }
        return classify(cp, rules != nullptr);
    }
    // byte mode, what phase 1 of the tile kernel leaves in the staging buffer: the code at LEAD bytes, 0 at continuation bytes
    uint32_t lead_code_at(int64_t p) const { return cont_at(p) ? LK_CODE_CONT : byte_code_at(p); }
    uint32_t code_at(int64_t p) const {
        if (u8) return lead_code_at(p);
        return (cps && p >= 0 && p < total) ? classify(cps[p], rules != nullptr) : 0u;
    }

    // stage 0: tile_first[t] = first string index s with row_off[s] >= t*kTile
    void build_tile_index() {
        tile_first.assign((size_t)n_tiles, n_str);
        int64_t prev_tile = -1;
        for (int64_t s = 0; s <= n_str; ++s) {
            int64_t tl = row_off[s] / kTile;
            for (int64_t w = prev_tile + 1; w <= tl && w < n_tiles; ++w) tile_first[(size_t)w] = s;
            if (tl > prev_tile) prev_tile = tl;
        }
    }

    void process_tile(int64_t t, int q_in, int tail_zero, bool write_summary) {
        const int64_t t0 = t * kTile;
        // string-start bits for chars t0 .. t0+4096+63
        lk_u64 Bw[kLanes + 1];
        memset(Bw, 0, sizeof(Bw));
        for (int64_t s = tile_first[(size_t)t]; s <= n_str; ++s) {
            int64_t rel = row_off[s] - t0;
            if (rel >= kTile + 64) break;
            Bw[rel >> 6] |= 1ull << (rel & 63);
        }
        lk_local loc[kLanes];
        lk_fwd fw[kLanes];
        lk_u64 Bl[kLanes];
        for (int j = 0; j < kLanes; ++j) {
            const int64_t base = t0 + 64 * j;
            uint32_t d[16];
            for (int k = 0; k < 16; ++k) {
                uint32_t v = 0;
                for (int b = 0; b < 4; ++b) v |= code_at(base + 4 * k + b) << (8 * b);
                d[k] = v;
            }
            lk_u64 plane[8];
            lk_bitslice64(d, plane);
            lk_halo h;
            h.prev = code_at(base - 1);
            h.next0 = code_at(base + 64);
            h.next1 = code_at(base + 65);
            Bl[j] = Bw[j];
            if (bm_a1) {
                lk_u64 st = 0, sp = 0;
                for (int i = 0; i < 64 && base + i < total; ++i) {
                    st |= (lk_u64)(bm_a1[base + i] != 0) << i;
                    sp |= (lk_u64)(bm_a2[base + i] != 0) << i;
                }
                loc[j] = lk_local();
                loc[j].start = st; loc[j].S = sp; loc[j].raw = ~0ull; loc[j].sym = 0;
            } else if (u8) {
                // the tile kernel's phase 2 in byte space: lead-only codes in, the planes the PREV_* columns and the token
                // stripping need are smeared over the continuation bytes by mask arithmetic (lane_math.h)
                lk_halo_bytes hb;
                uint32_t codes4 = 0;
                for (int k = 1; k <= 4; ++k) codes4 |= lead_code_at(base - k) << (8 * (4 - k));
                int cin_left = 0;
                lk_owner_before(codes4, &hb.prev, &cin_left);
                if (hb.prev != byte_code_at(base - 1)) abort();   // the owner state agrees with the per-byte definition
                hb.next_codes = 0;
                for (int k = 0; k < 8; ++k) hb.next_codes |= (lk_u64)code_at(base + 64 + k) << (8 * k);
                hb.next_B = (uint32_t)(Bw[j + 1] & 0xFFFFull);
                lk_u64 Ss = 0;
                const lk_u64 C = lk_take_cont_plane(plane);
                for (int i = 0; i < 64; ++i)
                    if (((C >> i) & 1ull) != (lk_u64)cont_at(base + i)) abort();
                lk_smear_planes<0x37u>(plane, C, hb.prev, cin_left);
                for (int i = 0; i < 64; ++i)   // smeared planes == per-byte definition
                    for (int b = 0; b < 8; ++b)
                        if (((0x37u >> b) & 1u) && ((plane[b] >> i) & 1ull) != ((byte_code_at(base + i) >> b) & 1u)) abort();
                if (rules) loc[j] = lk_rules_generic_bytes(plane, C, hb, Bw[j], *rules, &Ss);
                else loc[j] = lk_rules_bytes(plane, C, hb, Bw[j], &Ss);
                if (!rules && !lk_rules_bytes_weird(plane, C, hb.next_codes)) {   // the fast form agrees with the general one
                    lk_u64 Sg = 0;
                    const lk_local g = lk_rules_bytes_general(plane, C, hb, Bw[j], &Sg);
                    if (g.raw != loc[j].raw || g.start != loc[j].start || g.sym != loc[j].sym || g.S != loc[j].S || Sg != Ss ||
                        g.t_camel_next != loc[j].t_camel_next || g.t_prevsym != loc[j].t_prevsym)
                        abort();
                }
                if (space_bits && base < total) {
                    const int64_t remain = total - base;
                    space_bits[base >> 6] = Ss & (remain >= 64 ? ~0ull : ((1ull << remain) - 1ull));
                }
            } else if (rules) {
                loc[j] = lk_rules_generic(plane, h, Bw[j], Bw[j + 1] & 3ull, *rules);
            } else {
                loc[j] = lk_rules(lk_decode(plane), h, Bw[j], Bw[j + 1] & 3ull);
            }
            fw[j] = lk_forward(loc[j].start, loc[j].S, Bw[j]);
        }
        // "wave" exclusive scan of the q transfer functions
        lk_qfn acc;
        acc.a = 0; acc.b = 0;  // identity on q >= 0
        int head = 0, seen_closing = 0;
        for (int j = 0; j < kLanes; ++j) {
            const int r = lk_qfn_apply(acc, q_in);
            if (!seen_closing) {
                head += fw[j].has_closing ? fw[j].head_starts : lk_popc(loc[j].start);
                seen_closing = fw[j].has_closing;
            }
            acc = lk_qfn_then(acc, lk_qfn_of(fw[j]));
            if (r > 0) lk_apply_extra(fw[j], r);
        }
        if (write_summary) {
            TileSummary& s = summ[(size_t)t];
            s.a = acc.a; s.b = acc.b; s.head_starts = head; s.has_closing = seen_closing;
            s.c_rel = kTile; s.p_rel = 0; s.tail_keep = 0;
            for (int j = 0; j < kLanes; ++j) {
                const lk_u64 cl = loc[j].S | Bl[j];
                if (!cl) continue;
                if (s.c_rel == kTile) s.c_rel = 64 * j + lk_ctz(cl);
                const int top = 63 - __builtin_clzll(cl);
                const int s_top = (int)((loc[j].S >> top) & 1ull);
                s.p_rel = 64 * j + top + s_top;
                s.tail_keep = 1 - s_top;
            }
            const int hs = s.c_rel > 0 ? s.c_rel - 1 : 0;
            s.head_sym = s.c_rel > 0 ? (int)((loc[hs >> 6].sym >> (hs & 63)) & 1ull) : 0;
            s.tail_sym = (int)(loc[kLanes - 1].sym >> 63);
        }
        // backward fill, carry travels from lane 63 down to lane 0
        // tail_zero < 0: provisional decision = "a start is still pending at the tile end" (its closing event,
        // wherever it is, will zero the open block)
        int cin = tail_zero >= 0 ? tail_zero : (lk_qfn_apply(acc, q_in) > 0);
        for (int j = kLanes - 1; j >= 0; --j) {
            const lk_u64 zall = fw[j].zs | fw[j].zb;
            const int gen_top = (j < kLanes - 1) ? (int)((fw[j + 1].zs | fw[j + 1].zb) & 1ull) : 0;
            lk_bwd bw = lk_backward_prepare(zall, loc[j].S, Bl[j], gen_top);
            const lk_u64 cleared = lk_backward_fill(bw, cin, loc[j].S);
            cin = bw.g | (bw.p & cin);
            const int64_t base = t0 + 64 * j;
            if (base >= total) continue;
            const lk_u64 valid = (total - base >= 64) ? ~0ull : ((1ull << (total - base)) - 1ull);
            const lk_u64 keep = ~cleared;
            const lk_u64 out = (((loc[j].raw & keep) | loc[j].sym | Bl[j])) & valid;
            if (bits) bits[base >> 6] = out;
            if (bm_out)
                for (int i = 0; i < 64 && base + i < total; ++i) bm_out[base + i] = (int8_t)((keep >> i) & 1);
            if (values) {
                for (int i = 0; i < 64 && base + i < total; ++i) {
                    const lk_u64 m = 1ull << i;
                    int v = (int)!!(loc[j].t_space & m) + !!(loc[j].t_sym & m) + !!(loc[j].t_prevsym & m) +
                            !!(loc[j].t_camel_next & m) + !!(loc[j].t_camel_prev & m);
                    v = (keep & m) ? v : 0;
                    v += !!(loc[j].sym & m);
                    if (Bl[j] & m) v = 1;
                    values[base + i] = (uint8_t)v;
                }
            }
        }
    }

    void run() {
        if (n_tiles == 0) return;
        build_tile_index();
        summ.resize((size_t)n_tiles);
        // stage 1: every tile with q_in = 0 and the provisional tail decision (pending start at the tile end)
        for (int64_t t = 0; t < n_tiles; ++t) {
            process_tile(t, 0, -1, true);
        }
        // stage 2: forward scan of q, backward scan of "starts before the next closing"
        // (q is 64-bit here: a tile only ever sees min(q, 2^20), it has at most 4096 closings to feed)
        auto apply64 = [](const TileSummary& f, long long q) -> long long {
            if (f.a <= LK_NEG_INF / 2) return f.b;
            return std::max<long long>(q + f.a, f.b);
        };
        std::vector<long long> q_in((size_t)n_tiles);
        std::vector<int> tz((size_t)n_tiles);
        long long q = 0;
        for (int64_t t = 0; t < n_tiles; ++t) {
            q_in[(size_t)t] = q;
            q = apply64(summ[(size_t)t], q);
        }
        long long H = 0;  // starts after the end of tile t before the first closing event
        for (int64_t t = n_tiles - 1; t >= 0; --t) {
            const long long q_end = apply64(summ[(size_t)t], q_in[(size_t)t]);
            tz[(size_t)t] = (q_end + H) > 0;
            H = summ[(size_t)t].head_starts + (summ[(size_t)t].has_closing ? 0 : H);
        }
        // stage 3: recompute the tiles whose assumptions were wrong
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int tz0 = summ[(size_t)t].b > 0;
            if (q_in[(size_t)t] != 0 || tz[(size_t)t] != tz0) {
                const TileSummary& sm = summ[(size_t)t];
                if (bits && !values && !bm_out && !rules && !u8 && sm.has_closing && q_in[(size_t)t] <= 1 &&
                    (q_in[(size_t)t] == 0 || sm.head_starts == 0)) {
                    // patch in place (mirrors k_scan_resolve)
                    const int64_t t0 = t * kTile, t_end = std::min<int64_t>(t0 + kTile, total);
                    if (q_in[(size_t)t] == 1) clear_range(t0, t0 + sm.c_rel, t_end, 0, sm.head_sym);
                    if (tz[(size_t)t] != tz0) clear_range(t0 + sm.p_rel, t_end, t_end, sm.tail_keep, sm.tail_sym);
                    ++n_patch;
                    continue;
                }
                process_tile(t, (int)std::min<long long>(q_in[(size_t)t], 1 << 20), tz[(size_t)t], false);
                ++n_fix;
            }
        }
    }
};

}  // namespace

extern "C" int fused_split_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, uint8_t* values_out,
                                 uint64_t* bits_out, int64_t* n_fix_out) {
    if (n_str < 0) return -1;
    Model m;
    m.cps = cps;
    m.row_off = row_off;
    m.n_str = n_str;
    m.total = n_str > 0 ? row_off[n_str] : 0;
    m.n_tiles = (m.total + kTile - 1) / kTile;
    m.values = values_out;
    m.bits = bits_out;
    m.run();
    if (n_fix_out) *n_fix_out = m.n_fix + m.n_patch;
    return 0;
}

// byte-space mode: UTF-8 bytes + byte offsets in, boundary bits at lead-byte positions out (kModeBytes of the HIP kernel)
extern "C" int fused_split_batch_utf8(const uint8_t* u8, const int64_t* byte_off, int64_t n_str, uint64_t* bits_out,
                                      uint64_t* space_out, int64_t* n_fix_out) {
    if (n_str < 0) return -1;
    Model m;
    m.cps = nullptr;
    m.u8 = u8;
    m.row_off = byte_off;
    m.n_str = n_str;
    m.total = n_str > 0 ? byte_off[n_str] : 0;
    m.n_tiles = (m.total + kTile - 1) / kTile;
    m.values = nullptr;
    m.bits = bits_out;
    m.space_bits = space_out;
    m.run();
    if (n_fix_out) *n_fix_out = m.n_fix + m.n_patch;
    return 0;
}

// compat _gen_block_mask through the same pipeline (mirrors kModeBlockMask of the HIP kernel)
// runtime rule tables: rows[3][LK_MAX_RULE_ROWS] column sets, n_rows[3] (same packing as the C ABI keeps internally)
extern "C" int fused_split_batch_rules(const uint32_t* cps, const int64_t* row_off, int64_t n_str, const uint32_t* rows,
                                       const int32_t* n_rows, uint64_t* bits_out, int64_t* n_fix_out) {
    if (n_str < 0) return -1;
    lk_rule_tables R;
    memset(&R, 0, sizeof(R));
    for (int t = 0; t < 3; ++t) {
        if (n_rows[t] < 0 || n_rows[t] > LK_MAX_RULE_ROWS) return -1;
        R.n_rows[t] = n_rows[t];
        for (int r = 0; r < n_rows[t]; ++r) R.row[t][r] = rows[t * LK_MAX_RULE_ROWS + r];
    }
    Model m;
    m.cps = cps;
    m.row_off = row_off;
    m.n_str = n_str;
    m.total = n_str > 0 ? row_off[n_str] : 0;
    m.n_tiles = (m.total + kTile - 1) / kTile;
    m.values = nullptr;
    m.bits = bits_out;
    m.rules = &R;
    m.run();
    if (n_fix_out) *n_fix_out = m.n_fix + m.n_patch;
    return 0;
}

// byte space with run-time rule tables (kModeBytesRules of the HIP kernel)
extern "C" int fused_split_batch_utf8_rules(const uint8_t* u8, const int64_t* byte_off, int64_t n_str, const uint32_t* rows,
                                            const int32_t* n_rows, uint64_t* bits_out, uint64_t* space_out) {
    if (n_str < 0) return -1;
    lk_rule_tables R;
    memset(&R, 0, sizeof(R));
    for (int t = 0; t < 3; ++t) {
        if (n_rows[t] < 0 || n_rows[t] > LK_MAX_RULE_ROWS) return -1;
        R.n_rows[t] = n_rows[t];
        for (int r = 0; r < n_rows[t]; ++r) R.row[t][r] = rows[t * LK_MAX_RULE_ROWS + r];
    }
    Model m;
    m.cps = nullptr;
    m.u8 = u8;
    m.row_off = byte_off;
    m.n_str = n_str;
    m.total = n_str > 0 ? byte_off[n_str] : 0;
    m.n_tiles = (m.total + kTile - 1) / kTile;
    m.values = nullptr;
    m.bits = bits_out;
    m.space_bits = space_out;
    m.rules = &R;
    m.run();
    return 0;
}

extern "C" int fused_block_mask(const int8_t* a1, const int8_t* a2, int64_t n, int8_t* out) {
    if (n <= 0) return 0;
    const int64_t row[2] = {0, n};
    Model m;
    m.cps = nullptr;
    m.row_off = row;
    m.n_str = 1;
    m.total = n;
    m.n_tiles = (n + kTile - 1) / kTile;
    m.values = nullptr;
    m.bits = nullptr;
    m.bm_a1 = a1; m.bm_a2 = a2; m.bm_out = out;
    m.run();
    int any1 = 0, any2 = 0;
    for (int64_t i = 0; i < n; ++i) { any1 |= a1[i] != 0; any2 |= a2[i] != 0; }
    out[0] = (any1 && !any2) ? 0 : 1;   // reference quirk: element 0 (latok.c:224 vs :211-216)
    return 0;
}


// test hook: 64 raw ASCII bytes -> their split codes through lk_bitslice64 + lk_ascii_code_planes (the table-free ASCII
// classification of the narrow-input tile kernels), un-sliced again; and the same bytes through the class table
extern "C" void fused_ascii_codes(const uint8_t* bytes64, uint8_t* planes_codes_out, uint8_t* table_codes_out) {
    uint32_t d[16];
    for (int k = 0; k < 16; ++k) d[k] = (uint32_t)bytes64[4 * k] | ((uint32_t)bytes64[4 * k + 1] << 8) | ((uint32_t)bytes64[4 * k + 2] << 16) |
                                        ((uint32_t)bytes64[4 * k + 3] << 24);
    lk_u64 raw[8], p[8];
    lk_bitslice64(d, raw);
    lk_ascii_code_planes(raw, p);
    for (int i = 0; i < 64; ++i) {
        uint32_t c = 0;
        for (int b = 0; b < 8; ++b) c |= (uint32_t)((p[b] >> i) & 1ull) << b;
        planes_codes_out[i] = (uint8_t)c;
        table_codes_out[i] = (uint8_t)classify(bytes64[i]);
    }
}

// test hook: lane_math.h lk_pext64 (code-point results from byte-space masks: compact_kernels.hip, k_lead_compress)
extern "C" unsigned long long fused_pext64(unsigned long long x, unsigned long long m) {
    const lk_u64 want = lk_pext64(x, m), b = lk_pext64(~x, m);
    static uint8_t tab[256];                    // the table form (k_lead_compress: 4-bit pexts from LDS) must agree as well
    static bool tab_ready = false;
    if (!tab_ready) { for (uint32_t i = 0; i < 256; ++i) tab[i] = lk_pext4_entry(i); tab_ready = true; }
    lk_u64 c = x, d = ~x, e = x, unused = 0;
    lk_pext64_lut<true>(&c, &d, m, tab);
    lk_pext64_lut<false>(&e, &unused, m, tab);
    if (c != want || d != b || e != want) return ~want;
    return want;
}

// test hook: lane_math.h lk_lead_entry_of + lk_lead_index (the byte-space kernel's table-driven decode of multi-byte chars):
// W = 4 bytes from byte b0 on; returns 1 when the sequence is cut short, else 0 with *cp = (stage-1 index << 6) | stage-2 index
// (a byte below 0xC0 starts nothing: stage 1's last entry, index 0)
extern "C" int fused_lead_decode(uint32_t W, uint32_t* cp) {
    uint32_t off2 = 0, R = 0;
    const bool bad = lk_lead_index(lk_lead_entry_of(W & 0xFFu), W, &off2, &R);
    *cp = ((off2 >> 1) << LK_B6_SHIFT) | (R & 0x3Fu);
    return bad ? 1 : 0;
}
