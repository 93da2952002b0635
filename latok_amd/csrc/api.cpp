// api.cpp -- host side of liblatok_hip.so: the C ABI declared in include/latok_hip.h.
//
// Thin by design: argument checks, workspace management, H2D/D2H staging for host-pointer calls, and the launch
// sequence of the pipeline (tile index -> tiles -> resolve/repair).  No compute happens on the
// host and there is no CPU fallback: without a HIP device every compute entry point fails.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/latok_hip.h"
#include "corpus_gen.h"
#include "flow_hazards.h"
#include "kernels.h"
#include "unicode_tables.inc"

static_assert(LATOK_TBL_SHIFT == latok::kTblShift, "table shift");
static_assert(LATOK_TBL_STAGE1_LEN == latok::kStage1Len, "stage-1 length");
static_assert(LATOK_TBL_NBLOCKS * (1 << LATOK_TBL_SHIFT) == latok::kStage2Len, "stage-2 length");
static_assert(LATOK_TILE_CHARS == latok::kTile, "tile size");

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(LATOK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    unsigned gen = 0;   // counts (re)allocations: fresh memory holds anything
    int ensure(size_t bytes) {
        if (bytes <= cap) return LATOK_OK;
        ++gen;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(LATOK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return LATOK_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// pinned, device-mapped host memory: small host-pointer calls stage their inputs here and have the kernels read and
// write it in place over the bus, so that a call costs launches + ONE synchronisation instead of five blocking copies
struct PinBuf {
    void* h = nullptr;   // host address
    void* d = nullptr;   // the same memory as the device sees it
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return LATOK_OK;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            release();
            return fail(LATOK_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        memset(h, 0, want);
        cap = want;
        return LATOK_OK;
    }
    void release() {
        if (h) (void)hipHostFree(h);
        h = d = nullptr;
        cap = 0;
    }
};

// Small-batch completion: the kernel stores a sequence number into pinned memory after its last output and the host
// polls that word -- the end-of-kernel signal takes ~10 us longer to come back through the runtime than the data does.
// Bounded: after kPollNs without the word (first launch of a code object, a busy stream, a faulted kernel) the caller
// falls back to hipStreamSynchronize, which also surfaces errors.  LATOK_SMALL_POLL=0 turns polling off.
constexpr long long kPollNs = 200000;
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
static bool poll_completion() {
    static const bool on = [] { const char* e = getenv("LATOK_SMALL_POLL"); return !(e && e[0] == '0'); }();
    return on;
}
static bool wait_completion_word(const unsigned long long* word, unsigned long long seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 256; ++i) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return true;
            cpu_relax();
        }
        if (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > kPollNs) return false;
    }
}

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// host-pointer calls up to this size take the pinned zero-copy path: the arrays are copied into pinned mapped memory by
// the host and the kernels read / write them over the bus, including the binary searches over row_off.  Measured against
// the staged-copy path (tools/batch_size_sweep.py, offsets of strings of ~105 chars, us per call): 16 K chars 70 / 112,
// 54 K 60 / 120, 108 K 75 / 137; the two meet near 1 M chars, where the host's own memcpy into the pinned area is what
// the call costs either way.
constexpr int64_t kSmallChars = 262144;
constexpr int64_t kSmallStrings = 16384;

// One context = one device, one stream, one set of tables / workspaces / staging buffers, one rule-table state and one
// lock.  Calls on the same context are serialised by its lock; calls on different contexts (other devices, or the same
// device twice) share nothing and run concurrently.  Every entry point works on the calling thread's CURRENT context
// (latok_ctx_set_current; default: the process-wide one that latok_init creates) -- the same model as hipSetDevice.
struct Ctx {
    std::mutex mu;
    bool inited = false;
    int device = -1, n_cu = 0;
    hipStream_t stream = nullptr;
    // tables
    DevBuf t1, t1rule, t2code, t2cls, cw;
    DevBuf tb6, tb6rule;   // byte space: its own class table, split codes / rule codes (build_byte_tables)
    // runtime rule tables (latok_set_rules); off = the built-in default_tokenizer.py tables
    bool rules_on = false;
    lk_rule_tables rules;
    // pipeline workspace (sized by tiles)
    DevBuf summ, seg_agg, fix_count;
    // staging for host-pointer calls and for the offsets API
    DevBuf h_cps, h_row, h_out, bits, space, kept, wcnt, wpref, counts, bases, scan_tot, scalar, h_aux, tile_first;
    PinBuf pin, pin_tot;   // pin_tot: 64 bytes the scans drop their grand totals into (read after a stream sync, no copy)
    unsigned long long small_seq = 0;   // completion word of the single-launch small-batch path (pin_tot word 2)
    DevBuf done_ctr;                    // workgroup counter of latok::DoneSignal (0 between launches)
    std::vector<uint32_t> hd_cps;       // host decode of small UTF-8 batches (host_decode_small)
    std::vector<int64_t> hd_row, hd_pos;
    unsigned done_ctr_seen = 0;
    DevBuf u_bytes, u_boff, u_cnt, u_row, u_pref;   // UTF-8 ingest: uploaded bytes / byte offsets, per-string cp counts, cp offsets
    DevBuf u_lead, u_bspace, u_cpbits, u_cpspace;   // code-point results from byte space (cp_masks_via_bytes): lead-byte mask, byte-space
                                                    // SPACE plane, the packed code-point masks
    DevBuf codes;              // featurize: rule code of every char (SplitParams::codes_out)
    // chunked host pipeline (compact_host_pipelined): copy streams, events and double buffers
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_in_ready[2] = {nullptr, nullptr}, ev_k_done[2] = {nullptr, nullptr}, ev_d2h_done[2] = {nullptr, nullptr};
    DevBuf pipe_in[2], pipe_row[2], pipe_counts[2], pipe_items[2], pipe_feat[2];
    PinBuf pipe_tot;
    DevBuf chain, chain_ctl;   // k_word_counts_scan: look-back state per workgroup, {ticket counter}
    unsigned scan_epoch = 0;
    bool chain_ready = false;
    unsigned chain_seen = 0, chain_ctl_seen = 0;   // DevBuf::gen of the allocations the state was last cleared in
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // batch flow (latok_flow_*): up to kFlowSlots device-resident batches in flight, each with its own stream and workspace
    struct FlowSlot {
        DevBuf summ, seg_agg, fix_count, tile_first;
        // compaction passes of the slot's batch (latok_flow_split_offsets / _token_spans): what the context's own calls keep
        // in bits / space / kept / wcnt / wpref / bases / chain / chain_ctl / scalar
        DevBuf bits, space, kept, wcnt, wpref, bases, chain, chain_ctl, scalar, codes, widened;   // codes / widened: featurize
        unsigned scan_epoch = 0, chain_seen = 0, chain_ctl_seen = 0;
        bool chain_ready = false;
        hipStream_t st = nullptr;
    };
    static constexpr int kFlowSlots = 4;
    FlowSlot flow[kFlowSlots];
    int flow_slots = 2;                  // slots in use
    bool flow_ready = false;
    unsigned long long flow_seq = 0;     // batches submitted so far
    latok::FlowHazards flow_held;        // memory ranges of the batches in flight, per slot (flow_hazards.h)
    const uint32_t* bench_cps_b = nullptr;   // latok_bench_set_second_input: what the odd steps of the flow measurements read
    const int64_t* bench_row_b = nullptr;
    hipEvent_t turn_event = nullptr;     // recorded behind the last kernel of every call (StreamTurn)
    hipStream_t turn_stream = nullptr;
    bool turn_stream_valid = false;
};
Ctx g_default;                       // latok_init / latok_shutdown
thread_local Ctx* tl_ctx = nullptr;  // latok_ctx_set_current; nullptr = g_default
Ctx* current_ctx() { return tl_ctx ? tl_ctx : &g_default; }

// HIP's current device is per host thread: whatever thread a call arrives on, allocations, events and launches of a
// context must happen with ITS device current (a worker thread starts on device 0).  Restores the caller's device.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const Ctx& c) {
        if (c.inited && hipGetDevice(&prev) == hipSuccess && prev != c.device) switched = hipSetDevice(c.device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define LATOK_ENTER()                        \
    Ctx& g = *current_ctx();                 \
    std::lock_guard<std::mutex> lk(g.mu);    \
    DeviceGuard device_guard_(g)

static size_t ws_summ_bytes(int64_t n_tiles) { return (size_t)(n_tiles > 0 ? n_tiles : 1) * 16; }
// one Fn64 + Hd64 pair per segment; plan_segments never makes a segment shorter than kWPB tiles (unless it is the
// only one), so n_segs <= t / kWPB + 1
static size_t ws_seg_bytes(int64_t n_tiles) {
    return ((size_t)(n_tiles > 0 ? n_tiles : 1) / latok::kWPB + 2) * (sizeof(latok::Fn64) + sizeof(latok::Hd64));
}
static size_t ws_first_bytes(int64_t n_tiles) { return (size_t)(n_tiles > 0 ? n_tiles : 1) * 8 + 8; }   // per-tile string index (stage 0)
int ensure_workspace(Ctx& g, int64_t n_tiles) {
    int rc;
    if ((rc = g.summ.ensure(ws_summ_bytes(n_tiles)))) return rc;
    if ((rc = g.seg_agg.ensure(ws_seg_bytes(n_tiles)))) return rc;
    if ((rc = g.fix_count.ensure(8))) return rc;
    if ((rc = g.tile_first.ensure(ws_first_bytes(n_tiles)))) return rc;
    return LATOK_OK;
}

int need_init(const Ctx& g) {
    if (!g.inited) return fail(LATOK_ERR_NOT_INIT, "latok_init() has not been called (no CPU fallback exists)");
    return LATOK_OK;
}

// The workspaces (tile summaries, bitmasks, ranks, staging) are shared by all calls of a context.  Host calls are serialised by its lock,
// but with caller streams the kernels of two calls could still overlap on the device: a call that runs on another
// stream than the previous one first waits (on the device) for that call's last kernel.
struct StreamTurn {
    Ctx& g;
    hipStream_t st;
    StreamTurn(Ctx& ctx, void* stream) : g(ctx), st(stream ? (hipStream_t)stream : ctx.stream) {
        if (g.inited && g.turn_event && g.turn_stream_valid && g.turn_stream != st)
            (void)hipStreamWaitEvent(st, g.turn_event, 0);
    }
    ~StreamTurn() {
        if (g.inited && g.turn_event) {
            (void)hipEventRecord(g.turn_event, st);
            g.turn_stream = st;
            g.turn_stream_valid = true;
        }
    }
};

// LATOK_ONE_SEGMENT=0 in the environment: small batches take the three-launch pipeline too (A/B, tests)
static bool one_segment_enabled() {
    static const bool on = [] { const char* e = getenv("LATOK_ONE_SEGMENT"); return !(e && e[0] == '0'); }();
    return on;
}

// enqueue the pipeline on device-resident data
int run_pipeline(Ctx& g, const uint32_t* d_cps, const int64_t* d_row, int64_t n_str, int64_t total, uint64_t* d_bits,
                 uint8_t* d_values, int mode, hipStream_t st, hipEvent_t tiles_begin = nullptr,
                 hipEvent_t tiles_end = nullptr, const int8_t* bm_a1 = nullptr, const int8_t* bm_a2 = nullptr,
                 const int* bm_flags = nullptr, uint64_t* d_space = nullptr, int64_t* d_tile_first = nullptr,
                 const uint8_t* d_u8 = nullptr, int unit_kind = 0, int stages = 7,   // stages: 1 = tile index, 2 = tiles, 4 = resolve
                 uint8_t* d_codes = nullptr,     // d_codes: also leave the rule code of every char (featurize)
                 latok::DoneSignal done = latok::DoneSignal{nullptr, 0, nullptr},   // completion word stored by the last launch
                 Ctx::FlowSlot* slot = nullptr,   // the workspace of a batch-flow slot instead of the context's own
                 uint64_t* d_lead = nullptr, uint16_t* d_lead_pref = nullptr, int64_t* d_lead_cnt = nullptr) {   // byte space: also leave the
                 // lead-byte mask, the leads of a tile before each word and the leads per tile (code-point results, mask_utf8_via_bytes)
    if (total <= 0 || n_str <= 0) return LATOK_OK;
    if (d_u8) {   // byte space: UTF-8 bytes in, positions are bytes; or (unit_kind 1 / 2) fixed-width code units, positions are chars
        if (mode != latok::kModeBits) return fail(LATOK_ERR_INVALID, "byte-space input supports the bitmask outputs only");
        if (((uintptr_t)d_u8 & 15) != 0)
            return fail(LATOK_ERR_INVALID, unit_kind ? "device code-unit pointer must be 16-byte aligned" : "device UTF-8 pointer must be 16-byte aligned");
        mode = unit_kind == 1 ? latok::kModeLatin1 : (unit_kind == 2 ? latok::kModeUcs2 : latok::kModeBytes);
    }
    // run-time rule tables (latok_set_rules): the same input form, rules interpreted from the kernel arguments
    if (g.rules_on && mode != latok::kModeBlockMask) mode = latok::mode_with_rules(mode);
    const int64_t n_tiles = (total + latok::kTile - 1) / latok::kTile;
    DevBuf& w_summ = slot ? slot->summ : g.summ;
    DevBuf& w_seg = slot ? slot->seg_agg : g.seg_agg;
    DevBuf& w_fix = slot ? slot->fix_count : g.fix_count;
    DevBuf& w_first = slot ? slot->tile_first : g.tile_first;
    int rc = slot ? LATOK_OK : ensure_workspace(g, n_tiles);   // (a flow slot is sized by flow_submit before anything is enqueued)
    if (rc) return rc;
    latok::SplitParams P;
    P.cps = d_cps;
    P.u8 = d_u8;
    P.row_off = d_row;
    P.n_str = n_str;
    P.total = total;
    P.n_tiles = n_tiles;
    // A batch of a FLOW leaves part of the chip to the other batch in flight: its persistent tile kernel is planned for 7/8 of
    // the CUs (4 free per XCD on MI355X), so the first workgroups of the NEXT batch's tile kernel start on the free CUs while
    // this one still runs, finish early and free CUs for the batch after it -- the start-up (table copy, first tile) and the
    // ragged end of every tile kernel then overlap another kernel's steady state instead of idling the chip.  Alone, the kernel
    // is as fast on 224 CUs as on 256 (it is memory bound); in the flow C2 goes from 96 to 85.5-87 us per batch.  Measured on
    // C2 (profiles/r03_ab_flow_cus.txt): fewer than 32 free CUs gain nothing (16: 97, 24: 96, 28: 94 us), 32: 85.5-87,
    // 48: 87-90, 64: 87-91, 128 (two kernels side by side on half the chip each): 89.  A segment's tiles go round-robin over
    // the workgroup's 12 waves, so a plan whose last round holds only a wave or two (216 CUs: 145 tiles = 12 rounds + 1 tile:
    // 93-96 us) wastes what the free CUs gain: of the candidate shares the first whose last round is at least half full is taken.
    int n_cu_eff = g.n_cu;
    if (slot && g.n_cu >= 64) {
        const int cand[3] = {g.n_cu * 7 / 8, g.n_cu * 13 / 16, g.n_cu * 3 / 4};
        int best = cand[0], best_fill = -1;
        for (int c = 0; c < 3; ++c) {
            int st_ = 0;
            int64_t ns_ = 0;
            latok::plan_segments(n_tiles, cand[c], &st_, &ns_);
            const int wpb = (d_u8 && unit_kind != 0 && !g.rules_on) ? 16 : latok::kWPB;   // waves per workgroup of the tile kernel (tile_wpb)
            const int fill = (st_ - 1) % wpb + 1;                  // waves busy in a segment's last round
            if (fill * 2 >= wpb) { best = cand[c]; break; }
            if (fill > best_fill) { best = cand[c]; best_fill = fill; }
        }
        n_cu_eff = best;
    }
    if (n_cu_eff < 8) n_cu_eff = g.n_cu < 8 ? g.n_cu : 8;
    latok::plan_segments(n_tiles, n_cu_eff, &P.seg_tiles, &P.n_segs);
    // a small UTF-32 batch: one segment, and (below) one launch for the three stages
    const bool one_launch = stages == 7 && !d_u8 && !tiles_begin && !tiles_end && n_tiles <= latok::kOneSegTiles &&
                            (mode == latok::kModeBits || mode == latok::kModeRules) && one_segment_enabled();
    if (one_launch) {
        P.seg_tiles = (int)(n_tiles < latok::kOneSegTiles ? latok::kOneSegTiles : n_tiles);
        P.n_segs = 1;
    }
    if ((size_t)P.n_segs * (sizeof(latok::Fn64) + sizeof(latok::Hd64)) > w_seg.cap || (size_t)n_tiles * 16 > w_summ.cap)
        return fail(LATOK_ERR_INVALID, "internal: workspace too small for %lld segments / %lld tiles", (long long)P.n_segs, (long long)n_tiles);
    if (d_codes && mode != latok::kModeBits && mode != latok::kModeRules)
        return fail(LATOK_ERR_INVALID, "internal: code bytes are written by the UTF-32 bitmask modes only");
    // rule codes (split code + NUM) when the rules are interpreted at run time or the code bytes are kept for featurize
    const uint8_t* tables = (const uint8_t*)((latok::mode_rules(mode) || d_codes) ? g.t1rule.p : g.t1.p);
    P.t1 = tables;
    P.t2 = tables + latok::kStage1Pad;
    if (latok::mode_base(mode) == latok::kModeBytes) {   // (byte space classifies through its own table, cut at 6 bits)
        P.t1 = (const uint8_t*)(latok::mode_rules(mode) ? g.tb6rule.p : g.tb6.p);
        P.t2 = P.t1 + latok::kB6Stage1Bytes;
    }
    if (latok::mode_rules(mode)) P.rules = g.rules;
    else memset(&P.rules, 0, sizeof(P.rules));
    P.bits_out = d_bits;
    P.values_out = d_values;
    P.space_out = d_space;
    P.lead_out = d_lead;
    P.lead_pref_out = d_lead_pref;
    P.lead_cnt_out = d_lead_cnt;
    P.codes_out = d_codes;
    if (!d_tile_first) d_tile_first = (int64_t*)w_first.p;   // the per-tile string index lives in the workspace
    P.tile_first = d_tile_first;
    P.summ = (int4*)w_summ.p;
    P.seg_fn = (latok::Fn64*)w_seg.p;
    P.seg_hd = (latok::Hd64*)((char*)w_seg.p + (size_t)P.n_segs * sizeof(latok::Fn64));
    P.fix_count = (int64_t*)w_fix.p;
    P.bm_a1 = bm_a1;
    P.bm_a2 = bm_a2;
    P.bm_flags = bm_flags;
    P.done = (stages & 4) ? done : latok::DoneSignal{nullptr, 0, nullptr};
    if (one_launch) {
        HIP_TRY(latok::launch_one_segment(P, mode, st));
        return LATOK_OK;
    }
    if (stages & 1) HIP_TRY(latok::launch_tile_index(P, st));   // (the kernel-timing loop of latok_bench_split_mask launches stage 1 alone)
    if (tiles_begin) HIP_TRY(hipEventRecord(tiles_begin, st));
    if (stages & 2) HIP_TRY(latok::launch_split_tiles(P, mode, g.n_cu, st, slot != nullptr));   // (the plan may leave CUs free; the grid never exceeds the chip)
    if (tiles_end) HIP_TRY(hipEventRecord(tiles_end, st));
    if (stages & 4) HIP_TRY(latok::launch_resolve_fix(P, mode, g.n_cu, st));
    return LATOK_OK;
}

int check_csr_host(const int64_t* row_off, int64_t n_str, int64_t* total_io) {
    if (n_str < 0) return fail(LATOK_ERR_INVALID, "n_str must be >= 0");
    if (n_str == 0) { *total_io = 0; return LATOK_OK; }
    if (!row_off) return fail(LATOK_ERR_INVALID, "row_off is NULL");
    if (row_off[0] != 0) return fail(LATOK_ERR_INVALID, "row_off[0] must be 0");
    for (int64_t s = 0; s < n_str; ++s)
        if (row_off[s + 1] < row_off[s]) return fail(LATOK_ERR_INVALID, "row_off must be non-decreasing");
    if (*total_io >= 0 && *total_io != row_off[n_str])
        return fail(LATOK_ERR_INVALID, "total_chars does not match row_off[n_str]");
    *total_io = row_off[n_str];
    return LATOK_OK;
}

int resolve_total_device(const int64_t* d_row, int64_t n_str, int64_t* total_io, hipStream_t st) {
    if (n_str < 0) return fail(LATOK_ERR_INVALID, "n_str must be >= 0");
    if (n_str == 0) { *total_io = 0; return LATOK_OK; }
    if (*total_io < 0) {
        HIP_TRY(hipMemcpyAsync(total_io, d_row + n_str, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LATOK_OK;
}

int split_common(Ctx& g, const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total, void* out, int mode,
                 int flags, void* stream) {
    int rc = need_init(g);
    if (rc) return rc;
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    if (flags & LATOK_DEVICE_PTRS) {
        if ((rc = resolve_total_device(row_off, n_str, &total, st))) return rc;
        if (total == 0) return LATOK_OK;
        if (!cps || !out) return fail(LATOK_ERR_INVALID, "NULL buffer");
        if (((uintptr_t)cps & 15) != 0) return fail(LATOK_ERR_INVALID, "device cps pointer must be 16-byte aligned");
        return run_pipeline(g, cps, row_off, n_str, total, mode == latok::kModeBits ? (uint64_t*)out : nullptr,
                            mode == latok::kModeValues ? (uint8_t*)out : nullptr, mode, st);
    }
    if ((rc = check_csr_host(row_off, n_str, &total))) return rc;
    if (total == 0) return LATOK_OK;
    if (!cps || !out) return fail(LATOK_ERR_INVALID, "NULL buffer");
    const size_t out_bytes = mode == latok::kModeBits ? (size_t)((total + 63) / 64) * 8 : (size_t)total;
    if (total <= kSmallChars && n_str <= kSmallStrings) {
        // small batch: stage in pinned mapped memory, kernels work on it in place, one synchronisation
        const size_t o_row = ((size_t)total * 4 + 15) & ~(size_t)15, o_out = o_row + (size_t)(n_str + 1) * 8;
        if ((rc = g.pin.ensure(o_out + ((out_bytes + 15) & ~(size_t)15)))) return rc;
        memcpy(g.pin.h, cps, (size_t)total * 4);
        memcpy((char*)g.pin.h + o_row, row_off, (size_t)(n_str + 1) * 8);
        char* d = (char*)g.pin.d;
        latok::DoneSignal done{nullptr, 0, nullptr};
        unsigned long long seq = 0;
        if (poll_completion()) {   // the last launch stores a completion word (see compact_common)
            if ((rc = g.pin_tot.ensure(64)) || (rc = g.done_ctr.ensure(64))) return rc;
            if (g.done_ctr.gen != g.done_ctr_seen) {
                g.done_ctr_seen = g.done_ctr.gen;
                HIP_TRY(hipMemsetAsync(g.done_ctr.p, 0, 64, st));
            }
            seq = ++g.small_seq;
            done = latok::DoneSignal{(unsigned long long*)g.pin_tot.d + 2, seq, (unsigned*)g.done_ctr.p};
        }
        rc = run_pipeline(g, (const uint32_t*)d, (const int64_t*)(d + o_row), n_str, total,
                          mode == latok::kModeBits ? (uint64_t*)(d + o_out) : nullptr,
                          mode == latok::kModeValues ? (uint8_t*)(d + o_out) : nullptr, mode, st, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, 0, 7, nullptr, done);
        if (rc) return rc;
        if (!(done.word && wait_completion_word((const unsigned long long*)g.pin_tot.h + 2, seq))) HIP_TRY(hipStreamSynchronize(st));
        memcpy(out, (char*)g.pin.h + o_out, out_bytes);
        return LATOK_OK;
    }
    if ((rc = g.h_cps.ensure((size_t)total * 4))) return rc;
    if ((rc = g.h_row.ensure((size_t)(n_str + 1) * 8))) return rc;
    if ((rc = g.h_out.ensure(out_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(g.h_cps.p, cps, (size_t)total * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g.h_row.p, row_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
    rc = run_pipeline(g, (const uint32_t*)g.h_cps.p, (const int64_t*)g.h_row.p, n_str, total,
                      mode == latok::kModeBits ? (uint64_t*)g.h_out.p : nullptr,
                      mode == latok::kModeValues ? (uint8_t*)g.h_out.p : nullptr, mode, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, g.h_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

}  // namespace

extern "C" {

int latok_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* latok_last_error(void) { return g_err.c_str(); }
const char* latok_version(void) { return "latok_hip 0.1 (gfx950)"; }

// ---- context lifecycle ---------------------------------------------------------------------------------------------
static void ctx_release(Ctx& g) {   // caller holds g.mu (or owns g exclusively)
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    g.pin.release();
    g.pin_tot.release();
    g.done_ctr.release();
    g.rules_on = false;
    for (DevBuf* b : {&g.t1, &g.t1rule, &g.tb6, &g.tb6rule, &g.t2code, &g.t2cls, &g.cw, &g.summ, &g.seg_agg, &g.fix_count, &g.h_cps, &g.h_row, &g.h_out,
                      &g.bits, &g.space, &g.kept, &g.wcnt, &g.wpref, &g.counts, &g.bases, &g.scan_tot, &g.tile_first, &g.u_bytes,
                      &g.u_boff, &g.u_cnt, &g.u_row, &g.u_pref, &g.u_lead, &g.u_bspace, &g.u_cpbits, &g.u_cpspace, &g.scalar, &g.h_aux, &g.chain, &g.chain_ctl, &g.codes})
        b->release();
    for (auto& e : g.ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (g.turn_event) (void)hipEventDestroy(g.turn_event);
    g.turn_event = nullptr;
    g.turn_stream_valid = false;
    g.chain_ready = false;
    g.chain_seen = g.chain_ctl_seen = 0;
    g.scan_epoch = 0;
    for (int i = 0; i < 2; ++i) {
        for (DevBuf* b : {&g.pipe_in[i], &g.pipe_row[i], &g.pipe_counts[i], &g.pipe_items[i], &g.pipe_feat[i]}) b->release();
        for (hipEvent_t* e : {&g.ev_in_ready[i], &g.ev_k_done[i], &g.ev_d2h_done[i]}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    }
    g.pipe_tot.release();
    for (auto& f : g.flow) {
        if (f.st) {
            (void)hipStreamSynchronize(f.st);
            (void)hipStreamDestroy(f.st);
        }
        f.st = nullptr;
        for (DevBuf* b : {&f.summ, &f.seg_agg, &f.fix_count, &f.tile_first, &f.bits, &f.space, &f.kept, &f.wcnt, &f.wpref, &f.bases,
                          &f.chain, &f.chain_ctl, &f.scalar, &f.codes, &f.widened})
            b->release();
        f.scan_epoch = f.chain_seen = f.chain_ctl_seen = 0;
        f.chain_ready = false;
    }
    g.flow_held.clear();
    g.bench_cps_b = nullptr;
    g.bench_row_b = nullptr;
    g.flow_ready = false;
    g.flow_seq = 0;
    if (g.s_h2d) (void)hipStreamDestroy(g.s_h2d);
    if (g.s_d2h) (void)hipStreamDestroy(g.s_d2h);
    g.s_h2d = g.s_d2h = nullptr;
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    g.inited = false;
    g.device = -1;
}

// Byte space's class table (split_code.h: LK_B6_*): the generated two-stage table (blocks of 128 code points, uint8 block ids) cut
// again at 64 code points -- stage 1 by cp >> 6 = every UTF-8 byte of the char but the last, as the uint16 OFFSET of a 64-entry
// stage-2 block, which the last byte's payload indexes.  blob = [stage 1, kB6Stage1Bytes | stage 2, kB6Stage2Bytes]; code[] =
// kClassCode or kClassRuleCode.  What the kernel relies on beyond the lookup itself is checked here: ASCII is blocks 0 and 1,
// back to back (its bytes are looked up without stage 1), and the last stage-1 entry (cp >= 0x110000) is a block of zeros.
static int build_byte_tables(const unsigned char* code, std::vector<uint8_t>& blob) {
    blob.assign(latok::kB6TablesBytes, 0);
    uint16_t* s1 = reinterpret_cast<uint16_t*>(blob.data());
    uint8_t* s2 = blob.data() + latok::kB6Stage1Bytes;
    int n_blocks = 0;
    for (int hi = 0; hi < latok::kB6Stage1Len; ++hi) {
        uint8_t v[64];
        const uint32_t cp0 = (uint32_t)hi << LK_B6_SHIFT;
        const uint32_t b7 = kStage1[cp0 >= 0x110000u ? LATOK_TBL_STAGE1_LEN - 1 : (cp0 >> LATOK_TBL_SHIFT)];
        for (int j = 0; j < 64; ++j) {
            const uint32_t in = cp0 >= 0x110000u ? 0u : ((cp0 + j) & ((1u << LATOK_TBL_SHIFT) - 1u));
            v[j] = code[kStage2[(b7 << LATOK_TBL_SHIFT) | in]];
        }
        int b = 0;
        while (b < n_blocks && memcmp(s2 + 64 * b, v, 64) != 0) ++b;
        if (b == n_blocks) {
            if (n_blocks == latok::kB6MaxBlocks) return fail(LATOK_ERR_INVALID, "internal: more than %d distinct 64-char class blocks", latok::kB6MaxBlocks);
            memcpy(s2 + 64 * n_blocks++, v, 64);
        }
        s1[hi] = (uint16_t)(64 * b);
    }
    bool last_zero = true;
    for (int j = 0; j < 64; ++j) last_zero = last_zero && s2[s1[latok::kB6Stage1Len - 1] + j] == 0;
    if (s1[0] != 0 || s1[1] != 64 || !last_zero) return fail(LATOK_ERR_INVALID, "internal: byte-space class table layout");
    return LATOK_OK;
}

/* test hook (not part of the ABI; needs no device): the byte-space class table as the kernels get it -- rule_codes 0: split
 * codes, 1: rule codes.  Writes kB6TablesBytes into out (cap_bytes >= that) and returns the offset of stage 2 inside it. */
extern "C" int latok_debug_byte_tables(int rule_codes, void* out, int64_t cap_bytes) {
    std::vector<uint8_t> blob;
    const int rc = build_byte_tables(rule_codes ? kClassRuleCode : kClassCode, blob);
    if (rc) return rc;
    if (!out || cap_bytes < (int64_t)blob.size()) return fail(LATOK_ERR_INVALID, "need %zu bytes", blob.size());
    memcpy(out, blob.data(), blob.size());
    return latok::kB6Stage1Bytes;
}

static int ctx_init_body(Ctx& g, int device) {
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    g.n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    for (auto& e : g.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&g.turn_event, hipEventDisableTiming));
    g.turn_stream_valid = false;

    // tables: stage-1 (padded), stage-2 as split codes (fused path) and as class ids + class words (parse matrix)
    std::vector<uint8_t> t1(latok::kStage1Pad, kStage1[LATOK_TBL_STAGE1_LEN - 1]);
    memcpy(t1.data(), kStage1, LATOK_TBL_STAGE1_LEN);
    std::vector<uint8_t> t2code(latok::kStage2Len);
    for (int i = 0; i < latok::kStage2Len; ++i) t2code[i] = kClassCode[kStage2[i]];
    int rc;
    std::vector<uint8_t> t2rule(latok::kStage2Len);
    for (int i = 0; i < latok::kStage2Len; ++i) t2rule[i] = kClassRuleCode[kStage2[i]];
    if ((rc = g.t1.ensure(t1.size() + t2code.size()))) return rc;   // [stage1 | stage2 codes], contiguous like in LDS
    if ((rc = g.t1rule.ensure(t1.size() + t2rule.size()))) return rc;   // same with rule codes (runtime rule tables)
    HIP_TRY(hipMemcpy(g.t1rule.p, t1.data(), t1.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy((char*)g.t1rule.p + t1.size(), t2rule.data(), t2rule.size(), hipMemcpyHostToDevice));
    {
        std::vector<uint8_t> blob;
        if ((rc = build_byte_tables(kClassCode, blob))) return rc;
        if ((rc = g.tb6.ensure(blob.size()))) return rc;
        HIP_TRY(hipMemcpy(g.tb6.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
        if ((rc = build_byte_tables(kClassRuleCode, blob))) return rc;
        if ((rc = g.tb6rule.ensure(blob.size()))) return rc;
        HIP_TRY(hipMemcpy(g.tb6rule.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    if ((rc = g.t2cls.ensure(sizeof(kStage2)))) return rc;
    if ((rc = g.cw.ensure(sizeof(kClassWord)))) return rc;
    if ((rc = g.scalar.ensure(64))) return rc;
    HIP_TRY(hipMemcpy(g.t1.p, t1.data(), t1.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy((char*)g.t1.p + t1.size(), t2code.data(), t2code.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.t2cls.p, kStage2, sizeof(kStage2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.cw.p, kClassWord, sizeof(kClassWord), hipMemcpyHostToDevice));
    g.device = device;
    g.inited = true;
    return LATOK_OK;
}

// bind a context to `device`: stream, events, Unicode tables.  The caller's current HIP device is left as it was.
static int ctx_init(Ctx& g, int device) {   // caller holds g.mu
    int n = latok_device_count();
    if (n <= 0) return fail(LATOK_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(LATOK_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    int prev = -1;
    (void)hipGetDevice(&prev);
    const int rc = ctx_init_body(g, device);
    if (rc) {   // a partly built context leaks nothing
        const std::string msg = g_err;
        ctx_release(g);
        g_err = msg;
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

int latok_init(int device) {
    Ctx& g = g_default;
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.inited) {
        if (g.device == device) return LATOK_OK;
        return fail(LATOK_ERR_INVALID, "the default context is already on device %d; use latok_ctx_create for other devices", g.device);
    }
    return ctx_init(g, device);
}

int latok_shutdown(void) {
    Ctx& g = g_default;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.inited) return LATOK_OK;
    DeviceGuard dg(g);
    ctx_release(g);
    return LATOK_OK;
}

int latok_ctx_create(int device, latok_ctx** ctx_out) {
    if (!ctx_out) return fail(LATOK_ERR_INVALID, "ctx_out is NULL");
    *ctx_out = nullptr;
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return fail(LATOK_ERR_NOMEM, "out of host memory");
    int rc;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        rc = ctx_init(*c, device);
    }
    if (rc) {
        delete c;
        return rc;
    }
    *ctx_out = reinterpret_cast<latok_ctx*>(c);
    return LATOK_OK;
}

int latok_ctx_destroy(latok_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return LATOK_OK;
    if (c == &g_default) return fail(LATOK_ERR_INVALID, "the default context is destroyed by latok_shutdown()");
    if (tl_ctx == c) tl_ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->mu);   // waits for a call that is still running on it
        DeviceGuard dg(*c);
        ctx_release(*c);
    }
    delete c;
    return LATOK_OK;
}

int latok_ctx_set_current(latok_ctx* ctx) {
    tl_ctx = reinterpret_cast<Ctx*>(ctx);   // NULL = the default context
    if (tl_ctx == &g_default) tl_ctx = nullptr;
    return LATOK_OK;
}

latok_ctx* latok_ctx_get_current(void) { return reinterpret_cast<latok_ctx*>(tl_ctx); }

int latok_ctx_device(latok_ctx* ctx) {
    const Ctx* c = ctx ? reinterpret_cast<const Ctx*>(ctx) : &g_default;
    return c->inited ? c->device : -1;
}

int latok_reserve(int64_t max_chars, int64_t max_strings) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (max_chars < 0 || max_strings < 0) return fail(LATOK_ERR_INVALID, "negative size");
    return ensure_workspace(g, (max_chars + latok::kTile - 1) / latok::kTile);
}

// one rule table: row-major int8 [rows x cols] as build_combo_matrix returns it -> per-row column sets
static int pack_rule_table(const char* name, const int8_t* idx, int rows, int cols, uint32_t* row_sets) {
    if (rows < 0 || rows > LK_MAX_RULE_ROWS)
        return fail(LATOK_ERR_INVALID, "%s: %d rows (0..%d supported)", name, rows, LK_MAX_RULE_ROWS);
    if (rows > 0 && (!idx || cols < 1)) return fail(LATOK_ERR_INVALID, "%s: NULL table or no columns", name);
    for (int r = 0; r < rows; ++r) {
        uint32_t set = 0;
        for (int c = 0; c < cols; ++c) {
            const int v = idx[r * cols + c];
            if (v == -1) {
                // the reference seeds a row's product from its FIRST entry (latok.c:329-338); a row that starts with
                // the -1 pad would multiply into the previous row's product there -- refuse instead of guessing
                if (c == 0) return fail(LATOK_ERR_INVALID, "%s: row %d starts with -1", name, r);
                continue;
            }
            if (v < 0 || v >= LK_N_FEATURES)
                return fail(LATOK_ERR_INVALID, "%s: feature id %d in row %d is outside 0..%d", name, v, r, LK_N_FEATURES - 1);
            set |= 1u << v;
        }
        row_sets[r] = set;
    }
    return LATOK_OK;
}

int latok_set_rules(const int8_t* c_split, int split_rows, int split_cols, const int8_t* c_mask, int mask_rows,
                    int mask_cols, const int8_t* c_sym, int sym_rows, int sym_cols) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    lk_rule_tables R;
    memset(&R, 0, sizeof(R));
    if ((rc = pack_rule_table("C_SPLIT", c_split, split_rows, split_cols, R.row[0]))) return rc;
    if ((rc = pack_rule_table("C_MASK", c_mask, mask_rows, mask_cols, R.row[1]))) return rc;
    if ((rc = pack_rule_table("C_SYM", c_sym, sym_rows, sym_cols, R.row[2]))) return rc;
    R.n_rows[0] = split_rows;
    R.n_rows[1] = mask_rows;
    R.n_rows[2] = sym_rows;
    g.rules = R;
    g.rules_on = true;
    return LATOK_OK;
}

int latok_reset_rules(void) {
    LATOK_ENTER();
    g.rules_on = false;
    return LATOK_OK;
}

int latok_rules_active(void) {
    LATOK_ENTER();
    return g.rules_on ? 1 : 0;
}

int latok_split_mask_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                           uint64_t* mask_bits_out, int flags, void* stream) {
    LATOK_ENTER();
    return split_common(g, cps, row_off, n_str, total_chars, mask_bits_out, latok::kModeBits, flags, stream);
}

int latok_split_values_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                             uint8_t* values_out, int flags, void* stream) {
    LATOK_ENTER();
    return split_common(g, cps, row_off, n_str, total_chars, values_out, latok::kModeValues, flags, stream);
}

// UTF-8 ingest: decode a CSR batch of UTF-8 strings into the library's device buffers (g.h_cps = packed code points,
// g.u_row = code-point row offsets).  Inputs are host or device pointers per `dev`.  One blocking 8-byte read.
// With `bytes_route`: when the batch has no continuation byte at all (pure ASCII, the common case) byte positions ARE
// code-point positions, so nothing is decoded; *bytes_route = the device pointers for the byte-space tile kernel, whose
// results are then valid in code-point units as they are.
struct BytesRoute {
    const uint8_t* d_u8 = nullptr;
    const int64_t* d_boff = nullptr;
};
static int decode_utf8_to_workspace(Ctx& g, const uint8_t* u8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                    bool dev, hipStream_t st, int64_t* total_cps_out, BytesRoute* bytes_route = nullptr) {
    int rc;
    *total_cps_out = 0;
    if (bytes_route) *bytes_route = BytesRoute();
    const uint8_t* d_u8 = u8;
    const int64_t* d_boff = byte_off;
    if (dev) {
        if ((rc = resolve_total_device(byte_off, n_str, &total_bytes, st))) return rc;
    } else {
        if ((rc = check_csr_host(byte_off, n_str, &total_bytes))) return rc;
    }
    if (n_str == 0) return LATOK_OK;
    if (total_bytes > 0 && !u8) return fail(LATOK_ERR_INVALID, "utf8 buffer is NULL");
    if (!dev) {
        if ((rc = g.u_bytes.ensure((size_t)total_bytes + 16))) return rc;
        if ((rc = g.u_boff.ensure((size_t)(n_str + 1) * 8))) return rc;
        if (total_bytes > 0) HIP_TRY(hipMemcpyAsync(g.u_bytes.p, u8, (size_t)total_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(g.u_boff.p, byte_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
        d_u8 = (const uint8_t*)g.u_bytes.p;
        d_boff = (const int64_t*)g.u_boff.p;
    }
    const int64_t n_blocks = latok::utf8_blocks(total_bytes);
    if ((rc = g.u_cnt.ensure((size_t)n_blocks * 16 + 16))) return rc;               // block counts | block bases
    if ((rc = g.u_row.ensure((size_t)(n_str + 1) * 8))) return rc;
    if ((rc = g.u_pref.ensure((size_t)(n_blocks * 256) * 2 + 16))) return rc;       // u16 prefix per 16-byte chunk
    if ((rc = g.scan_tot.ensure((size_t)latok::scan_blocks(n_blocks) * 8))) return rc;
    int64_t* d_cnt = (int64_t*)g.u_cnt.p;
    int64_t* d_base = d_cnt + n_blocks;
    HIP_TRY(latok::launch_utf8_block_counts(d_u8, total_bytes, d_cnt, st));
    if ((rc = g.pin_tot.ensure(64))) return rc;
    HIP_TRY(latok::launch_exclusive_scan(d_cnt, n_blocks, d_base, (int64_t*)g.scalar.p, (int64_t*)g.scan_tot.p, st,
                                         (int64_t*)g.pin_tot.d));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t total_cps = *(volatile const int64_t*)g.pin_tot.h;
    if (bytes_route && total_cps == total_bytes && ((uintptr_t)d_u8 & 15) == 0) {
        bytes_route->d_u8 = d_u8;
        bytes_route->d_boff = d_boff;
        *total_cps_out = total_cps;
        return LATOK_OK;
    }
    if ((rc = g.h_cps.ensure((size_t)total_cps * 4 + 16))) return rc;
    HIP_TRY(latok::launch_utf8_decode(d_u8, total_bytes, d_boff, n_str, d_base, (uint16_t*)g.u_pref.p, total_cps,
                                      (uint32_t*)g.h_cps.p, (int64_t*)g.u_row.p, st));
    *total_cps_out = total_cps;
    return LATOK_OK;
}

// featurize: per-token column sums on the tile grid (split_kernels.hip: k_features_tiles)
static int enqueue_features(Ctx& g, const uint8_t* d_codes, const int64_t* d_row, int64_t n_str, int64_t total, const uint64_t* d_bits,
                            const uint64_t* d_space, const uint64_t* d_kept, const int64_t* d_rank, const int64_t* d_tile_cnt,
                            const uint16_t* d_pref, const int64_t* d_tile_first, void* d_spans4, int8_t* d_feat, bool out32,
                            const int64_t* d_n_tokens, int64_t cap, hipStream_t st, latok::DoneSignal done) {
    latok::FeatParams F;
    F.codes = d_codes;
    F.row_off = d_row;
    F.n_str = n_str;
    F.total = total;
    F.n_tiles = (total + latok::kTile - 1) / latok::kTile;
    F.bits = d_bits;
    F.kept = d_kept;
    F.tile_rank = d_rank;
    F.tile_cnt = d_tile_cnt;
    F.word_pref = d_pref;
    F.tile_first = d_tile_first;
    F.space = d_space;
    F.features = d_feat;
    F.spans4 = d_spans4;
    F.out32 = out32;
    F.n_tokens_dev = d_n_tokens;
    F.cap = cap;
    F.done = done;
    HIP_TRY(latok::launch_features_tiles(F, g.n_cu, st));
    return LATOK_OK;
}

// the single-pass scan of k_word_counts_scan keeps its state between launches: entries carry an epoch, so the array is
// cleared only when it is (re)allocated or when the 18-bit epoch wraps
struct ScanState {   // the chained scan's look-back words + epoch of one workspace set (the context's own, or a flow slot's)
    DevBuf& chain;
    DevBuf& chain_ctl;
    unsigned& scan_epoch;
    unsigned& chain_seen;
    unsigned& chain_ctl_seen;
    bool& chain_ready;
};
static ScanState scan_state(Ctx& g, Ctx::FlowSlot* slot) {
    if (slot) return ScanState{slot->chain, slot->chain_ctl, slot->scan_epoch, slot->chain_seen, slot->chain_ctl_seen, slot->chain_ready};
    return ScanState{g.chain, g.chain_ctl, g.scan_epoch, g.chain_seen, g.chain_ctl_seen, g.chain_ready};
}
static int next_scan_epoch(ScanState c, int64_t n_blocks, hipStream_t st, unsigned* epoch_out) {
    int rc;
    if ((rc = c.chain.ensure((size_t)n_blocks * 8 + 64))) return rc;
    if ((rc = c.chain_ctl.ensure(64))) return rc;
    c.scan_epoch = (c.scan_epoch + 1) & 0x3FFFFu;
    // (the state array may have been re-allocated by this call or by an earlier reserve: fresh memory holds anything)
    if (c.chain.gen != c.chain_seen || c.chain_ctl.gen != c.chain_ctl_seen || c.scan_epoch == 0 || !c.chain_ready) {
        c.chain_seen = c.chain.gen;
        c.chain_ctl_seen = c.chain_ctl.gen;
        HIP_TRY(hipMemsetAsync(c.chain.p, 0, c.chain.cap, st));
        HIP_TRY(hipMemsetAsync(c.chain_ctl.p, 0, 64, st));
        c.scan_epoch = 1;
        c.chain_ready = true;
    }
    *epoch_out = c.scan_epoch;
    return LATOK_OK;
}

// Device-side core of the compaction entry points, on one device-resident (chunk of a) batch: per-string boundary
// offsets (spans = false), token spans, or token spans + feature sums (feats).  Launch sequence:
//   tile index -> tiles -> resolve            the two bitmasks (boundaries, SPACE)
//   k_word_counts, k_scan_chained             items per word / tile, tile ranks and the total
//   k_counts_scatter (or k_string_counts + k_features_tiles)     per-string counts + the records, written only if the
//                                             total fits `cap`
// Nothing is synchronised here: the total and the int32-overflow flag land in the pinned pair p_tot[0..1] (h_tot = the
// host's view of the same two words) when the stream gets there.
static int enqueue_compaction_dev(Ctx& g, bool spans, bool feats, bool o32, const uint32_t* d_cps, const uint8_t* d_u8, int unit_kind,
                                  const int64_t* d_row, int64_t n_str, int64_t total, void* d_counts, void* d_items, int8_t* d_feat,
                                  int64_t cap, int64_t* p_tot, volatile int64_t* h_tot, hipStream_t st,
                                  latok::DoneSignal done = latok::DoneSignal{nullptr, 0, nullptr},
                                  Ctx::FlowSlot* slot = nullptr,     // slot: the workspaces of a batch-flow slot (offsets / spans only)
                                  const uint64_t* pre_bits = nullptr, const uint64_t* pre_space = nullptr) {   // the two bitmasks are
                                  // already there (code-point masks packed from byte space: cp_masks_via_bytes): only the string index is launched
    int rc;
    DevBuf& w_bits = slot ? slot->bits : g.bits;
    DevBuf& w_space = slot ? slot->space : g.space;
    DevBuf& w_kept = slot ? slot->kept : g.kept;
    DevBuf& w_wcnt = slot ? slot->wcnt : g.wcnt;
    DevBuf& w_bases = slot ? slot->bases : g.bases;
    DevBuf& w_wpref = slot ? slot->wpref : g.wpref;
    DevBuf& w_first = slot ? slot->tile_first : g.tile_first;
    DevBuf& w_scalar = slot ? slot->scalar : g.scalar;
    DevBuf& w_codes = slot ? slot->codes : g.codes;
    DevBuf& w_widened = slot ? slot->widened : g.h_cps;
    const ScanState sc = scan_state(g, slot);
    if (d_u8 && unit_kind && feats) {
        // featurize re-reads the code points: widen once, on the device
        if (((uintptr_t)d_u8 & (size_t)(unit_kind - 1)) != 0) return fail(LATOK_ERR_INVALID, "misaligned code units");
        if ((rc = w_widened.ensure((size_t)total * 4 + 16))) return rc;
        HIP_TRY(latok::launch_widen_units(d_u8, unit_kind, total, (uint32_t*)w_widened.p, st));
        d_cps = (const uint32_t*)w_widened.p;
        d_u8 = nullptr;
    }
    const int64_t words = (total + 63) / 64;
    if ((rc = w_bits.ensure((size_t)words * 8 + 8))) return rc;
    if (spans && (rc = w_space.ensure((size_t)words * 8 + 8))) return rc;
    if (spans && (rc = w_kept.ensure((size_t)words * 8 + 8))) return rc;
    const int64_t c_tiles = (words + 63) / 64;
    if ((rc = w_wcnt.ensure((size_t)c_tiles * 8 + 8))) return rc;       // items per tile
    if ((rc = w_bases.ensure((size_t)c_tiles * 8 + 8))) return rc;      // rank of each tile's first item
    if ((rc = w_wpref.ensure((size_t)words * 2 + 8))) return rc;        // items of the tile before each word
    if ((rc = w_scalar.ensure(64))) return rc;
    unsigned epoch = 0;
    if ((rc = next_scan_epoch(sc, latok::count_blocks(words), st, &epoch))) return rc;
    uint64_t* d_bits = pre_bits ? const_cast<uint64_t*>(pre_bits) : (uint64_t*)w_bits.p;
    uint64_t* d_space = spans ? (pre_bits ? const_cast<uint64_t*>(pre_space) : (uint64_t*)w_space.p) : nullptr;
    uint64_t* d_kept = spans ? (uint64_t*)w_kept.p : nullptr;
    const uint64_t* d_item_mask = spans ? d_kept : d_bits;
    int64_t* d_rank = (int64_t*)w_bases.p;
    int64_t* d_tcnt = (int64_t*)w_wcnt.p;
    uint16_t* d_pref = (uint16_t*)w_wpref.p;
    if ((rc = w_first.ensure((size_t)((total + latok::kTile - 1) / latok::kTile) * 8 + 8))) return rc;
    int64_t* d_tile_first = (int64_t*)w_first.p;
    uint8_t* d_codes = nullptr;
    if (feats) {   // the tile kernel leaves the rule code of every char: 1 B/char for k_features_tiles instead of 4 B/char + tables
        const size_t code_bytes = (size_t)total + latok::kTile + 256;   // read (never used) up to a tile behind the last char
        if ((rc = w_codes.ensure(code_bytes))) return rc;
        d_codes = (uint8_t*)w_codes.p;
        const size_t tail0 = (size_t)total & ~(size_t)(latok::kTile - 1);
        HIP_TRY(hipMemsetAsync(d_codes + tail0, 0, code_bytes - tail0, st));
    }
    if ((rc = run_pipeline(g, d_cps, d_row, n_str, total, d_bits, nullptr, latok::kModeBits, st, nullptr, nullptr, nullptr,
                           nullptr, nullptr, d_space, d_tile_first, d_u8, unit_kind, pre_bits ? 1 : 7, d_codes,
                           latok::DoneSignal{nullptr, 0, nullptr}, slot)))
        return rc;
    if (h_tot) {   // pinned pair of the context's own calls: cleared by the host
        h_tot[0] = 0;
        h_tot[1] = 0;
    } else {       // a flow's result words live wherever the caller put them: cleared on the stream
        HIP_TRY(hipMemsetAsync(p_tot, 0, 16, st));
    }
    int64_t* d_total = (int64_t*)w_scalar.p;
    int* d_err = (int*)(p_tot + 1);
    HIP_TRY(latok::launch_word_counts_scan(spans, d_bits, d_space, words, total, d_kept, d_tcnt, d_pref, d_rank,
                                           (unsigned long long*)sc.chain.p, (unsigned*)sc.chain_ctl.p, epoch, d_total, p_tot, d_err + 1, st));   // (the scan's own flag: the upper half of the pinned word)
    if (feats) {   // spans and sums come from one kernel
        HIP_TRY(latok::launch_string_counts(o32, d_item_mask, d_rank, d_pref, d_row, n_str, total, d_total, d_counts, d_err, st));
        return enqueue_features(g, d_codes, d_row, n_str, total, d_bits, d_space, d_kept, d_rank, d_tcnt, d_pref, d_tile_first, d_items,
                                d_feat, o32, d_total, cap, st, done);
    }
    HIP_TRY(latok::launch_counts_scatter(spans ? 1 : 0, o32, d_bits, d_space, d_item_mask, d_rank, d_tcnt, d_pref, words, total, d_row,
                                         n_str, d_tile_first, d_items, d_total, cap, d_counts, d_err, st, done));
    return LATOK_OK;
}

// Large host-pointer batches: a chunked pipeline over three streams.  The batch is cut into chunks of whole strings
// (~8 M chars); chunk c + 1 is on its way up the bus (copy stream) while chunk c runs its kernels (the context's stream)
// and the records of chunk c - 1 go down (second copy stream); inputs and outputs are double-buffered on the device, a
// chunk's records land in a buffer sized for the worst case (one item per char), so nothing waits for a total before it
// is enqueued.  Per chunk the host waits once (for its total) before it can place the chunk's records behind the
// previous ones in the caller's arrays.  With pinned host arrays (latok_host_alloc) both copy directions run at bus
// speed concurrently; pageable arrays work too (the runtime stages them).
// (test hook: LATOK_PIPE_CHUNK_CHARS in the environment shrinks the chunks, so that a test can push hundreds of chunks
// through the double buffers with batches the oracle checks in seconds)
static int64_t pipe_chunk_chars() {
    static const int64_t v = [] {
        const char* e = getenv("LATOK_PIPE_CHUNK_CHARS");
        const long long x = e ? atoll(e) : 0;
        return (int64_t)(x >= 64 ? x : (8ll << 20));
    }();
    return v;
}
#define kPipeChunkChars (pipe_chunk_chars())
#define kPipeMinChars (2 * pipe_chunk_chars())

static int ensure_pipe(Ctx& g) {
    if (g.s_h2d) return LATOK_OK;
    HIP_TRY(hipStreamCreateWithFlags(&g.s_h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g.s_d2h, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&g.ev_in_ready[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_k_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_d2h_done[i], hipEventDisableTiming));
    }
    return LATOK_OK;
}

static int compact_host_pipelined_body(Ctx& g, bool spans, bool feats, bool o32, const void* data, size_t unit_bytes, int unit_kind,
                                       bool as_u8, const int64_t* row_off, int64_t n_str, int64_t total, void* counts_out,
                                       void* items_out, int64_t items_cap, int64_t* n_items_out, int8_t* features_out,
                                       hipStream_t st);

static int compact_host_pipelined(Ctx& g, bool spans, bool feats, bool o32, const void* data, size_t unit_bytes, int unit_kind,
                                  bool as_u8, const int64_t* row_off, int64_t n_str, int64_t total, void* counts_out,
                                  void* items_out, int64_t items_cap, int64_t* n_items_out, int8_t* features_out, hipStream_t st) {
    const int rc = compact_host_pipelined_body(g, spans, feats, o32, data, unit_bytes, unit_kind, as_u8, row_off, n_str, total,
                                               counts_out, items_out, items_cap, n_items_out, features_out, st);
    if (rc != LATOK_OK && g.s_h2d) {
        // a failure in the middle leaves copies in flight that read and write the CALLER's arrays: drain them before the
        // error is returned (the message of the failure is kept)
        const std::string msg = g_err;
        (void)hipStreamSynchronize(g.s_h2d);
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(g.s_d2h);
        g_err = msg;
    }
    return rc;
}

static int compact_host_pipelined_body(Ctx& g, bool spans, bool feats, bool o32, const void* data, size_t unit_bytes, int unit_kind,
                                       bool as_u8, const int64_t* row_off, int64_t n_str, int64_t total, void* counts_out,
                                       void* items_out, int64_t items_cap, int64_t* n_items_out, int8_t* features_out,
                                       hipStream_t st) {
    int rc;
    if ((rc = ensure_pipe(g))) return rc;
    const size_t elt = o32 ? 4 : 8;
    const size_t item_bytes = (feats ? 4 : (spans ? 2 : 1)) * elt;
    // chunk boundaries (string ids): cut where the cumulative char count passes multiples of the chunk size
    std::vector<int64_t> cut(1, 0);
    int64_t max_chars = 0, max_strs = 0;
    while (cut.back() < n_str) {
        const int64_t s0 = cut.back();
        const int64_t* e = std::upper_bound(row_off + s0 + 1, row_off + n_str + 1, row_off[s0] + kPipeChunkChars);
        int64_t s1 = (e - row_off) - 1;            // last string that still ends within the chunk size
        if (s1 <= s0) s1 = s0 + 1;                 // a single string longer than a chunk is a chunk of its own
        cut.push_back(s1);
        max_chars = std::max(max_chars, row_off[s1] - row_off[s0]);
        max_strs = std::max(max_strs, s1 - s0);
    }
    const int n_chunks = (int)cut.size() - 1;
    for (int i = 0; i < 2; ++i) {
        if ((rc = g.pipe_in[i].ensure((size_t)max_chars * unit_bytes + 64))) return rc;
        if ((rc = g.pipe_row[i].ensure((size_t)(max_strs + 1) * 8))) return rc;
        if ((rc = g.pipe_counts[i].ensure((size_t)max_strs * elt + 16))) return rc;
        if ((rc = g.pipe_items[i].ensure((size_t)max_chars * item_bytes + 64))) return rc;      // at most one item per char
        if (feats && (rc = g.pipe_feat[i].ensure((size_t)max_chars * LATOK_FEATURE_COUNT + 64))) return rc;
    }
    if ((rc = g.pipe_tot.ensure(8 * 16))) return rc;
    {   // size every workspace for the largest chunk now: growing one later would free it under a chunk that is still running
        const size_t w = (size_t)((max_chars + 63) / 64), t = (size_t)((max_chars + latok::kTile - 1) / latok::kTile);
        if ((rc = ensure_workspace(g, (int64_t)t))) return rc;
        if ((rc = g.bits.ensure(w * 8 + 8))) return rc;
        if (spans && ((rc = g.space.ensure(w * 8 + 8)) || (rc = g.kept.ensure(w * 8 + 8)))) return rc;
        if ((rc = g.wcnt.ensure(((w + 63) / 64) * 8 + 8)) || (rc = g.bases.ensure(((w + 63) / 64) * 8 + 8))) return rc;
        if ((rc = g.wpref.ensure(w * 2 + 8))) return rc;
        if ((rc = g.tile_first.ensure(t * 8 + 8))) return rc;
        if (feats && (rc = g.codes.ensure((size_t)max_chars + latok::kTile + 256))) return rc;
        if (as_u8 && unit_kind && feats && (rc = g.h_cps.ensure((size_t)max_chars * 4 + 16))) return rc;
        if ((rc = g.chain.ensure((size_t)latok::count_blocks((int64_t)w) * 8 + 64))) return rc;
    }
    int64_t running = 0;
    bool overflow = false, too_long = false;
    std::vector<int64_t> n_of(n_chunks, 0);
    auto finish = [&](int c) -> int {   // chunk c's kernels are enqueued: wait for its total, send its records down
        const int slot = c & 1;
        HIP_TRY(hipEventSynchronize(g.ev_k_done[slot]));
        volatile int64_t* h = (volatile int64_t*)g.pipe_tot.h + 2 * (c & 7);
        const int64_t n = h[0];
        if (h[1] >> 32) { g.chain_ready = false; return fail(LATOK_ERR_HIP, "internal: the scan's look-back state was corrupt (the call is safe to repeat)"); }
        if (h[1] & 0xFFFFFFFFll) too_long = true;
        n_of[c] = n;
        const int64_t s0 = cut[c], ns = cut[c + 1] - cut[c];
        HIP_TRY(hipStreamWaitEvent(g.s_d2h, g.ev_k_done[slot], 0));
        HIP_TRY(hipMemcpyAsync((char*)counts_out + (size_t)s0 * elt, g.pipe_counts[slot].p, (size_t)ns * elt, hipMemcpyDeviceToHost, g.s_d2h));
        if (running + n > items_cap || (n > 0 && !items_out)) overflow = true;
        if (!overflow && n > 0) {
            HIP_TRY(hipMemcpyAsync((char*)items_out + (size_t)running * item_bytes, g.pipe_items[slot].p, (size_t)n * item_bytes,
                                   hipMemcpyDeviceToHost, g.s_d2h));
            if (feats)
                HIP_TRY(hipMemcpyAsync(features_out + (size_t)running * LATOK_FEATURE_COUNT, g.pipe_feat[slot].p,
                                       (size_t)n * LATOK_FEATURE_COUNT, hipMemcpyDeviceToHost, g.s_d2h));
        }
        HIP_TRY(hipEventRecord(g.ev_d2h_done[slot], g.s_d2h));
        running += n;
        return LATOK_OK;
    };
    auto upload = [&](int c) -> int {   // chunk c's code units and row offsets (its device buffers are free once chunk c - 2 is computed)
        const int slot = c & 1;
        const int64_t s0 = cut[c], ns = cut[c + 1] - s0, c0 = row_off[s0], nc = row_off[cut[c + 1]] - c0;
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(g.s_h2d, g.ev_k_done[slot], 0));
        if (nc > 0)
            HIP_TRY(hipMemcpyAsync(g.pipe_in[slot].p, (const char*)data + (size_t)c0 * unit_bytes, (size_t)nc * unit_bytes,
                                   hipMemcpyHostToDevice, g.s_h2d));
        HIP_TRY(hipMemcpyAsync(g.pipe_row[slot].p, row_off + s0, (size_t)(ns + 1) * 8, hipMemcpyHostToDevice, g.s_h2d));
        HIP_TRY(hipEventRecord(g.ev_in_ready[slot], g.s_h2d));
        return LATOK_OK;
    };
    if ((rc = upload(0))) return rc;
    for (int c = 0; c < n_chunks; ++c) {
        const int slot = c & 1;
        const int64_t s0 = cut[c], ns = cut[c + 1] - s0, c0 = row_off[s0], nc = row_off[cut[c + 1]] - c0;
        // compute: behind the upload, and behind the download of chunk c - 2 (it read the same output buffers)
        HIP_TRY(hipStreamWaitEvent(st, g.ev_in_ready[slot], 0));
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(st, g.ev_d2h_done[slot], 0));
        HIP_TRY(latok::launch_rebase_rows((int64_t*)g.pipe_row[slot].p, ns + 1, c0, st));
        volatile int64_t* h = (volatile int64_t*)g.pipe_tot.h + 2 * (c & 7);
        int64_t* p = (int64_t*)g.pipe_tot.d + 2 * (c & 7);
        if (nc > 0) {
            rc = enqueue_compaction_dev(g, spans, feats, o32, as_u8 ? nullptr : (const uint32_t*)g.pipe_in[slot].p,
                                        as_u8 ? (const uint8_t*)g.pipe_in[slot].p : nullptr, unit_kind, (const int64_t*)g.pipe_row[slot].p,
                                        ns, nc, g.pipe_counts[slot].p, g.pipe_items[slot].p, (int8_t*)g.pipe_feat[slot].p, nc, p, h, st);
            if (rc) return rc;
        } else {   // only empty strings in this chunk
            h[0] = 0;
            h[1] = 0;
            HIP_TRY(hipMemsetAsync(g.pipe_counts[slot].p, 0, (size_t)ns * elt, st));
        }
        HIP_TRY(hipEventRecord(g.ev_k_done[slot], st));
        // the next upload is queued before the host waits for anything: the copy stream never runs dry
        // (chunk c + 1 shares its buffers with chunk c - 1, whose kernels were enqueued an iteration ago)
        if (c + 1 < n_chunks && (rc = upload(c + 1))) return rc;
        if (c >= 1 && (rc = finish(c - 1))) return rc;
    }
    if ((rc = finish(n_chunks - 1))) return rc;
    HIP_TRY(hipStreamSynchronize(g.s_d2h));
    HIP_TRY(hipStreamSynchronize(st));
    *n_items_out = running;
    if (too_long) return fail(LATOK_ERR_INVALID, "a string is too long for LATOK_OUT_INT32; use the 64-bit form");
    if (running > items_cap) return fail(LATOK_ERR_INVALID, "output capacity too small: need %lld", (long long)running);
    if (running > 0 && !items_out) return fail(LATOK_ERR_INVALID, "output buffer is NULL");
    return LATOK_OK;
}

// Shared body of the compaction entry points: argument checks, staging of host-pointer batches, one synchronisation.
// Small UTF-8 host batches (one string per call is the usual C caller): decoded by the host with the device decoder's rule
// -- one code point per lead byte, as many continuation bytes as the lead announces (utf8_decode.h) -- into UTF-32, so
// that the call takes the pinned small-batch path of the code-point form; byte-space results are mapped back through the
// byte position of every char.  Only when every string is structurally well-formed (each lead followed by exactly its
// continuation bytes inside the string, no stray continuation byte): the two device paths define what malformed input
// means, and they keep doing so.  cps / cp_row / bytepos: code points, code-point row offsets, byte position of every
// char (+ one entry for the end).
static bool host_decode_small(const uint8_t* u8, const int64_t* boff, int64_t n_str, std::vector<uint32_t>& cps,
                              std::vector<int64_t>& cp_row, std::vector<int64_t>& bytepos) {
    const int64_t total = boff[n_str];
    cps.clear(); bytepos.clear();
    cps.reserve((size_t)total); bytepos.reserve((size_t)total + 1);
    cp_row.assign((size_t)n_str + 1, 0);
    for (int64_t s = 0; s < n_str; ++s) {
        const int64_t end = boff[s + 1];
        for (int64_t i = boff[s]; i < end;) {
            const uint32_t b0 = u8[i];
            int extra = 0;
            uint32_t cp = b0;
            if (b0 >= 0x80u) {
                if (b0 < 0xC0u) return false;                       // stray continuation byte
                if (b0 >= 0xF0u) { cp = b0 & 0x07u; extra = 3; }
                else if (b0 >= 0xE0u) { cp = b0 & 0x0Fu; extra = 2; }
                else { cp = b0 & 0x1Fu; extra = 1; }
                if (i + extra >= end) return false;                 // truncated at the string's end
                for (int j = 1; j <= extra; ++j) {
                    const uint32_t b = u8[i + j];
                    if ((b & 0xC0u) != 0x80u) return false;         // truncated sequence
                    cp = (cp << 6) | (b & 0x3Fu);
                }
            }
            cps.push_back(cp);
            bytepos.push_back(i);
            i += 1 + extra;
        }
        cp_row[(size_t)s + 1] = (int64_t)cps.size();
    }
    bytepos.push_back(total);
    return true;
}

static int cp_masks_via_bytes(Ctx& g, const uint8_t* d_u8, const int64_t* d_boff, int64_t n_str, int64_t total_bytes, uint64_t* d_out,
                              uint64_t* d_out_space, int64_t cap_words, int64_t* d_cp_row, hipStream_t st, int64_t* total_cps_out,
                              int* fallback_out);
static int utf8_on_device(Ctx& g, const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes, bool dev, hipStream_t st,
                          const uint8_t** d_u8, const int64_t** d_boff);
static int compact_common(Ctx& g, bool spans, const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total, void* counts_out,
                          void* items_out, int64_t items_cap, int64_t* n_items_out, int flags, void* stream, int8_t* features_out,
                          const uint8_t* utf8, bool byte_space, int unit_kind);

// A small UTF-8 host batch through the UTF-32 small-batch path (host_decode_small); *taken = false: not this route (too
// large, or malformed), nothing was done.  Byte space: a char position becomes the byte position of that char, relative
// to its string.
static int compact_small_utf8_host(Ctx& g, bool spans, const uint8_t* utf8, bool byte_space, const int64_t* byte_off, int64_t n_str,
                                   int64_t total_bytes, void* counts_out, void* items_out, int64_t items_cap, int64_t* n_items_out,
                                   int flags, void* stream, int8_t* features_out, bool* taken) {
    *taken = false;
    int64_t tb = total_bytes;
    if (check_csr_host(byte_off, n_str, &tb) != LATOK_OK || tb <= 0 || tb > kSmallChars) return LATOK_OK;
    if (!host_decode_small(utf8, byte_off, n_str, g.hd_cps, g.hd_row, g.hd_pos)) return LATOK_OK;
    *taken = true;
    const int rc = compact_common(g, spans, g.hd_cps.data(), g.hd_row.data(), n_str, (int64_t)g.hd_cps.size(), counts_out, items_out,
                                  items_cap, n_items_out, flags, stream, features_out, nullptr, false, 0);
    if (rc != LATOK_OK || !byte_space || !items_out) return rc;
    const bool o32 = (flags & LATOK_OUT_INT32) != 0;
    const int64_t per = spans ? 2 : 1;
    int64_t k = 0;
    for (int64_t s = 0; s < n_str; ++s) {
        const int64_t n = o32 ? (int64_t)((const int32_t*)counts_out)[s] : ((const int64_t*)counts_out)[s];
        const int64_t c0 = g.hd_row[(size_t)s], b0 = byte_off[s];
        for (int64_t j = 0; j < n * per; ++j, ++k) {
            if (o32) { int32_t* v = (int32_t*)items_out + k; *v = (int32_t)(g.hd_pos[(size_t)(c0 + *v)] - b0); }
            else { int64_t* v = (int64_t*)items_out + k; *v = g.hd_pos[(size_t)(c0 + *v)] - b0; }
        }
    }
    return LATOK_OK;
}

static int compact_common(Ctx& g, bool spans, const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total,
                          void* counts_out, void* items_out, int64_t items_cap, int64_t* n_items_out, int flags,
                          void* stream, int8_t* features_out = nullptr, const uint8_t* utf8 = nullptr,
                          bool byte_space = false, int unit_kind = 0) {
    const bool feats = features_out != nullptr;
    const bool o32 = (flags & LATOK_OUT_INT32) != 0;
    int rc = need_init(g);
    if (rc) return rc;
    if (!n_items_out) return fail(LATOK_ERR_INVALID, "the total-count output pointer is NULL");
    *n_items_out = 0;
    if (items_cap < 0) return fail(LATOK_ERR_INVALID, "negative capacity");
    const bool dev = (flags & LATOK_DEVICE_PTRS) != 0;
    if (o32 && !dev && row_off && n_str > 0) {   // before anything is staged (device-resident row offsets: the kernel checks)
        for (int64_t s = 0; s < n_str; ++s)
            if (row_off[s + 1] - row_off[s] > 0x7FFFFFFFll)
                return fail(LATOK_ERR_INVALID, "string %lld is too long for LATOK_OUT_INT32; use the 64-bit form", (long long)s);
    }
    if (!dev && utf8 && unit_kind == 0 && row_off && n_str > 0 && n_str <= kSmallStrings && counts_out) {
        bool taken = false;
        rc = compact_small_utf8_host(g, spans, utf8, byte_space, row_off, n_str, total, counts_out, items_out, items_cap, n_items_out,
                                     flags, stream, features_out, &taken);
        if (taken) return rc;
    }
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const size_t elt = o32 ? 4 : 8;                                   // width of counts and of every record field
    const size_t item_bytes = (feats ? 4 : (spans ? 2 : 1)) * elt;
    const uint32_t* d_cps = cps;
    const int64_t* d_row = row_off;
    const uint8_t* d_u8 = nullptr;   // byte space: the tile kernel reads the UTF-8 bytes itself, results are byte offsets
    const uint64_t *pre_bits = nullptr, *pre_space = nullptr;   // code-point masks packed from byte space (UTF-8 in code-point units)
    const size_t unit_bytes = (utf8 && byte_space) ? (unit_kind ? (size_t)unit_kind : 1) : 4;
    if (!dev && !(utf8 && !byte_space)) {
        // host pointers, fixed-width units or UTF-8 in byte space: checked here; large batches take the chunked pipeline
        if ((rc = check_csr_host(row_off, n_str, &total))) return rc;
        if (n_str == 0) return LATOK_OK;
        if (!counts_out) return fail(LATOK_ERR_INVALID, "counts_out is NULL");
        if (total >= kPipeMinChars)
            return compact_host_pipelined(g, spans, feats, o32, utf8 ? (const void*)utf8 : (const void*)cps, unit_bytes, unit_kind,
                                          utf8 != nullptr, row_off, n_str, total, counts_out, items_out, items_cap, n_items_out,
                                          features_out, st);
    }
    // Small host batches of narrow units (the C-extension caller of INTEGRATION.md section C hands over ONE str per call in
    // its PEP 393 kind) are widened to UTF-32 by the host straight into the pinned area and take the small-batch path below:
    // positions are chars either way, so the results are the same, and the call costs one launch instead of staged copies
    // (kind 1, one 105-char string: 110 -> 17 us).
    bool widen = false;
    if (!dev && utf8 && byte_space && n_str > 0 && total > 0 && total <= kSmallChars && n_str <= kSmallStrings) {
        widen = unit_kind == 1 || unit_kind == 2;   // (small UTF-8 batches were decoded by the host above)
    }
    if (widen) {
        d_cps = nullptr;   // (set below, with the pinned area)
    } else if (utf8 && byte_space) {
        // UTF-8 in byte space, or (unit_kind 1 / 2) PEP 393 code units: `total` positions of unit_bytes each
        if (dev) {
            if ((rc = resolve_total_device(row_off, n_str, &total, st))) return rc;
            d_u8 = utf8;
        } else if (n_str > 0) {
            if ((rc = g.u_bytes.ensure((size_t)total * unit_bytes + 16))) return rc;
            if ((rc = g.u_boff.ensure((size_t)(n_str + 1) * 8))) return rc;
            if (total > 0) HIP_TRY(hipMemcpyAsync(g.u_bytes.p, utf8, (size_t)total * unit_bytes, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(g.u_boff.p, row_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
            d_u8 = (const uint8_t*)g.u_bytes.p;
            d_row = (const int64_t*)g.u_boff.p;
        }
        d_cps = nullptr;
    } else if (utf8) {   // row_off = byte offsets, total = bytes; results are in code-point units
        int64_t tb = total;
        if (dev) {
            if ((rc = resolve_total_device(row_off, n_str, &tb, st))) return rc;
        } else if ((rc = check_csr_host(row_off, n_str, &tb))) {
            return rc;
        }
        bool done_via_bytes = false;
        if (!feats && n_str > 0 && tb > kSmallChars && (!dev || ((uintptr_t)utf8 & 15) == 0)) {
            // large batches: the byte-space kernel + the masks packed at the lead bytes (no UTF-32 copy of the batch); the compaction
            // then runs on the code-point masks
            const uint8_t* b8;
            const int64_t* boff;
            if ((rc = utf8_on_device(g, utf8, row_off, n_str, tb, dev, st, &b8, &boff))) return rc;
            const int64_t words_b = (tb + 63) / 64;
            if ((rc = g.u_cpbits.ensure((size_t)words_b * 8 + 8)) || (spans && (rc = g.u_cpspace.ensure((size_t)words_b * 8 + 8))) ||
                (rc = g.u_row.ensure((size_t)(n_str + 1) * 8)))
                return rc;
            int fallback = 0;
            int64_t total_cps = 0;
            if ((rc = cp_masks_via_bytes(g, b8, boff, n_str, tb, (uint64_t*)g.u_cpbits.p, spans ? (uint64_t*)g.u_cpspace.p : nullptr, words_b,
                                         (int64_t*)g.u_row.p, st, &total_cps, &fallback)))
                return rc;
            if (!fallback) {
                pre_bits = (const uint64_t*)g.u_cpbits.p;
                pre_space = spans ? (const uint64_t*)g.u_cpspace.p : nullptr;
                d_cps = nullptr;
                d_row = (const int64_t*)g.u_row.p;
                total = total_cps;
                done_via_bytes = true;
            }
        }
        if (!done_via_bytes) {   // small batches, featurize (re-reads code points), malformed input: decode on the device first
            BytesRoute br;
            if ((rc = decode_utf8_to_workspace(g, utf8, row_off, n_str, total, dev, st, &total, feats ? nullptr : &br))) return rc;
            if (br.d_u8) {   // no multi-byte char in the batch: byte space == code-point space, skip the decode
                d_u8 = br.d_u8;
                d_row = br.d_boff;
                d_cps = nullptr;
            } else {
                d_cps = (const uint32_t*)g.h_cps.p;
                d_row = (const int64_t*)g.u_row.p;
            }
        }
    } else if (dev) {
        if ((rc = resolve_total_device(row_off, n_str, &total, st))) return rc;
        if (total > 0 && ((uintptr_t)cps & 15) != 0)
            return fail(LATOK_ERR_INVALID, "device cps pointer must be 16-byte aligned");
    }
    if (n_str == 0) return LATOK_OK;
    if (!counts_out) return fail(LATOK_ERR_INVALID, "counts_out is NULL");
    if (total == 0) {   // only empty strings: all counts are 0
        if (dev) HIP_TRY(hipMemsetAsync(counts_out, 0, (size_t)n_str * elt, st));
        else memset(counts_out, 0, (size_t)n_str * elt);
        return LATOK_OK;
    }
    // small host batch: inputs and every output live in pinned mapped memory; nothing is copied by the runtime and the
    // call synchronises once (a string of ~100 chars: ~110 us of blocking copies otherwise)
    const bool small = !dev && (!utf8 || widen) && total <= kSmallChars && n_str <= kSmallStrings;
    size_t po_row = 0, po_counts = 0, po_items = 0, po_feat = 0;
    if (small) {
        po_row = ((size_t)total * 4 + 15) & ~(size_t)15;
        po_counts = po_row + (((size_t)(n_str + 1) * 8 + 15) & ~(size_t)15);
        po_items = po_counts + (((size_t)n_str * elt + 15) & ~(size_t)15);
        po_feat = po_items + (size_t)total * item_bytes;      // at most one item per char
        if ((rc = g.pin.ensure(po_feat + (feats ? (size_t)total * LATOK_FEATURE_COUNT : 0) + 64))) return rc;
        if (!widen) {
            memcpy(g.pin.h, cps, (size_t)total * 4);
        } else if (unit_kind == 2) {
            uint32_t* w = (uint32_t*)g.pin.h;
            for (int64_t i = 0; i < total; ++i) { uint16_t u; memcpy(&u, utf8 + 2 * i, 2); w[i] = u; }
        } else {
            uint32_t* w = (uint32_t*)g.pin.h;
            for (int64_t i = 0; i < total; ++i) w[i] = utf8[i];
        }
        memcpy((char*)g.pin.h + po_row, row_off, (size_t)(n_str + 1) * 8);
        d_cps = (const uint32_t*)g.pin.d;
        d_row = (const int64_t*)((char*)g.pin.d + po_row);
    } else if (!dev && !utf8) {
        if ((rc = g.h_cps.ensure((size_t)total * 4 + 16))) return rc;
        if ((rc = g.h_row.ensure((size_t)(n_str + 1) * 8))) return rc;
        HIP_TRY(hipMemcpyAsync(g.h_cps.p, cps, (size_t)total * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(g.h_row.p, row_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
        d_cps = (const uint32_t*)g.h_cps.p;
        d_row = (const int64_t*)g.h_row.p;
    }
    if ((rc = g.pin_tot.ensure(64))) return rc;
    volatile int64_t* h_tot = (volatile int64_t*)g.pin_tot.h;
    int64_t* p_tot = (int64_t*)g.pin_tot.d;
    // where the records go: the caller's device buffers, the pinned area, or (host pointers, mid-size batch) device staging
    // sized for the worst case of one item per char
    void* d_counts = counts_out;
    void* d_items = items_out;
    int8_t* d_feat = features_out;
    int64_t cap = items_out ? items_cap : 0;
    if (small) {
        d_counts = (char*)g.pin.d + po_counts;
        d_items = (char*)g.pin.d + po_items;
        d_feat = (int8_t*)((char*)g.pin.d + po_feat);
        cap = total;
    } else if (!dev) {
        if ((rc = g.counts.ensure((size_t)n_str * 8))) return rc;
        if ((rc = g.h_out.ensure((size_t)total * item_bytes))) return rc;
        if (feats && (rc = g.h_aux.ensure((size_t)total * LATOK_FEATURE_COUNT))) return rc;
        d_counts = g.counts.p;
        d_items = g.h_out.p;
        d_feat = (int8_t*)g.h_aux.p;
        cap = total;
    }
    bool polled = false;
    if (small && total <= latok::kTile) {
        // at most one tile (tokenize(text) / featurize(text): one string per call): one single-wave launch does everything
        // and stores a completion word the host polls
        latok::SplitParams P;
        memset(&P, 0, sizeof(P));
        P.cps = d_cps;
        P.row_off = d_row;
        P.n_str = n_str;
        P.total = total;
        P.n_tiles = 1;
        const uint8_t* tables = (const uint8_t*)((g.rules_on || feats) ? g.t1rule.p : g.t1.p);   // featurize needs rule codes
        P.t1 = tables;
        P.t2 = tables + latok::kStage1Pad;
        if (g.rules_on) P.rules = g.rules;
        h_tot[0] = 0;
        h_tot[1] = 0;
        const unsigned long long seq = ++g.small_seq;
        unsigned long long* d_done = poll_completion() ? (unsigned long long*)(p_tot + 2) : nullptr;
        HIP_TRY(latok::launch_small_batch(P, g.rules_on, feats ? 2 : (spans ? 1 : 0), o32, d_counts, d_items, d_feat, p_tot, d_done, seq, st));
        polled = d_done && wait_completion_word((const unsigned long long*)(h_tot + 2), seq);
    } else {
        // several tiles in pinned memory: the last kernel's workgroups count themselves in and the last one stores the
        // completion word (latok::DoneSignal)
        latok::DoneSignal done{nullptr, 0, nullptr};
        unsigned long long seq = 0;
        if (small && poll_completion()) {
            if ((rc = g.done_ctr.ensure(64))) return rc;
            if (g.done_ctr.gen != g.done_ctr_seen) {
                g.done_ctr_seen = g.done_ctr.gen;
                HIP_TRY(hipMemsetAsync(g.done_ctr.p, 0, 64, st));
            }
            seq = ++g.small_seq;
            done = latok::DoneSignal{(unsigned long long*)(p_tot + 2), seq, (unsigned*)g.done_ctr.p};
        }
        if ((rc = enqueue_compaction_dev(g, spans, feats, o32, d_cps, d_u8, unit_kind, d_row, n_str, total, d_counts, d_items, d_feat,
                                         cap, p_tot, h_tot, st, done, nullptr, pre_bits, pre_space)))
            return rc;
        polled = done.word && wait_completion_word((const unsigned long long*)(h_tot + 2), seq);
    }
    // the one synchronisation: total and flag are in pinned memory now (a polled small batch has seen its completion
    // word, which the kernel stores after everything else; the launch itself retires on the stream a moment later)
    if (!polled) HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_items = h_tot[0];
    *n_items_out = n_items;
    if (h_tot[1] >> 32) { g.chain_ready = false; return fail(LATOK_ERR_HIP, "internal: the scan's look-back state was corrupt (the call is safe to repeat)"); }
    if (h_tot[1] & 0xFFFFFFFFll) return fail(LATOK_ERR_INVALID, "a string is too long for LATOK_OUT_INT32; use the 64-bit form");
    const bool fits = n_items <= items_cap && (n_items == 0 || items_out);
    if (small) {
        memcpy(counts_out, (char*)g.pin.h + po_counts, (size_t)n_str * elt);
        if (fits && n_items > 0) {
            memcpy(items_out, (char*)g.pin.h + po_items, (size_t)n_items * item_bytes);
            if (feats) memcpy(features_out, (char*)g.pin.h + po_feat, (size_t)n_items * LATOK_FEATURE_COUNT);
        }
    } else if (!dev) {
        HIP_TRY(hipMemcpyAsync(counts_out, g.counts.p, (size_t)n_str * elt, hipMemcpyDeviceToHost, st));
        if (fits && n_items > 0) {
            HIP_TRY(hipMemcpyAsync(items_out, g.h_out.p, (size_t)n_items * item_bytes, hipMemcpyDeviceToHost, st));
            if (feats) HIP_TRY(hipMemcpyAsync(features_out, g.h_aux.p, (size_t)n_items * LATOK_FEATURE_COUNT, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (n_items > items_cap) return fail(LATOK_ERR_INVALID, "output capacity too small: need %lld", (long long)n_items);
    if (n_items > 0 && !items_out) return fail(LATOK_ERR_INVALID, "output buffer is NULL");
    return LATOK_OK;
}

int latok_split_offsets_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total,
                              int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap, int64_t* n_offsets_out,
                              int flags, void* stream) {
    LATOK_ENTER();
    return compact_common(g, false, cps, row_off, n_str, total, counts_out, offsets_out, offsets_cap, n_offsets_out, flags,
                          stream);
}

int latok_token_spans_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total,
                            int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out, int flags,
                            void* stream) {
    LATOK_ENTER();
    return compact_common(g, true, cps, row_off, n_str, total, counts_out, spans_out, spans_cap, n_tokens_out, flags, stream);
}

int latok_utf8_decode_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                            uint32_t* cps_out, int64_t cps_cap, int64_t* cp_row_off_out, int64_t* total_cps_out, int flags,
                            void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (!total_cps_out) return fail(LATOK_ERR_INVALID, "total_cps_out is NULL");
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const bool dev = (flags & LATOK_DEVICE_PTRS) != 0;
    int64_t total_cps = 0;
    if ((rc = decode_utf8_to_workspace(g, utf8, byte_off, n_str, total_bytes, dev, st, &total_cps))) return rc;
    *total_cps_out = total_cps;
    if (n_str == 0) return LATOK_OK;
    if (total_cps > cps_cap) return fail(LATOK_ERR_INVALID, "cps_cap too small: need %lld", (long long)total_cps);
    if (!cp_row_off_out || (total_cps > 0 && !cps_out)) return fail(LATOK_ERR_INVALID, "NULL output buffer");
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (total_cps > 0) HIP_TRY(hipMemcpyAsync(cps_out, g.h_cps.p, (size_t)total_cps * 4, kind, st));
    HIP_TRY(hipMemcpyAsync(cp_row_off_out, g.u_row.p, (size_t)(n_str + 1) * 8, kind, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

// Code-point results of a UTF-8 batch WITHOUT a UTF-32 copy of it (the reference reads code points, latok.c:53-55,79; a UTF-8
// caller has bytes): the byte-space tile kernel on the bytes, which also leaves the lead-byte mask and the lead counts per word
// and per tile; one scan of the tile counts (k_scan_chained); then k_lead_compress packs the boundary bits at lead bytes (and,
// for token spans, the SPACE plane) and turns the byte offsets into code-point offsets.  HBM traffic: the bytes once + ~5 bits per
// byte of masks and ranks, against 1 + 4 + 4 bytes per char through the staged decoder.  Everything is on the device; nothing
// waits for the host between the launches; the code-point total, the capacity check and the malformed-input flag are read
// after one synchronisation.  *fallback_out = 1: the batch holds a continuation byte that the byte-space model and the
// decoder treat differently (malformed UTF-8): the caller takes the decoder.
//   d_out / d_out_space: where the packed masks go (cap_words words each; d_out_space NULL: boundaries only), d_cp_row [n_str + 1]
static int cp_masks_via_bytes(Ctx& g, const uint8_t* d_u8, const int64_t* d_boff, int64_t n_str, int64_t total_bytes, uint64_t* d_out,
                              uint64_t* d_out_space, int64_t cap_words, int64_t* d_cp_row, hipStream_t st, int64_t* total_cps_out,
                              int* fallback_out) {
    int rc;
    *fallback_out = 0;
    const int64_t words_b = (total_bytes + 63) / 64, c_tiles = (words_b + 63) / 64;
    if ((rc = g.bits.ensure((size_t)words_b * 8 + 8)) || (rc = g.u_lead.ensure((size_t)words_b * 8 + 8)) ||
        (d_out_space && (rc = g.u_bspace.ensure((size_t)words_b * 8 + 8))) ||
        (rc = g.wcnt.ensure((size_t)c_tiles * 8 + 8)) || (rc = g.bases.ensure((size_t)c_tiles * 8 + 8)) ||
        (rc = g.wpref.ensure((size_t)words_b * 2 + 8)) || (rc = g.scalar.ensure(64)) || (rc = g.pin_tot.ensure(64)))
        return rc;
    unsigned epoch = 0;
    if ((rc = next_scan_epoch(scan_state(g, nullptr), latok::count_blocks(words_b), st, &epoch))) return rc;
    uint64_t* d_bmask = (uint64_t*)g.bits.p;
    uint64_t* d_lead = (uint64_t*)g.u_lead.p;
    uint64_t* d_bspace = d_out_space ? (uint64_t*)g.u_bspace.p : nullptr;
    if ((rc = run_pipeline(g, nullptr, d_boff, n_str, total_bytes, d_bmask, nullptr, latok::kModeBits, st, nullptr, nullptr, nullptr, nullptr,
                           nullptr, d_bspace, nullptr, d_u8, 0, 7, nullptr, latok::DoneSignal{nullptr, 0, nullptr}, nullptr, d_lead,
                           (uint16_t*)g.wpref.p, (int64_t*)g.wcnt.p)))
        return rc;
    volatile int64_t* h_tot = (volatile int64_t*)g.pin_tot.h;
    int64_t* p_tot = (int64_t*)g.pin_tot.d;
    h_tot[0] = 0;
    h_tot[1] = 0;
    h_tot[3] = 0;
    int* d_err = (int*)(p_tot + 1);
    HIP_TRY(latok::launch_tile_scan((const int64_t*)g.wcnt.p, c_tiles, (int64_t*)g.bases.p, (unsigned long long*)g.chain.p, (unsigned*)g.chain_ctl.p,
                                    epoch, (int64_t*)g.scalar.p, p_tot, d_err + 1, st));
    HIP_TRY(latok::launch_lead_compress(d_bmask, d_bspace, d_lead, (const int64_t*)g.bases.p, (const int64_t*)g.wcnt.p,
                                        (const uint16_t*)g.wpref.p, words_b, total_bytes, d_boff, n_str, (const int64_t*)g.scalar.p, d_out,
                                        d_out_space, cap_words, d_cp_row, (int*)(p_tot + 3), st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_tot[1] != 0) { g.chain_ready = false; return fail(LATOK_ERR_HIP, "internal: the chained scan did not complete (flag %lld)", (long long)h_tot[1]); }
    if (h_tot[3] != 0) { *fallback_out = 1; return LATOK_OK; }
    *total_cps_out = h_tot[0];
    return LATOK_OK;
}
// bytes + byte offsets on the device, from the caller's pointers (uploaded when they are host pointers)
static int utf8_on_device(Ctx& g, const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes, bool dev, hipStream_t st,
                          const uint8_t** d_u8, const int64_t** d_boff) {
    int rc;
    *d_u8 = utf8;
    *d_boff = byte_off;
    if (dev) return LATOK_OK;
    if ((rc = g.u_bytes.ensure((size_t)total_bytes + 16))) return rc;
    if ((rc = g.u_boff.ensure((size_t)(n_str + 1) * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(g.u_bytes.p, utf8, (size_t)total_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g.u_boff.p, byte_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
    *d_u8 = (const uint8_t*)g.u_bytes.p;
    *d_boff = (const int64_t*)g.u_boff.p;
    return LATOK_OK;
}
static int mask_utf8_via_bytes(Ctx& g, const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes, bool dev,
                               uint64_t* mask_bits_out, int64_t mask_cap_words, int64_t* cp_row_off_out, int64_t* total_cps_out,
                               hipStream_t st, int* fallback_out) {
    int rc;
    const uint8_t* d_u8;
    const int64_t* d_boff;
    if ((rc = utf8_on_device(g, utf8, byte_off, n_str, total_bytes, dev, st, &d_u8, &d_boff))) return rc;
    const int64_t words_b = (total_bytes + 63) / 64;
    const int64_t out_words = mask_cap_words < words_b ? mask_cap_words : words_b;   // (a batch has at most one char per byte)
    uint64_t* d_out = mask_bits_out;
    int64_t* d_cp_row = cp_row_off_out;
    if (!dev) {
        if ((rc = g.h_out.ensure((size_t)out_words * 8 + 8)) || (rc = g.u_row.ensure((size_t)(n_str + 1) * 8))) return rc;
        d_out = (uint64_t*)g.h_out.p;
        d_cp_row = (int64_t*)g.u_row.p;
    }
    int64_t total_cps = 0;
    if ((rc = cp_masks_via_bytes(g, d_u8, d_boff, n_str, total_bytes, d_out, nullptr, out_words, d_cp_row, st, &total_cps, fallback_out))) return rc;
    if (*fallback_out) return LATOK_OK;
    *total_cps_out = total_cps;
    const int64_t words = (total_cps + 63) / 64;
    if (words > mask_cap_words) return fail(LATOK_ERR_INVALID, "mask_cap_words too small: need %lld", (long long)words);
    if (!dev) {
        if (words > 0) HIP_TRY(hipMemcpyAsync(mask_bits_out, d_out, (size_t)words * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(cp_row_off_out, d_cp_row, (size_t)(n_str + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LATOK_OK;
}

int latok_split_mask_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                uint64_t* mask_bits_out, int64_t mask_cap_words, int64_t* cp_row_off_out,
                                int64_t* total_cps_out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (!total_cps_out) return fail(LATOK_ERR_INVALID, "total_cps_out is NULL");
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const bool dev = (flags & LATOK_DEVICE_PTRS) != 0;
    // large batches: byte space + compaction of the mask (no UTF-32 copy)
    {
        int64_t tb = total_bytes;
        if (dev) {
            if ((rc = resolve_total_device(byte_off, n_str, &tb, st))) return rc;
        } else if ((rc = check_csr_host(byte_off, n_str, &tb))) {
            return rc;
        }
        if (n_str > 0 && tb > kSmallChars && utf8 && cp_row_off_out && mask_bits_out && mask_cap_words >= 0 &&
            (!dev || ((uintptr_t)utf8 & 15) == 0)) {
            int fallback = 0;
            rc = mask_utf8_via_bytes(g, utf8, byte_off, n_str, tb, dev, mask_bits_out, mask_cap_words, cp_row_off_out, total_cps_out, st,
                                     &fallback);
            if (rc || !fallback) return rc;
        }
    }
    int64_t total = 0;
    BytesRoute br;
    if ((rc = decode_utf8_to_workspace(g, utf8, byte_off, n_str, total_bytes, dev, st, &total, &br))) return rc;
    *total_cps_out = total;
    if (n_str == 0) return LATOK_OK;
    const int64_t words = (total + 63) / 64;
    if (words > mask_cap_words) return fail(LATOK_ERR_INVALID, "mask_cap_words too small: need %lld", (long long)words);
    if (!cp_row_off_out || (words > 0 && !mask_bits_out)) return fail(LATOK_ERR_INVALID, "NULL output buffer");
    uint64_t* d_bits = mask_bits_out;
    if (!dev) {
        if ((rc = g.h_out.ensure((size_t)words * 8 + 8))) return rc;
        d_bits = (uint64_t*)g.h_out.p;
    }
    if (br.d_u8) {   // no multi-byte char: the byte-space kernel on the bytes, code-point offsets = byte offsets
        if ((rc = run_pipeline(g, nullptr, br.d_boff, n_str, total, d_bits, nullptr, latok::kModeBits, st, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, br.d_u8)))
            return rc;
    } else if ((rc = run_pipeline(g, (const uint32_t*)g.h_cps.p, (const int64_t*)g.u_row.p, n_str, total, d_bits, nullptr,
                                  latok::kModeBits, st))) {
        return rc;
    }
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (!dev && words > 0) HIP_TRY(hipMemcpyAsync(mask_bits_out, d_bits, (size_t)words * 8, kind, st));
    HIP_TRY(hipMemcpyAsync(cp_row_off_out, br.d_u8 ? (const void*)br.d_boff : (const void*)g.u_row.p, (size_t)(n_str + 1) * 8,
                           kind, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

int latok_split_offsets_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                   int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap,
                                   int64_t* n_offsets_out, int flags, void* stream) {
    LATOK_ENTER();
    static const uint8_t empty = 0;
    return compact_common(g, false, nullptr, byte_off, n_str, total_bytes, counts_out, offsets_out, offsets_cap,
                          n_offsets_out, flags, stream, nullptr, utf8 ? utf8 : &empty);
}

int latok_token_spans_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                 int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                 int flags, void* stream) {
    LATOK_ENTER();
    static const uint8_t empty = 0;
    return compact_common(g, true, nullptr, byte_off, n_str, total_bytes, counts_out, spans_out, spans_cap, n_tokens_out,
                          flags, stream, nullptr, utf8 ? utf8 : &empty);
}

/* byte-space UTF-8 entry points: the tile kernel reads the bytes (1 B/char for ASCII), all positions are byte offsets */
int latok_split_mask_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                      uint64_t* mask_bits_out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    if (flags & LATOK_DEVICE_PTRS) {
        if ((rc = resolve_total_device(byte_off, n_str, &total_bytes, st))) return rc;
        if (total_bytes == 0) return LATOK_OK;
        if (!utf8 || !mask_bits_out) return fail(LATOK_ERR_INVALID, "NULL buffer");
        return run_pipeline(g, nullptr, byte_off, n_str, total_bytes, mask_bits_out, nullptr, latok::kModeBits, st, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, utf8);
    }
    if ((rc = check_csr_host(byte_off, n_str, &total_bytes))) return rc;
    if (total_bytes == 0) return LATOK_OK;
    if (!utf8 || !mask_bits_out) return fail(LATOK_ERR_INVALID, "NULL buffer");
    const size_t out_bytes = (size_t)((total_bytes + 63) / 64) * 8;
    if ((rc = g.u_bytes.ensure((size_t)total_bytes + 16))) return rc;
    if ((rc = g.u_boff.ensure((size_t)(n_str + 1) * 8))) return rc;
    if ((rc = g.h_out.ensure(out_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(g.u_bytes.p, utf8, (size_t)total_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g.u_boff.p, byte_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
    if ((rc = run_pipeline(g, nullptr, (const int64_t*)g.u_boff.p, n_str, total_bytes, (uint64_t*)g.h_out.p, nullptr,
                           latok::kModeBits, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           (const uint8_t*)g.u_bytes.p)))
        return rc;
    HIP_TRY(hipMemcpyAsync(mask_bits_out, g.h_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

int latok_split_offsets_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                         int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap,
                                         int64_t* n_offsets_out, int flags, void* stream) {
    LATOK_ENTER();
    static const uint8_t empty = 0;
    return compact_common(g, false, nullptr, byte_off, n_str, total_bytes, counts_out, offsets_out, offsets_cap,
                          n_offsets_out, flags, stream, nullptr, utf8 ? utf8 : &empty, true);
}

int latok_token_spans_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                       int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                       int flags, void* stream) {
    LATOK_ENTER();
    static const uint8_t empty = 0;
    return compact_common(g, true, nullptr, byte_off, n_str, total_bytes, counts_out, spans_out, spans_cap, n_tokens_out,
                          flags, stream, nullptr, utf8 ? utf8 : &empty, true);
}

/* PEP 393 buffers (the reference's own input, latok.c:53-55,79): fixed-width code units of 1, 2 or 4 bytes */
static int check_kind(int kind) {
    if (kind != 1 && kind != 2 && kind != 4) return fail(LATOK_ERR_INVALID, "kind must be 1 (Latin-1), 2 (UCS-2) or 4 (UCS-4), got %d", kind);
    return LATOK_OK;
}

int latok_split_mask_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                uint64_t* mask_bits_out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = check_kind(kind);
    if (rc) return rc;
    if (kind == 4) return split_common(g, (const uint32_t*)units, row_off, n_str, total_chars, mask_bits_out, latok::kModeBits, flags, stream);
    if ((rc = need_init(g))) return rc;
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const bool dev = (flags & LATOK_DEVICE_PTRS) != 0;
    int64_t total = total_chars;
    if (dev) {
        if ((rc = resolve_total_device(row_off, n_str, &total, st))) return rc;
    } else if ((rc = check_csr_host(row_off, n_str, &total))) {
        return rc;
    }
    if (total == 0) return LATOK_OK;
    if (!units || !mask_bits_out) return fail(LATOK_ERR_INVALID, "NULL buffer");
    const uint8_t* d_units = (const uint8_t*)units;
    const int64_t* d_row = row_off;
    uint64_t* d_bits = mask_bits_out;
    const size_t out_bytes = (size_t)((total + 63) / 64) * 8;
    if (!dev) {
        if ((rc = g.u_bytes.ensure((size_t)total * kind + 16))) return rc;
        if ((rc = g.u_boff.ensure((size_t)(n_str + 1) * 8))) return rc;
        if ((rc = g.h_out.ensure(out_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(g.u_bytes.p, units, (size_t)total * kind, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(g.u_boff.p, row_off, (size_t)(n_str + 1) * 8, hipMemcpyHostToDevice, st));
        d_units = (const uint8_t*)g.u_bytes.p;
        d_row = (const int64_t*)g.u_boff.p;
        d_bits = (uint64_t*)g.h_out.p;
    }
    rc = run_pipeline(g, nullptr, d_row, n_str, total, d_bits, nullptr, latok::kModeBits, st, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, nullptr, d_units, kind);
    if (rc) return rc;
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(mask_bits_out, d_bits, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LATOK_OK;
}

int latok_split_offsets_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                   int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap, int64_t* n_offsets_out,
                                   int flags, void* stream) {
    LATOK_ENTER();
    int rc = check_kind(kind);
    if (rc) return rc;
    if (kind == 4)
        return compact_common(g, false, (const uint32_t*)units, row_off, n_str, total_chars, counts_out, offsets_out, offsets_cap,
                              n_offsets_out, flags, stream);
    static const uint8_t empty = 0;
    return compact_common(g, false, nullptr, row_off, n_str, total_chars, counts_out, offsets_out, offsets_cap, n_offsets_out,
                          flags, stream, nullptr, units ? (const uint8_t*)units : &empty, true, kind);
}

int latok_token_spans_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                 int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                 int flags, void* stream) {
    LATOK_ENTER();
    int rc = check_kind(kind);
    if (rc) return rc;
    if (kind == 4)
        return compact_common(g, true, (const uint32_t*)units, row_off, n_str, total_chars, counts_out, spans_out, spans_cap,
                              n_tokens_out, flags, stream);
    static const uint8_t empty = 0;
    return compact_common(g, true, nullptr, row_off, n_str, total_chars, counts_out, spans_out, spans_cap, n_tokens_out, flags,
                          stream, nullptr, units ? (const uint8_t*)units : &empty, true, kind);
}

int latok_token_features_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                    int64_t* counts_out, int64_t* spans4_out, int8_t* features_out, int64_t cap,
                                    int64_t* n_tokens_out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = check_kind(kind);
    if (rc) return rc;
    if (!features_out && cap > 0) return fail(LATOK_ERR_INVALID, "features_out is NULL");
    static int8_t dummy = 0;
    int8_t* f = features_out ? features_out : &dummy;
    if (kind == 4)
        return compact_common(g, true, (const uint32_t*)units, row_off, n_str, total_chars, counts_out, spans4_out, cap, n_tokens_out,
                              flags, stream, f);
    static const uint8_t empty = 0;
    return compact_common(g, true, nullptr, row_off, n_str, total_chars, counts_out, spans4_out, cap, n_tokens_out, flags, stream,
                          f, units ? (const uint8_t*)units : &empty, true, kind);
}

int latok_token_features_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total,
                               int64_t* counts_out, int64_t* spans4_out, int8_t* features_out, int64_t cap,
                               int64_t* n_tokens_out, int flags, void* stream) {
    LATOK_ENTER();
    if (!features_out && cap > 0) return fail(LATOK_ERR_INVALID, "features_out is NULL");
    int8_t dummy = 0;
    return compact_common(g, true, cps, row_off, n_str, total, counts_out, spans4_out, cap, n_tokens_out, flags, stream,
                          features_out ? features_out : &dummy);
}

int latok_parse_matrix(const uint32_t* cps, int64_t n, int8_t* matrix_out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (n < 0) return fail(LATOK_ERR_INVALID, "n must be >= 0");
    if (n == 0) return LATOK_OK;
    if (!cps || !matrix_out) return fail(LATOK_ERR_INVALID, "NULL buffer");
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const uint8_t* t1 = (const uint8_t*)g.t1.p;
    const uint8_t* t2 = (const uint8_t*)g.t2cls.p;
    const uint16_t* cw = (const uint16_t*)g.cw.p;
    if (flags & LATOK_DEVICE_PTRS) {
        HIP_TRY(latok::launch_parse_matrix(cps, n, t1, t2, cw, matrix_out, st));
        return LATOK_OK;
    }
    if (n <= latok::kSmallMatrixChars) {
        // one string per call (the reference's pattern): chars and matrix pass through pinned memory, one workgroup, and the
        // call returns when it has seen the kernel's completion word
        const size_t po_out = align16((size_t)n * 4);
        if ((rc = g.pin.ensure(po_out + (size_t)n * LATOK_FEATURE_COUNT + 64))) return rc;
        if ((rc = g.pin_tot.ensure(64))) return rc;
        memcpy(g.pin.h, cps, (size_t)n * 4);
        const unsigned long long seq = ++g.small_seq;
        unsigned long long* d_done = poll_completion() ? (unsigned long long*)g.pin_tot.d + 2 : nullptr;
        HIP_TRY(latok::launch_parse_matrix_small((const uint32_t*)g.pin.d, (int)n, t1, t2, cw, (int8_t*)((char*)g.pin.d + po_out), d_done,
                                                 seq, st));
        if (!(d_done && wait_completion_word((const unsigned long long*)g.pin_tot.h + 2, seq))) HIP_TRY(hipStreamSynchronize(st));
        memcpy(matrix_out, (char*)g.pin.h + po_out, (size_t)n * LATOK_FEATURE_COUNT);
        return LATOK_OK;
    }
    if ((rc = g.h_cps.ensure((size_t)n * 4))) return rc;
    if ((rc = g.h_out.ensure((size_t)n * LATOK_FEATURE_COUNT))) return rc;
    HIP_TRY(hipMemcpyAsync(g.h_cps.p, cps, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(latok::launch_parse_matrix((const uint32_t*)g.h_cps.p, n, t1, t2, cw, (int8_t*)g.h_out.p, st));
    HIP_TRY(hipMemcpyAsync(matrix_out, g.h_out.p, (size_t)n * LATOK_FEATURE_COUNT, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

int latok_combine_matrix_rows(const int8_t* m, int64_t rows, int64_t cols, int64_t stride_r, int64_t stride_c,
                              const int8_t* idx, int idx_ndim, int irows, int icols, int8_t* out, int flags,
                              void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (idx_ndim != 1 && idx_ndim != 2) return fail(LATOK_ERR_INVALID, "must specify 2d numpy array args");
    if (rows < 0 || cols < 0 || irows < 0 || icols < 0) return fail(LATOK_ERR_INVALID, "negative shape");
    if (cols == 0) return LATOK_OK;
    if (!m || !idx || !out) return fail(LATOK_ERR_INVALID, "NULL buffer");
    const int n_idx = idx_ndim == 2 ? irows * icols : icols;
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    if (flags & LATOK_DEVICE_PTRS) {
        HIP_TRY(latok::launch_combine_rows((const uint8_t*)m, stride_r, stride_c, cols, idx, idx_ndim, irows, icols, out, st));
        return LATOK_OK;
    }
    // the reference does not bounds-check idx (latok.c:324-327); we refuse out-of-range rows instead of reading wild
    for (int i = 0; i < n_idx; ++i) {
        const uint8_t r = (uint8_t)idx[i];
        if (r != 255 && (int64_t)r >= rows) return fail(LATOK_ERR_INVALID, "idx value %d out of range for %lld rows", (int)r, (long long)rows);
    }
    if (cols <= latok::kSmallMatrixChars && rows <= 64) {
        // the matrix of one string: gathered straight into pinned memory, one workgroup, completion word (see latok_parse_matrix)
        const size_t n_m = (size_t)rows * (size_t)cols;
        const size_t po_idx = align16(n_m), po_out = po_idx + align16((size_t)n_idx);
        if ((rc = g.pin.ensure(po_out + (size_t)cols + 64))) return rc;
        if ((rc = g.pin_tot.ensure(64))) return rc;
        int8_t* hm = (int8_t*)g.pin.h;
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < cols; ++c) hm[(size_t)(r * cols + c)] = m[r * stride_r + c * stride_c];
        if (n_idx) memcpy((char*)g.pin.h + po_idx, idx, (size_t)n_idx);
        const unsigned long long seq = ++g.small_seq;
        unsigned long long* d_done = poll_completion() ? (unsigned long long*)g.pin_tot.d + 2 : nullptr;
        HIP_TRY(latok::launch_combine_rows((const uint8_t*)g.pin.d, cols, 1, cols, (const int8_t*)((char*)g.pin.d + po_idx), idx_ndim,
                                           irows, icols, (int8_t*)((char*)g.pin.d + po_out), st, d_done, seq));
        if (!(d_done && wait_completion_word((const unsigned long long*)g.pin_tot.h + 2, seq))) HIP_TRY(hipStreamSynchronize(st));
        memcpy(out, (char*)g.pin.h + po_out, (size_t)cols);
        return LATOK_OK;
    }
    // host: gather the (possibly strided) matrix into a dense rows x cols copy, upload, run, download
    std::vector<int8_t> dense((size_t)rows * (size_t)cols);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) dense[(size_t)(r * cols + c)] = m[r * stride_r + c * stride_c];
    if ((rc = g.h_cps.ensure(dense.size() + 16))) return rc;
    if ((rc = g.h_aux.ensure((size_t)n_idx + 16))) return rc;
    if ((rc = g.h_out.ensure((size_t)cols))) return rc;
    if (!dense.empty()) HIP_TRY(hipMemcpyAsync(g.h_cps.p, dense.data(), dense.size(), hipMemcpyHostToDevice, st));
    if (n_idx) HIP_TRY(hipMemcpyAsync(g.h_aux.p, idx, (size_t)n_idx, hipMemcpyHostToDevice, st));
    HIP_TRY(latok::launch_combine_rows((const uint8_t*)g.h_cps.p, cols, 1, cols, (const int8_t*)g.h_aux.p, idx_ndim,
                                       irows, icols, (int8_t*)g.h_out.p, st));
    HIP_TRY(hipMemcpyAsync(out, g.h_out.p, (size_t)cols, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LATOK_OK;
}

int latok_block_mask(const int8_t* a1, const int8_t* a2, int64_t n, int8_t* out, int flags, void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (n < 0) return fail(LATOK_ERR_INVALID, "n must be >= 0");
    if (n == 0) return LATOK_OK;
    if (!a1 || !a2 || !out) return fail(LATOK_ERR_INVALID, "must specify two aligning 1d numpy array args");
    StreamTurn turn(g, stream);
    hipStream_t st = turn.st;
    const bool dev = (flags & LATOK_DEVICE_PTRS) != 0;
    // the block mask of ONE array pair is the batch pipeline over a single "string" [0, n) whose planes are a1 / a2
    const int64_t row[2] = {0, n};
    if (!dev && n <= latok::kTile) {
        // at most one tile: one single-wave launch on pinned memory, completion word (see compact_common)
        const size_t an = align16((size_t)n);
        if ((rc = g.pin.ensure(3 * an + 16 + 64))) return rc;
        if ((rc = g.pin_tot.ensure(64))) return rc;
        memcpy(g.pin.h, a1, (size_t)n);
        memcpy((char*)g.pin.h + an, a2, (size_t)n);
        memcpy((char*)g.pin.h + 3 * an, row, 16);
        latok::SplitParams P;
        memset(&P, 0, sizeof(P));
        P.row_off = (const int64_t*)((char*)g.pin.d + 3 * an);
        P.n_str = 1;
        P.total = n;
        P.n_tiles = 1;
        P.bm_a1 = (const int8_t*)g.pin.d;
        P.bm_a2 = (const int8_t*)((char*)g.pin.d + an);
        P.values_out = (uint8_t*)((char*)g.pin.d + 2 * an);
        const unsigned long long seq = ++g.small_seq;
        unsigned long long* d_done = poll_completion() ? (unsigned long long*)g.pin_tot.d + 2 : nullptr;
        HIP_TRY(latok::launch_small_block_mask(P, d_done, seq, st));
        if (!(d_done && wait_completion_word((const unsigned long long*)g.pin_tot.h + 2, seq))) HIP_TRY(hipStreamSynchronize(st));
        memcpy(out, (char*)g.pin.h + 2 * an, (size_t)n);
        return LATOK_OK;
    }
    if ((rc = g.h_row.ensure(16))) return rc;
    HIP_TRY(hipMemcpyAsync(g.h_row.p, row, 16, hipMemcpyHostToDevice, st));
    const int8_t *d1 = a1, *d2 = a2;
    int8_t* dout = out;
    if (dev) {
        if ((((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)out) & 3) != 0)
            return fail(LATOK_ERR_INVALID, "device pointers must be 4-byte aligned");
    } else {
        if ((rc = g.h_cps.ensure((size_t)n + 16))) return rc;
        if ((rc = g.h_aux.ensure((size_t)n + 16))) return rc;
        if ((rc = g.h_out.ensure((size_t)n + 16))) return rc;
        HIP_TRY(hipMemcpyAsync(g.h_cps.p, a1, (size_t)n, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(g.h_aux.p, a2, (size_t)n, hipMemcpyHostToDevice, st));
        d1 = (const int8_t*)g.h_cps.p;
        d2 = (const int8_t*)g.h_aux.p;
        dout = (int8_t*)g.h_out.p;
    }
    int* d_flags = (int*)((char*)g.scalar.p + 16);
    HIP_TRY(latok::launch_any_nonzero(d1, d2, n, d_flags, st));
    if ((rc = run_pipeline(g, nullptr, (const int64_t*)g.h_row.p, 1, n, nullptr, (uint8_t*)dout, latok::kModeBlockMask, st,
                           nullptr, nullptr, d1, d2, d_flags)))
        return rc;
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LATOK_OK;
}

void* latok_dev_alloc(size_t bytes) {
    LATOK_ENTER();
    void* p = nullptr;
    if (!g.inited) { fail(LATOK_ERR_NOT_INIT, "latok_init() has not been called"); return nullptr; }
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { fail(LATOK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
int latok_dev_free(void* p) {
    LATOK_ENTER();
    if (p) HIP_TRY(hipFree(p));
    return LATOK_OK;
}
void* latok_host_alloc(size_t bytes) {
    LATOK_ENTER();
    void* p = nullptr;
    if (!g.inited) { fail(LATOK_ERR_NOT_INIT, "latok_init() has not been called"); return nullptr; }
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { fail(LATOK_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
int latok_host_free(void* p) {
    LATOK_ENTER();
    if (p) HIP_TRY(hipHostFree(p));
    return LATOK_OK;
}
int latok_memcpy_h2d(void* d, const void* s, size_t n) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return LATOK_OK;
}
int latok_memcpy_d2h(void* d, const void* s, size_t n) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return LATOK_OK;
}
int latok_memset_dev(void* d, int v, size_t n) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d, v, n, g.stream));
    return LATOK_OK;
}
static int flow_drain(Ctx& g);   // batch flow (below): its three streams
int latok_sync(void) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return flow_drain(g);
}
int latok_device_props(int* n_cu, int64_t* hbm_bytes, char* name_out, int name_cap) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g.device));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (name_out && name_cap > 0) snprintf(name_out, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    return LATOK_OK;
}

int latok_corpus_offsets(uint64_t seed, uint64_t sid0, int64_t n_str, int64_t len_lo, int64_t len_hi,
                         int64_t* row_off_out) {
    if (n_str < 0 || len_lo < 0 || len_hi < len_lo || !row_off_out) return fail(LATOK_ERR_INVALID, "bad corpus shape");
    int64_t acc = 0;
    row_off_out[0] = 0;
    for (int64_t s = 0; s < n_str; ++s) {
        acc += latok_corpus_length(seed, sid0 + (uint64_t)s, len_lo, len_hi);
        row_off_out[s + 1] = acc;
    }
    return LATOK_OK;
}
int latok_corpus_fill_host(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off,
                           uint32_t* cps_out) {
    if (n_str < 0 || !row_off || (!cps_out && n_str > 0 && row_off[n_str] > 0)) return fail(LATOK_ERR_INVALID, "bad corpus args");
    for (int64_t s = 0; s < n_str; ++s)
        latok_corpus_string(seed, model, sid0 + (uint64_t)s, cps_out + row_off[s], row_off[s + 1] - row_off[s]);
    return LATOK_OK;
}
int latok_corpus_fill_device(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off_dev,
                             uint32_t* cps_out_dev, void* stream) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(latok::launch_corpus_fill(seed, model, sid0, n_str, row_off_dev, cps_out_dev,
                                      stream ? (hipStream_t)stream : g.stream));
    return LATOK_OK;
}
int latok_utf8_bytes(const uint32_t* cps, int64_t n, int64_t* bytes_out, int flags) {
    if (!bytes_out || n < 0) return fail(LATOK_ERR_INVALID, "bad args");
    if (!(flags & LATOK_DEVICE_PTRS)) {
        int64_t t = 0;
        for (int64_t i = 0; i < n; ++i) t += latok_utf8_len(cps[i]);
        *bytes_out = t;
        return LATOK_OK;
    }
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    HIP_TRY(latok::launch_utf8_bytes(cps, n, (unsigned long long*)g.scalar.p, g.stream));
    unsigned long long t = 0;
    HIP_TRY(hipMemcpyAsync(&t, g.scalar.p, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *bytes_out = (int64_t)t;
    return LATOK_OK;
}

/* test hook (not part of the ABI in include/latok_hip.h): set the scan epoch of the current context, so that the wrap of
 * the 18-bit epoch of k_word_counts_scan's state can be exercised without 262 144 calls */
int latok_debug_set_scan_epoch(unsigned epoch) {
    LATOK_ENTER();
    g.scan_epoch = epoch & 0x3FFFFu;
    return LATOK_OK;
}

/* test hook (not part of the ABI; needs no device): the flow's routing of batches to slots (flow_hazards.h) driven without a
 * GPU.  One call = one batch touching n ranges (lo[i], bytes[i], is_write[i]); returns the slot it is enqueued on, *drained_out = 1
 * when the flow had to be drained first.  n < 0: forget everything (= latok_flow_wait).  State of the calling thread. */
extern "C" int latok_debug_flow_route(int n_slots, const uint64_t* lo, const uint64_t* bytes, const int* is_write, int n, int* drained_out) {
    static thread_local latok::FlowHazards held;
    static thread_local unsigned long long seq = 0;
    if (drained_out) *drained_out = 0;
    if (n < 0) { held.clear(); seq = 0; return 0; }
    if (n_slots < 1 || n_slots > latok::FlowHazards::kMaxSlots || n > 16) return fail(LATOK_ERR_INVALID, "1..4 slots, <= 16 ranges");
    latok::FlowRange r[16];
    for (int i = 0; i < n; ++i) r[i] = latok::flow_range((const void*)(uintptr_t)lo[i], (size_t)bytes[i], is_write[i] != 0);
    const int turn = (int)(seq % (unsigned)n_slots);
    int s = held.route(n_slots, turn, r, n);
    if (s == latok::FlowHazards::kDrainFirst) {
        held.clear();
        if (drained_out) *drained_out = 1;
        s = turn;
    }
    held.note(s, r, n);
    ++seq;
    return s;
}

/* test hook (not part of the ABI; needs no device): the host decoder of small UTF-8 batches (host_decode_small).  Returns 1 and
 * fills cps_out[<= total bytes], cp_row_out[n_str + 1], bytepos_out[<= total bytes + 1] when every string is well formed,
 * 0 when the batch is left to the device paths, < 0 on a bad argument. */
extern "C" int latok_debug_host_decode_utf8(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, uint32_t* cps_out,
                                            int64_t* cp_row_out, int64_t* bytepos_out, int64_t* n_cps_out) {
    int64_t total = -1;
    if (!byte_off || n_str <= 0 || check_csr_host(byte_off, n_str, &total) != LATOK_OK) return LATOK_ERR_INVALID;
    std::vector<uint32_t> cps;
    std::vector<int64_t> row, pos;
    if (!host_decode_small(utf8, byte_off, n_str, cps, row, pos)) return 0;
    memcpy(cps_out, cps.data(), cps.size() * 4);
    memcpy(cp_row_out, row.data(), row.size() * 8);
    memcpy(bytepos_out, pos.data(), pos.size() * 8);
    *n_cps_out = (int64_t)cps.size();
    return 1;
}


// ---- batch flow: several device-resident batches in flight on one context --------------------------------------------
// A batch is three dependent launches (string index, tiles, resolve) and only the tile kernel fills the chip; back to back on
// one stream the two latency-bound launches and the three kernel boundaries cost 11-14 us of a 108 us step on C2.  A flow
// gives every batch in flight its own stream and workspace (two slots, used in turn) and NO dependency between the streams:
// the tile kernel of batch i+1 takes over the CUs as the workgroups of batch i retire, and the small launches of one stream
// run in the shadow of the other stream's tile kernel.  (Measured first: one stream per STAGE with events between them, so that
// the tile kernels stay strictly back to back -- 115-120 us per step, slower than serial: an inter-queue event wait costs more
// than the launch it hides.)
static int flow_setup(Ctx& g) {
    if (g.flow_ready) return LATOK_OK;
    g.flow_slots = 2;   // (3 and 4 slots measured: nothing over 2, profiles/r03_ab_flow_slots.txt)
    for (int i = 0; i < g.flow_slots; ++i) {
        Ctx::FlowSlot& f = g.flow[i];
        if (!f.st) HIP_TRY(hipStreamCreateWithFlags(&f.st, hipStreamNonBlocking));
    }
    g.flow_ready = true;
    return LATOK_OK;
}
static int flow_drain(Ctx& g) {
    if (!g.flow_ready) return LATOK_OK;
    // The streams are POLLED for up to 2 ms before the call blocks on them: a blocking wait comes back ~15 us after the last
    // kernel has ended (the runtime's wake-up), which is 1.5 % of a 20-batch flow on C2 (same-box A/B, K = 20: 89.1-90.7 ->
    // 86.1-88.7 us per batch).
    constexpr bool drain_poll = true;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < g.flow_slots; ++i) {
        bool done = false;
        if (drain_poll) {
            for (unsigned spins = 0;; ++spins) {
                const hipError_t q = hipStreamQuery(g.flow[i].st);
                if (q == hipSuccess) { done = true; break; }
                if (q != hipErrorNotReady) return fail(LATOK_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
                cpu_relax();
                if ((spins & 63) == 63 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > 2000)
                    break;
            }
        }
        if (!done) HIP_TRY(hipStreamSynchronize(g.flow[i].st));
    }
    g.flow_held.clear();   // nothing in flight: no buffer is being read or written
    return LATOK_OK;
}
// Reserve what a batch needs in a slot.  A buffer that has to grow is reallocated only once nothing in flight can still use it.
struct SlotNeed {
    DevBuf* buf;
    size_t bytes;
};
static int flow_reserve(Ctx& g, std::initializer_list<SlotNeed> needs) {
    bool grow = false;
    for (const SlotNeed& n : needs) grow = grow || n.buf->cap < n.bytes;
    if (!grow) return LATOK_OK;
    int rc = flow_drain(g);
    for (const SlotNeed& n : needs)
        if (!rc) rc = n.buf->ensure(n.bytes);
    return rc;
}
// Slots are used in turn -- except that a batch which touches memory a batch still in flight writes (or writes memory one
// reads) goes to THAT batch's slot, whose stream orders the two; when batches of several slots are in its way the flow is
// drained first (flow_hazards.h; callers that alternate buffers never hit either; an event per batch to order such pairs
// across streams would cost every batch ~3 us).  *slot_out = the slot to enqueue on; flow_note after the batch is enqueued.
static int flow_pick(Ctx& g, const latok::FlowRange* r, int n, int* slot_out) {
    const int turn = (int)(g.flow_seq % (unsigned)g.flow_slots);
    int s = g.flow_held.route(g.flow_slots, turn, r, n);
    if (s == latok::FlowHazards::kDrainFirst) {
        const int rc = flow_drain(g);
        if (rc) return rc;
        s = turn;
    }
    // a caller that never waits and never repeats a buffer: forget what an idle slot held; bound the list of a busy one
    if (g.flow_held.held(s) >= latok::FlowHazards::kPruneAt) {
        const hipError_t q = hipStreamQuery(g.flow[s].st);
        if (q == hipSuccess) g.flow_held.clear_slot(s);
        else if (q != hipErrorNotReady) return fail(LATOK_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        else if (g.flow_held.held(s) >= 4 * latok::FlowHazards::kPruneAt) {
            HIP_TRY(hipStreamSynchronize(g.flow[s].st));
            g.flow_held.clear_slot(s);
        }
    }
    *slot_out = s;
    return LATOK_OK;
}
// units: UTF-32 code points (unit_kind 4), PEP 393 units (1 / 2; positions are chars) or UTF-8 bytes (0; byte space)
static int flow_submit(Ctx& g, const void* units, int unit_kind, const int64_t* row_off, int64_t n_str, int64_t total, uint64_t* mask,
                       int* slot_used = nullptr) {
    int rc = flow_setup(g);
    if (rc) return rc;
    if (n_str <= 0 || total <= 0) return LATOK_OK;
    if (!units || !row_off || !mask) return fail(LATOK_ERR_INVALID, "NULL buffer");
    if (((uintptr_t)units & 15) != 0) return fail(LATOK_ERR_INVALID, "device input pointer must be 16-byte aligned");
    const uint32_t* cps = unit_kind == 4 ? (const uint32_t*)units : nullptr;
    const uint8_t* u8 = unit_kind == 4 ? nullptr : (const uint8_t*)units;
    const int64_t n_tiles = (total + latok::kTile - 1) / latok::kTile;
    const size_t unit_bytes = unit_kind == 0 ? 1 : (size_t)unit_kind;
    const latok::FlowRange touched[3] = {latok::flow_range(mask, (size_t)((total + 63) / 64) * 8, true),
                                         latok::flow_range(units, (size_t)total * unit_bytes, false),
                                         latok::flow_range(row_off, (size_t)(n_str + 1) * 8, false)};
    int slot = 0;
    if ((rc = flow_pick(g, touched, 3, &slot))) return rc;
    Ctx::FlowSlot& f = g.flow[slot];
    if ((rc = flow_reserve(g, {{&f.summ, ws_summ_bytes(n_tiles)}, {&f.seg_agg, ws_seg_bytes(n_tiles)}, {&f.tile_first, ws_first_bytes(n_tiles)},
                               {&f.fix_count, 8}})))
        return rc;
    if ((rc = run_pipeline(g, cps, row_off, n_str, total, mask, nullptr, latok::kModeBits, f.st, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, u8, unit_kind == 4 ? 0 : unit_kind, 7, nullptr, latok::DoneSignal{nullptr, 0, nullptr}, &f)))
        return rc;
    g.flow_held.note(slot, touched, 3);
    if (slot_used) *slot_used = slot;
    ++g.flow_seq;
    return LATOK_OK;
}
// offsets (spans = false) or token spans of one batch: everything latok_split_offsets_batch / latok_token_spans_batch launch,
// on the slot's stream and workspaces; the item total and the error flags land in result[0..1] when the stream gets there
static int flow_submit_compact(Ctx& g, bool spans, const void* units, int unit_kind, const int64_t* row_off, int64_t n_str, int64_t total,
                               void* counts, void* items, int64_t cap, int64_t* result, int flags, int8_t* feat = nullptr, bool feats = false) {
    int rc = flow_setup(g);
    if (rc) return rc;
    if (!result) return fail(LATOK_ERR_INVALID, "NULL result pointer");
    if (((uintptr_t)result & 7) != 0) return fail(LATOK_ERR_INVALID, "result pointer must be 8-byte aligned");
    const size_t rec = (flags & LATOK_OUT_INT32) ? 4 : 8;   // bytes of a count / of one field of a record
    if (n_str <= 0 || total <= 0) {   // nothing to launch: counts of empty strings are zero, no items
        const latok::FlowRange w[2] = {latok::flow_range(result, 16, true), latok::flow_range(counts, n_str > 0 ? (size_t)n_str * rec : 0, true)};
        int s0 = 0;
        if ((rc = flow_pick(g, w, 2, &s0))) return rc;
        HIP_TRY(hipMemsetAsync(result, 0, 16, g.flow[s0].st));
        if (n_str > 0 && counts) HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n_str * rec, g.flow[s0].st));
        g.flow_held.note(s0, w, 2);
        return LATOK_OK;
    }
    if (!units || !row_off || !counts || ((!items || (feats && !feat)) && cap > 0)) return fail(LATOK_ERR_INVALID, "NULL buffer");
    if (cap < 0) return fail(LATOK_ERR_INVALID, "capacity must be >= 0");
    if (feats && unit_kind == 0) return fail(LATOK_ERR_INVALID, "featurize reads code points or PEP 393 units, not UTF-8 bytes");
    if (((uintptr_t)units & 15) != 0) return fail(LATOK_ERR_INVALID, "device input pointer must be 16-byte aligned");
    const bool o32 = (flags & LATOK_OUT_INT32) != 0;
    if (((uintptr_t)items & 15) != 0 || ((uintptr_t)counts & (o32 ? 3 : 7)) != 0) return fail(LATOK_ERR_INVALID, "misaligned output buffer");
    const uint32_t* cps = unit_kind == 4 ? (const uint32_t*)units : nullptr;
    const uint8_t* u8 = unit_kind == 4 ? nullptr : (const uint8_t*)units;
    const int64_t n_tiles = (total + latok::kTile - 1) / latok::kTile;
    const int64_t words = (total + 63) / 64, c_tiles = (words + 63) / 64;
    // every output of the batch -- records, counts, result words, feature sums -- and its inputs
    const size_t unit_bytes = unit_kind == 0 ? 1 : (size_t)unit_kind;
    const size_t fields = feats ? 4 : (spans ? 2 : 1);
    const latok::FlowRange touched[6] = {latok::flow_range(items, (size_t)cap * fields * rec, true),
                                         latok::flow_range(counts, (size_t)n_str * rec, true),
                                         latok::flow_range(result, 16, true),
                                         latok::flow_range(feat, feats ? (size_t)cap * LATOK_FEATURE_COUNT : 0, true),
                                         latok::flow_range(units, (size_t)total * unit_bytes, false),
                                         latok::flow_range(row_off, (size_t)(n_str + 1) * 8, false)};
    int slot = 0;
    if ((rc = flow_pick(g, touched, 6, &slot))) return rc;
    Ctx::FlowSlot& f = g.flow[slot];
    if ((rc = flow_reserve(g, {{&f.summ, ws_summ_bytes(n_tiles)}, {&f.seg_agg, ws_seg_bytes(n_tiles)}, {&f.tile_first, ws_first_bytes(n_tiles)},
                               {&f.fix_count, 8}, {&f.bits, (size_t)words * 8 + 8}, {&f.space, spans ? (size_t)words * 8 + 8 : 0},
                               {&f.kept, spans ? (size_t)words * 8 + 8 : 0}, {&f.wcnt, (size_t)c_tiles * 8 + 8}, {&f.bases, (size_t)c_tiles * 8 + 8},
                               {&f.wpref, (size_t)words * 2 + 8}, {&f.scalar, 64}, {&f.chain, (size_t)latok::count_blocks(words) * 8 + 64},
                               {&f.chain_ctl, 64}, {&f.codes, feats ? (size_t)total + latok::kTile + 256 : 0},
                               {&f.widened, feats && unit_kind != 4 ? (size_t)total * 4 + 16 : 0}})))
        return rc;
    if ((rc = enqueue_compaction_dev(g, spans, feats, o32, cps, u8, unit_kind == 4 ? 0 : unit_kind, row_off, n_str, total, counts, items, feat,
                                     cap, result, nullptr, f.st, latok::DoneSignal{nullptr, 0, nullptr}, &f)))
        return rc;
    g.flow_held.note(slot, touched, 6);
    ++g.flow_seq;
    return LATOK_OK;
}

int latok_flow_split_mask(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                          uint64_t* mask_dev) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (total_chars < 0) {
        if ((rc = resolve_total_device(row_off_dev, n_str, &total_chars, g.stream))) return rc;
    }
    return flow_submit(g, cps_dev, 4, row_off_dev, n_str, total_chars, mask_dev);
}
int latok_flow_split_mask_kind(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                               uint64_t* mask_dev) {
    LATOK_ENTER();
    int rc = check_kind(kind);
    if (rc) return rc;
    if ((rc = need_init(g))) return rc;
    if (total_chars < 0) {
        if ((rc = resolve_total_device(row_off_dev, n_str, &total_chars, g.stream))) return rc;
    }
    return flow_submit(g, units_dev, kind, row_off_dev, n_str, total_chars, mask_dev);
}
int latok_flow_split_mask_utf8_bytes(const uint8_t* utf8_dev, const int64_t* byte_off_dev, int64_t n_str, int64_t total_bytes,
                                     uint64_t* mask_dev) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (total_bytes < 0) {
        if ((rc = resolve_total_device(byte_off_dev, n_str, &total_bytes, g.stream))) return rc;
    }
    return flow_submit(g, utf8_dev, 0, byte_off_dev, n_str, total_bytes, mask_dev);
}
static int flow_compact_entry(bool spans, const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_units,
                              void* counts_dev, void* items_dev, int64_t cap, int64_t* result_dev, int flags) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (kind != 0 && (rc = check_kind(kind))) return rc;
    if (flags & ~(LATOK_OUT_INT32 | LATOK_DEVICE_PTRS)) return fail(LATOK_ERR_INVALID, "unknown flag");
    if (total_units < 0) {
        if ((rc = resolve_total_device(row_off_dev, n_str, &total_units, g.stream))) return rc;
    }
    return flow_submit_compact(g, spans, units_dev, kind, row_off_dev, n_str, total_units, counts_dev, items_dev, cap, result_dev, flags);
}
int latok_flow_split_offsets(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_units,
                             void* counts_dev, void* offsets_dev, int64_t offsets_cap, int64_t* result_dev, int flags) {
    return flow_compact_entry(false, units_dev, kind, row_off_dev, n_str, total_units, counts_dev, offsets_dev, offsets_cap, result_dev, flags);
}
int latok_flow_token_spans(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_units,
                           void* counts_dev, void* spans_dev, int64_t spans_cap, int64_t* result_dev, int flags) {
    return flow_compact_entry(true, units_dev, kind, row_off_dev, n_str, total_units, counts_dev, spans_dev, spans_cap, result_dev, flags);
}
int latok_flow_token_features(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                              void* counts_dev, void* spans4_dev, int8_t* features_dev, int64_t cap, int64_t* result_dev, int flags) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if ((rc = check_kind(kind))) return rc;
    if (flags & ~(LATOK_OUT_INT32 | LATOK_DEVICE_PTRS)) return fail(LATOK_ERR_INVALID, "unknown flag");
    if (total_chars < 0) {
        if ((rc = resolve_total_device(row_off_dev, n_str, &total_chars, g.stream))) return rc;
    }
    return flow_submit_compact(g, true, units_dev, kind, row_off_dev, n_str, total_chars, counts_dev, spans4_dev, cap, result_dev, flags,
                               features_dev, true);
}
int latok_flow_wait(void) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    return flow_drain(g);
}

int latok_bench_stream_read(const void* buf_dev, int64_t bytes, int warmup, int iters, float* ms_out) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (!buf_dev || !ms_out || bytes < 16384 || iters <= 0 || warmup < 0)
        return fail(LATOK_ERR_INVALID, "stream_read: need a device buffer of >= 16 KiB, iters > 0");
    if (((uintptr_t)buf_dev & 15) != 0) return fail(LATOK_ERR_INVALID, "device pointer must be 16-byte aligned");
    StreamTurn turn(g, nullptr);
    hipStream_t st = turn.st;
    uint32_t* sink = (uint32_t*)g.scalar.p + 8;
    for (int i = 0; i < warmup; ++i) HIP_TRY(latok::launch_stream_read(buf_dev, bytes, sink, g.n_cu, st));
    HIP_TRY(hipEventRecord(g.ev[0], st));
    for (int i = 0; i < iters; ++i) HIP_TRY(latok::launch_stream_read(buf_dev, bytes, sink, g.n_cu, st));
    HIP_TRY(hipEventRecord(g.ev[1], st));
    HIP_TRY(hipEventSynchronize(g.ev[1]));
    HIP_TRY(hipEventElapsedTime(ms_out, g.ev[0], g.ev[1]));
    return LATOK_OK;
}

int latok_bench_split_mask(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total,
                           uint64_t* mask_dev, int warmup, int iters, float* ms_total_out, float* ms_tiles_out,
                           int64_t* n_fix_tiles_out) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (iters < 0 || warmup < 0) return fail(LATOK_ERR_INVALID, "iters and warmup must be >= 0");
    if (((uintptr_t)cps_dev & 15) != 0) return fail(LATOK_ERR_INVALID, "device cps pointer must be 16-byte aligned");
    StreamTurn turn(g, nullptr);
    hipStream_t st = turn.st;
    if ((rc = resolve_total_device(row_off_dev, n_str, &total, st))) return rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st))) return rc;
    // (1) whole pipeline, `iters` passes between one pair of events on the launch stream
    if (ms_total_out) {
        HIP_TRY(hipEventRecord(g.ev[0], st));
        for (int i = 0; i < iters; ++i)
            if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st))) return rc;
        HIP_TRY(hipEventRecord(g.ev[1], st));
        HIP_TRY(hipEventSynchronize(g.ev[1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev[0], g.ev[1]));
        *ms_total_out = ms;
    }
    // (2) the dominant kernel alone: `iters` launches of stage 1 back to back between ONE pair of events (the kernel is
    //     idempotent: it reads the code points, row_off and the tile index of the batch, which stage 0 left in place, and
    //     rewrites the same provisional bitmask and summaries).  An event pair around every launch inside the pipeline
    //     charges each interval with the markers' own dispatch, ~6 us per launch (rocprofv3 kernel trace of such a run:
    //     5.8 us of idle queue in front of every k_tiles_main, 102 us by events against 96 us kernel time).
    if (ms_tiles_out) {
        HIP_TRY(hipEventRecord(g.ev[2], st));
        for (int i = 0; i < iters; ++i)
            if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st, nullptr, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 2)))
                return rc;
        HIP_TRY(hipEventRecord(g.ev[3], st));
        HIP_TRY(hipEventSynchronize(g.ev[3]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g.ev[2], g.ev[3]));
        *ms_tiles_out = ms;
        // leave a resolved mask behind
        if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (n_fix_tiles_out) {
        *n_fix_tiles_out = 0;
        if (total > 0) HIP_TRY(hipMemcpy(n_fix_tiles_out, g.fix_count.p, 8, hipMemcpyDeviceToHost));
    }
    return LATOK_OK;
}


// ---- several contexts timed as one job ---------------------------------------------------------------------------------
}  // extern "C"
struct latok_gate {
    int parties = 1;
    std::atomic<int> arrived{0};
    std::atomic<unsigned> phase{0};
    std::atomic<bool> broken{false};
    std::atomic<unsigned> ready{0};      // shared gates: kGateReady once the creator has finished setting the object up
};
constexpr unsigned kGateReady = 0x6A7E0001u;
extern "C" {

int latok_gate_create(int parties, latok_gate** gate_out) {
    if (parties < 1 || !gate_out) return fail(LATOK_ERR_INVALID, "gate: parties must be >= 1");
    latok_gate* g = new (std::nothrow) latok_gate;
    if (!g) return fail(LATOK_ERR_NOMEM, "out of host memory");
    g->parties = parties;
    *gate_out = g;
    return LATOK_OK;
}
int latok_gate_destroy(latok_gate* gate) {
    delete gate;
    return LATOK_OK;
}
static int gate_map_shared(const char* name, bool create, int parties, latok_gate** gate_out) {
    if (!name || name[0] != '/' || !gate_out) return fail(LATOK_ERR_INVALID, "gate: shared name must start with '/'");
    const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return fail(LATOK_ERR_INVALID, "gate: shm_open(%s) failed: %s", name, strerror(errno));
    if (create && ftruncate(fd, (off_t)sizeof(latok_gate)) != 0) {
        close(fd);
        shm_unlink(name);
        return fail(LATOK_ERR_NOMEM, "gate: ftruncate failed: %s", strerror(errno));
    }
    if (!create) {
        // the creator may be between shm_open and ftruncate (a zero-length object: touching the mapping would be SIGBUS) or
        // before its placement-new: attach only to an object of full size whose `ready` word says it is set up
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size < (off_t)sizeof(latok_gate)) {
            close(fd);
            return fail(LATOK_ERR_INVALID, "gate: %s is not ready yet", name);
        }
    }
    void* p = mmap(nullptr, sizeof(latok_gate), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(LATOK_ERR_NOMEM, "gate: mmap failed: %s", strerror(errno));
    latok_gate* g;
    if (create) {
        g = new (p) latok_gate;
        g->parties = parties;
        g->ready.store(kGateReady, std::memory_order_release);      // last: everything above is visible to whoever sees it
    } else {
        g = reinterpret_cast<latok_gate*>(p);
        if (g->ready.load(std::memory_order_acquire) != kGateReady) {
            munmap(p, sizeof(latok_gate));
            return fail(LATOK_ERR_INVALID, "gate: %s is not ready yet", name);
        }
    }
    *gate_out = g;
    return LATOK_OK;
}
int latok_gate_create_shared(const char* name, int parties, latok_gate** gate_out) {
    if (parties < 1) return fail(LATOK_ERR_INVALID, "gate: parties must be >= 1");
    return gate_map_shared(name, true, parties, gate_out);
}
int latok_gate_attach_shared(const char* name, latok_gate** gate_out) { return gate_map_shared(name, false, 0, gate_out); }
int latok_gate_detach_shared(latok_gate* gate) {
    if (gate) munmap(gate, sizeof(latok_gate));
    return LATOK_OK;
}
int latok_gate_unlink_shared(const char* name) {
    if (name) shm_unlink(name);
    return LATOK_OK;
}
int latok_gate_break(latok_gate* gate) {
    if (gate) gate->broken.store(true, std::memory_order_release);
    return LATOK_OK;
}
int latok_gate_wait(latok_gate* gate, double timeout_s) {
    if (!gate) return LATOK_OK;
    if (gate->broken.load(std::memory_order_acquire)) return fail(LATOK_ERR_INVALID, "gate: broken by another party");
    const unsigned ph = gate->phase.load(std::memory_order_acquire);
    if (gate->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == gate->parties) {   // last one in opens the gate
        gate->arrived.store(0, std::memory_order_relaxed);
        gate->phase.store(ph + 1, std::memory_order_release);
        return LATOK_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (gate->phase.load(std::memory_order_acquire) != ph) return LATOK_OK;
        if (gate->broken.load(std::memory_order_acquire)) return fail(LATOK_ERR_INVALID, "gate: broken by another party");
        cpu_relax();
        if ((spins & 1023) == 1023) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(LATOK_ERR_INVALID, "gate: %d parties expected, not all arrived within %.1f s", gate->parties, timeout_s);
            std::this_thread::yield();   // more host threads than cores (rehearsals): let the others arrive
        }
    }
}

static inline int64_t mono_ns() {
    return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static thread_local int tl_bench_used_graph = 0;
/* measurement aid (not part of the ABI in include/latok_hip.h): 1 when this thread's last latok_bench_split_mask_gated replayed
 * a captured hipGraph (LATOK_BENCH_GRAPH=1 and the capture succeeded), 0 when it launched the passes one by one */
int latok_debug_bench_used_graph(void) { return tl_bench_used_graph; }

int latok_bench_split_mask_gated(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total,
                                 uint64_t* mask_dev, int iters, latok_gate* gate, float* ms_events_out, int64_t* t0_ns_out,
                                 int64_t* t1_ns_out) {
    LATOK_ENTER();
    tl_bench_used_graph = 0;
    int rc = need_init(g);
    if (rc) return rc;
    if (iters < 1) return fail(LATOK_ERR_INVALID, "iters must be >= 1");
    if (((uintptr_t)cps_dev & 15) != 0) return fail(LATOK_ERR_INVALID, "device cps pointer must be 16-byte aligned");
    StreamTurn turn(g, nullptr);
    hipStream_t st = turn.st;
    if ((rc = resolve_total_device(row_off_dev, n_str, &total, st))) return rc;
    // LATOK_BENCH_GRAPH=1 (bench.py --launch threads, N > 1): the K passes are captured into ONE hipGraph outside the
    // timed region and replayed by one call inside it -- with N host threads of one process launching 3 kernels per 0.1 ms
    // step each, the threads would otherwise meet in the runtime's launch path.  Same kernels, same order, same stream.
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    {
        const char* e = getenv("LATOK_BENCH_GRAPH");
        if (e && e[0] == '1') {
            // (the workspaces are sized by the caller's warm-up passes; one eager pass here makes sure of it: nothing may
            // allocate during a capture)
            if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st))) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                for (int i = 0; i < iters && !rc; ++i)
                    rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st);
                hipError_t ce = hipStreamEndCapture(st, &graph);
                if (!rc && ce == hipSuccess && graph) ce = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
                if (rc || ce != hipSuccess || !gexec) {      // no graph on this runtime: the eager form below
                    if (gexec) (void)hipGraphExecDestroy(gexec);
                    if (graph) (void)hipGraphDestroy(graph);
                    graph = nullptr;
                    gexec = nullptr;
                    rc = LATOK_OK;
                    (void)hipGetLastError();
                }
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = latok_gate_wait(gate, 120.0))) {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    const int64_t t0 = mono_ns();
    HIP_TRY(hipEventRecord(g.ev[0], st));
    if (gexec) {
        tl_bench_used_graph = 1;
        if (hipGraphLaunch(gexec, st) != hipSuccess) rc = fail(LATOK_ERR_HIP, "hipGraphLaunch failed");
    } else {
        for (int i = 0; i < iters; ++i)
            if ((rc = run_pipeline(g, cps_dev, row_off_dev, n_str, total, mask_dev, nullptr, latok::kModeBits, st))) break;
    }
    if (!rc) {
        hipError_t e = hipEventRecord(g.ev[1], st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(LATOK_ERR_HIP, "timed region failed: %s", hipGetErrorString(e));
    }
    const int64_t t1 = mono_ns();
    const int rc_gate = latok_gate_wait(gate, 120.0);   // also on failure: the other threads must not wait for this one
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc) return rc;
    if (rc_gate) return rc_gate;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g.ev[0], g.ev[1]));
    if (ms_events_out) *ms_events_out = ms;
    if (t0_ns_out) *t0_ns_out = t0;
    if (t1_ns_out) *t1_ns_out = t1;
    return LATOK_OK;
}

int latok_bench_split_mask_flow_gated(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total,
                                      uint64_t* mask_a_dev, uint64_t* mask_b_dev, int iters, latok_gate* gate,
                                      float* ms_events_out, int64_t* t0_ns_out, int64_t* t1_ns_out) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (iters < 1) return fail(LATOK_ERR_INVALID, "iters must be >= 1");
    if (!mask_a_dev || !mask_b_dev) return fail(LATOK_ERR_INVALID, "NULL buffer");
    if ((rc = resolve_total_device(row_off_dev, n_str, &total, g.stream))) return rc;
    if ((rc = flow_setup(g)) || (rc = flow_drain(g))) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    if ((rc = latok_gate_wait(gate, 120.0))) return rc;
    const int64_t t0 = mono_ns();
    HIP_TRY(hipEventRecord(g.ev[0], g.flow[g.flow_seq % (unsigned)g.flow_slots].st));   // the stream of the first submission
    int last_slot = 0;
    const uint32_t* cps_b = g.bench_cps_b ? g.bench_cps_b : cps_dev;     // (latok_bench_set_second_input: a copy at another address)
    const int64_t* row_b = g.bench_row_b ? g.bench_row_b : row_off_dev;
    for (int i = 0; i < iters; ++i)
        if ((rc = flow_submit(g, (i & 1) ? cps_b : cps_dev, 4, (i & 1) ? row_b : row_off_dev, n_str, total, (i & 1) ? mask_b_dev : mask_a_dev,
                              &last_slot)))
            break;
    if (!rc) {
        hipError_t e = hipEventRecord(g.ev[1], g.flow[last_slot].st);   // ... of the last one
        if (e != hipSuccess) rc = fail(LATOK_ERR_HIP, "timed region failed: %s", hipGetErrorString(e));
    }
    const int rc_drain = flow_drain(g);
    const int64_t t1 = mono_ns();
    const int rc_gate = latok_gate_wait(gate, 120.0);   // also on failure: the other ranks must not wait for this one
    if (rc) return rc;
    if (rc_drain) return rc_drain;
    if (rc_gate) return rc_gate;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g.ev[0], g.ev[1]));
    if (ms_events_out) *ms_events_out = ms;
    if (t0_ns_out) *t0_ns_out = t0;
    if (t1_ns_out) *t1_ns_out = t1;
    return LATOK_OK;
}

int latok_bench_set_second_input(const uint32_t* cps_b_dev, const int64_t* row_off_b_dev) {
    LATOK_ENTER();
    if ((cps_b_dev == nullptr) != (row_off_b_dev == nullptr)) return fail(LATOK_ERR_INVALID, "both pointers or neither");
    if (((uintptr_t)cps_b_dev & 15) != 0) return fail(LATOK_ERR_INVALID, "device cps pointer must be 16-byte aligned");
    g.bench_cps_b = cps_b_dev;
    g.bench_row_b = row_off_b_dev;
    return LATOK_OK;
}

int latok_bench_tiles_flow(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total, uint64_t* mask_a_dev,
                           uint64_t* mask_b_dev, int iters, float* ms_out) {
    LATOK_ENTER();
    int rc = need_init(g);
    if (rc) return rc;
    if (iters < 1 || !ms_out || !mask_a_dev || !mask_b_dev) return fail(LATOK_ERR_INVALID, "iters >= 1, two mask buffers and ms_out are needed");
    if ((rc = resolve_total_device(row_off_dev, n_str, &total, g.stream))) return rc;
    if (total <= 0) { *ms_out = 0.f; return LATOK_OK; }
    // one whole pass per slot first: workspaces sized, the per-tile string index of the batch in place in BOTH slots
    const uint32_t* cps_b = g.bench_cps_b ? g.bench_cps_b : cps_dev;
    const int64_t* row_b = g.bench_row_b ? g.bench_row_b : row_off_dev;
    int slot_of[2] = {0, 1};
    for (int i = 0; i < 2; ++i)
        if ((rc = flow_submit(g, i ? cps_b : cps_dev, 4, i ? row_b : row_off_dev, n_str, total, i ? mask_b_dev : mask_a_dev, &slot_of[i]))) return rc;
    if ((rc = flow_drain(g))) return rc;
    const int64_t t0 = mono_ns();
    for (int i = 0; i < iters && !rc; ++i) {
        Ctx::FlowSlot& f = g.flow[slot_of[i & 1]];
        rc = run_pipeline(g, (i & 1) ? cps_b : cps_dev, (i & 1) ? row_b : row_off_dev, n_str, total, (i & 1) ? mask_b_dev : mask_a_dev, nullptr,
                          latok::kModeBits, f.st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 2, nullptr,
                          latok::DoneSignal{nullptr, 0, nullptr}, &f);
    }
    const int rc_drain = flow_drain(g);
    const int64_t t1 = mono_ns();
    if (rc) return rc;
    if (rc_drain) return rc_drain;
    *ms_out = (float)((t1 - t0) / 1e6);
    // leave resolved masks behind
    for (int i = 0; i < 2; ++i)
        if ((rc = flow_submit(g, cps_dev, 4, row_off_dev, n_str, total, i ? mask_b_dev : mask_a_dev))) return rc;
    return flow_drain(g);
}

}  // extern "C"
