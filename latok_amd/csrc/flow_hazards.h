// Batch flow: which slot (= which stream) may a new batch go to?  Host-side bookkeeping only, no device call -- it is unit
// tested on CPU through latok_debug_router_* (tests/test_host_api.py).
//
// The reference's contract (latok/core/default_tokenizer.py:137-160): every tokenize call is independent of the one before.
// A flow keeps batches in flight on several streams with NO ordering between the streams, so two batches of which one writes
// memory the other reads or writes must share a stream.  Every slot keeps the memory ranges of ALL batches submitted to it
// since it was last known idle (not only of its last batch: A->X on slot 0, B on 1, C on 0, D->X on 1 must order D behind A);
// a new batch is routed by OVERLAP of ranges (not by equality of base pointers):
//   no conflict with anything in flight      -> the slots in turn
//   conflicts with batches of ONE slot       -> that slot (its stream orders them)
//   conflicts with batches of SEVERAL slots  -> the caller drains the flow first, then the slots in turn
// A conflict is write/write, write-after-read or read-after-write; two reads never conflict.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace latok {

struct FlowRange {
    uintptr_t lo = 0, hi = 0;   // [lo, hi)
    bool write = false;
};
inline FlowRange flow_range(const void* p, size_t bytes, bool write) {
    FlowRange r;
    r.lo = (uintptr_t)p;
    r.hi = p && bytes ? (uintptr_t)p + bytes : (uintptr_t)p;   // empty when there is nothing behind the pointer
    r.write = write;
    return r;
}

class FlowHazards {
public:
    static constexpr int kMaxSlots = 4;
    // a slot whose list reaches this length should be checked for idleness (and, at 4x this, waited for) before it grows further
    static constexpr size_t kPruneAt = 64;
    static constexpr int kDrainFirst = -1;

    // The slot a batch touching r[0..n) has to go to: `turn` (the caller's round-robin choice) when nothing in flight
    // conflicts, the one conflicting slot, or kDrainFirst when batches of more than one slot conflict.
    int route(int n_slots, int turn, const FlowRange* r, int n) const {
        int hit = -2;
        for (int s = 0; s < n_slots && s < kMaxSlots; ++s) {
            if (!conflicts(held_[s], r, n)) continue;
            if (hit >= 0) return kDrainFirst;
            hit = s;
        }
        return hit >= 0 ? hit : turn;
    }
    // the batch has been enqueued on `slot`
    void note(int slot, const FlowRange* r, int n) {
        std::vector<FlowRange>& h = held_[slot];
        for (int i = 0; i < n; ++i) {
            if (r[i].hi <= r[i].lo) continue;
            bool merged = false;
            for (FlowRange& o : h) {   // callers that alternate a few buffers keep the lists at a few entries
                if (o.lo <= r[i].lo && r[i].hi <= o.hi && (o.write || !r[i].write)) { merged = true; break; }
                if (r[i].lo <= o.lo && o.hi <= r[i].hi && (r[i].write || !o.write)) { o = r[i]; merged = true; break; }
            }
            if (!merged) h.push_back(r[i]);
        }
    }
    size_t held(int slot) const { return held_[slot].size(); }
    void clear_slot(int slot) { held_[slot].clear(); }   // the slot's stream is idle
    void clear() {
        for (auto& h : held_) h.clear();
    }

private:
    static bool conflicts(const std::vector<FlowRange>& h, const FlowRange* r, int n) {
        for (const FlowRange& o : h)
            for (int i = 0; i < n; ++i)
                if (r[i].lo < r[i].hi && r[i].lo < o.hi && o.lo < r[i].hi && (r[i].write || o.write)) return true;
        return false;
    }
    std::vector<FlowRange> held_[kMaxSlots];
};

}  // namespace latok
