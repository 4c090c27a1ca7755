// wl_view.h -- device view of one library's whitelist tables + the exact-lookup primitive.
#pragma once
#include <cstdint>

struct WlView {
    const uint32_t *offA;
    const uint16_t *tailA;
    const uint32_t *valA;  // nullptr => rank == sorted position
    const uint32_t *offB;
    const uint16_t *headB;
    uint32_t *valid;
    uint32_t *corrected;
    const uint32_t *prior;
    uint32_t bitsA, bitsB;  // head / tail bits
    uint32_t n;             // 0 => library not configured
    uint32_t pad;
};

// Exact whitelist membership (Whitelist::check_and_update, barcode/src/whitelist.rs:494-517):
// returns the canonical rank of the (translated) barcode or 0xFFFFFFFF.
__device__ __forceinline__ uint32_t wl_lookup(const WlView &w, uint32_t key) {
    const uint32_t head = key >> w.bitsB;  // bitsB <= 16 < 32 always; bitsA may be 0
    const uint32_t tail = key & ((1u << w.bitsB) - 1u);
    uint32_t lo = w.offA[head];
    const uint32_t hi = w.offA[head + 1];
    for (; lo < hi; ++lo) {
        const uint32_t t = w.tailA[lo];
        if (t >= tail) {
            if (t == tail) return w.valA ? w.valA[lo] : lo;
            break;
        }
    }
    return 0xFFFFFFFFu;
}
