// wl_view.h -- device view of one library's whitelist tables + the lookup primitives.
#pragma once
#include <cstdint>

struct WlView {
    const uint32_t *offA;
    const uint16_t *tailA;  // n entries (+2 of padding), 4-byte aligned base
    const uint32_t *valA;   // nullptr => rank == sorted position
    const uint32_t *offB;
    const uint16_t *headB;  // n entries (+2 of padding), 4-byte aligned base
    uint32_t *valid;
    uint32_t *corrected;
    const uint32_t *prior;
    const uint32_t *offE;   // exact-lookup index: bins of the top (key bits - shiftE) bits over tailA
    uint32_t bitsA, bitsB;  // head / tail bits
    uint32_t n;             // 0 => library not configured
    uint32_t shiftE;
};

// multi-dword loads from 4-byte aligned addresses (one memory instruction, one address per lane)
struct __attribute__((aligned(4))) U32x2 {
    uint32_t a, b;
};
struct __attribute__((aligned(4))) U32x4 {
    uint32_t w[4];
};

// Visit the u16 entries arr[lo, hi) in rounds of 16: the 8 dword loads of a round are independent
// (one memory latency per round instead of one per entry -- a per-entry load/compare/branch loop is a
// serial latency chain and made the lookups latency bound).  f(value, position) is called for every
// entry of the range.  arr must be 4-byte aligned and padded by 2 entries.
template <uint32_t DWORDS = 8, typename F>
__device__ __forceinline__ void scan_u16_range(const uint16_t *__restrict__ arr, uint32_t lo, uint32_t hi, F f) {
    const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(arr);
    for (uint32_t p = lo & ~1u; p < hi; p += 2u * DWORDS) {
        uint32_t d[DWORDS];
#pragma unroll
        for (uint32_t k = 0; k < DWORDS; k++) d[k] = (p + 2u * k < hi) ? w[(p >> 1) + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 2u * DWORDS; k++) {
            const uint32_t pos = p + k;
            if (pos >= lo && pos < hi) f((d[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu, pos);
        }
    }
}

// Exact whitelist membership (Whitelist::check_and_update, barcode/src/whitelist.rs:494-517):
// returns the canonical rank of the (translated) barcode or 0xFFFFFFFF.
__device__ __forceinline__ uint32_t wl_lookup(const WlView &w, uint32_t key) {
    // the exact index has ~1.4 keys per bin (the pigeonhole bins of table A hold ~11 and needed two
    // rounds of 8 loads per read): its top bits include the whole head, so comparing the stored tail
    // (the low bitsB bits) inside the bin is an exact test
    const uint32_t bin = (uint32_t)((uint64_t)key >> w.shiftE);
    const uint32_t tail = key & ((1u << w.bitsB) - 1u);
    const uint32_t lo = w.offE[bin];
    const uint32_t hi = w.offE[bin + 1];
    uint32_t found = 0xFFFFFFFFu;
    scan_u16_range<4>(w.tailA, lo, hi, [&](uint32_t t, uint32_t pos) {
        if (t == tail) found = pos;
    });
    if (found == 0xFFFFFFFFu) return found;
    return w.valA ? w.valA[found] : found;
}
