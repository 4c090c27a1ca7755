// wl_view.h -- device view of one library's whitelist tables + the lookup primitives.
#pragma once
#include <cstdint>

struct WlView {
    const uint32_t *offA;
    const uint16_t *tailA;  // n entries (+2 of padding), 4-byte aligned base
    const uint32_t *valA;   // nullptr => rank == sorted position
    const uint32_t *offB;
    const uint16_t *headB;  // n entries (+2 of padding), 4-byte aligned base
    const uint32_t *valB;   // rank of every entry of table B
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

// Visit the u16 entries arr[lo, hi) in rounds of 2*DWORDS entries fetched with 16-byte loads (one memory
// latency per round instead of one per entry -- a per-entry load/compare/branch loop is a serial latency
// chain and made the lookups latency bound).  f(value, position) is called for every entry of the range.
// arr must be 4-byte aligned and padded by 32 bytes.
template <uint32_t DWORDS = 8, typename F>
__device__ __forceinline__ void scan_u16_range(const uint16_t *__restrict__ arr, uint32_t lo, uint32_t hi, F f) {
    static_assert(DWORDS % 4 == 0, "rounds are made of 16-byte loads");
    const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(arr);
    for (uint32_t p = lo & ~1u; p < hi; p += 2u * DWORDS) {
        U32x4 v[DWORDS / 4];  // unconditional: the tables are padded by 32 bytes
#pragma unroll
        for (uint32_t k = 0; k < DWORDS / 4; k++) v[k] = *reinterpret_cast<const U32x4 *>(w + (p >> 1) + 4u * k);
#pragma unroll
        for (uint32_t k = 0; k < 2u * DWORDS; k++) {
            const uint32_t pos = p + k;
            if (pos >= lo && pos < hi) f((v[k >> 3].w[(k >> 1) & 3u] >> (16u * (k & 1u))) & 0xFFFFu, pos);
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
    // both bounds of the bin in one 8-byte load, then its first 8 entries in one 16-byte load (the tables
    // are padded by 32 bytes): two memory instructions and two cache lines per lookup
    const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offE + bin);
    const uint32_t lo = b2.a, hi = b2.b;
    uint32_t found = 0xFFFFFFFFu;
    if (hi > lo) {
        const uint32_t p0 = lo & ~1u;
        const U32x4 d = *reinterpret_cast<const U32x4 *>(reinterpret_cast<const uint32_t *>(w.tailA) + (lo >> 1));
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t pos = p0 + k;
            const uint32_t t = (d.w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
            if (pos >= lo && pos < hi && t == tail) found = pos;
        }
        if (hi > p0 + 8u && found == 0xFFFFFFFFu)  // a bin of more than 7 keys: rare
            scan_u16_range<4>(w.tailA, p0 + 8u, hi, [&](uint32_t t, uint32_t pos) {
                if (t == tail) found = pos;
            });
    }
    if (found == 0xFFFFFFFFu) return found;
    return w.valA ? w.valA[found] : found;
}
