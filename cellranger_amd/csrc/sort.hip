// sort.hip -- LSD radix sort of 64-bit molecule keys / 32-bit group hashes (optionally with a 32-bit
// payload) built on wave64 ballot multisplit, plus the small device-wide scan used by the compactions.
//
// This is the "sort" the reference gets from shardio's sorted shards + std HashMap grouping
// (cr_lib/src/barcode_sort.rs:97-162, par_proc.rs:131-152); integer work, HBM bound: every pass
// streams the keys once for the digit histogram and once for the scatter.
//
// 64-bit keys without payload (the molecule-key sort): "onesweep" -- k_global_hist counts the digits of ALL
// passes in one read of the keys, then one k_radix_scatter<..., ONESWEEP> launch per pass whose chunks get
// their global offsets from a decoupled look-back (no per-pass histogram read).  CRGPU_SORT=classic selects
// the path below for these keys too.
// Other sorts (32-bit hashes + payload, keys + read ordinals, the partition by owner rank): one pass = 3 launches:
//   k_radix_hist     per-block digit histogram           -> block_hist[digit][block]
//   k_scan_digits    one workgroup per digit: exclusive scan over blocks + digit total
//   k_radix_scatter  stable scatter; each block derives its digit bases from the digit totals
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "block_utils.h"
#include <algorithm>

#include "common.h"

// One workgroup of 1024 threads per CU with a 16 K-key chunk (128 KB of LDS).  What bounds the scatter is
// the efficiency of its writes: every resident workgroup keeps one partially written cache line open per
// digit, and a chunk adds CHUNK/256 keys to it.  With many small workgroups (256 threads, 4 K keys, four
// per CU) the open lines (1024 x 256 x 128 B = 32 MB) do not survive in the 4 MB L2s between two chunks and
// runs of 128 B go out as partial lines; with one large workgroup per CU the open set is 8 MB and the runs
// are 512 B.  Measured at 400 M reads: 13.1 -> 10.9 ms per step going from 4 K to 8 K keys at 256 threads.
// A/B switches for the scatter's key stream: -DSORT_NT_IN / -DSORT_NT_OUT use non-temporal loads / stores
#ifdef SORT_NT_IN
#define SORT_LOAD_IN(p) __builtin_nontemporal_load(p)
#else
#define SORT_LOAD_IN(p) (*(p))
#endif
#ifdef SORT_NT_OUT
#define SORT_STORE_OUT(v, p) __builtin_nontemporal_store((v), (p))
#else
#define SORT_STORE_OUT(v, p) (*(p) = (v))
#endif
#ifndef SORT_BLOCK
#define SORT_BLOCK 1024
#endif
#ifndef SORT_ITEMS
#define SORT_ITEMS 16      // keys per thread per chunk
#endif
#ifndef SORT_ITEMS_KV64
#define SORT_ITEMS_KV64 10  // 64-bit keys with a payload: 12 bytes per element, 10 K elements = 120 KB of LDS (8 K: 7 x 6.1 ms at 1 B records)
#endif
#define SORT_WAVES (SORT_BLOCK / 64)
#define RADIX_BITS 8
#define RADIX 256
#ifndef SORT_LB
#define SORT_LB 8  // predecessor statuses read per round trip of the onesweep look-back
#endif
#ifndef SORT_MAX_BLOCKS
#define SORT_MAX_BLOCKS 1024
#define OR_MAX_LOW_BITS 16u  // low key bits the finishing step may be left with (two levels of bucket arrays in or_run_through_memory)
#endif

// ---- exclusive scan of a small u32 array (block counts of the compactions), single workgroup -----
__global__ __launch_bounds__(1024) void k_scan_small(uint32_t *__restrict__ data, uint64_t n,
                                                     uint32_t *__restrict__ total_out) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    // coalesced rounds of 1024 elements with a running carry
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + tid;
        const uint32_t v = i < n ? data[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        uint32_t wpre = 0, tot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            const uint32_t t = wave_sums[w];
            if (w < wave) wpre += t;
            tot += t;
        }
        const uint32_t carry = carry_s;
        if (i < n) data[i] = carry + wpre + x - v;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0 && total_out) *total_out = carry_s;
}

int cr_scan_small(crgpu_ctx *ctx, uint32_t *d_data, uint64_t n, uint32_t *d_total_out) {
    CrTimer t(ctx, CRGPU_T_SCAN);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, d_data, n, d_total_out);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ---- digit extraction -------------------------------------------------------------------------
// RadixDigit   (key >> shift) & mask              (radix pass)
// OwnerDiv     (key >> shift) / width             (owner rank of the barcode, crgpu_partition_keys_dev:
//                                                  contiguous barcode-rank ranges, like shardio's make_chunks)
// OwnerBounds  index of the range [bounds[r], bounds[r+1]) that holds (key >> shift); `bounds` has
//              width+1 ascending entries in device memory (histogram-balanced ranges)
// PayloadDigit (payload >> shift) & mask         (partition by the top bits of the read ordinal that travels as payload)
// Every functor is called as dig(key, payload); only PayloadDigit looks at the payload.
struct RadixDigit {
    uint32_t shift, mask;
    template <typename K>
    __device__ __forceinline__ uint32_t operator()(K key, uint32_t = 0u) const {
        return (uint32_t)(key >> shift) & mask;
    }
};
struct PayloadDigit {
    uint32_t shift, mask;
    template <typename K>
    __device__ __forceinline__ uint32_t operator()(K, uint32_t val) const {
        return (val >> shift) & mask;
    }
};
struct OwnerDiv {
    uint32_t shift, width;
    template <typename K>
    __device__ __forceinline__ uint32_t operator()(K key, uint32_t = 0u) const {
        return (uint32_t)((uint64_t)key >> shift) / width;
    }
};
struct OwnerBounds {
    uint32_t shift, width;
    const uint32_t *bounds;
    template <typename K>
    __device__ __forceinline__ uint32_t operator()(K key, uint32_t = 0u) const {
        const uint32_t v = (uint32_t)((uint64_t)key >> shift);
        uint32_t lo = 0, hi = width;  // first r with bounds[r+1] > v
        while (lo + 1 < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (bounds[mid] <= v) lo = mid; else hi = mid;
        }
        return lo;
    }
};

template <typename K, bool HAS_VALS>
struct SortCfg {
    static constexpr int ITEMS = (sizeof(K) == 8 && HAS_VALS) ? SORT_ITEMS_KV64 : SORT_ITEMS;
    static constexpr uint32_t CHUNK = SORT_BLOCK * ITEMS;
    // LDS: staged keys (+ payloads), per-wave digit counts (u16: they stay below CHUNK <= 16384), digit bases,
    // global deltas, scan scratch
    static constexpr size_t lds_bytes(int bits) {
        return (size_t)CHUNK * (sizeof(K) + (HAS_VALS ? 4 : 0)) + (size_t)SORT_WAVES * (1u << bits) * 2 +
               2 * (size_t)(1u << bits) * 4 + SORT_WAVES * 4 + 16;
    }
};

// ---- pass 1: per-block digit histogram ------------------------------------------------------------
template <typename K, typename DIG, int BITS>
__global__ __launch_bounds__(SORT_BLOCK) void k_radix_hist(const K *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                           uint64_t n, uint64_t tile, DIG dig,
                                                           uint32_t *__restrict__ block_hist, uint32_t n_blocks) {
    constexpr uint32_t RADIX_T = 1u << BITS;
    __shared__ uint32_t h[RADIX_T];
    if (threadIdx.x < RADIX_T) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < n ? lo + tile : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += SORT_BLOCK) atomicAdd(&h[dig(keys[i], vals ? vals[i] : 0u)], 1u);
    __syncthreads();
    if (threadIdx.x < RADIX_T) block_hist[(uint64_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// ---- pass 2: one workgroup per digit scans that digit's row of block counts ------------------------
__global__ __launch_bounds__(256) void k_scan_digits(uint32_t *__restrict__ block_hist, uint32_t n_blocks,
                                                     uint32_t *__restrict__ digit_totals) {
    __shared__ uint32_t lds[8];
    uint32_t *row = block_hist + (uint64_t)blockIdx.x * n_blocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? row[i] : 0u;
        uint32_t tot;
        const uint32_t pre = block_excl_scan_256(v, lds, &tot);
        if (i < n_blocks) row[i] = carry + pre;
        carry += tot;
    }
    if (threadIdx.x == 0) digit_totals[blockIdx.x] = carry;
}

// ---- pass 3: stable scatter ------------------------------------------------------------------------
// Order inside a chunk: wave-major, then item, then lane, so a wave owns 64*ITEMS consecutive keys.
// The chunk is first ranked by digit inside LDS (ballot multisplit: rank of a key = same-digit keys of
// earlier waves + of this wave's earlier items + of lower lanes), then copied out so that neighbouring
// threads write neighbouring addresses: the global stores are contiguous runs per digit (a direct
// scatter issues ~56 separate 8-byte stores per wave instruction).
//
// The ranking loop is the ALU-heavy part (the kernel is issue-bound, not HBM-bound, when written
// naively), so it is kept lean: per digit bit one v_bfe_i32 + one ballot + xnor/and on the two mask
// halves, ranks from v_mbcnt, and the per-(wave, digit) running count is a plain LDS read by every lane
// followed by a write from the lowest lane of each digit group (a wave runs in lockstep and its LDS
// operations complete in order, so no atomic or cross-lane shuffle is needed).
//
// ONESWEEP: no per-block histogram pass.  Chunks are handed out by a ticket counter (so every chunk's
// predecessors are already running or done), `digit_totals` holds the EXCLUSIVE prefix of the global digit
// counts of this pass (k_global_hist counts all passes in one read of the keys), and the number of same-digit
// keys in earlier chunks comes from a decoupled look-back over `status`: one 64-bit word per (chunk, digit),
// flag in the top two bits (1 = this chunk's count, 2 = inclusive prefix up to this chunk), published with
// agent-scope atomic stores.  A chunk publishes its counts before it starts waiting, and waits only on lower
// tickets, so the chain always makes progress.
// -DSORT_PHASE_TIMING: thread 0 of every workgroup adds the shader-clock ticks of each phase of a chunk to g_sort_phase
// (attribution builds only, scripts/sort_phases.py; crgpu_debug_sort_phases reads and clears the counters)
#ifdef SORT_PHASE_TIMING
__device__ unsigned long long g_sort_phase[16];
#define SORT_PH(i)                                                          \
    do {                                                                    \
        if (tid == 0) {                                                     \
            const unsigned long long t_ = __builtin_readcyclecounter();     \
            atomicAdd(&g_sort_phase[i], t_ - ph_t);                         \
            ph_t = t_;                                                      \
        }                                                                   \
    } while (0)
extern "C" int crgpu_debug_sort_phases(unsigned long long *out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sort_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sort_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
#define SORT_PH(i) ((void)0)
#endif
#define OS_AGG (1ull << 62)
#define OS_INC (2ull << 62)
#define OS_VAL ((1ull << 62) - 1ull)
#ifndef OS_GROUP
#define OS_GROUP 8u  // consecutive chunks handed to one XCD
#endif
// the ticket of a workgroup's next chunk: a counter per XCD (n_xcc > 1) or one global sequence; 0xFFFFFFFF once a pass has aborted
__device__ __forceinline__ uint32_t take_ticket_fn(uint32_t *abort_word, uint32_t *ticket, uint32_t n_xcc) {
    uint32_t xcc = 0;
    if (n_xcc > 1) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xFu;
        if (xcc >= n_xcc) xcc %= n_xcc;
    }
    const uint32_t ab = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t k = atomicAdd(ticket + xcc * 32u, 1u);
    const uint32_t c = n_xcc > 1 ? ((k / OS_GROUP) * n_xcc + xcc) * OS_GROUP + (k % OS_GROUP) : k;
    return ab != 0u ? 0xFFFFFFFFu : c;
}
// ST32: status words of 32 bits (flag in the top two, counts below 2^30) for sorts of fewer than 2^30 keys -- the look-back
// reads ~21 predecessor rows per chunk (scripts/sort_phases.py), a third of the bytes of the keys themselves with 64-bit
// words: 25.8 -> 24.9 ms for the seven passes over 796 M keys.
template <typename K, bool HAS_VALS, typename DIG, int BITS, bool ONESWEEP = false, bool ST32 = false>
__global__ __launch_bounds__(SORT_BLOCK) void k_radix_scatter(const K *__restrict__ keys_in, K *__restrict__ keys_out,
                                                              const uint32_t *__restrict__ vals_in,
                                                              uint32_t *__restrict__ vals_out, uint64_t n, uint64_t tile,
                                                              DIG dig, const uint32_t *__restrict__ block_offs,
                                                              const uint32_t *__restrict__ digit_totals,
                                                              uint32_t n_blocks, unsigned long long *__restrict__ status_,
                                                              uint32_t *__restrict__ ticket, uint32_t *__restrict__ abort_word,
                                                              uint32_t pass_tag) {
    using os_word = typename std::conditional<ST32, uint32_t, unsigned long long>::type;
    constexpr int OS_FSHIFT = ST32 ? 30 : 62;
    constexpr os_word W_AGG = (os_word)1 << OS_FSHIFT, W_INC = (os_word)2 << OS_FSHIFT, W_VAL = W_AGG - (os_word)1;
    os_word *__restrict__ status = reinterpret_cast<os_word *>(status_);
    constexpr int ITEMS = SortCfg<K, HAS_VALS>::ITEMS;
    constexpr uint32_t CHUNK = SortCfg<K, HAS_VALS>::CHUNK;
    constexpr uint32_t RADIX_T = 1u << BITS;
    static_assert(RADIX_T <= SORT_BLOCK, "one thread per digit in the per-digit phases");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K *skeys = reinterpret_cast<K *>(smem);                                                    // CHUNK
    uint32_t *svals = reinterpret_cast<uint32_t *>(smem + (size_t)CHUNK * sizeof(K));         // CHUNK if HAS_VALS
    uint16_t(*wcount)[RADIX_T] = reinterpret_cast<uint16_t(*)[RADIX_T]>(svals + (HAS_VALS ? CHUNK : 0));
    // wcount: per-wave same-digit counts -> LDS position of (wave, digit)
    uint32_t *base = reinterpret_cast<uint32_t *>(&wcount[SORT_WAVES][0]);  // running global offset of each digit
    uint32_t *gdelta = base + RADIX_T;        // global position = LDS position + gdelta[digit]
    uint32_t *lds = gdelta + RADIX_T;         // SORT_WAVES words of scan scratch
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    uint32_t *s_chunk = lds + SORT_WAVES;     // ONESWEEP: the ticket of the current chunk
    uint32_t *s_abort = s_chunk + 1;          // ONESWEEP: a waiter of this workgroup saw (or raised) the abort word
    if (ONESWEEP && tid == 0) *s_abort = 0u;  // ordered before its first use by the barrier behind the ticket

    if (!ONESWEEP) {
        const uint32_t digit_base = block_excl_scan<SORT_BLOCK>(tid < RADIX_T ? digit_totals[tid] : 0u, lds, nullptr);
        if (tid < RADIX_T) base[tid] = digit_base + block_offs[(uint64_t)tid * n_blocks + blockIdx.x];
    }
    const uint64_t lo = ONESWEEP ? 0 : (uint64_t)blockIdx.x * tile;
    const uint64_t hi = ONESWEEP ? n : (lo + tile < n ? lo + tile : n);
    const uint64_t n_chunks = (n + CHUNK - 1) / CHUNK;

#ifdef SORT_PHASE_TIMING
    unsigned long long ph_t = __builtin_readcyclecounter();
#endif
    bool have_next = false;  // ONESWEEP: *s_chunk already holds the ticket of the chunk to come
    for (uint64_t chunk = lo;; chunk += CHUNK) {
        uint64_t cidx = 0;
        if (ONESWEEP) {
            // n_blocks carries the number of XCDs here.  Each XCD hands out its own share of the chunks: groups of
            // OS_GROUP consecutive chunks go round-robin over the XCDs, so that the boundary lines of neighbouring
            // chunks are mostly written through one L2, which merges them (chunks of one global ticket sequence land
            // on arbitrary XCDs and every run boundary went to HBM as two partial lines).  The look-back still
            // follows the global chunk order; a chunk's predecessors are either earlier tickets of its own XCD or
            // chunks of other XCDs that their counters reach without waiting on anything later.
            // abort_word (shared by all passes of one sort): 0, or 1 + the pass whose look-back watchdog fired.  Once it is
            // set nobody takes another chunk -- of this pass or of the passes queued behind it -- so the buffers stay as
            // the failed pass found them and the host redoes the sort from that pass on with the classic kernels.
            // (the abort word and the ticket are requested together: two dependent round trips were 6 % of a chunk's time;
            // a ticket taken after an abort is harmless -- nobody works on it and the redo does not use tickets)
            // Every chunk but a workgroup's first gets its ticket while the chunk before it is copied out (thread 0 asks before
            // the copy-out loop and puts the value into *s_chunk behind it): all workgroups shift alike, so nobody waits longer
            // for a predecessor (24.8 -> 24.2 ms for seven passes over 796 M keys).  Requesting the next chunk's KEYS there as
            // well, into the registers the LDS scatter has freed, costs 45 % (36 ms; profiles/r02_b_sort_phases_and_streams.txt) -- the second
            // time this was measured: reads issued in front of the copy-out stores hold the stores up.
            if (!have_next) {
                if (tid == 0) *s_chunk = take_ticket_fn(abort_word, ticket, n_blocks);
                __syncthreads();
            }
            cidx = *s_chunk;
            if (cidx >= n_chunks) break;  // uniform
            chunk = cidx * CHUNK;
        } else if (chunk >= hi) {
            break;
        }
        SORT_PH(0);  // ticket
        // every wave clears its own row of counters: the row's last readers of the previous chunk were this wave (LDS scatter)
        // and the per-digit threads two barriers ago, so no barrier is needed here (-0.13 ms per 1 B reads)
        {
            uint32_t *row = reinterpret_cast<uint32_t *>(&wcount[wave][0]);
            for (uint32_t x = lane; x < RADIX_T / 2; x += 64) row[x] = 0u;
        }
        SORT_PH(1);  // counters cleared

        K key[ITEMS];
        uint32_t val[ITEMS];
        uint32_t dr[ITEMS];  // (digit << 16) | rank inside (wave, digit); rank < 64 * ITEMS
        const uint32_t chunk_n = hi - chunk < CHUNK ? (uint32_t)(hi - chunk) : CHUNK;
        const uint32_t wave_off = wave * (64 * ITEMS) + lane;  // chunk-local index of item 0
        const K *kin = keys_in + chunk + wave_off;
        const uint32_t *vin = HAS_VALS ? vals_in + chunk + wave_off : nullptr;
        if (chunk_n == CHUNK) {
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                key[it] = SORT_LOAD_IN(&kin[it * 64]);
                if (HAS_VALS) val[it] = vin[it * 64];
            }
#ifdef SORT_PHASE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SORT_PH(2);  // keys arrived
#endif
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                const uint32_t d = dig(key[it], HAS_VALS ? val[it] : 0u);
                dr[it] = (d << 16) | wave_multisplit_rank<BITS, true>(d, true, wcount[wave]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                const bool ok = wave_off + it * 64 < chunk_n;
                key[it] = ok ? kin[it * 64] : (K)0;
                if (HAS_VALS) val[it] = ok ? vin[it * 64] : 0u;
            }
#pragma unroll
            for (int it = 0; it < ITEMS; it++) {
                const bool ok = wave_off + it * 64 < chunk_n;
                const uint32_t d = ok ? dig(key[it], HAS_VALS ? val[it] : 0u) : 0u;
                dr[it] = (d << 16) | wave_multisplit_rank<BITS, false>(d, ok, wcount[wave]);
            }
        }
        SORT_PH(3);  // own wave ranked
        __syncthreads();
        SORT_PH(4);  // all waves ranked
        // one thread per digit: chunk-local start of the digit (exclusive scan over digits), per-wave
        // starts inside it, and the shift from LDS position to global position
        uint32_t tot = 0, run0 = 0;
        {
            if (tid < RADIX_T)
                for (int w = 0; w < SORT_WAVES; w++) tot += wcount[w][tid];
            // (pass_tag bit 31: test switch CRGPU_SORT_FORCE_ABORT -- chunk 0 never publishes, the chain stalls)
            if (ONESWEEP && tid < RADIX_T && !((pass_tag >> 31) && cidx == 0))  // let the successors go on as early as possible
                __hip_atomic_store(&status[cidx * RADIX_T + tid], (cidx == 0 ? W_INC : W_AGG) | (os_word)tot, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            // (no trailing barrier: the scratch words are next written in the next chunk's scan, three barriers from here:
            // 24.1 -> 23.7 ms per 1 B reads.  The barrier at the END of the loop has to stay although nothing in LDS needs it:
            // without it fast waves request the next chunk's keys while others still copy out, +0.2 ms)
            uint32_t run = block_excl_scan<SORT_BLOCK, false>(tot, lds, nullptr);  // chunk-local start of digit tid
            run0 = run;
            if (tid < RADIX_T) {
                for (int w = 0; w < SORT_WAVES; w++) {
                    const uint32_t c = wcount[w][tid];
                    wcount[w][tid] = (uint16_t)run;
                    run += c;
                }
            }
        }
        __syncthreads();
        SORT_PH(5);  // digit scan
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            if (chunk_n == CHUNK || wave_off + it * 64 < chunk_n) {
                const uint32_t p = wcount[wave][dr[it] >> 16] + (dr[it] & 0xFFFFu);
                skeys[p] = key[it];
                if (HAS_VALS) svals[p] = val[it];
            }
        }
        SORT_PH(6);  // LDS scatter issued
        // global start of every digit's run.  ONESWEEP: the look-back comes as late as possible (after the LDS
        // scatter above) so that the predecessors have had time to publish
        if (tid < RADIX_T) {
            uint32_t g;
            if (ONESWEEP) {
                os_word excl = 0;
                if (cidx > 0) {
                    // Watchdog: the chain cannot stall as long as every XCD receives workgroups (each of them takes its
                    // XCD's chunks in ascending order).  Should that ever not hold (CU masking, a shared device), a waiter
                    // gives up after ~2^22 polls and raises the abort word; everybody stops waiting and stops taking
                    // chunks, and the host redoes the sort from this pass on with the classic kernels -- no hung GPU,
                    // no failed call.
                    const uint32_t poll_limit = (pass_tag >> 31) ? (1u << 16) : (1u << 22);  // the forced stall need not take seconds
#ifdef SORT_PHASE_TIMING
                    uint32_t ph_polls = 0, ph_depth = 0;
#endif
                    auto wait_for = [&](uint64_t c, os_word sv) {
                        uint32_t polls = 0;
                        while ((sv >> OS_FSHIFT) == 0) {
                            __builtin_amdgcn_s_sleep(1);
                            sv = __hip_atomic_load(&status[c * RADIX_T + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SORT_PHASE_TIMING
                            ph_polls++;
#endif
                            if ((++polls & 0xFFFu) == 0u) {
                                if (polls >= poll_limit) atomicCAS(abort_word, 0u, pass_tag & 0x7FFFFFFFu);
                                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                    *s_abort = 1u;
                                    return W_INC;
                                }
                            }
                        }
                        return sv;
                    };
                    // SORT_LB statuses per round trip (the loads of a round are independent).  Wider rounds are SLOWER: 16 per
                    // round +1.0 ms, 32 per round +3.1 ms for the seven passes over 796 M keys (profiles/r02_b_sort_phases_and_streams.txt)
                    // -- every status word read crosses the fabric (other XCDs wrote it), and a round of 32 reads as many
                    // bytes as the chunk's own keys.
                    uint64_t p = cidx;  // chunks [0, p) are still to be added; chunk 0 always carries an inclusive prefix
                    bool done = false;
                    while (!done) {
                        const uint32_t nb = p < SORT_LB ? (uint32_t)p : (uint32_t)SORT_LB;
                        os_word v[SORT_LB];
#pragma unroll
                        for (uint32_t j = 0; j < SORT_LB; j++)
                            v[j] = j < nb ? __hip_atomic_load(&status[(p - 1 - j) * RADIX_T + tid], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT)
                                          : (os_word)0;
#pragma unroll
                        for (uint32_t j = 0; j < SORT_LB; j++) {
                            if (done || j >= nb) continue;
                            const os_word x = wait_for(p - 1 - j, v[j]);
#ifdef SORT_PHASE_TIMING
                            ph_depth++;
#endif
                            excl += x & W_VAL;
                            done = (x >> OS_FSHIFT) == 2;
                        }
                        p -= nb;
                    }
                    __hip_atomic_store(&status[cidx * RADIX_T + tid], W_INC | (os_word)(excl + tot), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
#ifdef SORT_PHASE_TIMING
                    if (tid == 0) {  // [11] polls of statuses that were not published yet, [12] predecessors walked, [13] chunks
                        atomicAdd(&g_sort_phase[11], (unsigned long long)ph_polls);
                        atomicAdd(&g_sort_phase[12], (unsigned long long)ph_depth);
                        atomicAdd(&g_sort_phase[13], 1ull);
                    }
#endif
                }
                g = digit_totals[tid] + (uint32_t)excl;
            } else {
                g = base[tid];
                base[tid] = g + tot;
            }
            gdelta[tid] = g - run0;
        }
        SORT_PH(7);  // look-back of thread 0
        __syncthreads();
        SORT_PH(8);  // look-back of all digits
        if (ONESWEEP && *s_abort) break;  // uniform: nothing of this chunk is written, the pass is redone
        // the next ticket travels during the copy-out (everybody read this chunk's ticket at the top of the loop)
        uint32_t next_ticket = 0;
        if (ONESWEEP && tid == 0) next_ticket = take_ticket_fn(abort_word, ticket, n_blocks);
        for (uint32_t p = tid; p < chunk_n; p += SORT_BLOCK) {
            const K k = skeys[p];
            const uint32_t pos = p + gdelta[dig(k, HAS_VALS ? svals[p] : 0u)];
            SORT_STORE_OUT(k, &keys_out[pos]);
            if (HAS_VALS) vals_out[pos] = svals[p];
        }
        if (ONESWEEP && tid == 0) *s_chunk = next_ticket;
        have_next = true;

        SORT_PH(9);  // copy-out issued
        __syncthreads();
        SORT_PH(10);  // everybody's copy-out issued
    }
}

static uint32_t sort_blocks(uint64_t n, uint64_t chunk, uint64_t *tile_out) {
    uint64_t nb = (n + chunk * 4 - 1) / (chunk * 4);
    if (nb < 1) nb = 1;
    if (nb > SORT_MAX_BLOCKS) nb = SORT_MAX_BLOCKS;
    uint64_t tile = (n + nb - 1) / nb;
    tile = (tile + chunk - 1) / chunk * chunk;
    nb = (n + tile - 1) / tile;
    if (nb < 1) nb = 1;
    *tile_out = tile;
    return (uint32_t)nb;
}

static uint32_t *digit_totals_buf(crgpu_ctx *ctx) { return ctx->d_sort_hist + 256 * 2048; }  // RADIX_MAX u32 behind the block histograms

// one counting-sort pass keyed by `dig` (stable).
// slot >= 0: the timing slot all three kernels are booked under (without units)
template <typename K, typename DIG, int BITS = RADIX_BITS>
static int radix_pass(crgpu_ctx *ctx, const K *d_in, K *d_out, const uint32_t *d_vin, uint32_t *d_vout, uint64_t n,
                      DIG dig, int slot = -1) {
    uint64_t tile;
    const uint32_t nb = sort_blocks(n, d_vin ? SortCfg<K, true>::CHUNK : SortCfg<K, false>::CHUNK, &tile);
    uint32_t *d_hist = ctx->d_sort_hist;
    uint32_t *d_tot = digit_totals_buf(ctx);
    {
        // the auxiliary 32-bit sort of the low-support stage is booked under that stage: the SORT slots are
        // the 64-bit molecule-key sort alone (bench.py prices them at 8 / 16 bytes per key)
        CrTimer t(ctx, slot >= 0 ? slot : (sizeof(K) == 8 ? CRGPU_T_SORT_HIST : CRGPU_T_DEDUP), (slot < 0 && sizeof(K) == 8) ? n : 0);
        hipLaunchKernelGGL((k_radix_hist<K, DIG, BITS>), dim3(nb), dim3(SORT_BLOCK), 0, ctx->stream, d_in, d_vin, n, tile, dig, d_hist, nb);
    }
    {
        CrTimer t(ctx, slot >= 0 ? slot : CRGPU_T_SCAN);
        hipLaunchKernelGGL(k_scan_digits, dim3(1u << BITS), dim3(256), 0, ctx->stream, d_hist, nb, d_tot);
    }
    CrTimer t(ctx, slot >= 0 ? slot : (sizeof(K) == 8 ? CRGPU_T_SORT : CRGPU_T_DEDUP), (slot < 0 && sizeof(K) == 8) ? n : 0);  // DEDUP counts its keys once
    // more than 64 KB of LDS per workgroup has to be requested per kernel, once
    const size_t lds_kv = SortCfg<K, true>::lds_bytes(BITS), lds_k = SortCfg<K, false>::lds_bytes(BITS);
    if (d_vin) {
        cr_allow_lds(ctx, (const void *)k_radix_scatter<K, true, DIG, BITS>, lds_kv);
        hipLaunchKernelGGL((k_radix_scatter<K, true, DIG, BITS>), dim3(nb), dim3(SORT_BLOCK), lds_kv, ctx->stream,
                           d_in, d_out, d_vin, d_vout, n, tile, dig, d_hist, d_tot, nb, nullptr, nullptr, nullptr, 0u);
    } else {
        cr_allow_lds(ctx, (const void *)k_radix_scatter<K, false, DIG, BITS>, lds_k);
        hipLaunchKernelGGL((k_radix_scatter<K, false, DIG, BITS>), dim3(nb), dim3(SORT_BLOCK), lds_k, ctx->stream,
                           d_in, d_out, d_vin, d_vout, n, tile, dig, d_hist, d_tot, nb, nullptr, nullptr, nullptr, 0u);
    }
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ---- onesweep: all passes' digit histograms in one read of the keys -------------------------------------
template <typename K>
__global__ __launch_bounds__(SORT_BLOCK) void k_global_hist(const K *__restrict__ keys, uint64_t n, SweepPlan plan,
                                                            uint32_t *__restrict__ ghist /* [pass][RADIX_MAX] */) {
    __shared__ uint32_t h[OS_MAX_PASSES * RADIX_MAX];
    for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += SORT_BLOCK) h[x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * SORT_BLOCK * 4;
    for (uint64_t base = (uint64_t)blockIdx.x * SORT_BLOCK * 4; base < n; base += stride) {
        K k[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t i = base + (uint64_t)j * SORT_BLOCK + threadIdx.x;
            k[j] = keys[i < n ? i : n - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (base + (uint64_t)j * SORT_BLOCK + threadIdx.x >= n) continue;
            for (uint32_t p = 0; p < plan.n_passes; p++)
                atomicAdd(&h[p * RADIX_MAX + ((uint32_t)(k[j] >> plan.shift[p]) & plan.mask[p])], 1u);
        }
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += SORT_BLOCK)
        if (h[x]) atomicAdd(&ghist[x], h[x]);
}
// one workgroup per pass: exclusive prefix over the digits, in place
__global__ __launch_bounds__(RADIX_MAX) void k_scan_global_hist(uint32_t *__restrict__ ghist) {
    __shared__ uint32_t lds[RADIX_MAX / 64];
    uint32_t *row = ghist + (uint64_t)blockIdx.x * RADIX_MAX;
    const uint32_t v = row[threadIdx.x];
    row[threadIdx.x] = block_excl_scan<RADIX_MAX>(v, lds, nullptr);
}

__global__ void k_xcc_probe(uint32_t *out) {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 0xFu;
}
// number of XCDs whose ids are 0..n-1 and all receive workgroups; 1 = do not split the tickets
static uint32_t probe_xccs(crgpu_ctx *ctx) {
    const uint32_t nb = 1024;
    uint32_t *d = nullptr;
    if (cr_pool_alloc(ctx, (void **)&d, nb * sizeof(uint32_t)) != CRGPU_OK) return 1;
    hipLaunchKernelGGL(k_xcc_probe, dim3(nb), dim3(64), 0, ctx->stream, d);
    std::vector<uint32_t> h(nb);
    const int rc = crgpu_memcpy_d2h(ctx, h.data(), d, nb * sizeof(uint32_t));
    cr_pool_free(ctx, d);
    if (rc != CRGPU_OK) return 1;
    uint32_t seen = 0, mx = 0;
    for (uint32_t v : h) {
        seen |= 1u << v;
        mx = v > mx ? v : mx;
    }
    const uint32_t n = mx + 1;
    if (n > 16 || seen != (n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u)) return 1;
    if (const char *e = getenv("CRGPU_SORT_XCC"))  // "0": one global ticket sequence (A/B)
        if (atoi(e) == 0) return 1;
    return n;
}

static bool onesweep_enabled() {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("CRGPU_SORT");
        on = (e && strcmp(e, "classic") == 0) ? 0 : 1;  // CRGPU_SORT=classic: histogram pass per radix pass (A/B, fallback)
    }
    return on == 1;
}

// keys only, 64-bit: global histograms once, then one look-back scatter per pass
// d_prehist (nullable): the digit histograms k_build_keys already counted for exactly these keys
// d_vals / d_vals_tmp (HAS_VALS): a 32-bit payload travels with every key
template <bool HAS_VALS>
static int onesweep_sort_u64(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                             uint64_t n, const SweepPlan &plan, const uint32_t *widths, bool *result_in_tmp,
                             const uint32_t *d_prehist) {
    typedef SortCfg<uint64_t, HAS_VALS> Cfg;
    const uint64_t n_chunks = (n + Cfg::CHUNK - 1) / Cfg::CHUNK;
    void *d_small = nullptr, *d_status = nullptr;
    if (!ctx->n_xcc) ctx->n_xcc = probe_xccs(ctx);
    const uint32_t n_xcc = ctx->n_xcc;
    // histograms + per pass 16 ticket counters, a cache line each + the abort word (a line of its own)
    const size_t small_bytes = (size_t)OS_MAX_PASSES * RADIX_MAX * 4 + OS_MAX_PASSES * 16 * 128 + 128;
    CR_TRY(cr_pool_alloc(ctx, &d_small, small_bytes));
    int rc = cr_pool_alloc(ctx, &d_status, n_chunks * RADIX_MAX * sizeof(unsigned long long));
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d_small);
        return rc;
    }
    uint32_t *ghist = (uint32_t *)d_small;
    uint32_t *tickets = ghist + OS_MAX_PASSES * RADIX_MAX;
    uint32_t *d_abort = tickets + OS_MAX_PASSES * 16 * 32;
    // test switch: CRGPU_SORT_FORCE_ABORT=<pass> stalls the look-back chain of that pass (chunk 0 never publishes)
    int force_pass = -1;
    if (const char *fa = getenv("CRGPU_SORT_FORCE_ABORT")) force_pass = atoi(fa);
    hipError_t e = hipMemsetAsync(d_small, 0, small_bytes, ctx->stream);
    {
        CrTimer t(ctx, CRGPU_T_SORT_HIST, n);
        if (d_prehist) {
            if (e == hipSuccess)
                e = hipMemcpyAsync(ghist, d_prehist, (size_t)OS_MAX_PASSES * RADIX_MAX * 4, hipMemcpyDeviceToDevice, ctx->stream);
        } else {
            hipLaunchKernelGGL(k_global_hist<uint64_t>, dim3(256), dim3(SORT_BLOCK), 0, ctx->stream, d_keys, n, plan, ghist);
        }
        hipLaunchKernelGGL(k_scan_global_hist, dim3(plan.n_passes), dim3(RADIX_MAX), 0, ctx->stream, ghist);
    }
    uint64_t *in = d_keys, *out = d_tmp;
    uint32_t *vin = d_vals, *vout = d_vals_tmp;
    const bool st64_forced = getenv("CRGPU_SORT_STATUS64") != nullptr;  // A/B and test switch, read per call like the others
    const bool st32 = n < (1ull << 30) && !st64_forced;
    for (uint32_t p = 0; p < plan.n_passes && e == hipSuccess; p++) {
        const bool wide = widths[p] == 9;
        const size_t lds = Cfg::lds_bytes(wide ? 9 : 8);
        const uint32_t radix = wide ? 512u : 256u;
        RadixDigit dig{plan.shift[p], plan.mask[p]};
        const uint32_t tag = (p + 1u) | ((int)p == force_pass && n_chunks > 1 ? 0x80000000u : 0u);
        CrTimer t(ctx, CRGPU_T_SORT, n);
        e = hipMemsetAsync(d_status, 0, n_chunks * radix * (st32 ? sizeof(uint32_t) : sizeof(unsigned long long)), ctx->stream);
        // one workgroup fits per CU, the rest queue up for tickets.  With per-XCD tickets every XCD must receive
        // workgroups whatever the dispatcher's rotation: always the full grid (idle workgroups leave after one atomic)
        const dim3 grid((unsigned)(n_xcc > 1 || n_chunks > 512 ? 512 : n_chunks));
#define OS_LAUNCH(BITS_, ST32_)                                                                                                     \
    do {                                                                                                                            \
        cr_allow_lds(ctx, (const void *)k_radix_scatter<uint64_t, HAS_VALS, RadixDigit, BITS_, true, ST32_>, lds);                  \
        hipLaunchKernelGGL((k_radix_scatter<uint64_t, HAS_VALS, RadixDigit, BITS_, true, ST32_>), grid, dim3(SORT_BLOCK), lds,     \
                           ctx->stream, in, out, vin, vout, n, 0, dig, nullptr, ghist + p * RADIX_MAX, n_xcc,                       \
                           (unsigned long long *)d_status, tickets + p * 16 * 32, d_abort, tag);                                    \
    } while (0)
        if (wide && st32) OS_LAUNCH(9, true);
        else if (wide) OS_LAUNCH(9, false);
        else if (st32) OS_LAUNCH(8, true);
        else OS_LAUNCH(8, false);
#undef OS_LAUNCH
        if (e == hipSuccess) e = hipGetLastError();
        uint64_t *t2 = in;
        in = out;
        out = t2;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
        *result_in_tmp = !*result_in_tmp;
    }
    uint32_t aborted = 0;  // 0, or 1 + the pass whose watchdog fired: that pass and all later ones wrote nothing usable
    if (e == hipSuccess && crgpu_memcpy_d2h(ctx, &aborted, d_abort, sizeof(aborted)) != CRGPU_OK) e = hipErrorUnknown;
    cr_pool_free(ctx, d_status);
    cr_pool_free(ctx, d_small);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "onesweep sort: %s", hipGetErrorString(e));
    if (aborted) {
        // the input of the failed pass is intact (later passes left at once): finish with the classic three-kernel passes
        const uint32_t p0 = aborted - 1u;
        static const bool verbose = getenv("CRGPU_SORT_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "[crgpu sort] look-back watchdog fired in pass %u of %u: finishing with the classic passes\n", p0, plan.n_passes);
        in = (p0 & 1u) ? d_tmp : d_keys;
        out = (p0 & 1u) ? d_keys : d_tmp;
        vin = (p0 & 1u) ? d_vals_tmp : d_vals;
        vout = (p0 & 1u) ? d_vals : d_vals_tmp;
        for (uint32_t p = p0; p < plan.n_passes; p++) {
            RadixDigit dig{plan.shift[p], plan.mask[p]};
            if (widths[p] == 9)
                CR_TRY((radix_pass<uint64_t, RadixDigit, 9>(ctx, in, out, HAS_VALS ? vin : nullptr, vout, n, dig)));
            else
                CR_TRY((radix_pass<uint64_t, RadixDigit, 8>(ctx, in, out, HAS_VALS ? vin : nullptr, vout, n, dig)));
            uint64_t *t2 = in;
            in = out;
            out = t2;
            uint32_t *tv = vin;
            vin = vout;
            vout = tv;
        }
        ctx->sort_fallbacks++;
    }
    return CRGPU_OK;
}

// Bits of the molecule keys that the radix passes leave to k_finish_runs: with L low bits left out, the passes sort on
// the top bits only (all 9 bits wide) and the finishing pass orders the short runs of equal top bits in LDS.  L is the
// largest value <= 16 that makes ceil((total - L) / 9) passes cover the rest: 61 bits -> 5 passes + 16 bits (instead
// of 7 passes), 64 -> 6 + 10, 50 -> 4 + 14.  0: keys too short to gain a pass, or not switched on.
// OFF by default (CRGPU_SORT_FINISH=1 turns it on; the whole GPU suite passes with it).  At 1 B records the five passes
// take 18.7 ms instead of 25.7, but the finishing costs more than the 7 ms it saves, as a kernel of its own (20 ms) and
// fused with the run-length pass (k_finish_emit, 23 ms against the 3.6 ms of the two compaction launches it replaces:
// 7.5 ms without any ordering work -- 389 K tiles of 2048 keys with a ticket, ten barriers and a look-back each -- 1.2 ms
// for the run bounds and 14.4 ms for the ranks).  The ranks are only 4.8 G compare steps, but 13 % of the keys sit in
// runs of 8 to 64 (the top genes of every cell: Zipf x log-normal cell sizes), their tiles take ten times as long as the
// others, and under ticket order every later tile waits in its look-back for the slow tile's count with all workgroup
// slots taken.  profiles/r02_finish_pass_ab.txt.
// How many low key bits the radix passes leave alone (0: none; the default).
//   CRGPU_SORT_FINISH=2     the fewest low bits (at most 8) whose omission saves a whole pass, provided at least 16 UMI bits
//                           stay above the cut: keys that agree on everything above it are then rare and few (the reads of one
//                           UMI whose UmiType bits differ; UMIs of one (barcode, feature) that share their leading bases), and
//                           k_order_runs puts those short runs in order -- 61 bits: 6 passes + 7 bits, 64 bits: 7 passes + 1 bit.
//                           Measured at 1 B records: the passes 25.9 -> 22.8 ms, k_order_runs 3.7 ms (8.1 with one lane walking
//                           every run through memory, 5.5 with in-register odd-even sorting of the runs inside a wave, 3.7 with
//                           waves 48 keys apart so that short runs never leave their wave): +0.6 ms, so it is not the default;
//                           fused into the run-length count (which reads the keys anyway) it would be -1.5 ms
//   CRGPU_SORT_FINISH=1     the earlier experiment: up to 16 low bits, finished by k_finish_runs / k_finish_emit (slower still)
uint32_t cr_sort_low_bits(uint32_t total_bits, uint32_t umi_bits) {
    const char *e = getenv("CRGPU_SORT_FINISH");
    if (!onesweep_enabled() || total_bits <= 27) return 0;
    if (e && atoi(e) == 0) return 0;   // round 3: leaving the few low bits that save a pass is the default (cr_repair_runs)
    const uint32_t p8 = (total_bits + 7) / 8, p9 = (total_bits + 8) / 9;
    const uint32_t full = p9 < p8 ? p9 : p8;
    if (e && atoi(e) == 1) {
        if (const char *lo = getenv("CRGPU_SORT_FINISH_LOW")) {  // experiment: exactly this many low bits
            const uint32_t low = (uint32_t)atoi(lo);
            return low >= 1 && low <= 16 && low < total_bits ? low : 0;
        }
        const uint32_t p = (total_bits - 16 + 8) / 9;      // passes of 9 bits for the top part
        if (9 * p >= total_bits) return 0;
        const uint32_t low = total_bits - 9 * p;          // <= 16 by construction
        return p < full ? low : 0;
    }
    // at most OR_MAX_LOW_BITS low bits, and at least 14 UMI bits above the cut (sixteen low bits for the 61-bit layout -- five passes --
    // make the runs whole groups of UMIs and the finishing step 24 ms: measured): runs of up to 12 keys are ordered in registers
    // (k_repair_runs), up to 64 by a wave (k_repair_medium_runs), longer ones by a workgroup in LDS (k_repair_long_runs); what does
    // not fit the lists or LDS is walked through memory by one lane (two levels of at most 256 buckets beyond eight bits)
    static_assert(OR_MAX_LOW_BITS >= 8 && OR_MAX_LOW_BITS <= 16, "or_run_through_memory: two levels of at most 256 buckets");
    uint32_t best_low = 0, best_passes = full;
    uint32_t umi_above = 14u;  // UMI bits that stay above the cut (CRGPU_SORT_UMI_ABOVE: A/B)
    if (const char *ua = getenv("CRGPU_SORT_UMI_ABOVE")) umi_above = (uint32_t)atoi(ua);
    for (uint32_t low = 1; low <= OR_MAX_LOW_BITS && low < total_bits; low++) {
        const uint32_t top = total_bits - low;
        const uint32_t q8 = (top + 7) / 8, q9 = (top + 8) / 9;
        const uint32_t q = q9 < q8 ? q9 : q8;
        // bit 0 is the UmiType bit, the UMI sits right above it: low - 1 of its bits fall below the cut
        if (q < best_passes && umi_bits >= umi_above + (low - 1u)) {  // the fewest passes; among equals the fewest low bits
            best_passes = q;
            best_low = low;
        }
    }
    return best_low;
}
bool cr_sort_finish_experiment() {
    const char *e = getenv("CRGPU_SORT_FINISH");
    return e && atoi(e) == 1;
}

bool cr_sweep_plan(uint32_t lo_bit, uint32_t hi_bit, SweepPlan *plan, uint32_t *widths) {
    if (hi_bit <= lo_bit || !onesweep_enabled()) return false;
    const uint32_t total = hi_bit - lo_bit;
    const uint32_t p8 = (total + 7) / 8, p9 = (total + 8) / 9;
    uint32_t n9 = 0;
    if (p9 < p8 && total > 8 * p9) n9 = total - 8 * p9;
    if (const char *e = getenv("CRGPU_SORT_DIGITS"))
        if (atoi(e) == 8) n9 = 0;
    const uint32_t passes = n9 ? p9 : p8;
    if (passes > OS_MAX_PASSES) return false;
    memset(plan, 0, sizeof(*plan));
    for (uint32_t pass = 0, sh = lo_bit; pass < passes && sh < hi_bit; pass++) {
        const uint32_t width = pass >= passes - n9 ? 9u : 8u;
        const uint32_t bits = hi_bit - sh < width ? hi_bit - sh : width;
        plan->shift[plan->n_passes] = sh;
        plan->mask[plan->n_passes] = (1u << bits) - 1u;
        widths[plan->n_passes++] = width;
        sh += width;
    }
    return true;
}

// ---- finishing pass: order the runs of equal top bits by their low bits ---------------------------------------------
// After the radix passes on bits [low, total) the keys are sorted by their top bits and stable otherwise.  Runs of equal
// top bits are short (keys of one (barcode, feature, library) whose UMIs share their leading bases, mostly copies of one
// key), so every key finds its place by counting the keys of its run that precede it (by low bits, ties by position:
// exactly the order the full stable sort produces).  A workgroup takes the whole runs whose head lies in its tile of
// FIN_TILE keys, stages tile + halo in LDS with one coalesced load (no dependent global loads) and writes back only the
// keys that move.  A run longer than FIN_HALO raises *fallback and its tile writes nothing: the host then sorts the
// buffer, still a permutation of the keys, again on all bits.
#define FIN_TILE 4096u
#define FIN_HALO 1024u     // keys staged behind the tile: the runs that begin in the tile end in there, or the sort is redone
#define FIN_CAP (1u + FIN_TILE + FIN_HALO)
template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_finish_runs(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n,
                                                     uint32_t low_bits, uint32_t *__restrict__ fallback) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);                                       // FIN_CAP keys
    uint16_t *spos = reinterpret_cast<uint16_t *>(smem + (size_t)FIN_CAP * 8);              // where each key belongs
    uint32_t *sv = reinterpret_cast<uint32_t *>(smem + (size_t)FIN_CAP * 8 + ((FIN_CAP * 2 + 15) & ~15u));  // payloads
    __shared__ uint32_t s_a, s_b, s_bad;
    const uint32_t tid = threadIdx.x;
    const uint64_t n_tiles = (n + FIN_TILE - 1) / FIN_TILE;
    const uint64_t lowmask = (1ull << low_bits) - 1ull;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * FIN_TILE, t1 = t0 + FIN_TILE < n ? t0 + FIN_TILE : n;
        // one coalesced load of the window: the key before the tile, the tile, the halo
        const uint64_t w0 = t0 ? t0 - 1 : 0, w1 = t1 + FIN_HALO < n ? t1 + FIN_HALO : n;
        const uint32_t cnt = (uint32_t)(w1 - w0), off0 = (uint32_t)(t0 - w0), off1 = (uint32_t)(t1 - w0);
        if (tid == 0) {
            s_a = t0 == 0 ? 0u : 0xFFFFFFFFu;
            s_b = 0xFFFFFFFFu;
            s_bad = 0u;
        }
        for (uint32_t i = tid; i < cnt; i += 256) {
            sk[i] = keys[w0 + i];
            if (HAS_VALS) sv[i] = vals[w0 + i];
        }
        __syncthreads();
        // [a, b): the whole runs whose head lies in the tile = first run head at or after t0 .. first one at or after t1
        // (every thread keeps its first hit, a wave reduces, four LDS atomics per tile: one atomic per head cost 25 ms)
        {
            uint32_t fa = 0xFFFFFFFFu, fb = 0xFFFFFFFFu;
            for (uint32_t i = tid + 1; i < cnt && fb == 0xFFFFFFFFu; i += 256) {
                if ((sk[i] >> low_bits) == (sk[i - 1] >> low_bits)) continue;
                if (i >= off0 && fa == 0xFFFFFFFFu) fa = i;
                if (i >= off1) fb = i;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t xa = __shfl_xor(fa, d), xb = __shfl_xor(fb, d);
                fa = xa < fa ? xa : fa;
                fb = xb < fb ? xb : fb;
            }
            if ((tid & 63u) == 0u) {
                if (fa != 0xFFFFFFFFu) atomicMin(&s_a, fa);
                if (fb != 0xFFFFFFFFu) atomicMin(&s_b, fb);
            }
        }
        __syncthreads();
        const uint32_t ra = s_a;
        uint32_t rb = s_b;
        if (rb == 0xFFFFFFFFu && w1 == n) rb = cnt;  // the last run ends with the array
        __syncthreads();
        if (ra == 0xFFFFFFFFu || ra >= off1) {
            // no run begins in this tile: it lies inside a run whose owner deals with it (or raises the fallback)
            continue;
        }
        if (rb == 0xFFFFFFFFu) {  // the last run of the tile goes on beyond the halo
            if (tid == 0) atomicOr(fallback, 1u);
            continue;
        }
        for (uint32_t i = ra + tid; i < rb; i += 256) {
            const uint64_t k = sk[i];
            const uint64_t top = k >> low_bits, low = k & lowmask;
            uint32_t s = i, e = i + 1, rank = 0;
            while (s > ra && (sk[s - 1] >> low_bits) == top) s--;
            while (e < rb && (sk[e] >> low_bits) == top) e++;
            if (e - s > FIN_HALO) s_bad = 1u;  // quadratic work beyond this: leave it to the full sort
            else {
                for (uint32_t j = s; j < i; j++) rank += (sk[j] & lowmask) <= low;   // earlier position: ties stay in front
                for (uint32_t j = i + 1; j < e; j++) rank += (sk[j] & lowmask) < low;
            }
            spos[i] = (uint16_t)(s + rank);
        }
        __syncthreads();
        const bool bad = s_bad != 0u;
        if (bad) {
            if (tid == 0) atomicOr(fallback, 1u);  // nothing of this tile is written: the buffer stays a permutation
        } else {
            for (uint32_t i = ra + tid; i < rb; i += 256) {
                const uint32_t p = spos[i];
                if (p != i) {
                    keys[w0 + p] = sk[i];
                    if (HAS_VALS) vals[w0 + p] = sv[i];
                }
            }
        }
        __syncthreads();
    }
}

// ---- ordering the short runs that a sort on the top bits leaves ------------------------------------------------------------
// After passes on the bits above `low`, keys that agree on those bits sit next to each other in arrival order.  With the
// cut of cr_sort_low_bits such runs are mostly single keys or copies of ONE key (the reads of a molecule); what needs work
// are the reads of one UMI whose UmiType bits differ and the rare UMIs of a (barcode, feature) that share their leading bases.
// One streaming pass: the lane that holds the first key of a run (neighbours from the lanes next to it, memory only at the
// wave's edges) owns the run: two keys are compared and swapped in registers, longer runs (rare) are insertion-sorted in
// place through memory, long disordered ones by an in-place bucket permutation on the low bits; a run of more than OR_MAX
// keys raises *bad and the caller sorts on all bits instead.  Rewriting a run never changes the bits above the cut, which is
// all a neighbouring lane looks at.
#define OR_STEP 48u     // keys between the waves of k_order_runs (64 keys each)
#define OR_CAP 32u      // runs up to this length: insertion sort
#define OR_MAX 65536u   // longer runs than this are not scanned by one lane: the caller sorts on all bits
// one lane puts the run [i, e) in order through memory: e found by scanning, nothing to do when it is already ordered
// in-place bucket permutation (American flag sort) of keys[a, e) on the `bits` (<= 8) bits above `shift`; cnt / nxt hold 2^bits
// entries; on return cnt[d] = end of bucket d (relative to a)
template <bool HAS_VALS>
__device__ void or_flag_permute(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t a, uint64_t e, uint32_t shift,
                                uint32_t bits, uint32_t *cnt, uint32_t *nxt) {
    const uint32_t nb = 1u << bits, dm = nb - 1u;
    for (uint32_t d = 0; d < nb; d++) cnt[d] = 0;
    for (uint64_t x = a; x < e; x++) cnt[(uint32_t)(keys[x] >> shift) & dm]++;
    uint32_t acc = 0;
    for (uint32_t d = 0; d < nb; d++) {
        nxt[d] = acc;
        acc += cnt[d];
        cnt[d] = acc;  // end of bucket d
    }
    for (uint32_t d = 0; d < nb; d++) {
        while (nxt[d] < cnt[d]) {
            const uint64_t k = keys[a + nxt[d]];
            const uint32_t kd = (uint32_t)(k >> shift) & dm;
            if (kd == d) {
                nxt[d]++;
            } else {
                const uint64_t t = a + nxt[kd];
                const uint64_t o = keys[t];
                keys[t] = k;
                keys[a + nxt[d]] = o;
                if (HAS_VALS) {
                    const uint32_t va = vals[a + nxt[d]], vb = vals[t];
                    vals[t] = va;
                    vals[a + nxt[d]] = vb;
                }
                nxt[kd]++;
            }
        }
    }
}

template <bool HAS_VALS>
__device__ void or_run_through_memory(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n, uint32_t low, uint64_t i,
                                      uint32_t *__restrict__ bad) {
    const uint64_t first = keys[i], top = first >> low;
    uint64_t e = i + 1, prev = first;
    bool disorder = false;
    while (e < n && e - i <= OR_MAX) {
        const uint64_t k = keys[e];
        if ((k >> low) != top) break;
        disorder |= k < prev;
        prev = k;
        e++;
    }
    if (e - i > OR_MAX) {
        atomicOr(bad, 1u);
        return;
    }
    if (!disorder) return;
    const uint64_t len = e - i;
    if (len <= OR_CAP) {  // insertion sort in place (only this lane touches the run)
        for (uint64_t a = i + 1; a < e; a++) {
            const uint64_t k = keys[a];
            const uint32_t v = HAS_VALS ? vals[a] : 0u;
            uint64_t b = a;
            while (b > i && keys[b - 1] > k) {
                keys[b] = keys[b - 1];
                if (HAS_VALS) vals[b] = vals[b - 1];
                b--;
            }
            if (b != a) {
                keys[b] = k;
                if (HAS_VALS) vals[b] = v;
            }
        }
        return;
    }
    // a long run that mixes low bits (many reads of one UMI with both UmiTypes, ...): in-place bucket permutation on the
    // low bits (American flag sort).  Up to 8 low bits: one level of at most 256 buckets; 9 .. OR_MAX_LOW_BITS: first on the
    // bits above the low eight (at most 8 buckets), then every such bucket on its low eight.  (The bucket arrays live in
    // scratch, per lane: 256 entries each is what a kernel with 256-thread workgroups can afford.)
    uint32_t cnt[256], nxt[256];
    if (low > OR_MAX_LOW_BITS) {  // cannot happen (cr_sort_low_bits): never overrun the arrays, let the caller sort on all bits
        atomicOr(bad, 1u);
        return;
    }
    if (low <= 8u) {
        or_flag_permute<HAS_VALS>(keys, vals, i, e, 0u, low, cnt, nxt);
        return;
    }
    uint32_t ends[1u << (OR_MAX_LOW_BITS - 8u)];
    const uint32_t hi_bits = low - 8u;
    or_flag_permute<HAS_VALS>(keys, vals, i, e, 8u, hi_bits, cnt, nxt);
    for (uint32_t d = 0; d < (1u << hi_bits); d++) ends[d] = cnt[d];
    uint32_t from = 0;
    for (uint32_t d = 0; d < (1u << hi_bits); d++) {
        if (ends[d] - from > 1u) or_flag_permute<HAS_VALS>(keys, vals, i + from, i + ends[d], 0u, 8u, cnt, nxt);
        from = ends[d];
    }
}

template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_order_runs(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n, uint32_t low,
                                                    uint32_t *__restrict__ bad) {
    // A wave looks at 64 consecutive keys but the waves are only OR_STEP keys apart: a wave owns the runs that START in its
    // first OR_STEP lanes, so a run of up to 64 - OR_STEP + 1 keys always ends inside the wave that owns it (a run that
    // leaves its wave costs one lane a serial walk through memory while the other 63 wait: with waves 64 apart that was
    // half of all waves and 4 of the pass's 5.5 ms).
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_waves = (n + OR_STEP - 1) / OR_STEP;
    const uint64_t wave_stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); gw < n_waves; gw += wave_stride) {
        const uint64_t i = gw * OR_STEP + lane;
        const bool in = i < n;
        const uint64_t ic = in ? i : n - 1;
        uint64_t key = keys[ic];
        uint32_t val = HAS_VALS ? vals[ic] : 0u;
        const uint64_t key0 = key;
        const uint32_t val0 = val;
        uint64_t left = __shfl_up(key, 1);
        if (lane == 0) left = ic > 0 ? keys[ic - 1] : ~key;
        uint64_t behind = ~key;  // the key behind the wave (lane 63 only)
        if (lane == 63 && ic + 1 < n) behind = keys[ic + 1];
        const uint64_t top = key >> low;
        const bool start = in && (i == 0 || (left >> low) != top);
        const unsigned long long starts = __ballot(start);
        const unsigned long long upto = ~0ull >> (63u - lane);  // lanes 0 .. lane
        const bool has_head = (starts & upto) != 0ull;          // else: part of a run that began before the wave
        const uint32_t rs = has_head ? 63u - (uint32_t)__clzll((long long)(starts & upto)) : 0u;
        const unsigned long long above = lane < 63u ? starts >> (lane + 1u) : 0ull;
        const uint32_t n_in = (uint32_t)__popcll(__ballot(in));  // lanes behind the last key hold a copy of it: not part of any run
        uint32_t re = above ? lane + 1u + (uint32_t)__ffsll((long long)above) - 1u : 64u;  // exclusive
        const bool to_wave_end = re == 64u;
        re = re < n_in ? re : n_in;
        // the wave's last run goes on behind the wave?  (also: keys behind n do not exist)
        const bool last_goes_on = __shfl((behind >> low) == top && in, 63);
        const bool owned = in && has_head && rs < OR_STEP;  // runs that start in the last lanes belong to the next wave
        const bool closed = owned && !(to_wave_end && last_goes_on);
        // a run that starts here and leaves the wave: its first lane does it through memory
        if (start && owned && to_wave_end && last_goes_on) or_run_through_memory<HAS_VALS>(keys, vals, n, low, i, bad);
        // closed runs with a descent somewhere: odd-even transposition across the lanes, in registers
        const unsigned long long desc = __ballot(closed && lane > rs && key < left);
        const unsigned long long mine = (re >= 64u ? ~0ull : ((1ull << re) - 1ull)) & ~((1ull << rs) - 1ull);  // lanes rs .. re - 1
        const bool need = closed && (desc & mine) != 0ull;
        uint32_t len = need ? re - rs : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(len, d);
            len = o > len ? o : len;
        }
        for (uint32_t t = 0; t < len; t++) {  // uniform: the longest run of the wave that needs work
            const bool low_side = ((lane - rs) & 1u) == (t & 1u);
            const uint32_t partner = low_side ? lane + 1u : lane - 1u;
            const bool ok = need && (low_side ? partner < re : lane > rs);
            const uint64_t pk = __shfl(key, (int)(partner & 63u));
            const uint32_t pv = HAS_VALS ? __shfl(val, (int)(partner & 63u)) : 0u;
            if (ok) {
                const bool take = low_side ? pk < key : pk > key;  // the lower lane keeps the smaller key
                if (take) {
                    key = pk;
                    val = pv;
                }
            }
        }
        if (need && (key != key0 || (HAS_VALS && val != val0))) {  // an equal key may have arrived with another read's ordinal
            keys[i] = key;
            if (HAS_VALS) vals[i] = val;
        }
    }
}

static int cr_order_runs_r02(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back) {
    *fell_back = false;
    if (n < 2 || low_bits == 0) return CRGPU_OK;
    uint32_t *d_flag = ctx->d_scalars + 52;
    {
        CrTimer t(ctx, CRGPU_T_SORT_HIST, n);  // booked beside the histogram slot: "sort, not a scatter pass"
        CR_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ctx->stream));
        const dim3 grid(cr_grid((n + OR_STEP - 1) / OR_STEP * 64u, 256, 256u * 16u));
        if (d_vals)
            hipLaunchKernelGGL(k_order_runs<true>, grid, dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits, d_flag);
        else
            hipLaunchKernelGGL(k_order_runs<false>, grid, dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits, d_flag);
        CR_HIP(ctx, hipGetLastError());
    }
    uint32_t flag = 0;
    CR_TRY(crgpu_memcpy_d2h(ctx, &flag, d_flag, sizeof(flag)));
    *fell_back = flag != 0;
    return CRGPU_OK;
}

// ---- the same in two steps: find the descents, repair only the runs that hold one (the default since round 3) -------------
// k_order_runs touches every key of the buffer one load per lane at a time (3.7 ms per 796 M keys, more than the radix pass
// it saves).  But almost every run of equal top bits is a single key or copies of ONE key: nothing to do.  So:
//   k_find_descents   a streaming read with the wave-blocked layout of the run-length passes (several loads per lane in
//                     flight, the left neighbour from the lane below): bit i of desc[] = key i continues the run of its left
//                     neighbour (same bits above `low`) and is SMALLER than it -- the run is out of order there.  One 64-bit
//                     ballot word per 64 keys is written: 1.6 % of the bytes read.
//   k_repair_runs     one lane per set bit; the lane whose bit is the FIRST descent of its run (no other one between the
//                     run's head and it) owns the run: up to RR_SHORT keys are put in order in its registers; a longer run's
//                     head goes to a device-wide list.
//   k_repair_medium_runs / k_repair_long_runs   the listed runs: up to 64 keys by a wave, more by a workgroup in LDS, what
//                     fits neither through memory by one lane (or_run_through_memory; *bad for runs beyond OR_MAX -> the
//                     caller sorts on all bits).  The lists spread the runs of a few hot barcodes over the whole device.
// Runs are disjoint, so the repairing lanes never touch each other's keys.
#define FD_ITEMS 8
__global__ __launch_bounds__(256) void k_find_descents(const uint64_t *__restrict__ keys, uint64_t n, uint32_t low,
                                                       unsigned long long *__restrict__ desc, uint32_t *__restrict__ n_words_set) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t span = 64ull * FD_ITEMS;
    const uint64_t n_spans = (n + span - 1) / span;
    const uint64_t wave_stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t set = 0;
    for (uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); gw < n_spans; gw += wave_stride) {
        const uint64_t w0 = gw * span;
        uint64_t key[FD_ITEMS];
        uint64_t carry = keys[w0 > 0 ? w0 - 1 : 0];
        bool carry_valid = w0 > 0;
#pragma unroll
        for (int j = 0; j < FD_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            key[j] = keys[i < n ? i : n - 1];
        }
#pragma unroll
        for (int j = 0; j < FD_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            uint64_t left = __shfl_up(key[j], 1);
            if (lane == 0) left = carry;
            const bool has_left = lane > 0 || carry_valid;
            const bool d = i < n && has_left && (key[j] >> low) == (left >> low) && key[j] < left;
            const unsigned long long m = __ballot(d);
            if (lane == 0 && w0 + (uint64_t)j * 64 < n) {
                desc[(w0 >> 6) + j] = m;
                set += m != 0ull;
            }
            carry = __shfl(key[j], 63);
            carry_valid = true;
        }
    }
    if (lane == 0 && set) atomicAdd(n_words_set, set);
}

#define RR_SHORT 12u  // runs up to this many keys are ordered in registers
// one descent: find the head of its run and, when it is the run's first descent, put the run in order
// later_list (nullable) / later_n / later_cap: a run longer than RR_SHORT is not walked through memory by this lane but noted
// (its head) in a device-wide list for k_repair_medium_runs
template <bool HAS_VALS>
__device__ __forceinline__ void repair_at(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n, uint32_t low,
                                          const unsigned long long *__restrict__ desc, uint64_t p, uint32_t *__restrict__ bad,
                                          unsigned long long *__restrict__ later_list = nullptr,
                                          uint32_t *__restrict__ later_n = nullptr, uint32_t later_cap = 0u) {
    // keys[p] < keys[p - 1], same top bits.  Walk back to the head of the run.  An earlier descent on the way means another
    // lane owns the run.  Only things that no repair changes are looked at: the descent bits (read-only here) and the TOP
    // bits of the keys (a repair permutes keys inside one run, whose top bits are all equal) -- the owner may already be
    // rewriting this run.
    const uint64_t top = keys[p] >> low;
    uint64_t h = p;
    for (;;) {
        h--;  // keys[h] belongs to the run (p - 1 does by construction)
        if ((desc[h >> 6] >> (h & 63u)) & 1ull) return;
        if (h == 0 || (keys[h - 1] >> low) != top) break;  // h is the head
        if (p - h > OR_MAX) {
            atomicOr(bad, 1u);
            return;
        }
    }
    // Short runs (nearly all of them: the reads of one molecule with a sequencing error in the last bases of the UMI) are
    // taken into registers with RR_SHORT independent loads, ordered there by a stable odd-even transposition and written back
    // where they changed -- one memory latency instead of the two dozen dependent accesses of the walk through memory.
    uint64_t k[RR_SHORT];
    uint32_t v[RR_SHORT];
#pragma unroll
    for (uint32_t j = 0; j < RR_SHORT; j++) {
        const uint64_t at = h + j < n ? h + j : n - 1;
        k[j] = keys[at];
        v[j] = HAS_VALS ? vals[at] : 0u;
    }
    uint32_t len = RR_SHORT + 1u;  // RR_SHORT + 1: the run goes on behind the loaded keys
#pragma unroll
    for (uint32_t j = RR_SHORT; j-- > 0;)
        if (h + j >= n || (k[j] >> low) != top) len = j;
    if (len > RR_SHORT) {
        // is the key right behind the window still part of the run?
        if (h + RR_SHORT < n && (keys[h + RR_SHORT] >> low) == top) {
            if (later_list) {
                const uint32_t slot = atomicAdd(later_n, 1u);
                if (slot < later_cap) {
                    later_list[slot] = h;
                    return;
                }
            }
            or_run_through_memory<HAS_VALS>(keys, vals, n, low, h, bad);
            return;
        }
        len = RR_SHORT;
    }
    uint64_t k0[RR_SHORT];
    uint32_t v0[RR_SHORT];
#pragma unroll
    for (uint32_t j = 0; j < RR_SHORT; j++) {
        k0[j] = k[j];
        v0[j] = v[j];
    }
    // stable: equal keys are never exchanged (adjacent exchanges of strictly descending pairs only)
#pragma unroll
    for (uint32_t t = 0; t < RR_SHORT; t++) {
#pragma unroll
        for (uint32_t j = t & 1u; j + 1 < RR_SHORT; j += 2) {
            const bool sw = j + 1 < len && k[j] > k[j + 1];
            const uint64_t a = k[j], b = k[j + 1];
            const uint32_t va = v[j], vb = v[j + 1];
            k[j] = sw ? b : a;
            k[j + 1] = sw ? a : b;
            v[j] = sw ? vb : va;
            v[j + 1] = sw ? va : vb;
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < RR_SHORT; j++)
        if (j < len && (k[j] != k0[j] || (HAS_VALS && v[j] != v0[j]))) {
            keys[h + j] = k[j];
            if (HAS_VALS) vals[h + j] = v[j];
        }
}

// The descents are few (a few per mille of the keys) and scattered: a lane that took them straight from its mask word
// would work while its 63 neighbours wait.  A workgroup therefore collects the descents of RR_WORDS mask words in LDS and
// hands them out one per thread.
#define RR_WORDS 2048u
#define RR_CAP 4096u
template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_repair_runs(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n, uint32_t low,
                                                     const unsigned long long *__restrict__ desc, uint64_t n_words,
                                                     uint32_t *__restrict__ bad, unsigned long long *__restrict__ med_list,
                                                     uint32_t *__restrict__ n_med, uint32_t med_cap) {
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_pos[RR_CAP];  // position inside the tile (RR_WORDS * 64 keys: 17 bits)
    const uint64_t n_tiles = (n_words + RR_WORDS - 1) / RR_WORDS;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = tile * RR_WORDS;
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < RR_WORDS; j += 256) {
            unsigned long long m = w0 + j < n_words ? desc[w0 + j] : 0ull;
            while (m) {
                const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1ull;
                const uint32_t slot = atomicAdd(&s_n, 1u);
                if (slot < RR_CAP) s_pos[slot] = j * 64u + b;
                else repair_at<HAS_VALS>(keys, vals, n, low, desc, (w0 + j) * 64 + b, bad, med_list, n_med, med_cap);  // a tile full of descents: at once
            }
        }
        __syncthreads();
        const uint32_t cnt = s_n < RR_CAP ? s_n : RR_CAP;
        for (uint32_t t = threadIdx.x; t < cnt; t += 256)
            repair_at<HAS_VALS>(keys, vals, n, low, desc, w0 * 64 + s_pos[t], bad, med_list, n_med, med_cap);
        __syncthreads();
        __syncthreads();
    }
}

// Runs of RR_SHORT + 1 .. 64 keys with a descent (the reads of a larger molecule, some with a sequencing error in the low UMI
// bases): one WAVE each -- a lane per key, the key's place = the number of keys of the run that go before it (by value, ties
// by position: the stable order), found with one broadcast per member.  The heads come from a device-wide list, so the
// waves share them evenly: inside k_repair_runs the tiles of a few hot barcodes held thousands of such runs each and
// made the kernel's makespan (8.6 ms with ten low bits), as did one lane walking a run through memory (tens of us).
// A run that goes on behind the wave's 64 keys is passed on to k_repair_long_runs.
template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_repair_medium_runs(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n,
                                                            uint32_t low, const unsigned long long *__restrict__ med_list,
                                                            const uint32_t *__restrict__ n_med, uint32_t med_cap,
                                                            unsigned long long *__restrict__ long_list, uint32_t *__restrict__ n_long,
                                                            uint32_t long_cap, uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cnt = *n_med < med_cap ? *n_med : med_cap;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < cnt; t += n_waves) {  // wave-uniform
        const uint64_t h = med_list[t];
        const uint64_t at = h + lane < n ? h + lane : n - 1;
        const uint64_t key = keys[at];
        const uint32_t val = HAS_VALS ? vals[at] : 0u;
        const uint64_t top = __shfl(key, 0) >> low;
        const unsigned long long in = __ballot(h + lane < n && (key >> low) == top);
        const uint32_t len = ~in ? (uint32_t)__ffsll((long long)~in) - 1u : 64u;  // the leading lanes that agree with the head
        const bool goes_on = len == 64u && h + 64u < n && (keys[h + 64u] >> low) == top;
        if (goes_on) {
            if (lane == 0) {
                const uint32_t slot = long_cap ? atomicAdd(n_long, 1u) : 0xFFFFFFFFu;
                if (slot < long_cap) long_list[slot] = h;
                else or_run_through_memory<HAS_VALS>(keys, vals, n, low, h, bad);
            }
            continue;
        }
        uint32_t rank = 0;
        for (uint32_t i = 0; i < len; i++) {  // uniform
            const uint64_t ki = __shfl(key, (int)i);
            rank += (ki < key || (ki == key && i < lane)) ? 1u : 0u;
        }
        if (lane < len && rank != lane) {  // (every key of the run is in a register by now: the stores cannot hit a key not yet read)
            keys[h + rank] = key;
            if (HAS_VALS) vals[h + rank] = val;
        }
    }
}

// Runs of more than 64 keys with a descent (the reads of a large molecule, a few of them with a sequencing error in the low UMI
// bases): one WORKGROUP each.  The keys of a run differ in their low bits only, so the run is sorted as 32-bit words
// (low bits << 12 | position in the run) by a bitonic network in LDS -- the position makes the order the stable one -- and
// written back from there.  One lane walking such a run through memory took milliseconds (3 us per key) and was the
// kernel's whole makespan once ten low bits were left to the repair step.  Runs beyond RL_CAP keys: through memory.
#define RL_CAP 4096u
template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_repair_long_runs(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n, uint32_t low,
                                                          const unsigned long long *__restrict__ long_list,
                                                          const uint32_t *__restrict__ n_long, uint32_t long_cap,
                                                          uint32_t *__restrict__ bad) {
    __shared__ uint32_t s_c[RL_CAP];
    __shared__ uint32_t s_v[HAS_VALS ? RL_CAP : 1];
    __shared__ uint32_t s_len;
    const uint32_t tid = threadIdx.x;
    const uint32_t cnt = *n_long < long_cap ? *n_long : long_cap;
    const uint64_t lowmask = (1ull << low) - 1ull;
    for (uint32_t it = blockIdx.x; it < cnt; it += gridDim.x) {
        const uint64_t h = long_list[it];
        const uint64_t top = keys[h] >> low;
        if (tid == 0) s_len = RL_CAP + 1u;
        __syncthreads();
        // the run's length: the first position behind h whose top bits differ (RL_CAP + 1: the run goes on behind the window)
        for (uint32_t j = tid; j <= RL_CAP; j += 256) {
            const bool out = h + j >= n || (keys[h + j] >> low) != top;
            if (out) atomicMin(&s_len, j);
        }
        __syncthreads();
        const uint32_t len = s_len;
        if (len > RL_CAP) {  // uniform
            if (tid == 0) or_run_through_memory<HAS_VALS>(keys, vals, n, low, h, bad);
            __syncthreads();
            continue;
        }
        uint32_t P = 128;
        while (P < len) P <<= 1;
        for (uint32_t j = tid; j < P; j += 256) {
            s_c[j] = j < len ? ((uint32_t)(keys[h + j] & lowmask) << 12) | j : 0xFFFFFFFFu;
            if (HAS_VALS && j < len) s_v[j] = vals[h + j];
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t x = tid; x < P / 2; x += 256) {
                    const uint32_t a = 2u * x - (x & (j - 1u));  // lower index of the pair at distance j
                    const uint32_t b = a + j;
                    const bool up = (a & k) == 0u;
                    const uint32_t ca = s_c[a], cb = s_c[b];
                    if ((ca > cb) == up) {
                        s_c[a] = cb;
                        s_c[b] = ca;
                    }
                }
                __syncthreads();
            }
        for (uint32_t j = tid; j < len; j += 256) {
            const uint32_t c = s_c[j];
            if ((c & 4095u) != j) {
                keys[h + j] = (top << low) | (uint64_t)(c >> 12);
                if (HAS_VALS) vals[h + j] = s_v[c & 4095u];
            }
        }
        __syncthreads();
    }
}

int cr_repair_runs(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back) {
    *fell_back = false;
    if (n < 2 || low_bits == 0) return CRGPU_OK;
    uint32_t *d_flag = ctx->d_scalars + 52, *d_set = ctx->d_scalars + 53, *d_nlong = ctx->d_scalars + 54, *d_nmed = ctx->d_scalars + 55;
    const uint64_t n_words = (n + 63) / 64;
    void *d_desc = nullptr, *d_long = nullptr, *d_med = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &d_desc, n_words * sizeof(unsigned long long)));
    struct Rel {
        crgpu_ctx *c;
        void *&p;
        ~Rel() { cr_pool_free(c, p); }
    } rel{ctx, d_desc}, rel_long{ctx, d_long}, rel_med{ctx, d_med};
    // heads of the runs of more than 64 keys that need work (a run has at least 65 keys: at most n / 65 of them; capped)
    uint32_t long_cap = (uint32_t)std::min<uint64_t>(n / 65 + 1, 1u << 20);
    if (cr_pool_alloc(ctx, &d_long, (uint64_t)long_cap * sizeof(unsigned long long)) != CRGPU_OK) {
        d_long = nullptr;  // not fatal: such runs are walked through memory
        long_cap = 0;
    }
    // ... and of the runs of 13 .. 64 keys (at most n / 13; capped: beyond it a lane walks the run through memory)
    uint32_t med_cap = (uint32_t)std::min<uint64_t>(n / 13 + 1, 1u << 24);
    if (cr_pool_alloc(ctx, &d_med, (uint64_t)med_cap * sizeof(unsigned long long)) != CRGPU_OK) {
        d_med = nullptr;
        med_cap = 0;
    }
    {
        CrTimer t(ctx, CRGPU_T_SORT_HIST, n);  // booked beside the histogram slot: "sort, not a scatter pass"
        CR_HIP(ctx, hipMemsetAsync(d_flag, 0, 4 * sizeof(uint32_t), ctx->stream));
        const uint64_t n_spans = (n + 64ull * FD_ITEMS - 1) / (64ull * FD_ITEMS);
        hipLaunchKernelGGL(k_find_descents, dim3(cr_grid(n_spans * 64u, 256, 256u * 8u)), dim3(256), 0, ctx->stream, d_keys, n, low_bits,
                           (unsigned long long *)d_desc, d_set);
        const dim3 grid(cr_grid((n_words + RR_WORDS - 1) / RR_WORDS, 1, 256u * 8u));
#define CR_REPAIR_LAUNCH(HV)                                                                                                          \
    hipLaunchKernelGGL(k_repair_runs<HV>, grid, dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits,                                   \
                       (const unsigned long long *)d_desc, n_words, d_flag, (unsigned long long *)d_med, d_nmed, med_cap);                \
    if (med_cap)                                                                                                                          \
        hipLaunchKernelGGL(k_repair_medium_runs<HV>, dim3(256u * 8u), dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits,             \
                           (const unsigned long long *)d_med, d_nmed, med_cap, (unsigned long long *)d_long, d_nlong, long_cap, d_flag); \
    if (long_cap)                                                                                                                         \
        hipLaunchKernelGGL(k_repair_long_runs<HV>, dim3(256u * 4u), dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits,               \
                           (const unsigned long long *)d_long, d_nlong, long_cap, d_flag)
        if (d_vals) {
            CR_REPAIR_LAUNCH(true);
        } else {
            CR_REPAIR_LAUNCH(false);
        }
#undef CR_REPAIR_LAUNCH
        CR_HIP(ctx, hipGetLastError());
    }
    uint32_t flag = 0;
    CR_TRY(crgpu_memcpy_d2h(ctx, &flag, d_flag, sizeof(flag)));
    *fell_back = flag != 0;
    return CRGPU_OK;
}
// the finishing step behind the passes on the top bits: CRGPU_SORT_FINISH=3 keeps round 2's k_order_runs (A/B)
static int order_low_bits(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back) {
    const char *e = getenv("CRGPU_SORT_FINISH");
    if (e && atoi(e) == 3) return cr_order_runs_r02(ctx, d_keys, d_vals, n, low_bits, fell_back);
    return cr_repair_runs(ctx, d_keys, d_vals, n, low_bits, fell_back);
}
int cr_order_runs(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back) {
    return order_low_bits(ctx, d_keys, d_vals, n, low_bits, fell_back);
}

static int finish_runs(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back) {
    *fell_back = false;
    uint32_t *d_flag = ctx->d_scalars + 52;
    const size_t lds = (size_t)FIN_CAP * 8 + ((FIN_CAP * 2 + 15) & ~15u) + (d_vals ? (size_t)FIN_CAP * 4 : 0);
    {
        CrTimer t(ctx, CRGPU_T_SORT_HIST, n);  // booked beside the histogram slot: "sort, not a scatter pass"
        CR_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ctx->stream));
        const dim3 grid(cr_grid((n + FIN_TILE - 1) / FIN_TILE, 1, 256u * 8u));
        if (d_vals) {
            cr_allow_lds(ctx, (const void *)k_finish_runs<true>, lds);
            hipLaunchKernelGGL(k_finish_runs<true>, grid, dim3(256), lds, ctx->stream, d_keys, d_vals, n, low_bits, d_flag);
        } else {
            cr_allow_lds(ctx, (const void *)k_finish_runs<false>, lds);
            hipLaunchKernelGGL(k_finish_runs<false>, grid, dim3(256), lds, ctx->stream, d_keys, d_vals, n, low_bits, d_flag);
        }
        CR_HIP(ctx, hipGetLastError());
    }
    uint32_t flag = 0;
    CR_TRY(crgpu_memcpy_d2h(ctx, &flag, d_flag, sizeof(flag)));
    *fell_back = flag != 0;
    return CRGPU_OK;
}

// low_left (nullable): the caller finishes the sort itself (cr_finish_emit: finishing pass fused with the run-length
// pass); *low_left = number of low bits the passes did not sort on (0: fully sorted)
template <typename K>
static int radix_sort(crgpu_ctx *ctx, K *d_keys, K *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp, uint64_t n,
                      uint32_t lo_bit, uint32_t hi_bit, bool *result_in_tmp, uint32_t *low_left = nullptr) {
    *result_in_tmp = false;
    if (low_left) *low_left = 0;
    if (n <= 1 || hi_bit <= lo_bit) return CRGPU_OK;
    CR_REQUIRE(ctx, n < 0xFFFFFFFFull, CRGPU_ERANGE, "sort: at most 2^32-2 keys per call");
    K *in = d_keys, *out = d_tmp;
    uint32_t *vin = d_vals, *vout = d_vals_tmp;
    // digit plan: 8-bit digits, except that 9-bit digits at the high end are used when they save a whole pass
    const uint32_t total = hi_bit - lo_bit;
    const uint32_t p8 = (total + 7) / 8, p9 = (total + 8) / 9;
    uint32_t n9 = 0;
    if (p9 < p8 && total > 8 * p9) n9 = total - 8 * p9;  // 61 bits: 7 passes, 5 of them 9 bits wide
    if (const char *e = getenv("CRGPU_SORT_DIGITS"))     // "8": 8-bit digits only (A/B)
        if (atoi(e) == 8) n9 = 0;
    const uint32_t passes = n9 ? p9 : p8;
    if (sizeof(K) == 8) {
        SweepPlan plan;
        uint32_t widths[OS_MAX_PASSES];
        // the passes sort on the top bits only when that saves one; k_finish_runs orders the `low` bits afterwards
        const uint32_t low = (lo_bit == 0 && ctx->layout.set && hi_bit == ctx->layout.total_bits())
                                 ? cr_sort_low_bits(hi_bit, ctx->layout.bits_umi) : 0u;
        const bool sweep = cr_sweep_plan(lo_bit + low, hi_bit, &plan, widths);
        KeyHistograms &gh = ctx->ghist;
        const bool have_hist = sweep && gh.valid && gh.d_keys == (const uint64_t *)d_keys && gh.n == n &&
                               memcmp(&gh.plan, &plan, sizeof(plan)) == 0;
        gh.valid = false;  // consumed (or stale) either way
        if (sweep) {
            uint64_t *k0 = (uint64_t *)d_keys, *k1 = (uint64_t *)d_tmp;
            if (d_vals)
                CR_TRY(onesweep_sort_u64<true>(ctx, k0, k1, d_vals, d_vals_tmp, n, plan, widths, result_in_tmp,
                                               have_hist ? gh.d_hist : nullptr));
            else
                CR_TRY(onesweep_sort_u64<false>(ctx, k0, k1, nullptr, nullptr, n, plan, widths, result_in_tmp,
                                                have_hist ? gh.d_hist : nullptr));
            if (!low) return CRGPU_OK;
            if (low_left) {
                *low_left = low;
                return CRGPU_OK;
            }
            bool fell_back = false;
            if (cr_sort_finish_experiment())
                CR_TRY(finish_runs(ctx, *result_in_tmp ? k1 : k0, d_vals ? (*result_in_tmp ? d_vals_tmp : d_vals) : nullptr, n, low,
                                   &fell_back));
            else
                CR_TRY(order_low_bits(ctx, *result_in_tmp ? k1 : k0, d_vals ? (*result_in_tmp ? d_vals_tmp : d_vals) : nullptr, n, low,
                                      &fell_back));
            if (!fell_back) return CRGPU_OK;
            // a run of equal top bits too long for the finishing pass (it may have moved keys inside other runs: the
            // buffer still holds every key): sort it again on all bits
            ctx->sort_refinished++;
            SweepPlan full;
            uint32_t fw[OS_MAX_PASSES];
            CR_REQUIRE(ctx, cr_sweep_plan(lo_bit, hi_bit, &full, fw), CRGPU_EHIP, "sort: no plan for the full key");
            bool flip = false;
            uint64_t *a = *result_in_tmp ? k1 : k0, *b = *result_in_tmp ? k0 : k1;
            uint32_t *va = *result_in_tmp ? d_vals_tmp : d_vals, *vb = *result_in_tmp ? d_vals : d_vals_tmp;
            if (d_vals)
                CR_TRY(onesweep_sort_u64<true>(ctx, a, b, va, vb, n, full, fw, &flip, nullptr));
            else
                CR_TRY(onesweep_sort_u64<false>(ctx, a, b, nullptr, nullptr, n, full, fw, &flip, nullptr));
            if (flip) *result_in_tmp = !*result_in_tmp;
            return CRGPU_OK;
        }
    }
    uint32_t shift = lo_bit;
    for (uint32_t pass = 0; pass < passes && shift < hi_bit; pass++) {
        const bool wide = pass >= passes - n9;
        const uint32_t width = wide ? 9u : 8u;
        const uint32_t bits = hi_bit - shift < width ? hi_bit - shift : width;
        RadixDigit dig{shift, (1u << bits) - 1u};
        if (wide)
            CR_TRY((radix_pass<K, RadixDigit, 9>(ctx, in, out, vin, vout, n, dig)));
        else
            CR_TRY((radix_pass<K, RadixDigit, 8>(ctx, in, out, vin, vout, n, dig)));
        shift += width;
        K *t = in;
        in = out;
        out = t;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
        *result_in_tmp = !*result_in_tmp;
    }
    return CRGPU_OK;
}

// Sort d_keys[0..n) ascending on bits [lo_bit, hi_bit).  d_tmp (n keys) and, when d_vals != NULL,
// d_vals_tmp (n u32) are ping-pong buffers; *result_in_tmp tells where the sorted data ended up.
int cr_radix_sort_u64(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                      uint64_t n, uint32_t lo_bit, uint32_t hi_bit, bool *result_in_tmp) {
    return radix_sort<uint64_t>(ctx, d_keys, d_tmp, d_vals, d_vals_tmp, n, lo_bit, hi_bit, result_in_tmp);
}
// all key bits by radix passes, whatever CRGPU_SORT_FINISH says (the fallback of the fused finishing pass)
int cr_radix_sort_u64_full(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp, uint64_t n,
                           uint32_t hi_bit, bool *result_in_tmp) {
    *result_in_tmp = false;
    if (n <= 1) return CRGPU_OK;
    SweepPlan full;
    uint32_t fw[OS_MAX_PASSES];
    ctx->ghist.valid = false;
    if (!cr_sweep_plan(0, hi_bit, &full, fw))  // CRGPU_SORT=classic: the generic path sorts all bits anyway
        return radix_sort<uint64_t>(ctx, d_keys, d_tmp, d_vals, d_vals_tmp, n, 0, hi_bit, result_in_tmp);
    if (d_vals) return onesweep_sort_u64<true>(ctx, d_keys, d_tmp, d_vals, d_vals_tmp, n, full, fw, result_in_tmp, nullptr);
    return onesweep_sort_u64<false>(ctx, d_keys, d_tmp, nullptr, nullptr, n, full, fw, result_in_tmp, nullptr);
}
int cr_radix_sort_u64_top(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                          uint64_t n, uint32_t hi_bit, bool *result_in_tmp, uint32_t *low_left) {
    return radix_sort<uint64_t>(ctx, d_keys, d_tmp, d_vals, d_vals_tmp, n, 0, hi_bit, result_in_tmp, low_left);
}

// ---- finishing pass fused with the run-length pass -----------------------------------------------------------------------
// The keys are sorted by their top bits (stable otherwise).  One pass over them orders every run of equal top bits by its
// low bits in LDS (as k_finish_runs does) AND emits the distinct keys (ignoring the utype bit 0) with the position of their
// first read -- what the two HeadFlag compaction launches did on fully sorted keys -- so the finishing costs no pass of
// its own.  Tiles are handed out by tickets and get the number of distinct keys before them from a decoupled look-back
// (one status word per tile), which keeps the output in key order.  WRITE_BACK (the per-read DupInfo path): the ordered
// keys and payloads also go back to the buffer.  A run that outgrows the staged window raises *fallback (nothing of that
// tile is emitted or written; the chain still moves on) and the host redoes everything the slow way.
#define FE_TILE 2048u
#define FE_HALO 512u
#define FE_CAP (1u + FE_TILE + FE_HALO)
#define FE_ROUNDS ((FE_CAP + 255u) / 256u)   // 11
#define FE_WORDS ((FE_CAP + 63u) / 64u)      // 41
template <bool WRITE_BACK>
__global__ __launch_bounds__(256) void k_finish_emit(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint64_t n,
                                                     uint32_t low_bits, uint64_t *__restrict__ ukey, uint32_t *__restrict__ upos,
                                                     unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket,
                                                     unsigned long long *__restrict__ n_out, uint32_t *__restrict__ fallback) {
    __shared__ __attribute__((aligned(16))) uint64_t sk[FE_CAP];
    __shared__ uint16_t slo[FE_CAP], spos[FE_CAP];
    __shared__ uint32_t sv[WRITE_BACK ? FE_CAP : 1];
    __shared__ unsigned long long s_head[FE_WORDS + 1];
    __shared__ uint32_t s_cnt[FE_ROUNDS * 4 + 4];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_tile, s_a, s_b, s_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t n_tiles = (n + FE_TILE - 1) / FE_TILE;
    const uint32_t lowmask = (1u << low_bits) - 1u;
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint64_t tile = s_tile;
        if (tile >= n_tiles) break;  // uniform
        const uint64_t t0 = tile * FE_TILE, t1 = t0 + FE_TILE < n ? t0 + FE_TILE : n;
        const uint64_t w0 = t0 ? t0 - 1 : 0, w1 = t1 + FE_HALO < n ? t1 + FE_HALO : n;
        const uint32_t cnt = (uint32_t)(w1 - w0), off0 = (uint32_t)(t0 - w0), off1 = (uint32_t)(t1 - w0);
        if (tid == 0) s_bad = 0u;
        for (uint32_t i = tid; i < cnt; i += 256) {
            const uint64_t k = keys[w0 + i];
            sk[i] = k;
            slo[i] = (uint16_t)((uint32_t)k & lowmask);
            if (WRITE_BACK) sv[i] = vals[w0 + i];
        }
        __syncthreads();
        // run heads (equal top bits) as a bit mask: one ballot per 64 positions
#pragma unroll 1
        for (uint32_t j = 0; j < FE_ROUNDS; j++) {
            const uint32_t i = j * 256u + tid;
            bool h = false;
            if (i < cnt) h = i == 0 ? (w0 == 0) : (sk[i] >> low_bits) != (sk[i - 1] >> low_bits);
            const unsigned long long m = __ballot(h);
            if (lane == 0 && j * 4u + wave < FE_WORDS + 1) s_head[j * 4u + wave] = m;
        }
        __syncthreads();
        if (tid == 0) {
            // [ra, rb): the whole runs whose head lies in the tile
            uint32_t ra = 0xFFFFFFFFu, rb = 0xFFFFFFFFu;
            for (uint32_t w = off0 >> 6; w < FE_WORDS && ra == 0xFFFFFFFFu; w++) {
                unsigned long long m = s_head[w];
                if (w == (off0 >> 6)) m &= ~0ull << (off0 & 63u);
                if (m) ra = w * 64u + (uint32_t)(__ffsll((long long)m) - 1);
            }
            for (uint32_t w = off1 >> 6; w < FE_WORDS && rb == 0xFFFFFFFFu; w++) {
                unsigned long long m = s_head[w];
                if (w == (off1 >> 6)) m &= ~0ull << (off1 & 63u);
                if (m) rb = w * 64u + (uint32_t)(__ffsll((long long)m) - 1);
            }
            if (rb == 0xFFFFFFFFu && w1 == n) rb = cnt;  // the last run ends with the array
            if (ra != 0xFFFFFFFFu && ra >= off1) ra = 0xFFFFFFFFu;  // no run begins in this tile
            if (ra != 0xFFFFFFFFu && rb == 0xFFFFFFFFu) s_bad = 1u;  // the tile's last run goes on beyond the halo
            s_a = ra;
            s_b = rb;
        }
        __syncthreads();
        const uint32_t ra = s_a, rb = s_b;
        const bool active = ra != 0xFFFFFFFFu && !s_bad;
        // place of every key inside its run: bounds from the bit mask, rank by the 16-bit low parts
        uint64_t mykey[FE_ROUNDS];
        uint32_t myval[FE_ROUNDS];
        if (active) {
#pragma unroll 1
            for (uint32_t j = 0; j < FE_ROUNDS; j++) {
                const uint32_t i = j * 256u + tid;
                if (i < ra || i >= rb) continue;
                // s = last head at or before i
                uint32_t w = i >> 6;
                unsigned long long m = s_head[w] & (~0ull >> (63u - (i & 63u)));
                while (!m) m = s_head[--w];  // position ra is a head: the walk ends there at the latest
                const uint32_t s0 = w * 64u + 63u - (uint32_t)__clzll((long long)m);
                // e = first head after i, or rb
                uint32_t e0 = rb;
                w = i >> 6;
                m = (i & 63u) == 63u ? 0ull : (s_head[w] & (~0ull << ((i & 63u) + 1u)));
                for (;;) {
                    if (m) {
                        const uint32_t c = w * 64u + (uint32_t)(__ffsll((long long)m) - 1);
                        e0 = c < rb ? c : rb;
                        break;
                    }
                    if (++w >= FE_WORDS || w * 64u >= rb) break;
                    m = s_head[w];
                }
                uint32_t rank = i - s0;
                if (e0 - s0 > 1u) {
                    if (e0 - s0 > FE_HALO) {
                        s_bad = 1u;  // quadratic work beyond this: the slow way
                    } else {
                        const uint32_t low = slo[i];
                        rank = 0;
                        for (uint32_t q = s0; q < i; q++) rank += slo[q] <= low;   // earlier position: ties stay in front
                        for (uint32_t q = i + 1; q < e0; q++) rank += slo[q] < low;
                    }
                }
                spos[i] = (uint16_t)(s0 + rank);
            }
        }
        __syncthreads();
        const bool ok = active && !s_bad;
        if (ok) {
            // permute in place through registers
#pragma unroll
            for (uint32_t j = 0; j < FE_ROUNDS; j++) {
                const uint32_t i = j * 256u + tid;
                if (i >= ra && i < rb) {
                    mykey[j] = sk[i];
                    if (WRITE_BACK) myval[j] = sv[i];
                }
            }
        }
        __syncthreads();
        if (ok) {
#pragma unroll
            for (uint32_t j = 0; j < FE_ROUNDS; j++) {
                const uint32_t i = j * 256u + tid;
                if (i >= ra && i < rb) {
                    const uint32_t p = spos[i];
                    sk[p] = mykey[j];
                    if (WRITE_BACK) sv[p] = myval[j];
                }
            }
        }
        __syncthreads();
        // distinct keys (ignoring the utype bit) among [ra, rb): ra starts a run of equal top bits, so it is one
        uint32_t below[FE_ROUNDS];
#pragma unroll
        for (uint32_t j = 0; j < FE_ROUNDS; j++) {
            const uint32_t i = j * 256u + tid;
            const bool d = ok && i >= ra && i < rb && (i == ra || (sk[i] >> 1) != (sk[i - 1] >> 1));
            const unsigned long long m = __ballot(d);
            below[j] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_cnt[j * 4u + wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the FE_ROUNDS * 4 = 44 (round, wave) counts + the look-back, by wave 0
            const uint32_t v = tid < FE_ROUNDS * 4u ? s_cnt[tid] : 0u;
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (tid >= (uint32_t)d) x += y;
            }
            if (tid < FE_ROUNDS * 4u) s_cnt[tid] = x - v;
            const uint32_t total = __shfl(x, 63);
            if (tid == 0 && tile > 0)
                __hip_atomic_store(&status[tile], OS_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long excl = wave_lookback(status, tile);  // every lower ticket is running or done
            if (tid == 0) {
                __hip_atomic_store(&status[tile], OS_INC | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tile + 1 == n_tiles) *n_out = excl + total;
                s_base = excl;
            }
        }
        __syncthreads();
        if (ok) {
            const unsigned long long base = s_base;
#pragma unroll
            for (uint32_t j = 0; j < FE_ROUNDS; j++) {
                const uint32_t i = j * 256u + tid;
                if (i < ra || i >= rb) continue;
                const uint64_t k = sk[i];
                if (i == ra || (k >> 1) != (sk[i - 1] >> 1)) {
                    const unsigned long long o = base + s_cnt[j * 4u + wave] + below[j];
                    ukey[o] = k;
                    upos[o] = (uint32_t)(w0 + i);
                }
                if (WRITE_BACK) {
                    keys[w0 + i] = k;
                    vals[w0 + i] = sv[i];
                }
            }
        } else if (s_bad && tid == 0) {
            atomicOr(fallback, 1u);
        }
        __syncthreads();
    }
}

// Finishes a sort that cr_radix_sort_u64_top left with `low_bits` unsorted low bits and emits the distinct keys:
// ukey / upos (room for n entries), *nd_out = their number.  *fell_back: a run was too long; ukey / upos are garbage, the
// keys (and payloads) are untouched or partly ordered inside their runs -- still the same multiset.
int cr_finish_emit(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, uint64_t *d_ukey,
                   uint32_t *d_upos, uint64_t *nd_out, bool *fell_back) {
    *fell_back = false;
    *nd_out = 0;
    if (n == 0) return CRGPU_OK;
    const uint64_t n_tiles = (n + FE_TILE - 1) / FE_TILE;
    void *d_status = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &d_status, n_tiles * sizeof(unsigned long long) + 64));
    uint32_t *d_ticket = ctx->d_scalars + 44, *d_flag = ctx->d_scalars + 52;
    unsigned long long *d_n = (unsigned long long *)(ctx->d_scalars + 8);
    hipError_t e;
    {
        CrTimer t(ctx, CRGPU_T_DEDUP, n);  // it is the run-length pass of the dedup family (which counts its keys here)
        e = hipMemsetAsync(d_status, 0, n_tiles * sizeof(unsigned long long), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_ticket, 0, sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(unsigned long long), ctx->stream);
        const dim3 grid((unsigned)(n_tiles < 2048 ? n_tiles : 2048));
        if (d_vals)
            hipLaunchKernelGGL(k_finish_emit<true>, grid, dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits, d_ukey, d_upos,
                               (unsigned long long *)d_status, d_ticket, d_n, d_flag);
        else
            hipLaunchKernelGGL(k_finish_emit<false>, grid, dim3(256), 0, ctx->stream, d_keys, d_vals, n, low_bits, d_ukey, d_upos,
                               (unsigned long long *)d_status, d_ticket, d_n, d_flag);
        if (e == hipSuccess) e = hipGetLastError();
    }
    cr_pool_free(ctx, d_status);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "finish + emit: %s", hipGetErrorString(e));
    uint32_t flag = 0;
    unsigned long long nd = 0;
    CR_TRY(crgpu_memcpy_d2h(ctx, &flag, d_flag, sizeof(flag)));
    CR_TRY(crgpu_memcpy_d2h(ctx, &nd, d_n, sizeof(nd)));
    *fell_back = flag != 0;
    *nd_out = nd;
    return CRGPU_OK;
}

int cr_radix_sort_u32(crgpu_ctx *ctx, uint32_t *d_keys, uint32_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                      uint64_t n, uint32_t lo_bit, uint32_t hi_bit, bool *result_in_tmp) {
    return radix_sort<uint32_t>(ctx, d_keys, d_tmp, d_vals, d_vals_tmp, n, lo_bit, hi_bit, result_in_tmp);
}

// Stable partition of keys by the owner rank of their barcode: one counting-sort pass.
// d_vin / d_vout (nullable): a 32-bit payload (the read ordinals) that travels with the keys
int cr_partition_by_owner_kv(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, const uint32_t *d_vin, uint32_t *d_vout,
                             uint64_t n, uint32_t sh_bc, uint32_t n_ranks, const uint32_t *bounds, uint64_t *counts_out) {
    CR_REQUIRE(ctx, n_ranks >= 1 && n_ranks <= RADIX, CRGPU_EINVAL, "partition: n_ranks must be 1..256");
    CR_REQUIRE(ctx, n <= 0x7FFFFFFFull, CRGPU_ERANGE, "partition: at most 2^31-1 keys per call");
    for (uint32_t r = 0; r < n_ranks; r++) counts_out[r] = 0;
    if (n == 0) return CRGPU_OK;
    std::vector<uint32_t> col_bounds;
    if (ctx->dense.valid) {
        // CRGPU_OPT_DENSE_BARCODE_KEYS: the keys hold columns of the BarcodeIndex, the owner ranges are whitelist ranks: the
        // range [lo, hi) of ranks is the range of the columns whose rank lies in it (columns ascend with the ranks)
        const std::vector<uint32_t> &back = ctx->dense.h_back;
        const uint32_t width = (ctx->n_canon + n_ranks - 1) / n_ranks;
        col_bounds.resize(n_ranks + 1);
        for (uint32_t r = 0; r <= n_ranks; r++) {
            const uint64_t lo = bounds ? bounds[r] : (uint64_t)r * (width ? width : 1u);
            col_bounds[r] = (uint32_t)(std::lower_bound(back.begin(), back.end(), (uint32_t)std::min<uint64_t>(lo, 0xFFFFFFFFull)) - back.begin());
        }
        if (bounds) {
            CR_REQUIRE(ctx, bounds[0] == 0 && bounds[n_ranks] >= ctx->n_canon, CRGPU_EINVAL,
                       "partition: bounds must start at 0 and end at or beyond the whitelist size");
            for (uint32_t r = 0; r < n_ranks; r++)
                CR_REQUIRE(ctx, bounds[r] <= bounds[r + 1], CRGPU_EINVAL, "partition: bounds must be ascending");
        }
        col_bounds[0] = 0;
        col_bounds[n_ranks] = ctx->dense.V;
        uint32_t *d_bounds = ctx->d_scalars + 760;
        CR_HIP(ctx, hipMemcpyAsync(d_bounds, col_bounds.data(), (n_ranks + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        CR_TRY((radix_pass<uint64_t, OwnerBounds>(ctx, d_in, d_out, d_vin, d_vout, n, OwnerBounds{sh_bc, n_ranks, d_bounds})));
    } else if (bounds) {
        // rank r owns canonical barcode ranks [bounds[r], bounds[r+1])
        CR_REQUIRE(ctx, bounds[0] == 0 && bounds[n_ranks] >= ctx->n_canon, CRGPU_EINVAL,
                   "partition: bounds must start at 0 and end at or beyond the whitelist size");
        for (uint32_t r = 0; r < n_ranks; r++)
            CR_REQUIRE(ctx, bounds[r] <= bounds[r + 1], CRGPU_EINVAL, "partition: bounds must be ascending");
        uint32_t *d_bounds = ctx->d_scalars + 760;  // 257 u32 inside the 1024-word scalar page
        CR_HIP(ctx, hipMemcpyAsync(d_bounds, bounds, (n_ranks + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's array may be a temporary
        CR_TRY((radix_pass<uint64_t, OwnerBounds>(ctx, d_in, d_out, d_vin, d_vout, n, OwnerBounds{sh_bc, n_ranks, d_bounds})));
    } else {
        // rank r owns canonical barcode ranks [r*width, (r+1)*width)
        const uint32_t width = (ctx->n_canon + n_ranks - 1) / n_ranks;
        CR_TRY((radix_pass<uint64_t, OwnerDiv>(ctx, d_in, d_out, d_vin, d_vout, n, OwnerDiv{sh_bc, width ? width : 1u})));
    }
    uint32_t totals[RADIX];
    CR_TRY(crgpu_memcpy_d2h(ctx, totals, digit_totals_buf(ctx), sizeof(totals)));
    for (uint32_t r = 0; r < n_ranks; r++) counts_out[r] = totals[r];
    return CRGPU_OK;
}

int cr_partition_by_owner(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, uint32_t sh_bc,
                          uint32_t n_ranks, const uint32_t *bounds, uint64_t *counts_out) {
    return cr_partition_by_owner_kv(ctx, d_in, d_out, nullptr, nullptr, n, sh_bc, n_ranks, bounds, counts_out);
}

// One stable counting pass of (64-bit value, 32-bit payload) pairs keyed by the top bits of the PAYLOAD: groups the
// per-read records of the DupInfo path by windows of read ordinals before they are scattered to the reads.
int cr_partition_by_payload(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, const uint32_t *d_vin, uint32_t *d_vout,
                            uint64_t n, uint32_t shift) {
    if (n == 0) return CRGPU_OK;
    return radix_pass<uint64_t, PayloadDigit, 9>(ctx, d_in, d_out, d_vin, d_vout, n, PayloadDigit{shift, 511u}, CRGPU_T_DEDUP);
}
