// barcode.hip -- pack, K1 (exact match + valid histogram) and K2 (posterior 1-mismatch correction).
//
// K1 replaces Whitelist::check_and_update (barcode/src/whitelist.rs:494-517) per read and
//    MakeShardHistograms::observe (cr_lib/src/make_shard_metrics.rs:171-188).
// K2 replaces Posterior::correct_barcode (barcode/src/corrector.rs:111-165) driven by
//    correct_barcode_in_read (cr_lib/src/stages/barcode_correction.rs:76-99).
//
// Integer / f64 work, HBM- and cache-latency bound: no MFMA.  Built with -ffp-contract=off so the
// f64 multiply and the running sum are never fused (the reference's rustc flags have no +fma,
// lib/rust/.cargo/config.toml:5-8).
#include <cstdlib>

#include "common.h"
#include "block_utils.h"
#include "wl_view.h"

int cr_make_views(crgpu_ctx *ctx, WlView *views);

// Streams that are touched once (barcodes, flag bytes, quality rows, the index output) go around the caches' normal
// replacement so that they do not evict the whitelist tables, which every cold lookup and every K2 probe needs in L2.
typedef uint32_t cr_u32x4 __attribute__((ext_vector_type(4)));
#ifdef CR_NO_NT
#define CR_LOAD_STREAM(p) (*(p))
#define CR_STORE_STREAM(v, p) (*(p) = (v))
#else
#define CR_LOAD_STREAM(p) __builtin_nontemporal_load(p)
#define CR_STORE_STREAM(v, p) __builtin_nontemporal_store((v), (p))
#endif

struct WlViewSet {
    WlView v[CRGPU_MAX_LIB];
    uint32_t n_canon;
    // one-library calls (the UNIFORM kernels): v[0] holds the tables of library `ulib`, and reads whose flag byte
    // names another library are misses
    uint32_t ulib;
};

// ------------------------------------------------------------------------------------------------
// pack: ASCII bases + ASCII qualities -> 2-bit word + N-flagged quality bytes
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint32_t c, bool &is_n) {
    // 'A' 0x41 'C' 0x43 'G' 0x47 'T' 0x54 : (c>>1)&3 = 0,1,3,2 ; swap the last two
    uint32_t x = (c >> 1) & 3u;
    x ^= x >> 1;
    is_n = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
    return is_n ? 0u : x;
}

__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              uint64_t n, uint32_t len, uint32_t *__restrict__ packed,
                                              uint8_t *__restrict__ qualn, uint8_t *__restrict__ flags,
                                              uint32_t *__restrict__ n_without_flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t key = 0;
        bool any_n = false;
        if (len == 16) {
            const uint4 s4 = *reinterpret_cast<const uint4 *>(seq + i * 16);
            const uint4 q4 = *reinterpret_cast<const uint4 *>(qual + i * 16);
            const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w};
            uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t outq = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t c = (sw[w] >> (8 * b)) & 0xFFu;
                    uint32_t q = (qw[w] >> (8 * b)) & 0xFFu;
                    bool is_n;
                    const uint32_t code = base_code(c, is_n);
                    key = (key << 2) | code;
                    any_n |= is_n;
                    q = q > 127u ? 127u : q;
                    outq |= (q | (is_n ? 0x80u : 0u)) << (8 * b);
                }
                qw[w] = outq;
            }
            *reinterpret_cast<uint4 *>(qualn + i * 16) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
        } else {
            for (uint32_t j = 0; j < len; j++) {
                const uint32_t c = seq[i * len + j];
                uint32_t q = qual[i * len + j];
                bool is_n;
                const uint32_t code = base_code(c, is_n);
                key = (key << 2) | code;
                any_n |= is_n;
                q = q > 127u ? 127u : q;
                qualn[i * len + j] = (uint8_t)(q | (is_n ? 0x80u : 0u));
            }
        }
        packed[i] = key;
        if (flags && any_n) flags[i] |= CRGPU_FLAG_CB_HAS_N;
        if (!flags && any_n) atomicOr(n_without_flags, 1u);  // rare; remembered for a pass A that runs without flags
    }
}

extern "C" int crgpu_pack_dev(crgpu_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, uint32_t len,
                              uint32_t *d_packed_out, uint8_t *d_qualn_out, uint8_t *d_flags_inout) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE, "sequence length %u unsupported (<= 16)", len);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq && d_qual && d_packed_out && d_qualn_out, CRGPU_EINVAL, "crgpu_pack_dev: NULL buffer");
    if (len == 16)
        CR_REQUIRE(ctx, ((uintptr_t)d_seq | (uintptr_t)d_qual | (uintptr_t)d_qualn_out) % 16 == 0, CRGPU_EINVAL,
                   "crgpu_pack_dev: 16-base buffers must be 16-byte aligned");
    CrTimer t(ctx, CRGPU_T_PACK, n);
    cr_invalidate(ctx);
    hipLaunchKernelGGL(k_pack, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_seq, d_qual, n, len, d_packed_out,
                       d_qualn_out, d_flags_inout, ctx->d_scalars + CR_SCALAR_N_WITHOUT_FLAGS);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// The same packing straight from whole read rows (R1 as the FASTQ holds it: row_stride bytes per read): bases
// [offset, offset + len) of every row, i.e. the slicing of RnaRead's barcode / UMI ranges
// (cr_types/src/rna_read.rs:103-138) done on the device, so the host uploads R1 once for both parts.
__global__ __launch_bounds__(256) void k_pack_rows(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, uint64_t n,
                                                   uint32_t row_stride, uint32_t offset, uint32_t len,
                                                   uint32_t *__restrict__ packed, uint8_t *__restrict__ qualn,
                                                   uint8_t *__restrict__ flags, uint32_t *__restrict__ n_without_flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t *s = seq + i * row_stride + offset, *q = qual + i * row_stride + offset;
        uint32_t key = 0;
        bool any_n = false;
        for (uint32_t j = 0; j < len; j++) {
            bool is_n;
            const uint32_t code = base_code(s[j], is_n);
            key = (key << 2) | code;
            any_n |= is_n;
            const uint32_t qq = q[j] > 127u ? 127u : q[j];
            qualn[i * len + j] = (uint8_t)(qq | (is_n ? 0x80u : 0u));
        }
        packed[i] = key;
        if (flags && any_n) flags[i] |= CRGPU_FLAG_CB_HAS_N;
        if (!flags && any_n) atomicOr(n_without_flags, 1u);
    }
}

extern "C" int crgpu_pack_rows_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows, uint64_t n,
                                   uint32_t row_stride, uint32_t offset, uint32_t len, uint32_t *d_packed_out,
                                   uint8_t *d_qualn_out, uint8_t *d_flags_inout) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE, "sequence length %u unsupported (<= 16)", len);
    CR_REQUIRE(ctx, (uint64_t)offset + len <= row_stride, CRGPU_EINVAL, "crgpu_pack_rows_dev: bases [%u, %u) lie outside a row of %u bytes",
               offset, offset + len, row_stride);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq_rows && d_qual_rows && d_packed_out && d_qualn_out, CRGPU_EINVAL, "crgpu_pack_rows_dev: NULL buffer");
    CrTimer t(ctx, CRGPU_T_PACK, n);
    cr_invalidate(ctx);
    hipLaunchKernelGGL(k_pack_rows, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_seq_rows, d_qual_rows, n, row_stride, offset,
                       len, d_packed_out, d_qualn_out, d_flags_inout, ctx->d_scalars + CR_SCALAR_N_WITHOUT_FLAGS);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// UmiExtractor::extract_umi (cr_types/src/rna_read.rs:103-138): the UMI of a read that ends early keeps
// max(min(read_len - offset, length), min_length) bases; a range beyond the end of the read fails check_range.
__global__ __launch_bounds__(256) void k_pack_rows_var(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                       const uint32_t *__restrict__ read_len, uint64_t n, uint32_t row_stride,
                                                       uint32_t offset, uint32_t length, uint32_t min_length,
                                                       uint32_t *__restrict__ packed, uint8_t *__restrict__ qualn,
                                                       uint8_t *__restrict__ len_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t rl = read_len[i] < row_stride ? read_len[i] : row_stride;
        const uint32_t avail = rl > offset ? rl - offset : 0u;            // saturating_sub
        uint32_t L = avail < length ? avail : length;
        L = L > min_length ? L : min_length;
        if (offset + L > rl) L = 0;                                        // check_range(&range, "UMI") fails
        const uint8_t *s = seq + i * row_stride + offset, *q = qual + i * row_stride + offset;
        uint32_t key = 0;
        for (uint32_t j = 0; j < length; j++) {
            uint8_t out = 0;
            if (j < L) {
                bool is_n;
                const uint32_t code = base_code(s[j], is_n);
                key = (key << 2) | code;
                const uint32_t qq = q[j] > 127u ? 127u : q[j];
                out = (uint8_t)(qq | (is_n ? 0x80u : 0u));
            }
            qualn[i * length + j] = out;
        }
        packed[i] = key;
        len_out[i] = (uint8_t)L;
    }
}

extern "C" int crgpu_pack_rows_var_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows,
                                       const uint32_t *d_read_len, uint64_t n, uint32_t row_stride, uint32_t offset, uint32_t length,
                                       uint32_t min_length, uint32_t *d_packed_out, uint8_t *d_qualn_out, uint8_t *d_len_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, length >= 1 && length <= 16 && min_length >= 1 && min_length <= length, CRGPU_ERANGE,
               "crgpu_pack_rows_var_dev: lengths %u..%u unsupported (1 <= min <= length <= 16)", min_length, length);
    CR_REQUIRE(ctx, (uint64_t)offset + length <= row_stride, CRGPU_EINVAL, "crgpu_pack_rows_var_dev: bases [%u, %u) lie outside a row of %u bytes",
               offset, offset + length, row_stride);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq_rows && d_qual_rows && d_read_len && d_packed_out && d_qualn_out && d_len_out, CRGPU_EINVAL,
               "crgpu_pack_rows_var_dev: NULL buffer");
    CrTimer t(ctx, CRGPU_T_PACK, n);
    cr_invalidate(ctx);
    hipLaunchKernelGGL(k_pack_rows_var, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_seq_rows, d_qual_rows, d_read_len, n,
                       row_stride, offset, length, min_length, d_packed_out, d_qualn_out, d_len_out);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// K1: exact match + valid-barcode histogram
// ------------------------------------------------------------------------------------------------
template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_match(const WlViewSet vs, const uint32_t *__restrict__ cb,
                                               const uint8_t *__restrict__ flags, uint64_t n,
                                               uint32_t *__restrict__ idx_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t key = cb[i];
        const uint32_t f = flags ? flags[i] : 0u;
        uint32_t rank = CRGPU_MISS;
        const uint32_t lib = f & CRGPU_FLAG_LIB_MASK;
        if (!(f & CRGPU_FLAG_CB_HAS_N)) {
            if (UNIFORM) {
                if (lib == vs.ulib) rank = wl_lookup(vs.v[0], key);
            } else {
                if (vs.v[lib].n) rank = wl_lookup(vs.v[lib], key);
            }
        }
        idx_out[i] = rank;
        if (rank != CRGPU_MISS) atomicAdd(UNIFORM ? &vs.v[0].valid[rank] : &vs.v[lib].valid[rank], 1u);
    }
}

// ---- which calls take the one-library kernels ---------------------------------------------------------------
// MAKE_SHARD and BARCODE_CORRECTION hand over the reads of ONE library at a time (a FASTQ / a library type per chunk),
// whatever the number of libraries in the GEM well.  k_lib_mask finds that out from the flag bytes when several
// whitelists are set (1 byte per read, one host read of 4 bytes); with a single whitelist nothing is scanned.
__global__ __launch_bounds__(256) void k_lib_mask(const uint8_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ mask_out) {
    uint32_t m = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        m |= 1u << (flags[i] & CRGPU_FLAG_LIB_MASK);
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
    if ((threadIdx.x & 63u) == 0u && m) atomicOr(mask_out, m);
}

// *ulib_out = the library all reads of the call belong to, or -1 (several libraries, or none with a whitelist)
static int pick_uniform_lib(crgpu_ctx *ctx, const uint8_t *d_flags, uint64_t n, int *ulib_out) {
    *ulib_out = -1;
    int n_set = 0, only = -1;
    for (int l = 0; l < CRGPU_MAX_LIB; l++)
        if (ctx->wl[l].set) {
            n_set++;
            only = l;
        }
    if (n_set == 0) return CRGPU_OK;
    if (n_set == 1) {  // reads of any other library are misses either way
        *ulib_out = only;
        return CRGPU_OK;
    }
    if (!d_flags) {  // no flag bytes: every read is of library 0
        if (ctx->wl[0].set) *ulib_out = 0;
        return CRGPU_OK;
    }
    uint32_t *d_mask = ctx->d_scalars + 28, mask = 0;
    CR_HIP(ctx, hipMemsetAsync(d_mask, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_lib_mask, dim3(cr_grid(n, 256 * 16)), dim3(256), 0, ctx->stream, d_flags, n, d_mask);
    CR_HIP(ctx, hipGetLastError());
    CR_TRY(crgpu_memcpy_d2h(ctx, &mask, d_mask, sizeof(mask)));
    if (mask != 0u && (mask & (mask - 1u)) == 0u) {
        const int l = __builtin_ctz(mask);
        if (l < CRGPU_MAX_LIB && ctx->wl[l].set) *ulib_out = l;
    }
    return CRGPU_OK;
}

// views for the kernels; ulib >= 0: the tables of that library are presented as v[0]
static int make_view_set(crgpu_ctx *ctx, WlViewSet &vs, int ulib) {
    CR_TRY(cr_make_views(ctx, vs.v));
    vs.n_canon = ctx->n_canon;
    vs.ulib = ulib > 0 ? (uint32_t)ulib : 0u;
    if (ulib > 0) {
        const WlView t = vs.v[0];
        vs.v[0] = vs.v[ulib];
        vs.v[ulib] = t;
    }
    return CRGPU_OK;
}

// ---- K1 with an LDS-binned histogram ---------------------------------------------------------------
// Scattered device-scope atomics run at ~20 G/s on this chip (one 64-B fabric request per lane), which
// made the histogram 5x more expensive than the lookups.  Instead:
//   k_match_binned  looks the reads up, writes idx_out, and appends (rank & 0x7FFF) as u16 to the
//                   staging region of its bucket (bucket = library slot, rank >> 15).  A tile of 4096
//                   reads is ranked by bucket in LDS and copied out in contiguous runs; one global
//                   atomic per (tile, non-empty bucket) reserves the space.
//   k_hist_buckets  one workgroup per (bucket, slice): 32768 u32 counters in LDS (128 KB), streams the
//                   bucket's u16 entries with 16-byte loads, then adds the non-zero counters to the
//                   library's valid table (contiguous atomics, a few per barcode).
#define BIN_SHIFT 15
#define BIN_SIZE (1u << BIN_SHIFT)
#define MB_ITEMS 16
#define MB_TILE (256 * MB_ITEMS)
#define MB_MAX_BUCKETS 1024
#define MB_BPT (MB_MAX_BUCKETS / 256)  // buckets per thread in the tile scan
#define MB_LKB 8                       // reads whose lookups advance together
#define MB_LKG 4                       // ... in a call with several libraries (per-lane table pointers cost registers)
#define MB_CURSOR_STRIDE 32u            // one 128-byte line per bucket cursor: atomics on one line serialise in one L2 channel

struct BinPlan {
    uint32_t lib_slot[CRGPU_MAX_LIB];  // library id -> slot (0xFFFFFFFF = not configured)
    uint32_t buckets_per_lib;
    uint32_t n_buckets;
    uint64_t cap;  // staging entries per bucket
};

// FROM_IDX: the ranks were already written to idx_out by k_lookup_hot; only the histogram staging runs.
template <bool UNIFORM, bool FROM_IDX = false>
__global__ __launch_bounds__(256) void k_match_binned(const WlViewSet vs, const BinPlan plan,
                                                      const uint32_t *__restrict__ cb, const uint8_t *__restrict__ flags,
                                                      uint64_t n, uint32_t *__restrict__ idx_out,
                                                      uint16_t *__restrict__ stage, uint32_t *__restrict__ cursor,
                                                      const uint32_t *__restrict__ skip_if = nullptr) {
    __shared__ uint32_t bcnt[MB_MAX_BUCKETS];    // tile counts, then global base of each bucket
    __shared__ uint32_t bstart[MB_MAX_BUCKETS];  // tile-local exclusive prefix
    __shared__ uint16_t sval[MB_TILE];
    __shared__ uint16_t sbkt[MB_TILE];
    __shared__ uint32_t lds[8];
    __shared__ uint32_t tile_hits;
    if (skip_if && *skip_if) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t NB = plan.n_buckets;
    const uint64_t n_tiles = (n + MB_TILE - 1) / MB_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (uint32_t b = tid; b < NB; b += 256) bcnt[b] = 0;
        __syncthreads();
        uint32_t bv[MB_ITEMS];    // (bucket << 16) | value, 0xFFFFFFFF = no hit
        uint16_t loff[MB_ITEMS];  // arrival order inside (tile, bucket)
        if constexpr (FROM_IDX) {
#pragma unroll
            for (int j = 0; j < MB_ITEMS; j++) {
                const uint64_t i = tile * MB_TILE + (uint64_t)j * 256 + tid;
                const uint32_t rank = i < n ? idx_out[i] : CRGPU_MISS;
                bv[j] = 0xFFFFFFFFu;
                loff[j] = 0;
                if (rank != CRGPU_MISS) {
                    const uint32_t b = rank >> BIN_SHIFT;
                    loff[j] = (uint16_t)atomicAdd(&bcnt[b], 1u);
                    bv[j] = (b << 16) | (rank & (BIN_SIZE - 1u));
                }
            }
        } else if constexpr (UNIFORM) {
            // One library: the lookups of MB_LKB reads advance together, phase by phase (keys -> index bins ->
            // bin contents), so that the loads of a phase are all in flight at once.  With one read at a
            // time behind its own branches the kernel was a serial chain of 3 memory latencies per read.
            const WlView &w = vs.v[0];
            const uint32_t *__restrict__ tw = reinterpret_cast<const uint32_t *>(w.tailA);
            const uint32_t tail_mask = (1u << w.bitsB) - 1u;
#pragma unroll
            for (int j0 = 0; j0 < MB_ITEMS; j0 += MB_LKB) {
                uint32_t key[MB_LKB], lo[MB_LKB], hi[MB_LKB];
                U32x4 d[MB_LKB];
                bool live[MB_LKB];
#pragma unroll
                for (int jj = 0; jj < MB_LKB; jj++) {
                    const uint64_t i = tile * MB_TILE + (uint64_t)(j0 + jj) * 256 + tid;
                    const bool ok = i < n;
                    key[jj] = ok ? cb[i] : 0u;
                    const uint32_t f = (ok && flags) ? flags[i] : 0u;
                    live[jj] = ok && !(f & CRGPU_FLAG_CB_HAS_N) && (f & CRGPU_FLAG_LIB_MASK) == vs.ulib;
                }
#pragma unroll
                for (int jj = 0; jj < MB_LKB; jj++) {
                    // both bounds of the bin in one 8-byte load
                    const uint32_t bin = (uint32_t)((uint64_t)key[jj] >> w.shiftE);
                    const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offE + bin);
                    lo[jj] = b2.a;
                    hi[jj] = b2.b;
                }
#pragma unroll
                for (int jj = 0; jj < MB_LKB; jj++) {
                    // the first 8 entries of the bin in one 16-byte load (the tables are padded by 32 bytes)
                    d[jj] = *reinterpret_cast<const U32x4 *>(tw + (lo[jj] >> 1));
                }
#pragma unroll
                for (int jj = 0; jj < MB_LKB; jj++) {
                    const int j = j0 + jj;
                    const uint64_t i = tile * MB_TILE + (uint64_t)j * 256 + tid;
                    const uint32_t tail = key[jj] & tail_mask;
                    const uint32_t p0 = lo[jj] & ~1u;
                    uint32_t found = CRGPU_MISS;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t pos = p0 + k;
                        const uint32_t t = (d[jj].w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                        if (pos >= lo[jj] && pos < hi[jj] && t == tail) found = pos;
                    }
                    if (hi[jj] > p0 + 8u && found == CRGPU_MISS && live[jj]) {  // a bin of more than 7 keys: rare
                        scan_u16_range<4>(w.tailA, p0 + 8u, hi[jj], [&](uint32_t t, uint32_t pos) {
                            if (t == tail) found = pos;
                        });
                    }
                    uint32_t rank = CRGPU_MISS;
                    if (live[jj] && found != CRGPU_MISS) rank = w.valA ? w.valA[found] : found;
                    bv[j] = 0xFFFFFFFFu;
                    loff[j] = 0;
                    if (i < n) idx_out[i] = rank;
                    if (rank != CRGPU_MISS) {
                        const uint32_t b = rank >> BIN_SHIFT;
                        loff[j] = (uint16_t)atomicAdd(&bcnt[b], 1u);
                        bv[j] = (b << 16) | (rank & (BIN_SIZE - 1u));
                    }
                }
            }
        } else {
            // Several libraries in one call: the same phases with per-lane table pointers (every library has its own
            // index and, for a translation whitelist, its own rank table).  Lanes without a live read follow the tables
            // of `safe_lib` (a library that has a whitelist) so that every load of a phase is unconditional.
            uint32_t safe_lib = 0;
            for (uint32_t l = 0; l < CRGPU_MAX_LIB; l++)
                if (plan.lib_slot[CRGPU_MAX_LIB - 1u - l] != 0xFFFFFFFFu) safe_lib = CRGPU_MAX_LIB - 1u - l;
#pragma unroll
            for (int j0 = 0; j0 < MB_ITEMS; j0 += MB_LKG) {
                uint32_t key[MB_LKG], lo[MB_LKG], hi[MB_LKG], lib[MB_LKG];
                const uint32_t *twl[MB_LKG];
                U32x4 d[MB_LKG];
                bool live[MB_LKG];
#pragma unroll
                for (int jj = 0; jj < MB_LKG; jj++) {
                    const uint64_t i = tile * MB_TILE + (uint64_t)(j0 + jj) * 256 + tid;
                    const bool ok = i < n;
                    key[jj] = ok ? cb[i] : 0u;
                    const uint32_t f = (ok && flags) ? flags[i] : 0u;
                    const uint32_t l = f & CRGPU_FLAG_LIB_MASK;
                    live[jj] = ok && !(f & CRGPU_FLAG_CB_HAS_N) && vs.v[l].n != 0u;
                    lib[jj] = live[jj] ? l : safe_lib;
                }
#pragma unroll
                for (int jj = 0; jj < MB_LKG; jj++) {
                    const WlView &w = vs.v[lib[jj]];
                    const uint32_t bin = (uint32_t)((uint64_t)key[jj] >> w.shiftE);
                    const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offE + bin);
                    lo[jj] = b2.a;
                    hi[jj] = b2.b;
                    twl[jj] = reinterpret_cast<const uint32_t *>(w.tailA);
                }
#pragma unroll
                for (int jj = 0; jj < MB_LKG; jj++) d[jj] = *reinterpret_cast<const U32x4 *>(twl[jj] + (lo[jj] >> 1));
#pragma unroll
                for (int jj = 0; jj < MB_LKG; jj++) {
                    const int j = j0 + jj;
                    const uint64_t i = tile * MB_TILE + (uint64_t)j * 256 + tid;
                    const WlView &w = vs.v[lib[jj]];
                    const uint32_t tail = key[jj] & ((1u << w.bitsB) - 1u);
                    const uint32_t p0 = lo[jj] & ~1u;
                    uint32_t found = CRGPU_MISS;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t pos = p0 + k;
                        const uint32_t t = (d[jj].w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                        if (pos >= lo[jj] && pos < hi[jj] && t == tail) found = pos;
                    }
                    if (hi[jj] > p0 + 8u && found == CRGPU_MISS && live[jj]) {  // a bin of more than 7 keys: rare
                        scan_u16_range<4>(w.tailA, p0 + 8u, hi[jj], [&](uint32_t t, uint32_t pos) {
                            if (t == tail) found = pos;
                        });
                    }
                    uint32_t rank = CRGPU_MISS;
                    if (live[jj] && found != CRGPU_MISS) rank = w.valA ? w.valA[found] : found;
                    bv[j] = 0xFFFFFFFFu;
                    loff[j] = 0;
                    if (i < n) idx_out[i] = rank;
                    if (rank != CRGPU_MISS) {
                        const uint32_t b = plan.lib_slot[lib[jj]] * plan.buckets_per_lib + (rank >> BIN_SHIFT);
                        loff[j] = (uint16_t)atomicAdd(&bcnt[b], 1u);
                        bv[j] = (b << 16) | (rank & (BIN_SIZE - 1u));
                    }
                }
            }
        }
        __syncthreads();
        // exclusive scan of the tile's bucket counts (MB_BPT consecutive buckets per thread) and one
        // global reservation per non-empty bucket
        {
            uint32_t c[MB_BPT], s = 0;
#pragma unroll
            for (int k = 0; k < MB_BPT; k++) {
                const uint32_t b = tid * MB_BPT + k;
                c[k] = b < NB ? bcnt[b] : 0u;
                s += c[k];
            }
            uint32_t tot;
            uint32_t run = block_excl_scan_256(s, lds, &tot);
            if (tid == 0) tile_hits = tot;
#pragma unroll
            for (int k = 0; k < MB_BPT; k++) {
                const uint32_t b = tid * MB_BPT + k;
                if (b < NB) {
                    bstart[b] = run;
                    run += c[k];
                    bcnt[b] = c[k] ? atomicAdd(&cursor[b * MB_CURSOR_STRIDE], c[k]) : 0u;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MB_ITEMS; j++)
            if (bv[j] != 0xFFFFFFFFu) {
                const uint32_t b = bv[j] >> 16;
                const uint32_t p = bstart[b] + loff[j];
                sval[p] = (uint16_t)(bv[j] & 0xFFFFu);
                sbkt[p] = (uint16_t)b;
            }
        __syncthreads();
        // contiguous runs per bucket: neighbouring p of one bucket go to neighbouring addresses
        const uint32_t hits = tile_hits;
        for (uint32_t p = tid; p < hits; p += 256) {
            const uint32_t b = sbkt[p];
            stage[(uint64_t)b * plan.cap + bcnt[b] + (p - bstart[b])] = sval[p];
        }
        __syncthreads();
    }
}

// ---- K1 with a hot-barcode table in LDS -------------------------------------------------------------
// k_match_binned is bound by the vector L1: two cache-line misses per read (index bin, bin contents) keep
// every CU's miss queue full (TCP_PENDING_STALL ~75 % of the cycles) while the ALUs idle.  Single-cell
// reads are anything but uniform over the whitelist, though: a few thousand cell barcodes carry ~90 % of
// them.  After a first small batch the HOT_CAP most frequent barcodes (by the valid histogram so far) go
// into an open-addressing table that every workgroup keeps in LDS (128 KB); a read is looked up there
// first and only the rest goes to the global tables.  Purely a cache: the result of every lookup is the
// same rank either way.
#define HOT_SLOTS 16384u   // 8192 buckets of two 8-byte entries (rank << 32 | key)
#define HOT_BUCKETS (HOT_SLOTS / 2u)
#define HOT_CAP 8192u      // load factor <= 0.5
#define HOT_EMPTY 0xFFFFFFFFFFFFFFFFull
// An entry lives in its home bucket or the next one (four slots); a barcode that finds all four taken is
// simply not cached.  The lookup is therefore two 16-byte LDS reads and four compares, no loop.
__device__ __forceinline__ uint32_t hot_hash(uint32_t key) { return (key * 0x9E3779B1u) >> 19; }  // 13 bits
__device__ __forceinline__ uint32_t hot_probe(const unsigned long long *s_hot, uint32_t key) {
    const uint32_t b0 = hot_hash(key), b1 = (b0 + 1u) & (HOT_BUCKETS - 1u);
    const uint4 x = *reinterpret_cast<const uint4 *>(s_hot + 2u * b0);
    const uint4 y = *reinterpret_cast<const uint4 *>(s_hot + 2u * b1);
    // an empty slot reads as key 0xFFFFFFFF with rank 0xFFFFFFFF == CRGPU_MISS: taking the minimum keeps a real rank
    uint32_t r = CRGPU_MISS;
    r = x.x == key && x.y < r ? x.y : r;
    r = x.z == key && x.w < r ? x.w : r;
    r = y.x == key && y.y < r ? y.y : r;
    r = y.z == key && y.w < r ? y.w : r;
    return r;
}
// the same, also telling which of the HOT_SLOTS entries answered (0xFFFF: none) -- the key of the hot-hit histogram
__device__ __forceinline__ uint32_t hot_probe_slot(const unsigned long long *s_hot, uint32_t key, uint32_t &slot) {
    const uint32_t b0 = hot_hash(key), b1 = (b0 + 1u) & (HOT_BUCKETS - 1u);
    const uint4 x = *reinterpret_cast<const uint4 *>(s_hot + 2u * b0);
    const uint4 y = *reinterpret_cast<const uint4 *>(s_hot + 2u * b1);
    uint32_t r = CRGPU_MISS, s = 0xFFFFu;
    if (x.x == key && x.y < r) { r = x.y; s = 2u * b0; }
    if (x.z == key && x.w < r) { r = x.w; s = 2u * b0 + 1u; }
    if (y.x == key && y.y < r) { r = y.y; s = 2u * b1; }
    if (y.z == key && y.w < r) { r = y.w; s = 2u * b1 + 1u; }
    slot = s;
    return r;
}
// monotone bucketing of a count >= 1 into 256 classes (8 per octave)
__device__ __forceinline__ uint32_t hot_class(uint32_t c) {
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    const uint32_t m = e >= 3u ? (c >> (e - 3u)) & 7u : (c << (3u - e)) & 7u;
    return e * 8u + m;
}
__global__ __launch_bounds__(256) void k_hot_hist(const uint32_t *__restrict__ valid, uint32_t n, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n; r += gridDim.x * 256u) {
        const uint32_t c = valid[r];
        if (c) atomicAdd(&h[hot_class(c)], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// sums[0] += reads of all barcodes, sums[1] += reads of the barcodes that enter the table (the caller estimates from them
// which share of the hits the table will NOT answer)
__global__ __launch_bounds__(256) void k_hot_build(const uint32_t *__restrict__ valid, const uint32_t *__restrict__ keys,
                                                   uint32_t n, const uint32_t *__restrict__ hist,
                                                   unsigned long long *__restrict__ image, unsigned long long *__restrict__ sums,
                                                   const bool sparse_keys) {
    // sparse_keys: keys[] is a list's own rank -> key array, 0xFFFFFFFF where the list has no key of that rank
    __shared__ uint32_t h[256];
    __shared__ uint32_t s_min;
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        // lowest class such that it and all higher classes together hold at most HOT_CAP barcodes
        uint32_t acc = 0, b = 256;
        while (b > 0 && acc + h[b - 1] <= HOT_CAP) {
            acc += h[b - 1];
            b--;
        }
        s_min = b;
    }
    __syncthreads();
    const uint32_t cmin = s_min;
    unsigned long long all = 0, hot = 0;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n; r += gridDim.x * 256u) {
        const uint32_t c = valid[r];
        all += c;
        if (!c || hot_class(c) < cmin) continue;
        const uint32_t key = keys[r];
        if (sparse_keys && key == 0xFFFFFFFFu) continue;  // (a count without a key of this list: a table the host wrote)
        const unsigned long long e = ((unsigned long long)r << 32) | key;
        const uint32_t b0 = hot_hash(key), b1 = (b0 + 1u) & (HOT_BUCKETS - 1u);
        const uint32_t slots[4] = {2u * b0, 2u * b0 + 1u, 2u * b1, 2u * b1 + 1u};
        for (int k = 0; k < 4; k++)
            if (atomicCAS(&image[slots[k]], HOT_EMPTY, e) == HOT_EMPTY) {
                hot += c;
                break;
            }
    }
    if (sums) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            all += __shfl_xor(all, d);
            hot += __shfl_xor(hot, d);
        }
        if ((threadIdx.x & 63u) == 0u) {
            atomicAdd(&sums[0], all);
            atomicAdd(&sums[1], hot);
        }
    }
}

// The lookups proper: no barrier inside the loop, every thread keeps LH_ITEMS reads in flight, 16 waves per
// CU.  (A single kernel that also staged the histogram had to stop all 16 waves of the one workgroup a CU
// can hold -- the table takes 128 KB -- at five barriers per tile and was latency bound.)
#define LH_THREADS 1024
#ifndef LH_ITEMS
#define LH_ITEMS 8
#endif
#ifndef LH_QITEMS
#define LH_QITEMS 4                  // item slots packed into the cold queue at a time
#endif
#define LH_QUEUE (64 * LH_QITEMS)    // entries of a wave's queue: every lane of every slot could be cold
// MODE 0: every hit goes through the staged histogram afterwards (k_stage_idx over idx).
// MODE 1 (LH_SPLIT): hits the table answers are written out as table slots (2 bytes per read) and counted by k_hist_hot_slots.
// MODE 2 (LH_COUNT): the table keeps 4-byte keys only (64 KB) and the other 64 KB of its LDS hold one counter per slot: a hit
//   the table answers bumps its slot's counter right here and takes its rank from the table image in global memory (128 KB,
//   cache resident); the counters are added to the VALID table when the workgroup ends.  No slot stream, no second kernel,
//   and the staged histogram sees the cold hits only.  A cold region that overflows counts its surplus hits by device atomics.
//   Opt-in (CRGPU_K1_MODE=count), measured SLOWER at 1 B reads: pass A 8.05 -> 9.05 ms on the 737 K list, 12.0 -> 12.65 on the
//   6.8 M one, cfg2 0.93 -> 1.13 -- the rank gather (64 different lines of the image per wave instruction) costs the lookup
//   more than the staging it saves; profiles/r03_count_stage_and_sort_ab.txt.
#define LH_FULL 0
#define LH_SPLIT 1
#define LH_COUNT 2
// 4-byte-key table: slot of `key` or 0xFFFF.  An empty slot holds 0xFFFFFFFF, so the all-T barcode is never cached (cold path).
__device__ __forceinline__ uint32_t hot_probe_keys(const uint32_t *s_key, uint32_t key) {
    const uint32_t b0 = hot_hash(key), b1 = (b0 + 1u) & (HOT_BUCKETS - 1u);
    const uint2 x = *reinterpret_cast<const uint2 *>(s_key + 2u * b0);
    const uint2 y = *reinterpret_cast<const uint2 *>(s_key + 2u * b1);
    uint32_t sl = 0xFFFFu;
    sl = y.y == key ? 2u * b1 + 1u : sl;
    sl = y.x == key ? 2u * b1 : sl;
    sl = x.y == key ? 2u * b0 + 1u : sl;
    sl = x.x == key ? 2u * b0 : sl;
    return key == 0xFFFFFFFFu ? 0xFFFFu : sl;
}
template <int MODE>
__global__ __launch_bounds__(LH_THREADS) void k_lookup_hot(const WlViewSet vs,
                                                           const unsigned long long *__restrict__ hot_image,
                                                           const uint32_t *__restrict__ cb, const uint8_t *__restrict__ flags,
                                                           uint64_t n, uint32_t *__restrict__ idx_out,
                                                           uint32_t *__restrict__ rec_i, uint32_t *__restrict__ rec_key,
                                                           uint8_t *__restrict__ rec_fl, uint32_t *__restrict__ rec_count,
                                                           uint32_t rec_cap, uint32_t rec_regions, uint32_t i_offset,
                                                           uint16_t *__restrict__ hot_slot_out, uint32_t *__restrict__ cold_rank,
                                                           uint32_t *__restrict__ cold_count, uint32_t cold_cap,
                                                           uint32_t cold_regions) {
    // hot_slot_out / cold_rank (both or neither): the histogram's inputs.  A hit answered by the LDS table is recorded as
    // the table slot that answered (2 bytes per read, 0xFFFF otherwise) and counted later in LDS by k_hist_hot_slots; a hit
    // found in the global tables is appended to this wave's region of cold_rank (no atomics, cursor in a register) and goes
    // through the staged histogram -- 15 % of the reads instead of all of them.
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_hot[];  // HOT_SLOTS, then the cold queues
    uint32_t *s_queue = reinterpret_cast<uint32_t *>(s_hot + HOT_SLOTS);          // LH_THREADS / 64 queues of LH_QUEUE
    constexpr bool SPLIT = MODE == LH_SPLIT, COUNT = MODE == LH_COUNT, COLD = MODE != LH_FULL;
    uint32_t *s_key = reinterpret_cast<uint32_t *>(s_hot), *s_cnt = s_key + HOT_SLOTS;  // COUNT: keys, then one counter per slot
    const uint32_t *__restrict__ hot_words = reinterpret_cast<const uint32_t *>(hot_image);  // [2 s] key, [2 s + 1] rank of slot s
    const uint32_t tid = threadIdx.x;
    const WlView &w = vs.v[0];
    const uint32_t *__restrict__ tw = reinterpret_cast<const uint32_t *>(w.tailA);
    const uint32_t tail_mask = (1u << w.bitsB) - 1u;
    if (COUNT) {
        for (uint32_t s = tid; s < HOT_SLOTS; s += LH_THREADS) {
            s_key[s] = hot_words[2u * s];
            s_cnt[s] = 0u;
        }
    } else {
        for (uint32_t s = tid; s < HOT_SLOTS; s += LH_THREADS) s_hot[s] = hot_image[s];
    }
    __syncthreads();
    const uint64_t chunk = (uint64_t)LH_THREADS * LH_ITEMS;
    // miss records (for K2): every wave appends to its own region, no atomics; the cursor lives in a register
    const uint32_t region = blockIdx.x * (LH_THREADS / 64) + (tid >> 6);
    uint32_t rec_cur = rec_i ? rec_count[region] : 0u;
    uint32_t cold_cur = COLD ? cold_count[region] : 0u;
    // the keys of the next chunk are requested before this chunk's table probes and global lookups
    uint32_t nkey[LH_ITEMS], nfl[LH_ITEMS];
#pragma unroll
    for (int j = 0; j < LH_ITEMS; j++) {
        const uint64_t i = (uint64_t)blockIdx.x * chunk + (uint64_t)j * LH_THREADS + tid;
        nkey[j] = i < n ? CR_LOAD_STREAM(&cb[i]) : 0u;
        nfl[j] = (i < n && flags) ? CR_LOAD_STREAM(&flags[i]) : 0u;
    }
    for (uint64_t base = (uint64_t)blockIdx.x * chunk; base < n; base += (uint64_t)gridDim.x * chunk) {
        uint32_t key[LH_ITEMS], rank[LH_ITEMS], cfl[LH_ITEMS], hslot[LH_ITEMS];
        bool todo[LH_ITEMS];  // live read that the LDS table did not answer
#pragma unroll
        for (int j = 0; j < LH_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * LH_THREADS + tid;
            key[j] = nkey[j];
            cfl[j] = nfl[j];
            todo[j] = i < n && !(nfl[j] & CRGPU_FLAG_CB_HAS_N) && (nfl[j] & CRGPU_FLAG_LIB_MASK) == vs.ulib;
            rank[j] = CRGPU_MISS;
        }
#pragma unroll
        for (int j = 0; j < LH_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)gridDim.x * chunk + (uint64_t)j * LH_THREADS + tid;
            nkey[j] = i < n ? CR_LOAD_STREAM(&cb[i]) : 0u;
            nfl[j] = (i < n && flags) ? CR_LOAD_STREAM(&flags[i]) : 0u;
        }
        if (COUNT) {
#pragma unroll
            for (int j = 0; j < LH_ITEMS; j++) {
                const uint32_t sl = hot_probe_keys(s_key, key[j]);
                hslot[j] = todo[j] ? sl : 0xFFFFu;
            }
#pragma unroll
            for (int j = 0; j < LH_ITEMS; j++)   // the ranks of the slots that answered: independent loads, issued together
                if (hslot[j] != 0xFFFFu) rank[j] = hot_words[2u * hslot[j] + 1u];
#pragma unroll
            for (int j = 0; j < LH_ITEMS; j++)
                if (hslot[j] != 0xFFFFu) {
                    todo[j] = false;
                    atomicAdd(&s_cnt[hslot[j]], 1u);
                }
        } else {
#pragma unroll
        for (int j = 0; j < LH_ITEMS; j++) {
            uint32_t sl = 0xFFFFu;
            const uint32_t r = SPLIT ? hot_probe_slot(s_hot, key[j], sl) : hot_probe(s_hot, key[j]);
            hslot[j] = 0xFFFFu;
            if (todo[j] && r != CRGPU_MISS) {
                rank[j] = r;
                todo[j] = false;
                hslot[j] = sl;
            }
        }
        }
        if (SPLIT) {
            // the table slots that answered this thread's LH_ITEMS reads, as one 16-byte store: the histogram does not care
            // which read a slot belongs to, so the stream is laid out by (chunk, thread), not by read
            static_assert(LH_ITEMS == 8, "eight 16-bit slots per 16-byte store");
            const uint64_t chunk_id = base / chunk;
            cr_u32x4 pk;
            pk.x = hslot[0] | (hslot[1] << 16);
            pk.y = hslot[2] | (hslot[3] << 16);
            pk.z = hslot[4] | (hslot[5] << 16);
            pk.w = hslot[6] | (hslot[7] << 16);
            CR_STORE_STREAM(pk, reinterpret_cast<cr_u32x4 *>(hot_slot_out) + chunk_id * LH_THREADS + tid);
        }
        // The rest (~20 % of the reads: ambient barcodes, the smaller cells, sequencing errors) goes to the global
        // tables.  Executing that path once per item slot costs every lane its instructions although only a fifth
        // of them need it, so each wave first packs the keys of its cold lanes densely into a private LDS queue
        // (a wave runs in lockstep and its LDS operations complete in order: no barrier), looks them up 64 at a
        // time, and hands the ranks back through the same queue.
        uint32_t *q = s_queue + (tid >> 6) * LH_QUEUE;
        const uint32_t lane = tid & 63u;
#pragma unroll
        for (int h = 0; h < LH_ITEMS; h += LH_QITEMS) {
            uint32_t dj[LH_QITEMS], tot = 0;
#pragma unroll
            for (int jj = 0; jj < LH_QITEMS; jj++) {
                const unsigned long long m = __ballot(todo[h + jj]);
                dj[jj] = tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                tot += (uint32_t)__popcll(m);
                if (todo[h + jj]) q[dj[jj]] = key[h + jj];
            }
            __builtin_amdgcn_wave_barrier();  // keep the compiler from moving queue reads above these writes
            for (uint32_t r0 = 0; r0 < tot; r0 += 64) {  // uniform: tot comes from ballots
                const uint32_t dpos = r0 + lane;
                uint32_t found = CRGPU_MISS;
                if (dpos < tot) {
                    const uint32_t k = q[dpos];
                    const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offE + (uint32_t)((uint64_t)k >> w.shiftE));
                    const uint32_t lo = b2.a, hi = b2.b;
                    if (hi > lo) {
                        const U32x4 d = *reinterpret_cast<const U32x4 *>(tw + (lo >> 1));
                        const uint32_t tail = k & tail_mask;
                        const uint32_t p0 = lo & ~1u;
#pragma unroll
                        for (uint32_t e = 0; e < 8; e++) {
                            const uint32_t pos = p0 + e;
                            const uint32_t t = (d.w[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu;
                            if (pos >= lo && pos < hi && t == tail) found = pos;
                        }
                        if (hi > p0 + 8u && found == CRGPU_MISS)  // a bin of more than 7 keys: rare
                            scan_u16_range<4>(w.tailA, p0 + 8u, hi, [&](uint32_t t, uint32_t pos) {
                                if (t == tail) found = pos;
                            });
                    }
                    if (w.valA && found != CRGPU_MISS) found = w.valA[found];  // (translated / partial list: position -> rank)
                    q[dpos] = found;
                }
                if (COLD) {  // the hits of this batch of 64 cold lookups, densely, behind the wave's earlier ones
                    const unsigned long long cm = __ballot(found != CRGPU_MISS);
                    const uint32_t pos = cold_cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                    if (found != CRGPU_MISS && pos < cold_cap) cold_rank[(uint64_t)region * cold_cap + pos] = found;
                    if (COUNT && found != CRGPU_MISS && pos >= cold_cap) atomicAdd(&w.valid[found], 1u);  // region full: counted here
                    cold_cur += (uint32_t)__popcll(cm);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jj = 0; jj < LH_QITEMS; jj++)
                if (todo[h + jj]) rank[h + jj] = q[dj[jj]];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < LH_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * LH_THREADS + tid;
            if (i < n) CR_STORE_STREAM(rank[j], &idx_out[i]);
            if (rec_i) {
                const bool miss = i < n && rank[j] == CRGPU_MISS;
                const unsigned long long mm = __ballot(miss);
                const uint32_t pos = rec_cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                if (miss && pos < rec_cap) {
                    const uint64_t o = (uint64_t)region * rec_cap + pos;
                    rec_i[o] = i_offset + (uint32_t)i;
                    rec_key[o] = key[j];
                    rec_fl[o] = (uint8_t)cfl[j];
                }
                rec_cur += (uint32_t)__popcll(mm);
            }
        }
    }
    if (rec_i && (tid & 63u) == 0u) {
        rec_count[region] = rec_cur;
        if (rec_cur > rec_cap) atomicOr(&rec_count[rec_regions], 1u);  // overflow: K2 falls back to scanning idx
    }
    if (SPLIT && (tid & 63u) == 0u) {
        cold_count[region] = cold_cur;
        if (cold_cur > cold_cap) atomicOr(&cold_count[cold_regions], 1u);  // overflow: this round is counted by k_hist_ranks_atomic
    }
    if (COUNT) {
        if ((tid & 63u) == 0u) cold_count[region] = cold_cur < cold_cap ? cold_cur : cold_cap;
        __syncthreads();
        for (uint32_t s = tid; s < HOT_SLOTS; s += LH_THREADS) {
            const uint32_t c = s_cnt[s];
            if (c) atomicAdd(&w.valid[hot_words[2u * s + 1u]], c);
        }
    }
}

// Histogram of the hits the LDS table answered: 16384 counters (one per table slot) in LDS, the slot stream read with
// 16-byte loads; at the end every non-zero counter is added to the valid count of the slot's barcode.
// skip_if (nullable): do nothing when *skip_if != 0 (a cold region overflowed: the round is counted another way).
#define HH_THREADS 1024
__global__ __launch_bounds__(HH_THREADS) void k_hist_hot_slots(const uint16_t *__restrict__ slots, uint64_t groups,
                                                               const unsigned long long *__restrict__ hot_image,
                                                               uint32_t *__restrict__ valid, const uint32_t *__restrict__ skip_if) {
    extern __shared__ uint32_t s_cnt[];  // HOT_SLOTS
    if (skip_if && *skip_if) return;
    for (uint32_t s = threadIdx.x; s < HOT_SLOTS; s += HH_THREADS) s_cnt[s] = 0;
    __syncthreads();
    // groups of eight slots (16 bytes), one per (chunk, thread) of k_lookup_hot; the stream is a 16-byte aligned pool block
    const uint4 *__restrict__ g = reinterpret_cast<const uint4 *>(slots);
    for (uint64_t k = (uint64_t)blockIdx.x * HH_THREADS + threadIdx.x; k < groups; k += (uint64_t)gridDim.x * HH_THREADS) {
        const uint4 v = g[k];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t s = (w[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
            if (s != 0xFFFFu) atomicAdd(&s_cnt[s], 1u);
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < HOT_SLOTS; s += HH_THREADS) {
        const uint32_t c = s_cnt[s];
        if (c) atomicAdd(&valid[(uint32_t)(hot_image[s] >> 32)], c);
    }
}

// plain device atomics on a table of counts: the fallback of the fallbacks (run_if nullable: only when *run_if != 0)
__global__ __launch_bounds__(256) void k_hist_ranks_atomic(const uint32_t *__restrict__ idx, uint64_t n, uint32_t *__restrict__ table,
                                                          const uint32_t *__restrict__ run_if) {
    if (run_if && *run_if == 0u) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t r = idx[i];
        if (r != CRGPU_MISS) atomicAdd(&table[r], 1u);
    }
}

// Histogram staging from the ranks in idx for at most 31 buckets (whitelists of up to ~1 M barcodes, one
// library): the same tile -> per-bucket runs -> staging regions as k_match_binned, but the position of a
// rank inside its (tile, bucket) comes from the sort's ballot multisplit instead of 4096 returning LDS
// atomics on two dozen counters (those serialised on their addresses).  Bucket 31 collects the misses.
#define SI_BITS 5
#ifndef SI_ITEMS
#define SI_ITEMS 32  // 8192 ranks per tile: 3.97 -> 3.45 ms per 400 M reads against 4096 (fewer barriers and cursor atomics)
#endif
#define SI_TILE (256 * SI_ITEMS)
#define SI_MISS_BUCKET 31u
__global__ __launch_bounds__(256) void k_stage_idx(const BinPlan plan, const uint32_t *__restrict__ idx, uint64_t n,
                                                   uint16_t *__restrict__ stage, uint32_t *__restrict__ cursor,
                                                   const uint32_t *__restrict__ skip_if = nullptr) {
    __shared__ uint32_t wcount[4][32];  // per-wave bucket counts -> tile-local start of (wave, bucket)
    __shared__ uint32_t gbase[32];      // global base of the tile's run of each bucket
    __shared__ uint32_t tstart[32];     // tile-local start of each bucket
    __shared__ uint16_t sval[SI_TILE];
    __shared__ uint8_t sbkt[SI_TILE];
    __shared__ uint32_t tile_hits;
    if (skip_if && *skip_if) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t n_tiles = (n + SI_TILE - 1) / SI_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (tid < 128) wcount[tid >> 5][tid & 31u] = 0;
        __syncthreads();
        uint32_t dr[SI_ITEMS];  // (bucket << 16) | rank inside (wave, bucket)
        uint16_t v[SI_ITEMS];
        const uint64_t wave_base = tile * SI_TILE + (uint64_t)wave * (64 * SI_ITEMS) + lane;  // wave-major order
        uint32_t r[SI_ITEMS];
#pragma unroll
        for (int j = 0; j < SI_ITEMS; j++) {
            const uint64_t i = wave_base + (uint64_t)j * 64;
            r[j] = i < n ? idx[i] : CRGPU_MISS;
        }
#pragma unroll
        for (int j = 0; j < SI_ITEMS; j++) {
            const uint32_t b = r[j] != CRGPU_MISS ? (r[j] >> BIN_SHIFT) : SI_MISS_BUCKET;
            v[j] = (uint16_t)(r[j] & (BIN_SIZE - 1u));
            dr[j] = (b << 16) | wave_multisplit_rank<SI_BITS, true>(b, true, wcount[wave]);
        }
        __syncthreads();
        if (tid < 32) {
            // one lane per bucket: totals, exclusive scan over the hit buckets, one global reservation each
            uint32_t c[4], tot = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                c[w] = wcount[w][tid];
                tot += c[w];
            }
            if (tid == SI_MISS_BUCKET) tot = 0;
            uint32_t x = tot;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (tid >= (uint32_t)d) x += y;
            }
            uint32_t run = x - tot;
            tstart[tid] = run;
            if (tid == 30) tile_hits = x;  // buckets 0..30 are the hit buckets
            gbase[tid] = tot ? atomicAdd(&cursor[tid * MB_CURSOR_STRIDE], tot) : 0u;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                wcount[w][tid] = run;
                run += c[w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SI_ITEMS; j++) {
            const uint32_t b = dr[j] >> 16;
            if (b != SI_MISS_BUCKET) {
                const uint32_t p = wcount[wave][b] + (dr[j] & 0xFFFFu);
                sval[p] = v[j];
                sbkt[p] = (uint8_t)b;
            }
        }
        __syncthreads();
        const uint32_t hits = tile_hits;
        for (uint32_t p = tid; p < hits; p += 256) {
            const uint32_t b = sbkt[p];
            stage[(uint64_t)b * plan.cap + gbase[b] + (p - tstart[b])] = sval[p];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void k_hist_buckets(const WlViewSet vs, const BinPlan plan, uint32_t slices,
                                                      const uint16_t *__restrict__ stage,
                                                      const uint32_t *__restrict__ cursor) {
    extern __shared__ uint32_t cnt[];  // BIN_SIZE counters
    const uint32_t b = blockIdx.x / slices, s = blockIdx.x % slices;
    const uint32_t total = cursor[b * MB_CURSOR_STRIDE];
    // slice boundaries in units of 8 entries (16-byte loads); the last slice takes the ragged tail
    const uint32_t groups = (total + 7u) / 8u;
    const uint32_t g_lo = (uint32_t)((uint64_t)groups * s / slices), g_hi = (uint32_t)((uint64_t)groups * (s + 1) / slices);
    if (g_lo >= g_hi) return;
    for (uint32_t c = threadIdx.x; c < BIN_SIZE; c += 512) cnt[c] = 0;
    __syncthreads();
    const uint16_t *src = stage + (uint64_t)b * plan.cap;
    for (uint32_t g = g_lo + threadIdx.x; g < g_hi; g += 512) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + (uint64_t)g * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const uint32_t valid = total - g * 8u;  // entries of this group that exist (>= 8 except in the tail)
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k < valid) atomicAdd(&cnt[(w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu], 1u);
    }
    __syncthreads();
    // bucket -> (library slot, rank range)
    const uint32_t slot = b / plan.buckets_per_lib, sub = b % plan.buckets_per_lib;
    uint32_t lib = 0;
    for (uint32_t l = 0; l < CRGPU_MAX_LIB; l++)
        if (plan.lib_slot[l] == slot) lib = l;
    uint32_t *dst = vs.v[lib].valid + (uint64_t)sub * BIN_SIZE;
    const uint32_t limit = vs.n_canon - sub * BIN_SIZE < BIN_SIZE ? vs.n_canon - sub * BIN_SIZE : BIN_SIZE;
    for (uint32_t c = threadIdx.x; c < limit; c += 512) {
        const uint32_t x = cnt[c];
        if (x) atomicAdd(&dst[c], x);
    }
}

#ifndef MB_STAGE_BUDGET
#define MB_STAGE_BUDGET (8ull << 30)  // bytes of u16 staging per super-batch: 6 launch rounds per 1 B reads (2 GB = 22 rounds cost 1.1 ms more in ramp-up / table loads)
#endif

// Histogram of a dense array of canonical ranks (CRGPU_MISS entries are skipped) into v[0].valid of `vs`, with the
// staging kernels of K1: k_stage_idx (needs plan.n_buckets <= SI_MISS_BUCKET) + k_hist_buckets.
static int hist_from_idx(crgpu_ctx *ctx, const WlViewSet &vs, BinPlan plan, const uint32_t *d_idx, uint64_t n) {
    if (n == 0) return CRGPU_OK;
    uint64_t sb = MB_STAGE_BUDGET / 2 / plan.n_buckets;
    sb = sb / MB_TILE * MB_TILE;
    if (sb < MB_TILE) sb = MB_TILE;
    if (sb > n) sb = (n + MB_TILE - 1) / MB_TILE * MB_TILE;
    plan.cap = sb;
    uint16_t *d_stage = nullptr;
    uint32_t *d_cursor = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_stage, (uint64_t)plan.n_buckets * plan.cap * sizeof(uint16_t)));
    int rc = cr_pool_alloc(ctx, (void **)&d_cursor, plan.n_buckets * MB_CURSOR_STRIDE * sizeof(uint32_t));
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d_stage);
        return rc;
    }
    uint32_t slices = 512 / plan.n_buckets;
    if (slices < 1) slices = 1;
    cr_allow_lds(ctx, (const void *)k_hist_buckets, BIN_SIZE * 4);
    hipError_t e = hipSuccess;
    for (uint64_t off = 0; off < n && e == hipSuccess; off += sb) {
        const uint64_t m = n - off < sb ? n - off : sb;
        e = hipMemsetAsync(d_cursor, 0, plan.n_buckets * MB_CURSOR_STRIDE * sizeof(uint32_t), ctx->stream);
        hipLaunchKernelGGL(k_stage_idx, dim3(cr_grid((m + SI_TILE - 1) / SI_TILE, 1, 256u * 6u)), dim3(256), 0, ctx->stream, plan,
                           d_idx + off, m, d_stage, d_cursor);
        hipLaunchKernelGGL(k_hist_buckets, dim3(plan.n_buckets * slices), dim3(512), BIN_SIZE * 4, ctx->stream, vs, plan, slices,
                           d_stage, d_cursor);
        if (e == hipSuccess) e = hipGetLastError();
    }
    cr_pool_free(ctx, d_stage);
    cr_pool_free(ctx, d_cursor);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "histogram of ranks: %s", hipGetErrorString(e));
    return CRGPU_OK;
}

extern "C" int crgpu_match_and_count_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_flags, uint64_t n,
                                         uint32_t *d_idx_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_match_and_count: no whitelist set");
    CR_REQUIRE(ctx, ctx->n_segments == 0, CRGPU_ESTATE,
               "crgpu_match_and_count: this context holds a segmented barcode space; run the barcode stage on the segment contexts");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_cb && d_idx_out, CRGPU_EINVAL, "crgpu_match_and_count: NULL buffer");
    // records of an earlier call about these buffers are stale now; the other sets stay (other libraries of the well)
    for (MissRecords &old : ctx->recs)
        if (old.valid && (old.d_cb == d_cb || old.d_idx == d_idx_out)) cr_drop_miss_records(ctx, old);
    cr_dense_drop(ctx);  // the VALID table changes
    if (!d_flags) {
        // NULL flags mean "no barcode holds an N".  A pack call that ran without a flags array and met an N left a mark:
        // such a barcode would be looked up with its N read as A and could count as a whitelist hit, which the
        // reference's check_and_update never does (whitelist.rs:494-517)
        uint32_t seen = 0;
        CR_TRY(crgpu_memcpy_d2h(ctx, &seen, ctx->d_scalars + CR_SCALAR_N_WITHOUT_FLAGS, sizeof(seen)));
        if (seen) {
            CR_HIP(ctx, hipMemsetAsync(ctx->d_scalars + CR_SCALAR_N_WITHOUT_FLAGS, 0, sizeof(uint32_t), ctx->stream));
            return cr_fail(ctx, CRGPU_EINVAL,
                           "crgpu_match_and_count: barcodes containing N were packed without a flags array; pass flags to "
                           "crgpu_pack*_dev and to this call (CRGPU_FLAG_CB_HAS_N)");
        }
    }
    int ulib = -1;
    CR_TRY(pick_uniform_lib(ctx, d_flags, n, &ulib));
    const bool uniform = ulib >= 0;
    WlViewSet vs;
    CR_TRY(make_view_set(ctx, vs, ulib));
    const WlTables &uw = ctx->wl[uniform ? ulib : 0];  // the call's library (one-library calls only)

    BinPlan plan;
    uint32_t n_slots = 0;
    // a one-library call stages into a single slot, which k_hist_buckets adds to v[0] (= that library's tables)
    for (int l = 0; l < CRGPU_MAX_LIB; l++)
        plan.lib_slot[l] = uniform ? (l == 0 ? n_slots++ : 0xFFFFFFFFu) : (ctx->wl[l].set ? n_slots++ : 0xFFFFFFFFu);
    plan.buckets_per_lib = (ctx->n_canon + BIN_SIZE - 1) / BIN_SIZE;
    plan.n_buckets = plan.buckets_per_lib * n_slots;

    if (plan.n_buckets > MB_MAX_BUCKETS) {
        // very large whitelist x many libraries: plain device atomics
        CrTimer t(ctx, CRGPU_T_MATCH, n);
        const dim3 grid(cr_grid(n, 256)), block(256);
        if (uniform)
            hipLaunchKernelGGL(k_match<true>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, n, d_idx_out);
        else
            hipLaunchKernelGGL(k_match<false>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, n, d_idx_out);
        CR_HIP(ctx, hipGetLastError());
        return CRGPU_OK;
    }

    // super-batches sized so that every bucket could take ALL reads of the batch
    uint64_t sb = MB_STAGE_BUDGET / 2 / plan.n_buckets;
    sb = sb / MB_TILE * MB_TILE;
    if (sb < MB_TILE) sb = MB_TILE;
    if (sb > n) sb = (n + MB_TILE - 1) / MB_TILE * MB_TILE;
    plan.cap = sb;  // multiple of 4096 -> every bucket region is 16-byte aligned
    uint16_t *d_stage = nullptr;
    uint32_t *d_cursor = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_stage, (uint64_t)plan.n_buckets * plan.cap * sizeof(uint16_t)));
    int rc = cr_pool_alloc(ctx, (void **)&d_cursor, plan.n_buckets * MB_CURSOR_STRIDE * sizeof(uint32_t));
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d_stage);
        return rc;
    }
    // slices per bucket: ~2 workgroups per CU in total (128 KB of LDS each -> one resident per CU)
    uint32_t slices = 512 / plan.n_buckets;
    if (slices < 1) slices = 1;
    cr_allow_lds(ctx, (const void *)k_hist_buckets, BIN_SIZE * 4);
    // hot-barcode table: one library without translation, and enough reads to pay for the sampling batch
    // (CRGPU_HOT_MIN_READS lowers the threshold so that the parity tests can drive this path with small inputs)
    uint64_t hot_min = 16ull << 20;
    if (const char *env = getenv("CRGPU_HOT_MIN_READS")) hot_min = strtoull(env, nullptr, 10);
    // (a translated or partial list brings its own rank -> key array; a rank without a key of this list never has a count)
    const uint32_t *d_hot_keys = uw.d_valA ? uw.d_key_of_rank : ctx->d_canon_keys;
    const bool use_hot = uniform && n >= hot_min && n >= 4ull * MB_TILE && d_hot_keys != nullptr &&
                         !(uw.d_valA && getenv("CRGPU_HOT_PLAIN_ONLY"));  // (A/B switch: round 2's rule)
    uint64_t first = 0;  // reads of the sampling batch (a multiple of MB_TILE)
    if (use_hot) {
        first = n / 4 < (4ull << 20) ? n / 4 : (4ull << 20);
        first = first / MB_TILE * MB_TILE;
    }
    const size_t hot_lds = HOT_SLOTS * sizeof(unsigned long long);                       // table image
    const size_t lookup_lds = hot_lds + (LH_THREADS / 64) * LH_QUEUE * sizeof(uint32_t);  // + per-wave cold queues
    if (use_hot && !ctx->d_hot_image) {
        if (hipMalloc((void **)&ctx->d_hot_image, hot_lds + 256 * sizeof(uint32_t) + 2 * sizeof(unsigned long long)) != hipSuccess) {
            cr_pool_free(ctx, d_stage);
            cr_pool_free(ctx, d_cursor);
            return cr_fail(ctx, CRGPU_ENOMEM, "hipMalloc hot table failed");
        }
    }
    // miss records for K2: one region per wave of the lookup kernel (its grid is pinned to 256 workgroups)
    uint32_t rec_slot = CR_REC_SETS;
    for (uint32_t k = 0; k < CR_REC_SETS && rec_slot == CR_REC_SETS; k++)
        if (!ctx->recs[k].valid) rec_slot = k;
    if (rec_slot == CR_REC_SETS) {
        rec_slot = ctx->rec_next;
        ctx->rec_next = (ctx->rec_next + 1u) % CR_REC_SETS;
    }
    MissRecords &rec = ctx->recs[rec_slot];
    cr_drop_miss_records(ctx, rec);
    if (use_hot && ctx->trust_buffers && n < 0xFFFFFFFFull && !getenv("CRGPU_NO_MISS_RECORDS")) {
        const uint64_t H = n - first;
        const uint64_t launches = (H + sb - 1) / sb + 1;
        rec.regions = 256u * (LH_THREADS / 64);
        rec.cap = (uint32_t)(H / rec.regions / 3 + 2048 + 512 * launches);  // a third of a wave's reads may miss
        if (const char *env = getenv("CRGPU_MISS_RECORD_CAP")) rec.cap = (uint32_t)strtoul(env, nullptr, 10);  // tests: force the overflow fallback
        const uint64_t C = (uint64_t)rec.regions * rec.cap;
        int rr = cr_pool_alloc(ctx, (void **)&rec.d_i, C * sizeof(uint32_t));
        if (rr == CRGPU_OK) rr = cr_pool_alloc(ctx, (void **)&rec.d_key, C * sizeof(uint32_t));
        if (rr == CRGPU_OK) rr = cr_pool_alloc(ctx, (void **)&rec.d_fl, C);
        if (rr == CRGPU_OK) rr = cr_pool_alloc(ctx, (void **)&rec.d_count, (rec.regions + 1) * sizeof(uint32_t));
        if (rr == CRGPU_OK &&
            hipMemsetAsync(rec.d_count, 0, (rec.regions + 1) * sizeof(uint32_t), ctx->stream) == hipSuccess) {
            rec.valid = true;
            rec.d_cb = d_cb;
            rec.d_flags = d_flags;
            rec.d_idx = d_idx_out;
            rec.n = n;
            rec.first = first;
            rec.ulib = ulib;
        } else {
            cr_drop_miss_records(ctx, rec);  // not fatal: K2 scans idx as before
        }
    }
    if (use_hot) {
        cr_allow_lds(ctx, (const void *)k_lookup_hot<LH_FULL>, lookup_lds);
        cr_allow_lds(ctx, (const void *)k_lookup_hot<LH_SPLIT>, lookup_lds);
        cr_allow_lds(ctx, (const void *)k_lookup_hot<LH_COUNT>, lookup_lds);
    }
    hipError_t e = hipSuccess;
    bool hot_ready = false;
    // Split histogram of the table rounds (k_lookup_hot's comment): hits the table answers are counted per table slot in
    // LDS, the others are appended to per-wave regions and staged -- a sixth of the entries the staging used to see, so the
    // rounds can be that much longer.  Sized after the sampling batch from the share of the reads the table's barcodes
    // carry; a region that overflows makes its round fall back to device atomics (cold_count[regions] != 0).
    const uint32_t cold_regions = 256u * (LH_THREADS / 64);
    uint16_t *d_hot_slot = nullptr;
    uint32_t *d_cold = nullptr, *d_cold_count = nullptr;
    uint64_t hot_round = 0;   // reads per table round (0: the split histogram is off)
    uint32_t cold_cap = 0;
    // CRGPU_K1_MODE=count: table hits counted inside the lookup kernel (LH_COUNT); =split / =full: the other two
    const char *k1_mode = getenv("CRGPU_K1_MODE");
    const bool count_mode = k1_mode && strcmp(k1_mode, "count") == 0;
    struct ColdRelease {
        crgpu_ctx *c;
        uint16_t *&a;
        uint32_t *&b, *&d;
        ~ColdRelease() {
            cr_pool_free(c, a);
            cr_pool_free(c, b);
            cr_pool_free(c, d);
        }
    } cold_release{ctx, d_hot_slot, d_cold, d_cold_count};
    for (uint64_t off = 0; off < n && e == hipSuccess;) {
        uint64_t m = n - off < sb ? n - off : sb;
        if (use_hot && off == 0 && m > first) m = first;
        if (hot_ready && hot_round) m = n - off < hot_round ? n - off : hot_round;
        {
            CrTimer t(ctx, CRGPU_T_MATCH, m);
            e = hipMemsetAsync(d_cursor, 0, plan.n_buckets * MB_CURSOR_STRIDE * sizeof(uint32_t), ctx->stream);
            if (hot_ready && hot_round) {
                ctx->k1_split_rounds++;
                const uint64_t C = (uint64_t)cold_regions * cold_cap;
                const uint32_t *d_over = d_cold_count + cold_regions;  // != 0 after the lookup: a cold region overflowed
                if (e == hipSuccess) e = hipMemsetAsync(d_cold_count, 0, (cold_regions + 1) * sizeof(uint32_t), ctx->stream);
                if (e == hipSuccess) e = hipMemsetAsync(d_cold, 0xFF, C * sizeof(uint32_t), ctx->stream);
                if (count_mode) {
                    d_over = nullptr;  // nothing to fall back to: a full region counts its surplus hits itself
                    hipLaunchKernelGGL(k_lookup_hot<LH_COUNT>, dim3(cr_grid((m + MB_TILE - 1) / MB_TILE, 1, 256u)), dim3(LH_THREADS),
                                       lookup_lds, ctx->stream, vs, ctx->d_hot_image, d_cb + off, d_flags ? d_flags + off : nullptr, m,
                                       d_idx_out + off, rec.valid ? rec.d_i : nullptr, rec.d_key, rec.d_fl, rec.d_count, rec.cap,
                                       rec.regions, (uint32_t)off, (uint16_t *)nullptr, d_cold, d_cold_count, cold_cap, cold_regions);
                } else {
                hipLaunchKernelGGL(k_lookup_hot<LH_SPLIT>, dim3(cr_grid((m + MB_TILE - 1) / MB_TILE, 1, 256u)), dim3(LH_THREADS), lookup_lds,
                                   ctx->stream, vs, ctx->d_hot_image, d_cb + off, d_flags ? d_flags + off : nullptr, m,
                                   d_idx_out + off, rec.valid ? rec.d_i : nullptr, rec.d_key, rec.d_fl, rec.d_count, rec.cap,
                                   rec.regions, (uint32_t)off, d_hot_slot, d_cold, d_cold_count, cold_cap, cold_regions);
                const uint64_t lh_chunk = (uint64_t)LH_THREADS * LH_ITEMS;
                hipLaunchKernelGGL(k_hist_hot_slots, dim3(128), dim3(HH_THREADS), HOT_SLOTS * sizeof(uint32_t), ctx->stream, d_hot_slot,
                                   (m + lh_chunk - 1) / lh_chunk * LH_THREADS, ctx->d_hot_image, uw.d_valid, d_over);
                }
                if (plan.n_buckets <= SI_MISS_BUCKET)
                    hipLaunchKernelGGL(k_stage_idx, dim3(cr_grid((C + SI_TILE - 1) / SI_TILE, 1, 256u * 6u)), dim3(256), 0, ctx->stream,
                                       plan, d_cold, C, d_stage, d_cursor, d_over);
                else
                    hipLaunchKernelGGL((k_match_binned<true, true>), dim3(cr_grid((C + MB_ITEMS - 1) / MB_ITEMS, 256, 256u * 6u)),
                                       dim3(256), 0, ctx->stream, vs, plan, (const uint32_t *)nullptr, (const uint8_t *)nullptr, C, d_cold,
                                       d_stage, d_cursor, d_over);
                if (!count_mode)
                    hipLaunchKernelGGL(k_hist_ranks_atomic, dim3(cr_grid(m, 256)), dim3(256), 0, ctx->stream, d_idx_out + off, m, uw.d_valid,
                                       d_over);
                if (getenv("CRGPU_K1_DEBUG")) {
                    std::vector<uint32_t> cc(cold_regions + 1);
                    (void)hipMemcpyAsync(cc.data(), d_cold_count, cc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
                    (void)hipStreamSynchronize(ctx->stream);
                    uint32_t mx = 0;
                    uint64_t tot = 0;
                    for (uint32_t r = 0; r < cold_regions; r++) {
                        mx = std::max(mx, cc[r]);
                        tot += cc[r];
                    }
                    fprintf(stderr, "K1 round off=%llu m=%llu cap=%u max_region=%u total_cold=%llu overflow=%u\n",
                            (unsigned long long)off, (unsigned long long)m, cold_cap, mx, (unsigned long long)tot, cc[cold_regions]);
                }
            } else if (hot_ready) {
                hipLaunchKernelGGL(k_lookup_hot<LH_FULL>, dim3(cr_grid((m + MB_TILE - 1) / MB_TILE, 1, 256u)), dim3(LH_THREADS), lookup_lds,
                                   ctx->stream, vs, ctx->d_hot_image, d_cb + off, d_flags ? d_flags + off : nullptr, m,
                                   d_idx_out + off, rec.valid ? rec.d_i : nullptr, rec.d_key, rec.d_fl, rec.d_count, rec.cap,
                                   rec.regions, (uint32_t)off, (uint16_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, 0u);
                if (plan.n_buckets <= SI_MISS_BUCKET)
                    hipLaunchKernelGGL(k_stage_idx, dim3(cr_grid((m + SI_TILE - 1) / SI_TILE, 1, 256u * 6u)), dim3(256), 0,
                                       ctx->stream, plan, d_idx_out + off, m, d_stage, d_cursor);
                else
                    hipLaunchKernelGGL((k_match_binned<true, true>), dim3(cr_grid((m + MB_ITEMS - 1) / MB_ITEMS, 256, 256u * 6u)),
                                       dim3(256), 0, ctx->stream, vs, plan, d_cb + off, d_flags ? d_flags + off : nullptr, m,
                                       d_idx_out + off, d_stage, d_cursor);
            } else {
                const dim3 grid(cr_grid((m + MB_ITEMS - 1) / MB_ITEMS, 256, 256u * 6u)), block(256);
                if (uniform)
                    hipLaunchKernelGGL(k_match_binned<true>, grid, block, 0, ctx->stream, vs, plan, d_cb + off,
                                       d_flags ? d_flags + off : nullptr, m, d_idx_out + off, d_stage, d_cursor);
                else
                    hipLaunchKernelGGL(k_match_binned<false>, grid, block, 0, ctx->stream, vs, plan, d_cb + off,
                                       d_flags ? d_flags + off : nullptr, m, d_idx_out + off, d_stage, d_cursor);
            }
            hipLaunchKernelGGL(k_hist_buckets, dim3(plan.n_buckets * slices), dim3(512), BIN_SIZE * 4, ctx->stream, vs, plan,
                               slices, d_stage, d_cursor);
            if (use_hot && off == 0) {
                // the most frequent barcodes so far -> table image (the histogram includes earlier batches of the caller)
                uint32_t *d_hh = reinterpret_cast<uint32_t *>(ctx->d_hot_image + HOT_SLOTS);
                if (e == hipSuccess) e = hipMemsetAsync(ctx->d_hot_image, 0xFF, hot_lds, ctx->stream);
                if (e == hipSuccess) e = hipMemsetAsync(d_hh, 0, 256 * sizeof(uint32_t), ctx->stream);
                hipLaunchKernelGGL(k_hot_hist, dim3(128), dim3(256), 0, ctx->stream, uw.d_valid, ctx->n_canon, d_hh);
                unsigned long long *d_sums = reinterpret_cast<unsigned long long *>(d_hh + 256);
                if (e == hipSuccess) e = hipMemsetAsync(d_sums, 0, 2 * sizeof(unsigned long long), ctx->stream);
                hipLaunchKernelGGL(k_hot_build, dim3(128), dim3(256), 0, ctx->stream, uw.d_valid, d_hot_keys,
                                   ctx->n_canon, d_hh, ctx->d_hot_image, d_sums, uw.d_valA != nullptr);
                hot_ready = true;
                // share of the hits that the table will not answer (one read-back per call) -> size of the cold regions
                unsigned long long sums[2] = {0, 0};
                // Where it pays (measured, profiles/r02_k1_split_ab.txt): whitelists beyond the 31 buckets of k_stage_idx (the
                // 6.8 M-entry list: 208 buckets, 49 staging rounds) -4 ms per 1 B reads; on the 737 K list the 2 bytes per read
                // of the slot stream cost the lookup what the shorter staging saves (8.05 against 8.07 ms) and 100 M-read calls
                // lose 5 %.  CRGPU_K1_SPLIT=1 / 0 forces it on / off.
                const char *split_env = getenv("CRGPU_K1_SPLIT");
                const bool want_split = count_mode || (split_env ? split_env[0] != '0' : plan.n_buckets > SI_MISS_BUCKET);
                if (e == hipSuccess && want_split && !getenv("CRGPU_K1_FULL_STAGING") &&
                    hipMemcpyAsync(sums, d_sums, sizeof(sums), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess && sums[0] > 0) {
                    const double cold_frac = 1.0 - (double)sums[1] / (double)sums[0];
                    // head room for the spread between the waves; in steps of 0.05 (and the round in steps of 32 Mi reads
                    // below) so that a caller's repeated calls ask the pool for the same block sizes -- the table's content,
                    // and with it this estimate, varies a little from call to call (insertion order)
                    const double f = std::ceil((cold_frac * 1.5 + 0.02) * 20.0) / 20.0;
                    if (f < 0.5) {
                        // all cold hits of a round may fall into one bucket of the staging area (plan.cap entries)
                        uint64_t per_region = plan.cap / cold_regions;
                        uint64_t round = per_region > 128 ? (uint64_t)((double)(per_region - 64) * cold_regions / f) : 0;
                        if (round > (1ull << 30)) round = 1ull << 30;
                        if (round > (32ull << 20)) round = round / (32ull << 20) * (32ull << 20);
                        if (round > n - first) round = n - first;
                        round = (round + MB_TILE - 1) / MB_TILE * MB_TILE;
                        uint64_t cap = (uint64_t)((double)round * f / cold_regions) + 64;
                        if (const char *env = getenv("CRGPU_COLD_CAP")) cap = strtoull(env, nullptr, 10);  // tests: force the overflow fallback
                        if ((round >= sb || round >= (n - first) / MB_TILE * MB_TILE) && round > 0 && cap * cold_regions <= plan.cap) {
                            const uint64_t lh_chunk = (uint64_t)LH_THREADS * LH_ITEMS;
                            int rr = count_mode ? CRGPU_OK
                                                : cr_pool_alloc(ctx, (void **)&d_hot_slot, (round + lh_chunk) / lh_chunk * lh_chunk * sizeof(uint16_t));
                            if (rr == CRGPU_OK) rr = cr_pool_alloc(ctx, (void **)&d_cold, cap * cold_regions * sizeof(uint32_t));
                            if (rr == CRGPU_OK) rr = cr_pool_alloc(ctx, (void **)&d_cold_count, (cold_regions + 1) * sizeof(uint32_t));
                            if (rr == CRGPU_OK) {
                                hot_round = round;
                                cold_cap = (uint32_t)cap;
                                cr_allow_lds(ctx, (const void *)k_hist_hot_slots, HOT_SLOTS * sizeof(uint32_t));
                            }  // else: not fatal, the rounds are staged in full as before
                        }
                    }
                }
            }
            if (e == hipSuccess) e = hipGetLastError();
        }
        off += m;
    }
    cr_pool_free(ctx, d_stage);
    cr_pool_free(ctx, d_cursor);
    if (e != hipSuccess) {
        cr_drop_miss_records(ctx, rec);
        return cr_fail(ctx, CRGPU_EHIP, "crgpu_match_and_count: %s", hipGetErrorString(e));
    }
    return CRGPU_OK;
}

void cr_drop_miss_records(crgpu_ctx *ctx) {
    for (MissRecords &r : ctx->recs) cr_drop_miss_records(ctx, r);
}
void cr_drop_miss_records(crgpu_ctx *ctx, MissRecords &r) {
    cr_pool_free(ctx, r.d_i);
    cr_pool_free(ctx, r.d_key);
    cr_pool_free(ctx, r.d_fl);
    cr_pool_free(ctx, r.d_count);
    r = MissRecords();
}

// ------------------------------------------------------------------------------------------------
// K2: posterior correction of the reads that missed
// ------------------------------------------------------------------------------------------------
#ifndef K2_SCAN_DWORDS
#define K2_SCAN_DWORDS 8  // entries of a pigeonhole bin per round = 2 * this (the bins hold ~11 entries)
#endif
#define MISS_ITEMS 64
// Compact the indices of the reads that missed.  One global atomic per 16384-read chunk: same-address
// atomics saturate near 88 per microsecond on this chip, so the reservation is aggregated over the
// whole workgroup and over 64 reads per thread (a 64-bit miss mask per thread).
// idx points at read i_base of the caller's arrays; run_if (nullable): do nothing unless *run_if != 0.
__global__ __launch_bounds__(256) void k_collect_miss(const uint32_t *__restrict__ idx, uint64_t n, uint32_t i_base,
                                                      const uint32_t *__restrict__ run_if,
                                                      uint32_t *__restrict__ miss_list,
                                                      unsigned long long *__restrict__ n_miss) {
    __shared__ __attribute__((aligned(8))) uint32_t lds[10];
    if (run_if && *run_if == 0u) return;
    const uint64_t chunk = 256ull * MISS_ITEMS;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        unsigned long long mask = 0;
        // unconditional loads from clamped addresses, 16 in flight: a load behind an `i < n` branch is not
        // issued before the previous one has been compared
#pragma unroll
        for (int j0 = 0; j0 < MISS_ITEMS; j0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const uint64_t i = c * chunk + (uint64_t)(j0 + jj) * 256 + threadIdx.x;
                v[jj] = idx[i < n ? i : n - 1];
            }
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const uint64_t i = c * chunk + (uint64_t)(j0 + jj) * 256 + threadIdx.x;
                if (i < n && v[jj] == CRGPU_MISS) mask |= 1ull << (j0 + jj);
            }
        }
        unsigned long long o = block_reserve_256((uint32_t)__popcll(mask), n_miss, lds);
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            miss_list[o++] = i_base + (uint32_t)(c * chunk + (uint64_t)j * 256 + threadIdx.x);
        }
    }
}

// exactly one differing 2-bit group between a and b (both < 2^16)?  returns the bit offset of that
// group (even) or -1.
__device__ __forceinline__ int one_base_diff(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    const uint32_t y = (x | (x >> 1)) & 0x5555u;
    if (y == 0u || (y & (y - 1u)) != 0u) return -1;
    return __ffs((int)y) - 1;
}

struct K2Params {
    const uint8_t *qualn;
    uint32_t len;
    const double *ptab;
    double max_expected, thresh;
    bool check_expected;
    uint32_t *idx_inout;
    uint8_t *corrected_out;
    bool have_flags;  // the call came with flag bytes: CRGPU_FLAG_CB_HAS_N tells which reads have an N
};
// qualities of read i (bit 7 = N), kept in two 64-bit registers: byte k of (lo, hi) = position k
struct K2Qual {
    unsigned long long lo, hi;
};
// false: the read cannot be corrected (no qualities and an N somewhere: its position is unknown)
__device__ __forceinline__ bool k2_load_qual(const K2Params &P, uint64_t i, uint32_t f, K2Qual &q) {
    const uint8_t *__restrict__ qualn = P.qualn;
    const uint32_t len = P.len;
#ifdef K2_EXP_NO_QUAL
    if (false) {
#else
    if (qualn) {
#endif
        if (len == 16) {
            const cr_u32x4 qv = CR_LOAD_STREAM(reinterpret_cast<const cr_u32x4 *>(qualn + i * 16));
            const uint4 q4 = make_uint4(qv.x, qv.y, qv.z, qv.w);
            q.lo = (unsigned long long)q4.x | ((unsigned long long)q4.y << 32);
            q.hi = (unsigned long long)q4.z | ((unsigned long long)q4.w << 32);
        } else {
            q.lo = 0ull;
            q.hi = 0ull;
            for (uint32_t k = 0; k < len; k++) {
                const unsigned long long b = qualn[i * len + k];
                if (k < 8) q.lo |= b << (8 * k); else q.hi |= b << (8 * (k - 8));
            }
        }
    } else {
        q.lo = q.hi = 0x4242424242424242ull;         // BC_MAX_QV = 66, corrector.rs:126 map_or
        if (f & CRGPU_FLAG_CB_HAS_N) return false;  // N position unknown without qualities
    }
    return true;
}
// movemask of the N bits: bit k = position k
__device__ __forceinline__ uint32_t k2_nmask(const K2Qual &q) {
    return (uint32_t)(((q.lo & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) |
           ((uint32_t)(((q.hi & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) << 8);
}
__device__ __forceinline__ uint32_t k2_qv(const K2Qual &q, uint32_t pos) {
    const uint32_t qv = (uint32_t)((pos < 8u ? q.lo : q.hi) >> (8u * (pos & 7u))) & 0x7Fu;
    return qv < 66u ? qv : 66u;  // corrector.rs:126
}
// the running maximum and sum of the likelihoods, candidates fed in position-major, A<C<G<T order (corrector.rs:146)
struct K2Best {
    bool have = false;
    double like = 0.0, total = 0.0;
    uint32_t rank = 0;
    __device__ __forceinline__ void feed(double l, uint32_t r) {
        if (!have) {
            have = true;
            like = l;
            rank = r;
        } else if (l > like || (l == like && r >= rank)) {
            // Ord::max on (NotNan, BarcodeSegment): ties go to the larger sequence == larger rank
            like = l;
            rank = r;
        }
        total += l;
    }
};
// the acceptance test (corrector.rs:152-160): the rank the read is corrected to, or CRGPU_MISS
__device__ __forceinline__ uint32_t k2_accept(const K2Params &P, const K2Qual &q, const K2Best &b) {
    if (!b.have) return CRGPU_MISS;
    double expected = 0.0;  // :154, uncapped qualities, in order; 0.0 without qualities
    if (P.check_expected)
        for (uint32_t k = 0; k < P.len; k++) expected += P.ptab[(uint32_t)((k < 8u ? q.lo : q.hi) >> (8u * (k & 7u))) & 0x7Fu];
    return (expected < P.max_expected && b.like / b.total >= P.thresh) ? b.rank : CRGPU_MISS;
}

// a read with ONE N: the N is "observed" -- all four bases are tried at its position (corrector.rs:128-131); a candidate
// built at any other position still contains the N and cannot match.  The four exact lookups and then the four prior counts
// are independent loads, issued together.
__device__ __forceinline__ uint32_t k2_solve_n(const WlView &w, const K2Params &P, uint32_t key, const K2Qual &q, uint32_t nmask) {
    const uint32_t pos = (uint32_t)__ffs((int)nmask) - 1u;
    const uint32_t sh = 2u * (P.len - 1u - pos);
    uint32_t r4[4], c4[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) r4[b] = wl_lookup(w, (key & ~(3u << sh)) | (b << sh));
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) c4[b] = r4[b] != CRGPU_MISS ? w.prior[r4[b]] : 0u;
    const double pq = P.ptab[k2_qv(q, pos)];
    K2Best best;
#pragma unroll
    for (uint32_t b = 0; b < 4; b++)
        if (r4[b] != CRGPU_MISS) best.feed(pq * (double)(1ll + (long long)c4[b]), r4[b]);  // :138-141, A<C<G<T
    return k2_accept(P, q, best);
}

// one missing read on its own: both pigeonhole bins scanned by this lane.  Returns the rank or CRGPU_MISS.
// have_q: q holds the read's qualities.  Otherwise the caller vouches that the read has no N (its flag byte) and that the
// expected-error veto is off, and load_q(q) fetches the quality line only if the candidates have to be weighed: a read with
// ONE candidate is corrected to it whatever its qualities are (likelihood / total == x / x == 1.0 for the positive finite x
// that any quality gives, corrector.rs:140-152), and that is 99 % of the misses on the 737 K list -- their random 16 bytes
// of a 128-byte line were a third of this pass (profiles/r02_k2_cost_attribution_ab.txt).
template <typename LoadQ>
__device__ __forceinline__ uint32_t k2_solve_own(const WlView &w, const K2Params &P, uint32_t key, K2Qual &q, bool have_q, LoadQ load_q) {
    const uint32_t len = P.len;
    const double *__restrict__ ptab = P.ptab;
    if (have_q) {
        const uint32_t nmask = k2_nmask(q);
        if (nmask) return (nmask & (nmask - 1u)) == 0u ? k2_solve_n(w, P, key, q, nmask) : CRGPU_MISS;
    }
    // candidate slots: bit (pos*4 + base)
    unsigned long long cand = 0ull;
    uint32_t posA = 0u, nA = 0u, posB = 0u, nB = 0u;
    const uint32_t tail_from = w.bitsA >> 1;  // first position that lies in the tail
    {
        const uint32_t head = key >> w.bitsB;
        const uint32_t tail = key & ((1u << w.bitsB) - 1u);
        const uint32_t hA = w.bitsA >> 1;
        // the four bin bounds are independent loads
        const U32x2 a2 = *reinterpret_cast<const U32x2 *>(w.offA + head);
        const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offB + tail);
        const uint32_t a_lo = a2.a, a_hi = a2.b, b_lo = b2.a, b_hi = b2.b;
        // mutation in the tail: same head -> bin A
        scan_u16_range<K2_SCAN_DWORDS>(w.tailA, a_lo, a_hi, [&](uint32_t t, uint32_t at) {
            const int bo = one_base_diff(t, tail);
            if (bo >= 0) {
                const uint32_t pos = len - 1u - (uint32_t)(bo >> 1);
                cand |= 1ull << (pos * 4u + ((t >> bo) & 3u));
                posA = at;  // bin A is the sorted whitelist itself: the entry's position gives its rank
                nA++;
            }
        });
        // mutation in the head: same tail -> bin B
        scan_u16_range<K2_SCAN_DWORDS>(w.headB, b_lo, b_hi, [&](uint32_t h, uint32_t at) {
            const int bo = one_base_diff(h, head);
            if (bo >= 0) {
                const uint32_t pos = hA - 1u - (uint32_t)(bo >> 1);
                cand |= 1ull << (pos * 4u + ((h >> bo) & 3u));
                posB = at;  // table B carries the ranks of its entries
                nB++;
            }
        });
    }
    if (!cand) return CRGPU_MISS;
    const bool lone = (cand & (cand - 1ull)) == 0ull;  // exactly one candidate, already known to be listed
    if (lone && !have_q) {
        // k2_accept for like == total and expected == 0.0 (no veto)
        const uint32_t r = nA ? (w.valA ? w.valA[posA] : posA) : w.valB[posB];
        return (0.0 < P.max_expected && 1.0 >= P.thresh) ? r : CRGPU_MISS;
    }
    if (!have_q) {
        load_q(q);
        const uint32_t nmask = k2_nmask(q);  // an N the flag byte did not announce
        if (nmask) return (nmask & (nmask - 1u)) == 0u ? k2_solve_n(w, P, key, q, nmask) : CRGPU_MISS;
    }
    K2Best best;
    while (cand) {
        const uint32_t slot = (uint32_t)__ffsll((long long)cand) - 1u;
        cand &= cand - 1ull;
        const uint32_t pos = slot >> 2, base = slot & 3u;
        const uint32_t sh = 2u * (len - 1u - pos);
        const uint32_t ckey = (key & ~(3u << sh)) | (base << sh);
        // a lone mutation of its half was seen at posA / posB: no second lookup (offE + tail lines) for its rank
        const uint32_t r = (nA == 1u && pos >= tail_from) ? (w.valA ? w.valA[posA] : posA)
                           : (nB == 1u && pos < tail_from) ? w.valB[posB] : wl_lookup(w, ckey);
        if (r == CRGPU_MISS) continue;
        const uint32_t qv = k2_qv(q, pos);
        if (lone && ptab[qv] > 0.0) {
            // the only candidate: likelihood / total == x / x == 1.0 for any positive finite x, whatever the
            // prior count is -- skip its random 4-byte load (the tables are 3x the L2 of an XCD)
            best.have = true;
            best.like = best.total = 1.0;
            best.rank = r;
            break;
        }
        const long long bc_count = 1ll + (long long)w.prior[r];    // Laplace smoothing, :138-139
        best.feed(ptab[qv] * (double)bc_count, r);                  // :140-141
    }
    return k2_accept(P, q, best);
}

// what a correction leaves behind.  rank_sink == NULL -> one device-scope atomicAdd on the library's CORRECTED table
// (scattered atomics run at ~20 G/s: 17 % of K2's time at 1 B reads); otherwise the rank is stored in *rank_sink (a
// compact, coalesced array) and the table is built afterwards by K1's staged LDS histogram
__device__ __forceinline__ void k2_commit(const WlView &w, const K2Params &P, uint64_t i, uint32_t rank, uint32_t *rank_sink,
                                          bool count = true, bool flag = true) {
    if (rank == CRGPU_MISS) return;
#ifndef K2_EXP_NO_IDX   // cost-attribution builds (scripts/ab.sh): results are wrong without these stores
    CR_STORE_STREAM(rank, &P.idx_inout[i]);
#endif
    if (flag && P.corrected_out) P.corrected_out[i] = 1;
#ifndef K2_EXP_NO_ATOMIC
    if (rank_sink)
        CR_STORE_STREAM(rank, rank_sink);
    else if (count)
        atomicAdd(&w.corrected[rank], 1u);
#endif
}

// one missing read: i = index in the caller's arrays, key = packed barcode, f = flag byte
template <bool UNIFORM>
__device__ __forceinline__ void k2_correct_one(const WlViewSet &vs, const K2Params &P, uint64_t i, uint32_t key, uint32_t f,
                                               uint32_t *rank_sink = nullptr) {
    const uint32_t lib = UNIFORM ? 0u : (f & CRGPU_FLAG_LIB_MASK);
    if (UNIFORM && (f & CRGPU_FLAG_LIB_MASK) != vs.ulib) return;
    const WlView &w = vs.v[lib];
    if (!UNIFORM && w.n == 0) return;
    K2Qual q{0ull, 0ull};
    // the flag byte says whether the read has an N (CRGPU_FLAG_CB_HAS_N is set by the pack kernels whenever bit 7 of a quality
    // byte is; pass A decides by it, too): without one, and without the veto, the quality line is fetched only on demand
    const bool lazy = P.qualn != nullptr && P.have_flags && !P.check_expected && !(f & CRGPU_FLAG_CB_HAS_N);
    if (!lazy && !k2_load_qual(P, i, f, q)) return;
    k2_commit(w, P, i, k2_solve_own(w, P, key, q, !lazy, [&](K2Qual &qq) { (void)k2_load_qual(P, i, f, qq); }), rank_sink);
}

// the misses as a list of read indices (k_collect_miss); run_if_zero (nullable): do nothing if *run_if_zero != 0 ...
template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_correct(const WlViewSet vs, const uint32_t *__restrict__ cb,
                                                 const uint8_t *__restrict__ flags, const uint32_t *__restrict__ miss_list,
                                                 const unsigned long long *__restrict__ n_miss_ptr, const K2Params P) {
    const uint32_t n_miss = (uint32_t)*n_miss_ptr;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_miss; j += stride) {
        const uint64_t i = miss_list[j];
        k2_correct_one<UNIFORM>(vs, P, i, cb[i], flags ? flags[i] : 0u);
    }
}

// ... and the misses as the records K1's lookup kernel left behind (coalesced reads instead of two random
// 128-byte lines per miss for cb and flags).  Does nothing when the records overflowed (k_collect_miss + k_correct
// then cover everything).
// rank_out (nullable) + rec_off: the rank every record was corrected to (CRGPU_MISS: not corrected), compact in record
// order: slot = rec_off[region] + position (rec_off = exclusive prefix of the region counts, k_region_offsets)
__global__ __launch_bounds__(256) void k_correct_records(const WlViewSet vs, const uint32_t *__restrict__ rec_i,
                                                         const uint32_t *__restrict__ rec_key, const uint8_t *__restrict__ rec_fl,
                                                         const uint32_t *__restrict__ rec_count, uint32_t rec_cap,
                                                         uint32_t rec_regions, const K2Params P,
                                                         const uint32_t *__restrict__ rec_off, uint32_t *__restrict__ rank_out,
                                                         uint32_t parts, const bool n_aside) {
    if (rec_count[rec_regions] != 0u) return;
    // A read with an N takes another path than the others (four exact lookups, its quality line, prior counts: a longer chain of
    // dependent loads), and a tenth of the misses of the cfg3 model are such reads: with them inline every wave ran both paths
    // one after the other.  They are set aside in a queue of the wave instead (LDS; a wave runs in lockstep and its LDS
    // operations complete in order) and worked off 64 at a time, so either path runs with all its lanes busy.
    __shared__ uint32_t s_q[256 / 64][3][128];  // per wave: read index, key, slot of the record (for the rank sink)
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const bool set_aside = n_aside && P.have_flags && P.qualn != nullptr;
    auto drain = [&](uint32_t from, uint32_t n_take, uint32_t base) {
        if (lane < n_take) {
            const uint32_t slot = s_q[wv][2][from + lane];
            uint32_t *sink = rank_out ? rank_out + base + slot : nullptr;
            k2_correct_one<true>(vs, P, s_q[wv][0][from + lane], s_q[wv][1][from + lane], vs.ulib | CRGPU_FLAG_CB_HAS_N, sink);
        }
    };
    // work items = (region, part): `parts` workgroups share a region's records (gridDim.x is a multiple of parts)
    for (uint32_t w = blockIdx.x; w < rec_regions * parts; w += gridDim.x) {
        const uint32_t r = w / parts, part = w % parts;
        const uint32_t cnt = rec_count[r];
        const uint32_t base = rank_out ? rec_off[r] : 0u;
        const uint32_t p_lo = (uint32_t)((uint64_t)cnt * part / parts), p_hi = (uint32_t)((uint64_t)cnt * (part + 1u) / parts);
        uint32_t qn = 0;  // wave-uniform
        for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 256) {  // uniform trip count: the queue bookkeeping is wave-wide
            const uint32_t p = p0 + threadIdx.x;
            const bool live = p < p_hi;
            const uint64_t o = (uint64_t)r * rec_cap + (live ? p : p_lo);
            const uint32_t i = CR_LOAD_STREAM(&rec_i[o]), key = CR_LOAD_STREAM(&rec_key[o]), fl = CR_LOAD_STREAM(&rec_fl[o]);
            uint32_t *sink = rank_out ? rank_out + base + p : nullptr;
            if (live && sink) *sink = CRGPU_MISS;
            const bool aside = live && set_aside && (fl & CRGPU_FLAG_CB_HAS_N) && (fl & CRGPU_FLAG_LIB_MASK) == vs.ulib;
            if (live && !aside) k2_correct_one<true>(vs, P, i, key, fl, sink);
            const unsigned long long m = __ballot(aside);
            if (m) {
                const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (aside) {
                    s_q[wv][0][at] = i;
                    s_q[wv][1][at] = key;
                    s_q[wv][2][at] = p;
                }
                qn += (uint32_t)__popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64u) {
                    drain(qn - 64u, 64u, base);
                    qn -= 64u;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        drain(0u, qn, base);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- K2 over the misses in BARCODE order ----------------------------------------------------------------------------
// A miss is a whitelist barcode with one substitution, and the same wrong sequence comes back many times (a cell's 10^4 - 10^5
// reads put ~8 % of them on its 48 neighbours: runs of a hundred equal keys once the misses are sorted).  The candidate set of
// a miss -- which listed barcodes lie one substitution away, their ranks and prior counts -- depends on its sequence only, so a
// wave that holds 64 consecutive sorted misses finds it ONCE per run of equal keys and lets the lanes of the run weigh the
// candidates with their own qualities.  With the 6.8 M-entry list the two pigeonhole bins hold ~104 entries each and the
// per-read scan took ~9 L2 misses and 1 KB of table lines per miss.
// The kernel is bound by dependent loads, so the search is laid out flat: up to KS_HEADS runs of the wave are taken together;
// the first lane of every run fetches its four bin bounds; the bins are cut into items of 8 entries (one 16-byte load) and ALL
// items of all the runs are spread over the 64 lanes, so the whole search is one or two rounds of independent loads whatever
// the bin length; a hit leaves its rank in the run's slot table in LDS (slot = position * 4 + base: at most one
// listed barcode per slot, and ascending slots are the accumulation order of corrector.rs:146).  Accumulation, tie rule and
// acceptance test are the ones of k2_solve_own (same K2Best / k2_accept).
// Reads with an N are not in the list: k_compact_records hands them to k_correct (the flag byte decides, as it does in pass A --
// CRGPU_FLAG_CB_HAS_N is set by the pack kernels whenever bit 7 of a quality byte is).  Their four candidates sit at the N,
// they need their quality line and the prior counts, and 10 % of the misses of the cfg3 model are such reads: inside this
// kernel (tried: marked in bit 31 of the index, candidates from the run's slot table plus an exact-match note) every wave
// held a few and waited for their loads, 3.2 -> 5.1 ms, which is what k_correct takes for them on its own.  Where the kernel
// reads the qualities of a read and finds an unannounced N, the read is appended to k_correct's list from here.  Corrected ranks are counted per run of equal ranks: one atomic per run.
#define KS_WAVES 4
#define KS_HEADS 8
__global__ __launch_bounds__(64 * KS_WAVES) void k_correct_sorted(const WlViewSet vs, const uint32_t *__restrict__ skey,
                                                                  const uint32_t *__restrict__ sidx, uint32_t n_miss,
                                                                  const K2Params P, uint32_t *__restrict__ late_list,
                                                                  unsigned long long *__restrict__ n_late) {
    __shared__ uint32_t s_rank[KS_WAVES][KS_HEADS][64], s_mask[KS_WAVES][KS_HEADS][2];
    __shared__ uint32_t s_run[KS_WAVES][KS_HEADS][5];          // key, a_lo, a_hi, b_lo, b_hi of the batch's runs
    __shared__ uint32_t s_first[KS_WAVES][2 * KS_HEADS + 1];   // first item of (run, bin); [2 * KS_HEADS] = number of items
    const WlView &w = vs.v[0];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t len = P.len;
    const uint32_t n_chunks = (n_miss + 63u) / 64u;
    const uint32_t hA = w.bitsA >> 1;
    const uint32_t tail_mask = (1u << w.bitsB) - 1u;
    const uint32_t *__restrict__ tA = reinterpret_cast<const uint32_t *>(w.tailA);
    const uint32_t *__restrict__ hB = reinterpret_cast<const uint32_t *>(w.headB);
    const uint32_t c_first = blockIdx.x * KS_WAVES + wv, c_step = gridDim.x * KS_WAVES;
    uint32_t key_next = c_first * 64u + lane < n_miss ? skey[c_first * 64u + lane] : 0xFFFFFFFFu;
    uint32_t i_next = c_first * 64u + lane < n_miss ? sidx[c_first * 64u + lane] : 0xFFFFFFFFu;
    for (uint32_t c = c_first; c < n_chunks; c += c_step) {
        const uint32_t key = key_next, i = i_next;
        {   // the next chunk's keys are on their way while this one is searched
            const uint32_t pn = (c + c_step) * 64u + lane;
            const bool ok = c + c_step < n_chunks && pn < n_miss;
            key_next = ok ? skey[pn] : 0xFFFFFFFFu;
            i_next = ok ? sidx[pn] : 0xFFFFFFFFu;
        }
        // A read whose run has ONE candidate is corrected to it whatever its qualities are (likelihood / total == x / x == 1.0 for
        // the positive finite x that any quality gives; corrector.rs:140-152) unless the expected-error veto is on: only the
        // reads of runs with several candidates fetch their quality line -- the random 64-byte access that bounds this kernel.
        const bool need_qual_always = P.check_expected;
        K2Qual q{0ull, 0ull};
        bool usable = i != 0xFFFFFFFFu;  // (records of another library / with an N carry no index)
        bool have_q = false;
        if (usable && need_qual_always) {
            usable = k2_load_qual(P, i, 0u, q);
            have_q = true;
        }
        const uint32_t left = __shfl_up(key, 1);
        const bool is_head = lane == 0u || key != left;
        const unsigned long long heads = __ballot(is_head), want = __ballot(usable);
        const uint32_t my_run = (uint32_t)__popcll(heads & (~0ull >> (63u - lane))) - 1u;  // my run's number in the wave
        const uint32_t n_runs = (uint32_t)__popcll(heads);
        // (a head lane) does any lane of my run want the candidates?
        const unsigned long long above = lane == 63u ? 0ull : heads >> (lane + 1u);
        const uint32_t run_len = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
        const unsigned long long run_lanes = (run_len >= 64u ? ~0ull : ((1ull << run_len) - 1ull)) << lane;
        const bool run_wanted = is_head && (want & run_lanes) != 0ull;
        uint32_t rank = CRGPU_MISS;
        for (uint32_t r0 = 0; r0 < n_runs; r0 += KS_HEADS) {  // wave-uniform
            const bool in_batch = my_run >= r0 && my_run < r0 + KS_HEADS;
            const bool mine = usable && in_batch;
            if (__ballot(mine) == 0ull) continue;
            const uint32_t b = my_run - r0;  // (lanes of the batch)
            if (lane < KS_HEADS) s_run[wv][lane][1] = s_run[wv][lane][2] = s_run[wv][lane][3] = s_run[wv][lane][4] = 0u;
            __builtin_amdgcn_wave_barrier();
            if (is_head && in_batch && run_wanted) {
                const U32x2 a2 = *reinterpret_cast<const U32x2 *>(w.offA + (key >> w.bitsB));
                const U32x2 b2 = *reinterpret_cast<const U32x2 *>(w.offB + (key & tail_mask));
                s_run[wv][b][0] = key;
                s_run[wv][b][1] = a2.a;
                s_run[wv][b][2] = a2.b;
                s_run[wv][b][3] = b2.a;
                s_run[wv][b][4] = b2.b;
                s_mask[wv][b][0] = 0u;
                s_mask[wv][b][1] = 0u;
            }
            __builtin_amdgcn_wave_barrier();
            {   // lanes 0 .. 2 * KS_HEADS - 1: items of 8 entries in (run, bin) = lane; exclusive prefix over these lanes
                uint32_t cnt = 0;
                if (lane < 2u * KS_HEADS) {
                    const uint32_t lo = s_run[wv][lane >> 1][1u + 2u * (lane & 1u)], hi = s_run[wv][lane >> 1][2u + 2u * (lane & 1u)];
                    cnt = hi > lo ? (hi - (lo & ~1u) + 7u) / 8u : 0u;
                }
                uint32_t inc = cnt;
#pragma unroll
                for (uint32_t d = 1; d < 2u * KS_HEADS; d <<= 1) {
                    const uint32_t y = __shfl_up(inc, d);
                    if (lane >= d) inc += y;
                }
                if (lane < 2u * KS_HEADS) s_first[wv][lane] = inc - cnt;
                if (lane == 2u * KS_HEADS - 1u) s_first[wv][2 * KS_HEADS] = inc;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t total = s_first[wv][2 * KS_HEADS];
            for (uint32_t it0 = 0; it0 < total; it0 += 64u) {  // wave-uniform
                const uint32_t it = it0 + lane;
                if (it >= total) continue;
                uint32_t g = 0;  // the last (run, bin) whose first item is <= it (empty ones share their successor's first item)
#pragma unroll
                for (uint32_t k = 1; k < 2u * KS_HEADS; k++) g += s_first[wv][k] <= it ? 1u : 0u;
                const uint32_t rb = g >> 1;
                const bool binB = (g & 1u) != 0u;
                const uint32_t qk = s_run[wv][rb][0];
                const uint32_t lo = s_run[wv][rb][binB ? 3 : 1], hi = s_run[wv][rb][binB ? 4 : 2];
                const uint32_t e0 = (lo & ~1u) + 8u * (it - s_first[wv][g]);
                const U32x4 d = *reinterpret_cast<const U32x4 *>((binB ? hB : tA) + (e0 >> 1));  // (tables padded by 32 bytes)
                const uint32_t other = binB ? qk >> w.bitsB : qk & tail_mask;  // the half that may differ in one base
                uint32_t hits = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
                    const uint32_t v = (d.w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                    if (e0 + k >= lo && e0 + k < hi && one_base_diff(v, other) >= 0) hits |= 1u << k;
                }
                while (hits) {
                    const uint32_t k = (uint32_t)__ffs((int)hits) - 1u;
                    hits &= hits - 1u;
                    const uint32_t dw = (k & 4u) ? ((k & 2u) ? d.w[3] : d.w[2]) : ((k & 2u) ? d.w[1] : d.w[0]);  // (no indexed register file)
                    const uint32_t v = (dw >> (16u * (k & 1u))) & 0xFFFFu;
                    const uint32_t bo = (uint32_t)one_base_diff(v, other);
                    const uint32_t base = (v >> bo) & 3u;
                    uint32_t pos, r;
                    if (!binB) {  // mutation in the tail: bin A is the sorted whitelist itself, position -> rank
                        pos = len - 1u - (bo >> 1);
                        r = w.valA ? w.valA[e0 + k] : e0 + k;
                    } else {      // mutation in the head: table B carries the ranks of its entries
                        pos = hA - 1u - (bo >> 1);
                        r = w.valB[e0 + k];
                    }
                    const uint32_t slot = pos * 4u + base;
                    s_rank[wv][rb][slot] = r;  // (its prior count is only needed when the run has a second candidate: fetched there)
                    atomicOr(&s_mask[wv][rb][slot >> 5], 1u << (slot & 31u));
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                unsigned long long cand = (unsigned long long)s_mask[wv][b][0] | ((unsigned long long)s_mask[wv][b][1] << 32);
                if (cand && (cand & (cand - 1ull)) == 0ull && !need_qual_always) {
                    if (1.0 >= P.thresh && 0.0 < P.max_expected) rank = s_rank[wv][b][(uint32_t)__ffsll((long long)cand) - 1u];  // k2_accept for like == total, expected == 0
                } else if (cand) {
                    if (!have_q) (void)k2_load_qual(P, i, 0u, q);
                    if (k2_nmask(q) != 0u) {  // an N the flag byte did not announce: k_correct takes the read
                        late_list[atomicAdd(n_late, 1ull)] = i;
                    } else {
                        K2Best best;
                        while (cand) {
                            const uint32_t slot = (uint32_t)__ffsll((long long)cand) - 1u;
                            cand &= cand - 1ull;
                            const uint32_t r = s_rank[wv][b][slot];
                            const long long bc_count = 1ll + (long long)w.prior[r];        // Laplace smoothing, :138-139
                            best.feed(P.ptab[k2_qv(q, slot >> 2)] * (double)bc_count, r);  // :140-141
                        }
                        rank = k2_accept(P, q, best);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // (the `corrected` bytes are written afterwards in read order, k_flag_corrected: a third random access per miss here)
        k2_commit(w, P, i, rank, nullptr, false, false);
#ifndef K2_EXP_NO_ATOMIC
        {   // one atomic per run of equal corrected ranks (a run of equal keys almost always agrees on its rank)
            const uint32_t lr = __shfl_up(rank, 1);
            const unsigned long long starts = __ballot(lane == 0u || rank != lr);
            const unsigned long long ab = lane == 63u ? 0ull : starts >> (lane + 1u);
            const uint32_t rl = ab ? (uint32_t)__ffsll((long long)ab) : 64u - lane;
            if (((starts >> lane) & 1ull) && rank != CRGPU_MISS) atomicAdd(&w.corrected[rank], rl);
        }
#endif
    }
}

// after k_correct_sorted, in record (= read) order: a recorded miss of the call's library whose index is no longer MISS was corrected
__global__ __launch_bounds__(256) void k_flag_corrected(const uint32_t *__restrict__ rec_i, const uint8_t *__restrict__ rec_fl,
                                                        const uint32_t *__restrict__ rec_count, uint32_t rec_cap, uint32_t rec_regions,
                                                        uint32_t ulib, const uint32_t *__restrict__ idx, uint8_t *__restrict__ corrected) {
    if (rec_count[rec_regions] != 0u) return;
    for (uint32_t r = blockIdx.x; r < rec_regions; r += gridDim.x) {
        const uint32_t cnt = rec_count[r];
        for (uint32_t p = threadIdx.x; p < cnt; p += 256) {
            const uint64_t o = (uint64_t)r * rec_cap + p;
            if ((CR_LOAD_STREAM(&rec_fl[o]) & CRGPU_FLAG_LIB_MASK) != ulib) continue;
            const uint32_t i = CR_LOAD_STREAM(&rec_i[o]);
            if (idx[i] != CRGPU_MISS) corrected[i] = 1;
        }
    }
}

// K1's miss records (one region per wave of the lookup, `cap` slots each) -> dense (key, read index) arrays for the sort.
// Records of another library keep their place with index 0xFFFFFFFF (k_correct_sorted skips them), and so do the reads with
// an N, whose read indices are appended to the miss list that k_correct works through (one reservation per workgroup and
// region).
__global__ __launch_bounds__(256) void k_compact_records(const uint32_t *__restrict__ rec_i, const uint32_t *__restrict__ rec_key,
                                                         const uint8_t *__restrict__ rec_fl, const uint32_t *__restrict__ rec_count,
                                                         uint32_t rec_cap, uint32_t rec_regions, const uint32_t *__restrict__ rec_off,
                                                         uint32_t ulib, uint32_t *__restrict__ out_key, uint32_t *__restrict__ out_i,
                                                         uint32_t *__restrict__ miss_list, unsigned long long *__restrict__ n_miss) {
    __shared__ __attribute__((aligned(8))) uint32_t lds[10];
    if (rec_count[rec_regions] != 0u) return;
    for (uint32_t r = blockIdx.x; r < rec_regions; r += gridDim.x) {
        const uint32_t cnt = rec_count[r], base = rec_off[r];
        uint32_t n_mine = 0;
        for (uint32_t p = threadIdx.x; p < cnt; p += 256) {
            const uint64_t o = (uint64_t)r * rec_cap + p;
            const uint32_t fl = CR_LOAD_STREAM(&rec_fl[o]);
            const bool lib_ok = (fl & CRGPU_FLAG_LIB_MASK) == ulib, has_n = (fl & CRGPU_FLAG_CB_HAS_N) != 0u;
            out_key[base + p] = CR_LOAD_STREAM(&rec_key[o]);
            out_i[base + p] = lib_ok && !has_n ? CR_LOAD_STREAM(&rec_i[o]) : 0xFFFFFFFFu;
            n_mine += lib_ok && has_n ? 1u : 0u;
        }
        unsigned long long o_n = block_reserve_256(n_mine, n_miss, lds);
        if (n_mine)
            for (uint32_t p = threadIdx.x; p < cnt; p += 256) {
                const uint64_t o = (uint64_t)r * rec_cap + p;
                const uint32_t fl = rec_fl[o];
                if ((fl & CRGPU_FLAG_LIB_MASK) == ulib && (fl & CRGPU_FLAG_CB_HAS_N)) miss_list[o_n++] = rec_i[o];
            }
    }
}

// exclusive prefix of the per-region record counts (a few thousand regions): off[r], off[n] = total; all zero when the
// records overflowed (k_correct_records then does nothing)
__global__ __launch_bounds__(1024) void k_region_offsets(const uint32_t *__restrict__ count, uint32_t n, uint32_t *__restrict__ off) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const bool overflow = count[n] != 0u;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t r = base + threadIdx.x;
        const uint32_t v = (r < n && !overflow) ? count[r] : 0u;
        uint32_t tot;
        const uint32_t pre = block_excl_scan<1024>(v, lds, &tot);
        const uint32_t carry = carry_s;
        if (r < n) off[r] = carry + pre;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = carry_s;
}

extern "C" int crgpu_set_posterior(crgpu_ctx *ctx, double max_expected_barcode_errors, double bc_confidence_threshold) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    ctx->max_expected_errors = max_expected_barcode_errors;
    ctx->confidence_threshold = bc_confidence_threshold;
    return CRGPU_OK;
}

// fake_quals: the quality bytes only carry the N flags (host path without qualities): the
// expected-error sum is 0.0 as in corrector.rs:154 map_or.
static int correct_dev_impl(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_qualn, const uint8_t *d_flags,
                            uint64_t n, uint32_t *d_idx_inout, uint8_t *d_corrected_out, bool fake_quals) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_correct: no whitelist set");
    CR_REQUIRE(ctx, ctx->n_segments == 0, CRGPU_ESTATE,
               "crgpu_correct: this context holds a segmented barcode space; run the barcode stage on the segment contexts");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_cb && d_idx_inout, CRGPU_EINVAL, "crgpu_correct: NULL buffer");
    CR_REQUIRE(ctx, n < 0xFFFFFFFFull, CRGPU_ERANGE, "crgpu_correct: batches are limited to 2^32-2 reads");
    cr_dense_drop(ctx);  // the CORRECTED table changes
    if (ctx->cb_len == 16 && d_qualn)
        CR_REQUIRE(ctx, (uintptr_t)d_qualn % 16 == 0, CRGPU_EINVAL, "crgpu_correct: quality buffer must be 16-byte aligned");
    // NotNan::try_from(threshold).ok()? (corrector.rs:152): a NaN threshold corrects nothing
    if (ctx->confidence_threshold != ctx->confidence_threshold) return CRGPU_OK;
    // the records K1 left for exactly these buffers (consumed here: a second call scans idx again) also say which
    // library the call is about; otherwise the flag bytes do
    MissRecords *recp = nullptr;
    for (MissRecords &r : ctx->recs)
        if (r.valid && r.d_cb == d_cb && r.d_flags == d_flags && r.d_idx == d_idx_inout && r.n == n) recp = &r;
    const bool use_rec = recp != nullptr;
    MissRecords none;
    MissRecords &rec = recp ? *recp : none;
    int ulib = -1;
    if (use_rec)
        ulib = rec.ulib;
    else
        CR_TRY(pick_uniform_lib(ctx, d_flags, n, &ulib));
    const bool uniform = ulib >= 0;
    WlViewSet vs;
    CR_TRY(make_view_set(ctx, vs, ulib));
    void *ws;
    CR_TRY(cr_scratch(ctx, n * sizeof(uint32_t), &ws));
    uint32_t *miss_list = (uint32_t *)ws;
    unsigned long long *n_miss = (unsigned long long *)ctx->d_scalars;
    CrTimer t(ctx, CRGPU_T_CORRECT, n);
    CR_HIP(ctx, hipMemsetAsync(n_miss, 0, sizeof(unsigned long long), ctx->stream));
    if (d_corrected_out) CR_HIP(ctx, hipMemsetAsync(d_corrected_out, 0, n, ctx->stream));
    // expected_errors < f64::MAX is always true for a finite sum: skip the sum for the default
    const bool check_expected = d_qualn && !fake_quals && ctx->max_expected_errors < 1.7976931348623157e308;
    const K2Params P{d_qualn, ctx->cb_len, ctx->d_ptab, ctx->max_expected_errors, ctx->confidence_threshold, check_expected,
                     d_idx_inout, d_corrected_out, d_flags != nullptr && getenv("CRGPU_K2_EAGER_QUAL") == nullptr};  // (A/B switch)
    if (use_rec) {
        // reads before rec.first (K1's sampling batch) are not in the records; everything is scanned when they overflowed
        const uint32_t *overflow = rec.d_count + rec.regions;
        if (rec.first)
            hipLaunchKernelGGL(k_collect_miss, dim3(cr_grid(rec.first, 256)), dim3(256), 0, ctx->stream, d_idx_inout, rec.first, 0u,
                               (const uint32_t *)nullptr, miss_list, n_miss);
        hipLaunchKernelGGL(k_collect_miss, dim3(cr_grid(n - rec.first, 256)), dim3(256), 0, ctx->stream, d_idx_inout + rec.first,
                           n - rec.first, (uint32_t)rec.first, overflow, miss_list, n_miss);
    } else {
        hipLaunchKernelGGL(k_collect_miss, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_idx_inout, n, 0u,
                           (const uint32_t *)nullptr, miss_list, n_miss);
    }
    // Barcode order (k_correct_sorted) for the recorded misses: pays once the pigeonhole bins are long (the 6.8 M-entry list:
    // ~104 entries per bin against ~11 with 737 K); CRGPU_K2_SORTED=0/1 forces either way.  The reads with an N join the
    // miss list of the k_correct launch below.
    bool sorted = use_rec && uniform && d_qualn && d_flags && (uint64_t)rec.regions * rec.cap < 0xFFFFFFFFull;
    if (const char *g = getenv("CRGPU_K2_SORTED")) sorted = sorted && atoi(g) != 0;
    else sorted = sorted && ctx->n_canon > (1u << 21);
    int sorted_rc = CRGPU_OK;
    if (sorted) {
        uint32_t *d_o = nullptr, *d_k = nullptr, *d_v = nullptr, *d_kt = nullptr, *d_vt = nullptr;
        int rc = cr_pool_alloc(ctx, (void **)&d_o, (rec.regions + 1) * sizeof(uint32_t));
        uint32_t total = 0;
        if (rc == CRGPU_OK) {
            hipLaunchKernelGGL(k_region_offsets, dim3(1), dim3(1024), 0, ctx->stream, rec.d_count, rec.regions, d_o);
            rc = crgpu_memcpy_d2h(ctx, &total, d_o + rec.regions, sizeof(total));  // 0 when the records overflowed
        }
        if (rc == CRGPU_OK && total) {
            const uint64_t bytes = (uint64_t)total * sizeof(uint32_t);
            if (cr_pool_alloc(ctx, (void **)&d_k, bytes) != CRGPU_OK || cr_pool_alloc(ctx, (void **)&d_v, bytes) != CRGPU_OK ||
                cr_pool_alloc(ctx, (void **)&d_kt, bytes) != CRGPU_OK || cr_pool_alloc(ctx, (void **)&d_vt, bytes) != CRGPU_OK)
                rc = cr_fail(ctx, CRGPU_ENOMEM, "crgpu_correct: no memory for the sorted misses");
        }
        if (rc == CRGPU_OK && total) {
            hipLaunchKernelGGL(k_compact_records, dim3(rec.regions), dim3(256), 0, ctx->stream, rec.d_i, rec.d_key, rec.d_fl, rec.d_count,
                               rec.cap, rec.regions, d_o, vs.ulib, d_k, d_v, miss_list, n_miss);
            bool in_tmp = false;
            {   // the passes of this sort belong to the CORRECT span that is already open (on its own the 32-bit sort books
                // itself under the count stage, whose candidate sort it normally is)
                const bool timing = ctx->timing;
                ctx->timing = false;
                const uint32_t kb = 2u * ctx->cb_len;
                uint32_t sb = kb;
                if (const char *g = getenv("CRGPU_K2_SORT_BITS")) sb = (uint32_t)atoi(g);  // A/B: only the top bits
                if (sb < 8u || sb > kb) sb = kb;
                rc = cr_radix_sort_u32(ctx, d_k, d_kt, d_v, d_vt, total, kb - sb, kb, &in_tmp);
                ctx->timing = timing;
            }
            if (rc == CRGPU_OK) {
                const uint32_t chunks = (total + 63u) / 64u;
                const uint32_t wgs = (chunks + KS_WAVES - 1) / KS_WAVES;
                uint32_t per_cu = 16u;
                if (const char *g = getenv("CRGPU_K2_WGS")) per_cu = (uint32_t)atoi(g);  // A/B
                if (per_cu < 1u || per_cu > 64u) per_cu = 16u;
                hipLaunchKernelGGL(k_correct_sorted, dim3(wgs < 256u * per_cu ? wgs : 256u * per_cu), dim3(64 * KS_WAVES), 0, ctx->stream, vs,
                                   in_tmp ? d_kt : d_k, in_tmp ? d_vt : d_v, total, P, miss_list, n_miss);
                if (d_corrected_out)
                    hipLaunchKernelGGL(k_flag_corrected, dim3(rec.regions), dim3(256), 0, ctx->stream, rec.d_i, rec.d_fl, rec.d_count,
                                       rec.cap, rec.regions, vs.ulib, d_idx_inout, d_corrected_out);
                if (hipGetLastError() != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_correct: launch failed");
            }
        }
        cr_pool_free(ctx, d_o);
        cr_pool_free(ctx, d_k);
        cr_pool_free(ctx, d_v);
        cr_pool_free(ctx, d_kt);
        cr_pool_free(ctx, d_vt);
        sorted_rc = rc;
    }
    // K2 is launched for the worst case and loops over the device-side count: no host round trip
    const dim3 grid(cr_grid((use_rec ? rec.first + n / 64 : n) / 8 + 1, 256)), block(256);
    if (uniform)
        hipLaunchKernelGGL(k_correct<true>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, miss_list, n_miss, P);
    else
        hipLaunchKernelGGL(k_correct<false>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, miss_list, n_miss, P);
    if (sorted) {
        CR_HIP(ctx, hipGetLastError());
        cr_drop_miss_records(ctx, rec);  // stream-ordered: the pool reuses the blocks only for later work
        return sorted_rc;
    }
    if (use_rec) {
        // the corrected ranks of the records go to a compact array and are counted by K1's staged LDS histogram (when its
        // bucket plan applies: one library, at most 31 rank buckets) instead of one scattered atomic per corrected read
        BinPlan plan;
        for (int l = 0; l < CRGPU_MAX_LIB; l++) plan.lib_slot[l] = l == 0 ? 0u : 0xFFFFFFFFu;
        plan.buckets_per_lib = (ctx->n_canon + BIN_SIZE - 1) / BIN_SIZE;
        plan.n_buckets = plan.buckets_per_lib;
        uint32_t *d_off = nullptr, *d_rank = nullptr;
        const uint64_t rec_total_cap = (uint64_t)rec.regions * rec.cap;
        static const bool no_staged = getenv("CRGPU_K2_ATOMIC_HIST") != nullptr;  // A/B switch
        bool staged = !no_staged && plan.n_buckets <= SI_MISS_BUCKET && rec_total_cap < 0xFFFFFFFFull;
        if (staged && (cr_pool_alloc(ctx, (void **)&d_off, (rec.regions + 1) * sizeof(uint32_t)) != CRGPU_OK ||
                       cr_pool_alloc(ctx, (void **)&d_rank, rec_total_cap * sizeof(uint32_t)) != CRGPU_OK)) {
            cr_pool_free(ctx, d_off);
            d_off = d_rank = nullptr;
            staged = false;  // not fatal: atomics as before
        }
        if (staged) hipLaunchKernelGGL(k_region_offsets, dim3(1), dim3(1024), 0, ctx->stream, rec.d_count, rec.regions, d_off);
        // one workgroup per region (the regions are K1's waves: 4096): with 2048 workgroups taking two regions each, the 256
        // that do not fit beside the 1792 resident ones ran almost alone at the end (K2 3.24 -> 2.87 ms per 500 M reads)
        uint32_t k2_parts = 4;  // four workgroups per region: 2.88 -> 2.74 ms per 500 M reads on top of the one-per-region gain
        if (const char *g = getenv("CRGPU_K2_PARTS")) k2_parts = (uint32_t)atoi(g);  // A/B
        if (k2_parts < 1 || k2_parts > 16) k2_parts = 1;
        uint32_t k2_grid = rec.regions * k2_parts;
        if (const char *g = getenv("CRGPU_K2_GRID")) k2_grid = (uint32_t)atoi(g) / k2_parts * k2_parts;  // A/B
        if (k2_grid < k2_parts) k2_grid = k2_parts;
        hipLaunchKernelGGL(k_correct_records, dim3(k2_grid), dim3(256), 0, ctx->stream, vs, rec.d_i, rec.d_key, rec.d_fl, rec.d_count,
                           rec.cap, rec.regions, P, d_off, d_rank, k2_parts, getenv("CRGPU_K2_N_INLINE") == nullptr);  // (A/B switch)
        int rc = CRGPU_OK;
        if (hipGetLastError() != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_correct: launch failed");
        uint32_t total = 0;
        if (rc == CRGPU_OK && staged) rc = crgpu_memcpy_d2h(ctx, &total, d_off + rec.regions, sizeof(total));
        if (rc == CRGPU_OK && staged && total) {
            WlViewSet vc = vs;
            vc.v[0].valid = vc.v[0].corrected;  // k_hist_buckets adds to `valid` of the call's library (v[0])
            rc = hist_from_idx(ctx, vc, plan, d_rank, total);
        }
        cr_pool_free(ctx, d_off);
        cr_pool_free(ctx, d_rank);
        cr_drop_miss_records(ctx, rec);  // stream-ordered: the pool reuses the blocks only for later work
        return rc;
    }
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

extern "C" int crgpu_correct_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_qualn, const uint8_t *d_flags,
                                 uint64_t n, uint32_t *d_idx_inout, uint8_t *d_corrected_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    return correct_dev_impl(ctx, d_cb, d_qualn, d_flags, n, d_idx_inout, d_corrected_out, false);
}

// ------------------------------------------------------------------------------------------------
// segmented constructs: combined rank of the segments' ranks + the whole-barcode histograms
// (MakeShardHistograms::observe, make_shard_metrics.rs:171-190; corrected_barcode_counts, barcode_correction.rs:401-407)
// ------------------------------------------------------------------------------------------------
struct SegIdx {
    const uint32_t *idx[CRGPU_MAX_SEGMENTS];
    uint32_t n[CRGPU_MAX_SEGMENTS];
    uint32_t n_seg;
};

// AFTER == false: out[i] = combined rank or MISS.  AFTER == true: a read that had no barcode and has one now gets its
// rank in out[i] and in fresh[i] (the array the CORRECTED histogram is made of); everything else is MISS in fresh.
template <bool AFTER>
__global__ __launch_bounds__(256) void k_combine_segments(const SegIdx S, uint64_t n, uint32_t *__restrict__ out,
                                                          uint32_t *__restrict__ fresh) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t r = 0;
        bool ok = true;
        for (uint32_t s = 0; s < S.n_seg; s++) {
            const uint32_t v = S.idx[s][i];
            ok = ok && v < S.n[s];
            r = r * S.n[s] + (ok ? v : 0u);
        }
        if (!ok) r = CRGPU_MISS;
        if constexpr (AFTER) {
            const bool is_new = out[i] == CRGPU_MISS && r != CRGPU_MISS;
            if (is_new) out[i] = r;
            fresh[i] = is_new ? r : CRGPU_MISS;
        } else {
            out[i] = r;
        }
    }
}

extern "C" int crgpu_combine_segments_dev(crgpu_ctx *ctx, int lib, const uint32_t *const *d_seg_idx, uint32_t n_segments,
                                          uint64_t n, int after_correction, uint32_t *d_idx_inout) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_dense_drop(ctx);
    CR_REQUIRE(ctx, ctx->n_segments > 0, CRGPU_ESTATE, "crgpu_combine_segments_dev: call crgpu_set_barcode_segments first");
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB && ctx->wl[lib].set, CRGPU_ESTATE, "library %d has no barcode space", lib);
    CR_REQUIRE(ctx, n_segments == ctx->n_segments && d_seg_idx, CRGPU_EINVAL, "crgpu_combine_segments_dev: %u segments expected",
               ctx->n_segments);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_idx_inout, CRGPU_EINVAL, "crgpu_combine_segments_dev: NULL output");
    SegIdx S{};
    S.n_seg = n_segments;
    for (uint32_t s = 0; s < n_segments; s++) {
        CR_REQUIRE(ctx, d_seg_idx[s], CRGPU_EINVAL, "crgpu_combine_segments_dev: NULL idx array of segment %u", s);
        S.idx[s] = d_seg_idx[s];
        S.n[s] = ctx->seg_n[s];
    }
    cr_invalidate(ctx);
    uint32_t *d_fresh = nullptr;
    if (after_correction) CR_TRY(cr_pool_alloc(ctx, (void **)&d_fresh, n * sizeof(uint32_t)));
    struct Release {
        crgpu_ctx *c;
        void *p;
        ~Release() { cr_pool_free(c, p); }
    } rel{ctx, d_fresh};
    CrTimer t(ctx, after_correction ? CRGPU_T_CORRECT : CRGPU_T_MATCH, n);
    if (after_correction)
        hipLaunchKernelGGL(k_combine_segments<true>, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, S, n, d_idx_inout, d_fresh);
    else
        hipLaunchKernelGGL(k_combine_segments<false>, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, S, n, d_idx_inout, d_fresh);
    CR_HIP(ctx, hipGetLastError());
    // the histogram of the new ranks: K1's staging kernels when the space fits their 31 buckets of 32768 ranks, device
    // atomics otherwise (a product space of millions of barcodes; ~20 G reads/s, same-barcode runs serialise)
    const WlTables &w = ctx->wl[lib];
    uint32_t *table = after_correction ? w.d_corrected : w.d_valid;
    const uint32_t *src = after_correction ? d_fresh : d_idx_inout;
    BinPlan plan;
    for (int l = 0; l < CRGPU_MAX_LIB; l++) plan.lib_slot[l] = l == 0 ? 0u : 0xFFFFFFFFu;
    plan.buckets_per_lib = (ctx->n_canon + BIN_SIZE - 1) / BIN_SIZE;
    plan.n_buckets = plan.buckets_per_lib;
    if (plan.n_buckets <= SI_MISS_BUCKET) {
        WlViewSet vs{};
        vs.n_canon = ctx->n_canon;
        vs.ulib = (uint32_t)lib;
        vs.v[0].valid = table;
        CR_TRY(hist_from_idx(ctx, vs, plan, src, n));
    } else {
        hipLaunchKernelGGL(k_hist_ranks_atomic, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, src, n, table, (const uint32_t *)nullptr);
        CR_HIP(ctx, hipGetLastError());
    }
    if (d_fresh) CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the block goes back to the pool
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// host-buffer convenience entry points (SURVEY.md 8b signatures)
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_u8(uint8_t *p, uint64_t n, uint8_t v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

struct HostBatch {
    uint8_t *d_seq = nullptr, *d_qual = nullptr, *d_qualn = nullptr, *d_flags = nullptr;
    uint32_t *d_cb = nullptr, *d_idx = nullptr;
    uint8_t *d_corr = nullptr;
    ~HostBatch() {
        hipFree(d_seq);
        hipFree(d_qual);
        hipFree(d_qualn);
        hipFree(d_flags);
        hipFree(d_cb);
        hipFree(d_idx);
        hipFree(d_corr);
    }
};

static int stage_host_batch(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n, HostBatch &b) {
    const uint32_t len = ctx->cb_len;
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB && ctx->wl[lib].set, CRGPU_ESTATE, "library %d has no whitelist", lib);
    CR_HIP(ctx, hipMalloc((void **)&b.d_seq, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_qual, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_qualn, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_flags, n));
    CR_HIP(ctx, hipMalloc((void **)&b.d_cb, n * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&b.d_idx, n * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&b.d_corr, n));
    CR_HIP(ctx, hipMemcpyAsync(b.d_seq, seq, n * len, hipMemcpyHostToDevice, ctx->stream));
    if (qual) {
        CR_HIP(ctx, hipMemcpyAsync(b.d_qual, qual, n * len, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // no qualities (corrector.rs:126 map_or): every base gets BC_MAX_QV
        hipLaunchKernelGGL(k_fill_u8, dim3(cr_grid(n * len, 256)), dim3(256), 0, ctx->stream, b.d_qual, n * len, (uint8_t)66);
    }
    hipLaunchKernelGGL(k_fill_u8, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, b.d_flags, n, (uint8_t)lib);
    CR_TRY(crgpu_pack_dev(ctx, b.d_seq, b.d_qual, n, len, b.d_cb, b.d_qualn, b.d_flags));
    return CRGPU_OK;
}

extern "C" int crgpu_match_and_count(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                                     uint32_t *idx_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_match_and_count: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, seq && idx_out, CRGPU_EINVAL, "crgpu_match_and_count: NULL buffer");
    HostBatch b;
    CR_TRY(stage_host_batch(ctx, lib, seq, qual, n, b));
    CR_TRY(crgpu_match_and_count_dev(ctx, b.d_cb, b.d_flags, n, b.d_idx));
    cr_drop_miss_records(ctx);  // the staging buffers are temporaries: a later call may get the same addresses back
    CR_TRY(crgpu_memcpy_d2h(ctx, idx_out, b.d_idx, n * sizeof(uint32_t)));
    return CRGPU_OK;
}

extern "C" int crgpu_correct(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                             uint32_t *idx_inout, uint8_t *corrected_flag_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_correct: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, seq && idx_inout, CRGPU_EINVAL, "crgpu_correct: NULL buffer");
    HostBatch b;
    CR_TRY(stage_host_batch(ctx, lib, seq, qual, n, b));
    CR_HIP(ctx, hipMemcpyAsync(b.d_idx, idx_inout, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    CR_TRY(correct_dev_impl(ctx, b.d_cb, b.d_qualn, b.d_flags, n, b.d_idx, b.d_corr, qual == nullptr));
    CR_TRY(crgpu_memcpy_d2h(ctx, idx_inout, b.d_idx, n * sizeof(uint32_t)));
    if (corrected_flag_out) CR_TRY(crgpu_memcpy_d2h(ctx, corrected_flag_out, b.d_corr, n));
    return CRGPU_OK;
}
