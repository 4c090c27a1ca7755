// dedup.hip -- count stage: molecule keys, run-length grouping, UMI correction, low-support
// filtering and (barcode, feature) counting on sorted 64-bit keys.
//
// Replaces, per (barcode, library type) group of the reference:
//   UmiInfo::new                      umi/src/info.rs:20-37        (validity of each UMI)
//   DupBuilder::observe               tx_annotation/src/mark_dups.rs:128-155
//   correct_umis                      mark_dups.rs:19-59
//   BarcodeDupMarker::new             mark_dups.rs:202-277  (two-phase count move)
//   determine_low_support_umigenes    mark_dups.rs:87-108
//   BarcodeDupMarker::process         mark_dups.rs:280-363  (which keys yield a UmiCount)
//   BcUmiInfo::feature_counts         cr_types/src/types.rs:180-188
//
// Key layout (KeyLayout in common.h), most significant first:
//   [barcode rank][feature][library][UMI 2-bit][nonTxomic]
// so that after ONE ascending sort
//   * equal keys (ignoring the last bit) are the reads of one (UMI, feature): run length = read count,
//   * a (barcode, feature, library) segment holds every UMI that correct_umis may compare,
//   * a (barcode, feature) segment is one matrix entry.
// Low-support grouping needs (barcode, library, UMI) across features: the DISTINCT keys are sorted a
// second time by a 32-bit hash of that triple (index as payload) and re-checked exactly.
#include "block_utils.h"
#include "common.h"

int cr_scan_small(crgpu_ctx *ctx, uint32_t *d_data, uint64_t n, uint32_t *d_total_out);
int cr_partition_by_owner(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, uint32_t sh_bc,
                          uint32_t n_ranks, const uint32_t *bounds, uint64_t *counts_out);

#define NONE32 0xFFFFFFFFu

struct crgpu_counts {
    uint64_t n_triplets = 0, n_molecules = 0;
    uint32_t *d_bc = nullptr, *d_feature = nullptr, *d_count = nullptr;  // triplets
    uint64_t *d_mkeys = nullptr;    // molecule keys (primary layout), n_molecules
    uint32_t *d_mreads = nullptr;   // read_count of each molecule
    uint32_t *d_corr_reads = nullptr;  // [library][barcode rank] reads whose UMI was corrected (BarcodeSummary), or NULL
    uint32_t *d_filt_reads = nullptr;  // [library][barcode rank] reads of molecules the targeted-panel filter removed, or NULL
    int32_t *d_mprobe = nullptr;    // probe_idx of each molecule's representative read (crgpu_records.d_probe_idx given), or NULL
    uint32_t *d_back = nullptr;     // CRGPU_OPT_DENSE_BARCODE_KEYS: column -> whitelist rank of the barcode field of d_mkeys (n_back), else NULL
    uint32_t n_back = 0;
    uint32_t n_canon = 0;
    KeyLayout layout;
};

// ------------------------------------------------------------------------------------------------
// key layout
// ------------------------------------------------------------------------------------------------
extern "C" int crgpu_set_key_layout(crgpu_ctx *ctx, uint32_t n_features, uint32_t umi_len, uint32_t n_libs,
                                    uint32_t multiplexing_lib_mask) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_set_key_layout: set the whitelist first");
    CR_REQUIRE(ctx, n_features >= 1, CRGPU_EINVAL, "n_features must be >= 1");
    CR_REQUIRE(ctx, umi_len >= 1 && umi_len <= 16, CRGPU_ERANGE, "umi_len must be 1..16");
    CR_REQUIRE(ctx, n_libs >= 1 && n_libs <= CRGPU_MAX_LIB, CRGPU_EINVAL, "n_libs must be 1..%d", CRGPU_MAX_LIB);
    KeyLayout L;
    L.bits_bc = cr_ceil_log2(ctx->n_canon);
    L.bits_feat = cr_ceil_log2(n_features);
    L.bits_lib = cr_ceil_log2(n_libs);
    L.bits_umi = 2 * umi_len;
    L.n_features = n_features;
    L.umi_len = umi_len;
    L.umi_min_len = umi_len;
    L.n_libs = n_libs;
    L.mux_mask = multiplexing_lib_mask;
    // with CRGPU_OPT_DENSE_BARCODE_KEYS the barcode field shrinks to the columns that occur (known when the first key is built)
    CR_REQUIRE(ctx, L.total_bits() <= 64 || (ctx->dense.on && L.total_bits() - L.bits_bc + 1u <= 64u), CRGPU_ERANGE,
               "molecule key needs %u bits (barcode %u + feature %u + library %u + umi %u + 1) > 64%s", L.total_bits(),
               L.bits_bc, L.bits_feat, L.bits_lib, L.bits_umi,
               ctx->dense.on ? "" : " (CRGPU_OPT_DENSE_BARCODE_KEYS shortens the barcode field to the barcodes that occur)");
    L.set = true;
    cr_dense_drop(ctx);
    ctx->dense.canon_bits = L.bits_bc;
    ctx->layout = L;
    cr_invalidate(ctx);
    return CRGPU_OK;
}

// targeted gene expression: DupBuilder::build(.., targeted_umi_min_read_count) + FeatureReference::target_set
// (mark_dups.rs:156-169,311-320; the threshold comes from _slfe_matrix_computer.mro:122-140).  on_target: n_features bytes
// (host), non-zero = the feature is in the target set; NULL or min_read_count == 0 switches the filter off.
extern "C" int crgpu_set_target_filter(crgpu_ctx *ctx, const uint8_t *on_target, uint32_t n_features, uint64_t min_read_count) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_on_target) CR_HIP(ctx, hipFree(ctx->d_on_target));
    ctx->d_on_target = nullptr;
    ctx->n_target_features = 0;
    ctx->target_min_reads = 0;
    if (!on_target || !min_read_count || !n_features) return CRGPU_OK;
    CR_HIP(ctx, hipMalloc((void **)&ctx->d_on_target, n_features));
    CR_HIP(ctx, hipMemcpy(ctx->d_on_target, on_target, n_features, hipMemcpyHostToDevice));
    ctx->n_target_features = n_features;
    ctx->target_min_reads = min_read_count;
    return CRGPU_OK;
}

// Per-read UMI lengths (UmiExtractor::extract_umi, cr_types/src/rna_read.rs:103-138: a read that ends early keeps
// max(min(read_len - offset, length), min_length) bases): UMIs of different lengths are different UmiSeqs, so the length
// becomes part of the key -- a tag (umi_len - length) right above the UMI bits.  Call after crgpu_set_key_layout.
extern "C" int crgpu_set_umi_min_len(crgpu_ctx *ctx, uint32_t umi_min_len) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->layout.set, CRGPU_ESTATE, "crgpu_set_umi_min_len: call crgpu_set_key_layout first");
    KeyLayout L = ctx->layout;
    CR_REQUIRE(ctx, umi_min_len >= 1 && umi_min_len <= L.umi_len, CRGPU_EINVAL, "umi_min_len must be 1..umi_len (%u)", L.umi_len);
    L.umi_min_len = umi_min_len;
    L.bits_ulen = cr_ceil_log2(L.umi_len - umi_min_len + 1);
    cr_dense_drop(ctx);
    L.bits_bc = ctx->dense.canon_bits;
    CR_REQUIRE(ctx, L.total_bits() <= 64 || (ctx->dense.on && L.total_bits() - L.bits_bc + 1u <= 64u), CRGPU_ERANGE,
               "molecule key needs %u bits with %u bits of UMI length > 64", L.total_bits(), L.bits_ulen);
    ctx->layout = L;
    cr_invalidate(ctx);
    return CRGPU_OK;
}

struct KL {  // device copy of the layout
    uint32_t sh_umi, sh_lib, sh_feat, sh_bc, bits_umi, bits_lib, bits_feat, bits_bc, umi_len, n_features, n_libs, mux_mask;
    uint32_t bits_ulen, sh_libid, umi_min_len;
};
static KL make_kl(const KeyLayout &L) {
    return KL{L.sh_umi(), L.sh_lib(), L.sh_feat(), L.sh_bc(), L.bits_umi, L.bits_lib, L.bits_feat, L.bits_bc,
              L.umi_len, L.n_features, L.n_libs, L.mux_mask, L.bits_ulen, L.sh_libid(), L.umi_min_len};
}
__device__ __forceinline__ uint64_t lowmask(uint32_t bits) { return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull); }
typedef uint32_t cr_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream4(const uint4 *p) {  // 16-byte load that does not stay in the caches
    const cr_v4u32 v = __builtin_nontemporal_load(reinterpret_cast<const cr_v4u32 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// CRGPU_OPT_DENSE_BARCODE_KEYS: column of a whitelist rank in the BarcodeIndex, or CRGPU_MISS (DenseIndex::d_fwd)
__device__ __forceinline__ uint32_t dense_column(const uint4 *__restrict__ fwd, uint32_t rank) {
    const uint4 e = fwd[rank >> 6];
    const unsigned long long bits = ((unsigned long long)e.y << 32) | e.x;
    const uint32_t bit = rank & 63u;
    if (!((bits >> bit) & 1ull)) return CRGPU_MISS;
    return e.z + (uint32_t)__popcll(bits & ((1ull << bit) - 1ull));
}

struct DevBuf {  // pooled temporary, returned to the context's pool at scope exit
    crgpu_ctx *ctx = nullptr;
    void *p = nullptr;
    ~DevBuf() { cr_pool_free(ctx, p); }
    template <typename T>
    T *as() { return (T *)p; }
};

static int dmalloc(crgpu_ctx *ctx, DevBuf &b, uint64_t bytes) {
    b.ctx = ctx;
    return cr_pool_alloc(ctx, &b.p, bytes);
}

static int read_u32(crgpu_ctx *ctx, const uint32_t *d, uint32_t *h) {
    return crgpu_memcpy_d2h(ctx, h, d, sizeof(uint32_t));
}

// ------------------------------------------------------------------------------------------------
// build keys (compacting)
// ------------------------------------------------------------------------------------------------
#define KEY_ITEMS 16
#define KEY_BATCH 4  // reads whose loads are issued together: the kernel is bound by load latency, not bandwidth
// a row of UMI qualities as LQW dwords (rows of 4k bytes are dword aligned); LQW = 0: byte path
template <int LQW>
struct __attribute__((aligned(4))) QRow {
    uint32_t w[LQW ? LQW : 1];
};
// ORDERED (the keys travel with their read ordinals: the DupInfo paths): the compaction keeps the read order -- chunks
// are handed out by a ticket counter and get their output offset from a decoupled look-back over `status` (one word
// per chunk: flag in the top two bits, 1 = this chunk's count, 2 = inclusive prefix), so equal keys leave in qname
// order.  The sharded path relies on it: the owner of a barcode sees only the position of a key in its receive buffer.
// Without ORDERED a chunk reserves its space with one atomic and the chunks land in arrival order.
#define BK_AGG (1ull << 62)
#define BK_INC (2ull << 62)
template <int LQW, bool HIST, bool ORDERED = false>
__global__ __launch_bounds__(256) void k_build_keys(const KL kl, const uint32_t *__restrict__ bc_idx,
                                                    const uint32_t *__restrict__ umi, const uint8_t *__restrict__ umi_q,
                                                    const uint32_t *__restrict__ feature, const uint8_t *__restrict__ flags,
                                                    uint64_t n, uint64_t *__restrict__ keys_out,
                                                    uint32_t *__restrict__ vals_out,
                                                    unsigned long long *__restrict__ n_out, const SweepPlan plan,
                                                    uint32_t *__restrict__ ghist, unsigned long long *__restrict__ status,
                                                    uint32_t *__restrict__ ticket, const uint8_t *__restrict__ ulen,
                                                    const uint4 *__restrict__ dense_fwd, uint32_t *__restrict__ n_unknown,
                                                    uint32_t *__restrict__ lb_abort) {
    // lb_abort: the watchdog word of the ORDERED look-back (block_utils.h)
    // dense_fwd (nullable): barcode rank -> column of the BarcodeIndex (CRGPU_OPT_DENSE_BARCODE_KEYS)
    // ulen (nullable, byte path LQW == 0 only): the UMI length of every read, umi_min_len .. umi_len
    __shared__ __attribute__((aligned(8))) uint32_t lds[10];
    __shared__ unsigned long long s_c;  // ORDERED: the chunk of this round, then its output offset
    __shared__ uint32_t s_ws[KEY_ITEMS * 4];  // ORDERED: kept keys of (item slot, wave), then their exclusive prefix
    // ghist != NULL: the digits of every emitted key are counted for all passes of the sort that follows, which
    // then needs no histogram read of its own (k_global_hist)
    __shared__ uint32_t s_hist[HIST ? OS_MAX_PASSES * RADIX_MAX : 1];
    if (HIST) {
        for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += 256) s_hist[x] = 0;
        __syncthreads();
    }
    const uint32_t L = kl.umi_len;
    const uint64_t chunk = 256ull * KEY_ITEMS;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const uint32_t umi_mask = (uint32_t)lowmask(kl.bits_umi), adj_mask = (uint32_t)lowmask(kl.bits_umi - 2u);
    for (uint64_t c = blockIdx.x;; c += gridDim.x) {
      if (ORDERED) {
          if (threadIdx.x == 0) s_c = atomicAdd(ticket, 1u);
          __syncthreads();
          c = s_c;
          __syncthreads();
      }
      if (c >= n_chunks) break;  // uniform
      uint64_t keys[KEY_ITEMS];
      uint32_t mask = 0;
#pragma unroll
      for (int j0 = 0; j0 < KEY_ITEMS; j0 += KEY_BATCH) {
        // every field of KEY_BATCH reads is requested before any is looked at (no load sits behind a branch)
        uint32_t vb[KEY_BATCH], vf[KEY_BATCH], vfl[KEY_BATCH], vu[KEY_BATCH], vl[KEY_BATCH];
        QRow<LQW> vq[KEY_BATCH];
#pragma unroll
        for (int jj = 0; jj < KEY_BATCH; jj++) {
            const uint64_t i = c * chunk + (uint64_t)(j0 + jj) * 256 + threadIdx.x;
            const bool ok = i < n;
            vb[jj] = ok ? bc_idx[i] : CRGPU_MISS;
            vf[jj] = ok ? feature[i] : CRGPU_NO_FEATURE;
            vfl[jj] = (ok && flags) ? flags[i] : 0u;
            vu[jj] = ok ? umi[i] : 0u;
            vl[jj] = (LQW == 0 && ok && ulen) ? ulen[i] : L;
            if (LQW) {
                if (ok) vq[jj] = *reinterpret_cast<const QRow<LQW> *>(umi_q + i * (uint64_t)(4 * LQW));
                else
                    for (int k = 0; k < (LQW ? LQW : 1); k++) vq[jj].w[k] = 0u;
            }
        }
#pragma unroll
        for (int jj = 0; jj < KEY_BATCH; jj++) {
            const int j = j0 + jj;
            const uint64_t i = c * chunk + (uint64_t)j * 256 + threadIdx.x;
            uint32_t b = vb[jj];
            const uint32_t f = vf[jj], fl = vfl[jj];
            const uint32_t lib = fl & CRGPU_FLAG_LIB_MASK;
            if (dense_fwd && b != CRGPU_MISS) {
                b = dense_column(dense_fwd, b);
                if (b == CRGPU_MISS) atomicAdd(n_unknown, 1u);  // a barcode without reads in the tables: the call fails
            }
            bool keep = b != CRGPU_MISS && f != CRGPU_NO_FEATURE && f < kl.n_features && lib < kl.n_libs;
            // this read's UMI length (a length outside umi_min_len .. umi_len: the reference's check_range fails, no UMI)
            const uint32_t Li = LQW == 0 ? vl[jj] : L;
            if (LQW == 0 && (Li < kl.umi_min_len || Li > L)) keep = false;
            const uint32_t Lc = (LQW == 0 && Li >= 1u && Li <= L) ? Li : L;
            const uint32_t u = vu[jj] & (LQW == 0 ? (uint32_t)lowmask(2u * Lc) : umi_mask);
            // UmiInfo::new (umi/src/info.rs:20-37)
            bool has_n = false, low_q = false;
            if (LQW) {
#pragma unroll
                for (int k = 0; k < (LQW ? LQW : 1); k++) {
                    const uint32_t w = vq[jj].w[k];
                    has_n |= (w & 0x80808080u) != 0u;
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) low_q |= (uint8_t)(((w >> (8 * bb)) & 0x7Fu) - 33u) < 10u;
                }
            } else if (keep) {
                for (uint32_t k = 0; k < Lc; k++) {
                    const uint32_t q = umi_q[i * L + k];
                    has_n |= (q & 0x80u) != 0u;
                    low_q |= (uint8_t)((q & 0x7Fu) - 33u) < 10u;  // u8 wrapping subtraction, UMI_MIN_QV = 10
                }
            }
            // is_homopolymer: every adjacent pair equal (true for a 1-base UMI)
            const bool homopolymer = ((u ^ (u >> 2)) & (LQW == 0 ? (uint32_t)lowmask(2u * Lc - 2u) : adj_mask)) == 0u;
            keep = keep && !(has_n || homopolymer || low_q);
            keys[j] = ((uint64_t)b << kl.sh_bc) | ((uint64_t)f << kl.sh_feat) | ((uint64_t)lib << kl.sh_libid) |
                      ((uint64_t)(L - Lc) << kl.sh_lib) | ((uint64_t)u << kl.sh_umi) | ((fl & CRGPU_FLAG_NONTXOMIC) ? 1ull : 0ull);
            if (keep) mask |= 1u << j;
            if (HIST && keep)
                for (uint32_t p = 0; p < plan.n_passes; p++)
                    atomicAdd(&s_hist[p * RADIX_MAX + ((uint32_t)(keys[j] >> plan.shift[p]) & plan.mask[p])], 1u);
        }
      }
      if (ORDERED) {
          // stable inside the chunk, too: output order = (item slot, wave, lane) = ascending read index
          uint32_t below[KEY_ITEMS];
          const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
          for (int j = 0; j < KEY_ITEMS; j++) {
              const unsigned long long m = __ballot((mask >> j) & 1u);
              below[j] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
              if (lane == 0) s_ws[j * 4 + wave] = (uint32_t)__popcll(m);
          }
          __syncthreads();
          if (threadIdx.x < KEY_ITEMS * 4) {  // one wave: exclusive scan of the 64 (slot, wave) counts + the look-back
              const uint32_t v = s_ws[threadIdx.x];
              uint32_t x = v;
#pragma unroll
              for (int d = 1; d < KEY_ITEMS * 4; d <<= 1) {
                  const uint32_t y = __shfl_up(x, d);
                  if (threadIdx.x >= (uint32_t)d) x += y;
              }
              s_ws[threadIdx.x] = x - v;
              const uint32_t total = __shfl(x, 63);
              if (threadIdx.x == 0 && c > 0)
                  __hip_atomic_store(&status[c], BK_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              // every lower ticket is held by a workgroup that is running or done: the chain always moves on
              const unsigned long long excl = wave_lookback(status, c, lb_abort);
              if (threadIdx.x == 0) {
                  if (excl != WAVE_LOOKBACK_ABORTED) {
                      __hip_atomic_store(&status[c], BK_INC | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      if (c + 1 == n_chunks) *n_out = excl + total;
                  }
                  s_c = excl;
              }
          }
          __syncthreads();
          const unsigned long long base = s_c;
          if (base == WAVE_LOOKBACK_ABORTED) break;  // uniform: the host reports the stall (lb_abort is set)
#pragma unroll
          for (int j = 0; j < KEY_ITEMS; j++)
              if (mask & (1u << j)) {
                  const unsigned long long o = base + s_ws[j * 4 + wave] + below[j];
                  keys_out[o] = keys[j];
                  if (vals_out) vals_out[o] = (uint32_t)(c * chunk + (uint64_t)j * 256 + threadIdx.x);  // read ordinal
              }
          __syncthreads();
      } else {
          // one global atomic per 4096-read chunk (same-address atomics saturate near 88 per microsecond); inside a wave
          // the kept keys leave item slot by item slot, so that a store instruction writes consecutive addresses
          const unsigned long long o = block_reserve_256((uint32_t)__popc(mask), n_out, lds);
          unsigned long long wo = __shfl(o, 0);  // the wave's base = the offset of its first lane
          const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
          for (int j = 0; j < KEY_ITEMS; j++) {
              const bool k = (mask >> j) & 1u;
              const unsigned long long m = __ballot(k);
              if (k) {
                  const unsigned long long p = wo + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                  keys_out[p] = keys[j];
                  if (vals_out) vals_out[p] = (uint32_t)(c * chunk + (uint64_t)j * 256 + threadIdx.x);  // read ordinal
              }
              wo += (uint32_t)__popcll(m);
          }
      }
    }
    if (HIST) {
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += 256)
            if (s_hist[x]) atomicAdd(&ghist[x], s_hist[x]);
    }
}

// ---- the same for the common case, four consecutive reads per lane --------------------------------------------------------
// k_build_keys is bound by the number of loads in flight, not by bytes (five loads of 1 to 12 bytes per read).  Here a lane
// takes four consecutive reads with 16-byte loads: barcode ranks, features and UMIs as one uint4 each, the four flag bytes
// as one dword, the four quality rows (4 * LQW dwords) as LQW uint4 -- 1.75 loads per read instead of 5.  Keys only (no
// ordinals, fixed UMI length, dword quality rows), n a multiple of 4, 16-byte aligned arrays; everything else takes
// k_build_keys.  The kept keys leave in thread-major order inside a chunk, as there: the sort does not care.
#ifndef KEY_VB
#define KEY_VB 1
#endif
template <int LQW, bool HIST>
__global__ __launch_bounds__(256) void k_build_keys_v4(const KL kl, const uint32_t *__restrict__ bc_idx, const uint32_t *__restrict__ umi,
                                                       const uint8_t *__restrict__ umi_q, const uint32_t *__restrict__ feature,
                                                       const uint8_t *__restrict__ flags, uint64_t n, uint64_t *__restrict__ keys_out,
                                                       unsigned long long *__restrict__ n_out, const SweepPlan plan,
                                                       uint32_t *__restrict__ ghist, const uint4 *__restrict__ dense_fwd,
                                                       uint32_t *__restrict__ n_unknown) {
    static_assert(LQW >= 1 && LQW <= 4, "dword quality rows");
    __shared__ __attribute__((aligned(8))) uint32_t lds[10];
    __shared__ uint32_t s_hist[HIST ? OS_MAX_PASSES * RADIX_MAX : 1];
    if (HIST) {
        for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += 256) s_hist[x] = 0;
        __syncthreads();
    }
    constexpr int NV = KEY_ITEMS / 4;  // vectors of four reads per thread and chunk
    const uint32_t L = kl.umi_len;
    const uint64_t chunk = 256ull * KEY_ITEMS;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const uint64_t n_vec = n / 4;  // n is a multiple of 4
    const uint32_t umi_mask = (uint32_t)lowmask(kl.bits_umi), adj_mask = (uint32_t)lowmask(kl.bits_umi - 2u);
    const uint4 *__restrict__ bc4 = reinterpret_cast<const uint4 *>(bc_idx);
    const uint4 *__restrict__ ft4 = reinterpret_cast<const uint4 *>(feature);
    const uint4 *__restrict__ um4 = reinterpret_cast<const uint4 *>(umi);
    const uint32_t *__restrict__ fl4 = reinterpret_cast<const uint32_t *>(flags);
    const uint4 *__restrict__ q4 = reinterpret_cast<const uint4 *>(umi_q);
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        uint64_t keys[KEY_ITEMS];
        uint32_t mask = 0;
#pragma unroll
      for (int v0 = 0; v0 < NV; v0 += KEY_VB) {  // KEY_VB vectors' loads are in flight together
        uint4 vb[KEY_VB], vf[KEY_VB], vu[KEY_VB], vq[KEY_VB][LQW];
        uint32_t vfl[KEY_VB];
#pragma unroll
        for (int vv = 0; vv < KEY_VB; vv++) {
            const int v = v0 + vv;
            const uint64_t iv = c * (chunk / 4) + (uint64_t)v * 256 + threadIdx.x;  // index of the vector
            const uint64_t ic = iv < n_vec ? iv : n_vec - 1;                         // loads from a clamped address
            if (dense_fwd) {
                // the rank -> column table has to stay in L2 beside 33 bytes of touch-once input per read: those go past it
                vb[vv] = ld_stream4(bc4 + ic);
                vf[vv] = ld_stream4(ft4 + ic);
                vu[vv] = ld_stream4(um4 + ic);
                vfl[vv] = fl4 ? __builtin_nontemporal_load(fl4 + ic) : 0u;
#pragma unroll
                for (int k = 0; k < LQW; k++) vq[vv][k] = ld_stream4(q4 + ic * LQW + k);
            } else {
                vb[vv] = bc4[ic];
                vf[vv] = ft4[ic];
                vu[vv] = um4[ic];
                vfl[vv] = fl4 ? fl4[ic] : 0u;
#pragma unroll
                for (int k = 0; k < LQW; k++) vq[vv][k] = q4[ic * LQW + k];
            }
        }
#pragma unroll
        for (int vv = 0; vv < KEY_VB; vv++) {
            const int v = v0 + vv;
            const uint64_t iv = c * (chunk / 4) + (uint64_t)v * 256 + threadIdx.x;
            uint32_t b4[4] = {vb[vv].x, vb[vv].y, vb[vv].z, vb[vv].w};
            if (dense_fwd) {  // rank -> column (four independent gathers from a table that stays in L2)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t rk = b4[r];
                    const uint32_t c = dense_column(dense_fwd, rk != CRGPU_MISS ? rk : 0u);
                    if (rk != CRGPU_MISS && c == CRGPU_MISS && iv < n_vec) atomicAdd(n_unknown, 1u);
                    b4[r] = rk != CRGPU_MISS ? c : CRGPU_MISS;
                }
            }
            const uint32_t f4[4] = {vf[vv].x, vf[vv].y, vf[vv].z, vf[vv].w};
            const uint32_t u4[4] = {vu[vv].x, vu[vv].y, vu[vv].z, vu[vv].w};
            uint32_t qw[4 * LQW];
#pragma unroll
            for (int k = 0; k < LQW; k++) {
                qw[4 * k + 0] = vq[vv][k].x;
                qw[4 * k + 1] = vq[vv][k].y;
                qw[4 * k + 2] = vq[vv][k].z;
                qw[4 * k + 3] = vq[vv][k].w;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = v * 4 + r;
                const uint32_t b = b4[r], f = f4[r], fl = (vfl[vv] >> (8 * r)) & 0xFFu;
                const uint32_t lib = fl & CRGPU_FLAG_LIB_MASK;
                bool keep = iv < n_vec && b != CRGPU_MISS && f != CRGPU_NO_FEATURE && f < kl.n_features && lib < kl.n_libs;
                const uint32_t u = u4[r] & umi_mask;
                // UmiInfo::new (umi/src/info.rs:20-37)
                bool has_n = false, low_q = false;
#pragma unroll
                for (int k = 0; k < LQW; k++) {
                    const uint32_t w = qw[r * LQW + k];
                    has_n |= (w & 0x80808080u) != 0u;
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) low_q |= (uint8_t)(((w >> (8 * bb)) & 0x7Fu) - 33u) < 10u;
                }
                const bool homopolymer = ((u ^ (u >> 2)) & adj_mask) == 0u;
                keep = keep && !(has_n || homopolymer || low_q);
                keys[j] = ((uint64_t)b << kl.sh_bc) | ((uint64_t)f << kl.sh_feat) | ((uint64_t)lib << kl.sh_libid) |
                          ((uint64_t)u << kl.sh_umi) | ((fl & CRGPU_FLAG_NONTXOMIC) ? 1ull : 0ull);
                if (keep) mask |= 1u << j;
                if (HIST && keep)
                    for (uint32_t p = 0; p < plan.n_passes; p++)
                        atomicAdd(&s_hist[p * RADIX_MAX + ((uint32_t)(keys[j] >> plan.shift[p]) & plan.mask[p])], 1u);
            }
        }
      }
        // one reservation per workgroup and chunk; inside a wave the kept keys leave item slot by item slot, so that a store
        // instruction writes consecutive addresses (thread-major order made 64 separate 100-byte pieces of every store)
        const unsigned long long o = block_reserve_256((uint32_t)__popc(mask), n_out, lds);
        unsigned long long wo = __shfl(o, 0);  // the wave's base = the offset of its first lane
        const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
        for (int j = 0; j < KEY_ITEMS; j++) {
            const bool k = (mask >> j) & 1u;
            const unsigned long long m = __ballot(k);
            if (k) keys_out[wo + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = keys[j];
            wo += (uint32_t)__popcll(m);
        }
    }
    if (HIST) {
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < plan.n_passes * RADIX_MAX; x += 256)
            if (s_hist[x]) atomicAdd(&ghist[x], s_hist[x]);
    }
    (void)L;
}

static int build_keys_impl(crgpu_ctx *ctx, const crgpu_records *recs, uint64_t *d_keys_out, uint32_t *d_vals_out,
                           uint64_t *n_keys_out) {
    if (!ctx || !recs || !n_keys_out) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->layout.set, CRGPU_ESTATE, "crgpu_build_keys: call crgpu_set_key_layout first");
    CR_REQUIRE(ctx, recs->umi_len == ctx->layout.umi_len, CRGPU_EINVAL, "records umi_len %u != layout umi_len %u",
               recs->umi_len, ctx->layout.umi_len);
    *n_keys_out = 0;
    if (recs->n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, recs->n <= 0x7FFFFFFFull, CRGPU_ERANGE, "crgpu_build_keys: at most 2^31-1 records per call");
    CR_REQUIRE(ctx, recs->d_bc_idx && recs->d_umi && recs->d_umi_qualn && recs->d_feature && d_keys_out, CRGPU_EINVAL,
               "crgpu_build_keys: NULL buffer");
    CR_REQUIRE(ctx, !recs->d_umi_len || ctx->layout.bits_ulen || ctx->layout.umi_min_len == ctx->layout.umi_len, CRGPU_ESTATE,
               "crgpu_build_keys: per-read UMI lengths need crgpu_set_umi_min_len");
    CR_REQUIRE(ctx, (recs->umi_len & 3u) != 0u || (uintptr_t)recs->d_umi_qualn % 4 == 0, CRGPU_EINVAL,
               "crgpu_build_keys: the UMI quality buffer must be 4-byte aligned");
    unsigned long long *d_n = (unsigned long long *)(ctx->d_scalars + 8);
    bool append = false;
    const uint64_t n_before = ctx->ghist.n;
    CR_TRY(cr_dense_ensure(ctx));  // CRGPU_OPT_DENSE_BARCODE_KEYS: the BarcodeIndex of the tables as they stand (else nothing)
    const uint4 *d_fwd = ctx->dense.valid ? ctx->dense.d_fwd : nullptr;
    uint32_t *d_unknown = ctx->d_scalars + 60, *d_lb_abort = ctx->d_scalars + 61;
    {
        CrTimer t(ctx, CRGPU_T_KEYS, recs->n);
        CR_HIP(ctx, hipMemsetAsync(d_n, 0, sizeof(*d_n), ctx->stream));
        CR_HIP(ctx, hipMemsetAsync(d_unknown, 0, 2 * sizeof(uint32_t), ctx->stream));
        const KL kl = make_kl(ctx->layout);
        // keys only (no read ordinals): count the sort's digit histograms on the way (1024 workgroups keep the
        // flush at a few million atomics)
        KeyHistograms &gh = ctx->ghist;
        const bool had = gh.valid;
        gh.valid = false;
        SweepPlan plan;
        uint32_t widths[OS_MAX_PASSES];
        memset(&plan, 0, sizeof(plan));
        uint32_t *d_hist = nullptr;
        if (ctx->trust_buffers && !getenv("CRGPU_NO_KEY_HIST") && cr_sweep_plan(cr_sort_low_bits(ctx->layout.total_bits(), ctx->layout.bits_umi), ctx->layout.total_bits(), &plan, widths)) {
            if (!gh.d_hist) CR_TRY(cr_pool_alloc(ctx, (void **)&gh.d_hist, (size_t)OS_MAX_PASSES * RADIX_MAX * sizeof(uint32_t)));
            d_hist = gh.d_hist;
            // a call that writes its keys right behind the previous call's (the libraries of a well, one after the other, into
            // one buffer) adds to that call's histograms: the sort of the whole buffer then still finds them
            append = had && !d_vals_out && gh.d_keys + gh.n == d_keys_out && memcmp(&gh.plan, &plan, sizeof(plan)) == 0;
            if (!append) CR_HIP(ctx, hipMemsetAsync(d_hist, 0, (size_t)OS_MAX_PASSES * RADIX_MAX * sizeof(uint32_t), ctx->stream));
        }
#ifndef KEY_GRID_HIST
#define KEY_GRID_HIST 1024u
#endif
        const dim3 grid(cr_grid(recs->n, 256, d_hist ? KEY_GRID_HIST : 256u * 8u));
        // with ordinals: order-preserving compaction (tickets + look-back status, one word per 4096-read chunk)
        DevBuf status_b;
        unsigned long long *d_status = nullptr;
        uint32_t *d_ticket = ctx->d_scalars + 44;
        if (d_vals_out) {
            const uint64_t n_chunks = (recs->n + 256ull * KEY_ITEMS - 1) / (256ull * KEY_ITEMS);
            CR_TRY(dmalloc(ctx, status_b, n_chunks * sizeof(unsigned long long)));
            d_status = status_b.as<unsigned long long>();
            CR_HIP(ctx, hipMemsetAsync(d_status, 0, n_chunks * sizeof(unsigned long long), ctx->stream));
            CR_HIP(ctx, hipMemsetAsync(d_ticket, 0, sizeof(uint32_t), ctx->stream));
        }
#define CR_BUILD_KEYS(LQW)                                                                                                  \
    if (d_vals_out && d_hist)                                                                                               \
        hipLaunchKernelGGL((k_build_keys<LQW, true, true>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi, \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, recs->n, d_keys_out, d_vals_out, d_n, plan, d_hist, \
                           d_status, d_ticket, recs->d_umi_len, d_fwd, d_unknown, d_lb_abort);                                          \
    else if (d_vals_out)                                                                                                    \
        hipLaunchKernelGGL((k_build_keys<LQW, false, true>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi, \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, recs->n, d_keys_out, d_vals_out, d_n, plan, d_hist, \
                           d_status, d_ticket, recs->d_umi_len, d_fwd, d_unknown, d_lb_abort);                                          \
    else if (d_hist)                                                                                                        \
        hipLaunchKernelGGL((k_build_keys<LQW, true>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi,     \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, recs->n, d_keys_out, d_vals_out, d_n, plan, d_hist, \
                           d_status, d_ticket, recs->d_umi_len, d_fwd, d_unknown, d_lb_abort);                                          \
    else                                                                                                                    \
        hipLaunchKernelGGL((k_build_keys<LQW, false>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi,    \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, recs->n, d_keys_out, d_vals_out, d_n, plan, d_hist, \
                           d_status, d_ticket, recs->d_umi_len, d_fwd, d_unknown, d_lb_abort)
        // four reads per lane with 16-byte loads where the layout allows it (see k_build_keys_v4); the 1 - 3 reads behind the
        // last multiple of four go through the scalar kernel, appended by the same counter
        const bool aligned16 = ((uintptr_t)recs->d_bc_idx | (uintptr_t)recs->d_umi | (uintptr_t)recs->d_feature |
                                (uintptr_t)recs->d_umi_qualn) % 16 == 0 && (uintptr_t)recs->d_flags % 4 == 0;
        const uint32_t lqw = (recs->umi_len & 3u) == 0u ? recs->umi_len / 4u : 0u;
        if (!d_vals_out && !recs->d_umi_len && lqw >= 1 && lqw <= 4 && aligned16 && recs->n >= 4096 && !getenv("CRGPU_KEYS_SCALAR")) {
            const uint64_t n4 = recs->n & ~3ull;
#define CR_BUILD_KEYS_V4(LQW)                                                                                                  \
    if (d_hist)                                                                                                                \
        hipLaunchKernelGGL((k_build_keys_v4<LQW, true>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi,      \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, n4, d_keys_out, d_n, plan, d_hist, d_fwd, d_unknown); \
    else                                                                                                                       \
        hipLaunchKernelGGL((k_build_keys_v4<LQW, false>), grid, dim3(256), 0, ctx->stream, kl, recs->d_bc_idx, recs->d_umi,     \
                           recs->d_umi_qualn, recs->d_feature, recs->d_flags, n4, d_keys_out, d_n, plan, d_hist, d_fwd, d_unknown)
            switch (lqw) {
                case 1: CR_BUILD_KEYS_V4(1); break;
                case 2: CR_BUILD_KEYS_V4(2); break;
                case 3: CR_BUILD_KEYS_V4(3); break;
                default: CR_BUILD_KEYS_V4(4); break;
            }
#undef CR_BUILD_KEYS_V4
            if (n4 < recs->n) {
                const uint64_t rest = recs->n - n4;
                const uint8_t *tail_flags = recs->d_flags ? recs->d_flags + n4 : nullptr;
                if (d_hist)
                    hipLaunchKernelGGL((k_build_keys<0, true>), dim3(1), dim3(256), 0, ctx->stream, kl, recs->d_bc_idx + n4, recs->d_umi + n4,
                                       recs->d_umi_qualn + n4 * recs->umi_len, recs->d_feature + n4, tail_flags, rest, d_keys_out,
                                       (uint32_t *)nullptr, d_n, plan, d_hist, d_status, d_ticket, (const uint8_t *)nullptr, d_fwd, d_unknown, d_lb_abort);
                else
                    hipLaunchKernelGGL((k_build_keys<0, false>), dim3(1), dim3(256), 0, ctx->stream, kl, recs->d_bc_idx + n4, recs->d_umi + n4,
                                       recs->d_umi_qualn + n4 * recs->umi_len, recs->d_feature + n4, tail_flags, rest, d_keys_out,
                                       (uint32_t *)nullptr, d_n, plan, d_hist, d_status, d_ticket, (const uint8_t *)nullptr, d_fwd, d_unknown, d_lb_abort);
            }
        } else
        switch (recs->d_umi_len ? 0u : recs->umi_len) {  // per-read lengths: the byte path
            case 4: CR_BUILD_KEYS(1); break;
            case 8: CR_BUILD_KEYS(2); break;
            case 12: CR_BUILD_KEYS(3); break;
            case 16: CR_BUILD_KEYS(4); break;
            default: CR_BUILD_KEYS(0); break;
        }
        if (d_hist) {
            if (!append) gh.d_keys = d_keys_out;
            gh.plan = plan;
            gh.valid = true;  // gh.n is filled in below, once the number of keys is known
        }
#undef CR_BUILD_KEYS
        CR_HIP(ctx, hipGetLastError());
    }
    unsigned long long h = 0;
    CR_TRY(crgpu_memcpy_d2h(ctx, &h, d_n, sizeof(h)));
    if (d_vals_out) {
        uint32_t stalled = 0;
        CR_TRY(crgpu_memcpy_d2h(ctx, &stalled, d_lb_abort, sizeof(stalled)));
        if (stalled) {
            ctx->ghist.valid = false;
            return cr_fail(ctx, CRGPU_EHIP, "crgpu_build_keys: the look-back of the order-preserving compaction stalled (watchdog); "
                                            "nothing was hung, the keys of this call are incomplete");
        }
    }
    if (d_fwd) {
        uint32_t unknown = 0;
        CR_TRY(crgpu_memcpy_d2h(ctx, &unknown, d_unknown, sizeof(unknown)));
        if (unknown) {
            ctx->ghist.valid = false;
            return cr_fail(ctx, CRGPU_ESTATE, "crgpu_build_keys: %u records carry a barcode that has no read in the VALID / CORRECTED tables "
                           "(CRGPU_OPT_DENSE_BARCODE_KEYS needs the tables of the whole well before the first key is built)", unknown);
        }
    }
    *n_keys_out = h;
    ctx->ghist.n = append ? n_before + h : h;
    return CRGPU_OK;
}

extern "C" int crgpu_build_keys_dev(crgpu_ctx *ctx, const crgpu_records *recs, uint64_t *d_keys_out,
                                    uint64_t *n_keys_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    return build_keys_impl(ctx, recs, d_keys_out, nullptr, n_keys_out);
}

extern "C" int crgpu_partition_keys_dev(crgpu_ctx *ctx, const uint64_t *d_keys, uint64_t n, uint32_t n_ranks,
                                        const uint32_t *bounds, uint64_t *d_keys_out, uint64_t *counts_out) {
    if (!ctx || !counts_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->layout.set, CRGPU_ESTATE, "crgpu_partition_keys: call crgpu_set_key_layout first");
    CR_REQUIRE(ctx, n == 0 || (d_keys && d_keys_out), CRGPU_EINVAL, "crgpu_partition_keys: NULL buffer");
    cr_invalidate(ctx);
    CR_TRY(cr_dense_ensure(ctx));
    return cr_partition_by_owner(ctx, d_keys, d_keys_out, n, ctx->layout.sh_bc(), n_ranks, bounds, counts_out);
}

// Histogram-balanced owner ranges: bounds_out[0] = 0, bounds_out[n_ranks] = n_canon, and every range holds
// about the same number of reads according to the VALID + CORRECTED tables of all libraries (call it after the
// tables have been all-reduced so that every rank derives the same bounds).
extern "C" int crgpu_balanced_bounds(crgpu_ctx *ctx, uint32_t n_ranks, uint32_t *bounds_out) {
    if (!ctx || !bounds_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_balanced_bounds: no whitelist set");
    CR_REQUIRE(ctx, n_ranks >= 1 && n_ranks <= 256, CRGPU_EINVAL, "crgpu_balanced_bounds: n_ranks must be 1..256");
    const uint32_t W = ctx->n_canon;
    std::vector<uint64_t> tot(W, 0);
    std::vector<uint32_t> tmp(W);
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        if (!ctx->wl[l].set) continue;
        for (int which = 0; which < 2; which++) {
            CR_TRY(crgpu_memcpy_d2h(ctx, tmp.data(), which ? ctx->wl[l].d_corrected : ctx->wl[l].d_valid, sizeof(uint32_t) * W));
            for (uint32_t r = 0; r < W; r++) tot[r] += tmp[r];
        }
    }
    uint64_t total = 0;
    for (uint32_t r = 0; r < W; r++) total += tot[r];
    bounds_out[0] = 0;
    uint64_t acc = 0;
    uint32_t r = 0;
    for (uint32_t k = 1; k < n_ranks; k++) {
        const uint64_t want = total * k / n_ranks;
        while (r < W && acc + tot[r] <= want) acc += tot[r++];
        bounds_out[k] = r;
    }
    bounds_out[n_ranks] = W;
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// generic two-pass stream compaction driven by a flag functor: out position of every flagged item
// ------------------------------------------------------------------------------------------------
#define CP_BLOCK 256
#ifndef CP_ITEMS
#define CP_ITEMS 8  // items per thread per round: their flag loads are all issued before the first compare
#endif
#define CP_ROUND (CP_BLOCK * CP_ITEMS)
#define CP_WAVES (CP_BLOCK / 64)

// Flags are evaluated at clamped indices and masked afterwards, so that the loads of a round are not chained
// behind `i < hi` branches (one load in flight per wave left these passes at ~2 TB/s).
template <typename Flag>
__global__ __launch_bounds__(CP_BLOCK) void k_cp_count(Flag flag, uint64_t n, uint64_t tile, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t ws[CP_WAVES];
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < n ? lo + tile : n;
    uint32_t c = 0;
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        bool f[CP_ITEMS];
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * CP_BLOCK + threadIdx.x;
            f[j] = flag(i < hi ? i : hi - 1);
        }
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * CP_BLOCK + threadIdx.x;
            c += (f[j] && i < hi) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < CP_WAVES; w++) t += ws[w];
        block_counts[blockIdx.x] = t;
    }
}

// Stable: inside a round the output order is (item slot, wave, lane) == ascending input index.
template <typename Flag, typename Emit>
__global__ __launch_bounds__(CP_BLOCK) void k_cp_write(Flag flag, Emit emit, uint64_t n, uint64_t tile,
                                                       const uint32_t *__restrict__ block_offs) {
    __shared__ uint32_t ws[CP_ITEMS * CP_WAVES];  // flagged items of (item slot, wave), then their exclusive prefix
    __shared__ uint32_t round_total;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < n ? lo + tile : n;
    uint32_t run = block_offs[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        bool f[CP_ITEMS];
        typename Emit::Pre pre[CP_ITEMS];  // what the emit needs from memory, requested together with the flags
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * CP_BLOCK + threadIdx.x;
            f[j] = flag(i < hi ? i : hi - 1);
            pre[j] = emit.pre(i < hi ? i : hi - 1);
        }
        uint32_t below[CP_ITEMS];
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * CP_BLOCK + threadIdx.x;
            f[j] = f[j] && i < hi;
            const unsigned long long m = __ballot(f[j]);
            below[j] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) ws[j * CP_WAVES + wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < CP_ITEMS * CP_WAVES) {  // 32 lanes of wave 0: exclusive scan in (slot, wave) order
            const uint32_t v = ws[threadIdx.x];
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < CP_ITEMS * CP_WAVES; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (threadIdx.x >= (uint32_t)d) x += y;
            }
            ws[threadIdx.x] = x - v;
            if (threadIdx.x == CP_ITEMS * CP_WAVES - 1) round_total = x;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = base + (uint64_t)j * CP_BLOCK + threadIdx.x;
            if (f[j]) emit(i, run + ws[j * CP_WAVES + wave] + below[j], pre[j]);
        }
        run += round_total;
        __syncthreads();
    }
}

static uint32_t cp_blocks(uint64_t n, uint64_t *tile_out) {
    uint64_t nb = (n + CP_ROUND * 4 - 1) / (CP_ROUND * 4);
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    uint64_t tile = (n + nb - 1) / nb;
    tile = (tile + CP_ROUND - 1) / CP_ROUND * CP_ROUND;
    nb = (n + tile - 1) / tile;
    if (nb < 1) nb = 1;
    *tile_out = tile;
    return (uint32_t)nb;
}

// d_block: workspace of >= 4096 u32.  *total_out (device u32) receives the number of flagged items.
template <typename Flag, typename Emit>
static int compact(crgpu_ctx *ctx, Flag flag, Emit emit, uint64_t n, uint32_t *d_block, uint32_t *d_total_out) {
    uint64_t tile;
    const uint32_t nb = cp_blocks(n, &tile);
    hipLaunchKernelGGL(k_cp_count<Flag>, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag, n, tile, d_block);
    CR_TRY(cr_scan_small(ctx, d_block, nb, d_total_out));
    hipLaunchKernelGGL((k_cp_write<Flag, Emit>), dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag, emit, n, tile, d_block);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ---- per-key state after UMI correction ---------------------------------------------------------------------------------
// st[k] (16 bits, zeroed before the kernels run): bit 0 = key k is corrected away (corr[k] holds its target -- corr is
// written ONLY for such keys and must not be read otherwise), bit 1 = low support, bits 2.. = inc1 = number of keys
// corrected onto k (phase 1 of BarcodeDupMarker::new moves one read of each, mark_dups.rs:228-232; at most 3 L).
// inc_all[k] (zeroed) = reads of all keys corrected onto k (phases 1 + 2, :242-246); meaningful where inc1 > 0.
// Corrections are rare (a few per cent of the keys), so everything here is a scattered atomic on a zeroed array and the
// common key writes nothing; the streaming passes afterwards read 2 bytes per key instead of four arrays.
#define ST_CORRECTED 1u
#define ST_LOW 2u
__device__ __forceinline__ uint32_t st_inc1(uint32_t s) { return s >> 2; }
__device__ __forceinline__ void st_or(uint16_t *st, uint64_t k, uint32_t bits) {
    atomicOr(reinterpret_cast<uint32_t *>(st) + (k >> 1), bits << (16u * (uint32_t)(k & 1u)));
}
// minidx[K] (set to all ones) = the smallest raw key corrected onto K.  The representative-read rule (mark_dups.rs:248-268)
// takes the lexicographically smallest raw UMI R corrected onto K among those with (R < K or K itself corrected away); R and K
// lie in one (barcode, feature, library) segment where the distinct keys are sorted by UMI, so UMI order is index order, and
// the smallest qualifying R is the smallest R of all whenever one qualifies: K corrected away -> every R qualifies; otherwise an
// R < K exists exactly when the smallest R is below K.  The readers apply that test (rep_source); a separate pass over all
// distinct keys (k_rep_utype, 0.75 ms per 1 B reads on the critical path) used to apply it before the atomicMin.
__device__ __forceinline__ void move_reads(uint32_t *corr, uint16_t *st, uint32_t *inc_all, uint32_t *minidx, uint64_t k,
                                           uint32_t target, uint32_t my_cnt) {
    corr[k] = target;
    st_or(st, k, ST_CORRECTED);
    atomicAdd(reinterpret_cast<uint32_t *>(st) + (target >> 1), 4u << (16u * (target & 1u)));  // inc1[target] += 1
    atomicAdd(&inc_all[target], my_cnt);
    atomicMin(&minidx[target], (uint32_t)k);
}
// the raw key whose representative read stands for target K (state word sK), or NONE32: K's own reads do
__device__ __forceinline__ uint32_t rep_source(uint32_t min_raw, uint32_t K, uint32_t sK) {
    return (min_raw != 0xFFFFFFFFu && (min_raw < K || (sK & ST_CORRECTED))) ? min_raw : 0xFFFFFFFFu;
}

// ---- functors --------------------------------------------------------------------------------------
struct HeadFlag {  // first element of a run of equal (key >> shift)
    const uint64_t *keys;
    uint32_t shift;
    __device__ __forceinline__ bool operator()(uint64_t i) const {
        const uint64_t prev = keys[i ? i - 1 : 0];  // no load behind a branch
        return (i == 0) | ((keys[i] >> shift) != (prev >> shift));
    }
};
struct EmitRun {  // distinct key + start position of its run
    const uint64_t *keys;
    uint64_t *ukey;
    uint32_t *upos;
    typedef uint64_t Pre;
    __device__ __forceinline__ Pre pre(uint64_t i) const { return keys[i]; }
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t o, Pre key) const {
        ukey[o] = key;
        upos[o] = (uint32_t)i;
    }
};
// targeted-panel filter (mark_dups.rs:311-320): a corrected key of an on-target feature whose read count stays below
// the threshold yields no UmiCount (and its reads carry is_filtered_target_umi); min_reads == 0: None
struct TargetFilter {
    const uint8_t *on_target;
    uint32_t n_features, sh_feat, bits_feat;
    uint64_t min_reads;
    __device__ __forceinline__ bool filtered(uint64_t key, uint32_t read_count, bool low) const {
        if (!min_reads) return false;
        const uint32_t f = (uint32_t)((key >> sh_feat) & lowmask(bits_feat));
        return f < n_features && on_target[f] != 0 && (uint64_t)read_count < min_reads && !low;
    }
};
struct MolFlag {  // distinct key that yields a UmiCount (mark_dups.rs:322-325 with rate 1.0)
    const uint16_t *st;  // per-key state word
    __device__ __forceinline__ bool operator()(uint64_t k) const {
        const uint32_t s = st[k];
        const bool landed = !(s & ST_CORRECTED) | (st_inc1(s) > 0u);  // some read's corrected key is k
        return landed & !(s & ST_LOW);
    }
};
struct MolFlagTargeted {  // the same with the targeted-panel filter: needs the key's final read count
    const uint16_t *st;
    const uint64_t *ukey;
    const uint32_t *upos, *inc_all;
    uint64_t n_keys, n_dist;
    TargetFilter tf;
    __device__ __forceinline__ bool operator()(uint64_t k) const {
        const uint32_t s = st[k];
        const bool landed = !(s & ST_CORRECTED) | (st_inc1(s) > 0u);
        if (!landed || (s & ST_LOW)) return false;
        const uint32_t end = k + 1 < n_dist ? upos[k + 1] : (uint32_t)n_keys;
        const uint32_t rc = ((s & ST_CORRECTED) ? 0u : end - upos[k]) + (st_inc1(s) ? inc_all[k] : 0u);
        return !tf.filtered(ukey[k], rc, false);
    }
};
struct EmitMol {
    const uint64_t *ukey;
    const uint32_t *upos, *inc_all, *minidx;
    const uint16_t *st;
    uint64_t n_keys, n_dist;
    uint64_t *mkeys;
    uint32_t *mreads;
    // UmiCount::probe_idx (mark_dups.rs:332-342): the probe of the representative read, when the records carry probes
    const uint32_t *rep_read = nullptr;
    const int32_t *probe = nullptr;
    int32_t *mprobe = nullptr;
    struct Pre {
        uint64_t key;
        uint32_t p0, p1, s;
    };
    __device__ __forceinline__ Pre pre(uint64_t k) const {
        Pre p;
        p.key = ukey[k];
        p.p0 = upos[k];
        p.p1 = upos[k + 1 < n_dist ? k + 1 : k];
        p.s = st[k];
        return p;
    }
    // the same without the next key's run start (the caller takes it from the neighbouring lane)
    __device__ __forceinline__ Pre pre_own(uint64_t k) const {
        Pre p;
        p.key = ukey[k];
        p.p0 = upos[k];
        p.p1 = 0u;
        p.s = st[k];
        return p;
    }
    __device__ __forceinline__ void operator()(uint64_t k, uint32_t o, const Pre &p) const {
        // Only a key that other keys were corrected onto (a few per cent) has anything in inc_all / minidx.
        // UmiType of the representative read (mark_dups.rs:250-268,326-329): the min (utype, qname)
        // read of the smallest qualifying raw UMI corrected onto k, else of k itself.
        uint32_t inc = 0u, mi = NONE32;
        if (st_inc1(p.s)) {
            inc = inc_all[k];
            mi = rep_source(minidx[k], (uint32_t)k, p.s);
        }
        const uint64_t bit = (mi != NONE32 ? ukey[mi] : p.key) & 1ull;
        mkeys[o] = (p.key & ~1ull) | bit;
        const uint32_t end = k + 1 < n_dist ? p.p1 : (uint32_t)n_keys;
        const uint32_t cnt = end - p.p0;
        // umigene_counts after both moves (mark_dups.rs:226-246): own reads stay only if not corrected away
        mreads[o] = ((p.s & ST_CORRECTED) ? 0u : cnt) + inc;
        if (mprobe) mprobe[o] = probe[rep_read[mi != NONE32 ? mi : (uint32_t)k]];
    }
};
struct EmitTriplet {
    const uint64_t *mkeys;
    uint32_t *tpos;
    struct Pre {};
    __device__ __forceinline__ Pre pre(uint64_t) const { return Pre(); }
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t o, Pre) const { tpos[o] = (uint32_t)i; }
};

// ---- run lengths of the sorted keys (DupBuilder::observe): distinct keys + the start of every run -------------------------
// The generic compaction evaluates HeadFlag with two loads per key (keys[i], keys[i - 1]) and EmitRun with a third; these
// passes are bound by load instructions in flight, not by bytes.  With the wave-blocked layout (a wave owns 64 * CP_ITEMS
// consecutive keys of a round) the left neighbour comes from the lane below or from the previous item slot's lane 63, and
// only a wave's first key of a round looks at memory: one load per key.
__device__ __forceinline__ bool rl_is_head(uint64_t key, uint64_t &carry_key, bool &carry_valid, uint32_t shift, uint32_t lane) {
    uint64_t left = __shfl_up(key, 1);
    if (lane == 0) left = carry_key;
    const bool head = (lane == 0 && !carry_valid) || (key >> shift) != (left >> shift);
    carry_key = __shfl(key, 63);
    carry_valid = true;
    return head;
}

__global__ __launch_bounds__(CP_BLOCK) void k_rl_count(const uint64_t *__restrict__ keys, const uint32_t shift, const uint64_t n,
                                                       const uint64_t tile, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t ws[CP_WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < n ? lo + tile : n;
    uint32_t c = 0;
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        uint64_t key[CP_ITEMS];
        const uint64_t w0 = base + (uint64_t)wave * (64u * CP_ITEMS);
        const uint64_t wb = w0 > 0 ? (w0 - 1 < n ? w0 - 1 : n - 1) : 0;
        uint64_t carry_key = keys[wb];
        bool carry_valid = w0 > 0;
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            key[j] = keys[i < hi ? i : hi - 1];
        }
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            const bool head = rl_is_head(key[j], carry_key, carry_valid, shift, lane);
            c += (head && i < hi) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) ws[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < CP_WAVES; w++) t += ws[w];
        block_counts[blockIdx.x] = t;
    }
}

// z_st / z_inc / z_min (nullable, together): the per-key state of the UMI correction starts out here -- 0, 0 and all ones for
// every distinct key written -- instead of in three memsets on the critical path behind this kernel
__global__ __launch_bounds__(CP_BLOCK) void k_rl_write(const uint64_t *__restrict__ keys, const uint32_t shift, const uint64_t n,
                                                       const uint64_t tile, const uint32_t *__restrict__ block_offs,
                                                       uint64_t *__restrict__ ukey, uint32_t *__restrict__ upos,
                                                       uint16_t *__restrict__ z_st, uint32_t *__restrict__ z_inc,
                                                       uint32_t *__restrict__ z_min) {
    __shared__ uint32_t ws[CP_ITEMS * CP_WAVES];  // heads of (wave, item slot), then their exclusive prefix
    __shared__ uint32_t round_total;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < n ? lo + tile : n;
    uint32_t run = block_offs[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        uint64_t key[CP_ITEMS];
        bool f[CP_ITEMS];
        uint32_t below[CP_ITEMS];
        const uint64_t w0 = base + (uint64_t)wave * (64u * CP_ITEMS);
        const uint64_t wb = w0 > 0 ? (w0 - 1 < n ? w0 - 1 : n - 1) : 0;
        uint64_t carry_key = keys[wb];
        bool carry_valid = w0 > 0;
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            key[j] = keys[i < hi ? i : hi - 1];
        }
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            f[j] = rl_is_head(key[j], carry_key, carry_valid, shift, lane) && i < hi;
            const unsigned long long m = __ballot(f[j]);
            below[j] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) ws[wave * CP_ITEMS + j] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < CP_ITEMS * CP_WAVES) {  // 32 lanes of wave 0: exclusive scan in (wave, slot) order = key order
            const uint32_t v = ws[threadIdx.x];
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < CP_ITEMS * CP_WAVES; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (threadIdx.x >= (uint32_t)d) x += y;
            }
            ws[threadIdx.x] = x - v;
            if (threadIdx.x == CP_ITEMS * CP_WAVES - 1) round_total = x;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++)
            if (f[j]) {
                const uint32_t o = run + ws[wave * CP_ITEMS + j] + below[j];
                ukey[o] = key[j];
                upos[o] = (uint32_t)(w0 + (uint64_t)j * 64 + lane);
                if (z_st) {
                    z_st[o] = 0;
                    z_inc[o] = 0u;
                    z_min[o] = 0xFFFFFFFFu;
                }
            }
        run += round_total;
        __syncthreads();
    }
}

static int run_lengths(crgpu_ctx *ctx, const uint64_t *keys, uint32_t shift, uint64_t n, uint64_t *ukey, uint32_t *upos,
                       uint32_t *d_block, uint32_t *d_total_out, uint16_t *z_st = nullptr, uint32_t *z_inc = nullptr,
                       uint32_t *z_min = nullptr) {
    uint64_t tile;
    const uint32_t nb = cp_blocks(n, &tile);
    hipLaunchKernelGGL(k_rl_count, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, keys, shift, n, tile, d_block);
    CR_TRY(cr_scan_small(ctx, d_block, nb, d_total_out));
    hipLaunchKernelGGL(k_rl_write, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, keys, shift, n, tile, d_block, ukey, upos, z_st, z_inc,
                       z_min);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ---- molecules and (barcode, feature) triplets from the same two launches over the distinct keys ---------------------
// The molecule compaction (count + write over st / ukey / upos) was followed by a second compaction over the molecule keys
// just to find the first molecule of every (barcode, feature) and by k_triplets: two more reads of the 3.2 GB molecule table
// per 1 B records.  Here the count launch counts molecules AND the molecules that open a triplet, and the write launch
// emits both (block offsets from two small scans), so both outputs keep the key order.  A molecule opens a triplet when no
// earlier molecule shares its (barcode, feature): inside a wave that is ballot arithmetic on the lanes' own keys, from one
// item slot to the next a wave-uniform carry in registers, and for a wave's first slot of a round one walk back over the
// distinct keys that yield no molecule (corrected away / low support), rarely more than one step.
// (A single launch with tickets and a decoupled look-back carrying both counts in one status word was measured first: with
// 2048-key chunks +2.5 ms, with 4096-key chunks of one 1024-thread workgroup per CU equal to the separate passes -- the
// same-address ticket atomics and one look-back per chunk cost what the saved reads gain; A/B in profiles/r02_mol_trip_ab.txt.)
// Wave-blocked layout: in a round a wave owns 64 * CP_ITEMS consecutive keys and item slot j is the j-th run of 64 of them,
// so what lies in front of slot j > 0 is slot j - 1 of the same wave (registers) and only slot 0 looks at memory.
struct MtCarry {       // wave-uniform: the run of equal (barcode, feature) that ends in front of the current slot
    uint64_t prefix;   // its prefix
    bool has_mol;      // it holds a molecule
    bool valid;        // there is something in front at all
};

// what lies in front of a wave's first slot of a round: the key before i0 and, if it yields no molecule, the keys before it
template <typename Flag>
__device__ __forceinline__ MtCarry mt_carry_from_memory(const Flag &flag, const uint64_t *__restrict__ ukey, uint32_t sh_feat,
                                                        uint64_t nd, uint64_t i0) {
    MtCarry c{0ull, false, false};
    if (i0 == 0 || i0 >= nd) return c;
    c.valid = true;
    uint64_t p = i0 - 1;
    c.prefix = ukey[p] >> sh_feat;
    c.has_mol = flag(p);
    while (!c.has_mol && p > 0) {
        p--;
        if ((ukey[p] >> sh_feat) != c.prefix) break;
        c.has_mol = flag(p);
    }
    return c;
}

// does this lane's molecule open a triplet?  Updates the carry to describe the run that ends with lane 63.
__device__ __forceinline__ bool mt_opens_triplet(bool is_mol, uint64_t key, uint32_t sh_feat, uint32_t lane, MtCarry &carry) {
    const uint64_t prefix = key >> sh_feat;
    const uint64_t left = __shfl_up(prefix, 1);
    const unsigned long long starts = __ballot(lane == 0 || prefix != left), mols = __ballot(is_mol);
    const unsigned long long upto = ~0ull >> (63u - lane);                       // lanes 0 .. lane
    const uint32_t rs = 63u - (uint32_t)__clzll((long long)(starts & upto));     // first lane of this lane's run
    const unsigned long long between = (upto >> 1) & ~((1ull << rs) - 1ull);     // lanes rs .. lane - 1
    const uint64_t prefix0 = __shfl(prefix, 0);
    const bool carried = carry.valid && carry.prefix == prefix0 && carry.has_mol;  // a molecule of lane 0's run lies in front
    const bool open = (mols & between) == 0ull && !(rs == 0u && carried);
    // the run that ends with lane 63
    const uint32_t rs_last = 63u - (uint32_t)__clzll((long long)starts);
    const bool last_has = (mols >> rs_last) != 0ull || (rs_last == 0u && carried);
    carry.prefix = __shfl(prefix, 63);
    carry.has_mol = last_has;
    carry.valid = true;
    return is_mol && open;
}

#define MT_WAVE_SPAN (64u * CP_ITEMS)
template <typename Flag>
__global__ __launch_bounds__(CP_BLOCK) void k_mt_count(const Flag flag, const uint64_t *__restrict__ ukey, const uint32_t sh_feat,
                                                       const uint64_t nd, const uint64_t tile, uint32_t *__restrict__ mol_counts,
                                                       uint32_t *__restrict__ head_counts) {
    __shared__ uint32_t ws[2 * CP_WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < nd ? lo + tile : nd;
    uint32_t cm = 0, ch = 0;
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        bool f[CP_ITEMS];
        uint64_t key[CP_ITEMS];
        const uint64_t w0 = base + (uint64_t)wave * MT_WAVE_SPAN;
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            const uint64_t ic = i < hi ? i : hi - 1;
            f[j] = flag(ic);
            key[j] = ukey[ic];
        }
        MtCarry carry = mt_carry_from_memory(flag, ukey, sh_feat, nd, w0);
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            const bool m = f[j] && i < hi;
            const bool h = mt_opens_triplet(m, key[j], sh_feat, lane, carry);
            cm += m ? 1u : 0u;
            ch += h ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        cm += __shfl_xor(cm, d);
        ch += __shfl_xor(ch, d);
    }
    if (lane == 0) {
        ws[wave] = cm;
        ws[CP_WAVES + wave] = ch;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tm = 0, th = 0;
        for (int w = 0; w < CP_WAVES; w++) {
            tm += ws[w];
            th += ws[CP_WAVES + w];
        }
        mol_counts[blockIdx.x] = tm;
        head_counts[blockIdx.x] = th;
    }
}

template <typename Flag>
__global__ __launch_bounds__(CP_BLOCK) void k_mt_write(const Flag flag, const EmitMol emit, const uint32_t sh_bc, const uint32_t sh_feat,
                                                       const uint32_t bits_feat, const uint64_t nd, const uint64_t tile,
                                                       const uint32_t *__restrict__ mol_offs, const uint32_t *__restrict__ head_offs,
                                                       uint32_t *__restrict__ tbc, uint32_t *__restrict__ tfeat,
                                                       uint32_t *__restrict__ tpos, const uint32_t *__restrict__ back) {
    // back (nullable): the keys' barcode field is a column of the BarcodeIndex; the triplets report whitelist ranks
    __shared__ uint32_t ws[CP_ITEMS * CP_WAVES];  // (molecules | heads << 16) of (wave, item slot), then their exclusive prefix
    __shared__ uint32_t round_total;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile;
    const uint64_t hi = lo + tile < nd ? lo + tile : nd;
    uint32_t run_m = mol_offs[blockIdx.x], run_h = head_offs[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += CP_ROUND) {
        bool f[CP_ITEMS], h[CP_ITEMS];
        EmitMol::Pre pre[CP_ITEMS];
        const uint64_t w0 = base + (uint64_t)wave * MT_WAVE_SPAN;
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            const uint64_t ic = i < hi ? i : hi - 1;
            f[j] = flag(ic);
            pre[j] = emit.pre_own(ic);
        }
        {
            // run start of the NEXT key: the lane above, the next slot's lane 0, and memory only behind the wave's last key
            const uint64_t il = w0 + (uint64_t)MT_WAVE_SPAN;  // the key behind this wave's span
            uint32_t after = emit.upos[il < nd ? il : nd - 1];
#pragma unroll
            for (int j = CP_ITEMS - 1; j >= 0; j--) {
                uint32_t nx = __shfl_down(pre[j].p0, 1);
                if (lane == 63u) nx = after;
                pre[j].p1 = nx;
                after = __shfl(pre[j].p0, 0);
            }
            // keys clamped to hi - 1 (the tail of the last round) repeat one key: their p1 is never used (f is false there),
            // and the last real key of the array takes n_keys inside emit()
        }
        MtCarry carry = mt_carry_from_memory(flag, emit.ukey, sh_feat, nd, w0);
        uint32_t below_m[CP_ITEMS], below_h[CP_ITEMS];
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++) {
            const uint64_t i = w0 + (uint64_t)j * 64 + lane;
            f[j] = f[j] && i < hi;
            h[j] = mt_opens_triplet(f[j], pre[j].key, sh_feat, lane, carry);
            const unsigned long long mm = __ballot(f[j]), mh = __ballot(h[j]);
            const unsigned long long lt = (1ull << lane) - 1ull;
            below_m[j] = (uint32_t)__popcll(mm & lt);
            below_h[j] = (uint32_t)__popcll(mh & lt);
            if (lane == 0) ws[wave * CP_ITEMS + j] = (uint32_t)__popcll(mm) | ((uint32_t)__popcll(mh) << 16);
        }
        __syncthreads();
        if (threadIdx.x < CP_ITEMS * CP_WAVES) {  // 32 lanes of wave 0: exclusive scan in (wave, slot) order = key order
            const uint32_t v = ws[threadIdx.x];
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < CP_ITEMS * CP_WAVES; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (threadIdx.x >= (uint32_t)d) x += y;
            }
            ws[threadIdx.x] = x - v;
            if (threadIdx.x == CP_ITEMS * CP_WAVES - 1) round_total = x;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CP_ITEMS; j++)
            if (f[j]) {
                const uint64_t i = w0 + (uint64_t)j * 64 + lane;
                const uint32_t w = ws[wave * CP_ITEMS + j];
                const uint32_t o = run_m + (w & 0xFFFFu) + below_m[j];
                emit(i, o, pre[j]);
                if (h[j]) {
                    const uint32_t t = run_h + (w >> 16) + below_h[j];
                    const uint32_t bcf = (uint32_t)(pre[j].key >> sh_bc);
                    tbc[t] = back ? back[bcf] : bcf;
                    tfeat[t] = (uint32_t)((pre[j].key >> sh_feat) & lowmask(bits_feat));
                    tpos[t] = o;
                }
            }
        run_m += round_total & 0xFFFFu;
        run_h += round_total >> 16;
        __syncthreads();
    }
}

// molecules per triplet = distance to the next triplet's first molecule (types.rs:180-188)
__global__ __launch_bounds__(256) void k_trip_counts(const uint32_t *__restrict__ tpos, uint64_t nt, uint64_t nm,
                                                     uint32_t *__restrict__ cnt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += stride)
        cnt[t] = (t + 1 < nt ? tpos[t + 1] : (uint32_t)nm) - tpos[t];
}

// ------------------------------------------------------------------------------------------------
// UMI correction (correct_umis, mark_dups.rs:19-59)
// ------------------------------------------------------------------------------------------------
// segment [s, e) of distinct keys sharing (barcode, feature, library) with key k: galloping search
__device__ __forceinline__ void segment_bounds(const uint64_t *__restrict__ ukey, uint64_t nd, uint64_t k, uint32_t shift,
                                               uint64_t &s, uint64_t &e) {
    const uint64_t pre = ukey[k] >> shift;
    // backward
    uint64_t lo = k, step = 1;
    while (lo >= step && (ukey[lo - step] >> shift) == pre) {
        lo -= step;
        step <<= 1;
    }
    // first index with prefix == pre lies in (lo - step, lo]  (or [0, lo])
    uint64_t a = lo >= step ? lo - step + 1 : 0, b = lo;
    while (a < b) {
        const uint64_t mid = (a + b) >> 1;
        if ((ukey[mid] >> shift) == pre) b = mid; else a = mid + 1;
    }
    s = a;
    // forward
    uint64_t hi = k;
    step = 1;
    while (hi + step < nd && (ukey[hi + step] >> shift) == pre) {
        hi += step;
        step <<= 1;
    }
    a = hi;
    b = hi + step < nd ? hi + step - 1 : nd - 1;  // last index with prefix == pre lies in [hi, b]
    while (a < b) {
        const uint64_t mid = (a + b + 1) >> 1;
        if ((ukey[mid] >> shift) == pre) a = mid; else b = mid - 1;
    }
    e = a + 1;
}

__device__ __forceinline__ uint32_t run_count(const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys, uint64_t k) {
    const uint32_t end = k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys;
    return end - upos[k];
}

#include "umi_correct.h"

// ------------------------------------------------------------------------------------------------
// low-support filter (determine_low_support_umigenes, mark_dups.rs:87-108)
// ------------------------------------------------------------------------------------------------
// The rule groups the keys of one (barcode, library, UMI) across features.  Instead of re-sorting the
// 64-bit keys in [barcode][library][UMI][feature] order (8 passes over 12-byte pairs) the distinct
// keys are sorted by a 32-bit hash of (barcode, library, UMI) with their index as payload (4 passes
// over 8-byte pairs); members of a group are then adjacent inside a run of equal hashes, and every
// comparison below re-checks the exact (barcode, library, UMI) so hash collisions cannot merge groups.
__device__ __forceinline__ uint64_t group_id(const KL &kl, uint64_t key) {  // (barcode, library, UMI), exact
    const uint64_t umi = (key >> kl.sh_umi) & lowmask(kl.bits_umi);
    const uint64_t lib = (key >> kl.sh_lib) & lowmask(kl.bits_lib + kl.bits_ulen);  // library and UMI length: both tell UmiSeqs apart
    const uint64_t bc = key >> kl.sh_bc;
    return ((bc << (kl.bits_lib + kl.bits_ulen)) | lib) << kl.bits_umi | umi;
}
__device__ __forceinline__ uint32_t group_hash(uint64_t g) {
    g ^= g >> 33;
    g *= 0xff51afd7ed558ccdull;
    g ^= g >> 33;
    g *= 0xc4ceb9fe1a85ec53ull;
    g ^= g >> 33;
    return (uint32_t)g;
}

// ---- candidate filter ---------------------------------------------------------------------------
// Only a (barcode, library, UMI) group with at least two member keys (the UMI seen under two features)
// can hold low-support keys, and such groups are rare (a few per cent of the keys).  Sorting every key by
// group just to find them cost more than the whole UMI correction, so a filter runs first: the distinct
// keys are sorted by barcode, so a workgroup takes a range of WHOLE barcodes (the barcodes whose first key
// lies in its tile of LF_TILE keys), hashes each key's group into a two-bit-per-entry LDS table ("seen",
// "seen again") and marks as candidates the keys whose entry was hit twice: every member of a multi-member
// group is marked (no false negatives); false positives are removed by the exact comparison later.
#define LF_TILE 8192u
#define LF_ENTRY_BITS 18u               // 2^18 two-bit entries = 64 KB of LDS, two workgroups per CU
#define LF_WORDS (1u << (LF_ENTRY_BITS - 4u))
__device__ __forceinline__ uint64_t mix64(uint64_t g) {
    g ^= g >> 33;
    g *= 0xff51afd7ed558ccdull;
    g ^= g >> 33;
    g *= 0xc4ceb9fe1a85ec53ull;
    g ^= g >> 33;
    return g;
}
#define LF_THREADS 1024u  // a workgroup may own one giant barcode: many threads + batched loads keep that pole short
#define LF_BATCH 4
// EMIT (CRGPU_CAND_EMIT=1; measured slower, see the driver): instead of a flag per key -- which a count + write compaction then turned into the list of
// (hash, index) pairs with two more reads of the flags and one of the keys -- the second pass writes the pairs of its
// candidates itself: a workgroup counts them, reserves its stretch of the list with ONE atomic per tile and appends.  The
// list is sorted by hash afterwards, so the order of the tiles in it does not matter.
#define LS_HASH_BITS 32u
struct CandEmit {
    uint32_t *hash, *val;       // the list (room for every key)
    unsigned long long *n_out;  // its length (device counter, zeroed by the host)
    uint32_t vbits;
};
// entry e of the table: bit 0 "seen", bit 1 "seen again"
__device__ __forceinline__ void lf_note(uint32_t *bm, uint32_t e) {
    const uint32_t sh = (e & 15u) * 2u;
    const uint32_t old = atomicOr(&bm[e >> 4], 1u << sh);
    if ((old >> sh) & 1u) atomicOr(&bm[e >> 4], 2u << sh);
}
// two: TWO entries per group (bits 0.. and 24.. of the mix; opt-in, see the driver): a group with two members has hit both of
// its entries twice, a single key is marked only if other keys hit BOTH of its entries
__device__ __forceinline__ bool lf_again(const uint32_t *bm, uint64_t mx, uint32_t emask, bool two) {
    const uint32_t e1 = (uint32_t)mx & emask, e2 = (uint32_t)(mx >> 24) & emask;
    const bool a1 = (bm[e1 >> 4] >> ((e1 & 15u) * 2u + 1u)) & 1u;
    return two ? a1 && ((bm[e2 >> 4] >> ((e2 & 15u) * 2u + 1u)) & 1u) : a1;
}
template <bool EMIT>
__global__ __launch_bounds__(LF_THREADS) void k_group_candidates(const KL kl, const uint64_t *__restrict__ ukey, uint64_t nd,
                                                                 uint8_t *__restrict__ cand, const CandEmit em, const bool two) {
    __shared__ uint32_t bm[LF_WORDS];
    __shared__ uint32_t s_first;
    __shared__ unsigned long long s_end;
    __shared__ uint32_t s_cnt;                 // EMIT: candidates of the tile so far
    __shared__ unsigned long long s_base;      // EMIT: the tile's stretch of the list
    const uint32_t tid = threadIdx.x;
    const uint64_t n_tiles = (nd + LF_TILE - 1) / LF_TILE;
    const uint32_t emask = (1u << LF_ENTRY_BITS) - 1u;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * LF_TILE;
        const uint64_t t1 = t0 + LF_TILE < nd ? t0 + LF_TILE : nd;
        if (tid == 0) {
            s_first = 0xFFFFFFFFu;
            s_cnt = 0u;
        }
        for (uint32_t w = tid; w < LF_WORDS; w += LF_THREADS) bm[w] = 0u;
        __syncthreads();
        // first barcode head of the tile
        {
            uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t j = 0; j < LF_TILE / LF_THREADS; j++) {
                const uint64_t p = t0 + j * LF_THREADS + tid;
                if (p < t1 && mine == 0xFFFFFFFFu && (p == 0 || (ukey[p] >> kl.sh_bc) != (ukey[p - 1] >> kl.sh_bc)))
                    mine = (uint32_t)(p - t0);
            }
            if (mine != 0xFFFFFFFFu) atomicMin(&s_first, mine);
        }
        __syncthreads();
        const uint32_t first = s_first;
        if (tid == 0) {
            // first barcode head at or after the end of the tile: galloping + binary search on the barcode field
            uint64_t b = nd;
            if (first != 0xFFFFFFFFu && t1 < nd) {
                const uint64_t bcv = ukey[t1 - 1] >> kl.sh_bc;
                uint64_t lo = t1 - 1, step = 1;  // ukey[lo] has barcode bcv
                while (lo + step < nd && (ukey[lo + step] >> kl.sh_bc) == bcv) {
                    lo += step;
                    step <<= 1;
                }
                uint64_t hi = lo + step < nd ? lo + step : nd;  // ukey[hi] differs (or hi == nd)
                while (lo + 1 < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if ((ukey[mid] >> kl.sh_bc) == bcv) lo = mid; else hi = mid;
                }
                b = hi;
            }
            s_end = b;
        }
        __syncthreads();
        if (first == 0xFFFFFFFFu) continue;  // the whole tile is inside a barcode an earlier tile owns (uniform)
        const uint64_t a = t0 + first, b = s_end;
        for (uint64_t k0 = a + tid; k0 < b; k0 += (uint64_t)LF_THREADS * LF_BATCH) {
            uint64_t key[LF_BATCH];
#pragma unroll
            for (int j = 0; j < LF_BATCH; j++) {
                const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                key[j] = k < b ? ukey[k] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < LF_BATCH; j++) {
                if (k0 + (uint64_t)j * LF_THREADS >= b) break;
                const uint64_t mx = mix64(group_id(kl, key[j]));
                lf_note(bm, (uint32_t)mx & emask);
                if (two) lf_note(bm, (uint32_t)(mx >> 24) & emask);
            }
        }
        __syncthreads();
        if (!EMIT) {
            for (uint64_t k0 = a + tid; k0 < b; k0 += (uint64_t)LF_THREADS * LF_BATCH) {
                uint64_t key[LF_BATCH];
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    key[j] = k < b ? ukey[k] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    if (k >= b) break;
                    cand[k] = (uint8_t)lf_again(bm, mix64(group_id(kl, key[j])), emask, two);
                }
            }
        } else {
            // count the range's candidates (a second read of its keys, out of L2), reserve the range's stretch of the list
            // with ONE global atomic, then read the keys once more and append: slots inside the stretch come from an LDS
            // counter that every wave bumps once per item slot (no barrier inside the passes)
            const uint32_t lane = tid & 63u;
            uint32_t mine = 0;
            for (uint64_t k0 = a + tid; k0 < b; k0 += (uint64_t)LF_THREADS * LF_BATCH) {
                uint64_t key[LF_BATCH];
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    key[j] = k < b ? ukey[k] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    mine += (k < b && lf_again(bm, mix64(group_id(kl, key[j])), emask, two)) ? 1u : 0u;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
            if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
            __syncthreads();
            if (tid == 0) {
                s_base = s_cnt ? atomicAdd(em.n_out, (unsigned long long)s_cnt) : 0ull;
                s_cnt = 0u;
            }
            __syncthreads();
            const unsigned long long base = s_base;
            for (uint64_t k0 = a + tid; k0 < b; k0 += (uint64_t)LF_THREADS * LF_BATCH) {
                uint64_t key[LF_BATCH];
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    key[j] = k < b ? ukey[k] : 0ull;
                }
#pragma unroll
                for (int j = 0; j < LF_BATCH; j++) {
                    const uint64_t k = k0 + (uint64_t)j * LF_THREADS;
                    const bool is_c = k < b && lf_again(bm, mix64(group_id(kl, key[j])), emask, two);
                    const unsigned long long m = __ballot(is_c);
                    if (!m) continue;  // wave-uniform
                    uint32_t wbase = 0;
                    if (lane == (uint32_t)(__ffsll((long long)m) - 1)) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
                    wbase = __shfl(wbase, __ffsll((long long)m) - 1);
                    if (is_c) {
                        const unsigned long long o = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        // the high half of the mix: independent of the low bits the filter consumed (as EmitHash)
                        const uint64_t g = mix64(group_id(kl, key[j]) ^ 0x9E3779B97F4A7C15ull);
                        em.hash[o] = LS_HASH_BITS >= 32u ? (uint32_t)g : ((uint32_t)g & ((1u << (LS_HASH_BITS & 31u)) - 1u));
                        em.val[o] = (em.vbits >= 32u ? 0u : ((uint32_t)(g >> 32) << em.vbits)) | (uint32_t)k;
                    }
                }
            }
        }
        __syncthreads();
    }
}
struct CandFlag {
    const uint8_t *cand;
    __device__ __forceinline__ bool operator()(uint64_t k) const { return cand[k] != 0; }
};
// Candidates are sorted by a 32-bit hash of their group (four passes over 8-byte pairs); members of a group are
// then adjacent inside a run of equal hashes (27 bits in three 9-bit passes were tried: the sort got 0.5 ms faster and
// k_low_support 0.6 ms slower on its longer runs).  val = (extra hash bits << vbits) | index: the bits of the u32 payload the
// index does not need carry more hash bits, so that most false collisions of the sort key are rejected without touching
// ukey; every comparison re-checks the exact (barcode, library, UMI) anyway.
struct EmitHash {  // emit of the candidate compaction: (hash, val) of candidate k straight from its key
    KL kl;
    const uint64_t *ukey;
    uint32_t vbits;
    uint32_t *hash, *val;
    typedef uint64_t Pre;
    __device__ __forceinline__ Pre pre(uint64_t k) const { return ukey[k]; }
    __device__ __forceinline__ void operator()(uint64_t k, uint32_t o, Pre key) const {
        // the high half of the mix: independent of the low bits the candidate filter consumed
        const uint64_t g = mix64(group_id(kl, key) ^ 0x9E3779B97F4A7C15ull);
        hash[o] = LS_HASH_BITS >= 32u ? (uint32_t)g : ((uint32_t)g & ((1u << (LS_HASH_BITS & 31u)) - 1u));
        const uint32_t extra = vbits >= 32u ? 0u : ((uint32_t)(g >> 32) << vbits);
        val[o] = extra | (uint32_t)k;
    }
};

__global__ __launch_bounds__(256) void k_low_support(const KL kl, const uint32_t *__restrict__ hash,
                                                     const uint32_t *__restrict__ val, uint64_t n_cand, uint64_t nd,
                                                     uint32_t vbits,
                                                     const uint64_t *__restrict__ ukey,
                                                     const uint32_t *__restrict__ upos, uint64_t n_keys,
                                                     uint16_t *__restrict__ st) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_cand; j += stride) {
        const uint32_t h = hash[j];
        // almost every run of equal hashes is a singleton: a single (umi, feature) key is its own
        // strict maximum and never low support
        const bool same_prev = j > 0 && hash[j - 1] == h;
        const bool same_next = j + 1 < n_cand && hash[j + 1] == h;
        if (!same_prev && !same_next) continue;
        uint64_t s = j, e = j + 1;
        while (s > 0 && hash[s - 1] == h) s--;
        while (e < n_cand && hash[e] == h) e++;
        const uint32_t vmask = vbits >= 32u ? 0xFFFFFFFFu : ((1u << vbits) - 1u);
        const uint32_t my_val = val[j];
        // cheap filter on the extra hash bits before any gather
        bool any = false;
        for (uint64_t t = s; t < e; t++) any |= t != j && ((val[t] ^ my_val) & ~vmask) == 0u;
        if (!any) continue;
        const uint32_t me = my_val & vmask;
        const uint64_t g = group_id(kl, ukey[me]);
        // counts after moving ONE read of each corrected key (mark_dups.rs:226-232); zero-count keys stay
        const uint32_t my_s = st[me];
        const uint32_t my_c1 = run_count(upos, nd, n_keys, me) - ((my_s & ST_CORRECTED) ? 1u : 0u) + st_inc1(my_s);
        uint32_t mx = my_c1, n_members = 0, n_max = 0;
        for (uint64_t t = s; t < e; t++) {
            if (((val[t] ^ my_val) & ~vmask) != 0u) continue;  // differs in the extra hash bits
            const uint32_t k = val[t] & vmask;
            if (group_id(kl, ukey[k]) != g) continue;  // hash collision: a different group
            const uint32_t ks = st[k];
            const uint32_t c1 = run_count(upos, nd, n_keys, k) - ((ks & ST_CORRECTED) ? 1u : 0u) + st_inc1(ks);
            n_members++;
            if (c1 > mx) {
                mx = c1;
                n_max = 1;
            } else if (c1 == mx) {
                n_max++;
            }
        }
        if (n_members < 2) continue;
        // low iff below the group's maximum, or the maximum is shared (mark_dups.rs:96-106)
        if (my_c1 < mx || n_max >= 2) st_or(st, me, ST_LOW);
    }
}

__global__ __launch_bounds__(256) void k_triplets(const KL kl, const uint64_t *__restrict__ mkeys,
                                                  const uint32_t *__restrict__ tpos, uint64_t nt, uint64_t nm,
                                                  uint32_t *__restrict__ bc, uint32_t *__restrict__ feat,
                                                  uint32_t *__restrict__ cnt, const uint32_t *__restrict__ back) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += stride) {
        const uint32_t p = tpos[t];
        const uint32_t end = t + 1 < nt ? tpos[t + 1] : (uint32_t)nm;
        const uint64_t key = mkeys[p];
        const uint32_t bcf = (uint32_t)(key >> kl.sh_bc);
        bc[t] = back ? back[bcf] : bcf;
        feat[t] = (uint32_t)((key >> kl.sh_feat) & lowmask(kl.bits_feat));
        cnt[t] = end - p;  // number of surviving molecules of (barcode, feature): types.rs:180-188
    }
}

// ------------------------------------------------------------------------------------------------
// per-read DupInfo (mark_dups.rs:61-72, BarcodeDupMarker::process :280-363)
// ------------------------------------------------------------------------------------------------
// representative read of a raw key = min (utype, qname) over its reads (mark_dups.rs:137-152): the run is
// sorted by the utype bit (LSB of the key); among its leading same-bit elements take the smallest ordinal.
__global__ __launch_bounds__(256) void k_rep_read(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                  const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                  uint32_t *__restrict__ rep_read) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += stride) {
        const uint32_t b = upos[k], e = k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys;
        const uint64_t bit = keys[b] & 1ull;
        uint32_t best = vals[b];
        for (uint32_t i = b + 1; i < e && (keys[i] & 1ull) == bit; i++) best = vals[i] < best ? vals[i] : best;
        rep_read[k] = best;
    }
}

struct __attribute__((aligned(4))) DupRec {
    uint32_t umi, read_count, flags;
};

// PACK8 (UMIs of at most 12 bases): the record is 8 bytes -- (flags << 24 | UMI) and the read count -- one aligned 8-byte store
// per read instead of a 12-byte one that may straddle two sectors
struct __attribute__((aligned(8))) DupRec8 {
    uint32_t umi_flags, read_count;
};
template <bool PACK8>
__global__ __launch_bounds__(256) void k_per_read(const KL kl, const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ vals,
                                                  const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                  const uint32_t *__restrict__ corr, const uint32_t *__restrict__ inc_all,
                                                  const uint16_t *__restrict__ st, const uint32_t *__restrict__ minidx,
                                                  const uint32_t *__restrict__ rep_read, void *__restrict__ packed_v,
                                                  const TargetFilter tf) {
    DupRec *__restrict__ packed = reinterpret_cast<DupRec *>(packed_v);
    DupRec8 *__restrict__ packed8 = reinterpret_cast<DupRec8 *>(packed_v);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += stride) {
        const uint32_t b = upos[k], e = k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys;
        const bool corrected = (st[k] & ST_CORRECTED) != 0u;
        const uint32_t K = corrected ? corr[k] : (uint32_t)k;  // the key this run's reads land on
        const uint32_t sK = st[K];
        const uint32_t endK = (uint64_t)K + 1 < nd ? upos[K + 1] : (uint32_t)n_keys;
        const uint32_t cntK = endK - upos[K];
        const bool is_target = st_inc1(sK) != 0u;
        const uint32_t read_count = ((sK & ST_CORRECTED) ? 0u : cntK) + (is_target ? inc_all[K] : 0u);  // umigene_counts[corrected_key]
        const uint32_t umi = (uint32_t)((ukey[K] >> kl.sh_umi) & lowmask(kl.bits_umi));
        const uint32_t mi = is_target ? rep_source(minidx[K], K, sK) : NONE32;
        const uint32_t rep_key = mi != NONE32 ? mi : K;  // umigene_min_key[corrected_key]
        const uint32_t rep = rep_read[rep_key];
        const bool lowK = (sK & ST_LOW) != 0u;
        const bool filt = tf.filtered(ukey[K], read_count, lowK);
        const uint8_t base = (uint8_t)(CRGPU_DUP_HAS | (corrected ? CRGPU_DUP_CORRECTED : 0) |
                                       (lowK ? CRGPU_DUP_LOW_SUPPORT : 0) | (filt ? CRGPU_DUP_FILTERED_TARGET : 0));
        for (uint32_t i = b; i < e; i++) {
            const uint32_t r = vals[i];
            // one 12-byte store per read: the position of a read in sorted order has nothing to do with its ordinal,
            // and three separate scattered stores cost three partial-line writes per read (38 GB per 200 M reads)
            const uint32_t fl = (uint32_t)(base | ((!lowK && !filt && r == rep) ? CRGPU_DUP_UMI_COUNT : 0));
            if (PACK8)
                packed8[r] = DupRec8{(fl << 24) | umi, read_count};
            else
                packed[r] = DupRec{umi, read_count, fl};
        }
    }
}

// the packed records -> the three output arrays of the ABI, streaming
template <bool PACK8>
__global__ __launch_bounds__(256) void k_unpack_dupinfo(const void *__restrict__ packed_v, uint64_t n, uint32_t *__restrict__ out_umi,
                                                        uint32_t *__restrict__ out_cnt, uint8_t *__restrict__ out_flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        uint32_t umi, cnt, fl;
        if (PACK8) {
            const DupRec8 d = reinterpret_cast<const DupRec8 *>(packed_v)[r];
            umi = d.umi_flags & 0xFFFFFFu;
            fl = d.umi_flags >> 24;
            cnt = d.read_count;
        } else {
            const DupRec d = reinterpret_cast<const DupRec *>(packed_v)[r];
            umi = d.umi;
            cnt = d.read_count;
            fl = d.flags;
        }
        if (out_umi) out_umi[r] = umi;
        if (out_cnt) out_cnt[r] = cnt;
        if (out_flags) out_flags[r] = (uint8_t)fl;
    }
}

// ---- the same through windows of read ordinals ---------------------------------------------------------------------------
// (experiment, off by default -- see the driver) A read's position in sorted-key order has nothing to do with its ordinal:
// k_per_read's 12-byte stores go all over the output (35 ms per 1 B records, 23 G stores/s).  Instead: one 8-byte record per SORTED position (coalesced), a stable
// counting pass that groups (record, ordinal) by the top 9 bits of the ordinal (cr_partition_by_payload), and a scatter
// whose stores then stay inside one window of n / 512 reads at a time -- small enough for the memory-side cache to merge
// them into whole lines.  Packed record: [processed UMI 32][read_count 27][flags 5]; a read count that does not fit
// raises *overflow and the host takes the direct path above.
#define PR_COUNT_BITS 27u
__device__ __forceinline__ uint64_t pack_duprec(uint32_t umi, uint32_t read_count, uint32_t flags) {
    return ((uint64_t)umi << 32) | ((uint64_t)read_count << 5) | (flags & 0x1Fu);
}
__global__ __launch_bounds__(256) void k_per_read_sorted(const KL kl, const uint64_t *__restrict__ ukey,
                                                         const uint32_t *__restrict__ vals, const uint32_t *__restrict__ upos,
                                                         uint64_t nd, uint64_t n_keys, const uint32_t *__restrict__ corr,
                                                         const uint32_t *__restrict__ inc_all, const uint16_t *__restrict__ st,
                                                         const uint32_t *__restrict__ minidx, const uint32_t *__restrict__ rep_read,
                                                         uint64_t *__restrict__ prec, uint32_t *__restrict__ overflow,
                                                         const TargetFilter tf) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += stride) {
        const uint32_t b = upos[k], e = k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys;
        const bool corrected = (st[k] & ST_CORRECTED) != 0u;
        const uint32_t K = corrected ? corr[k] : (uint32_t)k;
        const uint32_t sK = st[K];
        const uint32_t endK = (uint64_t)K + 1 < nd ? upos[K + 1] : (uint32_t)n_keys;
        const uint32_t cntK = endK - upos[K];
        const bool is_target = st_inc1(sK) != 0u;
        const uint32_t read_count = ((sK & ST_CORRECTED) ? 0u : cntK) + (is_target ? inc_all[K] : 0u);
        if (read_count >> PR_COUNT_BITS) *overflow = 1u;
        const uint32_t umi = (uint32_t)((ukey[K] >> kl.sh_umi) & lowmask(kl.bits_umi));
        const uint32_t mi = is_target ? rep_source(minidx[K], K, sK) : NONE32;
        const uint32_t rep = rep_read[mi != NONE32 ? mi : K];
        const bool lowK = (sK & ST_LOW) != 0u;
        const bool filt = tf.filtered(ukey[K], read_count, lowK);
        const uint32_t base = CRGPU_DUP_HAS | (corrected ? CRGPU_DUP_CORRECTED : 0u) | (lowK ? CRGPU_DUP_LOW_SUPPORT : 0u) |
                              (filt ? CRGPU_DUP_FILTERED_TARGET : 0u);
        for (uint32_t i = b; i < e; i++)
            prec[i] = pack_duprec(umi, read_count, base | ((!lowK && !filt && vals[i] == rep) ? CRGPU_DUP_UMI_COUNT : 0u));
    }
}
// (record, ordinal) pairs grouped by ordinal window -> the output arrays (any may be NULL) or the packed 12-byte records
__global__ __launch_bounds__(256) void k_scatter_records(const uint64_t *__restrict__ prec, const uint32_t *__restrict__ ordinal,
                                                         uint64_t n_keys, uint32_t *__restrict__ out_umi,
                                                         uint32_t *__restrict__ out_cnt, uint8_t *__restrict__ out_flags,
                                                         DupRec *__restrict__ packed_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_keys; j += stride) {
        const uint64_t d = prec[j];
        const uint32_t r = ordinal[j];
        const uint32_t umi = (uint32_t)(d >> 32), cnt = (uint32_t)((d >> 5) & ((1u << PR_COUNT_BITS) - 1u)), fl = (uint32_t)(d & 0x1Fu);
        if (packed_out) packed_out[r] = DupRec{umi, cnt, fl};
        if (out_umi) out_umi[r] = umi;
        if (out_cnt) out_cnt[r] = cnt;
        if (out_flags) out_flags[r] = (uint8_t)fl;
    }
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
// BarcodeSummary::observe (cr_lib/src/aligner.rs:54-67), column umi_corrected_reads: reads whose raw (umi, feature) is a
// key of umi_corrections.  Corrections are rare (the UMI error rate), so these are few scattered atomics.
__global__ __launch_bounds__(256) void k_corrected_reads(const KL kl, const uint64_t *__restrict__ ukey,
                                                         const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                         const uint16_t *__restrict__ st, uint32_t W,
                                                         uint32_t *__restrict__ tab, const uint32_t *__restrict__ back) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += stride) {
        if (!(st[k] & ST_CORRECTED)) continue;
        const uint64_t key = ukey[k];
        const uint32_t run = (k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys) - upos[k];
        const uint32_t lib = (uint32_t)((key >> kl.sh_libid) & lowmask(kl.bits_lib));
        const uint32_t bcf = (uint32_t)(key >> kl.sh_bc);
        atomicAdd(&tab[(size_t)lib * W + (back ? back[bcf] : bcf)], run);
    }
}

// reads of the keys that the targeted-panel filter took out of the molecule table: they still count as
// candidate_dup_reads in the BarcodeSummary (aligner.rs:54-67 looks at is_low_support_umi only)
__global__ __launch_bounds__(256) void k_filtered_reads(const KL kl, const uint64_t *__restrict__ ukey,
                                                        const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                        const uint16_t *__restrict__ st, const uint32_t *__restrict__ inc_all,
                                                        const TargetFilter tf, uint32_t W, uint32_t *__restrict__ tab,
                                                        const uint32_t *__restrict__ back) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += stride) {
        const uint32_t s = st[k];
        const bool landed = !(s & ST_CORRECTED) | (st_inc1(s) > 0u);
        if (!landed || (s & ST_LOW)) continue;
        const uint32_t end = k + 1 < nd ? upos[k + 1] : (uint32_t)n_keys;
        const uint32_t rc = ((s & ST_CORRECTED) ? 0u : end - upos[k]) + (st_inc1(s) ? inc_all[k] : 0u);
        const uint64_t key = ukey[k];
        if (!tf.filtered(key, rc, false)) continue;
        const uint32_t lib = (uint32_t)((key >> kl.sh_libid) & lowmask(kl.bits_lib));
        const uint32_t bcf = (uint32_t)(key >> kl.sh_bc);
        atomicAdd(&tab[(size_t)lib * W + (back ? back[bcf] : bcf)], rc);
    }
}

// columns umis and candidate_dup_reads of the BarcodeSummary from the molecule table: every molecule has exactly one read
// with is_umi_count(), and the reads of the kept (not low-support) molecules are the candidate_dup_reads.  Molecules are
// sorted by barcode, so a thread sums a run of its MS_ITEMS consecutive molecules before it touches the table.
constexpr int MS_ITEMS = 8;
__global__ __launch_bounds__(256) void k_molecule_sums(const KL kl, const uint64_t *__restrict__ mkeys,
                                                       const uint32_t *__restrict__ mreads, uint64_t nm, uint32_t W,
                                                       uint32_t *__restrict__ umis, uint32_t *__restrict__ cand,
                                                       const uint32_t *__restrict__ back) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * MS_ITEMS;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * MS_ITEMS; base < nm; base += stride) {
        uint64_t key[MS_ITEMS];
        uint32_t rd[MS_ITEMS];
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
            const uint64_t i = base + j < nm ? base + j : nm - 1;
            key[j] = mkeys[i];
            rd[j] = mreads[i];
        }
        size_t cur = ~(size_t)0;
        uint32_t n_u = 0, n_c = 0;
#pragma unroll
        for (int j = 0; j < MS_ITEMS; j++) {
            if (base + j >= nm) break;
            const uint32_t lib = (uint32_t)((key[j] >> kl.sh_libid) & lowmask(kl.bits_lib));
            const uint32_t bcf = (uint32_t)(key[j] >> kl.sh_bc);
            const size_t slot = (size_t)lib * W + (back ? back[bcf] : bcf);
            if (slot != cur) {
                if (n_u) {
                    atomicAdd(&umis[cur], n_u);
                    atomicAdd(&cand[cur], n_c);
                }
                cur = slot;
                n_u = n_c = 0;
            }
            n_u += 1;
            n_c += rd[j];
        }
        if (n_u) {
            atomicAdd(&umis[cur], n_u);
            atomicAdd(&cand[cur], n_c);
        }
    }
}

struct SummaryFlag {  // barcode rank with at least one read of this library (a BarcodeSummary row exists)
    const uint32_t *valid, *corrected;
    uint32_t lo;
    __device__ __forceinline__ bool operator()(uint64_t i) const { return (valid[lo + i] + corrected[lo + i]) != 0u; }
};
struct EmitSummary {
    const uint32_t *valid, *corrected, *umis, *cand, *corr_reads, *filt_reads;  // the last four may be NULL
    uint32_t lo, lib;
    crgpu_barcode_summary_row *rows;
    struct Pre {};
    __device__ __forceinline__ Pre pre(uint64_t) const { return Pre(); }
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t o, Pre) const {
        const uint32_t r = lo + (uint32_t)i;
        crgpu_barcode_summary_row w;
        w.barcode_rank = r;
        w.library = lib;
        w.reads = (uint64_t)valid[r] + corrected[r];
        w.umis = umis ? umis[r] : 0u;
        w.candidate_dup_reads = (uint64_t)(cand ? cand[r] : 0u) + (filt_reads ? filt_reads[r] : 0u);
        w.umi_corrected_reads = corr_reads ? corr_reads[r] : 0u;
        rows[o] = w;
    }
};

struct PerRead {
    bool summary = false;        // keep the per-barcode corrected-read table for crgpu_counts_barcode_summary
    uint64_t n_reads = 0;        // entries of the output arrays
    uint32_t *d_vals = nullptr;  // read ordinal of every key (sorted along with the keys)
    uint32_t *out_umi = nullptr, *out_cnt = nullptr;
    uint8_t *out_flags = nullptr;
    struct DupRec *packed_out = nullptr;  // non-NULL: leave the packed 12-byte records here (n_reads entries), no unpacking
    const int32_t *d_probe = nullptr;     // probe index per read ordinal (crgpu_records.d_probe_idx): molecules get d_mprobe
};

// Two branches of the count stage that only read the distinct keys run side by side: the UMI correction stays on the
// context's stream, the search for low-support candidates (filter, compaction, hash sort) goes to the second one.  Between
// side() and join() every launch, scratch read-back and timer that goes through ctx->stream lands on the second stream; the
// ledger books the whole region as ONE span on the main stream (inner timers are off: overlapping spans would count the
// shared time twice).  Temporaries of both branches must stay alive until join(): the pool hands a freed block to the next
// caller at once, which is only safe on one in-order stream.  CRGPU_DEDUP_OVERLAP=0 keeps everything on one stream.
struct CrFork {
    crgpu_ctx *ctx;
    hipStream_t main = nullptr;
    bool on_side = false, was_timing = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    explicit CrFork(crgpu_ctx *c) : ctx(c) {}
    // CRGPU_DEDUP_OVERLAP: 0 = never, 2 = for every input (tests), otherwise from 2^20 distinct keys on
    static bool enabled(const crgpu_ctx *c, uint64_t nd) {
        const char *e = getenv("CRGPU_DEDUP_OVERLAP");
        if (!c->stream2 || (e && e[0] == '0')) return false;
        return (e && e[0] == '2') ? nd > 0 : nd >= (1u << 20);
    }
    // call after the main branch has been enqueued; start_of_region = the event recorded before it
    int side(hipEvent_t start_of_region) {
        main = ctx->stream;
        was_timing = ctx->timing;
        t0 = start_of_region;
        if (hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0) != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "hipStreamWaitEvent failed");
        ctx->stream = ctx->stream2;
        ctx->timing = false;
        on_side = true;
        return CRGPU_OK;
    }
    // back to the main stream for the host code that follows; the side branch stays open (inner timers stay off) until join()
    void pause() {
        if (on_side) ctx->stream = main;
    }
    int join() {
        if (!on_side) return CRGPU_OK;
        on_side = false;
        hipError_t e = hipEventRecord(ctx->ev_join, ctx->stream2);
        ctx->stream = main;
        ctx->timing = was_timing;
        if (e == hipSuccess) e = hipStreamWaitEvent(main, ctx->ev_join, 0);
        if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "joining the second stream failed: %s", hipGetErrorString(e));
        if (t0) {
            hipEventRecord(t1, main);
            ctx->spans.push_back({CRGPU_T_DEDUP, t0, t1, 0});
            t0 = t1 = nullptr;
        }
        return CRGPU_OK;
    }
    ~CrFork() {  // error path: nothing of the side branch may outlive the call
        if (on_side) {
            (void)hipStreamSynchronize(ctx->stream2);
            ctx->stream = main;
            ctx->timing = was_timing;
        }
        if (t0) {
            ctx->event_pool.push_back(t0);
            ctx->event_pool.push_back(t1);
        }
    }
};

static int count_keys_impl(crgpu_ctx *ctx, uint64_t *d_keys_inout, uint64_t n_keys, crgpu_counts **out, PerRead pr) {
    if (!ctx || !out) return CRGPU_EINVAL;
    *out = nullptr;
    CR_REQUIRE(ctx, ctx->layout.set, CRGPU_ESTATE, "crgpu_count_keys: call crgpu_set_key_layout first");
    CR_REQUIRE(ctx, n_keys <= 0x7FFFFFFFull, CRGPU_ERANGE, "crgpu_count_keys: at most 2^31-1 keys per call");
    // CRGPU_OPT_DENSE_BARCODE_KEYS: the index the keys were built with; if something (crgpu_invalidate, ...) dropped it in
    // between, the tables -- unchanged by the host's promise -- give the same index again
    CR_TRY(cr_dense_ensure(ctx));
    const KeyLayout &L = ctx->layout;
    const KL kl = make_kl(L);
    crgpu_counts *res = new (std::nothrow) crgpu_counts();
    if (!res) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    res->layout = L;
    res->n_canon = ctx->n_canon;
    if (n_keys == 0) {
        *out = res;
        return CRGPU_OK;
    }
    CR_REQUIRE(ctx, L.total_bits() <= 64, CRGPU_ESTATE,
               "crgpu_count_keys: the key layout needs the dense barcode index (CRGPU_OPT_DENSE_BARCODE_KEYS) and none is valid -- "
               "were the keys built before the last change of a histogram table?");
    CR_REQUIRE(ctx, d_keys_inout, CRGPU_EINVAL, "crgpu_count_keys: NULL keys");
    struct Guard {
        crgpu_ctx *c;
        crgpu_counts *r;
        bool armed = true;
        ~Guard() {
            if (armed) crgpu_counts_free(c, r);
        }
    } guard{ctx, res};

    uint32_t *d_block = ctx->d_sort_hist;      // 4096 u32 block counters of the compactions
    uint32_t *d_total = ctx->d_scalars + 16;   // device-side totals
    if (ctx->dense.valid) {  // the result keeps its own copy of the column -> rank map: the context's may change before it is read
        res->n_back = ctx->dense.V;
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_back, (size_t)ctx->dense.V * sizeof(uint32_t)));
        CR_HIP(ctx, hipMemcpyAsync(res->d_back, ctx->dense.d_back, (size_t)ctx->dense.V * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                   ctx->stream));
    }

    // 0. the per-key state of the UMI correction (section "per-key state") has to start out zeroed: 10 bytes per distinct key.
    //    The number of distinct keys is not known yet, but it is at most n_keys: the arrays are sized by that and zeroed NOW on
    //    the second stream, in the shadow of the sort (which is bound by the latency of its chunks, not by bandwidth) instead
    //    of 0.8 ms on the critical path behind the run lengths.  CRGPU_NO_PREZERO=1: as before.
    DevBuf corr_b, incall_b, st_b, minidx_b;
    // (measured at 1 B records: the 8 GB of zeroing cost the sort passes more than the 0.8 ms they save -- sized by n_keys they
    // are twice what the distinct keys need: OFF unless CRGPU_PREZERO=1, kept for the A/B record)
    const bool prezero = ctx->stream2 && n_keys >= (1u << 20) && getenv("CRGPU_PREZERO") != nullptr;
    // experiment: the run-length write pass zeroes the state of every distinct key it emits (arrays sized by n_keys, the bound on
    // the number of distinct keys known before that pass) instead of three memsets behind it
    // (measured at 1 B records: k_rl_write 2.2 -> 3.3 ms with the three extra stores per distinct key, against 0.8 ms of memsets:
    // OFF unless CRGPU_STATE_FUSED_ZERO=1, kept for the A/B record)
    const bool fused_zero = !prezero && getenv("CRGPU_STATE_FUSED_ZERO") && !getenv("CRGPU_RL_GENERIC") && !cr_sort_finish_experiment();
    const uint64_t st_cap = (prezero || fused_zero) ? n_keys : 0;
    if (fused_zero) {
        CR_TRY(dmalloc(ctx, minidx_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, corr_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, incall_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, st_b, ((st_cap + 1) & ~1ull) * sizeof(uint16_t) + 4));
    }
    if (prezero) {
        CR_TRY(dmalloc(ctx, minidx_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, corr_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, incall_b, st_cap * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, st_b, ((st_cap + 1) & ~1ull) * sizeof(uint16_t) + 4));
        // the blocks may have been in use by work queued on the main stream: the second stream starts behind it
        CR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        CR_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        CR_HIP(ctx, hipMemsetAsync(incall_b.p, 0, st_cap * sizeof(uint32_t), ctx->stream2));
        CR_HIP(ctx, hipMemsetAsync(st_b.p, 0, ((st_cap + 1) & ~1ull) * sizeof(uint16_t) + 4, ctx->stream2));
        CR_HIP(ctx, hipMemsetAsync(minidx_b.p, 0xFF, st_cap * sizeof(uint32_t), ctx->stream2));
        CR_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
    }
    struct PrezeroGuard {  // an error return in between must not leave the second stream writing into blocks the pool hands out again
        crgpu_ctx *c;
        bool armed;
        ~PrezeroGuard() {
            if (armed) (void)hipStreamSynchronize(c->stream2);
        }
    } prezero_guard{ctx, prezero};

    // 1. sort the keys: fully, or on their top bits with the finishing left to the run-length pass (CRGPU_SORT_FINISH)
    DevBuf tmp, vtmp;
    CR_TRY(dmalloc(ctx, tmp, n_keys * sizeof(uint64_t)));
    if (pr.d_vals) CR_TRY(dmalloc(ctx, vtmp, n_keys * sizeof(uint32_t)));
    bool in_tmp = false;
    uint32_t low_left = 0;
    CR_TRY(cr_radix_sort_u64_top(ctx, d_keys_inout, tmp.as<uint64_t>(), pr.d_vals, pr.d_vals ? vtmp.as<uint32_t>() : nullptr, n_keys,
                                 L.total_bits(), &in_tmp, &low_left));
    uint64_t *keys_rw = in_tmp ? tmp.as<uint64_t>() : d_keys_inout;
    uint32_t *vals_rw = pr.d_vals ? (in_tmp ? vtmp.as<uint32_t>() : pr.d_vals) : nullptr;

    // 2. distinct (barcode, feature, library, UMI) keys and their run starts (DupBuilder::observe)
    DevBuf ukey_b, upos_b;
    CR_TRY(dmalloc(ctx, ukey_b, n_keys * sizeof(uint64_t)));
    CR_TRY(dmalloc(ctx, upos_b, (n_keys + 1) * sizeof(uint32_t)));
    uint64_t *ukey = ukey_b.as<uint64_t>();
    uint32_t *upos = upos_b.as<uint32_t>();
    uint64_t nd = 0;
    bool emitted = false;
    if (low_left) {
        bool fell_back = false;
        if (cr_sort_finish_experiment()) {
            CR_TRY(cr_finish_emit(ctx, keys_rw, vals_rw, n_keys, low_left, ukey, upos, &nd, &fell_back));
            emitted = !fell_back;
        } else {
            CR_TRY(cr_order_runs(ctx, keys_rw, vals_rw, n_keys, low_left, &fell_back));  // then the run lengths as usual
        }
        if (fell_back) {
            // a run of equal top bits too long for the fused pass: sort the buffer (the same multiset) on all bits
            ctx->sort_refinished++;
            uint64_t *other = in_tmp ? d_keys_inout : tmp.as<uint64_t>();
            uint32_t *vother = pr.d_vals ? (in_tmp ? pr.d_vals : vtmp.as<uint32_t>()) : nullptr;
            bool flip = false;
            CR_TRY(cr_radix_sort_u64_full(ctx, keys_rw, other, vals_rw, vother, n_keys, L.total_bits(), &flip));
            if (flip) {
                keys_rw = other;
                vals_rw = vother;
            }
        }
    }
    const uint64_t *keys = keys_rw;
    const uint32_t *vals = vals_rw;
    if (!emitted) {
        uint32_t nd32 = 0;
        {
            CrTimer t(ctx, CRGPU_T_DEDUP, n_keys);  // the family's unit: one sorted key (counted here, once per call)
            if (getenv("CRGPU_RL_GENERIC"))  // A/B: the generic compaction with its three loads per key
                CR_TRY(compact(ctx, HeadFlag{keys, 1u}, EmitRun{keys, ukey, upos}, n_keys, d_block, d_total));
            else
                CR_TRY(run_lengths(ctx, keys, 1u, n_keys, ukey, upos, d_block, d_total, fused_zero ? st_b.as<uint16_t>() : nullptr,
                                   fused_zero ? incall_b.as<uint32_t>() : nullptr, fused_zero ? minidx_b.as<uint32_t>() : nullptr));
        }
        CR_TRY(read_u32(ctx, d_total, &nd32));
        nd = nd32;
    }

    // 3. UMI correction + the read moves (state layout: umi_correct.h)
    const uint64_t st_bytes = ((nd + 1) & ~1ull) * sizeof(uint16_t) + 4;
    const bool zeroed = prezero || (fused_zero && !emitted);
    if (!prezero && !fused_zero) {
        CR_TRY(dmalloc(ctx, minidx_b, nd * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, corr_b, nd * sizeof(uint32_t)));   // written (and valid) only where st says "corrected"
        CR_TRY(dmalloc(ctx, incall_b, nd * sizeof(uint32_t)));
        CR_TRY(dmalloc(ctx, st_b, st_bytes));
    } else if (prezero) {
        CR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));  // the zeroing of step 0
        prezero_guard.armed = false;
    }
    uint32_t *corr = corr_b.as<uint32_t>(), *inc_all = incall_b.as<uint32_t>(), *minidx = minidx_b.as<uint32_t>();
    uint16_t *st = st_b.as<uint16_t>();
    // the candidate search of step 4 needs nothing of step 3: it runs beside it on the second stream (CrFork)
    const bool overlap = CrFork::enabled(ctx, nd);
    CrFork fork(ctx);
    DevBuf heads_b, giant_b, best_b;  // step 3's temporaries, alive until the branches have joined
    bool edges_small_on_side = false;
    uint64_t es_tiles = 0;
    uint32_t *es_first = nullptr, *es_last = nullptr, *es_ngiant = nullptr;
    GiantItem *es_items = nullptr;
    if (overlap) {
        CR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        if (ctx->timing) {
            fork.t0 = cr_take_event(ctx);
            fork.t1 = cr_take_event(ctx);
            CR_HIP(ctx, hipEventRecord(fork.t0, ctx->stream));
        }
    }
    {
        const bool timing_was = ctx->timing;
        if (overlap) ctx->timing = false;  // one span for the whole region (CrFork::join)
        struct TimingBack {
            crgpu_ctx *c;
            bool v;
            ~TimingBack() { c->timing = v; }
        } timing_back{ctx, timing_was};
        CrTimer t(ctx, CRGPU_T_DEDUP);
        if (!zeroed) {
            CR_HIP(ctx, hipMemsetAsync(inc_all, 0, nd * sizeof(uint32_t), ctx->stream));
            CR_HIP(ctx, hipMemsetAsync(st, 0, st_bytes, ctx->stream));
            CR_HIP(ctx, hipMemsetAsync(minidx, 0xFF, nd * sizeof(uint32_t), ctx->stream));
        }
        const uint64_t n_tiles = (nd + UC_TILE - 1) / UC_TILE;
        CR_TRY(dmalloc(ctx, heads_b, 2 * n_tiles * sizeof(uint32_t)));  // per tile: first / last segment head
        uint32_t *tile_first = heads_b.as<uint32_t>(), *tile_last = tile_first + n_tiles;
#ifndef UC_GRID_WG
#define UC_GRID_WG 5u  // workgroups per CU: what LDS and registers allow (4 left a fifth of the CU idle: 4.39 -> 3.67 ms; 6 with 80 VGPRs: 4.2)
#endif
        hipLaunchKernelGGL(k_correct_umis_tiled, dim3(cr_grid(n_tiles, 1, 256u * UC_GRID_WG)), dim3(256), 0, ctx->stream, kl, ukey,
                           upos, nd, n_keys, tile_first, tile_last, corr, st, inc_all, minidx);
        if (n_tiles > 1) {
            const size_t lds_small = (2 * UES_CAP + UES_BUCKETS * 8) * sizeof(uint32_t);
            const size_t lds_large = (2 * UE_CAP + UE_BUCKETS * 8) * sizeof(uint32_t);
            cr_allow_lds(ctx, (const void *)k_correct_umis_edges<false>, lds_large);
            cr_allow_lds(ctx, (const void *)k_giant_probe, lds_large);
            // work list of the segments with more than UE_CAP keys: one item per chunk of UE_CAP keys
            const uint64_t max_items = nd / (UE_CAP / 2) + 16;
            CR_TRY(dmalloc(ctx, giant_b, max_items * sizeof(GiantItem) + 16));
            CR_TRY(dmalloc(ctx, best_b, nd * sizeof(unsigned long long)));
            uint32_t *n_giant = reinterpret_cast<uint32_t *>(giant_b.as<unsigned char>());
            GiantItem *items = reinterpret_cast<GiantItem *>(giant_b.as<unsigned char>() + 16);
            unsigned long long *best = best_b.as<unsigned long long>();
            CR_HIP(ctx, hipMemsetAsync(n_giant, 0, sizeof(uint32_t), ctx->stream));
            // the small edge segments only need the tile heads: with two streams they go to the second one, behind the candidate
            // search (which ends before this branch does), and leave this branch the large segments and the giant chain
            // (measured at 1 B records: 61.66 against 61.45 ms per step with everything on the main stream -- no gain, the candidate
            // branch is not idle for long enough: OFF unless CRGPU_EDGES_SIDE=1)
            edges_small_on_side = overlap && ctx->ev_aux && getenv("CRGPU_EDGES_SIDE") != nullptr;
            if (edges_small_on_side) {
                CR_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));  // the tile heads are written
                es_tiles = n_tiles;
                es_first = tile_first;
                es_last = tile_last;
                es_items = items;
                es_ngiant = n_giant;
            } else
                hipLaunchKernelGGL(k_correct_umis_edges<true>, dim3(cr_grid(n_tiles - 1, 1, 256u * 6u)), dim3(UES_THREADS),
                                   lds_small, ctx->stream, kl, ukey, upos, nd, n_keys, tile_first, tile_last, corr, st, inc_all,
                                   minidx, items, n_giant);
            hipLaunchKernelGGL(k_correct_umis_edges<false>, dim3(cr_grid(n_tiles - 1, 1, 256u * 2u)), dim3(UE_THREADS),
                               lds_large, ctx->stream, kl, ukey, upos, nd, n_keys, tile_first, tile_last, corr, st, inc_all,
                               minidx, items, n_giant);
            // the three kernels loop over the device-side item count (usually a few hundred, often zero)
            hipLaunchKernelGGL(k_giant_init, dim3(512), dim3(UE_THREADS), 0, ctx->stream, kl, ukey, upos, nd, n_keys, items,
                               n_giant, best);
            hipLaunchKernelGGL(k_giant_probe, dim3(512), dim3(UE_THREADS), lds_large, ctx->stream, kl, ukey, upos, nd, n_keys,
                               items, n_giant, best);
            hipLaunchKernelGGL(k_giant_final, dim3(512), dim3(UE_THREADS), 0, ctx->stream, upos, nd, n_keys, items, n_giant, best,
                               corr, st, inc_all, minidx);
        }
        CR_HIP(ctx, hipGetLastError());
    }

    // 4. low support: candidate keys (UMI seen under several features of a barcode) -> grouped by
    //    (barcode, library, UMI) through a 32-bit hash sort -> exact comparison of the phase-1 counts
    {
        DevBuf cand_b, h_b, v_b;
        bool side_edges_done = false;
        auto side_edges = [&]() -> int {  // the small edge segments of step 3, on the stream the candidate search ran on
            if (!edges_small_on_side || side_edges_done || !fork.on_side) return CRGPU_OK;
            side_edges_done = true;
            CR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux, 0));
            const size_t lds_small = (2 * UES_CAP + UES_BUCKETS * 8) * sizeof(uint32_t);
            hipLaunchKernelGGL(k_correct_umis_edges<true>, dim3(cr_grid(es_tiles - 1, 1, 256u * 6u)), dim3(UES_THREADS), lds_small,
                               ctx->stream, kl, ukey, upos, nd, n_keys, es_first, es_last, corr, st, inc_all, minidx, es_items, es_ngiant);
            CR_HIP(ctx, hipGetLastError());
            return CRGPU_OK;
        };
        if (!getenv("CRGPU_CAND_EMIT")) CR_TRY(dmalloc(ctx, cand_b, nd));
        CR_TRY(dmalloc(ctx, h_b, nd * sizeof(uint32_t)));   // room for every key; the candidates are ~1/5 of them
        CR_TRY(dmalloc(ctx, v_b, nd * sizeof(uint32_t)));
        const uint32_t vbits = cr_ceil_log2(nd ? nd : 1);  // bits the distinct-key index needs inside the payload
        uint32_t n_cand32 = 0;
        if (overlap) CR_TRY(fork.side(fork.t0));  // from here to join(): ctx->stream is the second stream
        // default: a flag per key + the count / write compaction.  CRGPU_CAND_EMIT=1: the filter kernel appends the (hash, index)
        // pairs itself -- measured SLOWER at 1 B records (k_group_candidates 2.6 -> 7.4 ms for the 2.6 ms of compaction it saves:
        // a third pass over every barcode range, whose longest ones are one workgroup's job), kept for the A/B record
        const bool cand_flags = getenv("CRGPU_CAND_EMIT") == nullptr;
        // CRGPU_CAND_FILTER=2: two table entries per group.  Measured at 1 B records: the filter kernel 2.6 -> 3.8 ms, the hash sort
        // only 20 % shorter (the large cells saturate the table either way), k_cp_count 1.6 -> 0.5: no change of the stage
        const char *cf2 = getenv("CRGPU_CAND_FILTER");
        const bool two_entries = cf2 && atoi(cf2) == 2;
        unsigned long long *d_ncand = (unsigned long long *)(ctx->d_scalars + 64);
        {
            CrTimer t(ctx, CRGPU_T_DEDUP);
            const uint64_t n_ftiles = (nd + LF_TILE - 1) / LF_TILE;
            if (cand_flags) {
                hipLaunchKernelGGL(k_group_candidates<false>, dim3(cr_grid(n_ftiles, 1, 256u * 2u)), dim3(LF_THREADS), 0, ctx->stream, kl,
                                   ukey, nd, cand_b.as<uint8_t>(), CandEmit{}, two_entries);
                CR_HIP(ctx, hipGetLastError());
                CR_TRY(compact(ctx, CandFlag{cand_b.as<uint8_t>()}, EmitHash{kl, ukey, vbits, h_b.as<uint32_t>(), v_b.as<uint32_t>()}, nd,
                               d_block, d_total));
            } else {
                CR_HIP(ctx, hipMemsetAsync(d_ncand, 0, sizeof(unsigned long long), ctx->stream));
                hipLaunchKernelGGL(k_group_candidates<true>, dim3(cr_grid(n_ftiles, 1, 256u * 2u)), dim3(LF_THREADS), 0, ctx->stream, kl,
                                   ukey, nd, (uint8_t *)nullptr, CandEmit{h_b.as<uint32_t>(), v_b.as<uint32_t>(), d_ncand, vbits}, two_entries);
                CR_HIP(ctx, hipGetLastError());
            }
        }
        if (!cand_flags) {
            unsigned long long nc64 = 0;
            CR_TRY(crgpu_memcpy_d2h(ctx, &nc64, d_ncand, sizeof(nc64)));
            n_cand32 = (uint32_t)nc64;
        } else
        CR_TRY(read_u32(ctx, d_total, &n_cand32));
        const uint64_t n_cand = n_cand32;
        ctx->last_distinct_keys = nd;
        ctx->last_low_support_candidates = n_cand;
        if (n_cand >= 2) {
            DevBuf ht_b, vt_b;
            CR_TRY(dmalloc(ctx, ht_b, n_cand * sizeof(uint32_t)));
            CR_TRY(dmalloc(ctx, vt_b, n_cand * sizeof(uint32_t)));
            bool s_in_tmp = false;
            CR_TRY(cr_radix_sort_u32(ctx, h_b.as<uint32_t>(), ht_b.as<uint32_t>(), v_b.as<uint32_t>(), vt_b.as<uint32_t>(), n_cand,
                                     0, LS_HASH_BITS, &s_in_tmp));
            CR_TRY(side_edges());
            CR_TRY(fork.join());  // k_low_support compares the phase-1 counts: it needs both branches
            {
                CrTimer t(ctx, CRGPU_T_DEDUP);
                hipLaunchKernelGGL(k_low_support, dim3(cr_grid(n_cand, 256)), dim3(256), 0, ctx->stream, kl,
                                   s_in_tmp ? ht_b.as<uint32_t>() : h_b.as<uint32_t>(),
                                   s_in_tmp ? vt_b.as<uint32_t>() : v_b.as<uint32_t>(), n_cand, nd, vbits, ukey, upos, n_keys, st);
                CR_HIP(ctx, hipGetLastError());
            }
        }
        CR_TRY(side_edges());
        CR_TRY(fork.join());  // (no candidates: nothing was joined above)
    }

    const TargetFilter tf{ctx->d_on_target, ctx->n_target_features, L.sh_feat(), L.bits_feat, ctx->d_on_target ? ctx->target_min_reads : 0};
    // 5b. optional per-read DupInfo.  Its scattered stores and the molecule / triplet passes of step 5 and 6 only share
    //     inputs: with the direct scatter the whole of 5b goes to the second stream and is joined at the end of the call
    //     (the ledger then books 5b + 5 + 5c + 6 as one span)
    CrFork fork2(ctx);
    DevBuf rep_b, packed_b;
    const bool windowed = getenv("CRGPU_DUPINFO_WINDOWED") != nullptr;
    const bool want_probe = vals && pr.d_probe;
    if (vals) CR_TRY(dmalloc(ctx, rep_b, nd * sizeof(uint32_t)));
    if (want_probe) {  // the molecule pass of step 5 needs the representative reads, too: before the streams part
        CrTimer t(ctx, CRGPU_T_DEDUP);
        hipLaunchKernelGGL(k_rep_read, dim3(cr_grid(nd, 256)), dim3(256), 0, ctx->stream, keys, vals, upos, nd, n_keys,
                           rep_b.as<uint32_t>());
        CR_HIP(ctx, hipGetLastError());
    }
    if (vals && !windowed && CrFork::enabled(ctx, nd)) {
        CR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        if (ctx->timing) {
            fork2.t0 = cr_take_event(ctx);
            fork2.t1 = cr_take_event(ctx);
            CR_HIP(ctx, hipEventRecord(fork2.t0, ctx->stream));
        }
        CR_TRY(fork2.side(fork2.t0));
    }
    // Beside the main stream these grid-stride kernels take 6 of the 8 workgroups a CU holds: with all 8 every wave slot was
    // theirs for their whole run time, and the main stream's next (often tiny) kernel waited milliseconds for one to end
    uint32_t side_wgs = fork2.on_side ? 256u * 6u : 256u * 8u;
    if (const char *g = getenv("CRGPU_SIDE_WGS")) side_wgs = 256u * (uint32_t)atoi(g);  // (A/B switch)
    if (side_wgs < 256u) side_wgs = 256u * 8u;
    if (vals) {
        CrTimer t(ctx, CRGPU_T_DEDUP);
        if (!want_probe)
            hipLaunchKernelGGL(k_rep_read, dim3(cr_grid(nd, 256, side_wgs)), dim3(256), 0, ctx->stream, keys, vals, upos, nd, n_keys,
                               rep_b.as<uint32_t>());
        // reads that never reach DupBuilder::observe get no DupInfo (mark_dups.rs:289-291): zeros.  The direct scatter into our
        // own packed temporary is followed by k_unpack_dupinfo, which writes all three arrays for EVERY read (zeros where the
        // temporary holds none): zeroing them first was 9 GB of fills per 1 B reads in front of the scatter.
        const bool unpack_covers_all = !windowed && !pr.packed_out;
        if (pr.packed_out) {
            CR_HIP(ctx, hipMemsetAsync(pr.packed_out, 0, pr.n_reads * sizeof(DupRec), ctx->stream));
        } else if (!unpack_covers_all) {
            if (pr.out_umi) CR_HIP(ctx, hipMemsetAsync(pr.out_umi, 0, pr.n_reads * sizeof(uint32_t), ctx->stream));
            if (pr.out_cnt) CR_HIP(ctx, hipMemsetAsync(pr.out_cnt, 0, pr.n_reads * sizeof(uint32_t), ctx->stream));
            if (pr.out_flags) CR_HIP(ctx, hipMemsetAsync(pr.out_flags, 0, pr.n_reads, ctx->stream));
        }
        // The direct scatter is the default.  CRGPU_DUPINFO_WINDOWED=1 takes the windowed one (k_per_read_sorted): measured at
        // 1 B records it is SLOWER (dedup family 119-138 ms against 77 ms, profiles/r02_dupinfo_windowed_ab.txt): 512 windows of
        // 2 M reads are 24 MB of output each, far beyond the 4 MB L2 of an XCD, and the three output arrays take three
        // scattered stores per read instead of one.  Kept as a tested experiment for windows that fit the L2 (two levels).
        bool direct = !windowed;
        if (!direct) {
            // CRGPU_DUPINFO_WINDOWED=2: two stable partition passes (18 bits of the ordinal, low digit first) leave windows of
            // 2^(bits - 18) reads -- 32 KB of output per array at 1 B reads, which the L2 merges into whole lines
            const char *wenv = getenv("CRGPU_DUPINFO_WINDOWED");
            const bool two_level = wenv && wenv[0] == '2';
            DevBuf prec_b, prec2_b, ord2_b, ord3_b;
            CR_TRY(dmalloc(ctx, prec_b, n_keys * sizeof(uint64_t)));
            CR_TRY(dmalloc(ctx, prec2_b, n_keys * sizeof(uint64_t)));
            CR_TRY(dmalloc(ctx, ord2_b, n_keys * sizeof(uint32_t)));
            if (two_level) CR_TRY(dmalloc(ctx, ord3_b, n_keys * sizeof(uint32_t)));
            uint32_t *d_over = ctx->d_scalars + 56, over = 0;
            CR_HIP(ctx, hipMemsetAsync(d_over, 0, sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(k_per_read_sorted, dim3(cr_grid(nd, 256)), dim3(256), 0, ctx->stream, kl, ukey, vals, upos, nd, n_keys,
                               corr, inc_all, st, minidx, rep_b.as<uint32_t>(), prec_b.as<uint64_t>(), d_over, tf);
            CR_HIP(ctx, hipGetLastError());
            CR_TRY(read_u32(ctx, d_over, &over));
            if (over) {
                direct = true;
            } else {
                const uint32_t bits = cr_ceil_log2(pr.n_reads ? pr.n_reads : 1);
                if (two_level && bits > 18) {
                    CR_TRY(cr_partition_by_payload(ctx, prec_b.as<uint64_t>(), prec2_b.as<uint64_t>(), vals, ord2_b.as<uint32_t>(), n_keys,
                                                   bits - 18));
                    CR_TRY(cr_partition_by_payload(ctx, prec2_b.as<uint64_t>(), prec_b.as<uint64_t>(), ord2_b.as<uint32_t>(),
                                                   ord3_b.as<uint32_t>(), n_keys, bits - 9));
                    hipLaunchKernelGGL(k_scatter_records, dim3(cr_grid(n_keys, 256)), dim3(256), 0, ctx->stream, prec_b.as<uint64_t>(),
                                       ord3_b.as<uint32_t>(), n_keys, pr.out_umi, pr.out_cnt, pr.out_flags, pr.packed_out);
                } else {
                    CR_TRY(cr_partition_by_payload(ctx, prec_b.as<uint64_t>(), prec2_b.as<uint64_t>(), vals, ord2_b.as<uint32_t>(), n_keys,
                                                   bits > 9 ? bits - 9 : 0));
                    hipLaunchKernelGGL(k_scatter_records, dim3(cr_grid(n_keys, 256)), dim3(256), 0, ctx->stream, prec2_b.as<uint64_t>(),
                                       ord2_b.as<uint32_t>(), n_keys, pr.out_umi, pr.out_cnt, pr.out_flags, pr.packed_out);
                }
                CR_HIP(ctx, hipGetLastError());
            }
        }
        if (direct) {
            // our own temporary may use the 8-byte layout
            const bool pack8 = !pr.packed_out && kl.bits_umi <= 24u && !getenv("CRGPU_DUPINFO_PACK12");
            if (!pr.packed_out) {
                const uint64_t bytes = pr.n_reads * (pack8 ? sizeof(DupRec8) : sizeof(DupRec));
                CR_TRY(dmalloc(ctx, packed_b, bytes));
                CR_HIP(ctx, hipMemsetAsync(packed_b.p, 0, bytes, ctx->stream));
            }
            DupRec *packed = pr.packed_out ? pr.packed_out : packed_b.as<DupRec>();
            if (pack8)
                hipLaunchKernelGGL(k_per_read<true>, dim3(cr_grid(nd, 256, side_wgs)), dim3(256), 0, ctx->stream, kl, ukey, vals, upos, nd, n_keys,
                                   corr, inc_all, st, minidx, rep_b.as<uint32_t>(), (void *)packed, tf);
            else
                hipLaunchKernelGGL(k_per_read<false>, dim3(cr_grid(nd, 256, side_wgs)), dim3(256), 0, ctx->stream, kl, ukey, vals, upos, nd, n_keys,
                                   corr, inc_all, st, minidx, rep_b.as<uint32_t>(), (void *)packed, tf);
            if (!pr.packed_out) {
                if (pack8)
                    hipLaunchKernelGGL(k_unpack_dupinfo<true>, dim3(cr_grid(pr.n_reads, 256, side_wgs)), dim3(256), 0, ctx->stream,
                                       (const void *)packed, pr.n_reads, pr.out_umi, pr.out_cnt, pr.out_flags);
                else
                    hipLaunchKernelGGL(k_unpack_dupinfo<false>, dim3(cr_grid(pr.n_reads, 256, side_wgs)), dim3(256), 0, ctx->stream,
                                       (const void *)packed, pr.n_reads, pr.out_umi, pr.out_cnt, pr.out_flags);
            }
            CR_HIP(ctx, hipGetLastError());
        }
    }
    fork2.pause();  // the rest of the call is enqueued on the main stream again; joined before the return

    // 5. molecules = distinct keys some read lands on and that are not low support; with them, in the same two launches,
    //    the (barcode, feature) triplets (6.): CRGPU_MOL_FUSED=0 keeps the separate passes (the A/B path)
    DevBuf mkeys_b, mreads_b, tpos_b;
    CR_TRY(dmalloc(ctx, mkeys_b, nd * sizeof(uint64_t)));
    CR_TRY(dmalloc(ctx, mreads_b, nd * sizeof(uint32_t)));
    uint32_t nm32 = 0, nt32 = 0;
    if (want_probe) CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_mprobe, nd * sizeof(int32_t)));
    const char *fused_env = getenv("CRGPU_MOL_FUSED");
    const bool fused = nd > 0 && !(fused_env && fused_env[0] == '0');
    if (fused) {
        CR_TRY(dmalloc(ctx, tpos_b, (nd + 1) * sizeof(uint32_t)));
        // the triplet arrays are sized before the number of triplets is known: one entry per distinct key is the bound
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_bc, nd * sizeof(uint32_t)));
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_feature, nd * sizeof(uint32_t)));
        uint32_t *d_heads = d_block + 4096;      // the second set of block counters
        uint32_t *d_total_h = ctx->d_scalars + 17;
        {
            CrTimer t(ctx, CRGPU_T_DEDUP);
            EmitMol emit{ukey, upos, inc_all, minidx, st, n_keys, nd, mkeys_b.as<uint64_t>(), mreads_b.as<uint32_t>()};
            if (want_probe) {
                emit.rep_read = rep_b.as<uint32_t>();
                emit.probe = pr.d_probe;
                emit.mprobe = res->d_mprobe;
            }
            uint64_t tile;
            const uint32_t nb = cp_blocks(nd, &tile);
            const MolFlagTargeted flag_t{st, ukey, upos, inc_all, n_keys, nd, tf};
            const MolFlag flag_p{st};
            if (tf.min_reads)
                hipLaunchKernelGGL(k_mt_count<MolFlagTargeted>, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag_t, ukey, L.sh_feat(), nd,
                                   tile, d_block, d_heads);
            else
                hipLaunchKernelGGL(k_mt_count<MolFlag>, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag_p, ukey, L.sh_feat(), nd, tile,
                                   d_block, d_heads);
            CR_TRY(cr_scan_small(ctx, d_block, nb, d_total));
            CR_TRY(cr_scan_small(ctx, d_heads, nb, d_total_h));
            if (tf.min_reads)
                hipLaunchKernelGGL(k_mt_write<MolFlagTargeted>, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag_t, emit, L.sh_bc(),
                                   L.sh_feat(), L.bits_feat, nd, tile, d_block, d_heads, res->d_bc, res->d_feature, tpos_b.as<uint32_t>(),
                                   res->d_back);
            else
                hipLaunchKernelGGL(k_mt_write<MolFlag>, dim3(nb), dim3(CP_BLOCK), 0, ctx->stream, flag_p, emit, L.sh_bc(), L.sh_feat(),
                                   L.bits_feat, nd, tile, d_block, d_heads, res->d_bc, res->d_feature, tpos_b.as<uint32_t>(), res->d_back);
            CR_HIP(ctx, hipGetLastError());
        }
        CR_TRY(read_u32(ctx, d_total, &nm32));
        CR_TRY(read_u32(ctx, d_total_h, &nt32));
    } else {
        CrTimer t(ctx, CRGPU_T_DEDUP);
        EmitMol emit{ukey, upos, inc_all, minidx, st, n_keys, nd, mkeys_b.as<uint64_t>(), mreads_b.as<uint32_t>()};
        if (want_probe) {
            emit.rep_read = rep_b.as<uint32_t>();
            emit.probe = pr.d_probe;
            emit.mprobe = res->d_mprobe;
        }
        if (tf.min_reads)
            CR_TRY(compact(ctx, MolFlagTargeted{st, ukey, upos, inc_all, n_keys, nd, tf}, emit, nd, d_block, d_total));
        else
            CR_TRY(compact(ctx, MolFlag{st}, emit, nd, d_block, d_total));
    }
    if (!fused) CR_TRY(read_u32(ctx, d_total, &nm32));
    const uint64_t nm = nm32;

    // 5c. optional: reads with a corrected UMI per (library, barcode) -- the one BarcodeSummary column that cannot be
    //     derived from the molecule table afterwards
    if (pr.summary) {
        const size_t bytes = ((size_t)1 << L.bits_lib) * ctx->n_canon * sizeof(uint32_t);
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_corr_reads, bytes));
        CrTimer t(ctx, CRGPU_T_DEDUP);
        CR_HIP(ctx, hipMemsetAsync(res->d_corr_reads, 0, bytes, ctx->stream));
        hipLaunchKernelGGL(k_corrected_reads, dim3(cr_grid(nd, 256)), dim3(256), 0, ctx->stream, kl, ukey, upos, nd, n_keys, st,
                           ctx->n_canon, res->d_corr_reads, res->d_back);
        if (tf.min_reads) {
            CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_filt_reads, bytes));
            CR_HIP(ctx, hipMemsetAsync(res->d_filt_reads, 0, bytes, ctx->stream));
            hipLaunchKernelGGL(k_filtered_reads, dim3(cr_grid(nd, 256)), dim3(256), 0, ctx->stream, kl, ukey, upos, nd, n_keys, st,
                               inc_all, tf, ctx->n_canon, res->d_filt_reads, res->d_back);
        }
        CR_HIP(ctx, hipGetLastError());
    }

    // 6. (barcode, feature) triplets = run lengths of the molecule keys at the feature boundary
    if (!fused) {
        CR_TRY(dmalloc(ctx, tpos_b, (nm + 1) * sizeof(uint32_t)));
        if (nm) {
            {
                CrTimer t(ctx, CRGPU_T_DEDUP);
                CR_TRY(compact(ctx, HeadFlag{mkeys_b.as<uint64_t>(), L.sh_feat()}, EmitTriplet{mkeys_b.as<uint64_t>(), tpos_b.as<uint32_t>()},
                               nm, d_block, d_total));
            }
            CR_TRY(read_u32(ctx, d_total, &nt32));
        }
    }
    const uint64_t nt = nt32;
    if (!fused) {
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_bc, nt * sizeof(uint32_t)));
        CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_feature, nt * sizeof(uint32_t)));
    }
    CR_TRY(cr_pool_alloc(ctx, (void **)&res->d_count, nt * sizeof(uint32_t)));
    if (nt) {
        CrTimer t(ctx, CRGPU_T_DEDUP);
        if (fused)
            hipLaunchKernelGGL(k_trip_counts, dim3(cr_grid(nt, 256)), dim3(256), 0, ctx->stream, tpos_b.as<uint32_t>(), nt, nm,
                               res->d_count);
        else
            hipLaunchKernelGGL(k_triplets, dim3(cr_grid(nt, 256)), dim3(256), 0, ctx->stream, kl, mkeys_b.as<uint64_t>(),
                               tpos_b.as<uint32_t>(), nt, nm, res->d_bc, res->d_feature, res->d_count, res->d_back);
        CR_HIP(ctx, hipGetLastError());
    }
    CR_TRY(fork2.join());
    res->n_triplets = nt;
    res->n_molecules = nm;
    res->d_mkeys = (uint64_t *)mkeys_b.p;
    res->d_mreads = (uint32_t *)mreads_b.p;
    mkeys_b.p = nullptr;
    mreads_b.p = nullptr;
    guard.armed = false;
    *out = res;
    return CRGPU_OK;
}

extern "C" int crgpu_count_keys_dev(crgpu_ctx *ctx, uint64_t *d_keys_inout, uint64_t n_keys, crgpu_counts **out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    PerRead pr;
    pr.summary = ctx->barcode_summary_on;
    return count_keys_impl(ctx, d_keys_inout, n_keys, out, pr);
}

extern "C" int crgpu_count_records_dev(crgpu_ctx *ctx, const crgpu_records *recs, crgpu_counts **out,
                                       uint32_t *d_processed_umi_out, uint32_t *d_read_count_out, uint8_t *d_dupflags_out) {
    if (!ctx || !recs || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, recs->n <= 0x7FFFFFFFull, CRGPU_ERANGE, "crgpu_count_records: at most 2^31-1 records per call");
    const uint64_t n = recs->n;
    DevBuf keys_b, vals_b;
    CR_TRY(dmalloc(ctx, keys_b, n * sizeof(uint64_t)));
    CR_TRY(dmalloc(ctx, vals_b, n * sizeof(uint32_t)));
    uint64_t n_keys = 0;
    CR_TRY(build_keys_impl(ctx, recs, keys_b.as<uint64_t>(), vals_b.as<uint32_t>(), &n_keys));
    PerRead pr;
    pr.summary = true;
    pr.n_reads = n;
    pr.d_vals = vals_b.as<uint32_t>();
    pr.out_umi = d_processed_umi_out;
    pr.out_cnt = d_read_count_out;
    pr.out_flags = d_dupflags_out;
    pr.d_probe = recs->d_probe_idx;
    if (n_keys == 0) {  // no read reaches DupBuilder::observe: no DupInfo anywhere (mark_dups.rs:289-291)
        if (d_processed_umi_out) CR_HIP(ctx, hipMemsetAsync(d_processed_umi_out, 0, n * sizeof(uint32_t), ctx->stream));
        if (d_read_count_out) CR_HIP(ctx, hipMemsetAsync(d_read_count_out, 0, n * sizeof(uint32_t), ctx->stream));
        if (d_dupflags_out) CR_HIP(ctx, hipMemsetAsync(d_dupflags_out, 0, n, ctx->stream));
    }
    return count_keys_impl(ctx, keys_b.as<uint64_t>(), n_keys, out, pr);
}

// ------------------------------------------------------------------------------------------------
// one GEM well on several GPUs, with per-read DupInfo (SURVEY 8e + mark_dups.rs:61-72)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_iota_u32(uint32_t *p, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}
// the records that came back from the owners, in the order of this rank's partitioned keys -> the reads they belong to
__global__ __launch_bounds__(256) void k_scatter_dupinfo(const DupRec *__restrict__ rec, const uint32_t *__restrict__ ordinal,
                                                         uint64_t n_keys, uint32_t *__restrict__ out_umi,
                                                         uint32_t *__restrict__ out_cnt, uint8_t *__restrict__ out_flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_keys; j += stride) {
        const DupRec d = rec[j];
        const uint32_t r = ordinal[j];
        if (out_umi) out_umi[r] = d.umi;
        if (out_cnt) out_cnt[r] = d.read_count;
        if (out_flags) out_flags[r] = (uint8_t)d.flags;
    }
}

// Collective.  Every rank passes its shard of the well's records (contiguous slices of the read stream in rank order:
// the qname rank of a read is its position in that stream) and gets back (a) the counts of the barcode range it owns,
// as crgpu_exchange_keys_dev + crgpu_count_keys_dev would give them, and (b) the DupInfo of ITS OWN reads:
//   keys + ordinals -> stable partition by owner -> keys to the owners (C2) -> dedup there with the position in the
//   receive buffer as ordinal (source-rank-major, stable == the well's read order) -> the packed 12-byte records travel
//   back along the same routes -> scattered to the reads by the ordinals that stayed at home.
extern "C" int crgpu_count_records_sharded_dev(crgpu_ctx *ctx, const crgpu_records *recs, crgpu_counts **out,
                                               uint32_t *d_processed_umi_out, uint32_t *d_read_count_out,
                                               uint8_t *d_dupflags_out) {
    if (!ctx || !recs || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, !recs->d_probe_idx, CRGPU_EINVAL,
               "crgpu_count_records_sharded: d_probe_idx is not carried across ranks (join probe indices on the host by the "
               "is_umi_count reads)");
    cr_invalidate(ctx);
    CR_REQUIRE(ctx, recs->n <= 0x7FFFFFFFull, CRGPU_ERANGE, "crgpu_count_records_sharded: at most 2^31-1 records per call");
    const uint64_t n = recs->n;
    const int W = ctx->n_ranks;
    DevBuf keys_b, vals_b, pkeys_b, pvals_b, recv_b, rrec_b, brec_b, iota_b;
    // Local steps only RECORD their status in front of a collective; the status travels with the count exchange and all ranks
    // leave together (a rank that returned early would leave its peers waiting inside RCCL for ever: comm.hip)
    uint64_t n_keys = 0;
    std::vector<uint32_t> bounds(W + 1);
    std::vector<uint64_t> send_cnt(W, 0), all((size_t)W * W, 0);
    auto prepare = [&]() -> int {
        CR_TRY(dmalloc(ctx, keys_b, (n ? n : 1) * sizeof(uint64_t)));
        CR_TRY(dmalloc(ctx, vals_b, (n ? n : 1) * sizeof(uint32_t)));
        CR_TRY(build_keys_impl(ctx, recs, keys_b.as<uint64_t>(), vals_b.as<uint32_t>(), &n_keys));
        // C2 with the ordinals kept at home
        CR_TRY(crgpu_balanced_bounds(ctx, (uint32_t)W, bounds.data()));
        CR_TRY(dmalloc(ctx, pkeys_b, (n_keys ? n_keys : 1) * sizeof(uint64_t)));
        CR_TRY(dmalloc(ctx, pvals_b, (n_keys ? n_keys : 1) * sizeof(uint32_t)));
        CR_TRY(cr_comm_test_failure(ctx));
        return cr_partition_by_owner_kv(ctx, keys_b.as<uint64_t>(), pkeys_b.as<uint64_t>(), vals_b.as<uint32_t>(), pvals_b.as<uint32_t>(),
                                        n_keys, ctx->layout.sh_bc(), (uint32_t)W, bounds.data(), send_cnt.data());
    };
    CR_TRY(cr_comm_exchange_counts(ctx, prepare(), send_cnt.data(), all.data(), 0x7FFFFFFFull, "crgpu_count_records_sharded"));
    std::vector<uint64_t> soff(W), sbytes(W), roff(W), rbytes(W);
    uint64_t n_recv = 0, so = 0;
    for (int p = 0; p < W; p++) {
        soff[p] = so;
        sbytes[p] = send_cnt[p];
        so += send_cnt[p];
        roff[p] = n_recv;
        rbytes[p] = all[(size_t)p * W + ctx->rank];
        n_recv += rbytes[p];
    }
    auto scaled = [&](const std::vector<uint64_t> &v, uint64_t f) {
        std::vector<uint64_t> o(v.size());
        for (size_t i = 0; i < v.size(); i++) o[i] = v[i] * f;
        return o;
    };
    CR_TRY(cr_comm_agree(ctx, dmalloc(ctx, recv_b, (n_recv ? n_recv : 1) * sizeof(uint64_t)), "crgpu_count_records_sharded"));
    {
        CrTimer t(ctx, CRGPU_T_COMM, n_keys);
        CR_TRY(cr_comm_alltoallv(ctx, pkeys_b.p, scaled(soff, 8).data(), scaled(sbytes, 8).data(), recv_b.p, scaled(roff, 8).data(),
                                 scaled(rbytes, 8).data()));
        CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->comm_bytes[1] += n_keys * sizeof(uint64_t);
    }
    // dedup of the owned range; the records of the received keys come back packed, in receive order
    auto dedup_owned = [&]() -> int {
        CR_TRY(dmalloc(ctx, rrec_b, (n_recv ? n_recv : 1) * sizeof(DupRec)));
        CR_TRY(dmalloc(ctx, iota_b, (n_recv ? n_recv : 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_iota_u32, dim3(cr_grid(n_recv ? n_recv : 1, 256)), dim3(256), 0, ctx->stream, iota_b.as<uint32_t>(), n_recv);
        PerRead pr;
        pr.summary = true;
        pr.n_reads = n_recv;
        pr.d_vals = iota_b.as<uint32_t>();
        pr.packed_out = rrec_b.as<DupRec>();
        CR_TRY(count_keys_impl(ctx, recv_b.as<uint64_t>(), n_recv, out, pr));
        // the records travel home along the reversed routes
        return dmalloc(ctx, brec_b, (n_keys ? n_keys : 1) * sizeof(DupRec));
    };
    {
        int rc = cr_comm_agree(ctx, dedup_owned(), "crgpu_count_records_sharded");
        if (rc == CRGPU_OK) {
            CrTimer t(ctx, CRGPU_T_COMM, n_recv);
            const uint64_t rs = sizeof(DupRec);
            rc = cr_comm_alltoallv(ctx, rrec_b.p, scaled(roff, rs).data(), scaled(rbytes, rs).data(), brec_b.p, scaled(soff, rs).data(),
                                   scaled(sbytes, rs).data());
            if (rc == CRGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "sync failed");
        }
        if (rc != CRGPU_OK) {
            if (*out) crgpu_counts_free(ctx, *out);
            *out = nullptr;
            return rc;
        }
    }
    CrTimer t(ctx, CRGPU_T_DEDUP);
    if (d_processed_umi_out) CR_HIP(ctx, hipMemsetAsync(d_processed_umi_out, 0, n * sizeof(uint32_t), ctx->stream));
    if (d_read_count_out) CR_HIP(ctx, hipMemsetAsync(d_read_count_out, 0, n * sizeof(uint32_t), ctx->stream));
    if (d_dupflags_out) CR_HIP(ctx, hipMemsetAsync(d_dupflags_out, 0, n, ctx->stream));
    if (n_keys)
        hipLaunchKernelGGL(k_scatter_dupinfo, dim3(cr_grid(n_keys, 256)), dim3(256), 0, ctx->stream, brec_b.as<DupRec>(),
                           pvals_b.as<uint32_t>(), n_keys, d_processed_umi_out, d_read_count_out, d_dupflags_out);
    CR_HIP(ctx, hipGetLastError());
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the temporaries above go back to the pool
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// K6 on the device: barcode index + CSC (barcode_index.rs:20-53, count_matrix.rs:382-448)
// ------------------------------------------------------------------------------------------------
struct CountTables {
    const uint32_t *t[2 * CRGPU_MAX_LIB];
    uint32_t n;
};
struct SeenFlag {  // barcode has a non-zero valid or corrected count in some library
    CountTables ct;
    __device__ __forceinline__ bool operator()(uint64_t r) const {
        uint32_t any = 0;
        for (uint32_t k = 0; k < ct.n; k++) any |= ct.t[k][r];
        return any != 0u;
    }
};
struct EmitCol {
    uint32_t *rank;
    struct Pre {};
    __device__ __forceinline__ Pre pre(uint64_t) const { return Pre(); }
    __device__ __forceinline__ void operator()(uint64_t r, uint32_t o, Pre) const { rank[o] = (uint32_t)r; }
};

// ---- CRGPU_OPT_DENSE_BARCODE_KEYS: the BarcodeIndex of the tables as they stand (cr_types/src/barcode_index.rs:20-53) ------
int cr_dense_ensure(crgpu_ctx *ctx) {
    DenseIndex &D = ctx->dense;
    if (!D.on || D.valid) return CRGPU_OK;
    CR_REQUIRE(ctx, ctx->canon_set && ctx->layout.set, CRGPU_ESTATE, "dense barcode keys: whitelist and key layout first");
    const uint32_t W = ctx->n_canon;
    SeenFlag seen;
    seen.ct.n = 0;
    for (int l = 0; l < CRGPU_MAX_LIB; l++)
        if (ctx->wl[l].set) {
            seen.ct.t[seen.ct.n++] = ctx->wl[l].d_valid;
            seen.ct.t[seen.ct.n++] = ctx->wl[l].d_corrected;
        }
    if (!D.d_fwd) CR_HIP(ctx, hipMalloc((void **)&D.d_fwd, (((size_t)W + 63) / 64) * sizeof(uint4)));
    if (!D.d_back) CR_HIP(ctx, hipMalloc((void **)&D.d_back, (size_t)W * sizeof(uint32_t)));
    uint32_t *d_total = ctx->d_scalars + 16;
    uint32_t V = 0;
    {
        CrTimer t(ctx, CRGPU_T_KEYS);
        CR_TRY(compact(ctx, seen, EmitCol{D.d_back}, W, ctx->d_sort_hist, d_total));
    }
    CR_TRY(read_u32(ctx, d_total, &V));
    KeyLayout L = ctx->layout;
    L.bits_bc = V > 1 ? cr_ceil_log2(V) : 1u;
    CR_REQUIRE(ctx, L.total_bits() <= 64, CRGPU_ERANGE,
               "molecule key needs %u bits even with %u barcodes in the index (barcode %u + feature %u + library %u + umi %u + 1) > 64",
               L.total_bits(), V, L.bits_bc, L.bits_feat, L.bits_lib, L.bits_umi + L.bits_ulen);
    D.h_back.resize(V);
    if (V) CR_TRY(crgpu_memcpy_d2h(ctx, D.h_back.data(), D.d_back, (uint64_t)V * sizeof(uint32_t)));
    {   // rank/select table on the host (V set bits, W / 64 words): 16 bytes per 64 ranks
        const size_t n_words = ((size_t)W + 63) / 64;
        std::vector<uint4> t(n_words, make_uint4(0u, 0u, 0u, 0u));
        for (uint32_t c = 0; c < V; c++) {
            const uint32_t r = D.h_back[c];
            if (r & 32u) t[r >> 6].y |= 1u << (r & 31u); else t[r >> 6].x |= 1u << (r & 31u);
        }
        uint32_t run = 0;
        for (size_t w = 0; w < n_words; w++) {
            t[w].z = run;
            run += (uint32_t)__builtin_popcount(t[w].x) + (uint32_t)__builtin_popcount(t[w].y);
        }
        CR_TRY(crgpu_memcpy_h2d(ctx, D.d_fwd, t.data(), n_words * sizeof(uint4)));
    }
    D.V = V;
    D.valid = true;
    ctx->layout = L;
    ctx->ghist.valid = false;
    return CRGPU_OK;
}

__global__ __launch_bounds__(256) void k_csc(const uint32_t *__restrict__ col_rank, uint64_t n_cols,
                                             const uint32_t *__restrict__ t_bc, const uint32_t *__restrict__ t_feat,
                                             const uint32_t *__restrict__ t_cnt, uint64_t nt,
                                             long long *__restrict__ indptr, int32_t *__restrict__ indices,
                                             int32_t *__restrict__ data) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // indptr[c] = first triplet whose barcode rank >= col_rank[c]; columns without counts get empty ranges
    for (uint64_t c = tid; c <= n_cols; c += stride) {
        if (c == n_cols) {
            indptr[c] = (long long)nt;
            continue;
        }
        const uint32_t r = col_rank[c];
        uint64_t lo = 0, hi = nt;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (t_bc[mid] < r) lo = mid + 1; else hi = mid;
        }
        indptr[c] = (long long)lo;
    }
    for (uint64_t i = tid; i < nt; i += stride) {
        indices[i] = (int32_t)t_feat[i];
        data[i] = (int32_t)t_cnt[i];
    }
}

struct MatrixDevImpl {
    crgpu_matrix_dev view;
    uint32_t *d_rank = nullptr;
    long long *d_indptr = nullptr;
    int32_t *d_indices = nullptr, *d_data = nullptr;
};

extern "C" int crgpu_assemble_matrix_dev(crgpu_ctx *ctx, const uint32_t *d_bc, const uint32_t *d_feature,
                                         const uint32_t *d_count, uint64_t n_triplets, crgpu_matrix_dev **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_assemble_matrix_dev: no whitelist set");
    CR_REQUIRE(ctx, n_triplets == 0 || (d_bc && d_feature && d_count), CRGPU_EINVAL, "NULL triplets");
    CR_REQUIRE(ctx, n_triplets < 0xFFFFFFFFull, CRGPU_ERANGE, "too many triplets");
    SeenFlag seen;
    seen.ct.n = 0;
    for (int l = 0; l < CRGPU_MAX_LIB; l++)
        if (ctx->wl[l].set) {
            seen.ct.t[seen.ct.n++] = ctx->wl[l].d_valid;
            seen.ct.t[seen.ct.n++] = ctx->wl[l].d_corrected;
        }
    MatrixDevImpl *m = new (std::nothrow) MatrixDevImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    struct Guard {
        crgpu_ctx *c;
        MatrixDevImpl *m;
        bool armed = true;
        ~Guard() {
            if (armed) crgpu_matrix_dev_free(c, &m->view);
        }
    } guard{ctx, m};
    const uint32_t W = ctx->n_canon;
    CR_TRY(cr_pool_alloc(ctx, (void **)&m->d_rank, (size_t)W * sizeof(uint32_t)));
    uint32_t *d_total = ctx->d_scalars + 16;
    uint32_t V = 0;
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        CR_TRY(compact(ctx, seen, EmitCol{m->d_rank}, W, ctx->d_sort_hist, d_total));
    }
    CR_TRY(read_u32(ctx, d_total, &V));
    CR_TRY(cr_pool_alloc(ctx, (void **)&m->d_indptr, ((size_t)V + 1) * sizeof(long long)));
    CR_TRY(cr_pool_alloc(ctx, (void **)&m->d_indices, n_triplets * sizeof(int32_t)));
    CR_TRY(cr_pool_alloc(ctx, (void **)&m->d_data, n_triplets * sizeof(int32_t)));
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        const uint64_t work = n_triplets > V ? n_triplets : (uint64_t)V + 1;
        hipLaunchKernelGGL(k_csc, dim3(cr_grid(work, 256)), dim3(256), 0, ctx->stream, m->d_rank, (uint64_t)V, d_bc, d_feature,
                           d_count, n_triplets, m->d_indptr, m->d_indices, m->d_data);
        CR_HIP(ctx, hipGetLastError());
    }
    m->view.n_barcodes = V;
    m->view.nnz = n_triplets;
    m->view.d_barcode_rank = m->d_rank;
    m->view.d_indptr = (const int64_t *)m->d_indptr;
    m->view.d_indices = m->d_indices;
    m->view.d_data = m->d_data;
    guard.armed = false;
    *out = &m->view;
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// aggr-style post-processing on the device
// ------------------------------------------------------------------------------------------------
// merge of two index-sorted columns; WRITE = false counts the merged entries
template <bool WRITE>
__global__ __launch_bounds__(256) void k_sum_columns(uint64_t n_cols, const long long *__restrict__ pa, const int32_t *__restrict__ ia,
                                                     const int32_t *__restrict__ da, const long long *__restrict__ pb,
                                                     const int32_t *__restrict__ ib, const int32_t *__restrict__ db,
                                                     uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off,
                                                     int32_t *__restrict__ io, int32_t *__restrict__ dout) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += stride) {
        long long i = pa[c], j = pb[c];
        const long long ie = pa[c + 1], je = pb[c + 1];
        uint32_t o = WRITE ? off[c] : 0u;
        while (i < ie || j < je) {
            const int32_t ra = i < ie ? ia[i] : 0x7FFFFFFF, rb = j < je ? ib[j] : 0x7FFFFFFF;
            if (WRITE) {
                io[o] = ra < rb ? ra : rb;
                dout[o] = (ra <= rb ? da[i] : 0) + (rb <= ra ? db[j] : 0);
            }
            o++;
            i += ra <= rb;
            j += rb <= ra;
        }
        if (!WRITE) cnt[c] = o;
    }
}
__global__ __launch_bounds__(256) void k_ranks_differ(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n,
                                                      uint32_t *__restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (a[i] != b[i]) *flag = 1u;
}
__global__ __launch_bounds__(256) void k_offsets_to_indptr(const uint32_t *__restrict__ off, uint64_t n, uint32_t total,
                                                           long long *__restrict__ indptr) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride)
        indptr[i] = i < n ? (long long)off[i] : (long long)total;
}
__global__ __launch_bounds__(256) void k_selected_lengths(const long long *__restrict__ indptr, const uint64_t *__restrict__ cols,
                                                          uint64_t n_sel, const uint32_t *__restrict__ rank_in,
                                                          uint32_t *__restrict__ len, uint32_t *__restrict__ rank_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_sel; k += stride) {
        const uint64_t c = cols[k];
        len[k] = (uint32_t)(indptr[c + 1] - indptr[c]);
        rank_out[k] = rank_in[c];
    }
}
// one wave per selected column: a contiguous copy of its entries
__global__ __launch_bounds__(256) void k_copy_columns(const long long *__restrict__ indptr, const uint64_t *__restrict__ cols,
                                                      uint64_t n_sel, const uint32_t *__restrict__ off,
                                                      const int32_t *__restrict__ ia, const int32_t *__restrict__ da,
                                                      int32_t *__restrict__ io, int32_t *__restrict__ dout) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave0; k < n_sel; k += n_waves) {
        const uint64_t c = cols[k];
        const long long s = indptr[c], e = indptr[c + 1];
        const uint32_t o = off[k];
        for (long long i = s + lane; i < e; i += 64) {
            io[o + (i - s)] = ia[i];
            dout[o + (i - s)] = da[i];
        }
    }
}

static int new_matrix_dev(crgpu_ctx *ctx, uint64_t V, uint64_t nnz, MatrixDevImpl **out) {
    MatrixDevImpl *m = new (std::nothrow) MatrixDevImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    int rc = cr_pool_alloc(ctx, (void **)&m->d_rank, (V ? V : 1) * sizeof(uint32_t));
    if (rc == CRGPU_OK) rc = cr_pool_alloc(ctx, (void **)&m->d_indptr, (V + 1) * sizeof(long long));
    if (rc == CRGPU_OK) rc = cr_pool_alloc(ctx, (void **)&m->d_indices, (nnz ? nnz : 1) * sizeof(int32_t));
    if (rc == CRGPU_OK) rc = cr_pool_alloc(ctx, (void **)&m->d_data, (nnz ? nnz : 1) * sizeof(int32_t));
    if (rc != CRGPU_OK) {
        crgpu_matrix_dev_free(ctx, &m->view);
        return rc;
    }
    m->view.n_barcodes = V;
    m->view.nnz = nnz;
    m->view.d_barcode_rank = m->d_rank;
    m->view.d_indptr = (const int64_t *)m->d_indptr;
    m->view.d_indices = m->d_indices;
    m->view.d_data = m->d_data;
    *out = m;
    return CRGPU_OK;
}

// CountMatrix.merge / merge_matrices (lib/python/cellranger/matrix.py:479-482,1319-1329): element-wise sum of two matrices
// of the same shape -- here two device CSCs over the same columns (same barcode ranks in the same order).
extern "C" int crgpu_sum_matrices_dev(crgpu_ctx *ctx, const crgpu_matrix_dev *a, const crgpu_matrix_dev *b, crgpu_matrix_dev **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, a && b, CRGPU_EINVAL, "crgpu_sum_matrices_dev: NULL matrix");
    CR_REQUIRE(ctx, a->n_barcodes == b->n_barcodes, CRGPU_EINVAL, "crgpu_sum_matrices_dev: %llu vs %llu columns",
               (unsigned long long)a->n_barcodes, (unsigned long long)b->n_barcodes);
    CR_REQUIRE(ctx, a->nnz + b->nnz < 0xFFFFFFFFull, CRGPU_ERANGE, "crgpu_sum_matrices_dev: too many entries");
    const uint64_t V = a->n_barcodes;
    uint32_t *d_flag = ctx->d_scalars + 48, *d_total = ctx->d_scalars + 16;
    DevBuf cnt_b;
    CR_TRY(dmalloc(ctx, cnt_b, (V + 1) * sizeof(uint32_t)));
    uint32_t *cnt = cnt_b.as<uint32_t>();
    const long long *pa = (const long long *)a->d_indptr, *pb = (const long long *)b->d_indptr;
    uint32_t differ = 0, total = 0;
    {
        CrTimer t(ctx, CRGPU_T_MATRIX, a->nnz + b->nnz);
        CR_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ctx->stream));
        if (V) {
            hipLaunchKernelGGL(k_ranks_differ, dim3(cr_grid(V, 256)), dim3(256), 0, ctx->stream, a->d_barcode_rank, b->d_barcode_rank, V, d_flag);
            hipLaunchKernelGGL(k_sum_columns<false>, dim3(cr_grid(V, 256)), dim3(256), 0, ctx->stream, V, pa, a->d_indices, a->d_data, pb,
                               b->d_indices, b->d_data, cnt, (const uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
        }
        CR_HIP(ctx, hipGetLastError());
        CR_TRY(cr_scan_small(ctx, cnt, V, d_total));
    }
    CR_TRY(read_u32(ctx, d_flag, &differ));
    CR_REQUIRE(ctx, !differ, CRGPU_EINVAL, "crgpu_sum_matrices_dev: the matrices hold different barcodes");
    CR_TRY(read_u32(ctx, d_total, &total));
    MatrixDevImpl *m = nullptr;
    CR_TRY(new_matrix_dev(ctx, V, total, &m));
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        if (V) {
            CR_HIP(ctx, hipMemcpyAsync(m->d_rank, a->d_barcode_rank, V * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            hipLaunchKernelGGL(k_sum_columns<true>, dim3(cr_grid(V, 256)), dim3(256), 0, ctx->stream, V, pa, a->d_indices, a->d_data, pb,
                               b->d_indices, b->d_data, (uint32_t *)nullptr, cnt, m->d_indices, m->d_data);
        }
        hipLaunchKernelGGL(k_offsets_to_indptr, dim3(cr_grid(V + 1, 256)), dim3(256), 0, ctx->stream, cnt, V, total, m->d_indptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            crgpu_matrix_dev_free(ctx, &m->view);
            return cr_fail(ctx, CRGPU_EHIP, "crgpu_sum_matrices_dev: kernel failed");
        }
    }
    *out = &m->view;
    return CRGPU_OK;
}

// CountMatrix.select_barcodes (matrix.py:860-875): the columns `cols` (host array of column positions) in the given order.
extern "C" int crgpu_select_barcodes_dev(crgpu_ctx *ctx, const crgpu_matrix_dev *a, const uint64_t *cols, uint64_t n_cols,
                                         crgpu_matrix_dev **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, a && (cols || n_cols == 0), CRGPU_EINVAL, "crgpu_select_barcodes_dev: NULL argument");
    for (uint64_t k = 0; k < n_cols; k++)
        CR_REQUIRE(ctx, cols[k] < a->n_barcodes, CRGPU_EINVAL, "crgpu_select_barcodes_dev: column %llu out of range",
                   (unsigned long long)cols[k]);
    DevBuf cols_b, len_b, rank_b;
    CR_TRY(dmalloc(ctx, cols_b, (n_cols ? n_cols : 1) * sizeof(uint64_t)));
    CR_TRY(dmalloc(ctx, len_b, (n_cols + 1) * sizeof(uint32_t)));
    CR_TRY(dmalloc(ctx, rank_b, (n_cols ? n_cols : 1) * sizeof(uint32_t)));
    uint32_t *d_total = ctx->d_scalars + 16, total = 0;
    const long long *pa = (const long long *)a->d_indptr;
    {
        CrTimer t(ctx, CRGPU_T_MATRIX, n_cols);
        if (n_cols) {
            CR_HIP(ctx, hipMemcpyAsync(cols_b.p, cols, n_cols * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_selected_lengths, dim3(cr_grid(n_cols, 256)), dim3(256), 0, ctx->stream, pa, cols_b.as<uint64_t>(), n_cols,
                               a->d_barcode_rank, len_b.as<uint32_t>(), rank_b.as<uint32_t>());
        }
        CR_HIP(ctx, hipGetLastError());
        CR_TRY(cr_scan_small(ctx, len_b.as<uint32_t>(), n_cols, d_total));
    }
    CR_TRY(read_u32(ctx, d_total, &total));
    MatrixDevImpl *m = nullptr;
    CR_TRY(new_matrix_dev(ctx, n_cols, total, &m));
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        if (n_cols) {
            CR_HIP(ctx, hipMemcpyAsync(m->d_rank, rank_b.p, n_cols * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            hipLaunchKernelGGL(k_copy_columns, dim3(cr_grid(n_cols * 64, 256)), dim3(256), 0, ctx->stream, pa, cols_b.as<uint64_t>(), n_cols,
                               len_b.as<uint32_t>(), a->d_indices, a->d_data, m->d_indices, m->d_data);
        }
        hipLaunchKernelGGL(k_offsets_to_indptr, dim3(cr_grid(n_cols + 1, 256)), dim3(256), 0, ctx->stream, len_b.as<uint32_t>(), n_cols,
                           total, m->d_indptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            crgpu_matrix_dev_free(ctx, &m->view);
            return cr_fail(ctx, CRGPU_EHIP, "crgpu_select_barcodes_dev: kernel failed");
        }
    }
    *out = &m->view;
    return CRGPU_OK;
}

extern "C" void crgpu_matrix_dev_free(crgpu_ctx *ctx, crgpu_matrix_dev *mv) {
    if (!mv || !ctx) return;
    CR_ENTER(ctx);
    MatrixDevImpl *m = reinterpret_cast<MatrixDevImpl *>(mv);  // view is the first member
    cr_pool_free(ctx, m->d_rank);
    cr_pool_free(ctx, m->d_indptr);
    cr_pool_free(ctx, m->d_indices);
    cr_pool_free(ctx, m->d_data);
    delete m;
}

extern "C" int crgpu_matrix_dev_download(crgpu_ctx *ctx, const crgpu_matrix_dev *mv, uint32_t *rank_out, int64_t *indptr_out,
                                         int32_t *indices_out, int32_t *data_out) {
    if (!ctx || !mv) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (rank_out) CR_TRY(crgpu_memcpy_d2h(ctx, rank_out, mv->d_barcode_rank, mv->n_barcodes * sizeof(uint32_t)));
    if (indptr_out) CR_TRY(crgpu_memcpy_d2h(ctx, indptr_out, mv->d_indptr, (mv->n_barcodes + 1) * sizeof(int64_t)));
    if (indices_out) CR_TRY(crgpu_memcpy_d2h(ctx, indices_out, mv->d_indices, mv->nnz * sizeof(int32_t)));
    if (data_out) CR_TRY(crgpu_memcpy_d2h(ctx, data_out, mv->d_data, mv->nnz * sizeof(int32_t)));
    return CRGPU_OK;
}

extern "C" int crgpu_counts_info(crgpu_ctx *ctx, const crgpu_counts *c, uint64_t *n_triplets, uint64_t *n_molecules) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (n_triplets) *n_triplets = c->n_triplets;
    if (n_molecules) *n_molecules = c->n_molecules;
    return CRGPU_OK;
}

extern "C" int crgpu_counts_triplets_dev(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t **d_bc, uint32_t **d_feature,
                                         uint32_t **d_count) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (d_bc) *d_bc = c->d_bc;
    if (d_feature) *d_feature = c->d_feature;
    if (d_count) *d_count = c->d_count;
    return CRGPU_OK;
}

extern "C" int crgpu_counts_triplets(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t *bc_out, uint32_t *feature_out,
                                     uint32_t *count_out) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    const uint64_t b = c->n_triplets * sizeof(uint32_t);
    if (bc_out) CR_TRY(crgpu_memcpy_d2h(ctx, bc_out, c->d_bc, b));
    if (feature_out) CR_TRY(crgpu_memcpy_d2h(ctx, feature_out, c->d_feature, b));
    if (count_out) CR_TRY(crgpu_memcpy_d2h(ctx, count_out, c->d_count, b));
    return CRGPU_OK;
}

// the molecule table on the host in the order ALIGN_AND_COUNT emits it: order[o] = device position of the o-th UmiCount
static int molecule_order(crgpu_ctx *ctx, const crgpu_counts *c, std::vector<uint64_t> &keys, std::vector<uint32_t> &reads,
                          std::vector<uint32_t> &order) {
    const uint64_t nm = c->n_molecules;
    keys.resize(nm);
    reads.resize(nm);
    order.resize(nm);
    CR_TRY(crgpu_memcpy_d2h(ctx, keys.data(), c->d_mkeys, nm * sizeof(uint64_t)));
    CR_TRY(crgpu_memcpy_d2h(ctx, reads.data(), c->d_mreads, nm * sizeof(uint32_t)));
    const KeyLayout &L = c->layout;
    // align_and_count.rs:314 sorts a barcode's UmiCounts by (library_idx, feature_idx, umi, ...):
    // the device order is (barcode, feature, library, umi); reorder inside each barcode.
    for (uint64_t i = 0; i < nm; i++) order[i] = (uint32_t)i;
    auto fld = [&](uint64_t k, uint32_t sh, uint32_t bits) { return (uint32_t)((k >> sh) & (bits >= 64 ? ~0ull : ((1ull << bits) - 1))); };
    // with a single library and one UMI length the device order (barcode, feature, umi) already is the required one;
    // otherwise UmiCount's derived Ord: library_idx, feature_idx, umi (2-bit), read_count, utype (types.rs:152-160)
    if (L.bits_lib != 0 || L.bits_ulen != 0) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t ka = keys[a], kb = keys[b];
        const uint32_t bca = (uint32_t)(ka >> L.sh_bc()), bcb = (uint32_t)(kb >> L.sh_bc());
        if (bca != bcb) return bca < bcb;
        const uint32_t la = fld(ka, L.sh_libid(), L.bits_lib), lb = fld(kb, L.sh_libid(), L.bits_lib);
        if (la != lb) return la < lb;
        const uint32_t fa = fld(ka, L.sh_feat(), L.bits_feat), fb = fld(kb, L.sh_feat(), L.bits_feat);
        if (fa != fb) return fa < fb;
        const uint32_t ua = fld(ka, L.sh_umi(), L.bits_umi), ub = fld(kb, L.sh_umi(), L.bits_umi);
        if (ua != ub) return ua < ub;
        if (reads[a] != reads[b]) return reads[a] < reads[b];
        return (ka & 1ull) < (kb & 1ull);
    });
    return CRGPU_OK;
}

extern "C" int crgpu_counts_molecules(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t *bc_out, uint8_t *lib_out,
                                      uint32_t *feature_out, uint32_t *umi_out, uint32_t *read_count_out,
                                      uint8_t *utype_out) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    const uint64_t nm = c->n_molecules;
    if (!nm) return CRGPU_OK;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> reads, order;
    CR_TRY(molecule_order(ctx, c, keys, reads, order));
    const KeyLayout &L = c->layout;
    auto fld = [&](uint64_t k, uint32_t sh, uint32_t bits) { return (uint32_t)((k >> sh) & (bits >= 64 ? ~0ull : ((1ull << bits) - 1))); };
    std::vector<uint32_t> back;
    if (c->d_back && bc_out) {
        back.resize(c->n_back);
        CR_TRY(crgpu_memcpy_d2h(ctx, back.data(), c->d_back, (uint64_t)c->n_back * sizeof(uint32_t)));
    }
    for (uint64_t o = 0; o < nm; o++) {
        const uint64_t k = keys[order[o]];
        if (bc_out) bc_out[o] = back.empty() ? (uint32_t)(k >> L.sh_bc()) : back[(uint32_t)(k >> L.sh_bc())];
        if (lib_out) lib_out[o] = (uint8_t)fld(k, L.sh_libid(), L.bits_lib);
        if (feature_out) feature_out[o] = fld(k, L.sh_feat(), L.bits_feat);
        if (umi_out) umi_out[o] = fld(k, L.sh_umi(), L.bits_umi);
        if (read_count_out) read_count_out[o] = reads[order[o]];
        if (utype_out) utype_out[o] = (uint8_t)(k & 1ull);
    }
    return CRGPU_OK;
}

extern "C" int crgpu_counts_probe_idx(crgpu_ctx *ctx, const crgpu_counts *c, int32_t *probe_idx_out) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    const uint64_t nm = c->n_molecules;
    if (!nm) return CRGPU_OK;
    CR_REQUIRE(ctx, c->d_mprobe, CRGPU_ESTATE,
               "crgpu_counts_probe_idx: these counts were made without crgpu_records.d_probe_idx (crgpu_count_records_dev / crgpu_count_host)");
    CR_REQUIRE(ctx, probe_idx_out, CRGPU_EINVAL, "crgpu_counts_probe_idx: NULL output");
    std::vector<uint64_t> keys;
    std::vector<uint32_t> reads, order;
    CR_TRY(molecule_order(ctx, c, keys, reads, order));
    std::vector<int32_t> probe(nm);
    CR_TRY(crgpu_memcpy_d2h(ctx, probe.data(), c->d_mprobe, nm * sizeof(int32_t)));
    for (uint64_t o = 0; o < nm; o++) probe_idx_out[o] = probe[order[o]];
    return CRGPU_OK;
}

extern "C" int crgpu_enable_barcode_summary(crgpu_ctx *ctx, int on) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    ctx->barcode_summary_on = on != 0;
    return CRGPU_OK;
}

extern "C" int crgpu_counts_barcode_summary(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t rank_lo, uint32_t rank_hi,
                                            crgpu_barcode_summary_row *rows_out, uint64_t cap, uint64_t *n_rows_out) {
    if (!ctx || !c || !n_rows_out) return CRGPU_EINVAL;
    *n_rows_out = 0;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_counts_barcode_summary: no whitelist set");
    const uint32_t W = ctx->n_canon;
    CR_REQUIRE(ctx, c->n_canon == W, CRGPU_ESTATE, "crgpu_counts_barcode_summary: the whitelist changed since the counts were made");
    CR_REQUIRE(ctx, c->n_molecules == 0 || c->d_corr_reads, CRGPU_ESTATE,
               "crgpu_counts_barcode_summary: these counts carry no corrected-read table "
               "(crgpu_enable_barcode_summary before crgpu_count_keys_dev, or use crgpu_count_records_dev)");
    if (rank_hi > W) rank_hi = W;
    if (rank_lo >= rank_hi) return CRGPU_OK;
    const KeyLayout &L = c->layout;
    const uint32_t slots = 1u << L.bits_lib;
    const KL kl = make_kl(L);
    DevBuf umis_b, cand_b, rows_b;
    const size_t tab_bytes = (size_t)slots * W * sizeof(uint32_t);
    if (c->n_molecules) {
        CR_TRY(dmalloc(ctx, umis_b, tab_bytes));
        CR_TRY(dmalloc(ctx, cand_b, tab_bytes));
        CrTimer t(ctx, CRGPU_T_DEDUP);
        CR_HIP(ctx, hipMemsetAsync(umis_b.p, 0, tab_bytes, ctx->stream));
        CR_HIP(ctx, hipMemsetAsync(cand_b.p, 0, tab_bytes, ctx->stream));
        hipLaunchKernelGGL(k_molecule_sums, dim3(cr_grid((c->n_molecules + MS_ITEMS - 1) / MS_ITEMS, 256)), dim3(256), 0,
                           ctx->stream, kl, c->d_mkeys, c->d_mreads, c->n_molecules, W, umis_b.as<uint32_t>(),
                           cand_b.as<uint32_t>(), c->d_back);
        CR_HIP(ctx, hipGetLastError());
    }
    const uint64_t span = rank_hi - rank_lo;
    CR_TRY(dmalloc(ctx, rows_b, span * sizeof(crgpu_barcode_summary_row)));
    uint32_t *d_total = ctx->d_scalars + 16;
    uint64_t n_rows = 0;
    for (uint32_t lib = 0; lib < slots && lib < CRGPU_MAX_LIB; lib++) {
        if (!ctx->wl[lib].set) continue;
        const uint32_t *valid = ctx->wl[lib].d_valid, *corrected = ctx->wl[lib].d_corrected;
        const size_t off = (size_t)lib * W;
        uint32_t n = 0;
        {
            CrTimer t(ctx, CRGPU_T_DEDUP);
            CR_TRY(compact(ctx, SummaryFlag{valid, corrected, rank_lo},
                           EmitSummary{valid, corrected, c->n_molecules ? umis_b.as<uint32_t>() + off : nullptr,
                                       c->n_molecules ? cand_b.as<uint32_t>() + off : nullptr,
                                       c->d_corr_reads ? c->d_corr_reads + off : nullptr,
                                       c->d_filt_reads ? c->d_filt_reads + off : nullptr, rank_lo, lib,
                                       rows_b.as<crgpu_barcode_summary_row>()},
                           span, ctx->d_sort_hist, d_total));
        }
        CR_TRY(read_u32(ctx, d_total, &n));
        if (rows_out && n_rows + n <= cap && n)
            CR_TRY(crgpu_memcpy_d2h(ctx, rows_out + n_rows, rows_b.p, (size_t)n * sizeof(crgpu_barcode_summary_row)));
        n_rows += n;
    }
    *n_rows_out = n_rows;
    CR_REQUIRE(ctx, !rows_out || n_rows <= cap, CRGPU_ERANGE, "crgpu_counts_barcode_summary: %llu rows, room for %llu",
               (unsigned long long)n_rows, (unsigned long long)cap);
    return CRGPU_OK;
}

extern "C" void crgpu_counts_free(crgpu_ctx *ctx, crgpu_counts *c) {
    if (!c || !ctx) return;
    CR_ENTER(ctx);
    cr_pool_free(ctx, c->d_bc);
    cr_pool_free(ctx, c->d_feature);
    cr_pool_free(ctx, c->d_count);
    cr_pool_free(ctx, c->d_mkeys);
    cr_pool_free(ctx, c->d_mreads);
    cr_pool_free(ctx, c->d_corr_reads);
    cr_pool_free(ctx, c->d_filt_reads);
    cr_pool_free(ctx, c->d_mprobe);
    cr_pool_free(ctx, c->d_back);
    delete c;
}

// ------------------------------------------------------------------------------------------------
// aggr: MERGE_MOLECULES on the barcode_idx column
// ------------------------------------------------------------------------------------------------
// MoleculeInfoWriter::trim_barcodes (cr_h5/src/molecule_info.rs:890-960) keeps the barcodes of pass_filter and, unless
// pass_only, every barcode that a molecule refers to, in ascending order, and rewrites barcode_idx to positions in the
// trimmed list; MERGE_MOLECULES' join (cr_aggr/src/merge_molecules.rs:131-330) then concatenates the samples with
// barcode_idx shifted by the number of barcodes retained before (bc_idx_offsets).  The H5 container, the gem-group /
// library look-up tables (two tiny maps applied per row) and the metrics JSON stay with the host.
__global__ __launch_bounds__(256) void k_mark_u64(const uint64_t *__restrict__ idx, uint64_t n, uint64_t limit, uint8_t *__restrict__ flag,
                                                  uint32_t *__restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t v = idx[i];
        if (v < limit) flag[v] = 1; else *bad = 1u;
    }
}
struct EmitRetained {
    uint32_t *retained, *newpos;
    struct Pre {};
    __device__ __forceinline__ Pre pre(uint64_t) const { return Pre(); }
    __device__ __forceinline__ void operator()(uint64_t k, uint32_t o, Pre) const {
        retained[o] = (uint32_t)k;
        newpos[k] = o;
    }
};
__global__ __launch_bounds__(256) void k_remap_u64(uint64_t *__restrict__ idx, uint64_t n, const uint32_t *__restrict__ newpos,
                                                   uint64_t offset) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) idx[i] = offset + newpos[idx[i]];
}

extern "C" int crgpu_trim_molecule_barcodes_dev(crgpu_ctx *ctx, uint64_t *d_barcode_idx_inout, uint64_t n_molecules,
                                                uint64_t n_barcodes, uint64_t *pass_filter_idx_inout, uint64_t n_pass,
                                                int pass_only, uint64_t barcode_idx_offset, uint64_t *retained_out,
                                                uint64_t *n_retained_out) {
    if (!ctx || !n_retained_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *n_retained_out = 0;
    CR_REQUIRE(ctx, n_barcodes < 0xFFFFFFFFull, CRGPU_ERANGE, "crgpu_trim_molecule_barcodes: at most 2^32-2 barcodes");
    CR_REQUIRE(ctx, n_molecules == 0 || d_barcode_idx_inout, CRGPU_EINVAL, "crgpu_trim_molecule_barcodes: NULL barcode_idx");
    CR_REQUIRE(ctx, n_pass == 0 || pass_filter_idx_inout, CRGPU_EINVAL, "crgpu_trim_molecule_barcodes: NULL pass_filter");
    cr_invalidate(ctx);
    DevBuf flag_b, ret_b, pos_b, pf_b;
    CR_TRY(dmalloc(ctx, flag_b, n_barcodes + 1));
    CR_TRY(dmalloc(ctx, ret_b, (n_barcodes + 1) * sizeof(uint32_t)));
    CR_TRY(dmalloc(ctx, pos_b, (n_barcodes + 1) * sizeof(uint32_t)));
    uint8_t *flag = flag_b.as<uint8_t>();
    uint32_t *d_bad = ctx->d_scalars + 60, *d_total = ctx->d_scalars + 16, bad = 0, kept = 0;
    {
        CrTimer t(ctx, CRGPU_T_MATRIX, n_molecules);
        CR_HIP(ctx, hipMemsetAsync(flag, 0, n_barcodes + 1, ctx->stream));
        CR_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
        if (n_pass) {
            CR_TRY(dmalloc(ctx, pf_b, n_pass * sizeof(uint64_t)));
            CR_HIP(ctx, hipMemcpyAsync(pf_b.p, pass_filter_idx_inout, n_pass * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_mark_u64, dim3(cr_grid(n_pass, 256)), dim3(256), 0, ctx->stream, pf_b.as<uint64_t>(), n_pass, n_barcodes, flag, d_bad);
        }
        if (!pass_only && n_molecules)
            hipLaunchKernelGGL(k_mark_u64, dim3(cr_grid(n_molecules, 256)), dim3(256), 0, ctx->stream, d_barcode_idx_inout, n_molecules,
                               n_barcodes, flag, d_bad);
        CR_HIP(ctx, hipGetLastError());
        if (n_barcodes)
            CR_TRY(compact(ctx, CandFlag{flag}, EmitRetained{ret_b.as<uint32_t>(), pos_b.as<uint32_t>()}, n_barcodes, ctx->d_sort_hist, d_total));
    }
    CR_TRY(read_u32(ctx, d_bad, &bad));
    CR_REQUIRE(ctx, !bad, CRGPU_EINVAL, "crgpu_trim_molecule_barcodes: a barcode index lies beyond the %llu barcodes",
               (unsigned long long)n_barcodes);
    if (n_barcodes) CR_TRY(read_u32(ctx, d_total, &kept));
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        // with pass_only a molecule may refer to a barcode that is not retained: the reference panics there
        // ("Error accessing invalid barcode index"); here such rows would read an undefined position, so check first
        if (pass_only && n_molecules) {
            DevBuf chk_b;
            CR_TRY(dmalloc(ctx, chk_b, n_barcodes + 1));
            CR_HIP(ctx, hipMemsetAsync(chk_b.p, 0, n_barcodes + 1, ctx->stream));
            hipLaunchKernelGGL(k_mark_u64, dim3(cr_grid(n_molecules, 256)), dim3(256), 0, ctx->stream, d_barcode_idx_inout, n_molecules,
                               n_barcodes, chk_b.as<uint8_t>(), d_bad);
            std::vector<uint8_t> used(n_barcodes), keep(n_barcodes);
            CR_TRY(crgpu_memcpy_d2h(ctx, used.data(), chk_b.p, n_barcodes));
            CR_TRY(crgpu_memcpy_d2h(ctx, keep.data(), flag, n_barcodes));
            for (uint64_t b = 0; b < n_barcodes; b++)
                CR_REQUIRE(ctx, !used[b] || keep[b], CRGPU_EINVAL,
                           "crgpu_trim_molecule_barcodes: molecules refer to barcode %llu, which pass_filter does not retain",
                           (unsigned long long)b);
        }
        if (n_molecules)
            hipLaunchKernelGGL(k_remap_u64, dim3(cr_grid(n_molecules, 256)), dim3(256), 0, ctx->stream, d_barcode_idx_inout, n_molecules,
                               pos_b.as<uint32_t>(), barcode_idx_offset);
        if (n_pass) {
            hipLaunchKernelGGL(k_remap_u64, dim3(cr_grid(n_pass, 256)), dim3(256), 0, ctx->stream, pf_b.as<uint64_t>(), n_pass,
                               pos_b.as<uint32_t>(), barcode_idx_offset);
            CR_TRY(crgpu_memcpy_d2h(ctx, pass_filter_idx_inout, pf_b.p, n_pass * sizeof(uint64_t)));
        }
        CR_HIP(ctx, hipGetLastError());
    }
    if (retained_out && kept) {
        std::vector<uint32_t> r(kept);
        CR_TRY(crgpu_memcpy_d2h(ctx, r.data(), ret_b.p, (size_t)kept * sizeof(uint32_t)));
        for (uint32_t i = 0; i < kept; i++) retained_out[i] = r[i];
    }
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_retained_out = kept;
    return CRGPU_OK;
}
