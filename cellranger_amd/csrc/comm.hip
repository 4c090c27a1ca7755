// comm.hip -- the exchange steps between the GPUs that share one GEM well, behind the C ABI.
//
//   C1  all-reduce(sum) of a per-library histogram table       crgpu_allreduce_counts
//       (the corrector's prior must be global: make_shard.rs:343-358 -> barcode_correction.rs:295-325)
//   C2  all-to-all of molecule keys by barcode-rank range       crgpu_exchange_keys_dev
//       (the reference: barcode-sorted shards + make_chunks, align_and_count.rs:505-524)
//   C3  gather of the ranks' disjoint triplet / CSC blocks      crgpu_gatherv_dev, crgpu_gather_triplets_dev
//
// Two transports behind one small interface (CrComm):
//   * RCCL (librccl linked directly; xGMI between the GPUs of a node): one process per GPU, the ranks meet through the
//     128-byte ncclUniqueId that the host ships from rank 0 to the others.  All-to-all = grouped ncclSend/ncclRecv:
//     xGMI is point-to-point, every pair of GPUs has its own link, so the n-1 transfers of a rank run concurrently.
//   * local group: several contexts inside ONE process (a host thread per GPU, or -- on the one-GPU test box -- several
//     ranks on the same device).  The ranks meet through a shared object whose address travels in the id; collectives
//     are device-to-device copies (hipMemcpyAsync, peer access between different devices) between rendezvous barriers.
// The orchestration above the transports (ranges, partition, count exchange, offsets, ordering) is the same code.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"

int cr_partition_by_owner(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n, uint32_t sh_bc,
                          uint32_t n_ranks, const uint32_t *bounds, uint64_t *counts_out);

#define CR_LOCAL_MAGIC "CRGPU-LOCAL-GROUP"
#define CR_MAX_RANKS 256
#define CR_LOCAL_TIMEOUT_S 600

// ---- in-process group ------------------------------------------------------------------------------------------------
struct LocalGroup {
    uint32_t n = 0;
    std::mutex m;
    std::condition_variable cv;
    uint32_t arrived = 0, refs = 0, joined = 0;
    uint64_t generation = 0;
    bool broken = false;
    // what a rank publishes for the collective in flight
    const void *ptr[CR_MAX_RANKS] = {nullptr};
    int device[CR_MAX_RANKS] = {0};
    bool has_rank[CR_MAX_RANKS] = {false};
    std::vector<uint64_t> nums[CR_MAX_RANKS];
    double dbl[CR_MAX_RANKS] = {0};

    // true when every rank has arrived; false when the group broke (a rank failed or left) or the wait timed out
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == n) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        cv.wait_for(lk, std::chrono::seconds(CR_LOCAL_TIMEOUT_S), [&] { return generation != gen || broken; });
        if (generation != gen) return true;  // completed (a rank may have left right afterwards: the NEXT barrier fails)
        broken = true;                       // timed out, or a rank left while this one was waiting
        cv.notify_all();
        return false;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
};

struct LocalId {
    char magic[24];
    LocalGroup *group;
    uint32_t n_ranks;
};
static_assert(sizeof(LocalId) <= CRGPU_UNIQUE_ID_BYTES, "local id must fit the unique id");

struct CrComm {
    ncclComm_t nccl = nullptr;
    LocalGroup *local = nullptr;
};

static int nccl_fail(crgpu_ctx *ctx, ncclResult_t r, const char *what) {
    return cr_fail(ctx, CRGPU_ECOMM, "%s: %s", what, ncclGetErrorString(r));
}
#define CR_NCCL(ctx, call)                                  \
    do {                                                    \
        ncclResult_t _r = (call);                           \
        if (_r != ncclSuccess) return nccl_fail((ctx), _r, #call); \
    } while (0)

extern "C" int crgpu_get_unique_id(void *id_out) {
    if (!id_out) return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_get_unique_id: id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == CRGPU_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return cr_fail(nullptr, CRGPU_ECOMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof(id));
    return CRGPU_OK;
}

extern "C" int crgpu_local_group_id(uint32_t n_ranks, void *id_out) {
    if (!id_out || n_ranks < 1 || n_ranks > CR_MAX_RANKS)
        return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_local_group_id: 1..%d ranks", CR_MAX_RANKS);
    LocalGroup *g = new (std::nothrow) LocalGroup();
    if (!g) return cr_fail(nullptr, CRGPU_ENOMEM, "out of host memory");
    g->n = n_ranks;
    g->refs = n_ranks;  // released by the destroy of each of the n contexts (a group nobody joins is leaked: 10 KB)
    LocalId id;
    memset(&id, 0, sizeof(id));
    strncpy(id.magic, CR_LOCAL_MAGIC, sizeof(id.magic) - 1);
    id.group = g;
    id.n_ranks = n_ranks;
    memset(id_out, 0, CRGPU_UNIQUE_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return CRGPU_OK;
}

int cr_comm_init(crgpu_ctx *ctx, int n_ranks, int rank, const void *unique_id) {
    // the collectives keep per-rank words in arrays of CR_MAX_RANKS entries
    CR_REQUIRE(ctx, n_ranks >= 1 && n_ranks <= CR_MAX_RANKS, CRGPU_EINVAL, "crgpu_create: at most %d ranks per communicator (%d asked)",
               CR_MAX_RANKS, n_ranks);
    CrComm *c = new (std::nothrow) CrComm();
    if (!c) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    LocalId lid;
    memcpy(&lid, unique_id, sizeof(lid));
    if (memcmp(lid.magic, CR_LOCAL_MAGIC, sizeof(CR_LOCAL_MAGIC)) == 0) {
        if ((int)lid.n_ranks != n_ranks) {
            delete c;
            return cr_fail(ctx, CRGPU_EINVAL, "crgpu_create: the local group id was made for %u ranks, not %d", lid.n_ranks, n_ranks);
        }
        c->local = lid.group;
        ctx->comm = c;  // from here on crgpu_destroy releases this context's reference to the group
        bool dup = false;
        {
            std::lock_guard<std::mutex> lk(c->local->m);
            dup = c->local->has_rank[rank];
            if (!dup) {
                c->local->has_rank[rank] = true;
                c->local->device[rank] = ctx->device;
                c->local->joined++;
            }
        }
        if (dup) {  // two contexts with one rank: the group can never be complete
            c->local->abort();
            return cr_fail(ctx, CRGPU_EINVAL, "crgpu_create: rank %d joined the local group twice", rank);
        }
        if (!c->local->barrier()) return cr_fail(ctx, CRGPU_ECOMM, "crgpu_create: the other ranks of the local group did not arrive");
        // peer access for the device-to-device copies (and the table sums, which dereference the other ranks' pointers)
        // between different GPUs of the process: a peer that cannot be reached fails the create instead of a later kernel
        int peer_fail = -1;
        for (int r = 0; r < n_ranks && peer_fail < 0; r++) {
            const int d = c->local->device[r];
            if (d == ctx->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ctx->device, d) != hipSuccess || !can) {
                peer_fail = d;
                break;
            }
            hipError_t e = hipDeviceEnablePeerAccess(d, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_fail = d;
            (void)hipGetLastError();
        }
        // every rank learns whether all of them reach their peers (one more rendezvous with the verdicts)
        c->local->nums[rank].assign(1, peer_fail >= 0 ? 1u : 0u);
        if (!c->local->barrier()) return cr_fail(ctx, CRGPU_ECOMM, "crgpu_create: the local group broke during setup");
        bool any_fail = false;
        for (int r = 0; r < n_ranks; r++) any_fail |= c->local->nums[r].size() == 1 && c->local->nums[r][0] != 0;
        if (!c->local->barrier()) return cr_fail(ctx, CRGPU_ECOMM, "crgpu_create: the local group broke during setup");
        if (peer_fail >= 0)
            return cr_fail(ctx, CRGPU_ECOMM, "crgpu_create: device %d cannot access its peer device %d (no xGMI / PCIe peer path)", ctx->device,
                           peer_fail);
        if (any_fail) return cr_fail(ctx, CRGPU_ECOMM, "crgpu_create: another rank of the local group cannot reach its peer devices");
        return CRGPU_OK;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->nccl, n_ranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return cr_fail(ctx, CRGPU_ECOMM, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, ncclGetErrorString(r));
    }
    ctx->comm = c;
    return CRGPU_OK;
}

void cr_comm_destroy(crgpu_ctx *ctx) {
    CrComm *c = ctx->comm;
    if (!c) return;
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->local) {
        LocalGroup *g = c->local;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->broken = true;  // a rank that leaves ends the group: the others get CRGPU_ECOMM instead of a hang
            g->cv.notify_all();
            last = --g->refs == 0;
        }
        if (last) delete g;
    }
    delete c;
    ctx->comm = nullptr;
}

// ---- transport primitives ----------------------------------------------------------------------------------------------
// every rank contributes k u64 words; all_out[r * k + j] = word j of rank r (host arrays)
static int comm_allgather_u64(crgpu_ctx *ctx, const uint64_t *mine, uint32_t k, uint64_t *all_out) {
    const int W = ctx->n_ranks;
    CrComm *c = ctx->comm;
    if (!c || W == 1) {
        memcpy(all_out, mine, k * sizeof(uint64_t));
        return CRGPU_OK;
    }
    if (c->local) {
        LocalGroup *g = c->local;
        g->nums[ctx->rank].assign(mine, mine + k);
        if (!g->barrier()) return cr_fail(ctx, CRGPU_ECOMM, "local group: a rank went away");
        for (int r = 0; r < W; r++) memcpy(all_out + (size_t)r * k, g->nums[r].data(), k * sizeof(uint64_t));
        if (!g->barrier()) return cr_fail(ctx, CRGPU_ECOMM, "local group: a rank went away");  // everybody has read
        return CRGPU_OK;
    }
    uint64_t *d = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d, (size_t)(W + 1) * k * sizeof(uint64_t)));
    hipError_t e = hipMemcpyAsync(d, mine, k * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = ncclAllGather(d, d + k, k, ncclUint64, c->nccl, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess)
        e = hipMemcpyAsync(all_out, d + k, (size_t)W * k * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
    cr_pool_free(ctx, d);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclAllGather");
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "allgather: %s", hipGetErrorString(e));
    return CRGPU_OK;
}

__global__ __launch_bounds__(256) void k_sum_tables(uint32_t *__restrict__ dst, const uint32_t *const *__restrict__ src,
                                                    uint32_t n_src, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t s = 0;
        for (uint32_t r = 0; r < n_src; r++) s += src[r][i];
        dst[i] = s;
    }
}

// element-wise sum over the ranks of a u32 device array, in place
static int comm_allreduce_u32(crgpu_ctx *ctx, uint32_t *d_buf, uint64_t n) {
    const int W = ctx->n_ranks;
    CrComm *c = ctx->comm;
    if (!c || W == 1 || n == 0) return CRGPU_OK;
    if (c->nccl) {
        CR_NCCL(ctx, ncclAllReduce(d_buf, d_buf, n, ncclUint32, ncclSum, c->nccl, ctx->stream));
        return CRGPU_OK;
    }
    LocalGroup *g = c->local;
    // everybody sums every rank's table into a private buffer, then replaces its own table
    uint32_t *d_sum = nullptr;
    const uint32_t **d_ptrs = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_sum, n * sizeof(uint32_t)));
    int rc = cr_pool_alloc(ctx, (void **)&d_ptrs, W * sizeof(void *));
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d_sum);
        return rc;
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);  // my table is final
    g->ptr[ctx->rank] = d_buf;
    bool ok = g->barrier();
    if (ok && e == hipSuccess) {
        const void *ptrs[CR_MAX_RANKS];
        for (int r = 0; r < W; r++) ptrs[r] = g->ptr[r];
        e = hipMemcpyAsync(d_ptrs, ptrs, W * sizeof(void *), hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(k_sum_tables, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_sum, d_ptrs, (uint32_t)W, n);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    ok = g->barrier() && ok;  // everybody has read every table
    if (ok && e == hipSuccess) e = hipMemcpyAsync(d_buf, d_sum, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
    cr_pool_free(ctx, d_sum);
    cr_pool_free(ctx, d_ptrs);
    if (!ok) return cr_fail(ctx, CRGPU_ECOMM, "local group: a rank went away");
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "allreduce: %s", hipGetErrorString(e));
    return CRGPU_OK;
}

// rank r sends send_bytes[p] bytes at send_off[p] of d_send to every p and receives recv_bytes[p] at recv_off[p] of
// d_recv from every p (host arrays of n_ranks entries; recv_bytes[p] must equal what p sends to r)
static int comm_alltoallv(crgpu_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                          void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes) {
    const int W = ctx->n_ranks;
    CrComm *c = ctx->comm;
    if (!c || W == 1) {
        if (send_bytes[0])
            CR_HIP(ctx, hipMemcpyAsync((char *)d_recv + recv_off[0], (const char *)d_send + send_off[0], send_bytes[0],
                                       hipMemcpyDeviceToDevice, ctx->stream));
        return CRGPU_OK;
    }
    if (c->nccl) {
        CR_NCCL(ctx, ncclGroupStart());
        ncclResult_t r = ncclSuccess;
        for (int p = 0; p < W && r == ncclSuccess; p++) {
            if (send_bytes[p]) r = ncclSend((const char *)d_send + send_off[p], send_bytes[p], ncclInt8, p, c->nccl, ctx->stream);
            if (r == ncclSuccess && recv_bytes[p])
                r = ncclRecv((char *)d_recv + recv_off[p], recv_bytes[p], ncclInt8, p, c->nccl, ctx->stream);
        }
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclSend/ncclRecv");
        if (r2 != ncclSuccess) return nccl_fail(ctx, r2, "ncclGroupEnd");
        return CRGPU_OK;
    }
    LocalGroup *g = c->local;
    hipError_t e = hipStreamSynchronize(ctx->stream);  // my send buffer is final
    g->ptr[ctx->rank] = d_send;
    g->nums[ctx->rank].assign(send_off, send_off + W);
    bool ok = g->barrier();
    if (ok && e == hipSuccess) {
        for (int p = 0; p < W && e == hipSuccess; p++) {  // pull my part of every rank's send buffer
            if (!recv_bytes[p]) continue;
            const char *src = (const char *)g->ptr[p] + g->nums[p][ctx->rank];
            e = hipMemcpyAsync((char *)d_recv + recv_off[p], src, recv_bytes[p], hipMemcpyDefault, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    ok = g->barrier() && ok;  // the senders may reuse their buffers
    if (!ok) return cr_fail(ctx, CRGPU_ECOMM, "local group: a rank went away");
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "alltoallv: %s", hipGetErrorString(e));
    return CRGPU_OK;
}

// ---- agreement on failure ----------------------------------------------------------------------------------------------------
// RCCL has no timeout: a rank that returns with a local error (out of memory, a launch failure, a limit) between two
// collectives leaves its peers waiting in the next one for ever.  So every local step in front of a data exchange only
// RECORDS its status; the status travels with the count exchange (or a one-word all-gather of its own), and all ranks leave
// together: the failing rank with its own error, the others with CRGPU_ECOMM naming it.
int cr_comm_agree(crgpu_ctx *ctx, int local_rc, const char *where) {
    const int W = ctx->n_ranks;
    if (W == 1 || !ctx->comm) return local_rc;
    const std::string own = ctx->err;  // the allgather below may overwrite the message
    uint64_t mine = (uint64_t)(uint32_t)local_rc;
    std::vector<uint64_t> all(W);
    const int rc = comm_allgather_u64(ctx, &mine, 1, all.data());
    if (rc != CRGPU_OK) return rc;  // the transport itself failed: every rank sees that
    if (local_rc != CRGPU_OK) {
        ctx->err = own;
        return local_rc;
    }
    for (int r = 0; r < W; r++)
        if ((int32_t)(uint32_t)all[r] != CRGPU_OK)
            return cr_fail(ctx, CRGPU_ECOMM, "%s: rank %d failed (error %d); the collective was abandoned on every rank", where, r,
                           (int)(int32_t)(uint32_t)all[r]);
    return CRGPU_OK;
}
// The W x W matrix of send counts (all[p * W + r] = what p sends to r) together with every rank's status.  On return all
// ranks agree: CRGPU_OK, the local error of a failing rank (CRGPU_ECOMM on the others), or CRGPU_ERANGE on every rank when
// some rank would receive more than max_recv elements.
int cr_comm_exchange_counts(crgpu_ctx *ctx, int local_rc, const uint64_t *send_cnt, uint64_t *all, uint64_t max_recv, const char *where) {
    const int W = ctx->n_ranks;
    const std::string own = ctx->err;
    std::vector<uint64_t> mine(W + 1), got((size_t)W * (W + 1));
    for (int p = 0; p < W; p++) mine[p] = local_rc == CRGPU_OK ? send_cnt[p] : 0;
    mine[W] = (uint64_t)(uint32_t)local_rc;
    const int rc = comm_allgather_u64(ctx, mine.data(), (uint32_t)(W + 1), got.data());
    if (rc != CRGPU_OK) return rc;
    if (local_rc != CRGPU_OK) {
        ctx->err = own;
        return local_rc;
    }
    for (int r = 0; r < W; r++) {
        const int32_t st = (int32_t)(uint32_t)got[(size_t)r * (W + 1) + W];
        if (st != CRGPU_OK)
            return cr_fail(ctx, CRGPU_ECOMM, "%s: rank %d failed (error %d); the collective was abandoned on every rank", where, r, (int)st);
        for (int p = 0; p < W; p++) all[(size_t)r * W + p] = got[(size_t)r * (W + 1) + p];
    }
    for (int r = 0; r < W; r++) {  // every rank holds the whole matrix: the limit is checked for all of them, by all of them
        uint64_t n_recv = 0;
        for (int p = 0; p < W; p++) n_recv += all[(size_t)p * W + r];
        if (n_recv > max_recv)
            return cr_fail(ctx, CRGPU_ERANGE, "%s: rank %d would own %llu keys (> %llu per call); split the well over more ranks", where, r,
                           (unsigned long long)n_recv, (unsigned long long)max_recv);
    }
    return CRGPU_OK;
}
// test hook: CRGPU_TEST_FAIL_EXCHANGE_RANK=<r> makes rank r's preparation of the key exchange fail
int cr_comm_test_failure(crgpu_ctx *ctx) {
    const char *e = getenv("CRGPU_TEST_FAIL_EXCHANGE_RANK");
    if (e && ctx->n_ranks > 1 && atoi(e) == ctx->rank)
        return cr_fail(ctx, CRGPU_ENOMEM, "forced failure of rank %d in front of the key exchange (CRGPU_TEST_FAIL_EXCHANGE_RANK)", ctx->rank);
    return CRGPU_OK;
}

int cr_comm_allgather_u64(crgpu_ctx *ctx, const uint64_t *mine, uint32_t k, uint64_t *all_out) {
    return comm_allgather_u64(ctx, mine, k, all_out);
}
int cr_comm_alltoallv(crgpu_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                      const uint64_t *recv_off, const uint64_t *recv_bytes) {
    return comm_alltoallv(ctx, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes);
}

// ---- entry points ----------------------------------------------------------------------------------------------------------
extern "C" int crgpu_barrier(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t one = 1, all[CR_MAX_RANKS];
    return comm_allgather_u64(ctx, &one, 1, all);
}

extern "C" int crgpu_allreduce_max_f64(crgpu_ctx *ctx, double *value_inout) {
    if (!ctx || !value_inout) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    uint64_t mine, all[CR_MAX_RANKS];
    memcpy(&mine, value_inout, sizeof(mine));
    CR_TRY(comm_allgather_u64(ctx, &mine, 1, all));
    double mx = *value_inout;
    for (int r = 0; r < ctx->n_ranks; r++) {
        double v;
        memcpy(&v, &all[r], sizeof(v));
        if (v > mx) mx = v;
    }
    *value_inout = mx;
    return CRGPU_OK;
}

// sum over the ranks of a small host array (MAKE_SHARD's feature counts, read totals): one all-gather of n words
extern "C" int crgpu_allreduce_sum_i64(crgpu_ctx *ctx, int64_t *values_inout, uint32_t n) {
    if (!ctx || (n && !values_inout)) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, n <= (1u << 20), CRGPU_ERANGE, "crgpu_allreduce_sum_i64: at most 2^20 values (a host-side helper, not a data path)");
    if (ctx->n_ranks == 1 || n == 0) return CRGPU_OK;
    const int W = ctx->n_ranks;
    std::vector<uint64_t> mine(n), all((size_t)W * n);
    memcpy(mine.data(), values_inout, n * sizeof(uint64_t));
    CR_TRY(comm_allgather_u64(ctx, mine.data(), n, all.data()));
    for (uint32_t j = 0; j < n; j++) {
        int64_t s = 0;
        for (int r = 0; r < W; r++) s += (int64_t)all[(size_t)r * n + j];
        values_inout[j] = s;
    }
    return CRGPU_OK;
}

extern "C" int crgpu_allreduce_counts(crgpu_ctx *ctx, int lib, int which) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_allreduce_counts: no whitelist set");
    CR_REQUIRE(ctx, which == CRGPU_COUNTS_VALID || which == CRGPU_COUNTS_CORRECTED, CRGPU_EINVAL,
               "crgpu_allreduce_counts: which must be CRGPU_COUNTS_VALID or CRGPU_COUNTS_CORRECTED");
    CR_REQUIRE(ctx, lib < CRGPU_MAX_LIB, CRGPU_EINVAL, "crgpu_allreduce_counts: library %d out of range", lib);
    CR_REQUIRE(ctx, lib < 0 || ctx->wl[lib].set, CRGPU_ESTATE, "crgpu_allreduce_counts: library %d has no whitelist", lib);
    if (ctx->n_ranks == 1) return CRGPU_OK;
    cr_dense_drop(ctx);
    CrTimer t(ctx, CRGPU_T_COMM, ctx->n_canon);
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        if ((lib >= 0 && l != lib) || !ctx->wl[l].set) continue;
        uint32_t *tab = which == CRGPU_COUNTS_VALID ? ctx->wl[l].d_valid : ctx->wl[l].d_corrected;
        CR_TRY(comm_allreduce_u32(ctx, tab, ctx->n_canon));
        ctx->comm_bytes[0] += (uint64_t)ctx->n_canon * sizeof(uint32_t);
    }
    return CRGPU_OK;
}

extern "C" int crgpu_exchange_keys_dev(crgpu_ctx *ctx, const uint64_t *d_keys, uint64_t n_keys, uint64_t **d_recv_out,
                                       uint64_t *n_recv_out, uint32_t *bounds_out) {
    if (!ctx || !d_recv_out || !n_recv_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *d_recv_out = nullptr;
    *n_recv_out = 0;
    CR_REQUIRE(ctx, ctx->layout.set, CRGPU_ESTATE, "crgpu_exchange_keys: call crgpu_set_key_layout first");
    CR_REQUIRE(ctx, n_keys == 0 || d_keys, CRGPU_EINVAL, "crgpu_exchange_keys: NULL keys");
    cr_invalidate(ctx);
    CR_TRY(cr_dense_ensure(ctx));
    const int W = ctx->n_ranks;
    // local preparation: nothing here returns early (cr_comm_exchange_counts carries the status to every rank)
    std::vector<uint32_t> bounds(W + 1);
    int rc = crgpu_balanced_bounds(ctx, (uint32_t)W, bounds.data());
    if (rc == CRGPU_OK && bounds_out) memcpy(bounds_out, bounds.data(), (W + 1) * sizeof(uint32_t));
    // stable partition of my keys by owner
    uint64_t *d_part = nullptr;
    if (rc == CRGPU_OK) rc = cr_pool_alloc(ctx, (void **)&d_part, (n_keys ? n_keys : 1) * sizeof(uint64_t));
    std::vector<uint64_t> send_cnt(W, 0), all(W * (size_t)W, 0);
    if (rc == CRGPU_OK) rc = cr_comm_test_failure(ctx);
    if (rc == CRGPU_OK)
        rc = cr_partition_by_owner(ctx, d_keys, d_part, n_keys, ctx->layout.sh_bc(), (uint32_t)W, bounds.data(), send_cnt.data());
    CrTimer t(ctx, CRGPU_T_COMM, n_keys);
    rc = cr_comm_exchange_counts(ctx, rc, send_cnt.data(), all.data(), 0x7FFFFFFFull, "crgpu_exchange_keys");
    uint64_t *d_recv = nullptr;
    uint64_t n_recv = 0;
    std::vector<uint64_t> soff(W), sbytes(W), roff(W), rbytes(W);
    if (rc == CRGPU_OK) {
        uint64_t so = 0;
        for (int p = 0; p < W; p++) {
            soff[p] = so * sizeof(uint64_t);
            sbytes[p] = send_cnt[p] * sizeof(uint64_t);
            so += send_cnt[p];
            const uint64_t from_p = all[(size_t)p * W + ctx->rank];
            roff[p] = n_recv * sizeof(uint64_t);
            rbytes[p] = from_p * sizeof(uint64_t);
            n_recv += from_p;
        }
        // the receive buffer is a local allocation: agree on it before anybody posts a transfer
        rc = cr_comm_agree(ctx, cr_pool_alloc(ctx, (void **)&d_recv, (n_recv ? n_recv : 1) * sizeof(uint64_t)), "crgpu_exchange_keys");
    }
    if (rc == CRGPU_OK) rc = comm_alltoallv(ctx, d_part, soff.data(), sbytes.data(), d_recv, roff.data(), rbytes.data());
    if (rc == CRGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_exchange_keys: sync failed");
    cr_pool_free(ctx, d_part);
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d_recv);
        return rc;
    }
    ctx->comm_bytes[1] += (uint64_t)n_keys * sizeof(uint64_t);
    *d_recv_out = d_recv;
    *n_recv_out = n_recv;
    return CRGPU_OK;
}

extern "C" int crgpu_gatherv_dev(crgpu_ctx *ctx, const void *d_src, uint64_t bytes, int root, void **d_out, uint64_t *bytes_out) {
    if (!ctx || !d_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *d_out = nullptr;
    const int W = ctx->n_ranks;
    CR_REQUIRE(ctx, root >= 0 && root < W, CRGPU_EINVAL, "crgpu_gatherv: root %d of %d ranks", root, W);
    CR_REQUIRE(ctx, bytes == 0 || d_src, CRGPU_EINVAL, "crgpu_gatherv: NULL source");
    cr_invalidate(ctx);
    CrTimer t(ctx, CRGPU_T_COMM, bytes);
    std::vector<uint64_t> all(W);
    CR_TRY(comm_allgather_u64(ctx, &bytes, 1, all.data()));
    std::vector<uint64_t> soff(W, 0), sbytes(W, 0), roff(W, 0), rbytes(W, 0);
    sbytes[root] = bytes;
    uint64_t total = 0;
    if (ctx->rank == root)
        for (int p = 0; p < W; p++) {
            roff[p] = total;
            rbytes[p] = all[p];
            total += all[p];
            if (bytes_out) bytes_out[p] = all[p];
        }
    void *d = nullptr;
    if (ctx->rank == root) CR_TRY(cr_pool_alloc(ctx, &d, total ? total : 1));
    int rc = comm_alltoallv(ctx, d_src, soff.data(), sbytes.data(), d, roff.data(), rbytes.data());
    if (rc == CRGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_gatherv: sync failed");
    if (rc != CRGPU_OK) {
        cr_pool_free(ctx, d);
        return rc;
    }
    *d_out = d;
    return CRGPU_OK;
}

// C3 in one exchange: ONE all-gather of the triplet counts, then the three arrays of every rank travel to root inside one
// grouped send / receive (3 (n - 1) point-to-point transfers in flight on root's links at once) and one stream
// synchronisation -- instead of three gathers with an all-gather and a synchronisation each.
extern "C" int crgpu_gather_triplets_dev(crgpu_ctx *ctx, const crgpu_counts *c, int root, uint32_t **d_bc_out,
                                         uint32_t **d_feature_out, uint32_t **d_count_out, uint64_t *n_total_out) {
    if (!ctx || !c || !d_bc_out || !d_feature_out || !d_count_out || !n_total_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *d_bc_out = *d_feature_out = *d_count_out = nullptr;
    *n_total_out = 0;
    const int W = ctx->n_ranks;
    CR_REQUIRE(ctx, root >= 0 && root < W, CRGPU_EINVAL, "crgpu_gather_triplets: root %d of %d ranks", root, W);
    uint64_t nt = 0;
    uint32_t *src[3] = {nullptr, nullptr, nullptr};
    int rc = crgpu_counts_info(ctx, c, &nt, nullptr);
    if (rc == CRGPU_OK) rc = crgpu_counts_triplets_dev(ctx, c, &src[0], &src[1], &src[2]);
    cr_invalidate(ctx);
    CrTimer t(ctx, CRGPU_T_COMM, nt);
    std::vector<uint64_t> all(W, 0);
    {   // counts + status of every rank
        std::vector<uint64_t> mine = {nt, (uint64_t)(uint32_t)rc}, got((size_t)W * 2);
        const std::string own = ctx->err;
        const int rc2 = comm_allgather_u64(ctx, mine.data(), 2, got.data());
        if (rc2 != CRGPU_OK) return rc2;
        if (rc != CRGPU_OK) {
            ctx->err = own;
            return rc;
        }
        for (int r = 0; r < W; r++) {
            if ((int32_t)(uint32_t)got[2 * r + 1] != CRGPU_OK)
                return cr_fail(ctx, CRGPU_ECOMM, "crgpu_gather_triplets: rank %d failed (error %d); abandoned on every rank", r,
                               (int)(int32_t)(uint32_t)got[2 * r + 1]);
            all[r] = got[2 * r];
        }
    }
    uint64_t total = 0;
    for (int p = 0; p < W; p++) total += all[p];
    void *out[3] = {nullptr, nullptr, nullptr};
    rc = CRGPU_OK;
    if (ctx->rank == root)
        for (int a = 0; a < 3 && rc == CRGPU_OK; a++) rc = cr_pool_alloc(ctx, &out[a], (total ? total : 1) * sizeof(uint32_t));
    rc = cr_comm_agree(ctx, rc, "crgpu_gather_triplets");
    CrComm *cm = ctx->comm;
    if (rc == CRGPU_OK) {
        if (!cm || W == 1) {
            for (int a = 0; a < 3 && nt; a++)
                if (hipMemcpyAsync(out[a], src[a], nt * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                    rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_gather_triplets: copy failed");
        } else if (cm->nccl) {
            ncclResult_t r = ncclGroupStart();
            for (int a = 0; a < 3 && r == ncclSuccess; a++) {
                if (nt) r = ncclSend(src[a], nt * sizeof(uint32_t), ncclInt8, root, cm->nccl, ctx->stream);
                if (ctx->rank == root) {
                    uint64_t off = 0;
                    for (int p = 0; p < W && r == ncclSuccess; p++) {
                        if (all[p]) r = ncclRecv((char *)out[a] + off * sizeof(uint32_t), all[p] * sizeof(uint32_t), ncclInt8, p, cm->nccl, ctx->stream);
                        off += all[p];
                    }
                }
            }
            const ncclResult_t r2 = ncclGroupEnd();
            if (r != ncclSuccess) rc = nccl_fail(ctx, r, "ncclSend/ncclRecv (triplets)");
            else if (r2 != ncclSuccess) rc = nccl_fail(ctx, r2, "ncclGroupEnd (triplets)");
        } else {
            LocalGroup *g = cm->local;
            hipError_t e = hipStreamSynchronize(ctx->stream);  // my triplets are final
            g->nums[ctx->rank].assign({(uint64_t)(uintptr_t)src[0], (uint64_t)(uintptr_t)src[1], (uint64_t)(uintptr_t)src[2]});
            bool ok = g->barrier();
            if (ok && e == hipSuccess && ctx->rank == root) {
                uint64_t off = 0;
                for (int p = 0; p < W && e == hipSuccess; p++) {
                    for (int a = 0; a < 3 && e == hipSuccess && all[p]; a++)
                        e = hipMemcpyAsync((char *)out[a] + off * sizeof(uint32_t), (const void *)(uintptr_t)g->nums[p][a],
                                           all[p] * sizeof(uint32_t), hipMemcpyDefault, ctx->stream);
                    off += all[p];
                }
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            }
            ok = g->barrier() && ok;  // the senders may free their triplets
            if (!ok) rc = cr_fail(ctx, CRGPU_ECOMM, "local group: a rank went away");
            else if (e != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_gather_triplets: %s", hipGetErrorString(e));
        }
    }
    if (rc == CRGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_gather_triplets: sync failed");
    if (rc != CRGPU_OK) {
        for (int a = 0; a < 3; a++) cr_pool_free(ctx, out[a]);
        return rc;
    }
    ctx->comm_bytes[2] += 3ull * nt * sizeof(uint32_t);
    if (ctx->rank == root) *n_total_out = total;
    *d_bc_out = (uint32_t *)out[0];
    *d_feature_out = (uint32_t *)out[1];
    *d_count_out = (uint32_t *)out[2];
    return CRGPU_OK;
}
