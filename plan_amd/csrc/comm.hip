// Multi-GPU exchange behind the C ABI: one process per GPU, RCCL over xGMI.
//
// No reference counterpart — the reference drives a query from one goroutine on one CPU
// (SURVEY.md §2, §8e). What a Go host needs to run the partitioned join queries is here as plain
// C entry points: communicator set-up from a 128-byte id the ranks share over any host channel,
// the count exchange, the all-to-all of column buffers as ONE group of ncclSend/ncclRecv pairs on
// the ctx stream (no host synchronisation between a partition's gathers and the exchange), a
// variable-length all-gather (broadcast of small build sides) and small reductions for merges.
//
// xGMI is point to point (7 links per GPU): a balanced all-to-all drives all links at once, so all
// columns of a stage go into one group call and RCCL schedules every peer pair concurrently.
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>

#include "common.h"

// ---- the in-process transport (ph_comm_init_local): the ranks are THREADS of one process, each with a ctx of its own — on one device (the
// one-GPU test box: RCCL refuses two ranks on one device) or on several (a host that drives all GPUs of a node from one process).
// A collective is: finish the own stream, post the send pointers, meet at a host barrier, copy every peer's rows out of ITS buffer with
// device-to-device copies on the own stream, finish, meet again (nobody reuses a send buffer before every peer has read it). Host-blocking and
// plain — it is the transport of the correctness tests, the collectives' SEMANTICS are the RCCL path's.
struct ph_local_group {
    int nranks = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<std::vector<const void *>> ptrs;   // per rank: the pointers it posted for the running collective
    std::vector<std::vector<int64_t>> vals;       // per rank: the host values it posted
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++arrived == nranks) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct ph_comm {
    ph_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    ph_local_group *local = nullptr;   // != nullptr: the in-process transport instead of RCCL
    int nranks = 1, rank = 0;
    // asynchronous collectives (ph_comm_allgather with async != 0) run on their own stream so that
    // the next kernels on the ctx stream overlap them
    hipStream_t cstream = nullptr;
    hipEvent_t ready = nullptr;
    static constexpr int RING = 4;
    hipEvent_t done[RING] = {};     // done[i % RING] = end of the i-th asynchronous collective
    int64_t issued = 0, waited = 0; // collectives issued / already waited for by the ctx stream
    int64_t *dev_words = nullptr;   // nranks*nranks + 64 words of device scratch
};

#define PH_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            ph::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return PH_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

extern "C" int ph_comm_unique_id(void *id_out) {
    PH_REQUIRE(id_out != nullptr, "ph_comm_unique_id: id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == PH_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    PH_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PH_OK;
}

extern "C" int ph_comm_init(ph_ctx *ctx, int32_t nranks, int32_t rank, const void *id, ph_comm **out) {
    PH_REQUIRE(ctx && id && out && nranks >= 1 && rank >= 0 && rank < nranks, "ph_comm_init: bad arguments");
    PH_HIP(hipSetDevice(ctx->device));
    ph_comm *c = new ph_comm();
    c->ctx = ctx;
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        ph::set_error("ncclCommInitRank(%d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
        delete c;
        return PH_EHIP;
    }
    if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[3], hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->dev_words, ((size_t)nranks * nranks + 2 * (size_t)nranks + 64) * 8) != hipSuccess) {
        ph::set_error("ph_comm_init: stream/event/scratch creation failed");
        ph_comm_destroy(c);
        return PH_EHIP;
    }
    *out = c;
    return PH_OK;
}

extern "C" int ph_local_group_create(int32_t nranks, ph_local_group **out) {
    PH_REQUIRE(out && nranks >= 1 && nranks <= 64, "ph_local_group_create: 1..64 ranks");
    ph_local_group *g = new ph_local_group();
    g->nranks = nranks;
    g->ptrs.resize((size_t)nranks);
    g->vals.resize((size_t)nranks);
    *out = g;
    return PH_OK;
}

extern "C" void ph_local_group_free(ph_local_group *g) { delete g; }

extern "C" int ph_comm_init_local(ph_ctx *ctx, ph_local_group *g, int32_t rank, ph_comm **out) {
    PH_REQUIRE(ctx && g && out && rank >= 0 && rank < g->nranks, "ph_comm_init_local: bad arguments");
    PH_HIP(hipSetDevice(ctx->device));
    ph_comm *c = new ph_comm();
    c->ctx = ctx;
    c->local = g;
    c->nranks = g->nranks;
    c->rank = rank;
    if (hipMalloc((void **)&c->dev_words, ((size_t)c->nranks * c->nranks + 2 * (size_t)c->nranks + 64) * 8) != hipSuccess) {
        ph::set_error("ph_comm_init_local: scratch allocation failed");
        delete c;
        return PH_EHIP;
    }
    *out = c;
    return PH_OK;
}

// ---- local transport primitives. `parts` describes what this rank READS: for every peer r, (offset into r's posted buffer k, offset into the own
// receive buffer k, bytes). Every rank posts `nbuf` send pointers.
namespace {
struct LocalCopy { int peer, buf; int64_t src_off, dst_off, bytes; };
int local_collective(ph_comm *c, const std::vector<const void *> &send, const std::vector<void *> &recv, const std::vector<LocalCopy> &copies) {
    ph_local_group *g = c->local;
    hipStream_t st = c->ctx->stream;
    PH_HIP(hipSetDevice(c->ctx->device));
    int rc = hipStreamSynchronize(st) == hipSuccess ? PH_OK : PH_EHIP;   // the send buffers are complete
    { std::lock_guard<std::mutex> lk(g->mu); g->ptrs[(size_t)c->rank] = send; }
    g->barrier();
    if (rc == PH_OK)
        for (const LocalCopy &cp : copies) {
            if (cp.bytes <= 0) continue;
            const void *src = nullptr;
            { std::lock_guard<std::mutex> lk(g->mu); src = g->ptrs[(size_t)cp.peer][(size_t)cp.buf]; }
            if (hipMemcpyAsync((char *)recv[(size_t)cp.buf] + cp.dst_off, (const char *)src + cp.src_off, (size_t)cp.bytes, hipMemcpyDefault, st) != hipSuccess) { rc = PH_EHIP; break; }
        }
    if (rc == PH_OK && hipStreamSynchronize(st) != hipSuccess) rc = PH_EHIP;
    g->barrier();   // every peer has read this rank's buffers
    if (rc != PH_OK) ph::set_error("local transport: a device copy failed");
    return rc;
}
// host values of every rank, rank-major (n per rank)
int local_allgather_host(ph_comm *c, const int64_t *mine, int n, std::vector<int64_t> *all) {
    ph_local_group *g = c->local;
    { std::lock_guard<std::mutex> lk(g->mu); g->vals[(size_t)c->rank].assign(mine, mine + n); }
    g->barrier();
    all->clear();
    { std::lock_guard<std::mutex> lk(g->mu); for (int r = 0; r < c->nranks; r++) all->insert(all->end(), g->vals[(size_t)r].begin(), g->vals[(size_t)r].end()); }
    g->barrier();
    return PH_OK;
}
}  // namespace

extern "C" void ph_comm_destroy(ph_comm *c) {
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    if (c->ctx) (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    if (c->dev_words) (void)hipFree(c->dev_words);
    delete c;
}

extern "C" int32_t ph_comm_nranks(const ph_comm *c) { return c ? c->nranks : 0; }
extern "C" int32_t ph_comm_rank(const ph_comm *c) { return c ? c->rank : -1; }

extern "C" int ph_comm_wait_keep(ph_comm *c, int32_t keep) {
    PH_REQUIRE(c != nullptr && keep >= 0 && keep < ph_comm::RING, "ph_comm_wait_keep: bad arguments (0 <= keep < %d)", ph_comm::RING);
    if (c->local) return PH_OK;   // the local transport's collectives are complete when they return
    // collectives complete in issue order on the side stream: waiting for the newest one that must
    // be finished covers the older ones
    const int64_t upto = c->issued - keep;   // wait for collectives [waited, upto)
    if (upto <= c->waited) return PH_OK;
    PH_HIP(hipStreamWaitEvent(c->ctx->stream, c->done[(upto - 1) % ph_comm::RING], 0));
    c->waited = upto;
    return PH_OK;
}

extern "C" int ph_comm_wait(ph_comm *c) { return ph_comm_wait_keep(c, 0); }

extern "C" int ph_comm_allgather(ph_comm *c, const void *send_dev, void *recv_dev, int64_t bytes, int32_t async) {
    PH_REQUIRE(c && bytes >= 0 && (bytes == 0 || (send_dev && recv_dev)), "ph_comm_allgather: bad arguments");
    if (bytes == 0) return PH_OK;
    if (c->local) {
        std::vector<LocalCopy> cps;
        for (int r = 0; r < c->nranks; r++) cps.push_back(LocalCopy{r, 0, 0, (int64_t)r * bytes, bytes});
        return local_collective(c, {send_dev}, {recv_dev}, cps);
    }
    if (!async) {
        PH_CHECK(ph_comm_wait(c));
        PH_NCCL(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclChar, c->comm, c->ctx->stream));
        return PH_OK;
    }
    // the previous asynchronous collective (if any) runs on the same side stream: ordered before this one
    PH_HIP(hipEventRecord(c->ready, c->ctx->stream));
    PH_HIP(hipStreamWaitEvent(c->cstream, c->ready, 0));
    PH_NCCL(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclChar, c->comm, c->cstream));
    // an event of the ring may only be re-recorded once the ctx stream has waited for its last use
    if (c->issued - c->waited >= ph_comm::RING) PH_CHECK(ph_comm_wait_keep(c, ph_comm::RING - 1));
    PH_HIP(hipEventRecord(c->done[c->issued % ph_comm::RING], c->cstream));
    c->issued++;
    return PH_OK;
}

extern "C" int ph_comm_allreduce_i64(ph_comm *c, int64_t *host_vals, int32_t n, int32_t op) {
    PH_REQUIRE(c && host_vals && n >= 1 && n <= 64 && op >= PH_RED_SUM && op <= PH_RED_MIN, "ph_comm_allreduce_i64: bad arguments (n <= 64)");
    if (c->local) {
        std::vector<int64_t> all;
        PH_CHECK(local_allgather_host(c, host_vals, n, &all));
        for (int i = 0; i < n; i++) {
            int64_t v = all[(size_t)i];
            for (int r = 1; r < c->nranks; r++) {
                const int64_t x = all[(size_t)r * n + i];
                v = op == PH_RED_SUM ? v + x : op == PH_RED_MAX ? std::max(v, x) : std::min(v, x);
            }
            host_vals[i] = v;
        }
        return PH_OK;
    }
    PH_CHECK(ph_comm_wait(c));
    hipStream_t st = c->ctx->stream;
    int64_t *d = c->dev_words + (size_t)c->nranks * c->nranks;
    PH_HIP(hipMemcpyAsync(d, host_vals, (size_t)n * 8, hipMemcpyHostToDevice, st));
    ncclRedOp_t rop = op == PH_RED_SUM ? ncclSum : op == PH_RED_MAX ? ncclMax : ncclMin;
    PH_NCCL(ncclAllReduce(d, d, (size_t)n, ncclInt64, rop, c->comm, st));
    return c->ctx->download_plain(host_vals, d, (int64_t)n * 8);
}

extern "C" int ph_comm_barrier(ph_comm *c) {
    int64_t one = 1;
    return ph_comm_allreduce_i64(c, &one, 1, PH_RED_SUM);
}

extern "C" int ph_comm_exchange_counts(ph_comm *c, const int64_t *send_counts_dev, int64_t *matrix_host) {
    PH_REQUIRE(c && send_counts_dev && matrix_host, "ph_comm_exchange_counts: bad arguments");
    if (c->local) {
        std::vector<int64_t> mine((size_t)c->nranks), all;
        PH_CHECK(c->ctx->download_plain(mine.data(), send_counts_dev, (int64_t)c->nranks * 8));
        PH_CHECK(local_allgather_host(c, mine.data(), c->nranks, &all));
        memcpy(matrix_host, all.data(), all.size() * 8);
        return PH_OK;
    }
    PH_CHECK(ph_comm_wait(c));
    PH_NCCL(ncclAllGather(send_counts_dev, c->dev_words, (size_t)c->nranks, ncclInt64, c->comm, c->ctx->stream));
    return c->ctx->download_plain(matrix_host, c->dev_words, (int64_t)c->nranks * c->nranks * 8);   // the stage's one round trip
}

extern "C" int ph_exchange_layout(const int64_t *matrix, int32_t nranks, int32_t rank, int64_t *send_off, int64_t *recv_off) {
    PH_REQUIRE(matrix && nranks >= 1 && rank >= 0 && rank < nranks && send_off && recv_off, "ph_exchange_layout: bad arguments");
    send_off[0] = recv_off[0] = 0;
    for (int r = 0; r < nranks; r++) {
        int64_t s = matrix[(size_t)rank * nranks + r], v = matrix[(size_t)r * nranks + rank];
        PH_REQUIRE(s >= 0 && v >= 0, "ph_exchange_layout: negative count");
        send_off[r + 1] = send_off[r] + s;      // rows this rank sends to rank r (its row of the matrix)
        recv_off[r + 1] = recv_off[r] + v;      // rows arriving from rank r (its column), in rank order
    }
    return PH_OK;
}

// Inside ncclGroupStart/End every rank must make the same calls: a rank that returned early would leave its peers
// blocked in the group. So everything that can fail by ARGUMENT (pointers, widths, the matrix) is validated first —
// from values that are identical on all ranks or local — the group holds nothing but the sends and receives, and
// the own-rows device copy comes after it.
extern "C" int ph_comm_exchange_columns(ph_comm *c, int32_t ncols, const void *const *send_dev, void *const *recv_dev,
                                        const int32_t *elem_bytes, const int64_t *matrix_host) {
    PH_REQUIRE(c && ncols >= 0 && (ncols == 0 || (send_dev && recv_dev && elem_bytes)) && matrix_host,
               "ph_comm_exchange_columns: bad arguments");
    std::vector<int64_t> so((size_t)c->nranks + 1), ro((size_t)c->nranks + 1);
    PH_CHECK(ph_exchange_layout(matrix_host, c->nranks, c->rank, so.data(), ro.data()));
    for (int32_t k = 0; k < ncols; k++) {
        PH_REQUIRE(elem_bytes[k] > 0, "ph_comm_exchange_columns: column %d has element width %d", k, elem_bytes[k]);
        PH_REQUIRE(so[(size_t)c->nranks] == 0 || send_dev[k], "ph_comm_exchange_columns: send buffer %d is NULL", k);
        PH_REQUIRE(ro[(size_t)c->nranks] == 0 || recv_dev[k], "ph_comm_exchange_columns: receive buffer %d is NULL", k);
    }
    if (c->local) {
        // this rank reads, from every peer s, the rows s sends to it: they start at s's send offset for this rank (the prefix of s's row of the matrix)
        std::vector<LocalCopy> cps;
        std::vector<const void *> sp(send_dev, send_dev + ncols);
        std::vector<void *> rp(recv_dev, recv_dev + ncols);
        for (int s2 = 0; s2 < c->nranks; s2++) {
            int64_t src_row = 0;
            for (int d = 0; d < c->rank; d++) src_row += matrix_host[(size_t)s2 * c->nranks + d];
            const int64_t rows = matrix_host[(size_t)s2 * c->nranks + c->rank];
            for (int32_t k = 0; k < ncols; k++) cps.push_back(LocalCopy{s2, k, src_row * elem_bytes[k], ro[(size_t)s2] * elem_bytes[k], rows * elem_bytes[k]});
        }
        return local_collective(c, sp, rp, cps);
    }
    PH_CHECK(ph_comm_wait(c));
    hipStream_t st = c->ctx->stream;
    PH_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int32_t k = 0; k < ncols; k++) {
        const int64_t w = elem_bytes[k];
        for (int r = 0; r < c->nranks; r++) {
            if (r == c->rank) continue;
            const int64_t ns = so[(size_t)r + 1] - so[(size_t)r], nr = ro[(size_t)r + 1] - ro[(size_t)r];
            // a failing call is remembered, the remaining calls are still made: the group stays symmetric
            if (ns > 0) { ncclResult_t e = ncclSend((const char *)send_dev[k] + so[(size_t)r] * w, (size_t)(ns * w), ncclChar, r, c->comm, st); if (bad == ncclSuccess) bad = e; }
            if (nr > 0) { ncclResult_t e = ncclRecv((char *)recv_dev[k] + ro[(size_t)r] * w, (size_t)(nr * w), ncclChar, r, c->comm, st); if (bad == ncclSuccess) bad = e; }
        }
    }
    ncclResult_t endr = ncclGroupEnd();
    if (bad != ncclSuccess || endr != ncclSuccess) {
        ph::set_error("ph_comm_exchange_columns: %s", ncclGetErrorString(bad != ncclSuccess ? bad : endr));
        return PH_EHIP;
    }
    const int me = c->rank;   // own rows: a device copy, no link involved
    const int64_t ns = so[(size_t)me + 1] - so[(size_t)me];
    for (int32_t k = 0; k < ncols && ns > 0; k++) {
        const int64_t w = elem_bytes[k];
        PH_HIP(hipMemcpyAsync((char *)recv_dev[k] + ro[(size_t)me] * w, (const char *)send_dev[k] + so[(size_t)me] * w, (size_t)(ns * w),
                              hipMemcpyDeviceToDevice, st));
    }
    return PH_OK;
}

// The rows of every rank behind the counts: one group of send/recv pairs (counts identical on all ranks)
static int allgather_rows_group(ph_comm *c, const void *send_dev, int64_t count, int32_t elem_bytes, void *recv_dev,
                                const int64_t *counts) {
    if (c->local) {
        std::vector<LocalCopy> cps;
        int64_t off = 0;
        for (int r = 0; r < c->nranks; r++) { cps.push_back(LocalCopy{r, 0, 0, off * elem_bytes, counts[r] * elem_bytes}); off += counts[r]; }
        (void)count;
        return local_collective(c, {send_dev}, {recv_dev}, cps);
    }
    hipStream_t st = c->ctx->stream;
    PH_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    int64_t off = 0, own_off = 0;
    for (int r = 0; r < c->nranks; r++) {
        const int64_t nr = counts[r];
        if (r == c->rank) own_off = off;
        else {
            if (count > 0) { ncclResult_t e = ncclSend(send_dev, (size_t)(count * elem_bytes), ncclChar, r, c->comm, st); if (bad == ncclSuccess) bad = e; }
            if (nr > 0) { ncclResult_t e = ncclRecv((char *)recv_dev + off * elem_bytes, (size_t)(nr * elem_bytes), ncclChar, r, c->comm, st); if (bad == ncclSuccess) bad = e; }
        }
        off += nr;
    }
    ncclResult_t endr = ncclGroupEnd();
    if (bad != ncclSuccess || endr != ncclSuccess) {
        ph::set_error("ph_comm_allgather_rows: %s", ncclGetErrorString(bad != ncclSuccess ? bad : endr));
        return PH_EHIP;
    }
    if (count > 0)
        PH_HIP(hipMemcpyAsync((char *)recv_dev + own_off * elem_bytes, send_dev, (size_t)(count * elem_bytes), hipMemcpyDeviceToDevice, st));
    return PH_OK;
}

// The capacity decision is COLLECTIVE: every rank contributes (count, capacity), and the rows are exchanged only
// when the total fits the SMALLEST capacity — otherwise every rank returns PH_ECAPACITY, together, with the
// counts filled in, and nobody has entered the send/recv group. (Round 2 decided per rank: a rank with a small
// local count sized its buffer too small, left with PH_ECAPACITY, and its peers blocked in the group.)
extern "C" int ph_comm_allgather_rows(ph_comm *c, const void *send_dev, int64_t count, int32_t elem_bytes, void *recv_dev,
                                      int64_t recv_capacity, int64_t *counts_host) {
    PH_REQUIRE(c && count >= 0 && elem_bytes > 0 && counts_host && (count == 0 || send_dev) && recv_capacity >= 0 &&
               (recv_capacity == 0 || recv_dev), "ph_comm_allgather_rows: bad arguments");
    PH_CHECK(ph_comm_wait(c));
    hipStream_t st = c->ctx->stream;
    int64_t *d = c->dev_words;   // [0, 2 n): every rank's (count, capacity); [2 n, 2 n + 2): this rank's pair
    const int64_t mine[2] = {count, recv_capacity};
    std::vector<int64_t> pairs((size_t)c->nranks * 2);
    if (c->local) PH_CHECK(local_allgather_host(c, mine, 2, &pairs));
    else {
        PH_HIP(hipMemcpyAsync(d + 2 * c->nranks, mine, 16, hipMemcpyHostToDevice, st));
        PH_NCCL(ncclAllGather(d + 2 * c->nranks, d, 2, ncclInt64, c->comm, st));
        PH_CHECK(c->ctx->download_plain(pairs.data(), d, (int64_t)c->nranks * 16));
    }
    int64_t total = 0, mincap = INT64_MAX;
    for (int r = 0; r < c->nranks; r++) { counts_host[r] = pairs[(size_t)r * 2]; total += counts_host[r]; mincap = std::min(mincap, pairs[(size_t)r * 2 + 1]); }
    if (total > mincap) {
        ph::set_error("ph_comm_allgather_rows: %lld rows in total, the smallest receive capacity over the ranks is %lld",
                      (long long)total, (long long)mincap);
        return PH_ECAPACITY;
    }
    if (total == 0) return PH_OK;
    return allgather_rows_group(c, send_dev, count, elem_bytes, recv_dev, counts_host);
}

// ... and the form that cannot run out of room: the library allocates exactly the total (ph_dev_alloc's pool; the
// caller frees with ph_dev_free). *recv_dev_out is a valid (possibly 1-row) allocation even when the total is 0.
extern "C" int ph_comm_allgather_rows_alloc(ph_comm *c, const void *send_dev, int64_t count, int32_t elem_bytes, void **recv_dev_out,
                                            int64_t *counts_host) {
    PH_REQUIRE(c && count >= 0 && elem_bytes > 0 && counts_host && recv_dev_out && (count == 0 || send_dev),
               "ph_comm_allgather_rows_alloc: bad arguments");
    PH_CHECK(ph_comm_wait(c));
    hipStream_t st = c->ctx->stream;
    int64_t *d = c->dev_words;
    if (c->local) {
        std::vector<int64_t> all;
        PH_CHECK(local_allgather_host(c, &count, 1, &all));
        memcpy(counts_host, all.data(), all.size() * 8);
    } else {
        PH_HIP(hipMemcpyAsync(d + c->nranks, &count, 8, hipMemcpyHostToDevice, st));
        PH_NCCL(ncclAllGather(d + c->nranks, d, 1, ncclInt64, c->comm, st));
        PH_CHECK(c->ctx->download_plain(counts_host, d, (int64_t)c->nranks * 8));
    }
    int64_t total = 0;
    for (int r = 0; r < c->nranks; r++) total += counts_host[r];
    // an allocation failure here is local, but it is a failure of the process (out of device memory), not a retry path
    PH_CHECK(ph_dev_alloc(c->ctx, std::max<int64_t>(total, 1) * elem_bytes, recv_dev_out));
    if (total == 0) return PH_OK;
    return allgather_rows_group(c, send_dev, count, elem_bytes, *recv_dev_out, counts_host);
}
