// Context, error reporting, device buffers and table residency.
//
// Table residency replaces the per-chunk materialisation of DataTable.Scan -> scanRows
// (reference pkg/storage/table.go:418-428, pkg/compute/executor_scan.go:158-241, which allocates a
// fresh 2048-row Chunk per call): the pruned columns are staged through pinned host memory and
// copied once with hipMemcpyAsync into HBM, in the narrow encodings of SURVEY.md §8(d).
#include <algorithm>

#include "common.h"

namespace ph {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

}  // namespace ph

extern "C" const char *ph_last_error(void) { return ph::g_err; }
extern "C" const char *ph_version(void) { return "planhip 0.1 (gfx950)"; }

int ph_ctx::ensure_scratch(int64_t bytes) {
    if (bytes <= scratch_bytes) return PH_OK;
    if (scratch) {
        PH_HIP(hipStreamSynchronize(stream));
        PH_HIP(hipFree(scratch));
        scratch = nullptr;
        scratch_bytes = 0;
    }
    bytes = ph::round_up(bytes, 1 << 20);
    PH_HIP(hipMalloc(&scratch, bytes));
    scratch_bytes = bytes;
    return PH_OK;
}

int ph_ctx::ensure_pinned(int64_t bytes) {
    if (bytes <= pinned_bytes) return PH_OK;
    if (pinned) {
        PH_HIP(hipStreamSynchronize(stream));
        PH_HIP(hipHostFree(pinned));
        pinned = nullptr;
        pinned_bytes = 0;
    }
    bytes = ph::round_up(bytes, 1 << 20);
    PH_HIP(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    pinned_bytes = bytes;
    return PH_OK;
}

constexpr int64_t PH_MAILBOX = 64 << 10;   // + 64 bytes behind it for the deferred-error words

int ph_ctx::deferred_words(int **out) {
    if (!deferred_dev) {
        PH_HIP(hipMalloc((void **)&deferred_dev, 64));
        PH_HIP(hipMemsetAsync(deferred_dev, 0, 64, stream));
    }
    *out = deferred_dev;
    return PH_OK;
}

int ph_ctx::finish_deferred() {
    deferred_pending = false;
    const int *d = reinterpret_cast<const int *>((const char *)mailbox + PH_MAILBOX);
    if (!d[0] && !d[1] && !d[2] && !d[3]) return PH_OK;
    const int ovf = d[0], miss = d[1], multi = d[2], unsorted = d[3];
    PH_HIP(hipMemsetAsync(deferred_dev, 0, 64, stream));
    if (ovf) { ph::set_error("deferred from ph_expr_eval: a row left the exact int64 decimal domain"); return PH_EOVERFLOW; }
    if (unsorted) {
        // bit 0: a sorted fill's keys, bit 2: a streaming aggregate's rows (the messages are told apart by ph_plan: it retires only the form that broke)
        if (unsorted & 4) ph::set_error("deferred: ph_agg_sink_sorted rows are not ordered by the group key tuple (take ph_agg_sink)");
        else ph::set_error("deferred: a sorted-input claim does not hold (ph_join_build_ex PH_JOIN_KEYS_SORTED_UNIQUE build keys are not strictly ascending)");
        return PH_ECONSTRAINT;
    }
    ph::set_error("deferred from ph_join_lookup_strict: %d probe rows without a match, %d with more than one", miss, multi);
    return PH_ECONSTRAINT;
}

// ---- mailbox publish: a kernel stores a small result into mapped host memory, the host polls a sequence word
namespace ph {
__global__ __launch_bounds__(256) void publish_kernel(const unsigned char *__restrict__ src, int64_t bytes, const int *__restrict__ deferred,
                                                      unsigned char *__restrict__ mbox, int *__restrict__ mbox_deferred,
                                                      unsigned long long *__restrict__ flag, unsigned long long seq) {
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(mbox)) & 15) == 0) {
        const int64_t nv = bytes >> 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(mbox);
        for (int64_t i = threadIdx.x; i < nv; i += 256) d4[i] = s4[i];
        for (int64_t i = (nv << 4) + threadIdx.x; i < bytes; i += 256) mbox[i] = src[i];
    } else {
        for (int64_t i = threadIdx.x; i < bytes; i += 256) mbox[i] = src[i];
    }
    if (deferred && threadIdx.x < 4) mbox_deferred[threadIdx.x] = deferred[threadIdx.x];
    __threadfence_system();   // this thread's stores are visible to the host ...
    __syncthreads();          // ... for every thread of the (single) workgroup ...
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the number is
}
}  // namespace ph

// wait until *flag == seq (written by a publish kernel on `stream`). The poll is bounded: every few thousand
// reads it asks the stream — idle without the number means the kernel never ran (an earlier fault)
static int poll_flag(const unsigned long long *flag, unsigned long long seq, hipStream_t stream) {
    for (unsigned long long spins = 1;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return PH_OK;
        if ((spins & 8191) == 0) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return PH_OK;
                ph::set_error("device-to-host mailbox: the stream went idle without publishing (%s)", hipGetErrorString(hipGetLastError()));
                return PH_EHIP;
            }
            if (q != hipErrorNotReady) { ph::set_error("device-to-host mailbox: %s", hipGetErrorString(q)); return PH_EHIP; }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

int ph_ctx::publish(const void *dev, int64_t bytes, bool with_deferred) {
    const int64_t MB = PH_MAILBOX;
    unsigned long long *flag = reinterpret_cast<unsigned long long *>((char *)mailbox + MB + 64);
    unsigned long long *flag_dev = reinterpret_cast<unsigned long long *>((char *)mailbox_dev + MB + 64);
    const unsigned long long seq = ++publish_seq;
    ph::publish_kernel<<<1, 256, 0, stream>>>((const unsigned char *)dev, bytes, with_deferred ? deferred_dev : nullptr,
                                              (unsigned char *)mailbox_dev, (int *)((char *)mailbox_dev + MB), flag_dev, seq);
    PH_HIP(hipGetLastError());
    return poll_flag(flag, seq, stream);
}

int ph_ctx::ensure_mailbox() {
    if (mailbox) return PH_OK;
    const int64_t MB = PH_MAILBOX;
    PH_HIP(hipHostMalloc(&mailbox, (size_t)MB + 128, hipHostMallocMapped | hipHostMallocCoherent));
    memset((char *)mailbox + MB, 0, 128);
    PH_HIP(hipHostGetDevicePointer(&mailbox_dev, mailbox, 0));
    return PH_OK;
}

int ph_ctx::arm_publish(int64_t bytes, unsigned long long **mbox_dev, unsigned long long **flag_dev, unsigned long long *seq, bool deferred_ok) {
    *seq = 0;
    *mbox_dev = nullptr;
    *flag_dev = nullptr;
    static const bool no_publish = getenv("PH_NO_PUBLISH") != nullptr || getenv("PH_NO_SCAN_TAIL_PUBLISH") != nullptr;
    if (no_publish || bytes > (int64_t)PH_MAILBOX || (!deferred_ok && deferred_pending && deferred_dev && !defer_hold)) return PH_OK;
    PH_CHECK(ensure_mailbox());
    *mbox_dev = reinterpret_cast<unsigned long long *>(mailbox_dev);
    *flag_dev = reinterpret_cast<unsigned long long *>((char *)mailbox_dev + PH_MAILBOX + 64);
    *seq = ++publish_seq;
    return PH_OK;
}

int ph_ctx::collect_armed(void *host, int64_t bytes, unsigned long long seq, bool deferred_ok) {
    if (!seq || seq != publish_seq || (!deferred_ok && deferred_pending && deferred_dev && !defer_hold)) return 1;
    PH_CHECK(poll_flag(reinterpret_cast<const unsigned long long *>((const char *)mailbox + PH_MAILBOX + 64), seq, stream));
    memcpy(host, mailbox, (size_t)bytes);
    return PH_OK;
}

int ph_ctx::arm_count(ph::ScanPublish *pub) {
    *pub = ph::ScanPublish{};
    static const bool early_count = !(getenv("PH_EARLY_COUNT") && getenv("PH_EARLY_COUNT")[0] == '0');
    if (!early_count || async_counts) return PH_OK;
    PH_CHECK(arm_publish(8, &pub->mbox, &pub->flag, &pub->seq, true));
    if (pub->seq && deferred_pending && deferred_dev && !defer_hold) {   // pending deferred-error words: checked with this count, as a publish would
        pub->deferred = deferred_dev;
        pub->mbox_deferred = reinterpret_cast<int *>((char *)mailbox_dev + PH_MAILBOX);
    }
    return PH_OK;
}

int ph_ctx::count_back(const ph::ScanPublish &pub, int64_t *host, const void *total_dev, int64_t cap, const char *what) {
    if (pub.seq) {
        // (the words were copied only if they were pending at arming time: pending ones that were NOT copied — a kernel between arming and the
        // scan set the mode — take the ordinary way)
        const bool pending_now = deferred_pending && deferred_dev && !defer_hold;
        const int rc = pending_now && !pub.deferred ? 1 : collect_armed(host, 8, pub.seq, true);
        if (rc < 0) return rc;
        if (rc == 0 && pub.deferred) PH_CHECK(finish_deferred());
        if (rc == 0) {
            if (cap >= 0 && *host > cap) { ph::set_error("%s: %lld rows, output capacity %lld", what, (long long)*host, (long long)cap); return PH_ECAPACITY; }
            return PH_OK;
        }
    }
    return download_count(host, total_dev, cap, what);
}

int ph_ctx::download(void *host, const void *dev, int64_t bytes, bool with_deferred) {
    if (bytes <= 0) return PH_OK;
    const int64_t MB = PH_MAILBOX;
    PH_CHECK(ensure_mailbox());
    const bool chk = with_deferred && !defer_hold && deferred_pending && deferred_dev;
    static const bool no_publish = getenv("PH_NO_PUBLISH") != nullptr;   // A/B switch: copy command + stream synchronisation
    if (bytes <= MB && !no_publish) {
        PH_CHECK(publish(dev, bytes, chk));
        memcpy(host, mailbox, (size_t)bytes);
        return chk ? finish_deferred() : PH_OK;
    }
    if (chk) PH_HIP(hipMemcpyAsync((char *)mailbox + MB, deferred_dev, 16, hipMemcpyDeviceToHost, stream));
    if (bytes <= MB) {
        PH_HIP(hipMemcpyAsync(mailbox, dev, (size_t)bytes, hipMemcpyDeviceToHost, stream));
        PH_HIP(hipStreamSynchronize(stream));
        memcpy(host, mailbox, (size_t)bytes);
        return chk ? finish_deferred() : PH_OK;
    }
    const int64_t CH = 32ll << 20;
    PH_CHECK(ensure_pinned(2 * CH));
    // double buffered: the copy of chunk k+1 overlaps the memcpy of chunk k out of staging
    hipEvent_t ev[2];
    PH_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    PH_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = PH_OK;
    int64_t nchunks = (bytes + CH - 1) / CH;
    for (int64_t c = 0; c <= nchunks && rc == PH_OK; c++) {
        if (c < nchunks) {
            int64_t off = c * CH, len = std::min(CH, bytes - off);
            if (hipMemcpyAsync((char *)pinned + (c & 1) * CH, (const char *)dev + off, (size_t)len,
                               hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipEventRecord(ev[c & 1], stream) != hipSuccess) rc = PH_EHIP;
        }
        if (c > 0 && rc == PH_OK) {
            int64_t off = (c - 1) * CH, len = std::min(CH, bytes - off);
            if (hipEventSynchronize(ev[(c - 1) & 1]) != hipSuccess) rc = PH_EHIP;
            else memcpy((char *)host + off, (char *)pinned + ((c - 1) & 1) * CH, (size_t)len);
        }
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc != PH_OK) ph::set_error("device-to-host copy of %lld bytes failed", (long long)bytes);
    if (rc == PH_OK && chk) rc = finish_deferred();
    return rc;
}

constexpr int PH_MAX_PENDING_COUNTS = 64;

int ph_ctx::download_count(int64_t *host, const void *dev, int64_t cap, const char *what) {
    if (!async_counts || pending_counts.size() >= (size_t)PH_MAX_PENDING_COUNTS) {
        PH_CHECK(download(host, dev, 8));
        if (cap >= 0 && *host > cap) { ph::set_error("%s: %lld rows, output capacity %lld", what, (long long)*host, (long long)cap); return PH_ECAPACITY; }
        return PH_OK;
    }
    if (!count_slots) {   // + one word behind the slots: the sequence number of the last count published
        PH_HIP(hipHostMalloc((void **)&count_slots, (PH_MAX_PENDING_COUNTS + 2) * 8, hipHostMallocMapped | hipHostMallocCoherent));
        memset(count_slots, 0, (PH_MAX_PENDING_COUNTS + 2) * 8);
        PH_HIP(hipHostGetDevicePointer((void **)&count_slots_dev, count_slots, 0));
    }
    const size_t slot = pending_counts.size();
    ph::publish_kernel<<<1, 256, 0, stream>>>((const unsigned char *)dev, 8, nullptr, (unsigned char *)(count_slots_dev + slot), nullptr,
                                              (unsigned long long *)(count_slots_dev + PH_MAX_PENDING_COUNTS), ++count_seq_issued);
    PH_HIP(hipGetLastError());
    pending_counts.push_back({host, cap, what});
    *host = -1;
    return PH_OK;
}

int ph_ctx::wait_counts() {
    if (pending_counts.empty()) return PH_OK;
    // the counts were published in stream order: the last one's number means all of them have arrived
    PH_CHECK(poll_flag((const unsigned long long *)(count_slots + PH_MAX_PENDING_COUNTS), count_seq_issued, stream));
    int rc = PH_OK;
    for (size_t i = 0; i < pending_counts.size(); i++) {
        const PendingCount &p = pending_counts[i];
        *p.host = count_slots[i];
        if (rc == PH_OK && p.cap >= 0 && *p.host > p.cap) {
            ph::set_error("%s: %lld rows, output capacity %lld", p.what, (long long)*p.host, (long long)p.cap);
            rc = PH_ECAPACITY;
        }
    }
    pending_counts.clear();
    return rc;
}

extern "C" int ph_ctx_set_async_counts(ph_ctx *ctx, int32_t on) {
    PH_REQUIRE(ctx, "ph_ctx_set_async_counts: ctx is NULL");
    if (!on) PH_CHECK(ctx->wait_counts());
    ctx->async_counts = on != 0;
    return PH_OK;
}

extern "C" int ph_ctx_wait_counts(ph_ctx *ctx) {
    PH_REQUIRE(ctx, "ph_ctx_wait_counts: ctx is NULL");
    return ctx->wait_counts();
}

extern "C" int ph_ctx_set_deferred_errors(ph_ctx *ctx, int32_t on) {
    PH_REQUIRE(ctx, "ph_ctx_set_deferred_errors: ctx is NULL");
    ctx->defer_errors = on != 0;
    ctx->defer_hold = on == 2;
    return PH_OK;
}

extern "C" int ph_ctx_check_deferred(ph_ctx *ctx) {
    PH_REQUIRE(ctx, "ph_ctx_check_deferred: ctx is NULL");
    if (!ctx->deferred_pending || !ctx->deferred_dev) return PH_OK;
    int d[4];
    const bool hold = ctx->defer_hold;
    ctx->defer_hold = false;   // this IS the call that reports a held error
    const int rc = ctx->download(d, ctx->deferred_dev, 16);
    ctx->defer_hold = hold;
    return rc;
}

int ph_ctx::pool_alloc(int64_t bytes, void **out) {
    int64_t sz = ph::round_up(bytes > 0 ? bytes : 1, bytes > (1 << 20) ? (2 << 20) : 4096);
    auto it = pool_free_blocks.find(sz);
    if (it != pool_free_blocks.end()) {
        *out = it->second;
        pool_free_blocks.erase(it);
        return PH_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)sz);
    if (e != hipSuccess) {
        // give cached blocks back to the driver and retry once
        PH_HIP(hipStreamSynchronize(stream));
        for (auto &kv : pool_free_blocks) { pool_sizes.erase(kv.second); (void)hipFree(kv.second); }
        pool_free_blocks.clear();
        PH_HIP(hipMalloc(&p, (size_t)sz));
    }
    pool_sizes[p] = sz;
    *out = p;
    return PH_OK;
}

void ph_ctx::pool_release(void *p) {
    if (!p) return;
    auto it = pool_sizes.find(p);
    if (it == pool_sizes.end()) { (void)hipFree(p); return; }
    pool_free_blocks.emplace(it->second, p);
}

void ph_ctx::pool_destroy() {
    for (auto &kv : pool_sizes) (void)hipFree(kv.first);
    pool_sizes.clear();
    pool_free_blocks.clear();
}

extern "C" int ph_ctx_create(int device, ph_ctx **out) {
    PH_REQUIRE(out != nullptr, "ph_ctx_create: out is NULL");
    int ndev = 0;
    PH_HIP(hipGetDeviceCount(&ndev));
    PH_REQUIRE(device >= 0 && device < ndev, "ph_ctx_create: device %d of %d", device, ndev);
    PH_HIP(hipSetDevice(device));
    ph_ctx *c = new ph_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->cu_count = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        ph::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return PH_EHIP;
    }
    c->own_stream = true;
    *out = c;
    return PH_OK;
}

extern "C" int ph_ctx_set_stream(ph_ctx *ctx, void *hip_stream) {
    PH_REQUIRE(ctx != nullptr, "ph_ctx_set_stream: ctx is NULL");
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) PH_HIP(hipStreamDestroy(ctx->stream));
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        PH_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return PH_OK;
}

extern "C" int ph_ctx_sync(ph_ctx *ctx) {
    PH_REQUIRE(ctx != nullptr, "ph_ctx_sync: ctx is NULL");
    PH_HIP(hipStreamSynchronize(ctx->stream));
    return PH_OK;
}

extern "C" void ph_ctx_destroy(ph_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pool_destroy();
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->scan_state) (void)hipFree(ctx->scan_state);
    if (ctx->deferred_dev) (void)hipFree(ctx->deferred_dev);
    if (ctx->scan_done_dev) (void)hipFree(ctx->scan_done_dev);
    if (ctx->count_slots) (void)hipHostFree(ctx->count_slots);
    if (ctx->count_event) (void)hipEventDestroy((hipEvent_t)ctx->count_event);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---------------------------------------------------------------- plain device buffers

extern "C" int ph_dev_alloc(ph_ctx *ctx, int64_t bytes, void **dev) {
    PH_REQUIRE(ctx && dev && bytes >= 0, "ph_dev_alloc: bad arguments");
    PH_HIP(hipSetDevice(ctx->device));
    return ctx->pool_alloc(bytes, dev);
}

extern "C" int ph_dev_free_many(ph_ctx *ctx, void *const *devs, int64_t n) {
    PH_REQUIRE(ctx != nullptr && n >= 0 && (n == 0 || devs), "ph_dev_free_many: bad arguments");
    for (int64_t i = 0; i < n; i++) ctx->pool_release(devs[i]);   // stream-ordered reuse; no synchronisation needed
    return PH_OK;
}

extern "C" int ph_dev_free(ph_ctx *ctx, void *dev) {
    PH_REQUIRE(ctx != nullptr, "ph_dev_free: ctx is NULL");
    ctx->pool_release(dev);  // stream-ordered reuse; no synchronisation needed
    return PH_OK;
}

// Pageable host memory (Go heap / numpy) -> pinned staging -> device, double buffered.
static int upload_staged(ph_ctx *ctx, void *dev, const void *host, int64_t bytes) {
    const int64_t CH = 32ll << 20;
    PH_CHECK(ctx->ensure_pinned(2 * CH));
    hipEvent_t ev[2];
    PH_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    PH_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = PH_OK;
    int k = 0;
    for (int64_t off = 0; off < bytes; off += CH, k ^= 1) {
        int64_t len = bytes - off < CH ? bytes - off : CH;
        char *stage = (char *)ctx->pinned + (int64_t)k * CH;
        if (off >= 2 * CH && hipEventSynchronize(ev[k]) != hipSuccess) { rc = PH_EHIP; break; }
        memcpy(stage, (const char *)host + off, (size_t)len);
        if (hipMemcpyAsync((char *)dev + off, stage, (size_t)len, hipMemcpyHostToDevice,
                           ctx->stream) != hipSuccess ||
            hipEventRecord(ev[k], ctx->stream) != hipSuccess) {
            rc = PH_EHIP;
            break;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = PH_EHIP;
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc != PH_OK) ph::set_error("staged upload of %lld bytes failed", (long long)bytes);
    return rc;
}

extern "C" int ph_dev_upload(ph_ctx *ctx, void *dev, const void *host, int64_t bytes) {
    PH_REQUIRE(ctx && (bytes == 0 || (dev && host)), "ph_dev_upload: bad arguments");
    if (bytes == 0) return PH_OK;
    PH_HIP(hipSetDevice(ctx->device));
    return upload_staged(ctx, dev, host, bytes);
}

extern "C" int ph_dev_download(ph_ctx *ctx, void *host, const void *dev, int64_t bytes) {
    PH_REQUIRE(ctx && (bytes == 0 || (dev && host)), "ph_dev_download: bad arguments");
    if (bytes == 0) return PH_OK;
    PH_HIP(hipSetDevice(ctx->device));
    return ctx->download(host, dev, bytes);
}

extern "C" int ph_dev_memset(ph_ctx *ctx, void *dev, int value, int64_t bytes) {
    PH_REQUIRE(ctx && (bytes == 0 || dev), "ph_dev_memset: bad arguments");
    if (bytes == 0) return PH_OK;
    PH_HIP(hipMemsetAsync(dev, value, (size_t)bytes, ctx->stream));
    return PH_OK;
}

// ---------------------------------------------------------------- column statistics

template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T *__restrict__ v, int64_t n,
                                                     long long *__restrict__ out) {
    long long lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        long long x = (long long)v[i];
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}

static int column_range(ph_ctx *ctx, int32_t type, const void *dev, int64_t n, int64_t *mn,
                        int64_t *mx) {
    PH_CHECK(ctx->ensure_scratch(64));
    long long init[2] = {INT64_MAX, INT64_MIN};
    PH_HIP(hipMemcpyAsync(ctx->scratch, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    long long *out = (long long *)ctx->scratch;
    switch (type) {
    case PH_I32: case PH_DATE:
        minmax_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t *)dev, n, out);
        break;
    case PH_I64: case PH_DEC64:
        minmax_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t *)dev, n, out);
        break;
    case PH_CODE8:
        minmax_kernel<uint8_t><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)dev, n, out);
        break;
    default:
        return PH_EUNSUPPORTED;
    }
    PH_HIP(hipGetLastError());
    long long res[2];
    PH_CHECK(ctx->download(res, ctx->scratch, sizeof res));
    *mn = res[0];
    *mx = res[1];
    return PH_OK;
}

// order statistics: out[0] |= 1 when a row is below its predecessor, |= 2 when it is not above it
template <typename T>
__global__ __launch_bounds__(256) void order_stat_kernel(const T *__restrict__ v, int64_t n, int *__restrict__ out) {
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T a = v[i], b = v[i + 1];
        f |= (b < a ? 1 : 0) | (b <= a ? 2 : 0);
    }
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(out, f);
}

static int column_order(ph_ctx *ctx, int32_t type, const void *dev, int64_t n, bool *ascending, bool *strict) {
    PH_CHECK(ctx->ensure_scratch(64));
    PH_HIP(hipMemsetAsync(ctx->scratch, 0, 8, ctx->stream));
    int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    if (grid < 1) grid = 1;
    int *out = (int *)ctx->scratch;
    if (type == PH_I32 || type == PH_DATE) order_stat_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t *)dev, n, out);
    else if (type == PH_I64 || type == PH_DEC64) order_stat_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t *)dev, n, out);
    else return PH_EUNSUPPORTED;
    PH_HIP(hipGetLastError());
    int f = 0;
    PH_CHECK(ctx->download(&f, out, 4));
    *ascending = (f & 1) == 0;
    *strict = (f & 2) == 0;
    return PH_OK;
}

// run statistic: out[0] |= 1 when some row i does not hold mn + i / c
template <typename T>
__global__ __launch_bounds__(256) void run_stat_kernel(const T *__restrict__ v, int64_t n, long long mn, int c, int *__restrict__ out) {
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) f |= (long long)v[i] != mn + i / c ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(out, f);
}

static int column_runs(ph_ctx *ctx, int32_t type, const void *dev, int64_t n, int64_t mn, int64_t mx, int32_t *run_len) {
    *run_len = 0;
    const __int128 span = (__int128)mx - (__int128)mn + 1;
    if (span <= 0 || span > n || n % (int64_t)span != 0) return PH_OK;
    const int64_t c = n / (int64_t)span;
    if (c < 2 || c > 64) return PH_OK;
    PH_CHECK(ctx->ensure_scratch(64));
    PH_HIP(hipMemsetAsync(ctx->scratch, 0, 8, ctx->stream));
    int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    int *out = (int *)ctx->scratch;
    if (type == PH_I32 || type == PH_DATE) run_stat_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t *)dev, n, (long long)mn, (int)c, out);
    else if (type == PH_I64 || type == PH_DEC64) run_stat_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t *)dev, n, (long long)mn, (int)c, out);
    else return PH_OK;
    PH_HIP(hipGetLastError());
    int f = 0;
    PH_CHECK(ctx->download(&f, out, 4));
    if (!f) *run_len = (int32_t)c;
    return PH_OK;
}

// ---------------------------------------------------------------- tables

extern "C" int ph_table_create(ph_ctx *ctx, int32_t ncols, const ph_col *host_cols, int64_t nrows,
                               ph_table **out) {
    PH_REQUIRE(ctx && host_cols && out && ncols > 0 && nrows >= 0, "ph_table_create: bad arguments");
    PH_REQUIRE(nrows < (1ll << 31), "ph_table_create: %lld rows exceed the int32 row-id domain",
               (long long)nrows);
    PH_HIP(hipSetDevice(ctx->device));
    ph_table *t = new ph_table();
    t->ctx = ctx;
    t->nrows = nrows;
    t->cols.resize((size_t)ncols);
    int64_t padded = ph::round_up(nrows > 0 ? nrows : 1, PH_ROW_PAD);
    int rc = PH_OK;
    for (int32_t c = 0; c < ncols && rc == PH_OK; c++) {
        const ph_col &h = host_cols[c];
        ph_table::column &d = t->cols[(size_t)c];
        d.type = h.type;
        d.scale = h.scale;
        if (h.type == PH_STR) {
            int64_t off_bytes = (padded + 1) * 4;
            if (hipMalloc(&d.data, (size_t)off_bytes) != hipSuccess ||
                hipMemsetAsync(d.data, 0, (size_t)off_bytes, ctx->stream) != hipSuccess) { rc = PH_EHIP; break; }
            rc = upload_staged(ctx, d.data, h.data, (nrows + 1) * 4);
            if (rc != PH_OK) break;
            d.aux_bytes = h.aux_bytes;
            if (hipMalloc(&d.aux, (size_t)(h.aux_bytes + 64)) != hipSuccess) { rc = PH_EHIP; break; }
            rc = upload_staged(ctx, d.aux, h.aux, h.aux_bytes);
        } else {
            int w = ph::type_width(h.type);
            if (w == 0) { ph::set_error("ph_table_create: column %d has unknown type %d", c, h.type); rc = PH_EINVAL; break; }
            if (hipMalloc(&d.data, (size_t)(padded * w)) != hipSuccess) { rc = PH_EHIP; break; }
            // zero the padding so out-of-range lanes read defined values
            if (hipMemsetAsync((char *)d.data + nrows * w, 0, (size_t)((padded - nrows) * w), ctx->stream) != hipSuccess) { rc = PH_EHIP; break; }
            rc = upload_staged(ctx, d.data, h.data, nrows * w);
            if (rc != PH_OK) break;
            if (h.type == PH_CODE8 && h.aux) { // dictionary: NUL-separated strings
                const char *p = (const char *)h.aux, *end = p + h.aux_bytes;
                while (p < end) {
                    size_t len = strnlen(p, (size_t)(end - p));
                    d.dict.emplace_back(p, len);
                    p += len + 1;
                }
            }
            if (nrows > 0 && (h.type == PH_I32 || h.type == PH_DATE || h.type == PH_I64 ||
                              h.type == PH_DEC64 || h.type == PH_CODE8)) {
                rc = column_range(ctx, h.type, d.data, nrows, &d.min, &d.max);
                d.has_range = rc == PH_OK;
                if (rc == PH_OK && h.type != PH_CODE8 && !h.validity) rc = column_order(ctx, h.type, d.data, nrows, &d.ascending, &d.strict);
                if (rc == PH_OK && d.ascending && !d.strict) rc = column_runs(ctx, h.type, d.data, nrows, d.min, d.max, &d.run_len);
            }
        }
        if (rc == PH_OK && h.validity) {
            int64_t vb = padded / 8;
            if (hipMalloc((void **)&d.validity, (size_t)vb) != hipSuccess ||
                hipMemsetAsync(d.validity, 0, (size_t)vb, ctx->stream) != hipSuccess) { rc = PH_EHIP; break; }
            rc = upload_staged(ctx, d.validity, h.validity, (nrows + 7) / 8);
        }
    }
    if (rc != PH_OK) {
        if (rc == PH_EHIP && ph_last_error()[0] == 0) ph::set_error("ph_table_create: HIP allocation/copy failed");
        ph_table_free(t);
        return rc;
    }
    ph::register_table(t);
    *out = t;
    return PH_OK;
}

// ---- the process-wide table registry (column base pointer -> table, column)
namespace ph {
static std::mutex g_tables_mu;
static std::map<const void *, std::pair<ph_table *, int>> g_table_cols;
void register_table(ph_table *t) {
    std::lock_guard<std::mutex> lock(g_tables_mu);
    for (size_t c = 0; c < t->cols.size(); c++) if (t->cols[c].data) g_table_cols[t->cols[c].data] = {t, (int)c};
}
void unregister_table(ph_table *t) {
    std::lock_guard<std::mutex> lock(g_tables_mu);
    for (auto &c : t->cols) {
        auto it = g_table_cols.find(c.data);
        if (c.data && it != g_table_cols.end() && it->second.first == t) g_table_cols.erase(it);
    }
}
bool lookup_table_col(const void *data, ph_table **t, int *col) {
    std::lock_guard<std::mutex> lock(g_tables_mu);
    auto it = g_table_cols.find(data);
    if (it == g_table_cols.end()) return false;
    *t = it->second.first;
    *col = it->second.second;
    return true;
}
}  // namespace ph

extern "C" int64_t ph_table_rows(const ph_table *t) { return t ? t->nrows : -1; }
extern "C" int32_t ph_table_ncols(const ph_table *t) { return t ? (int32_t)t->cols.size() : -1; }

extern "C" int ph_table_col(const ph_table *t, int32_t c, ph_col *out) {
    PH_REQUIRE(t && out && c >= 0 && c < (int32_t)t->cols.size(), "ph_table_col: bad column %d", c);
    const ph_table::column &d = t->cols[(size_t)c];
    out->type = d.type;
    out->scale = d.scale;
    out->data = d.data;
    out->validity = d.validity;
    out->aux = d.aux;
    out->aux_bytes = d.aux_bytes;
    return PH_OK;
}

extern "C" int ph_table_col_range(const ph_table *t, int32_t c, int64_t *mn, int64_t *mx) {
    PH_REQUIRE(t && c >= 0 && c < (int32_t)t->cols.size(), "ph_table_col_range: bad column %d", c);
    const ph_table::column &d = t->cols[(size_t)c];
    if (!d.has_range) { ph::set_error("column %d has no range statistics", c); return PH_EUNSUPPORTED; }
    if (mn) *mn = d.min;
    if (mx) *mx = d.max;
    return PH_OK;
}

extern "C" int ph_table_col_stats(const ph_table *t, int32_t c, int32_t *flags) {
    PH_REQUIRE(t && flags && c >= 0 && c < (int32_t)t->cols.size(), "ph_table_col_stats: bad column %d", c);
    const ph_table::column &d = t->cols[(size_t)c];
    *flags = (d.ascending ? PH_STAT_ASCENDING : 0) | (d.strict ? PH_STAT_STRICT : 0);
    for (auto &u : t->unique_keys) if (u.size() == 1 && u[0] == c) *flags |= PH_STAT_DECLARED_UNIQUE;
    return PH_OK;
}

extern "C" int32_t ph_table_col_run_len(const ph_table *t, int32_t c) {
    if (!t || c < 0 || c >= (int32_t)t->cols.size()) return 0;
    return t->cols[(size_t)c].run_len;
}

// ---- co-located column groups
struct ColocateParams {
    const void *src[8];
    int off[8], width[8];
    int ncols, stride;
};
__global__ __launch_bounds__(256) void colocate_kernel(ColocateParams P, int64_t n, unsigned char *__restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        unsigned char *row = out + r * P.stride;
        for (int c = 0; c < P.ncols; c++) {   // offsets are multiples of the width: aligned stores
            if (P.width[c] == 8) *reinterpret_cast<unsigned long long *>(row + P.off[c]) = ((const unsigned long long *)P.src[c])[r];
            else if (P.width[c] == 4) *reinterpret_cast<unsigned *>(row + P.off[c]) = ((const unsigned *)P.src[c])[r];
            else row[P.off[c]] = ((const unsigned char *)P.src[c])[r];
        }
    }
}

// build the co-located copy of column set `set` (sorted, distinct) on `ctx`'s stream; t->mu is held by the caller
static int colocate_locked(ph_ctx *ctx, ph_table *t, const std::vector<int> &set, bool explicit_request) {
    ph_table::colgroup g;
    g.cols = set;
    int total = 0;
    for (int c : set) {
        PH_REQUIRE(c >= 0 && c < (int)t->cols.size(), "ph_table_colocate: bad column %d", c);
        const int w = ph::type_width(t->cols[(size_t)c].type);
        if (w == 0 || t->cols[(size_t)c].validity) { ph::set_error("ph_table_colocate: column %d is not a fixed-width column without NULLs", c); return PH_EUNSUPPORTED; }
        g.width.push_back(w);
        total += w;
    }
    // widest first, so that every value is aligned to its width; rows of a power-of-two stride never straddle a 64-byte sector
    g.off.assign(set.size(), 0);
    int at = 0;
    for (int w : {8, 4, 1})
        for (size_t i = 0; i < set.size(); i++)
            if (g.width[i] == w) { g.off[i] = at; at += w; }
    int stride = 8;
    while (stride < total && stride < 64) stride *= 2;
    if (stride < total) stride = (int)ph::round_up(total, 16);
    g.stride = stride;
    const int64_t padded = ph::round_up(t->nrows > 0 ? t->nrows : 1, PH_ROW_PAD);
    g.bytes = padded * stride;
    // a copy the library builds on its own (the second sparse gather of a column set) stays inside the table's budget; one the host asked
    // for by name is the host's decision
    if (!explicit_request && t->colocate_bytes + g.bytes > t->colocate_budget) {
        ph::set_error("ph_table_colocate: %lld bytes beyond the table's co-location budget (%lld of %lld used)", (long long)g.bytes,
                      (long long)t->colocate_bytes, (long long)t->colocate_budget);
        return PH_ECAPACITY;
    }
    PH_HIP(hipSetDevice(ctx->device));
    PH_HIP(hipMalloc(&g.data, (size_t)g.bytes));
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipFree(g.data); ph::set_error("ph_table_colocate: hipEventCreate failed"); return PH_EHIP; }
    auto fail = [&]() { (void)hipEventDestroy(ev); (void)hipFree(g.data); ph::set_error("ph_table_colocate: launch failed"); return PH_EHIP; };
    if (hipMemsetAsync(g.data, 0, (size_t)g.bytes, ctx->stream) != hipSuccess) return fail();
    if (t->nrows > 0) {
        ColocateParams P{};
        P.ncols = (int)set.size();
        P.stride = stride;
        for (size_t i = 0; i < set.size(); i++) { P.src[i] = t->cols[(size_t)set[i]].data; P.off[i] = g.off[i]; P.width[i] = g.width[i]; }
        colocate_kernel<<<(int)std::min<int64_t>((t->nrows + 255) / 256, 256 * 16), 256, 0, ctx->stream>>>(P, t->nrows, (unsigned char *)g.data);
        if (hipGetLastError() != hipSuccess) return fail();
    }
    if (hipEventRecord(ev, ctx->stream) != hipSuccess) return fail();
    g.ready = ev;
    g.built_on = ctx->stream;
    t->colocate_bytes += g.bytes;
    t->groups.push_back(g);
    return PH_OK;
}

// Found (or built) and ORDERED: when this returns PH_OK every read of out->data queued on ctx's stream afterwards sees the finished copy —
// the builder's stream by stream order, any other stream through the group's event (a stream-side wait; once a consumer has seen the event
// complete nobody waits again). PH_EUNSUPPORTED: no such group and none built.
int ph::colocated_group_for(ph_ctx *ctx, ph_table *t, const std::vector<int> &tc, bool may_build, bool explicit_request, ph_table::colgroup *out) {
    std::lock_guard<std::mutex> lock(t->mu);
    auto find = [&]() -> ph_table::colgroup * {
        for (auto &g : t->groups) {
            bool all = true;
            for (int c : tc) all = all && std::find(g.cols.begin(), g.cols.end(), c) != g.cols.end();
            if (all) return &g;
        }
        return nullptr;
    };
    ph_table::colgroup *g = find();
    if (!g) {
        if (!may_build) return PH_EUNSUPPORTED;
        std::vector<int> set = tc;
        std::sort(set.begin(), set.end());
        if (std::adjacent_find(set.begin(), set.end()) != set.end()) return PH_EUNSUPPORTED;
        for (int c : set) if (c < 0 || c >= (int)t->cols.size() || t->cols[(size_t)c].validity) return PH_EUNSUPPORTED;
        if (!explicit_request) {
            int &seen = t->sparse_gathers[set];
            if (seen < 0 || ++seen < 2) return PH_EUNSUPPORTED;
            if (colocate_locked(ctx, t, set, false) != PH_OK) { seen = -1; return PH_EUNSUPPORTED; }   // (no room / no memory for the copy: not tried again)
        } else PH_CHECK(colocate_locked(ctx, t, set, true));
        g = find();
        if (!g) return PH_EUNSUPPORTED;
    }
    if (!g->complete && g->built_on != ctx->stream) {
        const hipError_t q = hipEventQuery((hipEvent_t)g->ready);
        if (q == hipSuccess) g->complete = true;
        else if (q == hipErrorNotReady) PH_HIP(hipStreamWaitEvent(ctx->stream, (hipEvent_t)g->ready, 0));
        else { ph::set_error("co-located group: %s", hipGetErrorString(q)); return PH_EHIP; }
    }
    *out = *g;
    return PH_OK;
}

extern "C" int ph_table_colocate(ph_table *t, int32_t ncols, const int32_t *cols) {
    PH_REQUIRE(t && t->ctx && cols && ncols >= 2 && ncols <= 8, "ph_table_colocate: 2..8 columns");
    std::vector<int> set(cols, cols + ncols);
    std::sort(set.begin(), set.end());
    PH_REQUIRE(std::adjacent_find(set.begin(), set.end()) == set.end(), "ph_table_colocate: a column is named twice");
    for (int c : set) PH_REQUIRE(c >= 0 && c < (int)t->cols.size(), "ph_table_colocate: bad column %d", c);
    for (int c : set)
        if (ph::type_width(t->cols[(size_t)c].type) == 0 || t->cols[(size_t)c].validity) { ph::set_error("ph_table_colocate: column %d is not a fixed-width column without NULLs", c); return PH_EUNSUPPORTED; }
    ph_table::colgroup g;
    return ph::colocated_group_for(t->ctx, t, set, true, true, &g);   // on the table's own ctx (load time)
}

extern "C" int32_t ph_table_colocated(const ph_table *t, int32_t ncols, const int32_t *cols) {
    if (!t || !cols || ncols < 1) return 0;
    std::lock_guard<std::mutex> lock(const_cast<ph_table *>(t)->mu);
    for (auto &g : t->groups) {
        bool all = true;
        for (int i = 0; i < ncols && all; i++) all = std::find(g.cols.begin(), g.cols.end(), (int)cols[i]) != g.cols.end();
        if (all) return 1;
    }
    return 0;
}

extern "C" int ph_table_set_replicated(ph_table *t, int32_t on) {
    PH_REQUIRE(t != nullptr, "ph_table_set_replicated: table is NULL");
    t->replicated = on != 0;
    return PH_OK;
}

extern "C" int ph_table_set_colocate_budget(ph_table *t, int64_t bytes) {
    PH_REQUIRE(t && bytes >= 0, "ph_table_set_colocate_budget: bad arguments");
    std::lock_guard<std::mutex> lock(t->mu);
    t->colocate_budget = bytes;
    return PH_OK;
}

extern "C" int64_t ph_table_colocate_bytes(const ph_table *t) {
    if (!t) return -1;
    std::lock_guard<std::mutex> lock(const_cast<ph_table *>(t)->mu);
    return t->colocate_bytes;
}

extern "C" int ph_table_declare_unique(ph_table *t, int32_t ncols, const int32_t *cols) {
    PH_REQUIRE(t && cols && ncols >= 1 && ncols <= 4, "ph_table_declare_unique: 1..4 columns");
    std::vector<int32_t> u(cols, cols + ncols);
    for (int32_t c : u) PH_REQUIRE(c >= 0 && c < (int32_t)t->cols.size(), "ph_table_declare_unique: bad column %d", c);
    std::sort(u.begin(), u.end());
    t->unique_keys.push_back(u);
    return PH_OK;
}

extern "C" void ph_table_free(ph_table *t) {
    if (!t) return;
    if (t->ctx) {
        (void)hipSetDevice(t->ctx->device);
        (void)hipStreamSynchronize(t->ctx->stream);
    }
    ph::unregister_table(t);
    for (auto &g : t->groups) { if (g.ready) (void)hipEventDestroy((hipEvent_t)g.ready); if (g.data) (void)hipFree(g.data); }
    for (auto &c : t->cols) {
        if (c.data) (void)hipFree(c.data);
        if (c.validity) (void)hipFree(c.validity);
        if (c.aux) (void)hipFree(c.aux);
    }
    delete t;
}
