// Internal declarations shared by the libplanhip.so translation units (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "planhip.h"

namespace ph {

void set_error(const char *fmt, ...);

#define PH_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ph::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                     \
            return PH_EHIP;                                                              \
        }                                                                                \
    } while (0)

#define PH_CHECK(expr)                 \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != PH_OK) return rc_;   \
    } while (0)

#define PH_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) {                  \
            ph::set_error(__VA_ARGS__); \
            return PH_EINVAL;           \
        }                               \
    } while (0)

constexpr int WAVE = 64;
constexpr int CU_COUNT = 256;

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// element width of a fixed-width column type (0 for PH_STR)
inline int type_width(int32_t t) {
    switch (t) {
    case PH_I32: case PH_DATE: case PH_F32: return 4;
    case PH_I64: case PH_DEC64: case PH_F64: return 8;
    case PH_CODE8: return 1;
    default: return 0;
    }
}

}  // namespace ph

namespace ph {
// what a scan needs to publish its total itself (ops.h exclusive_scan_i32): the mapped mailbox, its sequence word and the number to store there
struct ScanPublish {
    unsigned long long *mbox = nullptr, *flag = nullptr;
    unsigned long long seq = 0;
    const int *deferred = nullptr;   // the ctx's deferred-error words ride along (as with publish_kernel) when some are pending
    int *mbox_deferred = nullptr;
};
}  // namespace ph

struct ph_ctx {
    int device = 0;
    int cu_count = ph::CU_COUNT;  // multiProcessorCount of the device
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // reusable device scratch (block partials, scan buffers, counters)
    void *scratch = nullptr;
    int64_t scratch_bytes = 0;
    // pinned host staging for uploads / small downloads
    void *pinned = nullptr;
    int64_t pinned_bytes = 0;
    int ensure_scratch(int64_t bytes);
    int ensure_pinned(int64_t bytes);
    // Synchronous device -> host copy through pinned staging. Copying straight into pageable
    // caller memory makes the runtime pin/unpin those pages on every call (milliseconds when the
    // caller's buffers come and go, as numpy / Go-heap buffers do).
    void *mailbox = nullptr;  // 64 KiB pinned, for counts / flags / small results
    // Small results reach the host WITHOUT a copy command and a stream synchronisation: the mailbox is mapped
    // into the device's address space (coherent host memory), a one-workgroup kernel stores the bytes there and
    // then a sequence number, and the host polls that word (publish(); ctx.hip).
    void *mailbox_dev = nullptr;          // device address of the mailbox
    unsigned long long publish_seq = 0;   // last sequence number handed to a publish kernel
    int64_t *count_slots_dev = nullptr;   // device address of count_slots
    unsigned long long count_seq_issued = 0;
    int publish(const void *dev, int64_t bytes, bool with_deferred);   // bytes <= 64 KiB -> mailbox; waits for it
    // A kernel that publishes its own result (ScanTail): arm_publish hands out the mailbox, the sequence word and the next number BEFORE the launch;
    // collect_armed waits for that number and copies the bytes out — unless another publish has used the mailbox since (returns 1: download as usual)
    // or the arming was refused (seq 0: PH_NO_PUBLISH, deferred words pending).
    int arm_publish(int64_t bytes, unsigned long long **mbox_dev, unsigned long long **flag_dev, unsigned long long *seq, bool deferred_ok = false);
    int collect_armed(void *host, int64_t bytes, unsigned long long seq, bool deferred_ok = false);
    int ensure_mailbox();
    // A row count that travels with the scan that produces it: arm_count before the scan is launched (pub is left empty when the count has to take
    // the ordinary way: deferred counts, PH_EARLY_COUNT=0), count_back instead of download_count behind the kernels that consume the offsets.
    int arm_count(ph::ScanPublish *pub);
    int count_back(const ph::ScanPublish &pub, int64_t *host, const void *total_dev, int64_t cap, const char *what);
    unsigned *scan_done_dev = nullptr;    // the ticket counter of fused scans' last-workgroup tails (zero between launches)
    // single-pass scan (ops_select.hip): tile states + ticket counter, reused across calls by epoch
    void *scan_state = nullptr;
    int64_t scan_tiles = 0;
    unsigned scan_ticket_base = 0;
    unsigned long long scan_epoch = 0;
    // with_deferred: also fetch (and report) a pending deferred error in the same synchronisation — unless the ctx
    // HOLDS deferred errors (ph_ctx_set_deferred_errors(ctx, 2)): then only ph_ctx_check_deferred reports them
    int download(void *host, const void *dev, int64_t bytes, bool with_deferred = true);
    // a read-back that never reports a deferred error (the multi-GPU collectives' count downloads: a rank that left a
    // collective sequence early with ITS deferred error would leave its peers blocked in the next collective)
    int download_plain(void *host, const void *dev, int64_t bytes) { return download(host, dev, bytes, false); }
    // Deferred errors (ph_ctx_set_deferred_errors): device words that kernels of calls which would
    // otherwise read a flag back (ph_expr_eval's overflow flag, ph_join_lookup_strict's miss / multi-
    // match counts) OR / add into; the next download() of this ctx fetches them in the same stream
    // synchronisation and fails with the deferred error. [0] overflow, [1] lookup misses, [2] lookup
    // multi-matches.
    // Asynchronous counts (ph_ctx_set_async_counts): a call that returns a row count to the host
    // (ph_filter_select, ph_join_probe_inner*) enqueues the 8-byte copy into a pinned slot of its own and
    // returns at once with *n_out = -1; ph_ctx_wait_counts waits for the LAST such copy only (an event,
    // not the stream) and fills the counts in. Work queued between the call and the wait runs while the
    // host wakes up and prepares the launches that depend on the count.
    struct PendingCount { int64_t *host; int64_t cap; const char *what; };
    std::vector<PendingCount> pending_counts;
    int64_t *count_slots = nullptr;   // pinned, PH_MAX_PENDING_COUNTS entries
    void *count_event = nullptr;      // hipEvent_t
    bool async_counts = false;
    int download_count(int64_t *host, const void *dev, int64_t cap, const char *what);
    int wait_counts();
    int *deferred_dev = nullptr;
    bool defer_errors = false, deferred_pending = false, defer_hold = false;
    int deferred_words(int **out);   // allocates (zeroed) on first use
    int finish_deferred();           // after a sync that also copied the words into the mailbox tail
    // Stream-ordered device memory pool: freed blocks are reused by later allocations of the
    // same rounded size without hipFree/hipMalloc (both synchronise the device). Safe because
    // every kernel and copy of a ctx runs on its one stream.
    std::multimap<int64_t, void *> pool_free_blocks;
    std::map<void *, int64_t> pool_sizes;
    int pool_alloc(int64_t bytes, void **out);
    void pool_release(void *p);
    void pool_destroy();
};

struct ph_table {
    ph_ctx *ctx = nullptr;
    int64_t nrows = 0;
    struct column {
        int32_t type = 0, scale = 0;
        void *data = nullptr;      // device, padded
        uint8_t *validity = nullptr;
        void *aux = nullptr;       // PH_STR bytes (device)
        int64_t aux_bytes = 0;
        std::vector<std::string> dict; // PH_CODE8 dictionary (host), from aux at load
        int64_t min = 0, max = 0;
        bool has_range = false;
        // order statistics of integer columns, one pass at load (like min / max): the values are non-decreasing in
        // storage order (a clustering column: lineitem by l_orderkey) / strictly ascending (a primary key in key order)
        bool ascending = false, strict = false;
        // an ascending column made of runs of ONE length over consecutive values (row i holds min + i / run_len: partsupp by ps_partkey, four
        // suppliers per part): the rows of a key are found by arithmetic. 0 = not that shape. Verified on the device at load, like the order.
        int32_t run_len = 0;
    };
    std::vector<column> cols;
    // column sets the catalog declares unique (PRIMARY KEY): ph_table_declare_unique
    std::vector<std::vector<int32_t>> unique_keys;
    // Resident tables are SHARED: created on one ctx, read by plans on any other ctx of the device, from other threads (the Go shim keeps
    // the tables on a process-wide context and gives every executor its own; SURVEY.md §8(b) "threading": only the table cache is shared,
    // mutex-guarded). Columns, statistics and declared keys are immutable after ph_table_create / ph_table_declare_unique (load time);
    // what a QUERY may change — the co-located copies and the count of sparse gathers that triggers one — sits behind `mu`.
    std::mutex mu;
    // co-located copies of column sets (ph_table_colocate): row r of the group = the set's values of row r side by side, so a
    // sparse gather of several columns reads ONE sector per row id instead of one per column
    struct colgroup {
        std::vector<int> cols, off, width;   // table column, byte offset in the group's row, width
        int stride = 0;                      // bytes per row (a power of two up to 64, else a multiple of 16)
        void *data = nullptr;
        int64_t bytes = 0;
        // ordering against consumers: the copy is filled by a kernel on `built_on`; `ready` (hipEvent_t) is recorded behind it. A
        // consumer on another stream waits for the event on ITS stream (stream-side) until some consumer has seen it complete.
        void *ready = nullptr;
        hipStream_t built_on = nullptr;
        bool complete = false;
    };
    std::deque<colgroup> groups;                      // (a deque: elements stay where they are when one is added)
    std::map<std::vector<int>, int> sparse_gathers;   // column set -> sparse multi-column gathers seen (the second one builds the group)
    bool replicated = false;                          // multi-rank plans (ph_plan_set_comm): every rank holds ALL rows of this table (ph_table_set_replicated)
    int64_t colocate_budget = 4ll << 30;              // bytes the library may spend on copies it builds ON ITS OWN (ph_table_set_colocate_budget)
    int64_t colocate_bytes = 0;                       // bytes held by all copies
};

namespace ph {
// process-wide registry: device column base pointer -> (resident table, column). Lets ph_gather_multi recognise a table's columns in the
// views it is given whatever ctx it is called on (the registry of the CALLING ctx, the first form, never saw tables of another ctx).
void register_table(ph_table *t);
void unregister_table(ph_table *t);
bool lookup_table_col(const void *data, ph_table **t, int *col);
// the co-located group covering `tc` (table columns) for a consumer on `ctx`: found, or built on ctx's stream when `may_build`; ordered
// against ctx's stream before it returns. Returns a COPY of the group's layout (the table may grow another group meanwhile).
int colocated_group_for(ph_ctx *ctx, ph_table *t, const std::vector<int> &tc, bool may_build, bool explicit_request, ph_table::colgroup *out);
}  // namespace ph

// rows a column allocation is padded to, so vector loads never leave the allocation
constexpr int64_t PH_ROW_PAD = 8192;
