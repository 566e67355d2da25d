// General hash aggregate: ph_agg_create / sink / finalize.
//
// Replaces GroupedAggrHashTable.AddChunk -> FindOrCreateGroups -> UpdateStates and FinalizeStates
// (reference pkg/compute/aggregate_hash.go:136-391, aggregate_exec.go:456-475,
// function_aggr.go:420-1365) for arbitrary group cardinality (Q3: ~114k groups at SF10, Q9: 175).
//
// Device layout (all in HBM, sized for 288 GB rather than for cache):
//   slots[cap]           int32 group id, EMPTY/LOCKED sentinels; open addressing, linear probing,
//                        cap a power of two — the reference's scheme minus the 16-bit salt (the
//                        full key compare that the salt avoids is one 8-byte load per key here)
//   gkeys[g][nkeys]      raw 64-bit key words, gnull[g] NULL bits (NULL keys form their own group,
//                        as Match's NULL = NULL rule for group columns does, util_match.go:25-301)
//   sum_lo/sum_hi[g][a]  128-bit two's-complement sums (Hugeint for INTEGER input, exact unscaled
//                        DECIMAL sums), cnt[g][a] non-NULL inputs, first_row[g] for first-seen order
// Inter-workgroup protocol: every access to slots/gkeys goes through agent-scope atomics (the L1 of
// a CU is never refreshed by other CUs' stores and the per-XCD L2s are not coherent for plain
// accesses); a creator stores the key words, fences, then publishes the group id.
// State updates are fire-and-forget HBM atomics; the 128-bit add is a returning add on the low
// word plus a carry/sign add on the high word, which commutes, so any interleaving gives the exact sum.
#include <algorithm>

#include "common.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

constexpr int AGG_MAX_KEYS = 4;
constexpr int AGG_MAX_AGGS = 16;
constexpr int SLOT_EMPTY = -1;
constexpr int SLOT_LOCKED = -2;

struct AggCol {
    int type;
    const void *data;
    const uint8_t *validity;
};

struct AggSinkParams {
    int nkeys, naggs, nargs;
    AggCol key[AGG_MAX_KEYS];
    AggCol arg[AGG_MAX_AGGS];
    int agg_kind[AGG_MAX_AGGS];
    int agg_arg[AGG_MAX_AGGS];
    const int32_t *sel;
    int64_t n;
    int positional;
    int64_t row_base;
    int32_t *slots;
    uint64_t mask;
    unsigned long long *gkeys;
    unsigned *gnull;
    unsigned long long *sum_lo;
    long long *sum_hi;
    unsigned long long *cnt;
    long long *first_row;
    int *ngroups;
    int64_t gcap;
    int *error_flag;
    int lds_slots;  // power of two; per-workgroup staging table entries
    const int *skip_from;  // batches with index >= *skip_from were refused by the growth guard
    int batch_index;
};

__device__ __forceinline__ unsigned long long load_key(const AggCol &c, int64_t r) {
    switch (c.type) {
    case PH_I32: case PH_DATE: return (unsigned long long)(long long)((const int32_t *)c.data)[r];
    case PH_CODE8: return ((const uint8_t *)c.data)[r];
    default: return (unsigned long long)((const int64_t *)c.data)[r];
    }
}

__device__ __forceinline__ uint64_t keys_hash(const unsigned long long *k, unsigned nullmask, int nkeys) {
    uint64_t h = mix64((uint64_t)nullmask + 0x9e3779b97f4a7c15ULL);
    for (int c = 0; c < nkeys; c++) h = mix64(h ^ k[c]);
    return h;
}

__device__ __forceinline__ void add128(unsigned long long *lo, long long *hi, long long v) {
    unsigned long long old = atomicAdd(lo, (unsigned long long)v);
    unsigned long long nw = old + (unsigned long long)v;
    long long delta = (nw < old ? 1 : 0) + (v < 0 ? -1 : 0);
    if (delta != 0) atomicAdd((unsigned long long *)hi, (unsigned long long)delta);
}

// Growth guard, one tiny launch in front of every batch: a batch may only run when the table can
// take all of its rows as new groups (the reference's Resize rule). The first batch that cannot is
// recorded and it and all later batches return immediately; the host then grows the table and
// re-enqueues from there. Batches are enqueued back to back without a host round trip per batch.
__global__ void agg_guard_kernel(const int *__restrict__ ngroups, long long gcap, long long m, int batch,
                                 int *__restrict__ skip_from) {
    if (batch >= *skip_from) return;
    if (gcap - (long long)*ngroups <= m) atomicMin(skip_from, batch);
}

__global__ __launch_bounds__(256) void agg_sink_kernel(AggSinkParams P) {
    if (P.batch_index >= __hip_atomic_load(P.skip_from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char agg_lds[];
    // [first S x i64][sum S*na x u64][gid S x i32][cnt S*na x u32]
    long long *l_first = reinterpret_cast<long long *>(agg_lds);
    unsigned long long *l_sum = reinterpret_cast<unsigned long long *>(l_first + P.lds_slots);
    int *l_gid = reinterpret_cast<int *>(l_sum + (size_t)P.lds_slots * P.naggs);
    unsigned *l_cnt = reinterpret_cast<unsigned *>(l_gid + P.lds_slots);
    for (int e = threadIdx.x; e < P.lds_slots; e += 256) {
        l_gid[e] = -1;
        l_first[e] = INT64_MAX;
        for (int a = 0; a < P.naggs; a++) {
            int kind = P.agg_kind[a];
            l_sum[e * P.naggs + a] = kind == PH_A_MIN ? (unsigned long long)INT64_MAX
                                     : kind == PH_A_MAX ? (unsigned long long)INT64_MIN : 0ull;
            l_cnt[e * P.naggs + a] = 0;
        }
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * 256) {
        int64_t r = P.sel ? P.sel[i] : i;
        unsigned long long k[AGG_MAX_KEYS];
        unsigned nullmask = 0;
#pragma unroll
        for (int c = 0; c < AGG_MAX_KEYS; c++) {
            k[c] = 0;
            if (c < P.nkeys) {
                if (bit_valid(P.key[c].validity, r)) k[c] = load_key(P.key[c], r);
                else nullmask |= 1u << c;
            }
        }
        uint64_t slot = keys_hash(k, nullmask, P.nkeys) & P.mask;
        int gid = -1;
        // find or create (FindOrCreateGroups :272-388). No lane ever waits inside a branch, so
        // lanes of one wave racing for the same new key cannot deadlock: the winner publishes in
        // the same iteration it locked the slot; the others see the id on a later iteration.
        for (int guard = 0; gid < 0; guard++) {
            int g = __hip_atomic_load(&P.slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == SLOT_EMPTY) {
                int old = atomicCAS(&P.slots[slot], SLOT_EMPTY, SLOT_LOCKED);
                if (old == SLOT_EMPTY) {
                    int ng = atomicAdd(P.ngroups, 1);
                    if (ng >= P.gcap) {  // cannot happen: the host grows before a batch (see sink)
                        atomicOr(P.error_flag, 1);
                        ng = 0;
                    }
                    for (int c = 0; c < P.nkeys; c++)
                        __hip_atomic_store(&P.gkeys[(int64_t)ng * P.nkeys + c], k[c], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&P.gnull[ng], nullmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __threadfence();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&P.slots[slot], ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gid = ng;
                }
                // lost the race: look at the slot again
            } else if (g == SLOT_LOCKED) {
                if (guard > (1 << 22)) { atomicOr(P.error_flag, 2); break; }  // bounded spin
            } else {
                bool eq = __hip_atomic_load(&P.gnull[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nullmask;
                for (int c = 0; eq && c < P.nkeys; c++)
                    eq = __hip_atomic_load(&P.gkeys[(int64_t)g * P.nkeys + c], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT) == k[c];
                if (eq) gid = g;
                else slot = (slot + 1) & P.mask;  // linear probing (:376-384)
            }
        }
        if (gid < 0) continue;
        // ---- LDS staging (per workgroup, direct mapped by group id): rows of a group that owns
        // its LDS entry accumulate with ds atomics and reach HBM once per workgroup, so a hot
        // group (Q9: 175 groups, Q1-like inputs: 4) no longer serialises every row on one HBM
        // atomic. Groups that lose the entry to another id, and values too large for a bounded
        // int64 partial, update HBM directly.
        const int e = gid & (P.lds_slots - 1);
        bool staged = false;
        {
            int cur = l_gid[e];
            if (cur == gid) staged = true;
            else if (cur == -1) {
                int old = atomicCAS(&l_gid[e], -1, gid);
                staged = old == -1 || old == gid;
            }
        }
        long long frow = (long long)(P.row_base + (P.sel ? r : i));  // row id (ascending with i)
        // first-seen row: almost every row is later than the recorded one, so test with a load
        // (L2-served, agent scope) and only issue the HBM atomic when it would lower the minimum
        if (staged) { if (frow < l_first[e]) atomicMin(&l_first[e], frow); }
        else if (frow < __hip_atomic_load(&P.first_row[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(&P.first_row[gid], frow);
        // UpdateStates (aggregate_exec.go:456-475): NULL inputs are skipped (IgnoreNull)
        for (int a = 0; a < P.naggs; a++) {
            int64_t st = (int64_t)gid * P.naggs + a;
            int ls = e * P.naggs + a;
            int kind = P.agg_kind[a];
            if (kind == PH_A_COUNT_STAR) {
                if (staged) atomicAdd(&l_cnt[ls], 1u);
                else atomicAdd(&P.cnt[st], 1ull);
                continue;
            }
            const AggCol &c = P.arg[P.agg_arg[a]];
            int64_t ar = P.positional ? i : r;
            if (!bit_valid(c.validity, ar)) continue;
            long long v = c.type == PH_I32 ? (long long)((const int32_t *)c.data)[ar]
                                           : ((const int64_t *)c.data)[ar];
            if (kind == PH_A_SUM || kind == PH_A_AVG) {
                // a workgroup sees at most 2^21 rows per launch: |v| < 2^40 keeps the partial in int64
                bool small = v > -(1ll << 40) && v < (1ll << 40);
                if (staged && small) {
                    atomicAdd(&l_sum[ls], (unsigned long long)v);
                    atomicAdd(&l_cnt[ls], 1u);
                } else {
                    atomicAdd(&P.cnt[st], 1ull);
                    add128(&P.sum_lo[st], &P.sum_hi[st], v);
                }
            } else if (kind == PH_A_COUNT) {
                if (staged) atomicAdd(&l_cnt[ls], 1u);
                else atomicAdd(&P.cnt[st], 1ull);
            } else if (kind == PH_A_MIN) {
                if (staged) { atomicMin((long long *)&l_sum[ls], v); atomicAdd(&l_cnt[ls], 1u); }
                else { atomicAdd(&P.cnt[st], 1ull); atomicMin((long long *)&P.sum_lo[st], v); }
            } else if (kind == PH_A_MAX) {
                if (staged) { atomicMax((long long *)&l_sum[ls], v); atomicAdd(&l_cnt[ls], 1u); }
                else { atomicAdd(&P.cnt[st], 1ull); atomicMax((long long *)&P.sum_lo[st], v); }
            }
        }
    }
    // ---- flush the staging table: one HBM update per (group, aggregate) per workgroup
    __syncthreads();
    for (int e = threadIdx.x; e < P.lds_slots; e += 256) {
        int gid = l_gid[e];
        if (gid < 0) continue;
        if (l_first[e] != INT64_MAX) atomicMin(&P.first_row[gid], l_first[e]);
        for (int a = 0; a < P.naggs; a++) {
            unsigned n = l_cnt[e * P.naggs + a];
            if (n == 0) continue;
            int64_t st = (int64_t)gid * P.naggs + a;
            int kind = P.agg_kind[a];
            atomicAdd(&P.cnt[st], (unsigned long long)n);
            long long v = (long long)l_sum[e * P.naggs + a];
            if (kind == PH_A_SUM || kind == PH_A_AVG) add128(&P.sum_lo[st], &P.sum_hi[st], v);
            else if (kind == PH_A_MIN) atomicMin((long long *)&P.sum_lo[st], v);
            else if (kind == PH_A_MAX) atomicMax((long long *)&P.sum_lo[st], v);
        }
    }
}

__global__ __launch_bounds__(256) void agg_init_kernel(unsigned long long *sum_lo, long long *sum_hi,
                                                       unsigned long long *cnt, long long *first_row,
                                                       int64_t g_begin, int64_t g_end, int naggs,
                                                       const int *kinds /* device copy */) {
    int64_t total = (g_end - g_begin) * naggs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t st = g_begin * naggs + i;
        int kind = kinds[i % naggs];
        sum_lo[st] = kind == PH_A_MIN ? (unsigned long long)INT64_MAX
                     : kind == PH_A_MAX ? (unsigned long long)INT64_MIN : 0ull;
        sum_hi[st] = 0;
        cnt[st] = 0;
    }
    for (int64_t g = g_begin + (int64_t)blockIdx.x * 256 + threadIdx.x; g < g_end; g += (int64_t)gridDim.x * 256)
        first_row[g] = INT64_MAX;
}

// re-insert groups [0, ng) into a fresh slot table (Resize, aggregate_hash.go:440-513)
__global__ __launch_bounds__(256) void agg_rehash_kernel(int32_t *slots, uint64_t mask,
                                                         const unsigned long long *gkeys, const unsigned *gnull,
                                                         int nkeys, int ng) {
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        unsigned long long k[AGG_MAX_KEYS] = {0, 0, 0, 0};
        for (int c = 0; c < nkeys; c++) k[c] = gkeys[(int64_t)g * nkeys + c];
        uint64_t slot = keys_hash(k, gnull[g], nkeys) & mask;
        while (atomicCAS(&slots[slot], SLOT_EMPTY, g) != SLOT_EMPTY) slot = (slot + 1) & mask;
    }
}

// ---- top-N pre-selection: one workgroup, 8 radix passes over the 64-bit order keys, then a
// compaction of the groups on the good side of the k-th value.
__device__ __forceinline__ unsigned long long order_key(long long v, int descending) {
    unsigned long long u = (unsigned long long)v ^ 0x8000000000000000ull;  // signed -> unsigned order
    return descending ? ~u : u;                                            // smaller key = better
}

// state[0] = prefix, state[1] = remaining k, hist[256] in global memory; no host round trips
__global__ __launch_bounds__(256) void topk_check_kernel(const unsigned long long *__restrict__ sum_lo,
                                                         const long long *__restrict__ sum_hi, int naggs, int a, int ng,
                                                         int *__restrict__ flags) {
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        long long lo = (long long)sum_lo[(int64_t)g * naggs + a];
        if (sum_hi[(int64_t)g * naggs + a] != (lo >> 63)) atomicOr(flags, 1);
    }
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const unsigned long long *__restrict__ sum_lo, int naggs, int a,
                                                        int ng, int descending, int pass,
                                                        const unsigned long long *__restrict__ state,
                                                        unsigned *__restrict__ hist) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long prefix = state[0];
    unsigned long long mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        unsigned long long key = order_key((long long)sum_lo[(int64_t)g * naggs + a], descending);
        if ((key & mask) == (prefix & mask)) atomicAdd(&lh[(key >> (8 * pass)) & 0xff], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void topk_resolve_kernel(unsigned long long *__restrict__ state, unsigned *__restrict__ hist, int pass) {
    if (threadIdx.x == 0) {
        long long rem = (long long)state[1];
        int b = 0;
        for (; b < 256; b++) {
            if ((long long)hist[b] >= rem) break;
            rem -= hist[b];
        }
        if (b == 256) b = 255;  // k > ng: everything qualifies
        state[0] |= (unsigned long long)b << (8 * pass);
        state[1] = (unsigned long long)rem;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
}

__global__ __launch_bounds__(256) void topk_collect_kernel(const unsigned long long *__restrict__ sum_lo, int naggs, int a,
                                                           int ng, int descending, long long k,
                                                           const unsigned long long *__restrict__ state,
                                                           int *__restrict__ out_ids, int *__restrict__ out_count, int cap) {
    unsigned long long kth = state[0];
    bool all = k >= ng;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        unsigned long long key = order_key((long long)sum_lo[(int64_t)g * naggs + a], descending);
        if (all || key <= kth) {
            int pos = atomicAdd(out_count, 1);
            if (pos < cap) out_ids[pos] = g;
        }
    }
}

// pack the selected groups' records contiguously so they come back in six copies
__global__ __launch_bounds__(256) void agg_pack_kernel(const int *__restrict__ ids, int n, int nkeys, int naggs,
                                                       const long long *first_row, const unsigned long long *gkeys,
                                                       const unsigned *gnull, const unsigned long long *sum_lo,
                                                       const long long *sum_hi, const unsigned long long *cnt,
                                                       long long *o_first, unsigned long long *o_keys, unsigned *o_null,
                                                       unsigned long long *o_lo, long long *o_hi, unsigned long long *o_cnt) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int g = ids[i];
        o_first[i] = first_row[g];
        o_null[i] = gnull[g];
        for (int c = 0; c < nkeys; c++) o_keys[(int64_t)i * nkeys + c] = gkeys[(int64_t)g * nkeys + c];
        for (int a = 0; a < naggs; a++) {
            o_lo[(int64_t)i * naggs + a] = sum_lo[(int64_t)g * naggs + a];
            o_hi[(int64_t)i * naggs + a] = sum_hi[(int64_t)g * naggs + a];
            o_cnt[(int64_t)i * naggs + a] = cnt[(int64_t)g * naggs + a];
        }
    }
}

}  // namespace ph

struct ph_agg {
    ph_ctx *ctx = nullptr;
    int nkeys = 0, naggs = 0;
    int key_types[ph::AGG_MAX_KEYS] = {0, 0, 0, 0};
    ph_aggspec aggs[ph::AGG_MAX_AGGS] = {};
    int64_t cap = 0, gcap = 0;
    int32_t *slots = nullptr;
    unsigned long long *gkeys = nullptr;
    unsigned *gnull = nullptr;
    unsigned long long *sum_lo = nullptr;
    long long *sum_hi = nullptr;
    unsigned long long *cnt = nullptr;
    long long *first_row = nullptr;
    int *counters = nullptr;  // [0] ngroups, [1] error flag, [2] first refused batch (growth guard)
    int *kinds_dev = nullptr;
    int64_t rows_sunk = 0;
};

namespace {

constexpr int64_t AGG_BATCH = 1 << 21;

int64_t next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

int agg_free_arrays(ph_agg *a) {
    void *ptrs[] = {a->slots, a->gkeys, a->gnull, a->sum_lo, a->sum_hi, a->cnt, a->first_row};
    for (void *p : ptrs) if (p) a->ctx->pool_release(p);
    a->slots = nullptr; a->gkeys = nullptr; a->gnull = nullptr; a->sum_lo = nullptr;
    a->sum_hi = nullptr; a->cnt = nullptr; a->first_row = nullptr;
    return PH_OK;
}

// (re)allocate for `cap` slots, carrying over `ng` existing groups
int agg_resize(ph_agg *a, int64_t cap, int ng) {
    ph_ctx *ctx = a->ctx;
    int64_t gcap = cap / 2;
    int32_t *slots = nullptr;
    unsigned long long *gkeys = nullptr, *sum_lo = nullptr, *cnt = nullptr;
    unsigned *gnull = nullptr;
    long long *sum_hi = nullptr, *first_row = nullptr;
    size_t na = (size_t)std::max(a->naggs, 1);
    PH_CHECK(ctx->pool_alloc(cap * 4, (void **)&slots));
    PH_CHECK(ctx->pool_alloc(gcap * a->nkeys * 8, (void **)&gkeys));
    PH_CHECK(ctx->pool_alloc(gcap * 4, (void **)&gnull));
    PH_CHECK(ctx->pool_alloc(gcap * (int64_t)na * 8, (void **)&sum_lo));
    PH_CHECK(ctx->pool_alloc(gcap * (int64_t)na * 8, (void **)&sum_hi));
    PH_CHECK(ctx->pool_alloc(gcap * (int64_t)na * 8, (void **)&cnt));
    PH_CHECK(ctx->pool_alloc(gcap * 8, (void **)&first_row));
    PH_HIP(hipMemsetAsync(slots, 0xff, (size_t)cap * 4, ctx->stream));
    if (ng > 0) {
        PH_HIP(hipMemcpyAsync(gkeys, a->gkeys, (size_t)ng * a->nkeys * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(gnull, a->gnull, (size_t)ng * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(sum_lo, a->sum_lo, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(sum_hi, a->sum_hi, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(cnt, a->cnt, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(first_row, a->first_row, (size_t)ng * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    int grid = (int)std::min<int64_t>(((gcap - ng) * (int64_t)na + 255) / 256 + 1, 2048);
    ph::agg_init_kernel<<<grid, 256, 0, ctx->stream>>>(sum_lo, sum_hi, cnt, first_row, ng, gcap, a->naggs, a->kinds_dev);
    PH_HIP(hipGetLastError());
    if (ng > 0) {
        ph::agg_rehash_kernel<<<std::min((ng + 255) / 256, 2048), 256, 0, ctx->stream>>>(slots, (uint64_t)cap - 1, gkeys, gnull, a->nkeys, ng);
        PH_HIP(hipGetLastError());
    }
    agg_free_arrays(a);  // stream-ordered: the copies above are already queued
    a->slots = slots; a->gkeys = gkeys; a->gnull = gnull; a->sum_lo = sum_lo; a->sum_hi = sum_hi;
    a->cnt = cnt; a->first_row = first_row;
    a->cap = cap;
    a->gcap = gcap;
    return PH_OK;
}

}  // namespace

extern "C" void ph_agg_free(ph_agg *a) {
    if (!a) return;
    agg_free_arrays(a);
    if (a->counters) a->ctx->pool_release(a->counters);
    if (a->kinds_dev) a->ctx->pool_release(a->kinds_dev);
    delete a;
}

extern "C" int ph_agg_create(ph_ctx *ctx, int32_t nkeys, const int32_t *key_types, int32_t naggs,
                             const ph_aggspec *aggs, int64_t expected_groups, ph_agg **out) {
    PH_REQUIRE(ctx && out && key_types && nkeys >= 1 && nkeys <= ph::AGG_MAX_KEYS && naggs >= 0 &&
                   naggs <= ph::AGG_MAX_AGGS && (naggs == 0 || aggs),
               "ph_agg_create: bad arguments (1..%d keys, 0..%d aggregates)", ph::AGG_MAX_KEYS, ph::AGG_MAX_AGGS);
    for (int c = 0; c < nkeys; c++) {
        int t = key_types[c];
        if (t != PH_I32 && t != PH_I64 && t != PH_DATE && t != PH_DEC64 && t != PH_CODE8) {
            ph::set_error("ph_agg_create: key type %d cannot be a device group key", t);
            return PH_EUNSUPPORTED;
        }
    }
    ph_agg *a = new ph_agg();
    a->ctx = ctx;
    a->nkeys = nkeys;
    a->naggs = naggs;
    int kinds[ph::AGG_MAX_AGGS] = {};
    for (int c = 0; c < nkeys; c++) a->key_types[c] = key_types[c];
    for (int i = 0; i < naggs; i++) { a->aggs[i] = aggs[i]; kinds[i] = aggs[i].kind; }
    int rc = PH_OK;
    if (ctx->pool_alloc(16, (void **)&a->counters) != PH_OK || hipMemsetAsync(a->counters, 0, 16, ctx->stream) != hipSuccess ||
        ctx->pool_alloc(sizeof kinds, (void **)&a->kinds_dev) != PH_OK ||
        hipMemcpyAsync(a->kinds_dev, kinds, sizeof kinds, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ph::set_error("ph_agg_create: device allocation failed");
        rc = PH_EHIP;
    }
    // initial capacity: the reference starts at 2*2048 entries (aggregate_exec.go:332-339)
    if (rc == PH_OK) rc = agg_resize(a, next_pow2(std::max<int64_t>(4096, 2 * expected_groups)), 0);
    if (rc != PH_OK) { ph_agg_free(a); return rc; }
    *out = a;
    return PH_OK;
}

extern "C" int ph_agg_group_count(ph_agg *a, int64_t *ngroups) {
    PH_REQUIRE(a && ngroups, "ph_agg_group_count: bad arguments");
    int c[2] = {0, 0};
    PH_CHECK(a->ctx->download(c, a->counters, 8));
    if (c[1]) { ph::set_error("ph_agg: device table error flag %d", c[1]); return PH_EHIP; }
    *ngroups = c[0];
    return PH_OK;
}

extern "C" int ph_agg_sink(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs,
                           const int32_t *sel, int64_t n, int32_t positional, int64_t row_base) {
    PH_REQUIRE(a && keys && n >= 0 && nargs >= 0 && nargs <= ph::AGG_MAX_AGGS && (nargs == 0 || args),
               "ph_agg_sink: bad arguments");
    ph::AggSinkParams P{};
    P.nkeys = a->nkeys;
    P.naggs = a->naggs;
    P.nargs = nargs;
    for (int c = 0; c < a->nkeys; c++) {
        PH_REQUIRE(keys[c].type == a->key_types[c], "ph_agg_sink: key %d has type %d, table was created for %d", c, keys[c].type, a->key_types[c]);
        P.key[c] = {keys[c].type, keys[c].data, keys[c].validity};
    }
    bool used[ph::AGG_MAX_AGGS] = {};
    for (int i = 0; i < a->naggs; i++)
        if (a->aggs[i].kind != PH_A_COUNT_STAR && a->aggs[i].arg >= 0 && a->aggs[i].arg < nargs) used[a->aggs[i].arg] = true;
    for (int c = 0; c < nargs; c++) {
        if (!used[c]) continue;  // placeholders of count(*) are never read
        int t = args[c].type;
        if (t != PH_I32 && t != PH_I64 && t != PH_DEC64 && t != PH_DATE) { ph::set_error("ph_agg_sink: argument %d has type %d", c, t); return PH_EUNSUPPORTED; }
        P.arg[c] = {t == PH_DATE ? PH_I32 : t, args[c].data, args[c].validity};
    }
    for (int i = 0; i < a->naggs; i++) {
        P.agg_kind[i] = a->aggs[i].kind;
        P.agg_arg[i] = a->aggs[i].arg;
        PH_REQUIRE(a->aggs[i].kind == PH_A_COUNT_STAR || (a->aggs[i].arg >= 0 && a->aggs[i].arg < nargs),
                   "ph_agg_sink: aggregate %d refers to argument %d of %d", i, a->aggs[i].arg, nargs);
    }
    P.positional = positional;
    P.ngroups = a->counters;
    P.error_flag = a->counters + 1;
    P.skip_from = a->counters + 2;
    const int nbatches = (int)((n + AGG_BATCH - 1) / AGG_BATCH);
    int start = 0;
    while (start < nbatches) {
        // make room for the first batch of this round, then enqueue every remaining batch behind
        // its growth guard; only one host round trip per round
        int64_t ng = 0;
        PH_CHECK(ph_agg_group_count(a, &ng));
        int64_t m0 = std::min(AGG_BATCH, n - (int64_t)start * AGG_BATCH);
        int64_t cap = a->cap;
        while (cap / 2 - ng <= m0) cap *= 2;   // Resize rule (aggregate_hash.go:214-217); gcap = cap/2
        if (cap != a->cap) PH_CHECK(agg_resize(a, cap, (int)ng));
        const int big = 0x7fffffff;
        PH_HIP(hipMemcpyAsync(a->counters + 2, &big, 4, hipMemcpyHostToDevice, a->ctx->stream));
        for (int b = start; b < nbatches; b++) {
            int64_t off = (int64_t)b * AGG_BATCH;
            int64_t m = std::min(AGG_BATCH, n - off);
            P.sel = sel ? sel + off : nullptr;
            P.n = m;
            P.row_base = row_base + off;
            P.batch_index = b;
            if (!sel || positional) {
                // identity selection / positional args: shift the base pointers instead
                for (int c = 0; c < a->nkeys && !sel; c++) {
                    int w = ph::type_width(keys[c].type);
                    P.key[c].data = (const char *)keys[c].data + off * w;
                    P.key[c].validity = keys[c].validity ? keys[c].validity + off / 8 : nullptr;
                }
                for (int c = 0; c < nargs && (!sel || positional); c++) {
                    if (!used[c]) continue;
                    int w = ph::type_width(args[c].type);
                    P.arg[c].data = (const char *)args[c].data + off * w;
                    P.arg[c].validity = args[c].validity ? args[c].validity + off / 8 : nullptr;
                }
            }
            P.slots = a->slots;
            P.mask = (uint64_t)a->cap - 1;
            P.gkeys = a->gkeys; P.gnull = a->gnull; P.sum_lo = a->sum_lo; P.sum_hi = a->sum_hi;
            P.cnt = a->cnt; P.first_row = a->first_row; P.gcap = a->gcap;
            // staging table: as many entries as fit 48 KiB (12 B + 12 B per aggregate each)
            int per_entry = 12 + 12 * std::max(a->naggs, 1);
            int slots = 64;
            while (slots * 2 * per_entry <= 48 * 1024 && slots < 4096) slots *= 2;
            P.lds_slots = slots;
            size_t lds = (size_t)slots * (8 + 4) + (size_t)slots * a->naggs * (8 + 4);
            if (b > start)
                ph::agg_guard_kernel<<<1, 1, 0, a->ctx->stream>>>(a->counters, a->gcap, m, b, a->counters + 2);
            // few long-lived workgroups: every flush costs one HBM atomic per live entry
            int grid = (int)std::min<int64_t>((m + 255) / 256, (int64_t)a->ctx->cu_count * 4);
            ph::agg_sink_kernel<<<grid, 256, lds, a->ctx->stream>>>(P);
            PH_HIP(hipGetLastError());
        }
        int refused = big;
        if (nbatches - start > 1) PH_CHECK(a->ctx->download(&refused, a->counters + 2, 4));
        start = refused == big ? nbatches : refused;
    }
    a->rows_sunk += n;
    return PH_OK;
}

extern "C" int ph_agg_finalize(ph_agg *a, int64_t max_groups, int64_t *first_row, int64_t *keys,
                               uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && max_groups >= 0, "ph_agg_finalize: bad arguments");
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(a, &ng));
    if (ng > max_groups) { ph::set_error("ph_agg_finalize: %lld groups, room for %lld", (long long)ng, (long long)max_groups); return PH_ECAPACITY; }
    if (ng == 0) return PH_OK;
    size_t na = (size_t)std::max(a->naggs, 1), g = (size_t)ng;
    std::vector<long long> fr(g);
    std::vector<unsigned long long> gk(g * a->nkeys), lo(g * na), cn(g * na);
    std::vector<long long> hi(g * na);
    std::vector<unsigned> gn(g);
    ph_ctx *cx = a->ctx;
    PH_CHECK(cx->download(fr.data(), a->first_row, (int64_t)(g * 8)));
    PH_CHECK(cx->download(gk.data(), a->gkeys, (int64_t)(g * a->nkeys * 8)));
    PH_CHECK(cx->download(gn.data(), a->gnull, (int64_t)(g * 4)));
    PH_CHECK(cx->download(lo.data(), a->sum_lo, (int64_t)(g * na * 8)));
    PH_CHECK(cx->download(hi.data(), a->sum_hi, (int64_t)(g * na * 8)));
    PH_CHECK(cx->download(cn.data(), a->cnt, (int64_t)(g * na * 8)));
    // first-seen order = the reference's insertion order (GroupedAggrHashTable.Scan, :424-438)
    std::vector<int64_t> order(g);
    for (size_t i = 0; i < g; i++) order[i] = (int64_t)i;
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return fr[(size_t)x] < fr[(size_t)y]; });
    for (size_t o = 0; o < g; o++) {
        size_t src = (size_t)order[o];
        if (first_row) first_row[o] = fr[src];
        for (int c = 0; c < a->nkeys; c++) {
            if (keys) keys[o * a->nkeys + c] = (int64_t)gk[src * a->nkeys + c];
            if (key_null) key_null[o * a->nkeys + c] = (gn[src] >> c) & 1;
        }
        for (int i = 0; i < a->naggs; i++) {
            if (sum_lo) sum_lo[o * a->naggs + i] = lo[src * na + i];
            if (sum_hi) {
                int kind = a->aggs[i].kind;
                // MIN/MAX keep their value in the low word: sign-extend it for the caller
                sum_hi[o * a->naggs + i] = (kind == PH_A_MIN || kind == PH_A_MAX)
                                               ? ((long long)lo[src * na + i] < 0 ? -1 : 0) : hi[src * na + i];
            }
            if (count) count[o * a->naggs + i] = cn[src * na + i];
        }
    }
    return PH_OK;
}

extern "C" int ph_agg_topk(ph_agg *a, int32_t agg_index, int32_t descending, int64_t k, int64_t max_groups,
                           int64_t *n_out, int64_t *first_row, int64_t *keys, uint8_t *key_null,
                           uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && n_out && k >= 0 && max_groups >= 0 && agg_index >= 0 && agg_index < a->naggs,
               "ph_agg_topk: bad arguments");
    int kind = a->aggs[agg_index].kind;
    PH_REQUIRE(kind == PH_A_SUM || kind == PH_A_MIN || kind == PH_A_MAX || kind == PH_A_AVG,
               "ph_agg_topk: aggregate %d is not ordered by its sum/min/max value", agg_index);
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(a, &ng));
    *n_out = 0;
    if (ng == 0 || k == 0) return PH_OK;
    ph_ctx *ctx = a->ctx;
    int cap = (int)std::min<int64_t>(max_groups, ng);
    int *ids = nullptr, *meta = nullptr;
    unsigned long long *state = nullptr;  // [0] prefix [1] remaining k, then 256 x u32 histogram
    PH_CHECK(ctx->pool_alloc((int64_t)std::max(cap, 1) * 4, (void **)&ids));
    PH_CHECK(ctx->pool_alloc(8, (void **)&meta));
    PH_CHECK(ctx->pool_alloc(16 + 1024, (void **)&state));
    PH_HIP(hipMemsetAsync(meta, 0, 8, ctx->stream));
    PH_HIP(hipMemsetAsync(state, 0, 16 + 1024, ctx->stream));
    unsigned long long kk = (unsigned long long)k;
    PH_HIP(hipMemcpyAsync(state + 1, &kk, 8, hipMemcpyHostToDevice, ctx->stream));
    unsigned *hist = (unsigned *)(state + 2);
    int tg = (int)std::min<int64_t>((ng + 255) / 256, ctx->cu_count * 2);
    ph::topk_check_kernel<<<tg, 256, 0, ctx->stream>>>(a->sum_lo, a->sum_hi, a->naggs, agg_index, (int)ng, meta + 1);
    for (int pass = 7; pass >= 0; pass--) {  // radix select, most significant byte first
        ph::topk_hist_kernel<<<tg, 256, 0, ctx->stream>>>(a->sum_lo, a->naggs, agg_index, (int)ng, descending, pass, state, hist);
        ph::topk_resolve_kernel<<<1, 256, 0, ctx->stream>>>(state, hist, pass);
    }
    ph::topk_collect_kernel<<<tg, 256, 0, ctx->stream>>>(a->sum_lo, a->naggs, agg_index, (int)ng, descending, (long long)k,
                                                          state, ids, meta, cap);
    PH_HIP(hipGetLastError());
    ctx->pool_release(state);
    int m[2] = {0, 0};
    int rc = ctx->download(m, meta, 8);
    if (rc == PH_OK && m[1]) { ph::set_error("ph_agg_topk: a sum does not fit int64; use ph_agg_finalize"); rc = PH_EOVERFLOW; }
    if (rc == PH_OK && m[0] > cap) { ph::set_error("ph_agg_topk: %d qualifying groups, room for %d", m[0], cap); rc = PH_ECAPACITY; }
    if (rc != PH_OK) { ctx->pool_release(ids); ctx->pool_release(meta); return rc; }
    size_t n = (size_t)m[0], na = (size_t)a->naggs, nk = (size_t)a->nkeys;
    struct Row { long long fr; std::vector<unsigned long long> k, lo, cn; std::vector<long long> hi; unsigned null; };
    std::vector<Row> rows(n);
    if (n > 0) {
        // one packed device buffer: [first n][null n (padded to 8)][keys n*nk][lo n*na][hi n*na][cnt n*na]
        size_t words = n + n + n * nk + 3 * n * na;
        unsigned long long *pack = nullptr;
        rc = ctx->pool_alloc((int64_t)words * 8, (void **)&pack);
        if (rc == PH_OK) {
            long long *o_first = (long long *)pack;
            unsigned *o_null = (unsigned *)(pack + n);
            unsigned long long *o_keys = pack + 2 * n, *o_lo = o_keys + n * nk;
            long long *o_hi = (long long *)(o_lo + n * na);
            unsigned long long *o_cnt = (unsigned long long *)(o_hi + n * na);
            ph::agg_pack_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(ids, (int)n, a->nkeys, a->naggs, a->first_row,
                a->gkeys, a->gnull, a->sum_lo, a->sum_hi, a->cnt, o_first, o_keys, o_null, o_lo, o_hi, o_cnt);
            std::vector<unsigned long long> host(words);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
            if (rc == PH_OK) rc = ctx->download(host.data(), pack, (int64_t)words * 8);
            if (rc == PH_OK) {
                const unsigned *hn = (const unsigned *)(host.data() + n);
                const unsigned long long *hk = host.data() + 2 * n, *hl = hk + n * nk, *hh = hl + n * na, *hc = hh + n * na;
                for (size_t i = 0; i < n; i++) {
                    Row &r = rows[i];
                    r.fr = (long long)host[i];
                    r.null = hn[i];
                    r.k.assign(hk + i * nk, hk + (i + 1) * nk);
                    r.lo.assign(hl + i * na, hl + (i + 1) * na);
                    r.hi.assign((const long long *)hh + i * na, (const long long *)hh + (i + 1) * na);
                    r.cn.assign(hc + i * na, hc + (i + 1) * na);
                }
            }
            ctx->pool_release(pack);
        }
    }
    ctx->pool_release(ids);
    ctx->pool_release(meta);
    if (rc != PH_OK) return rc;
    std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.fr < y.fr; });
    for (size_t o = 0; o < n; o++) {
        const Row &r = rows[o];
        if (first_row) first_row[o] = r.fr;
        for (size_t c = 0; c < nk; c++) {
            if (keys) keys[o * nk + c] = (int64_t)r.k[c];
            if (key_null) key_null[o * nk + c] = (r.null >> c) & 1;
        }
        for (size_t i = 0; i < na; i++) {
            if (sum_lo) sum_lo[o * na + i] = r.lo[i];
            if (sum_hi) sum_hi[o * na + i] = r.hi[i];
            if (count) count[o * na + i] = r.cn[i];
        }
    }
    *n_out = (int64_t)n;
    return PH_OK;
}
