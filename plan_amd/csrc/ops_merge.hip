// Merge-style N:1 lookup: both sides ordered by the join key.
//
// The reference joins with one hash table whatever the children's order (pkg/compute/executor_join.go:54-264,
// join_table.go:197-288). When the build key is a primary key stored in key order AND the probe rows arrive
// ordered by the key (a clustered table behind order-preserving filters and joins: Q9's surviving lineitem rows
// against orders), no table is needed at all: a block of probe rows spans one contiguous slice of the build
// keys, found with two searches; the slice streams through LDS once and every probe row finds its key there
// with a binary search in LDS. The build column is read once (coalesced), nothing is written but the answers —
// against a direct table that costs a fill of 4 B x key range plus a random slot read per probe.
// Both orders are VERIFIED on the device (a violation raises the ctx's deferred PH_ECONSTRAINT word, the caller
// falls back to ph_join_build + ph_join_lookup): the probe order for every row; the build order for every pair a
// block stages through LDS, and — in blocks of very sparse probes, which search the column instead — for the pairs
// around every position a search ends at. A probe key that is absent answers -1 and counts as a miss.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

constexpr int ML_ROWS = 2048;    // probe rows per 256-thread workgroup
constexpr int ML_CHUNK = 4096;   // build keys staged in LDS at a time (32 KiB)
constexpr int ML_MAX_CHUNKS = 64;   // a block whose slice is longer searches the column itself (sparse probes)

template <int KW>
__device__ __forceinline__ long long ml_key(const void *col, int64_t i) {
    return KW == 4 ? (long long)((const int32_t *)col)[i] : ((const long long *)col)[i];
}

// first index in [lo, hi) whose key is >= k (hi if none): 64-ary search, the wave reads 64 pivots per step
template <int KW>
__device__ __forceinline__ int64_t wave_lower_bound(const void *a, int64_t lo, int64_t hi, long long k, int lane) {
    while (hi > lo) {
        const int64_t step = (hi - lo + 63) / 64;
        const int64_t idx = lo + (int64_t)lane * step;
        const bool less = idx < hi && ml_key<KW>(a, idx) < k;
        const int c = __popcll(__ballot(less));   // sorted: the lanes that see a smaller key are a prefix
        if (c == 0) return lo;
        const int64_t nlo = lo + (int64_t)(c - 1) * step + 1;
        hi = lo + (int64_t)c * step < hi ? lo + (int64_t)c * step : hi;
        lo = nlo;
    }
    return lo;
}

template <int KW, bool SEL>
__global__ __launch_bounds__(256) void merge_lookup_kernel(const void *__restrict__ bkeys, int64_t nb, const void *__restrict__ pkeys,
                                                           const int32_t *__restrict__ psel, int64_t n, int32_t *__restrict__ out,
                                                           int *__restrict__ stats, int *__restrict__ unsorted) {
    __shared__ long long bk[ML_CHUNK + 1];   // the chunk and the key behind it (order check across the chunk's end)
    __shared__ long long s_b[2];
    constexpr int PT = ML_ROWS / 256;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = (int64_t)blockIdx.x * ML_ROWS, p1 = p0 + ML_ROWS < n ? p0 + ML_ROWS : n;
    long long k[PT];
    int32_t res[PT];
    bool bad = false;
#pragma unroll
    for (int q = 0; q < PT; q++) {
        const int64_t i = p0 + q * 256 + threadIdx.x;
        res[q] = -1;
        k[q] = 0;
        if (i < p1) {
            k[q] = ml_key<KW>(pkeys, SEL ? (int64_t)psel[i] : i);
            if (i + 1 < n) bad = bad || k[q] > ml_key<KW>(pkeys, SEL ? (int64_t)psel[i + 1] : i + 1);   // the probe order
        }
    }
    if (threadIdx.x < 64) {   // wave 0: the slice of the build keys this block's probe keys span
        const long long klo = ml_key<KW>(pkeys, SEL ? (int64_t)psel[p0] : p0);
        const long long khi = ml_key<KW>(pkeys, SEL ? (int64_t)psel[p1 - 1] : p1 - 1);
        const int64_t b0 = wave_lower_bound<KW>(bkeys, 0, nb, klo, lane);
        // unique ascending integers: the keys in [klo, khi] are at most khi - klo + 1 rows
        const unsigned long long span = (unsigned long long)khi - (unsigned long long)klo;
        const int64_t cap1 = span < (unsigned long long)(nb - b0) ? b0 + (int64_t)span + 1 : nb;
        const int64_t b1 = khi == INT64_MAX ? nb : wave_lower_bound<KW>(bkeys, b0, cap1 < nb ? cap1 : nb, khi + 1, lane);
        if (lane == 0) { s_b[0] = b0; s_b[1] = b1; }
    }
    __syncthreads();
    const int64_t b0 = s_b[0], b1 = s_b[1];
    if (b1 - b0 > (int64_t)ML_MAX_CHUNKS * ML_CHUNK) {
        // sparse probes: the slice is too long to stream for 1024 rows — every row searches the column
#pragma unroll
        for (int q = 0; q < PT; q++) {
            const int64_t i = p0 + q * 256 + threadIdx.x;
            if (i >= p1) continue;
            int64_t lo = b0, hi = b1;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ml_key<KW>(bkeys, mid) < k[q]) lo = mid + 1; else hi = mid; }
            if (lo < b1 && ml_key<KW>(bkeys, lo) == k[q]) res[q] = (int32_t)lo;
            // the build order (strict) around the position the search ended at: a duplicated or descending pair there
            // is what could have misled it (the streaming branch below checks every pair it stages)
            if (lo < nb) {
                const long long at = ml_key<KW>(bkeys, lo);
                bad = bad || (lo > 0 && ml_key<KW>(bkeys, lo - 1) >= at) || (lo + 1 < nb && at >= ml_key<KW>(bkeys, lo + 1));
            }
        }
    } else {
        for (int64_t c = b0; c < b1; c += ML_CHUNK) {
            const int m = (int)(b1 - c < ML_CHUNK ? b1 - c : ML_CHUNK);
            __syncthreads();   // the previous chunk's searches are done
            {   // all 16 reads of a thread in flight before the first store (a load -> store loop waits per element)
                long long v[ML_CHUNK / 256];
#pragma unroll
                for (int q = 0; q < ML_CHUNK / 256; q++) {
                    const int e = q * 256 + threadIdx.x;
                    v[q] = e <= m && c + e < nb ? ml_key<KW>(bkeys, c + e) : INT64_MAX;
                }
#pragma unroll
                for (int q = 0; q < ML_CHUNK / 256; q++) bk[q * 256 + threadIdx.x] = v[q];
                if (threadIdx.x == 0) bk[ML_CHUNK] = m == ML_CHUNK && c + m < nb ? ml_key<KW>(bkeys, c + m) : INT64_MAX;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < m; e += 256) bad = bad || (c + e + 1 < nb && bk[e] >= bk[e + 1]);   // the build order (strict)
            const long long first = bk[0], last = bk[m - 1];
#pragma unroll
            for (int q = 0; q < PT; q++) {
                if (res[q] >= 0 || k[q] < first || k[q] > last || p0 + q * 256 + (int64_t)threadIdx.x >= p1) continue;
                int lo = 0, hi = m;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (bk[mid] < k[q]) lo = mid + 1; else hi = mid; }
                if (lo < m && bk[lo] == k[q]) res[q] = (int32_t)(c + lo);
            }
        }
    }
    int misses = 0;
#pragma unroll
    for (int q = 0; q < PT; q++) {
        const int64_t i = p0 + q * 256 + threadIdx.x;
        if (i < p1) { out[i] = res[q]; misses += res[q] < 0; }
    }
    for (int o = 32; o > 0; o >>= 1) misses += __shfl_xor(misses, o);
    if (lane == 0 && misses && stats) atomicAdd(stats, misses);
    if (bad) atomicOr(unsorted, 1);
}

}  // namespace ph

extern "C" int ph_merge_lookup(ph_ctx *ctx, const ph_col *build_key, int64_t n_build, const ph_col *probe_key, const int32_t *sel,
                               int64_t n, int32_t strict, int32_t *out_build_dev) {
    PH_REQUIRE(ctx && build_key && probe_key && n_build >= 0 && n >= 0 && n_build < (1ll << 31) && (n == 0 || out_build_dev),
               "ph_merge_lookup: bad arguments");
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : (t == PH_I64 || t == PH_DEC64) ? 8 : 0; };
    const int kw = width(build_key->type);
    if (kw == 0 || width(probe_key->type) != kw || build_key->validity || probe_key->validity) {
        ph::set_error("ph_merge_lookup: one 4- or 8-byte integer key column of the same width on both sides, no NULLs");
        return PH_EUNSUPPORTED;
    }
    if (n == 0) return PH_OK;
    if (n_build == 0) { PH_HIP(hipMemsetAsync(out_build_dev, 0xff, (size_t)n * 4, ctx->stream)); return PH_OK; }
    int *words = nullptr;
    PH_CHECK(ctx->deferred_words(&words));
    const int grid = (int)((n + ph::ML_ROWS - 1) / ph::ML_ROWS);
    int *stats = strict ? words + 1 : nullptr;   // a miss of a strict lookup is the deferred word of ph_join_lookup_strict
#define PH_ML(KWV, SELV) ph::merge_lookup_kernel<KWV, SELV><<<grid, 256, 0, ctx->stream>>>(build_key->data, n_build, probe_key->data, sel, n, out_build_dev, stats, words + 3)
    if (kw == 4) { if (sel) PH_ML(4, true); else PH_ML(4, false); }
    else { if (sel) PH_ML(8, true); else PH_ML(8, false); }
#undef PH_ML
    PH_HIP(hipGetLastError());
    ctx->deferred_pending = true;
    return PH_OK;
}

// ------------------------------------------------------------------ pairs against a CLUSTERED key column, no table
// A join whose build side is a big table stored in key order (lineitem by l_orderkey) and whose probe side is small: building any table over
// the build side (the reference always does, join_table.go:85-288; the direct table with duplicate chains here) costs a pass — several — over
// ALL its rows, for a probe side that touches a sliver of them (Q21 at SF10: 0.7 M probe rows against 60 M lines, 2.5 ms of table building for
// 3.6 M pairs). Sorted keys need no table: every probe row finds the run of its key with a binary search over the column and the pairs are
// (probe position, every row of the run). Two launches around one offset scan; the pairs come out in probe order, a run's rows ascending.
namespace ph {

template <int KW>
__device__ __forceinline__ long long sp_key(const void *data, int64_t r) {
    return KW == 4 ? (long long)((const int32_t *)data)[r] : ((const int64_t *)data)[r];
}

// first row whose key is >= k
template <int KW>
__device__ __forceinline__ int64_t sp_lower_bound(const void *bkey, int64_t nb, long long k) {
    int64_t lo = 0, hi = nb;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sp_key<KW>(bkey, mid) < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int KW>
__global__ __launch_bounds__(256) void sorted_pairs_count_kernel(const void *__restrict__ bkey, int64_t nb, const void *__restrict__ pkey, const uint8_t *pvalid,
                                                                 const int32_t *__restrict__ sel, int64_t n, int32_t *__restrict__ first, int32_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = sel ? sel[i] : i;
        int64_t lo = 0, cnt = 0;
        if (!pvalid || bit_valid(pvalid, r)) {   // a NULL key matches nothing (prepareKeys, join_table.go:152)
            const long long k = sp_key<KW>(pkey, r);
            lo = sp_lower_bound<KW>(bkey, nb, k);
            // the run: short in the shape this form is for (a handful of lines per order): walk it; a long run finishes with a second search
            int64_t hi = lo;
            while (hi < nb && hi - lo < 16 && sp_key<KW>(bkey, hi) == k) hi++;
            if (hi < nb && hi - lo == 16 && sp_key<KW>(bkey, hi) == k) hi = sp_lower_bound<KW>(bkey, nb, k + 1);
            cnt = hi - lo;
        }
        first[i] = (int32_t)lo;
        counts[i] = (int32_t)cnt;
    }
}

__global__ __launch_bounds__(256) void sorted_pairs_emit_kernel(const int32_t *__restrict__ first, const int32_t *__restrict__ offs, const int64_t *__restrict__ total,
                                                                const int32_t *__restrict__ sel, int64_t n, int rowids, int64_t cap,
                                                                int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t o = offs[i], e = i + 1 < n ? offs[i + 1] : *total;
        const int32_t p = rowids && sel ? sel[i] : (int32_t)i;
        for (int64_t q = o; q < e && q < cap; q++) { out_probe[q] = p; out_build[q] = first[i] + (int32_t)(q - o); }
    }
}
}  // namespace ph

// out_probe: the probe ROW id (sel[i], or i without a selection) — as ph_join_probe_inner reports it
extern "C" int ph_join_sorted_pairs(ph_ctx *ctx, const ph_col *build_key, int64_t n_build, const ph_col *probe_key, const int32_t *sel, int64_t n,
                                    int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    PH_REQUIRE(ctx && build_key && probe_key && n_out && n_build >= 0 && n >= 0 && n_build < (1ll << 31) && n < (1ll << 31) && cap >= 0,
               "ph_join_sorted_pairs: bad arguments");
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : (t == PH_I64 || t == PH_DEC64) ? 8 : 0; };
    const int kw = width(build_key->type);
    if (kw == 0 || width(probe_key->type) != kw || build_key->validity) {
        ph::set_error("ph_join_sorted_pairs: one 4- or 8-byte integer key column of the same width on both sides, no NULLs on the build side");
        return PH_EUNSUPPORTED;
    }
    *n_out = 0;
    if (n == 0 || n_build == 0) return PH_OK;
    int32_t *first = nullptr, *counts = nullptr;
    int64_t *total = nullptr;
    PH_CHECK(ctx->pool_alloc(n * 4, (void **)&first));
    int rc = ctx->pool_alloc(n * 4 + 64, (void **)&counts);
    if (rc == PH_OK) rc = ctx->pool_alloc(16, (void **)&total);
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    if (rc == PH_OK) {
        if (kw == 4) ph::sorted_pairs_count_kernel<4><<<grid, 256, 0, ctx->stream>>>(build_key->data, n_build, probe_key->data, probe_key->validity, sel, n, first, counts);
        else ph::sorted_pairs_count_kernel<8><<<grid, 256, 0, ctx->stream>>>(build_key->data, n_build, probe_key->data, probe_key->validity, sel, n, first, counts);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    ph::ScanPublish pub;
    if (rc == PH_OK) rc = ctx->arm_count(&pub);
    if (rc == PH_OK) rc = ph::exclusive_scan_i32(ctx, counts, n, total, pub.seq ? &pub : nullptr);
    if (rc == PH_OK) {
        ph::sorted_pairs_emit_kernel<<<grid, 256, 0, ctx->stream>>>(first, counts, total, sel, n, 1, cap, out_probe_dev, out_build_dev);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    int64_t m = 0;
    if (rc == PH_OK) rc = pub.seq ? ctx->count_back(pub, &m, total, -1, "ph_join_sorted_pairs") : ctx->download(&m, total, 8);   // (the count is needed here: never deferred)
    ctx->pool_release(first);
    if (counts) ctx->pool_release(counts);
    if (total) ctx->pool_release(total);
    PH_CHECK(rc);
    *n_out = m;
    if (m > cap) { ph::set_error("ph_join_sorted_pairs: %lld pairs, room for %lld", (long long)m, (long long)cap); return PH_ECAPACITY; }
    return PH_OK;
}

// ------------------------------------------------------------------ N:1 lookup into a table stored in runs of one length, no table
// The build table is clustered by its first key in runs of a constant length over consecutive values (ph_table_col_run_len: partsupp by
// ps_partkey, four rows per part), the join key is (that column, a second column) and unique: the rows of a first key are at
// (key - min) * run_len .. + run_len, the second key picks the row among them — one short read of the second key column per probe row, where
// the node table took a build over the (reduced) build side and two dependent random reads per probe (Q9 at SF10: 211 us -> ~40 us).
namespace ph {
template <int KW, int KW2>
__global__ __launch_bounds__(256) void run_lookup_kernel(const void *__restrict__ bkey2, int64_t nruns, long long kmin, int c, const void *__restrict__ pk1,
                                                         const uint8_t *pv1, const void *__restrict__ pk2, const uint8_t *pv2, const int32_t *__restrict__ sel,
                                                         int64_t n, int32_t *__restrict__ out, int *__restrict__ stats) {
    // Four probe rows per thread and step, every stage's loads issued for all four before any is used: the row ids, the two keys (two scattered
    // reads per row when the keys are table columns behind row ids), then the run's second keys (ONE 16-byte read for runs of four 4-byte keys)
    constexpr int U = 4;
    int misses = 0, multi = 0;
    const bool quad = KW2 == 4 && c == 4 && (reinterpret_cast<uintptr_t>(bkey2) & 15) == 0;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * U; i0 < n; i0 += (int64_t)gridDim.x * 256 * U) {
        int64_t r[U];
        long long d[U], k2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = i0 + u < n ? (sel ? sel[i0 + u] : i0 + u) : -1;
#pragma unroll
        for (int u = 0; u < U; u++) {
            live[u] = r[u] >= 0 && (!pv1 || bit_valid(pv1, r[u])) && (!pv2 || bit_valid(pv2, r[u]));   // a NULL key matches nothing (prepareKeys, join_table.go:152)
            d[u] = live[u] ? sp_key<KW>(pk1, r[u]) - kmin : -1;
            k2[u] = live[u] ? sp_key<KW2>(pk2, r[u]) : 0;
            live[u] = live[u] && d[u] >= 0 && d[u] < nruns;
        }
        int32_t row[U];
        if (quad) {
            int4 run[U];
#pragma unroll
            for (int u = 0; u < U; u++) run[u] = live[u] ? reinterpret_cast<const int4 *>(bkey2)[d[u]] : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = (int)k2[u];
                const int f0 = run[u].x == k, f1 = run[u].y == k, f2 = run[u].z == k, f3 = run[u].w == k;
                const int found = live[u] ? f0 + f1 + f2 + f3 : 0;
                row[u] = found ? (int32_t)(d[u] * 4 + (f0 ? 0 : f1 ? 1 : f2 ? 2 : 3)) : -1;
                multi += found > 1;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                row[u] = -1;
                if (!live[u]) continue;
                const int64_t base = d[u] * c;
                int found = 0;
                for (int j = 0; j < c; j++)
                    if (sp_key<KW2>(bkey2, base + j) == k2[u]) { if (!found) row[u] = (int32_t)(base + j); found++; }
                multi += found > 1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i0 + u < n) { out[i0 + u] = row[u]; misses += row[u] < 0; }
    }
    for (int o = 32; o > 0; o >>= 1) { misses += __shfl_xor(misses, o); multi += __shfl_xor(multi, o); }
    if ((threadIdx.x & 63) == 0 && stats) {
        if (misses) atomicAdd(stats, misses);
        if (multi) atomicAdd(stats + 1, multi);
    }
}
}  // namespace ph

extern "C" int ph_join_run_lookup(ph_ctx *ctx, const ph_col *build_key2, int64_t n_build, int64_t key1_min, int32_t run_len, const ph_col *probe_keys,
                                  const int32_t *sel, int64_t n, int32_t strict, int32_t *out_build_dev) {
    PH_REQUIRE(ctx && build_key2 && probe_keys && n_build >= 0 && n >= 0 && n_build < (1ll << 31) && run_len >= 1 && n_build % run_len == 0 && (n == 0 || out_build_dev),
               "ph_join_run_lookup: bad arguments");
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : (t == PH_I64 || t == PH_DEC64) ? 8 : 0; };
    const int kw = width(probe_keys[0].type), kw2 = width(build_key2->type);
    if (kw == 0 || kw2 == 0 || width(probe_keys[1].type) != kw2 || build_key2->validity) {
        ph::set_error("ph_join_run_lookup: 4- or 8-byte integer keys, the second of the same width on both sides, no NULLs on the build side");
        return PH_EUNSUPPORTED;
    }
    if (n == 0) return PH_OK;
    int *words = nullptr;
    PH_CHECK(ctx->deferred_words(&words));
    int *stats = strict ? words + 1 : nullptr;   // [1] misses, [2] several rows for one key: the deferred words of ph_join_lookup_strict
    const int grid = (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16);
    const int64_t nruns = n_build / run_len;
#define PH_RL(A, B) ph::run_lookup_kernel<A, B><<<grid, 256, 0, ctx->stream>>>(build_key2->data, nruns, (long long)key1_min, (int)run_len, probe_keys[0].data, probe_keys[0].validity, \
                                                                             probe_keys[1].data, probe_keys[1].validity, sel, n, out_build_dev, stats)
    if (kw == 4 && kw2 == 4) PH_RL(4, 4); else if (kw == 4) PH_RL(4, 8); else if (kw2 == 4) PH_RL(8, 4); else PH_RL(8, 8);
#undef PH_RL
    PH_HIP(hipGetLastError());
    if (strict) ctx->deferred_pending = true;
    return PH_OK;
}

// ------------------------------------------------------------------ children per parent: count(col) over a LEFT join grouped by the parent's key
// Agg(parent key; count(child column)) <- LEFT JOIN(parent, child ON parent key = child key) with a unique parent key (Q13: orders per customer)
// is, per parent row, the number of child rows with its key — NULL where there is none (CountOp finalises a group without non-NULL inputs as NULL,
// aggr_ops.go). The reference emits the pairs (join_scan.go NextLeftJoin) and folds them again (GroupedAggrHashTable.AddChunk); here the child
// keys are counted into an array over their value range (one atomic add per child row, the array in L2) and every parent row reads its count.
namespace ph {
template <int KW>
__global__ __launch_bounds__(256) void count_by_key_kernel(const void *__restrict__ key, const uint8_t *__restrict__ valid, const int32_t *__restrict__ sel, int64_t n,
                                                           long long kmin, unsigned long long range, int32_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = sel ? sel[i] : i;
        if (valid && !bit_valid(valid, r)) continue;   // a NULL key matches nothing
        const unsigned long long d = (unsigned long long)(sp_key<KW>(key, r) - kmin);
        if (d < range) atomicAdd(&counts[d], 1);
    }
}

template <int KW>
__global__ __launch_bounds__(256) void counts_lookup_kernel(const int32_t *__restrict__ counts, long long kmin, unsigned long long range, const void *__restrict__ key,
                                                            const uint8_t *__restrict__ valid, const int32_t *__restrict__ sel, int64_t n, int64_t *__restrict__ out,
                                                            uint8_t *__restrict__ out_valid) {
    // one thread per output row, a wave's 64 validity bits stored as one 8-byte word (n is padded to whole waves by the grid)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    long long c = 0;
    if (i < n) {
        const int64_t r = sel ? sel[i] : i;
        if (!valid || bit_valid(valid, r)) {
            const unsigned long long d = (unsigned long long)(sp_key<KW>(key, r) - kmin);
            if (d < range) c = counts[d];
        }
        out[i] = c;
    }
    const unsigned long long m = __ballot(c > 0);
    if ((threadIdx.x & 63) == 0 && (i & ~63ll) < n) reinterpret_cast<unsigned long long *>(out_valid)[i >> 6] = m;
}
}  // namespace ph

extern "C" int ph_count_by_key(ph_ctx *ctx, const ph_col *child_key, const int32_t *child_sel, int64_t n_child, int64_t key_min, int64_t key_range,
                               const ph_col *parent_key, const int32_t *parent_sel, int64_t n_parent, int64_t *out_counts_dev, uint8_t *out_valid_dev) {
    PH_REQUIRE(ctx && child_key && parent_key && n_child >= 0 && n_parent >= 0 && key_range >= 1 && key_range <= (1ll << 28) && (n_parent == 0 || (out_counts_dev && out_valid_dev)),
               "ph_count_by_key: bad arguments (the key range is limited to 2^28 values)");
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : (t == PH_I64 || t == PH_DEC64) ? 8 : 0; };
    const int kw = width(child_key->type);
    if (kw == 0 || width(parent_key->type) != kw) { ph::set_error("ph_count_by_key: 4- or 8-byte integer keys of one width"); return PH_EUNSUPPORTED; }
    if (n_parent == 0) return PH_OK;
    int32_t *counts = nullptr;
    PH_CHECK(ctx->pool_alloc(key_range * 4, (void **)&counts));
    int rc = hipMemsetAsync(counts, 0, (size_t)key_range * 4, ctx->stream) == hipSuccess ? PH_OK : PH_EHIP;
    if (rc == PH_OK && n_child > 0) {
        const int grid = (int)std::min<int64_t>((n_child + 255) / 256, (int64_t)ctx->cu_count * 16);
        if (kw == 4) ph::count_by_key_kernel<4><<<grid, 256, 0, ctx->stream>>>(child_key->data, child_key->validity, child_sel, n_child, (long long)key_min, (unsigned long long)key_range, counts);
        else ph::count_by_key_kernel<8><<<grid, 256, 0, ctx->stream>>>(child_key->data, child_key->validity, child_sel, n_child, (long long)key_min, (unsigned long long)key_range, counts);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    if (rc == PH_OK) {
        const int grid = (int)((n_parent + 255) / 256);
        if (kw == 4) ph::counts_lookup_kernel<4><<<grid, 256, 0, ctx->stream>>>(counts, (long long)key_min, (unsigned long long)key_range, parent_key->data, parent_key->validity, parent_sel, n_parent, out_counts_dev, out_valid_dev);
        else ph::counts_lookup_kernel<8><<<grid, 256, 0, ctx->stream>>>(counts, (long long)key_min, (unsigned long long)key_range, parent_key->data, parent_key->validity, parent_sel, n_parent, out_counts_dev, out_valid_dev);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    ctx->pool_release(counts);
    if (rc != PH_OK) ph::set_error("ph_count_by_key: launch failed");
    return rc;
}
