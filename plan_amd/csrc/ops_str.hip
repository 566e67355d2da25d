// String keys: VARCHAR group-by / join keys that are not small dictionaries.
//
// The reference hashes a VARCHAR key with util.HashBytes (pkg/chunk/hash.go:182-207, pkg/util/hash.go:13-65) and compares
// candidate rows byte by byte (Match / TemplatedMatchType, pkg/compute/util_match.go:25-301) inside its group table
// (aggregate_hash.go:201-391) and its join table (join_table.go:85-336). On the device the two steps are ONE primitive,
// string interning: every row's string is looked up in an open-addressing table keyed by the same hash_bytes and
// verified by the same byte compare; the first row to claim a slot becomes the string's REPRESENTATIVE, and every row
// gets the representative's row id as its code. Equal strings -> equal codes, different strings -> different codes
// (also when their hashes collide: the compare decides), so the integer group-by and join machinery (ph_agg_*,
// ph_join_*) runs unchanged on the int32 codes, and a code leads straight back to the bytes of its string.
// NULL strings get the code -1 (ph_strdict_build) / -2 (ph_strdict_lookup: never equal to a build-side NULL).
#include <algorithm>

#include "common.h"
#include "device_util.h"

struct ph_strdict {
    ph_ctx *ctx = nullptr;
    const int32_t *off = nullptr;   // the build column (not owned: it must outlive the dictionary)
    const char *bytes = nullptr;
    int32_t *slots = nullptr;       // representative row + 1, 0 = empty
    int64_t cap = 0;
};

namespace ph {

__device__ __forceinline__ bool str_equal(const char *a, int la, const char *b, int lb) {
    if (la != lb) return false;
    for (int i = 0; i < la; i++) if (a[i] != b[i]) return false;
    return true;
}

// BUILD = true: insert-or-find over the dictionary's own column; false: find only, the probe column is another one
template <bool BUILD>
__global__ __launch_bounds__(256) void str_intern_kernel(const int32_t *__restrict__ doff, const char *__restrict__ dbytes,
                                                         const int32_t *__restrict__ poff, const char *__restrict__ pbytes,
                                                         const uint8_t *__restrict__ validity, const int32_t *__restrict__ sel, int64_t i0, int64_t n,
                                                         int32_t *slots, unsigned long long mask, int32_t *__restrict__ codes) {
    for (int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = sel ? sel[i] : i;
        if (!bit_valid(validity, row)) { codes[i] = BUILD ? -1 : -2; continue; }
        const char *s = pbytes + poff[row];
        const int len = poff[row + 1] - poff[row];
        unsigned long long slot = hash_bytes((const uint8_t *)s, (uint64_t)len) & mask;
        int code = -2;
        for (unsigned long long step = 0; step <= mask; step++) {
            // a slot changes once, from empty to its row: a CACHED non-zero value is final, only an empty reading has to be confirmed at the
            // coherence point (a column of few distinct strings — 25 country codes in 680 k rows — hammered 25 slots with device-scope loads: 325 us)
            int cur = slots[slot];   // (a plain load: L2 / L1 may serve it)
            if (cur == 0) cur = __hip_atomic_load(&slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0) {
                if (!BUILD) break;                                  // not in the dictionary
                const int prev = atomicCAS(&slots[slot], 0, (int)row + 1);
                if (prev == 0) { code = (int)row; break; }          // this row represents its string
                cur = prev;
            }
            const int r = cur - 1;
            if (str_equal(s, len, dbytes + doff[r], doff[r + 1] - doff[r])) { code = r; break; }
            slot = (slot + 1) & mask;
        }
        codes[i] = code;
    }
}

}  // namespace ph

extern "C" int ph_strdict_build(ph_ctx *ctx, const ph_col *col, const int32_t *sel, int64_t n, int32_t *codes_out_dev, ph_strdict **out) {
    PH_REQUIRE(ctx && col && out && n >= 0 && (n == 0 || codes_out_dev), "ph_strdict_build: bad arguments");
    if (col->type != PH_STR || !col->data || (!col->aux && col->aux_bytes > 0)) { ph::set_error("ph_strdict_build: a PH_STR column (offsets + bytes)"); return PH_EUNSUPPORTED; }
    PH_REQUIRE(n < (1ll << 30), "ph_strdict_build: %lld rows", (long long)n);
    ph_strdict *d = new ph_strdict();
    d->ctx = ctx;
    d->off = (const int32_t *)col->data;
    d->bytes = (const char *)col->aux;
    d->cap = 1024;
    while (d->cap < 2 * n) d->cap <<= 1;     // at most half full
    int rc = ctx->pool_alloc(d->cap * 4, (void **)&d->slots);
    if (rc != PH_OK) { delete d; return rc; }
    if (hipMemsetAsync(d->slots, 0, (size_t)d->cap * 4, ctx->stream) != hipSuccess) { ctx->pool_release(d->slots); delete d; ph::set_error("ph_strdict_build: memset failed"); return PH_EHIP; }
    if (n > 0) {
        // A column of FEW distinct strings (25 country codes in 680 k rows) starts with every resident thread finding its slot empty and
        // claiming it: half a million compare-and-swaps on 25 addresses, one after the other at the memory side (283 us). A first launch of
        // eight workgroups over the first rows fills the table with what is common; the rest mostly finds (cached reads, no atomics).
        const int64_t head = n > 65536 ? 8192 : 0;
        if (head) ph::str_intern_kernel<true><<<8, 256, 0, ctx->stream>>>(d->off, d->bytes, d->off, d->bytes, col->validity, sel, 0, head, d->slots,
                                                                          (unsigned long long)d->cap - 1, codes_out_dev);
        const int grid = (int)std::min<int64_t>((n - head + 255) / 256, (int64_t)ctx->cu_count * 8);
        ph::str_intern_kernel<true><<<grid, 256, 0, ctx->stream>>>(d->off, d->bytes, d->off, d->bytes, col->validity, sel, head, n, d->slots,
                                                                     (unsigned long long)d->cap - 1, codes_out_dev);
        if (hipGetLastError() != hipSuccess) { ctx->pool_release(d->slots); delete d; ph::set_error("ph_strdict_build: launch failed"); return PH_EHIP; }
    }
    *out = d;
    return PH_OK;
}

extern "C" int ph_strdict_lookup(ph_strdict *d, const ph_col *col, const int32_t *sel, int64_t n, int32_t *codes_out_dev) {
    PH_REQUIRE(d && col && n >= 0 && (n == 0 || codes_out_dev), "ph_strdict_lookup: bad arguments");
    if (col->type != PH_STR || !col->data) { ph::set_error("ph_strdict_lookup: a PH_STR column (offsets + bytes)"); return PH_EUNSUPPORTED; }
    if (n == 0) return PH_OK;
    ph_ctx *ctx = d->ctx;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
    ph::str_intern_kernel<false><<<grid, 256, 0, ctx->stream>>>(d->off, d->bytes, (const int32_t *)col->data, (const char *)col->aux, col->validity, sel, 0, n,
                                                                  d->slots, (unsigned long long)d->cap - 1, codes_out_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" void ph_strdict_free(ph_strdict *d) {
    if (!d) return;
    if (d->slots) d->ctx->pool_release(d->slots);
    delete d;
}

// strings of rows rows_host[0..n) of a PH_STR table column, for the host to print group keys that are codes
extern "C" int ph_table_strings(ph_ctx *ctx, const ph_table *t, int32_t c, const int64_t *rows_host, int64_t n, int32_t *out_offsets, char *out_bytes,
                                int64_t out_capacity) {
    PH_REQUIRE(ctx && t && c >= 0 && c < (int32_t)t->cols.size() && n >= 0 && (n == 0 || (rows_host && out_offsets)), "ph_table_strings: bad arguments");
    const ph_table::column &col = t->cols[(size_t)c];
    if (col.type != PH_STR) { ph::set_error("ph_table_strings: column %d is not a PH_STR column", c); return PH_EUNSUPPORTED; }
    if (out_offsets) out_offsets[0] = 0;
    if (n == 0) return PH_OK;
    // the rows' strings gathered on the device (ph_substring over the row ids: the whole string of each), then two downloads — the offsets
    // and the bytes. (Two small downloads PER ROW, the first form, were 19 ms of Q21's 82 at SF10: 4 000 group keys.)
    std::vector<int32_t> r32((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        PH_REQUIRE(rows_host[i] >= 0 && rows_host[i] < t->nrows, "ph_table_strings: row %lld out of range", (long long)rows_host[i]);
        r32[(size_t)i] = (int32_t)rows_host[i];
    }
    void *rows_dev = nullptr, *off_dev = nullptr, *bytes_dev = nullptr;
    const int64_t cap = out_capacity > 0 ? out_capacity : 1;
    int rc = ctx->pool_alloc(n * 4, &rows_dev);
    if (rc == PH_OK) rc = ctx->pool_alloc((n + 1) * 4, &off_dev);
    if (rc == PH_OK) rc = ctx->pool_alloc(cap + 64, &bytes_dev);
    int64_t nbytes = 0;
    if (rc == PH_OK) rc = ph_dev_upload(ctx, rows_dev, r32.data(), n * 4);
    if (rc == PH_OK) {
        ph_col v{};
        v.type = PH_STR; v.data = col.data; v.aux = col.aux; v.aux_bytes = col.aux_bytes; v.validity = col.validity;
        rc = ph_substring(ctx, &v, 1, INT64_MAX, (const int32_t *)rows_dev, n, (int32_t *)off_dev, (uint8_t *)bytes_dev, cap, &nbytes);
    }
    if (rc == PH_OK) rc = ctx->download(out_offsets, off_dev, (n + 1) * 4);
    if (rc == PH_OK && nbytes > 0) rc = ctx->download(out_bytes, bytes_dev, nbytes);
    if (rows_dev) ctx->pool_release(rows_dev);
    if (off_dev) ctx->pool_release(off_dev);
    if (bytes_dev) ctx->pool_release(bytes_dev);
    if (rc != PH_OK) return rc;
    return PH_OK;
}
