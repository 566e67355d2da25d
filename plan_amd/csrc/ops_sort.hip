// ORDER BY on the device: ph_sort_rows.
//
// Replaces LocalSort.SinkChunk / Sort for fixed-size keys (reference pkg/compute/sort_local.go:64-250,
// key layout sort_layout.go:29-88, encoders sort_encoder.go:33-114, RadixScatter
// sort_radix.go:242-380): every ORDER BY column becomes a byte-comparable key —
//   [1 byte: 0 = NULL, 1 = value (the reference always sorts NULLs first, sort_layout.go:46)]
//   [value bytes, big-endian, sign bit flipped; all value bytes inverted for DESC]
//   INTEGER: 4 bytes; DATE: (year, month, day) = the order of the day number; DECIMAL:
//   dec.Int64(2) -> (whole, frac): the value ROUNDED half-even to two decimals (a reference
//   quirk: finer decimals compare equal when they round to the same cents)
// and rows are ordered by memcmp of the concatenated keys. Ties keep no defined order in the
// reference (its radix/pdq sort is not stable); here they keep their input order.
//
// Device form: each column is normalised to an unsigned 64-bit word with the same order (value
// XOR sign bit, inverted for DESC; 0 for NULL) plus the NULL flag, and the permutation is sorted
// column by column from the LAST ORDER BY column to the first with stable LSD radix passes over
// the bytes that actually vary (an OR-reduction of key XOR first key finds them: a date column
// needs two passes, a constant column none). A pass is histogram -> scan -> stable scatter of
// (key word, row word) pairs; the row word carries the column's NULL flag in bit 31, which is the
// most significant digit of the column.
#include <algorithm>

#include "common.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

constexpr int SORT_TILE = 256;          // rows ranked together (one per thread, in order)
constexpr int SORT_TILES_PER_WG = 16;   // consecutive tiles of a workgroup's chunk

struct SortCol {
    int type;       // PH_I32 / PH_DATE / PH_CODE8 / PH_DEC64
    int scale;      // DEC64
    const void *data;
    const uint8_t *validity;
    int descending;
};

__device__ __forceinline__ long long round_cents(long long x, int scale) {
    // dec.Int64(2): the unscaled value at scale 2, half-even (sort_encoder.go:65-70)
    if (scale == 2) return x;
    if (scale < 2) {
        for (int s = scale; s < 2; s++) x *= 10;
        return x;
    }
    long long p = 1;
    for (int s = 2; s < scale; s++) p *= 10;
    long long q = x / p, r = x % p;          // truncation toward zero
    long long ar = r < 0 ? -r : r, twice = 2 * ar;
    if (twice > p || (twice == p && (q & 1))) q += x < 0 ? -1 : 1;
    return q;
}

// perm[i] = row id at position i (in/out across columns); writes the (key, row|null<<31) pairs
__global__ __launch_bounds__(256) void sort_norm_kernel(SortCol C, const int32_t *__restrict__ perm, int64_t n,
                                                        unsigned long long *__restrict__ keys, unsigned *__restrict__ rows) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = perm[i];
        unsigned long long k = 0;
        unsigned nullbit = 1;   // digit of the NULL byte: 0 = NULL (first), 1 = value
        if (bit_valid(C.validity, r)) {
            long long v;
            switch (C.type) {
            case PH_I32: case PH_DATE: v = ((const int32_t *)C.data)[r]; break;
            case PH_CODE8: v = ((const uint8_t *)C.data)[r]; break;
            default: v = round_cents(((const int64_t *)C.data)[r], C.scale); break;
            }
            k = (unsigned long long)v ^ (1ull << 63);
            if (C.descending) k = ~k;
        } else {
            nullbit = 0;
        }
        keys[i] = k;
        rows[i] = (unsigned)r | (nullbit << 31);
    }
}

// OR of key ^ key[0] (bytes that vary) and of nullbit ^ nullbit[0]
__global__ __launch_bounds__(256) void sort_diff_kernel(const unsigned long long *__restrict__ keys,
                                                        const unsigned *__restrict__ rows, int64_t n,
                                                        unsigned long long *__restrict__ out) {
    const unsigned long long k0 = keys[0];
    const unsigned r0 = rows[0] >> 31;
    unsigned long long d = 0, dn = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        d |= keys[i] ^ k0;
        dn |= (rows[i] >> 31) ^ r0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        d |= __shfl_xor(d, o);
        dn |= __shfl_xor(dn, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (d) atomicOr(&out[0], d);
        if (dn) atomicOr(&out[1], dn);
    }
}

__device__ __forceinline__ int sort_digit(unsigned long long k, unsigned rw, int shift) {
    return shift < 64 ? (int)((k >> shift) & 0xFF) : (int)(rw >> 31);
}

__global__ __launch_bounds__(256) void sort_hist_kernel(const unsigned long long *__restrict__ keys,
                                                        const unsigned *__restrict__ rows, int64_t n, int shift,
                                                        int32_t *__restrict__ counts) {
    __shared__ int hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * SORT_TILE * SORT_TILES_PER_WG;
    const int64_t i1 = i0 + SORT_TILE * SORT_TILES_PER_WG < n ? i0 + SORT_TILE * SORT_TILES_PER_WG : n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) atomicAdd(&hist[sort_digit(keys[i], rows[i], shift)], 1);
    __syncthreads();
    counts[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

// Stable: a workgroup walks its chunk tile by tile in order; inside a tile a row's slot is the
// digit's cursor + rows of the same digit in earlier waves + rows of the same digit in lower
// lanes of its own wave (eight ballots find the lanes that share its digit).
__global__ __launch_bounds__(256) void sort_scatter_kernel(const unsigned long long *__restrict__ keys_in,
                                                           const unsigned *__restrict__ rows_in, int64_t n, int shift,
                                                           const int32_t *__restrict__ offsets,
                                                           unsigned long long *__restrict__ keys_out,
                                                           unsigned *__restrict__ rows_out) {
    __shared__ int cursor[256];
    __shared__ int wcount[4][256];
    cursor[threadIdx.x] = offsets[(int64_t)threadIdx.x * gridDim.x + blockIdx.x];
    for (int w = 0; w < 4; w++) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * SORT_TILE * SORT_TILES_PER_WG;
    for (int t = 0; t < SORT_TILES_PER_WG; t++) {
        const int64_t base = i0 + (int64_t)t * SORT_TILE;
        if (base >= n) break;   // workgroup-uniform
        const int64_t i = base + threadIdx.x;
        const bool live = i < n;
        unsigned long long k = 0;
        unsigned rw = 0;
        int d = 0;
        if (live) {
            k = keys_in[i];
            rw = rows_in[i];
            d = sort_digit(k, rw, shift);
        }
        // lanes of this wave with the same digit
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long has = __ballot(live && ((d >> b) & 1));
            same &= ((d >> b) & 1) ? has : ~has;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        if (live && rank == 0) wcount[wv][d] = __popcll(same);   // the digit's lowest lane
        __syncthreads();
        if (live) {
            int pos = cursor[d] + rank;
            for (int w = 0; w < wv; w++) pos += wcount[w][d];
            keys_out[pos] = k;
            rows_out[pos] = rw;
        }
        __syncthreads();
        const int add = wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] + wcount[3][threadIdx.x];
        cursor[threadIdx.x] += add;
        for (int w = 0; w < 4; w++) wcount[w][threadIdx.x] = 0;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sort_rows_out_kernel(const unsigned *__restrict__ rows, int64_t n, int32_t *__restrict__ perm) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        perm[i] = (int32_t)(rows[i] & 0x7FFFFFFFu);
}

__global__ __launch_bounds__(256) void sort_iota_kernel(const int32_t *__restrict__ sel, int64_t n, int32_t *__restrict__ perm) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        perm[i] = sel ? sel[i] : (int32_t)i;
}

}  // namespace ph

extern "C" int ph_sort_rows(ph_ctx *ctx, const ph_col *keys, const int32_t *descending, int32_t nkeys, const int32_t *sel,
                            int64_t n, int32_t *out_rows_dev) {
    PH_REQUIRE(ctx && keys && descending && nkeys >= 1 && nkeys <= 8 && n >= 0 && n < (1ll << 31) && (n == 0 || out_rows_dev),
               "ph_sort_rows: bad arguments (1..8 keys)");
    for (int c = 0; c < nkeys; c++) {
        const int t = keys[c].type;
        if (t != PH_I32 && t != PH_DATE && t != PH_CODE8 && t != PH_DEC64) {
            // the reference's RadixScatter has no case for BIGINT / DOUBLE keys either (sort_radix.go:257-321)
            ph::set_error("ph_sort_rows: key %d has type %d (INTEGER, DATE, DECIMAL and ordered dictionary codes sort on the device)", c, t);
            return PH_EUNSUPPORTED;
        }
    }
    if (n == 0) return PH_OK;
    hipStream_t st = ctx->stream;
    auto grid = [&](int64_t m) { return (int)std::min<int64_t>((m + 255) / 256, (int64_t)ctx->cu_count * 8); };
    const int64_t chunk = (int64_t)ph::SORT_TILE * ph::SORT_TILES_PER_WG;
    const int nwg = (int)((n + chunk - 1) / chunk);
    char *tmp = nullptr;
    const int64_t o_k1 = ph::round_up(n * 8, 16), o_r0 = 2 * o_k1, o_r1 = o_r0 + ph::round_up(n * 4, 16);
    const int64_t o_cnt = o_r1 + ph::round_up(n * 4, 16), o_misc = o_cnt + ph::round_up((int64_t)256 * nwg * 4, 16);
    PH_CHECK(ctx->pool_alloc(o_misc + 64, (void **)&tmp));
    unsigned long long *kbuf[2] = {(unsigned long long *)tmp, (unsigned long long *)(tmp + o_k1)};
    unsigned *rbuf[2] = {(unsigned *)(tmp + o_r0), (unsigned *)(tmp + o_r1)};
    int32_t *counts = (int32_t *)(tmp + o_cnt);
    unsigned long long *diff = (unsigned long long *)(tmp + o_misc);   // [0] key diff, [1] null diff
    int64_t *total = (int64_t *)(tmp + o_misc + 16);
    int rc = PH_OK;
    ph::sort_iota_kernel<<<grid(n), 256, 0, st>>>(sel, n, out_rows_dev);
    for (int c = nkeys - 1; c >= 0 && rc == PH_OK; c--) {   // LSD over the ORDER BY columns
        ph::SortCol C{keys[c].type, keys[c].scale, keys[c].data, keys[c].validity, descending[c] ? 1 : 0};
        int cur = 0;
        ph::sort_norm_kernel<<<grid(n), 256, 0, st>>>(C, out_rows_dev, n, kbuf[0], rbuf[0]);
        if (hipMemsetAsync(diff, 0, 16, st) != hipSuccess) { rc = PH_EHIP; break; }
        ph::sort_diff_kernel<<<grid(n), 256, 0, st>>>(kbuf[0], rbuf[0], n, diff);
        unsigned long long d[2] = {0, 0};
        if ((rc = ctx->download(d, diff, 16)) != PH_OK) break;
        for (int pass = 0; pass <= 8 && rc == PH_OK; pass++) {
            const int shift = pass * 8;   // pass 8 = the NULL byte
            const bool varies = pass < 8 ? ((d[0] >> shift) & 0xFF) != 0 : d[1] != 0;
            if (!varies) continue;
            ph::sort_hist_kernel<<<nwg, 256, 0, st>>>(kbuf[cur], rbuf[cur], n, shift, counts);
            rc = ph::exclusive_scan_i32(ctx, counts, (int64_t)256 * nwg, total);
            ph::sort_scatter_kernel<<<nwg, 256, 0, st>>>(kbuf[cur], rbuf[cur], n, shift, counts, kbuf[cur ^ 1], rbuf[cur ^ 1]);
            cur ^= 1;
        }
        ph::sort_rows_out_kernel<<<grid(n), 256, 0, st>>>(rbuf[cur], n, out_rows_dev);
    }
    if (rc == PH_OK && hipGetLastError() != hipSuccess) rc = PH_EHIP;
    ctx->pool_release(tmp);
    if (rc == PH_EHIP && ph_last_error()[0] == 0) ph::set_error("ph_sort_rows: HIP failure");
    return rc;
}
