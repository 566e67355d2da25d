// Fused Scan(filter) -> HashAggregate kernels for the resident-table mode (the measured mode).
//
// They replace the reference's per-2048-row pull loop  aggExecutor.Execute -> scanExecutor.Execute
// -> runFilterExec -> executeExprs -> GroupedAggrHashTable.AddChunk -> UpdateStates
// (pkg/compute/executor_aggr.go:110-142, executor_scan.go:144-241, expr_exec.go:85-486,
// aggregate_hash.go:136-391, function_aggr.go:1034-1161) by ONE pass over the device-resident
// columns. Both are HBM-bandwidth-bound integer kernels (no MFMA): every column byte is loaded
// exactly once with 16-byte-per-lane coalesced loads, predicates and decimal arithmetic run in
// registers on unscaled int64, and aggregation state never leaves the CU until the last tile.
//
//   filter_sumprod  (TPC-H Q6 shape): range predicates on <=3 columns, SUM(a*b) -> 1 group.
//       24 B/row algorithmic (shipdate 4 + discount 8 + quantity 4 + extendedprice 8).
//   lowcard_chain   (TPC-H Q1 shape): one range predicate, group key = dense index of two
//       dictionary-code columns (<= 8 live slots), accumulators
//       {Σq, Σe, Σe(A1+B1 d), Σe(A1+B1 d)(A2+B2 t), Σd, count}.
//       34 B/row algorithmic (qty 4 + ext/disc/tax 3x8 + 2 code bytes + shipdate 4).
//       Group state is a per-thread-private column of LDS (ds_add_u64, conflict-free because
//       consecutive lanes hit consecutive banks), merged once per workgroup at the end.
#include <mutex>

#include "common.h"
#include "scan_kernels.h"

namespace ph {

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ long long wave_sum_i64(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct alignas(16) i64x2 { long long x, y; };
struct alignas(16) i32x4 { int x, y, z, w; };

typedef long long v2i64 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// Streaming loads: every byte of these columns is read exactly once per launch, so the loads are
// marked non-temporal (global_load_dwordx4 ... nt) to keep them from displacing anything useful in
// L2/MALL. NT = false keeps the default cache policy (A/B switch: PH_SCAN_NT=0).
template <bool NT> __device__ __forceinline__ i64x2 ld_i64x2(const int64_t *p) {
    if (NT) {
        v2i64 v = __builtin_nontemporal_load(reinterpret_cast<const v2i64 *>(p));
        return i64x2{v.x, v.y};
    }
    return *reinterpret_cast<const i64x2 *>(p);
}
template <bool NT> __device__ __forceinline__ i32x4 ld_i32x4(const int32_t *p) {
    if (NT) {
        v4i32 v = __builtin_nontemporal_load(reinterpret_cast<const v4i32 *>(p));
        return i32x4{v.x, v.y, v.z, v.w};
    }
    return *reinterpret_cast<const i32x4 *>(p);
}
template <bool NT> __device__ __forceinline__ unsigned ld_u32(const uint8_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p));
    return *reinterpret_cast<const unsigned *>(p);
}

// ------------------------------------------------------------------ fused merge + publish (ScanTail)
// one partial word of this workgroup, stored where the other XCDs can read it (a plain store would stay in this XCD's write-back L2 until a fence)
__device__ __forceinline__ void scan_tail_put(long long *partials, int nacc, int j, long long v) {
    __hip_atomic_store(partials + (int64_t)blockIdx.x * nacc + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL threads of every workgroup (256 threads) after its scan_tail_put calls; lds: 2 * SCAN_TAIL_MAX_ACC + 1 words nothing else uses any more.
// A 64-bit partial is folded as two halves (a += low 32 bits, b += high 32 bits signed: no carries between LDS atomics), first-row minima as the
// maximum of the complement (zero is every word's identity).
__device__ __forceinline__ void scan_tail(const long long *partials, const ScanTail &T, unsigned long long *lds) {
    // Everything the workgroups tell each other travels in device-scope stores / atomics (performed at the device's coherent level, not in an XCD's
    // L2), so ordering is all that is needed: s_waitcnt returns when this wave's stores have been acknowledged, the barrier collects the waves, then
    // the ticket. A device-scope fence here (__threadfence: an L2 write-back + invalidate per workgroup) cost 36-50 us per launch.
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    int *s_flag = reinterpret_cast<int *>(lds + 2 * SCAN_TAIL_MAX_ACC);
    if (threadIdx.x == 0) *s_flag = __hip_atomic_fetch_add(T.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    for (int j = threadIdx.x; j < 2 * SCAN_TAIL_MAX_ACC; j += 256) lds[j] = 0;
    __syncthreads();
    if (!*s_flag) return;
    unsigned long long *la = lds, *lb = lds + SCAN_TAIL_MAX_ACC;
    const int total = (int)gridDim.x * T.nacc;
    for (int i0 = 0; i0 < total; i0 += 256 * 16) {
        long long v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = i0 + k * 256 + (int)threadIdx.x;
            v[k] = i < total ? __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = i0 + k * 256 + (int)threadIdx.x;
            if (i >= total) break;
            const int j = i % T.nacc;
            if (T.min_stride > 0 && j % T.min_stride == T.min_stride - 1) atomicMax(la + j, ~(unsigned long long)v[k]);
            else {
                atomicAdd(la + j, (unsigned long long)v[k] & 0xffffffffull);
                atomicAdd(lb + j, (unsigned long long)(v[k] >> 32));   // arithmetic shift: the signed high half
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < T.nacc; j += 256) {
        const unsigned long long a = la[j];
        const long long b = (long long)lb[j];
        unsigned long long lo;
        long long hi;
        if (T.min_stride > 0 && j % T.min_stride == T.min_stride - 1) {
            lo = ~a;
            hi = 0;
        } else {   // a + (b << 32) in 128 bits: a < 2^63 (2^31 workgroups x 2^32), b a signed sum of signed halves
            lo = a + ((unsigned long long)b << 32);
            hi = (b >> 32) + (lo < a ? 1 : 0);
        }
        T.out_lo[j] = lo;
        T.out_hi[j] = hi;
        if (T.mbox) {
            T.mbox[j] = lo;
            T.mbox[T.nacc + j] = (unsigned long long)hi;
        }
    }
    if (T.mbox) __threadfence_system();   // the mailbox words are visible to the host ...
    __syncthreads();                      // ... for every wave ...
    if (threadIdx.x == 0) {
        *T.done = 0;
        if (T.mbox) __hip_atomic_store(T.flag, T.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the number is
    }
}

// ------------------------------------------------------------------ filter_sumprod (Q6 shape)

struct FsTile {
    i32x4 p0;      // int32 predicate column (dates)
    i32x4 p2;      // int32 predicate column (quantity)
    i64x2 b0, b1;  // int64 predicate column that is also the second factor (discount)
    i64x2 a0, a1;  // int64 first factor (extendedprice)
};

template <bool NT> __device__ __forceinline__ FsTile fs_load(const FilterSumProdParams &P, int64_t row) {
    FsTile t;
    t.p0 = ld_i32x4<NT>(P.p0 + row);
    t.p2 = ld_i32x4<NT>(P.p2 + row);
    t.b0 = ld_i64x2<NT>(P.b + row);
    t.b1 = ld_i64x2<NT>(P.b + row + 2);
    t.a0 = ld_i64x2<NT>(P.a + row);
    t.a1 = ld_i64x2<NT>(P.a + row + 2);
    return t;
}

__device__ __forceinline__ void fs_row(const FilterSumProdParams &P, bool in_range, int p0, int p2,
                                       long long b, long long a, long long &sum, unsigned &cnt) {
    bool pass = in_range && p0 >= P.p0_lo && p0 <= P.p0_hi && p2 >= P.p2_lo && p2 <= P.p2_hi &&
                b >= P.b_lo && b <= P.b_hi;
    if (pass) {
        sum += a * b;
        cnt += 1;
    }
}

template <bool NT> __global__ __launch_bounds__(256) void filter_sumprod_kernel(FilterSumProdParams P) {
    // tile = 1024 rows per workgroup iteration, 4 consecutive rows per lane
    const int64_t tile_rows = 1024;
    long long sum = 0;
    unsigned cnt = 0;
    int64_t first = P.row_begin + (int64_t)blockIdx.x * tile_rows + threadIdx.x * 4;
    const int64_t stride = (int64_t)gridDim.x * tile_rows;
    // row_begin is a multiple of 4 and columns are padded to PH_ROW_PAD rows, so a 4-row vector
    // load that starts below row_end stays inside the allocation.
    if (first < P.row_end) {
        FsTile cur = fs_load<NT>(P, first);
        for (int64_t row = first; row < P.row_end; row += stride) {
            int64_t nrow = row + stride;
            FsTile nxt = cur;
            if (nrow < P.row_end) nxt = fs_load<NT>(P, nrow);
            fs_row(P, row + 0 < P.row_end, cur.p0.x, cur.p2.x, cur.b0.x, cur.a0.x, sum, cnt);
            fs_row(P, row + 1 < P.row_end, cur.p0.y, cur.p2.y, cur.b0.y, cur.a0.y, sum, cnt);
            fs_row(P, row + 2 < P.row_end, cur.p0.z, cur.p2.z, cur.b1.x, cur.a1.x, sum, cnt);
            fs_row(P, row + 3 < P.row_end, cur.p0.w, cur.p2.w, cur.b1.y, cur.a1.y, sum, cnt);
            cur = nxt;
        }
    }
    sum = wave_sum_i64(sum);
    long long c64 = wave_sum_i64((long long)cnt);
    __shared__ long long ws[2][4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        ws[0][w] = sum;
        ws[1][w] = c64;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long s = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3], c = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
        if (P.tail.done) {
            scan_tail_put(P.partials, 2, 0, s);
            scan_tail_put(P.partials, 2, 1, c);
        } else {
            P.partials[(int64_t)blockIdx.x * 2 + 0] = s;
            P.partials[(int64_t)blockIdx.x * 2 + 1] = c;
        }
    }
    __shared__ unsigned long long tail_lds[2 * SCAN_TAIL_MAX_ACC + 1];
    if (P.tail.done) scan_tail(P.partials, P.tail, tail_lds);
}

// ------------------------------------------------------------------ lowcard_chain (Q1 shape)

struct LcTile {
    i32x4 p;          // predicate column (shipdate)
    i32x4 q;          // int32 summed column (quantity)
    i64x2 e0, e1;     // extendedprice
    i64x2 d0, d1;     // discount
    i64x2 t0, t1;     // tax
    unsigned k0, k1;  // 4 code bytes each
};

template <bool NT> __device__ __forceinline__ LcTile lc_load(const LowcardChainParams &P, int64_t row) {
    LcTile t;
    t.p = ld_i32x4<NT>(P.p + row);
    t.q = ld_i32x4<NT>(P.q + row);
    t.e0 = ld_i64x2<NT>(P.e + row);
    t.e1 = ld_i64x2<NT>(P.e + row + 2);
    t.d0 = ld_i64x2<NT>(P.d + row);
    t.d1 = ld_i64x2<NT>(P.d + row + 2);
    t.t0 = ld_i64x2<NT>(P.t + row);
    t.t1 = ld_i64x2<NT>(P.t + row + 2);
    t.k0 = ld_u32<NT>(P.k0 + row);
    t.k1 = ld_u32<NT>(P.k1 + row);
    return t;
}

// LDS layout per workgroup (nslots group slots, 256 threads):
//   u64 acc64[slot][5][256]   Σq, Σe, Σe·f1, Σe·f1·f2, Σd      (one 8-byte column per thread)
//   u32 acc32[slot][2][256]   count, first row id (min)
// Every thread owns one column, so the 64 lanes of a wave always touch 64 consecutive words
// whatever slot each lane is in (slot strides are multiples of 256 words): no bank conflicts,
// no inter-lane contention, plain ds_add/ds_min without return.
__device__ __forceinline__ void lc_row(const LowcardChainParams &P, unsigned long long *acc64,
                                       unsigned *acc32, bool in_range, unsigned row, int p, int q,
                                       long long e, long long d, long long t, unsigned k0,
                                       unsigned k1) {
    if (in_range && p >= P.p_lo && p <= P.p_hi) {
        unsigned slot = k0 * (unsigned)P.nk1 + k1;
        unsigned long long *a = acc64 + (size_t)slot * 5 * 256;
        unsigned *c = acc32 + (size_t)slot * 2 * 256;
        long long dp = e * (P.A1 + P.B1 * d);
        long long ch = dp * (P.A2 + P.B2 * t);
        atomicAdd(a + 0 * 256, (unsigned long long)(long long)q);
        atomicAdd(a + 1 * 256, (unsigned long long)e);
        atomicAdd(a + 2 * 256, (unsigned long long)dp);
        atomicAdd(a + 3 * 256, (unsigned long long)ch);
        atomicAdd(a + 4 * 256, (unsigned long long)d);
        atomicAdd(c + 0 * 256, 1u);
        atomicMin(c + 1 * 256, row);
    }
}

template <bool NT, int U> __global__ __launch_bounds__(256) void lowcard_chain_kernel(LowcardChainParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_acc[];
    const int ns = P.nslots;
    unsigned long long *lds64 = lds_acc;
    unsigned *lds32 = reinterpret_cast<unsigned *>(lds_acc + (size_t)ns * 5 * 256);
    for (int i = threadIdx.x; i < ns * 5 * 256; i += 256) lds64[i] = 0;
    for (int s = 0; s < ns; s++) {
        lds32[(s * 2 + 0) * 256 + threadIdx.x] = 0;
        lds32[(s * 2 + 1) * 256 + threadIdx.x] = 0xffffffffu;
    }
    __syncthreads();
    unsigned long long *acc64 = lds64 + threadIdx.x;
    unsigned *acc32 = lds32 + threadIdx.x;

    // U tiles of 1024 rows per iteration (a lane owns 4 consecutive rows of each), register
    // double buffered: the loads of iteration i+1 are in flight while iteration i is aggregated.
    const int64_t tile_rows = 1024;
    int64_t first = P.row_begin + (int64_t)blockIdx.x * tile_rows * U + threadIdx.x * 4;
    const int64_t stride = (int64_t)gridDim.x * tile_rows * U;
    if (first < P.row_end) {
        LcTile cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (first + u * tile_rows < P.row_end) cur[u] = lc_load<NT>(P, first + u * tile_rows);
        for (int64_t row0 = first; row0 < P.row_end; row0 += stride) {
            int64_t nrow = row0 + stride;
#pragma unroll
            for (int u = 0; u < U; u++) {
                nxt[u] = cur[u];
                if (nrow + u * tile_rows < P.row_end) nxt[u] = lc_load<NT>(P, nrow + u * tile_rows);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                int64_t row = row0 + u * tile_rows;
                if (row >= P.row_end) break;
                const LcTile &c = cur[u];
                unsigned r = (unsigned)row;
                lc_row(P, acc64, acc32, row + 0 < P.row_end, r + 0, c.p.x, c.q.x, c.e0.x, c.d0.x, c.t0.x,
                       c.k0 & 0xff, c.k1 & 0xff);
                lc_row(P, acc64, acc32, row + 1 < P.row_end, r + 1, c.p.y, c.q.y, c.e0.y, c.d0.y, c.t0.y,
                       (c.k0 >> 8) & 0xff, (c.k1 >> 8) & 0xff);
                lc_row(P, acc64, acc32, row + 2 < P.row_end, r + 2, c.p.z, c.q.z, c.e1.x, c.d1.x, c.t1.x,
                       (c.k0 >> 16) & 0xff, (c.k1 >> 16) & 0xff);
                lc_row(P, acc64, acc32, row + 3 < P.row_end, r + 3, c.p.w, c.q.w, c.e1.y, c.d1.y, c.t1.y,
                       c.k0 >> 24, c.k1 >> 24);
            }
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = nxt[u];
        }
    }
    __syncthreads();
    // workgroup merge: wave w reduces accumulator rows w, w+4, ...; partial layout
    // [slot][LC_NACC+1] = {Σq, Σe, Σe·f1, Σe·f1·f2, Σd, count, first_row}
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_slot = LC_NACC + 1;
    for (int j = w; j < ns * per_slot; j += 4) {
        int s = j / per_slot, a = j % per_slot;
        long long v;
        if (a < 5) {
            const unsigned long long *r = lds64 + (size_t)(s * 5 + a) * 256;
            v = (long long)(r[lane] + r[lane + 64] + r[lane + 128] + r[lane + 192]);
            v = wave_sum_i64(v);
        } else if (a == 5) {
            const unsigned *r = lds32 + (size_t)(s * 2 + 0) * 256;
            v = (long long)r[lane] + r[lane + 64] + r[lane + 128] + r[lane + 192];
            v = wave_sum_i64(v);
        } else {
            const unsigned *r = lds32 + (size_t)(s * 2 + 1) * 256;
            unsigned m = min(min(r[lane], r[lane + 64]), min(r[lane + 128], r[lane + 192]));
            for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
            v = (long long)m;
        }
        if (lane == 0) {
            if (P.tail.done) scan_tail_put(P.partials, ns * per_slot, j, v);
            else P.partials[(int64_t)blockIdx.x * ns * per_slot + j] = v;
        }
    }
    if (P.tail.done) {
        __syncthreads();   // the LDS accumulators have been read by every wave: their first 2 KiB serve the tail
        scan_tail(P.partials, P.tail, lds_acc);
    }
}

// The end of a merge wave (one wave per accumulator word j; lane 0 holds the merged word): store it, and — when the merge is to publish (T.done) — take
// a ticket; the wave that finishes last copies all words into the mapped mailbox and stores the sequence number (what publish_kernel does as one more
// launch). Every host-visible store comes from that one wave, in publish_kernel's order: words, system fence, number.
__device__ __forceinline__ void merge_finish(const ScanTail &T, int j, unsigned long long lo, long long hi, unsigned long long *out_lo, long long *out_hi) {
    const int lane = threadIdx.x;
    if (!T.done) {
        if (lane == 0) {
            out_lo[j] = lo;
            out_hi[j] = hi;
        }
        return;
    }
    unsigned ticket = 0;
    if (lane == 0) {
        __hip_atomic_store(out_lo + j, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // where the other XCDs can read them (not this XCD's write-back L2)
        __hip_atomic_store(out_hi + j, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);   // acknowledged before the ticket
        ticket = __hip_atomic_fetch_add(T.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = (unsigned)__shfl((int)ticket, 0);
    if (ticket != gridDim.x - 1) return;
    for (int k = lane; k < T.nacc; k += 64) {
        T.mbox[k] = __hip_atomic_load(out_lo + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T.mbox[T.nacc + k] = (unsigned long long)__hip_atomic_load(out_hi + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence_system();
    if (lane == 0) {
        *T.done = 0;
        __hip_atomic_store(T.flag, T.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------ partial merge
// out[j] = 128-bit sum over blocks of int64 partials[b*nacc + j]. One wave per accumulator.
// When min_stride > 0, accumulators with j % min_stride == min_stride-1 are minima (first row ids).
__global__ __launch_bounds__(64) void merge_partials_kernel(const long long *__restrict__ partials,
                                                            int nblocks, int nacc, int min_stride,
                                                            unsigned long long *__restrict__ out_lo,
                                                            long long *__restrict__ out_hi, ScanTail T) {
    int j = blockIdx.x;
    int lane = threadIdx.x;
    if (min_stride > 0 && j % min_stride == min_stride - 1) {
        long long m = INT64_MAX;
        for (int b = lane; b < nblocks; b += 64) {
            long long v = partials[(int64_t)b * nacc + j];
            m = v < m ? v : m;
        }
        for (int o = 32; o > 0; o >>= 1) {
            long long v = __shfl_xor(m, o);
            m = v < m ? v : m;
        }
        merge_finish(T, j, (unsigned long long)m, 0, out_lo, out_hi);
        return;
    }
    // accumulate positives and negatives as unsigned magnitudes to keep carries simple
    unsigned long long lo = 0;
    long long hi = 0;
    for (int b = lane; b < nblocks; b += 64) {
        long long v = partials[(int64_t)b * nacc + j];
        unsigned long long nlo = lo + (unsigned long long)v;
        hi += (nlo < lo ? 1 : 0) + (v < 0 ? -1 : 0);
        lo = nlo;
    }
    // tree-combine 128-bit lane values
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long olo = __shfl_xor(lo, o);
        long long ohi = __shfl_xor(hi, o);
        unsigned long long nlo = lo + olo;
        hi = hi + ohi + (nlo < lo ? 1 : 0);
        lo = nlo;
    }
    merge_finish(T, j, lo, hi, out_lo, out_hi);
}

// out[j] over blocks with a per-word operation: op = (opmask >> 2*(j % stride)) & 3 — 0: 128-bit sum,
// 1: signed min, 2: signed max (MIN/MAX accumulators and first-row ids of the generated kernels)
__global__ __launch_bounds__(64) void merge_partials_ops_kernel(const long long *__restrict__ partials,
                                                                int nblocks, int nacc, int stride, unsigned long long opmask,
                                                                unsigned long long *__restrict__ out_lo,
                                                                long long *__restrict__ out_hi, ScanTail T) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int op = (int)((opmask >> (2 * (j % stride))) & 3);
    if (op != 0) {
        long long m = op == 1 ? INT64_MAX : INT64_MIN;
        for (int b = lane; b < nblocks; b += 64) {
            long long v = partials[(int64_t)b * nacc + j];
            m = op == 1 ? (v < m ? v : m) : (v > m ? v : m);
        }
        for (int o = 32; o > 0; o >>= 1) {
            long long v = __shfl_xor(m, o);
            m = op == 1 ? (v < m ? v : m) : (v > m ? v : m);
        }
        merge_finish(T, j, (unsigned long long)m, m < 0 ? -1 : 0, out_lo, out_hi);
        return;
    }
    unsigned long long lo = 0;
    long long hi = 0;
    for (int b = lane; b < nblocks; b += 64) {
        long long v = partials[(int64_t)b * nacc + j];
        unsigned long long nlo = lo + (unsigned long long)v;
        hi += (nlo < lo ? 1 : 0) + (v < 0 ? -1 : 0);
        lo = nlo;
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long olo = __shfl_xor(lo, o);
        long long ohi = __shfl_xor(hi, o);
        unsigned long long nlo = lo + olo;
        hi = hi + ohi + (nlo < lo ? 1 : 0);
        lo = nlo;
    }
    merge_finish(T, j, lo, hi, out_lo, out_hi);
}

static bool scan_nt() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("PH_SCAN_NT");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

int launch_filter_sumprod(ph_ctx *ctx, const FilterSumProdParams &P, int grid) {
    if (scan_nt()) filter_sumprod_kernel<true><<<grid, 256, 0, ctx->stream>>>(P);
    else filter_sumprod_kernel<false><<<grid, 256, 0, ctx->stream>>>(P);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

int launch_lowcard_chain(ph_ctx *ctx, const LowcardChainParams &P, int grid) {
    size_t lds = (size_t)P.nslots * (5 * 256 * sizeof(unsigned long long) + 2 * 256 * sizeof(unsigned));
    // the attribute belongs to the device's copy of the function: once per device, under a lock
    // (several ctxs / threads may launch concurrently)
    static std::mutex mu;
    static bool attr_set_dev[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    bool &attr_set = attr_set_dev[ctx->device & 63];
    if (!attr_set) {
        PH_HIP(hipFuncSetAttribute((const void *)lowcard_chain_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PH_HIP(hipFuncSetAttribute((const void *)lowcard_chain_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PH_HIP(hipFuncSetAttribute((const void *)lowcard_chain_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PH_HIP(hipFuncSetAttribute((const void *)lowcard_chain_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int u = 1;
    if (const char *e = getenv("PH_SCAN_UNROLL")) u = atoi(e);
    if (!scan_nt()) lowcard_chain_kernel<false, 1><<<grid, 256, lds, ctx->stream>>>(P);
    else if (u == 3) lowcard_chain_kernel<true, 3><<<grid, 256, lds, ctx->stream>>>(P);
    else if (u == 2) lowcard_chain_kernel<true, 2><<<grid, 256, lds, ctx->stream>>>(P);
    else lowcard_chain_kernel<true, 1><<<grid, 256, lds, ctx->stream>>>(P);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

int launch_merge_partials(ph_ctx *ctx, const long long *partials, int nblocks, int nacc,
                          int min_stride, unsigned long long *out_lo, long long *out_hi, const ScanTail *publish) {
    ScanTail T = {};
    if (publish && publish->mbox) T = *publish;
    merge_partials_kernel<<<nacc, 64, 0, ctx->stream>>>(partials, nblocks, nacc, min_stride, out_lo,
                                                        out_hi, T);
    PH_HIP(hipGetLastError());
    return PH_OK;
}


int launch_merge_partials_ops(ph_ctx *ctx, const long long *partials, int nblocks, int nacc, int stride,
                              unsigned long long opmask, unsigned long long *out_lo, long long *out_hi, const ScanTail *publish) {
    ScanTail T = {};
    if (publish && publish->mbox) T = *publish;
    merge_partials_ops_kernel<<<nacc, 64, 0, ctx->stream>>>(partials, nblocks, nacc, stride, opmask, out_lo, out_hi, T);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

}  // namespace ph
