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

#include <chrono>
#include <sstream>

#include "agg_sink_src.h"
#include "common.h"
#include "device_util.h"
#include "ops.h"
#include "scan_jit.h"

namespace ph {

}  // namespace ph
#include "agg_sink.inc"
namespace ph {

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

// Workgroup-wide radix select over n keys held in LDS (8 passes, most significant byte first):
// returns the k-th smallest key, or ~0 with *all = true when there are fewer than k keys.
constexpr int TOPK_CHUNK = 1024;   // keys one workgroup selects from: small chunks = many workgroups selecting in parallel, each over 4 keys per thread
constexpr int TOPK_FINAL = 4096;   // candidates the last workgroup selects from in LDS (32 KiB); more: from memory

__device__ unsigned long long wg_radix_select(const unsigned long long *keys, int n, long long k, unsigned *lh,
                                              unsigned long long *s_state, bool *all) {
    unsigned long long prefix = 0;
    long long rem = k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long *wtot = reinterpret_cast<long long *>(s_state + 2);
    *all = false;
    if (n < k) { *all = true; return ~0ull; }   // uniform over the workgroup
    // bytes no key differs in need no pass: all keys would land in ONE histogram bin, i.e. n
    // same-address LDS atomics (aggregate values rarely use the top bytes: 4 of 8 passes for Q3)
    unsigned long long diff = 0;
    const unsigned long long k0 = keys[0];
    for (int i = threadIdx.x; i < n; i += 256) diff |= keys[i] ^ k0;
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o);
    if (lane == 0) s_state[4 + wv] = diff;
    __syncthreads();
    diff = s_state[4] | s_state[5] | s_state[6] | s_state[7];
    __syncthreads();
    for (int pass = 7; pass >= 0; pass--) {
        if (((diff >> (8 * pass)) & 0xff) == 0) { prefix |= k0 & (0xffull << (8 * pass)); continue; }
        lh[threadIdx.x] = 0;
        __syncthreads();
        const unsigned long long mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
        for (int i = threadIdx.x; i < n; i += 256) {
            const unsigned long long key = keys[i];
            if ((key & mask) == (prefix & mask)) atomicAdd(&lh[(key >> (8 * pass)) & 0xff], 1u);
        }
        __syncthreads();
        const unsigned v = lh[threadIdx.x];
        long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            long long y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        for (int w = 0; w < wv; w++) incl += wtot[w];
        if (incl >= rem && incl - (long long)v < rem) {
            s_state[0] = prefix | ((unsigned long long)threadIdx.x << (8 * pass));
            s_state[1] = (unsigned long long)(rem - (incl - (long long)v));
        } else if (threadIdx.x == 255 && incl < rem) {   // fewer than k keys: everything qualifies
            s_state[0] = ~0ull;
            s_state[1] = 0;
        }
        __syncthreads();
        prefix = s_state[0];
        rem = (long long)s_state[1];
        if (prefix == ~0ull && rem == 0 && pass == 7) { *all = true; return ~0ull; }
    }
    return prefix;
}

// ONE launch: every workgroup selects the k best of its chunk of <= TOPK_CHUNK groups in LDS and
// appends them (all ties of its k-th value included) to a candidate list; the last workgroup to
// finish (a ticket counter) selects the k best of the candidates the same way and writes the ids
// of the groups at least that good. The group count is read on the device. Eight launches with
// grid-wide histograms took 77 us for Q3's 113 k groups (10 % of the query's kernel time).
// meta[0] = qualifying groups, meta[1] = "a sum does not fit int64" flag
__global__ __launch_bounds__(256) void topk_select_kernel(const unsigned long long *__restrict__ sum_lo,
                                                          const long long *__restrict__ sum_hi,
                                                          const unsigned long long *__restrict__ cnt, int naggs, int a,
                                                          const int *__restrict__ ngroups, int descending, int is_sum, long long k,
                                                          int *__restrict__ cand_ids, unsigned long long *__restrict__ cand_keys,
                                                          int *__restrict__ cand_count, int *__restrict__ done,
                                                          int *__restrict__ out_ids, int *__restrict__ meta, int cap,
                                                          unsigned long long *__restrict__ wg_kth) {
    __shared__ unsigned long long skeys[TOPK_FINAL];
    __shared__ unsigned lh[256];
    __shared__ unsigned long long s_state[8];
    __shared__ int s_last;
    const int ng = *ngroups;
    const int base = blockIdx.x * TOPK_CHUNK;
    const int n = ng - base < TOPK_CHUNK ? (ng - base > 0 ? ng - base : 0) : TOPK_CHUNK;
    bool wide = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int64_t st = (int64_t)(base + i) * naggs + a;
        const bool live = cnt[st] != 0;
        const long long lo = is_sum == 2 ? (long long)cnt[st] : (long long)sum_lo[st];   // 2: COUNT / COUNT(*) rank by their count word
        if (is_sum == 1 && live && sum_hi[st] != (lo >> 63)) wide = true;
        // an aggregate no input ever reached is NULL, and NULLs sort first whatever the direction
        // (sort_layout.go:46): the best possible key
        skeys[i] = live ? order_key(lo, descending) : 0ull;
    }
    if (wide) atomicOr(meta + 1, 1);
    __syncthreads();
    if (n > 0) {
        bool all;
        const unsigned long long kth = wg_radix_select(skeys, n, k, lh, s_state, &all);
        // this workgroup's k-th best key bounds the GLOBAL k-th best from above (it holds k keys at least that good): the last workgroup prunes
        // the candidate list with the smallest of these bounds before it selects
        if (threadIdx.x == 0) __hip_atomic_store(&wg_kth[blockIdx.x], all ? ~0ull : kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ONE reserving add on the global candidate counter per workgroup (its candidates are counted in LDS first): adds on one
        // address execute one after the other at the memory side (~12 ns each), and 111 workgroups x ~11 candidates were 14 of
        // this kernel's 33 us
        __shared__ int s_ncand, s_cbase;
        if (threadIdx.x == 0) s_ncand = 0;
        __syncthreads();
        int mypos[TOPK_CHUNK / 256];
#pragma unroll
        for (int q = 0; q < TOPK_CHUNK / 256; q++) {
            const int i = q * 256 + threadIdx.x;
            mypos[q] = (i < n && (all || skeys[i] <= kth)) ? atomicAdd(&s_ncand, 1) : -1;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cbase = s_ncand ? atomicAdd(cand_count, s_ncand) : 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TOPK_CHUNK / 256; q++) {
            if (mypos[q] < 0) continue;
            const int i = q * 256 + threadIdx.x, pos = s_cbase + mypos[q];
            // agent-scope stores: written through to the device's coherence point, where the last workgroup reads them.
            // (Ordinary stores + __threadfence() made every workgroup write back its whole L2 — including the group
            // states the previous kernel had just written: 39 us for a kernel whose own work is ~10.)
            __hip_atomic_store(&cand_ids[pos], base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one entry per group at most
            __hip_atomic_store(&cand_keys[pos], skeys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    else if (threadIdx.x == 0) __hip_atomic_store(&wg_kth[blockIdx.x], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a workgroup past the groups: no bound)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's candidate stores are performed
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    // final selection over the candidates (read at the coherence point: other workgroups wrote them)
    const int m = __hip_atomic_load(cand_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long kth = ~0ull;
    bool all = k >= m;
    if (!all && m > TOPK_FINAL) {
        // Many workgroups (Q10: 560 x ~20 candidates): only candidates at least as good as the smallest of the workgroups' own k-th keys can be
        // among the global k best — usually a few dozen. They are selected from in LDS; the select straight from memory over all 11 k
        // candidates was 50 of this kernel's 82 us.
        unsigned long long bound = ~0ull;
        for (int w = threadIdx.x; w < (int)gridDim.x; w += 256) {
            const unsigned long long b = __hip_atomic_load(&wg_kth[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bound = b < bound ? b : bound;
        }
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(bound, o); bound = y < bound ? y : bound; }
        if ((threadIdx.x & 63) == 0) s_state[4 + (threadIdx.x >> 6)] = bound;
        __shared__ int s_kept;
        if (threadIdx.x == 0) s_kept = 0;
        __syncthreads();
        for (int w = 0; w < 4; w++) bound = s_state[4 + w] < bound ? s_state[4 + w] : bound;
        __syncthreads();
        for (int i0 = 0; i0 < m; i0 += 256 * 8) {   // (eight loads in flight per thread: one after the other they were most of this stage)
            unsigned long long key[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * 256 + (int)threadIdx.x;
                key[u] = i < m ? __hip_atomic_load(&cand_keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * 256 + (int)threadIdx.x;
                if (i < m && key[u] <= bound) { const int pos = atomicAdd(&s_kept, 1); if (pos < TOPK_FINAL) skeys[pos] = key[u]; }
            }
        }
        __syncthreads();
        const int kept = s_kept;
        if (kept <= TOPK_FINAL && kept >= k) { kth = wg_radix_select(skeys, kept, k, lh, s_state, &all); }
        else kth = wg_radix_select(cand_keys, m, k, lh, s_state, &all);   // (ties by the thousand: the select straight from memory)
    } else if (!all) {
        if (m <= TOPK_FINAL) {
            for (int i = threadIdx.x; i < m; i += 256)
                skeys[i] = __hip_atomic_load(&cand_keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            kth = wg_radix_select(skeys, m, k, lh, s_state, &all);
        } else {
            // more candidates than LDS holds (a huge k, or one value shared by thousands of groups):
            // the same select straight from memory
            kth = wg_radix_select(cand_keys, m, k, lh, s_state, &all);
        }
    }
    for (int i0 = 0; i0 < m; i0 += 256 * 8) {
        unsigned long long key[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 256 + (int)threadIdx.x;
            key[u] = i < m ? __hip_atomic_load(&cand_keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 256 + (int)threadIdx.x;
            if (i < m && (all || key[u] <= kth)) {
                const int pos = atomicAdd(meta, 1);
                if (pos < cap) out_ids[pos] = __hip_atomic_load(&cand_ids[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Records of the selected groups, contiguous, behind a 4-int header {count, sum-too-wide flag,
// group count, table error flag}: header and the first records come back in ONE copy.
// record = [first_row][null mask][keys nkeys][sum_lo naggs][sum_hi naggs][count naggs] (8-byte words)
__global__ __launch_bounds__(256) void agg_pack_kernel(const int *__restrict__ ids, const int *__restrict__ meta,
                                                       const int *__restrict__ counters, int cap, int nkeys, int naggs,
                                                       const long long *first_row, const unsigned long long *gkeys,
                                                       const unsigned *gnull, const unsigned long long *sum_lo,
                                                       const long long *sum_hi, const unsigned long long *cnt,
                                                       unsigned long long *__restrict__ out) {
    const int n = meta[0] < cap ? meta[0] : cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int *h = reinterpret_cast<int *>(out);
        h[0] = meta[0]; h[1] = meta[1]; h[2] = counters[0]; h[3] = counters[1];
    }
    const int rec = 2 + nkeys + 3 * naggs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int g = ids ? ids[i] : i;   // no id list: groups 0..n-1 (ph_agg_finalize)
        unsigned long long *o = out + 2 + (int64_t)i * rec;
        o[0] = (unsigned long long)first_row[g];
        o[1] = gnull[g];
        for (int c = 0; c < nkeys; c++) o[2 + c] = gkeys[(int64_t)g * nkeys + c];
        for (int a = 0; a < naggs; a++) {
            o[2 + nkeys + a] = sum_lo[(int64_t)g * naggs + a];
            o[2 + nkeys + naggs + a] = (unsigned long long)sum_hi[(int64_t)g * naggs + a];
            o[2 + nkeys + 2 * naggs + a] = cnt[(int64_t)g * naggs + a];
        }
    }
}

// ------------------------------------------------------------------ bulk build of an empty table
// First sink into an EMPTY table that the caller expects to hold many groups (Q3: 113 k groups from
// 298 k rows). Creating a group in the global table costs ~9 coherent memory operations, and
// scattered device atomics run at ~20 G/s on this part, so such inputs were bound by creation.
// Here the rows are first grouped by bits of their key hash (count -> scan -> scatter of the key
// words, row id and argument values: plain stores), then ONE workgroup per partition aggregates
// its rows in a private LDS hash table (all rows of a key are in one partition) and writes every
// finished group once: group ids reserved with one counter add per workgroup, state arrays with
// plain stores, the slot claimed with one CAS (all keys are distinct, so no key is compared).
// Rows that find no room in LDS go through find_or_create as in the ordinary sink.
template <int NK>
__device__ __forceinline__ void bulk_row_keys(const AggSinkParams &S, int64_t r, unsigned long long *k, unsigned *nullmask) {
    *nullmask = 0;
#pragma unroll
    for (int c = 0; c < NK; c++) {
        k[c] = 0;
        if (bit_valid(S.key[c].validity, r)) k[c] = load_key(S.key[c], r);
        else *nullmask |= 1u << c;
    }
}

// BU rows per thread with their key reads issued together: keys without NULLs (PLAINK) sit behind
// one wave-uniform branch per column on its width instead of a per-row type switch.
constexpr int BU = 4;

template <int NK, bool PLAINK, int TPB = 256>
__device__ __forceinline__ void bulk_keys_batch(const AggSinkParams &S, int64_t base, int64_t i1, int64_t (&ii)[BU], int64_t (&rr)[BU],
                                                unsigned long long (&k)[BU][AGG_MAX_KEYS], unsigned (&nm)[BU]) {
#pragma unroll
    for (int u = 0; u < BU; u++) {
        ii[u] = base + u * TPB + threadIdx.x;
        const int64_t ic = ii[u] < i1 ? ii[u] : i1 - 1;
        rr[u] = S.sel ? (int64_t)S.sel[ic] : ic;
        nm[u] = 0;
#pragma unroll
        for (int c = 0; c < AGG_MAX_KEYS; c++) k[u][c] = 0;
    }
    if (PLAINK) {
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const int t = S.key[c].type;
            if (t == PH_I32 || t == PH_DATE) {
#pragma unroll
                for (int u = 0; u < BU; u++) k[u][c] = (unsigned long long)(long long)((const int32_t *)S.key[c].data)[rr[u]];
            } else if (t == PH_CODE8) {
#pragma unroll
                for (int u = 0; u < BU; u++) k[u][c] = ((const uint8_t *)S.key[c].data)[rr[u]];
            } else {
#pragma unroll
                for (int u = 0; u < BU; u++) k[u][c] = ((const unsigned long long *)S.key[c].data)[rr[u]];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < BU; u++) bulk_row_keys<NK>(S, rr[u], k[u], &nm[u]);
    }
}

// TPB = 1024 (the second form): the grid is the scatter's — one workgroup per row range, 512 of them for 32 M rows — so 256 threads
// each kept two waves on a SIMD and ~4 MB of key reads in flight on the whole chip: 72 us for 256 MB; four times the threads per range
template <int NK, bool PLAINK, int TPB = 256>
__global__ __launch_bounds__(TPB) void bulk_count_kernel(BulkParams B) {
    extern __shared__ int hist[];
    for (int e = threadIdx.x; e < B.nparts; e += TPB) hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * B.rows_per_wg, i1 = i0 + B.rows_per_wg < B.S.n ? i0 + B.rows_per_wg : B.S.n;
    for (int64_t base = i0; base < i1; base += TPB * BU) {
        int64_t ii[BU], rr[BU];
        unsigned long long k[BU][AGG_MAX_KEYS];
        unsigned nm[BU];
        bulk_keys_batch<NK, PLAINK, TPB>(B.S, base, i1, ii, rr, k, nm);
#pragma unroll
        for (int u = 0; u < BU; u++)
            if (ii[u] < i1) atomicAdd(&hist[(keys_hash(k[u], nm[u], NK) >> B.shift) & (B.nparts - 1)], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B.nparts; e += TPB) B.counts[(int64_t)e * gridDim.x + blockIdx.x] = hist[e];
}

template <int NK, bool PLAINK>
__global__ __launch_bounds__(256) void bulk_scatter_kernel(BulkParams B) {
    extern __shared__ int cursor[];
    for (int e = threadIdx.x; e < B.nparts; e += 256) cursor[e] = B.counts[(int64_t)e * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t n = B.S.n;
    const int64_t i0 = (int64_t)blockIdx.x * B.rows_per_wg, i1 = i0 + B.rows_per_wg < n ? i0 + B.rows_per_wg : n;
    for (int64_t base = i0; base < i1; base += 256 * BU) {
        int64_t ii[BU], rr[BU];
        unsigned long long k[BU][AGG_MAX_KEYS];
        unsigned nm[BU];
        bulk_keys_batch<NK, PLAINK>(B.S, base, i1, ii, rr, k, nm);
        // the carried argument values of all BU rows first (independent loads in flight together; read row
        // by row behind each cursor add they were BU x nused dependent HBM latencies per thread and step)
        long long av[BU][BK_MAX_ARGS];
        unsigned vb[BU];
#pragma unroll
        for (int u = 0; u < BU; u++) vb[u] = 0;
#pragma unroll
        for (int j = 0; j < BK_MAX_ARGS; j++) {
            if (j >= B.nused) break;   // wave-uniform
            const AggCol &c = B.S.arg[B.used_col[j]];
            const bool w32 = c.type == PH_I32;
#pragma unroll
            for (int u = 0; u < BU; u++) {
                const int64_t ic = ii[u] < i1 ? ii[u] : i1 - 1;
                const int64_t ar = B.S.positional ? ic : rr[u];
                av[u][j] = w32 ? (long long)((const int32_t *)c.data)[ar] : ((const int64_t *)c.data)[ar];
                if (bit_valid(c.validity, ar)) vb[u] |= 1u << j; else av[u][j] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < BU; u++) {
            if (ii[u] >= i1) continue;
            const int64_t i = ii[u], r = rr[u];
            const int pos = atomicAdd(&cursor[(keys_hash(k[u], nm[u], NK) >> B.shift) & (B.nparts - 1)], 1);
            unsigned long long *rec = B.rec + (int64_t)pos * B.rec_words;
            const unsigned long long rowid = (unsigned long long)(B.S.row_base + (B.S.sel ? r : i));
            const unsigned long long tail = (unsigned long long)nm[u] | ((unsigned long long)vb[u] << 8);
            if (NK == 1 && B.nused == 1) {   // the common record {key, row, value, masks}: two 16-byte stores
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 a, b;
                a.x = k[u][0]; a.y = rowid; b.x = (unsigned long long)av[u][0]; b.y = tail;
                reinterpret_cast<u64x2 *>(rec)[0] = a;
                reinterpret_cast<u64x2 *>(rec)[1] = b;
                continue;
            }
#pragma unroll
            for (int c = 0; c < NK; c++) rec[c] = k[u][c];
            rec[NK] = rowid;
#pragma unroll
            for (int j = 0; j < BK_MAX_ARGS; j++)
                if (j < B.nused) rec[NK + 1 + j] = (unsigned long long)av[u][j];
            rec[NK + 1 + B.nused] = tail;
        }
    }
}

// The finished groups of one workgroup's LDS table -> the global table (1024-thread workgroups): rank the ready entries,
// reserve their ids with ONE add, plain stores of the state, one CAS per group for its slot.
template <int NK>
__device__ __forceinline__ void bulk_write_groups(const BulkParams &B, int T, const long long *l_first, const unsigned long long *l_key,
                                                  const unsigned long long *l_lo, const long long *l_hi, const int *l_state,
                                                  const unsigned *l_cnt, int *s_wsum, int *s_base_p) {
    const AggSinkParams &S = B.S;
    const int na = S.naggs;
    int &s_base = *s_base_p;
    const int per = T / 1024 > 0 ? T / 1024 : 1;
    int mine = 0;
    for (int q = 0; q < per; q++) {
        const int e = threadIdx.x * per + q;
        if (e < T && l_state[e] >= 0) mine++;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int rank = incl - mine, total = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wv) rank += s_wsum[w];
        total += s_wsum[w];
    }
    if (threadIdx.x == 0) {
        s_base = total ? atomicAdd(S.ngroups, total) : 0;
        if (total && (long long)s_base + total > S.gcap) { atomicOr(B.overflow, 1); s_base = -1; }
    }
    __syncthreads();
    if (s_base < 0) return;   // the table is too small: the host grows it and runs this kernel again
    for (int q = 0; q < per; q++) {
        const int e = threadIdx.x * per + q;
        if (e >= T || l_state[e] < 0) continue;
        const int gid = s_base + rank++;
        unsigned long long k[AGG_MAX_KEYS] = {0, 0, 0, 0};
        const unsigned nullmask = (unsigned)l_state[e];
#pragma unroll
        for (int c = 0; c < NK; c++) {
            k[c] = l_key[c * T + e];
            __hip_atomic_store(&S.gkeys[(int64_t)gid * NK + c], k[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&S.gnull[gid], nullmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // arrays are not pre-initialised
        S.first_row[gid] = l_first[e];
        for (int a = 0; a < na; a++) {
            const int kind = S.agg_kind[a];
            const unsigned long long lo = l_lo[e * na + a];
            S.cnt[(int64_t)gid * na + a] = l_cnt[e * na + a];
            S.sum_lo[(int64_t)gid * na + a] = lo;
            S.sum_hi[(int64_t)gid * na + a] = (kind == PH_A_SUM || kind == PH_A_AVG) ? l_hi[e * na + a] : 0;
        }
        // keys visible before the id (same publication rule as find_or_create); every key of this
        // build is distinct, so the first free slot of the probe sequence is claimed without a compare
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t slot = keys_hash(k, nullmask, NK) & S.mask;
        for (int guard = 0; guard < (1 << 24); guard++) {
            if (atomicCAS(&S.slots[slot], SLOT_EMPTY, gid) == SLOT_EMPTY) break;
            if ((guard & 255) == 255 && __hip_atomic_load(S.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            slot = (slot + 1) & S.mask;
        }
    }
}

template <int NK>
__global__ __launch_bounds__(1024) void bulk_build_kernel(BulkParams B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
    __shared__ int s_nent, s_base, s_wsum[16], s_flagged;
    const AggSinkParams &S = B.S;
    const int T = B.lds_entries, na = S.naggs;
    const int64_t n = S.n;
    // [first T x i64][key NK*T x u64][lo T*na x u64][hi T*na x i64][state T x i32][cnt T*na x u32]
    long long *l_first = reinterpret_cast<long long *>(bk_lds);
    unsigned long long *l_key = reinterpret_cast<unsigned long long *>(l_first + T);
    unsigned long long *l_lo = l_key + (size_t)NK * T;
    long long *l_hi = reinterpret_cast<long long *>(l_lo + (size_t)T * na);
    int *l_state = reinterpret_cast<int *>(l_hi + (size_t)T * na);
    unsigned *l_cnt = reinterpret_cast<unsigned *>(l_state + T);
    for (int e = threadIdx.x; e < T; e += 1024) {
        l_state[e] = L_EMPTY;
        l_first[e] = INT64_MAX;
        for (int a = 0; a < na; a++) {
            const int kind = S.agg_kind[a];
            l_lo[e * na + a] = kind == PH_A_MIN ? (unsigned long long)INT64_MAX : kind == PH_A_MAX ? (unsigned long long)INT64_MIN : 0ull;
            l_hi[e * na + a] = 0;
            l_cnt[e * na + a] = 0;
        }
    }
    if (threadIdx.x == 0) { s_nent = 0; s_flagged = 0; }
    __syncthreads();
    const int p = blockIdx.x, nwg = (int)((n + B.rows_per_wg - 1) / B.rows_per_wg);
    const int64_t start = B.counts[(int64_t)p * nwg];
    const int64_t end = p + 1 < B.nparts ? (int64_t)B.counts[(int64_t)(p + 1) * nwg] : *B.total;
    // Two phases over the partition's rows. Phase 0 inserts into / accumulates in the LDS table and
    // only FLAGS rows that found no room (table closed or probe window exhausted). Phase 1 runs
    // with the table frozen: a flagged row looks its key up again and accumulates in LDS when the
    // key got in after all, else takes the global path. Deciding "not in LDS" while other lanes
    // may still be inserting that very key would create the group twice (once by find_or_create,
    // once by the write-out below, which claims slots without comparing keys).
    const bool rec4 = NK == 1 && B.nused == 1;   // the common record {key, row, value, masks}: two 16-byte reads
    for (int phase = 0; phase < 2; phase++) {
    // nearly always no row was flagged (the partitions are sized for a quarter-full table): phase 1
    // would re-read every line of the partition just to look at the flag words
    if (phase == 1 && !s_flagged) break;
    for (int64_t t = start + threadIdx.x; t < end; t += 1024) {
        unsigned long long *rec = B.rec + t * B.rec_words;
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 ra = {0, 0}, rb = {0, 0};
        if (rec4) { ra = reinterpret_cast<const u64x2 *>(rec)[0]; rb = reinterpret_cast<const u64x2 *>(rec)[1]; }
        const unsigned long long flags = rec4 ? rb.y : rec[NK + 1 + B.nused];
        const bool no_room = flags >> 63;
        if (phase == 1 && !no_room) continue;
        unsigned long long k[AGG_MAX_KEYS] = {0, 0, 0, 0};
        if (rec4) k[0] = ra.x;
        else {
#pragma unroll
            for (int c = 0; c < NK; c++) k[c] = rec[c];
        }
        const unsigned nullmask = (unsigned)(flags & 0xFF);
        const unsigned vbits = (unsigned)((flags >> 8) & 0xFF);
        const long long frow = (long long)(rec4 ? ra.y : rec[NK]);
        int ent = -1;
        int idx = (int)lds_hash<NK>(k, nullmask) & (T - 1);
        for (int probes = 0, spins = 0; probes < 32 && spins < (1 << 16);) {
            int st = __hip_atomic_load(&l_state[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (st == L_EMPTY) {
                if (phase == 1) break;   // frozen table: an empty slot ends the probe sequence, the key is not here
                if (__hip_atomic_load(&s_nent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= T - T / 4) break;
                int old = atomicCAS(&l_state[idx], L_EMPTY, L_LOCKED);
                if (old == L_EMPTY) {
                    atomicAdd(&s_nent, 1);
#pragma unroll
                    for (int c = 0; c < NK; c++) l_key[c * T + idx] = k[c];
                    __hip_atomic_store(&l_state[idx], (int)nullmask, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ent = idx;
                    break;
                }
                spins++;
            } else if (st == L_LOCKED) {
                spins++;
            } else {
                bool eq = st == (int)nullmask;
#pragma unroll
                for (int c = 0; c < NK; c++) eq = eq && l_key[c * T + idx] == k[c];
                if (eq) { ent = idx; break; }
                idx = (idx + 1) & (T - 1);
                probes++;
            }
        }
        int gid = -1;
        if (ent < 0) {
            if (phase == 0) { rec[NK + 1 + B.nused] = flags | (1ull << 63); s_flagged = 1; continue; }   // decided in phase 1
            // no room in LDS: the ordinary global path for this row. Once the table has run out of
            // ids this attempt is void (the host grows the table and runs the build again).
            if (__hip_atomic_load(S.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                __hip_atomic_load(B.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
            gid = find_or_create(S, k, nullmask, keys_hash(k, nullmask, NK));
            if (gid < 0) continue;
            if (frow < __hip_atomic_load(&S.first_row[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&S.first_row[gid], frow);
        } else {
            if (frow < l_first[ent]) atomicMin(&l_first[ent], frow);
            // a flag left by an earlier, voided attempt over the same records must not send the row
            // through phase 1 a second time
            if (phase == 0 && no_room) rec[NK + 1 + B.nused] = flags & ~(1ull << 63);
        }
        const long long v0 = (long long)(rec4 ? rb.x : 0ull);
        for (int a = 0; a < na; a++) {
            if (!((S.agg_mask >> a) & 1)) continue;
            const int kind = S.agg_kind[a];
            bool valid = true;
            long long v = 0;
            if (kind != PH_A_COUNT_STAR) {
                int j = 0;
                for (; j < B.nused; j++) if (B.used_col[j] == S.agg_arg[a]) break;
                valid = (vbits >> j) & 1;
                v = rec4 ? v0 : (long long)rec[NK + 1 + j];
            }
            if (!valid) continue;
            if (ent >= 0) {
                const int ls = ent * na + a;
                atomicAdd(&l_cnt[ls], 1u);
                if (kind == PH_A_SUM || kind == PH_A_AVG) {   // exact 128-bit in LDS: returning add + carry
                    unsigned long long old = atomicAdd(&l_lo[ls], (unsigned long long)v);
                    long long delta = (old + (unsigned long long)v < old ? 1 : 0) + (v < 0 ? -1 : 0);
                    if (delta) atomicAdd((unsigned long long *)&l_hi[ls], (unsigned long long)delta);
                } else if (kind == PH_A_MIN) atomicMin((long long *)&l_lo[ls], v);
                else if (kind == PH_A_MAX) atomicMax((long long *)&l_lo[ls], v);
            } else {
                const int64_t st = (int64_t)gid * na + a;
                atomicAdd(&S.cnt[st], 1ull);
                if (kind == PH_A_SUM || kind == PH_A_AVG) add128(&S.sum_lo[st], &S.sum_hi[st], v);
                else if (kind == PH_A_MIN) atomicMin((long long *)&S.sum_lo[st], v);
                else if (kind == PH_A_MAX) atomicMax((long long *)&S.sum_lo[st], v);
            }
        }
    }
    __syncthreads();   // phase 0 done: the table takes no more keys
    }
    bulk_write_groups<NK>(B, T, l_first, l_key, l_lo, l_hi, l_state, l_cnt, s_wsum, &s_base);
}

// ------------------------------------------------------------------ bulk build, second form (big first sinks)
// The form above scatters one 32-byte record per row straight to its partition: with 512 partitions x 512
// workgroups the L2 holds a small part of the open lines, so HBM sees 32-byte partial writes (65 k groups, 32 M rows:
// scatter 1.0 ms, build 0.53 ms). This form
//   * keeps the partition count at (expected groups) / (groups one LDS table takes) — 128 for 65 k groups — and
//     gets its parallelism from SLICES: `slices` workgroups aggregate disjoint row ranges of one partition in LDS
//     tables of their own, write their groups as partial states, and one workgroup per partition merges them;
//   * stages every 256 x BU-row chunk in LDS ordered by partition, so a chunk leaves as runs of consecutive rows
//     per partition (16 rows at 128 partitions), into column arrays (key words, argument words, 4-byte row ids, and
//     flag words only when some key or argument has NULLs): 20 bytes per row for one key and one argument.
// A table that runs out of room in LDS voids the attempt (bit 1 of *overflow): the host runs the first form.
template <int NK, bool PLAIN, int BUV, int TPB>
__global__ __launch_bounds__(TPB) void bulk2_scatter_kernel(Bulk2Params Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sc_lds[];
    const BulkParams &B = Q.B;
    const AggSinkParams &S = B.S;
    constexpr int CH = TPB * BUV;
    const int P = B.nparts, W = Q.W;
    int *hist = reinterpret_cast<int *>(sc_lds);
    int *off = hist + P;
    int *cursor = off + P;
    unsigned long long *s_w = reinterpret_cast<unsigned long long *>(cursor + P + (P & 1));
    uint32_t *s_row = reinterpret_cast<uint32_t *>(s_w + (size_t)W * CH);
    int *s_dst = reinterpret_cast<int *>(s_row + CH);
    uint32_t *s_flags = reinterpret_cast<uint32_t *>(s_dst + CH);
    __shared__ int s_wsum[TPB / 64];
    for (int e = threadIdx.x; e < P; e += TPB) { hist[e] = 0; cursor[e] = B.counts[(int64_t)e * gridDim.x + blockIdx.x]; }
    __syncthreads();
    const int64_t n = S.n;
    const int64_t i0 = (int64_t)blockIdx.x * B.rows_per_wg, i1 = i0 + B.rows_per_wg < n ? i0 + B.rows_per_wg : n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t ii[BUV], rr[BUV];
    unsigned long long k[BUV][NK];
    unsigned nm[BUV], vb[BUV];
    long long av[BUV][BK_MAX_ARGS];
    // the reads of one chunk (keys, argument values): issued for chunk c + 1 as soon as chunk c sits in LDS, so they are in
    // flight while c is written out (two 256-thread workgroups per CU hide little latency by themselves)
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < BUV; u++) {
            ii[u] = base + u * TPB + threadIdx.x;
            const int64_t ic = ii[u] < i1 ? ii[u] : i1 - 1;
            rr[u] = S.sel ? (int64_t)S.sel[ic] : ic;
            nm[u] = 0;
            vb[u] = 0xFu;
        }
#pragma unroll
        for (int c = 0; c < NK; c++) {
            const int t = S.key[c].type;   // wave-uniform
            if (t == PH_I32 || t == PH_DATE) {
#pragma unroll
                for (int u = 0; u < BUV; u++) k[u][c] = (unsigned long long)(long long)((const int32_t *)S.key[c].data)[rr[u]];
            } else if (t == PH_CODE8) {
#pragma unroll
                for (int u = 0; u < BUV; u++) k[u][c] = ((const uint8_t *)S.key[c].data)[rr[u]];
            } else {
#pragma unroll
                for (int u = 0; u < BUV; u++) k[u][c] = ((const unsigned long long *)S.key[c].data)[rr[u]];
            }
            if (!PLAIN && S.key[c].validity) {
#pragma unroll
                for (int u = 0; u < BUV; u++)
                    if (!bit_valid(S.key[c].validity, rr[u])) { k[u][c] = 0; nm[u] |= 1u << c; }
            }
        }
#pragma unroll
        for (int j = 0; j < BK_MAX_ARGS; j++) {
            if (j >= B.nused) break;   // wave-uniform
            const AggCol &c = S.arg[B.used_col[j]];
            const bool w32 = c.type == PH_I32;
#pragma unroll
            for (int u = 0; u < BUV; u++) {
                const int64_t ic = ii[u] < i1 ? ii[u] : i1 - 1;
                const int64_t ar = S.positional ? ic : rr[u];
                av[u][j] = w32 ? (long long)((const int32_t *)c.data)[ar] : ((const int64_t *)c.data)[ar];
                if (!PLAIN && !bit_valid(c.validity, ar)) { vb[u] &= ~(1u << j); av[u][j] = 0; }
            }
        }
    };
    if (i0 < i1) fetch(i0);
    for (int64_t base = i0; base < i1; base += CH) {
        int part[BUV], rank[BUV];
        uint32_t rowv[BUV];
        bool live[BUV];
#pragma unroll
        for (int u = 0; u < BUV; u++) { live[u] = ii[u] < i1; rowv[u] = (uint32_t)(S.sel ? rr[u] : ii[u]); }
        // rank of every row inside its partition's run of this chunk
#pragma unroll
        for (int u = 0; u < BUV; u++) {
            unsigned long long kk[AGG_MAX_KEYS] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NK; c++) kk[c] = k[u][c];
            part[u] = (int)((keys_hash(kk, nm[u], NK) >> B.shift) & (uint64_t)(P - 1));
            rank[u] = live[u] ? atomicAdd(&hist[part[u]], 1) : 0;
        }
        __syncthreads();
        {   // exclusive scan of the chunk's histogram: `per` consecutive partitions per thread
            const int per = (P + TPB - 1) / TPB;
            int sum = 0;
            for (int q = 0; q < per; q++) { const int e = threadIdx.x * per + q; if (e < P) sum += hist[e]; }
            int incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wv; w++) run += s_wsum[w];
            for (int q = 0; q < per; q++) { const int e = threadIdx.x * per + q; if (e < P) { off[e] = run; run += hist[e]; } }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BUV; u++) {
            if (!live[u]) continue;
            const int j = off[part[u]] + rank[u];
#pragma unroll
            for (int c = 0; c < NK; c++) s_w[(size_t)j * W + c] = k[u][c];
#pragma unroll
            for (int a = 0; a < BK_MAX_ARGS; a++)
                if (a < B.nused) s_w[(size_t)j * W + NK + a] = (unsigned long long)av[u][a];
            s_row[j] = rowv[u];
            s_dst[j] = cursor[part[u]] + rank[u];
            if (!PLAIN) s_flags[j] = nm[u] | (vb[u] << 8);
        }
        if (base + CH < i1) fetch(base + CH);
        __syncthreads();
        const int m = (int)(i1 - base < CH ? i1 - base : CH);
        for (int j = threadIdx.x; j < m; j += TPB) {
            const int64_t d = s_dst[j];
            if (W == 2) {   // the common record {key, argument}: one 16-byte store
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                reinterpret_cast<u64x2 *>(Q.w64)[d] = reinterpret_cast<const u64x2 *>(s_w)[j];
            } else {
                for (int c = 0; c < W; c++) Q.w64[d * W + c] = s_w[(size_t)j * W + c];
            }
            Q.rowid[d] = s_row[j];
            if (!PLAIN) Q.flags[d] = s_flags[j];
        }
        for (int e = threadIdx.x; e < P; e += TPB) { cursor[e] += hist[e]; hist[e] = 0; }
        __syncthreads();
    }
}

// ---- second partition level (more groups than 512 partitions of LDS tables hold: up to 8192 bins). The first level's records,
// already ordered by their top partition bits, are partitioned again by ALL bin bits: a chunk of consecutive records lies in one
// or two first-level partitions, so it touches ~bins / parts1 bins and leaves in runs again. Two words of LDS per bin (`hist`
// holds the chunk's counts, then — scanned in place — its offsets; `cursor` the bin's next output position) + the staged chunk.
constexpr int B2R_T = 1024, B2R_U = 2, B2R_CH = B2R_T * B2R_U;

template <int NK>
__global__ __launch_bounds__(B2R_T) void bulk2_count_rec_kernel(Bulk2Params Q, Bulk2Rec In, int64_t n) {
    extern __shared__ int b2_hist[];
    const BulkParams &B = Q.B;
    for (int e = threadIdx.x; e < B.nparts; e += B2R_T) b2_hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * B.rows_per_wg, i1 = i0 + B.rows_per_wg < n ? i0 + B.rows_per_wg : n;
    constexpr int CU4 = 4;   // records of a thread in flight
    for (int64_t base = i0; base < i1; base += B2R_T * CU4) {
        unsigned long long k[CU4][AGG_MAX_KEYS];
        unsigned nm[CU4];
        bool live[CU4];
#pragma unroll
        for (int u = 0; u < CU4; u++) {
            const int64_t i = base + u * B2R_T + threadIdx.x;
            live[u] = i < i1;
            const int64_t ic = live[u] ? i : i1 - 1;
#pragma unroll
            for (int c = 0; c < AGG_MAX_KEYS; c++) k[u][c] = c < NK ? In.w64[ic * Q.W + c] : 0ull;
            nm[u] = In.flags ? In.flags[ic] & 0xFF : 0u;
        }
#pragma unroll
        for (int u = 0; u < CU4; u++)
            if (live[u]) atomicAdd(&b2_hist[(int)((keys_hash(k[u], nm[u], NK) >> B.shift) & (uint64_t)(B.nparts - 1))], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B.nparts; e += B2R_T) B.counts[(int64_t)e * gridDim.x + blockIdx.x] = b2_hist[e];
}

template <int NK>
__global__ __launch_bounds__(B2R_T) void bulk2_scatter_rec_kernel(Bulk2Params Q, Bulk2Rec In, int64_t n, int sub_bins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char b2_lds[];
    const BulkParams &B = Q.B;
    const int P = B.nparts, W = Q.W;
    int *hist = reinterpret_cast<int *>(b2_lds);
    int *cursor = hist + P;
    unsigned long long *s_w = reinterpret_cast<unsigned long long *>(cursor + P);   // (P is even)
    uint32_t *s_row = reinterpret_cast<uint32_t *>(s_w + (size_t)W * B2R_CH);
    int *s_dst = reinterpret_cast<int *>(s_row + B2R_CH);
    uint32_t *s_flags = reinterpret_cast<uint32_t *>(s_dst + B2R_CH);
    __shared__ int s_wsum[B2R_T / 64], s_lo, s_hi;
    for (int e = threadIdx.x; e < P; e += B2R_T) { hist[e] = 0; cursor[e] = B.counts[(int64_t)e * gridDim.x + blockIdx.x]; }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * B.rows_per_wg, i1 = i0 + B.rows_per_wg < n ? i0 + B.rows_per_wg : n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int MAXPER = 8;                // bins per thread of the scan: up to 8192 bins
    for (int64_t base = i0; base < i1; base += B2R_CH) {
        unsigned long long k[B2R_U][AGG_MAX_KEYS + BK_MAX_ARGS];
        uint32_t row[B2R_U], fl[B2R_U];
        int bin[B2R_U], rank[B2R_U];
        bool live[B2R_U];
#pragma unroll
        for (int u = 0; u < B2R_U; u++) {
            const int64_t i = base + u * B2R_T + threadIdx.x;
            live[u] = i < i1;
            const int64_t ic = live[u] ? i : i1 - 1;
#pragma unroll
            for (int c = 0; c < AGG_MAX_KEYS + BK_MAX_ARGS; c++) k[u][c] = c < W ? In.w64[ic * W + c] : 0ull;
            row[u] = In.rowid[ic];
            fl[u] = In.flags ? In.flags[ic] : 0xF00u;
        }
#pragma unroll
        for (int u = 0; u < B2R_U; u++) {
            unsigned long long kk[AGG_MAX_KEYS] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NK; c++) kk[c] = k[u][c];
            bin[u] = (int)((keys_hash(kk, fl[u] & 0xFF, NK) >> B.shift) & (uint64_t)(P - 1));
            rank[u] = live[u] ? atomicAdd(&hist[bin[u]], 1) : 0;
        }
        // the input is ordered by the first level's partition: the chunk's bins lie between those of its first and its last record
        if (threadIdx.x == 0) s_lo = bin[0] & ~(sub_bins - 1);
        {
            const int64_t last = (i1 - base < B2R_CH ? i1 - base : B2R_CH) - 1;
#pragma unroll
            for (int u = 0; u < B2R_U; u++)
                if ((int64_t)u * B2R_T + threadIdx.x == last) s_hi = bin[u] | (sub_bins - 1);
        }
        __syncthreads();
        const int lo = s_lo, hi = s_hi, span = hi - lo + 1;   // (a chunk over several first-level partitions: a longer span, still right)
        const int per = (span + B2R_T - 1) / B2R_T;
        int cnt[MAXPER];
        {
            int sum = 0;
#pragma unroll
            for (int q = 0; q < MAXPER; q++) { const int e = lo + threadIdx.x * per + q; cnt[q] = (q < per && e <= hi) ? hist[e] : 0; sum += cnt[q]; }
            int incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wv; w++) run += s_wsum[w];
#pragma unroll
            for (int q = 0; q < MAXPER; q++) { const int e = lo + threadIdx.x * per + q; if (q < per && e <= hi) { hist[e] = run; run += cnt[q]; } }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < B2R_U; u++) {
            if (!live[u]) continue;
            const int j = hist[bin[u]] + rank[u];
#pragma unroll
            for (int c = 0; c < AGG_MAX_KEYS + BK_MAX_ARGS; c++) if (c < W) s_w[(size_t)j * W + c] = k[u][c];
            s_row[j] = row[u];
            s_dst[j] = cursor[bin[u]] + rank[u];
            if (Q.flags) s_flags[j] = fl[u];
        }
        __syncthreads();
        const int m = (int)(i1 - base < B2R_CH ? i1 - base : B2R_CH);
        for (int j = threadIdx.x; j < m; j += B2R_T) {
            const int64_t d = s_dst[j];
            if (W == 2) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                reinterpret_cast<u64x2 *>(Q.w64)[d] = reinterpret_cast<const u64x2 *>(s_w)[j];
            } else {
                for (int c = 0; c < W; c++) Q.w64[d * W + c] = s_w[(size_t)j * W + c];
            }
            Q.rowid[d] = s_row[j];
            if (Q.flags) Q.flags[d] = s_flags[j];
        }
#pragma unroll
        for (int q = 0; q < MAXPER; q++) { const int e = lo + threadIdx.x * per + q; if (q < per && e <= hi) { cursor[e] += cnt[q]; hist[e] = 0; } }
        __syncthreads();
    }
}

// one workgroup per bin: the partial body with its groups written straight into the table (no slices, no merge)
// Two workgroups per CU: each runs three short, latency-bound phases (read, build, write out) with barriers between them, so
// 64 VGPRs a lane (8 waves per SIMD) let the second one fill the first one's waits (4 M groups of 32 M rows: 793 -> 725 us).
// With 8 waves per SIMD the lock wait of the insert path needs its s_sleep: without it older spinning waves starve the
// holder of the issue slot (measured: the same kernel took 50 ms).
template <int NK, bool PLAIN>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void bulk2_build_direct_kernel(Bulk2Params Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
    __shared__ int s_nent, s_wsum[16], s_void;
    if (__hip_atomic_load(Q.B.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) return;
    bulk2_partial_body<NK, PLAIN, true>(Q, bk_lds, s_nent, s_wsum, s_void);
}

// the LDS table of the build workgroups: [first T x i64][key NK*T x u64][lo T*na x u64][hi T*na x i64][state T x i32][cnt T*na x u32]
struct BulkLds {
    long long *first;
    unsigned long long *key, *lo;
    long long *hi;
    int *state;
    unsigned *cnt;
};
template <int NK>
__device__ __forceinline__ BulkLds bulk_lds_init(unsigned char *lds, int T, const AggSinkParams &S) {
    BulkLds L;
    const int na = S.naggs;
    L.first = reinterpret_cast<long long *>(lds);
    L.key = reinterpret_cast<unsigned long long *>(L.first + T);
    L.lo = L.key + (size_t)NK * T;
    L.hi = reinterpret_cast<long long *>(L.lo + (size_t)T * na);
    L.state = reinterpret_cast<int *>(L.hi + (size_t)T * na);
    L.cnt = reinterpret_cast<unsigned *>(L.state + T);
    for (int e = threadIdx.x; e < T; e += blockDim.x) {
        L.state[e] = L_EMPTY;
        L.first[e] = INT64_MAX;
        for (int a = 0; a < na; a++) {
            const int kind = S.agg_kind[a];
            L.lo[e * na + a] = kind == PH_A_MIN ? (unsigned long long)INT64_MAX : kind == PH_A_MAX ? (unsigned long long)INT64_MIN : 0ull;
            L.hi[e * na + a] = 0;
            L.cnt[e * na + a] = 0;
        }
    }
    return L;
}

// find the key's entry or insert it; -1 when the table takes no more keys (three quarters full) or the probe window ends
template <int NK>
__device__ __forceinline__ int bulk_lds_find_insert(const BulkLds &L, int T, const unsigned long long *k, unsigned nullmask, int *s_nent) {
    int idx = (int)lds_hash<NK>(k, nullmask) & (T - 1);
    for (int probes = 0, spins = 0; probes < 32 && spins < (1 << 16);) {
        const int st = __hip_atomic_load(&L.state[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (st == L_EMPTY) {
            if (__hip_atomic_load(s_nent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= T - T / 4) return -1;
            const int old = atomicCAS(&L.state[idx], L_EMPTY, L_LOCKED);
            if (old == L_EMPTY) {
                atomicAdd(s_nent, 1);
#pragma unroll
                for (int c = 0; c < NK; c++) L.key[c * T + idx] = k[c];
                __hip_atomic_store(&L.state[idx], (int)nullmask, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return idx;
            }
            spins++;
        } else if (st == L_LOCKED) {
            spins++;
        } else {
            bool eq = st == (int)nullmask;
#pragma unroll
            for (int c = 0; c < NK; c++) eq = eq && L.key[c * T + idx] == k[c];
            if (eq) return idx;
            idx = (idx + 1) & (T - 1);
            probes++;
        }
    }
    return -1;
}

// slice `s` of partition `p`: its rows aggregated in LDS, its groups written as partial states (bulk2_partial_body, agg_sink.inc;
// the plan-specialised form of the same body is compiled at run time: agg_bulk2_partial_spec)
template <int NK, bool PLAIN>
__global__ __launch_bounds__(1024) void bulk2_partial_kernel(Bulk2Params Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
    __shared__ int s_nent, s_wsum[16], s_void;
    bulk2_partial_body<NK, PLAIN>(Q, bk_lds, s_nent, s_wsum, s_void);
}

// partition p: the partial states of its slices merged in LDS, the finished groups written to the table
template <int NK>
__global__ __launch_bounds__(1024) void bulk2_merge_kernel(Bulk2Params Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
    __shared__ int s_nent, s_base, s_wsum[16], s_void;
    const BulkParams &B = Q.B;
    const AggSinkParams &S = B.S;
    const int T = B.lds_entries, na = S.naggs;
    if (__hip_atomic_load(B.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) return;   // a slice ran out of room (the previous launch)
    const BulkLds L = bulk_lds_init<NK>(bk_lds, T, S);
    if (threadIdx.x == 0) { s_nent = 0; s_void = 0; }
    __syncthreads();
    const int p = blockIdx.x;
    for (int sl = 0; sl < Q.slices; sl++) {
        const int wg = p * Q.slices + sl, cnt = Q.partial_n[wg];
        const unsigned long long *in = Q.partials + (int64_t)wg * Q.pcap * Q.pwords;
        for (int t = threadIdx.x; t < cnt; t += 1024) {
            const unsigned long long *o = in + (int64_t)t * Q.pwords;
            unsigned long long k[AGG_MAX_KEYS] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NK; c++) k[c] = o[c];
            const long long frow = (long long)o[NK];
            const unsigned nullmask = (unsigned)o[NK + 1];
            const int ent = bulk_lds_find_insert<NK>(L, T, k, nullmask, &s_nent);
            if (ent < 0) { s_void = 1; continue; }
            if (frow < L.first[ent]) atomicMin(&L.first[ent], frow);
            for (int a = 0; a < na; a++) {
                const int kind = S.agg_kind[a];
                const unsigned long long lo = o[NK + 2 + 3 * a];
                const long long hi = (long long)o[NK + 3 + 3 * a];
                const unsigned c = (unsigned)o[NK + 4 + 3 * a];
                if (c == 0) continue;   // this slice saw no input of the aggregate
                const int ls = ent * na + a;
                atomicAdd(&L.cnt[ls], c);
                if (kind == PH_A_SUM || kind == PH_A_AVG) {   // 128-bit add: low words with carry into the high ones
                    const unsigned long long old = atomicAdd(&L.lo[ls], lo);
                    const long long delta = hi + (old + lo < old ? 1 : 0);
                    if (delta) atomicAdd((unsigned long long *)&L.hi[ls], (unsigned long long)delta);
                } else if (kind == PH_A_MIN) atomicMin((long long *)&L.lo[ls], (long long)lo);
                else if (kind == PH_A_MAX) atomicMax((long long *)&L.lo[ls], (long long)lo);
            }
        }
    }
    __syncthreads();
    if (s_void) { if (threadIdx.x == 0) atomicOr(B.overflow, 2); return; }
    bulk_write_groups<NK>(B, T, L.first, L.key, L.lo, L.hi, L.state, L.cnt, s_wsum, &s_base);
}

// key column c of all groups -> dense column + validity bits; a thread converts 8 groups
// (one group per lane and step: a wave reads 64 consecutive records and composes the validity bytes with a ballot. The first form gave a lane
// 8 consecutive groups — a 64-byte stride between the lanes of one load — and ran at 0.9 TB/s: 518 + 337 us of Q18's 3 ms at SF10.)
__global__ __launch_bounds__(256) void agg_keys_kernel(const unsigned long long *__restrict__ gkeys,
                                                       const unsigned *__restrict__ gnull, int nkeys, int c, int type,
                                                       int ng, void *__restrict__ out, uint8_t *__restrict__ valid) {
    const int lane = threadIdx.x & 63;
    for (int64_t g0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) & ~63ll; g0 < ng; g0 += (int64_t)gridDim.x * 256) {
        const int64_t g = g0 + lane;
        bool ok = false;
        if (g < ng) {
            const unsigned long long k = gkeys[g * nkeys + c];
            ok = !((gnull[g] >> c) & 1);
            if (type == PH_I32 || type == PH_DATE) ((int32_t *)out)[g] = (int32_t)k;
            else if (type == PH_CODE8) ((uint8_t *)out)[g] = (uint8_t)k;
            else ((unsigned long long *)out)[g] = k;
        }
        const unsigned long long bits = __ballot(ok);
        if (valid && (lane & 7) == 0 && g0 + lane < ng) valid[(g0 + lane) >> 3] = (uint8_t)(bits >> lane);
    }
}

// aggregate a of all groups -> dense int64 column + validity bits (a group no input reached is NULL). flag |= 1 when a SUM does not fit int64
__global__ __launch_bounds__(256) void agg_values_kernel(const unsigned long long *__restrict__ sum_lo, const long long *__restrict__ sum_hi,
                                                         const unsigned long long *__restrict__ cnt, int naggs, int a, int kind, int ng,
                                                         long long *__restrict__ out, uint8_t *__restrict__ valid, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const bool counting = kind == PH_A_COUNT || kind == PH_A_COUNT_STAR;
    bool wide = false;
    for (int64_t g0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) & ~63ll; g0 < ng; g0 += (int64_t)gridDim.x * 256) {
        const int64_t g = g0 + lane;
        bool ok = false;
        if (g < ng) {
            const int64_t st = g * naggs + a;
            const unsigned long long c = cnt[st];
            ok = c != 0;                         // SumOp / CountOp / MinMaxOp.Finalize: NULL when never set (COUNT: when 0)
            const long long lo = (long long)sum_lo[st];
            if (kind == PH_A_SUM && c != 0 && sum_hi[st] != (lo >> 63)) wide = true;
            out[g] = counting ? (long long)c : lo;
        }
        const unsigned long long bits = __ballot(ok);
        if (valid && (lane & 7) == 0 && g0 + lane < ng) valid[(g0 + lane) >> 3] = (uint8_t)(bits >> lane);
    }
    if (__ballot(wide) && lane == 0) atomicOr(flag, 1);
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
    int *counters = nullptr;  // [0] ngroups, [1] error flag, [2] need-grow flag of the running sink; [4..7] ph_agg_topk's state words (zeroed with the
                              // counters by the first sink's clearing launch: the first top-k of a table needs no memset launch of its own)
    bool topk_state_used = false;
    int *kinds_dev = nullptr;
    int64_t rows_sunk = 0;
    int64_t expected_groups = 0;   // ph_agg_create's hint: selects the bulk build of the first sink
    int kinds_host[ph::AGG_MAX_AGGS] = {};  // source of the asynchronous upload to kinds_dev
    bool fresh = true;             // counters not cleared yet (the first sink's one clearing launch does it)
    bool sorted_built = false;     // groups were written by ph_agg_sink_sorted: the slot array is not valid, no further sinks
};

namespace {


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

// everything a sink needs cleared, in ONE launch (four memsets before): the slot array of a new table
// (-1), counter words [c0, c0 + nc), the per-workgroup progress words
__global__ __launch_bounds__(256) void agg_clear_kernel(int32_t *__restrict__ slots, int64_t cap, int *__restrict__ counters, int c0, int nc,
                                                        int *__restrict__ progress, int nprog) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    if (slots) {
        int4 *s4 = reinterpret_cast<int4 *>(slots);   // cap is a power of two >= 4096
        for (int64_t i = t; i < cap / 4; i += step) s4[i] = make_int4(-1, -1, -1, -1);
    }
    if (progress) for (int64_t i = t; i < nprog; i += step) progress[i] = 0;
    if (t < nc) counters[c0 + t] = 0;
}

int agg_clear(ph_agg *a, bool slots, int c0, int nc, int *progress, int nprog) {
    const int64_t work = std::max<int64_t>(slots ? a->cap / 4 : 0, nprog);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, (int64_t)a->ctx->cu_count * 4));
    agg_clear_kernel<<<grid, 256, 0, a->ctx->stream>>>(slots ? a->slots : nullptr, a->cap, a->counters, c0, nc, progress, nprog);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// (re)allocate for `cap` slots, carrying over `ng` existing groups; clear_slots = false leaves the new
// slot array to the caller's agg_clear (first sinks: one launch for slots + counters + progress)
int agg_resize(ph_agg *a, int64_t cap, int ng, bool clear_slots = true) {
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
    if (clear_slots || ng > 0) PH_HIP(hipMemsetAsync(slots, 0xff, (size_t)cap * 4, ctx->stream));
    if (ng > 0) {
        PH_HIP(hipMemcpyAsync(gkeys, a->gkeys, (size_t)ng * a->nkeys * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(gnull, a->gnull, (size_t)ng * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(sum_lo, a->sum_lo, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(sum_hi, a->sum_hi, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(cnt, a->cnt, (size_t)ng * na * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(first_row, a->first_row, (size_t)ng * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // group state is initialised by whoever creates the group (find_or_create / the bulk build), not here
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

// initial capacity: the reference starts at 2*2048 entries (aggregate_exec.go:332-339)
int agg_ensure(ph_agg *a, int64_t min_cap = 0) {
    if (a->cap != 0) return PH_OK;
    PH_CHECK(agg_resize(a, std::max(min_cap, next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups))), 0, false));
    PH_CHECK(agg_clear(a, true, 0, a->fresh ? 8 : 0, nullptr, 0));
    a->fresh = false;
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
    int *kinds = a->kinds_host;   // lives as long as the handle: the upload below needs no sync
    for (int c = 0; c < nkeys; c++) a->key_types[c] = key_types[c];
    for (int i = 0; i < naggs; i++) { a->aggs[i] = aggs[i]; kinds[i] = aggs[i].kind; }
    int rc = PH_OK;
    if (ctx->pool_alloc(32, (void **)&a->counters) != PH_OK) {   // cleared by the first sink's clearing launch (a->fresh)
        ph::set_error("ph_agg_create: device allocation failed");
        rc = PH_EHIP;
    }
    // initial capacity: the reference starts at 2*2048 entries (aggregate_exec.go:332-339)
    // the table itself is allocated by the first call that needs it (agg_ensure): the first sink knows
    // its row count and sizes the table once instead of replacing the hint-sized one
    a->expected_groups = expected_groups;
    if (rc != PH_OK) { ph_agg_free(a); return rc; }
    *out = a;
    return PH_OK;
}

extern "C" int ph_agg_group_count(ph_agg *a, int64_t *ngroups) {
    PH_REQUIRE(a && ngroups, "ph_agg_group_count: bad arguments");
    if (a->cap == 0) { *ngroups = 0; return PH_OK; }   // nothing sunk yet
    int c[2] = {0, 0};
    PH_CHECK(a->ctx->download(c, a->counters, 8));
    if (c[1]) { ph::set_error("ph_agg: device table error flag %d", c[1]); return PH_EHIP; }
    *ngroups = c[0];
    return PH_OK;
}

namespace {

template <int NK>
int bulk_launch(ph_agg *a, ph::BulkParams &B, int nwg, size_t lds, int64_t nc, int64_t *total_dev, bool build_only) {
    hipStream_t st = a->ctx->stream;
    if (!build_only) {
        bool plaink = true;
        for (int c = 0; c < NK; c++) plaink = plaink && !B.S.key[c].validity;
        if (plaink) ph::bulk_count_kernel<NK, true><<<nwg, 256, (size_t)B.nparts * 4, st>>>(B);
        else ph::bulk_count_kernel<NK, false><<<nwg, 256, (size_t)B.nparts * 4, st>>>(B);
        PH_CHECK(ph::exclusive_scan_i32(a->ctx, B.counts, nc, total_dev));
        if (plaink) ph::bulk_scatter_kernel<NK, true><<<nwg, 256, (size_t)B.nparts * 4, st>>>(B);
        else ph::bulk_scatter_kernel<NK, false><<<nwg, 256, (size_t)B.nparts * 4, st>>>(B);
    }
    // up to 120 KiB of LDS per workgroup: above the default dynamic limit
    PH_HIP(hipFuncSetAttribute((const void *)ph::bulk_build_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ph::bulk_build_kernel<NK><<<B.nparts, 1024, lds, st>>>(B);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

int bulk2_spec_kernel(ph_ctx *ctx, const ph::AggSinkParams &P, bool plain, int slots, size_t lds, ph::JitKernel *out);

// Second form of the bulk build (see bulk2_scatter_kernel): PH_EUNSUPPORTED = not this shape, or the attempt was void
// (nothing sunk, table and counters clean) — the caller runs the first form.
template <int NK>
int bulk2_launch_scatter(ph_agg *a, ph::Bulk2Params &Q, int nwg, size_t lds_scatter, int bu, int tpb, bool plain, int64_t nc, int64_t *total_dev) {
    hipStream_t st = a->ctx->stream;
    {
        if (plain) ph::bulk_count_kernel<NK, true, 1024><<<nwg, 1024, (size_t)Q.B.nparts * 4, st>>>(Q.B);
        else ph::bulk_count_kernel<NK, false, 1024><<<nwg, 1024, (size_t)Q.B.nparts * 4, st>>>(Q.B);
        PH_CHECK(ph::exclusive_scan_i32(a->ctx, Q.B.counts, nc, total_dev));
#define PH_B2S(PL, BU, TP)                                                                                                                \
    do {                                                                                                                                  \
        PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_scatter_kernel<NK, PL, BU, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
        ph::bulk2_scatter_kernel<NK, PL, BU, TP><<<nwg, TP, lds_scatter, st>>>(Q);                                                       \
    } while (0)
        // the chunk is bu x 256 rows whatever the shape of the workgroup: 1024 threads x bu / 4 rows (more waves per CU to
        // hide the reads and the write-out behind) or 256 threads x bu rows
        if (tpb == 1024) {
            if (plain) { if (bu == 16) PH_B2S(true, 4, 1024); else if (bu == 8) PH_B2S(true, 2, 1024); else PH_B2S(true, 1, 1024); }
            else { if (bu == 16) PH_B2S(false, 4, 1024); else if (bu == 8) PH_B2S(false, 2, 1024); else PH_B2S(false, 1, 1024); }
        } else if (tpb == 512) {
            if (plain) { if (bu == 8) PH_B2S(true, 4, 512); else if (bu == 4) PH_B2S(true, 2, 512); else PH_B2S(true, 1, 512); }
            else { if (bu == 8) PH_B2S(false, 4, 512); else if (bu == 4) PH_B2S(false, 2, 512); else PH_B2S(false, 1, 512); }
        } else {
            if (plain) { if (bu == 8) PH_B2S(true, 8, 256); else if (bu == 4) PH_B2S(true, 4, 256); else PH_B2S(true, 2, 256); }
            else { if (bu == 8) PH_B2S(false, 8, 256); else if (bu == 4) PH_B2S(false, 4, 256); else PH_B2S(false, 2, 256); }
        }
#undef PH_B2S
    }
    PH_HIP(hipGetLastError());
    return PH_OK;
}

template <int NK>
int bulk2_launch(ph_agg *a, ph::Bulk2Params &Q, int nwg, size_t lds_build, size_t lds_scatter, int bu, int tpb, bool plain, int64_t nc,
                 int64_t *total_dev, bool merge_only) {
    hipStream_t st = a->ctx->stream;
    if (!merge_only) {
        PH_CHECK(bulk2_launch_scatter<NK>(a, Q, nwg, lds_scatter, bu, tpb, plain, nc, total_dev));
        ph::JitKernel spec{};
        if (bulk2_spec_kernel(a->ctx, Q.B.S, plain, Q.B.lds_entries, lds_build, &spec) == PH_OK) {
            // the body specialised for this shape through hiprtc (the generic one spends ~370 instructions per 64 rows on
            // aggregate-kind and argument-slot interpretation: 320 us per 32 M rows)
            ph::Bulk2Params copy = Q;
            size_t size = sizeof copy;
            void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
            PH_HIP(hipModuleLaunchKernel(spec.fn, (unsigned)(Q.B.nparts * Q.slices), 1, 1, 1024, 1, 1, 0, st, nullptr, config));
        } else if (plain) {
            PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_partial_kernel<NK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            ph::bulk2_partial_kernel<NK, true><<<Q.B.nparts * Q.slices, 1024, lds_build, st>>>(Q);
        } else {
            PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_partial_kernel<NK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            ph::bulk2_partial_kernel<NK, false><<<Q.B.nparts * Q.slices, 1024, lds_build, st>>>(Q);
        }
    }
    PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_merge_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ph::bulk2_merge_kernel<NK><<<Q.B.nparts, 1024, lds_build, st>>>(Q);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// two partition levels + one build workgroup per bin (more than 512 bins: up to 8192, ~4 M groups)
template <int NK>
int bulk2_two_level_launch(ph_agg *a, ph::Bulk2Params &Q1, ph::Bulk2Params &Q2, int nwg1, int nwg2, size_t lds_build, size_t lds_scatter1, int bu, int tpb,
                           bool plain, int64_t n, int64_t *total1, int64_t *total2, bool build_only) {
    hipStream_t st = a->ctx->stream;
    if (!build_only) {
        // level 1: the ordinary count + staged scatter into 64 partitions by the top bin bits
        PH_CHECK(bulk2_launch_scatter<NK>(a, Q1, nwg1, lds_scatter1, bu, tpb, plain, (int64_t)Q1.B.nparts * nwg1, total1));
        // level 2: its records into all bins
        const ph::Bulk2Rec In{Q1.w64, Q1.rowid, Q1.flags};
        const size_t lds2 = (size_t)Q2.B.nparts * 8 + (size_t)ph::B2R_CH * (8 * Q2.W + 12);
        PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_scatter_rec_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        ph::bulk2_count_rec_kernel<NK><<<nwg2, ph::B2R_T, (size_t)Q2.B.nparts * 4, st>>>(Q2, In, n);
        PH_CHECK(ph::exclusive_scan_i32(a->ctx, Q2.B.counts, (int64_t)Q2.B.nparts * nwg2, total2));
        ph::bulk2_scatter_rec_kernel<NK><<<nwg2, ph::B2R_T, lds2, st>>>(Q2, In, n, Q2.B.nparts / Q1.B.nparts);
    }
    if (plain) {
        PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_build_direct_kernel<NK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ph::bulk2_build_direct_kernel<NK, true><<<Q2.B.nparts, 1024, lds_build, st>>>(Q2);
    } else {
        PH_HIP(hipFuncSetAttribute((const void *)ph::bulk2_build_direct_kernel<NK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ph::bulk2_build_direct_kernel<NK, false><<<Q2.B.nparts, 1024, lds_build, st>>>(Q2);
    }
    PH_HIP(hipGetLastError());
    return PH_OK;
}

int bulk_sink_v2(ph_agg *a, const ph::AggSinkParams &P, const bool *used, int64_t n) {
    static const bool off = getenv("PH_AGG_BULK_V1") != nullptr;
    if (off || n < (4ll << 20) || n >= (1ll << 31)) return PH_EUNSUPPORTED;
    ph_ctx *ctx = a->ctx;
    ph::Bulk2Params Q{};
    ph::BulkParams &B = Q.B;
    B.S = P;
    bool plain = true;
    for (int c = 0; c < P.nargs; c++) {
        if (!used[c]) continue;
        if (B.nused == ph::BK_MAX_ARGS) return PH_EUNSUPPORTED;
        B.used_col[B.nused++] = c;
        plain = plain && !P.arg[c].validity;
    }
    for (int c = 0; c < P.nargs; c++) if (used[c]) B.S.arg_used |= 1u << c;   // (the specialised partial build derives the argument slots from it)
    const int nk = a->nkeys, na = a->naggs;
    for (int c = 0; c < nk; c++) plain = plain && !P.key[c].validity;
    const int per_entry = 4 + 8 * nk + 8 + 20 * na;
    int T = 256;
    while (T * 2 * per_entry <= 120 * 1024 && T < 4096) T *= 2;
    if (T * per_entry > 120 * 1024) return PH_EUNSUPPORTED;
    if (const char *te = getenv("PH_AGG_BULK_T")) { const int v = atoi(te); if (v >= 256 && v <= T && (v & (v - 1)) == 0) T = v; }
    B.lds_entries = T;
    // partitions: a quarter-full LDS table per partition (every slice of a partition sees all of its groups)
    static const int fill_div = [] { const char *e = getenv("PH_AGG_BULK_FILL"); const int v = e ? atoi(e) : 4; return v == 2 || v == 3 || v == 4 || v == 8 ? v : 4; }();
    const int64_t want_parts = (a->expected_groups + T / fill_div - 1) / (T / fill_div);
    int nparts = 16;
    while (nparts < want_parts) nparts *= 2;
    const bool two_level = nparts > 512;   // more bins than one staged scatter serves in runs: a first level of 64 partitions, then all bins
    if (nparts > 8192 || getenv("PH_AGG_BULK_ONE_LEVEL") != nullptr) { if (two_level) return PH_EUNSUPPORTED; }
    const int bins = nparts;
    int log_bins = 0;
    while ((1 << log_bins) < bins) log_bins++;
    if (two_level) nparts = 64;
    B.nparts = nparts;
    B.shift = two_level ? 40 + (log_bins - 6) : 40;
    int slices = 1;
    while (!two_level && nparts * slices < 512 && slices < 64) slices *= 2;   // two workgroups per CU in all: every slice pays a table initialisation and a write-out
    if (const char *se = getenv("PH_AGG_BULK_SLICES")) slices = std::max(1, std::min(atoi(se), 64));
    Q.slices = slices;
    Q.W = nk + B.nused;
    // chunk of the scatter: what fits 64 KiB of LDS beside the three partition tables (two workgroups per CU)
    const int row_bytes = 8 * Q.W + 8 + (plain ? 0 : 4);
    const size_t tables = (size_t)(3 * nparts + (nparts & 1)) * 4;
    int bu = 8;
    while (bu > 2 && tables + (size_t)256 * bu * row_bytes > 64 * 1024) bu /= 2;
    if (const char *be = getenv("PH_AGG_BULK_BU")) { const int v = atoi(be); if (v == 2 || v == 4 || v == 8) bu = v; }
    int tpb = 1024;
    if (const char *te = getenv("PH_AGG_BULK_TPB")) { const int v = atoi(te); if (v == 256 || v == 512 || v == 1024) tpb = v; }
    if (tpb == 1024 && bu < 4) bu = 4;   // at least a row per thread
    // one 1024-thread workgroup per CU can stage 4096 rows: runs twice as long per partition (fewer partial lines written)
    if (tpb == 1024 && bu == 8 && !getenv("PH_AGG_BULK_BU") && tables + (size_t)256 * 16 * row_bytes <= 120 * 1024) bu = 16;
    if (const char *be = getenv("PH_AGG_BULK_BU")) { if (atoi(be) == 16 && tpb == 1024 && tables + (size_t)256 * 16 * row_bytes <= 120 * 1024) bu = 16; }
    const size_t lds_scatter = tables + (size_t)256 * bu * row_bytes;
    if (lds_scatter > 124 * 1024) return PH_EUNSUPPORTED;
    const int ch = 256 * bu;
    B.rows_per_wg = std::max<int64_t>(ch, ph::round_up((n + 511) / 512, ch));
    const int nwg = (int)((n + B.rows_per_wg - 1) / B.rows_per_wg);
    const int64_t nc = (int64_t)nparts * nwg;
    int64_t cap = a->cap ? a->cap : next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups));
    cap = std::max(cap, next_pow2(4 * a->expected_groups));
    bool new_table = false;
    if (cap != a->cap) { PH_CHECK(agg_resize(a, cap, 0, false)); new_table = true; }
    Q.pcap = T;
    Q.pwords = nk + 2 + 3 * na;
    char *tmp = nullptr;
    const int64_t o_w64 = ph::round_up(nc * 4, 16), o_row = o_w64 + (int64_t)Q.W * n * 8, o_flags = o_row + ph::round_up(n * 4, 16),
                  o_part = o_flags + (plain ? 0 : ph::round_up(n * 4, 16)), o_pn = o_part + (two_level ? 0 : (int64_t)nparts * slices * Q.pcap * Q.pwords * 8),
                  o_total = o_pn + ph::round_up((int64_t)nparts * slices * 4, 16);
    PH_CHECK(ctx->pool_alloc(o_total + 16, (void **)&tmp));
    B.counts = (int32_t *)tmp;
    Q.w64 = (unsigned long long *)(tmp + o_w64);
    Q.rowid = (uint32_t *)(tmp + o_row);
    Q.flags = plain ? nullptr : (uint32_t *)(tmp + o_flags);
    Q.partials = (unsigned long long *)(tmp + o_part);
    Q.partial_n = (int *)(tmp + o_pn);
    int64_t *total_dev = (int64_t *)(tmp + o_total);
    B.total = total_dev;
    B.overflow = a->counters + 2;
    const size_t lds_build = (size_t)T * per_entry;
    // second level: records of the first, partitioned again by all bin bits
    ph::Bulk2Params Q2{};
    char *tmp2 = nullptr;
    int nwg2 = 0;
    int64_t *total2 = nullptr;
    if (two_level) {
        const int64_t rows2 = std::max<int64_t>(ph::B2R_CH, ph::round_up((n + 511) / 512, ph::B2R_CH));
        nwg2 = (int)((n + rows2 - 1) / rows2);
        const int64_t nc2 = (int64_t)bins * nwg2;
        const int64_t p_w64 = ph::round_up(nc2 * 4, 16), p_row = p_w64 + (int64_t)Q.W * n * 8, p_flags = p_row + ph::round_up(n * 4, 16),
                      p_total = p_flags + (plain ? 0 : ph::round_up(n * 4, 16));
        if (ctx->pool_alloc(p_total + 16, (void **)&tmp2) != PH_OK) { ctx->pool_release(tmp); ph::set_error("ph_agg_sink: allocation failed"); return PH_EHIP; }
        Q2 = Q;
        Q2.B.nparts = bins;
        Q2.B.shift = 40;
        Q2.B.rows_per_wg = rows2;
        Q2.B.counts = (int32_t *)tmp2;
        Q2.w64 = (unsigned long long *)(tmp2 + p_w64);
        Q2.rowid = (uint32_t *)(tmp2 + p_row);
        Q2.flags = plain ? nullptr : (uint32_t *)(tmp2 + p_flags);
        Q2.slices = 1;
        Q2.B.lds_entries = T / 2;   // a bin holds T / 4 groups on average: half-full tables, TWO build workgroups per CU (they close at 3/4 = 1.5x the hint)
        total2 = (int64_t *)(tmp2 + p_total);
        Q2.B.total = total2;
    }
    int rc = PH_OK;
    bool settled = false, voided = false;
    for (int attempt = 0; attempt < 6 && rc == PH_OK; attempt++) {
        B.S.slots = a->slots;
        B.S.mask = (uint64_t)a->cap - 1;
        B.S.gkeys = a->gkeys; B.S.gnull = a->gnull; B.S.sum_lo = a->sum_lo; B.S.sum_hi = a->sum_hi;
        B.S.cnt = a->cnt; B.S.first_row = a->first_row; B.S.gcap = a->gcap;
        if (attempt == 0) {
            if ((rc = agg_clear(a, new_table, 0, 8, nullptr, 0)) != PH_OK) break;
            a->fresh = false;
        } else if (hipMemsetAsync(a->counters, 0, 12, ctx->stream) != hipSuccess) { rc = PH_EHIP; break; }
        if (two_level) {
            Q2.B.S = B.S;   // (the table pointers of this attempt)
            switch (nk) {
            case 1: rc = bulk2_two_level_launch<1>(a, Q, Q2, nwg, nwg2, lds_build / 2, lds_scatter, bu, tpb, plain, n, total_dev, total2, attempt > 0); break;
            case 2: rc = bulk2_two_level_launch<2>(a, Q, Q2, nwg, nwg2, lds_build / 2, lds_scatter, bu, tpb, plain, n, total_dev, total2, attempt > 0); break;
            case 3: rc = bulk2_two_level_launch<3>(a, Q, Q2, nwg, nwg2, lds_build / 2, lds_scatter, bu, tpb, plain, n, total_dev, total2, attempt > 0); break;
            default: rc = bulk2_two_level_launch<4>(a, Q, Q2, nwg, nwg2, lds_build / 2, lds_scatter, bu, tpb, plain, n, total_dev, total2, attempt > 0); break;
            }
        } else
        switch (nk) {
        case 1: rc = bulk2_launch<1>(a, Q, nwg, lds_build, lds_scatter, bu, tpb, plain, nc, total_dev, attempt > 0); break;
        case 2: rc = bulk2_launch<2>(a, Q, nwg, lds_build, lds_scatter, bu, tpb, plain, nc, total_dev, attempt > 0); break;
        case 3: rc = bulk2_launch<3>(a, Q, nwg, lds_build, lds_scatter, bu, tpb, plain, nc, total_dev, attempt > 0); break;
        default: rc = bulk2_launch<4>(a, Q, nwg, lds_build, lds_scatter, bu, tpb, plain, nc, total_dev, attempt > 0); break;
        }
        if (rc != PH_OK) break;
        int c[3] = {0, 0, 0};
        if ((rc = ctx->download(c, a->counters, 12)) != PH_OK) break;
        if (c[2] & 2) { voided = true; break; }          // an LDS table ran out of room: more groups than the hint said
        if (!c[1] && !c[2]) { settled = true; break; }
        rc = agg_resize(a, next_pow2(4 * std::max<int64_t>(c[0], a->gcap)), 0);   // ids ran out: a larger table, the merge again
    }
    ctx->pool_release(tmp);
    if (tmp2) ctx->pool_release(tmp2);
    if (rc != PH_OK) return rc;
    if (settled) return PH_OK;
    // void: leave an empty table and clean counters behind
    (void)voided;
    PH_CHECK(agg_clear(a, true, 0, 8, nullptr, 0));
    return PH_EUNSUPPORTED;
}

// First sink into an empty table with a high expected cardinality: see bulk_build_kernel.
// Returns PH_EUNSUPPORTED when the shape is not handled (the caller runs the ordinary sink).
int bulk_sink(ph_agg *a, const ph::AggSinkParams &P, const bool *used, int64_t n) {
    {   // big inputs, up to ~260 k expected groups: the staged, sliced form
        const int rc2 = bulk_sink_v2(a, P, used, n);
        if (rc2 != PH_EUNSUPPORTED) return rc2;
    }
    ph_ctx *ctx = a->ctx;
    ph::BulkParams B{};
    B.S = P;
    for (int c = 0; c < P.nargs; c++) {
        if (!used[c]) continue;
        if (B.nused == ph::BK_MAX_ARGS) return PH_EUNSUPPORTED;
        B.used_col[B.nused++] = c;
    }
    const int nk = a->nkeys, na = a->naggs;
    const int per_entry = 4 + 8 * nk + 8 + 20 * na;
    int T = 256;
    while (T * 2 * per_entry <= 120 * 1024 && T < 4096) T *= 2;
    if (T * per_entry > 120 * 1024) return PH_EUNSUPPORTED;
    B.lds_entries = T;
    int64_t want_parts = (a->expected_groups + T / 4 - 1) / (T / 4);
    // one 1024-thread workgroup builds one partition: big inputs get at least two partitions per CU
    // (65 k groups from 32 M rows used 64 partitions = a quarter of the CUs: build 1.09 -> 0.51 ms)
    int nparts = n >= (4ll << 20) ? 512 : 8;
    if (const char *pe = getenv("PH_AGG_BULK_PARTS")) nparts = std::max(8, std::min(atoi(pe), 4096));
    while (nparts < want_parts && nparts < 4096) nparts *= 2;
    B.nparts = nparts;
    B.shift = 40;
    B.rows_per_wg = std::max<int64_t>(1024, ph::round_up((n + 511) / 512, 256));
    const int nwg = (int)((n + B.rows_per_wg - 1) / B.rows_per_wg);
    const int64_t nc = (int64_t)nparts * nwg;
    // capacity: when it is affordable make growth impossible (capacity > rows), else start from
    // the hint and let the build report an overflow
    const bool sure = n <= (4ll << 20);
    int64_t cap = a->cap ? a->cap : next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups));
    if (sure) while (cap / 2 <= n) cap *= 2;
    else cap = std::max(cap, next_pow2(4 * a->expected_groups));
    bool new_table = false;
    if (cap != a->cap) { PH_CHECK(agg_resize(a, cap, 0, false)); new_table = true; }
    // partition records
    char *tmp = nullptr;
    B.rec_words = nk + 1 + B.nused + 1;
    const int64_t o_rec = ph::round_up(nc * 4, 16), o_total = o_rec + (int64_t)B.rec_words * n * 8;
    PH_CHECK(ctx->pool_alloc(o_total + 16, (void **)&tmp));
    B.counts = (int32_t *)tmp;
    B.rec = (unsigned long long *)(tmp + o_rec);
    int64_t *total_dev = (int64_t *)(tmp + o_total);
    B.total = total_dev;
    B.overflow = a->counters + 2;
    const size_t lds = (size_t)T * per_entry;
    int rc = PH_OK;
    bool settled = false;   // the last attempt ran without running out of ids
    for (int attempt = 0; attempt < 6 && rc == PH_OK; attempt++) {
        B.S.slots = a->slots;
        B.S.mask = (uint64_t)a->cap - 1;
        B.S.gkeys = a->gkeys; B.S.gnull = a->gnull; B.S.sum_lo = a->sum_lo; B.S.sum_hi = a->sum_hi;
        B.S.cnt = a->cnt; B.S.first_row = a->first_row; B.S.gcap = a->gcap;
        if (attempt == 0) {   // one clearing launch: the new table's slots and all four counter words
            if ((rc = agg_clear(a, new_table, 0, 8, nullptr, 0)) != PH_OK) break;
            a->fresh = false;
        } else if (hipMemsetAsync(a->counters, 0, 12, ctx->stream) != hipSuccess) { rc = PH_EHIP; break; }
        switch (nk) {
        case 1: rc = bulk_launch<1>(a, B, nwg, lds, nc, total_dev, attempt > 0); break;
        case 2: rc = bulk_launch<2>(a, B, nwg, lds, nc, total_dev, attempt > 0); break;
        case 3: rc = bulk_launch<3>(a, B, nwg, lds, nc, total_dev, attempt > 0); break;
        default: rc = bulk_launch<4>(a, B, nwg, lds, nc, total_dev, attempt > 0); break;
        }
        if (rc != PH_OK) break;
        if (sure) { settled = true; break; }
        int c[3] = {0, 0, 0};
        if ((rc = ctx->download(c, a->counters, 12)) != PH_OK) break;
        if (!c[1] && !c[2]) { settled = true; break; }   // neither the row-by-row path nor a partition ran out of ids
        // c[0] = ids asked for so far (every partition adds its need even when it then backs off)
        rc = agg_resize(a, next_pow2(4 * std::max<int64_t>(c[0], a->gcap)), 0);
    }
    ctx->pool_release(tmp);
    if (rc == PH_OK && !settled) {
        // every attempt overflowed (a hint too small by orders of magnitude): the table was just
        // replaced by an empty, larger one. Clear the voided attempt's counters and let the caller
        // run the ordinary growing sink over the same rows; nothing has been sunk.
        if (hipMemsetAsync(a->counters, 0, 12, ctx->stream) != hipSuccess) return PH_EHIP;
        return PH_EUNSUPPORTED;
    }
    return rc;
}

}  // namespace

namespace {

// ---- plan-specialised sink: the same device source (agg_sink.inc), compiled through hiprtc with
// the sink's shape as compile-time constants. Key = everything the SK_* macros fold.
// does any aggregate of the call count its own non-NULL inputs (an argument column with a validity mask)?
// Otherwise every count is the LDS entry's row count and the per-aggregate counters are not allocated.
bool agg_spec_need_cnt(const ph::AggSinkParams &P) {
    for (int a = 0; a < P.naggs; a++)
        if (((P.agg_mask >> a) & 1) && P.agg_kind[a] != PH_A_COUNT_STAR && P.arg[P.agg_arg[a]].validity) return true;
    return false;
}
int agg_spec_entry_bytes(const ph::AggSinkParams &P) { return 12 + 8 * P.nkeys + 8 * P.naggs + (agg_spec_need_cnt(P) ? 4 * P.naggs : 0); }

std::string agg_spec_defines(const ph::AggSinkParams &P, int threads, int slots, size_t lds, std::string *key) {
    std::ostringstream d, k;
    auto list = [&](const char *name, int n, auto f) {   // #define name(i) ((i)==0?v0:(i)==1?v1:...:0)
        d << "#define " << name << "(i) (";
        for (int i = 0; i < n; i++) d << "(i)==" << i << "?" << f(i) << ":";
        d << "0)\n";
    };
    if (getenv("PH_AGG_HASH2")) d << "#define PH_LDS_HASH_2MUL 1\n";
    d << "#define SPEC_SLOTS " << slots << "\n#define SPEC_LDS_BYTES " << lds << "\n#define SPEC_NEED_CNT " << (agg_spec_need_cnt(P) ? 1 : 0) << "\n";
    d << "#define PH_SPEC 1\n#define SPEC_T " << threads << "\n#define SPEC_NK " << P.nkeys << "\n#define SPEC_NA " << P.naggs << "\n"
      << "#define SPEC_HAS_SEL " << (P.sel ? 1 : 0) << "\n#define SPEC_POSITIONAL " << (P.positional ? 1 : 0) << "\n"
      << "#define SPEC_AGG_MASK " << (P.agg_mask & ((1u << P.naggs) - 1u)) << "u\n#define SPEC_ARG_USED " << P.arg_used << "u\n";
    list("SPEC_KIND_OF", P.naggs, [&](int i) { return P.agg_kind[i]; });
    list("SPEC_ARG_OF", P.naggs, [&](int i) { return P.agg_kind[i] == PH_A_COUNT_STAR ? 0 : P.agg_arg[i]; });
    list("SPEC_KEYTYPE_OF", P.nkeys, [&](int i) { return P.key[i].type; });
    list("SPEC_KEYNULL_OF", P.nkeys, [&](int i) { return P.key[i].validity ? 1 : 0; });
    list("SPEC_ARGTYPE_OF", P.nargs, [&](int i) { return ((P.arg_used >> i) & 1) ? P.arg[i].type : 0; });
    list("SPEC_ARGNULL_OF", P.nargs, [&](int i) { return ((P.arg_used >> i) & 1) && P.arg[i].validity ? 1 : 0; });
    *key = "aggsink:" + d.str();
    return d.str();
}

// the specialised kernel of this sink shape, or PH_EUNSUPPORTED (no hiprtc, PH_AGG_JIT=0, compile trouble)
int agg_spec_kernel(ph_ctx *ctx, const ph::AggSinkParams &P, int threads, int slots, size_t lds, ph::JitKernel *out) {
    const char *e = getenv("PH_AGG_JIT");   // read per call: tests compare both kernels in one process
    if (e && atoi(e) == 0) return PH_EUNSUPPORTED;
    std::string key;
    std::string defs = agg_spec_defines(P, threads, slots, lds, &key);
    if (ph::jit_cached(ctx, key, out)) return PH_OK;   // no 20 KB source concatenation on a hit
    int rc = ph::jit_module(ctx, key, defs + AGG_SINK_SRC, "agg_sink_spec", out);
    return rc == PH_OK ? PH_OK : PH_EUNSUPPORTED;
}

}  // namespace

namespace {
// the partial build of the bulk form (agg_bulk2_partial_spec) for this shape, or PH_EUNSUPPORTED
int bulk2_spec_kernel(ph_ctx *ctx, const ph::AggSinkParams &P, bool plain, int slots, size_t lds, ph::JitKernel *out) {
    const char *e = getenv("PH_AGG_JIT");
    if (e && atoi(e) == 0) return PH_EUNSUPPORTED;
    std::string key;
    std::string defs = agg_spec_defines(P, 1024, slots, lds, &key) + "#define PH_SPEC_BULK2 1\n#define SPEC_BULK_PLAIN " + (plain ? "1" : "0") + "\n";
    key = "aggbulk2:" + key + (plain ? "p" : "n");
    if (ph::jit_cached(ctx, key, out)) return PH_OK;
    int rc = ph::jit_module(ctx, key, defs + AGG_SINK_SRC, "agg_bulk2_partial_spec", out);
    return rc == PH_OK ? PH_OK : PH_EUNSUPPORTED;
}
}  // namespace

// Build check without a device: the sink source specialised for a Q9-like shape (two INTEGER keys,
// SUM of a decimal, positional arguments) and a nullable two-aggregate shape compile for gfx950.
extern "C" int ph_agg_jit_selfcheck(int32_t which) {
    ph::AggSinkParams P{};
    if (which == 0) {
        P.nkeys = 2; P.naggs = 1; P.nargs = 1;
        P.key[0].type = PH_I32; P.key[1].type = PH_I32;
        P.arg[0].type = PH_DEC64;
        P.agg_kind[0] = PH_A_SUM; P.agg_arg[0] = 0;
        P.positional = 1; P.agg_mask = ~0u; P.arg_used = 1;
    } else if (which == 1) {
        static const uint8_t dummy = 0;
        P.nkeys = 3; P.naggs = 4; P.nargs = 2;
        P.key[0].type = PH_I64; P.key[1].type = PH_DATE; P.key[2].type = PH_CODE8; P.key[1].validity = &dummy;
        P.arg[0].type = PH_I32; P.arg[1].type = PH_DEC64; P.arg[1].validity = &dummy;
        P.agg_kind[0] = PH_A_MIN; P.agg_kind[1] = PH_A_AVG; P.agg_kind[2] = PH_A_COUNT_STAR; P.agg_kind[3] = PH_A_MAX;
        P.agg_arg[0] = 0; P.agg_arg[1] = 1; P.agg_arg[2] = -1; P.agg_arg[3] = 1;
        P.sel = reinterpret_cast<const int32_t *>(&dummy);
        P.agg_mask = 0xb; P.arg_used = 3;
    } else if (which == 2 || which == 3) {
        // the partial build of the bulk form: one BIGINT key, SUM + COUNT(*) without NULLs / two keys, NULL-able argument, MIN + AVG + COUNT
        static const uint8_t dummy = 0;
        std::string key, log;
        if (which == 2) {
            P.nkeys = 1; P.naggs = 2; P.nargs = 1;
            P.key[0].type = PH_I64; P.arg[0].type = PH_I64;
            P.agg_kind[0] = PH_A_SUM; P.agg_arg[0] = 0; P.agg_kind[1] = PH_A_COUNT_STAR; P.agg_arg[1] = -1;
            P.agg_mask = ~0u; P.arg_used = 1;
        } else {
            P.nkeys = 2; P.naggs = 3; P.nargs = 3;
            P.key[0].type = PH_I64; P.key[1].type = PH_I32; P.key[0].validity = &dummy;
            P.arg[0].type = PH_DEC64; P.arg[2].type = PH_I32; P.arg[2].validity = &dummy;
            P.agg_kind[0] = PH_A_MIN; P.agg_arg[0] = 2; P.agg_kind[1] = PH_A_AVG; P.agg_arg[1] = 0; P.agg_kind[2] = PH_A_COUNT; P.agg_arg[2] = 2;
            P.agg_mask = ~0u; P.arg_used = 5;
        }
        const int slots = which == 2 ? 2048 : 1024;
        std::string src = agg_spec_defines(P, 1024, slots, (size_t)slots * (4 + 8 * P.nkeys + 8 + 20 * P.naggs), &key) +
                          "#define PH_SPEC_BULK2 1\n#define SPEC_BULK_PLAIN " + (which == 2 ? "1" : "0") + "\n" + AGG_SINK_SRC;
        int rc = ph::jit_compile_only(src, "gfx950", &log);
        if (rc != PH_OK) ph::set_error("ph_agg_jit_selfcheck(%d): %s", which, log.c_str());
        return rc;
    } else {
        ph::set_error("ph_agg_jit_selfcheck: shapes 0..3");
        return PH_EINVAL;
    }
    std::string key, log;
    const int per_entry = agg_spec_entry_bytes(P);
    const int slots = which == 0 ? 2048 : 512;
    std::string src = agg_spec_defines(P, which == 0 ? 1024 : 256, slots, (size_t)slots * per_entry, &key) + AGG_SINK_SRC;
    return ph::jit_compile_only(src, "gfx950", &log);
}

extern "C" int ph_agg_sink(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs,
                           const int32_t *sel, int64_t n, int32_t positional, int64_t row_base) {
    return ph_agg_sink_masked(a, keys, args, nargs, sel, n, positional, row_base, 0xFFFFFFFFu);
}

// ------------------------------------------------------------------ streaming aggregate (sorted input)
// When the rows arrive ordered by the group key — the planner's StreamAggregate case: Q3 groups the
// output of a join whose probe side is clustered by the key — every group is one RUN of adjacent rows
// and no hash table is needed: mark the run heads (a row whose key tuple differs from its
// predecessor's), scan their counts, and let the thread of each head reduce its run and write the
// group's record, in first-seen order by construction. Three launches over n rows where the bulk build
// of the hash table takes four plus the table's initialisation (Q3: 298 k rows -> 113 k groups, 56 -> ~20 us).
// The order is VERIFIED: a key tuple lexicographically below its predecessor's raises a deferred
// PH_ECONSTRAINT of the ctx (the caller sinks again with ph_agg_sink).
namespace ph {

struct SortedAgg {
    int nkeys, naggs, nargs;
    AggCol key[AGG_MAX_KEYS];
    AggCol arg[AGG_MAX_AGGS];
    int agg_kind[AGG_MAX_AGGS];
    int agg_arg[AGG_MAX_AGGS];
    int64_t n, row_base;
    unsigned long long *gkeys;
    unsigned *gnull;
    unsigned long long *sum_lo;
    long long *sum_hi;
    unsigned long long *cnt;
    long long *first_row;
    int *counters;       // [0] group count (written here)
    int *violation;      // deferred-error word of the ctx
};

// Round 4: tiles instead of one-row-per-thread walks. A workgroup takes a tile of SA_TILE rows, a lane SA_V consecutive rows: the keys are read
// once per pass with plain coalesced loads, run heads come from comparing neighbours, group ids from the tile's head count (scanned over the tiles)
// plus a workgroup scan. Every aggregate is reduced as a SEGMENTED SCAN over the tile: a lane first folds its own rows (runs that begin and end
// inside its rows are written at once), then lanes combine across the wave with shuffles and across the four waves through LDS; the lane that
// holds a run's LAST row in the tile adds what the scan carried in and writes the group's record — once, with plain stores. A run that crosses
// tiles leaves two partials per tile in a small side array (what precedes the tile's first head, what follows its last head) and a fix-up
// launch, one lane per tile, adds them up — per TILE, so a run of a million rows (a table clustered by a low-cardinality key) costs a thousand
// steps of one lane, not a million dependent loads (round 3 withdrew the form for such inputs; ADVICE r3: a single long run among short ones
// was still walked row by row). 60 M rows x (8 B key + 4 B argument), 15 M groups (Q18's subquery): 1 325 us -> see profiles/README.md.
constexpr int SA_V = 4;                    // rows per lane
constexpr int SA_TILE = 256 * SA_V;        // rows per workgroup

struct SaPart {                            // the partial state of one aggregate over some rows of one run
    unsigned long long lo;                 // SUM / AVG: the 128-bit sum; MIN / MAX: the value
    long long hi;
    unsigned long long cnt;                // non-NULL inputs (COUNT_STAR: rows)
};

__device__ __forceinline__ SaPart sa_empty(int kind) {
    SaPart p;
    p.lo = kind == PH_A_MIN ? (unsigned long long)INT64_MAX : kind == PH_A_MAX ? (unsigned long long)INT64_MIN : 0ull;
    p.hi = 0; p.cnt = 0;
    return p;
}
__device__ __forceinline__ SaPart sa_combine(int kind, const SaPart &a, const SaPart &b) {   // a's rows precede b's
    SaPart r;
    r.cnt = a.cnt + b.cnt;
    if (kind == PH_A_MIN) { r.lo = (long long)b.lo < (long long)a.lo ? b.lo : a.lo; r.hi = 0; }
    else if (kind == PH_A_MAX) { r.lo = (long long)b.lo > (long long)a.lo ? b.lo : a.lo; r.hi = 0; }
    else { r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0); }
    return r;
}
__device__ __forceinline__ SaPart sa_value(int kind, long long v) {
    SaPart p;
    p.lo = (unsigned long long)v; p.hi = (kind == PH_A_MIN || kind == PH_A_MAX) ? 0 : (v < 0 ? -1 : 0); p.cnt = 1;
    return p;
}
__device__ __forceinline__ SaPart sa_shfl_up(const SaPart &p, int o) {
    SaPart r;
    r.lo = __shfl_up(p.lo, o); r.hi = __shfl_up(p.hi, o); r.cnt = __shfl_up(p.cnt, o);
    return r;
}
// (The descriptor lives in DEVICE memory and the kernels take a pointer: indexed by the aggregate / key number it is a handful of scalar loads.
// Passed by value, the same indexing made the compiler copy the whole struct into scratch — 712 bytes per lane — and the kernel ran at a third
// of the row-per-thread form it replaces.)
__device__ __forceinline__ void sa_store(const SortedAgg &S, int64_t g, int a, int kind, const SaPart &p) {
    const int64_t st = g * S.naggs + a;
    S.cnt[st] = p.cnt;
    if (kind == PH_A_COUNT || kind == PH_A_COUNT_STAR) { S.sum_lo[st] = 0; S.sum_hi[st] = 0; }
    else if (kind == PH_A_MIN || kind == PH_A_MAX) { S.sum_lo[st] = p.lo; S.sum_hi[st] = 0; }   // (cnt == 0: the value is the identity; readers test cnt)
    else { S.sum_lo[st] = p.lo; S.sum_hi[st] = p.hi; }
}

// is row i a run head: its key tuple differs from row i-1's? (*descending: it is lexicographically BELOW it — the order claim is broken)
__device__ __forceinline__ bool sa_is_head(const SortedAgg &S, int64_t i, bool *descending) {
    if (i == 0) return true;
    bool differ = false, below = false;
#pragma unroll
    for (int c = 0; c < AGG_MAX_KEYS; c++) {
        if (c >= S.nkeys || differ) continue;
        const long long a = (long long)load_key(S.key[c], i - 1), b = (long long)load_key(S.key[c], i);
        if (a != b) { differ = true; below = b < a; }
    }
    *descending = below;
    return differ;
}

// a lane's SA_V consecutive values of a 4- or 8-byte column as 16-byte loads (the lane's first row is a multiple of SA_V: aligned whenever the column
// is; the ragged last lanes and unaligned columns take the scalar loads)
__device__ __forceinline__ void sa_load4(int type, const void *data, int64_t base, int64_t n, long long v[SA_V]) {
    const bool w4 = type == PH_I32 || type == PH_DATE;
    if (base + SA_V <= n && type != PH_CODE8 && ((reinterpret_cast<uintptr_t>(data) & 15) == 0)) {
        if (w4) {
            const int4 x = *reinterpret_cast<const int4 *>((const int32_t *)data + base);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
            const longlong2 a = *reinterpret_cast<const longlong2 *>((const int64_t *)data + base);
            const longlong2 b = *reinterpret_cast<const longlong2 *>((const int64_t *)data + base + 2);
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < SA_V; r++) {
        const int64_t i = base + r;
        v[r] = i < n ? (type == PH_CODE8 ? (long long)((const uint8_t *)data)[i] : w4 ? (long long)((const int32_t *)data)[i] : ((const int64_t *)data)[i]) : 0;
    }
}

// head flags of a lane's SA_V rows as a bit mask (rows >= n: no head); k0[] receives the first key's values
__device__ __forceinline__ unsigned sa_lane_heads(const SortedAgg &S, int64_t base, bool *bad, long long k0[SA_V]) {
    unsigned h = 0;
    sa_load4(S.key[0].type, S.key[0].data, base, S.n, k0);
    if (S.nkeys == 1) {   // one key: the lane's own keys in 16-byte loads + its predecessor's last
        long long prev = base > 0 && base - 1 < S.n ? (long long)load_key(S.key[0], base - 1) : 0;
#pragma unroll
        for (int r = 0; r < SA_V; r++) {
            const int64_t i = base + r;
            if (i < S.n) {
                if (i == 0 || k0[r] != prev) { h |= 1u << r; if (i > 0 && k0[r] < prev) *bad = true; }
                prev = k0[r];
            }
        }
        return h;
    }
#pragma unroll
    for (int r = 0; r < SA_V; r++) {
        const int64_t i = base + r;
        if (i >= S.n) break;
        bool desc = false;
        if (sa_is_head(S, i, &desc)) h |= 1u << r;
        *bad = *bad || desc;
    }
    return h;
}

__device__ __forceinline__ void sorted_heads_body(const SortedAgg &S, int32_t *__restrict__ counts) {
    const int64_t base = (int64_t)blockIdx.x * SA_TILE + (int64_t)threadIdx.x * SA_V;
    bool bad = false;
    long long k0[SA_V];
    int heads = __popc(sa_lane_heads(S, base, &bad, k0));
    if (bad) atomicOr(S.violation, 4);
    for (int o = 32; o > 0; o >>= 1) heads += __shfl_xor(heads, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = heads;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// two entries per kernel: the descriptor by pointer (any shape), and BY VALUE for tables with ONE aggregate — then every index into the
// descriptor is a constant, it stays in the kernel-argument segment, and the call needs no descriptor upload (a pageable host-to-device copy
// costs ~10 us of launch time: a third of the whole aggregate over Q3's 298 k rows)
__global__ __launch_bounds__(256) void sorted_heads_kernel(const SortedAgg *__restrict__ Sp, int32_t *__restrict__ counts) { sorted_heads_body(*Sp, counts); }
__global__ __launch_bounds__(256) void sorted_heads_kernel_v(const SortedAgg S, int32_t *__restrict__ counts) { sorted_heads_body(S, counts); }

// One pass (round 4, late): the tile's first group id comes from a decoupled look-back over the tiles' head counts INSIDE the groups kernel — the
// heads pass (a second read of the key columns) and the scan launch between the two are gone. Tiles are taken from a ticket counter, so a tile
// only ever waits for tiles that already run; states as in scan_lookback_kernel (ops_select.hip). state == nullptr: the two-pass form.
struct SaLook {
    unsigned long long *state;
    unsigned *ticket;
    unsigned ticket_base;
    unsigned long long epoch;
    int32_t *block_off;   // out: every tile's first group id (the fix-up kernel reads it)
    int64_t *total;       // out: the number of groups
};
constexpr unsigned long long SA_AGG = 1ull, SA_INCL = 2ull;

// side[(tile * naggs + a) * 2 + 0] = the partial of the rows before the tile's first head (the whole tile when it has none),
// side[.. + 1] = the partial from the tile's last head to its end (unused without a head)
template <bool ONE>
__device__ __forceinline__ void sorted_groups_body(const SortedAgg &S, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total,
                                                   SaPart *__restrict__ side, const SaLook &L) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ unsigned s_tile;
    __shared__ long long s_prefix;
    if (L.state) {
        if (tid == 0) s_tile = atomicAdd(L.ticket, 1u) - L.ticket_base;
        __syncthreads();
    }
    const unsigned tile = L.state ? s_tile : blockIdx.x;
    const int64_t base = (int64_t)tile * SA_TILE + (int64_t)tid * SA_V;
    bool bad = false;
    long long k0[SA_V];
    const unsigned h = sa_lane_heads(S, base, &bad, k0);
    if (L.state && bad) atomicOr(S.violation, 4);   // (the two-pass form checks the order in its heads pass)
    const int nh = __popc(h);
    // exclusive scan of the head counts over the workgroup -> the lane's first group id
    __shared__ int wc[4];
    __shared__ SaPart wpart[4];
    __shared__ int wflag[4];
    int incl = nh;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    if (lane == 63) wc[wv] = incl;
    __syncthreads();
    int excl = incl - nh;
    for (int q = 0; q < wv; q++) excl += wc[q];
    const int tile_heads = wc[0] + wc[1] + wc[2] + wc[3];
    if (L.state) {   // the tile's first group id: look back over the predecessors' published head counts / inclusive prefixes
        // The FIRST WAVE looks at 64 predecessors at a time (one lane each): up to the nearest inclusive prefix everything must be published; the
        // window's counts are added with one wave reduction. (One thread walking back tile by tile made 58 k tiles a serial chain: 1.09 ms.)
        if (wv == 0) {
            const unsigned long long tag = L.epoch << 34;
            if (lane == 0) __hip_atomic_store(&L.state[tile], tag | ((tile == 0 ? SA_INCL : SA_AGG) << 32) | (unsigned)tile_heads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long prefix = 0;
            long long hi = (long long)tile - 1;   // the window is tiles hi, hi - 1, .., hi - 63
            unsigned spins = 0;
            while (hi >= 0) {
                const long long q = hi - lane;
                unsigned long long st = 0;
                if (q >= 0) st = __hip_atomic_load(&L.state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long flag = (q >= 0 && (st >> 34) == L.epoch) ? (st >> 32) & 3ull : 0ull;
                const unsigned long long incl_mask = __ballot(flag == SA_INCL);
                const unsigned long long ready_mask = __ballot(flag != 0 || q < 0);
                // lanes 0 .. first (the nearest inclusive prefix, or the whole window) must all be published
                const int first = incl_mask ? __ffsll((long long)incl_mask) - 1 : 63;
                const unsigned long long need = first == 63 ? ~0ull : ((1ull << (first + 1)) - 1ull);
                if ((ready_mask & need) != need) { if (++spins > (1u << 24)) break; continue; }   // not all published in this call yet: look again
                long long v = (q >= 0 && lane <= first) ? (long long)(unsigned)st : 0;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                prefix += v;
                if (incl_mask) break;
                hi -= 64;
            }
            if (lane == 0) {
                if (tile != 0) __hip_atomic_store(&L.state[tile], tag | (SA_INCL << 32) | (unsigned)(prefix + tile_heads), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = prefix;
                L.block_off[tile] = (int32_t)prefix;
                if ((int64_t)(tile + 1) * SA_TILE >= S.n) { *L.total = prefix + tile_heads; S.counters[0] = (int)(prefix + tile_heads); }   // the last tile
            }
        }
        __syncthreads();
    }
    const int64_t goff = (L.state ? s_prefix : (int64_t)block_off[tile]) + excl;     // group id of the lane's first head; goff - 1: the run its leading rows continue
    if (!L.state && tile == 0 && tid == 0) S.counters[0] = (int)*total;
    // the heads' keys and first rows
    {
        int k = 0;
#pragma unroll
        for (int r = 0; r < SA_V; r++) {
            if (!((h >> r) & 1)) continue;
            const int64_t i = base + r, g = goff + k++;
            S.first_row[g] = S.row_base + i;
            S.gnull[g] = 0;
            S.gkeys[g * S.nkeys] = (unsigned long long)k0[r];
#pragma unroll
            for (int c = 1; c < AGG_MAX_KEYS; c++) if (c < S.nkeys) S.gkeys[g * S.nkeys + c] = load_key(S.key[c], i);
        }
    }
    int rows_here = 0;
#pragma unroll
    for (int r = 0; r < SA_V; r++) rows_here += base + r < S.n ? 1 : 0;
    const int na = ONE ? 1 : S.naggs;
    for (int a = 0; a < na; a++) {
        const int kind = ONE ? S.agg_kind[0] : S.agg_kind[a];
        // the lane's own rows: `pre` = before its first head (all rows without one), interior runs written at once, `suf` = from its last head on
        SaPart pre = sa_empty(kind), cur = sa_empty(kind);
        int k = 0;
        bool seen_head = false;
        const AggCol *col = kind == PH_A_COUNT_STAR ? nullptr : (ONE ? &S.arg[0] : &S.arg[S.agg_arg[a]]);   // (ONE: the host put the argument first)
        long long vv[SA_V] = {0, 0, 0, 0};
        if (col) sa_load4(col->type, col->data, base, S.n, vv);
#pragma unroll
        for (int r = 0; r < SA_V; r++) {
            const int64_t i = base + r;
            if (i >= S.n) break;
            if ((h >> r) & 1) {
                if (seen_head) sa_store(S, goff + k - 1, a, kind, cur);   // a run that began and ended inside this lane's rows
                else pre = cur;
                cur = sa_empty(kind);
                seen_head = true;
                k++;
            }
            if (!col) cur.cnt++;
            else if (!col->validity || bit_valid(col->validity, i)) {
                cur = sa_combine(kind, cur, sa_value(kind, vv[r]));
            }
        }
        if (!seen_head) { pre = cur; }
        // segmented inclusive scan over the lanes: element = (has a head ? the part from its last head : all its rows)
        SaPart x = cur;           // (== pre when the lane has no head)
        int f = seen_head ? 1 : 0;
        for (int o = 1; o < 64; o <<= 1) {
            const SaPart y = sa_shfl_up(x, o);
            const int fy = __shfl_up(f, o);
            if (lane >= o && !f) { x = sa_combine(kind, y, x); f = fy; }
        }
        // carry into this lane = the scanned value of the previous lane (and whether a head precedes inside the wave)
        SaPart carry = sa_shfl_up(x, 1);
        int cflag = __shfl_up(f, 1);
        if (lane == 0) { carry = sa_empty(kind); cflag = 0; }
        if (lane == 63) { wpart[wv] = x; wflag[wv] = f; }
        __syncthreads();
        // earlier waves: their tails extend into this wave as long as no head intervenes
        SaPart wcarry = sa_empty(kind);
        int wcf = 0;
        for (int q = 0; q < wv; q++) {
            if (wflag[q]) { wcarry = wpart[q]; wcf = 1; }
            else wcarry = sa_combine(kind, wcarry, wpart[q]);
        }
        if (!cflag) { carry = sa_combine(kind, wcarry, carry); cflag = wcf; }
        // the run that ENDS at this lane's first head (or continues through the lane): carried part + `pre`
        if (seen_head && rows_here > 0) {
            const SaPart done = sa_combine(kind, carry, pre);
            if (cflag) sa_store(S, goff - 1, a, kind, done);                                   // its head lies in this tile: whole here? only if it began here
            else side[((int64_t)tile * S.naggs + a) * 2 + 0] = done;                      // it began in an earlier tile: the tile's leading partial
        }
        // the tile's trailing partial: the last lane's inclusive value
        if (tid == 255) {
            const SaPart incl_all = f ? x : sa_combine(kind, wcarry, x);
            const bool any = f || wcf;
            // (x of the last lane already folds its wave's tail; wcarry folds the earlier waves' when no head sits in between)
            if (tile_heads == 0) side[((int64_t)tile * S.naggs + a) * 2 + 0] = incl_all;  // the whole tile inside one run
            else side[((int64_t)tile * S.naggs + a) * 2 + 1] = any ? incl_all : incl_all;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sorted_groups_kernel(const SortedAgg *__restrict__ Sp, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total,
                                                            SaPart *__restrict__ side, SaLook L) { sorted_groups_body<false>(*Sp, block_off, total, side, L); }
__global__ __launch_bounds__(256) void sorted_groups_kernel_v(const SortedAgg S, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total,
                                                              SaPart *__restrict__ side, SaLook L) { sorted_groups_body<true>(S, block_off, total, side, L); }

// one lane per tile with a head: its last run = its trailing partial + the following tiles without a head + the next tile's leading partial
template <bool ONE>
__device__ __forceinline__ void sorted_fixup_body(const SortedAgg &S, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total, int64_t ntiles,
                                                  const SaPart *__restrict__ side) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const int64_t off = block_off[t], next = t + 1 < ntiles ? block_off[t + 1] : *total;
    if (next == off) return;                       // no head in this tile: some earlier tile's lane walks over it
    const int64_t g = next - 1;                    // the tile's last head's group
    const int na = ONE ? 1 : S.naggs;
    for (int a = 0; a < na; a++) {
        const int kind = ONE ? S.agg_kind[0] : S.agg_kind[a];
        SaPart acc = side[(t * S.naggs + a) * 2 + 1];
        for (int64_t u = t + 1; u < ntiles; u++) {
            const int64_t uo = block_off[u], un = u + 1 < ntiles ? block_off[u + 1] : *total;
            acc = sa_combine(kind, acc, side[(u * S.naggs + a) * 2 + 0]);
            if (un != uo) break;                   // tile u has a head: its leading partial closed the run
        }
        sa_store(S, g, a, kind, acc);
    }
}
__global__ __launch_bounds__(256) void sorted_fixup_kernel(const SortedAgg *__restrict__ Sp, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total, int64_t ntiles,
                                                           const SaPart *__restrict__ side) { sorted_fixup_body<false>(*Sp, block_off, total, ntiles, side); }
__global__ __launch_bounds__(256) void sorted_fixup_kernel_v(const SortedAgg S, const int32_t *__restrict__ block_off, const int64_t *__restrict__ total, int64_t ntiles,
                                                             const SaPart *__restrict__ side) { sorted_fixup_body<true>(S, block_off, total, ntiles, side); }

}  // namespace ph

extern "C" int ph_agg_sink_sorted(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs, int64_t n, int64_t row_base) {
    PH_REQUIRE(a && keys && n >= 0 && nargs >= 0 && nargs <= ph::AGG_MAX_AGGS && (nargs == 0 || args), "ph_agg_sink_sorted: bad arguments");
    if (a->rows_sunk != 0) { ph::set_error("ph_agg_sink_sorted: only the first sink into an empty table"); return PH_EUNSUPPORTED; }
    ph::SortedAgg S{};
    S.nkeys = a->nkeys; S.naggs = a->naggs; S.nargs = nargs; S.n = n; S.row_base = row_base;
    for (int c = 0; c < a->nkeys; c++) {
        PH_REQUIRE(keys[c].type == a->key_types[c], "ph_agg_sink_sorted: key %d has type %d, table was created for %d", c, keys[c].type, a->key_types[c]);
        if (keys[c].validity) { ph::set_error("ph_agg_sink_sorted: NULL-able keys take ph_agg_sink"); return PH_EUNSUPPORTED; }
        S.key[c] = {keys[c].type, keys[c].data, nullptr};
    }
    for (int c = 0; c < nargs; c++) {
        const int t = args[c].type;
        if (t != PH_I32 && t != PH_I64 && t != PH_DEC64 && t != PH_DATE) { ph::set_error("ph_agg_sink_sorted: argument %d has type %d", c, t); return PH_EUNSUPPORTED; }
        S.arg[c] = {t == PH_DATE ? PH_I32 : t, args[c].data, args[c].validity};
    }
    for (int i = 0; i < a->naggs; i++) {
        S.agg_kind[i] = a->aggs[i].kind;
        S.agg_arg[i] = a->aggs[i].arg;
        PH_REQUIRE(a->aggs[i].kind == PH_A_COUNT_STAR || (a->aggs[i].arg >= 0 && a->aggs[i].arg < nargs),
                   "ph_agg_sink_sorted: aggregate %d refers to argument %d of %d", i, a->aggs[i].arg, nargs);
    }
    if (n == 0) return PH_OK;
    ph_ctx *ctx = a->ctx;
    // a run per row at most: capacity for n groups; the slot array is not used by this form (no later sinks)
    int64_t cap = a->cap ? a->cap : next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups));
    while (cap / 2 < n) cap *= 2;
    if (cap != a->cap) PH_CHECK(agg_resize(a, cap, 0, false));
    S.gkeys = a->gkeys; S.gnull = a->gnull; S.sum_lo = a->sum_lo; S.sum_hi = a->sum_hi; S.cnt = a->cnt; S.first_row = a->first_row;
    S.counters = a->counters;
    int *words = nullptr;
    PH_CHECK(ctx->deferred_words(&words));
    S.violation = words + 3;
    const int64_t nb = (n + ph::SA_TILE - 1) / ph::SA_TILE;
    PH_CHECK(ctx->ensure_scratch(ph::round_up(nb * 4, 8) + 64));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(nb * 4, 8));
    ph::SaPart *side = nullptr;   // two partials per tile and aggregate (runs that cross tiles)
    PH_CHECK(ctx->pool_alloc(nb * a->naggs * 2 * (int64_t)sizeof(ph::SaPart) + 64, (void **)&side));
    PH_CHECK(agg_clear(a, false, 0, 8, nullptr, 0));   // the four counter words + the top-k state words
    a->fresh = false;
    ph::SortedAgg *Sd = nullptr;
    int rc = PH_OK;
    const bool one = a->naggs == 1;
    // Up to 8192 tiles (8 M rows) ONE pass: the tiles' first group ids from a look-back inside the groups kernel — two launches and a second read of
    // the keys less (Q3's 298 k rows: -10 us). Above that the heads pass + scan + groups pass: with 58 k tiles in flight every workgroup idles
    // through its look-back's round trips while holding its registers (60 M rows: 0.82 ms in one pass, 0.49 ms in two). PH_STREAM_AGG_TWO_PASS=1 /
    // PH_STREAM_AGG_ONE_PASS=1 force either form (the parity tests run both).
    ph::SaLook L{};
    const bool two_pass = getenv("PH_STREAM_AGG_TWO_PASS") != nullptr, one_pass = getenv("PH_STREAM_AGG_ONE_PASS") != nullptr;
    if (!two_pass && nb < (1ll << 31) && (nb <= 8192 || one_pass)) {
        PH_CHECK(ph::scan_state_acquire(ctx, nb, &L.state, &L.ticket, &L.ticket_base, &L.epoch));
        L.block_off = counts;
        L.total = total;
    }
    if (one) {   // the by-value entries: the one aggregate's argument first, every descriptor index a constant
        if (S.agg_kind[0] != PH_A_COUNT_STAR && S.agg_arg[0] != 0) { S.arg[0] = S.arg[S.agg_arg[0]]; S.agg_arg[0] = 0; }
        if (!L.state) {
            ph::sorted_heads_kernel_v<<<(int)nb, 256, 0, ctx->stream>>>(S, counts);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
            if (rc == PH_OK) rc = ph::exclusive_scan_i32(ctx, counts, nb, total);
        }
        if (rc == PH_OK) {
            ph::sorted_groups_kernel_v<<<(int)nb, 256, 0, ctx->stream>>>(S, counts, total, side, L);
            ph::sorted_fixup_kernel_v<<<(int)((nb + 255) / 256), 256, 0, ctx->stream>>>(S, counts, total, nb, side);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
        }
    } else {
        rc = ctx->pool_alloc((int64_t)sizeof(ph::SortedAgg), (void **)&Sd);
        if (rc == PH_OK && hipMemcpyAsync(Sd, &S, sizeof S, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PH_EHIP;   // (pageable source: staged before the call returns)
        if (rc == PH_OK && !L.state) {
            ph::sorted_heads_kernel<<<(int)nb, 256, 0, ctx->stream>>>(Sd, counts);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
            if (rc == PH_OK) rc = ph::exclusive_scan_i32(ctx, counts, nb, total);
        }
        if (rc == PH_OK) {
            ph::sorted_groups_kernel<<<(int)nb, 256, 0, ctx->stream>>>(Sd, counts, total, side, L);
            ph::sorted_fixup_kernel<<<(int)((nb + 255) / 256), 256, 0, ctx->stream>>>(Sd, counts, total, nb, side);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
        }
    }
    ctx->pool_release(side);   // (stream-ordered reuse)
    if (Sd) ctx->pool_release(Sd);
    if (rc != PH_OK) { ph::set_error("ph_agg_sink_sorted: launch failed"); return rc; }
    ctx->deferred_pending = true;
    a->rows_sunk += n;
    a->sorted_built = true;
    return PH_OK;
}

extern "C" int ph_agg_keys_dev(ph_agg *a, int32_t key_index, void *out_data_dev, uint8_t *out_validity_dev,
                               int64_t capacity, int64_t *ngroups) {
    PH_REQUIRE(a && ngroups && key_index >= 0 && key_index < a->nkeys && capacity >= 0, "ph_agg_keys_dev: bad arguments");
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(a, &ng));
    *ngroups = ng;
    if (ng > capacity) { ph::set_error("ph_agg_keys_dev: %lld groups, room for %lld", (long long)ng, (long long)capacity); return PH_ECAPACITY; }
    if (ng == 0) return PH_OK;
    PH_REQUIRE(out_data_dev, "ph_agg_keys_dev: out_data_dev is NULL");
    ph::agg_keys_kernel<<<(int)std::min<int64_t>((ng + 255) / 256, 256 * 16), 256, 0, a->ctx->stream>>>(a->gkeys, a->gnull, a->nkeys, key_index,
                                                                              a->key_types[key_index], (int)ng, out_data_dev,
                                                                              out_validity_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_agg_values_dev(ph_agg *a, int32_t agg_index, int64_t *out_dev, uint8_t *out_validity_dev, int64_t capacity, int64_t *ngroups) {
    PH_REQUIRE(a && ngroups && agg_index >= 0 && agg_index < a->naggs && capacity >= 0, "ph_agg_values_dev: bad arguments");
    const int kind = a->aggs[agg_index].kind;
    if (kind == PH_A_AVG) { ph::set_error("ph_agg_values_dev: AVG is a quotient the caller owns (sum and count are separate aggregates)"); return PH_EUNSUPPORTED; }
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(a, &ng));
    *ngroups = ng;
    if (ng > capacity) { ph::set_error("ph_agg_values_dev: %lld groups, room for %lld", (long long)ng, (long long)capacity); return PH_ECAPACITY; }
    if (ng == 0) return PH_OK;
    PH_REQUIRE(out_dev, "ph_agg_values_dev: out_dev is NULL");
    int *flag = nullptr;
    PH_CHECK(a->ctx->pool_alloc(16, (void **)&flag));
    PH_HIP(hipMemsetAsync(flag, 0, 4, a->ctx->stream));
    ph::agg_values_kernel<<<(int)std::min<int64_t>((ng + 255) / 256, 256 * 16), 256, 0, a->ctx->stream>>>(a->sum_lo, a->sum_hi, a->cnt, a->naggs, agg_index, kind, (int)ng,
                                                                                (long long *)out_dev, out_validity_dev, flag);
    int wide = 0;
    int rc = hipGetLastError() == hipSuccess ? a->ctx->download(&wide, flag, 4) : PH_EHIP;
    a->ctx->pool_release(flag);
    PH_CHECK(rc);
    if (wide) { ph::set_error("ph_agg_values_dev: a sum does not fit int64"); return PH_EOVERFLOW; }
    return PH_OK;
}

extern "C" int ph_agg_sink_masked(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs,
                                  const int32_t *sel, int64_t n, int32_t positional, int64_t row_base,
                                  uint32_t agg_mask) {
    PH_REQUIRE(a && keys && n >= 0 && nargs >= 0 && nargs <= ph::AGG_MAX_AGGS && (nargs == 0 || args),
               "ph_agg_sink: bad arguments");
    if (a->sorted_built) { ph::set_error("ph_agg_sink: the table was filled by ph_agg_sink_sorted (no hash slots): no further sinks"); return PH_EUNSUPPORTED; }
    ph::AggSinkParams P{};
    P.agg_mask = agg_mask;
    P.nkeys = a->nkeys;
    P.naggs = a->naggs;
    P.nargs = nargs;
    for (int c = 0; c < a->nkeys; c++) {
        PH_REQUIRE(keys[c].type == a->key_types[c], "ph_agg_sink: key %d has type %d, table was created for %d", c, keys[c].type, a->key_types[c]);
        P.key[c] = {keys[c].type, keys[c].data, keys[c].validity};
    }
    bool used[ph::AGG_MAX_AGGS] = {};
    for (int i = 0; i < a->naggs; i++)
        if (((agg_mask >> i) & 1) && a->aggs[i].kind != PH_A_COUNT_STAR && a->aggs[i].arg >= 0 && a->aggs[i].arg < nargs)
            used[a->aggs[i].arg] = true;
    for (int c = 0; c < nargs; c++) {
        if (!used[c]) continue;  // placeholders of count(*) are never read
        int t = args[c].type;
        if (t != PH_I32 && t != PH_I64 && t != PH_DEC64 && t != PH_DATE) { ph::set_error("ph_agg_sink: argument %d has type %d", c, t); return PH_EUNSUPPORTED; }
        P.arg[c] = {t == PH_DATE ? PH_I32 : t, args[c].data, args[c].validity};
    }
    for (int i = 0; i < a->naggs; i++) {
        P.agg_kind[i] = a->aggs[i].kind;
        P.agg_arg[i] = a->aggs[i].arg;
        PH_REQUIRE(!((agg_mask >> i) & 1) || a->aggs[i].kind == PH_A_COUNT_STAR || (a->aggs[i].arg >= 0 && a->aggs[i].arg < nargs),
                   "ph_agg_sink: aggregate %d refers to argument %d of %d", i, a->aggs[i].arg, nargs);
    }
    P.positional = positional;
    P.ngroups = a->counters;
    P.error_flag = a->counters + 1;
    P.need_grow = a->counters + 2;
    P.sel = sel;
    P.n = n;
    P.row_base = row_base;
    static const bool no_bulk = getenv("PH_AGG_NO_BULK") != nullptr;
    // ... and big first sinks whose expected groups overflow the largest LDS table the incremental
    // kernel can have (144 KiB at 70 %): their rows would update the global table one by one with
    // device-scope atomics (3000 groups / 32 M rows: 4.9 ms against 1.5 ms for the bulk build)
    int64_t bulk_min = 32768;
    if (n >= (4ll << 20)) {
        const int sper = agg_spec_entry_bytes(P);
        int t = 64;
        while ((size_t)t * 2 * sper <= 144 * 1024 && t < 8192) t *= 2;
        bulk_min = std::min<int64_t>(bulk_min, (int64_t)t * 7 / 10 + 1);
    }
    if (const char *be = getenv("PH_AGG_BULK_MIN")) bulk_min = atoll(be);
    if (!no_bulk && a->rows_sunk == 0 && a->expected_groups >= bulk_min && n >= 65536) {
        int brc = bulk_sink(a, P, used, n);
        if (brc != PH_EUNSUPPORTED) {
            if (brc == PH_OK) a->rows_sunk += n;
            return brc;
        }
    }
    for (int c = 0; c < nargs; c++) if (used[c]) P.arg_used |= 1u << c;
    P.sel = sel;
    P.n = n;
    P.row_base = row_base;
    if (n == 0) return PH_OK;
    // LDS table: as many entries as fit 32 KiB (at most 1024): state 4 B, first row 8 B, 8 B per
    // key column, 12 B per aggregate
    int per_entry = 12 + 8 * a->nkeys + 12 * a->naggs;
    int slots = 64;
    while (slots * 2 * per_entry <= 32 * 1024 && slots < 1024) slots *= 2;
    P.lds_slots = slots;
    size_t lds = (size_t)slots * per_entry;
    // as many workgroups as are resident at once (static chunk assignment: a workgroup that had to
    // wait for a free CU would double the run time), each alive for the whole call
    int occ = 0;
    void (*kernel)(ph::AggSinkParams) = a->nkeys == 1 ? ph::agg_sink_kernel<1> : a->nkeys == 2 ? ph::agg_sink_kernel<2>
                                        : a->nkeys == 3 ? ph::agg_sink_kernel<3> : ph::agg_sink_kernel<4>;
    // sinks large enough to repay a one-time compile (~0.5 s per shape and process) run the kernel
    // specialised for this shape; small ones, and everything when hiprtc is unavailable, the generic one
    ph::JitKernel spec{};
    bool have_spec = false;
    int spec_threads = 256;
    {
        ph::AggSinkParams Q = P;   // the shape is complete at this point (pointers do not enter the key)
        for (int c = 0; c < nargs; c++) if (used[c]) Q.arg_used |= 1u << c;
        // 512-thread workgroups share one LDS table among 8 waves: half of the flushes (every workgroup pays
        // one find-or-create + state update of device-scope atomics per group it saw: 800 workgroups x 175
        // groups cost more than the 3.3 M rows of Q9's aggregate themselves). Measured, two int32 keys /
        // 175 groups / one sum: 3.3 M rows 100 -> 78 us, 32 M rows 370 -> 309 us; 1024 threads (one
        // workgroup per CU by registers) 74 / 409 us.
        const char *te = getenv("PH_AGG_T");
        spec_threads = te ? atoi(te) : 512;
        if (spec_threads != 256 && spec_threads != 512 && spec_threads != 1024) spec_threads = 256;
        // The specialised kernel's table is static, so it may pass 64 KiB: when the creator expects more
        // groups than half of the 32 KiB table holds, the table grows (up to 128 KiB: one 1024-thread
        // workgroup per CU) — rows whose group finds no room in LDS go to the global table one by
        // one (1000 groups in a 256-entry table: 4.3 ms per 32 M rows).
        int sslots = slots;
        const int sper = agg_spec_entry_bytes(Q);
        size_t budget = 32 * 1024;
        if (const char *le = getenv("PH_AGG_LDS_KB")) budget = (size_t)std::max(8, std::min(atoi(le), 144)) * 1024;
        auto fit = [&](size_t b) { int t = 64; while ((size_t)t * 2 * sper <= b && t < 8192) t *= 2; return t; };
        sslots = fit(budget);
        if (!getenv("PH_AGG_LDS_KB"))   // the smallest table the expected groups fill to 70 % at most
            for (size_t b = budget; b <= 144 * 1024; b = b < 128 * 1024 ? b * 2 : b + 16 * 1024)
                if ((int64_t)fit(b) * 7 / 10 >= a->expected_groups) { budget = b; sslots = fit(b); break; }
        if (budget > 64 * 1024 && !te) spec_threads = 1024;
        have_spec = n >= (1 << 20) && agg_spec_kernel(a->ctx, Q, spec_threads, sslots, (size_t)sslots * sper, &spec) == PH_OK;
        if (have_spec) { slots = sslots; P.lds_slots = slots; lds = (size_t)slots * sper; }
    }
    const int threads = have_spec ? spec_threads : 256;
    if (have_spec) PH_HIP(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spec.fn, threads, 0));
    else PH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds));
    const int occ_raw = occ;
    occ = std::max(1, std::min(occ, 4));
    if (const char *oe = getenv("PH_AGG_OCC")) occ = std::max(1, std::min(atoi(oe), 8));   // tuning knob (workgroups per CU)
    // small inputs are latency bound (a new group costs a chain of dependent HBM atomics): give
    // every thread one row before giving any thread a second one
    const int64_t resident = (int64_t)a->ctx->cu_count * occ;
    int chunk = threads;
    int max_chunk = have_spec ? 4 * ph::AGG_CHUNK : ph::AGG_CHUNK;   // fewer growth checks (two barriers + one device-scope read each)
    if (const char *ce = getenv("PH_AGG_CHUNK")) max_chunk = std::max(256, std::min(atoi(ce), 1 << 16));
    while (chunk < max_chunk && (n + chunk - 1) / chunk > resident) chunk *= 2;
    P.chunk = chunk;
    const int64_t nchunks = (n + chunk - 1) / chunk;
    // staged partials are int64 sums of |v| < 2^40 and u32 counts: at most 2^22 rows per workgroup
    const int64_t min_grid = (n + (1ll << 22) - 1) >> 22;
    const int grid = (int)std::max<int64_t>(std::min<int64_t>(nchunks, resident), min_grid);
    // this call creates at most n groups in total, so a slack of n is always enough
    P.slack = std::min<long long>((long long)grid * (chunk + slots), n);
    int *progress = nullptr;
    PH_CHECK(a->ctx->pool_alloc((int64_t)grid * 4, (void **)&progress));
    P.progress = progress;
    int rc = PH_OK;
    bool new_table = false;
    // First sink into an empty table of up to 8 M rows: size the table so that growth is impossible
    // (capacity/2 > rows). The group count and the need-grow flag then never cross PCIe — two host
    // round trips (~50 us of idle GPU) against one larger memset of the slot array (32 MiB: 9 us).
    if (rc == PH_OK && a->rows_sunk == 0 && n <= (8ll << 20)) {
        int64_t want = a->cap ? a->cap : next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups));
        while (want / 2 <= n) want *= 2;
        if (want != a->cap) { rc = agg_resize(a, want, 0, false); new_table = true; }
    }
    if (rc == PH_OK && a->cap == 0) {
        rc = agg_resize(a, next_pow2(std::max<int64_t>(4096, 2 * a->expected_groups)), 0, false);
        new_table = true;
    }
    // one clearing launch: a new table's slots, the counters (all four on first use, else the need-grow
    // word), the progress words
    if (rc == PH_OK) rc = agg_clear(a, new_table, a->fresh ? 0 : 2, a->fresh ? 8 : 1, progress, grid);
    a->fresh = false;
    // the table cannot hold more groups than rows were sunk into it: while that bound plus this
    // call's rows fits, no growth is possible and neither the group count nor the need-grow flag
    // has to cross PCIe
    const bool sure = a->gcap - a->rows_sunk > n;
    if (sure) P.slack = -1;  // the in-kernel check can never be needed: switch it off
    static const bool timing = getenv("PH_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    const double t_b = now();
    double t_c = 0, t_d = 0, t_e = 0;
    while (rc == PH_OK) {
        int64_t ng = a->rows_sunk;
        if (!sure) {
            if (a->rows_sunk == 0) ng = 0;   // an empty table: nothing to ask the device
            else if ((rc = ph_agg_group_count(a, &ng)) != PH_OK) break;
            int64_t cap = a->cap;
            while (cap / 2 - ng <= P.slack) cap *= 2;   // gcap = cap/2 must exceed the in-flight rows
            if (cap != a->cap && (rc = agg_resize(a, cap, (int)ng)) != PH_OK) break;
        }
        P.slots = a->slots;
        P.mask = (uint64_t)a->cap - 1;
        P.gkeys = a->gkeys; P.gnull = a->gnull; P.sum_lo = a->sum_lo; P.sum_hi = a->sum_hi;
        P.cnt = a->cnt; P.first_row = a->first_row; P.gcap = a->gcap;
        t_c = now();
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (timing) { (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1); (void)hipEventRecord(ev0, a->ctx->stream); }
        if (have_spec) {
            ph::AggSinkParams copy = P;
            size_t size = sizeof copy;
            void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
            if (hipModuleLaunchKernel(spec.fn, (unsigned)grid, 1, 1, (unsigned)threads, 1, 1, 0, a->ctx->stream, nullptr, config) != hipSuccess) { rc = PH_EHIP; break; }
        } else {
            kernel<<<grid, 256, lds, a->ctx->stream>>>(P);
            if (hipGetLastError() != hipSuccess) { rc = PH_EHIP; break; }
        }
        if (timing) {
            (void)hipEventRecord(ev1, a->ctx->stream);
            (void)hipEventSynchronize(ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            fprintf(stderr, "[ph_agg_sink] kernel %.1f us by events (grid %d, lds %zu, chunk %d, occupancy %d)\n", ms * 1e3, grid, lds, chunk, occ_raw);
            (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
        }
        t_d = now();
        if (sure || a->gcap - ng - n > P.slack) break;  // even all-new groups cannot trip the check
        int grow = 0;
        if ((rc = a->ctx->download(&grow, a->counters + 2, 4)) != PH_OK || !grow) break;
        if (hipMemsetAsync(a->counters + 2, 0, 4, a->ctx->stream) != hipSuccess) rc = PH_EHIP;
    }
    t_e = now();
    if (timing) fprintf(stderr, "[ph_agg_sink] spec lookup %.0f us, resize/count %.0f us, launch %.0f us, after %.0f us (spec=%d sure=%d)\n",
                        t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, (int)have_spec, (int)sure);
    a->ctx->pool_release(progress);
    if (rc == PH_EHIP) ph::set_error("ph_agg_sink: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    if (rc != PH_OK) return rc;
    a->rows_sunk += n;
    return PH_OK;
}

// One host round trip: the records of up to max_groups groups are packed on the device (the kernel
// reads the group count there) and header + records come back in one copy when they fit the mailbox
// (64 KiB; otherwise the header first, then exactly the records that exist).
// the groups ids_dev[0 .. meta_dev[0]) (ids_dev == nullptr: all of them, meta_dev = the table's counters) to the host, first-seen order
static int agg_fetch_impl(ph_agg *a, const int *ids_dev, const int *meta_dev, int64_t max_groups, int64_t *ngroups, int64_t *first_row, int64_t *keys,
                          uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    ph_ctx *cx = a->ctx;
    const size_t rec = 2 + (size_t)a->nkeys + 3 * (size_t)a->naggs;
    const int64_t cap = std::min<int64_t>(max_groups, a->gcap);
    unsigned long long *pack = nullptr;
    PH_CHECK(cx->pool_alloc((int64_t)(2 + (size_t)cap * rec) * 8, (void **)&pack));
    ph::agg_pack_kernel<<<(int)std::max<int64_t>(1, std::min<int64_t>((cap + 255) / 256, 1024)), 256, 0, cx->stream>>>(
        ids_dev, meta_dev, a->counters, (int)cap, a->nkeys, a->naggs, a->first_row, a->gkeys, a->gnull, a->sum_lo, a->sum_hi, a->cnt, pack);
    std::vector<unsigned long long> host(2 + (size_t)cap * rec);
    int rc = hipGetLastError() == hipSuccess ? PH_OK : PH_EHIP;
    const bool one_copy = (int64_t)host.size() * 8 <= (64 << 10);
    if (rc == PH_OK) rc = cx->download(host.data(), pack, one_copy ? (int64_t)host.size() * 8 : 16);
    int64_t ng = 0;
    if (rc == PH_OK) {
        const int *h = reinterpret_cast<const int *>(host.data());   // {groups to fetch, -, the table's group count, its error flag}
        ng = h[0];
        *ngroups = ng;
        if (h[3]) { ph::set_error("ph_agg: device table error flag %d", h[3]); rc = PH_EHIP; }
        else if (ng > max_groups) { ph::set_error("ph_agg_fetch: %lld groups, room for %lld", (long long)ng, (long long)max_groups); rc = PH_ECAPACITY; }
        else if (!one_copy && ng > 0) rc = cx->download(host.data() + 2, pack + 2, (int64_t)((size_t)ng * rec) * 8);
    }
    cx->pool_release(pack);
    if (rc != PH_OK || ng == 0) return rc;
    size_t na = (size_t)std::max(a->naggs, 1), g = (size_t)ng;
    std::vector<long long> fr(g);
    std::vector<unsigned long long> gk(g * a->nkeys), lo(g * na), cn(g * na);
    std::vector<long long> hi(g * na);
    std::vector<unsigned> gn(g);
    for (size_t i = 0; i < g; i++) {
        const unsigned long long *o = host.data() + 2 + i * rec;
        fr[i] = (long long)o[0];
        gn[i] = (unsigned)o[1];
        for (int c = 0; c < a->nkeys; c++) gk[i * a->nkeys + c] = o[2 + c];
        for (int q = 0; q < a->naggs; q++) {
            lo[i * na + q] = o[2 + a->nkeys + q];
            hi[i * na + q] = (long long)o[2 + a->nkeys + a->naggs + q];
            cn[i * na + q] = o[2 + a->nkeys + 2 * a->naggs + q];
        }
    }
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

extern "C" int ph_agg_fetch(ph_agg *a, int64_t max_groups, int64_t *ngroups, int64_t *first_row, int64_t *keys,
                            uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && ngroups && max_groups >= 0, "ph_agg_fetch: bad arguments");
    *ngroups = 0;
    if (a->cap == 0) return PH_OK;   // nothing sunk yet
    return agg_fetch_impl(a, nullptr, a->counters, max_groups, ngroups, first_row, keys, key_null, sum_lo, sum_hi, count);
}

// HAVING on the device: the aggregate values as columns (ph_agg_values_dev), the conjuncts as a chain of ph_filter_select over the group
// ids, the surviving groups packed and fetched — instead of every group travelling to the host for the comparison
extern "C" int ph_agg_fetch_where(ph_agg *a, int32_t nconj, const int32_t *agg_index, const int32_t *op, const ph_const *k, const int32_t *value_scale,
                                  int64_t max_groups, int64_t *ngroups, int64_t *first_row, int64_t *keys, uint8_t *key_null, uint64_t *sum_lo,
                                  int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && ngroups && max_groups >= 0 && nconj >= 1 && agg_index && op && k && value_scale, "ph_agg_fetch_where: bad arguments");
    *ngroups = 0;
    if (a->cap == 0) return PH_OK;
    ph_ctx *cx = a->ctx;
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(a, &ng));
    if (ng == 0) return PH_OK;
    std::vector<void *> tmp;
    auto release = [&]() { for (void *q : tmp) cx->pool_release(q); };
    auto alloc = [&](int64_t bytes, void **out) { int rc = cx->pool_alloc(bytes, out); if (rc == PH_OK) tmp.push_back(*out); return rc; };
    const int32_t *sel = nullptr;
    int64_t cnt = ng;
    int rc = PH_OK;
    for (int32_t c = 0; c < nconj && rc == PH_OK && cnt > 0; c++) {
        if (agg_index[c] < 0 || agg_index[c] >= a->naggs) { ph::set_error("ph_agg_fetch_where: aggregate %d out of range", agg_index[c]); rc = PH_EINVAL; break; }
        void *vals = nullptr, *valid = nullptr, *out = nullptr;
        if ((rc = alloc(ng * 8, &vals)) != PH_OK || (rc = alloc((ng + 7) / 8 + 64, &valid)) != PH_OK || (rc = alloc(cnt * 4, &out)) != PH_OK) break;
        int64_t n2 = 0;
        if ((rc = ph_agg_values_dev(a, agg_index[c], (int64_t *)vals, (uint8_t *)valid, ng, &n2)) != PH_OK) break;
        ph_col v{};
        const int kind = a->aggs[agg_index[c]].kind;
        v.type = (kind == PH_A_COUNT || kind == PH_A_COUNT_STAR) ? PH_DEC64 : PH_DEC64;   // values travel as int64 at the argument's scale
        v.scale = value_scale[c];
        v.data = vals; v.validity = (const uint8_t *)valid;
        int64_t m = 0;
        ph_const kc = k[c];
        if (kc.type == PH_I32) {   // an INTEGER literal against a DECIMAL / HUGEINT value is cast to the value's type by the binder (DecimalSizeCheck + tryCastInt32ToDecimal)
            kc.type = PH_DEC64; kc.scale = v.scale;
            for (int s = 0; s < v.scale; s++) kc.i *= 10;
        }
        rc = ph_filter_select(cx, &v, ng, op[c], &kc, sel, cnt, (int32_t *)out, &m);
        sel = (const int32_t *)out;
        cnt = m;
    }
    if (rc != PH_OK) { release(); return rc; }
    if (cnt == 0) { release(); return PH_OK; }
    if (cnt > max_groups) { release(); *ngroups = cnt; ph::set_error("ph_agg_fetch_where: %lld groups, room for %lld", (long long)cnt, (long long)max_groups); return PH_ECAPACITY; }
    int *meta = nullptr;
    if ((rc = alloc(16, (void **)&meta)) != PH_OK) { release(); return rc; }
    const int m2[2] = {(int)cnt, 0};
    rc = ph_dev_upload(cx, meta, m2, 8);
    if (rc == PH_OK) rc = agg_fetch_impl(a, sel, meta, max_groups, ngroups, first_row, keys, key_null, sum_lo, sum_hi, count);
    release();
    return rc;
}

extern "C" int ph_agg_finalize(ph_agg *a, int64_t max_groups, int64_t *first_row, int64_t *keys,
                               uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && max_groups >= 0, "ph_agg_finalize: bad arguments");
    int64_t ng = 0;
    return ph_agg_fetch(a, max_groups, &ng, first_row, keys, key_null, sum_lo, sum_hi, count);
}

extern "C" int ph_agg_topk(ph_agg *a, int32_t agg_index, int32_t descending, int64_t k, int64_t max_groups,
                           int64_t *n_out, int64_t *first_row, int64_t *keys, uint8_t *key_null,
                           uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count) {
    PH_REQUIRE(a && n_out && k >= 0 && max_groups >= 0 && agg_index >= 0 && agg_index < a->naggs,
               "ph_agg_topk: bad arguments");
    int kind = a->aggs[agg_index].kind;
    if (kind != PH_A_SUM && kind != PH_A_MIN && kind != PH_A_MAX && kind != PH_A_COUNT && kind != PH_A_COUNT_STAR) {
        // AVG orders by sum/count, COUNT(DISTINCT) by a side table: neither is a word of the group's state
        ph::set_error("ph_agg_topk: aggregate %d (kind %d) is not ordered by its sum / min / max / count word; use ph_agg_finalize", agg_index, kind);
        return PH_EUNSUPPORTED;
    }
    *n_out = 0;
    if (k == 0 || max_groups == 0 || a->cap == 0) return PH_OK;
    ph_ctx *ctx = a->ctx;
    // The group count stays on the device: every kernel reads it there, grids are sized from the
    // table's capacity, and the header + first records come back in one copy (one host round trip
    // for the whole call when <= 256 groups qualify, which is the LIMIT case this is for).
    const int cap = (int)std::min<int64_t>(max_groups, a->gcap);
    const size_t na = (size_t)a->naggs, nk = (size_t)a->nkeys, rec = 2 + nk + 3 * na;
    int *ids = nullptr, *cand_ids = nullptr;
    unsigned long long *cand_keys = nullptr;
    int *state = nullptr;                 // [0] candidate count [1] ticket [2] meta: qualifying count [3] meta: too-wide flag
    unsigned long long *pack = nullptr;   // header (2 words) + cap records
    PH_CHECK(ctx->pool_alloc((int64_t)std::max(cap, 1) * 4, (void **)&ids));
    PH_CHECK(ctx->pool_alloc(a->gcap * 4, (void **)&cand_ids));
    PH_CHECK(ctx->pool_alloc(a->gcap * 8, (void **)&cand_keys));
    const bool own_state = a->topk_state_used;   // the words beside the counters are clean once: zeroed by the table's first clearing launch
    if (own_state) {
        PH_CHECK(ctx->pool_alloc(64, (void **)&state));
        PH_HIP(hipMemsetAsync(state, 0, 64, ctx->stream));
    } else {
        state = a->counters + 4;
        a->topk_state_used = true;
    }
    PH_CHECK(ctx->pool_alloc((int64_t)(2 + (size_t)cap * rec) * 8, (void **)&pack));
    int *meta = state + 2;
    const int tg = (int)((a->gcap + ph::TOPK_CHUNK - 1) / ph::TOPK_CHUNK);   // grid from the capacity: the count stays on the device
    unsigned long long *wg_kth = nullptr;   // every workgroup's own k-th best key (the last workgroup prunes the candidates with their minimum)
    PH_CHECK(ctx->pool_alloc((int64_t)tg * 8 + 64, (void **)&wg_kth));
    ph::topk_select_kernel<<<tg, 256, 0, ctx->stream>>>(a->sum_lo, a->sum_hi, a->cnt, a->naggs, agg_index, a->counters, descending,
                                                        kind == PH_A_SUM ? 1 : (kind == PH_A_COUNT || kind == PH_A_COUNT_STAR) ? 2 : 0, (long long)k, cand_ids, cand_keys, state, state + 1,
                                                        ids, meta, cap, wg_kth);
    ph::agg_pack_kernel<<<std::max(1, std::min((cap + 255) / 256, 64)), 256, 0, ctx->stream>>>(
        ids, meta, a->counters, cap, a->nkeys, a->naggs, a->first_row, a->gkeys, a->gnull, a->sum_lo, a->sum_hi, a->cnt, pack);
    int rc = hipGetLastError() == hipSuccess ? PH_OK : PH_EHIP;
    const size_t first_recs = std::min<size_t>((size_t)cap, 256);
    std::vector<unsigned long long> host(2 + first_recs * rec);
    if (rc == PH_OK) rc = ctx->download(host.data(), pack, (int64_t)host.size() * 8);
    int m[4] = {0, 0, 0, 0};
    memcpy(m, host.data(), sizeof m);
    if (rc == PH_OK && m[3]) { ph::set_error("ph_agg: device table error flag %d", m[3]); rc = PH_EHIP; }
    if (rc == PH_OK && m[1]) { ph::set_error("ph_agg_topk: a sum does not fit int64; use ph_agg_finalize"); rc = PH_EOVERFLOW; }
    if (rc == PH_OK && m[0] > cap) { ph::set_error("ph_agg_topk: %d qualifying groups, room for %d", m[0], cap); rc = PH_ECAPACITY; }
    const size_t n = rc == PH_OK ? (size_t)m[0] : 0;
    if (rc == PH_OK && n > first_recs) {  // many ties / a large k: fetch the remaining records
        host.resize(2 + n * rec);
        rc = ctx->download(host.data() + 2 + first_recs * rec, pack + 2 + first_recs * rec, (int64_t)((n - first_recs) * rec) * 8);
    }
    if (own_state) ctx->pool_release(state);
    ctx->pool_release(ids);
    ctx->pool_release(cand_ids);
    ctx->pool_release(cand_keys);
    ctx->pool_release(pack);
    ctx->pool_release(wg_kth);
    if (rc != PH_OK) return rc;
    struct Row { long long fr; std::vector<unsigned long long> k, lo, cn; std::vector<long long> hi; unsigned null; };
    std::vector<Row> rows(n);
    for (size_t i = 0; i < n; i++) {
        const unsigned long long *o = host.data() + 2 + i * rec;
        Row &r = rows[i];
        r.fr = (long long)o[0];
        r.null = (unsigned)o[1];
        r.k.assign(o + 2, o + 2 + nk);
        r.lo.assign(o + 2 + nk, o + 2 + nk + na);
        r.hi.assign((const long long *)(o + 2 + nk + na), (const long long *)(o + 2 + nk + 2 * na));
        r.cn.assign(o + 2 + nk + 2 * na, o + 2 + nk + 3 * na);
    }
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
