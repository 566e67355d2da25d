// Hash join: ph_join_build / ph_join_probe_inner / ph_join_probe_mark.
//
// Replaces JoinHashTable.Build -> Finalize/InsertHashesLoop and Probe -> Scan.NextInnerJoin /
// ScanKeyMatches (reference pkg/compute/join_table.go:85-336, join_scan.go:30-300,
// util_match.go:25-301). Same table shape as the reference: a bucket-head table with
// cap = max(nextpow2(2n), 1024) and a per-row `next` link holding the previous head (the reference
// stores it in the row's hash slot, join_table.go:268-288); heads are swapped in with atomicExch so
// a whole build batch inserts in one launch. Rows with a NULL key never enter the table and never
// probe (prepareKeys / filterNullValues, :152-195).
// Probe output is deterministic: pairs ordered by probe position, then by chain order; a count
// pass, a scan over workgroup totals and a write pass place them (no atomics on the output).
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

constexpr int JOIN_MAX_KEYS = 4;

struct JoinCol {
    int type;
    const void *data;
    const uint8_t *validity;
};

struct JoinSide {
    int nkeys;
    JoinCol key[JOIN_MAX_KEYS];
    const int32_t *sel;  // position -> row id (NULL = identity)
    int64_t n;
};

__device__ __forceinline__ unsigned long long jkey(const JoinCol &c, int64_t r) {
    switch (c.type) {
    case PH_I32: case PH_DATE: return (unsigned long long)(long long)((const int32_t *)c.data)[r];
    case PH_CODE8: return ((const uint8_t *)c.data)[r];
    default: return (unsigned long long)((const int64_t *)c.data)[r];
    }
}

__device__ __forceinline__ bool load_keys(const JoinSide &S, int64_t r, unsigned long long *k, uint64_t *h) {
    uint64_t hh = 0x9e3779b97f4a7c15ULL;
#pragma unroll
    for (int c = 0; c < JOIN_MAX_KEYS; c++) {
        k[c] = 0;
        if (c < S.nkeys) {
            if (!bit_valid(S.key[c].validity, r)) return false;
            k[c] = jkey(S.key[c], r);
            hh = mix64(hh ^ k[c]);
        }
    }
    *h = hh;
    return true;
}

// Bloom bitmap (two bits per key inside ONE 32-bit word, >= 16 bits per build key, ~1 % false
// positives; only for build sides small enough that it stays cache-resident): a selective probe
// (Q3: 1 % of the probe rows match) rejects most rows with one cached 4-byte read instead of a
// random read of the 4n-entry head table — every false positive costs one to four random HBM
// reads in the chain walk, which is what bounds the probe.
struct Bloom {
    unsigned *bits;       // NULL = no filter
    uint64_t word_mask;   // number of 32-bit words - 1
    // word index = (partition << hi_shift) | ((h >> 34) & inner_mask), partition = (h >> pshift) & pmask:
    // the partitioned build gives every bucket-range partition its own contiguous slice of the
    // bitmap so one workgroup can build both in LDS; pmask = 0 (inner_mask = word_mask) otherwise
    uint32_t pmask, pshift, hi_shift;
    uint64_t inner_mask;
    // small build sides (<= 256 K keys) also get a 1 Mbit single-hash bitmap that a probe workgroup
    // copies into LDS (128 KiB): it rejects most non-matching probes on the CU, so only the
    // survivors read the bitmap above through L2 (whose request rate bounded the candidate kernel)
    unsigned *coarse;
};

constexpr int CO_WORDS = 32768;   // 1 Mbit
static int g_cu_count = 256;       // set from the context before the coarse kernel is launched
__device__ __forceinline__ unsigned coarse_bit(uint64_t h) { return (unsigned)(h >> 40) & (CO_WORDS * 32 - 1); }

__device__ __forceinline__ unsigned bloom_mask(uint64_t b) { return (1u << (b & 31)) | (1u << ((b >> 5) & 31)); }

__device__ __forceinline__ uint64_t bloom_word(const Bloom &bl, uint64_t h) {
    return (((h >> bl.pshift) & bl.pmask) << bl.hi_shift) | ((h >> 34) & bl.inner_mask);
}

__device__ __forceinline__ bool bloom_maybe(const Bloom &bl, uint64_t h) {
    if (!bl.bits) return true;
    const unsigned m = bloom_mask(h >> 24);  // bits disjoint from the low bits that pick the bucket
    return (bl.bits[bloom_word(bl, h)] & m) == m;
}

__global__ __launch_bounds__(256) void join_build_kernel(JoinSide B, int32_t *__restrict__ head, uint64_t mask,
                                                         int32_t *__restrict__ next, int *__restrict__ count,
                                                         Bloom bl) {
    int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B.n; i += (int64_t)gridDim.x * 256) {
        int64_t r = B.sel ? B.sel[i] : i;
        unsigned long long k[JOIN_MAX_KEYS];
        uint64_t h;
        if (!load_keys(B, r, k, &h)) { next[i] = -2; continue; }  // NULL key: not inserted
        next[i] = atomicExch(&head[h & mask], (int32_t)i);        // head insertion
        if (bl.bits) {
            atomicOr(&bl.bits[bloom_word(bl, h)], bloom_mask(h >> 24));
            if (bl.coarse) atomicOr(&bl.coarse[coarse_bit(h) >> 5], 1u << (coarse_bit(h) & 31));
        }
        local++;
    }
    // one counter update per workgroup: same-address device atomics serialise at ~10 ns each
    __shared__ int s_local[4];
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) s_local[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0 && s_local[0] + s_local[1] + s_local[2] + s_local[3]) atomicAdd(count, s_local[0] + s_local[1] + s_local[2] + s_local[3]);
}

// clears what a build needs cleared in ONE launch: head table (-1) and Bloom bitmap of the atomic
// build, the coarse bitmap of tiny build sides, the inserted-row counter
__global__ __launch_bounds__(256) void join_init_kernel(int32_t *__restrict__ head, int64_t cap, unsigned *__restrict__ bits,
                                                        int64_t words, unsigned *__restrict__ coarse, int *__restrict__ count) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    if (head) {
        int4 *h4 = reinterpret_cast<int4 *>(head);   // cap is a power of two >= 1024
        for (int64_t i = t; i < cap / 4; i += step) h4[i] = make_int4(-1, -1, -1, -1);
    }
    if (bits) {
        uint4 *b4 = reinterpret_cast<uint4 *>(bits);  // words is a power of two >= 2048
        for (int64_t i = t; i < words / 4; i += step) b4[i] = make_uint4(0, 0, 0, 0);
    }
    if (coarse) for (int64_t i = t; i < CO_WORDS; i += step) coarse[i] = 0;
    if (t < 4) count[t] = 0;
}

// ---- partitioned build (no global atomics). Scattered device atomics run at ~20 G/s on this
// part (they execute at the memory side, one 64-B request each), which bounded the atomicExch
// build at ~10 G rows/s with a bitmap and ~20 G rows/s without. Here rows are first grouped by
// the top bits of their bucket index (count per workgroup -> scan -> scatter of (position, hash),
// plain stores), then ONE workgroup per partition links its rows into a 64 KiB head slice and
// its slice of the Bloom bitmap in LDS (ds atomics) and writes both out whole.
constexpr int PB_SLICE_LOG = 14;              // head entries per partition: 16384 = 64 KiB of LDS
constexpr int PB_SLICE = 1 << PB_SLICE_LOG;
constexpr int PB_MAX_PARTS = 4096;

template <int KW> __device__ __forceinline__ unsigned long long load_kw(const void *col, int64_t r) {
    return KW == 4 ? (unsigned long long)(long long)((const int32_t *)col)[r]
           : KW == 1 ? (unsigned long long)((const uint8_t *)col)[r] : ((const unsigned long long *)col)[r];
}

// streaming form: the probe columns are read once, front to back; marking the loads non-temporal
// keeps them from evicting the Bloom bitmap (4 MiB for Q3's 1.46 M order keys: the size of one
// XCD's L2) that every surviving row reads at random
template <int KW> __device__ __forceinline__ unsigned long long load_kw_nt(const void *col, int64_t r) {
    return KW == 4 ? (unsigned long long)(long long)__builtin_nontemporal_load((const int32_t *)col + r)
           : KW == 1 ? (unsigned long long)__builtin_nontemporal_load((const uint8_t *)col + r)
                     : __builtin_nontemporal_load((const unsigned long long *)col + r);
}

// The vectorised candidate kernels: a wave covers 512 consecutive probe rows in G = 8/R groups of
// 64*R rows; in a group a lane owns R = 16/KW CONSECUTIVE rows, so its key read is ONE 16-byte
// non-temporal load and the wave's load instruction covers 1 KiB without gaps (a lane that owned 8
// consecutive 8-byte keys read them with four loads at a 64-byte stride: every sector was fetched four
// times through the non-temporal path and the kernel ran twice as long as the scalar form).
// Slot s = g*R + e of a lane is local row (g*64 + lane)*R + e of the wave's 512.
typedef int dc_v4i __attribute__((ext_vector_type(4)));
typedef int dc_v2i __attribute__((ext_vector_type(2)));
typedef long long dc_v2l __attribute__((ext_vector_type(2)));

template <int W, int R>   // R consecutive elements of width W (4, 8 signed; 1 unsigned) starting at element i0 (i0 % R == 0)
__device__ __forceinline__ void dc_loadr(const void *col, int64_t i0, long long *v) {
    if (W == 8) {
        const dc_v2l a = __builtin_nontemporal_load(reinterpret_cast<const dc_v2l *>((const long long *)col + i0));
        v[0] = a.x; v[1] = a.y;
        if (R == 4) {   // an 8-byte filter column beside 4-byte keys: two loads (32-byte stride, half of each used per instruction)
            const dc_v2l b = __builtin_nontemporal_load(reinterpret_cast<const dc_v2l *>((const long long *)col + i0) + 1);
            v[2] = b.x; v[3] = b.y;
        }
    } else if (W == 4 && R == 4) {
        const dc_v4i a = __builtin_nontemporal_load(reinterpret_cast<const dc_v4i *>((const int32_t *)col + i0));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if (W == 4) {   // R == 2
        const dc_v2i a = __builtin_nontemporal_load(reinterpret_cast<const dc_v2i *>((const int32_t *)col + i0));
        v[0] = a.x; v[1] = a.y;
    } else if (R == 4) {   // W == 1
        const unsigned x = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>((const uint8_t *)col + i0));
        v[0] = x & 0xff; v[1] = (x >> 8) & 0xff; v[2] = (x >> 16) & 0xff; v[3] = x >> 24;
    } else {
        const unsigned x = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>((const uint8_t *)col + i0));
        v[0] = x & 0xff; v[1] = x >> 8;
    }
}

template <int W>
__device__ __forceinline__ long long dc_load1(const void *col, int64_t i) {
    return W == 4 ? (long long)((const int32_t *)col)[i] : W == 8 ? ((const long long *)col)[i] : (long long)((const uint8_t *)col)[i];
}

// keys (and the pushed-down range filter) of a lane's 8 slots; `full` = the whole 2048-row block is inside n
template <int KW, int WK, int NK>
__device__ __forceinline__ void dc_block_keys(const void *keycol, const void *keycol2, const void *wdata, long long wlo, long long whi,
                                              int64_t wave_row0, int lane, bool full, bool have, int64_t n, long long (&k)[8],
                                              long long (&k2)[8], bool (&ok)[8]) {
    constexpr int R = 16 / KW, G = 8 / R;
    constexpr int WW = WK == 1 ? 4 : WK == 2 ? 8 : 1;
    if (full) {
        if (WK != 0) {
            long long w[8];
#pragma unroll
            for (int g = 0; g < G; g++) dc_loadr<WW, R>(wdata, wave_row0 + (int64_t)(g * 64 + lane) * R, &w[g * R]);
#pragma unroll
            for (int s = 0; s < 8; s++) ok[s] = w[s] >= wlo && w[s] <= whi;
        } else {
#pragma unroll
            for (int s = 0; s < 8; s++) ok[s] = true;
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            dc_loadr<KW, R>(keycol, wave_row0 + (int64_t)(g * 64 + lane) * R, &k[g * R]);
            if (NK == 2) dc_loadr<KW, R>(keycol2, wave_row0 + (int64_t)(g * 64 + lane) * R, &k2[g * R]);
        }
        if (NK != 2) {
#pragma unroll
            for (int s = 0; s < 8; s++) k2[s] = 0;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int64_t i = wave_row0 + (int64_t)((s / R) * 64 + lane) * R + (s % R);
            ok[s] = have && i < n;
            const int64_t ic = ok[s] ? i : 0;
            if (WK != 0) { const long long w = dc_load1<WW>(wdata, ic); ok[s] = ok[s] && w >= wlo && w <= whi; }
            k[s] = dc_load1<KW>(keycol, ic);
            k2[s] = NK == 2 ? dc_load1<KW>(keycol2, ic) : 0;
        }
    }
}

// rank of a lane's surviving slots inside its wave in ascending row order (group, lane, element):
// first[g] = survivors of the wave that precede this lane's slots of group g; returns the wave's total
template <int R>
__device__ __forceinline__ int dc_wave_ranks(const bool (&take)[8], int (&first)[8 / R]) {
    constexpr int G = 8 / R;
    int before = 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        int below = 0, tot = 0;
#pragma unroll
        for (int e = 0; e < R; e++) {
            const unsigned long long bal = __ballot(take[g * R + e]);
            below += __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
            tot += __popcll(bal);
        }
        first[g] = before + below;
        before += tot;
    }
    return before;
}

// Straight-line forms of the two partition passes for the common build shape (one or two key
// columns of one width, no NULL keys): PU rows per thread, their selection / key reads issued
// together (the generic kernels below pay two or three dependent memory latencies per row).
constexpr int PU = 4;

template <int KW, int NK, bool SEL>
__device__ __forceinline__ void part_hashes(const void *k0, const void *k1, const int32_t *sel, int64_t base, int64_t i1,
                                            uint64_t (&h)[PU], bool (&ok)[PU]) {
    int64_t r[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const int64_t i = base + u * 256 + threadIdx.x;
        ok[u] = i < i1;
        const int64_t ic = ok[u] ? i : i1 - 1;
        r[u] = SEL ? (int64_t)sel[ic] : ic;
    }
    unsigned long long a[PU], b[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        a[u] = load_kw<KW>(k0, r[u]);
        b[u] = NK == 2 ? load_kw<KW>(k1, r[u]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < PU; u++) {
        uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ a[u]);   // load_keys' hash
        if (NK == 2) hh = mix64(hh ^ b[u]);
        h[u] = hh;
    }
}

template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(256) void part_count_fast_kernel(const void *k0, const void *k1, const int32_t *sel, int64_t n,
                                                              uint64_t mask, int nparts, int64_t rows_per_wg,
                                                              int32_t *__restrict__ counts) {
    extern __shared__ int hist[];
    for (int e = threadIdx.x; e < nparts; e += 256) hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < n ? i0 + rows_per_wg : n;
    for (int64_t base = i0; base < i1; base += 256 * PU) {
        uint64_t h[PU];
        bool ok[PU];
        part_hashes<KW, NK, SEL>(k0, k1, sel, base, i1, h, ok);
#pragma unroll
        for (int u = 0; u < PU; u++)
            if (ok[u]) atomicAdd(&hist[(h[u] & mask) >> PB_SLICE_LOG], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nparts; e += 256) counts[(int64_t)e * gridDim.x + blockIdx.x] = hist[e];
}

template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(256) void part_scatter_fast_kernel(const void *k0, const void *k1, const int32_t *sel, int64_t n,
                                                                uint64_t mask, int nparts, int64_t rows_per_wg,
                                                                const int32_t *__restrict__ offsets,
                                                                ulonglong2 *__restrict__ part_rec) {
    extern __shared__ int cursor[];
    for (int e = threadIdx.x; e < nparts; e += 256) cursor[e] = offsets[(int64_t)e * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < n ? i0 + rows_per_wg : n;
    for (int64_t base = i0; base < i1; base += 256 * PU) {
        uint64_t h[PU];
        bool ok[PU];
        part_hashes<KW, NK, SEL>(k0, k1, sel, base, i1, h, ok);
#pragma unroll
        for (int u = 0; u < PU; u++) {
            if (!ok[u]) continue;
            const int pos = atomicAdd(&cursor[(h[u] & mask) >> PB_SLICE_LOG], 1);
            part_rec[pos] = make_ulonglong2(h[u], (unsigned long long)(base + u * 256 + threadIdx.x));
        }
    }
}

__global__ __launch_bounds__(256) void part_count_kernel(JoinSide B, uint64_t mask, int nparts, int64_t rows_per_wg,
                                                         int32_t *__restrict__ counts) {
    extern __shared__ int hist[];
    for (int e = threadIdx.x; e < nparts; e += 256) hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < B.n ? i0 + rows_per_wg : B.n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        int64_t r = B.sel ? B.sel[i] : i;
        unsigned long long k[JOIN_MAX_KEYS];
        uint64_t h;
        if (load_keys(B, r, k, &h)) atomicAdd(&hist[(h & mask) >> PB_SLICE_LOG], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nparts; e += 256) counts[(int64_t)e * gridDim.x + blockIdx.x] = hist[e];
}

__global__ __launch_bounds__(256) void part_scatter_kernel(JoinSide B, uint64_t mask, int nparts, int64_t rows_per_wg,
                                                           const int32_t *__restrict__ offsets, ulonglong2 *__restrict__ part_rec,
                                                           int32_t *__restrict__ next) {
    extern __shared__ int cursor[];
    for (int e = threadIdx.x; e < nparts; e += 256) cursor[e] = offsets[(int64_t)e * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < B.n ? i0 + rows_per_wg : B.n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        int64_t r = B.sel ? B.sel[i] : i;
        unsigned long long k[JOIN_MAX_KEYS];
        uint64_t h;
        if (!load_keys(B, r, k, &h)) { next[i] = -2; continue; }  // NULL key: not inserted
        const int pos = atomicAdd(&cursor[(h & mask) >> PB_SLICE_LOG], 1);
        part_rec[pos] = make_ulonglong2(h, (unsigned long long)i);   // one 16-byte store per row
    }
}

__global__ __launch_bounds__(1024) void part_build_kernel(const int32_t *__restrict__ offsets, int nwg, int nparts,
                                                          const int64_t *__restrict__ total,
                                                          const ulonglong2 *__restrict__ part_rec,
                                                          int32_t *__restrict__ head, int32_t *__restrict__ next, Bloom bl,
                                                          int bloom_words) {
    extern __shared__ int lds[];
    int *lhead = lds;
    unsigned *lbloom = reinterpret_cast<unsigned *>(lds + PB_SLICE);
    const int p = blockIdx.x;
    for (int e = threadIdx.x; e < PB_SLICE; e += 1024) lhead[e] = -1;
    for (int e = threadIdx.x; e < bloom_words; e += 1024) lbloom[e] = 0;
    __syncthreads();
    const int64_t start = offsets[(int64_t)p * nwg];
    const int64_t end = p + 1 < nparts ? (int64_t)offsets[(int64_t)(p + 1) * nwg] : *total;
    for (int64_t t = start + threadIdx.x; t < end; t += 1024) {
        const ulonglong2 rec = part_rec[t];
        const int32_t i = (int32_t)rec.y;
        const unsigned long long h = rec.x;
        next[i] = atomicExch(&lhead[h & (PB_SLICE - 1)], i);   // head insertion, as the atomic build does
        if (bl.bits) atomicOr(&lbloom[(h >> 34) & bl.inner_mask], bloom_mask(h >> 24));
        if (bl.coarse) atomicOr(&bl.coarse[coarse_bit(h) >> 5], 1u << (coarse_bit(h) & 31));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PB_SLICE; e += 1024) head[(int64_t)p * PB_SLICE + e] = lhead[e];
    if (bl.bits)
        for (int e = threadIdx.x; e < bloom_words; e += 1024) bl.bits[((uint64_t)p << bl.hi_shift) + e] = lbloom[e];
}

constexpr int JP_ROUNDS = 8;
constexpr int JP_CHUNK = 256 * JP_ROUNDS;

// ---- large build sides (no bitmap; > 4 M keys): the NODE TABLE.
// The atomic build above issues one scattered device atomic per row (~20 G/s on this part: 15 M keys
// 0.65 ms), and a chain step of a probe costs three dependent random reads (head, next, build key).
// Here the build is partitioned by head slice like part_build_kernel, but the partition records ARE
// the table: node p = {packed key, build row id, next node} (16 bytes, partition order), written by
// the scatter pass, linked by ONE workgroup per 128 KiB head slice in LDS (ds atomics) which fills
// the `next` fields of its own contiguous records and writes the head slice whole — no global
// atomics, no scattered 4-byte stores (the per-row next[] of the small-table path is addressed by
// original row position: 15 M scattered stores made part_build_kernel take 0.35 ms here). A probe's
// chain step is head -> node: two random reads instead of three, the key compare needs no third.
// Keys: one column (any integer width) or two 4-byte columns, packed into 64 bits.
struct BigNode {
    unsigned long long key;
    int32_t row;    // build row id (sel applied)
    int32_t next;   // next node of the bucket, -1 = end
};

constexpr int BG_SLICE_LOG = 15;               // head entries per partition: 32768 = 128 KiB of LDS
constexpr int BG_SLICE = 1 << BG_SLICE_LOG;
constexpr int BG_MAX_PARTS = 8192;

template <int KW, int NK> __device__ __forceinline__ unsigned long long big_pack(unsigned long long a, unsigned long long b) {
    return NK == 2 ? ((a << 32) | (b & 0xffffffffull)) : a;
}
__device__ __forceinline__ uint64_t big_hash(unsigned long long key) { return mix64(0x9e3779b97f4a7c15ULL ^ key); }

constexpr int BG_T = 1024;   // threads of a partition-pass workgroup: one workgroup per CU, 16 waves in flight

// rows of one workgroup, PU at a time: packed key + validity (NULL keys never enter the table)
template <int KW, int NK, bool SEL>
__device__ __forceinline__ void big_keys(const void *k0, const void *k1, const uint8_t *v0, const uint8_t *v1, const int32_t *sel,
                                         int64_t base, int64_t i1, unsigned long long (&key)[PU], int32_t (&row)[PU], bool (&ok)[PU]) {
    int64_t r[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const int64_t i = base + u * BG_T + threadIdx.x;
        ok[u] = i < i1;
        const int64_t ic = ok[u] ? i : i1 - 1;
        r[u] = SEL ? (int64_t)sel[ic] : ic;
        row[u] = (int32_t)r[u];
    }
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const unsigned long long a = load_kw<KW>(k0, r[u]);
        const unsigned long long b = NK == 2 ? load_kw<KW>(k1, r[u]) : 0ull;
        key[u] = big_pack<KW, NK>(a, b);
        if (v0) ok[u] = ok[u] && bit_valid(v0, r[u]);
        if (NK == 2 && v1) ok[u] = ok[u] && bit_valid(v1, r[u]);
    }
}

template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(BG_T) void big_count_kernel(const void *k0, const void *k1, const uint8_t *v0, const uint8_t *v1,
                                                        const int32_t *sel, int64_t n, uint64_t mask, int nparts, int64_t rows_per_wg,
                                                        int32_t *__restrict__ counts) {
    extern __shared__ int hist[];
    for (int e = threadIdx.x; e < nparts; e += BG_T) hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < n ? i0 + rows_per_wg : n;
    for (int64_t base = i0; base < i1; base += BG_T * PU) {
        unsigned long long key[PU];
        int32_t row[PU];
        bool ok[PU];
        big_keys<KW, NK, SEL>(k0, k1, v0, v1, sel, base, i1, key, row, ok);
#pragma unroll
        for (int u = 0; u < PU; u++)
            if (ok[u]) atomicAdd(&hist[(big_hash(key[u]) & mask) >> BG_SLICE_LOG], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nparts; e += BG_T) counts[(int64_t)e * gridDim.x + blockIdx.x] = hist[e];
}

template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(BG_T) void big_scatter_kernel(const void *k0, const void *k1, const uint8_t *v0, const uint8_t *v1,
                                                          const int32_t *sel, int64_t n, uint64_t mask, int nparts, int64_t rows_per_wg,
                                                          const int32_t *__restrict__ offsets, BigNode *__restrict__ nodes) {
    extern __shared__ int cursor[];
    for (int e = threadIdx.x; e < nparts; e += BG_T) cursor[e] = offsets[(int64_t)e * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg, i1 = i0 + rows_per_wg < n ? i0 + rows_per_wg : n;
    for (int64_t base = i0; base < i1; base += BG_T * PU) {
        unsigned long long key[PU];
        int32_t row[PU];
        bool ok[PU];
        big_keys<KW, NK, SEL>(k0, k1, v0, v1, sel, base, i1, key, row, ok);
#pragma unroll
        for (int u = 0; u < PU; u++) {
            if (!ok[u]) continue;
            const int pos = atomicAdd(&cursor[(big_hash(key[u]) & mask) >> BG_SLICE_LOG], 1);
            BigNode nd;
            nd.key = key[u]; nd.row = row[u]; nd.next = -1;
            nodes[pos] = nd;   // one 16-byte store; a workgroup's records of a partition are contiguous
        }
    }
}

// one workgroup per head slice: links the partition's (contiguous) nodes in LDS, writes head + next
__global__ __launch_bounds__(1024) void big_build_kernel(const int32_t *__restrict__ offsets, int nwg, int nparts,
                                                         const int64_t *__restrict__ total, BigNode *__restrict__ nodes,
                                                         int32_t *__restrict__ head) {
    extern __shared__ int lhead[];
    for (int e = threadIdx.x; e < BG_SLICE; e += 1024) lhead[e] = -1;
    __syncthreads();
    const int p = blockIdx.x;
    const int64_t b0 = offsets[(int64_t)p * nwg];
    const int64_t b1 = p + 1 < nparts ? (int64_t)offsets[(int64_t)(p + 1) * nwg] : *total;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 1024) {
        const unsigned long long key = nodes[i].key;
        nodes[i].next = atomicExch(&lhead[big_hash(key) & (BG_SLICE - 1)], (int)i);   // head insertion, as everywhere
    }
    __syncthreads();
    for (int e = threadIdx.x; e < BG_SLICE; e += 1024) head[(int64_t)p * BG_SLICE + e] = lhead[e];
}

// probe side of the node table. BU probes per lane issue their key reads, then their head reads,
// then walk their chains together (every chain step is ONE 16-byte node read).
constexpr int BU = 4;

// MODE 0: lookup (out[i] = last matching build row or -1; stats: misses, multi-matches)
// MODE 1: count pass of the inner probe (ccnt[i] = matches, saturating at 65535; cmatch[i] = a
//         matching build row; block_counts[blk] = matches of the 2048-row block)
// MODE 2: mark (found[i] = 0 / 1)
template <int KW, int NK, bool SELP, int MODE>
__global__ __launch_bounds__(256) void big_probe_kernel(const void *__restrict__ pk0, const void *__restrict__ pk1,
                                                        const uint8_t *__restrict__ pv0, const uint8_t *__restrict__ pv1,
                                                        const int32_t *__restrict__ psel, int64_t n,
                                                        const int32_t *__restrict__ head, uint64_t mask,
                                                        const BigNode *__restrict__ nodes, int32_t *__restrict__ out,
                                                        uint16_t *__restrict__ ccnt, int32_t *__restrict__ block_counts,
                                                        uint8_t *__restrict__ found, int *__restrict__ stats) {
    static_assert(JP_CHUNK == 256 * 2 * BU, "a workgroup iteration covers half a block");
    int misses = 0, multi = 0;
    __shared__ int s_tot[4];
    // MODE 1 runs one workgroup per 2048-row block (its total is a plain store); the others grid-stride
    const int64_t step = MODE == 1 ? (int64_t)gridDim.x * JP_CHUNK : (int64_t)gridDim.x * 256 * BU;
    for (int64_t base0 = (int64_t)blockIdx.x * (MODE == 1 ? JP_CHUNK : 256 * BU); base0 < n; base0 += step) {
        int blocktotal = 0;
        for (int half = 0; half < (MODE == 1 ? 2 : 1); half++) {
            const int64_t base = base0 + half * 256 * BU;
            int64_t i[BU], r[BU];
            bool ok[BU];
            unsigned long long k[BU];
            int b[BU], c[BU];
            int32_t hit[BU];
#pragma unroll
            for (int u = 0; u < BU; u++) {
                i[u] = base + u * 256 + threadIdx.x;
                ok[u] = i[u] < n;
                r[u] = ok[u] ? i[u] : 0;
                c[u] = 0;
                hit[u] = -1;
            }
            if (SELP) {
#pragma unroll
                for (int u = 0; u < BU; u++) r[u] = psel[r[u]];
            }
#pragma unroll
            for (int u = 0; u < BU; u++) {
                const unsigned long long a = load_kw<KW>(pk0, r[u]);
                const unsigned long long bb = NK == 2 ? load_kw<KW>(pk1, r[u]) : 0ull;
                k[u] = big_pack<KW, NK>(a, bb);
                if (pv0) ok[u] = ok[u] && bit_valid(pv0, r[u]);      // NULL keys never match
                if (NK == 2 && pv1) ok[u] = ok[u] && bit_valid(pv1, r[u]);
            }
#pragma unroll
            for (int u = 0; u < BU; u++) {
                const int hb = head[big_hash(k[u]) & mask];
                b[u] = ok[u] ? hb : -1;
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < BU; u++) more = more || b[u] >= 0;
            while (more) {
                BigNode nd[BU];
#pragma unroll
                for (int u = 0; u < BU; u++) nd[u] = nodes[b[u] >= 0 ? b[u] : 0];
                more = false;
#pragma unroll
                for (int u = 0; u < BU; u++) {
                    if (b[u] >= 0) {
                        if (nd[u].key == k[u]) { c[u]++; hit[u] = nd[u].row; }
                        b[u] = nd[u].next;
                    }
                    more = more || b[u] >= 0;
                }
            }
#pragma unroll
            for (int u = 0; u < BU; u++) {
                if (i[u] >= n) continue;
                if (MODE == 0) { out[i[u]] = hit[u]; misses += c[u] == 0; multi += c[u] > 1; }
                else if (MODE == 1) { ccnt[i[u]] = (uint16_t)(c[u] > 65535 ? 65535 : c[u]); out[i[u]] = hit[u]; blocktotal += c[u]; }
                else found[i[u]] = c[u] > 0 ? 1 : 0;
            }
        }
        if (MODE == 1) {
            for (int o = 32; o > 0; o >>= 1) blocktotal += __shfl_xor(blocktotal, o);
            if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = blocktotal;
            __syncthreads();
            if (threadIdx.x == 0) block_counts[base0 / JP_CHUNK] = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
            __syncthreads();
        }
    }
    if (MODE == 0) {
        for (int o = 32; o > 0; o >>= 1) { misses += __shfl_xor(misses, o); multi += __shfl_xor(multi, o); }
        if ((threadIdx.x & 63) == 0) {
            if (misses) atomicAdd(stats, misses);
            if (multi) atomicAdd(stats + 1, multi);
        }
    }
}

// emit pass of the inner probe: one wave per 2048-row block, pairs ordered by probe position then
// chain order; single matches come from the count pass, multi-matches walk their chain again
template <int KW, int NK, bool SELP>
__global__ __launch_bounds__(256) void big_emit_kernel(const void *__restrict__ pk0, const void *__restrict__ pk1,
                                                       const int32_t *__restrict__ psel, int64_t n,
                                                       const int32_t *__restrict__ head, uint64_t mask,
                                                       const BigNode *__restrict__ nodes, const uint16_t *__restrict__ ccnt,
                                                       const int32_t *__restrict__ cmatch, const int32_t *__restrict__ block_off,
                                                       int64_t nb, int64_t cap, int32_t *__restrict__ out_probe,
                                                       int32_t *__restrict__ out_build) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        int64_t running = block_off[blk];
        const int64_t i0 = blk * JP_CHUNK, i1 = i0 + JP_CHUNK < n ? i0 + JP_CHUNK : n;
        for (int64_t t0 = i0; t0 < i1; t0 += 64) {
            const int64_t i = t0 + lane;
            int c = i < i1 ? (int)ccnt[i] : 0;
            const int64_t r = i < i1 ? (SELP ? (int64_t)psel[i] : i) : 0;
            unsigned long long key = 0;
            if (c > 1) {   // rare: walk the chain again (also recounts a saturated counter)
                const unsigned long long a = load_kw<KW>(pk0, r);
                const unsigned long long bb = NK == 2 ? load_kw<KW>(pk1, r) : 0ull;
                key = big_pack<KW, NK>(a, bb);
                if (c == 65535) {
                    c = 0;
                    for (int b = head[big_hash(key) & mask]; b >= 0; b = nodes[b].next) c += nodes[b].key == key;
                }
            }
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (c == 1) {
                const int64_t pos = running + incl - 1;
                if (pos < cap) { out_probe[pos] = (int32_t)r; out_build[pos] = cmatch[i]; }
            } else if (c > 1) {
                int64_t pos = running + incl - c;
                for (int b = head[big_hash(key) & mask]; b >= 0; b = nodes[b].next)
                    if (nodes[b].key == key) {
                        if (pos < cap) { out_probe[pos] = (int32_t)r; out_build[pos] = nodes[b].row; }
                        pos++;
                    }
            }
            running += __shfl(incl, 63);
        }
    }
}

__device__ __forceinline__ bool keys_equal(const JoinSide &B, int64_t brow, const unsigned long long *k) {
    for (int c = 0; c < B.nkeys; c++)
        if (jkey(B.key[c], brow) != k[c]) return false;
    return true;
}

// number of build rows matching probe position i
__device__ __forceinline__ int probe_count(const JoinSide &B, const JoinSide &Pr, const int32_t *head, uint64_t mask,
                                           const int32_t *next, int64_t i, const Bloom &bl) {
    int64_t r = Pr.sel ? Pr.sel[i] : i;
    unsigned long long k[JOIN_MAX_KEYS];
    uint64_t h;
    if (!load_keys(Pr, r, k, &h)) return 0;
    if (!bloom_maybe(bl, h)) return 0;
    int c = 0;
    for (int b = head[h & mask]; b >= 0; b = next[b]) {
        int64_t brow = B.sel ? B.sel[b] : b;
        c += keys_equal(B, brow, k) ? 1 : 0;
    }
    return c;
}

__device__ __forceinline__ bool range_pred(const RangePred &W, int64_t r) {
    if (W.kind == 0) return true;
    if (W.kind < 0 || !bit_valid(W.validity, r)) return false;  // NULL never selects
    long long v = W.kind == 1 ? (long long)((const int32_t *)W.data)[r]
                  : W.kind == 2 ? ((const int64_t *)W.data)[r] : (long long)((const uint8_t *)W.data)[r];
    return v >= W.lo && v <= W.hi;
}


// ---- selective probes (a Bloom bitmap exists). In the two-pass kernels above a lane that has to
// walk a chain (three or four dependent random reads) holds up its wave while the other lanes
// idle, and both passes pay for it. Here the work is split by density, with no same-address
// atomics anywhere (one returning atomic per workgroup on a shared counter costs ~7 ns each,
// serialised — 0.1 ms for a 32M-row probe):
//   join_cand_kernel   streams the probe keys, tests the bitmap and writes the surviving
//                      positions of its 2048-row block, in order, to the block's own slice of
//                      the candidate array
//   join_chain_kernel  one wave per block slice: walks the candidates' chains, stores each
//                      candidate's match count and the block total
//   (scan of the block totals)
//   join_emit_kernel   one wave per block slice: prefix-sums the counts and writes the pairs
// Output order is probe order, as before.
__global__ __launch_bounds__(256) void join_cand_kernel(JoinSide Pr, Bloom bl, RangePred where, uint16_t *__restrict__ cand,
                                                        int32_t *__restrict__ ccount) {
    // round rr covers positions base + rr*256 + thread (lane-consecutive rows: coalesced reads);
    // ordered output = round-major, so the slot of a survivor is the number of survivors in
    // earlier rounds and earlier waves plus those in lower lanes of its own ballot
    const int64_t base = (int64_t)blockIdx.x * JP_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int wc[JP_ROUNDS][4];
    unsigned long long bal[JP_ROUNDS];
    // Stage by stage over all 8 rounds, every load unconditional (rows that dropped out read row
    // 0): the 8 loads of a stage are independent and go out back to back, so a wave pays three
    // or four memory latencies in total instead of three or four per round — the kernel was
    // latency bound (PMC: 77 % of wave cycles waiting), not byte bound.
    int64_t r[JP_ROUNDS];
    bool ok[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        const int64_t i = base + rr * 256 + threadIdx.x;
        ok[rr] = i < Pr.n;
        const int64_t ic = ok[rr] ? i : 0;
        r[rr] = Pr.sel ? (int64_t)Pr.sel[ic] : ic;
    }
    if (where.kind != 0) {
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) ok[rr] = range_pred(where, r[rr]) && ok[rr];
    }
    uint64_t h[JP_ROUNDS] = {};
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        unsigned long long k[JOIN_MAX_KEYS];
        ok[rr] = load_keys(Pr, ok[rr] ? r[rr] : 0, k, &h[rr]) && ok[rr];
    }
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        const bool take = bloom_maybe(bl, h[rr]) && ok[rr];
        bal[rr] = __ballot(take);
        if (lane == 0) wc[rr][wv] = __popcll(bal[rr]);
    }
    __syncthreads();
    uint16_t *dst = cand + base;
    int before = 0;
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        int off = before;
        for (int k = 0; k < wv; k++) off += wc[rr][k];
        if ((bal[rr] >> lane) & 1) dst[off + __popcll(bal[rr] & ((1ull << lane) - 1ull))] = (uint16_t)(rr * 256 + threadIdx.x);
        before += wc[rr][0] + wc[rr][1] + wc[rr][2] + wc[rr][3];
    }
    if (threadIdx.x == 0) ccount[blockIdx.x] = before;
}

// The common probe shape — one key column without NULLs, optional integer-range filter without
// NULLs — compiled without any per-row type dispatch: with the switches of jkey()/range_pred()
// in the way every load sat in its own basic block behind an s_waitcnt vmcnt(0), so the "stages"
// of join_cand_kernel still paid one memory latency per load. Here the 8 loads of a stage are
// straight-line code and are issued back to back.
template <int KW, int WK, bool SEL, int NK>
__global__ __launch_bounds__(256) void join_cand_fast_kernel(const void *__restrict__ keycol, const void *__restrict__ keycol2,
                                                             const int32_t *__restrict__ sel,
                                                             int64_t n, Bloom bl, const void *__restrict__ wdata, long long wlo,
                                                             long long whi, uint16_t *__restrict__ cand,
                                                             int32_t *__restrict__ ccount) {
    const int64_t base = (int64_t)blockIdx.x * JP_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int wc[JP_ROUNDS][4];
    unsigned long long bal[JP_ROUNDS];
    int64_t r[JP_ROUNDS];
    bool ok[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        const int64_t i = base + rr * 256 + threadIdx.x;
        ok[rr] = i < n;
        const int64_t ic = ok[rr] ? i : 0;
        r[rr] = SEL ? (int64_t)sel[ic] : ic;
    }
    if (WK != 0) {
        long long w[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++)
            w[rr] = SEL ? (WK == 1 ? (long long)((const int32_t *)wdata)[r[rr]]
                           : WK == 2 ? ((const int64_t *)wdata)[r[rr]] : (long long)((const uint8_t *)wdata)[r[rr]])
                        : (WK == 1 ? (long long)__builtin_nontemporal_load((const int32_t *)wdata + r[rr])
                           : WK == 2 ? (long long)__builtin_nontemporal_load((const int64_t *)wdata + r[rr])
                                     : (long long)__builtin_nontemporal_load((const uint8_t *)wdata + r[rr]));
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            ok[rr] = ok[rr] && w[rr] >= wlo && w[rr] <= whi;
            if (!ok[rr]) r[rr] = 0;
        }
    }
    unsigned long long k[JP_ROUNDS], k2[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        k[rr] = SEL ? load_kw<KW>(keycol, r[rr]) : load_kw_nt<KW>(keycol, r[rr]);
        k2[rr] = NK == 2 ? (SEL ? load_kw<KW>(keycol2, r[rr]) : load_kw_nt<KW>(keycol2, r[rr])) : 0ull;
    }
    unsigned word[JP_ROUNDS], msk[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ k[rr]);   // load_keys' hash
        if (NK == 2) hh = mix64(hh ^ k2[rr]);
        msk[rr] = bloom_mask(hh >> 24);
        word[rr] = bl.bits[bloom_word(bl, hh)];
    }
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        bal[rr] = __ballot(ok[rr] && (word[rr] & msk[rr]) == msk[rr]);
        if (lane == 0) wc[rr][wv] = __popcll(bal[rr]);
    }
    __syncthreads();
    uint16_t *dst = cand + base;
    int before = 0;
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        int off = before;
        for (int q = 0; q < wv; q++) off += wc[rr][q];
        if ((bal[rr] >> lane) & 1) dst[off + __popcll(bal[rr] & ((1ull << lane) - 1ull))] = (uint16_t)(rr * 256 + threadIdx.x);
        before += wc[rr][0] + wc[rr][1] + wc[rr][2] + wc[rr][3];
    }
    if (threadIdx.x == 0) ccount[blockIdx.x] = before;
}

// Vectorised form of join_cand_fast_kernel for the identity selection over 16-byte aligned columns
// (see dc_block_keys): the round-per-row form spends ~85 VALU instructions per row beside the hash
// and was VALU bound (Q3: 60 M rows in 165 us; its columns stream in ~115).
template <int KW, int WK, int NK>
__global__ __launch_bounds__(256) void join_cand_vec_kernel(const void *__restrict__ keycol, const void *__restrict__ keycol2, int64_t n,
                                                            Bloom bl, const void *__restrict__ wdata, long long wlo, long long whi,
                                                            uint16_t *__restrict__ cand, int32_t *__restrict__ ccount) {
    constexpr int R = 16 / KW, G = 8 / R;
    const int64_t base = (int64_t)blockIdx.x * JP_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int wcv[4];
    long long k[8], k2[8];
    bool ok[8];
    dc_block_keys<KW, WK, NK>(keycol, keycol2, wdata, wlo, whi, base + wv * 512, lane, base + JP_CHUNK <= n, true, n, k, k2, ok);
    unsigned word[8], msk[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ (unsigned long long)k[s]);   // load_keys' hash
        if (NK == 2) hh = mix64(hh ^ (unsigned long long)k2[s]);
        msk[s] = bloom_mask(hh >> 24);
        word[s] = bl.bits[ok[s] ? bloom_word(bl, hh) : 0];   // filtered rows read word 0: one request per wave
    }
    bool take[8];
#pragma unroll
    for (int s = 0; s < 8; s++) take[s] = ok[s] && (word[s] & msk[s]) == msk[s];
    int first[G];
    const int wave_cands = dc_wave_ranks<R>(take, first);
    if (lane == 0) wcv[wv] = wave_cands;
    __syncthreads();
    int off = 0;
    for (int q = 0; q < wv; q++) off += wcv[q];
#pragma unroll
    for (int g = 0; g < G; g++) {
        uint16_t *dst = cand + base + off + first[g];
#pragma unroll
        for (int e = 0; e < R; e++)
            if (take[g * R + e]) *dst++ = (uint16_t)(wv * 512 + (g * 64 + lane) * R + e);
    }
    if (threadIdx.x == 0) ccount[blockIdx.x] = wcv[0] + wcv[1] + wcv[2] + wcv[3];
}

// The same kernel for tiny build sides: 1024 threads = four 256-thread groups, each owning one
// 2048-row block per step; the workgroup first copies the coarse bitmap into LDS (one workgroup per
// CU, so 128 KiB x 256 of traffic in all) and a probe reads the L2 bitmap only when its coarse
// bit is set (the others read word 0: one request per wave instead of one per lane).
template <int KW, int WK, bool SEL, int NK>
__global__ __launch_bounds__(1024) void join_cand_coarse_kernel(const void *__restrict__ keycol, const void *__restrict__ keycol2,
                                                                const int32_t *__restrict__ sel, int64_t n, Bloom bl,
                                                                const void *__restrict__ wdata, long long wlo, long long whi,
                                                                uint16_t *__restrict__ cand, int32_t *__restrict__ ccount,
                                                                int64_t nb) {
    extern __shared__ unsigned co_lds[];          // CO_WORDS words, then the per-group wave counts
    int (*wc)[JP_ROUNDS][4] = reinterpret_cast<int (*)[JP_ROUNDS][4]>(co_lds + CO_WORDS);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(bl.coarse);
        uint4 *dst = reinterpret_cast<uint4 *>(co_lds);
        for (int e = threadIdx.x; e < CO_WORDS / 4; e += 1024) dst[e] = src[e];
    }
    __syncthreads();
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int lane = tid & 63, wv = tid >> 6;
    for (int64_t it = 0; (it * gridDim.x + blockIdx.x) * 4 < nb; it++) {
        const int64_t blk = (it * gridDim.x + blockIdx.x) * 4 + grp;
        const bool have = blk < nb;
        const int64_t base = blk * JP_CHUNK;
        unsigned long long bal[JP_ROUNDS];
        int64_t r[JP_ROUNDS];
        bool ok[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            const int64_t i = base + rr * 256 + tid;
            ok[rr] = have && i < n;
            const int64_t ic = ok[rr] ? i : 0;
            r[rr] = SEL ? (int64_t)sel[ic] : ic;
        }
        if (WK != 0) {
            long long w[JP_ROUNDS];
#pragma unroll
            for (int rr = 0; rr < JP_ROUNDS; rr++)
                w[rr] = WK == 1 ? (long long)((const int32_t *)wdata)[r[rr]]
                        : WK == 2 ? ((const int64_t *)wdata)[r[rr]] : (long long)((const uint8_t *)wdata)[r[rr]];
#pragma unroll
            for (int rr = 0; rr < JP_ROUNDS; rr++) {
                ok[rr] = ok[rr] && w[rr] >= wlo && w[rr] <= whi;
                if (!ok[rr]) r[rr] = 0;
            }
        }
        unsigned long long k[JP_ROUNDS], k2[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            k[rr] = load_kw<KW>(keycol, r[rr]);
            k2[rr] = NK == 2 ? load_kw<KW>(keycol2, r[rr]) : 0ull;
        }
        unsigned word[JP_ROUNDS], msk[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ k[rr]);   // load_keys' hash
            if (NK == 2) hh = mix64(hh ^ k2[rr]);
            msk[rr] = bloom_mask(hh >> 24);
            const unsigned cb = coarse_bit(hh);
            ok[rr] = ok[rr] && ((co_lds[cb >> 5] >> (cb & 31)) & 1u);
            word[rr] = bl.bits[ok[rr] ? bloom_word(bl, hh) : 0];
        }
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            bal[rr] = __ballot(ok[rr] && (word[rr] & msk[rr]) == msk[rr]);
            if (lane == 0) wc[grp][rr][wv] = __popcll(bal[rr]);
        }
        __syncthreads();
        if (have) {
            uint16_t *dst = cand + base;
            int before = 0;
#pragma unroll
            for (int rr = 0; rr < JP_ROUNDS; rr++) {
                int off = before;
                for (int q = 0; q < wv; q++) off += wc[grp][rr][q];
                if ((bal[rr] >> lane) & 1) dst[off + __popcll(bal[rr] & ((1ull << lane) - 1ull))] = (uint16_t)(rr * 256 + tid);
                before += wc[grp][rr][0] + wc[grp][rr][1] + wc[grp][rr][2] + wc[grp][rr][3];
            }
            if (tid == 0) ccount[blk] = before;
        }
        __syncthreads();   // the counts are rewritten in the next step
    }
}

template <int KW, int WK>
static void launch_cand_fast(bool has_sel, int nb, hipStream_t st, const JoinSide &P, const Bloom &bl, const RangePred &w,
                             uint16_t *cand, int32_t *ccount) {
#define PH_CAND_ARGS P.key[0].data, P.key[1].data, P.sel, P.n, bl, w.data, w.lo, w.hi, cand, ccount
    auto aligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    static const bool no_vec = getenv("PH_JOIN_CAND_VEC") && atoi(getenv("PH_JOIN_CAND_VEC")) == 0;
    if (!no_vec && !bl.coarse && !has_sel && KW != 1 && aligned(P.key[0].data) && (P.nkeys == 1 || aligned(P.key[1].data)) && (WK == 0 || aligned(w.data))) {
        if (P.nkeys == 2) join_cand_vec_kernel<KW == 1 ? 4 : KW, WK, 2><<<nb, 256, 0, st>>>(P.key[0].data, P.key[1].data, P.n, bl, w.data, w.lo, w.hi, cand, ccount);
        else join_cand_vec_kernel<KW == 1 ? 4 : KW, WK, 1><<<nb, 256, 0, st>>>(P.key[0].data, P.key[1].data, P.n, bl, w.data, w.lo, w.hi, cand, ccount);
        return;
    }
    if (bl.coarse) {   // tiny build side: coarse bitmap in LDS, one 1024-thread workgroup per CU
        const size_t lds = (size_t)CO_WORDS * 4 + 4 * JP_ROUNDS * 4 * sizeof(int);
        const int grid = std::min((nb + 3) / 4, g_cu_count);
#define PH_COARSE(SELV, NKV)                                                                                            \
    do {                                                                                                                \
        (void)hipFuncSetAttribute((const void *)join_cand_coarse_kernel<KW, WK, SELV, NKV>,                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
        join_cand_coarse_kernel<KW, WK, SELV, NKV><<<grid, 1024, lds, st>>>(PH_CAND_ARGS, (int64_t)nb);                 \
    } while (0)
        if (P.nkeys == 2) { if (has_sel) PH_COARSE(true, 2); else PH_COARSE(false, 2); }
        else { if (has_sel) PH_COARSE(true, 1); else PH_COARSE(false, 1); }
#undef PH_COARSE
    } else if (P.nkeys == 2) {
        if (has_sel) join_cand_fast_kernel<KW, WK, true, 2><<<nb, 256, 0, st>>>(PH_CAND_ARGS);
        else join_cand_fast_kernel<KW, WK, false, 2><<<nb, 256, 0, st>>>(PH_CAND_ARGS);
    } else {
        if (has_sel) join_cand_fast_kernel<KW, WK, true, 1><<<nb, 256, 0, st>>>(PH_CAND_ARGS);
        else join_cand_fast_kernel<KW, WK, false, 1><<<nb, 256, 0, st>>>(PH_CAND_ARGS);
    }
#undef PH_CAND_ARGS
}

template <int KW>
static void launch_cand_fast_k(int wk, bool has_sel, int nb, hipStream_t st, const JoinSide &P, const Bloom &bl,
                               const RangePred &w, uint16_t *cand, int32_t *ccount) {
    switch (wk) {
    case 0: launch_cand_fast<KW, 0>(has_sel, nb, st, P, bl, w, cand, ccount); break;
    case 1: launch_cand_fast<KW, 1>(has_sel, nb, st, P, bl, w, cand, ccount); break;
    case 2: launch_cand_fast<KW, 2>(has_sel, nb, st, P, bl, w, cand, ccount); break;
    default: launch_cand_fast<KW, 3>(has_sel, nb, st, P, bl, w, cand, ccount); break;
    }
}

// true when the fast kernel took the launch
static bool try_cand_fast(int nb, hipStream_t st, const JoinSide &P, const Bloom &bl, const RangePred &w, uint16_t *cand,
                          int32_t *ccount) {
    if (P.nkeys > 2 || P.key[0].validity || w.kind < 0 || (w.kind != 0 && w.validity)) return false;
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
    const int kw = width(P.key[0].type);
    if (P.nkeys == 2 && (P.key[1].validity || width(P.key[1].type) != kw)) return false;
    if (kw == 4) launch_cand_fast_k<4>(w.kind, P.sel != nullptr, nb, st, P, bl, w, cand, ccount);
    else if (kw == 1) launch_cand_fast_k<1>(w.kind, P.sel != nullptr, nb, st, P, bl, w, cand, ccount);
    else launch_cand_fast_k<8>(w.kind, P.sel != nullptr, nb, st, P, bl, w, cand, ccount);
    return true;
}

__global__ __launch_bounds__(256) void join_chain_kernel(JoinSide B, JoinSide Pr, const int32_t *__restrict__ head,
                                                         uint64_t mask, const int32_t *__restrict__ next,
                                                         const uint16_t *__restrict__ cand, const int32_t *__restrict__ ccount,
                                                         uint16_t *__restrict__ ccnt, int32_t *__restrict__ cmatch,
                                                         int32_t *__restrict__ block_counts, int64_t nb) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        const int cnt = ccount[blk];
        int total = 0;
        for (int t0 = 0; t0 < cnt; t0 += 64) {
            const int t = t0 + lane;
            int c = 0;
            if (t < cnt) {
                // probe_count, also remembering the matching build row: with a unique build key
                // (the usual N:1 join) the emit pass then writes the pair without a second walk
                const int64_t i = blk * JP_CHUNK + cand[blk * JP_CHUNK + t];
                const int64_t r = Pr.sel ? Pr.sel[i] : i;
                unsigned long long k[JOIN_MAX_KEYS];
                uint64_t h;
                int32_t hit = -1;
                if (load_keys(Pr, r, k, &h)) {
                    for (int b = head[h & mask]; b >= 0; b = next[b]) {
                        int64_t brow = B.sel ? B.sel[b] : b;
                        if (keys_equal(B, brow, k)) { c++; hit = (int32_t)brow; }
                    }
                }
                ccnt[blk * JP_CHUNK + t] = (uint16_t)(c > 65535 ? 65535 : c);
                cmatch[blk * JP_CHUNK + t] = hit;
            }
            total += c;
        }
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        if (lane == 0) block_counts[blk] = total;
    }
}

// Chain walks for CU candidates per lane at once, without type dispatch: one key column without
// NULLs on both sides, same key width (check_probe guarantees the width class). Every stage's
// loads (keys, bucket heads, then per chain step the node's row id / key / link) are issued for
// all CU candidates before any is used; finished candidates keep reading node 0 so the code stays
// straight-line. Same results as the generic loop in join_chain_kernel.
constexpr int CU = 4;

template <int KW, bool SELP, bool SELB, int NK>
__global__ __launch_bounds__(256) void join_chain_fast_kernel(const void *__restrict__ bkey, const void *__restrict__ bkey2,
                                                              const int32_t *__restrict__ bsel,
                                                              const void *__restrict__ pkey, const void *__restrict__ pkey2,
                                                              const int32_t *__restrict__ psel,
                                                              const int32_t *__restrict__ head, uint64_t mask,
                                                              const int32_t *__restrict__ next,
                                                              const uint16_t *__restrict__ cand, const int32_t *__restrict__ ccount,
                                                              uint16_t *__restrict__ ccnt, int32_t *__restrict__ cmatch,
                                                              int32_t *__restrict__ block_counts, int64_t nb) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        const int cnt = ccount[blk];
        int total = 0;
        for (int t0 = 0; t0 < cnt; t0 += 64 * CU) {
            int t[CU], c[CU], b[CU];
            int32_t hit[CU];
            int64_t r[CU];
            unsigned long long k[CU], k2[CU];
            bool ok[CU];
#pragma unroll
            for (int u = 0; u < CU; u++) {
                t[u] = t0 + u * 64 + lane;
                ok[u] = t[u] < cnt;
                c[u] = 0;
                hit[u] = -1;
                r[u] = blk * JP_CHUNK + cand[blk * JP_CHUNK + (ok[u] ? t[u] : 0)];
            }
            if (SELP) {
#pragma unroll
                for (int u = 0; u < CU; u++) r[u] = psel[r[u]];
            }
#pragma unroll
            for (int u = 0; u < CU; u++) {
                k[u] = load_kw<KW>(pkey, r[u]);
                k2[u] = NK == 2 ? load_kw<KW>(pkey2, r[u]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < CU; u++) {
                uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ k[u]);
                if (NK == 2) hh = mix64(hh ^ k2[u]);
                const int hb = head[hh & mask];
                b[u] = ok[u] ? hb : -1;
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < CU; u++) more = more || b[u] >= 0;
            while (more) {
                int64_t brow[CU];
                int nx[CU];
                unsigned long long bk[CU], bk2[CU];
#pragma unroll
                for (int u = 0; u < CU; u++) {
                    const int bb = b[u] >= 0 ? b[u] : 0;
                    nx[u] = next[bb];
                    brow[u] = SELB ? (int64_t)bsel[bb] : (int64_t)bb;
                }
#pragma unroll
                for (int u = 0; u < CU; u++) {
                    bk[u] = load_kw<KW>(bkey, brow[u]);
                    bk2[u] = NK == 2 ? load_kw<KW>(bkey2, brow[u]) : 0ull;
                }
                more = false;
#pragma unroll
                for (int u = 0; u < CU; u++) {
                    if (b[u] >= 0) {
                        if (bk[u] == k[u] && bk2[u] == k2[u]) { c[u]++; hit[u] = (int32_t)brow[u]; }
                        b[u] = nx[u];   // -1 ends the chain, -2 cannot occur on a chain (NULL keys are never linked)
                    }
                    more = more || b[u] >= 0;
                }
            }
#pragma unroll
            for (int u = 0; u < CU; u++) {
                if (ok[u]) {
                    ccnt[blk * JP_CHUNK + t[u]] = (uint16_t)(c[u] > 65535 ? 65535 : c[u]);
                    cmatch[blk * JP_CHUNK + t[u]] = hit[u];
                    total += c[u];
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        if (lane == 0) block_counts[blk] = total;
    }
}

template <int KW>
static void launch_chain_fast(int grid, hipStream_t st, const JoinSide &B, const JoinSide &P, const int32_t *head, uint64_t mask,
                              const int32_t *next, const uint16_t *cand, const int32_t *ccount, uint16_t *ccnt, int32_t *cmatch,
                              int32_t *counts, int64_t nb) {
#define PH_CHAIN_ARGS B.key[0].data, B.key[1].data, B.sel, P.key[0].data, P.key[1].data, P.sel, head, mask, next, cand, ccount, ccnt, cmatch, counts, nb
#define PH_CHAIN_LAUNCH(NKV)                                                                                   \
    do {                                                                                                       \
        if (P.sel && B.sel) join_chain_fast_kernel<KW, true, true, NKV><<<grid, 256, 0, st>>>(PH_CHAIN_ARGS);   \
        else if (P.sel) join_chain_fast_kernel<KW, true, false, NKV><<<grid, 256, 0, st>>>(PH_CHAIN_ARGS);      \
        else if (B.sel) join_chain_fast_kernel<KW, false, true, NKV><<<grid, 256, 0, st>>>(PH_CHAIN_ARGS);      \
        else join_chain_fast_kernel<KW, false, false, NKV><<<grid, 256, 0, st>>>(PH_CHAIN_ARGS);                \
    } while (0)
    if (P.nkeys == 2) PH_CHAIN_LAUNCH(2);
    else PH_CHAIN_LAUNCH(1);
#undef PH_CHAIN_LAUNCH
#undef PH_CHAIN_ARGS
}

static bool try_chain_fast(int grid, hipStream_t st, const JoinSide &B, const JoinSide &P, const int32_t *head, uint64_t mask,
                           const int32_t *next, const uint16_t *cand, const int32_t *ccount, uint16_t *ccnt, int32_t *cmatch,
                           int32_t *counts, int64_t nb) {
    if (P.nkeys > 2 || P.key[0].validity || B.key[0].validity) return false;
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
    const int kw = width(P.key[0].type);
    if (kw != width(B.key[0].type)) return false;
    if (P.nkeys == 2 && (P.key[1].validity || B.key[1].validity || width(P.key[1].type) != kw || width(B.key[1].type) != kw)) return false;
    if (kw == 4) launch_chain_fast<4>(grid, st, B, P, head, mask, next, cand, ccount, ccnt, cmatch, counts, nb);
    else if (kw == 1) launch_chain_fast<1>(grid, st, B, P, head, mask, next, cand, ccount, ccnt, cmatch, counts, nb);
    else launch_chain_fast<8>(grid, st, B, P, head, mask, next, cand, ccount, ccnt, cmatch, counts, nb);
    return true;
}

__global__ __launch_bounds__(256) void join_emit_kernel(JoinSide B, JoinSide Pr, const int32_t *__restrict__ head,
                                                        uint64_t mask, const int32_t *__restrict__ next,
                                                        const uint16_t *__restrict__ cand, const int32_t *__restrict__ ccount,
                                                        const uint16_t *__restrict__ ccnt, const int32_t *__restrict__ cmatch,
                                                        const int32_t *__restrict__ block_off,
                                                        int64_t nb, int64_t cap, int32_t *__restrict__ out_probe,
                                                        int32_t *__restrict__ out_build) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        const int cnt = ccount[blk];
        int64_t running = block_off[blk];
        for (int t0 = 0; t0 < cnt; t0 += 64) {
            const int t = t0 + lane;
            int c = 0;
            int64_t i = 0;
            if (t < cnt) {
                i = blk * JP_CHUNK + cand[blk * JP_CHUNK + t];
                c = ccnt[blk * JP_CHUNK + t];
                if (c == 65535) c = probe_count(B, Pr, head, mask, next, i, Bloom{});  // saturated: recount
            }
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (c == 1) {  // the chain pass already found the one matching build row
                const int64_t pos = running + incl - 1;
                if (pos < cap) {
                    out_probe[pos] = (int32_t)(Pr.sel ? Pr.sel[i] : i);
                    out_build[pos] = cmatch[blk * JP_CHUNK + t];
                }
            } else if (c > 1) {
                int64_t pos = running + incl - c;
                int64_t r = Pr.sel ? Pr.sel[i] : i;
                unsigned long long k[JOIN_MAX_KEYS];
                uint64_t h;
                load_keys(Pr, r, k, &h);
                for (int b = head[h & mask]; b >= 0; b = next[b]) {
                    int64_t brow = B.sel ? B.sel[b] : b;
                    if (keys_equal(B, brow, k)) {
                        if (pos < cap) {
                            out_probe[pos] = (int32_t)r;
                            out_build[pos] = (int32_t)brow;
                        }
                        pos++;
                    }
                }
            }
            running += __shfl(incl, 63);
        }
    }
}

__global__ __launch_bounds__(256) void join_mark_kernel(JoinSide B, JoinSide Pr, const int32_t *__restrict__ head,
                                                        uint64_t mask, const int32_t *__restrict__ next,
                                                        uint8_t *__restrict__ found, Bloom bl) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Pr.n; i += (int64_t)gridDim.x * 256)
        found[i] = probe_count(B, Pr, head, mask, next, i, bl) > 0 ? 1 : 0;
}

// ---- lookup probe (N:1 joins: the build keys are unique, a primary key). One kernel, no
// candidate / scan / emit pipeline: out[i] = the matching build row of probe position i, or -1.
// The intermediate of a join chain stays positional (late materialisation): every further N:1 join
// is one such lookup addressed through the SAME row-id array, and columns are gathered once at
// the end instead of after every join. Straight-line like join_chain_fast_kernel: the LU probes of
// a lane issue their key reads, then their head reads, then walk their chains together.
constexpr int LU = 4;

template <int KW, bool SELP, bool SELB, int NK>
__global__ __launch_bounds__(256) void join_lookup_fast_kernel(const void *__restrict__ bkey, const void *__restrict__ bkey2,
                                                               const int32_t *__restrict__ bsel,
                                                               const void *__restrict__ pkey, const void *__restrict__ pkey2,
                                                               const int32_t *__restrict__ psel, int64_t n,
                                                               const int32_t *__restrict__ head, uint64_t mask,
                                                               const int32_t *__restrict__ next, Bloom bl,
                                                               int32_t *__restrict__ out, int *__restrict__ stats) {
    int misses = 0, multi = 0;
    for (int64_t base = (int64_t)blockIdx.x * 256 * LU; base < n; base += (int64_t)gridDim.x * 256 * LU) {
        int64_t i[LU], r[LU];
        bool ok[LU];
        unsigned long long k[LU], k2[LU];
        int b[LU], c[LU];
        int32_t hit[LU];
#pragma unroll
        for (int u = 0; u < LU; u++) {
            i[u] = base + u * 256 + threadIdx.x;
            ok[u] = i[u] < n;
            r[u] = ok[u] ? i[u] : 0;
            c[u] = 0;
            hit[u] = -1;
        }
        if (SELP) {
#pragma unroll
            for (int u = 0; u < LU; u++) r[u] = psel[r[u]];
        }
#pragma unroll
        for (int u = 0; u < LU; u++) {
            k[u] = load_kw<KW>(pkey, r[u]);
            k2[u] = NK == 2 ? load_kw<KW>(pkey2, r[u]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < LU; u++) {
            uint64_t hh = mix64(0x9e3779b97f4a7c15ULL ^ k[u]);
            if (NK == 2) hh = mix64(hh ^ k2[u]);
            const int hb = head[hh & mask];   // no bitmap test: a lookup expects its rows to match (foreign keys)
            b[u] = ok[u] ? hb : -1;
        }
        bool more = false;
#pragma unroll
        for (int u = 0; u < LU; u++) more = more || b[u] >= 0;
        while (more) {
            int64_t brow[LU];
            int nx[LU];
            unsigned long long bk[LU], bk2[LU];
#pragma unroll
            for (int u = 0; u < LU; u++) {
                const int bb = b[u] >= 0 ? b[u] : 0;
                nx[u] = next[bb];
                brow[u] = SELB ? (int64_t)bsel[bb] : (int64_t)bb;
            }
#pragma unroll
            for (int u = 0; u < LU; u++) {
                bk[u] = load_kw<KW>(bkey, brow[u]);
                bk2[u] = NK == 2 ? load_kw<KW>(bkey2, brow[u]) : 0ull;
            }
            more = false;
#pragma unroll
            for (int u = 0; u < LU; u++) {
                if (b[u] >= 0) {
                    if (bk[u] == k[u] && bk2[u] == k2[u]) { c[u]++; hit[u] = (int32_t)brow[u]; }
                    b[u] = nx[u];
                }
                more = more || b[u] >= 0;
            }
        }
#pragma unroll
        for (int u = 0; u < LU; u++) {
            if (ok[u]) {
                out[i[u]] = hit[u];
                misses += c[u] == 0;
                multi += c[u] > 1;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { misses += __shfl_xor(misses, o); multi += __shfl_xor(multi, o); }
    if ((threadIdx.x & 63) == 0) {
        if (misses) atomicAdd(stats, misses);
        if (multi) atomicAdd(stats + 1, multi);
    }
}

// any key shape (NULL-able keys, three or four key columns, mixed widths)
__global__ __launch_bounds__(256) void join_lookup_kernel(JoinSide B, JoinSide Pr, const int32_t *__restrict__ head, uint64_t mask,
                                                          const int32_t *__restrict__ next, Bloom bl, int32_t *__restrict__ out,
                                                          int *__restrict__ stats) {
    int misses = 0, multi = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Pr.n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = Pr.sel ? Pr.sel[i] : i;
        unsigned long long k[JOIN_MAX_KEYS];
        uint64_t h;
        int c = 0;
        int32_t hit = -1;
        if (load_keys(Pr, r, k, &h) && bloom_maybe(bl, h))
            for (int b = head[h & mask]; b >= 0; b = next[b]) {
                const int64_t brow = B.sel ? B.sel[b] : b;
                if (keys_equal(B, brow, k)) { c++; hit = (int32_t)brow; }
            }
        out[i] = hit;
        misses += c == 0;
        multi += c > 1;
    }
    for (int o = 32; o > 0; o >>= 1) { misses += __shfl_xor(misses, o); multi += __shfl_xor(multi, o); }
    if ((threadIdx.x & 63) == 0) {
        if (misses) atomicAdd(stats, misses);
        if (multi) atomicAdd(stats + 1, multi);
    }
}

// Dense probes (no bitmap): every position is a candidate. Filling the slices with the identity lets
// them share the chain / emit kernels (and their straight-line fast forms) with selective probes.
__global__ __launch_bounds__(256) void join_cand_all_kernel(int64_t n, uint16_t *__restrict__ cand, int32_t *__restrict__ ccount) {
    const int64_t base = (int64_t)blockIdx.x * JP_CHUNK;
    for (int t = threadIdx.x; t < JP_CHUNK; t += 256) cand[base + t] = (uint16_t)t;
    if (threadIdx.x == 0) ccount[blockIdx.x] = (int32_t)(n - base < JP_CHUNK ? n - base : JP_CHUNK);
}

// MARK / SEMI / ANTI over a selective probe: the candidates' match counts decide the flags
__global__ __launch_bounds__(256) void join_mark_set_kernel(const uint16_t *__restrict__ cand, const int32_t *__restrict__ ccount,
                                                            const uint16_t *__restrict__ ccnt, uint8_t *__restrict__ found,
                                                            int64_t nb) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        const int cnt = ccount[blk];
        for (int t = lane; t < cnt; t += 64)
            if (ccnt[blk * JP_CHUNK + t]) found[blk * JP_CHUNK + cand[blk * JP_CHUNK + t]] = 1;
    }
}


// ---- dense integer keys: the DIRECT table (ph_join_build_range).
// When the caller knows the key column's value range (column statistics) and the build side fills
// that range densely — a primary-key column: orders 15 M keys over a 60 M range, supplier, customer
// — the bucket-head table is addressed by key - lo instead of by a hash: no hash arithmetic, no key
// compare (a chain holds one key's rows only), no collisions, one random 4-byte read per probe, and
// probes in key order (lineitem by l_orderkey) read the table front to back. The reference always
// hashes (join_table.go:197-288); the result is the same pair set. Build, no global atomics for
// unique keys: scatter positions with plain stores (counting the rows stored) -> count the occupied
// slots in one streaming pass: fewer slots than rows means some rows lost a duplicate key, and only
// then the two kernels after it do any work — verify (a row that does not find itself in its slot
// gets next = -3) and link (the losers are chained in front of the winner with atomicExch).
// count[0] rows stored, count[1] occupied slots, count[2] keys outside [lo, hi].
// build-side rows: key column, validity, selection and an optional pushed-down range filter (Filter ->
// build in one pass: a direct table is sized by the key RANGE, so the number of rows that pass need
// not be known on the host — no selection vector, no count read-back before the build)
// slot of a key: key - lo, in range iff < range. 4-byte keys compute it in 32 bits (build_direct clamps the range
// of a 4-byte table to the int32 domain, so the wrapped difference of two int32 values below `range` is exact:
// a false hit would need key + 2^32 < lo + range <= 2^31) — the 64-bit form is two VALU instructions per
// subtract / compare / shift where the candidate kernels spend ~39 per key.
template <int KW>
__device__ __forceinline__ bool direct_slot(long long k, long long lo, unsigned long long range, unsigned long long *off) {
    if (KW == 4) {
        const unsigned o = (unsigned)((int)k - (int)lo);
        *off = o;
        return o < (unsigned)range;
    }
    *off = (unsigned long long)(k - lo);
    return *off < range;
}

struct DirectSrc {
    const void *kcol; const uint8_t *valid; const int32_t *sel;
    int wkind;   // 0 none, 1 int32, 2 int64, 3 uint8 column wdata, rows with wlo <= value <= whi are built
    const void *wdata; long long wlo, whi;
};

template <int KW, bool SEL>
__device__ __forceinline__ bool direct_key(const DirectSrc &S, int64_t i, long long lo, unsigned long long range, unsigned long long *off,
                                           bool *oor) {
    const int64_t r = SEL ? (int64_t)S.sel[i] : i;
    *oor = false;
    if (S.valid && !bit_valid(S.valid, r)) return false;   // NULL key: never inserted, never matches
    if (S.wkind) {
        const long long w = S.wkind == 1 ? (long long)((const int32_t *)S.wdata)[r] : S.wkind == 2 ? ((const long long *)S.wdata)[r]
                                                                                                     : (long long)((const uint8_t *)S.wdata)[r];
        if (w < S.wlo || w > S.whi) return false;
    }
    const long long k = (long long)load_kw<KW>(S.kcol, r);
    const unsigned long long o = (unsigned long long)(k - lo);
    *off = o;
    *oor = o >= range;
    return o < range;
}

// (same-address device atomics cost ~10 ns each, serialised: one per WORKGROUP and one workgroup per
// CU, not one per wave — 8192 of them made this kernel 60 us slower)
constexpr int DT = 1024;

__device__ __forceinline__ void direct_block_add(int v0, int v1, int *__restrict__ c0, int *__restrict__ c1) {
    __shared__ int part[2][DT / 64];
    __syncthreads();   // a second call in one kernel reuses the array
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = v0; part[1][threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int w = 0; w < DT / 64; w++) { a += part[0][w]; b += part[1][w]; }
        if (a) atomicAdd(c0, a);
        if (b && c1) atomicAdd(c1, b);
    }
}

template <int KW, bool SEL>
__global__ __launch_bounds__(DT) void direct_scatter_kernel(DirectSrc S, int64_t n, long long lo,
                                                             unsigned long long range, int32_t *__restrict__ direct,
                                                             int *__restrict__ count, const int *__restrict__ gate) {
    constexpr int U = 4;
    if (gate && *gate == 0) return;   // the sorted fill already built the table
    int ins = 0, out = 0;
    for (int64_t base = (int64_t)blockIdx.x * DT * U; base < n; base += (int64_t)gridDim.x * DT * U) {
        unsigned long long off[U];
        bool ok[U], oor[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * DT + threadIdx.x;
            oor[u] = false;
            ok[u] = i < n && direct_key<KW, SEL>(S, i, lo, range, &off[u], &oor[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (ok[u]) direct[off[u]] = (int32_t)(base + u * DT + threadIdx.x);
            ins += ok[u];
            out += oor[u];
        }
    }
    direct_block_add(ins, out, count, count + 2);
}

// SORTED build keys (a primary-key column in storage order: orders by o_orderkey) need neither the
// initialisation pass nor the scatter: a workgroup that owns rows [r0, r1) owns the slots from
// key[r0] - lo up to key[r1] - lo, composes them 16 K at a time in LDS (-1, then its rows' positions)
// and writes them out whole — one streaming write of the table instead of a fill, a scatter of
// partial lines and a counting read (60 M slots: 40 us against 36 + 96 + 56). Rows of equal keys are
// adjacent, so the occupied-slot count is known here too. The kernel VERIFIES the order it relies on:
// on the first descending pair (or key outside the range) it raises count[3], and the general passes,
// which otherwise leave at once, run instead.
constexpr int SF_ROWS = 512;     // rows per step of a 256-thread workgroup (eight workgroups per CU overlap each other's latencies)
constexpr int SF_TILE = 4096;    // slots composed in LDS at a time
constexpr int SF_GATED_ROWS = 2048;   // rows per step of the gated form

// GATED (declared sorted-unique keys only): the build child's Filter rides along — rows whose wdata value is
// outside [wlo, whi] keep their slot empty — and the fill also writes the occupancy bitmap the candidate pass
// of inner probes tests first (words wholly inside a tile with plain stores, the two edge words of a tile
// with atomic ORs into the pre-cleared bitmap). The rows stored are the bitmap's set bits (ph_join_count).
template <int KW, int WK>   // WK: 0 plain, else the gate column's kind (1 int32, 2 int64, 3 uint8) — a template parameter
                            // because a run-time switch between three load widths put waits into the key prefetch
__global__ __launch_bounds__(256) void direct_sorted_fill_kernel(const void *__restrict__ kcol, int64_t n, long long lo,
                                                                unsigned long long range, int64_t cap4, int32_t *__restrict__ direct,
                                                                int *__restrict__ count, int *__restrict__ partials,
                                                                int *__restrict__ declared, const void *__restrict__ wdata,
                                                                long long wlo, long long whi, unsigned *__restrict__ dbits) {
    constexpr bool GATED = WK != 0;
    // the gated form has no slot image in LDS: bigger chunks (more key bytes in flight per workgroup; with 512 rows
    // it moved 135 MB in 62 us) and a bitmap tile that spans a chunk of 4-slots-per-row keys
    constexpr int ROWS = GATED ? SF_GATED_ROWS : SF_ROWS, TILE = GATED ? 16384 : SF_TILE;
    // declared != NULL: the caller stated (column statistics) that the keys are sorted and unique. The
    // kernel still verifies both, but a violation becomes a deferred error of the ctx (*declared) instead of
    // a fallback: none of the general passes is launched behind this kernel.
    // GATED: only the occupied slots are written (straight to the table) — the bitmap is authoritative and no
    // probe reads a slot whose bit is clear, so the 4 x range bytes of "-1" are never written
    __shared__ int tile[WK != 0 ? 1 : SF_TILE];
    __shared__ long long kk[ROWS + 2];   // key[r0 - 1] (the run test of the first row), the chunk's keys, key[r1]
    long long *keys = kk + 1;
    // the gate column's values of the chunk's rows, RAW as loaded (bytes travel four to a dword): any
    // arithmetic on a prefetched value — a compare, even the widening of a byte — is scheduled next to its
    // load and waits there for the whole prefetch (the kernel took 150 us instead of 70)
    using GT = typename std::conditional<WK == 2, long long, int>::type;
    constexpr int GN = WK == 3 ? (ROWS / 4 + 255) / 256 : ROWS / 256;
    __shared__ GT graw[WK == 0 ? 1 : WK == 3 ? ROWS / 4 : ROWS];
    auto gate_pass = [&](int e) {
        const long long v = WK == 3 ? (long long)reinterpret_cast<const unsigned char *>(graw)[e] : (long long)graw[WK == 3 ? 0 : e];
        return v >= wlo && v <= whi;
    };
    __shared__ unsigned lbits[GATED ? TILE / 32 + 1 : 1];   // occupancy words of the tile being composed
    __shared__ int s_bad;
    int stored = 0, runs = 0;
    const int64_t nchunks = (n + ROWS - 1) / ROWS;
    const long long beyond = (long long)(lo + (long long)range);   // above every valid key: the "next key" of the last row
    // software pipeline: the keys of the NEXT chunk are in flight while this one is composed
    constexpr int KR = (ROWS + 2 + 255) / 256;
    long long kreg[KR];
    GT greg[GN];
    auto fetch = [&](int64_t c) {
        const int64_t r0 = c * ROWS;
#pragma unroll
        for (int q = 0; q < KR; q++) {
            const int64_t i = r0 - 1 + q * 256 + threadIdx.x;
            kreg[q] = (c < nchunks && i >= 0 && i < n && q * 256 + (int)threadIdx.x < ROWS + 2) ? (long long)load_kw<KW>(kcol, i) : beyond;
        }
        if (WK == 3) {   // 512 flag bytes = 128 aligned dwords (the host checked the column's alignment)
#pragma unroll
            for (int q = 0; q < GN; q++) {
                const int64_t d = r0 / 4 + q * 256 + threadIdx.x;
                greg[q] = (c < nchunks && q * 256 + (int)threadIdx.x < ROWS / 4 && d * 4 < n) ? ((const int *)wdata)[d] : 0;
            }
        } else if (GATED) {
#pragma unroll
            for (int q = 0; q < GN; q++) {
                const int64_t i = r0 + q * 256 + threadIdx.x;
                greg[q] = (c < nchunks && i < n) ? ((const GT *)wdata)[i] : (GT)0;
            }
        }
    };
    fetch(blockIdx.x);
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t r0 = c * ROWS, r1 = r0 + ROWS < n ? r0 + ROWS : n;
        const int m = (int)(r1 - r0);
        __syncthreads();   // the previous chunk's readers are done
        if (threadIdx.x == 0) s_bad = 0;
#pragma unroll
        for (int q = 0; q < KR; q++)
            if (q * 256 + (int)threadIdx.x < ROWS + 2) kk[q * 256 + threadIdx.x] = kreg[q];
        if (WK == 3) {
#pragma unroll
            for (int q = 0; q < GN; q++) if (q * 256 + (int)threadIdx.x < ROWS / 4) graw[q * 256 + threadIdx.x] = greg[q];
        }
        else if (GATED) {
#pragma unroll
            for (int q = 0; q < GN; q++) graw[q * 256 + threadIdx.x] = greg[q];
        }
        if (GATED) for (int e = threadIdx.x; e <= TILE / 32; e += 256) lbits[e] = 0;
        fetch(c + gridDim.x);
        __syncthreads();
        if (GATED) {
            // Gated form: three barriers per chunk instead of seven. Rows are verified while they are stored
            // (a chunk that breaks the claim raises the deferred error; what it wrote stays inside the table
            // and the table is void anyway), only occupied slots are written, straight to the table, and the
            // occupancy words are composed in LDS beside them.
            // the span is clamped to the table: a key outside [lo, lo + range) is reported (deferred error) but must
            // not move the span — and with it the stores below — outside the allocation
            int64_t s0 = c == 0 ? 0 : keys[0] - lo;
            int64_t s1 = r1 == n ? cap4 : keys[m] - lo;
            s0 = s0 < 0 ? 0 : s0 > cap4 ? cap4 : s0;
            s1 = s1 > cap4 ? cap4 : s1 < s0 ? s0 : s1;
            bool bad = false;
            for (int64_t t = s0; t < s1 || t == s0; t += TILE) {
                const int w = (int)(s1 - t < TILE ? s1 - t : TILE);
                if (t != s0) {
                    __syncthreads();   // the previous tile's words have been read
                    for (int e = threadIdx.x; e <= TILE / 32; e += 256) lbits[e] = 0;
                    __syncthreads();
                }
                for (int e = threadIdx.x; e < m; e += 256) {
                    const long long k = keys[e];
                    if (t == s0) {
                        bad = bad || (unsigned long long)(k - lo) >= range || k > keys[e + 1];
                        runs += (r0 + e == 0) || keys[e - 1] != k;
                        stored++;
                    }
                    const int64_t off = k - lo - t;
                    if (off >= 0 && off < w && (unsigned long long)(k - lo) < range && gate_pass(e)) {
                        direct[t + off] = (int32_t)(r0 + e);
                        const int b = (int)(t & 31) + (int)off;
                        atomicOr(&lbits[b >> 5], 1u << (b & 31));
                    }
                }
                const int anybad = __syncthreads_or(bad);   // a chunk that breaks the claim stops after this tile: the table is void
                const int sh = (int)(t & 31), nwords = w > 0 ? (sh + w + 31) >> 5 : 0;
                for (int jw = threadIdx.x; jw < nwords; jw += 256) {
                    const unsigned bits = lbits[jw];
                    const bool whole = jw * 32 >= sh && (jw + 1) * 32 <= sh + w;
                    if (whole) dbits[(t >> 5) + jw] = bits;
                    else if (bits) atomicOr(&dbits[(t >> 5) + jw], bits);
                }
                if (anybad) break;
            }
            if (bad) { atomicOr(count + 3, 1); if (declared) atomicOr(declared, 1); }
            continue;
        }
        bool bad = false;
        for (int e = threadIdx.x; e < m; e += 256) {
            const unsigned long long off = (unsigned long long)(keys[e] - lo);
            bad = bad || off >= range || keys[e] > keys[e + 1];
            // the next chunk's first key bounds this chunk's slot span: outside the range it voids the table too
            if (e == m - 1 && r1 < n) bad = bad || (unsigned long long)(keys[m] - lo) >= range;
            runs += (r0 + e == 0) || keys[e - 1] != keys[e];
            stored++;
        }
        if (bad) s_bad = 1;
        __syncthreads();
        if (s_bad) {   // nothing written for this chunk: the general passes rebuild everything (or the deferred error says so)
            if (threadIdx.x == 0) { atomicOr(count + 3, 1); if (declared) atomicOr(declared, 1); }
            continue;
        }
        // keys[m] is the NEXT chunk's first key and has not been checked by this chunk: clamp the span to the table
        int64_t s0 = c == 0 ? 0 : keys[0] - lo;
        int64_t s1 = r1 == n ? cap4 : keys[m] - lo;
        s0 = s0 < 0 ? 0 : s0 > cap4 ? cap4 : s0;
        s1 = s1 > cap4 ? cap4 : s1 < s0 ? s0 : s1;
        for (int64_t t = s0; t < s1; t += SF_TILE) {
            const int w = (int)(s1 - t < SF_TILE ? s1 - t : SF_TILE);
            __syncthreads();
            for (int e = threadIdx.x; e < w; e += 256) tile[e] = -1;
            __syncthreads();
            for (int e = threadIdx.x; e < m; e += 256) {
                const int64_t off = keys[e] - lo - t;
                if (off >= 0 && off < w) tile[off] = (int32_t)(r0 + e);   // equal keys: any of them; the chains are linked afterwards
            }
            __syncthreads();
            // 16-byte stores over the part of [t, t + w) that is 16-byte aligned in the table, 4-byte stores at the ends
            const int head = (int)((4 - (t & 3)) & 3) < w ? (int)((4 - (t & 3)) & 3) : w;
            const int nq = (w - head) / 4;
            if ((int)threadIdx.x < head) direct[t + threadIdx.x] = tile[threadIdx.x];
            int4 *d4 = reinterpret_cast<int4 *>(direct + t + head);
            for (int q = threadIdx.x; q < nq; q += 256) {
                const int e = head + 4 * q;
                d4[q] = make_int4(tile[e], tile[e + 1], tile[e + 2], tile[e + 3]);
            }
            for (int e = head + 4 * nq + threadIdx.x; e < w; e += 256) direct[t + e] = tile[e];
        }
    }
    // per-workgroup partial counts (2048 workgroups adding to one counter would serialise: ~10 ns each)
    __shared__ int part[2][4];
    for (int o = 32; o > 0; o >>= 1) { stored += __shfl_xor(stored, o); runs += __shfl_xor(runs, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = stored; part[1][threadIdx.x >> 6] = runs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int st = part[0][0] + part[0][1] + part[0][2] + part[0][3], rn = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (partials) { partials[2 * blockIdx.x] = st; partials[2 * blockIdx.x + 1] = rn; }
        if (declared && st != rn) atomicOr(declared, 1);   // two adjacent rows share a key
    }
}

// rows stored by a gated fill = bits set in its occupancy bitmap (only ph_join_count asks)
__global__ __launch_bounds__(256) void direct_popcount_kernel(const unsigned *__restrict__ dbits, int64_t words, int *__restrict__ count) {
    __shared__ int part[4];
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) c += __popc(dbits[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(count, part[0] + part[1] + part[2] + part[3]);
}

// the general passes' initialisation when the sorted fill gave up
__global__ __launch_bounds__(256) void direct_refill_kernel(int32_t *__restrict__ direct, int64_t cap4, const int *__restrict__ count) {
    if (count[3] == 0) return;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    int4 *d4 = reinterpret_cast<int4 *>(direct);
    for (int64_t i = t; i < cap4 / 4; i += step) d4[i] = make_int4(-1, -1, -1, -1);
}
// one workgroup: the fill's partial counts -> count[0] (rows stored), count[1] (slots occupied), or nothing
// when the fill gave up (the general passes count for themselves)
__global__ __launch_bounds__(256) void direct_recount_kernel(int *__restrict__ count, const int *__restrict__ partials, int nparts) {
    if (count[3] != 0) return;
    int a = 0, b = 0;
    for (int e = threadIdx.x; e < nparts; e += 256) { a += partials[2 * e]; b += partials[2 * e + 1]; }
    __shared__ int sa[4], sb[4];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { count[0] = sa[0] + sa[1] + sa[2] + sa[3]; count[1] = sb[0] + sb[1] + sb[2] + sb[3]; }
}

// small build sides (<= 256 K rows): ONE kernel after the initialisation — head insertion with
// atomicExch as in join_build_kernel (a few microseconds of scattered atomics at this size, against
// four more launches at ~5 us each), which links duplicate keys on the spot
template <int KW, bool SEL>
__global__ __launch_bounds__(DT) void direct_small_kernel(DirectSrc S, int64_t n, long long lo,
                                                         unsigned long long range, int32_t *__restrict__ direct,
                                                         int32_t *__restrict__ next, int *__restrict__ count,
                                                         unsigned *__restrict__ coarse, int cshift, unsigned *__restrict__ dbits) {
    int ins = 0, first = 0, out = 0;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        unsigned long long off;
        bool oor = false;
        if (direct_key<KW, SEL>(S, i, lo, range, &off, &oor)) {
            const int32_t old = atomicExch(&direct[off], (int32_t)i);
            if (coarse) { const unsigned cb = (unsigned)(off >> cshift); atomicOr(&coarse[cb >> 5], 1u << (cb & 31)); }
            if (dbits) atomicOr(&dbits[off >> 5], 1u << (off & 31));
            next[i] = old;
            ins++;
            first += old < 0;
        } else next[i] = -2;
        out += oor;
    }
    direct_block_add(ins, first, count, count + 1);   // rows stored, slots occupied
    direct_block_add(out, 0, count + 2, nullptr);
}

// occupied slots of the table (cap4 is a multiple of 4; the padding slots are -1)
// ... and, for tables of <= 8 M slots, the occupancy bitmap: a thread's 4 slots are a nibble, 8 lanes a word
__global__ __launch_bounds__(DT) void direct_occupied_kernel(const int32_t *__restrict__ direct, int64_t cap4, int *__restrict__ count,
                                                            unsigned *__restrict__ dbits, const int *__restrict__ gate) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    if (gate && *gate == 0) return;
    const v4i *d4 = reinterpret_cast<const v4i *>(direct);
    const int64_t nq = cap4 / 4;
    int occ = 0;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i - (threadIdx.x & 63) < nq; i += (int64_t)gridDim.x * DT) {   // wave-uniform trip count
        v4i v = {-1, -1, -1, -1};
        if (i < nq) v = __builtin_nontemporal_load(d4 + i);
        occ += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
        if (dbits) {
            unsigned w = ((v.x >= 0) | ((v.y >= 0) << 1) | ((v.z >= 0) << 2) | ((v.w >= 0) << 3)) << (4 * (threadIdx.x & 7));
            w |= __shfl_xor(w, 1);
            w |= __shfl_xor(w, 2);
            w |= __shfl_xor(w, 4);
            if ((threadIdx.x & 7) == 0 && i < nq) dbits[i >> 3] = w;   // slots 4i .. 4i+31
        }
    }
    direct_block_add(occ, 0, count + 1, nullptr);
}

template <int KW, bool SEL>
__global__ __launch_bounds__(256) void direct_verify_kernel(DirectSrc S, int64_t n, long long lo,
                                                            unsigned long long range, const int32_t *__restrict__ direct,
                                                            int32_t *__restrict__ next, const int *__restrict__ count) {
    constexpr int U = 4;
    if (count[0] == count[1]) return;   // every stored row owns its slot: unique keys (the expected case)
    for (int64_t base = (int64_t)blockIdx.x * 256 * U; base < n; base += (int64_t)gridDim.x * 256 * U) {
        unsigned long long off[U];
        bool ok[U], oor[U];
        int32_t d[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            oor[u] = false;
            ok[u] = i < n && direct_key<KW, SEL>(S, i, lo, range, &off[u], &oor[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = direct[ok[u] ? off[u] : 0];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i >= n) continue;
            const bool mine = ok[u] && d[u] == (int32_t)i;
            next[i] = !ok[u] ? -2 : mine ? -1 : -3;
        }
    }
}

template <int KW, bool SEL>
__global__ __launch_bounds__(256) void direct_dups_kernel(DirectSrc S, int64_t n, long long lo,
                                                          unsigned long long range, int32_t *__restrict__ direct,
                                                          int32_t *__restrict__ next, const int *__restrict__ count) {
    if (count[0] == count[1]) return;   // unique keys: nothing to link
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (next[i] != -3) continue;
        unsigned long long off;
        bool oor;
        if (direct_key<KW, SEL>(S, i, lo, range, &off, &oor)) next[i] = atomicExch(&direct[off], (int32_t)i);
    }
}

// probe side, lookup (MODE 0) and mark (MODE 2): DU probes per lane issue their key reads, then
// their table reads together.
constexpr int DU = 4;

template <int KW, bool SELP, bool SELB, int MODE>
__global__ __launch_bounds__(256) void direct_probe_kernel(const void *__restrict__ pkey, const uint8_t *__restrict__ pvalid,
                                                           const int32_t *__restrict__ psel, int64_t n, long long lo,
                                                           unsigned long long range, const int32_t *__restrict__ direct,
                                                           const int32_t *__restrict__ next, const int32_t *__restrict__ bsel,
                                                           const int *__restrict__ bcount, int32_t nbuild, int32_t *__restrict__ out,
                                                           uint8_t *__restrict__ found, int *__restrict__ stats,
                                                           const unsigned *__restrict__ abits) {
    // abits: the table's occupancy bitmap is AUTHORITATIVE (gated sorted fill: only occupied slots were ever
    // written, the rest of the slot array is uninitialised memory) — a slot is read only when its bit is set
    const bool dups = bcount[0] != bcount[1];   // rows stored vs slots occupied
    int misses = 0, multi = 0;
    for (int64_t base = (int64_t)blockIdx.x * 256 * DU; base < n; base += (int64_t)gridDim.x * 256 * DU) {
        int64_t r[DU];
        bool ok[DU];
        long long k[DU];
        int32_t b[DU];
        int c[DU];
#pragma unroll
        for (int u = 0; u < DU; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            ok[u] = i < n;
            r[u] = ok[u] ? i : 0;
        }
        if (SELP) {
#pragma unroll
            for (int u = 0; u < DU; u++) r[u] = psel[r[u]];
        }
#pragma unroll
        for (int u = 0; u < DU; u++) {
            k[u] = (long long)load_kw<KW>(pkey, r[u]);
            if (pvalid) ok[u] = ok[u] && bit_valid(pvalid, r[u]);
        }
#pragma unroll
        for (int u = 0; u < DU; u++) {
            unsigned long long off;
            ok[u] = direct_slot<KW>(k[u], lo, range, &off) && ok[u];
            if (abits) ok[u] = ok[u] && ((abits[(ok[u] ? off : 0) >> 5] >> (off & 31)) & 1u);
            const int32_t d = direct[ok[u] ? off : 0];
            b[u] = ok[u] && (unsigned)d < (unsigned)nbuild ? d : -1;   // anything but a build row reads as empty
            c[u] = b[u] >= 0 ? 1 : 0;
        }
        if (dups) {   // duplicate build keys: a chain holds every row of the key; report the last, count all
#pragma unroll
            for (int u = 0; u < DU; u++)
                if (b[u] >= 0)
                    for (int32_t x = next[b[u]]; x >= 0; x = next[x]) { c[u]++; if (MODE == 0) b[u] = x; }
        }
        if (SELB) {
#pragma unroll
            for (int u = 0; u < DU; u++) b[u] = b[u] >= 0 ? bsel[b[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < DU; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i >= n) continue;
            if (MODE == 0) { out[i] = b[u]; misses += c[u] == 0; multi += c[u] > 1; }
            else found[i] = c[u] > 0 ? 1 : 0;
        }
    }
    if (MODE == 0) {
        for (int o = 32; o > 0; o >>= 1) { misses += __shfl_xor(misses, o); multi += __shfl_xor(multi, o); }
        if ((threadIdx.x & 63) == 0) {
            if (misses) atomicAdd(stats, misses);
            if (multi) atomicAdd(stats + 1, multi);
        }
    }
}

// inner probe, pass 1: 256 threads per 2048-row block stream the probe keys (after the pushed-down
// range filter WK: 0 none, 1 int32, 2 int64, 3 uint8 column), read the table, and write the block's
// matching positions in order (ballot ranks, no atomics) with the slot's chain head beside them —
// the shape of join_cand_fast_kernel, but the table read IS the exact test, so there is no chain pass.
// COARSE: a sparse table (few build keys in a small range: Q9's 109 k pink parts of 2 M) also has a
// 1 Mbit bitmap of occupied slot groups that the probe workgroup keeps in LDS; only probes whose
// group is occupied read the table through L2 (the others read slot 0: one request per wave).
struct DirectCand {
    const void *keycol; const uint8_t *pvalid; const int32_t *sel; int64_t n; long long lo; unsigned long long range;
    const int32_t *direct; const int32_t *next; const int *bcount; const void *wdata; long long wlo, whi;
    uint16_t *cand; int32_t *cmatch; uint16_t *ccnt; int32_t *ccount; int32_t *block_counts;
    int cshift;   // coarse bit = slot >> cshift
    const unsigned *dbits;   // one bit per slot (tables of <= 8 M slots: L2 resident) or NULL
    const uint8_t *bflags;   // residual predicate on the BUILD row (a byte per build row, non-zero = keep) or NULL
    int32_t nbuild;          // build rows: a slot holding anything else reads as empty (a fill that gave up on keys
                             // declared sorted leaves slots unwritten until the deferred error is seen)
};

template <int KW, int WK, bool SEL, bool COARSE>
__device__ __forceinline__ void direct_cand_block(const DirectCand &D, int64_t blk, bool have, int tid, int (*wc)[4], int *wtot,
                                                  const unsigned *co_lds, bool dups) {
    const int64_t base = blk * JP_CHUNK;
    const int lane = tid & 63, wv = tid >> 6;
    unsigned long long bal[JP_ROUNDS];
    int64_t r[JP_ROUNDS];
    bool ok[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        const int64_t i = base + rr * 256 + tid;
        ok[rr] = have && i < D.n;
        const int64_t ic = ok[rr] ? i : 0;
        r[rr] = SEL ? (int64_t)D.sel[ic] : ic;
    }
    if (WK != 0) {
        long long w[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++)
            w[rr] = SEL ? (WK == 1 ? (long long)((const int32_t *)D.wdata)[r[rr]]
                           : WK == 2 ? ((const int64_t *)D.wdata)[r[rr]] : (long long)((const uint8_t *)D.wdata)[r[rr]])
                        : (WK == 1 ? (long long)__builtin_nontemporal_load((const int32_t *)D.wdata + r[rr])
                           : WK == 2 ? (long long)__builtin_nontemporal_load((const int64_t *)D.wdata + r[rr])
                                     : (long long)__builtin_nontemporal_load((const uint8_t *)D.wdata + r[rr]));
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            ok[rr] = ok[rr] && w[rr] >= D.wlo && w[rr] <= D.whi;
            if (!ok[rr]) r[rr] = 0;
        }
    }
    long long k[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        k[rr] = (long long)(SEL ? load_kw<KW>(D.keycol, r[rr]) : load_kw_nt<KW>(D.keycol, r[rr]));
        if (D.pvalid) ok[rr] = ok[rr] && bit_valid(D.pvalid, r[rr]);
    }
    int32_t d[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        const unsigned long long off = (unsigned long long)(k[rr] - D.lo);
        ok[rr] = ok[rr] && off < D.range;
        if (COARSE) {
            const unsigned cb = (unsigned)((ok[rr] ? off : 0) >> D.cshift);
            ok[rr] = ok[rr] && ((co_lds[cb >> 5] >> (cb & 31)) & 1u);
        }
        k[rr] = (long long)(ok[rr] ? off : 0);
    }
    if (D.dbits) {   // exact occupancy bit from L2 first: only real matches read the (much larger) slot array
        unsigned bw[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) bw[rr] = D.dbits[k[rr] >> 5];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) { ok[rr] = ok[rr] && ((bw[rr] >> (k[rr] & 31)) & 1u); if (!ok[rr]) k[rr] = 0; }
    }
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) { d[rr] = D.direct[k[rr]]; d[rr] = (unsigned)d[rr] < (unsigned)D.nbuild ? d[rr] : -1; }
    if (D.bflags && !dups) {   // unique keys: the slot's row either passes the residual predicate or the probe row has no pair
        uint8_t fl[JP_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) fl[rr] = D.bflags[ok[rr] && d[rr] >= 0 ? d[rr] : 0];
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) if (!fl[rr]) d[rr] = -1;
    }
    int total = 0;
    int c[JP_ROUNDS];
#pragma unroll
    for (int rr = 0; rr < JP_ROUNDS; rr++) {
        bool take = ok[rr] && d[rr] >= 0;
        c[rr] = take ? 1 : 0;
        if (dups && take) {
            c[rr] = 0;
            for (int32_t x = d[rr]; x >= 0; x = D.next[x]) c[rr] += !D.bflags || D.bflags[x];
            take = c[rr] > 0;
        }
        total += c[rr];
        bal[rr] = __ballot(take);
        if (lane == 0) wc[rr][wv] = __popcll(bal[rr]);
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0) wtot[wv] = total;
    __syncthreads();
    if (have) {
        int before = 0;
#pragma unroll
        for (int rr = 0; rr < JP_ROUNDS; rr++) {
            int off = before;
            for (int q = 0; q < wv; q++) off += wc[rr][q];
            if ((bal[rr] >> lane) & 1) {
                const int64_t slot = base + off + __popcll(bal[rr] & ((1ull << lane) - 1ull));
                D.cand[slot] = (uint16_t)(rr * 256 + tid);
                D.cmatch[slot] = d[rr];
                D.ccnt[slot] = (uint16_t)(c[rr] > 65535 ? 65535 : c[rr]);
            }
            before += wc[rr][0] + wc[rr][1] + wc[rr][2] + wc[rr][3];
        }
        if (tid == 0) {
            D.ccount[blk] = before;
            D.block_counts[blk] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        }
    }
}

template <int KW, int WK, bool SEL>
__global__ __launch_bounds__(256) void direct_cand_kernel(DirectCand D) {
    __shared__ int wc[JP_ROUNDS][4];
    __shared__ int wtot[4];
    direct_cand_block<KW, WK, SEL, false>(D, blockIdx.x, true, threadIdx.x, wc, wtot, nullptr, D.bcount[0] != D.bcount[1]);
}

// one 1024-thread workgroup per CU = four 256-thread groups, each owning one block per step
template <int KW, int WK, bool SEL>
__global__ __launch_bounds__(1024) void direct_cand_coarse_kernel(DirectCand D, const unsigned *__restrict__ coarse, int64_t nb) {
    extern __shared__ unsigned dco_lds[];          // CO_WORDS words, then the per-group wave counts
    int (*wc)[JP_ROUNDS][4] = reinterpret_cast<int (*)[JP_ROUNDS][4]>(dco_lds + CO_WORDS);
    int (*wtot)[4] = reinterpret_cast<int (*)[4]>(dco_lds + CO_WORDS + 4 * JP_ROUNDS * 4);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(coarse);
        uint4 *dst = reinterpret_cast<uint4 *>(dco_lds);
        for (int e = threadIdx.x; e < CO_WORDS / 4; e += 1024) dst[e] = src[e];
    }
    __syncthreads();
    const bool dups = D.bcount[0] != D.bcount[1];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (int64_t it = 0; (it * gridDim.x + blockIdx.x) * 4 < nb; it++) {
        const int64_t blk = (it * gridDim.x + blockIdx.x) * 4 + grp;
        direct_cand_block<KW, WK, SEL, true>(D, blk, blk < nb, tid, wc[grp], wtot[grp], dco_lds, dups);
        __syncthreads();   // the counts are rewritten in the next step
    }
}

// Vectorised form for the common probe shape (identity selection, no NULLs, 16-byte aligned columns):
// a lane owns 8 CONSECUTIVE rows and reads them with 16-byte non-temporal loads; ranks come from 8
// ballots. The round-per-row form above spends ~75 VALU instructions per row (64-bit index
// arithmetic, a scalar load and a ballot round per row) and was VALU bound: 60 M 4-byte keys took
// 174 us where the key column streams in 40.
template <int KW, bool COARSE, bool WAVE>
__device__ __forceinline__ void direct_cand_block_keys(const DirectCand &D, int64_t blk, bool have, int tid, int *wcv, int *wtot,
                                                       const unsigned *co_lds, bool dups, long long (&k)[8], bool (&ok)[8],
                                                       int sub, int &run_cands, int &run_total);

template <int KW, int WK, bool COARSE>
__device__ __forceinline__ void direct_cand_block_vec(const DirectCand &D, int64_t blk, bool have, int tid, int *wcv, int *wtot,
                                                      const unsigned *co_lds, bool dups) {
    const int64_t base = blk * JP_CHUNK;
    const int lane = tid & 63, wv = tid >> 6;
    long long k[8], k2[8];
    bool ok[8];
    dc_block_keys<KW, WK, 1>(D.keycol, nullptr, D.wdata, D.wlo, D.whi, base + wv * 512, lane, have && base + JP_CHUNK <= D.n, have, D.n, k, k2, ok);
    int rc = 0, rt = 0;
    direct_cand_block_keys<KW, COARSE, false>(D, blk, have, tid, wcv, wtot, co_lds, dups, k, ok, 0, rc, rt);
}

// the same with the keys (and filter results) already in registers. WAVE: one wave works through the
// block alone, step `sub` of four (the persistent coarse kernel): the offsets inside the block run in
// run_cands / run_total instead of LDS counts, and there is no barrier
template <int KW, bool COARSE, bool WAVE>
__device__ __forceinline__ void direct_cand_block_keys(const DirectCand &D, int64_t blk, bool have, int tid, int *wcv, int *wtot,
                                                       const unsigned *co_lds, bool dups, long long (&k)[8], bool (&ok)[8],
                                                       int sub, int &run_cands, int &run_total) {
    constexpr int R = 16 / KW, G = 8 / R;
    const int64_t base = blk * JP_CHUNK;
    const int lane = tid & 63, wv = WAVE ? sub : tid >> 6;
    int32_t d[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        unsigned long long off;
        ok[s] = direct_slot<KW>(k[s], D.lo, D.range, &off) && ok[s];
        if (COARSE) {
            const unsigned cb = (unsigned)((ok[s] ? off : 0) >> D.cshift);
            ok[s] = ok[s] && ((co_lds[cb >> 5] >> (cb & 31)) & 1u);
        }
        k[s] = (long long)(ok[s] ? off : 0);
    }
    if (D.dbits) {   // exact occupancy bit from L2 first: only real matches read the (much larger) slot array
        unsigned bw[8];
#pragma unroll
        for (int s = 0; s < 8; s++) bw[s] = D.dbits[k[s] >> 5];
#pragma unroll
        for (int s = 0; s < 8; s++) { ok[s] = ok[s] && ((bw[s] >> (k[s] & 31)) & 1u); if (!ok[s]) k[s] = 0; }
    }
#pragma unroll
    for (int s = 0; s < 8; s++) { d[s] = D.direct[k[s]]; d[s] = (unsigned)d[s] < (unsigned)D.nbuild ? d[s] : -1; }
    if (D.bflags && !dups) {   // unique keys: the slot's row either passes the residual predicate or the probe row has no pair
        uint8_t fl[8];
#pragma unroll
        for (int s = 0; s < 8; s++) fl[s] = D.bflags[ok[s] && d[s] >= 0 ? d[s] : 0];
#pragma unroll
        for (int s = 0; s < 8; s++) if (!fl[s]) d[s] = -1;
    }
    int total = 0;
    int c[8];
    bool take[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        take[s] = ok[s] && d[s] >= 0;
        c[s] = take[s] ? 1 : 0;
        if (dups && take[s]) {
            c[s] = 0;
            for (int32_t x = d[s]; x >= 0; x = D.next[x]) c[s] += !D.bflags || D.bflags[x];
            take[s] = c[s] > 0;
        }
        total += c[s];
    }
    int first[G];
    const int wave_cands = dc_wave_ranks<R>(take, first);
    if (dups) { for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o); } else total = wave_cands;
    if (!WAVE) {
        if (lane == 0) { wcv[wv] = wave_cands; wtot[wv] = total; }
        __syncthreads();
    }
    if (have) {
        int off = run_cands;
        if (!WAVE) for (int q = 0; q < wv; q++) off += wcv[q];
#pragma unroll
        for (int g = 0; g < G; g++) {
            int64_t slot = base + off + first[g];
#pragma unroll
            for (int e = 0; e < R; e++) {
                const int s = g * R + e;
                if (take[s]) {
                    D.cand[slot] = (uint16_t)(wv * 512 + (g * 64 + lane) * R + e);
                    D.cmatch[slot] = d[s];
                    if (dups) D.ccnt[slot] = (uint16_t)(c[s] > 65535 ? 65535 : c[s]);
                    slot++;
                }
            }
        }
        if (!WAVE && tid == 0) {
            D.ccount[blk] = wcv[0] + wcv[1] + wcv[2] + wcv[3];
            D.block_counts[blk] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        }
    }
    if (WAVE) { run_cands += wave_cands; run_total += total; }
}

template <int KW, int WK>
__global__ __launch_bounds__(256) void direct_cand_vec_kernel(DirectCand D) {
    __shared__ int wcv[4], wtot[4];
    direct_cand_block_vec<KW, WK, false>(D, blockIdx.x, true, threadIdx.x, wcv, wtot, nullptr, D.bcount[0] != D.bcount[1]);
}

template <int KW, int WK>
__global__ __launch_bounds__(1024) void direct_cand_coarse_vec_kernel(DirectCand D, const unsigned *__restrict__ coarse, int64_t nb) {
    extern __shared__ unsigned dcv_lds[];          // CO_WORDS words, then the per-group wave counts
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(coarse);
        uint4 *dst = reinterpret_cast<uint4 *>(dcv_lds);
        for (int e = threadIdx.x; e < CO_WORDS / 4; e += 1024) dst[e] = src[e];
    }
    __syncthreads();
    const bool dups = D.bcount[0] != D.bcount[1];
    // One workgroup per CU (the bitmap takes the LDS), and every WAVE works alone: it takes a whole 2048-row
    // block in four 512-row steps, ranks its survivors with ballots and keeps the block's running offset in
    // a register — no LDS counts, no barrier. (With four waves per block and a barrier per step the 16
    // waves of the CU moved in lockstep through key read -> bitmap test -> table read -> barrier: 240 MB of
    // l_partkey took 127 us.) The keys of the next step are requested before the current step's table reads.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t blk = (int64_t)blockIdx.x * 16 + wave; blk < nb; blk += (int64_t)gridDim.x * 16) {
        const bool full = (blk + 1) * JP_CHUNK <= D.n;
        int run_cands = 0, run_total = 0;
        long long k[8], k2[8], kn[8];
        bool ok[8], okn[8];
        dc_block_keys<KW, WK, 1>(D.keycol, nullptr, D.wdata, D.wlo, D.whi, blk * JP_CHUNK, lane, full, true, D.n, k, k2, ok);
#pragma unroll 1
        for (int sub = 0; sub < 4; sub++) {   // not unrolled: four copies of the body overflow the instruction cache
            if (sub < 3) dc_block_keys<KW, WK, 1>(D.keycol, nullptr, D.wdata, D.wlo, D.whi, blk * JP_CHUNK + (sub + 1) * 512, lane, full, true, D.n, kn, k2, okn);
            direct_cand_block_keys<KW, true, true>(D, blk, true, lane, nullptr, nullptr, dcv_lds, dups, k, ok, sub, run_cands, run_total);
            if (sub < 3) {
#pragma unroll
                for (int s = 0; s < 8; s++) { k[s] = kn[s]; ok[s] = okn[s]; }
            }
        }
        if (lane == 0) { D.ccount[blk] = run_cands; D.block_counts[blk] = run_total; }
    }
}

// inner probe, pass 2: one wave per block writes the pairs at the block's offset, ordered by probe
// position then chain order; a candidate of a unique key is one pair straight from cmatch
template <bool SELP, bool SELB>
__global__ __launch_bounds__(256) void direct_emit_kernel(const int32_t *__restrict__ psel, const int32_t *__restrict__ next,
                                                          const int32_t *__restrict__ bsel, const uint16_t *__restrict__ cand,
                                                          const int32_t *__restrict__ cmatch, const uint16_t *__restrict__ ccnt,
                                                          const int32_t *__restrict__ ccount, const int32_t *__restrict__ block_off,
                                                          const int *__restrict__ bcount, const uint8_t *__restrict__ bflags,
                                                          int64_t nb, int64_t cap, int32_t *__restrict__ out_probe,
                                                          int32_t *__restrict__ out_build) {
    const int lane = threadIdx.x & 63;
    const bool dups = bcount[0] != bcount[1];
    const bool walk1 = dups && bflags != nullptr;   // a single pair of a chain need not be the chain's head
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nb; blk += nw) {
        const int cnt = ccount[blk];
        int64_t running = block_off[blk];
        for (int t0 = 0; t0 < cnt; t0 += 64) {
            const int t = t0 + lane;
            int c = 0;
            int32_t first = -1;
            int64_t r = 0;
            if (t < cnt) {
                const int64_t i = blk * JP_CHUNK + cand[blk * JP_CHUNK + t];
                r = SELP ? (int64_t)psel[i] : i;
                c = dups ? (int)ccnt[blk * JP_CHUNK + t] : 1;   // the candidate pass only writes counts when chains exist
                first = cmatch[blk * JP_CHUNK + t];
                if (c == 65535) { c = 0; for (int32_t x = first; x >= 0; x = next[x]) c += !bflags || bflags[x]; }   // saturated: recount
            }
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (c == 1 && !walk1) {
                const int64_t pos = running + incl - 1;
                if (pos < cap) { out_probe[pos] = (int32_t)r; out_build[pos] = SELB ? bsel[first] : first; }
            } else if (c >= 1) {
                int64_t pos = running + incl - c;
                for (int32_t x = first; x >= 0; x = next[x]) {
                    if (bflags && !bflags[x]) continue;
                    if (pos < cap) { out_probe[pos] = (int32_t)r; out_build[pos] = SELB ? bsel[x] : x; }
                    pos++;
                }
            }
            running += __shfl(incl, 63);
        }
    }
}

// mark probe with a pushed-down range filter on the probe side, vectorised shape only (identity selection,
// no NULLs, 16-byte aligned columns): found[i] = filter(i) && key i is in the table. One pass, one byte
// per row out, no counts — a semi-join flag array that a later probe takes as its residual predicate.
template <int KW, int WK>
__global__ __launch_bounds__(256) void direct_mark_where_kernel(const void *__restrict__ keycol, int64_t n, long long lo, unsigned long long range,
                                                                const int32_t *__restrict__ direct, const unsigned *__restrict__ dbits,
                                                                int32_t nbuild, const void *__restrict__ wdata, long long wlo, long long whi,
                                                                uint8_t *__restrict__ found) {
    constexpr int R = 16 / KW, G = 8 / R;
    const int64_t base = (int64_t)blockIdx.x * JP_CHUNK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool full = base + JP_CHUNK <= n;
    long long k[8], k2[8];
    bool ok[8];
    dc_block_keys<KW, WK, 1>(keycol, nullptr, wdata, wlo, whi, base + wv * 512, lane, full, true, n, k, k2, ok);
    int32_t d[8];
    if (dbits) {   // membership is all a mark needs: the occupancy bitmap (L2 resident) instead of the 32x larger slot array
#pragma unroll
        for (int s = 0; s < 8; s++) {
            unsigned long long off;
            ok[s] = direct_slot<KW>(k[s], lo, range, &off) && ok[s];
            const unsigned long long o = ok[s] ? off : 0;
            d[s] = ((dbits[o >> 5] >> (o & 31)) & 1u) ? 0 : -1;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            unsigned long long off;
            ok[s] = direct_slot<KW>(k[s], lo, range, &off) && ok[s];
            d[s] = direct[ok[s] ? off : 0];
            d[s] = (unsigned)d[s] < (unsigned)nbuild ? d[s] : -1;
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int64_t row = base + wv * 512 + (int64_t)(g * 64 + lane) * R;
        if (full) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < R; e++) w |= (ok[g * R + e] && d[g * R + e] >= 0 ? 1u : 0u) << (8 * e);
            if (R == 4) *reinterpret_cast<unsigned *>(found + row) = w;
            else *reinterpret_cast<unsigned short *>(found + row) = (unsigned short)w;
        } else {
#pragma unroll
            for (int e = 0; e < R; e++)
                if (row + e < n) found[row + e] = ok[g * R + e] && d[g * R + e] >= 0 ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------- radix-partitioned hash join
// Big build sides whose keys are NOT dense in a range (the direct table does not apply). The node table
// above answers a probe with two dependent random reads in HBM — a 128-byte line each for 4 + 16 bytes,
// whatever order the probe keys arrive in: 60 M probes of 15 M random 62-bit keys 3.1 ms, 7.7 GB moved. Here both
// sides are partitioned by key hash first, so that a probe only ever touches a table that is resident on chip:
//   build  rows -> P x SUB bins (count, scan, LDS-staged scatter of {key, row}); ONE workgroup per bin builds the
//          bin's open-addressing table (S slots of {key, row}, linear probing, duplicates in slots of their own)
//          IN LDS and writes its image out whole — no global atomic, no random write;
//   probe  rows -> P partitions (count, scan, LDS-staged scatter of {key, row}: 12 bytes per row in runs), then
//          partition p's rows probe the SUB images of partition p (2 MiB: resident in the L2 of the XCD that the
//          workgroups of p are dealt to — consecutive workgroup ids of one residue mod 8, see rj_map) and emit pairs
//          through one reserving atomic per wave and probe step.
// h = big_hash(key): bits [36, 36 + logSUB) pick the table inside a partition, the bits above them the partition,
// the low bits the slot.
constexpr int RJ_T = 1024;
constexpr int RJ_U = 4;                      // rows per thread and chunk
constexpr int RJ_CH = RJ_T * RJ_U;           // rows staged per chunk
constexpr int RJ_SLOTS = 8192;               // slots of one table image: 128 KiB of LDS while it is built
constexpr int RJ_SHIFT = 36;

struct RjSide {
    const void *k0, *k1;
    const uint8_t *v0, *v1;
    const int32_t *sel;
    int64_t n;
};

struct RjPart {
    int bins, shift;                         // bin = (h >> shift) & (bins - 1)
    int64_t rows_per_wg;
    int32_t *counts;                         // [bins][nwg] -> offsets after the scan
    ulonglong2 *rec;                         // records {key, row} in bin order (16 bytes: one store, runs of 256 bytes and more per bin)
};

template <int KW, int NK, bool SEL>
__device__ __forceinline__ void rj_keys(const RjSide &S, int64_t base, int64_t i1, unsigned long long (&key)[RJ_U], int32_t (&row)[RJ_U], bool (&ok)[RJ_U]) {
    int64_t r[RJ_U];
#pragma unroll
    for (int u = 0; u < RJ_U; u++) {
        const int64_t i = base + u * RJ_T + threadIdx.x;
        ok[u] = i < i1;
        const int64_t ic = ok[u] ? i : i1 - 1;
        r[u] = SEL ? (int64_t)S.sel[ic] : ic;
        row[u] = (int32_t)r[u];
    }
#pragma unroll
    for (int u = 0; u < RJ_U; u++) {
        const unsigned long long a = load_kw<KW>(S.k0, r[u]);
        const unsigned long long b = NK == 2 ? load_kw<KW>(S.k1, r[u]) : 0ull;
        key[u] = big_pack<KW, NK>(a, b);
        if (S.v0) ok[u] = ok[u] && bit_valid(S.v0, r[u]);          // NULL keys neither enter a table nor match
        if (NK == 2 && S.v1) ok[u] = ok[u] && bit_valid(S.v1, r[u]);
    }
}

template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(RJ_T) void rj_count_kernel(RjSide S, RjPart Q) {
    extern __shared__ int rj_hist[];
    for (int e = threadIdx.x; e < Q.bins; e += RJ_T) rj_hist[e] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * Q.rows_per_wg, i1 = i0 + Q.rows_per_wg < S.n ? i0 + Q.rows_per_wg : S.n;
    for (int64_t base = i0; base < i1; base += RJ_CH) {
        unsigned long long key[RJ_U];
        int32_t row[RJ_U];
        bool ok[RJ_U];
        rj_keys<KW, NK, SEL>(S, base, i1, key, row, ok);
#pragma unroll
        for (int u = 0; u < RJ_U; u++)
            if (ok[u]) atomicAdd(&rj_hist[(int)((big_hash(key[u]) >> Q.shift) & (uint64_t)(Q.bins - 1))], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Q.bins; e += RJ_T) Q.counts[(int64_t)e * gridDim.x + blockIdx.x] = rj_hist[e];
}

// every chunk of RJ_CH rows is ordered by bin in LDS first and leaves as runs of consecutive records per bin.
// Two words of LDS per bin (8192 bins + the staged chunk = 128 KiB): `hist` holds the chunk's counts, then — scanned in
// place — the chunk offsets; `cursor` the bin's next position in the output.
template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(RJ_T) void rj_scatter_kernel(RjSide S, RjPart Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rj_lds[];
    const int B = Q.bins;
    int *hist = reinterpret_cast<int *>(rj_lds);
    int *cursor = hist + B;
    ulonglong2 *s_rec = reinterpret_cast<ulonglong2 *>(cursor + B);   // (B is even: 16-byte aligned)
    int *s_dst = reinterpret_cast<int *>(s_rec + RJ_CH);
    __shared__ int s_wsum[RJ_T / 64], s_m;
    for (int e = threadIdx.x; e < B; e += RJ_T) { hist[e] = 0; cursor[e] = Q.counts[(int64_t)e * gridDim.x + blockIdx.x]; }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * Q.rows_per_wg, i1 = i0 + Q.rows_per_wg < S.n ? i0 + Q.rows_per_wg : S.n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int MAXPER = 8;                // bins per thread of the scan: up to 8192 bins
    const int per = (B + RJ_T - 1) / RJ_T;
    for (int64_t base = i0; base < i1; base += RJ_CH) {
        unsigned long long key[RJ_U];
        int32_t row[RJ_U];
        bool ok[RJ_U];
        int bin[RJ_U], rank[RJ_U];
        rj_keys<KW, NK, SEL>(S, base, i1, key, row, ok);
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            bin[u] = (int)((big_hash(key[u]) >> Q.shift) & (uint64_t)(B - 1));
            rank[u] = ok[u] ? atomicAdd(&hist[bin[u]], 1) : 0;
        }
        __syncthreads();
        int cnt[MAXPER];
        {   // exclusive scan of the chunk's histogram, in place: `per` consecutive bins per thread
            int sum = 0;
#pragma unroll
            for (int q = 0; q < MAXPER; q++) { const int e = threadIdx.x * per + q; cnt[q] = (q < per && e < B) ? hist[e] : 0; sum += cnt[q]; }
            int incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wv; w++) run += s_wsum[w];
#pragma unroll
            for (int q = 0; q < MAXPER; q++) { const int e = threadIdx.x * per + q; if (q < per && e < B) { hist[e] = run; run += cnt[q]; } }
            if (threadIdx.x == RJ_T - 1) s_m = run;   // rows of the chunk that have a key
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            if (!ok[u]) continue;
            const int j = hist[bin[u]] + rank[u];
            s_rec[j] = make_ulonglong2(key[u], (unsigned long long)(unsigned)row[u]);
            s_dst[j] = cursor[bin[u]] + rank[u];
        }
        __syncthreads();
        const int m = s_m;
        for (int j = threadIdx.x; j < m; j += RJ_T) {
            Q.rec[s_dst[j]] = s_rec[j];
        }
#pragma unroll
        for (int q = 0; q < MAXPER; q++) { const int e = threadIdx.x * per + q; if (q < per && e < B) { cursor[e] += cnt[q]; hist[e] = 0; } }
        __syncthreads();
    }
}

// workgroup id -> (partition, slice) such that all slices of a partition have ids of ONE residue mod 8 (workgroups are
// dealt round-robin over the 8 XCDs: they share an L2) and follow each other in that XCD's stream. Needs parts % 8 == 0.
__device__ __forceinline__ void rj_map(int b, int slices, int *p, int *s) {
    const int q = b >> 3, x = b & 7;
    *p = (q / slices) * 8 + x;
    *s = q % slices;
}

// ONE workgroup per bin: its keys into a bucketed open-addressing table in LDS — RJ_BUCKETS buckets of 16 slots {key, row}, a
// 16-byte tag word per bucket (one byte per slot: 0 = empty, else 0x80 | 7 hash bits) — and the image written out whole.
// A probe reads its bucket's tag word (ONE 16-byte read), then only the slots whose tag matches (one, rarely two), and moves
// to the next bucket only when this one is full (16 keys where 9 are expected: ~2 % of the buckets). Tried and measured
// (15 M keys, 60 M probes, probe kernel alone): linear probing over single slots 4.7 ms — a wave takes as many steps as its
// unluckiest lane, 24 at 45 % load; eight lanes reading one 8-slot bucket per probe row (one L2 request per row) 2.1 ms —
// eight times the instructions; this form 1.15 ms, close to what the request path between the CUs and the L2 passes
// (~140 requests/ns on the whole chip, two requests per probe).
// flag |= 1 when a bin holds more keys than 7/8 of the slots (the host keeps the hash tables then)
constexpr int RJ_BUCKETS = RJ_SLOTS / 16;
__device__ __forceinline__ unsigned rj_tag(uint64_t h) { return 0x80u | (unsigned)((h >> 24) & 0x7f); }
__device__ __forceinline__ int rj_bucket(uint64_t h) { return (int)(h & (uint64_t)(RJ_BUCKETS - 1)); }

__global__ __launch_bounds__(RJ_T) void rj_tables_kernel(RjPart Q, int nwg, const int64_t *__restrict__ total, ulonglong2 *__restrict__ slots,
                                                         uint4 *__restrict__ tags, int *__restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rj_lds[];
    ulonglong2 *l_slot = reinterpret_cast<ulonglong2 *>(rj_lds);
    unsigned char *l_tag = reinterpret_cast<unsigned char *>(l_slot + RJ_SLOTS);
    int *l_cnt = reinterpret_cast<int *>(l_tag + RJ_SLOTS);
    const int bin = blockIdx.x;
    for (int e = threadIdx.x; e < RJ_SLOTS; e += RJ_T) l_slot[e] = make_ulonglong2(0ull, 0ull);
    for (int e = threadIdx.x; e < RJ_SLOTS / 4; e += RJ_T) reinterpret_cast<unsigned *>(l_tag)[e] = 0u;
    for (int e = threadIdx.x; e < RJ_BUCKETS; e += RJ_T) l_cnt[e] = 0;
    __syncthreads();
    const int64_t start = Q.counts[(int64_t)bin * nwg];
    const int64_t end = bin + 1 < Q.bins ? (int64_t)Q.counts[(int64_t)(bin + 1) * nwg] : *total;
    if (end - start > RJ_SLOTS - RJ_SLOTS / 8) {   // uniform: the table would have no room to spare
        if (threadIdx.x == 0) atomicOr(flag, 1);
        return;
    }
    for (int64_t t = start + threadIdx.x; t < end; t += RJ_T) {
        const ulonglong2 r = Q.rec[t];
        const uint64_t h = big_hash(r.x);
        int b = rj_bucket(h);
        for (int tries = 0; tries < RJ_BUCKETS; tries++) {
            const int pos = atomicAdd(&l_cnt[b], 1);
            if (pos < 16) {
                l_slot[b * 16 + pos] = r;
                l_tag[b * 16 + pos] = (unsigned char)rj_tag(h);
                break;
            }
            b = (b + 1) & (RJ_BUCKETS - 1);
        }
    }
    __syncthreads();
    ulonglong2 *out = slots + (int64_t)bin * RJ_SLOTS;
    for (int e = threadIdx.x; e < RJ_SLOTS; e += RJ_T) out[e] = l_slot[e];
    uint4 *tout = tags + (int64_t)bin * RJ_BUCKETS;
    for (int e = threadIdx.x; e < RJ_BUCKETS; e += RJ_T) tout[e] = reinterpret_cast<const uint4 *>(l_tag)[e];
}

struct RjProbe {
    int parts, log_sub, slices;
    const int32_t *counts;                   // probe-side offsets [parts][nwg]
    int nwg;
    const int64_t *total;                    // probe rows that have a key
    const ulonglong2 *rec;                   // probe records {key, row} in partition order
    const ulonglong2 *tables;                // slot images [bins][RJ_SLOTS]
    const uint4 *tags;                       // tag words [bins][RJ_BUCKETS]
    int32_t *out_probe, *out_build;
    int64_t cap;
    unsigned long long *n_out;               // pairs (also beyond cap: the caller learns how many there are)
};

// Pairs of one chunk are staged in LDS (one LDS add per wave and probe step reserves their places) and leave with ONE
// reserving add on the global pair counter per workgroup and chunk: adds on one address execute one after the other at
// the memory side (~12 ns each) — a reserving add per wave and step made 60 M probes take 35 ms.
constexpr int RJ_STAGE = 8192;               // pairs staged per chunk (RJ_CH probe rows); beyond it a pair reserves its place by itself

struct RjStage {
    int32_t *p, *b;                          // [RJ_STAGE] each, LDS
    int *cnt;                                // LDS word: pairs staged so far
};

__device__ __forceinline__ void rj_emit(bool match, int prow, int brow, const RjProbe &R, const RjStage &G) {
    const unsigned long long m = __ballot(match);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(G.cnt, __popcll(m));
    base = __shfl(base, leader);
    if (match) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < RJ_STAGE) { G.p[pos] = prow; G.b[pos] = brow; }
        else {   // (a chunk with more than two matches per probe row on average)
            const int64_t g = (int64_t)atomicAdd(R.n_out, 1ull);
            if (g < R.cap) { R.out_probe[g] = prow; R.out_build[g] = brow; }
        }
    }
}

// all rows of the chunk probed: the staged pairs to the output
__device__ __forceinline__ void rj_flush(const RjProbe &R, const RjStage &G) {
    __syncthreads();
    const int cnt = *G.cnt < RJ_STAGE ? *G.cnt : RJ_STAGE;
    __shared__ unsigned long long s_gbase;
    if (threadIdx.x == 0) s_gbase = cnt ? atomicAdd(R.n_out, (unsigned long long)cnt) : 0ull;
    __syncthreads();
    const int64_t gb = (int64_t)s_gbase;
    for (int i = threadIdx.x; i < cnt; i += RJ_T) {
        const int64_t pos = gb + i;
        if (pos < R.cap) { R.out_probe[pos] = G.p[i]; R.out_build[pos] = G.b[i]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) *G.cnt = 0;
    __syncthreads();
}

// RJ_U rows per lane against the table images of their bins. Wave-uniform rounds: every seeking row reads its bucket's tag
// word; rows with candidate slots read them one at a time (reads of a lane's RJ_U rows are issued together); a row whose
// bucket is full goes on to the next bucket.
__device__ __forceinline__ void rj_probe_rows(const unsigned long long (&key)[RJ_U], const int (&prow)[RJ_U], bool (&seek)[RJ_U],
                                              const int (&bin)[RJ_U], const uint64_t (&hh)[RJ_U], const RjProbe &R, const RjStage &G) {
    int b[RJ_U];
    unsigned long long T[RJ_U];
#pragma unroll
    for (int u = 0; u < RJ_U; u++) { b[u] = rj_bucket(hh[u]); T[u] = (unsigned long long)rj_tag(hh[u]) * 0x0101010101010101ull; }
    for (int round = 0; round < RJ_BUCKETS; round++) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < RJ_U; u++) any = any || seek[u];
        if (__ballot(any) == 0ull) break;
        uint4 tg[RJ_U];
#pragma unroll
        for (int u = 0; u < RJ_U; u++) tg[u] = seek[u] ? R.tags[(int64_t)bin[u] * RJ_BUCKETS + b[u]] : make_uint4(0, 0, 0, 0);
        unsigned long long zlo[RJ_U], zhi[RJ_U];   // 0x80 in every byte whose tag equals the row's
        bool full[RJ_U];
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            const unsigned long long lo = (unsigned long long)tg[u].x | ((unsigned long long)tg[u].y << 32);
            const unsigned long long hi = (unsigned long long)tg[u].z | ((unsigned long long)tg[u].w << 32);
            const unsigned long long xl = lo ^ T[u], xh = hi ^ T[u];
            const unsigned long long k7 = 0x7f7f7f7f7f7f7f7full;
            zlo[u] = seek[u] ? ~(((xl & k7) + k7) | xl | k7) : 0ull;
            zhi[u] = seek[u] ? ~(((xh & k7) + k7) | xh | k7) : 0ull;
            full[u] = (lo & hi & 0x8080808080808080ull) == 0x8080808080808080ull;
        }
        for (int cand = 0; cand < 16; cand++) {
            bool more = false;
#pragma unroll
            for (int u = 0; u < RJ_U; u++) more = more || (zlo[u] | zhi[u]) != 0ull;
            if (__ballot(more) == 0ull) break;
            ulonglong2 e[RJ_U];
            bool have[RJ_U];
#pragma unroll
            for (int u = 0; u < RJ_U; u++) {
                have[u] = (zlo[u] | zhi[u]) != 0ull;
                int pos = 0;
                if (zlo[u]) { pos = __ffsll((long long)zlo[u]) - 1; zlo[u] &= zlo[u] - 1; pos >>= 3; }
                else if (zhi[u]) { pos = __ffsll((long long)zhi[u]) - 1; zhi[u] &= zhi[u] - 1; pos = 8 + (pos >> 3); }
                e[u] = have[u] ? R.tables[(int64_t)bin[u] * RJ_SLOTS + b[u] * 16 + pos] : make_ulonglong2(0ull, 0ull);
            }
#pragma unroll
            for (int u = 0; u < RJ_U; u++) rj_emit(have[u] && e[u].x == key[u], prow[u], (int)(unsigned)e[u].y, R, G);
        }
#pragma unroll
        for (int u = 0; u < RJ_U; u++) { seek[u] = seek[u] && full[u]; b[u] = (b[u] + 1) & (RJ_BUCKETS - 1); }
    }
}

#define RJ_STAGE_DECL                                                                     \
    __shared__ int32_t rj_sp[RJ_STAGE], rj_sb[RJ_STAGE];                                  \
    __shared__ int rj_scnt;                                                               \
    const RjStage G{rj_sp, rj_sb, &rj_scnt};                                              \
    if (threadIdx.x == 0) rj_scnt = 0;                                                    \
    __syncthreads()

__global__ __launch_bounds__(RJ_T) void rj_probe_kernel(RjProbe R) {
    RJ_STAGE_DECL;
    int p, sl;
    rj_map((int)blockIdx.x, R.slices, &p, &sl);
    const int64_t start = R.counts[(int64_t)p * R.nwg];
    const int64_t end = p + 1 < R.parts ? (int64_t)R.counts[(int64_t)(p + 1) * R.nwg] : *R.total;
    const int64_t len = end - start, t0 = start + len * sl / R.slices, t1 = start + len * (sl + 1) / R.slices;
    const int subs_mask = (1 << R.log_sub) - 1;
    for (int64_t tb = t0; tb < t1; tb += RJ_CH) {
        unsigned long long key[RJ_U];
        int prow[RJ_U], bin[RJ_U];
        uint64_t hh[RJ_U];
        bool seek[RJ_U];
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            const int64_t t = tb + u * RJ_T + threadIdx.x;
            seek[u] = t < t1;
            const ulonglong2 r = R.rec[seek[u] ? t : t1 - 1];
            key[u] = r.x;
            prow[u] = (int)(unsigned)r.y;
        }
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            hh[u] = big_hash(key[u]);
            bin[u] = (p << R.log_sub) | (int)((hh[u] >> RJ_SHIFT) & (uint64_t)subs_mask);
        }
        rj_probe_rows(key, prow, seek, bin, hh, R, G);
        rj_flush(R, G);
    }
}

// small probe sides: straight from the probe columns, no partitioning (the images are the same tables)
template <int KW, int NK, bool SEL>
__global__ __launch_bounds__(RJ_T) void rj_probe_direct_kernel(RjSide S, RjProbe R, int log_parts) {
    RJ_STAGE_DECL;
    for (int64_t base = (int64_t)blockIdx.x * RJ_CH; base < S.n; base += (int64_t)gridDim.x * RJ_CH) {
        unsigned long long key[RJ_U];
        int32_t prow[RJ_U];
        bool seek[RJ_U];
        int bin[RJ_U];
        uint64_t hh[RJ_U];
        rj_keys<KW, NK, SEL>(S, base, S.n, key, prow, seek);
#pragma unroll
        for (int u = 0; u < RJ_U; u++) {
            hh[u] = big_hash(key[u]);
            bin[u] = (int)((hh[u] >> RJ_SHIFT) & (uint64_t)((1 << (log_parts + R.log_sub)) - 1));
        }
        rj_probe_rows(key, prow, seek, bin, hh, R, G);
        rj_flush(R, G);
    }
}
#undef RJ_STAGE_DECL

}  // namespace ph

struct ph_join {
    ph_ctx *ctx = nullptr;
    ph::JoinSide build{};
    int32_t *sel_copy = nullptr;
    int32_t *head = nullptr, *next = nullptr;
    int64_t cap = 0;
    int64_t count = 0;       // -1 = not fetched from count_dev yet
    int *count_dev = nullptr;
    ph::Bloom bloom{};
    ph::BigNode *nodes = nullptr;   // node table (large build sides): replaces next[]
    int big_kw = 0, big_nk = 0;     // key width / count of the node table's packed key
    int32_t *direct = nullptr;      // direct table (dense integer keys): replaces head[], addressed by key - dlo
    int64_t dlo = 0;
    unsigned long long drange = 0;
    int dkw = 0;
    int dcshift = 0;                // sparse direct tables: bloom.coarse bit = slot >> dcshift (occupied slot groups)
    int64_t count_from_bits = 0;    // gated sorted fill: words of dbits whose set bits are the rows stored
    bool bits_authoritative = false;   // ... and only the occupied slots of `direct` were ever written: every probe tests dbits first
    bool exists_only = false;          // PH_JOIN_EXISTS_ONLY: eflags is the whole table (one byte per key value of the range): mark probes only
    uint8_t *eflags = nullptr;
    unsigned *dbits = nullptr;      // direct tables of <= 8 M slots: one occupancy bit per slot
    ulonglong2 *rj_tables = nullptr;   // radix-partitioned form: 2^(rj_log_parts + rj_log_sub) table images of RJ_SLOTS {key, row} slots
    uint4 *rj_tags = nullptr;          // ... and their tag words (one per bucket of 16 slots)
    int rj_log_parts = 0, rj_log_sub = 0;
};

extern "C" void ph_join_free(ph_join *j) {
    if (!j) return;
    if (j->sel_copy) j->ctx->pool_release(j->sel_copy);
    if (j->head) j->ctx->pool_release(j->head);
    if (j->next) j->ctx->pool_release(j->next);
    if (j->bloom.bits) j->ctx->pool_release(j->bloom.bits);
    if (j->bloom.coarse) j->ctx->pool_release(j->bloom.coarse);
    if (j->count_dev) j->ctx->pool_release(j->count_dev);
    if (j->nodes) j->ctx->pool_release(j->nodes);
    if (j->direct) j->ctx->pool_release(j->direct);
    if (j->dbits) j->ctx->pool_release(j->dbits);
    if (j->eflags) j->ctx->pool_release(j->eflags);
    if (j->rj_tables) j->ctx->pool_release(j->rj_tables);
    if (j->rj_tags) j->ctx->pool_release(j->rj_tags);
    delete j;
}

static int fill_side(ph::JoinSide *S, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n) {
    S->nkeys = nkeys;
    for (int c = 0; c < nkeys; c++) {
        int t = keys[c].type;
        if (t != PH_I32 && t != PH_I64 && t != PH_DATE && t != PH_DEC64 && t != PH_CODE8) {
            ph::set_error("join key %d has type %d (device join keys are integers, dates, decimals and dictionary codes)", c, t);
            return PH_EUNSUPPORTED;
        }
        S->key[c] = {t, keys[c].data, keys[c].validity};
    }
    S->sel = sel;
    S->n = n;
    return PH_OK;
}

// node-table build: count -> scan -> scatter (records = nodes) -> link per 128 KiB head slice
static int build_big(ph_join *j, int kw, int nparts) {
    ph_ctx *ctx = j->ctx;
    const ph::JoinSide &B = j->build;
    const int64_t n = B.n;
    j->big_kw = kw;
    j->big_nk = B.nkeys;
    // 256-thread workgroups, a multiple of the CU count; long runs per (workgroup, partition) keep
    // the scatter's 16-byte stores in whole lines
    // one 1024-thread workgroup per CU: with 1024 partitions a (workgroup, partition) run is ~57 nodes
    // = 0.9 KiB of consecutive 16-byte stores (whole lines except at the two ends)
    const int nwg = (int)std::min<int64_t>(ctx->cu_count, std::max<int64_t>(1, n / 4096));
    const int64_t rows_per_wg = ph::round_up((n + nwg - 1) / nwg, ph::BG_T * ph::PU);
    const int nwg_used = (int)((n + rows_per_wg - 1) / rows_per_wg);
    const int64_t nc = (int64_t)nparts * nwg_used;
    int32_t *counts = nullptr;
    PH_CHECK(ctx->pool_alloc(nc * 4, (void **)&counts));
    if (ctx->pool_alloc(std::max<int64_t>(n, 1) * (int64_t)sizeof(ph::BigNode), (void **)&j->nodes) != PH_OK ||
        ctx->pool_alloc(16, (void **)&j->count_dev) != PH_OK) { ctx->pool_release(counts); return PH_EHIP; }
    const uint64_t mask = (uint64_t)j->cap - 1;
    const size_t hl = (size_t)nparts * 4;
    int rc = PH_OK;
#define PH_BIG_ARGS B.key[0].data, B.key[1].data, B.key[0].validity, B.key[1].validity, B.sel, n, mask, nparts, rows_per_wg
#define PH_BIG_BUILD(KWV, NKV, SELV)                                                                               \
    do {                                                                                                           \
        ph::big_count_kernel<KWV, NKV, SELV><<<nwg_used, ph::BG_T, hl, ctx->stream>>>(PH_BIG_ARGS, counts);             \
        rc = ph::exclusive_scan_i32(ctx, counts, nc, (int64_t *)j->count_dev);                                     \
        ph::big_scatter_kernel<KWV, NKV, SELV><<<nwg_used, ph::BG_T, hl, ctx->stream>>>(PH_BIG_ARGS, counts, j->nodes); \
    } while (0)
    if (B.nkeys == 2) { if (B.sel) PH_BIG_BUILD(4, 2, true); else PH_BIG_BUILD(4, 2, false); }
    else if (kw == 4) { if (B.sel) PH_BIG_BUILD(4, 1, true); else PH_BIG_BUILD(4, 1, false); }
    else { if (B.sel) PH_BIG_BUILD(8, 1, true); else PH_BIG_BUILD(8, 1, false); }
#undef PH_BIG_BUILD
#undef PH_BIG_ARGS
    static std::mutex mu;
    static bool raised[64] = {};
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!raised[ctx->device & 63]) {   // 128 KiB head slice: above the default dynamic LDS limit
            if (hipFuncSetAttribute((const void *)ph::big_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess) rc = PH_EHIP;
            raised[ctx->device & 63] = true;
        }
    }
    if (rc == PH_OK) {
        ph::big_build_kernel<<<nparts, 1024, (size_t)ph::BG_SLICE * 4, ctx->stream>>>(counts, nwg_used, nparts, (const int64_t *)j->count_dev,
                                                                                   j->nodes, j->head);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    ctx->pool_release(counts);
    if (rc != PH_OK) ph::set_error("ph_join_build: node-table build failed");
    return rc;
}


// ---- radix-partitioned join: host side
static void radix_free(ph_join *j) {
    if (j->rj_tables) j->ctx->pool_release(j->rj_tables);
    if (j->rj_tags) j->ctx->pool_release(j->rj_tags);
    j->rj_tables = nullptr;
    j->rj_tags = nullptr;
}

static int rj_raise_lds(const void *f, int bytes) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? PH_OK : PH_EHIP; }

// count -> scan -> staged scatter of one side into `bins` bins; records in Q->pkey / Q->prow (allocated here, n entries),
// offsets in Q->counts ([bins][nwg]), rows with a key in *total_dev
static int rj_partition(ph_ctx *ctx, const ph::JoinSide &B, int kw, int bins, int shift, ph::RjPart *Q, int *nwg_out, int64_t *total_dev) {
    const int64_t n = B.n;
    ph::RjSide S{B.key[0].data, B.key[1].data, B.key[0].validity, B.key[1].validity, B.sel, n};
    const int64_t target = std::max<int64_t>(1, std::min<int64_t>(bins > 1024 ? ctx->cu_count : (int64_t)ctx->cu_count * 2, n / ph::RJ_CH));
    Q->bins = bins;
    Q->shift = shift;
    Q->rows_per_wg = ph::round_up((n + target - 1) / target, ph::RJ_CH);
    const int nwg = (int)((n + Q->rows_per_wg - 1) / Q->rows_per_wg);
    *nwg_out = nwg;
    const int64_t nc = (int64_t)bins * nwg;
    PH_CHECK(ctx->pool_alloc(nc * 4, (void **)&Q->counts));
    PH_CHECK(ctx->pool_alloc(n * 16, (void **)&Q->rec));
    const size_t hl = (size_t)bins * 4;
    const size_t sl = (size_t)bins * 8 + (size_t)ph::RJ_CH * 20;
    if (bins > 8192) { ph::set_error("radix join: more than 8192 bins"); return PH_EUNSUPPORTED; }
    int rc = PH_OK;
#define PH_RJ_PART(KWV, NKV, SELV)                                                                                              \
    do {                                                                                                                        \
        rc = rj_raise_lds((const void *)ph::rj_scatter_kernel<KWV, NKV, SELV>, 144 * 1024);                                     \
        if (rc != PH_OK) break;                                                                                                 \
        ph::rj_count_kernel<KWV, NKV, SELV><<<nwg, ph::RJ_T, hl, ctx->stream>>>(S, *Q);                                         \
        rc = ph::exclusive_scan_i32(ctx, Q->counts, nc, total_dev);                                                             \
        if (rc != PH_OK) break;                                                                                                 \
        ph::rj_scatter_kernel<KWV, NKV, SELV><<<nwg, ph::RJ_T, sl, ctx->stream>>>(S, *Q);                                       \
    } while (0)
    if (B.nkeys == 2) { if (B.sel) PH_RJ_PART(4, 2, true); else PH_RJ_PART(4, 2, false); }
    else if (kw == 4) { if (B.sel) PH_RJ_PART(4, 1, true); else PH_RJ_PART(4, 1, false); }
    else { if (B.sel) PH_RJ_PART(8, 1, true); else PH_RJ_PART(8, 1, false); }
#undef PH_RJ_PART
    if (rc == PH_OK && hipGetLastError() != hipSuccess) rc = PH_EHIP;
    return rc;
}

static void rj_part_free(ph_ctx *ctx, ph::RjPart *Q) {
    if (Q->counts) ctx->pool_release(Q->counts);
    if (Q->rec) ctx->pool_release(Q->rec);
    Q->counts = nullptr; Q->rec = nullptr;
}

// PH_OK with j->rj_tables set, or PH_EUNSUPPORTED (a bin overflows its table: heavy duplicates / skew — the caller builds the node table)
static int build_radix(ph_join *j, int kw) {
    ph_ctx *ctx = j->ctx;
    const ph::JoinSide &B = j->build;
    const int64_t n = B.n;
    // bins of at most ~4.6 k keys (56 % of a table image) on average; a partition = the SUB images one XCD's L2 keeps while its rows probe (2 MiB)
    int log_bins = 6;
    while (((int64_t)1 << log_bins) * (ph::RJ_SLOTS * 9 / 16) < n && log_bins < 13) log_bins++;
    if (((int64_t)1 << log_bins) * (ph::RJ_SLOTS * 9 / 16) < n) return PH_EUNSUPPORTED;   // beyond 37 M build rows: the node table
    const int log_sub = std::min(4, log_bins - 3);
    j->rj_log_sub = log_sub;
    j->rj_log_parts = log_bins - log_sub;
    j->big_kw = kw;
    j->big_nk = B.nkeys;
    ph::RjPart Q{};
    int nwg = 0;
    int64_t *total_dev = nullptr;
    int *flag = nullptr;
    PH_CHECK(ctx->pool_alloc(16, (void **)&j->count_dev));
    PH_CHECK(ctx->pool_alloc(16, (void **)&flag));
    total_dev = (int64_t *)j->count_dev;   // (the low word is the row count ph_join_count reads)
    int rc = hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess ? PH_OK : PH_EHIP;
    if (rc == PH_OK) rc = rj_partition(ctx, B, kw, 1 << log_bins, ph::RJ_SHIFT, &Q, &nwg, total_dev);
    const int64_t tbytes = ((int64_t)ph::RJ_SLOTS << log_bins) * 16;
    if (rc == PH_OK) rc = ctx->pool_alloc(tbytes, (void **)&j->rj_tables);
    if (rc == PH_OK) rc = ctx->pool_alloc(tbytes / 16, (void **)&j->rj_tags);
    if (rc == PH_OK) rc = rj_raise_lds((const void *)ph::rj_tables_kernel, 144 * 1024);
    int f = 0;
    if (rc == PH_OK) {
        ph::rj_tables_kernel<<<1 << log_bins, ph::RJ_T, (size_t)ph::RJ_SLOTS * 17 + (size_t)ph::RJ_BUCKETS * 4, ctx->stream>>>(Q, nwg, total_dev, j->rj_tables, j->rj_tags, flag);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    if (rc == PH_OK) rc = ctx->download_plain(&f, flag, 4);
    rj_part_free(ctx, &Q);
    ctx->pool_release(flag);
    if (rc == PH_OK && f) {
        radix_free(j);
        ctx->pool_release(j->count_dev);
        j->count_dev = nullptr;
        return PH_EUNSUPPORTED;
    }
    if (rc != PH_OK) ph::set_error("ph_join_build: radix build failed");
    return rc;
}

static int radix_probe_inner(ph_join *j, const ph::JoinSide &P, int64_t n, int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    ph_ctx *ctx = j->ctx;
    unsigned long long *cnt = nullptr;
    PH_CHECK(ctx->pool_alloc(16, (void **)&cnt));
    PH_HIP(hipMemsetAsync(cnt, 0, 8, ctx->stream));
    ph::RjProbe R{};
    R.parts = 1 << j->rj_log_parts;
    R.log_sub = j->rj_log_sub;
    R.tables = j->rj_tables;
    R.tags = j->rj_tags;
    R.out_probe = out_probe_dev; R.out_build = out_build_dev; R.cap = cap; R.n_out = cnt;
    int rc = PH_OK;
    const char *pm = getenv("PH_JOIN_RADIX_PART_MIN");   // read per call: the tests probe both ways
    const int64_t part_min = pm ? atoll(pm) : (1ll << 20);
    if (n < part_min) {
        // small probe sides: straight from the probe columns
        ph::RjSide S{P.key[0].data, P.key[1].data, P.key[0].validity, P.key[1].validity, P.sel, n};
        const int grid = (int)std::min<int64_t>((n + ph::RJ_CH - 1) / ph::RJ_CH, (int64_t)ctx->cu_count * 2);
#define PH_RJ_PD(KWV, NKV, SELV) ph::rj_probe_direct_kernel<KWV, NKV, SELV><<<grid, ph::RJ_T, 0, ctx->stream>>>(S, R, j->rj_log_parts)
        if (j->big_nk == 2) { if (P.sel) PH_RJ_PD(4, 2, true); else PH_RJ_PD(4, 2, false); }
        else if (j->big_kw == 4) { if (P.sel) PH_RJ_PD(4, 1, true); else PH_RJ_PD(4, 1, false); }
        else { if (P.sel) PH_RJ_PD(8, 1, true); else PH_RJ_PD(8, 1, false); }
#undef PH_RJ_PD
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    } else {
        ph::RjPart Q{};
        int nwg = 0;
        int64_t *total_dev = nullptr;
        rc = ctx->pool_alloc(16, (void **)&total_dev);
        if (rc == PH_OK) rc = rj_partition(ctx, P, j->big_kw, R.parts, ph::RJ_SHIFT + j->rj_log_sub, &Q, &nwg, total_dev);
        if (rc == PH_OK) {
            R.counts = Q.counts; R.nwg = nwg; R.total = total_dev; R.rec = Q.rec;
            // as many slices as keep ONE partition's workgroups on an XCD at a time (two 1024-thread workgroups per CU x 32 CUs)
            int slices = 64;
            while (slices > 1 && n / ((int64_t)R.parts * slices) < 1024) slices /= 2;
            R.slices = slices;
            ph::rj_probe_kernel<<<R.parts * slices, ph::RJ_T, 0, ctx->stream>>>(R);
            if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
        }
        rj_part_free(ctx, &Q);
        if (total_dev) ctx->pool_release(total_dev);
    }
    if (rc == PH_OK) rc = ctx->download_count(n_out, cnt, cap, "ph_join_probe_inner");
    ctx->pool_release(cnt);
    return rc;
}

// ---- direct table (dense integer keys): host side
#define PH_DIRECT_KS(KERNEL, GRID, THREADS, ...)                                                            \
    do {                                                                                                    \
        if (kw == 4) { if (B.sel) KERNEL<4, true><<<GRID, THREADS, 0, ctx->stream>>>(__VA_ARGS__); else KERNEL<4, false><<<GRID, THREADS, 0, ctx->stream>>>(__VA_ARGS__); } \
        else { if (B.sel) KERNEL<8, true><<<GRID, THREADS, 0, ctx->stream>>>(__VA_ARGS__); else KERNEL<8, false><<<GRID, THREADS, 0, ctx->stream>>>(__VA_ARGS__); }         \
    } while (0)

// workgroups of the sorted fill resident per CU: its chunks are assigned statically (chunk c, c + grid, ...),
// so a workgroup that had to wait for a free CU would start its share when the others are done — 8 per CU
// were launched where the static LDS (20.5 KB) lets 7 in, and the kernel took twice its time
static int sorted_fill_occupancy(int kw, bool gated) {
    int occ = 0;
    const void *f = kw == 4 ? (gated ? (const void *)ph::direct_sorted_fill_kernel<4, 3> : (const void *)ph::direct_sorted_fill_kernel<4, 0>)
                            : (gated ? (const void *)ph::direct_sorted_fill_kernel<8, 3> : (const void *)ph::direct_sorted_fill_kernel<8, 0>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, f, 256, 0) != hipSuccess || occ < 1) occ = 4;
    return std::min(occ, 8);
}

namespace ph {
// min / max of the build keys (NULL keys and rows outside the selection skipped): out = {min, max}, preset to {INT64_MAX, INT64_MIN}
__global__ void join_key_range_init_kernel(long long *out) { out[0] = INT64_MAX; out[1] = INT64_MIN; }

template <int KW>
__global__ __launch_bounds__(256) void join_key_range_kernel(const void *__restrict__ kcol, const uint8_t *__restrict__ valid,
                                                             const int32_t *__restrict__ sel, int64_t n, long long *__restrict__ out) {
    long long lo = INT64_MAX, hi = INT64_MIN;
    constexpr int U = 8;   // keys of a thread in flight together (two workgroups per CU: little else hides the latency)
    for (int64_t base = (int64_t)blockIdx.x * 256 * U; base < n; base += (int64_t)gridDim.x * 256 * U) {
        long long x[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            ok[u] = i < n;
            const int64_t r = ok[u] ? (sel ? (int64_t)sel[i] : i) : 0;
            ok[u] = ok[u] && (!valid || bit_valid(valid, r));
            x[u] = KW == 4 ? (long long)((const int32_t *)kcol)[r] : ((const long long *)kcol)[r];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            lo = ok[u] && x[u] < lo ? x[u] : lo;
            hi = ok[u] && x[u] > hi ? x[u] : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    // ONE pair of atomics per workgroup: atomics on one address execute one after the other at the memory side
    // (~12 ns each; a pair per wave of a 2048-workgroup grid was 0.2 ms for a 25 us pass)
    __shared__ long long s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; }
        if (lo <= hi) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
    }
}
}  // namespace ph

namespace ph {
// Existence-only table (SEMI / ANTI / mark joins: which build rows hold a key never matters, nor how many): one BYTE per key value of the
// range. The general direct build of a key column with duplicates is slot scatter + occupancy + verify + duplicate chains — 0.9 ms for
// 15 M o_custkey values, 1.2 ms for 38 M l_orderkey values; the flags are one pass of plain byte stores.
template <int KW, bool SEL>
__global__ __launch_bounds__(256) void direct_bits_kernel(const void *__restrict__ key, const uint8_t *__restrict__ valid, const int32_t *__restrict__ sel,
                                                         int64_t n, long long lo, unsigned long long range, uint8_t *__restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = SEL ? (int64_t)sel[i] : i;
        if (valid && !bit_valid(valid, r)) continue;
        unsigned long long off;
        if (!direct_slot<KW>((long long)load_kw<KW>(key, r), lo, range, &off)) continue;
        flags[off] = 1;   // a plain, idempotent byte store: scattered atomics run at the memory side (~25 G/s, and same-word ORs of a clustered
                          // key column one after the other: the bit form took 1.4 ms for 38 M l_orderkey values), byte stores of neighbouring keys coalesce
    }
}

template <int KW, bool SELP>
__global__ __launch_bounds__(256) void direct_mark_bits_kernel(const void *__restrict__ pkey, const uint8_t *__restrict__ pvalid, const int32_t *__restrict__ psel,
                                                              int64_t n, long long lo, unsigned long long range, const uint8_t *__restrict__ flags,
                                                              uint8_t *__restrict__ found) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = SELP ? (int64_t)psel[i] : i;
        unsigned long long off = 0;
        bool ok = !pvalid || bit_valid(pvalid, r);
        ok = direct_slot<KW>((long long)load_kw<KW>(pkey, r), lo, range, &off) && ok;
        found[i] = ok && flags[off] ? 1 : 0;
    }
}
}  // namespace ph

static int build_exists(ph_join *j, int kw, int64_t lo, int64_t range) {
    ph_ctx *ctx = j->ctx;
    const int64_t n = j->build.n;
    if (kw == 4) {   // (as build_direct: the slot arithmetic of 4-byte keys is 32-bit)
        const int64_t hi = std::min<int64_t>(lo + range - 1, INT32_MAX);
        lo = std::min<int64_t>(std::max<int64_t>(lo, INT32_MIN), INT32_MAX);
        range = hi >= lo ? hi - lo + 1 : 1;
    }
    const int64_t bytes = ph::round_up(range, 256);
    j->dkw = kw;
    j->dlo = lo;
    j->drange = (unsigned long long)range;
    if (ctx->pool_alloc(bytes, (void **)&j->eflags) != PH_OK || ctx->pool_alloc(16, (void **)&j->count_dev) != PH_OK) { ph::set_error("ph_join_build: allocation failed"); return PH_EHIP; }
    PH_HIP(hipMemsetAsync(j->eflags, 0, (size_t)bytes, ctx->stream));
    PH_HIP(hipMemsetAsync(j->count_dev, 0, 16, ctx->stream));
    if (n > 0) {
        const ph::JoinSide &B = j->build;
        const int grid = (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16);
        if (kw == 4) { if (B.sel) ph::direct_bits_kernel<4, true><<<grid, 256, 0, ctx->stream>>>(B.key[0].data, B.key[0].validity, B.sel, n, (long long)lo, j->drange, j->eflags);
                       else ph::direct_bits_kernel<4, false><<<grid, 256, 0, ctx->stream>>>(B.key[0].data, B.key[0].validity, nullptr, n, (long long)lo, j->drange, j->eflags); }
        else { if (B.sel) ph::direct_bits_kernel<8, true><<<grid, 256, 0, ctx->stream>>>(B.key[0].data, B.key[0].validity, B.sel, n, (long long)lo, j->drange, j->eflags);
               else ph::direct_bits_kernel<8, false><<<grid, 256, 0, ctx->stream>>>(B.key[0].data, B.key[0].validity, nullptr, n, (long long)lo, j->drange, j->eflags); }
        PH_HIP(hipGetLastError());
    }
    j->build.sel = nullptr;   // (nothing refers to build rows afterwards)
    j->exists_only = true;
    j->count = n;             // (rows offered; the table itself knows key values, not rows)
    return PH_OK;
}

static int build_direct(ph_join *j, int kw, int64_t lo, int64_t range, const ph::RangePred &where_in, bool declared_sorted_unique) {
    ph_ctx *ctx = j->ctx;
    const int64_t n = j->build.n;
    // A SMALL dense table over keys declared sorted and unique (a dimension's primary key: supplier, 100 k rows)
    // goes through the gated sorted fill too, with a gate that always passes (the key column against the whole
    // integer range): one kernel without atomics instead of direct_small_kernel's atomic exchange per row
    // (device-scope atomics to random lines run at ~10 G/s: 15 us for 100 k rows, 4 us for the fill).
    ph::RangePred where = where_in;
    if (declared_sorted_unique && where.kind == 0 && n >= 4096 && n <= (256 << 10) && range <= 8 * n && !j->build.sel && !j->build.key[0].validity) {
        where.kind = kw == 4 ? 1 : 2;
        where.data = j->build.key[0].data;
        where.validity = nullptr;
        where.lo = INT64_MIN;
        where.hi = INT64_MAX;
    }
    if (kw == 4) {   // a 4-byte key cannot lie outside the int32 domain: the slot arithmetic of 4-byte tables is 32-bit (direct_slot)
        const int64_t hi = std::min<int64_t>(lo + range - 1, INT32_MAX);
        lo = std::min<int64_t>(std::max<int64_t>(lo, INT32_MIN), INT32_MAX);
        range = hi >= lo ? hi - lo + 1 : 1;   // (an empty intersection: one slot no stated-range key can claim)
    }
    const int64_t cap4 = ph::round_up(range, 4);
    j->dkw = kw;
    j->dlo = lo;
    j->drange = (unsigned long long)range;
    if (ctx->pool_alloc(cap4 * 4, (void **)&j->direct) != PH_OK || ctx->pool_alloc(std::max<int64_t>(n, 1) * 4, (void **)&j->next) != PH_OK ||
        ctx->pool_alloc(16, (void **)&j->count_dev) != PH_OK) { ph::set_error("ph_join_build_range: allocation failed"); return PH_EHIP; }
    if (j->build.sel && n > 0) {   // own copy: the table outlives the caller's selection buffer
        PH_CHECK(ctx->pool_alloc(n * 4, (void **)&j->sel_copy));
        PH_HIP(hipMemcpyAsync(j->sel_copy, j->build.sel, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        j->build.sel = j->sel_copy;
    }
    const ph::JoinSide &B = j->build;
    // sparse small build side (the range check in join_build_impl let it in because the range itself is
    // small): bitmap of occupied slot groups for the LDS filter of big selective probes
    if (n <= (256 << 10) && range > 8 * n) {
        PH_CHECK(ctx->pool_alloc(ph::CO_WORDS * 4, (void **)&j->bloom.coarse));
        while (((range - 1) >> j->dcshift) >= (int64_t)ph::CO_WORDS * 32) j->dcshift++;
    }
    // occupancy bitmap for the candidate pass of inner probes: 1 MiB at most, so that it stays in L2 while
    // the slot array (32 x larger) is only read for rows that match
    int64_t dwords = 0;
    const char *sfe0 = getenv("PH_JOIN_SORTED_FILL");
    // a filtered build over keys declared sorted and unique: the gated sorted fill writes the bitmap in its one
    // pass, so the bitmap may be larger (clustered probes stream it; 128 M slots = 16 MiB)
    const bool gated_fill = declared_sorted_unique && where.kind != 0 && (where.kind != 3 || (reinterpret_cast<uintptr_t>(where.data) & 3) == 0) && !(sfe0 && atoi(sfe0) == 0) && (n > (256 << 10) || (n >= 4096 && range <= 8 * n)) && !j->build.sel &&
                            !j->build.key[0].validity && range <= (128ll << 20) && lo <= INT64_MAX - range - 1;
    if (range <= (8ll << 20) || gated_fill) {
        dwords = ph::round_up(cap4, 128) / 32;   // the occupied kernel writes whole words up to cap4
        PH_CHECK(ctx->pool_alloc(dwords * 4, (void **)&j->dbits));
    }
    // sorted keys (verified on the device): one streaming fill instead of initialisation + scatter + count.
    // Plain shapes only (no selection, NULLs or pushed-down filter; no occupancy bitmap to derive).
    const char *sfe = getenv("PH_JOIN_SORTED_FILL");   // read per call: the test builds both ways in one process
    const bool no_sorted = sfe && atoi(sfe) == 0;
    const bool try_sorted = gated_fill || (!no_sorted && n > (256 << 10) && !B.sel && !B.key[0].validity && where.kind == 0 && !j->dbits &&
                            lo <= INT64_MAX - range - 1);   // the kernel uses lo + range as the "key after the last"
    // (the sorted fill writes every slot itself: only the counters are cleared here)
    ph::join_init_kernel<<<try_sorted && !j->dbits ? 1 : ctx->cu_count * 4, 256, 0, ctx->stream>>>(try_sorted ? nullptr : j->direct, cap4, j->dbits, dwords,
                                                                                     j->bloom.coarse, j->count_dev);
    if (n > 0) {
        const int grid = (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 8);
        const ph::DirectSrc S{B.key[0].data, B.key[0].validity, B.sel, where.kind, where.data, where.lo, where.hi};
        if (n <= (256 << 10) && !gated_fill) {   // (a small declared-sorted table takes the gated fill below: its slots are not initialised)
            const int grids = (int)std::min<int64_t>((n + ph::DT - 1) / ph::DT, (int64_t)ctx->cu_count);
            PH_DIRECT_KS(ph::direct_small_kernel, grids, ph::DT, S, n, (long long)lo, j->drange, j->direct, j->next, j->count_dev,
                         j->bloom.coarse, j->dcshift, j->dbits);
            PH_HIP(hipGetLastError());
            j->count = -1;
            return PH_OK;
        }
        const int *gate = nullptr;
        int *partials = nullptr;
        if (try_sorted && declared_sorted_unique) {
            // keys the caller's statistics declare sorted and unique: the fill alone (it verifies the claim; a
            // violation is a deferred PH_ECONSTRAINT of the ctx and the caller builds again without the claim).
            // The six launches of the general passes, ~4.6 us each although they would leave at once, are not made.
            int *words = nullptr;
            PH_CHECK(ctx->deferred_words(&words));
            const int frows = where.kind != 0 ? ph::SF_GATED_ROWS : ph::SF_ROWS;
            const int gridf = (int)std::min<int64_t>((n + frows - 1) / frows, (int64_t)ctx->cu_count * sorted_fill_occupancy(kw, where.kind != 0));
            if (where.kind != 0) {   // the build child's Filter rides along; the fill also writes the occupancy bitmap
                unsigned *fill_bits = j->dbits;
                j->bits_authoritative = true;
#define PH_SF_GATED(KWV, WKV) ph::direct_sorted_fill_kernel<KWV, WKV><<<gridf, 256, 0, ctx->stream>>>(B.key[0].data, n, (long long)lo, j->drange, cap4, j->direct, j->count_dev, nullptr, words + 3, where.data, where.lo, where.hi, fill_bits)
                if (kw == 4) { if (where.kind == 1) PH_SF_GATED(4, 1); else if (where.kind == 2) PH_SF_GATED(4, 2); else PH_SF_GATED(4, 3); }
                else { if (where.kind == 1) PH_SF_GATED(8, 1); else if (where.kind == 2) PH_SF_GATED(8, 2); else PH_SF_GATED(8, 3); }
#undef PH_SF_GATED
                PH_HIP(hipGetLastError());
                ctx->deferred_pending = true;
                j->count = -1;   // the rows that pass = the bitmap's set bits, counted when ph_join_count asks
                j->count_from_bits = dwords;
                return PH_OK;
            }
            if (kw == 4) ph::direct_sorted_fill_kernel<4, 0><<<gridf, 256, 0, ctx->stream>>>(B.key[0].data, n, (long long)lo, j->drange, cap4, j->direct, j->count_dev, nullptr, words + 3, nullptr, 0, 0, nullptr);
            else ph::direct_sorted_fill_kernel<8, 0><<<gridf, 256, 0, ctx->stream>>>(B.key[0].data, n, (long long)lo, j->drange, cap4, j->direct, j->count_dev, nullptr, words + 3, nullptr, 0, 0, nullptr);
            PH_HIP(hipGetLastError());
            ctx->deferred_pending = true;
            j->count = n;   // every row is stored when the claim holds (count[0] == count[1] == 0 on the device: no chains)
            return PH_OK;
        }
        if (try_sorted) {
            const int gridf = (int)std::min<int64_t>((n + ph::SF_ROWS - 1) / ph::SF_ROWS, (int64_t)ctx->cu_count * sorted_fill_occupancy(kw, false));
            PH_CHECK(ctx->pool_alloc((int64_t)gridf * 8, (void **)&partials));
            if (kw == 4) ph::direct_sorted_fill_kernel<4, 0><<<gridf, 256, 0, ctx->stream>>>(B.key[0].data, n, (long long)lo, j->drange, cap4, j->direct, j->count_dev, partials, nullptr, nullptr, 0, 0, nullptr);
            else ph::direct_sorted_fill_kernel<8, 0><<<gridf, 256, 0, ctx->stream>>>(B.key[0].data, n, (long long)lo, j->drange, cap4, j->direct, j->count_dev, partials, nullptr, nullptr, 0, 0, nullptr);
            ph::direct_recount_kernel<<<1, 256, 0, ctx->stream>>>(j->count_dev, partials, gridf);
            ph::direct_refill_kernel<<<ctx->cu_count * 4, 256, 0, ctx->stream>>>(j->direct, cap4, j->count_dev);
            ctx->pool_release(partials);   // stream-ordered reuse
            gate = j->count_dev + 3;
        }
        // one 1024-thread workgroup per CU for the two passes that end in a counter update
        const int gridc = (int)std::min<int64_t>((n + ph::DT * 4 - 1) / (ph::DT * 4), (int64_t)ctx->cu_count);
        const int grido = (int)std::min<int64_t>((cap4 / 4 + ph::DT - 1) / ph::DT, (int64_t)ctx->cu_count);
        PH_DIRECT_KS(ph::direct_scatter_kernel, gridc, ph::DT, S, n, (long long)lo, j->drange, j->direct, j->count_dev, gate);
        ph::direct_occupied_kernel<<<grido, ph::DT, 0, ctx->stream>>>(j->direct, cap4, j->count_dev, j->dbits, gate);
        PH_DIRECT_KS(ph::direct_verify_kernel, grid, 256, S, n, (long long)lo, j->drange, j->direct, j->next, j->count_dev);
        const int grid1 = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
        PH_DIRECT_KS(ph::direct_dups_kernel, grid1, 256, S, n, (long long)lo, j->drange, j->direct, j->next, j->count_dev);
    }
    PH_HIP(hipGetLastError());
    j->count = n == 0 ? 0 : -1;
    return PH_OK;
}
#undef PH_DIRECT_KS

namespace ph {
// The mark probe with the table's occupancy bitmap STAGED IN LDS. Every probe row that passes the filter reads
// one bitmap word at a random position; from L2 that is one request per row, and the L1 -> L2 request path
// (~140 k requests / us on the whole device, PMC: TCP pending stalls half of the kernel) bounds the plain kernel
// at 48 us for Q3's 15 M orders rows where the columns stream in 20. One 1024-thread workgroup per CU keeps a
// 1 Mbit image of the bitmap in LDS: the bitmap itself when it has at most 2^20 bits, else its words FOLDED
// (image word w = OR of bitmap words w, w + 32768, ...): a clear image bit answers "absent", a set one is
// confirmed in the real bitmap (L2) only when the image is folded. Waves work alone on 512-row sub-blocks.
template <int KW, int WK>
__global__ __launch_bounds__(1024) void direct_mark_where_lds_kernel(const void *__restrict__ keycol, int64_t n, long long lo, unsigned long long range,
                                                                     const unsigned *__restrict__ dbits, int64_t dwords,
                                                                     const void *__restrict__ wdata, long long wlo, long long whi,
                                                                     uint8_t *__restrict__ found) {
    extern __shared__ unsigned mw_img[];   // CO_WORDS words
    constexpr int R = 16 / KW, G = 8 / R;
    const bool folded = dwords > CO_WORDS;
    {   // 32 image words per thread, their reads issued together (a word-by-word loop is 64 dependent L2 latencies: 45 us)
        unsigned v[CO_WORDS / 1024];
#pragma unroll
        for (int q = 0; q < CO_WORDS / 1024; q++) v[q] = 0;
        for (int64_t fold = 0; fold < dwords; fold += CO_WORDS) {
#pragma unroll
            for (int q = 0; q < CO_WORDS / 1024; q++) {
                const int64_t x = fold + q * 1024 + threadIdx.x;
                v[q] |= x < dwords ? dbits[x] : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < CO_WORDS / 1024; q++) mw_img[q * 1024 + threadIdx.x] = v[q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nsub = (n + 511) / 512;
    // two sub-blocks of a wave in flight: their key / filter reads are issued together, then their bitmap tests,
    // then their confirmations (one wave per SIMD x 4 has little else to hide a memory round trip behind)
    constexpr int NB = 2;   // (four: no faster, 40 vs 39 us)
    const int64_t stride = (int64_t)gridDim.x * 16;
    for (int64_t sb0 = (int64_t)blockIdx.x * 16 + wave; sb0 < nsub; sb0 += stride * NB) {
        long long k[NB][8], k2[8];
        bool ok[NB][8];
        unsigned long long off[NB][8];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int64_t sb = sb0 + u * stride;
            const bool have = sb < nsub;
            const int64_t row0 = have ? sb * 512 : 0;
            dc_block_keys<KW, WK, 1>(keycol, nullptr, wdata, wlo, whi, row0, lane, have && row0 + 512 <= n, have, n, k[u], k2, ok[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; u++)
#pragma unroll
            for (int s = 0; s < 8; s++) {
                ok[u][s] = direct_slot<KW>(k[u][s], lo, range, &off[u][s]) && ok[u][s];
                const unsigned b = ok[u][s] ? (unsigned)off[u][s] & (CO_WORDS * 32 - 1) : 0u;
                ok[u][s] = ok[u][s] && ((mw_img[b >> 5] >> (b & 31)) & 1u);
            }
        if (folded) {   // confirm the survivors in the real bitmap
            unsigned wd[NB][8];
#pragma unroll
            for (int u = 0; u < NB; u++)
#pragma unroll
                for (int s = 0; s < 8; s++) wd[u][s] = dbits[ok[u][s] ? off[u][s] >> 5 : 0];
#pragma unroll
            for (int u = 0; u < NB; u++)
#pragma unroll
                for (int s = 0; s < 8; s++) ok[u][s] = ok[u][s] && ((wd[u][s] >> (off[u][s] & 31)) & 1u);
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int64_t sb = sb0 + u * stride;
            if (sb >= nsub) break;   // wave-uniform
            const int64_t row0 = sb * 512;
            const bool full = row0 + 512 <= n;
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int64_t row = row0 + (int64_t)(g * 64 + lane) * R;
                if (full) {
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < R; e++) w |= (ok[u][g * R + e] ? 1u : 0u) << (8 * e);
                    if (R == 4) *reinterpret_cast<unsigned *>(found + row) = w;
                    else *reinterpret_cast<unsigned short *>(found + row) = (unsigned short)w;
                } else {
#pragma unroll
                    for (int e = 0; e < R; e++)
                        if (row + e < n) found[row + e] = ok[u][g * R + e] ? 1 : 0;
                }
            }
        }
    }
}
}  // namespace ph

static bool direct_probe_ok(const ph_join *j, const ph::JoinSide &P) {
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
    return P.nkeys == 1 && width(P.key[0].type) == j->dkw;
}

template <int MODE>
static void launch_direct_probe(ph_join *j, const ph::JoinSide &P, int64_t n, int grid, int32_t *out, uint8_t *found, int *stats) {
    hipStream_t st = j->ctx->stream;
    const int32_t *bsel = j->build.sel;
    // the occupancy bitmap in front of the slot array: required when only occupied slots were ever written (gated sorted fill), and a FILTER when
    // the table is sparse — the bitmap (<= 1 MiB) stays in L2, the slot array (32 x larger) is then read for the few rows that can match (a SEMI
    // mark pass of 60 M keys against 2 000 parts of a 2 M-key range: 615 -> ~100 us); a probe that mostly hits would only pay the extra read
    const unsigned *hint = j->dbits && (j->bits_authoritative || (int64_t)j->build.n * 4 < (int64_t)j->drange) ? j->dbits : nullptr;
#define PH_DP_ARGS P.key[0].data, P.key[0].validity, P.sel, n, (long long)j->dlo, j->drange, j->direct, j->next, bsel, j->count_dev, (int32_t)j->build.n, out, found, stats, hint
#define PH_DP_LAUNCH(KWV)                                                                                                      \
    do {                                                                                                                       \
        if (P.sel && bsel) ph::direct_probe_kernel<KWV, true, true, MODE><<<grid, 256, 0, st>>>(PH_DP_ARGS);                   \
        else if (P.sel) ph::direct_probe_kernel<KWV, true, false, MODE><<<grid, 256, 0, st>>>(PH_DP_ARGS);                     \
        else if (bsel) ph::direct_probe_kernel<KWV, false, true, MODE><<<grid, 256, 0, st>>>(PH_DP_ARGS);                      \
        else ph::direct_probe_kernel<KWV, false, false, MODE><<<grid, 256, 0, st>>>(PH_DP_ARGS);                               \
    } while (0)
    if (j->dkw == 4) PH_DP_LAUNCH(4); else PH_DP_LAUNCH(8);
#undef PH_DP_LAUNCH
#undef PH_DP_ARGS
}

template <int KW, int WK>
static void launch_direct_cand(ph_join *j, const ph::JoinSide &P, int64_t n, int nb, const ph::RangePred &w, const uint8_t *bflags, uint16_t *cand,
                               int32_t *cmatch, uint16_t *ccnt, int32_t *ccount, int32_t *counts) {
    ph::DirectCand D{P.key[0].data, P.key[0].validity, P.sel, n, (long long)j->dlo, j->drange, j->direct, j->next, j->count_dev,
                     w.data, w.lo, w.hi, cand, cmatch, ccnt, ccount, counts, j->dcshift, j->dbits, bflags, (int32_t)j->build.n};
    hipStream_t st = j->ctx->stream;
    auto aligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = !P.sel && !P.key[0].validity && aligned(P.key[0].data) && (WK == 0 || aligned(w.data));
    // the LDS bitmap of occupied slot groups already rejects most probes: a second filter stage (one more
    // dependent L2 read per survivor) made the kernel slower (Q9: 119 -> 151 us)
    if (j->bloom.coarse && nb >= 64 && !j->bits_authoritative) D.dbits = nullptr;
    if (vec && j->bloom.coarse && nb >= 64 && !j->bits_authoritative) {
        const size_t lds = (size_t)ph::CO_WORDS * 4;
        const int grid = std::min((nb + 15) / 16, j->ctx->cu_count);   // one block per wave and step
        (void)hipFuncSetAttribute((const void *)ph::direct_cand_coarse_vec_kernel<KW, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ph::direct_cand_coarse_vec_kernel<KW, WK><<<grid, 1024, lds, st>>>(D, j->bloom.coarse, (int64_t)nb);
        return;
    }
    if (vec) { ph::direct_cand_vec_kernel<KW, WK><<<nb, 256, 0, st>>>(D); return; }
    if (j->bloom.coarse && nb >= 64 && !j->bits_authoritative) {   // sparse table: occupied-group bitmap in LDS, one 1024-thread workgroup per CU
        const size_t lds = (size_t)ph::CO_WORDS * 4 + 4 * ph::JP_ROUNDS * 4 * sizeof(int) + 4 * 4 * sizeof(int);
        const int grid = std::min((nb + 3) / 4, j->ctx->cu_count);
        if (P.sel) {
            (void)hipFuncSetAttribute((const void *)ph::direct_cand_coarse_kernel<KW, WK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            ph::direct_cand_coarse_kernel<KW, WK, true><<<grid, 1024, lds, st>>>(D, j->bloom.coarse, (int64_t)nb);
        } else {
            (void)hipFuncSetAttribute((const void *)ph::direct_cand_coarse_kernel<KW, WK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            ph::direct_cand_coarse_kernel<KW, WK, false><<<grid, 1024, lds, st>>>(D, j->bloom.coarse, (int64_t)nb);
        }
        return;
    }
    if (P.sel) ph::direct_cand_kernel<KW, WK, true><<<nb, 256, 0, st>>>(D);
    else ph::direct_cand_kernel<KW, WK, false><<<nb, 256, 0, st>>>(D);
}

static int direct_probe_inner(ph_join *j, const ph::JoinSide &P, int64_t n, const ph::RangePred &where, const uint8_t *bflags,
                              int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    ph_ctx *ctx = j->ctx;
    const int64_t nb = (n + ph::JP_CHUNK - 1) / ph::JP_CHUNK;
    const int64_t o_ccount = ph::round_up(nb * 4, 8) + 64;
    const int64_t o_cand = ph::round_up(o_ccount + nb * 4, 8), o_ccnt = o_cand + nb * ph::JP_CHUNK * 2;
    const int64_t o_cmatch = o_ccnt + nb * ph::JP_CHUNK * 2;
    PH_CHECK(ctx->ensure_scratch(o_cmatch + nb * ph::JP_CHUNK * 4));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(nb * 4, 8));
    int32_t *ccount = (int32_t *)((char *)ctx->scratch + o_ccount);
    uint16_t *cand = (uint16_t *)((char *)ctx->scratch + o_cand), *ccnt = (uint16_t *)((char *)ctx->scratch + o_ccnt);
    int32_t *cmatch = (int32_t *)((char *)ctx->scratch + o_cmatch);
#define PH_DC_K(KWV)                                                                                                 \
    switch (where.kind) {                                                                                            \
    case 0: launch_direct_cand<KWV, 0>(j, P, n, (int)nb, where, bflags, cand, cmatch, ccnt, ccount, counts); break;          \
    case 1: launch_direct_cand<KWV, 1>(j, P, n, (int)nb, where, bflags, cand, cmatch, ccnt, ccount, counts); break;          \
    case 2: launch_direct_cand<KWV, 2>(j, P, n, (int)nb, where, bflags, cand, cmatch, ccnt, ccount, counts); break;          \
    default: launch_direct_cand<KWV, 3>(j, P, n, (int)nb, where, bflags, cand, cmatch, ccnt, ccount, counts); break;         \
    }
    if (j->dkw == 4) { PH_DC_K(4) } else { PH_DC_K(8) }
#undef PH_DC_K
    PH_HIP(hipGetLastError());
    ph::ScanPublish pub;   // the pair count travels with the scan: the host has it while the pairs are still being written
    PH_CHECK(ctx->arm_count(&pub));
    PH_CHECK(ph::exclusive_scan_i32(ctx, counts, nb, total, pub.seq ? &pub : nullptr));
    const int wave_grid = (int)std::min<int64_t>((nb + 3) / 4, (int64_t)ctx->cu_count * 8);
    const int32_t *bsel = j->build.sel;
#define PH_DE_ARGS P.sel, j->next, bsel, cand, cmatch, ccnt, ccount, counts, j->count_dev, bflags, nb, cap, out_probe_dev, out_build_dev
    if (P.sel && bsel) ph::direct_emit_kernel<true, true><<<wave_grid, 256, 0, ctx->stream>>>(PH_DE_ARGS);
    else if (P.sel) ph::direct_emit_kernel<true, false><<<wave_grid, 256, 0, ctx->stream>>>(PH_DE_ARGS);
    else if (bsel) ph::direct_emit_kernel<false, true><<<wave_grid, 256, 0, ctx->stream>>>(PH_DE_ARGS);
    else ph::direct_emit_kernel<false, false><<<wave_grid, 256, 0, ctx->stream>>>(PH_DE_ARGS);
#undef PH_DE_ARGS
    PH_HIP(hipGetLastError());
    return ctx->count_back(pub, n_out, total, cap, "ph_join_probe_inner");
}

// probe-side shape check of a node table: same packing as the build side, no other key shape
static bool big_probe_ok(const ph_join *j, const ph::JoinSide &P) {
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
    if (P.nkeys != j->big_nk) return false;
    for (int c = 0; c < P.nkeys; c++) if (width(P.key[c].type) != j->big_kw) return false;
    return true;
}

template <int MODE>
static void launch_big_probe(ph_join *j, const ph::JoinSide &P, int64_t n, int grid, int32_t *out, uint16_t *ccnt, int32_t *block_counts,
                             uint8_t *found, int *stats) {
    hipStream_t st = j->ctx->stream;
    const uint64_t mask = (uint64_t)j->cap - 1;
#define PH_BP_ARGS P.key[0].data, P.key[1].data, P.key[0].validity, P.key[1].validity, P.sel, n, j->head, mask, j->nodes, out, ccnt, block_counts, found, stats
    if (j->big_nk == 2) { if (P.sel) ph::big_probe_kernel<4, 2, true, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); else ph::big_probe_kernel<4, 2, false, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); }
    else if (j->big_kw == 4) { if (P.sel) ph::big_probe_kernel<4, 1, true, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); else ph::big_probe_kernel<4, 1, false, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); }
    else { if (P.sel) ph::big_probe_kernel<8, 1, true, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); else ph::big_probe_kernel<8, 1, false, MODE><<<grid, 256, 0, st>>>(PH_BP_ARGS); }
#undef PH_BP_ARGS
}

static int big_probe_inner(ph_join *j, const ph::JoinSide &P, int64_t n, int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap,
                           int64_t *n_out) {
    ph_ctx *ctx = j->ctx;
    const int64_t nb = (n + ph::JP_CHUNK - 1) / ph::JP_CHUNK;
    const int64_t o_total = ph::round_up(nb * 4, 8), o_ccnt = o_total + 64, o_cmatch = ph::round_up(o_ccnt + nb * ph::JP_CHUNK * 2, 8);
    PH_CHECK(ctx->ensure_scratch(o_cmatch + nb * ph::JP_CHUNK * 4));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + o_total);
    uint16_t *ccnt = (uint16_t *)((char *)ctx->scratch + o_ccnt);
    int32_t *cmatch = (int32_t *)((char *)ctx->scratch + o_cmatch);
    launch_big_probe<1>(j, P, n, (int)std::min<int64_t>(nb, (int64_t)ctx->cu_count * 16), cmatch, ccnt, counts, nullptr, nullptr);
    PH_HIP(hipGetLastError());
    ph::ScanPublish pub;
    PH_CHECK(ctx->arm_count(&pub));
    PH_CHECK(ph::exclusive_scan_i32(ctx, counts, nb, total, pub.seq ? &pub : nullptr));
    const int wave_grid = (int)std::min<int64_t>((nb + 3) / 4, (int64_t)ctx->cu_count * 8);
    const uint64_t mask = (uint64_t)j->cap - 1;
#define PH_BE_ARGS P.key[0].data, P.key[1].data, P.sel, n, j->head, mask, j->nodes, ccnt, cmatch, counts, nb, cap, out_probe_dev, out_build_dev
    if (j->big_nk == 2) { if (P.sel) ph::big_emit_kernel<4, 2, true><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); else ph::big_emit_kernel<4, 2, false><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); }
    else if (j->big_kw == 4) { if (P.sel) ph::big_emit_kernel<4, 1, true><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); else ph::big_emit_kernel<4, 1, false><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); }
    else { if (P.sel) ph::big_emit_kernel<8, 1, true><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); else ph::big_emit_kernel<8, 1, false><<<wave_grid, 256, 0, ctx->stream>>>(PH_BE_ARGS); }
#undef PH_BE_ARGS
    PH_HIP(hipGetLastError());
    return ctx->count_back(pub, n_out, total, cap, "ph_join_probe_inner");
}

// the radix form answers inner probes; marks and lookups go through the node table, built here on first use
static int ensure_nodes(ph_join *j) {
    if (j->nodes || !j->rj_tables) return PH_OK;
    ph_ctx *ctx = j->ctx;
    const int bparts = (int)(j->cap >> ph::BG_SLICE_LOG);
    if (bparts < 2 || bparts > ph::BG_MAX_PARTS) { ph::set_error("ph_join: this probe kind needs the node table, which does not take %lld build rows", (long long)j->build.n); return PH_EUNSUPPORTED; }
    PH_CHECK(ctx->pool_alloc(j->cap * 4, (void **)&j->head));
    if (j->count_dev) { ctx->pool_release(j->count_dev); j->count_dev = nullptr; }   // (build_big allocates its own; the count is the same)
    return build_big(j, j->big_kw, bparts);
}

static int join_build_impl(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n, bool have_range,
                           int64_t key_lo, int64_t key_hi, bool fk_probes, ph_join **out,
                           const ph::RangePred &where = ph::RangePred{0, nullptr, nullptr, 0, 0}, bool sorted_unique = false, bool exists_only = false) {
    PH_REQUIRE(ctx && keys && out && nkeys >= 1 && nkeys <= ph::JOIN_MAX_KEYS && n >= 0 && n < (1ll << 31),
               "ph_join_build: bad arguments (1..%d keys)", ph::JOIN_MAX_KEYS);
    ph_join *j = new ph_join();
    j->ctx = ctx;
    int rc = fill_side(&j->build, keys, nkeys, sel, n);
    if (rc != PH_OK) { delete j; return rc; }
    if (!have_range && nkeys == 1 && where.kind == 0 && n >= (1ll << 20)) {
        // No range from the caller and a build side big enough for the node table (whose probes cost a 128-byte line per row,
        // hashing away whatever order the probe keys have: 15 M keys + 60 M probes 1.53 ms): read the key range off the column
        // first — one streaming pass and one host round trip (~40 us for 15 M keys) — and let the density test below decide.
        // Keys that are dense in their range (any surrogate key, also one that arrives here as an intermediate result without
        // statistics) get the direct table: build + probe 0.11 + 0.45 ms for the same sizes.
        const char *ar = getenv("PH_JOIN_AUTO_RANGE");   // read per call: the tests build both forms over the same keys
        const int t = keys[0].type;
        const int kw = (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8;
        if (!(ar && atoi(ar) == 0) && kw != 1) {
            long long *mm = nullptr;
            if (ctx->pool_alloc(16, (void **)&mm) != PH_OK) { ph_join_free(j); ph::set_error("ph_join_build: allocation failed"); return PH_EHIP; }
            long long res[2] = {0, -1};
            const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 2);
            bool ok = true;   // (a copy of the two initial words from pageable host memory costs ~0.2 ms: a one-thread kernel instead)
            ph::join_key_range_init_kernel<<<1, 1, 0, ctx->stream>>>(mm);
            {
                const ph::JoinSide &Bs = j->build;
                if (kw == 4) ph::join_key_range_kernel<4><<<grid, 256, 0, ctx->stream>>>(Bs.key[0].data, Bs.key[0].validity, Bs.sel, n, mm);
                else ph::join_key_range_kernel<8><<<grid, 256, 0, ctx->stream>>>(Bs.key[0].data, Bs.key[0].validity, Bs.sel, n, mm);
                ok = hipGetLastError() == hipSuccess && ctx->download_plain(res, mm, sizeof res) == PH_OK;
            }
            ctx->pool_release(mm);
            if (!ok) { ph_join_free(j); ph::set_error("ph_join_build: key range pass failed"); return PH_EHIP; }
            if (res[0] <= res[1]) { have_range = true; key_lo = res[0]; key_hi = res[1]; }
        }
    }
    if (have_range && nkeys == 1 && n > 0 && key_hi >= key_lo) {
        // dense keys: a direct table when the range is at most 8 slots per build row (a primary-key
        // column, possibly filtered) and at most 2^30 slots; sparse build sides keep the hash tables,
        // whose Bloom bitmap rejects most probes from cache
        const char *dz = getenv("PH_JOIN_DIRECT");   // read per call: the tests build both forms over the same keys
        const int t = keys[0].type;
        const int kw = (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8;
        const unsigned long long span = (unsigned long long)key_hi - (unsigned long long)key_lo;
        // ... or the range itself is small (<= 4 M slots = 16 MiB, and the build side <= 256 K rows so that it
        // gets the occupied-group bitmap): the table read is the exact test, no chain pass
        if (exists_only && where.kind == 0 && kw != 1 && span < (64ull << 20) && n >= (256 << 10) && !(dz && atoi(dz) == 0)) {
            // only "is there a build row with this key" will be asked, and the build side is big: one bit per key value (<= 8 MiB)
            int rce = build_exists(j, kw, key_lo, (int64_t)span + 1);
            if (rce != PH_OK) { ph_join_free(j); return rce; }
            *out = j;
            return PH_OK;
        }
        const bool dense = (int64_t)span + 1 <= std::max<int64_t>(8 * n, 4096);
        const bool small_range = span < (4ull << 20) && n <= (256 << 10);
        if (!(dz && atoi(dz) == 0) && kw != 1 && span < (1ull << 30) && (dense || small_range)) {
            int rcd = build_direct(j, kw, key_lo, (int64_t)span + 1, where, sorted_unique);
            if (rcd != PH_OK) { ph_join_free(j); return rcd; }
            *out = j;
            return PH_OK;
        }
    }
    if (where.kind != 0) {   // only the direct table is built through a filter: its size does not depend on how many rows pass
        ph::set_error("ph_join_build_where: the key range [%lld, %lld] does not give a direct table for %lld build rows; run "
                      "ph_filter_select and ph_join_build", (long long)key_lo, (long long)key_hi, (long long)n);
        ph_join_free(j);
        return PH_EUNSUPPORTED;
    }
    // pointer table: cap = max(nextpow2(2n), 1024) (pointerTableCap, join_table.go:197-199)
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    j->cap = cap;
    auto fail = [&](const char *what) { ph::set_error("ph_join_build: %s failed", what); ph_join_free(j); return PH_EHIP; };
    {   // large build sides whose keys are not dense in a range: both sides partitioned by key hash, tables built in LDS (see rj_*)
        const char *re = getenv("PH_JOIN_RADIX"), *rm = getenv("PH_JOIN_RADIX_MIN");   // read per call: the tests build every form over the same keys
        const int64_t radix_min = rm ? atoll(rm) : (4ll << 20) + 1;
        auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
        const ph::JoinSide &Bs = j->build;
        const int kw = width(Bs.key[0].type);
        const bool packable = (Bs.nkeys == 1 && kw != 1) || (Bs.nkeys == 2 && kw == 4 && width(Bs.key[1].type) == 4);
        if (!(re && atoi(re) == 0) && !fk_probes && packable && n >= radix_min) {
            if (sel) {   // own copy: a probe kind the radix form does not answer builds the node table later, from the same rows
                if (ctx->pool_alloc(n * 4, (void **)&j->sel_copy) != PH_OK) return fail("alloc(sel)");
                if (hipMemcpyAsync(j->sel_copy, sel, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return fail("copy sel");
                j->build.sel = j->sel_copy;
            }
            const int rcr = build_radix(j, kw);
            if (rcr == PH_OK) { j->count = -1; *out = j; return PH_OK; }
            if (rcr != PH_EUNSUPPORTED) { ph_join_free(j); return rcr; }
        }
    }
    if (ctx->pool_alloc(cap * 4, (void **)&j->head) != PH_OK) return fail("alloc(head)");
    {   // large build sides with a packable key: the node table (see BigNode)
        const char *bm = getenv("PH_JOIN_BIG_MIN");   // read per call: the tests lower it to cover this path at small sizes
        // foreign-key probes (nearly every probe row matches): a Bloom bitmap rejects nothing, and a chain
        // step of the node table is one 16-byte read where the chained layout needs next + one read per
        // key column — Q9's 3.3 M composite-key lookups into 0.43 M partsupp rows: 136 -> 60 us
        const int64_t big_min = bm ? atoll(bm) : fk_probes ? (32 << 10) : (4ll << 20) + 1;
        auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
        const ph::JoinSide &Bs = j->build;
        const int kw = width(Bs.key[0].type);
        const bool packable = (Bs.nkeys == 1 && kw != 1) || (Bs.nkeys == 2 && kw == 4 && width(Bs.key[1].type) == 4);
        const int bparts = (int)(cap >> ph::BG_SLICE_LOG);
        if (n >= big_min && packable && bparts >= 2 && bparts <= ph::BG_MAX_PARTS) {
            int rcb = build_big(j, kw, bparts);
            if (rcb != PH_OK) { ph_join_free(j); return rcb; }
            j->count = -1;
            *out = j;
            return PH_OK;
        }
    }
    if (ctx->pool_alloc(std::max<int64_t>(n, 1) * 4, (void **)&j->next) != PH_OK) return fail("alloc(next)");
    if (sel && n > 0 && !j->sel_copy) {  // keep our own copy: the table outlives the caller's selection buffer
        if (ctx->pool_alloc(n * 4, (void **)&j->sel_copy) != PH_OK) return fail("alloc(sel)");
        if (hipMemcpyAsync(j->sel_copy, sel, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return fail("copy sel");
        j->build.sel = j->sel_copy;
    }
    const int nparts = (int)(cap >> ph::PB_SLICE_LOG);
    static const bool no_part = getenv("PH_JOIN_ATOMIC_BUILD") != nullptr;
    // only where the atomic build needs two atomics per row (a bitmap is built): without one it runs
    // at ~21 G rows/s, faster than the three passes of the partitioned build (~12 G rows/s)
    static const int64_t part_max = getenv("PH_JOIN_PART_MAX") ? atoll(getenv("PH_JOIN_PART_MAX")) : (4ll << 20);
    const bool partitioned = !no_part && n >= (128 << 10) && n <= part_max && nparts >= 2 && nparts <= ph::PB_MAX_PARTS;
    int64_t bits = 0;
    if (n > 0 && n <= (4ll << 20)) {  // bitmap of >= 16 bits per key, at most 16 MiB
        bits = 1 << 16;
        while (bits < 16 * n) bits <<= 1;
        if (ctx->pool_alloc(bits / 8, (void **)&j->bloom.bits) != PH_OK) return fail("alloc(bloom)");
        j->bloom.word_mask = (uint64_t)(bits / 32) - 1;
        j->bloom.inner_mask = j->bloom.word_mask;
        if (partitioned) {
            int logp = 0, logw = 0;
            while ((1 << logp) < nparts) logp++;
            while ((1ll << logw) < bits / 32) logw++;
            j->bloom.pmask = (uint32_t)nparts - 1;
            j->bloom.pshift = ph::PB_SLICE_LOG;
            j->bloom.hi_shift = (uint32_t)(logw - logp);
            j->bloom.inner_mask = ((uint64_t)(bits / 32) >> logp) - 1;
        }
    }
    if (bits && n <= (256ll << 10)) {   // tiny build side: the LDS-resident coarse bitmap of the probe
        if (ctx->pool_alloc(ph::CO_WORDS * 4, (void **)&j->bloom.coarse) != PH_OK) return fail("alloc(coarse)");
    }
    // number of inserted (non-NULL-key) rows: stays on the device until ph_join_count asks, so
    // building a table costs no host round trip
    if (ctx->pool_alloc(16, (void **)&j->count_dev) != PH_OK) return fail("alloc(count)");
    int *count = j->count_dev;
    // one initialisation launch for whatever this build needs cleared (the coarse bitmap, the row
    // counter and — for the atomic build — the head table and the bitmap): up to four memsets before
    if (n > 0) {
        const bool atomic_build = !partitioned;
        ph::join_init_kernel<<<ctx->cu_count * 2, 256, 0, ctx->stream>>>(atomic_build ? j->head : nullptr, cap,
                                                                          atomic_build && bits ? j->bloom.bits : nullptr, bits / 32,
                                                                          j->bloom.coarse, count);
        if (hipGetLastError() != hipSuccess) return fail("join_init_kernel launch");
    } else if (hipMemsetAsync(count, 0, 16, ctx->stream) != hipSuccess || hipMemsetAsync(j->head, 0xff, (size_t)cap * 4, ctx->stream) != hipSuccess)
        return fail("memset");
    if (partitioned) {
        const int64_t rows_per_wg = std::max<int64_t>(1024, ph::round_up((n + 511) / 512, 256));
        const int nwg = (int)((n + rows_per_wg - 1) / rows_per_wg);
        int32_t *counts = nullptr;
        ulonglong2 *part_rec = nullptr;
        const int64_t nc = (int64_t)nparts * nwg;
        if (ctx->pool_alloc(nc * 4, (void **)&counts) != PH_OK || ctx->pool_alloc(n * 16, (void **)&part_rec) != PH_OK)
            return fail("alloc(partition scratch)");
        const uint64_t mask = (uint64_t)cap - 1;
        // common shape (one or two keys of one width, no NULL keys): straight-line passes
        const ph::JoinSide &Bs = j->build;
        auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
        const int kw = width(Bs.key[0].type);
        const bool fast = Bs.nkeys <= 2 && !Bs.key[0].validity && kw != 1 &&
                          (Bs.nkeys == 1 || (!Bs.key[1].validity && width(Bs.key[1].type) == kw));
        const size_t hl = (size_t)nparts * 4;
        int rc2 = PH_OK;
#define PH_PART_FAST(KWV, NKV, SELV)                                                                                      \
    do {                                                                                                                  \
        ph::part_count_fast_kernel<KWV, NKV, SELV><<<nwg, 256, hl, ctx->stream>>>(Bs.key[0].data, Bs.key[1].data, Bs.sel, Bs.n, \
                                                                                 mask, nparts, rows_per_wg, counts);     \
        rc2 = ph::exclusive_scan_i32(ctx, counts, nc, (int64_t *)count);                                                  \
        ph::part_scatter_fast_kernel<KWV, NKV, SELV><<<nwg, 256, hl, ctx->stream>>>(Bs.key[0].data, Bs.key[1].data, Bs.sel,    \
                                                                                   Bs.n, mask, nparts, rows_per_wg, counts, \
                                                                                   part_rec);                             \
    } while (0)
        if (fast && kw == 4 && Bs.nkeys == 1 && Bs.sel) PH_PART_FAST(4, 1, true);
        else if (fast && kw == 4 && Bs.nkeys == 1) PH_PART_FAST(4, 1, false);
        else if (fast && kw == 4 && Bs.sel) PH_PART_FAST(4, 2, true);
        else if (fast && kw == 4) PH_PART_FAST(4, 2, false);
        else if (fast && Bs.nkeys == 1 && Bs.sel) PH_PART_FAST(8, 1, true);
        else if (fast && Bs.nkeys == 1) PH_PART_FAST(8, 1, false);
        else if (fast && Bs.sel) PH_PART_FAST(8, 2, true);
        else if (fast) PH_PART_FAST(8, 2, false);
        else {
            ph::part_count_kernel<<<nwg, 256, hl, ctx->stream>>>(j->build, mask, nparts, rows_per_wg, counts);
            rc2 = ph::exclusive_scan_i32(ctx, counts, nc, (int64_t *)count);   // total = inserted rows (low word read as int)
            ph::part_scatter_kernel<<<nwg, 256, hl, ctx->stream>>>(j->build, mask, nparts, rows_per_wg, counts, part_rec, j->next);
        }
#undef PH_PART_FAST
        const int bloom_words = bits ? (int)((bits / 32) / nparts) : 0;
        static bool lds_raised = false;
        if (!lds_raised) {  // 64 KiB head slice + up to 32 KiB bitmap slice: above the default dynamic LDS limit
            if (hipFuncSetAttribute((const void *)ph::part_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
                return fail("hipFuncSetAttribute");
            lds_raised = true;
        }
        ph::part_build_kernel<<<nparts, 1024, (size_t)(ph::PB_SLICE + bloom_words) * 4, ctx->stream>>>(
            counts, nwg, nparts, (const int64_t *)count, part_rec, j->head, j->next, j->bloom, bloom_words);
        const bool bad = rc2 != PH_OK || hipGetLastError() != hipSuccess;
        ctx->pool_release(counts); ctx->pool_release(part_rec);
        if (bad) return fail("partitioned build launch");
    } else {
        if (n > 0) {
            int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
            ph::join_build_kernel<<<grid, 256, 0, ctx->stream>>>(j->build, j->head, (uint64_t)cap - 1, j->next, count, j->bloom);
            if (hipGetLastError() != hipSuccess) return fail("join_build_kernel launch");
        }
    }
    j->count = n == 0 ? 0 : -1;
    *out = j;
    return PH_OK;
}

extern "C" int ph_join_build(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n, ph_join **out) {
    return join_build_impl(ctx, keys, nkeys, sel, n, false, 0, 0, false, out);
}

extern "C" int ph_join_build_range(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n, int64_t key_lo,
                                   int64_t key_hi, ph_join **out) {
    return join_build_impl(ctx, keys, nkeys, sel, n, true, key_lo, key_hi, false, out);
}

extern "C" int ph_join_build_ex(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n, int32_t flags,
                                int64_t key_lo, int64_t key_hi, ph_join **out) {
    PH_REQUIRE((flags & ~(PH_JOIN_KEY_RANGE | PH_JOIN_FK_PROBES | PH_JOIN_KEYS_SORTED_UNIQUE | PH_JOIN_EXISTS_ONLY)) == 0, "ph_join_build_ex: unknown flags %d", flags);
    return join_build_impl(ctx, keys, nkeys, sel, n, (flags & PH_JOIN_KEY_RANGE) != 0, key_lo, key_hi, (flags & PH_JOIN_FK_PROBES) != 0, out,
                           ph::RangePred{0, nullptr, nullptr, 0, 0}, (flags & PH_JOIN_KEYS_SORTED_UNIQUE) != 0, (flags & PH_JOIN_EXISTS_ONLY) != 0);
}

extern "C" int ph_join_build_where(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const ph_col *where_col, int32_t where_op,
                                   const ph_const *where_k, const int32_t *sel, int64_t n, int64_t key_lo, int64_t key_hi,
                                   ph_join **out) {
    return ph_join_build_where_ex(ctx, keys, nkeys, where_col, where_op, where_k, sel, n, 0, key_lo, key_hi, out);
}

extern "C" int ph_join_build_where_ex(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const ph_col *where_col, int32_t where_op,
                                      const ph_const *where_k, const int32_t *sel, int64_t n, int32_t flags, int64_t key_lo,
                                      int64_t key_hi, ph_join **out) {
    PH_REQUIRE(where_col && where_k, "ph_join_build_where: bad arguments");
    ph::RangePred where{};
    if (!ph::lower_range_pred(where_col, where_op, where_k, &where) || where.validity) {
        ph::set_error("ph_join_build_where: only integer-range predicates over a column without NULLs are fused into the build");
        return PH_EUNSUPPORTED;
    }
    if (where.kind < 0) { where.kind = 1; where.data = keys[0].data; where.lo = 1; where.hi = 0; }   // never true: an empty range over any column
    return join_build_impl(ctx, keys, nkeys, sel, n, true, key_lo, key_hi, (flags & PH_JOIN_FK_PROBES) != 0, out, where,
                           (flags & PH_JOIN_KEYS_SORTED_UNIQUE) != 0);
}

extern "C" const char *ph_join_kind(const ph_join *j) {
    return !j ? "" : j->exists_only ? "bitmap" : j->direct ? "direct" : j->rj_tables ? "radix" : j->nodes ? "nodes" : j->bloom.bits ? "chained+bloom" : "chained";
}

extern "C" int ph_join_pairs_ordered(const ph_join *j) { return j && !j->rj_tables ? 1 : 0; }

extern "C" int64_t ph_join_count(const ph_join *cj) {
    ph_join *j = const_cast<ph_join *>(cj);
    if (!j) return -1;
    if (j->count < 0 && j->direct && j->count_from_bits > 0) {
        int *tmp = nullptr, c = 0;
        if (j->ctx->pool_alloc(16, (void **)&tmp) != PH_OK) return -1;
        if (hipMemsetAsync(tmp, 0, 4, j->ctx->stream) != hipSuccess) return -1;
        ph::direct_popcount_kernel<<<j->ctx->cu_count, 256, 0, j->ctx->stream>>>(j->dbits, j->count_from_bits, tmp);
        const int rc = j->ctx->download(&c, tmp, 4);
        j->ctx->pool_release(tmp);
        if (rc != PH_OK) return -1;
        j->count = c;
    }
    if (j->count < 0 && j->direct) {
        int c[4] = {0, 0, 0, 0};
        if (j->ctx->download(c, j->count_dev, 16) != PH_OK) return -1;
        if (c[2] != 0) {
            ph::set_error("ph_join_build_range: %d build keys lie outside the stated range [%lld, %lld]", c[2], (long long)j->dlo,
                          (long long)(j->dlo + (int64_t)j->drange - 1));
            return -1;
        }
        j->count = c[0];
    }
    if (j->count < 0) {
        int c = 0;
        if (j->ctx->download(&c, j->count_dev, 4) != PH_OK) return -1;
        j->count = c;
    }
    return j->count;
}

static int check_probe(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, ph::JoinSide *P) {
    PH_REQUIRE(j && keys && n >= 0, "ph_join_probe: bad arguments");
    PH_CHECK(fill_side(P, keys, j->build.nkeys, sel, n));
    for (int c = 0; c < j->build.nkeys; c++) {
        int a = P->key[c].type, b = j->build.key[c].type;
        bool w32a = a == PH_I32 || a == PH_DATE, w32b = b == PH_I32 || b == PH_DATE;
        bool w64a = a == PH_I64 || a == PH_DEC64, w64b = b == PH_I64 || b == PH_DEC64;
        PH_REQUIRE((w32a && w32b) || (w64a && w64b) || (a == PH_CODE8 && b == PH_CODE8),
                   "ph_join_probe: key %d types differ (probe %d, build %d); cast on the host side first", c, a, b);
    }
    return PH_OK;
}

#define PH_NOT_EXISTS_ONLY(j, what) do { if ((j) && (j)->exists_only) { ph::set_error(what ": the table was built with PH_JOIN_EXISTS_ONLY (mark probes only)"); return PH_EUNSUPPORTED; } } while (0)

static int probe_inner_impl(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, const ph::RangePred &where,
                            int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out, const uint8_t *bflags = nullptr);

extern "C" int ph_join_probe_inner(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n,
                                   int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    return probe_inner_impl(j, keys, sel, n, ph::RangePred{0, nullptr, nullptr, 0, 0}, out_probe_dev, out_build_dev, cap, n_out);
}

extern "C" int ph_join_probe_inner_where(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op,
                                         const ph_const *where_k, const int32_t *sel, int64_t n, int32_t *out_probe_dev,
                                         int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    PH_REQUIRE(j && where_col && where_k, "ph_join_probe_inner_where: bad arguments");
    ph::RangePred where{};
    if ((!j->bloom.bits && !j->direct) || !ph::lower_range_pred(where_col, where_op, where_k, &where) ||
        (j->direct && (where.validity || where.kind < 0))) {
        ph::set_error("ph_join_probe_inner_where: only integer-range predicates over a probe of a table with a Bloom bitmap (or of a "
                      "direct table, over a column without NULLs) are fused; run ph_filter_select and ph_join_probe_inner");
        return PH_EUNSUPPORTED;
    }
    return probe_inner_impl(j, keys, sel, n, where, out_probe_dev, out_build_dev, cap, n_out);
}

// Inner probe with a RESIDUAL predicate on the build row: of the pairs ph_join_probe_inner[_where] would
// emit, those whose build row r has build_flags_dev[r] != 0. What a join whose build child is
// Filter / SemiJoin(build table) becomes when the filter's result is a flag per build-table row
// instead of a selection: the build side stays the whole table (a primary-key column in storage order
// builds in one pass, and the table can be shared by every plan that joins on that key), and neither
// the filter's row count nor a filtered key list ever reaches the host. Direct tables built without a
// selection only; PH_EUNSUPPORTED otherwise (the caller filters, builds and probes).
extern "C" int ph_join_probe_inner_residual(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op,
                                            const ph_const *where_k, const uint8_t *build_flags_dev, const int32_t *sel, int64_t n,
                                            int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out) {
    PH_REQUIRE(j && build_flags_dev, "ph_join_probe_inner_residual: bad arguments");
    ph::RangePred where{0, nullptr, nullptr, 0, 0};
    const bool has_where = where_col != nullptr;
    if (!j->direct || j->build.sel || (has_where && (!where_k || !ph::lower_range_pred(where_col, where_op, where_k, &where) || where.validity || where.kind < 0))) {
        ph::set_error("ph_join_probe_inner_residual: only direct tables built without a selection, and integer-range probe filters over a "
                      "column without NULLs");
        return PH_EUNSUPPORTED;
    }
    return probe_inner_impl(j, keys, sel, n, where, out_probe_dev, out_build_dev, cap, n_out, build_flags_dev);
}

static int probe_inner_impl(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, const ph::RangePred &where,
                            int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out, const uint8_t *bflags) {
    PH_NOT_EXISTS_ONLY(j, "ph_join_probe_inner");
    ph::JoinSide P{};
    PH_CHECK(check_probe(j, keys, sel, n, &P));
    PH_REQUIRE(n_out && cap >= 0 && (cap == 0 || (out_probe_dev && out_build_dev)), "ph_join_probe_inner: bad output arguments");
    *n_out = 0;
    if (n == 0 || j->build.n == 0) return PH_OK;
    ph_ctx *ctx = j->ctx;
    if (j->direct) {
        if (!direct_probe_ok(j, P)) { ph::set_error("ph_join_probe_inner: probe key shape differs from the direct table's"); return PH_EUNSUPPORTED; }
        return direct_probe_inner(j, P, n, where, bflags, out_probe_dev, out_build_dev, cap, n_out);
    }
    if (j->rj_tables) {
        if (!big_probe_ok(j, P)) { ph::set_error("ph_join_probe_inner: probe key shape differs from the table's"); return PH_EUNSUPPORTED; }
        return radix_probe_inner(j, P, n, out_probe_dev, out_build_dev, cap, n_out);
    }
    if (j->nodes) {
        if (!big_probe_ok(j, P)) { ph::set_error("ph_join_probe_inner: probe key shape differs from the node table's"); return PH_EUNSUPPORTED; }
        return big_probe_inner(j, P, n, out_probe_dev, out_build_dev, cap, n_out);
    }
    int64_t nb = (n + ph::JP_CHUNK - 1) / ph::JP_CHUNK;
    const bool selective = j->bloom.bits != nullptr;  // candidate lists pay off when most probes miss
    const int64_t o_ccount = ph::round_up(nb * 4, 8) + 64;
    const int64_t o_cand = ph::round_up(o_ccount + nb * 4, 8), o_ccnt = o_cand + nb * ph::JP_CHUNK * 2;
    const int64_t o_cmatch = o_ccnt + nb * ph::JP_CHUNK * 2;
    PH_CHECK(ctx->ensure_scratch(o_cmatch + nb * ph::JP_CHUNK * 4));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(nb * 4, 8));
    uint64_t mask = (uint64_t)j->cap - 1;
    ph::ScanPublish pub;
    {
        int32_t *ccount = (int32_t *)((char *)ctx->scratch + o_ccount);
        uint16_t *cand = (uint16_t *)((char *)ctx->scratch + o_cand), *ccnt = (uint16_t *)((char *)ctx->scratch + o_ccnt);
        int32_t *cmatch = (int32_t *)((char *)ctx->scratch + o_cmatch);
        const int wave_grid = (int)std::min<int64_t>((nb + 3) / 4, (int64_t)ctx->cu_count * 8);
        if (!selective) ph::join_cand_all_kernel<<<(int)nb, 256, 0, ctx->stream>>>(n, cand, ccount);
        else if ((ph::g_cu_count = ctx->cu_count, !ph::try_cand_fast((int)nb, ctx->stream, P, j->bloom, where, cand, ccount)))
            ph::join_cand_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, j->bloom, where, cand, ccount);
        if (!ph::try_chain_fast(wave_grid, ctx->stream, j->build, P, j->head, mask, j->next, cand, ccount, ccnt, cmatch, counts, nb))
            ph::join_chain_kernel<<<wave_grid, 256, 0, ctx->stream>>>(j->build, P, j->head, mask, j->next, cand, ccount, ccnt, cmatch, counts, nb);
        PH_HIP(hipGetLastError());
        PH_CHECK(ctx->arm_count(&pub));
        PH_CHECK(ph::exclusive_scan_i32(ctx, counts, nb, total, pub.seq ? &pub : nullptr));
        ph::join_emit_kernel<<<wave_grid, 256, 0, ctx->stream>>>(j->build, P, j->head, mask, j->next, cand, ccount, ccnt, cmatch, counts, nb, cap,
                                                                 out_probe_dev, out_build_dev);
        PH_HIP(hipGetLastError());
    }
    return ctx->count_back(pub, n_out, total, cap, "ph_join_probe_inner");
}

// Filter -> semi-join mark in one pass: found_dev[i] = (row i passes the comparison) && (its key is in the
// table). Direct tables, identity selection, columns without NULLs and 16-byte aligned; PH_EUNSUPPORTED
// otherwise (the caller runs ph_filter_select + ph_join_probe_mark).
extern "C" int ph_join_probe_mark_where(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op, const ph_const *where_k,
                                        int64_t n, uint8_t *found_dev) {
    PH_REQUIRE(j && keys && where_col && where_k && n >= 0 && (n == 0 || found_dev), "ph_join_probe_mark_where: bad arguments");
    ph::JoinSide P{};
    PH_CHECK(check_probe(j, keys, nullptr, n, &P));
    ph::RangePred w{};
    auto aligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((!j->direct && !j->exists_only) || !direct_probe_ok(j, P) || P.key[0].validity || !ph::lower_range_pred(where_col, where_op, where_k, &w) || w.validity ||
        w.kind <= 0 || !aligned(P.key[0].data) || !aligned(w.data) || !aligned(found_dev)) {
        ph::set_error("ph_join_probe_mark_where: only direct tables, an integer-range filter and aligned columns without NULLs");
        return PH_EUNSUPPORTED;
    }
    if (n == 0) return PH_OK;
    ph_ctx *ctx = j->ctx;
    if (j->build.n == 0) { PH_HIP(hipMemsetAsync(found_dev, 0, (size_t)n, ctx->stream)); return PH_OK; }
    const int nb = (int)((n + ph::JP_CHUNK - 1) / ph::JP_CHUNK);
    const char *mle = getenv("PH_JOIN_MARK_LDS");   // read per call: the test marks both ways in one process
    if (j->dbits && !(mle && atoi(mle) == 0) && n >= (1 << 20) && j->drange <= (8ull << 20)) {
        // big probe side, bitmap of at most 8 Mbit: the LDS-staged (folded) image, one workgroup per CU
        const int64_t dwords = (int64_t)((j->drange + 31) / 32);
        const size_t lds = (size_t)ph::CO_WORDS * 4;
        const int grid = ctx->cu_count;
#define PH_MWL(KWV, WKV)                                                                                                                   \
    do {                                                                                                                                   \
        (void)hipFuncSetAttribute((const void *)ph::direct_mark_where_lds_kernel<KWV, WKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        ph::direct_mark_where_lds_kernel<KWV, WKV><<<grid, 1024, lds, ctx->stream>>>(P.key[0].data, n, (long long)j->dlo, j->drange, j->dbits, dwords, \
                                                                                      w.data, w.lo, w.hi, found_dev);                      \
    } while (0)
        if (j->dkw == 4) { if (w.kind == 1) PH_MWL(4, 1); else if (w.kind == 2) PH_MWL(4, 2); else PH_MWL(4, 3); }
        else { if (w.kind == 1) PH_MWL(8, 1); else if (w.kind == 2) PH_MWL(8, 2); else PH_MWL(8, 3); }
#undef PH_MWL
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (j->exists_only) { ph::set_error("ph_join_probe_mark_where: a bitmap table of this size takes ph_join_probe_mark behind the filter"); return PH_EUNSUPPORTED; }
#define PH_MW(KWV, WKV) ph::direct_mark_where_kernel<KWV, WKV><<<nb, 256, 0, ctx->stream>>>(P.key[0].data, n, (long long)j->dlo, j->drange, j->direct, j->dbits, (int32_t)j->build.n, w.data, w.lo, w.hi, found_dev)
    if (j->dkw == 4) { if (w.kind == 1) PH_MW(4, 1); else if (w.kind == 2) PH_MW(4, 2); else PH_MW(4, 3); }
    else { if (w.kind == 1) PH_MW(8, 1); else if (w.kind == 2) PH_MW(8, 2); else PH_MW(8, 3); }
#undef PH_MW
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_join_probe_mark(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, uint8_t *found_dev) {
    ph::JoinSide P{};
    PH_CHECK(check_probe(j, keys, sel, n, &P));
    PH_REQUIRE(n == 0 || found_dev, "ph_join_probe_mark: found_dev is NULL");
    if (n == 0) return PH_OK;
    ph_ctx *ctx = j->ctx;
    if (j->build.n == 0) { PH_HIP(hipMemsetAsync(found_dev, 0, (size_t)n, ctx->stream)); return PH_OK; }
    if (j->exists_only) {
        if (!direct_probe_ok(j, P)) { ph::set_error("ph_join_probe_mark: probe key shape differs from the table's"); return PH_EUNSUPPORTED; }
        const int grid = (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16);
        if (j->dkw == 4) { if (P.sel) ph::direct_mark_bits_kernel<4, true><<<grid, 256, 0, ctx->stream>>>(P.key[0].data, P.key[0].validity, P.sel, n, (long long)j->dlo, j->drange, j->eflags, found_dev);
                           else ph::direct_mark_bits_kernel<4, false><<<grid, 256, 0, ctx->stream>>>(P.key[0].data, P.key[0].validity, nullptr, n, (long long)j->dlo, j->drange, j->eflags, found_dev); }
        else { if (P.sel) ph::direct_mark_bits_kernel<8, true><<<grid, 256, 0, ctx->stream>>>(P.key[0].data, P.key[0].validity, P.sel, n, (long long)j->dlo, j->drange, j->eflags, found_dev);
               else ph::direct_mark_bits_kernel<8, false><<<grid, 256, 0, ctx->stream>>>(P.key[0].data, P.key[0].validity, nullptr, n, (long long)j->dlo, j->drange, j->eflags, found_dev); }
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (j->direct) {
        if (!direct_probe_ok(j, P)) { ph::set_error("ph_join_probe_mark: probe key shape differs from the direct table's"); return PH_EUNSUPPORTED; }
        launch_direct_probe<2>(j, P, n, (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16), nullptr, found_dev, nullptr);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (j->rj_tables && !j->nodes) PH_CHECK(ensure_nodes(j));
    if (j->nodes) {
        if (!big_probe_ok(j, P)) { ph::set_error("ph_join_probe_mark: probe key shape differs from the node table's"); return PH_EUNSUPPORTED; }
        launch_big_probe<2>(j, P, n, (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16), nullptr, nullptr, nullptr, found_dev, nullptr);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    const uint64_t mask = (uint64_t)j->cap - 1;
    if (j->bloom.bits) {
        // selective probe: same candidate slices + chain pass as ph_join_probe_inner (a lane that
        // walks a chain no longer holds up 63 idle ones); the flags follow from the match counts
        const int64_t nb = (n + ph::JP_CHUNK - 1) / ph::JP_CHUNK;
        const int64_t o_ccount = ph::round_up(nb * 4, 8) + 64, o_cand = ph::round_up(o_ccount + nb * 4, 8);
        const int64_t o_ccnt = o_cand + nb * ph::JP_CHUNK * 2, o_cmatch = o_ccnt + nb * ph::JP_CHUNK * 2;
        PH_CHECK(ctx->ensure_scratch(o_cmatch + nb * ph::JP_CHUNK * 4));
        int32_t *counts = (int32_t *)ctx->scratch, *ccount = (int32_t *)((char *)ctx->scratch + o_ccount);
        uint16_t *cand = (uint16_t *)((char *)ctx->scratch + o_cand), *ccnt = (uint16_t *)((char *)ctx->scratch + o_ccnt);
        int32_t *cmatch = (int32_t *)((char *)ctx->scratch + o_cmatch);
        const int wave_grid = (int)std::min<int64_t>((nb + 3) / 4, (int64_t)ctx->cu_count * 8);
        const ph::RangePred none{0, nullptr, nullptr, 0, 0};
        PH_HIP(hipMemsetAsync(found_dev, 0, (size_t)n, ctx->stream));
        ph::g_cu_count = ctx->cu_count;
        if (!ph::try_cand_fast((int)nb, ctx->stream, P, j->bloom, none, cand, ccount))
            ph::join_cand_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, j->bloom, none, cand, ccount);
        if (!ph::try_chain_fast(wave_grid, ctx->stream, j->build, P, j->head, mask, j->next, cand, ccount, ccnt, cmatch, counts, nb))
            ph::join_chain_kernel<<<wave_grid, 256, 0, ctx->stream>>>(j->build, P, j->head, mask, j->next, cand, ccount, ccnt, cmatch, counts, nb);
        ph::join_mark_set_kernel<<<wave_grid, 256, 0, ctx->stream>>>(cand, ccount, ccnt, found_dev, nb);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    ph::join_mark_kernel<<<grid, 256, 0, ctx->stream>>>(j->build, P, j->head, mask, j->next, found_dev, j->bloom);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ lookup probe (N:1)
template <int KW>
static void launch_lookup_fast(int grid, hipStream_t st, const ph::JoinSide &B, const ph::JoinSide &P, int64_t n, const int32_t *head,
                               uint64_t mask, const int32_t *next, const ph::Bloom &bl, int32_t *out, int *stats) {
#define PH_LU_ARGS B.key[0].data, B.key[1].data, B.sel, P.key[0].data, P.key[1].data, P.sel, n, head, mask, next, bl, out, stats
#define PH_LU_LAUNCH(NKV)                                                                                       \
    do {                                                                                                        \
        if (P.sel && B.sel) ph::join_lookup_fast_kernel<KW, true, true, NKV><<<grid, 256, 0, st>>>(PH_LU_ARGS);  \
        else if (P.sel) ph::join_lookup_fast_kernel<KW, true, false, NKV><<<grid, 256, 0, st>>>(PH_LU_ARGS);     \
        else if (B.sel) ph::join_lookup_fast_kernel<KW, false, true, NKV><<<grid, 256, 0, st>>>(PH_LU_ARGS);     \
        else ph::join_lookup_fast_kernel<KW, false, false, NKV><<<grid, 256, 0, st>>>(PH_LU_ARGS);               \
    } while (0)
    if (P.nkeys == 2) PH_LU_LAUNCH(2);
    else PH_LU_LAUNCH(1);
#undef PH_LU_LAUNCH
#undef PH_LU_ARGS
}

extern "C" int ph_join_lookup(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, int32_t *out_build_dev, int32_t *stats_dev);

extern "C" int ph_join_lookup_strict(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, int32_t *out_build_dev) {
    PH_REQUIRE(j, "ph_join_lookup_strict: join is NULL");
    int *words = nullptr;
    PH_CHECK(j->ctx->deferred_words(&words));
    int rc = ph_join_lookup(j, keys, sel, n, out_build_dev, words + 1);   // [1] misses, [2] multi-matches
    if (rc == PH_OK && n > 0) j->ctx->deferred_pending = true;
    return rc;
}

extern "C" int ph_join_lookup(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, int32_t *out_build_dev,
                              int32_t *stats_dev) {
    PH_NOT_EXISTS_ONLY(j, "ph_join_lookup");
    ph::JoinSide P{};
    PH_CHECK(check_probe(j, keys, sel, n, &P));
    if (n == 0) return PH_OK;
    PH_REQUIRE(out_build_dev != nullptr, "ph_join_lookup: out_build_dev is NULL");
    ph_ctx *ctx = j->ctx;
    int *stats = stats_dev;
    int *scratch = nullptr;
    if (!stats) {   // the kernel always counts; without a caller buffer the counts are dropped
        PH_CHECK(ctx->pool_alloc(8, (void **)&scratch));
        stats = scratch;
    }
    if (j->direct) {
        int rcb = PH_OK;
        if (!direct_probe_ok(j, P)) { ph::set_error("ph_join_lookup: probe key shape differs from the direct table's"); rcb = PH_EUNSUPPORTED; }
        else {
            launch_direct_probe<0>(j, P, n, (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16), out_build_dev, nullptr, stats);
            if (hipGetLastError() != hipSuccess) { ph::set_error("ph_join_lookup: kernel launch failed"); rcb = PH_EHIP; }
        }
        if (scratch) ctx->pool_release(scratch);
        return rcb;
    }
    if (j->rj_tables && !j->nodes) PH_CHECK(ensure_nodes(j));
    if (j->nodes) {
        int rcb = PH_OK;
        if (!big_probe_ok(j, P)) { ph::set_error("ph_join_lookup: probe key shape differs from the node table's"); rcb = PH_EUNSUPPORTED; }
        else {
            launch_big_probe<0>(j, P, n, (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 16), out_build_dev, nullptr, nullptr, nullptr, stats);
            if (hipGetLastError() != hipSuccess) { ph::set_error("ph_join_lookup: kernel launch failed"); rcb = PH_EHIP; }
        }
        if (scratch) ctx->pool_release(scratch);
        return rcb;
    }
    const ph::JoinSide &B = j->build;
    const uint64_t mask = (uint64_t)j->cap - 1;
    const int grid = (int)std::min<int64_t>((n + 256 * ph::LU - 1) / (256 * ph::LU), (int64_t)ctx->cu_count * 8);
    auto width = [](int t) { return (t == PH_I32 || t == PH_DATE) ? 4 : t == PH_CODE8 ? 1 : 8; };
    const int kw = width(P.key[0].type);
    bool fast = B.n > 0 && P.nkeys <= 2 && !P.key[0].validity && !B.key[0].validity && kw == width(B.key[0].type) && kw != 1;
    if (fast && P.nkeys == 2)
        fast = !P.key[1].validity && !B.key[1].validity && width(P.key[1].type) == kw && width(B.key[1].type) == kw;
    if (B.n == 0) {
        ph::JoinSide Bz = B;
        ph::join_lookup_kernel<<<std::min<int64_t>((n + 255) / 256, 2048), 256, 0, ctx->stream>>>(Bz, P, j->head, mask, j->next, j->bloom, out_build_dev, stats);
    } else if (fast && kw == 4) launch_lookup_fast<4>(grid, ctx->stream, B, P, n, j->head, mask, j->next, j->bloom, out_build_dev, stats);
    else if (fast) launch_lookup_fast<8>(grid, ctx->stream, B, P, n, j->head, mask, j->next, j->bloom, out_build_dev, stats);
    else ph::join_lookup_kernel<<<(int)std::min<int64_t>((n + 255) / 256, 2048), 256, 0, ctx->stream>>>(B, P, j->head, mask, j->next, j->bloom, out_build_dev, stats);
    const bool bad = hipGetLastError() != hipSuccess;
    if (scratch) ctx->pool_release(scratch);
    if (bad) { ph::set_error("ph_join_lookup: kernel launch failed"); return PH_EHIP; }
    return PH_OK;
}
