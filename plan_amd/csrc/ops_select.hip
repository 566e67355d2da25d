// Operator-granular filter: ph_filter_select, plus the shared block-scan, gather and partition
// kernels.
//
// Replaces ExprExec.executeSelect -> execSelectCompare -> selectOperation -> selectFlat ->
// selectFlatLoop (reference pkg/compute/expr_exec.go:342-442,
// function_operator_boolean.go:393-521, 672-868) for one `column OP constant` comparison over a
// device batch, narrowing an optional input selection exactly as execSelectAnd chains conjuncts
// (expr_exec.go:444-486). The output selection is ascending, like the reference's trueSel:
// every workgroup owns a contiguous run of candidates, lanes evaluate consecutive candidates, a
// wavefront ballot + popcount gives each passing lane its rank, and an exclusive scan over the
// workgroup totals places the runs (count pass, scan, write pass — no atomics, deterministic).
#include <algorithm>

#include "common.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

enum SelKind {
    SK_NEVER = 0,
    SK_RANGE_I32,   // lo <= v <= hi on int32 (INTEGER, DATE)
    SK_NE_I32,
    SK_RANGE_I64,   // DECIMAL '>' literal
    SK_RANGE_U8,    // dictionary code = code of the literal
    SK_NE_U8,
    SK_F32_DEC,     // float32(decimal) OP float32 literal
    SK_F32,         // float32 column
    SK_F64,         // float64 column ('<' only)
    SK_STR,         // PH_STR: =, !=, LIKE, NOT LIKE on offsets+bytes
    SK_CMP2_I32,    // column OP column, 4-byte integers (INTEGER, DATE): ph_filter_select_cols
    SK_CMP2_I64,    // column OP column, 8-byte integers (DECIMAL of one scale)
    SK_IN_I32,      // value IN (a short list): ph_filter_select_in
    SK_IN_U8
};

struct SelParams {
    int kind;
    int op;
    const void *data;
    const uint8_t *validity;
    const char *bytes;
    long long lo, hi;
    float kf;
    double kd;
    double div;  // 10^scale
    int plen;
    int contains;  // LIKE / NOT LIKE pattern is %literal% (no _ and no inner %): substring search
    int lit2_at, lit1_len, lit2_len;   // pattern %A%B% (no _): A = pat[1 .. 1+lit1_len), B = pat[lit2_at .. lit2_at+lit2_len); lit2_at == 0: not that shape
    char pat[96];
    const void *data2;          // SK_CMP2_*: the right-hand column
    const uint8_t *validity2;
    int nin;                    // SK_IN_I32: the list
    int inv[16];
    unsigned inbits[8];         // SK_IN_U8: one bit per dictionary code
};

__device__ __forceinline__ bool icmp(int op, long long a, long long b) {
    switch (op) {
    case PH_EQ: return a == b;
    case PH_NE: return a != b;
    case PH_LT: return a < b;
    case PH_LE: return a <= b;
    case PH_GT: return a > b;
    case PH_GE: return a >= b;
    default: return false;
    }
}

__device__ __forceinline__ bool fcmp(int op, double v, double k) {
    switch (op) {
    case PH_GT: return v > k;
    case PH_GE: return v >= k;
    case PH_LT: return v < k;
    case PH_LE: return v <= k;
    default: return false;
    }
}

__device__ __forceinline__ bool sel_pred(const SelParams &P, int64_t r) {
    if (!bit_valid(P.validity, r)) return false;  // NULL never selects (selectFlatLoop :842-866)
    switch (P.kind) {
    case SK_RANGE_I32: {
        long long v = ((const int32_t *)P.data)[r];
        return v >= P.lo && v <= P.hi;
    }
    case SK_NE_I32: return ((const int32_t *)P.data)[r] != (int32_t)P.lo;
    case SK_RANGE_I64: {
        long long v = ((const int64_t *)P.data)[r];
        return v >= P.lo && v <= P.hi;
    }
    case SK_RANGE_U8: {
        long long v = ((const uint8_t *)P.data)[r];
        return v >= P.lo && v <= P.hi;
    }
    case SK_NE_U8: return ((const uint8_t *)P.data)[r] != (uint8_t)P.lo;
    case SK_F32_DEC: {
        // tryCastDecimalToFloat32 (function_cast.go:349-354): decimal -> float64 -> float32.
        // IEEE division of two exactly representable doubles is the correctly rounded value of
        // the decimal, which is what the reference's string round trip produces.
        float v = (float)((double)((const int64_t *)P.data)[r] / P.div);
        return fcmp(P.op, (double)v, (double)P.kf);
    }
    case SK_F32: return fcmp(P.op, (double)((const float *)P.data)[r], (double)P.kf);
    case SK_F64: return fcmp(P.op, ((const double *)P.data)[r], P.kd);
    case SK_IN_I32: {
        const int v = ((const int32_t *)P.data)[r];
        bool hit = false;
        for (int q = 0; q < P.nin; q++) hit = hit || v == P.inv[q];
        return hit;
    }
    case SK_IN_U8: {
        const unsigned v = ((const uint8_t *)P.data)[r];
        return (P.inbits[v >> 5] >> (v & 31)) & 1u;
    }
    case SK_CMP2_I32: return bit_valid(P.validity2, r) && icmp(P.op, ((const int32_t *)P.data)[r], ((const int32_t *)P.data2)[r]);
    case SK_CMP2_I64: return bit_valid(P.validity2, r) && icmp(P.op, ((const int64_t *)P.data)[r], ((const int64_t *)P.data2)[r]);
    case SK_STR: {
        const int32_t *off = (const int32_t *)P.data;
        const char *s = P.bytes + off[r];
        int slen = off[r + 1] - off[r];
        if (P.op == PH_LIKE) return like_match(s, slen, P.pat, P.plen);
        if (P.op == PH_NOTLIKE) return !like_match(s, slen, P.pat, P.plen);
        bool eq = slen == P.plen;
        for (int i = 0; eq && i < slen; i++) eq = s[i] == P.pat[i];
        return P.op == PH_EQ ? eq : !eq;
    }
    default: return false;
    }
}

constexpr int SEL_ROUNDS = 8;
constexpr int SEL_CHUNK = 256 * SEL_ROUNDS;
constexpr int SEL_STR_LDS = 24 * 1024;  // staged string bytes per round (256 rows)

// `flags` (optional): the count pass leaves every wave's ballot there ([block][round][wave]) and
// the write pass reads it back instead of evaluating the predicate a second time — used for the
// string predicates, where a LIKE match costs far more than the 1 bit per row of the memo.
__global__ __launch_bounds__(256) void select_count_kernel(SelParams P, const int32_t *__restrict__ sel_in,
                                                           int64_t n_in, int32_t *__restrict__ block_counts,
                                                           unsigned long long *__restrict__ flags) {
    int64_t base = (int64_t)blockIdx.x * SEL_CHUNK;
    int cnt = 0;
    // string predicates over consecutive rows: the 256 strings of a round are one contiguous byte
    // range, copied to LDS with coalesced 4-byte reads so the matcher's byte-by-byte walk never
    // touches global memory (a LIKE over 2M part names was bound by 1-byte global loads)
    __shared__ __attribute__((aligned(16))) char sbuf[SEL_STR_LDS];
    __shared__ int soff[257];   // string offsets of the round's rows
    __shared__ char spat[96];   // the pattern too: indexing the kernel argument costs a global load per character
    const bool stage = P.kind == SK_STR && sel_in == nullptr;
    if (stage) {
        if (threadIdx.x < 96) spat[threadIdx.x] = P.pat[threadIdx.x];
        __syncthreads();
    }
    for (int r = 0; r < SEL_ROUNDS; r++) {
        int64_t i = base + r * 256 + threadIdx.x;
        bool pass = false;
        bool staged = false;
        if (stage) {
            const int64_t first = base + r * 256;
            if (first < n_in) {   // workgroup-uniform
                const int64_t last = first + 256 < n_in ? first + 256 : n_in;
                const int nrow = (int)(last - first);
                __syncthreads();  // previous round's readers are done
                if (threadIdx.x <= nrow) soff[threadIdx.x] = ((const int32_t *)P.data)[first + threadIdx.x];
                if (threadIdx.x == 0) soff[nrow] = ((const int32_t *)P.data)[last];
                __syncthreads();
                const int64_t b0 = soff[0] & ~3, b1 = soff[nrow];
                if (b1 - b0 <= SEL_STR_LDS) {
                    for (int64_t w = threadIdx.x; w < (b1 - b0 + 3) / 4; w += 256)
                        reinterpret_cast<int *>(sbuf)[w] = *reinterpret_cast<const int *>(P.bytes + b0 + w * 4);
                    __syncthreads();
                    staged = true;
                    if (P.contains) {
                        // %literal%: every thread tests byte positions of the whole staged range
                        // (no per-string loop, no divergence); a hit marks the row that owns the
                        // position unless the literal would run past that row's end
                        __shared__ unsigned rowhit[8];
                        if (threadIdx.x < 8) rowhit[threadIdx.x] = 0;
                        __syncthreads();
                        const int L = P.plen - 2;
                        const int64_t s0 = soff[0] - b0, s1 = b1 - b0;
                        // four positions per step from two aligned LDS words, compared against the
                        // literal's first (up to) four bytes at once: the slow path (rest of the
                        // literal, row lookup) runs only on real prefix hits, so waves rarely
                        // diverge into it
                        unsigned lit4 = 0;
                        for (int c = 0; c < 4 && c < L; c++) lit4 |= (unsigned)(unsigned char)spat[1 + c] << (8 * c);
                        const unsigned m4 = L >= 4 ? 0xFFFFFFFFu : (1u << (8 * L)) - 1u;
                        const unsigned *sw = reinterpret_cast<const unsigned *>(sbuf);
                        const int nword = (int)((s1 + 3) / 4);
                        for (int j = threadIdx.x; j < nword; j += 256) {
                            const unsigned long long win =
                                (unsigned long long)sw[j] | ((unsigned long long)(j + 1 < SEL_STR_LDS / 4 ? sw[j + 1] : 0u) << 32);
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if ((((unsigned)(win >> (8 * k)) ^ lit4) & m4) != 0) continue;
                                const int64_t x = (int64_t)j * 4 + k;
                                if (x < s0 || x + L > s1) continue;
                                bool m = true;
                                for (int c = 4; m && c < L; c++) m = sbuf[x + c] == spat[1 + c];
                                if (!m) continue;
                                int lo = 0, hi = nrow;   // last row with offset <= position
                                while (hi - lo > 1) {
                                    int mid = (lo + hi) >> 1;
                                    if (soff[mid] - b0 <= x) lo = mid; else hi = mid;
                                }
                                if (x + L <= soff[lo + 1] - b0) atomicOr(&rowhit[lo >> 5], 1u << (lo & 31));
                            }
                        }
                        __syncthreads();
                        if (i < n_in && bit_valid(P.validity, i)) {
                            bool hit = (rowhit[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1;
                            pass = hit == (P.op == PH_LIKE);
                        }
                    } else if (i < n_in && bit_valid(P.validity, i)) {
                        const char *str = sbuf + (soff[threadIdx.x] - b0);
                        const int slen = soff[threadIdx.x + 1] - soff[threadIdx.x];
                        bool m;
                        if (P.op == PH_LIKE || P.op == PH_NOTLIKE) m = like_match(str, slen, spat, P.plen) == (P.op == PH_LIKE);
                        else {
                            bool eq = slen == P.plen;
                            for (int c = 0; eq && c < slen; c++) eq = str[c] == spat[c];
                            m = P.op == PH_EQ ? eq : !eq;
                        }
                        pass = m;
                    }
                }
            } else staged = true;
        }
        if (!staged && i < n_in) {
            int64_t row = sel_in ? sel_in[i] : i;
            pass = sel_pred(P, row);
        }
        cnt += pass ? 1 : 0;
        if (flags) {
            unsigned long long m = __ballot(pass);
            if ((threadIdx.x & 63) == 0) flags[((int64_t)blockIdx.x * SEL_ROUNDS + r) * 4 + (threadIdx.x >> 6)] = m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// %literal% over consecutive rows without staging the bytes: a workgroup owns 2048 rows, keeps only their
// offsets in LDS (8 KiB) and scans their byte range straight from global memory, 16 bytes per thread
// and step (every load independent of the others); a prefix hit verifies the rest of the literal and
// finds its row by binary search in the LDS offsets. Two barriers per 2048 rows — select_count_kernel's
// staged form pays five per 256 rows, three of them behind dependent global loads, and was latency
// bound (LIKE '%green%' over 2 M part names: 83 us for 66 MB).
__global__ __launch_bounds__(256) void like_contains_count_kernel(SelParams P, int64_t n_in, int32_t *__restrict__ block_counts,
                                                                  unsigned long long *__restrict__ flags) {
    __shared__ int soff[SEL_CHUNK + 1];
    __shared__ unsigned rowhit[SEL_CHUNK / 32];
    __shared__ char spat[96];
    const int64_t base = (int64_t)blockIdx.x * SEL_CHUNK;
    const int nrow = (int)(base + SEL_CHUNK < n_in ? SEL_CHUNK : n_in - base);
    const int32_t *off = (const int32_t *)P.data;
    for (int e = threadIdx.x; e <= nrow; e += 256) soff[e] = off[base + e];
    if (threadIdx.x < SEL_CHUNK / 32) rowhit[threadIdx.x] = 0;
    if (threadIdx.x < 96) spat[threadIdx.x] = P.pat[threadIdx.x];
    __syncthreads();
    const int L = P.plen - 2;
    const int64_t s0 = soff[0], s1 = soff[nrow];
    unsigned lit4 = 0;
    for (int c = 0; c < 4 && c < L; c++) lit4 |= (unsigned)(unsigned char)spat[1 + c] << (8 * c);
    const unsigned m4 = L >= 4 ? 0xFFFFFFFFu : (1u << (8 * L)) - 1u;
    const int64_t a0 = s0 & ~15ll;
    for (int64_t p = a0 + (int64_t)threadIdx.x * 16; p < s1; p += 256 * 16) {
        unsigned w[5] = {0, 0, 0, 0, 0};
        if (p + 20 <= s1) {   // the window and its 4-byte overlap lie inside this block's bytes
            const uint4 v = *reinterpret_cast<const uint4 *>(P.bytes + p);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            w[4] = *reinterpret_cast<const unsigned *>(P.bytes + p + 16);
        } else {              // tail of the block's range: byte by byte, nothing is read past s1
            for (int b = 0; b < 20 && p + b < s1; b++) w[b >> 2] |= (unsigned)(unsigned char)P.bytes[p + b] << (8 * (b & 3));
        }
        unsigned hits = 0;   // branch-free prefix test of the 16 positions first: ONE copy of the (rare) hit path below
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned long long two = (unsigned long long)w[k >> 2] | ((unsigned long long)w[(k >> 2) + 1] << 32);
            hits |= ((((unsigned)(two >> (8 * (k & 3))) ^ lit4) & m4) == 0 ? 1u : 0u) << k;
        }
        while (hits) {
            const int k = __ffs(hits) - 1;
            hits &= hits - 1;
            const int64_t x = p + k;
            if (x < s0 || x + L > s1) continue;
            bool m = true;
            for (int c = 4; m && c < L; c++) m = P.bytes[x + c] == spat[1 + c];
            if (!m) continue;
            int lo = 0, hi = nrow;   // last row with offset <= position
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (soff[mid] <= x) lo = mid; else hi = mid;
            }
            if (x + L <= soff[lo + 1]) atomicOr(&rowhit[lo >> 5], 1u << (lo & 31));
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int r = 0; r < SEL_ROUNDS; r++) {
        const int lr = r * 256 + threadIdx.x;
        const int64_t i = base + lr;
        bool pass = false;
        if (i < n_in && bit_valid(P.validity, i)) pass = (((rowhit[lr >> 5] >> (lr & 31)) & 1) != 0) == (P.op == PH_LIKE);
        cnt += pass ? 1 : 0;
        const unsigned long long m = __ballot(pass);
        if ((threadIdx.x & 63) == 0) flags[((int64_t)blockIdx.x * SEL_ROUNDS + r) * 4 + (threadIdx.x >> 6)] = m;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// %A%B% over consecutive rows, like like_contains_count_kernel: the row matches when SOME occurrence of A ends at or before the start of SOME
// occurrence of B, i.e. the earliest end of A <= the latest start of B — two words of LDS per row (atomicMin / atomicMax of the hits), no
// per-row scan (Q13's o_comment NOT LIKE '%special%requests%': 15 M rows, 730 MB; the row-per-thread matcher ran 2.0 ms at SF10).
__global__ __launch_bounds__(256) void like_contains2_count_kernel(SelParams P, int64_t n_in, int32_t *__restrict__ block_counts,
                                                                   unsigned long long *__restrict__ flags) {
    __shared__ int soff[SEL_CHUNK + 1];
    __shared__ int endA[SEL_CHUNK], startB[SEL_CHUNK];
    __shared__ char spat[96];
    const int64_t base = (int64_t)blockIdx.x * SEL_CHUNK;
    const int nrow = (int)(base + SEL_CHUNK < n_in ? SEL_CHUNK : n_in - base);
    const int32_t *off = (const int32_t *)P.data;
    for (int e = threadIdx.x; e <= nrow; e += 256) soff[e] = off[base + e];
    for (int e = threadIdx.x; e < SEL_CHUNK; e += 256) { endA[e] = 0x7fffffff; startB[e] = -1; }
    if (threadIdx.x < 96) spat[threadIdx.x] = P.pat[threadIdx.x];
    __syncthreads();
    const int LA = P.lit1_len, LB = P.lit2_len, atB = P.lit2_at;
    const int64_t s0 = soff[0], s1 = soff[nrow];
    unsigned litA = 0, litB = 0;
    for (int c = 0; c < 4 && c < LA; c++) litA |= (unsigned)(unsigned char)spat[1 + c] << (8 * c);
    for (int c = 0; c < 4 && c < LB; c++) litB |= (unsigned)(unsigned char)spat[atB + c] << (8 * c);
    const unsigned mA = LA >= 4 ? 0xFFFFFFFFu : (1u << (8 * LA)) - 1u, mB = LB >= 4 ? 0xFFFFFFFFu : (1u << (8 * LB)) - 1u;
    const int64_t a0 = s0 & ~15ll;
    for (int64_t p = a0 + (int64_t)threadIdx.x * 16; p < s1; p += 256 * 16) {
        unsigned w[5] = {0, 0, 0, 0, 0};
        if (p + 20 <= s1) {
            const uint4 v = *reinterpret_cast<const uint4 *>(P.bytes + p);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            w[4] = *reinterpret_cast<const unsigned *>(P.bytes + p + 16);
        } else {
            for (int b = 0; b < 20 && p + b < s1; b++) w[b >> 2] |= (unsigned)(unsigned char)P.bytes[p + b] << (8 * (b & 3));
        }
        unsigned hits = 0;   // bit k: a prefix of A at position k, bit 16 + k: of B
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned long long two = (unsigned long long)w[k >> 2] | ((unsigned long long)w[(k >> 2) + 1] << 32);
            const unsigned four = (unsigned)(two >> (8 * (k & 3)));
            hits |= (((four ^ litA) & mA) == 0 ? 1u : 0u) << k;
            hits |= (((four ^ litB) & mB) == 0 ? 1u : 0u) << (16 + k);
        }
        while (hits) {
            const int h = __ffs(hits) - 1;
            hits &= hits - 1;
            const bool isB = h >= 16;
            const int L = isB ? LB : LA, at = isB ? atB : 1;
            const int64_t x = p + (h & 15);
            if (x < s0 || x + L > s1) continue;
            bool m = true;
            for (int c = 4; m && c < L; c++) m = P.bytes[x + c] == spat[at + c];
            if (!m) continue;
            int lo = 0, hi = nrow;   // last row with offset <= position
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (soff[mid] <= x) lo = mid; else hi = mid;
            }
            if (x + L > soff[lo + 1]) continue;   // straddles two rows
            const int rel = (int)(x - soff[lo]);
            if (isB) atomicMax(&startB[lo], rel); else atomicMin(&endA[lo], rel + L);
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int r = 0; r < SEL_ROUNDS; r++) {
        const int lr = r * 256 + threadIdx.x;
        const int64_t i = base + lr;
        bool pass = false;
        if (i < n_in && bit_valid(P.validity, i)) pass = (endA[lr] <= startB[lr]) == (P.op == PH_LIKE);
        cnt += pass ? 1 : 0;
        const unsigned long long m = __ballot(pass);
        if ((threadIdx.x & 63) == 0) flags[((int64_t)blockIdx.x * SEL_ROUNDS + r) * 4 + (threadIdx.x >> 6)] = m;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__device__ __forceinline__ void scan_publish(const ScanPublish &S, long long total) {   // one thread: the word, then the number
    if (!S.mbox) return;
    S.mbox[0] = (unsigned long long)total;
    if (S.deferred) for (int k = 0; k < 4; k++) S.mbox_deferred[k] = S.deferred[k];
    __threadfence_system();
    __hip_atomic_store(S.flag, S.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// exclusive scan of n ints in place, total to out_total; one workgroup
__global__ __launch_bounds__(1024) void scan_kernel(int32_t *__restrict__ v, int64_t n,
                                                    int64_t *__restrict__ out_total, ScanPublish S) {
    __shared__ long long wsum[16];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        long long x = i < n ? v[i] : 0;
        long long incl = x;
        for (int o = 1; o < 64; o <<= 1) {
            long long y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < w; k++) woff += wsum[k];
        long long c = carry;
        if (i < n) v[i] = (int32_t)(c + woff + incl - x);
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out_total = carry; scan_publish(S, carry); }
}

// the same for n <= 16384 in ONE step: a thread owns 16 consecutive elements (the loop above pays three
// barriers per 1024 elements: 12 us for the 7 324 block totals of a 15 M-row probe, 3 us here)
__global__ __launch_bounds__(1024) void scan_small_kernel(int32_t *__restrict__ v, int n, int64_t *__restrict__ out_total, ScanPublish S) {
    constexpr int E = 16;
    __shared__ long long wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = threadIdx.x * E;
    int x[E];
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = i0 + e < n ? v[i0 + e] : 0;
    long long mine = 0;
#pragma unroll
    for (int e = 0; e < E; e++) mine += x[e];
    long long incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        long long y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    long long run = incl - mine;
    for (int k = 0; k < w; k++) run += wsum[k];
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (i0 + e < n) v[i0 + e] = (int32_t)run;
        run += x[e];
    }
    if (threadIdx.x == 1023) { *out_total = run; scan_publish(S, run); }
}

__global__ __launch_bounds__(256) void select_write_kernel(SelParams P, const int32_t *__restrict__ sel_in,
                                                           int64_t n_in, const int32_t *__restrict__ block_off,
                                                           int32_t *__restrict__ sel_out,
                                                           const unsigned long long *__restrict__ flags) {
    int64_t base = (int64_t)blockIdx.x * SEL_CHUNK;
    __shared__ int ws[4];
    int running = block_off[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < SEL_ROUNDS; r++) {
        int64_t i = base + r * 256 + threadIdx.x;
        int64_t row = 0;
        bool pass = false;
        unsigned long long m;
        if (flags) {
            m = flags[((int64_t)blockIdx.x * SEL_ROUNDS + r) * 4 + w];
            pass = (m >> lane) & 1;
            if (pass) row = sel_in ? sel_in[i] : i;
        } else {
            if (i < n_in) {
                row = sel_in ? sel_in[i] : i;
                pass = sel_pred(P, row);
            }
            m = __ballot(pass);
        }
        int rank = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) ws[w] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; k++) woff += ws[k];
        int total = ws[0] + ws[1] + ws[2] + ws[3];
        if (pass) sel_out[running + woff + rank] = (int32_t)row;
        running += total;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ vectorised range path
// Identity selection + integer range predicate (the common pushed-down conjunct: dates, quantities,
// decimals lowered to integers, dictionary codes): each lane loads 4 consecutive values with one
// 16-byte (int32) or two 16-byte (int64) loads, so the column streams at HBM rate; ranks come from a
// wave scan of the per-lane pass counts. Same count -> scan -> write structure, same ordered output.
constexpr int VSEL_ROUNDS = 4;
constexpr int VSEL_CHUNK = 256 * 4 * VSEL_ROUNDS;  // 4096 rows per workgroup

template <typename T> struct Vec4 { T v[4]; };

template <typename T> __device__ __forceinline__ Vec4<T> load4(const T *p);
template <> __device__ __forceinline__ Vec4<int32_t> load4(const int32_t *p) {
    int4 x = *reinterpret_cast<const int4 *>(p);
    return Vec4<int32_t>{{x.x, x.y, x.z, x.w}};
}
template <> __device__ __forceinline__ Vec4<int64_t> load4(const int64_t *p) {
    longlong2 a = *reinterpret_cast<const longlong2 *>(p), b = *reinterpret_cast<const longlong2 *>(p + 2);
    return Vec4<int64_t>{{a.x, a.y, b.x, b.y}};
}
template <> __device__ __forceinline__ Vec4<uint8_t> load4(const uint8_t *p) {
    unsigned x = *reinterpret_cast<const unsigned *>(p);
    return Vec4<uint8_t>{{(uint8_t)x, (uint8_t)(x >> 8), (uint8_t)(x >> 16), (uint8_t)(x >> 24)}};
}

template <typename T>
__device__ __forceinline__ unsigned range_mask4(const T *data, const uint8_t *validity, long long lo, long long hi,
                                                int64_t row, int64_t n) {
    // rows [row, row+4); the column allocation is padded, only rows < n may pass
    unsigned m = 0;
    if (row + 3 < n) {
        Vec4<T> x = load4<T>(data + row);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            long long v = (long long)x.v[j];
            if (v >= lo && v <= hi && bit_valid(validity, row + j)) m |= 1u << j;
        }
    } else {
        for (int j = 0; j < 4 && row + j < n; j++) {
            long long v = (long long)data[row + j];
            if (v >= lo && v <= hi && bit_valid(validity, row + j)) m |= 1u << j;
        }
    }
    return m;
}

template <typename T>
__global__ __launch_bounds__(256) void vsel_count_kernel(const T *__restrict__ data, const uint8_t *__restrict__ validity,
                                                         long long lo, long long hi, int64_t n,
                                                         int32_t *__restrict__ block_counts) {
    int64_t base = (int64_t)blockIdx.x * VSEL_CHUNK;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < VSEL_ROUNDS; r++) {
        int64_t row = base + (int64_t)r * 1024 + threadIdx.x * 4;
        if (row < n) cnt += __popc(range_mask4<T>(data, validity, lo, hi, row, n));
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

template <typename T>
__global__ __launch_bounds__(256) void vsel_write_kernel(const T *__restrict__ data, const uint8_t *__restrict__ validity,
                                                         long long lo, long long hi, int64_t n,
                                                         const int32_t *__restrict__ block_off,
                                                         int32_t *__restrict__ sel_out) {
    int64_t base = (int64_t)blockIdx.x * VSEL_CHUNK;
    __shared__ int ws[VSEL_ROUNDS][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned masks[VSEL_ROUNDS];
    int incl[VSEL_ROUNDS];
#pragma unroll
    for (int r = 0; r < VSEL_ROUNDS; r++) {
        int64_t row = base + (int64_t)r * 1024 + threadIdx.x * 4;
        masks[r] = row < n ? range_mask4<T>(data, validity, lo, hi, row, n) : 0u;
        int c = __popc(masks[r]);
        int x = c;
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        incl[r] = x;
        if (lane == 63) ws[r][w] = x;
    }
    __syncthreads();
    int running = block_off[blockIdx.x];
#pragma unroll
    for (int r = 0; r < VSEL_ROUNDS; r++) {
        int woff = 0;
        for (int k = 0; k < w; k++) woff += ws[r][k];
        int pos = running + woff + incl[r] - __popc(masks[r]);
        int64_t row = base + (int64_t)r * 1024 + threadIdx.x * 4;
        unsigned m = masks[r];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (m & (1u << j)) sel_out[pos++] = (int32_t)(row + j);
        running += ws[r][0] + ws[r][1] + ws[r][2] + ws[r][3];
    }
}

template <typename T>
static int run_vsel(ph_ctx *ctx, const SelParams &P, int64_t n, int32_t *sel_out, int32_t *counts, int64_t nb,
                    int64_t *total, const ScanPublish *pub) {
    vsel_count_kernel<T><<<(int)nb, 256, 0, ctx->stream>>>((const T *)P.data, P.validity, P.lo, P.hi, n, counts);
    PH_HIP(hipGetLastError());
    PH_CHECK(exclusive_scan_i32(ctx, counts, nb, total, pub));
    vsel_write_kernel<T><<<(int)nb, 256, 0, ctx->stream>>>((const T *)P.data, P.validity, P.lo, P.hi, n, counts, sel_out);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ host side

static bool lower_select(const ph_col *col, int32_t op, const ph_const *k, SelParams *P) {
    // which (type, op) pairs exist: selectOperation (function_operator_boolean.go:393-504)
    memset(P, 0, sizeof *P);
    P->kind = SK_NEVER;
    P->op = op;
    P->data = col->data;
    P->validity = col->validity;
    P->lo = INT64_MIN;
    P->hi = INT64_MAX;
    auto range = [&](long long v) {
        switch (op) {
        case PH_EQ: P->lo = P->hi = v; return true;
        case PH_LT: if (v == INT64_MIN) return false; P->hi = v - 1; return true;
        case PH_LE: P->hi = v; return true;
        case PH_GT: if (v == INT64_MAX) return false; P->lo = v + 1; return true;
        case PH_GE: P->lo = v; return true;
        default: return false;
        }
    };
    switch (col->type) {
    case PH_I32:
        if (k->type != PH_I32) return false;
        if (op == PH_NE) { P->kind = SK_NE_I32; P->lo = (int32_t)k->i; return true; }
        if (op >= PH_EQ && op <= PH_GE && range((int32_t)k->i)) P->kind = SK_RANGE_I32;
        return true;
    case PH_DATE:
        if (k->type != PH_DATE) return false;
        // DATE has <,<=,>,>= only ('=' falls in the default: return 0 branch)
        if (op >= PH_LT && op <= PH_GE && range((int32_t)k->i)) P->kind = SK_RANGE_I32;
        return true;
    case PH_I64:
        return true;  // no BIGINT comparison is implemented: selects nothing
    case PH_CODE8:
        // VARCHAR '=' / '!=' on a dictionary column: the caller resolves the literal to its
        // code (PH_I32 constant; a code outside 0..255 = literal not in the dictionary)
        // ... or to a RUN of codes (k.type = PH_CODE8, codes k.i .. k.scale): `LIKE 'prefix%'` or a sorted IN list over a
        // dictionary in byte order is a run of codes; '=' selects the codes inside it
        if (k->type == PH_CODE8) {
            if (op == PH_EQ && k->i <= k->scale && k->scale >= 0 && k->i <= 255) { P->lo = k->i < 0 ? 0 : k->i; P->hi = k->scale > 255 ? 255 : k->scale; P->kind = SK_RANGE_U8; }
            return op == PH_EQ;
        }
        if (k->type != PH_I32) return false;
        if (op == PH_EQ) { if (k->i >= 0 && k->i <= 255) { P->lo = P->hi = k->i; P->kind = SK_RANGE_U8; } return true; }
        if (op == PH_NE) { if (k->i >= 0 && k->i <= 255) { P->lo = k->i; P->kind = SK_NE_U8; } else { P->lo = 0; P->hi = 255; P->kind = SK_RANGE_U8; } return true; }
        return true;
    case PH_DEC64:
        if (k->type == PH_F32) {
            if (op == PH_GT || op == PH_GE || op == PH_LE) {  // FLOAT has no '<', '=' or '!='
                P->kind = SK_F32_DEC;
                P->kf = (float)k->f;
                P->div = 1;
                for (int i = 0; i < col->scale; i++) P->div *= 10;
            }
            return true;
        }
        if (k->type == PH_DEC64) {
            if (op != PH_GT) return true;  // only DECIMAL '>' exists
            if (k->scale > col->scale) return false;
            long long v = k->i;
            for (int i = k->scale; i < col->scale; i++)
                if (__builtin_mul_overflow(v, 10ll, &v)) return false;
            if (range(v)) P->kind = SK_RANGE_I64;
            return true;
        }
        return false;
    case PH_F32:
        if (k->type != PH_F32) return false;
        if (op == PH_GT || op == PH_GE || op == PH_LE) { P->kind = SK_F32; P->kf = (float)k->f; }
        return true;
    case PH_F64:
        if (k->type != PH_F64 && k->type != PH_F32) return false;
        if (op == PH_LT) { P->kind = SK_F64; P->kd = k->f; }
        return true;
    case PH_STR:
        if (k->type != PH_STR || !k->s) return false;
        if (op != PH_EQ && op != PH_NE && op != PH_LIKE && op != PH_NOTLIKE) return true;
        P->plen = (int)strlen(k->s);
        if (P->plen >= (int)sizeof P->pat) return false;
        memcpy(P->pat, k->s, (size_t)P->plen);
        if ((op == PH_LIKE || op == PH_NOTLIKE) && P->plen >= 3 && P->pat[0] == '%' && P->pat[P->plen - 1] == '%') {
            P->contains = 1;
            int inner = 0, at = 0;
            bool plain = true;
            for (int c = 1; c < P->plen - 1; c++) {
                if (P->pat[c] == '%' || P->pat[c] == '_') P->contains = 0;
                if (P->pat[c] == '%') { inner++; at = c; }
                if (P->pat[c] == '_' || P->pat[c] == '\\') plain = false;
            }
            // %A%B%: A somewhere, B somewhere behind it (wildcardMatch's % = any run of characters, function_operator_like.go)
            if (plain && inner == 1 && at > 1 && at < P->plen - 2) { P->lit1_len = at - 1; P->lit2_at = at + 1; P->lit2_len = P->plen - 1 - (at + 1); }
        }
        P->bytes = (const char *)col->aux;
        P->kind = SK_STR;
        return true;
    default:
        return false;
    }
}

bool lower_range_pred(const ph_col *col, int32_t op, const ph_const *k, RangePred *out) {
    SelParams P;
    if (!lower_select(col, op, k, &P)) return false;
    out->data = P.data;
    out->validity = P.validity;
    out->lo = P.lo;
    out->hi = P.hi;
    switch (P.kind) {
    case SK_NEVER: out->kind = -1; return true;
    case SK_RANGE_I32: out->kind = 1; return true;
    case SK_RANGE_I64: out->kind = 2; return true;
    case SK_RANGE_U8: out->kind = 3; return true;
    default: return false;
    }
}

// Larger inputs: tiles of 4096 scanned by one workgroup each (in place, exclusive within the
// tile, tile total aside), the tile totals scanned by scan_kernel, then added back. The single
// workgroup loop above costs ~1.5-2.5 ns per element (two barriers per 1024 elements).
constexpr int SCAN_TILE = 4096;

__global__ __launch_bounds__(1024) void scan_tile_kernel(int32_t *__restrict__ v, int64_t n, int32_t *__restrict__ tile_sum) {
    __shared__ int wsum[16];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 4;
    int x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = base + j < n ? v[base + j] : 0;
    const int mine = x[0] + x[1] + x[2] + x[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = incl - mine;
    for (int k = 0; k < w; k++) off += wsum[k];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) v[base + j] = off;
        off += x[j];
    }
    if (threadIdx.x == 1023) tile_sum[blockIdx.x] = off;
}

__global__ __launch_bounds__(1024) void scan_add_kernel(int32_t *__restrict__ v, int64_t n, const int32_t *__restrict__ tile_off) {
    const int add = tile_off[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 4;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (base + j < n) v[base + j] += add;
}

// Single-pass form (decoupled look-back): ONE launch for any n. A workgroup takes the next tile
// from a ticket counter (tiles start in order, so a tile only ever waits for tiles already running),
// scans its 4096 elements, publishes its aggregate, walks back over the predecessors' published
// aggregates / inclusive prefixes, publishes its own inclusive prefix and writes its elements.
// The tile states live in a per-context buffer that is never cleared: every entry carries the
// epoch of the call that wrote it ((epoch << 34) | (flag << 32) | value), stale entries read as
// "not yet published", and the ticket counter keeps counting across calls (the host passes its
// base). A scan of Q3's 29 k candidate-block counts was three launches (tile sums, their scan,
// add-back: ~15 us); row counts stay below 2^31, so 32 value bits are enough.
constexpr unsigned long long SC_AGG = 1ull, SC_INCL = 2ull;

__global__ __launch_bounds__(1024) void scan_lookback_kernel(int32_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ state,
                                                             unsigned *__restrict__ ticket, unsigned ticket_base,
                                                             unsigned long long epoch, int64_t *__restrict__ total, ScanPublish S) {
    __shared__ int wsum[16];
    __shared__ unsigned s_tile;
    __shared__ long long s_prefix;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const unsigned tile = s_tile;
    const int64_t base = (int64_t)tile * SCAN_TILE + (int64_t)threadIdx.x * 4;
    int x[4];
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = base + j < n ? v[base + j] : 0;
    const int mine = x[0] + x[1] + x[2] + x[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = incl - mine, agg = 0;
    for (int k = 0; k < 16; k++) {
        if (k < w) off += wsum[k];
        agg += wsum[k];
    }
    if (threadIdx.x == 0) {
        const unsigned long long tag = epoch << 34;
        long long prefix = 0;
        if (tile == 0) {
            __hip_atomic_store(&state[0], tag | (SC_INCL << 32) | (unsigned)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(&state[tile], tag | (SC_AGG << 32) | (unsigned)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            for (long long p = (long long)tile - 1; p >= 0;) {
                const unsigned long long st = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long flag = (st >> 32) & 3ull;
                if ((st >> 34) != epoch || flag == 0) {          // not published in this call yet: look again
                    if (++spins > (1u << 27)) break;             // never reached in practice: every wave must terminate
                    continue;
                }
                prefix += (long long)(unsigned)st;
                if (flag == SC_INCL) break;
                p--;
            }
            __hip_atomic_store(&state[tile], tag | (SC_INCL << 32) | (unsigned)(prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_prefix = prefix;
        if ((int64_t)(tile + 1) * SCAN_TILE >= n) { *total = prefix + agg; scan_publish(S, prefix + agg); }   // the last tile
    }
    __syncthreads();
    off += (int)s_prefix;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) v[base + j] = off;
        off += x[j];
    }
}

int scan_state_acquire(ph_ctx *ctx, int64_t nt, unsigned long long **state, unsigned **ticket, unsigned *ticket_base, unsigned long long *epoch) {
    if (nt > ctx->scan_tiles) {   // (re)allocate the state buffer: zero = epoch 0, which no call uses
        if (ctx->scan_state) { PH_HIP(hipStreamSynchronize(ctx->stream)); PH_HIP(hipFree(ctx->scan_state)); ctx->scan_state = nullptr; }
        const int64_t cap = std::max<int64_t>(nt * 2, 4096);
        PH_HIP(hipMalloc(&ctx->scan_state, (size_t)cap * 8 + 64));
        PH_HIP(hipMemsetAsync(ctx->scan_state, 0, (size_t)cap * 8 + 64, ctx->stream));
        ctx->scan_tiles = cap;
        ctx->scan_ticket_base = 0;
    }
    *state = (unsigned long long *)ctx->scan_state;
    *ticket = (unsigned *)(*state + ctx->scan_tiles);
    ctx->scan_epoch = (ctx->scan_epoch % ((1ull << 30) - 1)) + 1;   // 1 .. 2^30-1, never 0
    *epoch = ctx->scan_epoch;
    *ticket_base = ctx->scan_ticket_base;
    ctx->scan_ticket_base += (unsigned)nt;   // wraps like the device counter
    return PH_OK;
}

int exclusive_scan_i32(ph_ctx *ctx, int32_t *dev, int64_t n, int64_t *total_dev, const ScanPublish *pub) {
    const ScanPublish S = pub ? *pub : ScanPublish{};
    if (n <= 4 * SCAN_TILE) {
        if (n > 1024) scan_small_kernel<<<1, 1024, 0, ctx->stream>>>(dev, (int)n, total_dev, S);
        else scan_kernel<<<1, 1024, 0, ctx->stream>>>(dev, n, total_dev, S);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    const int64_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    static const bool three_pass = getenv("PH_SCAN_THREE_PASS") != nullptr;
    if (!three_pass && nt < (1ll << 31)) {
        unsigned long long *state = nullptr, epoch = 0;
        unsigned *ticket = nullptr, ticket_base = 0;
        PH_CHECK(scan_state_acquire(ctx, nt, &state, &ticket, &ticket_base, &epoch));
        scan_lookback_kernel<<<(int)nt, 1024, 0, ctx->stream>>>(dev, n, state, ticket, ticket_base, epoch, total_dev, S);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    int32_t *tiles = nullptr;
    PH_CHECK(ctx->pool_alloc(nt * 4, (void **)&tiles));
    scan_tile_kernel<<<(int)nt, 1024, 0, ctx->stream>>>(dev, n, tiles);
    int rc = exclusive_scan_i32(ctx, tiles, nt, total_dev, pub);
    scan_add_kernel<<<(int)nt, 1024, 0, ctx->stream>>>(dev, n, tiles);
    if (rc == PH_OK && hipGetLastError() != hipSuccess) rc = PH_EHIP;
    ctx->pool_release(tiles);
    return rc;
}

}  // namespace ph

static int run_select(ph_ctx *ctx, ph::SelParams &P, const int32_t *sel_in, int64_t n_in, int32_t *sel_out, int64_t *n_out);

extern "C" int ph_filter_select(ph_ctx *ctx, const ph_col *col, int64_t n, int32_t op,
                                const ph_const *k, const int32_t *sel_in, int64_t n_in,
                                int32_t *sel_out, int64_t *n_out) {
    PH_REQUIRE(ctx && col && k && n_out && n >= 0 && n_in >= 0, "ph_filter_select: bad arguments");
    PH_REQUIRE(sel_in || n_in == n, "ph_filter_select: without sel_in, n_in must equal n");
    PH_REQUIRE(n_in == 0 || sel_out, "ph_filter_select: sel_out is NULL");
    *n_out = 0;
    if (n_in == 0) return PH_OK;
    ph::SelParams P;
    if (!ph::lower_select(col, op, k, &P)) {
        ph::set_error("ph_filter_select: column type %d with constant type %d is outside the device path", col->type, k->type);
        return PH_EUNSUPPORTED;
    }
    return run_select(ctx, P, sel_in, n_in, sel_out, n_out);
}

// Two conjuncts over ONE column in one pass (l_shipdate >= a AND l_shipdate < b: execSelectAnd runs the second over the first's selection,
// expr_exec.go:430-486 — two passes and a gather): both lower to a value range, the ranges intersect. PH_EUNSUPPORTED when either is no range
// (a string, a float, '!='): the caller runs them one after the other.
extern "C" int ph_filter_select_and(ph_ctx *ctx, const ph_col *col, int64_t n, int32_t op1, const ph_const *k1, int32_t op2, const ph_const *k2,
                                    const int32_t *sel_in, int64_t n_in, int32_t *sel_out, int64_t *n_out) {
    PH_REQUIRE(ctx && col && k1 && k2 && n_out && n >= 0 && n_in >= 0, "ph_filter_select_and: bad arguments");
    PH_REQUIRE(sel_in || n_in == n, "ph_filter_select_and: without sel_in, n_in must equal n");
    PH_REQUIRE(n_in == 0 || sel_out, "ph_filter_select_and: sel_out is NULL");
    *n_out = 0;
    if (n_in == 0) return PH_OK;
    ph::SelParams P, Q;
    if (!ph::lower_select(col, op1, k1, &P) || !ph::lower_select(col, op2, k2, &Q)) {
        ph::set_error("ph_filter_select_and: column type %d with these constants is outside the device path", col->type);
        return PH_EUNSUPPORTED;
    }
    if (P.kind == ph::SK_NEVER || Q.kind == ph::SK_NEVER) return PH_OK;   // a (type, op) pair the reference does not implement selects nothing
    const bool range = P.kind == Q.kind && (P.kind == ph::SK_RANGE_I32 || P.kind == ph::SK_RANGE_I64 || P.kind == ph::SK_RANGE_U8);
    if (!range) { ph::set_error("ph_filter_select_and: the two conjuncts are not value ranges of one kind"); return PH_EUNSUPPORTED; }
    P.lo = std::max(P.lo, Q.lo);
    P.hi = std::min(P.hi, Q.hi);
    if (P.lo > P.hi) return PH_OK;
    return run_select(ctx, P, sel_in, n_in, sel_out, n_out);
}

// `col IN (v1, .., vk)` in one pass: InExpr is `in(a,x) OR in(a,y) ..` — execSelectOr (expr_exec.go:488-530) evaluates every child over the input
// and unites the selections: k passes, k counts, one union. INTEGER columns (up to 16 values) and dictionary codes (any set of codes: a bitmap) — '=' exists for both (selectOperation); a value outside
// the column's type matches nothing.
extern "C" int ph_filter_select_in(ph_ctx *ctx, const ph_col *col, int64_t n, const int64_t *values, int32_t nvalues, const int32_t *sel_in, int64_t n_in,
                                   int32_t *sel_out, int64_t *n_out) {
    PH_REQUIRE(ctx && col && n_out && n >= 0 && n_in >= 0 && nvalues >= 0 && (nvalues == 0 || values), "ph_filter_select_in: bad arguments");
    PH_REQUIRE(sel_in || n_in == n, "ph_filter_select_in: without sel_in, n_in must equal n");
    PH_REQUIRE(n_in == 0 || sel_out, "ph_filter_select_in: sel_out is NULL");
    *n_out = 0;
    if (col->type != PH_I32 && col->type != PH_CODE8) { ph::set_error("ph_filter_select_in: INTEGER or dictionary-code columns"); return PH_EUNSUPPORTED; }
    if (nvalues > (col->type == PH_CODE8 ? 256 : 16)) { ph::set_error("ph_filter_select_in: at most 16 INTEGER values / 256 codes"); return PH_EUNSUPPORTED; }
    if (n_in == 0 || nvalues == 0) return PH_OK;
    ph::SelParams P;
    memset(&P, 0, sizeof P);
    P.kind = col->type == PH_I32 ? ph::SK_IN_I32 : ph::SK_IN_U8;
    P.op = PH_EQ;
    P.data = col->data; P.validity = col->validity;
    for (int32_t q = 0; q < nvalues; q++) {
        const int64_t v = values[q];
        if (col->type == PH_I32 ? (v < INT32_MIN || v > INT32_MAX) : (v < 0 || v > 255)) continue;
        if (col->type == PH_I32) P.inv[P.nin++] = (int)v;
        else { P.inbits[v >> 5] |= 1u << (v & 31); P.nin++; }   // a set of codes: one bit each (any number of them)
    }
    if (P.nin == 0) return PH_OK;
    return run_select(ctx, P, sel_in, n_in, sel_out, n_out);
}

static int run_select(ph_ctx *ctx, ph::SelParams &P, const int32_t *sel_in, int64_t n_in, int32_t *sel_out, int64_t *n_out) {
    if (P.kind == ph::SK_NEVER) return PH_OK;
    // the 4-values-per-lane path needs 16-byte (uint8: 4-byte) aligned column data
    uintptr_t addr = (uintptr_t)P.data;
    bool aligned = P.kind == ph::SK_RANGE_U8 ? addr % 4 == 0 : addr % 16 == 0;
    bool vec = aligned && sel_in == nullptr &&
               (P.kind == ph::SK_RANGE_I32 || P.kind == ph::SK_RANGE_I64 || P.kind == ph::SK_RANGE_U8);
    int64_t chunk = vec ? ph::VSEL_CHUNK : ph::SEL_CHUNK;
    int64_t nb = (n_in + chunk - 1) / chunk;
    const bool memo = P.kind == ph::SK_STR;
    PH_CHECK(ctx->ensure_scratch(ph::round_up(nb * 4, 8) + 64 + (memo ? nb * ph::SEL_ROUNDS * 4 * 8 : 0)));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(nb * 4, 8));
    unsigned long long *flags = memo ? (unsigned long long *)((char *)ctx->scratch + ph::round_up(nb * 4, 8) + 64) : nullptr;
    // the count travels with the scan (ScanPublish): the host has it while the write pass runs, and no publish launch follows
    ph::ScanPublish pub;
    PH_CHECK(ctx->arm_count(&pub));
    const unsigned long long armed = pub.seq;
    auto count_back = [&]() -> int { return ctx->count_back(pub, n_out, total, -1, "ph_filter_select"); };
    if (vec) {
        int rc = P.kind == ph::SK_RANGE_I32   ? ph::run_vsel<int32_t>(ctx, P, n_in, sel_out, counts, nb, total, armed ? &pub : nullptr)
                 : P.kind == ph::SK_RANGE_I64 ? ph::run_vsel<int64_t>(ctx, P, n_in, sel_out, counts, nb, total, armed ? &pub : nullptr)
                                              : ph::run_vsel<uint8_t>(ctx, P, n_in, sel_out, counts, nb, total, armed ? &pub : nullptr);
        PH_CHECK(rc);
        return count_back();
    }
    static const bool no_direct_like = getenv("PH_LIKE_STAGED") != nullptr;   // the staged form, for the parity test
    if (P.kind == ph::SK_STR && P.contains && sel_in == nullptr && P.plen >= 3 && !no_direct_like &&
        (reinterpret_cast<uintptr_t>(P.bytes) & 15) == 0)
        ph::like_contains_count_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, n_in, counts, flags);
    else if (P.kind == ph::SK_STR && P.lit2_at > 0 && (P.op == PH_LIKE || P.op == PH_NOTLIKE) && sel_in == nullptr && memo && !no_direct_like &&
             (reinterpret_cast<uintptr_t>(P.bytes) & 15) == 0)
        ph::like_contains2_count_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, n_in, counts, flags);
    else
        ph::select_count_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, sel_in, n_in, counts, flags);
    PH_HIP(hipGetLastError());
    PH_CHECK(ph::exclusive_scan_i32(ctx, counts, nb, total, armed ? &pub : nullptr));
    ph::select_write_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, sel_in, n_in, counts, sel_out, flags);
    PH_HIP(hipGetLastError());
    return count_back();
}

// column OP column (selectBinary with two FLAT vectors, function_operator_boolean.go:506-521): the (type, op) pairs
// are selectOperation's — INTEGER has all six, DATE the four orderings, DECIMAL only '>'; anything else selects nothing
extern "C" int ph_filter_select_cols(ph_ctx *ctx, const ph_col *a, const ph_col *b, int64_t n, int32_t op, const int32_t *sel_in,
                                     int64_t n_in, int32_t *sel_out, int64_t *n_out) {
    PH_REQUIRE(ctx && a && b && n_out && n >= 0 && n_in >= 0, "ph_filter_select_cols: bad arguments");
    PH_REQUIRE(sel_in || n_in == n, "ph_filter_select_cols: without sel_in, n_in must equal n");
    PH_REQUIRE(n_in == 0 || sel_out, "ph_filter_select_cols: sel_out is NULL");
    *n_out = 0;
    if (n_in == 0) return PH_OK;
    ph::SelParams P;
    memset(&P, 0, sizeof P);
    P.kind = ph::SK_NEVER;
    P.op = op;
    P.data = a->data; P.validity = a->validity; P.data2 = b->data; P.validity2 = b->validity;
    const bool ordering = op == PH_LT || op == PH_LE || op == PH_GT || op == PH_GE;
    if (a->type == PH_I32 && b->type == PH_I32) { if (op >= PH_EQ && op <= PH_GE) P.kind = ph::SK_CMP2_I32; }
    else if (a->type == PH_DATE && b->type == PH_DATE) { if (ordering) P.kind = ph::SK_CMP2_I32; }
    else if (a->type == PH_DEC64 && b->type == PH_DEC64 && a->scale == b->scale) { if (op == PH_GT) P.kind = ph::SK_CMP2_I64; }
    else if ((a->type == PH_I64 && b->type == PH_I64) || (a->type == PH_CODE8 && b->type == PH_CODE8)) { /* no BIGINT comparison exists; dictionary codes of two columns are not comparable */ }
    else {
        ph::set_error("ph_filter_select_cols: column types %d and %d are outside the device path", a->type, b->type);
        return PH_EUNSUPPORTED;
    }
    if (P.kind == ph::SK_NEVER) return PH_OK;
    const int64_t nb = (n_in + ph::SEL_CHUNK - 1) / ph::SEL_CHUNK;
    PH_CHECK(ctx->ensure_scratch(ph::round_up(nb * 4, 8) + 64));
    int32_t *counts = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(nb * 4, 8));
    ph::ScanPublish pub;
    PH_CHECK(ctx->arm_count(&pub));
    ph::select_count_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, sel_in, n_in, counts, nullptr);
    PH_HIP(hipGetLastError());
    PH_CHECK(ph::exclusive_scan_i32(ctx, counts, nb, total, pub.seq ? &pub : nullptr));
    ph::select_write_kernel<<<(int)nb, 256, 0, ctx->stream>>>(P, sel_in, n_in, counts, sel_out, nullptr);
    PH_HIP(hipGetLastError());
    return ctx->count_back(pub, n_out, total, -1, "ph_filter_select_cols");
}

// ------------------------------------------------------------------ union (OR / IN lists)

namespace ph {
__global__ __launch_bounds__(256) void sel_flag_kernel(const int32_t *__restrict__ sel, int64_t n, uint8_t *__restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) flags[sel[i]] = 1;
}
}  // namespace ph

namespace ph {
__global__ __launch_bounds__(256) void iota_kernel(int32_t *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (int32_t)i;
}
}  // namespace ph

extern "C" int ph_dev_iota(ph_ctx *ctx, int32_t *out_dev, int64_t n) {
    PH_REQUIRE(ctx && n >= 0 && (n == 0 || out_dev) && n < (1ll << 31), "ph_dev_iota: bad arguments");
    if (n == 0) return PH_OK;
    ph::iota_kernel<<<(int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8), 256, 0, ctx->stream>>>(out_dev, n);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_sel_mark(ph_ctx *ctx, const int32_t *sel_dev, int64_t n, uint8_t *marks_dev) {
    PH_REQUIRE(ctx && n >= 0 && (n == 0 || (sel_dev && marks_dev)), "ph_sel_mark: bad arguments");
    if (n == 0) return PH_OK;
    ph::sel_flag_kernel<<<(int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8), 256, 0, ctx->stream>>>(sel_dev, n, marks_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_sel_union(ph_ctx *ctx, const int32_t *const *sels_dev, const int64_t *counts, int32_t k,
                            int64_t n_rows, int32_t *out_sel_dev, int64_t *n_out) {
    PH_REQUIRE(ctx && n_out && k >= 0 && n_rows >= 0 && (k == 0 || (sels_dev && counts)), "ph_sel_union: bad arguments");
    *n_out = 0;
    int64_t total = 0;
    for (int i = 0; i < k; i++) { PH_REQUIRE(counts[i] >= 0 && (counts[i] == 0 || sels_dev[i]), "ph_sel_union: bad selection %d", i); total += counts[i]; }
    if (total == 0 || n_rows == 0) return PH_OK;
    PH_REQUIRE(out_sel_dev, "ph_sel_union: out_sel_dev is NULL");
    // one flag byte per row, set by every child's rows, compacted by the vectorised byte filter
    uint8_t *flags = nullptr;
    PH_CHECK(ctx->pool_alloc(ph::round_up(n_rows, 16) + 16, (void **)&flags));
    int rc = PH_OK;
    if (hipMemsetAsync(flags, 0, (size_t)n_rows, ctx->stream) != hipSuccess) rc = PH_EHIP;
    for (int i = 0; i < k && rc == PH_OK; i++) {
        if (counts[i] == 0) continue;
        int grid = (int)std::min<int64_t>((counts[i] + 255) / 256, (int64_t)ctx->cu_count * 8);
        ph::sel_flag_kernel<<<grid, 256, 0, ctx->stream>>>(sels_dev[i], counts[i], flags);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    if (rc == PH_OK) {
        ph_col c{};
        c.type = PH_CODE8;
        c.data = flags;
        ph_const one{};
        one.type = PH_I32;
        one.i = 1;
        rc = ph_filter_select(ctx, &c, n_rows, PH_EQ, &one, nullptr, n_rows, out_sel_dev, n_out);
    }
    ctx->pool_release(flags);
    if (rc == PH_EHIP && ph_last_error()[0] == 0) ph::set_error("ph_sel_union: HIP failure");
    return rc;
}

// ------------------------------------------------------------------ CASE building blocks

namespace ph {
__global__ __launch_bounds__(256) void sel_set_kernel(const int32_t *__restrict__ sel, int64_t n, uint8_t *__restrict__ flags, uint8_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) flags[sel[i]] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_kernel(const T *__restrict__ vals, const uint8_t *__restrict__ vvalid,
                                                      const int32_t *__restrict__ sel, int64_t n, T *__restrict__ out,
                                                      unsigned *__restrict__ ovalid) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = sel ? sel[i] : i;
        const bool ok = bit_valid(vvalid, i);
        if (ok) out[r] = vals[i];
        if (ovalid && ok) atomicOr(&ovalid[r >> 5], 1u << (r & 31));
    }
}
}  // namespace ph

extern "C" int ph_sel_difference(ph_ctx *ctx, const int32_t *parent_dev, int64_t n_parent, const int32_t *child_dev,
                                 int64_t n_child, int64_t n_rows, int32_t *out_sel_dev, int64_t *n_out) {
    PH_REQUIRE(ctx && n_out && n_rows >= 0 && n_child >= 0 && (n_child == 0 || child_dev) && (parent_dev || n_parent == n_rows),
               "ph_sel_difference: bad arguments");
    *n_out = 0;
    if (n_rows == 0 || n_parent == 0) return PH_OK;
    PH_REQUIRE(out_sel_dev, "ph_sel_difference: out_sel_dev is NULL");
    uint8_t *flags = nullptr;
    PH_CHECK(ctx->pool_alloc(ph::round_up(n_rows, 16) + 16, (void **)&flags));
    int rc = PH_OK;
    auto grid = [&](int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8); };
    if (hipMemsetAsync(flags, parent_dev ? 0 : 1, (size_t)n_rows, ctx->stream) != hipSuccess) rc = PH_EHIP;
    if (rc == PH_OK && parent_dev) ph::sel_set_kernel<<<grid(n_parent), 256, 0, ctx->stream>>>(parent_dev, n_parent, flags, 1);
    if (rc == PH_OK && n_child) ph::sel_set_kernel<<<grid(n_child), 256, 0, ctx->stream>>>(child_dev, n_child, flags, 0);
    if (rc == PH_OK && hipGetLastError() != hipSuccess) rc = PH_EHIP;
    if (rc == PH_OK) {
        ph_col c{};
        c.type = PH_CODE8;
        c.data = flags;
        ph_const one{};
        one.type = PH_I32;
        one.i = 1;
        rc = ph_filter_select(ctx, &c, n_rows, PH_EQ, &one, nullptr, n_rows, out_sel_dev, n_out);
    }
    ctx->pool_release(flags);
    if (rc == PH_EHIP && ph_last_error()[0] == 0) ph::set_error("ph_sel_difference: HIP failure");
    return rc;
}

extern "C" int ph_scatter(ph_ctx *ctx, const ph_col *values, const int32_t *sel_dev, int64_t n, void *out_data_dev,
                          uint8_t *out_validity_dev) {
    PH_REQUIRE(ctx && values && n >= 0, "ph_scatter: bad arguments");
    if (n == 0) return PH_OK;
    PH_REQUIRE(values->data && out_data_dev, "ph_scatter: NULL buffer");
    PH_REQUIRE(((uintptr_t)out_validity_dev & 3) == 0, "ph_scatter: validity bitmap must be 4-byte aligned");
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
    switch (values->type) {
    case PH_I32:
        ph::scatter_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t *)values->data, values->validity, sel_dev, n,
                                                                   (int32_t *)out_data_dev, (unsigned *)out_validity_dev);
        break;
    case PH_DEC64:
        ph::scatter_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t *)values->data, values->validity, sel_dev, n,
                                                                   (int64_t *)out_data_dev, (unsigned *)out_validity_dev);
        break;
    default:
        ph::set_error("ph_scatter: value type %d (FillSwitch fills INTEGER and DECIMAL results only, expr_exec.go:565-572)", values->type);
        return PH_EUNSUPPORTED;
    }
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ gather

namespace ph {
// Eight elements per thread per step, stage by stage: the 8 index reads go out together, then the
// 8 dependent value reads, then the stores — a gather is two dependent memory latencies per element
// and one element per thread per grid-stride step left it latency bound (~1 TB/s on 13 MB sources).
template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ src, const int32_t *__restrict__ idx,
                                                     int64_t n, T *__restrict__ out) {
    constexpr int U = 8;
    for (int64_t base = (int64_t)blockIdx.x * 256 * U; base < n; base += (int64_t)gridDim.x * 256 * U) {
        int32_t ix[U];
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            ix[u] = idx[i < n ? i : 0];
            ix[u] = ix[u] < 0 ? 0 : ix[u];   // a negative row id (a strict lookup's miss, reported later) reads row 0
        }
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[ix[u]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i < n) out[i] = v[u];
        }
    }
}
}  // namespace ph

extern "C" int ph_gather(ph_ctx *ctx, const ph_col *col, const int32_t *idx_dev, int64_t n, void *out_dev) {
    PH_REQUIRE(ctx && col && (n == 0 || (idx_dev && out_dev)), "ph_gather: bad arguments");
    if (n == 0) return PH_OK;
    int grid = (int)std::min<int64_t>((n + 2047) / 2048, 256 * 16);
    switch (ph::type_width(col->type)) {
    case 1: ph::gather_kernel<uint8_t><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)col->data, idx_dev, n, (uint8_t *)out_dev); break;
    case 4: ph::gather_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t *)col->data, idx_dev, n, (int32_t *)out_dev); break;
    case 8: ph::gather_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t *)col->data, idx_dev, n, (int64_t *)out_dev); break;
    default: ph::set_error("ph_gather: column type %d is not fixed width", col->type); return PH_EUNSUPPORTED;
    }
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ---- validity of a row-id vector (the NULL-extended side of a LEFT OUTER join: row id -1 = no build row) and dictionary codes as INTEGERs
namespace ph {
__global__ __launch_bounds__(256) void rowid_validity_kernel(const int32_t *__restrict__ ids, int64_t n, uint8_t *__restrict__ bitmap) {
    // a lane composes the byte of eight consecutive rows: plain byte stores, no atomics
    const int64_t nbytes = (n + 7) / 8;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nbytes; b += (int64_t)gridDim.x * 256) {
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t i = b * 8 + k;
            if (i < n && ids[i] >= 0) v |= 1u << k;
        }
        bitmap[b] = (uint8_t)v;
    }
}
__global__ __launch_bounds__(256) void widen_codes_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ sel, int64_t n, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = sel ? (sel[i] < 0 ? 0 : sel[i]) : i;
        out[i] = (int32_t)codes[r];
    }
}
}  // namespace ph

extern "C" int ph_rowid_validity(ph_ctx *ctx, const int32_t *ids_dev, int64_t n, uint8_t *bitmap_dev) {
    PH_REQUIRE(ctx && n >= 0 && (n == 0 || (ids_dev && bitmap_dev)), "ph_rowid_validity: bad arguments");
    if (n == 0) return PH_OK;
    const int grid = (int)std::min<int64_t>(((n + 7) / 8 + 255) / 256, 256 * 8);
    ph::rowid_validity_kernel<<<grid, 256, 0, ctx->stream>>>(ids_dev, n, bitmap_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_widen_codes(ph_ctx *ctx, const ph_col *col, const int32_t *sel, int64_t n, int32_t *out_dev) {
    PH_REQUIRE(ctx && col && n >= 0 && (n == 0 || out_dev), "ph_widen_codes: bad arguments");
    PH_REQUIRE(col->type == PH_CODE8 && col->data, "ph_widen_codes: column type %d is not PH_CODE8", col->type);
    if (n == 0) return PH_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    ph::widen_codes_kernel<<<grid, 256, 0, ctx->stream>>>((const uint8_t *)col->data, sel, n, out_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// Several columns through ONE row-id array in one pass (late materialisation of a join chain's probe
// side: Q9 needs six lineitem columns at the rows that survived the first join). A gather is two
// dependent memory latencies per element; one launch per column pays them once per column, here
// the index read is shared and every column's value read of a row is in flight together.
namespace ph {
constexpr int GM_MAX = 8;
struct GatherMulti {
    const void *src[GM_MAX];
    void *dst[GM_MAX];
    int width[GM_MAX];
    int ncols;
};

template <int U>
__global__ __launch_bounds__(256) void gather_multi_kernel(GatherMulti G, const int32_t *__restrict__ idx, int64_t n) {
    for (int64_t base = (int64_t)blockIdx.x * 256 * U; base < n; base += (int64_t)gridDim.x * 256 * U) {
        int32_t ix[U];
        unsigned long long v[U][GM_MAX];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            ix[u] = idx[i < n ? i : 0];
            ix[u] = ix[u] < 0 ? 0 : ix[u];
        }
#pragma unroll
        for (int c = 0; c < GM_MAX; c++) {
            if (c >= G.ncols) break;   // wave-uniform
#pragma unroll
            for (int u = 0; u < U; u++)
                v[u][c] = G.width[c] == 8 ? ((const unsigned long long *)G.src[c])[ix[u]]
                          : G.width[c] == 4 ? (unsigned long long)((const uint32_t *)G.src[c])[ix[u]]
                                            : (unsigned long long)((const uint8_t *)G.src[c])[ix[u]];
        }
#pragma unroll
        for (int c = 0; c < GM_MAX; c++) {
            if (c >= G.ncols) break;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + u * 256 + threadIdx.x;
                if (i >= n) continue;
                if (G.width[c] == 8) ((unsigned long long *)G.dst[c])[i] = v[u][c];
                else if (G.width[c] == 4) ((uint32_t *)G.dst[c])[i] = (uint32_t)v[u][c];
                else ((uint8_t *)G.dst[c])[i] = (uint8_t)v[u][c];
            }
        }
    }
}
}  // namespace ph

namespace ph {
// the same gather out of a co-located group: one row of `stride` bytes holds every requested value
struct GatherGroup {
    const unsigned char *rows;
    int stride, ncols;
    int off[GM_MAX], width[GM_MAX];
    void *dst[GM_MAX];
};
template <int U>
__global__ __launch_bounds__(256) void gather_group_kernel(GatherGroup G, const int32_t *__restrict__ idx, int64_t n) {
    for (int64_t base = (int64_t)blockIdx.x * 256 * U; base < n; base += (int64_t)gridDim.x * 256 * U) {
        const unsigned char *row[U];
        unsigned long long v[U][GM_MAX];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            int ix = idx[i < n ? i : 0];
            ix = ix < 0 ? 0 : ix;               // (as ph_gather_multi: a negative row id reads row 0)
            row[u] = G.rows + (int64_t)ix * G.stride;
        }
        if (G.stride <= 32) {
            // the whole row in two 16-byte reads (one when it is 16 bytes), the values picked out of the registers: two requests
            // per row id where a read per column is one request each — the request path, not the bytes, bounds this kernel
            unsigned d[U][8];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint4 lo = *reinterpret_cast<const uint4 *>(row[u]);
                uint4 hi = make_uint4(0, 0, 0, 0);
                if (G.stride == 32) hi = *reinterpret_cast<const uint4 *>(row[u] + 16);
                else if (G.stride == 8) { /* 8-byte rows: the upper half of `lo` belongs to the next row, unused */ }
                d[u][0] = lo.x; d[u][1] = lo.y; d[u][2] = lo.z; d[u][3] = lo.w;
                d[u][4] = hi.x; d[u][5] = hi.y; d[u][6] = hi.z; d[u][7] = hi.w;
            }
#pragma unroll
            for (int c = 0; c < GM_MAX; c++) {
                if (c >= G.ncols) break;   // wave-uniform
                const int w0 = G.off[c] >> 2;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    unsigned a = 0, b = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) { a = w0 == k ? d[u][k] : a; b = (w0 + 1 == k) ? d[u][k] : b; }
                    v[u][c] = G.width[c] == 8 ? ((unsigned long long)b << 32) | a
                              : G.width[c] == 4 ? (unsigned long long)a
                                                : (unsigned long long)((a >> (8 * (G.off[c] & 3))) & 0xffu);
                }
            }
        } else {
#pragma unroll
        for (int c = 0; c < GM_MAX; c++) {
            if (c >= G.ncols) break;   // wave-uniform
#pragma unroll
            for (int u = 0; u < U; u++)
                v[u][c] = G.width[c] == 8 ? *reinterpret_cast<const unsigned long long *>(row[u] + G.off[c])
                          : G.width[c] == 4 ? (unsigned long long)*reinterpret_cast<const uint32_t *>(row[u] + G.off[c])
                                            : (unsigned long long)row[u][G.off[c]];
        }
        }
#pragma unroll
        for (int c = 0; c < GM_MAX; c++) {
            if (c >= G.ncols) break;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + u * 256 + threadIdx.x;
                if (i >= n) continue;
                if (G.width[c] == 8) ((unsigned long long *)G.dst[c])[i] = v[u][c];
                else if (G.width[c] == 4) ((uint32_t *)G.dst[c])[i] = (uint32_t)v[u][c];
                else ((uint8_t *)G.dst[c])[i] = (uint8_t)v[u][c];
            }
        }
    }
}
}  // namespace ph

// The views name columns of ONE resident table (recognised by their base pointers in the process-wide registry, whatever ctx created the
// table): through a co-located group that covers them when the table has one. A sparse gather (at most an eighth of the rows) of three or
// more columns that comes a SECOND time builds the group itself, inside the table's budget (ph_table_set_colocate_budget) — the layout
// follows the access pattern the plans show, as the statistics follow the data. The copy is built on the CALLER's stream and ordered
// against every other consumer's by an event (ph::colocated_group_for); the table's mutable state is behind its mutex, so two queries on two
// contexts may meet here. PH_OK = done, PH_EUNSUPPORTED = not this shape (the caller gathers column by column in one pass as before).
static int gather_through_group(ph_ctx *ctx, int32_t ncols, const ph_col *cols, const int32_t *idx_dev, int64_t n, void *const *out_dev) {
    static const bool off = getenv("PH_COLOCATE") && atoi(getenv("PH_COLOCATE")) == 0;
    if (off || ncols < 2) return PH_EUNSUPPORTED;
    ph_table *t = nullptr;
    std::vector<int> tc((size_t)ncols);
    for (int c = 0; c < ncols; c++) {
        ph_table *tt = nullptr;
        int col = -1;
        if (!ph::lookup_table_col(cols[c].data, &tt, &col) || (t && tt != t)) return PH_EUNSUPPORTED;
        t = tt;
        tc[(size_t)c] = col;
        if (ph::type_width(cols[c].type) != ph::type_width(t->cols[(size_t)col].type)) return PH_EUNSUPPORTED;
    }
    if (t->ctx && t->ctx->device != ctx->device) return PH_EUNSUPPORTED;
    ph_table::colgroup grp;
    const bool may_build = ncols >= 3 && n * 8 <= t->nrows;
    {
        const int rcg = ph::colocated_group_for(ctx, t, tc, may_build, false, &grp);
        if (rcg != PH_OK) return rcg == PH_EUNSUPPORTED ? rcg : rcg;
    }
    const ph_table::colgroup *g = &grp;
    ph::GatherGroup G{};
    G.rows = (const unsigned char *)g->data;
    G.stride = g->stride;
    G.ncols = ncols;
    for (int c = 0; c < ncols; c++) {
        const size_t k = (size_t)(std::find(g->cols.begin(), g->cols.end(), tc[(size_t)c]) - g->cols.begin());
        G.off[c] = g->off[k];
        G.width[c] = g->width[k];
        G.dst[c] = out_dev[c];
    }
    // rows per thread in flight (every row is one sector read by up to ncols adjacent loads)
    static const int gu = getenv("PH_GATHER_GROUP_U") ? atoi(getenv("PH_GATHER_GROUP_U")) : 1;   // measured on Q9 (3.27 M ascending ids into 60 M rows, a read per column): 1: 114 us, 2: 143, 4: 233, 8: 294
    if (gu == 8 && ncols <= 5) {
        const int grid = (int)std::min<int64_t>((n + 2047) / 2048, 256 * 16);
        ph::gather_group_kernel<8><<<grid, 256, 0, ctx->stream>>>(G, idx_dev, n);
    } else if (gu >= 4) {
        const int grid = (int)std::min<int64_t>((n + 1023) / 1024, 256 * 16);
        ph::gather_group_kernel<4><<<grid, 256, 0, ctx->stream>>>(G, idx_dev, n);
    } else if (gu == 1) {
        const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 32);
        ph::gather_group_kernel<1><<<grid, 256, 0, ctx->stream>>>(G, idx_dev, n);
    } else {
        const int grid = (int)std::min<int64_t>((n + 511) / 512, 256 * 16);
        ph::gather_group_kernel<2><<<grid, 256, 0, ctx->stream>>>(G, idx_dev, n);
    }
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_gather_multi(ph_ctx *ctx, int32_t ncols, const ph_col *cols, const int32_t *idx_dev, int64_t n,
                               void *const *out_dev) {
    PH_REQUIRE(ctx && ncols >= 1 && ncols <= ph::GM_MAX && cols && out_dev && (n == 0 || idx_dev),
               "ph_gather_multi: bad arguments (1..%d columns)", ph::GM_MAX);
    if (n == 0) return PH_OK;
    for (int c = 0; c < ncols; c++) PH_REQUIRE(cols[c].data && out_dev[c], "ph_gather_multi: column %d has a NULL pointer", c);
    {
        const int rcg = gather_through_group(ctx, ncols, cols, idx_dev, n, out_dev);
        if (rcg != PH_EUNSUPPORTED) return rcg;
    }
    ph::GatherMulti G{};
    G.ncols = ncols;
    for (int c = 0; c < ncols; c++) {
        G.width[c] = ph::type_width(cols[c].type);
        if (G.width[c] == 0) { ph::set_error("ph_gather_multi: column %d (type %d) is not fixed width", c, cols[c].type); return PH_EUNSUPPORTED; }
        PH_REQUIRE(cols[c].data && out_dev[c], "ph_gather_multi: column %d has a NULL pointer", c);
        G.src[c] = cols[c].data;
        G.dst[c] = out_dev[c];
    }
    // 1, 2, 4 or 8 rows per thread and 4 .. 64 workgroups per CU all run within 8 % of each other on Q9's
    // shape (5 columns at 5.4 % of 60 M ascending row ids: 245-265 us): the kernel is bound by the rate of
    // 64-byte sector reads (~4.2 TB/s of sectors for 0.1 TB/s of values), not by latency
    int grid = (int)std::min<int64_t>((n + 511) / 512, 256 * 16);
    ph::gather_multi_kernel<2><<<grid, 256, 0, ctx->stream>>>(G, idx_dev, n);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ partition (multi-GPU shuffle prep)
// dest = mix64(key) % nparts. Three kernels: per-workgroup histogram (LDS atomics), exclusive scan
// over [part][block], scatter of row ids. No reference counterpart.

namespace ph {

constexpr int PART_CHUNK = 2048;
constexpr int PART_MAX = 64;

__device__ __forceinline__ uint64_t part_key(const void *data, int width, int64_t r) {
    return width == 8 ? (uint64_t)((const int64_t *)data)[r] : (uint64_t)(int64_t)((const int32_t *)data)[r];
}

__global__ __launch_bounds__(256) void part_hist_kernel(const void *data, int width, const int32_t *sel,
                                                        int64_t n, int nparts, int nblocks,
                                                        int32_t *__restrict__ hist /* [part][block] */) {
    __shared__ int h[PART_MAX];
    if (threadIdx.x < PART_MAX) h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * PART_CHUNK;
    for (int r = 0; r < PART_CHUNK / 256; r++) {
        int64_t i = base + r * 256 + threadIdx.x;
        if (i < n) {
            int64_t row = sel ? sel[i] : i;
            int d = (int)(mix64(part_key(data, width, row)) % (uint64_t)nparts);
            atomicAdd(&h[d], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < nparts) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter: a row's slot = partition cursor + same-partition rows of earlier waves in this
// 256-row round + same-partition lower lanes (six ballots give every lane the mask of the lanes that
// share its partition), so rows keep their input order inside every partition and the permutation
// is reproducible run to run (first-seen group order survives an exchange).
__global__ __launch_bounds__(256) void part_scatter_kernel(const void *data, int width, const int32_t *sel,
                                                           int64_t n, int nparts, int nblocks,
                                                           const int32_t *__restrict__ offs,
                                                           int32_t *__restrict__ perm) {
    __shared__ int cur[PART_MAX];
    __shared__ int wcount[4][PART_MAX];
    if (threadIdx.x < PART_MAX) {
        cur[threadIdx.x] = threadIdx.x < nparts ? offs[(int64_t)threadIdx.x * nblocks + blockIdx.x] : 0;
        for (int w = 0; w < 4; w++) wcount[w][threadIdx.x] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int64_t base = (int64_t)blockIdx.x * PART_CHUNK;
    for (int r = 0; r < PART_CHUNK / 256; r++) {
        int64_t i = base + r * 256 + threadIdx.x;
        const bool live = i < n;
        int64_t row = 0;
        int d = 0;
        if (live) {
            row = sel ? sel[i] : i;
            d = (int)(mix64(part_key(data, width, row)) % (uint64_t)nparts);
        }
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 6; b++) {
            unsigned long long m = __ballot((d >> b) & 1);
            same &= ((d >> b) & 1) ? m : ~m;
        }
        const int rank = __popcll(same & lt);
        if (live && rank == 0) wcount[wv][d] = __popcll(same);
        __syncthreads();
        if (live) {
            int pos = cur[d] + rank;
            for (int w = 0; w < wv; w++) pos += wcount[w][d];
            perm[pos] = (int32_t)row;
        }
        __syncthreads();
        if (threadIdx.x < PART_MAX) {
            int t = 0;
            for (int w = 0; w < 4; w++) { t += wcount[w][threadIdx.x]; wcount[w][threadIdx.x] = 0; }
            cur[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

// counts[p] = first offset of partition p+1 - first offset of partition p (after the scan)
__global__ void part_counts_kernel(const int32_t *__restrict__ offs, int nblocks, int nparts, int64_t n,
                                   int64_t *__restrict__ counts) {
    int p = threadIdx.x;
    if (p >= nparts) return;
    int64_t a = offs[(int64_t)p * nblocks];
    int64_t b = p + 1 < nparts ? offs[(int64_t)(p + 1) * nblocks] : n;
    counts[p] = b - a;
}

}  // namespace ph

extern "C" int ph_partition_dev(ph_ctx *ctx, const ph_col *key, const int32_t *sel, int64_t n, int32_t nparts,
                                int64_t *counts_dev, int32_t *perm_dev) {
    PH_REQUIRE(ctx && key && counts_dev && nparts >= 1 && nparts <= ph::PART_MAX && n >= 0 && n < (1ll << 31),
               "ph_partition: bad arguments (1 <= nparts <= %d)", ph::PART_MAX);
    int w = ph::type_width(key->type);
    PH_REQUIRE(w == 4 || w == 8, "ph_partition: key must be a 32- or 64-bit integer column");
    PH_REQUIRE(key->validity == nullptr, "ph_partition: NULL-able partition keys are not supported");
    if (n == 0) { PH_HIP(hipMemsetAsync(counts_dev, 0, (size_t)nparts * 8, ctx->stream)); return PH_OK; }
    PH_REQUIRE(perm_dev != nullptr, "ph_partition: perm_dev is NULL");
    int64_t nb = (n + ph::PART_CHUNK - 1) / ph::PART_CHUNK;
    int64_t cells = nb * nparts;
    PH_CHECK(ctx->ensure_scratch(ph::round_up(cells * 4, 8) + 64));
    int32_t *hist = (int32_t *)ctx->scratch;
    int64_t *total = (int64_t *)((char *)ctx->scratch + ph::round_up(cells * 4, 8));
    ph::part_hist_kernel<<<(int)nb, 256, 0, ctx->stream>>>(key->data, w, sel, n, nparts, (int)nb, hist);
    PH_HIP(hipGetLastError());
    // first block offset of each partition = exclusive scan over the part-major layout
    PH_CHECK(ph::exclusive_scan_i32(ctx, hist, cells, total));
    ph::part_counts_kernel<<<1, 64, 0, ctx->stream>>>(hist, (int)nb, nparts, n, counts_dev);
    ph::part_scatter_kernel<<<(int)nb, 256, 0, ctx->stream>>>(key->data, w, sel, n, nparts, (int)nb, hist, perm_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_partition(ph_ctx *ctx, const ph_col *key, const int32_t *sel, int64_t n, int32_t nparts,
                            int64_t *counts_host, int32_t *perm_dev) {
    PH_REQUIRE(ctx && counts_host && nparts >= 1 && nparts <= ph::PART_MAX, "ph_partition: bad arguments (1 <= nparts <= %d)", ph::PART_MAX);
    int64_t *counts_dev = nullptr;
    PH_CHECK(ctx->pool_alloc((int64_t)nparts * 8, (void **)&counts_dev));
    int rc = ph_partition_dev(ctx, key, sel, n, nparts, counts_dev, perm_dev);
    if (rc == PH_OK) rc = ctx->download(counts_host, counts_dev, (int64_t)nparts * 8);   // one round trip
    ctx->pool_release(counts_dev);
    return rc;
}
