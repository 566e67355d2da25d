// Device-side helpers shared by the operator kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ph {

__device__ __forceinline__ bool bit_valid(const uint8_t *mask, int64_t i) {
    // pkg/util/bitmap.go:72-77 — 1 bit per row, LSB first; NULL mask = all valid (:171-173)
    return mask == nullptr || ((mask[i >> 3] >> (i & 7)) & 1);
}

// reference hash primitives (pkg/chunk/hash.go:26-41) — bit-identical by construction
__device__ __host__ __forceinline__ uint64_t murmurhash64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    return x;
}
__device__ __host__ __forceinline__ uint64_t combine_hash(uint64_t a, uint64_t b) {
    return (a * 0xbf58476d1ce4e5b9ULL) ^ b;
}
constexpr uint64_t NULL_HASH = 0xbf58476d1ce4e5b9ULL;

// util.HashBytes (pkg/util/hash.go:13-65): MurmurHash64A-style, seed 0xe17a1465
__device__ __host__ inline uint64_t hash_bytes(const uint8_t *p, uint64_t len) {
    const uint64_t M = 0xc6a4a7935bd1e995ULL;
    const int R = 47;
    uint64_t h = 0xe17a1465ULL ^ (len * M);
    uint64_t nblocks = len / 8;
    for (uint64_t i = 0; i < nblocks; i++) {
        uint64_t k = 0;
        for (int b = 0; b < 8; b++) k |= (uint64_t)p[8 * i + b] << (8 * b);  // little-endian load
        k *= M;
        k ^= k >> R;
        k *= M;
        h ^= k;
        h *= M;
    }
    const uint8_t *t = p + 8 * nblocks;
    uint64_t rem = len & 7;
    if (rem) {
        for (uint64_t b = rem; b-- > 0;) h ^= (uint64_t)t[b] << (8 * b);
        h *= M;
    }
    h ^= h >> R;
    h *= M;
    h ^= h >> R;
    return h;
}

// table-placement mixer for the device hash tables (placement needs no parity with the reference)
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

// days since 1970-01-01 -> (year, month, day); what the reference's scan materialises as
// Date{Year,Month,Day} (executor_scan.go:419-423, pkg/common/date.go:8-12)
__device__ __host__ __forceinline__ void civil_from_days(int32_t z, int32_t *y, int32_t *m, int32_t *d) {
    z += 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    int32_t yy = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    uint32_t mp = (5u * doy + 2u) / 153u;
    *d = (int32_t)(doy - (153u * mp + 2u) / 5u + 1u);
    *m = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    *y = yy + (*m <= 2);
}

// LIKE with % and _ (same greedy/backtracking matcher as wildcardMatch,
// function_operator_boolean.go:336-377)
__device__ __host__ __forceinline__ bool like_match(const char *s, int slen, const char *pat, int plen) {
    int p = 0, t = 0, after_pct = -1, t_at_pct = -1;
    while (t < slen) {
        if (p < plen && pat[p] == '%') {
            p++;
            after_pct = p;
            if (p >= plen) return true;
            t_at_pct = t;
        } else if (p < plen && (pat[p] == '_' || pat[p] == s[t])) {
            p++;
            t++;
        } else {
            if (after_pct < 0 || t_at_pct < 0) return false;
            p = after_pct;
            t_at_pct++;
            t = t_at_pct;
        }
    }
    while (p < plen && pat[p] == '%') p++;
    return p >= plen;
}

}  // namespace ph
