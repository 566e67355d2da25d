// Plan-specialised fused Agg <- Scan(filter) kernels, compiled at ph_scan_plan_create with hiprtc.
//
// The two precompiled fused kernels (scan_kernels.hip) cover exactly the Q1 and Q6 shapes. Every
// other low-cardinality plan used to fall to the operator chain, whose aggregate sink interprets
// key / aggregate descriptors per row (~750 instructions per row-wave, 11-16 % of HBM peak). Here the
// plan's shape — which columns are read and how wide they are, the range predicates, the dense
// group index over dictionary-code columns, every aggregate's product of affine column factors —
// becomes the SOURCE of one kernel with the lowcard_chain structure (one 256-thread workgroup per
// CU, 4 consecutive rows per lane, 16-byte non-temporal loads of only the columns the plan names,
// register double buffering, per-thread-private LDS accumulator columns or plain registers for an
// ungrouped aggregate). Constants (bounds, affine coefficients) are kernel parameters, so the
// compiled module is cached by shape and reused across literals.
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace ph {

struct JitAffine { int64_t A = 0, B = 0; int col = -1; };   // A + B * column(col slot)

struct JitShape {
    std::vector<int> col_width;            // loaded columns: 1 (code byte), 4, 8 bytes
    struct Pred { int col; bool ne; };     // range lo <= v <= hi, or v != k; constants are parameters
    std::vector<Pred> preds;
    std::vector<int> group_col;            // loaded-column slots of the group keys (code bytes)
    std::vector<int> group_card;           // dictionary size of each
    int nslots = 1;                        // product of group_card (1 = ungrouped)
    int lds_cols = 256;                    // accumulator columns per (slot, accumulator): 256 = one per thread
    // largest power-of-two column count (256..16) whose accumulators fit the CU's LDS; 0 = none does
    static int fit_lds_cols(int nslots, int naccs) {
        for (int c = 256; c >= 16; c >>= 1)
            if ((int64_t)nslots * (naccs * 8 + 8) * c <= 156 * 1024) return c;
        return 0;
    }
    // accumulators: op 0 = SUM of a product of affine factors, 1 = MIN, 2 = MAX of one column
    struct Acc { int op; std::vector<std::pair<int, bool>> factors; };  // (column slot, pure: A=0,B=1)
    std::vector<Acc> accs;
    std::string key() const;               // cache key (shape only)
};

constexpr int JIT_MAX_CONSTS = 64;
constexpr int JIT_MAX_COLS = 12;

struct JitParams {                          // passed by value to the generated kernel
    const void *col[JIT_MAX_COLS];
    long long row_begin, row_end;
    long long *partials;                    // [grid][nslots][naccs + 2]: accumulators, count, first row
    long long k[JIT_MAX_CONSTS];            // predicate bounds, then affine (A, B) pairs in factor order
};

std::string jit_generate(const JitShape &s);

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
};
// compile (or fetch from the in-process cache) the kernel of a shape for the ctx's device
int jit_get(ph_ctx *ctx, const JitShape &s, JitKernel *out);
// compile only (no device needed): status + log, for the CPU-side build check
int jit_compile_check(const JitShape &s, const char *arch, std::string *log);
int jit_launch(ph_ctx *ctx, const JitKernel &k, const JitParams &p, int grid);
// generic: compile `src` for the ctx's device (cached by key for the process lifetime) and return `entry`
int jit_module(ph_ctx *ctx, const std::string &key, const std::string &src, const char *entry, JitKernel *out);
int jit_compile_only(const std::string &src, const char *arch, std::string *log);
// cache probe by key alone: a hit skips generating the source (tens of microseconds per call)
bool jit_cached(ph_ctx *ctx, const std::string &key, JitKernel *out);

}  // namespace ph
