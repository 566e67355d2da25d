// Parameter blocks of the fused scan kernels (scan_kernels.hip), filled by scan_plan.hip.
#pragma once
#include "common.h"

namespace ph {

constexpr int LC_NACC = 6;   // Σq, Σe, Σe·f1, Σe·f1·f2, Σd, count
constexpr int LC_MAX_SLOTS = 12;

// One launch per fused scan: every workgroup stores its partial words with device-scope stores, takes a ticket, and the workgroup that finishes LAST
// reads all partials (coalesced, every load independent), folds them in LDS into 128-bit sums (what merge_partials_kernel does as a second launch),
// stores them to out_lo / out_hi and, when mbox is set, into the mapped mailbox followed by the sequence number (what publish_kernel does as a
// third), and zeroes the ticket for the next launch. done == NULL: no tail (the caller launches the merge itself).
constexpr int SCAN_TAIL_MAX_ACC = 128;
struct ScanTail {
    unsigned *done;               // zero on entry, reset by the last workgroup
    unsigned long long *out_lo;   // [nacc]
    long long *out_hi;            // [nacc]
    int32_t nacc, min_stride;     // as launch_merge_partials
    unsigned long long *mbox;     // device address of the mailbox: lo[nacc] then hi[nacc]; NULL = no publish
    unsigned long long *flag;     // its sequence word
    unsigned long long seq;
};

struct FilterSumProdParams {
    const int32_t *p0;  // int32 range-predicate column
    const int32_t *p2;  // int32 range-predicate column
    const int64_t *b;   // int64 range-predicate column, second factor of the product
    const int64_t *a;   // int64 first factor
    int32_t p0_lo, p0_hi, p2_lo, p2_hi;
    int64_t b_lo, b_hi;
    int64_t row_begin, row_end;
    long long *partials;  // [grid][2] = {Σ a*b, count}
    ScanTail tail;
};

struct LowcardChainParams {
    const int32_t *p;   // int32 range-predicate column
    const int32_t *q;   // int32 summed column
    const int64_t *e, *d, *t;
    const uint8_t *k0, *k1;
    int32_t p_lo, p_hi;
    int32_t nk1;        // slot = k0 * nk1 + k1
    int32_t nslots;
    int64_t A1, B1, A2, B2;  // f1 = A1 + B1*d, f2 = A2 + B2*t
    int64_t row_begin, row_end;
    long long *partials;  // [grid][nslots][LC_NACC+1]: sums, count, first row id
    ScanTail tail;
};

int launch_filter_sumprod(ph_ctx *ctx, const FilterSumProdParams &P, int grid);
int launch_lowcard_chain(ph_ctx *ctx, const LowcardChainParams &P, int grid);
// publish (may be NULL; ignored without a mailbox): the merge's last wave also publishes the merged words (ScanTail: done, nacc, mbox, flag, seq)
int launch_merge_partials(ph_ctx *ctx, const long long *partials, int nblocks, int nacc,
                          int min_stride, unsigned long long *out_lo, long long *out_hi, const ScanTail *publish = nullptr);

int launch_merge_partials_ops(ph_ctx *ctx, const long long *partials, int nblocks, int nacc, int stride,
                              unsigned long long opmask, unsigned long long *out_lo, long long *out_hi, const ScanTail *publish = nullptr);

}  // namespace ph
