// Shared between the operator translation units.
#pragma once
#include "common.h"

namespace ph {
// in-place exclusive scan of n int32 on the ctx stream; *total_dev receives the sum
// pub (optional, from ph_ctx::arm_publish): the thread that stores the total also stores it into the mapped mailbox and then the sequence
// number — the host learns a selection's count while the write pass that follows the scan is still running, and without a publish launch
int exclusive_scan_i32(ph_ctx *ctx, int32_t *dev, int64_t n, int64_t *total_dev, const ScanPublish *pub = nullptr);
// The decoupled look-back's per-context tile states for ONE kernel of `tiles` tiles (see scan_lookback_kernel: entries carry the epoch of the call
// that wrote them, the ticket counter keeps counting across calls): state[tiles], the ticket word, its base for this call and the call's epoch.
int scan_state_acquire(ph_ctx *ctx, int64_t tiles, unsigned long long **state, unsigned **ticket, unsigned *ticket_base, unsigned long long *epoch);

// A `column OP constant` comparison lowered to an integer range test, for kernels that evaluate a
// pushed-down filter inline (the fused filter+probe). kind 0 = no predicate.
struct RangePred {
    int kind;  // 0 none, 1 int32, 2 int64, 3 uint8 (dictionary code), -1 never true
    const void *data;
    const uint8_t *validity;
    long long lo, hi;
};
// false when the comparison does not lower to a range (strings, float compares, !=): the caller
// falls back to ph_filter_select + the plain probe
bool lower_range_pred(const ph_col *col, int32_t op, const ph_const *k, RangePred *out);
}  // namespace ph
