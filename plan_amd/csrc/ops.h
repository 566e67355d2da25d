// Shared between the operator translation units.
#pragma once
#include "common.h"

namespace ph {
// in-place exclusive scan of n int32 on the ctx stream; *total_dev receives the sum
int exclusive_scan_i32(ph_ctx *ctx, int32_t *dev, int64_t n, int64_t *total_dev);
}  // namespace ph
