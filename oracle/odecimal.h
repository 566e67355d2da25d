/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under plan_amd/ may include, link or call this.
 *
 * odecimal — CPU restatement of the decimal arithmetic the reference's compute path runs on:
 * the third-party Go module github.com/govalues/decimal v0.1.28 (reference go.mod:15), whose
 * source is NOT under /root/reference. Restated from the module's published behaviour:
 *   value = (-1)^neg * coef / 10^scale, coef <= 10^19-1 (19 digits), scale in [0,19];
 *   Add/Sub: scale = max(scales), exact, else rounded half-to-even to 19 digits;
 *   Mul: scale = sum of scales, same rounding rule;
 *   Quo: quotient rounded half-to-even to 19 significant digits, trailing zeros trimmed down to
 *        the preferred scale max(0, sd - se);
 *   Int64(s): (whole, frac) after half-to-even rounding to scale s;
 *   NewFromInt64(w, f, s): w + f/10^s with trailing zeros of f removed;
 *   String(): exactly `scale` fractional digits.
 * Reference call sites this stands in for: pkg/common/decimal.go:19-45,
 * pkg/compute/function_operator_binary.go:134-207, function_aggr.go:687-689, 886-895,
 * function_cast.go:337-404, pkg/chunk/vector.go:121-137, 245-264, pkg/chunk/value.go:37-46.
 * Pin: the reference's SF1 goldens (tests/golden/plan_q{1,3,6,9}.txt) — every decimal printed
 * there went through Mul/Sub/Add/Quo/Int64/NewFromInt64/String.
 */
#ifndef ODECIMAL_H
#define ODECIMAL_H
#include <stdint.h>

typedef struct {
    uint8_t neg;
    int8_t scale;
    uint64_t coef;
} odec;

#define ODEC_OK 0
#define ODEC_OVERFLOW 1
#define ODEC_DIVZERO 2
#define ODEC_MAX_PREC 19

int odec_new(int64_t value, int scale, odec *out); /* MustNew(value, scale) */
int odec_new_from_int64(int64_t whole, int64_t frac, int scale, odec *out);
int odec_add(odec a, odec b, odec *out);
int odec_sub(odec a, odec b, odec *out);
int odec_mul(odec a, odec b, odec *out);
int odec_quo(odec a, odec b, odec *out);
odec odec_neg(odec a);
int odec_cmp(odec a, odec b);
int odec_is_zero(odec a);
/* Int64(scale): returns 1 when representable (ok), 0 otherwise */
int odec_int64(odec d, int scale, int64_t *whole, int64_t *frac);
/* String(); buf must hold >= 48 bytes; returns length */
int odec_string(odec d, char *buf);
/* Float64(): nearest double of the decimal text */
double odec_float64(odec d);
/* unscaled int64 at a given scale (exact or fails) — helper for tests */
int odec_to_unscaled(odec d, int scale, __int128 *out);

#endif
