/* ORACLE — TEST INFRASTRUCTURE ONLY. See odecimal.h for what this restates and why. */
#include "odecimal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static u128 p10(int k) {
    u128 r = 1;
    while (k-- > 0) r *= 10;
    return r;
}

static int ndigits(u128 v) {
    int n = 0;
    while (v) {
        v /= 10;
        n++;
    }
    return n ? n : 1;
}

/* drop `drop` low digits of c, rounding half-to-even; `sticky` = a non-zero tail below c */
static u128 rsh_half_even(u128 c, int drop, int sticky) {
    if (drop <= 0) return c;
    u128 d = p10(drop);
    u128 q = c / d, rem = c % d, half = d / 2;
    if (rem > half || (rem == half && (sticky || (q & 1)))) q++;
    return q;
}

/* fit (coef, scale) into 19 digits / scale <= 19, never going below min_scale */
static int fit(int neg, u128 coef, int scale, int min_scale, int sticky, odec *out) {
    for (;;) {
        int drop = ndigits(coef) - ODEC_MAX_PREC;
        if (scale - ODEC_MAX_PREC > drop) drop = scale - ODEC_MAX_PREC;
        if (drop <= 0) break;
        if (scale - drop < min_scale) return ODEC_OVERFLOW;
        coef = rsh_half_even(coef, drop, sticky);
        sticky = 0;
        scale -= drop;
    }
    if (scale < 0) return ODEC_OVERFLOW;
    out->neg = (coef != 0) ? (uint8_t)neg : 0;
    out->coef = (uint64_t)coef;
    out->scale = (int8_t)scale;
    return ODEC_OK;
}

int odec_new(int64_t value, int scale, odec *out) {
    if (scale < 0 || scale > 19) return ODEC_OVERFLOW;
    out->neg = value < 0;
    out->coef = value < 0 ? (uint64_t)0 - (uint64_t)value : (uint64_t)value;
    out->scale = (int8_t)scale;
    return ODEC_OK;
}

int odec_new_from_int64(int64_t whole, int64_t frac, int scale, odec *out) {
    if (scale < 0 || scale > 19) return ODEC_OVERFLOW;
    if ((whole > 0 && frac < 0) || (whole < 0 && frac > 0)) return ODEC_OVERFLOW;
    int neg = whole < 0 || frac < 0;
    u128 w = whole < 0 ? (u128)((uint64_t)0 - (uint64_t)whole) : (u128)whole;
    u128 f = frac < 0 ? (u128)((uint64_t)0 - (uint64_t)frac) : (u128)frac;
    if (f >= p10(scale) && scale > 0) return ODEC_OVERFLOW;
    if (scale == 0 && f != 0) return ODEC_OVERFLOW;
    /* trailing zeros of the fractional part are removed */
    if (f == 0) {
        scale = 0;
    } else {
        while (f % 10 == 0) {
            f /= 10;
            scale--;
        }
    }
    u128 coef = w * p10(scale) + f;
    return fit(neg, coef, scale, 0, 0, out);
}

odec odec_neg(odec a) {
    if (a.coef != 0) a.neg = !a.neg;
    return a;
}

int odec_is_zero(odec a) { return a.coef == 0; }

static int add_signed(odec a, odec b, odec *out) {
    int s = a.scale > b.scale ? a.scale : b.scale;
    u128 x = (u128)a.coef * p10(s - a.scale);
    u128 y = (u128)b.coef * p10(s - b.scale);
    if (a.neg == b.neg) return fit(a.neg, x + y, s, 0, 0, out);
    if (x >= y) return fit(a.neg, x - y, s, 0, 0, out);
    return fit(b.neg, y - x, s, 0, 0, out);
}

int odec_add(odec a, odec b, odec *out) { return add_signed(a, b, out); }
int odec_sub(odec a, odec b, odec *out) { return add_signed(a, odec_neg(b), out); }

int odec_mul(odec a, odec b, odec *out) {
    u128 c = (u128)a.coef * (u128)b.coef;
    return fit(a.neg != b.neg, c, a.scale + b.scale, 0, 0, out);
}

int odec_cmp(odec a, odec b) {
    if (a.coef == 0 && b.coef == 0) return 0;
    if (a.neg != b.neg) return a.neg ? -1 : 1;
    int s = a.scale > b.scale ? a.scale : b.scale;
    u128 x = (u128)a.coef * p10(s - a.scale);
    u128 y = (u128)b.coef * p10(s - b.scale);
    int r = x < y ? -1 : (x > y ? 1 : 0);
    return a.neg ? -r : r;
}

int odec_quo(odec a, odec b, odec *out) {
    if (b.coef == 0) return ODEC_DIVZERO;
    int pref = a.scale - b.scale;
    if (pref < 0) pref = 0;
    if (a.coef == 0) {
        out->neg = 0;
        out->coef = 0;
        out->scale = (int8_t)pref;
        return ODEC_OK;
    }
    u128 n = a.coef;
    int scale = a.scale - b.scale;
    if (scale < 0) {
        n *= p10(-scale);
        scale = 0;
    }
    u128 d = b.coef;
    u128 c = n / d, r = n % d;
    /* extend with fractional digits until 20 significant digits (one guard) or scale 20 */
    while (r != 0 && ndigits(c) <= ODEC_MAX_PREC && scale <= ODEC_MAX_PREC) {
        r *= 10;
        c = c * 10 + r / d;
        r %= d;
        scale++;
    }
    odec q;
    int rc = fit(a.neg != b.neg, c, scale, 0, r != 0, &q);
    if (rc != ODEC_OK) return rc;
    while (q.scale > pref && q.coef % 10 == 0) { /* Trim(pref) */
        q.coef /= 10;
        q.scale--;
    }
    *out = q;
    return ODEC_OK;
}

int odec_int64(odec d, int scale, int64_t *whole, int64_t *frac) {
    if (scale < 0 || scale > 19) return 0;
    u128 c = d.coef;
    if (scale < d.scale)
        c = rsh_half_even(c, d.scale - scale, 0);
    else
        c *= p10(scale - d.scale);
    u128 y = p10(scale);
    u128 w = c / y, f = c % y;
    if (w > (u128)INT64_MAX || f > (u128)INT64_MAX) return 0;
    *whole = d.neg ? -(int64_t)w : (int64_t)w;
    *frac = d.neg ? -(int64_t)f : (int64_t)f;
    return 1;
}

int odec_string(odec d, char *buf) {
    char digs[24];
    int n = snprintf(digs, sizeof digs, "%llu", (unsigned long long)d.coef);
    char *p = buf;
    if (d.neg) *p++ = '-';
    if (d.scale == 0) {
        memcpy(p, digs, (size_t)n);
        p += n;
    } else if (n > d.scale) {
        memcpy(p, digs, (size_t)(n - d.scale));
        p += n - d.scale;
        *p++ = '.';
        memcpy(p, digs + n - d.scale, (size_t)d.scale);
        p += d.scale;
    } else {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < d.scale - n; i++) *p++ = '0';
        memcpy(p, digs, (size_t)n);
        p += n;
    }
    *p = 0;
    return (int)(p - buf);
}

double odec_float64(odec d) {
    char buf[48];
    odec_string(d, buf);
    return strtod(buf, NULL);
}

int odec_to_unscaled(odec d, int scale, __int128 *out) {
    if (scale < d.scale) {
        u128 q = p10(d.scale - scale);
        if ((u128)d.coef % q != 0) return 0;
        u128 c = (u128)d.coef / q;
        *out = d.neg ? -(__int128)c : (__int128)c;
        return 1;
    }
    u128 c = (u128)d.coef * p10(scale - d.scale);
    *out = d.neg ? -(__int128)c : (__int128)c;
    return 1;
}
