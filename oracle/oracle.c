/* ORACLE — TEST INFRASTRUCTURE ONLY. See oracle.h. Operator-level restatements. */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VS ORACLE_VECTOR_SIZE

/* ------------------------------------------------------------------ helpers */

static inline int row_valid(const uint8_t *mask, int64_t i) {
    /* Bitmap.RowIsValid (pkg/util/bitmap.go:72-77); nil mask = all valid (:171-173) */
    return mask == NULL || ((mask[i >> 3] >> (i & 7)) & 1);
}

typedef struct {
    int32_t y, m, d;
} odate; /* pkg/common/date.go:8-12 */

static odate date_from_days(int32_t z) {
    /* what the scan materialises: time.Date(1970,1,1+days) -> (Y,M,D) (executor_scan.go:419-423) */
    z += 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    int32_t y = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    uint32_t mp = (5u * doy + 2u) / 153u;
    odate r;
    r.d = (int32_t)(doy - (153u * mp + 2u) / 5u + 1u);
    r.m = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    r.y = y + (r.m <= 2);
    return r;
}

static int date_cmp(odate a, odate b) {
    /* Date.Less builds two time.Time and compares (date.go:18-26); for valid calendar dates that
     * is the lexicographic order of (Y,M,D). Date.Equal compares the fields (:14-16). */
    if (a.y != b.y) return a.y < b.y ? -1 : 1;
    if (a.m != b.m) return a.m < b.m ? -1 : 1;
    if (a.d != b.d) return a.d < b.d ? -1 : 1;
    return 0;
}

static odec dec_from_unscaled(int64_t v, int scale) {
    /* the loader splits unscaled into (whole, frac) and calls NewFromInt64, which drops the
     * fraction's trailing zeros (executor_scan.go:447-460, pkg/chunk/vector.go:257-264) */
    int64_t p = 1;
    for (int i = 0; i < scale; i++) p *= 10;
    odec d;
    if (odec_new_from_int64(v / p, v % p, scale, &d) != ODEC_OK) abort();
    return d;
}

int oracle_like(const char *s, int64_t slen, const char *pattern) {
    /* wildcardMatch (function_operator_boolean.go:336-377): % = any run, _ = one byte,
     * backtracking to the last % */
    int64_t plen = (int64_t)strlen(pattern);
    int64_t p = 0, t = 0, after_pct = -1, t_at_pct = -1;
    while (t < slen) {
        if (p < plen && pattern[p] == '%') {
            p++;
            after_pct = p;
            if (p >= plen) return 1;
            t_at_pct = t;
        } else if (p < plen && (pattern[p] == '_' || pattern[p] == s[t])) {
            p++;
            t++;
        } else {
            if (after_pct < 0 || t_at_pct < 0) return 0;
            p = after_pct;
            t_at_pct++;
            t = t_at_pct;
        }
    }
    while (p < plen && pattern[p] == '%') p++;
    return p >= plen;
}

/* ------------------------------------------------------------------ select */

static int cmp_result(int c, int32_t op) {
    switch (op) {
    case OP_EQ: return c == 0;
    case OP_NE: return c != 0;
    case OP_LT: return c < 0;
    case OP_LE: return c <= 0;
    case OP_GT: return c > 0;
    case OP_GE: return c >= 0;
    default: return 0;
    }
}

/* which (physical type, op) pairs selectOperation implements (function_operator_boolean.go:393-504);
 * everything else returns 0 rows */
static int select_supported(int phys, int32_t op) {
    enum { P_INT32, P_DATE, P_FLOAT, P_DOUBLE, P_DECIMAL, P_VARCHAR, P_INT64 };
    switch (op) {
    case OP_EQ: case OP_NE: return phys == P_INT32 || phys == P_VARCHAR;
    case OP_GT: return phys == P_INT32 || phys == P_DATE || phys == P_FLOAT || phys == P_DECIMAL;
    case OP_GE: return phys == P_INT32 || phys == P_DATE || phys == P_FLOAT;
    case OP_LT: return phys == P_INT32 || phys == P_DATE || phys == P_DOUBLE;
    case OP_LE: return phys == P_INT32 || phys == P_DATE || phys == P_FLOAT;
    case OP_LIKE: case OP_NOTLIKE: return phys == P_VARCHAR;
    default: return 0;
    }
}

int64_t oracle_select(const ocol *col, int32_t op, const oconst *k, const int64_t *sel_in,
                      int64_t n_in, int64_t *sel_out) {
    enum { P_INT32, P_DATE, P_FLOAT, P_DOUBLE, P_DECIMAL, P_VARCHAR, P_INT64 };
    int phys;
    /* the comparison's operand type after the binder's implicit casts */
    if (col->type == OT_DECIMAL && k->type == OT_FLOAT) phys = P_FLOAT; /* MaxLType(DECIMAL,FLOAT)=FLOAT */
    else if (col->type == OT_DECIMAL) phys = P_DECIMAL;
    else if (col->type == OT_INT32) phys = P_INT32;
    else if (col->type == OT_INT64) phys = P_INT64;
    else if (col->type == OT_DATE) phys = P_DATE;
    else if (col->type == OT_FLOAT) phys = P_FLOAT;
    else if (col->type == OT_DOUBLE) phys = P_DOUBLE;
    else phys = P_VARCHAR;
    if (!select_supported(phys, op)) return 0;

    odate kd = {0, 0, 0};
    odec kdec = {0, 0, 0};
    float kf = (float)k->f;
    if (phys == P_DATE) kd = date_from_days((int32_t)k->i);
    if (phys == P_DECIMAL) kdec = dec_from_unscaled(k->i, k->scale);

    int64_t out = 0;
    /* chunk loop: one selectFlatLoop per <=2048 input rows */
    for (int64_t base = 0; base < n_in; base += VS) {
        int64_t cnt = n_in - base < VS ? n_in - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t r = sel_in ? sel_in[base + j] : base + j;
            if (!row_valid(col->validity, r)) continue; /* NULL never selects (:842-866) */
            int res = 0;
            switch (phys) {
            case P_INT32: {
                int32_t v = ((const int32_t *)col->data)[r];
                int32_t c = (int32_t)k->i;
                res = cmp_result(v < c ? -1 : (v > c ? 1 : 0), op);
                break;
            }
            case P_DATE: {
                odate v = date_from_days(((const int32_t *)col->data)[r]);
                res = cmp_result(date_cmp(v, kd), op);
                break;
            }
            case P_FLOAT: {
                float v;
                if (col->type == OT_DECIMAL) {
                    odec d = dec_from_unscaled(((const int64_t *)col->data)[r], col->scale);
                    v = (float)odec_float64(d); /* tryCastDecimalToFloat32 */
                } else {
                    v = ((const float *)col->data)[r];
                }
                res = cmp_result(v < kf ? -1 : (v > kf ? 1 : (v == kf ? 0 : 2)), op);
                break;
            }
            case P_DOUBLE: {
                double v = ((const double *)col->data)[r];
                res = cmp_result(v < k->f ? -1 : (v > k->f ? 1 : (v == k->f ? 0 : 2)), op);
                break;
            }
            case P_DECIMAL: {
                odec d = dec_from_unscaled(((const int64_t *)col->data)[r], col->scale);
                res = cmp_result(odec_cmp(d, kdec), op);
                break;
            }
            case P_VARCHAR: {
                const char *s;
                int64_t slen;
                if (col->type == OT_CODE8) {
                    s = col->dict[((const uint8_t *)col->data)[r]];
                    slen = (int64_t)strlen(s);
                } else {
                    const int32_t *off = (const int32_t *)col->data;
                    s = (const char *)col->dict + off[r];
                    slen = off[r + 1] - off[r];
                }
                if (op == OP_LIKE) res = oracle_like(s, slen, k->s);
                else if (op == OP_NOTLIKE) res = !oracle_like(s, slen, k->s);
                else {
                    int64_t kl = (int64_t)strlen(k->s);
                    int eq = (kl == slen) && memcmp(s, k->s, (size_t)slen) == 0;
                    res = (op == OP_EQ) ? eq : !eq;
                }
                break;
            }
            default: break;
            }
            if (res) sel_out[out++] = r;
        }
    }
    return out;
}

/* ------------------------------------------------------------------ hash */

#define NULL_HASH 0xbf58476d1ce4e5b9ULL

static inline uint64_t murmurhash64(uint64_t x) { /* pkg/chunk/hash.go:26-33 */
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    return x;
}

static inline uint64_t combine_hash(uint64_t a, uint64_t b) { /* hash.go:39-41 */
    return (a * 0xbf58476d1ce4e5b9ULL) ^ b;
}

static uint64_t hash_bytes(const uint8_t *p, uint64_t len) { /* pkg/util/hash.go:13-65 */
    const uint64_t M = 0xc6a4a7935bd1e995ULL, SEED = 0xe17a1465ULL;
    const int R = 47;
    uint64_t h = SEED ^ (len * M);
    uint64_t nblocks = len / 8;
    for (uint64_t i = 0; i < nblocks; i++) {
        uint64_t k;
        memcpy(&k, p + 8 * i, 8);
        k *= M;
        k ^= k >> R;
        k *= M;
        h ^= k;
        h *= M;
    }
    const uint8_t *t = p + 8 * nblocks;
    switch (len & 7) {
    case 7: h ^= (uint64_t)t[6] << 48; /* fallthrough */
    case 6: h ^= (uint64_t)t[5] << 40; /* fallthrough */
    case 5: h ^= (uint64_t)t[4] << 32; /* fallthrough */
    case 4: h ^= (uint64_t)t[3] << 24; /* fallthrough */
    case 3: h ^= (uint64_t)t[2] << 16; /* fallthrough */
    case 2: h ^= (uint64_t)t[1] << 8;  /* fallthrough */
    case 1: h ^= (uint64_t)t[0]; h *= M; /* fallthrough */
    default: break;
    }
    h ^= h >> R;
    h *= M;
    h ^= h >> R;
    return h;
}

static uint64_t hash_value(const ocol *c, int64_t r) {
    if (c->type != OT_CONST32 && !row_valid(c->validity, r)) return NULL_HASH;
    switch (c->type) {
    case OT_INT32: /* HashFuncInt32: uint32 zero-extended (hash.go:53-55) */
        return murmurhash64((uint64_t)(uint32_t)((const int32_t *)c->data)[r]);
    case OT_CONST32:
        return murmurhash64((uint64_t)(uint32_t)((const int32_t *)c->data)[0]);
    case OT_INT64:
        return murmurhash64((uint64_t)((const int64_t *)c->data)[r]);
    case OT_DATE: { /* h(Y)^h(M)^h(D) (hash.go:127-129) */
        odate d = date_from_days(((const int32_t *)c->data)[r]);
        return murmurhash64((uint64_t)(int64_t)d.y) ^ murmurhash64((uint64_t)(int64_t)d.m) ^
               murmurhash64((uint64_t)(int64_t)d.d);
    }
    case OT_DECIMAL: { /* h(neg)^h(coef)^h(scale) (hash.go:144-151) */
        odec d = dec_from_unscaled(((const int64_t *)c->data)[r], c->scale);
        return murmurhash64(d.neg) ^ murmurhash64(d.coef) ^ murmurhash64((uint64_t)(int64_t)d.scale);
    }
    case OT_CODE8: {
        const char *s = c->dict[((const uint8_t *)c->data)[r]];
        return hash_bytes((const uint8_t *)s, strlen(s));
    }
    case OT_VARCHAR: {
        const int32_t *off = (const int32_t *)c->data;
        return hash_bytes((const uint8_t *)c->dict + off[r], (uint64_t)(off[r + 1] - off[r]));
    }
    default: abort();
    }
}

void oracle_hash(const ocol *cols, int32_t ncols, int64_t n, uint64_t *out) {
    for (int64_t r = 0; r < n; r++) {
        uint64_t h = hash_value(&cols[0], r);
        for (int32_t c = 1; c < ncols; c++) h = combine_hash(h, hash_value(&cols[c], r));
        out[r] = h;
    }
}

/* ------------------------------------------------------------------ decimal expressions */

int oracle_eval_decimal(const ocol *cols, const orpn *prog, int32_t nprog, const int64_t *sel,
                        int64_t n, odec *out) {
    odec stack[16];
    for (int64_t base = 0; base < n; base += VS) { /* one chunk per executeExprs call */
        int64_t cnt = n - base < VS ? n - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t r = sel ? sel[base + j] : base + j;
            int sp = 0;
            for (int32_t p = 0; p < nprog; p++) {
                const orpn *o = &prog[p];
                int rc = ODEC_OK;
                switch (o->op) {
                case OX_COL: {
                    const ocol *c = &cols[o->col];
                    if (c->type == OT_DECIMAL)
                        stack[sp++] = dec_from_unscaled(((const int64_t *)c->data)[r], c->scale);
                    else if (c->type == OT_INT32)
                        rc = odec_new_from_int64(((const int32_t *)c->data)[r], 0, 0, &stack[sp++]);
                    else
                        return -2;
                    break;
                }
                case OX_CONST_INT: rc = odec_new_from_int64(o->ival, 0, o->scale, &stack[sp++]); break;
                case OX_CONST_DEC: stack[sp++] = dec_from_unscaled(o->ival, o->scale); break;
                case OX_ADD: sp--; rc = odec_add(stack[sp - 1], stack[sp], &stack[sp - 1]); break;
                case OX_SUB: sp--; rc = odec_sub(stack[sp - 1], stack[sp], &stack[sp - 1]); break;
                case OX_MUL: sp--; rc = odec_mul(stack[sp - 1], stack[sp], &stack[sp - 1]); break;
                default: return -2;
                }
                if (rc != ODEC_OK) return rc;
            }
            out[base + j] = stack[0];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ hash aggregate */

typedef struct { /* aggrHTEntry (aggregate_hash.go:14-18) */
    uint16_t salt;
    uint16_t page_offset;
    uint32_t page_nr; /* 0 = empty */
} aggr_entry;

typedef struct { /* State[T] (function_aggr.go:420-425) */
    uint8_t isset;
    uint64_t count;
    ohuge h;
    odec d;
    double f;
} agg_state;

#define AGG_MAX_KEYS 8
#define AGG_MAX_AGGS 16

struct oagg {
    int32_t nkeys, naggs;
    int32_t key_type[AGG_MAX_KEYS];
    const char *const *key_dict[AGG_MAX_KEYS];
    oaggspec aggs[AGG_MAX_AGGS];
    int32_t arg_type[AGG_MAX_AGGS]; /* input type of each aggregate (OT_INT32/OT_ODEC/0=count(*)) */
    int64_t cap, count;
    uint64_t bitmask;
    aggr_entry *entries;
    int tuples_per_block;
    /* payload tuples in insertion order: [null bits | group values | hash | states] */
    int64_t groups_cap;
    int64_t *first_row;
    int64_t *key_vals;   /* count * nkeys */
    uint8_t *key_null;   /* count * nkeys */
    uint64_t *hashes;
    agg_state *states;   /* count * naggs */
    int error;
};

#define AGG_BLOCK_SIZE (256 * 1024 - 8) /* aggregate.go:62-66 */
#define HASH_PREFIX_SHIFT 48 /* (HASH_WIDTH-2)*8, aggregate_hash.go:126 */

static void aggr_resize(oagg *t, int64_t size) { /* Resize (aggregate_hash.go:440-513) */
    t->cap = size;
    t->bitmask = (uint64_t)size - 1;
    free(t->entries);
    t->entries = (aggr_entry *)calloc((size_t)size, sizeof(aggr_entry));
    for (int64_t g = 0; g < t->count; g++) {
        uint64_t idx = t->hashes[g] & t->bitmask;
        while (t->entries[idx].page_nr > 0) {
            idx++;
            if (idx >= (uint64_t)t->cap) idx = 0;
        }
        t->entries[idx].salt = (uint16_t)(t->hashes[g] >> HASH_PREFIX_SHIFT);
        t->entries[idx].page_nr = (uint32_t)(1 + g / t->tuples_per_block);
        t->entries[idx].page_offset = (uint16_t)(g % t->tuples_per_block);
    }
}

oagg *oracle_agg_create(const ocol *key_proto, int32_t nkeys, const ocol *arg_proto,
                        const oaggspec *aggs, int32_t naggs) {
    if (nkeys > AGG_MAX_KEYS || naggs > AGG_MAX_AGGS || nkeys < 1) return NULL;
    oagg *t = (oagg *)calloc(1, sizeof *t);
    t->nkeys = nkeys;
    t->naggs = naggs;
    for (int32_t c = 0; c < nkeys; c++) {
        t->key_type[c] = key_proto[c].type;
        t->key_dict[c] = key_proto[c].dict;
    }
    for (int32_t a = 0; a < naggs; a++) {
        t->aggs[a] = aggs[a];
        t->arg_type[a] = aggs[a].arg >= 0 ? arg_proto[aggs[a].arg].type : 0;
    }
    /* the row width only decides how many tuples share one 256 KiB block, i.e. how a group's
     * ordinal splits into (pageNr, pageOffset); it has no effect on results */
    int row_width = 8 + 8 * nkeys + 8 + 48 * naggs;
    t->tuples_per_block = AGG_BLOCK_SIZE / row_width;
    t->groups_cap = 4096;
    size_t na = (size_t)(naggs ? naggs : 1);
    t->first_row = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->groups_cap);
    t->key_vals = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->groups_cap * (size_t)nkeys);
    t->key_null = (uint8_t *)malloc((size_t)t->groups_cap * (size_t)nkeys);
    t->hashes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)t->groups_cap);
    t->states = (agg_state *)calloc((size_t)t->groups_cap * na, sizeof(agg_state));
    aggr_resize(t, 2 * VS); /* initial capacity 2*DefaultVectorSize (aggregate_exec.go:332-339) */
    return t;
}

void oracle_agg_free(oagg *t) {
    if (!t) return;
    free(t->entries); free(t->first_row); free(t->key_vals); free(t->key_null);
    free(t->hashes); free(t->states); free(t);
}

static int64_t key_value(const ocol *k, int64_t i) {
    switch (k->type) {
    case OT_INT32: case OT_DATE: return ((const int32_t *)k->data)[i];
    case OT_INT64: case OT_DECIMAL: return ((const int64_t *)k->data)[i];
    case OT_CODE8: return ((const uint8_t *)k->data)[i];
    case OT_CONST32: return ((const int32_t *)k->data)[0];
    default: abort();
    }
}

static int key_match(const oagg *t, const ocol *keys, int64_t i, int64_t g) {
    /* Match/TemplatedMatchType (util_match.go:25-301) with FuncEqual predicates on the group
     * columns: NULL matches NULL, NULL never matches a value */
    for (int32_t c = 0; c < t->nkeys; c++) {
        int v = keys[c].type == OT_CONST32 ? 1 : row_valid(keys[c].validity, i);
        int gn = t->key_null[g * t->nkeys + c];
        if (v == gn) return 0; /* one NULL, one not */
        if (!v) continue;
        int64_t a = key_value(&keys[c], i), b = t->key_vals[g * t->nkeys + c];
        if (a == b) continue;
        if (keys[c].type == OT_CODE8 && strcmp(keys[c].dict[a], keys[c].dict[b]) == 0) continue;
        return 0;
    }
    return 1;
}

static void huge_add_value(ohuge *r, uint64_t value, int positive) {
    /* HugeintAdd.addValue (function_aggr.go:623-632) */
    r->lower += value;
    int overflow = r->lower < value;
    if ((overflow ^ positive) == 0) r->upper += -1 + 2 * (int64_t)positive;
}

static int state_update(agg_state *s, int32_t kind, const ocol *arg, int64_t i) {
    /* UnaryScatter/UnaryFlatLoop/UnaryScatterLoop (function_aggr.go:1034-1161): NULL inputs are
     * skipped (IgnoreNull); Sum/Avg/Count Operation = AddValues + AddNumber (:790-800,
     * :862-871, :926-936); MinMaxOp.Operation (:988-1001) */
    if (!row_valid(arg->validity, i)) return 0;
    switch (kind) {
    case OA_SUM:
        s->isset = 1;
        s->count++;
        if (arg->type == OT_INT32) {
            int32_t v = ((const int32_t *)arg->data)[i];
            huge_add_value(&s->h, (uint64_t)(int64_t)v, v >= 0);
        } else if (arg->type == OT_ODEC) {
            return odec_add(s->d, ((const odec *)arg->data)[i], &s->d); /* DecimalAdd (:687-689) */
        } else abort();
        break;
    case OA_AVG:
        s->count++;
        if (arg->type == OT_INT32) s->f += (double)((const int32_t *)arg->data)[i]; /* :733-738 */
        else if (arg->type == OT_ODEC) return odec_add(s->d, ((const odec *)arg->data)[i], &s->d);
        else abort();
        break;
    case OA_COUNT:
        /* CountOp also runs AddNumber on a dummy Hugeint (:926-936); it never reaches the output */
        s->count++;
        break;
    case OA_MIN: case OA_MAX: {
        if (arg->type != OT_ODEC) abort();
        odec v = ((const odec *)arg->data)[i];
        if (!s->isset) { s->d = v; s->isset = 1; }
        else if (kind == OA_MAX ? odec_cmp(v, s->d) > 0 : odec_cmp(v, s->d) < 0) s->d = v;
        break;
    }
    default: abort();
    }
    return 0;
}

static int state_finalize(const agg_state *s, int32_t kind, int32_t arg_type, oaggval *out) {
    /* SumOp.Finalize (:813-823), AvgOp.Finalize (:873-900), CountOp.Finalize (:950-962),
     * MinMaxOp.Finalize (:1017-1027) */
    memset(out, 0, sizeof *out);
    switch (kind) {
    case OA_SUM:
        if (!s->isset) { out->kind = OV_NULL; return 0; }
        if (arg_type == OT_INT32) { out->kind = OV_HUGEINT; out->h = s->h; }
        else { out->kind = OV_DECIMAL; out->d = s->d; }
        return 0;
    case OA_AVG:
        if (s->count == 0) { out->kind = OV_NULL; return 0; }
        if (arg_type == OT_INT32) { out->kind = OV_DOUBLE; out->f = s->f / (double)s->count; }
        else {
            odec c;
            odec_new((int64_t)s->count, 0, &c);
            out->kind = OV_DECIMAL;
            return odec_quo(s->d, c, &out->d);
        }
        return 0;
    case OA_COUNT:
        if (s->count == 0) { out->kind = OV_NULL; return 0; } /* a zero count comes out NULL */
        out->kind = OV_HUGEINT;
        out->h.lower = s->count;
        return 0;
    case OA_MIN: case OA_MAX:
        if (!s->isset) { out->kind = OV_NULL; return 0; }
        out->kind = OV_DECIMAL;
        out->d = s->d;
        return 0;
    default: abort();
    }
}

static void agg_grow(oagg *t) {
    int64_t oc = t->groups_cap;
    t->groups_cap *= 2;
    size_t na = (size_t)(t->naggs ? t->naggs : 1);
    t->first_row = (int64_t *)realloc(t->first_row, sizeof(int64_t) * (size_t)t->groups_cap);
    t->key_vals = (int64_t *)realloc(t->key_vals, sizeof(int64_t) * (size_t)t->groups_cap * (size_t)t->nkeys);
    t->key_null = (uint8_t *)realloc(t->key_null, (size_t)t->groups_cap * (size_t)t->nkeys);
    t->hashes = (uint64_t *)realloc(t->hashes, sizeof(uint64_t) * (size_t)t->groups_cap);
    t->states = (agg_state *)realloc(t->states, sizeof(agg_state) * (size_t)t->groups_cap * na);
    memset(t->states + (size_t)oc * na, 0, sizeof(agg_state) * (size_t)(t->groups_cap - oc) * na);
}

int oracle_agg_sink(oagg *t, const ocol *keys, const ocol *args, const int64_t *row_ids,
                    int64_t cnt) {
    return oracle_agg_sink_filtered(t, keys, args, row_ids, cnt, 0xFFFFFFFFu);
}

int oracle_agg_sink_filtered(oagg *t, const ocol *keys, const ocol *args, const int64_t *row_ids,
                             int64_t cnt, uint32_t agg_mask) {
    uint64_t hashes[VS];
    uint64_t ht_off[VS];
    int64_t addr[VS]; /* group ordinal per row (the reference keeps row pointers) */
    int64_t cur[VS], nomatch[VS];
    if (cnt > VS) return -3;
    if (t->error) return t->error;
    if (cnt == 0) return 0;
    /* groups.Hash (AddChunk2, aggregate_hash.go:136-153) */
    for (int64_t j = 0; j < cnt; j++) {
        uint64_t h = hash_value(&keys[0], j);
        for (int32_t c = 1; c < t->nkeys; c++) h = combine_hash(h, hash_value(&keys[c], j));
        hashes[j] = h;
    }
    /* FindOrCreateGroups (:201-391); ResizeThreshold = int(float32(cap)/1.5) (:538-540) */
    if (t->cap - t->count <= cnt || t->count > (int64_t)((float)t->cap / 1.5f))
        aggr_resize(t, t->cap * 2);
    for (int64_t j = 0; j < cnt; j++) {
        ht_off[j] = hashes[j] & t->bitmask;
        cur[j] = j;
    }
    int64_t remaining = cnt;
    while (remaining > 0) {
        int64_t nm = 0;
        for (int64_t i = 0; i < remaining; i++) {
            int64_t j = cur[i];
            aggr_entry *e = &t->entries[ht_off[j]];
            uint16_t salt = (uint16_t)(hashes[j] >> HASH_PREFIX_SHIFT);
            if (e->page_nr == 0) {
                /* empty cell: claim it, append a new tuple, InitStates */
                if (t->count == t->groups_cap) agg_grow(t);
                int64_t g = t->count++;
                t->first_row[g] = row_ids ? row_ids[j] : j;
                t->hashes[g] = hashes[j];
                for (int32_t c = 0; c < t->nkeys; c++) {
                    int v = keys[c].type == OT_CONST32 ? 1 : row_valid(keys[c].validity, j);
                    t->key_null[g * t->nkeys + c] = (uint8_t)!v;
                    t->key_vals[g * t->nkeys + c] = v ? key_value(&keys[c], j) : 0;
                }
                e->salt = salt;
                e->page_nr = (uint32_t)(1 + g / t->tuples_per_block);
                e->page_offset = (uint16_t)(g % t->tuples_per_block);
                addr[j] = g;
            } else if (e->salt == salt) {
                int64_t g = (int64_t)(e->page_nr - 1) * t->tuples_per_block + e->page_offset;
                if (key_match(t, keys, j, g)) addr[j] = g;
                else nomatch[nm++] = j;
            } else {
                nomatch[nm++] = j;
            }
        }
        for (int64_t i = 0; i < nm; i++) { /* linear probing (:376-384) */
            int64_t j = nomatch[i];
            ht_off[j]++;
            if (ht_off[j] >= (uint64_t)t->cap) ht_off[j] = 0;
            cur[i] = j;
        }
        remaining = nm;
    }
    /* update loop (:176-198): one pass per aggregate over the chunk */
    for (int32_t a = 0; a < t->naggs; a++) {
        /* AddChunk's filter (:178-197): aggregates not listed are skipped, their states untouched */
        if (!((agg_mask >> a) & 1)) continue;
        const ocol *arg = t->aggs[a].arg >= 0 ? &args[t->aggs[a].arg] : NULL;
        for (int64_t j = 0; j < cnt; j++) {
            agg_state *s = &t->states[addr[j] * t->naggs + a];
            if (arg == NULL) { s->count++; continue; } /* count(*) */
            int rc = state_update(s, t->aggs[a].kind, arg, j);
            if (rc) { t->error = rc; return rc; }
        }
    }
    return 0;
}

int64_t oracle_agg_count(const oagg *t) { return t->count; }

int oracle_agg_group(const oagg *t, int64_t g, int64_t *first_row, int64_t *key_vals,
                     uint8_t *key_null, oaggval *vals) {
    if (g < 0 || g >= t->count) return -1;
    if (first_row) *first_row = t->first_row[g];
    for (int32_t c = 0; c < t->nkeys; c++) {
        if (key_vals) key_vals[c] = t->key_vals[g * t->nkeys + c];
        if (key_null) key_null[c] = t->key_null[g * t->nkeys + c];
    }
    for (int32_t a = 0; a < t->naggs; a++) {
        int rc = state_finalize(&t->states[g * t->naggs + a], t->aggs[a].kind, t->arg_type[a],
                                &vals[a]);
        if (rc) return rc;
    }
    return 0;
}

/* gather one chunk of a column into a positional buffer (what SliceIndice/DICT views amount to) */
static ocol gather_col(const ocol *c, const int64_t *sel, int64_t base, int64_t cnt, void *buf,
                       uint8_t *vbuf) {
    ocol o = *c;
    if (c->type == OT_CONST32) return o;
    size_t w = (c->type == OT_INT32 || c->type == OT_DATE || c->type == OT_FLOAT) ? 4
               : (c->type == OT_CODE8) ? 1
               : (c->type == OT_ODEC) ? sizeof(odec) : 8;
    for (int64_t j = 0; j < cnt; j++) {
        int64_t r = sel ? sel[base + j] : base + j;
        memcpy((char *)buf + (size_t)j * w, (const char *)c->data + (size_t)r * w, w);
    }
    o.data = buf;
    if (c->validity) {
        memset(vbuf, 0, VS / 8);
        for (int64_t j = 0; j < cnt; j++) {
            int64_t r = sel ? sel[base + j] : base + j;
            if (row_valid(c->validity, r)) vbuf[j >> 3] |= (uint8_t)(1u << (j & 7));
        }
        o.validity = vbuf;
    }
    return o;
}

int64_t oracle_groupby(const ocol *keys, int32_t nkeys, const ocol *args, int32_t nargs,
                       const oaggspec *aggs, int32_t naggs, const int64_t *sel, int64_t n,
                       int64_t *group_first_row, int64_t *group_keys, uint8_t *group_key_null,
                       oaggval *vals, int64_t max_groups) {
    oagg *t = oracle_agg_create(keys, nkeys, args, aggs, naggs);
    if (!t) return -1;
    int ncols = nkeys + nargs;
    char *bufs = (char *)malloc((size_t)ncols * VS * sizeof(odec));
    uint8_t *vbufs = (uint8_t *)malloc((size_t)ncols * (VS / 8));
    int64_t rid[VS];
    ocol k[AGG_MAX_KEYS], a[AGG_MAX_AGGS];
    int rc = 0;
    for (int64_t base = 0; base < n && rc == 0; base += VS) {
        int64_t cnt = n - base < VS ? n - base : VS;
        for (int32_t c = 0; c < nkeys; c++)
            k[c] = gather_col(&keys[c], sel, base, cnt, bufs + (size_t)c * VS * sizeof(odec),
                              vbufs + (size_t)c * (VS / 8));
        for (int32_t c = 0; c < nargs; c++)
            a[c] = gather_col(&args[c], sel, base, cnt, bufs + (size_t)(nkeys + c) * VS * sizeof(odec),
                              vbufs + (size_t)(nkeys + c) * (VS / 8));
        for (int64_t j = 0; j < cnt; j++) rid[j] = sel ? sel[base + j] : base + j;
        rc = oracle_agg_sink(t, k, a, rid, cnt);
    }
    int64_t ng = rc ? -1 : t->count;
    for (int64_t g = 0; g < ng && g < max_groups; g++) {
        if (oracle_agg_group(t, g, &group_first_row[g], group_keys ? &group_keys[g * nkeys] : NULL,
                             group_key_null ? &group_key_null[g * nkeys] : NULL,
                             &vals[g * naggs]) != 0) { ng = -1; break; }
    }
    free(bufs);
    free(vbufs);
    oracle_agg_free(t);
    return ng;
}

/* ------------------------------------------------------------------ CASE */

int oracle_case_decimal(const ocol *cols, const ocol *when_col, int32_t when_op, const oconst *when_k,
                        const orpn *then_prog, int32_t nthen, const orpn *else_prog, int32_t nelse,
                        int64_t n, odec *out, uint8_t *out_null) {
    int64_t *tsel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int64_t *fsel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    odec *tmp = (odec *)malloc(sizeof(odec) * (size_t)(n ? n : 1));
    int rc = 0;
    /* execSelectExpr on the WHEN (:156-166): true rows, the rest are the false rows */
    int64_t tn = oracle_select(when_col, when_op, when_k, NULL, n, tsel), fn = 0, ti = 0;
    for (int64_t r = 0; r < n; r++) {
        if (ti < tn && tsel[ti] == r) ti++;
        else fsel[fn++] = r;
    }
    for (int64_t r = 0; r < n; r++) out_null[r] = 1;
    /* THEN on the true rows, filled at those rows (:182-204); ELSE on the others (:212-241).
     * A NULL input makes the value NULL: FillSwitch copies the validity bit (:603) */
    for (int pass = 0; pass < 2 && rc == 0; pass++) {
        const int64_t *sel = pass == 0 ? tsel : fsel;
        int64_t cnt = pass == 0 ? tn : fn;
        if (cnt == 0) continue;
        rc = oracle_eval_decimal(cols, pass == 0 ? then_prog : else_prog, pass == 0 ? nthen : nelse, sel, cnt, tmp);
        if (rc) break;
        const orpn *prog = pass == 0 ? then_prog : else_prog;
        int np = pass == 0 ? nthen : nelse;
        for (int64_t i = 0; i < cnt; i++) {
            int valid = 1;
            for (int p = 0; p < np; p++)
                if (prog[p].op == OX_COL && !row_valid(cols[prog[p].col].validity, sel[i])) valid = 0;
            out[sel[i]] = tmp[i];
            out_null[sel[i]] = (uint8_t)!valid;
        }
    }
    free(tsel); free(fsel); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ column OP column */

/* selectBinary over two FLAT vectors (function_operator_boolean.go:506-521 -> selectFlat :672-778 ->
 * selectFlatLoop :780-868): the same (type, op) table as against a constant; a NULL on either side never selects. */
int64_t oracle_select_cols(const ocol *a, int32_t op, const ocol *b, const int64_t *sel_in, int64_t n_in,
                           int64_t *sel_out) {
    enum { P_INT32, P_DATE, P_FLOAT, P_DOUBLE, P_DECIMAL, P_VARCHAR, P_INT64 };
    int phys;
    if (a->type == OT_INT32 && b->type == OT_INT32) phys = P_INT32;
    else if (a->type == OT_DATE && b->type == OT_DATE) phys = P_DATE;
    else if (a->type == OT_DECIMAL && b->type == OT_DECIMAL) phys = P_DECIMAL;
    else if (a->type == OT_INT64 && b->type == OT_INT64) phys = P_INT64;
    else return 0;
    if (!select_supported(phys, op)) return 0;
    int64_t out = 0;
    for (int64_t base = 0; base < n_in; base += VS) {
        int64_t cnt = n_in - base < VS ? n_in - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t r = sel_in ? sel_in[base + j] : base + j;
            if (!row_valid(a->validity, r) || !row_valid(b->validity, r)) continue;
            int c;
            if (phys == P_INT32) {
                int32_t x = ((const int32_t *)a->data)[r], y = ((const int32_t *)b->data)[r];
                c = x < y ? -1 : x > y;
            } else if (phys == P_DATE) {
                c = date_cmp(date_from_days(((const int32_t *)a->data)[r]), date_from_days(((const int32_t *)b->data)[r]));
            } else {
                c = odec_cmp(dec_from_unscaled(((const int64_t *)a->data)[r], a->scale), dec_from_unscaled(((const int64_t *)b->data)[r], b->scale));
            }
            if (cmp_result(c, op)) sel_out[out++] = r;
        }
    }
    return out;
}

/* ------------------------------------------------------------------ OR of comparisons */

int64_t oracle_select_or(const ocol *cols, const int32_t *ops, const oconst *ks, int32_t k,
                         const int64_t *sel_in, int64_t n_in, int64_t *sel_out) {
    /* curSel / curCount (expr_exec.go:490-491): the rows no child has accepted yet */
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_in ? n_in : 1));
    int64_t *t = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_in ? n_in : 1));
    int64_t cur_n = n_in, res = 0;
    for (int64_t i = 0; i < n_in; i++) cur[i] = sel_in ? sel_in[i] : i;
    for (int32_t c = 0; c < k; c++) {
        int64_t tn = oracle_select(&cols[c], ops[c], &ks[c], cur, cur_n, t);
        if (tn > 0) { /* :517-526: append the true rows, continue on the false ones */
            int64_t w = 0, ti = 0;
            for (int64_t i = 0; i < cur_n; i++) {
                if (ti < tn && t[ti] == cur[i]) { sel_out[res++] = cur[i]; ti++; }
                else cur[w++] = cur[i];
            }
            cur_n = w;
        }
    }
    free(cur);
    free(t);
    return res;
}

/* ------------------------------------------------------------------ ORDER BY */

static void enc_be32(uint8_t *p, int32_t v) { /* encodeInt32 (sort_encoder.go:98-101) */
    uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)(u >> 24) ^ 0x80; p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
}
static void enc_be64(uint8_t *p, int64_t v) { /* encodeInt64 (:107-110) */
    uint64_t u = (uint64_t)v;
    for (int b = 0; b < 8; b++) p[b] = (uint8_t)(u >> (56 - 8 * b));
    p[0] ^= 0x80;
}
static void civil_from_days(int32_t z, int32_t *y, int32_t *m, int32_t *d) {
    z += 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    int32_t yy = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    uint32_t mp = (5 * doy + 2) / 153;
    *d = (int32_t)(doy - (153 * mp + 2) / 5 + 1);
    *m = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    *y = yy + (*m <= 2);
}

typedef struct { const uint8_t *key; int64_t pos; int32_t len; } sort_ref;
static int sort_ref_cmp(const void *a, const void *b) {
    const sort_ref *x = (const sort_ref *)a, *y = (const sort_ref *)b;
    int c = memcmp(x->key, y->key, (size_t)x->len);
    if (c) return c;
    return x->pos < y->pos ? -1 : x->pos > y->pos;
}

int oracle_sort_rows(const ocol *cols, const int32_t *descending, int32_t nkeys, const int64_t *sel,
                     int64_t n, int64_t *rows_out, int32_t *key_len_out, uint8_t *keys_out) {
    int32_t width = 0;
    for (int32_t c = 0; c < nkeys; c++) {
        switch (cols[c].type) {
        case OT_INT32: width += 1 + 4; break;
        case OT_CODE8: width += 1 + 1; break;
        case OT_DATE: width += 1 + 12; break;
        case OT_DECIMAL: width += 1 + 16; break;
        default: return -2;
        }
    }
    if (key_len_out) *key_len_out = width;
    uint8_t *keys = (uint8_t *)calloc((size_t)(n ? n : 1), (size_t)width);
    sort_ref *refs = (sort_ref *)malloc(sizeof(sort_ref) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        int64_t r = sel ? sel[i] : i;
        uint8_t *p = keys + (size_t)i * (size_t)width;
        for (int32_t c = 0; c < nkeys; c++) {
            const ocol *k = &cols[c];
            int vw = k->type == OT_INT32 ? 4 : k->type == OT_CODE8 ? 1 : k->type == OT_DATE ? 12 : 16;
            if (!row_valid(k->validity, r)) { p[0] = 0; memset(p + 1, 0, (size_t)vw); p += 1 + vw; continue; }
            p[0] = 1;
            if (k->type == OT_INT32) enc_be32(p + 1, ((const int32_t *)k->data)[r]);
            else if (k->type == OT_CODE8) p[1] = ((const uint8_t *)k->data)[r];
            else if (k->type == OT_DATE) {
                int32_t y, m, d;
                civil_from_days(((const int32_t *)k->data)[r], &y, &m, &d);
                enc_be32(p + 1, y); enc_be32(p + 5, m); enc_be32(p + 9, d);
            } else {
                odec v = dec_from_unscaled(((const int64_t *)k->data)[r], k->scale);
                int64_t whole, frac;
                if (!odec_int64(v, 2, &whole, &frac)) { free(keys); free(refs); return -1; } /* `ok` of dec.Int64 */
                enc_be64(p + 1, whole); enc_be64(p + 9, frac);
            }
            if (descending[c]) for (int b = 1; b <= vw; b++) p[b] = (uint8_t)~p[b];
            p += 1 + vw;
        }
        refs[i].key = keys + (size_t)i * (size_t)width;
        refs[i].pos = i;
        refs[i].len = width;
    }
    qsort(refs, (size_t)n, sizeof(sort_ref), sort_ref_cmp);
    for (int64_t i = 0; i < n; i++) {
        rows_out[i] = sel ? sel[refs[i].pos] : refs[i].pos;
        if (keys_out) memcpy(keys_out + (size_t)i * (size_t)width, refs[i].key, (size_t)width);
    }
    free(keys);
    free(refs);
    return 0;
}

/* ------------------------------------------------------------------ hash join */

struct ojoin {
    int32_t nkeys;
    ocol keys[4];      /* build-side key columns (borrowed) */
    int64_t count;     /* rows kept (NULL keys dropped) */
    int64_t *row;      /* build row id per kept tuple */
    uint64_t *hash;    /* hash per tuple */
    int64_t *next;     /* chain: previous head, stored "in the row's hash slot" (:268-288) */
    int64_t *buckets;  /* pointer table: tuple index or -1 */
    int64_t cap;
    uint64_t bitmask;
};

static int keys_all_valid(const ocol *keys, int32_t nkeys, int64_t r) {
    for (int32_t c = 0; c < nkeys; c++)
        if (!row_valid(keys[c].validity, r)) return 0;
    return 1;
}

static uint64_t next_pow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

ojoin *oracle_join_build(const ocol *keys, int32_t nkeys, const int64_t *sel, int64_t n) {
    if (nkeys > 4) return NULL;
    ojoin *j = (ojoin *)calloc(1, sizeof *j);
    j->nkeys = nkeys;
    memcpy(j->keys, keys, sizeof(ocol) * (size_t)nkeys);
    j->row = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    j->hash = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    /* Build (join_table.go:85-137): per chunk, drop NULL keys (prepareKeys :152-195), hash, append */
    for (int64_t i = 0; i < n; i++) {
        int64_t r = sel ? sel[i] : i;
        if (!keys_all_valid(keys, nkeys, r)) continue;
        uint64_t h = hash_value(&keys[0], r);
        for (int32_t c = 1; c < nkeys; c++) h = combine_hash(h, hash_value(&keys[c], r));
        j->row[j->count] = r;
        j->hash[j->count] = h;
        j->count++;
    }
    /* Finalize (:208-246): pointer table cap = max(nextpow2(2n), 1024) (:197-199) */
    uint64_t cap = next_pow2((uint64_t)j->count * 2);
    if (cap < 1024) cap = 1024;
    j->cap = (int64_t)cap;
    j->bitmask = cap - 1;
    j->buckets = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    for (uint64_t b = 0; b < cap; b++) j->buckets[b] = -1;
    j->next = (int64_t *)malloc(sizeof(int64_t) * (size_t)(j->count ? j->count : 1));
    for (int64_t t = 0; t < j->count; t++) { /* InsertHashesLoop (:268-288): head insertion */
        uint64_t b = j->hash[t] & j->bitmask;
        j->next[t] = j->buckets[b];
        j->buckets[b] = t;
    }
    return j;
}

void oracle_join_free(ojoin *j) {
    if (!j) return;
    free(j->row); free(j->hash); free(j->next); free(j->buckets); free(j);
}

int64_t oracle_join_count(const ojoin *j) { return j->count; }

static int join_key_equal(const ojoin *j, const ocol *pk, int64_t probe_row, int64_t build_row) {
    for (int32_t c = 0; c < j->nkeys; c++) {
        const ocol *b = &j->keys[c], *p = &pk[c];
        switch (b->type) {
        case OT_INT32: case OT_DATE:
            if (((const int32_t *)p->data)[probe_row] != ((const int32_t *)b->data)[build_row]) return 0;
            break;
        case OT_INT64: case OT_DECIMAL:
            if (((const int64_t *)p->data)[probe_row] != ((const int64_t *)b->data)[build_row]) return 0;
            break;
        case OT_CODE8:
            if (strcmp(p->dict[((const uint8_t *)p->data)[probe_row]],
                       b->dict[((const uint8_t *)b->data)[build_row]]) != 0) return 0;
            break;
        default: abort();
        }
    }
    return 1;
}

static void probe_chunk(const ojoin *j, const ocol *keys, int32_t nkeys, const int64_t *sel,
                        int64_t base, int64_t cnt, int64_t *out_probe, int64_t *out_build,
                        int64_t max, int64_t *total, uint8_t *found) {
    int64_t ptr[VS];
    int64_t live[VS];
    int64_t nlive = 0;
    /* Probe (:324-336): NULL keys dropped, hash, bucket heads (ApplyBitmask2 :303-322),
     * initSelVec keeps rows whose bucket is non-empty (join_scan.go:30-45) */
    for (int64_t i = 0; i < cnt; i++) {
        int64_t r = sel ? sel[base + i] : base + i;
        if (!keys_all_valid(keys, nkeys, r)) continue;
        uint64_t h = hash_value(&keys[0], r);
        for (int32_t c = 1; c < nkeys; c++) h = combine_hash(h, hash_value(&keys[c], r));
        ptr[i] = j->buckets[h & j->bitmask];
        if (ptr[i] >= 0) live[nlive++] = i;
    }
    /* InnerJoin rounds (:235-261): emit every live row whose current tuple matches, then
     * advancePointers (:263-278) for ALL live rows */
    while (nlive > 0) {
        for (int64_t q = 0; q < nlive; q++) {
            int64_t i = live[q];
            int64_t r = sel ? sel[base + i] : base + i;
            int64_t brow = j->row[ptr[i]];
            if (join_key_equal(j, keys, r, brow)) {
                if (found) found[base + i] = 1;
                if (out_probe && *total < max) {
                    out_probe[*total] = r;
                    out_build[*total] = brow;
                }
                (*total)++;
            }
        }
        int64_t nn = 0;
        for (int64_t q = 0; q < nlive; q++) {
            int64_t i = live[q];
            ptr[i] = j->next[ptr[i]];
            if (ptr[i] >= 0) live[nn++] = i;
        }
        nlive = nn;
    }
}

int64_t oracle_join_probe_inner(const ojoin *j, const ocol *keys, int32_t nkeys,
                                const int64_t *sel, int64_t n, int64_t *out_probe,
                                int64_t *out_build, int64_t max) {
    int64_t total = 0;
    if (j->count == 0) return 0;
    for (int64_t base = 0; base < n; base += VS) {
        int64_t cnt = n - base < VS ? n - base : VS;
        probe_chunk(j, keys, nkeys, sel, base, cnt, out_probe, out_build, max, &total, NULL);
    }
    return total;
}

void oracle_join_probe_mark(const ojoin *j, const ocol *keys, int32_t nkeys, const int64_t *sel,
                            int64_t n, uint8_t *found) {
    memset(found, 0, (size_t)n);
    if (j->count == 0) return;
    int64_t total = 0;
    for (int64_t base = 0; base < n; base += VS) {
        int64_t cnt = n - base < VS ? n - base : VS;
        probe_chunk(j, keys, nkeys, sel, base, cnt, NULL, NULL, 0, &total, found);
    }
}

/* ---- substring: substringStartEnd / substringFunc, function_operator_binary.go:553-625 ---- */
static int substring_start_end(int64_t slen, int64_t offset, int64_t length, int64_t *start, int64_t *end) {
    if (length == 0) return 0;
    if (offset > 0) {
        *start = slen < offset - 1 ? slen : offset - 1;       /* from the start */
    } else if (offset < 0) {
        *start = slen + offset > 0 ? slen + offset : 0;       /* from the end */
    } else {
        *start = 0;
        length--;
        if (length <= 0) return 0;
    }
    if (length > 0) {
        *end = slen < *start + length ? slen : *start + length;   /* left -> right */
    } else {
        *end = *start;                                            /* right -> left */
        *start = *start + length > 0 ? *start + length : 0;
    }
    return *start != *end;
}

int64_t oracle_substring(const char *s, int64_t slen, int64_t offset, int64_t length, char *out) {
    int64_t start = 0, end = 0;
    if (!substring_start_end(slen, offset, length, &start, &end)) return 0;
    memcpy(out, s + start, (size_t)(end - start));
    return end - start;
}

/* ---- cross product: CrossProductExec.Execute, join_cross.go:109-230 ---- */
void oracle_cross_pairs(int64_t n_left, int64_t n_right, int64_t chunk, int64_t *out_l, int64_t *out_r) {
    int64_t k = 0;
    for (int64_t base = 0; base < n_left; base += chunk) {           /* one LHS chunk at a time */
        int64_t card = n_left - base < chunk ? n_left - base : chunk;
        for (int64_t r = 0; r < n_right; r++)                         /* NextValue: the next RHS row */
            for (int64_t i = 0; i < card; i++) {                      /* LHS columns referenced, RHS row constant */
                out_l[k] = base + i;
                out_r[k] = r;
                k++;
            }
    }
}
