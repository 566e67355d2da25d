/* ORACLE — TEST INFRASTRUCTURE ONLY. See oracle.h.
 * Query drivers, round 3: TPC-H Q4, Q5, Q12, Q14, Q19 as physical pipelines over the operator restatements of
 * oracle.c — filter (execSelectAnd / execSelectOr, expr_exec.go:444-530; column-vs-column selectBinary), hash join
 * build / probe incl. the SEMI variant (join_scan.go:90-120, 166-180), CASE (executeCase, expr_exec.go:144-246), hash
 * aggregate, and the FLOAT arithmetic Q14's select list binds to (function_scalar.go:476-512, 960-1010;
 * function_cast.go:349-354). The join ORDER is a planner's choice (the result does not depend on it); the value
 * semantics are the reference's. Pinned by cases/tpch/1g/plan/q{4,5,12,14,19}.txt through tests/test_golden_tpch.py. */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VS ORACLE_VECTOR_SIZE

static ocol mkcol(int32_t type, int32_t scale, const void *data) {
    ocol c;
    memset(&c, 0, sizeof c);
    c.type = type;
    c.scale = scale;
    c.data = data;
    return c;
}
static ocol mkcode(const uint8_t *data, const char *const *dict) {
    ocol c = mkcol(OT_CODE8, 0, data);
    c.dict = dict;
    return c;
}
static oconst kdate(int32_t d) { oconst k; memset(&k, 0, sizeof k); k.type = OT_DATE; k.i = d; return k; }
static oconst kint(int64_t v) { oconst k; memset(&k, 0, sizeof k); k.type = OT_INT32; k.i = v; return k; }
static oconst kstr(const char *s) { oconst k; memset(&k, 0, sizeof k); k.type = OT_VARCHAR; k.s = s; return k; }
static int64_t *i64buf(int64_t n) { return (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1)); }

/* e * (1 - d) */
static const orpn DISC_PRICE[5] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0}, {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}};

/* ------------------------------------------------------------------ Q4
 * Agg(o_orderpriority; count(*)) <- SemiJoin(o_orderkey = l_orderkey) probe Scan(orders, date range)
 *                                       build Scan(lineitem, l_commitdate < l_receiptdate)        (EXISTS subquery) */
int64_t oracle_q4(const oracle_tpch *T, int32_t date_ge, int32_t date_lt, oracle_q4_row *out, int64_t max) {
    int64_t *lsel = i64buf(T->n_lineitem);
    ocol lc = mkcol(OT_DATE, 0, T->l_commitdate), lr = mkcol(OT_DATE, 0, T->l_receiptdate);
    int64_t nl = oracle_select_cols(&lc, OP_LT, &lr, NULL, T->n_lineitem, lsel);
    ocol lkey = mkcol(OT_INT64, 0, T->l_orderkey);
    ojoin *j = oracle_join_build(&lkey, 1, lsel, nl);
    int64_t *o1 = i64buf(T->n_orders), *o2 = i64buf(T->n_orders);
    ocol od = mkcol(OT_DATE, 0, T->o_orderdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t n1 = oracle_select(&od, OP_GE, &k1, NULL, T->n_orders, o1);
    int64_t n2 = oracle_select(&od, OP_LT, &k2, o1, n1, o2);
    uint8_t *found = (uint8_t *)malloc((size_t)(n2 > 0 ? n2 : 1));
    ocol okey = mkcol(OT_INT64, 0, T->o_orderkey);
    oracle_join_probe_mark(j, &okey, 1, o2, n2, found);      /* NextSemiOrAntiJoin: the probe rows with a match */
    oracle_join_free(j);
    ocol kproto[1] = {mkcode(NULL, T->orderpriority_dict)};
    oaggspec aggs[1] = {{OA_COUNT, -1}};
    oagg *t = oracle_agg_create(kproto, 1, NULL, aggs, 1);
    uint8_t pr[VS];
    int rc = 0;
    int64_t m = 0;
    for (int64_t i = 0; i <= n2 && rc == 0; i++) {
        if (i < n2 && found[i]) pr[m++] = T->o_orderpriority[o2[i]];
        if (m == VS || (i == n2 && m > 0)) {
            ocol keys[1] = {mkcode(pr, T->orderpriority_dict)};
            rc = oracle_agg_sink(t, keys, NULL, NULL, m);
            m = 0;
        }
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[1];
        oaggval v;
        oracle_agg_group(t, g, NULL, kv, NULL, &v);
        out[g].code = (int32_t)kv[0];
        out[g].count = v.h;
    }
    oracle_agg_free(t);
    free(lsel); free(o1); free(o2); free(found);
    return ng;
}

/* ------------------------------------------------------------------ Q5
 * Agg(n_name; sum(e*(1-d))) <- Join((l_suppkey, c_nationkey) = (s_suppkey, s_nationkey)) <- Join(l_orderkey = o_orderkey)
 *   probe lineitem, build <- Join(o_custkey = c_custkey) probe orders[date range], build <- Join(c_nationkey = n_nationkey)
 *   probe customer, build <- Join(n_regionkey = r_regionkey) probe nation, build region[r_name = ..] */
int64_t oracle_q5(const oracle_tpch *T, const char *region, int32_t date_ge, int32_t date_lt, oracle_q5_row *out, int64_t max) {
    int64_t rsel[5], np[25], nb[25];
    ocol rn = mkcode(T->r_name, T->region_dict);
    oconst kr = kstr(region);
    int64_t nr = oracle_select(&rn, OP_EQ, &kr, NULL, 5, rsel);
    ocol rk = mkcol(OT_INT32, 0, T->r_regionkey);
    ojoin *jr = oracle_join_build(&rk, 1, rsel, nr);
    ocol nrk = mkcol(OT_INT32, 0, T->n_regionkey);
    int64_t nn = oracle_join_probe_inner(jr, &nrk, 1, NULL, 25, np, nb, 25);     /* nations of the region */
    oracle_join_free(jr);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, np, nn);
    /* customer x nation */
    int64_t *c_row = i64buf(T->n_customer), *c_nat = i64buf(T->n_customer);
    ocol cn = mkcol(OT_INT32, 0, T->c_nationkey);
    int64_t nc = oracle_join_probe_inner(jn, &cn, 1, NULL, T->n_customer, c_row, c_nat, T->n_customer);
    oracle_join_free(jn);
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    ojoin *jc = oracle_join_build(&ck, 1, c_row, nc);          /* build row ids = customer rows */
    /* orders[date range] x customer' */
    int64_t *o1 = i64buf(T->n_orders), *o2 = i64buf(T->n_orders);
    ocol od = mkcol(OT_DATE, 0, T->o_orderdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t n1 = oracle_select(&od, OP_GE, &k1, NULL, T->n_orders, o1);
    int64_t n2 = oracle_select(&od, OP_LT, &k2, o1, n1, o2);
    int64_t *o_row = i64buf(n2), *o_cust = i64buf(n2);
    ocol oc = mkcol(OT_INT32, 0, T->o_custkey);
    int64_t no = oracle_join_probe_inner(jc, &oc, 1, o2, n2, o_row, o_cust, n2);
    oracle_join_free(jc);
    /* lineitem x orders' (the build side is the join output: positional keys) */
    int64_t *bk = i64buf(no);
    for (int64_t i = 0; i < no; i++) bk[i] = T->o_orderkey[o_row[i]];
    ocol bkc = mkcol(OT_INT64, 0, bk);
    ojoin *jo = oracle_join_build(&bkc, 1, NULL, no);
    int64_t cap = T->n_lineitem;
    int64_t *l_row = i64buf(cap), *l_ord = i64buf(cap);
    ocol lk = mkcol(OT_INT64, 0, T->l_orderkey);
    int64_t nl = oracle_join_probe_inner(jo, &lk, 1, NULL, T->n_lineitem, l_row, l_ord, cap);
    oracle_join_free(jo);
    /* ... x supplier on (l_suppkey, c_nationkey) = (s_suppkey, s_nationkey) */
    ocol sk[2] = {mkcol(OT_INT32, 0, T->s_suppkey), mkcol(OT_INT32, 0, T->s_nationkey)};
    ojoin *js = oracle_join_build(sk, 2, NULL, T->n_supplier);
    int32_t *p_supp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1)), *p_nat = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl > 0 ? nl : 1));
    for (int64_t i = 0; i < nl; i++) {
        p_supp[i] = T->l_suppkey[l_row[i]];
        p_nat[i] = T->c_nationkey[o_cust[l_ord[i]]];
    }
    ocol pk[2] = {mkcol(OT_INT32, 0, p_supp), mkcol(OT_INT32, 0, p_nat)};
    int64_t *f_pos = i64buf(nl), *f_sup = i64buf(nl);
    int64_t nf = oracle_join_probe_inner(js, pk, 2, NULL, nl, f_pos, f_sup, nl);
    oracle_join_free(js);
    /* aggregate by n_name */
    ocol kproto[1] = {mkcode(NULL, T->nation_dict)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    static odec v[VS];
    int64_t ext[VS], disc[VS];
    uint8_t name[VS];
    int rc = 0;
    for (int64_t base = 0; base < nf && rc == 0; base += VS) {
        int64_t cnt = nf - base < VS ? nf - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t pos = f_pos[base + j], l = l_row[pos];
            ext[j] = T->l_extendedprice[l];
            disc[j] = T->l_discount[l];
            name[j] = T->n_name[T->c_nationkey[o_cust[l_ord[pos]]]];   /* n_nationkey = its row (the fixed NATION table) */
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, cnt, v);
        ocol keys[1] = {mkcode(name, T->nation_dict)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[1];
        oaggval val;
        oracle_agg_group(t, g, NULL, kv, NULL, &val);
        out[g].nation = (int32_t)kv[0];
        out[g].revenue = val.d;
    }
    oracle_agg_free(t);
    free(c_row); free(c_nat); free(o1); free(o2); free(o_row); free(o_cust); free(bk); free(l_row); free(l_ord);
    free(p_supp); free(p_nat); free(f_pos); free(f_sup);
    (void)nb;
    return ng;
}

/* ------------------------------------------------------------------ Q12
 * Agg(l_shipmode; sum(case when prio = '1-URGENT' or prio = '2-HIGH' then 1 else 0 end), sum(case when prio <> .. and prio <> ..
 * then 1 else 0 end)) <- Join(l_orderkey = o_orderkey) probe Scan(lineitem, shipmode IN (..), commit < receipt, ship < commit,
 * receipt range), build Scan(orders) */
int64_t oracle_q12(const oracle_tpch *T, const char *mode1, const char *mode2, int32_t date_ge, int32_t date_lt, oracle_q12_row *out, int64_t max) {
    int64_t n = T->n_lineitem;
    int64_t *s1 = i64buf(n), *s2 = i64buf(n);
    ocol sm = mkcode(T->l_shipmode, T->shipmode_dict);
    ocol orc[2] = {sm, sm};
    int32_t ops[2] = {OP_EQ, OP_EQ};
    oconst ks[2] = {kstr(mode1), kstr(mode2)};
    int64_t c = oracle_select_or(orc, ops, ks, 2, NULL, n, s1);                 /* l_shipmode in (m1, m2) */
    ocol lc = mkcol(OT_DATE, 0, T->l_commitdate), lr = mkcol(OT_DATE, 0, T->l_receiptdate), ls = mkcol(OT_DATE, 0, T->l_shipdate);
    c = oracle_select_cols(&lc, OP_LT, &lr, s1, c, s2);                         /* l_commitdate < l_receiptdate */
    c = oracle_select_cols(&ls, OP_LT, &lc, s2, c, s1);                         /* l_shipdate < l_commitdate */
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    c = oracle_select(&lr, OP_GE, &k1, s1, c, s2);
    c = oracle_select(&lr, OP_LT, &k2, s2, c, s1);
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    ojoin *jo = oracle_join_build(&ok, 1, NULL, T->n_orders);
    int64_t *l_row = i64buf(c), *o_row = i64buf(c);
    ocol lk = mkcol(OT_INT64, 0, T->l_orderkey);
    int64_t nj = oracle_join_probe_inner(jo, &lk, 1, s1, c, l_row, o_row, c);
    oracle_join_free(jo);
    ocol kproto[1] = {mkcode(NULL, T->shipmode_dict)};
    ocol aproto[2] = {mkcol(OT_INT32, 0, NULL), mkcol(OT_INT32, 0, NULL)};
    oaggspec aggs[2] = {{OA_SUM, 0}, {OA_SUM, 1}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 2);
    uint8_t mode[VS], prio[VS];
    int32_t hi[VS], lo[VS];
    int64_t ts[VS], ts2[VS];
    int rc = 0;
    for (int64_t base = 0; base < nj && rc == 0; base += VS) {
        int64_t cnt = nj - base < VS ? nj - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            mode[j] = T->l_shipmode[l_row[base + j]];
            prio[j] = T->o_orderpriority[o_row[base + j]];
            hi[j] = lo[j] = 0;                                   /* ELSE 0 */
        }
        ocol pc = mkcode(prio, T->orderpriority_dict);
        ocol pcs[2] = {pc, pc};
        oconst ku[2] = {kstr("1-URGENT"), kstr("2-HIGH")};
        int32_t eq[2] = {OP_EQ, OP_EQ};
        int64_t th = oracle_select_or(pcs, eq, ku, 2, NULL, cnt, ts);          /* WHEN a OR b: THEN 1 at the true rows */
        for (int64_t i = 0; i < th; i++) hi[ts[i]] = 1;
        int64_t tl = oracle_select(&pc, OP_NE, &ku[0], NULL, cnt, ts);         /* WHEN a AND b */
        tl = oracle_select(&pc, OP_NE, &ku[1], ts, tl, ts2);
        for (int64_t i = 0; i < tl; i++) lo[ts2[i]] = 1;
        ocol keys[1] = {mkcode(mode, T->shipmode_dict)};
        ocol args[2] = {mkcol(OT_INT32, 0, hi), mkcol(OT_INT32, 0, lo)};
        rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[1];
        oaggval v[2];
        oracle_agg_group(t, g, NULL, kv, NULL, v);
        out[g].mode = (int32_t)kv[0];
        out[g].high = v[0].h;
        out[g].low = v[1].h;
    }
    oracle_agg_free(t);
    free(s1); free(s2); free(l_row); free(o_row);
    return ng;
}

/* ------------------------------------------------------------------ Q14
 * Agg(; sum(case when p_type like 'PROMO%' then e*(1-d) else 0 end), sum(e*(1-d))) <- Join(l_partkey = p_partkey)
 * probe Scan(lineitem, shipdate range), build Scan(part); the select list's 100.00 * a / b is FLOAT arithmetic */
int32_t oracle_q14(const oracle_tpch *T, const char *like_pattern, int32_t date_ge, int32_t date_lt, float *promo_revenue, odec *promo, odec *total) {
    int64_t n = T->n_lineitem;
    int64_t *s1 = i64buf(n), *s2 = i64buf(n);
    ocol ls = mkcol(OT_DATE, 0, T->l_shipdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t c = oracle_select(&ls, OP_GE, &k1, NULL, n, s1);
    c = oracle_select(&ls, OP_LT, &k2, s1, c, s2);
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, NULL, T->n_part);
    int64_t *l_row = i64buf(c), *p_row = i64buf(c);
    ocol lp = mkcol(OT_INT32, 0, T->l_partkey);
    int64_t nj = oracle_join_probe_inner(jp, &lp, 1, s2, c, l_row, p_row, c);
    oracle_join_free(jp);
    static const int32_t one = 1;
    ocol kproto[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol aproto[2] = {mkcol(OT_ODEC, 0, NULL), mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[2] = {{OA_SUM, 0}, {OA_SUM, 1}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 2);
    static odec vc[VS], va[VS];
    static uint8_t vnull[VS];
    int64_t ext[VS], disc[VS];
    uint8_t ty[VS];
    const orpn zero[1] = {{OX_CONST_INT, 0, 0, 0}};
    oconst kp = kstr(like_pattern);
    int rc = 0;
    for (int64_t base = 0; base < nj && rc == 0; base += VS) {
        int64_t cnt = nj - base < VS ? nj - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            ext[j] = T->l_extendedprice[l_row[base + j]];
            disc[j] = T->l_discount[l_row[base + j]];
            ty[j] = T->p_type[p_row[base + j]];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        ocol wc = mkcode(ty, T->type_dict);
        rc = oracle_case_decimal(cols, &wc, OP_LIKE, &kp, DISC_PRICE, 5, zero, 1, cnt, vc, vnull);
        if (rc == 0) rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, cnt, va);
        ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[2] = {mkcol(OT_ODEC, 0, vc), mkcol(OT_ODEC, 0, va)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int32_t res = rc ? -1 : 1;
    if (rc == 0 && oracle_agg_count(t) == 1) {
        int64_t kv[1];
        oaggval v[2];
        oracle_agg_group(t, 0, NULL, kv, NULL, v);
        if (v[0].kind == OV_DECIMAL && v[1].kind == OV_DECIMAL) {
            *promo = v[0].d;
            *total = v[1].d;
            /* 100.00 is a FLOAT literal; `*` and `/` resolve to their FLOAT overloads with both decimals cast
             * decimal -> float64 -> float32 (tryCastDecimalToFloat32); each operation rounds to float32 */
            volatile float a = (float)odec_float64(v[0].d), b = (float)odec_float64(v[1].d);
            volatile float m = 100.00f * a;
            *promo_revenue = m / b;
            res = 0;
        }
    }
    oracle_agg_free(t);
    free(s1); free(s2); free(l_row); free(p_row);
    return res;
}

/* ------------------------------------------------------------------ Q19
 * Agg(; sum(e*(1-d))) <- Filter(OR of three conjunctions over both sides) <- Join(l_partkey = p_partkey) probe lineitem, build part.
 * (p_partkey = l_partkey is common to the three OR branches: the join condition.) execSelectOr evaluates a branch on the
 * rows no earlier branch accepted; every branch is an execSelectAnd chain; IN lists are ORs of '='. */
typedef struct { const char *brand; const char *cntr[4]; int32_t qlo, qhi, smax; } q19_branch;

static int64_t and_step(const ocol *col, int32_t op, const oconst *k, int64_t *cur, int64_t n, int64_t *tmp) {
    int64_t m = oracle_select(col, op, k, cur, n, tmp);
    memcpy(cur, tmp, sizeof(int64_t) * (size_t)m);
    return m;
}
static int64_t or_step(const ocol *col, const char *const *vals, int nv, int64_t *cur, int64_t n, int64_t *tmp) {
    ocol cs[4];
    int32_t ops[4];
    oconst ks[4];
    for (int i = 0; i < nv; i++) { cs[i] = *col; ops[i] = OP_EQ; ks[i] = kstr(vals[i]); }
    int64_t m = oracle_select_or(cs, ops, ks, nv, cur, n, tmp);
    memcpy(cur, tmp, sizeof(int64_t) * (size_t)m);
    return m;
}

int32_t oracle_q19(const oracle_tpch *T, odec *revenue) {
    static const q19_branch BR[3] = {{"Brand#23", {"SM CASE", "SM BOX", "SM PACK", "SM PKG"}, 5, 15, 5},
                                     {"Brand#15", {"MED BAG", "MED BOX", "MED PKG", "MED PACK"}, 14, 24, 10},
                                     {"Brand#44", {"LG CASE", "LG BOX", "LG PACK", "LG PKG"}, 28, 38, 15}};
    static const char *const MODES[2] = {"AIR", "AIR REG"};
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, NULL, T->n_part);
    int64_t n = T->n_lineitem;
    int64_t *l_row = i64buf(n), *p_row = i64buf(n);
    ocol lp = mkcol(OT_INT32, 0, T->l_partkey);
    int64_t nj = oracle_join_probe_inner(jp, &lp, 1, NULL, n, l_row, p_row, n);
    oracle_join_free(jp);
    static const int32_t one = 1;
    ocol kproto[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    static odec v[VS];
    int64_t ext[VS], disc[VS], cur[VS], acc[VS], rest[VS], tmp[VS];
    int32_t qty[VS], size[VS];
    uint8_t brand[VS], cntr[VS], mode[VS], instr[VS];
    int rc = 0;
    for (int64_t base = 0; base < nj && rc == 0; base += VS) {
        int64_t cnt = nj - base < VS ? nj - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t l = l_row[base + j], p = p_row[base + j];
            ext[j] = T->l_extendedprice[l]; disc[j] = T->l_discount[l]; qty[j] = T->l_quantity[l];
            mode[j] = T->l_shipmode[l]; instr[j] = T->l_shipinstruct[l];
            brand[j] = T->p_brand[p]; cntr[j] = T->p_container[p]; size[j] = T->p_size[p];
        }
        ocol cb = mkcode(brand, T->brand_dict), cc = mkcode(cntr, T->container_dict), cm = mkcode(mode, T->shipmode_dict),
             ci = mkcode(instr, T->shipinstruct_dict), cq = mkcol(OT_INT32, 0, qty), cz = mkcol(OT_INT32, 0, size);
        int64_t nrest = cnt, nacc = 0;
        for (int64_t j = 0; j < cnt; j++) rest[j] = j;
        for (int b = 0; b < 3 && nrest > 0; b++) {     /* execSelectOr: branch b on the rows not accepted yet */
            int64_t m = nrest;
            memcpy(cur, rest, sizeof(int64_t) * (size_t)nrest);
            oconst kb = kstr(BR[b].brand), kq1 = kint(BR[b].qlo), kq2 = kint(BR[b].qhi), kz1 = kint(1), kz2 = kint(BR[b].smax), kin = kstr("DELIVER IN PERSON");
            m = and_step(&cb, OP_EQ, &kb, cur, m, tmp);
            m = or_step(&cc, BR[b].cntr, 4, cur, m, tmp);
            m = and_step(&cq, OP_GE, &kq1, cur, m, tmp);
            m = and_step(&cq, OP_LE, &kq2, cur, m, tmp);
            m = and_step(&cz, OP_GE, &kz1, cur, m, tmp);
            m = and_step(&cz, OP_LE, &kz2, cur, m, tmp);
            m = or_step(&cm, MODES, 2, cur, m, tmp);
            m = and_step(&ci, OP_EQ, &kin, cur, m, tmp);
            /* accepted rows leave `rest` (or_step may have reordered cur: membership test) */
            uint8_t hit[VS];
            memset(hit, 0, (size_t)cnt);
            for (int64_t i = 0; i < m; i++) { hit[cur[i]] = 1; acc[nacc++] = cur[i]; }
            int64_t w = 0;
            for (int64_t i = 0; i < nrest; i++) if (!hit[rest[i]]) rest[w++] = rest[i];
            nrest = w;
        }
        if (nacc == 0) continue;
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, DISC_PRICE, 5, acc, nacc, v);
        ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, nacc);
    }
    int32_t res = rc ? -1 : 1;
    if (rc == 0 && oracle_agg_count(t) == 1) {
        int64_t kv[1];
        oaggval val;
        oracle_agg_group(t, 0, NULL, kv, NULL, &val);
        if (val.kind == OV_DECIMAL) { *revenue = val.d; res = 0; }
    }
    oracle_agg_free(t);
    free(l_row); free(p_row);
    return res;
}

/* ------------------------------------------------------------------ Q18
 * Limit <- Order <- Agg(c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice; sum(l_quantity))
 *   <- Join(l_orderkey = o_orderkey) probe lineitem, build <- Join(o_custkey = c_custkey) probe <- SemiJoin(o_orderkey = l_orderkey)
 *      probe orders, build Filter(sum > k) <- Agg(l_orderkey; sum(l_quantity)) <- lineitem;   build customer */
int64_t oracle_q18(const oracle_tpch *T, const int64_t *o_totalprice, int64_t qty_gt, oracle_q18_row *out, int64_t max) {
    /* the subquery: group lineitem by l_orderkey, HAVING sum(l_quantity) > k (sum(INTEGER) is a HUGEINT; '>' exists for it) */
    ocol kproto[1] = {mkcol(OT_INT64, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *sub = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    int rc = 0;
    for (int64_t base = 0; base < T->n_lineitem && rc == 0; base += VS) {
        int64_t cnt = T->n_lineitem - base < VS ? T->n_lineitem - base : VS;
        ocol keys[1] = {mkcol(OT_INT64, 0, T->l_orderkey + base)};
        ocol args[1] = {mkcol(OT_INT32, 0, T->l_quantity + base)};
        rc = oracle_agg_sink(sub, keys, args, NULL, cnt);
    }
    if (rc) { oracle_agg_free(sub); return -1; }
    int64_t ngs = oracle_agg_count(sub), nbig = 0;
    int64_t *big = i64buf(ngs);
    for (int64_t g = 0; g < ngs; g++) {
        int64_t kv[1];
        oaggval v;
        oracle_agg_group(sub, g, NULL, kv, NULL, &v);
        if (v.kind == OV_HUGEINT && (v.h.upper > 0 || (v.h.upper == 0 && v.h.lower > (uint64_t)qty_gt))) big[nbig++] = kv[0];
    }
    oracle_agg_free(sub);
    ocol bk = mkcol(OT_INT64, 0, big);
    ojoin *js = oracle_join_build(&bk, 1, NULL, nbig);
    uint8_t *found = (uint8_t *)malloc((size_t)(T->n_orders > 0 ? T->n_orders : 1));
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    oracle_join_probe_mark(js, &ok, 1, NULL, T->n_orders, found);            /* SEMI: the orders with a qualifying key */
    oracle_join_free(js);
    int64_t *osel = i64buf(T->n_orders), no = 0;
    for (int64_t i = 0; i < T->n_orders; i++) if (found[i]) osel[no++] = i;
    /* x customer */
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    ojoin *jc = oracle_join_build(&ck, 1, NULL, T->n_customer);
    int64_t *o_row = i64buf(no), *c_row = i64buf(no);
    ocol oc = mkcol(OT_INT32, 0, T->o_custkey);
    int64_t n1 = oracle_join_probe_inner(jc, &oc, 1, osel, no, o_row, c_row, no);
    oracle_join_free(jc);
    /* lineitem x that */
    int64_t *bkey = i64buf(n1);
    for (int64_t i = 0; i < n1; i++) bkey[i] = T->o_orderkey[o_row[i]];
    ocol bkc = mkcol(OT_INT64, 0, bkey);
    ojoin *jo = oracle_join_build(&bkc, 1, NULL, n1);
    int64_t cap = T->n_lineitem;
    int64_t *l_row = i64buf(cap), *b_pos = i64buf(cap);
    ocol lk = mkcol(OT_INT64, 0, T->l_orderkey);
    int64_t n2 = oracle_join_probe_inner(jo, &lk, 1, NULL, T->n_lineitem, l_row, b_pos, cap);
    oracle_join_free(jo);
    /* final aggregate. c_name = 'Customer#' + nine digits of the key: a VARCHAR group key — hashed and compared as bytes through a
     * dictionary built over the (few) names that reach the aggregate */
    char (*names)[20] = (char (*)[20])malloc(sizeof(char[20]) * (size_t)(n1 > 0 ? n1 : 1));
    const char **dict = (const char **)malloc(sizeof(char *) * (size_t)(n1 > 0 ? n1 : 1));
    if (n1 > 255) { n2 = 0; rc = 1; }   /* the dictionary stand-in covers the handful of orders this query keeps */
    for (int64_t i = 0; i < n1 && i < 256; i++) { snprintf(names[i], 20, "Customer#%09d", T->c_custkey[c_row[i]]); dict[i] = names[i]; }
    ocol kp[5] = {mkcode(NULL, dict), mkcol(OT_INT32, 0, NULL), mkcol(OT_INT64, 0, NULL), mkcol(OT_DATE, 0, NULL), mkcol(OT_DECIMAL, 2, NULL)};
    ocol ap[1] = {mkcol(OT_INT32, 0, NULL)};
    oagg *t = oracle_agg_create(kp, 5, ap, aggs, 1);
    uint8_t nm[VS];
    int32_t ckv[VS], odv[VS], qty[VS];
    int64_t okv[VS], tpv[VS];
    for (int64_t base = 0; base < n2 && rc == 0; base += VS) {
        int64_t cnt = n2 - base < VS ? n2 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t b = b_pos[base + j], o = o_row[b];
            nm[j] = (uint8_t)b;                                   /* code of the order's customer name (one dictionary entry per kept order) */
            ckv[j] = T->c_custkey[c_row[b]];
            okv[j] = T->o_orderkey[o];
            odv[j] = T->o_orderdate[o];
            tpv[j] = o_totalprice[o];
            qty[j] = T->l_quantity[l_row[base + j]];
        }
        ocol keys[5] = {mkcode(nm, dict), mkcol(OT_INT32, 0, ckv), mkcol(OT_INT64, 0, okv), mkcol(OT_DATE, 0, odv), mkcol(OT_DECIMAL, 2, tpv)};
        ocol args[1] = {mkcol(OT_INT32, 0, qty)};
        rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[5];
        oaggval v;
        oracle_agg_group(t, g, NULL, kv, NULL, &v);
        out[g].c_custkey = (int32_t)kv[1];
        out[g].o_orderkey = kv[2];
        out[g].o_orderdate = (int32_t)kv[3];
        out[g].o_totalprice = kv[4];
        out[g].sum_qty = v.h;
    }
    oracle_agg_free(t);
    free(big); free(found); free(osel); free(o_row); free(c_row); free(bkey); free(l_row); free(b_pos); free(names); free(dict);
    return ng;
}

int64_t oracle_q18_text(oracle_q18_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap);

/* ------------------------------------------------------------------ Q17 (cases/tpch/query/q17.sql)
 * Project(sum / 7.0) <- Agg(; sum(l_extendedprice)) <- Filter(l_quantity < 0.2 * avg) <- Join(l_partkey = sub.l_partkey)
 *   [Join(l_partkey = p_partkey) probe Scan(lineitem), build Scan(part, p_brand = .. and p_container = ..)]
 *   x [Agg(l_partkey; avg(l_quantity)) <- Scan(lineitem)]   (the correlated subquery, decorrelated into an aggregate by its correlation key)
 * Typing (l_quantity is INTEGER in this schema): avg(INTEGER) accumulates and divides in float64 (AvgOp, aggregate_hash.go:733-738,
 * :873-900) -> DOUBLE; 0.2 is a FLOAT literal (bindAConst builder_binder.go:264-273) and `*` has only (T, T) overloads
 * (function_scalar.go:389-425), so it is cast float32 -> float64 (tryCastFloat32ToFloat64 function_cast.go:411-414) and multiplied in
 * float64; `<` has a DOUBLE overload (function_scalar.go:1429-1435, lessFloat64Op function_operator_boolean.go:466) and l_quantity is cast
 * int32 -> float64 (function_cast.go:332-335). sum(DECIMAL) / 7.0: `/` has FLOAT and DECIMAL overloads only (:478-500), the literal is
 * FLOAT, so the sum is cast decimal -> float64 -> float32 and divided in float32.
 * Returns 0 ok / 1 when the sum is NULL (no row passes) / -1 on error. */
int32_t oracle_q17(const oracle_tpch *T, const char *brand, const char *container, float fraction, float divisor, float *avg_yearly, odec *sum_out) {
    /* part[brand, container] */
    int64_t *ps1 = i64buf(T->n_part), *ps2 = i64buf(T->n_part);
    ocol pb = mkcode(T->p_brand, T->brand_dict), pc = mkcode(T->p_container, T->container_dict);
    oconst kb = kstr(brand), kc = kstr(container);
    int64_t np = oracle_select(&pb, OP_EQ, &kb, NULL, T->n_part, ps1);
    np = oracle_select(&pc, OP_EQ, &kc, ps1, np, ps2);
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, ps2, np);
    int64_t cap = T->n_lineitem;
    int64_t *l_row = i64buf(cap), *p_row = i64buf(cap);
    ocol lp = mkcol(OT_INT32, 0, T->l_partkey);
    int64_t nj = oracle_join_probe_inner(jp, &lp, 1, NULL, T->n_lineitem, l_row, p_row, cap);
    oracle_join_free(jp);
    /* the subquery: avg(l_quantity) per l_partkey over all of lineitem */
    ocol kproto[1] = {mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec aggs[1] = {{OA_AVG, 0}};
    oagg *sub = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    int rc = 0;
    for (int64_t base = 0; base < T->n_lineitem && rc == 0; base += VS) {
        int64_t cnt = T->n_lineitem - base < VS ? T->n_lineitem - base : VS;
        ocol keys[1] = {mkcol(OT_INT32, 0, T->l_partkey + base)};
        ocol args[1] = {mkcol(OT_INT32, 0, T->l_quantity + base)};
        rc = oracle_agg_sink(sub, keys, args, NULL, cnt);
    }
    int64_t ngs = rc ? 0 : oracle_agg_count(sub);
    int32_t *gkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ngs > 0 ? ngs : 1));
    double *gavg = (double *)malloc(sizeof(double) * (size_t)(ngs > 0 ? ngs : 1));
    for (int64_t g = 0; g < ngs; g++) {
        int64_t kv[1];
        oaggval v;
        oracle_agg_group(sub, g, NULL, kv, NULL, &v);
        gkey[g] = (int32_t)kv[0];
        gavg[g] = v.kind == OV_DOUBLE ? v.f : 0.0;
    }
    oracle_agg_free(sub);
    /* joined rows x the subquery's groups, then the Filter */
    ocol gk = mkcol(OT_INT32, 0, gkey);
    ojoin *jg = oracle_join_build(&gk, 1, NULL, ngs);
    int32_t *jkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nj > 0 ? nj : 1));
    for (int64_t i = 0; i < nj; i++) jkey[i] = T->l_partkey[l_row[i]];
    ocol jk = mkcol(OT_INT32, 0, jkey);
    int64_t *j_row = i64buf(nj), *g_row = i64buf(nj);
    int64_t n2 = oracle_join_probe_inner(jg, &jk, 1, NULL, nj, j_row, g_row, nj);
    oracle_join_free(jg);
    static const int32_t one = 1;
    ocol k1[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol a1[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec sum1[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(k1, 1, a1, sum1, 1);
    static odec ext[VS];
    int64_t fill = 0;
    const double frac64 = (double)fraction;
    for (int64_t i = 0; i <= n2 && rc == 0; i++) {
        if (i < n2) {
            const int64_t l = l_row[j_row[i]];
            volatile double thr = frac64 * gavg[g_row[i]];
            if ((double)T->l_quantity[l] < thr) odec_new(T->l_extendedprice[l], 2, &ext[fill++]);
        }
        if (fill == VS || (i == n2 && fill > 0)) {
            ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
            ocol args[1] = {mkcol(OT_ODEC, 0, ext)};
            rc = oracle_agg_sink(t, keys, args, NULL, fill);
            fill = 0;
        }
    }
    int32_t res = rc ? -1 : 1;
    if (rc == 0 && oracle_agg_count(t) == 1) {
        int64_t kv[1];
        oaggval v;
        oracle_agg_group(t, 0, NULL, kv, NULL, &v);
        if (v.kind == OV_DECIMAL) {
            *sum_out = v.d;
            volatile float a = (float)odec_float64(v.d);
            *avg_yearly = a / divisor;
            res = 0;
        }
    }
    oracle_agg_free(t);
    free(ps1); free(ps2); free(l_row); free(p_row); free(gkey); free(gavg); free(jkey); free(j_row); free(g_row);
    return res;
}

int64_t oracle_q17_text(float avg_yearly, int is_null, char *buf, int64_t cap);

/* ------------------------------------------------------------------ Q15 (cases/tpch/query/q15.sql)
 * Order(s_suppkey) <- Project <- Join(s_suppkey = supplier_no) probe Scan(supplier)
 *   build <- Join(total_revenue = max) [the scalar subquery's one row; `=` on DECIMAL exists only as a join condition: the hash join
 *            hashes and matches the decimals (chunk/hash.go, util_match.go), executeSelect has no DECIMAL case for FuncEqual
 *            (function_operator_boolean.go:395-407)]
 *        probe  q15_revenue0 = Agg(l_suppkey; sum(l_extendedprice * (1 - l_discount))) <- Scan(lineitem, l_shipdate in [d, d + 3 months))
 *        build  Agg(; max(total_revenue)) <- q15_revenue0        (MinMaxOp on DECIMAL: function_aggr.go:968-1027)
 * Returns the rows (s_suppkey, total_revenue) in supplier order, -1 on error. */
int64_t oracle_q15(const oracle_tpch *T, int32_t date_ge, int32_t date_lt, oracle_q15_row *out, int64_t max) {
    int64_t n = T->n_lineitem;
    int64_t *s1 = i64buf(n), *s2 = i64buf(n);
    ocol ls = mkcol(OT_DATE, 0, T->l_shipdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t c = oracle_select(&ls, OP_GE, &k1, NULL, n, s1);
    c = oracle_select(&ls, OP_LT, &k2, s1, c, s2);
    ocol kproto[1] = {mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *rev = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    static odec v[VS];
    int64_t ext[VS], disc[VS];
    int32_t supp[VS];
    int rc = 0;
    for (int64_t base = 0; base < c && rc == 0; base += VS) {
        int64_t cnt = c - base < VS ? c - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t r = s2[base + j];
            ext[j] = T->l_extendedprice[r]; disc[j] = T->l_discount[r]; supp[j] = T->l_suppkey[r];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, cnt, v);
        ocol keys[1] = {mkcol(OT_INT32, 0, supp)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(rev, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? 0 : oracle_agg_count(rev);
    int32_t *gkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1));
    odec *gval = (odec *)malloc(sizeof(odec) * (size_t)(ng > 0 ? ng : 1));
    /* the scalar subquery: max over the groups (an ungrouped aggregate over the CTE's rows) */
    static const int32_t one = 1;
    ocol k1p[1] = {mkcol(OT_CONST32, 0, &one)};
    oaggspec mx[1] = {{OA_MAX, 0}};
    oagg *tmax = oracle_agg_create(k1p, 1, aproto, mx, 1);
    for (int64_t g = 0; g < ng; g++) {
        int64_t kv[1];
        oaggval val;
        oracle_agg_group(rev, g, NULL, kv, NULL, &val);
        gkey[g] = (int32_t)kv[0];
        gval[g] = val.d;
    }
    for (int64_t base = 0; base < ng && rc == 0; base += VS) {
        int64_t cnt = ng - base < VS ? ng - base : VS;
        ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[1] = {mkcol(OT_ODEC, 0, gval + base)};
        rc = oracle_agg_sink(tmax, keys, args, NULL, cnt);
    }
    int64_t nout = rc ? -1 : 0;
    if (rc == 0 && oracle_agg_count(tmax) == 1) {
        int64_t kv[1];
        oaggval m;
        oracle_agg_group(tmax, 0, NULL, kv, NULL, &m);
        /* the groups whose revenue equals the maximum (Decimal.Equal), joined with supplier on the key, in supplier order */
        for (int64_t s = 0; s < T->n_supplier && m.kind == OV_DECIMAL; s++)
            for (int64_t g = 0; g < ng; g++)
                if (gkey[g] == T->s_suppkey[s] && odec_cmp(gval[g], m.d) == 0) {
                    if (nout < max) { out[nout].s_suppkey = gkey[g]; out[nout].total_revenue = gval[g]; }
                    nout++;
                }
    }
    oracle_agg_free(rev); oracle_agg_free(tmax);
    free(s1); free(s2); free(gkey); free(gval);
    return nout;
}

/* ------------------------------------------------------------------ Q22 (cases/tpch/query/q22.sql)
 * Order(cntrycode) <- Agg(cntrycode; count(*), sum(c_acctbal)) <- ANTI Join(c_custkey = o_custkey) probe
 *   Filter(c_acctbal > scalar) <- Project(substring(c_phone from 1 for 2) as cntrycode, ..) <- Scan(customer, cntrycode IN (..)), build Scan(orders);
 *   scalar = Agg(; avg(c_acctbal)) <- Scan(customer, c_acctbal > 0.00 and cntrycode IN (..)).
 * substring: substringFunc (function_operator_binary.go:553-625); IN binds to an OR of `=` on VARCHAR (equalStrOp); `c_acctbal > 0.00` is DECIMAL
 * against a FLOAT literal: float32 compare (greatFloat32Op); avg(DECIMAL) = govalues Quo(sum, count) (AvgOp.Finalize, function_aggr.go:873-900);
 * `c_acctbal > (subquery)` is DECIMAL > DECIMAL: greatDecimalOp, an exact comparison (function_operator_boolean.go:431-442); NOT EXISTS is the
 * ANTI join (join_scan.go:102-120). c_phone: 15 bytes per row. Returns the groups in first-seen order, -1 on error. */
int64_t oracle_q22(const oracle_tpch *T, const char *c_phone, const int64_t *c_acctbal, const char *const *codes, int32_t ncodes,
                   oracle_q22_row *out, int64_t max) {
    const int64_t n = T->n_customer;
    /* cntrycode as a dictionary column: the dictionary is the distinct substrings in first-seen order (at most 25 country codes) */
    static char dict_store[256][4];
    const char *dict[256];
    int32_t ndict = 0;
    uint8_t *code = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        char sub[16];
        const int64_t len = oracle_substring(c_phone + 15 * i, 15, 1, 2, sub);
        sub[len] = 0;
        int32_t k = 0;
        while (k < ndict && strcmp(dict[k], sub) != 0) k++;
        if (k == ndict) {
            if (ndict == 256 || len > 3) { free(code); return -1; }
            memcpy(dict_store[ndict], sub, (size_t)len + 1);
            dict[ndict] = dict_store[ndict];
            ndict++;
        }
        code[i] = (uint8_t)k;
    }
    ocol cc = mkcode(code, dict);
    cc.dict_size = ndict;
    int64_t *in_sel = i64buf(n), *s1 = i64buf(n), *s2 = i64buf(n);
    ocol orc[16]; int32_t ops[16]; oconst ks[16];
    if (ncodes > 16) { free(code); return -1; }
    for (int32_t k = 0; k < ncodes; k++) { orc[k] = cc; ops[k] = OP_EQ; ks[k] = kstr(codes[k]); }
    const int64_t nin = oracle_select_or(orc, ops, ks, ncodes, NULL, n, in_sel);
    /* the scalar subquery */
    ocol bal = mkcol(OT_DECIMAL, 2, c_acctbal);
    oconst zero; memset(&zero, 0, sizeof zero); zero.type = OT_FLOAT; zero.f = 0.00;
    const int64_t npos = oracle_select(&bal, OP_GT, &zero, in_sel, nin, s1);
    static const int32_t one = 1;
    ocol k1[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol a1[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec av[1] = {{OA_AVG, 0}};
    oagg *ta = oracle_agg_create(k1, 1, a1, av, 1);
    static odec v[VS];
    int rc = 0;
    for (int64_t base = 0; base < npos && rc == 0; base += VS) {
        int64_t cnt = npos - base < VS ? npos - base : VS;
        for (int64_t j = 0; j < cnt; j++) odec_new(c_acctbal[s1[base + j]], 2, &v[j]);
        ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        rc = oracle_agg_sink(ta, keys, args, NULL, cnt);
    }
    int64_t nout = -1;
    oaggval avg;
    memset(&avg, 0, sizeof avg);
    if (rc == 0 && oracle_agg_count(ta) == 1) { int64_t kv[1]; oracle_agg_group(ta, 0, NULL, kv, NULL, &avg); }
    oracle_agg_free(ta);
    if (rc == 0 && avg.kind == OV_DECIMAL) {
        /* c_acctbal > avg (exact), then the ANTI join */
        int64_t nf = 0;
        for (int64_t j = 0; j < nin; j++) {
            odec b;
            odec_new(c_acctbal[in_sel[j]], 2, &b);
            if (odec_cmp(b, avg.d) > 0) s2[nf++] = in_sel[j];
        }
        ocol ok = mkcol(OT_INT32, 0, T->o_custkey);
        ojoin *jo = oracle_join_build(&ok, 1, NULL, T->n_orders);
        uint8_t *found = (uint8_t *)malloc((size_t)(nf > 0 ? nf : 1));
        ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
        oracle_join_probe_mark(jo, &ck, 1, s2, nf, found);
        oracle_join_free(jo);
        ocol kp[1] = {cc};
        kp[0].data = NULL;
        oaggspec ag[2] = {{OA_COUNT, -1}, {OA_SUM, 0}};
        oagg *t = oracle_agg_create(kp, 1, a1, ag, 2);
        uint8_t kc[VS];
        int64_t fill = 0;
        for (int64_t j = 0; j <= nf && rc == 0; j++) {
            if (j < nf && !found[j]) { kc[fill] = code[s2[j]]; odec_new(c_acctbal[s2[j]], 2, &v[fill]); fill++; }
            if (fill == VS || (j == nf && fill > 0)) {
                ocol keys[1] = {cc};
                keys[0].data = kc;
                ocol args[1] = {mkcol(OT_ODEC, 0, v)};
                rc = oracle_agg_sink(t, keys, args, NULL, fill);
                fill = 0;
            }
        }
        nout = rc ? -1 : oracle_agg_count(t);
        for (int64_t g = 0; g < nout && g < max; g++) {
            int64_t kv[1];
            oaggval val[2];
            oracle_agg_group(t, g, NULL, kv, NULL, val);
            memset(out[g].cntrycode, 0, sizeof out[g].cntrycode);
            strncpy(out[g].cntrycode, dict[kv[0]], 3);
            out[g].numcust = val[0].h;
            out[g].totacctbal = val[1].d;
        }
        oracle_agg_free(t);
        free(found);
    }
    free(code); free(in_sel); free(s1); free(s2);
    return nout;
}

/* ------------------------------------------------------------------ Q20 (cases/tpch/query/q20.sql)
 * Order(s_name) <- Project(s_name, s_address) <- SEMI Join(s_suppkey = ps_suppkey) probe Join(s_nationkey = n_nationkey)[supplier, nation(n_name = ..)]
 *   build Filter(ps_availqty > 0.5 * sum) <- Join((ps_partkey, ps_suppkey) = (l_partkey, l_suppkey))
 *           probe SEMI Join(ps_partkey = p_partkey) probe Scan(partsupp), build Scan(part, p_name like 'lime%')
 *           build Agg(l_partkey, l_suppkey; sum(l_quantity)) <- Scan(lineitem, l_shipdate in [d, d + 1 year))   (the correlated subquery by its keys)
 * Typing: sum(INTEGER) is HUGEINT; 0.5 is a FLOAT literal, MaxLType(FLOAT, HUGEINT) = FLOAT: the sum is cast tryCastBigintToFloat32
 * (function_cast.go:365-374) and multiplied in float32; ps_availqty (INTEGER) > FLOAT compares in float32 (tryCastInt32ToFloat32,
 * greatFloat32Op). A partsupp row without lineitems in the year has a NULL subquery: the comparison is not true (the inner join drops it).
 * Returns the qualifying suppliers' keys in s_suppkey order (= ORDER BY s_name: the name is the zero-padded key), -1 on error. */
int64_t oracle_q20(const oracle_tpch *T, const int32_t *p_name_off, const char *p_name_bytes, int64_t n_ps, const int32_t *ps_partkey,
                   const int32_t *ps_suppkey, const int32_t *ps_availqty, const char *like_pattern, const char *nation, int32_t date_ge, int32_t date_lt,
                   float fraction, int32_t *out, int64_t max) {
    /* part[p_name like ..] */
    int64_t *psel = i64buf(T->n_part);
    ocol pname = mkcol(OT_VARCHAR, 0, p_name_off);
    pname.dict = (const char *const *)p_name_bytes;
    oconst kl = kstr(like_pattern);
    const int64_t np = oracle_select(&pname, OP_LIKE, &kl, NULL, T->n_part, psel);
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, psel, np);
    uint8_t *ps_hit = (uint8_t *)malloc((size_t)(n_ps > 0 ? n_ps : 1));
    ocol psp = mkcol(OT_INT32, 0, ps_partkey);
    oracle_join_probe_mark(jp, &psp, 1, NULL, n_ps, ps_hit);        /* SEMI: partsupp rows of those parts */
    oracle_join_free(jp);
    int64_t *ps_sel = i64buf(n_ps), nps = 0;
    for (int64_t i = 0; i < n_ps; i++) if (ps_hit[i]) ps_sel[nps++] = i;
    /* the subquery's aggregate */
    const int64_t n = T->n_lineitem;
    int64_t *s1 = i64buf(n), *s2 = i64buf(n);
    ocol ls = mkcol(OT_DATE, 0, T->l_shipdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t c = oracle_select(&ls, OP_GE, &k1, NULL, n, s1);
    c = oracle_select(&ls, OP_LT, &k2, s1, c, s2);
    ocol kproto[2] = {mkcol(OT_INT32, 0, NULL), mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *sub = oracle_agg_create(kproto, 2, aproto, aggs, 1);
    int32_t kp[VS], ksu[VS], q[VS];
    int rc = 0;
    for (int64_t base = 0; base < c && rc == 0; base += VS) {
        int64_t cnt = c - base < VS ? c - base : VS;
        for (int64_t j = 0; j < cnt; j++) { const int64_t r = s2[base + j]; kp[j] = T->l_partkey[r]; ksu[j] = T->l_suppkey[r]; q[j] = T->l_quantity[r]; }
        ocol keys[2] = {mkcol(OT_INT32, 0, kp), mkcol(OT_INT32, 0, ksu)};
        ocol args[1] = {mkcol(OT_INT32, 0, q)};
        rc = oracle_agg_sink(sub, keys, args, NULL, cnt);
    }
    const int64_t ng = rc ? 0 : oracle_agg_count(sub);
    int32_t *gp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1)), *gs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1));
    float *gsum = (float *)malloc(sizeof(float) * (size_t)(ng > 0 ? ng : 1));
    for (int64_t g = 0; g < ng; g++) {
        int64_t kv[2];
        oaggval v;
        oracle_agg_group(sub, g, NULL, kv, NULL, &v);
        gp[g] = (int32_t)kv[0]; gs[g] = (int32_t)kv[1];
        gsum[g] = v.h.upper == -1 ? -(float)(UINT64_MAX - v.h.lower) - 1 : (float)v.h.lower + (float)v.h.upper * (float)UINT64_MAX;   /* tryCastBigintToFloat32 */
    }
    oracle_agg_free(sub);
    /* partsupp x the groups on both keys, then the Filter */
    ocol gk[2] = {mkcol(OT_INT32, 0, gp), mkcol(OT_INT32, 0, gs)};
    ojoin *jg = oracle_join_build(gk, 2, NULL, ng);
    ocol pkeys[2] = {mkcol(OT_INT32, 0, ps_partkey), mkcol(OT_INT32, 0, ps_suppkey)};
    int64_t *j_ps = i64buf(nps), *j_g = i64buf(nps);
    const int64_t nj = oracle_join_probe_inner(jg, pkeys, 2, ps_sel, nps, j_ps, j_g, nps);
    oracle_join_free(jg);
    int32_t *good = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nj > 0 ? nj : 1));
    int64_t ngood = 0;
    for (int64_t i = 0; i < nj; i++) {
        volatile float thr = fraction * gsum[j_g[i]];
        if ((float)ps_availqty[j_ps[i]] > thr) good[ngood++] = ps_suppkey[j_ps[i]];
    }
    /* supplier x nation[name], SEMI on the qualifying keys */
    int64_t nsel[25];
    ocol nn = mkcode(T->n_name, T->nation_dict);
    oconst kn = kstr(nation);
    const int64_t cn = oracle_select(&nn, OP_EQ, &kn, NULL, 25, nsel);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, nsel, cn);
    int64_t *s_row = i64buf(T->n_supplier), *s_nat = i64buf(T->n_supplier);
    ocol sn = mkcol(OT_INT32, 0, T->s_nationkey);
    const int64_t ns = oracle_join_probe_inner(jn, &sn, 1, NULL, T->n_supplier, s_row, s_nat, T->n_supplier);
    oracle_join_free(jn);
    ocol gd = mkcol(OT_INT32, 0, good);
    ojoin *js = oracle_join_build(&gd, 1, NULL, ngood);
    uint8_t *found = (uint8_t *)malloc((size_t)(ns > 0 ? ns : 1));
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    oracle_join_probe_mark(js, &sk, 1, s_row, ns, found);
    oracle_join_free(js);
    int64_t nout = rc ? -1 : 0;
    for (int64_t i = 0; i < ns && rc == 0; i++)
        if (found[i]) { if (nout < max) out[nout] = T->s_suppkey[s_row[i]]; nout++; }
    /* (s_row ascends: the probe keeps supplier order; the keys ascend with it) */
    free(psel); free(ps_hit); free(ps_sel); free(s1); free(s2); free(gp); free(gs); free(gsum); free(j_ps); free(j_g); free(good); free(s_row); free(s_nat); free(found);
    return nout;
}

/* ------------------------------------------------------------------ Q21 (cases/tpch/query/q21.sql)
 * Limit 100 <- Order(numwait desc, s_name) <- Agg(s_name; count(*))
 *   <- ANTI Join(l3.l_orderkey = l1.l_orderkey AND l3.l_suppkey <> l1.l_suppkey) build Scan(lineitem l3, l_receiptdate > l_commitdate)
 *   <- SEMI Join(l2.l_orderkey = l1.l_orderkey AND l2.l_suppkey <> l1.l_suppkey) build Scan(lineitem l2)
 *   <- Join(o_orderkey = l_orderkey) build Scan(orders, o_orderstatus = 'F') <- Join(s_suppkey = l_suppkey) probe Scan(lineitem l1, receipt > commit)
 *      build Join(s_nationkey = n_nationkey)[supplier, nation(n_name = ..)].
 * A join's non-equi conjunct is evaluated over the key matches (the hash join's pairs, then the condition as a select over the joined
 * columns: join_scan.go's Next* with the extra conditions), EXISTS / NOT EXISTS keep the probe rows with / without a surviving match.
 * l2 / l3's key matches are the lines of l1's own order. Returns the groups (supplier key, count) in first-seen order, -1 on error. */
int64_t oracle_q21(const oracle_tpch *T, const uint8_t *o_orderstatus, const char *nation, oracle_q21_row *out, int64_t max) {
    /* supplier x nation[name] */
    int64_t nsel[25];
    ocol nn = mkcode(T->n_name, T->nation_dict);
    oconst kn = kstr(nation);
    const int64_t cn = oracle_select(&nn, OP_EQ, &kn, NULL, 25, nsel);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, nsel, cn);
    int64_t *s_row = i64buf(T->n_supplier), *s_nat = i64buf(T->n_supplier);
    ocol sn = mkcol(OT_INT32, 0, T->s_nationkey);
    const int64_t ns = oracle_join_probe_inner(jn, &sn, 1, NULL, T->n_supplier, s_row, s_nat, T->n_supplier);
    oracle_join_free(jn);
    /* l1 = lineitem[l_receiptdate > l_commitdate] x those suppliers */
    const int64_t n = T->n_lineitem;
    int64_t *late = i64buf(n);
    ocol lr = mkcol(OT_DATE, 0, T->l_receiptdate), lc = mkcol(OT_DATE, 0, T->l_commitdate);
    const int64_t nlate = oracle_select_cols(&lr, OP_GT, &lc, NULL, n, late);
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    ojoin *js = oracle_join_build(&sk, 1, s_row, ns);
    int64_t *l1 = i64buf(nlate), *l1s = i64buf(nlate);
    ocol lsup = mkcol(OT_INT32, 0, T->l_suppkey);
    const int64_t n1 = oracle_join_probe_inner(js, &lsup, 1, late, nlate, l1, l1s, nlate);
    oracle_join_free(js);
    /* x orders[o_orderstatus = 'F'] */
    int64_t *osel = i64buf(T->n_orders), no = 0;
    for (int64_t i = 0; i < T->n_orders; i++) if (o_orderstatus[i] == 'F') osel[no++] = i;      /* equalStrOp on the one-character status */
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    ojoin *jo = oracle_join_build(&ok, 1, osel, no);
    int64_t *kb = i64buf(n1);
    for (int64_t i = 0; i < n1; i++) kb[i] = T->l_orderkey[l1[i]];
    ocol kbc = mkcol(OT_INT64, 0, kb);
    int64_t *p2 = i64buf(n1), *o2 = i64buf(n1);
    const int64_t n2 = oracle_join_probe_inner(jo, &kbc, 1, NULL, n1, p2, o2, n1);
    oracle_join_free(jo);
    /* rows so far: lineitem row l1[p2[i]], supplier row l1s[p2[i]] */
    int64_t *cur = i64buf(n2), *cur_s = i64buf(n2);
    for (int64_t i = 0; i < n2; i++) { cur[i] = l1[p2[i]]; cur_s[i] = l1s[p2[i]]; }
    /* EXISTS l2 / NOT EXISTS l3: the join on l_orderkey, then `<>` on the suppliers over the pairs */
    ocol lk = mkcol(OT_INT64, 0, T->l_orderkey);
    int64_t ncur = n2;
    for (int pass = 0; pass < 2; pass++) {
        ojoin *jl = pass == 0 ? oracle_join_build(&lk, 1, NULL, n) : oracle_join_build(&lk, 1, late, nlate);
        int64_t *key = i64buf(ncur);
        for (int64_t i = 0; i < ncur; i++) key[i] = T->l_orderkey[cur[i]];
        ocol kc = mkcol(OT_INT64, 0, key);
        const int64_t cap = 8 * ncur + 64;                     /* at most seven lines per order */
        int64_t *pp = i64buf(cap), *bb = i64buf(cap);
        const int64_t np = oracle_join_probe_inner(jl, &kc, 1, NULL, ncur, pp, bb, cap);
        oracle_join_free(jl);
        uint8_t *hit = (uint8_t *)calloc((size_t)(ncur > 0 ? ncur : 1), 1);
        for (int64_t i = 0; i < np && np <= cap; i++)
            if (T->l_suppkey[bb[i]] != T->l_suppkey[cur[pp[i]]]) hit[pp[i]] = 1;      /* notEqualOp[int32] over the joined pair */
        int64_t m = 0;
        for (int64_t i = 0; i < ncur; i++)
            if ((pass == 0) == (hit[i] != 0)) { cur[m] = cur[i]; cur_s[m] = cur_s[i]; m++; }
        ncur = np <= cap ? m : 0;
        free(key); free(pp); free(bb); free(hit);
    }
    /* Agg(s_name; count(*)): s_name is the supplier's key in text, a VARCHAR key — grouped here by the key */
    ocol kp[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec ag[1] = {{OA_COUNT, -1}};
    oagg *t = oracle_agg_create(kp, 1, NULL, ag, 1);
    int32_t kv32[VS];
    int rc = 0;
    for (int64_t base = 0; base < ncur && rc == 0; base += VS) {
        int64_t cnt = ncur - base < VS ? ncur - base : VS;
        for (int64_t j = 0; j < cnt; j++) kv32[j] = T->s_suppkey[cur_s[base + j]];   /* (a build over a selection reports its ROW ids) */
        ocol keys[1] = {mkcol(OT_INT32, 0, kv32)};
        rc = oracle_agg_sink(t, keys, NULL, NULL, cnt);
    }
    const int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[1];
        oaggval v;
        oracle_agg_group(t, g, NULL, kv, NULL, &v);
        out[g].s_suppkey = (int32_t)kv[0];
        out[g].numwait = v.h;
    }
    oracle_agg_free(t);
    free(s_row); free(s_nat); free(late); free(l1); free(l1s); free(osel); free(kb); free(p2); free(o2); free(cur); free(cur_s);
    return ng;
}

/* ------------------------------------------------------------------ text */
/* extract(year from date): Date.Year (pkg/common/date.go) */
static int32_t year_of_days2(int32_t z) {
    z += 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    int32_t y = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    uint32_t mp = (5u * doy + 2u) / 153u;
    int32_t m = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    return y + (m <= 2);
}


/* ------------------------------------------------------------------ Q7 (cases/tpch/query/q7.sql)
 * Agg(supp_nation, cust_nation, l_year; sum(volume)) <- Filter((n1 = A and n2 = B) or (n1 = B and n2 = A))
 *   <- Join(c_nationkey = n2.n_nationkey) <- Join(s_nationkey = n1.n_nationkey) <- Join(c_custkey = o_custkey)
 *   <- Join(o_orderkey = l_orderkey) <- Join(s_suppkey = l_suppkey) probe Scan(lineitem, l_shipdate between), build Scan(supplier).
 * Every join is N:1 from lineitem's side, so the result does not depend on the order the optimizer picks; the pair
 * condition is evaluated by executeSelect's OR of two ANDs (expr_exec.go:342-530) on the joined chunk. */
int64_t oracle_q7(const oracle_tpch *T, const char *nation_a, const char *nation_b, int32_t date_ge, int32_t date_le, oracle_q7_row *out, int64_t max) {
    int64_t n = T->n_lineitem;
    int64_t *s1 = i64buf(n), *s2 = i64buf(n);
    ocol ls = mkcol(OT_DATE, 0, T->l_shipdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_le);
    int64_t c = oracle_select(&ls, OP_GE, &k1, NULL, n, s1);          /* BETWEEN = >= AND <= (bindBetweenExpr) */
    c = oracle_select(&ls, OP_LE, &k2, s1, c, s2);
    /* x supplier */
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    ojoin *js = oracle_join_build(&sk, 1, NULL, T->n_supplier);
    int64_t *l1 = i64buf(c), *sr = i64buf(c);
    ocol lsup = mkcol(OT_INT32, 0, T->l_suppkey);
    int64_t n1 = oracle_join_probe_inner(js, &lsup, 1, s2, c, l1, sr, c);
    oracle_join_free(js);
    /* x orders */
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    ojoin *jo = oracle_join_build(&ok, 1, NULL, T->n_orders);
    int64_t *kbuf = i64buf(n1);
    for (int64_t i = 0; i < n1; i++) kbuf[i] = T->l_orderkey[l1[i]];
    ocol pk = mkcol(OT_INT64, 0, kbuf);
    int64_t *p2 = i64buf(n1), *orow = i64buf(n1);
    int64_t n2 = oracle_join_probe_inner(jo, &pk, 1, NULL, n1, p2, orow, n1);
    oracle_join_free(jo);
    /* x customer */
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    ojoin *jc = oracle_join_build(&ck, 1, NULL, T->n_customer);
    int32_t *cbuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    for (int64_t i = 0; i < n2; i++) cbuf[i] = T->o_custkey[orow[i]];
    ocol pc = mkcol(OT_INT32, 0, cbuf);
    int64_t *p3 = i64buf(n2), *crow = i64buf(n2);
    int64_t n3 = oracle_join_probe_inner(jc, &pc, 1, NULL, n2, p3, crow, n2);
    oracle_join_free(jc);
    /* x nation n1 (supplier's), x nation n2 (customer's): n_nationkey = row of the fixed NATION table; the joins keep every row */
    ocol kproto[3] = {mkcode(NULL, T->nation_dict), mkcode(NULL, T->nation_dict), mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 3, aproto, aggs, 1);
    static odec v[VS];
    int64_t ext[VS], disc[VS], t1[VS], t2[VS], t3[VS], t4[VS], keep[VS];
    uint8_t sn[VS], cn[VS];
    int32_t yr[VS];
    oconst ka = kstr(nation_a), kb = kstr(nation_b);
    int rc = 0;
    for (int64_t base = 0; base < n3 && rc == 0; base += VS) {
        int64_t cnt = n3 - base < VS ? n3 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t q3 = p3[base + j], q2 = p2[q3], l = l1[q2];
            ext[j] = T->l_extendedprice[l];
            disc[j] = T->l_discount[l];
            sn[j] = T->n_name[T->s_nationkey[sr[q2]]];
            cn[j] = T->n_name[T->c_nationkey[crow[base + j]]];
            yr[j] = year_of_days2(T->l_shipdate[l]);
        }
        ocol snc = mkcode(sn, T->nation_dict), cnc = mkcode(cn, T->nation_dict);
        /* (n1 = A and n2 = B) or (n1 = B and n2 = A): execSelectOr unites the two branches' selections in row order */
        int64_t a1 = oracle_select(&snc, OP_EQ, &ka, NULL, cnt, t1);
        a1 = oracle_select(&cnc, OP_EQ, &kb, t1, a1, t2);
        int64_t b1 = oracle_select(&snc, OP_EQ, &kb, NULL, cnt, t3);
        b1 = oracle_select(&cnc, OP_EQ, &ka, t3, b1, t4);
        int64_t m = 0, ia = 0, ib = 0;
        while (ia < a1 || ib < b1) {
            if (ib >= b1 || (ia < a1 && t2[ia] < t4[ib])) keep[m++] = t2[ia++];
            else if (ia >= a1 || t4[ib] < t2[ia]) keep[m++] = t4[ib++];
            else { keep[m++] = t2[ia++]; ib++; }
        }
        /* the filter's selection narrows the chunk (SliceIndice): the aggregate sees the surviving rows, positionally */
        for (int64_t i = 0; i < m; i++) {
            const int64_t j = keep[i];   /* keep[i] >= i: in place */
            ext[i] = ext[j]; disc[i] = disc[j]; sn[i] = sn[j]; cn[i] = cn[j]; yr[i] = yr[j];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, m, v);
        ocol keys[3] = {mkcode(sn, T->nation_dict), mkcode(cn, T->nation_dict), mkcol(OT_INT32, 0, yr)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0 && m > 0) rc = oracle_agg_sink(t, keys, args, NULL, m);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[3];
        oaggval val;
        oracle_agg_group(t, g, NULL, kv, NULL, &val);
        out[g].supp_nation = (int32_t)kv[0];
        out[g].cust_nation = (int32_t)kv[1];
        out[g].l_year = (int32_t)kv[2];
        out[g].revenue = val.d;
    }
    oracle_agg_free(t);
    free(s1); free(s2); free(l1); free(sr); free(kbuf); free(p2); free(orow); free(cbuf); free(p3); free(crow);
    return ng;
}

/* ------------------------------------------------------------------ Q8 (cases/tpch/query/q8.sql)
 * Agg(o_year; sum(case when nation = X then volume else 0 end), sum(volume)) over part[p_type] x lineitem x supplier x orders[date range]
 * x customer x nation n1 x region[r_name] and nation n2 (the supplier's); the select list divides the two sums: DECIMAL `/` =
 * govalues Quo, typed as its first argument (BindDecimalDivide, function_scalar.go:507-514), printed at that scale. */
int64_t oracle_q8(const oracle_tpch *T, const char *nation, const char *region, const char *ptype, int32_t date_ge, int32_t date_le,
                  oracle_q8_row *out, int64_t max) {
    /* part[p_type = ..] */
    int64_t *psel = i64buf(T->n_part);
    ocol pt = mkcode(T->p_type, T->type_dict);
    oconst kt = kstr(ptype);
    int64_t np = oracle_select(&pt, OP_EQ, &kt, NULL, T->n_part, psel);
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, psel, np);
    int64_t cap = T->n_lineitem;
    int64_t *l1 = i64buf(cap), *pr = i64buf(cap);
    ocol lp = mkcol(OT_INT32, 0, T->l_partkey);
    int64_t n1 = oracle_join_probe_inner(jp, &lp, 1, NULL, T->n_lineitem, l1, pr, cap);
    oracle_join_free(jp);
    /* x orders[o_orderdate between] */
    int64_t *o1 = i64buf(T->n_orders), *o2 = i64buf(T->n_orders);
    ocol od = mkcol(OT_DATE, 0, T->o_orderdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_le);
    int64_t no = oracle_select(&od, OP_GE, &k1, NULL, T->n_orders, o1);
    no = oracle_select(&od, OP_LE, &k2, o1, no, o2);
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    ojoin *jo = oracle_join_build(&ok, 1, o2, no);
    int64_t *kbuf = i64buf(n1);
    for (int64_t i = 0; i < n1; i++) kbuf[i] = T->l_orderkey[l1[i]];
    ocol pko = mkcol(OT_INT64, 0, kbuf);
    int64_t *p2 = i64buf(n1), *orow = i64buf(n1);
    int64_t n2 = oracle_join_probe_inner(jo, &pko, 1, NULL, n1, p2, orow, n1);
    oracle_join_free(jo);
    /* nations of the region, customers of those nations */
    int64_t rsel[5], nrow[25], nreg[25];
    ocol rn = mkcode(T->r_name, T->region_dict);
    oconst kr = kstr(region);
    int64_t nr = oracle_select(&rn, OP_EQ, &kr, NULL, 5, rsel);
    ocol rk = mkcol(OT_INT32, 0, T->r_regionkey);
    ojoin *jr = oracle_join_build(&rk, 1, rsel, nr);
    ocol nrk = mkcol(OT_INT32, 0, T->n_regionkey);
    int64_t nn = oracle_join_probe_inner(jr, &nrk, 1, NULL, 25, nrow, nreg, 25);
    oracle_join_free(jr);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, nrow, nn);
    int64_t *c_row = i64buf(T->n_customer), *c_nat = i64buf(T->n_customer);
    ocol cn = mkcol(OT_INT32, 0, T->c_nationkey);
    int64_t nc = oracle_join_probe_inner(jn, &cn, 1, NULL, T->n_customer, c_row, c_nat, T->n_customer);
    oracle_join_free(jn);
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    ojoin *jc = oracle_join_build(&ck, 1, c_row, nc);
    int32_t *cbuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    for (int64_t i = 0; i < n2; i++) cbuf[i] = T->o_custkey[orow[i]];
    ocol pc = mkcol(OT_INT32, 0, cbuf);
    int64_t *p3 = i64buf(n2), *crow = i64buf(n2);
    int64_t n3 = oracle_join_probe_inner(jc, &pc, 1, NULL, n2, p3, crow, n2);
    oracle_join_free(jc);
    /* x supplier x nation n2: both keep every row (foreign keys into whole tables); s_suppkey = row + 1 is NOT assumed: a join */
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    ojoin *js = oracle_join_build(&sk, 1, NULL, T->n_supplier);
    int32_t *sbuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n3 > 0 ? n3 : 1));
    for (int64_t i = 0; i < n3; i++) sbuf[i] = T->l_suppkey[l1[p2[p3[i]]]];
    ocol ps = mkcol(OT_INT32, 0, sbuf);
    int64_t *p4 = i64buf(n3), *srow = i64buf(n3);
    int64_t n4 = oracle_join_probe_inner(js, &ps, 1, NULL, n3, p4, srow, n3);
    oracle_join_free(js);
    ocol kproto[1] = {mkcol(OT_INT32, 0, NULL)};
    ocol aproto[2] = {mkcol(OT_ODEC, 0, NULL), mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[2] = {{OA_SUM, 0}, {OA_SUM, 1}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 2);
    static odec vc[VS], va[VS];
    static uint8_t vnull[VS];
    int64_t ext[VS], disc[VS];
    uint8_t nat[VS];
    int32_t yr[VS];
    const orpn zero[1] = {{OX_CONST_INT, 0, 0, 0}};
    oconst kn = kstr(nation);
    int rc = 0;
    for (int64_t base = 0; base < n4 && rc == 0; base += VS) {
        int64_t cnt = n4 - base < VS ? n4 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t q4 = p4[base + j], q3 = p3[q4], q2 = p2[q3], l = l1[q2];
            ext[j] = T->l_extendedprice[l];
            disc[j] = T->l_discount[l];
            nat[j] = T->n_name[T->s_nationkey[srow[base + j]]];
            yr[j] = year_of_days2(T->o_orderdate[orow[q3]]);
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        ocol wc = mkcode(nat, T->nation_dict);
        rc = oracle_case_decimal(cols, &wc, OP_EQ, &kn, DISC_PRICE, 5, zero, 1, cnt, vc, vnull);   /* case when nation = X then volume else 0 end */
        if (rc == 0) rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, cnt, va);
        ocol keys[1] = {mkcol(OT_INT32, 0, yr)};
        ocol args[2] = {mkcol(OT_ODEC, 0, vc), mkcol(OT_ODEC, 0, va)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[1];
        oaggval v[2];
        oracle_agg_group(t, g, NULL, kv, NULL, v);
        out[g].o_year = (int32_t)kv[0];
        out[g].nation_volume = v[0].d;
        out[g].volume = v[1].d;
        if (odec_quo(v[0].d, v[1].d, &out[g].mkt_share) != 0) ng = -1;
    }
    oracle_agg_free(t);
    free(psel); free(l1); free(pr); free(o1); free(o2); free(kbuf); free(p2); free(orow); free(c_row); free(c_nat); free(cbuf);
    free(p3); free(crow); free(sbuf); free(p4); free(srow);
    (void)nreg;
    return ng;
}

/* ------------------------------------------------------------------ Q11 (cases/tpch/query/q11.sql)
 * Agg(ps_partkey; sum(ps_supplycost * ps_availqty)) over partsupp x supplier x nation[n_name], HAVING sum > (the same sum over
 * all those rows) * 0.0001000000. The literal is FLOAT (bindAConst, builder_binder.go:264-273), so the subquery's select list is
 * FLOAT arithmetic — the DECIMAL sum cast decimal -> float64 -> float32, multiplied in float32 (as Q14's) — and the HAVING compares
 * a DECIMAL column with a FLOAT one: both sides as float32 (MaxLType(DECIMAL, FLOAT) = FLOAT; greaterFloatOp). ps_availqty is
 * INTEGER: cast to DECIMAL(.., 0) for the product (scale 2). Returns the groups that pass, in first-seen order. */
int64_t oracle_q11(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int64_t *ps_supplycost,
                   const int32_t *ps_availqty, const char *nation, float fraction, oracle_q11_row *out, int64_t max) {
    int64_t nsel[25];
    ocol nn = mkcode(T->n_name, T->nation_dict);
    oconst kn = kstr(nation);
    int64_t cn = oracle_select(&nn, OP_EQ, &kn, NULL, 25, nsel);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, nsel, cn);
    int64_t *s_row = i64buf(T->n_supplier), *s_nat = i64buf(T->n_supplier);
    ocol sn = mkcol(OT_INT32, 0, T->s_nationkey);
    int64_t ns = oracle_join_probe_inner(jn, &sn, 1, NULL, T->n_supplier, s_row, s_nat, T->n_supplier);
    oracle_join_free(jn);
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    ojoin *js = oracle_join_build(&sk, 1, s_row, ns);
    int64_t *p_row = i64buf(n_ps), *p_sup = i64buf(n_ps);
    ocol pk = mkcol(OT_INT32, 0, ps_suppkey);
    int64_t np = oracle_join_probe_inner(js, &pk, 1, NULL, n_ps, p_row, p_sup, n_ps);
    oracle_join_free(js);
    static const int32_t one = 1;
    ocol kproto[1] = {mkcol(OT_INT32, 0, NULL)}, kproto1[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 1), *tt = oracle_agg_create(kproto1, 1, aproto, aggs, 1);
    static odec v[VS];
    int64_t cost[VS];
    int32_t qty[VS], part[VS];
    const orpn prod[3] = {{OX_COL, 0, 0, 0}, {OX_COL, 1, 0, 0}, {OX_MUL, 0, 0, 0}};
    int rc = 0;
    for (int64_t base = 0; base < np && rc == 0; base += VS) {
        int64_t cnt = np - base < VS ? np - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t r = p_row[base + j];
            cost[j] = ps_supplycost[r];
            qty[j] = ps_availqty[r];
            part[j] = ps_partkey[r];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, cost), mkcol(OT_INT32, 0, qty)};
        rc = oracle_eval_decimal(cols, prod, 3, NULL, cnt, v);
        ocol keys[1] = {mkcol(OT_INT32, 0, part)}, keys1[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
        if (rc == 0) rc = oracle_agg_sink(tt, keys1, args, NULL, cnt);
    }
    int64_t nout = rc ? -1 : 0;
    if (rc == 0 && oracle_agg_count(tt) == 1) {
        int64_t kv[1];
        oaggval total;
        oracle_agg_group(tt, 0, NULL, kv, NULL, &total);
        volatile float tf = (float)odec_float64(total.d);
        volatile float threshold = tf * fraction;
        int64_t ng = oracle_agg_count(t);
        for (int64_t g = 0; g < ng; g++) {
            oaggval val;
            oracle_agg_group(t, g, NULL, kv, NULL, &val);
            volatile float gv = (float)odec_float64(val.d);
            if (!(gv > threshold)) continue;
            if (nout < max) { out[nout].ps_partkey = (int32_t)kv[0]; out[nout].value = val.d; }
            nout++;
        }
    }
    oracle_agg_free(t); oracle_agg_free(tt);
    free(s_row); free(s_nat); free(p_row); free(p_sup);
    return nout;
}

typedef struct { char *buf; int64_t cap, len; } sbuf2;
static void put(sbuf2 *s, const char *t) {
    int64_t n = (int64_t)strlen(t);
    if (s->len + n < s->cap) memcpy(s->buf + s->len, t, (size_t)n);
    s->len += n;
}
static int64_t done(sbuf2 *s) { if (s->len < s->cap) s->buf[s->len] = 0; else if (s->cap > 0) s->buf[s->cap - 1] = 0; return s->len; }

/* ORDER BY through the reference's key encoding (oracle_sort_rows: LocalSort's byte-comparable keys) */
static void order_by_code(const int32_t *codes, int64_t n, int64_t *order) {
    uint8_t *c8 = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) c8[i] = (uint8_t)codes[i];
    ocol k = mkcol(OT_CODE8, 0, c8);
    int32_t desc = 0;
    oracle_sort_rows(&k, &desc, 1, NULL, n, order, NULL, NULL);
    free(c8);
}

int64_t oracle_q4_text(oracle_q4_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) codes[i] = rows[i].code;
    order_by_code(codes, n, ord);
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        put(&s, dict[rows[ord[i]].code]); put(&s, "\t");
        oracle_format_hugeint(rows[ord[i]].count, t); put(&s, t); put(&s, "\n");
    }
    free(codes); free(ord);
    return done(&s);
}

int64_t oracle_q5_text(oracle_q5_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    int64_t *un = i64buf(n), *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) { __int128 u = 0; odec_to_unscaled(rows[i].revenue, 4, &u); un[i] = (int64_t)u; }
    ocol k = mkcol(OT_DECIMAL, 4, un);
    int32_t desc = 1;
    oracle_sort_rows(&k, &desc, 1, NULL, n, ord, NULL, NULL);      /* ORDER BY revenue DESC */
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        put(&s, dict[rows[ord[i]].nation]); put(&s, "\t");
        oracle_format_decimal(rows[ord[i]].revenue, 4, t); put(&s, t); put(&s, "\n");
    }
    free(un); free(ord);
    return done(&s);
}

int64_t oracle_q12_text(oracle_q12_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\t\n");
    int32_t *codes = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) codes[i] = rows[i].mode;
    order_by_code(codes, n, ord);
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        put(&s, dict[rows[ord[i]].mode]); put(&s, "\t");
        oracle_format_hugeint(rows[ord[i]].high, t); put(&s, t); put(&s, "\t");
        oracle_format_hugeint(rows[ord[i]].low, t); put(&s, t); put(&s, "\n");
    }
    free(codes); free(ord);
    return done(&s);
}

int64_t oracle_q14_text(float promo_revenue, int is_null, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\n");
    char t[64];
    if (is_null) put(&s, "NULL");
    else { oracle_format_double((double)promo_revenue, t); put(&s, t); }   /* Value.String FLOAT: %v of float64(float32) */
    put(&s, "\n");
    return done(&s);
}

int64_t oracle_q19_text(const odec *revenue, int is_null, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\n");
    char t[64];
    if (is_null) put(&s, "NULL");
    else { oracle_format_decimal(*revenue, 4, t); put(&s, t); }
    put(&s, "\n");
    return done(&s);
}

int64_t oracle_q18_text(oracle_q18_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\t\t\t\t\n");
    int64_t *tp = i64buf(n), *ord = i64buf(n);
    int32_t *od = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) { tp[i] = rows[i].o_totalprice; od[i] = rows[i].o_orderdate; }
    ocol k[2] = {mkcol(OT_DECIMAL, 2, tp), mkcol(OT_DATE, 0, od)};
    int32_t desc[2] = {1, 0};
    oracle_sort_rows(k, desc, 2, NULL, n, ord, NULL, NULL);          /* ORDER BY o_totalprice DESC, o_orderdate */
    char t[64];
    for (int64_t i = 0; i < n && i < limit; i++) {
        const oracle_q18_row *r = &rows[ord[i]];
        snprintf(t, sizeof t, "Customer#%09d\t%d\t%lld\t", r->c_custkey, r->c_custkey, (long long)r->o_orderkey); put(&s, t);
        oracle_format_date(r->o_orderdate, t); put(&s, t); put(&s, "\t");
        odec d;
        odec_new(r->o_totalprice, 2, &d);
        oracle_format_decimal(d, 2, t); put(&s, t); put(&s, "\t");
        oracle_format_hugeint(r->sum_qty, t); put(&s, t); put(&s, "\n");
    }
    free(tp); free(ord); free(od);
    return done(&s);
}

static int q7_cmp_dict(const char *const *dict, int32_t a, int32_t b) { return strcmp(dict[a], dict[b]); }

int64_t oracle_q7_text(oracle_q7_row *rows, int64_t n, const char *const *nation_names, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\t\t\n");
    /* ORDER BY supp_nation, cust_nation, l_year (VARCHAR keys in byte order): a handful of rows, insertion sort */
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) {
        int64_t j = i;
        while (j > 0) {
            const oracle_q7_row *a = &rows[ord[j - 1]], *b = &rows[i];
            int c = q7_cmp_dict(nation_names, a->supp_nation, b->supp_nation);
            if (c == 0) c = q7_cmp_dict(nation_names, a->cust_nation, b->cust_nation);
            if (c == 0) c = (a->l_year > b->l_year) - (a->l_year < b->l_year);
            if (c <= 0) break;
            ord[j] = ord[j - 1];
            j--;
        }
        ord[j] = i;
    }
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        const oracle_q7_row *r = &rows[ord[i]];
        put(&s, nation_names[r->supp_nation]); put(&s, "\t");
        put(&s, nation_names[r->cust_nation]); put(&s, "\t");
        sprintf(t, "%d", r->l_year); put(&s, t); put(&s, "\t");
        oracle_format_decimal(r->revenue, 4, t); put(&s, t); put(&s, "\n");
    }
    free(ord);
    return done(&s);
}

int64_t oracle_q8_text(oracle_q8_row *rows, int64_t n, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) {       /* ORDER BY o_year */
        int64_t j = i;
        while (j > 0 && rows[ord[j - 1]].o_year > rows[i].o_year) { ord[j] = ord[j - 1]; j--; }
        ord[j] = i;
    }
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        const oracle_q8_row *r = &rows[ord[i]];
        sprintf(t, "%d", r->o_year); put(&s, t); put(&s, "\t");
        oracle_format_decimal(r->mkt_share, 4, t); put(&s, t); put(&s, "\n");   /* the quotient's column type is its first argument's: DECIMAL(38,4) */
    }
    free(ord);
    return done(&s);
}

int64_t oracle_q11_text(oracle_q11_row *rows, int64_t n, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    int64_t *un = i64buf(n), *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) { __int128 u = 0; odec_to_unscaled(rows[i].value, 2, &u); un[i] = (int64_t)u; }
    ocol k = mkcol(OT_DECIMAL, 2, un);
    int32_t desc = 1;
    oracle_sort_rows(&k, &desc, 1, NULL, n, ord, NULL, NULL);      /* ORDER BY value DESC */
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        sprintf(t, "%d", rows[ord[i]].ps_partkey); put(&s, t); put(&s, "\t");
        oracle_format_decimal(rows[ord[i]].value, 2, t); put(&s, t); put(&s, "\n");
    }
    free(un); free(ord);
    return done(&s);
}

int64_t oracle_q17_text(float avg_yearly, int is_null, char *buf, int64_t cap) {
    return oracle_q14_text(avg_yearly, is_null, buf, cap);   /* one FLOAT column: the same rendering */
}

/* s_name = 'Supplier#' + nine digits of the key (TPC-H 4.2.3); address and phone of supplier row i as the generator made them */
int64_t oracle_q15_text(const oracle_q15_row *rows, int64_t n, const int32_t *s_suppkey, int64_t n_supplier, const int32_t *addr_off, const char *addr_bytes,
                        const char *phone_bytes, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\t\t\t\n");
    char t[96];
    for (int64_t i = 0; i < n; i++) {   /* (already in supplier order = ORDER BY s_suppkey: the keys ascend) */
        int64_t r = -1;
        for (int64_t j = 0; j < n_supplier; j++) if (s_suppkey[j] == rows[i].s_suppkey) { r = j; break; }
        if (r < 0) continue;
        sprintf(t, "%d\tSupplier#%09d\t", rows[i].s_suppkey, rows[i].s_suppkey); put(&s, t);
        int32_t len = addr_off[r + 1] - addr_off[r];
        memcpy(t, addr_bytes + addr_off[r], (size_t)len); t[len] = 0; put(&s, t); put(&s, "\t");
        memcpy(t, phone_bytes + 15 * r, 15); t[15] = 0; put(&s, t); put(&s, "\t");
        oracle_format_decimal(rows[i].total_revenue, 4, t); put(&s, t); put(&s, "\n");
    }
    return done(&s);
}

int64_t oracle_q22_text(oracle_q22_row *rows, int64_t n, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\t\n");
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) {   /* ORDER BY cntrycode: byte order of the strings */
        int64_t j = i;
        while (j > 0 && strcmp(rows[ord[j - 1]].cntrycode, rows[i].cntrycode) > 0) { ord[j] = ord[j - 1]; j--; }
        ord[j] = i;
    }
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        const oracle_q22_row *r = &rows[ord[i]];
        put(&s, r->cntrycode); put(&s, "\t");
        oracle_format_hugeint(r->numcust, t); put(&s, t); put(&s, "\t");
        oracle_format_decimal(r->totacctbal, 2, t); put(&s, t); put(&s, "\n");
    }
    free(ord);
    return done(&s);
}

int64_t oracle_q20_text(const int32_t *keys, int64_t n, const int32_t *s_suppkey, int64_t n_supplier, const int32_t *addr_off, const char *addr_bytes, char *buf,
                        int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    char t[96];
    for (int64_t i = 0; i < n; i++) {   /* ORDER BY s_name = key order (zero-padded), which is the order the keys come in */
        int64_t r = -1;
        for (int64_t j = 0; j < n_supplier; j++) if (s_suppkey[j] == keys[i]) { r = j; break; }
        if (r < 0) continue;
        sprintf(t, "Supplier#%09d\t", keys[i]); put(&s, t);
        int32_t len = addr_off[r + 1] - addr_off[r];
        memcpy(t, addr_bytes + addr_off[r], (size_t)len); t[len] = 0; put(&s, t); put(&s, "\n");
    }
    return done(&s);
}

int64_t oracle_q21_text(oracle_q21_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap) {
    sbuf2 s = {buf, cap, 0};
    put(&s, "#\t\n");
    int64_t *ord = i64buf(n);
    for (int64_t i = 0; i < n; i++) {   /* ORDER BY numwait DESC, s_name (= the key: zero-padded) */
        int64_t j = i;
        while (j > 0) {
            const oracle_q21_row *a = &rows[ord[j - 1]], *b = &rows[i];
            const int after = a->numwait.lower < b->numwait.lower || (a->numwait.lower == b->numwait.lower && a->s_suppkey > b->s_suppkey);
            if (!after) break;
            ord[j] = ord[j - 1]; j--;
        }
        ord[j] = i;
    }
    char t[64];
    for (int64_t i = 0; i < n && i < limit; i++) {
        sprintf(t, "Supplier#%09d\t", rows[ord[i]].s_suppkey); put(&s, t);
        oracle_format_hugeint(rows[ord[i]].numwait, t); put(&s, t); put(&s, "\n");
    }
    free(ord);
    return done(&s);
}
