/* ORACLE — TEST INFRASTRUCTURE ONLY. See oracle.h.
 * Query drivers: the physical pipelines of the reference for TPC-H Q1/Q6/Q3/Q9, pulled one
 * 2048-row chunk at a time like execOps' loop (pkg/compute/executor.go:151-188), built from
 * the operator restatements in oracle.c, plus the result-text formatting. */
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

/* ------------------------------------------------------------------ Q1 */
/* Plan: Order <- Project <- Agg <- Scan(lineitem, l_shipdate <= const)  (SURVEY §3.1/3.2;
 * scanExecutor.Execute executor_scan.go:144-241, aggExecutor.Execute executor_aggr.go:106-142) */
int32_t oracle_q1(const oracle_lineitem *L, int32_t shipdate_le, oracle_q1_row *out,
                  int32_t max_groups) {
    ocol kproto[2] = {mkcode(NULL, L->returnflag_dict), mkcode(NULL, L->linestatus_dict)};
    ocol aproto[5] = {mkcol(OT_INT32, 0, NULL), mkcol(OT_ODEC, 0, NULL), mkcol(OT_ODEC, 0, NULL),
                      mkcol(OT_ODEC, 0, NULL), mkcol(OT_ODEC, 0, NULL)};
    /* sum(qty), sum(ext), sum(ext*(1-disc)), sum(ext*(1-disc)*(1+tax)), avg(qty), avg(ext),
     * avg(disc), count(*) */
    oaggspec aggs[8] = {{OA_SUM, 0}, {OA_SUM, 1}, {OA_SUM, 2}, {OA_SUM, 3},
                        {OA_AVG, 0}, {OA_AVG, 1}, {OA_AVG, 4}, {OA_COUNT, -1}};
    oagg *t = oracle_agg_create(kproto, 2, aproto, aggs, 8);
    const orpn p_ext[] = {{OX_COL, 0, 0, 0}};
    const orpn p_disc[] = {{OX_COL, 1, 0, 0}};
    const orpn p_dp[] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0},
                         {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}};
    const orpn p_ch[] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0},
                         {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0},
                         {OX_COL, 2, 0, 0}, {OX_ADD, 0, 0, 0}, {OX_MUL, 0, 0, 0}};
    oconst k;
    memset(&k, 0, sizeof k);
    k.type = OT_DATE;
    k.i = shipdate_le;

    static odec v_ext[VS], v_dp[VS], v_ch[VS], v_disc[VS];
    int64_t sel[VS], rid[VS];
    uint8_t rf[VS], ls[VS];
    int32_t qty[VS];
    int rc = 0;
    for (int64_t base = 0; base < L->n && rc == 0; base += VS) {
        int64_t cnt = L->n - base < VS ? L->n - base : VS;
        ocol sd = mkcol(OT_DATE, 0, L->l_shipdate + base);
        int64_t m = oracle_select(&sd, OP_LE, &k, NULL, cnt, sel); /* pushed-down filter */
        if (m == 0) continue;
        ocol cols[3] = {mkcol(OT_DECIMAL, 2, L->l_extendedprice + base),
                        mkcol(OT_DECIMAL, 2, L->l_discount + base),
                        mkcol(OT_DECIMAL, 2, L->l_tax + base)};
        rc |= oracle_eval_decimal(cols, p_ext, 1, sel, m, v_ext);
        rc |= oracle_eval_decimal(cols, p_disc, 1, sel, m, v_disc);
        rc |= oracle_eval_decimal(cols, p_dp, 5, sel, m, v_dp);
        rc |= oracle_eval_decimal(cols, p_ch, 9, sel, m, v_ch);
        for (int64_t j = 0; j < m; j++) {
            int64_t r = base + sel[j];
            rf[j] = L->l_returnflag[r];
            ls[j] = L->l_linestatus[r];
            qty[j] = L->l_quantity[r];
            rid[j] = r;
        }
        ocol keys[2] = {mkcode(rf, L->returnflag_dict), mkcode(ls, L->linestatus_dict)};
        ocol args[5] = {mkcol(OT_INT32, 0, qty), mkcol(OT_ODEC, 0, v_ext), mkcol(OT_ODEC, 0, v_dp),
                        mkcol(OT_ODEC, 0, v_ch), mkcol(OT_ODEC, 0, v_disc)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, rid, m);
    }
    int32_t ng = rc ? -1 : (int32_t)oracle_agg_count(t);
    for (int32_t g = 0; g < ng && g < max_groups; g++) {
        int64_t kv[2];
        oaggval v[8];
        if (oracle_agg_group(t, g, NULL, kv, NULL, v) != 0) { ng = -1; break; }
        oracle_q1_row *o = &out[g];
        memset(o, 0, sizeof *o);
        o->returnflag = (uint8_t)kv[0];
        o->linestatus = (uint8_t)kv[1];
        o->sum_qty = v[0].h;
        o->sum_base_price = v[1].d;
        o->sum_disc_price = v[2].d;
        o->sum_charge = v[3].d;
        o->avg_qty = v[4].f;
        o->avg_price = v[5].d;
        o->avg_disc = v[6].d;
        o->count_order = v[7].h.lower;
    }
    oracle_agg_free(t);
    return ng;
}

/* ------------------------------------------------------------------ Q6 */
/* Plan: Agg(no group by -> constant key 1, executor_aggr.go:37-48) <- Scan(lineitem, 5 conjuncts).
 * execSelectAnd narrows the selection conjunct by conjunct (expr_exec.go:444-486). */
int32_t oracle_q6(const oracle_lineitem *L, int32_t date_ge, int32_t date_lt, float disc_lo,
                  float disc_hi, int32_t qty_lt, odec *revenue) {
    static const int32_t one = 1;
    ocol kproto[1] = {mkcol(OT_CONST32, 0, &one)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 1, aproto, aggs, 1);
    const orpn prog[] = {{OX_COL, 0, 0, 0}, {OX_COL, 1, 0, 0}, {OX_MUL, 0, 0, 0}};
    oconst k1, k2, k3, k4, k5;
    memset(&k1, 0, sizeof k1);
    k1.type = OT_DATE; k1.i = date_ge;
    k2 = k1; k2.i = date_lt;
    memset(&k3, 0, sizeof k3);
    k3.type = OT_FLOAT; k3.f = disc_lo;
    k4 = k3; k4.f = disc_hi;
    memset(&k5, 0, sizeof k5);
    k5.type = OT_INT32; k5.i = qty_lt;
    static odec v[VS];
    int64_t s1[VS], s2[VS];
    int rc = 0;
    for (int64_t base = 0; base < L->n && rc == 0; base += VS) {
        int64_t cnt = L->n - base < VS ? L->n - base : VS;
        ocol sd = mkcol(OT_DATE, 0, L->l_shipdate + base);
        ocol dc = mkcol(OT_DECIMAL, 2, L->l_discount + base);
        ocol qt = mkcol(OT_INT32, 0, L->l_quantity + base);
        int64_t m = oracle_select(&sd, OP_GE, &k1, NULL, cnt, s1);
        if (m) m = oracle_select(&sd, OP_LT, &k2, s1, m, s2);
        if (m) m = oracle_select(&dc, OP_GE, &k3, s2, m, s1);
        if (m) m = oracle_select(&dc, OP_LE, &k4, s1, m, s2);
        if (m) m = oracle_select(&qt, OP_LT, &k5, s2, m, s1);
        if (m == 0) continue;
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, L->l_extendedprice + base), dc};
        rc = oracle_eval_decimal(cols, prog, 3, s1, m, v);
        ocol keys[1] = {mkcol(OT_CONST32, 0, &one)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, m);
    }
    int32_t res;
    if (rc) res = -1;
    else if (oracle_agg_count(t) == 0) res = 1; /* no input row: the ungrouped aggregate is NULL */
    else {
        oaggval val;
        oracle_agg_group(t, 0, NULL, NULL, NULL, &val);
        if (val.kind == OV_NULL) res = 1;
        else { *revenue = val.d; res = 0; }
    }
    oracle_agg_free(t);
    return res;
}

/* ------------------------------------------------------------------ Q3 */
/* Plan: Agg <- HashJoin(l_orderkey = o_orderkey) <- [Scan(lineitem, l_shipdate > d),
 *       HashJoin(o_custkey = c_custkey) <- [Scan(orders, o_orderdate < d),
 *                                            Scan(customer, c_mktsegment = seg)]]
 * joinExecutor (executor_join.go:54-264): children[1] is built, children[0] probes. */
int64_t oracle_q3(const oracle_lineitem *L, const oracle_orders *O, const oracle_customer *C,
                  const char *segment, int32_t date, oracle_q3_row *out, int64_t max) {
    oconst ks, kd;
    memset(&ks, 0, sizeof ks);
    ks.type = OT_VARCHAR; ks.s = segment;
    memset(&kd, 0, sizeof kd);
    kd.type = OT_DATE; kd.i = date;

    /* customer filter + build */
    int64_t *csel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(C->n ? C->n : 1));
    ocol cseg = mkcode(C->c_mktsegment, C->mktsegment_dict);
    int64_t nc = oracle_select(&cseg, OP_EQ, &ks, NULL, C->n, csel);
    ocol ckey = mkcol(OT_INT32, 0, C->c_custkey);
    ojoin *jc = oracle_join_build(&ckey, 1, csel, nc);

    /* orders filter + probe */
    int64_t *osel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(O->n ? O->n : 1));
    ocol odate = mkcol(OT_DATE, 0, O->o_orderdate);
    int64_t no = oracle_select(&odate, OP_LT, &kd, NULL, O->n, osel);
    ocol ocust = mkcol(OT_INT32, 0, O->o_custkey);
    int64_t *j1_o = (int64_t *)malloc(sizeof(int64_t) * (size_t)(no ? no : 1));
    int64_t *j1_c = (int64_t *)malloc(sizeof(int64_t) * (size_t)(no ? no : 1));
    int64_t n1 = oracle_join_probe_inner(jc, &ocust, 1, osel, no, j1_o, j1_c, no); /* N:1 */
    oracle_join_free(jc);

    /* materialise the join output that the next join builds on */
    int64_t *j1_okey = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n1 ? n1 : 1));
    for (int64_t i = 0; i < n1; i++) j1_okey[i] = O->o_orderkey[j1_o[i]];
    ocol bkey = mkcol(OT_INT64, 0, j1_okey);
    ojoin *jo = oracle_join_build(&bkey, 1, NULL, n1);

    /* lineitem filter + probe */
    int64_t *lsel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L->n ? L->n : 1));
    ocol lship = mkcol(OT_DATE, 0, L->l_shipdate);
    int64_t nl = oracle_select(&lship, OP_GT, &kd, NULL, L->n, lsel);
    ocol lkey = mkcol(OT_INT64, 0, L->l_orderkey);
    int64_t *j2_l = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nl ? nl : 1));
    int64_t *j2_b = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nl ? nl : 1));
    int64_t n2 = oracle_join_probe_inner(jo, &lkey, 1, lsel, nl, j2_l, j2_b, nl);
    oracle_join_free(jo);

    /* aggregate: group by (l_orderkey, o_orderdate, o_shippriority), sum(ext*(1-disc)) */
    ocol kproto[3] = {mkcol(OT_INT64, 0, NULL), mkcol(OT_DATE, 0, NULL), mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 3, aproto, aggs, 1);
    const orpn prog[] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0},
                         {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}};
    static odec v[VS];
    int64_t k0[VS], ext[VS], disc[VS];
    int32_t k1[VS], k2[VS];
    int rc = 0;
    for (int64_t base = 0; base < n2 && rc == 0; base += VS) {
        int64_t cnt = n2 - base < VS ? n2 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t l = j2_l[base + j], o = j1_o[j2_b[base + j]];
            k0[j] = L->l_orderkey[l];
            k1[j] = O->o_orderdate[o];
            k2[j] = O->o_shippriority[o];
            ext[j] = L->l_extendedprice[l];
            disc[j] = L->l_discount[l];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, prog, 5, NULL, cnt, v);
        ocol keys[3] = {mkcol(OT_INT64, 0, k0), mkcol(OT_DATE, 0, k1), mkcol(OT_INT32, 0, k2)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[3];
        oaggval val;
        oracle_agg_group(t, g, NULL, kv, NULL, &val);
        out[g].l_orderkey = kv[0];
        out[g].o_orderdate = (int32_t)kv[1];
        out[g].o_shippriority = (int32_t)kv[2];
        out[g].revenue = val.d;
    }
    oracle_agg_free(t);
    free(csel); free(osel); free(j1_o); free(j1_c); free(j1_okey); free(lsel); free(j2_l); free(j2_b);
    return ng;
}

/* ------------------------------------------------------------------ Q9 */
static int32_t year_of_days(int32_t z) { /* extract(year from date): Date.Year */
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

/* Six-way join + group by (nation, o_year). Every join is N:1 from lineitem's side, so the
 * intermediate result is carried as parallel arrays of base-table row ids. The nation join
 * (s_nationkey = n_nationkey, 25 rows) is the identity on the key and is folded into the group
 * key: n_name <-> n_nationkey is one to one. */
int64_t oracle_q9(const oracle_lineitem *L, const oracle_orders *O, const oracle_part *P,
                  const oracle_partsupp *PS, const oracle_supplier *S, const char *like_pattern,
                  oracle_q9_row *out, int64_t max) {
    oconst kl;
    memset(&kl, 0, sizeof kl);
    kl.type = OT_VARCHAR; kl.s = like_pattern;
    int64_t *psel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P->n ? P->n : 1));
    ocol pname = mkcol(OT_VARCHAR, 0, P->p_name_off);
    pname.dict = (const char *const *)P->p_name_bytes;
    int64_t np = oracle_select(&pname, OP_LIKE, &kl, NULL, P->n, psel);
    ocol pkey = mkcol(OT_INT32, 0, P->p_partkey);
    ojoin *jp = oracle_join_build(&pkey, 1, psel, np);

    int64_t cap = L->n ? L->n : 1;
    int64_t *cur_l = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t *tmp_b = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    ocol lpart = mkcol(OT_INT32, 0, L->l_partkey);
    int64_t n = oracle_join_probe_inner(jp, &lpart, 1, NULL, L->n, cur_l, tmp_b, cap);
    oracle_join_free(jp);

    /* partsupp on (partkey, suppkey) */
    ocol pskeys[2] = {mkcol(OT_INT32, 0, PS->ps_partkey), mkcol(OT_INT32, 0, PS->ps_suppkey)};
    ojoin *jps = oracle_join_build(pskeys, 2, NULL, PS->n);
    ocol lkeys[2] = {mkcol(OT_INT32, 0, L->l_partkey), mkcol(OT_INT32, 0, L->l_suppkey)};
    int64_t *l2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int64_t *ps2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int64_t n2 = oracle_join_probe_inner(jps, lkeys, 2, cur_l, n, l2, ps2, n);
    oracle_join_free(jps);

    /* supplier */
    ocol skey = mkcol(OT_INT32, 0, S->s_suppkey);
    ojoin *js = oracle_join_build(&skey, 1, NULL, S->n);
    ocol lsupp = mkcol(OT_INT32, 0, L->l_suppkey);
    /* probe positions must map back to (l2, ps2): probe with an explicit positional key copy */
    int32_t *ksupp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 ? n2 : 1));
    for (int64_t i = 0; i < n2; i++) ksupp[i] = L->l_suppkey[l2[i]];
    ocol ksc = mkcol(OT_INT32, 0, ksupp);
    int64_t *pos3 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n2 ? n2 : 1));
    int64_t *s3 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n2 ? n2 : 1));
    int64_t n3 = oracle_join_probe_inner(js, &ksc, 1, NULL, n2, pos3, s3, n2);
    oracle_join_free(js);
    (void)lsupp;

    /* orders */
    ocol okey = mkcol(OT_INT64, 0, O->o_orderkey);
    ojoin *jo = oracle_join_build(&okey, 1, NULL, O->n);
    int64_t *kord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n3 ? n3 : 1));
    for (int64_t i = 0; i < n3; i++) kord[i] = L->l_orderkey[l2[pos3[i]]];
    ocol koc = mkcol(OT_INT64, 0, kord);
    int64_t *pos4 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n3 ? n3 : 1));
    int64_t *o4 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n3 ? n3 : 1));
    int64_t n4 = oracle_join_probe_inner(jo, &koc, 1, NULL, n3, pos4, o4, n3);
    oracle_join_free(jo);

    /* aggregate */
    ocol kproto[2] = {mkcol(OT_INT32, 0, NULL), mkcol(OT_INT32, 0, NULL)};
    ocol aproto[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kproto, 2, aproto, aggs, 1);
    /* l_extendedprice*(1-l_discount) - ps_supplycost*l_quantity */
    const orpn prog[] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0},
                         {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}, {OX_COL, 2, 0, 0},
                         {OX_COL, 3, 0, 0}, {OX_MUL, 0, 0, 0}, {OX_SUB, 0, 0, 0}};
    static odec v[VS];
    int64_t ext[VS], disc[VS], cost[VS];
    int32_t qty[VS], k0[VS], k1[VS];
    int rc = 0;
    for (int64_t base = 0; base < n4 && rc == 0; base += VS) {
        int64_t cnt = n4 - base < VS ? n4 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t p3 = pos4[base + j];      /* index into the supplier-join output */
            int64_t p2 = pos3[p3];            /* index into the partsupp-join output */
            int64_t l = l2[p2];
            ext[j] = L->l_extendedprice[l];
            disc[j] = L->l_discount[l];
            qty[j] = L->l_quantity[l];
            cost[j] = PS->ps_supplycost[ps2[p2]];
            k0[j] = S->s_nationkey[s3[p3]];
            k1[j] = year_of_days(O->o_orderdate[o4[base + j]]);
        }
        ocol cols[4] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc),
                        mkcol(OT_DECIMAL, 2, cost), mkcol(OT_INT32, 0, qty)};
        rc = oracle_eval_decimal(cols, prog, 9, NULL, cnt, v);
        ocol keys[2] = {mkcol(OT_INT32, 0, k0), mkcol(OT_INT32, 0, k1)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t kv[2];
        oaggval val;
        oracle_agg_group(t, g, NULL, kv, NULL, &val);
        out[g].nationkey = (int32_t)kv[0];
        out[g].o_year = (int32_t)kv[1];
        out[g].sum_profit = val.d;
    }
    oracle_agg_free(t);
    free(psel); free(cur_l); free(tmp_b); free(l2); free(ps2); free(ksupp); free(pos3); free(s3);
    free(kord); free(pos4); free(o4);
    (void)n2;
    return ng;
}

/* ------------------------------------------------------------------ text */

int oracle_format_decimal(odec d, int type_scale, char *buf) {
    /* Vector.GetValue DECIMAL: Int64(Typ.Scale) (vector.go:121-137); Value.String:
     * NewFromInt64(w, f, scale).String() (value.go:37-46); falls back to d.String() */
    int64_t w, f;
    if (!odec_int64(d, type_scale, &w, &f)) return odec_string(d, buf);
    odec r;
    if (odec_new_from_int64(w, f, type_scale, &r) != ODEC_OK) return odec_string(d, buf);
    return odec_string(r, buf);
}

int oracle_format_double(double v, char *buf) {
    /* Value.String DOUBLE: fmt "%v" of a float64 (value.go:52-55) = strconv 'g' with the
     * shortest round-tripping digits; the %e form is used when exp < -4 || exp >= 6 (the
     * threshold precision is 6 when the digits are the shortest) */
    if (v != v) return sprintf(buf, "NaN");
    if (v == 0) return sprintf(buf, (1 / v < 0) ? "-0" : "0");
    if (v > 1.7976931348623157e308) return sprintf(buf, "+Inf");
    if (v < -1.7976931348623157e308) return sprintf(buf, "-Inf");
    char tmp[64];
    int prec;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(tmp, sizeof tmp, "%.*e", prec - 1, v);
        if (strtod(tmp, NULL) == v) break;
    }
    /* tmp = [-]d.ddddde[+-]XX */
    char digits[32];
    int nd = 0;
    const char *p = tmp;
    int neg = 0;
    if (*p == '-') { neg = 1; p++; }
    for (; *p && *p != 'e'; p++)
        if (*p != '.') digits[nd++] = *p;
    int exp = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    char *o = buf;
    if (neg) *o++ = '-';
    if (exp < -4 || exp >= 6) {
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, digits + 1, (size_t)(nd - 1));
            o += nd - 1;
        }
        o += sprintf(o, "e%c%02d", exp < 0 ? '-' : '+', exp < 0 ? -exp : exp);
    } else if (exp < 0) {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -exp - 1; i++) *o++ = '0';
        memcpy(o, digits, (size_t)nd);
        o += nd;
    } else {
        for (int i = 0; i <= exp; i++) *o++ = i < nd ? digits[i] : '0';
        if (nd > exp + 1) {
            *o++ = '.';
            memcpy(o, digits + exp + 1, (size_t)(nd - exp - 1));
            o += nd - exp - 1;
        }
    }
    *o = 0;
    return (int)(o - buf);
}

int oracle_format_hugeint(ohuge h, char *buf) {
    /* Value.String HUGEINT: big.Int(upper)<<64 + lower (value.go:60-66) */
    __int128 v = ((__int128)h.upper << 64) + (__int128)(unsigned __int128)h.lower;
    char tmp[48];
    int n = 0, neg = v < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    do {
        tmp[n++] = (char)('0' + (int)(u % 10));
        u /= 10;
    } while (u);
    char *o = buf;
    if (neg) *o++ = '-';
    while (n) *o++ = tmp[--n];
    *o = 0;
    return (int)(o - buf);
}

int oracle_format_date(int32_t days, char *buf) {
    int32_t z = days + 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    int32_t y = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    uint32_t mp = (5u * doy + 2u) / 153u;
    int32_t d = (int32_t)(doy - (153u * mp + 2u) / 5u + 1u);
    int32_t m = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    y += (m <= 2);
    return sprintf(buf, "%04d-%02d-%02d", y, m, d); /* time.DateOnly */
}

typedef struct {
    char *buf;
    int64_t cap, len;
} sbuf;

static void sb_put(sbuf *s, const char *t) {
    int64_t n = (int64_t)strlen(t);
    if (s->len + n < s->cap) memcpy(s->buf + s->len, t, (size_t)n);
    s->len += n;
}

static void sb_header(sbuf *s, int ncols) { /* "#" + names joined by tab; names are empty */
    sb_put(s, "#");
    for (int i = 1; i < ncols; i++) sb_put(s, "\t");
    sb_put(s, "\n");
}

static int64_t sb_finish(sbuf *s) {
    if (s->len < s->cap) s->buf[s->len] = 0;
    return s->len;
}

static const char *const *g_rf, *const *g_ls;
static int q1_cmp(const void *a, const void *b) {
    const oracle_q1_row *x = (const oracle_q1_row *)a, *y = (const oracle_q1_row *)b;
    int c = strcmp(g_rf[x->returnflag], g_rf[y->returnflag]);
    if (c) return c;
    return strcmp(g_ls[x->linestatus], g_ls[y->linestatus]);
}

int64_t oracle_q1_text(const oracle_q1_row *rows, int32_t n, const char *const *rf_dict,
                       const char *const *ls_dict, char *buf, int64_t cap) {
    oracle_q1_row *r = (oracle_q1_row *)malloc(sizeof(oracle_q1_row) * (size_t)(n ? n : 1));
    memcpy(r, rows, sizeof(oracle_q1_row) * (size_t)n);
    g_rf = rf_dict;
    g_ls = ls_dict;
    qsort(r, (size_t)n, sizeof *r, q1_cmp); /* order by l_returnflag, l_linestatus */
    sbuf s = {buf, cap, 0};
    sb_header(&s, 10);
    char t[64];
    for (int32_t i = 0; i < n; i++) {
        sb_put(&s, rf_dict[r[i].returnflag]); sb_put(&s, "\t");
        sb_put(&s, ls_dict[r[i].linestatus]); sb_put(&s, "\t");
        oracle_format_hugeint(r[i].sum_qty, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(r[i].sum_base_price, 2, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(r[i].sum_disc_price, 4, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(r[i].sum_charge, 6, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_double(r[i].avg_qty, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(r[i].avg_price, 2, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(r[i].avg_disc, 2, t); sb_put(&s, t); sb_put(&s, "\t");
        ohuge c = {r[i].count_order, 0};
        oracle_format_hugeint(c, t); sb_put(&s, t); sb_put(&s, "\n");
    }
    free(r);
    return sb_finish(&s);
}

int64_t oracle_q6_text(const odec *revenue, int is_null, char *buf, int64_t cap) {
    sbuf s = {buf, cap, 0};
    sb_header(&s, 1);
    char t[64];
    if (is_null) sb_put(&s, "NULL");
    else { oracle_format_decimal(*revenue, 4, t); sb_put(&s, t); }
    sb_put(&s, "\n");
    return sb_finish(&s);
}

static int q3_cmp(const void *a, const void *b) {
    const oracle_q3_row *x = (const oracle_q3_row *)a, *y = (const oracle_q3_row *)b;
    int c = odec_cmp(y->revenue, x->revenue); /* revenue desc */
    if (c) return c;
    if (x->o_orderdate != y->o_orderdate) return x->o_orderdate < y->o_orderdate ? -1 : 1;
    return 0;
}

int64_t oracle_q3_text(oracle_q3_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap) {
    qsort(rows, (size_t)n, sizeof *rows, q3_cmp);
    sbuf s = {buf, cap, 0};
    sb_header(&s, 4);
    char t[64];
    for (int64_t i = 0; i < n && i < limit; i++) {
        sprintf(t, "%lld", (long long)rows[i].l_orderkey); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(rows[i].revenue, 4, t); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_date(rows[i].o_orderdate, t); sb_put(&s, t); sb_put(&s, "\t");
        sprintf(t, "%d", rows[i].o_shippriority); sb_put(&s, t); sb_put(&s, "\n");
    }
    return sb_finish(&s);
}

static const char *const *g_nat;
static int q9_cmp(const void *a, const void *b) {
    const oracle_q9_row *x = (const oracle_q9_row *)a, *y = (const oracle_q9_row *)b;
    int c = strcmp(g_nat[x->nationkey], g_nat[y->nationkey]);
    if (c) return c;
    return (x->o_year > y->o_year) ? -1 : (x->o_year < y->o_year); /* o_year desc */
}

int64_t oracle_q9_text(oracle_q9_row *rows, int64_t n, const char *const *nation_names,
                       char *buf, int64_t cap) {
    g_nat = nation_names;
    qsort(rows, (size_t)n, sizeof *rows, q9_cmp);
    sbuf s = {buf, cap, 0};
    sb_header(&s, 3);
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        sb_put(&s, nation_names[rows[i].nationkey]); sb_put(&s, "\t");
        sprintf(t, "%d", rows[i].o_year); sb_put(&s, t); sb_put(&s, "\t");
        oracle_format_decimal(rows[i].sum_profit, 4, t); sb_put(&s, t); sb_put(&s, "\n");
    }
    return sb_finish(&s);
}
