/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Round 4: the pipelines of the four remaining TPC-H queries with a reference golden — Q2, Q10, Q13, Q16
 * (cases/tpch/query/q{2,10,13,16}.sql), the ones that read the generator's COMMENT text. They pin what no earlier golden
 * reached: the LEFT OUTER join (NextLeftJoin, join_scan.go:67-88) with count() over its NULL-extended side and CountOp's
 * NULL-for-zero finalize (function_aggr.go:950-962) as a GROUP KEY of the aggregate above (q13.txt's first row is
 * `NULL\t50005`), COUNT(DISTINCT) through the distinct side table (SinkDistinctGrouping / DistinctGrouping,
 * aggregate_exec.go:76-105, 201-304), NOT IN as an ANTI join (builder_plan.go:497-505), a correlated min() decorrelated into
 * an aggregate by its key and joined back on (key, value) — DECIMAL `=` exists only as a join condition —, LIKE with several
 * '%' (wildcardMatch, function_operator_boolean.go:336-377) over VARCHAR text, and VARCHAR columns in the select list.
 * Built from the same blocks as the other pipelines (oracle_select / oracle_agg_* / oracle_join_*), 2048 rows at a time.
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VS ORACLE_VECTOR_SIZE

static ocol mkcol(int32_t type, int32_t scale, const void *data) {
    ocol c;
    memset(&c, 0, sizeof c);
    c.type = type; c.scale = scale; c.data = data;
    return c;
}
static ocol mkcode(const uint8_t *data, const char *const *dict) {
    ocol c = mkcol(OT_CODE8, 0, data);
    c.dict = dict;
    return c;
}
static ocol mkstr(const int32_t *off, const char *bytes) {
    ocol c = mkcol(OT_VARCHAR, 0, off);
    c.dict = (const char *const *)bytes;
    return c;
}
static oconst kint(int64_t v) { oconst k; memset(&k, 0, sizeof k); k.type = OT_INT32; k.i = v; return k; }
static oconst kdate(int32_t d) { oconst k; memset(&k, 0, sizeof k); k.type = OT_DATE; k.i = d; return k; }
static oconst kstr(const char *s) { oconst k; memset(&k, 0, sizeof k); k.type = OT_VARCHAR; k.s = s; return k; }
static int64_t *i64buf(int64_t n) { return (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1)); }
static odec dec_unscaled(int64_t v, int scale) {
    int64_t p = 1;
    for (int i = 0; i < scale; i++) p *= 10;
    odec d;
    if (odec_new_from_int64(v / p, v % p, scale, &d) != ODEC_OK) abort();
    return d;
}
typedef struct { char *buf; int64_t cap, len; } sbuf3;
static void put(sbuf3 *s, const char *t) {
    int64_t n = (int64_t)strlen(t);
    if (s->len + n < s->cap) memcpy(s->buf + s->len, t, (size_t)n);
    s->len += n;
}
static void putn(sbuf3 *s, const char *t, int64_t n) {
    if (s->len + n < s->cap) memcpy(s->buf + s->len, t, (size_t)n);
    s->len += n;
}
static int64_t done(sbuf3 *s) { if (s->len < s->cap) s->buf[s->len] = 0; else if (s->cap > 0) s->buf[s->cap - 1] = 0; return s->len; }

/* ------------------------------------------------------------------ Q16
 * Order(supplier_cnt desc, p_brand, p_type, p_size) <- Agg(p_brand, p_type, p_size; count(distinct ps_suppkey))
 *   <- ANTI Join(ps_suppkey = s_suppkey)  probe Join(ps_partkey = p_partkey) probe Scan(partsupp),
 *                                                build Scan(part, p_brand <> .. and p_type not like .. and p_size in (..))
 *                                         build Scan(supplier, s_comment like '%Customer%Complaints%')
 * `<>` and NOT LIKE on VARCHAR: notEqualStrOp / notLikeOp; the IN list is an OR of INTEGER `=` (execSelectOr); NOT IN (subquery) is the
 * ANTI join; count(distinct x): the raw rows create the groups (AddChunk with the distinct aggregate filtered out), the distinct
 * (group keys, x) rows — a side table keyed by all four — are sunk into the aggregate at finalize (aggregate_exec.go:201-304). */
int64_t oracle_q16(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int32_t *s_comment_off,
                   const char *s_comment_bytes, const char *brand_ne, const char *type_notlike, const int32_t *sizes, int32_t nsizes,
                   const char *comment_like, oracle_q16_row *out, int64_t max) {
    /* part side */
    int64_t *p1 = i64buf(T->n_part), *p2 = i64buf(T->n_part), *p3 = i64buf(T->n_part);
    ocol pb = mkcode(T->p_brand, T->brand_dict), pt = mkcode(T->p_type, T->type_dict), psz = mkcol(OT_INT32, 0, T->p_size);
    oconst kb = kstr(brand_ne), kt = kstr(type_notlike);
    int64_t c = oracle_select(&pb, OP_NE, &kb, NULL, T->n_part, p1);
    c = oracle_select(&pt, OP_NOTLIKE, &kt, p1, c, p2);
    {
        ocol cols[16];
        int32_t ops[16];
        oconst ks[16];
        if (nsizes > 16) nsizes = 16;
        for (int i = 0; i < nsizes; i++) { cols[i] = psz; ops[i] = OP_EQ; ks[i] = kint(sizes[i]); }
        c = oracle_select_or(cols, ops, ks, nsizes, p2, c, p3);
    }
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, p3, c);
    int64_t *ps_row = i64buf(n_ps), *p_row = i64buf(n_ps);
    ocol psk = mkcol(OT_INT32, 0, ps_partkey);
    int64_t n1 = oracle_join_probe_inner(jp, &psk, 1, NULL, n_ps, ps_row, p_row, n_ps);
    oracle_join_free(jp);
    /* suppliers with a complaint */
    int64_t *ssel = i64buf(T->n_supplier);
    ocol sc = mkstr(s_comment_off, s_comment_bytes);
    oconst kl = kstr(comment_like);
    int64_t ns = oracle_select(&sc, OP_LIKE, &kl, NULL, T->n_supplier, ssel);
    ocol sk = mkcol(OT_INT32, 0, T->s_suppkey);
    ojoin *js = oracle_join_build(&sk, 1, ssel, ns);
    uint8_t *found = (uint8_t *)malloc((size_t)(n1 > 0 ? n1 : 1));
    ocol pss = mkcol(OT_INT32, 0, ps_suppkey);
    oracle_join_probe_mark(js, &pss, 1, ps_row, n1, found);     /* probe rows = the partsupp rows of the pairs, in pair order */
    oracle_join_free(js);
    int64_t n2 = 0;
    for (int64_t i = 0; i < n1; i++) if (!found[i]) { ps_row[n2] = ps_row[i]; p_row[n2] = p_row[i]; n2++; }   /* ANTI: rows without a match */
    /* the aggregate with its distinct side table */
    ocol kp[3] = {mkcode(NULL, T->brand_dict), mkcode(NULL, T->type_dict), mkcol(OT_INT32, 0, NULL)};
    ocol ap[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec aggs[1] = {{OA_COUNT, 0}};
    oagg *t = oracle_agg_create(kp, 3, ap, aggs, 1);
    ocol dkp[4] = {mkcode(NULL, T->brand_dict), mkcode(NULL, T->type_dict), mkcol(OT_INT32, 0, NULL), mkcol(OT_INT32, 0, NULL)};
    oagg *d = oracle_agg_create(dkp, 4, ap, aggs, 1);
    uint8_t bv[VS], tv[VS];
    int32_t sv[VS], kv[VS];
    int rc = 0;
    for (int64_t base = 0; base < n2 && rc == 0; base += VS) {
        int64_t cnt = n2 - base < VS ? n2 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t p = p_row[base + j];
            bv[j] = T->p_brand[p]; tv[j] = T->p_type[p]; sv[j] = T->p_size[p]; kv[j] = ps_suppkey[ps_row[base + j]];
        }
        ocol keys[4] = {mkcode(bv, T->brand_dict), mkcode(tv, T->type_dict), mkcol(OT_INT32, 0, sv), mkcol(OT_INT32, 0, kv)};
        ocol args[1] = {mkcol(OT_INT32, 0, kv)};
        rc = oracle_agg_sink_filtered(t, keys, args, NULL, cnt, 0u);          /* groups only: the one aggregate is DISTINCT */
        if (rc == 0) rc = oracle_agg_sink_filtered(d, keys, args, NULL, cnt, 0u);   /* the distinct table's rows are its groups */
    }
    /* finalize: the distinct (keys, argument) rows feed the aggregate */
    int64_t nd = rc ? 0 : oracle_agg_count(d);
    for (int64_t base = 0; base < nd && rc == 0; base += VS) {
        int64_t cnt = nd - base < VS ? nd - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            int64_t k4[4];
            oaggval unused;
            oracle_agg_group(d, base + j, NULL, k4, NULL, &unused);
            bv[j] = (uint8_t)k4[0]; tv[j] = (uint8_t)k4[1]; sv[j] = (int32_t)k4[2]; kv[j] = (int32_t)k4[3];
        }
        ocol keys[3] = {mkcode(bv, T->brand_dict), mkcode(tv, T->type_dict), mkcol(OT_INT32, 0, sv)};
        ocol args[1] = {mkcol(OT_INT32, 0, kv)};
        rc = oracle_agg_sink_filtered(t, keys, args, NULL, cnt, 1u);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t k3[3];
        oaggval v;
        oracle_agg_group(t, g, NULL, k3, NULL, &v);
        out[g].brand = (int32_t)k3[0]; out[g].type = (int32_t)k3[1]; out[g].size = (int32_t)k3[2];
        out[g].supplier_cnt = v.kind == OV_HUGEINT ? v.h : (ohuge){0, 0};
        out[g].cnt_null = v.kind == OV_NULL;
    }
    oracle_agg_free(t); oracle_agg_free(d);
    free(p1); free(p2); free(p3); free(ps_row); free(p_row); free(ssel); free(found);
    return ng;
}

static const char *const *g16_brand, *const *g16_type;
static int q16_cmp(const void *a, const void *b) {
    const oracle_q16_row *x = (const oracle_q16_row *)a, *y = (const oracle_q16_row *)b;
    if (x->supplier_cnt.lower != y->supplier_cnt.lower) return x->supplier_cnt.lower > y->supplier_cnt.lower ? -1 : 1;   /* desc */
    int c = strcmp(g16_brand[x->brand], g16_brand[y->brand]);
    if (c) return c;
    c = strcmp(g16_type[x->type], g16_type[y->type]);
    if (c) return c;
    return x->size < y->size ? -1 : x->size > y->size;
}
int64_t oracle_q16_text(oracle_q16_row *rows, int64_t n, const char *const *brand_dict, const char *const *type_dict, char *buf, int64_t cap) {
    g16_brand = brand_dict; g16_type = type_dict;
    qsort(rows, (size_t)n, sizeof *rows, q16_cmp);
    sbuf3 s = {buf, cap, 0};
    put(&s, "#\t\t\t\n");
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        put(&s, brand_dict[rows[i].brand]); put(&s, "\t");
        put(&s, type_dict[rows[i].type]); put(&s, "\t");
        sprintf(t, "%d\t", rows[i].size); put(&s, t);
        if (rows[i].cnt_null) put(&s, "NULL"); else { oracle_format_hugeint(rows[i].supplier_cnt, t); put(&s, t); }
        put(&s, "\n");
    }
    return done(&s);
}

/* ------------------------------------------------------------------ Q13
 * Order(custdist desc, c_count desc) <- Agg(c_count; count(*)) <- Agg(c_custkey; count(o_orderkey))
 *   <- LEFT Join(c_custkey = o_custkey)  probe Scan(customer),  build Scan(orders, o_comment not like '%..%..%')
 * NextLeftJoin (join_scan.go:67-88): per probe chunk the inner matches, then the probe rows without one with the build side's columns
 * NULL. count(o_orderkey) skips the NULLs (IgnoreNull), and CountOp.Finalize (function_aggr.go:950-962) turns a count of 0 into NULL:
 * the customers without a qualifying order form the group whose key is NULL (the golden's first row). The ON clause's NOT LIKE belongs to
 * the build side (it names only orders). */
int64_t oracle_q13(const oracle_tpch *T, const int32_t *o_comment_off, const char *o_comment_bytes, const char *notlike, oracle_q13_row *out, int64_t max) {
    int64_t *osel = i64buf(T->n_orders);
    ocol oc = mkstr(o_comment_off, o_comment_bytes);
    oconst kl = kstr(notlike);
    int64_t no = oracle_select(&oc, OP_NOTLIKE, &kl, NULL, T->n_orders, osel);
    ocol ock = mkcol(OT_INT32, 0, T->o_custkey);
    ojoin *j = oracle_join_build(&ock, 1, osel, no);
    /* the inner aggregate, fed probe chunk by probe chunk: the chunk's pairs (o_orderkey valid), then its unmatched rows (o_orderkey NULL) */
    ocol kp[1] = {mkcol(OT_INT32, 0, NULL)};
    ocol ap[1] = {mkcol(OT_INT64, 0, NULL)};
    oaggspec aggs[1] = {{OA_COUNT, 0}};
    oagg *inner = oracle_agg_create(kp, 1, ap, aggs, 1);
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    int64_t cap = 1 << 16;
    int64_t *pr = i64buf(cap), *br = i64buf(cap);
    int32_t kv[VS];
    int64_t av[VS];
    uint8_t valid[VS / 8];
    int rc = 0;
    for (int64_t base = 0; base < T->n_customer && rc == 0; base += VS) {
        const int64_t cnt = T->n_customer - base < VS ? T->n_customer - base : VS;
        int64_t *sel = i64buf(cnt);
        for (int64_t i = 0; i < cnt; i++) sel[i] = base + i;
        int64_t m = oracle_join_probe_inner(j, &ck, 1, sel, cnt, pr, br, cap);
        if (m > cap) { cap = m; free(pr); free(br); pr = i64buf(cap); br = i64buf(cap); m = oracle_join_probe_inner(j, &ck, 1, sel, cnt, pr, br, cap); }
        uint8_t matched[VS];
        memset(matched, 0, sizeof matched);
        for (int64_t b2 = 0; b2 < m && rc == 0; b2 += VS) {
            const int64_t c2 = m - b2 < VS ? m - b2 : VS;
            for (int64_t i = 0; i < c2; i++) { kv[i] = T->c_custkey[pr[b2 + i]]; av[i] = T->o_orderkey[br[b2 + i]]; matched[pr[b2 + i] - base] = 1; }
            ocol keys[1] = {mkcol(OT_INT32, 0, kv)}, args[1] = {mkcol(OT_INT64, 0, av)};
            rc = oracle_agg_sink(inner, keys, args, NULL, c2);
        }
        int64_t nu = 0;
        for (int64_t i = 0; i < cnt; i++) if (!matched[i]) { kv[nu] = T->c_custkey[base + i]; av[nu] = 0; nu++; }
        if (nu > 0 && rc == 0) {
            memset(valid, 0, sizeof valid);                       /* the build side's columns are NULL for these rows */
            ocol keys[1] = {mkcol(OT_INT32, 0, kv)}, args[1] = {mkcol(OT_INT64, 0, av)};
            args[0].validity = valid;
            rc = oracle_agg_sink(inner, keys, args, NULL, nu);
        }
        free(sel);
    }
    oracle_join_free(j);
    /* the outer aggregate over the inner one's groups: key = c_count, a HUGEINT that is NULL where the count was 0 */
    int64_t ngi = rc ? 0 : oracle_agg_count(inner);
    ocol okp[1] = {mkcol(OT_INT64, 0, NULL)};
    ocol oap[1] = {mkcol(OT_INT32, 0, NULL)};
    oaggspec star[1] = {{OA_COUNT, -1}};   /* count(*) */
    oagg *outer = oracle_agg_create(okp, 1, oap, star, 1);
    int32_t ones[VS];
    for (int i = 0; i < VS; i++) ones[i] = 1;
    for (int64_t base = 0; base < ngi && rc == 0; base += VS) {
        const int64_t cnt = ngi - base < VS ? ngi - base : VS;
        memset(valid, 0xFF, sizeof valid);
        for (int64_t i = 0; i < cnt; i++) {
            int64_t k1[1];
            oaggval v;
            oracle_agg_group(inner, base + i, NULL, k1, NULL, &v);
            if (v.kind == OV_HUGEINT) av[i] = (int64_t)v.h.lower;
            else { av[i] = 0; valid[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
        }
        ocol keys[1] = {mkcol(OT_INT64, 0, av)}, args[1] = {mkcol(OT_INT32, 0, ones)};
        keys[0].validity = valid;
        rc = oracle_agg_sink(outer, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(outer);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t k1[1];
        uint8_t kn[1];
        oaggval v;
        oracle_agg_group(outer, g, NULL, k1, kn, &v);
        out[g].c_count = k1[0];
        out[g].c_count_null = kn[0];
        out[g].custdist = v.kind == OV_HUGEINT ? v.h : (ohuge){0, 0};
    }
    oracle_agg_free(inner); oracle_agg_free(outer);
    free(osel); free(pr); free(br);
    return ng;
}

static int q13_cmp(const void *a, const void *b) {
    const oracle_q13_row *x = (const oracle_q13_row *)a, *y = (const oracle_q13_row *)b;
    if (x->custdist.lower != y->custdist.lower) return x->custdist.lower > y->custdist.lower ? -1 : 1;   /* custdist desc */
    if (x->c_count_null != y->c_count_null) return x->c_count_null ? -1 : 1;                            /* NULLs first (sort_layout.go:46) */
    return x->c_count > y->c_count ? -1 : x->c_count < y->c_count;                                       /* c_count desc */
}
int64_t oracle_q13_text(oracle_q13_row *rows, int64_t n, char *buf, int64_t cap) {
    qsort(rows, (size_t)n, sizeof *rows, q13_cmp);
    sbuf3 s = {buf, cap, 0};
    put(&s, "#\t\n");
    char t[64];
    for (int64_t i = 0; i < n; i++) {
        if (rows[i].c_count_null) put(&s, "NULL"); else { sprintf(t, "%lld", (long long)rows[i].c_count); put(&s, t); }
        put(&s, "\t");
        oracle_format_hugeint(rows[i].custdist, t); put(&s, t); put(&s, "\n");
    }
    return done(&s);
}

/* ------------------------------------------------------------------ Q2
 * Limit <- Order(s_acctbal desc, n_name, s_name, p_partkey) <- Join(ps_partkey = sub.ps_partkey and ps_supplycost = sub.min)
 *   probe  Join(p_partkey = ps_partkey) [part filtered by p_size = .. and p_type like ..] x partsupp x supplier x nation x region[r_name = ..]
 *   build  Agg(ps_partkey; min(ps_supplycost)) <- partsupp x supplier x nation x region[r_name = ..]   (the correlated subquery by its key)
 * The region-side chain is the same in both branches: the partsupp rows whose supplier is of the region. min(DECIMAL): MinMaxOp
 * (function_aggr.go:968-1027); the value equality is a join condition (hash + Match on the decimals), exact. */
int64_t oracle_q2(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int64_t *ps_supplycost, int32_t size,
                  const char *type_like, const char *region, oracle_q2_row *out, int64_t max) {
    /* region -> nations -> suppliers of the region */
    int64_t rsel[8], nsel[32];
    ocol rn = mkcode(T->r_name, T->region_dict);
    oconst kr = kstr(region);
    int64_t nr = oracle_select(&rn, OP_EQ, &kr, NULL, 5, rsel);
    ocol rk = mkcol(OT_INT32, 0, T->r_regionkey);
    ojoin *jr = oracle_join_build(&rk, 1, rsel, nr);
    uint8_t nf[32];
    ocol nrk = mkcol(OT_INT32, 0, T->n_regionkey);
    oracle_join_probe_mark(jr, &nrk, 1, NULL, 25, nf);
    oracle_join_free(jr);
    int64_t nn = 0;
    for (int64_t i = 0; i < 25; i++) if (nf[i]) nsel[nn++] = i;
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, nsel, nn);
    int64_t *s_row = i64buf(T->n_supplier), *n_row = i64buf(T->n_supplier);
    ocol snk = mkcol(OT_INT32, 0, T->s_nationkey);
    int64_t nsup = oracle_join_probe_inner(jn, &snk, 1, NULL, T->n_supplier, s_row, n_row, T->n_supplier);
    oracle_join_free(jn);
    int32_t *skey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nsup > 0 ? nsup : 1));
    for (int64_t i = 0; i < nsup; i++) skey[i] = T->s_suppkey[s_row[i]];
    ocol skc = mkcol(OT_INT32, 0, skey);
    ojoin *js = oracle_join_build(&skc, 1, NULL, nsup);
    int64_t *ps_row = i64buf(n_ps), *sp = i64buf(n_ps);
    ocol pssk = mkcol(OT_INT32, 0, ps_suppkey);
    int64_t nps = oracle_join_probe_inner(js, &pssk, 1, NULL, n_ps, ps_row, sp, n_ps);      /* partsupp rows of the region's suppliers */
    oracle_join_free(js);
    /* the subquery: min(ps_supplycost) by ps_partkey over those rows */
    ocol kp[1] = {mkcol(OT_INT32, 0, NULL)};
    ocol ap[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_MIN, 0}};
    oagg *sub = oracle_agg_create(kp, 1, ap, aggs, 1);
    static odec dv[VS];
    int32_t kv[VS];
    int rc = 0;
    for (int64_t base = 0; base < nps && rc == 0; base += VS) {
        const int64_t cnt = nps - base < VS ? nps - base : VS;
        for (int64_t i = 0; i < cnt; i++) { kv[i] = ps_partkey[ps_row[base + i]]; dv[i] = dec_unscaled(ps_supplycost[ps_row[base + i]], 2); }
        ocol keys[1] = {mkcol(OT_INT32, 0, kv)}, args[1] = {mkcol(OT_ODEC, 0, dv)};
        rc = oracle_agg_sink(sub, keys, args, NULL, cnt);
    }
    int64_t ngs = rc ? 0 : oracle_agg_count(sub);
    int32_t *mkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ngs > 0 ? ngs : 1));
    int64_t *mval = i64buf(ngs);
    for (int64_t g = 0; g < ngs; g++) {
        int64_t k1[1];
        oaggval v;
        oracle_agg_group(sub, g, NULL, k1, NULL, &v);
        __int128 u = 0;
        odec_to_unscaled(v.d, 2, &u);
        mkey[g] = (int32_t)k1[0]; mval[g] = (int64_t)u;
    }
    oracle_agg_free(sub);
    /* the outer branch: the same partsupp rows joined with the filtered parts ... */
    int64_t *p1 = i64buf(T->n_part), *p2 = i64buf(T->n_part);
    ocol psz = mkcol(OT_INT32, 0, T->p_size), pt = mkcode(T->p_type, T->type_dict);
    oconst ks = kint(size), kt = kstr(type_like);
    int64_t np = oracle_select(&psz, OP_EQ, &ks, NULL, T->n_part, p1);
    np = oracle_select(&pt, OP_LIKE, &kt, p1, np, p2);
    ocol pk = mkcol(OT_INT32, 0, T->p_partkey);
    ojoin *jp = oracle_join_build(&pk, 1, p2, np);
    int32_t *ppk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nps > 0 ? nps : 1));
    int64_t *pcost = i64buf(nps);
    for (int64_t i = 0; i < nps; i++) { ppk[i] = ps_partkey[ps_row[i]]; pcost[i] = ps_supplycost[ps_row[i]]; }
    ocol ppkc = mkcol(OT_INT32, 0, ppk);
    int64_t *a_row = i64buf(nps), *p_row = i64buf(nps);
    int64_t n1 = oracle_join_probe_inner(jp, &ppkc, 1, NULL, nps, a_row, p_row, nps);      /* a_row: position in the region's partsupp rows */
    oracle_join_free(jp);
    /* ... and with the subquery's rows on (partkey, cost = min) */
    ocol bk[2] = {mkcol(OT_INT32, 0, mkey), mkcol(OT_DECIMAL, 2, mval)};
    ojoin *jm = oracle_join_build(bk, 2, NULL, ngs);
    int32_t *qk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int64_t *qc = i64buf(n1);
    for (int64_t i = 0; i < n1; i++) { qk[i] = ppk[a_row[i]]; qc[i] = pcost[a_row[i]]; }
    ocol qcols[2] = {mkcol(OT_INT32, 0, qk), mkcol(OT_DECIMAL, 2, qc)};
    int64_t *f_row = i64buf(n1), *g_row = i64buf(n1);
    int64_t n2 = rc ? -1 : oracle_join_probe_inner(jm, qcols, 2, NULL, n1, f_row, g_row, n1);
    oracle_join_free(jm);
    for (int64_t i = 0; i < n2 && i < max; i++) {
        const int64_t a = a_row[f_row[i]];        /* the partsupp row of the region branch */
        out[i].s_row = (int32_t)s_row[sp[a]];
        out[i].nation = (int32_t)n_row[sp[a]];
        out[i].p_row = (int32_t)p_row[f_row[i]];
    }
    free(s_row); free(n_row); free(skey); free(ps_row); free(sp); free(mkey); free(mval); free(p1); free(p2); free(ppk); free(pcost);
    free(a_row); free(p_row); free(qk); free(qc); free(f_row); free(g_row);
    return n2;
}

static const oracle_tpch *g2_T;
static const int64_t *g2_acctbal;
static int q2_cmp(const void *a, const void *b) {
    const oracle_q2_row *x = (const oracle_q2_row *)a, *y = (const oracle_q2_row *)b;
    const int64_t ax = g2_acctbal[x->s_row], ay = g2_acctbal[y->s_row];
    if (ax != ay) return ax > ay ? -1 : 1;                                                    /* s_acctbal desc */
    int c = strcmp(g2_T->nation_dict[g2_T->n_name[x->nation]], g2_T->nation_dict[g2_T->n_name[y->nation]]);
    if (c) return c;
    if (g2_T->s_suppkey[x->s_row] != g2_T->s_suppkey[y->s_row]) return g2_T->s_suppkey[x->s_row] < g2_T->s_suppkey[y->s_row] ? -1 : 1;   /* s_name: zero padded */
    return g2_T->p_partkey[x->p_row] < g2_T->p_partkey[y->p_row] ? -1 : g2_T->p_partkey[x->p_row] > g2_T->p_partkey[y->p_row];
}
int64_t oracle_q2_text(oracle_q2_row *rows, int64_t n, int32_t limit, const oracle_tpch *T, const int64_t *s_acctbal, const uint8_t *p_mfgr,
                       const int32_t *addr_off, const char *addr_bytes, const char *phone_bytes, const int32_t *cmnt_off, const char *cmnt_bytes,
                       char *buf, int64_t cap) {
    g2_T = T; g2_acctbal = s_acctbal;
    qsort(rows, (size_t)n, sizeof *rows, q2_cmp);
    sbuf3 s = {buf, cap, 0};
    put(&s, "#\t\t\t\t\t\t\t\n");
    char t[160];
    for (int64_t i = 0; i < n && i < limit; i++) {
        const int64_t r = rows[i].s_row;
        oracle_format_decimal(dec_unscaled(s_acctbal[r], 2), 2, t); put(&s, t); put(&s, "\t");
        sprintf(t, "Supplier#%09d\t", T->s_suppkey[r]); put(&s, t);
        put(&s, T->nation_dict[T->n_name[rows[i].nation]]); put(&s, "\t");
        sprintf(t, "%d\tManufacturer#%d\t", T->p_partkey[rows[i].p_row], p_mfgr[rows[i].p_row] + 1); put(&s, t);
        putn(&s, addr_bytes + addr_off[r], addr_off[r + 1] - addr_off[r]); put(&s, "\t");
        putn(&s, phone_bytes + 15 * r, 15); put(&s, "\t");
        putn(&s, cmnt_bytes + cmnt_off[r], cmnt_off[r + 1] - cmnt_off[r]); put(&s, "\n");
    }
    return done(&s);
}

/* ------------------------------------------------------------------ Q10
 * Limit <- Order(revenue desc) <- Agg(c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment; sum(l_extendedprice * (1 - l_discount)))
 *   <- Join(c_nationkey = n_nationkey) <- Join(c_custkey = o_custkey) <- Join(l_orderkey = o_orderkey)
 *        probe Scan(lineitem, l_returnflag = 'R'), build Scan(orders, o_orderdate in [d, d + 3 months)); build customer; build nation
 * The four VARCHAR group keys (c_name = 'Customer#' + the key, c_phone, c_address, c_comment) and c_acctbal are columns of the customer ROW the
 * key c_custkey names, n_name of the nation that row names: equal c_custkey means every other key equal, so the groups — and their
 * first-seen order — are those of (c_custkey, c_acctbal, n_name); the group table's byte compare of the strings (pinned by Q18's c_name) has
 * nothing left to separate. */
int64_t oracle_q10(const oracle_tpch *T, const uint8_t *l_returnflag, const char *const *returnflag_dict, const int64_t *c_acctbal, const char *flag,
                   int32_t date_ge, int32_t date_lt, oracle_q10_row *out, int64_t max) {
    int64_t *o1 = i64buf(T->n_orders), *o2 = i64buf(T->n_orders);
    ocol od = mkcol(OT_DATE, 0, T->o_orderdate);
    oconst k1 = kdate(date_ge), k2 = kdate(date_lt);
    int64_t no = oracle_select(&od, OP_GE, &k1, NULL, T->n_orders, o1);
    no = oracle_select(&od, OP_LT, &k2, o1, no, o2);
    ocol ok = mkcol(OT_INT64, 0, T->o_orderkey);
    ojoin *jo = oracle_join_build(&ok, 1, o2, no);
    int64_t *lsel = i64buf(T->n_lineitem);
    ocol lf = mkcode(l_returnflag, returnflag_dict);
    oconst kf = kstr(flag);
    int64_t nl = oracle_select(&lf, OP_EQ, &kf, NULL, T->n_lineitem, lsel);
    int64_t *l_row = i64buf(nl), *o_row = i64buf(nl);
    ocol lk = mkcol(OT_INT64, 0, T->l_orderkey);
    int64_t n1 = oracle_join_probe_inner(jo, &lk, 1, lsel, nl, l_row, o_row, nl);
    oracle_join_free(jo);
    ocol ck = mkcol(OT_INT32, 0, T->c_custkey);
    ojoin *jc = oracle_join_build(&ck, 1, NULL, T->n_customer);
    int32_t *ock = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    for (int64_t i = 0; i < n1; i++) ock[i] = T->o_custkey[o_row[i]];
    ocol ockc = mkcol(OT_INT32, 0, ock);
    int64_t *a_row = i64buf(n1), *c_row = i64buf(n1);
    int64_t n2 = oracle_join_probe_inner(jc, &ockc, 1, NULL, n1, a_row, c_row, n1);
    oracle_join_free(jc);
    ocol nk = mkcol(OT_INT32, 0, T->n_nationkey);
    ojoin *jn = oracle_join_build(&nk, 1, NULL, 25);
    int32_t *cnk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    for (int64_t i = 0; i < n2; i++) cnk[i] = T->c_nationkey[c_row[i]];
    ocol cnkc = mkcol(OT_INT32, 0, cnk);
    int64_t *b_row = i64buf(n2), *n_row = i64buf(n2);
    int64_t n3 = oracle_join_probe_inner(jn, &cnkc, 1, NULL, n2, b_row, n_row, n2);
    oracle_join_free(jn);
    ocol kp[3] = {mkcol(OT_INT32, 0, NULL), mkcol(OT_DECIMAL, 2, NULL), mkcode(NULL, T->nation_dict)};
    ocol ap[1] = {mkcol(OT_ODEC, 0, NULL)};
    oaggspec aggs[1] = {{OA_SUM, 0}};
    oagg *t = oracle_agg_create(kp, 3, ap, aggs, 1);
    static const orpn DISC_PRICE[5] = {{OX_COL, 0, 0, 0}, {OX_CONST_INT, 0, 1, 0}, {OX_COL, 1, 0, 0}, {OX_SUB, 0, 0, 0}, {OX_MUL, 0, 0, 0}};
    static odec v[VS];
    int64_t ext[VS], disc[VS], bal[VS];
    int32_t ckv[VS];
    uint8_t nat[VS];
    int32_t *crow_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T->n_customer > 0 ? T->n_customer : 1));   /* custkey - 1 -> row (keys are 1..n in order) */
    (void)crow_of;
    int rc = 0;
    for (int64_t base = 0; base < n3 && rc == 0; base += VS) {
        const int64_t cnt = n3 - base < VS ? n3 - base : VS;
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t b = b_row[base + j], a = a_row[b], l = l_row[a], c = c_row[b];
            ext[j] = T->l_extendedprice[l]; disc[j] = T->l_discount[l];
            ckv[j] = T->c_custkey[c]; bal[j] = c_acctbal[c]; nat[j] = T->n_name[n_row[base + j]];
        }
        ocol cols[2] = {mkcol(OT_DECIMAL, 2, ext), mkcol(OT_DECIMAL, 2, disc)};
        rc = oracle_eval_decimal(cols, DISC_PRICE, 5, NULL, cnt, v);
        ocol keys[3] = {mkcol(OT_INT32, 0, ckv), mkcol(OT_DECIMAL, 2, bal), mkcode(nat, T->nation_dict)};
        ocol args[1] = {mkcol(OT_ODEC, 0, v)};
        if (rc == 0) rc = oracle_agg_sink(t, keys, args, NULL, cnt);
    }
    int64_t ng = rc ? -1 : oracle_agg_count(t);
    for (int64_t g = 0; g < ng && g < max; g++) {
        int64_t k3[3];
        oaggval val;
        oracle_agg_group(t, g, NULL, k3, NULL, &val);
        out[g].c_custkey = (int32_t)k3[0];
        out[g].nation_code = (int32_t)k3[2];
        out[g].revenue = val.d;
    }
    oracle_agg_free(t);
    free(o1); free(o2); free(lsel); free(l_row); free(o_row); free(ock); free(a_row); free(c_row); free(cnk); free(b_row); free(n_row); free(crow_of);
    return ng;
}

static int q10_cmp(const void *a, const void *b) {
    const oracle_q10_row *x = (const oracle_q10_row *)a, *y = (const oracle_q10_row *)b;
    int c = odec_cmp(y->revenue, x->revenue);   /* revenue desc */
    if (c) return c;
    return x->c_custkey < y->c_custkey ? -1 : x->c_custkey > y->c_custkey;
}
int64_t oracle_q10_text(oracle_q10_row *rows, int64_t n, int32_t limit, const oracle_tpch *T, const int64_t *c_acctbal, const int32_t *addr_off,
                        const char *addr_bytes, const char *phone_bytes, const int32_t *cmnt_off, const char *cmnt_bytes, char *buf, int64_t cap) {
    qsort(rows, (size_t)n, sizeof *rows, q10_cmp);
    sbuf3 s = {buf, cap, 0};
    put(&s, "#\t\t\t\t\t\t\t\n");
    char t[160];
    for (int64_t i = 0; i < n && i < limit; i++) {
        const int64_t r = rows[i].c_custkey - 1;   /* customer keys are 1..n in row order */
        if (r < 0 || r >= T->n_customer || T->c_custkey[r] != rows[i].c_custkey) continue;
        sprintf(t, "%d\tCustomer#%09d\t", rows[i].c_custkey, rows[i].c_custkey); put(&s, t);
        oracle_format_decimal(rows[i].revenue, 4, t); put(&s, t); put(&s, "\t");
        oracle_format_decimal(dec_unscaled(c_acctbal[r], 2), 2, t); put(&s, t); put(&s, "\t");
        put(&s, T->nation_dict[rows[i].nation_code]); put(&s, "\t");
        putn(&s, addr_bytes + addr_off[r], addr_off[r + 1] - addr_off[r]); put(&s, "\t");
        putn(&s, phone_bytes + 15 * r, 15); put(&s, "\t");
        putn(&s, cmnt_bytes + cmnt_off[r], cmnt_off[r + 1] - cmnt_off[r]); put(&s, "\n");
    }
    return done(&s);
}
