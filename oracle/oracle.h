/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 * Nothing under plan_amd/ includes, links or calls it; the product path never routes here.
 *
 * What it is: a plain-C, single-threaded CPU restatement of the reference's hot path
 * (pkg/compute: PhysicalScan -> Filter -> HashJoin build/probe -> HashAggregate over
 * pkg/chunk vectors), chunk-at-a-time (2048 rows, pkg/util/util.go:123-125), with the
 * reference's value semantics: selection vectors, decimal = (neg, coef, scale) exact integer
 * arithmetic, Hugeint sums, float64 AVG for INTEGER input, salted linear-probing group table,
 * chained-bucket join table, groups emitted in first-seen order, print-time decimal rounding.
 * Each function cites the reference file:line it follows.
 *
 * Pinned by: the reference's own SF1 goldens cases/tpch/1g/plan/q{1,3,6,9}.txt (copied to
 * tests/golden/plan_q*.txt), reproduced byte-for-byte by oracle_q*_text() on the data of
 * include/tpchgen.h (tests/test_golden_tpch.py). The reference itself (Go 1.24 + cgo) cannot be
 * built in this environment, so there is no oracle/_ref.
 *
 * Input columns use the narrow encodings the reference's own loader reads from parquet
 * (executor_scan.go:410-466): INTEGER int32, BIGINT int64, DATE int32 days since 1970-01-01,
 * DECIMAL int64 unscaled + scale, VARCHAR as uint8 dictionary code + dictionary.
 * Validity: 1 bit per row, LSB first, NULL pointer = all valid (pkg/util/bitmap.go:27-44).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#include "odecimal.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_VECTOR_SIZE 2048

typedef enum {
    OT_INT32 = 1,
    OT_INT64 = 2,
    OT_DATE = 3,    /* int32 days since epoch; materialised as Date{Y,M,D} (pkg/common/date.go:8-12) */
    OT_DECIMAL = 4, /* int64 unscaled, `scale` */
    OT_CODE8 = 5,   /* uint8 dictionary code of a VARCHAR column */
    OT_FLOAT = 6,   /* float32 */
    OT_DOUBLE = 7,
    OT_ODEC = 8,    /* array of odec (results of decimal expressions) */
    OT_VARCHAR = 9, /* offsets(int32[n+1]) + bytes: data=offsets, dict=(const char*const*)bytes */
    OT_CONST32 = 10 /* one int32 repeated for every row (CONST vector; the constant group key of
                       an ungrouped aggregate, executor_aggr.go:37-48) */
} otype;

typedef struct {
    int32_t type;
    int32_t scale;
    const void *data;
    const uint8_t *validity;
    const char *const *dict; /* OT_CODE8: code -> NUL-terminated string */
    int32_t dict_size;
} ocol;

typedef enum { OP_EQ = 1, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_LIKE, OP_NOTLIKE } ocmp;

typedef struct {
    int32_t type; /* OT_INT32, OT_DATE (days), OT_FLOAT (f), OT_DECIMAL (i, scale), OT_VARCHAR (s) */
    int32_t scale;
    int64_t i;
    double f;
    const char *s;
} oconst;

/* Filter: one comparison `col OP const`, narrowing an optional input selection.
 * Follows ExprExec.executeSelect/execSelectCompare (expr_exec.go:342-442) and
 * selectOperation/selectFlat/selectFlatLoop (function_operator_boolean.go:393-521, 672-868),
 * including which (type, op) pairs are implemented at all: an unsupported pair selects NOTHING
 * (:447-459), e.g. `=` on BIGINT, `<` on FLOAT. A DECIMAL column compared with a FLOAT constant
 * is cast decimal -> float64 -> float32 first (tryCastDecimalToFloat32, function_cast.go:349-354).
 * sel_in == NULL means identity. Returns the number of rows written to sel_out (ascending). */
int64_t oracle_select(const ocol *col, int32_t op, const oconst *k, const int64_t *sel_in,
                      int64_t n_in, int64_t *sel_out);

/* column OP column over the same rows (selectBinary with two FLAT vectors, function_operator_boolean.go:506-521):
 * INTEGER all six operators, DATE the four orderings, DECIMAL '>' — the pairs selectOperation has; others select
 * nothing. Q4 / Q12: l_commitdate < l_receiptdate. */
int64_t oracle_select_cols(const ocol *a, int32_t op, const ocol *b, const int64_t *sel_in, int64_t n_in,
                           int64_t *sel_out);

/* OR of k comparisons (`a IN (x, y)` binds to in(a,x) OR in(a,y); `in` selects like `=` for
 * INTEGER and VARCHAR and nothing for other types, function_operator_boolean.go:419-429).
 * execSelectOr (expr_exec.go:488-530): child i is evaluated on the rows every earlier child
 * rejected, and its true rows are appended — so the output is child-major, not ascending.
 * Returns the number of rows written to sel_out. */
int64_t oracle_select_or(const ocol *cols, const int32_t *ops, const oconst *ks, int32_t k,
                         const int64_t *sel_in, int64_t n_in, int64_t *sel_out);

/* Chunk.Hash (pkg/chunk/chunk.go:160-166, hash.go:26-41, 182-413; util.HashBytes
 * pkg/util/hash.go:13-65): hash of the first column, combined with the others. */
void oracle_hash(const ocol *cols, int32_t ncols, int64_t n, uint64_t *out);

/* Decimal expression programs (RPN) — ExprExec.execute/executeFunc (expr_exec.go:85-340) with
 * binaryExecSwitch decimal ops (function_operator_binary.go:134-207, 267-481). */
typedef enum {
    OX_COL = 1,    /* push column `col` (DECIMAL as is; INT32 cast to decimal scale 0,
                      tryCastInt32ToDecimal function_cast.go:337-347) */
    OX_CONST_INT,  /* push integer literal ival cast to decimal (NewFromInt64(v,0,s) -> scale 0) */
    OX_CONST_DEC,  /* push decimal literal (ival unscaled, scale) */
    OX_ADD,
    OX_SUB,
    OX_MUL
} oxop;

typedef struct {
    int32_t op;
    int32_t col;
    int64_t ival;
    int32_t scale;
} orpn;

/* Evaluates prog for rows sel[0..n) (NULL = identity over n rows); out[i] is the value of
 * row sel[i]. Returns 0, or an ODEC_* error code (the reference panics -> query error). */
int oracle_eval_decimal(const ocol *cols, const orpn *prog, int32_t nprog, const int64_t *sel,
                        int64_t n, odec *out);

/* CASE WHEN <col OP const> THEN <then_prog> ELSE <else_prog> END over rows 0..n
 * (executeCase, expr_exec.go:144-246; FillSwitch/TemplatedFillLoop :559-606): the WHEN is a
 * select; THEN is evaluated on its true rows and filled into the result at those rows, ELSE on
 * the remaining rows. out[r] / out_null[r] per row. Returns 0 or an ODEC_* error. */
int oracle_case_decimal(const ocol *cols, const ocol *when_col, int32_t when_op, const oconst *when_k,
                        const orpn *then_prog, int32_t nthen, const orpn *else_prog, int32_t nelse,
                        int64_t n, odec *out, uint8_t *out_null);

/* ---- hash aggregate ---- */
typedef enum { OA_SUM = 1, OA_AVG, OA_COUNT, OA_MIN, OA_MAX } oaggkind;

typedef struct {
    int32_t kind;
    int32_t arg; /* index into args */
} oaggspec;

typedef struct {
    uint64_t lower;
    int64_t upper;
} ohuge; /* pkg/common/hugeint.go:8-11 */

typedef enum { OV_NULL = 0, OV_HUGEINT, OV_DECIMAL, OV_DOUBLE } ovalkind;

typedef struct {
    int32_t kind;
    ohuge h;
    odec d;
    double f;
} oaggval;

/* GroupedAggrHashTable (aggregate_hash.go:101-134 create, :136-199 AddChunk,
 * :201-391 FindOrCreateGroups, :440-513 Resize) + aggregate states (function_aggr.go:420-1032)
 * + FinalizeStates (:1330-1365).
 * Incremental form (what aggExecutor.Execute drives chunk by chunk, executor_aggr.go:110-142):
 * create -> sink(<=2048 positional rows)* -> read groups in first-seen (insertion) order.
 * keys: OT_INT32/OT_INT64/OT_DATE/OT_DECIMAL/OT_CODE8/OT_CONST32; args: OT_INT32 or OT_ODEC. */
typedef struct oagg oagg;
oagg *oracle_agg_create(const ocol *key_proto, int32_t nkeys, const ocol *arg_proto,
                        const oaggspec *aggs, int32_t naggs);
/* keys[c].data / args[c].data are positional buffers of `cnt` rows; row_ids (optional) are
 * reported back as group_first_row. Returns 0 or an error code. */
int oracle_agg_sink(oagg *t, const ocol *keys, const ocol *args, const int64_t *row_ids,
                    int64_t cnt);
/* AddChunk with its `filter []int` (aggregate_hash.go:155-199): groups are found or created for
 * every row, but only the aggregates whose bit is set in agg_mask are updated. The reference
 * sinks raw rows with the non-DISTINCT aggregates' filter and, at finalize, the rows of each
 * DISTINCT aggregate's own (group keys + argument) table with filter {i}
 * (aggregate_exec.go:74-99 SinkDistinctGrouping, :201-304 FinalizeDistinct/DistinctGrouping). */
int oracle_agg_sink_filtered(oagg *t, const ocol *keys, const ocol *args, const int64_t *row_ids,
                             int64_t cnt, uint32_t agg_mask);
int64_t oracle_agg_count(const oagg *t);
int oracle_agg_group(const oagg *t, int64_t g, int64_t *first_row, int64_t *key_vals,
                     uint8_t *key_null, oaggval *vals);
void oracle_agg_free(oagg *t);

/* One-shot form over whole columns addressed by row id: rows sel[0..n) (NULL = identity) are
 * consumed 2048 at a time. Outputs in first-seen order: group_first_row[g], group_keys[g*nkeys+c]
 * (widened to int64; codes for OT_CODE8), group_key_null, vals[g*naggs+a].
 * Returns the number of groups, or -1 on a decimal error. */
int64_t oracle_groupby(const ocol *keys, int32_t nkeys, const ocol *args, int32_t nargs,
                       const oaggspec *aggs, int32_t naggs, const int64_t *sel, int64_t n,
                       int64_t *group_first_row, int64_t *group_keys, uint8_t *group_key_null,
                       oaggval *vals, int64_t max_groups);

/* ---- ORDER BY ----
 * LocalSort for fixed-size keys: every row's key is the concatenation, per ORDER BY column, of
 * [1 = value / 0 = NULL (NULLs first, sort_layout.go:46)] + the encoder's bytes (INT32: big-endian
 * with the sign bit flipped; DATE: year, month, day as such int32s; DECIMAL: dec.Int64(2) ->
 * whole and frac as such int64s, sort_encoder.go:33-114), value bytes inverted for DESC
 * (TemplatedRadixScatter, sort_radix.go:324-380; a NULL's value bytes are zero). Rows are ordered
 * by memcmp of the keys (the reference's radix sort + pdqsort over the same bytes,
 * sort_local.go:128-250); equal keys are returned in input order here (undefined there).
 * key_len_out (optional) receives the key width; keys_out (optional, n * width bytes) the sorted keys.
 * Column types: OT_INT32, OT_DATE (days), OT_DECIMAL (unscaled int64 + scale), OT_CODE8 (compared
 * by code: what a VARCHAR key with a dictionary in ascending byte order amounts to). */
int oracle_sort_rows(const ocol *cols, const int32_t *descending, int32_t nkeys, const int64_t *sel,
                     int64_t n, int64_t *rows_out, int32_t *key_len_out, uint8_t *keys_out);

/* ---- hash join ---- */
typedef struct ojoin ojoin;

/* JoinHashTable.Build/prepareKeys/hash/Finalize/InsertHashesLoop (join_table.go:85-288):
 * rows with a NULL key are dropped, chains are head-inserted, cap = max(nextpow2(2n),1024). */
ojoin *oracle_join_build(const ocol *keys, int32_t nkeys, const int64_t *sel, int64_t n);
void oracle_join_free(ojoin *j);
int64_t oracle_join_count(const ojoin *j);

/* Inner probe: JoinHashTable.Probe (join_table.go:324-336), Scan.NextInnerJoin/InnerJoin/
 * advancePointers (join_scan.go:182-278), Match (util_match.go:25-301). Emits (probe row id,
 * build row id) pairs in the reference's order: per 2048-row probe chunk, per chain round.
 * Returns the pair count (pairs beyond `max` are counted but not stored). */
int64_t oracle_join_probe_inner(const ojoin *j, const ocol *keys, int32_t nkeys,
                                const int64_t *sel, int64_t n, int64_t *out_probe,
                                int64_t *out_build, int64_t max);

/* Semi/anti/mark: found[i] = 1 when probe row sel[i] has a match (ScanKeyMatches,
 * join_scan.go:166-180). */
void oracle_join_probe_mark(const ojoin *j, const ocol *keys, int32_t nkeys, const int64_t *sel,
                            int64_t n, uint8_t *found);

/* ---- query drivers (the pipelines of SURVEY.md §3.2/3.3) ---- */
typedef struct {
    uint8_t returnflag, linestatus; /* dictionary codes */
    ohuge sum_qty;
    odec sum_base_price, sum_disc_price, sum_charge;
    double avg_qty;
    odec avg_price, avg_disc;
    uint64_t count_order;
} oracle_q1_row;

typedef struct {
    const int32_t *l_quantity;
    const int64_t *l_extendedprice, *l_discount, *l_tax;
    const uint8_t *l_returnflag, *l_linestatus;
    const int32_t *l_shipdate;
    const int64_t *l_orderkey;
    const int32_t *l_partkey, *l_suppkey;
    const char *const *returnflag_dict; /* code -> string, for the VARCHAR hash/compare */
    const char *const *linestatus_dict;
    int64_t n;
} oracle_lineitem;

/* Q1 (cases/tpch/query/q1.sql): rows in first-seen order; returns group count. */
int32_t oracle_q1(const oracle_lineitem *L, int32_t shipdate_le, oracle_q1_row *out,
                  int32_t max_groups);
/* Q6: revenue = sum(l_extendedprice*l_discount); returns 0 ok / 1 when the sum is NULL. */
int32_t oracle_q6(const oracle_lineitem *L, int32_t date_ge, int32_t date_lt, float disc_lo,
                  float disc_hi, int32_t qty_lt, odec *revenue);

typedef struct {
    const int64_t *o_orderkey;
    const int32_t *o_custkey, *o_orderdate, *o_shippriority;
    int64_t n;
} oracle_orders;

typedef struct {
    const int32_t *c_custkey;
    const uint8_t *c_mktsegment;
    const char *const *mktsegment_dict;
    int32_t dict_size;
    int64_t n;
} oracle_customer;

typedef struct {
    int64_t l_orderkey;
    odec revenue;
    int32_t o_orderdate, o_shippriority;
} oracle_q3_row;

/* Q3 before ORDER BY/LIMIT: all groups in first-seen order. Returns the group count
 * (rows beyond max are counted, not stored). */
int64_t oracle_q3(const oracle_lineitem *L, const oracle_orders *O, const oracle_customer *C,
                  const char *segment, int32_t date, oracle_q3_row *out, int64_t max);

typedef struct {
    const int32_t *p_partkey;
    const int32_t *p_name_off; /* n+1 offsets */
    const char *p_name_bytes;
    int64_t n;
} oracle_part;

typedef struct {
    const int32_t *ps_partkey, *ps_suppkey;
    const int64_t *ps_supplycost;
    int64_t n;
} oracle_partsupp;

typedef struct {
    const int32_t *s_suppkey, *s_nationkey;
    int64_t n;
} oracle_supplier;

typedef struct {
    int32_t nationkey;
    int32_t o_year;
    odec sum_profit;
} oracle_q9_row;

int64_t oracle_q9(const oracle_lineitem *L, const oracle_orders *O, const oracle_part *P,
                  const oracle_partsupp *PS, const oracle_supplier *S, const char *like_pattern,
                  oracle_q9_row *out, int64_t max);

/* ---- round 3: Q4, Q5, Q12, Q14, Q19 (cases/tpch/query/q{4,5,12,14,19}.sql) — SEMI join, the six-table join chain,
 * IN / OR lists, CASE (integer and decimal branches, LIKE in a WHEN), column-vs-column comparisons and FLOAT
 * arithmetic over aggregate results. Pinned by cases/tpch/1g/plan/q{4,5,12,14,19}.txt (tests/golden/). */
typedef struct {
    /* lineitem */
    int64_t n_lineitem;
    const int64_t *l_orderkey, *l_extendedprice, *l_discount;
    const int32_t *l_partkey, *l_suppkey, *l_quantity, *l_shipdate, *l_commitdate, *l_receiptdate;
    const uint8_t *l_shipmode, *l_shipinstruct;
    /* orders */
    int64_t n_orders;
    const int64_t *o_orderkey;
    const int32_t *o_custkey, *o_orderdate;
    const uint8_t *o_orderpriority;
    /* customer, supplier, part */
    int64_t n_customer;
    const int32_t *c_custkey, *c_nationkey;
    int64_t n_supplier;
    const int32_t *s_suppkey, *s_nationkey;
    int64_t n_part;
    const int32_t *p_partkey, *p_size;
    const uint8_t *p_brand, *p_type, *p_container;
    /* nation (25), region (5) */
    const int32_t *n_nationkey, *n_regionkey, *r_regionkey;
    const uint8_t *n_name, *r_name;
    /* dictionaries (code -> string) */
    const char *const *shipmode_dict, *const *shipinstruct_dict, *const *orderpriority_dict, *const *brand_dict, *const *type_dict,
        *const *container_dict, *const *nation_dict, *const *region_dict;
} oracle_tpch;

typedef struct { int32_t code; ohuge count; } oracle_q4_row;          /* o_orderpriority code, count(*) */
typedef struct { int32_t nation; odec revenue; } oracle_q5_row;       /* n_name code, sum */
typedef struct { int32_t mode; ohuge high, low; } oracle_q12_row;     /* l_shipmode code, the two sums */
/* each returns the number of groups (first-seen order), -1 on a decimal error */
int64_t oracle_q4(const oracle_tpch *T, int32_t date_ge, int32_t date_lt, oracle_q4_row *out, int64_t max);
int64_t oracle_q5(const oracle_tpch *T, const char *region, int32_t date_ge, int32_t date_lt, oracle_q5_row *out, int64_t max);
int64_t oracle_q12(const oracle_tpch *T, const char *mode1, const char *mode2, int32_t date_ge, int32_t date_lt, oracle_q12_row *out, int64_t max);
/* promo_revenue = 100.00 * sum(case when p_type like pattern then e*(1-d) else 0 end) / sum(e*(1-d)), in float32 as the
 * binder types it (the literal is FLOAT; MaxLType(FLOAT, DECIMAL) = FLOAT): returns 0 ok, 1 NULL sums, -1 decimal error */
int32_t oracle_q14(const oracle_tpch *T, const char *like_pattern, int32_t date_ge, int32_t date_lt, float *promo_revenue, odec *promo, odec *total);
int32_t oracle_q19(const oracle_tpch *T, odec *revenue);   /* the query's own constants; 0 ok, 1 NULL, -1 error */
/* Q18 (cases/tpch/query/q18.sql): the IN (select l_orderkey .. group by l_orderkey having sum(l_quantity) > k) subquery as a
 * SEMI join against an aggregate with HAVING (HUGEINT '>': greatHugeintOp), then customer x orders x lineitem grouped by the five
 * select-list columns — c_name a VARCHAR key (hash = util.HashBytes, compare bytes). o_totalprice unscaled at scale 2. */
typedef struct { int32_t c_custkey; int64_t o_orderkey; int32_t o_orderdate; int64_t o_totalprice; ohuge sum_qty; } oracle_q18_row;
typedef struct { const int64_t *o_totalprice; } oracle_tpch_q18_extra;
int64_t oracle_q18(const oracle_tpch *T, const int64_t *o_totalprice, int64_t qty_gt, oracle_q18_row *out, int64_t max);
int64_t oracle_q18_text(oracle_q18_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap);   /* ORDER BY o_totalprice DESC, o_orderdate LIMIT */
/* Q7 / Q8 (cases/tpch/query/q7.sql, q8.sql): six- and eight-table join chains, OR of conjunctions over two nation joins, CASE,
 * EXTRACT(year), and Q8's DECIMAL division in the select list (govalues Quo, typed as its first argument) */
typedef struct { int32_t supp_nation, cust_nation, l_year; odec revenue; } oracle_q7_row;     /* nation = row of NATION */
typedef struct { int32_t o_year; odec nation_volume, volume, mkt_share; } oracle_q8_row;
int64_t oracle_q7(const oracle_tpch *T, const char *nation_a, const char *nation_b, int32_t date_ge, int32_t date_le, oracle_q7_row *out, int64_t max);
int64_t oracle_q8(const oracle_tpch *T, const char *nation, const char *region, const char *ptype, int32_t date_ge, int32_t date_le,
                  oracle_q8_row *out, int64_t max);
int64_t oracle_q7_text(oracle_q7_row *rows, int64_t n, const char *const *nation_names, char *buf, int64_t cap);   /* ORDER BY the three keys */
int64_t oracle_q8_text(oracle_q8_row *rows, int64_t n, char *buf, int64_t cap);                                     /* ORDER BY o_year */
/* Q11 (cases/tpch/query/q11.sql): HAVING against an uncorrelated scalar subquery whose select list is FLOAT arithmetic */
typedef struct { int32_t ps_partkey; odec value; } oracle_q11_row;
int64_t oracle_q11(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int64_t *ps_supplycost,
                   const int32_t *ps_availqty, const char *nation, float fraction, oracle_q11_row *out, int64_t max);
int64_t oracle_q11_text(oracle_q11_row *rows, int64_t n, char *buf, int64_t cap);   /* ORDER BY value DESC */
/* Q15 (cases/tpch/query/q15.sql): the suppliers whose quarter revenue equals the maximum, in s_suppkey order */
typedef struct { int32_t s_suppkey; odec total_revenue; } oracle_q15_row;
int64_t oracle_q15(const oracle_tpch *T, int32_t date_ge, int32_t date_lt, oracle_q15_row *out, int64_t max);
int64_t oracle_q15_text(const oracle_q15_row *rows, int64_t n, const int32_t *s_suppkey, int64_t n_supplier, const int32_t *addr_off, const char *addr_bytes,
                        const char *phone_bytes, char *buf, int64_t cap);
/* Q22 (cases/tpch/query/q22.sql): c_phone = 15 bytes per customer row, c_acctbal unscaled at scale 2, codes = the IN list */
typedef struct { char cntrycode[4]; ohuge numcust; odec totacctbal; } oracle_q22_row;
int64_t oracle_q22(const oracle_tpch *T, const char *c_phone, const int64_t *c_acctbal, const char *const *codes, int32_t ncodes,
                   oracle_q22_row *out, int64_t max);
int64_t oracle_q22_text(oracle_q22_row *rows, int64_t n, char *buf, int64_t cap);   /* ORDER BY cntrycode */
/* Q20 (cases/tpch/query/q20.sql): the qualifying suppliers' keys in s_name order; p_name as offsets + bytes, partsupp's three columns */
int64_t oracle_q20(const oracle_tpch *T, const int32_t *p_name_off, const char *p_name_bytes, int64_t n_ps, const int32_t *ps_partkey,
                   const int32_t *ps_suppkey, const int32_t *ps_availqty, const char *like_pattern, const char *nation, int32_t date_ge, int32_t date_lt,
                   float fraction, int32_t *out, int64_t max);
int64_t oracle_q20_text(const int32_t *keys, int64_t n, const int32_t *s_suppkey, int64_t n_supplier, const int32_t *addr_off, const char *addr_bytes, char *buf,
                        int64_t cap);
/* Q21 (cases/tpch/query/q21.sql): groups (supplier, count of its waiting lines); o_orderstatus = one raw byte per order ('F' / 'O' / 'P') */
typedef struct { int32_t s_suppkey; ohuge numwait; } oracle_q21_row;
int64_t oracle_q21(const oracle_tpch *T, const uint8_t *o_orderstatus, const char *nation, oracle_q21_row *out, int64_t max);
int64_t oracle_q21_text(oracle_q21_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap);   /* ORDER BY numwait DESC, s_name LIMIT */
/* Q17 (cases/tpch/query/q17.sql): 0 ok / 1 the sum is NULL / -1 error; avg_yearly = float32(sum) / divisor, the threshold fraction * avg in float64 */
int32_t oracle_q17(const oracle_tpch *T, const char *brand, const char *container, float fraction, float divisor, float *avg_yearly, odec *sum_out);
int64_t oracle_q17_text(float avg_yearly, int is_null, char *buf, int64_t cap);
int64_t oracle_q4_text(oracle_q4_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap);    /* ORDER BY o_orderpriority */
int64_t oracle_q5_text(oracle_q5_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap);    /* ORDER BY revenue DESC */
int64_t oracle_q12_text(oracle_q12_row *rows, int64_t n, const char *const *dict, char *buf, int64_t cap);  /* ORDER BY l_shipmode */
int64_t oracle_q14_text(float promo_revenue, int is_null, char *buf, int64_t cap);
int64_t oracle_q19_text(const odec *revenue, int is_null, char *buf, int64_t cap);

/* ---- round 4: Q2, Q10, Q13, Q16 (oqueries3.c) — the queries that read the generator's COMMENT text. VARCHAR columns come as offsets (int32[n+1]) +
 * bytes; s_phone / c_phone are 15 bytes per row; acctbal / supplycost unscaled at scale 2. */
typedef struct { int32_t brand, type, size; ohuge supplier_cnt; uint8_t cnt_null; } oracle_q16_row;     /* dictionary codes, p_size, count(distinct) */
int64_t oracle_q16(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int32_t *s_comment_off,
                   const char *s_comment_bytes, const char *brand_ne, const char *type_notlike, const int32_t *sizes, int32_t nsizes,
                   const char *comment_like, oracle_q16_row *out, int64_t max);
int64_t oracle_q16_text(oracle_q16_row *rows, int64_t n, const char *const *brand_dict, const char *const *type_dict, char *buf, int64_t cap);
typedef struct { int64_t c_count; uint8_t c_count_null; ohuge custdist; } oracle_q13_row;
int64_t oracle_q13(const oracle_tpch *T, const int32_t *o_comment_off, const char *o_comment_bytes, const char *notlike, oracle_q13_row *out, int64_t max);
int64_t oracle_q13_text(oracle_q13_row *rows, int64_t n, char *buf, int64_t cap);
typedef struct { int32_t s_row, nation, p_row; } oracle_q2_row;     /* rows of supplier / NATION / part */
int64_t oracle_q2(const oracle_tpch *T, int64_t n_ps, const int32_t *ps_partkey, const int32_t *ps_suppkey, const int64_t *ps_supplycost, int32_t size,
                  const char *type_like, const char *region, oracle_q2_row *out, int64_t max);
int64_t oracle_q2_text(oracle_q2_row *rows, int64_t n, int32_t limit, const oracle_tpch *T, const int64_t *s_acctbal, const uint8_t *p_mfgr,
                       const int32_t *addr_off, const char *addr_bytes, const char *phone_bytes, const int32_t *cmnt_off, const char *cmnt_bytes,
                       char *buf, int64_t cap);
typedef struct { int32_t c_custkey, nation_code; odec revenue; } oracle_q10_row;
int64_t oracle_q10(const oracle_tpch *T, const uint8_t *l_returnflag, const char *const *returnflag_dict, const int64_t *c_acctbal, const char *flag,
                   int32_t date_ge, int32_t date_lt, oracle_q10_row *out, int64_t max);
int64_t oracle_q10_text(oracle_q10_row *rows, int64_t n, int32_t limit, const oracle_tpch *T, const int64_t *c_acctbal, const int32_t *addr_off,
                        const char *addr_bytes, const char *phone_bytes, const int32_t *cmnt_off, const char *cmnt_bytes, char *buf, int64_t cap);

/* ---- result text: Chunk.SaveToFile (pkg/chunk/chunk.go:196-220), Vector.GetValue
 * (vector.go:76-186), Value.String (value.go:26-70), headline "#\t..." of execQuery
 * (executor_bench.go:229-238). The ORDER BY / LIMIT tail of each query is applied here with a
 * plain sort — the reference's sort operator is outside the hot path. */
int oracle_format_decimal(odec d, int type_scale, char *buf);      /* GetValue DECIMAL + String */
int oracle_format_double(double v, char *buf);                    /* Go %v of a float64 */
int oracle_format_hugeint(ohuge h, char *buf);
int oracle_format_date(int32_t days, char *buf);

int64_t oracle_q1_text(const oracle_q1_row *rows, int32_t n, const char *const *rf_dict,
                       const char *const *ls_dict, char *buf, int64_t cap);
int64_t oracle_q6_text(const odec *revenue, int is_null, char *buf, int64_t cap);
int64_t oracle_q3_text(oracle_q3_row *rows, int64_t n, int32_t limit, char *buf, int64_t cap);
int64_t oracle_q9_text(oracle_q9_row *rows, int64_t n, const char *const *nation_names,
                       char *buf, int64_t cap);

/* substring(s FROM offset FOR length): substringFunc + substringStartEnd
 * (pkg/compute/function_operator_binary.go:553-625). Writes the result bytes to out (caller gives
 * at least slen bytes) and returns their count. */
int64_t oracle_substring(const char *s, int64_t slen, int64_t offset, int64_t length, char *out);

/* CrossProductExec.Execute (pkg/compute/join_cross.go:109-230) as row-id pairs in the order the
 * reference emits rows: for every left chunk of `chunk` rows, for every right row, the chunk's left
 * rows. out_l / out_r: n_left * n_right entries. */
void oracle_cross_pairs(int64_t n_left, int64_t n_right, int64_t chunk, int64_t *out_l, int64_t *out_r);

/* LIKE with % and _ (wildcardMatch, function_operator_boolean.go) */
int oracle_like(const char *s, int64_t slen, const char *pattern);

#ifdef __cplusplus
}
#endif
#endif
