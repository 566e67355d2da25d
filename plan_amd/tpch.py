"""TPC-H as resident tables + the operator subtrees the reference's planner hands buildOperatorExec
(pkg/compute/executor.go:305-350), written as ph_plan descriptors (include/planhip.h, "resident plans").

The reference plans these queries in Go (parser -> binder -> optimizer, out of scope here: SURVEY.md §2 rows
12-13); what reaches the executors is a PhysicalOperator tree of Scan / Filter / Join / Project / Agg nodes with
pushed-down conjuncts, join keys and pruned output columns. The builders below write that tree down by hand for
each query — the join ORDER is the planner's business and is fixed here; every PHYSICAL choice (table forms,
lookups vs pairs, fused filters, merge / streaming forms) is made inside the library from the tables' statistics.
Tests, bench.py and the C++ host layer (plan_amd/csrc/host/tpch_plans.cpp mirrors these builders) share the shapes.
"""
import numpy as np

from . import hip, tpchgen

# table -> [(column, ph type, scale, dictionary or None)]; the order is the resident table's column order
SCHEMA = {
    "lineitem": [("l_orderkey", hip.PH_I64, 0, None), ("l_partkey", hip.PH_I32, 0, None), ("l_suppkey", hip.PH_I32, 0, None),
                 ("l_quantity", hip.PH_I32, 0, None), ("l_extendedprice", hip.PH_DEC64, 2, None), ("l_discount", hip.PH_DEC64, 2, None),
                 ("l_tax", hip.PH_DEC64, 2, None), ("l_returnflag", hip.PH_CODE8, 0, tpchgen.RETURNFLAG_DICT),
                 ("l_linestatus", hip.PH_CODE8, 0, tpchgen.LINESTATUS_DICT), ("l_shipdate", hip.PH_DATE, 0, None),
                 ("l_commitdate", hip.PH_DATE, 0, None), ("l_receiptdate", hip.PH_DATE, 0, None),
                 ("l_shipmode", hip.PH_CODE8, 0, tpchgen.SHIPMODE_DICT), ("l_shipinstruct", hip.PH_CODE8, 0, tpchgen.SHIPINSTRUCT_DICT),
                 ("l_linenumber", hip.PH_I32, 0, None)],
    "orders": [("o_orderkey", hip.PH_I64, 0, None), ("o_custkey", hip.PH_I32, 0, None), ("o_orderdate", hip.PH_DATE, 0, None),
               ("o_shippriority", hip.PH_I32, 0, None), ("o_orderpriority", hip.PH_CODE8, 0, tpchgen.ORDERPRIORITY_DICT),
               ("o_totalprice", hip.PH_DEC64, 2, None), ("o_orderstatus", hip.PH_CODE8, 0, tpchgen.ORDERSTATUS_DICT), ("o_comment", hip.PH_STR, 0, None)],
    "customer": [("c_custkey", hip.PH_I32, 0, None), ("c_nationkey", hip.PH_I32, 0, None), ("c_mktsegment", hip.PH_CODE8, 0, tpchgen.MKTSEGMENT_DICT),
                 ("c_name", hip.PH_STR, 0, None), ("c_phone", hip.PH_STR, 0, None), ("c_acctbal", hip.PH_DEC64, 2, None),
                 ("c_address", hip.PH_STR, 0, None), ("c_comment", hip.PH_STR, 0, None)],
    "part": [("p_partkey", hip.PH_I32, 0, None), ("p_name", hip.PH_STR, 0, None), ("p_brand", hip.PH_CODE8, 0, "part_brand"),
             ("p_type", hip.PH_CODE8, 0, "part_type"), ("p_size", hip.PH_I32, 0, None), ("p_container", hip.PH_CODE8, 0, "part_container"),
             ("p_mfgr", hip.PH_CODE8, 0, tpchgen.MFGR_DICT)],
    "partsupp": [("ps_partkey", hip.PH_I32, 0, None), ("ps_suppkey", hip.PH_I32, 0, None), ("ps_supplycost", hip.PH_DEC64, 2, None), ("ps_availqty", hip.PH_I32, 0, None)],
    "supplier": [("s_suppkey", hip.PH_I32, 0, None), ("s_nationkey", hip.PH_I32, 0, None), ("s_name", hip.PH_STR, 0, None),
                 ("s_address", hip.PH_STR, 0, None), ("s_phone", hip.PH_STR, 0, None), ("s_acctbal", hip.PH_DEC64, 2, None), ("s_comment", hip.PH_STR, 0, None)],
    "nation": [("n_nationkey", hip.PH_I32, 0, None), ("n_name", hip.PH_CODE8, 0, "nation_names"), ("n_regionkey", hip.PH_I32, 0, None)],
    "region": [("r_regionkey", hip.PH_I32, 0, None), ("r_name", hip.PH_CODE8, 0, "region_names")],
}
_NAMED_DICTS = {"nation_names": tpchgen.nation_names, "region_names": tpchgen.region_names, "part_brand": tpchgen.part_brand_dict,
                "part_type": tpchgen.part_type_dict, "part_container": tpchgen.part_container_dict}
# cases/tpch/query/ddl.sql: PRIMARY KEY of every table (lineitem's (l_orderkey, l_linenumber) is not loaded)
PRIMARY_KEY = {"orders": ["o_orderkey"], "customer": ["c_custkey"], "part": ["p_partkey"], "partsupp": ["ps_partkey", "ps_suppkey"],
               "supplier": ["s_suppkey"], "nation": ["n_nationkey"], "region": ["r_regionkey"]}


def nation_columns():
    return {"n_nationkey": np.arange(25, dtype=np.int32), "n_name": np.arange(25, dtype=np.uint8),
            "n_regionkey": np.array(tpchgen.nation_regions(), dtype=np.int32)}


def region_columns():
    return {"r_regionkey": np.arange(5, dtype=np.int32), "r_name": np.arange(5, dtype=np.uint8)}


class Database:
    """Resident tables (ph_table) of one TPC-H database. `data`: {table: {column: numpy array}} as tests/tpch_data.load
    or the generator calls give it; only the tables present are loaded. Primary keys are declared to the library."""

    def __init__(self, ctx, data):
        self.ctx, self.tables, self.index = ctx, {}, {}
        for name, cols in SCHEMA.items():
            src = data.get(name)
            if name == "nation" and src is None:
                src = nation_columns()
            if name == "region" and src is None:
                src = region_columns()
            if src is None:
                continue
            specs, idx = [], {}
            for cname, typ, scale, dic in cols:
                if typ == hip.PH_STR:
                    if cname + "_off" not in src:
                        continue
                    specs.append(dict(typ=typ, arr=src[cname + "_off"], aux=src[cname + "_bytes"]))
                    n = len(src[cname + "_off"]) - 1
                else:
                    if cname not in src:
                        continue
                    d = _NAMED_DICTS[dic]() if isinstance(dic, str) else dic
                    specs.append(dict(typ=typ, arr=src[cname], scale=scale, dictionary=d))
                    n = len(src[cname])
                idx[cname] = len(specs) - 1
            t = hip.Table(ctx, specs, n)
            pk = PRIMARY_KEY.get(name)
            if pk and all(c in idx for c in pk):
                hip.table_declare_unique(t, [idx[c] for c in pk])
            self.tables[name], self.index[name] = t, idx

    def on(self, ctx):
        """the same resident tables seen from another ctx: plans built from the view are created and run on `ctx` (its own stream) while
        the tables stay where they were loaded — the arrangement of a host that keeps one table cache for every query's context"""
        import copy
        v = copy.copy(self)
        v.ctx = ctx
        return v

    def t(self, name):
        return self.tables[name]

    def c(self, table, *names):
        return [self.index[table][n] for n in names]

    def rows(self, table):
        return self.tables[table].nrows

    def free(self):
        for t in self.tables.values():
            t.free()
        self.tables = {}


def _pred(db, table, column, op, k):
    return hip.pred(db.index[table][column], op, k)


def q3_plan(db, segment="HOUSEHOLD", date=None, topk=10):
    """cases/tpch/query/q3.sql:
       Agg(l_orderkey, o_orderdate, o_shippriority; sum(l_extendedprice * (1 - l_discount)))
        <- Join(l_orderkey = o_orderkey)  probe lineitem[l_shipdate > d]
             build <- Join(o_custkey = c_custkey)  probe orders[o_orderdate < d], build customer[c_mktsegment = seg]
       and ORDER BY revenue DESC, o_orderdate LIMIT 10 above it (announced with ph_plan_set_topk)."""
    date = tpchgen.days(1995, 3, 29) if date is None else date
    p = hip.Plan(db.ctx)
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey"), [_pred(db, "customer", "c_mktsegment", hip.PH_EQ, hip.const(hip.PH_STR, s=segment))])
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"),
                    [_pred(db, "orders", "o_orderdate", hip.PH_LT, hip.const(hip.PH_DATE, i=date))])
    j1 = p.join(orders, cust, [1], [0], [0, 2, 3])                       # -> o_orderkey, o_orderdate, o_shippriority
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipdate", hip.PH_GT, hip.const(hip.PH_DATE, i=date))])
    j2 = p.join(line, j1, [0], [0], [0, 1, 2, 4, 5])                     # -> l_orderkey, ext, disc, o_orderdate, o_shippriority
    revenue = hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL])
    p.agg(j2, [hip.pe_col(0), hip.pe_col(3), hip.pe_col(4)], [(hip.PH_A_SUM, revenue)])
    p.create()
    if topk:
        p.set_topk(0, topk, descending=True)
    return p


def q9_plan(db, pattern="%pink%"):
    """cases/tpch/query/q9.sql: part[p_name like pattern] and the five joins hanging off lineitem, the profit
    expression, group by (nation, year):
       Agg(n_name, o_year; sum(amount))
        <- Project(n_name, extract(year from o_orderdate), l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity)
        <- Join(s_nationkey = n_nationkey) <- Join(l_orderkey = o_orderkey) <- Join(l_suppkey = s_suppkey)
        <- Join((l_partkey, l_suppkey) = (ps_partkey, ps_suppkey)) <- Join(l_partkey = p_partkey) <- lineitem"""
    p = hip.Plan(db.ctx)
    part = p.scan(db.t("part"), db.c("part", "p_partkey"), [_pred(db, "part", "p_name", hip.PH_LIKE, hip.const(hip.PH_STR, s=pattern))])
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"))
    j1 = p.join(line, part, [1], [0], [0, 1, 2, 3, 4, 5])
    ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey", "ps_supplycost"))
    j2 = p.join(j1, ps, [1, 2], [0, 1], [0, 2, 3, 4, 5, 8])            # l_orderkey, l_suppkey, qty, ext, disc, ps_supplycost
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
    j3 = p.join(j2, supp, [1], [0], [0, 2, 3, 4, 5, 7])                  # l_orderkey, qty, ext, disc, cost, s_nationkey
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_orderdate"))
    j4 = p.join(j3, orders, [0], [0], [1, 2, 3, 4, 5, 7])                # qty, ext, disc, cost, s_nationkey, o_orderdate
    nation = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name"))
    j5 = p.join(j4, nation, [4], [0], [0, 1, 2, 3, 5, 7])                # qty, ext, disc, cost, o_orderdate, n_name
    amount = hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL, hip.X_COL(3), hip.X_COL(0), hip.X_MUL, hip.X_SUB])
    proj = p.project(j5, [hip.pe_col(5), hip.pe_year(4), amount])
    p.agg(proj, [hip.pe_col(0), hip.pe_col(1)], [(hip.PH_A_SUM, hip.pe_col(2))])
    return p.create()


def q1_plan(db, cutoff=None):
    """Agg <- Scan: ph_plan hands it to the fused scan kernels (the ph_scan_plan path)"""
    from . import queries
    cutoff = queries.q1_shipdate_cutoff() if cutoff is None else cutoff
    p = hip.Plan(db.ctx)
    s = p.scan(db.t("lineitem"), db.c("lineitem", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"),
               [_pred(db, "lineitem", "l_shipdate", hip.PH_LE, hip.const(hip.PH_DATE, i=cutoff))])
    one = hip.X_CONST(1)
    dp = [hip.X_COL(1), one, hip.X_COL(2), hip.X_SUB, hip.X_MUL]
    ch = dp + [one, hip.X_COL(3), hip.X_ADD, hip.X_MUL]
    p.agg(s, [hip.pe_col(4), hip.pe_col(5)],
          [(hip.PH_A_SUM, hip.pe_col(0)), (hip.PH_A_SUM, hip.pe_col(1)), (hip.PH_A_SUM, hip.pe_dec(dp)), (hip.PH_A_SUM, hip.pe_dec(ch)),
           (hip.PH_A_AVG, hip.pe_col(0)), (hip.PH_A_AVG, hip.pe_col(1)), (hip.PH_A_AVG, hip.pe_col(2)), (hip.PH_A_COUNT_STAR, None)])
    return p.create()


def q3_top(r, limit=10):
    """ORDER BY revenue DESC, o_orderdate LIMIT k over the (preselected) group rows -> (okey, revenue, odate, prio)"""
    rows = [(int(r["keys"][g][0]), r["sum"][g][0], int(r["keys"][g][1]), int(r["keys"][g][2])) for g in range(r["ngroups"])]
    return sorted(rows, key=lambda x: (-x[1], x[2]))[:limit]


def q9_rows(r):
    """(nation code, year, sum_profit unscaled at scale 4) per group"""
    return [(int(r["keys"][g][0]), int(r["keys"][g][1]), r["sum"][g][0]) for g in range(r["ngroups"])]


# ---------------------------------------------------------------- round 3: Q4, Q5, Q12, Q14, Q19

def _k(typ, **kw):
    return hip.const(typ, **kw)


def _s(x):
    return hip.const(hip.PH_STR, s=x)


def q4_plan(db, d1=None, d2=None):
    """cases/tpch/query/q4.sql: Agg(o_orderpriority; count(*)) <- SemiJoin(o_orderkey = l_orderkey)
       probe orders[date range], build lineitem[l_commitdate < l_receiptdate] (the EXISTS subquery)"""
    d1 = tpchgen.days(1997, 7, 1) if d1 is None else d1
    d2 = tpchgen.days(1997, 10, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    lc, lr = db.c("lineitem", "l_commitdate", "l_receiptdate")
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey"), bools=hip.bool_tree(("colcmp", lc, hip.PH_LT, lr)))
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_orderpriority"),
                    [_pred(db, "orders", "o_orderdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "orders", "o_orderdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    j = p.join(orders, line, [0], [0], [1], join_type=hip.PH_JT_SEMI)
    p.agg(j, [hip.pe_col(0)], [(hip.PH_A_COUNT_STAR, None)])
    return p.create()


def q5_plan(db, region="AMERICA", d1=None, d2=None):
    """cases/tpch/query/q5.sql: the six-table chain; the last join carries the two-column condition
       (l_suppkey, c_nationkey) = (s_suppkey, s_nationkey)"""
    d1 = tpchgen.days(1994, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1995, 1, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    reg = p.scan(db.t("region"), db.c("region", "r_regionkey"), [_pred(db, "region", "r_name", hip.PH_EQ, _s(region))])
    nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name", "n_regionkey"))
    jn = p.join(nat, reg, [2], [0], [0, 1])                               # n_nationkey, n_name
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_nationkey"))
    jc = p.join(cust, jn, [1], [0], [0, 1, 3])                            # c_custkey, c_nationkey, n_name
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey"),
                    [_pred(db, "orders", "o_orderdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "orders", "o_orderdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    jo = p.join(orders, jc, [1], [0], [0, 3, 4])                          # o_orderkey, c_nationkey, n_name
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"))
    jl = p.join(line, jo, [0], [0], [1, 2, 3, 5, 6])                      # l_suppkey, ext, disc, c_nationkey, n_name
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
    js = p.join(jl, supp, [0, 3], [0, 1], [1, 2, 4])                      # ext, disc, n_name
    p.agg(js, [hip.pe_col(2)], [(hip.PH_A_SUM, hip.pe_dec([hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]))])
    return p.create()


def q7_plan(db, a="FRANCE", b="ARGENTINA", d1=None, d2=None):
    """cases/tpch/query/q7.sql: lineitem[l_shipdate between] x supplier x orders x customer with the two nation joins, the pair
       condition (n1 = a and n2 = b) or (n1 = b and n2 = a) as a Filter over both nation names, group by (n1, n2, year(l_shipdate)).
       The planner's part here: each nation scan carries n_name IN (a, b), which the pair condition implies, so the supplier and the
       customer side are reduced before they meet lineitem."""
    d1 = tpchgen.days(1995, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1996, 12, 31) if d2 is None else d2
    p = hip.Plan(db.ctx)
    nn = db.c("nation", "n_name")[0]
    two = lambda: hip.bool_tree(("in", nn, [_s(a), _s(b)]))
    n1 = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name"), bools=two())
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
    js = p.join(supp, n1, [1], [0], [0, 3])                                # s_suppkey, n1.n_name
    n2 = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name"), bools=two())
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_nationkey"))
    jc = p.join(cust, n2, [1], [0], [0, 3])                                # c_custkey, n2.n_name
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey", "l_shipdate", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LE, _k(hip.PH_DATE, i=d2))])
    j1 = p.join(line, js, [1], [0], [0, 2, 3, 4, 6])                       # l_orderkey, l_shipdate, ext, disc, n1
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey"))
    j2 = p.join(j1, orders, [0], [0], [1, 2, 3, 4, 6])                     # l_shipdate, ext, disc, n1, o_custkey
    j3 = p.join(j2, jc, [4], [0], [0, 1, 2, 3, 6])                         # l_shipdate, ext, disc, n1, n2
    pair = hip.bool_tree(("or", ("and", ("cmp", 3, hip.PH_EQ, _s(a)), ("cmp", 4, hip.PH_EQ, _s(b))),
                                ("and", ("cmp", 3, hip.PH_EQ, _s(b)), ("cmp", 4, hip.PH_EQ, _s(a)))))
    f = p.filter(j3, bools=pair)
    proj = p.project(f, [hip.pe_col(3), hip.pe_col(4), hip.pe_year(0), hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL])])
    p.agg(proj, [hip.pe_col(0), hip.pe_col(1), hip.pe_col(2)], [(hip.PH_A_SUM, hip.pe_col(3))])
    return p.create()


def q8_plan(db, nation="ARGENTINA", region="AMERICA", ptype="ECONOMY BURNISHED TIN", d1=None, d2=None):
    """cases/tpch/query/q8.sql: eight tables — part[p_type] x lineitem x orders[date range] x customer x nation n1 x region[r_name] for the
       customer side, supplier x nation n2 for the CASE: Agg(year(o_orderdate); sum(case when n2 = nation then volume else 0 end), sum(volume));
       the select list divides the two sums on the host (DECIMAL `/`)"""
    d1 = tpchgen.days(1995, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1996, 12, 31) if d2 is None else d2
    p = hip.Plan(db.ctx)
    reg = p.scan(db.t("region"), db.c("region", "r_regionkey"), [_pred(db, "region", "r_name", hip.PH_EQ, _s(region))])
    n1 = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_regionkey"))
    jn = p.join(n1, reg, [1], [0], [0])                                    # n_nationkey (of the region)
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_nationkey"))
    jc = p.join(cust, jn, [1], [0], [0])                                   # c_custkey
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey", "o_orderdate"),
                    [_pred(db, "orders", "o_orderdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "orders", "o_orderdate", hip.PH_LE, _k(hip.PH_DATE, i=d2))])
    jo = p.join(orders, jc, [1], [0], [0, 2])                              # o_orderkey, o_orderdate
    part = p.scan(db.t("part"), db.c("part", "p_partkey"), [_pred(db, "part", "p_type", hip.PH_EQ, _s(ptype))])
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_partkey", "l_suppkey", "l_extendedprice", "l_discount"))
    j1 = p.join(line, part, [1], [0], [0, 2, 3, 4])                        # l_orderkey, l_suppkey, ext, disc
    j2 = p.join(j1, jo, [0], [0], [1, 2, 3, 5])                            # l_suppkey, ext, disc, o_orderdate
    n2 = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name"))
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
    js = p.join(supp, n2, [1], [0], [0, 3])                                # s_suppkey, n2.n_name
    j3 = p.join(j2, js, [0], [0], [1, 2, 3, 5])                            # ext, disc, o_orderdate, n2.n_name
    dp = [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]
    mine = hip.pe_case(hip.bool_tree(("cmp", 3, hip.PH_EQ, _s(nation))), dp, [hip.X_CONST(0)], keep=p._keep)
    p.agg(j3, [hip.pe_year(2)], [(hip.PH_A_SUM, mine), (hip.PH_A_SUM, hip.pe_dec(dp))])
    return p.create()


def q11_plans(db, nation="JAPAN"):
    """cases/tpch/query/q11.sql: (grouped plan, scalar plan) — Agg(ps_partkey; sum(ps_supplycost * ps_availqty)) and the same sum ungrouped over
       partsupp x supplier x nation[n_name]; HAVING sum > scalar * 0.0001 is applied by the caller in float32 (q11_rows)"""
    def chain(p):
        nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey"), [_pred(db, "nation", "n_name", hip.PH_EQ, _s(nation))])
        supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
        js = p.join(supp, nat, [1], [0], [0])
        ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey", "ps_supplycost", "ps_availqty"))
        return p.join(ps, js, [1], [0], [0, 2, 3])                         # ps_partkey, ps_supplycost, ps_availqty
    value = lambda: hip.pe_dec([hip.X_COL(1), hip.X_COL(2), hip.X_MUL])
    g = hip.Plan(db.ctx)
    g.agg(chain(g), [hip.pe_col(0)], [(hip.PH_A_SUM, value())])
    t = hip.Plan(db.ctx)
    t.agg(chain(t), [], [(hip.PH_A_SUM, value())])
    return g.create(), t.create()


def q11_rows(grouped, total, fraction=0.0001):
    """the groups whose DECIMAL sum, as float32, exceeds float32(total) * float32(fraction): (ps_partkey, unscaled value at scale 2)"""
    from decimal import Decimal
    f32 = lambda unscaled, scale: np.float32(float(Decimal(unscaled).scaleb(-scale)))
    if total["ngroups"] == 0:
        return []
    thr = np.float32(f32(total["sum"][0][0], total["scale"][0]) * np.float32(fraction))
    return [(int(grouped["keys"][g][0]), grouped["sum"][g][0]) for g in range(grouped["ngroups"]) if f32(grouped["sum"][g][0], grouped["scale"][0]) > thr]


def q11_set_having(grouped_plan, total, fraction=0.0001):
    """the same HAVING on the device: the threshold float32(total) * float32(fraction) becomes a FLOAT constant of ph_plan_set_having, the
       DECIMAL sums compare with it in float32 where the groups are, and only the survivors are fetched. Returns False when the total is NULL"""
    from decimal import Decimal
    if total["ngroups"] == 0:
        return False
    thr = np.float32(np.float32(float(Decimal(total["sum"][0][0]).scaleb(-total["scale"][0]))) * np.float32(fraction))
    grouped_plan.set_having([hip.pred(1, hip.PH_GT, hip.const(hip.PH_F32, f=float(thr)))])
    return True


def q12_plan(db, modes=("FOB", "TRUCK"), d1=None, d2=None):
    """cases/tpch/query/q12.sql: integer CASE sums over lineitem[shipmode IN, two column-vs-column date comparisons, receipt range]
       joined with orders"""
    d1 = tpchgen.days(1996, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1997, 1, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    sm, lc, lr, ls = db.c("lineitem", "l_shipmode", "l_commitdate", "l_receiptdate", "l_shipdate")
    where = hip.bool_tree(("and", ("in", sm, [_s(m) for m in modes]), ("colcmp", lc, hip.PH_LT, lr), ("colcmp", ls, hip.PH_LT, lc)))
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_shipmode"),
                  [_pred(db, "lineitem", "l_receiptdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_receiptdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))],
                  bools=where)
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_orderpriority"))
    j = p.join(line, orders, [0], [0], [1, 3])                            # l_shipmode, o_orderpriority
    one, zero = [hip.X_CONST(1)], [hip.X_CONST(0)]
    high = hip.pe_case(hip.bool_tree(("or", ("cmp", 1, hip.PH_EQ, _s("1-URGENT")), ("cmp", 1, hip.PH_EQ, _s("2-HIGH")))), one, zero, result_int=True, keep=p._keep)
    low = hip.pe_case(hip.bool_tree(("and", ("cmp", 1, hip.PH_NE, _s("1-URGENT")), ("cmp", 1, hip.PH_NE, _s("2-HIGH")))), one, zero, result_int=True, keep=p._keep)
    p.agg(j, [hip.pe_col(0)], [(hip.PH_A_SUM, high), (hip.PH_A_SUM, low)])
    return p.create()


def q14_plan(db, pattern="PROMO%", d1=None, d2=None):
    """cases/tpch/query/q14.sql: sum(case when p_type like 'PROMO%' then e*(1-d) else 0 end), sum(e*(1-d)); the select list's
       100.00 * a / b is FLOAT arithmetic over the one result row (q14_promo_revenue)"""
    d1 = tpchgen.days(1996, 4, 1) if d1 is None else d1
    d2 = tpchgen.days(1996, 5, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    part = p.scan(db.t("part"), db.c("part", "p_partkey", "p_type"))
    j = p.join(line, part, [0], [0], [1, 2, 4])                           # ext, disc, p_type
    dp = [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]
    promo = hip.pe_case(hip.bool_tree(("cmp", 2, hip.PH_LIKE, _s(pattern))), dp, [hip.X_CONST(0)], keep=p._keep)
    p.agg(j, [], [(hip.PH_A_SUM, promo), (hip.PH_A_SUM, hip.pe_dec(dp))])
    return p.create()


def q14_promo_revenue(r):
    """100.00 * a / b as the binder types it: the literal is FLOAT, so both sums are cast decimal -> float64 -> float32
    (tryCastDecimalToFloat32) and `*`, `/` are the FLOAT overloads (function_scalar.go:476-512, 960-1010)"""
    if r["ngroups"] == 0:
        return None
    from decimal import Decimal
    a = np.float32(float(Decimal(r["sum"][0][0]).scaleb(-r["scale"][0])))
    b = np.float32(float(Decimal(r["sum"][0][1]).scaleb(-r["scale"][1])))
    return np.float32(np.float32(100.0) * a) / b


def q15_plan(db, d1=None, d2=None):
    """cases/tpch/query/q15.sql: the CTE q15_revenue0 = Agg(l_suppkey; sum(l_extendedprice * (1 - l_discount))) over one quarter of lineitem.
       What stands above it reads only its group rows (10 k x SF): the maximum, the DECIMAL equality and the supplier's columns (q15_rows)"""
    d1 = tpchgen.days(1995, 12, 1) if d1 is None else d1
    d2 = tpchgen.days(1996, 3, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_suppkey", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    p.agg(line, [hip.pe_col(0)], [(hip.PH_A_SUM, hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL]))])
    return p.create()


def q15_rows_plan(db, d1=None, d2=None):
    """Q15 whole, as the reference's tree, in ONE plan whose root is the final join (ph_plan_fetch_rows): Join(s_suppkey = supplier_no) probe
       Scan(supplier), build Join(total_revenue = max) probe CTE, build Agg(; max(total_revenue)) <- CTE. The CTE node has two parents (lowered
       once per run); `=` on DECIMAL runs as the hash join the reference runs it as. Rows: s_suppkey, s_name, s_address, s_phone, total_revenue"""
    d1 = tpchgen.days(1995, 12, 1) if d1 is None else d1
    d2 = tpchgen.days(1996, 3, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_suppkey", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    cte = p.agg(line, [hip.pe_col(0)], [(hip.PH_A_SUM, hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL]))])
    top = p.agg(cte, [], [(hip.PH_A_MAX, hip.pe_col(1))])
    j1 = p.join(cte, top, [1], [0], [0, 1])
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_name", "s_address", "s_phone"))
    p.join(supp, j1, [0], [0], [0, 1, 2, 3, 5])
    return p.create()


def q15_rows_text(r):
    """ORDER BY s_suppkey + the reference's text over ph_plan_fetch_rows' columns"""
    key, name, addr, phone, rev = r["columns"]
    order = np.argsort(key, kind="stable")
    return "#\t\t\t\t\n" + "".join(f"{int(key[i])}\t{name[i]}\t{addr[i]}\t{phone[i]}\t{dec_text(int(rev[i]), r['scales'][4])}\n" for i in order)


def q15_rows(r):
    """[(s_suppkey, total_revenue unscaled)] of the groups whose exact DECIMAL sum equals the maximum, ORDER BY s_suppkey"""
    if r["ngroups"] == 0:
        return []
    top = max(r["sum"][g][0] for g in range(r["ngroups"]))
    return sorted((int(r["keys"][g][0]), r["sum"][g][0]) for g in range(r["ngroups"]) if r["sum"][g][0] == top)


def q15_text(db, rows, supp_keys, scale=4):
    """the reference's text: s_suppkey, s_name, s_address, s_phone (rows of the resident supplier table), total_revenue"""
    pos = [int(np.nonzero(supp_keys == k)[0][0]) for k, _ in rows]
    cols = [hip.table_strings(db.ctx, db.t("supplier"), db.c("supplier", c)[0], pos) for c in ("s_name", "s_address", "s_phone")]
    out = ["#\t\t\t\t"]
    for i, (k, v) in enumerate(rows):
        out.append(f"{k}\t{cols[0][i]}\t{cols[1][i]}\t{cols[2][i]}\t{dec_text(v, scale)}")
    return "\n".join(out) + "\n"


def q20_plans(db, pattern="lime%", nation="VIETNAM", d1=None, d2=None):
    """cases/tpch/query/q20.sql as two plans: (1) partsupp SEMI part[p_name like ..] joined on BOTH keys with the correlated subquery's aggregate
       (Agg(l_partkey, l_suppkey; sum(l_quantity)) over one year of lineitem, its groups stay on the device), grouped by the three columns the
       FLOAT predicate reads (ps_suppkey, ps_availqty, sum) — the plan has no FLOAT arithmetic, q20_keys applies it to the groups;
       (2) the suppliers of the nation: supplier x nation[n_name = ..] by key"""
    d1 = tpchgen.days(1993, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1994, 1, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    sub_scan = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_suppkey", "l_quantity"),
                      [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    sub = p.agg(sub_scan, [hip.pe_col(0), hip.pe_col(1)], [(hip.PH_A_SUM, hip.pe_col(2))])          # l_partkey, l_suppkey, sum
    part = p.scan(db.t("part"), db.c("part", "p_partkey"), [_pred(db, "part", "p_name", hip.PH_LIKE, _s(pattern))])
    ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey", "ps_availqty"))
    j1 = p.join(ps, part, [0], [0], [0, 1, 2], join_type=hip.PH_JT_SEMI)
    j2 = p.join(j1, sub, [0, 1], [0, 1], [1, 2, 5])                                                # ps_suppkey, ps_availqty, sum
    p.agg(j2, [hip.pe_col(0), hip.pe_col(1), hip.pe_col(2)], [(hip.PH_A_COUNT_STAR, None)])
    s = hip.Plan(db.ctx)
    nat = s.scan(db.t("nation"), db.c("nation", "n_nationkey"), [_pred(db, "nation", "n_name", hip.PH_EQ, _s(nation))])
    supp = s.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey"))
    js = s.join(supp, nat, [1], [0], [0])
    s.agg(js, [hip.pe_col(0)], [(hip.PH_A_COUNT_STAR, None)])
    return p.create(), s.create()


def q20_keys(r, rs, fraction=0.5):
    """s_suppkey of the result, ascending: the groups with float32(ps_availqty) > float32(fraction) * float32(sum) (sum(INTEGER) is HUGEINT, cast
       tryCastBigintToFloat32; INTEGER > FLOAT compares in float32), SEMI-joined with the nation's suppliers"""
    f = np.float32(fraction)
    good = {int(r["keys"][g][0]) for g in range(r["ngroups"]) if np.float32(int(r["keys"][g][1])) > np.float32(f * np.float32(int(r["keys"][g][2])))}
    return sorted(int(rs["keys"][g][0]) for g in range(rs["ngroups"]) if int(rs["keys"][g][0]) in good)


def q20_text(db, keys, supp_keys):
    """s_name, s_address ORDER BY s_name (the zero-padded key: key order)"""
    pos = [int(np.nonzero(supp_keys == k)[0][0]) for k in keys]
    cols = [hip.table_strings(db.ctx, db.t("supplier"), db.c("supplier", c)[0], pos) for c in ("s_name", "s_address")]
    return "#\t\n" + "".join(f"{a}\t{b}\n" for a, b in zip(*cols))


def q21_plan(db, nation="BRAZIL", residual=True):
    """cases/tpch/query/q21.sql. EXISTS / NOT EXISTS carry a non-equi conjunct beside the key (l2.l_suppkey <> l1.l_suppkey): a SEMI / ANTI join on
       l_orderkey with that RESIDUAL condition (the library filters the key matches and marks the l1 rows that keep one).
       residual=False: the same query without residual conditions, as round 3 expressed it — the INNER join on l_orderkey emits the pairs, a Filter
       compares the two supplier columns, and the l1 rows that keep a pair, identified by lineitem's primary key (l_orderkey, l_linenumber), are an
       aggregate below the SEMI / ANTI join that closes the step (the l1 side has two parents there: lowered once per run)"""
    p = hip.Plan(db.ctx)
    late = hip.bool_tree(("colcmp", db.c("lineitem", "l_receiptdate")[0], hip.PH_GT, db.c("lineitem", "l_commitdate")[0]))

    def l1_side():
        nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey"), [_pred(db, "nation", "n_name", hip.PH_EQ, _s(nation))])
        supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey", "s_name"))
        js = p.join(supp, nat, [1], [0], [0, 2])                                                  # s_suppkey, s_name
        l1 = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey", "l_linenumber"), bools=late)
        j1 = p.join(l1, js, [1], [0], [0, 1, 2, 4])                                               # l_orderkey, l_suppkey, l_linenumber, s_name
        orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey"), [_pred(db, "orders", "o_orderstatus", hip.PH_EQ, _s("F"))])
        return p.join(j1, orders, [0], [0], [0, 1, 2, 3])

    def with_other_supplier(rows, only_late):
        other = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey"), bools=late if only_late else None)
        pairs = p.join(rows, other, [0], [0], [0, 2, 1, 5])                                       # l_orderkey, l_linenumber, l1.l_suppkey, other.l_suppkey
        differ = p.filter(pairs, bools=hip.bool_tree(("colcmp", 2, hip.PH_NE, 3)))
        return p.agg(differ, [hip.pe_col(0), hip.pe_col(1)], [(hip.PH_A_COUNT_STAR, None)])       # the l1 rows with such a line

    l1 = l1_side()
    if residual:
        other_supplier = hip.bool_tree(("colcmp", 5, hip.PH_NE, 1))                               # [l1: 0..3 | other: 4 l_orderkey, 5 l_suppkey]
        l2 = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey"))
        j3 = p.join(l1, l2, [0], [0], [0, 1, 2, 3], join_type=hip.PH_JT_SEMI, residual=other_supplier)
        l3 = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_suppkey"), bools=late)
        j4 = p.join(j3, l3, [0], [0], [3], join_type=hip.PH_JT_ANTI, residual=other_supplier)   # s_name
    else:
        e = with_other_supplier(l1, False)                                                        # (l1: two parents each, lowered once per run)
        j3 = p.join(l1, e, [0, 2], [0, 1], [0, 1, 2, 3], join_type=hip.PH_JT_SEMI)
        n = with_other_supplier(j3, True)
        j4 = p.join(j3, n, [0, 2], [0, 1], [3], join_type=hip.PH_JT_ANTI)                      # s_name
    p.agg(j4, [hip.pe_col(0)], [(hip.PH_A_COUNT_STAR, None)])
    return p.create()


def q21_text(db, p, r, limit=100):
    """ORDER BY numwait DESC, s_name LIMIT 100 + the reference's text; s_name comes back as a row of supplier.s_name"""
    typ, _sc, _t, col = hip.plan_key_info(p, 0)
    assert typ == hip.PH_STR
    names = hip.table_strings(db.ctx, db.t("supplier"), col, [int(r["keys"][g][0]) for g in range(r["ngroups"])])
    rows = sorted(((names[g], int(r["count"][g][0])) for g in range(r["ngroups"])), key=lambda x: (-x[1], x[0]))[:limit]
    return "#\t\n" + "".join(f"{a}\t{c}\n" for a, c in rows)


Q22_CODES = ("10", "11", "26", "22", "19", "20", "27")


def _q22_in(col, codes):
    return hip.bool_tree(("or",) + tuple(("cmp", col, hip.PH_EQ, _s(c)) for c in codes))


def q22_scalar_plan(db, codes=Q22_CODES):
    """cases/tpch/query/q22.sql, the scalar subquery: avg(c_acctbal) over customer[c_acctbal > 0.00 (a FLOAT literal: float32 compare),
       substring(c_phone from 1 for 2) IN (..)] as its SUM and COUNT — avg(DECIMAL) is the quotient of the two (q22_threshold)"""
    p = hip.Plan(db.ctx)
    cust = p.scan(db.t("customer"), db.c("customer", "c_phone", "c_acctbal"), [_pred(db, "customer", "c_acctbal", hip.PH_GT, _k(hip.PH_F32, f=0.00))])
    pr = p.project(cust, [hip.pe_substr(0, 1, 2), hip.pe_col(1)])
    f = p.filter(pr, bools=_q22_in(0, codes))
    p.agg(f, [], [(hip.PH_A_SUM, hip.pe_col(1)), (hip.PH_A_COUNT, hip.pe_col(1))])
    return p.create()


def q22_threshold(r):
    """c_acctbal (scale 2) > avg  <=>  unscaled c_acctbal > floor(sum / count): the comparison is DECIMAL > DECIMAL, exact (greatDecimalOp), and the
       19-digit quotient lies strictly between two cents unless it is a whole number of cents — either way the floor decides"""
    if r["ngroups"] == 0 or r["count"][0][1] == 0:
        return None
    return r["sum"][0][0] // r["count"][0][1]


def q22_plan(db, threshold, codes=Q22_CODES):
    """Agg(cntrycode; count(*), sum(c_acctbal)) <- ANTI Join(c_custkey = o_custkey) probe Filter(cntrycode IN (..)) <- Project(substring) <-
       Scan(customer, c_acctbal > avg), build Scan(orders)"""
    p = hip.Plan(db.ctx)
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_phone", "c_acctbal"),
                  [_pred(db, "customer", "c_acctbal", hip.PH_GT, _k(hip.PH_DEC64, i=threshold, scale=2))])
    pr = p.project(cust, [hip.pe_col(0), hip.pe_substr(1, 1, 2), hip.pe_col(2)])
    f = p.filter(pr, bools=_q22_in(1, codes))
    orders = p.scan(db.t("orders"), db.c("orders", "o_custkey"))
    j = p.join(f, orders, [0], [0], [1, 2], join_type=hip.PH_JT_ANTI)          # cntrycode, c_acctbal
    p.agg(j, [hip.pe_col(0)], [(hip.PH_A_COUNT_STAR, None), (hip.PH_A_SUM, hip.pe_col(1))])
    return p.create()


def q22_text(db, p, r):
    """ORDER BY cntrycode + the reference's text; the group key is a VARCHAR computed in the plan: its strings are rows of a relation the plan owns"""
    typ, _sc, tab, col = hip.plan_key_info(p, 0)
    assert typ == hip.PH_STR
    names = hip.table_strings(db.ctx, tab, col, [int(r["keys"][g][0]) for g in range(r["ngroups"])])
    rows = sorted((names[g], int(r["count"][g][0]), r["sum"][g][1]) for g in range(r["ngroups"]))
    return "#\t\t\n" + "".join(f"{c}\t{n}\t{dec_text(v, 2)}\n" for c, n, v in rows)


def q17_plan(db, brand="Brand#54", container="LG BAG"):
    """cases/tpch/query/q17.sql: the correlated avg(l_quantity) subquery is an aggregate by its correlation key BELOW a join (its groups
       stay on the device; SUM and COUNT, the float64 average is taken where the DOUBLE predicate is). The reference's
       Agg(sum(l_extendedprice)) <- Filter(l_quantity < 0.2 * avg) runs as Agg(l_quantity, sum, count; sum(l_extendedprice)): the predicate
       reads only those three columns, so filtering the groups and adding their exact DECIMAL sums (q17_avg_yearly) is the same result;
       the plan has no DOUBLE arithmetic"""
    p = hip.Plan(db.ctx)
    sub_scan = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_quantity"))
    sub = p.agg(sub_scan, [hip.pe_col(0)], [(hip.PH_A_SUM, hip.pe_col(1)), (hip.PH_A_COUNT, hip.pe_col(1))])   # l_partkey, sum, count
    part = p.scan(db.t("part"), db.c("part", "p_partkey"),
                  [_pred(db, "part", "p_brand", hip.PH_EQ, _s(brand)), _pred(db, "part", "p_container", hip.PH_EQ, _s(container))])
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_quantity", "l_extendedprice"))
    j1 = p.join(line, part, [0], [0], [0, 1, 2])
    j2 = p.join(j1, sub, [0], [0], [1, 2, 4, 5])                          # l_quantity, l_extendedprice, sum(l_quantity), count(l_quantity)
    p.agg(j2, [hip.pe_col(0), hip.pe_col(2), hip.pe_col(3)], [(hip.PH_A_SUM, hip.pe_col(1))])
    return p.create()


def q17_whole_plan(db, brand="Brand#54", container="LG BAG", fraction=0.2):
    """Q17 with its DOUBLE predicate in the plan (PH_PE_FLOAT): Agg(; sum(l_extendedprice)) <- Filter(flag = 1) <- Project(flag =
       float64(l_quantity) < float64(0.2f) * (float64(sum) / float64(count)), l_extendedprice) <- the two joins. One group comes back; the select
       list's float32 division stays with the caller (q17_avg_of_sum)"""
    p = hip.Plan(db.ctx)
    sub_scan = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_quantity"))
    sub = p.agg(sub_scan, [hip.pe_col(0)], [(hip.PH_A_SUM, hip.pe_col(1)), (hip.PH_A_COUNT, hip.pe_col(1))])
    part = p.scan(db.t("part"), db.c("part", "p_partkey"),
                  [_pred(db, "part", "p_brand", hip.PH_EQ, _s(brand)), _pred(db, "part", "p_container", hip.PH_EQ, _s(container))])
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_quantity", "l_extendedprice"))
    j1 = p.join(line, part, [0], [0], [0, 1, 2])
    j2 = p.join(j1, sub, [0], [0], [1, 2, 4, 5])                          # l_quantity, l_extendedprice, sum, count
    flag = hip.pe_float([hip.X_COL(0), hip.X_F32(fraction), hip.X_COL(2), hip.X_COL(3), hip.X_OP(hip.PH_X_DIV), hip.X_MUL, hip.X_OP(hip.PH_X_LT)], wide=True)
    pr = p.project(j2, [flag, hip.pe_col(1)])
    f = p.filter(pr, [hip.pred(0, hip.PH_EQ, hip.const(hip.PH_I32, i=1))])
    p.agg(f, [], [(hip.PH_A_SUM, hip.pe_col(1))])
    return p.create()


def q17_avg_of_sum(r, divisor=7.0):
    """float32(sum) / float32(divisor) over the one group of q17_whole_plan; (None, 0) when no row passed"""
    from decimal import Decimal
    if r["ngroups"] == 0 or r["count"][0][0] == 0:
        return None, 0
    total = r["sum"][0][0]
    return np.float32(np.float32(float(Decimal(total).scaleb(-r["scale"][0]))) / np.float32(divisor)), total


def q20_whole_plan(db, pattern="lime%", nation="VIETNAM", d1=None, d2=None, fraction=0.5):
    """Q20 as ONE plan whose root is the SEMI join (ph_plan_fetch_rows: s_name, s_address): the FLOAT predicate ps_availqty > 0.5 * sum is a
       PH_PE_FLOAT flag under a Filter, the qualifying partsupp rows are the build side of SEMI Join(s_suppkey = ps_suppkey) whose probe side is
       supplier x nation[n_name]"""
    d1 = tpchgen.days(1993, 1, 1) if d1 is None else d1
    d2 = tpchgen.days(1994, 1, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    sub_scan = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_suppkey", "l_quantity"),
                      [_pred(db, "lineitem", "l_shipdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "lineitem", "l_shipdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    sub = p.agg(sub_scan, [hip.pe_col(0), hip.pe_col(1)], [(hip.PH_A_SUM, hip.pe_col(2))])
    part = p.scan(db.t("part"), db.c("part", "p_partkey"), [_pred(db, "part", "p_name", hip.PH_LIKE, _s(pattern))])
    ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey", "ps_availqty"))
    j1 = p.join(ps, part, [0], [0], [0, 1, 2], join_type=hip.PH_JT_SEMI)
    j2 = p.join(j1, sub, [0, 1], [0, 1], [1, 2, 5])                                                # ps_suppkey, ps_availqty, sum
    flag = hip.pe_float([hip.X_COL(1), hip.X_F32(fraction), hip.X_COL(2), hip.X_MUL, hip.X_OP(hip.PH_X_GT)])
    pr = p.project(j2, [hip.pe_col(0), flag])
    good = p.filter(pr, [hip.pred(1, hip.PH_EQ, hip.const(hip.PH_I32, i=1))])
    nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey"), [_pred(db, "nation", "n_name", hip.PH_EQ, _s(nation))])
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey", "s_name", "s_address"))
    js = p.join(supp, nat, [1], [0], [0, 2, 3])                                                    # s_suppkey, s_name, s_address
    p.join(js, good, [0], [0], [1, 2], join_type=hip.PH_JT_SEMI)
    return p.create()


def q17_avg_yearly(r, fraction=0.2, divisor=7.0):
    """(float32 avg_yearly or None, exact DECIMAL sum unscaled): the groups (l_quantity, sum, count) that pass
       float64(l_quantity) < float64(float32(fraction)) * (float64(sum) / float64(count)) — avg(INTEGER) is float64, the FLOAT literal is
       cast to DOUBLE for `*`, `<` is the DOUBLE overload — their sums added exactly, then float32(sum) / float32(divisor)"""
    from decimal import Decimal
    f = float(np.float32(fraction))
    total, any_row = 0, False
    for g in range(r["ngroups"]):
        q, s, c = (int(x) for x in r["keys"][g])
        if float(q) < f * (float(s) / float(c)):
            total += r["sum"][g][0]
            any_row = True
    if not any_row:
        return None, 0
    a = np.float32(float(Decimal(total).scaleb(-r["scale"][0])))
    return np.float32(a / np.float32(divisor)), total


Q19_BRANCHES = (("Brand#23", ("SM CASE", "SM BOX", "SM PACK", "SM PKG"), 5, 15, 5),
                ("Brand#15", ("MED BAG", "MED BOX", "MED PKG", "MED PACK"), 14, 24, 10),
                ("Brand#44", ("LG CASE", "LG BOX", "LG PACK", "LG PKG"), 28, 38, 15))


def q19_plan(db):
    """cases/tpch/query/q19.sql: the conjuncts common to the three OR branches (the join condition, l_shipmode IN (..),
       l_shipinstruct = ..) are what DistributivityRule + filter push-down take out of the OR; the rest is a Filter over the join"""
    p = hip.Plan(db.ctx)
    sm = db.c("lineitem", "l_shipmode")[0]
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_partkey", "l_quantity", "l_extendedprice", "l_discount"),
                  [_pred(db, "lineitem", "l_shipinstruct", hip.PH_EQ, _s("DELIVER IN PERSON"))],
                  bools=hip.bool_tree(("in", sm, [_s("AIR"), _s("AIR REG")])))
    part = p.scan(db.t("part"), db.c("part", "p_partkey", "p_brand", "p_size", "p_container"))
    j = p.join(line, part, [0], [0], [1, 2, 3, 5, 6, 7])                  # qty, ext, disc, p_brand, p_size, p_container
    I = lambda v: _k(hip.PH_I32, i=v)
    branches = tuple(("and", ("cmp", 3, hip.PH_EQ, _s(b)), ("in", 5, [_s(c) for c in cn]), ("cmp", 0, hip.PH_GE, I(q1)), ("cmp", 0, hip.PH_LE, I(q2)),
                      ("cmp", 4, hip.PH_GE, I(1)), ("cmp", 4, hip.PH_LE, I(sz))) for b, cn, q1, q2, sz in Q19_BRANCHES)
    f = p.filter(j, bools=hip.bool_tree(("or",) + branches))
    p.agg(f, [], [(hip.PH_A_SUM, hip.pe_dec([hip.X_COL(1), hip.X_CONST(1), hip.X_COL(2), hip.X_SUB, hip.X_MUL]))])
    return p.create()


def dec_text(unscaled, scale):
    """Value.String of a DECIMAL cell at `scale` (trailing zeros of the fraction trimmed, as NewFromInt64 does)"""
    neg = "-" if unscaled < 0 else ""
    w, f = divmod(abs(int(unscaled)), 10 ** scale)
    frac = (("%0" + str(scale) + "d") % f).rstrip("0") if scale else ""
    return f"{neg}{w}" + (f".{frac}" if frac else "")


def q18_plan(db, qty_gt=314, topk=0):
    """cases/tpch/query/q18.sql: the IN (select l_orderkey from lineitem group by l_orderkey having sum(l_quantity) > k) subquery is an
       aggregate BELOW a SEMI join (its groups stay on the device), HAVING is the Filter above it; five group keys, c_name a VARCHAR that is
       no small dictionary (1.5 M distinct names at SF10): interned on the device (ph_strdict), two narrow keys packed into one key word"""
    p = hip.Plan(db.ctx)
    sub_scan = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_quantity"))
    sub = p.agg(sub_scan, [hip.pe_col(0)], [(hip.PH_A_SUM, hip.pe_col(1))])                 # l_orderkey, sum(l_quantity)
    having = p.filter(sub, [hip.pred(1, hip.PH_GT, _k(hip.PH_I32, i=qty_gt))])
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey", "o_orderdate", "o_totalprice"))
    j1 = p.join(orders, having, [0], [0], [0, 1, 2, 3], join_type=hip.PH_JT_SEMI)
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_name"))
    j2 = p.join(j1, cust, [1], [0], [0, 2, 3, 4, 5])                                        # o_orderkey, o_orderdate, o_totalprice, c_custkey, c_name
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_quantity"))
    j3 = p.join(line, j2, [0], [0], [1, 6, 5, 2, 3, 4])                                     # l_quantity, c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice
    p.agg(j3, [hip.pe_col(1), hip.pe_col(2), hip.pe_col(3), hip.pe_col(4), hip.pe_col(5)], [(hip.PH_A_SUM, hip.pe_col(0))])
    return p.create()


def q18_text(db, p, r, limit=100):
    """ORDER BY o_totalprice DESC, o_orderdate LIMIT 100 + the reference's text; c_name comes back as a row of customer.c_name"""
    import datetime
    typ, _s, _t, col = hip.plan_key_info(p, 0)
    assert typ == hip.PH_STR
    rows = [[int(x) for x in r["keys"][g]] + [r["sum"][g][0]] for g in range(r["ngroups"])]
    names = hip.table_strings(db.ctx, db.t("customer"), col, [row[0] for row in rows])
    rows = sorted(zip(names, rows), key=lambda nr: (-nr[1][4], nr[1][3]))[:limit]
    out = ["#\t\t\t\t\t"]
    for name, (_c, ck, ok, od, tp, q) in rows:
        d = datetime.date(1970, 1, 1) + datetime.timedelta(days=od)
        out.append(f"{name}\t{ck}\t{ok}\t{d.isoformat()}\t{dec_text(tp, 2)}\t{q}")
    return "\n".join(out) + "\n"


# ---------------------------------------------------------------- round 4: Q16, Q13, Q2, Q10 — the queries that read the generator's COMMENT text

Q16_SIZES = (14, 7, 21, 24, 35, 33, 2, 20)


def q16_plan(db, brand_ne="Brand#35", type_notlike="ECONOMY BURNISHED%", sizes=Q16_SIZES, comment_like="%Customer%Complaints%"):
    """cases/tpch/query/q16.sql: Agg(p_brand, p_type, p_size; count(DISTINCT ps_suppkey)) <- ANTI Join(ps_suppkey = s_suppkey) [NOT IN]
       probe Join(ps_partkey = p_partkey) probe partsupp, build part[p_brand <> .., p_type NOT LIKE .., p_size IN (..)];
       build supplier[s_comment LIKE '%Customer%Complaints%'] — PH_A_COUNT_DISTINCT: the library keeps the distinct side table"""
    p = hip.Plan(db.ctx)
    ty, sz = db.c("part", "p_type", "p_size")
    part = p.scan(db.t("part"), db.c("part", "p_partkey", "p_brand", "p_type", "p_size"), [_pred(db, "part", "p_brand", hip.PH_NE, _s(brand_ne))],
                  bools=hip.bool_tree(("and", ("cmp", ty, hip.PH_NOTLIKE, _s(type_notlike)), ("in", sz, [_k(hip.PH_I32, i=v) for v in sizes]))))
    ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey"))
    j1 = p.join(ps, part, [0], [0], [1, 3, 4, 5])                                              # ps_suppkey, p_brand, p_type, p_size
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey"), [_pred(db, "supplier", "s_comment", hip.PH_LIKE, _s(comment_like))])
    j2 = p.join(j1, supp, [0], [0], [0, 1, 2, 3], join_type=hip.PH_JT_ANTI)
    p.agg(j2, [hip.pe_col(1), hip.pe_col(2), hip.pe_col(3)], [(hip.PH_A_COUNT_DISTINCT, hip.pe_col(0))])
    return p.create()


def q16_text(r):
    """ORDER BY supplier_cnt DESC, p_brand, p_type, p_size + the reference's text"""
    bd, td = tpchgen.part_brand_dict(), tpchgen.part_type_dict()
    rows = sorted(((bd[int(r["keys"][g][0])], td[int(r["keys"][g][1])], int(r["keys"][g][2]), r["count"][g][0]) for g in range(r["ngroups"])),
                  key=lambda x: (-x[3], x[0], x[1], x[2]))
    return "#\t\t\t\n" + "".join(f"{b}\t{t}\t{s}\t{c}\n" for b, t, s, c in rows)


def q13_plan(db, notlike="%pending%accounts%"):
    """cases/tpch/query/q13.sql: Agg(c_count; count(*)) <- Agg(c_custkey; count(o_orderkey)) <- LEFT Join(c_custkey = o_custkey) probe customer,
       build orders[o_comment NOT LIKE ..]. count() over the NULL-extended side is 0 for a customer without orders and finalises to NULL
       (CountOp.Finalize): the NULL is the group key of the aggregate above (key_null in the result)"""
    p = hip.Plan(db.ctx)
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey"))
    orders = p.scan(db.t("orders"), db.c("orders", "o_custkey", "o_orderkey"), [_pred(db, "orders", "o_comment", hip.PH_NOTLIKE, _s(notlike))])
    j = p.join(cust, orders, [0], [0], [0, 2], join_type=hip.PH_JT_LEFT)                       # c_custkey, o_orderkey (NULL without a match)
    inner = p.agg(j, [hip.pe_col(0)], [(hip.PH_A_COUNT, hip.pe_col(1))])                       # c_custkey, c_count
    p.agg(inner, [hip.pe_col(1)], [(hip.PH_A_COUNT_STAR, None)])
    return p.create()


def q13_text(r):
    """ORDER BY custdist DESC, c_count DESC (NULLs first) + the reference's text"""
    kn = r.get("key_null")
    rows = [(None if kn is not None and kn[g][0] else int(r["keys"][g][0]), r["count"][g][0]) for g in range(r["ngroups"])]
    rows.sort(key=lambda x: (-x[1], 0 if x[0] is None else 1, -(x[0] or 0)))
    return "#\t\n" + "".join(f"{'NULL' if k is None else k}\t{c}\n" for k, c in rows)


def q2_plan(db, size=48, type_like="%TIN", region="MIDDLE EAST"):
    """cases/tpch/query/q2.sql as ONE plan whose root is the final join (ph_plan_fetch_rows). The correlated min(ps_supplycost) is an
       aggregate by ps_partkey over partsupp x supplier x nation x region[r_name = ..] — the same subtree the outer branch joins with part, a
       node with two parents — joined back on (ps_partkey, ps_supplycost = min): DECIMAL `=` runs as the hash join the reference runs it as.
       Rows: s_acctbal, s_name, n_name, p_partkey, p_mfgr, s_address, s_phone, s_comment"""
    p = hip.Plan(db.ctx)
    reg = p.scan(db.t("region"), db.c("region", "r_regionkey"), [_pred(db, "region", "r_name", hip.PH_EQ, _s(region))])
    nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name", "n_regionkey"))
    jn = p.join(nat, reg, [2], [0], [0, 1])                                                    # n_nationkey, n_name
    supp = p.scan(db.t("supplier"), db.c("supplier", "s_suppkey", "s_nationkey", "s_acctbal", "s_name", "s_address", "s_phone", "s_comment"))
    js = p.join(supp, jn, [1], [0], [0, 2, 3, 4, 5, 6, 8])                                     # s_suppkey, s_acctbal, s_name, s_address, s_phone, s_comment, n_name
    ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_suppkey", "ps_supplycost"))
    jps = p.join(ps, js, [1], [0], [0, 2, 4, 5, 6, 7, 8, 9])                                   # ps_partkey, ps_supplycost, s_acctbal, s_name, s_address, s_phone, s_comment, n_name
    sub = p.agg(jps, [hip.pe_col(0)], [(hip.PH_A_MIN, hip.pe_col(1))])                         # ps_partkey, min(ps_supplycost)
    ty = db.c("part", "p_type")[0]
    part = p.scan(db.t("part"), db.c("part", "p_partkey", "p_mfgr"), [_pred(db, "part", "p_size", hip.PH_EQ, _k(hip.PH_I32, i=size))],
                  bools=hip.bool_tree(("cmp", ty, hip.PH_LIKE, _s(type_like))))
    jp = p.join(jps, part, [0], [0], [0, 1, 2, 3, 4, 5, 6, 7, 9])                              # + p_mfgr
    p.join(jp, sub, [0, 1], [0, 1], [2, 3, 7, 0, 8, 4, 5, 6])
    return p.create()


def q2_text(r, limit=100):
    """ORDER BY s_acctbal DESC, n_name, s_name, p_partkey LIMIT 100 + the reference's text over ph_plan_fetch_rows' columns"""
    bal, name, nat, pk, mf, addr, phone, cmnt = r["columns"]
    nn = tpchgen.nation_names()
    rows = sorted(range(r["nrows"]), key=lambda i: (-int(bal[i]), nn[int(nat[i])], name[i], int(pk[i])))[:limit]
    return "#\t\t\t\t\t\t\t\n" + "".join(
        f"{dec_text(int(bal[i]), 2)}\t{name[i]}\t{nn[int(nat[i])]}\t{int(pk[i])}\t{tpchgen.MFGR_DICT[int(mf[i])]}\t{addr[i]}\t{phone[i]}\t{cmnt[i]}\n" for i in rows)


def q10_plan(db, flag="R", d1=None, d2=None, topk=20):
    """cases/tpch/query/q10.sql: Agg(c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment; sum(l_extendedprice * (1 - l_discount)))
       over lineitem[l_returnflag = 'R'] x orders[one quarter] x customer x nation, ORDER BY revenue DESC LIMIT 20 announced as a top-k.
       Seven group keys, four of them VARCHAR columns that are no small dictionaries: interned on the device, narrow keys (INTEGER, string
       codes, dictionary codes) packed two per key word"""
    d1 = tpchgen.days(1993, 3, 1) if d1 is None else d1
    d2 = tpchgen.days(1993, 6, 1) if d2 is None else d2
    p = hip.Plan(db.ctx)
    orders = p.scan(db.t("orders"), db.c("orders", "o_orderkey", "o_custkey"),
                    [_pred(db, "orders", "o_orderdate", hip.PH_GE, _k(hip.PH_DATE, i=d1)), _pred(db, "orders", "o_orderdate", hip.PH_LT, _k(hip.PH_DATE, i=d2))])
    line = p.scan(db.t("lineitem"), db.c("lineitem", "l_orderkey", "l_extendedprice", "l_discount"), [_pred(db, "lineitem", "l_returnflag", hip.PH_EQ, _s(flag))])
    j1 = p.join(line, orders, [0], [0], [1, 2, 4])                                             # ext, disc, o_custkey
    cust = p.scan(db.t("customer"), db.c("customer", "c_custkey", "c_name", "c_acctbal", "c_phone", "c_nationkey", "c_address", "c_comment"))
    j2 = p.join(j1, cust, [2], [0], [0, 1, 3, 4, 5, 6, 7, 8, 9])                               # ext, disc, c_custkey, c_name, c_acctbal, c_phone, c_nationkey, c_address, c_comment
    nat = p.scan(db.t("nation"), db.c("nation", "n_nationkey", "n_name"))
    j3 = p.join(j2, nat, [6], [0], [0, 1, 2, 3, 4, 5, 10, 7, 8])                               # ext, disc, c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment
    revenue = hip.pe_dec([hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL])
    p.agg(j3, [hip.pe_col(c) for c in (2, 3, 4, 5, 6, 7, 8)], [(hip.PH_A_SUM, revenue)])
    p.create()
    if topk:
        p.set_topk(0, topk, descending=True)
    return p


def q10_text(db, p, r, limit=20):
    """ORDER BY revenue DESC LIMIT 20 + the reference's text; the VARCHAR keys come back as rows of customer's columns"""
    order = sorted(range(r["ngroups"]), key=lambda g: (-r["sum"][g][0], int(r["keys"][g][0])))[:limit]
    strs = {}
    for k in (1, 3, 5, 6):
        typ, _sc, _t, col = hip.plan_key_info(p, k)
        assert typ == hip.PH_STR
        strs[k] = hip.table_strings(db.ctx, db.t("customer"), col, [int(r["keys"][g][k]) for g in order])
    nn = tpchgen.nation_names()
    out = ["#\t\t\t\t\t\t\t"]
    for i, g in enumerate(order):
        k = r["keys"][g]
        out.append(f"{int(k[0])}\t{strs[1][i]}\t{dec_text(r['sum'][g][0], 4)}\t{dec_text(int(k[2]), 2)}\t{nn[int(k[4])]}\t{strs[5][i]}\t{strs[3][i]}\t{strs[6][i]}")
    return "\n".join(out) + "\n"
