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
                 ("l_linestatus", hip.PH_CODE8, 0, tpchgen.LINESTATUS_DICT), ("l_shipdate", hip.PH_DATE, 0, None)],
    "orders": [("o_orderkey", hip.PH_I64, 0, None), ("o_custkey", hip.PH_I32, 0, None), ("o_orderdate", hip.PH_DATE, 0, None),
               ("o_shippriority", hip.PH_I32, 0, None)],
    "customer": [("c_custkey", hip.PH_I32, 0, None), ("c_nationkey", hip.PH_I32, 0, None), ("c_mktsegment", hip.PH_CODE8, 0, tpchgen.MKTSEGMENT_DICT)],
    "part": [("p_partkey", hip.PH_I32, 0, None), ("p_name", hip.PH_STR, 0, None)],
    "partsupp": [("ps_partkey", hip.PH_I32, 0, None), ("ps_suppkey", hip.PH_I32, 0, None), ("ps_supplycost", hip.PH_DEC64, 2, None)],
    "supplier": [("s_suppkey", hip.PH_I32, 0, None), ("s_nationkey", hip.PH_I32, 0, None)],
    "nation": [("n_nationkey", hip.PH_I32, 0, None), ("n_name", hip.PH_CODE8, 0, "nation_names")],
}
# cases/tpch/query/ddl.sql: PRIMARY KEY of every table (lineitem's (l_orderkey, l_linenumber) is not loaded)
PRIMARY_KEY = {"orders": ["o_orderkey"], "customer": ["c_custkey"], "part": ["p_partkey"], "partsupp": ["ps_partkey", "ps_suppkey"],
               "supplier": ["s_suppkey"], "nation": ["n_nationkey"]}


def nation_columns():
    names = tpchgen.nation_names()
    return {"n_nationkey": np.arange(25, dtype=np.int32), "n_name": np.arange(25, dtype=np.uint8)}, names


class Database:
    """Resident tables (ph_table) of one TPC-H database. `data`: {table: {column: numpy array}} as tests/tpch_data.load
    or the generator calls give it; only the tables present are loaded. Primary keys are declared to the library."""

    def __init__(self, ctx, data):
        self.ctx, self.tables, self.index = ctx, {}, {}
        for name, cols in SCHEMA.items():
            src = data.get(name)
            if name == "nation" and src is None:
                src, _ = nation_columns()
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
                    d = tpchgen.nation_names() if dic == "nation_names" else dic
                    specs.append(dict(typ=typ, arr=src[cname], scale=scale, dictionary=d))
                    n = len(src[cname])
                idx[cname] = len(specs) - 1
            t = hip.Table(ctx, specs, n)
            pk = PRIMARY_KEY.get(name)
            if pk and all(c in idx for c in pk):
                hip.table_declare_unique(t, [idx[c] for c in pk])
            self.tables[name], self.index[name] = t, idx

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
