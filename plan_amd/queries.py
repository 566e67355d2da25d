"""TPC-H plan fragments of the hot path, expressed as the descriptors the C-ABI takes.

Each builder mirrors what the reference's planner hands `buildOperatorExec` for that query's
Agg <- Scan(filter) / join sub-tree (pkg/compute/executor.go:305-350): pushed-down conjuncts,
group-by columns, aggregate argument expressions. Column indices refer to the tables made by
`lineitem_table` etc. below.
"""
import numpy as np

from . import hip, tpchgen

# lineitem column positions in the resident table
L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE, \
    L_ORDERKEY, L_PARTKEY, L_SUPPKEY = range(10)


def lineitem_table(ctx, L, keys=False):
    """L: dict of numpy columns (tpchgen.lineitem). keys=True also loads the join key columns."""
    n = len(L["l_shipdate"])
    cols = [
        (hip.PH_I32, L["l_quantity"]),
        (hip.PH_DEC64, L["l_extendedprice"], 2),
        (hip.PH_DEC64, L["l_discount"], 2),
        (hip.PH_DEC64, L["l_tax"], 2),
        (hip.PH_CODE8, L["l_returnflag"], 0, None, tpchgen.RETURNFLAG_DICT),
        (hip.PH_CODE8, L["l_linestatus"], 0, None, tpchgen.LINESTATUS_DICT),
        (hip.PH_DATE, L["l_shipdate"]),
    ]
    if keys:
        cols += [(hip.PH_I64, L["l_orderkey"]), (hip.PH_I32, L["l_partkey"]),
                 (hip.PH_I32, L["l_suppkey"])]
    return hip.Table(ctx, cols, n)


def q1_shipdate_cutoff():
    # l_shipdate <= date '1998-12-01' - interval '112 day'   (cases/tpch/query/q1.sql)
    return tpchgen.days(1998, 12, 1) - 112


def q1_plan(ctx, table, cutoff=None):
    """Q1: 8 aggregates grouped by (l_returnflag, l_linestatus), filter l_shipdate <= cutoff."""
    if cutoff is None:
        cutoff = q1_shipdate_cutoff()
    e, d, t = hip.X_COL(L_EXTENDEDPRICE), hip.X_COL(L_DISCOUNT), hip.X_COL(L_TAX)
    one = hip.X_CONST(1, 0)  # INTEGER literal -> DECIMAL scale 0 (function_cast.go:337-347)
    disc_price = [e, one, d, hip.X_SUB, hip.X_MUL]
    charge = disc_price + [one, t, hip.X_ADD, hip.X_MUL]
    aggs = [
        hip.aggexpr(hip.PH_A_SUM, [hip.X_COL(L_QUANTITY)]),
        hip.aggexpr(hip.PH_A_SUM, [e]),
        hip.aggexpr(hip.PH_A_SUM, disc_price),
        hip.aggexpr(hip.PH_A_SUM, charge),
        hip.aggexpr(hip.PH_A_AVG, [hip.X_COL(L_QUANTITY)]),
        hip.aggexpr(hip.PH_A_AVG, [e]),
        hip.aggexpr(hip.PH_A_AVG, [d]),
        hip.aggexpr(hip.PH_A_COUNT_STAR),
    ]
    preds = [hip.pred(L_SHIPDATE, hip.PH_LE, hip.const(hip.PH_DATE, i=cutoff))]
    return hip.ScanPlan(ctx, table, preds, [L_RETURNFLAG, L_LINESTATUS], aggs)


def q6_constants():
    """The literals as the reference's binder/constant folder produces them: decimal-point
    literals are FLOAT (float32) and `0.03 - 0.01` folds in float32 (builder_binder.go:264-273,
    rule_constant_folding.go:34-70)."""
    lo = np.float32(0.03) - np.float32(0.01)
    hi = np.float32(0.03) + np.float32(0.01)
    return tpchgen.days(1994, 1, 1), tpchgen.days(1995, 1, 1), float(lo), float(hi), 24


def q6_plan(ctx, table, consts=None):
    d1, d2, lo, hi, qty = consts or q6_constants()
    preds = [
        hip.pred(L_SHIPDATE, hip.PH_GE, hip.const(hip.PH_DATE, i=d1)),
        hip.pred(L_SHIPDATE, hip.PH_LT, hip.const(hip.PH_DATE, i=d2)),
        hip.pred(L_DISCOUNT, hip.PH_GE, hip.const(hip.PH_F32, f=lo)),
        hip.pred(L_DISCOUNT, hip.PH_LE, hip.const(hip.PH_F32, f=hi)),
        hip.pred(L_QUANTITY, hip.PH_LT, hip.const(hip.PH_I32, i=qty)),
    ]
    aggs = [hip.aggexpr(hip.PH_A_SUM, [hip.X_COL(L_EXTENDEDPRICE), hip.X_COL(L_DISCOUNT), hip.X_MUL])]
    return hip.ScanPlan(ctx, table, preds, [], aggs)
