"""Generates the TPC-H tables the hot path reads, as numpy columns (test/bench helper)."""
from plan_amd import tpchgen


def load(num, den, q9=True, text=False):
    """text=True adds the COMMENT / address columns Q2, Q10, Q13 and Q16 read (cut from the generator's 300 MiB text pool)"""
    sf = (num, den)
    t = {
        "sf": sf,
        "lineitem": tpchgen.lineitem(sf, columns=[
            "l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice",
            "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate",
            "l_commitdate", "l_receiptdate", "l_shipinstruct", "l_shipmode", "l_linenumber"]),
        "orders": tpchgen.orders(sf, columns=[
            "o_orderkey", "o_custkey", "o_orderdate", "o_shippriority", "o_orderpriority", "o_totalprice", "o_orderstatus"] + (["o_comment"] if text else [])),
        "customer": tpchgen.customer(sf, text=text),
    }
    if q9:
        t["part"] = tpchgen.part(sf)
        t["partsupp"] = tpchgen.partsupp(sf)
        t["supplier"] = tpchgen.supplier(sf, text=text)
    return t
