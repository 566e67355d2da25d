"""The clean-room TPC-H generator against the publicly known first rows of `dbgen -s 1` output (part.tbl,
orders.tbl, lineitem.tbl as printed in countless tutorials) for the columns added in round 3. The goldens of the
queries that read them (tests/test_golden_tpch.py) are the final proof; these rows localise a regression."""
from plan_amd import tpchgen as g

SF1 = (1, 1)


def test_part_rows():
    P = g.part(SF1, 0, 5)
    td, cd, bd = g.part_type_dict(), g.part_container_dict(), g.part_brand_dict()
    got = [(int(P["p_partkey"][i]), bd[P["p_brand"][i]], td[P["p_type"][i]], int(P["p_size"][i]), cd[P["p_container"][i]]) for i in range(5)]
    assert got == [(1, "Brand#13", "PROMO BURNISHED COPPER", 7, "JUMBO PKG"), (2, "Brand#13", "LARGE BRUSHED BRASS", 1, "LG CASE"),
                   (3, "Brand#42", "STANDARD POLISHED BRASS", 21, "WRAP CASE"), (4, "Brand#34", "SMALL PLATED BRASS", 14, "MED DRUM"),
                   (5, "Brand#32", "STANDARD POLISHED TIN", 15, "SM PKG")]
    assert len(set(td)) == 150 and len(set(cd)) == 40 and len(set(bd)) == 25


def test_orders_priorities():
    O = g.orders(SF1, 0, 8, columns=["o_orderkey", "o_orderpriority"])
    got = [(int(k), g.ORDERPRIORITY_DICT[c]) for k, c in zip(O["o_orderkey"], O["o_orderpriority"])]
    assert got == [(1, "5-LOW"), (2, "1-URGENT"), (3, "5-LOW"), (4, "5-LOW"), (5, "5-LOW"), (6, "4-NOT SPECIFIED"), (7, "2-HIGH"), (32, "2-HIGH")]


def test_lineitem_ship_instructions_and_modes():
    L = g.lineitem(SF1, 0, 3, columns=["l_orderkey", "l_shipinstruct", "l_shipmode"])
    got = [(int(k), g.SHIPINSTRUCT_DICT[a], g.SHIPMODE_DICT[b]) for k, a, b in zip(L["l_orderkey"], L["l_shipinstruct"], L["l_shipmode"])]
    assert got == [(1, "DELIVER IN PERSON", "TRUCK"), (1, "TAKE BACK RETURN", "MAIL"), (1, "TAKE BACK RETURN", "REG AIR"), (1, "NONE", "AIR"),
                   (1, "NONE", "FOB"), (1, "DELIVER IN PERSON", "MAIL"), (2, "TAKE BACK RETURN", "RAIL"), (3, "NONE", "AIR"),
                   (3, "TAKE BACK RETURN", "RAIL"), (3, "DELIVER IN PERSON", "SHIP"), (3, "NONE", "TRUCK"), (3, "TAKE BACK RETURN", "FOB"),
                   (3, "TAKE BACK RETURN", "RAIL")]


def test_nation_region_table():
    names, regions, rn = g.nation_names(), g.nation_regions(), g.region_names()
    america = sorted(n for n, r in zip(names, regions) if rn[r] == "AMERICA")
    assert america == ["ARGENTINA", "BRAZIL", "CANADA", "PERU", "UNITED STATES"]     # the five rows of cases/tpch/1g/plan/q5.txt
    assert [regions.count(r) for r in range(5)] == [5, 5, 5, 5, 5]
