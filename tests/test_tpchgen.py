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


def test_partsupp_availqty_first_rows():
    """the first rows of dbgen's partsupp.tbl at SF1 as publicly known: 1|2|3325|771.64|, 1|2502|8076|993.49|, 1|5002|3956|337.09|,
    1|7502|4069|357.84|, 2|3|8895|378.49|, 2|2503|4969|915.27|, 2|5003|8539|438.37|, 2|7503|3025|306.39|, 3|4|4651|920.92|"""
    ps = g.partsupp(SF1, 0, 3)
    assert ps["ps_partkey"][:9].tolist() == [1, 1, 1, 1, 2, 2, 2, 2, 3]
    assert ps["ps_suppkey"][:9].tolist() == [2, 2502, 5002, 7502, 3, 2503, 5003, 7503, 4]
    assert ps["ps_availqty"][:9].tolist() == [3325, 8076, 3956, 4069, 8895, 4969, 8539, 3025, 4651]
    assert ps["ps_supplycost"][:9].tolist() == [77164, 99349, 33709, 35784, 37849, 91527, 43837, 30639, 92092]


def test_supplier_address_and_phone_first_rows():
    """the first rows of dbgen's supplier.tbl at SF1 as publicly known: 1|Supplier#000000001| N kD4on9OM Ipw3,gf0JBoQDd7tgrzrddZ|17|27-918-335-1736|,
    2|Supplier#000000002|89eJ5ksX3ImxJQBvxObC,|5|15-679-861-2259|, 3|Supplier#000000003|q1,G3Pj6OjIuUYfUoH18BFTKP5aU9bEV3|1|11-383-516-1199|;
    and the row cases/tpch/1g/plan/q15.txt prints for supplier 7895"""
    def strs(S, c):
        o, b = S[c + "_off"], S[c + "_bytes"].tobytes()
        return [b[o[i]:o[i + 1]].decode() for i in range(len(o) - 1)]
    S = g.supplier(SF1, 0, 3)
    assert strs(S, "s_name") == ["Supplier#000000001", "Supplier#000000002", "Supplier#000000003"]
    assert strs(S, "s_address") == [" N kD4on9OM Ipw3,gf0JBoQDd7tgrzrddZ", "89eJ5ksX3ImxJQBvxObC,", "q1,G3Pj6OjIuUYfUoH18BFTKP5aU9bEV3"]
    assert strs(S, "s_phone") == ["27-918-335-1736", "15-679-861-2259", "11-383-516-1199"]
    assert S["s_nationkey"].tolist() == [17, 5, 1]
    S = g.supplier(SF1, 7894, 1)
    assert (strs(S, "s_address"), strs(S, "s_phone")) == (["NYl,i8UhxTykLxGJ2voIRn20Ugk1KTzz"], ["14-559-808-3306"])


def test_customer_phone_and_acctbal_first_rows():
    """customer.tbl at SF1 as publicly known: 1|..|15|25-989-741-2988|711.56|BUILDING|, 2|..|13|23-768-687-3665|121.65|AUTOMOBILE|,
    3|..|1|11-719-748-3364|7498.12|AUTOMOBILE|, 4|..|4|14-128-190-5944|2866.83|MACHINERY|, 5|..|3|13-750-942-6364|794.47|HOUSEHOLD|"""
    C = g.customer(SF1, 0, 5)
    b = C["c_phone_bytes"].tobytes()
    assert [b[i * 15:(i + 1) * 15].decode() for i in range(5)] == ["25-989-741-2988", "23-768-687-3665", "11-719-748-3364", "14-128-190-5944", "13-750-942-6364"]
    assert C["c_acctbal"].tolist() == [71156, 12165, 749812, 286683, 79447]
    assert C["c_nationkey"].tolist() == [15, 13, 1, 4, 3]


def _strs(T, c, k=None):
    o, b = T[c + "_off"], T[c + "_bytes"].tobytes()
    return [b[o[i]:o[i + 1]].decode() for i in range(len(o) - 1 if k is None else k)]


def test_comment_columns_first_rows():
    """Round 4: the COMMENT columns — substrings of the pregenerated 300 MiB text (include/tpchgen.h). The first rows of dbgen's SF1 files as
    publicly known:
      supplier.tbl  1|..|5755.94|each slyly above the careful|   2|..|4032.68| slyly bold instructions. idle dependen|
                    3|..|4192.40|blithely silent requests after the express dependencies are sl|   4|..|4641.08|riously even requests above the exp|
                    5|..|-283.84|. slyly regular pinto bea|
      customer.tbl  1|Customer#000000001|IVhzIApeRb ot,c,E|15|..|to the even, regular platelets. regular, ironic epitaphs nag e|
                    2|Customer#000000002|XSTf4,NCwDVaWNe6tEgvwfmRchLXak|13|..|l accounts. blithely ironic theodolites integrate boldly: caref|
      orders.tbl    1|36901|O|173665.47|1996-01-02|5-LOW|Clerk#000000951|0|nstructions sleep furiously among |
                    2|78002|O|46929.18|1996-12-01|1-URGENT|Clerk#000000880|0| foxes. pending accounts at the pending, silent asymptot|
      nation.tbl    0|ALGERIA|0| haggle. carefully final deposits detect slyly agai|
                    1|ARGENTINA|1|al foxes promise slyly according to the regular accounts. bold requests alon|
      region.tbl    0|AFRICA|lar deposits. blithely final packages cajole. regular waters are final requests. regular accounts are according to |
                    1|AMERICA|hs use ironic, even requests. s|
    A 115-character comment does not come out right by chance: the grammar, the word lists and weights, the sentence stream and the columns'
    own streams are all pinned by these rows (and again by the 120 comments of the reference's q2.txt / q10.txt, tests/test_golden_tpch.py)."""
    S = g.supplier(SF1, 0, 5, text=True)
    assert _strs(S, "s_comment") == ["each slyly above the careful", " slyly bold instructions. idle dependen",
                                     "blithely silent requests after the express dependencies are sl", "riously even requests above the exp",
                                     ". slyly regular pinto bea"]
    assert S["s_acctbal"].tolist() == [575594, 403268, 419240, 464108, -28384]
    C = g.customer(SF1, 0, 2, text=True)
    assert _strs(C, "c_comment") == ["to the even, regular platelets. regular, ironic epitaphs nag e", "l accounts. blithely ironic theodolites integrate boldly: caref"]
    assert _strs(C, "c_address") == ["IVhzIApeRb ot,c,E", "XSTf4,NCwDVaWNe6tEgvwfmRchLXak"]
    O = g.orders(SF1, 0, 2, columns=["o_orderkey", "o_comment"])
    assert _strs(O, "o_comment") == ["nstructions sleep furiously among ", " foxes. pending accounts at the pending, silent asymptot"]
    assert g.nation_comments()[:2] == [" haggle. carefully final deposits detect slyly agai", "al foxes promise slyly according to the regular accounts. bold requests alon"]
    assert g.region_comments()[:2] == ["lar deposits. blithely final packages cajole. regular waters are final requests. regular accounts are according to ",
                                       "hs use ironic, even requests. s"]


def test_supplier_complaint_injection():
    """10 suppliers in 10 000 carry "Customer ... Complaints" or "... Recommends" written over their comment (TPC-H 4.2.3): Q16's
    `s_comment like '%Customer%Complaints%'` selects the first kind. s_complaint flags them without the text pool; with the text the two agree."""
    S = g.supplier(SF1, text=True)
    com = _strs(S, "s_comment")
    flagged = [i for i, c in enumerate(com) if "Customer" in c]
    assert 2 <= len(flagged) <= 30
    complaints = [i for i in flagged if "Complaints" in com[i][com[i].index("Customer"):]]
    assert complaints == [int(i) for i in S["s_complaint"].nonzero()[0]] and all("Customer " in com[i] for i in flagged)
    assert all(("Complaints" in com[i]) != ("Recommends" in com[i]) for i in flagged)
    # generated from any row on: shards equal the whole
    T = g.supplier(SF1, 5000, 100, text=True)
    assert _strs(T, "s_comment") == com[5000:5100]


def test_part_manufacturer_follows_the_brand():
    P = g.part(SF1, 0, 1000)
    bd = g.part_brand_dict()
    assert all(g.MFGR_DICT[m][-1] == bd[b][6] for m, b in zip(P["p_mfgr"], P["p_brand"]))
