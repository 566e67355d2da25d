"""Pins the oracle (and the data generator) with the REFERENCE'S OWN fixtures: ALL 22 SF1 result
files cases/tpch/1g/plan/q{1..22}.txt (copied to tests/golden/plan_q*.txt). The oracle's
pipelines, run on include/tpchgen.h data, must reproduce them byte for byte."""
import os

import oracle_lib as O
from plan_amd import tpchgen

G = os.path.join(os.path.dirname(__file__), "golden")


def golden(name):
    return open(os.path.join(G, name)).read()


def test_q1_matches_reference_golden(sf1):
    # cases/tpch/query/q1.sql: l_shipdate <= date '1998-12-01' - interval '112 day'
    rows = O.q1(sf1["lineitem"], tpchgen.days(1998, 12, 1) - 112)
    assert len(rows) == 4
    assert O.q1_text(rows) == golden("plan_q1.txt")


def test_q6_matches_reference_golden(sf1):
    import numpy as np
    # float32 constants as the binder folds them: float32(0.03) -/+ float32(0.01)
    lo = np.float32(0.03) - np.float32(0.01)
    hi = np.float32(0.03) + np.float32(0.01)
    rc, d = O.q6(sf1["lineitem"], tpchgen.days(1994, 1, 1), tpchgen.days(1995, 1, 1), lo, hi, 24)
    assert rc == 0
    assert O.q6_text(rc, d) == golden("plan_q6.txt")


def test_q3_matches_reference_golden(sf1):
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    # 11 378 groups for this repo's q3.sql parameters (HOUSEHOLD, 1995-03-29); an independent pandas
    # merge/groupby over the same tables gives the same count (SURVEY's 11 620 is the count for
    # the TPC-H default parameters BUILDING / 1995-03-15)
    assert n == 11378
    assert O.q3_text(rows, n, 10) == golden("plan_q3.txt")


def test_q9_matches_reference_golden(sf1):
    n, rows = O.q9(sf1, "%pink%")
    assert n == 175
    assert O.q9_text(rows, n, tpchgen.nation_names()) == golden("plan_q9.txt")


def test_duckdb_goldens_agree_on_sums():
    # the DuckDB answers for the same data carry the same exact sums (text differs only in
    # trailing zeros and in AVG(decimal) being a double there)
    p = [l.split("\t") for l in golden("plan_q1.txt").split("\n")[1:] if l]
    d = [l.split("\t") for l in golden("duckdb_q1.txt").split("\n")[1:] if l]
    from decimal import Decimal
    for a, b in zip(p, d):
        assert a[:2] == b[:2]
        for i in (2, 3, 4, 5, 9):
            assert Decimal(a[i]) == Decimal(b[i])


# ---- round 3: five more of the reference's goldens. They pin what Q1/3/6/9 never reach: the SEMI join (Q4), the
# six-table join chain with a two-column join condition (Q5), IN / OR lists, column-vs-column comparisons and integer
# CASE (Q12), CASE with LIKE in the WHEN and FLOAT arithmetic over aggregates (Q14), OR of conjunctions over both join
# sides (Q19) — and the generator columns those queries read.

def test_q4_matches_reference_golden(sf1):
    # o_orderdate >= date '1997-07-01' and < date '1997-07-01' + interval '3 month'
    assert O.q4_text(sf1, tpchgen.days(1997, 7, 1), tpchgen.days(1997, 10, 1)) == golden("plan_q4.txt")


def test_q5_matches_reference_golden(sf1):
    assert O.q5_text(sf1, "AMERICA", tpchgen.days(1994, 1, 1), tpchgen.days(1995, 1, 1)) == golden("plan_q5.txt")


def test_q12_matches_reference_golden(sf1):
    assert O.q12_text(sf1, "FOB", "TRUCK", tpchgen.days(1996, 1, 1), tpchgen.days(1997, 1, 1)) == golden("plan_q12.txt")


def test_q14_matches_reference_golden(sf1):
    # l_shipdate >= date '1996-04-01' and < + interval '1 month'; the result is a FLOAT (float32 arithmetic, printed as %v)
    assert O.q14_text(sf1, "PROMO%", tpchgen.days(1996, 4, 1), tpchgen.days(1996, 5, 1)) == golden("plan_q14.txt")


def test_q15_matches_reference_golden(sf1):
    # a CTE used twice: Agg(l_suppkey; sum) and max() over its groups, DECIMAL equality as a join condition, the supplier's generated
    # s_address / s_phone in the select list (pins the generator's v-string and phone streams at supplier 7895)
    assert O.q15_text(sf1, tpchgen.days(1995, 12, 1), tpchgen.days(1996, 3, 1)) == golden("plan_q15.txt")


def test_q20_matches_reference_golden(sf1):
    # nested IN subqueries as SEMI joins, LIKE with a prefix pattern, a correlated sum by two keys joined back on both, INTEGER > FLOAT x HUGEINT in
    # float32; 177 suppliers with their generated s_address
    assert O.q20_text(sf1) == golden("plan_q20.txt")


def test_q21_matches_reference_golden(sf1):
    # EXISTS / NOT EXISTS whose join carries a non-equi condition (l2.l_suppkey <> l1.l_suppkey) beside the key, a column-vs-column filter,
    # o_orderstatus (pins the generator's derivation of it from the lines' status), ORDER BY count DESC, VARCHAR LIMIT 100
    assert O.q21_text(sf1) == golden("plan_q21.txt")


def test_q22_matches_reference_golden(sf1):
    # substring() as a filter operand and as the group key (pins oracle_substring with a reference fixture), IN over VARCHAR, avg(DECIMAL) as a
    # scalar subquery compared DECIMAL > DECIMAL, DECIMAL > FLOAT literal, NOT EXISTS as an ANTI join; pins the generator's c_phone / c_acctbal
    assert O.q22_text(sf1) == golden("plan_q22.txt")


def test_q17_matches_reference_golden(sf1):
    # a correlated subquery decorrelated into an aggregate by its key (avg(INTEGER) = float64), joined back; FLOAT literal x DOUBLE =
    # float64 arithmetic and the DOUBLE '<'; sum(DECIMAL) / 7.0 in float32
    assert O.q17_text(sf1) == golden("plan_q17.txt")


def test_q19_matches_reference_golden(sf1):
    assert O.q19_text(sf1) == golden("plan_q19.txt")


def test_q18_matches_reference_golden(sf1):
    # an aggregate with HAVING (HUGEINT '>') under a SEMI join, five group keys (c_name a VARCHAR), ORDER BY DECIMAL DESC, DATE LIMIT 100;
    # also pins the generator's o_totalprice (discount applied before tax, truncated to cents each time)
    assert O.q18_text(sf1) == golden("plan_q18.txt")


def test_q7_matches_reference_golden(sf1):
    # six tables, two nation joins, (n1 = 'FRANCE' and n2 = 'ARGENTINA') or (n1 = 'ARGENTINA' and n2 = 'FRANCE'), l_shipdate BETWEEN,
    # EXTRACT(year) as a group key, ORDER BY two VARCHAR keys and an INTEGER
    assert O.q7_text(sf1, "FRANCE", "ARGENTINA", tpchgen.days(1995, 1, 1), tpchgen.days(1996, 12, 31)) == golden("plan_q7.txt")


def test_q8_matches_reference_golden(sf1):
    # eight tables; sum(case when nation = 'ARGENTINA' then volume else 0 end) / sum(volume): DECIMAL division (govalues Quo) typed as
    # its first argument, DECIMAL(38,4), printed at scale 4
    assert O.q8_text(sf1, "ARGENTINA", "AMERICA", "ECONOMY BURNISHED TIN", tpchgen.days(1995, 1, 1), tpchgen.days(1996, 12, 31)) == golden("plan_q8.txt")


def test_q11_matches_reference_golden(sf1):
    # HAVING sum(ps_supplycost * ps_availqty) > (select sum(..) * 0.0001000000 ..): the literal is FLOAT, so the threshold and the
    # comparison are float32; 1225 rows ordered by value DESC. Also pins the generator's ps_availqty stream.
    assert O.q11_text(sf1) == golden("plan_q11.txt")


# ---- round 4: the last four goldens — the queries that read the generator's COMMENT text (the pregenerated text pool, include/tpchgen.h).

def test_q16_matches_reference_golden(sf1):
    # COUNT(DISTINCT) through the distinct side table, NOT IN as an ANTI join, `<>` / NOT LIKE / an IN list over part, and the
    # "Customer ... Complaints" injection into s_comment: 18 341 groups ordered by (supplier_cnt desc, p_brand, p_type, p_size)
    assert O.q16_text(sf1) == golden("plan_q16.txt")


def test_q13_matches_reference_golden(sf1):
    # LEFT OUTER join (NextLeftJoin), count(o_orderkey) over the NULL-extended side, CountOp's NULL-for-zero finalize as the group key of the
    # aggregate above (first row `NULL\t50005`), NOT LIKE with two '%' over o_comment
    assert O.q13_text(sf1) == golden("plan_q13.txt")
    # ... and with the specification's default substitution parameters ('special', 'requests') the publicly known qualification answer
    # of TPC-H Q13 at SF1 comes out (answers/q13.out: 0|50005, 9|6641, 10|6532, 11|6014, 8|5937, 12|5639, 13|5024, 19|4793, 7|4687,
    # 17|4587, 18|4529, 20|4516, 15|4505, 14|4446, 16|4273, 21|4190 ...): a second, independent pin of the text pool behind o_comment.
    # (cases/tpch/1g/duckdb/q13.txt was made with yet another pattern and is not reproduced by either.)
    rows = [tuple(l.split("\t")) for l in O.q13_text(sf1, notlike="%special%requests%").split("\n")[1:] if l]
    assert rows[:16] == [("NULL", "50005"), ("9", "6641"), ("10", "6532"), ("11", "6014"), ("8", "5937"), ("12", "5639"), ("13", "5024"), ("19", "4793"),
                         ("7", "4687"), ("17", "4587"), ("18", "4529"), ("20", "4516"), ("15", "4505"), ("14", "4446"), ("16", "4273"), ("21", "4190")]


def test_q2_matches_reference_golden(sf1):
    # a correlated min() decorrelated into an aggregate by its key and joined back on (key, DECIMAL value); LIKE '%TIN' over a dictionary column;
    # 100 rows of s_acctbal, s_name, n_name, p_partkey, p_mfgr, s_address, s_phone and s_comment — pins the generator's text pool and s_acctbal
    assert O.q2_text(sf1) == golden("plan_q2.txt")


def test_q10_matches_reference_golden(sf1):
    # seven group keys (four VARCHAR columns of the customer row), ORDER BY revenue DESC LIMIT 20; c_address and c_comment in the select list
    assert O.q10_text(sf1) == golden("plan_q10.txt")
