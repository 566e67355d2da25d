"""Resident plans (ph_plan_*, include/planhip.h): operator subtrees over resident tables lowered inside the library
from the tables' statistics. Parity against the reference's SF1 goldens and the oracle, through the C-ABI."""
import os

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, pipelines, tpch, tpchgen

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def db(ctx, sf1):
    d = tpch.Database(ctx, sf1)
    yield d
    d.free()


def golden(name):
    return open(os.path.join(GOLDEN, name)).read()


def test_table_statistics(db):
    """order statistics gathered at load + the declared primary keys: what the plans' choices rest on"""
    o, l, ps = db.t("orders"), db.t("lineitem"), db.t("partsupp")
    A, S, U = hip.PH_STAT_ASCENDING, hip.PH_STAT_STRICT, hip.PH_STAT_DECLARED_UNIQUE
    assert hip.table_col_stats(o, db.c("orders", "o_orderkey")[0]) == A | S | U      # primary key in storage order
    assert hip.table_col_stats(l, db.c("lineitem", "l_orderkey")[0]) == A             # clustered, not unique
    assert hip.table_col_stats(o, db.c("orders", "o_custkey")[0]) == 0
    assert hip.table_col_stats(ps, db.c("partsupp", "ps_partkey")[0]) == A            # composite key: neither column alone
    assert hip.table_col_stats(db.t("customer"), db.c("customer", "c_custkey")[0]) == A | S | U


def test_q3_plan_matches_golden_and_oracle(ctx, db, sf1):
    """Q3 as ONE descriptor: the library picks the gated fills, the semi-join marks, the fused Filter -> probe, the
    streaming aggregate and the top-k preselection itself — the top 10 equal cases/tpch/1g/plan/q3.txt, all 11 378
    groups the oracle's."""
    p = tpch.q3_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    assert pipelines.q3_text(tpch.q3_top(r)) == golden("plan_q3.txt")
    for phrase in ("gated sorted fill", "semi-join marks in one pass", "fused filter+probe", "streaming aggregate"):
        assert phrase in ex, ex
    assert "conservative" not in ex
    p.free()
    p = tpch.q3_plan(db, topk=0)
    p.run()
    r = p.fetch()
    p.free()
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    want = {(rows[i].l_orderkey, rows[i].revenue.unscaled(4), rows[i].o_orderdate, rows[i].o_shippriority) for i in range(n)}
    got = {(int(r["keys"][g][0]), r["sum"][g][0], int(r["keys"][g][1]), int(r["keys"][g][2])) for g in range(r["ngroups"])}
    assert r["ngroups"] == n == 11378 and got == want


def test_having_and_topk_are_exclusive_so_no_group_is_lost(ctx, db, sf1):
    """GROUP BY .. HAVING agg > c ORDER BY agg LIMIT k ('>' is the one comparison selectOperation has for DECIMAL): HAVING runs in the aggregate's output phase, before Order and Limit
    (executor_aggr.go:143-263). A top-k preselection taken first would hand back the best k groups, the HAVING would reject some of
    them and the groups that should have moved up would be gone. The library refuses the combination in both directions
    (PH_EUNSUPPORTED); with the HAVING on the device and the sort above it, Q3's ten smallest revenues ABOVE a floor equal the oracle's."""
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    rev = sorted(rows[i].revenue.unscaled(4) for i in range(n))
    cap = rev[4]                                     # removes the five best groups of the unrestricted (ascending) top 10
    having = [hip.pred(3, hip.PH_GT, hip.const(hip.PH_DEC64, i=cap, scale=4))]
    p = tpch.q3_plan(db, topk=0)
    p.set_having(having)
    with pytest.raises(hip.PlanHipError) as e:
        p.set_topk(0, 10, descending=False)
    assert e.value.code == hip.PH_EUNSUPPORTED
    p.run()
    r = p.fetch()
    p.free()
    want = sorted(((rows[i].revenue.unscaled(4), rows[i].o_orderdate, rows[i].l_orderkey) for i in range(n) if rows[i].revenue.unscaled(4) > cap))
    got = sorted((r["sum"][g][0], int(r["keys"][g][1]), int(r["keys"][g][0])) for g in range(r["ngroups"]))
    assert r["ngroups"] == len(want) and len(want) <= n - 5 and got[:10] == want[:10] and got == want
    # the other direction: a plan that announced its top-k refuses a HAVING
    p = tpch.q3_plan(db, topk=10)
    with pytest.raises(hip.PlanHipError) as e:
        p.set_having(having)
    assert e.value.code == hip.PH_EUNSUPPORTED
    p.free()


@pytest.mark.parametrize("segment,ymd", [("AUTOMOBILE", (1994, 1, 1)), ("HOUSEHOLD", (1998, 12, 1)), ("NOSUCHSEGMENT", (1995, 3, 15))])
def test_q3_plan_other_parameters(ctx, db, sf1, segment, ymd):
    date = tpchgen.days(*ymd)
    p = tpch.q3_plan(db, segment=segment, date=date)
    p.run()
    r = p.fetch()
    p.free()
    n, rows = O.q3(sf1, segment, date)
    assert pipelines.q3_text(tpch.q3_top(r)) == O.q3_text(rows, n)


def test_q9_plan_matches_golden(ctx, db, monkeypatch):
    """Q9: LIKE, five joins (one composite), Project, 175 groups; the library finds partsupp's rows by arithmetic (stored in runs of four
    by ps_partkey: the run lookup), uses strict N:1 lookups and the merge lookup for orders; with the run lookup switched off it reduces
    partsupp by the part keys' domain and builds the node table — the same rows"""
    for no_run, phrases in ((False, ("run lookup", "strict N:1 lookup", "merge lookup")),
                            (True, ("reduced by the probe key's domain", "strict N:1 lookup", "merge lookup"))):
        if no_run:
            monkeypatch.setenv("PH_PLAN_NO_RUN_LOOKUP", "1")
        p = tpch.q9_plan(db)
        for _ in range(2):   # a plan can be run again
            p.run()
            r = p.fetch()
            assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == golden("plan_q9.txt")
        ex = p.explain()
        for phrase in phrases:
            assert phrase in ex, ex
        assert "conservative" not in ex
        p.free()


@pytest.mark.parametrize("pattern", ["%green%", "%zzzz%", "%a%"])
def test_q9_plan_other_patterns(ctx, db, sf1, pattern):
    p = tpch.q9_plan(db, pattern=pattern)
    p.run()
    r = p.fetch()
    p.free()
    n, rows = O.q9(sf1, pattern)
    assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == O.q9_text(rows, n, tpchgen.nation_names())


def test_q1_plan_takes_the_fused_scan(ctx, db, sf1):
    """Agg <- Scan inside a ph_plan is the fused scan kernel; same numbers as the oracle"""
    from plan_amd import queries
    p = tpch.q1_plan(db)
    p.run()
    r = p.fetch()
    assert "fused scan plan" in p.explain()
    p.free()
    want = O.q1(sf1["lineitem"], queries.q1_shipdate_cutoff())
    assert r["ngroups"] == len(want) == 4
    for g, w in enumerate(want):
        assert tuple(r["keys"][g]) == (w.returnflag, w.linestatus)
        assert r["sum"][g][:4] == [w.sum_qty.value(), w.sum_base_price.unscaled(2), w.sum_disc_price.unscaled(4), w.sum_charge.unscaled(6)]
        assert r["count"][g][7] == w.count_order


def shuffled(table, rng):
    if "p_name_off" in table:
        return table   # (part stays in order: Q9 reads its VARCHAR column, offsets + bytes)
    table = {k: v for k, v in table.items() if not (k.endswith("_off") or k.endswith("_bytes"))}   # (Q3 / Q9 read no other VARCHAR column)
    n = len(next(iter(table.values())))
    perm = rng.permutation(n)
    return {k: np.ascontiguousarray(v[perm]) for k, v in table.items()}


def test_plans_on_shuffled_tables_take_the_general_forms(ctx, sf1):
    """row-shuffled tables: no column is ascending, so no sorted fill, no merge lookup, no streaming aggregate — the
    plans run their general forms from the start (optimistic about foreign keys only) and give the same results"""
    rng = np.random.default_rng(7)
    data = {k: (shuffled(v, rng) if isinstance(v, dict) else v) for k, v in sf1.items()}
    d = tpch.Database(ctx, data)
    try:
        p = tpch.q3_plan(d)
        p.run()
        r = p.fetch()
        ex = p.explain()
        p.free()
        assert pipelines.q3_text(tpch.q3_top(r)) == golden("plan_q3.txt")
        assert "hash aggregate" in ex and "sorted fill" not in ex and "streaming" not in ex, ex
        p = tpch.q9_plan(d)
        p.run()
        r = p.fetch()
        ex = p.explain()
        p.free()
        assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == golden("plan_q9.txt")
        assert "merge lookup" not in ex, ex
    finally:
        d.free()


def test_q9_plan_with_dangling_foreign_keys_reruns_conservatively(ctx, sf001):
    """lineitem rows whose supplier / order does not exist: the strict lookups raise the deferred PH_ECONSTRAINT, the
    plan runs again with counted lookups that drop those rows, and the result equals the oracle's"""
    t = {k: (dict(v) if isinstance(v, dict) else v) for k, v in sf001.items()}
    L = t["lineitem"] = {k: v.copy() for k, v in t["lineitem"].items()}
    rng = np.random.default_rng(3)
    bad = rng.choice(len(L["l_suppkey"]), 500, replace=False)
    L["l_suppkey"][bad[:250]] = 10_000_000          # no such supplier (and no such partsupp row)
    L["l_orderkey"][bad[250:]] = L["l_orderkey"][bad[250:]] + 8    # key values dbgen never uses: no such order
    d = tpch.Database(ctx, t)
    try:
        p = tpch.q9_plan(d, pattern="%a%")
        p.run()
        r = p.fetch()
        ex = p.explain()
        assert "conservative" in ex and "counted N:1 lookup" in ex, ex
        n, rows = O.q9(t, "%a%")
        assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == O.q9_text(rows, n, tpchgen.nation_names())
        p.run()                                     # later runs start conservatively
        r2 = p.fetch()
        assert tpch.q9_rows(r2) == tpch.q9_rows(r) and "optimistic" not in p.explain()
        p.free()
    finally:
        d.free()


# ---------------------------------------------------------------- round 3: five more reference goldens through ph_plan

def _by_name(rows, dic):
    return sorted(rows, key=lambda r: dic[r[0]])


def q4_text(r):
    dic = tpchgen.ORDERPRIORITY_DICT
    rows = [(int(r["keys"][g][0]), r["count"][g][0]) for g in range(r["ngroups"])]
    return "#\t\n" + "".join(f"{dic[c]}\t{n}\n" for c, n in _by_name(rows, dic))


def q5_text(r):
    dic = tpchgen.nation_names()
    rows = sorted(((int(r["keys"][g][0]), r["sum"][g][0]) for g in range(r["ngroups"])), key=lambda x: -x[1])
    return "#\t\n" + "".join(f"{dic[c]}\t{tpch.dec_text(s, 4)}\n" for c, s in rows)


def q12_text(r):
    dic = tpchgen.SHIPMODE_DICT
    rows = [(int(r["keys"][g][0]), r["sum"][g][0], r["sum"][g][1]) for g in range(r["ngroups"])]
    return "#\t\t\n" + "".join(f"{dic[c]}\t{h}\t{l}\n" for c, h, l in _by_name(rows, dic))


def test_q4_semi_join_plan_matches_golden(ctx, db, monkeypatch):
    """SEMI join against a build side with duplicate keys (lineitem rows per order) behind a column-vs-column filter, two ways: the orders of the
    quarter binary-search lineitem (clustered by the order key), the pairs — the late-line filter applied to them — mark their orders; and, with
    that switched off, an existence table over every late line, probed by the orders"""
    for off, phrase in ((False, "through the pairs of the table-less join"), (True, "semi (marks + selection)")):
        if off:
            monkeypatch.setenv("PH_PLAN_NO_EXISTS_PAIRS", "1")
        p = tpch.q4_plan(db)
        p.run()
        r = p.fetch()
        ex = p.explain()
        p.free()
        assert q4_text(r) == golden("plan_q4.txt"), ex
        assert phrase in ex, ex


def test_q5_six_table_chain_matches_golden(ctx, db):
    p = tpch.q5_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    assert q5_text(r) == golden("plan_q5.txt"), ex


def test_q12_case_and_in_list_matches_golden(ctx, db):
    p = tpch.q12_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    assert q12_text(r) == golden("plan_q12.txt"), ex


def test_q14_case_like_and_float_result_matches_golden(ctx, db, sf1):
    p = tpch.q14_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    rc, f, a, b = O.q14(sf1, "PROMO%", tpchgen.days(1996, 4, 1), tpchgen.days(1996, 5, 1))
    assert r["ngroups"] == 1 and r["sum"][0] == [a.unscaled(4), b.unscaled(4)], ex     # both decimal sums bit-exact
    assert f"#\n{float(tpch.q14_promo_revenue(r))!r}\n" == golden("plan_q14.txt")


def test_q19_or_of_conjunctions_matches_golden(ctx, db):
    p = tpch.q19_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    assert r["ngroups"] == 1, ex
    assert f"#\n{tpch.dec_text(r['sum'][0][0], 4)}\n" == golden("plan_q19.txt"), ex


def test_q7_two_nation_joins_and_pair_filter_match_golden(ctx, db, sf1):
    """six tables, the nation table joined twice, an OR of conjunctions over both nation names as a Filter above the joins,
    EXTRACT(year) as a group key: the four groups equal the oracle's and the text cases/tpch/1g/plan/q7.txt"""
    p = tpch.q7_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    dic = tpchgen.nation_names()
    rows, n = O.q7_rows(sf1, "FRANCE", "ARGENTINA", tpchgen.days(1995, 1, 1), tpchgen.days(1996, 12, 31))
    want = {(rows[i].supp_nation, rows[i].cust_nation, rows[i].l_year): rows[i].revenue.unscaled(4) for i in range(n)}
    got = {(int(r["keys"][g][0]), int(r["keys"][g][1]), int(r["keys"][g][2])): r["sum"][g][0] for g in range(r["ngroups"])}
    assert got == want, ex
    text = "#\t\t\t\n" + "".join(f"{dic[a]}\t{dic[b]}\t{y}\t{tpch.dec_text(v, 4)}\n" for (a, b, y), v in sorted(got.items(), key=lambda kv: (dic[kv[0][0]], dic[kv[0][1]], kv[0][2])))
    assert text == golden("plan_q7.txt"), ex


def test_q8_eight_tables_and_case_sums_match_oracle(ctx, db, sf1):
    """eight tables; both sums per year equal the oracle's bit for bit (the oracle's quotient of them is what q8.txt pins:
    tests/test_golden_tpch.py; the host layer's own DECIMAL division: tests/test_host_layer.py)"""
    p = tpch.q8_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    rows, n = O.q8_rows(sf1, "ARGENTINA", "AMERICA", "ECONOMY BURNISHED TIN", tpchgen.days(1995, 1, 1), tpchgen.days(1996, 12, 31))
    want = {rows[i].o_year: (rows[i].nation_volume.unscaled(4), rows[i].volume.unscaled(4)) for i in range(n)}
    got = {int(r["keys"][g][0]): (r["sum"][g][0], r["sum"][g][1]) for g in range(r["ngroups"])}
    assert got == want and sorted(got) == [1995, 1996], ex


def test_q11_having_against_a_scalar_subquery_matches_golden(ctx, db, sf1):
    """Q11: the grouped and the ungrouped sum(ps_supplycost * ps_availqty) over partsupp x supplier x nation[JAPAN] from two plans; the
    float32 HAVING threshold applied over the fetched groups gives the oracle's 1225 rows and the golden's text"""
    g, t = tpch.q11_plans(db)
    g.run(); t.run()
    rg, rt = g.fetch(), t.fetch()
    ex = g.explain()
    t.free()
    rows = tpch.q11_rows(rg, rt)
    # ... and with the HAVING applied on the device (ph_plan_set_having): the same groups, only they are fetched
    assert tpch.q11_set_having(g, rt)
    g.run()
    rd = g.fetch()
    assert rd["ngroups"] == len(rows) < rg["ngroups"]
    assert sorted((int(rd["keys"][i][0]), rd["sum"][i][0]) for i in range(rd["ngroups"])) == sorted(rows)
    g.free()
    orows, n = O.q11_rows(sf1)
    assert sorted(rows) == sorted((orows[i].ps_partkey, orows[i].value.unscaled(2)) for i in range(n)), ex
    text = "#\t\n" + "".join(f"{k}\t{tpch.dec_text(v, 2)}\n" for k, v in sorted(rows, key=lambda kv: -kv[1]))
    assert text == golden("plan_q11.txt"), ex


def test_q15_cte_maximum_and_supplier_strings_match_golden(ctx, db, sf1):
    """Q15: the CTE's aggregate as the plan, its maximum and the DECIMAL equality over the fetched groups, the supplier's generated VARCHAR
    columns read back from the resident table: the oracle's rows and cases/tpch/1g/plan/q15.txt"""
    p = tpch.q15_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    rows = tpch.q15_rows(r)
    orows, n = O.q15_rows(sf1, tpchgen.days(1995, 12, 1), tpchgen.days(1996, 3, 1))
    assert rows == [(orows[i].s_suppkey, orows[i].total_revenue.unscaled(4)) for i in range(n)], ex
    assert tpch.q15_text(db, rows, sf1["supplier"]["s_suppkey"], r["scale"][0]) == golden("plan_q15.txt"), ex


def test_q20_two_key_join_with_the_subquery_aggregate_matches_golden(ctx, db, sf1):
    """Q20: SEMI join on a LIKE-filtered part, a join on two keys whose build side is an aggregate by those keys, the float32 predicate over
    the fetched groups, the nation's suppliers from a second plan: the oracle's keys and cases/tpch/1g/plan/q20.txt (177 generated addresses)"""
    p, s = tpch.q20_plans(db)
    p.run(); s.run()
    r, rs = p.fetch(), s.fetch()
    ex = p.explain()
    p.free(); s.free()
    keys = tpch.q20_keys(r, rs)
    assert keys == O.q20_keys(sf1).tolist(), ex
    assert tpch.q20_text(db, keys, sf1["supplier"]["s_suppkey"]) == golden("plan_q20.txt"), ex
    assert "groups stay on the device" in ex


@pytest.mark.parametrize("residual", [True, False])
def test_q21_exists_with_a_non_equi_condition_matches_golden(ctx, db, sf1, residual):
    """Q21's EXISTS / NOT EXISTS with `l2.l_suppkey <> l1.l_suppkey`, two ways: SEMI / ANTI joins with a RESIDUAL condition (the key matches
    filtered, the l1 rows that keep one marked), and without residual conditions (the pairs of an N:M join on l_orderkey filtered by a
    column-vs-column <>, the l1 rows that keep a pair as an aggregate by lineitem's primary key below a two-key SEMI / ANTI join). A VARCHAR
    group key: the oracle's groups and cases/tpch/1g/plan/q21.txt"""
    p = tpch.q21_plan(db, residual=residual)
    p.run()
    r = p.fetch()
    ex = p.explain()
    text = tpch.q21_text(db, p, r)
    p.free()
    orows, n = O.q21_rows(sf1)
    assert r["ngroups"] == n, ex
    assert text == golden("plan_q21.txt"), ex
    assert ("with a residual condition" in ex) == residual, ex


def test_residual_condition_of_an_inner_join_filters_its_pairs(ctx, db, sf1):
    """an INNER join with a residual condition = the join, then a Filter over its rows (two plans, the same rows): partsupp x part[p_size >= 40] on
    the part key where ps_availqty < p_size (a quarter of a percent of the pairs)"""
    def plan(as_residual):
        p = hip.Plan(db.ctx)
        part = p.scan(db.t("part"), db.c("part", "p_partkey", "p_size"), [tpch._pred(db, "part", "p_size", hip.PH_GE, hip.const(hip.PH_I32, i=40))])
        ps = p.scan(db.t("partsupp"), db.c("partsupp", "ps_partkey", "ps_availqty", "ps_suppkey"))
        cond = hip.bool_tree(("colcmp", 1, hip.PH_LT, 4))                                           # ps_availqty < p_size  [partsupp 0..2 | part 3, 4]
        if as_residual:
            j = p.join(ps, part, [0], [0], [0, 2, 4], residual=cond)
        else:
            j0 = p.join(ps, part, [0], [0], [0, 1, 2, 3, 4])
            f = p.filter(j0, bools=cond)
            j = p.project(f, [hip.pe_col(0), hip.pe_col(2), hip.pe_col(4)])
        p.agg(j, [hip.pe_col(2)], [(hip.PH_A_COUNT_STAR, None), (hip.PH_A_SUM, hip.pe_col(1))])
        return p.create()
    res = []
    for as_residual in (True, False):
        p = plan(as_residual)
        p.run()
        r = p.fetch()
        ex = p.explain()
        p.free()
        res.append(sorted((int(r["keys"][g][0]), int(r["count"][g][0]), int(r["sum"][g][1])) for g in range(r["ngroups"])))
        assert ("residual condition over the pairs" in ex) == as_residual, ex
    assert res[0] == res[1] and len(res[0]) > 0


def test_q22_substring_keys_anti_join_and_scalar_average_match_golden(ctx, db, sf1):
    """Q22: substring() computed in the plan as a filter operand (IN = OR of =) and as the group key, a scalar avg(DECIMAL) subquery as its own
    plan, DECIMAL > DECIMAL as an exact threshold, NOT EXISTS as an ANTI join: the oracle's groups and cases/tpch/1g/plan/q22.txt"""
    s = tpch.q22_scalar_plan(db)
    s.run()
    thr = tpch.q22_threshold(s.fetch())
    s.free()
    p = tpch.q22_plan(db, thr)
    p.run()
    r = p.fetch()
    ex = p.explain()
    text = tpch.q22_text(db, p, r)
    p.free()
    orows, n = O.q22_rows(sf1)
    assert r["ngroups"] == n, ex
    assert text == golden("plan_q22.txt"), ex


@pytest.mark.parametrize("offset,length,op", [(1, 3, hip.PH_NE), (-4, 3, hip.PH_EQ), (3, 5, hip.PH_NE)])
def test_plan_substring_as_filter_operand_and_group_key_matches_oracle(ctx, db, sf1, offset, length, op):
    """PH_PE_SUBSTR on its own: substring(p_name ..) with a positive and a negative offset, = and <> against a constant, as the group key of
    count(*) / sum(p_size) — against oracle_substring over the same strings (the reference semantics: function_operator_binary.go:553-625)"""
    P = sf1["part"]
    off, by = P["p_name_off"], P["p_name_bytes"].tobytes()
    subs = [O.substring(by[off[i]:off[i + 1]], offset, length) for i in range(20000)]
    k = max(set(subs), key=subs.count)                       # the most frequent value as the constant
    keep = [(s == k) == (op == hip.PH_EQ) for s in subs]
    want = {}
    for i, s in enumerate(subs):
        if keep[i]:
            c, v = want.get(s, (0, 0))
            want[s] = (c + 1, v + int(P["p_size"][i]))
    p = hip.Plan(ctx)
    part = p.scan(db.t("part"), db.c("part", "p_name", "p_size", "p_partkey"), [hip.pred(db.c("part", "p_partkey")[0], hip.PH_LE, hip.const(hip.PH_I32, i=20000))])
    pr = p.project(part, [hip.pe_substr(0, offset, length), hip.pe_col(1)])
    f = p.filter(pr, bools=hip.bool_tree(("cmp", 0, op, hip.const(hip.PH_STR, s=k.decode()))))
    p.agg(f, [hip.pe_col(0)], [(hip.PH_A_COUNT_STAR, None), (hip.PH_A_SUM, hip.pe_col(1))])
    p.create()
    p.run()
    r = p.fetch()
    typ, _sc, tab, col = hip.plan_key_info(p, 0)
    assert typ == hip.PH_STR
    names = hip.table_strings(ctx, tab, col, [int(r["keys"][g][0]) for g in range(r["ngroups"])])
    got = {names[g].encode(): (int(r["count"][g][0]), r["sum"][g][1]) for g in range(r["ngroups"])}
    ex = p.explain()
    p.free()
    assert got == want, ex


def test_q15_as_one_plan_with_a_join_root_matches_golden(ctx, db, sf1):
    """a plan whose ROOT is a join (ph_plan_fetch_rows): the CTE with two parents, max() over it as an ungrouped aggregate below a join on a
    DECIMAL key, the supplier's three VARCHAR columns gathered on the device: cases/tpch/1g/plan/q15.txt"""
    p = tpch.q15_rows_plan(db)
    p.run()
    r = p.fetch_rows()
    ex = p.explain()
    p.free()
    assert r["nrows"] == 1 and r["types"][1] == hip.PH_STR, ex
    assert tpch.q15_rows_text(r) == golden("plan_q15.txt"), ex
    assert "lowered before in this run, reused" in ex


def test_q17_and_q20_with_their_float_predicates_inside_the_plan_match_the_goldens(ctx, db, sf1):
    """PH_PE_FLOAT: Q17's DOUBLE predicate (l_quantity < 0.2 * avg, float64 with the FLOAT literal widened) and Q20's FLOAT one (ps_availqty >
    0.5 * sum, float32) as flag columns under a Filter; Q20 whole as a plan whose root is the SEMI join (rows: s_name, s_address)"""
    p = tpch.q17_whole_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    f, total = tpch.q17_avg_of_sum(r)
    rc, of, osum = O.q17(sf1)
    assert rc == 0 and total == osum.unscaled(2), ex
    assert f"#\n{float(f)!r}\n" == golden("plan_q17.txt"), ex
    p = tpch.q20_whole_plan(db)
    p.run()
    rows = p.fetch_rows()
    ex = p.explain()
    p.free()
    names, addrs = rows["columns"]
    text = "#\t\n" + "".join(f"{a}\t{b}\n" for a, b in sorted(zip(names, addrs)))
    assert text == golden("plan_q20.txt"), ex


def test_q17_decorrelated_average_joined_back_matches_golden(ctx, db, sf1):
    """Q17: an aggregate by the correlation key below a join whose payload is its SUM and COUNT; the DOUBLE predicate and the float32
    division over the fetched groups: the oracle's exact sum and cases/tpch/1g/plan/q17.txt"""
    p = tpch.q17_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    f, total = tpch.q17_avg_yearly(r)
    rc, of, osum = O.q17(sf1)
    assert rc == 0 and total == osum.unscaled(2), ex
    assert f"#\n{float(f)!r}\n" == golden("plan_q17.txt"), ex
    assert float(f) == of
    assert "groups stay on the device" in ex


def test_q18_subquery_aggregate_varchar_key_matches_golden(ctx, db):
    """an aggregate below a SEMI join (its 1.5 M groups stay on the device, HAVING is a Filter over them), five group keys — c_name a
    VARCHAR interned on the device, two narrow keys packed into one key word: cases/tpch/1g/plan/q18.txt byte for byte"""
    p = tpch.q18_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    text = tpch.q18_text(db, p, r)
    p.free()
    assert text == golden("plan_q18.txt"), ex
    assert "groups stay on the device" in ex


@pytest.mark.parametrize("jt", ["inner", "semi", "anti"])
def test_join_rooted_plan_rows_match_the_oracle_join(ctx, jt):
    """a plan whose root is a join over two ad-hoc tables with duplicate keys on both sides and a Filter under the build side: the rows
    ph_plan_fetch_rows returns are the oracle's pair set (INNER: probe key, probe payload, build payload) / its marked probe rows (SEMI, ANTI)"""
    rng = np.random.default_rng({"inner": 1, "semi": 2, "anti": 3}[jt])
    nb, npr = 30_000, 90_000
    bk = rng.integers(0, 20_000, nb).astype(np.int32)
    bv = rng.integers(0, 1000, nb).astype(np.int64)
    pk = rng.integers(0, 25_000, npr).astype(np.int32)
    pv = np.arange(npr, dtype=np.int64) * 7
    B = hip.Table(ctx, [dict(typ=hip.PH_I32, arr=bk), dict(typ=hip.PH_DEC64, arr=bv, scale=2)], nb)
    P = hip.Table(ctx, [dict(typ=hip.PH_I32, arr=pk), dict(typ=hip.PH_I64, arr=pv)], npr)
    p = hip.Plan(ctx)
    build = p.scan(B, [0, 1], [hip.pred(1, hip.PH_GT, hip.const(hip.PH_DEC64, i=30000, scale=2))])     # bv > 300.00
    probe = p.scan(P, [0, 1])
    if jt == "inner":
        p.join(probe, build, [0], [0], [0, 1, 3])
    else:
        p.join(probe, build, [0], [0], [0, 1], join_type=hip.PH_JT_SEMI if jt == "semi" else hip.PH_JT_ANTI)
    p.create()
    p.run()
    r = p.fetch_rows()
    ex = p.explain()
    p.free()
    bsel = np.nonzero(bv > 30000)[0].astype(np.int64)
    oj = O.Join([O.col(O.OT_INT32, bk)], bsel, len(bsel))
    if jt == "inner":
        m, wp, wb = oj.probe_inner([O.col(O.OT_INT32, pk)], None, npr, 1 << 22)
        want = sorted(zip(pk[wp].tolist(), pv[wp].tolist(), bv[wb].tolist()))
        got = sorted(zip(r["columns"][0].tolist(), r["columns"][1].tolist(), r["columns"][2].tolist()))
        assert r["nrows"] == m and r["scales"][2] == 2, ex
    else:
        f = oj.probe_mark([O.col(O.OT_INT32, pk)], None, npr).astype(bool)
        keep = f if jt == "semi" else ~f
        want = sorted(zip(pk[keep].tolist(), pv[keep].tolist()))
        got = sorted(zip(r["columns"][0].tolist(), r["columns"][1].tolist()))
    assert got == want, ex
    B.free(); P.free()


def test_tables_of_one_context_serve_concurrent_plans_on_two_others(sf1):
    """VERDICT r3 item 2 / SURVEY §8(b) threading (psql server path, cmd/main/main.go:71-122: handles independent, only the table cache
    shared). The arrangement of the Go shim: tables live on context A; two threads each own a context (B, C) and create, run and fetch
    plans over those tables AT THE SAME TIME — Q3 on B, Q9 on C (whose second late materialisation makes the library build the
    co-located copy of five lineitem columns while Q3 is reading the same table). Every iteration's result equals the reference's
    golden; the copy is found through the process-wide registry although the calling ctx never created a table, and it is built once."""
    import threading
    a, b, c = hip.Ctx(0), hip.Ctx(0), hip.Ctx(0)
    db = tpch.Database(a, sf1)
    lt = db.t("lineitem")
    assert lt.colocate_bytes() == 0
    errors, texts = [], {"q3": [], "q9": []}

    def worker(name, ctx, build, render):
        try:
            view = db.on(ctx)
            for _ in range(4):
                p = build(view)
                p.run()
                r = p.fetch()
                texts[name].append((render(r), p.explain()))
                p.free()
        except Exception as e:   # noqa: BLE001 - reported by the main thread
            errors.append((name, repr(e)))

    t3 = threading.Thread(target=worker, args=("q3", b, tpch.q3_plan, lambda r: pipelines.q3_text(tpch.q3_top(r))))
    t9 = threading.Thread(target=worker, args=("q9", c, tpch.q9_plan, lambda r: pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names())))
    t3.start(); t9.start(); t3.join(); t9.join()
    assert not errors, errors
    assert len(texts["q3"]) == 4 and len(texts["q9"]) == 4
    for text, ex in texts["q3"]:
        assert text == golden("plan_q3.txt"), ex
    for text, ex in texts["q9"]:
        assert text == golden("plan_q9.txt"), ex
    held = lt.colocate_bytes()
    assert held > 0, "the co-located copy was never built: the calling ctx did not find the table's columns"
    # one more round on yet another pair of roles (the copy exists: every consumer is ordered against its build by the event)
    p = tpch.q9_plan(db.on(b)); p.run(); r = p.fetch(); p.free()
    assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == golden("plan_q9.txt")
    assert lt.colocate_bytes() == held
    db.free()
    for x in (c, b, a):
        x.close()


def test_colocate_budget_is_the_hosts_veto(sf1):
    """ph_table_set_colocate_budget(0): the library builds no co-located copy on its own (Q9 still gives the golden, through the column
    arrays); a budget with room lets the second run build it; ph_table_colocate_bytes shows what is held."""
    ctx = hip.Ctx(0)
    db = tpch.Database(ctx, sf1)
    lt = db.t("lineitem")
    lt.set_colocate_budget(0)
    for _ in range(3):
        p = tpch.q9_plan(db); p.run(); r = p.fetch(); p.free()
        assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == golden("plan_q9.txt")
    assert lt.colocate_bytes() == 0
    lt.set_colocate_budget(1 << 30)
    db2 = tpch.Database(ctx, sf1)          # a fresh table: the vetoed one does not try again
    l2 = db2.t("lineitem")
    for _ in range(3):
        p = tpch.q9_plan(db2); p.run(); r = p.fetch(); p.free()
        assert pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()) == golden("plan_q9.txt")
    assert 0 < l2.colocate_bytes() <= 1 << 30
    db.free(); db2.free()
    ctx.close()


# ---- round 4: the last four reference goldens (the queries that read the generator's COMMENT text)

def test_q16_count_distinct_and_not_in_match_golden(ctx, db, sf1):
    """Q16: PH_A_COUNT_DISTINCT (a distinct side table inside the library, re-sunk into the aggregate: SinkDistinctGrouping /
    DistinctGrouping), NOT IN as an ANTI join, `<>` / NOT LIKE / IN over dictionary columns, LIKE '%Customer%Complaints%' over the
    suppliers' generated comments: 18 341 groups, cases/tpch/1g/plan/q16.txt byte for byte"""
    p = tpch.q16_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    p.free()
    assert "count(distinct)" in ex, ex
    assert tpch.q16_text(r) == golden("plan_q16.txt"), ex
    rows, n = O.q16_rows(sf1)
    assert r["ngroups"] == n == 18341


def test_q13_left_join_and_null_count_key_match_golden(ctx, db, sf1, monkeypatch):
    """Q13: count(o_orderkey) over the NULL-extended side of a LEFT join grouped by the customer key, the count's NULL-for-zero as the GROUP KEY
    of the aggregate above (key_null in the result: the golden's first row is NULL\t50005), NOT LIKE with two '%' over 1.5 M order comments.
    Two physical forms, the same text: children per parent (ph_count_by_key: no pairs) and, with that switched off, PH_JT_LEFT as the
    reference runs it (NextLeftJoin: pairs + unmatched rows, then the hash aggregate)"""
    for off, phrase in ((False, "children per parent"), (True, "LEFT OUTER")):
        if off:
            monkeypatch.setenv("PH_PLAN_NO_COUNT_PUSHDOWN", "1")
        p = tpch.q13_plan(db)
        p.run()
        r = p.fetch()
        ex = p.explain()
        p.free()
        assert phrase in ex, ex
        assert r["key_null"] is not None and int(r["key_null"].sum()) == 1
        assert tpch.q13_text(r) == golden("plan_q13.txt"), ex
    monkeypatch.delenv("PH_PLAN_NO_COUNT_PUSHDOWN")
    # ... and the specification's default pattern: the publicly known answer's first rows
    p = tpch.q13_plan(db, notlike="%special%requests%")
    p.run()
    r = p.fetch()
    p.free()
    assert tpch.q13_text(r) == O.q13_text(sf1, notlike="%special%requests%")
    assert tpch.q13_text(r).split("\n")[1:4] == ["NULL\t50005", "9\t6641", "10\t6532"]


def test_q2_minimum_joined_back_with_text_columns_matches_golden(ctx, db):
    """Q2 as ONE join-rooted plan: the correlated min() as an aggregate by its key over a subtree with two parents, joined back on (key,
    DECIMAL value); four VARCHAR columns of supplier (s_comment from the text pool) gathered on the device for the 100 rows"""
    p = tpch.q2_plan(db)
    p.run()
    r = p.fetch_rows()
    ex = p.explain()
    p.free()
    assert tpch.q2_text(r) == golden("plan_q2.txt"), ex
    # ORDER BY s_acctbal DESC .. LIMIT k announced (ph_plan_set_rows_topk): exactly the rows at least as good as the k-th come back — ties
    # of the k-th value included — and the text is the same; ascending over an INTEGER column (p_partkey) likewise
    bal = np.asarray(r["columns"][0], dtype=np.int64)
    for k in (100, 7, 1):
        p = tpch.q2_plan(db)
        p.set_rows_topk(0, k, descending=True)
        p.run()
        rk = p.fetch_rows()
        ex = p.explain()
        p.free()
        kth = np.sort(bal)[::-1][k - 1]
        assert "LIMIT" in ex and rk["nrows"] == int((bal >= kth).sum()) and int(np.asarray(rk["columns"][0], dtype=np.int64).min()) == kth, ex
        assert tpch.q2_text(rk, limit=k) == tpch.q2_text(r, limit=k)
    pk = np.asarray(r["columns"][3], dtype=np.int64)
    p = tpch.q2_plan(db)
    p.set_rows_topk(3, 50, descending=False)
    p.run()
    rk = p.fetch_rows()
    p.free()
    assert rk["nrows"] == int((pk <= np.sort(pk)[49]).sum()) and sorted(np.asarray(rk["columns"][3]).tolist()) == sorted(pk[pk <= np.sort(pk)[49]].tolist())


def test_q10_seven_group_keys_match_golden(ctx, db):
    """Q10: seven group keys — four VARCHAR columns interned on the device, narrow keys packed two per key word (a dictionary code widened to
    an INTEGER for it) — top-k preselection for ORDER BY revenue DESC LIMIT 20; c_address / c_comment come back as rows of their columns"""
    p = tpch.q10_plan(db)
    p.run()
    r = p.fetch()
    ex = p.explain()
    assert tpch.q10_text(db, p, r) == golden("plan_q10.txt"), ex
    p.free()
