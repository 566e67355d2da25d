"""Parity at BASELINE.json's full size (SF10 lineitem, ~60M rows) through size-independent
properties, checked against plain numpy (not the oracle, which would take minutes here):
linearity over row-range shards, counts and integer sums of the filtered rows, and for the Q3 join
the probe row set against numpy's set membership."""
import numpy as np
import pytest

from plan_amd import hip, pipelines, queries, tpchgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf10():
    sf = (10, 1)
    L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax",
                                      "l_returnflag", "l_linestatus", "l_shipdate"])
    assert len(L["l_shipdate"]) == 59986052   # SF10 lineitem cardinality
    return L


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


def test_q1_sf10_properties(ctx, sf10):
    L = sf10
    n = len(L["l_shipdate"])
    t = queries.lineitem_table(ctx, L)
    p = queries.q1_plan(ctx, t)
    p.run()
    full = p.fetch()
    # (1) linearity: the whole table equals the sum over three ragged row-range shards
    cuts = [0, 20_000_004, 20_000_004 + 7 * 4, n]
    acc = {}
    for a, b in zip(cuts[:-1], cuts[1:]):
        p.run(a, b)
        r = p.fetch()
        for g in range(r["ngroups"]):
            k = tuple(r["keys"][g])
            s, c = acc.setdefault(k, ([0] * 8, [0] * 8))
            for i in range(8):
                s[i] += r["sum"][g][i]
                c[i] += r["count"][g][i]
    assert len(acc) == full["ngroups"] == 4
    for g in range(4):
        s, c = acc[tuple(full["keys"][g])]
        assert s == full["sum"][g] and c == full["count"][g]
    # (2) independent numpy check of counts and exact integer sums per group
    m = L["l_shipdate"] <= queries.q1_shipdate_cutoff()
    for g in range(4):
        f, st = full["keys"][g]
        sel = m & (L["l_returnflag"] == f) & (L["l_linestatus"] == st)
        e, d, tx = L["l_extendedprice"][sel], L["l_discount"][sel], L["l_tax"][sel]
        assert full["count"][g][7] == int(sel.sum())
        assert full["sum"][g][0] == int(L["l_quantity"][sel].sum(dtype=np.int64))
        assert full["sum"][g][1] == int(e.sum())
        dp = e * (100 - d)
        assert full["sum"][g][2] == int(dp.sum())
        assert full["sum"][g][3] == sum(int(x) for x in np.array_split(dp * (100 + tx), 64) for x in [x.sum()])
    p.free()
    # (3) Q6 on the same table
    p6 = queries.q6_plan(ctx, t)
    p6.run()
    r6 = p6.fetch()
    d1, d2, lo, hi, q = queries.q6_constants()
    m6 = (L["l_shipdate"] >= d1) & (L["l_shipdate"] < d2) & (L["l_discount"] >= 2) & (L["l_discount"] <= 4) & (L["l_quantity"] < q)
    assert r6["count"][0][0] == int(m6.sum())
    assert r6["sum"][0][0] == int((L["l_extendedprice"][m6] * L["l_discount"][m6]).sum())
    p6.free()
    t.free()


def test_q3_sf10_join_rows_match_numpy(ctx, sf10):
    sf = (10, 1)
    Od = tpchgen.orders(sf, columns=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    C = tpchgen.customer(sf)
    L = {k: sf10[k] for k in ("l_orderkey", "l_extendedprice", "l_discount", "l_shipdate")}
    pipe = pipelines.Q3Pipeline(ctx, L, Od, C)
    r = pipe.run(want_groups=True)
    pipe.free()
    date = tpchgen.days(1995, 3, 29)
    cust = C["c_custkey"][C["c_mktsegment"] == tpchgen.MKTSEGMENT_DICT.index("HOUSEHOLD")]
    omask = (Od["o_orderdate"] < date) & np.isin(Od["o_custkey"], cust)
    okeys = Od["o_orderkey"][omask]
    lmask = (L["l_shipdate"] > date) & np.isin(L["l_orderkey"], okeys)
    assert r["join_rows"] == int(lmask.sum())
    keys = L["l_orderkey"][lmask]
    rev = L["l_extendedprice"][lmask] * (100 - L["l_discount"][lmask])
    uk, inv = np.unique(keys, return_inverse=True)
    sums = np.bincount(inv, weights=None, minlength=len(uk)) * 0
    want = np.zeros(len(uk), np.int64)
    np.add.at(want, inv, rev)
    got = dict((g[0], g[1]) for g in r["groups"])
    assert len(got) == len(uk) == r["ngroups"]
    assert all(got[int(k)] == int(v) for k, v in zip(uk[:50000], want[:50000]))
    assert sum(got.values()) == int(rev.sum())          # checksum of all groups


def test_q9_sf10_join_rows_and_profit_match_numpy(ctx, sf10):
    """Q9 at SF10 (the size bench.py's Q9 figure is quoted on; the LDS coarse-bitmap candidate
    kernel and the semi-join reduction of partsupp are selected by these build-side sizes): the
    join-row count, the 175 (nation, year) keys and the exact profit sum of every group against
    numpy lookups (np.isin / searchsorted / np.add.at), independent of the oracle."""
    sf = (10, 1)
    L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity",
                                      "l_extendedprice", "l_discount"])
    Od = tpchgen.orders(sf, columns=["o_orderkey", "o_orderdate"])
    P, PS, S = tpchgen.part(sf), tpchgen.partsupp(sf), tpchgen.supplier(sf)
    assert len(L["l_orderkey"]) == 59986052 and np.array_equal(L["l_orderkey"], sf10["l_orderkey"])
    pipe = pipelines.Q9Pipeline(ctx, L, Od, P, PS, S)
    r = pipe.run()
    pipe.free()
    # p_name like '%pink%': a name is five colour words joined by blanks, the pattern has no blank
    pink_words = [i for i, w in enumerate(tpchgen.colors()) if "pink" in w]
    pink = P["p_partkey"][np.isin(P["p_name_colors"].reshape(-1, 5), pink_words).any(axis=1)]
    lm = np.isin(L["l_partkey"], pink)
    lp, ls = L["l_partkey"][lm], L["l_suppkey"][lm]
    # partsupp on (partkey, suppkey), supplier on suppkey, orders on orderkey: each N:1
    pskey = PS["ps_partkey"].astype(np.int64) << 32 | PS["ps_suppkey"].astype(np.int64)
    order = np.argsort(pskey, kind="stable")
    want_ps = lp.astype(np.int64) << 32 | ls.astype(np.int64)
    pos = np.searchsorted(pskey[order], want_ps)
    assert np.array_equal(pskey[order][pos], want_ps)            # every lineitem has its partsupp row
    cost = PS["ps_supplycost"][order][pos]
    so = np.argsort(S["s_suppkey"], kind="stable")
    spos = np.searchsorted(S["s_suppkey"][so], ls)
    assert np.array_equal(S["s_suppkey"][so][spos], ls)
    nat = S["s_nationkey"][so][spos]
    opos = np.searchsorted(Od["o_orderkey"], L["l_orderkey"][lm])   # o_orderkey ascends
    assert np.array_equal(Od["o_orderkey"][opos], L["l_orderkey"][lm])
    year = (Od["o_orderdate"][opos].astype("datetime64[D]").astype("datetime64[Y]").astype(np.int64) + 1970)
    # amount = l_extendedprice*(1-l_discount) [scale 4] - ps_supplycost*l_quantity [scale 2 -> 4]
    amount = L["l_extendedprice"][lm] * (100 - L["l_discount"][lm]) - cost * L["l_quantity"][lm].astype(np.int64) * 100
    assert r["join_rows"] == int(lm.sum())
    gid = nat.astype(np.int64) * 4096 + year
    uk, inv = np.unique(gid, return_inverse=True)
    want = np.zeros(len(uk), np.int64)
    np.add.at(want, inv, amount)
    got = {a * 4096 + b: c for a, b, c in r["rows"]}
    assert r["ngroups"] == len(uk) == 175
    assert got == {int(k): int(v) for k, v in zip(uk, want)}


def test_whole_plans_agree_with_their_two_step_forms_at_sf10(ctx):
    """Q15, Q17 and Q20 at SF10, each in two independently lowered formulations whose SF1 results both equal the reference's golden
    (tests/test_gpu_plan.py): the whole tree as one plan (join-rooted rows / FLOAT predicate inside the plan) against the aggregate-rooted
    plan whose upper operators run over the fetched groups. At full size the two must give the same rows — exact DECIMAL sums included."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from plan_amd import tpch
    sf = (10, 1)
    data = {"sf": sf,
            "lineitem": tpchgen.lineitem(sf, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_shipdate"]),
            "part": tpchgen.part(sf), "partsupp": tpchgen.partsupp(sf), "supplier": tpchgen.supplier(sf)}
    db = tpch.Database(ctx, data)
    try:
        # Q15
        p = tpch.q15_rows_plan(db); p.run(); whole = p.fetch_rows(); p.free()
        p = tpch.q15_plan(db); p.run(); rows = tpch.q15_rows(p.fetch()); p.free()
        assert whole["nrows"] == len(rows) >= 1
        assert sorted(zip(whole["columns"][0].tolist(), whole["columns"][4].tolist())) == rows
        assert all(n == f"Supplier#{int(k):09d}" for k, n in zip(whole["columns"][0], whole["columns"][1]))
        # Q17
        p = tpch.q17_whole_plan(db); p.run(); f1, s1 = tpch.q17_avg_of_sum(p.fetch()); p.free()
        p = tpch.q17_plan(db); p.run(); f2, s2 = tpch.q17_avg_yearly(p.fetch()); p.free()
        assert s1 == s2 and s1 > 0 and float(f1) == float(f2)
        # Q20
        p = tpch.q20_whole_plan(db); p.run(); w = p.fetch_rows(); p.free()
        g, s = tpch.q20_plans(db); g.run(); s.run(); keys = tpch.q20_keys(g.fetch(), s.fetch()); g.free(); s.free()
        assert w["nrows"] == len(keys) > 1000
        assert sorted(w["columns"][0]) == [f"Supplier#{k:09d}" for k in keys]
    finally:
        db.free()
