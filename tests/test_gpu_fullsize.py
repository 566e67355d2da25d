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
