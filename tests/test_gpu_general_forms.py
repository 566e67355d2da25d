"""Pipeline-level parity of the GENERAL-FORM plans ("no pipeline-level parity test on unclustered data").

Q3Pipeline / Q9Pipeline pick the gated sorted fill, ph_merge_lookup and ph_agg_sink_sorted from statistics computed
at load, and fall back to the general forms (bulk hash aggregate, general direct / node builds, table lookups) when a
statistic is false or — after a deferred PH_ECONSTRAINT — turns out to be wrong. Generator-ordered tables never take
those paths; here every table is row-shuffled (one permutation per table, all its columns), so
  * with honest statistics the general forms run from the start, and
  * with statistics that are deliberately WRONG (claimed sorted / unique / clustered on shuffled data) the
    optimistic forms run first, the device detects the broken claim, and the whole query reruns.
Both must give the oracle's groups and the reference's golden text."""
import os

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, pipelines, tpchgen

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


def shuffled(table, rng):
    table = {k: v for k, v in table.items() if k in ("p_name_off", "p_name_bytes") or not (k.endswith("_off") or k.endswith("_bytes"))}   # (the pipelines read no other VARCHAR column)
    n = len(next(iter(table.values())))
    fixed = {k: v for k, v in table.items() if k in ("p_name_off", "p_name_bytes")}
    perm = rng.permutation(n)
    out = {k: np.ascontiguousarray(v[perm]) for k, v in table.items() if k not in fixed}
    if fixed:   # VARCHAR column as offsets + bytes: rebuild both for the permuted rows
        off, by = table["p_name_off"], table["p_name_bytes"]
        lens = (off[1:] - off[:-1])[perm]
        noff = np.zeros(n + 1, dtype=off.dtype)
        np.cumsum(lens, out=noff[1:])
        nby = np.empty(int(noff[-1]), dtype=by.dtype)
        starts = off[:-1][perm]
        for i in range(n):   # 200 k short strings at SF1
            nby[noff[i]:noff[i + 1]] = by[starts[i]:starts[i] + lens[i]]
        out["p_name_off"], out["p_name_bytes"] = noff, nby
    return out


@pytest.fixture(scope="module")
def sf1_shuffled(sf1):
    rng = np.random.default_rng(20261004)
    return {k: (shuffled(v, rng) if isinstance(v, dict) else v) for k, v in sf1.items()}


@pytest.fixture(scope="module")
def q3_want(sf1):
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    return {(rows[i].l_orderkey, rows[i].revenue.unscaled(4), rows[i].o_orderdate, rows[i].o_shippriority) for i in range(n)}


def test_q3_general_form_on_shuffled_tables(ctx, sf1_shuffled, q3_want):
    """statistics honestly false: general direct build (scatter passes), residual / hash-table probe, bulk hash
    aggregate instead of the streaming one — all 11 378 groups equal the oracle's, the top 10 the golden"""
    t = sf1_shuffled
    p = pipelines.Q3Pipeline(ctx, t["lineitem"], t["orders"], t["customer"])
    assert not (p.o_key_sorted_unique or p.c_key_sorted_unique or p.l_key_sorted)
    r = p.run(want_groups=True)
    p.free()
    assert r["ngroups"] == 11378 and set(r["groups"]) == q3_want
    assert pipelines.q3_text(r["top"]) == open(os.path.join(GOLDEN, "plan_q3.txt")).read()


@pytest.mark.parametrize("claim", ["o_key_sorted_unique", "c_key_sorted_unique", "l_key_sorted", "all"])
def test_q3_wrong_statistic_forces_the_retry(ctx, sf1_shuffled, q3_want, claim):
    """a statistic that claims order on shuffled data: the gated fill / the streaming aggregate verify the claim on
    the device, the deferred PH_ECONSTRAINT aborts the attempt, the statistics are dropped and the rerun is right"""
    t = sf1_shuffled
    p = pipelines.Q3Pipeline(ctx, t["lineitem"], t["orders"], t["customer"])
    for name in (["o_key_sorted_unique", "c_key_sorted_unique", "l_key_sorted"] if claim == "all" else [claim]):
        setattr(p, name, True)
    p.time_stages = False
    r = p.run(want_groups=True)
    assert not (p.o_key_sorted_unique or p.c_key_sorted_unique or p.l_key_sorted), "the wrong claim was never detected"
    p.free()
    assert r["ngroups"] == 11378 and set(r["groups"]) == q3_want
    assert pipelines.q3_text(r["top"]) == open(os.path.join(GOLDEN, "plan_q3.txt")).read()
    ctx.check_deferred()   # nothing left pending for the next query


def test_q9_general_form_on_shuffled_tables(ctx, sf1_shuffled, sf1):
    """honest statistics on shuffled tables: table lookup instead of the merge lookup, general builds for supplier
    and orders; the 175 rows equal the golden text and the oracle"""
    t = sf1_shuffled
    p = pipelines.Q9Pipeline(ctx, t["lineitem"], t["orders"], t["part"], t["partsupp"], t["supplier"])
    assert not (p.o_key_sorted_unique or p.s_key_sorted_unique or p.l_key_sorted)
    p.time_stages = False
    r = p.run()
    p.free()
    text = pipelines.q9_text(r["rows"], tpchgen.nation_names())
    assert text == open(os.path.join(GOLDEN, "plan_q9.txt")).read()
    n, rows = O.q9(sf1, "%pink%")
    assert text == O.q9_text(rows, n, tpchgen.nation_names())


@pytest.mark.parametrize("claim", ["o_key_sorted_unique+l_key_sorted", "s_key_sorted_unique", "all"])
def test_q9_wrong_statistic_forces_the_retry(ctx, sf1_shuffled, claim):
    """claimed order on shuffled data: the merge lookup / the declared sorted fills detect it (deferred
    PH_ECONSTRAINT at the final download) and the query reruns with counted lookups over general tables"""
    t = sf1_shuffled
    p = pipelines.Q9Pipeline(ctx, t["lineitem"], t["orders"], t["part"], t["partsupp"], t["supplier"])
    names = ["o_key_sorted_unique", "l_key_sorted", "s_key_sorted_unique"] if claim == "all" else claim.split("+")
    for name in names:
        setattr(p, name, True)
    p.time_stages = False
    calls = []
    inner = p._run
    p._run = lambda strict: (calls.append(strict), inner(strict))[1]
    r = p.run()
    p.free()
    assert calls == [True, False], "the wrong claim was never detected"
    assert pipelines.q9_text(r["rows"], tpchgen.nation_names()) == open(os.path.join(GOLDEN, "plan_q9.txt")).read()
    ctx.check_deferred()
