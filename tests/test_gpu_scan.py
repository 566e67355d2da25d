"""GPU parity of the fused Agg <- Scan(filter) path (Q1 / Q6 shapes) against the oracle.
Everything goes through the C-ABI (plan_amd.hip is a thin ctypes binding of include/planhip.h)."""
import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, queries, tpchgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


def oracle_q1_ints(L, cutoff):
    rows = O.q1(L, cutoff)
    out = []
    for r in rows:
        out.append(dict(key=(r.returnflag, r.linestatus), sum_qty=r.sum_qty.value(),
                        sum_base=r.sum_base_price.unscaled(2), sum_disc=r.sum_disc_price.unscaled(4),
                        sum_charge=r.sum_charge.unscaled(6), avg_qty=r.avg_qty,
                        count=r.count_order))
    return out


def check_q1(ctx, L, cutoff, row_begin=0, row_end=None):
    t = queries.lineitem_table(ctx, L)
    p = queries.q1_plan(ctx, t, cutoff)
    assert p.kind == "lowcard_chain"
    p.run(row_begin, row_end)
    r = p.fetch()
    p.free()
    t.free()
    sl = slice(row_begin, row_end)
    want = oracle_q1_ints({k: v[sl] for k, v in L.items()}, cutoff)
    assert r["ngroups"] == len(want)
    assert r["scale"][:4] == [0, 2, 4, 6]
    for g, w in enumerate(want):  # same (first-seen) order, bit-exact integers
        assert tuple(r["keys"][g]) == w["key"]
        s, c = r["sum"][g], r["count"][g]
        assert s[0] == w["sum_qty"] and s[1] == w["sum_base"]
        assert s[2] == w["sum_disc"] and s[3] == w["sum_charge"]
        assert c[7] == w["count"]
        # avg(INTEGER) is float64 sum / float64 count in the reference; tolerance 1e-9 relative
        assert abs(s[4] / c[4] - w["avg_qty"]) <= 1e-9 * abs(w["avg_qty"])
    return r


def test_q1_sf001(ctx, sf001):
    check_q1(ctx, sf001["lineitem"], queries.q1_shipdate_cutoff())


def test_q1_sf1(ctx, sf1):
    r = check_q1(ctx, sf1["lineitem"], queries.q1_shipdate_cutoff())
    assert sum(c[7] for c in r["count"]) == 5870362  # golden Σcount_order


def test_q1_ragged_ranges(ctx, sf001):
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    for b, e in [(0, 1), (0, 3), (4, 1027), (1024, 1024), (0, n - 1), (60000, n), (8, 2049)]:
        check_q1(ctx, L, queries.q1_shipdate_cutoff(), b, e)


def test_q1_filter_extremes(ctx, sf001):
    L = sf001["lineitem"]
    check_q1(ctx, L, tpchgen.days(1990, 1, 1))   # selects nothing -> no groups
    check_q1(ctx, L, tpchgen.days(2000, 1, 1))   # selects everything


def oracle_q6(L, consts):
    rc, d = O.q6(L, *consts)
    return None if rc == 1 else d.unscaled(4)


def check_q6(ctx, L, consts, row_begin=0, row_end=None):
    t = queries.lineitem_table(ctx, L)
    p = queries.q6_plan(ctx, t, consts)
    assert p.kind == "filter_sumprod"
    p.run(row_begin, row_end)
    r = p.fetch()
    p.free()
    t.free()
    sl = slice(row_begin, row_end)
    want = oracle_q6({k: v[sl] for k, v in L.items()}, consts)
    if want is None:
        assert r["ngroups"] == 0
    else:
        assert r["ngroups"] == 1 and r["scale"] == [4]
        assert r["sum"][0][0] == want
    return r


def test_q6_sf001(ctx, sf001):
    check_q6(ctx, sf001["lineitem"], queries.q6_constants())


def test_q6_sf1_matches_golden(ctx, sf1):
    r = check_q6(ctx, sf1["lineitem"], queries.q6_constants())
    assert r["sum"][0][0] == 616600517967  # cases/tpch/1g/plan/q6.txt: 61660051.7967


def test_q6_ragged_and_empty(ctx, sf001):
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    c = queries.q6_constants()
    for b, e in [(0, 0), (0, 5), (4, 1030), (0, n - 3), (59996, n)]:
        check_q6(ctx, L, c, b, e)
    # a year with no shipments -> the aggregate has no group
    check_q6(ctx, L, (tpchgen.days(2001, 1, 1), tpchgen.days(2002, 1, 1)) + c[2:])


def test_q6_float32_boundary_semantics(ctx):
    """discount values straddling the float32 constants: 0.02 passes `>= float32(0.03)-float32(0.01)`,
    0.04 passes `<= float32(0.03)+float32(0.01)`; checked row by row vs the oracle."""
    n = 4096
    rng = np.random.default_rng(7)
    L = dict(l_quantity=rng.integers(1, 51, n).astype(np.int32),
             l_extendedprice=rng.integers(90000, 10500000, n).astype(np.int64),
             l_discount=rng.integers(-3, 12, n).astype(np.int64),
             l_tax=rng.integers(0, 9, n).astype(np.int64),
             l_returnflag=rng.integers(0, 3, n).astype(np.uint8),
             l_linestatus=rng.integers(0, 2, n).astype(np.uint8),
             l_shipdate=np.full(n, tpchgen.days(1994, 6, 1), dtype=np.int32))
    check_q6(ctx, L, queries.q6_constants())


def test_overflow_is_refused(ctx):
    n = 2048
    big = np.full(n, 3_000_000_000_000_000, dtype=np.int64)
    L = dict(l_quantity=np.ones(n, np.int32), l_extendedprice=big, l_discount=big.copy(),
             l_tax=np.zeros(n, np.int64), l_returnflag=np.zeros(n, np.uint8),
             l_linestatus=np.zeros(n, np.uint8),
             l_shipdate=np.full(n, tpchgen.days(1994, 6, 1), dtype=np.int32))
    t = queries.lineitem_table(ctx, L)
    p = queries.q6_plan(ctx, t, (0, 20000, -1e30, 1e30, 100))
    with pytest.raises(hip.PlanHipError) as e:
        p.run()
    assert e.value.code == hip.PH_EOVERFLOW
    p.free()
    t.free()


def test_q3_pipeline_sf1_matches_reference_golden(ctx, sf1):
    """Q3 through the device operators: all 11 378 groups equal the oracle's, and the
    ORDER BY revenue DESC, o_orderdate LIMIT 10 tail equals cases/tpch/1g/plan/q3.txt."""
    import os
    from plan_amd import pipelines
    p = pipelines.Q3Pipeline(ctx, sf1["lineitem"], sf1["orders"], sf1["customer"])
    r = p.run(want_groups=True)
    p.free()
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    want = {(rows[i].l_orderkey, rows[i].revenue.unscaled(4), rows[i].o_orderdate, rows[i].o_shippriority)
            for i in range(n)}
    assert r["ngroups"] == n == 11378 and set(r["groups"]) == want
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "plan_q3.txt")).read()
    assert pipelines.q3_text(r["top"]) == golden


@pytest.mark.parametrize("segment,ymd", [("AUTOMOBILE", (1994, 1, 1)), ("FURNITURE", (1992, 2, 1)), ("HOUSEHOLD", (1998, 12, 1)),
                                         ("NOSUCHSEGMENT", (1995, 3, 15))])
def test_q3_pipeline_other_parameters_match_oracle(ctx, sf1, segment, ymd):
    """Q3 with other segments and cut-off dates than the golden's (other selectivities of both build-side gates,
    an empty result, a segment that is not in the dictionary): the top 10 equal the oracle's text."""
    from plan_amd import pipelines
    date = tpchgen.days(*ymd)
    p = pipelines.Q3Pipeline(ctx, sf1["lineitem"], sf1["orders"], sf1["customer"], segment=segment, date=date)
    p.time_stages = False
    r = p.run()
    p.free()
    n, rows = O.q3(sf1, segment, date)
    assert pipelines.q3_text(r["top"]) == O.q3_text(rows, n)


@pytest.mark.parametrize("pattern", ["%green%", "%zzzz%", "%a%"])
def test_q9_pipeline_other_patterns_match_oracle(ctx, sf1, pattern):
    """Q9 with other LIKE patterns: a different 5 % of the parts, no part at all, and 95 % of them (the part side
    then takes other table forms and probe kernels than the golden run's)."""
    from plan_amd import pipelines
    p = pipelines.Q9Pipeline(ctx, sf1["lineitem"], sf1["orders"], sf1["part"], sf1["partsupp"], sf1["supplier"], pattern=pattern)
    p.time_stages = False
    r = p.run()
    p.free()
    n, rows = O.q9(sf1, pattern)
    assert pipelines.q9_text(r["rows"], tpchgen.nation_names()) == O.q9_text(rows, n, tpchgen.nation_names())


def test_q3_partitioned_two_ranks_on_one_gpu(sf1):
    """The N>1 form of Q3 (broadcast customer keys, hash-partition orders' and lineitem rows by
    order key, exchange, local build/probe/aggregate, top-10 merge) with 2 ranks run as 2 threads
    on this GPU (plan_amd.dist.ThreadGroup stands in for the RCCL process group). Each rank owns
    half of the orders (and their lineitems) and half of the customers; the union of the ranks'
    groups must equal the single-GPU/oracle result and the merged top-10 the reference golden."""
    import os
    import threading
    import torch
    from plan_amd import dist as pd, pipelines
    N = 2
    L, Od, C = sf1["lineitem"], sf1["orders"], sf1["customer"]
    no, nc = len(Od["o_orderkey"]), len(C["c_custkey"])
    # lineitem rows of an order range: order keys are ascending in both tables
    def shard(r):
        o0, o1 = r * no // N, (r + 1) * no // N
        k0 = Od["o_orderkey"][o0]
        k1 = Od["o_orderkey"][o1] if o1 < no else np.iinfo(np.int64).max
        l0, l1 = np.searchsorted(L["l_orderkey"], [k0, k1])
        c0, c1 = r * nc // N, (r + 1) * nc // N
        return ({k: v[l0:l1] for k, v in L.items()}, {k: v[o0:o1] for k, v in Od.items()},
                {k: v[c0:c1] for k, v in C.items()})
    grp = pd.ThreadGroup(N)
    results, local, plain, errors = [None] * N, [None] * N, [None] * N, []

    def run(r):
        try:
            grp.bind(r)
            torch.cuda.set_device(0)
            c = hip.Ctx(0)
            Ls, Os, Cs = shard(r)
            p = pipelines.Q3Pipeline(c, Ls, Os, Cs)
            assert p.copartitioned                      # the statistic the partition-wise plan rests on
            p.allow_partitionwise = False               # the hash-partitioned exchange plan ...
            p.semijoin_reduce = False                   # ... sending every lineitem row the date filter keeps
            plain[r] = p.run(want_groups=True)
            p.semijoin_reduce = True                    # ... and with the semi-join reduction in front of the exchange
            results[r] = p.run(want_groups=True)
            p.allow_partitionwise = True                # the partition-wise join over the same shards
            local[r] = p.run(want_groups=True)
            p.free()
            c.close()
        except Exception as e:   # noqa: BLE001 - surface any rank's failure in the main thread
            errors.append(e)
            grp.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    n, rows = O.q3(sf1, "HOUSEHOLD", tpchgen.days(1995, 3, 29))
    want = {(rows[i].l_orderkey, rows[i].revenue.unscaled(4), rows[i].o_orderdate, rows[i].o_shippriority)
            for i in range(n)}
    got0, got1 = set(results[0]["groups"]), set(results[1]["groups"])
    assert not (got0 & got1)                     # groups are disjoint across ranks
    assert got0 | got1 == want
    assert min(len(got0), len(got1)) > n // 4    # and reasonably balanced
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "plan_q3.txt")).read()
    assert pipelines.q3_text(results[0]["top"]) == golden == pipelines.q3_text(results[1]["top"])
    assert results[0]["timings"]["exchange_bytes_sent"] > 0
    # the semi-join reduction changes what travels, not the result: far fewer lineitem rows are exchanged
    assert [set(x["groups"]) for x in plain] == [set(x["groups"]) for x in results]
    assert pipelines.q3_text(plain[0]["top"]) == golden
    assert results[0]["timings"]["exchange_bytes_sent"] * 20 < plain[0]["timings"]["exchange_bytes_sent"]
    # partition-wise: the same groups in all (each rank keeps the orders of its own key range), the same
    # top 10 on every rank, and no row exchange
    assert set(local[0]["groups"]) | set(local[1]["groups"]) == want and not (set(local[0]["groups"]) & set(local[1]["groups"]))
    assert pipelines.q3_text(local[0]["top"]) == golden == pipelines.q3_text(local[1]["top"])
    assert "exchange_bytes_sent" not in local[0]["timings"]


def test_q9_pipeline_sf1_matches_reference_golden(ctx, sf1):
    """Q9: LIKE filter + four hash joins (one on a composite key) + profit expression + group by
    (nation, year) on the device; the 175 rows must equal cases/tpch/1g/plan/q9.txt."""
    import os
    from plan_amd import pipelines
    p = pipelines.Q9Pipeline(ctx, sf1["lineitem"], sf1["orders"], sf1["part"], sf1["partsupp"], sf1["supplier"])
    r = p.run()
    # the measured form: no per-stage syncs, host round trips pipelined (asynchronous counts), twice in a row
    p.time_stages = False
    r2 = [p.run(), p.run()]
    p.free()
    assert r["ngroups"] == 175
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "plan_q9.txt")).read()
    assert pipelines.q9_text(r["rows"], tpchgen.nation_names()) == golden
    for rr in r2:
        assert pipelines.q9_text(rr["rows"], tpchgen.nation_names()) == golden and rr["join_rows"] == r["join_rows"]
    n, rows = O.q9(sf1, "%pink%")
    want = {(rows[i].nationkey, rows[i].o_year): rows[i].sum_profit.unscaled(4) for i in range(n)}
    assert {(a, b): c for a, b, c in r["rows"]} == want


def test_q9_pipeline_with_dangling_foreign_keys_falls_back_to_counted_lookups(ctx, sf001):
    """Q9's strict lookups assume every lineitem row finds its supplier / partsupp / order. With
    lineitem rows whose l_suppkey or l_orderkey point nowhere, the deferred PH_ECONSTRAINT arrives with the
    final download, the pipeline runs again with counted lookups that drop those rows (inner-join
    semantics), and the groups equal the oracle's on the same damaged tables — in the stage-timed
    form and in the pipelined one."""
    from plan_amd import pipelines
    t = dict(sf001)
    L = {k: v.copy() for k, v in sf001["lineitem"].items()}
    bad_s = (np.arange(len(L["l_orderkey"])) % 37) == 0
    L["l_suppkey"][bad_s] = 10**6                                 # no such supplier (and no such partsupp row)
    bad_o = (np.arange(len(L["l_orderkey"])) % 41) == 0
    L["l_orderkey"][bad_o] = L["l_orderkey"][bad_o] + 9           # keys 9, 10, .. of a group of 32 are never used
    t["lineitem"] = L
    n, rows = O.q9(t, "%pink%")
    want = {(rows[i].nationkey, rows[i].o_year): rows[i].sum_profit.unscaled(4) for i in range(n)}
    assert 0 < len(want) <= 175
    for timed in (True, False):
        p = pipelines.Q9Pipeline(ctx, t["lineitem"], t["orders"], t["part"], t["partsupp"], t["supplier"])
        p.time_stages = timed
        r = p.run()
        r2 = p.run()
        p.free()
        assert {(a, b): c for a, b, c in r["rows"]} == want == {(a, b): c for a, b, c in r2["rows"]}


def test_generic_scan_plan_runs_unfused_shapes(ctx, sf001):
    """Descriptors outside the two fused shapes run as the operator chain on the device
    (ph_scan_plan_kind == "generic") and must agree with the oracle's group-by."""
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    t = queries.lineitem_table(ctx, L, keys=True)
    # group by l_suppkey (INTEGER, ~100 groups), filters incl. '!=' and a dictionary-string '=',
    # aggregates incl. MIN/MAX and an expression with two column factors in one term
    preds = [hip.pred(queries.L_SHIPDATE, hip.PH_GE, hip.const(hip.PH_DATE, i=tpchgen.days(1994, 1, 1))),
             hip.pred(queries.L_QUANTITY, hip.PH_NE, hip.const(hip.PH_I32, i=25)),
             hip.pred(queries.L_RETURNFLAG, hip.PH_EQ, hip.const(hip.PH_STR, s="N"))]
    e, d, tx = hip.X_COL(queries.L_EXTENDEDPRICE), hip.X_COL(queries.L_DISCOUNT), hip.X_COL(queries.L_TAX)
    aggs = [hip.aggexpr(hip.PH_A_SUM, [e, d, hip.X_MUL, tx, hip.X_ADD]),      # ext*disc + tax  (scale 4)
            hip.aggexpr(hip.PH_A_MIN, [e]), hip.aggexpr(hip.PH_A_MAX, [e]),
            hip.aggexpr(hip.PH_A_AVG, [hip.X_COL(queries.L_QUANTITY)]), hip.aggexpr(hip.PH_A_COUNT_STAR)]
    p = hip.ScanPlan(ctx, t, preds, [queries.L_SUPPKEY], aggs)
    assert p.kind == "generic"
    p.run()
    r = p.fetch()
    assert r["scale"][:3] == [4, 2, 2]
    # oracle: same filters, expression, group-by
    sel = O.select(O.col(O.OT_DATE, L["l_shipdate"]), O.OP_GE, O.const(O.OT_DATE, i=tpchgen.days(1994, 1, 1)), n=n)
    sel = O.select(O.col(O.OT_INT32, L["l_quantity"]), O.OP_NE, O.const(O.OT_INT32, i=25), sel)
    sel = O.select(O.col(O.OT_CODE8, L["l_returnflag"], dictionary=O.cdict(O.RF)), O.OP_EQ, O.const(O.OT_VARCHAR, s="N"), sel)
    cols = [O.col(O.OT_DECIMAL, L["l_extendedprice"], 2), O.col(O.OT_DECIMAL, L["l_discount"], 2), O.col(O.OT_DECIMAL, L["l_tax"], 2)]
    rc, v = O.eval_decimal(cols, [(O.OX_COL, 0, 0, 0), (O.OX_COL, 1, 0, 0), (O.OX_MUL, 0, 0, 0), (O.OX_COL, 2, 0, 0), (O.OX_ADD, 0, 0, 0)], None, n)
    rc2, ve = O.eval_decimal(cols, [(O.OX_COL, 0, 0, 0)], None, n)
    args = [O.col(O.OT_ODEC, v), O.col(O.OT_ODEC, ve), O.col(O.OT_INT32, L["l_quantity"])]
    ng, first, gk, gn, vals = O.groupby([O.col(O.OT_INT32, L["l_suppkey"])], args,
                                        [(O.OA_SUM, 0), (O.OA_MIN, 1), (O.OA_MAX, 1), (O.OA_AVG, 2), (O.OA_COUNT, -1)], sel, len(sel), 4096)
    assert r["ngroups"] == ng > 50
    got = {int(r["keys"][g][0]): g for g in range(ng)}
    for og in range(ng):
        g = got[int(gk[og][0])]
        assert int(r["first_row"][g]) == int(first[og])
        assert r["sum"][g][0] == vals[og * 5 + 0].d.unscaled(4)
        assert r["sum"][g][1] == vals[og * 5 + 1].d.unscaled(2) and r["sum"][g][2] == vals[og * 5 + 2].d.unscaled(2)
        assert abs(r["sum"][g][3] / r["count"][g][3] - vals[og * 5 + 3].f) <= 1e-9 * vals[og * 5 + 3].f
        assert r["count"][g][4] == vals[og * 5 + 4].h.value()
    # ungrouped generic plan over a row range that selects nothing
    p2 = hip.ScanPlan(ctx, t, [hip.pred(queries.L_QUANTITY, hip.PH_NE, hip.const(hip.PH_I32, i=7))], [],
                      [hip.aggexpr(hip.PH_A_MAX, [e])])
    assert p2.kind == "jit"        # `!=` + MAX over NULL-free columns: a generated kernel since round 2
    p2.run(0, 4096)
    r2 = p2.fetch()
    m = L["l_quantity"][:4096] != 7
    assert r2["ngroups"] == 1 and r2["sum"][0][0] == int(L["l_extendedprice"][:4096][m].max())
    p.free(); p2.free(); t.free()


def test_q9_partitioned_two_ranks_on_one_gpu(sf1):
    """Q9's N>1 form: pink part keys, their partsupp rows and supplier are broadcast, lineitem x
    orders is hash-partitioned by order key and exchanged; 2 ranks as 2 threads on this GPU, every
    table sharded by row ranges. The merged 175 rows must equal the reference golden."""
    import os
    import threading
    import torch
    from plan_amd import dist as pd, pipelines
    N = 2
    L, Od, P, PS, S = sf1["lineitem"], sf1["orders"], sf1["part"], sf1["partsupp"], sf1["supplier"]
    no, npart, ns = len(Od["o_orderkey"]), len(P["p_partkey"]), len(S["s_suppkey"])

    def shard(r):
        o0, o1 = r * no // N, (r + 1) * no // N
        k0 = Od["o_orderkey"][o0]
        k1 = Od["o_orderkey"][o1] if o1 < no else np.iinfo(np.int64).max
        l0, l1 = np.searchsorted(L["l_orderkey"], [k0, k1])
        p0, p1 = r * npart // N, (r + 1) * npart // N
        s0, s1 = r * ns // N, (r + 1) * ns // N
        off = P["p_name_off"][p0:p1 + 1]
        Pr = {"p_partkey": P["p_partkey"][p0:p1], "p_name_off": (off - off[0]).astype(np.int32),
              "p_name_bytes": P["p_name_bytes"][off[0]:off[-1]]}
        return ({k: v[l0:l1] for k, v in L.items()}, {k: v[o0:o1] for k, v in Od.items()}, Pr,
                {k: v[4 * p0:4 * p1] for k, v in PS.items()}, {k: v[s0:s1] for k, v in S.items()})
    grp = pd.ThreadGroup(N)
    results, local, errors = [None] * N, [None] * N, []

    def run(r):
        try:
            grp.bind(r)
            torch.cuda.set_device(0)
            c = hip.Ctx(0)
            p = pipelines.Q9Pipeline(c, *shard(r))
            assert p.copartitioned
            p.allow_partitionwise = False      # the order-key stage hash-partitioned and exchanged
            results[r] = p.run()
            p.allow_partitionwise = True       # ... or rank-local (the shards are co-partitioned by order key)
            local[r] = p.run()
            p.free()
            c.close()
        except Exception as e:   # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
            grp.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, "\n".join(errors)
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "plan_q9.txt")).read()
    for r in range(N):
        assert results[r]["ngroups"] == 175
        assert pipelines.q9_text(results[r]["rows"], tpchgen.nation_names()) == golden
    assert results[0]["timings"]["exchange_bytes_sent"] > 0
    assert results[0]["join_rows"] + results[1]["join_rows"] > 300000   # both ranks did real work
    for r in range(N):   # the partition-wise plan: the same 175 rows, nothing exchanged for the orders join
        assert pipelines.q9_text(local[r]["rows"], tpchgen.nation_names()) == golden
        assert "exchange_bytes_sent" not in local[r]["timings"]


def test_fused_plan_partials_merge_across_shards(ctx, sf001):
    """The multi-GPU merge of fused plans: three row-range shards (three 'ranks') run Q1 and Q6
    separately, their raw device partials are concatenated rank-major and ph_scan_plan_fetch_merged
    must give the whole table's result, groups in global first-seen order."""
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    cuts = [0, 20000, 41000, n]
    for make, check_full in ((queries.q1_plan, True), (queries.q6_plan, False)):
        words, plans, tabs = [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            t = queries.lineitem_table(ctx, {k: v[a:b] for k, v in L.items()})
            p = make(ctx, t)
            p.run()
            ptr, nw = p.partials_dev()
            words.append(ctx.download(hip.vp(ptr), np.uint64, nw))
            plans.append(p); tabs.append(t)
        merged = plans[0].fetch_merged(np.concatenate(words), 3)
        tw = queries.lineitem_table(ctx, L)
        pw = make(ctx, tw)
        pw.run()
        whole = pw.fetch()
        assert merged["ngroups"] == whole["ngroups"]
        assert merged["keys"].tolist() == whole["keys"].tolist()      # same first-seen order
        assert merged["sum"] == whole["sum"] and merged["count"] == whole["count"]
        for x in plans + [pw]:
            x.free()
        for x in tabs + [tw]:
            x.free()
