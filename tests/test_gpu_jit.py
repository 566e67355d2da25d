"""Plan-specialised fused Agg <- Scan(filter) kernels (ph_scan_plan_kind == "jit": generated from
the plan's shape and compiled with hiprtc at plan creation) against the oracle: the same parity bar
as the two precompiled fused kernels — bit-exact integer sums / counts / min / max, groups in
first-seen order — for shapes neither of them covers, and against them on their own shapes."""
import os

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


@pytest.fixture()
def force_jit():
    os.environ["PH_SCAN_JIT"] = "1"
    yield
    os.environ.pop("PH_SCAN_JIT", None)


def test_q1_q6_through_generated_kernels_equal_precompiled(ctx, sf1, force_jit):
    """PH_SCAN_JIT=1 sends the Q1 and Q6 shapes through generated kernels: identical results to the
    precompiled lowcard_chain / filter_sumprod kernels, whole table and ragged row ranges."""
    L = sf1["lineitem"]
    n = len(L["l_shipdate"])
    t = queries.lineitem_table(ctx, L)
    for make in (queries.q1_plan, queries.q6_plan):
        pj = make(ctx, t)
        assert pj.kind == "jit"
        os.environ["PH_SCAN_JIT"] = "0"
        pp = make(ctx, t)
        os.environ["PH_SCAN_JIT"] = "1"
        assert pp.kind in ("lowcard_chain", "filter_sumprod")
        for a, b in ((0, n), (4 * 1000, n - 12345), (0, 8), (n // 4 * 4 - 4, n), (1024, 1024)):
            pj.run(a, b); pp.run(a, b)
            rj, rp = pj.fetch(), pp.fetch()
            assert rj["ngroups"] == rp["ngroups"] and rj["keys"].tolist() == rp["keys"].tolist()
            assert rj["sum"] == rp["sum"] and rj["count"] == rp["count"] and rj["scale"] == rp["scale"]
            if pp.kind == "lowcard_chain":   # filter_sumprod keeps no first-row word (one group: row 0)
                assert rj["first_row"].tolist() == rp["first_row"].tolist()
        pj.free(); pp.free()
    t.free()


def _table(ctx, L, extra_code):
    n = len(L["l_shipdate"])
    cols = [(hip.PH_I32, L["l_quantity"]), (hip.PH_DEC64, L["l_extendedprice"], 2), (hip.PH_DEC64, L["l_discount"], 2),
            (hip.PH_DEC64, L["l_tax"], 2), (hip.PH_CODE8, L["l_returnflag"], 0, None, tpchgen.RETURNFLAG_DICT),
            (hip.PH_CODE8, L["l_linestatus"], 0, None, tpchgen.LINESTATUS_DICT), (hip.PH_DATE, L["l_shipdate"]),
            (hip.PH_CODE8, extra_code, 0, None, ["m0", "m1", "m2", "m3", "m4"]), (hip.PH_I64, L["l_orderkey"])]
    return hip.Table(ctx, cols, n)


def _oracle_groups(L, extra, sel, key_cols, args, aggs):
    keys = []
    for k in key_cols:
        if k == "rf":
            keys.append(O.col(O.OT_CODE8, L["l_returnflag"], dictionary=O.cdict(O.RF)))
        elif k == "ls":
            keys.append(O.col(O.OT_CODE8, L["l_linestatus"], dictionary=O.cdict(O.LS)))
        else:
            keys.append(O.col(O.OT_INT32, extra.astype(np.int32)))
    if not keys:
        keys = [O.col(O.OT_INT32, np.zeros(len(L["l_shipdate"]), np.int32))]
    return O.groupby(keys, args, aggs, sel, len(sel), 4096)


def test_generated_kernels_match_oracle_on_new_shapes(ctx, sf001):
    """Shapes outside both precompiled kernels: three group columns, MIN/MAX, `!=`, a float-literal
    compare on a decimal column, a four-factor product, an ungrouped plan with two accumulators,
    COUNT(*) alone — each against the oracle's select / expression / group-by."""
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    extra = (L["l_quantity"] % 5).astype(np.uint8)
    t = _table(ctx, L, extra)
    Q, E, D, T, RF, LS, SD, MODE, OK = range(9)
    e, d, tx, q = hip.X_COL(E), hip.X_COL(D), hip.X_COL(T), hip.X_COL(Q)
    one = hip.X_CONST(1, 0)
    ocols = [O.col(O.OT_DECIMAL, L["l_extendedprice"], 2), O.col(O.OT_DECIMAL, L["l_discount"], 2),
             O.col(O.OT_DECIMAL, L["l_tax"], 2), O.col(O.OT_INT32, L["l_quantity"])]
    oe, od, ot, oq = (O.OX_COL, 0, 0, 0), (O.OX_COL, 1, 0, 0), (O.OX_COL, 2, 0, 0), (O.OX_COL, 3, 0, 0)
    MUL, ADD, SUB = (O.OX_MUL, 0, 0, 0), (O.OX_ADD, 0, 0, 0), (O.OX_SUB, 0, 0, 0)
    oone = (O.OX_CONST_INT, 0, 1, 0)

    def osel(preds):
        sel = None
        for col, op, k in preds:
            sel = O.select(col, op, k, sel, n=n)
        return sel if sel is not None else np.arange(n, dtype=np.int64)

    cases = []
    # (a) three group columns, MIN/MAX + sums, `!=` and a date range
    cases.append(dict(
        preds=[hip.pred(SD, hip.PH_GE, hip.const(hip.PH_DATE, i=tpchgen.days(1993, 6, 1))),
               hip.pred(SD, hip.PH_LT, hip.const(hip.PH_DATE, i=tpchgen.days(1997, 1, 1))),
               hip.pred(Q, hip.PH_NE, hip.const(hip.PH_I32, i=25))],
        opreds=[(O.col(O.OT_DATE, L["l_shipdate"]), O.OP_GE, O.const(O.OT_DATE, i=tpchgen.days(1993, 6, 1))),
                (O.col(O.OT_DATE, L["l_shipdate"]), O.OP_LT, O.const(O.OT_DATE, i=tpchgen.days(1997, 1, 1))),
                (O.col(O.OT_INT32, L["l_quantity"]), O.OP_NE, O.const(O.OT_INT32, i=25))],
        groups=[RF, LS, MODE], okeys=["rf", "ls", "mode"],
        aggs=[hip.aggexpr(hip.PH_A_MIN, [e]), hip.aggexpr(hip.PH_A_MAX, [e]), hip.aggexpr(hip.PH_A_SUM, [e, d, hip.X_MUL]),
              hip.aggexpr(hip.PH_A_AVG, [q]), hip.aggexpr(hip.PH_A_COUNT_STAR)],
        oprogs=[[oe], [oe], [oe, od, MUL], None, None], oargs_int=[None, None, None, "qty", None],
        oaggs=[O.OA_MIN, O.OA_MAX, O.OA_SUM, O.OA_AVG, O.OA_COUNT], scales=[2, 2, 4, 0, 0]))
    # (b) one group column, a four-factor product ext*(1-disc)*(1+tax)*qty, float-literal compare
    cases.append(dict(
        preds=[hip.pred(D, hip.PH_GE, hip.const(hip.PH_F32, f=float(np.float32(0.05))))],
        opreds=[(O.col(O.OT_DECIMAL, L["l_discount"], 2), O.OP_GE, O.const(O.OT_FLOAT, f=float(np.float32(0.05))))],
        groups=[LS], okeys=["ls"],
        aggs=[hip.aggexpr(hip.PH_A_SUM, [e, one, d, hip.X_SUB, hip.X_MUL, one, tx, hip.X_ADD, hip.X_MUL, q, hip.X_MUL]),
              hip.aggexpr(hip.PH_A_SUM, [tx])],
        oprogs=[[oe, oone, od, SUB, MUL, oone, ot, ADD, MUL, oq, MUL], [ot]], oargs_int=[None, None],
        oaggs=[O.OA_SUM, O.OA_SUM], scales=[6, 2]))
    # (c) ungrouped, two accumulators + count
    cases.append(dict(
        preds=[hip.pred(Q, hip.PH_LT, hip.const(hip.PH_I32, i=10))],
        opreds=[(O.col(O.OT_INT32, L["l_quantity"]), O.OP_LT, O.const(O.OT_INT32, i=10))],
        groups=[], okeys=[],
        aggs=[hip.aggexpr(hip.PH_A_SUM, [e, tx, hip.X_MUL]), hip.aggexpr(hip.PH_A_MAX, [d]), hip.aggexpr(hip.PH_A_COUNT_STAR)],
        oprogs=[[oe, ot, MUL], [od], None], oargs_int=[None, None, None],
        oaggs=[O.OA_SUM, O.OA_MAX, O.OA_COUNT], scales=[4, 2, 0]))
    for case in cases:
        p = hip.ScanPlan(ctx, t, case["preds"], case["groups"], case["aggs"])
        assert p.kind == "jit", hip.lib().ph_last_error()
        p.run()
        r = p.fetch()
        sel = osel(case["opreds"])
        args, specs = [], []
        for i, (prog, asint, kind) in enumerate(zip(case["oprogs"], case["oargs_int"], case["oaggs"])):
            if prog is not None:
                rc, v = O.eval_decimal(ocols, prog, None, n)
                assert rc == 0
                args.append(O.col(O.OT_ODEC, v))
                specs.append((kind, len(args) - 1))
            elif asint == "qty":
                args.append(O.col(O.OT_INT32, L["l_quantity"]))
                specs.append((kind, len(args) - 1))
            else:
                specs.append((kind, -1))
        ng, first, gk, gn, vals = _oracle_groups(L, extra, sel, case["okeys"], args, specs)
        na = len(specs)
        assert r["ngroups"] == ng and r["scale"] == case["scales"]
        for g in range(ng):   # same first-seen order
            assert int(r["first_row"][g]) == int(first[g])
            if case["okeys"]:
                assert [int(x) for x in r["keys"][g]] == [int(x) for x in gk[g][:len(case["okeys"])]]
            for a, (kind, _) in enumerate(specs):
                v = vals[g * na + a]
                if kind == O.OA_COUNT:
                    assert r["count"][g][a] == v.h.value()
                elif kind == O.OA_AVG:
                    assert abs(r["sum"][g][a] / r["count"][g][a] - v.f) <= 1e-9 * abs(v.f)
                else:
                    assert r["sum"][g][a] == v.d.unscaled(case["scales"][a]), (a, g)
        p.free()
    # COUNT(*) alone over a predicate, empty selections and a selects-nothing range
    p = hip.ScanPlan(ctx, t, [hip.pred(Q, hip.PH_GE, hip.const(hip.PH_I32, i=49))], [], [hip.aggexpr(hip.PH_A_COUNT_STAR)])
    assert p.kind == "jit"
    p.run()
    assert p.fetch()["count"][0][0] == int((L["l_quantity"] >= 49).sum())
    p.run(0, 0)
    assert p.fetch()["ngroups"] == 0
    p.free()
    p = hip.ScanPlan(ctx, t, [hip.pred(Q, hip.PH_GT, hip.const(hip.PH_I32, i=50))], [RF], [hip.aggexpr(hip.PH_A_SUM, [e])])
    p.run()
    assert p.fetch()["ngroups"] == 0
    p.free()
    # BIGINT compares do not exist in the reference's selectOperation: selects nothing, as there
    p = hip.ScanPlan(ctx, t, [hip.pred(OK, hip.PH_GT, hip.const(hip.PH_I32, i=5))], [RF], [hip.aggexpr(hip.PH_A_SUM, [e])])
    p.run()
    assert p.fetch()["ngroups"] == 0
    p.free()
    t.free()


def test_generated_kernel_partials_merge_across_shards(ctx, sf001):
    """ph_scan_plan_fetch_merged over a generated kernel's raw words: sums add, MIN/MAX take the
    min / max, first-seen rows order by rank — three shards equal the whole table."""
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    extra = (L["l_quantity"] % 5).astype(np.uint8)
    E, LS, MODE = 1, 5, 7
    aggs = [hip.aggexpr(hip.PH_A_MIN, [hip.X_COL(E)]), hip.aggexpr(hip.PH_A_MAX, [hip.X_COL(E)]),
            hip.aggexpr(hip.PH_A_SUM, [hip.X_COL(E)]), hip.aggexpr(hip.PH_A_COUNT_STAR)]
    cuts = [0, 20000, 41000, n]
    words, plans, tabs = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        t = _table(ctx, {k: v[a:b] for k, v in L.items()}, extra[a:b])
        p = hip.ScanPlan(ctx, t, [], [MODE, LS], aggs)
        assert p.kind == "jit"
        p.run()
        ptr, nw = p.partials_dev()
        words.append(ctx.download(hip.vp(ptr), np.uint64, nw))
        plans.append(p); tabs.append(t)
    merged = plans[0].fetch_merged(np.concatenate(words), 3)
    tw = _table(ctx, L, extra)
    pw = hip.ScanPlan(ctx, tw, [], [MODE, LS], aggs)
    pw.run()
    whole = pw.fetch()
    assert merged["ngroups"] == whole["ngroups"] == 10
    assert merged["keys"].tolist() == whole["keys"].tolist()
    assert merged["sum"] == whole["sum"] and merged["count"] == whole["count"]
    for x in plans + [pw]:
        x.free()
    for x in tabs + [tw]:
        x.free()


def test_many_slots_share_lds_columns_and_nullable_inputs_fall_back(ctx, sf001):
    """30 group slots x 9 accumulators do not fit as per-thread-private LDS columns: the generated
    kernel shares 16 columns per accumulator between lanes (LDS atomics keep it exact) — checked
    against numpy. A NULL-able input is not generated: the plan is "generic" (the operator chain)."""
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    extra = (L["l_quantity"] % 5).astype(np.uint8)
    t = _table(ctx, L, extra)
    E, D, T, RF, LS, MODE = 1, 2, 3, 4, 5, 7
    names = {E: "l_extendedprice", D: "l_discount", T: "l_tax"}
    aggs = [hip.aggexpr(hip.PH_A_SUM, [hip.X_COL(c)]) for c in (E, D, T)] + \
           [hip.aggexpr(hip.PH_A_MIN, [hip.X_COL(c)]) for c in (E, D, T)] + \
           [hip.aggexpr(hip.PH_A_MAX, [hip.X_COL(c)]) for c in (E, D, T)]
    p = hip.ScanPlan(ctx, t, [], [RF, LS, MODE], aggs)   # 30 slots x 9 accumulators
    assert p.kind == "jit"
    p.run()
    r = p.fetch()
    gid = (L["l_returnflag"].astype(np.int64) * 2 + L["l_linestatus"]) * 5 + extra
    assert r["ngroups"] == len(np.unique(gid))
    for g in range(r["ngroups"]):
        k = r["keys"][g]
        m = gid == (int(k[0]) * 2 + int(k[1])) * 5 + int(k[2])
        assert int(r["first_row"][g]) == int(np.nonzero(m)[0][0]) and r["count"][g][0] == int(m.sum())
        for i, c in enumerate((E, D, T)):
            v = L[names[c]][m]
            assert r["sum"][g][i] == int(v.sum()) and r["sum"][g][3 + i] == int(v.min()) and r["sum"][g][6 + i] == int(v.max())
    p.free()
    t.free()
    valid = np.ones(n, bool)
    valid[::7] = False
    tn = hip.Table(ctx, [dict(typ=hip.PH_DEC64, arr=L["l_extendedprice"], scale=2, validity=np.packbits(valid, bitorder="little")),
                         (hip.PH_CODE8, L["l_linestatus"], 0, None, tpchgen.LINESTATUS_DICT)], n)
    p = hip.ScanPlan(ctx, tn, [], [1], [hip.aggexpr(hip.PH_A_SUM, [hip.X_COL(0)])])
    assert p.kind == "generic"
    p.run()
    r = p.fetch()
    for g in range(r["ngroups"]):
        m = (L["l_linestatus"] == r["keys"][g][0]) & valid
        assert r["sum"][g][0] == int(L["l_extendedprice"][m].sum()) and r["count"][g][0] == int(m.sum())
    p.free()
    tn.free()
