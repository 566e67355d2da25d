"""More operator shapes: DISTINCT aggregates, OR / IN lists, CASE.

DISTINCT aggregates: the reference keeps, per DISTINCT aggregate, a table grouped
by (group keys + argument) and re-sinks its rows into the main table with filter {i}
(aggregate_exec.go:74-99, 201-304; AddChunk's filter, aggregate_hash.go:155-199). The oracle
composition below follows that; numpy's np.unique is the independent check; the device path is the
same composition over ph_agg_sink_masked + ph_agg_keys_dev."""
import numpy as np
import pytest

import oracle_lib as O


def make_data(n=50000, seed=3):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 37, n).astype(np.int32)             # group key
    x = rng.integers(0, 400, n).astype(np.int32)            # distinct argument (many repeats)
    xv = rng.random(n) > 0.1                                # 10 % NULL arguments
    y = rng.integers(-5000, 5000, n).astype(np.int32)       # ordinary SUM argument
    gv = rng.random(n) > 0.02                               # a few NULL group keys (their own group)
    return g, gv, x, xv, y


def numpy_answer(g, gv, x, xv, y):
    """{group or None: (count(distinct x), sum(distinct x), sum(y), count(*))}"""
    out = {}
    keys = np.where(gv, g, -1)
    for k in np.unique(keys):
        m = keys == k
        d = np.unique(x[m & xv])
        out[None if k == -1 else int(k)] = (len(d), int(d.sum()), int(y[m].sum()), int(m.sum()))
    return out


def oracle_answer(g, gv, x, xv, y):
    # main table: count(distinct x), sum(distinct x), sum(y), count(*); raw rows feed aggregates 2,3
    aggs = [(O.OA_COUNT, 0), (O.OA_SUM, 0), (O.OA_SUM, 1), (O.OA_COUNT, -1)]
    main = O.Agg([(O.OT_INT32, 0)], [(O.OT_INT32, 0), (O.OT_INT32, 0)], aggs)
    dist = O.Agg([(O.OT_INT32, 0), (O.OT_INT32, 0)], [], [])          # (group key, x), no aggregates
    main.sink([(g, gv)], [(x, xv), (y, None)], mask=0b1100)
    dist.sink([(g, gv), (x, xv)], [])
    rows = dist.groups()
    dg = np.array([0 if r[1][0] is None else r[1][0] for r in rows], np.int32)
    dgv = np.array([r[1][0] is not None for r in rows])
    dx = np.array([0 if r[1][1] is None else r[1][1] for r in rows], np.int32)
    dxv = np.array([r[1][1] is not None for r in rows])
    main.sink([(dg, dgv)], [(dx, dxv), (dx, None)], mask=0b0011)
    out = {}
    for first, keys, vals in main.groups():
        out[keys[0]] = (vals[0].h.value(), vals[1].h.value() if vals[1].kind != O.OV_NULL else 0,
                        vals[2].h.value() if vals[2].kind != O.OV_NULL else 0, vals[3].h.value())
    return out


def test_oracle_filtered_sink_distinct_composition_matches_numpy():
    d = make_data()
    assert oracle_answer(*d) == numpy_answer(*d)


@pytest.mark.gpu
def test_device_distinct_aggregates_match_oracle():
    from plan_amd import hip
    ctx = hip.Ctx(0)
    g, gv, x, xv, y = make_data(n=300000, seed=8)
    want = oracle_answer(g, gv, x, xv, y)
    n = len(g)
    vb = lambda v: np.packbits(v, bitorder="little")
    dg = hip.DevColumn(ctx, hip.PH_I32, g, validity=vb(gv))
    dx = hip.DevColumn(ctx, hip.PH_I32, x, validity=vb(xv))
    dy = hip.DevColumn(ctx, hip.PH_I32, y)
    main = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_COUNT, 0), (hip.PH_A_SUM, 0), (hip.PH_A_SUM, 1), (hip.PH_A_COUNT_STAR, -1)], 64)
    dist = hip.Agg(ctx, [hip.PH_I32, hip.PH_I32], [], 1024)
    main.sink([dg], [dx, dy], None, n, mask=0b1100)
    dist.sink([dg, dx], [], None, n)
    p0, v0, nd, c0 = dist.key_column(0, hip.PH_I32)
    p1, v1, nd1, c1 = dist.key_column(1, hip.PH_I32)
    assert nd == nd1 == len({(None if not a else int(b), None if not c else int(e)) for a, b, c, e in zip(gv, g, xv, x)})
    main.sink([c0], [c1, c1], None, nd, mask=0b0011)
    r = main.finalize()
    got = {}
    for i in range(r["ngroups"]):
        k = None if r["key_null"][i][0] else int(r["keys"][i][0])
        got[k] = (int(r["count"][i][0]), r["sum"][i][1], r["sum"][i][2], int(r["count"][i][3]))
    assert got == want
    for p in (p0, v0, p1, v1):
        ctx.free(p)
    main.free(); dist.free()
    for d in (dg, dx, dy):
        d.free()
    ctx.close()


# ------------------------------------------------------------------ OR / IN lists

def in_list_data(n=100000, seed=21):
    rng = np.random.default_rng(seed)
    size = rng.integers(1, 51, n).astype(np.int32)
    sv = rng.random(n) > 0.05
    mode = rng.integers(0, 7, n).astype(np.uint8)   # dictionary codes of l_shipmode-like strings
    return size, sv, mode


MODES = ["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"]


def test_oracle_select_or_is_the_union_in_child_major_order():
    size, sv, mode = in_list_data()
    n = len(size)
    c = O.col(O.OT_INT32, size, validity=np.packbits(sv, bitorder="little"))
    kids = [(c, O.OP_EQ, O.const(O.OT_INT32, i=v)) for v in (49, 14, 23, 14)]
    got = O.select_or(kids, n=n)
    want = np.concatenate([np.flatnonzero(sv & (size == v)) for v in (49, 14, 23)])   # the repeated 14 adds nothing
    assert np.array_equal(got, want)
    # `in` on BIGINT is not implemented by the reference: selects nothing
    big = O.col(O.OT_INT64, size.astype(np.int64))
    assert len(O.select_or([(big, O.OP_EQ, O.const(O.OT_INT64, i=14))], n=n)) == 0


@pytest.mark.gpu
def test_device_in_list_union_matches_oracle():
    from plan_amd import hip
    ctx = hip.Ctx(0)
    size, sv, mode = in_list_data(n=1_000_003, seed=4)
    n = len(size)
    vb = np.packbits(sv, bitorder="little")
    dsize = hip.DevColumn(ctx, hip.PH_I32, size, validity=vb)
    osize = O.col(O.OT_INT32, size, validity=vb)
    dmode = hip.DevColumn(ctx, hip.PH_CODE8, mode)
    omode = O.col(O.OT_CODE8, mode, dictionary=O.cdict(MODES))
    # p_size IN (49, 14, 23, 45, 19, 3, 36, 9) (Q16), l_shipmode IN ('MAIL', 'SHIP') (Q12), and a mix
    cases = [[(dsize, osize, hip.PH_I32, O.OT_INT32, v) for v in (49, 14, 23, 45, 19, 3, 36, 9)],
             [(dmode, omode, "code", O.OT_VARCHAR, s) for s in ("MAIL", "SHIP")],
             [(dsize, osize, hip.PH_I32, O.OT_INT32, 7), (dmode, omode, "code", O.OT_VARCHAR, "TRUCK"),
              (dsize, osize, hip.PH_I32, O.OT_INT32, 1000)]]
    for case in cases:
        sels, counts, kids = [], [], []
        for dcol, ocol, dt, ot, v in case:
            if dt == "code":
                k = hip.const(hip.PH_I32, i=MODES.index(v))
                kids.append((ocol, O.OP_EQ, O.const(O.OT_VARCHAR, s=v)))
            else:
                k = hip.const(dt, i=v)
                kids.append((ocol, O.OP_EQ, O.const(ot, i=v)))
            s, c = hip.filter_select(ctx, dcol, n, hip.PH_EQ, k)
            sels.append(s); counts.append(c)
        out, m = hip.sel_union(ctx, sels, counts, n)
        want = np.sort(O.select_or(kids, n=n))
        got = ctx.download(out, np.int32, m).astype(np.int64)
        assert m == len(want) and np.array_equal(got, want)
        for s in sels + [out]:
            ctx.free(s)
    # no children / nothing selected
    out, m = hip.sel_union(ctx, [], [], n)
    assert m == 0
    ctx.free(out)
    dsize.free(); dmode.free()
    ctx.close()


# ------------------------------------------------------------------ CASE

@pytest.mark.gpu
def test_device_case_when_matches_oracle():
    """sum(case when l_shipdate < D then ext * (1 - disc) else 0.0000 end)-style CASE (Q14/Q12/Q8):
    WHEN -> ph_filter_select, THEN/ELSE -> ph_expr_eval over the true / false rows, results filled
    back with ph_scatter (executeCase + FillSwitch, expr_exec.go:144-246, 559-606)."""
    from plan_amd import hip
    ctx = hip.Ctx(0)
    rng = np.random.default_rng(17)
    n = 200_000
    ext = rng.integers(90000, 10500000, n).astype(np.int64)
    disc = rng.integers(0, 11, n).astype(np.int64)
    ship = rng.integers(8000, 10600, n).astype(np.int32)
    ev = rng.random(n) > 0.03           # NULL prices
    evb = np.packbits(ev, bitorder="little")
    D = 9500
    d_ext = hip.DevColumn(ctx, hip.PH_DEC64, ext, 2, validity=evb)
    d_disc = hip.DevColumn(ctx, hip.PH_DEC64, disc, 2)
    d_ship = hip.DevColumn(ctx, hip.PH_DATE, ship)
    ocols = [O.col(O.OT_DECIMAL, ext, 2, validity=evb), O.col(O.OT_DECIMAL, disc, 2)]
    then_o = [(O.OX_COL, 0, 0, 0), (O.OX_CONST_INT, 0, 1, 0), (O.OX_COL, 1, 0, 0), (O.OX_SUB, 0, 0, 0), (O.OX_MUL, 0, 0, 0)]
    else_o = [(O.OX_CONST_DEC, 0, 0, 4)]
    rc, want, wnull = O.case_decimal(ocols, O.col(O.OT_DATE, ship), O.OP_LT, O.const(O.OT_DATE, i=D), then_o, else_o, n)
    assert rc == 0
    # device
    tsel, tn = hip.filter_select(ctx, d_ship, n, hip.PH_LT, hip.const(hip.PH_DATE, i=D))
    fsel, fn = hip.sel_difference(ctx, None, n, tsel, tn, n)
    assert tn + fn == n and tn == int((ship < D).sum())
    then_p = [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]
    else_p = [hip.X_CONST(0, 4)]
    out = ctx.alloc(n * 8)
    oval = ctx.upload(np.zeros((n + 31) // 32 * 4, np.uint8))
    for sel, cnt, prog in ((tsel, tn, then_p), (fsel, fn, else_p)):
        v, vv = hip.expr_eval(ctx, [d_ext, d_disc], prog, sel, cnt, want_validity=True)
        c = hip.Col()
        c.type, c.scale, c.data, c.validity = hip.PH_DEC64, 4, v, vv
        hip.scatter(ctx, c, sel, cnt, out, oval)
        ctx.free(v); ctx.free(vv)
    got = ctx.download(out, np.int64, n)
    gvalid = np.unpackbits(ctx.download(oval, np.uint8, (n + 31) // 32 * 4), bitorder="little")[:n].astype(bool)
    assert np.array_equal(~gvalid, wnull.astype(bool))
    assert np.array_equal(gvalid, ev | (ship >= D))        # NULL only where THEN read a NULL price
    wu = np.array(O.odec_unscaled(want, 4), dtype=np.int64)
    assert np.array_equal(got[gvalid], wu[gvalid])
    for p in (tsel, fsel, out, oval):
        ctx.free(p)
    for d in (d_ext, d_disc, d_ship):
        d.free()
    ctx.close()


# ------------------------------------------------------------------ fused Filter -> probe

@pytest.mark.gpu
def test_fused_filter_probe_equals_filter_then_probe(sf001):
    """ph_join_probe_inner_where = ph_filter_select + ph_join_probe_inner (same pairs, same order),
    and refuses shapes it does not fuse."""
    from plan_amd import hip, tpchgen
    ctx = hip.Ctx(0)
    Od, L = sf001["orders"], sf001["lineitem"]
    no, nl = len(Od["o_orderkey"]), len(L["l_orderkey"])
    okey = hip.DevColumn(ctx, hip.PH_I64, Od["o_orderkey"])
    lkey = hip.DevColumn(ctx, hip.PH_I64, L["l_orderkey"])
    ship = hip.DevColumn(ctx, hip.PH_DATE, L["l_shipdate"])
    disc = hip.DevColumn(ctx, hip.PH_DEC64, L["l_discount"], 2)
    bsel = ctx.upload(np.arange(0, no, 7, dtype=np.int32))          # every 7th order is built
    j = hip.Join(ctx, [okey], bsel, (no + 6) // 7)
    for wcol, op, k in ((ship, hip.PH_GT, hip.const(hip.PH_DATE, i=tpchgen.days(1995, 3, 15))),
                        (ship, hip.PH_LE, hip.const(hip.PH_DATE, i=tpchgen.days(1992, 1, 1))),     # nothing passes
                        (disc, hip.PH_EQ, hip.const(hip.PH_DEC64, i=5, scale=2))):
        s, c = hip.filter_select(ctx, wcol, nl, op, k)
        m0, p0, b0 = j.probe_inner([lkey], s, c, max(c, 1))
        got = j.probe_inner_where([lkey], wcol, op, k, None, nl, nl)
        assert got is not None
        m1, p1, b1 = got
        assert m0 == m1
        assert np.array_equal(ctx.download(p0, np.int32, m0), ctx.download(p1, np.int32, m1))
        assert np.array_equal(ctx.download(b0, np.int32, m0), ctx.download(b1, np.int32, m1))
        for p in (s, p0, b0, p1, b1):
            ctx.free(p)
    # a float comparison is not an integer range: not fused
    assert j.probe_inner_where([lkey], disc, hip.PH_GE, hip.const(hip.PH_F32, f=0.05), None, nl, nl) is None
    j.free()
    ctx.free(bsel)
    for d in (okey, lkey, ship, disc):
        d.free()
    ctx.close()


# ------------------------------------------------------------------ ORDER BY

def sort_case(n, seed):
    rng = np.random.default_rng(seed)
    rev = rng.integers(-5 * 10**6, 5 * 10**6, n).astype(np.int64)       # DECIMAL(15,4): rounds to cents for the sort
    rev[rng.integers(0, n, n // 10)] = 1234549                          # 123.4549 / 123.4550 / 123.4551 around a half
    rev[rng.integers(0, n, n // 10)] = 1234550
    rev[rng.integers(0, n, n // 10)] = 1234551
    date = rng.integers(8000, 8060, n).astype(np.int32)
    prio = rng.integers(-3, 4, n).astype(np.int32)
    code = rng.integers(0, 7, n).astype(np.uint8)
    vr = rng.random(n) > 0.05
    vd = rng.random(n) > 0.05
    return rev, date, prio, code, vr, vd


def test_oracle_sort_orders_like_python():
    """the byte-comparable keys order rows like the tuple (NULL first, value rounded to cents, ...)"""
    n = 3000
    rev, date, prio, code, vr, vd = sort_case(n, 5)
    vb = lambda v: np.packbits(v, bitorder="little")
    cols = [O.col(O.OT_DECIMAL, rev, 4, validity=vb(vr)), O.col(O.OT_DATE, date, validity=vb(vd)), O.col(O.OT_INT32, prio)]
    rows, keys = O.sort_rows(cols, [True, False, True], n=n)
    from decimal import Decimal, ROUND_HALF_EVEN
    cents = [int((Decimal(int(x)) / Decimal(10000)).quantize(Decimal("0.01"), rounding=ROUND_HALF_EVEN) * 100) for x in rev]

    def pykey(r):
        return ((0, 0) if not vr[r] else (1, -cents[r]), (0, 0) if not vd[r] else (1, int(date[r])), -int(prio[r]), r)
    assert rows.tolist() == sorted(range(n), key=pykey)
    assert all(bytes(keys[i]) <= bytes(keys[i + 1]) for i in range(n - 1))


@pytest.mark.gpu
def test_device_sort_matches_oracle():
    """ph_sort_rows = the oracle's LocalSort restatement: same row order (ties in input order on
    both sides), for DECIMAL desc + DATE asc + INTEGER desc + dictionary code keys, NULLs first,
    with and without a selection, across sizes that span one tile to many workgroups."""
    from plan_amd import hip
    ctx = hip.Ctx(0)
    for n, seed in ((1, 1), (255, 2), (5000, 3), (300_000, 4)):
        rev, date, prio, code, vr, vd = sort_case(n, seed)
        vb = lambda v: np.packbits(v, bitorder="little")
        d = [hip.DevColumn(ctx, hip.PH_DEC64, rev, 4, validity=vb(vr)), hip.DevColumn(ctx, hip.PH_DATE, date, validity=vb(vd)),
             hip.DevColumn(ctx, hip.PH_I32, prio), hip.DevColumn(ctx, hip.PH_CODE8, code)]
        o = [O.col(O.OT_DECIMAL, rev, 4, validity=vb(vr)), O.col(O.OT_DATE, date, validity=vb(vd)),
             O.col(O.OT_INT32, prio), O.col(O.OT_CODE8, code)]
        for pick, desc in (([0, 1, 2], [True, False, True]), ([3, 0], [False, False]), ([1], [True]), ([2, 3, 1, 0], [False, True, True, True])):
            for sel in (None, np.sort(np.random.default_rng(seed).choice(n, max(1, n // 2), replace=False))):
                m = n if sel is None else len(sel)
                sd = None if sel is None else ctx.upload(sel.astype(np.int32))
                out = hip.sort_rows(ctx, [d[i] for i in pick], desc, sd, m)
                got = ctx.download(out, np.int32, m).astype(np.int64)
                want, _ = O.sort_rows([o[i] for i in pick], desc, sel=None if sel is None else sel.astype(np.int64), n=m)
                assert np.array_equal(got, want), (n, pick, desc, sel is not None)
                ctx.free(out)
                if sd is not None:
                    ctx.free(sd)
        for c in d:
            c.free()
    big = hip.DevColumn(ctx, hip.PH_I64, np.arange(10, dtype=np.int64))
    with pytest.raises(hip.PlanHipError) as e:      # BIGINT keys: no RadixScatter case in the reference either
        hip.sort_rows(ctx, [big], [False], None, 10)
    assert e.value.code == hip.PH_EUNSUPPORTED
    big.free()
    ctx.close()


@pytest.mark.gpu
def test_device_substring_and_cross_pairs_match_oracle():
    """ph_substring over a PH_STR column (NULL rows, a selection, every branch of substringStartEnd)
    and ph_cross_pairs against their oracle restatements. oracle_substring's positive-offset branch is pinned by Q22's golden
    (tests/test_golden_tpch.py); the other branches and the cross product follow the Go source alone (parity unpinned)."""
    from plan_amd import hip
    ctx = hip.Ctx(0)
    rng = np.random.default_rng(17)
    n = 5000
    words = [bytes(rng.integers(97, 123, rng.integers(0, 30)).astype(np.uint8)) for _ in range(n)]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(w) for w in words])
    data = np.frombuffer(b"".join(words), dtype=np.uint8)
    valid = rng.random(n) > 0.1
    col = hip.DevColumn(ctx, hip.PH_STR, off, validity=np.packbits(valid, bitorder="little"), aux=data)
    sel = np.sort(rng.choice(n, 1777, replace=False)).astype(np.int32)
    dsel = ctx.upload(sel)
    for o, ln in ((1, 2), (3, 100), (-3, 2), (0, 3), (0, 1), (4, -2), (40, 3), (-40, 3), (5, 0), (2, 2**62)):
        for s, rows in ((None, np.arange(n)), (dsel, sel)):
            po, pb, nb = hip.substring(ctx, col, o, ln, s, len(rows))
            goff = ctx.download(po, np.int32, len(rows) + 1)
            gb = ctx.download(pb, np.uint8, nb).tobytes() if nb else b""
            want = [O.substring(words[r], o, ln) if valid[r] else b"" for r in rows]
            assert goff[0] == 0 and goff[-1] == nb == sum(len(w) for w in want)
            assert [gb[goff[i]:goff[i + 1]] for i in range(len(rows))] == want
            ctx.free(po); ctx.free(pb)
    for nl, nr in ((7, 3), (2048, 2), (1, 1), (0, 5), (300, 0)):
        ol, orr = hip.cross_pairs(ctx, nl, nr)
        wl, wr = O.cross_pairs(nl, nr, chunk=max(nl, 1))     # one device batch = one left chunk
        assert np.array_equal(ctx.download(ol, np.int32, nl * nr), wl) and np.array_equal(ctx.download(orr, np.int32, nl * nr), wr)
        ctx.free(ol); ctx.free(orr)
    col.free()
    ctx.close()
