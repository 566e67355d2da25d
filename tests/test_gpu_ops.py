"""GPU parity of the operator-granular kernels (filter, hash, expressions, hash aggregate, hash
join, gather, partition) against the oracle, through the C-ABI."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, tpchgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


def rnd_validity(rng, n, p_null):
    bits = rng.random(n) >= p_null
    return np.packbits(bits, bitorder="little"), bits


def dl(ctx, p, dtype, n):
    return ctx.download(p, dtype, n) if n else np.empty(0, dtype)


# ------------------------------------------------------------------ filter

SEL_CASES = [
    # (hip type, oracle type, scale, generator, [(op, hip const, oracle const)])
    ("i32", hip.PH_I32, O.OT_INT32, 0),
    ("date", hip.PH_DATE, O.OT_DATE, 0),
    ("dec_float", hip.PH_DEC64, O.OT_DECIMAL, 2),
    ("dec_dec", hip.PH_DEC64, O.OT_DECIMAL, 2),
    ("i64", hip.PH_I64, O.OT_INT64, 0),
]


@pytest.mark.parametrize("name,ht,ot,scale", SEL_CASES)
@pytest.mark.parametrize("n", [1, 63, 2048, 70001])
def test_filter_select_all_ops(ctx, name, ht, ot, scale, n):
    rng = np.random.default_rng(n * 7 + len(name))
    if name in ("i32", "date"):
        data = rng.integers(8000, 8100, n).astype(np.int32)
        hk = lambda: hip.const(ht, i=8050)
        ok = lambda: O.const(ot, i=8050)
    elif name == "dec_float":
        data = rng.integers(-2, 12, n).astype(np.int64)
        kf = float(np.float32(0.03) + np.float32(0.01))
        hk = lambda: hip.const(hip.PH_F32, f=kf)
        ok = lambda: O.const(O.OT_FLOAT, f=kf)
    elif name == "dec_dec":
        data = rng.integers(0, 1000, n).astype(np.int64)
        hk = lambda: hip.const(hip.PH_DEC64, i=5, scale=0)   # 5 == 5.00
        ok = lambda: O.const(O.OT_DECIMAL, i=500, scale=2)
    else:
        data = rng.integers(0, 100, n).astype(np.int64)
        hk = lambda: hip.const(hip.PH_I64, i=50)
        ok = lambda: O.const(O.OT_INT64, i=50)
    vbytes, _ = rnd_validity(rng, n, 0.1)
    d = hip.DevColumn(ctx, ht, data, scale, validity=vbytes)
    oc = O.col(ot, data, scale, validity=vbytes)
    for op in (hip.PH_EQ, hip.PH_NE, hip.PH_LT, hip.PH_LE, hip.PH_GT, hip.PH_GE):
        try:
            sel, cnt = hip.filter_select(ctx, d, n, op, hk())
        except hip.PlanHipError as e:
            assert e.code == hip.PH_EUNSUPPORTED
            continue
        want = O.select(oc, op, ok(), n=n)   # includes "this (type, op) selects nothing" cases
        got = dl(ctx, sel, np.int32, cnt)
        assert cnt == len(want), (name, op)
        assert np.array_equal(got.astype(np.int64), want), (name, op)
        ctx.free(sel)
    d.free()


def test_filter_and_chain_and_strings(ctx, sf001):
    L = sf001["lineitem"]
    n = len(L["l_shipdate"])
    ship = hip.DevColumn(ctx, hip.PH_DATE, L["l_shipdate"])
    disc = hip.DevColumn(ctx, hip.PH_DEC64, L["l_discount"], 2)
    qty = hip.DevColumn(ctx, hip.PH_I32, L["l_quantity"])
    d1, d2 = tpchgen.days(1994, 1, 1), tpchgen.days(1995, 1, 1)
    lo = float(np.float32(0.03) - np.float32(0.01))
    hi_ = float(np.float32(0.03) + np.float32(0.01))
    s1, c1 = hip.filter_select(ctx, ship, n, hip.PH_GE, hip.const(hip.PH_DATE, i=d1))
    s2, c2 = hip.filter_select(ctx, ship, n, hip.PH_LT, hip.const(hip.PH_DATE, i=d2), s1, c1)
    s3, c3 = hip.filter_select(ctx, disc, n, hip.PH_GE, hip.const(hip.PH_F32, f=lo), s2, c2)
    s4, c4 = hip.filter_select(ctx, disc, n, hip.PH_LE, hip.const(hip.PH_F32, f=hi_), s3, c3)
    s5, c5 = hip.filter_select(ctx, qty, n, hip.PH_LT, hip.const(hip.PH_I32, i=24), s4, c4)
    w = O.select(O.col(O.OT_DATE, L["l_shipdate"]), O.OP_GE, O.const(O.OT_DATE, i=d1), n=n)
    w = O.select(O.col(O.OT_DATE, L["l_shipdate"]), O.OP_LT, O.const(O.OT_DATE, i=d2), w)
    w = O.select(O.col(O.OT_DECIMAL, L["l_discount"], 2), O.OP_GE, O.const(O.OT_FLOAT, f=lo), w)
    w = O.select(O.col(O.OT_DECIMAL, L["l_discount"], 2), O.OP_LE, O.const(O.OT_FLOAT, f=hi_), w)
    w = O.select(O.col(O.OT_INT32, L["l_quantity"]), O.OP_LT, O.const(O.OT_INT32, i=24), w)
    assert c5 == len(w) and np.array_equal(dl(ctx, s5, np.int32, c5).astype(np.int64), w)
    # strings: LIKE / NOT LIKE / = / != over offsets + bytes
    P = sf001["part"]
    npart = len(P["p_partkey"])
    name = hip.DevColumn(ctx, hip.PH_STR, P["p_name_off"], aux=P["p_name_bytes"])
    oname = O.col(O.OT_VARCHAR, P["p_name_off"], dictionary=P["p_name_bytes"])
    first = bytes(P["p_name_bytes"][P["p_name_off"][0]:P["p_name_off"][1]]).decode()
    for op, pat in [(hip.PH_LIKE, "%pink%"), (hip.PH_NOTLIKE, "%pink%"), (hip.PH_LIKE, "%"),
                    (hip.PH_LIKE, "a%e_ %"), (hip.PH_LIKE, ""), (hip.PH_EQ, first), (hip.PH_NE, first),
                    (hip.PH_LIKE, "%ss__%")]:
        s, c = hip.filter_select(ctx, name, npart, op, hip.const(hip.PH_STR, s=pat))
        want = O.select(oname, op, O.const(O.OT_VARCHAR, s=pat), n=npart)
        assert c == len(want), (op, pat)
        assert np.array_equal(dl(ctx, s, np.int32, c).astype(np.int64), want)
        ctx.free(s)
    for d in (ship, disc, qty, name):
        d.free()


def test_two_conjuncts_over_one_column_in_one_pass(ctx):
    """ph_filter_select_and = ph_filter_select twice (execSelectAnd): date and integer ranges, '=' inside a range, an empty intersection,
    a (type, op) pair the reference lacks (DATE '=') selecting nothing, over all rows and over a selection; '!=' is refused."""
    rng = np.random.default_rng(8)
    n = 300_001
    d = rng.integers(8000, 10500, n).astype(np.int32)
    cases = [(hip.PH_DATE, hip.PH_GE, 9000, hip.PH_LT, 9365), (hip.PH_DATE, hip.PH_LT, 9365, hip.PH_GT, 8999), (hip.PH_I32, hip.PH_GE, 9000, hip.PH_EQ, 9100),
             (hip.PH_I32, hip.PH_GT, 9500, hip.PH_LT, 9400), (hip.PH_DATE, hip.PH_EQ, 9100, hip.PH_LT, 9365), (hip.PH_I32, hip.PH_LE, 9000, hip.PH_LE, 8500)]
    sel_in = np.sort(rng.choice(n, 100_000, replace=False)).astype(np.int32)
    ds = ctx.upload(sel_in)
    for typ, o1, a, o2, b in cases:
        col = hip.DevColumn(ctx, typ, d)
        k1, k2 = hip.const(typ, i=a), hip.const(typ, i=b)
        for s_in, m_in in ((None, n), (ds, len(sel_in))):
            s1, c1 = hip.filter_select(ctx, col, n, o1, k1, s_in, m_in)
            s2, c2 = hip.filter_select(ctx, col, n, o2, k2, s1, c1)
            sa, ca = hip.filter_select_and(ctx, col, n, o1, k1, o2, k2, s_in, m_in)
            assert ca == c2 and np.array_equal(dl(ctx, sa, np.int32, ca), dl(ctx, s2, np.int32, c2)), (typ, o1, a, o2, b)
            for q in (s1, s2, sa):
                ctx.free(q)
        col.free()
    col = hip.DevColumn(ctx, hip.PH_I32, d)
    with pytest.raises(hip.PlanHipError) as e:
        hip.filter_select_and(ctx, col, n, hip.PH_NE, hip.const(hip.PH_I32, i=9000), hip.PH_LT, hip.const(hip.PH_I32, i=9365))
    assert e.value.code == hip.PH_EUNSUPPORTED
    col.free(); ctx.free(ds)


def test_in_list_in_one_pass(ctx):
    """ph_filter_select_in = the union of the equalities (execSelectOr over InExpr's children): INTEGER values incl. negatives, repeated and
    absent ones, dictionary codes incl. a code no row can hold, NULL rows, a selection; DATE columns are refused (the reference has no DATE '=')"""
    rng = np.random.default_rng(12)
    n = 250_003
    v = rng.integers(-50, 50, n).astype(np.int32)
    codes = rng.integers(0, 40, n).astype(np.uint8)
    valid = rng.random(n) > 0.02
    vb = np.packbits(valid, bitorder="little")
    sel_in = np.sort(rng.choice(n, 90_000, replace=False)).astype(np.int32)
    ds = ctx.upload(sel_in)
    for typ, data, lists in ((hip.PH_I32, v, [[14, 7, 21, 24, 35, 33, 2, 20], [-50, 49, 49, 1000], [3], [2**40]]),
                             (hip.PH_CODE8, codes, [[1, 5, 39], [999, 7], [300], list(range(0, 40, 2)) + [255]])):   # (any number of codes: a bitmap)
        col = hip.DevColumn(ctx, typ, data, validity=vb)
        for vals in lists:
            for s_in, rows in ((None, np.arange(n)), (ds, sel_in)):
                want = rows[np.isin(data[rows].astype(np.int64), vals) & valid[rows]]
                s, c = hip.filter_select_in(ctx, col, n, vals, s_in, len(rows))
                assert c == len(want) and np.array_equal(dl(ctx, s, np.int32, c), want.astype(np.int32)), (typ, vals)
                ctx.free(s)
        col.free()
    col = hip.DevColumn(ctx, hip.PH_DATE, v)
    with pytest.raises(hip.PlanHipError) as e:
        hip.filter_select_in(ctx, col, n, [1, 2])
    assert e.value.code == hip.PH_EUNSUPPORTED
    col.free(); ctx.free(ds)


def test_filter_like_contains_edge_cases(ctx):
    """%literal% takes the position-parallel substring path: literals that span two rows must not
    match, matches at the first/last byte of a row must, empty and shorter-than-literal rows never
    do, NULL rows never select, and rows too long for the LDS stage fall back to the generic
    matcher (wildcardMatch, function_operator_boolean.go:336-377) with the same answers."""
    rng = np.random.default_rng(5)
    base = ["pi", "nk", "pink", "", "xpink", "pinkx", "p", "ink", "pinpink", "pin", "kpi", "nkp", "PINK", "pi nk",
            "a" * 30000 + "pink", "b" * 40, "pink" + "c" * 26000, "nk"]
    words = base + [("".join(rng.choice(list("pinkx "), int(rng.integers(0, 12))))) for _ in range(3000)]
    off = np.zeros(len(words) + 1, np.int32)
    off[1:] = np.cumsum([len(w) for w in words])
    b = np.frombuffer("".join(words).encode(), dtype=np.uint8)
    valid = np.ones(len(words), bool)
    valid[[4, 100, 101]] = False
    vb = np.packbits(valid, bitorder="little")
    col = hip.DevColumn(ctx, hip.PH_STR, off, aux=b, validity=vb)
    ocol = O.col(O.OT_VARCHAR, off, validity=vb, dictionary=b)
    for op, pat in [(hip.PH_LIKE, "%pink%"), (hip.PH_NOTLIKE, "%pink%"), (hip.PH_LIKE, "%k%"), (hip.PH_LIKE, "%pi nk%"),
                    (hip.PH_LIKE, "%nkp%"), (hip.PH_LIKE, "%p_nk%"), (hip.PH_LIKE, "%pin%pink%"),
                    # %A%B% (the two-literal kernel: earliest end of A <= latest start of B, inside ONE row): B may not overlap A, order
                    # matters, one-byte literals, literals at the row's ends, rows longer than the LDS stage
                    (hip.PH_NOTLIKE, "%pin%pink%"), (hip.PH_LIKE, "%pi%nk%"), (hip.PH_LIKE, "%nk%pi%"), (hip.PH_LIKE, "%p%p%"),
                    (hip.PH_LIKE, "%pink%pink%"), (hip.PH_LIKE, "%ink%k%"), (hip.PH_NOTLIKE, "%k%p%"), (hip.PH_LIKE, "%aaaaa%pink%"),
                    (hip.PH_LIKE, "%pink%ccccc%"), (hip.PH_LIKE, "%x %x%")]:
        s, c = hip.filter_select(ctx, col, len(words), op, hip.const(hip.PH_STR, s=pat))
        want = O.select(ocol, op, O.const(O.OT_VARCHAR, s=pat), n=len(words))
        assert c == len(want), (op, pat, c, len(want))
        assert np.array_equal(dl(ctx, s, np.int32, c).astype(np.int64), want), (op, pat)
        ctx.free(s)
    col.free()


def test_filter_dictionary_code_equality(ctx, sf001):
    C = sf001["customer"]
    n = len(C["c_custkey"])
    seg = hip.DevColumn(ctx, hip.PH_CODE8, C["c_mktsegment"])
    oseg = O.col(O.OT_CODE8, C["c_mktsegment"], dictionary=O.cdict(O.SEG))
    for op in (hip.PH_EQ, hip.PH_NE):
        for lit in ("HOUSEHOLD", "NOPE"):
            code = O.SEG.index(lit) if lit in O.SEG else 999
            s, c = hip.filter_select(ctx, seg, n, op, hip.const(hip.PH_I32, i=code))
            want = O.select(oseg, op, O.const(O.OT_VARCHAR, s=lit), n=n)
            assert c == len(want) and np.array_equal(dl(ctx, s, np.int32, c).astype(np.int64), want)
            ctx.free(s)
    seg.free()


# ------------------------------------------------------------------ hash

def test_hash_bit_identical(ctx):
    n = 50000
    rng = np.random.default_rng(3)
    i32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    i64 = rng.integers(-2**62, 2**62, n).astype(np.int64)
    date = rng.integers(-30000, 60000, n).astype(np.int32)
    dec = (rng.integers(-10**9, 10**9, n) * rng.choice([1, 10, 100, 1000], n)).astype(np.int64)
    dec[:5] = [0, 100, -100, 1050, -7]
    code = rng.integers(0, 3, n).astype(np.uint8)
    v, _ = rnd_validity(rng, n, 0.05)
    dh = np.array([hip.hash_bytes(s.encode()) for s in O.RF], dtype=np.uint64)
    dh_dev = ctx.upload(dh)
    specs = [
        [(hip.PH_I32, O.OT_INT32, i32, 0, None)],
        [(hip.PH_I64, O.OT_INT64, i64, 0, v)],
        [(hip.PH_DATE, O.OT_DATE, date, 0, None)],
        [(hip.PH_DEC64, O.OT_DECIMAL, dec, 4, None)],
        [(hip.PH_CODE8, O.OT_CODE8, code, 0, None)],
        [(hip.PH_I64, O.OT_INT64, i64, 0, None), (hip.PH_DATE, O.OT_DATE, date, 0, v),
         (hip.PH_I32, O.OT_INT32, i32, 0, None), (hip.PH_CODE8, O.OT_CODE8, code, 0, None)],
    ]
    for spec in specs:
        dcols = [hip.DevColumn(ctx, ht, a, sc, validity=val) for ht, _, a, sc, val in spec]
        ocols = [O.col(ot, a, sc, validity=val, dictionary=O.cdict(O.RF) if ot == O.OT_CODE8 else None)
                 for _, ot, a, sc, val in spec]
        dhs = [dh_dev if ht == hip.PH_CODE8 else None for ht, *_ in spec]
        out = hip.hash_cols(ctx, dcols, n, dhs)
        got = ctx.download(out, np.uint64, n)
        want = O.hash_cols(ocols, n)
        assert np.array_equal(got, want)
        ctx.free(out)
        for d in dcols:
            d.free()
    # strings
    words = ["", "a", "pink", "12345678", "123456789abcdef", "hello world, this is longer than 16"]
    off = np.zeros(len(words) + 1, np.int32)
    off[1:] = np.cumsum([len(w) for w in words])
    b = np.frombuffer("".join(words).encode(), dtype=np.uint8)
    sc = hip.DevColumn(ctx, hip.PH_STR, off, aux=b)
    out = hip.hash_cols(ctx, [sc], len(words))
    got = ctx.download(out, np.uint64, len(words))
    want = O.hash_cols([O.col(O.OT_VARCHAR, off, dictionary=b)], len(words))
    assert np.array_equal(got, want)
    assert [int(x) for x in got] == [hip.hash_bytes(w.encode()) for w in words]
    sc.free()


# ------------------------------------------------------------------ expressions

def test_expr_eval_matches_decimal_semantics(ctx):
    n = 30000
    rng = np.random.default_rng(11)
    ext = rng.integers(90000, 10500000, n).astype(np.int64)
    disc = rng.integers(0, 11, n).astype(np.int64)
    tax = rng.integers(0, 9, n).astype(np.int64)
    qty = rng.integers(1, 51, n).astype(np.int32)
    cost = rng.integers(100, 100001, n).astype(np.int64)
    sel = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
    dcols = [hip.DevColumn(ctx, hip.PH_DEC64, ext, 2), hip.DevColumn(ctx, hip.PH_DEC64, disc, 2),
             hip.DevColumn(ctx, hip.PH_DEC64, tax, 2), hip.DevColumn(ctx, hip.PH_I32, qty),
             hip.DevColumn(ctx, hip.PH_DEC64, cost, 2)]
    ocols = [O.col(O.OT_DECIMAL, ext, 2), O.col(O.OT_DECIMAL, disc, 2), O.col(O.OT_DECIMAL, tax, 2),
             O.col(O.OT_INT32, qty), O.col(O.OT_DECIMAL, cost, 2)]
    progs = {
        "disc_price": ([hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL],
                       [(O.OX_COL, 0, 0, 0), (O.OX_CONST_INT, 0, 1, 0), (O.OX_COL, 1, 0, 0),
                        (O.OX_SUB, 0, 0, 0), (O.OX_MUL, 0, 0, 0)], 4),
        "charge": ([hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL, hip.X_CONST(1),
                    hip.X_COL(2), hip.X_ADD, hip.X_MUL],
                   [(O.OX_COL, 0, 0, 0), (O.OX_CONST_INT, 0, 1, 0), (O.OX_COL, 1, 0, 0), (O.OX_SUB, 0, 0, 0),
                    (O.OX_MUL, 0, 0, 0), (O.OX_CONST_INT, 0, 1, 0), (O.OX_COL, 2, 0, 0), (O.OX_ADD, 0, 0, 0),
                    (O.OX_MUL, 0, 0, 0)], 6),
        "q9_amount": ([hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL, hip.X_COL(4),
                       hip.X_COL(3), hip.X_MUL, hip.X_SUB],
                      [(O.OX_COL, 0, 0, 0), (O.OX_CONST_INT, 0, 1, 0), (O.OX_COL, 1, 0, 0), (O.OX_SUB, 0, 0, 0),
                       (O.OX_MUL, 0, 0, 0), (O.OX_COL, 4, 0, 0), (O.OX_COL, 3, 0, 0), (O.OX_MUL, 0, 0, 0),
                       (O.OX_SUB, 0, 0, 0)], 4),
    }
    sel_dev = ctx.upload(sel)
    for name, (hp, op, scale) in progs.items():
        assert hip.expr_scale(dcols, hp) == scale
        for s_dev, s_np, m in ((None, None, n), (sel_dev, sel.astype(np.int64), len(sel))):
            out, _ = hip.expr_eval(ctx, dcols, hp, s_dev, m)
            got = ctx.download(out, np.int64, m)
            rc, want = O.eval_decimal(ocols, op, s_np, m)
            assert rc == 0
            assert [int(x) for x in got] == O.odec_unscaled(want, scale), name
            ctx.free(out)
    # overflow is detected, not wrapped
    big = hip.DevColumn(ctx, hip.PH_DEC64, np.full(100, 4_000_000_000, np.int64), 2)
    with pytest.raises(hip.PlanHipError) as e:
        hip.expr_eval(ctx, [big], [hip.X_COL(0), hip.X_COL(0), hip.X_MUL, hip.X_COL(0), hip.X_MUL], None, 100)
    assert e.value.code == hip.PH_EOVERFLOW
    # NULL in -> NULL out
    v, bits = rnd_validity(rng, n, 0.2)
    dn = hip.DevColumn(ctx, hip.PH_DEC64, disc, 2, validity=v)
    out, val = hip.expr_eval(ctx, [dcols[0], dn], [hip.X_COL(0), hip.X_COL(1), hip.X_MUL], None, n, True)
    gv = np.unpackbits(ctx.download(val, np.uint8, (n + 7) // 8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(gv, bits)
    got = ctx.download(out, np.int64, n)
    assert np.array_equal(got[bits], (ext * disc)[bits])
    for d in dcols + [big, dn]:
        d.free()


# ------------------------------------------------------------------ hash aggregate

def agg_compare(ctx, key_specs, arg_specs, aggs, n, sel=None, expected=16):
    """key_specs/arg_specs: (hip type, oracle type, array, scale, validity)"""
    dk = [hip.DevColumn(ctx, ht, a, sc, validity=v) for ht, _, a, sc, v in key_specs]
    da = [hip.DevColumn(ctx, ht, a, sc, validity=v) for ht, _, a, sc, v in arg_specs]
    ok = [O.col(ot, a, sc, validity=v, dictionary=O.cdict([str(i) for i in range(256)]) if ot == O.OT_CODE8 else None)
          for _, ot, a, sc, v in key_specs]
    # oracle args: decimals go in as ODEC arrays evaluated from the unscaled column
    oa = []
    keep = []
    for _, ot, a, sc, v in arg_specs:
        if ot == O.OT_DECIMAL:
            rc, od = O.eval_decimal([O.col(O.OT_DECIMAL, a, sc)], [(O.OX_COL, 0, 0, 0)], None, len(a))
            keep.append(od)
            oa.append(O.col(O.OT_ODEC, od, validity=v))
        else:
            oa.append(O.col(ot, a, sc, validity=v))
    agg = hip.Agg(ctx, [ht for ht, *_ in key_specs], aggs, expected)
    sel_dev = ctx.upload(sel.astype(np.int32)) if sel is not None else None
    m = len(sel) if sel is not None else n
    agg.sink(dk, da, sel_dev, m)
    r = agg.finalize()
    oaggs = [(k if k != hip.PH_A_COUNT_STAR else O.OA_COUNT, a if k != hip.PH_A_COUNT_STAR else -1) for k, a in aggs]
    ng, first, gk, gn, vals = O.groupby(ok, oa, oaggs, None if sel is None else sel.astype(np.int64), m,
                                        max(r["ngroups"], 1) + 8)
    assert r["ngroups"] == ng
    # The device emits groups in strict first-seen order. The reference's insertion order is the
    # same except where linear-probe collisions inside one 2048-row chunk defer an earlier row to
    # a later probing round (FindOrCreateGroups :272-388), so groups are matched by key here and
    # the first-seen row of every group is compared instead of the ordinal.
    assert np.all(np.diff(r["first_row"]) > 0)
    def keyof(nulls, vals_):
        return tuple(None if nulls[c] else int(vals_[c]) for c in range(len(key_specs)))
    gpu_index = {keyof(r["key_null"][g], r["keys"][g]): g for g in range(ng)}
    assert len(gpu_index) == ng
    for og in range(ng):
        g = gpu_index[keyof(gn[og], gk[og])]
        assert int(r["first_row"][g]) == int(first[og])      # row id of the group's first row
        for a, (kind, ai) in enumerate(aggs):
            v = vals[og * len(aggs) + a]
            cnt = int(r["count"][g][a])
            if v.kind == O.OV_NULL:
                assert cnt == 0
                continue
            scale = arg_specs[ai][3] if ai >= 0 else 0
            if kind in (hip.PH_A_COUNT, hip.PH_A_COUNT_STAR):
                assert cnt == v.h.value()
            elif kind == hip.PH_A_SUM:
                want = v.h.value() if v.kind == O.OV_HUGEINT else v.d.unscaled(scale)
                assert r["sum"][g][a] == want
            elif kind == hip.PH_A_AVG:
                if v.kind == O.OV_DOUBLE:
                    assert abs(r["sum"][g][a] / cnt - v.f) <= 1e-9 * max(abs(v.f), 1e-300)
                else:  # decimal: sum/count compared exactly via cross-multiplication to 19 digits
                    from fractions import Fraction
                    exact = Fraction(r["sum"][g][a], cnt * 10 ** scale)
                    got = Fraction(int(v.d.coef) * (-1 if v.d.neg else 1), 10 ** int(v.d.scale))
                    assert abs(exact - got) <= abs(exact) * Fraction(1, 10 ** 18) + Fraction(1, 10 ** 19)
            else:  # MIN / MAX
                s = r["sum"][g][a]
                assert s == v.d.unscaled(scale)
    agg.free()
    for d in dk + da:
        d.free()
    return r


def test_agg_q3_like_keys(ctx):
    rng = np.random.default_rng(5)
    n = 200_000
    okey = rng.integers(1, 40_000, n).astype(np.int64)
    odate = (8000 + okey % 700).astype(np.int32)
    prio = np.zeros(n, np.int32)
    rev = rng.integers(-10**9, 10**11, n).astype(np.int64)
    agg_compare(ctx,
                [(hip.PH_I64, O.OT_INT64, okey, 0, None), (hip.PH_DATE, O.OT_DATE, odate, 0, None),
                 (hip.PH_I32, O.OT_INT32, prio, 0, None)],
                [(hip.PH_DEC64, O.OT_DECIMAL, rev, 4, None)],
                [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1), (hip.PH_A_AVG, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0)],
                n)


def test_agg_nulls_and_selection(ctx):
    rng = np.random.default_rng(6)
    n = 50_000
    k0 = rng.integers(0, 7, n).astype(np.int32)
    k1 = rng.integers(0, 3, n).astype(np.uint8)
    q = rng.integers(-50, 51, n).astype(np.int32)
    d = rng.integers(-10**12, 10**12, n).astype(np.int64)
    vk, _ = rnd_validity(rng, n, 0.1)
    vq, _ = rnd_validity(rng, n, 0.3)
    vd, _ = rnd_validity(rng, n, 0.3)
    sel = np.sort(rng.choice(n, n // 2, replace=False))
    for s in (None, sel):
        agg_compare(ctx,
                    [(hip.PH_I32, O.OT_INT32, k0, 0, vk), (hip.PH_CODE8, O.OT_CODE8, k1, 0, None)],
                    [(hip.PH_I32, O.OT_INT32, q, 0, vq), (hip.PH_DEC64, O.OT_DECIMAL, d, 2, vd)],
                    [(hip.PH_A_SUM, 0), (hip.PH_A_AVG, 0), (hip.PH_A_COUNT, 0), (hip.PH_A_SUM, 1),
                     (hip.PH_A_AVG, 1), (hip.PH_A_MIN, 1), (hip.PH_A_MAX, 1), (hip.PH_A_COUNT_STAR, -1)],
                    n, sel=s)


def test_agg_many_groups_grows_table(ctx):
    rng = np.random.default_rng(8)
    n = 600_000
    k = rng.integers(0, 300_000, n).astype(np.int64)   # ~260k groups from a 4096-slot start
    v = rng.integers(0, 10**6, n).astype(np.int64)
    r = agg_compare(ctx, [(hip.PH_I64, O.OT_INT64, k, 0, None)], [(hip.PH_DEC64, O.OT_DECIMAL, v, 2, None)],
                    [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], n, expected=16)
    assert r["ngroups"] == len(np.unique(k))


def test_agg_bulk_build_of_an_empty_table(ctx):
    """A first sink with a high expected cardinality takes the partitioned bulk build (rows grouped
    by key hash, one workgroup aggregates a partition in LDS and writes every group once): three
    keys with NULLs, NULL arguments, every aggregate kind, a selection; then a wrong hint (few
    groups: all rows land in a handful of partitions and most overflow the LDS tables' ids into
    the row-by-row path) and a second, ordinary sink into the table the bulk build produced."""
    rng = np.random.default_rng(31)
    n = 700_000
    k0 = rng.integers(0, 90_000, n).astype(np.int64)
    k1 = rng.integers(9000, 9030, n).astype(np.int32)
    k2 = rng.integers(0, 3, n).astype(np.int32)
    vk, _ = rnd_validity(rng, n, 0.02)
    va, _ = rnd_validity(rng, n, 0.1)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    q = rng.integers(1, 51, n).astype(np.int32)
    sel = np.sort(rng.choice(n, 500_000, replace=False))
    keys = [(hip.PH_I64, O.OT_INT64, k0, 0, vk), (hip.PH_DATE, O.OT_DATE, k1, 0, None), (hip.PH_I32, O.OT_INT32, k2, 0, None)]
    args = [(hip.PH_DEC64, O.OT_DECIMAL, v, 2, va), (hip.PH_I32, O.OT_INT32, q, 0, None)]
    aggs = [(hip.PH_A_SUM, 0), (hip.PH_A_AVG, 1), (hip.PH_A_COUNT, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0), (hip.PH_A_COUNT_STAR, -1)]
    r = agg_compare(ctx, keys, args, aggs, n, sel=sel, expected=200_000)
    assert r["ngroups"] > 300_000
    # wrong hint: 5 groups
    k5 = rng.integers(0, 5, n).astype(np.int64)
    r = agg_compare(ctx, [(hip.PH_I64, O.OT_INT64, k5, 0, None)], [(hip.PH_DEC64, O.OT_DECIMAL, v, 2, None)],
                    [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], n, expected=100_000)
    assert r["ngroups"] == 5
    # bulk build, then an ordinary sink that mostly revisits its groups and adds some
    ka = rng.integers(0, 150_000, 400_000).astype(np.int64)
    kb = rng.integers(100_000, 200_000, 100_000).astype(np.int64)
    va_ = rng.integers(0, 1000, 400_000).astype(np.int64)
    vb_ = rng.integers(0, 1000, 100_000).astype(np.int64)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 100_000)
    da, dva = hip.DevColumn(ctx, hip.PH_I64, ka), hip.DevColumn(ctx, hip.PH_I64, va_)
    db, dvb = hip.DevColumn(ctx, hip.PH_I64, kb), hip.DevColumn(ctx, hip.PH_I64, vb_)
    agg.sink([da], [dva], None, len(ka))
    agg.sink([db], [dvb], None, len(kb), row_base=len(ka))
    r = agg.finalize()
    allk, allv = np.concatenate([ka, kb]), np.concatenate([va_, vb_])
    uk, inv = np.unique(allk, return_inverse=True)
    sums = np.bincount(inv, weights=allv.astype(np.float64)).astype(np.int64)
    cnts = np.bincount(inv)
    firsts = np.full(len(uk), len(allk)); np.minimum.at(firsts, inv, np.arange(len(allk)))
    assert r["ngroups"] == len(uk)
    order = np.argsort(firsts)
    assert np.array_equal(r["keys"][:, 0], uk[order]) and np.array_equal(r["first_row"], firsts[order])
    assert [x[0] for x in r["sum"]] == sums[order].tolist() and np.array_equal(r["count"][:, 1], cnts[order])
    agg.free()
    for d in (da, dva, db, dvb):
        d.free()


def test_agg_bulk_build_outgrows_its_hint(ctx):
    """More than 4 M rows: the bulk build starts from the caller's hint instead of a table that
    could hold a group per row, partitions that cannot reserve their ids back off, the host grows
    the table and reruns the build over the same partition records."""
    rng = np.random.default_rng(32)
    n = 4_600_000
    k = rng.integers(0, 1_200_000, n).astype(np.int64)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 40_000)   # hint 30x too low
    dk, dv = hip.DevColumn(ctx, hip.PH_I64, k), hip.DevColumn(ctx, hip.PH_I64, v)
    agg.sink([dk], [dv], None, n)
    r = agg.finalize(python_ints=False)
    uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
    order = np.argsort(first)
    assert r["ngroups"] == len(uk)
    assert np.array_equal(r["keys"][:, 0], uk[order]) and np.array_equal(r["first_row"], first[order])
    sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv, v)
    assert np.array_equal(r["sum_lo"][:, 0].astype(np.int64), sums[order])
    assert np.array_equal(r["sum_hi"][:, 0], np.where(sums[order] < 0, -1, 0))
    assert np.array_equal(r["count"][:, 1], np.bincount(inv)[order])
    agg.free(); dk.free(); dv.free()


def test_agg_bulk_build_second_form(ctx):
    """More than 4 M rows and up to ~260 k expected groups: the bulk build's second form — chunks staged in LDS by
    partition, `slices` workgroups per partition aggregating in LDS tables of their own, one merge workgroup per
    partition. Against the oracle: two keys (one with NULLs), NULL arguments, every aggregate kind, a selection.
    Against numpy: 65 536 groups of 5 M plain rows (keys, first rows, sums, counts), and sums whose 128-bit carries
    and borrows cross the slices of a partition."""
    rng = np.random.default_rng(41)
    n = 4_400_000
    k0 = rng.integers(0, 11_000, n).astype(np.int64)
    k1 = rng.integers(0, 4, n).astype(np.int32)
    vk, _ = rnd_validity(rng, n, 0.01)
    va, _ = rnd_validity(rng, n, 0.2)
    v = rng.integers(-10**9, 10**9, n).astype(np.int64)
    q = rng.integers(1, 51, n).astype(np.int32)
    sel = np.sort(rng.choice(n, 4_250_000, replace=False))
    keys = [(hip.PH_I64, O.OT_INT64, k0, 0, vk), (hip.PH_I32, O.OT_INT32, k1, 0, None)]
    args = [(hip.PH_DEC64, O.OT_DECIMAL, v, 2, va), (hip.PH_I32, O.OT_INT32, q, 0, None)]
    aggs = [(hip.PH_A_SUM, 0), (hip.PH_A_AVG, 1), (hip.PH_A_COUNT, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0), (hip.PH_A_COUNT_STAR, -1)]
    r = agg_compare(ctx, keys, args, aggs, n, sel=sel, expected=50_000)
    assert r["ngroups"] == 4 * 11_001
    # plain rows, 65 536 groups
    n = 5_000_000
    k = rng.integers(0, 65_536, n).astype(np.int64)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 65_536)
    dk, dv = hip.DevColumn(ctx, hip.PH_I64, k), hip.DevColumn(ctx, hip.PH_I64, v)
    agg.sink([dk], [dv], None, n, row_base=1000)
    r = agg.finalize(python_ints=False)
    uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
    order = np.argsort(first)
    assert r["ngroups"] == len(uk)
    assert np.array_equal(r["keys"][:, 0], uk[order]) and np.array_equal(r["first_row"], first[order] + 1000)
    sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv, v)
    assert np.array_equal(r["sum_lo"][:, 0].astype(np.int64), sums[order])
    assert np.array_equal(r["sum_hi"][:, 0], np.where(sums[order] < 0, -1, 0))
    assert np.array_equal(r["count"][:, 1], np.bincount(inv)[order])
    agg.free(); dk.free(); dv.free()
    # more groups than 512 partitions of LDS tables hold: two partition levels, one build workgroup per bin (1.5 M groups, NULL arguments)
    n = 5_200_000
    k = rng.integers(0, 1_500_000, n).astype(np.int64)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    bits = rng.random(n) < 0.9
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1), (hip.PH_A_MAX, 0)], 1_500_000)
    dk, dv = hip.DevColumn(ctx, hip.PH_I64, k), hip.DevColumn(ctx, hip.PH_I64, v, validity=np.packbits(bits, bitorder="little"))
    agg.sink([dk], [dv], None, n)
    r = agg.finalize(python_ints=False)
    uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
    order = np.argsort(first)
    assert r["ngroups"] == len(uk)
    assert np.array_equal(r["keys"][:, 0], uk[order]) and np.array_equal(r["first_row"], first[order])
    sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv[bits], v[bits])
    mx = np.full(len(uk), np.iinfo(np.int64).min); np.maximum.at(mx, inv[bits], v[bits])
    cnt = np.bincount(inv[bits], minlength=len(uk))
    assert np.array_equal(r["sum_lo"][:, 0].astype(np.int64), sums[order]) and np.array_equal(r["count"][:, 0], cnt[order])
    assert np.array_equal(r["count"][:, 1], np.bincount(inv)[order])
    has = cnt[order] > 0
    assert np.array_equal(r["sum_lo"][:, 2].astype(np.int64)[has], mx[order][has])
    agg.free(); dk.free(); dv.free()
    # 128-bit sums: +-2^62 per row, 3000 groups
    n = 4_300_000
    k = rng.integers(0, 3000, n).astype(np.int32)
    sign = rng.integers(0, 2, n).astype(np.int64) * 2 - 1
    v = sign * (2**62)
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0)], 3000)
    dk, dv = hip.DevColumn(ctx, hip.PH_I32, k), hip.DevColumn(ctx, hip.PH_I64, v)
    agg.sink([dk], [dv], None, n)
    r = agg.finalize()
    net = np.zeros(3000, np.int64); np.add.at(net, k, sign)
    assert r["ngroups"] == 3000
    for g in range(3000):
        key = int(r["keys"][g][0])
        assert r["sum"][g][0] == int(net[key]) * 2**62 and r["sum"][g][1] == -(2**62) and r["sum"][g][2] == 2**62
    agg.free(); dk.free(); dv.free()


def test_agg_single_hot_group_and_empty(ctx):
    n = 300_000
    k = np.zeros(n, np.int32)
    v = np.full(n, 2**40, np.int64)   # sum needs more than 64 bits? 3e5 * 2^40 < 2^63; use 128-bit anyway
    r = agg_compare(ctx, [(hip.PH_I32, O.OT_INT32, k, 0, None)], [(hip.PH_DEC64, O.OT_DECIMAL, v, 0, None)],
                    [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], n)
    assert r["ngroups"] == 1 and r["sum"][0][0] == n * 2**40
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_COUNT_STAR, -1)])
    assert agg.finalize()["ngroups"] == 0
    agg.free()


def test_agg_128bit_sums(ctx):
    """sums beyond int64: 128-bit accumulation must carry and borrow exactly"""
    n = 4096
    k = (np.arange(n) % 2).astype(np.int32)
    v = np.where(np.arange(n) % 4 < 2, 2**62, -(2**62)).astype(np.int64)
    v[0] = 2**62 + 12345
    dk = hip.DevColumn(ctx, hip.PH_I32, k)
    dv = hip.DevColumn(ctx, hip.PH_I64, v)
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0)])
    agg.sink([dk], [dv], None, n)
    r = agg.finalize()
    for g in range(2):
        assert r["sum"][g][0] == int(v[k == r["keys"][g][0]].astype(object).sum())
    agg.free(); dk.free(); dv.free()


# ------------------------------------------------------------------ hash join

def join_compare(ctx, bkeys, pkeys, bsel=None, psel=None, key_range=None, kind=None):
    """bkeys/pkeys: lists of (hip type, oracle type, array, validity)"""
    db = [hip.DevColumn(ctx, ht, a, validity=v) for ht, _, a, v in bkeys]
    dp = [hip.DevColumn(ctx, ht, a, validity=v) for ht, _, a, v in pkeys]
    ob = [O.col(ot, a, validity=v) for _, ot, a, v in bkeys]
    op = [O.col(ot, a, validity=v) for _, ot, a, v in pkeys]
    nb, np_ = len(bkeys[0][2]), len(pkeys[0][2])
    bs = ctx.upload(bsel.astype(np.int32)) if bsel is not None else None
    ps = ctx.upload(psel.astype(np.int32)) if psel is not None else None
    mb = len(bsel) if bsel is not None else nb
    mp = len(psel) if psel is not None else np_
    j = hip.Join(ctx, db, bs, mb, key_range=key_range)
    if kind is not None:
        assert j.kind == kind
    assert j.pairs_ordered() == (j.kind != "radix")
    oj = O.Join(ob, None if bsel is None else bsel.astype(np.int64), mb)
    assert j.count() == oj.count()
    cap = 1 << 22
    m, opr, obd = j.probe_inner(dp, ps, mp, cap)
    wm, wp, wb = oj.probe_inner(op, None if psel is None else psel.astype(np.int64), mp, cap)
    assert m == wm
    got = np.stack([dl(ctx, opr, np.int32, m), dl(ctx, obd, np.int32, m)], 1).astype(np.int64)
    want = np.stack([wp, wb], 1)
    # bit-exact row SET: same pairs; order differs (the reference emits per chain round)
    assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))])
    assert np.all(np.diff(got[:, 0]) >= 0) if psel is None and j.pairs_ordered() else True   # device order: by probe row
    f = j.probe_mark(dp, ps, mp)
    gf = dl(ctx, f, np.uint8, mp)
    wf = oj.probe_mark(op, None if psel is None else psel.astype(np.int64), mp)
    assert np.array_equal(gf, wf)
    j.free()
    for d in db + dp:
        d.free()
    return m


def test_join_unique_and_duplicate_keys(ctx):
    rng = np.random.default_rng(21)
    b = rng.permutation(200_000)[:120_000].astype(np.int64)           # unique build keys
    p = rng.integers(0, 250_000, 400_000).astype(np.int64)
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)])
    bd = rng.integers(0, 5000, 40_000).astype(np.int32)               # ~8 duplicates per key
    pd_ = rng.integers(0, 6000, 30_000).astype(np.int32)
    m = join_compare(ctx, [(hip.PH_I32, O.OT_INT32, bd, None)], [(hip.PH_I32, O.OT_INT32, pd_, None)])
    assert m > 100_000


def test_join_composite_nulls_selections_empty(ctx):
    rng = np.random.default_rng(22)
    nb, np_ = 30_000, 50_000
    b0 = rng.integers(0, 300, nb).astype(np.int32); b1 = rng.integers(0, 40, nb).astype(np.int32)
    p0 = rng.integers(0, 320, np_).astype(np.int32); p1 = rng.integers(0, 44, np_).astype(np.int32)
    vb, _ = rnd_validity(rng, nb, 0.1)
    vp, _ = rnd_validity(rng, np_, 0.1)
    bsel = np.sort(rng.choice(nb, nb // 2, replace=False))
    psel = np.sort(rng.choice(np_, np_ // 3, replace=False))
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, b0, vb), (hip.PH_I32, O.OT_INT32, b1, None)],
                 [(hip.PH_I32, O.OT_INT32, p0, None), (hip.PH_I32, O.OT_INT32, p1, vp)], bsel, psel)
    # nothing matches / empty build
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, np.arange(10, dtype=np.int32), None)],
                 [(hip.PH_I32, O.OT_INT32, np.arange(100, 200, dtype=np.int32), None)])
    j = hip.Join(ctx, [hip.DevColumn(ctx, hip.PH_I32, np.zeros(1, np.int32))], None, 0)
    assert j.count() == 0
    j.free()


def test_join_partitioned_build_and_fast_kernels(ctx):
    """Build sides of 128 K .. 4 M rows take the atomic-free partitioned build (rows grouped by
    bucket range, linked in LDS) and one-/two-key probes without NULLs take the type-specialised
    candidate / chain kernels: duplicates, a skewed key, NULL keys and selections on both sides."""
    rng = np.random.default_rng(23)
    nb, np_ = 400_000, 900_000
    # unique keys, selections on both sides (fast 8-byte kernels, SELP/SELB variants)
    b = rng.permutation(1_000_000)[:nb].astype(np.int64)
    p = rng.integers(0, 1_200_000, np_).astype(np.int64)
    bsel = np.sort(rng.choice(nb, 300_000, replace=False))
    psel = np.sort(rng.choice(np_, 500_000, replace=False))
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)], bsel, psel)
    # duplicates plus one heavily repeated key (a whole partition's worth of rows in one bucket)
    bd = rng.integers(0, 60_000, 200_000).astype(np.int32)
    bd[:20_000] = 7
    pd_ = rng.integers(0, 90_000, 150_000).astype(np.int32)
    pd_[:3] = 7
    m = join_compare(ctx, [(hip.PH_I32, O.OT_INT32, bd, None)], [(hip.PH_I32, O.OT_INT32, pd_, None)])
    assert m > 60_000
    # two keys: fast path without NULLs, generic kernels with NULLs on either side
    b0 = rng.integers(0, 3000, 180_000).astype(np.int32); b1 = rng.integers(0, 200, 180_000).astype(np.int32)
    p0 = rng.integers(0, 3200, 260_000).astype(np.int32); p1 = rng.integers(0, 220, 260_000).astype(np.int32)
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, b0, None), (hip.PH_I32, O.OT_INT32, b1, None)],
                 [(hip.PH_I32, O.OT_INT32, p0, None), (hip.PH_I32, O.OT_INT32, p1, None)])
    vb, _ = rnd_validity(rng, 180_000, 0.05)
    vp, _ = rnd_validity(rng, 260_000, 0.05)
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, b0, vb), (hip.PH_I32, O.OT_INT32, b1, None)],
                 [(hip.PH_I32, O.OT_INT32, p0, None), (hip.PH_I32, O.OT_INT32, p1, vp)])


def test_join_direct_table_dense_keys(ctx):
    """ph_join_build_range: a dense key range (<= 8 slots per build row) builds a direct table
    addressed by key - lo. Every probe form returns what the hash tables return (oracle = the
    reference's chained table): unique keys with selections on both sides, duplicate build keys
    (linked after the plain-store scatter), NULL keys on both sides, a negative lower bound, probe
    keys outside the range, the fused Filter -> probe, the lookup probe; sparse ranges and
    PH_JOIN_DIRECT=0 keep the hash tables; a build key outside the stated range is an error."""
    import os
    rng = np.random.default_rng(31)
    nb, np_ = 300_000, 700_000
    b = (rng.permutation(1_000_000)[:nb] + 5_000).astype(np.int64)          # unique, range 1e6 <= 8 * nb
    p = rng.integers(0, 1_100_000, np_).astype(np.int64)                     # some below lo, some above hi
    bsel = np.sort(rng.choice(nb, 200_000, replace=False))
    psel = np.sort(rng.choice(np_, 400_000, replace=False))
    rngk = (5_000, 1_004_999)
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=rngk, kind="direct")
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)], bsel, psel, key_range=rngk, kind="direct")
    # duplicates (~8 per key, one key 5000 times), int32 keys, negative lower bound, NULLs on both sides
    bd = (rng.integers(0, 5000, 40_000) - 2500).astype(np.int32)
    bd[:5000] = 17
    pd_ = (rng.integers(0, 6000, 30_000) - 3000).astype(np.int32)
    vb, _ = rnd_validity(rng, len(bd), 0.1)
    vp, _ = rnd_validity(rng, len(pd_), 0.1)
    m = join_compare(ctx, [(hip.PH_I32, O.OT_INT32, bd, None)], [(hip.PH_I32, O.OT_INT32, pd_, None)], key_range=(-2500, 2499), kind="direct")
    assert m > 100_000
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, bd, vb)], [(hip.PH_I32, O.OT_INT32, pd_, vp)],
                 np.sort(rng.choice(len(bd), 30_000, replace=False)), np.sort(rng.choice(len(pd_), 20_000, replace=False)),
                 key_range=(-2500, 2499), kind="direct")
    # a sparse build side in a SMALL range (<= 4 M slots) is direct too, with the occupied-group bitmap
    # that big probes keep in LDS; sparse in a large range / switched off: the hash tables, same results
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b[:1000], None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=rngk, kind="direct")
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b[:20_000], None)], [(hip.PH_I64, O.OT_INT64, p, None)], None, psel, key_range=(0, 3_999_999), kind="direct")
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b[:1000], None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=(0, 2**31), kind="chained+bloom")
    os.environ["PH_JOIN_DIRECT"] = "0"
    try:
        join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=rngk, kind="chained+bloom")
    finally:
        os.environ.pop("PH_JOIN_DIRECT")
    # lookup probe + fused Filter -> probe against the hash table's answers
    db, dp = hip.DevColumn(ctx, hip.PH_I64, b), hip.DevColumn(ctx, hip.PH_I64, p)
    w = rng.integers(0, 100, np_).astype(np.int32)
    dw = hip.DevColumn(ctx, hip.PH_I32, w)
    jd, jh = hip.Join(ctx, [db], None, nb, key_range=rngk), hip.Join(ctx, [db], None, nb)
    assert jd.kind == "direct" and jh.kind == "chained+bloom" and jd.count() == jh.count() == nb
    st = ctx.upload(np.zeros(2, np.int32))
    got = ctx.download(jd.lookup([dp], None, np_, st), np.int32, np_)
    want = ctx.download(jh.lookup([dp], None, np_), np.int32, np_)
    assert np.array_equal(got, want) and ctx.download(st, np.int32, 2).tolist() == [int((want < 0).sum()), 0]
    rd = jd.probe_inner_where([dp], dw, hip.PH_LT, hip.const(hip.PH_I32, i=40), None, np_, np_)
    rh = jh.probe_inner_where([dp], dw, hip.PH_LT, hip.const(hip.PH_I32, i=40), None, np_, np_)
    assert rd is not None and rh is not None and rd[0] == rh[0] == int(((want >= 0) & (w < 40)).sum())
    for x, y in ((rd[1], rh[1]), (rd[2], rh[2])):
        assert np.array_equal(ctx.download(x, np.int32, rd[0]), ctx.download(y, np.int32, rd[0]))
    jd.free(); jh.free()
    # a build key outside the stated range: reported by ph_join_count
    jbad = hip.Join(ctx, [db], None, nb, key_range=(5_000, 900_000))
    assert jbad.kind == "direct" and jbad.count() == -1 and "outside the stated range" in hip.last_error()
    jbad.free()
    for c in (db, dp, dw):
        c.free()


def test_join_unhinted_big_build_reads_its_key_range(ctx):
    """ph_join_build without a range and a build side of a million keys or more: the library reads min / max off
    the key column (NULL keys and unselected rows skipped) and builds the direct table when the keys are dense in
    their range — sorted keys (the verified one-pass fill), shuffled keys with NULLs behind a selection (general
    passes), duplicates — and keeps the hash tables when they are not (random 62-bit keys) or when
    PH_JOIN_AUTO_RANGE=0. All against the oracle's pairs."""
    import os
    rng = np.random.default_rng(77)
    nb, np_ = 1_200_000, 1_500_000
    dense = np.sort(rng.choice(5_000_000, nb, replace=False)).astype(np.int64) + 10**12     # 24 % dense, far from zero
    p = rng.integers(10**12 - 1000, 10**12 + 5_001_000, np_).astype(np.int64)
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, dense, None)], [(hip.PH_I64, O.OT_INT64, p, None)], kind="direct")
    shuf = rng.permutation(dense)
    shuf[::1000] = shuf[1::1000]                                                             # duplicate keys
    vb, _ = rnd_validity(rng, nb, 0.01)
    bsel = np.sort(rng.choice(nb, 1_100_000, replace=False))
    psel = np.sort(rng.choice(np_, 700_000, replace=False))
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, shuf, vb)], [(hip.PH_I64, O.OT_INT64, p, None)], bsel, psel, kind="direct")
    k32 = rng.permutation(np.arange(-600_000, 600_000, dtype=np.int32))
    p32 = rng.integers(-700_000, 700_000, np_).astype(np.int32)
    join_compare(ctx, [(hip.PH_I32, O.OT_INT32, k32, None)], [(hip.PH_I32, O.OT_INT32, p32, None)], kind="direct")
    sparse = rng.integers(0, 2**62, nb).astype(np.int64)
    ps = np.concatenate([sparse[:400_000], rng.integers(0, 2**62, 300_000).astype(np.int64)])
    j = hip.Join(ctx, [hip.DevColumn(ctx, hip.PH_I64, sparse)], None, nb)
    assert j.kind != "direct"
    j.free()
    join_compare(ctx, [(hip.PH_I64, O.OT_INT64, sparse, None)], [(hip.PH_I64, O.OT_INT64, ps, None)])
    os.environ["PH_JOIN_AUTO_RANGE"] = "0"
    try:
        d = hip.DevColumn(ctx, hip.PH_I64, dense)
        j = hip.Join(ctx, [d], None, nb)
        assert j.kind != "direct" and j.count() == nb
        j.free(); d.free()
    finally:
        os.environ.pop("PH_JOIN_AUTO_RANGE")


def test_gather_multi_through_a_colocated_copy(ctx):
    """ph_table_colocate: columns of widths 8 / 4 / 1 side by side per row; ph_gather_multi over views of them reads the
    copy (any subset, any order of the columns, negative row ids read row 0 as before) and gives what the column arrays
    give; a sparse gather of three columns that comes a second time builds the copy by itself."""
    rng = np.random.default_rng(5)
    n = 300_000
    a = rng.integers(-2**62, 2**62, n).astype(np.int64)
    b = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    c = rng.integers(0, 200, n).astype(np.uint8)
    d = rng.integers(0, 10**9, n).astype(np.int64)
    e = rng.integers(0, 50, n).astype(np.int32)
    t = hip.Table(ctx, [(hip.PH_I64, a), (hip.PH_I32, b), (hip.PH_CODE8, c, 0, None, [str(i) for i in range(200)]), (hip.PH_DEC64, d, 2), (hip.PH_I32, e)], n)
    cols = [hip.TableColumn(t, i) for i in range(5)]
    host = [a, b, c, d, e]
    idx = rng.integers(0, n, 20_000).astype(np.int32)
    idx[:5] = -1
    di = ctx.upload(idx)
    want_idx = np.where(idx < 0, 0, idx)
    def check(which):
        outs = hip.gather_multi(ctx, [cols[i] for i in which], di, len(idx))
        for o, i in zip(outs, which):
            got = ctx.download(o, host[i].dtype, len(idx))
            assert np.array_equal(got, host[i][want_idx]), i
            ctx.free(o)
    assert not t.colocated([0, 1, 3])
    check([3, 0, 1])                      # first sparse gather of this set: the column arrays
    assert not t.colocated([0, 1, 3])
    check([3, 0, 1])                      # the second builds the copy and reads it
    assert t.colocated([0, 1, 3]) and not t.colocated([0, 1, 2, 3, 4])
    check([1, 3]); check([0, 1, 3])
    t.colocate([0, 1, 2, 3, 4])           # the planner's set, widths 8, 4, 1, 8, 4
    assert t.colocated([4, 2])
    for which in ([0, 1, 2, 3, 4], [4, 2, 0], [2, 1], [3, 4, 0, 2]):
        check(which)
    ctx.free(di)
    t.free()


def test_join_radix_partitioned_form(ctx, monkeypatch):
    """Big build sides whose keys are not dense in a range: both sides partitioned by key hash, the tables built in
    LDS and written out as images, probes against the images of their partition (kind "radix"; PH_JOIN_RADIX_MIN
    lowers the 4 M-row threshold here). Random 62-bit keys with duplicates on both sides, NULL keys and
    selections on both sides, one INTEGER key, two INTEGER keys (packed), the unpartitioned probe of small probe
    sides, a bin that overflows its table (one key repeated 9000 times: the hash tables instead), and the probe
    kinds the radix form hands to the node table (lookup, mark) — all against the oracle / the node table."""
    monkeypatch.setenv("PH_JOIN_RADIX_MIN", "100000")
    monkeypatch.setenv("PH_JOIN_AUTO_RANGE", "0")
    rng = np.random.default_rng(91)
    nb, np_ = 600_000, 1_400_000
    b = rng.integers(0, 2**62, nb).astype(np.int64)
    b[::50] = b[1::50]                                           # duplicate build keys
    p = np.concatenate([b[rng.integers(0, nb, 900_000)], rng.integers(0, 2**62, np_ - 900_000).astype(np.int64)])
    rng.shuffle(p)
    for part_min in ("1", str(1 << 40)):                         # partitioned probe / straight from the probe column
        monkeypatch.setenv("PH_JOIN_RADIX_PART_MIN", part_min)
        join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, None)], [(hip.PH_I64, O.OT_INT64, p, None)], kind="radix")
        vb, _ = rnd_validity(rng, nb, 0.02)
        vp, _ = rnd_validity(rng, np_, 0.02)
        bsel = np.sort(rng.choice(nb, 500_000, replace=False))
        psel = np.sort(rng.choice(np_, 1_200_000, replace=False))
        join_compare(ctx, [(hip.PH_I64, O.OT_INT64, b, vb)], [(hip.PH_I64, O.OT_INT64, p, vp)], bsel, psel, kind="radix")
        k32 = rng.integers(-2**31, 2**31 - 1, nb).astype(np.int32)
        p32 = np.concatenate([k32[rng.integers(0, nb, 700_000)], rng.integers(-2**31, 2**31 - 1, 700_000).astype(np.int32)])
        join_compare(ctx, [(hip.PH_I32, O.OT_INT32, k32, None)], [(hip.PH_I32, O.OT_INT32, p32, None)], kind="radix")
        ka, kb = rng.integers(0, 200_000, nb).astype(np.int32), rng.integers(0, 4, nb).astype(np.int32)
        pa, pb = rng.integers(0, 220_000, np_).astype(np.int32), rng.integers(0, 4, np_).astype(np.int32)
        join_compare(ctx, [(hip.PH_I32, O.OT_INT32, ka, None), (hip.PH_I32, O.OT_INT32, kb, None)],
                     [(hip.PH_I32, O.OT_INT32, pa, None), (hip.PH_I32, O.OT_INT32, pb, None)], kind="radix")
    monkeypatch.delenv("PH_JOIN_RADIX_PART_MIN")
    # a bin that cannot hold its keys: the node table takes the build
    hot = b.copy()
    hot[:9000] = 12345
    d = hip.DevColumn(ctx, hip.PH_I64, hot)
    j = hip.Join(ctx, [d], None, nb)
    assert j.kind != "radix" and j.count() == nb
    j.free(); d.free()
    # lookups and marks of a radix join go through the node table, built on first use
    ub = rng.permutation(np.unique(b))
    db, dp = hip.DevColumn(ctx, hip.PH_I64, ub), hip.DevColumn(ctx, hip.PH_I64, p)
    jr = hip.Join(ctx, [db], None, len(ub))
    monkeypatch.setenv("PH_JOIN_RADIX", "0")
    jn = hip.Join(ctx, [db], None, len(ub))
    monkeypatch.delenv("PH_JOIN_RADIX")
    assert jr.kind == "radix" and jn.kind != "radix" and jr.count() == jn.count() == len(ub)
    got = ctx.download(jr.lookup([dp], None, np_), np.int32, np_)
    want = ctx.download(jn.lookup([dp], None, np_), np.int32, np_)
    assert np.array_equal(got, want) and (want >= 0).sum() > 800_000
    mr, ar, br = jr.probe_inner([dp], None, np_, np_)
    pairs = np.stack([ctx.download(ar, np.int32, mr), ctx.download(br, np.int32, mr)], 1)
    pairs = pairs[np.argsort(pairs[:, 0], kind="stable")]
    hit = np.nonzero(want >= 0)[0]
    assert mr == len(hit) and np.array_equal(pairs[:, 0], hit) and np.array_equal(pairs[:, 1], want[hit])
    jr.free(); jn.free(); db.free(); dp.free()


def test_join_direct_table_sorted_fill(ctx):
    """Build keys in storage order (sorted) and a range above 8 M slots take the one-pass sorted fill of
    the direct table; the kernel verifies the order, so unsorted keys of the same shape fall back to
    the general passes: sorted unique keys, sorted keys with duplicate runs (one across a 2048-row
    chunk boundary), one descending pair in the middle, and PH_JOIN_SORTED_FILL=0 all give the
    reference's pairs / marks / counts."""
    import os
    rng = np.random.default_rng(61)
    n, np_ = 1_300_000, 600_000
    uniq = np.sort(rng.choice(10_000_000, n, replace=False)).astype(np.int64) + 77
    dup = np.sort(rng.integers(0, 9_500_000, n)).astype(np.int64) + 77
    dup[2046:2051] = dup[2046]                                         # a run across the first chunk boundary
    dup[-3:] = dup[-1]
    swapped = uniq.copy()
    swapped[700_000], swapped[700_001] = uniq[700_001], uniq[700_000]  # one descending pair
    p = rng.integers(0, 10_000_200, np_).astype(np.int64)
    p[:1000] = dup[2046]
    rngk = (77, 10_000_076)
    for keys in (uniq, dup, swapped):
        join_compare(ctx, [(hip.PH_I64, O.OT_INT64, keys, None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=rngk, kind="direct")
    os.environ["PH_JOIN_SORTED_FILL"] = "0"
    try:
        join_compare(ctx, [(hip.PH_I64, O.OT_INT64, dup, None)], [(hip.PH_I64, O.OT_INT64, p, None)], key_range=rngk, kind="direct")
    finally:
        os.environ.pop("PH_JOIN_SORTED_FILL")
    # int32 keys, lookup of every build key finds a row of its run
    k32 = np.sort(rng.integers(0, 9_000_000, 1_200_000)).astype(np.int32)
    d32 = hip.DevColumn(ctx, hip.PH_I32, k32)
    j = hip.Join(ctx, [d32], None, len(k32), key_range=(0, 9_000_000))
    assert j.kind == "direct" and j.count() == len(k32)
    got = ctx.download(j.lookup([d32], None, len(k32)), np.int32, len(k32))
    assert np.all(got >= 0) and np.array_equal(k32[got], k32)
    j.free(); d32.free()


def test_join_keys_declared_sorted_unique(ctx):
    """PH_JOIN_KEYS_SORTED_UNIQUE: with a true claim the direct table is built by the fill alone and
    answers like the undeclared build; with a false claim (one descending pair; one duplicate) the
    build call still returns, probes stay in bounds, and the next read-back fails with the deferred
    PH_ECONSTRAINT — after which an undeclared build of the same keys is correct."""
    rng = np.random.default_rng(91)
    n = 1_200_000                                                   # 9 M slots: above the bitmap limit, within 8 slots per row
    keys = np.sort(rng.choice(9_000_000, n, replace=False)).astype(np.int64) + 5
    rngk = (5, 9_000_004)
    p = rng.integers(0, 9_000_100, 500_000).astype(np.int64)
    dk, dp = hip.DevColumn(ctx, hip.PH_I64, keys), hip.DevColumn(ctx, hip.PH_I64, p)
    jd = hip.Join(ctx, [dk], None, n, key_range=rngk, sorted_unique=True)
    ju = hip.Join(ctx, [dk], None, n, key_range=rngk)
    assert jd.kind == ju.kind == "direct" and jd.count() == ju.count() == n
    a = ctx.download(jd.lookup([dp], None, len(p)), np.int32, len(p))
    b = ctx.download(ju.lookup([dp], None, len(p)), np.int32, len(p))
    assert np.array_equal(a, b) and (a >= 0).sum() > 40_000
    ctx.check_deferred()
    jd.free(); ju.free()
    for kind in ("descending", "duplicate"):
        bad = keys.copy()
        if kind == "descending":
            bad[600_000], bad[600_001] = keys[600_001], keys[600_000]
        else:
            bad[700_001] = bad[700_000]   # (still non-decreasing: only the uniqueness half of the claim fails)
        dbad = hip.DevColumn(ctx, hip.PH_I64, bad)
        jb = hip.Join(ctx, [dbad], None, n, key_range=rngk, sorted_unique=True)
        rows = jb.lookup([dp], None, len(p))                      # slots may be unwritten: anything but a build row reads as empty
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(rows, np.int32, len(p))
        assert e.value.code == hip.PH_ECONSTRAINT and "SORTED_UNIQUE" in str(e.value)
        jb.free()
        jg = hip.Join(ctx, [dbad], None, n, key_range=rngk)     # the caller's second attempt, without the claim
        got = ctx.download(jg.lookup([dp], None, len(p)), np.int32, len(p))
        hit = got >= 0
        assert np.array_equal(bad[got[hit]], p[hit]) and np.array_equal(hit, np.isin(p, bad))
        jg.free(); dbad.free()
    dk.free(); dp.free()


def test_join_sorted_fill_understated_range_stays_in_bounds(ctx):
    """sorted build keys whose tail lies ABOVE the stated key_hi (an understated statistic), longer
    than one 2048-row chunk and ending in a far sentinel, and keys below key_lo in later chunks. The fill derives
    a chunk's slot span from raw keys, so it must clamp the span to the table: the build reports the error
    (ph_join_count = -1 for the undeclared build, the deferred PH_ECONSTRAINT for PH_JOIN_KEYS_SORTED_UNIQUE,
    plain and gated) and guard buffers allocated around the table keep their pattern."""
    rng = np.random.default_rng(97)
    n = 1_200_000
    inside = np.sort(rng.choice(9_000_000, n - 6000, replace=False)).astype(np.int64) + 100
    tail = np.arange(6000, dtype=np.int64) * 3 + 9_500_000             # 6000 rows (three chunks) above key_hi
    tail[-1] = 2**40                                                   # a far sentinel as the last key
    hi_keys = np.concatenate([inside, tail])
    low_keys = np.concatenate([np.arange(5000, dtype=np.int64) - 6000, inside[: n - 5000] + 20_000])   # chunks 0..2 below key_lo
    rngk = (100, 9_000_099)
    flag = np.ones(n, np.uint8)
    dfl = hip.DevColumn(ctx, hip.PH_CODE8, flag)
    p = rng.integers(0, 9_000_200, 300_000).astype(np.int64)
    dp = hip.DevColumn(ctx, hip.PH_I64, p)
    pattern = np.full(1 << 20, 0x5A, np.uint8)
    for keys in (hi_keys, low_keys):
        assert np.all(np.diff(keys) > 0)
        dk = hip.DevColumn(ctx, hip.PH_I64, keys)
        for form in ("range", "declared", "gated"):
            g0 = ctx.upload(pattern)
            if form == "range":
                j = hip.Join(ctx, [dk], None, n, key_range=rngk)
            elif form == "declared":
                j = hip.Join(ctx, [dk], None, n, key_range=rngk, sorted_unique=True)
            else:
                j = hip.Join.build_where(ctx, [dk], dfl, hip.PH_EQ, hip.const(hip.PH_I32, i=1), None, n, rngk, sorted_unique=True)
            g1 = ctx.upload(pattern)
            assert j is not None and j.kind == "direct"
            rows = j.lookup([dp], None, len(p))                        # probes stay in bounds whatever the fill wrote
            if form == "range":
                ctx.download(rows, np.int32, len(p))
                assert j.count() == -1 and "outside the stated range" in hip.last_error()
            else:
                with pytest.raises(hip.PlanHipError) as e:
                    ctx.download(rows, np.int32, len(p))
                assert e.value.code == hip.PH_ECONSTRAINT
            ctx.sync()
            assert np.array_equal(ctx.download(g0, np.uint8, len(pattern)), pattern)
            assert np.array_equal(ctx.download(g1, np.uint8, len(pattern)), pattern)
            j.free()
            ctx.free(g0); ctx.free(g1); ctx.free(rows)
        dk.free()
    dfl.free(); dp.free()


def test_join_build_where_sorted_unique_gated_fill(ctx):
    """ph_join_build_where_ex with PH_JOIN_KEYS_SORTED_UNIQUE: the Filter rides along in the sorted fill,
    which also writes the occupancy bitmap (here 9 M slots: above the general passes' bitmap limit). Pairs,
    lookups and the row count equal ph_filter_select + ph_join_build_range; int32 and int64 keys; a gate
    nothing passes; a false claim is the deferred PH_ECONSTRAINT."""
    rng = np.random.default_rng(93)
    n = 1_300_000
    for dt, typ in ((np.int64, hip.PH_I64), (np.int32, hip.PH_I32)):
        keys = (np.sort(rng.choice(9_000_000, n, replace=False)) + 7).astype(dt)
        rngk = (7, 9_000_006)
        flag = (rng.random(n) < 0.1).astype(np.uint8)
        flag[:3] = 1; flag[-3:] = 1                                    # first and last tile edges
        p = np.concatenate([rng.integers(0, 9_000_100, 700_000), keys[rng.integers(0, n, 100_000)]]).astype(dt)
        p.sort()                                                        # clustered probes, like lineitem by order key
        ship = rng.integers(0, 100, len(p)).astype(np.int32)
        dk, dfl, dp, dsh = (hip.DevColumn(ctx, typ, keys), hip.DevColumn(ctx, hip.PH_CODE8, flag), hip.DevColumn(ctx, typ, p),
                            hip.DevColumn(ctx, hip.PH_I32, ship))
        for code in (1, 5):                                             # 5: nothing passes
            jw = hip.Join.build_where(ctx, [dk], dfl, hip.PH_EQ, hip.const(hip.PH_I32, i=code), None, n, rngk, sorted_unique=True)
            assert jw is not None and jw.kind == "direct"
            fs, fn = hip.filter_select(ctx, dfl, n, hip.PH_EQ, hip.const(hip.PH_I32, i=code))
            jf = hip.Join(ctx, [dk], fs, fn, key_range=rngk)
            assert jw.count() == jf.count() == int((flag == code).sum())
            mw, pw, bw = jw.probe_inner([dp], None, len(p), len(p))
            mf, pf, bf = jf.probe_inner([dp], None, len(p), len(p))
            assert mw == mf == int(np.isin(p, keys[flag == code]).sum())
            assert np.array_equal(ctx.download(pw, np.int32, mw), ctx.download(pf, np.int32, mf))
            assert np.array_equal(ctx.download(bw, np.int32, mw), ctx.download(bf, np.int32, mf))
            fw = jw.probe_inner_where([dp], dsh, hip.PH_GT, hip.const(hip.PH_I32, i=40), None, len(p), len(p))
            assert fw is not None
            want = np.isin(p, keys[flag == code]) & (ship > 40)
            assert fw[0] == int(want.sum()) and np.array_equal(ctx.download(fw[1], np.int32, fw[0]), np.nonzero(want)[0])
            assert np.array_equal(keys[ctx.download(fw[2], np.int32, fw[0])], p[want])
            lw = ctx.download(jw.lookup([dp], None, len(p)), np.int32, len(p))
            lf = ctx.download(jf.lookup([dp], None, len(p)), np.int32, len(p))
            assert np.array_equal(lw, lf)
            ctx.check_deferred()
            for b in (pw, bw, pf, bf, fs, fw[1], fw[2]):
                ctx.free(b)
            jw.free(); jf.free()
        # gates over 4- and 8-byte columns (a date range, a decimal threshold) through the same fill
        gd = rng.integers(8000, 10000, n).astype(np.int32)
        gv = rng.integers(0, 10**9, n).astype(np.int64)
        dgd, dgv = hip.DevColumn(ctx, hip.PH_DATE, gd), hip.DevColumn(ctx, hip.PH_DEC64, gv, 2)   # (BIGINT has no comparison in the reference)
        for wcol, op, kc, keep in ((dgd, hip.PH_LT, hip.const(hip.PH_DATE, i=8200), gd < 8200),
                                   (dgv, hip.PH_GT, hip.const(hip.PH_DEC64, i=900_000_000, scale=2), gv > 900_000_000)):
            jw = hip.Join.build_where(ctx, [dk], wcol, op, kc, None, n, rngk, sorted_unique=True)
            assert jw is not None and jw.kind == "direct" and jw.count() == int(keep.sum())
            lw = ctx.download(jw.lookup([dp], None, len(p)), np.int32, len(p))
            pos = np.searchsorted(keys, p)
            hit = (pos < n) & (keys[np.minimum(pos, n - 1)] == p)
            hit &= keep[np.minimum(pos, n - 1)]
            assert np.array_equal(lw >= 0, hit) and np.array_equal(lw[hit], pos[hit])
            ctx.check_deferred()
            jw.free()
        dgd.free(); dgv.free()
        bad = keys.copy()
        bad[650_000], bad[650_001] = keys[650_001], keys[650_000]
        dbad = hip.DevColumn(ctx, typ, bad)
        jb = hip.Join.build_where(ctx, [dbad], dfl, hip.PH_EQ, hip.const(hip.PH_I32, i=1), None, n, rngk, sorted_unique=True)
        rows = jb.lookup([dp], None, len(p))
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(rows, np.int32, len(p))
        assert e.value.code == hip.PH_ECONSTRAINT
        jb.free()
        for c in (dk, dfl, dp, dsh, dbad):
            c.free()


def test_sorted_pairs_and_run_lookup_equal_table_joins(ctx):
    """The two table-less joins against a table stored in key order answer like a built table:
    ph_join_sorted_pairs (binary search of a clustered key column with duplicates: the lineitem-by-order shape) against ph_join_build +
    ph_join_probe_inner, incl. absent keys, a probe selection, runs longer than the walk's 16 rows, the capacity protocol;
    ph_join_run_lookup (runs of one length by the first key, the second key picks the row: the partsupp shape) against numpy, incl. misses,
    NULL probe keys, keys outside the range, and the strict form's deferred error; ph_table_col_run_len finds the shape and refuses near misses."""
    rng = np.random.default_rng(202)
    # ---- sorted pairs
    norders = 300_000
    per = rng.integers(1, 8, norders)
    per[1000] = 40                                                   # one long run
    okeys = (np.arange(norders, dtype=np.int64) // 8) * 32 + np.arange(norders) % 8 + 1
    bk = np.repeat(okeys, per)
    nb = len(bk)
    db = hip.DevColumn(ctx, hip.PH_I64, bk)
    probe = np.concatenate([okeys[rng.integers(0, norders, 20_000)], np.array([okeys[1000], -3, int(okeys[-1]) + 9], dtype=np.int64),
                            rng.integers(0, int(okeys[-1]), 2000)]).astype(np.int64)
    rng.shuffle(probe)
    dp = hip.DevColumn(ctx, hip.PH_I64, probe)
    first = np.searchsorted(bk, probe, "left"); last = np.searchsorted(bk, probe, "right")
    want = [(i, r) for i in range(len(probe)) for r in range(first[i], last[i])]
    op, ob, m = hip.join_sorted_pairs(ctx, db, nb, dp, None, len(probe), len(want) + 10)
    got = list(zip(ctx.download(op, np.int32, m).tolist(), ctx.download(ob, np.int32, m).tolist()))
    assert m == len(want) and got == want
    sel = np.sort(rng.choice(len(probe), 5000, replace=False)).astype(np.int32)
    ds = ctx.upload(sel)
    want_s = [(int(i), r) for i in sel for r in range(first[i], last[i])]
    op2, ob2, m2 = hip.join_sorted_pairs(ctx, db, nb, dp, ds, len(sel), len(want_s))
    assert m2 == len(want_s) and list(zip(ctx.download(op2, np.int32, m2).tolist(), ctx.download(ob2, np.int32, m2).tolist())) == want_s
    with pytest.raises(hip.PlanHipError) as e:
        hip.join_sorted_pairs(ctx, db, nb, dp, None, len(probe), 100)
    assert e.value.code == hip.PH_ECAPACITY
    for q in (op, ob, op2, ob2, ds):
        ctx.free(q)
    db.free(); dp.free()
    # ---- run lookup
    nparts, c = 50_000, 4
    pk = np.repeat(np.arange(7, 7 + nparts, dtype=np.int32), c)
    sk = np.empty(nparts * c, np.int32)
    for j in range(c):
        sk[j::c] = (np.arange(nparts) * 3 + j * 1250) % 5000 + 1     # four distinct suppliers per part
    t = hip.Table(ctx, [(hip.PH_I32, pk), (hip.PH_I32, sk)], nparts * c)
    assert t.col_run_len(0) == c and t.col_run_len(1) == 0
    rows = rng.integers(0, nparts * c, 100_000)
    p1, p2 = pk[rows].copy(), sk[rows].copy()
    p2[:500] = 6000                                                  # no such supplier
    p1[500:600] = 3                                                  # below the range
    p1[600:700] = 7 + nparts                                         # above it
    valid = np.ones(len(rows), bool); valid[700:800] = False
    d1 = hip.DevColumn(ctx, hip.PH_I32, p1, validity=np.packbits(valid, bitorder="little"))
    d2 = hip.DevColumn(ctx, hip.PH_I32, p2)
    got = ctx.download(hip.join_run_lookup(ctx, t.col(1), nparts * c, 7, c, [d1, d2], None, len(rows)), np.int32, len(rows))
    want = rows.astype(np.int32).copy(); want[:800] = -1
    assert np.array_equal(got, want)
    ctx.check_deferred()
    strict = hip.join_run_lookup(ctx, t.col(1), nparts * c, 7, c, [d1, d2], None, len(rows), strict=True)
    with pytest.raises(hip.PlanHipError) as e:
        ctx.download(strict, np.int32, 4)
    assert e.value.code == hip.PH_ECONSTRAINT
    pk2 = pk.copy(); pk2[1000] = pk2[999]                            # one run of five, one of three: not the shape
    t2 = hip.Table(ctx, [(hip.PH_I32, np.sort(pk2))], nparts * c)
    assert t2.col_run_len(0) == 0
    t.free(); t2.free(); d1.free(); d2.free()


def test_count_by_key_is_left_join_plus_count(ctx):
    """ph_count_by_key against numpy: children per parent with child / parent selections, NULL child keys, keys outside the range, parents
    without children (count 0, validity bit clear: the NULL of count() over the NULL-extended row), int32 and int64 keys"""
    rng = np.random.default_rng(77)
    for dt, typ in ((np.int32, hip.PH_I32), (np.int64, hip.PH_I64)):
        nparent, nchild = 100_003, 1_000_000
        pkeys = (np.arange(nparent) + 5).astype(dt)
        ckeys = rng.integers(0, nparent + 20, nchild).astype(dt)          # some below 5 / above the parents' keys
        ckeys[rng.integers(0, nchild, 50_000)] = 17                        # a hot key
        cvalid = rng.random(nchild) > 0.01
        dc = hip.DevColumn(ctx, typ, ckeys, validity=np.packbits(cvalid, bitorder="little"))
        dp = hip.DevColumn(ctx, typ, pkeys)
        csel = np.sort(rng.choice(nchild, 700_000, replace=False)).astype(np.int32)
        psel = np.sort(rng.choice(nparent, 60_000, replace=False)).astype(np.int32)
        for cs, ps in ((None, None), (csel, psel)):
            crow = np.arange(nchild) if cs is None else cs
            prow = np.arange(nparent) if ps is None else ps
            live = ckeys[crow][cvalid[crow]]
            hist = np.bincount(live[(live >= 5) & (live < 5 + nparent)].astype(np.int64) - 5, minlength=nparent)
            want = hist[pkeys[prow].astype(np.int64) - 5]
            dcs = ctx.upload(cs) if cs is not None else None
            dps = ctx.upload(ps) if ps is not None else None
            out, val = hip.count_by_key(ctx, dc, dcs, len(crow), 5, nparent, dp, dps, len(prow))
            got = ctx.download(out, np.int64, len(prow))
            bits = np.unpackbits(ctx.download(val, np.uint8, (len(prow) + 7) // 8), bitorder="little")[:len(prow)].astype(bool)
            assert np.array_equal(got, want) and np.array_equal(bits, want > 0) and (want == 0).any()
            for q in (out, val, dcs, dps):
                if q is not None:
                    ctx.free(q)
        dc.free(); dp.free()


def test_merge_lookup_equals_table_lookup(ctx):
    """ph_merge_lookup (both sides ordered by the key, no table) answers like ph_join_build + ph_join_lookup:
    dense clustered probes (blocks stream their slice of the build keys through LDS), very sparse probes (blocks
    search the column), a probe selection, int32 keys, absent keys, keys below / above every build key, empty
    sides; strict misses and either broken order are the deferred PH_ECONSTRAINT."""
    rng = np.random.default_rng(101)
    nb = 2_000_000
    i = np.arange(nb, dtype=np.int64)
    bk = (i // 8) * 32 + i % 8 + 1                                     # the o_orderkey pattern
    for dt, typ in ((np.int64, hip.PH_I64), (np.int32, hip.PH_I32)):
        b = bk.astype(dt)
        db = hip.DevColumn(ctx, typ, b)
        jt = hip.Join(ctx, [db], None, nb, key_range=(int(b[0]), int(b[-1])))
        dense = np.sort(np.concatenate([b[rng.integers(0, nb, 900_000)], rng.integers(-5, int(b[-1]) + 50, 100_000).astype(dt)]))
        sparse = np.sort(np.concatenate([b[rng.integers(0, nb, 1500)], np.array([-7, int(b[-1]) + 3], dtype=dt)]))
        for p in (dense, sparse):
            dp = hip.DevColumn(ctx, typ, p)
            got = ctx.download(hip.merge_lookup(ctx, db, nb, dp, None, len(p)), np.int32, len(p))
            want = ctx.download(jt.lookup([dp], None, len(p)), np.int32, len(p))
            assert np.array_equal(got, want) and (got >= 0).sum() > 1000 and (got < 0).sum() >= 2
            ctx.check_deferred()
            dp.free()
        # a selection that keeps the order; strict lookups of present keys only
        dp = hip.DevColumn(ctx, typ, dense)
        sel = np.nonzero(np.isin(dense, b))[0].astype(np.int32)[::3]
        ds = ctx.upload(sel)
        got = ctx.download(hip.merge_lookup(ctx, db, nb, dp, ds, len(sel), strict=True), np.int32, len(sel))
        assert np.array_equal(b[got], dense[sel])
        ctx.check_deferred()
        # strict with a miss, probes out of order, build out of order: deferred errors
        miss = hip.merge_lookup(ctx, db, nb, dp, None, len(dense), strict=True)
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(miss, np.int32, 4)
        assert e.value.code == hip.PH_ECONSTRAINT
        bad = dense.copy(); bad[500_000], bad[500_001] = dense[-1], dense[0]
        dbad = hip.DevColumn(ctx, typ, bad)
        r = hip.merge_lookup(ctx, db, nb, dbad, None, len(bad))
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(r, np.int32, 4)
        assert e.value.code == hip.PH_ECONSTRAINT
        b2 = b.copy(); b2[1_000_000], b2[1_000_001] = b[1_000_001], b[1_000_000]
        db2 = hip.DevColumn(ctx, typ, b2)
        r = hip.merge_lookup(ctx, db2, nb, dp, None, len(dense))
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(r, np.int32, 4)
        assert e.value.code == hip.PH_ECONSTRAINT
        # sparse probes (the block searches the column itself) against a build column with a duplicated pair / a
        # descending pair exactly where a search ends: the local order check raises the deferred error
        dsp = hip.DevColumn(ctx, typ, sparse)
        hitpos = int(np.searchsorted(b, sparse[len(sparse) // 2]))
        for kind in ("dup", "desc"):
            b3 = b.copy()
            if kind == "dup":
                b3[hitpos + 1] = b3[hitpos]
            else:
                b3[hitpos], b3[hitpos + 1] = b[hitpos + 1], b[hitpos]
            db3 = hip.DevColumn(ctx, typ, b3)
            r = hip.merge_lookup(ctx, db3, nb, dsp, None, len(sparse))
            with pytest.raises(hip.PlanHipError) as e:
                ctx.download(r, np.int32, 4)
            assert e.value.code == hip.PH_ECONSTRAINT
            db3.free()
        dsp.free()
        # empty sides
        assert (ctx.download(hip.merge_lookup(ctx, db, 0, dp, None, 1000), np.int32, 1000) == -1).all()
        hip.merge_lookup(ctx, db, nb, dp, None, 0)
        ctx.check_deferred()
        jt.free(); ctx.free(ds)
        for c in (db, dp, dbad, db2):
            c.free()


def test_join_small_dense_table_declared_sorted_unique(ctx):
    """A small dense table (a dimension's primary key: 100 k rows) over keys declared sorted and unique is built by
    the gated sorted fill with an always-true gate (no atomics): lookups, marks, pairs and the count equal the
    undeclared build (direct_small_kernel); 4- and 8-byte keys; a false claim is the deferred PH_ECONSTRAINT."""
    rng = np.random.default_rng(109)
    n = 100_000
    for dt, typ in ((np.int32, hip.PH_I32), (np.int64, hip.PH_I64)):
        keys = (np.sort(rng.choice(3 * n, n, replace=False)) + 11).astype(dt)   # 3 slots per row
        rngk = (11, 3 * n + 10)
        p = rng.integers(0, 3 * n + 40, 700_000).astype(dt)
        dk, dp = hip.DevColumn(ctx, typ, keys), hip.DevColumn(ctx, typ, p)
        jd = hip.Join(ctx, [dk], None, n, key_range=rngk, sorted_unique=True)
        ju = hip.Join(ctx, [dk], None, n, key_range=rngk)
        assert jd.kind == ju.kind == "direct" and jd.count() == ju.count() == n
        a = ctx.download(jd.lookup([dp], None, len(p)), np.int32, len(p))
        b = ctx.download(ju.lookup([dp], None, len(p)), np.int32, len(p))
        assert np.array_equal(a, b) and 200_000 < (a >= 0).sum() < 260_000
        assert np.array_equal(ctx.download(jd.probe_mark([dp], None, len(p)), np.uint8, len(p)) != 0, a >= 0)
        md, pd_, bd = jd.probe_inner([dp], None, len(p), len(p))
        assert md == int((a >= 0).sum()) and np.array_equal(ctx.download(bd, np.int32, md), a[a >= 0])
        ctx.check_deferred()
        jd.free(); ju.free()
        bad = keys.copy(); bad[40_000], bad[40_001] = keys[40_001], keys[40_000]
        dbad = hip.DevColumn(ctx, typ, bad)
        jb = hip.Join(ctx, [dbad], None, n, key_range=rngk, sorted_unique=True)
        r = jb.lookup([dp], None, len(p))
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(r, np.int32, 4)
        assert e.value.code == hip.PH_ECONSTRAINT
        jb.free()
        for x in (pd_, bd):
            ctx.free(x)
        for c in (dk, dp, dbad):
            c.free()


def test_join_build_where_equals_filter_then_build(ctx):
    """ph_join_build_where (Filter -> build fused into a direct table): pairs, marks, lookups and the
    row count equal ph_filter_select + ph_join_build_range over the same rows (build row ids are rows
    of the unfiltered input in both); duplicates among the passing rows; a predicate nothing passes;
    PH_EUNSUPPORTED (None) where the range gives no direct table."""
    rng = np.random.default_rng(51)
    nb, np_ = 600_000, 900_000
    bk = (rng.permutation(1_000_000)[:nb] + 10).astype(np.int32)
    bk[:50] = bk[100:150]                                             # duplicate keys
    seg = rng.integers(0, 5, nb).astype(np.uint8)
    pk = rng.integers(0, 1_000_100, np_).astype(np.int32)
    dbk, dseg, dpk = hip.DevColumn(ctx, hip.PH_I32, bk), hip.DevColumn(ctx, hip.PH_CODE8, seg), hip.DevColumn(ctx, hip.PH_I32, pk)
    rngk = (10, 1_000_009)
    for code in (2, 9):                                               # 9: nothing passes
        jw = hip.Join.build_where(ctx, [dbk], dseg, hip.PH_EQ, hip.const(hip.PH_I32, i=code), None, nb, rngk)
        assert jw is not None and jw.kind == "direct"
        fs, fn = hip.filter_select(ctx, dseg, nb, hip.PH_EQ, hip.const(hip.PH_I32, i=code))
        jf = hip.Join(ctx, [dbk], fs, fn, key_range=rngk)
        assert jw.count() == jf.count() == int((seg == code).sum())
        mw, pw, bw = jw.probe_inner([dpk], None, np_, 2 * np_)
        mf, pf, bf = jf.probe_inner([dpk], None, np_, 2 * np_)
        assert mw == mf
        a = np.stack([ctx.download(pw, np.int32, mw), ctx.download(bw, np.int32, mw)], 1)
        b = np.stack([ctx.download(pf, np.int32, mf), ctx.download(bf, np.int32, mf)], 1)
        assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))])
        if mw:
            assert np.all(seg[a[:, 1]] == code) and np.array_equal(bk[a[:, 1]], pk[a[:, 0]])
        assert np.array_equal(ctx.download(jw.probe_mark([dpk], None, np_), np.uint8, np_), ctx.download(jf.probe_mark([dpk], None, np_), np.uint8, np_))
        jw.free(); jf.free()
    assert hip.Join.build_where(ctx, [dbk], dseg, hip.PH_EQ, hip.const(hip.PH_I32, i=2), None, nb, (0, 2**31)) is None
    for c in (dbk, dseg, dpk):
        c.free()


def test_direct_table_edge_cases(ctx):
    """Empty and tiny inputs through the direct-table entry points: zero build rows, zero probe rows, one
    row, a range of one value, probe keys all outside the range, ph_join_build_where over zero rows,
    residual flags that reject everything."""
    one = hip.DevColumn(ctx, hip.PH_I64, np.array([42], np.int64))
    some = hip.DevColumn(ctx, hip.PH_I64, np.array([40, 41, 42, 43, 42, 1000], np.int64))
    j0 = hip.Join(ctx, [one], None, 0, key_range=(42, 42))                 # no build rows: whatever table, no matches
    assert j0.count() == 0
    m, a, b = j0.probe_inner([some], None, 6, 16)
    assert m == 0 and ctx.download(j0.probe_mark([some], None, 6), np.uint8, 6).sum() == 0
    j0.free()
    j1 = hip.Join(ctx, [one], None, 1, key_range=(42, 42))                 # a range of one value
    assert j1.kind == "direct" and j1.count() == 1
    m, a, b = j1.probe_inner([some], None, 6, 16)
    assert m == 2 and ctx.download(a, np.int32, 2).tolist() == [2, 4] and ctx.download(b, np.int32, 2).tolist() == [0, 0]
    assert ctx.download(j1.lookup([some], None, 6), np.int32, 6).tolist() == [-1, -1, 0, -1, 0, -1]
    m, a, b = j1.probe_inner([some], None, 0, 16)                         # zero probe rows
    assert m == 0
    far = hip.DevColumn(ctx, hip.PH_I64, np.array([-5, 10**12, 41, 43], np.int64))
    assert ctx.download(j1.probe_mark([far], None, 4), np.uint8, 4).tolist() == [0, 0, 0, 0]
    flags0 = ctx.upload(np.zeros(1, np.uint8))
    r = j1.probe_inner_residual([some], None, 0, None, flags0, None, 6, 16)
    assert r is not None and r[0] == 0
    flags1 = ctx.upload(np.ones(1, np.uint8))
    r = j1.probe_inner_residual([some], None, 0, None, flags1, None, 6, 16)
    assert r is not None and r[0] == 2
    j1.free()
    seg = hip.DevColumn(ctx, hip.PH_CODE8, np.array([1], np.uint8))
    jw = hip.Join.build_where(ctx, [one], seg, hip.PH_EQ, hip.const(hip.PH_I32, i=1), None, 0, (42, 42))
    assert jw is None or jw.count() == 0                                   # zero rows: no direct table is fine, so is an empty one
    if jw is not None:
        jw.free()
    for c in (one, some, far, seg):
        c.free()


def test_async_counts(ctx):
    """ph_ctx_set_async_counts: filter_select / probe_inner return at once with the count pending (-1)
    and ph_ctx_wait_counts fills all of them in; an overflowing pair list is reported by the wait;
    switching the mode off waits too."""
    rng = np.random.default_rng(81)
    n = 300_000
    v = rng.integers(0, 100, n).astype(np.int32)
    dv = hip.DevColumn(ctx, hip.PH_I32, v)
    bk = np.arange(1000, dtype=np.int32)
    dbk = hip.DevColumn(ctx, hip.PH_I32, bk)
    j = hip.Join(ctx, [dbk], None, len(bk))
    ctx.set_async_counts(True)
    try:
        s1, c1 = hip.filter_select(ctx, dv, n, hip.PH_LT, hip.const(hip.PH_I32, i=10), defer=True)
        s2, c2 = hip.filter_select(ctx, dv, n, hip.PH_GE, hip.const(hip.PH_I32, i=90), defer=True)
        m, op, ob = j.probe_inner([dv], None, n, n, defer=True)
        assert c1.value == c2.value == m.value == -1
        ctx.wait_counts()
        assert c1.value == int((v < 10).sum()) and c2.value == int((v >= 90).sum()) and m.value == n
        assert np.array_equal(ctx.download(s1, np.int32, c1.value), np.flatnonzero(v < 10))
        m2, op2, ob2 = j.probe_inner([dv], None, n, 1000, defer=True)      # capacity too small: reported by the wait
        with pytest.raises(hip.PlanHipError) as e:
            ctx.wait_counts()
        assert e.value.code == hip.PH_ECAPACITY and m2.value == n
        s3, c3 = hip.filter_select(ctx, dv, n, hip.PH_EQ, hip.const(hip.PH_I32, i=5), defer=True)
    finally:
        ctx.set_async_counts(False)                                        # waits for c3
    assert c3.value == int((v == 5).sum())
    s4, c4 = hip.filter_select(ctx, dv, n, hip.PH_EQ, hip.const(hip.PH_I32, i=5))
    assert c4 == c3.value
    j.free(); dv.free(); dbk.free()


def test_join_mark_where_lds_staged_bitmap(ctx, monkeypatch):
    """Big probe sides run ph_join_probe_mark_where with the occupancy bitmap staged in LDS: the bitmap itself
    (range <= 2^20 slots) or its folded image confirmed in L2 (here 1.5 M and 5 M slots); the flags equal the plain
    kernel's (PH_JOIN_MARK_LDS=0) and numpy, 4- and 8-byte keys, a ragged tail, keys outside the range."""
    rng = np.random.default_rng(103)
    no = 1_500_037
    odate = rng.integers(9000, 9200, no).astype(np.int32)
    dodate = hip.DevColumn(ctx, hip.PH_DATE, odate)
    cut = hip.const(hip.PH_DATE, i=9100)
    for dt, ht, span in ((np.int32, hip.PH_I32, 900_000), (np.int32, hip.PH_I32, 1_500_000), (np.int64, hip.PH_I64, 5_000_000)):
        nc = span // 5
        ckeys = np.sort(rng.choice(span, nc, replace=False) + 1).astype(dt)
        dck = hip.DevColumn(ctx, ht, ckeys)
        jc = hip.Join(ctx, [dck], None, nc, key_range=(1, span))
        assert jc.kind == "direct"
        ocust = rng.integers(-3, span + 40, no).astype(dt)
        docust = hip.DevColumn(ctx, ht, ocust)
        f = jc.probe_mark_where([docust], dodate, hip.PH_LT, cut, no)
        assert f is not None
        got = ctx.download(f, np.uint8, no)
        monkeypatch.setenv("PH_JOIN_MARK_LDS", "0")
        f0 = jc.probe_mark_where([docust], dodate, hip.PH_LT, cut, no)
        monkeypatch.delenv("PH_JOIN_MARK_LDS")
        plain = ctx.download(f0, np.uint8, no)
        want = (np.isin(ocust, ckeys) & (odate < 9100)).astype(np.uint8)
        assert np.array_equal(got, want) and np.array_equal(plain, want) and 50_000 < want.sum() < no // 4
        ctx.free(f); ctx.free(f0)
        jc.free(); dck.free(); docust.free()
    dodate.free()


def test_join_mark_where_and_residual_probe(ctx):
    """ph_join_probe_mark_where (Filter -> semi-join mark in one pass) equals filter_select + probe_mark,
    and ph_join_probe_inner_residual over a table built on ALL rows equals the inner probe of a table
    built on the flagged rows only — unique build keys and duplicate ones (chains are walked with the
    flag tested per row), with and without a probe-side filter, 4- and 8-byte keys."""
    rng = np.random.default_rng(71)
    nc, no, nl = 200_000, 700_000, 900_000
    ckeys = (rng.permutation(400_000)[:nc] + 1).astype(np.int32)
    dck = hip.DevColumn(ctx, hip.PH_I32, ckeys)
    jc = hip.Join(ctx, [dck], None, nc, key_range=(1, 400_000))
    assert jc.kind == "direct"
    ocust = rng.integers(1, 400_001, no).astype(np.int32)
    odate = rng.integers(9000, 9200, no).astype(np.int32)
    docust, dodate = hip.DevColumn(ctx, hip.PH_I32, ocust), hip.DevColumn(ctx, hip.PH_DATE, odate)
    cut = hip.const(hip.PH_DATE, i=9100)
    f = jc.probe_mark_where([docust], dodate, hip.PH_LT, cut, no)
    assert f is not None
    flags = ctx.download(f, np.uint8, no)
    want = (np.isin(ocust, ckeys) & (odate < 9100)).astype(np.uint8)
    assert np.array_equal(flags, want)
    for dt, ht, keys_of in ((np.int64, hip.PH_I64, "unique"), (np.int32, hip.PH_I32, "dups")):
        if keys_of == "unique":
            okey = (np.arange(no) * 3 + 10).astype(dt)               # sorted unique (sorted fill, range 2.1 M: bitmap)
        else:
            okey = np.sort(rng.integers(0, 250_000, no)).astype(dt)  # ~3 rows per key
        lkey = rng.integers(0, int(okey.max()) + 50, nl).astype(dt)
        lship = rng.integers(9000, 9200, nl).astype(np.int32)
        dok, dlk, dls = hip.DevColumn(ctx, ht, okey), hip.DevColumn(ctx, ht, lkey), hip.DevColumn(ctx, hip.PH_DATE, lship)
        rngk = (int(okey.min()), int(okey.max()))
        jall = hip.Join(ctx, [dok], None, no, key_range=rngk)
        assert jall.kind == "direct"
        sel = np.flatnonzero(flags).astype(np.int32)
        jsel = hip.Join(ctx, [dok], ctx.upload(sel), len(sel), key_range=rngk)
        for with_where in (False, True):
            if with_where:
                got = jall.probe_inner_residual([dlk], dls, hip.PH_GT, cut, f, None, nl, 4 * nl)
                ref = jsel.probe_inner_where([dlk], dls, hip.PH_GT, cut, None, nl, 4 * nl)
            else:
                got = jall.probe_inner_residual([dlk], None, 0, None, f, None, nl, 4 * nl)
                ref = jsel.probe_inner([dlk], None, nl, 4 * nl)
            assert got is not None and ref is not None and got[0] == ref[0] > 0
            a = np.stack([ctx.download(got[1], np.int32, got[0]), ctx.download(got[2], np.int32, got[0])], 1)
            b = np.stack([ctx.download(ref[1], np.int32, ref[0]), ctx.download(ref[2], np.int32, ref[0])], 1)
            assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))])
            assert np.all(flags[a[:, 1]] == 1) and np.array_equal(okey[a[:, 1]], lkey[a[:, 0]])
        jall.free(); jsel.free()
        for c in (dok, dlk, dls):
            c.free()
    hj = hip.Join(ctx, [dck], None, nc)                               # a hash table: not fused
    assert hj.probe_mark_where([docust], dodate, hip.PH_LT, cut, no) is None
    hj.free(); jc.free()
    for c in (dck, docust, dodate):
        c.free()


def test_join_fk_probe_hint_takes_node_table(ctx):
    """ph_join_build_ex with PH_JOIN_FK_PROBES: build sides of >= 32 K rows take the node table (no
    Bloom bitmap); lookups, pairs and marks equal the default table's (composite 4-byte keys and one
    8-byte key)."""
    rng = np.random.default_rng(41)
    a = rng.integers(0, 200_000, 300_000).astype(np.int32)
    b = rng.integers(0, 8, 300_000).astype(np.int32)
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    ba, bb = pairs[:, 0].copy(), pairs[:, 1].copy()
    pa = np.concatenate([ba[::3], rng.integers(0, 200_000, 50_000).astype(np.int32)])
    pb = np.concatenate([bb[::3], rng.integers(0, 9, 50_000).astype(np.int32)])
    da, db, dpa, dpb = (hip.DevColumn(ctx, hip.PH_I32, x) for x in (ba, bb, pa, pb))
    jn, jd = hip.Join(ctx, [da, db], None, len(ba), fk_probes=True), hip.Join(ctx, [da, db], None, len(ba))
    assert jn.kind == "nodes" and jd.kind == "chained+bloom" and jn.count() == jd.count() == len(ba)
    n = len(pa)
    assert np.array_equal(ctx.download(jn.lookup([dpa, dpb], None, n), np.int32, n), ctx.download(jd.lookup([dpa, dpb], None, n), np.int32, n))
    assert np.array_equal(ctx.download(jn.probe_mark([dpa, dpb], None, n), np.uint8, n), ctx.download(jd.probe_mark([dpa, dpb], None, n), np.uint8, n))
    mn, pn, bn = jn.probe_inner([dpa, dpb], None, n, n)
    md, pd_, bd = jd.probe_inner([dpa, dpb], None, n, n)
    assert mn == md and np.array_equal(ctx.download(pn, np.int32, mn), ctx.download(pd_, np.int32, md))
    assert np.array_equal(ctx.download(bn, np.int32, mn), ctx.download(bd, np.int32, md))
    jn.free(); jd.free()
    k8 = rng.permutation(500_000)[:100_000].astype(np.int64) * 1000
    d8 = hip.DevColumn(ctx, hip.PH_I64, k8)
    j8 = hip.Join(ctx, [d8], None, len(k8), key_range=(0, int(k8.max())), fk_probes=True)   # sparse range: not direct
    assert j8.kind == "nodes"
    assert np.array_equal(ctx.download(j8.lookup([d8], None, len(k8)), np.int32, len(k8)), np.arange(len(k8)))
    j8.free()
    small = hip.Join(ctx, [d8], None, 1000, fk_probes=True)                                 # below 32 K rows: default tables
    assert small.kind == "chained+bloom"
    small.free()
    for c in (da, db, dpa, dpb, d8):
        c.free()


def test_deferred_errors_and_strict_lookup(ctx):
    """ph_ctx_set_deferred_errors: an overflowing ph_expr_eval returns PH_OK and the NEXT call that
    reads back fails with PH_EOVERFLOW, once; ph_join_lookup_strict reports a missing / duplicated
    build key the same way (PH_ECONSTRAINT) and ph_gather / ph_date_extract stay in bounds for the
    -1 rows it produced; without violations nothing is reported. Off again: errors are immediate."""
    big = hip.DevColumn(ctx, hip.PH_DEC64, np.full(1000, 2**62, np.int64), 2)
    ok = hip.DevColumn(ctx, hip.PH_DEC64, np.arange(1000, dtype=np.int64), 2)
    prog = [hip.X_COL(0), hip.X_COL(0), hip.X_MUL]
    with pytest.raises(hip.PlanHipError) as e:
        hip.expr_eval(ctx, [big], prog, None, 1000)
    assert e.value.code == hip.PH_EOVERFLOW
    ctx.set_deferred_errors(True)
    try:
        out, _ = hip.expr_eval(ctx, [ok], prog, None, 1000)            # no overflow: nothing pending afterwards
        assert ctx.download(out, np.int64, 3).tolist() == [0, 1, 4]
        out2, _ = hip.expr_eval(ctx, [big], prog, None, 1000)          # overflow: reported by the next read-back
        with pytest.raises(hip.PlanHipError) as e:
            ctx.download(out2, np.int64, 1)
        assert e.value.code == hip.PH_EOVERFLOW and "deferred" in str(e.value)
        assert ctx.download(out, np.int64, 2).tolist() == [0, 1]       # reported once
        ctx.check_deferred()
        # strict lookup: build keys 0..99 except 7, probe 0..99 -> one miss
        bk = np.array([k for k in range(100) if k != 7], np.int32)
        pk = np.arange(100, dtype=np.int32)
        dbk, dpk = hip.DevColumn(ctx, hip.PH_I32, bk), hip.DevColumn(ctx, hip.PH_I32, pk)
        dates = hip.DevColumn(ctx, hip.PH_DATE, np.arange(99, dtype=np.int32) + 9000)
        for key_range in (None, (0, 99)):
            j = hip.Join(ctx, [dbk], None, len(bk), key_range=key_range)
            rows = j.lookup_strict([dpk], None, 100)
            g = hip.gather(ctx, dbk, rows, 100)                         # row -1 reads row 0: in bounds
            y = hip.date_extract(ctx, hip.PH_PART_YEAR, dates, rows, 100)
            with pytest.raises(hip.PlanHipError) as e:
                ctx.download(g, np.int32, 100)
            assert e.value.code == hip.PH_ECONSTRAINT and "1 probe rows without a match" in str(e.value)
            got = ctx.download(rows, np.int32, 100)
            assert got[7] == -1 and np.array_equal(bk[got[got >= 0]], pk[got >= 0])
            rows2 = j.lookup_strict([dbk], None, len(bk))               # every key present: clean
            assert np.array_equal(ctx.download(rows2, np.int32, len(bk)), np.arange(len(bk)))
            ctx.check_deferred()
            j.free()
            for q in (rows, rows2, g, y):
                ctx.free(q)
        for c in (dbk, dpk, dates):
            c.free()
    finally:
        ctx.set_deferred_errors(False)
    with pytest.raises(hip.PlanHipError):
        hip.expr_eval(ctx, [big], prog, None, 1000)
    big.free(); ok.free()


def test_q3_pipeline_operator_granular(ctx, sf001):
    """Q3 assembled from the operator kernels: filter -> join -> join -> expr -> group by."""
    L, Od, C = sf001["lineitem"], sf001["orders"], sf001["customer"]
    date = tpchgen.days(1995, 3, 29)
    nl, no, nc = len(L["l_orderkey"]), len(Od["o_orderkey"]), len(C["c_custkey"])
    seg = hip.DevColumn(ctx, hip.PH_CODE8, C["c_mktsegment"])
    ck = hip.DevColumn(ctx, hip.PH_I32, C["c_custkey"])
    oc = hip.DevColumn(ctx, hip.PH_I32, Od["o_custkey"]); od = hip.DevColumn(ctx, hip.PH_DATE, Od["o_orderdate"])
    ok = hip.DevColumn(ctx, hip.PH_I64, Od["o_orderkey"]); osp = hip.DevColumn(ctx, hip.PH_I32, Od["o_shippriority"])
    lk = hip.DevColumn(ctx, hip.PH_I64, L["l_orderkey"]); ls = hip.DevColumn(ctx, hip.PH_DATE, L["l_shipdate"])
    le = hip.DevColumn(ctx, hip.PH_DEC64, L["l_extendedprice"], 2); ld = hip.DevColumn(ctx, hip.PH_DEC64, L["l_discount"], 2)
    cs, cn = hip.filter_select(ctx, seg, nc, hip.PH_EQ, hip.const(hip.PH_I32, i=O.SEG.index("HOUSEHOLD")))
    j1 = hip.Join(ctx, [ck], cs, cn)
    os_, on = hip.filter_select(ctx, od, no, hip.PH_LT, hip.const(hip.PH_DATE, i=date))
    m1, p1, _b1 = j1.probe_inner([oc], os_, on, on)
    j2 = hip.Join(ctx, [ok], p1, m1)            # build on the surviving orders rows
    lsel, ln = hip.filter_select(ctx, ls, nl, hip.PH_GT, hip.const(hip.PH_DATE, i=date))
    m2, lrow, orow = j2.probe_inner([lk], lsel, ln, ln)
    rev, _ = hip.expr_eval(ctx, [le, ld], [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL], lrow, m2)
    gk = hip.gather(ctx, lk, lrow, m2); gd = hip.gather(ctx, od, orow, m2); gp = hip.gather(ctx, osp, orow, m2)
    mk = lambda t, p, sc=0: hip.Col(t, sc, p, None, None, 0)
    agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_DATE, hip.PH_I32], [(hip.PH_A_SUM, 0)], 1024)
    agg.sink([mk(hip.PH_I64, gk), mk(hip.PH_DATE, gd), mk(hip.PH_I32, gp)], [mk(hip.PH_DEC64, rev, 4)], None, m2,
             positional=True)
    r = agg.finalize()
    n, rows = O.q3(sf001, "HOUSEHOLD", date)
    want = {(rows[i].l_orderkey, rows[i].o_orderdate, rows[i].o_shippriority): rows[i].revenue.unscaled(4)
            for i in range(n)}
    got = {tuple(int(x) for x in r["keys"][g]): r["sum"][g][0] for g in range(r["ngroups"])}
    assert got == want and n > 50


# ------------------------------------------------------------------ gather / partition

def test_gather_and_partition(ctx):
    rng = np.random.default_rng(31)
    n = 100_000
    for ht, dt in ((hip.PH_CODE8, np.uint8), (hip.PH_I32, np.int32), (hip.PH_I64, np.int64)):
        a = rng.integers(0, 200, n).astype(dt)
        idx = rng.integers(0, n, 7777).astype(np.int32)
        d = hip.DevColumn(ctx, ht, a)
        out = hip.gather(ctx, d, ctx.upload(idx), len(idx))
        assert np.array_equal(ctx.download(out, dt, len(idx)), a[idx])
        d.free()
    keys = rng.integers(0, 50_000, n).astype(np.int64)
    dk = hip.DevColumn(ctx, hip.PH_I64, keys)
    for nparts in (1, 2, 8):
        counts, perm = hip.partition(ctx, dk, None, n, nparts)
        p = ctx.download(perm, np.int32, n)
        assert sum(counts) == n and sorted(p.tolist()) == list(range(n))   # a permutation
        start = 0
        owner = {}
        for part, c in enumerate(counts):
            for kv in np.unique(keys[p[start:start + c]]):
                assert owner.setdefault(int(kv), part) == part                # a key lives in one partition
            start += c
    dk.free()


def test_agg_topk_preselection(ctx):
    """ph_agg_topk returns exactly the groups whose aggregate is >= (<=) the k-th best value."""
    rng = np.random.default_rng(41)
    n = 400_000
    k = rng.integers(0, 90_000, n).astype(np.int64)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    v[rng.integers(0, n, 5000)] = 0          # ties around zero
    dk, dv = hip.DevColumn(ctx, hip.PH_I64, k), hip.DevColumn(ctx, hip.PH_I64, v)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)])
    agg.sink([dk], [dv], None, n)
    full = agg.finalize(python_ints=False)
    sums = full["sum_lo"][:, 0].view(np.int64)
    for kk, desc in [(10, True), (10, False), (1, True), (1000, True), (10**7, True)]:
        r = agg.topk(0, kk, descending=desc, cap=full["ngroups"])
        order = np.sort(sums)[::-1] if desc else np.sort(sums)
        kth = order[min(kk, len(order)) - 1]
        want = set(full["keys"][:, 0][(sums >= kth) if desc else (sums <= kth)].tolist())
        assert set(r["keys"][:, 0].tolist()) == want and r["ngroups"] == len(want)
        assert np.all(np.diff(r["first_row"]) > 0)
        got_sum = dict(zip(r["keys"][:, 0].tolist(), r["sum_lo"][:, 0].view(np.int64).tolist()))
        for key, s_ in zip(full["keys"][:, 0].tolist(), sums.tolist()):
            if key in got_sum:
                assert got_sum[key] == s_
    # ORDER BY count(*) [DESC] LIMIT k (Q21's numwait): COUNT aggregates rank by their count word
    counts = full["count"][:, 1].astype(np.int64)
    for kk, desc in [(10, True), (100, False), (1, True)]:
        r = agg.topk(1, kk, descending=desc, cap=full["ngroups"])
        order = np.sort(counts)[::-1] if desc else np.sort(counts)
        kth = order[min(kk, len(order)) - 1]
        want = set(full["keys"][:, 0][(counts >= kth) if desc else (counts <= kth)].tolist())
        assert set(r["keys"][:, 0].tolist()) == want and r["ngroups"] == len(want)
    agg.free(); dk.free(); dv.free()


def test_agg_topk_many_groups_prunes_candidates(ctx):
    """ph_agg_topk over 1.5 M groups: thousands of workgroups each hand in their k best (more candidates than the last workgroup holds in LDS);
    it prunes them with the smallest of the workgroups' own k-th keys before it selects — same groups as a full sort, ties of the k-th
    value included, descending and ascending, SUM and COUNT(*)"""
    rng = np.random.default_rng(47)
    n = 4_000_000
    k = rng.integers(0, 1_500_000, n).astype(np.int64)
    v = rng.integers(-10**5, 10**5, n).astype(np.int64)
    dk, dv = hip.DevColumn(ctx, hip.PH_I64, k), hip.DevColumn(ctx, hip.PH_I64, v)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 1_500_000)
    agg.sink([dk], [dv], None, n)
    full = agg.finalize(python_ints=False)
    sums = full["sum_lo"][:, 0].view(np.int64)
    counts = full["count"][:, 1].astype(np.int64)
    for a, vals in ((0, sums), (1, counts)):
        for kk, desc in [(20, True), (100, False), (3, True)]:
            r = agg.topk(a, kk, descending=desc, cap=full["ngroups"])
            order = np.sort(vals)[::-1] if desc else np.sort(vals)
            kth = order[kk - 1]
            want = set(full["keys"][:, 0][(vals >= kth) if desc else (vals <= kth)].tolist())
            assert set(r["keys"][:, 0].tolist()) == want and r["ngroups"] == len(want), (a, kk, desc, r["ngroups"], len(want))
    agg.free(); dk.free(); dv.free()


def test_filter_select_unaligned_column_pointer(ctx):
    """a column view that starts in the middle of an allocation (not 16-byte aligned) must take the
    scalar path and still give the right, ordered selection"""
    rng = np.random.default_rng(77)
    n = 10007
    data = rng.integers(0, 100, n + 3).astype(np.int32)
    d = hip.DevColumn(ctx, hip.PH_I32, data)
    c = d.col()
    c.data = c.data + 4 * 3          # skip 3 values: 12-byte offset
    sel, cnt = hip.filter_select(ctx, c, n, hip.PH_LT, hip.const(hip.PH_I32, i=50))
    want = np.nonzero(data[3:] < 50)[0]
    assert cnt == len(want) and np.array_equal(dl(ctx, sel, np.int32, cnt), want.astype(np.int32))
    d.free()


def test_partition_is_stable_and_device_counts_match(ctx):
    """ph_partition_dev: rows keep their input order inside every partition (the permutation is
    reproducible, so first-seen group order survives an exchange), with and without a selection;
    the device counts equal the host counts of ph_partition."""
    rng = np.random.default_rng(5)
    n = 300_001
    keys = rng.integers(0, 1 << 40, n).astype(np.int64)
    dk = hip.DevColumn(ctx, hip.PH_I64, keys)
    sel = np.sort(rng.choice(n, 123_457, replace=False)).astype(np.int32)
    dsel = ctx.upload(sel)
    for nparts in (2, 3, 8, 64):
        for s, rows in ((None, np.arange(n, dtype=np.int32)), (dsel, sel)):
            m = len(rows)
            cdev, perm = hip.partition_dev(ctx, dk, s, m, nparts)
            counts = ctx.download(cdev, np.int64, nparts)
            p = ctx.download(perm, np.int32, m)
            hc, perm2 = hip.partition(ctx, dk, s, m, nparts)
            assert counts.tolist() == hc and np.array_equal(p, ctx.download(perm2, np.int32, m))   # reproducible
            start = 0
            seen = np.zeros(n, bool)
            for c in counts:
                part = p[start:start + c]
                assert np.all(np.diff(part) > 0)          # input order kept inside the partition
                seen[part] = True
                start += c
            assert start == m and seen[rows].all() and seen.sum() == m
            for q in (cdev, perm, perm2):
                ctx.free(q)
    dk.free()


def test_agg_topk_null_groups_and_avg(ctx):
    """a SUM/MIN/MAX no input ever reached is NULL and NULLs sort first, so such groups
    must always be among the preselected ones; AVG is not ranked by its sum word and is refused."""
    rng = np.random.default_rng(43)
    n = 50_000
    k = rng.integers(0, 2000, n).astype(np.int64)
    v = rng.integers(1, 10**6, n).astype(np.int64)
    valid = np.ones(n, bool)
    valid[np.isin(k, [7, 8, 9])] = False          # three groups whose every input is NULL
    bits = np.packbits(valid, bitorder="little")
    dk = hip.DevColumn(ctx, hip.PH_I64, k)
    dv = hip.DevColumn(ctx, hip.PH_I64, v, validity=bits)
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_MAX, 0), (hip.PH_A_AVG, 0)])
    agg.sink([dk], [dv], None, n)
    full = agg.finalize(python_ints=False)
    nullkeys = set(full["keys"][:, 0][full["count"][:, 0] == 0].tolist())
    assert nullkeys == {7, 8, 9}
    live = full["count"][:, 0] > 0
    for a in (0, 1):
        vals = full["sum_lo"][:, a].view(np.int64)[live]
        for desc in (True, False):
            r = agg.topk(a, 5, descending=desc, cap=full["ngroups"])
            got = set(r["keys"][:, 0].tolist())
            assert nullkeys <= got                     # NULLs first, whatever the direction
            # the 5 best rows are the 3 NULL groups and then the 2 best live ones: all preselected
            order = np.sort(vals)[::-1] if desc else np.sort(vals)
            best2 = set(full["keys"][:, 0][live][(vals >= order[1]) if desc else (vals <= order[1])].tolist())
            assert best2 <= got and len(got) <= 3 + len(best2) + 3
    with pytest.raises(hip.PlanHipError) as e:
        agg.topk(2, 5)
    assert e.value.code == hip.PH_EUNSUPPORTED
    agg.free(); dk.free(); dv.free()


def test_rccl_communicator_single_rank_roundtrip(ctx):
    """The multi-GPU entry points of the C ABI on real hardware with a 1-rank RCCL communicator
    (a one-GPU box cannot hold two RCCL ranks): id -> init, count matrix, the grouped all-to-all
    (self copy), variable all-gather, reductions and an asynchronous all-gather with wait_keep."""
    from plan_amd import dist as pd
    g = pd.RcclGroup(ctx, 1, 0, pd.RcclGroup.unique_id())
    try:
        assert hip.lib().ph_comm_nranks(g.h) == 1 and hip.lib().ph_comm_rank(g.h) == 0
        rng = np.random.default_rng(9)
        n = 70_001
        keys = rng.integers(0, 1 << 30, n).astype(np.int64)
        vals = rng.integers(0, 1 << 20, n).astype(np.int32)
        dk, dv = hip.DevColumn(ctx, hip.PH_I64, keys), hip.DevColumn(ctx, hip.PH_I32, vals)
        cdev, perm = hip.partition_dev(ctx, dk, None, n, 1)
        matrix = g.exchange_counts(cdev)
        assert matrix.tolist() == [[n]]
        sk, sv = hip.gather(ctx, dk, perm, n), hip.gather(ctx, dv, perm, n)
        rk, rv = ctx.alloc(n * 8), ctx.alloc(n * 4)
        g.exchange_columns([sk, sv], [rk, rv], [8, 4], matrix)
        assert np.array_equal(ctx.download(rk, np.int64, n), keys) and np.array_equal(ctx.download(rv, np.int32, n), vals)
        out = ctx.alloc(n * 8)
        import ctypes
        cnt = (hip.i64 * 1)()
        rc = hip.lib().ph_comm_allgather_rows(g.h, dk.data, hip.i64(n), hip.i32(8), out, hip.i64(n), cnt)
        assert rc == 0 and cnt[0] == n and np.array_equal(ctx.download(out, np.int64, n), keys)
        rc = hip.lib().ph_comm_allgather_rows(g.h, dk.data, hip.i64(n), hip.i32(8), out, hip.i64(n - 1), cnt)
        assert rc == hip.PH_ECAPACITY and cnt[0] == n      # the capacity decision is collective (min over the ranks)
        got, counts = g.allgather_rows(dk.data, n, 8)       # the allocating form: sized from the counts
        assert counts == [n] and np.array_equal(ctx.download(got, np.int64, n), keys)
        ctx.free(got)
        # a held deferred error is not reported by the collectives' read-backs, only by check_deferred
        ctx.set_deferred_errors(2)
        bad = hip.merge_lookup(ctx, dk, n, dk, None, n)     # unsorted keys on both sides: the deferred PH_ECONSTRAINT
        assert g.allreduce([3], "max") == [3]
        assert g.exchange_counts(cdev).tolist() == [[n]]
        ctx.download(bad, np.int32, 4)                      # an ordinary read-back stays silent while the error is held
        with pytest.raises(hip.PlanHipError) as e:
            ctx.check_deferred()
        assert e.value.code == hip.PH_ECONSTRAINT
        ctx.set_deferred_errors(False)
        ctx.free(bad)
        assert g.allreduce([5, -7], "sum") == [5, -7] and g.allreduce([5], "max") == [5]
        g.barrier()
        for i in range(6):        # more asynchronous collectives than the event ring holds
            g.wait(keep=1)
            g.allgather(dk.data, out, 4096, async_=True)
        g.wait()
        assert np.array_equal(ctx.download(out, np.int64, 512), keys[:512])
        for q in (cdev, perm, sk, sv, rk, rv, out):
            ctx.free(q)
        dk.free(); dv.free()
    finally:
        g.close()


def test_join_lookup_matches_inner_probe_on_unique_keys(ctx):
    """ph_join_lookup (N:1 lookup probe): for unique build keys out[i] is the build row the pair-
    emitting probe reports for probe row i, -1 where there is none; the stats words count misses and
    multi-matches. One and two key columns, 4- and 8-byte keys, selections on both sides, a Bloom-
    filtered (small) and a plain (large) table."""
    rng = np.random.default_rng(123)
    for nb, npr, dt, ht in ((50_000, 300_000, np.int32, hip.PH_I32), (5_000_000, 400_000, np.int64, hip.PH_I64)):
        bk = rng.permutation(nb * 3).astype(dt)[:nb]            # unique
        pk = rng.integers(0, nb * 3, npr).astype(dt)
        dbk, dpk = hip.DevColumn(ctx, ht, bk), hip.DevColumn(ctx, ht, pk)
        bsel = np.sort(rng.choice(nb, nb // 2, replace=False)).astype(np.int32)
        psel = np.sort(rng.choice(npr, npr // 3, replace=False)).astype(np.int32)
        for bs, ps in ((None, None), (bsel, psel)):
            brows = np.arange(nb) if bs is None else bs
            prows = np.arange(npr) if ps is None else ps
            j = hip.Join(ctx, [dbk], None if bs is None else ctx.upload(bs), len(brows))
            stats = ctx.upload(np.zeros(2, np.int32))
            out = j.lookup([dpk], None if ps is None else ctx.upload(ps), len(prows), stats)
            got = ctx.download(out, np.int32, len(prows))
            where = {int(k): int(r) for k, r in zip(bk[brows], brows)}
            want = np.array([where.get(int(k), -1) for k in pk[prows]], np.int32)
            assert np.array_equal(got, want)
            assert ctx.download(stats, np.int32, 2).tolist() == [int((want < 0).sum()), 0]
            j.free()
        dbk.free(); dpk.free()
    # composite key + a duplicated build key (multi-match is counted, a matching row is still reported)
    a = rng.integers(0, 1000, 20_000).astype(np.int32)
    b = rng.integers(0, 50, 20_000).astype(np.int32)
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    ba, bb = pairs[:, 0].copy(), pairs[:, 1].copy()
    ba = np.concatenate([ba, ba[:7]]); bb = np.concatenate([bb, bb[:7]])      # 7 duplicated keys
    da, db = hip.DevColumn(ctx, hip.PH_I32, ba), hip.DevColumn(ctx, hip.PH_I32, bb)
    pa, pb = hip.DevColumn(ctx, hip.PH_I32, a), hip.DevColumn(ctx, hip.PH_I32, b)
    j = hip.Join(ctx, [da, db], None, len(ba))
    stats = ctx.upload(np.zeros(2, np.int32))
    got = ctx.download(j.lookup([pa, pb], None, len(a), stats), np.int32, len(a))
    assert np.all(got >= 0) and np.array_equal(ba[got], a) and np.array_equal(bb[got], b)
    dup = set(zip(ba[:7].tolist(), bb[:7].tolist()))
    st = ctx.download(stats, np.int32, 2).tolist()
    assert st == [0, sum((x, y) in dup for x, y in zip(a.tolist(), b.tolist()))]
    j.free()
    for c in (da, db, pa, pb):
        c.free()


@pytest.mark.parametrize("form", ["PH_STREAM_AGG_ONE_PASS", "PH_STREAM_AGG_TWO_PASS"])
def test_streaming_aggregate_over_sorted_input_equals_hash_aggregate(ctx, form, monkeypatch):
    """(both forms: the tiles' first group ids from a look-back inside the groups kernel, and from a heads pass + scan)
    ph_agg_sink_sorted: rows ordered by the group-key tuple (runs of 1..40 rows, one run of 100 000
    rows, a run across a 2048-row block boundary; two keys; SUM / AVG / COUNT / MIN / MAX / COUNT(*) with
    a NULL-able argument) give the groups ph_agg_sink gives, record for record and in first-seen order;
    a further sink is refused; rows that are NOT ordered are a deferred PH_ECONSTRAINT."""
    monkeypatch.setenv(form, "1")
    rng = np.random.default_rng(101)
    lens = rng.integers(1, 41, 60_000)
    lens[1000] = 100_000
    k0 = np.repeat(np.arange(len(lens), dtype=np.int64) * 3, lens)
    k1 = np.repeat(rng.integers(0, 5, len(lens)).astype(np.int32), lens)
    n = len(k0)
    v0 = rng.integers(-10**12, 10**12, n).astype(np.int64)
    v1 = rng.integers(0, 1000, n).astype(np.int32)
    valid = rng.random(n) > 0.1
    d0, d1 = hip.DevColumn(ctx, hip.PH_I64, k0), hip.DevColumn(ctx, hip.PH_I32, k1)
    a0 = hip.DevColumn(ctx, hip.PH_DEC64, v0, 2, validity=np.packbits(valid, bitorder="little"))
    a1 = hip.DevColumn(ctx, hip.PH_I32, v1)
    aggs = [(hip.PH_A_SUM, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 1), (hip.PH_A_AVG, 1), (hip.PH_A_COUNT_STAR, -1), (hip.PH_A_COUNT, 0)]
    res = []
    for sorted_form in (True, False):
        agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_I32], aggs, 1024)
        if sorted_form:
            assert agg.sink_sorted([d0, d1], [a0, a1], n)
            with pytest.raises(hip.PlanHipError):
                agg.sink([d0, d1], [a0, a1], None, n)
        else:
            agg.sink([d0, d1], [a0, a1], None, n)
        r = agg.finalize(python_ints=False, room=len(lens))
        res.append({k: np.asarray(r[k]) for k in ("first_row", "keys", "key_null", "sum_lo", "sum_hi", "count")})
        assert r["ngroups"] == len(lens)
        agg.free()
    ctx.check_deferred()
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    assert np.array_equal(res[0]["first_row"], np.concatenate([[0], np.cumsum(lens)[:-1]]))
    # top-k works on the streamed table too (groups whose every input is NULL sort first, in both forms)
    tops = []
    for sorted_form in (True, False):
        agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_I32], [(hip.PH_A_SUM, 1)], 1024)
        if sorted_form:
            assert agg.sink_sorted([d0, d1], [a0, a1], n)
        else:
            agg.sink([d0, d1], [a0, a1], None, n)
        top = agg.topk(0, 5)
        tops.append(sorted(zip(top["keys"][:, 0].tolist(), top["sum_lo"][:, 0].tolist())))
        agg.free()
    sums = np.add.reduceat(v1.astype(np.int64), np.concatenate([[0], np.cumsum(lens)[:-1]]))
    assert tops[0] == tops[1] and {k for k, _ in tops[0]} >= set((np.argsort(-sums, kind="stable")[:1] * 3).tolist())
    # not ordered by the key: the claim is verified on the device
    kb = k0.copy()
    kb[500_000:500_010] = 1
    db = hip.DevColumn(ctx, hip.PH_I64, kb)
    agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_I32], aggs, 1024)
    assert agg.sink_sorted([db, d1], [a0, a1], n)
    with pytest.raises(hip.PlanHipError) as e:
        agg.group_count()
    assert e.value.code == hip.PH_ECONSTRAINT
    agg.free()
    for c in (d0, d1, a0, a1, db):
        c.free()


def test_specialised_sink_equals_generic_sink_and_numpy(ctx):
    """ph_agg_sink calls of >= 2^20 rows run the hiprtc-specialised form of the sink kernel (same
    source, the shape as constants): for several shapes — NULL-able keys and arguments, selections,
    positional arguments, MIN/MAX/AVG/COUNT, low and high cardinality — its groups equal the generic
    kernel's (PH_AGG_JIT=0) word for word, and the sums equal numpy's."""
    import os
    rng = np.random.default_rng(77)
    n = (1 << 20) + 12345
    shapes = [dict(card=4, nullable=False, sel=False), dict(card=175, nullable=True, sel=False),
              dict(card=3000, nullable=False, sel=True), dict(card=400_000, nullable=True, sel=True)]
    for sh in shapes:
        k0 = rng.integers(0, sh["card"], n).astype(np.int32)
        k1 = rng.integers(0, 3, n).astype(np.int64)
        v0 = rng.integers(-10**6, 10**6, n).astype(np.int64)
        v1 = rng.integers(0, 1000, n).astype(np.int32)
        kvalid = rng.random(n) > 0.05 if sh["nullable"] else np.ones(n, bool)
        vvalid = rng.random(n) > 0.1 if sh["nullable"] else np.ones(n, bool)
        bits = lambda m: np.packbits(m, bitorder="little") if sh["nullable"] else None
        d0 = hip.DevColumn(ctx, hip.PH_I32, k0, validity=bits(kvalid)); d1 = hip.DevColumn(ctx, hip.PH_I64, k1)
        a0 = hip.DevColumn(ctx, hip.PH_DEC64, v0, 2, validity=bits(vvalid)); a1 = hip.DevColumn(ctx, hip.PH_I32, v1)
        rows = np.sort(rng.choice(n, n - 1000, replace=False)).astype(np.int32) if sh["sel"] else None
        dsel = ctx.upload(rows) if rows is not None else None
        m = len(rows) if rows is not None else n
        aggs = [(hip.PH_A_SUM, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 1), (hip.PH_A_AVG, 1), (hip.PH_A_COUNT_STAR, -1), (hip.PH_A_COUNT, 0)]
        res = {}
        for mode in ("1", "0"):
            os.environ["PH_AGG_JIT"] = mode
            agg = hip.Agg(ctx, [hip.PH_I32, hip.PH_I64], aggs, 1024)
            agg.sink([d0, d1], [a0, a1], dsel, m)
            r = agg.finalize(python_ints=False)
            order = np.lexsort((r["keys"][:, 1], r["keys"][:, 0], r["key_null"][:, 0]))
            res[mode] = {k: np.asarray(r[k])[order] for k in ("first_row", "keys", "key_null", "sum_lo", "sum_hi", "count")}
            agg.free()
        os.environ.pop("PH_AGG_JIT")
        for k in res["1"]:
            assert np.array_equal(res["1"][k], res["0"][k]), (sh, k)
        # numpy: SUM(v0) and COUNT(*) per group over the sunk rows
        sl = rows if rows is not None else np.arange(n)
        key = np.where(kvalid[sl], k0[sl].astype(np.int64), -1) * 4 + k1[sl]
        uk, inv = np.unique(key, return_inverse=True)
        sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv, np.where(vvalid[sl], v0[sl], 0))
        cnts = np.bincount(inv)
        got_key = np.where(res["1"]["key_null"][:, 0] == 1, -1, res["1"]["keys"][:, 0]) * 4 + res["1"]["keys"][:, 1]
        o2 = np.argsort(got_key)
        assert np.array_equal(got_key[o2], uk)
        assert np.array_equal(res["1"]["sum_lo"][:, 0].view(np.int64)[o2], sums) and np.array_equal(res["1"]["count"][:, 4][o2], cnts)
        for c in (d0, d1, a0, a1):
            c.free()


def test_node_table_join_matches_chained_table(ctx):
    """Large build sides (> 4 M keys) use the node table (partitioned build, 16-byte nodes). With
    PH_JOIN_BIG_MIN lowered the same joins run through both table forms: inner pairs (duplicate
    build keys, misses, selections on both sides, NULL keys), marks and lookups must be identical
    sets — and equal to numpy for the single-key case. One 8-byte key, one 4-byte key, two 4-byte keys."""
    import os
    rng = np.random.default_rng(2024)
    nb, npr = 300_000, 700_000
    cases = []
    k8 = rng.integers(0, 400_000, nb).astype(np.int64)
    cases.append(([(hip.PH_I64, k8)], [(hip.PH_I64, rng.integers(0, 450_000, npr).astype(np.int64))]))
    k4 = rng.integers(0, 400_000, nb).astype(np.int32)
    cases.append(([(hip.PH_I32, k4)], [(hip.PH_I32, rng.integers(0, 450_000, npr).astype(np.int32))]))
    a, b = rng.integers(0, 2000, nb).astype(np.int32), rng.integers(0, 300, nb).astype(np.int32)
    cases.append(([(hip.PH_I32, a), (hip.PH_I32, b)],
                  [(hip.PH_I32, rng.integers(0, 2100, npr).astype(np.int32)), (hip.PH_I32, rng.integers(0, 310, npr).astype(np.int32))]))
    bvalid = rng.random(nb) > 0.03
    pvalid = rng.random(npr) > 0.03
    bsel = np.sort(rng.choice(nb, nb * 2 // 3, replace=False)).astype(np.int32)
    psel = np.sort(rng.choice(npr, npr // 2, replace=False)).astype(np.int32)
    for bcols, pcols in cases:
        for nullable, use_sel in ((False, False), (True, True)):
            bdev = [hip.DevColumn(ctx, t, v, validity=np.packbits(bvalid, bitorder="little") if nullable else None) for t, v in bcols]
            pdev = [hip.DevColumn(ctx, t, v, validity=np.packbits(pvalid, bitorder="little") if nullable else None) for t, v in pcols]
            bs = ctx.upload(bsel) if use_sel else None
            ps = ctx.upload(psel) if use_sel else None
            nbb, npp = (len(bsel), len(psel)) if use_sel else (nb, npr)
            res = {}
            for mode in ("1000", "1000000000"):     # node table / chained table
                os.environ["PH_JOIN_BIG_MIN"] = mode
                j = hip.Join(ctx, bdev, bs, nbb)
                cnt = j.count()
                m, op, ob = j.probe_inner(pdev, ps, npp, npp * 4)
                pairs = np.stack([ctx.download(op, np.int32, m), ctx.download(ob, np.int32, m)], 1)
                assert np.all(np.diff(pairs[:, 0]) >= 0)                       # ordered by probe row
                mark = ctx.download(j.probe_mark(pdev, ps, npp), np.uint8, npp)
                stats = ctx.upload(np.zeros(2, np.int32))
                look = ctx.download(j.lookup(pdev, ps, npp, stats), np.int32, npp)
                st = ctx.download(stats, np.int32, 2).tolist()
                res[mode] = (cnt, set(map(tuple, pairs.tolist())), mark, look >= 0, st)
                j.free()
            os.environ.pop("PH_JOIN_BIG_MIN")
            big, old = res["1000"], res["1000000000"]
            assert big[0] == old[0] and big[1] == old[1] and np.array_equal(big[2], old[2])
            assert np.array_equal(big[3], old[3]) and big[4] == old[4]
            assert np.array_equal(big[2].astype(bool), big[3]) and len(big[1]) > 1000
            if len(bcols) == 1 and not nullable:       # independent check against numpy
                bk, pk = bcols[0][1], pcols[0][1]
                assert np.array_equal(big[2].astype(bool), np.isin(pk, bk))
                uk, cnts = np.unique(bk, return_counts=True)
                assert len(big[1]) == int(cnts[np.searchsorted(uk, pk[np.isin(pk, bk)])].sum())
            for c in bdev + pdev:
                c.free()


def test_generated_expression_kernel_equals_interpreter(ctx):
    """ph_expr_eval batches of >= 2^18 rows run a kernel generated from the RPN (hiprtc); its values,
    result validity and overflow refusal must equal the interpreting kernel's (PH_EXPR_JIT=0), whose
    decimal semantics test_expr_eval_matches_decimal_semantics pins to the oracle."""
    import os
    rng = np.random.default_rng(31)
    n = (1 << 18) + 777
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    b = rng.integers(0, 11, n).astype(np.int64)
    q = rng.integers(1, 51, n).astype(np.int32)
    va, vb = rng.random(n) > 0.05, rng.random(n) > 0.05
    sel = np.sort(rng.choice(n, n - 5000, replace=False)).astype(np.int32)
    progs = [[hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL],                       # a * (1 - b)
             [hip.X_COL(0), hip.X_COL(1), hip.X_COL(2), hip.X_MUL, hip.X_SUB],                          # a - b * q
             [hip.X_COL(0), hip.X_CONST(25, 1), hip.X_ADD, hip.X_COL(2), hip.X_MUL, hip.X_COL(1), hip.X_ADD]]
    for nullable in (False, True):
        bits = lambda m: np.packbits(m, bitorder="little") if nullable else None
        cols = [hip.DevColumn(ctx, hip.PH_DEC64, a, 2, validity=bits(va)), hip.DevColumn(ctx, hip.PH_DEC64, b, 2, validity=bits(vb)),
                hip.DevColumn(ctx, hip.PH_I32, q)]
        for prog in progs:
            for s, m in ((None, n), (ctx.upload(sel), len(sel))):
                got = {}
                for mode in ("1", "0"):
                    os.environ["PH_EXPR_JIT"] = mode
                    out, val = hip.expr_eval(ctx, cols, prog, s, m, want_validity=nullable)
                    got[mode] = (ctx.download(out, np.int64, m), ctx.download(val, np.uint8, (m + 7) // 8) if nullable else None)
                    ctx.free(out)
                os.environ.pop("PH_EXPR_JIT")
                assert np.array_equal(got["1"][0], got["0"][0])
                if nullable:
                    assert np.array_equal(got["1"][1], got["0"][1]) and got["1"][1].sum() > 0
        for c in cols:
            c.free()
    # a product that leaves int64 is refused by both kernels
    big = hip.DevColumn(ctx, hip.PH_DEC64, np.full(n, 4 * 10**9, np.int64), 2)
    for mode in ("1", "0"):
        os.environ["PH_EXPR_JIT"] = mode
        with pytest.raises(hip.PlanHipError) as e:
            hip.expr_eval(ctx, [big], [hip.X_COL(0), hip.X_COL(0), hip.X_MUL, hip.X_COL(0), hip.X_MUL], None, n)
        assert e.value.code == hip.PH_EOVERFLOW
    os.environ.pop("PH_EXPR_JIT")
    big.free()


def test_gather_multi_equals_per_column_gathers(ctx):
    """ph_gather_multi (several columns through one row-id array in one pass) against numpy, for
    1-, 4- and 8-byte columns, 1..8 columns, ragged sizes."""
    rng = np.random.default_rng(8)
    nsrc = 123_457
    srcs = [(hip.PH_CODE8, rng.integers(0, 200, nsrc).astype(np.uint8)), (hip.PH_I32, rng.integers(-9, 9**9, nsrc).astype(np.int32)),
            (hip.PH_I64, rng.integers(-9**15, 9**15, nsrc).astype(np.int64)), (hip.PH_DATE, rng.integers(0, 20000, nsrc).astype(np.int32)),
            (hip.PH_DEC64, rng.integers(0, 10**12, nsrc).astype(np.int64)), (hip.PH_I32, rng.integers(0, 50, nsrc).astype(np.int32)),
            (hip.PH_I64, rng.integers(0, 9, nsrc).astype(np.int64)), (hip.PH_CODE8, rng.integers(0, 3, nsrc).astype(np.uint8))]
    dev = [hip.DevColumn(ctx, t, v) for t, v in srcs]
    for n in (0, 1, 511, 512, 70_001):
        idx = rng.integers(0, nsrc, max(n, 1)).astype(np.int32)
        didx = ctx.upload(idx)
        for k in (1, 3, 8):
            outs = hip.gather_multi(ctx, dev[:k], didx, n)
            for (t, v), o in zip(srcs[:k], outs):
                assert np.array_equal(ctx.download(o, v.dtype, n), v[idx[:n]])
                ctx.free(o)
        ctx.free(didx)
    for c in dev:
        c.free()


def test_single_pass_scan_through_large_selections(ctx):
    """The decoupled look-back scan (one launch, tile states reused across calls by epoch) places
    the rows of every count/scan/write operator: filter selections far beyond 4 x 4096 blocks must
    stay exact and ordered over many consecutive calls of different sizes on one context."""
    rng = np.random.default_rng(12)
    for n in (40_000_000, 17_000_001, 40_000_000, 9_999_999, 33_333_333):
        v = rng.integers(0, 100, n).astype(np.int32)
        d = hip.DevColumn(ctx, hip.PH_I32, v)
        for thr in (1, 37, 99):
            sel, cnt = hip.filter_select(ctx, d, n, hip.PH_LT, hip.const(hip.PH_I32, i=thr))
            want = np.nonzero(v < thr)[0]
            got = ctx.download(sel, np.int32, cnt)
            assert cnt == len(want) and np.array_equal(got, want.astype(np.int32))
            ctx.free(sel)
        d.free()


def _str_col(ctx, strings, nulls=None):
    """list of str -> (DevColumn PH_STR, offsets, bytes)"""
    enc = [s.encode() for s in strings]
    off = np.zeros(len(enc) + 1, np.int32)
    np.cumsum([len(b) for b in enc], out=off[1:])
    byts = np.frombuffer(b"".join(enc), dtype=np.uint8) if off[-1] else np.zeros(1, np.uint8)
    val = None
    if nulls is not None:
        val = np.packbits(~np.asarray(nulls, bool), bitorder="little")
    return hip.DevColumn(ctx, hip.PH_STR, off, validity=val, aux=byts)


def test_string_keys_interning_groups_and_joins(ctx):
    """VERDICT r2 item 6 — VARCHAR keys that are not small dictionaries. ph_strdict_build interns a PH_STR column (hash =
    util.HashBytes, candidates verified byte by byte): equal strings get equal codes, different strings different codes, a code is a
    row holding the string, NULLs get -1. 200 k rows over 30 k distinct strings (so every table slot sees collisions that only the
    byte compare resolves), strings that share long prefixes and differ in the last byte, the empty string, a selection.
    The group-by over the codes equals numpy's over the strings; ph_strdict_lookup resolves a second column against the dictionary
    (absent strings and NULLs: -2), and the integer join over the codes gives the pairs of the string join."""
    rng = np.random.default_rng(2026)
    base = [f"Customer#{i:09d}" for i in range(30_000)] + ["", "a", "aa", "aaa" * 40 + "x", "aaa" * 40 + "y"]
    pick = rng.integers(0, len(base), 200_000)
    strings = [base[i] for i in pick]
    nulls = rng.random(len(strings)) < 0.01
    col = _str_col(ctx, strings, nulls)
    n = len(strings)
    d = hip.StrDict(ctx, col, None, n)
    codes = ctx.download(d.codes, np.int32, n)
    assert np.all(codes[nulls] == -1) and np.all(codes[~nulls] >= 0)
    live = np.nonzero(~nulls)[0]
    assert all(strings[codes[i]] == strings[i] and not nulls[codes[i]] for i in live[:5000])          # a code is a row holding the string
    ids = {}
    for i in live:                                                                                       # equal <-> equal, over all rows
        assert ids.setdefault(strings[i], codes[i]) == codes[i]
    assert len(set(ids.values())) == len(ids)
    # group by the codes: count(*) and sum(v) per string == numpy over the strings
    v = rng.integers(0, 1000, n).astype(np.int64)
    dv = hip.DevColumn(ctx, hip.PH_I64, v)
    key = hip.Col(); key.type, key.data, key.validity = hip.PH_I32, d.codes, col.validity
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, 0)], 40_000)
    agg.sink([key], [dv, dv], None, n)
    r = agg.finalize(room=40_000)
    want = {}
    for i in range(n):
        k = None if nulls[i] else strings[i]
        s, c = want.get(k, (0, 0))
        want[k] = (s + int(v[i]), c + 1)
    got = {}
    for g in range(r["ngroups"]):
        k = None if r["key_null"][g][0] else strings[int(r["keys"][g][0])]
        got[k] = (r["sum"][g][0], int(r["count"][g][1]))
    assert got == want
    # a selection interns only its rows
    sel = np.sort(rng.choice(n, 5000, replace=False)).astype(np.int32)
    ds = ctx.upload(sel)
    d2 = hip.StrDict(ctx, col, ds, len(sel))
    c2 = ctx.download(d2.codes, np.int32, len(sel))
    assert all((c2[j] == -1) if nulls[sel[j]] else strings[c2[j]] == strings[sel[j]] for j in range(len(sel)))
    # lookup of another column: present strings resolve to the same code, absent ones and NULLs to -2
    probe = [base[i] for i in rng.integers(0, len(base), 50_000)] + ["Customer#999999999", "aaa" * 40 + "z", "b"]
    pn = np.zeros(len(probe), bool); pn[7] = True
    pc = _str_col(ctx, probe, pn)
    lk = ctx.download(d.lookup(pc, None, len(probe)), np.int32, len(probe))
    for j, s in enumerate(probe):
        assert lk[j] == (-2 if pn[j] or s not in ids else ids[s]), (j, s)
    # the integer join over the codes = the string join
    bcol = hip.Col(); bcol.type, bcol.data, bcol.validity = hip.PH_I32, d.codes, col.validity
    j = hip.Join(ctx, [bcol], None, n)
    pk = hip.Col(); pk.type, pk.data = hip.PH_I32, d.lookup(pc, None, len(probe))
    m, pp, bb = j.probe_inner([pk], None, len(probe), 4 * n)
    pp, bb = ctx.download(pp, np.int32, m), ctx.download(bb, np.int32, m)
    counts = {}
    for i in live:
        counts[strings[i]] = counts.get(strings[i], 0) + 1
    assert m == sum(counts.get(s, 0) for jx, s in enumerate(probe) if not pn[jx])
    assert all(strings[b] == probe[a] for a, b in zip(pp[:20000], bb[:20000]))
    j.free(); agg.free(); d.free(); d2.free()
    for c in (col, dv, pc):
        c.free()


@pytest.mark.gpu
def test_having_on_the_device_matches_the_oracle_filter(ctx):
    """ph_agg_fetch_where: the groups whose SUM(DECIMAL) exceeds a DECIMAL / a FLOAT constant (float32 compare, as ph_filter_select) and whose
    COUNT(*) exceeds k — against oracle_select over the same group values (executeSelect's rules: function_operator_boolean.go:393-521)"""
    rng = np.random.default_rng(21)
    n, card = 300_000, 5_000
    keys = rng.integers(0, card, n).astype(np.int32)
    vals = rng.integers(-50_000, 200_000, n).astype(np.int64)          # DECIMAL scale 2
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], card)
    agg.sink([hip.DevColumn(ctx, hip.PH_I32, keys)], [hip.DevColumn(ctx, hip.PH_DEC64, vals, scale=2)], None, n)
    full = agg.finalize()
    sums = np.array([s[0] for s in full["sum"]], np.int64)
    cnts = full["count"][:, 1]
    for k_sum, k_cnt in ((hip.const(hip.PH_DEC64, i=4_500_000, scale=2), 55), (hip.const(hip.PH_F32, f=45000.37), 62), (hip.const(hip.PH_I32, i=47000), 1)):
        # ('>' is the one comparison DECIMAL and HUGEINT both have in selectOperation: '>=' on them selects nothing there — and here)
        got = agg.finalize(where=[(0, hip.PH_GT, k_sum, 2), (1, hip.PH_GT, hip.const(hip.PH_I32, i=k_cnt - 1), 0)])
        assert agg.finalize(where=[(1, hip.PH_GE, hip.const(hip.PH_I32, i=1), 0)])["ngroups"] == 0
        if k_sum.type == hip.PH_F32:
            s1 = O.select(O.col(O.OT_DECIMAL, sums, scale=2), O.OP_GT, O.const(O.OT_FLOAT, f=k_sum.f), n=len(sums))
        elif k_sum.type == hip.PH_DEC64:
            s1 = np.nonzero(sums > k_sum.i)[0]
        else:
            s1 = np.nonzero(sums > k_sum.i * 100)[0]
        keep = [g for g in s1 if cnts[g] >= k_cnt]
        assert 0 < len(keep) < full["ngroups"]
        assert got["ngroups"] == len(keep)
        assert got["keys"][:, 0].tolist() == full["keys"][keep, 0].tolist()                   # first-seen order kept
        assert [s[0] for s in got["sum"]] == [full["sum"][g][0] for g in keep]
    agg.free()


@pytest.mark.gpu
def test_float_eval_follows_the_float_and_double_overloads(ctx):
    """ph_float_eval against numpy's IEEE float32 / float64 arithmetic, operation by operation as mulFloat32 / mulFloat64 and the casts
    tryCastInt32ToFloat32 / tryCastDecimalToFloat32 (decimal -> float64 -> float32) do it; comparisons that selectOperation lacks for the type
    ('<' on FLOAT, '>' on DOUBLE) are never true. Parity unpinned beyond Q17 / Q20, whose goldens run through these two programs."""
    rng = np.random.default_rng(5)
    n = 200_000
    q = rng.integers(1, 51, n).astype(np.int32)
    s = rng.integers(1, 4000, n).astype(np.int64)                     # a HUGEINT sum, carried as a scale-0 decimal
    c = rng.integers(1, 80, n).astype(np.int64)
    d = rng.integers(-10**9, 10**9, n).astype(np.int64)               # DECIMAL scale 2
    Q, S, C, D = (hip.DevColumn(ctx, hip.PH_I32, q), hip.DevColumn(ctx, hip.PH_DEC64, s, scale=0), hip.DevColumn(ctx, hip.PH_DEC64, c, scale=0),
                  hip.DevColumn(ctx, hip.PH_DEC64, d, scale=2))
    f32 = np.float32
    # Q17's: float64(q) < float64(0.2f) * (float64(s) / float64(c))
    got = ctx.download(hip.float_eval(ctx, [Q, S, C], [hip.X_COL(0), hip.X_F32(0.2), hip.X_COL(1), hip.X_COL(2), hip.X_OP(hip.PH_X_DIV), hip.X_MUL,
                                                      hip.X_OP(hip.PH_X_LT)], None, n, wide=True), np.int32, n)
    want = (q.astype(np.float64) < np.float64(f32(0.2)) * (s.astype(np.float64) / c.astype(np.float64)))
    assert np.array_equal(got.astype(bool), want) and 0 < want.sum() < n
    # Q20's: float32(q) > 0.5f * float32(s), and '>=' / '<=' beside it; '<' on FLOAT is never true
    for op, fn in ((hip.PH_X_GT, np.greater), (hip.PH_X_GE, np.greater_equal), (hip.PH_X_LE, np.less_equal), (hip.PH_X_LT, lambda a, b: np.zeros(n, bool))):
        got = ctx.download(hip.float_eval(ctx, [Q, S], [hip.X_COL(0), hip.X_F32(0.5), hip.X_COL(1), hip.X_MUL, hip.X_OP(op)], None, n), np.int32, n)
        assert np.array_equal(got.astype(bool), fn(q.astype(f32), f32(0.5) * s.astype(f32)))
    # values: 100.00f * float32(decimal) / float32(decimal') — Q14's select list — every operation rounded to float32
    got = ctx.download(hip.float_eval(ctx, [D, S], [hip.X_F32(100.0), hip.X_COL(0), hip.X_MUL, hip.X_COL(1), hip.X_OP(hip.PH_X_DIV)], None, n, truth=False), np.float32, n)
    want = (f32(100.0) * (d.astype(np.float64) / 100.0).astype(f32)) / s.astype(f32)
    assert np.array_equal(got, want.astype(f32))
    # '>' on DOUBLE does not exist: never true
    got = ctx.download(hip.float_eval(ctx, [Q, S], [hip.X_COL(0), hip.X_COL(1), hip.X_OP(hip.PH_X_GT)], None, n, wide=True), np.int32, n)
    assert not got.any()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [4, 8])
def test_existence_only_table_marks_like_the_oracle(ctx, kw):
    """PH_JOIN_EXISTS_ONLY: a big build side with duplicates, a key range, NULL keys and a build selection as one flag byte per key value; its
    marks equal oracle_join_probe_mark's (probe selection and NULL probe keys included); every other probe is refused"""
    rng = np.random.default_rng(kw)
    nb, npr = 700_000, 900_000
    dt, ht, ot = (np.int32, hip.PH_I32, O.OT_INT32) if kw == 4 else (np.int64, hip.PH_I64, O.OT_INT64)
    bk = rng.integers(1000, 3_000_000, nb).astype(dt)
    bvalid = np.packbits(rng.random(nb) > 0.01, bitorder="little")
    bsel = np.sort(rng.choice(nb, 400_000, replace=False))
    pk = rng.integers(0, 3_100_000, npr).astype(dt)
    pvalid = np.packbits(rng.random(npr) > 0.02, bitorder="little")
    psel = np.sort(rng.choice(npr, 500_000, replace=False))
    B = hip.DevColumn(ctx, ht, bk, validity=bvalid)
    P = hip.DevColumn(ctx, ht, pk, validity=pvalid)
    j = hip.Join(ctx, [B], ctx.upload(bsel.astype(np.int32)), len(bsel), key_range=(1000, 2_999_999), exists_only=True)
    assert j.kind == "bitmap"
    oj = O.Join([O.col(ot, bk, validity=bvalid)], bsel.astype(np.int64), len(bsel))
    got = dl(ctx, j.probe_mark([P], ctx.upload(psel.astype(np.int32)), len(psel)), np.uint8, len(psel))
    want = oj.probe_mark([O.col(ot, pk, validity=pvalid)], psel.astype(np.int64), len(psel))
    assert np.array_equal(got, want) and 0 < want.sum() < len(psel)
    got = dl(ctx, j.probe_mark([P], None, npr), np.uint8, npr)
    assert np.array_equal(got, oj.probe_mark([O.col(ot, pk, validity=pvalid)], None, npr))
    with pytest.raises(hip.PlanHipError):
        j.probe_inner([P], None, npr, 1 << 20)
    j.free()
    B.free(); P.free()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["PH_STREAM_AGG_ONE_PASS", "PH_STREAM_AGG_TWO_PASS"])
def test_streaming_aggregate_long_runs_are_reduced_tile_by_tile(ctx, form, monkeypatch):
    """(both forms) Round 4: the streaming aggregate reduces every 1024-row tile as a segmented scan and joins the pieces of a run that crosses
    tiles in a fix-up pass, one step per TILE — so the shapes round 3 withdrew from (a table clustered by a low-cardinality key: 8 runs of
    50 000 rows) and the one it still walked row by row (ONE long run among short ones) both give the hash aggregate's groups, record for
    record: SUM, MIN, MAX, COUNT(*) and a NULL-able argument, runs that end exactly at a tile boundary, a last partial tile."""
    monkeypatch.setenv(form, "1")
    rng = np.random.default_rng(9)
    cases = {
        "low cardinality": np.repeat(np.arange(8, dtype=np.int64), 50_000),
        "one long run among short ones": np.concatenate([np.repeat(np.arange(3000, dtype=np.int64), rng.integers(1, 6, 3000)), np.full(300_000, 5000, np.int64),
                                                        np.repeat(np.arange(6000, 9000, dtype=np.int64), rng.integers(1, 6, 3000))]),
        "runs ending at tile boundaries": np.repeat(np.arange(40, dtype=np.int64), 1024),
        "every row a group": np.arange(5000, dtype=np.int64),
    }
    aggs = [(hip.PH_A_SUM, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0), (hip.PH_A_COUNT_STAR, -1), (hip.PH_A_COUNT, 1)]
    for name, keys in cases.items():
        n = len(keys)
        vals = rng.integers(-10**15, 10**15, n).astype(np.int64)
        v2 = rng.integers(0, 100, n).astype(np.int32)
        valid = rng.random(n) > 0.3
        K, V = hip.DevColumn(ctx, hip.PH_I64, keys), hip.DevColumn(ctx, hip.PH_I64, vals)
        W = hip.DevColumn(ctx, hip.PH_I32, v2, validity=np.packbits(valid, bitorder="little"))
        res = []
        for sorted_form in (True, False):
            agg = hip.Agg(ctx, [hip.PH_I64], aggs, 1024)
            if sorted_form:
                assert agg.sink_sorted([K], [V, W], n), name
            else:
                agg.sink([K], [V, W], None, n)
            r = agg.finalize(python_ints=False, room=len(np.unique(keys)))
            res.append({k: np.asarray(r[k]) for k in ("first_row", "keys", "sum_lo", "sum_hi", "count")})
            agg.free()
        ctx.check_deferred()
        for k in res[0]:
            assert np.array_equal(res[0][k], res[1][k]), (name, k)
        uk, first = np.unique(keys, return_index=True)
        assert np.array_equal(res[0]["keys"][:, 0], uk) and np.array_equal(res[0]["first_row"], first), name
        sums = np.add.reduceat(vals.astype(object), first)
        got = [(int(h) << 64) + int(l) for l, h in zip(res[0]["sum_lo"][:, 0], res[0]["sum_hi"][:, 0])]
        assert got == [int(x) for x in sums], name
        for c in (K, V, W):
            c.free()
    # (ordered first key, UNordered second key): equal (k0, k1) tuples are no longer adjacent — the claim is verified on the device
    k0 = np.repeat(np.arange(1000, dtype=np.int64), 4)
    k1 = np.tile(np.array([2, 1, 2, 1], np.int32), 1000)
    K0, K1, V = hip.DevColumn(ctx, hip.PH_I64, k0), hip.DevColumn(ctx, hip.PH_I32, k1), hip.DevColumn(ctx, hip.PH_I64, np.ones(4000, np.int64))
    agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_I32], [(hip.PH_A_SUM, 0)], 1024)
    assert agg.sink_sorted([K0, K1], [V], 4000)
    with pytest.raises(hip.PlanHipError) as e:
        agg.group_count()
    assert e.value.code == hip.PH_ECONSTRAINT and "ph_agg_sink_sorted" in str(e.value)
    agg.free()
    for c in (K0, K1, V):
        c.free()
