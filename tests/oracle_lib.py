"""ctypes binding of oracle/liboracle.so — the CPU restatement used as the parity checker.
Test infrastructure only: nothing under plan_amd/ imports this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

OT_INT32, OT_INT64, OT_DATE, OT_DECIMAL, OT_CODE8, OT_FLOAT, OT_DOUBLE, OT_ODEC, OT_VARCHAR, OT_CONST32 = range(1, 11)
OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_LIKE, OP_NOTLIKE = range(1, 9)
OX_COL, OX_CONST_INT, OX_CONST_DEC, OX_ADD, OX_SUB, OX_MUL = range(1, 7)
OA_SUM, OA_AVG, OA_COUNT, OA_MIN, OA_MAX = range(1, 6)
OV_NULL, OV_HUGEINT, OV_DECIMAL, OV_DOUBLE = range(4)

i64 = ctypes.c_int64
i32 = ctypes.c_int32


class ODec(ctypes.Structure):
    _fields_ = [("neg", ctypes.c_uint8), ("scale", ctypes.c_int8), ("coef", ctypes.c_uint64)]

    def unscaled(self, scale):
        """Exact unscaled integer at `scale` (raises if not exact)."""
        v = int(self.coef)
        if scale >= self.scale:
            v *= 10 ** (scale - self.scale)
        else:
            q, r = divmod(v, 10 ** (self.scale - scale))
            assert r == 0
            v = q
        return -v if self.neg else v

    def __str__(self):
        buf = ctypes.create_string_buffer(64)
        lib().odec_string(self, buf)
        return buf.value.decode()


ODEC_DTYPE = np.dtype([("neg", np.uint8), ("scale", np.int8), ("coef", np.uint64)], align=True)


class OCol(ctypes.Structure):
    _fields_ = [("type", i32), ("scale", i32), ("data", ctypes.c_void_p),
                ("validity", ctypes.c_void_p), ("dict", ctypes.POINTER(ctypes.c_char_p)),
                ("dict_size", i32)]


class OConst(ctypes.Structure):
    _fields_ = [("type", i32), ("scale", i32), ("i", i64), ("f", ctypes.c_double),
                ("s", ctypes.c_char_p)]


class ORpn(ctypes.Structure):
    _fields_ = [("op", i32), ("col", i32), ("ival", i64), ("scale", i32)]


class OAggSpec(ctypes.Structure):
    _fields_ = [("kind", i32), ("arg", i32)]


class OHuge(ctypes.Structure):
    _fields_ = [("lower", ctypes.c_uint64), ("upper", ctypes.c_int64)]

    def value(self):
        return (int(self.upper) << 64) + int(self.lower)


class OAggVal(ctypes.Structure):
    _fields_ = [("kind", i32), ("h", OHuge), ("d", ODec), ("f", ctypes.c_double)]


class Q1Row(ctypes.Structure):
    _fields_ = [("returnflag", ctypes.c_uint8), ("linestatus", ctypes.c_uint8),
                ("sum_qty", OHuge), ("sum_base_price", ODec), ("sum_disc_price", ODec),
                ("sum_charge", ODec), ("avg_qty", ctypes.c_double), ("avg_price", ODec),
                ("avg_disc", ODec), ("count_order", ctypes.c_uint64)]


class Lineitem(ctypes.Structure):
    _fields_ = [("l_quantity", ctypes.c_void_p), ("l_extendedprice", ctypes.c_void_p),
                ("l_discount", ctypes.c_void_p), ("l_tax", ctypes.c_void_p),
                ("l_returnflag", ctypes.c_void_p), ("l_linestatus", ctypes.c_void_p),
                ("l_shipdate", ctypes.c_void_p), ("l_orderkey", ctypes.c_void_p),
                ("l_partkey", ctypes.c_void_p), ("l_suppkey", ctypes.c_void_p),
                ("returnflag_dict", ctypes.POINTER(ctypes.c_char_p)),
                ("linestatus_dict", ctypes.POINTER(ctypes.c_char_p)), ("n", i64)]


class Orders(ctypes.Structure):
    _fields_ = [("o_orderkey", ctypes.c_void_p), ("o_custkey", ctypes.c_void_p),
                ("o_orderdate", ctypes.c_void_p), ("o_shippriority", ctypes.c_void_p), ("n", i64)]


class Customer(ctypes.Structure):
    _fields_ = [("c_custkey", ctypes.c_void_p), ("c_mktsegment", ctypes.c_void_p),
                ("mktsegment_dict", ctypes.POINTER(ctypes.c_char_p)), ("dict_size", i32),
                ("n", i64)]


class Q3Row(ctypes.Structure):
    _fields_ = [("l_orderkey", i64), ("revenue", ODec), ("o_orderdate", i32),
                ("o_shippriority", i32)]


class Part(ctypes.Structure):
    _fields_ = [("p_partkey", ctypes.c_void_p), ("p_name_off", ctypes.c_void_p),
                ("p_name_bytes", ctypes.c_void_p), ("n", i64)]


class PartSupp(ctypes.Structure):
    _fields_ = [("ps_partkey", ctypes.c_void_p), ("ps_suppkey", ctypes.c_void_p),
                ("ps_supplycost", ctypes.c_void_p), ("n", i64)]


class Supplier(ctypes.Structure):
    _fields_ = [("s_suppkey", ctypes.c_void_p), ("s_nationkey", ctypes.c_void_p), ("n", i64)]


class Q9Row(ctypes.Structure):
    _fields_ = [("nationkey", i32), ("o_year", i32), ("sum_profit", ODec)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(path)
        for f in ("oracle_select", "oracle_groupby", "oracle_join_count", "oracle_join_probe_inner",
                  "oracle_q3", "oracle_q9", "oracle_q1_text", "oracle_q6_text", "oracle_q3_text",
                  "oracle_q9_text", "oracle_agg_count", "oracle_select_or"):
            getattr(L, f).restype = i64
        L.oracle_join_build.restype = ctypes.c_void_p
        L.oracle_agg_create.restype = ctypes.c_void_p
        L.odec_float64.restype = ctypes.c_double
        _LIB = L
    return _LIB


def cdict(strings):
    arr = (ctypes.c_char_p * len(strings))(*[s.encode() for s in strings])
    return arr


def ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def col(typ, data, scale=0, validity=None, dictionary=None):
    c = OCol()
    c.type = typ
    c.scale = scale
    c.data = None if data is None else data.ctypes.data
    c.validity = None if validity is None else validity.ctypes.data
    if dictionary is not None:
        if typ == OT_VARCHAR:  # dictionary = bytes array
            c.dict = ctypes.cast(ctypes.c_void_p(dictionary.ctypes.data),
                                 ctypes.POINTER(ctypes.c_char_p))
        else:
            c.dict = ctypes.cast(dictionary, ctypes.POINTER(ctypes.c_char_p))
            c.dict_size = len(dictionary)
    c._keep = (data, validity, dictionary)
    return c


def const(typ, i=0, f=0.0, s=None, scale=0):
    k = OConst()
    k.type, k.scale, k.i, k.f = typ, scale, int(i), float(f)
    k.s = None if s is None else s.encode()
    return k


def select(c, op, k, sel_in=None, n=None):
    if sel_in is not None:
        n = len(sel_in)
    out = np.empty(max(n, 1), dtype=np.int64)
    m = lib().oracle_select(ctypes.byref(c), i32(op), ctypes.byref(k), ptr(sel_in), i64(n), ptr(out))
    return out[:m].copy()


def hash_cols(cols, n):
    arr = (OCol * len(cols))(*cols)
    out = np.empty(n, dtype=np.uint64)
    lib().oracle_hash(arr, i32(len(cols)), i64(n), ptr(out))
    return out


def eval_decimal(cols, prog, sel, n):
    arr = (OCol * len(cols))(*cols)
    p = (ORpn * len(prog))(*[ORpn(*x) for x in prog])
    out = np.zeros(n, dtype=ODEC_DTYPE)
    rc = lib().oracle_eval_decimal(arr, p, i32(len(prog)), ptr(sel), i64(n), ptr(out))
    return rc, out


def case_decimal(cols, when_col, when_op, when_k, then_prog, else_prog, n):
    """CASE WHEN when_col OP when_k THEN then_prog ELSE else_prog END -> (rc, odec array, null flags)"""
    arr = (OCol * len(cols))(*cols)
    tp = (ORpn * len(then_prog))(*[ORpn(*x) for x in then_prog])
    ep = (ORpn * len(else_prog))(*[ORpn(*x) for x in else_prog])
    out = np.zeros(n, dtype=ODEC_DTYPE)
    nul = np.zeros(n, dtype=np.uint8)
    rc = lib().oracle_case_decimal(arr, ctypes.byref(when_col), i32(when_op), ctypes.byref(when_k), tp, i32(len(then_prog)),
                                   ep, i32(len(else_prog)), i64(n), ptr(out), ptr(nul))
    return rc, out, nul


def odec_unscaled(arr, scale):
    """numpy ODEC array -> python ints at `scale` (exact)."""
    out = []
    for neg, sc, coef in zip(arr["neg"], arr["scale"], arr["coef"]):
        v = int(coef)
        if scale >= sc:
            v *= 10 ** (scale - int(sc))
        else:
            q, r = divmod(v, 10 ** (int(sc) - scale))
            assert r == 0
            v = q
        out.append(-v if neg else v)
    return out


def select_or(children, sel_in=None, n=None):
    """children: list of (ocol, op, oconst). Returns the selected rows in the reference's order."""
    k = len(children)
    cols = (OCol * k)(*[c for c, _, _ in children])
    ops = (i32 * k)(*[o for _, o, _ in children])
    ks = (OConst * k)(*[kk for _, _, kk in children])
    n_in = len(sel_in) if sel_in is not None else n
    out = np.empty(max(n_in, 1), dtype=np.int64)
    m = lib().oracle_select_or(cols, ops, ks, i32(k), ptr(sel_in), i64(n_in), ptr(out))
    return out[:m].copy()


def sort_rows(cols, descending, sel=None, n=None):
    """ORDER BY restatement: (sorted row ids, sorted key bytes [n, width])"""
    k = len(cols)
    arr = (OCol * k)(*cols)
    desc = (i32 * k)(*[1 if d else 0 for d in descending])
    n_in = len(sel) if sel is not None else n
    rows = np.empty(max(n_in, 1), dtype=np.int64)
    width = i32()
    keys = np.zeros(max(n_in, 1) * 80, dtype=np.uint8)
    rc = lib().oracle_sort_rows(arr, desc, i32(k), ptr(sel), i64(n_in), ptr(rows), ctypes.byref(width), ptr(keys))
    assert rc == 0, rc
    return rows[:n_in].copy(), keys[:n_in * width.value].reshape(n_in, width.value).copy()


def groupby(keys, args, aggs, sel, n, max_groups):
    ka = (OCol * len(keys))(*keys)
    aa = (OCol * max(len(args), 1))(*args)
    sp = (OAggSpec * len(aggs))(*[OAggSpec(k, a) for k, a in aggs])
    first = np.zeros(max_groups, dtype=np.int64)
    gk = np.zeros(max_groups * len(keys), dtype=np.int64)
    gn = np.zeros(max_groups * len(keys), dtype=np.uint8)
    vals = (OAggVal * (max_groups * len(aggs)))()
    ng = lib().oracle_groupby(ka, i32(len(keys)), aa, i32(len(args)), sp, i32(len(aggs)),
                              ptr(sel), i64(n), ptr(first), ptr(gk), ptr(gn), vals, i64(max_groups))
    return ng, first, gk.reshape(max_groups, len(keys)), gn.reshape(max_groups, len(keys)), vals


class Agg:
    """Incremental GroupedAggrHashTable of the oracle (create -> sink(<= 2048 rows, filter)* ->
    groups), i.e. what aggExecutor and the DISTINCT finalisation drive chunk by chunk.
    Columns are given as (type, numpy array, scale, bool validity array or None)."""

    def __init__(self, key_specs, arg_specs, aggs):
        self.key_specs, self.arg_specs, self.aggs = key_specs, arg_specs, aggs
        kp = (OCol * len(key_specs))(*[col(t, None, sc) for t, sc in key_specs])
        ap = (OCol * max(len(arg_specs), 1))(*[col(t, None, sc) for t, sc in arg_specs])
        sp = (OAggSpec * max(len(aggs), 1))(*[OAggSpec(k, a) for k, a in aggs])
        self.h = ctypes.c_void_p(lib().oracle_agg_create(kp, i32(len(key_specs)), ap, sp, i32(len(aggs))))

    def sink(self, keys, args, mask=0xFFFFFFFF, row_base=0):
        """keys/args: lists of (numpy array, bool validity or None) for the same n rows"""
        n = len(keys[0][0])
        for base in range(0, n, 2048):
            cnt = min(2048, n - base)

            def mk(specs, cols):
                out = []
                for (t, sc), (arr, valid) in zip(specs, cols):
                    a = np.ascontiguousarray(arr[base:base + cnt])
                    v = None if valid is None else np.packbits(valid[base:base + cnt], bitorder="little")
                    out.append(col(t, a, sc, validity=v))
                return out
            kc, ac = mk(self.key_specs, keys), mk(self.arg_specs, args)
            rid = np.arange(row_base + base, row_base + base + cnt, dtype=np.int64)
            rc = lib().oracle_agg_sink_filtered(self.h, (OCol * len(kc))(*kc), (OCol * max(len(ac), 1))(*ac),
                                                ptr(rid), i64(cnt), ctypes.c_uint32(mask))
            assert rc == 0, rc

    def groups(self):
        ng = lib().oracle_agg_count(self.h)
        nk, na = len(self.key_specs), len(self.aggs)
        out = []
        for g in range(ng):
            first = i64()
            kv = (i64 * nk)()
            kn = (ctypes.c_uint8 * nk)()
            vals = (OAggVal * max(na, 1))()
            assert lib().oracle_agg_group(self.h, i64(g), ctypes.byref(first), kv, kn, vals) == 0
            out.append((first.value, [None if kn[c] else int(kv[c]) for c in range(nk)], [vals[a] for a in range(na)]))
        return out

    def __del__(self):
        if self.h:
            lib().oracle_agg_free(self.h)
            self.h = None


class Join:
    def __init__(self, keys, sel, n):
        self.keys = keys
        arr = (OCol * len(keys))(*keys)
        self.h = ctypes.c_void_p(lib().oracle_join_build(arr, i32(len(keys)), ptr(sel), i64(n)))
        self.nkeys = len(keys)

    def count(self):
        return lib().oracle_join_count(self.h)

    def probe_inner(self, keys, sel, n, cap):
        arr = (OCol * len(keys))(*keys)
        op = np.empty(max(cap, 1), dtype=np.int64)
        ob = np.empty(max(cap, 1), dtype=np.int64)
        m = lib().oracle_join_probe_inner(self.h, arr, i32(len(keys)), ptr(sel), i64(n), ptr(op),
                                          ptr(ob), i64(cap))
        return m, op[:min(m, cap)].copy(), ob[:min(m, cap)].copy()

    def probe_mark(self, keys, sel, n):
        arr = (OCol * len(keys))(*keys)
        found = np.zeros(max(n, 1), dtype=np.uint8)
        lib().oracle_join_probe_mark(self.h, arr, i32(len(keys)), ptr(sel), i64(n), ptr(found))
        return found[:n]

    def __del__(self):
        if self.h:
            lib().oracle_join_free(self.h)
            self.h = None


# ---- query drivers ----
RF = ["A", "N", "R"]
LS = ["F", "O"]
SEG = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]


def _lineitem(t):
    L = Lineitem()
    keep = [cdict(RF), cdict(LS)]
    for name in ("l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag",
                 "l_linestatus", "l_shipdate", "l_orderkey", "l_partkey", "l_suppkey"):
        if name in t:
            setattr(L, name, t[name].ctypes.data)
    L.returnflag_dict = ctypes.cast(keep[0], ctypes.POINTER(ctypes.c_char_p))
    L.linestatus_dict = ctypes.cast(keep[1], ctypes.POINTER(ctypes.c_char_p))
    L.n = len(t["l_shipdate"])
    L._keep = (keep, t)
    return L


def q1(lineitem, shipdate_le):
    L = _lineitem(lineitem)
    rows = (Q1Row * 16)()
    n = lib().oracle_q1(ctypes.byref(L), i32(shipdate_le), rows, i32(16))
    return [rows[i] for i in range(n)]


def q1_text(rows):
    arr = (Q1Row * max(len(rows), 1))(*rows)
    buf = ctypes.create_string_buffer(1 << 16)
    lib().oracle_q1_text(arr, i32(len(rows)), cdict(RF), cdict(LS), buf, i64(len(buf)))
    return buf.value.decode()


def q6(lineitem, date_ge, date_lt, lo, hi, qty_lt):
    L = _lineitem(lineitem)
    d = ODec()
    rc = lib().oracle_q6(ctypes.byref(L), i32(date_ge), i32(date_lt), ctypes.c_float(lo),
                         ctypes.c_float(hi), i32(qty_lt), ctypes.byref(d))
    return rc, d


def q6_text(rc, d):
    buf = ctypes.create_string_buffer(256)
    lib().oracle_q6_text(ctypes.byref(d), ctypes.c_int(rc), buf, i64(len(buf)))
    return buf.value.decode()


def q3(t, segment, date, cap=1 << 22):
    L = _lineitem(t["lineitem"])
    O = Orders()
    for name in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"):
        setattr(O, name, t["orders"][name].ctypes.data)
    O.n = len(t["orders"]["o_orderkey"])
    C = Customer()
    C.c_custkey = t["customer"]["c_custkey"].ctypes.data
    C.c_mktsegment = t["customer"]["c_mktsegment"].ctypes.data
    seg = cdict(SEG)
    C.mktsegment_dict = ctypes.cast(seg, ctypes.POINTER(ctypes.c_char_p))
    C.dict_size = 5
    C.n = len(t["customer"]["c_custkey"])
    rows = (Q3Row * cap)()
    n = lib().oracle_q3(ctypes.byref(L), ctypes.byref(O), ctypes.byref(C), segment.encode(),
                        i32(date), rows, i64(cap))
    return n, rows


def q3_text(rows, n, limit=10):
    buf = ctypes.create_string_buffer(1 << 16)
    lib().oracle_q3_text(rows, i64(n), i32(limit), buf, i64(len(buf)))
    return buf.value.decode()


def q9(t, pattern, cap=4096):
    L = _lineitem(t["lineitem"])
    O = Orders()
    for name in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"):
        setattr(O, name, t["orders"][name].ctypes.data)
    O.n = len(t["orders"]["o_orderkey"])
    P = Part()
    P.p_partkey = t["part"]["p_partkey"].ctypes.data
    P.p_name_off = t["part"]["p_name_off"].ctypes.data
    P.p_name_bytes = t["part"]["p_name_bytes"].ctypes.data
    P.n = len(t["part"]["p_partkey"])
    PS = PartSupp()
    for name in ("ps_partkey", "ps_suppkey", "ps_supplycost"):
        setattr(PS, name, t["partsupp"][name].ctypes.data)
    PS.n = len(t["partsupp"]["ps_partkey"])
    S = Supplier()
    S.s_suppkey = t["supplier"]["s_suppkey"].ctypes.data
    S.s_nationkey = t["supplier"]["s_nationkey"].ctypes.data
    S.n = len(t["supplier"]["s_suppkey"])
    rows = (Q9Row * cap)()
    n = lib().oracle_q9(ctypes.byref(L), ctypes.byref(O), ctypes.byref(P), ctypes.byref(PS),
                        ctypes.byref(S), pattern.encode(), rows, i64(cap))
    return n, rows


def q9_text(rows, n, nation_names):
    buf = ctypes.create_string_buffer(1 << 16)
    lib().oracle_q9_text(rows, i64(n), cdict(nation_names), buf, i64(len(buf)))
    return buf.value.decode()


def substring(b, offset, length):
    """oracle_substring over python bytes"""
    out = ctypes.create_string_buffer(max(len(b), 1))
    lib().oracle_substring.restype = i64
    n = lib().oracle_substring(ctypes.c_char_p(b), i64(len(b)), i64(offset), i64(length), out)
    return out.raw[:n]


def cross_pairs(n_left, n_right, chunk=2048):
    ol = np.empty(max(n_left * n_right, 1), np.int64)
    orr = np.empty(max(n_left * n_right, 1), np.int64)
    lib().oracle_cross_pairs(i64(n_left), i64(n_right), i64(chunk), ptr(ol), ptr(orr))
    return ol[:n_left * n_right], orr[:n_left * n_right]


# ---------------------------------------------------------------- round 3: Q4, Q5, Q12, Q14, Q19

_VP, _DP = ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p)


class Tpch(ctypes.Structure):
    _fields_ = ([("n_lineitem", i64)] + [(c, _VP) for c in ("l_orderkey", "l_extendedprice", "l_discount", "l_partkey", "l_suppkey", "l_quantity",
                                                           "l_shipdate", "l_commitdate", "l_receiptdate", "l_shipmode", "l_shipinstruct")] +
                [("n_orders", i64)] + [(c, _VP) for c in ("o_orderkey", "o_custkey", "o_orderdate", "o_orderpriority")] +
                [("n_customer", i64), ("c_custkey", _VP), ("c_nationkey", _VP), ("n_supplier", i64), ("s_suppkey", _VP), ("s_nationkey", _VP),
                 ("n_part", i64), ("p_partkey", _VP), ("p_size", _VP), ("p_brand", _VP), ("p_type", _VP), ("p_container", _VP),
                 ("n_nationkey", _VP), ("n_regionkey", _VP), ("r_regionkey", _VP), ("n_name", _VP), ("r_name", _VP)] +
                [(c, _DP) for c in ("shipmode_dict", "shipinstruct_dict", "orderpriority_dict", "brand_dict", "type_dict", "container_dict",
                                    "nation_dict", "region_dict")])


class Q4Row(ctypes.Structure):
    _fields_ = [("code", i32), ("count", OHuge)]


class Q5Row(ctypes.Structure):
    _fields_ = [("nation", i32), ("revenue", ODec)]


class Q12Row(ctypes.Structure):
    _fields_ = [("mode", i32), ("high", OHuge), ("low", OHuge)]


class Q7Row(ctypes.Structure):
    _fields_ = [("supp_nation", i32), ("cust_nation", i32), ("l_year", i32), ("revenue", ODec)]


class Q8Row(ctypes.Structure):
    _fields_ = [("o_year", i32), ("nation_volume", ODec), ("volume", ODec), ("mkt_share", ODec)]


def tpch_struct(t):
    """oracle_tpch over the numpy tables of tpch_data.load (+ the fixed nation / region tables); returns (struct, keepalive)"""
    from plan_amd import tpchgen
    T, keep = Tpch(), []

    def put(name, arr):
        arr = np.ascontiguousarray(arr)
        keep.append(arr)
        setattr(T, name, arr.ctypes.data)

    L, Od, C = t["lineitem"], t["orders"], t["customer"]
    T.n_lineitem, T.n_orders, T.n_customer = len(L["l_orderkey"]), len(Od["o_orderkey"]), len(C["c_custkey"])
    for c in ("l_orderkey", "l_extendedprice", "l_discount", "l_partkey", "l_suppkey", "l_quantity", "l_shipdate", "l_commitdate",
              "l_receiptdate", "l_shipmode", "l_shipinstruct"):
        put(c, L[c])
    for c in ("o_orderkey", "o_custkey", "o_orderdate", "o_orderpriority"):
        put(c, Od[c])
    put("c_custkey", C["c_custkey"]); put("c_nationkey", C["c_nationkey"])
    if "supplier" in t:
        T.n_supplier = len(t["supplier"]["s_suppkey"])
        put("s_suppkey", t["supplier"]["s_suppkey"]); put("s_nationkey", t["supplier"]["s_nationkey"])
    if "part" in t:
        P = t["part"]
        T.n_part = len(P["p_partkey"])
        for c in ("p_partkey", "p_size", "p_brand", "p_type", "p_container"):
            put(c, P[c])
    put("n_nationkey", np.arange(25, dtype=np.int32)); put("n_regionkey", np.array(tpchgen.nation_regions(), dtype=np.int32))
    put("r_regionkey", np.arange(5, dtype=np.int32)); put("n_name", np.arange(25, dtype=np.uint8)); put("r_name", np.arange(5, dtype=np.uint8))
    for field, strings in (("shipmode_dict", tpchgen.SHIPMODE_DICT), ("shipinstruct_dict", tpchgen.SHIPINSTRUCT_DICT),
                           ("orderpriority_dict", tpchgen.ORDERPRIORITY_DICT), ("brand_dict", tpchgen.part_brand_dict()),
                           ("type_dict", tpchgen.part_type_dict()), ("container_dict", tpchgen.part_container_dict()),
                           ("nation_dict", tpchgen.nation_names()), ("region_dict", tpchgen.region_names())):
        d = cdict(strings)
        keep.append(d)
        setattr(T, field, ctypes.cast(d, ctypes.POINTER(ctypes.c_char_p)))
    return T, keep


def _text(fn, *args, cap=1 << 16):
    buf = ctypes.create_string_buffer(cap)
    getattr(lib(), fn)(*args, buf, i64(cap))
    return buf.value.decode()


def q4_text(t, date_ge, date_lt):
    from plan_amd import tpchgen
    T, keep = tpch_struct(t)
    rows = (Q4Row * 16)()
    n = lib().oracle_q4(ctypes.byref(T), i32(date_ge), i32(date_lt), rows, i64(16))
    assert n >= 0
    return _text("oracle_q4_text", rows, i64(n), cdict(tpchgen.ORDERPRIORITY_DICT))


def q5_text(t, region, date_ge, date_lt):
    from plan_amd import tpchgen
    T, keep = tpch_struct(t)
    rows = (Q5Row * 32)()
    n = lib().oracle_q5(ctypes.byref(T), region.encode(), i32(date_ge), i32(date_lt), rows, i64(32))
    assert n >= 0
    return _text("oracle_q5_text", rows, i64(n), cdict(tpchgen.nation_names()))


def q7_rows(t, nation_a, nation_b, date_ge, date_le):
    T, keep = tpch_struct(t)
    rows = (Q7Row * 64)()
    n = lib().oracle_q7(ctypes.byref(T), nation_a.encode(), nation_b.encode(), i32(date_ge), i32(date_le), rows, i64(64))
    assert n >= 0
    return rows, n


def q7_text(t, nation_a, nation_b, date_ge, date_le):
    from plan_amd import tpchgen
    rows, n = q7_rows(t, nation_a, nation_b, date_ge, date_le)
    return _text("oracle_q7_text", rows, i64(n), cdict(tpchgen.nation_names()))


def q8_rows(t, nation, region, ptype, date_ge, date_le):
    T, keep = tpch_struct(t)
    rows = (Q8Row * 16)()
    n = lib().oracle_q8(ctypes.byref(T), nation.encode(), region.encode(), ptype.encode(), i32(date_ge), i32(date_le), rows, i64(16))
    assert n >= 0
    return rows, n


def q8_text(t, nation, region, ptype, date_ge, date_le):
    rows, n = q8_rows(t, nation, region, ptype, date_ge, date_le)
    return _text("oracle_q8_text", rows, i64(n))


class Q11Row(ctypes.Structure):
    _fields_ = [("ps_partkey", i32), ("value", ODec)]


def q11_rows(t, nation="JAPAN", fraction=0.0001):
    T, keep = tpch_struct(t)
    PS = t["partsupp"]
    arrs = [np.ascontiguousarray(PS[c]) for c in ("ps_partkey", "ps_suppkey", "ps_supplycost", "ps_availqty")]
    n_ps = len(arrs[0])
    rows = (Q11Row * (n_ps // 4 + 1))()
    lib().oracle_q11.restype = i64
    n = lib().oracle_q11(ctypes.byref(T), i64(n_ps), *[ctypes.c_void_p(a.ctypes.data) for a in arrs], nation.encode(), ctypes.c_float(fraction),
                         rows, i64(len(rows)))
    assert n >= 0
    return rows, n


def q11_text(t, nation="JAPAN", fraction=0.0001):
    rows, n = q11_rows(t, nation, fraction)
    return _text("oracle_q11_text", rows, i64(n), cap=1 << 20)


def q12_text(t, mode1, mode2, date_ge, date_lt):
    from plan_amd import tpchgen
    T, keep = tpch_struct(t)
    rows = (Q12Row * 16)()
    n = lib().oracle_q12(ctypes.byref(T), mode1.encode(), mode2.encode(), i32(date_ge), i32(date_lt), rows, i64(16))
    assert n >= 0
    return _text("oracle_q12_text", rows, i64(n), cdict(tpchgen.SHIPMODE_DICT))


def q14(t, pattern, date_ge, date_lt):
    """(rc, float32 promo_revenue, promo sum, total sum)"""
    T, keep = tpch_struct(t)
    f, a, b = ctypes.c_float(), ODec(), ODec()
    lib().oracle_q14.restype = i32
    rc = lib().oracle_q14(ctypes.byref(T), pattern.encode(), i32(date_ge), i32(date_lt), ctypes.byref(f), ctypes.byref(a), ctypes.byref(b))
    return rc, f.value, a, b


def q14_text(t, pattern, date_ge, date_lt):
    rc, f, _a, _b = q14(t, pattern, date_ge, date_lt)
    assert rc >= 0
    return _text("oracle_q14_text", ctypes.c_float(f), i32(rc))


def q20_keys(t, pattern="lime%", nation="VIETNAM", date_ge=None, date_lt=None, fraction=0.5):
    from plan_amd import tpchgen
    date_ge = tpchgen.days(1993, 1, 1) if date_ge is None else date_ge
    date_lt = tpchgen.days(1994, 1, 1) if date_lt is None else date_lt
    T, keep = tpch_struct(t)
    P, PS = t["part"], t["partsupp"]
    arrs = [np.ascontiguousarray(P["p_name_off"]), np.ascontiguousarray(P["p_name_bytes"])] + [np.ascontiguousarray(PS[c]) for c in ("ps_partkey", "ps_suppkey", "ps_availqty")]
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    out = np.zeros(len(t["supplier"]["s_suppkey"]), np.int32)
    lib().oracle_q20.restype = i64
    n = lib().oracle_q20(ctypes.byref(T), vp(arrs[0]), vp(arrs[1]), i64(len(arrs[2])), vp(arrs[2]), vp(arrs[3]), vp(arrs[4]), pattern.encode(), nation.encode(),
                         i32(date_ge), i32(date_lt), ctypes.c_float(fraction), vp(out), i64(len(out)))
    assert n >= 0
    return out[:n]


def q20_text(t, **kw):
    keys = np.ascontiguousarray(q20_keys(t, **kw))
    S = t["supplier"]
    key, off, ab = np.ascontiguousarray(S["s_suppkey"]), np.ascontiguousarray(S["s_address_off"]), np.ascontiguousarray(S["s_address_bytes"])
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    return _text("oracle_q20_text", vp(keys), i64(len(keys)), vp(key), i64(len(key)), vp(off), vp(ab), cap=1 << 18)


class Q21Row(ctypes.Structure):
    _fields_ = [("s_suppkey", i32), ("numwait", OHuge)]


def q21_rows(t, nation="BRAZIL"):
    T, keep = tpch_struct(t)
    st = np.frombuffer(b"FOP", np.uint8)[np.ascontiguousarray(t["orders"]["o_orderstatus"])]   # codes -> the raw bytes the restatement compares
    st = np.ascontiguousarray(st)
    rows = (Q21Row * len(t["supplier"]["s_suppkey"]))()
    lib().oracle_q21.restype = i64
    n = lib().oracle_q21(ctypes.byref(T), ctypes.c_void_p(st.ctypes.data), nation.encode(), rows, i64(len(rows)))
    assert n >= 0
    return rows, n


def q21_text(t, nation="BRAZIL", limit=100):
    rows, n = q21_rows(t, nation)
    return _text("oracle_q21_text", rows, i64(n), i32(limit), cap=1 << 18)


Q22_CODES = ("10", "11", "26", "22", "19", "20", "27")


class Q22Row(ctypes.Structure):
    _fields_ = [("cntrycode", ctypes.c_char * 4), ("numcust", OHuge), ("totacctbal", ODec)]


def q22_rows(t, codes=Q22_CODES):
    T, keep = tpch_struct(t)
    C = t["customer"]
    ph, bal = np.ascontiguousarray(C["c_phone_bytes"]), np.ascontiguousarray(C["c_acctbal"])
    arr = (ctypes.c_char_p * len(codes))(*[c.encode() for c in codes])
    rows = (Q22Row * 32)()
    lib().oracle_q22.restype = i64
    n = lib().oracle_q22(ctypes.byref(T), ctypes.c_void_p(ph.ctypes.data), ctypes.c_void_p(bal.ctypes.data), arr, i32(len(codes)), rows, i64(32))
    assert 0 <= n <= 32
    return rows, n


def q22_text(t, codes=Q22_CODES):
    rows, n = q22_rows(t, codes)
    return _text("oracle_q22_text", rows, i64(n))


class Q15Row(ctypes.Structure):
    _fields_ = [("s_suppkey", i32), ("total_revenue", ODec)]


def q15_rows(t, date_ge, date_lt):
    T, keep = tpch_struct(t)
    rows = (Q15Row * 64)()
    lib().oracle_q15.restype = i64
    n = lib().oracle_q15(ctypes.byref(T), i32(date_ge), i32(date_lt), rows, i64(64))
    assert 0 <= n <= 64
    return rows, n


def q15_text(t, date_ge, date_lt):
    rows, n = q15_rows(t, date_ge, date_lt)
    S = t["supplier"]
    key, off = np.ascontiguousarray(S["s_suppkey"]), np.ascontiguousarray(S["s_address_off"])
    ab, pb = np.ascontiguousarray(S["s_address_bytes"]), np.ascontiguousarray(S["s_phone_bytes"])
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    return _text("oracle_q15_text", rows, i64(n), vp(key), i64(len(key)), vp(off), vp(ab), vp(pb))


def q17(t, brand="Brand#54", container="LG BAG", fraction=0.2, divisor=7.0):
    """(rc, float32 avg_yearly, decimal sum)"""
    T, keep = tpch_struct(t)
    f, s = ctypes.c_float(), ODec()
    lib().oracle_q17.restype = i32
    rc = lib().oracle_q17(ctypes.byref(T), brand.encode(), container.encode(), ctypes.c_float(fraction), ctypes.c_float(divisor), ctypes.byref(f), ctypes.byref(s))
    return rc, f.value, s


def q17_text(t, **kw):
    rc, f, _s = q17(t, **kw)
    assert rc >= 0
    return _text("oracle_q17_text", ctypes.c_float(f), i32(rc))


def q19(t):
    T, keep = tpch_struct(t)
    d = ODec()
    lib().oracle_q19.restype = i32
    rc = lib().oracle_q19(ctypes.byref(T), ctypes.byref(d))
    return rc, d


def q19_text(t):
    rc, d = q19(t)
    assert rc >= 0
    return _text("oracle_q19_text", ctypes.byref(d), i32(rc))


class Q18Row(ctypes.Structure):
    _fields_ = [("c_custkey", i32), ("o_orderkey", i64), ("o_orderdate", i32), ("o_totalprice", i64), ("sum_qty", OHuge)]


def q18(t, qty_gt=314):
    T, keep = tpch_struct(t)
    tp = np.ascontiguousarray(t["orders"]["o_totalprice"])
    rows = (Q18Row * 4096)()
    lib().oracle_q18.restype = i64
    n = lib().oracle_q18(ctypes.byref(T), ctypes.c_void_p(tp.ctypes.data), i64(qty_gt), rows, i64(4096))
    assert n >= 0
    return n, rows


def q18_text(t, qty_gt=314, limit=100):
    n, rows = q18(t, qty_gt)
    return _text("oracle_q18_text", rows, i64(n), i32(limit))


# ---------------------------------------------------------------- round 4: Q2, Q10, Q13, Q16 (oracle/oqueries3.c)

class Q16Row(ctypes.Structure):
    _fields_ = [("brand", i32), ("type", i32), ("size", i32), ("supplier_cnt", OHuge), ("cnt_null", ctypes.c_uint8)]


class Q13Row(ctypes.Structure):
    _fields_ = [("c_count", i64), ("c_count_null", ctypes.c_uint8), ("custdist", OHuge)]


class Q2Row(ctypes.Structure):
    _fields_ = [("s_row", i32), ("nation", i32), ("p_row", i32)]


class Q10Row(ctypes.Structure):
    _fields_ = [("c_custkey", i32), ("nation_code", i32), ("revenue", ODec)]


def _vp(a):
    return ctypes.c_void_p(np.ascontiguousarray(a).ctypes.data)


def q16_rows(t, brand_ne="Brand#35", type_notlike="ECONOMY BURNISHED%", sizes=(14, 7, 21, 24, 35, 33, 2, 20), comment_like="%Customer%Complaints%"):
    T, keep = tpch_struct(t)
    PS, S = t["partsupp"], t["supplier"]
    arrs = [np.ascontiguousarray(PS["ps_partkey"]), np.ascontiguousarray(PS["ps_suppkey"]), np.ascontiguousarray(S["s_comment_off"]),
            np.ascontiguousarray(S["s_comment_bytes"])]
    sz = np.array(sizes, np.int32)
    n_groups_max = 150 * 25 * len(sizes) + 16
    rows = (Q16Row * n_groups_max)()
    lib().oracle_q16.restype = i64
    n = lib().oracle_q16(ctypes.byref(T), i64(len(arrs[0])), _vp(arrs[0]), _vp(arrs[1]), _vp(arrs[2]), _vp(arrs[3]), brand_ne.encode(), type_notlike.encode(),
                         _vp(sz), i32(len(sz)), comment_like.encode(), rows, i64(n_groups_max))
    assert 0 <= n <= n_groups_max
    return rows, n


def q16_text(t, **kw):
    from plan_amd import tpchgen
    rows, n = q16_rows(t, **kw)
    return _text("oracle_q16_text", rows, i64(n), cdict(tpchgen.part_brand_dict()), cdict(tpchgen.part_type_dict()), cap=1 << 21)


def q13_rows(t, notlike="%pending%accounts%"):
    T, keep = tpch_struct(t)
    Od = t["orders"]
    off, b = np.ascontiguousarray(Od["o_comment_off"]), np.ascontiguousarray(Od["o_comment_bytes"])
    rows = (Q13Row * 256)()
    lib().oracle_q13.restype = i64
    n = lib().oracle_q13(ctypes.byref(T), _vp(off), _vp(b), notlike.encode(), rows, i64(256))
    assert 0 <= n <= 256
    return rows, n


def q13_text(t, **kw):
    rows, n = q13_rows(t, **kw)
    return _text("oracle_q13_text", rows, i64(n))


def q2_text(t, size=48, type_like="%TIN", region="MIDDLE EAST", limit=100):
    T, keep = tpch_struct(t)
    PS, S, P = t["partsupp"], t["supplier"], t["part"]
    a = [np.ascontiguousarray(PS[c]) for c in ("ps_partkey", "ps_suppkey", "ps_supplycost")]
    cap = len(a[0])
    rows = (Q2Row * cap)()
    lib().oracle_q2.restype = i64
    n = lib().oracle_q2(ctypes.byref(T), i64(cap), _vp(a[0]), _vp(a[1]), _vp(a[2]), i32(size), type_like.encode(), region.encode(), rows, i64(cap))
    assert 0 <= n <= cap
    s = [np.ascontiguousarray(S[c]) for c in ("s_acctbal", "s_address_off", "s_address_bytes", "s_phone_bytes", "s_comment_off", "s_comment_bytes")]
    mf = np.ascontiguousarray(P["p_mfgr"])
    return _text("oracle_q2_text", rows, i64(n), i32(limit), ctypes.byref(T), _vp(s[0]), _vp(mf), _vp(s[1]), _vp(s[2]), _vp(s[3]), _vp(s[4]), _vp(s[5]), cap=1 << 18)


def q10_rows(t, flag="R", date_ge=None, date_lt=None):
    from plan_amd import tpchgen
    date_ge = tpchgen.days(1993, 3, 1) if date_ge is None else date_ge
    date_lt = tpchgen.days(1993, 6, 1) if date_lt is None else date_lt
    T, keep = tpch_struct(t)
    rf = np.ascontiguousarray(t["lineitem"]["l_returnflag"])
    bal = np.ascontiguousarray(t["customer"]["c_acctbal"])
    cap = len(bal)
    rows = (Q10Row * cap)()
    lib().oracle_q10.restype = i64
    n = lib().oracle_q10(ctypes.byref(T), _vp(rf), cdict(tpchgen.RETURNFLAG_DICT), _vp(bal), flag.encode(), i32(date_ge), i32(date_lt), rows, i64(cap))
    assert 0 <= n <= cap
    return rows, n, T, keep


def q10_text(t, limit=20, **kw):
    rows, n, T, keep = q10_rows(t, **kw)
    C = t["customer"]
    c = [np.ascontiguousarray(C[k]) for k in ("c_acctbal", "c_address_off", "c_address_bytes", "c_phone_bytes", "c_comment_off", "c_comment_bytes")]
    return _text("oracle_q10_text", rows, i64(n), i32(limit), ctypes.byref(T), _vp(c[0]), _vp(c[1]), _vp(c[2]), _vp(c[3]), _vp(c[4]), _vp(c[5]), cap=1 << 16)
