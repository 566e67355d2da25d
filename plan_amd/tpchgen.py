"""ctypes front end of the clean-room TPC-H generator (include/tpchgen.h).

Synthetic-data source for bench.py and the tests. Columns come back as numpy arrays in the
device encodings of SURVEY.md §8(d): INTEGER int32, BIGINT int64, DECIMAL(15,2) int64 unscaled,
DATE int32 days since 1970-01-01, VARCHAR(1)/c_mktsegment uint8 dictionary codes.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RETURNFLAG_DICT = ["A", "N", "R"]
LINESTATUS_DICT = ["F", "O"]
MKTSEGMENT_DICT = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]

_i64 = ctypes.c_int64


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtpchgen.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = ctypes.CDLL(path)
        for f in ("tpchgen_orders_count", "tpchgen_customer_count", "tpchgen_part_count",
                  "tpchgen_supplier_count", "tpchgen_lineitem_count", "tpchgen_lineitem",
                  "tpchgen_orders", "tpchgen_customer", "tpchgen_part", "tpchgen_partsupp",
                  "tpchgen_supplier"):
            getattr(_LIB, f).restype = _i64
        _LIB.tpchgen_days_from_civil.restype = ctypes.c_int32
    return _LIB


def _dict(sym, n):
    arr = (ctypes.c_char_p * n).in_dll(lib(), sym)
    return [arr[i].decode() for i in range(n)]


def nation_names():
    return _dict("TPCHGEN_NATION_NAMES", 25)


def region_names():
    return _dict("TPCHGEN_REGION_NAMES", 5)


def nation_regions():
    return list((ctypes.c_int32 * 25).in_dll(lib(), "TPCHGEN_NATION_REGION"))


SHIPMODE_DICT = ["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"]
SHIPINSTRUCT_DICT = ["COLLECT COD", "DELIVER IN PERSON", "NONE", "TAKE BACK RETURN"]
ORDERPRIORITY_DICT = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]


def _fn_dict(fn, n):
    f = getattr(lib(), fn)
    f.restype = ctypes.POINTER(ctypes.c_char_p)
    arr = f()
    return [arr[i].decode() for i in range(n)]


def part_type_dict():
    return _fn_dict("tpchgen_part_type_dict", 150)


def part_container_dict():
    return _fn_dict("tpchgen_part_container_dict", 40)


def part_brand_dict():
    return _fn_dict("tpchgen_part_brand_dict", 25)


def colors():
    return _dict("TPCHGEN_COLORS", 92)


def days(y, m, d):
    """Civil date -> days since 1970-01-01."""
    return int(lib().tpchgen_days_from_civil(ctypes.c_int32(y), ctypes.c_int32(m),
                                             ctypes.c_int32(d)))


_LINEITEM = [("l_orderkey", np.int64), ("l_partkey", np.int32), ("l_suppkey", np.int32),
             ("l_linenumber", np.int32), ("l_quantity", np.int32),
             ("l_extendedprice", np.int64), ("l_discount", np.int64), ("l_tax", np.int64),
             ("l_returnflag", np.uint8), ("l_linestatus", np.uint8), ("l_shipdate", np.int32),
             ("l_commitdate", np.int32), ("l_receiptdate", np.int32), ("l_shipinstruct", np.uint8), ("l_shipmode", np.uint8)]
_ORDERS = [("o_orderkey", np.int64), ("o_custkey", np.int32), ("o_orderdate", np.int32),
           ("o_shippriority", np.int32), ("o_totalprice", np.int64), ("o_orderstatus", np.uint8), ("o_orderpriority", np.uint8),
           ("o_comment", np.uint8), ("o_comment_len", np.uint8)]
_CUSTOMER = [("c_custkey", np.int32), ("c_nationkey", np.int32), ("c_mktsegment", np.uint8), ("c_phone", np.uint8), ("c_acctbal", np.int64),
             ("c_address", np.uint8), ("c_address_len", np.uint8), ("c_comment", np.uint8), ("c_comment_len", np.uint8)]
_PART = [("p_partkey", np.int32), ("p_name_colors", np.uint8), ("p_brand", np.uint8), ("p_type", np.uint8), ("p_size", np.int32),
         ("p_container", np.uint8), ("p_mfgr", np.uint8)]
_PARTSUPP = [("ps_partkey", np.int32), ("ps_suppkey", np.int32), ("ps_supplycost", np.int64), ("ps_availqty", np.int32)]
_SUPPLIER = [("s_suppkey", np.int32), ("s_nationkey", np.int32), ("s_address", np.uint8), ("s_address_len", np.uint8), ("s_phone", np.uint8),
             ("s_acctbal", np.int64), ("s_comment", np.uint8), ("s_comment_len", np.uint8), ("s_complaint", np.uint8)]
S_ADDRESS_STRIDE, S_PHONE_LEN = 40, 15
O_COMMENT_STRIDE, C_COMMENT_STRIDE, S_COMMENT_STRIDE = 80, 120, 104
MFGR_DICT = [f"Manufacturer#{i}" for i in range(1, 6)]


def _varlen(cols, name, stride, n):
    """fixed-stride characters + lengths (as the generator writes them) -> <name>_off (int32[n+1]) and <name>_bytes"""
    lens = cols.pop(name + "_len").astype(np.int64)
    a = cols.pop(name).reshape(n, stride)
    off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    cols[name + "_off"] = off
    cols[name + "_bytes"] = a[np.arange(stride)[None, :] < lens[:, None]]


def text_pool():
    """the 300 MiB text every COMMENT column is cut from (built on first use), as a numpy uint8 view"""
    size = _i64()
    f = lib().tpchgen_text_pool
    f.restype = ctypes.c_void_p
    p = f(ctypes.byref(size))
    return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(size.value,))


def _fixed_comments(fn, n):
    out = []
    for i in range(n):
        buf = ctypes.create_string_buffer(128)
        k = getattr(lib(), fn)(ctypes.c_int32(i), buf)
        out.append(buf.raw[:k].decode())
    return out


def nation_comments():
    return _fixed_comments("tpchgen_nation_comment", 25)


def region_comments():
    return _fixed_comments("tpchgen_region_comment", 5)


def _gen(fn, layout, nrows, sf, first, n, columns, width=None):
    want = [c for c, _ in layout] if columns is None else list(columns)
    cols = {}
    ptrs = []
    for name, dt in layout:
        if name in want:
            w = (width or {}).get(name, 1)
            cols[name] = np.empty(nrows * w, dtype=dt)
            ptrs.append(ctypes.c_void_p(cols[name].ctypes.data))
        else:
            ptrs.append(ctypes.c_void_p(None))
    struct = (ctypes.c_void_p * len(layout))(*ptrs)
    got = fn(_i64(sf[0]), _i64(sf[1]), _i64(first), _i64(n), struct)
    assert got == nrows, (got, nrows)
    return cols


def orders_count(sf):
    return int(lib().tpchgen_orders_count(_i64(sf[0]), _i64(sf[1])))


def lineitem_count(sf, first_order=0, n_orders=None):
    if n_orders is None:
        n_orders = orders_count(sf) - first_order
    return int(lib().tpchgen_lineitem_count(_i64(sf[0]), _i64(sf[1]), _i64(first_order),
                                            _i64(n_orders)))


def lineitem(sf, first_order=0, n_orders=None, columns=None):
    """sf = (num, den). Rows of orders [first_order, first_order + n_orders)."""
    if n_orders is None:
        n_orders = orders_count(sf) - first_order
    nrows = lineitem_count(sf, first_order, n_orders)
    return _gen(lib().tpchgen_lineitem, _LINEITEM, nrows, sf, first_order, n_orders, columns)


ORDERSTATUS_DICT = ["F", "O", "P"]


def orders(sf, first=0, n=None, columns=None):
    """o_orderstatus comes as a code into ORDERSTATUS_DICT (the generator writes the raw byte 'F' / 'O' / 'P')"""
    if n is None:
        n = orders_count(sf) - first
    want = [c for c, _ in _ORDERS if not c.startswith("o_comment")] if columns is None else list(columns)
    if "o_comment" in want:   # (asked for by name: 19..78 characters of the text pool per order, as o_comment_off / o_comment_bytes)
        want.append("o_comment_len")
    cols = _gen(lib().tpchgen_orders, _ORDERS, n, sf, first, n, want, width={"o_comment": O_COMMENT_STRIDE})
    if "o_comment" in cols:
        _varlen(cols, "o_comment", O_COMMENT_STRIDE, n)
    if "o_orderstatus" in cols:
        lut = np.zeros(256, np.uint8)
        lut[ord("O")], lut[ord("P")] = 1, 2
        cols["o_orderstatus"] = lut[cols["o_orderstatus"]]
    return cols


def customer(sf, first=0, n=None, columns=None, text=False):
    """c_name is 'Customer#' + the key as nine digits (TPC-H 4.2.3): derived here as offsets + bytes (c_name_off / c_name_bytes).
    text=True (or naming them in `columns`) adds c_address and c_comment (Q10's select list), as offsets + bytes"""
    if n is None:
        n = int(lib().tpchgen_customer_count(_i64(sf[0]), _i64(sf[1]))) - first
    if columns is None:
        want = [c for c, _ in _CUSTOMER if text or not (c.startswith("c_address") or c.startswith("c_comment"))]
    else:
        want = list(columns)
        for c in ("c_address", "c_comment"):
            if c in want:
                want.append(c + "_len")
    cols = _gen(lib().tpchgen_customer, _CUSTOMER, n, sf, first, n, want, width={"c_phone": 15, "c_address": S_ADDRESS_STRIDE, "c_comment": C_COMMENT_STRIDE})
    if "c_address" in cols:
        _varlen(cols, "c_address", S_ADDRESS_STRIDE, n)
    if "c_comment" in cols:
        _varlen(cols, "c_comment", C_COMMENT_STRIDE, n)
    if "c_phone" in cols:
        cols["c_phone_bytes"] = cols.pop("c_phone")
        cols["c_phone_off"] = np.arange(0, 15 * (n + 1), 15, dtype=np.int32)
    if columns is None or "c_name" in columns:
        keys = np.arange(first + 1, first + n + 1, dtype=np.int64)
        buf = np.empty((n, 18), dtype=np.uint8)
        buf[:, :9] = np.frombuffer(b"Customer#", dtype=np.uint8)
        for d in range(9):
            buf[:, 17 - d] = ord("0") + (keys // 10 ** d) % 10
        cols["c_name_off"] = np.arange(0, 18 * (n + 1), 18, dtype=np.int32)
        cols["c_name_bytes"] = buf.reshape(-1)
    return cols


def part(sf, first=0, n=None):
    """p_partkey plus p_name as (offsets int32[n+1], bytes uint8[]) built from the 5 colour words."""
    if n is None:
        n = int(lib().tpchgen_part_count(_i64(sf[0]), _i64(sf[1]))) - first
    cols = _gen(lib().tpchgen_part, _PART, n, sf, first, n, None, width={"p_name_colors": 5})
    words = colors()
    wl = np.array([len(w) for w in words], dtype=np.int32)
    codes = cols["p_name_colors"].reshape(n, 5)
    lens = wl[codes].sum(axis=1) + 4
    off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    # assemble the bytes word by word (vectorised over rows)
    buf = np.full(int(off[-1]), ord(" "), dtype=np.uint8)
    wbytes = [np.frombuffer(w.encode(), dtype=np.uint8) for w in words]
    pos = off[:-1].astype(np.int64).copy()
    for k in range(5):
        ck = codes[:, k]
        for c in np.unique(ck):
            rows = np.nonzero(ck == c)[0]
            wb = wbytes[c]
            idx = pos[rows][:, None] + np.arange(len(wb))[None, :]
            buf[idx] = wb[None, :]
        pos += wl[ck] + 1
    cols["p_name_off"] = off
    cols["p_name_bytes"] = buf
    return cols


def partsupp(sf, first_part=0, n_parts=None, columns=None):
    if n_parts is None:
        n_parts = int(lib().tpchgen_part_count(_i64(sf[0]), _i64(sf[1]))) - first_part
    return _gen(lib().tpchgen_partsupp, _PARTSUPP, 4 * n_parts, sf, first_part, n_parts, columns)


def supplier(sf, first=0, n=None, columns=None, text=False):
    """s_name is 'Supplier#' + the key as nine digits (TPC-H 4.2.3); s_name, s_address, s_phone come as offsets + bytes (<col>_off / <col>_bytes).
    text=True (or naming it) adds s_comment — with the "Customer ... Complaints / Recommends" injection — as offsets + bytes; s_acctbal and
    s_complaint (1 = the injection Q16's LIKE selects) always come along by default (no text pool needed)."""
    if n is None:
        n = int(lib().tpchgen_supplier_count(_i64(sf[0]), _i64(sf[1]))) - first
    want = None if columns is None else list(columns)
    if want is None:
        raw = ["s_suppkey", "s_nationkey", "s_address", "s_address_len", "s_phone", "s_acctbal", "s_complaint"] + (["s_comment", "s_comment_len"] if text else [])
    else:
        raw = [c for c in want if c in ("s_suppkey", "s_nationkey", "s_acctbal", "s_complaint")]
        if "s_address" in want:
            raw += ["s_address", "s_address_len"]
        if "s_phone" in want:
            raw.append("s_phone")
        if "s_comment" in want:
            raw += ["s_comment", "s_comment_len"]
    cols = _gen(lib().tpchgen_supplier, _SUPPLIER, n, sf, first, n, raw, width={"s_address": S_ADDRESS_STRIDE, "s_phone": S_PHONE_LEN, "s_comment": S_COMMENT_STRIDE})
    if "s_address" in cols:
        _varlen(cols, "s_address", S_ADDRESS_STRIDE, n)
    if "s_comment" in cols:
        _varlen(cols, "s_comment", S_COMMENT_STRIDE, n)
    if "s_phone" in cols:
        cols["s_phone_bytes"] = cols.pop("s_phone")
        cols["s_phone_off"] = np.arange(0, S_PHONE_LEN * (n + 1), S_PHONE_LEN, dtype=np.int32)
    if want is None or "s_name" in want:
        keys = np.arange(first + 1, first + n + 1, dtype=np.int64)
        buf = np.empty((n, 18), dtype=np.uint8)
        buf[:, :9] = np.frombuffer(b"Supplier#", dtype=np.uint8)
        for d in range(9):
            buf[:, 17 - d] = ord("0") + (keys // 10 ** d) % 10
        cols["s_name_off"] = np.arange(0, 18 * (n + 1), 18, dtype=np.int32)
        cols["s_name_bytes"] = buf.reshape(-1)
    return cols
