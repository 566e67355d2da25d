"""Column load path: parquet / Arrow -> device encodings (plan_amd/loader.py)."""
import datetime
import decimal
import os

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, loader, queries, tpchgen

pa = pytest.importorskip("pyarrow")
import pyarrow.parquet as pq  # noqa: E402


def lineitem_arrow(L):
    n = len(L["l_shipdate"])
    epoch = datetime.date(1970, 1, 1)

    def dec(a):  # unscaled int64 (scale 2) -> decimal128(15,2)
        return pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in a], pa.decimal128(15, 2))
    rf = np.array(tpchgen.RETURNFLAG_DICT)[L["l_returnflag"]]
    ls = np.array(tpchgen.LINESTATUS_DICT)[L["l_linestatus"]]
    return pa.table({
        "l_quantity": pa.array(L["l_quantity"], pa.int32()),
        "l_extendedprice": dec(L["l_extendedprice"]),
        "l_discount": dec(L["l_discount"]),
        "l_tax": dec(L["l_tax"]),
        "l_returnflag": pa.array(rf.tolist(), pa.string()),
        "l_linestatus": pa.array(ls.tolist(), pa.string()),
        "l_shipdate": pa.array(L["l_shipdate"], pa.int32()).cast(pa.date32()),
    })


def test_arrow_columns_become_device_encodings(sf001, tmp_path):
    L = sf001["lineitem"]
    path = os.path.join(tmp_path, "lineitem.parquet")
    pq.write_table(lineitem_arrow(L), path)
    tbl = pq.read_table(path)
    want = {"l_quantity": (hip.PH_I32, 0), "l_extendedprice": (hip.PH_DEC64, 2), "l_discount": (hip.PH_DEC64, 2),
            "l_tax": (hip.PH_DEC64, 2), "l_returnflag": (hip.PH_CODE8, 0), "l_linestatus": (hip.PH_CODE8, 0),
            "l_shipdate": (hip.PH_DATE, 0)}
    for name, (typ, scale) in want.items():
        t, data, sc, val, d, aux = loader.arrow_to_spec(tbl.column(name))
        assert (t, sc, val) == (typ, scale, None)
        assert np.array_equal(data, L[name]), name       # same bytes the generator produced
        if typ == hip.PH_CODE8:
            assert d == (tpchgen.RETURNFLAG_DICT if name == "l_returnflag" else tpchgen.LINESTATUS_DICT)


def test_nulls_long_strings_and_offsets():
    a = pa.array([1, None, 3, None, 5, 6, 7, 8, 9], pa.int32()).slice(1, 7)   # offset not a multiple of 8
    t, data, sc, val, d, aux = loader.arrow_to_spec(a)
    assert t == hip.PH_I32 and data.tolist() == [0, 3, 0, 5, 6, 7, 8]
    assert np.unpackbits(val, bitorder="little")[:7].tolist() == [0, 1, 0, 1, 1, 1, 1]
    s = pa.array([f"name {i}" for i in range(1000)] + [None], pa.string())
    t, off, sc, val, d, byts = loader.arrow_to_spec(s)
    assert t == hip.PH_STR and len(off) == 1002 and off[0] == 0
    assert bytes(byts[off[7]:off[8]]).decode() == "name 7"
    dec = pa.array([decimal.Decimal("-1.50"), None, decimal.Decimal("12345678901234.56")], pa.decimal128(16, 2))
    t, data, sc, val, d, aux = loader.arrow_to_spec(dec)
    assert (t, sc) == (hip.PH_DEC64, 2) and data.tolist() == [-150, 0, 1234567890123456]
    with pytest.raises(ValueError):
        loader.arrow_to_spec(pa.array([decimal.Decimal(10) ** 30], pa.decimal128(38, 2)))


@pytest.mark.gpu
def test_parquet_to_resident_table_to_q1(sf001, tmp_path):
    L = sf001["lineitem"]
    path = os.path.join(tmp_path, "lineitem.parquet")
    pq.write_table(lineitem_arrow(L), path)
    ctx = hip.Ctx(0)
    cols = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
    t = loader.table_from_parquet(ctx, path, cols)
    p = queries.q1_plan(ctx, t)
    p.run()
    r = p.fetch()
    want = O.q1(L, queries.q1_shipdate_cutoff())
    assert r["ngroups"] == len(want) == 4
    for g, w in enumerate(want):
        assert tuple(r["keys"][g]) == (w.returnflag, w.linestatus)
        assert r["sum"][g][3] == w.sum_charge.unscaled(6) and r["count"][g][7] == w.count_order
    p.free(); t.free(); ctx.close()


@pytest.mark.gpu
def test_arrow_c_data_interface_load_matches_the_python_loader(sf001):
    """ph_table_create_arrow: a record batch exported through the Arrow C data interface becomes a resident
    table inside the LIBRARY — ints, dates, decimal128, low-cardinality strings as dictionary codes, long strings as offsets +
    bytes, NULLs with an unaligned slice offset — with the same device bytes the Python loader produces, and Q1 over it equals
    the oracle."""
    L = sf001["lineitem"]
    ctx = hip.Ctx(0)
    tbl = lineitem_arrow(L)
    t = loader.table_from_arrow_c(ctx, tbl)
    assert t.column_names == tbl.column_names and t.nrows == len(L["l_shipdate"])
    assert t.dicts[4] == tpchgen.RETURNFLAG_DICT and t.dicts[5] == tpchgen.LINESTATUS_DICT
    for c, name in enumerate(tbl.column_names):
        col = t.col(c)
        typ, data, scale, _v, _d, _a = loader.arrow_to_spec(tbl.column(name))
        assert (col.type, col.scale) == (typ, scale)
        got = ctx.download(hip.vp(col.data), data.dtype, len(data))
        assert np.array_equal(got, data), name
    p = queries.q1_plan(ctx, t)
    p.run()
    r = p.fetch()
    want = O.q1(L, queries.q1_shipdate_cutoff())
    assert r["ngroups"] == 4 and all(r["sum"][g][3] == w.sum_charge.unscaled(6) for g, w in enumerate(want))
    p.free(); t.free()
    # NULLs behind a slice offset that is not a multiple of 8, a long-string column, a dictionary-encoded array
    ints = pa.array([1, None, 3, None, 5, 6, 7, 8, 9, 10, None], pa.int32()).slice(1, 9)
    names = pa.array([f"name {i}" for i in range(300)][:9], pa.string())
    many = pa.array([f"long string number {i}" for i in range(1000)], pa.string())
    dic = pa.array(["b", "a", None, "b", "c", "a", "a", "b", "c"], pa.string()).dictionary_encode()
    t2 = loader.table_from_arrow_c(ctx, pa.record_batch([ints, names, dic], names=["i", "s", "d"]))
    c0, c2 = t2.col(0), t2.col(2)
    assert c0.type == hip.PH_I32 and ctx.download(hip.vp(c0.data), np.int32, 9).tolist() == [0, 3, 0, 5, 6, 7, 8, 9, 10]
    assert np.unpackbits(ctx.download(hip.vp(c0.validity), np.uint8, 2), bitorder="little")[:9].tolist() == [0, 1, 0, 1, 1, 1, 1, 1, 1]
    assert hip.table_col_range_of(t2, 0) == (0, 10)          # NULL slots do not reach the statistics
    assert c2.type == hip.PH_CODE8 and t2.dicts[2] == ["a", "b", "c"]
    assert ctx.download(hip.vp(c2.data), np.uint8, 9).tolist() == [1, 0, 0, 1, 2, 0, 0, 1, 2]
    t3 = loader.table_from_arrow_c(ctx, pa.record_batch([many], names=["m"]))
    c = t3.col(0)
    off = ctx.download(hip.vp(c.data), np.int32, 1001)
    byts = ctx.download(hip.vp(c.aux), np.uint8, int(c.aux_bytes))
    assert c.type == hip.PH_STR and bytes(byts[off[7]:off[8]]).decode() == "long string number 7" and off[1000] == c.aux_bytes
    t2.free(); t3.free(); ctx.close()
