"""Column load path: Arrow / parquet columns -> the device encodings, whole columns at a time.

Replaces the reference's COPY FROM parquet + scan materialisation (pkg/compute/executor_scan.go:272-309
readers, :410-466 parquetColToValue + Vector.SetValue, one VALUE at a time into 24-byte Decimal /
12-byte Date / malloc'd String cells) — SURVEY.md §8f rank 3. The parquet physical encodings are
already the narrow ones the device wants (DATE = int32 days, DECIMAL = unscaled integer), so the
columns go straight from Arrow buffers to pinned staging to HBM.

Arrow's validity bitmap has the same convention as pkg/util/bitmap.go (1 bit per row, LSB first,
1 = valid), so it is passed through untouched when the column has NULLs.
"""
import numpy as np

from . import hip


def _validity(arr, n):
    if arr.null_count == 0:
        return None
    buf = arr.buffers()[0]
    bits = np.frombuffer(buf, dtype=np.uint8)
    if arr.offset % 8 == 0:
        return bits[arr.offset // 8: arr.offset // 8 + (n + 7) // 8].copy()
    unpacked = np.unpackbits(bits, bitorder="little")[arr.offset: arr.offset + n]
    return np.packbits(unpacked, bitorder="little")


def arrow_to_spec(arr):
    """pyarrow.Array / ChunkedArray -> (ph type, numpy data, scale, validity, dictionary, aux)
    in plan_amd.hip.host_col's argument order."""
    import pyarrow as pa
    import pyarrow.compute as pc
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    n = len(arr)
    t = arr.type
    val = _validity(arr, n)
    if pa.types.is_int32(t):
        return (hip.PH_I32, arr.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int32, copy=False), 0, val, None, None)
    if pa.types.is_int64(t):
        return (hip.PH_I64, arr.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64, copy=False), 0, val, None, None)
    if pa.types.is_date32(t):
        days = arr.cast(pa.int32()).fill_null(0).to_numpy(zero_copy_only=False)
        return (hip.PH_DATE, days.astype(np.int32, copy=False), 0, val, None, None)
    if pa.types.is_decimal(t):
        if t.precision > 18:
            raise ValueError(f"DECIMAL({t.precision},{t.scale}) does not fit the int64 device encoding")
        # decimal128 values are 16-byte little-endian two's-complement unscaled integers
        raw = np.frombuffer(arr.buffers()[1], dtype=np.int64).reshape(-1, 2)[arr.offset: arr.offset + n]
        lo = raw[:, 0].copy()
        if not np.array_equal(raw[:, 1], lo >> 63):
            raise ValueError("decimal value outside the int64 range")
        if val is not None:
            lo[~np.unpackbits(val, bitorder="little")[:n].astype(bool)] = 0
        return (hip.PH_DEC64, lo, t.scale, val, None, None)
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        enc = pc.dictionary_encode(arr)
        if isinstance(enc, pa.ChunkedArray):
            enc = enc.combine_chunks()
        d = enc.dictionary.to_pylist()
        if len(d) <= 256:   # VARCHAR with few distinct values -> uint8 codes + dictionary
            order = sorted(range(len(d)), key=lambda i: d[i])          # codes in dictionary order
            remap = np.zeros(max(len(d), 1), np.uint8)
            for new, old in enumerate(order):
                remap[old] = new
            codes = remap[enc.indices.fill_null(0).to_numpy(zero_copy_only=False)]
            return (hip.PH_CODE8, codes.astype(np.uint8), 0, val, [d[i] for i in order], None)
        s = arr.cast(pa.string())
        off = np.frombuffer(s.buffers()[1], dtype=np.int32)[s.offset: s.offset + n + 1]
        data = np.frombuffer(s.buffers()[2], dtype=np.uint8)
        base = int(off[0])
        return (hip.PH_STR, (off - base).astype(np.int32), 0, val, None, data[base: int(off[-1])].copy())
    raise ValueError(f"arrow type {t} has no device encoding")


def table_from_arrow(ctx, tbl, columns=None):
    """pyarrow.Table -> resident hip.Table (column order = `columns` or the table's)."""
    names = list(columns) if columns is not None else tbl.column_names
    specs = [arrow_to_spec(tbl.column(c)) for c in names]
    t = hip.Table(ctx, specs, tbl.num_rows)
    t.column_names = names
    return t


def table_from_parquet(ctx, path, columns=None):
    """Reads only the pruned columns (the reference's scan reads the plan's pruned column list,
    executor_scan.go:61-143) and loads them resident."""
    import pyarrow.parquet as pq
    return table_from_arrow(ctx, pq.read_table(path, columns=columns), columns)


# ---------------------------------------------------------------- through the C-ABI (Arrow C data interface)

class _ArrowSchema(__import__("ctypes").Structure):
    pass


class _ArrowArray(__import__("ctypes").Structure):
    pass


def _declare_arrow_structs():
    import ctypes
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    _ArrowSchema._fields_ = [("format", ctypes.c_char_p), ("name", ctypes.c_char_p), ("metadata", vp), ("flags", i64), ("n_children", i64),
                             ("children", vp), ("dictionary", vp), ("release", ctypes.CFUNCTYPE(None, ctypes.POINTER(_ArrowSchema))), ("private_data", vp)]
    _ArrowArray._fields_ = [("length", i64), ("null_count", i64), ("offset", i64), ("n_buffers", i64), ("n_children", i64), ("buffers", vp),
                            ("children", vp), ("dictionary", vp), ("release", ctypes.CFUNCTYPE(None, ctypes.POINTER(_ArrowArray))), ("private_data", vp)]


_declare_arrow_structs()


def table_from_arrow_c(ctx, tbl, columns=None):
    """pyarrow.Table / RecordBatch -> resident table through ph_table_create_arrow: the batch is exported with the Arrow C data
    interface (what Go's arrow/cdata or any parquet reader hands a C library) and the LIBRARY maps the buffers to the device
    encodings — the load path a host without Python uses. Returns a hip.Table (column_names, dictionaries attached)."""
    import ctypes
    import pyarrow as pa
    if isinstance(tbl, pa.Table):
        names = list(columns) if columns is not None else tbl.column_names
        tbl = tbl.select(names).combine_chunks()
        batch = tbl.to_batches()[0] if tbl.num_rows else pa.RecordBatch.from_pylist([], schema=tbl.schema)
    else:
        batch = tbl
    sch, arr = _ArrowSchema(), _ArrowArray()
    batch._export_to_c(ctypes.addressof(arr), ctypes.addressof(sch))
    t = hip.Table.__new__(hip.Table)
    t.ctx, t.h = ctx, hip.vp()
    try:
        hip.check(hip.lib().ph_table_create_arrow(ctx.h, ctypes.byref(sch), ctypes.byref(arr), None, hip.i32(0), ctypes.byref(t.h)))
    finally:
        arr.release(ctypes.byref(arr))      # the consumer releases what it was given (C data interface protocol)
        sch.release(ctypes.byref(sch))
    t.nrows, t.ncols, t.column_names = batch.num_rows, batch.num_columns, batch.schema.names
    lib = hip.lib()
    lib.ph_table_dict_entry.restype = ctypes.c_char_p
    t.dicts = [[lib.ph_table_dict_entry(t.h, hip.i32(c), hip.i32(k)).decode() for k in range(max(lib.ph_table_dict_size(t.h, hip.i32(c)), 0))]
               for c in range(t.ncols)]
    return t
