"""ctypes binding of libplanhip.so (include/planhip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PH_JOIN_KEY_RANGE, PH_JOIN_FK_PROBES, PH_JOIN_KEYS_SORTED_UNIQUE, PH_JOIN_EXISTS_ONLY = 1, 2, 4, 8
PH_OK, PH_EINVAL, PH_EHIP, PH_EUNSUPPORTED, PH_EOVERFLOW, PH_ECAPACITY, PH_ECONSTRAINT = 0, -1, -2, -3, -4, -5, -6
PH_I32, PH_I64, PH_DATE, PH_DEC64, PH_CODE8, PH_F32, PH_F64, PH_STR = range(1, 9)
PH_EQ, PH_NE, PH_LT, PH_LE, PH_GT, PH_GE, PH_LIKE, PH_NOTLIKE = range(1, 9)
PH_X_COL, PH_X_CONST, PH_X_ADD, PH_X_SUB, PH_X_MUL = range(1, 6)
PH_A_SUM, PH_A_AVG, PH_A_COUNT, PH_A_MIN, PH_A_MAX, PH_A_COUNT_STAR, PH_A_COUNT_DISTINCT = range(1, 8)
PH_COMM_ID_BYTES = 128
PH_RED_SUM, PH_RED_MAX, PH_RED_MIN = 1, 2, 3

i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p

NP_TYPES = {PH_I32: np.int32, PH_I64: np.int64, PH_DATE: np.int32, PH_DEC64: np.int64,
            PH_CODE8: np.uint8, PH_F32: np.float32, PH_F64: np.float64}


class PlanHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"planhip error {code}: {msg}")
        self.code = code


class Col(ctypes.Structure):
    _fields_ = [("type", i32), ("scale", i32), ("data", vp), ("validity", vp), ("aux", vp),
                ("aux_bytes", i64)]


class Const(ctypes.Structure):
    _fields_ = [("type", i32), ("scale", i32), ("i", i64), ("f", ctypes.c_double),
                ("s", ctypes.c_char_p)]


class Pred(ctypes.Structure):
    _fields_ = [("col", i32), ("op", i32), ("k", Const)]


class Rpn(ctypes.Structure):
    _fields_ = [("op", i32), ("col", i32), ("ival", i64), ("scale", i32)]


class AggExpr(ctypes.Structure):
    _fields_ = [("kind", i32), ("nprog", i32), ("prog", Rpn * 12)]


class AggSpec(ctypes.Structure):
    _fields_ = [("kind", i32), ("arg", i32)]


class AggResult(ctypes.Structure):
    _fields_ = [("ngroups", i64), ("first_row", ctypes.POINTER(i64)), ("keys", ctypes.POINTER(i64)),
                ("sum_lo", ctypes.POINTER(ctypes.c_uint64)), ("sum_hi", ctypes.POINTER(i64)),
                ("count", ctypes.POINTER(ctypes.c_uint64)), ("scale", ctypes.POINTER(i32)),
                ("nkeys", i32), ("naggs", i32), ("key_null", ctypes.POINTER(ctypes.c_uint8))]


class RowsResult(ctypes.Structure):
    _fields_ = [("nrows", i64), ("ncols", i32), ("type", ctypes.POINTER(i32)), ("scale", ctypes.POINTER(i32)),
                ("values", ctypes.POINTER(ctypes.POINTER(i64))), ("offsets", ctypes.POINTER(ctypes.POINTER(i32))),
                ("bytes", ctypes.POINTER(ctypes.c_void_p))]


def _preload_torch_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm bundles its own libamdhip64.so / libhiprtc.so /
    librccl.so (SONAMEs libamdhip64.so.7, libhiprtc.so.7, librccl.so.1 — the ones libplanhip.so
    asks for), and its libraries ask for them by the unversioned names: if ROCm's copies were loaded
    first (through libplanhip.so's NEEDED entries) a later `import torch` would bring in a second
    runtime, whose device discovery then fails ("No HIP GPUs are available").
    So when torch is installed it is imported FIRST, and libplanhip.so then binds to the copies torch
    has loaded, by SONAME. Loading only torch's .so files here and importing torch later was tried:
    the process then aborts at exit ("double free or corruption") because the libraries' static
    destructors run in an order torch never sees (measured on the GPU box, round 2)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    try:
        import torch  # noqa: F401  (initialises torch's bundled ROCm libraries in torch's own order)
    except Exception:  # noqa: BLE001 - a broken torch install: fall back to its bare libraries
        pass
    for name in ("libamdhip64.so", "libhiprtc.so", "librccl.so"):
        cand = os.path.join(os.path.dirname(spec.origin), "lib", name)
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libplanhip.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing — the HIP extension is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(path)
        for name, rt in (("ph_last_error", ctypes.c_char_p), ("ph_version", ctypes.c_char_p),
                         ("ph_scan_plan_kind", ctypes.c_char_p), ("ph_join_kind", ctypes.c_char_p), ("ph_table_rows", i64),
                         ("ph_hash_bytes", ctypes.c_uint64), ("ph_join_count", i64),
                         ("ph_comm_nranks", i32), ("ph_comm_rank", i32)):
            getattr(L, name).restype = rt  # a missing symbol raises: the ABI must be complete
        _LIB = L
    return _LIB


def last_error():
    return lib().ph_last_error().decode()


def check(rc):
    if rc != PH_OK:
        raise PlanHipError(rc, lib().ph_last_error().decode())


def const(typ, i=0, f=0.0, s=None, scale=0):
    k = Const()
    k.type, k.scale, k.i, k.f = typ, scale, int(i), float(f)
    k.s = None if s is None else s.encode()
    return k


def pred(col, op, k):
    p = Pred()
    p.col, p.op, p.k = col, op, k
    return p


def aggexpr(kind, prog=()):
    a = AggExpr()
    a.kind = kind
    a.nprog = len(prog)
    for j, (op, col, ival, scale) in enumerate(prog):
        a.prog[j] = Rpn(op, col, ival, scale)
    return a


def X_COL(c):
    return (PH_X_COL, c, 0, 0)


def X_CONST(v, scale=0):
    return (PH_X_CONST, -1, v, scale)


X_ADD = (PH_X_ADD, -1, 0, 0)
X_SUB = (PH_X_SUB, -1, 0, 0)
X_MUL = (PH_X_MUL, -1, 0, 0)


class Ctx:
    """One HIP device + one stream (ph_ctx)."""

    def __init__(self, device=0, stream=None):
        self.h = vp()
        check(lib().ph_ctx_create(i32(device), ctypes.byref(self.h)))
        self.device = device
        if stream is not None:
            check(lib().ph_ctx_set_stream(self.h, vp(stream)))

    def sync(self):
        check(lib().ph_ctx_sync(self.h))

    def set_deferred_errors(self, on=True):
        """flags that would cost a host round trip each (expression overflow) are reported by the next
        call that reads anything back (ph_ctx_set_deferred_errors)"""
        check(lib().ph_ctx_set_deferred_errors(self.h, i32(2 if on == 2 else 1 if on else 0)))   # 2: held until check_deferred()

    def check_deferred(self):
        check(lib().ph_ctx_check_deferred(self.h))

    def set_async_counts(self, on=True):
        """row counts of filter_select / probe_inner* are filled in by wait_counts() (ph_ctx_set_async_counts)"""
        check(lib().ph_ctx_set_async_counts(self.h, i32(1 if on else 0)))

    def wait_counts(self):
        check(lib().ph_ctx_wait_counts(self.h))

    def close(self):
        if self.h:
            lib().ph_ctx_destroy(self.h)
            self.h = None

    # -- plain device buffers
    def alloc(self, nbytes):
        p = vp()
        check(lib().ph_dev_alloc(self.h, i64(nbytes), ctypes.byref(p)))
        return p

    def free(self, p):
        check(lib().ph_dev_free(self.h, p))

    def free_many(self, ptrs):
        ptrs = [p for p in ptrs if p]
        if ptrs:
            arr = (vp * len(ptrs))(*[vp(p.value) if isinstance(p, vp) else vp(p) for p in ptrs])
            check(lib().ph_dev_free_many(self.h, arr, i64(len(ptrs))))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        check(lib().ph_dev_upload(self.h, p, vp(arr.ctypes.data), i64(arr.nbytes)))
        return p

    def download(self, p, dtype, n):
        out = np.empty(n, dtype=dtype)
        check(lib().ph_dev_download(self.h, vp(out.ctypes.data), p, i64(out.nbytes)))
        return out


def host_col(typ, arr, scale=0, validity=None, dictionary=None, aux=None):
    """ph_col over host numpy memory (for ph_table_create). Returns (Col, keepalive)."""
    c = Col()
    c.type, c.scale = typ, scale
    arr = np.ascontiguousarray(arr)
    c.data = arr.ctypes.data
    keep = [arr]
    if validity is not None:
        validity = np.ascontiguousarray(validity, dtype=np.uint8)
        c.validity = validity.ctypes.data
        keep.append(validity)
    if dictionary is not None:
        blob = b"".join(s.encode() + b"\0" for s in dictionary)
        buf = ctypes.create_string_buffer(blob, len(blob))
        c.aux = ctypes.cast(buf, vp)
        c.aux_bytes = len(blob)
        keep.append(buf)
    if aux is not None:
        aux = np.ascontiguousarray(aux)
        c.aux = aux.ctypes.data
        c.aux_bytes = aux.nbytes
        keep.append(aux)
    return c, keep


class Table:
    """Device-resident table (ph_table). cols: list of (type, array, scale, dictionary|None)."""

    def __init__(self, ctx, cols, nrows):
        self.ctx = ctx
        arr = (Col * len(cols))()
        keep = []
        for j, spec in enumerate(cols):
            c, k = host_col(*spec) if not isinstance(spec, dict) else host_col(**spec)
            arr[j] = c
            keep.append(k)
        self.h = vp()
        check(lib().ph_table_create(ctx.h, i32(len(cols)), arr, i64(nrows), ctypes.byref(self.h)))
        self.nrows = nrows
        self.ncols = len(cols)

    def col(self, c):
        out = Col()
        check(lib().ph_table_col(self.h, i32(c), ctypes.byref(out)))
        return out

    def col_run_len(self, c):
        """ph_table_col_run_len: > 0 = the column is stored in runs of that one length over consecutive values"""
        lib().ph_table_col_run_len.restype = ctypes.c_int32
        return int(lib().ph_table_col_run_len(self.h, i32(c)))

    def col_range(self, c):
        mn, mx = i64(), i64()
        check(lib().ph_table_col_range(self.h, i32(c), ctypes.byref(mn), ctypes.byref(mx)))
        return mn.value, mx.value

    def colocate(self, cols):
        """ph_table_colocate: a co-located (row-major) copy of these columns beside the column arrays; ph_gather_multi over
        views of them then reads one sector per row id"""
        check(lib().ph_table_colocate(self.h, i32(len(cols)), (i32 * len(cols))(*cols)))

    def colocated(self, cols):
        return bool(lib().ph_table_colocated(self.h, i32(len(cols)), (i32 * len(cols))(*cols)))

    def set_replicated(self, on=True):
        """multi-rank plans: this rank holds ALL rows of the table (default: a shard)"""
        check(lib().ph_table_set_replicated(self.h, i32(1 if on else 0)))

    def set_colocate_budget(self, nbytes):
        """bytes of HBM the library may spend on its own on co-located copies of this table (0 = never; default 4 GiB)"""
        check(lib().ph_table_set_colocate_budget(self.h, i64(nbytes)))

    def colocate_bytes(self):
        lib().ph_table_colocate_bytes.restype = i64
        return int(lib().ph_table_colocate_bytes(self.h))

    def free(self):
        if self.h:
            lib().ph_table_free(self.h)
            self.h = None


def _result(rp):
    r = rp.contents
    ng, nk, na = r.ngroups, r.nkeys, r.naggs
    out = {
        "ngroups": ng,
        "first_row": np.array([r.first_row[g] for g in range(ng)], dtype=np.int64),
        "keys": np.array([r.keys[i] for i in range(ng * nk)], dtype=np.int64).reshape(ng, nk),
        "scale": [r.scale[a] for a in range(na)],
        # python ints: exact 128-bit sums
        "sum": [[(int(r.sum_hi[g * na + a]) << 64) + int(r.sum_lo[g * na + a]) for a in range(na)]
                for g in range(ng)],
        "count": [[int(r.count[g * na + a]) for a in range(na)] for g in range(ng)],
        # 1 = the group's key is NULL (None when the result has no NULL-able key)
        "key_null": np.array([r.key_null[i] for i in range(ng * nk)], dtype=np.uint8).reshape(ng, nk) if r.key_null else None,
    }
    lib().ph_agg_result_free(rp)
    return out


class ScanPlan:
    """Fused Agg <- Scan(filter) over a resident table (ph_scan_plan)."""

    def __init__(self, ctx, table, preds, group_cols, aggs):
        self.ctx, self.table = ctx, table
        pa = (Pred * max(len(preds), 1))(*preds)
        ga = (i32 * max(len(group_cols), 1))(*group_cols)
        aa = (AggExpr * len(aggs))(*aggs)
        self.h = vp()
        check(lib().ph_scan_plan_create(ctx.h, table.h, pa, i32(len(preds)), ga,
                                        i32(len(group_cols)), aa, i32(len(aggs)),
                                        ctypes.byref(self.h)))
        self._keep = (pa, ga, aa)

    @property
    def kind(self):
        return lib().ph_scan_plan_kind(self.h).decode()

    def run(self, row_begin=0, row_end=None):
        if row_end is None:
            row_end = self.table.nrows
        check(lib().ph_scan_plan_run(self.h, i64(row_begin), i64(row_end)))

    def fetch(self):
        rp = ctypes.POINTER(AggResult)()
        check(lib().ph_scan_plan_fetch(self.h, ctypes.byref(rp)))
        return _result(rp)

    def partials_dev(self):
        """(device pointer, number of 64-bit words) of the raw partial result of the last run."""
        dev, n = vp(), i32()
        check(lib().ph_scan_plan_partials_dev(self.h, ctypes.byref(dev), ctypes.byref(n)))
        return dev.value, n.value

    def fetch_merged(self, words, nranks):
        """words: uint64 numpy array, the ranks' raw partials concatenated rank-major."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        rp = ctypes.POINTER(AggResult)()
        check(lib().ph_scan_plan_fetch_merged(self.h, vp(words.ctypes.data), i32(nranks), ctypes.byref(rp)))
        return _result(rp)

    def free(self):
        if self.h:
            lib().ph_scan_plan_free(self.h)
            self.h = None


# ---------------------------------------------------------------- operator-granular API

class DevColumn:
    """A device-resident column built from a numpy array (ph_dev_alloc + ph_dev_upload)."""

    def __init__(self, ctx, typ, arr, scale=0, validity=None, aux=None):
        self.ctx, self.type, self.scale = ctx, typ, scale
        arr = np.ascontiguousarray(arr)
        self.n = len(arr) - 1 if typ == PH_STR else len(arr)
        self.ptrs = []
        self.data = ctx.upload(arr) if arr.nbytes else ctx.alloc(8)
        self.ptrs.append(self.data)
        self.validity = None
        if validity is not None:
            self.validity = ctx.upload(np.ascontiguousarray(validity, dtype=np.uint8))
            self.ptrs.append(self.validity)
        self.aux, self.aux_bytes = None, 0
        if aux is not None:
            aux = np.ascontiguousarray(aux)
            self.aux = ctx.upload(aux) if aux.nbytes else ctx.alloc(8)
            self.aux_bytes = aux.nbytes
            self.ptrs.append(self.aux)

    def col(self):
        c = Col()
        c.type, c.scale = self.type, self.scale
        c.data = self.data
        c.validity = self.validity
        c.aux = self.aux
        c.aux_bytes = self.aux_bytes
        return c

    def free(self):
        for p in self.ptrs:
            self.ctx.free(p)
        self.ptrs = []


class TableColumn(DevColumn):
    """Column c of a resident Table, usable wherever a DevColumn is (col(), data, type, scale, n); freed with its table."""

    def __init__(self, table, c):
        self.table, self.index = table, c
        v = table.col(c)
        self.ctx, self.type, self.scale, self.data, self.validity, self.n = table.ctx, v.type, v.scale, v.data, v.validity, table.nrows
        self.aux, self.aux_bytes = v.aux, v.aux_bytes
        self.ptrs = []

    def col(self):
        return self.table.col(self.index)

    def free(self):
        pass


def _cols(cols):
    return (Col * max(len(cols), 1))(*[c.col() if isinstance(c, DevColumn) else c for c in cols])


def filter_select(ctx, col, n, op, k, sel_in=None, n_in=None, defer=False):
    """Returns (sel_out_dev, count). sel_in: device pointer or None. defer (with Ctx.set_async_counts):
    the count comes back as the ctypes int64 itself, valid after Ctx.wait_counts()."""
    if n_in is None:
        n_in = n
    out = ctx.alloc(max(n_in, 1) * 4)
    cnt = i64()
    c = col.col() if isinstance(col, DevColumn) else col
    check(lib().ph_filter_select(ctx.h, ctypes.byref(c), i64(n), i32(op), ctypes.byref(k),
                                 sel_in, i64(n_in), out, ctypes.byref(cnt)))
    return (out, cnt) if defer else (out, cnt.value)


def filter_select_and(ctx, col, n, op1, k1, op2, k2, sel_in=None, n_in=None):
    """ph_filter_select_and: two conjuncts over one column in one pass; (sel_out_dev, count). PH_EUNSUPPORTED when they are not two value ranges."""
    if n_in is None:
        n_in = n
    out = ctx.alloc(max(n_in, 1) * 4)
    cnt = i64()
    c = col.col() if isinstance(col, DevColumn) else col
    try:
        check(lib().ph_filter_select_and(ctx.h, ctypes.byref(c), i64(n), i32(op1), ctypes.byref(k1), i32(op2), ctypes.byref(k2), sel_in, i64(n_in), out, ctypes.byref(cnt)))
    except PlanHipError:
        ctx.free(out)
        raise
    return out, cnt.value


def filter_select_in(ctx, col, n, values, sel_in=None, n_in=None):
    """ph_filter_select_in: col IN (values) in one pass over an INTEGER or dictionary-code column; (sel_out_dev, count)"""
    if n_in is None:
        n_in = n
    out = ctx.alloc(max(n_in, 1) * 4)
    cnt = i64()
    c = col.col() if isinstance(col, DevColumn) else col
    vals = (ctypes.c_int64 * max(len(values), 1))(*values)
    try:
        check(lib().ph_filter_select_in(ctx.h, ctypes.byref(c), i64(n), vals, i32(len(values)), sel_in, i64(n_in), out, ctypes.byref(cnt)))
    except PlanHipError:
        ctx.free(out)
        raise
    return out, cnt.value


def sel_union(ctx, sels, counts, n_rows):
    """OR of predicates: ascending union of the children's selections (device pointers).
    Returns (sel_out_dev, count)."""
    k = len(sels)
    arr = (vp * max(k, 1))(*sels)
    cnts = (i64 * max(k, 1))(*[int(c) for c in counts])
    out = ctx.alloc(max(min(n_rows, sum(int(c) for c in counts)), 1) * 4)
    n = i64()
    check(lib().ph_sel_union(ctx.h, arr, cnts, i32(k), i64(n_rows), out, ctypes.byref(n)))
    return out, n.value


def sel_difference(ctx, parent, n_parent, child, n_child, n_rows):
    """falseSel: rows of parent (None = identity over n_rows) not in child. Returns (sel_dev, count)."""
    out = ctx.alloc(max(n_parent, 1) * 4)
    n = i64()
    check(lib().ph_sel_difference(ctx.h, parent, i64(n_parent), child, i64(n_child), i64(n_rows), out, ctypes.byref(n)))
    return out, n.value


def scatter(ctx, values, sel, n, out_data, out_validity=None):
    """FillSwitch: out[sel[i]] = values[i] (values: positional ph_col / DevColumn)"""
    c = values.col() if isinstance(values, DevColumn) else values
    check(lib().ph_scatter(ctx.h, ctypes.byref(c), sel, i64(n), out_data, out_validity))


def sort_rows(ctx, keys, descending, sel, n):
    """ORDER BY: device array of the n row ids (sel[i] or i) in sorted order."""
    out = ctx.alloc(max(n, 1) * 4)
    desc = (i32 * len(keys))(*[1 if d else 0 for d in descending])
    check(lib().ph_sort_rows(ctx.h, _cols(keys), desc, i32(len(keys)), sel, i64(n), out))
    return out


def hash_cols(ctx, cols, n, dict_hashes=None):
    out = ctx.alloc(max(n, 1) * 8)
    dh = None
    if dict_hashes is not None:
        dh = (vp * len(cols))(*[d if d is not None else None for d in dict_hashes])
    check(lib().ph_hash(ctx.h, _cols(cols), dh, i32(len(cols)), i64(n), out))
    return out


def hash_bytes(b):
    return int(lib().ph_hash_bytes(ctypes.c_char_p(b), ctypes.c_uint64(len(b))))


def expr_scale(cols, prog):
    p = (Rpn * len(prog))(*[Rpn(*x) for x in prog])
    s = i32()
    check(lib().ph_expr_scale(_cols(cols), p, i32(len(prog)), ctypes.byref(s)))
    return s.value


def expr_eval(ctx, cols, prog, sel, n, want_validity=False):
    p = (Rpn * len(prog))(*[Rpn(*x) for x in prog])
    out = ctx.alloc(max(n, 1) * 8)
    val = ctx.alloc((n + 7) // 8 + 8) if want_validity else None
    check(lib().ph_expr_eval(ctx.h, _cols(cols), i32(len(cols)), p, i32(len(prog)), sel, i64(n),
                             out, val))
    return out, val


def float_eval(ctx, cols, prog, sel, n, truth=True, wide=False):
    """ph_float_eval: FLOAT / DOUBLE program over device columns -> device pointer of int32 truth values (truth=True) or float32 values"""
    arr = (Col * len(cols))(*_cols(cols))
    pr = (Rpn * len(prog))(*[Rpn(*x) for x in prog])
    out = ctx.alloc(max(n, 1) * 4)
    check(lib().ph_float_eval(ctx.h, arr, i32(len(cols)), pr, i32(len(prog)), i32(1 if wide else 0), sel, i64(n), i32(PH_I32 if truth else PH_F32), out, None))
    return out


class Agg:
    def __init__(self, ctx, key_types, aggs, expected_groups=1024):
        self.ctx = ctx
        self.nkeys, self.naggs = len(key_types), len(aggs)
        kt = (i32 * len(key_types))(*key_types)
        sp = (AggSpec * max(len(aggs), 1))(*[AggSpec(k, a) for k, a in aggs])
        self.h = vp()
        check(lib().ph_agg_create(ctx.h, i32(len(key_types)), kt, i32(len(aggs)), sp,
                                  i64(expected_groups), ctypes.byref(self.h)))

    def sink_sorted(self, keys, args, n, row_base=0):
        """ph_agg_sink_sorted: streaming aggregate over rows ordered by the group key (first sink only);
        False when the shape is not supported (the caller sinks with sink())"""
        rc = lib().ph_agg_sink_sorted(self.h, _cols(keys), _cols(args), i32(len(args)), i64(n), i64(row_base))
        if rc == PH_EUNSUPPORTED:
            return False
        check(rc)
        return True

    def sink(self, keys, args, sel, n, positional=False, row_base=0, mask=None):
        """mask: bit a set = aggregate a is updated (AddChunk's filter); None = all"""
        if mask is None:
            check(lib().ph_agg_sink(self.h, _cols(keys), _cols(args), i32(len(args)), sel, i64(n),
                                    i32(1 if positional else 0), i64(row_base)))
        else:
            check(lib().ph_agg_sink_masked(self.h, _cols(keys), _cols(args), i32(len(args)), sel, i64(n),
                                           i32(1 if positional else 0), i64(row_base), ctypes.c_uint32(mask)))

    def key_column(self, c, key_type, scale=0):
        """key column c of all groups as a device column (data ptr, validity ptr, ngroups, ph_col)"""
        ng = self.group_count()
        w = 4 if key_type in (PH_I32, PH_DATE) else 1 if key_type == PH_CODE8 else 8
        data = self.ctx.alloc(max(ng, 1) * w)
        valid = self.ctx.alloc((max(ng, 1) + 7) // 8 + 8)
        n = i64()
        check(lib().ph_agg_keys_dev(self.h, i32(c), data, valid, i64(ng), ctypes.byref(n)))
        col = Col()
        col.type, col.scale, col.data, col.validity = key_type, scale, data, valid
        return data, valid, n.value, col

    def group_count(self):
        n = i64()
        check(lib().ph_agg_group_count(self.h, ctypes.byref(n)))
        return n.value

    def finalize(self, python_ints=True, room=1024, where=None):
        """ph_agg_fetch with room for `room` groups first (one host round trip when they fit), again
        with the reported count when there are more. where = [(agg_index, op, Const, value_scale), ...]: ph_agg_fetch_where — only the
        groups whose aggregates satisfy every conjunct (HAVING on the device)"""
        na = max(self.naggs, 1)
        P = lambda a: vp(a.ctypes.data)
        m = max(room, 1)
        if where:
            wi = (i32 * len(where))(*[w[0] for w in where])
            wo = (i32 * len(where))(*[w[1] for w in where])
            wk = (Const * len(where))(*[w[2] for w in where])
            ws = (i32 * len(where))(*[w[3] for w in where])
        while True:
            first = np.zeros(m, np.int64)
            keys = np.zeros(m * self.nkeys, np.int64)
            knull = np.zeros(m * self.nkeys, np.uint8)
            lo = np.zeros(m * na, np.uint64)
            hi = np.zeros(m * na, np.int64)
            cnt = np.zeros(m * na, np.uint64)
            n = i64()
            if where:
                rc = lib().ph_agg_fetch_where(self.h, i32(len(where)), wi, wo, wk, ws, i64(m), ctypes.byref(n), P(first), P(keys), P(knull), P(lo), P(hi), P(cnt))
            else:
                rc = lib().ph_agg_fetch(self.h, i64(m), ctypes.byref(n), P(first), P(keys), P(knull), P(lo), P(hi), P(cnt))
            if rc == PH_ECAPACITY and n.value > m:
                m = n.value
                continue
            check(rc)
            break
        ng = n.value
        out = dict(ngroups=ng, first_row=first[:ng], keys=keys.reshape(m, self.nkeys)[:ng],
                   key_null=knull.reshape(m, self.nkeys)[:ng],
                   sum_lo=lo.reshape(m, na)[:ng, :self.naggs], sum_hi=hi.reshape(m, na)[:ng, :self.naggs],
                   count=cnt.reshape(m, na)[:ng, :self.naggs].astype(np.int64))
        if python_ints:  # exact 128-bit python ints (one C-level tolist per word array, then plain int arithmetic)
            hl, ll = hi[:ng * na].tolist(), lo[:ng * na].tolist()
            nag = self.naggs
            out["sum"] = [[(hl[g * na + a] << 64) + ll[g * na + a] for a in range(nag)] for g in range(ng)]
        return out

    def topk(self, agg_index, k, descending=True, cap=4096):
        """Groups whose aggregate `agg_index` is at least as good as the k-th best (>= k with ties)."""
        first = np.zeros(cap, np.int64)
        keys = np.zeros(cap * self.nkeys, np.int64)
        knull = np.zeros(cap * self.nkeys, np.uint8)
        na = max(self.naggs, 1)
        lo = np.zeros(cap * na, np.uint64)
        hi = np.zeros(cap * na, np.int64)
        cnt = np.zeros(cap * na, np.uint64)
        n = i64()
        P = lambda a: vp(a.ctypes.data)
        check(lib().ph_agg_topk(self.h, i32(agg_index), i32(1 if descending else 0), i64(k), i64(cap),
                                ctypes.byref(n), P(first), P(keys), P(knull), P(lo), P(hi), P(cnt)))
        m = n.value
        return dict(ngroups=m, first_row=first[:m], keys=keys.reshape(cap, self.nkeys)[:m],
                    key_null=knull.reshape(cap, self.nkeys)[:m], sum_lo=lo.reshape(cap, na)[:m, :self.naggs],
                    sum_hi=hi.reshape(cap, na)[:m, :self.naggs],
                    count=cnt.reshape(cap, na)[:m, :self.naggs].astype(np.int64))

    def free(self):
        if self.h:
            lib().ph_agg_free(self.h)
            self.h = None


class Join:
    def __init__(self, ctx, keys, sel, n, key_range=None, fk_probes=False, sorted_unique=False, exists_only=False):
        """key_range = (lo, hi) of the single key column from column statistics (Table.col_range): a
        direct table for dense keys; fk_probes: the probe side is a foreign key into these keys (node
        table, no Bloom bitmap) — ph_join_build_ex"""
        self.ctx = ctx
        self.h = vp()
        if key_range is None and not fk_probes and not sorted_unique and not exists_only:
            check(lib().ph_join_build(ctx.h, _cols(keys), i32(len(keys)), sel, i64(n), ctypes.byref(self.h)))
        else:
            flags = ((PH_JOIN_KEY_RANGE if key_range is not None else 0) | (PH_JOIN_FK_PROBES if fk_probes else 0) |
                     (PH_JOIN_KEYS_SORTED_UNIQUE if sorted_unique else 0) | (PH_JOIN_EXISTS_ONLY if exists_only else 0))
            lo, hi = key_range if key_range is not None else (0, 0)
            check(lib().ph_join_build_ex(ctx.h, _cols(keys), i32(len(keys)), sel, i64(n), i32(flags), i64(lo), i64(hi),
                                         ctypes.byref(self.h)))

    @classmethod
    def build_where(cls, ctx, keys, where_col, where_op, where_k, sel, n, key_range, sorted_unique=False):
        """ph_join_build_where_ex (Filter -> build fused, direct tables only); None when the shape is not fused.
        sorted_unique: PH_JOIN_KEYS_SORTED_UNIQUE (the gated sorted fill + occupancy bitmap)"""
        self = cls.__new__(cls)
        self.ctx, self.h = ctx, vp()
        w = where_col.col() if isinstance(where_col, DevColumn) else where_col
        rc = lib().ph_join_build_where_ex(ctx.h, _cols(keys), i32(len(keys)), ctypes.byref(w), i32(where_op), ctypes.byref(where_k),
                                          sel, i64(n), i32(PH_JOIN_KEYS_SORTED_UNIQUE if sorted_unique else 0),
                                          i64(key_range[0]), i64(key_range[1]), ctypes.byref(self.h))
        if rc == PH_EUNSUPPORTED:
            return None
        check(rc)
        return self

    @property
    def kind(self):
        return lib().ph_join_kind(self.h).decode()

    def pairs_ordered(self):
        """ph_join_pairs_ordered: inner probes emit their pairs in probe-row order (every form except 'radix')"""
        return bool(lib().ph_join_pairs_ordered(self.h))

    def count(self):
        return int(lib().ph_join_count(self.h))

    def probe_inner(self, keys, sel, n, cap, defer=False):
        """defer (with Ctx.set_async_counts): the pair count comes back as the ctypes int64 itself, valid
        after Ctx.wait_counts()"""
        op = self.ctx.alloc(max(cap, 1) * 4)
        ob = self.ctx.alloc(max(cap, 1) * 4)
        m = i64()
        check(lib().ph_join_probe_inner(self.h, _cols(keys), sel, i64(n), op, ob, i64(cap), ctypes.byref(m)))
        return (m if defer else m.value), op, ob

    def probe_inner_where(self, keys, where_col, where_op, where_k, sel, n, cap):
        """Filter -> probe in one pass; returns None when the shape is not fused (caller runs
        filter_select + probe_inner)"""
        op = self.ctx.alloc(max(cap, 1) * 4)
        ob = self.ctx.alloc(max(cap, 1) * 4)
        m = i64()
        w = where_col.col() if isinstance(where_col, DevColumn) else where_col
        rc = lib().ph_join_probe_inner_where(self.h, _cols(keys), ctypes.byref(w), i32(where_op), ctypes.byref(where_k),
                                             sel, i64(n), op, ob, i64(cap), ctypes.byref(m))
        if rc == PH_EUNSUPPORTED:
            self.ctx.free(op)
            self.ctx.free(ob)
            return None
        check(rc)
        return m.value, op, ob

    def probe_inner_residual(self, keys, where_col, where_op, where_k, build_flags, sel, n, cap):
        """ph_join_probe_inner_residual: pairs whose build row has build_flags[row] != 0; where_col may be
        None. Returns None when the shape is not supported (direct tables only)."""
        op = self.ctx.alloc(max(cap, 1) * 4)
        ob = self.ctx.alloc(max(cap, 1) * 4)
        m = i64()
        w = None
        if where_col is not None:
            w = ctypes.byref(where_col.col() if isinstance(where_col, DevColumn) else where_col)
        rc = lib().ph_join_probe_inner_residual(self.h, _cols(keys), w, i32(where_op), ctypes.byref(where_k) if where_k is not None else None,
                                                build_flags, sel, i64(n), op, ob, i64(cap), ctypes.byref(m))
        if rc == PH_EUNSUPPORTED:
            self.ctx.free(op)
            self.ctx.free(ob)
            return None
        check(rc)
        return m.value, op, ob

    def probe_mark_where(self, keys, where_col, where_op, where_k, n):
        """ph_join_probe_mark_where: byte flags (filter && key present) or None when not supported"""
        f = self.ctx.alloc(max(n, 1) + 16)
        w = where_col.col() if isinstance(where_col, DevColumn) else where_col
        rc = lib().ph_join_probe_mark_where(self.h, _cols(keys), ctypes.byref(w), i32(where_op), ctypes.byref(where_k), i64(n), f)
        if rc == PH_EUNSUPPORTED:
            self.ctx.free(f)
            return None
        check(rc)
        return f

    def lookup(self, keys, sel, n, stats=None):
        """N:1 lookup probe: device int32[n] of matching build rows (-1 = none); stats: optional
        device int32[2] (misses, multi-matches), zeroed by the caller"""
        out = self.ctx.alloc(max(n, 1) * 4)
        check(lib().ph_join_lookup(self.h, _cols(keys), sel, i64(n), out, stats))
        return out

    def lookup_strict(self, keys, sel, n):
        """ph_join_lookup_strict: a miss / multi-match is a deferred PH_ECONSTRAINT error of the ctx"""
        out = self.ctx.alloc(max(n, 1) * 4)
        check(lib().ph_join_lookup_strict(self.h, _cols(keys), sel, i64(n), out))
        return out

    def probe_mark(self, keys, sel, n):
        f = self.ctx.alloc(max(n, 1))
        check(lib().ph_join_probe_mark(self.h, _cols(keys), sel, i64(n), f))
        return f

    def free(self):
        if self.h:
            lib().ph_join_free(self.h)
            self.h = None


def merge_lookup(ctx, build_key, n_build, probe_key, sel, n, strict=False):
    """ph_merge_lookup: N:1 lookup of probe keys that arrive in key order into a unique, ascending key column;
    device int32[n] of build rows (-1 = none)"""
    out = ctx.alloc(max(n, 1) * 4)
    b = build_key.col() if isinstance(build_key, DevColumn) else build_key
    p = probe_key.col() if isinstance(probe_key, DevColumn) else probe_key
    check(lib().ph_merge_lookup(ctx.h, ctypes.byref(b), i64(n_build), ctypes.byref(p), sel, i64(n), i32(1 if strict else 0), out))
    return out


def join_sorted_pairs(ctx, build_key, n_build, probe_key, sel, n, cap):
    """ph_join_sorted_pairs: inner pairs against a key column stored in ascending order with duplicates, no table; (probe rows, build rows, count)"""
    op, ob = ctx.alloc(max(cap, 1) * 4), ctx.alloc(max(cap, 1) * 4)
    b = build_key.col() if isinstance(build_key, DevColumn) else build_key
    p = probe_key.col() if isinstance(probe_key, DevColumn) else probe_key
    m = ctypes.c_int64(0)
    check(lib().ph_join_sorted_pairs(ctx.h, ctypes.byref(b), i64(n_build), ctypes.byref(p), sel, i64(n), op, ob, i64(cap), ctypes.byref(m)))
    return op, ob, m.value


def join_run_lookup(ctx, build_key2, n_build, key1_min, run_len, probe_keys, sel, n, strict=False):
    """ph_join_run_lookup: N:1 lookup on (first key, second key) into a table stored in runs of run_len by the first key; device int32[n] (-1 = none)"""
    out = ctx.alloc(max(n, 1) * 4)
    b = build_key2.col() if isinstance(build_key2, DevColumn) else build_key2
    arr = (Col * 2)(*[k.col() if isinstance(k, DevColumn) else k for k in probe_keys])
    check(lib().ph_join_run_lookup(ctx.h, ctypes.byref(b), i64(n_build), i64(key1_min), i32(run_len), arr, sel, i64(n), i32(1 if strict else 0), out))
    return out


def count_by_key(ctx, child_key, child_sel, n_child, key_min, key_range, parent_key, parent_sel, n_parent):
    """ph_count_by_key: child rows counted by key, every parent row reads its count; (device int64[n_parent], device validity bits: count > 0)"""
    out, val = ctx.alloc(max(n_parent, 1) * 8), ctx.alloc((n_parent + 63) // 64 * 8 + 64)
    c = child_key.col() if isinstance(child_key, DevColumn) else child_key
    p = parent_key.col() if isinstance(parent_key, DevColumn) else parent_key
    check(lib().ph_count_by_key(ctx.h, ctypes.byref(c), child_sel, i64(n_child), i64(key_min), i64(key_range), ctypes.byref(p), parent_sel, i64(n_parent), out, val))
    return out, val


def gather(ctx, col, idx_dev, n):
    c = col.col() if isinstance(col, DevColumn) else col
    w = {PH_CODE8: 1, PH_I32: 4, PH_DATE: 4, PH_F32: 4}.get(c.type, 8)
    out = ctx.alloc(max(n, 1) * w)
    check(lib().ph_gather(ctx.h, ctypes.byref(c), idx_dev, i64(n), out))
    return out


def gather_multi(ctx, cols, idx_dev, n):
    """ph_gather_multi: several columns through one row-id array in one pass -> list of device pointers"""
    cs = [c.col() if isinstance(c, DevColumn) else c for c in cols]
    w = [{PH_CODE8: 1, PH_I32: 4, PH_DATE: 4, PH_F32: 4}.get(c.type, 8) for c in cs]
    outs = [ctx.alloc(max(n, 1) * wi) for wi in w]
    arr = (Col * len(cs))(*cs)
    po = (vp * len(cs))(*[vp(o.value) for o in outs])
    check(lib().ph_gather_multi(ctx.h, i32(len(cs)), arr, idx_dev, i64(n), po))
    return outs


def partition(ctx, key, sel, n, nparts):
    c = key.col() if isinstance(key, DevColumn) else key
    counts = (i64 * nparts)()
    perm = ctx.alloc(max(n, 1) * 4)
    check(lib().ph_partition(ctx.h, ctypes.byref(c), sel, i64(n), i32(nparts), counts, perm))
    return [counts[p] for p in range(nparts)], perm


def partition_dev(ctx, key, sel, n, nparts):
    """ph_partition_dev: (device int64[nparts] counts, device permutation); no host round trip."""
    c = key.col() if isinstance(key, DevColumn) else key
    counts = ctx.alloc(max(nparts, 1) * 8)
    perm = ctx.alloc(max(n, 1) * 4)
    check(lib().ph_partition_dev(ctx.h, ctypes.byref(c), sel, i64(n), i32(nparts), counts, perm))
    return counts, perm


def read_reduce(ctx, dev, nbytes, out_words_dev, grid=256):
    check(lib().ph_dev_read_reduce(ctx.h, dev, i64(nbytes), out_words_dev, i32(grid)))


PH_PART_YEAR, PH_PART_MONTH, PH_PART_DAY = 1, 2, 3


def date_extract(ctx, part, col, sel, n):
    c = col.col() if isinstance(col, DevColumn) else col
    out = ctx.alloc(max(n, 1) * 4)
    check(lib().ph_date_extract(ctx.h, i32(part), ctypes.byref(c), sel, i64(n), out))
    return out


def substring(ctx, col, offset, length, sel, n):
    """ph_substring: (offsets dev int32[n+1], bytes dev, nbytes) of substring(col FROM offset FOR length)"""
    c = col.col() if isinstance(col, DevColumn) else col
    cap = max(int(c.aux_bytes), 1) + 64
    off = ctx.alloc((max(n, 0) + 1) * 4)
    out = ctx.alloc(cap)
    nb = i64()
    check(lib().ph_substring(ctx.h, ctypes.byref(c), i64(offset), i64(length), sel, i64(n), off, out, i64(cap), ctypes.byref(nb)))
    return off, out, nb.value


def cross_pairs(ctx, n_left, n_right):
    tot = max(n_left * n_right, 1)
    ol, orr = ctx.alloc(tot * 4), ctx.alloc(tot * 4)
    check(lib().ph_cross_pairs(ctx.h, i64(n_left), i64(n_right), ol, orr))
    return ol, orr


# ---------------------------------------------------------------- resident plans (ph_plan_*)

PH_PN_SCAN, PH_PN_FILTER, PH_PN_JOIN, PH_PN_PROJECT, PH_PN_AGG = range(1, 6)
PH_JT_INNER, PH_JT_SEMI, PH_JT_ANTI, PH_JT_LEFT = 1, 2, 3, 4
PH_PE_COL, PH_PE_DECIMAL, PH_PE_YEAR = 1, 2, 3
PH_STAT_ASCENDING, PH_STAT_STRICT, PH_STAT_DECLARED_UNIQUE = 1, 2, 4


PH_PE_CASE = 4
PH_PE_SUBSTR = 5
PH_PE_FLOAT = 6
PH_X_DIV, PH_X_LT, PH_X_LE, PH_X_GT, PH_X_GE = 6, 7, 8, 9, 10
PH_B_CMP, PH_B_AND, PH_B_OR = 1, 2, 3
PH_COLREF = 9


class Bool(ctypes.Structure):
    _fields_ = [("kind", i32), ("col", i32), ("op", i32), ("k", Const), ("first_child", i32), ("nchildren", i32)]


class PlanExpr(ctypes.Structure):
    _fields_ = [("kind", i32), ("col", i32), ("nprog", i32), ("prog", Rpn * 12), ("nwhen", i32), ("when", ctypes.POINTER(Bool)),
                ("nelse", i32), ("else_prog", Rpn * 12), ("result_int", i32), ("sub_offset", i64), ("sub_length", i64), ("float_wide", i32)]


def bool_tree(expr):
    """Nested tuples -> a flat ph_bool array (node 0 = root, children of a node contiguous):
         ("cmp", col, op, Const) | ("colcmp", col, op, other_col) | ("and", child, ...) | ("or", child, ...)
         ("in", col, [Const, ...])  =  OR of '=' comparisons, as the reference binds IN lists"""
    nodes = []

    def norm(e):
        if e[0] == "in":
            return ("or",) + tuple(("cmp", e[1], PH_EQ, k) for k in e[2])
        return e

    def emit(e, at):
        e = norm(e)
        b = Bool()
        if e[0] == "cmp":
            b.kind, b.col, b.op, b.k = PH_B_CMP, e[1], e[2], e[3]
        elif e[0] == "colcmp":
            b.kind, b.col, b.op = PH_B_CMP, e[1], e[2]
            b.k = const(PH_COLREF, i=e[3])
        else:
            kids = e[1:]
            b.kind = PH_B_AND if e[0] == "and" else PH_B_OR
            b.first_child, b.nchildren = len(nodes), len(kids)
            base = len(nodes)
            nodes.extend([None] * len(kids))
            for j, kid in enumerate(kids):
                emit(kid, base + j)
        nodes[at] = b

    nodes.append(None)
    emit(expr, 0)
    return (Bool * len(nodes))(*nodes)


class PlanAgg(ctypes.Structure):
    _fields_ = [("kind", i32), ("arg", PlanExpr)]


class PlanNode(ctypes.Structure):
    _fields_ = [("kind", i32), ("child", i32 * 2), ("table", vp), ("ncols", i32), ("cols", ctypes.POINTER(i32)),
                ("npreds", i32), ("preds", ctypes.POINTER(Pred)), ("nbools", i32), ("bools", ctypes.POINTER(Bool)),
                ("join_type", i32), ("nkeys", i32),
                ("probe_keys", ctypes.POINTER(i32)), ("build_keys", ctypes.POINTER(i32)), ("nout", i32), ("out", ctypes.POINTER(i32)),
                ("nexprs", i32), ("exprs", ctypes.POINTER(PlanExpr)), ("ngroups", i32), ("groups", ctypes.POINTER(PlanExpr)),
                ("naggs", i32), ("aggs", ctypes.POINTER(PlanAgg))]


def pe_col(c):
    e = PlanExpr()
    e.kind, e.col = PH_PE_COL, c
    return e


def pe_year(c):
    e = PlanExpr()
    e.kind, e.col = PH_PE_YEAR, c
    return e


def X_F32(v):
    """a FLOAT literal of a PH_PE_FLOAT program (its float32 bits)"""
    return (PH_X_CONST, -1, int(np.float32(v).view(np.uint32)), 0)


def X_OP(op):
    return (op, -1, 0, 0)


def pe_float(prog, truth=True, wide=False):
    """FLOAT (wide=False) / DOUBLE arithmetic ending — truth=True — in a comparison whose 1 / 0 a Filter above tests (ph_float_eval)"""
    e = PlanExpr()
    e.kind, e.col, e.nprog, e.result_int, e.float_wide = PH_PE_FLOAT, -1, len(prog), 1 if truth else 0, 1 if wide else 0
    for i, r in enumerate(prog):
        e.prog[i] = Rpn(*r)
    return e


def pe_substr(c, offset, length):
    """substring(<VARCHAR column c> FROM offset FOR length): a VARCHAR computed in the plan (filter by = / <> / IN, group key)"""
    e = PlanExpr()
    e.kind, e.col, e.sub_offset, e.sub_length = PH_PE_SUBSTR, c, offset, length
    return e


def pe_dec(prog):
    e = PlanExpr()
    e.kind, e.col, e.nprog = PH_PE_DECIMAL, -1, len(prog)
    for j, x in enumerate(prog):
        e.prog[j] = Rpn(*x)
    return e


def pe_case(when, then_prog, else_prog, result_int=False, keep=None):
    """CASE WHEN <when: bool_tree()> THEN <then_prog> ELSE <else_prog> END; keep: a list the WHEN array is appended to
    (it must outlive Plan.create)"""
    e = PlanExpr()
    e.kind, e.col, e.nprog, e.nelse, e.result_int = PH_PE_CASE, -1, len(then_prog), len(else_prog), 1 if result_int else 0
    for j, x in enumerate(then_prog):
        e.prog[j] = Rpn(*x)
    for j, x in enumerate(else_prog):
        e.else_prog[j] = Rpn(*x)
    e.nwhen, e.when = len(when), when
    if keep is not None:
        keep.append(when)
    return e


def _i32arr(v):
    return (i32 * max(len(v), 1))(*v)


class Plan:
    """ph_plan: an operator subtree Agg <- [Project] <- Join* <- Scan over resident tables, described node by node
    (children first, the aggregate last) and lowered inside the library from the tables' statistics."""

    def __init__(self, ctx):
        self.ctx, self.nodes, self._keep, self.h = ctx, [], [], None

    def _add(self, n):
        self.nodes.append(n)
        return len(self.nodes) - 1

    def scan(self, table, cols, preds=(), bools=None):
        """preds: simple conjuncts over TABLE columns; bools: one more conjunct of any shape (bool_tree())"""
        n = PlanNode()
        n.kind, n.child[0], n.child[1] = PH_PN_SCAN, -1, -1
        n.table = table.h
        ca, pa = _i32arr(cols), (Pred * max(len(preds), 1))(*preds)
        self._keep += [ca, pa, table, bools]
        n.ncols, n.cols = len(cols), ca
        n.npreds, n.preds = len(preds), pa
        if bools is not None:
            n.nbools, n.bools = len(bools), bools
        return self._add(n)

    def filter(self, child, preds=(), bools=None):
        n = PlanNode()
        n.kind, n.child[0], n.child[1] = PH_PN_FILTER, child, -1
        pa = (Pred * max(len(preds), 1))(*preds)
        self._keep += [pa, bools]
        n.npreds, n.preds = len(preds), pa
        if bools is not None:
            n.nbools, n.bools = len(bools), bools
        return self._add(n)

    def join(self, probe, build, probe_keys, build_keys, out, join_type=PH_JT_INNER, residual=None):
        """residual: a bool_tree over [probe columns | build columns] — the join's non-equi condition (INNER keeps the pairs that satisfy it,
        SEMI / ANTI the probe rows with / without such a pair)"""
        n = PlanNode()
        n.kind, n.child[0], n.child[1] = PH_PN_JOIN, probe, build
        pk, bk, oa = _i32arr(probe_keys), _i32arr(build_keys), _i32arr(out)
        self._keep += [pk, bk, oa, residual]
        if residual is not None:
            n.nbools, n.bools = len(residual), residual
        n.join_type, n.nkeys, n.probe_keys, n.build_keys = join_type, len(probe_keys), pk, bk
        n.nout, n.out = len(out), oa
        return self._add(n)

    def project(self, child, exprs):
        n = PlanNode()
        n.kind, n.child[0], n.child[1] = PH_PN_PROJECT, child, -1
        ea = (PlanExpr * len(exprs))(*exprs)
        self._keep.append(ea)
        n.nexprs, n.exprs = len(exprs), ea
        return self._add(n)

    def agg(self, child, groups, aggs):
        """aggs: list of (ph_aggkind, PlanExpr or None)"""
        n = PlanNode()
        n.kind, n.child[0], n.child[1] = PH_PN_AGG, child, -1
        ga = (PlanExpr * max(len(groups), 1))(*groups)
        aa = (PlanAgg * len(aggs))()
        for j, (kind, arg) in enumerate(aggs):
            aa[j].kind = kind
            if arg is not None:
                aa[j].arg = arg
        self._keep += [ga, aa]
        n.ngroups, n.groups, n.naggs, n.aggs = len(groups), ga, len(aggs), aa
        return self._add(n)

    def create(self):
        arr = (PlanNode * len(self.nodes))(*self.nodes)
        self.h = vp()
        check(lib().ph_plan_create(self.ctx.h, arr, i32(len(self.nodes)), ctypes.byref(self.h)))
        return self

    def set_having(self, conjuncts):
        """conjuncts: hip.pred(result column, op, const) over the root's aggregate columns; raises PlanHipError(PH_EUNSUPPORTED) when the
        plan cannot apply them on the device (the caller filters the fetched groups)"""
        arr = (Pred * max(len(conjuncts), 1))(*conjuncts)
        self._keep.append(arr)
        check(lib().ph_plan_set_having(self.h, i32(len(conjuncts)), arr))

    def set_comm(self, comm, broadcast_rows=None):
        """multi-rank execution (ph_plan_set_comm): comm = a Comm (RCCL or the in-process transport); every rank creates, runs and fetches the same
        plan over its shard and receives the complete result"""
        check(lib().ph_plan_set_comm(self.h, comm.h if comm is not None else None))
        if broadcast_rows is not None:
            check(lib().ph_plan_set_broadcast_rows(self.h, i64(broadcast_rows)))

    def set_topk(self, agg_index, k, descending=True):
        check(lib().ph_plan_set_topk(self.h, i32(agg_index), i32(1 if descending else 0), i64(k)))

    def set_rows_topk(self, col, k, descending=True):
        """ph_plan_set_rows_topk: a join-rooted plan under ORDER BY <column col> [DESC] LIMIT k returns only the rows that can be among the first k"""
        check(lib().ph_plan_set_rows_topk(self.h, i32(col), i32(1 if descending else 0), i64(k)))

    def run(self):
        check(lib().ph_plan_run(self.h))

    def fetch(self):
        rp = ctypes.POINTER(AggResult)()
        check(lib().ph_plan_fetch(self.h, ctypes.byref(rp)))
        return _result(rp)

    def fetch_rows(self):
        """rows of a plan whose root is a join / filter / project (ph_plan_fetch_rows): {"nrows", "types", "scales", "columns"} — a fixed-width
        column is an int64 numpy array, a VARCHAR column a list of str"""
        rp = ctypes.POINTER(RowsResult)()
        check(lib().ph_plan_fetch_rows(self.h, ctypes.byref(rp)))
        r = rp.contents
        n, cols = r.nrows, []
        for c in range(r.ncols):
            if r.values[c]:
                cols.append(np.ctypeslib.as_array(r.values[c], shape=(max(n, 1),))[:n].copy())
            else:
                off = np.ctypeslib.as_array(r.offsets[c], shape=(n + 1,))
                raw = ctypes.string_at(r.bytes[c], int(off[n])) if n else b""
                cols.append([raw[off[i]:off[i + 1]].decode() for i in range(n)])
        out = dict(nrows=n, types=[r.type[c] for c in range(r.ncols)], scales=[r.scale[c] for c in range(r.ncols)], columns=cols)
        lib().ph_rows_result_free(rp)
        return out

    def explain(self):
        lib().ph_plan_explain.restype = ctypes.c_char_p
        return lib().ph_plan_explain(self.h).decode()

    def free(self):
        if self.h:
            lib().ph_plan_free(self.h)
            self.h = None


def table_col_stats(table, c):
    f = i32()
    check(lib().ph_table_col_stats(table.h, i32(c), ctypes.byref(f)))
    return f.value


def table_declare_unique(table, cols):
    check(lib().ph_table_declare_unique(table.h, i32(len(cols)), _i32arr(cols)))


def table_col_range_of(table, c):
    mn, mx = i64(), i64()
    check(lib().ph_table_col_range(table.h, i32(c), ctypes.byref(mn), ctypes.byref(mx)))
    return mn.value, mx.value


class StrDict:
    """ph_strdict: VARCHAR keys as int32 codes (the representative row of each distinct string)"""

    def __init__(self, ctx, col, sel, n):
        self.ctx, self.h = ctx, vp()
        c = col.col() if isinstance(col, DevColumn) else col
        self.codes = ctx.alloc(max(n, 1) * 4)
        check(lib().ph_strdict_build(ctx.h, ctypes.byref(c), sel, i64(n), self.codes, ctypes.byref(self.h)))
        self._keep = col

    def lookup(self, col, sel, n):
        c = col.col() if isinstance(col, DevColumn) else col
        out = self.ctx.alloc(max(n, 1) * 4)
        check(lib().ph_strdict_lookup(self.h, ctypes.byref(c), sel, i64(n), out))
        return out

    def free(self):
        if self.h:
            lib().ph_strdict_free(self.h)
            self.h = None
            self.ctx.free(self.codes)


def plan_key_info(plan, k):
    """(type, scale, table handle or None, column) of group key k of a ph_plan's last run"""
    t, s, tab, c = i32(), i32(), vp(), i32()
    check(lib().ph_plan_key_info(plan.h, i32(k), ctypes.byref(t), ctypes.byref(s), ctypes.byref(tab), ctypes.byref(c)))
    return t.value, s.value, tab.value, c.value


def table_strings(ctx, table, c, rows):
    """strings of the given rows of PH_STR column c (ph_table_strings); `table` is a Table or the raw handle plan_key_info returned"""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    n = len(rows)
    off = np.zeros(n + 1, np.int32)
    cap = 1 << 20
    buf = ctypes.create_string_buffer(cap)
    check(lib().ph_table_strings(ctx.h, table.h if hasattr(table, "h") else vp(table), i32(c), vp(rows.ctypes.data), i64(n), vp(off.ctypes.data), buf, i64(cap)))
    return [buf.raw[off[i]:off[i + 1]].decode() for i in range(n)]


class LocalGroup:
    """the rendezvous object of the in-process transport (ph_local_group): the ranks are threads of this process"""

    def __init__(self, nranks):
        self.h, self.nranks = vp(), nranks
        check(lib().ph_local_group_create(i32(nranks), ctypes.byref(self.h)))

    def free(self):
        if self.h:
            lib().ph_local_group_free(self.h)
            self.h = None


class Comm:
    """ph_comm over the in-process transport (group = LocalGroup) — the RCCL form lives in plan_amd.dist.RcclGroup"""

    def __init__(self, ctx, group, rank):
        self.h, self.ctx, self.rank, self.nranks = vp(), ctx, rank, group.nranks
        check(lib().ph_comm_init_local(ctx.h, group.h, i32(rank), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib().ph_comm_destroy(self.h)
            self.h = None
