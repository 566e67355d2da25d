"""CPU-side checks: the C-ABI library loads and exports every symbol include/planhip.h declares,
the generator library likewise, and the host-only entry points agree with the oracle."""
import ctypes
import os
import re

import numpy as np

import oracle_lib as O
from plan_amd import hip, tpchgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_planhip_exports_every_declared_symbol():
    lib = hip.lib()
    names = declared("planhip.h", "ph_")
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"libplanhip.so does not export {n}"


def test_tpchgen_exports_every_declared_symbol():
    lib = tpchgen.lib()
    for n in declared("tpchgen.h", "tpchgen_"):
        assert hasattr(lib, n), f"libtpchgen.so does not export {n}"


def test_errors_are_codes_not_aborts():
    lib = hip.lib()
    rc = lib.ph_ctx_sync(None)
    assert rc == hip.PH_EINVAL and b"ctx is NULL" in lib.ph_last_error()
    rc = lib.ph_table_col(None, 0, None)
    assert rc == hip.PH_EINVAL


def test_hash_bytes_matches_reference_hashbytes():
    # util.HashBytes restated twice (oracle C, product C++): must agree on every length class
    rng = np.random.default_rng(0)
    for ln in list(range(0, 40)) + [63, 64, 65, 200]:
        b = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        off = np.array([0, ln], dtype=np.int32)
        arr = np.frombuffer(b, dtype=np.uint8) if ln else np.zeros(1, np.uint8)
        want = int(O.hash_cols([O.col(O.OT_VARCHAR, off, dictionary=arr)], 1)[0])
        assert hip.hash_bytes(b) == want


def test_expr_scale_follows_binder_rules():
    c = hip.Col(); c.type, c.scale = hip.PH_DEC64, 2
    q = hip.Col(); q.type, q.scale = hip.PH_I32, 0
    cols = [c, c, c, q]
    one = hip.X_CONST(1)
    dp = [hip.X_COL(0), one, hip.X_COL(1), hip.X_SUB, hip.X_MUL]
    assert hip.expr_scale(cols, dp) == 4                      # Mul: 2 + max(0, 2)
    assert hip.expr_scale(cols, dp + [one, hip.X_COL(2), hip.X_ADD, hip.X_MUL]) == 6
    assert hip.expr_scale(cols, dp + [hip.X_COL(2), hip.X_COL(3), hip.X_MUL, hip.X_SUB]) == 4
    try:
        hip.expr_scale(cols, [hip.X_COL(0), hip.X_MUL])
        assert False
    except hip.PlanHipError as e:
        assert e.code == hip.PH_EUNSUPPORTED


def test_generator_shards_equal_whole():
    """ranged generation (what each rank does) reproduces the same rows as one pass"""
    sf = (1, 100)
    whole = tpchgen.lineitem(sf)
    n_orders = tpchgen.orders_count(sf)
    parts = [tpchgen.lineitem(sf, a, b - a) for a, b in [(0, 1), (1, 5000), (5000, 5003), (5003, n_orders)]]
    for k in whole:
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts])), k
    assert len(whole["l_orderkey"]) == 60175   # SF0.01 lineitem cardinality


def test_plan_specialised_kernels_generate_and_compile_for_gfx950():
    """The hiprtc path needs no device to compile: every canned plan shape must generate a source
    that compiles for gfx950 (the generated source is what ph_scan_plan_create builds on the GPU)."""
    lib = hip.lib()
    buf = ctypes.create_string_buffer(1 << 16)
    for which in range(4):
        rc = lib.ph_scan_jit_selfcheck(which, buf, 1 << 16)
        assert rc == hip.PH_OK, lib.ph_last_error().decode()
        src = buf.value.decode()
        assert 'extern "C" __global__' in src and "__builtin_nontemporal_load" in src
    lib.ph_scan_jit_selfcheck(0, buf, 1 << 16)
    q1 = buf.value.decode()
    assert q1.count("__builtin_nontemporal_load") == 4 + 3 * 2   # 2 int32 + 2 byte columns, 3 int64 columns x 2
    assert lib.ph_scan_jit_selfcheck(9, None, 0) == hip.PH_EINVAL


def test_specialised_aggregate_sink_source_compiles_for_gfx950():
    """agg_sink.inc through hiprtc with a sink shape as compile-time constants (what ph_agg_sink
    does for calls of >= 2^20 rows): both canned sink shapes, and two shapes of the bulk build's partial
    kernel (agg_bulk2_partial_spec), compile for gfx950 without a device."""
    lib = hip.lib()
    for which in (0, 1, 2, 3):
        assert lib.ph_agg_jit_selfcheck(which) == hip.PH_OK, lib.ph_last_error().decode()
    assert lib.ph_agg_jit_selfcheck(7) == hip.PH_EINVAL


def test_generated_expression_kernels_compile_for_gfx950():
    lib = hip.lib()
    for which in range(3):
        assert lib.ph_expr_jit_selfcheck(which) == hip.PH_OK, lib.ph_last_error().decode()
    assert lib.ph_expr_jit_selfcheck(5) == hip.PH_EINVAL


def test_planhost_libraries_load_and_export_their_entry_points():
    """libplanhost.so (the C++ operator layer) and libplantpch.so (TPC-H resident plans + the C entry points bench.py
    uses) load next to libplanhip.so; no compute call without a GPU."""
    hip.lib()
    host = ctypes.CDLL(os.path.join(ROOT, "plan_amd", "libplanhost.so"))
    tpch = ctypes.CDLL(os.path.join(ROOT, "plan_amd", "libplantpch.so"))
    assert host is not None
    for n in ("planhost_tpch_load", "planhost_tpch_run", "planhost_tpch_rows", "planhost_tpch_free", "planhost_last_error"):
        assert hasattr(tpch, n), n
    tpch.planhost_last_error.restype = ctypes.c_char_p
    assert tpch.planhost_tpch_load(None, ctypes.c_int64(1), ctypes.c_int64(1), None) == hip.PH_EINVAL
    assert b"bad arguments" in tpch.planhost_last_error()


def test_plan_descriptor_validation_without_a_device():
    """ph_plan_create checks the descriptor on the host: the root is an aggregate, or a join / filter / project whose rows come back —
    never a scan; children precede parents"""
    lib = hip.lib()
    n = (hip.PlanNode * 2)()
    n[0].kind, n[1].kind = hip.PH_PN_SCAN, hip.PH_PN_SCAN
    out = hip.vp()
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_EINVAL
    assert b"root" in lib.ph_last_error()
    n[1].kind = hip.PH_PN_JOIN          # an acceptable root, but its children (0, 0 by default) do not both precede it as distinct inputs
    n[1].child[0], n[1].child[1] = 1, 0
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_EINVAL
    assert b"precede" in lib.ph_last_error()
    assert lib.ph_plan_create(None, n, hip.i32(2), ctypes.byref(out)) == hip.PH_EINVAL


def test_plan_descriptor_rejects_boolean_trees_that_point_backwards():
    """an AND / OR node's children must FOLLOW it in the flat array — a node naming itself (or an earlier node) as its child
    would send eval_bool into unbounded recursion. Checked by ph_plan_create on the host, for a scan's / filter's tree and for a CASE's WHEN."""
    lib = hip.lib()
    out = hip.vp()
    cols = (hip.i32 * 1)(0)
    # Filter(Scan) with the tree [OR(first_child=1, n=2), AND(first_child=1, n=1)  <- names itself, CMP]
    b = (hip.Bool * 3)()
    b[0].kind, b[0].first_child, b[0].nchildren = hip.PH_B_OR, 1, 2
    b[1].kind, b[1].first_child, b[1].nchildren = hip.PH_B_AND, 1, 1
    b[2].kind, b[2].col, b[2].op, b[2].k = hip.PH_B_CMP, 0, hip.PH_EQ, hip.const(hip.PH_I32, i=1)
    n = (hip.PlanNode * 2)()
    n[0].kind, n[0].child[0], n[0].child[1] = hip.PH_PN_SCAN, -1, -1
    n[0].table, n[0].ncols, n[0].cols = hip.vp(1), 1, cols
    n[1].kind, n[1].child[0], n[1].child[1] = hip.PH_PN_FILTER, 0, -1
    n[1].nbools, n[1].bools = 3, b
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_EINVAL
    assert b"follow their parent" in lib.ph_last_error()
    b[1].first_child = 2                     # now a proper tree: OR(AND(CMP), CMP)
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_OK
    lib.ph_plan_free(out)
    # the same inside a CASE's WHEN (an aggregate argument)
    w = (hip.Bool * 2)()
    w[0].kind, w[0].first_child, w[0].nchildren = hip.PH_B_AND, 0, 1      # names itself
    w[1].kind, w[1].col, w[1].op, w[1].k = hip.PH_B_CMP, 0, hip.PH_EQ, hip.const(hip.PH_I32, i=1)
    e = hip.PlanExpr()
    e.kind, e.nprog, e.nelse, e.nwhen, e.when = hip.PH_PE_CASE, 1, 1, 2, w
    e.prog[0] = hip.X_CONST(1)
    e.else_prog[0] = hip.X_CONST(0)
    aggs = (hip.PlanAgg * 1)()
    aggs[0].kind, aggs[0].arg = hip.PH_A_SUM, e
    n[1] = hip.PlanNode()
    n[1].kind, n[1].child[0], n[1].child[1] = hip.PH_PN_AGG, 0, -1
    n[1].naggs, n[1].aggs = 1, aggs
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_EINVAL
    assert b"malformed expression" in lib.ph_last_error()
    w[0].first_child = 1
    assert lib.ph_plan_create(hip.vp(1), n, hip.i32(2), ctypes.byref(out)) == hip.PH_OK
    lib.ph_plan_free(out)
