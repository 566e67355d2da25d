"""The C++ host layer (plan_amd/csrc/host): the reference's OperatorExec interface and pkg/chunk
types mirrored in C++, driving the C-ABI from 2048-row chunks. host_tester plays the role of the
reference's `tester tpch1g --query_id N`: it pulls the executor tree like execOps and prints the
result in the reference's text format, so these tests compare whole result files."""
import os
import subprocess

import pytest

import oracle_lib as O
from plan_amd import tpchgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTER = os.path.join(ROOT, "plan_amd", "host_tester")
G = os.path.join(os.path.dirname(__file__), "golden")


def run(*args):
    return subprocess.run([TESTER, *args], check=True, capture_output=True, text=True, timeout=600).stdout


def test_chunk_serialize_roundtrip_and_value_text():
    """Chunk.Serialize -> Deserialize (the stub fixture format, chunk.go:168-194) keeps every cell,
    NULLs included, and Value.String prints them the reference's way."""
    want = ("-2\t0\t-0.07\ta\t1992-01-01\t25.5\t0\n"
            "NULL\t10000000000\t10.43\tbb\t1993-02-04\t25.833333333333332\t7\n"
            "0\t20000000000\t20.93\tNULL\t1994-03-11\t26.166666666666668\t14\n"
            "1\t30000000000\t31.43\t\t1995-04-15\t26.5\t21\n"
            "2\t40000000000\t41.93\teeeee\t1996-05-19\t26.833333333333332\t-18446744073709551588\n")
    assert run("roundtrip") == want


@pytest.mark.gpu
def test_q1_through_operator_interface_matches_reference_golden():
    # source -> gpuFilterExecutor -> gpuAggExecutor, SF1: the whole q1.txt, AVG rounding included
    assert run("q1", "1", "1") == open(os.path.join(G, "plan_q1.txt")).read()


@pytest.mark.gpu
def test_q1_via_stub_replay_matches_oracle(sf001):
    # chunks go through Serialize/Deserialize first (the reference's stubExecutor fixture path)
    rows = O.q1(sf001["lineitem"], tpchgen.days(1998, 12, 1) - 112)
    assert run("q1", "1", "100", "stub") == O.q1_text(rows)


@pytest.mark.gpu
def test_q6_through_operator_interface_matches_reference_golden():
    assert run("q6", "1", "1") == open(os.path.join(G, "plan_q6.txt")).read()


@pytest.mark.gpu
def test_q3_two_joins_through_operator_interface_matches_reference_golden():
    out = run("q3", "1", "1").split("\n")
    assert out[0] == "#\t\t\t"
    # the aggregate emits [group columns..., aggregates...]; the Project above it (outside the
    # hot path) puts them in select-list order: l_orderkey, revenue, o_orderdate, o_shippriority
    rows = [[k, rev, d, p] for k, d, p, rev in (l.split("\t") for l in out[1:] if l)]
    assert len(rows) == 11378
    from decimal import Decimal
    rows.sort(key=lambda r: (-Decimal(r[1]), r[2]))      # ORDER BY revenue DESC, o_orderdate
    text = "#\t\t\t\n" + "".join("\t".join(r) + "\n" for r in rows[:10])
    assert text == open(os.path.join(G, "plan_q3.txt")).read()


@pytest.mark.gpu
def test_resident_scan_agg_executor_matches_reference_golden():
    """gpuScanAggExecutor: table loaded once, Agg <- Scan(filter) as one fused plan, group rows
    finalised into pkg/chunk types (Hugeint / Decimal / double) — the measured mode behind the
    operator interface."""
    assert run("q1", "1", "1", "resident") == open(os.path.join(G, "plan_q1.txt")).read()
    assert run("q6", "1", "1", "resident") == open(os.path.join(G, "plan_q6.txt")).read()


@pytest.mark.gpu
def test_semi_and_anti_join_executors(sf001):
    """customer SEMI / ANTI JOIN orders on the customer key: the emitted customer keys are exactly
    those with / without an order (in TPC-H every third customer key never orders)."""
    import numpy as np
    ck = sf001["customer"]["c_custkey"]
    has = np.isin(ck, sf001["orders"]["o_custkey"])
    semi = [int(x) for x in run("semi", "1", "100").split("\n")[1:] if x]
    anti = [int(x) for x in run("anti", "1", "100").split("\n")[1:] if x]
    assert semi == ck[has].tolist() and anti == ck[~has].tolist()
    assert all(k % 3 == 0 for k in anti[:50]) and len(anti) >= len(ck) // 3


def test_vector_formats_unify_slice_and_serialize():
    """FLAT / CONST / DICT / SEQUENCE vectors through ToUnifiedFormat, Slice (selection of a
    selection included) and Serialize (vector_format.go:64-97, chunk.go:82-93)."""
    full = [f"{5 + 3 * i}\tk\tNULL\t{-10000000000 + 10000000000 * i}\t{'NULL' if i == 3 else 100 + i}" for i in range(6)]
    first = [full[i] for i in (4, 1, 3, 0)]
    second = [first[i] for i in (2, 0)]
    want = "\n".join(full + ["--"] + first + ["--"] + second + ["--"] + second) + "\n"
    assert run("formats") == want


@pytest.mark.gpu
def test_left_join_executor(sf001):
    """customer LEFT JOIN orders: every (custkey, orderkey) pair plus one (custkey, NULL) row per
    customer without orders (NextLeftJoin, join_scan.go:67-88)."""
    ck = sf001["customer"]["c_custkey"]
    oc, ok = sf001["orders"]["o_custkey"], sf001["orders"]["o_orderkey"]
    got = sorted(tuple(l.split("\t")) for l in run("left", "1", "100").split("\n")[1:] if l)
    import numpy as np
    never = ck[~np.isin(ck, oc)]
    want = sorted([(str(int(c)), str(int(o))) for c, o in zip(oc, ok)] + [(str(int(c)), "NULL") for c in never])
    assert got == want and len(never) > 0


@pytest.mark.gpu
def test_order_executor(sf001):
    """gpuOrderExecutor (orderExecutor, executor_order.go:56-138): customer ordered by
    c_mktsegment DESC (VARCHAR key), c_custkey % 97, c_custkey DESC."""
    C = sf001["customer"]
    segs = [tpchgen.MKTSEGMENT_DICT[c] for c in C["c_mktsegment"]]
    rows = sorted(zip(segs, (C["c_custkey"] % 97).tolist(), C["c_custkey"].tolist()),
                  key=lambda r: ([-b for b in r[0].encode()] + [1], r[1], -r[2]))
    got = [tuple(l.split("\t")) for l in run("order", "1", "100").split("\n")[1:] if l]
    assert got == [(s, str(a), str(b)) for s, a, b in rows]
