"""Parity AT THE SIZE THE NUMBERS ARE QUOTED ON. bench.py's `tpch_operator_interface` times the eighteen reproduced
TPC-H queries at SF10 behind the C++ OperatorExec layer; form selection inside ph_plan depends on table sizes, so the SF1 goldens do not
cover the SF10 code paths. Here every one of those queries runs at SF10 through the same entry point bench.py times
(planhost_tpch_run -> limitExecutor <- gpuOrderExecutor <- gpuResidentPlanExecutor -> ph_plan) and its result text must equal, byte for
byte, the oracle's text for the same generator data — the reference's own method: whole result files (executor_bench.go:127-136).

The oracle runs live, in a child process on the host cores (scripts/oracle_sf10_texts.py: ~1.5 min with 6 workers) while the device side
loads and runs; its texts must also equal the committed fixtures tests/golden/sf10/oracle_q*.txt, which is what bench.py stamps its
`parity` field from."""
import ctypes
import os
import subprocess
import sys

import pytest

import oracle_queries as OQ
from plan_amd import hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "sf10")


@pytest.fixture(scope="module")
def oracle_dir(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("oracle_sf10"))
    log = open(os.path.join(out, "log.txt"), "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "oracle_sf10_texts.py"), out, "10", "1", "--jobs", "6"],
                            stdout=log, stderr=subprocess.STDOUT)
    state = {"proc": proc, "out": out}
    yield state
    if proc.poll() is None:
        proc.kill()
        proc.wait()


@pytest.fixture(scope="module")
def device_texts():
    """all eighteen queries at SF10 through the C++ operator layer, one load (the generator inside libplantpch makes the same rows)"""
    hip.lib()
    lib = ctypes.CDLL(os.path.join(ROOT, "plan_amd", "libplantpch.so"))
    lib.planhost_last_error.restype = ctypes.c_char_p
    ctx = hip.Ctx(0)
    db = ctypes.c_void_p()
    rc = lib.planhost_tpch_load(ctx.h, ctypes.c_int64(10), ctypes.c_int64(1), ctypes.byref(db))
    assert rc == 0, lib.planhost_last_error()
    texts, explains = {}, {}
    cap = 1 << 22
    buf, ex = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(1 << 16)
    for q in OQ.QUERIES:
        avg, best = ctypes.c_double(), ctypes.c_double()
        rc = lib.planhost_tpch_run(db, ctypes.c_int32(q), ctypes.c_int32(1), ctypes.c_int32(0), ctypes.byref(avg), ctypes.byref(best), buf, ctypes.c_int64(cap),
                                   ex, ctypes.c_int64(1 << 16))
        assert rc == 0, (q, lib.planhost_last_error())
        texts[q], explains[q] = buf.value.decode(), ex.value.decode()
    lib.planhost_tpch_free(db)
    ctx.close()
    return texts, explains


def wait_for(state, q, timeout=840):
    import time
    path = os.path.join(state["out"], f"q{q}.txt")
    t0 = time.time()
    while not os.path.exists(path):
        if state["proc"].poll() is not None and not os.path.exists(path):
            raise AssertionError("the oracle run ended without q%d: %s" % (q, open(os.path.join(state["out"], "log.txt")).read()[-2000:]))
        if time.time() - t0 > timeout:
            raise AssertionError("the oracle run did not produce q%d in %d s" % (q, timeout))
        time.sleep(0.5)
    return open(path).read()


@pytest.mark.parametrize("q", OQ.QUERIES)
def test_sf10_result_text_equals_the_oracle(q, oracle_dir, device_texts):
    texts, explains = device_texts
    want = wait_for(oracle_dir, q)
    assert want == open(os.path.join(FIX, f"oracle_q{q}.txt")).read(), f"tests/golden/sf10/oracle_q{q}.txt is stale: regenerate with scripts/oracle_sf10_texts.py"
    assert texts[q] == want, explains[q]
