"""Q3 / Q9 pipelines against the oracle (tests/oracle_lib, test infrastructure) for other parameters than the goldens':
segments, cut-off dates, LIKE patterns, at SF1 (python scripts/soak_pipelines.py). Not part of the product path."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import oracle_lib as O
import tpch_data
from plan_amd import hip, pipelines, tpchgen

ctx = hip.Ctx(0)
t = tpch_data.load(1, 1)
for seg, (y, m, d) in (("BUILDING", (1995, 3, 15)), ("AUTOMOBILE", (1994, 1, 1)), ("MACHINERY", (1997, 6, 30)), ("FURNITURE", (1992, 2, 1)),
                       ("HOUSEHOLD", (1998, 12, 1)), ("NOSUCHSEGMENT", (1995, 3, 15))):
    date = tpchgen.days(y, m, d)
    p = pipelines.Q3Pipeline(ctx, t["lineitem"], t["orders"], t["customer"], segment=seg, date=date)
    p.time_stages = False
    r = p.run()
    p.free()
    n, rows = O.q3(t, seg, date)
    want = O.q3_text(rows, n)
    got = pipelines.q3_text(r["top"])
    print("q3", seg, (y, m, d), "join rows", r["join_rows"], "oracle groups", n, "ok" if got == want else "MISMATCH", flush=True)
    assert got == want
for pat in ("%green%", "%pink%", "%almond%", "%zzzz%", "%a%"):
    p = pipelines.Q9Pipeline(ctx, t["lineitem"], t["orders"], t["part"], t["partsupp"], t["supplier"], pattern=pat)
    p.time_stages = False
    r = p.run()
    p.free()
    n, rows = O.q9(t, pat)
    want = O.q9_text(rows, n, tpchgen.nation_names())
    got = pipelines.q9_text(r["rows"], tpchgen.nation_names())
    print("q9", pat, "join rows", r["join_rows"], "groups", r["ngroups"], "ok" if got == want else "MISMATCH", flush=True)
    assert got == want
print("soak ok")
