"""Device-memory stability of the resident plans (ph_plan): every TPC-H plan run + fetched repeatedly at SF1; free memory after
5 and after 45 runs of each, and the result text of the last run against the first (python scripts/leak_check_plans.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tpch_data
from plan_amd import hip, tpch

ctx = hip.Ctx(0)
data = tpch_data.load(1, 1)
db = tpch.Database(ctx, data)


def run(p, name=""):
    p.run()
    r = p.fetch()
    if name == "q22":   # the key of a computed VARCHAR is the row of its representative in the plan's relation — whichever equal string interned first
        return tpch.q22_text(db, p, r)
    return (r["ngroups"], tuple(map(tuple, r["sum"][:4])), tuple(map(tuple, r["keys"][:4])))


def q22():
    s = tpch.q22_scalar_plan(db)
    s.run()
    thr = tpch.q22_threshold(s.fetch())
    s.free()
    return tpch.q22_plan(db, thr)


plans = {"q3": lambda: tpch.q3_plan(db), "q9": lambda: tpch.q9_plan(db), "q15": lambda: tpch.q15_plan(db), "q17": lambda: tpch.q17_plan(db),
         "q18": lambda: tpch.q18_plan(db), "q20": lambda: tpch.q20_plans(db)[0], "q21": lambda: tpch.q21_plan(db), "q22": q22}
bad = 0
for name, make in plans.items():
    p = make()
    first = run(p, name)
    for _ in range(4):
        run(p, name)
    ctx.sync()
    f0 = torch.cuda.mem_get_info()[0]
    last = None
    for _ in range(40):
        last = run(p, name)
    ctx.sync()
    f1 = torch.cuda.mem_get_info()[0]
    same = last == first
    bad += (not same) or (f0 - f1) > (64 << 20)
    print(f"{name}: free MiB after 5 runs {f0 >> 20}, after 45 runs {f1 >> 20}, delta {(f0 - f1) >> 20}; results stable: {same}", flush=True)
    p.free()
db.free()
print("leak check done:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
