import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from plan_amd import hip, tpchgen
stage = int(sys.argv[1])
if stage == 8:
    import torch
ctx = hip.Ctx(0)
import tpch_data
t = tpch_data.load(1, 10, q9=False)
L, O, C = t["lineitem"], t["orders"], t["customer"]
if stage >= 1:
    D = hip.DevColumn
    c_key = D(ctx, hip.PH_I32, C["c_custkey"]); c_seg = D(ctx, hip.PH_CODE8, C["c_mktsegment"])
    o_key = D(ctx, hip.PH_I64, O["o_orderkey"]); o_cust = D(ctx, hip.PH_I32, O["o_custkey"]); o_date = D(ctx, hip.PH_DATE, O["o_orderdate"])
    l_key = D(ctx, hip.PH_I64, L["l_orderkey"]); l_ship = D(ctx, hip.PH_DATE, L["l_shipdate"])
    nc, no, nl = len(C["c_custkey"]), len(O["o_orderkey"]), len(L["l_orderkey"])
    date = tpchgen.days(1995, 3, 29)
if stage >= 2:
    cs, cn = hip.filter_select(ctx, c_seg, nc, hip.PH_EQ, hip.const(hip.PH_I32, i=3))
    print("filter", cn)
if stage >= 3:
    ck = hip.gather(ctx, c_key, cs, cn)
    c = hip.Col(); c.type, c.data = hip.PH_I32, ck
    j1 = hip.Join(ctx, [c], None, cn)
    print("build", j1.count())
if stage >= 4:
    fused = j1.probe_inner_where([o_cust], o_date, hip.PH_LT, hip.const(hip.PH_DATE, i=date), None, no, no)
    print("probe_where", fused[0] if fused else None)
if stage >= 5:
    m1, orow, _c = fused
    j2 = hip.Join(ctx, [o_key], orow, m1)
    print("build2", j2.count())
if stage >= 6:
    f2 = j2.probe_inner_where([l_key], l_ship, hip.PH_GT, hip.const(hip.PH_DATE, i=date), None, nl, nl)
    print("probe2", f2[0])
if stage == 7:
    import torch
if stage == 9:
    import torch.distributed
ctx.close()
print("closed ok", flush=True)
