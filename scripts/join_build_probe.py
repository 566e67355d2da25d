"""join build at 15M keys (and a dense 60M-row probe), for rocprofv3: python scripts/join_build_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip, tpchgen
ctx = hip.Ctx(0)
O = tpchgen.orders((10, 1), columns=["o_orderkey"])
L = tpchgen.lineitem((10, 1), columns=["l_orderkey"])
ok = hip.DevColumn(ctx, hip.PH_I64, O["o_orderkey"])
lk = hip.DevColumn(ctx, hip.PH_I64, L["l_orderkey"])
no, nl = len(O["o_orderkey"]), len(L["l_orderkey"])
for _ in range(5):
    j = hip.Join(ctx, [ok], None, no)
    m, a, b = j.probe_inner([lk], None, nl, nl)
    out = j.lookup([lk], None, nl)
    ctx.free(a); ctx.free(b); ctx.free(out)
    j.free()
ctx.sync()
print("pairs", m)
