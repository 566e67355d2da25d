"""String-filter timing probe for rocprofv3: python scripts/like_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip, tpchgen

ctx = hip.Ctx(0)
P = tpchgen.part((10, 1))
n = len(P["p_name_off"]) - 1
name = hip.DevColumn(ctx, hip.PH_STR, P["p_name_off"], aux=P["p_name_bytes"])
print("rows", n, "bytes", len(P["p_name_bytes"]))
for op, pat in [(hip.PH_LIKE, "%green%"), (hip.PH_LIKE, "zz%"), (hip.PH_EQ, "x"), (hip.PH_LIKE, "%gr_en%")]:
    for _ in range(3):
        s, c = hip.filter_select(ctx, name, n, op, hip.const(hip.PH_STR, s=pat))
        ctx.free(s)
    print(pat, c)
