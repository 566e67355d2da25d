import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
rng = np.random.default_rng(21)
n, card = 300_000, 5_000
keys = rng.integers(0, card, n).astype(np.int32)
vals = rng.integers(-50_000, 200_000, n).astype(np.int64)
agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], card)
agg.sink([hip.DevColumn(ctx, hip.PH_I32, keys)], [hip.DevColumn(ctx, hip.PH_DEC64, vals, scale=2)], None, n)
full = agg.finalize()
sums = np.array([s[0] for s in full["sum"]], np.int64); cnts = full["count"][:, 1]
print("groups", full["ngroups"], "sum>4.5M:", int((sums > 4_500_000).sum()), "cnt>=55:", int((cnts >= 55).sum()))
for w in ([(0, hip.PH_GT, hip.const(hip.PH_DEC64, i=4_500_000, scale=2), 2)], [(1, hip.PH_GE, hip.const(hip.PH_I32, i=55), 0)],
          [(1, hip.PH_GE, hip.const(hip.PH_DEC64, i=55, scale=0), 0)], [(0, hip.PH_GT, hip.const(hip.PH_F32, f=45000.37), 2)]):
    try:
        print(w[0][0], w[0][2].type, agg.finalize(where=w)["ngroups"])
    except Exception as e:
        print("ERR", e)
