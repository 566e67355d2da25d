"""gather_multi variants on the Q9 shape: 5 lineitem-sized columns at 5.4 % ascending row ids"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
n = 59_986_052
rng = np.random.default_rng(5)
ids = np.flatnonzero(rng.random(n) < 0.0545).astype(np.int32)
m = len(ids)
cols = [hip.DevColumn(ctx, hip.PH_I32, np.zeros(n, np.int32)), hip.DevColumn(ctx, hip.PH_I64, np.zeros(n, np.int64)),
        hip.DevColumn(ctx, hip.PH_DEC64, np.zeros(n, np.int64), 2), hip.DevColumn(ctx, hip.PH_DEC64, np.zeros(n, np.int64), 2),
        hip.DevColumn(ctx, hip.PH_I32, np.zeros(n, np.int32))]
d = ctx.upload(ids)
for u in (1, 2, 4, 8):
    for g in (4, 8, 16, 64):
        os.environ["PH_GM_U"], os.environ["PH_GM_GRID"] = str(u), str(g)
        for rep in range(2):
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(10):
                outs = hip.gather_multi(ctx, cols, d, m)
                for o in outs: ctx.free(o)
            ctx.sync(); dt = (time.perf_counter() - t0) / 10
        print(f"U={u} grid=256*{g}: {dt*1e6:.0f} us for {m} rows x 5 columns", flush=True)
