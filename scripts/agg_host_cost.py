"""host-side cost of ph_agg_sink (no sync) and wall time with sync, specialised vs generic kernel"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
n = 32_000_000
vals = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 10**6, n).astype(np.int64))
keys = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 175, n).astype(np.int64))
for mode in ("1", "0", "1", "0"):
    os.environ["PH_AGG_JIT"] = mode
    host, wall = [], []
    for _ in range(8):
        agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 175)
        ctx.sync()
        t0 = time.perf_counter()
        agg.sink([keys], [vals], None, n)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        host.append(t1 - t0); wall.append(t2 - t0)
        agg.free()
    print(f"PH_AGG_JIT={mode}: host call {min(host)*1e6:.0f} us (median {sorted(host)[4]*1e6:.0f}), call+sync {min(wall)*1e6:.0f} us")
