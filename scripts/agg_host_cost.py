"""ph_agg_sink wall time (call + sync), specialised vs generic kernel, at several row counts"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
N = 64_000_000
vals = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 10**6, N).astype(np.int64))
keys = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 175, N).astype(np.int64))
for n in (8_000_000, 16_000_000, 32_000_000, 64_000_000):
    for mode in ("1", "0"):
        os.environ["PH_AGG_JIT"] = mode
        wall = []
        for _ in range(6):
            agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], 175)
            ctx.sync()
            t0 = time.perf_counter()
            agg.sink([keys], [vals], None, n)
            ctx.sync()
            wall.append(time.perf_counter() - t0)
            agg.free()
        print(f"n={n/1e6:.0f}M PH_AGG_JIT={mode}: call+sync {min(wall)*1e6:.0f} us", flush=True)
