"""Q9-shaped aggregate sink (3.27 M rows, two int32 keys, 175 groups, one positional DEC64 sum) under
different workgroups-per-CU settings: python scripts/agg_small_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
for n in (3_269_208, 32_000_000):
    k0 = hip.DevColumn(ctx, hip.PH_I32, rng.integers(0, 25, n).astype(np.int32))
    k1 = hip.DevColumn(ctx, hip.PH_I32, rng.integers(1992, 1999, n).astype(np.int32))
    v = hip.DevColumn(ctx, hip.PH_DEC64, rng.integers(0, 10**8, n).astype(np.int64), 4)
    for occ in ("256", "512", "1024"):
        os.environ["PH_AGG_T"] = occ
        best = 1e9
        for rep in range(6):
            agg = hip.Agg(ctx, [hip.PH_I32, hip.PH_I32], [(hip.PH_A_SUM, 0)], 1024)
            ctx.sync(); t0 = time.perf_counter()
            agg.sink([k0, k1], [v], None, n, positional=True)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
            agg.free()
        print(f"n={n} threads/workgroup={occ}: {best*1e6:.0f} us", flush=True)
