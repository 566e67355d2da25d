"""A/B of the two-level bulk aggregate against the one-level forms: python scripts/agg_two_level_ab.py (set PH_AGG_BULK_ONE_LEVEL=1 for B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

n = 32_000_000
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
vals = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 10**6, n).astype(np.int64))
for card in ([int(x) for x in sys.argv[1:]] or (200_000, 500_000, 1_000_000, 2_000_000, 4_000_000)):
    keys = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, card, n).astype(np.int64))
    best = 1e9
    for it in range(12):
        agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], card)
        ctx.sync()
        t0 = time.perf_counter()
        agg.sink([keys], [vals], None, n)
        ctx.sync()
        dt = time.perf_counter() - t0
        if it: best = min(best, dt)
        g = agg.group_count()
        agg.free()
    print(f"card {card:>8} groups {g:>8} best {best*1e3:.3f} ms", flush=True)
    keys.free()
