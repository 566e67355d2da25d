"""ph_sort_rows timing: python scripts/bench_sort.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip
ctx = hip.Ctx(0)
rng = np.random.default_rng(1)
for n in (113_000, 10_000_000):
    rev = hip.DevColumn(ctx, hip.PH_DEC64, rng.integers(0, 5 * 10**9, n).astype(np.int64), 4)
    date = hip.DevColumn(ctx, hip.PH_DATE, rng.integers(8000, 10500, n).astype(np.int32))
    for keys, desc, label in (([rev, date], [True, False], "revenue desc, date"), ([date], [False], "date")):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); out = hip.sort_rows(ctx, keys, desc, None, n); ctx.sync(); best = min(best, time.perf_counter() - t0); ctx.free(out)
        print(f"sort {n} rows by ({label}): {best*1e3:.3f} ms  {n/best/1e9:.2f} G rows/s")
    rev.free(); date.free()
