"""Streaming aggregate (ph_agg_sink_sorted) at Q18's subquery shape: 60 M rows ordered by an 8-byte key, ~4 rows per group, SUM of a 4-byte argument.
usage: python scripts/bench_stream_agg.py [rows]"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plan_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
rng = np.random.default_rng(1)
lens = rng.integers(1, 8, n // 4 + 8)
keys = np.repeat(np.arange(len(lens), dtype=np.int64) * 4, lens)[:n]
vals = rng.integers(1, 51, n).astype(np.int32)
ctx = hip.Ctx(0)
K, V = hip.DevColumn(ctx, hip.PH_I64, keys), hip.DevColumn(ctx, hip.PH_I32, vals)
ng = len(np.unique(keys))
for rep in range(6):
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0)], ng)
    ctx.sync()
    t0 = time.perf_counter()
    agg.sink_sorted([K], [V], n)
    ctx.sync()
    dt = time.perf_counter() - t0
    got = agg.group_count()
    if rep == 5:
        print(f"{n} rows -> {got} groups: {dt * 1e3:.3f} ms, {n * 12 / dt / 1e12:.2f} TB/s of key + argument bytes")
    assert got == ng
    agg.free()
