"""Gated sorted fill against the plain one, for rocprofv3: python scripts/gated_fill_probe.py [density ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

ctx = hip.Ctx(0)
ctx.set_deferred_errors(True)
n = 15_000_000
i = np.arange(n, dtype=np.int64)
keys = (i // 8) * 32 + i % 8 + 1          # the o_orderkey pattern: 8 of every 32 values
rngk = (int(keys[0]), int(keys[-1]))
dk = hip.DevColumn(ctx, hip.PH_I64, keys)
rng = np.random.default_rng(0)
for dens in [float(x) for x in sys.argv[1:]] or [0.1, 1.0]:
    flag = (rng.random(n) < dens).astype(np.uint8)
    dfl = hip.DevColumn(ctx, hip.PH_CODE8, flag)
    for _ in range(5):
        j = hip.Join.build_where(ctx, [dk], dfl, hip.PH_EQ, hip.const(hip.PH_I32, i=1), None, n, rngk, sorted_unique=True)
        j.free()
    dfl.free()
for _ in range(5):
    j = hip.Join(ctx, [dk], None, n, key_range=rngk, sorted_unique=True)
    j.free()
ctx.sync()
ctx.check_deferred()
print("ok")
