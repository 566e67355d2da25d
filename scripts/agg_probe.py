"""One aggregate configuration, for rocprofv3: python scripts/agg_probe.py <cardinality> [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

card = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32_000_000
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
vals = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 10**6, n).astype(np.int64))
keys = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, card, n).astype(np.int64))
for _ in range(5):
    agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], card)
    agg.sink([keys], [vals], None, n)
    print(agg.group_count())
    agg.free()
