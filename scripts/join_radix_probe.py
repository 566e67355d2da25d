"""The radix-partitioned join alone, for rocprofv3: 15 M random 62-bit build keys, 60 M probes (all matching, random order)."""
import os, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
no, nl = 15_000_000, 60_000_000
keys = rng.integers(0, 2**62, no).astype(np.int64)
sp = hip.DevColumn(ctx, hip.PH_I64, keys)
spp = hip.DevColumn(ctx, hip.PH_I64, keys[rng.integers(0, no, nl)])
for _ in range(3):
    t0 = time.perf_counter()
    j = hip.Join(ctx, [sp], None, no)
    ctx.sync(); t1 = time.perf_counter()
    m, a, b = j.probe_inner([spp], None, nl, nl)
    ctx.sync(); t2 = time.perf_counter()
    print(j.kind, m, f"build {(t1-t0)*1e3:.3f} ms probe {(t2-t1)*1e3:.3f} ms")
    ctx.free(a); ctx.free(b); j.free()
