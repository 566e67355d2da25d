"""Does the gated sorted fill depend on where its key column lives? The same 15 M order keys as a DevColumn (ph_dev_alloc: the ctx's pool)
and as a column of a resident table (hipMalloc per column), the same flag bytes; wall time of ph_join_build_where_ex, best of 10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip, tpchgen

ctx = hip.Ctx(0)
O = tpchgen.orders((10, 1), columns=["o_orderkey"])
key = O["o_orderkey"]
n = len(key)
rng = np.random.default_rng(1)
flags = (rng.random(n) < 0.1).astype(np.uint8)
krange = (int(key.min()), int(key.max()))
def best(f, reps=10):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ctx.sync(); ts.append(time.perf_counter() - t0)
    return min(ts)
dkey = hip.DevColumn(ctx, hip.PH_I64, key)
dfl = hip.DevColumn(ctx, hip.PH_CODE8, flags)
tab = hip.Table(ctx, [(hip.PH_I64, key), (hip.PH_CODE8, flags, 0, None, [str(i) for i in range(2)])], n)
tkey, tfl = hip.TableColumn(tab, 0), hip.TableColumn(tab, 1)
one = hip.const(hip.PH_I32, i=1)
for name, k, f in (("pool key, pool flags", dkey, dfl), ("table key, pool flags", tkey, dfl), ("pool key, table flags", dkey, tfl), ("table key, table flags", tkey, tfl)):
    def run():
        j = hip.Join.build_where(ctx, [k], f, hip.PH_EQ, one, None, n, krange, sorted_unique=True)
        j.free()
    run()
    addr = lambda c: int(c.data.value if hasattr(c.data, "value") else c.data)
    print(f"{name}: {best(run)*1e3:.3f} ms  (key at {addr(k) % (2 << 20):#x} mod 2 MiB, flags at {addr(f) % (2 << 20):#x})")
