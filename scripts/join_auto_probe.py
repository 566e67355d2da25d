"""ph_join_build without a range: with and without the automatic key-range pass (PH_JOIN_AUTO_RANGE), build + probe wall time."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from plan_amd import hip, tpchgen

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ctx.sync(); ts.append(time.perf_counter() - t0)
    return min(ts)
L = tpchgen.lineitem((10, 1), columns=["l_orderkey"])
O = tpchgen.orders((10, 1), columns=["o_orderkey"])
nl, no = len(L["l_orderkey"]), len(O["o_orderkey"])
ok = hip.DevColumn(ctx, hip.PH_I64, O["o_orderkey"])
lk = hip.DevColumn(ctx, hip.PH_I64, L["l_orderkey"])
perm = hip.DevColumn(ctx, hip.PH_I64, rng.permutation(O["o_orderkey"]))
sparse_keys = rng.integers(0, 2**62, no).astype(np.int64)
sp = hip.DevColumn(ctx, hip.PH_I64, sparse_keys)
spp = hip.DevColumn(ctx, hip.PH_I64, sparse_keys[rng.integers(0, no, nl)])
for auto in ("0", "1"):
    os.environ["PH_JOIN_AUTO_RANGE"] = auto
    for label, col, sel, m, pk in (("orders keys, storage order", ok, None, no, lk), ("orders keys, shuffled", perm, None, no, lk),
                                   ("10% of the orders (selection)", ok, ctx.upload(np.sort(rng.choice(no, no // 10, replace=False)).astype(np.int32)), no // 10, lk),
                                   ("random 62-bit keys, random probes", sp, None, no, spp)):
        tb = best(lambda: hip.Join(ctx, [col], sel, m).free(), 3)
        j = hip.Join(ctx, [col], sel, m)
        def probe():
            mm, a, b = j.probe_inner([pk], None, nl, nl)
            ctx.free(a); ctx.free(b)
        tp = best(probe, 3)
        print(f"auto={auto} {label}: {j.kind}: build {tb*1e3:.3f} ms, inner probe of {nl/1e6:.0f}M rows {tp*1e3:.3f} ms, total {(tb+tp)*1e3:.3f} ms")
        j.free()
