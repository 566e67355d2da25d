"""Soak of the call-specialised aggregate sink against the generic kernel: random shapes at >= 2^20 rows
(python scripts/soak_sink.py [rounds] [seed])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx = hip.Ctx(0)
n = (1 << 20) + 777
POOL = [(hip.PH_A_SUM, 0), (hip.PH_A_SUM, 1), (hip.PH_A_AVG, 0), (hip.PH_A_COUNT, 0), (hip.PH_A_COUNT, 1), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 1),
        (hip.PH_A_COUNT_STAR, -1)]
for it in range(rounds):
    card = int(rng.choice([1, 7, 300, 900, 2500, 5000, 200_000]))
    nk = int(rng.integers(1, 3))
    nullable_k, nullable_a = rng.random() < 0.4, rng.random() < 0.5
    use_sel = rng.random() < 0.4
    hint = int(rng.choice([16, card, 4 * card + 40_000]))
    k0 = rng.integers(0, card, n).astype(np.int64)
    k1 = rng.integers(0, 2, n).astype(np.int32)
    v0 = rng.integers(-10**12, 10**12, n).astype(np.int64)      # some values beyond the 2^40 bound of the LDS partials
    v1 = rng.integers(0, 1000, n).astype(np.int32)
    kv = np.packbits(rng.random(n) > 0.04, bitorder="little") if nullable_k else None
    av = np.packbits(rng.random(n) > 0.1, bitorder="little") if nullable_a else None
    cols = [hip.DevColumn(ctx, hip.PH_I64, k0, validity=kv), hip.DevColumn(ctx, hip.PH_I32, k1)][:nk]
    types = [hip.PH_I64, hip.PH_I32][:nk]
    args = [hip.DevColumn(ctx, hip.PH_DEC64, v0, 2, validity=av), hip.DevColumn(ctx, hip.PH_I32, v1)]
    aggs = [POOL[i] for i in rng.choice(len(POOL), int(rng.integers(1, 6)), replace=False)]
    rows = np.sort(rng.choice(n, n - 5000, replace=False)).astype(np.int32) if use_sel else None
    dsel = ctx.upload(rows) if rows is not None else None
    m = len(rows) if rows is not None else n
    res = {}
    for mode in ("1", "0"):
        os.environ["PH_AGG_JIT"] = mode
        agg = hip.Agg(ctx, types, aggs, hint)
        agg.sink(cols, args, dsel, m)
        r = agg.finalize(python_ints=False)
        order = np.lexsort(tuple(r["keys"][:, c] for c in range(nk - 1, -1, -1)) + (r["key_null"][:, 0],))
        res[mode] = {k: np.asarray(r[k])[order] for k in ("first_row", "keys", "key_null", "sum_lo", "sum_hi", "count")}
        agg.free()
    os.environ.pop("PH_AGG_JIT")
    ok = all(np.array_equal(res["1"][k], res["0"][k]) for k in res["1"])
    print(f"round {it}: card {card} nk {nk} nullk {nullable_k} nulla {nullable_a} sel {use_sel} hint {hint} aggs {[a[0] for a in aggs]} groups {len(res['1']['keys'])} {'ok' if ok else 'MISMATCH'}", flush=True)
    assert ok
    for c in cols + args:
        c.free()
    if dsel is not None:
        ctx.free(dsel)
print("soak ok")
