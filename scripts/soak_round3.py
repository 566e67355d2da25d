"""Randomised soak of the round-3 forms against numpy: the bulk aggregate's second form (n >= 4 M rows: key width, NULLs, group counts
around the hint, several aggregates) and the radix join (build / probe sizes, duplicate rates, NULL keys, selections, probe kernels).
python scripts/soak_round3.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PH_JOIN_RADIX_MIN", "200000")
os.environ.setdefault("PH_JOIN_AUTO_RANGE", "0")
import numpy as np
from plan_amd import hip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = hip.Ctx(0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    rng = np.random.default_rng(1000 + it)
    it += 1
    if it % 2:   # ---- aggregate
        n = int(rng.integers(4_200_000, 6_500_000))
        card = int(rng.choice([3000, 20_000, 65_536, 200_000]))
        hint = int(card * rng.choice([0.5, 1.0, 1.5]))
        wide = bool(rng.integers(0, 2))
        k = rng.integers(0, card, n).astype(np.int64 if wide else np.int32)
        v = rng.integers(-10**6, 10**6, n).astype(np.int64)
        nulls = bool(rng.integers(0, 2))
        val = None
        if nulls:
            bits = rng.random(n) < 0.9
            val = np.packbits(bits, bitorder="little")
        agg = hip.Agg(ctx, [hip.PH_I64 if wide else hip.PH_I32], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT, 0), (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0), (hip.PH_A_COUNT_STAR, -1)], hint)
        dk = hip.DevColumn(ctx, hip.PH_I64 if wide else hip.PH_I32, k)
        dv = hip.DevColumn(ctx, hip.PH_I64, v, validity=val)
        agg.sink([dk], [dv], None, n)
        r = agg.finalize(python_ints=False)
        uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
        order = np.argsort(first)
        assert r["ngroups"] == len(uk), (it, r["ngroups"], len(uk))
        assert np.array_equal(r["keys"][:, 0], uk[order]) and np.array_equal(r["first_row"], first[order]), it
        ok = bits if nulls else np.ones(n, bool)
        sums = np.zeros(len(uk), np.int64); np.add.at(sums, inv[ok], v[ok])
        cnt = np.bincount(inv[ok], minlength=len(uk))
        mn = np.full(len(uk), np.iinfo(np.int64).max); np.minimum.at(mn, inv[ok], v[ok])
        mx = np.full(len(uk), np.iinfo(np.int64).min); np.maximum.at(mx, inv[ok], v[ok])
        assert np.array_equal(r["sum_lo"][:, 0].astype(np.int64), sums[order]), it
        assert np.array_equal(r["count"][:, 1], cnt[order]) and np.array_equal(r["count"][:, 4], np.bincount(inv)[order]), it
        has = cnt[order] > 0
        assert np.array_equal(r["sum_lo"][:, 2].astype(np.int64)[has], mn[order][has]) and np.array_equal(r["sum_lo"][:, 3].astype(np.int64)[has], mx[order][has]), it
        agg.free(); dk.free(); dv.free()
        print(f"iter {it}: agg n={n} card={card} hint={hint} wide={wide} nulls={nulls} ok", flush=True)
    else:        # ---- radix join
        nb = int(rng.integers(250_000, 1_500_000))
        npr = int(rng.integers(50_000, 3_000_000))
        dup = float(rng.choice([0.0, 0.05]))
        b = rng.integers(0, 2**61, nb).astype(np.int64)
        if dup:
            idx = rng.integers(0, nb, int(nb * dup))
            b[idx] = b[(idx + 1) % nb]
        p = np.concatenate([b[rng.integers(0, nb, npr // 2)], rng.integers(0, 2**61, npr - npr // 2).astype(np.int64)])
        rng.shuffle(p)
        os.environ["PH_JOIN_RADIX_PART_MIN"] = str(int(rng.choice([1, 1 << 40])))
        db, dp = hip.DevColumn(ctx, hip.PH_I64, b), hip.DevColumn(ctx, hip.PH_I64, p)
        j = hip.Join(ctx, [db], None, nb)
        assert j.kind == "radix", j.kind
        cap = npr * 3
        m, pr, br = j.probe_inner([dp], None, npr, cap)
        got = np.stack([ctx.download(pr, np.int32, m), ctx.download(br, np.int32, m)], 1).astype(np.int64)
        order = np.argsort(b, kind="stable")
        sb = b[order]
        lo, hi = np.searchsorted(sb, p, "left"), np.searchsorted(sb, p, "right")
        want_m = int((hi - lo).sum())
        assert m == want_m, (it, m, want_m)
        rows = np.repeat(np.arange(npr), hi - lo)
        starts = np.repeat(lo, hi - lo)
        offs = np.arange(want_m) - np.repeat(np.cumsum(hi - lo) - (hi - lo), hi - lo)
        want = np.stack([rows, order[starts + offs]], 1)
        g = got[np.lexsort((got[:, 1], got[:, 0]))]
        w = want[np.lexsort((want[:, 1], want[:, 0]))]
        assert np.array_equal(g, w), it
        ctx.free(pr); ctx.free(br); j.free(); db.free(); dp.free()
        print(f"iter {it}: join nb={nb} np={npr} dup={dup} part_min={os.environ['PH_JOIN_RADIX_PART_MIN']} pairs={m} ok", flush=True)
print("soak done:", it, "iterations")
