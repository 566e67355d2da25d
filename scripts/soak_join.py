"""Soak of the ordered-input join forms against the general ones: gated sorted fill vs filter + build, merge lookup
vs table lookup, LDS-staged mark vs plain mark, on random sizes / densities (python scripts/soak_join.py [rounds] [seed])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plan_amd import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ctx = hip.Ctx(0)
ctx.set_deferred_errors(True)
for it in range(rounds):
    wide = rng.random() < 0.5
    dt, typ = (np.int64, hip.PH_I64) if wide else (np.int32, hip.PH_I32)
    n = int(rng.choice([5_000, 90_000, 300_000, 1_400_000]))
    slots_per_row = float(rng.choice([1.0, 1.7, 4.0, 7.5]))
    span = int(n * slots_per_row)
    lo = int(rng.integers(-1000, 1000))
    keys = (np.sort(rng.choice(span, n, replace=False)) + lo).astype(dt)
    rngk = (lo, lo + span - 1)
    dens = float(rng.choice([0.0, 0.02, 0.3, 1.0]))
    flag = (rng.random(n) < dens).astype(np.uint8)
    npq = int(rng.choice([3_000, 200_000, 1_300_000]))
    probes = rng.integers(lo - 20, lo + span + 20, npq).astype(dt)
    if rng.random() < 0.6:
        probes.sort()
    dk, dfl, dp = hip.DevColumn(ctx, typ, keys), hip.DevColumn(ctx, hip.PH_CODE8, flag), hip.DevColumn(ctx, typ, probes)
    one = hip.const(hip.PH_I32, i=1)
    jw = hip.Join.build_where(ctx, [dk], dfl, hip.PH_EQ, one, None, n, rngk, sorted_unique=True)
    fs, fn = hip.filter_select(ctx, dfl, n, hip.PH_EQ, one)
    jf = hip.Join(ctx, [dk], fs, fn, key_range=rngk)
    ok = jw is not None and jw.count() == jf.count() == int(flag.sum())
    a = ctx.download(jw.lookup([dp], None, npq), np.int32, npq)
    b = ctx.download(jf.lookup([dp], None, npq), np.int32, npq)
    ok = ok and np.array_equal(a, b)
    mw = jw.probe_inner([dp], None, npq, npq); mf = jf.probe_inner([dp], None, npq, npq)
    ok = ok and mw[0] == mf[0] and np.array_equal(ctx.download(mw[2], np.int32, mw[0]), ctx.download(mf[2], np.int32, mf[0]))
    # mark with a filter (LDS-staged for big probe sides) against numpy
    wcol = rng.integers(0, 100, npq).astype(np.int32)
    dw = hip.DevColumn(ctx, hip.PH_I32, wcol)
    f = jw.probe_mark_where([dp], dw, hip.PH_LT, hip.const(hip.PH_I32, i=50), npq)
    if f is not None:
        got = ctx.download(f, np.uint8, npq)
        ok = ok and np.array_equal(got != 0, np.isin(probes, keys[flag == 1]) & (wcol < 50))
    # merge lookup over the whole key column (needs ordered probes)
    sp = np.sort(probes)
    dsp = hip.DevColumn(ctx, typ, sp)
    ju = hip.Join(ctx, [dk], None, n, key_range=rngk)
    ml = ctx.download(hip.merge_lookup(ctx, dk, n, dsp, None, npq), np.int32, npq)
    tl = ctx.download(ju.lookup([dsp], None, npq), np.int32, npq)
    ok = ok and np.array_equal(ml, tl)
    ctx.check_deferred()
    print(f"round {it}: {'i64' if wide else 'i32'} n {n} span {span} density {dens} probes {npq} marked {'lds/plain' if f is not None else 'n/a'} {'ok' if ok else 'MISMATCH'}", flush=True)
    assert ok
    for j in (jw, jf, ju):
        j.free()
    for c in (dk, dfl, dp, dw, dsp):
        c.free()
print("soak ok")
