"""Micro-benchmarks of the operator-granular kernels (device time via wall clock around
synchronised calls; best of a few runs). Usage: python scripts/bench_ops.py"""
import sys, time
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


n = 32_000_000
vals = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, 10**6, n).astype(np.int64))
for label, card in (("4 groups", 4), ("175 groups", 175), ("65k groups", 65536), ("4M groups", 4_000_000)):
    keys = hip.DevColumn(ctx, hip.PH_I64, rng.integers(0, card, n).astype(np.int64))
    def run():
        agg = hip.Agg(ctx, [hip.PH_I64], [(hip.PH_A_SUM, 0), (hip.PH_A_COUNT_STAR, -1)], card)
        agg.sink([keys], [vals], None, n)
        agg.group_count()
        agg.free()
    t = best(run, 3)
    print(f"agg sink {n/1e6:.0f}M rows, {label}: {t*1e3:.2f} ms  {n/t/1e9:.2f} Grows/s")
    keys.free()

L = tpchgen.lineitem((10, 1), columns=["l_shipdate", "l_orderkey"])
nl = len(L["l_shipdate"])
ship = hip.DevColumn(ctx, hip.PH_DATE, L["l_shipdate"])
k = hip.const(hip.PH_DATE, i=tpchgen.days(1995, 3, 29))
def filt():
    s, c = hip.filter_select(ctx, ship, nl, hip.PH_GT, k)
    ctx.free(s)
t = best(filt)
print(f"filter_select {nl/1e6:.0f}M int32 rows (54% pass): {t*1e3:.3f} ms  {nl*4/t/1e9:.0f} GB/s of column bytes")
k2 = hip.const(hip.PH_DATE, i=tpchgen.days(1998, 11, 1))
def filt2():
    s, c = hip.filter_select(ctx, ship, nl, hip.PH_GT, k2)
    ctx.free(s)
t = best(filt2)
print(f"filter_select {nl/1e6:.0f}M int32 rows (<1% pass): {t*1e3:.3f} ms  {nl*4/t/1e9:.0f} GB/s of column bytes")

O = tpchgen.orders((10, 1), columns=["o_orderkey"])
ok = hip.DevColumn(ctx, hip.PH_I64, O["o_orderkey"])
lk = hip.DevColumn(ctx, hip.PH_I64, L["l_orderkey"])
no = len(O["o_orderkey"])
for frac, label in ((0.1, "10% of orders built"), (1.0, "all orders built")):
    m = int(no * frac)
    sel = ctx.upload(np.sort(rng.choice(no, m, replace=False)).astype(np.int32))
    tb = best(lambda: hip.Join(ctx, [ok], sel, m).free(), 3)
    j = hip.Join(ctx, [ok], sel, m)
    def probe():
        mm, a, b = j.probe_inner([lk], None, nl, nl)
        ctx.free(a); ctx.free(b)
    tp = best(probe, 3)
    print(f"join build {m/1e6:.1f}M keys: {tb*1e3:.3f} ms; probe {nl/1e6:.0f}M rows ({label}): {tp*1e3:.3f} ms  {nl/tp/1e9:.1f} Grows/s")
    j.free(); ctx.free(sel)

# the same 15 M order keys as a DIRECT table (ph_join_build_range with the column's value range): keys in
# storage order take the verified one-pass sorted fill; a random permutation the general passes
krange = (int(O["o_orderkey"].min()), int(O["o_orderkey"].max()))
perm = hip.DevColumn(ctx, hip.PH_I64, rng.permutation(O["o_orderkey"]))
for label, col in (("storage order", ok), ("random order", perm)):
    tb = best(lambda: hip.Join(ctx, [col], None, no, key_range=krange).free(), 3)
    j = hip.Join(ctx, [col], None, no, key_range=krange)
    assert j.kind == "direct"
    def lookup():
        ctx.free(j.lookup([lk], None, nl))
    tl = best(lookup, 3)
    def probe():
        mm, a, b = j.probe_inner([lk], None, nl, nl)
        ctx.free(a); ctx.free(b)
    tp = best(probe, 3)
    print(f"direct table {no/1e6:.1f}M keys ({label}): build {tb*1e3:.3f} ms; lookup of {nl/1e6:.0f}M rows {tl*1e3:.3f} ms; inner probe {tp*1e3:.3f} ms")
    j.free()
