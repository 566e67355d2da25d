"""Q9 step time with and without per-stage syncs (SF10), and where the host time goes"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from plan_amd import hip, pipelines, tpchgen
sf = (10, 1)
L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"])
Od = tpchgen.orders(sf, columns=["o_orderkey", "o_orderdate"])
P, PS, S = tpchgen.part(sf), tpchgen.partsupp(sf), tpchgen.supplier(sf)
ctx = hip.Ctx(0)
pipe = pipelines.Q9Pipeline(ctx, L, Od, P, PS, S)
for mode in (False, True, False):
    pipe.time_stages = mode
    for _ in range(3):
        pipe.run()
    ctx.sync()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); r = pipe.run(); ctx.sync(); ts.append(time.perf_counter() - t0)
    print(f"time_stages={mode}: min {min(ts)*1e3:.3f} ms  median {sorted(ts)[7]*1e3:.3f} ms", r["timings"] if mode else "")
pipe.time_stages = False
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    pipe.run()
ctx.sync(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
