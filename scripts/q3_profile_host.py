"""Host-side profile of the Q3 pipeline (where the non-kernel time of a step goes)."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
from plan_amd import hip, pipelines, tpchgen
sf = (10, 1)
n_ord = tpchgen.orders_count(sf)
L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
Od = tpchgen.orders(sf, columns=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
C = tpchgen.customer(sf)
ctx = hip.Ctx(0)
pipe = pipelines.Q3Pipeline(ctx, L, Od, C)
for _ in range(2):
    pipe.run()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    r = pipe.run()
pr.disable()
print({k: round(v * 1e3, 3) if isinstance(v, float) else v for k, v in r["timings"].items()})
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
