"""where the host time of a pipeline step goes: python scripts/host_overhead.py q3|q9"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plan_amd import hip, pipelines, tpchgen
q = sys.argv[1] if len(sys.argv) > 1 else "q9"
torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = hip.Ctx(0, stream=st.cuda_stream)
sf = (10, 1)
if q == "q9":
    L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"])
    Od = tpchgen.orders(sf, columns=["o_orderkey", "o_orderdate"])
    pipe = pipelines.Q9Pipeline(ctx, L, Od, tpchgen.part(sf), tpchgen.partsupp(sf), tpchgen.supplier(sf))
else:
    L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    Od = tpchgen.orders(sf, columns=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    pipe = pipelines.Q3Pipeline(ctx, L, Od, tpchgen.customer(sf))
pipe.time_stages = False
for _ in range(5):
    pipe.run()
ctx.sync(); t0 = time.perf_counter()
for _ in range(30):
    pipe.run()
ctx.sync(); print(f"{q}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    pipe.run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
