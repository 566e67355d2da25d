"""host timeline of a Q3 run (where the GPU waits for the host): python scripts/q3_gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plan_amd import hip, pipelines, tpchgen, dist
torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = hip.Ctx(0, stream=st.cuda_stream)
sf = (10, 1)
L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
Od = tpchgen.orders(sf, columns=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
pipe = pipelines.Q3Pipeline(ctx, L, Od, tpchgen.customer(sf))
pipe.time_stages = False
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((name, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)
for obj, name in ((hip.Agg, "topk"), (hip.Agg, "free"), (hip.Ctx, "free_many"), (hip.Join, "probe_inner_where"), (hip.Join, "probe_mark_where"),
                  (hip.Join, "free"), (hip, "expr_eval"), (hip, "gather"), (hip, "gather_multi"), (hip.Agg, "__init__"), (hip.Agg, "sink_sorted"),
                  (dist, "merge_topk"), (hip.Ctx, "set_deferred_errors")):
    wrap(obj, name)
_bw = hip.Join.build_where.__func__
def bw(cls, *a, **k):
    t0 = time.perf_counter(); r = _bw(cls, *a, **k); marks.append(("build_where", t0, time.perf_counter())); return r
hip.Join.build_where = classmethod(bw)
for _ in range(5):
    pipe.run()
marks.clear()
ctx.sync(); T0 = time.perf_counter()
ends = []
for _ in range(20):
    t0 = time.perf_counter(); pipe.run(); ends.append((t0, time.perf_counter()))
ctx.sync(); print(f"{(time.perf_counter() - T0) / 20 * 1e3:.3f} ms per step")
t0, t1 = ends[6]
print(f"run 6: {1e6*(t1-t0):.0f} us")
for m in marks:
    if t0 - 5e-6 <= m[1] <= t1 + 30e-6:
        print(f"   {m[0]:22s} @{1e6*(m[1]-t0):7.1f} us, took {1e6*(m[2]-m[1]):6.1f}")
