"""host time between the last download of one Q9 run and the first launch of the next:
python scripts/q9_gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plan_amd import hip, pipelines, tpchgen
torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = hip.Ctx(0, stream=st.cuda_stream)
sf = (10, 1)
L = tpchgen.lineitem(sf, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"])
Od = tpchgen.orders(sf, columns=["o_orderkey", "o_orderdate"])
pipe = pipelines.Q9Pipeline(ctx, L, Od, tpchgen.part(sf), tpchgen.partsupp(sf), tpchgen.supplier(sf))
pipe.time_stages = False
marks = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((tag, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)
wrap(hip.Agg, "finalize", "finalize")
wrap(hip, "filter_like", "like") if hasattr(hip, "filter_like") else None
wrap(hip.Ctx, "wait_counts", "wait_counts")
from plan_amd import dist
for obj, name in ((hip.Agg, "free"), (hip.Ctx, "free_many"), (hip.Ctx, "set_async_counts"), (hip.Ctx, "set_deferred_errors"),
                  (dist, "merge_group_partials"), (hip, "filter_select"), (hip.Agg, "__init__"), (hip.Agg, "sink")):
    wrap(obj, name, "x:" + name)
for _ in range(5):
    pipe.run()
marks.clear()
ctx.sync(); T0 = time.perf_counter()
ends = []
for _ in range(20):
    t0 = time.perf_counter(); pipe.run(); ends.append((t0, time.perf_counter()))
ctx.sync(); print(f"{(time.perf_counter() - T0) / 20 * 1e3:.3f} ms per step")
fin = [m for m in marks if m[0] == "finalize"]
for i in range(3, 8):
    t0, t1 = ends[i]
    f = fin[i]
    w = [m for m in marks if m[0] == "wait_counts" and t0 <= m[1] <= t1]
    print(f"run {i}: total {1e6*(t1-t0):.0f} us | until finalize call {1e6*(f[1]-t0):.0f} | in finalize {1e6*(f[2]-f[1]):.0f} | after finalize {1e6*(t1-f[2]):.0f} | "
          + " ".join(f"wait@{1e6*(m[1]-t0):.0f}+{1e6*(m[2]-m[1]):.0f}" for m in w))
    if i == 5:
        for m in marks:
            if m[0].startswith("x:") and f[1] - 60e-6 <= m[1] <= t1 + 60e-6:
                print(f"     {m[0][2:]:24s} @{1e6*(m[1]-f[2]):8.1f} us after finalize, took {1e6*(m[2]-m[1]):6.1f}")
