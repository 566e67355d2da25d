"""Device-memory stability of the join pipelines: free memory after 20 and after 220 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plan_amd import hip, pipelines, tpchgen

ctx = hip.Ctx(0)
sf = (2, 1)
n_ord = tpchgen.orders_count(sf)
L = tpchgen.lineitem(sf, 0, n_ord)
Od = tpchgen.orders(sf, 0, n_ord)
C = tpchgen.customer(sf, 0, n_ord // 10)
P, PS, S = tpchgen.part(sf), tpchgen.partsupp(sf), tpchgen.supplier(sf)
for name, pipe in (("q3", pipelines.Q3Pipeline(ctx, L, Od, C)), ("q9", pipelines.Q9Pipeline(ctx, L, Od, P, PS, S))):
    pipe.time_stages = False
    for _ in range(20):
        pipe.run()
    ctx.sync()
    f0 = torch.cuda.mem_get_info()[0]
    for _ in range(200):
        pipe.run()
    ctx.sync()
    f1 = torch.cuda.mem_get_info()[0]
    print(name, "free MiB after 20 steps", f0 >> 20, "after 220 steps", f1 >> 20, "delta MiB", (f0 - f1) >> 20)
    pipe.free()
