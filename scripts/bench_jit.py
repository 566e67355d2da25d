"""Generated (hiprtc) fused scan kernels against the precompiled ones and against the HBM roofline,
SF10 lineitem: python scripts/bench_jit.py. GB/s = bytes of the columns the plan reads / event time."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from plan_amd import hip, queries, tpchgen

torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = hip.Ctx(0, stream=stream.cuda_stream)
L = tpchgen.lineitem((10, 1), columns=["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
n = len(L["l_shipdate"])
t = queries.lineitem_table(ctx, L)
Q, E, D, T, RF, LS, SD = range(7)
W = {Q: 4, E: 8, D: 8, T: 8, RF: 1, LS: 1, SD: 4}


def timeit(plan, reps=20):
    for _ in range(3):
        plan.run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream); plan.run(); b.record(stream)
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(d) / len(d)


def report(name, plan, cols):
    ms = timeit(plan)
    b = n * sum(W[c] for c in cols)
    print(f"{name:58s} {plan.kind:14s} {ms:7.3f} ms  {b / ms / 1e6:7.0f} GB/s  {b / ms / 1e6 / 8000:5.1%} of 8 TB/s", flush=True)
    plan.free()


e, d, tx, q = hip.X_COL(E), hip.X_COL(D), hip.X_COL(T), hip.X_COL(Q)
one = hip.X_CONST(1, 0)
for mode in ("0", "1"):
    os.environ["PH_SCAN_JIT"] = mode
    report(f"Q1 (PH_SCAN_JIT={mode})", queries.q1_plan(ctx, t), [Q, E, D, T, RF, LS, SD])
    report(f"Q6 (PH_SCAN_JIT={mode})", queries.q6_plan(ctx, t), [SD, D, Q, E])
os.environ.pop("PH_SCAN_JIT")
cut = hip.pred(SD, hip.PH_LE, hip.const(hip.PH_DATE, i=queries.q1_shipdate_cutoff()))
report("group by linestatus: sum(ext)", hip.ScanPlan(ctx, t, [], [LS], [hip.aggexpr(hip.PH_A_SUM, [e])]), [LS, E])
report("group by flag,status: sum(ext), count(*) where shipdate<=", hip.ScanPlan(ctx, t, [cut], [RF, LS], [hip.aggexpr(hip.PH_A_SUM, [e]), hip.aggexpr(hip.PH_A_COUNT_STAR)]), [RF, LS, E, SD])
report("group by flag,status: min/max/sum(ext*(1-disc))", hip.ScanPlan(ctx, t, [], [RF, LS], [hip.aggexpr(hip.PH_A_MIN, [e]), hip.aggexpr(hip.PH_A_MAX, [e]), hip.aggexpr(hip.PH_A_SUM, [e, one, d, hip.X_SUB, hip.X_MUL])]), [RF, LS, E, D])
report("ungrouped: sum(ext*disc*tax), sum(qty) where qty != 7", hip.ScanPlan(ctx, t, [hip.pred(Q, hip.PH_NE, hip.const(hip.PH_I32, i=7))], [], [hip.aggexpr(hip.PH_A_SUM, [e, d, hip.X_MUL, tx, hip.X_MUL]), hip.aggexpr(hip.PH_A_SUM, [q])]), [Q, E, D, T])
report("group by flag: 8 accumulators (Q1's) ", hip.ScanPlan(ctx, t, [cut], [RF], [hip.aggexpr(hip.PH_A_SUM, [q]), hip.aggexpr(hip.PH_A_SUM, [e]), hip.aggexpr(hip.PH_A_SUM, [e, one, d, hip.X_SUB, hip.X_MUL]), hip.aggexpr(hip.PH_A_SUM, [e, one, d, hip.X_SUB, hip.X_MUL, one, tx, hip.X_ADD, hip.X_MUL]), hip.aggexpr(hip.PH_A_AVG, [q]), hip.aggexpr(hip.PH_A_AVG, [e]), hip.aggexpr(hip.PH_A_AVG, [d]), hip.aggexpr(hip.PH_A_COUNT_STAR)]), [Q, E, D, T, RF, SD])
report("count(*) where shipdate<=", hip.ScanPlan(ctx, t, [cut], [], [hip.aggexpr(hip.PH_A_COUNT_STAR)]), [SD])
# the operator chain on the same narrow shape, for the record
os.environ["PH_SCAN_JIT"] = "0"
report("group by linestatus: sum(ext)  [operator chain]", hip.ScanPlan(ctx, t, [], [LS], [hip.aggexpr(hip.PH_A_SUM, [e])]), [LS, E])
