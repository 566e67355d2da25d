import sys, time
sys.path.insert(0, '.')
from plan_amd import hip, pipelines, tpchgen
sf=(10,1)
n_ord=tpchgen.orders_count(sf)
C=tpchgen.customer(sf)
ctx=hip.Ctx(0)
ck=hip.DevColumn(ctx,hip.PH_I32,C['c_custkey']); cseg=hip.DevColumn(ctx,hip.PH_CODE8,C['c_mktsegment'])
nc=len(C['c_custkey'])
for it in range(6):
    t0=time.perf_counter()
    cs,cn=hip.filter_select(ctx,cseg,nc,hip.PH_EQ,hip.const(hip.PH_I32,i=3))
    t1=time.perf_counter()
    j=hip.Join(ctx,[ck],cs,cn)
    t2=time.perf_counter()
    j.free(); ctx.free(cs)
    t3=time.perf_counter()
    print(it, 'filter %.3f ms build %.3f ms free %.3f ms'%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3), cn)
