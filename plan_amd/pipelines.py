"""Join pipelines assembled from the operator-granular C-ABI calls — what `gpuJoinExecutor` +
`gpuFilterExecutor` + the aggregate sink do for the reference's Q3 plan
(Agg <- HashJoin(l_orderkey=o_orderkey) <- [Scan(lineitem), HashJoin(o_custkey=c_custkey) <-
[Scan(orders), Scan(customer)]]), single GPU or hash-partitioned over N ranks.
"""
import time

import numpy as np

from . import dist, hip, tpchgen


def _raw(typ, ptr, scale=0):
    return hip.Col(typ, scale, ptr, None, None, 0)


class Q3Pipeline:
    """TPC-H Q3 on device-resident customer / orders / lineitem shards.

    Single GPU: filter -> build(customer) -> probe(orders) -> build(orders') -> probe(lineitem)
    -> revenue expression -> group by (l_orderkey, o_orderdate, o_shippriority).
    N ranks: the filtered customer keys are broadcast (small build side), the surviving orders
    and the filtered lineitem rows are hash-partitioned by order key (ph_partition + ph_gather)
    and exchanged with one all-to-all each, so every rank builds/probes/aggregates a disjoint
    key range; the final top-10 is merged from N x 10 rows.
    """

    def __init__(self, ctx, L, O, C, segment="HOUSEHOLD", date=None):
        self.ctx = ctx
        self.date = tpchgen.days(1995, 3, 29) if date is None else date
        self.seg_code = tpchgen.MKTSEGMENT_DICT.index(segment) if segment in tpchgen.MKTSEGMENT_DICT else 999
        D = hip.DevColumn
        self.nc, self.no, self.nl = len(C["c_custkey"]), len(O["o_orderkey"]), len(L["l_orderkey"])
        self.c_key = D(ctx, hip.PH_I32, C["c_custkey"])
        self.c_seg = D(ctx, hip.PH_CODE8, C["c_mktsegment"])
        self.o_key = D(ctx, hip.PH_I64, O["o_orderkey"])
        self.o_cust = D(ctx, hip.PH_I32, O["o_custkey"])
        self.o_date = D(ctx, hip.PH_DATE, O["o_orderdate"])
        self.o_prio = D(ctx, hip.PH_I32, O["o_shippriority"])
        self.l_key = D(ctx, hip.PH_I64, L["l_orderkey"])
        self.l_ext = D(ctx, hip.PH_DEC64, L["l_extendedprice"], 2)
        self.l_disc = D(ctx, hip.PH_DEC64, L["l_discount"], 2)
        self.l_ship = D(ctx, hip.PH_DATE, L["l_shipdate"])
        self.cols = [self.c_key, self.c_seg, self.o_key, self.o_cust, self.o_date, self.o_prio,
                     self.l_key, self.l_ext, self.l_disc, self.l_ship]
        self.revenue_prog = [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]

    def free(self):
        for c in self.cols:
            c.free()

    # -- helpers for the partitioned path (torch tensors as exchange buffers)
    def _gather_t(self, col, idx_ptr, n, dtype):
        import torch
        out = torch.empty(max(n, 1), dtype=dtype, device="cuda")
        if n:
            c = col.col() if isinstance(col, hip.DevColumn) else col
            hip.check(hip.lib().ph_gather(self.ctx.h, hip.ctypes.byref(c), idx_ptr, hip.i64(n),
                                          hip.vp(out.data_ptr())))
        return out[:n]

    def run(self, limit=10, want_groups=False):
        import ctypes
        ctx, date = self.ctx, self.date
        N = dist.world()
        t = {}
        frees = []
        tic = time.perf_counter

        def stage(name, t0):
            ctx.sync()
            t[name] = t.get(name, 0.0) + (tic() - t0)

        # ---- customer filter, build side of join 1
        t0 = tic()
        cs, cn = hip.filter_select(ctx, self.c_seg, self.nc, hip.PH_EQ, hip.const(hip.PH_I32, i=self.seg_code))
        frees.append(cs)
        if N == 1:
            j1 = hip.Join(ctx, [self.c_key], cs, cn)
        else:
            import torch
            mine = self._gather_t(self.c_key, cs, cn, torch.int32)
            ctx.sync()
            allkeys = dist.allgather_rows(mine)       # broadcast of the small build side
            torch.cuda.synchronize()
            self._keep = allkeys
            j1 = hip.Join(ctx, [_raw(hip.PH_I32, allkeys.data_ptr())], None, allkeys.numel())
        stage("customer_filter_build", t0)

        # ---- orders filter + probe join 1
        t0 = tic()
        os_, on = hip.filter_select(ctx, self.o_date, self.no, hip.PH_LT, hip.const(hip.PH_DATE, i=date))
        m1, orow, _c = j1.probe_inner([self.o_cust], os_, on, max(on, 1))
        frees += [os_, orow, _c]
        stage("orders_filter_probe", t0)

        # ---- build side of join 2 (partitioned by o_orderkey when N > 1)
        t0 = tic()
        if N == 1:
            j2 = hip.Join(ctx, [self.o_key], orow, m1)
            b_date, b_prio = self.o_date.col(), self.o_prio.col()   # addressed by orders row id
        else:
            import torch
            counts, perm = hip.partition(ctx, self.o_key, orow, m1, N)
            frees.append(perm)
            send = [self._gather_t(self.o_key, perm, m1, torch.int64),
                    self._gather_t(self.o_date, perm, m1, torch.int32),
                    self._gather_t(self.o_prio, perm, m1, torch.int32)]
            ctx.sync()
            (rk, rd, rp), _ = dist.exchange_columns(send, counts)
            torch.cuda.synchronize()
            self._keep2 = (rk, rd, rp)
            j2 = hip.Join(ctx, [_raw(hip.PH_I64, rk.data_ptr())], None, rk.numel())
            b_date, b_prio = _raw(hip.PH_DATE, rd.data_ptr()), _raw(hip.PH_I32, rp.data_ptr())
        stage("orders_partition_build", t0)

        # ---- lineitem filter (+ partition/exchange) + probe join 2
        t0 = tic()
        lsel, ln = hip.filter_select(ctx, self.l_ship, self.nl, hip.PH_GT, hip.const(hip.PH_DATE, i=date))
        frees.append(lsel)
        stage("lineitem_filter", t0)
        if N == 1:
            p_key, p_ext, p_disc, p_sel, p_n = self.l_key, self.l_ext, self.l_disc, lsel, ln
        else:
            import torch
            t0 = tic()
            counts, perm = hip.partition(ctx, self.l_key, lsel, ln, N)
            frees.append(perm)
            send = [self._gather_t(self.l_key, perm, ln, torch.int64),
                    self._gather_t(self.l_ext, perm, ln, torch.int64),
                    self._gather_t(self.l_disc, perm, ln, torch.int64)]
            ctx.sync()
            stage("lineitem_partition", t0)
            t0 = tic()
            (lk, le, ld), _ = dist.exchange_columns(send, counts)
            import torch as _t
            _t.cuda.synchronize()
            t["lineitem_exchange"] = tic() - t0
            t["exchange_bytes_sent"] = int(sum(counts) - counts[dist.rank()]) * 24
            self._keep3 = (lk, le, ld)
            p_key = _raw(hip.PH_I64, lk.data_ptr())
            p_ext, p_disc = _raw(hip.PH_DEC64, le.data_ptr(), 2), _raw(hip.PH_DEC64, ld.data_ptr(), 2)
            p_sel, p_n = None, lk.numel()
        t0 = tic()
        m2, prow, brow = j2.probe_inner([p_key], p_sel, p_n, max(p_n, 1))
        frees += [prow, brow]
        stage("lineitem_probe", t0)
        t["probe_rows"] = p_n

        # ---- revenue expression + aggregate
        t0 = tic()
        rev, _ = hip.expr_eval(ctx, [p_ext, p_disc], self.revenue_prog, prow, m2)
        gk = hip.gather(ctx, p_key, prow, m2)
        gd = hip.gather(ctx, b_date, brow, m2)
        gp = hip.gather(ctx, b_prio, brow, m2)
        frees += [rev, gk, gd, gp]
        agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_DATE, hip.PH_I32], [(hip.PH_A_SUM, 0)], max(m2 // 2, 1024))
        agg.sink([_raw(hip.PH_I64, gk), _raw(hip.PH_DATE, gd), _raw(hip.PH_I32, gp)],
                 [_raw(hip.PH_DEC64, rev, 4)], None, m2, positional=True)
        stage("expr_aggregate", t0)
        t0 = tic()
        r = agg.finalize(python_ints=False)
        stage("finalize_download", t0)

        # ORDER BY revenue DESC, o_orderdate LIMIT k over this rank's groups (vectorised; the
        # revenue of one order fits int64, which the high word confirms)
        rev_lo = r["sum_lo"][:, 0].view(np.int64)
        assert np.array_equal(r["sum_hi"][:, 0], rev_lo >> 63), "Q3 revenue left the int64 range"
        keys = r["keys"]
        ng = r["ngroups"]
        if ng > limit:   # O(n) selection of everything >= the k-th largest revenue, then a tiny sort
            kth = np.partition(rev_lo, ng - limit)[ng - limit]
            pick = np.nonzero(rev_lo >= kth)[0]
        else:
            pick = np.arange(ng)
        order = pick[np.lexsort((keys[pick, 1], -rev_lo[pick]))][:limit]
        cand = [(int(keys[g, 0]), int(rev_lo[g]), int(keys[g, 1]), int(keys[g, 2])) for g in order]
        top = dist.merge_topk(cand, limit, key=lambda x: (-x[1], x[2]))   # revenue desc, o_orderdate
        groups = None
        if want_groups:
            groups = list(zip(keys[:, 0].tolist(), rev_lo.tolist(), keys[:, 1].tolist(), keys[:, 2].tolist()))
        agg.free()
        j1.free()
        j2.free()
        for p in frees:
            ctx.free(p)
        self._keep = self._keep2 = self._keep3 = None
        return dict(ngroups=r["ngroups"], groups=groups, top=top, join_rows=m2, timings=t)


def q3_text(top):
    """Result text in the reference's format (headline '#' + tabs; Value.String rules) for the
    ORDER BY revenue DESC, o_orderdate LIMIT 10 tail of Q3."""
    import datetime
    lines = ["#\t\t\t"]
    for okey, rev, odate, prio in top:
        neg = "-" if rev < 0 else ""
        w, f = divmod(abs(rev), 10000)
        frac = ("%04d" % f).rstrip("0")
        dec = f"{neg}{w}" + (f".{frac}" if frac else "")
        d = datetime.date(1970, 1, 1) + datetime.timedelta(days=odate)
        lines.append(f"{okey}\t{dec}\t{d.isoformat()}\t{prio}")
    return "\n".join(lines) + "\n"
