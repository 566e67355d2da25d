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


def copartitioned_by_order_key(ctx, o_key_range, l_orderkey):
    """Table statistic of a sharded database (one all-gather of four numbers per rank at load): orders and
    lineitem are CO-PARTITIONED by order key when the ranks' o_orderkey ranges are pairwise disjoint and every
    rank's l_orderkey values lie inside its own range (TPC-H tables split by row ranges are). Then
    lineitem JOIN orders is rank-local — a partition-wise join."""
    if dist.world() == 1:
        return False
    l_lo = int(l_orderkey.min()) if len(l_orderkey) else 0
    l_hi = int(l_orderkey.max()) if len(l_orderkey) else -1
    o_lo, o_hi = o_key_range if o_key_range is not None else (0, -1)
    rec = dist.allgather_records(ctx, np.array([[o_lo, o_hi, l_lo, l_hi]], dtype=np.int64))
    ranges = sorted((int(a), int(b)) for a, b, _, _ in rec if b >= a)
    disjoint = all(ranges[i][1] < ranges[i + 1][0] for i in range(len(ranges) - 1))
    inside = all(int(d) < int(c) or (int(a) <= int(c) and int(d) <= int(b)) for a, b, c, d in rec)
    return bool(disjoint and inside)


def _run_with_fallback(ctx, attempt, relax, can_relax):
    """One query attempt with the optimistic forms (statistics trusted, strict lookups, deferred errors), and — only
    when a deferred PH_ECONSTRAINT says a claim did not hold — a second one after relax().
    One rank: the error surfaces at the next read-back and aborts the attempt there.
    Several ranks: the decision is COLLECTIVE. The ctx HOLDS deferred errors (no read-back in the middle of the
    pipeline reports them, so no rank leaves the sequence of collectives alone — its peers would block in the next
    exchange; the kernels stay in bounds after a broken claim by construction), every rank checks at the end of its
    attempt, and one all-reduce(max) of the flag decides for all of them whether the query runs again."""
    multi = dist.world() > 1
    ctx.set_deferred_errors(2 if multi else True)
    try:
        failed, err, res = 0, None, None
        try:
            res = attempt()
            if multi:
                ctx.check_deferred()
        except hip.PlanHipError as e:
            if e.code != hip.PH_ECONSTRAINT or not can_relax():
                raise
            failed, err = 1, e
        if multi:
            failed = dist.agree_max(ctx, failed)
        if not failed:
            return res
        relax()
        res = attempt()
        if multi:
            ctx.check_deferred()
        return res
    finally:
        ctx.set_deferred_errors(False)


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
        self.time_stages = True   # sync + time every stage (reporting); off in measured steps
        self.probe_events = None  # (start, end) torch events recorded around the lineitem probe calls
        self.date = tpchgen.days(1995, 3, 29) if date is None else date
        self.seg_code = tpchgen.MKTSEGMENT_DICT.index(segment) if segment in tpchgen.MKTSEGMENT_DICT else 999
        D = hip.DevColumn
        self.nc, self.no, self.nl = len(C["c_custkey"]), len(O["o_orderkey"]), len(L["l_orderkey"])
        self.c_key = D(ctx, hip.PH_I32, C["c_custkey"])
        # column statistics (what ph_table_col_range keeps for a table): dense primary keys build direct tables
        self.c_key_range = (int(C["c_custkey"].min()), int(C["c_custkey"].max())) if self.nc else None
        self.c_seg = D(ctx, hip.PH_CODE8, C["c_mktsegment"])
        self.o_key = D(ctx, hip.PH_I64, O["o_orderkey"])
        self.o_key_range = (int(O["o_orderkey"].min()), int(O["o_orderkey"].max())) if self.no else None
        # column statistics, computed once at load like the range: a primary key in storage order
        self.o_key_sorted_unique = bool(self.no > 1 and np.all(np.diff(O["o_orderkey"]) > 0))
        self.c_key_sorted_unique = bool(self.nc > 1 and np.all(np.diff(C["c_custkey"]) > 0))
        self.o_cust = D(ctx, hip.PH_I32, O["o_custkey"])
        self.o_date = D(ctx, hip.PH_DATE, O["o_orderdate"])
        self.o_prio = D(ctx, hip.PH_I32, O["o_shippriority"])
        self.l_key = D(ctx, hip.PH_I64, L["l_orderkey"])
        # column statistics: lineitem clustered by order key -> the join output arrives ordered by the group key
        self.l_key_sorted = bool(self.nl > 1 and np.all(np.diff(L["l_orderkey"]) >= 0))
        self.l_ext = D(ctx, hip.PH_DEC64, L["l_extendedprice"], 2)
        self.l_disc = D(ctx, hip.PH_DEC64, L["l_discount"], 2)
        self.l_ship = D(ctx, hip.PH_DATE, L["l_shipdate"])
        self.cols = [self.c_key, self.c_seg, self.o_key, self.o_cust, self.o_date, self.o_prio,
                     self.l_key, self.l_ext, self.l_disc, self.l_ship]
        self.revenue_prog = [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL]
        # orders and lineitem co-partitioned by order key (see copartitioned_by_order_key): lineitem JOIN orders
        # and the group-by on l_orderkey are rank-local; only the customer keys (the small build side) cross
        # the links, and the top-k candidates at the end
        self.allow_partitionwise = True   # False: always the hash-partitioned exchange plan
        self.semijoin_reduce = True       # the exchange plan filters lineitem by the all-gathered qualifying order keys first
        self.copartitioned = copartitioned_by_order_key(ctx, self.o_key_range, L["l_orderkey"])

    def free(self):
        for c in self.cols:
            c.free()

    def lineitem_filter_rows(self):
        """rows of this rank's lineitem that pass l_shipdate > date (reporting only)"""
        s, c = hip.filter_select(self.ctx, self.l_ship, self.nl, hip.PH_GT, hip.const(hip.PH_DATE, i=self.date))
        self.ctx.free(s)
        return c

    def orders_filter_rows(self):
        """rows of this rank's orders that pass o_orderdate < date (reporting only)"""
        s, c = hip.filter_select(self.ctx, self.o_date, self.no, hip.PH_LT, hip.const(hip.PH_DATE, i=self.date))
        self.ctx.free(s)
        return c

    def run(self, limit=10, want_groups=False):
        """the revenue expression's overflow flag is a deferred error (seen at the result download); so is
        a violation of the sorted-and-unique statistic the orders build relies on, after which the query
        runs again without it"""
        return _run_with_fallback(self.ctx, lambda: self._run(limit, want_groups), self._drop_statistics,
                                  lambda: self.o_key_sorted_unique or self.l_key_sorted or self.c_key_sorted_unique)

    def _drop_statistics(self):
        self.o_key_sorted_unique = self.l_key_sorted = self.c_key_sorted_unique = False   # a statistic did not hold: the general forms

    def _run(self, limit, want_groups):
        ctx, date = self.ctx, self.date
        N = dist.world()
        W = N   # ranks the customer keys are gathered over
        if N > 1 and self.copartitioned and self.allow_partitionwise:
            N = 1   # partition-wise join: everything behind the customer build is this rank's own (see __init__)
        t = {}
        frees = []
        tic = time.perf_counter

        def stage(name, t0):
            if self.time_stages:   # a sync per stage costs ~20 us of idle GPU each: reporting runs only
                ctx.sync()
                t[name] = t.get(name, 0.0) + (tic() - t0)

        # ---- customer filter, build side of join 1
        t0 = tic()
        j1 = None
        if W == 1 and self.c_key_range is not None:
            # Filter(c_mktsegment = ..) under the build child, fused into the build: c_custkey is a dense
            # primary key, so the table is a direct table sized by the key range and the number of
            # customers that pass never has to reach the host (no selection vector, no read-back)
            j1 = hip.Join.build_where(ctx, [self.c_key], self.c_seg, hip.PH_EQ, hip.const(hip.PH_I32, i=self.seg_code),
                                      None, self.nc, self.c_key_range,
                                      sorted_unique=self.c_key_sorted_unique and not getattr(self, "no_gated_fill", False))
            if j1 is not None:
                cn = j1.count() if self.time_stages else 0   # reporting only
        if j1 is not None:
            pass
        elif W == 1:
            cs, cn = hip.filter_select(ctx, self.c_seg, self.nc, hip.PH_EQ, hip.const(hip.PH_I32, i=self.seg_code))
            frees.append(cs)
            # build on the gathered keys (no selection inside the table: one dependent read less per
            # chain step of the probe); the build-side row ids of this join are never used
            ck = hip.gather(ctx, self.c_key, cs, cn)
            frees.append(ck)
            j1 = hip.Join(ctx, [_raw(hip.PH_I32, ck)], None, cn, key_range=self.c_key_range)
        else:
            cs, cn = hip.filter_select(ctx, self.c_seg, self.nc, hip.PH_EQ, hip.const(hip.PH_I32, i=self.seg_code))
            frees.append(cs)
            mine = hip.gather(ctx, self.c_key, cs, cn)
            allkeys, nall = dist.allgather_rows(ctx, mine, cn, np.int32)   # broadcast of the small build side
            frees += [mine, allkeys]
            if not hasattr(self, "_c_key_range_all"):   # column statistics over all shards, once per table
                self._c_key_range_all = dist.global_range(ctx, self.c_key_range)
            j1 = hip.Join(ctx, [_raw(hip.PH_I32, allkeys)], None, nall, key_range=self._c_key_range_all)
        stage("customer_filter_build", t0)

        # ---- orders filter + probe join 1
        t0 = tic()
        # The build child of join 2 is  orders SEMI JOIN customer WHERE o_orderdate < date  (customer gives
        # the output nothing but its key, which is unique). Residual form: ONE pass marks the orders rows
        # that qualify (a byte per orders row: no pair list, no count read-back), join 2 is built over the
        # WHOLE orders table (its primary key in storage order: one streaming fill of a direct table,
        # independent of the mark), and the lineitem probe tests the flag of the orders row it finds.
        residual = None
        gated = False   # the flags were consumed by join 2's build (no residual test in the probe)
        if N == 1 and self.o_key_range is not None and j1.kind == "direct" and not getattr(self, "no_residual", False):
            flags = j1.probe_mark_where([self.o_cust], self.o_date, hip.PH_LT, hip.const(hip.PH_DATE, i=date), self.no)
            if flags is not None:
                frees.append(flags)
                stage("orders_filter_probe", t0)
                t0 = tic()
                j2 = None
                if self.o_key_sorted_unique and not getattr(self, "no_gated_fill", False):
                    # the flags ride along in the verified sorted fill: only qualifying orders get a slot, and the
                    # fill writes the occupancy bitmap the lineitem probe tests before it touches the slot array
                    j2 = hip.Join.build_where(ctx, [self.o_key], _raw(hip.PH_CODE8, flags), hip.PH_EQ, hip.const(hip.PH_I32, i=1),
                                              None, self.no, self.o_key_range, sorted_unique=True)
                    gated = j2 is not None and j2.kind == "direct"
                    if j2 is not None and not gated:
                        j2.free()
                        j2 = None
                if j2 is None:
                    j2 = hip.Join(ctx, [self.o_key], None, self.no, key_range=self.o_key_range, sorted_unique=self.o_key_sorted_unique)
                if j2.kind == "direct":
                    residual = flags
                    b_date, b_prio = self.o_date.col(), self.o_prio.col()   # addressed by orders row id
                    m1 = 0
                    if self.time_stages:   # reporting only: how many orders rows qualify
                        qs, m1 = hip.filter_select(ctx, _raw(hip.PH_CODE8, flags), self.no, hip.PH_EQ, hip.const(hip.PH_I32, i=1))
                        ctx.free(qs)
                    stage("orders_partition_build", t0)
                else:
                    j2.free()
        fused = None
        if residual is None:
            fused = j1.probe_inner_where([self.o_cust], self.o_date, hip.PH_LT, hip.const(hip.PH_DATE, i=date),
                                         None, self.no, self.no)
        if residual is not None:
            pass
        elif fused is not None:
            m1, orow, _c = fused
            frees += [orow, _c]
        else:
            os_, on = hip.filter_select(ctx, self.o_date, self.no, hip.PH_LT, hip.const(hip.PH_DATE, i=date))
            m1, orow, _c = j1.probe_inner([self.o_cust], os_, on, max(on, 1))
            frees += [os_, orow, _c]
        if residual is None:
            stage("orders_filter_probe", t0)

        # ---- build side of join 2 (partitioned by o_orderkey when N > 1)
        t0 = tic()
        if residual is not None:
            pass
        elif N == 1:
            j2 = hip.Join(ctx, [self.o_key], orow, m1)
            b_date, b_prio = self.o_date.col(), self.o_prio.col()   # addressed by orders row id
        else:
            # partition -> gathers -> count matrix -> all-to-all, all stream-ordered; the count
            # matrix is the stage's one host round trip
            counts_dev, perm = hip.partition_dev(ctx, self.o_key, orow, m1, N)
            send = [(hip.gather(ctx, self.o_key, perm, m1), np.int64),
                    (hip.gather(ctx, self.o_date, perm, m1), np.int32),
                    (hip.gather(ctx, self.o_prio, perm, m1), np.int32)]
            (rk, rd, rp), nrecv, _sent = dist.exchange(ctx, send, counts_dev, m1)
            frees += [counts_dev, perm, rk, rd, rp] + [p for p, _ in send]
            j2 = hip.Join(ctx, [_raw(hip.PH_I64, rk)], None, nrecv)
            b_date, b_prio = _raw(hip.PH_DATE, rd), _raw(hip.PH_I32, rp)
        if residual is None:
            stage("orders_partition_build", t0)

        # ---- lineitem filter (+ partition/exchange) + probe join 2
        t0 = tic()
        fused2 = None
        if N == 1:
            p_key, p_ext, p_disc = self.l_key, self.l_ext, self.l_disc
            if self.probe_events:
                self.probe_events[0].record()
            if residual is not None and not gated:
                fused2 = j2.probe_inner_residual([p_key], self.l_ship, hip.PH_GT, hip.const(hip.PH_DATE, i=date), residual,
                                                 None, self.nl, self.nl)
            else:
                fused2 = j2.probe_inner_where([p_key], self.l_ship, hip.PH_GT, hip.const(hip.PH_DATE, i=date),
                                              None, self.nl, self.nl)
            if fused2 is not None:
                if self.probe_events:
                    self.probe_events[1].record()
                stage("lineitem_filter_probe", t0)
                t["probe_rows"] = self.nl   # rows streamed by the fused filter+probe
        if fused2 is None and N > 1 and self.semijoin_reduce:
            # SEMI-JOIN REDUCTION in front of the exchange: every rank learns the order keys that qualify anywhere
            # (8 B per qualifying order, all-gathered: 12 MB per rank at SF10) and sends only the lineitem rows
            # whose order qualifies — 0.5 % of the rows the date filter keeps, so the all-to-all moves ~24 B x 0.3 M
            # rows per rank instead of 24 B x 32 M, and the partition + gathers in front of it shrink likewise
            qk = hip.gather(ctx, self.o_key, orow, m1)
            allq, nq = dist.allgather_rows(ctx, qk, m1, np.int64)
            frees += [qk, allq]
            jq = hip.Join(ctx, [_raw(hip.PH_I64, allq)], None, nq)
            red = jq.probe_inner_where([self.l_key], self.l_ship, hip.PH_GT, hip.const(hip.PH_DATE, i=date), None, self.nl, self.nl)
            if red is not None:
                ln, lsel, _b = red
                frees += [lsel, _b]
            else:
                ls0, l0 = hip.filter_select(ctx, self.l_ship, self.nl, hip.PH_GT, hip.const(hip.PH_DATE, i=date))
                ln, lsel, _b = jq.probe_inner([self.l_key], ls0, l0, max(l0, 1))
                frees += [ls0, lsel, _b]
            jq.free()
            t["semijoin_keys_gathered"] = nq
            stage("lineitem_filter", t0)
        elif fused2 is None:
            lsel, ln = hip.filter_select(ctx, self.l_ship, self.nl, hip.PH_GT, hip.const(hip.PH_DATE, i=date))
            frees.append(lsel)
            stage("lineitem_filter", t0)
        if fused2 is not None:
            pass
        elif N == 1:
            p_key, p_ext, p_disc, p_sel, p_n = self.l_key, self.l_ext, self.l_disc, lsel, ln
        else:
            t0 = tic()
            counts_dev, perm = hip.partition_dev(ctx, self.l_key, lsel, ln, N)
            send = [(hip.gather(ctx, self.l_key, perm, ln), np.int64),
                    (hip.gather(ctx, self.l_ext, perm, ln), np.int64),
                    (hip.gather(ctx, self.l_disc, perm, ln), np.int64)]
            stage("lineitem_partition", t0)
            t0 = tic()
            (lk, le, ld), nrecv, sent = dist.exchange(ctx, send, counts_dev, ln)
            stage("lineitem_exchange", t0)
            t["exchange_bytes_sent"] = sent * 24
            frees += [counts_dev, perm, lk, le, ld] + [p for p, _ in send]
            p_key = _raw(hip.PH_I64, lk)
            p_ext, p_disc = _raw(hip.PH_DEC64, le, 2), _raw(hip.PH_DEC64, ld, 2)
            p_sel, p_n = None, nrecv
        if fused2 is not None:
            m2, prow, brow = fused2
            frees += [prow, brow]
        else:
            t0 = tic()
            m2, prow, brow = j2.probe_inner([p_key], p_sel, p_n, max(p_n, 1))
            frees += [prow, brow]
            stage("lineitem_probe", t0)
            t["probe_rows"] = p_n

        # ---- revenue expression + aggregate
        t0 = tic()
        rev, _ = hip.expr_eval(ctx, [p_ext, p_disc], self.revenue_prog, prow, m2)
        gk = hip.gather(ctx, p_key, prow, m2)
        gd, gp = hip.gather_multi(ctx, [b_date, b_prio], brow, m2)
        frees += [rev, gk, gd, gp]
        agg = hip.Agg(ctx, [hip.PH_I64, hip.PH_DATE, hip.PH_I32], [(hip.PH_A_SUM, 0)], max(m2 // 2, 1024))
        gkeys, gargs = [_raw(hip.PH_I64, gk), _raw(hip.PH_DATE, gd), _raw(hip.PH_I32, gp)], [_raw(hip.PH_DEC64, rev, 4)]
        # the pairs come out in lineitem order; lineitem clustered by order key (a column statistic) makes
        # every group one run of adjacent rows: the streaming aggregate instead of the hash table
        streamed = N == 1 and fused2 is not None and self.l_key_sorted and not getattr(self, "no_stream_agg", False) and \
            agg.sink_sorted(gkeys, gargs, m2)
        if not streamed:
            agg.sink(gkeys, gargs, None, m2, positional=True)
        stage("expr_aggregate", t0)
        t0 = tic()
        # the total group count is reporting only (one more host round trip): not in measured steps
        ngroups_total = agg.group_count() if (self.time_stages or want_groups) else None
        # the intermediates go back to the (stream-ordered) pool BEFORE the host blocks in the fetch: the
        # bookkeeping runs while the GPU is still busy instead of between two queries
        ctx.free_many(frees)
        frees = []
        if want_groups:
            r = agg.finalize(python_ints=False)
        else:   # ORDER BY revenue DESC ... LIMIT: only the groups at least as good as the k-th
            r = agg.topk(0, limit, descending=True)
        stage("finalize_download", t0)

        # ORDER BY revenue DESC, o_orderdate LIMIT k over this rank's groups (vectorised; the
        # revenue of one order fits int64, which the high word confirms)
        rev_lo = r["sum_lo"][:, 0].view(np.int64)
        keys = r["keys"]
        ng = r["ngroups"]
        if ng <= 64 and not want_groups:
            # the usual case after the device top-k (k rows + ties): plain Python over a handful of rows — a dozen
            # numpy calls on 10-element arrays cost more host time between two queries than the sort itself
            rl, hl, kl = rev_lo.tolist(), r["sum_hi"][:, 0].tolist(), keys.tolist()
            assert all(h == (v >> 63) for h, v in zip(hl, rl)), "Q3 revenue left the int64 range"
            cand = sorted(((kl[g][0], rl[g], kl[g][1], kl[g][2]) for g in range(ng)), key=lambda x: (-x[1], x[2]))[:limit]
            top = dist.merge_topk(cand, limit, key=lambda x: (-x[1], x[2]), ctx=ctx)   # revenue desc, o_orderdate
            agg.free()
            j1.free()
            j2.free()
            ctx.free_many(frees)
            return dict(ngroups=ngroups_total, groups=None, top=top, join_rows=m2, build_rows=cn + m1, timings=t)
        assert np.array_equal(r["sum_hi"][:, 0], rev_lo >> 63), "Q3 revenue left the int64 range"
        if ng > limit:   # O(n) selection of everything >= the k-th largest revenue, then a tiny sort
            kth = np.partition(rev_lo, ng - limit)[ng - limit]
            pick = np.nonzero(rev_lo >= kth)[0]
        else:
            pick = np.arange(ng)
        order = pick[np.lexsort((keys[pick, 1], -rev_lo[pick]))][:limit]
        cand = [(int(keys[g, 0]), int(rev_lo[g]), int(keys[g, 1]), int(keys[g, 2])) for g in order]
        top = dist.merge_topk(cand, limit, key=lambda x: (-x[1], x[2]), ctx=ctx)   # revenue desc, o_orderdate
        groups = None
        if want_groups:
            groups = list(zip(keys[:, 0].tolist(), rev_lo.tolist(), keys[:, 1].tolist(), keys[:, 2].tolist()))
        agg.free()
        j1.free()
        j2.free()
        ctx.free_many(frees)
        return dict(ngroups=ngroups_total, groups=groups, top=top, join_rows=m2, build_rows=cn + m1, timings=t)


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


class Q9Pipeline:
    """TPC-H Q9 (cases/tpch/query/q9.sql) on one GPU: LIKE filter on part, four N:1 hash joins from
    lineitem (part, partsupp on the composite key, supplier, orders), the profit expression
    l_extendedprice*(1-l_discount) - ps_supplycost*l_quantity, group by (nation, year).
    The nation join (25 rows, key = s_nationkey) is the identity on the key and is folded into the
    group key; names are attached when the result text is produced.
    The intermediate result is carried as aligned arrays of base-table row ids; after each join the
    earlier arrays are re-gathered by the probe positions that survived."""

    def __init__(self, ctx, L, O, P, PS, S, pattern="%pink%"):
        self.ctx, self.pattern = ctx, pattern
        self.time_stages = True
        D = hip.DevColumn
        self.n = dict(l=len(L["l_orderkey"]), o=len(O["o_orderkey"]), p=len(P["p_partkey"]),
                      ps=len(PS["ps_partkey"]), s=len(S["s_suppkey"]))
        self.p_key = D(ctx, hip.PH_I32, P["p_partkey"])
        self.p_key_range = (int(P["p_partkey"].min()), int(P["p_partkey"].max())) if self.n["p"] else None
        self.p_name = D(ctx, hip.PH_STR, P["p_name_off"], aux=P["p_name_bytes"])
        self.ps_part = D(ctx, hip.PH_I32, PS["ps_partkey"])
        self.ps_supp = D(ctx, hip.PH_I32, PS["ps_suppkey"])
        self.ps_cost = D(ctx, hip.PH_DEC64, PS["ps_supplycost"], 2)
        self.s_key = D(ctx, hip.PH_I32, S["s_suppkey"])
        # column statistics: dense primary keys (supplier, orders) build direct tables
        self.s_key_range = (int(S["s_suppkey"].min()), int(S["s_suppkey"].max())) if self.n["s"] else None
        self.s_key_sorted_unique = bool(self.n["s"] > 1 and np.all(np.diff(S["s_suppkey"]) > 0))
        self.o_key_range = (int(O["o_orderkey"].min()), int(O["o_orderkey"].max())) if self.n["o"] else None
        self.o_key_sorted_unique = bool(self.n["o"] > 1 and np.all(np.diff(O["o_orderkey"]) > 0))
        self.s_nat = D(ctx, hip.PH_I32, S["s_nationkey"])
        self.o_key = D(ctx, hip.PH_I64, O["o_orderkey"])
        self.o_date = D(ctx, hip.PH_DATE, O["o_orderdate"])
        # The five lineitem columns that are fetched together at the rows surviving the part join live in ONE resident table with a
        # co-located copy (ph_table_colocate: the planner names the set; ph_plan finds it by itself on the second run): the late
        # materialisation then reads one 64-byte sector per surviving row instead of one per column
        nl = len(L["l_orderkey"])
        self.l_tab = hip.Table(ctx, [(hip.PH_I64, L["l_orderkey"]), (hip.PH_I32, L["l_suppkey"]), (hip.PH_I32, L["l_quantity"]),
                                     (hip.PH_DEC64, L["l_extendedprice"], 2), (hip.PH_DEC64, L["l_discount"], 2)], nl)
        if nl >= (1 << 16):
            self.l_tab.colocate([0, 1, 2, 3, 4])
        self.l_key = hip.TableColumn(self.l_tab, 0)
        # column statistic: lineitem clustered by order key -> the surviving rows reach the orders join in key order
        self.l_key_sorted = bool(self.n["l"] > 1 and np.all(np.diff(L["l_orderkey"]) >= 0))
        # shards co-partitioned by order key: the one large join, lineitem x orders, needs no exchange
        self.allow_partitionwise = True
        self.copartitioned = copartitioned_by_order_key(ctx, self.o_key_range, L["l_orderkey"])
        self.l_part = D(ctx, hip.PH_I32, L["l_partkey"])
        self.l_supp = hip.TableColumn(self.l_tab, 1)
        self.l_qty = hip.TableColumn(self.l_tab, 2)
        self.l_ext = hip.TableColumn(self.l_tab, 3)
        self.l_disc = hip.TableColumn(self.l_tab, 4)
        self.cols = [self.p_key, self.p_name, self.ps_part, self.ps_supp, self.ps_cost, self.s_key,
                     self.s_nat, self.o_key, self.o_date, self.l_part]

    def free(self):
        for c in self.cols:
            c.free()
        self.l_tab.free()

    def run(self):
        """The three N:1 joins first run as STRICT lookups (foreign keys into primary keys: no
        statistics read-back; a violation is a deferred PH_ECONSTRAINT, seen at the final download)
        with the expression's overflow flag deferred the same way — then, only if that error came,
        again with the counted lookups, which drop unmatched rows as an inner join must."""
        st = {"strict": True}

        def relax():
            self.ctx.set_async_counts(False)   # counts of the abandoned attempt land in variables that are still alive
            st["strict"] = False
        try:
            return _run_with_fallback(self.ctx, lambda: self._run(strict=st["strict"]), relax, lambda: st["strict"])
        finally:
            self.ctx.set_async_counts(False)   # waits for counts still in flight (their variables live in self._counts)
            self._counts = []

    def _run(self, strict):
        """N == 1: everything local. N > 1 (one process per GPU, tables sharded by row ranges):
        the small build sides are broadcast — pink part keys, the partsupp rows of pink parts (found
        with a semi-join against the broadcast part keys), supplier — and the one large join,
        lineitem x orders, is hash-partitioned by order key on both sides and exchanged with an
        all-to-all each (multi-stage: part-key stage local after the broadcasts, order-key stage
        partitioned). The 175 partial groups are merged at the end."""
        ctx = self.ctx
        N = dist.world()
        t, frees = {}, []
        tic = time.perf_counter

        def stage(name, t0):
            if self.time_stages:
                ctx.sync()
                t[name] = tic() - t0

        def gat(col, idx, n):
            p = hip.gather(ctx, col, idx, n)
            frees.append(p)
            return p

        def bcast(col, idx, n, dtype):
            """gather rows idx of col and all-gather them over the ranks -> (device pointer, count)"""
            mine = gat(col, idx, n)
            allv, total = dist.allgather_rows(ctx, mine, n, dtype)
            frees.append(allv)
            return allv, total

        t0 = tic()
        # Pipelined host round trips (N == 1): a count the host needs is asked for asynchronously, work that
        # does not depend on it is queued behind it, and only then the host waits — so the GPU runs the
        # supplier and orders builds while the LIKE count travels, and the partsupp semi-join while the
        # pair count of the part join does.
        pipelined = N == 1 and not self.time_stages
        # orders: both sides of the last join are ordered by the key (o_orderkey a primary key in storage order,
        # the intermediate in lineitem order, lineitem clustered by l_orderkey): a merge lookup, no table
        local_orders = N == 1 or (self.copartitioned and self.allow_partitionwise)   # partition-wise join on the order key
        merge_orders = (strict and local_orders and self.o_key_sorted_unique and self.l_key_sorted
                        and not getattr(self, "no_merge_lookup", False))
        js = jo = None
        if pipelined:
            ctx.set_async_counts(True)
            psel, np_c = hip.filter_select(ctx, self.p_name, self.n["p"], hip.PH_LIKE, hip.const(hip.PH_STR, s=self.pattern), defer=True)
            self._counts = [np_c]
            js = hip.Join(ctx, [self.s_key], None, self.n["s"], key_range=self.s_key_range, sorted_unique=strict and self.s_key_sorted_unique)
            if not merge_orders:
                jo = hip.Join(ctx, [self.o_key], None, self.n["o"], key_range=self.o_key_range, sorted_unique=strict and self.o_key_sorted_unique)
            ctx.wait_counts()
            np_ = np_c.value
        else:
            psel, np_ = hip.filter_select(ctx, self.p_name, self.n["p"], hip.PH_LIKE,
                                          hip.const(hip.PH_STR, s=self.pattern))
        frees.append(psel)
        if N == 1:
            pk = hip.gather(ctx, self.p_key, psel, np_)   # as in Q3: no selection inside the table
            frees.append(pk)
            j = hip.Join(ctx, [_raw(hip.PH_I32, pk)], None, np_, key_range=self.p_key_range)
        else:
            pk, npk = bcast(self.p_key, psel, np_, np.int32)
            if not hasattr(self, "_ranges_all"):   # column statistics over all shards, once per table
                self._ranges_all = {k: dist.global_range(ctx, r) for k, r in
                                    (("p", self.p_key_range), ("s", self.s_key_range), ("o", self.o_key_range))}
            j = hip.Join(ctx, [_raw(hip.PH_I32, pk)], None, npk, key_range=self._ranges_all["p"])
        stage("part_like_build", t0)
        t0 = tic()
        n1, lrow, prow_part = j.probe_inner([self.l_part], None, self.n["l"], self.n["l"], defer=pipelined)
        frees += [lrow, prow_part]
        stage("lineitem_probe_part", t0)

        t0 = tic()
        # Semi-join reduction of the build side: ps_partkey = l_partkey and l_partkey is a pink part,
        # so only partsupp rows of pink parts can match (4 per pink part, ~5 % of partsupp). They are
        # found with a mark probe of partsupp against the part table and only they are built: a
        # 0.4 M-row partitioned build instead of an 8 M-row atomic one (0.37 ms), and a chain walk in
        # a cache-resident table. partsupp stays the BUILD side: probing WITH the intermediate keeps
        # it in lineitem order, so every later gather by its row ids walks the base columns forwards
        # (3.13 vs 3.42 ms per query with the sides swapped).
        f = j.probe_mark([self.ps_part], None, self.n["ps"])
        frees.append(f)
        fsel, fn = hip.filter_select(ctx, _raw(hip.PH_CODE8, f), self.n["ps"], hip.PH_EQ, hip.const(hip.PH_I32, i=1), defer=pipelined)
        frees.append(fsel)
        if pipelined:   # both counts (pairs of the part join, partsupp rows of pink parts) in one wait
            self._counts += [n1, fn]
            ctx.wait_counts()
            n1, fn = n1.value, fn.value
            ctx.set_async_counts(False)
        j.free()
        if N == 1:
            bp, bs, bc = hip.gather_multi(ctx, [self.ps_part, self.ps_supp, self.ps_cost], fsel, fn)
            frees += [bp, bs, bc]
            jps = hip.Join(ctx, [_raw(hip.PH_I32, bp), _raw(hip.PH_I32, bs)], None, fn, fk_probes=True)
            ps_cost = _raw(hip.PH_DEC64, bc, 2)
        else:
            # ... and broadcast, so every rank can resolve its own lineitem rows
            bp, nb = bcast(self.ps_part, fsel, fn, np.int32)
            bs, _ = bcast(self.ps_supp, fsel, fn, np.int32)
            bc, _ = bcast(self.ps_cost, fsel, fn, np.int64)
            jps = hip.Join(ctx, [_raw(hip.PH_I32, bp), _raw(hip.PH_I32, bs)], None, nb, fk_probes=True)
            ps_cost = _raw(hip.PH_DEC64, bc, 2)
        # LATE MATERIALISATION, ONCE: the six lineitem columns the rest of the query needs are fetched
        # at the surviving rows in one pass (ph_gather_multi: every column read of a row in flight
        # together; one launch per column — or a selection inside every later kernel — paid the two
        # dependent latencies of a gather per column). From here on the intermediate is positional
        # and dense: the per-join materialisation of Scan.gatherResult (join_scan.go:250-278), done once.
        # (l_partkey of a surviving row is its matched part's key: read through the pair's build row from
        # the 109 k gathered part keys, cache resident, instead of one more 64-byte sector per row of lineitem)
        d_part = gat(_raw(hip.PH_I32, pk), prow_part, n1)
        d_supp, c_okey, d_ext, d_disc, d_qty = hip.gather_multi(
            ctx, [self.l_supp, self.l_key, self.l_ext, self.l_disc, self.l_qty], lrow, n1)
        frees += [d_supp, c_okey, d_ext, d_disc, d_qty]
        if getattr(self, "keep_gather_ids", False):   # bench.py times this kernel on the very same row ids afterwards
            self.kept_gather_ids = (ctx.upload(ctx.download(lrow, np.int32, n1)), n1)
        # The two joins below are N:1 (partsupp's composite primary key, supplier's key): LOOKUP
        # probes — one kernel each, no candidate/scan/emit pipeline, no re-gather of earlier columns.
        stats = None
        if not strict:
            stats = ctx.alloc(8)
            frees.append(stats)
            hip.check(hip.lib().ph_dev_memset(ctx.h, stats, 0, hip.i64(8)))
        lookup = (lambda j, keys, n: j.lookup_strict(keys, None, n)) if strict else (lambda j, keys, n: j.lookup(keys, None, n, stats))
        psrow = lookup(jps, [_raw(hip.PH_I32, d_part), _raw(hip.PH_I32, d_supp)], n1)
        jps.free()
        frees.append(psrow)
        stage("partsupp_join", t0)

        t0 = tic()
        if N == 1:
            if js is None:
                js = hip.Join(ctx, [self.s_key], None, self.n["s"], key_range=self.s_key_range, sorted_unique=strict and self.s_key_sorted_unique)
            s_nat = self.s_nat
        else:
            ident = ctx.upload(np.arange(self.n["s"], dtype=np.int32))
            frees.append(ident)
            sk, nsk = bcast(self.s_key, ident, self.n["s"], np.int32)
            sn, _ = bcast(self.s_nat, ident, self.n["s"], np.int32)
            js = hip.Join(ctx, [_raw(hip.PH_I32, sk)], None, nsk, key_range=self._ranges_all["s"])
            s_nat = _raw(hip.PH_I32, sn)
        srow = lookup(js, [_raw(hip.PH_I32, d_supp)], n1)
        frees.append(srow)
        js.free()
        # one read for both joins: rows without a match (none in TPC-H: foreign keys) would have to
        # leave the intermediate, rows with several matches would need the pair-emitting probe
        misses, multi = (0, 0) if strict else ctx.download(stats, np.int32, 2).tolist()
        if multi:
            raise hip.PlanHipError(hip.PH_EUNSUPPORTED, "Q9: partsupp / supplier keys are not unique; use probe_inner")
        n3 = n1
        if misses:   # inner-join semantics: keep the positions both lookups resolved
            okp, c1 = hip.filter_select(ctx, _raw(hip.PH_I32, psrow), n1, hip.PH_GE, hip.const(hip.PH_I32, i=0))
            ok2, n3 = hip.filter_select(ctx, _raw(hip.PH_I32, srow), n1, hip.PH_GE, hip.const(hip.PH_I32, i=0), okp, c1)
            frees += [okp, ok2]
            psrow, srow = gat(_raw(hip.PH_I32, psrow), ok2, n3), gat(_raw(hip.PH_I32, srow), ok2, n3)
            c_okey, d_ext, d_disc, d_qty = hip.gather_multi(
                ctx, [_raw(hip.PH_I64, c_okey), _raw(hip.PH_DEC64, d_ext, 2), _raw(hip.PH_DEC64, d_disc, 2), _raw(hip.PH_I32, d_qty)], ok2, n3)
            frees += [c_okey, d_ext, d_disc, d_qty]
        stage("supplier_join", t0)

        # ---- the profit expression, positional: l_extendedprice * (1 - l_discount) - ps_supplycost *
        # l_quantity in ONE program (scale 4), evaluated before the orders join so that the join (and,
        # on several GPUs, the exchange) carries one 8-byte amount instead of four columns
        t0 = tic()
        c_cost, c_nat = gat(ps_cost, psrow, n3), gat(s_nat, srow, n3)
        c_amount, _vb = hip.expr_eval(ctx, [_raw(hip.PH_DEC64, d_ext, 2), _raw(hip.PH_DEC64, d_disc, 2), _raw(hip.PH_DEC64, c_cost, 2), _raw(hip.PH_I32, d_qty)],
                                      [hip.X_COL(0), hip.X_CONST(1), hip.X_COL(1), hip.X_SUB, hip.X_MUL,
                                       hip.X_COL(2), hip.X_COL(3), hip.X_MUL, hip.X_SUB], None, n3)
        frees.append(c_amount)
        if local_orders:
            # jo: built below on this rank's orders unless the pipelined form already queued it
            o_date = self.o_date.col()
            m = n3
        else:
            # order-key stage: both sides hash-partitioned by order key and exchanged
            counts_dev, perm = hip.partition_dev(ctx, _raw(hip.PH_I64, c_okey), None, n3, N)
            send = [(gat(_raw(hip.PH_I64, c_okey), perm, n3), np.int64),
                    (gat(_raw(hip.PH_DEC64, c_amount), perm, n3), np.int64),
                    (gat(_raw(hip.PH_I32, c_nat), perm, n3), np.int32)]
            recv, m, sent = dist.exchange(ctx, send, counts_dev, n3)
            ocounts_dev, operm = hip.partition_dev(ctx, self.o_key, None, self.n["o"], N)
            osend = [(gat(self.o_key, operm, self.n["o"]), np.int64), (gat(self.o_date, operm, self.n["o"]), np.int32)]
            orecv, mo, osent = dist.exchange(ctx, osend, ocounts_dev, self.n["o"])
            frees += [counts_dev, perm, ocounts_dev, operm] + recv + orecv
            t["exchange_bytes_sent"] = sent * 20 + osent * 12
            c_okey, c_amount, c_nat = recv
            jo = hip.Join(ctx, [_raw(hip.PH_I64, orecv[0])], None, mo, key_range=self._ranges_all["o"])
            o_date = _raw(hip.PH_DATE, orecv[1])
        # orders is the BUILD side (o_orderkey is its primary key) and the intermediate looks its order
        # up: N:1 again, so one lookup kernel and the intermediate stays positional — no pair emission,
        # no gathers of the amount / nation columns. Building 15 M order keys costs 0.37 ms with the
        # node table (was 0.65 ms with one atomic per row, which is why round 1 built the 3.3 M-row
        # intermediate instead and probed it with all 15 M orders: 0.83 ms for the stage).
        if merge_orders:
            orow = hip.merge_lookup(ctx, self.o_key, self.n["o"], _raw(hip.PH_I64, c_okey), None, m, strict=True)
        else:
            if jo is None:
                jo = hip.Join(ctx, [self.o_key], None, self.n["o"], key_range=self.o_key_range, sorted_unique=strict and self.o_key_sorted_unique)
            if not strict:
                hip.check(hip.lib().ph_dev_memset(ctx.h, stats, 0, hip.i64(8)))
            orow = lookup(jo, [_raw(hip.PH_I64, c_okey)], m)
            jo.free()
        frees.append(orow)
        misses, multi = (0, 0) if strict else ctx.download(stats, np.int32, 2).tolist()
        if multi:
            raise hip.PlanHipError(hip.PH_EUNSUPPORTED, "Q9: order keys are not unique; use probe_inner")
        amount, nat, n4 = c_amount, c_nat, m
        if misses:   # lineitems without an order leave the result (none in TPC-H)
            okp, n4 = hip.filter_select(ctx, _raw(hip.PH_I32, orow), m, hip.PH_GE, hip.const(hip.PH_I32, i=0))
            frees.append(okp)
            amount, nat = gat(_raw(hip.PH_DEC64, c_amount, 4), okp, n4), gat(_raw(hip.PH_I32, c_nat), okp, n4)
            orow = gat(_raw(hip.PH_I32, orow), okp, n4)
        stage("orders_join", t0)

        t0 = tic()
        year = hip.date_extract(ctx, hip.PH_PART_YEAR, o_date, orow, n4)
        frees.append(year)
        agg = hip.Agg(ctx, [hip.PH_I32, hip.PH_I32], [(hip.PH_A_SUM, 0)], 1024)
        agg.sink([_raw(hip.PH_I32, nat), _raw(hip.PH_I32, year)], [_raw(hip.PH_DEC64, amount, 4)], None, n4,
                 positional=True)
        # the intermediates go back to the (stream-ordered) pool BEFORE the host blocks in the fetch: the
        # bookkeeping runs while the GPU is still busy instead of between two queries
        ctx.free_many(frees)
        r = agg.finalize()
        agg.free()
        stage("expr_aggregate", t0)
        ng, sl = r["ngroups"], r["sum"]
        if dist.world() == 1:   # the groups are final: no merge, no per-group dictionary (30 us of host time between two queries)
            keys = r["keys"]
            rows = list(zip(keys[:ng, 0].tolist(), keys[:ng, 1].tolist(), [x[0] for x in sl]))
            return dict(ngroups=ng, rows=rows, join_rows=n4, timings=t)
        kl, cl = r["keys"].tolist(), r["count"].tolist()
        mine = {(kl[g][0], kl[g][1]): ([sl[g][0]], [cl[g][0]]) for g in range(ng)}
        merged = dist.merge_group_partials(mine, ctx=ctx)
        rows = [(k[0], k[1], v[0][0]) for k, v in merged.items()]
        return dict(ngroups=len(rows), rows=rows, join_rows=n4, timings=t)


def q9_text(rows, nation_names):
    """ORDER BY nation, o_year DESC + the reference's text format (sum_profit at scale 4)."""
    out = ["#\t\t"]
    for nat, year, s in sorted(rows, key=lambda x: (nation_names[x[0]], -x[1])):
        neg = "-" if s < 0 else ""
        w, f = divmod(abs(s), 10000)
        frac = ("%04d" % f).rstrip("0")
        out.append(f"{nation_names[nat]}\t{year}\t{neg}{w}" + (f".{frac}" if frac else ""))
    return "\n".join(out) + "\n"
