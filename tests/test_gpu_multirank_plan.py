"""The N-GPU split INSIDE ph_plan (north_star: "hash-join build-side and group-by hash tables partition by key-hash across
the GPUs ... RCCL all-to-all for the shuffle" — behind the operator interface, not in a Python pipeline). Two ranks = two THREADS of this
process, each with a context of its own on the one GPU of the test box, joined by the in-process transport (ph_comm_init_local: RCCL refuses two
ranks on one device; the collectives' semantics are the RCCL path's). Every rank loads its SHARD of the database (row ranges, as the generator
makes them for any order / row range), creates the same plan, announces the communicator, runs and fetches: every rank must receive the complete
result, and it must equal the reference's golden / the oracle."""
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip, pipelines, tpch, tpchgen

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SF = (1, 1)
NR = 2


def golden(name):
    return open(os.path.join(GOLDEN, name)).read()


def shard(rank, nranks=NR, by_rows=False):
    """rank's share of the SF1 database: orders / lineitem by order ranges (a lineitem row lives with its order), the other tables by row
    ranges; NATION and REGION whole on every rank. by_rows: lineitem and orders split at DIFFERENT order boundaries, so that nothing is co-located"""
    no, nc, ns, np_ = tpchgen.orders_count(SF), 150000, 10000, 200000
    cut = lambda n, r: (n * r // nranks, n * (r + 1) // nranks - n * r // nranks)
    o0, on = cut(no, rank)
    l0, ln = (o0, on) if not by_rows else ((no * rank // nranks + (777 if rank else 0)) if rank else 0, 0)
    if by_rows:   # lineitem's cut shifted against orders': rows of some orders sit on the other rank
        lo_cut = no // nranks + 777
        l0, ln = (0, lo_cut) if rank == 0 else (lo_cut, no - lo_cut)
    c0, cn = cut(nc, rank)
    s0, sn = cut(ns, rank)
    p0, pn = cut(np_, rank)
    return {"sf": SF,
            "lineitem": tpchgen.lineitem(SF, l0, ln),
            "orders": tpchgen.orders(SF, o0, on),
            "customer": tpchgen.customer(SF, c0, cn),
            "supplier": tpchgen.supplier(SF, s0, sn),
            "part": tpchgen.part(SF, p0, pn),
            "partsupp": tpchgen.partsupp(SF, p0, pn)}


def run_ranks(build, render, by_rows=False, broadcast_rows=None):
    """every rank: own ctx, own shard, the same plan; returns [(text, explain)] per rank"""
    group = hip.LocalGroup(NR)
    out, errors = [None] * NR, []

    def worker(rank):
        ctx = db = comm = None
        try:
            ctx = hip.Ctx(0)
            comm = hip.Comm(ctx, group, rank)
            db = tpch.Database(ctx, shard(rank, by_rows=by_rows))
            db.t("nation").set_replicated()
            db.t("region").set_replicated()
            p = build(db)
            p.set_comm(comm, broadcast_rows)
            p.run()
            r = p.fetch_rows() if getattr(p, "rows_root", False) else p.fetch()
            out[rank] = (render(db, p, r), p.explain())
            p.free()
        except Exception as e:   # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))
        finally:
            if db is not None:
                db.free()
            if comm is not None:
                comm.close()
            if ctx is not None:
                ctx.close()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(NR)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    alive = [t for t in ts if t.is_alive()]
    group.free() if not alive else None
    assert not alive, "a rank is stuck in a collective (another rank left the sequence): " + repr(errors)
    assert not errors, errors
    return out


def test_q3_two_ranks_colocated_by_order_ranges():
    """Q3: customer's qualifying keys are broadcast (a small build side), orders SEMI customer runs locally, lineitem x orders is co-located by the
    ranks' key ranges (no exchange), the groups are disjoint by the order key's ranges: per-rank top-k, concatenated"""
    res = run_ranks(lambda db: tpch.q3_plan(db), lambda db, p, r: pipelines.q3_text(tpch.q3_top(r)))
    for text, ex in res:
        assert text == golden("plan_q3.txt"), ex
        assert "broadcast" in ex and "co-located by key range" in ex, ex


def test_q3_two_ranks_hash_partitioned_exchange():
    """... and with lineitem split at other order boundaries than orders (nothing co-located) and the broadcast limit lowered: both sides of the
    big join are hash-partitioned by the order key and exchanged all-to-all inside the plan; the aggregate's input is partitioned by l_orderkey again
    for the top-k's whole groups"""
    res = run_ranks(lambda db: tpch.q3_plan(db), lambda db, p, r: pipelines.q3_text(tpch.q3_top(r)), by_rows=True, broadcast_rows=100_000)
    for text, ex in res:
        assert text == golden("plan_q3.txt"), ex
        assert "exchange (build side of join" in ex and "exchange (probe side of join" in ex, ex


def test_q9_q5_q1_two_ranks_merge_partial_groups():
    """Q9 (five joins, 175 groups), Q5 (six tables) and Q1 (the fused scan) on two ranks: the root aggregate's partial states — exact 128-bit
    sums, counts — are merged at fetch; every rank gets the golden's groups"""
    res = run_ranks(lambda db: tpch.q9_plan(db), lambda db, p, r: pipelines.q9_text(tpch.q9_rows(r), tpchgen.nation_names()), by_rows=True)
    for text, ex in res:
        assert text == golden("plan_q9.txt"), ex
    def q5_text(r):
        dic = tpchgen.nation_names()
        rows = sorted(((int(r["keys"][g][0]), r["sum"][g][0]) for g in range(r["ngroups"])), key=lambda x: -x[1])
        return "#\t\n" + "".join(f"{dic[c]}\t{tpch.dec_text(s, 4)}\n" for c, s in rows)
    res = run_ranks(lambda db: tpch.q5_plan(db), lambda db, p, r: q5_text(r))
    for text, ex in res:
        assert text == golden("plan_q5.txt"), ex
    res = run_ranks(lambda db: tpch.q1_plan(db), lambda db, p, r: sorted((tuple(int(x) for x in r["keys"][g]), tuple(r["sum"][g][:4]), r["count"][g][7]) for g in range(r["ngroups"])))
    import tpch_data
    from plan_amd import queries
    want = O.q1(tpch_data.load(1, 1, q9=False)["lineitem"], queries.q1_shipdate_cutoff())
    exp = sorted(((w.returnflag, w.linestatus), (w.sum_qty.value(), w.sum_base_price.unscaled(2), w.sum_disc_price.unscaled(4), w.sum_charge.unscaled(6)), w.count_order) for w in want)
    for got, ex in res:
        assert got == exp, ex


def test_q18_two_ranks_varchar_key_through_a_broadcast_table():
    """Q18: the subquery's aggregate by l_orderkey is whole per rank (disjoint key ranges), customer is broadcast WITH its VARCHAR c_name (strings
    travel: lengths + bytes), the c_name group key then names rows of the replicated temporary table — the same on both ranks — and the result text
    is the golden's"""
    res = run_ranks(lambda db: tpch.q18_plan(db), lambda db, p, r: q18_text_any_table(db, p, r))
    for text, ex in res:
        assert text == golden("plan_q18.txt"), ex
        assert "VARCHAR" in ex, ex


def q18_text_any_table(db, p, r, limit=100):
    """tpch.q18_text with the c_name strings read from whatever table the plan reports for the key (the broadcast temporary table)"""
    import datetime
    typ, _s, table, col = hip.plan_key_info(p, 0)
    assert typ == hip.PH_STR and table
    rows = [[int(x) for x in r["keys"][g]] + [r["sum"][g][0]] for g in range(r["ngroups"])]

    names = hip.table_strings(db.ctx, table, col, [row[0] for row in rows])
    rows = sorted(zip(names, rows), key=lambda nr: (-nr[1][4], nr[1][3]))[:limit]
    out = ["#\t\t\t\t\t"]
    for name, (_c, ck, ok, od, tp, q) in rows:
        d = datetime.date(1970, 1, 1) + datetime.timedelta(days=od)
        out.append(f"{name}\t{ck}\t{ok}\t{d.isoformat()}\t{tpch.dec_text(tp, 2)}\t{q}")
    return "\n".join(out) + "\n"
