#!/usr/bin/env python3
"""bench.py — throughput of the hot path on MI355X, one process per GPU.

A "step" is one pass of the fused Agg <- Scan(filter) pipeline (TPC-H Q1: filter + 4-group hash
aggregate, 8 aggregates) over the rank's device-resident lineitem shard, plus — for N > 1 — the
merge of the per-rank partial group rows (a few hundred bytes, all-gathered over RCCL).
The data path needs no collective: lineitem is partitioned by order ranges across ranks
(weak scaling: every rank holds an SF`--sf` shard of an SF(sf*N) database).

Prints ONE JSON line on rank 0 (see the driver contract); extra objects:
  roofline     dominant kernel (lowcard_chain_kernel): algorithmic bytes (34 B/row) / its average
               launch duration measured with HIP events on the launch stream
  cpu_baseline the oracle (CPU restatement of the reference path, 1 thread) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q1_BYTES_PER_ROW = 34  # qty 4 + ext/disc/tax 3x8 + flag 1 + status 1 + shipdate 4
Q6_BYTES_PER_ROW = 24
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pmc_traffic(args, world, nrows):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r01_pmc_summary.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command; KiB units, FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md §HBM). Only valid for
    the exact workload those passes ran (SF10, 1 GPU); null otherwise."""
    if world != 1 or args.sf != 10 or nrows != 59986052:
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        kern = "lowcard_chain_kernel" if args.query == "q1" else "filter_sumprod_kernel"
        fetch = [e["avg"] for e in d[f"{args.query} FETCH_SIZE"] if kern in e["kernel"]][0]
        write = [e["avg"] for e in d.get(f"{args.query} WRITE_SIZE", []) if kern in e["kernel"]]
        return fetch * 1024 * 2 + (write[0] * 1024 if write else 0)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sf", type=int, default=10, help="scale factor of each rank's lineitem shard")
    ap.add_argument("--query", default="q1", choices=["q1", "q6", "q3", "q9"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-q3", action="store_true", help="skip the Q3 companion measurement of the default (q1, N=1) run")
    ap.add_argument("--cpu-rows", type=int, default=16_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run",
                  file=sys.stderr)
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from plan_amd import hip, queries, tpchgen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # Rehearsal knob for a one-GPU box: PH_BENCH_BACKEND=gloo puts every rank on device 0 and runs
    # the collectives over gloo on CPU tensors (the real runs use nccl = RCCL, one GPU per rank).
    backend = os.environ.get("PH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.query == "q3":
        return bench_q3(args, rank, local_rank, world)
    if args.query == "q9":
        return bench_q9(args, rank, local_rank, world)

    # ---- this rank's shard: orders [rank*n, (rank+1)*n) of an SF(sf*world) database
    sf_total = (args.sf * world, 1)
    orders_per_rank = tpchgen.orders_count((args.sf, 1))
    t0 = time.time()
    cols = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag",
            "l_linestatus", "l_shipdate"]
    L = tpchgen.lineitem(sf_total, rank * orders_per_rank, orders_per_rank, columns=cols)
    nrows = len(L["l_shipdate"])
    gen_s = time.time() - t0

    # a non-default torch stream, shared with the library so torch.cuda.Event sees its kernels
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = hip.Ctx(local_rank, stream=stream.cuda_stream)
    t0 = time.time()
    table = queries.lineitem_table(ctx, L)
    load_s = time.time() - t0
    if args.query == "q1":
        plan = queries.q1_plan(ctx, table)
        bytes_per_row = Q1_BYTES_PER_ROW
    else:
        plan = queries.q6_plan(ctx, table)
        bytes_per_row = Q6_BYTES_PER_ROW

    # ---- N > 1: every step ends with the cross-rank exchange of the raw partial result (a few
    # hundred bytes, all-gathered on the device over RCCL behind the scan kernels). Two plans
    # (two partial-result buffers) alternate, and the all-gather of step i is asynchronous: it
    # overlaps the scan of step i+1, which writes the OTHER buffer; a buffer's collective is waited
    # for (stream-side) before its plan runs again, and all are drained before the closing barrier.
    # The merged group rows are decoded once after the timed loop, like the N = 1 case fetches its
    # result once after the loop.
    make_plan = (lambda: queries.q1_plan(ctx, table)) if args.query == "q1" else (lambda: queries.q6_plan(ctx, table))
    plans = [plan] + ([make_plan()] if world > 1 else [])
    nbuf = len(plans)
    locals_, gaths, works = [], [], [None] * nbuf
    if world > 1:
        for pl in plans:
            ptr, nwords = pl.partials_dev()

            class _DevView:  # zero-copy torch view of the plan's device result words
                __cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i8", "data": (ptr, False), "version": 2}
            locals_.append(torch.as_tensor(_DevView(), device="cuda"))
            gaths.append(torch.empty(world * nwords, dtype=torch.int64, device=cdev if backend != "nccl" else "cuda"))
    state = {"i": 0}

    def step():
        b = state["i"] % nbuf
        state["i"] += 1
        if works[b] is not None:      # the exchange that read this plan's partials must be done
            works[b].wait()
            works[b] = None
        plans[b].run()
        if world > 1:
            if backend == "nccl":
                if state.get("sync_only"):
                    dist.all_gather_into_tensor(gaths[b], locals_[b])
                else:
                    try:
                        works[b] = dist.all_gather_into_tensor(gaths[b], locals_[b], async_op=True)
                    except (RuntimeError, TypeError):   # every rank fails alike: fall back to the blocking form
                        state["sync_only"] = True
                        dist.all_gather_into_tensor(gaths[b], locals_[b])
            else:  # gloo rehearsal: through host memory
                torch.cuda.synchronize()
                works[b] = dist.all_gather_into_tensor(gaths[b], locals_[b].cpu(), async_op=True)

    def drain():
        for b in range(nbuf):
            if works[b] is not None:
                works[b].wait()
                works[b] = None

    def merged_result():
        if world == 1:
            return plan.fetch()
        drain()
        torch.cuda.synchronize()
        last = (state["i"] - 1) % nbuf
        return plans[last].fetch_merged(gaths[last].cpu().numpy().view(np.uint64), world)

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    result = merged_result()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    result = merged_result()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([nrows], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_rows = int(tot.item())
    else:
        total_rows = nrows

    # ---- roofline: per-launch duration of the scan kernel sequence with HIP events on the
    # launch stream (events bracket one ph_scan_plan_run = scan kernel + the 1-wave merge kernel;
    # no exchange inside the bracket)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    for a, b in ev:
        a.record(stream)
        plan.run()
        b.record(stream)
    torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) for a, b in ev)
    avg_ms = sum(durs) / len(durs)
    achieved = nrows * bytes_per_row / (avg_ms * 1e-3) / 1e9

    # ---- sanity: the result of the timed query must be self-consistent
    ngroups = result["ngroups"]
    rows_out = sum(c[-1] for c in result["count"]) if args.query == "q1" else None

    out = None
    if rank == 0:
        value = total_rows * args.steps / elapsed
        out = {
            "metric": "rows/sec through hash-agg (Q1)" if args.query == "q1" else "rows/sec through filter+SUM (Q6)",
            "value": value,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": f"TPC-H {args.query.upper()} fused filter+hash-aggregate over SF{args.sf} "
                            f"lineitem per GPU ({nrows} rows on rank 0, {total_rows} total), "
                            "table resident in HBM",
                "rows_per_gpu": nrows,
                "kernel_family": plan.kind,
                "groups": ngroups,
                "rows_aggregated": rows_out,
                "parallelism": f"row-range shards x{world}, per-step all-gather of the partial group rows (asynchronous, double-buffered: it overlaps the next step's scan)",
                "generate_s": round(gen_s, 2),
                "pcie_load_s": round(load_s, 2),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args, world, nrows),
                "kernel": "lowcard_chain_kernel" if args.query == "q1" else "filter_sumprod_kernel",
                "avg_launch_ms": avg_ms,
                "min_launch_ms": durs[0],
                "algorithmic_bytes_per_launch": nrows * bytes_per_row,
            },
        }

    # ---- CPU baseline (rank 0, N=1 only): the oracle on a bounded sample of the same rows
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        m = min(nrows, args.cpu_rows)
        sample = {k: v[:m] for k, v in L.items()}
        t0 = time.perf_counter()
        if args.query == "q1":
            O.q1(sample, queries.q1_shipdate_cutoff())
        else:
            O.q6(sample, *queries.q6_constants())
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": m / dt,
            "unit": "rows/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {m} rows of the same lineitem shard, oracle {args.query} pipeline "
                      f"(chunked 2048-row CPU restatement of the reference path), {dt:.1f} s, "
                      f"host has {os.cpu_count()} logical CPUs",
        }
    for pl in plans:
        pl.free()
    table.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    # ---- the metric's second half (hash-join probe, Q3) on the same GPU, N = 1 only: extra fields of
    # the same line, measured after (outside) the Q1 timing; a failure here never costs the Q1 line
    if rank == 0 and world == 1 and args.query == "q1" and not args.no_q3:
        try:
            del L
            a3 = argparse.Namespace(**vars(args))
            a3.query, a3.steps, a3.warmup = "q3", min(args.steps, 30), min(args.warmup, 3)
            q3 = bench_q3(a3, 0, local_rank, 1, emit=False)
            out["q3_single_gpu"] = {"metric": q3["metric"], "value": q3["value"], "unit": q3["unit"],
                                    "ms_per_step": q3["ms_per_step"], "steps": q3["steps"],
                                    "probe_rows_per_s": q3["config"]["probe_rows_per_s"],
                                    "stage_ms": q3["config"]["stage_ms"], "roofline": q3["roofline"]}
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the Q1 line
            out["q3_single_gpu"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out))


def bench_q3(args, rank, local_rank, world, emit=True):
    """Q3: customer |x| orders |x| lineitem hash joins + 3-column group-by, assembled from the
    operator-granular kernels; for N > 1 the join sides are hash-partitioned by order key and
    exchanged with RCCL all-to-all (plan_amd/pipelines.py). A step = one whole Q3."""
    import torch
    import torch.distributed as dist

    from plan_amd import hip, pipelines, tpchgen

    sf_total = (args.sf * world, 1)
    n_ord = tpchgen.orders_count((args.sf, 1))
    n_cust = n_ord // 10
    L = tpchgen.lineitem(sf_total, rank * n_ord, n_ord,
                         columns=["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    Od = tpchgen.orders(sf_total, rank * n_ord, n_ord,
                        columns=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    C = tpchgen.customer(sf_total, rank * n_cust, n_cust)
    nrows = len(L["l_orderkey"])
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)   # the library launches on this stream, so torch events time its kernels
    ctx = hip.Ctx(local_rank, stream=stream.cuda_stream)
    pipe = pipelines.Q3Pipeline(ctx, L, Od, C)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    pipe.time_stages = False   # the measured steps run without a host sync per stage
    for _ in range(args.warmup):
        r = pipe.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = pipe.run()
    barrier()
    elapsed = time.perf_counter() - t0
    # stage times for the report: a few extra steps, outside the timed region, with a sync per stage
    pipe.time_stages = True
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    pipe.probe_events = ev
    probe_dev_ms = 0.0
    agg_t, stage_steps = {}, min(args.steps, 10)
    for _ in range(stage_steps):
        r = pipe.run()
        for k, v in r["timings"].items():
            agg_t[k] = agg_t.get(k, 0) + v
        if "lineitem_filter_probe" in r["timings"]:
            torch.cuda.synchronize()
            probe_dev_ms += ev[0].elapsed_time(ev[1])
    barrier()
    total_rows = nrows
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([nrows], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        total_rows = int(tot.item())
    if rank == 0:
        k = args.steps
        probe_rows = agg_t["probe_rows"] / stage_steps
        fused = "lineitem_filter_probe" in agg_t
        probe_ms = agg_t["lineitem_filter_probe" if fused else "lineitem_probe"] / stage_steps * 1e3
        host_probe_ms = probe_ms
        if fused and probe_dev_ms > 0:
            probe_ms = probe_dev_ms / stage_steps   # HIP events on the launch stream: the stage's kernels only
        pairs = r["join_rows"]
        # probe algorithmic bytes, counted once (BASELINE.md's Q3 row: 16 B per probe row read =
        # selection entry 4 + key 8 + bucket head 4); per output pair next 4 + build key 8 + the
        # pair 8 + its selection entry 4. Implementation passes (candidate slices, the second
        # chain walk of the emit kernel) are NOT counted.
        probe_bytes = probe_rows * 16 + pairs * 24
        if fused:
            # fused Filter -> probe: every lineitem row's l_shipdate (4 B) is read, and the 16 B of
            # a probe only for the rows the filter keeps (counted once, outside the timed loop)
            kept = pipe.lineitem_filter_rows()
            probe_bytes = nrows * 4 + kept * 16 + pairs * 24
        out = {
            "metric": "rows/sec through hash-join probe + hash-agg (Q3)",
            "value": total_rows * k / elapsed, "unit": "rows/s", "n_gpus": world, "steps": k,
            "warmup": args.warmup, "ms_per_step": elapsed / k * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {
                "workload": f"TPC-H Q3 over SF{args.sf} customer/orders/lineitem shards per GPU "
                            f"({nrows} lineitem rows on rank 0), tables resident in HBM",
                "groups_rank0": r["ngroups"], "join_rows_rank0": pairs,
                "parallelism": f"hash-partition by order key x{world}, all-to-all over RCCL" if world > 1 else "single GPU",
                "stage_ms": {kk: round(v / stage_steps * 1e3, 3) for kk, v in agg_t.items() if kk not in ("probe_rows", "exchange_bytes_sent")},
                "stage_ms_note": f"{stage_steps} extra steps after the timed region, one host sync per stage (the timed steps have none)",
                "exchange_bytes_sent_rank0": agg_t.get("exchange_bytes_sent", 0) / stage_steps,
                "probe_rows_per_s": probe_rows / (probe_ms * 1e-3),
                "top1": list(r["top"][0]) if r["top"] else None,
            },
            "roofline": {"bound": "hbm", "achieved": probe_bytes / (probe_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": probe_bytes / (probe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "join_cand_fast_kernel+join_chain_fast_kernel+scan+join_emit_kernel (lineitem Filter+probe stage)",
                         "avg_launch_ms": probe_ms, "timing": "HIP events on the launch stream around the stage" if fused and probe_dev_ms > 0 else "host clock around the stage",
                         "host_timed_stage_ms": host_probe_ms},
        }
        if emit:
            print(json.dumps(out))
    else:
        out = None
    pipe.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return out


def bench_q9(args, rank, local_rank, world):
    """Q9: LIKE + four hash joins (one composite) + profit expression + 175-group aggregate from the
    operator-granular kernels (plan_amd/pipelines.py Q9Pipeline). A step = one whole Q9."""
    import torch
    import torch.distributed as dist

    from plan_amd import hip, pipelines, tpchgen

    sf_total = (args.sf * world, 1)
    n_ord = tpchgen.orders_count((args.sf, 1))
    n_part, n_supp = n_ord * 2 // 15, n_ord // 150
    L = tpchgen.lineitem(sf_total, rank * n_ord, n_ord, columns=["l_orderkey", "l_partkey", "l_suppkey", "l_quantity",
                                                                  "l_extendedprice", "l_discount"])
    Od = tpchgen.orders(sf_total, rank * n_ord, n_ord, columns=["o_orderkey", "o_orderdate"])
    P = tpchgen.part(sf_total, rank * n_part, n_part)
    PS = tpchgen.partsupp(sf_total, rank * n_part, n_part)
    S = tpchgen.supplier(sf_total, rank * n_supp, n_supp)
    nrows = len(L["l_orderkey"])
    ctx = hip.Ctx(local_rank)
    pipe = pipelines.Q9Pipeline(ctx, L, Od, P, PS, S)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    pipe.time_stages = False   # the measured steps run without a host sync per stage
    for _ in range(args.warmup):
        r = pipe.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = pipe.run()
    barrier()
    elapsed = time.perf_counter() - t0
    # stage times for the report: a few extra steps, outside the timed region, with a sync per stage
    pipe.time_stages = True
    agg_t, stage_steps = {}, min(args.steps, 10)
    for _ in range(stage_steps):
        r = pipe.run()
        for k, v in r["timings"].items():
            agg_t[k] = agg_t.get(k, 0) + v
    barrier()
    total_rows = nrows
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([nrows], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        total_rows = int(tot.item())
    if rank == 0:
        k = args.steps
        out = {
            "metric": "rows/sec through 4 hash joins + hash-agg (Q9)", "value": total_rows * k / elapsed,
            "unit": "rows/s", "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"TPC-H Q9 over SF{args.sf} shards per GPU ({nrows} lineitem rows on rank 0), tables resident in HBM",
                       "groups": r["ngroups"], "join_rows_rank0": r["join_rows"],
                       "stage_ms": {kk: round(v / stage_steps * 1e3, 3) for kk, v in agg_t.items() if kk != "exchange_bytes_sent"},
                       "stage_ms_note": f"{stage_steps} extra steps after the timed region, one host sync per stage"},
            "roofline": None,
        }
        print(json.dumps(out))
    pipe.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
