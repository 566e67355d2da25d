"""world_size-2 gloo (CPU) tests of the N>1 path: the partition/all-to-all exchange plumbing and
the partial-group / top-k merges used by bench.py and the Q3 pipeline. The device kernels are
replaced by numpy stand-ins here (same dest = mix64(key) % N rule as ph_partition); what is
under test is the exchange protocol itself, with the send/receive offsets computed by the C ABI's
host routine ph_exchange_layout (the one the RCCL path uses)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plan_amd import dist as pd


def mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xbf58476d1ce4e5b9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94d049bb133111eb)
    x ^= x >> np.uint64(31)
    return x


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        # each rank owns different rows of a probe side and a build side
        n = 5000 + 37 * rank
        keys = rng.integers(0, 3000, n).astype(np.int64)
        vals = rng.integers(0, 10**6, n).astype(np.int64)
        dest = (mix64(keys) % np.uint64(world)).astype(np.int64)
        order = np.argsort(dest, kind="stable")
        counts = [int((dest == d).sum()) for d in range(world)]
        (rk, rv), rc = pd.exchange_columns([keys[order], vals[order]], counts)
        assert len(rk) == len(rv) == sum(rc)
        # the offsets come from the ABI's host routine: this rank's column of the count matrix
        matrix = np.array(pd._gather_objects(counts), dtype=np.int64)
        so, ro = pd.layout(matrix, rank)
        assert so[-1] == n and ro[-1] == len(rk) and [ro[s + 1] - ro[s] for s in range(world)] == rc
        # rows arrive grouped by source rank, each group in the sender's (stable) order
        mine = keys[order][so[rank]:so[rank + 1]]
        assert np.array_equal(rk[ro[rank]:ro[rank + 1]], mine)
        # every received key belongs to this rank's partition
        assert np.all((mix64(rk) % np.uint64(world)).astype(np.int64) == rank)
        # local aggregate on disjoint keys; the union over ranks must equal the global aggregate
        local = {}
        for k, v in zip(rk.tolist(), rv.tolist()):
            s, c = local.get((k,), ([0], [0]))
            local[(k,)] = ([s[0] + v], [c[0] + 1])
        allk = pd.allgather_rows_host(keys)
        allv = pd.allgather_rows_host(vals)
        merged = pd.merge_group_partials(local)
        want = {}
        for k, v in zip(allk.tolist(), allv.tolist()):
            s, c = want.get((k,), ([0], [0]))
            want[(k,)] = ([s[0] + v], [c[0] + 1])
        assert merged == want
        top = pd.merge_topk([(k[0], s[0]) for k, (s, c) in local.items()], 10, key=lambda x: (-x[1], x[0]))
        wtop = sorted([(k[0], s[0]) for k, (s, c) in want.items()], key=lambda x: (-x[1], x[0]))[:10]
        assert top == wtop
        # column statistics over all shards (key ranges for the direct join tables of the N > 1 pipelines)
        assert pd.global_range(None, (10 * rank - 3, 10 * rank + 5)) == (-3, 10 * (world - 1) + 5)
        assert pd.global_range(None, None if rank == 1 else (0, 1)) is None
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_partition_exchange_merge_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_single_process_paths_are_identity():
    cols, rc = pd.exchange_columns([np.arange(5)], [5])
    assert rc == [5] and cols[0].tolist() == [0, 1, 2, 3, 4]
    assert pd.allgather_rows_host(np.arange(3)).tolist() == [0, 1, 2]
    assert pd.merge_topk([(3, 1), (1, 2)], 1, key=lambda x: x[0]) == [(1, 2)]
    assert pd.merge_group_partials({(1,): ([2], [3])}) == {(1,): ([2], [3])}
