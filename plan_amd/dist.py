"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests). The reference has no distributed anything;
this is the exchange step SURVEY.md §8(e) designs for the join queries:

    hash-partition rows by join key -> exchange the per-destination counts -> all-to-all of every
    needed column (variable split sizes) -> local build / probe / aggregate on the received rows.

xGMI is point-to-point (7 links per GPU), so one balanced all-to-all keeps every link busy at once;
columns are exchanged as separate contiguous buffers (no row packing) so the receiving kernels read
them exactly like resident table columns.
"""
import threading

import torch
import torch.distributed as dist

_tls = threading.local()


class ThreadGroup:
    """N ranks as N threads of ONE process sharing one GPU — a test double for the process group,
    so the partitioned pipelines can be exercised end to end on a single-GPU box. Collectives are
    a barrier plus reads of the other ranks' published tensors (same device, so plain copies)."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n

    def bind(self, rank):
        _tls.group, _tls.rank = self, rank

    def share(self, rank, obj):
        """all ranks publish obj; returns the list of everyone's objects"""
        self.barrier.wait()          # previous round fully consumed
        self.slots[rank] = obj
        self.barrier.wait()
        out = list(self.slots)
        self.barrier.wait()
        return out


def _tg():
    return getattr(_tls, "group", None)


def world():
    if _tg() is not None:
        return _tg().n
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    if _tg() is not None:
        return _tls.rank
    return dist.get_rank() if dist.is_initialized() else 0


def exchange_counts(send_counts, device):
    """send_counts[d] rows go to rank d. Returns recv_counts[s] = rows arriving from rank s."""
    n = world()
    if n == 1:
        return list(send_counts)
    if _tg() is not None:
        allc = _tg().share(rank(), list(send_counts))
        return [int(allc[s][rank()]) for s in range(n)]
    inp = torch.tensor(list(send_counts), dtype=torch.int64, device=device)
    out = torch.empty(n, dtype=torch.int64, device=device)
    dist.all_to_all_single(out, inp)
    return [int(x) for x in out.tolist()]


def exchange_columns(columns, send_counts, recv_counts=None):
    """columns: list of 1-D tensors already ordered by destination rank (rows of dest 0 first),
    all with sum(send_counts) rows. Returns (received columns, recv_counts)."""
    n = world()
    dev = columns[0].device if columns else torch.device("cpu")
    if recv_counts is None:
        recv_counts = exchange_counts(send_counts, dev)
    if n == 1:
        return list(columns), recv_counts
    if _tg() is not None:
        me = rank()
        offs = [0]
        for c in send_counts:
            offs.append(offs[-1] + int(c))
        parts = _tg().share(me, (columns, offs))
        res = [torch.cat([parts[s][0][ci][parts[s][1][me]:parts[s][1][me + 1]] for s in range(n)])
               for ci in range(len(columns))]
        _tg().barrier.wait()   # nobody frees its send buffers before everyone has copied
        return res, recv_counts
    total = int(sum(recv_counts))
    out = []
    for c in columns:
        r = torch.empty(total, dtype=c.dtype, device=c.device)
        dist.all_to_all_single(r, c.contiguous(), list(recv_counts), list(send_counts))
        out.append(r)
    return out, recv_counts


def allgather_rows(column):
    """Variable-length all-gather of one column (broadcast of a small build side)."""
    n = world()
    if n == 1:
        return column
    if _tg() is not None:
        res = torch.cat(_tg().share(rank(), column))
        _tg().barrier.wait()
        return res
    cnt = torch.tensor([column.numel()], dtype=torch.int64, device=column.device)
    cnts = [torch.empty(1, dtype=torch.int64, device=column.device) for _ in range(n)]
    dist.all_gather(cnts, cnt)
    sizes = [int(c.item()) for c in cnts]
    m = max(sizes) if sizes else 0
    pad = torch.zeros(m, dtype=column.dtype, device=column.device)
    pad[:column.numel()] = column
    parts = [torch.empty(m, dtype=column.dtype, device=column.device) for _ in range(n)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def merge_group_partials(groups):
    """groups: {key tuple: (sums list, counts list)} of this rank. Returns the merged dict on
    every rank (Q1/Q6-style tiny merges: a few hundred bytes, so an object all-gather is fine)."""
    n = world()
    if n == 1:
        return dict(groups)
    if _tg() is not None:
        gathered = _tg().share(rank(), groups)
    else:
        gathered = [None] * n
        dist.all_gather_object(gathered, groups)
    merged = {}
    for part in gathered:
        for k, (s, c) in part.items():
            if k not in merged:
                merged[k] = ([0] * len(s), [0] * len(c))
            ms, mc = merged[k]
            for a in range(len(s)):
                ms[a] += s[a]
                mc[a] += c[a]
    return merged


def merge_topk(rows, k, key):
    """rows: this rank's candidate rows; returns the global top-k under `key` on every rank."""
    n = world()
    if n == 1:
        return sorted(rows, key=key)[:k]
    if _tg() is not None:
        gathered = _tg().share(rank(), sorted(rows, key=key)[:k])
    else:
        gathered = [None] * n
        dist.all_gather_object(gathered, sorted(rows, key=key)[:k])
    allrows = [r for part in gathered for r in part]
    return sorted(allrows, key=key)[:k]
