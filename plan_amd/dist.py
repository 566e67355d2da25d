"""Multi-GPU plumbing: a thin caller of the exchange entry points of the C ABI
(include/planhip.h: ph_partition_dev, ph_comm_* — RCCL over xGMI, one process per GPU).

The reference has no distributed anything; this is the exchange step SURVEY.md §8(e)
designs for the join queries:

    hash-partition rows by join key (ph_partition_dev, counts stay on the device)
    -> count matrix (ph_comm_exchange_counts: one all-gather + the stage's one host round trip)
    -> all-to-all of every needed column as ONE group of send/recv pairs (ph_comm_exchange_columns)
    -> local build / probe / aggregate on the received rows.

Three interchangeable backends carry the same protocol:
  RcclGroup    the product path: ph_comm over RCCL, device buffers only, stream-ordered
  ThreadGroup  N ranks as N threads of one process sharing one GPU (test double for one-GPU boxes)
  gloo         torch.distributed over gloo (CPU test double; also the one-GPU bench rehearsal)
The two doubles move the rows through host memory with offsets from ph_exchange_layout, the same
host routine the RCCL path uses, so the offset arithmetic under test is the shipped one.
"""
import ctypes
import threading

import numpy as np

from . import hip

_tls = threading.local()
_proc = {"group": None}   # process-wide RcclGroup (one ctx per process in the real runs)


# ---------------------------------------------------------------- backends

class RcclGroup:
    """ph_comm over RCCL. `id_bytes`: the 128-byte id from unique_id() of rank 0."""

    def __init__(self, ctx, nranks, rank, id_bytes):
        self.ctx, self.n, self.rank = ctx, nranks, rank
        self.h = hip.vp()
        buf = ctypes.create_string_buffer(bytes(id_bytes), hip.PH_COMM_ID_BYTES)
        hip.check(hip.lib().ph_comm_init(ctx.h, hip.i32(nranks), hip.i32(rank), buf, ctypes.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(hip.PH_COMM_ID_BYTES)
        hip.check(hip.lib().ph_comm_unique_id(buf))
        return buf.raw

    def bind(self):
        _proc["group"] = self

    def close(self):
        if self.h:
            hip.lib().ph_comm_destroy(self.h)
            self.h = None
        if _proc["group"] is self:
            _proc["group"] = None

    # -- thin wrappers
    def allgather(self, send_ptr, recv_ptr, nbytes, async_=False):
        hip.check(hip.lib().ph_comm_allgather(self.h, hip.vp(_addr(send_ptr)), hip.vp(_addr(recv_ptr)), hip.i64(nbytes),
                                              hip.i32(1 if async_ else 0)))

    def wait(self, keep=0):
        hip.check(hip.lib().ph_comm_wait_keep(self.h, hip.i32(keep)))

    def allreduce(self, vals, op="sum"):
        a = (hip.i64 * len(vals))(*[int(v) for v in vals])
        code = {"sum": hip.PH_RED_SUM, "max": hip.PH_RED_MAX, "min": hip.PH_RED_MIN}[op]
        hip.check(hip.lib().ph_comm_allreduce_i64(self.h, a, hip.i32(len(vals)), hip.i32(code)))
        return [a[i] for i in range(len(vals))]

    def barrier(self):
        hip.check(hip.lib().ph_comm_barrier(self.h))

    def exchange_counts(self, counts_dev):
        m = (hip.i64 * (self.n * self.n))()
        hip.check(hip.lib().ph_comm_exchange_counts(self.h, counts_dev, m))
        return np.frombuffer(m, dtype=np.int64).reshape(self.n, self.n).copy()

    def exchange_columns(self, send_ptrs, recv_ptrs, widths, matrix):
        k = len(send_ptrs)
        sp = (hip.vp * k)(*[hip.vp(_addr(p)) for p in send_ptrs])
        rp = (hip.vp * k)(*[hip.vp(_addr(p)) for p in recv_ptrs])
        wd = (hip.i32 * k)(*widths)
        m = np.ascontiguousarray(matrix, dtype=np.int64)
        hip.check(hip.lib().ph_comm_exchange_columns(self.h, hip.i32(k), sp, rp, wd, hip.vp(m.ctypes.data)))

    def allgather_rows(self, send_ptr, count, width):
        """ph_comm_allgather_rows_alloc: the library sizes the output from the all-gathered counts (no rank can
        leave between the count exchange and the send/recv group). Returns (device pointer, counts)."""
        counts = (hip.i64 * self.n)()
        out = hip.vp()
        hip.check(hip.lib().ph_comm_allgather_rows_alloc(self.h, hip.vp(_addr(send_ptr)), hip.i64(count), hip.i32(width),
                                                         ctypes.byref(out), counts))
        return out, [counts[r] for r in range(self.n)]


class ThreadGroup:
    """N ranks as N threads of ONE process sharing one GPU — a test double for the process group,
    so the partitioned pipelines can be exercised end to end on a single-GPU box. Collectives are
    a barrier plus reads of the other ranks' published host copies."""

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


def _addr(p):
    if p is None:
        return None
    return p.value if isinstance(p, ctypes.c_void_p) else int(p)


def _tg():
    return getattr(_tls, "group", None)


def _rccl():
    return _proc["group"]


def _td():
    import sys
    td = sys.modules.get("torch.distributed")   # never imported from here: a process that has not
    return td if td is not None and td.is_available() and td.is_initialized() else None   # set it up is one rank


def world():
    if _tg() is not None:
        return _tg().n
    if _rccl() is not None:
        return _rccl().n
    td = _td()
    return td.get_world_size() if td else 1


def rank():
    if _tg() is not None:
        return _tls.rank
    if _rccl() is not None:
        return _rccl().rank
    td = _td()
    return td.get_rank() if td else 0


def init_rccl(ctx):
    """Create and bind the process-wide RcclGroup; the id travels over the already initialised
    torch.distributed group (any backend: it is a 128-byte host object)."""
    td = _td()
    if td is None:
        raise RuntimeError("init_rccl needs torch.distributed initialised to share the communicator id")
    box = [RcclGroup.unique_id() if td.get_rank() == 0 else None]
    td.broadcast_object_list(box, src=0)
    g = RcclGroup(ctx, td.get_world_size(), td.get_rank(), box[0])
    g.bind()
    return g


def layout(matrix, me):
    """send/recv offsets of every peer from the count matrix — ph_exchange_layout (host-only ABI)"""
    m = np.ascontiguousarray(matrix, dtype=np.int64)
    n = m.shape[0]
    so, ro = (hip.i64 * (n + 1))(), (hip.i64 * (n + 1))()
    hip.check(hip.lib().ph_exchange_layout(hip.vp(m.ctypes.data), hip.i32(n), hip.i32(me), so, ro))
    return [so[i] for i in range(n + 1)], [ro[i] for i in range(n + 1)]


# ---------------------------------------------------------------- host-memory doubles

def _gather_objects(obj):
    """every rank's obj, in rank order (ThreadGroup or torch.distributed object gather)"""
    if _tg() is not None:
        return _tg().share(rank(), obj)
    td = _td()
    out = [None] * td.get_world_size()
    td.all_gather_object(out, obj)
    return out


def _count_matrix_host(send_counts):
    return np.array(_gather_objects([int(c) for c in send_counts]), dtype=np.int64)


def _exchange_host(cols, matrix):
    """cols: numpy arrays ordered by destination. Returns the received arrays (source-rank order)."""
    n, me = world(), rank()
    so, ro = layout(matrix, me)
    if _tg() is not None:
        parts = _tg().share(me, cols)
        res = []
        for ci in range(len(cols)):
            pieces = []
            for s in range(n):
                s_off, _ = layout(matrix, s)
                pieces.append(parts[s][ci][s_off[me]:s_off[me + 1]])
            res.append(np.concatenate(pieces) if pieces else cols[ci][:0])
        _tg().barrier.wait()   # nobody drops its send buffers before everyone has copied
        return res
    import torch
    td = _td()
    out = []
    for c in cols:
        c = np.ascontiguousarray(c)
        r = torch.empty(ro[n], dtype=torch.from_numpy(c[:0]).dtype)
        td.all_to_all_single(r, torch.from_numpy(c), [ro[s + 1] - ro[s] for s in range(n)],
                             [so[d + 1] - so[d] for d in range(n)])
        out.append(r.numpy())
    return out


# ---------------------------------------------------------------- the protocol

def exchange_columns(columns, send_counts):
    """HOST form (numpy arrays ordered by destination rank, rows of dest 0 first): the protocol the
    CPU tests drive. Returns (received arrays, recv_counts)."""
    n = world()
    if n == 1:
        return list(columns), [int(c) for c in send_counts]
    matrix = _count_matrix_host(send_counts)
    recv = _exchange_host([np.ascontiguousarray(c) for c in columns], matrix)
    return recv, [int(matrix[s][rank()]) for s in range(n)]


def exchange(ctx, cols, counts_dev, n_rows):
    """DEVICE form. cols: list of (device pointer, numpy dtype) with n_rows rows ordered by
    destination (ph_partition_dev's permutation applied); counts_dev: that call's device counts.
    Returns (received device pointers — ctx.alloc'd, the caller frees —, rows received, rows this
    rank sent to OTHER ranks)."""
    n, me = world(), rank()
    g = _rccl()
    widths = [np.dtype(dt).itemsize for _, dt in cols]
    if g is not None and _tg() is None:
        matrix = g.exchange_counts(counts_dev)
        _, ro = layout(matrix, me)
        total = ro[n]
        recv = [ctx.alloc(max(total, 1) * w) for w in widths]
        g.exchange_columns([p for p, _ in cols], recv, widths, matrix)
        return recv, total, int(matrix[me].sum() - matrix[me][me])
    counts = ctx.download(counts_dev, np.int64, n)
    matrix = _count_matrix_host(counts)
    host = [ctx.download(p, dt, n_rows) if n_rows else np.empty(0, dt) for p, dt in cols]
    got = _exchange_host(host, matrix)
    total = len(got[0]) if got else 0
    recv = [ctx.upload(a) if len(a) else ctx.alloc(8) for a in got]
    return recv, total, int(matrix[me].sum() - matrix[me][me])


def allgather_rows(ctx, ptr, count, dtype):
    """Variable-length all-gather of one device column (broadcast of a small build side).
    Returns (device pointer — ctx.alloc'd —, total rows)."""
    n = world()
    w = np.dtype(dtype).itemsize
    g = _rccl()
    if g is not None and _tg() is None:
        out, counts = g.allgather_rows(ptr, count, w)
        return out, sum(counts)
    mine = ctx.download(ptr, dtype, count) if count else np.empty(0, dtype)
    allv = np.concatenate(_gather_objects(mine))
    return (ctx.upload(allv) if len(allv) else ctx.alloc(8)), len(allv)


def allgather_rows_host(column):
    """HOST form of allgather_rows (numpy)."""
    if world() == 1:
        return column
    return np.concatenate(_gather_objects(np.ascontiguousarray(column)))


def allgather_records(ctx, records):
    """records: int64 numpy matrix [rows, width] of this rank (small merges: partial groups, top-k
    candidates). Returns all ranks' records concatenated in rank order. ctx None = host form."""
    records = np.ascontiguousarray(records, dtype=np.int64)
    if world() == 1:
        return records
    width = records.shape[1]
    g = _rccl()
    if ctx is not None and g is not None and _tg() is None:
        flat = records.reshape(-1)
        src = ctx.upload(flat) if len(flat) else ctx.alloc(8)
        out, total = allgather_rows(ctx, src, len(flat), np.int64)
        res = ctx.download(out, np.int64, total).reshape(-1, width) if total else records[:0]
        ctx.free(src)
        ctx.free(out)
        return res
    return np.concatenate(_gather_objects(records)).reshape(-1, width)


def agree_max(ctx, value):
    """max over all ranks of a small integer (an error flag): every rank calls it at the same point"""
    if world() == 1:
        return int(value)
    g = _rccl()
    if g is not None and _tg() is None:
        return int(g.allreduce([int(value)], "max")[0])
    return int(max(_gather_objects(int(value))))


def global_range(ctx, lo_hi):
    """(min, max) over all ranks of a column's value range — column statistics of a table whose rows
    are sharded over the ranks (computed once per table, not per query). None when any rank has none."""
    if world() == 1:
        return lo_hi
    rec = np.array([[1, lo_hi[0], lo_hi[1]]] if lo_hi is not None else [[0, 0, 0]], dtype=np.int64)
    allr = allgather_records(ctx, rec)
    if not np.all(allr[:, 0] == 1):
        return None
    return int(allr[:, 1].min()), int(allr[:, 2].max())


def merge_group_partials(groups, ctx=None):
    """groups: {key tuple of ints: (sums list of python ints < 2^127, counts list)} of this rank.
    Returns the merged dict on every rank (Q1/Q9-style tiny merges). Records travel as int64 words:
    keys, then (lo, hi) of every sum, then the counts."""
    if world() == 1:
        return dict(groups)
    nk = na = 0
    for k, (s, c) in groups.items():
        nk, na = len(k), len(s)
        break
    shape = allgather_records(ctx, np.array([[nk, na]], dtype=np.int64))
    nk, na = int(shape[:, 0].max()), int(shape[:, 1].max())
    rec = np.zeros((len(groups), nk + 3 * na), dtype=np.int64)
    M = (1 << 64) - 1
    for i, (k, (s, c)) in enumerate(groups.items()):
        rec[i, :nk] = k
        for a in range(na):
            lo = s[a] & M
            rec[i, nk + 2 * a] = lo - (1 << 64) if lo >= (1 << 63) else lo
            rec[i, nk + 2 * a + 1] = s[a] >> 64
            rec[i, nk + 2 * na + a] = c[a]
    allr = allgather_records(ctx, rec)
    merged = {}
    for row in allr.tolist():
        k = tuple(row[:nk])
        ms, mc = merged.setdefault(k, ([0] * na, [0] * na))
        for a in range(na):
            ms[a] += (row[nk + 2 * a] & M) + (row[nk + 2 * a + 1] << 64)
            mc[a] += row[nk + 2 * na + a]
    return merged


def merge_topk(rows, k, key, ctx=None):
    """rows: this rank's candidate rows (tuples of ints); the global top-k under `key`, on every rank."""
    mine = sorted(rows, key=key)[:k]
    if world() == 1:
        return mine
    width = len(mine[0]) if mine else 0
    w = int(allgather_records(ctx, np.array([[width]], dtype=np.int64)).max())
    rec = np.array(mine, dtype=np.int64).reshape(len(mine), w) if w else np.zeros((0, 0), np.int64)
    allr = allgather_records(ctx, rec) if w else rec
    return sorted([tuple(r) for r in allr.tolist()], key=key)[:k]
