"""Seeded random shapes through the aggregate and the join against the oracle: key counts and
types, NULLs, selections, aggregate mixes, cardinalities and sizes on both sides of the thresholds
that switch kernels (LDS pre-aggregation / bulk build at 64 K rows with a high hint; atomic /
partitioned join build at 128 K rows; Bloom bitmap up to 4 M keys; one- and two-key fast kernels)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from plan_amd import hip
from test_gpu_ops import agg_compare, join_compare, rnd_validity

pytestmark = pytest.mark.gpu
SEED_BASE = int(os.environ.get("PH_TEST_SEED_BASE", "0"))   # soak runs: other seeds through the same shapes


@pytest.fixture(scope="module")
def ctx():
    c = hip.Ctx(0)
    yield c
    c.close()


KEY_TYPES = [(hip.PH_I32, O.OT_INT32, np.int32), (hip.PH_I64, O.OT_INT64, np.int64), (hip.PH_DATE, O.OT_DATE, np.int32)]


@pytest.mark.parametrize("seed", range(12))
def test_random_aggregates(ctx, seed):
    rng = np.random.default_rng(1000 + SEED_BASE + seed)
    n = int(rng.choice([300, 5_000, 70_000, 150_000, 400_000]))
    nk = int(rng.integers(1, 4))
    card = int(rng.choice([1, 3, 40, 2_000, 60_000]))
    per_key = max(1, int(round(card ** (1.0 / nk))))
    keys = []
    for _ in range(nk):
        ht, ot, dt = KEY_TYPES[int(rng.integers(0, len(KEY_TYPES)))]
        vals = rng.integers(-per_key // 2, per_key // 2 + 1, n).astype(dt)
        v = rnd_validity(rng, n, 0.03)[0] if rng.random() < 0.4 else None
        keys.append((ht, ot, vals, 0, v))
    dec = rng.integers(-10**7, 10**7, n).astype(np.int64)
    qty = rng.integers(1, 51, n).astype(np.int32)
    vd = rnd_validity(rng, n, 0.1)[0] if rng.random() < 0.5 else None
    args = [(hip.PH_DEC64, O.OT_DECIMAL, dec, 2, vd), (hip.PH_I32, O.OT_INT32, qty, 0, None)]
    pool = [(hip.PH_A_SUM, 0), (hip.PH_A_SUM, 1), (hip.PH_A_AVG, 0), (hip.PH_A_AVG, 1), (hip.PH_A_COUNT, 0),
            (hip.PH_A_MIN, 0), (hip.PH_A_MAX, 0), (hip.PH_A_COUNT_STAR, -1)]
    aggs = [pool[i] for i in rng.choice(len(pool), int(rng.integers(1, 5)), replace=False)]
    sel = np.sort(rng.choice(n, int(n * 0.6), replace=False)) if rng.random() < 0.4 else None
    expected = int(rng.choice([16, card, 4 * card + 40_000]))   # low, right and high (bulk build) hints
    r = agg_compare(ctx, keys, args, aggs, n, sel=sel, expected=expected)
    assert r["ngroups"] >= 1


@pytest.mark.parametrize("seed", range(10))
def test_random_joins(ctx, seed):
    rng = np.random.default_rng(2000 + SEED_BASE + seed)
    nb = int(rng.choice([50, 20_000, 140_000, 300_000]))
    np_ = int(rng.choice([1_000, 90_000, 500_000]))
    nk = int(rng.integers(1, 3))
    ht, ot, dt = KEY_TYPES[int(rng.integers(0, 2))]
    dom = max(2, int(nb * float(rng.choice([0.3, 1.0, 3.0])))) if nk == 1 else max(2, int(np.sqrt(nb * 2)))
    with_nulls = rng.random() < 0.4
    b, p = [], []
    for c in range(nk):
        bv = rng.integers(0, dom, nb).astype(dt)
        pv = rng.integers(0, int(dom * 1.3) + 1, np_).astype(dt)
        b.append((ht, ot, bv, rnd_validity(rng, nb, 0.05)[0] if with_nulls and c == 0 else None))
        p.append((ht, ot, pv, rnd_validity(rng, np_, 0.05)[0] if with_nulls and c == nk - 1 else None))
    bsel = np.sort(rng.choice(nb, max(1, nb * 3 // 4), replace=False)) if rng.random() < 0.5 else None
    psel = np.sort(rng.choice(np_, max(1, np_ // 2), replace=False)) if rng.random() < 0.5 else None
    join_compare(ctx, b, p, bsel, psel)


def test_error_codes_and_capacity_reports(ctx):
    """Every entry point answers a shape it does not run, or a buffer that is too small, with an
    error code (never a different result): join output capacity, top-k room, scatter value type,
    key-type mismatch, too many keys."""
    rng = np.random.default_rng(77)
    b = hip.DevColumn(ctx, hip.PH_I32, rng.integers(0, 100, 5000).astype(np.int32))       # ~50 duplicates per key
    p = hip.DevColumn(ctx, hip.PH_I32, rng.integers(0, 100, 4000).astype(np.int32))
    j = hip.Join(ctx, [b], None, 5000)
    with pytest.raises(hip.PlanHipError) as e:
        j.probe_inner([p], None, 4000, 1000)          # ~200 000 pairs do not fit 1000
    assert e.value.code == hip.PH_ECAPACITY
    m, op, ob = j.probe_inner([p], None, 4000, 400_000)
    assert m > 150_000
    ctx.free(op); ctx.free(ob)
    p64 = hip.DevColumn(ctx, hip.PH_I64, np.arange(10, dtype=np.int64))
    with pytest.raises(hip.PlanHipError) as e:
        j.probe_inner([p64], None, 10, 10)            # 64-bit probe key against a 32-bit build key
    assert e.value.code == hip.PH_EINVAL
    j.free()
    with pytest.raises(hip.PlanHipError):
        hip.Join(ctx, [b, b, b, b, b], None, 5000)    # more than 4 key columns
    # top-k: more qualifying groups than the caller has room for
    k = hip.DevColumn(ctx, hip.PH_I32, np.arange(3000, dtype=np.int32))
    v = hip.DevColumn(ctx, hip.PH_I64, np.ones(3000, np.int64))                            # 3000 groups tie
    agg = hip.Agg(ctx, [hip.PH_I32], [(hip.PH_A_SUM, 0)], 4096)
    agg.sink([k], [v], None, 3000)
    with pytest.raises(hip.PlanHipError) as e:
        agg.topk(0, 10, cap=100)
    assert e.value.code == hip.PH_ECAPACITY
    assert len(agg.topk(0, 10, cap=4096)["first_row"]) == 3000                             # all tie for the top
    agg.free()
    # FillSwitch fills INTEGER and DECIMAL results only
    c = hip.Col()
    c.type, c.data = hip.PH_I64, v.data
    out = ctx.alloc(3000 * 8)
    with pytest.raises(hip.PlanHipError) as e:
        hip.scatter(ctx, c, None, 3000, out)
    assert e.value.code == hip.PH_EUNSUPPORTED
    ctx.free(out)
    for d in (b, p, p64, k, v):
        d.free()
