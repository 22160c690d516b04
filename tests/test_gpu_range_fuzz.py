"""Seeded random sweep over the round-3 forms of the partition plans, at the row counts from which they are planned (>= 2^22): one 4-byte
integer key over a dense, strided or offset domain (range partitions and the direct-indexed aggregation where the domain is dense, the
hashed / slot-indexed forms where it is not), value columns that do or do not fit the spare bits of the key word, one and two partition
levels (2e5 .. 4e6 groups), every aggregate -- and the build over the same keys (look-up table or routed ids).  Everything against the oracle."""
import os

import numpy as np
import pytest

import checker as ck
import golden_util as gu

pytestmark = pytest.mark.gpu
VAL_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint32, np.float32, np.float64]
OPS = ["sum", "min", "max", "count", "avg", "var"]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def make_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([4_200_000, 4_700_023, 6_000_011]))
    G = int(rng.choice([200_000, 1_000_000, 4_000_000]))
    kdt = np.int32 if rng.random() < 0.6 else np.uint32
    stride = int(rng.choice([1, 1, 1, 2, 7, 400]))
    lo = int(rng.choice([0, 1, -G // 2, 1_000_000_000])) if kdt == np.int32 else int(rng.choice([0, 5, 2**32 - 1 - G * stride - 10]))
    lo = max(np.iinfo(kdt).min, min(lo, np.iinfo(kdt).max - G * stride))
    key = (lo + rng.integers(0, G, n) * stride).astype(kdt)
    if rng.random() < 0.25: key[rng.integers(1 << 20, n, 3)] = np.iinfo(kdt).max - 3      # keys far outside whatever the first 2^20 rows show
    if rng.random() < 0.2: key[: n // 5] = key[0]                                          # one heavy group
    aggs = []
    for _ in range(int(rng.integers(1, 4))):
        vdt = VAL_DTYPES[rng.integers(len(VAL_DTYPES))]
        op = OPS[rng.integers(len(OPS))]
        if np.dtype(vdt).kind == "f":
            v = np.round(rng.uniform(-100, 100, n), 3).astype(vdt)
        elif vdt in (np.int32, np.uint32) and rng.random() < 0.6:                          # narrow ranges: these travel inside the key word
            w = int(rng.choice([2, 5, 15, 200]))
            base = int(rng.choice([0, 1, -7])) if vdt == np.int32 else int(rng.choice([0, 3]))
            v = (base + rng.integers(0, w, n)).astype(vdt)
            if rng.random() < 0.3: v[n - 11] = base + w + 1000                             # ... until a late row does not fit its field
        else:
            info = np.iinfo(vdt)
            v = rng.integers(max(info.min, -2**31), min(info.max, 2**31 - 1), n, endpoint=True).astype(vdt)
        aggs.append((op, v))
    hint = int(rng.choice([0, G + 100, G * 2]))
    return n, key, aggs, hint


@pytest.mark.parametrize("seed", range(int(os.environ.get("AQG_FUZZ_SEEDS", "24"))))
def test_partition_plans_over_one_integer_key(gpu, oracle, seed):
    n, key, aggs, hint = make_case(int(os.environ.get("AQG_FUZZ_BASE", "3000")) + seed)
    o = oracle.groupby([key])
    ops = [ck.RED_NAMES[op] for op, _ in aggs]
    try:
        gb = gpu.groupby_agg([key], ops, [v for _, v in aggs], hint=hint)
    except Exception as e:                                   # the documented limit: 8 accumulators per call (three VARs over 8-byte integers)
        assert "too many accumulators" in str(e), e
        return
    assert gb.ngroups == o["ngroups"], seed
    assert np.array_equal(gb.first_rows(), o["first_rows"]), seed
    assert np.array_equal(gb.keys(0, key.dtype), key[o["first_rows"]]), seed
    for j, (op, v) in enumerate(aggs):
        got, want = gb.result(j, ops[j], ck.tag_of(v)), oracle.grouped_reduce(ops[j], v, o)
        if v.dtype.kind == "f" and op in ("sum", "avg", "var"):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-6), (seed, j, op)
        else:
            assert gu.same_bits(got, want), (seed, j, op, gb.plan)
    gb.destroy()
    if seed % 3 == 0:                                          # the build over the same key column
        b = gpu.groupby_build([key], hint=hint)
        assert b.ngroups == o["ngroups"]
        assert np.array_equal(b.reversemap(), o["reversemap"]) and np.array_equal(b.counts(), o["counts"]) and np.array_equal(b.first_rows(), o["first_rows"]), (seed, b.plan)
        b.destroy()


def make_wide_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([4_300_003, 5_000_017]))
    nk = int(rng.choice([3, 4, 5, 6]))
    distinct = int(rng.choice([n, n, n // 2, n // 3]))           # all rows distinct (the row map) or a few rows per tuple
    r = rng.permutation(n).astype(np.int64) if distinct == n else rng.integers(0, distinct, n)
    keys, rest = [], r.copy()
    widths = [int(rng.choice([3, 13, 100, 1000, 70_000])) for _ in range(nk - 1)]
    for w in widths:
        lo = int(rng.choice([0, 1, -50, 2_000_000_000]))
        dt = np.int32 if lo < 0 or rng.random() < 0.7 else np.uint32
        keys.append((lo + rest % w).astype(dt))
        rest = rest // w
    keys.append(rest.astype(np.int32))                            # what is left of the tuple number: the tuples stay distinct where r is
    if rng.random() < 0.3: keys[0][rng.integers(1 << 20, n, 2)] += 5_000_000     # values far outside the sampled range of a packed column
    order = rng.permutation(nk)
    keys = [keys[i] for i in order]
    aggs = []
    for _ in range(int(rng.integers(1, 3))):
        vdt = VAL_DTYPES[rng.integers(len(VAL_DTYPES))]
        op = OPS[rng.integers(5)]
        v = np.round(rng.uniform(-100, 100, n), 3).astype(vdt) if np.dtype(vdt).kind == "f" else rng.integers(-100 if np.dtype(vdt).kind == "i" else 0, 100, n).astype(vdt)
        aggs.append((op, v))
    return n, keys, aggs, distinct


@pytest.mark.parametrize("seed", range(int(os.environ.get("AQG_FUZZ_SEEDS", "12"))))
def test_wide_tuples_packed_and_row_map(gpu, oracle, seed):
    """tuples of 3 .. 6 four-byte columns with more than 2^20 groups expected: the wide-tuple plan with the key columns packed under sampled
    ranges (or not, when they do not fit or a row misses its range), all-distinct inputs through the row map"""
    n, keys, aggs, distinct = make_wide_case(int(os.environ.get("AQG_FUZZ_BASE", "5000")) + seed)
    o = oracle.groupby(keys)
    ops = [ck.RED_NAMES[op] for op, _ in aggs]
    gb = gpu.groupby_agg(keys, ops, [v for _, v in aggs], hint=min(n, distinct + 1000))
    assert gb.ngroups == o["ngroups"], seed
    assert np.array_equal(gb.first_rows(), o["first_rows"]), seed
    for k, c in enumerate(keys):
        assert np.array_equal(gb.keys(k, c.dtype), c[o["first_rows"]]), (seed, k)
    for j, (op, v) in enumerate(aggs):
        got, want = gb.result(j, ops[j], ck.tag_of(v)), oracle.grouped_reduce(ops[j], v, o)
        if v.dtype.kind == "f" and op in ("sum", "avg") and o["ngroups"] != n:
            assert np.allclose(got, want, rtol=1e-9, atol=1e-6), (seed, j, op)
        else:
            assert gu.same_bits(got, want), (seed, j, op, gb.plan)
    gb.destroy()
