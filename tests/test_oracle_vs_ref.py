"""Pins the C restatement (oracle/aq_oracle.c) against the REAL reference library
(oracle/_ref/libaqref.so, built from /root/reference by oracle/Makefile) on seeded
random inputs, bit for bit.  Skipped where the reference build is absent."""
import numpy as np
import pytest

import checker as ck

NUM_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]
BIN_DTYPES = [np.int16, np.int32, np.int64, np.uint32, np.float32, np.float64]


def rand(rng, dt, n, small=False):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.uniform(-100, 100, n)).astype(dt)
    hi = 12 if small else min(np.iinfo(dt).max, 20000)
    lo = 1 if (dt.kind == "u" or small) else max(np.iinfo(dt).min, -20000)
    return rng.integers(lo, hi, n, endpoint=True).astype(dt)


def same(a, b):
    return a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("dt", NUM_DTYPES)
def test_type_rules(oracle, ref, dt):
    t = ck.NP2TAG[np.dtype(dt)]
    assert oracle.long_type(t) == ref.long_type(t)
    assert oracle.fp_type(t) == ref.fp_type(t)
    for dt2 in NUM_DTYPES:
        t2 = ck.NP2TAG[np.dtype(dt2)]
        assert oracle.coercion(t, t2) == ref.coercion(t, t2), (dt, dt2)
    for op in range(9):
        assert oracle.reduce_out_dtype(op, t) == ref.reduce_out_dtype(op, t)
    for op in range(12):
        assert oracle.scan_out_dtype(op, t) == ref.scan_out_dtype(op, t)


@pytest.mark.parametrize("dt", NUM_DTYPES)
@pytest.mark.parametrize("n", [0, 1, 2, 7, 1000])
def test_reduce(oracle, ref, dt, n):
    rng = np.random.default_rng(n * 31 + np.dtype(dt).num)
    x = rand(rng, dt, n)
    for name, op in ck.RED_NAMES.items():
        if n == 0 and name in ("avg",):
            continue  # 0/0 = NaN on both sides
        a, b = oracle.reduce(op, x), ref.reduce(op, x)
        assert np.array(a).tobytes() == np.array(b).tobytes(), (name, a, b)


@pytest.mark.parametrize("dt", NUM_DTYPES)
@pytest.mark.parametrize("n", [0, 1, 5, 257])
def test_scan(oracle, ref, dt, n):
    rng = np.random.default_rng(n * 17 + np.dtype(dt).num)
    x = rand(rng, dt, n)
    x[x == 0] = 1
    for name, op in ck.SCAN_NAMES.items():
        if name in ("vars", "stddevs", "varw", "stddevw"):
            continue  # not callable in the reference (printf in loop / out-of-bounds read)
        for w in (0, 1, 2, 3, 10, 256, 257, 1000):
            if w == 0 and name in ("ratiow", "sumw", "avgw"):
                continue  # undefined in the reference (reads ret[-1] / divides by zero)
            a, b = oracle.scan(op, x, w), ref.scan(op, x, w)
            assert same(a, b), (name, w, dt)


@pytest.mark.parametrize("lt", BIN_DTYPES)
@pytest.mark.parametrize("rt", BIN_DTYPES)
def test_ewise_free_ops(oracle, ref, lt, rt):
    rng = np.random.default_rng(np.dtype(lt).num * 100 + np.dtype(rt).num)
    n = 301
    l, r = rand(rng, lt, n), rand(rng, rt, n)
    r[r == 0] = 3
    l[l == 0] = 5
    tl, tr = ck.NP2TAG[np.dtype(lt)], ck.NP2TAG[np.dtype(rt)]
    for op in (ck.OP_ADD, ck.OP_SUB, ck.OP_MUL, ck.OP_DIV, ck.OP_GT):
        assert oracle.ewise_out_dtype(op, tl, tr) == ref.ewise_out_dtype(op, tl, tr), (op, lt, rt)
        assert same(oracle.ewise(op, l, r), ref.ewise(op, l, r)), (op, lt, rt, "vv")
        assert same(oracle.ewise(op, l, r[7]), ref.ewise(op, l, r[7])), (op, lt, rt, "vs")
        assert same(oracle.ewise(op, l[9], r), ref.ewise(op, l[9], r)), (op, lt, rt, "sv")


@pytest.mark.parametrize("lt", BIN_DTYPES)
@pytest.mark.parametrize("rt", BIN_DTYPES)
def test_ewise_aqop(oracle, ref, lt, rt):
    rng = np.random.default_rng(np.dtype(lt).num * 7 + np.dtype(rt).num)
    n = 130
    l, r = rand(rng, lt, n, small=True), rand(rng, rt, n, small=True)
    for op in (ck.OP_LT, ck.OP_GE, ck.OP_LE, ck.OP_EQ, ck.OP_NE):
        assert same(oracle.ewise(op, l, r, ot=ck.BOOL), ref.ewise(op, l, r, ot=ck.BOOL)), (op, lt, rt)
    if np.dtype(lt).kind != "f" and np.dtype(rt).kind != "f":
        for op in (ck.OP_AND, ck.OP_OR, ck.OP_XOR):
            assert same(oracle.ewise(op, l, r, ot=ck.INT32), ref.ewise(op, l, r, ot=ck.INT32)), (op, lt, rt)


@pytest.mark.parametrize("dt", [np.int32, np.int64, np.float32, np.float64, np.uint8])
def test_unary(oracle, ref, dt):
    rng = np.random.default_rng(5)
    x = np.abs(rand(rng, dt, 100))
    assert same(oracle.unary(ck.UN_SQRT, x), ref.unary(ck.UN_SQRT, x))
    if np.dtype(dt).kind == "f":
        for p in (0, 2, 6, 7, 20):
            assert same(oracle.unary(ck.UN_TRUNCATE, x, p), ref.unary(ck.UN_TRUNCATE, x, p)), p


@pytest.mark.parametrize("lt", BIN_DTYPES)
@pytest.mark.parametrize("rt", BIN_DTYPES)
def test_corr(oracle, ref, lt, rt):
    rng = np.random.default_rng(11)
    x, y = rand(rng, lt, 500, small=True), rand(rng, rt, 500, small=True)
    a, b = oracle.corr(x, y), ref.corr(x, y)
    assert np.float64(a).tobytes() == np.float64(b).tobytes(), (a, b)


@pytest.mark.parametrize("dt", NUM_DTYPES)
def test_gather_compact(oracle, ref, dt):
    rng = np.random.default_rng(3)
    x = rand(rng, dt, 400)
    idx = rng.integers(0, 400, 1000).astype(np.uint32)
    assert same(oracle.gather(x, idx), ref.gather(x, idx))
    mask = rng.integers(0, 2, 400).astype(np.uint8)
    assert same(oracle.compact(x, mask), ref.compact(x, mask))
    assert oracle.compact(x, np.zeros(400, np.uint8)).size == 0


def test_hash_kats(oracle, ref):
    # SURVEY 8c a16
    assert oracle.hash_scalar(np.int32(7)) == ref.hash_scalar(np.int32(7)) == 6018027440424182935
    assert oracle.hash_tuple([np.int32(3), np.int32(4)]) == ref.hash_tuple([np.int32(3), np.int32(4)]) == 11708105269577805707
    rng = np.random.default_rng(1)
    for v in rng.integers(-2**31, 2**31 - 1, 50):
        assert oracle.hash_scalar(np.int32(v)) == ref.hash_scalar(np.int32(v))
    for _ in range(20):
        vals = [np.int32(v) for v in rng.integers(-1000, 1000, 6)]
        for k in (1, 2, 3, 6):
            assert oracle.hash_tuple(vals[:k]) == ref.hash_tuple(vals[:k])


GB_CASES = [
    (1, 1000, 10), (1, 5000, 100), (1, 3000, 3000), (2, 4000, 20), (3, 4000, 8), (6, 3000, 3), (1, 1, 1), (2, 17, 2),
]


@pytest.mark.parametrize("nk,n,card", GB_CASES)
def test_groupby(oracle, ref, nk, n, card):
    rng = np.random.default_rng(nk * 1000 + n + card)
    keys = [rng.integers(-card, card, n).astype(np.int32) for _ in range(nk)]
    a, b = oracle.groupby(keys), ref.groupby(keys)
    assert a["ngroups"] == b["ngroups"]
    for f in ("reversemap", "counts", "first_rows", "offsets", "row_ids"):
        assert np.array_equal(a[f], b[f]), f
    for vdt in (np.int32, np.float32, np.int64, np.float64):
        v = rand(rng, vdt, n, small=True)
        for name in ("sum", "min", "max", "count", "avg", "first", "last", "var"):
            x, y = oracle.grouped_reduce(ck.RED_NAMES[name], v, a), ref.grouped_reduce(ck.RED_NAMES[name], v, b)
            assert same(x, y), (name, vdt)


def test_groupby_int64_key(oracle, ref):
    rng = np.random.default_rng(8)
    k = (rng.integers(0, 50, 2000).astype(np.int64) << 33) - 7
    a, b = oracle.groupby([k]), ref.groupby([k])
    for f in ("reversemap", "counts", "first_rows", "offsets", "row_ids"):
        assert np.array_equal(a[f], b[f]), f
