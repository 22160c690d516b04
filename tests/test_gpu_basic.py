"""GPU parity, part 1: generators, reductions, element-wise ops, fused group-by -- HIP path (through the
C-ABI) vs the oracle on the same seeded inputs, and vs the golden vectors dumped from the reference."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu

pytestmark = pytest.mark.gpu

NUM_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]
BIN_DTYPES = [np.int16, np.int32, np.int64, np.uint32, np.float32, np.float64]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def rand(rng, dt, n, small=False):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return np.round(rng.uniform(-100, 100, n), 6).astype(dt)
    hi = 12 if small else min(np.iinfo(dt).max, 20000)
    lo = 1 if (dt.kind == "u" or small) else max(np.iinfo(dt).min, -20000)
    return rng.integers(lo, hi, n, endpoint=True).astype(dt)


def test_generators_match_oracle(gpu, oracle):
    for col in range(11):
        for base, n in ((0, 1000), (123456789, 4099), (0, 3)):
            a = gpu.gen_column(col, 42, base, n, 10**9, 100).to_host()
            b = oracle.gen_column(col, 42, base, n, 10**9, 100)
            assert a.tobytes() == b.tobytes(), (col, base, n)


@pytest.mark.parametrize("dt", NUM_DTYPES)
@pytest.mark.parametrize("n", [0, 1, 3, 1000, 100003])
def test_reduce(gpu, oracle, dt, n):
    rng = np.random.default_rng(n + np.dtype(dt).num)
    x = rand(rng, dt, n)
    for name, op in ck.RED_NAMES.items():
        if n == 0 and name == "avg":
            continue
        a, b = gpu.reduce(op, x), oracle.reduce(op, x)
        if np.dtype(dt).kind == "f" and name in ("sum", "avg", "var", "stddev") and n > 1:
            # floating sums: tree order on device, sequential in the reference -> bound (n-1) * 2^-53 * sum|x|
            tol = max(1.0, float(np.sum(np.abs(x.astype(np.float64)) ** (2 if name in ("var", "stddev") else 1)))) * n * 2.0 ** -52
            assert abs(float(a) - float(b)) <= tol * (1 if name != "stddev" else 1e3), (name, a, b)
        else:
            assert np.array(a).tobytes() == np.array(b).tobytes(), (name, dt, n, a, b)


def test_reduce_golden(gpu):
    for c in [c for c in gu.load() if c["fn"] == "reduce"]:
        x = gu.dec(c["x"])
        if x.dtype.kind == "f" and c["op"] in ("sum", "avg", "var", "stddev") and x.size > 2:
            continue
        if x.size == 0 and c["op"] == "avg":
            continue
        got = gpu.reduce(ck.RED_NAMES[c["op"]], x)
        assert gu.scalar_same(got, gu.dec_scalar(c["out"])), (c["op"], c["x"]["dtype"], c["src"])


@pytest.mark.parametrize("lt", BIN_DTYPES)
@pytest.mark.parametrize("rt", BIN_DTYPES)
def test_ewise(gpu, oracle, lt, rt):
    rng = np.random.default_rng(np.dtype(lt).num * 100 + np.dtype(rt).num)
    for n in (5, 1003):
        l, r = rand(rng, lt, n), rand(rng, rt, n)
        r[r == 0] = 3
        l[l == 0] = 5
        for op in (ck.OP_ADD, ck.OP_SUB, ck.OP_MUL, ck.OP_DIV, ck.OP_GT):
            assert gu.same_bits(gpu.ewise(op, l, r), oracle.ewise(op, l, r)), (op, lt, rt, "vv")
            assert gu.same_bits(gpu.ewise(op, l, r[2]), oracle.ewise(op, l, r[2])), (op, lt, rt, "vs")
            assert gu.same_bits(gpu.ewise(op, l[1], r), oracle.ewise(op, l[1], r)), (op, lt, rt, "sv")
        for op in (ck.OP_LT, ck.OP_GE, ck.OP_LE, ck.OP_EQ, ck.OP_NE):
            assert gu.same_bits(gpu.ewise(op, l, r, ot=ck.BOOL), oracle.ewise(op, l, r, ot=ck.BOOL)), (op, lt, rt)
        if np.dtype(lt).kind != "f" and np.dtype(rt).kind != "f":
            for op in (ck.OP_AND, ck.OP_OR, ck.OP_XOR, ck.OP_MOD):
                assert gu.same_bits(gpu.ewise(op, l, r, ot=ck.INT32), oracle.ewise(op, l, r, ot=ck.INT32)), (op, lt, rt)


def test_ewise_golden(gpu):
    name2tag = {"bool": ck.BOOL}
    for c in [c for c in gu.load() if c["fn"] == "ewise"]:
        ot = name2tag[c["ot"]] if c["ot"] else None
        got = gpu.ewise(ck.OP_NAMES[c["op"]], gu.operand(c, "l"), gu.operand(c, "r"), ot=ot)
        assert gu.same_bits(got, gu.dec(c["out"])), (c["op"], c["l"]["dtype"], c["r"]["dtype"], c["src"])


def test_unary(gpu, oracle):
    rng = np.random.default_rng(5)
    for dt in (np.int32, np.int64, np.float32, np.float64, np.uint8):
        x = np.abs(rand(rng, dt, 1000))
        assert gu.same_bits(gpu.unary(ck.UN_SQRT, x), oracle.unary(ck.UN_SQRT, x))
        if np.dtype(dt).kind == "f":
            for p in (0, 2, 6, 7, 20):
                assert gu.same_bits(gpu.unary(ck.UN_TRUNCATE, x, p), oracle.unary(ck.UN_TRUNCATE, x, p)), (dt, p)


def test_corr(gpu, oracle):
    rng = np.random.default_rng(11)
    for lt in (np.int16, np.int32, np.int64, np.uint32):
        for rt in (np.int16, np.int32, np.uint32):
            x, y = rand(rng, lt, 5000, small=True), rand(rng, rt, 5000, small=True)
            a, b = gpu.corr(x, y), oracle.corr(x, y)
            assert np.float64(a).tobytes() == np.float64(b).tobytes(), (lt, rt, a, b)


def check_agg(gpu, oracle, keys, vals, hint):
    """fused group-by vs oracle: group order, keys, first rows, every aggregate"""
    ogb = oracle.groupby(keys)
    names = ["sum", "min", "max", "count", "avg", "var"]
    for v in vals:
        ops = [ck.RED_NAMES[nm] for nm in names]
        gb = gpu.groupby_agg(keys, ops, [v] * len(ops), hint=hint)
        assert gb.ngroups == ogb["ngroups"]
        assert np.array_equal(gb.first_rows(), ogb["first_rows"])
        assert np.array_equal(gb.counts(), ogb["counts"])
        for k, key in enumerate(keys):
            assert np.array_equal(gb.keys(k, key.dtype), key[ogb["first_rows"]])
        for j, nm in enumerate(names):
            got = gb.result(j, ops[j], ck.tag_of(v))
            want = oracle.grouped_reduce(ops[j], v, ogb)
            if v.dtype.kind == "f" and nm in ("sum", "avg", "var"):
                w, g = want.astype(np.float64), got.astype(np.float64)
                scale = np.maximum(1.0, np.abs(w))
                assert np.all(np.abs(g - w) <= scale * len(v) * 2.0 ** -50), nm
            else:
                assert gu.same_bits(got, want), (nm, v.dtype)
        gb.destroy()


GB_CASES = [(3, 40000, 8, 0), (6, 30000, 3, 0), (3, 5000, 40, 0), (1, 1000, 10, 0), (1, 50000, 100, 128), (1, 30000, 3000, 0), (2, 40000, 20, 0), (2, 40000, 300, 0),
            (1, 1, 1, 0), (2, 17, 2, 0), (1, 200000, 100000, 0), (1, 100003, 100, 100)]


@pytest.mark.parametrize("nk,n,card,hint", GB_CASES)
def test_groupby_agg(gpu, oracle, nk, n, card, hint):
    rng = np.random.default_rng(nk * 1000 + n + card)
    keys = [rng.integers(-card, card, n).astype(np.int32) for _ in range(nk)]
    vals = [rand(rng, np.int32, n, small=True), np.round(rng.uniform(0, 100, n), 6).astype(np.float32),
            rand(rng, np.int16, n), rand(rng, np.float64, n)]
    check_agg(gpu, oracle, keys, vals, hint)


def test_groupby_agg_key_dtypes(gpu, oracle):
    rng = np.random.default_rng(77)
    n = 20000
    v = [rand(rng, np.int32, n, small=True)]
    check_agg(gpu, oracle, [(rng.integers(0, 50, n).astype(np.int64) << 33) - 7], v, 0)
    check_agg(gpu, oracle, [rng.integers(0, 50, n).astype(np.int16), rng.integers(0, 5, n).astype(np.uint8)], v, 0)
    # sentinel keys: INT_MIN (the LDS empty mark) and an all-ones packed pair
    k = rng.integers(0, 4, n).astype(np.int32)
    k[k == 0] = np.iinfo(np.int32).min
    check_agg(gpu, oracle, [k], v, 0)
    k2 = rng.integers(-1, 1, n).astype(np.int32)
    check_agg(gpu, oracle, [k2, k2.copy()], v, 0)


def test_groupby_build(gpu, oracle):
    rng = np.random.default_rng(9)
    for nk, n, card in ((1, 5000, 7), (1, 100001, 100), (2, 30000, 40), (1, 50000, 20000)):
        keys = [rng.integers(1, card, n, endpoint=True).astype(np.int32) for _ in range(nk)]
        o = oracle.groupby(keys)
        g = gpu.groupby_build(keys)
        assert g.ngroups == o["ngroups"]
        assert np.array_equal(g.reversemap(), o["reversemap"])
        assert np.array_equal(g.counts(), o["counts"])
        assert np.array_equal(g.first_rows(), o["first_rows"])
        g.destroy()


def test_groupby_golden(gpu):
    for c in [c for c in gu.load() if c["fn"] == "groupby"]:
        keys = [gu.dec(k) for k in c["keys"]]
        g = gpu.groupby_build(keys)
        assert g.ngroups == c["ngroups"]
        assert np.array_equal(g.reversemap(), gu.dec(c["reversemap"]))
        assert np.array_equal(g.counts(), gu.dec(c["counts"]))
        assert np.array_equal(g.first_rows(), gu.dec(c["first_rows"]))
        g.destroy()
        for a in c["aggs"]:
            if a["op"] in ("first", "last"):
                continue
            x = gu.dec(a["x"])
            gb = gpu.groupby_agg(keys, [ck.RED_NAMES[a["op"]]], [x])
            got, want = gb.result(0, ck.RED_NAMES[a["op"]], ck.tag_of(x)), gu.dec(a["out"])
            if x.dtype.kind == "f" and a["op"] in ("sum", "avg", "var"):
                assert np.allclose(got.astype(np.float64), want.astype(np.float64), rtol=1e-12, atol=1e-9), a["op"]
            else:
                assert gu.same_bits(got, want), (a["op"], c["src"])
            gb.destroy()
