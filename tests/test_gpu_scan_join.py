"""GPU parity, part 2: prefix scans, sliding windows, shifts, gather / mask filter, ht_postproc row lists,
grouped reductions over row lists, hash join -- HIP path (through the C-ABI) vs the oracle and the golden vectors."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu

pytestmark = pytest.mark.gpu

NUM_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]
EXACT_SCANS = ["mins", "maxs", "minw", "maxw", "deltas", "prev", "aggnext", "ratiow"]
SUM_SCANS = ["sums", "avgs", "sumw", "avgw"]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def rand(rng, dt, n, small=False):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return np.round(rng.uniform(0.5, 100, n), 6).astype(dt)
    hi = 12 if small else min(np.iinfo(dt).max, 20000)
    lo = 1 if (dt.kind == "u" or small) else max(np.iinfo(dt).min, -20000)
    x = rng.integers(lo, hi, n, endpoint=True).astype(dt)
    x[x == 0] = 1
    return x


def ulp_close(a, b, ulps):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


@pytest.mark.parametrize("dt", NUM_DTYPES)
@pytest.mark.parametrize("n", [1, 5, 2048, 2049, 70001])
def test_scans_exact(gpu, oracle, dt, n):
    """integer/index-valued scans and every min/max/shift: bit-exact"""
    rng = np.random.default_rng(n * 17 + np.dtype(dt).num)
    x = rand(rng, dt, n)
    for name in EXACT_SCANS:
        for w in (0, 1, 2, 3, 10, 100, 2047, 2048, 5000, 100000):
            if w == 0 and name == "ratiow":
                continue
            if name in ("mins", "maxs", "deltas", "prev", "aggnext") and w != 1:
                continue
            a, b = gpu.scan(ck.SCAN_NAMES[name], x, w), oracle.scan(ck.SCAN_NAMES[name], x, w)
            assert gu.same_bits(a, b), (name, w, dt, n)


@pytest.mark.parametrize("dt", NUM_DTYPES)
@pytest.mark.parametrize("n", [1, 5, 2049, 70001])
def test_scans_sums(gpu, oracle, dt, n):
    """sums/sumw on integers: bit-exact (128-bit).  avgs on integers: bit-exact (one division of an exact sum).
    avgw on integers: the device rounds the exact window mean once; the reference accumulates a floating recurrence
    (aggregations.h:270-271), so both are compared with the exact rational.  Floating inputs: tree order vs sequential."""
    rng = np.random.default_rng(n * 3 + np.dtype(dt).num)
    x = rand(rng, dt, n)
    is_int = np.dtype(dt).kind != "f"
    for name in SUM_SCANS:
        for w in (1, 2, 3, 10, 100, 2048, 5000, 100000):
            if name in ("sums", "avgs") and w != 1:
                continue
            a, b = gpu.scan(ck.SCAN_NAMES[name], x, w), oracle.scan(ck.SCAN_NAMES[name], x, w)
            if is_int and name in ("sums", "sumw", "avgs"):
                assert gu.same_bits(a, b), (name, w, dt, n)
            elif is_int:  # avgw
                ww = min(w, n)
                xs = [int(v) for v in x]
                pref = np.concatenate([[0], np.cumsum(np.array(xs, dtype=object))])
                idx = np.arange(n)
                lens = np.minimum(idx + 1, ww)
                exact = np.array([float((pref[i + 1] - pref[i + 1 - l])) / float(l) for i, l in zip(idx, lens)])
                assert ulp_close(a, exact, 1), (name, w, dt)                       # device: <= 1 ulp of the exact mean
                if np.dtype(dt).kind == "u" and np.dtype(dt).itemsize >= 4:
                    continue   # reference quirk: (arr[i] - arr[i-w]) wraps for unsigned 4/8-byte inputs (aggregations.h:271)
                drift = 4.0 * np.spacing(float(np.max(np.abs(exact))) + 1.0) * (idx + 2)   # ~2 roundings per step at the largest magnitude
                assert np.all(np.abs(b - exact) <= drift), (name, w, dt)           # reference: inside its recurrence drift
                assert np.all(np.abs(a - b) <= drift)
            else:
                scale = np.maximum(1.0, np.abs(b.astype(np.float64)))
                # avgw: the reference subtracts arr[i]-arr[i-w] in T (float32 rounding per step), then drifts
                eps = float(np.finfo(dt).eps) if name == "avgw" else 2.0 ** -52
                sabs = float(np.sum(np.abs(x.astype(np.float64))))
                # any summation order: |err| <= (n-1) u sum|x| (both sides); avgw additionally carries the reference's per-step T rounding
                tol = 2 * n * 2.0 ** -52 * sabs + (100.0 * eps * float(np.max(np.abs(x))) * (np.arange(n) + 8) if name == "avgw" else 0)
                assert np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol), (name, w, dt)


def test_scans_golden(gpu):
    for c in [c for c in gu.load() if c["fn"] == "scan"]:
        x = gu.dec(c["x"])
        name = c["op"]
        if x.dtype.kind == "f" and name in SUM_SCANS:
            continue
        got = gpu.scan(ck.SCAN_NAMES[name], x, c["w"])
        want = gu.dec(c["out"])
        if name == "avgw":
            if x.dtype.kind == "u" and x.dtype.itemsize >= 4:
                continue   # reference wraps (arr[i] - arr[i-w]) in unsigned arithmetic
            assert np.allclose(got, want, rtol=1e-12, atol=1e-9), (c["w"], c["src"], c["x"]["dtype"])
        else:
            assert gu.same_bits(got, want), (name, c["w"], c["x"]["dtype"], c["src"])


def test_vars_windows(gpu, oracle):
    """vars/stddevs (floating recurrence in the reference) and varw/stddevw (undefined in the reference, D9):
    compared with the oracle's restatement within a relative tolerance"""
    rng = np.random.default_rng(4)
    for dt in (np.int32, np.float32, np.float64):
        x = rand(rng, dt, 5000, small=(np.dtype(dt).kind != "f"))
        for name in ("vars", "stddevs"):
            a, b = gpu.scan(ck.SCAN_NAMES[name], x), oracle.scan(ck.SCAN_NAMES[name], x)
            assert np.allclose(a, b, rtol=1e-8, atol=1e-8), (name, dt)
        for name in ("varw", "stddevw"):
            for w in (2, 5, 100):
                a, b = gpu.scan(ck.SCAN_NAMES[name], x, w), oracle.scan(ck.SCAN_NAMES[name], x, w)
                assert np.allclose(a, b, rtol=1e-7, atol=1e-6), (name, dt, w)


@pytest.mark.parametrize("dt", NUM_DTYPES)
def test_gather_compact(gpu, oracle, dt):
    rng = np.random.default_rng(3)
    for n in (1, 400, 100001):
        x = rand(rng, dt, n)
        idx = rng.integers(0, n, 3 * n + 7).astype(np.uint32)
        assert gu.same_bits(gpu.gather(x, idx), oracle.gather(x, idx))
        mask = rng.integers(0, 2, n).astype(np.uint8)
        assert gu.same_bits(gpu.compact(x, mask), oracle.compact(x, mask))
        assert np.array_equal(gpu.mask_to_index(mask), np.nonzero(mask)[0].astype(np.uint32))
        assert gpu.compact(x, np.zeros(n, np.uint8)).size == 0
        assert gu.same_bits(gpu.compact(x, np.ones(n, np.uint8)), x)


def test_gather_compact_golden(gpu):
    for c in [c for c in gu.load() if c["fn"] == "gather"]:
        assert gu.same_bits(gpu.gather(gu.dec(c["x"]), gu.dec(c["idx"])), gu.dec(c["out"]))
    for c in [c for c in gu.load() if c["fn"] == "compact"]:
        assert gu.same_bits(gpu.compact(gu.dec(c["x"]), gu.dec(c["mask"])), gu.dec(c["out"]))


PP_CASES = [(3, 20000, 6), (6, 20000, 2), (1, 1000, 10), (1, 100001, 100), (1, 30000, 3000), (2, 40000, 20), (1, 1, 1), (1, 4097, 2), (1, 200000, 70000), (1, 65536, 257)]


@pytest.mark.parametrize("nk,n,card", PP_CASES)
def test_postproc_and_grouped_reduce(gpu, oracle, nk, n, card):
    """reversemap, counts, ht_postproc offsets and DESCENDING row-id lists: bit-exact; out[g] = op(col[vecs[g]])"""
    rng = np.random.default_rng(nk * 1000 + n + card)
    keys = [rng.integers(1, card, n, endpoint=True).astype(np.int32) for _ in range(nk)]
    o = oracle.groupby(keys)
    g = gpu.groupby_build(keys)
    off, rows = g.postproc()
    assert g.ngroups == o["ngroups"]
    assert np.array_equal(off[:-1], o["offsets"]) and off[-1] == n
    assert np.array_equal(rows, o["row_ids"])
    for vdt in (np.int32, np.float32, np.int64, np.int16):
        v = rand(rng, vdt, n, small=True)
        for name in ("sum", "min", "max", "count", "avg", "first", "last", "var"):
            got = gpu.grouped_reduce(g, ck.RED_NAMES[name], v)
            want = oracle.grouped_reduce(ck.RED_NAMES[name], v, o)
            if np.dtype(vdt).kind == "f" and name in ("sum", "avg", "var"):
                scale = np.maximum(1.0, np.abs(want.astype(np.float64)))
                assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= scale * n * 2.0 ** -50), name
            else:
                assert gu.same_bits(got, want), (name, vdt)
    g.destroy()


def test_postproc_golden(gpu):
    for c in [c for c in gu.load() if c["fn"] == "groupby"]:
        keys = [gu.dec(k) for k in c["keys"]]
        g = gpu.groupby_build(keys)
        off, rows = g.postproc()
        assert np.array_equal(off[:-1], gu.dec(c["offsets"]))
        assert np.array_equal(rows, gu.dec(c["row_ids"]))
        for a in c["aggs"]:
            x = gu.dec(a["x"])
            got, want = gpu.grouped_reduce(g, ck.RED_NAMES[a["op"]], x), gu.dec(a["out"])
            if x.dtype.kind == "f" and a["op"] in ("sum", "avg", "var"):
                assert np.allclose(got.astype(np.float64), want.astype(np.float64), rtol=1e-12, atol=1e-9)
            else:
                assert gu.same_bits(got, want), (a["op"], c["src"])
        g.destroy()


def test_join(gpu, oracle):
    rng = np.random.default_rng(21)
    for nb, npr, card, dt in ((100, 5000, 100, np.int32), (1000, 20000, 300, np.int32), (50, 1000, 200, np.int64), (1, 10, 2, np.int16)):
        build = rng.integers(1, card, nb, endpoint=True).astype(dt)
        probe = rng.integers(1, card + 20, npr, endpoint=True).astype(dt)
        pr, br = gpu.join_pairs(build, probe)
        opr, obr = oracle.join_pairs(build, probe)
        assert np.array_equal(pr, opr) and np.array_equal(br, obr), (nb, npr, card)
    # unique-key dimension lookup (h2o join + group-by shape)
    dim = rng.permutation(np.arange(1, 101, dtype=np.int32))
    fact = rng.integers(1, 120, 100000, endpoint=True).astype(np.int32)
    look = gpu.join_lookup(dim, fact)
    pos = {int(k): i for i, k in enumerate(dim)}
    want = np.array([pos.get(int(k), 0xFFFFFFFF) for k in fact], dtype=np.uint32)
    assert np.array_equal(look, want)


def test_large_properties(gpu):
    """size-independent properties at 1e8 rows (BASELINE configs are 1e8..1e9): sum of group sums == column sum,
    group counts add up, window min <= value, prefix max is monotone, round trip of compaction"""
    n = 100_000_000
    id1 = gpu.gen_column(ck.GEN_ID1, 42, 0, n, n, 100)
    v1 = gpu.gen_column(ck.GEN_V1, 42, 0, n, n, 100)
    gb = gpu.groupby_agg([id1], [ck.RED_SUM, ck.RED_COUNT], [v1, v1], hint=128)
    sums = ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32))
    cnts = gb.result(1, ck.RED_COUNT, ck.INT32)
    assert gb.ngroups == 100
    assert sum(sums) == int(gpu.reduce(ck.RED_SUM, v1))
    assert int(cnts.sum()) == n
    assert sorted(gb.keys(0, np.int32).tolist()) == list(range(1, 101))
    first = gb.first_rows()
    assert np.all(np.diff(first.astype(np.int64)) > 0)          # group order = first occurrence
    price = gpu.gen_column(ck.GEN_PRICE, 42, 0, n, n, 100)
    mw = gpu.scan(ck.SCAN_MINW, price, 10, keep=True)
    d = gpu.ewise(ck.OP_SUB, price, mw, keep=True)                # price - minw(10, price) >= 0
    assert int(gpu.reduce(ck.RED_MIN, d)) >= 0
    assert int(gpu.reduce(ck.RED_MAX, mw)) <= int(gpu.reduce(ck.RED_MAX, price))
    sm = gpu.scan(ck.SCAN_SUMS, v1, keep=True)
    last = sm.to_host()[-1]
    assert (int(last["hi"]) << 64) + int(last["lo"]) == sum(sums)


def test_join_match_count_beyond_32_bits(gpu):
    """70,000 x 70,000 rows with one key: 4.9e9 pairs.  The count is exact in 64 bits; the pair form refuses (its outputs are
    addressed by uint32 offsets) instead of expanding through a wrapped 32-bit scan (join.hip)."""
    import ctypes as C
    import aquery2_amd.capi as capi
    k = np.full(70_000, 7, np.int32)
    assert gpu.join_count(k, k) == 70_000 * 70_000
    bd, pd = gpu._dev(k), gpu._dev(k)
    pr, br = gpu.empty(16, np.uint32), gpu.empty(16, np.uint32)
    m = C.c_uint64()
    rc = gpu.lib.aqg_join_pairs(gpu.ctx, bd.tag, C.c_void_p(bd.ptr), C.c_uint32(bd.n), C.c_void_p(pd.ptr), C.c_uint32(pd.n),
                                C.c_void_p(pr.ptr), C.c_void_p(br.ptr), C.c_uint64(1 << 40), C.byref(m))
    assert rc == 6 and m.value == 70_000 * 70_000          # AQG_ERR_OVERFLOW, true count reported
    # exactly 2^32 pairs used to read as zero matches
    a, b = np.full(65_536, 3, np.int32), np.full(65_536, 3, np.int32)
    assert gpu.join_count(a, b) == 1 << 32
