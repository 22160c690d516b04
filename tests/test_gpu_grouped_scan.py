"""Per-group scans, windows, shifts, reductions of scan results and two-column aggregates over ALL groups of a build in a fixed
number of launches (aqg_grouped_flatten / aqg_grouped_scan[_flat] / aqg_grouped_reduce_flat / aqg_grouped_corr): the device form of
the generated loop  for g: out[g] = f(col[vecs[g]])  (engine/ast.py:722-789, mem_opt.cpp:53-63).  The oracle is the COMPOSITION the
reference itself runs: its group-by (row lists in ht_postproc order, descending row ids) and then its scan / reduction / corr over
every group's gathered rows, laid out in the flat buffer sliced by the offsets.  Bit-exact for integer results; floating sums within
the bounds of tests/test_gpu_scan_fuzz.py, applied per group."""
import os

import numpy as np
import pytest

import checker as ck
import golden_util as gu
from test_gpu_basic import rand

pytestmark = pytest.mark.gpu
DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def flat_groups(ogb):
    """[(start, count)] of every group in the flat layout"""
    off = np.concatenate([[0], np.cumsum(ogb["counts"].astype(np.int64))])
    return [(int(off[g]), int(off[g + 1] - off[g])) for g in range(ogb["ngroups"])]


def compose(ogb, x, fn, out_dtype):
    """flat[offsets[g] + i] = fn(x[vecs[g]])[i]"""
    out = np.zeros(x.size, dtype=out_dtype)
    rows = ogb["row_ids"]
    for s, c in flat_groups(ogb):
        out[s:s + c] = fn(x[rows[s:s + c]])
    return out


def pos_in_group(ogb, n):
    p = np.zeros(n, np.int64)
    for s, c in flat_groups(ogb):
        p[s:s + c] = np.arange(c)
    return p


def keys_for(rng, n, G):
    k = rng.integers(0, G, n).astype(np.int32)
    if n >= 4 and rng.integers(3) == 0:
        k[: n // 2] = k[0]                       # one long group in front: carries over many tiles
    return k


@pytest.mark.parametrize("esz_dtype", [np.uint8, np.int16, np.int32, np.float32, np.int64, np.float64])
@pytest.mark.parametrize("n,G", [(1, 1), (5000, 7), (70_001, 300), (300_001, 70_000)])
def test_flatten_is_the_gather_through_the_row_lists(gpu, oracle, esz_dtype, n, G):
    rng = np.random.default_rng(n + G)
    keys = keys_for(rng, n, G)
    x = rand(rng, esz_dtype, n, small=False) if np.dtype(esz_dtype).kind != "f" else rng.standard_normal(n).astype(esz_dtype)
    ogb = oracle.groupby([keys])
    gb = gpu.groupby_build([keys])
    assert np.array_equal(gpu.group_offsets(gb)[:-1], ogb["offsets"]) and gpu.group_offsets(gb)[-1] == n
    assert gu.same_bits(gpu.grouped_flatten(gb, x), x[ogb["row_ids"]])


SCANS = ["sums", "avgs", "mins", "maxs", "deltas", "prev", "aggnext"]
WINDOWS = ["sumw", "avgw", "minw", "maxw", "ratiow"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("AQG_FUZZ_SEEDS", "40"))))
def test_grouped_scan_random_shapes(gpu, oracle, seed):
    rng = np.random.default_rng(int(os.environ.get("AQG_FUZZ_BASE", "9100")) + seed)
    dt = DTYPES[rng.integers(len(DTYPES))]
    n = int(rng.choice([1, 7, 2047, 2049, 70_001, 300_001]))
    G = int(rng.choice([1, 3, 100, 300, 5000, 70_000]))
    fp = np.dtype(dt).kind == "f"
    x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rand(rng, dt, n, small=True)
    keys = keys_for(rng, n, G)
    ogb = oracle.groupby([keys])
    gb = gpu.groupby_build([keys])
    pos = pos_in_group(ogb, n)
    xf = x[ogb["row_ids"]]
    absx = np.abs(xf.astype(np.float64))
    cum_abs = compose(ogb, np.abs(x.astype(np.float64)), np.cumsum, np.float64)
    for name in rng.choice(SCANS, 3, replace=False):
        op = ck.SCAN_NAMES[str(name)]
        want = compose(ogb, x, lambda v: oracle.scan(op, v), ck.TAG2NP[oracle.scan_out_dtype(op, ck.tag_of(x))])
        got = gpu.grouped_scan(gb, op, x)
        if fp and name in ("sums", "avgs"):
            bound = 2.0 ** -52 * (pos + 2) * cum_abs
            if name == "avgs":
                bound = bound / (pos + 1)
            assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound + 1e-12), (seed, name, dt, n, G)
        else:
            assert gu.same_bits(got, want), (seed, name, dt, n, G)
    for name in rng.choice(WINDOWS, 3, replace=False):
        w = int(rng.choice([1, 2, 3, 5, 10, 64, 100, 1000, 2500, 30_000, n, n + 3]))
        op = ck.SCAN_NAMES[str(name)]
        if name == "avgw" and np.dtype(dt).kind == "u" and np.dtype(dt).itemsize >= 4:
            continue                                          # the reference wraps arr[i] - arr[i-w] for unsigned 4/8-byte inputs
        want = compose(ogb, x, lambda v: oracle.scan(op, v, w), ck.TAG2NP[oracle.scan_out_dtype(op, ck.tag_of(x))])
        got = gpu.grouped_scan(gb, op, x, w)
        if name in ("minw", "maxw") or (name == "sumw" and not fp) or name == "ratiow":
            assert gu.same_bits(got, want), (seed, name, dt, w, n, G)
        else:
            eps_in = float(np.finfo(dt).eps) if fp else 2.0 ** -52
            bound = 4 * eps_in * float(np.max(absx)) * (pos + 2) + 1e-9
            assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound), (seed, name, dt, w, n, G)
    # the two-step form (flatten once, scan the flat column) gives the same bits as the one-call form
    xflat = gpu.grouped_flatten(gb, x, keep=True)
    op = ck.SCAN_NAMES["mins"]
    assert gu.same_bits(gpu.grouped_scan(gb, op, xflat, flat=True), gpu.grouped_scan(gb, op, x))


@pytest.mark.parametrize("dt", [np.int8, np.int32, np.uint16, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("n,G", [(1, 1), (4097, 5), (120_001, 1000), (300_001, 90_000)])
def test_reductions_of_flat_columns(gpu, oracle, dt, n, G):
    """max(ratios(x[vecs[g]])) and friends (tests/q4.a:23): out[g] = op(flat[offsets[g] .. offsets[g+1]))"""
    rng = np.random.default_rng(n * 3 + G)
    fp = np.dtype(dt).kind == "f"
    keys = keys_for(rng, n, G)
    x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rand(rng, dt, n, small=True)
    ogb = oracle.groupby([keys])
    gb = gpu.groupby_build([keys])
    flat = x[ogb["row_ids"]]                     # any flat column will do
    groups = flat_groups(ogb)
    # the oracle's per-group reduction over the flat column: its row lists are then simply consecutive positions
    fgb = dict(ngroups=ogb["ngroups"], offsets=ogb["offsets"], counts=ogb["counts"], row_ids=np.arange(n, dtype=np.uint32))
    for name in ("sum", "min", "max", "avg", "count", "first", "last", "var"):
        op = ck.RED_NAMES[name]
        got = gpu.grouped_reduce_flat(gb, op, flat)
        want = oracle.grouped_reduce(op, flat, fgb)
        if fp and name in ("sum", "avg", "var"):
            for g, (s, c) in enumerate(groups):
                a = np.abs(flat[s:s + c].astype(np.float64))
                scale = float(np.sum(a)) if name != "var" else float(np.sum(a * a)) + float(np.sum(a)) ** 2
                assert abs(float(got[g]) - float(want[g])) <= (c + 2) * 2.0 ** -50 * max(scale, 1e-300) + 1e-12, (name, dt, g)
        else:
            assert gu.same_bits(got, want), (name, dt, n, G)


@pytest.mark.parametrize("dx,dy", [(np.int32, np.int32), (np.int8, np.uint16), (np.uint32, np.int32), (np.int16, np.int16)])
@pytest.mark.parametrize("n,G", [(50, 4), (100_003, 100), (400_001, 10_000)])
def test_grouped_corr(gpu, oracle, dx, dy, n, G):
    """h2o Q9 `corr(v1, v2) BY id2, id4` (benchmark/h2o/groupby.sql:20): five 128-bit sums per group, the reference's formula"""
    rng = np.random.default_rng(n + G + np.dtype(dx).itemsize)
    keys = keys_for(rng, n, G)
    x, y = rand(rng, dx, n, small=False), rand(rng, dy, n, small=False)
    ogb = oracle.groupby([keys])
    gb = gpu.groupby_build([keys])
    got = gpu.grouped_corr(gb, x, y)
    rows = ogb["row_ids"]
    want = np.array([oracle.corr(x[rows[s:s + c]], y[rows[s:s + c]]) for s, c in flat_groups(ogb)])
    assert gu.same_bits(got, want), (dx, dy, n, G)


def test_q7_shape_at_size(gpu, oracle):
    """benchmark/quries/Aquery/q7.a `SELECT stocksymbol, avgs(5, price) ... ASSUMING ASC time GROUP BY stocksymbol` and the frozen sample's
    avgw(10, sales[vecs[i]], col[i]) (mem_opt.cpp:61): 1e7 rows over 1e5 symbols, ONE call for all groups"""
    n, G = 10_000_000, 100_000
    rng = np.random.default_rng(77)
    sym = rng.integers(0, G, n).astype(np.int32)
    price = (rng.integers(50, 500, n)).astype(np.int32)
    ogb = oracle.groupby([sym])
    gb = gpu.groupby_build([sym])
    assert gb.ngroups == ogb["ngroups"]
    for name, w in (("avgw", 5), ("maxw", 10), ("sums", 0)):
        op = ck.SCAN_NAMES[name]
        got = gpu.grouped_scan(gb, op, price, w)
        want = compose(ogb, price, lambda v: oracle.scan(op, v, w), ck.TAG2NP[oracle.scan_out_dtype(op, ck.INT32)])
        if name == "avgw":
            assert np.all(np.abs(got - want) <= 1e-9 * np.abs(want) + 1e-9)
        else:
            assert gu.same_bits(got, want), name
