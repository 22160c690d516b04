"""GPU parity of the library's own exchange (aqg_groupby_agg_sharded): a table cut into row-range shards, every shard
grouped by its own rank, ONE all-gather of the shard tables, re-aggregation -- against the oracle's group-by of the WHOLE table:
group order (global first occurrence), keys, global first rows, every aggregate.  Ranks are threads with their own contexts on the
one GPU of the box and a copy-and-barrier all-gather (RCCL refuses two ranks on one device); the RCCL transport itself runs with a
world of one.  The same C entry point is what bench.py --gpus N calls."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu

pytestmark = pytest.mark.gpu


def shards_of(n, world, uneven=True):
    cuts = [0] + sorted(np.random.default_rng(n + world).integers(0, n, world - 1).tolist()) + [n] if uneven else [n * r // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def run_sharded(world, keys, ops, vals, hint=0, gmax=0, uneven=True, empty_rank=None):
    import aquery2_amd
    n = len(keys[0])
    sh = shards_of(n, world, uneven)
    if empty_rank is not None:
        lo, hi = sh[empty_rank]
        sh[empty_rank] = (hi, hi)
        if empty_rank > 0: sh[empty_rank - 1] = (sh[empty_rank - 1][0], hi)
        else: sh[1] = (lo, sh[1][1])
    tr = aquery2_amd.ThreadRanks(world)
    def body(rank, dev, comm):
        lo, hi = sh[rank]
        k = [np.ascontiguousarray(c[lo:hi]) for c in keys]
        up = {}                                              # one device column per distinct host column (as a caller would hold them)
        for c in vals:
            if c is not None and id(c) not in up:
                up[id(c)] = dev.to_device(np.ascontiguousarray(c[lo:hi]))
        v = [up[id(c)] if c is not None else None for c in vals]
        res = None
        gb = None
        for _ in range(2):                                   # the second call reuses every buffer of the first
            gb = comm.groupby_agg_sharded(k, ops, v, row_base=lo, hint=hint, gmax=gmax, handle=gb)
        res = {"G": gb.ngroups, "keys": [gb.keys(i, c.dtype) for i, c in enumerate(keys)], "first": gb.first_rows64(),
               "res": [gb.result(j, op, ck.tag_of(vals[j]) if vals[j] is not None else ck.INT32) for j, op in enumerate(ops)]}
        assert gb.first_rows() is None
        gb.destroy()
        return res
    try:
        return tr.run(body)
    finally:
        tr.close()


def check(world, keys, ops, vals, oracle, **kw):
    out = run_sharded(world, keys, ops, vals, **kw)
    o = oracle.groupby(keys)
    for r in out:
        assert r["G"] == o["ngroups"]
        assert np.array_equal(r["first"], o["first_rows"].astype(np.int64))
        for i, c in enumerate(keys):
            assert np.array_equal(r["keys"][i], c[o["first_rows"]])
        for j, op in enumerate(ops):
            x = vals[j] if vals[j] is not None else vals[0]
            want = oracle.grouped_reduce(op, x, o)
            got = r["res"][j]
            if x.dtype.kind == "f" and op in (ck.RED_SUM, ck.RED_AVG):
                w, g = want.astype(np.float64), got.astype(np.float64)
                assert np.all(np.abs(g - w) <= np.maximum(1.0, np.abs(w)) * len(x) * 2.0 ** -50), op
            elif x.dtype.kind == "f" and op in (ck.RED_VAR, ck.RED_STDDEV):
                # moments summed in another order: q - s*s/(n+1) cancels, so the bound is relative to the second moment
                w, g = want.astype(np.float64), got.astype(np.float64)
                q = oracle.grouped_reduce(ck.RED_SUM, (x.astype(np.float64) ** 2), o).astype(np.float64)
                cnt = o["counts"].astype(np.float64)
                tol = (q / (cnt + 1)) * len(x) * 2.0 ** -48 + 1e-300
                if op == ck.RED_STDDEV:
                    tol = np.sqrt(tol) + tol / np.maximum(np.abs(w), 1e-300)
                assert np.all(np.abs(g - w) <= tol), op
            else:
                assert gu.same_bits(got, want), (op, x.dtype)


def test_q1_q4_shapes_three_and_five_shards(oracle):
    rng = np.random.default_rng(1)
    n = 400_003
    id1 = rng.integers(1, 101, n).astype(np.int32)
    v1, v2 = rng.integers(1, 6, n).astype(np.int32), rng.integers(1, 16, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6).astype(np.float32)
    check(3, [id1], [ck.RED_SUM], [v1], oracle, hint=128, gmax=128)                              # h2o Q1: one collective
    check(5, [id1], [ck.RED_AVG, ck.RED_AVG, ck.RED_AVG], [v1, v2, v3], oracle, hint=128, gmax=128)   # h2o Q4
    check(2, [id1], [ck.RED_SUM, ck.RED_COUNT, ck.RED_MIN, ck.RED_MAX, ck.RED_AVG], [v1, v1, v3, v2, v1], oracle, empty_rank=0)


def test_multi_key_high_cardinality_variable_payload(oracle):
    """h2o Q5 / Q10 shapes: the shard tables have different sizes (gmax = 0: the group counts are exchanged first) and nearly
    disjoint key sets; several key columns of different widths; negative keys and the key equal to the table's empty mark"""
    rng = np.random.default_rng(2)
    n = 250_000
    id6 = rng.integers(1, 40_000, n).astype(np.int32)
    v1, v2 = rng.integers(-5, 6, n).astype(np.int32), rng.integers(1, 16, n).astype(np.int16)
    v3 = np.round(rng.uniform(0, 100, n), 6).astype(np.float32)
    check(4, [id6], [ck.RED_SUM, ck.RED_SUM, ck.RED_SUM], [v1, v2, v3], oracle)                   # Q5
    ids = [rng.integers(-3, 4, n).astype(np.int32), rng.integers(0, 50, n).astype(np.int16), rng.integers(0, 2, n).astype(np.uint8),
           rng.integers(-2**40, 2**40, n).astype(np.int64) // 2**38]
    ids[0][::1000] = -2**31
    check(3, ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], oracle)                                   # Q10 shape: wide tuple
    d = rng.standard_normal(n)
    check(2, [id6, ids[1]], [ck.RED_MAX, ck.RED_MIN, ck.RED_AVG, ck.RED_SUM], [d, v1, d, d], oracle)


def test_second_moments_and_eight_byte_sums(oracle):
    """VAR / STDDEV travel as {sum, sum of squares, count}; a partial that needs 128 bits per shard (sums of 8-byte integers, sums of
    squares) as two 8-byte columns.  Integer results are bit-identical to the oracle's whole-table aggregates."""
    rng = np.random.default_rng(3)
    n = 300_007
    id1 = rng.integers(1, 101, n).astype(np.int32)
    v1 = rng.integers(-50_000, 50_000, n).astype(np.int32)
    v2 = rng.integers(-300, 300, n).astype(np.int16)
    u4 = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    big = rng.integers(-2**62, 2**62, n).astype(np.int64)             # a group's sum passes 64 bits
    ubig = rng.integers(0, 2**64, n, dtype=np.uint64)
    v3 = np.round(rng.uniform(0, 100, n), 6).astype(np.float32)
    d = rng.standard_normal(n) * 1e3
    check(3, [id1], [ck.RED_VAR, ck.RED_STDDEV], [v1, v1], oracle, hint=128, gmax=128)
    check(4, [id1], [ck.RED_VAR, ck.RED_AVG, ck.RED_SUM], [v2, v2, v1], oracle)
    check(2, [id1], [ck.RED_STDDEV], [u4], oracle, hint=128)
    check(3, [id1], [ck.RED_SUM, ck.RED_AVG, ck.RED_MIN], [big, big, big], oracle)
    check(2, [id1], [ck.RED_SUM, ck.RED_AVG], [ubig, ubig], oracle, empty_rank=1)
    check(3, [id1], [ck.RED_VAR, ck.RED_STDDEV], [v3, d], oracle)
    check(5, [id1, v2 % 3], [ck.RED_VAR, ck.RED_COUNT], [d, d], oracle)
    mid = rng.integers(-2**40, 2**40, n).astype(np.int64)            # 8-byte integers: sum and sum of squares both as two columns, three merge calls
    check(3, [id1], [ck.RED_VAR, ck.RED_STDDEV, ck.RED_AVG, ck.RED_MAX], [mid, mid, mid, v1], oracle)


def test_group_order_is_global_first_occurrence(oracle):
    """keys that first appear late in an early shard and early in a late one: the merged order must follow global row ids"""
    n = 90_000
    k = np.zeros(n, np.int32)
    k[:30_000] = np.arange(30_000) % 7
    k[29_990:30_000] = 100 + np.arange(10)            # new keys at the very end of where shard 0 may end
    k[30_000:60_000] = 50 - (np.arange(30_000) % 13)
    k[60_000:] = (np.arange(30_000) * 7919) % 211
    v = np.arange(n, dtype=np.int32) % 1000
    check(3, [k], [ck.RED_SUM], [v], oracle, uneven=False)
    check(4, [k], [ck.RED_MAX, ck.RED_COUNT], [v, v], oracle)


@pytest.mark.parametrize("two_keys", [False, True])
def test_a_shard_over_gmax_fails_the_call_on_every_rank_instead_of_hanging(two_keys):
    """one skewed shard holds more groups than gmax: that rank still joins the all-gather (with an empty table and its status in the
    header) and EVERY rank returns AQG_ERR_OVERFLOW -- before, the rank returned alone and the others waited in the collective for ever.
    One key column takes the one-launch merge, two the concatenation path; both read the status words."""
    import aquery2_amd
    world, per = 3, 4000
    rng = np.random.default_rng(5)
    def shard_keys(rank):
        k = rng.integers(0, 8, per).astype(np.int32)
        if rank == 1:
            k = rng.integers(0, 500, per).astype(np.int32)            # 500 groups against gmax = 16
        return [k, (k % 3).astype(np.int32)] if two_keys else [k]
    data = [(shard_keys(r), rng.integers(1, 9, per).astype(np.int32)) for r in range(world)]
    tr = aquery2_amd.ThreadRanks(world)
    def body(rank, dev, comm):
        k, v = data[rank]
        codes = []
        for _ in range(2):                                             # the communicator stays usable after a failed call
            try:
                comm.groupby_agg_sharded(k, [ck.RED_SUM], [v], row_base=rank * per, hint=64, gmax=16)
                codes.append(0)
            except aquery2_amd.capi.AqgError as e:
                codes.append(e.code)
        ok = comm.groupby_agg_sharded([c % 4 for c in k], [ck.RED_SUM], [v], row_base=rank * per, hint=64, gmax=16)   # every shard within bounds again
        return codes, ok.ngroups
    try:
        out = tr.run(body)
    finally:
        tr.close()
    assert [c for c, _ in out] == [[6, 6]] * world, out                # AQG_ERR_OVERFLOW everywhere, twice
    assert len({g for _, g in out}) == 1 and out[0][1] >= 4


def _cuts(n, world, rng, empty=None, tiny=None):
    c = [0] + sorted(rng.integers(0, n + 1, world - 1).tolist()) + [n]
    if tiny is not None and world > 2:                           # a shard of a few rows in the middle: halos reach across it
        c[tiny + 1] = min(n, c[tiny] + 3)
        c = [0] + sorted(c[1:-1]) + [n]
    sh = [(c[r], c[r + 1]) for r in range(world)]
    if empty is not None:
        lo, hi = sh[empty]
        sh[empty] = (lo, lo)
        if empty + 1 < world: sh[empty + 1] = (lo, sh[empty + 1][1])
        else: sh[empty - 1] = (sh[empty - 1][0], hi)
    return sh


@pytest.mark.parametrize("world,empty,tiny", [(2, None, None), (3, 1, None), (4, None, 1), (5, 0, 2)])
def test_reductions_over_row_range_shards(oracle, world, empty, tiny):
    """aqg_reduce_sharded / aqg_corr_sharded: every reduction of the WHOLE column from its shards with ONE all-gather of raw moments;
    integer results bit for bit, floating sums within the summation-order bound (server/aggregations.h:19-32,71-86,332-407)"""
    import aquery2_amd
    rng = np.random.default_rng(100 * world + (empty or 0))
    n = 200_003
    cols = {dt: (np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if np.dtype(dt).kind == "f" else rng.integers(-120 if np.dtype(dt).kind == "i" else 0, 121, n).astype(dt))
            for dt in (np.int8, np.int32, np.uint32, np.int64, np.uint64, np.float32, np.float64)}
    cols[np.int64] = cols[np.int64] * (1 << 40)                   # sums that need more than 64 bits per shard pair
    y = rng.integers(-30000, 30000, n).astype(np.int32)
    sh = _cuts(n, world, rng, empty, tiny)
    tr = aquery2_amd.ThreadRanks(world)
    def body(rank, dev, comm):
        lo, hi = sh[rank]
        out = {}
        for dt, x in cols.items():
            xs = dev.to_device(np.ascontiguousarray(x[lo:hi]))
            out[dt] = {name: comm.reduce_sharded(op, xs) for name, op in ck.RED_NAMES.items() if not (name == "avg" and n == 0)}
        out["corr"] = comm.corr_sharded(np.ascontiguousarray(cols[np.int32][lo:hi]), np.ascontiguousarray(y[lo:hi]))
        return out
    try:
        res = tr.run(body)
    finally:
        tr.close()
    for r in res:
        for dt, x in cols.items():
            fp = np.dtype(dt).kind == "f"
            for name, op in ck.RED_NAMES.items():
                want, got = oracle.reduce(op, x), r[dt][name]
                if fp and name in ("sum", "avg", "var", "stddev"):
                    scale = float(np.sum(np.abs(x.astype(np.float64)))) if name in ("sum", "avg") else float(np.sum(x.astype(np.float64) ** 2))
                    tol = n * 2.0 ** -52 * scale / (n if name == "avg" else 1) + 1e-9
                    if name in ("var", "stddev"):
                        tol = max(tol, abs(float(want)) * 1e-9)
                    assert abs(float(got) - float(want)) <= tol, (dt, name, got, want)
                else:
                    assert gu.scalar_same(got, want), (dt, name, got, want)
        assert r["corr"] == oracle.corr(cols[np.int32], y)


@pytest.mark.parametrize("world,empty,tiny", [(2, None, None), (3, 2, None), (4, None, 1), (5, 0, 3)])
def test_scans_windows_and_shifts_over_row_range_shards_inside_the_library(oracle, world, empty, tiny):
    """aqg_scan_sharded: every rank's rows of the scan of the WHOLE column -- sums / avgs resume from the exact totals of the earlier
    shards, mins / maxs from their min / max, windows take the last w rows of the shards before (across a 3-row shard and an empty
    one), shifts their neighbour rows; windows shorter and longer than the shards and than the column (aggregations.h:89-281,439-485)"""
    import aquery2_amd
    rng = np.random.default_rng(7 * world + (tiny or 0))
    n = 60_013
    sh = _cuts(n, world, rng, empty, tiny)
    cases = []
    for dt in (np.int32, np.int16, np.uint32, np.int64, np.float32, np.float64):
        fp = np.dtype(dt).kind == "f"
        x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rng.integers(1, 90, n).astype(dt)
        ops = [("sums", 0), ("avgs", 0), ("mins", 0), ("maxs", 0), ("deltas", 0), ("prev", 0), ("aggnext", 0), ("ratiow", 1), ("ratiow", 7), ("ratiow", n + 5),
               ("sumw", 3), ("avgw", 100), ("minw", 2), ("maxw", 50), ("minw", 20_000), ("maxw", 0), ("sumw", n + 10), ("minw", n + 1)]
        if dt in (np.uint32,):
            ops = [o for o in ops if o[0] != "avgw"]              # the reference wraps arr[i] - arr[i-w] for unsigned 4/8-byte inputs
        cases.append((dt, x, ops))
    neg = -np.abs(np.round(rng.uniform(1, 1000, n), 3)).astype(np.float64)    # all negative: the running max a window degrades to has no seed
    cases.append((np.float64, neg, [("maxw", 0), ("maxs", 0)]))
    tr = aquery2_amd.ThreadRanks(world)
    def body(rank, dev, comm):
        lo, hi = sh[rank]
        out = []
        for dt, x, ops in cases:
            xs = dev.to_device(np.ascontiguousarray(x[lo:hi]))
            out.append([comm.scan_sharded(ck.SCAN_NAMES[name], xs, w) for name, w in ops])
        return out
    try:
        res = tr.run(body)
    finally:
        tr.close()
    for ci, (dt, x, ops) in enumerate(cases):
        fp = np.dtype(dt).kind == "f"
        for oi, (name, w) in enumerate(ops):
            want = oracle.scan(ck.SCAN_NAMES[name], x, w)
            got = np.concatenate([res[r][ci][oi] for r in range(world)])
            assert got.size == n
            if name in ("mins", "maxs", "minw", "maxw", "deltas", "prev", "aggnext", "ratiow") or (not fp and name in ("sums", "sumw", "avgs")):
                assert gu.same_bits(got, want), (dt, name, w)
            else:
                eps_in = float(np.finfo(dt).eps) if fp else 2.0 ** -52
                bound = 4 * eps_in * float(np.max(np.abs(x.astype(np.float64)))) * (np.arange(n) + 2) + 1e-9
                assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound), (dt, name, w)


def test_typed_and_string_keys_over_row_range_shards(oracle):
    """aqg_groupby_agg_sharded over key columns that are not plain integers (server/hasher.h:97-144): dates, times, timestamps and 128-bit
    integers travel as their normalised integer columns and come back in the caller's types; astring_view keys as codes of ONE dictionary
    merged over the shards in global first-occurrence order (aqg_str_encode_sharded).  Against the oracle's typed group-by of the whole table."""
    import aquery2_amd
    import keycases
    rng = np.random.default_rng(77)
    n, world = 30_011, 3
    dates = np.zeros((n, 4), np.uint8); dates[:, 0] = rng.integers(1, 29, n); dates[:, 1] = rng.integers(1, 4, n); dates[:, 2:] = np.array([2022], np.int16).view(np.uint8)
    times = np.zeros((n, 8), np.uint8); times[:, 4] = rng.integers(0, 3, n); times[:, 5] = rng.integers(0, 5, n); times[:, 7] = rng.integers(0, 255, n)   # (byte 7 is padding: ignored)
    stamps = np.concatenate([dates, times], axis=1)[:, :12].copy()
    big = np.zeros(n, ck.I128); big["lo"] = rng.integers(0, 50, n).astype(np.uint64) << np.uint64(60); big["hi"] = rng.integers(-2, 3, n)
    plain = rng.integers(0, 4, n).astype(np.int16)
    strs = [b"sym%d" % v for v in rng.integers(0, 300, n)]
    v = rng.integers(-50, 50, n).astype(np.int32)
    cuts = [0, 9_000, 9_003, n]
    cases = {"date": [(ck.DATE, dates)], "time+plain": [(ck.TIME, times), (ck.INT16, plain)], "timestamp": [(ck.TIMESTAMP, stamps)], "int128": [(ck.INT128, big), (ck.DATE, dates)]}
    tr = aquery2_amd.ThreadRanks(world)
    def body(rank, dev, comm):
        lo, hi = cuts[rank], cuts[rank + 1]
        out = {}
        vd = dev.to_device(np.ascontiguousarray(v[lo:hi]))
        for name, cols in cases.items():
            kd = [dev.key_col(tag, np.ascontiguousarray(data[lo:hi])) if tag in (ck.DATE, ck.TIME, ck.TIMESTAMP) else dev.to_device(np.ascontiguousarray(data[lo:hi])) for tag, data in cols]
            gb = comm.groupby_agg_sharded(kd, [ck.RED_SUM, ck.RED_COUNT], [vd, vd], row_base=lo, hint=0)
            eb = [ck.KEY_ELEM_BYTES.get(tag, np.asarray(data).dtype.itemsize) for tag, data in cols]
            out[name] = (gb.ngroups, gb.first_rows64(), [gb.keys_raw(k, e) for k, e in enumerate(eb)], gb.result(0, ck.RED_SUM, ck.INT32))
            gb.destroy()
        codes, nd = comm.str_encode_sharded(strs[lo:hi])
        gb = comm.groupby_agg_sharded([codes], [ck.RED_SUM], [vd], row_base=lo, hint=0)
        out["str"] = (nd, codes.to_host(), gb.ngroups, gb.first_rows64(), gb.result(0, ck.RED_SUM, ck.INT32))
        gb.destroy()
        return out
    try:
        res = tr.run(body)
    finally:
        tr.close()
    for name, cols in cases.items():
        o = oracle.groupby_typed(cols)
        want_sum = np.bincount(o["reversemap"].astype(np.int64), weights=v.astype(np.float64), minlength=o["ngroups"]).astype(np.int64)
        for r in res:
            G, first, keys, sums = r[name]
            assert G == o["ngroups"], name
            assert np.array_equal(first, o["first_rows"].astype(np.int64)), name
            assert ck.i128_to_int(sums) == want_sum.tolist(), name
            for k, (tag, data) in enumerate(cols):
                host = np.ascontiguousarray(data).reshape(n, -1).view(np.uint8).reshape(n, -1)
                want = host[o["first_rows"]].copy()
                if tag == ck.TIME: want[:, 7] = 0                     # the padding byte comes back cleared
                if tag == ck.TIMESTAMP: want[:, 11] = 0
                assert np.array_equal(keys[k], want), (name, k)
    o = oracle.groupby_typed([(ck.STR, strs)])
    want_sum = np.bincount(o["reversemap"].astype(np.int64), weights=v.astype(np.float64), minlength=o["ngroups"]).astype(np.int64)
    allcodes = np.concatenate([r["str"][1] for r in res])
    assert np.array_equal(allcodes, o["reversemap"])                 # global first-occurrence ids, the reference's
    for r in res:
        nd, _, G, first, sums = r["str"]
        assert nd == o["ngroups"] == G
        assert np.array_equal(first, o["first_rows"].astype(np.int64)) and ck.i128_to_int(sums) == want_sum.tolist()


def test_rccl_transport_world_of_one(gpu_dev, oracle):
    """the RCCL path itself (librccl opened with dlopen, ncclGetUniqueId / ncclCommInitRank / ncclAllGather on the context's stream)
    with the only world a one-GPU box allows; the sharded call then equals the plain one"""
    import aquery2_amd
    rng = np.random.default_rng(3)
    n = 300_000
    id1, v1 = rng.integers(1, 101, n).astype(np.int32), rng.integers(1, 6, n).astype(np.int32)
    comm = aquery2_amd.Comm(gpu_dev, 0, 1, nccl_id=aquery2_amd.Comm.unique_id())
    gb = comm.groupby_agg_sharded([id1], [ck.RED_SUM, ck.RED_AVG], [v1, v1], row_base=5_000_000_000, hint=128, gmax=128)
    o = oracle.groupby([id1])
    assert gb.ngroups == o["ngroups"] == 100
    assert np.array_equal(gb.keys(0, np.int32), id1[o["first_rows"]])
    assert np.array_equal(gb.first_rows64(), o["first_rows"].astype(np.int64) + 5_000_000_000)
    assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v1, o))
    assert gu.same_bits(gb.result(1, ck.RED_AVG, ck.INT32), oracle.grouped_reduce(ck.RED_AVG, v1, o))
    gb.destroy()
    comm.destroy()


@pytest.fixture(scope="module")
def gpu_dev():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def _bench(args, env_extra, nproc=1, torchrun=False):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    if nproc > 1 or torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", "29511",
               os.path.join(root, "bench.py"), "--gpus", str(nproc)] + args
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["q1", "join"])
def test_bench_multi_rank_path_rehearsed_on_one_gpu(workload):
    """`bench.py --gpus 2` end to end with both ranks on GPU 0 and the all-gather over gloo (AQG_BENCH_REHEARSAL): the same
    aqg_groupby_agg_sharded / aqg_groupby_exchange calls the RCCL run makes, its own sum check included"""
    line = _bench(["--rows", "3e6", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--workload", workload], {"AQG_BENCH_REHEARSAL": "1"}, nproc=2)
    assert line["n_gpus"] == 2 and line["config"]["groups"] == 100 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0


def test_bench_self_merge_world_of_one():
    line = _bench(["--rows", "3e6", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], {"AQG_BENCH_SELFMERGE": "1"})
    assert line["n_gpus"] == 1 and line["config"]["groups"] == 100 and "secondary" not in line


def test_rccl_transport_after_torch_has_loaded_its_own_rocm_stack():
    """bench.py --gpus N imports torch (whose wheel bundles libamdhip64 / libhsa-runtime64 / librccl) BEFORE the library makes its
    communicator: the library must open the RCCL that sits next to the HIP runtime it is bound to (a fresh process, torch first)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import torch
torch.cuda.init(); torch.zeros(4, device="cuda"); torch.cuda.synchronize()
import sys; sys.path.insert(0, "tests")
import numpy as np, aquery2_amd, checker as ck
side = torch.cuda.Stream(device=0); torch.cuda.set_stream(side)
dev = aquery2_amd.Device(0, stream=torch.cuda.current_stream().cuda_stream)
comm = aquery2_amd.Comm(dev, 0, 1, nccl_id=aquery2_amd.Comm.unique_id())
k = (np.arange(100_000, dtype=np.int32) * 7919) % 101; v = np.arange(100_000, dtype=np.int32)
gb = comm.groupby_agg_sharded([k], [ck.RED_SUM, ck.RED_VAR], [v, v], row_base=0, hint=128, gmax=128)
want = np.bincount(k, weights=v.astype(np.float64), minlength=101)
got = ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32))
assert gb.ngroups == 101 and [int(x) for x in got] == [int(want[key]) for key in gb.keys(0, np.int32)]
print("OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


@pytest.mark.parametrize("force_torch", [False, True])
def test_bench_multi_rank_code_path_with_real_rccl_world_of_one(force_torch):
    """bench.py under torchrun with ONE rank and AQG_BENCH_FORCE_COMM: torch's process group on RCCL, the RCCL id broadcast, the
    library's own communicator (or, forced, the fallback transport over torch.distributed's all-gather) and the prepared sharded call --
    everything `--gpus N` runs except a second GPU"""
    env = {"AQG_BENCH_FORCE_COMM": "1"}
    if force_torch:
        env["AQG_BENCH_FORCE_TORCH_ALLGATHER"] = "1"
    line = _bench(["--rows", "3e6", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], env, nproc=1, torchrun=True)
    assert line["n_gpus"] == 1 and line["config"]["groups"] == 100
    assert ("torch.distributed" in line["config"]["exchange"]) == force_torch
