"""GPU parity on the BASELINE.json configurations' own shapes (seed-42 h2o / time-series columns generated on both
sides by the same counter-based generator), at sizes the oracle finishes in seconds, plus size-independent properties
at 1e8 rows."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu

pytestmark = pytest.mark.gpu
K = 100


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def cols(gpu, oracle, n, which):
    dev = {c: gpu.gen_column(c, 42, 0, n, n, K) for c in which}
    host = {c: oracle.gen_column(c, 42, 0, n, n, K) for c in which}
    return dev, host


def test_config1_q1_sum_by_single_int_key(gpu, oracle):
    n = 5_000_003
    dev, host = cols(gpu, oracle, n, [ck.GEN_ID1, ck.GEN_V1])
    gb = gpu.groupby_agg([dev[ck.GEN_ID1]], [ck.RED_SUM], [dev[ck.GEN_V1]], hint=128)
    o = oracle.groupby([host[ck.GEN_ID1]])
    assert gb.ngroups == o["ngroups"] == 100
    assert np.array_equal(gb.keys(0, np.int32), host[ck.GEN_ID1][o["first_rows"]])
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, host[ck.GEN_V1], o))


def test_config2_q5_high_cardinality_and_q2_multi_key(gpu, oracle):
    n = 3_000_000
    dev, host = cols(gpu, oracle, n, [ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID6, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3])
    # Q5: sum(v1), sum(v2), sum(v3) by id6   (id6 ~ U{1..n/K}: 30,000 groups)
    ops = [ck.RED_SUM] * 3
    gb = gpu.groupby_agg([dev[ck.GEN_ID6]], ops, [dev[ck.GEN_V1], dev[ck.GEN_V2], dev[ck.GEN_V3]], hint=n // K + 1024)
    o = oracle.groupby([host[ck.GEN_ID6]])
    assert gb.ngroups == o["ngroups"]
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    assert np.array_equal(gb.keys(0, np.int32), host[ck.GEN_ID6][o["first_rows"]])
    for j, c in enumerate((ck.GEN_V1, ck.GEN_V2)):
        assert gu.same_bits(gb.result(j, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, host[c], o))
    got, want = gb.result(2, ck.RED_SUM, ck.FLOAT), oracle.grouped_reduce(ck.RED_SUM, host[ck.GEN_V3], o)
    assert np.all(np.abs(got - want) <= np.maximum(1.0, np.abs(want)) * 1000 * 2.0 ** -52)   # <= ~100 rows per group
    # Q2: sum(v1) by id1, id2 (1e4 groups, two int32 keys packed into one 64-bit word)
    gb2 = gpu.groupby_agg([dev[ck.GEN_ID1], dev[ck.GEN_ID2]], [ck.RED_SUM], [dev[ck.GEN_V1]])
    o2 = oracle.groupby([host[ck.GEN_ID1], host[ck.GEN_ID2]])
    assert gb2.ngroups == o2["ngroups"] == 10000
    assert np.array_equal(gb2.first_rows(), o2["first_rows"])
    assert gu.same_bits(gb2.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, host[ck.GEN_V1], o2))


def test_config2_q10_six_keys(gpu, oracle):
    n = 300_000
    ids = [ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6]
    dev, host = cols(gpu, oracle, n, ids + [ck.GEN_V3])
    gb = gpu.groupby_agg([dev[c] for c in ids], [ck.RED_SUM, ck.RED_COUNT], [dev[ck.GEN_V3], dev[ck.GEN_V3]])
    o = oracle.groupby([host[c] for c in ids])
    assert gb.ngroups == o["ngroups"]
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    assert np.array_equal(gb.counts(), o["counts"])
    for k, c in enumerate(ids):
        assert np.array_equal(gb.keys(k, np.int32), host[c][o["first_rows"]])
    # sum(v3): float32 values accumulated in double; any summation order stays inside (n - 1) 2^-53 sum|x| of the reference's
    got, want = gb.result(0, ck.RED_SUM, ck.FLOAT).astype(np.float64), oracle.grouped_reduce(ck.RED_SUM, host[ck.GEN_V3], o).astype(np.float64)
    cnt = o["counts"].astype(np.float64)
    assert np.all(np.abs(got - want) <= np.maximum(cnt - 1, 0) * 2.0 ** -53 * np.abs(want) + 1e-300)
    assert np.array_equal(gb.result(1, ck.RED_COUNT, ck.FLOAT), oracle.grouped_reduce(ck.RED_COUNT, host[ck.GEN_V3], o))


def test_config3_moving_windows_over_ordered_series(gpu, oracle):
    """stock.a-style queries over a time series: max(price - mins(price)), avgs / sums windows w in {3,5,10,100}"""
    n = 4_000_001
    dev, host = cols(gpu, oracle, n, [ck.GEN_TIMESTAMP, ck.GEN_PRICE])
    p_d, p_h = dev[ck.GEN_PRICE], host[ck.GEN_PRICE]
    assert np.array_equal(host[ck.GEN_TIMESTAMP][:3], [1, 2, 3])
    # q2 of tests/stock.a
    d = gpu.ewise(ck.OP_SUB, p_d, gpu.scan(ck.SCAN_MINS, p_d, keep=True), keep=True)
    want = oracle.reduce(ck.RED_MAX, oracle.ewise(ck.OP_SUB, p_h, oracle.scan(ck.SCAN_MINS, p_h)))
    assert int(gpu.reduce(ck.RED_MAX, d)) == int(want)
    for w in (3, 5, 10, 100):
        assert gu.same_bits(gpu.scan(ck.SCAN_SUMW, p_d, w), oracle.scan(ck.SCAN_SUMW, p_h, w)), w
        assert gu.same_bits(gpu.scan(ck.SCAN_MINW, p_d, w), oracle.scan(ck.SCAN_MINW, p_h, w)), w
        assert gu.same_bits(gpu.scan(ck.SCAN_MAXW, p_d, w), oracle.scan(ck.SCAN_MAXW, p_h, w)), w
        a, b = gpu.scan(ck.SCAN_AVGW, p_d, w), oracle.scan(ck.SCAN_AVGW, p_h, w)
        # device: exact window mean rounded once; reference: floating recurrence drifting by ~ulp per step
        assert np.all(np.abs(a - b) <= 4.0 * np.spacing(500.0) * (np.arange(n) + 2)), w
    assert gu.same_bits(gpu.scan(ck.SCAN_SUMS, p_d), oracle.scan(ck.SCAN_SUMS, p_h))
    assert gu.same_bits(gpu.scan(ck.SCAN_AVGS, p_d), oracle.scan(ck.SCAN_AVGS, p_h))
    assert gu.same_bits(gpu.scan(ck.SCAN_DELTAS, p_d), oracle.scan(ck.SCAN_DELTAS, p_h))
    assert gu.same_bits(gpu.scan(ck.SCAN_RATIOW, p_d, 1), oracle.scan(ck.SCAN_RATIOW, p_h, 1))


def test_config4_join_then_groupby(gpu, oracle):
    """fact JOIN small(id4, w) ON id4, then sum(v1 * w) by id1 -- composed from the C-ABI pieces"""
    n = 2_000_000
    dev, host = cols(gpu, oracle, n, [ck.GEN_ID1, ck.GEN_ID4, ck.GEN_V1])
    rng = np.random.default_rng(4)
    dim_key = rng.permutation(np.arange(1, K + 1, dtype=np.int32))
    dim_w = rng.integers(1, 50, K).astype(np.int32)
    look = gpu.join_lookup(dim_key, dev[ck.GEN_ID4])                      # build row of every fact row
    pos = np.zeros(K + 1, np.uint32)
    pos[dim_key] = np.arange(K, dtype=np.uint32)
    assert np.array_equal(look, pos[host[ck.GEN_ID4]])
    w_of_row = gpu.gather(dim_w, look)
    prod = gpu.ewise(ck.OP_MUL, dev[ck.GEN_V1], w_of_row, ot=ck.INT64, keep=True)   # aqop_mul with an int64 Ret
    gb = gpu.groupby_agg([dev[ck.GEN_ID1]], [ck.RED_SUM], [prod], hint=128)
    o = oracle.groupby([host[ck.GEN_ID1]])
    want_prod = host[ck.GEN_V1].astype(np.int64) * dim_w[pos[host[ck.GEN_ID4]]].astype(np.int64)
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT64), oracle.grouped_reduce(ck.RED_SUM, want_prod, o))


def test_config4_fused_star_join_groupby_sum(gpu, oracle):
    """aqg_join_groupby_sum == the composed pipeline == the oracle, incl. unmatched fact rows (inner join), duplicate and
    sentinel dimension keys, negative values and the first-occurrence order among JOINED rows"""
    rng = np.random.default_rng(44)
    for n, nb, lo, wlim in ((2_000_003, 100, 1, 2**31 - 1), (2_000_003, 100, 1, 60), (50_001, 37, -5, 2**31 - 1), (50_001, 37, -5, 7), (7, 3, 0, 2**31 - 1),
                            (3, 4096, 0, 3)):        # small |w|: the one-accumulator path; large: the split halves
        fk = rng.integers(lo, lo + nb + 20, n).astype(np.int32)                # some keys have no partner
        gkey = rng.integers(-40, 40, n).astype(np.int32)
        val = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
        dim_key = rng.permutation(np.arange(lo, lo + nb, dtype=np.int32))
        if nb == 37:
            dim_key[5] = dim_key[30]                                             # duplicate: the lowest row wins
            dim_key[7] = np.iinfo(np.int32).min                                 # the LDS empty mark as a key
            fk[::11] = np.iinfo(np.int32).min
        dim_w = rng.integers(-wlim - 1, wlim, nb).astype(np.int32)
        gb = gpu.join_groupby_sum(dim_key, dim_w, fk, gkey, val)
        # oracle: lookup (lowest build row), drop unmatched rows, exact products, grouped sum in first-occurrence order
        first = {}
        for r, k in enumerate(dim_key.tolist()):
            first.setdefault(k, r)
        look = np.array([first.get(k, -1) for k in fk.tolist()], dtype=np.int64)
        m = look >= 0
        rows = np.nonzero(m)[0]
        if len(rows) == 0:
            assert gb.ngroups == 0
            continue
        prod = [int(v) * int(w) for v, w in zip(val[m].tolist(), dim_w[look[m]].tolist())]
        o = oracle.groupby([np.ascontiguousarray(gkey[m])])
        assert gb.ngroups == o["ngroups"]
        assert np.array_equal(gb.keys(0, np.int32), gkey[m][o["first_rows"]])
        assert np.array_equal(gb.first_rows(), rows[o["first_rows"]].astype(np.uint32))      # row ids of the FACT table
        want = [0] * o["ngroups"]
        for g, p_ in zip(o["reversemap"].tolist(), prod):
            want[g] += p_
        assert ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT64)) == want
        gb.destroy()


def test_config4_fused_unsigned(gpu):
    rng = np.random.default_rng(45)
    n, nb = 300_000, 64
    fk = rng.integers(0, nb, n).astype(np.uint32)
    gkey = rng.integers(0, 9, n).astype(np.uint32)
    val = rng.integers(0, 2**32 - 1, n, dtype=np.uint64).astype(np.uint32)
    dim_key = np.arange(nb, dtype=np.uint32)
    dim_w = rng.integers(0, 2**32 - 1, nb, dtype=np.uint64).astype(np.uint32)
    gb = gpu.join_groupby_sum(dim_key, dim_w, fk, gkey, val)
    want = {}
    for g, v, f in zip(gkey.tolist(), val.tolist(), fk.tolist()):
        want[g] = want.get(g, 0) + v * int(dim_w[f])
    keys = gb.keys(0, np.uint32).tolist()
    assert ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.UINT64)) == [want[k] for k in keys]


def test_full_size_properties_q5_and_windows(gpu):
    """1e8 rows: Q5 invariants (sums of group sums, counts) and window invariants"""
    n = 100_000_000
    id6 = gpu.gen_column(ck.GEN_ID6, 42, 0, n, n, K)
    v1 = gpu.gen_column(ck.GEN_V1, 42, 0, n, n, K)
    v2 = gpu.gen_column(ck.GEN_V2, 42, 0, n, n, K)
    gb = gpu.groupby_agg([id6], [ck.RED_SUM, ck.RED_SUM, ck.RED_COUNT], [v1, v2, v1], hint=n // K + 1024)
    assert gb.ngroups <= n // K and gb.ngroups > 0.99 * (n // K)
    s1 = gb.result(0, ck.RED_SUM, ck.INT32)
    s2 = gb.result(1, ck.RED_SUM, ck.INT32)
    assert int(s1["lo"].astype(np.int64).sum()) == int(gpu.reduce(ck.RED_SUM, v1)) and not s1["hi"].any()
    assert int(s2["lo"].astype(np.int64).sum()) == int(gpu.reduce(ck.RED_SUM, v2))
    assert int(gb.result(2, ck.RED_COUNT, ck.INT32).sum()) == n
    first = gb.first_rows().astype(np.int64)
    assert np.all(np.diff(first) > 0)
    price = gpu.gen_column(ck.GEN_PRICE, 42, 0, n, n, K)
    for w in (5, 100):
        mw = gpu.scan(ck.SCAN_MINW, price, w, keep=True)
        xw = gpu.scan(ck.SCAN_MAXW, price, w, keep=True)
        assert int(gpu.reduce(ck.RED_MIN, gpu.ewise(ck.OP_SUB, price, mw, keep=True))) >= 0
        assert int(gpu.reduce(ck.RED_MIN, gpu.ewise(ck.OP_SUB, xw, price, keep=True))) >= 0
        mw.free(); xw.free()
    sw = gpu.scan(ck.SCAN_SUMW, price, n, keep=True)      # window = whole column: last element is the column sum
    last = sw.to_host()[-1]
    assert (int(last["hi"]) << 64) + int(last["lo"]) == int(gpu.reduce(ck.RED_SUM, price))


def test_h2o_q5_q3_q7_at_the_full_1e9_rows(gpu):
    """BASELINE config 2 at its full size, through the plan the bench takes (two levels of range partitions, value columns inside the key word, the
    direct-indexed aggregation): invariants that do not need the oracle -- the plan bits, the group count, every key once, groups in first-occurrence
    order, sums of group sums / counts against whole-column reductions, avg = sum / count, max and min inside the columns' ranges and met"""
    from aquery2_amd import capi
    n = 1_000_000_000
    id6, id3, v1, v2, v3 = (gpu.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID6, ck.GEN_ID3, ck.GEN_V1, ck.GEN_V2, ck.GEN_V3))
    want = capi.PLAN_PART_TWO | capi.PLAN_PACKED_VALUES | capi.PLAN_RANGE_PARTITIONS
    # Q5: sum(v1), sum(v2), sum(v3), count BY id6
    gb = gpu.groupby_agg([id6], [ck.RED_SUM, ck.RED_SUM, ck.RED_SUM, ck.RED_COUNT], [v1, v2, v3, v1], hint=n // K + 1024)
    assert gb.plan & ~capi.PLAN_SORTED_TAIL == want, gb.plan      # (tools/fuzz_more.sh forces the ordering tail onto every partition plan)
    G = gb.ngroups
    assert 0.999 * (n // K) < G <= n // K
    keys = gb.keys(0, np.int32)
    assert keys.min() >= 1 and keys.max() <= n // K and np.unique(keys).size == G                   # every key once
    first = gb.first_rows().astype(np.int64)
    assert np.all(np.diff(first) > 0) and first[0] == 0 and first[-1] < n                           # first-occurrence order
    s1, s2 = gb.result(0, ck.RED_SUM, ck.INT32), gb.result(1, ck.RED_SUM, ck.INT32)
    assert int(s1["lo"].astype(np.int64).sum()) == int(gpu.reduce(ck.RED_SUM, v1)) and not s1["hi"].any()
    assert int(s2["lo"].astype(np.int64).sum()) == int(gpu.reduce(ck.RED_SUM, v2)) and not s2["hi"].any()
    cnt = gb.result(3, ck.RED_COUNT, ck.INT32)
    assert int(cnt.sum()) == n
    s3 = gb.result(2, ck.RED_SUM, ck.FLOAT)
    tot3 = float(gpu.reduce(ck.RED_SUM, v3))
    assert abs(float(s3.sum()) - tot3) <= 1e-9 * abs(tot3)                                          # (float sums: order-dependent in the last bits)
    gb.destroy()
    # Q3: sum(v1), avg(v3) BY id3 -- avg against the sums of Q5's kind
    gb = gpu.groupby_agg([id3], [ck.RED_SUM, ck.RED_AVG, ck.RED_SUM, ck.RED_COUNT], [v1, v3, v3, v3], hint=n // K + 1024)
    assert gb.plan & ~capi.PLAN_SORTED_TAIL == want, gb.plan      # (tools/fuzz_more.sh forces the ordering tail onto every partition plan)
    avg, s3, cnt = gb.result(1, ck.RED_AVG, ck.FLOAT), gb.result(2, ck.RED_SUM, ck.FLOAT), gb.result(3, ck.RED_COUNT, ck.FLOAT)
    assert np.array_equal(avg, s3 / cnt.astype(np.float64)) and int(cnt.sum()) == n
    gb.destroy()
    # Q7: max(v1), min(v2) BY id3 -- no value plane at all
    gb = gpu.groupby_agg([id3], [ck.RED_MAX, ck.RED_MIN], [v1, v2], hint=n // K + 1024)
    assert gb.plan & ~capi.PLAN_SORTED_TAIL == want, gb.plan      # (tools/fuzz_more.sh forces the ordering tail onto every partition plan)
    mx, mn = gb.result(0, ck.RED_MAX, ck.INT32), gb.result(1, ck.RED_MIN, ck.INT32)
    assert mx.max() == int(gpu.reduce(ck.RED_MAX, v1)) and mx.min() >= int(gpu.reduce(ck.RED_MIN, v1))
    assert mn.min() == int(gpu.reduce(ck.RED_MIN, v2)) and mn.max() <= int(gpu.reduce(ck.RED_MAX, v2))
    gb.destroy()
    for c in (id6, id3, v1, v2, v3): c.free()


def test_shard_exchange_pack_and_merge(gpu, oracle):
    """SURVEY 8e on one device: three row-range shards grouped separately, their tables packed (aqg_groupby_pack) into the
    buffer an all_gather would fill, merged by aqg_groupby_merge_packed == the group-by of the whole table"""
    import aquery2_amd
    rng = np.random.default_rng(81)
    n, world = 900_001, 3
    key32 = rng.integers(-20, 21, n).astype(np.int32)
    key32[: n // 3] = rng.integers(0, 5, n // 3)                 # the first shard sees only a few of the keys
    # 8-byte keys: -0x7FFFFFFFFFFFFFFF is the empty mark of the one-workgroup merge, -1 that of the device tables
    lut = np.array([-0x7FFFFFFFFFFFFFFF, -1, 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max] + [int(v) * 3_000_000_019 for v in range(-18, 18)], dtype=np.int64)
    key64 = lut[key32.astype(np.int64) + 20]
    val = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    bounds = [0, n // 3, n // 3, n]                             # the second shard is EMPTY
    # world x gmax <= 2048: one workgroup merges (merge_small_kernel); above: concatenation + the generic group-by
    for key, tag, gmax in ((key32, ck.INT32, 64), (key32, ck.INT32, 1024), (key64, ck.INT64, 64)):
        gathered = gpu.empty(world * (gmax + 1) * 2, np.int64)
        o = oracle.groupby([key])
        for op in (ck.RED_SUM, ck.RED_MIN, ck.RED_MAX, ck.RED_COUNT):
            for r in range(world):
                lo, hi = bounds[r], bounds[r + 1]
                gb = gpu.groupby_agg([key[lo:hi]], [op], [val[lo:hi]])
                gpu.groupby_pack(gb, 0, gmax, gathered.ptr + r * (gmax + 1) * 2 * 8)
                gb.destroy()
            merged = gpu.groupby_merge_packed(gathered.ptr, world, gmax, tag, op)
            assert merged.ngroups == o["ngroups"]
            assert np.array_equal(merged.keys(0, key.dtype), key[o["first_rows"]])            # global first-occurrence order
            want = oracle.grouped_reduce(op, val, o)
            got = merged.result(0, ck.RED_SUM if op == ck.RED_COUNT else op, ck.INT64)
            if op in (ck.RED_SUM, ck.RED_COUNT):
                assert ck.i128_to_int(got) == (ck.i128_to_int(want) if op == ck.RED_SUM else [int(c) for c in want]), (gmax, op)
            else:
                assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (gmax, op)
            merged.destroy()
        gathered.free()
    # sums that need more than 64 bits: int64 partials near the limits, merged exactly into 128 bits
    big = np.array([2**62, 2**62, 2**62, -2**62, 2**63 - 1, -2**63], dtype=np.int64)
    gathered = gpu.empty(6 * 2 * 2, np.int64)
    for r in range(6):
        gb = gpu.groupby_agg([np.array([7], np.int32)], [ck.RED_MAX], [big[r:r + 1]])       # MAX of one row: the value itself
        gpu.groupby_pack(gb, 0, 1, gathered.ptr + r * 2 * 2 * 8)
        gb.destroy()
    merged = gpu.groupby_merge_packed(gathered.ptr, 6, 1, ck.INT32, ck.RED_SUM)
    assert merged.ngroups == 1 and ck.i128_to_int(merged.result(0, ck.RED_SUM, ck.INT64)) == [sum(int(v) for v in big)]
    merged.destroy()


def test_maximum_row_count(gpu):
    """n = AQG_MAX_ROWS - 1 = 2^32 - 2^20 - 1 rows (sizes are uint32_t, server/vector_type.hpp:66; the last 2^20 counts are
    rejected, see include/aqg.h): every index computation near the 32-bit limit.
    Properties only (17 GB per column): Q1 sums add up to the column sum, counts to n; scans / windows / shifts agree at
    the far end with values recomputed on the host from the device's own tail; unaligned tail (n % 4 == 3)."""
    import aquery2_amd
    n = 2**32 - 2**20 - 1
    with pytest.raises(aquery2_amd.capi.AqgError, match="AQG_MAX_ROWS"):
        gpu.gen_column(ck.GEN_V1, 42, 0, 2**32 - 1, 2**32 - 1, K)
    id1 = gpu.gen_column(ck.GEN_ID1, 42, 0, n, n, K)
    v1 = gpu.gen_column(ck.GEN_V1, 42, 0, n, n, K)
    total = int(gpu.reduce(ck.RED_SUM, v1))
    assert n <= total <= 5 * n
    gb = gpu.groupby_agg([id1], [ck.RED_SUM, ck.RED_COUNT], [v1, v1], hint=128)
    assert gb.ngroups == 100
    assert sum(ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32))) == total
    assert int(gb.result(1, ck.RED_COUNT, ck.INT32).astype(np.uint64).sum()) == n
    assert int(gb.counts().astype(np.uint64).sum()) == n
    assert np.array_equal(np.sort(gb.keys(0, np.int32)), np.arange(1, 101))
    assert gb.first_rows().max() < 5000
    gb.destroy()
    # last rows of running sum / max / window sum / deltas against a host recomputation of the tail
    tail = 1000
    sums = gpu.scan(ck.SCAN_SUMS, v1, keep=True)
    got_last = ck.i128_to_int(aquery2_amd.DevBuf(gpu, sums.ptr + (n - tail) * 16, ck.I128, tail, owned=False).to_host())
    assert got_last[-1] == total
    v_tail = aquery2_amd.DevBuf(gpu, v1.ptr + (n - tail) * 4, np.int32, tail, owned=False).to_host().astype(np.int64)
    assert got_last == list(np.int64(total) - np.concatenate([np.cumsum(v_tail[::-1])[::-1][1:], [0]]))
    sums.free()
    w = gpu.scan(ck.SCAN_SUMW, v1, 7, keep=True)
    w_tail = ck.i128_to_int(aquery2_amd.DevBuf(gpu, w.ptr + (n - 100) * 16, ck.I128, 100, owned=False).to_host())
    ref = np.convolve(v_tail, np.ones(7, dtype=np.int64))[: tail][-100:]
    assert w_tail == ref.tolist()
    w.free()
    dl = gpu.scan(ck.SCAN_DELTAS, v1, keep=True)
    d_tail = aquery2_amd.DevBuf(gpu, dl.ptr + (n - 100) * 4, np.int32, 100, owned=False).to_host()
    assert np.array_equal(d_tail, np.diff(v_tail)[-100:].astype(np.int32))
    dl.free()
    mx = gpu.scan(ck.SCAN_MAXS, v1, keep=True)
    assert int(aquery2_amd.DevBuf(gpu, mx.ptr + (n - 1) * 4, np.int32, 1, owned=False).to_host()[0]) == 5
    mx.free()
    plus = gpu.ewise(ck.OP_ADD, v1, np.int32(1), keep=True)                 # int32 + int32 -> int32
    assert int(gpu.reduce(ck.RED_SUM, plus)) == total + n
    plus.free()
    import ctypes
    kept = ctypes.c_uint32()
    mask = gpu.ewise(ck.OP_GT, v1, np.int32(4), keep=True)
    out = gpu.empty(n, np.int32)
    gpu._chk(gpu.lib.aqg_compact(gpu.ctx, ck.INT32, ctypes.c_void_p(v1.ptr), ctypes.c_void_p(mask.ptr), ctypes.c_uint32(n), ctypes.c_void_p(out.ptr),
                                 ctypes.byref(kept)), "aqg_compact")
    fives = aquery2_amd.DevBuf(gpu, out.ptr, np.int32, kept.value, owned=False)
    assert int(gpu.reduce(ck.RED_MIN, fives)) == 5 and int(gpu.reduce(ck.RED_SUM, fives)) == 5 * kept.value
    assert abs(kept.value / n - 0.2) < 1e-3


def test_scans_and_windows_over_row_range_shards(gpu, oracle):
    """three row-range shards of one ordered series; every shard gets only the halo an exchange would bring
    (last w-1 rows / boundary rows / the fold of everything before) and its results equal the whole-column scan"""
    import aquery2_amd
    from aquery2_amd import shard
    n = 300_007
    price = oracle.gen_column(ck.GEN_PRICE, 42, 0, n, n, K)
    bounds = [0, 100_000, 100_050, n]                         # the middle shard is shorter than some of the windows' halos allow: w <= 51 below
    dev_shards = [gpu.to_device(np.ascontiguousarray(price[bounds[r]:bounds[r + 1]])) for r in range(3)]
    def tail(r, h):                                             # what exchange_tails would deliver to rank r + 1
        lo = max(bounds[r + 1] - h, 0)
        return gpu.to_device(np.ascontiguousarray(price[lo:bounds[r + 1]]))
    for name, w in (("sumw", 7), ("avgw", 50), ("minw", 3), ("maxw", 51), ("ratiow", 1)):
        op = ck.SCAN_NAMES[name]
        h = w if name == "ratiow" else w - 1
        want = oracle.scan(op, price, w)
        for r in range(3):
            halo = tail(r - 1, h) if r else None
            got = shard.window_scan_with_halo(gpu, op, dev_shards[r], halo, w).to_host()
            ref = want[bounds[r]:bounds[r + 1]]
            if name == "avgw":
                assert np.all(np.abs(got - ref) <= 1e-9 * np.abs(ref) * (np.arange(bounds[r], bounds[r + 1]) + 2)), (name, r)
            else:
                assert gu.same_bits(got, ref), (name, r)
    for name in ("deltas", "prev", "aggnext"):
        op = ck.SCAN_NAMES[name]
        want = oracle.scan(op, price)
        for r in range(3):
            prev_last = gpu.to_device(price[bounds[r] - 1:bounds[r]].copy()) if r else None
            next_first = gpu.to_device(price[bounds[r + 1]:bounds[r + 1] + 1].copy()) if r < 2 else None
            got = shard.shift_scan_with_neighbours(gpu, op, dev_shards[r], prev_last, next_first).to_host()
            assert gu.same_bits(got, want[bounds[r]:bounds[r + 1]]), (name, r)
    for name, fold in (("mins", np.min), ("maxs", np.max)):
        op = ck.SCAN_NAMES[name]
        want = oracle.scan(op, price)
        for r in range(3):
            carry = gpu.to_device(np.array([fold(price[:bounds[r]])], dtype=price.dtype)) if r else None
            got = shard.running_minmax_with_carry(gpu, op, dev_shards[r], carry).to_host()
            assert gu.same_bits(got, want[bounds[r]:bounds[r + 1]]), (name, r)
    # running sums / avgs: the carry is the exact sum of every earlier row (128 bits for integer columns) and avgs also need
    # the number of earlier rows; integer results are bit-identical to the whole-column scan
    for dt in (np.int32, np.int64, np.uint32, np.float64):
        col = price.astype(dt) if dt != np.int64 else price.astype(np.int64) * 3_000_000_007 - 5_000_000_000_000
        for name in ("sums", "avgs"):
            op = ck.SCAN_NAMES[name]
            want = oracle.scan(op, col)
            for r in range(3):
                part = np.ascontiguousarray(col[bounds[r]:bounds[r + 1]])
                before = col[:bounds[r]]
                carry = float(np.sum(before)) if dt == np.float64 else sum(int(v) for v in before.tolist())
                if r == 0:
                    carry = -0.0 if dt == np.float64 else 0
                got = shard.running_sums_with_carry(gpu, op, gpu.to_device(part), carry, bounds[r]).to_host()
                ref = want[bounds[r]:bounds[r + 1]]
                if dt == np.float64:
                    assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(np.abs(ref), 1.0)), (name, r)
                else:
                    assert gu.same_bits(got, ref), (dt, name, r)


def test_merge_packed_against_hand_packed_tables(gpu):
    """aqg_groupby_merge_packed fed with hand-packed shard tables ({count, 0; key, partial} int64 pairs): both the one-workgroup
    merge (world x gmax <= 2048) and the concatenation + generic group-by, checked against a plain dictionary merge in rank order"""
    rng = np.random.default_rng(2024)
    pool = np.array([0, -1, 1, -0x7FFFFFFFFFFFFFFF, np.iinfo(np.int64).min, np.iinfo(np.int64).max] + [int(v) for v in rng.integers(-2**62, 2**62, 300)], dtype=np.int64)
    for world, gmax in ((1, 1), (2, 7), (8, 128), (16, 128), (5, 700), (64, 32), (3, 2000)):
        for op in (ck.RED_SUM, ck.RED_MIN, ck.RED_MAX):
            host = np.zeros((world, gmax + 1, 2), dtype=np.int64)
            merged_ref = {}
            for r in range(world):
                cnt = int(rng.integers(0, min(gmax, len(pool)) + 1))
                keys = rng.choice(pool, cnt, replace=False)
                vals = rng.integers(-2**63, 2**63 - 1, cnt, dtype=np.int64) if op != ck.RED_SUM else rng.integers(-2**62, 2**62, cnt, dtype=np.int64) * 2
                host[r, 0, 0] = cnt
                host[r, 1:cnt + 1, 0] = keys
                host[r, 1:cnt + 1, 1] = vals
                for k, v in zip(keys.tolist(), vals.tolist()):
                    if k not in merged_ref:
                        merged_ref[k] = v
                    else:
                        merged_ref[k] = merged_ref[k] + v if op == ck.RED_SUM else (min(merged_ref[k], v) if op == ck.RED_MIN else max(merged_ref[k], v))
            dev = gpu.to_device(host.reshape(-1))
            merged = gpu.groupby_merge_packed(dev.ptr, world, gmax, ck.INT64, op)
            assert merged.ngroups == len(merged_ref), (world, gmax, op)
            assert merged.keys(0, np.int64).tolist() == list(merged_ref.keys()), (world, gmax, op)      # dict order = first occurrence
            got = merged.result(0, op, ck.INT64)
            if op == ck.RED_SUM:
                assert ck.i128_to_int(got) == list(merged_ref.values()), (world, gmax)
            else:
                assert got.astype(np.int64).tolist() == list(merged_ref.values()), (world, gmax, op)
            merged.destroy(); dev.free()
    # a corrupt header is refused by both paths
    import aquery2_amd
    for world, gmax in ((2, 8), (3, 2000)):
        host = np.zeros((world, gmax + 1, 2), dtype=np.int64)
        host[1, 0, 0] = gmax + 1
        dev = gpu.to_device(host.reshape(-1))
        with pytest.raises(aquery2_amd.capi.AqgError):
            gpu.groupby_merge_packed(dev.ptr, world, gmax, ck.INT64, ck.RED_SUM)
        dev.free()
