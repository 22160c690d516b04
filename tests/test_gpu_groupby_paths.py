"""GPU parity for the group-by paths that only large inputs reach (n >= 2^20 rows):
  * the big LDS table with several passes over the rows (a few thousand to ~25,000 groups; agg_kernel<BLOCK = 512 / 1024>)
  * the partition pipeline (more groups than that; partition.hip): 4- and 8-byte packed keys, the all-ones key, 1/2/4/8-byte
    values, row-index operands (FIRST / LAST through aqg_grouped_reduce)
against the oracle's first-occurrence group order, keys, first rows, counts and every aggregate."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu
from test_gpu_basic import check_agg, rand

pytestmark = pytest.mark.gpu
N = 1_300_003


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def vals_of(rng, n):
    return [rand(rng, np.int32, n, small=True), np.round(rng.uniform(0, 100, n), 6).astype(np.float32), rand(rng, np.int16, n), rand(rng, np.float64, n)]


@pytest.mark.parametrize("hint", [0, 10000, 20000])
def test_dense_domain_direct_indexed(gpu, oracle, hint):
    """100 x 100 key values: the product of the column ranges fits the LDS tables -> direct-indexed (dense.hip)"""
    rng = np.random.default_rng(21 + hint)
    keys = [rng.integers(-50, 50, N).astype(np.int32), rng.integers(1000, 1100, N).astype(np.int32)]
    check_agg(gpu, oracle, keys, vals_of(rng, N), hint)


def test_dense_domain_wide_tuple_and_passes(gpu, oracle):
    """five narrow columns (a tuple wider than 64 bits once an int64 column is in it) with a 40 x 12 x 5 x 2 x 9 = 43,200-slot
    domain: several passes of the direct-indexed table; keys are emitted from a representative row"""
    rng = np.random.default_rng(23)
    keys = [rng.integers(-20, 20, N).astype(np.int64), rng.integers(0, 12, N).astype(np.uint8), rng.integers(-2, 3, N).astype(np.int16),
            rng.integers(0, 2, N).astype(np.int32), rng.integers(100, 109, N).astype(np.uint16)]
    check_agg(gpu, oracle, keys, [rand(rng, np.int32, N, small=True)], 43_200)


@pytest.mark.parametrize("hint", [0, 10000, 20000])
def test_big_lds_table_multipass(gpu, oracle, hint):
    """sparse key values (huge ranges, so no dense domain): the hashed 150 KB LDS table with several passes"""
    rng = np.random.default_rng(21 + hint)
    keys = [rng.integers(0, 100, N).astype(np.int32) * 9_999_991, rng.integers(0, 100, N).astype(np.int32) * -7_777_777]
    check_agg(gpu, oracle, keys, vals_of(rng, N), hint)


def test_big_lds_table_few_accumulators(gpu, oracle):
    """1 and 2 accumulators take the 1024-thread instantiation; one 4-byte key takes the K32 slot layout"""
    rng = np.random.default_rng(22)
    for keys in ([rng.integers(0, 9000, N).astype(np.int32) * 100_003], [rng.integers(0, 90, N).astype(np.int16), rng.integers(0, 90, N).astype(np.int32) * 1_000_003],
                 [rng.integers(0, 9000, N).astype(np.int32)]):
        o = oracle.groupby(keys)
        v = rand(rng, np.int32, N, small=True)
        w = rand(rng, np.float64, N)
        for ops, vs in (([ck.RED_SUM], [v]), ([ck.RED_SUM, ck.RED_MIN], [v, w]), ([ck.RED_COUNT], [v])):
            gb = gpu.groupby_agg(keys, ops, vs, hint=9000)
            assert gb.ngroups == o["ngroups"]
            assert np.array_equal(gb.first_rows(), o["first_rows"])
            for j, (op, x) in enumerate(zip(ops, vs)):
                assert gu.same_bits(gb.result(j, op, ck.tag_of(x)), oracle.grouped_reduce(op, x, o)), (op, x.dtype)
            gb.destroy()


@pytest.mark.parametrize("card", [60_000, 400_000])
def test_partition_path_one_key(gpu, oracle, card):
    rng = np.random.default_rng(card)
    keys = [rng.integers(-card // 2, card // 2, N).astype(np.int32)]
    check_agg(gpu, oracle, keys, vals_of(rng, N), card)


def test_partition_path_wide_keys_and_sentinels(gpu, oracle):
    rng = np.random.default_rng(5)
    k64 = (rng.integers(0, 150_000, N).astype(np.int64) << 21) - 1          # includes -1: the packed key equal to the empty mark
    assert (k64 == -1).any()
    v = [rand(rng, np.int32, N, small=True), rng.integers(0, 255, N).astype(np.uint8)]
    check_agg(gpu, oracle, [k64], v, 200_000)
    k2 = [rng.integers(0, 700, N).astype(np.int32), rng.integers(-300, 300, N).astype(np.int32)]
    check_agg(gpu, oracle, k2, v, 500_000)
    # 8-byte values: sums keep two 64-bit accumulators (low / high halves), so fewer aggregates fit one call
    x = rand(rng, np.int64, N)
    o = oracle.groupby(k2)
    ops = [ck.RED_SUM, ck.RED_MIN, ck.RED_MAX, ck.RED_AVG]
    gb = gpu.groupby_agg(k2, ops, [x] * 4, hint=500_000)
    assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
    for j, op in enumerate(ops):
        assert gu.same_bits(gb.result(j, op, ck.INT64), oracle.grouped_reduce(op, x, o)), op
    gb.destroy()
    k3 = [rng.integers(0, 300, N).astype(np.int16), rng.integers(0, 200, N).astype(np.uint8), rng.integers(0, 2, N).astype(np.int32)]
    check_agg(gpu, oracle, k3, v[:1], 150_000)


def test_partition_path_retries_from_a_small_hint(gpu, oracle):
    rng = np.random.default_rng(6)
    keys = [rng.integers(0, 250_000, N).astype(np.int32)]
    check_agg(gpu, oracle, keys, [rand(rng, np.int32, N, small=True)], 0)


def test_grouped_reduce_first_last_high_cardinality(gpu, oracle):
    """FIRST / LAST = arg-min / arg-max of the row id, which the partition pipeline carries with every record"""
    rng = np.random.default_rng(8)
    keys = [rng.integers(0, 180_000, N).astype(np.int32)]
    x = rand(rng, np.int32, N)
    o = oracle.groupby(keys)
    g = gpu.groupby_build(keys)
    assert g.ngroups == o["ngroups"]
    assert np.array_equal(g.reversemap(), o["reversemap"])
    for op in (ck.RED_FIRST, ck.RED_LAST, ck.RED_SUM, ck.RED_MAX):
        assert gu.same_bits(gpu.grouped_reduce(g, op, x), oracle.grouped_reduce(op, x, o)), op
    g.destroy()


def test_build_over_dense_domain_and_large_count_histograms(gpu, oracle):
    """aqg_groupby_build (reversemap + counts + postproc) where the first pass takes the dense plan (two narrow key columns), and where
    the second pass keeps 20,000 group counts in an LDS histogram (sparse keys)"""
    rng = np.random.default_rng(61)
    for keys in ([rng.integers(0, 100, N).astype(np.int32), rng.integers(-50, 50, N).astype(np.int16)],
                 [rng.integers(0, 20_000, N).astype(np.int32) * 104_729],
                 [rng.integers(0, 40, N).astype(np.int64), rng.integers(0, 30, N).astype(np.uint8), rng.integers(0, 9, N).astype(np.int32)]):
        o = oracle.groupby(keys)
        g = gpu.groupby_build(keys)
        assert g.ngroups == o["ngroups"]
        assert np.array_equal(g.reversemap(), o["reversemap"])
        assert np.array_equal(g.counts(), o["counts"])
        assert np.array_equal(g.first_rows(), o["first_rows"])
        x = rand(rng, np.int32, N, small=True)
        assert gu.same_bits(gpu.grouped_reduce(g, ck.RED_SUM, x), oracle.grouped_reduce(ck.RED_SUM, x, o))
        g.destroy()


def test_dense_domain_sampled_ranges_miss_and_recover(gpu, oracle):
    """>= 2^22 rows: the dense plan takes the key ranges from the first 2^20 rows; values that only appear later must trigger the
    exact-range re-run (and give the same result as ever)"""
    rng = np.random.default_rng(71)
    n = 4_400_011
    a = rng.integers(0, 60, n).astype(np.int32)
    b = rng.integers(0, 70, n).astype(np.int16)
    a[3_000_000:] += 25                                       # new maxima after the sampled prefix
    b[4_399_000:] -= 9                                        # new minima at the very end
    v = rand(rng, np.int32, n, small=True)
    o = oracle.groupby([a, b])
    gb = gpu.groupby_agg([a, b], [ck.RED_SUM, ck.RED_COUNT], [v, v], hint=8000)
    assert gb.ngroups == o["ngroups"]
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    assert np.array_equal(gb.keys(0, np.int32), a[o["first_rows"]]) and np.array_equal(gb.keys(1, np.int16), b[o["first_rows"]])
    assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v, o))
    # the handle remembers: a second call on the same handle goes straight to exact ranges (same answer)
    gb2 = gpu.groupby_agg([a, b], [ck.RED_SUM, ck.RED_COUNT], [v, v], hint=8000, handle=gb)
    assert gb2.ngroups == o["ngroups"] and gu.same_bits(gb2.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v, o))
    g = gpu.groupby_build([a, b], hint=8000)
    assert np.array_equal(g.reversemap(), o["reversemap"]) and np.array_equal(g.counts(), o["counts"])
    g.destroy()


@pytest.mark.parametrize("n", [77, 1_000_003])
def test_fast_kernel_8_byte_value_columns(gpu, oracle, n):
    """one or two 4-byte keys (or one 8-byte key) with 1..4 accumulators over 8-byte columns (agg32_kernel<.., VW = 8>): doubles, int64 sums that
    need all 128 bits of the reference's __int128 (two 64-bit half sums), order-preserving MIN / MAX of negative int64"""
    rng = np.random.default_rng(5 + n)
    f64 = np.round(rng.uniform(-1e6, 1e6, n), 3)
    i64 = rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
    u64 = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    i32 = rand(rng, np.int32, n, small=True)
    u32 = rng.integers(0, 1 << 32, n, dtype=np.uint32)
    f32 = np.round(rng.uniform(0, 100, n), 6).astype(np.float32)
    i8, u8 = rng.integers(-128, 128, n).astype(np.int8), rng.integers(0, 256, n).astype(np.uint8)
    i16, u16 = rng.integers(-2**15, 2**15, n).astype(np.int16), rng.integers(0, 2**16, n).astype(np.uint16)
    b8 = (rng.random(n) < 0.4).astype(np.uint8)
    shapes = [(["sum"], [f64]), (["avg"], [f64]), (["min", "max"], [f64, f64]), (["var"], [f64]), (["sum"], [i64]), (["sum"], [u64]),
              (["min", "max"], [i64, u64]), (["sum", "avg"], [i64, f64]), (["sum", "count", "max"], [f64, f64, i64]), (["avg", "avg", "sum"], [f64, -f64, u64]),
              # 4-byte columns beside 8-byte ones
              (["sum", "avg"], [i32, f64]), (["sum", "sum", "max"], [f32, i64, u32]), (["min", "var"], [u32, f64]), (["avg", "max", "min", "sum"], [i32, f64, f32, f64]),
              # 1- and 2-byte columns (masks, small integers), alone and beside wider ones
              (["sum"], [b8]), (["sum", "min", "max"], [i8, i8, u8]), (["var", "avg"], [i16, u16]), (["sum", "sum", "max", "min"], [b8, f64, i16, i32]), (["count", "sum"], [u16, i8])]
    # one 8-byte key column takes the same kernel (its bits are the packed key): -1 is the table's empty mark, INT64_MIN and huge
    # unsigned values exercise the hash of the high half
    k64 = rng.integers(-3, 60, n).astype(np.int64) * 3_000_000_019
    k64[k64 == -3 * 3_000_000_019] = -1
    k64[k64 == -2 * 3_000_000_019] = np.iinfo(np.int64).min
    ku64 = rng.integers(0, 50, n).astype(np.uint64) * np.uint64(368_934_881_474_191_032) + np.uint64(7)
    for keys in ([rng.integers(-40, 40, n).astype(np.int32)], [rng.integers(0, 9, n).astype(np.int32), rng.integers(0, 7, n).astype(np.uint32) * np.uint32(600_000_000)],
                 [k64], [ku64]):
        ogb = oracle.groupby(keys)
        for names, vals in shapes + [(["sum", "count"], [rand(rng, np.int32, n, small=True)] * 2)]:
            ops = [ck.RED_NAMES[nm] for nm in names]
            gb = gpu.groupby_agg(keys, ops, vals, hint=128)
            assert gb.ngroups == ogb["ngroups"]
            assert np.array_equal(gb.first_rows(), ogb["first_rows"])
            for k, key in enumerate(keys):
                assert np.array_equal(gb.keys(k, key.dtype), key[ogb["first_rows"]])
            for j, (nm, v) in enumerate(zip(names, vals)):
                got = gb.result(j, ops[j], ck.tag_of(v))
                want = oracle.grouped_reduce(ops[j], v, ogb)
                if v.dtype.kind == "f" and nm in ("sum", "avg", "var"):
                    w, g = want.astype(np.float64), got.astype(np.float64)
                    scale = np.maximum(1.0, np.abs(w)) if nm != "var" else np.maximum(1.0, np.abs(w)) * 1e3
                    assert np.all(np.abs(g - w) <= scale * len(v) * 2.0 ** -50), (nm, names)
                else:
                    assert gu.same_bits(got, want), (nm, v.dtype, names)
            gb.destroy()


@pytest.mark.parametrize("n", [13, 200_003, 1_500_000])
def test_build_and_postproc_with_8_byte_and_mixed_keys(gpu, oracle, n):
    """aqg_groupby_build (reversemap, counts, first rows) and ht_postproc (descending row lists) for the key shapes whose first pass
    takes the fast kernel's 8-byte layouts: one int64 / uint64 column (with the -1 and INT64_MIN sentinels), two 4-byte columns;
    plus shapes that stay on the generic path: int64 + int32, int16 + int64 + uint8 (hasher.h:146-199)"""
    rng = np.random.default_rng(31 + n)
    k64 = rng.integers(-3, 40, n).astype(np.int64) * 5_000_000_029
    k64[k64 == -3 * 5_000_000_029] = -1
    k64[k64 == -2 * 5_000_000_029] = np.iinfo(np.int64).min
    shapes = [[k64], [rng.integers(0, 30, n).astype(np.uint64) * np.uint64(600_000_000_000_000_007)],
              [rng.integers(-5, 5, n).astype(np.int32), rng.integers(0, 6, n).astype(np.uint32) * np.uint32(700_000_001)],
              [k64, rng.integers(0, 3, n).astype(np.int32)],
              [rng.integers(0, 4, n).astype(np.int16), rng.integers(0, 5, n).astype(np.int64) << 40, rng.integers(0, 2, n).astype(np.uint8)]]
    for keys in shapes:
        o = oracle.groupby(keys)
        g = gpu.groupby_build(keys)
        assert g.ngroups == o["ngroups"]
        assert np.array_equal(g.reversemap(), o["reversemap"])
        assert np.array_equal(g.counts(), o["counts"])
        assert np.array_equal(g.first_rows(), o["first_rows"])
        for k, key in enumerate(keys):
            assert np.array_equal(g.keys(k, key.dtype), key[o["first_rows"]])
        off, rows = g.postproc()
        assert np.array_equal(off[:-1], o["offsets"]) and off[-1] == n
        assert np.array_equal(rows, o["row_ids"])
        g.destroy()


def test_handle_reuse_with_the_tail_still_queued(gpu, oracle):
    """small-table calls return once the group count is known, with first rows / rank / emit still queued on the stream: reusing the
    handle at once for another input, reading results right after a call, and the prepared call of bench.py must all see their
    own call's results (stream order)"""
    rng = np.random.default_rng(77)
    n = 600_011
    cases = []
    for t in range(3):
        keys = [rng.integers(-30 - t, 30 + t, n).astype(np.int32)]
        v = rng.integers(-1000, 1000, n).astype(np.int32)
        o = oracle.groupby(keys)
        cases.append((gpu.to_device(keys[0]), gpu.to_device(v), keys[0], o, ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, v, o))))
    gb = None
    for it in range(12):
        dk, dv, hk, o, want = cases[it % 3]
        gb = gpu.groupby_agg([dk], [ck.RED_SUM], [dv], hint=128, handle=gb)
        if it % 2:                                              # every other call: straight into the next one, nothing read in between
            continue
        assert gb.ngroups == o["ngroups"]
        assert np.array_equal(gb.keys(0, np.int32), hk[o["first_rows"]])
        assert np.array_equal(gb.first_rows(), o["first_rows"])
        assert ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32)) == want
    gb.destroy()
    dk, dv, hk, o, want = cases[1]
    run, pgb = gpu.prepare_groupby_agg([dk], [ck.RED_SUM], [dv], hint=128)
    for _ in range(5):
        run()
    assert pgb.ngroups == o["ngroups"] and ck.i128_to_int(pgb.result(0, ck.RED_SUM, ck.INT32)) == want
    pgb.destroy()


def test_handle_reuse_gaining_a_key_column(gpu, oracle):
    """a reused handle whose later calls have more key columns and more groups than the column buffers of the earlier ones were
    sized for: every group-sized buffer is tracked by itself (keys_out[1] was allocated for 10 groups and then
    written with 50,000)"""
    rng = np.random.default_rng(91)
    n = 300_000
    v = rng.integers(-100, 100, n).astype(np.int32)
    calls = [([rng.integers(0, 100_000, n).astype(np.int32)], 0),
             ([rng.integers(0, 5, n).astype(np.int32), rng.integers(0, 2, n).astype(np.int32)], 0),
             ([rng.integers(0, 250, n).astype(np.int32), rng.integers(0, 200, n).astype(np.int32)], 0),
             ([rng.integers(0, 40, n).astype(np.int32), rng.integers(0, 40, n).astype(np.int16), rng.integers(0, 30, n).astype(np.int8)], 0)]
    gb = None
    for keys, hint in calls:
        o = oracle.groupby(keys)
        gb = gpu.groupby_agg(keys, [ck.RED_SUM, ck.RED_COUNT], [v, v], hint=hint, handle=gb)
        assert gb.ngroups == o["ngroups"]
        assert np.array_equal(gb.first_rows(), o["first_rows"])
        assert np.array_equal(gb.counts(), o["counts"])
        for k, key in enumerate(keys):
            assert np.array_equal(gb.keys(k, key.dtype), key[o["first_rows"]])
        assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v, o))
    gb.destroy()


def test_wide_tuples_through_the_hash_partition_pipeline(gpu, oracle):
    """h2o Q10 shape at a size the wide-tuple partition plan takes (tuples wider than 8 bytes, more than 2^20 groups): six int32 key
    columns, sum(v3) and count; three key columns of mixed widths with an int32 sum; then a tuple that dominates the input, whose
    partition overflows LDS and sends the call back to the HBM table"""
    rng = np.random.default_rng(10)
    n = 2_400_007
    ids = [rng.integers(1, 101, n).astype(np.int32), rng.integers(1, 101, n).astype(np.int32), rng.integers(1, n // 100, n).astype(np.int32),
           rng.integers(1, 101, n).astype(np.int32), rng.integers(1, 101, n).astype(np.int32), rng.integers(1, n // 100, n).astype(np.int32)]
    v3 = np.round(rng.uniform(0, 100, n), 6).astype(np.float32)
    v1 = rng.integers(-5, 6, n).astype(np.int32)
    o = oracle.groupby(ids)
    assert o["ngroups"] > (1 << 20)
    for hint in (0, n):
        gb = gpu.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_SUM], [v3, v3, v1], hint=hint)
        assert gb.ngroups == o["ngroups"]
        assert np.array_equal(gb.first_rows(), o["first_rows"])
        assert np.array_equal(gb.counts(), o["counts"])
        for k in range(6):
            assert np.array_equal(gb.keys(k, np.int32), ids[k][o["first_rows"]])
        got, want = gb.result(0, ck.RED_SUM, ck.FLOAT), oracle.grouped_reduce(ck.RED_SUM, v3, o)
        assert np.all(np.abs(got - want) <= np.maximum(1.0, np.abs(want)) * 64 * 2.0 ** -50)
        assert gu.same_bits(gb.result(2, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v1, o))
        gb.destroy()
    # mixed widths: int64 + int16 + uint8 + int32 (15 bytes)
    mk = [rng.integers(-2**40, 2**40, n).astype(np.int64), rng.integers(-300, 300, n).astype(np.int16), rng.integers(0, 3, n).astype(np.uint8), ids[2]]
    o2 = oracle.groupby(mk)
    gb = gpu.groupby_agg(mk, [ck.RED_SUM, ck.RED_MIN], [v1, v3], hint=n)
    assert gb.ngroups == o2["ngroups"] and np.array_equal(gb.first_rows(), o2["first_rows"])
    assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v1, o2))
    assert gu.same_bits(gb.result(1, ck.RED_MIN, ck.FLOAT), oracle.grouped_reduce(ck.RED_MIN, v3, o2))
    gb.destroy()
    # one tuple on half of the rows: its partition cannot fit LDS; the call must still be right (HBM table)
    dom = [c.copy() for c in ids]
    half = rng.random(n) < 0.5
    for c in dom:
        c[half] = 7
    o3 = oracle.groupby(dom)
    gb = gpu.groupby_agg(dom, [ck.RED_COUNT], [v1], hint=n)
    assert gb.ngroups == o3["ngroups"] and np.array_equal(gb.first_rows(), o3["first_rows"]) and np.array_equal(gb.counts(), o3["counts"])
    gb.destroy()


def test_wide_partition_overflow_by_chance_retries_with_another_seed_then_falls_back():
    """partitions sized far too tightly (AQG_PW_SIGMA = -2: below the mean): a partition is over by a little, the plan is tried once
    more with another seed of the partition hash, is over again and the call falls back to the HBM table -- three attempts, one exact
    result.  A fresh process: the sizing knob is read once."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
import numpy as np
sys.path.insert(0, "tests")
import aquery2_amd, checker as ck
gpu, oracle = aquery2_amd.Device(0), ck.load_oracle()
rng = np.random.default_rng(11)
n = 2_400_007
ids = [rng.integers(1, 101, n).astype(np.int32), rng.integers(1, n // 100, n).astype(np.int32), rng.integers(1, 101, n).astype(np.int32), rng.integers(1, n // 100, n).astype(np.int32)]
v1 = rng.integers(-5, 6, n).astype(np.int32)
o = oracle.groupby(ids)
gb = gpu.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v1, v1], hint=n)
assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"]) and np.array_equal(gb.counts(), o["counts"])
assert np.array_equal(ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32)), ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, v1, o)))
print("OK", gb.ngroups)
'''
    env = dict(os.environ, AQG_PW_SIGMA="-2", AQG_DEBUG_FLAGS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    gave_up = [l for l in out.stderr.splitlines() if "wide partition plan gave up" in l]
    assert len(gave_up) == 2 and "seed 0)" in gave_up[0] and "seed 0)" not in gave_up[1], out.stderr[-2000:]


def test_dense_plan_reuses_sampled_ranges_and_survives_changed_data(gpu, oracle):
    """a handle keeps the sampled key ranges of its last dense plan: a second call over the same device columns skips the sampling
    pass; when the columns have been overwritten with values outside those ranges, the kernels' per-row check sends the call
    through the exact ranges once and the result is still the oracle's"""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 5_000_011                                             # >= 2^22 rows: ranges come from the first 2^20
    k1, k2 = rng.integers(1, 101, n).astype(np.int32), rng.integers(1, 101, n).astype(np.int32)
    v = rng.integers(-9, 10, n).astype(np.int32)
    d1, d2, dv = gpu.to_device(k1), gpu.to_device(k2), gpu.to_device(v)
    gb = None
    for _ in range(2):
        gb = gpu.groupby_agg([d1, d2], [ck.RED_SUM, ck.RED_COUNT], [dv, dv], hint=16384, handle=gb)
        o = oracle.groupby([k1, k2])
        assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
        assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v, o))
    k1b = k1.copy()
    k1b[2_000_000:] = rng.integers(50, 181, n - 2_000_000).astype(np.int32)      # beyond the cached [1, 100], behind the sampled rows
    gpu._chk(gpu.lib.aqg_h2d(gpu.ctx, C.c_void_p(d1.ptr), k1b.ctypes.data_as(C.c_void_p), C.c_size_t(k1b.nbytes)), "aqg_h2d")
    for _ in range(2):
        gb = gpu.groupby_agg([d1, d2], [ck.RED_SUM, ck.RED_COUNT], [dv, dv], hint=32768, handle=gb)
        o = oracle.groupby([k1b, k2])
        assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
        assert np.array_equal(gb.counts(), o["counts"])
        assert gu.same_bits(gb.result(0, ck.RED_SUM, ck.INT32), oracle.grouped_reduce(ck.RED_SUM, v, o))
    gb.destroy()
