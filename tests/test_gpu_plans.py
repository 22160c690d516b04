"""Every group-by plan of DESIGN.md section 4.1 pinned in the committed -m gpu suite: the plans only large inputs reach at default
thresholds (the two-level partition plan, the round-1 pipeline, the ordering tail over packed keys and over wide tuples) are FORCED
by their measurement switches in a fresh process (the switches are read once per process), the plan actually taken is read back
(aqg_groupby_plan) and every output is compared with the oracle.  Groups have several rows each, so merging inside the partition
aggregation and inside the ordering tail's partitions is exercised, not only one-row groups.  The last test is h2o Q10 at 1e8 rows at
DEFAULT thresholds (wide-tuple partitions + the ordering tail), checked against its own input."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import checker as ck

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import sys
import numpy as np
sys.path.insert(0, "tests")
import aquery2_amd, checker as ck, golden_util as gu
from aquery2_amd import capi
gpu, oracle = aquery2_amd.Device(0), ck.load_oracle()
def check(keys, ops, vals, hint, want_plan):
    o = oracle.groupby(keys)
    gb = gpu.groupby_agg(keys, ops, vals, hint=hint)
    assert gb.plan == want_plan, ("plan", gb.plan, want_plan)
    assert gb.ngroups == o["ngroups"], (gb.ngroups, o["ngroups"])
    assert np.array_equal(gb.first_rows(), o["first_rows"])
    for k, c in enumerate(keys):
        assert np.array_equal(gb.keys(k, c.dtype), c[o["first_rows"]]), k
    for j, (op, v) in enumerate(zip(ops, vals)):
        got, want = gb.result(j, op, ck.tag_of(v)), oracle.grouped_reduce(op, v, o)
        if v.dtype.kind == "f" and op in (ck.RED_SUM, ck.RED_AVG):
            assert np.allclose(got, want, rtol=1e-12, atol=1e-9), j
        else:
            assert gu.same_bits(got, want), j
    print("OK", gb.ngroups, flush=True)
rng = np.random.default_rng(41)
'''


def run_forced(env, body):
    out = subprocess.run([sys.executable, "-c", PRELUDE + body], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env), cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


PACKED = r'''
n = 3_000_017
key = rng.integers(0, 300_000, n).astype(np.int32)           # ~10 rows per group
v1, v3 = rng.integers(-9, 10, n).astype(np.int32), np.round(rng.uniform(0, 100, n), 3).astype(np.float32)
check([key], [ck.RED_SUM, ck.RED_COUNT, ck.RED_MIN, ck.RED_SUM], [v1, v1, v1, v3], 400_000, %s)
key8 = (key.astype(np.int64) << 33) | 5                        # an 8-byte packed key takes the same plans
check([key8], [ck.RED_SUM, ck.RED_MAX], [v1, v3], 400_000, %s)
'''


def test_two_level_partition_plan_forced():
    """AQG_P1_MAX=1: every partition plan takes two levels (what h2o Q3 / Q5 / Q7 take at 1e9 rows): p2_hist, p2_scatter x 2, p1_agg"""
    run_forced({"AQG_P1_MAX": "1"}, PACKED % ("capi.PLAN_PART_TWO", "capi.PLAN_PART_TWO"))


PACKVALS = r"""
n = 4_700_023                                                  # (value packing is planned from 2^22 rows on)
key = rng.integers(0, 400_000, n).astype(np.int32)
v1, v2 = rng.integers(1, 6, n).astype(np.int32), rng.integers(-3, 12, n).astype(np.int32)
v3 = np.round(rng.uniform(0, 100, n), 3).astype(np.float32)
PT = capi.PLAN_PART_TWO
check([key], [ck.RED_SUM, ck.RED_SUM, ck.RED_SUM], [v1, v2, v3], 500_000, PT | capi.PLAN_PACKED_VALUES)            # h2o Q5: {key|v1|v2, row, v3}
check([key], [ck.RED_MAX, ck.RED_MIN, ck.RED_COUNT], [v1, v2, v1], 500_000, PT | capi.PLAN_PACKED_VALUES)          # h2o Q7: no value plane at all
check([key], [ck.RED_AVG, ck.RED_VAR, ck.RED_MIN], [v2, v1, v2], 500_000, PT | capi.PLAN_PACKED_VALUES)            # sums of squares of a packed field
wide = v1.copy(); wide[::7] = 1 << 20                          # a range that leaves no room next to the key: travels as its own plane
check([key], [ck.RED_SUM, ck.RED_SUM], [wide, v2], 500_000, PT | capi.PLAN_PACKED_VALUES)
late = v1.copy(); late[n - 5] = 77                             # a value outside the range of the sampled first 2^20 rows: caught while packing, the call repeats unpacked
check([key], [ck.RED_SUM, ck.RED_MAX], [late, late], 500_000, PT)
bigkey = key.copy(); bigkey[n - 9] = (1 << 30) + 5             # a KEY above the sampled maximum would run into the value fields: caught the same way
check([bigkey], [ck.RED_SUM, ck.RED_SUM], [v1, v2], 500_000, PT)
"""


def test_two_level_plan_with_narrow_value_columns_inside_the_key_word():
    """AQG_P1_MAX=1 at >= 2^22 rows: the two-level plan packs 4-byte integer value columns of a small sampled range into the spare bits of a
    4-byte key word (h2o Q5 / Q7 at 1e9 rows), verifies every row while packing and repeats the call unpacked when a row does not fit"""
    run_forced({"AQG_P1_MAX": "1", "AQG_DISABLE_RANGED": "1"}, PACKVALS)


RANGED = r"""
n = 4_700_023
PT, PK, RG = capi.PLAN_PART_TWO, capi.PLAN_PACKED_VALUES, capi.PLAN_RANGE_PARTITIONS
key = rng.integers(0, 400_000, n).astype(np.int32)
v1, v2 = rng.integers(1, 6, n).astype(np.int32), rng.integers(-3, 12, n).astype(np.int32)
v3, v4 = np.round(rng.uniform(0, 100, n), 3).astype(np.float32), rng.integers(-2**40, 2**40, n).astype(np.int64)
check([key], [ck.RED_SUM, ck.RED_SUM, ck.RED_SUM], [v1, v2, v3], 500_000, PT | PK | RG)                              # h2o Q5
check([key], [ck.RED_MAX, ck.RED_MIN, ck.RED_COUNT], [v1, v2, v1], 500_000, PT | PK | RG)                             # h2o Q7
check([key], [ck.RED_AVG, ck.RED_VAR, ck.RED_SUM, ck.RED_MAX], [v3, v2, v4, v4], 500_000, PT | PK | RG)               # 8-byte planes, squares, counts
check([key - 200_000], [ck.RED_SUM, ck.RED_MIN], [v3, v3], 500_000, PT | RG)                                          # a domain around zero: nothing packs beside a negative key
check([(key.astype(np.int64) + 2**32 - 500_000).astype(np.uint32)], [ck.RED_SUM, ck.RED_COUNT], [v1, v1], 500_000, PT | RG)   # uint32 keys up to 2^32 - 1
sparse = key * 37                                              # 1.5e7 key values for 4e5 groups: still range partitions (more of them)
check([sparse], [ck.RED_SUM, ck.RED_SUM], [v1, v3], 500_000, PT | PK | RG)
wide = key.astype(np.int64) * 5003                             # a domain of 2e9: hashed
check([wide.astype(np.int32)], [ck.RED_SUM, ck.RED_SUM], [v1, v3], 500_000, PT)
skew = key.copy(); skew[2_000_000 : 2_000_000 + n // 3] = 77      # one key holds a third of the rows (behind the sampled first 2^20)
check([skew], [ck.RED_SUM, ck.RED_MAX], [v1, v3], 500_000, PT | PK | RG)
late = key.copy(); late[n - 9] = 900_000                       # a key far outside the sampled range: flagged by the scatter, the call repeats hashed
check([late], [ck.RED_SUM, ck.RED_SUM], [v1, v3], 500_000, PT)
near = key.copy(); near[n - 9] = 400_700                       # just outside the sample's maximum: inside the slack the domain is given, still ranged
check([near], [ck.RED_SUM, ck.RED_SUM], [v3, v3], 500_000, PT | RG)
"""


def test_one_level_plan_over_a_dense_key_domain_takes_range_partitions():
    """the same shapes at the default thresholds: 4e5 groups take ONE partition level, and a dense key domain takes its range form too
    (p1_hist / p1_scatter with order-preserving bins, p1_agg_direct_kernel over the [bin][chunk] starts), with the narrow value columns in the key word"""
    one = RANGED.replace("PT, PK, RG = capi.PLAN_PART_TWO, capi.PLAN_PACKED_VALUES, capi.PLAN_RANGE_PARTITIONS", "PT, PK, RG = capi.PLAN_PART_ONE, capi.PLAN_PACKED_VALUES, capi.PLAN_RANGE_PARTITIONS")
    # (a domain of 1.5e7 values for 4e5 groups would need more bins than the hashed plan was given: hashed here, the value column still packed)
    one = one.replace("check([sparse], [ck.RED_SUM, ck.RED_SUM], [v1, v3], 500_000, PT | PK | RG)", "check([sparse], [ck.RED_SUM, ck.RED_SUM], [v1, v3], 500_000, PT | PK)")
    assert one.count("500_000, PT | PK)") == 1
    run_forced({}, one)                                     # <= 256 range bins: the cursor scatter of the two-level plan as the only level
    run_forced({"AQG_DISABLE_P1_CURSORS": "1"}, one)        # the chunk-histogram scatter (what more than 256 range bins take)


def test_two_level_plan_over_a_dense_key_domain_takes_range_partitions():
    """one 4-byte integer key column whose sampled values fill their range (h2o id3 / id6): order-preserving bins umulhi(key - kmin, M) at both
    levels and p1_agg_direct_kernel (accumulators indexed by the key, no key table); a key outside the sampled domain repeats the call hashed"""
    run_forced({"AQG_P1_MAX": "1"}, RANGED)


ROWS = r"""
n = 2_600_003
def check_rows(keys, ops, vals, hint):
    o = oracle.groupby(keys)
    assert o["ngroups"] == n
    gb = gpu.groupby_agg(keys, ops, vals, hint=hint)
    assert gb.plan & capi.PLAN_ROW_EMIT, gb.plan
    assert gb.ngroups == n and np.array_equal(gb.first_rows(), np.arange(n, dtype=np.uint32))
    for k, c in enumerate(keys):
        assert np.array_equal(gb.keys(k, c.dtype), c), k
    for j, (op, v) in enumerate(zip(ops, vals)):
        got, want = gb.result(j, op, ck.tag_of(v)), oracle.grouped_reduce(op, v, o)
        assert gu.same_bits(got, want), (j, op)
    print("OK", gb.ngroups, flush=True)
key = rng.permutation(n).astype(np.int32) - 1_000_000
i32, f32 = rng.integers(-2**31, 2**31, n).astype(np.int32), np.round(rng.normal(0, 1e3, n), 2).astype(np.float32)
f32[::5] = -0.0
i64, f64, u16 = rng.integers(-2**62, 2**62, n).astype(np.int64), rng.normal(0, 1e6, n), rng.integers(0, 2**16, n).astype(np.uint16)
check_rows([key], [ck.RED_SUM, ck.RED_AVG, ck.RED_MIN, ck.RED_MAX, ck.RED_COUNT], [i32, i32, f32, f32, i32], n + 1000)
check_rows([key], [ck.RED_SUM, ck.RED_VAR, ck.RED_STDDEV, ck.RED_SUM], [f32, i32, f32, i64], n + 1000)
check_rows([key.astype(np.int64) * 3], [ck.RED_SUM, ck.RED_MAX, ck.RED_AVG, ck.RED_MIN], [f64, i64, u16, u16], n + 1000)
ids = [(key % 1000).astype(np.int32), (key // 1000).astype(np.int32), np.zeros(n, np.int32), (key % 7).astype(np.int32)]       # a 16-byte tuple, all distinct
check_rows(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_AVG], [f32, f32, i32], n + 1000)
"""


def test_every_row_its_own_group_is_emitted_as_a_map_of_the_input():
    """G == n out of a partition plan (a grouping by a unique key: h2o Q10 at 1e9 rows): emit_rows_kernel writes the result columns from the
    input rows -- every aggregate and value type against the oracle's per-group reductions (groups of one row: D9's n + 1 included), the
    -0.0 a floating sum turns into +0.0, packed keys and a wide tuple"""
    run_forced({}, ROWS)


def test_round1_partition_pipeline_forced():
    """AQG_DISABLE_P1=1: the round-1 pipeline of partition.hip (packed keys beyond 4096 partitions at default thresholds): part_* kernels"""
    run_forced({"AQG_DISABLE_P1": "1"}, PACKED % ("capi.PLAN_PART_ROUND1", "capi.PLAN_PART_ROUND1"))


def test_one_level_partition_plan_is_what_mid_cardinality_takes():
    run_forced({}, PACKED % ("capi.PLAN_PART_ONE", "capi.PLAN_PART_ONE"))


def test_ordering_tail_over_packed_keys_forced():
    """AQG_SORTED_TAIL_MIN=1: the records of a partition plan are ORDERED by first row (pn_level_hist, p2_scatter, sorted_emit_kernel) instead
    of ranked through a bitmap -- what any partition plan takes from 2^24 groups on; here with ~10 rows per group"""
    run_forced({"AQG_SORTED_TAIL_MIN": "1"}, PACKED % ("capi.PLAN_PART_ONE | capi.PLAN_SORTED_TAIL", "capi.PLAN_PART_ONE | capi.PLAN_SORTED_TAIL"))
    run_forced({"AQG_SORTED_TAIL_MIN": "1", "AQG_P1_MAX": "1"}, PACKED % ("capi.PLAN_PART_TWO | capi.PLAN_SORTED_TAIL", "capi.PLAN_PART_TWO | capi.PLAN_SORTED_TAIL"))


WIDE = r'''
n = 3_200_011
r = rng.integers(0, 1_400_000, n)                              # ~2.3 rows per tuple, 1.2e6 distinct tuples of 16 bytes
ids = [(r %% 100 + 1).astype(np.int32), (r // 100 %% 1000).astype(np.int32), (r // 100_000 + 7).astype(np.int32), ((r * 7) %% 13).astype(np.int32)]
v1, v3 = rng.integers(-9, 10, n).astype(np.int32), np.round(rng.uniform(0, 100, n), 3).astype(np.float32)
check(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_SUM], [v1, v1, v3], 1_300_000, %s)      # (2.5 rows per tuple: the partitions are sized for the wider spread)
'''


def test_wide_tuple_partition_plan_with_and_without_the_ordering_tail():
    """tuples wider than 8 bytes with more than 2^20 groups expected (h2o Q10's plan): pw_hash, the hashed tile scatters, pw_agg -- at
    default thresholds with the bitmap tail, and with the ordering tail forced (wide tuples fetch their key columns through first rows)"""
    run_forced({}, WIDE % "capi.PLAN_PART_WIDE")
    run_forced({"AQG_SORTED_TAIL_MIN": "1"}, WIDE % "capi.PLAN_PART_WIDE | capi.PLAN_SORTED_TAIL")


NEARLY = r"""
n = 2_700_017
r = rng.permutation(n)                                          # every row its own 16-byte tuple ...
r[rng.integers(0, n, 4000)] = r[rng.integers(0, n, 4000)]      # ... except ~4000 rows that repeat another row's
ids = [(r % 1000).astype(np.int32), (r // 1000).astype(np.int32), np.zeros(n, np.int32), (r % 7).astype(np.int32)]
v1, v3 = rng.integers(-9, 10, n).astype(np.int32), np.round(rng.uniform(0, 100, n), 3).astype(np.float32)
W = capi.PLAN_PART_WIDE
check(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_MAX], [v3, v1, v1], n, W)                 # hint ~ rows: ids and values are read only where a partition has a duplicate
check(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_MIN], [v1, v1, v3], 1_700_000, W)         # a hint well below the rows: everything loaded up front
"""


def test_wide_tuples_nearly_all_distinct_put_off_their_distinct_partitions():
    """pw_agg when the caller could emit from the input rows (every row its own group): a partition of distinct rows is marked and counted, nothing
    accumulated or written; here ~4000 of 2.7e6 rows repeat another row, so most partitions are put off, G != n, and the second launch (mode 2)
    aggregates exactly the marked ones -- against the oracle, with and without the late loads of row ids and values; and the same with
    AQG_DISABLE_PW_DEFER=1 (one launch over everything)"""
    run_forced({}, NEARLY)
    run_forced({"AQG_DISABLE_PW_DEFER": "1"}, NEARLY)


WIDEPACK = r"""
n = 4_400_021                                                  # (key packing is planned from 2^22 rows on)
r = rng.integers(0, 1_900_000, n)                              # ~2.3 rows per tuple
ids = [(r % 100 + 1).astype(np.int32), (r // 100 % 1000 - 500).astype(np.int32), (r // 100_000 + 7).astype(np.int32), ((r * 7) % 13).astype(np.int32), (r % 3).astype(np.uint32)]
v1, v3 = rng.integers(-9, 10, n).astype(np.int32), np.round(rng.uniform(0, 100, n), 3).astype(np.float32)
W, PK = capi.PLAN_PART_WIDE, capi.PLAN_PACKED_KEYS
check(ids, [ck.RED_SUM, ck.RED_COUNT, ck.RED_SUM], [v1, v1, v3], 1_800_000, W | PK)          # 7 + 10 + 5 + 4 + 2 bits: one dword plane instead of five
big = [c.copy() for c in ids]; big[2] = (big[2].astype(np.int64) * 40_000_000).astype(np.int32)   # a column that needs all its 32 bits takes a plane by itself, the other four share one
check(big, [ck.RED_SUM, ck.RED_COUNT], [v1, v1], 1_800_000, W | PK)
late = [c.copy() for c in ids]; late[1][n - 3] = 70_000       # a key outside the range of the sampled first 2^20 rows: caught while packing, the call repeats unpacked
check(late, [ck.RED_SUM, ck.RED_MIN], [v3, v1], 1_800_000, W)
"""


def test_wide_tuples_travel_packed_under_sampled_ranges():
    """tuples wider than 8 bytes whose 4-byte integer columns have narrow sampled ranges (h2o Q10's six id columns need 76 bits, not 192): the
    hash pass also writes the columns packed into fewer dword planes, every row verified; a miss repeats the call unpacked"""
    run_forced({}, WIDEPACK)


def test_h2o_q10_at_1e8_rows_default_thresholds_groups_of_two():
    """h2o Q10 `sum(v3), count(*) BY id1 .. id6` (benchmark/h2o/groupby.sql:23) at 1e8 rows with the DEFAULT thresholds: hint 1e8 >= 2^24, so
    the call takes the wide-tuple partitions AND the ordering tail.  The table is the first 5e7 rows of the seed-42 columns (all tuples
    distinct) twice over, so every group has exactly two rows and the result is known from the input alone: groups = 5e7, group g = row g,
    count 2, sum(v3) = 2 * v3[g] (exact in double).  tools/q10_check.py is the 1e9-row form (every group one row)."""
    import aquery2_amd
    from aquery2_amd import capi
    d = aquery2_amd.Device(0)
    try:
        half, n = 50_000_000, 100_000_000
        def doubled(col, dtype):
            a = d.gen_column(col, 42, 0, half, 1_000_000_000, 100)
            b = d.empty(n, dtype)
            for k in range(2):
                d._chk(d.lib.aqg_d2d(d.ctx, C.c_void_p(b.ptr + k * half * 4), C.c_void_p(a.ptr), C.c_size_t(half * 4)), "aqg_d2d")
            d.sync(); a.free()
            return b
        ids = [doubled(c, np.int32) for c in (ck.GEN_ID1, ck.GEN_ID2, ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6)]
        v3 = doubled(ck.GEN_V3, np.float32)
        gb = d.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=n)
        assert gb.plan == capi.PLAN_PART_WIDE | capi.PLAN_SORTED_TAIL | capi.PLAN_PACKED_KEYS, gb.plan      # (six id columns in three dword planes)
        assert gb.ngroups == half, gb.ngroups                         # (seed 42: the first 5e7 tuples are all distinct)
        tmp = d.empty(half, np.int32)
        for k in range(6):
            d._chk(d.lib.aqg_groupby_keys(gb.h, k, C.c_void_p(tmp.ptr)), "aqg_groupby_keys")
            first_half = aquery2_amd.DevBuf(d, ids[k].ptr, np.int32, half, owned=False)
            assert int(d.reduce(ck.RED_SUM, d.ewise(ck.OP_NE, tmp, first_half, keep=True))) == 0, k
        assert np.array_equal(gb.first_rows(), np.arange(half, dtype=np.uint32))
        cnt = gb.result(1, ck.RED_COUNT, ck.FLOAT)
        assert int(cnt.min()) == 2 and int(cnt.max()) == 2
        s = gb.result(0, ck.RED_SUM, ck.FLOAT)
        h3 = aquery2_amd.DevBuf(d, v3.ptr, np.float32, half, owned=False).to_host()
        assert np.array_equal(s, 2.0 * h3.astype(np.float64))
    finally:
        d.close()


def test_packed_keys_beyond_the_partition_plans_take_the_wide_tuple_plan():
    """more than 2^25 groups expected over a key of <= 8 bytes (a nearly unique 4-byte key, a pair of them): the packed-key partition plans end
    there, and the call is handled like a wide tuple -- hash partitions, representative rows, the ordering tail -- instead of falling to the
    HBM table (2e8 unique keys: 105 -> 11 ms)"""
    import aquery2_amd
    import golden_util as gu
    from aquery2_amd import capi
    d = aquery2_amd.Device(0)
    oracle = ck.load_oracle()
    try:
        n = 40_000_003
        i = np.arange(n, dtype=np.int64)
        key = ((i * 2654435761) % (1 << 31)).astype(np.int32)       # all distinct
        key[1::10] = key[0::10][: key[1::10].size]                   # ... but every tenth row repeats its neighbour: 3.6e7 groups
        v = (i % 1000 - 500).astype(np.int32)
        for keys in ([key], [(key & 0xFFFF).astype(np.uint16), (key >> 16).astype(np.int16)]):
            o = oracle.groupby(keys)
            gb = d.groupby_agg(keys, [ck.RED_SUM, ck.RED_COUNT, ck.RED_MIN], [v, v, v], hint=n)
            assert gb.plan & capi.PLAN_PART_WIDE, gb.plan
            assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
            for k, c in enumerate(keys):
                assert np.array_equal(gb.keys(k, c.dtype), c[o["first_rows"]]), k
            for j, op in enumerate((ck.RED_SUM, ck.RED_COUNT, ck.RED_MIN)):
                assert gu.same_bits(gb.result(j, op, ck.tag_of(v)), oracle.grouped_reduce(op, v, o)), j
            gb.destroy()
    finally:
        d.close()


@pytest.mark.parametrize("n,G", [(9_000_017, 4_000_000), (4_200_000, -1)])
def test_grouped_reduce_partitioned_on_the_dense_group_id(n, G):
    """aqg_grouped_reduce beyond the LDS tables: rows {id, value} partitioned on the build's dense group id (one and two levels of
    order-preserving bins, cursors from the offsets: no histogram pass), direct-indexed LDS accumulators, results written in id order.
    Every op and value dtype it serves against the oracle's per-group reduction; 8-byte integer sums still take the hashed plans."""
    import aquery2_amd
    import golden_util as gu
    from aquery2_amd import capi
    d = aquery2_amd.Device(0)
    oracle = ck.load_oracle()
    try:
        rng = np.random.default_rng(n + G)
        key = rng.integers(0, G, n).astype(np.int32) if G > 0 else rng.permutation(n).astype(np.int32)     # (G < 0: nearly every row its own group)
        key[: n // 50] = key[0]                                   # one large group among many small ones
        o = oracle.groupby([key])
        gb = d.groupby_build([key])
        assert gb.ngroups == o["ngroups"] and gb.ngroups > (3 << 20)
        for dt in (np.int8, np.int16, np.int32, np.uint16, np.uint32, np.float32, np.float64, np.int64):
            fp = np.dtype(dt).kind == "f"
            x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rng.integers(-100 if np.dtype(dt).kind == "i" else 0, 100, n).astype(dt)
            for name in ("sum", "avg", "min", "max", "var", "stddev"):
                op = ck.RED_NAMES[name]
                got, want = d.grouped_reduce(gb, op, x), oracle.grouped_reduce(op, x, o)
                served = not (dt == np.int64 and name not in ("min", "max"))
                assert (gb.plan == capi.PLAN_GID_PARTITION) == served, (dt, name, gb.plan)
                if fp and name in ("sum", "avg", "var", "stddev"):
                    assert np.allclose(got, want, rtol=1e-9, atol=1e-6), (dt, name)
                else:
                    assert gu.same_bits(got, want), (dt, name)
        # (the narrow int32 / uint32 columns above travelled inside the id word; here one late row does not fit the field its sample gave it)
        x = rng.integers(1, 6, n).astype(np.int32)
        x[n - 3] = 1 << 20
        for name in ("sum", "max"):
            assert gu.same_bits(d.grouped_reduce(gb, ck.RED_NAMES[name], x), oracle.grouped_reduce(ck.RED_NAMES[name], x, o)), name
    finally:
        d.close()


BUILD = r'''
n = 3_000_017
def check_build(keys, want_plan):
    o = oracle.groupby(keys)
    gb = gpu.groupby_build(keys)
    assert gb.plan & want_plan == want_plan and (want_plan or not gb.plan & capi.PLAN_BUILD_PARTITIONED), ("plan", gb.plan, want_plan)
    assert gb.ngroups == o["ngroups"]
    assert np.array_equal(gb.reversemap(), o["reversemap"]) and np.array_equal(gb.counts(), o["counts"]) and np.array_equal(gb.first_rows(), o["first_rows"])
    for k, c in enumerate(keys):
        assert np.array_equal(gb.keys(k, c.dtype), c[o["first_rows"]]), k
    off, rows = gb.postproc()
    assert np.array_equal(off[:-1], o["offsets"]) and np.array_equal(rows, o["row_ids"])
    v = rng.integers(-9, 10, n).astype(np.int32)
    assert gu.same_bits(gpu.grouped_reduce(gb, ck.RED_SUM, v), oracle.grouped_reduce(ck.RED_SUM, v, o))
    print("OK", gb.ngroups, flush=True)
key = rng.integers(0, %d, n).astype(np.int32)
key[::1000] = np.int32(-2**31)                                  # the value that doubles as the LDS tables' empty mark
check_build([key], %s)
check_build([(key.astype(np.int64) << 33) | 5], %s)            # an 8-byte key word
check_build([(key %% 1000).astype(np.int16), (key // 1000).astype(np.int32)], %s)   # two columns packed into one word
'''


LOOKUP = r"""
n = 4_500_037
def check_build(keys, lookup):
    o = oracle.groupby(keys)
    gb = gpu.groupby_build(keys)
    assert bool(gb.plan & capi.PLAN_BUILD_LOOKUP) == lookup and gb.plan & capi.PLAN_BUILD_PARTITIONED, ("plan", gb.plan, lookup)
    assert gb.ngroups == o["ngroups"]
    assert np.array_equal(gb.reversemap(), o["reversemap"]) and np.array_equal(gb.counts(), o["counts"]) and np.array_equal(gb.first_rows(), o["first_rows"])
    assert np.array_equal(gb.keys(0, keys[0].dtype), keys[0][o["first_rows"]])
    off, rows = gb.postproc()
    assert np.array_equal(off[:-1], o["offsets"]) and np.array_equal(rows, o["row_ids"])
    v = rng.integers(-9, 10, n).astype(np.int32)
    assert gu.same_bits(gpu.grouped_reduce(gb, ck.RED_SUM, v), oracle.grouped_reduce(ck.RED_SUM, v, o))
    print("OK", gb.ngroups, flush=True)
key = rng.integers(0, 300_000, n).astype(np.int32)
check_build([key], True)                                       # a domain of 3e5 values: ids through the key -> id table
check_build([key - 150_000], True)                             # around zero
check_build([(key.astype(np.int64) + 2**32 - 400_000).astype(np.uint32)], True)
check_build([key * 9], False)                                  # 2.7e6 values: beyond the table's 2^21, the routed form
late = key.copy(); late[n - 5] = 1_900_000                     # a key outside the sampled domain: caught by the look-up pass, the call repeats routed
check_build([late], False)
hole = key.copy(); hole[: 1 << 20] = np.arange(1 << 20, dtype=np.int32) % 200_000      # (the first 2^20 rows hold 2e5 values, the column 3e5: the sample is spread over the column and sees them)
check_build([hole], True)
srt = np.sort(key)                                             # a column sorted by its key: its first rows show one end of the range only
check_build([srt], True)
"""


def test_build_over_a_small_dense_key_domain_goes_through_a_lookup_table():
    """aqg_groupby_build above the LDS tables over one 4-byte integer key whose sampled domain has at most 2^21 values: the group table from the
    partition plan, then reversemap[row] = table[key - kmin] in row order (no partitioned rows kept, no routing back); keys outside the sampled
    domain repeat the call through the routed form"""
    run_forced({}, LOOKUP)


@pytest.mark.parametrize("G", [300_000, 2_500_000])
def test_build_through_the_partition_plans(G):
    """aqg_groupby_build above the LDS tables: the group table from the one- / two-level partition plan (counts only), then the id of every row
    from one more pass over the rows still lying partitioned -- reversemap, counts, first rows, keys, ht_postproc and a grouped reduction
    against the oracle; 4- and 8-byte key words, a packed pair, the key value that doubles as the empty mark.  With the switch off
    (AQG_DISABLE_BUILD_PARTITION=1) the same calls take the HBM table and give the same results."""
    plan = "capi.PLAN_BUILD_PARTITIONED"
    run_forced({}, BUILD % (G, plan, plan, plan))
    run_forced({"AQG_DISABLE_BUILD_PARTITION": "1"}, BUILD % (G, "0", "0", "0"))
