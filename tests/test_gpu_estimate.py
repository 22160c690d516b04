"""GPU parity for calls WITHOUT a group-count hint over inputs large enough (>= 2^22 rows) that the library first estimates the count
from a spread sample (estimate_groups / sample_distinct_kernel in csrc/groupby.hip) -- the only way the header layer ever calls
(include/aquery/hasher.h: max_groups_hint = 0).  Whatever the estimate says the result must be the oracle's: few groups, the all-ones
key (the sample table's empty mark), wide tuples (counted by their hash), every row its own group, and keys clustered in runs (the
case the clustering check exists for: a table sorted by its key)."""
import numpy as np
import pytest

import checker as ck
from test_gpu_basic import check_agg, rand

pytestmark = pytest.mark.gpu
N = (1 << 22) + 200_003


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def vals_of(rng):
    return [rand(rng, np.int32, N, small=True), np.round(rng.uniform(0, 100, N), 6).astype(np.float32)]


def test_few_groups_and_the_all_ones_key(gpu, oracle):
    rng = np.random.default_rng(401)
    k32 = rng.integers(-50, 50, N).astype(np.int32)                       # -1 among them: 0xFFFFFFFF, a 4-byte tuple (not the 8-byte empty mark)
    check_agg(gpu, oracle, [k32], vals_of(rng), 0)
    k64 = rng.integers(-3, 300_000, N).astype(np.int64)                    # -1 as an 8-byte tuple IS the sample table's empty mark
    check_agg(gpu, oracle, [k64], vals_of(rng), 0)
    only = np.full(N, -1, np.int64)                                        # nothing but the empty mark: one group
    check_agg(gpu, oracle, [only], vals_of(rng), 0)


def test_wide_tuples_are_counted_by_their_hash(gpu, oracle):
    rng = np.random.default_rng(402)
    a, b = rng.integers(0, 500, N).astype(np.int64) * 1_000_003, rng.integers(0, 400, N).astype(np.int32)
    check_agg(gpu, oracle, [a, b, (b % 3).astype(np.int32)], vals_of(rng), 0)           # 16-byte tuples, ~2e5 groups


def test_every_row_its_own_group(gpu, oracle):
    rng = np.random.default_rng(403)
    check_agg(gpu, oracle, [rng.permutation(N).astype(np.int32)], vals_of(rng), 0)


@pytest.mark.parametrize("run", [3, 9, 700])
def test_keys_clustered_in_runs(gpu, oracle, run):
    """a column sorted by its key: the sample's blocks share no tuples, the estimate goes by d * n / s"""
    rng = np.random.default_rng(404 + run)
    key = (np.arange(N) // run).astype(np.int32) * 7 - 1_000_000
    check_agg(gpu, oracle, [key], vals_of(rng), 0)
    two = [(np.arange(N) // (run * 50)).astype(np.int32), ((np.arange(N) // run) % 50).astype(np.int16)]      # the same runs as a two-column tuple
    check_agg(gpu, oracle, two, vals_of(rng), 0)


def test_float_keys_with_both_zeros(gpu):
    """(the oracle's group-by takes integer tuples; the reference semantics of floating keys -- 0.0 and -0.0 one group -- are pinned by
    tests/test_gpu_keys.py against the reference's own ids: here numpy restates them for a column large enough to be estimated)"""
    rng = np.random.default_rng(405)
    f = rng.integers(-500, 500, N).astype(np.float32) / 4
    f[::7] = -0.0
    v = rand(rng, np.int32, N, small=True)
    _, first, inv = np.unique(f + np.float32(0.0), return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                               # groups in first-occurrence order
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    gid = rank[inv]
    gb = gpu.groupby_agg([f], [ck.RED_SUM, ck.RED_COUNT], [v, v], hint=0)
    assert gb.ngroups == len(first)
    assert np.array_equal(gb.first_rows(), first[order].astype(np.uint32))
    assert ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32)) == np.bincount(gid, weights=v.astype(np.float64)).astype(np.int64).tolist()
    assert np.array_equal(gb.result(1, ck.RED_COUNT, ck.INT32), np.bincount(gid).astype(np.uint64))
    gb.destroy()
