"""Seeded random sweep over the pieces between the big kernels: mask compaction / mask -> index / gather with sizes around the
2048-row tiles and every element width, hash joins (pairs, unique-key lookup) over random key dtypes and cardinalities, against numpy
and the oracle.  AQG_FUZZ_SEEDS / AQG_FUZZ_BASE lengthen the sweep (3400 more seeds were run at the end of round 1: all green)."""
import os

import numpy as np
import pytest

import checker as ck
import golden_util as gu
from test_gpu_basic import rand

pytestmark = pytest.mark.gpu
DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.int64, np.uint64, np.float64]
KEY_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("AQG_FUZZ_SEEDS", "40"))))
def test_filter_gather_join_random_shapes(gpu, oracle, seed):
    rng = np.random.default_rng(int(os.environ.get("AQG_FUZZ_BASE", "4000")) + seed)
    n = int(rng.choice([0, 1, 7, 2047, 2048, 2049, 4097, 65_537, 300_001, 1_000_003]))
    dt = DTYPES[rng.integers(len(DTYPES))]
    x = rand(rng, dt, n)
    density = float(rng.choice([0.0, 0.01, 0.5, 0.99, 1.0]))
    mask = rng.random(n) < density
    # ColRef::operator[](mask) as a true compaction (D11), and the row ids of the survivors
    assert gu.same_bits(gpu.compact(x, mask), x[mask]), (n, dt, density)
    assert np.array_equal(gpu.mask_to_index(mask), np.nonzero(mask)[0].astype(np.uint32))
    # ColRef::operator[](index vector)
    if n:
        idx = rng.integers(0, n, int(rng.choice([1, 5, 4096, 100_003]))).astype(np.uint32)
        assert gu.same_bits(gpu.gather(x, idx), x[idx])
    # joins: every (probe row, build row) pair in the oracle's order; unique-key lookup
    kdt = KEY_DTYPES[rng.integers(len(KEY_DTYPES))]
    info = np.iinfo(kdt)
    card = int(rng.choice([1, 3, 50, 120]))
    card = min(card, int(info.max) - int(info.min))
    lo = int(info.min) if rng.random() < 0.3 else (int(info.max) - card if rng.random() < 0.3 else 0)
    nb, npr = int(rng.choice([1, 9, 200, 3000])), int(rng.choice([1, 10, 5000, 70_001]))
    build = np.array([lo + v for v in rng.integers(0, card, nb).tolist()], dtype=kdt)             # python ints: no wrap near the dtype's ends
    probe = np.array([min(lo + v, int(info.max)) for v in rng.integers(0, card + 3, npr).tolist()], dtype=kdt)
    pr, br = gpu.join_pairs(build, probe)
    opr, obr = oracle.join_pairs(build, probe)
    assert np.array_equal(pr, opr) and np.array_equal(br, obr), (kdt, nb, npr, card)
    dim = np.unique(build)
    rng.shuffle(dim)
    look = gpu.join_lookup(dim, probe)
    pos = {int(k): i for i, k in enumerate(dim.tolist())}
    want = np.array([pos.get(int(k), 0xFFFFFFFF) for k in probe.tolist()], dtype=np.uint32)
    assert np.array_equal(look, want), (kdt, len(dim), npr)
