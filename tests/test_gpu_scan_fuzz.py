"""Seeded random sweep over prefix scans, sliding windows and shifts: every numeric dtype, sizes from 1 to a few million rows
(so that tiles, halos, the chained scan's links and the multi-workgroup aggregate scan all take part), window lengths from 1 to
beyond the column -- against the oracle (bit-exact for integer inputs and for min / max / shifts of floats; stated bounds for
floating sums)."""
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


# AQG_FUZZ_SEEDS / AQG_FUZZ_BASE: a longer or different sweep (1500 more seeds were run at the end of round 1: all green)
@pytest.mark.parametrize("seed", range(int(os.environ.get("AQG_FUZZ_SEEDS", "48"))))
def test_scan_random_shapes(gpu, oracle, seed):
    rng = np.random.default_rng(int(os.environ.get("AQG_FUZZ_BASE", "7000")) + seed)
    dt = DTYPES[rng.integers(len(DTYPES))]
    n = int(rng.choice([1, 7, 2047, 2049, 70_001, 1_234_567, 3_000_001]))
    fp = np.dtype(dt).kind == "f"
    x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rand(rng, dt, n, small=True)
    for name in rng.choice(["sums", "avgs", "mins", "maxs", "deltas", "prev", "aggnext"], 3, replace=False):
        op = ck.SCAN_NAMES[str(name)]
        got, want = gpu.scan(op, x), oracle.scan(op, x)
        if fp and name in ("sums", "avgs"):
            # any two summation orders of the first i+1 values differ by at most (i+1) * eps * sum|x| (results are double)
            bound = 2.0 ** -52 * (np.arange(n) + 2) * np.cumsum(np.abs(x.astype(np.float64)))
            if name == "avgs":
                bound = bound / (np.arange(n) + 1)
            assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound + 1e-12), (seed, name, dt)
        else:
            assert gu.same_bits(got, want), (seed, name, dt)
    for name in rng.choice(["sumw", "avgw", "minw", "maxw", "ratiow"], 3, replace=False):
        w = int(rng.choice([1, 2, 3, 5, 10, 64, 100, 1000, 2500, n, n + 3]))
        if w == 0:
            continue
        op = ck.SCAN_NAMES[str(name)]
        if name == "avgw" and np.dtype(dt).kind == "u" and np.dtype(dt).itemsize >= 4:
            continue                                          # the reference wraps arr[i] - arr[i-w] for unsigned 4/8-byte inputs
        got, want = gpu.scan(op, x, w), oracle.scan(op, x, w)
        if name in ("minw", "maxw") or (name == "sumw" and not fp) or (name == "ratiow"):
            assert gu.same_bits(got, want), (seed, name, dt, w)
        else:                                                 # avgw: the reference's floating recurrence drifts; floating sumw: summation order
            # the device value is the window's sum / mean rounded once; the reference's recurrence adds one rounding of a term of
            # magnitude <= 2 max|x| per step, in the precision of its element type
            eps_in = float(np.finfo(dt).eps) if fp else 2.0 ** -52
            bound = 4 * eps_in * float(np.max(np.abs(x.astype(np.float64)))) * (np.arange(n) + 2) + 1e-9
            assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= bound), (seed, name, dt, w)
