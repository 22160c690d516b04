"""GPU parity at the edges: empty and tiny inputs, sizes around every tile boundary (64 / 256 / 2048 / 4096 / 8192) and columns
that start at addresses aligned only to their element size (the vector paths need 16 bytes and must fall back cleanly)."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu
from test_gpu_basic import rand

pytestmark = pytest.mark.gpu
SIZES = [0, 1, 2, 3, 5, 63, 64, 65, 255, 257, 2047, 2049, 4097, 8193, 12289]


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def view(gpu, a, off):
    """device copy of `a` whose first element sits `off` elements past a 256-byte aligned allocation"""
    import aquery2_amd
    pad = np.concatenate([np.zeros(off, a.dtype), a]) if off else a
    base = gpu.to_device(pad if len(pad) else np.zeros(1, a.dtype))
    v = aquery2_amd.DevBuf(gpu, base.ptr + off * a.itemsize, a.dtype, len(a), owned=False)
    v._base = base
    return v


@pytest.mark.parametrize("off", [0, 1, 3])
def test_ewise_and_reduce_sizes(gpu, oracle, off):
    rng = np.random.default_rng(100 + off)
    for n in SIZES:
        for lt, rt, op in ((np.int32, np.int32, ck.OP_ADD), (np.int8, np.int16, ck.OP_MUL), (np.float32, np.float64, ck.OP_DIV),
                           (np.int64, np.int32, ck.OP_GT), (np.uint8, np.uint8, ck.OP_SUB)):
            x, y = rand(rng, lt, n, small=True), rand(rng, rt, n, small=True)
            if op == ck.OP_DIV:
                y = np.where(y == 0, 1, y).astype(rt)
            assert gu.same_bits(gpu.ewise(op, view(gpu, x, off), view(gpu, y, off)), oracle.ewise(op, x, y)), (n, lt, rt, op)
            assert gu.same_bits(gpu.ewise(op, view(gpu, x, off), y[:1][0] if n else rt(3)), oracle.ewise(op, x, y[:1][0] if n else rt(3))), (n, "scalar")
        for dt in (np.int32, np.int8, np.uint64, np.float32):
            x = rand(rng, dt, n, small=True)
            for op in (ck.RED_SUM, ck.RED_MIN, ck.RED_MAX, ck.RED_COUNT) + ((ck.RED_AVG, ck.RED_FIRST, ck.RED_LAST) if n else ()):
                a, b = gpu.reduce(op, view(gpu, x, off)), oracle.reduce(op, x)
                if np.dtype(dt).kind == "f" and op in (ck.RED_SUM, ck.RED_AVG):
                    assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(b))), (n, dt, op)
                else:
                    assert np.asarray(a).tobytes() == np.asarray(b).tobytes() or a == b, (n, dt, op, a, b)


@pytest.mark.parametrize("off", [0, 1])
def test_scan_sizes_and_windows(gpu, oracle, off):
    rng = np.random.default_rng(200 + off)
    for n in SIZES:
        x = rand(rng, np.int32, n, small=True)
        xv = view(gpu, x, off)
        for name in ("sums", "mins", "maxs", "avgs", "deltas", "prev", "aggnext"):
            assert gu.same_bits(gpu.scan(ck.SCAN_NAMES[name], xv), oracle.scan(ck.SCAN_NAMES[name], x)), (n, name)
        for w in (1, 2, 3, 7, 64, 100, max(n, 1), n + 5):
            for name in ("sumw", "minw", "maxw"):
                assert gu.same_bits(gpu.scan(ck.SCAN_NAMES[name], xv, w), oracle.scan(ck.SCAN_NAMES[name], x, w)), (n, name, w)
            a, b = gpu.scan(ck.SCAN_AVGW, xv, w), oracle.scan(ck.SCAN_AVGW, x, w)
            assert np.all(np.abs(a - b) <= 1e-9 * np.maximum(1.0, np.abs(b)) * (np.arange(n) + 2)), (n, "avgw", w)
        f = np.round(rng.uniform(-50, 50, n), 3).astype(np.float64)
        fv = view(gpu, f, off)
        for name in ("mins", "maxs", "deltas"):
            assert gu.same_bits(gpu.scan(ck.SCAN_NAMES[name], fv), oracle.scan(ck.SCAN_NAMES[name], f)), (n, name, "f64")
        for w in (1, 3, 100):
            assert gu.same_bits(gpu.scan(ck.SCAN_MINW, fv, w), oracle.scan(ck.SCAN_MINW, f, w)), (n, "minw f64", w)


@pytest.mark.parametrize("off", [0, 1])
def test_gather_compact_sizes(gpu, off):
    rng = np.random.default_rng(300 + off)
    for n in SIZES:
        for dt in (np.int32, np.int8, np.float64):
            x = rand(rng, dt, max(n, 1), small=True)
            idx = rng.integers(0, len(x), n).astype(np.uint32)
            assert gu.same_bits(gpu.gather(view(gpu, x, off), view(gpu, idx, off)), x[idx]), (n, dt)
            m = rng.integers(0, 2, n).astype(np.uint8)
            xs = x[:n]
            assert gu.same_bits(gpu.compact(view(gpu, xs, off), view(gpu, m, off)), xs[m != 0]), (n, dt)
        m = rng.integers(0, 2, n).astype(np.uint8)
        assert np.array_equal(gpu.mask_to_index(view(gpu, m, off)), np.nonzero(m)[0].astype(np.uint32)), n


@pytest.mark.parametrize("off", [0, 1])
def test_groupby_and_join_sizes(gpu, oracle, off):
    rng = np.random.default_rng(400 + off)
    for n in SIZES:
        k1 = rng.integers(-3, 4, n).astype(np.int32)
        k2 = rng.integers(0, 3, n).astype(np.int16)
        v = rand(rng, np.int32, n, small=True)
        for keys in ([k1], [k1, k2]):
            o = oracle.groupby(keys)
            g = gpu.groupby_build([view(gpu, k, off) for k in keys])
            assert g.ngroups == o["ngroups"], n
            if n:
                assert np.array_equal(g.reversemap(), o["reversemap"]) and np.array_equal(g.counts(), o["counts"])
                offs, rows = g.postproc()
                assert np.array_equal(offs, np.concatenate([[0], np.cumsum(o["counts"])]).astype(np.uint32))
                for gi in range(o["ngroups"]):
                    seg = rows[offs[gi]:offs[gi + 1]]
                    assert np.all(np.diff(seg.astype(np.int64)) < 0) and np.all(o["reversemap"][seg] == gi)
            g.destroy()
            gb = gpu.groupby_agg([view(gpu, k, off) for k in keys], [ck.RED_SUM, ck.RED_MAX, ck.RED_COUNT], [view(gpu, v, off)] * 3)
            assert gb.ngroups == o["ngroups"], n
            if n:
                assert np.array_equal(gb.first_rows(), o["first_rows"])
                for j, op in enumerate((ck.RED_SUM, ck.RED_MAX, ck.RED_COUNT)):
                    assert gu.same_bits(gb.result(j, op, ck.INT32), oracle.grouped_reduce(op, v, o)), (n, op)
            gb.destroy()
        nb = n // 3
        bk = rng.permutation(np.arange(nb + 2, dtype=np.int32))[:nb]
        pk = rng.integers(0, nb + 4, n).astype(np.int32)
        look = gpu.join_lookup(view(gpu, bk, off), view(gpu, pk, off))
        pos = {int(k): i for i, k in enumerate(bk.tolist())}
        assert np.array_equal(look, np.array([pos.get(int(k), 0xFFFFFFFF) for k in pk.tolist()], dtype=np.uint32)), n


@pytest.mark.parametrize("dt", [np.int32, np.int16, np.float32, np.int64, np.float64])
def test_running_minmax_changes_across_many_links(gpu, oracle, dt):
    """mins / maxs of a drifting series whose running extreme keeps changing to the last row: every chain link of the
    single-pass scan needs the prefix handed over by its predecessors, over several 64-link look-back windows
    (aggregations.h:350-381: mins seeds with max, maxs with min)"""
    n = 6_000_011                       # 367 links of 16384 rows (4-byte types), 733 links of 8192 rows (8-byte types)
    rng = np.random.default_rng(11)
    span = 30000 if np.dtype(dt).itemsize == 2 else 1_000_000
    drift = np.linspace(span, -span, n)
    x = (drift + rng.integers(-span // 50, span // 50, n)).astype(dt)
    for arr in (x, x[::-1].copy()):
        d = gpu.to_device(arr)
        assert gu.same_bits(gpu.scan(ck.SCAN_MINS, d), oracle.scan(ck.SCAN_MINS, arr))
        assert gu.same_bits(gpu.scan(ck.SCAN_MAXS, d), oracle.scan(ck.SCAN_MAXS, arr))
        d.free()


@pytest.mark.parametrize("dt", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.float32, np.float64])
def test_scan_resume_every_dtype_and_tile_boundary(gpu, oracle, dt):
    """aqg_scan_resume: sums / avgs of the second part of a column, resumed from the carry of the first part, equal the tail of
    the whole-column scan -- for every numeric dtype and for splits around the 2048-row tile (aggregations.h:203-236)"""
    rng = np.random.default_rng(3)
    n = 10_000
    x = rand(rng, dt, n, small=np.dtype(dt).kind == "f")
    for split in (0, 1, 2047, 2048, 2049, 9_999):
        head, tail = x[:split], np.ascontiguousarray(x[split:])
        for name in ("sums", "avgs"):
            op = ck.SCAN_NAMES[name]
            want = oracle.scan(op, x)[split:]
            if np.dtype(dt).kind == "f":
                carry = float(np.sum(head.astype(np.float64))) if split else -0.0
            else:
                carry = sum(int(v) for v in head.tolist())
            got = gpu.scan_resume(op, tail, carry, split)
            if np.dtype(dt).kind == "f":
                assert np.all(np.abs(got - want) <= 1e-9 * np.maximum(np.abs(want), 1.0)), (name, split)
            else:
                assert gu.same_bits(got, want), (name, split)


def test_col_pin_async_upload_is_stream_ordered(gpu, oracle):
    """aqg_col_pin returns with the upload still running (page-locked chunks, DMA on a copy stream); calls that follow read the
    complete column (the context's stream waits for it), for page-aligned and unaligned host ranges, sizes below and above the
    registration threshold and a second sight of the same column (cached)"""
    rng = np.random.default_rng(17)
    base = rng.integers(-1000, 1000, 9_000_011).astype(np.int32)
    alive = []                      # the mirror cache is keyed by host address: borrowed columns live until the session ends
    al = (-base.ctypes.data % 4096) // 4          # first page-aligned element: a copy that starts in page-locked memory must not run past it
    for off, n in ((0, 9_000_011), (3, 5_000_000), (1025, 1_000), (7, 300_001), (al, 4_000_003), (al, 1 << 20), (al + 1024, 2_097_152 + 5), (al + 1, 3_000_000)):
        a = base[off:off + n]
        d = gpu.col_pin(a)
        assert int(gpu.reduce(ck.RED_SUM, d)) == int(a.sum(dtype=np.int64))
        d2 = gpu.col_pin(a)
        assert d2.ptr == d.ptr
        hk = np.ascontiguousarray(a % 7)
        alive.append(hk)
        k = gpu.col_pin(hk)
        gb = gpu.groupby_agg([k], [ck.RED_SUM], [d], hint=0)
        o = oracle.groupby([hk])
        assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
        gb.destroy()
    gpu.col_unpin_all()


def test_col_pin_of_memory_somebody_else_has_page_locked(gpu, oracle):
    """pages the HOST APPLICATION has page-locked (here: the test, through hipHostRegister) are neither registered again nor copied
    from directly -- the runtime keeps registrations in a map keyed by their start, overlapping ones cannot be undone, and a direct copy
    that starts in locked memory and runs past its end faults (profiles/r3_hostregister_abort.md): such chunks are staged.  Whole
    column, a slice that starts inside the locked part and ends behind it, one that ends inside it; then Q1 over the mirrors."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    rng = np.random.default_rng(23)
    base = rng.integers(-1000, 1000, 10_000_000).astype(np.int32)
    addr = base.ctypes.data
    lo = (addr + (8 << 20) + 4095) & ~4095                       # [lo, lo + 16 MB): locked by "the application"
    assert hip.hipHostRegister(ctypes.c_void_p(lo), ctypes.c_size_t(16 << 20), ctypes.c_uint(0)) == 0
    try:
        i_lo = (lo - addr) // 4
        alive = []
        for off, n in ((0, 10_000_000), (i_lo + 1000, 6_000_000), (5, i_lo + 2_000_000), (i_lo, 1 << 20)):
            a = base[off:off + n]
            d = gpu.col_pin(a)
            assert gpu.col_pin_last() == (0, 1, 0), gpu.col_pin_last()       # the one chunk touches foreign locked pages: staged
            assert int(gpu.reduce(ck.RED_SUM, d)) == int(a.sum(dtype=np.int64)), (off, n)
            hk = np.ascontiguousarray(a % 7)
            alive.append(hk)
            gb = gpu.groupby_agg([gpu.col_pin(hk)], [ck.RED_SUM], [d], hint=0)
            assert gpu.col_pin_last()[1] == 0                                  # (the key column is memory of its own)
            o = oracle.groupby([hk])
            assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
            gb.destroy()
        gpu.col_unpin_all()
    finally:
        assert hip.hipHostUnregister(ctypes.c_void_p(lo)) == 0   # the application's registration is still its own to undo


def test_two_contexts_pin_overlapping_slices_of_one_array(oracle):
    """two contexts of one process (one per GPU thread; the header layer's runtime next to a harness's) page-lock through ONE registry:
    the second context's overlapping slice is staged instead of being registered over the first's pages; unpinning in either order
    and pinning again works (before: each context only knew its own registrations)"""
    import threading
    import aquery2_amd
    rng = np.random.default_rng(29)
    base = rng.integers(-1000, 1000, 12_000_000).astype(np.int32)
    devs = [aquery2_amd.Device(0), aquery2_amd.Device(0)]
    slices = [base[0:8_000_000], base[2_000_000:11_000_000]]
    try:
        for order in ((0, 1), (1, 0)):
            out, how = [None, None], [None, None]
            def body(r):
                d = devs[r].col_pin(slices[r])
                how[r] = devs[r].col_pin_last()
                out[r] = int(devs[r].reduce(ck.RED_SUM, d))
            th = [threading.Thread(target=body, args=(r,)) for r in range(2)]
            for t in th: t.start()
            for t in th: t.join()
            assert out == [int(s.sum(dtype=np.int64)) for s in slices]
            assert sorted(how) == [(0, 1, 0), (1, 0, 0)], how               # whoever came second found the first's pages and staged its chunk
            for r in order:
                devs[r].col_unpin_all()
    finally:
        for d in devs: d.close()


@pytest.mark.parametrize("off", [0, 1])
def test_long_window_minmax(gpu, oracle, off):
    """minw / maxw with w >= 128 (the van Herk / Gil-Werman kernel): every element type, windows around the segment and tile
    borders, inputs shorter than / equal to / a few rows longer than one tile, misaligned columns (aggregations.h:127-167)"""
    rng = np.random.default_rng(300 + off)
    for dt in (np.int32, np.int8, np.uint16, np.int64, np.uint64, np.float32, np.float64):
        fp = np.dtype(dt).kind == "f"
        for n, ws in ((6169, (128, 1000)), (6170, (129, 1023)), (20_011, (128, 1024, 2047, 3000, 5000)), (300_007, (1000, 2500))):
            x = np.round(rng.uniform(-1000, 1000, n), 3).astype(dt) if fp else rand(rng, dt, n, small=False)
            xv = view(gpu, x, off)
            for w in ws:
                for name in ("minw", "maxw"):
                    assert gu.same_bits(gpu.scan(ck.SCAN_NAMES[name], xv, w), oracle.scan(ck.SCAN_NAMES[name], x, w)), (dt, n, name, w)
    # a sorted and a reverse-sorted series: every window's best sits at one of its ends
    x = np.arange(50_000, dtype=np.int32)
    for y in (x, x[::-1].copy()):
        yv = view(gpu, y, off)
        for w in (128, 777, 4096):
            assert gu.same_bits(gpu.scan(ck.SCAN_MAXW, yv, w), oracle.scan(ck.SCAN_MAXW, y, w)), w
            assert gu.same_bits(gpu.scan(ck.SCAN_MINW, yv, w), oracle.scan(ck.SCAN_MINW, y, w)), w
