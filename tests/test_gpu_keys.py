"""GPU parity for key columns that are not plain integers (missing #2): floating keys (0.0 / -0.0 one group, NaNs
singletons), date_t / time_t / timestamp_t, __int128, astring_view (dictionary codes) -- aqg_groupby_agg and aqg_groupby_build
against the ids the REAL reference produced (tests/golden/ref_golden_keys.json) on the same seeded inputs."""
import json
import os

import numpy as np
import pytest

import checker as ck
import keycases

pytestmark = pytest.mark.gpu
GOLD = {c["name"]: c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden_keys.json")))["cases"]}


@pytest.fixture(scope="module")
def gpu():
    import aquery2_amd
    d = aquery2_amd.Device(0)
    yield d
    d.close()


def dev_cols(gpu, cols):
    out = []
    for tag, data in cols:
        if tag in (ck.DATE, ck.TIME, ck.TIMESTAMP, ck.STR):
            out.append(gpu.key_col(tag, data))
        else:
            out.append(gpu.to_device(data))
    return out


def elem_bytes(tag, data):
    if tag == ck.STR: return 4
    if tag in ck.KEY_ELEM_BYTES: return ck.KEY_ELEM_BYTES[tag]
    return np.asarray(data).dtype.itemsize


@pytest.mark.parametrize("name", sorted(GOLD))
def test_key_types_against_reference_ids(gpu, name):
    cols = dict(keycases.cases())[name]
    g = GOLD[name]
    n = len(g["reversemap"])
    dk = dev_cols(gpu, cols)
    v = np.arange(n, dtype=np.int32) % 1000 - 300
    first, rev = np.array(g["first_rows"], np.uint32), np.array(g["reversemap"], np.int64)
    want_sum = np.bincount(rev, weights=v.astype(np.float64), minlength=g["ngroups"]).astype(np.int64)
    want_cnt = np.bincount(rev, minlength=g["ngroups"])
    for hint in (0, 64):
        gb = gpu.groupby_agg(dk, [ck.RED_SUM, ck.RED_COUNT], [v, v], hint=hint)
        assert gb.ngroups == g["ngroups"]
        assert np.array_equal(gb.first_rows(), first)
        assert ck.i128_to_int(gb.result(0, ck.RED_SUM, ck.INT32)) == want_sum.tolist()
        assert np.array_equal(gb.result(1, ck.RED_COUNT, ck.INT32), want_cnt.astype(np.uint64))
        for k, (tag, data) in enumerate(cols):
            if tag == ck.STR:
                continue                                  # the code column's keys are codes; the header layer maps them back through first rows
            eb = elem_bytes(tag, data)
            host = np.ascontiguousarray(data).reshape(n, -1).view(np.uint8).reshape(n, eb)
            assert np.array_equal(gb.keys_raw(k, eb), host[first]), (name, k)
        gb.destroy()
    b = gpu.groupby_build(dk)
    assert b.ngroups == g["ngroups"]
    assert np.array_equal(b.reversemap(), np.array(g["reversemap"], np.uint32))
    assert np.array_equal(b.counts(), want_cnt.astype(np.uint32))
    b.destroy()


def test_floating_keys_without_special_values_take_the_bit_patterns(gpu, oracle):
    rng = np.random.default_rng(5)
    n = 2_000_003
    k = (rng.integers(0, 5000, n) * 0.5 + 0.25).astype(np.float64)
    v = rng.integers(-9, 9, n).astype(np.int32)
    gb = gpu.groupby_agg([k], [ck.RED_SUM], [v], hint=0)
    o = oracle.groupby_typed([(ck.DOUBLE, k)])
    assert gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
    assert np.array_equal(gb.keys(0, np.float64), k[o["first_rows"]])
    gb.destroy()


def _pooled_strings(rng, n, npool, dup_every=7):
    """n row pointers into a pool of short NUL-terminated strings; every `dup_every`-th pool entry repeats the CONTENT of another entry
    behind its own pointer (astring_view keys compare content, server/types.h:281-334).  Returns (pointer array, expected codes)."""
    import ctypes as C
    contents = [b"sym%07d" % (i * 7919 % 10_000_019) if i % 3 else b"s%d" % i for i in range(npool)]
    for i in range(dup_every, npool, dup_every):
        contents[i] = contents[i - dup_every + 1]
    bufs = [C.create_string_buffer(c) for c in contents]
    addrs = np.array([C.addressof(b) for b in bufs], dtype=np.uint64)
    canon = {}
    cid = np.array([canon.setdefault(c, len(canon)) for c in contents], dtype=np.int64)
    idx = rng.integers(0, npool, n)
    ptrs = np.ascontiguousarray(addrs[idx])
    cont = cid[idx]
    _, first = np.unique(cont, return_index=True)
    rank = np.empty(len(canon), np.int64); rank[:] = -1
    rank[cont[np.sort(first)]] = np.arange(first.size)
    return ptrs, rank[cont].astype(np.uint32), bufs


@pytest.mark.parametrize("n,npool", [(70_000, 50), (3_000_000, 200_000)])
def test_string_dictionary_built_on_the_device(gpu, n, npool):
    """aqg_str_encode from 2^16 rows on: bytes + offsets uploaded once, hashed on the device, numbered by first occurrence through the
    group-by build over {hash, length}, every row compared with its group's first row -- codes equal to content-equality ids"""
    import ctypes as C
    rng = np.random.default_rng(n)
    ptrs, want, keep = _pooled_strings(rng, n, npool)
    out = gpu.empty(n, np.uint32)
    nd = C.c_uint32()
    gpu._chk(gpu.lib.aqg_str_encode(gpu.ctx, C.c_void_p(ptrs.ctypes.data), C.c_uint32(n), C.c_void_p(out.ptr), C.byref(nd)), "aqg_str_encode")
    got = out.to_host()
    assert nd.value == int(want.max()) + 1
    assert np.array_equal(got, want)


def test_string_dictionary_host_and_device_paths_agree():
    """the same column through the host map (AQG_STR_HOST=1) in a fresh process: identical codes"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "tests")
import aquery2_amd
from test_gpu_keys import _pooled_strings
gpu = aquery2_amd.Device(0)
ptrs, want, keep = _pooled_strings(np.random.default_rng(1), 400_000, 3000)
out = gpu.empty(400_000, np.uint32)
gpu._chk(gpu.lib.aqg_str_encode(gpu.ctx, C.c_void_p(ptrs.ctypes.data), C.c_uint32(400_000), C.c_void_p(out.ptr), None), "aqg_str_encode")
assert np.array_equal(out.to_host(), want)
print("OK")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, AQG_STR_HOST="1"), cwd=root)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
