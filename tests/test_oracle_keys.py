"""Group-by over key columns that are not plain integers (reference server/hasher.h:97-144 + tuple ==): the oracle's restatement
of the equality rules against golden ids dumped from the REAL reference (tests/golden/ref_golden_keys.json, made by
oracle/gen_golden_keys.py over the seeded inputs of tests/keycases.py), and against the reference itself where it is built."""
import json
import os

import numpy as np

import checker as ck
import keycases

GOLD = {c["name"]: c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden_keys.json")))["cases"]}


def test_oracle_matches_the_reference_golden_ids(oracle):
    cases = keycases.cases()
    assert len(cases) == len(GOLD) >= 16
    for name, cols in cases:
        r, g = oracle.groupby_typed(cols), GOLD[name]
        assert r["ngroups"] == g["ngroups"], name
        assert r["reversemap"].tolist() == g["reversemap"], name
        assert r["first_rows"].tolist() == g["first_rows"], name


def test_quirks_pinned_by_the_reference(oracle):
    """what the golden ids say about the reference: 0.0 and -0.0 are one group, every NaN is its own, the padding byte of time_t
    does not matter, string views compare contents"""
    d = np.array([0.0, -0.0, 0.0, np.nan, np.nan, 1.5, -0.0, 1.5])
    assert oracle.groupby_typed([(ck.DOUBLE, d)])["reversemap"].tolist() == [0, 0, 0, 1, 2, 3, 0, 3]
    t = np.zeros((3, 8), np.uint8); t[:, 0] = 5; t[1, 7] = 0xAB; t[2, 6] = 1
    assert oracle.groupby_typed([(ck.TIME, t)])["reversemap"].tolist() == [0, 0, 1]
    assert oracle.groupby_typed([(ck.STR, [b"abc", b"abc", b"abd", b"", b"ab", b"", b"abd"])])["reversemap"].tolist() == [0, 0, 1, 2, 3, 2, 1]
    g = GOLD["float64"]
    assert g["ngroups"] > 30          # the NaN rows of the seeded case each made a group


def test_oracle_vs_real_reference_where_built(oracle, ref):
    if ref is None:
        import pytest
        pytest.skip("oracle/_ref not built here")
    for name, cols in keycases.cases():
        a, b = oracle.groupby_typed(cols), ref.groupby_typed(cols)
        assert a["ngroups"] == b["ngroups"] and np.array_equal(a["reversemap"], b["reversemap"]) and np.array_equal(a["first_rows"], b["first_rows"]), name
    # raw string pointers are POINTERS under the reference's tuple == (mem_opt.cpp:22 groups by ColRef<const char*>)
    import ctypes as C
    bufs = [C.create_string_buffer(b) for b in (b"abc", b"abc", b"abd")]
    ptrs = np.array([C.addressof(bufs[0]), C.addressof(bufs[1]), C.addressof(bufs[2]), C.addressof(bufs[0])], np.uint64)
    assert ref.groupby_typed([(ck.UINT64, ptrs)])["reversemap"].tolist() == [0, 1, 2, 0]
