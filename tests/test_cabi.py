"""CPU-side checks of the drop-in boundary: libaqg.so loads and exports every symbol include/aqg.h declares;
the pure type-rule entry points agree with the oracle (no GPU needed); the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import checker as ck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "aqg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aqg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import aquery2_amd
    lib = aquery2_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_type_rules_match_oracle(oracle):
    import aquery2_amd
    lib = aquery2_amd.load_library()
    tags = [ck.INT8, ck.INT16, ck.INT32, ck.INT64, ck.UINT8, ck.UINT16, ck.UINT32, ck.UINT64, ck.FLOAT, ck.DOUBLE]
    for a in tags:
        assert lib.aqg_long_type(a) == oracle.long_type(a)
        assert lib.aqg_fp_type(a) == oracle.fp_type(a)
        assert lib.aqg_dtype_size(a) == ck.TAG2NP[a].itemsize
        for b in tags:
            assert lib.aqg_coercion(a, b) == oracle.coercion(a, b), (a, b)
            for op in range(14):
                assert lib.aqg_ewise_out_dtype(op, a, b) == oracle.ewise_out_dtype(op, a, b), (op, a, b)
        for op in range(9):
            assert lib.aqg_reduce_out_dtype(op, a) == oracle.reduce_out_dtype(op, a)
        for op in range(16):
            assert lib.aqg_scan_out_dtype(op, a) == oracle.scan_out_dtype(op, a)


def test_no_cpu_fallback_without_gpu():
    import aquery2_amd
    lib = aquery2_amd.load_library()
    if lib.aqg_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(aquery2_amd.AqgError):
        aquery2_amd.Device(0)


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under aquery2_amd/ or include/ may import, include or link it"""
    bad = []
    for top in ("aquery2_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            if "build" in dp.split(os.sep):
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"#include\s+[\"<][^\">]*oracle|import\s+checker|liboracle|libaqref|aq_oracle\.h", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
