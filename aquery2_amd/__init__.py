"""aquery2_amd -- MI355X-native execution library for the AQuery column-batch hot path.

The product is the C-ABI shared library `libaqg.so` (include/aqg.h) built from the hand-written
HIP kernels under aquery2_amd/csrc/, plus the host C++ headers under include/aquery/ that mirror
the reference's header-level API.  This Python package is a thin ctypes harness over the C-ABI
used by tests/ and bench.py; it contains no compute of its own and no CPU fallback.
"""
from . import capi  # noqa: F401
from .capi import Device, DevBuf, AqgError, Comm, ThreadRanks, lib_path, load_library  # noqa: F401
