"""Decode tests/golden/ref_golden.json (made by oracle/gen_golden.py from the real reference)."""
import json
import os

import numpy as np

import checker as ck

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden.json")
_NAME2DT = {"int128": ck.I128, "uint128": ck.U128, "bool": np.dtype(np.uint8)}


def dec(e):
    dt = _NAME2DT.get(e["dtype"], None) or np.dtype(e["dtype"])
    return np.frombuffer(bytes.fromhex(e["hex"]), dtype=dt, count=e["n"]).copy()


def dec_scalar(e):
    if "int" in e:
        return int(e["int"])
    return dec(e)[0]


def load():
    with open(PATH) as f:
        return json.load(f)["cases"]


def operand(c, side):
    a = dec(c[side])
    return a[0] if c[side + "_scalar"] else a


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.dtype.itemsize == b.dtype.itemsize and a.tobytes() == b.tobytes()


def scalar_same(got, want):
    if isinstance(want, int) and not isinstance(want, (np.integer,)):
        return int(got) == want
    return np.asarray(got).tobytes() == np.asarray(want).tobytes()
