#!/usr/bin/env python3
"""Generate tests/golden/ref_golden.json from the REAL reference library.

Runs only where oracle/_ref/libaqref.so exists (built by oracle/Makefile from the
reference sources under /root/reference).  The fixture holds inputs and the
reference's outputs as raw little-endian bytes (hex), so comparisons are bit-exact.
Inputs: the reference's own tiny data files (data/moving_avg.csv, data/test.csv, the
rows of tests/stock.a -- values typed in below as data, not code) plus seeded random
columns.
    python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checker as ck  # noqa: E402

ref = ck.load_ref()
if ref is None:
    sys.exit("oracle/_ref/libaqref.so missing: run `make -C oracle` where /root/reference is mounted")

TAGNAME = {ck.INT8: "int8", ck.INT16: "int16", ck.INT32: "int32", ck.INT64: "int64", ck.UINT8: "uint8",
           ck.UINT16: "uint16", ck.UINT32: "uint32", ck.UINT64: "uint64", ck.FLOAT: "float32",
           ck.DOUBLE: "float64", ck.BOOL: "bool", ck.INT128: "int128", ck.UINT128: "uint128"}


def enc(a):
    a = np.ascontiguousarray(a)
    name = "int128" if a.dtype == ck.I128 else "uint128" if a.dtype == ck.U128 else a.dtype.name
    return {"dtype": name, "n": int(a.size), "hex": a.tobytes().hex()}


def enc_scalar(v, tag):
    if tag in (ck.INT128, ck.UINT128):
        return {"dtype": TAGNAME[tag], "int": str(int(v))}
    return enc(np.array([v], dtype=ck.TAG2NP[tag]))


cases = []

# ---- the reference's own fixtures ------------------------------------------------------------
# data/moving_avg.csv (Month,sales): file order, and ascending Month (what ASSUMING ASC Mont yields)
mavg_file = np.array([100, 140, 130, 140, 120], np.int32)
mavg_sorted = np.array([100, 120, 140, 140, 130], np.int32)
# tests/stock.a:3-18 rows (timestamp 1..16, price)
stock_price = np.array([15, 19, 16, 17, 15, 13, 5, 8, 7, 13, 11, 14, 10, 5, 2, 5], np.int32)
stock_ts = np.arange(1, 17, dtype=np.int32)
# data/test.csv columns a,b,c,d (20 rows)
test_csv = np.loadtxt(os.path.join("/root/reference", "data", "test.csv"), delimiter=",", skiprows=1, dtype=np.int32)


def add_scan(name, x, w, src):
    op = ck.SCAN_NAMES[name]
    out = ref.scan(op, x, w)
    cases.append({"fn": "scan", "op": name, "w": int(w), "x": enc(x), "out": enc(out), "src": src})


def add_reduce(name, x, src):
    op = ck.RED_NAMES[name]
    t = ck.tag_of(x)
    out = ref.reduce(op, x)
    cases.append({"fn": "reduce", "op": name, "x": enc(x), "out": enc_scalar(out, ref.reduce_out_dtype(op, t)), "src": src})


def add_ewise(name, l, r, src, ot=None):
    out = ref.ewise(ck.OP_NAMES[name], l, r, ot=ot)
    cases.append({"fn": "ewise", "op": name, "l": enc(np.atleast_1d(l)), "l_scalar": bool(np.ndim(l) == 0),
                  "r": enc(np.atleast_1d(r)), "r_scalar": bool(np.ndim(r) == 0), "out": enc(out),
                  "ot": None if ot is None else TAGNAME[ot], "src": src})


def add_groupby(keys, vals, src):
    gb = ref.groupby(keys)
    c = {"fn": "groupby", "keys": [enc(k) for k in keys], "ngroups": gb["ngroups"], "src": src}
    for f in ("reversemap", "counts", "first_rows", "offsets", "row_ids"):
        c[f] = enc(gb[f])
    aggs = []
    for v in vals:
        for name in ("sum", "min", "max", "count", "avg", "first", "last", "var"):
            aggs.append({"op": name, "x": enc(v), "out": enc(ref.grouped_reduce(ck.RED_NAMES[name], v, gb))})
    c["aggs"] = aggs
    cases.append(c)


scan_ops = ["sums", "avgs", "mins", "maxs", "sumw", "avgw", "minw", "maxw", "ratiow", "deltas", "prev", "aggnext"]
for src, arr in (("data/moving_avg.csv sorted by Month", mavg_sorted), ("data/moving_avg.csv file order", mavg_file),
                 ("tests/stock.a price", stock_price)):
    for name in scan_ops:
        for w in (1, 2, 3, 5, 10, 100):
            add_scan(name, arr, w, src)
    for name in ck.RED_NAMES:
        add_reduce(name, arr, src)

# stock.a q1..q4 building blocks: price - min(timestamp) ; price - mins(price) ; price - timestamp > 1
add_ewise("sub", stock_price, np.int32(1), "tests/stock.a q1: price - min(timestamp)")
add_ewise("sub", stock_price, np.ascontiguousarray(ref.scan(ck.SCAN_MINS, stock_price)), "tests/stock.a q2: price - mins(price)")
add_ewise("sub", stock_price, stock_ts, "tests/stock.a q3: price - timestamp")
add_ewise("gt", stock_price - stock_ts, np.int32(1), "tests/stock.a q3: price - timestamp > 1")
add_ewise("mul", stock_price, stock_ts, "tests/stock.a q3: price * timestamp")
rev = stock_price[::-1].copy()
add_ewise("sub", rev, np.ascontiguousarray(ref.scan(ck.SCAN_MINS, rev)), "tests/stock.a q4: desc timestamp")

# tests/q1.sql: group by a,b,d over data/test.csv, sum(c)
a, b, c, d = (np.ascontiguousarray(test_csv[:, i]) for i in range(4))
add_groupby([a, b, d], [c], "tests/q1.sql on data/test.csv (group by a,b,d)")
add_groupby([a], [c, d], "data/test.csv group by a")
add_groupby([a, b], [c], "data/test.csv group by a,b")

# quirks pinned by the survey (D8, float sum)
add_reduce("max", np.array([-1.0, -2.0], np.float64), "D8: max seeds with numeric_limits<T>::min()")
add_reduce("max", np.array([-1.0, -2.0], np.float32), "D8 float")
add_reduce("max", np.array([-5, -7], np.int32), "max of negatives (int)")
add_reduce("sum", np.array([0.1, 0.2, 0.3], np.float32), "sum(float) accumulates in double")
add_reduce("sum", np.array([4000000000, 4000000000], np.uint32), "sum(unsigned) is uint128")
for name in ck.RED_NAMES:
    if name != "avg":
        add_reduce(name, np.array([], np.int32), "empty column")

# ---- seeded random columns ---------------------------------------------------------------------
rng = np.random.default_rng(42)
NUM = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]


def rnd(dt, n):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return np.round(rng.uniform(0, 100, n), 6).astype(dt)
    return rng.integers(1, min(np.iinfo(dt).max, 500), n, endpoint=True).astype(dt)


for dt in NUM:
    x = rnd(dt, 61)
    for name in ck.RED_NAMES:
        add_reduce(name, x, "random seed 42")
    for name in scan_ops:
        for w in (1, 4, 61, 200):
            add_scan(name, x, w, "random seed 42")
BIN = [np.int16, np.int32, np.int64, np.uint32, np.float32, np.float64]
for lt in BIN:
    for rt in BIN:
        l, r = rnd(lt, 41), rnd(rt, 41)
        for name in ("add", "sub", "mul", "div", "gt"):
            add_ewise(name, l, r, "random seed 42")
            add_ewise(name, l, r[3], "random seed 42 vec-scalar")
            add_ewise(name, l[5], r, "random seed 42 scalar-vec")
        for name in ("lt", "ge", "le", "eq", "ne"):
            add_ewise(name, l, r, "random seed 42 aqop", ot=ck.BOOL)
for nk, n, card in ((1, 500, 7), (1, 2000, 100), (2, 1500, 12), (3, 1200, 5), (6, 800, 2), (1, 300, 300)):
    keys = [rng.integers(1, card, n, endpoint=True).astype(np.int32) for _ in range(nk)]
    vals = [rng.integers(1, 5, n, endpoint=True).astype(np.int32), np.round(rng.uniform(0, 100, n), 6).astype(np.float32)]
    add_groupby(keys, vals, f"random h2o-shaped nk={nk} n={n} card={card}")

# corr / gather / compact / hash
for lt in (np.int32, np.float32, np.float64):
    for rt in (np.int32, np.float32):
        x, y = rnd(lt, 200), rnd(rt, 200)
        cases.append({"fn": "corr", "x": enc(x), "y": enc(y), "out": enc(np.array([ref.corr(x, y)])), "src": "random seed 42"})
x = rnd(np.int32, 100)
idx = rng.integers(0, 100, 250).astype(np.uint32)
cases.append({"fn": "gather", "x": enc(x), "idx": enc(idx), "out": enc(ref.gather(x, idx)), "src": "random"})
mask = rng.integers(0, 2, 100).astype(np.uint8)
cases.append({"fn": "compact", "x": enc(x), "mask": enc(mask), "out": enc(ref.compact(x, mask)), "src": "random (selected tail only, D11)"})
cases.append({"fn": "hash_scalar", "v": 7, "out": str(ref.hash_scalar(np.int32(7)))})
cases.append({"fn": "hash_tuple", "v": [3, 4], "out": str(ref.hash_tuple([np.int32(3), np.int32(4)]))})

out = os.path.join(ROOT, "tests", "golden", "ref_golden.json")
with open(out, "w") as f:
    json.dump({"generator": "oracle/gen_golden.py", "reference": "sunyinqi0508/AQuery2 @ /root/reference, g++ -O2 -std=c++20",
               "cases": cases}, f, separators=(",", ":"))
print(f"{len(cases)} cases -> {out} ({os.path.getsize(out)} bytes)")
