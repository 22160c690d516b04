#!/usr/bin/env python3
"""Generate tests/golden/ref_golden_keys.json from the REAL reference library: group ids (first-occurrence order) and first rows
of AQHashTable over key columns that are not plain integers -- floating columns (0.0 / -0.0 / NaN), date_t, time_t (with junk in
its padding byte), timestamp_t, __int128, astring_view, raw string pointers -- alone and paired with an int column
(reference server/hasher.h:66-199).  Runs only where oracle/_ref/libaqref.so exists.
    python oracle/gen_golden_keys.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checker as ck  # noqa: E402
import keycases  # noqa: E402

ref = ck.load_ref()
if ref is None:
    sys.exit("oracle/_ref/libaqref.so missing: run `make -C oracle` where /root/reference is mounted")
out = []
for name, cols in keycases.cases():
    r = ref.groupby_typed(cols)
    out.append({"name": name, "ngroups": int(r["ngroups"]), "reversemap": r["reversemap"].tolist(), "first_rows": r["first_rows"].tolist()})
path = os.path.join(ROOT, "tests", "golden", "ref_golden_keys.json")
with open(path, "w") as f:
    json.dump({"generator": "oracle/gen_golden_keys.py", "inputs": "tests/keycases.py (seeded)", "cases": out}, f)
print(len(out), "cases ->", path)
