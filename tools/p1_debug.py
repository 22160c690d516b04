"""scratch: one-level partition plan against the oracle on a few shapes (run on the GPU box)"""
import sys, os
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd as A
import checker as ck
from test_gpu_basic import rand
d = A.Device(0)
orc = ck.load_oracle()
n, card = 1_300_003, 60_000
rng = np.random.default_rng(card)
k = rng.integers(-card // 2, card // 2, n).astype(np.int32)
vals = [rand(rng, np.int32, n, small=True)]
o = orc.groupby([k])
for v in vals:
  for sel in (["sum"], ["count"]):
    ops = [ck.RED_NAMES[x] for x in sel]
    try:
        gb = d.groupby_agg([k], ops, [v] * len(ops), hint=card)
        ok = gb.ngroups == o["ngroups"] and np.array_equal(gb.first_rows(), o["first_rows"])
        print(v.dtype, sel, "groups", gb.ngroups, o["ngroups"], "ok" if ok else "MISMATCH", flush=True)
    except Exception as e:
        print(v.dtype, sel, "FAILED", e, flush=True)
