import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import aquery2_amd as A, checker as ck
d = A.Device(0)
K = 100
n = 2**32 - 2**20 - 1
try:
    d.gen_column(ck.GEN_V1, 42, 0, 2**32 - 1, 2**32 - 1, K)
except Exception as e:
    print("expected:", e)
id1 = d.gen_column(ck.GEN_ID1, 42, 0, n, n, K)
v1 = d.gen_column(ck.GEN_V1, 42, 0, n, n, K)
print("tags", id1.tag, v1.tag, v1.n, hex(v1.ptr))
print(d.reduce(ck.RED_SUM, v1))
