"""Seeded inputs of the key-type parity cases (group-by over key columns that are not plain integers).  The same function feeds
oracle/gen_golden_keys.py (expected ids dumped from the real reference into tests/golden/ref_golden_keys.json), the oracle tests
and the GPU tests; only inputs live here -- the expectations come from the reference."""
import numpy as np

import checker as ck


def _dates(rng, n, distinct):
    pool = np.zeros((distinct, 4), np.uint8)
    pool[:, 0] = rng.integers(1, 29, distinct); pool[:, 1] = rng.integers(1, 13, distinct)
    pool[:, 2:4] = rng.integers(1990, 2030, distinct).astype("<i2").view(np.uint8).reshape(-1, 2)
    return pool[rng.integers(0, distinct, n)]


def _times(rng, n, distinct):
    pool = np.zeros((distinct, 8), np.uint8)
    pool[:, 0:4] = rng.integers(0, 1000, distinct).astype("<u4").view(np.uint8).reshape(-1, 4)
    pool[:, 4] = rng.integers(0, 60, distinct); pool[:, 5] = rng.integers(0, 60, distinct); pool[:, 6] = rng.integers(0, 24, distinct)
    t = pool[rng.integers(0, distinct, n)].copy()
    t[:, 7] = rng.integers(0, 256, n)            # junk in the padding byte: the reference compares fields only
    return t


def cases():
    rng = np.random.default_rng(20260)
    n = 3000
    out = []
    for tag, dt in ((ck.DOUBLE, np.float64), (ck.FLOAT, np.float32)):
        v = rng.integers(-6, 7, n).astype(dt) / 2
        v[rng.integers(0, n, 40)] = np.nan
        v[rng.integers(0, n, 40)] = -0.0
        w = v.copy(); w[::7] = -np.nan
        j = rng.integers(0, 3, n).astype(np.int32)
        out += [(f"{dt.__name__}", [(tag, v)]), (f"{dt.__name__}_negnan", [(tag, w)]), (f"{dt.__name__}_int", [(tag, v), (ck.INT32, j)])]
        out.append((f"{dt.__name__}_plain", [(tag, rng.integers(-50, 50, n).astype(dt) * 0.25 + 0.125)]))
    d, t = _dates(rng, n, 90), _times(rng, n, 120)
    ts = np.concatenate([d, t], axis=1)
    j = rng.integers(0, 4, n).astype(np.int32)
    out += [("date", [(ck.DATE, d)]), ("time_padding_junk", [(ck.TIME, t)]), ("timestamp", [(ck.TIMESTAMP, ts)]),
            ("date_int", [(ck.DATE, d), (ck.INT32, j)]), ("timestamp_int", [(ck.TIMESTAMP, ts), (ck.INT32, j)])]
    big = np.zeros(n, ck.I128)
    big["lo"] = rng.integers(0, 4, n).astype(np.uint64) * np.uint64(0x8000000000000001)
    big["hi"] = rng.integers(-2, 3, n)
    out.append(("int128_int", [(ck.INT128, big), (ck.INT32, j)]))
    words = [b"", b"a", b"ab", b"abc", b"abd", b"jan", b"feb", b"a much longer month name than the others", b"\xc3\xa9t\xc3\xa9"]
    s = [words[i] for i in rng.integers(0, len(words), n)]
    out += [("astring_view", [(ck.STR, s)]), ("astring_view_int", [(ck.STR, s), (ck.INT32, j)])]
    return out
