"""The C restatement (oracle/) against the committed golden vectors that were dumped from the
REAL reference library (tests/golden/ref_golden.json, oracle/gen_golden.py) -- runs anywhere,
including the GPU box where /root/reference does not exist -- plus the KATs SURVEY.md 8c lists."""
import numpy as np
import pytest

import checker as ck
import golden_util as gu

CASES = gu.load()


def by_fn(fn):
    return [c for c in CASES if c["fn"] == fn]


def test_golden_scans(oracle):
    cs = by_fn("scan")
    assert len(cs) > 300
    for c in cs:
        got = oracle.scan(ck.SCAN_NAMES[c["op"]], gu.dec(c["x"]), c["w"])
        assert gu.same_bits(got, gu.dec(c["out"])), (c["op"], c["w"], c["x"]["dtype"], c["src"])


def test_golden_reductions(oracle):
    cs = by_fn("reduce")
    assert len(cs) > 100
    for c in cs:
        got = oracle.reduce(ck.RED_NAMES[c["op"]], gu.dec(c["x"]))
        assert gu.scalar_same(got, gu.dec_scalar(c["out"])), (c["op"], c["x"]["dtype"], c["src"])


def test_golden_ewise(oracle):
    cs = by_fn("ewise")
    assert len(cs) > 500
    name2tag = {"bool": ck.BOOL}
    for c in cs:
        ot = name2tag[c["ot"]] if c["ot"] else None
        got = oracle.ewise(ck.OP_NAMES[c["op"]], gu.operand(c, "l"), gu.operand(c, "r"), ot=ot)
        assert gu.same_bits(got, gu.dec(c["out"])), (c["op"], c["l"]["dtype"], c["r"]["dtype"], c["src"])


def test_golden_groupby(oracle):
    cs = by_fn("groupby")
    assert len(cs) >= 9
    for c in cs:
        keys = [gu.dec(k) for k in c["keys"]]
        gb = oracle.groupby(keys)
        assert gb["ngroups"] == c["ngroups"]
        for f in ("reversemap", "counts", "first_rows", "offsets", "row_ids"):
            assert np.array_equal(gb[f], gu.dec(c[f])), (f, c["src"])
        for a in c["aggs"]:
            got = oracle.grouped_reduce(ck.RED_NAMES[a["op"]], gu.dec(a["x"]), gb)
            assert gu.same_bits(got, gu.dec(a["out"])), (a["op"], c["src"])


def test_golden_misc(oracle):
    for c in by_fn("corr"):
        assert gu.same_bits(np.array([oracle.corr(gu.dec(c["x"]), gu.dec(c["y"]))]), gu.dec(c["out"]))
    for c in by_fn("gather"):
        assert gu.same_bits(oracle.gather(gu.dec(c["x"]), gu.dec(c["idx"])), gu.dec(c["out"]))
    for c in by_fn("compact"):
        assert gu.same_bits(oracle.compact(gu.dec(c["x"]), gu.dec(c["mask"])), gu.dec(c["out"]))
    for c in by_fn("hash_scalar"):
        assert oracle.hash_scalar(np.int32(c["v"])) == int(c["out"])
    for c in by_fn("hash_tuple"):
        assert oracle.hash_tuple([np.int32(v) for v in c["v"]]) == int(c["out"])


# ---- known-answer values quoted in SURVEY.md 8c (human-readable pins) -------------------------
SALES = np.array([100, 120, 140, 140, 130], np.int32)   # data/moving_avg.csv ascending Month
SALES_FILE = np.array([100, 140, 130, 140, 120], np.int32)
PRICE = np.array([15, 19, 16, 17, 15, 13, 5, 8, 7, 13, 11, 14, 10, 5, 2, 5], np.int32)  # tests/stock.a
TS = np.arange(1, 17, dtype=np.int32)


def test_kat_moving_avg(oracle):
    s = lambda name, x, w=0: oracle.scan(ck.SCAN_NAMES[name], x, w)
    assert s("avgw", SALES, 3).tolist() == [100, 110, 120, 133.33333333333334, 136.66666666666669]
    assert ck.i128_to_int(s("sumw", SALES, 3)) == [100, 220, 360, 400, 410]
    assert s("minw", SALES, 2).tolist() == [100, 100, 120, 140, 130]
    assert s("maxw", SALES, 2).tolist() == [100, 120, 140, 140, 140]
    assert s("avgs", SALES).tolist() == [100, 110, 120, 125, 126]
    assert ck.i128_to_int(s("sums", SALES)) == [100, 220, 360, 500, 630]
    assert s("mins", SALES).tolist() == [100] * 5
    assert s("maxs", SALES).tolist() == [100, 120, 140, 140, 140]
    assert s("deltas", SALES).tolist() == [0, 20, 20, 0, -10]
    assert s("prev", SALES).tolist() == [100, 100, 120, 140, 140]
    assert s("aggnext", SALES).tolist() == [120, 140, 140, 130, 130]
    r = s("ratiow", SALES, 1)
    assert r.dtype == np.float32
    assert [float(v) for v in r] == [1, 1.2000000476837158, 1.1666666269302368, 1, 0.92857140302658081]
    assert s("avgw", SALES_FILE, 3).tolist() == [100, 120, 123.33333333333333, 136.66666666666666, 130]
    assert s("minw", SALES_FILE, 2).tolist() == [100, 100, 130, 130, 120]
    red = lambda name, x: oracle.reduce(ck.RED_NAMES[name], x)
    assert (red("sum", SALES_FILE), red("avg", SALES_FILE), red("max", SALES_FILE), red("min", SALES_FILE)) == (630, 126.0, 140, 100)


def test_kat_stock(oracle):
    red = lambda name, x: oracle.reduce(ck.RED_NAMES[name], x)
    q1 = oracle.ewise(ck.OP_SUB, PRICE, red("min", TS))
    assert red("max", q1) == 18
    q2 = oracle.ewise(ck.OP_SUB, PRICE, oracle.scan(ck.SCAN_MINS, PRICE))
    assert red("max", q2) == 9
    rev = PRICE[::-1].copy()
    assert red("max", oracle.ewise(ck.OP_SUB, rev, oracle.scan(ck.SCAN_MINS, rev))) == 17
    mask = oracle.ewise(ck.OP_GT, oracle.ewise(ck.OP_SUB, PRICE, TS), np.int32(1))
    assert "".join(str(int(v)) for v in mask) == "1111110001010000"


def test_kat_types_and_quirks(oracle):
    assert oracle.reduce_out_dtype(ck.RED_SUM, ck.INT32) == ck.INT128
    assert oracle.reduce_out_dtype(ck.RED_SUM, ck.UINT32) == ck.UINT128
    assert oracle.reduce_out_dtype(ck.RED_SUM, ck.FLOAT) == ck.DOUBLE
    assert oracle.scan_out_dtype(ck.SCAN_AVGW, ck.INT32) == ck.DOUBLE
    assert oracle.ewise_out_dtype(ck.OP_DIV, ck.INT32, ck.INT32) == ck.FLOAT
    assert oracle.ewise_out_dtype(ck.OP_MUL, ck.INT32, ck.INT32) == ck.INT128
    assert oracle.reduce(ck.RED_SUM, np.array([0.1, 0.2, 0.3], np.float32)) == 0.60000001639127731
    assert oracle.reduce(ck.RED_MAX, np.array([-1.0, -2.0])) == 2.2250738585072014e-308  # D8
    assert oracle.hash_scalar(np.int32(7)) == 6018027440424182935
    assert oracle.hash_tuple([np.int32(3), np.int32(4)]) == 11708105269577805707


def test_kat_groupby_test_csv(oracle):
    c = [c for c in by_fn("groupby") if c["src"].startswith("tests/q1.sql")][0]
    a, b, d = (gu.dec(k) for k in c["keys"])
    gb = oracle.groupby([a, b, d])
    assert gb["ngroups"] == 16
    first = gb["first_rows"]
    order = [(int(a[i]), int(b[i]), int(d[i])) for i in first]
    assert order == [(1, 1, 2), (2, 1, 2), (2, 4, 4), (1, 2, 2), (1, 2, 4), (4, 2, 4), (2, 1, 3), (3, 2, 2), (1, 2, 3),
                     (3, 3, 4), (2, 2, 1), (2, 3, 4), (2, 4, 2), (3, 4, 2), (2, 3, 2), (1, 2, 1)]
    g2 = gb["row_ids"][gb["offsets"][1]: gb["offsets"][1] + gb["counts"][1]]
    assert g2.tolist() == [12, 7, 1]  # the second group, key (2,1,2): descending row ids inside a group
    cc = gu.dec(c["aggs"][0]["x"])
    sums = ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, cc, gb))
    assert sums == [2, 7, 3, 2, 6, 1, 3, 5, 3, 4, 3, 4, 1, 1, 2, 3]
