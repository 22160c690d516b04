"""ctypes front end of the CPU checkers (test infrastructure).

`Checker` wraps a shared library exporting the checker ABI of oracle/aq_oracle.h:
  - oracle/liboracle.so        prefix aqo_  (plain-C restatement)
  - oracle/_ref/libaqref.so    prefix aqr_  (the real reference headers; only where
                                             /root/reference was mounted at build time)
Inputs and outputs are numpy arrays; 128-bit results come back as the structured
dtype I128 (lo: u8, hi: i8) -- see `i128_to_int`.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# dtype tags: include/aqg.h (reference server/aquery_types.h:1-5)
INT32, FLOAT, STR, DOUBLE, LDOUBLE, INT64, INT128, INT16, DATE, TIME, INT8 = range(11)
UINT32, UINT64, UINT128, UINT16, UINT8, BOOL = 11, 12, 13, 14, 15, 16
TIMESTAMP = 18
ERROR = 22
KEY_ELEM_BYTES = {DATE: 4, TIME: 8, TIMESTAMP: 12}          # key columns of these tags travel as (n, bytes) uint8 arrays

I128 = np.dtype([("lo", "<u8"), ("hi", "<i8")])
U128 = np.dtype([("lo", "<u8"), ("hi", "<u8")])

TAG2NP = {
    INT8: np.dtype(np.int8), INT16: np.dtype(np.int16), INT32: np.dtype(np.int32), INT64: np.dtype(np.int64),
    UINT8: np.dtype(np.uint8), UINT16: np.dtype(np.uint16), UINT32: np.dtype(np.uint32), UINT64: np.dtype(np.uint64),
    FLOAT: np.dtype(np.float32), DOUBLE: np.dtype(np.float64), BOOL: np.dtype(np.uint8),
    INT128: I128, UINT128: U128,
}
NP2TAG = {v: k for k, v in TAG2NP.items() if k not in (BOOL,)}
NP2TAG[np.dtype(np.bool_)] = BOOL

# op enums (include/aqg.h)
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_AND, OP_OR, OP_XOR, OP_GT, OP_LT, OP_GE, OP_LE, OP_EQ, OP_NE = range(14)
VEC_VEC, VEC_SCALAR, SCALAR_VEC = 0, 1, 2
UN_SQRT, UN_TRUNCATE = 0, 1
RED_SUM, RED_MIN, RED_MAX, RED_COUNT, RED_AVG, RED_VAR, RED_STDDEV, RED_FIRST, RED_LAST = range(9)
(SCAN_SUMS, SCAN_AVGS, SCAN_MINS, SCAN_MAXS, SCAN_SUMW, SCAN_AVGW, SCAN_MINW, SCAN_MAXW, SCAN_RATIOW,
 SCAN_DELTAS, SCAN_PREV, SCAN_NEXT, SCAN_VARS, SCAN_STDDEVS, SCAN_VARW, SCAN_STDDEVW) = range(16)
(GEN_ID1, GEN_ID2, GEN_ID3, GEN_ID4, GEN_ID5, GEN_ID6, GEN_V1, GEN_V2, GEN_V3, GEN_TIMESTAMP, GEN_PRICE) = range(11)
GEN_DTYPE = {GEN_V3: np.float32}

RED_NAMES = {"sum": RED_SUM, "min": RED_MIN, "max": RED_MAX, "count": RED_COUNT, "avg": RED_AVG,
             "var": RED_VAR, "stddev": RED_STDDEV, "first": RED_FIRST, "last": RED_LAST}
SCAN_NAMES = {"sums": SCAN_SUMS, "avgs": SCAN_AVGS, "mins": SCAN_MINS, "maxs": SCAN_MAXS, "sumw": SCAN_SUMW,
              "avgw": SCAN_AVGW, "minw": SCAN_MINW, "maxw": SCAN_MAXW, "ratiow": SCAN_RATIOW,
              "deltas": SCAN_DELTAS, "prev": SCAN_PREV, "aggnext": SCAN_NEXT, "vars": SCAN_VARS,
              "stddevs": SCAN_STDDEVS, "varw": SCAN_VARW, "stddevw": SCAN_STDDEVW}
OP_NAMES = {"add": OP_ADD, "sub": OP_SUB, "mul": OP_MUL, "div": OP_DIV, "mod": OP_MOD, "and": OP_AND, "or": OP_OR,
            "xor": OP_XOR, "gt": OP_GT, "lt": OP_LT, "ge": OP_GE, "le": OP_LE, "eq": OP_EQ, "ne": OP_NE}


def tag_of(a):
    return NP2TAG[np.asarray(a).dtype]


def i128_to_int(a):
    """structured I128/U128 array -> list of python ints"""
    a = np.atleast_1d(a)
    signed = a.dtype == I128
    out = []
    for lo, hi in zip(a["lo"].tolist(), a["hi"].tolist()):
        out.append((hi << 64) + lo if signed else (hi << 64) | lo)
    return out


def scalar_from16(buf, tag):
    """decode the 16-byte result slot of reduce() into a python/numpy scalar"""
    dt = TAG2NP[tag]
    v = np.frombuffer(bytes(buf), dtype=dt, count=1)[0]
    if tag in (INT128, UINT128):
        return i128_to_int(np.array([v], dtype=dt))[0]
    return v


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class CheckerError(RuntimeError):
    pass


class Checker:
    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)
        f = self._f
        f("time_groupby_sum").restype = C.c_double
        f("hash_scalar").restype = C.c_uint64
        f("hash_tuple").restype = C.c_uint64

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise CheckerError(f"{what}: status {rc}")

    # -- type rules
    def long_type(self, t): return self._f("long_type")(t)
    def fp_type(self, t): return self._f("fp_type")(t)
    def coercion(self, a, b): return self._f("coercion")(a, b)
    def ewise_out_dtype(self, op, lt, rt): return self._f("ewise_out_dtype")(op, lt, rt)
    def reduce_out_dtype(self, op, t): return self._f("reduce_out_dtype")(op, t)
    def scan_out_dtype(self, op, t): return self._f("scan_out_dtype")(op, t)

    # -- element-wise
    def ewise(self, op, l, r, ot=None):
        l_is_vec, r_is_vec = np.ndim(l) > 0, np.ndim(r) > 0
        la, ra = np.ascontiguousarray(np.atleast_1d(l)), np.ascontiguousarray(np.atleast_1d(r))
        kind = VEC_VEC if (l_is_vec and r_is_vec) else (VEC_SCALAR if l_is_vec else SCALAR_VEC)
        n = la.size if l_is_vec else ra.size
        lt, rt = tag_of(la), tag_of(ra)
        if ot is None:
            ot = self.ewise_out_dtype(op, lt, rt)
        if ot == ERROR:
            raise CheckerError("ewise: no result dtype")
        out = np.empty(n, dtype=TAG2NP[ot])
        self._chk(self._f("ewise")(op, kind, lt, _p(la), rt, _p(ra), ot, _p(out), C.c_uint32(n)), "ewise")
        return out

    def unary(self, op, x, param=0):
        x = np.ascontiguousarray(x)
        t = tag_of(x)
        ot = DOUBLE if op == UN_SQRT else t
        out = np.empty(x.size, dtype=TAG2NP[ot])
        self._chk(self._f("unary")(op, t, _p(x), C.c_uint32(x.size), C.c_uint32(param), ot, _p(out)), "unary")
        return out

    # -- reductions
    def reduce(self, op, x):
        x = np.ascontiguousarray(x)
        t = tag_of(x)
        buf = (C.c_ubyte * 16)()
        self._chk(self._f("reduce")(op, t, _p(x), C.c_uint32(x.size), buf), "reduce")
        return scalar_from16(buf, self.reduce_out_dtype(op, t))

    def corr(self, x, y):
        x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
        out = C.c_double()
        self._chk(self._f("corr")(tag_of(x), _p(x), tag_of(y), _p(y), C.c_uint32(x.size), C.byref(out)), "corr")
        return out.value

    # -- scans
    def scan(self, op, x, w=0):
        x = np.ascontiguousarray(x)
        t = tag_of(x)
        ot = self.scan_out_dtype(op, t)
        out = np.zeros(x.size, dtype=TAG2NP[ot])
        self._chk(self._f("scan")(op, t, _p(x), C.c_uint32(x.size), C.c_uint32(w), _p(out)), "scan")
        return out

    # -- gather / filter
    def gather(self, x, idx):
        x, idx = np.ascontiguousarray(x), np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.empty(idx.size, dtype=x.dtype)
        self._chk(self._f("gather")(tag_of(x), _p(x), _p(idx), C.c_uint32(idx.size), _p(out)), "gather")
        return out

    def compact(self, x, mask):
        x, mask = np.ascontiguousarray(x), np.ascontiguousarray(mask).astype(np.uint8)
        out = np.empty(x.size, dtype=x.dtype)
        m = C.c_uint32()
        self._chk(self._f("compact")(tag_of(x), _p(x), _p(mask), C.c_uint32(x.size), _p(out), C.byref(m)), "compact")
        return out[:m.value].copy()

    # -- hashing
    def hash_scalar(self, v):
        a = np.atleast_1d(np.asarray(v))
        return self._f("hash_scalar")(tag_of(a), _p(a))

    def hash_tuple(self, vals):
        arrs = [np.atleast_1d(np.asarray(v)) for v in vals]
        dts = (C.c_int * len(arrs))(*[tag_of(a) for a in arrs])
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        return self._f("hash_tuple")(len(arrs), dts, ptrs)

    # -- group by
    def groupby(self, keys, postproc=True):
        keys = [np.ascontiguousarray(k) for k in keys]
        n = keys[0].size
        dts = (C.c_int * len(keys))(*[tag_of(k) for k in keys])
        ptrs = (C.c_void_p * len(keys))(*[k.ctypes.data for k in keys])
        rev = np.empty(max(n, 1), np.uint32)
        counts = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(max(n, 1), np.uint32)
        row_ids = np.empty(max(n, 1), np.uint32)
        first = np.zeros(max(n, 1), np.uint32)
        G = C.c_uint32()
        self._chk(self._f("groupby")(len(keys), dts, ptrs, C.c_uint32(n), _p(rev), C.byref(G), _p(counts),
                                     _p(offsets) if postproc else None, _p(row_ids) if postproc else None,
                                     _p(first)), "groupby")
        g = G.value
        res = dict(ngroups=g, reversemap=rev[:n].copy(), counts=counts[:g].copy(), first_rows=first[:g].copy())
        if postproc:
            res.update(offsets=offsets[:g].copy(), row_ids=row_ids[:n].copy())
        return res

    def groupby_typed(self, cols):
        """group ids in first-occurrence order for key columns of any type the reference hashes: cols = [(tag, data)] with
        data = numpy array (FLOAT / DOUBLE / integers / I128), an (n, bytes) uint8 array (DATE / TIME / TIMESTAMP), or a list of
        bytes objects (STR: astring_view, compared by content; every row gets its own buffer)"""
        keep, ptrs, dts, n = [], [], [], None
        for tag, data in cols:
            if tag == STR:
                bufs = [C.create_string_buffer(b) for b in data]
                arr = (C.c_char_p * len(bufs))(*[C.cast(b, C.c_char_p) for b in bufs])
                keep += [bufs, arr]
                ptrs.append(C.cast(arr, C.c_void_p).value); m = len(bufs)
            else:
                a = np.ascontiguousarray(data)
                keep.append(a)
                ptrs.append(a.ctypes.data); m = a.shape[0]
            dts.append(tag)
            assert n is None or n == m
            n = m
        rev, first, G = np.empty(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32), C.c_uint32()
        self._chk(self._f("groupby_typed")(len(cols), (C.c_int * len(cols))(*dts), (C.c_void_p * len(cols))(*ptrs), C.c_uint32(n),
                                           _p(rev), C.byref(G), _p(first)), "groupby_typed")
        return dict(ngroups=G.value, reversemap=rev[:n].copy(), first_rows=first[:G.value].copy())

    def grouped_reduce(self, op, x, gb):
        x = np.ascontiguousarray(x)
        t = tag_of(x)
        ot = self.reduce_out_dtype(op, t)
        G = gb["ngroups"]
        out = np.zeros(G, dtype=TAG2NP[ot])
        self._chk(self._f("grouped_reduce")(op, t, _p(x), C.c_uint32(G), _p(gb["offsets"]), _p(gb["counts"]),
                                            _p(gb["row_ids"]), _p(out)), "grouped_reduce")
        return out

    def join_pairs(self, build, probe):
        build, probe = np.ascontiguousarray(build), np.ascontiguousarray(probe)
        m = C.c_uint64()
        self._chk(self._f("join_pairs")(tag_of(build), _p(build), C.c_uint32(build.size), _p(probe),
                                        C.c_uint32(probe.size), None, None, C.c_uint64(0), C.byref(m)), "join")
        pr, br = np.empty(m.value, np.uint32), np.empty(m.value, np.uint32)
        self._chk(self._f("join_pairs")(tag_of(build), _p(build), C.c_uint32(build.size), _p(probe),
                                        C.c_uint32(probe.size), _p(pr), _p(br), C.c_uint64(m.value), C.byref(m)), "join")
        return pr, br

    def gen_column(self, col, seed, row_base, n, n_total, K):
        out = np.empty(n, dtype=GEN_DTYPE.get(col, np.int32))
        self._chk(self._f("gen_column")(col, C.c_uint64(seed), C.c_uint64(row_base), C.c_uint32(n),
                                        C.c_uint64(n_total), C.c_uint32(K), _p(out)), "gen_column")
        return out

    def time_groupby_sum(self, keys, vals):
        keys = [np.ascontiguousarray(k) for k in keys]
        vals = [np.ascontiguousarray(v) for v in vals]
        kd = (C.c_int * len(keys))(*[tag_of(k) for k in keys])
        kp = (C.c_void_p * len(keys))(*[k.ctypes.data for k in keys])
        vd = (C.c_int * len(vals))(*[tag_of(v) for v in vals])
        vp = (C.c_void_p * len(vals))(*[v.ctypes.data for v in vals])
        G = C.c_uint32()
        split = (C.c_double * 3)()
        t = self._f("time_groupby_sum")(len(keys), kd, kp, len(vals), vd, vp, C.c_uint32(keys[0].size), C.byref(G), split)
        return t, G.value, list(split)


def oracle_path():
    return os.path.join(ROOT, "oracle", "liboracle.so")


def ref_path(fast=False):
    return os.path.join(ROOT, "oracle", "_ref", "libaqref_fast.so" if fast else "libaqref.so")


def load_oracle():
    p = oracle_path()
    if not os.path.exists(p):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return Checker(p, "aqo_")


def load_ref(fast=False):
    p = ref_path(fast)
    return Checker(p, "aqr_") if os.path.exists(p) else None
